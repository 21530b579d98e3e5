// MEASUREMENT BUILD of the HALO_PHASE kernel with parts knocked out (KO template parameter of conv_igemm_dma_kernel.h):
// NOT part of the default library: `make KO=1` compiles this file and the dispatcher's RS_HALO_KO = 1..4 branch
// (-DRS_HALO_KO_BUILD) -- the results of a knocked-out launch are wrong by construction.  scripts/halo_knockout.sh builds
// that way, times them on dec3's shape and restores the default build.
#define RS_CONV_INSTANTIATE
#include "conv_igemm_dma_kernel.h"

void rs_conv_launch_bf16_halo_phase_ko(int ko, int grid, hipStream_t s, const ConvArgsT<bf16_t>& a) {
  if (ko == 1) conv_igemm_dma<bf16_t, 256, 128, 4, 2, 128, true, EPI_EVAL, HALO_PHASE, 1><<<grid, 512, 0, s>>>(a);
  else if (ko == 2) conv_igemm_dma<bf16_t, 256, 128, 4, 2, 128, true, EPI_EVAL, HALO_PHASE, 2><<<grid, 512, 0, s>>>(a);
  else if (ko == 3) conv_igemm_dma<bf16_t, 256, 128, 4, 2, 128, true, EPI_EVAL, HALO_PHASE, 3><<<grid, 512, 0, s>>>(a);
  else conv_igemm_dma<bf16_t, 256, 128, 4, 2, 128, true, EPI_EVAL, HALO_PHASE, 4><<<grid, 512, 0, s>>>(a);
}

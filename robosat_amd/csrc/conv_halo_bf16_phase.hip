// Instantiations of the HALO_PHASE form of the LDS-DMA convolution kernel (conv_igemm_dma_kernel.h): bf16 DecoderBlock
// (reference robosat/unet.py:63-73) in phase form, one output parity per block, the 2x2 taps of the parity read from ONE
// source halo per channel chunk.  `tile` = BN (128 | 64).
#define RS_CONV_INSTANTIATE
#include "conv_igemm_dma_kernel.h"

void rs_conv_launch_bf16_halo_phase(int tile, int rowb, int grid, hipStream_t s, const ConvArgsT<bf16_t>& a) {
  launch_halo<HALO_PHASE, true, EPI_EVAL>(tile, grid, s, a);
}

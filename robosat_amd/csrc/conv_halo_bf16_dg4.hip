// Instantiations of the HALO_DG4 form of the LDS-DMA convolution kernel (conv_igemm_dma_kernel.h): the bf16 data gradient
// of DecoderBlock wrt its pre-upsample inputs (loss.backward() through robosat/unet.py:63-73) -- a 4x4 / stride-2
// convolution over dz -- as four 2x2 convolutions over dz's parity planes, each plane's halo in LDS once per channel chunk.
// `tile` = BN (128 | 64).
#define RS_CONV_INSTANTIATE
#include "conv_igemm_dma_kernel.h"

void rs_conv_launch_bf16_halo_dg4(int tile, int rowb, int grid, hipStream_t s, const ConvArgsT<bf16_t>& a) {
  launch_halo<HALO_DG4, false, EPI_EVAL>(tile, grid, s, a);
}

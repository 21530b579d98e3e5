// Instantiations of the HALO_33 form of the LDS-DMA convolution kernel (conv_igemm_dma_kernel.h): bf16, 3x3 / stride 1 /
// pad 1 with the patch's source halo in LDS once per channel chunk -- Bottleneck.conv2 of layer1-3 (reference
// robosat/unet.py:127-130 via torchvision's Bottleneck) forward with the three epilogue kinds (eval, train-mode statistics)
// and its data gradient into the preceding BatchNorm (EPI_BWD).  `tile` = BN (128 | 64), `rowb` = the epilogue kind.
#define RS_CONV_INSTANTIATE
#include "conv_igemm_dma_kernel.h"

void rs_conv_launch_bf16_halo33(int tile, int rowb, int grid, hipStream_t s, const ConvArgsT<bf16_t>& a) {
  if (rowb == EPI_STATS) launch_halo<HALO_33, false, EPI_STATS>(tile, grid, s, a);
  else if (rowb == EPI_BWD) launch_halo<HALO_33, false, EPI_BWD>(tile, grid, s, a);
  else launch_halo<HALO_33, false, EPI_EVAL>(tile, grid, s, a);
}

// Instantiations of the LDS-DMA implicit-GEMM convolution kernel (conv_igemm_dma_kernel.h; design notes in
// conv_igemm_dma.hip): bf16 activations, direct form, epilogue EPI_EVAL -- every tile and K-chunk row size.
#define RS_CONV_INSTANTIATE
#include "conv_igemm_dma_kernel.h"

RS_CONV_DEFINE_LAUNCHER(rs_conv_launch_bf16_plain_eval, bf16_t, false, EPI_EVAL)

// conv_wgrad_f32.h -- what conv_wgrad.hip (plan, register-staged kernel, stem) and conv_wgrad_f32_dma.hip (the LDS-DMA kernel)
// share: the launch arguments and the tile variants.
#pragma once
#include "common.h"

struct WgradArgs {
  const float* dy;
  const float* src1;
  const float* src2;
  float* out;  // [splits][Cout][K]  (K = taps * Cin, or kh*32 for the stem)
  int N, Hs, Ws, C1, C2, Hv, Wv, ups;
  int kw, stride, pad, Ho, Wo, Cout;
  int M, K, tiles_co, tiles_ci, tiles_k, chunks_per_split;
  rs_fastdiv div_howo, div_wo;  // PHASE: of the source grid (Hs*Ws, Ws): the rows m enumerate source pixels
};

enum { V128x128 = 0, V128x64, V64x128, V64x64, V32x128, V32x32, VSTEM };

// conv_wgrad_f32_dma.hip: the same blocks (tile variant, tap / phase combination, split of 32-pixel chunks) with both operands
// copied HBM -> LDS by LDS-DMA as they lie (pixel-major) and read back one dword per MFMA operand.  Never the stem.
__attribute__((visibility("hidden"))) int rs_wgrad_f32_dma_launch(int variant, bool phase, int grid, hipStream_t s, const WgradArgs& a);

// conv_wgrad_wino_f32.hip (round 6): DecoderBlock's weight gradient in the Winograd domain of the forward's F(2x2, 2x2) form
// (9/16 of the phase form's multiply-adds); knob wgrad_f32_wino.  Workspace: splits x 36 x Cout x Cin floats.
__attribute__((visibility("hidden"))) bool rs_wgrad_f32_wino_ok(const rs_conv_desc* d);
__attribute__((visibility("hidden"))) long rs_wgrad_f32_wino_workspace_floats(const rs_conv_desc* d);
__attribute__((visibility("hidden"))) int rs_wgrad_f32_wino_launch(const rs_conv_desc* d, const float* dz, const float* src1, const float* src2,
                                                                   float* dw, float* workspace, hipStream_t s);

// conv_wgrad_wino33_f32.hip (round 6): stride-1 3x3 / pad-1 convolutions in the Winograd domain of F(2x2, 3x3) (16/36 of the
// multiply-adds); knob wgrad_f32_wino33.  Workspace: splits x 16 x Cout x Cin floats + the reduction's.
__attribute__((visibility("hidden"))) bool rs_wgrad_f32_wino33_ok(const rs_conv_desc* d);
__attribute__((visibility("hidden"))) long rs_wgrad_f32_wino33_workspace_floats(const rs_conv_desc* d);
__attribute__((visibility("hidden"))) int rs_wgrad_f32_wino33_launch(const rs_conv_desc* d, const float* dy, const float* src, float* dw,
                                                                     float* workspace, hipStream_t s);

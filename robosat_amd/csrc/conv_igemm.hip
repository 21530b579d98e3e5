// rs_conv2d_fwd: the fp32 entry point of every nn.Conv2d / F.interpolate / torch.cat call of UNet.forward (reference
// robosat/unet.py:122-141) and, through the `ups = 2` gather, of the data-gradient convolutions of loss.backward()
// (tools/train.py:186).  Dispatch only: every non-stem launch runs the LDS-DMA implicit-GEMM kernel (conv_igemm_dma.hip, fp32
// instantiation), the 7x7 / stride-2 stem its own kernel (stem_f32.hip).  (Until round 6 this file held the register-staged
// implicit-GEMM kernel the stem was the last user of: K walked as 7 rows x 32 with 147 of 224 entries real.)
#include "common.h"

// stem_f32.hip
__attribute__((visibility("hidden"))) int rs_stem_f32_launch(const rs_conv_desc* d, int bands, const float* x, const float* w,
                                                             const float* scale, const float* shift, float* out, void* stream);

namespace {

__global__ void pack_stem_weight_kernel(const float* __restrict__ w, float* __restrict__ out, int Cout, int kh, int kw,
                                        int Cin) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  const int total = Cout * kh * 32;
  if (idx >= total) return;
  const int c = idx & 3, s = (idx >> 2) & 7, r = (idx >> 5) % kh, co = (idx >> 5) / kh;
  float v = 0.f;
  if (s < kw && c < Cin) v = w[((co * kh + r) * kw + s) * Cin + c];
  out[idx] = v;
}

enum Tile { T128x128 = 0, T128x64, T128x32, T64x64, TSTEM, T256x128, NTILES };
// (report names by tile index; 6..8 are bf16-only kernels of conv_igemm_dma.hip's table, 9 = conv1x1_ew_f32.hip: the fp32 1x1
// kernel with the epilogue on its own waves)
constexpr int kNamedTiles = 10;
const char* const kTileNames[kNamedTiles] = {"conv_igemm_f32<128x128>", "conv_igemm_f32<128x64>", "conv_igemm_f32<128x32>",
                                             "conv_igemm_f32<64x64>", "stem_conv_f32<128x64>", "conv_igemm_f32<256x128>",
                                             "", "", "", "conv1x1_ew_f32<128x64>"};

bool valid(const rs_conv_desc* d) {
  if (!d) return false;
  if (d->N <= 0 || d->Hs <= 0 || d->Ws <= 0 || d->Ho <= 0 || d->Wo <= 0) return false;
  if (d->kh <= 0 || d->kw <= 0 || d->stride <= 0 || d->pad < 0) return false;
  if (d->Cout <= 0 || (d->Cout % 32) != 0) return false;
  if (d->ups < 0 || d->ups > 2) return false;
  if (d->stem) {
    if (d->C1 != 4 || d->C2 != 0 || d->kw > 8 || d->ups != 0 || (d->Cout % 64) != 0) return false;
  } else {
    if (d->C1 <= 0 || (d->C1 % 32) != 0 || d->C2 < 0 || (d->C2 % 32) != 0) return false;
  }
  return true;
}

}  // namespace

extern "C" int rs_conv2d_tile(const rs_conv_desc* d) {
  if (!valid(d)) return RS_EINVAL;
  return d->stem ? (int)TSTEM : rs_conv_dma_tile(d);  // same index order as kTileNames
}

extern "C" const char* rs_conv2d_tile_name(int tile) { return (tile >= 0 && tile < kNamedTiles) ? kTileNames[tile] : ""; }

extern "C" int rs_conv2d_fwd(const rs_conv_desc* d, const float* src1, const float* src2, const float* weight,
                             const float* scale, const float* shift, const float* residual, const float* relu_mask,
                             float* out, rs_stream_t stream) {
  if (!valid(d) || !src1 || !weight || !out) return RS_EINVAL;
  if (d->C2 > 0 && !src2) return RS_EINVAL;
  if (!d->stem) return rs_conv_dma_f32(d, src1, src2, weight, scale, shift, residual, relu_mask, out, stream);
  if (residual || relu_mask) return RS_EINVAL;  // (nothing in the network adds to or masks the stem's output)
  // stem = 3: the caller vouches that the input's 4th band and the filter's c = 3 entries are zeros (RGB): they are skipped
  return rs_stem_f32_launch(d, d->stem == 3 ? 3 : 4, src1, weight, scale, shift, out, stream);
}

extern "C" int rs_pack_stem_weight(const float* w_krsc, float* packed, int Cout, int kh, int kw, int Cin,
                                   rs_stream_t stream) {
  if (!w_krsc || !packed || Cout <= 0 || kh <= 0 || kw <= 0 || kw > 8 || Cin <= 0 || Cin > 4) return RS_EINVAL;
  const int total = Cout * kh * 32;
  pack_stem_weight_kernel<<<rs_cdiv(total, 256), 256, 0, (hipStream_t)stream>>>(w_krsc, packed, Cout, kh, kw, Cin);
  return RS_LAUNCH_RESULT();
}

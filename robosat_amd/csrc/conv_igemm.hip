// Implicit-GEMM convolution for gfx950 on v_mfma_f32_32x32x2_f32 (exact fp32, 157 TFLOP/s chip peak).
//
// Replaces every nn.Conv2d / F.interpolate / torch.cat call of UNet.forward (reference robosat/unet.py:122-141)
// and, through the `ups = 2` gather, every data-gradient convolution of loss.backward() (tools/train.py:186).
//
// GEMM view:  M = N*Ho*Wo output pixels,  N = Cout,  K = kh*kw*Cin walked in 32-wide chunks
// (one filter tap x 32 consecutive input channels: 128 contiguous bytes per output pixel in NHWC).
//
//   block  = 256 threads = 4 waves, tile BM x BN, K-chunk 32, LDS double buffered, one barrier per chunk;
//            the next chunk is fetched into registers while the MFMAs of the current one run.
//   gather = per output pixel: (n, oy*stride-pad, ox*stride-pad) is decoded once; per chunk only the tap offset
//            is added.  `ups = 1` reads the source at (y>>1, x>>1) (nearest x2, DecoderBlock, unet.py:73);
//            `ups = 2` additionally zeroes odd coordinates (zero-insertion: adjoint of a stride-2 conv);
//            channels [0,C1) come from src1 and [C1,C1+C2) from src2 (torch.cat, unet.py:134-137).
//   LDS    = rows of 32 floats padded to 36: ds_write_b128 (8 lanes = one 128-B row) and ds_read_b128
//            (16-lane groups hit 16 distinct 4-bank slots because 36*r/4 = 9r is a bijection mod 16) are
//            conflict free.
//   MFMA   = lane l feeds A[i = l&31][k = l>>5], B[k = l>>5][j = l&31].  Each lane reads 4 consecutive k per
//            operand with one b128 (k = 8j + 4*(l>>5) + t) and issues 4 MFMAs; A and B use the same k
//            permutation, so every k of the chunk is consumed exactly once.
//            The WEIGHT fragment is the A operand, so D[i][j] has i = cout, j = pixel and a lane's registers
//            4g..4g+3 hold 4 consecutive couts of one pixel.
//   store  = the accumulators are staged through the (now idle) LDS as [pixel][cout] with ds_write_b128, then
//            read back row-wise so every global access of the epilogue (output store, residual / ReLU-mask read)
//            is a 16-byte access on consecutive couts: whole 128-B lines per wave instruction.  Epilogue:
//            per-cout scale/shift (eval BatchNorm), residual add, ReLU, ReLU-mask (backward).
//
// The 7x7/2 stem (Cin = 3, padded to NHWC4) uses the same kernel with STEM = 1: a chunk is one filter ROW,
// 8 taps x 4 channels = 32 contiguous floats, per-tap bounds checks, weights packed [Cout][7][8][4].
#include "common.h"

namespace {

struct ConvArgs {
  const float* src1;
  const float* src2;
  const float* wgt;
  const float* scale;
  const float* shift;
  const float* res;
  const float* mask;
  float* out;
  int Hs, Ws, C1, C2, Hv, Wv, ups;
  int kw, stride, pad, Ho, Wo, Cout;
  int M, cpt, nk, Kw, relu, ntiles;
};

constexpr int LDK = 36;  // padded LDS row (floats)

template <int BM, int BN, int WGM, int WGN, int STEM>
__global__ __launch_bounds__(256, 2) void conv_igemm_f32(const ConvArgs p) {
  static_assert(WGM * WGN == 4, "4 waves per block");
  constexpr int WM = BM / WGM, WN = BN / WGN;
  constexpr int TM = WM / 32, TN = WN / 32;
  constexpr int AR = BM / 32, BR = BN / 32;  // float4 loads per thread per chunk
  constexpr int BUF = (BM + BN) * LDK;
  static_assert(TM >= 1 && TN >= 1, "wave tile must hold one 32x32 MFMA tile");

  __shared__ __attribute__((aligned(16))) float lds[2 * BUF];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int wm = wave / WGN, wn = wave % WGN;

  const int bid = rs_xcd_remap(blockIdx.x, gridDim.x);
  const int mt = bid / p.ntiles, nt = bid - mt * p.ntiles;
  const int m0 = mt * BM, n0 = nt * BN;

  const int lrow = tid >> 3;  // 0..31: row within a 32-row slab
  const int c4 = tid & 7;     // which float4 of the 32-float chunk row

  // ---- decode this thread's output pixels once -------------------------------------------------------------
  int rn[AR], ry[AR], rx[AR];
  const int HoWo = p.Ho * p.Wo;
#pragma unroll
  for (int i = 0; i < AR; ++i) {
    const int m = m0 + lrow + 32 * i;
    if (m < p.M) {
      const int n = m / HoWo;
      const int rem = m - n * HoWo;
      const int oy = rem / p.Wo;
      const int ox = rem - oy * p.Wo;
      rn[i] = n;
      ry[i] = oy * p.stride - p.pad;
      rx[i] = ox * p.stride - p.pad;
    } else {
      rn[i] = -1;
      ry[i] = 0;
      rx[i] = 0;
    }
  }

  f32x4 ra[AR], rb[BR];
  int lr = 0, ls = 0, lc = 0, lk = 0;  // next chunk to fetch: tap row / tap col / channel chunk / linear index

  auto load_chunk = [&]() __attribute__((always_inline)) {
    const float* src;
    int Cs, cs;
    if (STEM) {
      src = p.src1;
      Cs = 4;
      cs = 0;
    } else {
      const int c0 = lc * 32;
      if (c0 < p.C1) {
        src = p.src1;
        Cs = p.C1;
        cs = c0 + c4 * 4;
      } else {
        src = p.src2;
        Cs = p.C2;
        cs = c0 - p.C1 + c4 * 4;
      }
    }
#pragma unroll
    for (int i = 0; i < AR; ++i) {
      const int iy = ry[i] + lr;
      const int ix = rx[i] + (STEM ? c4 : ls);
      bool ok = (rn[i] >= 0) && ((unsigned)iy < (unsigned)p.Hv) && ((unsigned)ix < (unsigned)p.Wv);
      int sy = iy, sx = ix;
      if (p.ups) {
        if (p.ups == 2) ok = ok && (((iy | ix) & 1) == 0);
        sy = iy >> 1;
        sx = ix >> 1;
      }
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if (ok) {
        const long off = (((long)rn[i] * p.Hs + sy) * p.Ws + sx) * Cs + cs;
        v = *reinterpret_cast<const f32x4*>(src + off);
      }
      ra[i] = v;
    }
#pragma unroll
    for (int i = 0; i < BR; ++i) {
      const int row = n0 + lrow + 32 * i;
      rb[i] = *reinterpret_cast<const f32x4*>(p.wgt + (long)row * p.Kw + lk * 32 + c4 * 4);
    }
    ++lk;
    if (STEM) {
      ++lr;
    } else if (++lc == p.cpt) {
      lc = 0;
      if (++ls == p.kw) {
        ls = 0;
        ++lr;
      }
    }
  };

  auto store_chunk = [&](int buf) __attribute__((always_inline)) {
    float* L = lds + buf * BUF;
#pragma unroll
    for (int i = 0; i < AR; ++i) *reinterpret_cast<f32x4*>(&L[(lrow + 32 * i) * LDK + c4 * 4]) = ra[i];
#pragma unroll
    for (int i = 0; i < BR; ++i) *reinterpret_cast<f32x4*>(&L[(BM + lrow + 32 * i) * LDK + c4 * 4]) = rb[i];
  };

  f32x16 acc[TN][TM];  // [cout sub-tile][pixel sub-tile]; D rows = couts, D cols = pixels
#pragma unroll
  for (int a = 0; a < TN; ++a)
#pragma unroll
    for (int b = 0; b < TM; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

  const int arow = wm * WM + (lane & 31);
  const int brow = BM + wn * WN + (lane & 31);
  const int kq = (lane >> 5) * 4;

  auto compute = [&](int buf) __attribute__((always_inline)) {
    const float* L = lds + buf * BUF;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      f32x4 a[TM], b[TN];
#pragma unroll
      for (int tm = 0; tm < TM; ++tm) a[tm] = *reinterpret_cast<const f32x4*>(&L[(arow + 32 * tm) * LDK + 8 * j + kq]);
#pragma unroll
      for (int tn = 0; tn < TN; ++tn) b[tn] = *reinterpret_cast<const f32x4*>(&L[(brow + 32 * tn) * LDK + 8 * j + kq]);
#pragma unroll
      for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int tm = 0; tm < TM; ++tm)
#pragma unroll
          for (int tn = 0; tn < TN; ++tn)
            acc[tn][tm] = __builtin_amdgcn_mfma_f32_32x32x2f32(b[tn][t], a[tm][t], acc[tn][tm], 0, 0, 0);
    }
  };

  // ---- main loop: fetch(k+1) -> regs | MFMA(k) from LDS | regs -> LDS(other buffer) | barrier --------------
  load_chunk();
  store_chunk(0);
  __syncthreads();
  for (int kc = 0; kc < p.nk; ++kc) {
    const bool more = (kc + 1) < p.nk;
    if (more) load_chunk();
    compute(kc & 1);
    if (more) store_chunk((kc + 1) & 1);
    __syncthreads();
  }

  // ---- epilogue: registers -> LDS [pixel][cout] -> 16-byte row-wise global accesses ----------------------------
  // (the loop's trailing barrier guarantees nobody still reads the pipeline buffers)
  constexpr int LDO = BN + 4;  // row stride: 8 consecutive pixel rows land on 8 distinct 4-bank groups
  static_assert(BM * LDO <= 2 * BUF, "staging tile must fit in the pipeline buffers");
  {
    const int prow = wm * WM + (lane & 31);
    const int ccol = wn * WN + 4 * (lane >> 5);
#pragma unroll
    for (int tn = 0; tn < TN; ++tn)
#pragma unroll
      for (int tm = 0; tm < TM; ++tm)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          f32x4 v;
          v[0] = acc[tn][tm][4 * g + 0];
          v[1] = acc[tn][tm][4 * g + 1];
          v[2] = acc[tn][tm][4 * g + 2];
          v[3] = acc[tn][tm][4 * g + 3];
          *reinterpret_cast<f32x4*>(&lds[(prow + 32 * tm) * LDO + ccol + 32 * tn + 8 * g]) = v;
        }
  }
  __syncthreads();
  {
    constexpr int CPR = BN / 4;                // float4 chunks per row
    constexpr int RPI = 256 / CPR;             // rows per iteration
    const int cc = tid % CPR, rr = tid / CPR;  // CPR is a power of two <= 32
    const int col = n0 + cc * 4;
    f32x4 sc = {1.f, 1.f, 1.f, 1.f}, sh = {0.f, 0.f, 0.f, 0.f};
    if (p.scale) sc = *reinterpret_cast<const f32x4*>(p.scale + col);
    if (p.shift) sh = *reinterpret_cast<const f32x4*>(p.shift + col);
#pragma unroll 4
    for (int row = rr; row < BM; row += RPI) {
      const int m = m0 + row;
      if (m >= p.M) break;
      const long o = (long)m * p.Cout + col;
      f32x4 v = *reinterpret_cast<const f32x4*>(&lds[row * LDO + cc * 4]);
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] = v[e] * sc[e] + sh[e];
      if (p.res) {
        const f32x4 r = *reinterpret_cast<const f32x4*>(p.res + o);
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] += r[e];
      }
      if (p.relu) {
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], 0.f);
      }
      if (p.mask) {
        const f32x4 z = *reinterpret_cast<const f32x4*>(p.mask + o);
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = z[e] > 0.f ? v[e] : 0.f;
      }
      *reinterpret_cast<f32x4*>(p.out + o) = v;
    }
  }
}

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

enum Tile { T128x128 = 0, T128x64, T128x32, T64x64, TSTEM, NTILES };
const char* const kTileNames[NTILES] = {"conv_igemm_f32<128x128>", "conv_igemm_f32<128x64>", "conv_igemm_f32<128x32>",
                                        "conv_igemm_f32<64x64>", "conv_igemm_f32<128x64,stem>"};
const int kTileBM[NTILES] = {128, 128, 128, 64, 128};
const int kTileBN[NTILES] = {128, 64, 32, 64, 64};

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

int pick_tile(const rs_conv_desc* d) {
  if (d->stem) return TSTEM;
  const long M = (long)d->N * d->Ho * d->Wo;
  // want >= 2 blocks per CU (512) so one block's barrier/fetch stalls hide under another's MFMAs; fall back to
  // smaller tiles for the small-M / large-K layers (layer4, center, dec0) rather than leaving CUs idle
  const long want = 512;
  if (d->Cout % 128 == 0 && (long)rs_cdiv(M, 128) * (d->Cout / 128) >= want) return T128x128;
  if (d->Cout % 64 == 0) {
    if ((long)rs_cdiv(M, 128) * (d->Cout / 64) >= want) return T128x64;
    return T64x64;
  }
  return T128x32;
}

}  // namespace

extern "C" int rs_conv2d_tile(const rs_conv_desc* d) { return valid(d) ? pick_tile(d) : RS_EINVAL; }

extern "C" const char* rs_conv2d_tile_name(int tile) { return (tile >= 0 && tile < NTILES) ? kTileNames[tile] : ""; }

extern "C" int rs_conv2d_fwd(const rs_conv_desc* d, const float* src1, const float* src2, const float* weight,
                             const float* scale, const float* shift, const float* residual, const float* relu_mask,
                             float* out, rs_stream_t stream) {
  if (!valid(d) || !src1 || !weight || !out) return RS_EINVAL;
  if (d->C2 > 0 && !src2) return RS_EINVAL;
  ConvArgs a;
  a.src1 = src1;
  a.src2 = src2;
  a.wgt = weight;
  a.scale = scale;
  a.shift = shift;
  a.res = residual;
  a.mask = relu_mask;
  a.out = out;
  a.Hs = d->Hs;
  a.Ws = d->Ws;
  a.C1 = d->C1;
  a.C2 = d->C2;
  a.ups = d->ups;
  a.Hv = d->ups == 0 ? d->Hs : (d->ups == 1 ? 2 * d->Hs : 2 * d->Hs - 1);
  a.Wv = d->ups == 0 ? d->Ws : (d->ups == 1 ? 2 * d->Ws : 2 * d->Ws - 1);
  a.kw = d->kw;
  a.stride = d->stride;
  a.pad = d->pad;
  a.Ho = d->Ho;
  a.Wo = d->Wo;
  a.Cout = d->Cout;
  const long M = (long)d->N * d->Ho * d->Wo;
  if (M >= (1L << 31)) return RS_EINVAL;
  a.M = (int)M;
  a.cpt = d->stem ? 1 : (d->C1 + d->C2) / 32;
  a.nk = d->stem ? d->kh : d->kh * d->kw * a.cpt;
  a.Kw = a.nk * 32;
  a.relu = d->relu;

  const int tile = pick_tile(d);
  a.ntiles = d->Cout / kTileBN[tile];
  const int grid = rs_cdiv(M, kTileBM[tile]) * a.ntiles;
  hipStream_t s = (hipStream_t)stream;
  switch (tile) {
    case T128x128: conv_igemm_f32<128, 128, 2, 2, 0><<<grid, 256, 0, s>>>(a); break;
    case T128x64: conv_igemm_f32<128, 64, 2, 2, 0><<<grid, 256, 0, s>>>(a); break;
    case T128x32: conv_igemm_f32<128, 32, 4, 1, 0><<<grid, 256, 0, s>>>(a); break;
    case T64x64: conv_igemm_f32<64, 64, 2, 2, 0><<<grid, 256, 0, s>>>(a); break;
    case TSTEM: conv_igemm_f32<128, 64, 2, 2, 1><<<grid, 256, 0, s>>>(a); break;
    default: return RS_EINVAL;
  }
  return RS_LAUNCH_RESULT();
}

extern "C" int rs_pack_stem_weight(const float* w_krsc, float* packed, int Cout, int kh, int kw, int Cin,
                                   rs_stream_t stream) {
  if (!w_krsc || !packed || Cout <= 0 || kh <= 0 || kw <= 0 || kw > 8 || Cin <= 0 || Cin > 4) return RS_EINVAL;
  const int total = Cout * kh * 32;
  pack_stem_weight_kernel<<<rs_cdiv(total, 256), 256, 0, (hipStream_t)stream>>>(w_krsc, packed, Cout, kh, kw, Cin);
  return RS_LAUNCH_RESULT();
}

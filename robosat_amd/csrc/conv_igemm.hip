// Implicit-GEMM convolution for gfx950 on v_mfma_f32_32x32x2_f32 (exact fp32, 157 TFLOP/s chip peak).
//
// Replaces every nn.Conv2d / F.interpolate / torch.cat call of UNet.forward (reference robosat/unet.py:122-141)
// and, through the `ups = 2` gather, every data-gradient convolution of loss.backward() (tools/train.py:186).
//
// GEMM view:  M = N*Ho*Wo output pixels,  N = Cout,  K = kh*kw*Cin walked in 32-wide chunks
// (one filter tap x 32 consecutive input channels: 128 contiguous bytes per output pixel in NHWC).
//
//   block  = 256 threads = 4 waves, tile BM x BN, K-chunk 32, LDS double buffered, one barrier per chunk;
//            the next chunk is fetched into registers while the MFMAs of the current one run.  The loop body is
//            ONE basic block: loads are buffer_load_dwordx4 through SRSRC descriptors whose hardware bounds
//            check returns zeros for padding / tail rows (offset = -1), so there is no branch around any load and
//            the compiler interleaves the address arithmetic + load issue of chunk k+1 with the MFMAs of chunk
//            k (PMC on the first version showed both waves of a SIMD doing their ~1250-cycle load phase at the
//            same time: MFMA busy 74 %).  The prefetch of the non-existent chunk nk is harmless (zeros / in-range).
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
  int N, Hs, Ws, C1, C2, Hv, Wv, ups;
  int kw, stride, pad, Ho, Wo, Cout;
  int M, cpt, nk, Kw, relu, ntiles;
};

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

// raw buffer descriptor (stride 0): loads at byte offsets >= bytes return 0
__device__ __forceinline__ __amdgpu_buffer_rsrc_t rs_make_rsrc(const float* base, long bytes) {
  const unsigned int n = bytes > 0xFFFFFFFEL ? 0xFFFFFFFEu : (unsigned int)(bytes < 0 ? 0 : bytes);
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(base), 0, (int)n, 0x00020000);
}

__device__ __forceinline__ f32x4 rs_buffer_load4(__amdgpu_buffer_rsrc_t r, int byte_off) {
  return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, byte_off, 0, 0));
}

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

  // ---- decode this thread's output pixels once; offsets are relative to the tile's first image so that 32-bit
  //      byte offsets are enough (validated on the host) -------------------------------------------------------
  const int HoWo = p.Ho * p.Wo;
  const int nfirst = m0 / HoWo;
  int rbase[AR], ry[AR], rx[AR];  // (n - nfirst) * Hs  (or -1 for rows past M), top-left input coordinates
#pragma unroll
  for (int i = 0; i < AR; ++i) {
    const int m = m0 + lrow + 32 * i;
    if (m < p.M) {
      const int n = m / HoWo;
      const int rem = m - n * HoWo;
      const int oy = rem / p.Wo;
      const int ox = rem - oy * p.Wo;
      rbase[i] = (n - nfirst) * p.Hs;
      ry[i] = oy * p.stride - p.pad;
      rx[i] = ox * p.stride - p.pad;
    } else {
      rbase[i] = 0;
      ry[i] = -64;  // taps add at most kh-1 <= 14: never inside [0, Hv)
      rx[i] = 0;
    }
  }
  const long img1 = (long)p.Hs * p.Ws * (STEM ? 4 : p.C1);
  const long img2 = (long)p.Hs * p.Ws * p.C2;
  const __amdgpu_buffer_rsrc_t rsrc1 = rs_make_rsrc(p.src1 + nfirst * img1, (long)(p.N - nfirst) * img1 * 4);
  const __amdgpu_buffer_rsrc_t rsrc2 = rs_make_rsrc(p.C2 ? p.src2 + nfirst * img2 : p.src1, (long)(p.N - nfirst) * img2 * 4);
  const __amdgpu_buffer_rsrc_t rsrcw = rs_make_rsrc(p.wgt, (long)p.Cout * p.Kw * 4);
  int woff[BR];
#pragma unroll
  for (int i = 0; i < BR; ++i) woff[i] = ((n0 + lrow + 32 * i) * p.Kw + c4 * 4) * 4;
  const int ush = p.ups ? 1 : 0;       // nearest / zero-insert x2: source coordinate = virtual >> 1
  const int upar = p.ups == 2 ? 1 : 0;  // zero-insert: odd virtual coordinates are zeros

  f32x4 ra[AR], rb[BR];
  int lr = 0, ls = 0, lc = 0, lk = 0;  // next chunk to fetch: tap row / tap col / channel chunk / linear index

  // issues a third of the loads of the NEXT chunk (parts 0..2; the last MFMA quarter stays load-free so that the
  // latency of the final loads is covered before the LDS store needs them); branch-free
  auto load_part = [&](int part) __attribute__((always_inline)) {
    const int c0 = lc * 32;
    const bool first = STEM || (c0 < p.C1);
    const __amdgpu_buffer_rsrc_t rs = first ? rsrc1 : rsrc2;
    const int Cs = STEM ? 4 : (first ? p.C1 : p.C2);
    const int cs = STEM ? 0 : ((first ? c0 : c0 - p.C1) + c4 * 4);
#pragma unroll
    for (int i = 0; i < AR; ++i) {
      if ((i % 3) != part) continue;
      const int iy = ry[i] + lr;
      const int ix = rx[i] + (STEM ? c4 : ls);
      bool ok = ((unsigned)iy < (unsigned)p.Hv) && ((unsigned)ix < (unsigned)p.Wv);
      ok = ok && ((((iy | ix) & upar)) == 0);
      const int pix = (rbase[i] + (iy >> ush)) * p.Ws + (ix >> ush);
      const int off = ok ? (pix * Cs + cs) * 4 : -1;
      ra[i] = rs_buffer_load4(rs, off);
    }
#pragma unroll
    for (int i = 0; i < BR; ++i) {
      if (((i + 1) % 3) != part) continue;
      rb[i] = rs_buffer_load4(rsrcw, woff[i] + lk * 128);
    }
  };

  auto advance = [&]() __attribute__((always_inline)) {
    ++lk;
    if (STEM) {
      ++lr;
    } else {
      ++lc;
      const int w1 = (lc == p.cpt) ? 1 : 0;
      lc = w1 ? 0 : lc;
      ls += w1;
      const int w2 = (ls == p.kw) ? 1 : 0;
      ls = w2 ? 0 : ls;
      lr += w2;
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

  // one quarter of a chunk = 8 of the 32 k: fragments via one ds_read_b128 per 32-row sub-tile, TM*TN*4 MFMAs
  auto read_frag = [&](const float* L, int j, f32x4 (&a)[TM], f32x4 (&b)[TN]) __attribute__((always_inline)) {
#pragma unroll
    for (int tm = 0; tm < TM; ++tm) a[tm] = *reinterpret_cast<const f32x4*>(&L[(arow + 32 * tm) * LDK + 8 * j + kq]);
#pragma unroll
    for (int tn = 0; tn < TN; ++tn) b[tn] = *reinterpret_cast<const f32x4*>(&L[(brow + 32 * tn) * LDK + 8 * j + kq]);
  };
  auto mma_frag = [&](const f32x4 (&a)[TM], const f32x4 (&b)[TN]) __attribute__((always_inline)) {
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
      for (int tm = 0; tm < TM; ++tm)
#pragma unroll
        for (int tn = 0; tn < TN; ++tn)
          acc[tn][tm] = __builtin_amdgcn_mfma_f32_32x32x2f32(b[tn][t], a[tm][t], acc[tn][tm], 0, 0, 0);
  };

  // ---- main loop.  Per quarter j: { prefetch the LDS fragments of quarter j+1 | address math + buffer loads of a
  //      third of chunk k+1 | 16 MFMAs of quarter j }; then regs -> LDS(other buffer) | barrier.  No branch inside;
  //      sched_barrier keeps each quarter's loads in that quarter (the scheduler otherwise sinks every load to the
  //      end of the chunk, right before its use) and the last quarter load-free (latency cover for the LDS store).
#pragma unroll
  for (int part = 0; part < 3; ++part) load_part(part);
  advance();
  store_chunk(0);
  __syncthreads();
  for (int kc = 0; kc < p.nk; ++kc) {
    const float* L = lds + (kc & 1) * BUF;
    f32x4 fa[2][TM], fb[2][TN];
    read_frag(L, 0, fa[0], fb[0]);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      if (j < 3) {
        read_frag(L, j + 1, fa[(j + 1) & 1], fb[(j + 1) & 1]);
        load_part(j);
      }
      mma_frag(fa[j & 1], fb[j & 1]);
      __builtin_amdgcn_sched_barrier(0);
    }
    advance();
    store_chunk((kc + 1) & 1);
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

enum Tile { T128x128 = 0, T128x64, T128x32, T64x64, TSTEM, T256x128, NTILES };
// (report names by tile index; 6..8 are bf16-only kernels of conv_igemm_dma.hip's table, 9 = conv1x1_ew_f32.hip: the fp32 1x1
// kernel with the epilogue on its own waves)
constexpr int kNamedTiles = 10;
const char* const kTileNames[kNamedTiles] = {"conv_igemm_f32<128x128>", "conv_igemm_f32<128x64>", "conv_igemm_f32<128x32>",
                                             "conv_igemm_f32<64x64>", "conv_igemm_f32<128x64,stem>", "conv_igemm_f32<256x128>",
                                             "", "", "", "conv1x1_ew_f32<128x64>"};
const int kTileBM[NTILES] = {128, 128, 128, 64, 128, 256};
const int kTileBN[NTILES] = {128, 64, 32, 64, 64, 128};

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
  // every non-stem convolution: the LDS-DMA kernel (conv_igemm_dma.hip, fp32 instantiation); the kernel below keeps the
  // 7x7 stem, whose rows (8 taps x 4 channels) need per-tap bounds checks inside a 128-byte row
  if (!d->stem) return rs_conv_dma_f32(d, src1, src2, weight, scale, shift, residual, relu_mask, out, stream);
  ConvArgs a;
  a.src1 = src1;
  a.src2 = src2;
  a.wgt = weight;
  a.scale = scale;
  a.shift = shift;
  a.res = residual;
  a.mask = relu_mask;
  a.out = out;
  a.N = d->N;
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
  {
    // the kernel addresses its inputs with 32-bit byte offsets relative to the first image of a tile: a tile of
    // <= 128 output pixels touches at most 128 / (Ho*Wo) + 2 images
    const long cmax = d->stem ? 4 : (d->C1 > d->C2 ? d->C1 : d->C2);
    const long img_bytes = (long)d->Hs * d->Ws * cmax * 4;
    const long span = (128 / ((long)d->Ho * d->Wo) + 2) * img_bytes;
    if (span >= (1L << 31)) return RS_EINVAL;
  }
  a.cpt = d->stem ? 1 : (d->C1 + d->C2) / 32;
  a.nk = d->stem ? d->kh : d->kh * d->kw * a.cpt;
  a.Kw = a.nk * 32;
  a.relu = d->relu;

  const int tile = pick_tile(d);
  a.ntiles = d->Cout / kTileBN[tile];
  const int grid = rs_cdiv(M, kTileBM[tile]) * a.ntiles;
  hipStream_t s = (hipStream_t)stream;
  if (tile != TSTEM) return RS_EINVAL;
  conv_igemm_f32<128, 64, 2, 2, 1><<<grid, 256, 0, s>>>(a);
  return RS_LAUNCH_RESULT();
}

extern "C" int rs_pack_stem_weight(const float* w_krsc, float* packed, int Cout, int kh, int kw, int Cin,
                                   rs_stream_t stream) {
  if (!w_krsc || !packed || Cout <= 0 || kh <= 0 || kw <= 0 || kw > 8 || Cin <= 0 || Cin > 4) return RS_EINVAL;
  const int total = Cout * kh * 32;
  pack_stem_weight_kernel<<<rs_cdiv(total, 256), 256, 0, (hipStream_t)stream>>>(w_krsc, packed, Cout, kh, kw, Cin);
  return RS_LAUNCH_RESULT();
}

// Weight-gradient convolution for gfx950 on v_mfma_f32_32x32x2_f32 (exact fp32).
//
// The filter gradients that autograd synthesises for every nn.Conv2d of UNet.forward when the reference calls
// loss.backward() (robosat/tools/train.py:186):
//
//     dW[co][ky][kx][ci] = sum over output pixels m=(n,oy,ox) of  dy[m][co] * in[n][oy*stride-pad+ky][ox*stride-pad+kx][ci]
//
// GEMM view: rows = Cout, cols = one filter tap x a slice of Cin, REDUCTION over the M = N*Ho*Wo output pixels.
// `in` is read through the same gather as the forward pass (nearest-x2 upsample, two-source concat, stem packing),
// so the upsampled / concatenated tensors are not materialised in the backward pass either.
//
//   block  = tile BMo couts x BNo cins of one tap, over a contiguous range of pixels (split-P), walked in chunks of
//            32 pixels; LDS double buffered, next chunk prefetched into registers during the MFMAs.
//   loads  = buffer_load_dwordx4 through SRSRC descriptors (hardware bounds check => zeros for padding / tail
//            pixels, offset = -1): branch-free, so the address math + loads of chunk k+1 interleave with the MFMAs
//            of chunk k; pixel -> (n, oy, ox) uses mul-hi division by constants prepared on the host.
//   LDS    = both tiles are stored as loaded, [32 pixels][channels]: the MFMA operands (A[i=co][k=pixel],
//            B[k=pixel][j=ci], lane l <-> channel l&31, pixel parity l>>5) are read with ds_read_b32 on consecutive
//            channels => conflict free; at 64 cycles per fp32 MFMA one b32 per operand per MFMA is ample.
//   split-P: partial tiles go to a workspace [split][Cout][K] and are summed by a second (streaming) kernel:
//            deterministic, no atomics.
#include "common.h"

namespace {

struct WgradArgs {
  const float* dy;
  const float* src1;
  const float* src2;
  float* out;  // [splits][Cout][K]  (K = taps * Cin, or kh*32 for the stem)
  int N, Hs, Ws, C1, C2, Hv, Wv, ups;
  int kw, stride, pad, Ho, Wo, Cout;
  int M, K, tiles_co, tiles_ci, tiles_k, chunks_per_split;
  rs_fastdiv div_howo, div_wo;
};

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ __amdgpu_buffer_rsrc_t wg_make_rsrc(const float* base, long bytes) {
  const unsigned int n = bytes > 0xFFFFFFFEL ? 0xFFFFFFFEu : (unsigned int)(bytes < 0 ? 0 : bytes);
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(base), 0, (int)n, 0x00020000);
}

__device__ __forceinline__ f32x4 wg_buffer_load4(__amdgpu_buffer_rsrc_t r, int byte_off) {
  return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, byte_off, 0, 0));
}

template <int BMo, int BNo, int WGM, int WGN, int STEM>
__global__ __launch_bounds__(64 * WGM * WGN) void conv_wgrad_f32(const WgradArgs p) {
  constexpr int NT = 64 * WGM * WGN;
  constexpr int WM = BMo / WGM, WN = BNo / WGN;
  constexpr int TM = WM / 32, TN = WN / 32;
  constexpr int AQ = BMo / 4, BQ = BNo / 4;  // float4 per tile row
  constexpr int AR = 8 * BMo / NT, BR = 8 * BNo / NT;  // float4 loads per thread per chunk
  constexpr int BUF = 32 * (BMo + BNo);
  static_assert(TM >= 1 && TN >= 1 && AR >= 1 && BR >= 1, "bad tile");

  __shared__ __attribute__((aligned(16))) float lds[2 * BUF];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int wm = wave / WGN, wn = wave % WGN;

  int bid = rs_xcd_remap(blockIdx.x, gridDim.x);
  const int tco = bid % p.tiles_co;
  bid /= p.tiles_co;
  const int tk = bid % p.tiles_k;
  const int split = bid / p.tiles_k;
  const int tap = tk / p.tiles_ci, tci = tk - tap * p.tiles_ci;
  const int ky = STEM ? tap : tap / p.kw;
  const int kx = STEM ? 0 : tap - ky * p.kw;
  const int co0 = tco * BMo;
  const int ci0 = tci * BNo;  // STEM: 0

  // source of this tile's input channels
  const float* src = p.src1;
  int Cs = STEM ? 4 : p.C1, cs = ci0;
  if (!STEM && ci0 >= p.C1) {
    src = p.src2;
    Cs = p.C2;
    cs = ci0 - p.C1;
  }

  const int chunk0 = split * p.chunks_per_split;
  const int total_chunks = (p.M + 31) >> 5;
  int chunk1 = chunk0 + p.chunks_per_split;
  if (chunk1 > total_chunks) chunk1 = total_chunks;
  const int HoWo = p.Ho * p.Wo;

  // 32-bit byte offsets relative to this split's first pixel (dy) / first image (input); validated on the host
  const int m_first = chunk0 << 5;
  const int n_first = (int)rs_div((unsigned)m_first, p.div_howo);
  const long img = (long)p.Hs * p.Ws * Cs;
  const __amdgpu_buffer_rsrc_t rsrc_dy = wg_make_rsrc(p.dy + (long)m_first * p.Cout, ((long)p.M - m_first) * p.Cout * 4);
  const __amdgpu_buffer_rsrc_t rsrc_x = wg_make_rsrc(src + n_first * img, (long)(p.N - n_first) * img * 4);
  const int ush = p.ups ? 1 : 0;
  const int upar = p.ups == 2 ? 1 : 0;

  f32x4 ra[AR], rb[BR];
  int lchunk = chunk0;  // next chunk to fetch

  // a third of the next chunk's loads (parts 0..2), branch-free
  auto load_part = [&](int part) __attribute__((always_inline)) {
    const int mbase = lchunk << 5;
#pragma unroll
    for (int i = 0; i < AR; ++i) {
      if ((i % 3) != part) continue;
      const int f = tid + NT * i;
      const int row = f / AQ, c4 = f - row * AQ;  // AQ is a power of two
      const int m = mbase + row;
      const int off = (m < p.M) ? ((m - m_first) * p.Cout + co0 + c4 * 4) * 4 : -1;
      ra[i] = wg_buffer_load4(rsrc_dy, off);
    }
#pragma unroll
    for (int i = 0; i < BR; ++i) {
      if (((i + 1) % 3) != part) continue;
      const int f = tid + NT * i;
      const int row = f / BQ, c4 = f - row * BQ;
      const int m = mbase + row;
      const int n = (int)rs_div((unsigned)m, p.div_howo);
      const int rem = m - n * HoWo;
      const int oy = (int)rs_div((unsigned)rem, p.div_wo);
      const int ox = rem - oy * p.Wo;
      const int iy = oy * p.stride - p.pad + ky;
      const int ix = ox * p.stride - p.pad + (STEM ? c4 : kx);
      bool ok = (m < p.M) && ((unsigned)iy < (unsigned)p.Hv) && ((unsigned)ix < (unsigned)p.Wv);
      ok = ok && (((iy | ix) & upar) == 0);
      const int pix = ((n - n_first) * p.Hs + (iy >> ush)) * p.Ws + (ix >> ush);
      const int off = ok ? (pix * Cs + (STEM ? 0 : cs + c4 * 4)) * 4 : -1;
      rb[i] = wg_buffer_load4(rsrc_x, off);
    }
  };

  auto store_chunk = [&](int buf) __attribute__((always_inline)) {
    float* L = lds + buf * BUF;
#pragma unroll
    for (int i = 0; i < AR; ++i) *reinterpret_cast<f32x4*>(&L[(tid + NT * i) * 4]) = ra[i];
#pragma unroll
    for (int i = 0; i < BR; ++i) *reinterpret_cast<f32x4*>(&L[32 * BMo + (tid + NT * i) * 4]) = rb[i];
  };

  f32x16 acc[TM][TN];
#pragma unroll
  for (int a = 0; a < TM; ++a)
#pragma unroll
    for (int b = 0; b < TN; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

  const int acol = wm * WM + (lane & 31);
  const int bcol = wn * WN + (lane & 31);
  const int kh2 = lane >> 5;

  // a quarter of a chunk: 8 of the 32 pixels = 4 MFMA steps
  auto compute_part = [&](const float* LA, int q) __attribute__((always_inline)) {
    const float* LB = LA + 32 * BMo;
#pragma unroll
    for (int t = 4 * q; t < 4 * q + 4; ++t) {
      const int prow = 2 * t + kh2;
      float a[TM], b[TN];
#pragma unroll
      for (int tm = 0; tm < TM; ++tm) a[tm] = LA[prow * BMo + acol + 32 * tm];
#pragma unroll
      for (int tn = 0; tn < TN; ++tn) b[tn] = LB[prow * BNo + bcol + 32 * tn];
#pragma unroll
      for (int tm = 0; tm < TM; ++tm)
#pragma unroll
        for (int tn = 0; tn < TN; ++tn)
          acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[tm], b[tn], acc[tm][tn], 0, 0, 0);
    }
  };

  if (chunk0 < chunk1) {
#pragma unroll
    for (int part = 0; part < 3; ++part) load_part(part);
    ++lchunk;
    store_chunk(0);
    __syncthreads();
    for (int c = chunk0; c < chunk1; ++c) {
      const float* LA = lds + ((c - chunk0) & 1) * BUF;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        if (q < 3) load_part(q);  // the prefetch past the last chunk reads zeros (m >= M) or pixels of the next split
        compute_part(LA, q);
        __builtin_amdgcn_sched_barrier(0);
      }
      ++lchunk;
      store_chunk((c - chunk0 + 1) & 1);
      __syncthreads();
    }
  }

  // D[i = co][j = ci]: j = lane&31, i = (r&3) + 8*(r>>2) + 4*(lane>>5); 128-B segments per store instruction
  float* out = p.out + (long)split * p.Cout * p.K;
  const int kbase = STEM ? tap * 32 : tap * (p.C1 + p.C2) + ci0;
#pragma unroll
  for (int tm = 0; tm < TM; ++tm)
#pragma unroll
    for (int tn = 0; tn < TN; ++tn)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int co = co0 + wm * WM + tm * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        const int kk = kbase + wn * WN + tn * 32 + (lane & 31);
        out[(long)co * p.K + kk] = acc[tm][tn][r];
      }
}

__global__ void reduce_splits_kernel(const float* __restrict__ ws, float* __restrict__ out, long n4, int splits) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n4) return;
  f32x4 s = *reinterpret_cast<const f32x4*>(ws + i * 4);
  for (int k = 1; k < splits; ++k) {
    const f32x4 v = *reinterpret_cast<const f32x4*>(ws + ((long)k * n4 + i) * 4);
#pragma unroll
    for (int e = 0; e < 4; ++e) s[e] += v[e];
  }
  *reinterpret_cast<f32x4*>(out + i * 4) = s;
}

// [Cout][kh][8][4] (packed stem gradient) -> KRSC [Cout][kh][kw][Cin]
__global__ void unpack_stem_weight_kernel(const float* __restrict__ packed, float* __restrict__ w, int Cout, int kh, int kw,
                                          int Cin) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  const int total = Cout * kh * kw * Cin;
  if (idx >= total) return;
  const int c = idx % Cin;
  int t = idx / Cin;
  const int s = t % kw;
  t /= kw;
  const int r = t % kh, co = t / kh;
  w[idx] = packed[((co * kh + r) * 8 + s) * 4 + c];
}

// forward-layout weights [Cout][kh][kw][Cin] -> data-gradient weights [Cin][kh][kw][Cout] with the taps flipped:
// dgrad of a convolution is itself a convolution of dy with these (conv_igemm.hip, ups = 0 / 2).
__global__ void pack_dgrad_weight_kernel(const float* __restrict__ w, float* __restrict__ out, int Cout, int taps, int Cin) {
  __shared__ float tile[32][33];
  const int tap = blockIdx.z;
  const int ci0 = blockIdx.x * 32, co0 = blockIdx.y * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 256 threads: 32 x 8
  for (int r = ty; r < 32; r += 8) {
    const int co = co0 + r, ci = ci0 + tx;
    tile[r][tx] = (co < Cout && ci < Cin) ? w[((long)co * taps + tap) * Cin + ci] : 0.f;
  }
  __syncthreads();
  const int ftap = taps - 1 - tap;
  for (int r = ty; r < 32; r += 8) {
    const int ci = ci0 + r, co = co0 + tx;
    if (ci < Cin && co < Cout) out[((long)ci * taps + ftap) * Cout + co] = tile[tx][r];
  }
}

struct Plan {
  int bmo, bno, variant, tiles_co, tiles_ci, taps, tiles_k, splits, chunks_per_split;
  long K;
};

enum { V128x128 = 0, V128x64, V64x128, V64x64, V32x128, V32x32, VSTEM };

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
  return (long)d->N * d->Ho * d->Wo < (1L << 31);
}

int largest_tile(int c) { return (c % 128 == 0) ? 128 : (c % 64 == 0) ? 64 : 32; }

Plan plan(const rs_conv_desc* d) {
  Plan pl;
  const long M = (long)d->N * d->Ho * d->Wo;
  if (d->stem) {
    pl.bmo = 64;
    pl.bno = 32;
    pl.variant = VSTEM;
    pl.taps = d->kh;
    pl.tiles_ci = 1;
    pl.K = (long)d->kh * 32;
  } else {
    pl.bmo = largest_tile(d->Cout);
    pl.bno = largest_tile(d->C1);
    if (d->C2 > 0) {
      const int b2 = largest_tile(d->C2);
      if (b2 < pl.bno) pl.bno = b2;
    }
    if (pl.bno == 32) pl.bmo = 32;
    if (pl.bmo == 32 && pl.bno == 64) pl.bno = 32;
    pl.variant = pl.bmo == 128 ? (pl.bno == 128 ? V128x128 : V128x64)
                 : pl.bmo == 64 ? (pl.bno == 128 ? V64x128 : V64x64)
                                : (pl.bno == 128 ? V32x128 : V32x32);
    pl.taps = d->kh * d->kw;
    pl.tiles_ci = (d->C1 + d->C2) / pl.bno;
    pl.K = (long)pl.taps * (d->C1 + d->C2);
  }
  pl.tiles_co = d->Cout / pl.bmo;
  pl.tiles_k = pl.taps * pl.tiles_ci;
  const long tiles = (long)pl.tiles_co * pl.tiles_k;
  const long chunks = (M + 31) / 32;
  long s = (1024 + tiles - 1) / tiles;       // aim at >= 1024 blocks ...
  const long smax = (chunks + 7) / 8;        // ... of at least 8 chunks (256 pixels) each
  if (s > smax) s = smax;
  if (s < 1) s = 1;
  // 32-bit byte offsets inside a split: dy spans pixels_per_split * Cout floats, the input spans the images the split
  // touches (+1 for the prefetch past its end); shrink the splits until both fit
  const long cmax = d->stem ? 4 : (d->C1 > d->C2 ? d->C1 : d->C2);
  const long img_bytes = (long)d->Hs * d->Ws * cmax * 4;
  const long howo = (long)d->Ho * d->Wo;
  for (;;) {
    pl.chunks_per_split = (int)((chunks + s - 1) / s);
    const long px = ((long)pl.chunks_per_split + 1) * 32;
    const long span_dy = px * d->Cout * 4;
    const long span_x = (px / howo + 2) * img_bytes;
    if ((span_dy < (1L << 31) && span_x < (1L << 31)) || pl.chunks_per_split == 1) break;
    s *= 2;
  }
  pl.splits = (int)((chunks + pl.chunks_per_split - 1) / pl.chunks_per_split);
  return pl;
}

}  // namespace

extern "C" long rs_conv2d_wgrad_workspace_bytes(const rs_conv_desc* d) {
  if (!valid(d)) return RS_EINVAL;
  const Plan pl = plan(d);
  return (long)pl.splits * d->Cout * pl.K * (long)sizeof(float);
}

extern "C" int rs_conv2d_wgrad(const rs_conv_desc* d, const float* dy, const float* src1, const float* src2, float* dw,
                               void* workspace, rs_stream_t stream) {
  if (!valid(d) || !dy || !src1 || !dw || !workspace) return RS_EINVAL;
  if (d->C2 > 0 && !src2) return RS_EINVAL;
  const Plan pl = plan(d);
  WgradArgs a;
  a.dy = dy;
  a.src1 = src1;
  a.src2 = src2;
  a.out = reinterpret_cast<float*>(workspace);
  a.N = d->N;
  a.Hs = d->Hs;
  a.Ws = d->Ws;
  a.C1 = d->C1;
  a.C2 = d->C2;
  a.ups = d->ups;
  a.div_howo = rs_make_fastdiv((unsigned)(d->Ho * d->Wo));
  a.div_wo = rs_make_fastdiv((unsigned)d->Wo);
  a.Hv = d->ups == 0 ? d->Hs : (d->ups == 1 ? 2 * d->Hs : 2 * d->Hs - 1);
  a.Wv = d->ups == 0 ? d->Ws : (d->ups == 1 ? 2 * d->Ws : 2 * d->Ws - 1);
  a.kw = d->kw;
  a.stride = d->stride;
  a.pad = d->pad;
  a.Ho = d->Ho;
  a.Wo = d->Wo;
  a.Cout = d->Cout;
  a.M = (int)((long)d->N * d->Ho * d->Wo);
  a.K = (int)pl.K;
  a.tiles_co = pl.tiles_co;
  a.tiles_ci = pl.tiles_ci;
  a.tiles_k = pl.tiles_k;
  a.chunks_per_split = pl.chunks_per_split;
  const int grid = pl.tiles_co * pl.tiles_k * pl.splits;
  hipStream_t s = (hipStream_t)stream;
  switch (pl.variant) {
    case V128x128: conv_wgrad_f32<128, 128, 2, 2, 0><<<grid, 256, 0, s>>>(a); break;
    case V128x64: conv_wgrad_f32<128, 64, 2, 2, 0><<<grid, 256, 0, s>>>(a); break;
    case V64x128: conv_wgrad_f32<64, 128, 2, 2, 0><<<grid, 256, 0, s>>>(a); break;
    case V64x64: conv_wgrad_f32<64, 64, 2, 2, 0><<<grid, 256, 0, s>>>(a); break;
    case V32x128: conv_wgrad_f32<32, 128, 1, 4, 0><<<grid, 256, 0, s>>>(a); break;
    case V32x32: conv_wgrad_f32<32, 32, 1, 1, 0><<<grid, 64, 0, s>>>(a); break;
    case VSTEM: conv_wgrad_f32<64, 32, 2, 1, 1><<<grid, 128, 0, s>>>(a); break;
    default: return RS_EINVAL;
  }
  const long n = (long)d->Cout * pl.K;  // multiple of 4
  reduce_splits_kernel<<<rs_cdiv(n / 4, 256), 256, 0, s>>>(a.out, dw, n / 4, pl.splits);
  return RS_LAUNCH_RESULT();
}

extern "C" int rs_unpack_stem_weight(const float* packed, float* w_krsc, int Cout, int kh, int kw, int Cin,
                                     rs_stream_t stream) {
  if (!packed || !w_krsc || Cout <= 0 || kh <= 0 || kw <= 0 || kw > 8 || Cin <= 0 || Cin > 4) return RS_EINVAL;
  const int total = Cout * kh * kw * Cin;
  unpack_stem_weight_kernel<<<rs_cdiv(total, 256), 256, 0, (hipStream_t)stream>>>(packed, w_krsc, Cout, kh, kw, Cin);
  return RS_LAUNCH_RESULT();
}

extern "C" int rs_pack_dgrad_weight(const float* w_krsc, float* out, int Cout, int kh, int kw, int Cin, rs_stream_t stream) {
  if (!w_krsc || !out || Cout <= 0 || kh <= 0 || kw <= 0 || Cin <= 0) return RS_EINVAL;
  dim3 grid(rs_cdiv(Cin, 32), rs_cdiv(Cout, 32), kh * kw);
  pack_dgrad_weight_kernel<<<grid, 256, 0, (hipStream_t)stream>>>(w_krsc, out, Cout, kh * kw, Cin);
  return RS_LAUNCH_RESULT();
}

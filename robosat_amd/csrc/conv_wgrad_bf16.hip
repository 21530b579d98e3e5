// Weight-gradient convolution for gfx950 on v_mfma_f32_32x32x16_bf16: bf16 dy and activations, fp32 accumulation and
// fp32 KRSC result (the optimizer's master gradient).  The bf16 training path of BASELINE configs[2].
//
//     dW[co][ky][kx][ci] = sum over output pixels m=(n,oy,ox) of  dy[m][co] * in[n][oy*stride-pad+ky][ox*stride-pad+kx][ci]
//
// i.e. the filter gradients autograd synthesises for every nn.Conv2d of UNet.forward when the reference calls
// loss.backward() (robosat/tools/train.py:186).  GEMM view as conv_wgrad.hip: rows = Cout tile, cols = one filter tap
// x a Cin tile, REDUCTION over pixels; `in` is read through the forward gather (nearest-x2 upsample, 2-source concat).
//
//   block  = BMo couts x BNo cins of one tap over a contiguous pixel range (split-P), walked in chunks of 64 pixels
//            (4 MFMA k-steps); LDS double buffered, the next chunk is fetched into registers during the MFMAs.
//   gather = one wave-sized job per chunk: threads 0..63 decode pixel m -> source pixel (mul-hi divisions) for chunk
//            k+2 into a double-buffered 64-entry LDS table; every staging thread then needs two ds_read_b128 and a
//            multiply-add per load instead of 8 coordinate decodes.
//   LDS    = channel-major [channel'][64 pixels] bf16 (128-byte rows, 16-byte pieces XOR-swizzled with (row>>1)&7).
//            Both operands arrive pixel-major from HBM ([pixel][channel]); a staging thread owns an 8-pixel x
//            8-channel block: eight 16-byte loads (consecutive lanes = consecutive channel octets of one pixel: full
//            lines), an 8x8 16-bit transpose in registers (32 v_perm_b32) and eight ds_write_b128.  LDS row
//            R = ch*Q + cq holds channel 8*cq + ch (Q = channels/8): consecutive lanes write consecutive rows (conflict
//            free); the epilogue's addressing undoes the permutation.  MFMA operands are then read exactly as in the
//            forward kernel: one ds_read_b128 = 8 pixels of the reduction per lane.
//   split-P: partial tiles -> workspace [split][Cout][K], summed by a streaming kernel: deterministic, no atomics.
#include "common.h"

namespace {

struct WgradArgsB {
  const bf16_t* dy;
  const bf16_t* src1;
  const bf16_t* src2;
  float* out;  // [splits][Cout][K]
  int N, Hs, Ws, C1, C2, Hv, Wv, ups;
  int kw, stride, pad, Ho, Wo, Cout;
  int M, K, tiles_co, tiles_ci, tiles_k, chunks_per_split;
  rs_fastdiv div_howo, div_wo;
};

typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ __amdgpu_buffer_rsrc_t wb_make_rsrc(const void* base, long bytes) {
  const unsigned int n = bytes > 0xFFFFFFFEL ? 0xFFFFFFFEu : (unsigned int)(bytes < 0 ? 0 : bytes);
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, (int)n, 0x00020000);
}

__device__ __forceinline__ u32x4 wb_buffer_load(__amdgpu_buffer_rsrc_t r, int byte_off) {
  return __builtin_amdgcn_raw_buffer_load_b128(r, byte_off, 0, 0);
}

// 8 pixels x 8 channels (in[e] = 8 channels of pixel e) -> out[ch] = 8 pixels of channel ch
__device__ __forceinline__ void wb_transpose8x8(const u32x4 (&in)[8], u32x4 (&out)[8]) {
#pragma unroll
  for (int w = 0; w < 4; ++w)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const unsigned int x = in[2 * j][w], y = in[2 * j + 1][w];
      out[2 * w][j] = __builtin_amdgcn_perm(y, x, 0x05040100u);      // (lo x, lo y)
      out[2 * w + 1][j] = __builtin_amdgcn_perm(y, x, 0x07060302u);  // (hi x, hi y)
    }
}

constexpr int PK = 64;    // pixels per chunk
constexpr int ROWB = 128;  // bytes per LDS row

template <int BMo, int BNo, int WGM, int WGN>
__global__ __launch_bounds__(64 * WGM * WGN, 2) void conv_wgrad_bf16(const WgradArgsB p) {
  constexpr int NT = 64 * WGM * WGN;
  constexpr int WM = BMo / WGM, WN = BNo / WGN;
  constexpr int TM = WM / 32, TN = WN / 32;
  constexpr int AQ = BMo / 8, BQ = BNo / 8;            // channel octets per tile
  constexpr int ASL = BMo < 64 ? 64 : BMo;             // staging slots of the A tile (one 8x8 block each), wave aligned
  constexpr int BSL = BNo < 64 ? 64 : BNo;
  constexpr int NJ = (ASL + BSL + NT - 1) / NT;        // staging blocks per thread
  constexpr int BUF = (BMo + BNo) * ROWB;
  static_assert(TM >= 1 && TN >= 1 && (ASL % 64) == 0 && (BSL % 64) == 0, "bad tile");
  static_assert(NT >= 64 && (ASL % NT == 0 || NT % ASL == 0 || NJ == 1), "slot layout");

  __shared__ __attribute__((aligned(16))) unsigned char smem[2 * BUF + 2 * PK * 4];
  int* tabs = reinterpret_cast<int*>(smem + 2 * BUF);

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
  const int ky = tap / p.kw;
  const int kx = tap - ky * p.kw;
  const int co0 = tco * BMo;
  const int ci0 = tci * BNo;

  const bf16_t* src = p.src1;
  int Cs = p.C1, cs = ci0;
  if (ci0 >= p.C1) {
    src = p.src2;
    Cs = p.C2;
    cs = ci0 - p.C1;
  }

  const int chunk0 = split * p.chunks_per_split;
  const int total_chunks = (p.M + PK - 1) / PK;
  int chunk1 = chunk0 + p.chunks_per_split;
  if (chunk1 > total_chunks) chunk1 = total_chunks;
  const int HoWo = p.Ho * p.Wo;

  const int m_first = chunk0 * PK;
  const int n_first = (int)rs_div((unsigned)m_first, p.div_howo);
  const long img = (long)p.Hs * p.Ws * Cs;
  const __amdgpu_buffer_rsrc_t rsrc_dy = wb_make_rsrc(p.dy + (long)m_first * p.Cout, ((long)p.M - m_first) * p.Cout * 2);
  const __amdgpu_buffer_rsrc_t rsrc_x = wb_make_rsrc(src + n_first * img, (long)(p.N - n_first) * img * 2);
  const int ush = p.ups ? 1 : 0;
  const int upar = p.ups == 2 ? 1 : 0;

  // pixel m -> source pixel of this block's tap (relative to image n_first), -1 = contributes zeros
  auto fill_table = [&](int chunk, int which) __attribute__((always_inline)) {
    if (tid < PK) {
      const int m = chunk * PK + tid;
      int pix = -1;
      if (m < p.M) {
        const int n = (int)rs_div((unsigned)m, p.div_howo);
        const int rem = m - n * HoWo;
        const int oy = (int)rs_div((unsigned)rem, p.div_wo);
        const int ox = rem - oy * p.Wo;
        const int iy = oy * p.stride - p.pad + ky;
        const int ix = ox * p.stride - p.pad + kx;
        const bool ok = ((unsigned)iy < (unsigned)p.Hv) && ((unsigned)ix < (unsigned)p.Wv) && (((iy | ix) & upar) == 0);
        if (ok) pix = ((n - n_first) * p.Hs + (iy >> ush)) * p.Ws + (ix >> ush);
      }
      tabs[which * PK + tid] = pix;
    }
  };

  // staging roles: slot = tid + NT*j; slots [0, ASL) stage dy blocks, [ASL, ASL+BSL) input blocks (wave uniform)
  u32x4 rg[NJ][8];
  int lchunk = chunk0;  // next chunk to fetch

  auto load_block = [&](int j, int e0, int e1, int which) __attribute__((always_inline)) {
    const int slot = tid + NT * j;
    if (slot < ASL) {
      const int cq = slot % AQ, pg = slot / AQ;
      const bool act = slot < BMo;
      const int mb = lchunk * PK + pg * 8;
#pragma unroll
      for (int e = e0; e < e1; ++e) {
        const int m = mb + e;
        const int off = (act && m < p.M) ? ((m - m_first) * p.Cout + co0 + cq * 8) * 2 : -1;
        rg[j][e] = wb_buffer_load(rsrc_dy, off);
      }
    } else if (slot < ASL + BSL) {
      const int s2 = slot - ASL;
      const int cq = s2 % BQ, pg = s2 / BQ;
      const bool act = s2 < BNo;
      const int cb = (cs + cq * 8) * 2, cs2 = Cs * 2;
      const i32x4 t0 = *reinterpret_cast<const i32x4*>(&tabs[which * PK + (pg & 7) * 8]);
      const i32x4 t1 = *reinterpret_cast<const i32x4*>(&tabs[which * PK + (pg & 7) * 8 + 4]);
#pragma unroll
      for (int e = e0; e < e1; ++e) {
        const int pix = e < 4 ? t0[e & 3] : t1[e & 3];
        const int off = (act && pix >= 0) ? pix * cs2 + cb : -1;
        rg[j][e] = wb_buffer_load(rsrc_x, off);
      }
    }
  };

  auto store_block = [&](int j, int buf) __attribute__((always_inline)) {
    unsigned char* L = smem + buf * BUF;
    const int slot = tid + NT * j;
    int rbase, cq, pg, Q;
    bool act;
    if (slot < ASL) {
      cq = slot % AQ;
      pg = slot / AQ;
      Q = AQ;
      rbase = 0;
      act = slot < BMo;
    } else {
      const int s2 = slot - ASL;
      cq = s2 % BQ;
      pg = s2 / BQ;
      Q = BQ;
      rbase = BMo;
      act = s2 < BNo && slot < ASL + BSL;
    }
    if (act) {
      u32x4 t[8];
      wb_transpose8x8(rg[j], t);
#pragma unroll
      for (int ch = 0; ch < 8; ++ch) {
        const int R = ch * Q + cq;  // tile-local LDS row of channel 8*cq + ch
        *reinterpret_cast<u32x4*>(L + (rbase + R) * ROWB + ((pg ^ ((R >> 1) & 7)) * 16)) = t[ch];
      }
    }
  };

  f32x16 acc[TM][TN];
#pragma unroll
  for (int a = 0; a < TM; ++a)
#pragma unroll
    for (int b = 0; b < TN; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

  const int frow = lane & 31;
  const int fl = (frow >> 1) & 7;
  int foff[4];
#pragma unroll
  for (int s = 0; s < 4; ++s) foff[s] = ((2 * s + (lane >> 5)) ^ fl) * 16;
  const int abase = (wm * WM + frow) * ROWB;
  const int bbase = (BMo + wn * WN + frow) * ROWB;

  auto read_frag = [&](const unsigned char* L, int s, bf16x8 (&a)[TM], bf16x8 (&b)[TN]) __attribute__((always_inline)) {
#pragma unroll
    for (int tm = 0; tm < TM; ++tm) a[tm] = *reinterpret_cast<const bf16x8*>(L + abase + 32 * tm * ROWB + foff[s]);
#pragma unroll
    for (int tn = 0; tn < TN; ++tn) b[tn] = *reinterpret_cast<const bf16x8*>(L + bbase + 32 * tn * ROWB + foff[s]);
  };
  auto mma_frag = [&](const bf16x8 (&a)[TM], const bf16x8 (&b)[TN]) __attribute__((always_inline)) {
#pragma unroll
    for (int tm = 0; tm < TM; ++tm)
#pragma unroll
      for (int tn = 0; tn < TN; ++tn)
        acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[tm], b[tn], acc[tm][tn], 0, 0, 0);
  };

  if (chunk0 < chunk1) {
    fill_table(chunk0, 0);
    fill_table(chunk0 + 1, 1);
    __syncthreads();
#pragma unroll
    for (int j = 0; j < NJ; ++j) load_block(j, 0, 8, 0);
    ++lchunk;
#pragma unroll
    for (int j = 0; j < NJ; ++j) store_block(j, 0);
    __syncthreads();
    for (int c = chunk0; c < chunk1; ++c) {
      const int it = c - chunk0;
      const unsigned char* L = smem + (it & 1) * BUF;
      const int tw = (it + 1) & 1;  // table of chunk c+1
      bf16x8 fa[2][TM], fb[2][TN];
      read_frag(L, 0, fa[0], fb[0]);
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        if (s < 3) {
          read_frag(L, s + 1, fa[(s + 1) & 1], fb[(s + 1) & 1]);
          // loads of chunk c+1: thirds (3 + 3 + 2 pixels of every block); past the last chunk they read zeros / the
          // next split's pixels, harmlessly
#pragma unroll
          for (int j = 0; j < NJ; ++j) load_block(j, s * 3, s == 2 ? 8 : s * 3 + 3, tw);
        }
        mma_frag(fa[s & 1], fb[s & 1]);
        __builtin_amdgcn_sched_barrier(0);
      }
      ++lchunk;
      fill_table(c + 2, it & 1);
#pragma unroll
      for (int j = 0; j < NJ; ++j) store_block(j, (it + 1) & 1);
      __syncthreads();
    }
  }

  // D[i][j]: i = (r&3) + 8*(r>>2) + 4*(lane>>5) is an LDS row of the dy tile, j = lane&31 one of the input tile;
  // LDS row R <-> channel 8*(R % Q) + R / Q
  float* out = p.out + (long)split * p.Cout * p.K;
  const int kbase = tap * (p.C1 + p.C2) + ci0;
#pragma unroll
  for (int tm = 0; tm < TM; ++tm)
#pragma unroll
    for (int tn = 0; tn < TN; ++tn) {
      const int Rb = wn * WN + tn * 32 + (lane & 31);
      const int kk = kbase + 8 * (Rb % BQ) + Rb / BQ;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int Ra = wm * WM + tm * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        const int co = co0 + 8 * (Ra % AQ) + Ra / AQ;
        out[(long)co * p.K + kk] = acc[tm][tn][r];
      }
    }
}

struct Plan {
  int bmo, bno, variant, tiles_co, tiles_ci, taps, tiles_k, splits, chunks_per_split;
  long K;
};

enum { V128x128 = 0, V128x64, V64x128, V64x64, V32x128, V32x32 };

bool valid(const rs_conv_desc* d) {
  if (!d) return false;
  if (d->N <= 0 || d->Hs <= 0 || d->Ws <= 0 || d->Ho <= 0 || d->Wo <= 0) return false;
  if (d->kh <= 0 || d->kw <= 0 || d->stride <= 0 || d->pad < 0) return false;
  if (d->Cout <= 0 || (d->Cout % 32) != 0) return false;
  if (d->ups < 0 || d->ups > 2 || d->stem) return false;
  if (d->C1 <= 0 || (d->C1 % 32) != 0 || d->C2 < 0 || (d->C2 % 32) != 0) return false;
  return (long)d->N * d->Ho * d->Wo < (1L << 31);
}

int largest_tile(int c) { return (c % 128 == 0) ? 128 : (c % 64 == 0) ? 64 : 32; }

Plan plan(const rs_conv_desc* d) {
  Plan pl;
  const long M = (long)d->N * d->Ho * d->Wo;
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
  pl.tiles_co = d->Cout / pl.bmo;
  pl.tiles_k = pl.taps * pl.tiles_ci;
  const long tiles = (long)pl.tiles_co * pl.tiles_k;
  const long chunks = (M + PK - 1) / PK;
  long s = (1024 + tiles - 1) / tiles;  // aim at >= 1024 blocks ...
  const long smax = (chunks + 7) / 8;   // ... of at least 8 chunks (512 pixels) each
  if (s > smax) s = smax;
  if (s < 1) s = 1;
  // 32-bit byte offsets inside a split: shrink the splits until dy and the input both fit
  const long cmax = d->C1 > d->C2 ? d->C1 : d->C2;
  const long img_bytes = (long)d->Hs * d->Ws * cmax * 2;
  const long howo = (long)d->Ho * d->Wo;
  for (;;) {
    pl.chunks_per_split = (int)((chunks + s - 1) / s);
    const long px = ((long)pl.chunks_per_split + 2) * PK;
    const long span_dy = px * d->Cout * 2;
    const long span_x = (px / howo + 2) * img_bytes;
    if ((span_dy < (1L << 31) && span_x < (1L << 31)) || pl.chunks_per_split == 1) break;
    s *= 2;
  }
  pl.splits = (int)((chunks + pl.chunks_per_split - 1) / pl.chunks_per_split);
  return pl;
}

}  // namespace

extern "C" long rs_conv2d_wgrad_bf16_workspace_bytes(const rs_conv_desc* d) {
  if (!valid(d)) return RS_EINVAL;
  int tb = 0, tslices = 0;
  if (rs_wgrad_thin_plan(d, &tb, &tslices)) {
    const long n = 32L * 9 * d->C1;
    return (tslices * n + rs_reduce_scratch_floats(n, tslices)) * (long)sizeof(float);
  }
  const Plan pl = plan(d);
  const long n = (long)d->Cout * pl.K;
  return (pl.splits * n + rs_reduce_scratch_floats(n, pl.splits)) * (long)sizeof(float);
}

extern "C" int rs_conv2d_wgrad_bf16(const rs_conv_desc* d, const rs_bf16* dy, const rs_bf16* src1, const rs_bf16* src2,
                                    float* dw, void* workspace, rs_stream_t stream) {
  if (!valid(d) || !dy || !src1 || !dw || !workspace) return RS_EINVAL;
  if (d->C2 > 0 && !src2) return RS_EINVAL;
  {
    int tb = 0, tslices = 0;
    if (rs_wgrad_thin_plan(d, &tb, &tslices)) {  // Cout = 32 3x3 layers: all nine taps in one block
      const int rc = rs_wgrad_thin_launch(d, dy, src1, reinterpret_cast<float*>(workspace), stream);
      if (rc) return rc;
      const long n = 32L * 9 * d->C1;
      float* ws = reinterpret_cast<float*>(workspace);
      return rs_reduce_splits(ws, dw, n, tslices, ws + (long)tslices * n, stream);
    }
  }
  const Plan pl = plan(d);
  WgradArgsB a;
  a.dy = reinterpret_cast<const bf16_t*>(dy);
  a.src1 = reinterpret_cast<const bf16_t*>(src1);
  a.src2 = reinterpret_cast<const bf16_t*>(src2);
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
    case V128x128: conv_wgrad_bf16<128, 128, 2, 2><<<grid, 256, 0, s>>>(a); break;
    case V128x64: conv_wgrad_bf16<128, 64, 2, 2><<<grid, 256, 0, s>>>(a); break;
    case V64x128: conv_wgrad_bf16<64, 128, 2, 2><<<grid, 256, 0, s>>>(a); break;
    case V64x64: conv_wgrad_bf16<64, 64, 2, 2><<<grid, 256, 0, s>>>(a); break;
    case V32x128: conv_wgrad_bf16<32, 128, 1, 4><<<grid, 256, 0, s>>>(a); break;
    case V32x32: conv_wgrad_bf16<32, 32, 1, 1><<<grid, 64, 0, s>>>(a); break;
    default: return RS_EINVAL;
  }
  const int rc = RS_LAUNCH_RESULT();
  if (rc) return rc;
  const long n = (long)d->Cout * pl.K;  // multiple of 4
  return rs_reduce_splits(a.out, dw, n, pl.splits, a.out + (long)pl.splits * n, stream);
}

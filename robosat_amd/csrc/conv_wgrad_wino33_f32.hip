// conv_wgrad_wino33_f32.hip -- the fp32 weight gradient of the stride-1 3x3 / pad-1 convolutions (torchvision Bottleneck.conv2 of
// layer1..layer4 behind robosat/unet.py:127-130, and any plain ConvRelu) in the WINOGRAD domain of F(2x2, 3x3) -- the filter
// gradient autograd synthesises under tools/train.py:186.  Round 6; the twin of conv_wgrad_wino_f32.hip (DecoderBlock).
//
//     forward  Y = A^T [ U (.) V ] A,   U = G g G^T,  V = B^T d B   (d: a tile's 4x4 input patch, Y: its 2x2 outputs)
//     gradient dU[xi] = sum over tiles of  Z[xi] * V[xi],   Z = A dY A^T  (4x4 from the tile's 2x2 dy values),   dg = G^T dU G
//     A^T = [1 1 1 0; 0 1 -1 -1],  B^T = [1 0 -1 0; 0 1 1 0; 0 -1 1 0; 0 1 0 -1],  G = [1 0 0; .5 .5 .5; .5 -.5 .5; 0 0 1]
// Sixteen [Cout x tiles] . [tiles x Cin] products instead of nine over four times as many rows: 16/36 of the multiply-adds.
//
// Mapping.  Sixteen accumulators of v_mfma_f32_32x32x2_f32 would be 256 registers, so TWO waves share a 32 x 32 (cout, cin)
// sub-tile: wave half h owns the transformed rows i = 2 h, 2 h + 1 (xi = 4 i + j): eight accumulators, and only its own half of
// both transforms (rows of Z from all four dy values; rows of V from three of the patch's four rows).
//   block   (32 WCO couts) x (32 WCI cins) x a split of the chunk sequence; 2 WCO WCI waves.  <2, 2>: 64 x 64, eight waves, one
//           block per CU; <1, 1>: 32 x 32, two waves (the 32-channel layers).
//   chunk   eight consecutive tiles of ONE tile row: their patches overlap, so the chunk's source pixels are fetched once -- four
//           rows x 18 pixels instead of 8 x 16 -- next to two rows x 16 pixels of dy; LDS-DMA, gather table (-1 = zeros: padding,
//           odd edges, the tail of a row), ring of three stages, as in conv_wgrad_wino_f32.hip.
//   output  partial tiles [split][16][Cout][Cin]; rs_reduce_splits (fixed order), then wino33_wgrad_finish_kernel applies G^T . G.
#include "conv_wgrad_wino_f32.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));

struct Wino33WgradArgs {
  const float* dy;   // [N][H][W][Cout]
  const float* src;  // [N][H][W][Cin]
  float* part;       // [splits][16][Cout][Cin]
  int N, H, W, Cin, Cout;
  int TY, CX;  // tile rows per image, chunks per tile row
  int nchunks;
  int tiles_co, tiles_ci;
  int chunks_per_split;
  rs_fastdiv div_tycx, div_cx;
};

template <int WCO, int WCI>
__global__ __launch_bounds__(128 * WCO * WCI, WCO * WCI == 4 ? 1 : 3) void conv_wgrad_wino33_f32(const Wino33WgradArgs p) {
  constexpr int PK = 8, NS = PK / 2, RING = 3;
  constexpr int NW = 2 * WCO * WCI;
  constexpr int BM = 32 * WCO, BN = 32 * WCI;
  constexpr int ROWA = BM * 4, ROWB = BN * 4;
  constexpr int PPA = ROWA / 16, PPB = ROWB / 16;
  constexpr int RIA = 1024 / ROWA, RIB = 1024 / ROWB;
  constexpr int NRA = 2 * 2 * PK, NRB = 4 * (2 * PK + 2);  // dy: two rows x 16 pixels; source: four rows x 18 pixels
  constexpr int IA = NRA / RIA, IB = NRB / RIB;
  static_assert(IA * RIA == NRA && IB * RIB == NRB, "whole instructions");
  constexpr int NI = (IA + IB + NW - 1) / NW;
  constexpr int ABYTES = NRA * ROWA, BBYTES = NRB * ROWB;
  constexpr int BUF = ABYTES + BBYTES;
  constexpr int NT = NRA + NRB;
  static_assert(RING * BUF + RING * NT * 4 <= (NW == 8 ? 160 : 53) * 1024, "one / three blocks per CU");
  static_assert(NT <= 64 * NW, "one table entry per thread");

  __shared__ __attribute__((aligned(16))) unsigned char smem[RING * BUF + RING * NT * 4];
  int* tabs = reinterpret_cast<int*>(smem + RING * BUF);  // [RING][NT]: dy rows (u * 16 + x), then source rows (r * 18 + x)

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int half = wave & 1, wsub = wave >> 1;  // transformed rows 2 half, 2 half + 1; sub-tile (wm, wn)
  const int wm = wsub / WCI, wn = wsub % WCI;

  int bid = rs_xcd_remap(blockIdx.x, gridDim.x);
  const int tco = bid % p.tiles_co;
  bid /= p.tiles_co;
  const int tci = bid % p.tiles_ci;
  const int split = bid / p.tiles_ci;
  const int co0 = tco * BM, ci0 = tci * BN;

  const int chunk0 = split * p.chunks_per_split;
  int chunk1 = chunk0 + p.chunks_per_split;
  if (chunk1 > p.nchunks) chunk1 = p.nchunks;
  const int n_first = (int)rs_div((unsigned)chunk0, p.div_tycx);
  const long img = (long)p.H * p.W;
  const __amdgpu_buffer_rsrc_t rsrc_dy = ww_make_rsrc(p.dy + n_first * img * p.Cout, (long)(p.N - n_first) * img * p.Cout * 4);
  const __amdgpu_buffer_rsrc_t rsrc_x = ww_make_rsrc(p.src + n_first * img * p.Cin, (long)(p.N - n_first) * img * p.Cin * 4);

  auto fill_table = [&](int chunk, int which) __attribute__((always_inline)) {
    if (tid < NT) {
      int pix = -1;
      if (chunk < p.nchunks) {
        const int n = (int)rs_div((unsigned)chunk, p.div_tycx);
        const int rem = chunk - n * p.TY * p.CX;
        const int ty = (int)rs_div((unsigned)rem, p.div_cx);
        const int cx = rem - ty * p.CX;
        int y, x;
        if (tid < NRA) {
          y = 2 * ty + (tid >> 4);
          x = 16 * cx + (tid & 15);
        } else {
          const int e = tid - NRA, r = e / 18;
          y = 2 * ty - 1 + r;
          x = 16 * cx - 1 + (e - 18 * r);
        }
        if ((unsigned)y < (unsigned)p.H && (unsigned)x < (unsigned)p.W) pix = ((n - n_first) * p.H + y) * p.W + x;
      }
      tabs[which * NT + tid] = pix;
    }
    rs_lds_writes_done();  // (read by other waves behind a later barrier, which hipcc emits bare: common.h)
  };

  const int ra_a = lane / PPA, pp_a = lane % PPA;
  const int ra_b = lane / PPB, pp_b = lane % PPB;
  const int cola = (co0 + pp_a * 4) * 4, colb = (ci0 + pp_b * 4) * 4;
  const int cout4 = p.Cout * 4, cin4 = p.Cin * 4;
  const unsigned int lds0 = __builtin_amdgcn_readfirstlane(ww_lds_addr(smem));
  int voff[NI];
  unsigned int fL = lds0;
  auto prepare_dma = [&](int buf, int which) __attribute__((always_inline)) {
    fL = lds0 + buf * BUF;
#pragma unroll
    for (int j = 0; j < NI; ++j) {
      const int ii = wave + NW * j;  // wave-uniform
      if (ii < IA) {
        const int pix = tabs[which * NT + RIA * ii + ra_a];
        voff[j] = pix >= 0 ? pix * cout4 + cola : -1;
      } else if (ii < IA + IB) {
        const int pix = tabs[which * NT + NRA + RIB * (ii - IA) + ra_b];
        voff[j] = pix >= 0 ? pix * cin4 + colb : -1;
      }
    }
  };
  auto issue_piece = [&](int j) __attribute__((always_inline)) {
    const int ii = wave + NW * j;
    if (ii < IA) ww_dma16(rsrc_dy, fL + ii * 1024, voff[j]);
    else if (ii < IA + IB) ww_dma16(rsrc_x, fL + ABYTES + (ii - IA) * 1024, voff[j]);
  };

  f32x16 acc[8];
#pragma unroll
  for (int x = 0; x < 8; ++x)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[x][r] = 0.f;

  // fragment addressing: k-step s, lane (i = lane & 31, k = lane >> 5): tile t = 2 s + k of the chunk, channel (sub-tile base + i);
  // dy (u, v) is row u * 16 + 2 t + v, source (r, c) row r * 18 + 2 t + c; this wave reads source rows half .. half + 2 only
  const int arow = (lane >> 5) * 2 * ROWA + (wm * 32 + (lane & 31)) * 4;
  const int brow = ABYTES + ((lane >> 5) * 2 + half * 18) * ROWB + (wn * 32 + (lane & 31)) * 4;
  auto read_frag = [&](const unsigned char* L, int s, float (&dyv)[4], float (&d)[12]) __attribute__((always_inline)) {
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
      for (int v = 0; v < 2; ++v) dyv[2 * u + v] = *reinterpret_cast<const float*>(L + arow + (16 * u + 4 * s + v) * ROWA);
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
      for (int c = 0; c < 4; ++c) d[4 * r + c] = *reinterpret_cast<const float*>(L + brow + (18 * r + 4 * s + c) * ROWB);
  };
  auto kstep = [&](const float (&dyv)[4], const float (&d)[12], auto issue, int s) __attribute__((always_inline)) {
    // rows i = 2 half, 2 half + 1 of Z = A dY A^T and V = B^T d B.  With d' = the three patch rows this wave read (half .. half + 2):
    //   half 0: Tz0 = dY0, Tz1 = dY0 + dY1;        Tv0 = d'0 - d'2 (rows 0, 2), Tv1 = d'1 + d'2 (rows 1, 2)
    //   half 1: Tz0 = dY0 - dY1, Tz1 = -dY1;       Tv0 = d'1 - d'0 (rows 2, 1), Tv1 = d'0 - d'2 (rows 1, 3)
    float tz[2][2], tv[2][4], Z[8], V[8];
#pragma unroll
    for (int v = 0; v < 2; ++v) {
      tz[0][v] = half ? dyv[v] - dyv[2 + v] : dyv[v];
      tz[1][v] = half ? -dyv[2 + v] : dyv[v] + dyv[2 + v];
    }
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      tv[0][c] = half ? d[4 + c] - d[c] : d[c] - d[8 + c];
      tv[1][c] = half ? d[c] - d[8 + c] : d[4 + c] + d[8 + c];
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      Z[4 * i + 0] = tz[i][0];
      Z[4 * i + 1] = tz[i][0] + tz[i][1];
      Z[4 * i + 2] = tz[i][0] - tz[i][1];
      Z[4 * i + 3] = -tz[i][1];
      V[4 * i + 0] = tv[i][0] - tv[i][2];
      V[4 * i + 1] = tv[i][1] + tv[i][2];
      V[4 * i + 2] = tv[i][2] - tv[i][1];
      V[4 * i + 3] = tv[i][1] - tv[i][3];
    }
#pragma unroll
    for (int x = 0; x < 8; ++x) {
      issue(s * 8 + x);
      acc[x] = __builtin_amdgcn_mfma_f32_32x32x2f32(Z[x], V[x], acc[x], 0, 0, 0);
    }
  };
  constexpr int NMMA = NS * 8;
  constexpr int PSTEP = NMMA / (2 * NI) >= 1 ? NMMA / (2 * NI) : 1;
  auto chunk_mma = [&](const unsigned char* L, bool fetch) __attribute__((always_inline)) {
    float dyv[2][4], d[2][12];
    read_frag(L, 0, dyv[0], d[0]);
#pragma unroll
    for (int s = 0; s < NS; ++s) {
      if (s + 1 < NS) read_frag(L, s + 1, dyv[(s + 1) & 1], d[(s + 1) & 1]);
      kstep(dyv[s & 1], d[s & 1], [&](int q) __attribute__((always_inline)) {
        if (fetch && q % PSTEP == 0 && q / PSTEP < NI) issue_piece(q / PSTEP);
      }, s);
    }
    if (fetch) {
#pragma unroll
      for (int q = (NMMA + PSTEP - 1) / PSTEP; q < NI; ++q) issue_piece(q);
    }
  };

  if (chunk0 < chunk1) {  // (the pipeline of conv_wgrad_wino_f32.hip: chunk c in stage / table (c - chunk0) % RING)
#pragma unroll
    for (int r = 0; r < RING; ++r) fill_table(chunk0 + r, r);
    __syncthreads();
#pragma unroll
    for (int r = 0; r < RING - 1; ++r) {
      if (chunk0 + r < chunk1) {
        prepare_dma(r, r);
#pragma unroll
        for (int q = 0; q < NI; ++q) issue_piece(q);
      }
    }
    ww_dma_wait();
    __syncthreads();
    int st = 0;
    for (int c = chunk0; c < chunk1; ++c) {
      const int stf = st == 0 ? RING - 1 : st - 1;
      const bool fetch = c + RING - 1 < chunk1;
      if (fetch) prepare_dma(stf, stf);
      chunk_mma(smem + st * BUF, fetch);
      if (fetch) {
        fill_table(c + RING, st);
        ww_dma_wait_but<(RING - 2) * NI>();
      } else {
        ww_dma_wait();
      }
      __syncthreads();
      st = st == RING - 1 ? 0 : st + 1;
    }
  }

  // D[i][j]: i = cout (tile-local) = (r & 3) + 8 (r >> 2) + 4 (lane >> 5), j = cin (tile-local) = lane & 31; xi = 8 half + x
  float* out = p.part + ((long)split * 16 + 8 * half) * p.Cout * p.Cin;
  const int ci = ci0 + wn * 32 + (lane & 31);
#pragma unroll
  for (int x = 0; x < 8; ++x)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int co = co0 + wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
      out[((long)x * p.Cout + co) * p.Cin + ci] = acc[x][r];
    }
}

// sum [16][Cout][Cin] -> dW [Cout][3][3][Cin]: dg = G^T dU G, G^T = [1 .5 .5 0; 0 .5 -.5 0; 0 .5 .5 1]
__global__ void wino33_wgrad_finish_kernel(const float* __restrict__ sum, float* __restrict__ dw, int Cout, int Cin, long total) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;  // over [Cout][Cin]
  if (i >= total) return;
  const int ci = (int)(i % Cin);
  const long co = i / Cin;
  float u[4][4], t[3][4];
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int b = 0; b < 4; ++b) u[a][b] = sum[((long)(4 * a + b) * Cout + co) * Cin + ci];
#pragma unroll
  for (int b = 0; b < 4; ++b) {
    const float h1 = 0.5f * u[1][b], h2 = 0.5f * u[2][b];
    t[0][b] = u[0][b] + (h1 + h2);
    t[1][b] = h1 - h2;
    t[2][b] = (h1 + h2) + u[3][b];
  }
#pragma unroll
  for (int a = 0; a < 3; ++a) {
    const float h1 = 0.5f * t[a][1], h2 = 0.5f * t[a][2];
    float* o = dw + (co * 9 + a * 3) * Cin + ci;
    o[0] = t[a][0] + (h1 + h2);
    o[Cin] = h1 - h2;
    o[2 * (long)Cin] = (h1 + h2) + t[a][3];
  }
}

struct W33Plan {
  int wide, tiles_co, tiles_ci, ty, cx, nchunks, splits, chunks_per_split;
};

bool w33_plan(const rs_conv_desc* d, W33Plan* pl) {
  if (!d || d->stem || d->ups != 0 || d->kh != 3 || d->kw != 3 || d->stride != 1 || d->pad != 1 || d->Ho != d->Hs || d->Wo != d->Ws || d->C2 != 0)
    return false;
  if (d->N <= 0 || d->Hs < 2 || d->Ws < 2 || d->C1 <= 0 || d->Cout <= 0 || (d->Cout % 32) || (d->C1 % 32)) return false;
  pl->wide = (d->Cout % 64 == 0 && d->C1 % 64 == 0) ? 1 : 0;
  const int b = pl->wide ? 64 : 32;
  pl->tiles_co = d->Cout / b;
  pl->tiles_ci = d->C1 / b;
  pl->ty = (d->Hs + 1) / 2;
  pl->cx = ((d->Ws + 1) / 2 + 7) / 8;
  const long chunks = (long)d->N * pl->ty * pl->cx;
  if (chunks >= (1L << 28)) return false;
  pl->nchunks = (int)chunks;
  const long tiles = (long)pl->tiles_co * pl->tiles_ci;
  const long target = pl->wide ? rs_knobs().wgrad_f32_wino33_blocks : 3L * rs_knobs().wgrad_f32_wino33_blocks;  // (one / three blocks per CU)
  long s = (target + tiles - 1) / tiles;
  const long smax = (chunks + 7) / 8;  // at least 8 chunks (64 tiles) per split
  if (s > smax) s = smax;
  if (s < 1) s = 1;
  const long cmax = d->C1 > d->Cout ? d->C1 : d->Cout;
  const long per_img = (long)pl->ty * pl->cx;
  for (;;) {  // 32-bit byte offsets inside a split
    pl->chunks_per_split = (int)((chunks + s - 1) / s);
    const long imgs = ((long)pl->chunks_per_split + 1) / per_img + 2;
    if (imgs * d->Hs * d->Ws * cmax * 4 < (1L << 31)) break;
    if (pl->chunks_per_split == 1) return false;
    s *= 2;
  }
  pl->splits = (int)((chunks + pl->chunks_per_split - 1) / pl->chunks_per_split);
  return tiles * pl->splits < (1L << 31);
}

}  // namespace

bool rs_wgrad_f32_wino33_ok(const rs_conv_desc* d) {
  W33Plan pl;
  return rs_knobs().wgrad_f32_wino33 != 0 && w33_plan(d, &pl);
}

long rs_wgrad_f32_wino33_workspace_floats(const rs_conv_desc* d) {
  W33Plan pl;
  if (!w33_plan(d, &pl)) return 0;
  const long n = 16L * d->Cout * d->C1;
  return pl.splits * n + rs_reduce_scratch_floats(n, pl.splits) + n;  // partial tiles, the reduction's scratch, their sum
}

int rs_wgrad_f32_wino33_launch(const rs_conv_desc* d, const float* dy, const float* src, float* dw, float* workspace, hipStream_t s) {
  W33Plan pl;
  if (!w33_plan(d, &pl)) return RS_EINVAL;
  Wino33WgradArgs a;
  a.dy = dy;
  a.src = src;
  a.part = workspace;
  a.N = d->N;
  a.H = d->Hs;
  a.W = d->Ws;
  a.Cin = d->C1;
  a.Cout = d->Cout;
  a.TY = pl.ty;
  a.CX = pl.cx;
  a.nchunks = pl.nchunks;
  a.tiles_co = pl.tiles_co;
  a.tiles_ci = pl.tiles_ci;
  a.chunks_per_split = pl.chunks_per_split;
  a.div_tycx = rs_make_fastdiv((unsigned)(pl.ty * pl.cx));
  a.div_cx = rs_make_fastdiv((unsigned)pl.cx);
  const int grid = pl.tiles_co * pl.tiles_ci * pl.splits;
  if (pl.wide) conv_wgrad_wino33_f32<2, 2><<<grid, 512, 0, s>>>(a);
  else conv_wgrad_wino33_f32<1, 1><<<grid, 128, 0, s>>>(a);
  int rc = RS_LAUNCH_RESULT();
  if (rc) return rc;
  const long total = (long)d->Cout * d->C1, n = 16 * total;
  float* scratch = workspace + (long)pl.splits * n;
  float* sum = scratch + rs_reduce_scratch_floats(n, pl.splits);
  rc = rs_reduce_splits(workspace, sum, n, pl.splits, scratch, s);  // (fixed order: deterministic)
  if (rc) return rc;
  wino33_wgrad_finish_kernel<<<rs_cdiv(total, 256), 256, 0, s>>>(sum, dw, d->Cout, d->C1, total);
  return RS_LAUNCH_RESULT();
}

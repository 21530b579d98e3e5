// conv_wgrad_wino_f32.hip -- the fp32 weight gradient of DecoderBlock (reference robosat/unet.py:63-73: conv3x3(pad 1) over a
// nearest-x2 upsample of cat[skip, prev]; the filter gradient autograd synthesises under tools/train.py:186) in the WINOGRAD
// domain of the forward's F(2x2, 2x2) form (conv_wino_f32.hip).  Round 6.
//
// The phase form (conv_wgrad.hip, PHASE) reduces, per output parity p = (py, px) and 2x2 tap (r, s), dz's parity plane against the
// shifted source: 16 multiply-adds per source pixel, cin and cout.  The forward of one parity on a 2x2 tile of source positions is
//     Y = A^T [ U (.) V ] A,   U = G g G^T,  V = B^T d B   (d: the tile's 3x3 source patch, Y: its 2x2 outputs of parity p)
// so its filter gradient is
//     dU[xi] = sum over tiles of  Z[xi] * V[xi],   Z = A dY A^T   (3x3 from the tile's 2x2 dz values),   dg = G^T dU G
// -- nine [Cout x tiles] . [tiles x Cin] products per parity instead of sixteen [Cout x pixels] . [pixels x Cin] over four times
// as many rows: 9/16 of the phase form's multiply-adds, 1/4 of the reference-shape count.  fp32 only, for the reason given in
// conv_wino_f32.hip: here the matrix cores are the bound (the fp32 train step is 59 % MFMA-busy over ALL its convolutions on both
// streams); the transforms are 5 additions (Z) and 12 subtractions (V) per lane and k-step, next to 9 x 64 MFMA cycles.
//
// Mapping.  A block owns (parity, 32*WGM couts, 32*WGN cins, a split of the tile sequence) and all nine xi; a wave a 32 x 32
// sub-tile: nine accumulators of v_mfma_f32_32x32x2_f32 (144 registers), K = tiles, two per MFMA.
//   HBM -> LDS   LDS-DMA as in conv_wgrad_f32_dma.hip: whole [pixel] rows of the block's channels, each lane its own offset from
//                a gather table (tile -> its four dz pixels of parity p and its nine source pixels; -1 = zeros: out of the image,
//                odd edge, tail).  PK = 8 tiles per chunk: 32 dz rows + 72 source rows, two stages, two blocks per CU.
//   fragments    lane (i = lane & 31, k = lane >> 5) reads channel i of tile 2 s + k: four dwords of dz, nine of the source patch,
//                transforms them in registers and feeds the nine MFMAs of the k-step.
//   output       partial tiles [split][parity][xi][Cout][Cin]; wino_wgrad_finish_kernel sums the splits in a fixed order, applies
//                G^T . G and adds the (parity, tap) pairs that make up each 3x3 filter tap (combine_phase_wgrad_f32_kernel's rule).
#include "conv_wgrad_wino_f32.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));

struct WinoWgradArgs {
  const float* dz;    // [N][2 Hs][2 Ws][Cout]
  const float* src1;  // [N][Hs][Ws][C1]
  const float* src2;  // [N][Hs][Ws][C2] or null
  float* part;        // [splits][4][9][Cout][C1 + C2]
  int N, Hs, Ws, C1, C2, Cout;
  int TY, TX;              // tiles per image
  int ntiles;              // N * TY * TX
  int tiles_co, tiles_ci;  // block tiles
  int chunks_per_split;    // chunks of PK tiles
  rs_fastdiv div_tytx, div_tx;
};

// PK tiles per chunk, RING stages.  Shipped: <2, 2, 8, 3> / <1, 2, 8, 3>, four / two waves, two blocks per CU.  Measured beside them
// (profiles/r06/wino_wgrad.txt): two stages (same time: the DMA's latency is not the bound), an eight-wave block on 128 couts x 64
// cins with 16-tile chunks and one block per CU (-35 % LDS-DMA bytes per MFMA, half the barriers: same time), two stages at three
// blocks per CU for the two-wave block (slower alone, same in the step); knock-outs: 80 TF executed as built, 95 without the DMA
// pieces, 85 without the transforms, 107 without either -- the phase form's kernel runs the same shapes at 110.
template <int WGM, int WGN, int PK, int RING>
__global__ __launch_bounds__(64 * WGM * WGN, WGM * WGN == 8 ? 1 : 2) void conv_wgrad_wino_f32(const WinoWgradArgs p) {
  constexpr int NS = PK / 2;
  constexpr int NW = WGM * WGN;
  constexpr int BM = 32 * WGM, BN = 32 * WGN;
  constexpr int ROWA = BM * 4, ROWB = BN * 4;          // bytes per LDS row (one pixel)
  constexpr int PPA = ROWA / 16, PPB = ROWB / 16;      // 16-byte pieces per row
  constexpr int RIA = 1024 / ROWA, RIB = 1024 / ROWB;  // rows per DMA instruction
  constexpr int NRA = PK * 4, NRB = PK * 9;            // rows per chunk
  constexpr int IA = NRA / RIA, IB = NRB / RIB;        // DMA instructions per chunk
  static_assert(IA * RIA == NRA && IB * RIB == NRB, "whole instructions");
  constexpr int NI = (IA + IB + NW - 1) / NW;          // per wave
  constexpr int ABYTES = NRA * ROWA, BBYTES = NRB * ROWB;
  constexpr int BUF = ABYTES + BBYTES;
  constexpr int NT = NRA + NRB;  // table entries per chunk
  static_assert(RING * BUF + RING * NT * 4 <= (NW == 8 ? 160 : 80) * 1024, "one / two blocks per CU");
  static_assert(NT <= 64 * NW, "one table entry per thread");

  __shared__ __attribute__((aligned(16))) unsigned char smem[RING * BUF + RING * NT * 4];
  int* tabs = reinterpret_cast<int*>(smem + RING * BUF);  // [RING][NT]: dz rows (tile * 4 + 2 u + v), then source rows (tile * 9 + 3 r + c)

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WGN, wn = wave % WGN;

  int bid = rs_xcd_remap(blockIdx.x, gridDim.x);
  const int tco = bid % p.tiles_co;
  bid /= p.tiles_co;
  const int tci = bid % p.tiles_ci;
  bid /= p.tiles_ci;
  const int par = bid & 3, split = bid >> 2;
  const int py = par >> 1, px = par & 1;
  const int co0 = tco * BM, ci0 = tci * BN;
  const int Cin = p.C1 + p.C2;

  const float* src = p.src1;
  int Cs = p.C1, cs = ci0;
  if (ci0 >= p.C1) {  // (a tile never straddles the two concat sources: BN divides both)
    src = p.src2;
    Cs = p.C2;
    cs = ci0 - p.C1;
  }
  constexpr int CPK = PK / kWwPK;  // plan chunks per chunk
  const int total_chunks = (p.ntiles + PK - 1) / PK;
  const int chunk0 = split * (p.chunks_per_split / CPK);  // (the plan makes chunks_per_split a multiple of CPK)
  int chunk1 = chunk0 + p.chunks_per_split / CPK;
  if (chunk1 > total_chunks) chunk1 = total_chunks;
  const int Ho = 2 * p.Hs, Wo = 2 * p.Ws;
  const int n_first = (int)rs_div((unsigned)(chunk0 * PK), p.div_tytx);
  const long img = (long)p.Hs * p.Ws * Cs, dimg = (long)Ho * Wo * p.Cout;
  const __amdgpu_buffer_rsrc_t rsrc_dz = ww_make_rsrc(p.dz + n_first * dimg, (long)(p.N - n_first) * dimg * 4);
  const __amdgpu_buffer_rsrc_t rsrc_x = ww_make_rsrc(src + n_first * img, (long)(p.N - n_first) * img * 4);

  auto fill_table = [&](int chunk, int which) __attribute__((always_inline)) {
    if (tid < NT) {
      const bool isb = tid >= NRA;
      const int e = isb ? tid - NRA : tid;
      const int tl = isb ? e / 9 : e >> 2, sub = isb ? e - tl * 9 : e & 3;
      const int t = chunk * PK + tl;
      int pix = -1;
      if (t < p.ntiles) {
        const int n = (int)rs_div((unsigned)t, p.div_tytx);
        const int rem = t - n * p.TY * p.TX;
        const int ty = (int)rs_div((unsigned)rem, p.div_tx);
        const int tx = rem - ty * p.TX;
        if (isb) {
          const int r = sub / 3, c = sub - 3 * r;
          const int y = 2 * ty - 1 + py + r, x = 2 * tx - 1 + px + c;
          if ((unsigned)y < (unsigned)p.Hs && (unsigned)x < (unsigned)p.Ws) pix = ((n - n_first) * p.Hs + y) * p.Ws + x;
        } else {
          const int a = 2 * ty + (sub >> 1), b = 2 * tx + (sub & 1);
          if (a < p.Hs && b < p.Ws) pix = ((n - n_first) * Ho + 2 * a + py) * Wo + 2 * b + px;
        }
      }
      tabs[which * NT + tid] = pix;
    }
    rs_lds_writes_done();  // (read by other waves behind a later barrier, which hipcc emits bare: common.h)
  };

  // ---- DMA roles: instruction ii = wave + NW j; ii < IA copies dz rows RIA ii.., else source rows RIB (ii - IA)..
  const int ra_a = lane / PPA, pp_a = lane % PPA;
  const int ra_b = lane / PPB, pp_b = lane % PPB;
  const int cola = (co0 + pp_a * 4) * 4, colb = (cs + pp_b * 4) * 4;
  const int cout4 = p.Cout * 4, cs4 = Cs * 4;
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
        voff[j] = pix >= 0 ? pix * cs4 + colb : -1;
      }
    }
  };
  auto issue_piece = [&](int j) __attribute__((always_inline)) {
    const int ii = wave + NW * j;
#if defined(RS_WW_KO) && (RS_WW_KO & 1)
    return;  // (knock-out build: no DMA in the steady state -- wrong results by construction)
#endif
    if (ii < IA) ww_dma16(rsrc_dz, fL + ii * 1024, voff[j]);
    else if (ii < IA + IB) ww_dma16(rsrc_x, fL + ABYTES + (ii - IA) * 1024, voff[j]);
  };

  f32x16 acc[9];
#pragma unroll
  for (int x = 0; x < 9; ++x)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[x][r] = 0.f;

  // fragment addressing: k-step s, lane (i = lane & 31, k = lane >> 5): tile 2 s + k, channel (sub-tile base + i)
  const int arow = (lane >> 5) * 4 * ROWA + (wm * 32 + (lane & 31)) * 4;
  const int brow = ABYTES + (lane >> 5) * 9 * ROWB + (wn * 32 + (lane & 31)) * 4;
  auto read_frag = [&](const unsigned char* L, int s, float (&dzv)[4], float (&d)[9]) __attribute__((always_inline)) {
#pragma unroll
    for (int q = 0; q < 4; ++q) dzv[q] = *reinterpret_cast<const float*>(L + arow + (8 * s + q) * ROWA);
#pragma unroll
    for (int q = 0; q < 9; ++q) d[q] = *reinterpret_cast<const float*>(L + brow + (18 * s + q) * ROWB);
  };
  auto kstep = [&](const float (&dzv)[4], const float (&d)[9], auto issue, int s) __attribute__((always_inline)) {
    // Z = A dY A^T (A = [1 0; 1 1; 0 1]); V = B^T d B (B^T = [1 -1 0; 0 1 0; 0 -1 1])
    float Z[9], V[9], T[9];
#if defined(RS_WW_KO) && (RS_WW_KO & 2)
#pragma unroll
    for (int x = 0; x < 9; ++x) {  // (knock-out build: no transforms)
      issue(s * 9 + x);
      acc[x] = __builtin_amdgcn_mfma_f32_32x32x2f32(dzv[x & 3], d[x], acc[x], 0, 0, 0);
    }
    return;
#endif
    const float t10 = dzv[0] + dzv[2], t11 = dzv[1] + dzv[3];
    Z[0] = dzv[0];
    Z[1] = dzv[0] + dzv[1];
    Z[2] = dzv[1];
    Z[3] = t10;
    Z[4] = t10 + t11;
    Z[5] = t11;
    Z[6] = dzv[2];
    Z[7] = dzv[2] + dzv[3];
    Z[8] = dzv[3];
#pragma unroll
    for (int r = 0; r < 3; ++r) {
      T[3 * r + 0] = d[3 * r + 0] - d[3 * r + 1];
      T[3 * r + 1] = d[3 * r + 1];
      T[3 * r + 2] = d[3 * r + 2] - d[3 * r + 1];
    }
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      V[c] = T[c] - T[3 + c];
      V[3 + c] = T[3 + c];
      V[6 + c] = T[6 + c] - T[3 + c];
    }
#pragma unroll
    for (int x = 0; x < 9; ++x) {
      issue(s * 9 + x);
      acc[x] = __builtin_amdgcn_mfma_f32_32x32x2f32(Z[x], V[x], acc[x], 0, 0, 0);
    }
  };
  constexpr int NMMA = NS * 9;
  constexpr int PSTEP = NMMA / (2 * NI) >= 1 ? NMMA / (2 * NI) : 1;  // front-loaded: the chunk's tail covers the latency
  auto chunk_mma = [&](const unsigned char* L, bool fetch) __attribute__((always_inline)) {
    float dzv[2][4], d[2][9];
    read_frag(L, 0, dzv[0], d[0]);
#pragma unroll
    for (int s = 0; s < NS; ++s) {
      if (s + 1 < NS) read_frag(L, s + 1, dzv[(s + 1) & 1], d[(s + 1) & 1]);
      kstep(dzv[s & 1], d[s & 1], [&](int q) __attribute__((always_inline)) {
        if (fetch && q % PSTEP == 0 && q / PSTEP < NI) issue_piece(q / PSTEP);
      }, s);
    }
    if (fetch) {
#pragma unroll
      for (int q = (NMMA + PSTEP - 1) / PSTEP; q < NI; ++q) issue_piece(q);
    }
  };

  if (chunk0 < chunk1) {
    // chunk c lives in stage / table (c - chunk0) % RING.  Iteration c runs chunk c's MFMAs with the pieces of chunk c + RING - 1 issued
    // between them (into the stage chunk c - 1 left at the last barrier), writes the table of chunk c + RING (over chunk c's, read
    // RING - 1 barriers ago) and waits for chunk c + 1's pieces: all but this wave's (RING - 2) NI youngest.
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
    int st = 0;  // (c - chunk0) % RING
    for (int c = chunk0; c < chunk1; ++c) {
      const int stf = st == 0 ? RING - 1 : st - 1;  // stage / table of chunk c + RING - 1
      const bool fetch = c + RING - 1 < chunk1;     // (block-uniform)
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

  // D[i][j]: i = cout (tile-local) = (r & 3) + 8 (r >> 2) + 4 (lane >> 5), j = cin (tile-local) = lane & 31
  float* out = p.part + ((long)(split * 4 + par) * 9) * p.Cout * Cin;
  const int ci = ci0 + wn * 32 + (lane & 31);
#pragma unroll
  for (int x = 0; x < 9; ++x)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int co = co0 + wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
      out[((long)x * p.Cout + co) * Cin + ci] = acc[x][r];
    }
}

// part [splits][4][9][Cout][Cin] -> dW [Cout][3][3][Cin]: splits summed in order, dg = G^T dU G per parity (G = [1 0; 1 1; 0 1]),
// then the (parity, tap) pairs of each filter tap: ky 0 -> (py 0, r 0), (1, 0); 1 -> (0, 1), (1, 0); 2 -> (0, 1), (1, 1); same in x
__global__ void wino_wgrad_finish_kernel(const float* __restrict__ part, float* __restrict__ dw, int splits, int Cout, int Cin, long total) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;  // over [Cout][Cin]
  if (i >= total) return;
  const int ci = (int)(i % Cin);
  const long co = i / Cin;
  const long plane = (long)Cout * Cin, sstride = 36 * plane;
  float g[4][2][2];
#pragma unroll
  for (int par = 0; par < 4; ++par) {
    float u[9];
#pragma unroll
    for (int x = 0; x < 9; ++x) {
      const float* q = part + (long)(par * 9 + x) * plane + co * Cin + ci;
      float s = 0.f;
      for (int k = 0; k < splits; ++k) s += q[k * sstride];
      u[x] = s;
    }
    g[par][0][0] = (u[0] + u[1]) + (u[3] + u[4]);
    g[par][0][1] = (u[1] + u[2]) + (u[4] + u[5]);
    g[par][1][0] = (u[3] + u[4]) + (u[6] + u[7]);
    g[par][1][1] = (u[4] + u[5]) + (u[7] + u[8]);
  }
#pragma unroll
  for (int ky = 0; ky < 3; ++ky)
#pragma unroll
    for (int kx = 0; kx < 3; ++kx) {
      const int ya[2][2] = {{0, ky == 0 ? 0 : 1}, {1, ky == 2 ? 1 : 0}};
      const int xa[2][2] = {{0, kx == 0 ? 0 : 1}, {1, kx == 2 ? 1 : 0}};
      float acc = 0.f;
#pragma unroll
      for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b) acc += g[2 * ya[a][0] + xa[b][0]][ya[a][1]][xa[b][1]];
      dw[(co * 9 + ky * 3 + kx) * Cin + ci] = acc;
    }
}

struct WwPlan {
  int wgm, wgn, tiles_co, tiles_ci, splits, chunks_per_split, ntiles, ty, tx;
};

bool ww_plan(const rs_conv_desc* d, WwPlan* pl) {
  if (!d || d->stem || d->ups != 1 || d->kh != 3 || d->kw != 3 || d->stride != 1 || d->pad != 1 || d->Ho != 2 * d->Hs || d->Wo != 2 * d->Ws)
    return false;
  if (d->N <= 0 || d->Hs < 2 || d->Ws < 2 || d->C1 <= 0 || d->C2 < 0 || d->Cout <= 0 || (d->Cout % 32) || (d->C1 % 64) || (d->C2 % 64)) return false;
  pl->wgn = 2;  // 64 cins per block (BN divides both concat sources)
  pl->wgm = (d->Cout % 64 == 0) ? 2 : 1;
  pl->tiles_co = d->Cout / (32 * pl->wgm);
  pl->tiles_ci = (d->C1 + d->C2) / 64;
  pl->ty = (d->Hs + 1) / 2;
  pl->tx = (d->Ws + 1) / 2;
  const long nt = (long)d->N * pl->ty * pl->tx;
  if (nt >= (1L << 28)) return false;
  pl->ntiles = (int)nt;
  const long chunks = (nt + kWwPK - 1) / kWwPK;
  const long tiles = 4L * pl->tiles_co * pl->tiles_ci;
  const long target = rs_knobs().wgrad_f32_wino_blocks;
  long s = (target + tiles - 1) / tiles;
  const long smax = (chunks + 15) / 16;  // at least 16 chunks (128 tiles) per split
  if (s > smax) s = smax;
  if (s < 1) s = 1;
  const long cmax = d->C1 > d->C2 ? d->C1 : d->C2;
  const long per_img = (long)pl->ty * pl->tx;
  for (;;) {  // 32-bit byte offsets inside a split (+1 chunk: the table runs one ahead)
    pl->chunks_per_split = (int)((chunks + s - 1) / s);
    const long imgs = ((long)pl->chunks_per_split + 1) * kWwPK / per_img + 2;
    const long span_dz = imgs * 4 * d->Hs * d->Ws * d->Cout * 4, span_x = imgs * d->Hs * d->Ws * cmax * 4;
    if (span_dz < (1L << 31) && span_x < (1L << 31)) break;
    if (pl->chunks_per_split == 1) return false;
    s *= 2;
  }
  pl->splits = (int)((chunks + pl->chunks_per_split - 1) / pl->chunks_per_split);
  if (4L * pl->tiles_co * pl->tiles_ci * pl->splits >= (1L << 31)) return false;
  return true;
}

}  // namespace

// Whether rs_conv2d_wgrad takes this launch through the Winograd domain (knob wgrad_f32_wino; geometry only, never the batch size
// beyond the 32-bit offset limits: the two forms differ in fp32 summation order).
bool rs_wgrad_f32_wino_ok(const rs_conv_desc* d) {
  WwPlan pl;
  return rs_knobs().wgrad_f32_wino != 0 && ww_plan(d, &pl);
}

long rs_wgrad_f32_wino_workspace_floats(const rs_conv_desc* d) {
  WwPlan pl;
  if (!ww_plan(d, &pl)) return 0;
  const long n = 36L * d->Cout * (d->C1 + d->C2);
  return pl.splits * n + rs_reduce_scratch_floats(n, pl.splits) + n;  // partial tiles, the reduction's scratch, their sum
}

int rs_wgrad_f32_wino_launch(const rs_conv_desc* d, const float* dz, const float* src1, const float* src2, float* dw, float* workspace,
                             hipStream_t s) {
  WwPlan pl;
  if (!ww_plan(d, &pl)) return RS_EINVAL;
  WinoWgradArgs a;
  a.dz = dz;
  a.src1 = src1;
  a.src2 = src2;
  a.part = workspace;
  a.N = d->N;
  a.Hs = d->Hs;
  a.Ws = d->Ws;
  a.C1 = d->C1;
  a.C2 = d->C2;
  a.Cout = d->Cout;
  a.TY = pl.ty;
  a.TX = pl.tx;
  a.ntiles = pl.ntiles;
  a.tiles_co = pl.tiles_co;
  a.tiles_ci = pl.tiles_ci;
  a.chunks_per_split = pl.chunks_per_split;
  a.div_tytx = rs_make_fastdiv((unsigned)(pl.ty * pl.tx));
  a.div_tx = rs_make_fastdiv((unsigned)pl.tx);
  const int grid = 4 * pl.tiles_co * pl.tiles_ci * pl.splits;
  if (pl.wgm == 2) conv_wgrad_wino_f32<2, 2, kWwPK, 3><<<grid, 256, 0, s>>>(a);
  else conv_wgrad_wino_f32<1, 2, kWwPK, 3><<<grid, 128, 0, s>>>(a);
  int rc = RS_LAUNCH_RESULT();
  if (rc) return rc;
  const long total = (long)d->Cout * (d->C1 + d->C2), n = 36 * total;
  float* scratch = workspace + (long)pl.splits * n;
  float* sum = scratch + rs_reduce_scratch_floats(n, pl.splits);
  rc = rs_reduce_splits(workspace, sum, n, pl.splits, scratch, s);  // (fixed order: deterministic)
  if (rc) return rc;
  wino_wgrad_finish_kernel<<<rs_cdiv(total, 256), 256, 0, s>>>(sum, dw, 1, d->Cout, d->C1 + d->C2, total);
  return RS_LAUNCH_RESULT();
}

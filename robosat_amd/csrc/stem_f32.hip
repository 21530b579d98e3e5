// stem_f32.hip -- resnet.conv1 (7x7 / stride 2 / pad 3, 3 or 4 bands -> 64; reference robosat/unet.py:122) in exact fp32 on
// v_mfma_f32_32x32x2_f32, every operand read from LDS.
//
// The implicit-GEMM stem this replaces walked K as 7 filter rows x (8 taps x 4 channels) = 224 with 147 of them real (the 8th
// tap and, for RGB, the 4th band are zeros), re-gathered every output pixel's 128-byte row from global memory once per filter
// row, and ran at 60 % of the fp32 MFMA peak on those padded FLOPs (0.32 ms of the 10.15 ms bs-16 predict pass).  Here:
//
//   tile   = 128 consecutive output pixels of ONE output row x all 64 couts; 4 waves as 2 (pixel halves) x 2 (cout halves), each
//            2 x 1 MFMA tiles of 32 x 32.  Blocks are persistent (two per CU) and own a CONTIGUOUS run of tiles: consecutive
//            tiles walk down the image, so five of a tile's seven input rows were fetched by the same CU's previous tile (its
//            XCD's L2 has them) and the filter is staged once per block, not once per tile.
//   strip  = the tile's receptive field: 7 input rows x 262 columns, fetched ONCE per tile with 16-byte buffer loads (out of
//            range = zeros = the padding) and written to LDS de-interleaved by column parity with an ODD pixel stride
//            (3 floats for RGB -- the zero band is dropped -- 5 for four bands): output pixel i's tap s sits at
//            [row][s & 1][i + (s >> 1)], so the 32 lanes of an MFMA operand read addresses 3i + const (5i + const): every
//            bank once.  The next tile's strip is fetched into registers before the current tile's MFMAs and written after its
//            epilogue.
//   K      = 75 k-steps of two (RGB; 100 for four bands) instead of 112: lanes 0-31 / 32-63 of an MFMA supply k and k + 1,
//            and the two must differ by a CONSTANT LDS offset for the step's address to be `lane base + immediate`: filter rows
//            (0,1) (2,3) (4,5) pair up (offset = one strip row), row 6 pairs its even taps with the odd ones (offset = the
//            parity plane), its last tap with a zero line.  150 real + 3 padding k against 224.
//   filter = [k-step][half][cout] in LDS (38 KB), re-ordered from the packed [64][7][8][4] layout once per block.
//   store  = as every convolution here: accumulators through LDS as [pixel][cout] (the staging tile aliases the strip), read back
//            row-wise into registers; then the NEXT strip is written and only after that the 16-byte stores go out, with the folded
//            BatchNorm scale / shift + ReLU (predict) or raw (train: bn_train_stats reads it).
//
// Measured (profiles/r06/stem_f32.txt; bs 16, 512^2): 319 -> 187-195 us, 68 % MFMA busy.  In-kernel stamps (-DRS_STEM_TRACE,
// scripts/debug/stem_trace.py): a tile is 9 700 cycles of MFMAs + 2 200 of everything else, and the CU's two blocks ALTERNATE rather than
// overlap -- a block's VALU work (epilogue arithmetic, fetch addresses) crawls while the other block's waves stream MFMAs on the same
// SIMDs (its LDS traffic and barriers do not); s_setprio, s_sleep between MFMA pairs and a staggered start changed nothing.
#include "common.h"

namespace {

struct StemArgs {
  const float* x;      // [N][H][W][4]
  const float* w;      // packed [64][7][8][4] (rs_pack_stem_weight)
  const float* scale;  // optional per-cout
  const float* shift;
  float* out;          // [N][Ho][Wo][64]
  int N, H, W, Ho, Wo, relu;
  int tpr;    // tiles per output row
  int ntile;  // N * Ho * tpr
#ifdef RS_STEM_TRACE
  long long* trace;  // [block][tile of the block][8] shader-clock stamps of wave 0 (measurement build: scripts/debug/stem_trace.py)
#endif
};

constexpr int SBM = 128, SBN = 64;  // tile: pixels x couts
constexpr int SCOLS = 2 * SBM + 6;  // strip columns: 2 * 127 + 7 taps, + 1 so that both parity planes hold XI_USED entries
constexpr int XI = 132;             // entries per parity plane (131 used)
constexpr int LDO = SBN + 4;        // staging row (floats)

template <int CIN>
struct Geom {
  static constexpr int PS = CIN == 3 ? 3 : 5;       // floats per strip pixel: odd => conflict-free operand reads
  static constexpr int ROWF = 2 * XI * PS;          // floats per strip row (two parity planes)
  static constexpr int STRIPF = 7 * ROWF;
  static constexpr int PAIR = 7 * CIN;              // k-steps per pair of filter rows
  static constexpr int NPAIR = 3 * PAIR;            // ... of rows (0,1) (2,3) (4,5)
  static constexpr int NEO = 3 * CIN;               // row 6, taps (0,1) (2,3) (4,5)
  static constexpr int NSTEP = NPAIR + NEO + CIN;   // + tap 6 | zero line
  static constexpr int WF = NSTEP * 2 * SBN;        // filter floats in LDS
  static constexpr int STAGEF = SBM * LDO;
  static constexpr int REGION = STRIPF > STAGEF ? STRIPF : STAGEF;  // the staging tile aliases the strip
  static constexpr int ZEROF = SBM * PS + 8;        // the zero line tap 7 of row 6 reads
  static constexpr int LDSF = WF + REGION + ZEROF;
};

// k-step t, lane half h -> filter element (r, s, c); s = 7: the zero tap (packed filters hold 0 there)
struct Tap {
  int r, s, c;
};
template <int CIN>
__host__ __device__ constexpr Tap stem_tap(int t, int h) {
  using G = Geom<CIN>;
  if (t < G::NPAIR) return Tap{2 * (t / G::PAIR) + h, (t % G::PAIR) / CIN, t % CIN};
  const int u = t - G::NPAIR;
  return Tap{6, 2 * (u / CIN) + h, u % CIN};
}
// ... and the step's immediate strip offset (floats) for the lanes of half 0; half 1 adds the group's constant
template <int CIN>
__host__ __device__ constexpr int stem_imm(int t) {
  using G = Geom<CIN>;
  const Tap k = stem_tap<CIN>(t, 0);
  return ((k.r * 2 + (k.s & 1)) * XI + (k.s >> 1)) * G::PS + k.c;
}

__device__ __forceinline__ __amdgpu_buffer_rsrc_t stem_rsrc(const float* base, long bytes) {
  const unsigned int n = bytes > 0x7FFFFFF0L ? 0x7FFFFFF0u : (unsigned int)(bytes < 0 ? 0 : bytes);
  const unsigned long b = (unsigned long)base;
  const unsigned int lo = __builtin_amdgcn_readfirstlane((unsigned int)b), hi = __builtin_amdgcn_readfirstlane((unsigned int)(b >> 32));
  return __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<void*>(((unsigned long)hi << 32) | lo), 0,
                                           (int)__builtin_amdgcn_readfirstlane(n), 0x00020000);
}

template <int CIN>
__global__ __launch_bounds__(256, CIN == 3 ? 2 : 1) void stem_conv_f32(const StemArgs p) {
  using G = Geom<CIN>;
  constexpr int PS = G::PS;
  constexpr int NPF = (7 * SCOLS + 255) / 256;  // strip pixels per thread
  static_assert(G::LDSF * 4 <= (CIN == 3 ? 80 : 160) * 1024, "LDS");

  __shared__ __attribute__((aligned(16))) float lds[G::LDSF];
  float* const wl = lds;                        // [NSTEP][2][64]
  float* const strip = lds + G::WF;             // [7][2][XI][PS]; the staging tile [128][LDO] after the MFMAs
  float* const zline = lds + G::WF + G::REGION;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int li = lane & 31, lh = lane >> 5;

#ifdef RS_STEM_TRACE
  if (tid == 0) p.trace[((long)blockIdx.x * 64 + 63) * 8 + 0] = (long long)__builtin_amdgcn_s_memtime();
#endif
  // this block's run of tiles
  const int t0 = (int)((long)blockIdx.x * p.ntile / gridDim.x);
  const int t1 = (int)((long)(blockIdx.x + 1) * p.ntile / gridDim.x);
  if (t0 >= t1) return;

  // ---- filter: packed [64][7][8][4] -> [k-step][half][cout]; the zero line.  Fourteen coalesced 16-byte loads per thread, all in
  //      flight together, each scattered to its k-step (the first version walked the LDS layout and gathered 4 bytes at a time: 38
  //      dependent L2 round trips per thread in front of the block's first tile)
  {
    constexpr int NW4 = SBN * 7 * 8 / 256;  // float4 per thread
    f32x4 wv[NW4];
#pragma unroll
    for (int k = 0; k < NW4; ++k) wv[k] = *reinterpret_cast<const f32x4*>(p.w + (tid + 256 * k) * 4);
#pragma unroll
    for (int k = 0; k < NW4; ++k) {
      const int e = tid + 256 * k;  // (co, r, s)
      const int s_ = e & 7, r = (e >> 3) % 7, co = e / 56;
      int t, h;
      if (r < 6) {
        t = (r >> 1) * G::PAIR + s_ * CIN, h = r & 1;
        if (s_ == 7) continue;  // (rows 0..5 have no k-step for the zero tap)
      } else if (s_ < 6) {
        t = G::NPAIR + (s_ >> 1) * CIN, h = s_ & 1;
      } else {
        t = G::NPAIR + G::NEO, h = s_ & 1;  // tap 6 | the zero tap (its packed entries are zeros)
      }
#pragma unroll
      for (int c = 0; c < CIN; ++c) wl[((t + c) * 2 + h) * SBN + co] = wv[k][c];
    }
  }
  for (int e = tid; e < G::ZEROF; e += 256) zline[e] = 0.f;

  // ---- strip fetch: pixel e = tid + 256 q of the 7 x SCOLS field -> registers; written to LDS one tile later -----------------
  f32x4 pf[NPF];
  int pdst[NPF];  // LDS offset (floats) of strip pixel e, or -1 past the field
  int prow[NPF], pcol[NPF], prel[NPF];  // its (row, column) in the field and byte offset relative to the field's first pixel
#pragma unroll
  for (int q = 0; q < NPF; ++q) {
    const int e = tid + 256 * q;
    const int r = e / SCOLS, j = e - r * SCOLS;
    const bool in = e < 7 * SCOLS;
    pdst[q] = in ? ((r * 2 + (j & 1)) * XI + (j >> 1)) * PS : -1;
    prow[q] = in ? r : -(1 << 24);  // (past the field: never inside the image)
    pcol[q] = j;
    prel[q] = (r * p.W + j) * 16;
  }
  const long img = (long)p.H * p.W * 4;
  // tile -> (image, output row, tile within the row), advanced tile by tile (no division in the loop)
  int cn, coy, ctx;
  {
    const int rowt = t0 / p.tpr;
    ctx = t0 - rowt * p.tpr;
    cn = rowt / p.Ho;
    coy = rowt - cn * p.Ho;
  }
  // (`live` = false: the same eight loads, all out of range -- the prefetch behind the block's last tile.  Unconditional on purpose: a
  // fetch under `if (tile + 1 < t1)` makes pf a merge of old and new values, hipcc then loads into temporaries and copies, and the copies
  // wait for the loads right where they were issued: the strip's round trip ended up in front of every tile's MFMAs.)
  auto fetch = [&](int n, int oy, int tx, bool live) __attribute__((always_inline)) {
    const __amdgpu_buffer_rsrc_t rs = stem_rsrc(p.x + n * img, live ? img * 4 : 0);
    const int iy0 = 2 * oy - 3, ix0 = 2 * tx * SBM - 3;
    const int base = (iy0 * p.W + ix0) * 16;
#pragma unroll
    for (int q = 0; q < NPF; ++q) {
      const bool ok = (unsigned)(iy0 + prow[q]) < (unsigned)p.H && (unsigned)(ix0 + pcol[q]) < (unsigned)p.W;
      pf[q] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, ok ? base + prel[q] : -1, 0, 0));
    }
  };
  auto put = [&]() __attribute__((always_inline)) {
#pragma unroll
    for (int q = 0; q < NPF; ++q) {
      if (pdst[q] >= 0) {
        float* d = strip + pdst[q];
        d[0] = pf[q][0];
        d[1] = pf[q][1];
        d[2] = pf[q][2];
        if (CIN == 4) d[3] = pf[q][3];
      }
    }
  };

  // operand bases (floats): pixel (wm * 64 + li) of the tile, + the half's constant of each group of k-steps
  const int apix = (wm * 64 + li) * PS;
  const float* const A1 = strip + apix + lh * G::ROWF;     // rows (2a, 2a + 1)
  const float* const A2 = strip + apix + lh * (XI * PS);   // row 6: taps (2b, 2b + 1)
  // row 6: tap 6 | the zero line.  Half 1 must land on zline + (pixel) * PS + c for the SAME immediate as half 0's tap 6:
  constexpr int IMM6 = ((6 * 2 + 0) * XI + 3) * PS;
  const float* const A3 = lh ? zline + apix - IMM6 : strip + apix;
  const float* const B = wl + lh * SBN + wn * 32 + li;

  // epilogue roles: 16 float4 per staged row, 16 rows per pass, 8 passes
  const int ecol = (tid & 15) * 4, erow = tid >> 4;
  f32x4 sc = {1.f, 1.f, 1.f, 1.f}, sh = {0.f, 0.f, 0.f, 0.f};
  if (p.scale) sc = *reinterpret_cast<const f32x4*>(p.scale + ecol);
  if (p.shift) sh = *reinterpret_cast<const f32x4*>(p.shift + ecol);

#ifdef RS_STEM_TRACE
#define STAMP(k) do { if (tid == 0) p.trace[((long)blockIdx.x * 64 + (tile - t0)) * 8 + (k)] = (long long)__builtin_amdgcn_s_memtime(); } while (0)
#else
#define STAMP(k)
#endif
#ifdef RS_STEM_TRACE
  if (tid == 0) p.trace[((long)blockIdx.x * 64 + 63) * 8 + 1] = (long long)__builtin_amdgcn_s_memtime();
#endif
  fetch(cn, coy, ctx, true);
  put();
  for (int tile = t0; tile < t1; ++tile) {
    STAMP(0);
    __syncthreads();  // the strip is in place
    STAMP(1);
    // the next tile's coordinates; its strip is in flight under the MFMAs
    int nn = cn, noy = coy, ntx = ctx + 1;
    if (ntx == p.tpr) {
      ntx = 0;
      if (++noy == p.Ho) noy = 0, ++nn;
    }
    fetch(nn, noy, ntx, tile + 1 < t1);
    STAMP(2);

    f32x16 acc0, acc1;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc0[r] = 0.f, acc1[r] = 0.f;
    // operands DEPTH steps ahead of their MFMAs (three ds_read_b32 per step, ~130 cycles of LDS latency against 128 cycles of MFMA per
    // step and wave: one step of distance -- what hipcc picks on its own -- leaves the latency exposed whenever the other block's
    // waves are reading too); sched_barrier pins the order
    constexpr int DEPTH = 3;
    float fb[DEPTH], fa0[DEPTH], fa1[DEPTH];
    auto frag = [&](int t) __attribute__((always_inline)) {
      const float* A = t < G::NPAIR ? A1 : (t < G::NPAIR + G::NEO ? A2 : A3);
      const int imm = stem_imm<CIN>(t);
      fb[t % DEPTH] = B[t * 2 * SBN];
      fa0[t % DEPTH] = A[imm];
      fa1[t % DEPTH] = A[imm + 32 * PS];
    };
#pragma unroll
    for (int t = 0; t < DEPTH; ++t) frag(t);
#pragma unroll
    for (int t = 0; t < G::NSTEP; ++t) {
      const float b = fb[t % DEPTH], a0 = fa0[t % DEPTH], a1 = fa1[t % DEPTH];
      __builtin_amdgcn_sched_barrier(0);
      acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(b, a0, acc0, 0, 0, 0);
      if (t + DEPTH < G::NSTEP) frag(t + DEPTH);
      acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(b, a1, acc1, 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
    }
    STAMP(3);
    __syncthreads();  // everybody is done with the strip: the staging tile takes its place
    STAMP(4);

    // ---- epilogue: registers -> LDS [pixel][cout] -> registers, row-wise; then the NEXT strip goes to LDS and only after that the
    //      16-byte stores are issued: the strip's s_waitcnt vmcnt must not find this tile's stores in the queue in front of it (they
    //      count in vmcnt on gfx9: waiting for the prefetched loads would wait for the stores' round trip to HBM -- 13.5 us per tile
    //      against the MFMAs' 8 in the first version)
    {
      const int prow = wm * 64 + li;
      const int ccol = wn * 32 + 4 * lh;
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        *reinterpret_cast<f32x4*>(&strip[prow * LDO + ccol + 8 * g]) = f32x4{acc0[4 * g], acc0[4 * g + 1], acc0[4 * g + 2], acc0[4 * g + 3]};
        *reinterpret_cast<f32x4*>(&strip[(prow + 32) * LDO + ccol + 8 * g]) = f32x4{acc1[4 * g], acc1[4 * g + 1], acc1[4 * g + 2], acc1[4 * g + 3]};
      }
    }
    __syncthreads();
    f32x4 ov[SBM / 16];
#pragma unroll
    for (int i = 0; i < SBM / 16; ++i) ov[i] = *reinterpret_cast<const f32x4*>(&strip[(erow + 16 * i) * LDO + ecol]);
    __syncthreads();  // the staging tile has been read: the next strip may land
    STAMP(5);
    put();
    STAMP(6);
    {
      const int ox0 = ctx * SBM;
      const int live = p.Wo - ox0 < SBM ? p.Wo - ox0 : SBM;
      float* const orow = p.out + (((long)cn * p.Ho + coy) * p.Wo + ox0) * SBN + ecol;
#pragma unroll
      for (int i = 0; i < SBM / 16; ++i) {
        const int row = erow + 16 * i;
        f32x4 v = ov[i];
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = v[e] * sc[e] + sh[e];
        if (p.relu) {
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], 0.f);
        }
        if (row < live) *reinterpret_cast<f32x4*>(orow + (long)row * SBN) = v;
      }
    }
    STAMP(7);
    cn = nn, coy = noy, ctx = ntx;
  }
#ifdef RS_STEM_TRACE
  if (tid == 0) p.trace[((long)blockIdx.x * 64 + 63) * 8 + 2] = (long long)__builtin_amdgcn_s_memtime();
#endif
}

}  // namespace

// conv_igemm.hip's rs_conv2d_fwd hands every stem launch here.  bands: 3 = the input's 4th band and the filter's c = 3 entries are
// zeros (RGB through rs_nchw_to_nhwc4 / rs_pack_stem_weight), anything else = four live bands.
__attribute__((visibility("hidden"))) int rs_stem_f32_launch(const rs_conv_desc* d, int bands, const float* x, const float* w,
                                                             const float* scale, const float* shift, float* out, void* stream) {
  if (d->kh != 7 || d->kw != 7 || d->stride != 2 || d->pad != 3 || d->Cout != SBN || d->C1 != 4) return RS_EINVAL;
  if (d->Ho != (d->Hs + 1) / 2 || d->Wo != (d->Ws + 1) / 2) return RS_EINVAL;
  if ((long)d->Hs * d->Ws * 16 >= (1L << 31)) return RS_EINVAL;  // 32-bit byte offsets within an image
  StemArgs a;
  a.x = x, a.w = w, a.scale = scale, a.shift = shift, a.out = out;
  a.N = d->N, a.H = d->Hs, a.W = d->Ws, a.Ho = d->Ho, a.Wo = d->Wo, a.relu = d->relu;
  a.tpr = rs_cdiv(d->Wo, SBM);
  const long ntile = (long)d->N * d->Ho * a.tpr;
  if (ntile <= 0 || ntile >= (1L << 31)) return RS_EINVAL;
  a.ntile = (int)ntile;
  int dev = 0, cus = 0;
  if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0)
    cus = 256;
  const long want = bands == 3 ? 2L * cus : cus;  // resident blocks: the LDS admits two (RGB) / one per CU
  const int grid = (int)(ntile < want ? ntile : want);
  if (bands == 3)
    stem_conv_f32<3><<<grid, 256, 0, (hipStream_t)stream>>>(a);
  else
    stem_conv_f32<4><<<grid, 256, 0, (hipStream_t)stream>>>(a);
  return RS_LAUNCH_RESULT();
}

#ifdef RS_STEM_TRACE
extern "C" int rs_stem_f32_trace(int N, int H, int W, const float* x, const float* w, float* out, long long* trace, int grid) {
  StemArgs a;
  a.x = x, a.w = w, a.scale = nullptr, a.shift = nullptr, a.out = out;
  a.N = N, a.H = H, a.W = W, a.Ho = H / 2, a.Wo = W / 2, a.relu = 1;
  a.tpr = rs_cdiv(a.Wo, SBM);
  a.ntile = N * a.Ho * a.tpr;
  a.trace = trace;
  stem_conv_f32<3><<<grid, 256, 0, 0>>>(a);
  return RS_LAUNCH_RESULT();
}
#endif

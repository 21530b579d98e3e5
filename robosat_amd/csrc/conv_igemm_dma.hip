// Implicit-GEMM convolution for gfx950 with LDS-DMA staging, templated on the activation type:
//   bf16_t : v_mfma_f32_32x32x16_bf16 (bf16 operands, fp32 accumulation; dense chip peak ~2.5 PFLOP/s): the bf16 training
//            path of BASELINE configs[2] (`rs train ... bf16`);
//   float  : v_mfma_f32_32x32x2_f32 (exact fp32, 157 TFLOP/s): every non-stem convolution of the fp32 parity path
//            (rs_conv2d_fwd; the 7x7 stem keeps the register-staged kernel of conv_igemm.hip).
// A K-chunk is always a 128-byte row per pixel (64 bf16 / 32 fp32 channels; 64-byte rows for the 32-channel bf16 layers),
// so both types share the LDS image, the swizzle and the fragment addressing; only the MFMA issue differs.
//
// Same operator as conv_igemm.hip (every nn.Conv2d / F.interpolate / torch.cat of UNet.forward, reference
// robosat/unet.py:122-141, and through `ups = 2` every data-gradient convolution of loss.backward(),
// tools/train.py:186), same GEMM view (M = N*Ho*Wo pixels, N = Cout, K = taps x Cin), same fused gather and
// epilogue.  What changes with 16x the MFMA rate is where the time goes, so the structure is re-balanced:
//
//   K-chunk = one filter tap x KC channels (KC = 64: 128-byte rows; KC = 32 for the 32-channel layers), i.e.
//            KC/16 MFMA k-steps per barrier.
//   gather  = the (tap, output pixel) -> source pixel map is computed ONCE per block into an LDS table
//            (taps x BM ints, -1 = padding / zero-insert hole / tail row); per chunk a thread fetches its rows'
//            entries with one ds_read and forms byte offsets with a multiply-add: the fp32 kernel's per-chunk
//            coordinate arithmetic (~10 VALU per load) would no longer hide under 8x shorter MFMA phases.
//   HBM->LDS = LDS-DMA (buffer_load_dwordx4 ... lds through SRSRC descriptors; offset -1 => the hardware writes zeros):
//            no VGPR round trip and no ds_write -- at bf16 MFMA rates the 32 KB per chunk a register-staged tile pushes
//            through ds_write_b128 (~79 B/clk/CU) costs as many LDS cycles as the MFMAs of the chunk take.  One wave
//            instruction = 1 KiB = 8 (16) whole rows, every lane with its own source offset (the gather).
//   LDS     = UNPADDED rows of KC bf16, 16-byte pieces XOR-swizzled: piece c of row r is stored at position
//            c ^ f(r), f(r) = (r>>1)&7 for 128-byte rows, (r>>2)&3 for 64-byte rows.  The DMA image is lane-linear
//            (lane l -> byte 16*l), so the swizzle is applied on the source side: lane l fetches piece (l % CPR) ^ f(row).
//            MFMA fragment reads (ds_read_b128: lane l reads row l&31, piece 2s + (l>>5)) hit 16 distinct 16-byte slots
//            in each of the instruction's four 16-lane groups.
//   MFMA    = 32x32x16: lane l feeds A[i = l&31][k = 8*(l>>5) .. +7] and B[k = 8*(l>>5) .. +7][j = l&31]: one b128
//            per 32-row sub-tile per k-step.  The weight fragment is the A operand: D[i = cout][j = pixel].
//   store   = accumulators -> LDS [pixel][cout] fp32 -> row-wise: 8 couts per thread, fp32 scale/shift, bf16 residual /
//            ReLU / ReLU-mask, one 16-byte bf16x8 store.
#include <stdlib.h>

#include "common.h"

namespace {

template <typename T>
struct ConvArgsT {
  const T* src1;
  const T* src2;
  const T* wgt;
  const float* scale;
  const float* shift;
  const T* res;
  const T* mask;
  T* out;
  float* stats;  // optional [M tiles][2][Cout]: per-tile sum / sum of squares of the STORED output (train-mode BatchNorm)
  // optional (data-gradient launches, with stats): the output is g = d loss / d z of a BatchNorm layer (ReLU mask
  // applied); the partial rows then hold sum g and sum g * xhat, xhat = (bn_y - bn_mean) * bn_invstd: the two reductions
  // of BatchNorm's backward, which otherwise cost a separate pass over (dz, z, y)
  const T* bn_y;
  const float* bn_mean;
  const float* bn_invstd;
  // optional second destination (the torch.cat split of a decoder data gradient): couts [0, csplit) go to `out` (row
  // stride csplit, `mask`), couts [csplit, Cout) to `out2` (row stride Cout - csplit, `mask2`); csplit % BN == 0
  T* out2;
  const T* mask2;
  int csplit;
  int N, Hs, Ws, C1, C2, Hv, Wv, ups;
  int kh, kw, stride, pad, Ho, Wo, Cout;
  int M, cpt, nk, Kw, relu, ntiles, ntaps, phase4, direct;
};

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

// 16 bytes of activations <-> fp32 lanes
template <typename T>
struct Piece;
template <>
struct Piece<float> {
  __device__ static __forceinline__ void load(const float* p, float (&v)[4]) {
    const f32x4 t = *reinterpret_cast<const f32x4*>(p);
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] = t[e];
  }
  __device__ static __forceinline__ void round(const float (&v)[4], float (&w)[4]) {
#pragma unroll
    for (int e = 0; e < 4; ++e) w[e] = v[e];
  }
  __device__ static __forceinline__ void store(float* p, const float (&v)[4]) {
    f32x4 t;
#pragma unroll
    for (int e = 0; e < 4; ++e) t[e] = v[e];
    *reinterpret_cast<f32x4*>(p) = t;
  }
};
template <>
struct Piece<bf16_t> {
  __device__ static __forceinline__ void load(const bf16_t* p, float (&v)[8]) {
    const bf16x8 t = *reinterpret_cast<const bf16x8*>(p);
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = (float)t[e];
  }
  __device__ static __forceinline__ void round(const float (&v)[8], float (&w)[8]) {
#pragma unroll
    for (int e = 0; e < 8; ++e) w[e] = (float)(bf16_t)v[e];
  }
  __device__ static __forceinline__ void store(bf16_t* p, const float (&v)[8]) {
    bf16x8 t;
#pragma unroll
    for (int e = 0; e < 8; ++e) t[e] = (bf16_t)v[e];
    *reinterpret_cast<bf16x8*>(p) = t;
  }
};

// one MFMA k-step on 16-byte operand fragments: D[i][j] += sum_k A[i][k] B[k][j]
__device__ __forceinline__ void mma16(f32x16& acc, const u32x4 a, const u32x4 b, bf16_t) {
  acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), acc, 0, 0, 0);
}
__device__ __forceinline__ void mma16(f32x16& acc, const u32x4 a, const u32x4 b, float) {
  const f32x4 fa = __builtin_bit_cast(f32x4, a), fb = __builtin_bit_cast(f32x4, b);
#pragma unroll
  for (int t = 0; t < 4; ++t) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[t], fb[t], acc, 0, 0, 0);
}

__device__ __forceinline__ __amdgpu_buffer_rsrc_t rb_make_rsrc(const void* base, long bytes) {
  const unsigned int n = bytes > 0xFFFFFFFEL ? 0xFFFFFFFEu : (unsigned int)(bytes < 0 ? 0 : bytes);
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, (int)n, 0x00020000);
}

// One LDS-DMA wave instruction (buffer_load_dwordx4 ... lds): lane l's 16 bytes at buffer offset `voff` land at LDS byte
// `lds_dst` + 16*l (lds_dst wave-uniform, in M0); offset -1 is out of range => zeros (scripts/probes/probe_glds.hip).
// Inline asm on purpose: through the builtin hipcc cannot tell that the DMA's destination (the OTHER pipeline buffer) is
// disjoint from the fragment reads that follow and drains the queue (s_waitcnt vmcnt(0)) before the first ds_read of
// every chunk.  As asm the copy is invisible to its counters, so the kernel waits itself (rb_dma_wait) ahead of the
// barrier that publishes the buffer.
__device__ __forceinline__ void rb_dma16(__amdgpu_buffer_rsrc_t r, unsigned int lds_dst, int voff) {
  unsigned int keep;
  asm volatile(
      "s_mov_b32 %0, m0\n\t"
      "s_mov_b32 m0, %2\n\t"
      "s_nop 0\n\t"
      "buffer_load_dwordx4 %1, %3, 0 offen lds\n\t"
      "s_mov_b32 m0, %0"
      : "=&s"(keep)
      : "v"(voff), "s"(lds_dst), "s"(r)
      : "memory");
}

__device__ __forceinline__ void rb_dma_wait() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
template <int N>
__device__ __forceinline__ void rb_dma_wait_n() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

__device__ __forceinline__ unsigned int rb_lds_addr(const void* p) {
  return (unsigned int)(unsigned long)(__attribute__((address_space(3))) const void*)p;
}

constexpr int kMaxK = 4;  // filter height / width up to 4 (the 4x4 stride-2 data gradient of an upsampled 3x3)

template <typename T, int BM, int BN, int WGM, int WGN, int ROWB, bool PHASE, int NBUF = 2>
__global__ __launch_bounds__(256, 2) void conv_igemm_dma(const ConvArgsT<T> p) {
  static_assert(NBUF == 2 || NBUF == 3, "2 or 3 pipeline buffers");
  static_assert(WGM * WGN == 4, "4 waves per block");
  static_assert(ROWB == 128 || ROWB == 64, "a K-chunk is a 128- or 64-byte row");
  constexpr int NW = 4;
  constexpr int ES = (int)sizeof(T);  // element size
  constexpr int EPP = 16 / ES;        // elements per 16-byte piece
  constexpr int KC = ROWB / ES;       // channels per chunk
  constexpr int WM = BM / WGM, WN = BN / WGN;
  constexpr int TM = WM / 32, TN = WN / 32;
  constexpr int CPR = ROWB / 16;      // 16-byte pieces per row
  constexpr int RI = 64 / CPR;        // rows per LDS-DMA wave instruction (1 KiB)
  constexpr int IA = BM / RI, IB = BN / RI;  // DMA instructions per chunk: pixel rows / weight rows
  constexpr int NI = (IA + IB + NW - 1) / NW;  // per wave
  constexpr int KS = CPR / 2;         // k-steps per chunk (two pieces each: lanes 0-31 / 32-63)
  constexpr int BUF = (BM + BN) * ROWB;  // bytes per pipeline buffer
  constexpr int LDO = BN + 4;         // epilogue staging row (floats)
  constexpr int PIPE = NBUF * BUF, STAGE = WGM * 32 * LDO * 4;  // staging: one 32-row sub-tile per wave row at a time
  constexpr int MAINB = PIPE > STAGE ? PIPE : STAGE;
  constexpr int TABN = (2 * kMaxK + 1) * BM;  // separable gather table (tap row | tap column) x tile row + output rows
  static_assert(TM >= 1 && TN >= 1, "wave tile must hold one 32x32 MFMA tile");
  static_assert((IA % NW) == 0 && IB >= 1 && (BM % RI) == 0 && (BN % RI) == 0, "DMA instruction split");
  static_assert(NBUF == 2 || ((IA + IB) % NW) == 0, "counted vmcnt waits need the same DMA count in every wave");

  __shared__ __attribute__((aligned(16))) unsigned char smem[MAINB + TABN * 4];
  int* taby = reinterpret_cast<int*>(smem + MAINB);  // [kh][BM]: ((n - nfirst)*Hs + iy) * Ws, or -1
  int* tabx = taby + kMaxK * BM;                     // [kw][BM]: ix, or -1
  int* orow = tabx + kMaxK * BM;                     // [BM]: output pixel index of the row, or -1 past M

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WGN, wn = wave % WGN;

  int bid = rs_xcd_remap(blockIdx.x, gridDim.x);
  // phase mode (p.phase4): conv3x3(pad 1) over a nearest-x2 upsampled source == four 2x2 convolutions on the SOURCE grid,
  // one per output parity (py, px), with pre-summed taps (rs_pack_phase_weight): 4/9 of the MACs and no duplicate
  // gathers.  Block -> (phase, tile); the problem rows m then enumerate SOURCE pixels (n, a, b) and the output row is
  // (n, 2a + py, 2b + px).
  int py = 0, px = 0;
  if (PHASE) {
    py = (bid >> 1) & 1;
    px = bid & 1;
    bid >>= 2;
  }
  const int mt = bid / p.ntiles, nt = bid - mt * p.ntiles;
  const int m0 = mt * BM, n0 = nt * BN;

  const int Hd = PHASE ? p.Hs : p.Ho, Wd = PHASE ? p.Ws : p.Wo;  // the grid the rows m enumerate
  const int HdWd = Hd * Wd;
  const int nfirst = m0 / HdWd;
  const int ush = p.ups ? 1 : 0;
  const int upar = p.ups == 2 ? 1 : 0;
  const int pad_y = PHASE ? 1 - py : p.pad, pad_x = PHASE ? 1 - px : p.pad;

  // ---- separable gather table, relative to the tile's first image: source pixel of (row, tap (r, s)) =
  //      taby[r][row] + tabx[s][row] when both are >= 0, else the tap contributes zeros ------------------------------
  for (int e = tid; e < (p.kh + p.kw + 1) * BM; e += 256) {
    const int t = e / BM, row = e - t * BM;
    const int m = m0 + row;
    int v = -1;
    if (m < p.M) {
      const int n = m / HdWd;
      const int rem = m - n * HdWd;
      const int oy = rem / Wd;
      const int ox = rem - oy * Wd;
      if (t < p.kh) {
        const int iy = oy * p.stride - pad_y + t;
        if (((unsigned)iy < (unsigned)p.Hv) && ((iy & upar) == 0)) v = ((n - nfirst) * p.Hs + (iy >> ush)) * p.Ws;
      } else if (t < p.kh + p.kw) {
        const int ix = ox * p.stride - pad_x + (t - p.kh);
        if (((unsigned)ix < (unsigned)p.Wv) && ((ix & upar) == 0)) v = ix >> ush;
      } else {
        v = PHASE ? (n * p.Ho + 2 * oy + py) * p.Wo + 2 * ox + px : m;
      }
    }
    if (t < p.kh) taby[t * BM + row] = v;
    else if (t < p.kh + p.kw) tabx[(t - p.kh) * BM + row] = v;
    else orow[row] = v;
  }

  const long img1 = (long)p.Hs * p.Ws * p.C1;
  const long img2 = (long)p.Hs * p.Ws * p.C2;
  const __amdgpu_buffer_rsrc_t rsrc1 = rb_make_rsrc(p.src1 + nfirst * img1, (long)(p.N - nfirst) * img1 * ES);
  const __amdgpu_buffer_rsrc_t rsrc2 = rb_make_rsrc(p.C2 ? p.src2 + nfirst * img2 : p.src1, (long)(p.N - nfirst) * img2 * ES);
  const __amdgpu_buffer_rsrc_t rsrcw =
      rb_make_rsrc(p.wgt + (long)(2 * py + px) * p.Cout * p.Kw, (long)p.Cout * p.Kw * ES);  // phase weights follow each other

  // ---- LDS-DMA roles.  Instruction ii = wave + 4j copies 1 KiB = RI whole rows: ii < IA pixel rows RI*ii.., else
  //      weight rows RI*(ii-IA)...  Lane l: row ra = l / CPR of the instruction, 16-byte position pp = l % CPR, which
  //      must receive channel piece pp ^ f(row) (the swizzle lives on the SOURCE address; the LDS image is lane-linear).
  const int ra = lane / CPR, pp = lane % CPR;
  const int fsw = ROWB == 128 ? ((4 * (wave & 1) + (ra >> 1)) & 7) : ((ra >> 2) & 3);  // f(RI*ii + ra): ii = wave (mod 2)
  const int gp = pp ^ fsw;
  const unsigned int lds0 = __builtin_amdgcn_readfirstlane(rb_lds_addr(smem));
  int wrow[NI];  // byte offset of this lane's piece in weight row (n0 + RI*jj + ra), chunk 0
#pragma unroll
  for (int j = 0; j < NI; ++j) wrow[j] = ((n0 + RI * (wave + NW * j - IA) + ra) * p.Kw + gp * EPP) * ES;
  __syncthreads();

  int lr = 0, ls = 0, lc = 0, lk = 0;  // next chunk to fetch: tap row / tap column / channel chunk / linear index
  auto issue_dma = [&](int buf) __attribute__((always_inline)) {
    const unsigned int L = lds0 + buf * BUF;
    const int c0 = lc * KC;
    const bool first = c0 < p.C1;
    const __amdgpu_buffer_rsrc_t rs = first ? rsrc1 : rsrc2;
    const int cs2 = (first ? p.C1 : p.C2) * ES;
    const int cb = ((first ? c0 : c0 - p.C1) + gp * EPP) * ES;
    int pix[NI];
#pragma unroll
    for (int j = 0; j < NI; ++j) {
      const int row = RI * (wave + NW * j) + ra;
      const int y = (NW * j < IA) ? taby[lr * BM + row] : 0, x = (NW * j < IA) ? tabx[ls * BM + row] : 0;
      pix[j] = (y | x) < 0 ? -1 : y + x;
    }
#pragma unroll
    for (int j = 0; j < NI; ++j) {
      const int ii = wave + NW * j;  // wave-uniform; IA % 4 == 0, so the role depends on j alone
      if (NW * j < IA) {
        rb_dma16(rs, L + ii * 1024, pix[j] >= 0 ? pix[j] * cs2 + cb : -1);
      } else if ((IB % NW) == 0 || ii < IA + IB) {
        rb_dma16(rsrcw, L + ii * 1024, wrow[j] + lk * ROWB);
      }
    }
    ++lk;  // advance to the following chunk
    ++lc;
    const int w1 = (lc == p.cpt) ? 1 : 0;
    lc = w1 ? 0 : lc;
    ls += w1;
    const int w2 = (ls == p.kw) ? 1 : 0;
    ls = w2 ? 0 : ls;
    lr += w2;
  };

  f32x16 acc[TN][TM];  // [cout sub-tile][pixel sub-tile]; D rows = couts, D cols = pixels
#pragma unroll
  for (int a = 0; a < TN; ++a)
#pragma unroll
    for (int b = 0; b < TM; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

  // fragment addressing: row (lane&31) of a 32-row sub-tile, piece 2s + (lane>>5), swizzled
  const int frow = lane & 31;
  const int fl = ROWB == 128 ? ((frow >> 1) & 7) : ((frow >> 2) & 3);
  int foff[KS];
#pragma unroll
  for (int s = 0; s < KS; ++s) foff[s] = ((2 * s + (lane >> 5)) ^ fl) * 16;
  const int abase = (wm * WM + frow) * ROWB;
  const int bbase = (BM + wn * WN + frow) * ROWB;

  auto read_frag = [&](const unsigned char* L, int s, u32x4 (&a)[TM], u32x4 (&b)[TN]) __attribute__((always_inline)) {
#pragma unroll
    for (int tm = 0; tm < TM; ++tm) a[tm] = *reinterpret_cast<const u32x4*>(L + abase + 32 * tm * ROWB + foff[s]);
#pragma unroll
    for (int tn = 0; tn < TN; ++tn) b[tn] = *reinterpret_cast<const u32x4*>(L + bbase + 32 * tn * ROWB + foff[s]);
  };
  auto mma_frag = [&](const u32x4 (&a)[TM], const u32x4 (&b)[TN]) __attribute__((always_inline)) {
#pragma unroll
    for (int tm = 0; tm < TM; ++tm)
#pragma unroll
      for (int tn = 0; tn < TN; ++tn) mma16(acc[tn][tm], b[tn], a[tm], T());
  };

  // ---- main loop: the chunks k+1 .. k+NBUF-1 stream HBM -> LDS by DMA while the MFMAs of chunk k run; one barrier per
  //      chunk.  Each wave first waits for ITS OWN DMA instructions of chunk k with a COUNTED s_waitcnt (loads complete in
  //      order: allowing NI*(chunks still in flight) outstanding == chunk k has landed), the barrier publishes everybody's,
  //      and only then the buffer freed by chunk k-1 is refilled.  With three buffers a chunk has two iterations to land:
  //      the short-K encoder layers, whose iteration is one DMA round trip and 8 MFMAs long, run ~1.5x faster per block.
  constexpr int AHEAD = NBUF - 1;
#pragma unroll
  for (int j = 0; j < AHEAD; ++j)
    if (j < p.nk) issue_dma(j);
  for (int kc = 0; kc < p.nk; ++kc) {
    if (NBUF == 3 && kc + 1 < p.nk) rb_dma_wait_n<NI>();  // chunk kc+1 may stay in flight
    else rb_dma_wait();
    __syncthreads();
    if (kc + AHEAD < p.nk) issue_dma((kc + AHEAD) % NBUF);  // (uniform) the buffer chunk kc-1 was read from
    const unsigned char* L = smem + (kc % NBUF) * BUF;
    u32x4 fa[2][TM], fb[2][TN];
    read_frag(L, 0, fa[0], fb[0]);
#pragma unroll
    for (int s = 0; s < KS; ++s) {
      if (s + 1 < KS) read_frag(L, s + 1, fa[(s + 1) & 1], fb[(s + 1) & 1]);
      mma_frag(fa[s & 1], fb[s & 1]);
    }
  }
  __syncthreads();  // every wave is done with the pipeline buffers: the epilogue stages through them

  // ---- direct epilogue (p.direct; no BatchNorm statistics): D[i = cout][j = pixel] puts 4 consecutive couts of one pixel
  //      in registers 4g..4g+3, so a lane can apply the epilogue and store them itself (16 bytes fp32 / 8 bytes bf16; the
  //      two half-waves fill 32 / 16 contiguous bytes per pixel row and L2 merges the rows across g, tn) -- no LDS round
  //      trip, no barrier.  Which of the two wins is shape dependent (measured; see pick_direct).
  if (p.direct) {
    const int hh = lane >> 5;
#pragma unroll
    for (int tm = 0; tm < TM; ++tm) {
      const int opix = orow[wm * WM + 32 * tm + (lane & 31)];
      if (opix < 0) continue;
#pragma unroll
      for (int tn = 0; tn < TN; ++tn)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const int c0 = n0 + wn * WN + 32 * tn + 8 * g + 4 * hh;
          const long o = (long)opix * p.Cout + c0;
          f32x4 v;
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const float a = acc[tn][tm][4 * g + e];
            v[e] = a * (p.scale ? p.scale[c0 + e] : 1.f) + (p.shift ? p.shift[c0 + e] : 0.f);
          }
          if (p.res) {
            const f32x4 r = rs_ld4(p.res + o);
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] += r[e];
          }
          if (p.relu) {
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], 0.f);
          }
          if (p.mask) {
            const f32x4 z = rs_ld4(p.mask + o);
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = z[e] > 0.f ? v[e] : 0.f;
          }
          rs_st4(p.out + o, v);
        }
    }
    return;
  }

  // ---- epilogue: registers -> LDS [pixel][cout] fp32 -> one 16-byte piece of couts per thread, row-wise stores.
  //      TM passes of WGM*32 rows each (pass t = sub-tile tm = t of every wave) keep the staging tile at
  //      WGM*32 x (BN+4) floats: the LDS footprint, hence the blocks per CU, is set by the pipeline buffers alone.
  float* lds = reinterpret_cast<float*>(smem);
  constexpr int TPR = BN / EPP;              // threads per row
  constexpr int RPI = 256 / TPR;             // rows per iteration
  const int cc = tid % TPR, rr = tid / TPR;
  const int col = n0 + cc * EPP;
  float sc[EPP], sh[EPP];
#pragma unroll
  for (int e = 0; e < EPP; ++e) {
    sc[e] = p.scale ? p.scale[col + e] : 1.f;
    sh[e] = p.shift ? p.shift[col + e] : 0.f;
  }
  T* outp = p.out;  // destination of this block's couts (block-uniform: a tile never straddles csplit)
  const T* maskp = p.mask;
  int ostride = p.Cout, ocol = col;
  if (p.out2) {
    if (n0 >= p.csplit) {
      outp = p.out2;
      maskp = p.mask2;
      ostride = p.Cout - p.csplit;
      ocol = col - p.csplit;
    } else {
      ostride = p.csplit;
    }
  }
  float st0[EPP], st1[EPP];  // BatchNorm statistics of this thread's rows (only when p.stats)
  float bmu[EPP], bis[EPP];
#pragma unroll
  for (int e = 0; e < EPP; ++e) {
    st0[e] = st1[e] = 0.f;
    bmu[e] = p.bn_y ? p.bn_mean[col + e] : 0.f;
    bis[e] = p.bn_y ? p.bn_invstd[col + e] : 0.f;
  }
#pragma unroll
  for (int tm = 0; tm < TM; ++tm) {
    if (tm > 0) __syncthreads();  // the previous pass has been read out
    {
      const int pr = wm * 32 + (lane & 31);  // pass-local row
      const int ccol = wn * WN + 4 * (lane >> 5);
#pragma unroll
      for (int tn = 0; tn < TN; ++tn)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          f32x4 v;
          v[0] = acc[tn][tm][4 * g + 0];
          v[1] = acc[tn][tm][4 * g + 1];
          v[2] = acc[tn][tm][4 * g + 2];
          v[3] = acc[tn][tm][4 * g + 3];
          *reinterpret_cast<f32x4*>(&lds[pr * LDO + ccol + 32 * tn + 8 * g]) = v;
        }
    }
    __syncthreads();
#pragma unroll 2
    for (int lrow = rr; lrow < WGM * 32; lrow += RPI) {
      const int row = (lrow >> 5) * WM + 32 * tm + (lrow & 31);  // tile row of pass-local row lrow
      const int opix = orow[row];
      if (opix >= 0) {
        const long o = (long)opix * ostride + ocol;
        float v[EPP];
#pragma unroll
        for (int h = 0; h < EPP / 4; ++h) {
          const f32x4 t = *reinterpret_cast<const f32x4*>(&lds[lrow * LDO + cc * EPP + 4 * h]);
#pragma unroll
          for (int e = 0; e < 4; ++e) v[4 * h + e] = t[e] * sc[4 * h + e] + sh[4 * h + e];
        }
        if (p.res) {
          float r[EPP];
          Piece<T>::load(p.res + o, r);
#pragma unroll
          for (int e = 0; e < EPP; ++e) v[e] += r[e];
        }
        if (p.relu) {
#pragma unroll
          for (int e = 0; e < EPP; ++e) v[e] = fmaxf(v[e], 0.f);
        }
        if (maskp) {
          float z[EPP];
          Piece<T>::load(maskp + o, z);
#pragma unroll
          for (int e = 0; e < EPP; ++e) v[e] = z[e] > 0.f ? v[e] : 0.f;
        }
        Piece<T>::store(outp + o, v);
        if (p.stats) {
          float w[EPP];
          Piece<T>::round(v, w);  // statistics of the values as stored (bf16-rounded on the bf16 path)
          if (p.bn_y) {
            float yv[EPP];
            Piece<T>::load(p.bn_y + o, yv);
#pragma unroll
            for (int e = 0; e < EPP; ++e) {
              st0[e] += w[e];
              st1[e] += w[e] * ((yv[e] - bmu[e]) * bis[e]);
            }
          } else {
#pragma unroll
            for (int e = 0; e < EPP; ++e) {
              st0[e] += w[e];
              st1[e] += w[e] * w[e];
            }
          }
        }
      }
    }
  }
  if (p.stats) {  // (uniform) block reduction over the RPI row lanes -> one partial row per M tile
    __syncthreads();
    float* r0 = lds;             // [RPI][BN]
    float* r1 = lds + RPI * BN;  // [RPI][BN]
#pragma unroll
    for (int e = 0; e < EPP; ++e) {
      r0[rr * BN + cc * EPP + e] = st0[e];
      r1[rr * BN + cc * EPP + e] = st1[e];
    }
    __syncthreads();
    if (tid < 2 * BN) {
      const float* r = tid < BN ? r0 : r1;
      const int c = tid < BN ? tid : tid - BN;
      float a = 0.f;
#pragma unroll 4
      for (int l = 0; l < RPI; ++l) a += r[l * BN + c];
      p.stats[((long)mt * 2 + (tid < BN ? 0 : 1)) * p.Cout + n0 + c] = a;
    }
  }
}

// fp32 KRSC [Cout][taps][Cin] -> bf16 data-gradient weights [Cin][taps][Cout], taps flipped (cf. conv_wgrad.hip)
__global__ void pack_dgrad_weight_bf16_kernel(const float* __restrict__ w, bf16_t* __restrict__ out, int Cout, int taps,
                                              int Cin) {
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
    if (ci < Cin && co < Cout) out[((long)ci * taps + ftap) * Cout + co] = (bf16_t)tile[tx][r];
  }
}

enum Tile { T128x128 = 0, T128x64, T128x32, T64x64, NTILES };
const char* const kTileNamesBf16[NTILES] = {"conv_igemm_bf16<128x128>", "conv_igemm_bf16<128x64>", "conv_igemm_bf16<128x32>",
                                            "conv_igemm_bf16<64x64>"};
const int kTileBM[NTILES] = {128, 128, 128, 64};
const int kTileBN[NTILES] = {128, 64, 32, 64};

bool valid(const rs_conv_desc* d) {
  if (!d) return false;
  if (d->N <= 0 || d->Hs <= 0 || d->Ws <= 0 || d->Ho <= 0 || d->Wo <= 0) return false;
  if (d->kh <= 0 || d->kw <= 0 || d->kh > kMaxK || d->kw > kMaxK || d->stride <= 0 || d->pad < 0) return false;
  if (d->Cout <= 0 || (d->Cout % 32) != 0) return false;
  if (d->ups < 0 || d->ups > 2 || d->stem) return false;
  if (d->C1 <= 0 || (d->C1 % 32) != 0 || d->C2 < 0 || (d->C2 % 32) != 0) return false;
  return true;
}

bool phase_ok(const rs_conv_desc* d) {
  return d->ups == 1 && d->kh == 3 && d->kw == 3 && d->stride == 1 && d->pad == 1 && d->Ho == 2 * d->Hs && d->Wo == 2 * d->Ws;
}

int pick_tile(const rs_conv_desc* d, bool phase4 = false) {
  const long M = phase4 ? (long)d->N * d->Hs * d->Ws * 4 : (long)d->N * d->Ho * d->Wo;  // (x4: the phases share the grid)
  const long want = 512;  // >= 2 blocks per CU
  if (d->Cout % 128 == 0 && (long)rs_cdiv(M, 128) * (d->Cout / 128) >= want) return T128x128;
  if (d->Cout % 64 == 0) {
    if ((long)rs_cdiv(M, 128) * (d->Cout / 64) >= want) return T128x64;
    return T64x64;
  }
  return T128x32;
}

// 128-byte rows (4 k-steps per barrier, 2 blocks per CU) or 64-byte rows (2 k-steps per barrier, 4 blocks per CU)?
// RS_CONV_ROWB=64|128 overrides (measurement knob).
int pick_rowb(const rs_conv_desc* d, int es, bool phase4 = false) {
  static const int forced = [] {
    const char* e = getenv("RS_CONV_ROWB");
    return e ? atoi(e) : 0;
  }();
  if (forced == 64 || forced == 128) return forced;
  // Measured on both paths (bs-32 bf16 train, bs-16 fp32 predict, per-layer A/B): 64-byte rows win when the grid can use the
  // doubled occupancy (>= 2048 blocks) and the K loop is short (<= 16 chunks of 128 bytes) -- the 1x1 convolutions at
  // 64^2..128^2 gain 25-35 % -- or, for fp32, at any K (the 64-cycle fp32 MFMAs hide the extra barriers); long-K layers
  // with few blocks (layer3/4, dec0/dec1) keep 128-byte rows (+8..20 % there).
  const int tile = pick_tile(d, phase4);
  const long blocks = (long)rs_cdiv((long)d->N * d->Ho * d->Wo, kTileBM[tile]) * (d->Cout / kTileBN[tile]);
  const long nk128 = (long)(phase4 ? 4 : d->kh * d->kw) * (d->C1 + d->C2) * es / 128;
  if (blocks >= 2048 && (nk128 <= 16 || es == 4)) return 64;
  return 128;
}

template <typename T, int ROWB, bool PHASE>
void launch(int tile, int grid, hipStream_t s, const ConvArgsT<T>& a) {
  // RS_CONV_NBUF=3 (measurement knob): three pipeline buffers with counted vmcnt waits for the 64-byte-row kernels (two
  // chunks in flight per block, but 3 instead of 4 blocks per CU).  Measured per layer on both paths: a wash to slightly
  // slower (fp32 predict 14.48 -> 14.79 ms of convolutions, bf16 train 17.5 -> 18.0) -- occupancy hides the DMA round trip
  // better than depth here -- so two buffers stay the default.
  static const int nbuf3 = [] {
    const char* e = getenv("RS_CONV_NBUF");
    return e && atoi(e) == 3;
  }();
  if (ROWB == 64 && nbuf3 && tile != T128x32) {
    switch (tile) {
      case T128x128: conv_igemm_dma<T, 128, 128, 2, 2, 64, PHASE, 3><<<grid, 256, 0, s>>>(a); break;
      case T128x64: conv_igemm_dma<T, 128, 64, 2, 2, 64, PHASE, 3><<<grid, 256, 0, s>>>(a); break;
      default: conv_igemm_dma<T, 64, 64, 2, 2, 64, PHASE, 3><<<grid, 256, 0, s>>>(a); break;
    }
    return;
  }
  switch (tile) {
    case T128x128: conv_igemm_dma<T, 128, 128, 2, 2, ROWB, PHASE><<<grid, 256, 0, s>>>(a); break;
    case T128x64: conv_igemm_dma<T, 128, 64, 2, 2, ROWB, PHASE><<<grid, 256, 0, s>>>(a); break;
    case T128x32: conv_igemm_dma<T, 128, 32, 4, 1, ROWB, PHASE><<<grid, 256, 0, s>>>(a); break;
    default: conv_igemm_dma<T, 64, 64, 2, 2, ROWB, PHASE><<<grid, 256, 0, s>>>(a); break;
  }
}

template <typename T>
int conv_fwd(const rs_conv_desc* d, const void* src1, const void* src2, const void* weight, const float* scale,
             const float* shift, const void* residual, const void* relu_mask, void* out, rs_stream_t stream,
             float* stats = nullptr, const void* bn_y = nullptr, const float* bn_mean = nullptr,
             const float* bn_invstd = nullptr, bool phase4 = false, void* out2 = nullptr, const void* mask2 = nullptr,
             int csplit = 0) {
  if (!valid(d) || !src1 || !weight || !out) return RS_EINVAL;
  if (out2 && (csplit <= 0 || csplit >= d->Cout || residual || stats)) return RS_EINVAL;
  if (phase4 && (!phase_ok(d) || stats)) return RS_EINVAL;
  if (d->C2 > 0 && !src2) return RS_EINVAL;
  constexpr long ES = (long)sizeof(T);
  ConvArgsT<T> a;
  a.src1 = reinterpret_cast<const T*>(src1);
  a.src2 = reinterpret_cast<const T*>(src2);
  a.wgt = reinterpret_cast<const T*>(weight);
  a.scale = scale;
  a.shift = shift;
  a.res = reinterpret_cast<const T*>(residual);
  a.mask = reinterpret_cast<const T*>(relu_mask);
  a.out = reinterpret_cast<T*>(out);
  a.stats = stats;
  a.bn_y = reinterpret_cast<const T*>(bn_y);
  a.bn_mean = bn_mean;
  a.bn_invstd = bn_invstd;
  a.out2 = reinterpret_cast<T*>(out2);
  a.mask2 = reinterpret_cast<const T*>(mask2);
  a.csplit = csplit;
  a.N = d->N;
  a.Hs = d->Hs;
  a.Ws = d->Ws;
  a.C1 = d->C1;
  a.C2 = d->C2;
  a.phase4 = phase4 ? 1 : 0;
  {
    static const int forced = [] {  // RS_CONV_DIRECT=0|1 overrides (measurement knob)
      const char* e = getenv("RS_CONV_DIRECT");
      return e ? atoi(e) : -1;
    }();
    a.direct = (forced >= 0 ? forced != 0 : false) && !stats;
  }
  a.ups = phase4 ? 0 : d->ups;  // phase mode gathers on the source grid itself
  a.Hv = a.ups == 0 ? d->Hs : (a.ups == 1 ? 2 * d->Hs : 2 * d->Hs - 1);
  a.Wv = a.ups == 0 ? d->Ws : (a.ups == 1 ? 2 * d->Ws : 2 * d->Ws - 1);
  a.kh = phase4 ? 2 : d->kh;
  a.kw = phase4 ? 2 : d->kw;
  a.stride = d->stride;
  a.pad = d->pad;
  a.Ho = d->Ho;
  a.Wo = d->Wo;
  a.Cout = d->Cout;
  if ((long)d->N * d->Ho * d->Wo >= (1L << 31)) return RS_EINVAL;
  const long M = phase4 ? (long)d->N * d->Hs * d->Ws : (long)d->N * d->Ho * d->Wo;  // rows per phase
  a.M = (int)M;
  {
    // 32-bit byte offsets relative to the first image of a tile (<= 128 rows: 128/(rows per image) + 2 images)
    const long cmax = d->C1 > d->C2 ? d->C1 : d->C2;
    const long img_bytes = (long)d->Hs * d->Ws * cmax * ES;
    const long rows_per_image = phase4 ? (long)d->Hs * d->Ws : (long)d->Ho * d->Wo;
    const long span = (128 / rows_per_image + 2) * img_bytes;
    if (span >= (1L << 31)) return RS_EINVAL;
    if ((long)d->Cout * d->kh * d->kw * (d->C1 + d->C2) * ES >= (1L << 31)) return RS_EINVAL;
  }
  // channels per K-chunk: a 128-byte row, or 64 bytes (the 32-channel bf16 layers; also half the LDS: 4 blocks per CU
  // instead of 2, which is what the short-K layers want -- see pick_rowb)
  const int kc128 = 128 / (int)ES;
  const bool can128 = d->C1 % kc128 == 0 && d->C2 % kc128 == 0;
  const int kc = (can128 && pick_rowb(d, (int)ES, phase4) == 128) ? kc128 : kc128 / 2;
  a.cpt = (d->C1 + d->C2) / kc;
  a.ntaps = a.kh * a.kw;
  a.nk = a.ntaps * a.cpt;
  a.Kw = a.nk * kc;
  a.relu = d->relu;

  const int tile = pick_tile(d, phase4);
  if (out2 && (csplit % kTileBN[tile]) != 0) return RS_EINVAL;
  a.direct = a.direct && !out2;
  a.ntiles = d->Cout / kTileBN[tile];
  const int grid = rs_cdiv(M, kTileBM[tile]) * a.ntiles * (phase4 ? 4 : 1);
  hipStream_t s = (hipStream_t)stream;
  if (phase4) {
    if (kc == kc128) launch<T, 128, true>(tile, grid, s, a);
    else launch<T, 64, true>(tile, grid, s, a);
  } else {
    if (kc == kc128) launch<T, 128, false>(tile, grid, s, a);
    else launch<T, 64, false>(tile, grid, s, a);
  }
  return RS_LAUNCH_RESULT();
}

}  // namespace

// fp32 entry for conv_igemm.hip (declared in common.h): every non-stem convolution of rs_conv2d_fwd
int rs_conv_dma_f32(const rs_conv_desc* d, const float* src1, const float* src2, const float* weight, const float* scale,
                    const float* shift, const float* residual, const float* relu_mask, float* out, void* stream) {
  return conv_fwd<float>(d, src1, src2, weight, scale, shift, residual, relu_mask, out, stream);
}

int rs_conv_dma_tile(const rs_conv_desc* d) { return valid(d) ? pick_tile(d) : RS_EINVAL; }

// For the roofline report (kernel names that map 1:1 to the launched symbol): tile index and K-chunk row bytes the
// dispatcher picks for `d` with activations of `es` bytes, in the direct (phase4 = 0) or phase form.
extern "C" int rs_conv2d_config(const rs_conv_desc* d, int es, int phase4, int* tile, int* rowb) {
  if (!valid(d) || (es != 2 && es != 4) || (phase4 && !phase_ok(d))) return RS_EINVAL;
  const int kc128 = 128 / es;
  const bool can128 = d->C1 % kc128 == 0 && d->C2 % kc128 == 0;
  if (tile) *tile = pick_tile(d, phase4 != 0);
  if (rowb) *rowb = (can128 && pick_rowb(d, es, phase4 != 0) == 128) ? 128 : 64;
  return 0;
}

// ---- phase form of the decoder convolutions ---------------------------------------------------------------------------
// fp32 KRSC [Cout][3][3][Cin] -> [4 phases (py, px)][Cout][2][2][Cin] in T: tap (r, s) of phase (py, px) is the sum of the
// original taps that land on the same source pixel: rows {0},{1,2} for py = 0 and {0,1},{2} for py = 1 (same in x).
template <typename T>
__global__ void pack_phase_weight_kernel(const float* __restrict__ w, T* __restrict__ out, int Cout, int Cin, long total) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  const int ci = (int)(i % Cin);
  long t = i / Cin;
  const int s = (int)(t & 1), r = (int)((t >> 1) & 1);
  t >>= 2;
  const int co = (int)(t % Cout);
  const int ph = (int)(t / Cout), py = ph >> 1, px = ph & 1;
  const int ky0 = py == 0 ? (r == 0 ? 0 : 1) : (r == 0 ? 0 : 2), ky1 = py == 0 ? (r == 0 ? 0 : 2) : (r == 0 ? 1 : 2);
  const int kx0 = px == 0 ? (s == 0 ? 0 : 1) : (s == 0 ? 0 : 2), kx1 = px == 0 ? (s == 0 ? 0 : 2) : (s == 0 ? 1 : 2);
  float acc = 0.f;
  for (int ky = ky0; ky <= ky1; ++ky)
    for (int kx = kx0; kx <= kx1; ++kx) acc += w[(((long)co * 3 + ky) * 3 + kx) * Cin + ci];
  out[i] = (T)acc;
}

// fp32 KRSC [Cout][3][3][Cin] -> data-gradient weights of the phase form, [Cin][4][4][Cout] in T: the gradient of
// DecoderBlock wrt its (pre-upsample) input is a 4x4 / stride-2 / pad-1 convolution over dz,
//   d_src[u][v][ci] = sum_{ty,tx,co} dz[2u - 1 + ty][2v - 1 + tx][co] * Wd[ci][ty][tx][co],
// Wd[ty] = sum of the taps ky with ((2u - 1 + ty) + ky - 1) >> 1 == u:  ty 0 -> {2}, 1 -> {1,2}, 2 -> {0,1}, 3 -> {0}.
template <typename T>
__global__ __launch_bounds__(256) void pack_dgrad_phase_weight_kernel(const float* __restrict__ w, T* __restrict__ out, int Cout,
                                                                      int Cin) {
  // block = 32 cins x 32 couts, all taps: reads coalesced along ci (the KRSC inner dimension), LDS transpose, writes
  // coalesced along co (the inner dimension of [Cin][4][4][Cout])
  __shared__ float tile[9][32][33];  // [tap][co][ci]
  const int ci0 = blockIdx.x * 32, co0 = blockIdx.y * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  for (int e = ty; e < 9 * 32; e += 8) {
    const int k = e / 32, co = e - k * 32;
    tile[k][co][tx] = (co0 + co < Cout && ci0 + tx < Cin) ? w[((long)(co0 + co) * 9 + k) * Cin + ci0 + tx] : 0.f;
  }
  __syncthreads();
  for (int e = ty; e < 16 * 32; e += 8) {
    const int t = e / 32, ci = e - t * 32;  // thread writes out[ci0+ci][t][co0+tx]
    const int ty4 = t >> 2, tx4 = t & 3;
    const int ky0 = ty4 == 0 ? 2 : (ty4 == 1 ? 1 : 0), ky1 = ty4 == 0 ? 2 : (ty4 == 1 ? 2 : (ty4 == 2 ? 1 : 0));
    const int kx0 = tx4 == 0 ? 2 : (tx4 == 1 ? 1 : 0), kx1 = tx4 == 0 ? 2 : (tx4 == 1 ? 2 : (tx4 == 2 ? 1 : 0));
    float acc = 0.f;
    for (int ky = ky0; ky <= ky1; ++ky)
      for (int kx = kx0; kx <= kx1; ++kx) acc += tile[ky * 3 + kx][tx][ci];
    if (ci0 + ci < Cin && co0 + tx < Cout) out[((long)(ci0 + ci) * 16 + t) * Cout + co0 + tx] = (T)acc;
  }
}

extern "C" int rs_pack_dgrad_phase_weight_dt(const float* w_krsc, void* out, int dtype, int Cout, int Cin, rs_stream_t stream) {
  if (!w_krsc || !out || Cout <= 0 || Cin <= 0) return RS_EINVAL;
  hipStream_t s = (hipStream_t)stream;
  const dim3 grid(rs_cdiv(Cin, 32), rs_cdiv(Cout, 32));
  if (dtype == RS_F32)
    pack_dgrad_phase_weight_kernel<float><<<grid, 256, 0, s>>>(w_krsc, reinterpret_cast<float*>(out), Cout, Cin);
  else if (dtype == RS_BF16)
    pack_dgrad_phase_weight_kernel<bf16_t><<<grid, 256, 0, s>>>(w_krsc, reinterpret_cast<bf16_t*>(out), Cout, Cin);
  else
    return RS_EINVAL;
  return RS_LAUNCH_RESULT();
}

extern "C" int rs_pack_phase_weight_dt(const float* w_krsc, void* out, int dtype, int Cout, int Cin, rs_stream_t stream) {
  if (!w_krsc || !out || Cout <= 0 || Cin <= 0) return RS_EINVAL;
  const long total = 16L * Cout * Cin;
  hipStream_t s = (hipStream_t)stream;
  if (dtype == RS_F32)
    pack_phase_weight_kernel<float><<<rs_cdiv(total, 256), 256, 0, s>>>(w_krsc, reinterpret_cast<float*>(out), Cout, Cin, total);
  else if (dtype == RS_BF16)
    pack_phase_weight_kernel<bf16_t><<<rs_cdiv(total, 256), 256, 0, s>>>(w_krsc, reinterpret_cast<bf16_t*>(out), Cout, Cin, total);
  else
    return RS_EINVAL;
  return RS_LAUNCH_RESULT();
}

// rs_conv2d_fwd with the torch.cat split of a decoder data gradient fused into the store: output channels [0, csplit) ->
// out1 [N][Ho][Wo][csplit] (zeroed where mask1 <= 0 if given), [csplit, Cout) -> out2 [N][Ho][Wo][Cout-csplit] (mask2).
extern "C" int rs_conv2d_fwd_split_dt(const rs_conv_desc* d, int dtype, const void* src1, const void* weight, void* out1,
                                      const void* mask1, void* out2, const void* mask2, int csplit, rs_stream_t stream) {
  if (!out2 || (d && d->C2 != 0)) return RS_EINVAL;
  if (dtype == RS_F32)
    return conv_fwd<float>(d, src1, nullptr, weight, nullptr, nullptr, nullptr, mask1, out1, stream, nullptr, nullptr, nullptr,
                           nullptr, false, out2, mask2, csplit);
  if (dtype == RS_BF16)
    return conv_fwd<bf16_t>(d, src1, nullptr, weight, nullptr, nullptr, nullptr, mask1, out1, stream, nullptr, nullptr,
                            nullptr, nullptr, false, out2, mask2, csplit);
  return RS_EINVAL;
}

extern "C" int rs_conv2d_fwd_phase_dt(const rs_conv_desc* d, int dtype, const void* src1, const void* src2,
                                      const void* weight_phase, const float* scale, const float* shift,
                                      const void* residual, const void* relu_mask, void* out, rs_stream_t stream) {
  if (dtype == RS_F32)
    return conv_fwd<float>(d, src1, src2, weight_phase, scale, shift, residual, relu_mask, out, stream, nullptr, nullptr,
                           nullptr, nullptr, true);
  if (dtype == RS_BF16)
    return conv_fwd<bf16_t>(d, src1, src2, weight_phase, scale, shift, residual, relu_mask, out, stream, nullptr, nullptr,
                            nullptr, nullptr, true);
  return RS_EINVAL;
}

extern "C" long rs_conv2d_bnstats_rows(const rs_conv_desc* d) {
  if (!valid(d)) return RS_EINVAL;
  return rs_cdiv((long)d->N * d->Ho * d->Wo, kTileBM[pick_tile(d)]);
}

extern "C" int rs_conv2d_dgrad_bnstats_dt(const rs_conv_desc* d, int dtype, const void* dy, const void* weight,
                                          const void* residual, const void* relu_mask, const void* bn_y,
                                          const float* bn_mean, const float* bn_invstd, void* out, float* stats_partial,
                                          rs_stream_t stream) {
  if (!stats_partial || !bn_y || !bn_mean || !bn_invstd || (d && d->C2 != 0)) return RS_EINVAL;
  if (dtype == RS_F32)
    return conv_fwd<float>(d, dy, nullptr, weight, nullptr, nullptr, residual, relu_mask, out, stream, stats_partial, bn_y,
                           bn_mean, bn_invstd);
  if (dtype == RS_BF16)
    return conv_fwd<bf16_t>(d, dy, nullptr, weight, nullptr, nullptr, residual, relu_mask, out, stream, stats_partial, bn_y,
                            bn_mean, bn_invstd);
  return RS_EINVAL;
}

extern "C" int rs_conv2d_fwd_bnstats_dt(const rs_conv_desc* d, int dtype, const void* src1, const void* src2,
                                        const void* weight, void* out, float* stats_partial, rs_stream_t stream) {
  if (!stats_partial) return RS_EINVAL;
  if (dtype == RS_F32) return conv_fwd<float>(d, src1, src2, weight, nullptr, nullptr, nullptr, nullptr, out, stream, stats_partial);
  if (dtype == RS_BF16) return conv_fwd<bf16_t>(d, src1, src2, weight, nullptr, nullptr, nullptr, nullptr, out, stream, stats_partial);
  return RS_EINVAL;
}

extern "C" int rs_conv2d_tile_bf16(const rs_conv_desc* d) { return valid(d) ? pick_tile(d) : RS_EINVAL; }

extern "C" const char* rs_conv2d_tile_name_bf16(int tile) {
  return (tile >= 0 && tile < NTILES) ? kTileNamesBf16[tile] : "";
}

extern "C" int rs_conv2d_fwd_bf16(const rs_conv_desc* d, const rs_bf16* src1, const rs_bf16* src2, const rs_bf16* weight,
                                  const float* scale, const float* shift, const rs_bf16* residual, const rs_bf16* relu_mask,
                                  rs_bf16* out, rs_stream_t stream) {
  return conv_fwd<bf16_t>(d, src1, src2, weight, scale, shift, residual, relu_mask, out, stream);
}

extern "C" int rs_pack_dgrad_weight_bf16(const float* w_krsc, rs_bf16* out, int Cout, int kh, int kw, int Cin,
                                         rs_stream_t stream) {
  if (!w_krsc || !out || Cout <= 0 || kh <= 0 || kw <= 0 || Cin <= 0) return RS_EINVAL;
  dim3 grid(rs_cdiv(Cin, 32), rs_cdiv(Cout, 32), kh * kw);
  pack_dgrad_weight_bf16_kernel<<<grid, 256, 0, (hipStream_t)stream>>>(w_krsc, reinterpret_cast<bf16_t*>(out), Cout, kh * kw,
                                                                       Cin);
  return RS_LAUNCH_RESULT();
}

// Weight-gradient convolution for gfx950 on v_mfma_f32_32x32x16_bf16: bf16 dy and activations, fp32 accumulation and
// fp32 KRSC result (the optimizer's master gradient).  The bf16 training path of BASELINE configs[2].
//
//     dW[co][ky][kx][ci] = sum over output pixels m=(n,oy,ox) of  dy[m][co] * in[n][oy*stride-pad+ky][ox*stride-pad+kx][ci]
//
// i.e. the filter gradients autograd synthesises for every nn.Conv2d of UNet.forward when the reference calls
// loss.backward() (robosat/tools/train.py:186).  GEMM view as conv_wgrad.hip: rows = Cout tile, cols = one filter tap
// x a Cin tile, REDUCTION over pixels; `in` is read through the forward gather (nearest-x2 upsample, 2-source concat).
// (The Cout = 32 3x3 layers take conv_wgrad_thin_bf16.hip instead: all nine taps per block.)
//
//   block  = BMo couts x BNo cins of one tap over a contiguous pixel range (split-P), walked in chunks of 64 pixels
//            (4 MFMA k-steps); two LDS buffers, one barrier per chunk.
//   HBM -> LDS = LDS-DMA (buffer_load_dwordx4 ... lds): both operands are pixel-major in HBM ([pixel][channel]) and are
//            copied AS THEY LIE -- no VGPR round trip, no ds_write, no register transpose.  A wave instruction moves
//            64 x 16 B = 1 KiB: whole [pixel] rows of the tile (256 / 128 / 64 bytes), each lane supplying its own global
//            offset (so the gather -- tap shift, padding, upsample, concat source -- costs one table lookup + one
//            multiply-add per lane, and out-of-image / tail rows are offset -1: the hardware writes zeros,
//            scripts/probes/probe_glds.hip).  The LDS image is lane-linear, so the bank swizzle is applied on the SOURCE
//            side: 64-byte piece P of LDS row R holds channel piece P ^ swz(R).
//   gather = threads 0..63 decode pixel m -> source pixel (mul-hi divisions) for chunk k+2 into a double-buffered
//            64-entry LDS table.
//   MFMA operands = ds_read_b64_tr_b16 (the LDS transposing read, semantics pinned by scripts/probes/probe_tr16.hip):
//            MFMA wants, per lane, 8 consecutive REDUCTION indices (pixels) of one channel, the tile is stored
//            pixel-major; within a 16-lane group 4-lane sets address four pixel rows and every lane receives its
//            channel's column.  Two reads = one 8-pixel fragment.  With the swizzle the four rows of a read sit in four
//            different 64-byte bank quarters: conflict free.
//   split-P: partial tiles -> workspace [split][Cout][K], summed by reduce.hip: deterministic, no atomics.
#include <stdlib.h>

#include <type_traits>

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
  int Ctot, ci_base;  // channels of the whole concat (tap stride inside K) and this launch's first channel in it
  rs_fastdiv div_howo, div_wo;
};

typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ __amdgpu_buffer_rsrc_t wb_make_rsrc(const void* base, long bytes) {
  const unsigned int n = bytes > 0xFFFFFFFEL ? 0xFFFFFFFEu : (unsigned int)(bytes < 0 ? 0 : bytes);
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, (int)n, 0x00020000);
}

// One LDS-DMA wave instruction: lane l's 16 bytes at buffer offset `voff` land at LDS byte `lds_dst` + 16*l (lds_dst
// wave-uniform, in M0).  Inline asm on purpose: through the builtin hipcc cannot tell that the DMA's LDS destination
// (the OTHER pipeline buffer) is disjoint from the operand reads that follow and drains the queue (s_waitcnt vmcnt(0))
// before the first ds_read of every chunk, serialising copy and MFMA.  As asm the copy is invisible to its counters, so
// the kernel waits itself: wb_dma_wait() ahead of the barrier that publishes the buffer.
// m0 is named as a clobber rather than saved/restored around every piece (see the note at rb_dma16s in
// conv_igemm_dma.hip: nothing else in this file makes the compiler use m0).
__device__ __forceinline__ void wb_dma16(__amdgpu_buffer_rsrc_t r, unsigned int lds_dst, int voff) {
  asm volatile(
      "s_mov_b32 m0, %1\n\t"
      "s_nop 0\n\t"
      "buffer_load_dwordx4 %0, %2, 0 offen lds"
      :
      : "v"(voff), "s"(lds_dst), "s"(r)
      : "memory", "m0");
}

__device__ unsigned int wb_zero_line[256];  // 1 KiB of zeros (a __device__ array is zero-initialised): DEAD == 1

__device__ __forceinline__ void wb_dma_wait() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }

__device__ __forceinline__ unsigned int wb_lds_addr(const void* p) {
  return (unsigned int)(unsigned long)(__attribute__((address_space(3))) const void*)p;
}

__device__ __forceinline__ bf16x8 wb_tr_read8(const unsigned char* p0, const unsigned char* p1) {
  typedef __attribute__((address_space(3))) s16x4* lds_ptr;
  const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_ptr)p0);
  const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_ptr)p1);
  s16x8 v;
  v[0] = lo[0];
  v[1] = lo[1];
  v[2] = lo[2];
  v[3] = lo[3];
  v[4] = hi[0];
  v[5] = hi[1];
  v[6] = hi[2];
  v[7] = hi[3];
  return __builtin_bit_cast(bf16x8, v);
}


// PK = pixels per chunk: 64 (4 k-steps per barrier, 64 KB of LDS for the 128x128 tile: 2 blocks per CU) or 32 (2 k-steps,
// half the LDS: 4 blocks per CU -- the short reductions of the small layers)
// PHASE: the layer is DecoderBlock (3x3 / pad 1 behind the nearest-x2 upsample).  Instead of nine taps over the UPSAMPLED
// pixels the block reduces one of 16 (parity, source offset) combinations over the SOURCE pixels:
//   G[(py,px),(r,s)][co][ci] = sum_{n,a,b} dz[n][2a+py][2b+px][co] * src[n][a-(1-py)+r][b-(1-px)+s][ci]
// (4/9 of the multiply-adds; dz rows are then gathered too: a second table), and combine_phase_wgrad_kernel adds the four
// G's that make up each filter tap: dW[ky][kx] = sum_{(py,r) : ky in S(py,r)} sum_{(px,s) : kx in S(px,s)} G, with
// S(0,0) = {0}, S(0,1) = {1,2}, S(1,0) = {0,1}, S(1,1) = {2} (the taps that hit the same source pixel).
// RING = chunk buffers.  SHIPPED: 3 for the tap-per-block launches (knob wgrad_ring), 2 for the phase form.
// 2: one chunk streams in while one is multiplied, the wave drains its DMA queue before every barrier (two or more blocks per CU cover
// each other's round trips).  3 (round 5): the tap-per-block launches run ONE block per CU (knob wgrad_blocks = 96) and a chunk's MFMAs
// are a fraction of a DMA round trip, so with two buffers every chunk paid one; with two chunks in flight behind the one being multiplied
// and counted waits (s_waitcnt vmcnt(NI): "all but the newest chunk's pieces have landed") the block keeps its CU's LDS-DMA path busy
// by itself.  The tail of a split drains (the last iterations wait with vmcnt(0) and issue nothing).
// History (profiles/r05/wgrad_ring.txt, profiles/r06/dma_order.txt): round 5 shipped the ring and conv_wgrad_phase4_bf16 as defaults, found
// 42-96 % of their launches not bit-reproducible beside an LDS-using neighbour on the CU, blamed the out-of-range pieces that padded the tail
// of a split ("they retire ahead of older loads, so the counted wait lies") and withdrew both.  Round 6 asked the hardware
// (scripts/probes/probe_dma_order.hip: LDS-DMA pieces retire IN ORDER whatever their lanes address -- 0 stale of 2.9 M pieces) and the
// kernel (scripts/dma_order_forensics.py: a wrong launch multiplies dy with the x rows of the chunk RING iterations EARLIER): the defect was
// the gather table.  Wave 0 writes chunk c + RING's table at the end of iteration c, everybody reads it behind the next barrier -- and hipcc
// emits that __syncthreads() as a BARE s_barrier in these loops (its `s_waitcnt lgkmcnt(0)` is dropped; see rs_lds_writes_done in common.h),
// so with LDS traffic from a neighbour the other waves' ds_read_b32 overtook the ds_write_b32 and fetched the old chunk's rows again.
// fill_table now waits for its own write.  With that one line every variant of the tail -- padded out of range, padded with real loads,
// drained -- is bit-identical to two buffers in 100 of 100 rounds; without it (RING = 4 / DEAD = 0, kept as the race screen's positive
// control and flagged by scripts/isa_audit.py) 43-85 of 100 are not.
// DEAD (RING = 4 only; round 6, the bisect of what exactly goes wrong with the padded tail -- profiles/r06/dma_order.txt): what a chunk past
// the end of the split issues.  0: out-of-range pieces into the dead ring slot (the round-5 control);  1: in-range loads of a 1 KiB zero
// line into the dead slot;  2: out-of-range pieces into a scratch KiB nobody ever reads;  3: whatever follows the split in memory (the next
// split's chunk: ordinary loads, out of range only behind the last pixel).
template <int BMo, int BNo, int WGM, int WGN, int PK, bool PHASE, int RING = 2, int DEAD = 0>
__global__ __launch_bounds__(64 * WGM * WGN, (WGM * WGN > 4 || RING > 2) ? 1 : 2) void conv_wgrad_bf16(const WgradArgsB p) {
  constexpr int NW = WGM * WGN;          // waves
  constexpr bool FIXLGKM = !(RING == 4 && DEAD == 0);  // (the RING = 4 / DEAD = 0 instantiation stays as it was: the race screen's positive control)
  constexpr int NS = PK / 16;            // MFMA k-steps per chunk
  constexpr int WM = BMo / WGM, WN = BNo / WGN;
  constexpr int TM = WM / 32, TN = WN / 32;
  constexpr int ROWA = BMo * 2, ROWB_ = BNo * 2;        // bytes per LDS row (one pixel)
  constexpr int WA = BMo / 32, WB = BNo / 32;           // 64-byte pieces per row
  constexpr int RPBA = WA >= 4 ? 1 : 4 / WA, RPBB = WB >= 4 ? 1 : 4 / WB;  // rows per 256-byte bank row
  constexpr int RIA = 1024 / ROWA, RIB = 1024 / ROWB_;  // rows per DMA instruction
  constexpr int IA = PK / RIA, IB = PK / RIB;           // DMA instructions per chunk
  constexpr int NI = (IA + IB) / NW;                    // per wave
  constexpr int ABYTES = PK * ROWA, BBYTES = PK * ROWB_;
  constexpr int BUF = ABYTES + BBYTES;
  static_assert(TM >= 1 && TN >= 1 && (IA % NW) == 0 && (IB % NW) == 0, "bad tile");

  static_assert(RING >= 2 && (RING - 2) * NI <= 63, "vmcnt is a 6-bit count");
  __shared__ __attribute__((aligned(16))) unsigned char smem[RING * BUF + (PHASE ? 2 : 1) * RING * PK * 4 + (DEAD == 2 ? 1024 : 0)];
  constexpr int SCRATCH = RING * BUF + (PHASE ? 2 : 1) * RING * PK * 4;
  int* tabs = reinterpret_cast<int*>(smem + RING * BUF);  // [RING][PK] input-row gather
  int* taba = tabs + RING * PK;                            // [RING][PK] dz-row gather (PHASE only)

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WGN, wn = wave % WGN;

  int bid = rs_xcd_remap(blockIdx.x, gridDim.x);
  const int tco = bid % p.tiles_co;
  bid /= p.tiles_co;
  const int tk = bid % p.tiles_k;
  const int split = bid / p.tiles_k;
  const int tap = tk / p.tiles_ci, tci = tk - tap * p.tiles_ci;
  const int ky = PHASE ? ((tap >> 1) & 1) : tap / p.kw;  // PHASE: tap = 4*(2*py+px) + 2*r + s
  const int kx = PHASE ? (tap & 1) : tap - ky * p.kw;
  const int py = (tap >> 3) & 1, px = (tap >> 2) & 1;
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
  const int Wd = PHASE ? p.Ws : p.Wo;                   // the pixel grid the reduction index m enumerates
  const int HoWo = PHASE ? p.Hs * p.Ws : p.Ho * p.Wo;   // (div_howo / div_wo are prepared for that grid)

  const int m_first = chunk0 * PK;
  const int n_first = (int)rs_div((unsigned)m_first, p.div_howo);
  const long img = (long)p.Hs * p.Ws * Cs;
  const long dimg = (long)p.Ho * p.Wo * p.Cout;
  const __amdgpu_buffer_rsrc_t rsrc_dy =
      PHASE ? wb_make_rsrc(p.dy + n_first * dimg, (long)(p.N - n_first) * dimg * 2)
            : wb_make_rsrc(p.dy + (long)m_first * p.Cout, ((long)p.M - m_first) * p.Cout * 2);
  const __amdgpu_buffer_rsrc_t rsrc_x = wb_make_rsrc(src + n_first * img, (long)(p.N - n_first) * img * 2);
  const int ush = p.ups ? 1 : 0;
  const int upar = p.ups == 2 ? 1 : 0;

  // pixel m -> source pixel of this block's tap (relative to image n_first), -1 = contributes zeros
  auto fill_table = [&](int chunk, int which) __attribute__((always_inline)) {
    if (tid < PK) {
      const int m = chunk * PK + tid;
      int pix = -1, pixa = -1;
      if (m < p.M) {
        const int n = (int)rs_div((unsigned)m, p.div_howo);
        const int rem = m - n * HoWo;
        const int oy = (int)rs_div((unsigned)rem, p.div_wo);
        const int ox = rem - oy * Wd;
        if (PHASE) {
          const int iy = oy - (1 - py) + ky, ix = ox - (1 - px) + kx;
          if (((unsigned)iy < (unsigned)p.Hs) && ((unsigned)ix < (unsigned)p.Ws)) pix = ((n - n_first) * p.Hs + iy) * p.Ws + ix;
          pixa = ((n - n_first) * p.Ho + 2 * oy + py) * p.Wo + 2 * ox + px;
        } else {
          const int iy = oy * p.stride - p.pad + ky;
          const int ix = ox * p.stride - p.pad + kx;
          const bool ok = ((unsigned)iy < (unsigned)p.Hv) && ((unsigned)ix < (unsigned)p.Wv) && (((iy | ix) & upar) == 0);
          if (ok) pix = ((n - n_first) * p.Hs + (iy >> ush)) * p.Ws + (ix >> ush);
        }
      }
      tabs[which * PK + tid] = pix;
      if (PHASE) taba[which * PK + tid] = pixa;
      if (FIXLGKM) rs_lds_writes_done();
    }
  };

  // ---- DMA roles: instruction ii = wave + NW*j (j < NI); ii < IA copies dy rows, else input rows.  Lane constants:
  //      row within the instruction, 16-byte position, and the (swizzled) channel piece it fetches.
  const int ra_a = lane / (ROWA / 16), pp_a = lane % (ROWA / 16);
  const int ra_b = lane / (ROWB_ / 16), pp_b = lane % (ROWB_ / 16);
  // swz(R) of the LDS row R = RIA * (wave + NW*j) + ra_a this lane fills: RIA*NW is a multiple of the swizzle period
  // (WA*RPBA rows), so the row's swizzle does not depend on j
  static_assert((RIA * NW) % (WA * RPBA) == 0 && (RIB * NW) % (WB * RPBB) == 0, "swizzle period");
  const int swa = WA > 1 ? (((RIA * wave + ra_a) / RPBA) & (WA - 1)) : 0;
  const int swb = WB > 1 ? (((RIB * wave + ra_b) / RPBB) & (WB - 1)) : 0;
  const int gpa = (((pp_a >> 2) ^ swa) << 2) | (pp_a & 3);  // global 16-byte piece (8 channels) of the row
  const int gpb = (((pp_b >> 2) ^ swb) << 2) | (pp_b & 3);
  const int cola = (co0 + gpa * 8) * 2;                     // byte offset of the piece inside a dy row
  const int colb = (cs + gpb * 8) * 2;
  const int cout2 = p.Cout * 2, cs2 = Cs * 2;

  const unsigned int lds0 = __builtin_amdgcn_readfirstlane(wb_lds_addr(smem));
  // Per chunk: first the byte offsets of all of this wave's pieces (one LDS round trip for the gather tables), then the
  // pieces themselves -- in the steady state BETWEEN the MFMAs of the previous chunk (an LDS-DMA instruction costs its
  // wave 60-180 cycles of issue; a burst at the top of the chunk puts that on every wave's critical path at once).
  int voff[NI];
  unsigned int fL = lds0;
  bool fdead = false;  // (wave-uniform) the chunk being fetched lies past the split
  const __amdgpu_buffer_rsrc_t rsrc_zero = wb_make_rsrc(wb_zero_line, 1024);
  // (`live` = false, RING > 2 only: a chunk past the block's range -- its pieces are still issued, out of range (zeros into a
  // buffer nobody reads), so that every wave issues the same number of DMA instructions per step: what the counted waits rely on)
  auto prepare_dma = [&](int chunk, int buf, int which, bool live = true) __attribute__((always_inline)) {
    fL = lds0 + buf * BUF;
    int pix[NI];
#pragma unroll
    for (int j = 0; j < NI; ++j)
      pix[j] = (NW * j >= IA) ? tabs[which * PK + RIB * (wave + NW * j - IA) + ra_b]
                              : (PHASE ? taba[which * PK + RIA * (wave + NW * j) + ra_a] : 0);
#pragma unroll
    for (int j = 0; j < NI; ++j) {
      const int ii = wave + NW * j;  // wave-uniform; IA is a multiple of NW, so the role depends on j alone
      if (NW * j < IA) {
        if (PHASE) {
          voff[j] = pix[j] >= 0 ? pix[j] * cout2 + cola : -1;
        } else {
          const int m = chunk * PK + RIA * ii + ra_a;
          voff[j] = (m - m_first) * cout2 + cola;  // rows >= M lie past the descriptor: zeros
        }
      } else {
        voff[j] = pix[j] >= 0 ? pix[j] * cs2 + colb : -1;
      }
      if (RING > 2 && !live && DEAD != 3) voff[j] = DEAD == 1 ? lane * 16 : -1;
    }
    fdead = RING > 2 && !live;
  };
  auto issue_piece = [&](int j) __attribute__((always_inline)) {
    const int ii = wave + NW * j;
    if (DEAD == 1 && fdead) wb_dma16(rsrc_zero, NW * j < IA ? fL + ii * 1024 : fL + ABYTES + (ii - IA) * 1024, voff[j]);
    else if (DEAD == 2 && fdead) wb_dma16(rsrc_dy, lds0 + SCRATCH, voff[j]);
    else if (NW * j < IA) wb_dma16(rsrc_dy, fL + ii * 1024, voff[j]);
    else wb_dma16(rsrc_x, fL + ABYTES + (ii - IA) * 1024, voff[j]);
  };

  f32x16 acc[TM][TN];
#pragma unroll
  for (int a = 0; a < TM; ++a)
#pragma unroll
    for (int b = 0; b < TN; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

  // ---- operand addressing for ds_read_b64_tr_b16: lane = 16*g + q; group g covers channels 16*(g&1)..+15 of a 32-wide
  //      piece and LDS rows R = 16s + 8*(g>>1) + 4t + (q>>2); lane q addresses row (q>>2), channels 4*(q&3)..+3.
  const int g = lane >> 4, q = lane & 15, jr = q >> 2;
  const int chb = (16 * (g & 1) + 4 * (q & 3)) * 2;
  // the two halves of a fragment read rows 16s + 8*(g>>1) + 4t + jr (t = 0, 1): swz(R) per half (for tiles up to 128 wide
  // the period divides 4 and both halves share it; the 256-wide tile's period is 8 rows)
  int aoff[TM][2], boff[TN][2];
#pragma unroll
  for (int t = 0; t < 2; ++t) {
    const int rr16 = 8 * (g >> 1) + 4 * t + jr;
    const int rsw_a = WA > 1 ? ((rr16 / RPBA) & (WA - 1)) : 0;
    const int rsw_b = WB > 1 ? ((rr16 / RPBB) & (WB - 1)) : 0;
#pragma unroll
    for (int tm = 0; tm < TM; ++tm) aoff[tm][t] = rr16 * ROWA + (((wm * TM + tm) ^ rsw_a) * 64) + chb;
#pragma unroll
    for (int tn = 0; tn < TN; ++tn) boff[tn][t] = ABYTES + rr16 * ROWB_ + (((wn * TN + tn) ^ rsw_b) * 64) + chb;
  }

  constexpr int NMMA = NS * TM * TN;
  constexpr int PSTEP = NMMA / (2 * NI) >= 1 ? NMMA / (2 * NI) : 1;  // front-loaded: the chunk's tail covers the latency
  constexpr int PIN = (NMMA + PSTEP - 1) / PSTEP < NI ? (NMMA + PSTEP - 1) / PSTEP : NI;
  auto chunk_mma = [&](const unsigned char* L, auto fetch_tag) __attribute__((always_inline)) {
    constexpr bool FETCH = decltype(fetch_tag)::value;
#pragma unroll
    for (int s = 0; s < NS; ++s) {
      bf16x8 fa[TM], fb[TN];
#pragma unroll
      for (int tm = 0; tm < TM; ++tm)
        fa[tm] = wb_tr_read8(L + aoff[tm][0] + (16 * s) * ROWA, L + aoff[tm][1] + (16 * s) * ROWA);
#pragma unroll
      for (int tn = 0; tn < TN; ++tn)
        fb[tn] = wb_tr_read8(L + boff[tn][0] + (16 * s) * ROWB_, L + boff[tn][1] + (16 * s) * ROWB_);
#pragma unroll
      for (int tm = 0; tm < TM; ++tm)
#pragma unroll
        for (int tn = 0; tn < TN; ++tn) {
          const int q = (s * TM + tm) * TN + tn;  // MFMA index within the chunk (compile-time after unrolling)
          if (FETCH && q % PSTEP == 0 && q / PSTEP < PIN) issue_piece(q / PSTEP);
          acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[tm], fb[tn], acc[tm][tn], 0, 0, 0);
        }
    }
    if (FETCH) {
#pragma unroll
      for (int q = PIN; q < NI; ++q) issue_piece(q);
    }
  };
  if (RING == 3) {
    // the tail DRAINS: the wait at the top of iteration c allows the pieces of chunk c + 1 only if that chunk exists (padding the tail with
    // out-of-range pieces instead, as RING = 4 below does, is equally correct -- LDS-DMA pieces retire in order -- and costs a few dead pieces)
    if (chunk0 < chunk1) {
#pragma unroll
      for (int r = 0; r < RING; ++r) fill_table(chunk0 + r, r);
      __syncthreads();
      prepare_dma(chunk0, 0, 0);
#pragma unroll
      for (int q = 0; q < NI; ++q) issue_piece(q);
      if (chunk0 + 1 < chunk1) {
        prepare_dma(chunk0 + 1, 1, 1);
#pragma unroll
        for (int q = 0; q < NI; ++q) issue_piece(q);
      }
      int slot = 0;  // ring slot of the chunk being multiplied
      for (int c = chunk0; c < chunk1; ++c) {
        if (c + 1 < chunk1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NI) : "memory");  // chunk c landed, chunk c + 1 may still fly
        else wb_dma_wait();
        __syncthreads();
        const int s2 = slot == 0 ? RING - 1 : slot - 1;
        if (c + 2 < chunk1) {
          prepare_dma(c + 2, s2, s2);
          chunk_mma(smem + slot * BUF, std::true_type());
        } else {
          chunk_mma(smem + slot * BUF, std::false_type());
        }
        fill_table(c + RING, slot);  // (this slot's table was read by prepare_dma RING - 1 iterations ago)
        slot = slot == RING - 1 ? 0 : slot + 1;
      }
    }
  } else if (RING > 3) {
    // RING = 4 (LATE): round 5's first version (padded tail, reads one iteration behind the wait).  DEAD = 0 is the race screen's POSITIVE
    // CONTROL: it keeps the defect (FIXLGKM = false: the gather table is published through a bare s_barrier); DEAD = 1-3 (`make EXP=1`) are
    // the bisect's variants of the padded tail, all with the fix, all clean.
    constexpr bool LATE = RING >= 4;
    constexpr int LOOPWAIT = (LATE ? RING - 3 : RING - 2) * NI;
    if (chunk0 < chunk1) {
#pragma unroll
      for (int r = 0; r < RING; ++r) fill_table(chunk0 + r, r);
      __syncthreads();
#pragma unroll
      for (int r = 0; r < RING - 1; ++r) {
        prepare_dma(chunk0 + r, r, r, chunk0 + r < chunk1);
#pragma unroll
        for (int q = 0; q < NI; ++q) issue_piece(q);
      }
      if (LATE) {  // chunk0 retired, and a barrier behind that wait, before the loop's first read
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"((RING - 2) * NI) : "memory");
        __syncthreads();
      }
      int slot = 0;  // ring slot of the chunk being multiplied
      for (int c = chunk0; c < chunk1; ++c) {
        // RING = 3: chunk c has landed when at most the pieces of the chunk behind it are still in flight; the barrier publishes
        // everybody's share and frees the slot chunk c - 1 was read from: chunk c + RING - 1 streams into it between the MFMAs
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(LOOPWAIT) : "memory");
        __syncthreads();
        const int s2 = slot == 0 ? RING - 1 : slot - 1;
        prepare_dma(c + RING - 1, s2, s2, c + RING - 1 < chunk1);
        chunk_mma(smem + slot * BUF, std::true_type());
        fill_table(c + RING, slot);  // (this slot's table was read by prepare_dma RING - 1 iterations ago)
        slot = slot == RING - 1 ? 0 : slot + 1;
      }
      wb_dma_wait();  // (the out-of-range pieces of the last steps)
    }
  } else if (chunk0 < chunk1) {
    fill_table(chunk0, 0);
    fill_table(chunk0 + 1, 1);
    __syncthreads();
    prepare_dma(chunk0, 0, 0);
#pragma unroll
    for (int q = 0; q < NI; ++q) issue_piece(q);
    wb_dma_wait();
    __syncthreads();
    int c = chunk0;
    for (; c + 1 < chunk1; ++c) {  // steady state: chunk c+1 streams into the other buffer between chunk c's MFMAs
      const int it = c - chunk0;
      // (the other buffer's last readers passed the barrier that ended iteration it-1)
      prepare_dma(c + 1, (it + 1) & 1, (it + 1) & 1);
      chunk_mma(smem + (it & 1) * BUF, std::true_type());
      fill_table(c + 2, it & 1);
      wb_dma_wait();  // this wave's share of chunk c+1 has landed; the barrier publishes everybody's
      __syncthreads();
    }
    chunk_mma(smem + ((c - chunk0) & 1) * BUF, std::false_type());
  }

  // D[i][j]: i = cout (tile-local) = (r&3) + 8*(r>>2) + 4*(lane>>5), j = cin (tile-local) = lane&31
  float* out = p.out + (long)split * p.Cout * p.K;
  const int kbase = tap * p.Ctot + p.ci_base + ci0;
#pragma unroll
  for (int tm = 0; tm < TM; ++tm)
#pragma unroll
    for (int tn = 0; tn < TN; ++tn) {
      const int kk = kbase + wn * WN + tn * 32 + (lane & 31);
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int co = co0 + wm * WM + tm * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        out[(long)co * p.K + kk] = acc[tm][tn][r];
      }
    }
}

// ---- phase form, FOUR source offsets per block (round 5) ---------------------------------------------------------------------
// The 16 (output parity, source offset) reductions of DecoderBlock's weight gradient pair each of the four dz parity planes with
// four shifted views of the source.  conv_wgrad_bf16<.., PHASE> gives every pair its own block: a 128 x 128 tile fetches
// 256 rows per 64-pixel chunk for ONE product, and the launches are bound by the CUs' LDS-DMA paths (MFMA busy 41 %,
// profiles/r05/pmc_mfma_per_kernel.txt).  Here a block owns one PLANE and all four of its offsets: the dz tile is fetched once and
// multiplied with four source tiles -- 640 rows per chunk for FOUR products (0.625 x the bytes per multiply-add).  8 waves as
// 2 (couts) x 4 (cins): wave tile 64 x 32 for each of the four offsets = 128 accumulator registers; 32-pixel chunks (two MFMA
// k-steps per barrier), two buffers of 40 KB, one block per CU.  Same gather tables, swizzle, transposing reads, partial-tile
// format ([split][Cout][16 x Ctot], tap = 4 * plane + offset) and combine as the kernel above.  Cout % 128 == 0, Cin tile 128.
__global__ __launch_bounds__(512, 2) void conv_wgrad_phase4_bf16(const WgradArgsB p) {
  constexpr int BMo = 128, BNo = 128, WGM = 2, WGN = 4, PK = 32, NOFF = 4;
  constexpr int NW = WGM * WGN;
  constexpr int NS = PK / 16;
  constexpr int WM = BMo / WGM, TM = WM / 32;               // 64, 2 (couts per wave)
  constexpr int ROW = 256;                                  // bytes per LDS row: 128 channels of one pixel (both operands)
  constexpr int RI = 1024 / ROW;                            // 4 rows per DMA instruction
  constexpr int IA = PK / RI, IB = PK / RI;                 // 8 instructions for the dz tile, 8 per source tile
  static_assert(IA == NW && IB == NW, "one dz instruction + one per offset for every wave");
  constexpr int NI = 1 + NOFF;
  constexpr int TBYTES = PK * ROW;                          // 8 KB per tile
  constexpr int BUF = (1 + NOFF) * TBYTES;                  // dz tile, then the four source tiles

  constexpr int RING = 3;  // chunk buffers: two chunks in flight behind the one being multiplied (one block per CU: nobody else
                           // covers a DMA round trip, and a chunk's 16 MFMAs per wave are a fraction of one)
  __shared__ __attribute__((aligned(16))) unsigned char smem[RING * BUF + RING * (1 + NOFF) * PK * 4];
  int* tabs = reinterpret_cast<int*>(smem + RING * BUF);  // [RING][NOFF][PK] source-row gather per offset
  int* taba = tabs + RING * NOFF * PK;                     // [RING][PK] dz-row gather

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WGN, wn = wave % WGN;

  int bid = rs_xcd_remap(blockIdx.x, gridDim.x);
  const int tco = bid % p.tiles_co;
  bid /= p.tiles_co;
  const int tk = bid % p.tiles_k;  // tiles_k = 4 planes x tiles_ci
  const int split = bid / p.tiles_k;
  const int plane = tk / p.tiles_ci, tci = tk - plane * p.tiles_ci;
  const int py = plane >> 1, px = plane & 1;
  const int co0 = tco * BMo;
  const int ci0 = tci * BNo;

  const bf16_t* src = p.src1;
  int Cs = p.C1, cs = ci0;
  if (ci0 >= p.C1) {
    src = p.src2;
    Cs = p.C2;
    cs = ci0 - p.C1;
  }

  // (the plan counts 64-pixel chunks)
  const int chunk0 = split * p.chunks_per_split * (64 / PK);
  const int total_chunks = (p.M + PK - 1) / PK;
  int chunk1 = chunk0 + p.chunks_per_split * (64 / PK);
  if (chunk1 > total_chunks) chunk1 = total_chunks;
  const int HsWs = p.Hs * p.Ws;

  const int m_first = chunk0 * PK;
  const int n_first = (int)rs_div((unsigned)m_first, p.div_howo);
  const long img = (long)p.Hs * p.Ws * Cs;
  const long dimg = (long)p.Ho * p.Wo * p.Cout;
  const __amdgpu_buffer_rsrc_t rsrc_dy = wb_make_rsrc(p.dy + n_first * dimg, (long)(p.N - n_first) * dimg * 2);
  const __amdgpu_buffer_rsrc_t rsrc_x = wb_make_rsrc(src + n_first * img, (long)(p.N - n_first) * img * 2);

  // pixel m = source pixel (n, a, b): dz row (2a + py, 2b + px); source row of offset (r, s): (a - (1 - py) + r, b - (1 - px) + s)
  auto fill_table = [&](int chunk, int which) __attribute__((always_inline)) {
    if (tid < (1 + NOFF) * PK) {
      const int o = tid / PK, i = tid - o * PK;  // o < NOFF: source offset o = 2 r + s;  o == NOFF: the dz rows
      const int m = chunk * PK + i;
      int pix = -1;
      if (m < p.M) {
        const int n = (int)rs_div((unsigned)m, p.div_howo);
        const int rem = m - n * HsWs;
        const int a = (int)rs_div((unsigned)rem, p.div_wo);
        const int b = rem - a * p.Ws;
        if (o == NOFF) {
          pix = ((n - n_first) * p.Ho + 2 * a + py) * p.Wo + 2 * b + px;
        } else {
          const int iy = a - (1 - py) + (o >> 1), ix = b - (1 - px) + (o & 1);
          if (((unsigned)iy < (unsigned)p.Hs) && ((unsigned)ix < (unsigned)p.Ws)) pix = ((n - n_first) * p.Hs + iy) * p.Ws + ix;
        }
      }
      if (o == NOFF) taba[which * PK + i] = pix;
      else tabs[(which * NOFF + o) * PK + i] = pix;
      rs_lds_writes_done();
    }
  };

  // DMA roles: piece 0 = rows RI*wave.. of the dz tile, piece 1 + o = the same rows of source tile o.  Lane: row ra of the
  // instruction, 16-byte position pp, fetching channel piece gp (64-byte pieces XOR-swizzled by the row: swz(R) = R & 3)
  const int ra = lane >> 4, pp = lane & 15;
  const int gp = (((pp >> 2) ^ ra) << 2) | (pp & 3);
  const int cola = (co0 + gp * 8) * 2;
  const int colb = (cs + gp * 8) * 2;
  const int cout2 = p.Cout * 2, cs2 = Cs * 2;
  const unsigned int lds0 = __builtin_amdgcn_readfirstlane(wb_lds_addr(smem));
  int voff[NI];
  unsigned int fL = lds0;
  // (`live` = 0: a chunk past the block's range -- its five pieces are still issued, out of range (zeros into a buffer nobody
  // reads), so that every wave issues the same number of DMA instructions per step: what the counted waits rely on)
  auto prepare_dma = [&](int buf, int which, int live) __attribute__((always_inline)) {
    fL = lds0 + buf * BUF;
    const int row = RI * wave + ra;
    const int pa = live ? taba[which * PK + row] : -1;
    voff[0] = pa >= 0 ? pa * cout2 + cola : -1;
#pragma unroll
    for (int o = 0; o < NOFF; ++o) {
      const int px_ = live ? tabs[(which * NOFF + o) * PK + row] : -1;
      voff[1 + o] = px_ >= 0 ? px_ * cs2 + colb : -1;
    }
  };
  auto issue_piece = [&](int j) __attribute__((always_inline)) {  // j: compile-time after unrolling
    if (j == 0) wb_dma16(rsrc_dy, fL + wave * 1024, voff[0]);
    else wb_dma16(rsrc_x, fL + j * TBYTES + wave * 1024, voff[j]);
  };

  f32x16 acc[NOFF][TM];
#pragma unroll
  for (int o = 0; o < NOFF; ++o)
#pragma unroll
    for (int a = 0; a < TM; ++a)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[o][a][r] = 0.f;

  // operand addressing for ds_read_b64_tr_b16 (as in conv_wgrad_bf16): lane = 16 g + q; group g covers channels 16 (g & 1) .. + 15
  // of a 32-wide piece and rows 8 (g >> 1) + 4 t + (q >> 2) of a 16-row k-step; lane q addresses row q >> 2, channels 4 (q & 3) ..
  const int g = lane >> 4, q = lane & 15, jr = q >> 2;
  const int chb = (16 * (g & 1) + 4 * (q & 3)) * 2;
  int aoff[TM][2], boff[2];
#pragma unroll
  for (int t = 0; t < 2; ++t) {
    const int rr16 = 8 * (g >> 1) + 4 * t + jr;
    const int rsw = rr16 & 3;
#pragma unroll
    for (int tm = 0; tm < TM; ++tm) aoff[tm][t] = rr16 * ROW + (((wm * TM + tm) ^ rsw) * 64) + chb;
    boff[t] = TBYTES + rr16 * ROW + ((wn ^ rsw) * 64) + chb;
  }

  constexpr int NMMA = NS * NOFF * TM;  // 16 MFMAs per chunk per wave
  auto chunk_mma = [&](const unsigned char* L, auto fetch_tag) __attribute__((always_inline)) {
    constexpr bool FETCH = decltype(fetch_tag)::value;
#pragma unroll
    for (int s = 0; s < NS; ++s) {
      bf16x8 fa[TM], fb[NOFF];
#pragma unroll
      for (int tm = 0; tm < TM; ++tm) fa[tm] = wb_tr_read8(L + aoff[tm][0] + (16 * s) * ROW, L + aoff[tm][1] + (16 * s) * ROW);
#pragma unroll
      for (int o = 0; o < NOFF; ++o)
        fb[o] = wb_tr_read8(L + boff[0] + o * TBYTES + (16 * s) * ROW, L + boff[1] + o * TBYTES + (16 * s) * ROW);
#pragma unroll
      for (int o = 0; o < NOFF; ++o)
#pragma unroll
        for (int tm = 0; tm < TM; ++tm) {
          const int i = (s * NOFF + o) * TM + tm;  // MFMA index within the chunk (compile-time after unrolling)
          if (FETCH && i % 2 == 0 && i / 2 < NI) issue_piece(i / 2);  // the five pieces of the next chunk under the first ten MFMAs
          acc[o][tm] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[tm], fb[o], acc[o][tm], 0, 0, 0);
        }
    }
    static_assert(NMMA / 2 >= NI, "every piece finds an MFMA to hide behind");
  };
  if (chunk0 < chunk1) {
    fill_table(chunk0, 0);
    fill_table(chunk0 + 1, 1);
    fill_table(chunk0 + 2, 2);
    __syncthreads();
    prepare_dma(0, 0, 1);
#pragma unroll
    for (int j = 0; j < NI; ++j) issue_piece(j);
    prepare_dma(1, 1, chunk0 + 1 < chunk1);
#pragma unroll
    for (int j = 0; j < NI; ++j) issue_piece(j);
    int slot = 0;  // ring slot of the chunk being multiplied
    for (int c = chunk0; c < chunk1; ++c) {
      // chunk c has landed when at most the NI pieces of chunk c + 1 are still in flight; the barrier publishes everybody's
      // share and frees the slot chunk c - 1 was read from: chunk c + 2 streams into it between this chunk's MFMAs
      asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NI) : "memory");
      __syncthreads();
      const int s2 = slot == 0 ? 2 : slot - 1;  // (slot + 2) % 3
      prepare_dma(s2, s2, c + 2 < chunk1);
      chunk_mma(smem + slot * BUF, std::true_type());
      fill_table(c + 3, slot);  // (this slot's table was read by prepare_dma two iterations ago)
      slot = slot == 2 ? 0 : slot + 1;
    }
    wb_dma_wait();  // (the out-of-range pieces of the last two steps)
  }

  // D[i][j]: i = cout (tile-local) = (r&3) + 8*(r>>2) + 4*(lane>>5), j = cin (tile-local) = lane&31; offset o -> tap 4 * plane + o
  float* out = p.out + (long)split * p.Cout * p.K;
#pragma unroll
  for (int o = 0; o < NOFF; ++o) {
    const int kk = (4 * plane + o) * p.Ctot + p.ci_base + ci0 + wn * 32 + (lane & 31);
#pragma unroll
    for (int tm = 0; tm < TM; ++tm)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int co = co0 + wm * WM + tm * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        out[(long)co * p.K + kk] = acc[o][tm][r];
      }
  }
}

struct Plan {
  int bmo, bno, variant, tiles_co, tiles_ci, taps, tiles_k, splits, chunks_per_split, pk, phase;
  int phase4;  // the 128 x 128 launch of a phase-form layer runs conv_wgrad_phase4_bf16 (one plane x four offsets per block)
  // two-source layers whose sources allow different Cin tile widths (dec3: 256 + 64 channels) run one launch per source,
  // each with its own widest tile (128-wide tiles feed twice the MFMAs per LDS-DMA byte of 64-wide ones): bno2 != 0
  int bno2, variant2, tiles_ci2;
  long K;
};

// G [Cout][16][Cin] (phase form, tap index 4*(2*py+px) + 2*r + s) -> dW [Cout][3][3][Cin]
__global__ void combine_phase_wgrad_kernel(const float* __restrict__ g, float* __restrict__ dw, int Cin, long total) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  const int ci = (int)(i % Cin);
  long t = i / Cin;
  const int kx = (int)(t % 3), ky = (int)((t / 3) % 3);
  const long co = t / 9;
  // (py, r) pairs whose tap set contains ky: ky 0 -> (0,0),(1,0); 1 -> (0,1),(1,0); 2 -> (0,1),(1,1)
  const int ya[2][2] = {{0, ky == 0 ? 0 : 1}, {1, ky == 2 ? 1 : 0}};
  const int xa[2][2] = {{0, kx == 0 ? 0 : 1}, {1, kx == 2 ? 1 : 0}};
  float acc = 0.f;
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b) {
      const int tap = 4 * (2 * ya[a][0] + xa[b][0]) + 2 * ya[a][1] + xa[b][1];
      acc += g[(co * 16 + tap) * Cin + ci];
    }
  dw[i] = acc;
}

enum { V128x128 = 0, V128x64, V64x128, V64x64, V32x128, V32x32, V256x128 };

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

bool phase_ok(const rs_conv_desc* d) {
  return d->ups == 1 && d->kh == 3 && d->kw == 3 && d->stride == 1 && d->pad == 1 && d->Ho == 2 * d->Hs && d->Wo == 2 * d->Ws;
}

Plan plan(const rs_conv_desc* d) {
  Plan pl;
  pl.phase = phase_ok(d) ? 1 : 0;
  const long M = pl.phase ? (long)d->N * d->Hs * d->Ws : (long)d->N * d->Ho * d->Wo;
  pl.bmo = largest_tile(d->Cout);
  pl.bno = largest_tile(d->C1);
  pl.bno2 = 0;
  if (d->C2 > 0) {
    const int b2 = largest_tile(d->C2);
    if (b2 != pl.bno && b2 >= 64 && pl.bno >= 64 && pl.bmo >= 64) pl.bno2 = b2;  // one launch per source
    else if (b2 < pl.bno) pl.bno = b2;
  }
  if (pl.bno == 32) pl.bmo = 32;
  if (pl.bmo == 32 && pl.bno == 64) pl.bno = 32;
  pl.variant = pl.bmo == 128 ? (pl.bno == 128 ? V128x128 : V128x64)
               : pl.bmo == 64 ? (pl.bno == 128 ? V64x128 : V64x64)
                              : (pl.bno == 128 ? V32x128 : V32x32);
  // 256 couts x 128 cins by 8 waves (round 2): three quarters of the 128x128 tile's LDS-DMA bytes per multiply-add, two waves
  // per SIMD inside ONE block (the tap-per-block launches run one block per CU); -0.15 ms on the bf16 bs-32 step
  if (!pl.phase && !pl.bno2 && pl.variant == V128x128 && d->Cout % 256 == 0) {
    pl.bmo = 256;
    pl.variant = V256x128;
  }
  pl.taps = pl.phase ? 16 : d->kh * d->kw;
  pl.tiles_ci = (pl.bno2 ? d->C1 : d->C1 + d->C2) / pl.bno;
  pl.tiles_ci2 = pl.bno2 ? d->C2 / pl.bno2 : 0;
  pl.variant2 = pl.bmo == 128 ? (pl.bno2 == 128 ? V128x128 : V128x64) : (pl.bno2 == 128 ? V64x128 : V64x64);
  pl.K = (long)pl.taps * (d->C1 + d->C2);
  pl.tiles_co = d->Cout / pl.bmo;
  pl.tiles_k = pl.taps * pl.tiles_ci;
  // conv_wgrad_phase4_bf16 (round 5; opt-in, knob wgrad_phase4 = 1): the phase form's 128 x 128 launches -- a quarter of the blocks, four
  // products each.  Not the default: see the note at RsKnobs::wgrad_phase4.
  pl.phase4 = (pl.phase && pl.variant == V128x128 && rs_knobs().wgrad_phase4 != 0) ? 1 : 0;
  const long tiles = pl.phase4 ? (long)pl.tiles_co * 4 * pl.tiles_ci  // (a second, narrower source follows the split count of the first)
                               : (long)pl.tiles_co * pl.taps * (pl.tiles_ci + pl.tiles_ci2);
  pl.pk = 64;  // pixels per chunk (32 was measured for the short reductions: no gain)
  const int PK = pl.pk;
  const long chunks = (M + PK - 1) / PK;
  // blocks a launch aims at (knobs wgrad_blocks / wgrad_blocks_phase, seeded by RS_WGRAD_BLOCKS / RS_WGRAD_BLOCKS_PHASE).  Measured on the bf16 bs-32 step
  // (scripts/wgrad_blocks_ab.sh; step time, not kernel time, is what counts: these launches run on the side stream BESIDE
  // the data-gradient chain).  The tap-per-block launches of the encoder: in isolation they are fastest at 512 blocks (a
  // block is prologue + few chunks + a 64 KB partial tile, and every extra split is another partial to write and reduce),
  // but one block per CU (256) leaves room for the main stream's kernels and gives the shortest step (25.3-25.4 ms vs 25.6
  // at 512 and 26.1 at the former 1024).  Round 5 (profiles/r05/wgrad_blocks_bf16.txt, median step): 22.64-22.70 ms at 256,
  // 22.51 / 22.52 at 192 / 128, 22.72 at 384 -> 192 (SHIPPED).  With the ring of three chunk buffers (opt-in, see RING above) (RING = 3: a lone block keeps its CU's
  // LDS-DMA path busy; the launches are 15-30 % shorter in isolation) fewer, longer blocks are the better neighbours: 22.20-22.31
  // at 96-112 against 22.32-22.39 for two buffers at 192, five alternating pairs (profiles/r05/wgrad_ring.txt) -> 96.  The phase
  // form's 16-tap launches are long reductions and want more, shorter blocks (1024 / 1536 / 2048: the same step time).
  const long target = pl.phase4 ? rs_knobs().wgrad_blocks_phase4 : (pl.phase ? rs_knobs().wgrad_blocks_phase : rs_knobs().wgrad_blocks);
  long s = (target + tiles - 1) / tiles;          // aim at >= `target` blocks ...
  const long smax = (chunks * PK / 64 + 7) / 8;   // ... of at least 512 pixels each
  if (s > smax) s = smax;
  if (s < 1) s = 1;
  // 32-bit byte offsets inside a split: shrink the splits until dy and the input both fit
  const long cmax = d->C1 > d->C2 ? d->C1 : d->C2;
  const long img_bytes = (long)d->Hs * d->Ws * cmax * 2;
  const long howo = pl.phase ? (long)d->Hs * d->Ws : (long)d->Ho * d->Wo;
  const long dimg_bytes = (long)d->Ho * d->Wo * d->Cout * 2;
  for (;;) {
    pl.chunks_per_split = (int)((chunks + s - 1) / s);
    const long px = ((long)pl.chunks_per_split + 2) * PK;
    const long span_dy = pl.phase ? (px / howo + 2) * dimg_bytes : px * d->Cout * 2;
    const long span_x = (px / howo + 2) * img_bytes;
    if ((span_dy < (1L << 31) && span_x < (1L << 31)) || pl.chunks_per_split == 1) break;
    s *= 2;
  }
  pl.splits = (int)((chunks + pl.chunks_per_split - 1) / pl.chunks_per_split);
  return pl;
}

}  // namespace

// Which formulation rs_conv2d_wgrad_bf16 uses for `d` (for the roofline report): 0 tap-per-block, 1 all-taps thin kernel,
// 2 phase form (executes 4/9 of the algorithmic multiply-adds).
extern "C" int rs_conv2d_wgrad_bf16_form(const rs_conv_desc* d) {
  if (!valid(d)) return RS_EINVAL;
  int tb = 0, ts = 0;
  if (rs_wgrad_thin_plan(d, &tb, &ts)) return 1;
  return phase_ok(d) ? 2 : 0;
}

// The tile rs_conv2d_wgrad_bf16 launches for `d` (for per-symbol reports: the kernel is instantiated per tile), as
// (couts << 16) | cins of the first launch, | (cins of the second source's launch) << 8 when the two concat sources run
// separate launches; 0 for the all-taps thin kernel.
extern "C" int rs_conv2d_wgrad_bf16_tile(const rs_conv_desc* d) {
  if (!valid(d)) return RS_EINVAL;
  int tb = 0, ts = 0;
  if (rs_wgrad_thin_plan(d, &tb, &ts)) return 0;
  const Plan pl = plan(d);
  return (pl.bmo << 16) | (pl.bno2 << 8) | pl.bno;
}

extern "C" long rs_conv2d_wgrad_bf16_workspace_bytes(const rs_conv_desc* d) {
  if (!valid(d)) return RS_EINVAL;
  int tb = 0, tslices = 0;
  if (rs_wgrad_thin_plan(d, &tb, &tslices)) {
    const long n = (long)d->Cout * 9 * d->C1;
    return (tslices * n + rs_reduce_scratch_floats(n, tslices)) * (long)sizeof(float);
  }
  const Plan pl = plan(d);
  const long n = (long)d->Cout * pl.K;
  return (pl.splits * n + rs_reduce_scratch_floats(n, pl.splits) + (pl.phase ? n : 0)) * (long)sizeof(float);
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
      const long n = (long)d->Cout * 9 * d->C1;
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
  a.div_howo = rs_make_fastdiv((unsigned)(pl.phase ? d->Hs * d->Ws : d->Ho * d->Wo));
  a.div_wo = rs_make_fastdiv((unsigned)(pl.phase ? d->Ws : d->Wo));
  a.Hv = d->ups == 0 ? d->Hs : (d->ups == 1 ? 2 * d->Hs : 2 * d->Hs - 1);
  a.Wv = d->ups == 0 ? d->Ws : (d->ups == 1 ? 2 * d->Ws : 2 * d->Ws - 1);
  a.kw = d->kw;
  a.stride = d->stride;
  a.pad = d->pad;
  a.Ho = d->Ho;
  a.Wo = d->Wo;
  a.Cout = d->Cout;
  a.M = (int)(pl.phase ? (long)d->N * d->Hs * d->Ws : (long)d->N * d->Ho * d->Wo);
  a.K = (int)pl.K;
  a.tiles_co = pl.tiles_co;
  a.chunks_per_split = pl.chunks_per_split;
  a.Ctot = d->C1 + d->C2;
  hipStream_t s = (hipStream_t)stream;
  // one launch, or one per concat source (Plan::bno2): each sees "its" source as a single-source problem whose channels
  // start at ci_base inside the concat
  const int nseg = pl.bno2 ? 2 : 1;
  const bool ring3 = rs_knobs().wgrad_ring == 3;  // (tap-per-block launches only: the phase form runs several blocks per CU)
  const bool ring4 = rs_knobs().wgrad_ring >= 4 && rs_knobs().wgrad_ring <= 7;  // (four buffers, reads one iteration behind the wait; tiles up to 128 wide)
  const int dead = rs_knobs().wgrad_ring - 4;  // 5, 6, 7: the DEAD variants of the 128 x 64 tile (the race screen's bisect)
  for (int seg = 0; seg < nseg; ++seg) {
    int variant = pl.variant;
    a.ci_base = 0;
    a.tiles_ci = pl.tiles_ci;
    if (pl.bno2) {
      a.src1 = reinterpret_cast<const bf16_t*>(seg == 0 ? src1 : src2);
      a.src2 = nullptr;
      a.C1 = seg == 0 ? d->C1 : d->C2;
      a.C2 = 0;
      a.ci_base = seg == 0 ? 0 : d->C1;
      a.tiles_ci = seg == 0 ? pl.tiles_ci : pl.tiles_ci2;
      variant = seg == 0 ? pl.variant : pl.variant2;
    }
    a.tiles_k = pl.taps * a.tiles_ci;
    const int grid = pl.tiles_co * a.tiles_k * pl.splits;
    if (pl.phase4 && variant == V128x128) {
      a.tiles_k = 4 * a.tiles_ci;  // planes x Cin tiles
      conv_wgrad_phase4_bf16<<<pl.tiles_co * a.tiles_k * pl.splits, 512, 0, s>>>(a);
    } else if (pl.phase) {
      switch (variant) {
        case V128x128: conv_wgrad_bf16<128, 128, 2, 2, 64, true><<<grid, 256, 0, s>>>(a); break;
        case V128x64: conv_wgrad_bf16<128, 64, 2, 2, 64, true><<<grid, 256, 0, s>>>(a); break;
        case V64x128: conv_wgrad_bf16<64, 128, 2, 2, 64, true><<<grid, 256, 0, s>>>(a); break;
        case V64x64: conv_wgrad_bf16<64, 64, 2, 2, 64, true><<<grid, 256, 0, s>>>(a); break;
        case V32x128: conv_wgrad_bf16<32, 128, 1, 4, 64, true><<<grid, 256, 0, s>>>(a); break;
        case V32x32: conv_wgrad_bf16<32, 32, 1, 1, 64, true><<<grid, 64, 0, s>>>(a); break;
        default: return RS_EINVAL;
      }
    } else {
      switch (variant) {
        case V128x128:
          if (ring3) conv_wgrad_bf16<128, 128, 2, 2, 64, false, 3><<<grid, 256, 0, s>>>(a);
          else conv_wgrad_bf16<128, 128, 2, 2, 64, false><<<grid, 256, 0, s>>>(a);
          break;
        case V128x64:
#ifdef RS_EXP_BUILD  // (`make EXP=1`: the bisect variants of scripts/dma_order_bisect.py)
          if (ring4 && dead == 1) conv_wgrad_bf16<128, 64, 2, 2, 64, false, 4, 1><<<grid, 256, 0, s>>>(a);
          else if (ring4 && dead == 2) conv_wgrad_bf16<128, 64, 2, 2, 64, false, 4, 2><<<grid, 256, 0, s>>>(a);
          else if (ring4 && dead == 3) conv_wgrad_bf16<128, 64, 2, 2, 64, false, 4, 3><<<grid, 256, 0, s>>>(a);
          else
#endif
          if (ring4) conv_wgrad_bf16<128, 64, 2, 2, 64, false, 4><<<grid, 256, 0, s>>>(a);
          else if (ring3) conv_wgrad_bf16<128, 64, 2, 2, 64, false, 3><<<grid, 256, 0, s>>>(a);
          else conv_wgrad_bf16<128, 64, 2, 2, 64, false><<<grid, 256, 0, s>>>(a);
          break;
        case V64x128:
          if (ring3) conv_wgrad_bf16<64, 128, 2, 2, 64, false, 3><<<grid, 256, 0, s>>>(a);
          else conv_wgrad_bf16<64, 128, 2, 2, 64, false><<<grid, 256, 0, s>>>(a);
          break;
        case V64x64:
          if (ring3) conv_wgrad_bf16<64, 64, 2, 2, 64, false, 3><<<grid, 256, 0, s>>>(a);
          else conv_wgrad_bf16<64, 64, 2, 2, 64, false><<<grid, 256, 0, s>>>(a);
          break;
        case V32x128: conv_wgrad_bf16<32, 128, 1, 4, 64, false><<<grid, 256, 0, s>>>(a); break;
        case V32x32: conv_wgrad_bf16<32, 32, 1, 1, 64, false><<<grid, 64, 0, s>>>(a); break;
        case V256x128:
          if (ring3) conv_wgrad_bf16<256, 128, 4, 2, 64, false, 3><<<grid, 512, 0, s>>>(a);
          else conv_wgrad_bf16<256, 128, 4, 2, 64, false><<<grid, 512, 0, s>>>(a);
          break;
        default: return RS_EINVAL;
      }
    }
    const int rcl = RS_LAUNCH_RESULT();
    if (rcl) return rcl;
  }
  const long n = (long)d->Cout * pl.K;  // multiple of 4
  float* scratch = a.out + (long)pl.splits * n;
  if (!pl.phase) return rs_reduce_splits(a.out, dw, n, pl.splits, scratch, stream);
  float* gbuf = scratch + rs_reduce_scratch_floats(n, pl.splits);  // G [Cout][16][Cin]
  const int rc2 = rs_reduce_splits(a.out, gbuf, n, pl.splits, scratch, stream);
  if (rc2) return rc2;
  const long total = (long)d->Cout * 9 * (d->C1 + d->C2);
  combine_phase_wgrad_kernel<<<rs_cdiv(total, 256), 256, 0, s>>>(gbuf, dw, d->C1 + d->C2, total);
  return RS_LAUNCH_RESULT();
}

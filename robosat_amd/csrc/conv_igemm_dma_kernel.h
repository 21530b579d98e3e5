// conv_igemm_dma_kernel.h -- the LDS-DMA implicit-GEMM convolution kernel of conv_igemm_dma.hip, as a header: one
// translation unit per (activation type, direct | phase form, epilogue kind) instantiates its tiles and row sizes
// (conv_igemm_dma_<type>_<form>.hip; `make -j` compiles them in parallel) and exports ONE launcher each, declared at the
// bottom of this file for the dispatcher in conv_igemm_dma.hip.  See that file's header for the design.
#pragma once
#include <stdlib.h>

#include <type_traits>
#include <utility>

#include "common.h"

enum { EPI_EVAL = 0, EPI_STATS = 1, EPI_BWD = 2 };
constexpr int kMaxK = 4;  // filter height / width up to 4 (the 4x4 stride-2 data gradient of an upsampled 3x3)
// (index 4 is the fp32 stem kernel of conv_igemm.hip: the two files share the index space of rs_conv2d_tile_name)
// T256x256: 8-wave blocks, one per CU, bf16 only (see pick_tile); T256x128: 8 waves as 4 x 2 (64x64 wave tiles)
// TTHIN: the all-taps kernels of conv_thin_bf16.hip (the 32-channel decoder tail in bf16), reported through the same index space
// THALO: the halo-once forms of this kernel (HALO template parameter below; bf16, 8 waves, a 2-D patch of 8 x 32 pixels
// per block), reported through the same index space
// TEW: conv1x1_ew_f32.hip (fp32 1x1 launches with the epilogue on its own waves), reported through the same index space
enum Tile { T128x128 = 0, T128x64, T128x32, T64x64, TSTEM_RESERVED, T256x128, T256x256, TTHIN, THALO, TEW, NTILES };
// HALO forms: the block's rows are a 2-D PATCH of the output grid; per K-group (one 128-byte channel chunk of one source
// plane) the patch's source HALO lands in LDS once and every filter tap reads it at a row offset -- the implicit-GEMM
// form above fetches each source pixel once per tap.  HALO_33: 3x3 / stride 1 / pad 1 (9 taps);  HALO_PHASE: one output
// parity of the DecoderBlock phase form (2x2 taps on the source grid);  HALO_DG4: the 4x4 / stride-2 data gradient of the
// phase form = four 2x2 convolutions, one per parity plane of dz, accumulated into the same output patch (16 taps).
enum { HALO_NONE = 0, HALO_33 = 1, HALO_PHASE = 2, HALO_DG4 = 3 };

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
  // optional, instead of `mask` in those launches: the ReLU mask as one BIT per element (bit e of byte i covers element
  // 8*i + e of the [M][Cout] tensor; rs_bn_apply_bits_dt writes it next to z).  The epilogue of a bottleneck's conv1 data
  // gradient moves four Cout-wide operands; the mask as bits is a sixteenth of one of them.
  const unsigned char* mask_bits;
  // optional second destination (the torch.cat split of a decoder data gradient): couts [0, csplit) go to `out` (row
  // stride csplit, `mask`), couts [csplit, Cout) to `out2` (row stride Cout - csplit, `mask2`); csplit % BN == 0
  T* out2;
  const T* mask2;
  int csplit;
  int N, Hs, Ws, C1, C2, Hv, Wv, ups;
  int kh, kw, stride, pad, Ho, Wo, Cout;
  int M, cpt, nk, Kw, relu, ntiles, ntaps, phase4;
  int tpx, tpi;  // HALO forms: patches per row of the grid the rows enumerate / per image (nk = K-groups: planes x chunks)
};


// One launcher per translation unit: the tile (enum Tile) and K-chunk row size (64 | 128) pick the instantiation.
#define RS_CONV_LAUNCHER(name, T) \
  __attribute__((visibility("hidden"))) void name(int tile, int rowb, int grid, hipStream_t s, const ConvArgsT<T>& a)
RS_CONV_LAUNCHER(rs_conv_launch_f32_plain_eval, float);
RS_CONV_LAUNCHER(rs_conv_launch_f32_plain_stats, float);
RS_CONV_LAUNCHER(rs_conv_launch_f32_plain_bwd, float);
RS_CONV_LAUNCHER(rs_conv_launch_f32_phase_eval, float);
RS_CONV_LAUNCHER(rs_conv_launch_bf16_plain_eval, bf16_t);
RS_CONV_LAUNCHER(rs_conv_launch_bf16_plain_stats, bf16_t);
RS_CONV_LAUNCHER(rs_conv_launch_bf16_plain_bwd, bf16_t);
RS_CONV_LAUNCHER(rs_conv_launch_bf16_phase_eval, bf16_t);
// halo-once forms (bf16): `tile` = BN (128 | 64), `rowb` = the epilogue kind (EPI_*) for the 3x3 form
RS_CONV_LAUNCHER(rs_conv_launch_bf16_halo33, bf16_t);
RS_CONV_LAUNCHER(rs_conv_launch_bf16_halo_phase, bf16_t);
RS_CONV_LAUNCHER(rs_conv_launch_bf16_halo_dg4, bf16_t);
__attribute__((visibility("hidden"))) void rs_conv_launch_bf16_halo_phase_ko(int ko, int grid, hipStream_t s, const ConvArgsT<bf16_t>& a);
// conv1x1_ew_f32.hip: fp32 1x1 / stride-1 launches with the epilogue on its own waves (knob conv1x1_ew: by rule K <= 64)
__attribute__((visibility("hidden"))) int rs_conv1x1_ew_f32_ok(const rs_conv_desc* d);
__attribute__((visibility("hidden"))) int rs_conv1x1_ew_f32_launch(const ConvArgsT<float>& a, hipStream_t s);
// conv1x1_np_f32.hip: fp32 1x1 / stride-1 launches with a sub-tile's epilogue between the next sub-tile's MFMAs (knob conv1x1_np)
__attribute__((visibility("hidden"))) int rs_conv1x1_np_f32_ok(const rs_conv_desc* d);
__attribute__((visibility("hidden"))) int rs_conv1x1_np_f32_launch(const ConvArgsT<float>& a, hipStream_t s);
// conv1x1_ew_bf16.hip: the train-mode forward's bf16 1x1 launches (statistics epilogue) in the same layout (knob conv1x1_ew_bf16)
__attribute__((visibility("hidden"))) int rs_conv1x1_ew_bf16_stats_ok(const rs_conv_desc* d);
__attribute__((visibility("hidden"))) int rs_conv1x1_ew_bf16_stats_launch(const ConvArgsT<bf16_t>& a, hipStream_t s);

#ifdef RS_CONV_INSTANTIATE  // ---- kernel + launcher body: only in the instantiating translation units --------------------
namespace {

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
  // (clipped below kDmaOOB: every offset the kernels form is < 2^31, and the out-of-range sentinel must stay out of range)
  const unsigned int n = bytes > 0xFFFE0000L ? 0xFFFE0000u : (unsigned int)(bytes < 0 ? 0 : bytes);
  // The inputs are wave-uniform (kernel arguments and blockIdx arithmetic) but 64-bit multiplies, integer divisions and
  // the clamp above run on the VALU: hipcc then carries the descriptor in VGPRs and only sometimes moves it back (it did not
  // once a select between two descriptors was itself lowered to v_cndmask: "invalid operand" in the LDS-DMA asm, whose
  // SRSRC must be SGPRs).  readfirstlane on the descriptor's INPUTS makes the uniformity provable (cdna_hip_programming.md T20).
  const unsigned long b = (unsigned long)base;
  const unsigned int lo = __builtin_amdgcn_readfirstlane((unsigned int)b), hi = __builtin_amdgcn_readfirstlane((unsigned int)(b >> 32));
  const unsigned int nn = __builtin_amdgcn_readfirstlane(n);
  return __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<void*>(((unsigned long)hi << 32) | lo), 0, (int)nn, 0x00020000);
}

// One LDS-DMA wave instruction (buffer_load_dwordx4 ... lds): lane l's 16 bytes at buffer offset `voff` + `soff` land at
// LDS byte `lds_dst` + 16*l (lds_dst wave-uniform, in M0); an out-of-range offset => zeros (scripts/probes/probe_glds.hip).
// Inline asm on purpose: through the builtin hipcc cannot tell that the DMA's destination (the OTHER pipeline buffer) is
// disjoint from the fragment reads that follow and drains the queue (s_waitcnt vmcnt(0)) before the first ds_read of
// every chunk.  As asm the copy is invisible to its counters, so the kernel waits itself (rb_dma_wait) ahead of the
// barrier that publishes the buffer.
// `soff` is a wave-uniform byte offset added to the address (the SOFFSET operand): the per-lane offsets of a
// (tap, source) stay in registers and the K loop advances through the channels with one SGPR.  kDmaOOB + soff is out of
// range for every rsrc rb_make_rsrc builds (zeros land in the LDS): the per-lane offset of a padding / past-the-end row.
// m0 is declared clobbered rather than saved and restored (two SALU per piece, eight pieces per chunk per wave): the
// compiler itself only touches m0 for M0-operand LDS builtins and dynamically indexed register arrays, and this file has
// neither (every register array is indexed by unrolled constants) -- check `grep m0` of the ISA after touching the kernel.
constexpr int kDmaOOB = (int)0xFFFF0000u;
__device__ __forceinline__ void rb_dma16s(__amdgpu_buffer_rsrc_t r, unsigned int lds_dst, int voff, int soff) {
  asm volatile(
      "s_mov_b32 m0, %1\n\t"
      "s_nop 0\n\t"
      "buffer_load_dwordx4 %0, %2, %3 offen lds"
      :
      : "v"(voff), "s"(lds_dst), "s"(r), "s"(soff)
      : "memory", "m0");
}

__device__ __forceinline__ void rb_dma_wait() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
template <int N>
__device__ __forceinline__ void rb_dma_wait_n() {  // all but the N youngest of this wave's DMA instructions have landed
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}
// f(std::integral_constant<int, 0>()), f(<1>), ...: an unrolled loop whose index is a constant EXPRESSION (the counted
// s_waitcnt immediates of the halo forms depend on the tap)
template <class F, int... I>
__device__ __forceinline__ void rb_for_each(F&& f, std::integer_sequence<int, I...>) {
  (f(std::integral_constant<int, I>()), ...);
}
// halo pieces per wave are spread over the taps 0 .. ntap-3 of the PREVIOUS K-group (the halo is published one step before
// the group that reads it starts): piece j goes with tap j*(ntap-2)/njh
constexpr int halo_tap_of_piece(int j, int ntap, int njh) { return j * (ntap - 2) / njh; }
constexpr int halo_pieces_at_tap(int t, int ntap, int njh) {
  int n = 0;
  for (int j = 0; j < njh; ++j) n += halo_tap_of_piece(j, ntap, njh) == t ? 1 : 0;
  return n;
}
__device__ __forceinline__ unsigned int rb_lds_addr(const void* p) {
  return (unsigned int)(unsigned long)(__attribute__((address_space(3))) const void*)p;
}


// (second launch bound = waves per SIMD the LDS footprint admits, capped at 4: the register allocation must not be what
// limits the blocks per CU of the short-K layers, whose co-resident blocks are what hides a block's DMA round trips)
constexpr int conv_waves_per_simd(int nw, int bm, int bn, int rowb, int es, int epi, int halo = 0) {
  if (halo) return 2;  // one 8-wave block per CU (the halo ring + weight ring fill the LDS)
  const int lds = 2 * (bm + bn) * rowb + 9 * bm * 4;
  const int blocks = 160 * 1024 / lds;
  const int w = blocks * nw / 4;
  // 8-wave tiles: 128 (256x256: 256) accumulator + operand registers; the fp32 data gradient into a BatchNorm holds three
  // fp32 epilogue operands per row (not a benchmarked path: no spills matter more than its occupancy)
  const int cap = nw == 8 ? (bm * bn >= 256 * 256 ? 1 : 2) : ((es == 4 && epi == 2 && bm * bn >= 128 * 128) ? 2 : 4);
  return w < 1 ? 1 : (w > cap ? cap : w);
}

// EPI selects the epilogue a launch needs, so that each instantiation carries only its own registers and loads:
//   EPI_EVAL  : scale/shift, residual, ReLU, ReLU mask, two-destination store (predict, and every plain data gradient)
//   EPI_STATS : raw output + per-tile BatchNorm partial sums (sum y, sum y^2): the train-mode forward
//   EPI_BWD   : residual, ReLU mask, + partial sums (sum g, sum g * xhat) against bn_y: data gradient into a BatchNorm
// KO (measurement builds only, conv_halo_ko.hip; results are WRONG for KO != 0): knock-outs of the halo main loop that say
// where its time goes -- 1: no waits / barriers, 2: no DMA, 3: no fragment reads (one set reused), 4: no MFMAs.
template <typename T, int BM, int BN, int WGM, int WGN, int ROWB, bool PHASE, int EPI, int HALO = HALO_NONE, int KO = 0>
__global__ __launch_bounds__(64 * WGM * WGN, conv_waves_per_simd(WGM * WGN, BM, BN, ROWB, (int)sizeof(T), EPI, HALO)) void conv_igemm_dma(
    const ConvArgsT<T> p) {
  static_assert(WGM * WGN == 4 || WGM * WGN == 8, "4 or 8 waves per block");
  static_assert(ROWB == 128 || ROWB == 64, "a K-chunk is a 128- or 64-byte row");
  constexpr int NBUF = 2;  // pipeline buffers (3 and 4 with counted vmcnt waits were measured: no gain)
  constexpr int NW = WGM * WGN;  // waves; the 8-wave blocks (256- and 512-row tiles) run one per CU
  constexpr int NT = 64 * NW;
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
  // ---- halo forms: geometry of the patch, its halo, the weight ring (see the HALO enum)
  constexpr bool HL = HALO != HALO_NONE;
  constexpr int PWD = 32, PHT = BM / 32;           // patch: PHT rows x 32 pixels (a 32-pixel MFMA sub-tile = one patch row)
  constexpr int HK = HALO == HALO_33 ? 3 : 2;      // taps per dimension of one K-group
  constexpr int HNT = HK * HK;                     // taps per K-group
  constexpr int HWD = PWD + HK - 1, HHT = PHT + HK - 1, HROWS = HHT * HWD;  // halo: HHT x HWD source pixels
  constexpr int NJH = ((HROWS + RI - 1) / RI + NW - 1) / NW;  // halo DMA instructions per wave (every wave issues exactly NJH)
  constexpr int HALOB = ((HROWS + RI - 1) / RI) * 1024;  // bytes per halo buffer (two of them): whole DMA instructions
  constexpr int HSCRATCH = 2 * HALOB;              // one KiB for the zeros of the DMA instructions past the halo's last row
  constexpr int HBOFF = 2 * HALOB + 1024;          // the weight ring
  constexpr int DRING = 4;                         // weight ring: step g in slot g & 3 (weights are issued three steps ahead)
  constexpr int BSLOT = BN * ROWB;                 // one (tap, chunk) of weights
  constexpr int NBW = IB / NW;                     // weight DMA instructions per wave per step
  constexpr int PIPE = HL ? HBOFF + DRING * BSLOT : NBUF * BUF;
  constexpr int STAGE = WGM * 32 * LDO * 4;  // staging: one 32-row sub-tile per wave row at a time
  constexpr int MAINB = PIPE > STAGE ? PIPE : STAGE;
  constexpr int TABN = HL ? BM : (2 * kMaxK + 1) * BM;  // separable gather table (tap row | tap column) x tile row + output rows
  static_assert(TM >= 1 && TN >= 1, "wave tile must hold one 32x32 MFMA tile");
  static_assert((IA % NW) == 0 && IB >= 1 && (BM % RI) == 0 && (BN % RI) == 0, "DMA instruction split");
  static_assert(!HL || (sizeof(T) == 2 && NW == 8 && (IB % NW) == 0 && WM % 32 == 0 && HNT >= 4 && KS % 2 == 0 && (HALO == HALO_PHASE) == PHASE),
                "halo forms: bf16, 8 waves, whole weight DMA instructions per wave");
  static_assert(MAINB + TABN * 4 <= 160 * 1024, "LDS");

  __shared__ __attribute__((aligned(16))) unsigned char smem[MAINB + TABN * 4];
  int* taby = reinterpret_cast<int*>(smem + MAINB);  // [kh][BM]: ((n - nfirst)*Hs + iy) * Ws, or -1
  int* tabx = taby + kMaxK * BM;                     // [kw][BM]: ix, or -1
  int* orow = HL ? taby : tabx + kMaxK * BM;         // [BM]: output pixel index of the row, or -1 past M (halo forms: the only table)

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
  // (an integer division runs on the VALU: tell the compiler the quotient is wave-uniform, or the buffer descriptors built
  // from it end up in VGPRs and the LDS-DMA asm cannot take them)
  // halo forms: M tile mt = patch (image hn, origin (hy0, hx0) on the Hd x Wd grid); the descriptors start at that image
  int hn = 0, hy0 = 0, hx0 = 0;
  if constexpr (HL) {
    hn = __builtin_amdgcn_readfirstlane(mt / p.tpi);
    const int rem = mt - hn * p.tpi;
    const int ty = __builtin_amdgcn_readfirstlane(rem / p.tpx);
    hy0 = ty * PHT;
    hx0 = (rem - ty * p.tpx) * PWD;
  }
  const int nfirst = HL ? hn : __builtin_amdgcn_readfirstlane(m0 / HdWd);
  const int ush = p.ups ? 1 : 0;
  const int upar = p.ups == 2 ? 1 : 0;
  const int pad_y = PHASE ? 1 - py : p.pad, pad_x = PHASE ? 1 - px : p.pad;

  // ---- separable gather table, relative to the tile's first image: source pixel of (row, tap (r, s)) =
  //      taby[r][row] + tabx[s][row] when both are >= 0, else the tap contributes zeros ------------------------------
  if constexpr (HL) {
    for (int e = tid; e < BM; e += NT) {  // row e of the tile = patch pixel (e / 32, e % 32)
      const int y = hy0 + e / PWD, x = hx0 + e % PWD;
      int v = -1;
      if (y < Hd && x < Wd) v = PHASE ? (hn * p.Ho + 2 * y + py) * p.Wo + 2 * x + px : (hn * Hd + y) * Wd + x;
      orow[e] = v;
    }
  }
  for (int e = tid; e < (HL ? 0 : (p.kh + p.kw + 1) * BM); e += NT) {
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
  constexpr int NIA = IA / NW;  // this wave's pieces j < NIA copy pixel rows, the others weight rows
  int wrow[NI];                 // byte offset of this lane's piece in weight row (n0 + RI*jj + ra), chunk 0
#pragma unroll
  for (int j = 0; j < NI; ++j) wrow[j] = ((n0 + RI * (wave + NW * j - IA) + ra) * p.Kw + gp * EPP) * ES;
  __syncthreads();

  // Per-lane byte offsets of the pixel pieces for the (tap, source) being fetched, chunk 0 of it; kDmaOOB for a padding
  // row.  Rebuilt from the gather tables only when the tap or the concat source changes -- the chunks in between differ
  // by the wave-uniform channel offset alone, which rides in the DMA's SOFFSET.
  int pbase[NIA];
  auto load_tap = [&](int (&dst)[NIA], int r, int s_, bool first) __attribute__((always_inline)) {
    const int cs2 = (first ? p.C1 : p.C2) * ES;
#pragma unroll
    for (int j = 0; j < NIA; ++j) {
      const int row = RI * (wave + NW * j) + ra;
      const int y = taby[r * BM + row], x = tabx[s_ * BM + row];
      dst[j] = (y | x) < 0 ? kDmaOOB : (y + x) * cs2 + gp * 16;
    }
  };
  // state of the chunk being fetched (wave-uniform): tap row / column, channel chunk, linear index; and what its pieces
  // need: destination buffer, source descriptor, channel byte offset within the source, weight byte offset.
  // (K order = the weights' memory order: taps outer, channel chunks inner.  Channel chunks outer / taps inner was tried
  // for the phase form -- same fabric traffic per launch (FETCH_SIZE), same time -- and dropped: its summation order
  // depends on the chunk size, which depends on the batch, and predictions must not.)
  int lr = 0, ls = 0, lc = 0, lk = 0;
  unsigned int fL = lds0;
  __amdgpu_buffer_rsrc_t frs = rsrc1;
  int fsa = 0, fsb = 0;
  auto begin_chunk = [&](int buf) __attribute__((always_inline)) {
    const int c0 = lc * KC;
    const bool first = c0 < p.C1;
    if (lc == 0 || c0 == p.C1) load_tap(pbase, lr, ls, first);  // (uniform) new tap, or the second concat source begins
    fL = lds0 + buf * BUF;
    frs = first ? rsrc1 : rsrc2;
    fsa = (first ? c0 : c0 - p.C1) * ES;
    fsb = lk * ROWB;
    ++lk;  // advance to the following chunk
    ++lc;
    const int w1 = (lc == p.cpt) ? 1 : 0;
    lc = w1 ? 0 : lc;
    ls += w1;
    const int w2 = (ls == p.kw) ? 1 : 0;
    ls = w2 ? 0 : ls;
    lr += w2;
  };
  auto issue_piece = [&](int j) __attribute__((always_inline)) {  // j: compile-time after unrolling
    const int ii = wave + NW * j;  // wave-uniform; IA % NW == 0, so the role depends on j alone
    if (j < NIA) {
      rb_dma16s(frs, fL + ii * 1024, pbase[j], fsa);
    } else if ((IB % NW) == 0 || ii < IA + IB) {
      rb_dma16s(rsrcw, fL + ii * 1024, wrow[j], fsb);
    }
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
  // ---- main loop: chunk k+1 streams HBM -> LDS by DMA while the MFMAs of chunk k run; one barrier per chunk.  Each wave
  //      first waits for ITS OWN DMA instructions of chunk k, the barrier publishes everybody's, and only then the buffer
  //      freed by chunk k-1 is refilled.
  if constexpr (!HL) {
  if (p.nk > 0) {
    begin_chunk(0);
#pragma unroll
    for (int q = 0; q < NI; ++q) issue_piece(q);
  }
  }
  // The pieces of chunk kc+1 are issued BETWEEN the MFMAs of chunk kc (one piece every PSTEP MFMAs from the start of the
  // chunk): an LDS-DMA instruction costs the issuing wave 60-180 cycles, which a burst at the top of the chunk would add
  // to the critical path of every wave of the block at once.
  constexpr int NMMA = KS * TM * TN;
  constexpr int PSTEP = NMMA / (2 * NI) >= 1 ? NMMA / (2 * NI) : 1;  // front-loaded: the tail of the chunk covers the latency
  constexpr int PIN = (NMMA + PSTEP - 1) / PSTEP < NI ? (NMMA + PSTEP - 1) / PSTEP : NI;  // pieces placed between MFMAs
  // One chunk: wait for it, publish it, then its MFMAs -- with the pieces of chunk kc+1 in between when FETCH.  Two loops
  // (steady state with FETCH, then the last chunk without) rather than a branch per piece: a diamond inside one loop
  // made hipcc keep the 64 accumulator registers of the two arms apart (64 v_mov per chunk).
  auto chunk = [&](int kc, auto fetch_tag) __attribute__((always_inline)) {
    constexpr bool FETCH = decltype(fetch_tag)::value;
    rb_dma_wait();
    __syncthreads();
    if (FETCH) begin_chunk((kc + 1) % NBUF);  // refills the buffer chunk kc-1 was read from
    const unsigned char* L = smem + (kc % NBUF) * BUF;
    u32x4 fa[2][TM], fb[2][TN];
    read_frag(L, 0, fa[0], fb[0]);
#pragma unroll
    for (int s = 0; s < KS; ++s) {
      if (s + 1 < KS) read_frag(L, s + 1, fa[(s + 1) & 1], fb[(s + 1) & 1]);
#pragma unroll
      for (int tm = 0; tm < TM; ++tm)
#pragma unroll
        for (int tn = 0; tn < TN; ++tn) {
          const int q = (s * TM + tm) * TN + tn;  // MFMA index within the chunk (compile-time after unrolling)
          if (FETCH && q % PSTEP == 0 && q / PSTEP < PIN) issue_piece(q / PSTEP);
          mma16(acc[tn][tm], fb[s & 1][tn], fa[s & 1][tm], T());
        }
    }
    if (FETCH) {
#pragma unroll
      for (int q = PIN; q < NI; ++q) issue_piece(q);  // (tiles with fewer MFMAs per chunk than pieces)
    }
  };
  if constexpr (!HL) {
    int kc = 0;
    for (; kc < p.nk - 1; ++kc) chunk(kc, std::true_type());
    for (; kc < p.nk; ++kc) chunk(kc, std::false_type());
  } else {
    // ================================ halo-once main loop ===================================================================
    // K = K-groups q (a 128-byte channel chunk of one source plane; HALO_DG4: 4 planes x chunks, else the chunks of the
    // concat sources) x HNT taps; step g = q * HNT + t.  LDS: two halo buffers (group q in q & 1) + a ring of DRING = 4
    // weight tiles (step g in g & 3).  The wait + barrier at the top of step g publish the weights of step g + 1 (and, at a
    // group's last tap, the next group's halo): one step EARLY, so that the first fragments of step g + 1 are read at the
    // end of step g and its MFMAs start right behind its barrier instead of behind an LDS round trip.  During step g the
    // weights of step g + 3 and, for t <= HNT - 3, this tap's share of the NEXT group's halo are issued between the MFMAs;
    // the counted wait at the top of every step leaves exactly the previous step's DMA instructions in flight.
    // Nothing is branched on inside the loop: past the end the pieces are still issued, out of range (zeros into buffers
    // nobody reads), so every wave issues the same count in every step -- what the counted waits rely on.
    constexpr int RPT = WM / 32;  // patch rows per wave row (= TM)
    constexpr int NHI = (HROWS + RI - 1) / RI;  // halo DMA instructions that carry rows; the surplus ones (ii >= NHI) land in a scratch KiB
    const int hi = lane >> 5, l31 = lane & 31;
    const int ctot = p.C1 + p.C2;
    // this lane's halo rows: instruction ii = wave + NW*j copies halo rows RI*ii .. (R = hyy * HWD + hxx)
    // (round 6) per piece the lane's BYTE offset in either concat source, not its pixel: the multiply by the source's channel count, the
    // swizzled piece and the padding test happen once per patch (DG4: once per parity plane) instead of once per issued piece -- the
    // main loop was issue-bound on exactly such arithmetic (3.3 VALU + 2.2 SALU per MFMA, profiles/r04/halo_variants.txt)
    int ho1[NJH], ho2[NJH];  // byte offset of the lane's row of piece j in source 1 / 2 for the group being FETCHED, kDmaOOB = zeros
    auto halo_pix = [&](int pa, int pb) __attribute__((always_inline)) {
      int hpix[NJH];
#pragma unroll
      for (int j = 0; j < NJH; ++j) {
        const int R = RI * (wave + NW * j) + ra;
        const int hyy = R / HWD, hxx = R - hyy * HWD;
        int v = -1;
        if (HALO == HALO_DG4) {  // plane (pa, pb) of dz: sub-grid pixel (u, v) = dz(2u + pa, 2v + pb); rows u0 - pa + {0, 1} feed taps ty = 2r + 1 - pa
          const int su = hy0 - pa + hyy, sv = hx0 - pb + hxx;
          if (R < HROWS && (unsigned)su < (unsigned)p.Ho && (unsigned)sv < (unsigned)p.Wo) v = (2 * su + pa) * p.Ws + 2 * sv + pb;
        } else {
          const int sy = hy0 - pad_y + hyy, sx = hx0 - pad_x + hxx;
          if (R < HROWS && (unsigned)sy < (unsigned)p.Hs && (unsigned)sx < (unsigned)p.Ws) v = sy * p.Ws + sx;
        }
        hpix[j] = v;
      }
#pragma unroll
      for (int j = 0; j < NJH; ++j) {
        ho1[j] = hpix[j] >= 0 ? hpix[j] * (p.C1 * ES) + gp * 16 : kDmaOOB;
        ho2[j] = hpix[j] >= 0 ? hpix[j] * (p.C2 * ES) + gp * 16 : kDmaOOB;
      }
    };
    int hwrow[NBW];  // byte offset of this lane's piece in weight row (n0 + RI*ii + ra), tap 0, channel 0
#pragma unroll
    for (int j = 0; j < NBW; ++j) hwrow[j] = ((n0 + RI * (wave + NW * j) + ra) * p.Kw + gp * EPP) * ES;
    struct Grp {  // K-group (wave-uniform)
      int c0;     // first channel of the chunk in the concatenated input
      int pa, pb;  // HALO_DG4: parity plane
      int live;   // inside the K range
    };
    const int Q = p.nk;
    auto group = [&](int q) __attribute__((always_inline)) {
      Grp g;
      g.live = q < Q ? 1 : 0;
      const int qq = g.live ? q : 0;
      const int pl = HALO == HALO_DG4 ? qq / p.cpt : 0;
      g.c0 = (qq - pl * p.cpt) * KC;
      g.pa = pl >> 1;
      g.pb = pl & 1;
      return g;
    };
    auto issue_halo = [&](int j, int buf, const Grp& g) __attribute__((always_inline)) {
      const int first = g.c0 < p.C1 ? 1 : 0;
      const int v = g.live ? (first ? ho1[j] : ho2[j]) : kDmaOOB;
      const int ii = wave + NW * j;  // (wave-uniform) instructions past the halo's last row write their zeros to the scratch KiB
      rb_dma16s(first ? rsrc1 : rsrc2, lds0 + (ii < NHI ? buf * HALOB + ii * 1024 : HSCRATCH), v, (first ? g.c0 : g.c0 - p.C1) * ES);
    };
    auto issue_w = [&](int j, int slot, const Grp& g, int tap) __attribute__((always_inline)) {  // tap: compile-time
      // weight tap of (group, tap): 3x3 -> tap; phase -> tap (the parity's own 2x2 block: rsrcw starts there);
      // DG4 -> (2r + 1 - pa) * 4 + 2s + 1 - pb
      const int tau = HALO == HALO_DG4 ? (2 * (tap >> 1) + 1 - g.pa) * 4 + 2 * (tap & 1) + 1 - g.pb : tap;
      rb_dma16s(rsrcw, lds0 + HBOFF + slot * BSLOT + (wave + NW * j) * 1024, g.live ? hwrow[j] : kDmaOOB, (tau * ctot + g.c0) * ES);
    };
    // fragment addressing: A = halo rows of the lane's pixel (patch row wm*RPT + tm, column l31) shifted by the tap,
    // swizzle key of THAT row; B as in the implicit-GEMM form
    int rb0[TM];
#pragma unroll
    for (int tm = 0; tm < TM; ++tm) rb0[tm] = (wm * RPT + tm) * HWD + l31;
    // (round 6) PRE: the HNT x TM fragment row addresses of a lane -- row R = its pixel's halo row shifted by the tap, 16-byte position
    // `hi` under that row's swizzle key -- are loop invariant: kept in registers where that is <= 18 of them (every phase / 4x4 form, the 3x3
    // form on 256-pixel patches), one v_xad_u32 per fragment read is left (k-step ^ , halo buffer +) where five VALU were
    constexpr bool PRE = HNT * TM <= 18;
    int ab[PRE ? HNT : 1][PRE ? TM : 1];
    if constexpr (PRE) {
      rb_for_each(
          [&](auto tc) __attribute__((always_inline)) {
            constexpr int t = decltype(tc)::value;
            constexpr int toff = (t / HK) * HWD + (t % HK);
#pragma unroll
            for (int tm = 0; tm < TM; ++tm) {
              const int R = rb0[tm] + toff;
              const int key = ROWB == 128 ? ((R >> 1) & 7) << 4 : ((R >> 2) & 3) << 4;
              ab[t][tm] = R * ROWB + ((hi << 4) ^ key);
              asm volatile("" : "+v"(ab[t][tm]));  // (a register from here on: not rematerialised inside the loop)
            }
          },
          std::make_integer_sequence<int, HNT>());
    }
    const int bfl = ROWB == 128 ? ((l31 >> 1) & 7) : ((l31 >> 2) & 3);
    int bfo[KS];
#pragma unroll
    for (int s = 0; s < KS; ++s) bfo[s] = HBOFF + (wn * WN + l31) * ROWB + (((2 * s + hi) ^ bfl) * 16);
    // fragments of k-step `ks` of the step with tap offset `toff` on halo buffer `hbuf`, weight slot `slot` (uniform)
    auto rd = [&](int hbuf, int slot, int toff, int ks, u32x4 (&a)[TM], u32x4 (&b)[TN], int t = 0) __attribute__((always_inline)) {
#pragma unroll
      for (int tm = 0; tm < TM; ++tm) {
        if constexpr (PRE) {  // (t: compile-time after unrolling) ((2 ks + hi) << 4) ^ key == ((hi << 4) ^ key) ^ (ks << 5)
          a[tm] = *reinterpret_cast<const u32x4*>(smem + ((ab[t][tm] ^ (ks << 5)) + hbuf * HALOB));
        } else {
          const int R = rb0[tm] + toff;
          const int key = ROWB == 128 ? ((R >> 1) & 7) << 4 : ((R >> 2) & 3) << 4;
          a[tm] = *reinterpret_cast<const u32x4*>(smem + hbuf * HALOB + R * ROWB + (((2 * ks + hi) << 4) ^ key));
        }
      }
#pragma unroll
      for (int tn = 0; tn < TN; ++tn) b[tn] = *reinterpret_cast<const u32x4*>(smem + slot * BSLOT + 32 * tn * ROWB + bfo[ks]);
    };

    Grp cur = group(0), nxt = group(1);
    halo_pix(cur.pa, cur.pb);
#pragma unroll
    for (int j = 0; j < NJH; ++j) issue_halo(j, 0, cur);
#pragma unroll
    for (int d = 0; d < 3; ++d)  // weights of steps 0, 1, 2 (HNT >= 4)
#pragma unroll
      for (int j = 0; j < NBW; ++j) issue_w(j, d, cur, d);
    if (HALO == HALO_DG4) halo_pix(nxt.pa, nxt.pb);  // (the pieces issued during group q fetch group q + 1)
    constexpr int NMMA = KS * TM * TN;
    u32x4 fa[2][TM], fb[2][TN];
    rb_dma_wait_n<NBW>();  // halo 0 and the weights of steps 0 and 1 have landed (step 2's may still fly)
    __syncthreads();
    __builtin_amdgcn_sched_barrier(0);
    rd(0, 0, 0, 0, fa[0], fb[0], 0);  // first fragments of step 0
    for (int q = 0; q < Q; ++q) {
      const int hbuf = q & 1, nbuf = (q + 1) & 1;
      // (opaque to the optimiser on purpose: the HNT x TM fragment row addresses are loop-invariant per lane, and hipcc would
      // hoist all of them out of the group loop -- up to 72 registers -- and spill; recomputed per tap they are a few VALU)
      if constexpr (!PRE) {
#pragma unroll
        for (int tm = 0; tm < TM; ++tm) asm volatile("" : "+v"(rb0[tm]));
      }
      rb_for_each(
          [&](auto tc) __attribute__((always_inline)) {
            constexpr int t = decltype(tc)::value;
            constexpr int tprev = (t + HNT - 1) % HNT;
            // (sched_barrier: the taps are straight-line code, and without it hipcc's machine scheduler moves fragment reads
            // and MFMAs across this wait + barrier)
            __builtin_amdgcn_sched_barrier(0);
            // in flight after this wait: what the previous step issued (NBW weight pieces + its halo pieces)
            if constexpr (KO != 1) {
              rb_dma_wait_n<NBW + halo_pieces_at_tap(tprev, HNT, NJH)>();
#ifdef RS_HALO_FENCED_BARRIER
              __syncthreads();
#else
              // a BARE s_barrier (round 6): __syncthreads() carries `s_waitcnt lgkmcnt(0)`, which here waits for the fragment reads of
              // THIS step that were issued just above it (the prefetch behind the previous step's last k-step) -- an LDS round trip in
              // front of every barrier, the very thing the prefetch was there to hide.  Nothing in this loop writes LDS with DS
              // instructions (the pipeline buffers are filled by LDS-DMA, covered by the counted vmcnt wait above), and a wave's reads
              // of the slots that are refilled behind this barrier were consumed by the MFMAs it issued before it.
              __builtin_amdgcn_s_barrier();
#endif
            }
            __builtin_amdgcn_sched_barrier(0);
            constexpr int NHT = halo_pieces_at_tap(t, HNT, NJH);  // halo pieces that go with this tap
            constexpr int NP = NBW + NHT;
            constexpr int PST = NMMA / (2 * NP) >= 1 ? NMMA / (2 * NP) : 1;  // one piece every PST MFMAs, front-loaded
            constexpr int t3 = (t + 3) % HNT;
            const int gstep = q * HNT + t;  // (wave-uniform)
            auto piece = [&](int i) __attribute__((always_inline)) {  // i: compile-time after unrolling
              if constexpr (KO == 2) return;
              if (i < NBW) {
                issue_w(i, (gstep + 3) & (DRING - 1), (t + 3 < HNT) ? cur : nxt, t3);
              } else {
                int seen = 0;
#pragma unroll
                for (int j = 0; j < NJH; ++j) {
                  if (halo_tap_of_piece(j, HNT, NJH) == t) {
                    if (seen == i - NBW) issue_halo(j, nbuf, nxt);
                    ++seen;
                  }
                }
              }
            };
            constexpr int toff = (t / HK) * HWD + (t % HK);
            constexpr int tn1 = (t + 1) % HNT;                        // the following step: its tap offset,
            constexpr int toff1 = (tn1 / HK) * HWD + (tn1 % HK);
            const int hbuf1 = (t + 1 < HNT) ? hbuf : nbuf;           // ... halo buffer and weight slot
            const int slot = gstep & (DRING - 1), slot1 = (gstep + 1) & (DRING - 1);
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
              // k-step ks + 1 of this step -- or, behind the last one, k-step 0 of the NEXT step (published by this step's barrier)
              if constexpr (KO != 3) {
                if (ks + 1 < KS) rd(hbuf, slot, toff, ks + 1, fa[(ks + 1) & 1], fb[(ks + 1) & 1], t);
                else rd(hbuf1, slot1, toff1, 0, fa[0], fb[0], tn1);
              }
#pragma unroll
              for (int tm = 0; tm < TM; ++tm)
#pragma unroll
                for (int tn = 0; tn < TN; ++tn) {
                  const int i = (ks * TM + tm) * TN + tn;  // MFMA index within the step (compile-time after unrolling)
                  if (i % PST == 0 && i / PST < NP) piece(i / PST);
                  if constexpr (KO == 4) {
                    asm volatile("" ::"v"(fb[ks & 1][tn]), "v"(fa[ks & 1][tm]));  // (the fragments stay live: the reads are not dead code)
                  } else if constexpr (KO == 3) {
                    mma16(acc[tn][tm], fb[0][tn], fa[0][tm], T());
                  } else {
                    mma16(acc[tn][tm], fb[ks & 1][tn], fa[ks & 1][tm], T());
                  }
                }
            }
#pragma unroll
            for (int i = (NMMA + PST - 1) / PST; i < NP; ++i) piece(i);  // (fewer MFMAs than pieces: never with these tiles)
          },
          std::make_integer_sequence<int, HNT>());
      cur = nxt;
      nxt = group(q + 2);
      if (HALO == HALO_DG4) halo_pix(nxt.pa, nxt.pb);
    }
    rb_dma_wait();  // the past-the-end pieces too: they write (zeros) into the buffers the epilogue stages through
    __builtin_amdgcn_sched_barrier(0);
  }
  __syncthreads();  // every wave is done with the pipeline buffers: the epilogue stages through them
  if constexpr (HL) __builtin_amdgcn_sched_barrier(0);  // (no staging write may be scheduled above that barrier)

  // ---- epilogue: registers -> LDS [pixel][cout] fp32 -> one 16-byte piece of couts per thread, row-wise stores.
  //      TM passes of WGM*32 rows each (pass t = sub-tile tm = t of every wave) keep the staging tile at
  //      WGM*32 x (BN+4) floats: the LDS footprint, hence the blocks per CU, is set by the pipeline buffers alone.
  float* lds = reinterpret_cast<float*>(smem);
  constexpr int TPR = BN / EPP;              // threads per row
  constexpr int RPI = NT / TPR;              // rows per iteration
  const int cc = tid % TPR, rr = tid / TPR;
  const int col = n0 + cc * EPP;
  const bool cvalid = col < p.Cout;  // false only in a ragged last N tile (Cout % EPP == 0: a piece never straddles)
  constexpr int NSC = EPI == EPI_EVAL ? EPP : 1, NST = EPI == EPI_EVAL ? 1 : EPP, NBN = EPI == EPI_BWD ? EPP : 1;
  float sc[NSC], sh[NSC];
  if constexpr (EPI == EPI_EVAL) {
#pragma unroll
    for (int e = 0; e < EPP; ++e) {
      sc[e] = (p.scale && cvalid) ? p.scale[col + e] : 1.f;
      sh[e] = (p.shift && cvalid) ? p.shift[col + e] : 0.f;
    }
  }
  T* outp = p.out;  // destination of this block's couts (block-uniform: a tile never straddles csplit)
  const T* maskp = p.mask;
  int ostride = p.Cout, ocol = col;
  if (EPI == EPI_EVAL && p.out2) {
    if (n0 >= p.csplit) {
      outp = p.out2;
      maskp = p.mask2;
      ostride = p.Cout - p.csplit;
      ocol = col - p.csplit;
    } else {
      ostride = p.csplit;
    }
  }
  float st0[NST], st1[NST];  // BatchNorm statistics of this thread's rows (EPI_STATS / EPI_BWD)
  float bmu[NBN], bis[NBN];
#pragma unroll
  for (int e = 0; e < NST; ++e) st0[e] = st1[e] = 0.f;
  if constexpr (EPI == EPI_BWD) {
#pragma unroll
    for (int e = 0; e < EPP; ++e) {
      bmu[e] = cvalid ? p.bn_mean[col + e] : 0.f;
      bis[e] = cvalid ? p.bn_invstd[col + e] : 0.f;
    }
  }
  // Row loop of a pass, in groups of G rows per thread: first ALL of the group's residual / ReLU-mask / BatchNorm-input
  // pieces are requested (raw 16-byte loads, unconditional: a row past M or a column past Cout reads offset 0 and is dropped
  // at the store), then the group is combined and stored.  Written as one loop with the loads inside, hipcc keeps each
  // row's loads behind the previous row's store (they may alias for all it knows): one HBM round trip per row.  The first
  // group's requests go out BEFORE the accumulators are staged, so their latency overlaps the LDS round trip + barrier.
  constexpr int NIT = WN / (2 * EPP);  // rows per thread per pass = WGM*32 / RPI
  constexpr int G = (ES == 4 && EPI == EPI_BWD) ? 1 : 2;  // (up to 3 tensors x G x 4 registers of loads in flight per thread)
  static_assert(NIT * RPI == WGM * 32 && NIT % G == 0, "row groups tile the pass");
  const bool has_res = EPI != EPI_STATS && p.res != nullptr, has_mask = EPI != EPI_STATS && maskp != nullptr;
  const bool has_bits = EPI == EPI_BWD && p.mask_bits != nullptr;
  constexpr bool has_bny = EPI == EPI_BWD;
  auto load_group = [&](int tm, int g0, long (&o)[G], bool (&ok)[G], u32x4 (&rr_)[G], u32x4 (&rm_)[G], u32x4 (&ry_)[G])
      __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < G; ++i) {
      const int lrow = rr + (g0 + i) * RPI;
      const int row = (lrow >> 5) * WM + 32 * tm + (lrow & 31);  // tile row of pass-local row lrow
      const int opix = orow[row];
      ok[i] = opix >= 0 && cvalid;
      o[i] = ok[i] ? (long)opix * ostride + ocol : 0;
    }
    if (has_res) {
#pragma unroll
      for (int i = 0; i < G; ++i) rr_[i] = *reinterpret_cast<const u32x4*>(p.res + o[i]);
    }
    if (has_mask) {
#pragma unroll
      for (int i = 0; i < G; ++i) rm_[i] = *reinterpret_cast<const u32x4*>(maskp + o[i]);
    }
    if constexpr (EPI == EPI_BWD) {
      if (has_bits) {
#pragma unroll
        for (int i = 0; i < G; ++i) rm_[i][0] = p.mask_bits[o[i] >> 3];
      }
    }
    if constexpr (has_bny) {
#pragma unroll
      for (int i = 0; i < G; ++i) ry_[i] = *reinterpret_cast<const u32x4*>(p.bn_y + o[i]);
    }
  };
  auto unpack = [&](const u32x4 raw, float (&v)[EPP]) __attribute__((always_inline)) {
    if constexpr (ES == 4) {
      const f32x4 t = __builtin_bit_cast(f32x4, raw);
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] = t[e];
    } else {
      const bf16x8 t = __builtin_bit_cast(bf16x8, raw);
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] = (float)t[e];
    }
  };
  auto finish_group = [&](int g0, const long (&o)[G], const bool (&ok)[G], const u32x4 (&rr_)[G], const u32x4 (&rm_)[G],
                          const u32x4 (&ry_)[G]) __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < G; ++i) {
      const int lrow = rr + (g0 + i) * RPI;
      float v[EPP];
#pragma unroll
      for (int h = 0; h < EPP / 4; ++h) {
        const f32x4 t = *reinterpret_cast<const f32x4*>(&lds[lrow * LDO + cc * EPP + 4 * h]);
#pragma unroll
        for (int e = 0; e < 4; ++e) v[4 * h + e] = EPI == EPI_EVAL ? t[e] * sc[(4 * h + e) % NSC] + sh[(4 * h + e) % NSC] : t[e];
      }
      if (has_res) {
        float r[EPP];
        unpack(rr_[i], r);
#pragma unroll
        for (int e = 0; e < EPP; ++e) v[e] += r[e];
      }
      if (EPI == EPI_EVAL && p.relu) {
#pragma unroll
        for (int e = 0; e < EPP; ++e) v[e] = fmaxf(v[e], 0.f);
      }
      if (has_mask) {
        float z[EPP];
        unpack(rm_[i], z);
#pragma unroll
        for (int e = 0; e < EPP; ++e) v[e] = z[e] > 0.f ? v[e] : 0.f;
      }
      if constexpr (EPI == EPI_BWD) {
        if (has_bits) {
          const unsigned int bits = EPP == 8 ? rm_[i][0] : (rm_[i][0] >> (o[i] & 4));  // (fp32 pieces are 4 elements: a nibble)
#pragma unroll
          for (int e = 0; e < EPP; ++e) v[e] = ((bits >> e) & 1u) ? v[e] : 0.f;
        }
      }
      if (ok[i]) Piece<T>::store(outp + o[i], v);
      if constexpr (EPI != EPI_EVAL) {
        float w[EPP];
        Piece<T>::round(v, w);  // statistics of the values as stored (bf16-rounded on the bf16 path)
        const float keep = ok[i] ? 1.f : 0.f;
        if constexpr (EPI == EPI_BWD) {
          float yv[EPP];
          unpack(ry_[i], yv);
#pragma unroll
          for (int e = 0; e < EPP; ++e) {
            st0[e % NST] += keep * w[e];
            st1[e % NST] += keep * (w[e] * ((yv[e] - bmu[e % NBN]) * bis[e % NBN]));
          }
        } else {
#pragma unroll
          for (int e = 0; e < EPP; ++e) {
            st0[e % NST] += keep * w[e];
            st1[e % NST] += keep * (w[e] * w[e]);
          }
        }
      }
    }
  };
#pragma unroll
  for (int tm = 0; tm < TM; ++tm) {
    long o[G];
    bool ok[G];
    u32x4 rr_[G], rm_[G], ry_[G];
    load_group(tm, 0, o, ok, rr_, rm_, ry_);  // in flight across the staging below
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
    finish_group(0, o, ok, rr_, rm_, ry_);
#pragma unroll
    for (int g0 = G; g0 < NIT; g0 += G) {
      load_group(tm, g0, o, ok, rr_, rm_, ry_);
      finish_group(g0, o, ok, rr_, rm_, ry_);
    }
  }
  if constexpr (EPI != EPI_EVAL) {  // block reduction over the RPI row lanes -> one partial row per M tile
    __syncthreads();
    float* r0 = lds;             // [RPI][BN]
    float* r1 = lds + RPI * BN;  // [RPI][BN]
#pragma unroll
    for (int e = 0; e < EPP; ++e) {
      r0[rr * BN + cc * EPP + e] = st0[e % NST];
      r1[rr * BN + cc * EPP + e] = st1[e % NST];
    }
    __syncthreads();
    if (tid < 2 * BN) {
      const float* r = tid < BN ? r0 : r1;
      const int c = tid < BN ? tid : tid - BN;
      float a = 0.f;
#pragma unroll 4
      for (int l = 0; l < RPI; ++l) a += r[l * BN + c];
      if (n0 + c < p.Cout) p.stats[((long)mt * 2 + (tid < BN ? 0 : 1)) * p.Cout + n0 + c] = a;
    }
  }
}


template <typename T, int ROWB, bool PHASE, int EPI>
void launch_rows(int tile, int grid, hipStream_t s, const ConvArgsT<T>& a) {
  if constexpr (sizeof(T) == 2 && EPI == EPI_EVAL) {
    if (tile == T256x256) {  // 8 waves, 1 block per CU
      conv_igemm_dma<T, 256, 256, 2, 4, ROWB, PHASE, EPI><<<grid, 512, 0, s>>>(a);
      return;
    }
  }
  switch (tile) {
    case T128x128: conv_igemm_dma<T, 128, 128, 2, 2, ROWB, PHASE, EPI><<<grid, 256, 0, s>>>(a); break;
    case T128x64: conv_igemm_dma<T, 128, 64, 2, 2, ROWB, PHASE, EPI><<<grid, 256, 0, s>>>(a); break;
    case T128x32: conv_igemm_dma<T, 128, 32, 4, 1, ROWB, PHASE, EPI><<<grid, 256, 0, s>>>(a); break;
    case T256x128: conv_igemm_dma<T, 256, 128, 4, 2, ROWB, PHASE, EPI><<<grid, 512, 0, s>>>(a); break;
    default: conv_igemm_dma<T, 64, 64, 2, 2, ROWB, PHASE, EPI><<<grid, 256, 0, s>>>(a); break;
  }
}

}  // namespace

// halo-once forms: 8 waves as 4 x 2 over a 256-pixel patch (8 rows x 32) x BN couts with 128-byte rows (64-channel
// chunks), or -- `bn` = 128 | 0x1000 -- over a 512-pixel patch (16 rows x 32) x 128 couts with 64-byte rows (32-channel
// chunks): wave tiles of 128 x 64, 2/3 of the weight DMA bytes and 3/4 of the fragment reads per MFMA
template <int HALO, bool PHASE, int EPI>
void launch_halo(int bn, int grid, hipStream_t s, const ConvArgsT<bf16_t>& a) {
  if (bn == (128 | 0x1000)) conv_igemm_dma<bf16_t, 512, 128, 4, 2, 64, PHASE, EPI, HALO><<<grid, 512, 0, s>>>(a);
  else if (bn == 128) conv_igemm_dma<bf16_t, 256, 128, 4, 2, 128, PHASE, EPI, HALO><<<grid, 512, 0, s>>>(a);
  else conv_igemm_dma<bf16_t, 256, 64, 4, 2, 128, PHASE, EPI, HALO><<<grid, 512, 0, s>>>(a);
}

// RS_CONV_INSTANTIATE(name, T, PHASE, EPI)
#define RS_CONV_DEFINE_LAUNCHER(name, T, PHASE, EPI)                                            \
  void name(int tile, int rowb, int grid, hipStream_t s, const ConvArgsT<T>& a) {               \
    if (rowb == 128) launch_rows<T, 128, PHASE, EPI>(tile, grid, s, a);                         \
    else launch_rows<T, 64, PHASE, EPI>(tile, grid, s, a);                                      \
  }
#endif  // RS_CONV_INSTANTIATE

// Shared device helpers for the gfx950 kernels.  wave = 64 lanes, 256 CUs in 8 XCDs.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/robosat_hip.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16_t;
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

// Four consecutive activations as fp32, whatever the storage type: fp32 = one 16-byte access, bf16 = one 8-byte
// access (+ v_cvt_pk_bf16_f32, round to nearest even, on the way out).  All arithmetic stays fp32.
__device__ __forceinline__ f32x4 rs_ld4(const float* p) { return *reinterpret_cast<const f32x4*>(p); }
__device__ __forceinline__ f32x4 rs_ld4(const bf16_t* p) {
  return __builtin_convertvector(*reinterpret_cast<const bf16x4*>(p), f32x4);
}
__device__ __forceinline__ void rs_st4(float* p, f32x4 v) { *reinterpret_cast<f32x4*>(p) = v; }
__device__ __forceinline__ void rs_st4(bf16_t* p, f32x4 v) {
  *reinterpret_cast<bf16x4*>(p) = __builtin_convertvector(v, bf16x4);
}
__device__ __forceinline__ float rs_ld1(const float* p) { return *p; }
__device__ __forceinline__ float rs_ld1(const bf16_t* p) { return (float)*p; }

// V = 4 or 8 consecutive activations as fp32 (V = 8: one 16-byte access in bf16, two in fp32).
template <int V>
struct rs_vecf {
  float v[V];
};
template <int V, typename T>
__device__ __forceinline__ rs_vecf<V> rs_ldv(const T* p) {
  rs_vecf<V> r;
  if constexpr (V == 4) {
    const f32x4 t = rs_ld4(p);
#pragma unroll
    for (int e = 0; e < 4; ++e) r.v[e] = t[e];
  } else if constexpr (sizeof(T) == 2) {
    const bf16x8 t = *reinterpret_cast<const bf16x8*>(p);
#pragma unroll
    for (int e = 0; e < 8; ++e) r.v[e] = (float)t[e];
  } else {
    const f32x4 a = *reinterpret_cast<const f32x4*>(p), b = *(reinterpret_cast<const f32x4*>(p) + 1);
#pragma unroll
    for (int e = 0; e < 4; ++e) r.v[e] = a[e], r.v[4 + e] = b[e];
  }
  return r;
}
template <int V, typename T>
__device__ __forceinline__ void rs_stv(T* p, const rs_vecf<V>& r) {
  if constexpr (V == 4) {
    rs_st4(p, f32x4{r.v[0], r.v[1], r.v[2], r.v[3]});
  } else if constexpr (sizeof(T) == 2) {
    bf16x8 t;
#pragma unroll
    for (int e = 0; e < 8; ++e) t[e] = (bf16_t)r.v[e];
    *reinterpret_cast<bf16x8*>(p) = t;
  } else {
    *reinterpret_cast<f32x4*>(p) = f32x4{r.v[0], r.v[1], r.v[2], r.v[3]};
    *(reinterpret_cast<f32x4*>(p) + 1) = f32x4{r.v[4], r.v[5], r.v[6], r.v[7]};
  }
}

// One pixel's 32 channels into fp32 registers: 16-byte loads issued together (4 in bf16, 8 in fp32).
__device__ __forceinline__ void rs_ld_row32(const bf16_t* p, float (&v)[32]) {
  bf16x8 t[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) t[j] = *reinterpret_cast<const bf16x8*>(p + j * 8);
#pragma unroll
  for (int j = 0; j < 4; ++j)
#pragma unroll
    for (int e = 0; e < 8; ++e) v[j * 8 + e] = (float)t[j][e];
}
__device__ __forceinline__ void rs_ld_row32(const float* p, float (&v)[32]) {
  f32x4 t[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) t[j] = *reinterpret_cast<const f32x4*>(p + j * 4);
#pragma unroll
  for (int j = 0; j < 8; ++j)
#pragma unroll
    for (int e = 0; e < 4; ++e) v[j * 4 + e] = t[j][e];
}
__device__ __forceinline__ void rs_st_row32(bf16_t* p, const float (&v)[32]) {
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    bf16x8 t;
#pragma unroll
    for (int e = 0; e < 8; ++e) t[e] = (bf16_t)v[j * 8 + e];
    *reinterpret_cast<bf16x8*>(p + j * 8) = t;
  }
}
__device__ __forceinline__ void rs_st_row32(float* p, const float (&v)[32]) {
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    f32x4 t;
#pragma unroll
    for (int e = 0; e < 4; ++e) t[e] = v[j * 4 + e];
    *reinterpret_cast<f32x4*>(p + j * 4) = t;
  }
}

#define RS_LAUNCH_RESULT() ((int)hipGetLastError())

// An LDS write that OTHER waves read behind the next barrier: the writing wave waits for its own DS operations first.  __syncthreads()
// is supposed to (its release fence carries `s_waitcnt lgkmcnt(0)`), but in the pipelined kernels whose LDS-DMA waits are inline asm hipcc
// DROPS that wait inside loops -- the barrier comes out as a bare s_barrier (round 6: a gather table written by wave 0 and read by the other
// waves three instructions behind the barrier was stale in 40-85 % of launches beside an LDS-using neighbour; profiles/r06/dma_order.txt).
// scripts/isa_audit.py (tests/test_isa_audit.py) checks every kernel of the library for exactly this: no s_barrier with a DS write pending.
__device__ __forceinline__ void rs_lds_writes_done() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }

// Blocks are dispatched round-robin over the 8 XCDs (block b -> XCD b % 8, observed; speed only, never
// correctness).  Remap so each XCD works on a contiguous run of tile indices and neighbouring tiles (which share
// input halos and weight panels) hit the same private L2.  Bijective for any grid size.
__device__ __forceinline__ int rs_xcd_remap(int b, int nwg) {
  const int xcd = b & 7, k = b >> 3;
  const int q = nwg >> 3, r = nwg & 7;
  const int base = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
  return base + k;
}

// v[lane] + v[lane ^ 16] / v[lane] + v[lane ^ 32] in every lane, on the VALU (gfx950's v_permlane16_swap / v_permlane32_swap: rows of 16 / 32 lanes
// change places between the two operands) instead of two ds_bpermute round trips through the LDS crossbar.  Same sums as
// `v += __shfl_xor(v, 16)` bit for bit (one commutative add per lane).  Inline asm: the builtin with the same value in both operands is
// folded by hipcc into `v + v` (ROCm 7.2); `s_nop 1` = the wait states the swap needs behind a VALU write of its operands.
__device__ __forceinline__ float rs_xor16_sum(float v) {
  float a = v, b = v;
  asm("s_nop 1\n\tv_permlane16_swap_b32 %0, %1" : "+v"(a), "+v"(b));
  return a + b;
}
__device__ __forceinline__ float rs_xor32_sum(float v) {
  float a = v, b = v;
  asm("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(a), "+v"(b));
  return a + b;
}

// sum over the 16 lanes of the lane's DPP row, in every lane of the row: v + ror 8, + ror 4, + ror 2, + ror 1 (VALU only: no LDS crossbar)
__device__ __forceinline__ float rs_row16_sum(float v) {
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x128, 0xf, 0xf, false));  // row_ror:8
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x124, 0xf, 0xf, false));  // row_ror:4
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x122, 0xf, 0xf, false));  // row_ror:2
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x121, 0xf, 0xf, false));  // row_ror:1
  return v;
}

__device__ __forceinline__ float rs_wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

__device__ __forceinline__ double rs_wave_sum(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

static inline int rs_cdiv(long a, long b) { return (int)((a + b - 1) / b); }

// Division of a 31-bit unsigned by a runtime-constant divisor with one mul-hi (Granlund-Montgomery round-up
// method): q = (umulhi(n, mul) + n) >> shift, exact for 0 <= n < 2^31 and 1 <= d < 2^31.
struct rs_fastdiv {
  unsigned int mul;
  unsigned int shift;
  unsigned int div;
};

static inline rs_fastdiv rs_make_fastdiv(unsigned int d) {
  rs_fastdiv f;
  unsigned int s = 0;
  while ((1ull << s) < d) ++s;  // s = ceil(log2 d)
  f.shift = s;
  f.mul = (unsigned int)((((1ull << s) - d) << 32) / d + 1);
  f.div = d;
  return f;
}

__host__ __device__ __forceinline__ unsigned int rs_div(unsigned int n, const rs_fastdiv f) {
#if defined(__HIP_DEVICE_COMPILE__)
  return (__umulhi(n, f.mul) + n) >> f.shift;
#else
  return (unsigned int)((((unsigned long long)n * f.mul) >> 32) + n) >> f.shift;
#endif
}

// knobs.hip (library-internal): the measurement / A-B switches of every dispatcher, seeded from the environment once and
// changed afterwards only through rs_set_knob() -- no dispatcher calls getenv().  Unset = the measured rules.
struct RsKnobs {
  int conv_tile = -1;          // RS_CONV_TILE: forced implicit-GEMM tile index (rs_conv2d_set_tuning), -1 = heuristics
  int conv_rowb = 0;           // RS_CONV_ROWB: forced K-chunk row bytes (64 | 128), 0 = heuristics
  int conv_big = 1;            // RS_CONV_BIG: the 8-wave 256x256 tile allowed (bf16, no fused statistics)
  int conv_min256 = 384;       // RS_CONV_MIN256: ... for launches with at least this many 256x256 blocks
  int conv_halo = 1;           // RS_CONV_HALO: halo-once forms allowed (bf16)
  int conv_halo_min = 192;     // RS_CONV_HALO_MIN: ... for launches with at least this many blocks
  int conv_halo512 = -1;       // RS_CONV_HALO512: the 512-pixel patch: -1 by rule, 0 never, 1 wherever it can run
  int conv1x1_ew = -1;         // RS_CONV1X1_EW: conv1x1_ew_f32 -- -1 by rule (K <= 64), 0 never, 1 wherever it can run
  int conv1x1_np = 0;          // RS_CONV1X1_NP: conv1x1_np_f32 (fp32 1x1 with the epilogue between the next sub-tile's MFMAs): -1 by rule, 0 never, 1 wherever it can run
  int conv1x1_ew_bf16 = 0;     // RS_CONV1X1_EW_BF16: conv1x1_ew_bf16 (train-mode 1x1 forward) -- 0 never, 1 wherever it can run
  int halo_ko = 0;             // RS_HALO_KO: knock-out variant of the halo kernel (`make KO=1` builds only)
  int wgrad_f32_phase = 1;     // RS_WGRAD_F32_PHASE: fp32 DecoderBlock weight gradient in phase form (0: direct form)
  int wgrad_f32_dma = -1;      // RS_WGRAD_F32_DMA: fp32 weight gradient by LDS-DMA (conv_wgrad_f32_dma.hip): -1 by rule, 0 never, 1 wherever it can run
  int wgrad_f32_blocks = 2048; // RS_WGRAD_F32_BLOCKS: block target of the fp32 weight-gradient launches
  int wgrad_f32_wino = 1;      // RS_WGRAD_F32_WINO: fp32 DecoderBlock weight gradient in the Winograd domain (conv_wgrad_wino_f32.hip); 0: phase form
  int wgrad_f32_wino_blocks = 1024;  // RS_WGRAD_F32_WINO_BLOCKS: block target of those launches
  int wgrad_f32_wino33 = 1;    // RS_WGRAD_F32_WINO33: fp32 weight gradient of the stride-1 3x3 convolutions in the Winograd domain (conv_wgrad_wino33_f32.hip); 0: nine taps
  int wgrad_f32_wino33_blocks = 256;  // RS_WGRAD_F32_WINO33_BLOCKS: block target of those launches (x 3 for the two-wave block)
  int wgrad_blocks = 96;       // RS_WGRAD_BLOCKS: block target of the tap-per-block bf16 weight-gradient launches (192 for two chunk buffers; 96 with the ring of three: profiles/r05/wgrad_ring.txt)
  int wgrad_blocks_phase = 1536;  // RS_WGRAD_BLOCKS_PHASE: ... of the phase-form launches
  int wgrad_phase4 = 1;        // RS_WGRAD_PHASE4: the phase form's 128 x 128 launches as one plane x four offsets per block (-0.2 ms on the bf16 step).  Round 5 withdrew it for a reproducibility failure whose cause round 6 found (a gather table published through a bare s_barrier: profiles/r06/dma_order.txt) and fixed
  int wgrad_blocks_phase4 = 256;  // RS_WGRAD_BLOCKS_PHASE4: block target of those launches (8-wave blocks, one per CU)
  int wgrad_ring = 3;          // RS_WGRAD_RING: chunk buffers of the tap-per-block bf16 weight-gradient kernel.  3 = two chunks in flight, counted waits (-0.1 ms on the step at 96 blocks); 2 = the drained two-buffer pipeline; 4 = the race screen's positive control (the round-5 defect kept on purpose, one tile); 5-7 = the bisect's variants (`make EXP=1`)
  int lovasz_xcd = 1;          // RS_LOVASZ_XCD: the Lovasz gradient scatter keeps an image's blocks on one XCD (0: natural order)
  int wino_wide = 1;           // ROBOSAT_WINO_WIDE: the 128 x 64 block of the fp32 Winograd DecoderBlock kernel
};
__attribute__((visibility("hidden"))) RsKnobs& rs_knobs();

// conv_wgrad_thin_bf16.hip (library-internal): all-taps-per-block weight gradient of the Cout = 32 3x3 layers.
// plan: 1 if `d` qualifies (+ grid size and number of fp32 partial slices [32][9*Cin] it writes); launch: the kernel.
__attribute__((visibility("hidden"))) int rs_wgrad_thin_plan(const rs_conv_desc* d, int* blocks, int* slices);
__attribute__((visibility("hidden"))) int rs_wgrad_thin_launch(const rs_conv_desc* d, const void* dy, const void* src,
                                                               float* partial, void* stream);

// conv_thin_bf16.hip (library-internal): the bf16 decoder-tail convolutions with all taps from one LDS halo tile.
// mode 0: 3x3 / pad 1, 32 -> 32;  1: phase form of DecoderBlock 128 -> 32 (weights from rs_pack_phase_weight_dt);
// 2: its 4x4 / stride-2 data gradient 32 -> 128 (weights from rs_pack_dgrad_phase_weight_dt).
__attribute__((visibility("hidden"))) int rs_conv_thin_bf16_launch(int mode, const void* src, const void* wgt, const void* mask,
                                                                   void* out, int N, int Hs, int Ws, int relu, void* stream);

// reduce.hip (library-internal): out[i] = sum over `splits` partial tiles of n floats (n % 4 == 0); `scratch` needs
// rs_reduce_scratch_floats(n, splits) floats (may be NULL when that is 0).
__attribute__((visibility("hidden"))) long rs_reduce_scratch_floats(long n, int splits);
__attribute__((visibility("hidden"))) int rs_reduce_splits(const float* ws, float* out, long n, int splits, float* scratch,
                                                           void* stream);

// conv_igemm_dma.hip (library-internal): the LDS-DMA implicit-GEMM kernel instantiated for fp32 (non-stem convolutions of
// rs_conv2d_fwd) and the tile it picks (index order: 128x128, 128x64, 128x32, 64x64).
__attribute__((visibility("hidden"))) int rs_conv_dma_f32(const rs_conv_desc* d, const float* src1, const float* src2,
                                                          const float* weight, const float* scale, const float* shift,
                                                          const float* residual, const float* relu_mask, float* out,
                                                          void* stream);
__attribute__((visibility("hidden"))) int rs_conv_dma_tile(const rs_conv_desc* d);

// The THIN convolutions of the decoder tail in bf16, all filter taps from one LDS halo tile, weights resident in registers.
//
//   dec5  ConvRelu 32 -> 32 3x3 at full resolution (reference robosat/unet.py:107,139)  and its data gradient   MODE 3x3
//   dec4  DecoderBlock 128 -> 32 behind the nearest-x2 upsample (unet.py:106,138) in phase form:
//         the four parity-specific 2x2 convolutions on the source grid (rs_pack_phase_weight)                   MODE PHASE
//   dec4  data gradient: the 4x4 / stride-2 convolution 32 -> 128 over dz (rs_pack_dgrad_phase_weight)          MODE DG4
//
// These layers hold 14 % of the forward FLOPs but move the two largest activations of the network (32 channels at 512^2).
// In bf16 they are HBM-bound by a wide margin (~150 flop/B against a machine balance of ~400), yet the generic implicit-GEMM
// kernel (conv_igemm_dma_kernel.h) re-gathers every input pixel once per filter tap through L2 -> LDS (9x / 4x / 16x the
// input bytes) and re-streams the weights for every 128-pixel tile: it runs at the ~12 TB/s the CUs' vector-memory path
// sustains, i.e. 1.5-2.6 TB/s of useful traffic.  Here:
//
//   weights   live in REGISTERS for the whole launch: a wave's share is taps x Cin/16 MFMA operand fragments (18 or 32
//             x 4 VGPRs) loaded once; blocks are persistent (one per CU) and walk patches of the image.
//   input     per patch ONE halo tile (patch + filter border, pixel-major rows as they lie in HBM) is copied to LDS by
//             LDS-DMA (buffer_load ... lds; out-of-image rows = out-of-range offsets = zeros) into a ring of three: the
//             halo two patches ahead is issued between the MFMAs of the current one, so two halos (~80-90 KB per CU) are
//             in flight -- with one block per CU that is what covers the HBM round trip (one halo ahead ran at 2 TB/s).  Every tap is then just an LDS row offset in
//             the fragment read -- nothing is fetched twice.  In phase form the four output parities share the halo:
//             wave w computes parity w from the same source pixels.
//   MFMA      v_mfma_f32_32x32x16_bf16, D[i = cout][j = pixel]: the weight fragment is the A operand.  Eight waves, two per
//             SIMD: a patch has 16 (row group | parity | cout group, sub-tile) work items of 32 pixels x 32 couts, two per
//             wave; both waves of a SIMD hold the same weights, and one's address / staging / store phases run under the
//             other's MFMAs (with four waves the epilogue VALU and the waits were ~55 % of the wave cycles).
//   output    accumulators -> bf16 patch image staged in the patch's own, consumed halo slot (16-byte pieces XOR-swizzled by
//             pixel) -> whole contiguous rows of the NHWC output with 16-byte stores, ReLU / ReLU-mask applied on the way
//             out; the mask patch arrives by LDS-DMA as well (a register load would sit behind the in-flight halos in
//             the compiler's vmcnt bookkeeping and drain them).
//
// Traffic per launch = input once (x 1.2-1.4 halo overlap) + output once (+ the mask once).
#include "common.h"

namespace {

enum { THIN_33 = 0, THIN_PHASE = 1, THIN_DG4 = 2 };

struct ThinConvArgs {
  const bf16_t* src;   // [N][Hs][Ws][CIN]
  const bf16_t* wgt;   // 3x3: [32][3][3][32] | PHASE: [4][32][2][2][128] | DG4: [128][4][4][32]
  const bf16_t* mask;  // optional, shaped like out: result zeroed where mask <= 0
  bf16_t* out;         // [N][Ho][Wo][COUT]
  int N, Hs, Ws, Ho, Wo;
  int relu;
  int ppx, ppi, total;  // patches per row of the base grid / per image / in all
};

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ __amdgpu_buffer_rsrc_t tc_make_rsrc(const void* base, long bytes) {
  const unsigned int n = bytes > 0x7FFF0000L ? 0x7FFF0000u : (unsigned int)(bytes < 0 ? 0 : bytes);
  const unsigned long b = (unsigned long)base;  // (descriptor inputs made provably wave-uniform: cdna_hip_programming.md T20)
  const unsigned int lo = __builtin_amdgcn_readfirstlane((unsigned int)b), hi = __builtin_amdgcn_readfirstlane((unsigned int)(b >> 32));
  return __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<void*>(((unsigned long)hi << 32) | lo), 0,
                                           (int)__builtin_amdgcn_readfirstlane(n), 0x00020000);
}

constexpr int kThinOOB = (int)0x80000000u;  // beyond every descriptor tc_make_rsrc builds: the DMA writes zeros

// One LDS-DMA wave instruction: lane l's 16 bytes at buffer offset `voff` land at LDS byte `lds_dst` + 16*l (see rb_dma16s in
// conv_igemm_dma_kernel.h for why this is inline asm and why m0 is a clobber).
__device__ __forceinline__ void tc_dma16(__amdgpu_buffer_rsrc_t r, unsigned int lds_dst, int voff) {
  asm volatile(
      "s_mov_b32 m0, %1\n\t"
      "s_nop 0\n\t"
      "buffer_load_dwordx4 %0, %2, 0 offen lds"
      :
      : "v"(voff), "s"(lds_dst), "s"(r)
      : "memory", "m0");
}
__device__ __forceinline__ void tc_dma_wait() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
__device__ __forceinline__ unsigned int tc_lds_addr(const void* p) {
  return (unsigned int)(unsigned long)(__attribute__((address_space(3))) const void*)p;
}

template <int MODE>
struct ThinGeom;
template <>
struct ThinGeom<THIN_33> {  // base grid = output grid; patch 16 x 32 pixels; sub-tile = one patch row
  static constexpr int CIN = 32, COUT = 32, NTAP = 9, PH = 16, PW = 32, HALO_H = 18, HALO_W = 34;
  static constexpr int OUT_ROWS = 16, OUT_PX = 32;  // output patch: rows x pixels per row
};
template <>
struct ThinGeom<THIN_PHASE> {  // base grid = source grid; patch 8 x 16 source pixels -> 16 x 32 output pixels
  static constexpr int CIN = 128, COUT = 32, NTAP = 4, PH = 8, PW = 16, HALO_H = 10, HALO_W = 18;
  static constexpr int OUT_ROWS = 16, OUT_PX = 32;
};
template <>
struct ThinGeom<THIN_DG4> {  // base grid = output grid (half the resolution of dz); patch 8 x 16 pixels
  static constexpr int CIN = 32, COUT = 128, NTAP = 16, PH = 8, PW = 16, HALO_H = 18, HALO_W = 34;
  static constexpr int OUT_ROWS = 8, OUT_PX = 16;
};

template <int N>
__device__ __forceinline__ void tc_dma_wait_n() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

template <int MODE>
__global__ __launch_bounds__(512, 2) void conv_thin_bf16(const ThinConvArgs p) {
  typedef ThinGeom<MODE> G;
  constexpr int CIN = G::CIN, COUT = G::COUT, NTAP = G::NTAP;
  constexpr int ROWB = CIN * 2;             // bytes per halo pixel
  constexpr int CPR = ROWB / 16;            // 16-byte pieces per halo row
  constexpr int RI = 64 / CPR;              // halo rows per DMA instruction (1 KiB)
  constexpr int KS = CIN / 16;              // MFMA k-steps per tap
  constexpr int NF = NTAP * KS;             // weight fragments per wave
  constexpr int HROWS = G::HALO_H * G::HALO_W;
  constexpr int NW = 8;                                // waves: two per SIMD, each with a full copy of its weight fragments and
  constexpr int ST = 2;                                // ST of the patch's sub-tiles -- one wave's VALU / LDS phases hide under its partner's MFMAs
  constexpr int NJ = ((HROWS + RI - 1) / RI + NW - 1) / NW;  // DMA instructions per wave per halo (every wave issues exactly NJ:
  constexpr int NINSTR = NW * NJ;                            // the counted vmcnt waits need that; surplus rows are out of range)
  constexpr int HALOB = NINSTR * 1024;
  constexpr int NB = 3;                        // halo ring: patch i computes, i+1 has landed or is landing, i+2 is issued
  constexpr int OROWB = G::OUT_PX * COUT * 2;  // bytes per output patch row
  constexpr int STAGEB = G::OUT_ROWS * OROWB;  // 32 KiB in every mode: staged in the patch's own (consumed) halo buffer
  constexpr int OPP = COUT / 8;                // 16-byte pieces per output pixel
  constexpr bool HAS_MASK = MODE != THIN_PHASE;  // (the forward DecoderBlock has no ReLU mask)
  constexpr int MASKB = HAS_MASK ? STAGEB : 0;   // the patch of the mask tensor, by DMA like the halo (no VGPR round trip)
  constexpr int NM = STAGEB / 1024 / NW;         // mask DMA instructions per wave
  static_assert(STAGEB == 32768 && STAGEB <= HALOB, "output patch image fits a halo buffer");

  __shared__ __attribute__((aligned(16))) unsigned char smem[NB * HALOB + MASKB + NINSTR * 64 * 2];
  unsigned char* maskbuf = smem + NB * HALOB;
  short* dtab = reinterpret_cast<short*>(smem + NB * HALOB + MASKB);  // per (DMA instruction, lane): hy << 10 | hx << 4 | piece, or -1

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int role = wave & 3, half = wave >> 2;  // role: 3x3 row group | PHASE output parity | DG4 cout group; half: which 2 of its 4 sub-tiles
  const int hi = lane >> 5, l31 = lane & 31;

  // ---- DMA lane constants (independent of the patch): halo row R = RI*ii + lane/CPR holds source pixel (R / HALO_W,
  //      R % HALO_W) of the halo; LDS position pp = lane % CPR receives channel piece pp ^ swz(R) (the swizzle lives on the
  //      SOURCE address: the DMA image is lane-linear).  swz: 64-byte rows (R >> 2) & 3, 256-byte rows R & 15.
  for (int e = tid; e < NINSTR * 64; e += 64 * NW) {
    const int ii = e >> 6, l = e & 63;
    const int R = RI * ii + l / CPR, pp = l % CPR;
    const int f = CPR == 4 ? ((R >> 2) & 3) : (R & 15);
    dtab[e] = R < HROWS ? (short)(((R / G::HALO_W) << 10) | ((R % G::HALO_W) << 4) | (pp ^ f)) : (short)-1;
  }

  // ---- this wave's weights -> registers: NF fragments of [32 rows][16 k]; lane holds row l31, k = 8*hi .. 8*hi + 7
  u32x4 wf[NF];
  {
    const bf16_t* wbase;
    if (MODE == THIN_33) wbase = p.wgt + (long)l31 * NTAP * CIN;                       // [cout][tap][cin]
    else if (MODE == THIN_PHASE) wbase = p.wgt + ((long)role * 32 + l31) * NTAP * CIN;  // [phase = role][cout][tap][cin]
    else wbase = p.wgt + ((long)role * 32 + l31) * NTAP * CIN;                          // [cout = 32*role + ..][tap][cin]
#pragma unroll
    for (int f = 0; f < NF; ++f) wf[f] = *reinterpret_cast<const u32x4*>(wbase + (f / KS) * CIN + (f % KS) * 16 + 8 * hi);
  }

  const __amdgpu_buffer_rsrc_t rsrc = tc_make_rsrc(p.src, (long)p.N * p.Hs * p.Ws * ROWB);
  const __amdgpu_buffer_rsrc_t rsrc_mask = tc_make_rsrc(p.mask ? (const void*)p.mask : (const void*)p.src,
                                                        p.mask ? (long)p.N * p.Ho * p.Wo * COUT * 2 : 0);
  const unsigned int lds0 = __builtin_amdgcn_readfirstlane(tc_lds_addr(smem));

  // patch -> (image, base-grid origin); halo origin in source coordinates, output origin
  struct Where {
    int n, by0, bx0;
    bool live;
  };
  auto locate = [&](int patch) __attribute__((always_inline)) {
    Where w;
    w.live = patch < p.total;
    const int pc = w.live ? patch : 0;
    w.n = pc / p.ppi;
    const int rem = pc - w.n * p.ppi;
    const int py = rem / p.ppx;
    w.by0 = py * G::PH;
    w.bx0 = (rem - py * p.ppx) * G::PW;
    return w;
  };
  // Halo piece j of this wave for patch `w` into ring slot `slot`.  A patch past the end still issues its pieces -- out of
  // range, zeros into a slot nobody reads: no branch between the MFMAs (a diamond there makes hipcc keep two copies of the
  // accumulators) and the same DMA count in every wave and iteration (the counted waits rely on it).
  auto issue_piece = [&](int j, int slot, const Where& w) __attribute__((always_inline)) {
    const int ii = wave + NW * j;  // wave-uniform
    const int c = dtab[ii * 64 + lane];
    const int hy0 = MODE == THIN_DG4 ? 2 * w.by0 - 1 : w.by0 - 1, hx0 = MODE == THIN_DG4 ? 2 * w.bx0 - 1 : w.bx0 - 1;
    const int sy = hy0 + (c >> 10), sx = hx0 + ((c >> 4) & 63);
    const bool ok = w.live && c >= 0 && (unsigned)sy < (unsigned)p.Hs && (unsigned)sx < (unsigned)p.Ws;
    const int voff = ok ? (((w.n * p.Hs + sy) * p.Ws + sx) * ROWB + (c & 15) * 16) : kThinOOB;
    tc_dma16(rsrc, lds0 + slot * HALOB + ii * 1024, voff);
  };
  // Mask piece j of this wave for the patch being computed: the output-shaped patch of `mask`, linear rows (OROWB bytes each)
  auto issue_mask = [&](int j, const Where& w) __attribute__((always_inline)) {
    const int ii = wave + NW * j;
    const int byte = ii * 1024 + lane * 16;
    const int row = byte / OROWB, inrow = byte - row * OROWB;
    const int oy = (MODE == THIN_PHASE ? 2 : 1) * w.by0 + row, ox = (MODE == THIN_PHASE ? 2 : 1) * w.bx0 + inrow / (COUT * 2);
    const bool ok = w.live && p.mask && oy < p.Ho && ox < p.Wo;
    const int voff = ok ? (((w.n * p.Ho + oy) * p.Wo + ox) * COUT * 2 + inrow % (COUT * 2)) : kThinOOB;
    tc_dma16(rsrc_mask, lds0 + NB * HALOB + ii * 1024, voff);
  };

  // ---- fragment addressing: sub-tile u = 2*half + t of this wave's role, lane pixel on the patch; halo row of tap (r, s) = baseR + r*HALO_W + s
  int baseR[ST];
#pragma unroll
  for (int t = 0; t < ST; ++t) {
    const int u = ST * half + t;
    if (MODE == THIN_33) baseR[t] = (4 * role + u) * G::HALO_W + l31;
    else if (MODE == THIN_PHASE) baseR[t] = (2 * u + (l31 >> 4) + (role >> 1)) * G::HALO_W + (l31 & 15) + (role & 1);
    else baseR[t] = (2 * (2 * u + (l31 >> 4))) * G::HALO_W + 2 * (l31 & 15);
  }
  // ---- staging addressing: where this lane's 4 consecutive couts (registers 4g .. 4g+3) of sub-tile t go.  The staged
  //      image is the output patch, rows of OROWB bytes; in PHASE mode a row holds its even columns first, then the odd ones
  //      (a wave writes one parity: consecutive lanes then sit 64 bytes apart instead of 128, which halves the bank conflicts
  //      of the 8-byte writes); the write-out undoes it.
  int sbase[ST];  // byte offset of the pixel in the staged patch image
  int sswz[ST];   // its swizzle key
#pragma unroll
  for (int t = 0; t < ST; ++t) {
    const int u = ST * half + t;
    int row, col;
    if (MODE == THIN_33) {
      row = 4 * role + u;
      col = l31;
    } else if (MODE == THIN_PHASE) {
      row = 2 * (2 * u + (l31 >> 4)) + (role >> 1);
      col = 16 * (role & 1) + (l31 & 15);  // staged position of output column 2*(l31 & 15) + (role & 1)
    } else {
      row = 2 * u + (l31 >> 4);
      col = l31 & 15;
    }
    sbase[t] = row * OROWB + col * COUT * 2;
    sswz[t] = OPP == 4 ? ((col >> 1) & 3) : (col & 15);
  }

  __syncthreads();  // dtab ready

  // ---- ring prologue: halos of this block's first two patches
  int patch = blockIdx.x;
  Where cur = locate(patch), nxt = locate(patch + gridDim.x);
#pragma unroll
  for (int j = 0; j < NJ; ++j) issue_piece(j, 0, cur);
#pragma unroll
  for (int j = 0; j < NJ; ++j) issue_piece(j, 1, nxt);
  tc_dma_wait_n<NJ>();  // halo 0 (this wave's pieces) has landed; halo 1 may still fly
  int slot = 0;
  while (cur.live) {
    // Every wave waited for ITS pieces of this halo before it got here (above; below, ahead of the write-out): barrier A makes
    // that everybody's -- and says the previous patch's staged output (and mask) have been read out, so the slot it was
    // staged in (the one halo i+2 goes to) and the mask buffer are free.
    __syncthreads();
    const Where nn = locate(patch + 2 * gridDim.x);
    const int slot2 = slot >= 1 ? slot - 1 : NB - 1;  // (slot + 2) % NB
    if (HAS_MASK && p.mask) {  // (uniform)
#pragma unroll
      for (int j = 0; j < NM; ++j) issue_mask(j, cur);  // needed one compute phase from now; older than this iteration's halo pieces
    }

    f32x16 acc[ST];
#pragma unroll
    for (int t = 0; t < ST; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

    unsigned char* H = smem + slot * HALOB;
    // (opaque to the optimiser on purpose: the NTAP x KS x ST fragment addresses below are loop-invariant per lane, and hipcc
    // would hoist all of them -- up to 64 registers next to the 128 weight registers -- out of the patch loop and spill;
    // recomputed per patch they are two VALU instructions per read)
#pragma unroll
    for (int t = 0; t < ST; ++t) asm volatile("" : "+v"(baseR[t]));
    constexpr int NG = NTAP * KS;  // fragment groups: one (tap, k-step) = ST pixel fragments against one weight fragment
    constexpr int NMMA = NG * ST;
    constexpr int PSTEP = NMMA / (2 * NJ) >= 1 ? NMMA / (2 * NJ) : 1;  // halo i+2's DMA pieces between the MFMAs, front-loaded
    constexpr int TW = MODE == THIN_33 ? 3 : (MODE == THIN_PHASE ? 2 : 4);
    // fragments of group g+1 are requested BEFORE the MFMAs of group g (the DMA statements between the MFMAs are asm with a
    // memory clobber: hipcc does not hoist LDS reads across them, so the prefetch is spelled out)
    auto read_group = [&](int g, u32x4 (&a)[ST]) __attribute__((always_inline)) {
      const int tap = g / KS, ks = g % KS;
      const int toff = (tap / TW) * G::HALO_W + (tap % TW);
#pragma unroll
      for (int t = 0; t < ST; ++t) {
        const int R = baseR[t] + toff;
        const int f = CPR == 4 ? ((R >> 2) & 3) : (R & 15);
        a[t] = *reinterpret_cast<const u32x4*>(H + R * ROWB + (((2 * ks + hi) ^ f) * 16));
      }
    };
    u32x4 a[2][ST];
    read_group(0, a[0]);
#pragma unroll
    for (int g = 0; g < NG; ++g) {
      if (g + 1 < NG) read_group(g + 1, a[(g + 1) & 1]);
#pragma unroll
      for (int t = 0; t < ST; ++t) {
        const int q = g * ST + t;  // MFMA index within the patch (compile-time after unrolling)
        if (q % PSTEP == 0 && q / PSTEP < NJ) issue_piece(q / PSTEP, slot2, nn);
        acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, wf[g]), __builtin_bit_cast(bf16x8, a[g & 1][t]), acc[t], 0, 0, 0);
      }
    }
#pragma unroll
    for (int j = (NMMA + PSTEP - 1) / PSTEP; j < NJ; ++j) issue_piece(j, slot2, nn);
    if (p.relu) {  // (uniform) ReLU on the fp32 accumulators: 16 v_max per sub-tile instead of unpack / compare / select per stored bf16
#pragma unroll
      for (int t = 0; t < ST; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = fmaxf(acc[t][r], 0.f);
    }

    __syncthreads();  // barrier B: every wave is done reading this halo -- its slot becomes the output staging area
    // ---- stage: D[i = cout][j = pixel], lane holds couts (r&3) + 8*(r>>2) + 4*hi of pixel l31 -> 8-byte groups of 4 couts
#pragma unroll
    for (int t = 0; t < ST; ++t)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        bf16x4 v;
        v[0] = (bf16_t)acc[t][4 * g + 0];
        v[1] = (bf16_t)acc[t][4 * g + 1];
        v[2] = (bf16_t)acc[t][4 * g + 2];
        v[3] = (bf16_t)acc[t][4 * g + 3];
        const int piece = (MODE == THIN_DG4 ? 4 * role : 0) + g;  // logical 16-byte piece of the pixel (8 couts)
        *reinterpret_cast<bf16x4*>(H + sbase[t] + ((piece ^ sswz[t]) * 16) + hi * 8) = v;
      }
    // All but this iteration's NJ halo pieces (halo i+2) have landed: halo i+1 (issued an iteration ago), this patch's
    // mask, and the previous patch's output stores.  The wait sits BEFORE the output stores on purpose: after them it
    // would wait for the stores as well (vmcnt counts them) -- one HBM write round trip per patch on the critical path.
    tc_dma_wait_n<NJ>();
    __syncthreads();  // barrier C: staging (and everybody's mask pieces) visible

    // ---- write out: the staged patch is OUT_ROWS rows of OROWB contiguous output bytes
    const int oy0 = (MODE == THIN_PHASE ? 2 : 1) * cur.by0, ox0 = (MODE == THIN_PHASE ? 2 : 1) * cur.bx0;
    constexpr int PPR = OROWB / 16;  // pieces per patch row
#pragma unroll 1
    for (int k = 0; k < STAGEB / 16 / (64 * NW); ++k) {
      const int e = tid + 64 * NW * k;
      const int row = e / PPR, within = e - row * PPR;
      const int px = within / OPP, pp = within - px * OPP;          // output column of the patch row, position inside the pixel
      const int spx = MODE == THIN_PHASE ? 16 * (px & 1) + (px >> 1) : px;  // where that column was staged
      const int piece = pp ^ (OPP == 4 ? ((spx >> 1) & 3) : (spx & 15));    // logical piece stored at position pp of it
      const int oy = oy0 + row, ox = ox0 + px;
      if (oy < p.Ho && ox < p.Wo) {
        const long o = (((long)cur.n * p.Ho + oy) * p.Wo + ox) * COUT + piece * 8;
        u32x4 v = *reinterpret_cast<const u32x4*>(H + row * OROWB + (spx * OPP + pp) * 16);
        if (HAS_MASK && p.mask) {  // zero where the mask value is <= 0 (or NaN): bf16 sign bit set or magnitude zero, on the raw bits
          const u32x4 z = *reinterpret_cast<const u32x4*>(maskbuf + row * OROWB + (px * OPP + piece) * 16);
#pragma unroll
          for (int c = 0; c < 4; ++c) {
            const unsigned int lo = z[c] & 0xFFFFu, hi16 = z[c] >> 16;
            const unsigned int keep = ((lo - 1u) < 0x7F80u ? 0x0000FFFFu : 0u) | ((hi16 - 1u) < 0x7F80u ? 0xFFFF0000u : 0u);
            v[c] &= keep;
          }
        }
        *reinterpret_cast<u32x4*>(p.out + o) = v;
      }
    }
    patch += gridDim.x;
    cur = nxt;
    nxt = nn;
    slot = slot + 1 == NB ? 0 : slot + 1;
  }
}

}  // namespace

// mode: 0 = 3x3 32 -> 32 (pad 1), 1 = phase form 128 -> 32 (output 2Hs x 2Ws), 2 = 4x4/s2 data gradient 32 -> 128 (output
// Hs/2 x Ws/2).  Returns RS_EINVAL when the launch does not fit (the caller then takes the generic kernel).
int rs_conv_thin_bf16_launch(int mode, const void* src, const void* wgt, const void* mask, void* out, int N, int Hs, int Ws, int relu,
                             void* stream) {
  if (!src || !wgt || !out || N <= 0 || Hs <= 0 || Ws <= 0) return RS_EINVAL;
  ThinConvArgs a;
  a.src = reinterpret_cast<const bf16_t*>(src);
  a.wgt = reinterpret_cast<const bf16_t*>(wgt);
  a.mask = reinterpret_cast<const bf16_t*>(mask);
  a.out = reinterpret_cast<bf16_t*>(out);
  a.N = N;
  a.Hs = Hs;
  a.Ws = Ws;
  a.relu = relu;
  int bh, bw, ph, pw;  // base grid, patch
  long in_bytes;
  if (mode == THIN_33) {
    a.Ho = Hs, a.Wo = Ws, bh = Hs, bw = Ws, ph = 16, pw = 32, in_bytes = (long)N * Hs * Ws * 64;
  } else if (mode == THIN_PHASE) {
    a.Ho = 2 * Hs, a.Wo = 2 * Ws, bh = Hs, bw = Ws, ph = 8, pw = 16, in_bytes = (long)N * Hs * Ws * 256;
  } else if (mode == THIN_DG4) {
    if ((Hs & 1) || (Ws & 1)) return RS_EINVAL;
    a.Ho = Hs / 2, a.Wo = Ws / 2, bh = Hs / 2, bw = Ws / 2, ph = 8, pw = 16, in_bytes = (long)N * Hs * Ws * 64;
  } else {
    return RS_EINVAL;
  }
  if (in_bytes >= 0x7FFF0000L) return RS_EINVAL;  // 32-bit DMA offsets over the whole input tensor
  a.ppx = rs_cdiv(bw, pw);
  a.ppi = a.ppx * rs_cdiv(bh, ph);
  const long total = (long)a.ppi * N;
  if (total >= (1L << 31)) return RS_EINVAL;
  a.total = (int)total;
  const int grid = a.total < 256 ? a.total : 256;  // persistent: one 8-wave block per CU
  hipStream_t s = (hipStream_t)stream;
  if (mode == THIN_33) conv_thin_bf16<THIN_33><<<grid, 512, 0, s>>>(a);
  else if (mode == THIN_PHASE) conv_thin_bf16<THIN_PHASE><<<grid, 512, 0, s>>>(a);
  else conv_thin_bf16<THIN_DG4><<<grid, 512, 0, s>>>(a);
  return RS_LAUNCH_RESULT();
}

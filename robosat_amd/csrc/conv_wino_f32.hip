// conv_wino_f32.hip -- DecoderBlock (reference robosat/unet.py:63-73: conv3x3(pad 1) over a nearest-x2 upsample of
// cat[skip, prev]) in fp32 as a WINOGRAD F(2x2, 2x2) convolution on the phase form.
//
// The phase form (conv_igemm_dma_kernel.h, PHASE = true) already turns the 3x3 convolution over the upsampled tensor into
// four 2x2 convolutions on the SOURCE grid, one per output parity (py, px), with pre-summed taps: 16 multiply-adds per source
// pixel, channel and cout instead of 36.  A 2x2 correlation producing a 2x2 block of outputs from a 3x3 block of inputs
// needs only 9 multiplies with the minimal-filtering transforms (all coefficients are 0 / +-1, so nothing is lost in fp32):
//     Y = A^T [ (G g G^T) (.) (B^T d B) ] A,   B^T = [1 -1 0; 0 1 0; 0 -1 1],  G = [1 0; 1 1; 0 1],  A^T = [1 1 0; 0 1 1]
// i.e. 9 GEMMs [tiles x Cin] . [Cin x Cout] per parity instead of 16: 9/16 of the phase form's MFMA work, 1/4 of the
// reference-shape count.  That trade only pays where the matrix cores are the bottleneck and the vector ALU / LDS have
// slack: exactly the fp32 path (v_mfma_f32_32x32x2_f32: 157 TFLOP/s, 64 cycles per instruction per SIMD), where the four
// generic phase layers are 48 % of a predict pass at 0.87-0.91 of the MFMA peak.  In bf16 the same layers are bound by the
// LDS-DMA stream, not by MFMA, and the transform would only add bytes: bf16 keeps the phase form.
//
// Mapping.  GEMM rows = TILES (2x2 blocks of source positions -> 4x4 output pixels over the four parities; each parity is
// its own set of work items), columns = couts, K = input channels of cat[skip, prev] in 16-channel chunks (64-byte rows).
//   block  = 8 waves, persistent: one block per CU walks work items (16*TG tiles x 32*CG couts x parity; TG x CG = 4 x 2, or
//            8 x 1 for the 32-cout layer).  A wave owns 16 tiles x 32 couts and keeps 9 x 2 accumulators of
//            v_mfma_f32_16x16x4_f32 (one pair per transformed position xi): 72 registers, two waves per SIMD -- while one
//            wave reads its patch and runs the transform's vector ops, the other's MFMAs keep the matrix core busy.
//   LDS    = per K-chunk the block's source HALO -- every source pixel any of its tiles touches, once: (2 PB + 1)^2 pixels
//            per PB x PB patch of tiles, 64 bytes each -- plus the transformed filters U[xi][cout][16 ch], both by LDS-DMA
//            (buffer_load ... lds; out-of-image pixels arrive as zeros), double buffered, one barrier per chunk.  A source
//            pixel is fetched ONCE per block and chunk (the generic phase kernel fetches it once per tap: 4x).  The fetch side
//            runs one chunk ahead ACROSS work items: the next item's first chunk streams in while this item's outputs are
//            stored (its halo table is built an item ahead, double buffered).
//   reads  = each lane reads the 3x3 patch of its tile (9 ds_read_b128: 4 channels of each of the 9 pixels), forms B^T d B
//            with 12 vector subtractions, reads the filter pieces one position ahead of their use (sched_barrier keeps hipcc
//            from sinking them in front of every MFMA group) and issues 72 MFMAs per chunk.  Halo rows are stored
//            even-x-first, sub-blocks padded to 8 rows, pieces XOR-swizzled with (row ^ row >> 1) & 3, lanes mapped to tiles
//            by a bit permutation: found by exhaustive search against ds_read_b128's four 16-lane groups, all 27 reads per
//            chunk are bank-conflict free (scripts/probes/wino_lds.py re-derives and checks it; SQ_LDS_BANK_CONFLICT = 0).
//   store  = A^T M A on the accumulators (3 adds per output element), ReLU, 16-byte stores straight from registers: a lane
//            holds 4 consecutive couts of its tile's pixels; no LDS staging, nothing to wait for.
// Weights: rs_pack_wino_phase_weight turns the phase pack [4][Cout][2][2][Cin] into U = G g G^T, [4][9][Cout][Cin].
// Which layers: decided on the layer's GEOMETRY alone (>= 8 tiles per image side), never on the batch size -- the two forms
// differ in fp32 summation order, and a tile's probabilities must not depend on the batch it travels in.
//
// Measured (MI355X, bs 16 at 512^2, scripts/bench_wino.py; generic phase kernel -> this kernel): dec0 0.68 -> 0.40 ms, dec1
// 1.24 -> 0.79, dec2 0.76 -> 0.48, dec3 2.39 -> 1.68, dec4 1.21 -> 0.79: 1.43-1.7x for 9/16 of the MFMA work, i.e. the matrix
// cores are 72-78 % busy here against 90 % in the generic kernel (SQ_VALU_MFMA_BUSY_CYCLES, profiles/r03).  What the rest is:
// the CU's LDS-DMA path (~23 B/clk) carries 12 B/clk here -- 55 KB per chunk, two thirds of it filters -- and the waves that
// issue those pieces stall on it in phase with each other (one barrier per chunk).  Variants measured on the way
// (profiles/r03/wino_variants.txt): 32x32x2 MFMAs with 144 accumulator registers at one wave per SIMD (104 TFLOP/s executed
// on dec3), the same with two dedicated DMA-producer waves and persistent blocks (101), with wave pairs splitting each
// chunk's channels and an LDS reduction at the end (110), this one (116); front-loading the DMA pieces: no change.
#define RS_CONV_INSTANTIATE  // (for the LDS-DMA helpers of the header; no kernel of it is instantiated here)
#include "conv_igemm_dma_kernel.h"

namespace {

struct WinoArgs {
  const float* src1;
  const float* src2;
  const float* u;  // [4][9][Cout][C1 + C2]
  float* out;      // [N][2 Hs][2 Ws][Cout]
  int N, Hs, Ws, C1, C2, Cout;
  int TY, TX;    // tiles per image: ceil(Hs / 2), ceil(Ws / 2)
  int BBY, BBX;  // PB x PB tile patches ("sub-blocks") per image
  int nsub;      // N * BBY * BBX
  int ncb;       // cout blocks: Cout / BN
  int relu;
  // DG instantiations only -- the DecoderBlock's DATA gradient (autograd of unet.py:63-73 under tools/train.py:186): `src1` is dz
  // [N][2 Hs][2 Ws][C1] (C2 = 0), `u` the transformed data-gradient filters [4][9][Cout][C1] (rs_pack_wino_dgrad_weight), the output
  // d cat[skip, prev] at SOURCE resolution [N][Hs][Ws][Cout], optionally split at cout `csplit` into out / out2 (torch.cat's backward
  // fused into the store) with a ReLU mask each (the tensor whose sign decides: the forward activation; null = none)
  float* out2;
  const float* mask1;
  const float* mask2;
  int csplit;
};

template <int PB>
struct WinoGeom {
  static constexpr int HW = 2 * PB + 1;  // halo width of a PB x PB patch of tiles ("sub-block")
  static constexpr int PITCH = HW;       // LDS rows per halo line
  static constexpr int HALF = PB + 1;    // even x first, then odd x
  // rows per sub-block, padded to a multiple of 8: the swizzle reads row bits 0..2, so every sub-block sees the pattern the
  // conflict-free search was run on
  static constexpr int SBROWS = (HW * PITCH + 7) / 8 * 8;
};

// lane (0..15) -> tile of the wave's 16: the bit permutation that belongs to the conflict-free LDS layout (found by exhaustive
// search over pitch / row order / swizzle / permutation against ds_read_b128's four 16-lane groups; scripts/probes/wino_lds.py)
template <int PB>
__device__ __forceinline__ int wino_lane_tile(int l) {
  if constexpr (PB == 8) return ((l & 1) << 1) | (((l >> 1) & 1) << 2) | ((l >> 2) & 1) | (l & 8);
  return ((l & 1) << 1) | (((l >> 1) & 1) << 3) | ((l >> 2) & 1) | (((l >> 3) & 1) << 2);
}
typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f32x2 pk_sub(f32x2 a, f32x2 b) {  // a - b on two floats: v_pk_add_f32 with the second operand negated
  f32x2 d;
  asm("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(d) : "v"(a), "v"(b));
  return d;
}
__device__ __forceinline__ int wino_swz(int row) { return (row ^ (row >> 1)) & 3; }  // 16-byte piece c of a row is stored at c ^ swz(row)

// TG tile groups (16 tiles each) x CG cout groups (32 couts each) = 8 waves per block
// NM = MFMA tiles of 16 couts per wave (a wave owns 16 tiles x 16*NM couts)
// DG (round 6): the 4x4 / stride-2 data gradient of the phase form in the same machinery.  d src(a, b) = sum over the four parity
// planes of dz, plane(pa, pb)(u, v) = dz(2 u + pa, 2 v + pb), of a 2x2 correlation with taps w4[2 r + 1 - pa][2 s + 1 - pb] at origin
// (a - pa, b - pb): exactly the forward's parity (py, px) = (1 - pa, 1 - pb) item on that plane -- same halo origin, same 3x3 patches,
// same transforms -- except that the four parities ACCUMULATE into one output tile instead of interleaving into four.  So a block
// owns an output item for four consecutive "units" (one per plane: its own halo table, source offsets and filter set, as for a
// forward item), keeps the accumulators across them and writes A^T M A once: 9/16 of the generic 4x4 kernel's multiply-adds.
template <int PB, int TG, int CG, int NM = 2, bool DG = false>
__global__ __launch_bounds__(512, 1) void conv_wino_f32_kernel(const WinoArgs p) {
  using G = WinoGeom<PB>;
  constexpr int NW = TG * CG;
  static_assert(NW == 8, "8 waves");
  constexpr int BMT = 16 * TG, BN = 16 * NM * CG;
  constexpr int SB = BMT / (PB * PB);
  static_assert(SB * PB * PB == BMT, "whole sub-blocks per block");
  constexpr int AROWS = SB * G::SBROWS;
  constexpr int IA = (AROWS + 15) / 16, IB = 9 * BN / 16;
  constexpr int AROWS_PAD = IA * 16;
  constexpr int NI = (IA + IB + NW - 1) / NW;  // DMA instructions per wave and chunk
  constexpr int STAGE = (AROWS_PAD + 9 * BN) * 64;
  constexpr int KC = 16;  // channels per chunk (64-byte rows)

  __shared__ __attribute__((aligned(16))) unsigned char smem[2 * STAGE + 2 * AROWS_PAD * 4];
  int* tabs = reinterpret_cast<int*>(smem + 2 * STAGE);  // [2][AROWS_PAD]: halo row -> source pixel relative to the item's first image, -1 = zeros

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int tg = wave % TG, cg = wave / TG;
  const int per_img = p.BBY * p.BBX;
  const int Cin = p.C1 + p.C2;
  const int nk = Cin / KC;
  const int ntiles = ((p.nsub + SB - 1) / SB) * p.ncb * (DG ? 1 : 4);  // work items: (m block, cout block[, parity]); DG: output items
  const int first = rs_xcd_remap(blockIdx.x, gridDim.x);
  // this block walks items first, first + grid, ...; DG: each output item as four consecutive units (the parity planes of dz)
  const int nitems = (ntiles - first + (int)gridDim.x - 1) / (int)gridDim.x * (DG ? 4 : 1);
  auto item_of = [&](int seq) __attribute__((always_inline)) {  // the (m block, cout block, parity) index of the block's unit `seq`
    return DG ? (((first + (seq >> 2) * (int)gridDim.x) << 2) | (seq & 3)) : first + seq * (int)gridDim.x;
  };
  const unsigned int lds0 = __builtin_amdgcn_readfirstlane(rb_lds_addr(smem));
  const long img1 = (long)p.Hs * p.Ws * p.C1 * (DG ? 4 : 1), img2 = (long)p.Hs * p.Ws * p.C2;  // (DG: dz images are 2 Hs x 2 Ws)

  // item -> (parity, m block, cout block)
  auto decode = [&](int it, int& py, int& px, int& mblk, int& nblk) __attribute__((always_inline)) {
    py = (it >> 1) & 1;
    px = it & 1;
    const int rest = it >> 2;
    mblk = rest / p.ncb;
    nblk = rest - mblk * p.ncb;
  };
  auto build_table = [&](int seq) __attribute__((always_inline)) {  // all threads; visible after the next barrier
    int py, px, mblk, nblk;
    decode(item_of(seq), py, px, mblk, nblk);
    const int sub0 = mblk * SB, nfirst = sub0 / per_img;
    int* tab = tabs + (seq & 1) * AROWS_PAD;
    for (int rho = tid; rho < AROWS_PAD; rho += 64 * NW) {
      int v = -1;
      if (rho < AROWS) {
        const int sb = rho / G::SBROWS, rem = rho - sb * G::SBROWS;
        const int hy = rem / G::PITCH, xs = rem - hy * G::PITCH;
        const int hx = hy >= G::HW ? -1 : (xs < G::HALF ? 2 * xs : 2 * (xs - G::HALF) + 1);
        const int sub = sub0 + sb;
        if (hx >= 0 && sub < p.nsub) {
          const int n = sub / per_img, r2 = sub - n * per_img;
          const int bby = r2 / p.BBX, bbx = r2 - bby * p.BBX;
          const int y = 2 * bby * PB - 1 + py + hy, x = 2 * bbx * PB - 1 + px + hx;
          if ((unsigned)y < (unsigned)p.Hs && (unsigned)x < (unsigned)p.Ws)
            v = DG ? ((n - nfirst) * 2 * p.Hs + 2 * y + 1 - py) * (2 * p.Ws) + 2 * x + 1 - px  // plane (1 - py, 1 - px) of dz, pixel (y, x)
                   : ((n - nfirst) * p.Hs + y) * p.Ws + x;
        }
      }
      tab[rho] = v;
    }
    rs_lds_writes_done();  // (read by every wave behind a later barrier, which hipcc emits bare: common.h)
  };

  // ---- fetch side: runs ONE chunk ahead of the MFMAs, across item boundaries (the next item's first chunk streams in while
  //      this item's outputs are stored).  Instruction ii = wave + NW j copies 16 rows of 64 bytes (4 lanes per row); ii < IA
  //      halo rows, else filter rows.  Per lane and instruction ONE register: the byte offset for the current item and source.
  const int ra = lane >> 2, pp = lane & 3;
  int doff[NI];
  int f_seq = 0, f_kc = 0, f_g = 0;  // item / chunk / global chunk being fetched
  __amdgpu_buffer_rsrc_t rsrc1 = rb_make_rsrc(p.src1, 0), rsrc2 = rsrc1, rsrcu = rsrc1;
  auto fetch_item = [&]() __attribute__((always_inline)) {  // f_seq changed: descriptors and filter offsets of the new item
    int py, px, mblk, nblk;
    decode(item_of(f_seq), py, px, mblk, nblk);
    const int nfirst = __builtin_amdgcn_readfirstlane((mblk * SB) / per_img);
    rsrc1 = rb_make_rsrc(p.src1 + nfirst * img1, (long)(p.N - nfirst) * img1 * 4);
    rsrc2 = rb_make_rsrc(p.C2 ? p.src2 + nfirst * img2 : p.src1, (long)(p.N - nfirst) * img2 * 4);
    rsrcu = rb_make_rsrc(p.u + (long)(2 * py + px) * 9 * p.Cout * Cin, (long)9 * p.Cout * Cin * 4);
#pragma unroll
    for (int j = 0; j < NI; ++j) {
      const int ii = wave + NW * j;
      if (ii >= IA) {
        const int w = 16 * (ii - IA) + ra;  // filter row: xi * BN + cout
        const int xi = w / BN, co = w - xi * BN;
        doff[j] = ((xi * p.Cout + nblk * BN + co) * Cin) * 4 + ((pp ^ wino_swz(w)) & 3) * 16;
      }
    }
  };
  auto set_source = [&](bool src_first) __attribute__((always_inline)) {
    const int cb = (src_first ? p.C1 : p.C2) * 4;
    const int* tab = tabs + (f_seq & 1) * AROWS_PAD;
#pragma unroll
    for (int j = 0; j < NI; ++j) {
      const int ii = wave + NW * j;
      if (ii < IA) {
        const int rho = 16 * ii + ra;
        const int pix = tab[rho];
        doff[j] = pix < 0 ? kDmaOOB : pix * cb + ((pp ^ wino_swz(rho)) & 3) * 16;
      }
    }
  };
  // issue chunk (f_seq, f_kc) into stage f_g & 1, then advance.  Past the block's last chunk (`past`: once, during its very
  // last MFMAs) the previous chunk's pieces are simply issued again into the free stage -- nobody reads them, and the MFMA
  // loop needs no second, fetch-less copy of itself (which cost registers: hipcc kept the two copies' live ranges apart).
  auto fetch_chunk = [&](auto interleave, bool past) __attribute__((always_inline)) {
    const int c0 = f_kc * KC;
    const bool src_first = c0 < p.C1;
    if (!past) {
      if (f_kc == 0) fetch_item();
      if (c0 == 0 || c0 == p.C1) set_source(src_first);
    }
    const unsigned int fL = lds0 + (f_g & 1) * STAGE;
    const int fsa = (src_first ? c0 : c0 - p.C1) * 4, fsu = c0 * 4;
    interleave([&](int j) __attribute__((always_inline)) {
      const int ii = wave + NW * j;  // wave-uniform
      if (ii < IA) {
        if (src_first) rb_dma16s(rsrc1, fL + ii * 1024, doff[j], fsa);
        else rb_dma16s(rsrc2, fL + ii * 1024, doff[j], fsa);
      } else if (ii < IA + IB) {
        rb_dma16s(rsrcu, fL + ii * 1024, doff[j], fsu);
      }
    });
    ++f_g;
    if (!past && ++f_kc == nk) {
      f_kc = 0;
      ++f_seq;
    }
  };

  // ---- fragment addressing (item-independent): lane = tile (lane & 15) of the wave's 16, 16-byte piece (lane >> 4) --------
  const int l15 = lane & 15, pc = lane >> 4;
  const int t = 16 * tg + wino_lane_tile<PB>(l15);
  const int tsb = t / (PB * PB), tq = t - tsb * (PB * PB);
  const int tty = tq / PB, ttx = tq - tty * PB;
  int addrA[3][3];
  {
    const int rho0 = tsb * G::SBROWS + 2 * tty * G::PITCH + ttx;
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        const int rho = rho0 + r * G::PITCH + (c == 0 ? 0 : (c == 1 ? G::HALF : 1));
        addrA[r][c] = rho * 64 + ((pc ^ wino_swz(rho)) & 3) * 16;
      }
  }
  // filter rows xi * BN + 32 cg + 16 m + l15: BN and 16 are multiples of 8, so the swizzle only sees l15
  const int addrB = AROWS_PAD * 64 + (16 * NM * cg + l15) * 64 + ((pc ^ wino_swz(l15)) & 3) * 16;  // + (xi * BN + 16 m) * 64

  build_table(0);
  __syncthreads();
  fetch_chunk([&](auto issue) __attribute__((always_inline)) {
#pragma unroll
    for (int j = 0; j < NI; ++j) issue(j);
  }, false);
  const int total = nitems * nk;
  const int Ho = 2 * p.Hs, Wo = 2 * p.Ws;

  int g = 0;
  f32x4 acc[9][NM];
  for (int seq = 0; seq < nitems; ++seq) {
    if (!DG || (seq & 3) == 0) {  // (DG: the four parity planes of an output item accumulate)
#pragma unroll
      for (int x = 0; x < 9; ++x)
#pragma unroll
        for (int m = 0; m < NM; ++m) acc[x][m] = f32x4{0.f, 0.f, 0.f, 0.f};
    }

    for (int kc = 0; kc < nk; ++kc, ++g) {
      rb_dma_wait();
      __syncthreads();  // chunk g is in stage g & 1; stage (g + 1) & 1 is free again
      const unsigned char* L = smem + (g & 1) * STAGE;
      f32x4 P[3][3];
#pragma unroll
      for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int c = 0; c < 3; ++c) P[r][c] = *reinterpret_cast<const f32x4*>(L + addrA[r][c]);
      const bool more = g + 1 < total;
      // LDS-DMA pieces of the NEXT chunk, spread evenly over this chunk's MFMAs: the CU's DMA path moves ~23 B/clk and this
      // kernel needs ~12 B/clk of it (55 KB per 4608 MFMA cycles); issued in a burst the pieces queue up and hold the issuing
      // waves (front-loading them was measured: no gain)
      constexpr int NMMA = 36 * NM, PSTEP = NMMA / NI >= 1 ? NMMA / NI : 1;
      // filter pieces are fetched one transformed position ahead of the MFMAs that use them (the first pair goes out before
      // the transform's vector ops): in source order hipcc otherwise parks an LDS round trip in front of every group of 8 MFMAs
      f32x4 Bq[9][NM];
#pragma unroll
      for (int m = 0; m < NM; ++m) Bq[0][m] = *reinterpret_cast<const f32x4*>(L + addrB + (16 * m) * 64);
      __builtin_amdgcn_sched_barrier(0);
      // V = B^T d B: along x, then along y ([d0 - d1, d1, d2 - d1] each way), on 2-float halves with v_pk_add_f32 (two fp32
      // subtractions per instruction; written as 4-float vector code hipcc emits scalar v_sub_f32: measured -4 % on the 3x3 form)
      f32x4 V[9];
      {
        f32x2 Tl[3][3], Th[3][3];
#pragma unroll
        for (int r = 0; r < 3; ++r) {
          const f32x2 l0 = {P[r][0][0], P[r][0][1]}, h0 = {P[r][0][2], P[r][0][3]};
          const f32x2 l1 = {P[r][1][0], P[r][1][1]}, h1 = {P[r][1][2], P[r][1][3]};
          const f32x2 l2 = {P[r][2][0], P[r][2][1]}, h2 = {P[r][2][2], P[r][2][3]};
          Tl[r][0] = pk_sub(l0, l1);
          Tl[r][1] = l1;
          Tl[r][2] = pk_sub(l2, l1);
          Th[r][0] = pk_sub(h0, h1);
          Th[r][1] = h1;
          Th[r][2] = pk_sub(h2, h1);
        }
#pragma unroll
        for (int c = 0; c < 3; ++c) {
          const f32x2 a = pk_sub(Tl[0][c], Tl[1][c]), b = pk_sub(Tl[2][c], Tl[1][c]);
          const f32x2 d = pk_sub(Th[0][c], Th[1][c]), e = pk_sub(Th[2][c], Th[1][c]);
          V[0 * 3 + c] = f32x4{a[0], a[1], d[0], d[1]};
          V[1 * 3 + c] = f32x4{Tl[1][c][0], Tl[1][c][1], Th[1][c][0], Th[1][c][1]};
          V[2 * 3 + c] = f32x4{b[0], b[1], e[0], e[1]};
        }
      }
      auto mfmas = [&](auto issue) __attribute__((always_inline)) {
#pragma unroll
        for (int x = 0; x < 9; ++x) {
          if (x + 1 < 9) {
#pragma unroll
            for (int m = 0; m < NM; ++m) Bq[x + 1][m] = *reinterpret_cast<const f32x4*>(L + addrB + ((x + 1) * BN + 16 * m) * 64);
            __builtin_amdgcn_sched_barrier(0);  // (keep these reads in front of the MFMAs below: hipcc sinks them to their first use)
          }
#pragma unroll
          for (int k = 0; k < 4; ++k)
#pragma unroll
            for (int m = 0; m < NM; ++m) {
              const int q = (x * 4 + k) * NM + m;  // MFMA index within the chunk (compile-time)
              if (q % PSTEP == 0 && q / PSTEP < NI) issue(q / PSTEP);
              acc[x][m] = __builtin_amdgcn_mfma_f32_16x16x4f32(Bq[x][m][k], V[x][k], acc[x][m], 0, 0, 0);
            }
        }
      };
      fetch_chunk([&](auto issue) __attribute__((always_inline)) { mfmas(issue); }, !more);

      if (kc == 0 && seq + 1 < nitems) build_table(seq + 1);  // (its buffer held item seq - 1's table: the fetch side left it a whole item ago)
    }

    // ---- Y = A^T M A, ReLU, store: lane = tile l15, couts 32 cg + 16 m + 4 pc + (0..3) -------------------------------------
    if (DG && (seq & 3) != 3) continue;
    int py, px, mblk, nblk;
    decode(item_of(seq), py, px, mblk, nblk);
    const int sub = mblk * SB + tsb;
    if constexpr (DG) {
      if (sub < p.nsub) {
        const int n = sub / per_img, r2 = sub - n * per_img;
        const int bby = r2 / p.BBX, bbx = r2 - bby * p.BBX;
        const int a0 = 2 * (bby * PB + tty), b0 = 2 * (bbx * PB + ttx);
        // destination of this wave's couts (block-uniform: a cout block never straddles csplit)
        const int co = nblk * BN + 16 * NM * cg + 4 * pc;
        const bool second = p.out2 != nullptr && nblk * BN >= p.csplit;
        float* ob = second ? p.out2 : p.out;
        const float* mb = second ? p.mask2 : p.mask1;
        const int ostride = p.out2 ? (second ? p.Cout - p.csplit : p.csplit) : p.Cout;
        const int ocol = second ? co - p.csplit : co;
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
          for (int v = 0; v < 2; ++v) {
            const int a = a0 + u, b = b0 + v;
            if (a >= p.Hs || b >= p.Ws) continue;
            const long o = ((long)(n * p.Hs + a) * p.Ws + b) * ostride + ocol;
#pragma unroll
            for (int m = 0; m < NM; ++m) {
              f32x4 y = (acc[u * 3 + v][m] + acc[u * 3 + v + 1][m]) + (acc[(u + 1) * 3 + v][m] + acc[(u + 1) * 3 + v + 1][m]);
              if (mb) {
                const f32x4 z = *reinterpret_cast<const f32x4*>(mb + o + 16 * m);
#pragma unroll
                for (int e = 0; e < 4; ++e) y[e] = z[e] > 0.f ? y[e] : 0.f;
              }
              *reinterpret_cast<f32x4*>(ob + o + 16 * m) = y;
            }
          }
      }
      continue;
    }
    if (sub < p.nsub) {
      const int n = sub / per_img, r2 = sub - n * per_img;
      const int bby = r2 / p.BBX, bbx = r2 - bby * p.BBX;
      const int a0 = 2 * (bby * PB + tty), b0 = 2 * (bbx * PB + ttx);
      float* obase = p.out + nblk * BN + 16 * NM * cg + 4 * pc;
#pragma unroll
      for (int u = 0; u < 2; ++u)
#pragma unroll
        for (int v = 0; v < 2; ++v) {
          const int a = a0 + u, b = b0 + v;
          if (a >= p.Hs || b >= p.Ws) continue;
          float* o = obase + ((long)(n * Ho + 2 * a + py) * Wo + 2 * b + px) * p.Cout;
#pragma unroll
          for (int m = 0; m < NM; ++m) {
            f32x4 y = (acc[u * 3 + v][m] + acc[u * 3 + v + 1][m]) + (acc[(u + 1) * 3 + v][m] + acc[(u + 1) * 3 + v + 1][m]);
            if (p.relu) {
#pragma unroll
              for (int e = 0; e < 4; ++e) y[e] = fmaxf(y[e], 0.f);
            }
            *reinterpret_cast<f32x4*>(o + 16 * m) = y;
          }
        }
    }
  }
  rb_dma_wait();  // (the re-issued pieces of the last chunk: landed before this block's LDS is handed to the next one)
}

// U = G g G^T per (parity, cout, cin): phase pack [4][Cout][2][2][Cin] -> [4][9][Cout][Cin]
__global__ void pack_wino_phase_weight_kernel(const float* __restrict__ w, float* __restrict__ u, int Cout, int Cin, long total) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;  // over [4][Cout][Cin]
  if (i >= total) return;
  const int ci = (int)(i % Cin);
  const long r = i / Cin;
  const int co = (int)(r % Cout), ph = (int)(r / Cout);
  const float* g = w + (((long)ph * Cout + co) * 4) * Cin + ci;  // [2][2][Cin]
  const float g00 = g[0], g01 = g[Cin], g10 = g[2 * (long)Cin], g11 = g[3 * (long)Cin];
  const float v[9] = {g00, g00 + g01, g01, g00 + g10, (g00 + g01) + (g10 + g11), g01 + g11, g10, g10 + g11, g11};
#pragma unroll
  for (int x = 0; x < 9; ++x) u[(((long)ph * 9 + x) * Cout + co) * Cin + ci] = v[x];
}

// Data-gradient filters: wd [Cin][4][4][Cout] (rs_pack_dgrad_phase_weight_dt: the 4x4 / stride-2 taps over dz) -> U [4][9][Cin][Cout].
// Unit (py, px) of an output item reads plane (1 - py, 1 - px) of dz with the 2x2 taps g[r][s] = wd[ci][2 r + py][2 s + px][co]
// (= w4[2 r + 1 - pa][2 s + 1 - pb]); U = G g G^T as in the forward pack.
__global__ void pack_wino_dgrad_weight_kernel(const float* __restrict__ wd, float* __restrict__ u, int Cin, int Cout, long total) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;  // over [4][Cin][Cout]
  if (i >= total) return;
  const int co = (int)(i % Cout);
  const long r = i / Cout;
  const int ci = (int)(r % Cin), ph = (int)(r / Cin), py = ph >> 1, px = ph & 1;
  const float* g = wd + (long)ci * 16 * Cout + co;  // [4][4][Cout]
  const float g00 = g[(long)(py * 4 + px) * Cout], g01 = g[(long)(py * 4 + 2 + px) * Cout];
  const float g10 = g[(long)((2 + py) * 4 + px) * Cout], g11 = g[(long)((2 + py) * 4 + 2 + px) * Cout];
  const float v[9] = {g00, g00 + g01, g01, g00 + g10, (g00 + g01) + (g10 + g11), g01 + g11, g10, g10 + g11, g11};
#pragma unroll
  for (int x = 0; x < 9; ++x) u[(((long)ph * 9 + x) * Cin + ci) * Cout + co] = v[x];
}

int wino_cus() {
  static const int n = [] {
    int dev = 0, cus = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0)
      cus = 256;
    return cus;
  }();
  return n;
}

bool wino_wide() {  // (A/B knob while the wide block is being measured: ROBOSAT_WINO_WIDE=0 keeps 64 tiles x 64 couts)
  return rs_knobs().wino_wide != 0;
}

struct WinoPlan {
  int pb, wgn;  // 8 | 4; 2 (64 couts per block) | 1 (32)
  int sb;       // sub-blocks per block
  bool worth;   // enough work items to fill the chip (else the generic phase kernel is the faster choice)
  bool wide;    // 128 tiles x 64 couts per block instead of 64 x 64 (see wino_plan)
};

bool wino_plan(const rs_conv_desc* d, WinoPlan* pl) {
  if (!d || d->N <= 0 || d->Hs <= 0 || d->Ws <= 0 || d->C1 <= 0 || (d->C1 % 16) || d->C2 < 0 || (d->C2 % 16) || d->Cout <= 0 ||
      (d->Cout % 32))
    return false;
  if (!(d->ups == 1 && d->kh == 3 && d->kw == 3 && d->stride == 1 && d->pad == 1 && d->Ho == 2 * d->Hs && d->Wo == 2 * d->Ws)) return false;
  const int ty = (d->Hs + 1) / 2, tx = (d->Ws + 1) / 2;
  if (ty < 4 || tx < 4) return false;  // (tiny layers: the halo of a 64-tile block would not fit; the generic kernel runs them)
  pl->pb = (ty >= 8 && tx >= 8) ? 8 : 4;
  pl->wgn = (d->Cout % 64 == 0) ? 2 : 1;
  pl->sb = 16 * (8 / pl->wgn) / (pl->pb * pl->pb);
  if (d->C1 + d->C2 < 32) return false;  // (the table of the next item is built during an item's first chunk: two chunks at least)
  const long cmax = d->C1 > d->C2 ? d->C1 : d->C2;
  if ((long)(pl->sb + 1) * d->Hs * d->Ws * cmax * 4 >= (1L << 31)) return false;  // 32-bit byte offsets within a block's images
  if ((long)9 * d->Cout * (d->C1 + d->C2) * 4 >= (1L << 31)) return false;
  if ((long)d->N * d->Ho * d->Wo >= (1L << 31)) return false;
  // one block per CU and no other block to hide behind: a launch with fewer work items than ~half the CUs (the `center`
  // block at small batches) is faster on the generic phase kernel's 4-blocks-per-CU grid
  // Worth it -- a decision on the layer's GEOMETRY alone, never on the batch size: a tile's probabilities must not depend on
  // its batch neighbours (the two forms differ in fp32 summation order), so the same kernel runs a layer at every N.  One
  // full 8x8 patch of tiles per image side at least (Hs, Ws >= 15): with fewer, a block's tiles straddle images and the
  // persistent one-block-per-CU grid is short of work items at any batch size (`center` at 512^2: 0.32 vs 0.21 ms).
  pl->worth = pl->pb == 8;
  // Block shape for the 64-cout layers: 128 tiles x 64 couts (a wave = 16 tiles x 64 couts: twice the MFMAs per transformed
  // patch and per filter byte; +3-5 % on dec1-dec3) when that still leaves two work items per CU, else 64 tiles x 64 couts
  // (dec0 at bs 16: 128 wide items on 256 CUs would halve the chip).  Either shape accumulates every output in the same order
  // -- they differ in which wave computes what --, so the choice may depend on the batch size without making a tile's
  // output depend on it.
  pl->wide = false;
  if (pl->pb == 8 && pl->wgn == 2 && wino_wide()) {
    const long nsub = (long)d->N * rs_cdiv(ty, 8) * rs_cdiv(tx, 8);
    pl->wide = (nsub + 1) / 2 * (d->Cout / 64) * 4 >= 2L * wino_cus();
    if (pl->wide) pl->sb = 2;
  }
  return true;
}

}  // namespace

extern "C" int rs_conv2d_phase_wino_ok(const rs_conv_desc* d) {
  WinoPlan pl;
  if (!wino_plan(d, &pl)) return 0;
  return pl.worth ? 1 : 2;
}

extern "C" const char* rs_conv2d_phase_wino_name(const rs_conv_desc* d) {
  WinoPlan pl;
  if (!wino_plan(d, &pl)) return "";
  if (pl.pb == 8) return pl.wide ? "conv_wino_f32<phase,p8,128x64>" : (pl.wgn == 2 ? "conv_wino_f32<phase,p8,64x64>" : "conv_wino_f32<phase,p8,128x32>");
  return pl.wgn == 2 ? "conv_wino_f32<phase,p4,64x64>" : "conv_wino_f32<phase,p4,128x32>";
}

extern "C" int rs_pack_wino_phase_weight(const float* w_phase, float* u, int Cout, int Cin, rs_stream_t stream) {
  if (!w_phase || !u || Cout <= 0 || Cin <= 0) return RS_EINVAL;
  const long total = 4L * Cout * Cin;
  pack_wino_phase_weight_kernel<<<rs_cdiv(total, 256), 256, 0, (hipStream_t)stream>>>(w_phase, u, Cout, Cin, total);
  return RS_LAUNCH_RESULT();
}

extern "C" int rs_conv2d_fwd_phase_wino(const rs_conv_desc* d, const float* src1, const float* src2, const float* u, float* out,
                                        rs_stream_t stream) {
  WinoPlan pl;
  if (!wino_plan(d, &pl) || !src1 || !u || !out || (d->C2 > 0 && !src2)) return RS_EINVAL;
  WinoArgs a;
  a.src1 = src1;
  a.src2 = src2;
  a.u = u;
  a.out = out;
  a.N = d->N;
  a.Hs = d->Hs;
  a.Ws = d->Ws;
  a.C1 = d->C1;
  a.C2 = d->C2;
  a.Cout = d->Cout;
  a.TY = (d->Hs + 1) / 2;
  a.TX = (d->Ws + 1) / 2;
  a.BBY = rs_cdiv(a.TY, pl.pb);
  a.BBX = rs_cdiv(a.TX, pl.pb);
  a.nsub = d->N * a.BBY * a.BBX;
  a.ncb = d->Cout / (32 * pl.wgn);
  a.relu = d->relu;
  a.out2 = nullptr;
  a.mask1 = a.mask2 = nullptr;
  a.csplit = 0;
  const long items = (long)rs_cdiv(a.nsub, pl.sb) * a.ncb * 4;
  if (items >= (1L << 31)) return RS_EINVAL;
  // persistent: one block per CU (its LDS stages fill the CU), items dealt round-robin
  const int grid = (int)(items < wino_cus() ? items : wino_cus());
  hipStream_t s = (hipStream_t)stream;
  if (pl.wide) conv_wino_f32_kernel<8, 8, 1, 4><<<grid, 512, 0, s>>>(a);  // 128 tiles x 64 couts
  else if (pl.pb == 8 && pl.wgn == 2) conv_wino_f32_kernel<8, 4, 2><<<grid, 512, 0, s>>>(a);
  else if (pl.pb == 8) conv_wino_f32_kernel<8, 8, 1><<<grid, 512, 0, s>>>(a);
  else if (pl.wgn == 2) conv_wino_f32_kernel<4, 4, 2><<<grid, 512, 0, s>>>(a);
  else conv_wino_f32_kernel<4, 8, 1><<<grid, 512, 0, s>>>(a);
  return RS_LAUNCH_RESULT();
}

// ---- the DecoderBlock's data gradient in the same form (round 6) ---------------------------------------------------------------------
// `d` is the FORWARD layer's descriptor (DecoderBlock: ups = 1, 3x3, pad 1; Hs x Ws the source grid, C1 + C2 the concatenated input
// channels, Cout the block's output channels): the gradient has Cout INPUT channels (dz, at 2 Hs x 2 Ws) and C1 + C2 OUTPUT channels.
namespace {
bool wino_dgrad_plan(const rs_conv_desc* d, rs_conv_desc* g, WinoPlan* pl) {
  if (!d || d->C1 <= 0 || d->C2 < 0 || d->Cout <= 0) return false;
  *g = *d;
  g->C1 = d->Cout;
  g->C2 = 0;
  g->Cout = d->C1 + d->C2;
  if (!wino_plan(g, pl) || !pl->worth) return false;
  // (dz images are four times the plan's: 32-bit byte offsets within a block's images)
  if ((long)(pl->sb + 1) * 4 * d->Hs * d->Ws * g->C1 * 4 >= (1L << 31)) return false;
  if (g->Cout % 64) return false;  // (64-cout blocks only: every DecoderBlock input of the U-Net is a multiple of 64 channels)
  return true;
}
}  // namespace

extern "C" int rs_conv2d_dgrad_phase_wino_ok(const rs_conv_desc* d) {
  rs_conv_desc g;
  WinoPlan pl;
  return wino_dgrad_plan(d, &g, &pl) ? 1 : 0;
}

extern "C" const char* rs_conv2d_dgrad_phase_wino_name(const rs_conv_desc* d) {
  rs_conv_desc g;
  WinoPlan pl;
  if (!wino_dgrad_plan(d, &g, &pl)) return "";
  return pl.wide ? "conv_wino_f32<dgrad4x4,p8,128x64>" : "conv_wino_f32<dgrad4x4,p8,64x64>";
}

extern "C" int rs_pack_wino_dgrad_weight(const float* wd4x4, float* u, int Cin, int Cout, rs_stream_t stream) {
  if (!wd4x4 || !u || Cin <= 0 || Cout <= 0) return RS_EINVAL;
  const long total = 4L * Cin * Cout;
  pack_wino_dgrad_weight_kernel<<<rs_cdiv(total, 256), 256, 0, (hipStream_t)stream>>>(wd4x4, u, Cin, Cout, total);
  return RS_LAUNCH_RESULT();
}

extern "C" int rs_conv2d_dgrad_phase_wino(const rs_conv_desc* d, const float* dz, const float* u, float* out, const float* mask,
                                          float* out2, const float* mask2, int csplit, rs_stream_t stream) {
  rs_conv_desc g;
  WinoPlan pl;
  if (!wino_dgrad_plan(d, &g, &pl) || !dz || !u || !out) return RS_EINVAL;
  if (out2 ? (csplit <= 0 || csplit >= g.Cout || (csplit % 64) != 0) : (csplit != 0 || mask2 != nullptr)) return RS_EINVAL;
  WinoArgs a;
  a.src1 = dz;
  a.src2 = nullptr;
  a.u = u;
  a.out = out;
  a.out2 = out2;
  a.mask1 = mask;
  a.mask2 = mask2;
  a.csplit = csplit;
  a.N = d->N;
  a.Hs = d->Hs;
  a.Ws = d->Ws;
  a.C1 = g.C1;
  a.C2 = 0;
  a.Cout = g.Cout;
  a.TY = (d->Hs + 1) / 2;
  a.TX = (d->Ws + 1) / 2;
  a.BBY = rs_cdiv(a.TY, pl.pb);
  a.BBX = rs_cdiv(a.TX, pl.pb);
  a.nsub = d->N * a.BBY * a.BBX;
  a.ncb = g.Cout / 64;
  a.relu = 0;
  // (the wide block where that leaves two OUTPUT items per CU: there is no parity factor in the item count here)
  bool wide = false;
  if (wino_wide()) wide = (long)((a.nsub + 1) / 2) * a.ncb >= 2L * wino_cus();
  const int sb = wide ? 2 : 1;
  const long items = (long)rs_cdiv(a.nsub, sb) * a.ncb;
  if (items >= (1L << 29)) return RS_EINVAL;
  const int grid = (int)(items < wino_cus() ? items : wino_cus());
  hipStream_t s = (hipStream_t)stream;
  if (wide) conv_wino_f32_kernel<8, 8, 1, 4, true><<<grid, 512, 0, s>>>(a);
  else conv_wino_f32_kernel<8, 4, 2, 2, true><<<grid, 512, 0, s>>>(a);
  return RS_LAUNCH_RESULT();
}

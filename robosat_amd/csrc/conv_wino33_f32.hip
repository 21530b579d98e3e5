// conv_wino33_f32.hip -- the stride-1 3x3 / pad-1 convolutions of the fp32 PREDICT path (the Bottleneck conv2 of every
// ResNet-50 block but the three strided ones, reference torchvision Bottleneck via robosat/unet.py:94,122-130, and dec5's
// ConvRelu, unet.py:32-44,139) as a WINOGRAD F(2x2, 3x3) convolution: each 2x2 block of outputs comes from a 4x4 input patch
// with 16 multiplies instead of 36,
//     Y = A^T [ (G g G^T) (.) (B^T d B) ] A,  B^T = [1 0 -1 0; 0 1 1 0; 0 -1 1 0; 0 1 0 -1],  G = [1 0 0; .5 .5 .5; .5 -.5 .5; 0 0 1],
//     A^T = [1 1 1 0; 0 1 -1 -1]
// i.e. 16 GEMMs [tiles x Cin] . [Cin x Cout] instead of 9 four times as tall: 4/9 of the multiply-adds.  Same reasoning and
// same machinery as the DecoderBlock's Winograd form (conv_wino_f32.hip, which has the long version): the fp32 matrix cores
// (157 TFLOP/s, v_mfma_f32_16x16x4_f32) are what bounds these layers at 0.75-0.84 of their peak in the generic kernel, the
// vector ALU and the LDS have slack for the transforms.  Input and output transforms have +-1 coefficients, the filter
// transform halves: nothing that matters in fp32 (the parity tests hold it to the generic kernel's bar).
//
//   block  = 8 waves, persistent (one per CU), work items = (16*TG tiles, 16*CG couts); a wave owns 16 tiles x 16 couts and
//            keeps 16 accumulators (one per transformed position): 64 registers, two waves per SIMD.
//   LDS    = per 16-channel chunk the block's source halo ((2*8 + 2)^2 pixels per 8x8 patch of tiles, 64 bytes each) + the
//            transformed filters U[16][couts][16 ch] by LDS-DMA, double buffered, fetched one chunk ahead across work items.
//   reads  = 16 ds_read_b128 for the lane's 4x4 patch, 32 vector add/sub for B^T d B, filter pieces one position ahead, 64
//            MFMAs per chunk; rows even-x-first at pitch 18, pieces XOR-swizzled with (row ^ row >> 1) & 3, lanes -> tiles by
//            a bit permutation: conflict-free for all 32 reads (exhaustive search; scripts/probes/wino_lds.py checks it).
//   store  = A^T M A (+ the eval-mode BatchNorm's scale / shift, ReLU) from registers: 4 consecutive couts per lane.
// Only the EVAL epilogue exists (predict, serve): the train-mode forward needs the raw output + BatchNorm partial sums and
// stays on the generic kernel.  Chosen by layer geometry (>= 8 tiles per image side, i.e. H, W >= 15), never by batch size.
#define RS_CONV_INSTANTIATE  // (for the LDS-DMA helpers of the header; no kernel of it is instantiated here)
#include "conv_igemm_dma_kernel.h"
#include "final_head.h"

namespace {

struct Wino33Args {
  const float* src;
  const float* u;      // [16][Cout][Cin]
  const float* scale;  // optional [Cout]
  const float* shift;  // optional [Cout]
  float* out;          // [N][H][W][Cout]
  int N, H, W, Cin, Cout;
  int BBY, BBX;  // 8x8 tile patches per image
  int nsub;      // N * BBY * BBX
  int ncb;       // cout blocks: Cout / BN
  int relu;
  // HEAD instantiation only -- dec5 + `self.final` (+ softmax / quantise / argmax) in one launch (reference unet.py:139-141,
  // tools/predict.py:87-103): the block's 32 couts never leave the CU, `out` is not written
  const float* hw;         // final.weight [hC][32]
  const float* hb;         // final.bias [hC] or null
  const double* hanchors;  // np.linspace(0, 1, 256) (mode 2)
  float* hout;             // NCHW fp32 logits / probabilities (modes 0, 1)
  unsigned char* hq;       // quantised probabilities (mode 2) / class indices (mode 3)
  int hC, hmode, hov;
  // STATS instantiations only -- the train-mode forward (torchvision BatchNorm2d under tools/train.py:169): `out` holds the RAW
  // convolution output and stats [m blocks][2][Cout] the per-block sum / sum of squares of it (rs_bn_finalize_stats's input)
  float* stats;
  // BWD instantiations only (round 6) -- the DATA gradient of such a layer (the same convolution over dy with the flipped, transposed
  // filters) arriving at a ReLU (+ BatchNorm) output: the ReLU mask as the forward activation (`mask`, sign) or one bit per element
  // (`mask_bits`, rs_bn_apply_bits_dt's), and -- with `stats` -- the partial sums (sum g, sum g * xhat) of BatchNorm's backward
  // against its input `bn_y` and statistics (what EPI_BWD of the generic kernel does: rs_conv2d_dgrad_bnstats_bits_dt)
  const float* mask;
  const unsigned char* mask_bits;
  const float* bn_y;
  const float* bn_mean;
  const float* bn_invstd;
};

typedef float f32x2 __attribute__((ext_vector_type(2)));
// two fp32 adds / subtractions per instruction (v_pk_add_f32; hipcc emits it for a + b but mostly not for a - b)
__device__ __forceinline__ f32x2 pk_add(f32x2 a, f32x2 b) {
  f32x2 d;
  asm("v_pk_add_f32 %0, %1, %2" : "=v"(d) : "v"(a), "v"(b));
  return d;
}
__device__ __forceinline__ f32x2 pk_sub(f32x2 a, f32x2 b) {
  f32x2 d;
  asm("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(d) : "v"(a), "v"(b));
  return d;
}
constexpr int kPB = 8, kHW = 2 * kPB + 2, kPITCH = kHW, kHALF = kHW / 2;
constexpr int kSBROWS = (kHW * kPITCH + 7) / 8 * 8;  // rows per sub-block, padded to the swizzle's period

__device__ __forceinline__ int w33_lane_tile(int l) { return (l & 1) | (((l >> 1) & 1) << 2) | (((l >> 2) & 1) << 1) | (l & 8); }
__device__ __forceinline__ int w33_swz(int row) { return (row ^ (row >> 1)) & 3; }

// HEAD: 0 = the layer alone; 1 / 2 / 3 = + self.final and logits-or-softmax / quantised probabilities / argmax (one
// instantiation per output kind: each carries only its own epilogue code -- the kernel's instructions are fetched cold on
// every launch, which a single-tile `rs serve` request pays for in full)
// MODE: 0 = the layer alone; 1 / 2 / 3 = + self.final (see above); 5 (round 6) = the layer's DATA gradient (BWD: see Wino33Args);
// 4 (round 6) = the layer alone + the partial sums of train-mode
// BatchNorm's statistics over the block's outputs (sum y, sum y^2 per cout and m block), summed in a fixed order: lane (its 2x2
// pixels) -> the wave's 16 tiles (DPP row rotations: no LDS) -> the block's tile groups (through LDS, picked up behind the first
// barrier of the block's next item like the head's exchange) -> one row of `stats` per m block.
template <int TG, int CG, int MODE = 0>
__global__ __launch_bounds__(512, 1) void conv_wino33_f32_kernel(const Wino33Args p) {
  constexpr int HEAD = (MODE >= 1 && MODE <= 3) ? MODE : 0;
  constexpr bool BWD = MODE == 5;
  constexpr bool STATS = MODE == 4 || BWD;  // (BWD: only when p.stats is given -- block-uniform)
  constexpr int NW = TG * CG;
  static_assert(NW == 8, "8 waves");
  static_assert(!HEAD || (TG == 4 && CG == 2), "the fused head: one 8x8 patch of tiles x all 32 couts per block");
  constexpr int BMT = 16 * TG, BN = 16 * CG;
  constexpr int SB = BMT / (kPB * kPB);
  static_assert(SB * kPB * kPB == BMT, "whole sub-blocks per block");
  constexpr int AROWS = SB * kSBROWS;
  constexpr int IA = (AROWS + 15) / 16, IB = 16 * BN / 16;
  constexpr int AROWS_PAD = IA * 16;
  constexpr int NI = (IA + IB + NW - 1) / NW;
  constexpr int STAGE = (AROWS_PAD + 16 * BN) * 64;
  constexpr int KC = 16;

  // HEAD: + the class weights [8][32] and biases [8], + the exchange of the two cout groups' partial logits [tile][pixel][8]
  //       (two exchange buffers: an item's second half runs behind the first barrier of the block's NEXT item)
  // (exchange rows of 36 floats: 16 lanes' 16-byte accesses at a 144-byte pitch fall into 16 different bank groups)
  constexpr int HEADW = kHeadMaxC * 32 + kHeadMaxC, XROW = 4 * kHeadMaxC + 4, XCH = BMT * XROW;
  constexpr int XST = 2 * TG * BN;  // STATS: per exchange buffer [2 sums][TG tile groups][BN couts]
  __shared__ __attribute__((aligned(16))) unsigned char smem[2 * STAGE + 2 * AROWS_PAD * 4 + (HEAD ? (HEADW + 2 * XCH) * 4 : 0) + (STATS ? 2 * XST * 4 : 0)];
  int* tabs = reinterpret_cast<int*>(smem + 2 * STAGE);
  float* hws = reinterpret_cast<float*>(smem + 2 * STAGE + 2 * AROWS_PAD * 4);
  float* xch = hws + HEADW;
  float* xst = reinterpret_cast<float*>(smem + 2 * STAGE + 2 * AROWS_PAD * 4);  // (STATS; never together with HEAD)

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int tg = wave % TG, cg = wave / TG;
  const int per_img = p.BBY * p.BBX;
  const int nk = p.Cin / KC;
  const int ntiles = ((p.nsub + SB - 1) / SB) * p.ncb;
  const int first = rs_xcd_remap(blockIdx.x, gridDim.x);
  const int nitems = (ntiles - first + (int)gridDim.x - 1) / (int)gridDim.x;
  const unsigned int lds0 = __builtin_amdgcn_readfirstlane(rb_lds_addr(smem));
  const long img = (long)p.H * p.W * p.Cin;

  auto decode = [&](int it, int& mblk, int& nblk) __attribute__((always_inline)) {
    mblk = it / p.ncb;
    nblk = it - mblk * p.ncb;
  };
  auto build_table = [&](int seq) __attribute__((always_inline)) {
    int mblk, nblk;
    decode(first + seq * (int)gridDim.x, mblk, nblk);
    const int sub0 = mblk * SB, nfirst = sub0 / per_img;
    int* tab = tabs + (seq & 1) * AROWS_PAD;
    for (int rho = tid; rho < AROWS_PAD; rho += 64 * NW) {
      int v = -1;
      if (rho < AROWS) {
        const int sb = rho / kSBROWS, rem = rho - sb * kSBROWS;
        const int hy = rem / kPITCH, xs = rem - hy * kPITCH;
        const int hx = hy >= kHW ? -1 : (xs < kHALF ? 2 * xs : 2 * (xs - kHALF) + 1);
        const int sub = sub0 + sb;
        if (hx >= 0 && sub < p.nsub) {
          const int n = sub / per_img, r2 = sub - n * per_img;
          const int bby = r2 / p.BBX, bbx = r2 - bby * p.BBX;
          const int y = 2 * bby * kPB - 1 + hy, x = 2 * bbx * kPB - 1 + hx;
          if ((unsigned)y < (unsigned)p.H && (unsigned)x < (unsigned)p.W) v = ((n - nfirst) * p.H + y) * p.W + x;
        }
      }
      tab[rho] = v;
    }
    rs_lds_writes_done();  // (read by every wave behind a later barrier, which hipcc emits bare: common.h)
  };

  // ---- fetch side (see conv_wino_f32.hip): one chunk ahead, across work items --------------------------------------------
  const int ra = lane >> 2, pp = lane & 3;
  int doff[NI];
  int f_seq = 0, f_kc = 0, f_g = 0;
  __amdgpu_buffer_rsrc_t rsrc = rb_make_rsrc(p.src, 0);
  const __amdgpu_buffer_rsrc_t rsrcu = rb_make_rsrc(p.u, (long)16 * p.Cout * p.Cin * 4);
  auto fetch_item = [&]() __attribute__((always_inline)) {
    int mblk, nblk;
    decode(first + f_seq * (int)gridDim.x, mblk, nblk);
    const int nfirst = __builtin_amdgcn_readfirstlane((mblk * SB) / per_img);
    rsrc = rb_make_rsrc(p.src + nfirst * img, (long)(p.N - nfirst) * img * 4);
    const int* tab = tabs + (f_seq & 1) * AROWS_PAD;
#pragma unroll
    for (int j = 0; j < NI; ++j) {
      const int ii = wave + NW * j;
      if (ii < IA) {
        const int rho = 16 * ii + ra;
        const int pix = tab[rho];
        doff[j] = pix < 0 ? kDmaOOB : pix * (p.Cin * 4) + ((pp ^ w33_swz(rho)) & 3) * 16;
      } else {
        const int w = 16 * (ii - IA) + ra;  // filter row: xi * BN + cout
        const int xi = w / BN, co = w - xi * BN;
        doff[j] = ((xi * p.Cout + nblk * BN + co) * p.Cin) * 4 + ((pp ^ w33_swz(w)) & 3) * 16;
      }
    }
  };
  // (`past`: beyond the block's last chunk the previous pieces are issued again into the free stage; see conv_wino_f32.hip)
  auto fetch_chunk = [&](auto interleave, bool past) __attribute__((always_inline)) {
    if (!past && f_kc == 0) fetch_item();
    const unsigned int fL = lds0 + (f_g & 1) * STAGE;
    const int fs = f_kc * KC * 4;
    interleave([&](int j) __attribute__((always_inline)) {
      const int ii = wave + NW * j;  // wave-uniform
      if (ii < IA) rb_dma16s(rsrc, fL + ii * 1024, doff[j], fs);
      else if (ii < IA + IB) rb_dma16s(rsrcu, fL + ii * 1024, doff[j], fs);
    });
    ++f_g;
    if (!past && ++f_kc == nk) {
      f_kc = 0;
      ++f_seq;
    }
  };

  // ---- fragment addressing: lane = tile (lane & 15) of the wave's 16, 16-byte piece (lane >> 4) ---------------------------
  const int l15 = lane & 15, pc = lane >> 4;
  const int t = 16 * tg + w33_lane_tile(l15);
  const int tsb = t / (kPB * kPB), tq = t - tsb * (kPB * kPB);
  const int tty = tq / kPB, ttx = tq - tty * kPB;
  int addrA[4][4];
  {
    const int rho0 = tsb * kSBROWS + 2 * tty * kPITCH + ttx;
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const int rho = rho0 + r * kPITCH + (c >> 1) + (c & 1) * kHALF;  // x = 2 ttx + c: even x first, then odd x
        addrA[r][c] = rho * 64 + ((pc ^ w33_swz(rho)) & 3) * 16;
      }
  }
  const int addrB = AROWS_PAD * 64 + (16 * cg + l15) * 64 + ((pc ^ w33_swz(l15)) & 3) * 16;  // + xi * BN * 64 (BN, 16 cg: multiples of 8)

  if constexpr (HEAD) {
    for (int f = threadIdx.x; f < HEADW; f += 64 * NW) {
      const int c = f < kHeadMaxC * 32 ? f / 32 : f - kHeadMaxC * 32;
      hws[f] = c >= p.hC ? 0.f : (f < kHeadMaxC * 32 ? p.hw[f] : (p.hb ? p.hb[c] : 0.f));
    }
  }
  build_table(0);
  __syncthreads();
  fetch_chunk([&](auto issue) __attribute__((always_inline)) {
#pragma unroll
    for (int j = 0; j < NI; ++j) issue(j);
  }, false);
  const int total = nitems * nk;

  // HEAD: the cg = 0 lanes' pixel of the previous item, waiting for the other cout group's half (finish_head)
  float hmine[kHeadMaxC];
  long hpix = -1;
  int hbuf = 0;
  auto finish_head = [&]() __attribute__((always_inline)) {
    if constexpr (HEAD) {
      if (cg == 0) {
        const float* xrow = xch + hbuf * XCH + (16 * tg + l15) * XROW + pc * kHeadMaxC;
        const f32x4 o0 = *reinterpret_cast<const f32x4*>(xrow), o1 = *reinterpret_cast<const f32x4*>(xrow + 4);
        float lg[kHeadMaxC];
#pragma unroll
        for (int c = 0; c < kHeadMaxC; ++c) lg[c] = (hmine[c] + (c < 4 ? o0[c] : o1[c - 4])) + hws[kHeadMaxC * 32 + c];
        if (hpix >= 0)
          rs_final_epilogue_rt(lg, p.hC, hpix, (long)p.H * p.W, HEAD == 1 ? (p.hmode & 1) : HEAD, p.hanchors, p.hq, p.hout, p.W, p.hov);
      }
    }
  };

  // STATS: the previous item's exchange buffer -> its row of `stats` (threads 0 .. 2 BN - 1: sum c of stat tid / BN over the tile groups)
  int st_buf = 0, st_row = 0, st_col = 0;
  auto finish_stats = [&]() __attribute__((always_inline)) {
    if constexpr (STATS) {
      if ((!BWD || p.stats) && threadIdx.x < 2 * BN) {
        const int which = threadIdx.x / BN, c = threadIdx.x - which * BN;
        const float* x = xst + st_buf * XST + which * TG * BN + c;
        float a = 0.f;
#pragma unroll
        for (int t = 0; t < TG; ++t) a += x[t * BN];
        p.stats[((long)st_row * 2 + which) * p.Cout + st_col + c] = a;
      }
    }
  };
  int g = 0;
  for (int seq = 0; seq < nitems; ++seq) {
    f32x4 acc[16];
#pragma unroll
    for (int x = 0; x < 16; ++x) acc[x] = f32x4{0.f, 0.f, 0.f, 0.f};

    for (int kc = 0; kc < nk; ++kc, ++g) {
      rb_dma_wait();
      __syncthreads();
      if (HEAD && kc == 0 && seq > 0) finish_head();  // (behind this barrier the other cout group's partial logits are in LDS)
      if (STATS && kc == 0 && seq > 0) finish_stats();
      const unsigned char* L = smem + (g & 1) * STAGE;
      f32x4 V[16];
      {
        // (the transform on 2-float halves: hipcc turns those into v_pk_add_f32 -- two lanes' worth of fp32 adds per
        //  instruction -- where the 4-float form came out as scalar v_sub_f32)
        f32x2 Pl[4][4], Ph[4][4];
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
          for (int c = 0; c < 4; ++c) {
            const f32x4 q = *reinterpret_cast<const f32x4*>(L + addrA[r][c]);
            Pl[r][c] = f32x2{q[0], q[1]};
            Ph[r][c] = f32x2{q[2], q[3]};
          }
        f32x2 Tl[4][4], Th[4][4];  // B^T d B: along x, then along y ([d0 - d2, d1 + d2, d2 - d1, d1 - d3] each way)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          Tl[r][0] = pk_sub(Pl[r][0], Pl[r][2]);
          Tl[r][1] = pk_add(Pl[r][1], Pl[r][2]);
          Tl[r][2] = pk_sub(Pl[r][2], Pl[r][1]);
          Tl[r][3] = pk_sub(Pl[r][1], Pl[r][3]);
          Th[r][0] = pk_sub(Ph[r][0], Ph[r][2]);
          Th[r][1] = pk_add(Ph[r][1], Ph[r][2]);
          Th[r][2] = pk_sub(Ph[r][2], Ph[r][1]);
          Th[r][3] = pk_sub(Ph[r][1], Ph[r][3]);
        }
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          const f32x2 l0 = pk_sub(Tl[0][c], Tl[2][c]), l1 = pk_add(Tl[1][c], Tl[2][c]), l2 = pk_sub(Tl[2][c], Tl[1][c]), l3 = pk_sub(Tl[1][c], Tl[3][c]);
          const f32x2 h0 = pk_sub(Th[0][c], Th[2][c]), h1 = pk_add(Th[1][c], Th[2][c]), h2 = pk_sub(Th[2][c], Th[1][c]), h3 = pk_sub(Th[1][c], Th[3][c]);
          V[0 * 4 + c] = f32x4{l0[0], l0[1], h0[0], h0[1]};
          V[1 * 4 + c] = f32x4{l1[0], l1[1], h1[0], h1[1]};
          V[2 * 4 + c] = f32x4{l2[0], l2[1], h2[0], h2[1]};
          V[3 * 4 + c] = f32x4{l3[0], l3[1], h3[0], h3[1]};
        }
      }
      const bool more = g + 1 < total;
      constexpr int NMMA = 64, PSTEP = NMMA / NI >= 1 ? NMMA / NI : 1;
      f32x4 Bq[16];
      Bq[0] = *reinterpret_cast<const f32x4*>(L + addrB);
      __builtin_amdgcn_sched_barrier(0);
      auto mfmas = [&](auto issue) __attribute__((always_inline)) {
#pragma unroll
        for (int x = 0; x < 16; ++x) {
          if (x + 1 < 16) {
            Bq[x + 1] = *reinterpret_cast<const f32x4*>(L + addrB + (x + 1) * BN * 64);
            __builtin_amdgcn_sched_barrier(0);  // (keep the read in front of the MFMAs below)
          }
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            const int q = x * 4 + k;
            if (q % PSTEP == 0 && q / PSTEP < NI) issue(q / PSTEP);
            acc[x] = __builtin_amdgcn_mfma_f32_16x16x4f32(Bq[x][k], V[x][k], acc[x], 0, 0, 0);
          }
        }
      };
      fetch_chunk([&](auto issue) __attribute__((always_inline)) { mfmas(issue); }, !more);
      if (kc == 0 && seq + 1 < nitems) build_table(seq + 1);
    }

    // ---- Y = A^T M A (y0 = m0 + m1 + m2, y1 = m1 - m2 - m3 each way), scale / shift, ReLU, store --------------------------
    int mblk, nblk;
    decode(first + seq * (int)gridDim.x, mblk, nblk);
    const int sub = mblk * SB + tsb;
    const bool live = sub < p.nsub;
    const int n = sub / per_img, r2 = sub - n * per_img;
    const int bby = r2 / p.BBX, bbx = r2 - bby * p.BBX;
    const int a0 = 2 * (bby * kPB + tty), b0 = 2 * (bbx * kPB + ttx);
    const int co = nblk * BN + 16 * cg + 4 * pc;
    f32x4 Y[2][2];
    if (live || HEAD) {
      f32x4 sc = {1.f, 1.f, 1.f, 1.f}, sh = {0.f, 0.f, 0.f, 0.f};
      if (p.scale) sc = *reinterpret_cast<const f32x4*>(p.scale + co);
      if (p.shift) sh = *reinterpret_cast<const f32x4*>(p.shift + co);
      f32x4 R[2][4];  // rows transformed: R[u][b] = sum_a A^T[u][a] M[a][b]
#pragma unroll
      for (int b = 0; b < 4; ++b) {
        R[0][b] = (acc[0 * 4 + b] + acc[1 * 4 + b]) + acc[2 * 4 + b];
        R[1][b] = (acc[1 * 4 + b] - acc[2 * 4 + b]) - acc[3 * 4 + b];
      }
#pragma unroll
      for (int u = 0; u < 2; ++u)
#pragma unroll
        for (int v = 0; v < 2; ++v) {
          f32x4 y = v == 0 ? (R[u][0] + R[u][1]) + R[u][2] : (R[u][1] - R[u][2]) - R[u][3];
          y = y * sc + sh;
          if (p.relu) {
#pragma unroll
            for (int e = 0; e < 4; ++e) y[e] = fmaxf(y[e], 0.f);
          }
          Y[u][v] = y;
        }
    }
    if constexpr (!HEAD) {
      f32x4 s0 = {0.f, 0.f, 0.f, 0.f}, s1 = {0.f, 0.f, 0.f, 0.f};
      if (live) {
        if constexpr (BWD) {
          // all of the item's mask / BatchNorm-input pieces are requested first (one HBM round trip, not four), then combined and stored
          long o[4];
          bool ok[4];
          f32x4 zm[4], yv[4];
          unsigned int bits[4];
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const int a = a0 + (q >> 1), b = b0 + (q & 1);
            ok[q] = a < p.H && b < p.W;
            o[q] = ok[q] ? ((long)(n * p.H + a) * p.W + b) * p.Cout + co : (long)co;
          }
          if (p.mask) {
#pragma unroll
            for (int q = 0; q < 4; ++q) zm[q] = *reinterpret_cast<const f32x4*>(p.mask + o[q]);
          }
          if (p.mask_bits) {
#pragma unroll
            for (int q = 0; q < 4; ++q) bits[q] = p.mask_bits[o[q] >> 3];
          }
          f32x4 mu = {0.f, 0.f, 0.f, 0.f}, is = mu;
          if (p.stats) {
#pragma unroll
            for (int q = 0; q < 4; ++q) yv[q] = *reinterpret_cast<const f32x4*>(p.bn_y + o[q]);
            mu = *reinterpret_cast<const f32x4*>(p.bn_mean + co);
            is = *reinterpret_cast<const f32x4*>(p.bn_invstd + co);
          }
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            f32x4 y = Y[q >> 1][q & 1];
            if (p.mask) {
#pragma unroll
              for (int e = 0; e < 4; ++e) y[e] = zm[q][e] > 0.f ? y[e] : 0.f;
            }
            if (p.mask_bits) {
              const unsigned int nib = bits[q] >> (o[q] & 4);
#pragma unroll
              for (int e = 0; e < 4; ++e) y[e] = ((nib >> e) & 1u) ? y[e] : 0.f;
            }
            if (!ok[q]) continue;
            *reinterpret_cast<f32x4*>(p.out + o[q]) = y;
            if (p.stats) {
              s0 += y;
              s1 += y * ((yv[q] - mu) * is);
            }
          }
        } else {
#pragma unroll
          for (int u = 0; u < 2; ++u)
#pragma unroll
            for (int v = 0; v < 2; ++v) {
              const int a = a0 + u, b = b0 + v;
              if (a >= p.H || b >= p.W) continue;
              *reinterpret_cast<f32x4*>(p.out + ((long)(n * p.H + a) * p.W + b) * p.Cout + co) = Y[u][v];
              if constexpr (STATS) {
                s0 += Y[u][v];
                s1 += Y[u][v] * Y[u][v];
              }
            }
        }
      }
      if constexpr (STATS) {
        if (!BWD || p.stats) {
          // over the wave's 16 tiles (lanes l15 of each 16-lane row): four DPP row rotations, every lane of the row ends with the row's sum
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            s0[e] = rs_row16_sum(s0[e]);
            s1[e] = rs_row16_sum(s1[e]);
          }
          st_buf = seq & 1;
          st_row = mblk;
          st_col = nblk * BN;
          if (l15 == 0) {
            float* x = xst + st_buf * XST + tg * BN + 16 * cg + 4 * pc;
            *reinterpret_cast<f32x4*>(x) = s0;
            *reinterpret_cast<f32x4*>(x + TG * BN) = s1;
            rs_lds_writes_done();  // (read behind the next barrier, which hipcc emits bare: common.h)
          }
        }
      }
    } else {
      // ---- self.final on the block's 32 channels: this lane's 4 couts -> the wave's 16 (lanes l15 + 16 pc) -> both cout
      //      groups (waves tg and tg + TG, through LDS) + bias; then one pixel per lane of the cg = 0 waves (pixel pc of tile
      //      l15).  The order of the sums is fixed by the layout, not by the batch.
      float part[4][kHeadMaxC];
#pragma unroll
      for (int c = 0; c < kHeadMaxC; ++c) {
        if (c < p.hC) {
          const f32x4 wv = *reinterpret_cast<const f32x4*>(hws + c * 32 + 16 * cg + 4 * pc);
#pragma unroll
          for (int px = 0; px < 4; ++px) {
            const f32x4 y = Y[px >> 1][px & 1];
            float v = fmaf(y[3], wv[3], fmaf(y[2], wv[2], fmaf(y[1], wv[1], y[0] * wv[0])));
            v = rs_xor32_sum(rs_xor16_sum(v));  // (the 16 couts: lanes l15 + 16 pc, on the VALU -- not __shfl_xor, see below)
            part[px][c] = v;
          }
        } else {
#pragma unroll
          for (int px = 0; px < 4; ++px) part[px][c] = 0.f;
        }
      }
      // the cg = 1 half goes to LDS; the cg = 0 lanes keep their pixel's half and finish it behind the next barrier (the first
      // chunk of the block's next item, or the one after the loop): no barrier of its own, and the softmax + stores run
      // beside the other waves' MFMAs.
      // Round 6 (profiles/r06/head_race.txt): as round 5 shipped it this epilogue was WRONG whenever an LDS-DMA kernel of another
      // stream shared the CUs -- up to 0.32 off in ~2 500 probabilities per launch, never alone (tests/test_gpu_race_screen.py
      // ::test_wino33_fused_head_fp32_beside_an_lds_user was red in 20 of 20 rounds).  Two changes, both needed for a clean screen:
      // (1) the writer waits for its own LDS stores (hipcc emits the next __syncthreads() as a bare s_barrier: common.h), (2) the
      // sum over the wave's 16 couts runs on the VALU (v_permlane16/32_swap) instead of two ds_bpermute round trips per value:
      // with (1) alone the sums of the lanes that receive lanes 48-63 were still stale in most launches of that build.  The
      // hardware mechanism behind (2) was not isolated (two probes of candidate hazards, scripts/probes/probe_bpermute_dma.hip and
      // probe_pk_ds_hazard.hip, are negative; a later build with __shfl_xor behind a run-time switch passed too): what is pinned
      // is the screen, which any later change of this epilogue must keep green.
      hbuf = seq & 1;
      if (cg == 1 && pc == 0) {
        float* xrow = xch + hbuf * XCH + (16 * tg + l15) * XROW;
#pragma unroll
        for (int px = 0; px < 4; ++px) {
          *reinterpret_cast<f32x4*>(xrow + px * kHeadMaxC) = f32x4{part[px][0], part[px][1], part[px][2], part[px][3]};
          *reinterpret_cast<f32x4*>(xrow + px * kHeadMaxC + 4) = f32x4{part[px][4], part[px][5], part[px][6], part[px][7]};
        }
        rs_lds_writes_done();  // (the cg = 0 waves read it right behind the next barrier, which hipcc emits bare: common.h)
      }
#pragma unroll
      for (int c = 0; c < kHeadMaxC; ++c) hmine[c] = pc == 0 ? part[0][c] : (pc == 1 ? part[1][c] : (pc == 2 ? part[2][c] : part[3][c]));
      const int a = a0 + (pc >> 1), b = b0 + (pc & 1);
      hpix = (live && a < p.H && b < p.W) ? (long)(n * p.H + a) * p.W + b : -1;
    }
  }
  if constexpr (HEAD) {
    __syncthreads();
    if (nitems > 0) finish_head();
  }
  if constexpr (STATS) {
    __syncthreads();
    if (nitems > 0) finish_stats();
  }
  rb_dma_wait();  // (the re-issued pieces of the last chunk: landed before this block's LDS is handed to the next one)
}

// U = G g G^T per (cout, cin): KRSC [Cout][3][3][Cin] -> [16][Cout][Cin]
__global__ void pack_wino33_weight_kernel(const float* __restrict__ w, float* __restrict__ u, int Cout, int Cin, long total) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;  // over [Cout][Cin]
  if (i >= total) return;
  const int ci = (int)(i % Cin), co = (int)(i / Cin);
  const float* g = w + (long)co * 9 * Cin + ci;
  float gg[3][3];
#pragma unroll
  for (int r = 0; r < 3; ++r)
#pragma unroll
    for (int c = 0; c < 3; ++c) gg[r][c] = g[(long)(r * 3 + c) * Cin];
  float t[4][3];  // G g
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    t[0][c] = gg[0][c];
    t[1][c] = 0.5f * ((gg[0][c] + gg[1][c]) + gg[2][c]);
    t[2][c] = 0.5f * ((gg[0][c] - gg[1][c]) + gg[2][c]);
    t[3][c] = gg[2][c];
  }
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const float v[4] = {t[r][0], 0.5f * ((t[r][0] + t[r][1]) + t[r][2]), 0.5f * ((t[r][0] - t[r][1]) + t[r][2]), t[r][2]};
#pragma unroll
    for (int c = 0; c < 4; ++c) u[((long)(r * 4 + c) * Cout + co) * Cin + ci] = v[c];
  }
}

int w33_cus() {
  static const int n = [] {
    int dev = 0, cus = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0)
      cus = 256;
    return cus;
  }();
  return n;
}

// 0: cannot run; 1: can and should; (there is no "can but should not": the form needs >= 8 tiles per image side to run at all)
int w33_plan(const rs_conv_desc* d, int* cgroups) {
  if (!d || d->N <= 0 || d->Hs <= 0 || d->Ws <= 0 || d->C1 < 32 || (d->C1 % 16) || d->C2 != 0 || d->Cout <= 0 || (d->Cout % 16)) return 0;
  if (!(d->ups == 0 && d->kh == 3 && d->kw == 3 && d->stride == 1 && d->pad == 1 && d->Ho == d->Hs && d->Wo == d->Ws && !d->stem)) return 0;
  if ((d->Hs + 1) / 2 < kPB || (d->Ws + 1) / 2 < kPB) return 0;
  *cgroups = (d->Cout % 32 == 0) ? 2 : 1;
  const int sb = 16 * (8 / *cgroups) / (kPB * kPB);
  if ((long)(sb + 1) * d->Hs * d->Ws * d->C1 * 4 >= (1L << 31)) return 0;
  if ((long)16 * d->Cout * d->C1 * 4 >= (1L << 31)) return 0;
  if ((long)d->N * d->Ho * d->Wo * d->Cout >= (1L << 62)) return 0;
  return 1;
}

}  // namespace

extern "C" int rs_conv2d_wino33_ok(const rs_conv_desc* d) {
  int cgn;
  return w33_plan(d, &cgn);
}

extern "C" const char* rs_conv2d_wino33_name(const rs_conv_desc* d) {
  int cgn;
  if (!w33_plan(d, &cgn)) return "";
  return cgn == 2 ? "conv_wino_f32<3x3,p8,64x32>" : "conv_wino_f32<3x3,p8,128x16>";
}

// dec5 + self.final: the Winograd form must run the layer as ONE cout block of 32 (dec5: num_filters = 32, unet.py:104-108)
extern "C" int rs_conv2d_wino33_head_ok(const rs_conv_desc* d, int C) {
  int cgn;
  return w33_plan(d, &cgn) && cgn == 2 && d->Cout == 32 && C >= 1 && C <= kHeadMaxC;
}

extern "C" const char* rs_conv2d_wino33_head_name(void) { return "conv_wino_f32<3x3+final,p8,64x32>"; }

extern "C" int rs_pack_wino33_weight(const float* w_krsc, float* u, int Cout, int Cin, rs_stream_t stream) {
  if (!w_krsc || !u || Cout <= 0 || Cin <= 0) return RS_EINVAL;
  const long total = (long)Cout * Cin;
  pack_wino33_weight_kernel<<<rs_cdiv(total, 256), 256, 0, (hipStream_t)stream>>>(w_krsc, u, Cout, Cin, total);
  return RS_LAUNCH_RESULT();
}

extern "C" int rs_conv2d_fwd_wino33(const rs_conv_desc* d, const float* src, const float* u, const float* scale, const float* shift,
                                    float* out, rs_stream_t stream) {
  int cgn;
  if (!w33_plan(d, &cgn) || !src || !u || !out) return RS_EINVAL;
  Wino33Args a;
  a.src = src;
  a.u = u;
  a.scale = scale;
  a.shift = shift;
  a.out = out;
  a.N = d->N;
  a.H = d->Hs;
  a.W = d->Ws;
  a.Cin = d->C1;
  a.Cout = d->Cout;
  a.BBY = rs_cdiv((d->Hs + 1) / 2, kPB);
  a.BBX = rs_cdiv((d->Ws + 1) / 2, kPB);
  a.nsub = d->N * a.BBY * a.BBX;
  a.ncb = d->Cout / (16 * cgn);
  a.relu = d->relu;
  a.hw = a.hb = nullptr;
  a.hanchors = nullptr;
  a.hout = nullptr;
  a.hq = nullptr;
  a.hC = a.hmode = a.hov = 0;
  a.stats = nullptr;
  a.mask = a.bn_y = a.bn_mean = a.bn_invstd = nullptr;
  a.mask_bits = nullptr;
  const int sb = 16 * (8 / cgn) / (kPB * kPB);
  const long items = (long)rs_cdiv(a.nsub, sb) * a.ncb;
  if (items >= (1L << 31)) return RS_EINVAL;
  const int grid = (int)(items < w33_cus() ? items : w33_cus());
  hipStream_t s = (hipStream_t)stream;
  if (cgn == 2) conv_wino33_f32_kernel<4, 2><<<grid, 512, 0, s>>>(a);
  else conv_wino33_f32_kernel<8, 1><<<grid, 512, 0, s>>>(a);
  return RS_LAUNCH_RESULT();
}

// rs_conv2d_fwd_wino33 on a 32-cout layer + `self.final` (1x1, C classes, bias) + what rs_final_conv1x1_dt /
// rs_final_conv1x1_quantize_dt / rs_final_conv1x1_argmax_dt do with the logits, in one launch: mode 0 logits / 1 softmax ->
// `out` fp32 NCHW [N][C][H][W]; 2 -> `qout` = quantised probabilities of the un-buffered crop (`anchors`, `overlap`);
// 3 -> `qout` = class indices [N][H][W].  The layer's own output never reaches memory.
extern "C" int rs_conv2d_fwd_wino33_head(const rs_conv_desc* d, const float* src, const float* u, const float* scale, const float* shift,
                                         const float* final_w, const float* final_b, int C, int mode, const double* anchors,
                                         int overlap, float* out, uint8_t* qout, rs_stream_t stream) {
  if (!rs_conv2d_wino33_head_ok(d, C) || !src || !u || !final_w || mode < 0 || mode > 3) return RS_EINVAL;
  if (mode <= 1 ? !out : !qout) return RS_EINVAL;
  if (mode == 2 && (!anchors || C < 2 || overlap < 0 || 2 * overlap >= d->Hs || 2 * overlap >= d->Ws)) return RS_EINVAL;
  if (mode == 3 && C < 2) return RS_EINVAL;
  Wino33Args a;
  a.src = src;
  a.u = u;
  a.scale = scale;
  a.shift = shift;
  a.out = nullptr;
  a.N = d->N;
  a.H = d->Hs;
  a.W = d->Ws;
  a.Cin = d->C1;
  a.Cout = d->Cout;
  a.BBY = rs_cdiv((d->Hs + 1) / 2, kPB);
  a.BBX = rs_cdiv((d->Ws + 1) / 2, kPB);
  a.nsub = d->N * a.BBY * a.BBX;
  a.ncb = 1;
  a.relu = d->relu;
  a.hw = final_w;
  a.hb = final_b;
  a.hanchors = anchors;
  a.hout = out;
  a.hq = qout;
  a.hC = C;
  a.hmode = mode;
  a.hov = overlap;
  a.stats = nullptr;
  a.mask = a.bn_y = a.bn_mean = a.bn_invstd = nullptr;
  a.mask_bits = nullptr;
  const long items = a.nsub;
  if (items >= (1L << 31)) return RS_EINVAL;
  const int grid = (int)(items < w33_cus() ? items : w33_cus());
  hipStream_t s = (hipStream_t)stream;
  if (mode <= 1) conv_wino33_f32_kernel<4, 2, 1><<<grid, 512, 0, s>>>(a);
  else if (mode == 2) conv_wino33_f32_kernel<4, 2, 2><<<grid, 512, 0, s>>>(a);
  else conv_wino33_f32_kernel<4, 2, 3><<<grid, 512, 0, s>>>(a);
  return RS_LAUNCH_RESULT();
}

// Train-mode forward of the same layers (round 6): RAW output + the partial sums of BatchNorm's statistics, one row of `stats`
// [rows][2][Cout] per block of 64 (Cout % 32 == 0) / 128 tiles -- rs_conv2d_wino33_stats_rows(d) rows, every entry written.
extern "C" long rs_conv2d_wino33_stats_rows(const rs_conv_desc* d) {
  int cgn;
  if (!w33_plan(d, &cgn)) return RS_EINVAL;
  const int sb = 16 * (8 / cgn) / (kPB * kPB);
  return rs_cdiv((long)d->N * rs_cdiv((d->Hs + 1) / 2, kPB) * rs_cdiv((d->Ws + 1) / 2, kPB), sb);
}

extern "C" int rs_conv2d_fwd_wino33_stats(const rs_conv_desc* d, const float* src, const float* u, float* out, float* stats,
                                          rs_stream_t stream) {
  int cgn;
  if (!w33_plan(d, &cgn) || !src || !u || !out || !stats) return RS_EINVAL;
  Wino33Args a;
  a.src = src;
  a.u = u;
  a.scale = a.shift = nullptr;
  a.out = out;
  a.N = d->N;
  a.H = d->Hs;
  a.W = d->Ws;
  a.Cin = d->C1;
  a.Cout = d->Cout;
  a.BBY = rs_cdiv((d->Hs + 1) / 2, kPB);
  a.BBX = rs_cdiv((d->Ws + 1) / 2, kPB);
  a.nsub = d->N * a.BBY * a.BBX;
  a.ncb = d->Cout / (16 * cgn);
  a.relu = 0;
  a.hw = a.hb = nullptr;
  a.hanchors = nullptr;
  a.hout = nullptr;
  a.hq = nullptr;
  a.hC = a.hmode = a.hov = 0;
  a.stats = stats;
  a.mask = a.bn_y = a.bn_mean = a.bn_invstd = nullptr;
  a.mask_bits = nullptr;
  const int sb = 16 * (8 / cgn) / (kPB * kPB);
  const long items = (long)rs_cdiv(a.nsub, sb) * a.ncb;
  if (items >= (1L << 31)) return RS_EINVAL;
  const int grid = (int)(items < w33_cus() ? items : w33_cus());
  hipStream_t s = (hipStream_t)stream;
  if (cgn == 2) conv_wino33_f32_kernel<4, 2, 4><<<grid, 512, 0, s>>>(a);
  else conv_wino33_f32_kernel<8, 1, 4><<<grid, 512, 0, s>>>(a);
  return RS_LAUNCH_RESULT();
}

// The data gradient of such a layer (round 6): rs_conv2d_fwd_wino33 over dy with u = rs_pack_wino33_weight of the data-gradient filters
// (rs_pack_dgrad_weight: flipped taps, [Cin][3][3][Cout]); `d` describes THAT convolution (C1 = dy's channels, Cout = the gradient's).
// The gradient arrives at a ReLU output: `mask` (the forward activation, its sign decides) or `mask_bits` (rs_bn_apply_bits_dt's, one
// bit per element; Cout % 8 == 0) or neither; with `stats` [rs_conv2d_wino33_stats_rows(d)][2][Cout] also at a BatchNorm output --
// the per-block partial sums (sum g, sum g * (bn_y - bn_mean) * bn_invstd) of the masked gradient, rs_bn_bwd_from_partials_dt's input
// (what rs_conv2d_dgrad_bnstats[_bits]_dt computes with nine taps: autograd of torchvision Bottleneck.conv2 / ConvRelu under
// tools/train.py:186).
extern "C" int rs_conv2d_dgrad_wino33(const rs_conv_desc* d, const float* dy, const float* u, const float* mask, const uint8_t* mask_bits,
                                      const float* bn_y, const float* bn_mean, const float* bn_invstd, float* out, float* stats,
                                      rs_stream_t stream) {
  int cgn;
  if (!w33_plan(d, &cgn) || cgn != 2 || !dy || !u || !out) return RS_EINVAL;  // (Cout % 32 == 0: every layer this is for)
  if (mask && mask_bits) return RS_EINVAL;
  if (mask_bits && (d->Cout % 8) != 0) return RS_EINVAL;
  if (stats ? (!bn_y || !bn_mean || !bn_invstd) : (bn_y || bn_mean || bn_invstd)) return RS_EINVAL;
  Wino33Args a;
  a.src = dy;
  a.u = u;
  a.scale = a.shift = nullptr;
  a.out = out;
  a.N = d->N;
  a.H = d->Hs;
  a.W = d->Ws;
  a.Cin = d->C1;
  a.Cout = d->Cout;
  a.BBY = rs_cdiv((d->Hs + 1) / 2, kPB);
  a.BBX = rs_cdiv((d->Ws + 1) / 2, kPB);
  a.nsub = d->N * a.BBY * a.BBX;
  a.ncb = d->Cout / (16 * cgn);
  a.relu = 0;
  a.hw = a.hb = nullptr;
  a.hanchors = nullptr;
  a.hout = nullptr;
  a.hq = nullptr;
  a.hC = a.hmode = a.hov = 0;
  a.stats = stats;
  a.mask = mask;
  a.mask_bits = mask_bits;
  a.bn_y = bn_y;
  a.bn_mean = bn_mean;
  a.bn_invstd = bn_invstd;
  const int sb = 16 * (8 / cgn) / (kPB * kPB);
  const long items = (long)rs_cdiv(a.nsub, sb) * a.ncb;
  if (items >= (1L << 31)) return RS_EINVAL;
  const int grid = (int)(items < w33_cus() ? items : w33_cus());
  hipStream_t s = (hipStream_t)stream;
  conv_wino33_f32_kernel<4, 2, 5><<<grid, 512, 0, s>>>(a);
  return RS_LAUNCH_RESULT();
}

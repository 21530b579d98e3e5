// conv_wgrad_f32_dma.hip -- the fp32 weight gradient (v_mfma_f32_32x32x2_f32, exact fp32) with both operands copied
// HBM -> LDS by LDS-DMA, as they lie in memory.  Round 5; the filter gradients autograd synthesises for every nn.Conv2d of
// UNet.forward under the reference's fp32 training (robosat/tools/train.py:180-188), same blocks and same partial-tile format
// as conv_wgrad.hip's register-staged kernel (which stays for the packed 7x7 stem; within a 32-pixel chunk the two pair the
// pixels of a k-step differently, so they agree to fp32 summation-order noise, not bit for bit).
//
//     dW[co][tap][ci] = sum over pixels m of  dy[m][co] * in[gather(m, tap)][ci]
//
// Why.  conv_wgrad.hip stages a chunk through VGPRs (16-byte global loads whose addresses cost two mul-hi divisions each,
// a 4x4 register transpose, ds_write_b128 into channel-major rows) so that the MFMA operands come back as ds_read_b128: 71 TF
// over the fp32 train step's 60 launches = 45 % of the fp32 MFMA peak (profiles/r05/train_fp32_per_layer.txt).  But the fp32
// MFMA takes ONE dword per lane per operand -- A[i = lane & 31][k = lane >> 5] -- and both tensors are pixel-major in HBM
// ([pixel][channel]): lane (i, k) simply reads channel i of pixel row k.  So nothing has to be transposed at all:
//   HBM -> LDS   buffer_load_dwordx4 ... lds: a wave instruction moves 1 KiB = whole [pixel] rows of the tile (2 / 4 / 8 rows
//                for 128 / 64 / 32 channels), each lane with its own global offset (gather table lookup + one multiply-add),
//                out-of-image / tail rows at offset -1 (the hardware writes zeros).  No VGPR round trip, no ds_write.
//   LDS image    [32 pixels][BMo or BNo channels] fp32, unpadded, unswizzled: a fragment read is 32 consecutive dwords per
//                half wave (lanes 0-31: pixel row 2s, lanes 32-63: row 2s + 1) -- conflict free as it stands.
//   MFMA         per k-step (2 pixels) TM + TN ds_read_b32 feed TM x TN v_mfma_f32_32x32x2_f32 (64 cycles each): the LDS
//                instruction count that sank round 1's first pixel-major version (register-staged, 89 TF) is small change
//                next to 256 MFMA cycles per k-step once the staging instructions are gone.
//   pipeline     two LDS buffers, one barrier per 32-pixel chunk (16 k-steps = 64 MFMAs per wave on the 128 x 128 tile), the
//                DMA pieces of chunk c + 1 issued between the MFMAs of chunk c, gather tables two chunks ahead.
// PHASE (DecoderBlock): one of the 16 (output parity, source offset) reductions over the SOURCE pixels; dz rows are gathered
// too (second table).  Partial tiles [split][Cout][K] as before; reduce.hip / the phase combine are unchanged.
#include <type_traits>

#include "conv_wgrad_f32.h"

namespace {

__device__ __forceinline__ __amdgpu_buffer_rsrc_t wd_make_rsrc(const void* base, long bytes) {
  const unsigned int n = bytes > 0xFFFFFFFEL ? 0xFFFFFFFEu : (unsigned int)(bytes < 0 ? 0 : bytes);
  // (the inputs are block-uniform, but 64-bit multiplies and divisions run on the VALU: readfirstlane makes the uniformity
  // provable, the LDS-DMA asm needs its SRSRC in SGPRs -- see rb_make_rsrc in conv_igemm_dma_kernel.h)
  const unsigned long b = (unsigned long)base;
  const unsigned int lo = __builtin_amdgcn_readfirstlane((unsigned int)b), hi = __builtin_amdgcn_readfirstlane((unsigned int)(b >> 32));
  const unsigned int nn = __builtin_amdgcn_readfirstlane(n);
  return __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<void*>(((unsigned long)hi << 32) | lo), 0, (int)nn, 0x00020000);
}

// One LDS-DMA wave instruction: lane l's 16 bytes at buffer offset `voff` land at LDS byte `lds_dst` + 16*l (lds_dst
// wave-uniform, in M0); an out-of-range offset writes zeros.  Inline asm for the reason given at wb_dma16 (conv_wgrad_bf16.hip):
// the compiler must not drain the queue before the fragment reads of the OTHER buffer; the kernel waits itself.
__device__ __forceinline__ void wd_dma16(__amdgpu_buffer_rsrc_t r, unsigned int lds_dst, int voff) {
  asm volatile(
      "s_mov_b32 m0, %1\n\t"
      "s_nop 0\n\t"
      "buffer_load_dwordx4 %0, %2, 0 offen lds"
      :
      : "v"(voff), "s"(lds_dst), "s"(r)
      : "memory", "m0");
}
__device__ __forceinline__ void wd_dma_wait() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
__device__ __forceinline__ unsigned int wd_lds_addr(const void* p) {
  return (unsigned int)(unsigned long)(__attribute__((address_space(3))) const void*)p;
}

// PK = pixels per pipeline chunk (a plan chunk is 32 pixels).  Instantiated with 32: 16-pixel chunks -- half the LDS, four
// blocks per CU instead of two -- were measured on every layer of the fp32 bs-8 step and change nothing (8.79 vs 8.74 ms
// of weight gradients, profiles/r05/wgrad_f32_dma.txt): occupancy is not what bounds this kernel.
template <int BMo, int BNo, int WGM, int WGN, bool PHASE, int PK>
__global__ __launch_bounds__(64 * WGM * WGN, PK == 16 ? 4 : 2) void conv_wgrad_f32_dma(const WgradArgs p) {
  constexpr int NS = PK / 2;             // MFMA k-steps per chunk
  constexpr int NW = WGM * WGN;          // waves
  constexpr int WM = BMo / WGM, WN = BNo / WGN;
  constexpr int TM = WM / 32, TN = WN / 32;
  constexpr int ROWA = BMo * 4, ROWB_ = BNo * 4;        // bytes per LDS row (one pixel)
  constexpr int PPA = ROWA / 16, PPB = ROWB_ / 16;      // 16-byte pieces per row
  constexpr int RIA = 1024 / ROWA, RIB = 1024 / ROWB_;  // rows per DMA instruction
  constexpr int IA = PK / RIA, IB = PK / RIB;           // DMA instructions per chunk
  constexpr int NI = (IA + IB) / NW;                    // per wave
  constexpr int ABYTES = PK * ROWA, BBYTES = PK * ROWB_;
  constexpr int BUF = ABYTES + BBYTES;
  static_assert(TM >= 1 && TN >= 1 && (IA % NW) == 0 && (IB % NW) == 0 && NI >= 1, "bad tile");
  static_assert(2 * BUF + 4 * PK * 4 <= (PK == 16 ? 40 : 80) * 1024, "four / two blocks per CU");
  static_assert(PK == 16 || PK == 32, "a plan chunk is 32 pixels");

  __shared__ __attribute__((aligned(16))) unsigned char smem[2 * BUF + (PHASE ? 4 : 2) * PK * 4];
  int* tabs = reinterpret_cast<int*>(smem + 2 * BUF);  // [2][PK] input-row gather: source pixel (relative to image n_first) or -1
  int* taba = tabs + 2 * PK;                            // [2][PK] dz-row gather (PHASE only)

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

  const float* src = p.src1;
  int Cs = p.C1, cs = ci0;
  if (ci0 >= p.C1) {  // (a tile never straddles the two concat sources: BNo divides both)
    src = p.src2;
    Cs = p.C2;
    cs = ci0 - p.C1;
  }

  const int chunk0 = split * p.chunks_per_split * (32 / PK);
  const int total_chunks = (p.M + PK - 1) / PK;
  int chunk1 = chunk0 + p.chunks_per_split * (32 / PK);
  if (chunk1 > total_chunks) chunk1 = total_chunks;
  const int Wd = PHASE ? p.Ws : p.Wo;                   // the pixel grid the reduction index m enumerates
  const int HoWo = PHASE ? p.Hs * p.Ws : p.Ho * p.Wo;   // (div_howo / div_wo are prepared for that grid)

  const int m_first = chunk0 * PK;
  const int n_first = (int)rs_div((unsigned)m_first, p.div_howo);
  const long img = (long)p.Hs * p.Ws * Cs;
  const long dimg = (long)p.Ho * p.Wo * p.Cout;
  const __amdgpu_buffer_rsrc_t rsrc_dy =
      PHASE ? wd_make_rsrc(p.dy + n_first * dimg, (long)(p.N - n_first) * dimg * 4)
            : wd_make_rsrc(p.dy + (long)m_first * p.Cout, ((long)p.M - m_first) * p.Cout * 4);
  const __amdgpu_buffer_rsrc_t rsrc_x = wd_make_rsrc(src + n_first * img, (long)(p.N - n_first) * img * 4);
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
    }
  };

  // ---- DMA roles: instruction ii = wave + NW*j (j < NI); ii < IA copies dy rows RIA*ii.., else input rows RIB*(ii-IA)...
  //      Lane l: row l / PP of the instruction, 16-byte piece l % PP of that row.  The LDS image is lane-linear = row-major.
  const int ra_a = lane / PPA, pp_a = lane % PPA;
  const int ra_b = lane / PPB, pp_b = lane % PPB;
  const int cola = (co0 + pp_a * 4) * 4;  // byte offset of the piece inside a dy row
  const int colb = (cs + pp_b * 4) * 4;
  const int cout4 = p.Cout * 4, cs4 = Cs * 4;

  const unsigned int lds0 = __builtin_amdgcn_readfirstlane(wd_lds_addr(smem));
  int voff[NI];
  unsigned int fL = lds0;
  auto prepare_dma = [&](int chunk, int buf, int which) __attribute__((always_inline)) {
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
          voff[j] = pix[j] >= 0 ? pix[j] * cout4 + cola : -1;
        } else {
          const int m = chunk * PK + RIA * ii + ra_a;
          voff[j] = (m - m_first) * cout4 + cola;  // rows >= M lie past the descriptor: zeros
        }
      } else {
        voff[j] = pix[j] >= 0 ? pix[j] * cs4 + colb : -1;
      }
    }
  };
  auto issue_piece = [&](int j) __attribute__((always_inline)) {
    const int ii = wave + NW * j;
    if (NW * j < IA) wd_dma16(rsrc_dy, fL + ii * 1024, voff[j]);
    else wd_dma16(rsrc_x, fL + ABYTES + (ii - IA) * 1024, voff[j]);
  };

  f32x16 acc[TM][TN];
#pragma unroll
  for (int a = 0; a < TM; ++a)
#pragma unroll
    for (int b = 0; b < TN; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

  // fragment addressing: k-step s, lane (i = lane & 31, k = lane >> 5): pixel row 2s + k, channel (sub-tile base + i)
  const int arow = (lane >> 5) * ROWA + (wm * WM + (lane & 31)) * 4;
  const int brow = ABYTES + (lane >> 5) * ROWB_ + (wn * WN + (lane & 31)) * 4;
  auto read_frag = [&](const unsigned char* L, int s, float (&a)[TM], float (&b)[TN]) __attribute__((always_inline)) {
#pragma unroll
    for (int tm = 0; tm < TM; ++tm) a[tm] = *reinterpret_cast<const float*>(L + arow + 2 * s * ROWA + tm * 128);
#pragma unroll
    for (int tn = 0; tn < TN; ++tn) b[tn] = *reinterpret_cast<const float*>(L + brow + 2 * s * ROWB_ + tn * 128);
  };

  constexpr int NMMA = NS * TM * TN;
  constexpr int PSTEP = NMMA / (2 * NI) >= 1 ? NMMA / (2 * NI) : 1;  // front-loaded: the chunk's tail covers the latency
  constexpr int PIN = (NMMA + PSTEP - 1) / PSTEP < NI ? (NMMA + PSTEP - 1) / PSTEP : NI;
  auto chunk_mma = [&](const unsigned char* L, auto fetch_tag) __attribute__((always_inline)) {
    constexpr bool FETCH = decltype(fetch_tag)::value;
    float fa[2][TM], fb[2][TN];
    read_frag(L, 0, fa[0], fb[0]);
#pragma unroll
    for (int s = 0; s < NS; ++s) {
      if (s + 1 < NS) read_frag(L, s + 1, fa[(s + 1) & 1], fb[(s + 1) & 1]);
#pragma unroll
      for (int tm = 0; tm < TM; ++tm)
#pragma unroll
        for (int tn = 0; tn < TN; ++tn) {
          const int q = (s * TM + tm) * TN + tn;  // MFMA index within the chunk (compile-time after unrolling)
          if (FETCH && q % PSTEP == 0 && q / PSTEP < PIN) issue_piece(q / PSTEP);
          acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[s & 1][tm], fb[s & 1][tn], acc[tm][tn], 0, 0, 0);
        }
    }
    if (FETCH) {
#pragma unroll
      for (int q = PIN; q < NI; ++q) issue_piece(q);
    }
  };
  if (chunk0 < chunk1) {
    fill_table(chunk0, 0);
    fill_table(chunk0 + 1, 1);
    __syncthreads();
    prepare_dma(chunk0, 0, 0);
#pragma unroll
    for (int q = 0; q < NI; ++q) issue_piece(q);
    wd_dma_wait();
    __syncthreads();
    int c = chunk0;
    for (; c + 1 < chunk1; ++c) {  // steady state: chunk c+1 streams into the other buffer between chunk c's MFMAs
      const int it = c - chunk0;
      // (the other buffer's last readers passed the barrier that ended iteration it-1)
      prepare_dma(c + 1, (it + 1) & 1, (it + 1) & 1);
      chunk_mma(smem + (it & 1) * BUF, std::true_type());
      fill_table(c + 2, it & 1);
      wd_dma_wait();  // this wave's share of chunk c+1 has landed; the barrier publishes everybody's
      __syncthreads();
    }
    chunk_mma(smem + ((c - chunk0) & 1) * BUF, std::false_type());
  }

  // D[i][j]: i = cout (tile-local) = (r&3) + 8*(r>>2) + 4*(lane>>5), j = cin (tile-local) = lane&31
  float* out = p.out + (long)split * p.Cout * p.K;
  const int kbase = tap * (p.C1 + p.C2) + ci0;
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

template <bool PHASE>
int launch(int variant, int grid, hipStream_t s, const WgradArgs& a) {
  switch (variant) {
    case V128x128: conv_wgrad_f32_dma<128, 128, 2, 2, PHASE, 32><<<grid, 256, 0, s>>>(a); break;
    case V128x64: conv_wgrad_f32_dma<128, 64, 2, 2, PHASE, 32><<<grid, 256, 0, s>>>(a); break;
    case V64x128: conv_wgrad_f32_dma<64, 128, 2, 2, PHASE, 32><<<grid, 256, 0, s>>>(a); break;
    case V64x64: conv_wgrad_f32_dma<64, 64, 2, 2, PHASE, 32><<<grid, 256, 0, s>>>(a); break;
    case V32x128: conv_wgrad_f32_dma<32, 128, 1, 4, PHASE, 32><<<grid, 256, 0, s>>>(a); break;
    case V32x32: conv_wgrad_f32_dma<32, 32, 1, 1, PHASE, 32><<<grid, 64, 0, s>>>(a); break;
    default: return RS_EINVAL;
  }
  return RS_LAUNCH_RESULT();
}

}  // namespace

int rs_wgrad_f32_dma_launch(int variant, bool phase, int grid, hipStream_t s, const WgradArgs& a) {
  return phase ? launch<true>(variant, grid, s, a) : launch<false>(variant, grid, s, a);
}

// Weight-gradient convolution for gfx950 on v_mfma_f32_32x32x2_f32 (exact fp32).
//
// The filter gradients that autograd synthesises for every nn.Conv2d of UNet.forward when the reference calls
// loss.backward() (robosat/tools/train.py:186):
//
//     dW[co][ky][kx][ci] = sum over output pixels m=(n,oy,ox) of  dy[m][co] * in[n][oy*stride-pad+ky][ox*stride-pad+kx][ci]
//
// GEMM view: rows = Cout, cols = one filter tap x a slice of Cin, REDUCTION over the M = N*Ho*Wo output pixels.
// `in` is read through the same gather as the forward pass (nearest-x2 upsample, two-source concat, stem packing),
// so the upsampled / concatenated tensors are not materialised in the backward pass either.
//
//   block  = tile BMo couts x BNo cins of one tap, over a contiguous range of pixels (split-P), walked in chunks of
//            32 pixels; LDS double buffered, next chunk prefetched into registers during the MFMAs.
//   loads  = buffer_load_dwordx4 through SRSRC descriptors (hardware bounds check => zeros for padding / tail
//            pixels, offset = -1): branch-free, so the address math + loads of chunk k+1 interleave with the MFMAs
//            of chunk k; pixel -> (n, oy, ox) uses mul-hi division by constants prepared on the host.
//   LDS    = channel-major [channel'][32 pixels (+4 pad)]: a thread fetches a 4-pixel x 4-channel block (four
//            16-byte loads, each wave instruction = whole 512-B rows), transposes it in registers for free and
//            writes four ds_write_b128 (4 consecutive pixels of one channel).  LDS row R = e*Q + c4 holds channel
//            4*c4 + e (Q = channels/4), so consecutive lanes write consecutive rows (stride 36 floats: conflict
//            free) and the MFMA operands are read exactly like the forward kernel: one ds_read_b128 = 4 pixels of
//            the reduction per lane, A[i = co'][k = pixel], B[k = pixel][j = ci'].  The channel permutation is undone
//            by the epilogue's addressing.  (The first version kept [pixel][channel] tiles and fed the MFMAs with
//            ds_read_b32: 4x the LDS instructions, 89 TF on dec3 against 134 TF for the forward kernel.)
//   split-P: partial tiles go to a workspace [split][Cout][K] and are summed by a second (streaming) kernel:
//            deterministic, no atomics.
//   PHASE  : DecoderBlock (3x3 / pad 1 behind the nearest-x2 upsample, robosat/unet.py:63-73).  Instead of nine taps over
//            the UPSAMPLED pixels the block reduces one of 16 (output parity, source offset) combinations over the SOURCE
//            pixels, G[(py,px),(r,s)][co][ci] = sum_{n,a,b} dz[n][2a+py][2b+px][co] * src[n][a-(1-py)+r][b-(1-px)+s][ci]
//            -- 16/36 of the multiply-adds -- and combine_phase_wgrad_f32_kernel adds the four G's that make up each
//            filter tap (the fp32 twin of conv_wgrad_bf16.hip's phase form; round 4).
#include "conv_wgrad_f32.h"

namespace {

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ __amdgpu_buffer_rsrc_t wg_make_rsrc(const float* base, long bytes) {
  const unsigned int n = bytes > 0xFFFFFFFEL ? 0xFFFFFFFEu : (unsigned int)(bytes < 0 ? 0 : bytes);
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(base), 0, (int)n, 0x00020000);
}

__device__ __forceinline__ f32x4 wg_buffer_load4(__amdgpu_buffer_rsrc_t r, int byte_off) {
  return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, byte_off, 0, 0));
}

// FORM: 0 plain, 1 the packed 7x7 stem, 2 phase form of DecoderBlock
template <int BMo, int BNo, int WGM, int WGN, int FORM>
__global__ __launch_bounds__(64 * WGM * WGN) void conv_wgrad_f32(const WgradArgs p) {
  constexpr int STEM = FORM == 1 ? 1 : 0;
  constexpr bool PHASE = FORM == 2;
  constexpr int NT = 64 * WGM * WGN;
  constexpr int WM = BMo / WGM, WN = BNo / WGN;
  constexpr int TM = WM / 32, TN = WN / 32;
  constexpr int AQ = BMo / 4, BQ = BNo / 4;  // channel quads per tile
  constexpr int TA = 2 * BMo, TB = 2 * BNo;  // threads staging the A / B tile (one 4-pixel x 4-channel block each)
  constexpr int LDT = 36;                    // padded LDS row (floats): 32 pixels + 4
  constexpr int BUF = (BMo + BNo) * LDT;
  static_assert(TM >= 1 && TN >= 1 && TA <= NT && TB <= NT && (TA % 64) == 0 && (TB % 64) == 0, "bad tile");

  __shared__ __attribute__((aligned(16))) float lds[2 * BUF];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int wm = wave / WGN, wn = wave % WGN;

  int bid = rs_xcd_remap(blockIdx.x, gridDim.x);
  const int tco = bid % p.tiles_co;
  bid /= p.tiles_co;
  const int tk = bid % p.tiles_k;
  const int split = bid / p.tiles_k;
  const int tap = tk / p.tiles_ci, tci = tk - tap * p.tiles_ci;
  const int ky = STEM ? tap : (PHASE ? ((tap >> 1) & 1) : tap / p.kw);  // PHASE: tap = 4*(2*py+px) + 2*r + s
  const int kx = STEM ? 0 : (PHASE ? (tap & 1) : tap - ky * p.kw);
  const int py = (tap >> 3) & 1, px = (tap >> 2) & 1;
  const int co0 = tco * BMo;
  const int ci0 = tci * BNo;  // STEM: 0

  // source of this tile's input channels
  const float* src = p.src1;
  int Cs = STEM ? 4 : p.C1, cs = ci0;
  if (!STEM && ci0 >= p.C1) {
    src = p.src2;
    Cs = p.C2;
    cs = ci0 - p.C1;
  }

  const int chunk0 = split * p.chunks_per_split;
  const int total_chunks = (p.M + 31) >> 5;
  int chunk1 = chunk0 + p.chunks_per_split;
  if (chunk1 > total_chunks) chunk1 = total_chunks;
  const int HoWo = PHASE ? p.Hs * p.Ws : p.Ho * p.Wo;  // pixels per image of the grid the rows enumerate
  const int Wrow = PHASE ? p.Ws : p.Wo;

  // 32-bit byte offsets relative to this split's first pixel (dy) / first image (input); validated on the host
  const int m_first = chunk0 << 5;
  const int n_first = (int)rs_div((unsigned)m_first, p.div_howo);
  const long img = (long)p.Hs * p.Ws * Cs;
  const long dyimg = (long)p.Ho * p.Wo * p.Cout;
  const __amdgpu_buffer_rsrc_t rsrc_dy = PHASE ? wg_make_rsrc(p.dy + n_first * dyimg, (long)(p.N - n_first) * dyimg * 4)
                                               : wg_make_rsrc(p.dy + (long)m_first * p.Cout, ((long)p.M - m_first) * p.Cout * 4);
  const __amdgpu_buffer_rsrc_t rsrc_x = wg_make_rsrc(src + n_first * img, (long)(p.N - n_first) * img * 4);
  const int ush = p.ups ? 1 : 0;
  const int upar = p.ups == 2 ? 1 : 0;

  const bool doA = tid < TA, doB = tid < TB;  // wave-uniform
  const int a_c4 = tid % AQ, a_pg = tid / AQ;  // channel quad / group of 4 pixels staged by this thread
  const int b_c4 = tid % BQ, b_pg = tid / BQ;

  f32x4 ra[4], rb[4];   // [pixel within the group] -> 4 channels
  int lchunk = chunk0;  // next chunk to fetch

  auto load_a = [&](int e) __attribute__((always_inline)) {
    const int m = (lchunk << 5) + a_pg * 4 + e;
    int off = (doA && m < p.M) ? ((m - m_first) * p.Cout + co0 + a_c4 * 4) * 4 : -1;
    if (PHASE) {  // row m = source pixel (n, a, b): dz pixel (2a + py, 2b + px) of the same image
      const int n = (int)rs_div((unsigned)m, p.div_howo);
      const int rem = m - n * HoWo;
      const int a = (int)rs_div((unsigned)rem, p.div_wo);
      const int b = rem - a * Wrow;
      const int pix = ((n - n_first) * p.Ho + 2 * a + py) * p.Wo + 2 * b + px;
      off = (doA && m < p.M) ? (pix * p.Cout + co0 + a_c4 * 4) * 4 : -1;
    }
    ra[e] = wg_buffer_load4(rsrc_dy, off);
  };
  auto load_b = [&](int e) __attribute__((always_inline)) {
    const int m = (lchunk << 5) + b_pg * 4 + e;
    const int n = (int)rs_div((unsigned)m, p.div_howo);
    const int rem = m - n * HoWo;
    const int oy = (int)rs_div((unsigned)rem, p.div_wo);
    const int ox = rem - oy * Wrow;
    const int iy = PHASE ? oy - (1 - py) + ky : oy * p.stride - p.pad + ky;
    const int ix = PHASE ? ox - (1 - px) + kx : ox * p.stride - p.pad + (STEM ? b_c4 : kx);
    bool ok = doB && (m < p.M) && ((unsigned)iy < (unsigned)(PHASE ? p.Hs : p.Hv)) && ((unsigned)ix < (unsigned)(PHASE ? p.Ws : p.Wv));
    ok = ok && (PHASE || (((iy | ix) & upar) == 0));
    const int pix = PHASE ? ((n - n_first) * p.Hs + iy) * p.Ws + ix : ((n - n_first) * p.Hs + (iy >> ush)) * p.Ws + (ix >> ush);
    const int off = ok ? (pix * Cs + (STEM ? 0 : cs + b_c4 * 4)) * 4 : -1;
    rb[e] = wg_buffer_load4(rsrc_x, off);
  };
  // a third of the next chunk's loads, branch-free (3 + 3 + 2)
  auto load_part = [&](int part) __attribute__((always_inline)) {
    if (part == 0) { load_a(0); load_a(3); load_b(2); }
    if (part == 1) { load_a(1); load_b(0); load_b(3); }
    if (part == 2) { load_a(2); load_b(1); }
  };

  // registers -> LDS with the free 4x4 transpose: row (e*Q + c4) <- pixels 4*pg..4*pg+3 of channel 4*c4+e
  auto store_chunk = [&](int buf) __attribute__((always_inline)) {
    float* L = lds + buf * BUF;
    if (doA) {
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        f32x4 v;
        v[0] = ra[0][e];
        v[1] = ra[1][e];
        v[2] = ra[2][e];
        v[3] = ra[3][e];
        *reinterpret_cast<f32x4*>(&L[(e * AQ + a_c4) * LDT + a_pg * 4]) = v;
      }
    }
    if (doB) {
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        f32x4 v;
        v[0] = rb[0][e];
        v[1] = rb[1][e];
        v[2] = rb[2][e];
        v[3] = rb[3][e];
        *reinterpret_cast<f32x4*>(&L[(BMo + e * BQ + b_c4) * LDT + b_pg * 4]) = v;
      }
    }
  };

  f32x16 acc[TM][TN];
#pragma unroll
  for (int a = 0; a < TM; ++a)
#pragma unroll
    for (int b = 0; b < TN; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

  const int arow = wm * WM + (lane & 31);
  const int brow = BMo + wn * WN + (lane & 31);
  const int kq = (lane >> 5) * 4;

  // one quarter of a chunk = 8 of the 32 pixels; lane reads pixels 8j + 4*(lane>>5) + t with one b128 per sub-tile
  auto read_frag = [&](const float* L, int j, f32x4 (&a)[TM], f32x4 (&b)[TN]) __attribute__((always_inline)) {
#pragma unroll
    for (int tm = 0; tm < TM; ++tm) a[tm] = *reinterpret_cast<const f32x4*>(&L[(arow + 32 * tm) * LDT + 8 * j + kq]);
#pragma unroll
    for (int tn = 0; tn < TN; ++tn) b[tn] = *reinterpret_cast<const f32x4*>(&L[(brow + 32 * tn) * LDT + 8 * j + kq]);
  };
  auto mma_frag = [&](const f32x4 (&a)[TM], const f32x4 (&b)[TN]) __attribute__((always_inline)) {
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
      for (int tm = 0; tm < TM; ++tm)
#pragma unroll
        for (int tn = 0; tn < TN; ++tn)
          acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[tm][t], b[tn][t], acc[tm][tn], 0, 0, 0);
  };

  if (chunk0 < chunk1) {
#pragma unroll
    for (int part = 0; part < 3; ++part) load_part(part);
    ++lchunk;
    store_chunk(0);
    __syncthreads();
    for (int c = chunk0; c < chunk1; ++c) {
      const float* L = lds + ((c - chunk0) & 1) * BUF;
      f32x4 fa[2][TM], fb[2][TN];
      read_frag(L, 0, fa[0], fb[0]);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        if (j < 3) {
          read_frag(L, j + 1, fa[(j + 1) & 1], fb[(j + 1) & 1]);
          load_part(j);  // the prefetch past the last chunk reads zeros (m >= M) or pixels of the next split
        }
        mma_frag(fa[j & 1], fb[j & 1]);
        __builtin_amdgcn_sched_barrier(0);
      }
      ++lchunk;
      store_chunk((c - chunk0 + 1) & 1);
      __syncthreads();
    }
  }

  // D[i][j]: i = (r&3) + 8*(r>>2) + 4*(lane>>5) is an LDS row of the dy tile, j = lane&31 one of the input tile;
  // LDS row R <-> channel 4*(R % Q) + R / Q
  float* out = p.out + (long)split * p.Cout * p.K;
  const int kbase = STEM ? tap * 32 : tap * (p.C1 + p.C2) + ci0;
#pragma unroll
  for (int tm = 0; tm < TM; ++tm)
#pragma unroll
    for (int tn = 0; tn < TN; ++tn) {
      const int Rb = wn * WN + tn * 32 + (lane & 31);
      const int kk = kbase + 4 * (Rb % BQ) + Rb / BQ;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int Ra = wm * WM + tm * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        const int co = co0 + 4 * (Ra % AQ) + Ra / AQ;
        out[(long)co * p.K + kk] = acc[tm][tn][r];
      }
    }
}

// [Cout][kh][8][4] (packed stem gradient) -> KRSC [Cout][kh][kw][Cin]
__global__ void unpack_stem_weight_kernel(const float* __restrict__ packed, float* __restrict__ w, int Cout, int kh, int kw,
                                          int Cin) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  const int total = Cout * kh * kw * Cin;
  if (idx >= total) return;
  const int c = idx % Cin;
  int t = idx / Cin;
  const int s = t % kw;
  t /= kw;
  const int r = t % kh, co = t / kh;
  w[idx] = packed[((co * kh + r) * 8 + s) * 4 + c];
}

// forward-layout weights [Cout][kh][kw][Cin] -> data-gradient weights [Cin][kh][kw][Cout] with the taps flipped:
// dgrad of a convolution is itself a convolution of dy with these (conv_igemm.hip, ups = 0 / 2).
__global__ void pack_dgrad_weight_kernel(const float* __restrict__ w, float* __restrict__ out, int Cout, int taps, int Cin) {
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
    if (ci < Cin && co < Cout) out[((long)ci * taps + ftap) * Cout + co] = tile[tx][r];
  }
}

// G [Cout][16][Cin] (phase form, tap index 4*(2*py+px) + 2*r + s) -> dW [Cout][3][3][Cin]: the (py, r) pairs whose tap
// set contains ky are ky 0 -> (0,0),(1,0); 1 -> (0,1),(1,0); 2 -> (0,1),(1,1) (same in x)
__global__ void combine_phase_wgrad_f32_kernel(const float* __restrict__ g, float* __restrict__ dw, int Cin, long total) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  const int ci = (int)(i % Cin);
  long t = i / Cin;
  const int kx = (int)(t % 3), ky = (int)((t / 3) % 3);
  const long co = t / 9;
  const int ya[2][2] = {{0, ky == 0 ? 0 : 1}, {1, ky == 2 ? 1 : 0}};
  const int xa[2][2] = {{0, kx == 0 ? 0 : 1}, {1, kx == 2 ? 1 : 0}};
  float acc = 0.f;
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b) acc += g[(co * 16 + 4 * (2 * ya[a][0] + xa[b][0]) + 2 * ya[a][1] + xa[b][1]) * Cin + ci];
  dw[i] = acc;
}

struct Plan {
  int bmo, bno, variant, tiles_co, tiles_ci, taps, tiles_k, splits, chunks_per_split;
  long K;
  bool phase;
};

bool phase_ok(const rs_conv_desc* d) {
  return !d->stem && d->ups == 1 && d->kh == 3 && d->kw == 3 && d->stride == 1 && d->pad == 1 && d->Ho == 2 * d->Hs &&
         d->Wo == 2 * d->Ws && rs_knobs().wgrad_f32_phase != 0;  // (knob wgrad_f32_phase / RS_WGRAD_F32_PHASE=0: the direct form, for A/B runs)
}


bool valid(const rs_conv_desc* d) {
  if (!d) return false;
  if (d->N <= 0 || d->Hs <= 0 || d->Ws <= 0 || d->Ho <= 0 || d->Wo <= 0) return false;
  if (d->kh <= 0 || d->kw <= 0 || d->stride <= 0 || d->pad < 0) return false;
  if (d->Cout <= 0 || (d->Cout % 32) != 0) return false;
  if (d->ups < 0 || d->ups > 2) return false;
  if (d->stem) {
    if (d->C1 != 4 || d->C2 != 0 || d->kw > 8 || d->ups != 0 || (d->Cout % 64) != 0) return false;
  } else {
    if (d->C1 <= 0 || (d->C1 % 32) != 0 || d->C2 < 0 || (d->C2 % 32) != 0) return false;
  }
  return (long)d->N * d->Ho * d->Wo < (1L << 31);
}

int largest_tile(int c) { return (c % 128 == 0) ? 128 : (c % 64 == 0) ? 64 : 32; }

Plan plan(const rs_conv_desc* d) {
  Plan pl;
  pl.phase = phase_ok(d);
  const long M = pl.phase ? (long)d->N * d->Hs * d->Ws : (long)d->N * d->Ho * d->Wo;
  if (d->stem) {
    pl.bmo = 64;
    pl.bno = 32;
    pl.variant = VSTEM;
    pl.taps = d->kh;
    pl.tiles_ci = 1;
    pl.K = (long)d->kh * 32;
  } else {
    pl.bmo = largest_tile(d->Cout);
    pl.bno = largest_tile(d->C1);
    if (d->C2 > 0) {
      const int b2 = largest_tile(d->C2);
      if (b2 < pl.bno) pl.bno = b2;
    }
    if (pl.bno == 32) pl.bmo = 32;
    if (pl.bmo == 32 && pl.bno == 64) pl.bno = 32;
    pl.variant = pl.bmo == 128 ? (pl.bno == 128 ? V128x128 : V128x64)
                 : pl.bmo == 64 ? (pl.bno == 128 ? V64x128 : V64x64)
                                : (pl.bno == 128 ? V32x128 : V32x32);
    pl.taps = pl.phase ? 16 : d->kh * d->kw;
    pl.tiles_ci = (d->C1 + d->C2) / pl.bno;
    pl.K = (long)pl.taps * (d->C1 + d->C2);
  }
  pl.tiles_co = d->Cout / pl.bmo;
  pl.tiles_k = pl.taps * pl.tiles_ci;
  const long tiles = (long)pl.tiles_co * pl.tiles_k;
  const long chunks = (M + 31) / 32;
  const long target = rs_knobs().wgrad_f32_blocks;
  long s = (target + tiles - 1) / tiles;     // aim at >= 2048 blocks (knob wgrad_f32_blocks; with the LDS-DMA kernel: 8.87 ms of weight gradients per bs-8 step against 9.55 at 1024 and 10.56 at 512, profiles/r05/wgrad_f32_blocks.txt) ...
  const long smax = (chunks + 7) / 8;        // ... of at least 8 chunks (256 pixels) each
  if (s > smax) s = smax;
  if (s < 1) s = 1;
  // 32-bit byte offsets inside a split: dy spans pixels_per_split * Cout floats, the input spans the images the split
  // touches (+1 for the prefetch past its end); shrink the splits until both fit
  const long cmax = d->stem ? 4 : (d->C1 > d->C2 ? d->C1 : d->C2);
  const long img_bytes = (long)d->Hs * d->Ws * cmax * 4;
  const long howo = pl.phase ? (long)d->Hs * d->Ws : (long)d->Ho * d->Wo;
  for (;;) {
    pl.chunks_per_split = (int)((chunks + s - 1) / s);
    const long px = ((long)pl.chunks_per_split + 1) * 32;
    const long span_dy = pl.phase ? (px / howo + 2) * (long)d->Ho * d->Wo * d->Cout * 4 : px * d->Cout * 4;
    const long span_x = (px / howo + 2) * img_bytes;
    if ((span_dy < (1L << 31) && span_x < (1L << 31)) || pl.chunks_per_split == 1) break;
    s *= 2;
  }
  pl.splits = (int)((chunks + pl.chunks_per_split - 1) / pl.chunks_per_split);
  return pl;
}

}  // namespace

extern "C" long rs_conv2d_wgrad_workspace_bytes(const rs_conv_desc* d) {
  if (!valid(d)) return RS_EINVAL;
  const Plan pl = plan(d);
  const long n = (long)d->Cout * pl.K;
  const long floats = pl.splits * n + rs_reduce_scratch_floats(n, pl.splits) + (pl.phase ? n : 0);  // (+ G for the combine)
  const long wino = pl.phase ? rs_wgrad_f32_wino_workspace_floats(d) : rs_wgrad_f32_wino33_workspace_floats(d);  // (whichever form the knobs pick at launch time)
  return (floats > wino ? floats : wino) * (long)sizeof(float);
}

extern "C" int rs_conv2d_wgrad_form(const rs_conv_desc* d) {
  if (!valid(d)) return RS_EINVAL;
  if (!phase_ok(d)) return (rs_knobs().wgrad_f32_dma != 0 && rs_wgrad_f32_wino33_ok(d)) ? 4 : 0;
  return (rs_knobs().wgrad_f32_dma != 0 && rs_wgrad_f32_wino_ok(d)) ? 3 : 2;
}

extern "C" int rs_conv2d_wgrad(const rs_conv_desc* d, const float* dy, const float* src1, const float* src2, float* dw,
                               void* workspace, rs_stream_t stream) {
  if (!valid(d) || !dy || !src1 || !dw || !workspace) return RS_EINVAL;
  if (d->C2 > 0 && !src2) return RS_EINVAL;
  const Plan pl = plan(d);
  if (pl.phase && rs_knobs().wgrad_f32_dma != 0 && rs_wgrad_f32_wino_ok(d))  // DecoderBlock in the Winograd domain (conv_wgrad_wino_f32.hip)
    return rs_wgrad_f32_wino_launch(d, dy, src1, src2, dw, reinterpret_cast<float*>(workspace), (hipStream_t)stream);
  if (!pl.phase && rs_knobs().wgrad_f32_dma != 0 && rs_wgrad_f32_wino33_ok(d))  // stride-1 3x3 in the Winograd domain (conv_wgrad_wino33_f32.hip)
    return rs_wgrad_f32_wino33_launch(d, dy, src1, dw, reinterpret_cast<float*>(workspace), (hipStream_t)stream);
  WgradArgs a;
  a.dy = dy;
  a.src1 = src1;
  a.src2 = src2;
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
  a.tiles_ci = pl.tiles_ci;
  a.tiles_k = pl.tiles_k;
  a.chunks_per_split = pl.chunks_per_split;
  const int grid = pl.tiles_co * pl.tiles_k * pl.splits;
  hipStream_t s = (hipStream_t)stream;
  // knob wgrad_f32_dma (-1 / 1: every non-stem launch; 0: the register-staged kernel below, for A/B runs): the LDS-DMA kernel
  // of conv_wgrad_f32_dma.hip -- same blocks and partial tiles, both operands copied HBM -> LDS as they lie
  const bool dma = !d->stem && rs_knobs().wgrad_f32_dma != 0;
  int rc = 0;
  if (dma) {
    rc = rs_wgrad_f32_dma_launch(pl.variant, pl.phase, grid, s, a);
  } else if (pl.phase) {
    switch (pl.variant) {
      case V128x128: conv_wgrad_f32<128, 128, 2, 2, 2><<<grid, 256, 0, s>>>(a); break;
      case V128x64: conv_wgrad_f32<128, 64, 2, 2, 2><<<grid, 256, 0, s>>>(a); break;
      case V64x128: conv_wgrad_f32<64, 128, 2, 2, 2><<<grid, 256, 0, s>>>(a); break;
      case V64x64: conv_wgrad_f32<64, 64, 2, 2, 2><<<grid, 256, 0, s>>>(a); break;
      case V32x128: conv_wgrad_f32<32, 128, 1, 4, 2><<<grid, 256, 0, s>>>(a); break;
      case V32x32: conv_wgrad_f32<32, 32, 1, 1, 2><<<grid, 64, 0, s>>>(a); break;
      default: return RS_EINVAL;
    }
    rc = RS_LAUNCH_RESULT();
  } else {
    switch (pl.variant) {
      case V128x128: conv_wgrad_f32<128, 128, 2, 2, 0><<<grid, 256, 0, s>>>(a); break;
      case V128x64: conv_wgrad_f32<128, 64, 2, 2, 0><<<grid, 256, 0, s>>>(a); break;
      case V64x128: conv_wgrad_f32<64, 128, 2, 2, 0><<<grid, 256, 0, s>>>(a); break;
      case V64x64: conv_wgrad_f32<64, 64, 2, 2, 0><<<grid, 256, 0, s>>>(a); break;
      case V32x128: conv_wgrad_f32<32, 128, 1, 4, 0><<<grid, 256, 0, s>>>(a); break;
      case V32x32: conv_wgrad_f32<32, 32, 1, 1, 0><<<grid, 64, 0, s>>>(a); break;
      case VSTEM: conv_wgrad_f32<64, 32, 2, 1, 1><<<grid, 128, 0, s>>>(a); break;
      default: return RS_EINVAL;
    }
    rc = RS_LAUNCH_RESULT();
  }
  if (rc) return rc;
  const long n = (long)d->Cout * pl.K;  // multiple of 4 (phase form: Cout x 16 x Cin)
  float* scratch = a.out + (long)pl.splits * n;
  if (!pl.phase) return rs_reduce_splits(a.out, dw, n, pl.splits, scratch, stream);
  float* gbuf = scratch + rs_reduce_scratch_floats(n, pl.splits);
  const int rc2 = rs_reduce_splits(a.out, gbuf, n, pl.splits, scratch, stream);
  if (rc2) return rc2;
  const long total = (long)d->Cout * 9 * (d->C1 + d->C2);
  combine_phase_wgrad_f32_kernel<<<rs_cdiv(total, 256), 256, 0, s>>>(gbuf, dw, d->C1 + d->C2, total);
  return RS_LAUNCH_RESULT();
}

extern "C" int rs_unpack_stem_weight(const float* packed, float* w_krsc, int Cout, int kh, int kw, int Cin,
                                     rs_stream_t stream) {
  if (!packed || !w_krsc || Cout <= 0 || kh <= 0 || kw <= 0 || kw > 8 || Cin <= 0 || Cin > 4) return RS_EINVAL;
  const int total = Cout * kh * kw * Cin;
  unpack_stem_weight_kernel<<<rs_cdiv(total, 256), 256, 0, (hipStream_t)stream>>>(packed, w_krsc, Cout, kh, kw, Cin);
  return RS_LAUNCH_RESULT();
}

extern "C" int rs_pack_dgrad_weight(const float* w_krsc, float* out, int Cout, int kh, int kw, int Cin, rs_stream_t stream) {
  if (!w_krsc || !out || Cout <= 0 || kh <= 0 || kw <= 0 || Cin <= 0) return RS_EINVAL;
  dim3 grid(rs_cdiv(Cin, 32), rs_cdiv(Cout, 32), kh * kw);
  pack_dgrad_weight_kernel<<<grid, 256, 0, (hipStream_t)stream>>>(w_krsc, out, Cout, kh * kw, Cin);
  return RS_LAUNCH_RESULT();
}

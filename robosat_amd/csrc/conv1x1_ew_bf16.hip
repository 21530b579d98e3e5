// conv1x1_ew_bf16.hip -- the bf16 1x1 / stride-1 convolutions of the TRAIN-mode forward (Bottleneck.conv1 / conv3, reference
// torchvision Bottleneck via robosat/unet.py:94,122-130 under tools/train.py:180-188): raw output + the per-tile BatchNorm
// partial sums (sum y, sum y^2 of the values as stored), with the EPILOGUE ON ITS OWN WAVES.
//
// Why (profiles/r04/k_slope_1x1.txt): such a launch costs 19 us of per-tile fixed work + 1.8 us per 32-channel chunk step in
// the generic kernel, and the two do not overlap; the chunk steps are bound by the CUs' LDS-DMA paths, not by MFMA issue, so a
// block can give half of its waves to the epilogue without slowing its main loop (conv1x1_ew_f32.hip is the fp32 sibling,
// where that trade did not pay: its main loop IS the matrix cores).  A block (one per CU, persistent) is
//   waves 0-3  main loop only: 128 pixels x 128 couts per tile (2 x 2 waves, 64 x 64 each), 32-channel chunks (64-byte rows)
//              by LDS-DMA into a ring of FIVE buffers -- four chunks in flight, across tiles, counted vmcnt waits -- and
//              v_mfma_f32_32x32x16_bf16; at the end of a tile the accumulators go to an LDS staging tile as bf16;
//   waves 4-7  epilogue only, one tile behind: 16-byte stores of the staged rows, the statistics of the stored values
//              (per-thread sums over its 8 rows -> the wave's 4 row lanes by two lane exchanges -> the four waves through
//              LDS -> one partial row per tile), a slice per chunk step of the tile the other four are computing.
// Both halves pass the same sequence of s_barrier instructions (one per chunk step + two at the end).
//
// STATUS (round 5, profiles/r05/ew_bf16_1x1.txt): correct on its first run (output bits identical to the generic kernel, partial
// rows equal to fp32 association) and NOT faster -- x 1.04-1.10 on the long-K / narrow launches, x 0.83-0.95 on the short-K / wide
// ones it was built for: one block per CU leaves four waves to issue a tile's stores where the generic kernel has sixteen.
// Measurement only: built with `make EXP=1`, reached with knob conv1x1_ew_bf16 = 1 (RS_CONV1X1_EW_BF16=1).
#define RS_CONV_INSTANTIATE
#include "conv_igemm_dma_kernel.h"

namespace {

constexpr int EB_BM = 128, EB_BN = 128, EB_ROWB = 64, EB_KC = 32;
constexpr int EB_RING = 5;
constexpr int EB_BUF = (EB_BM + EB_BN) * EB_ROWB;  // bytes per ring buffer: pixel rows, then filter rows
constexpr int EB_SROW = EB_BN * 2 + 16;            // staging row (bytes, bf16): 272 = 68 dwords
constexpr int EB_STAGE = EB_BM * EB_SROW;
constexpr int EB_RED = 4 * 2 * EB_BN * 4;          // [wave][sum | sum of squares][cout] floats
constexpr int EB_NG = EB_BM / 16;                  // epilogue groups per tile: 16 rows x 16 pieces of 16 bytes (8 couts)

typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ void eb_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

__global__ __launch_bounds__(512, 2) void conv1x1_ew_bf16_stats_kernel(const ConvArgsT<bf16_t> p) {
  __shared__ __attribute__((aligned(16))) unsigned char smem[EB_RING * EB_BUF + EB_STAGE + EB_RED];
  unsigned char* stage = smem + EB_RING * EB_BUF;
  float* red = reinterpret_cast<float*>(smem + EB_RING * EB_BUF + EB_STAGE);

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int K = p.C1, nk = K / EB_KC;
  const int ntiles = p.Cout / EB_BN;
  const int items = ((p.M + EB_BM - 1) / EB_BM) * ntiles;
  const int first = rs_xcd_remap(blockIdx.x, gridDim.x);
  const int nitems = first < items ? (items - first + (int)gridDim.x - 1) / (int)gridDim.x : 0;
  const int total = nitems * nk;

  if (wave < 4) {
    // ================================================ main-loop waves ================================================
    const int wm = wave >> 1, wn = wave & 1;
    const int ra = lane >> 2, pp = lane & 3;
    const int gp = pp ^ ((ra >> 2) & 3);
    const unsigned int lds0 = __builtin_amdgcn_readfirstlane(rb_lds_addr(smem));
    const __amdgpu_buffer_rsrc_t rsrcw = rb_make_rsrc(p.wgt, (long)p.Cout * K * 2);
    __amdgpu_buffer_rsrc_t rsrca = rb_make_rsrc(p.src1, 0);
    int f_seq = 0, f_kc = 0, f_b = 0;  // fetch cursor: item, chunk of it, ring buffer
    int voff[4];                       // two pixel-row instructions (relative to the item's first row), two filter-row instructions
    auto fetch_item = [&]() __attribute__((always_inline)) {
      const int it = first + f_seq * (int)gridDim.x;
      const int mt = __builtin_amdgcn_readfirstlane(it / ntiles);
      const int nt = it - mt * ntiles;
      const int m0 = mt * EB_BM, n0 = nt * EB_BN;
      rsrca = rb_make_rsrc(p.src1 + (long)m0 * K, (long)(p.M - m0) * K * 2);  // (rows past M are past the descriptor: zeros)
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        voff[j] = (16 * (wave + 4 * j) + ra) * (K * 2) + gp * 16;
        voff[2 + j] = (n0 + 16 * (wave + 4 * j) + ra) * (K * 2) + gp * 16;
      }
    };
    auto issue = [&](int j) __attribute__((always_inline)) {  // j compile-time: instruction wave + 4 j of the chunk's 16
      const unsigned int dst = lds0 + f_b * EB_BUF + (wave + 4 * j) * 1024;
      if (j < 2) rb_dma16s(rsrca, dst, voff[j], f_kc * EB_ROWB);
      else rb_dma16s(rsrcw, dst, voff[j], f_kc * EB_ROWB);
    };
    auto advance = [&]() __attribute__((always_inline)) {
      f_b = f_b == EB_RING - 1 ? 0 : f_b + 1;
      if (++f_kc == nk) {
        f_kc = 0;
        ++f_seq;
      }
    };

    const int frow = lane & 31;
    const int fl = (frow >> 2) & 3;
    int foff[2];
#pragma unroll
    for (int s = 0; s < 2; ++s) foff[s] = ((2 * s + (lane >> 5)) ^ fl) * 16;
    const int abase = (wm * 64 + frow) * EB_ROWB;
    const int bbase = (EB_BM + wn * 64 + frow) * EB_ROWB;

    f32x16 acc[2][2];  // [cout sub-tile][pixel sub-tile]
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int b = 0; b < 2; ++b)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

    {  // prologue: the first four chunks
      const int n0c = total < EB_RING - 1 ? total : EB_RING - 1;
      for (int c = 0; c < n0c; ++c) {
        if (f_kc == 0) fetch_item();
#pragma unroll
        for (int j = 0; j < 4; ++j) issue(j);
        advance();
      }
    }
    int g = 0, c_b = 0;
    for (int seq = 0; seq < nitems; ++seq) {
      for (int kc = 0; kc < nk; ++kc, ++g) {
        // chunk g has landed when at most the younger chunks' instructions of this wave are outstanding (4 per chunk)
        const int younger = total - 1 - g;
        if (younger >= 3) rb_dma_wait_n<12>();
        else if (younger == 2) rb_dma_wait_n<8>();
        else if (younger == 1) rb_dma_wait_n<4>();
        else rb_dma_wait();
        eb_barrier();  // chunk g is in buffer c_b; the buffer chunk g - 1 was read from is free again
        const bool more = g + EB_RING - 1 < total;
        if (more && f_kc == 0) fetch_item();
        const unsigned char* L = smem + c_b * EB_BUF;
        u32x4 fa[2][2], fb[2][2];
#pragma unroll
        for (int s = 0; s < 2; ++s) {
#pragma unroll
          for (int t = 0; t < 2; ++t) {
            fa[s][t] = *reinterpret_cast<const u32x4*>(L + abase + 32 * t * EB_ROWB + foff[s]);
            fb[s][t] = *reinterpret_cast<const u32x4*>(L + bbase + 32 * t * EB_ROWB + foff[s]);
          }
        }
#pragma unroll
        for (int s = 0; s < 2; ++s)
#pragma unroll
          for (int tm = 0; tm < 2; ++tm)
#pragma unroll
            for (int tn = 0; tn < 2; ++tn) {
              const int q = (s * 2 + tm) * 2 + tn;
              if (q % 2 == 0) {
                if (more) issue(q / 2);
              }
              mma16(acc[tn][tm], fb[s][tn], fa[s][tm], bf16_t());
            }
        if (more) advance();
        c_b = c_b == EB_RING - 1 ? 0 : c_b + 1;
      }
      // ---- accumulators -> staging as bf16 (D rows = couts 8 q + 4 (lane >> 5) + e, D columns = pixels lane & 31)
#pragma unroll
      for (int tm = 0; tm < 2; ++tm) {
        const int pr = wm * 64 + 32 * tm + (lane & 31);
#pragma unroll
        for (int tn = 0; tn < 2; ++tn) {
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            typedef __bf16 bf16x4_t __attribute__((ext_vector_type(4)));
            bf16x4_t v;
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = (bf16_t)acc[tn][tm][4 * q + e];
            const int col = wn * 64 + 32 * tn + 8 * q + 4 * (lane >> 5);
            *reinterpret_cast<u32x2*>(stage + pr * EB_SROW + col * 2) = __builtin_bit_cast(u32x2, v);
          }
#pragma unroll
          for (int r = 0; r < 16; ++r) acc[tn][tm][r] = 0.f;
        }
      }
    }
    eb_barrier();  // publishes the last tile's staging
    eb_barrier();  // (the epilogue waves' partial sums)
    rb_dma_wait();
  } else {
    // ================================================ epilogue waves =================================================
    const int e = tid - 256;
    const int c16 = e & 15, r16 = e >> 4;
    const int w4 = wave - 4;
    const int gpi = nk > 1 ? (EB_NG + nk - 2) / (nk - 1) : EB_NG;  // groups per chunk step (steps 0 .. nk-2 drain a tile)
    long obase = 0;
    int m_first = 0, mt_cur = 0, n0_cur = 0;
    float st0[8], st1[8];
    auto begin_tile = [&](int seq) __attribute__((always_inline)) {
      const int it = first + seq * (int)gridDim.x;
      mt_cur = it / ntiles;
      n0_cur = (it - mt_cur * ntiles) * EB_BN;
      m_first = mt_cur * EB_BM + r16;
      obase = (long)m_first * p.Cout + n0_cur + 8 * c16;
#pragma unroll
      for (int q = 0; q < 8; ++q) st0[q] = st1[q] = 0.f;
    };
    auto drain = [&](int lo, int hi) __attribute__((always_inline)) {
#pragma unroll
      for (int j = 0; j < EB_NG; ++j) {
        if (j >= lo && j < hi && m_first + 16 * j < p.M) {
          const u32x4 raw = *reinterpret_cast<const u32x4*>(stage + (16 * j + r16) * EB_SROW + c16 * 16);
          *reinterpret_cast<u32x4*>(p.out + obase + (long)(16 * j) * p.Cout) = raw;
          const bf16x8 t = __builtin_bit_cast(bf16x8, raw);
#pragma unroll
          for (int q = 0; q < 8; ++q) {
            const float w = (float)t[q];
            st0[q] += w;
            st1[q] += w * w;
          }
        }
      }
      if (lo < EB_NG && hi >= EB_NG) {  // the tile's last group is in this slice: the wave's 4 row lanes, then LDS
#pragma unroll
        for (int q = 0; q < 8; ++q) {
          st0[q] += __shfl_xor(st0[q], 16);
          st0[q] += __shfl_xor(st0[q], 32);
          st1[q] += __shfl_xor(st1[q], 16);
          st1[q] += __shfl_xor(st1[q], 32);
        }
        if (lane < 16) {
#pragma unroll
          for (int q = 0; q < 8; ++q) {
            red[(w4 * 2 + 0) * EB_BN + 8 * c16 + q] = st0[q];
            red[(w4 * 2 + 1) * EB_BN + 8 * c16 + q] = st1[q];
          }
        }
      }
    };
    auto finish = [&]() __attribute__((always_inline)) {  // one partial row of the tile: the four waves' sums in wave order
      const int stat = e >> 7, c = e & 127;
      const float a = ((red[(0 * 2 + stat) * EB_BN + c] + red[(1 * 2 + stat) * EB_BN + c]) + red[(2 * 2 + stat) * EB_BN + c]) +
                      red[(3 * 2 + stat) * EB_BN + c];
      p.stats[((long)mt_cur * 2 + stat) * p.Cout + n0_cur + c] = a;
    };
    if (nitems > 0) {
      for (int kc = 0; kc < nk; ++kc) eb_barrier();  // tile 0 is being computed
      for (int seq = 1; seq < nitems; ++seq) {
        begin_tile(seq - 1);
        for (int kc = 0; kc < nk - 1; ++kc) {
          eb_barrier();
          drain(kc * gpi, (kc + 1) * gpi);
        }
        eb_barrier();
        finish();
      }
      eb_barrier();
      begin_tile(nitems - 1);
      drain(0, EB_NG);
      eb_barrier();
      finish();
    } else {
      eb_barrier();
      eb_barrier();
    }
  }
}

}  // namespace

// 1: this launch can take the epilogue-wave kernel (geometry only)
int rs_conv1x1_ew_bf16_stats_ok(const rs_conv_desc* d) {
  if (!d || d->stem || d->kh != 1 || d->kw != 1 || d->stride != 1 || d->pad != 0 || d->ups != 0 || d->C2 != 0) return 0;
  if (d->Ho != d->Hs || d->Wo != d->Ws) return 0;
  if (d->C1 < 2 * EB_KC || (d->C1 % EB_KC) || d->Cout <= 0 || (d->Cout % EB_BN)) return 0;
  if ((long)d->Cout * d->C1 * 2 >= (1L << 31) || (long)(EB_BM + 16) * d->C1 * 2 >= (1L << 31)) return 0;
  return 1;
}

int rs_conv1x1_ew_bf16_stats_launch(const ConvArgsT<bf16_t>& a, hipStream_t s) {
  int dev = 0, cus = 0;
  if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0)
    cus = 256;
  const long items = (long)rs_cdiv(a.M, EB_BM) * (a.Cout / EB_BN);
  if (items <= 0 || items >= (1L << 31) || !a.stats) return RS_EINVAL;
  const int grid = (int)(items < cus ? items : cus);
  conv1x1_ew_bf16_stats_kernel<<<grid, 512, 0, s>>>(a);
  return RS_LAUNCH_RESULT();
}

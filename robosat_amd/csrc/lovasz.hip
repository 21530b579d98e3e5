// LovaszLoss2d (reference robosat/losses.py:86-119) on the GPU, all images of the batch in one set of launches.
//
// Per image n, over the flattened C*H*W vector i (NCHW order, i = c*HW + hw) with the one-hot mask m_i as labels:
//     err_i   = 1 - (2 m_i - 1) * x_i
//     sort err descending (permutation pi), lab_r = m_{pi(r)}
//     inter_r = gts - cumsum(lab)_r ;  union_r = gts + cumsum(1 - lab)_r ;  jac_r = 1 - inter_r / union_r
//     delta_0 = jac_0, delta_r = jac_r - jac_{r-1}
//     loss_n  = sum_r relu(err_{pi(r)}) * delta_r ;   loss = mean_n loss_n
// gts = sum(m) = H*W exactly (one label per pixel).  All of inter/union/jac/delta are evaluated in fp32 with the same
// operations as the reference, so for the same permutation they are bit-identical to torch's.
// Gradient (what autograd gives the reference): d loss / d x_i = -(2 m_i - 1) * [err_i > 0] * delta_{rank(i)} / N.
//
// Sort = LSD radix sort, 4 passes x 8 bits, keys = order-preserving uint32 image of err (inverted for descending),
// payload = i | (m_i << 31).  Each pass: per-block digit histograms -> per-image exclusive scan (digit-major) ->
// stable scatter.  The scatter works on 8192-element tiles: every wave ranks its 2048 elements among equal digits
// (64-bit ballots per batch of 64 + wave-private running counters in LDS), the tile is first sorted by digit INSIDE LDS,
// and only then written out -- consecutive threads then write consecutive addresses of a digit's run, instead of 4-byte
// stores sprayed over 256 buckets (which cost the first version 0.39 ms per pass against a 0.07 ms traffic bound).
// The rest is a segmented prefix sum (block sums -> scan -> apply) fused with the Jaccard deltas, the dot product
// (fp64 partials) and the gradient scatter.  HBM-bound integer/byte work throughout.
#include "common.h"

namespace {

// elements per 256-thread block of the sort kernels: 8192 (4 waves x 32 batches of 64; 68 KB of LDS: two blocks per CU) or,
// for the large sorts, 4096 (four blocks per CU: -9 % on the bs-32 step's 16.8 M keys, -14 % at 33.5 M; +7 % on a bs-8 step's
// 4.2 M keys, where twice as many per-block histograms outweigh it: profiles/r05/lovasz.txt).  Chosen by the total key count;
// the sort is exact and stable either way.
constexpr long kSmallSortKeys = 8L << 20;
inline int sort_chunk(long total_keys) { return total_keys >= kSmallSortKeys ? 4096 : 8192; }
constexpr int kScanChunk = 1024;  // elements per 256-thread block in the prefix-sum kernels

typedef unsigned int lv_u32x4 __attribute__((ext_vector_type(4)));
typedef long long lv_i64x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ uint32_t desc_key(float f) {
  const uint32_t u = __float_as_uint(f);
  const uint32_t asc = (u & 0x80000000u) ? ~u : (u | 0x80000000u);
  return ~asc;
}

__device__ __forceinline__ float key_to_float(uint32_t key) {
  const uint32_t asc = ~key;
  const uint32_t u = (asc & 0x80000000u) ? (asc & 0x7fffffffu) : ~asc;
  return __uint_as_float(u);
}

__global__ void lovasz_keys_kernel(const float* __restrict__ x, const long long* __restrict__ tgt, uint32_t* __restrict__ keys,
                                   uint32_t* __restrict__ vals, long P, long HW) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const long n = blockIdx.y;
  if (i >= P) return;
  const long c = i / HW, hw = i - c * HW;
  const uint32_t m = (tgt[n * HW + hw] == c) ? 1u : 0u;
  const float v = x[n * P + i];
  const float err = 1.f - (m ? v : -v);  // 1 - (2m-1)*x
  keys[n * P + i] = desc_key(err);
  vals[n * P + i] = (uint32_t)i | (m << 31);
}

// The same for HW % 4 == 0 (every tile size the tools produce): four consecutive pixels of one class plane per thread -- one
// 16-byte load of the logits, two of the int64 labels, two 16-byte stores (round 5: the scalar kernel ran at ~2.5 TB/s).
__global__ __launch_bounds__(256) void lovasz_keys4_kernel(const float* __restrict__ x, const long long* __restrict__ tgt,
                                                           uint32_t* __restrict__ keys, uint32_t* __restrict__ vals, long P, long HW) {
  const long i = ((long)blockIdx.x * blockDim.x + threadIdx.x) * 4;
  const long n = blockIdx.y;
  if (i >= P) return;
  const long c = i / HW, hw = i - c * HW;  // (HW % 4 == 0: the four elements share the class plane)
  const f32x4 v = *reinterpret_cast<const f32x4*>(x + n * P + i);
  const lv_i64x2 t0 = *reinterpret_cast<const lv_i64x2*>(tgt + n * HW + hw), t1 = *reinterpret_cast<const lv_i64x2*>(tgt + n * HW + hw + 2);
  const long long t[4] = {t0[0], t0[1], t1[0], t1[1]};
  lv_u32x4 k, w;
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const uint32_t m = (t[e] == c) ? 1u : 0u;
    const float err = 1.f - (m ? v[e] : -v[e]);
    k[e] = desc_key(err);
    w[e] = (uint32_t)(i + e) | (m << 31);
  }
  *reinterpret_cast<lv_u32x4*>(keys + n * P + i) = k;
  *reinterpret_cast<lv_u32x4*>(vals + n * P + i) = w;
}

// counts[n][b][256]
template <int CHUNK>
__global__ __launch_bounds__(256) void radix_hist_kernel(const uint32_t* __restrict__ keys, uint32_t* __restrict__ counts,
                                                         long P, int nblk, int shift) {
  __shared__ uint32_t hist[256];
  const int tid = threadIdx.x;
  const long n = blockIdx.y;
  const int b = blockIdx.x;
  hist[tid] = 0;
  __syncthreads();
  const long base = (long)b * CHUNK;
  if ((P & 3) == 0) {  // 16-byte loads: four consecutive keys per thread
    for (int k = 0; k < CHUNK / 1024; ++k) {
      const long i = base + (k * 256 + tid) * 4;
      if (i < P) {
        // (counting the lanes that share a digit by ballot and adding once per group was measured: -50 % on the passes whose
        // digits are concentrated -- sign + exponent -- and +60 % on the uniform ones; a wash over a sort: profiles/r05/lovasz.txt)
        const lv_u32x4 v = *reinterpret_cast<const lv_u32x4*>(keys + n * P + i);
#pragma unroll
        for (int e = 0; e < 4; ++e) atomicAdd(&hist[(v[e] >> shift) & 255u], 1u);
      }
    }
  } else {
    for (int k = 0; k < CHUNK / 256; ++k) {
      const long i = base + k * 256 + tid;
      if (i < P) atomicAdd(&hist[(keys[n * P + i] >> shift) & 255u], 1u);
    }
  }
  __syncthreads();
  counts[(n * nblk + b) * 256 + tid] = hist[tid];
}

// exclusive scan of counts in (digit-major, block-minor) order, per image; in place: counts -> offsets.  One block of 1024 per
// image: digit d = tid & 255, quarter g = tid >> 8 of the blocks (round 4: 256 threads walked all blocks twice -- 22-30 us of
// dependent round trips for a few hundred KB, four times per sort).
__global__ __launch_bounds__(1024) void radix_scan_kernel(uint32_t* __restrict__ counts, int nblk) {
  __shared__ uint32_t part[4][256];
  __shared__ uint32_t tot[256];
  const int d = threadIdx.x & 255, g = threadIdx.x >> 8;
  uint32_t* c = counts + (long)blockIdx.x * nblk * 256;
  const int q = (nblk + 3) / 4;
  const int b0 = g * q, b1 = (b0 + q) < nblk ? (b0 + q) : nblk;
  uint32_t s = 0;
  for (int b = b0; b < b1; ++b) s += c[(long)b * 256 + d];
  part[g][d] = s;
  __syncthreads();
  if (g == 0) tot[d] = (part[0][d] + part[1][d]) + (part[2][d] + part[3][d]);
  __syncthreads();
  if (threadIdx.x < 64) {  // exclusive scan of the 256 digit totals by one wave: 4 digits per lane
    const int l = threadIdx.x;
    const uint32_t t0 = tot[4 * l], t1 = tot[4 * l + 1], t2 = tot[4 * l + 2], t3 = tot[4 * l + 3];
    const uint32_t mine = t0 + t1 + t2 + t3;
    uint32_t inc = mine;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const uint32_t t = __shfl_up(inc, o, 64);
      if (l >= o) inc += t;
    }
    const uint32_t ex = inc - mine;
    tot[4 * l] = ex;
    tot[4 * l + 1] = ex + t0;
    tot[4 * l + 2] = ex + t0 + t1;
    tot[4 * l + 3] = ex + t0 + t1 + t2;
  }
  __syncthreads();
  uint32_t run = tot[d];
  for (int gg = 0; gg < g; ++gg) run += part[gg][d];
  for (int b = b0; b < b1; ++b) {
    const uint32_t t = c[(long)b * 256 + d];
    c[(long)b * 256 + d] = run;
    run += t;
  }
}

template <int CHUNK>
__global__ __launch_bounds__(256) void radix_scatter_kernel(const uint32_t* __restrict__ keys, const uint32_t* __restrict__ vals,
                                                            uint32_t* __restrict__ okeys, uint32_t* __restrict__ ovals,
                                                            const uint32_t* __restrict__ offsets, long P, int nblk, int shift,
                                                            int xcd_affine) {
  // Round 5: each element's rank inside its wave is fixed in the FIRST sweep -- one ballot match per batch of 64, the leader
  // of every digit group adds the group's size to the wave's counter with a RETURNING LDS atomic (= the elements of that digit
  // in the wave's earlier batches) and hands it to the group by a lane read -- and kept in a register; the second sweep is then
  // one table read + one 8-byte store per element.  Rounds 1-4 counted first (an atomic per element), matched again in the
  // second sweep and read + updated the running counters there: ~8 scattered LDS operations per element against ~4 now
  // (scattered = bank conflicts by construction: SQ_LDS_BANK_CONFLICT was 64 % of this kernel's LDS cycles).  Same output.
  typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
  constexpr int WCH = CHUNK / 4;       // elements per wave
  constexpr int NB = WCH / 64;         // batches of 64 per wave
  __shared__ u32x2 lkv[CHUNK];         // the tile, sorted by digit: (key, value)
  __shared__ uint32_t loc[4][256];     // per wave: histogram, then position of the wave's first element of each digit in the tile
  __shared__ uint32_t gdelta[256];     // global position of a digit's run minus its position in the tile
  __shared__ uint32_t wsum[4];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  long n = blockIdx.y;
  int b = blockIdx.x;
  if (xcd_affine) {  // all tiles of an image on one XCD (see lovasz_apply_kernel)
    const long L = (long)blockIdx.y * gridDim.x + blockIdx.x;
    const long k = L >> 3;
    n = (L & 7) + 8 * (k / nblk);
    b = (int)(k % nblk);
  }
  const long start = (long)b * CHUNK + wave * WCH;

  uint32_t key[NB], val[NB], rnk[NB];
#pragma unroll
  for (int w = 0; w < 4; ++w) loc[w][tid] = 0;
  __syncthreads();
  const unsigned long long lt = (1ull << lane) - 1ull;
#pragma unroll
  for (int k = 0; k < NB; ++k) {  // fully unrolled: key[] / val[] / rnk[] stay in registers
    const long i = start + k * 64 + lane;
    const bool valid = i < P;
    key[k] = 0xffffffffu;
    val[k] = 0;
    if (valid) {
      key[k] = keys[n * P + i];
      val[k] = vals[n * P + i];
    }
    const uint32_t d = (key[k] >> shift) & 255u;
    unsigned long long peers = __ballot(valid);
#pragma unroll
    for (int bit = 0; bit < 8; ++bit) {
      const bool one = (d >> bit) & 1u;
      const unsigned long long bm = __ballot(valid && one);
      peers &= one ? bm : ~bm;
    }
    // loc[wave][] is private to this wave and a wave's LDS operations execute in issue order: batch k + 1's atomic sees batch k's
    const int leader = __ffsll((long long)peers) - 1;
    uint32_t before = 0;
    if (valid && leader == lane) before = atomicAdd(&loc[wave][d], (uint32_t)__popcll(peers));
    before = __shfl(before, leader & 63, 64);
    rnk[k] = before + (uint32_t)__popcll(peers & lt);
  }
  __syncthreads();
  {  // digit-major, wave-minor exclusive offsets inside the tile (thread d owns digit d)
    const uint32_t c0 = loc[0][tid], c1 = loc[1][tid], c2 = loc[2][tid], c3 = loc[3][tid];
    const uint32_t tot = c0 + c1 + c2 + c3;
    uint32_t inc = tot;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const uint32_t t = __shfl_up(inc, o, 64);
      if (lane >= o) inc += t;
    }
    if (lane == 63) wsum[wave] = inc;
    __syncthreads();
    uint32_t woff = 0;
#pragma unroll
    for (int w = 0; w < 4; ++w)
      if (w < wave) woff += wsum[w];
    const uint32_t ex = woff + inc - tot;  // elements of smaller digits in the tile
    loc[0][tid] = ex;
    loc[1][tid] = ex + c0;
    loc[2][tid] = ex + c0 + c1;
    loc[3][tid] = ex + c0 + c1 + c2;
    gdelta[tid] = offsets[(n * nblk + b) * 256 + tid] - ex;
  }
  __syncthreads();
#pragma unroll
  for (int k = 0; k < NB; ++k) {
    if ((start + k * 64 + lane) < P) {
      u32x2 kv;
      kv[0] = key[k];
      kv[1] = val[k];
      lkv[loc[wave][(key[k] >> shift) & 255u] + rnk[k]] = kv;
    }
  }
  __syncthreads();
  const long tile0 = (long)b * CHUNK;
  const int count = (int)((P - tile0) < CHUNK ? (P - tile0) : CHUNK);
  for (int i = tid; i < count; i += 256) {
    const u32x2 kv = lkv[i];
    const uint32_t g = gdelta[(kv[0] >> shift) & 255u] + (uint32_t)i;
    okeys[n * P + g] = kv[0];
    ovals[n * P + g] = kv[1];
  }
}

__device__ __forceinline__ uint32_t block_exclusive_scan_256(uint32_t v, uint32_t* smem /*[4]*/, uint32_t* total) {
  // inclusive scan inside the wave
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  uint32_t inc = v;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const uint32_t t = __shfl_up(inc, o, 64);
    if (lane >= o) inc += t;
  }
  if (lane == 63) smem[wave] = inc;
  __syncthreads();
  uint32_t woff = 0;
#pragma unroll
  for (int w = 0; w < 4; ++w)
    if (w < wave) woff += smem[w];
  *total = smem[0] + smem[1] + smem[2] + smem[3];
  return woff + inc - v;
}

// per-block number of labels among the sorted elements: bsum[n][b]
__global__ __launch_bounds__(256) void lovasz_blocksum_kernel(const uint32_t* __restrict__ vals, uint32_t* __restrict__ bsum,
                                                              long P, int nblk) {
  __shared__ uint32_t sm[4];
  const long n = blockIdx.y;
  const long r0 = (long)blockIdx.x * kScanChunk + threadIdx.x * 4;
  uint32_t s = 0;
  if ((P & 3) == 0 && r0 + 3 < P) {  // (P % 4 == 0: 16-byte aligned whatever the image)
    const lv_u32x4 v = *reinterpret_cast<const lv_u32x4*>(vals + n * P + r0);
    s = (v[0] >> 31) + (v[1] >> 31) + (v[2] >> 31) + (v[3] >> 31);
  } else {
#pragma unroll
    for (int e = 0; e < 4; ++e)
      if (r0 + e < P) s += vals[n * P + r0 + e] >> 31;
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
  if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) bsum[n * nblk + blockIdx.x] = sm[0] + sm[1] + sm[2] + sm[3];
}

// in-place exclusive scan of bsum[n][0..nblk) (one block per image)
__global__ __launch_bounds__(256) void lovasz_scan_blocks_kernel(uint32_t* __restrict__ bsum, int nblk) {
  __shared__ uint32_t sm[4];
  uint32_t* a = bsum + (long)blockIdx.x * nblk;
  const int per = (nblk + 255) / 256;
  const int i0 = threadIdx.x * per;
  uint32_t s = 0;
  for (int k = 0; k < per; ++k)
    if (i0 + k < nblk) s += a[i0 + k];
  uint32_t total;
  uint32_t run = block_exclusive_scan_256(s, sm, &total);
  for (int k = 0; k < per; ++k)
    if (i0 + k < nblk) {
      const uint32_t t = a[i0 + k];
      a[i0 + k] = run;
      run += t;
    }
}

// Jaccard deltas, loss partials and the gradient scatter
__global__ __launch_bounds__(256) void lovasz_apply_kernel(const uint32_t* __restrict__ keys, const uint32_t* __restrict__ vals,
                                                           const uint32_t* __restrict__ boff, double* __restrict__ partial,
                                                           float* __restrict__ grad, long P, long HW, int nblk, float inv_n,
                                                           int xcd_affine) {
  __shared__ uint32_t sm[4];
  __shared__ double red[4];
  // The gradient scatter writes 4-byte values all over ONE image's 4 * P bytes (2 MB at 524 288 keys).  Blocks are dealt
  // round-robin over the 8 XCDs, each with an L2 of its own: with the natural (block, image) order every XCD holds a slice of
  // every cache line of the image.  When the image count divides by 8 and an image's gradient is at most half an L2 (2 MB: the
  // two-class 512^2 tiles of configs[2]), all blocks of an image run on ONE XCD instead (block L of the launch sits on XCD L % 8):
  // its L2 collects the image's lines whole before they leave -- lovasz_apply 217 -> 130 us at bs 32; at 4 MB per image (four
  // classes) the lines evict each other and the natural order is 3 % faster, so the rule stops there (profiles/r05/lovasz.txt).
  long n = blockIdx.y;
  int bx = blockIdx.x;
  if (xcd_affine) {
    const long L = (long)blockIdx.y * gridDim.x + blockIdx.x;
    const int x = (int)(L & 7);
    const long k = L >> 3;
    n = x + 8 * (k / nblk);
    bx = (int)(k % nblk);
  }
  const long r0 = (long)bx * kScanChunk + threadIdx.x * 4;
  uint32_t lab[4], key[4], idx[4];
  uint32_t s = 0;
  if ((P & 3) == 0 && r0 + 3 < P) {
    const lv_u32x4 v = *reinterpret_cast<const lv_u32x4*>(vals + n * P + r0), k = *reinterpret_cast<const lv_u32x4*>(keys + n * P + r0);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      lab[e] = v[e] >> 31;
      idx[e] = v[e] & 0x7fffffffu;
      key[e] = k[e];
      s += lab[e];
    }
  } else {
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      lab[e] = 0;
      key[e] = 0;
      idx[e] = 0;
      if (r0 + e < P) {
        const uint32_t v = vals[n * P + r0 + e];
        lab[e] = v >> 31;
        idx[e] = v & 0x7fffffffu;
        key[e] = keys[n * P + r0 + e];
      }
      s += lab[e];
    }
  }
  uint32_t total;
  uint32_t cs = block_exclusive_scan_256(s, sm, &total) + boff[n * nblk + bx];  // labels strictly before r0
  const float gts = (float)HW;
  double acc = 0;
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const long r = r0 + e;
    if (r < P) {
      const float cs_prev = (float)cs;
      cs += lab[e];
      const float cs_inc = (float)cs;
      // reference: inter = gts - cumsum(lab); union = gts + cumsum(1 - lab); iou = 1 - inter / union
      const float jac = 1.f - (gts - cs_inc) / (gts + ((float)(r + 1) - cs_inc));
      float delta = jac;
      if (r > 0) {
        const float jac_p = 1.f - (gts - cs_prev) / (gts + ((float)r - cs_prev));
        delta = jac - jac_p;
      }
      const float err = key_to_float(key[e]);
      const float pos = err > 0.f ? err : 0.f;
      acc += (double)(pos * delta);
      if (grad) {
        const float g = err > 0.f ? delta : 0.f;
        grad[n * P + idx[e]] = (lab[e] ? -g : g) * inv_n;
      }
    }
  }
  acc = rs_wave_sum(acc);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) partial[n * nblk + bx] = (red[0] + red[1]) + (red[2] + red[3]);
}

// (deterministic: a fixed element -> thread assignment and a fixed tree; 1024 threads with four independent accumulators each --
// round 4's 256 threads with one dependent chain took 41 us for the 16 384 partials of a bs-32 step)
__global__ __launch_bounds__(1024) void lovasz_finalize_kernel(const double* __restrict__ partial, long count, float inv_n,
                                                               float* __restrict__ loss) {
  __shared__ double red[1024];
  double a[4] = {0, 0, 0, 0};
  long i = threadIdx.x;
  for (; i + 3 * 1024 < count; i += 4 * 1024) {
    a[0] += partial[i];
    a[1] += partial[i + 1024];
    a[2] += partial[i + 2 * 1024];
    a[3] += partial[i + 3 * 1024];
  }
  for (; i < count; i += 1024) a[0] += partial[i];
  red[threadIdx.x] = (a[0] + a[1]) + (a[2] + a[3]);
  __syncthreads();
  for (int o = 512; o > 0; o >>= 1) {
    if (threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) loss[0] = (float)(red[0] * (double)inv_n);
}

__global__ void scale_by_scalar_kernel(const float* __restrict__ src, const float* __restrict__ scalar, float* __restrict__ dst,
                                       long n) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) dst[i] = src[i] * scalar[0];
}

struct Carve {
  long P;
  int nblk_sort, nblk_scan;
  size_t keys0, vals0, keys1, vals1, counts, bsum, partial, total;
};

Carve carve(int N, int C, int H, int W) {
  Carve c;
  c.P = (long)C * H * W;
  c.nblk_sort = rs_cdiv(c.P, sort_chunk((long)N * c.P));
  c.nblk_scan = rs_cdiv(c.P, kScanChunk);
  const size_t arr = ((size_t)N * c.P * sizeof(uint32_t) + 255) & ~(size_t)255;
  size_t o = 0;
  c.keys0 = o; o += arr;
  c.vals0 = o; o += arr;
  c.keys1 = o; o += arr;
  c.vals1 = o; o += arr;
  c.counts = o; o += (((size_t)N * c.nblk_sort * 256 * sizeof(uint32_t)) + 255) & ~(size_t)255;
  c.bsum = o; o += (((size_t)N * c.nblk_scan * sizeof(uint32_t)) + 255) & ~(size_t)255;
  c.partial = o; o += (((size_t)N * c.nblk_scan * sizeof(double)) + 255) & ~(size_t)255;
  c.total = o;
  return c;
}

}  // namespace

extern "C" long rs_lovasz_workspace_bytes(int N, int C, int H, int W) {
  if (N <= 0 || C <= 0 || H <= 0 || W <= 0 || (long)C * H * W >= (1L << 31)) return RS_EINVAL;
  return (long)carve(N, C, H, W).total;
}

extern "C" int rs_lovasz_fwd(const float* logits, const long long* targets, float* loss, float* grad_unit, int N, int C,
                             int H, int W, void* workspace, rs_stream_t stream) {
  if (!logits || !targets || !loss || !workspace || N <= 0 || C <= 0 || H <= 0 || W <= 0 || (long)C * H * W >= (1L << 31))
    return RS_EINVAL;
  const Carve cv = carve(N, C, H, W);
  char* ws = reinterpret_cast<char*>(workspace);
  uint32_t* k0 = reinterpret_cast<uint32_t*>(ws + cv.keys0);
  uint32_t* v0 = reinterpret_cast<uint32_t*>(ws + cv.vals0);
  uint32_t* k1 = reinterpret_cast<uint32_t*>(ws + cv.keys1);
  uint32_t* v1 = reinterpret_cast<uint32_t*>(ws + cv.vals1);
  uint32_t* counts = reinterpret_cast<uint32_t*>(ws + cv.counts);
  uint32_t* bsum = reinterpret_cast<uint32_t*>(ws + cv.bsum);
  double* partial = reinterpret_cast<double*>(ws + cv.partial);
  hipStream_t s = (hipStream_t)stream;
  const long P = cv.P, HW = (long)H * W;
  const bool small_chunk = sort_chunk((long)N * P) == 4096;
  // the sort's scatter passes keep an image's tiles on one XCD too (its digit runs of 64-128 bytes then meet their neighbours'
  // in one L2): -5 % on a bs-32 sort, -3.5 % with four classes, +3 % at 8 images (one image per XCD: no slack) -> from 16 images
  const int sc_affine = (N % 8 == 0 && N >= 16 && rs_knobs().lovasz_xcd != 0) ? 1 : 0;

  if ((HW & 3) == 0) lovasz_keys4_kernel<<<dim3(rs_cdiv(P / 4, 256), N), 256, 0, s>>>(logits, targets, k0, v0, P, HW);
  else lovasz_keys_kernel<<<dim3(rs_cdiv(P, 256), N), 256, 0, s>>>(logits, targets, k0, v0, P, HW);
  for (int pass = 0; pass < 4; ++pass) {
    const uint32_t* ik = (pass & 1) ? k1 : k0;
    const uint32_t* iv = (pass & 1) ? v1 : v0;
    uint32_t* ok = (pass & 1) ? k0 : k1;
    uint32_t* ov = (pass & 1) ? v0 : v1;
    if (small_chunk) radix_hist_kernel<4096><<<dim3(cv.nblk_sort, N), 256, 0, s>>>(ik, counts, P, cv.nblk_sort, pass * 8);
    else radix_hist_kernel<8192><<<dim3(cv.nblk_sort, N), 256, 0, s>>>(ik, counts, P, cv.nblk_sort, pass * 8);
    radix_scan_kernel<<<N, 1024, 0, s>>>(counts, cv.nblk_sort);
    if (small_chunk) radix_scatter_kernel<4096><<<dim3(cv.nblk_sort, N), 256, 0, s>>>(ik, iv, ok, ov, counts, P, cv.nblk_sort, pass * 8, sc_affine);
    else radix_scatter_kernel<8192><<<dim3(cv.nblk_sort, N), 256, 0, s>>>(ik, iv, ok, ov, counts, P, cv.nblk_sort, pass * 8, sc_affine);
  }
  // after 4 passes the sorted data is back in (k0, v0)
  lovasz_blocksum_kernel<<<dim3(cv.nblk_scan, N), 256, 0, s>>>(v0, bsum, P, cv.nblk_scan);
  lovasz_scan_blocks_kernel<<<N, 256, 0, s>>>(bsum, cv.nblk_scan);
  const float inv_n = 1.f / (float)N;
  const int xcd_affine = (N % 8 == 0 && P * 4 <= (2L << 20) && rs_knobs().lovasz_xcd != 0) ? 1 : 0;
  lovasz_apply_kernel<<<dim3(cv.nblk_scan, N), 256, 0, s>>>(k0, v0, bsum, partial, grad_unit, P, HW, cv.nblk_scan, inv_n, xcd_affine);
  lovasz_finalize_kernel<<<1, 1024, 0, s>>>(partial, (long)N * cv.nblk_scan, inv_n, loss);
  return RS_LAUNCH_RESULT();
}

extern "C" int rs_scale_by_scalar(const float* src, const float* scalar, float* dst, long n, rs_stream_t stream) {
  if (!src || !scalar || !dst || n <= 0) return RS_EINVAL;
  scale_by_scalar_kernel<<<rs_cdiv(n, 256), 256, 0, (hipStream_t)stream>>>(src, scalar, dst, n);
  return RS_LAUNCH_RESULT();
}

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

constexpr int kSortChunk = 8192;  // elements per 256-thread block: 4 waves x 32 batches of 64
constexpr int kScanChunk = 1024;  // elements per 256-thread block in the prefix-sum kernels

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

// counts[n][b][256]
__global__ __launch_bounds__(256) void radix_hist_kernel(const uint32_t* __restrict__ keys, uint32_t* __restrict__ counts,
                                                         long P, int nblk, int shift) {
  __shared__ uint32_t hist[256];
  const int tid = threadIdx.x;
  const long n = blockIdx.y;
  const int b = blockIdx.x;
  hist[tid] = 0;
  __syncthreads();
  const long base = (long)b * kSortChunk;
  for (int k = 0; k < kSortChunk / 256; ++k) {
    const long i = base + k * 256 + tid;
    if (i < P) atomicAdd(&hist[(keys[n * P + i] >> shift) & 255u], 1u);
  }
  __syncthreads();
  counts[(n * nblk + b) * 256 + tid] = hist[tid];
}

// exclusive scan of counts in (digit-major, block-minor) order, per image; in place: counts -> offsets
__global__ __launch_bounds__(256) void radix_scan_kernel(uint32_t* __restrict__ counts, int nblk) {
  __shared__ uint32_t tot[256];
  const int d = threadIdx.x;
  uint32_t* c = counts + (long)blockIdx.x * nblk * 256;
  uint32_t s = 0;
  for (int b = 0; b < nblk; ++b) s += c[(long)b * 256 + d];
  tot[d] = s;
  __syncthreads();
  if (d == 0) {
    uint32_t run = 0;
    for (int k = 0; k < 256; ++k) {
      const uint32_t t = tot[k];
      tot[k] = run;
      run += t;
    }
  }
  __syncthreads();
  uint32_t run = tot[d];
  for (int b = 0; b < nblk; ++b) {
    const uint32_t t = c[(long)b * 256 + d];
    c[(long)b * 256 + d] = run;
    run += t;
  }
}

__global__ __launch_bounds__(256) void radix_scatter_kernel(const uint32_t* __restrict__ keys, const uint32_t* __restrict__ vals,
                                                            uint32_t* __restrict__ okeys, uint32_t* __restrict__ ovals,
                                                            const uint32_t* __restrict__ offsets, long P, int nblk, int shift) {
  constexpr int WCH = kSortChunk / 4;  // elements per wave
  constexpr int NB = WCH / 64;         // batches of 64 per wave
  __shared__ uint32_t lk[kSortChunk];  // the tile, sorted by digit
  __shared__ uint32_t lv[kSortChunk];
  __shared__ uint32_t loc[4][256];     // per wave: histogram, then running position of each digit inside the tile
  __shared__ uint32_t gdelta[256];     // global position of a digit's run minus its position in the tile
  __shared__ uint32_t wsum[4];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const long n = blockIdx.y;
  const int b = blockIdx.x;
  const long start = (long)b * kSortChunk + wave * WCH;

  uint32_t key[NB], val[NB];
#pragma unroll
  for (int w = 0; w < 4; ++w) loc[w][tid] = 0;
  __syncthreads();
#pragma unroll
  for (int k = 0; k < NB; ++k) {
    const long i = start + k * 64 + lane;
    key[k] = 0xffffffffu;
    val[k] = 0;
    if (i < P) {
      key[k] = keys[n * P + i];
      val[k] = vals[n * P + i];
      atomicAdd(&loc[wave][(key[k] >> shift) & 255u], 1u);
    }
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
  const unsigned long long lt = (1ull << lane) - 1ull;
#pragma unroll
  for (int k = 0; k < NB; ++k) {  // fully unrolled: key[] / val[] stay in registers
    const bool valid = (start + k * 64 + lane) < P;
    const uint32_t d = (key[k] >> shift) & 255u;
    unsigned long long peers = __ballot(valid);
#pragma unroll
    for (int bit = 0; bit < 8; ++bit) {
      const bool one = (d >> bit) & 1u;
      const unsigned long long bm = __ballot(valid && one);
      peers &= one ? bm : ~bm;
    }
    // loc[wave][] is private to this wave and a wave's LDS operations execute in issue order: the leader's update below
    // is seen by the next batch's read without a block barrier
    uint32_t pos = 0;
    if (valid) pos = loc[wave][d] + (uint32_t)__popcll(peers & lt);
    __builtin_amdgcn_wave_barrier();
    if (valid && (__ffsll((long long)peers) - 1) == lane) loc[wave][d] += (uint32_t)__popcll(peers);
    __builtin_amdgcn_wave_barrier();
    if (valid) {
      lk[pos] = key[k];
      lv[pos] = val[k];
    }
  }
  __syncthreads();
  const long tile0 = (long)b * kSortChunk;
  const int count = (int)((P - tile0) < kSortChunk ? (P - tile0) : kSortChunk);
  for (int i = tid; i < count; i += 256) {
    const uint32_t kk = lk[i];
    const uint32_t g = gdelta[(kk >> shift) & 255u] + (uint32_t)i;
    okeys[n * P + g] = kk;
    ovals[n * P + g] = lv[i];
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
#pragma unroll
  for (int e = 0; e < 4; ++e)
    if (r0 + e < P) s += vals[n * P + r0 + e] >> 31;
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
                                                           float* __restrict__ grad, long P, long HW, int nblk, float inv_n) {
  __shared__ uint32_t sm[4];
  __shared__ double red[4];
  const long n = blockIdx.y;
  const long r0 = (long)blockIdx.x * kScanChunk + threadIdx.x * 4;
  uint32_t lab[4], key[4], idx[4];
  uint32_t s = 0;
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
  uint32_t total;
  uint32_t cs = block_exclusive_scan_256(s, sm, &total) + boff[n * nblk + blockIdx.x];  // labels strictly before r0
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
  if (threadIdx.x == 0) partial[n * nblk + blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}

__global__ __launch_bounds__(256) void lovasz_finalize_kernel(const double* __restrict__ partial, long count, float inv_n,
                                                              float* __restrict__ loss) {
  __shared__ double red[256];
  double s = 0;
  for (long i = threadIdx.x; i < count; i += 256) s += partial[i];
  red[threadIdx.x] = s;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
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
  c.nblk_sort = rs_cdiv(c.P, kSortChunk);
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

  lovasz_keys_kernel<<<dim3(rs_cdiv(P, 256), N), 256, 0, s>>>(logits, targets, k0, v0, P, HW);
  for (int pass = 0; pass < 4; ++pass) {
    const uint32_t* ik = (pass & 1) ? k1 : k0;
    const uint32_t* iv = (pass & 1) ? v1 : v0;
    uint32_t* ok = (pass & 1) ? k0 : k1;
    uint32_t* ov = (pass & 1) ? v0 : v1;
    radix_hist_kernel<<<dim3(cv.nblk_sort, N), 256, 0, s>>>(ik, counts, P, cv.nblk_sort, pass * 8);
    radix_scan_kernel<<<N, 256, 0, s>>>(counts, cv.nblk_sort);
    radix_scatter_kernel<<<dim3(cv.nblk_sort, N), 256, 0, s>>>(ik, iv, ok, ov, counts, P, cv.nblk_sort, pass * 8);
  }
  // after 4 passes the sorted data is back in (k0, v0)
  lovasz_blocksum_kernel<<<dim3(cv.nblk_scan, N), 256, 0, s>>>(v0, bsum, P, cv.nblk_scan);
  lovasz_scan_blocks_kernel<<<N, 256, 0, s>>>(bsum, cv.nblk_scan);
  const float inv_n = 1.f / (float)N;
  lovasz_apply_kernel<<<dim3(cv.nblk_scan, N), 256, 0, s>>>(k0, v0, bsum, partial, grad_unit, P, HW, cv.nblk_scan, inv_n);
  lovasz_finalize_kernel<<<1, 256, 0, s>>>(partial, (long)N * cv.nblk_scan, inv_n, loss);
  return RS_LAUNCH_RESULT();
}

extern "C" int rs_scale_by_scalar(const float* src, const float* scalar, float* dst, long n, rs_stream_t stream) {
  if (!src || !scalar || !dst || n <= 0) return RS_EINVAL;
  scale_by_scalar_kernel<<<rs_cdiv(n, 256), 256, 0, (hipStream_t)stream>>>(src, scalar, dst, n);
  return RS_LAUNCH_RESULT();
}

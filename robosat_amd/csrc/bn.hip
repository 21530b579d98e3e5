// Train-mode BatchNorm2d for NHWC activations viewed as an [M = N*H*W][C] matrix: per-channel column reductions and
// elementwise normalise / backward kernels.  HBM-bound: every kernel streams its operands once with 16-byte accesses
// (4 consecutive channels per thread, consecutive threads on consecutive channel quads => full 128-B segments).
//
// Reference semantics (torch.nn.BatchNorm2d as used by torchvision's ResNet-50 under robosat/unet.py:122-130, train
// mode in tools/train.py:169): normalise with the batch mean and BIASED variance, eps 1e-5; update running_mean /
// running_var with momentum 0.1 and the UNBIASED variance; num_batches_tracked += 1.
//
// Reductions: stage 1 writes per-row-split partial sums in fp64 (one thread accumulates its rows in fp64, the
// block combines its row lanes through LDS), stage 2 combines the splits -- deterministic, no atomics.
#include <atomic>

#include "common.h"

namespace {

constexpr int kMaxSplits = 512;

struct Geometry {
  int Q;     // channel quads per row
  int qb;    // quads per block
  int rpb;   // row lanes per block
  int gx;    // blocks along channels
  int R;     // row splits
  long rows_per_split;
};

Geometry geometry(long M, int C, int V = 4) {
  Geometry g;
  g.Q = C / V;
  g.qb = g.Q < 256 ? g.Q : 256;
  g.rpb = 256 / g.qb;
  g.gx = rs_cdiv(g.Q, g.qb);
  // aim at ~2048 blocks, at least 64 rows per row lane
  long want = 2048 / g.gx;
  if (want < 1) want = 1;
  long maxsplit = M / ((long)g.rpb * 64);
  if (maxsplit < 1) maxsplit = 1;
  long R = want < maxsplit ? want : maxsplit;
  if (R > kMaxSplits) R = kMaxSplits;
  g.rows_per_split = (M + R - 1) / R;
  g.R = rs_cdiv(M, g.rows_per_split);
  return g;
}

// MODE 0: s0 = sum y,  s1 = sum y^2                       (forward statistics)
// MODE 1: s0 = sum g,  s1 = sum g * (y - mean) * invstd     (backward; g = dz * (z > 0) when zmask != null)
// A thread owns V channels of the row lanes rl, rl + rpb, ...; U rows per trip: all loads of a trip are issued before the
// fp64 accumulation consumes them.
template <int MODE, typename T, int V>
__global__ __launch_bounds__(256) void bn_partial_kernel(const T* __restrict__ y, const T* __restrict__ dz,
                                                         const T* __restrict__ zmask, const float* __restrict__ mean,
                                                         const float* __restrict__ invstd, double* __restrict__ part,
                                                         long M, int C, int qb, int rpb, long rows_per_split) {
  __shared__ double red[256 * 2 * V];
  const int tid = threadIdx.x;
  const int Q = C / V;
  const int ql = tid % qb, rl = tid / qb;
  const int q = blockIdx.x * qb + ql;
  const long r0 = (long)blockIdx.y * rows_per_split;
  long r1 = r0 + rows_per_split;
  if (r1 > M) r1 = M;
  double s0[V], s1[V];
#pragma unroll
  for (int e = 0; e < V; ++e) s0[e] = 0, s1[e] = 0;
  const bool active = (q < Q) && (rl < rpb);
  if (active) {
    float mu[V], is[V];
#pragma unroll
    for (int e = 0; e < V; ++e) mu[e] = MODE == 1 ? mean[q * V + e] : 0.f, is[e] = MODE == 1 ? invstd[q * V + e] : 0.f;
    constexpr int U = V == 8 ? 2 : 4;
    const long step = (long)rpb * U;
    for (long rb = r0 + rl; rb < r1; rb += step) {
      rs_vecf<V> v[U], g[U], z[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const long r = rb + (long)u * rpb;
        const bool in = r < r1;
        const long o = (in ? r : rb) * C + q * V;
        v[u] = rs_ldv<V>(y + o);
        if (MODE == 1) {
          g[u] = rs_ldv<V>(dz + o);
          if (zmask) z[u] = rs_ldv<V>(zmask + o);
        }
        if (!in) {
#pragma unroll
          for (int e = 0; e < V; ++e) v[u].v[e] = 0.f, g[u].v[e] = 0.f;
        }
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        if (MODE == 0) {
#pragma unroll
          for (int e = 0; e < V; ++e) {
            s0[e] += (double)v[u].v[e];
            s1[e] += (double)v[u].v[e] * (double)v[u].v[e];
          }
        } else {
          if (zmask) {
#pragma unroll
            for (int e = 0; e < V; ++e) g[u].v[e] = z[u].v[e] > 0.f ? g[u].v[e] : 0.f;
          }
#pragma unroll
          for (int e = 0; e < V; ++e) {
            s0[e] += (double)g[u].v[e];
            s1[e] += (double)g[u].v[e] * (double)((v[u].v[e] - mu[e]) * is[e]);
          }
        }
      }
    }
  }
  // (component-major: red[component][thread] -- consecutive threads write consecutive 8-byte words; the thread-major layout
  // of rounds 1-4 put all 64 lanes of a store on the same two banks: 76 % of this kernel's LDS cycles were conflicts)
#pragma unroll
  for (int e = 0; e < V; ++e) {
    red[e * 256 + tid] = s0[e];
    red[(V + e) * 256 + tid] = s1[e];
  }
  __syncthreads();
  if (active && rl == 0) {
    for (int g = 1; g < rpb; ++g) {
      const double* o = red + g * qb + ql;
#pragma unroll
      for (int e = 0; e < V; ++e) {
        s0[e] += o[e * 256];
        s1[e] += o[(V + e) * 256];
      }
    }
    double* p0 = part + ((long)blockIdx.y * 2) * C + q * V;
    double* p1 = p0 + C;
#pragma unroll
    for (int e = 0; e < V; ++e) {
      p0[e] = s0[e];
      p1[e] = s1[e];
    }
  }
}

// Stage 2: 16 channels x 16 split lanes per block; lane l sums splits l, l+16, ... in fp64, LDS tree over the lanes.
// SC1: the rows were published inside this launch by blocks on other XCDs (sc1 write-through stores): read them with
// agent-scope loads, which bypass this XCD's non-coherent L2 lines.
template <typename P, bool SC1 = false>
__device__ __forceinline__ void bn_reduce_splits(const P* __restrict__ part, int R, int C, int c, int lane, bool ok,
                                                 double& s0, double& s1) {
  __shared__ double red[2][16][17];
  double a = 0, b = 0;
  if (ok) {
    for (int r = lane; r < R; r += 16) {
      if (SC1) {
        a += (double)__hip_atomic_load(part + ((long)r * 2) * C + c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        b += (double)__hip_atomic_load(part + ((long)r * 2 + 1) * C + c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      } else {
        a += (double)part[((long)r * 2) * C + c];
        b += (double)part[((long)r * 2 + 1) * C + c];
      }
    }
  }
  const int cl = threadIdx.x & 15;
  red[0][lane][cl] = a;
  red[1][lane][cl] = b;
  __syncthreads();
  s0 = 0;
  s1 = 0;
  if (lane == 0) {
#pragma unroll
    for (int l = 0; l < 16; ++l) {
      s0 += red[0][l][cl];
      s1 += red[1][l][cl];
    }
  }
}

struct BnStatsOut {
  float eps, momentum;
  const float *gamma, *beta;
  float *mean, *invstd, *scale, *shift, *running_mean, *running_var;
  long long* num_batches_tracked;
};

// channel group `cg` (16 channels) of the forward finalize, by one whole block: R partial rows [r][2][C] -> statistics
template <typename P, bool SC1 = false>
__device__ __forceinline__ void bn_stats_finalize_body(const P* __restrict__ part, int R, long M, int C, int cg,
                                                       const BnStatsOut& o) {
  const int c = cg * 16 + (threadIdx.x & 15), lane = threadIdx.x >> 4;
  if (cg == 0 && threadIdx.x == 0 && o.num_batches_tracked) *o.num_batches_tracked += 1;
  double s, ss;
  bn_reduce_splits<P, SC1>(part, R, C, c, lane, c < C, s, ss);
  if (c >= C || lane != 0) return;
  const double mu = s / (double)M;
  double var = ss / (double)M - mu * mu;
  if (var < 0) var = 0;
  const float is = (float)(1.0 / sqrt(var + (double)o.eps));
  o.mean[c] = (float)mu;
  o.invstd[c] = is;
  const float sc = o.gamma[c] * is;
  o.scale[c] = sc;
  o.shift[c] = o.beta[c] - (float)mu * sc;
  if (o.running_mean) {
    const double unbiased = M > 1 ? var * ((double)M / (double)(M - 1)) : var;
    o.running_mean[c] = (1.f - o.momentum) * o.running_mean[c] + o.momentum * (float)mu;
    o.running_var[c] = (1.f - o.momentum) * o.running_var[c] + o.momentum * (float)unbiased;
  }
}

template <typename P>
__global__ __launch_bounds__(256) void bn_stats_finalize_kernel(const P* __restrict__ part, int R, long M, int C, BnStatsOut o) {
  bn_stats_finalize_body(part, R, M, C, blockIdx.x, o);
}

// coef[0][c] = k1 = gamma*invstd, coef[1][c] = k2 = k1*sum(g)/M, coef[2][c] = k3 = k1*invstd*sum(g*xhat)/M
struct BnBwdOut {
  const float *gamma, *invstd;
  float *dgamma, *dbeta, *coef;
};

template <bool SC1 = false>
__device__ __forceinline__ void bn_bwd_finalize_body(const double* __restrict__ part, int R, long M, int C, int cg,
                                                     const BnBwdOut& o) {
  const int c = cg * 16 + (threadIdx.x & 15), lane = threadIdx.x >> 4;
  double s0, s1;
  bn_reduce_splits<double, SC1>(part, R, C, c, lane, c < C, s0, s1);
  if (c >= C || lane != 0) return;
  o.dbeta[c] = (float)s0;
  o.dgamma[c] = (float)s1;
  const double k1 = (double)o.gamma[c] * (double)o.invstd[c];
  o.coef[c] = (float)k1;
  o.coef[C + c] = (float)(k1 * s0 / (double)M);
  o.coef[2 * C + c] = (float)(k1 * (double)o.invstd[c] * s1 / (double)M);
}

__global__ __launch_bounds__(256) void bn_bwd_finalize_kernel(const double* __restrict__ part, int R, long M, int C, BnBwdOut o) {
  bn_bwd_finalize_body(part, R, M, C, blockIdx.x, o);
}

// out = relu?( y * scale[c] + shift[c] (+ residual) )
template <typename T>
__global__ void bn_apply_kernel(const T* __restrict__ y, const float* __restrict__ scale, const float* __restrict__ shift,
                                const T* __restrict__ res, T* __restrict__ out, long total4, int Q, int relu) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total4) return;
  const int q = (int)(i % Q);
  f32x4 v = rs_ld4(y + i * 4);
  const f32x4 sc = *reinterpret_cast<const f32x4*>(scale + q * 4);
  const f32x4 sh = *reinterpret_cast<const f32x4*>(shift + q * 4);
  f32x4 r = {0, 0, 0, 0};
  if (res) r = rs_ld4(res + i * 4);
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    float t = v[e] * sc[e] + sh[e] + r[e];
    v[e] = relu ? fmaxf(t, 0.f) : t;
  }
  rs_st4(out + i * 4, v);
}

// dy = k1*g - k2 - k3*(y - mean),  g = dz*(z>0) (or dz);  optionally dmasked = g (gradient of the residual branch)
template <typename T>
__global__ void bn_bwd_apply_kernel(const T* __restrict__ dz, const T* __restrict__ zmask, const T* __restrict__ y,
                                    const float* __restrict__ mean, const float* __restrict__ coef, T* __restrict__ dy,
                                    T* __restrict__ dmasked, long total4, int Q) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total4) return;
  const int q = (int)(i % Q);
  const int C = Q * 4;
  f32x4 g = rs_ld4(dz + i * 4);
  if (zmask) {
    const f32x4 z = rs_ld4(zmask + i * 4);
#pragma unroll
    for (int e = 0; e < 4; ++e) g[e] = z[e] > 0.f ? g[e] : 0.f;
  }
  const f32x4 v = rs_ld4(y + i * 4);
  const f32x4 mu = *reinterpret_cast<const f32x4*>(mean + q * 4);
  const f32x4 k1 = *reinterpret_cast<const f32x4*>(coef + q * 4);
  const f32x4 k2 = *reinterpret_cast<const f32x4*>(coef + C + q * 4);
  const f32x4 k3 = *reinterpret_cast<const f32x4*>(coef + 2 * C + q * 4);
  f32x4 o;
#pragma unroll
  for (int e = 0; e < 4; ++e) o[e] = k1[e] * g[e] - k2[e] - k3[e] * (v[e] - mu[e]);
  rs_st4(dy + i * 4, o);
  if (dmasked) rs_st4(dmasked + i * 4, g);
}

// ---- streaming forms of the two apply passes (round 2) -------------------------------------------------------------------
// One thread = 8 consecutive channels (one 16-byte access in bf16, two in fp32), a block-iteration = 2048 consecutive
// elements.  When C divides 2048 a thread meets the SAME 8 channels in every iteration, so its per-channel coefficients
// live in registers for the whole launch; a block issues the loads of kBnIter iterations before the first use (4-8 requests
// of 16 bytes in flight per lane instead of 1-2 of 8 bytes).  Other channel counts use the 4-channel kernels above.
constexpr int kBnIter = 4;

struct BnRawB { bf16x8 v; };
struct BnRawF { f32x4 a, b; };
__device__ __forceinline__ BnRawB bn_ld8(const bf16_t* p) { return {__builtin_nontemporal_load(reinterpret_cast<const bf16x8*>(p))}; }
__device__ __forceinline__ BnRawF bn_ld8(const float* p) {
  return {__builtin_nontemporal_load(reinterpret_cast<const f32x4*>(p)), __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(p) + 1)};
}
__device__ __forceinline__ float bn_get(const BnRawB& r, int e) { return (float)r.v[e]; }
__device__ __forceinline__ float bn_get(const BnRawF& r, int e) { return e < 4 ? r.a[e] : r.b[e - 4]; }
__device__ __forceinline__ void bn_st8(bf16_t* p, const float (&v)[8]) {
  bf16x8 t;
#pragma unroll
  for (int e = 0; e < 8; ++e) t[e] = (bf16_t)v[e];
  *reinterpret_cast<bf16x8*>(p) = t;
}
__device__ __forceinline__ void bn_st8(float* p, const float (&v)[8]) {
  f32x4 a, b;
#pragma unroll
  for (int e = 0; e < 4; ++e) a[e] = v[e], b[e] = v[4 + e];
  *reinterpret_cast<f32x4*>(p) = a;
  *(reinterpret_cast<f32x4*>(p) + 1) = b;
}

template <typename T, bool RES>
__global__ __launch_bounds__(256) void bn_apply_stream_kernel(const T* __restrict__ y, const float* __restrict__ scale,
                                                              const float* __restrict__ shift, const T* __restrict__ res,
                                                              T* __restrict__ out, unsigned char* __restrict__ bits, long total,
                                                              int C, int relu) {
  const int c0 = (threadIdx.x * 8) & (C - 1);  // (C divides 2048: a power of two)
  float sc[8], sh[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) sc[e] = scale[c0 + e], sh[e] = shift[c0 + e];
  const long base = (long)blockIdx.x * (kBnIter * 2048) + threadIdx.x * 8;
  decltype(bn_ld8(y)) v[kBnIter], r[kBnIter];
#pragma unroll
  for (int it = 0; it < kBnIter; ++it) {
    const long i = base + it * 2048;
    if (i < total) {
      v[it] = bn_ld8(y + i);
      if (RES) r[it] = bn_ld8(res + i);
    }
  }
#pragma unroll
  for (int it = 0; it < kBnIter; ++it) {
    const long i = base + it * 2048;
    if (i < total) {
      float o[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        float t = bn_get(v[it], e) * sc[e] + sh[e];
        if (RES) t += bn_get(r[it], e);
        o[e] = relu ? fmaxf(t, 0.f) : t;
      }
      bn_st8(out + i, o);
      if (bits) {  // the ReLU mask the backward needs, one bit per element: (stored value > 0)
        unsigned int b = 0;
#pragma unroll
        for (int e = 0; e < 8; ++e) b |= ((float)(T)o[e] > 0.f ? 1u : 0u) << e;
        bits[i >> 3] = (unsigned char)b;
      }
    }
  }
}

// MASK: g = dz * (z > 0) first;  DM: that g is also stored (the gradient of the residual branch)
template <typename T, bool MASK, bool DM>
__global__ __launch_bounds__(256) void bn_bwd_apply_stream_kernel(const T* __restrict__ g, const T* __restrict__ zmask,
                                                                  const T* __restrict__ y, const float* __restrict__ mean,
                                                                  const float* __restrict__ coef, T* __restrict__ dy,
                                                                  T* __restrict__ dmasked, long total, int C) {
  constexpr int IT = MASK ? 2 : kBnIter;  // (three operands per iteration when masked: same number of loads in flight)
  const int c0 = (threadIdx.x * 8) & (C - 1);
  float mu[8], k1[8], k2[8], k3[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) mu[e] = mean[c0 + e], k1[e] = coef[c0 + e], k2[e] = coef[C + c0 + e], k3[e] = coef[2 * C + c0 + e];
  const long base = (long)blockIdx.x * (IT * 2048) + threadIdx.x * 8;
  decltype(bn_ld8(y)) gv[IT], yv[IT], zv[IT];
#pragma unroll
  for (int it = 0; it < IT; ++it) {
    const long i = base + it * 2048;
    if (i < total) {
      gv[it] = bn_ld8(g + i);
      yv[it] = bn_ld8(y + i);
      if (MASK) zv[it] = bn_ld8(zmask + i);
    }
  }
#pragma unroll
  for (int it = 0; it < IT; ++it) {
    const long i = base + it * 2048;
    if (i < total) {
      float o[8], gm[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        gm[e] = bn_get(gv[it], e);
        if (MASK) gm[e] = bn_get(zv[it], e) > 0.f ? gm[e] : 0.f;
        o[e] = k1[e] * gm[e] - k2[e] - k3[e] * (bn_get(yv[it], e) - mu[e]);
      }
      bn_st8(dy + i, o);
      if (DM) bn_st8(dmasked + i, gm);
    }
  }
}

inline bool bn_streamable(long M, int C) {
  return C >= 8 && C <= 2048 && (2048 % C) == 0 && ((M * C) % 8) == 0;
}

template <typename T>
int bn_apply_launch(const T* y, const float* scale, const float* shift, const T* res, T* out, unsigned char* bits, long M, int C,
                    int relu, hipStream_t s) {
  const long total = M * C;
  if (bn_streamable(M, C)) {
    const int grid = rs_cdiv(total, kBnIter * 2048L);
    if (res)
      bn_apply_stream_kernel<T, true><<<grid, 256, 0, s>>>(y, scale, shift, res, out, bits, total, C, relu);
    else
      bn_apply_stream_kernel<T, false><<<grid, 256, 0, s>>>(y, scale, shift, res, out, bits, total, C, relu);
    return RS_LAUNCH_RESULT();
  }
  if (bits) return RS_EINVAL;  // (the bit mask comes from the streaming form only)
  bn_apply_kernel<T><<<rs_cdiv(total / 4, 256), 256, 0, s>>>(y, scale, shift, res, out, total / 4, C / 4, relu);
  return RS_LAUNCH_RESULT();
}

template <typename T>
void bn_bwd_apply_launch(const T* g, const T* zmask, const T* y, const float* mean, const float* coef, T* dy, T* dmasked, long M,
                         int C, hipStream_t s) {
  const long total = M * C;
  if (bn_streamable(M, C)) {
    const int grid = rs_cdiv(total, (zmask ? 2 : kBnIter) * 2048L);
    if (zmask && dmasked)
      bn_bwd_apply_stream_kernel<T, true, true><<<grid, 256, 0, s>>>(g, zmask, y, mean, coef, dy, dmasked, total, C);
    else if (zmask)
      bn_bwd_apply_stream_kernel<T, true, false><<<grid, 256, 0, s>>>(g, zmask, y, mean, coef, dy, nullptr, total, C);
    else if (dmasked)
      bn_bwd_apply_stream_kernel<T, false, true><<<grid, 256, 0, s>>>(g, nullptr, y, mean, coef, dy, dmasked, total, C);
    else
      bn_bwd_apply_stream_kernel<T, false, false><<<grid, 256, 0, s>>>(g, nullptr, y, mean, coef, dy, nullptr, total, C);
    return;
  }
  bn_bwd_apply_kernel<T><<<rs_cdiv(total / 4, 256), 256, 0, s>>>(g, zmask, y, mean, coef, dy, dmasked, total / 4, C / 4);
}

template <typename T>
int bn_train_stats_t(const T* y, long M, int C, float eps, float momentum, const float* gamma, const float* beta, float* mean,
                     float* invstd, float* scale, float* shift, float* running_mean, float* running_var,
                     long long* num_batches_tracked, void* workspace, hipStream_t s) {
  const int V = (C % 8) == 0 ? 8 : 4;
  const Geometry g = geometry(M, C, V);
  double* part = reinterpret_cast<double*>(workspace);
  if (V == 8)
    bn_partial_kernel<0, T, 8><<<dim3(g.gx, g.R), 256, 0, s>>>(y, nullptr, nullptr, nullptr, nullptr, part, M, C, g.qb, g.rpb,
                                                               g.rows_per_split);
  else
    bn_partial_kernel<0, T, 4><<<dim3(g.gx, g.R), 256, 0, s>>>(y, nullptr, nullptr, nullptr, nullptr, part, M, C, g.qb, g.rpb,
                                                               g.rows_per_split);
  const BnStatsOut o = {eps, momentum, gamma, beta, mean, invstd, scale, shift, running_mean, running_var, num_batches_tracked};
  bn_stats_finalize_kernel<double><<<rs_cdiv(C, 16), 256, 0, s>>>(part, g.R, M, C, o);
  return RS_LAUNCH_RESULT();
}

template <typename T>
int bn_bwd_t(const T* dz, const T* zmask, const T* y, const float* mean, const float* invstd, const float* gamma, T* dy,
             T* dmasked, float* dgamma, float* dbeta, long M, int C, void* workspace, hipStream_t s) {
  const int V = (C % 8) == 0 ? 8 : 4;
  const Geometry g = geometry(M, C, V);
  double* part = reinterpret_cast<double*>(workspace);
  float* coef = reinterpret_cast<float*>(part + (long)g.R * 2 * C);  // 3*C floats behind the partials
  if (V == 8)
    bn_partial_kernel<1, T, 8><<<dim3(g.gx, g.R), 256, 0, s>>>(y, dz, zmask, mean, invstd, part, M, C, g.qb, g.rpb,
                                                               g.rows_per_split);
  else
    bn_partial_kernel<1, T, 4><<<dim3(g.gx, g.R), 256, 0, s>>>(y, dz, zmask, mean, invstd, part, M, C, g.qb, g.rpb,
                                                               g.rows_per_split);
  const BnBwdOut o = {gamma, invstd, dgamma, dbeta, coef};
  bn_bwd_finalize_kernel<<<rs_cdiv(C, 16), 256, 0, s>>>(part, g.R, M, C, o);
  bn_bwd_apply_launch(dz, zmask, y, mean, coef, dy, dmasked, M, C, s);
  return RS_LAUNCH_RESULT();
}

}  // namespace

extern "C" int rs_bn_train_stats_dt(const void* y, int dtype, long M, int C, float eps, float momentum, const float* gamma,
                                    const float* beta, float* mean, float* invstd, float* scale, float* shift,
                                    float* running_mean, float* running_var, long long* num_batches_tracked,
                                    void* workspace, rs_stream_t stream) {
  if (!y || !gamma || !beta || !mean || !invstd || !scale || !shift || !workspace || M <= 0 || C <= 0 || (C & 3))
    return RS_EINVAL;
  if ((running_mean == nullptr) != (running_var == nullptr)) return RS_EINVAL;
  hipStream_t s = (hipStream_t)stream;
  if (dtype == RS_F32)
    return bn_train_stats_t(reinterpret_cast<const float*>(y), M, C, eps, momentum, gamma, beta, mean, invstd, scale, shift,
                            running_mean, running_var, num_batches_tracked, workspace, s);
  if (dtype == RS_BF16)
    return bn_train_stats_t(reinterpret_cast<const bf16_t*>(y), M, C, eps, momentum, gamma, beta, mean, invstd, scale, shift,
                            running_mean, running_var, num_batches_tracked, workspace, s);
  return RS_EINVAL;
}

extern "C" int rs_bn_train_stats(const float* y, long M, int C, float eps, float momentum, const float* gamma,
                                 const float* beta, float* mean, float* invstd, float* scale, float* shift,
                                 float* running_mean, float* running_var, long long* num_batches_tracked, void* workspace,
                                 rs_stream_t stream) {
  return rs_bn_train_stats_dt(y, RS_F32, M, C, eps, momentum, gamma, beta, mean, invstd, scale, shift, running_mean,
                              running_var, num_batches_tracked, workspace, stream);
}

// Many conv-epilogue partial rows -> statistics in ONE launch (round 2: was reduce + finalize, 71 + 71 launches of ~5 us on
// the critical path of a step).  Level 1: block (cg, sl) sums slice sl of the rows for channel group cg into one fp64 row
// [sl][2][C].  The block that arrives LAST at its channel group's counter then sums the <= 64 slice rows in slice order
// (deterministic, whoever it is) and finalizes.  Counters: zero at module load, every launch leaves its set at zero; 16 sets in
// rotation so that launches in flight on different streams never share one.
__device__ unsigned int g_bn_arrivals[16 * 128];

struct BnFinOut {
  BnStatsOut st;
  BnBwdOut bw;
};

template <int MODE>  // 0: forward statistics, 1: backward coefficients
__global__ __launch_bounds__(256) void bn_reduce_finalize_kernel(const float* __restrict__ part, int R, int C, int rows_per_slice,
                                                                 double* __restrict__ out, long M, int set, BnFinOut o) {
  __shared__ int is_last;
  const int c = blockIdx.x * 16 + (threadIdx.x & 15), lane = threadIdx.x >> 4;
  const int r0 = blockIdx.y * rows_per_slice;
  int r1 = r0 + rows_per_slice;
  if (r1 > R) r1 = R;
  double s0, s1;
  bn_reduce_splits(part + (long)r0 * 2 * C, r1 - r0, C, c, lane, c < C, s0, s1);
  // Publishing a slice row across XCDs without flushing an L2 that is full of the producing convolution's output: the row goes
  // out as write-through (sc1) 8-byte stores, every wave waits for its stores, then ONE relaxed device-scope arrival per block;
  // the last arriver reads the rows back with sc1 loads (platform guide, in-launch split-K reduction recipe).
  if (c < C && lane == 0) {
    __hip_atomic_store(out + ((long)blockIdx.y * 2) * C + c, s0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(out + ((long)blockIdx.y * 2 + 1) * C + c, s1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  if (set < 0) return;  // (more channel groups than counters: the host launches the finalize kernel behind this one)
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (threadIdx.x == 0) {
    unsigned int* ctr = g_bn_arrivals + set * 128 + blockIdx.x;
    const unsigned int before = __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    is_last = before == gridDim.y - 1;
    if (is_last) __hip_atomic_store(ctr, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // (all arrivals of this launch have happened)
  }
  __syncthreads();
  if (!is_last) return;
  if (MODE == 0)
    bn_stats_finalize_body<double, true>(out, (int)gridDim.y, M, C, blockIdx.x, o.st);
  else
    bn_bwd_finalize_body<true>(out, (int)gridDim.y, M, C, blockIdx.x, o.bw);
}

static int bn_next_counter_set() {
  static std::atomic<unsigned int> seq{0};
  return (int)(seq.fetch_add(1) & 15u);
}

// Finalize from the per-tile fp32 partial sums the convolution epilogue wrote (rs_conv2d_fwd_bnstats_dt): replaces the
// separate read pass of rs_bn_train_stats.
extern "C" int rs_bn_finalize_stats(const float* partial, long rows, long M, int C, float eps, float momentum,
                                    const float* gamma, const float* beta, float* mean, float* invstd, float* scale,
                                    float* shift, float* running_mean, float* running_var, long long* num_batches_tracked,
                                    void* workspace, rs_stream_t stream) {
  if (!partial || rows <= 0 || rows >= (1L << 31) || !gamma || !beta || !mean || !invstd || !scale || !shift || M <= 0 ||
      C <= 0)
    return RS_EINVAL;
  if ((running_mean == nullptr) != (running_var == nullptr)) return RS_EINVAL;
  hipStream_t s = (hipStream_t)stream;
  if (rows > 256 && workspace) {  // two levels: <= 64 slices of the rows in parallel, then the usual fp64 finalize
    int slices = (int)((rows + 63) / 64);
    if (slices > 64) slices = 64;
    const int rps = (int)((rows + slices - 1) / slices);
    slices = (int)((rows + rps - 1) / rps);
    double* part2 = reinterpret_cast<double*>(workspace);
    BnFinOut o = {};
    o.st = {eps, momentum, gamma, beta, mean, invstd, scale, shift, running_mean, running_var, num_batches_tracked};
    const bool merged = C <= 2048;
    bn_reduce_finalize_kernel<0><<<dim3(rs_cdiv(C, 16), slices), 256, 0, s>>>(partial, (int)rows, C, rps, part2, M,
                                                                              merged ? bn_next_counter_set() : -1, o);
    if (!merged) bn_stats_finalize_kernel<double><<<rs_cdiv(C, 16), 256, 0, s>>>(part2, slices, M, C, o.st);
    return RS_LAUNCH_RESULT();
  }
  const BnStatsOut o = {eps, momentum, gamma, beta, mean, invstd, scale, shift, running_mean, running_var, num_batches_tracked};
  bn_stats_finalize_kernel<float><<<rs_cdiv(C, 16), 256, 0, s>>>(partial, (int)rows, M, C, o);
  return RS_LAUNCH_RESULT();
}

// rs_bn_apply_dt that also writes the ReLU mask of `out` as one bit per element (`bits`: M*C/8 bytes; bit e of byte i = element
// 8*i + e is > 0) -- what the data-gradient epilogues read instead of `out` itself (rs_conv2d_dgrad_bnstats_bits_dt).  Needs a
// channel count that divides 2048 (every BatchNorm of the ResNet-50 encoder); RS_EINVAL otherwise.
extern "C" int rs_bn_apply_bits_dt(const void* y, const float* scale, const float* shift, const void* residual, void* out,
                                   unsigned char* bits, int dtype, long M, int C, int relu, rs_stream_t stream) {
  if (!y || !scale || !shift || !out || M <= 0 || C <= 0 || (C & 3)) return RS_EINVAL;
  hipStream_t s = (hipStream_t)stream;
  if (dtype == RS_F32)
    return bn_apply_launch(reinterpret_cast<const float*>(y), scale, shift, reinterpret_cast<const float*>(residual),
                           reinterpret_cast<float*>(out), bits, M, C, relu, s);
  if (dtype == RS_BF16)
    return bn_apply_launch(reinterpret_cast<const bf16_t*>(y), scale, shift, reinterpret_cast<const bf16_t*>(residual),
                           reinterpret_cast<bf16_t*>(out), bits, M, C, relu, s);
  return RS_EINVAL;
}

extern "C" int rs_bn_apply_dt(const void* y, const float* scale, const float* shift, const void* residual, void* out,
                              int dtype, long M, int C, int relu, rs_stream_t stream) {
  return rs_bn_apply_bits_dt(y, scale, shift, residual, out, nullptr, dtype, M, C, relu, stream);
}

extern "C" int rs_bn_apply(const float* y, const float* scale, const float* shift, const float* residual, float* out,
                           long M, int C, int relu, rs_stream_t stream) {
  return rs_bn_apply_dt(y, scale, shift, residual, out, RS_F32, M, C, relu, stream);
}

extern "C" int rs_bn_bwd_dt(const void* dz, const void* zmask, const void* y, const float* mean, const float* invstd,
                            const float* gamma, void* dy, void* dmasked, float* dgamma, float* dbeta, int dtype, long M,
                            int C, void* workspace, rs_stream_t stream) {
  if (!dz || !y || !mean || !invstd || !gamma || !dy || !dgamma || !dbeta || !workspace || M <= 0 || C <= 0 || (C & 3))
    return RS_EINVAL;
  hipStream_t s = (hipStream_t)stream;
  if (dtype == RS_F32)
    return bn_bwd_t(reinterpret_cast<const float*>(dz), reinterpret_cast<const float*>(zmask),
                    reinterpret_cast<const float*>(y), mean, invstd, gamma, reinterpret_cast<float*>(dy),
                    reinterpret_cast<float*>(dmasked), dgamma, dbeta, M, C, workspace, s);
  if (dtype == RS_BF16)
    return bn_bwd_t(reinterpret_cast<const bf16_t*>(dz), reinterpret_cast<const bf16_t*>(zmask),
                    reinterpret_cast<const bf16_t*>(y), mean, invstd, gamma, reinterpret_cast<bf16_t*>(dy),
                    reinterpret_cast<bf16_t*>(dmasked), dgamma, dbeta, M, C, workspace, s);
  return RS_EINVAL;
}

// BatchNorm backward whose two reductions were already done by the data-gradient convolution that produced g
// (rs_conv2d_dgrad_bnstats_dt): finalize the per-tile partials, then ONE streaming pass dy = k1*g - k2 - k3*(y - mean).
extern "C" int rs_bn_bwd_from_partials_dt(const void* g, const void* y, const float* mean, const float* invstd,
                                          const float* gamma, void* dy, float* dgamma, float* dbeta, const float* partial,
                                          long rows, int dtype, long M, int C, void* workspace, rs_stream_t stream) {
  if (!g || !y || !mean || !invstd || !gamma || !dy || !dgamma || !dbeta || !partial || !workspace || rows <= 0 ||
      rows >= (1L << 31) || M <= 0 || C <= 0 || (C & 3))
    return RS_EINVAL;
  hipStream_t s = (hipStream_t)stream;
  // workspace: [64][2][C] doubles (first-level sums) + 3*C floats (coefficients)
  double* part2 = reinterpret_cast<double*>(workspace);
  float* coef = reinterpret_cast<float*>(part2 + 64L * 2 * C);
  int slices = (int)((rows + 63) / 64);
  if (slices > 64) slices = 64;
  const int rps = (int)((rows + slices - 1) / slices);
  slices = (int)((rows + rps - 1) / rps);
  BnFinOut o = {};
  o.bw = {gamma, invstd, dgamma, dbeta, coef};
  const bool merged = C <= 2048;
  bn_reduce_finalize_kernel<1><<<dim3(rs_cdiv(C, 16), slices), 256, 0, s>>>(partial, (int)rows, C, rps, part2, M,
                                                                            merged ? bn_next_counter_set() : -1, o);
  if (!merged) bn_bwd_finalize_kernel<<<rs_cdiv(C, 16), 256, 0, s>>>(part2, slices, M, C, o.bw);
  if (dtype == RS_F32)
    bn_bwd_apply_launch<float>(reinterpret_cast<const float*>(g), nullptr, reinterpret_cast<const float*>(y), mean, coef,
                               reinterpret_cast<float*>(dy), nullptr, M, C, s);
  else if (dtype == RS_BF16)
    bn_bwd_apply_launch<bf16_t>(reinterpret_cast<const bf16_t*>(g), nullptr, reinterpret_cast<const bf16_t*>(y), mean, coef,
                                reinterpret_cast<bf16_t*>(dy), nullptr, M, C, s);
  else
    return RS_EINVAL;
  return RS_LAUNCH_RESULT();
}

extern "C" int rs_bn_bwd(const float* dz, const float* zmask, const float* y, const float* mean, const float* invstd,
                         const float* gamma, float* dy, float* dmasked, float* dgamma, float* dbeta, long M, int C,
                         void* workspace, rs_stream_t stream) {
  return rs_bn_bwd_dt(dz, zmask, y, mean, invstd, gamma, dy, dmasked, dgamma, dbeta, RS_F32, M, C, workspace, stream);
}

extern "C" long rs_bn_workspace_bytes(long M, int C) {
  if (M <= 0 || C <= 0 || (C & 3)) return RS_EINVAL;
  const Geometry g4 = geometry(M, C, 4), g8 = geometry(M, C, 8);  // (either vector width may run)
  return (long)(g4.R > g8.R ? g4.R : g8.R) * 2 * C * (long)sizeof(double);
}

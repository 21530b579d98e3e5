// Per-pixel classification losses and the confusion counts of the training loop, on NCHW logits [N][C][H][W] and
// int64 label maps [N][H][W] -- the tensors the reference's criteria receive (robosat/tools/train.py:185,226).
//
//   CrossEntropyLoss2d (robosat/losses.py:8-25):  NLLLoss(weight)(log_softmax(x, 1), t)
//       = sum_p w[t_p] * (lse_p - x[t_p][p]) / sum_p w[t_p]
//   FocalLoss2d (robosat/losses.py:28-50), gamma = 2 by default:
//       = sum_p w[t_p] * ( -(1 - s_t)^gamma * log s_t ) / sum_p w[t_p],   s = softmax(x, 1)
//   Metrics.add (robosat/metrics.py:27-41): argmax over classes, then the reference's pred/actual quotient trick.
//
// HBM-bound: one thread per pixel, plane-wise coalesced reads; reductions in fp64, two deterministic stages.
#include "common.h"

namespace {

constexpr int kMaxC = 8;
constexpr int kLossBlocks = 1024;

template <int C>
__device__ __forceinline__ void load_logits(const float* __restrict__ x, long n, long hw, long HW, float (&v)[C]) {
#pragma unroll
  for (int c = 0; c < C; ++c) v[c] = x[(n * C + c) * HW + hw];
}

// per-pixel value of the (unnormalised) loss and, optionally, d/dx_c of it
template <int C>
__device__ __forceinline__ float pixel_loss(const float (&v)[C], int t, int mode, float gamma, float (*grad)[C]) {
  float mx = v[0];
#pragma unroll
  for (int c = 1; c < C; ++c) mx = fmaxf(mx, v[c]);
  float e[C], sum = 0.f;
#pragma unroll
  for (int c = 0; c < C; ++c) {
    e[c] = expf(v[c] - mx);
    sum += e[c];
  }
  const float lse = logf(sum) + mx;
  float xt = 0.f;
#pragma unroll
  for (int c = 0; c < C; ++c) xt = (c == t) ? v[c] : xt;
  const float logp = xt - lse;  // log softmax at the target
  if (mode == 0) {
    if (grad) {
#pragma unroll
      for (int c = 0; c < C; ++c) (*grad)[c] = e[c] / sum - (c == t ? 1.f : 0.f);
    }
    return -logp;
  }
  const float pt = expf(logp);
  const float om = 1.f - pt;
  const float pw = powf(om, gamma);
  if (grad) {
    // L = -(1-pt)^g * log pt ;  dL/dx_c = [ g (1-pt)^(g-1) pt log pt - (1-pt)^g ] * (1[c==t] - p_c)
    const float pwm1 = (gamma == 0.f) ? 0.f : gamma * powf(om, gamma - 1.f);
    const float k = pwm1 * pt * logp - pw;
#pragma unroll
    for (int c = 0; c < C; ++c) (*grad)[c] = k * ((c == t ? 1.f : 0.f) - e[c] / sum);
  }
  return -pw * logp;
}

template <int C>
__global__ __launch_bounds__(256) void nll_fwd_kernel(const float* __restrict__ x, const long long* __restrict__ tgt,
                                                      const float* __restrict__ weight, double* __restrict__ partial,
                                                      long P, long HW, int mode, float gamma) {
  __shared__ double red[2][4];
  double sl = 0, sw = 0;
  for (long p = (long)blockIdx.x * 256 + threadIdx.x; p < P; p += (long)gridDim.x * 256) {
    const long n = p / HW, hw = p - n * HW;
    float v[C];
    load_logits<C>(x, n, hw, HW, v);
    const int t = (int)tgt[p];
    const float w = weight ? weight[t] : 1.f;
    const float l = pixel_loss<C>(v, t, mode, gamma, nullptr);
    sl += (double)(w * l);
    sw += (double)w;
  }
  sl = rs_wave_sum(sl);
  sw = rs_wave_sum(sw);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (lane == 0) {
    red[0][wave] = sl;
    red[1][wave] = sw;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    partial[blockIdx.x * 2] = (red[0][0] + red[0][1]) + (red[0][2] + red[0][3]);
    partial[blockIdx.x * 2 + 1] = (red[1][0] + red[1][1]) + (red[1][2] + red[1][3]);
  }
}

// loss = sum(w*l) / sum(w); stats[0] = loss, stats[1] = sum(w)
__global__ void nll_finalize_kernel(const double* __restrict__ partial, int nblocks, float* __restrict__ loss,
                                    float* __restrict__ stats) {
  __shared__ double red[2][256];
  double sl = 0, sw = 0;
  for (int b = threadIdx.x; b < nblocks; b += 256) {
    sl += partial[b * 2];
    sw += partial[b * 2 + 1];
  }
  red[0][threadIdx.x] = sl;
  red[1][threadIdx.x] = sw;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if (threadIdx.x < o) {
      red[0][threadIdx.x] += red[0][threadIdx.x + o];
      red[1][threadIdx.x] += red[1][threadIdx.x + o];
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    const float l = (float)(red[0][0] / red[1][0]);
    loss[0] = l;
    stats[0] = l;
    stats[1] = (float)red[1][0];
  }
}

// dx[c][p] = gout * w[t_p] * dl_p/dx_c / sum(w)
template <int C>
__global__ __launch_bounds__(256) void nll_bwd_kernel(const float* __restrict__ x, const long long* __restrict__ tgt,
                                                      const float* __restrict__ weight, const float* __restrict__ stats,
                                                      const float* __restrict__ gout, float* __restrict__ dx, long P, long HW,
                                                      int mode, float gamma) {
  const long p = (long)blockIdx.x * 256 + threadIdx.x;
  if (p >= P) return;
  const long n = p / HW, hw = p - n * HW;
  float v[C], g[C];
  load_logits<C>(x, n, hw, HW, v);
  const int t = (int)tgt[p];
  const float w = weight ? weight[t] : 1.f;
  pixel_loss<C>(v, t, mode, gamma, &g);
  const float k = (gout ? gout[0] : 1.f) * w / stats[1];
#pragma unroll
  for (int c = 0; c < C; ++c) dx[(n * C + c) * HW + hw] = k * g[c];
}

// counts[0..3] += (tn, fn, fp, tp) with the reference's naming: q = argmax/actual as floats; nan -> tn, +inf -> fn,
// 0 -> fp, 1 -> tp, anything else (only possible for C > 2) is dropped (robosat/metrics.py:35-41).
template <int C>
__global__ __launch_bounds__(256) void confusion_kernel(const float* __restrict__ x, const long long* __restrict__ tgt,
                                                        unsigned long long* __restrict__ counts, long P, long HW) {
  unsigned int c4[4] = {0, 0, 0, 0};
  for (long p = (long)blockIdx.x * 256 + threadIdx.x; p < P; p += (long)gridDim.x * 256) {
    const long n = p / HW, hw = p - n * HW;
    float v[C];
    load_logits<C>(x, n, hw, HW, v);
    int pred = 0;
    float best = v[0];
#pragma unroll
    for (int c = 1; c < C; ++c)
      if (v[c] > best) {  // torch.argmax: first maximal index
        best = v[c];
        pred = c;
      }
    const int act = (int)tgt[p];
    if (act == 0) {
      if (pred == 0) c4[0]++;  // 0/0 = nan
      else c4[1]++;            // k/0 = inf
    } else if (pred == 0) {
      c4[2]++;  // 0/k = 0
    } else if (pred == act) {
      c4[3]++;  // k/k = 1
    }
  }
  __shared__ unsigned int red[4];
  if (threadIdx.x < 4) red[threadIdx.x] = 0;
  __syncthreads();
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    unsigned int s = c4[i];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
    if ((threadIdx.x & 63) == 0) atomicAdd(&red[i], s);
  }
  __syncthreads();
  if (threadIdx.x < 4) atomicAdd(&counts[threadIdx.x], (unsigned long long)red[threadIdx.x]);
}

template <int C>
int run_fwd(const float* x, const long long* t, const float* w, float* loss, float* stats, double* part, long P, long HW,
            int mode, float gamma, hipStream_t s) {
  const long nb = (P + 255) / 256;
  const int grid = nb < kLossBlocks ? (int)nb : kLossBlocks;
  nll_fwd_kernel<C><<<grid, 256, 0, s>>>(x, t, w, part, P, HW, mode, gamma);
  nll_finalize_kernel<<<1, 256, 0, s>>>(part, grid, loss, stats);
  return RS_LAUNCH_RESULT();
}

template <int C>
int run_bwd(const float* x, const long long* t, const float* w, const float* stats, const float* gout, float* dx, long P,
            long HW, int mode, float gamma, hipStream_t s) {
  nll_bwd_kernel<C><<<rs_cdiv(P, 256), 256, 0, s>>>(x, t, w, stats, gout, dx, P, HW, mode, gamma);
  return RS_LAUNCH_RESULT();
}

template <int C>
int run_conf(const float* x, const long long* t, unsigned long long* counts, long P, long HW, hipStream_t s) {
  const long nb = (P + 255) / 256;
  const int grid = nb < kLossBlocks ? (int)nb : kLossBlocks;
  confusion_kernel<C><<<grid, 256, 0, s>>>(x, t, counts, P, HW);
  return RS_LAUNCH_RESULT();
}


// ---- mIoULoss2d (robosat/losses.py:53-83) ------------------------------------------------------------------------
// soft IoU per (class c, image n): inter = sum_p s_c m_c, union = sum_p (s_c + m_c - s_c m_c); miou = 1 - mean(inter/union);
// the reference returns Python max(miou, weighted NLL): whichever is larger, gradient through that branch only.
// Per-(n,c) sums: A = sum_{p: t=c} s_c, B = sum_p s_c, cnt = #{p: t=c}  =>  inter = A, union = B + cnt - A.
template <int C>
__global__ __launch_bounds__(256) void miou_partial_kernel(const float* __restrict__ x, const long long* __restrict__ tgt,
                                                           const float* __restrict__ weight, double* __restrict__ partial,
                                                           long HW, int nblk) {
  // partial[(n*nblk + b)][3*C + 2]
  __shared__ double red[4][3 * C + 2];
  const long n = blockIdx.y;
  double acc[3 * C + 2];
#pragma unroll
  for (int i = 0; i < 3 * C + 2; ++i) acc[i] = 0;
  for (long hw = (long)blockIdx.x * 256 + threadIdx.x; hw < HW; hw += (long)nblk * 256) {
    float v[C];
    load_logits<C>(x, n, hw, HW, v);
    const int t = (int)tgt[n * HW + hw];
    float mx = v[0];
#pragma unroll
    for (int c = 1; c < C; ++c) mx = fmaxf(mx, v[c]);
    float e[C], sum = 0.f;
#pragma unroll
    for (int c = 0; c < C; ++c) {
      e[c] = expf(v[c] - mx);
      sum += e[c];
    }
    float xt = 0.f;
#pragma unroll
    for (int c = 0; c < C; ++c) {
      const float sc = e[c] / sum;
      acc[c] += (c == t) ? (double)sc : 0.0;
      acc[C + c] += (double)sc;
      acc[2 * C + c] += (c == t) ? 1.0 : 0.0;
      xt = (c == t) ? v[c] : xt;
    }
    const float w = weight ? weight[t] : 1.f;
    acc[3 * C] += (double)(w * ((logf(sum) + mx) - xt));
    acc[3 * C + 1] += (double)w;
  }
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int i = 0; i < 3 * C + 2; ++i) {
    const double s = rs_wave_sum(acc[i]);
    if (lane == 0) red[wave][i] = s;
  }
  __syncthreads();
  if (threadIdx.x < 3 * C + 2) {
    const int i = threadIdx.x;
    partial[(n * nblk + blockIdx.x) * (3 * C + 2) + i] = (red[0][i] + red[1][i]) + (red[2][i] + red[3][i]);
  }
}

// stats: [0] loss, [1] sum w, [2] branch (0 = miou, 1 = nll), [3 + n*C + c] = g for m=1, [3 + N*C + n*C + c] = g for m=0,
// [3 + 2*N*C] = the miou branch's value, [4 + 2*N*C] = the nll branch's numerator sum_i w_i l_i (data-parallel ranks
// exchange these two and [1] to take the reference's ONE branch decision over the global batch: robosat_amd/losses.py)
__global__ void miou_finalize_kernel(const double* __restrict__ partial, int N, int C, int nblk, float* __restrict__ loss,
                                     float* __restrict__ stats) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  const int S = 3 * C + 2;
  double sl = 0, sw = 0, rsum = 0;
  for (int n = 0; n < N; ++n) {
    for (int c = 0; c < C; ++c) {
      double A = 0, B = 0, cnt = 0;
      for (int b = 0; b < nblk; ++b) {
        const double* p = partial + ((long)n * nblk + b) * S;
        A += p[c];
        B += p[C + c];
        cnt += p[2 * C + c];
      }
      const double U = B + cnt - A;
      rsum += A / U;
      const double k = 1.0 / ((double)C * (double)N);
      stats[3 + n * C + c] = (float)(-k / U);
      stats[3 + N * C + n * C + c] = (float)(k * A / (U * U));
    }
    for (int b = 0; b < nblk; ++b) {
      const double* p = partial + ((long)n * nblk + b) * S;
      sl += p[3 * C];
      sw += p[3 * C + 1];
    }
  }
  const float miou = (float)(1.0 - rsum / ((double)C * (double)N));
  const float nll = (float)(sl / sw);
  const bool use_nll = nll > miou;  // Python max(miou, nll): returns nll only if nll > miou
  loss[0] = use_nll ? nll : miou;
  stats[0] = loss[0];
  stats[1] = (float)sw;
  stats[2] = use_nll ? 1.f : 0.f;
  stats[3 + 2 * N * C] = miou;
  stats[4 + 2 * N * C] = (float)sl;
}

template <int C>
__global__ __launch_bounds__(256) void miou_bwd_kernel(const float* __restrict__ x, const long long* __restrict__ tgt,
                                                       const float* __restrict__ weight, const float* __restrict__ stats,
                                                       const float* __restrict__ gout, float* __restrict__ dx, int N, long HW) {
  const long hw = (long)blockIdx.x * 256 + threadIdx.x;
  const long n = blockIdx.y;
  if (hw >= HW) return;
  float v[C], g[C];
  load_logits<C>(x, n, hw, HW, v);
  const int t = (int)tgt[n * HW + hw];
  const float go = gout ? gout[0] : 1.f;
  if (stats[2] > 0.5f) {
    const float w = weight ? weight[t] : 1.f;
    pixel_loss<C>(v, t, 0, 0.f, &g);
    const float k = go * w / stats[1];
#pragma unroll
    for (int c = 0; c < C; ++c) dx[(n * C + c) * HW + hw] = k * g[c];
    return;
  }
  float mx = v[0];
#pragma unroll
  for (int c = 1; c < C; ++c) mx = fmaxf(mx, v[c]);
  float sum = 0.f;
#pragma unroll
  for (int c = 0; c < C; ++c) {
    v[c] = expf(v[c] - mx);
    sum += v[c];
  }
  float dot = 0.f;
#pragma unroll
  for (int c = 0; c < C; ++c) {
    v[c] = v[c] / sum;
    g[c] = (c == t) ? stats[3 + n * C + c] : stats[3 + N * C + n * C + c];
    dot += g[c] * v[c];
  }
#pragma unroll
  for (int c = 0; c < C; ++c) dx[(n * C + c) * HW + hw] = go * v[c] * (g[c] - dot);
}

constexpr int kMiouBlocks = 64;  // per image

template <int C>
int run_miou_fwd(const float* x, const long long* t, const float* w, float* loss, float* stats, double* part, int N, long HW,
                 hipStream_t s) {
  const long nb = (HW + 255) / 256;
  const int nblk = nb < kMiouBlocks ? (int)nb : kMiouBlocks;
  miou_partial_kernel<C><<<dim3(nblk, N), 256, 0, s>>>(x, t, w, part, HW, nblk);
  miou_finalize_kernel<<<1, 64, 0, s>>>(part, N, C, nblk, loss, stats);
  return RS_LAUNCH_RESULT();
}

template <int C>
int run_miou_bwd(const float* x, const long long* t, const float* w, const float* stats, const float* gout, float* dx, int N,
                 long HW, hipStream_t s) {
  miou_bwd_kernel<C><<<dim3(rs_cdiv(HW, 256), N), 256, 0, s>>>(x, t, w, stats, gout, dx, N, HW);
  return RS_LAUNCH_RESULT();
}

#define RS_DISPATCH_C(C, CALL)              \
  switch (C) {                              \
    case 1: { constexpr int K = 1; return CALL; } \
    case 2: { constexpr int K = 2; return CALL; } \
    case 3: { constexpr int K = 3; return CALL; } \
    case 4: { constexpr int K = 4; return CALL; } \
    case 5: { constexpr int K = 5; return CALL; } \
    case 6: { constexpr int K = 6; return CALL; } \
    case 7: { constexpr int K = 7; return CALL; } \
    default: { constexpr int K = 8; return CALL; } \
  }

}  // namespace

extern "C" long rs_nll_loss_workspace_bytes(void) { return (long)kLossBlocks * 2 * (long)sizeof(double); }

extern "C" int rs_nll_loss_fwd(const float* logits, const long long* targets, const float* weight, float* loss, float* stats,
                               int N, int C, int H, int W, int mode, float gamma, void* workspace, rs_stream_t stream) {
  if (!logits || !targets || !loss || !stats || !workspace || N <= 0 || C <= 0 || C > kMaxC || H <= 0 || W <= 0 ||
      mode < 0 || mode > 1)
    return RS_EINVAL;
  const long HW = (long)H * W, P = (long)N * HW;
  hipStream_t s = (hipStream_t)stream;
  double* part = reinterpret_cast<double*>(workspace);
  RS_DISPATCH_C(C, run_fwd<K>(logits, targets, weight, loss, stats, part, P, HW, mode, gamma, s));
}

extern "C" int rs_nll_loss_bwd(const float* logits, const long long* targets, const float* weight, const float* stats,
                               const float* grad_out, float* dlogits, int N, int C, int H, int W, int mode, float gamma,
                               rs_stream_t stream) {
  if (!logits || !targets || !stats || !dlogits || N <= 0 || C <= 0 || C > kMaxC || H <= 0 || W <= 0 || mode < 0 || mode > 1)
    return RS_EINVAL;
  const long HW = (long)H * W, P = (long)N * HW;
  hipStream_t s = (hipStream_t)stream;
  RS_DISPATCH_C(C, run_bwd<K>(logits, targets, weight, stats, grad_out, dlogits, P, HW, mode, gamma, s));
}

extern "C" int rs_confusion_counts(const float* scores, const long long* targets, unsigned long long* counts, int N, int C,
                                   int H, int W, rs_stream_t stream) {
  if (!scores || !targets || !counts || N <= 0 || C <= 0 || C > kMaxC || H <= 0 || W <= 0) return RS_EINVAL;
  const long HW = (long)H * W, P = (long)N * HW;
  hipStream_t s = (hipStream_t)stream;
  RS_DISPATCH_C(C, run_conf<K>(scores, targets, counts, P, HW, s));
}

extern "C" long rs_miou_loss_workspace_bytes(int N, int C) {
  if (N <= 0 || C <= 0 || C > kMaxC) return RS_EINVAL;
  return (long)N * kMiouBlocks * (3 * C + 2) * (long)sizeof(double);
}

extern "C" int rs_miou_loss_fwd(const float* logits, const long long* targets, const float* weight, float* loss, float* stats,
                                int N, int C, int H, int W, void* workspace, rs_stream_t stream) {
  if (!logits || !targets || !loss || !stats || !workspace || N <= 0 || C <= 0 || C > kMaxC || H <= 0 || W <= 0)
    return RS_EINVAL;
  const long HW = (long)H * W;
  hipStream_t s = (hipStream_t)stream;
  double* part = reinterpret_cast<double*>(workspace);
  RS_DISPATCH_C(C, run_miou_fwd<K>(logits, targets, weight, loss, stats, part, N, HW, s));
}

extern "C" int rs_miou_loss_bwd(const float* logits, const long long* targets, const float* weight, const float* stats,
                                const float* grad_out, float* dlogits, int N, int C, int H, int W, rs_stream_t stream) {
  if (!logits || !targets || !stats || !dlogits || N <= 0 || C <= 0 || C > kMaxC || H <= 0 || W <= 0) return RS_EINVAL;
  const long HW = (long)H * W;
  hipStream_t s = (hipStream_t)stream;
  RS_DISPATCH_C(C, run_miou_bwd<K>(logits, targets, weight, stats, grad_out, dlogits, N, HW, s));
}

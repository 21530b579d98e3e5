// The callers either side of the U-Net on the device: integer / byte work, HBM-bound.
//
//   rs_confusion_matrix   robosat/metrics.py:27-41 generalised to C classes: counts[actual][predicted] (the reference's four
//                         counters are the C = 2 matrix: tn = [0][0], "fn" = [0][1], "fp" = [1][0], tp = [1][1])
//   rs_label_histogram_u8 robosat/tools/weights.py:41-47: np.bincount over every label tile of the training set
//   rs_softvote_masks     robosat/tools/masks.py:42-84: un-quantise K probability PNGs, weighted average, argmax
//   rs_augment_tiles      robosat/tools/train.py:248-260 + robosat/transforms.py:127-221: random horizontal flip + up to
//                         three 90-degree rotations + ToTensor + Normalize, from a cache of decoded uint8 tiles in HBM
#include "common.h"

namespace {

// ---- C x C confusion matrix ------------------------------------------------------------------------------------------
// predicted = first maximum over the class scores (torch.argmax ties -> lowest index; NaN never wins unless first).
// One thread per pixel; per-block histogram in LDS (C*C <= 64 bins), one 64-bit atomic per non-empty bin per block.
__global__ __launch_bounds__(256) void confusion_matrix_kernel(const float* __restrict__ scores, const long long* __restrict__ targets,
                                                               unsigned long long* __restrict__ counts, long P, long HW, int C) {
  __shared__ unsigned int hist[64];
  if (threadIdx.x < 64) hist[threadIdx.x] = 0;
  __syncthreads();
  for (long pix = (long)blockIdx.x * blockDim.x + threadIdx.x; pix < P; pix += (long)gridDim.x * blockDim.x) {
    const long n = pix / HW, hw = pix - n * HW;
    const float* s = scores + n * C * HW + hw;
    int best = 0;
    float bv = s[0];
    for (int c = 1; c < C; ++c) {
      const float v = s[c * HW];
      if (v > bv) {
        bv = v;
        best = c;
      }
    }
    const long long t = targets[pix];
    if (t >= 0 && t < C) atomicAdd(&hist[(int)t * C + best], 1u);
  }
  __syncthreads();
  if (threadIdx.x < C * C && hist[threadIdx.x]) atomicAdd(&counts[threadIdx.x], (unsigned long long)hist[threadIdx.x]);
}

// ---- label histogram -------------------------------------------------------------------------------------------------
// 16 labels per thread per load; per-wave private LDS histograms (4 per block) to thin out same-address atomics on label
// maps that are one class almost everywhere.
__global__ __launch_bounds__(256) void label_histogram_kernel(const uint8_t* __restrict__ labels, long n,
                                                              unsigned long long* __restrict__ counts) {
  __shared__ unsigned int hist[4][256];
  for (int i = threadIdx.x; i < 4 * 256; i += 256) (&hist[0][0])[i] = 0;
  __syncthreads();
  unsigned int* h = hist[threadIdx.x >> 6];
  const long nvec = n / 16;
  typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
  const u32x4* v = reinterpret_cast<const u32x4*>(labels);
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < nvec; i += (long)gridDim.x * blockDim.x) {
    const u32x4 q = v[i];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const unsigned int wv = q[e];
      atomicAdd(&h[wv & 255u], 1u);
      atomicAdd(&h[(wv >> 8) & 255u], 1u);
      atomicAdd(&h[(wv >> 16) & 255u], 1u);
      atomicAdd(&h[wv >> 24], 1u);
    }
  }
  if (blockIdx.x == 0)  // tail
    for (long i = nvec * 16 + threadIdx.x; i < n; i += 256) atomicAdd(&h[labels[i]], 1u);
  __syncthreads();
  const unsigned int tot = hist[0][threadIdx.x] + hist[1][threadIdx.x] + hist[2][threadIdx.x] + hist[3][threadIdx.x];
  if (tot) atomicAdd(&counts[threadIdx.x], (unsigned long long)tot);
}

// ---- soft vote ---------------------------------------------------------------------------------------------------------
// q: [K][P][Cq] quantised foreground-class probabilities (Cq = classes - 1 bytes per pixel, as `rs predict` writes them);
// class 0 = 1 - sum of the others.  out[p] = argmax_c sum_k w_k * prob_k[c] / sum_k w_k, float64, models summed in order
// and the first maximum taken -- numpy's np.argmax(np.average(probs, axis=0, weights=w), axis=0) on the same values.
__global__ __launch_bounds__(256) void softvote_kernel(const uint8_t* __restrict__ q, const double* __restrict__ weights,
                                                       const double* __restrict__ anchors, uint8_t* __restrict__ out, int K, long P,
                                                       int Cq) {
  __shared__ double anc[256];
  anc[threadIdx.x] = anchors[threadIdx.x];
  __syncthreads();
  const long p = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= P) return;
  double acc[8];
#pragma unroll
  for (int c = 0; c < 8; ++c) acc[c] = 0.0;
  double scl = 0.0;
  for (int k = 0; k < K; ++k) {
    const double wk = weights ? weights[k] : 1.0;
    scl += wk;
    const uint8_t* qp = q + ((long)k * P + p) * Cq;
    double bg = 1.0;
    // masks.py:51-52: background = 1 - foreground (binary).  C > 2: 1 - (f1 + f2 + ...), subtracted one by one.
    for (int c = 0; c < Cq; ++c) {
      const double f = anc[qp[c]];
      bg -= f;
      acc[c + 1] += weights ? f * wk : f;
    }
    acc[0] += weights ? bg * wk : bg;
  }
  int best = 0;
  double bv = weights ? acc[0] / scl : acc[0] / (double)K;
  for (int c = 1; c <= Cq; ++c) {
    const double v = weights ? acc[c] / scl : acc[c] / (double)K;
    if (v > bv) {
      bv = v;
      best = c;
    }
  }
  out[p] = (uint8_t)best;
}

// ---- augmentation from a decoded-tile cache ----------------------------------------------------------------------------
// op = f + 2*k: PIL FLIP_LEFT_RIGHT when f, then k times ROTATE_90 (counter-clockwise; out[i][j] = in[j][S-1-i]).
// One thread per output pixel: all C channels of the image (ToTensor + Normalize: (v/255 - mean)/std in fp32, IEEE
// division -- the same expression as rs_u8_to_nhwc4_norm, bit-identical to the host ops) + the mask label.
struct AugArgs {
  const uint8_t* images;  // [T][S][S][C]
  const uint8_t* masks;   // [T][S][S] (may be NULL)
  const int* index;       // [N] tile of the cache
  const int* op;          // [N]
  float* out;             // [N][C][S][S]
  long long* out_mask;    // [N][S][S]
  float mean[4], stdv[4];
  int N, S, C;
};

__global__ __launch_bounds__(256) void augment_kernel(const AugArgs a) {
  const long SS = (long)a.S * a.S;
  const long pix = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (pix >= (long)a.N * SS) return;
  const int n = (int)(pix / SS);
  const long rem = pix - n * SS;
  int y = (int)(rem / a.S), x = (int)(rem - (long)y * a.S);
  const int op = a.op[n];
  for (int r = (op >> 1) & 3; r > 0; --r) {  // undo the rotations, last first
    const int py = x, px = a.S - 1 - y;
    y = py;
    x = px;
  }
  if (op & 1) x = a.S - 1 - x;
  const long src = ((long)a.index[n] * SS + (long)y * a.S + x);
  const uint8_t* ip = a.images + src * a.C;
  for (int c = 0; c < a.C; ++c) a.out[((long)n * a.C + c) * SS + rem] = ((float)ip[c] / 255.0f - a.mean[c]) / a.stdv[c];
  if (a.masks) a.out_mask[pix] = (long long)a.masks[src];
}

}  // namespace

extern "C" int rs_confusion_matrix(const float* scores, const int64_t* targets, int64_t* counts, int N, int C, int H, int W,
                                   rs_stream_t stream) {
  if (!scores || !targets || !counts || N <= 0 || C <= 0 || C > 8 || H <= 0 || W <= 0) return RS_EINVAL;
  const long HW = (long)H * W, P = (long)N * HW;
  const int grid = (int)(rs_cdiv(P, 256) < 4096 ? rs_cdiv(P, 256) : 4096);
  confusion_matrix_kernel<<<grid, 256, 0, (hipStream_t)stream>>>(scores, reinterpret_cast<const long long*>(targets),
                                                                 reinterpret_cast<unsigned long long*>(counts), P, HW, C);
  return RS_LAUNCH_RESULT();
}

extern "C" int rs_label_histogram_u8(const uint8_t* labels, long n, int64_t* counts256, rs_stream_t stream) {
  if (!labels || !counts256 || n <= 0 || ((uintptr_t)labels & 15)) return RS_EINVAL;
  const long nvec = n / 16 > 0 ? n / 16 : 1;
  const int grid = (int)(rs_cdiv(nvec, 256) < 2048 ? rs_cdiv(nvec, 256) : 2048);
  label_histogram_kernel<<<grid, 256, 0, (hipStream_t)stream>>>(labels, n, reinterpret_cast<unsigned long long*>(counts256));
  return RS_LAUNCH_RESULT();
}

extern "C" int rs_softvote_masks(const uint8_t* q, const double* weights, const double* anchors, uint8_t* out, int K, long P,
                                 int C, rs_stream_t stream) {
  if (!q || !anchors || !out || K <= 0 || P <= 0 || C < 2 || C > 8) return RS_EINVAL;
  softvote_kernel<<<rs_cdiv(P, 256), 256, 0, (hipStream_t)stream>>>(q, weights, anchors, out, K, P, C - 1);
  return RS_LAUNCH_RESULT();
}

extern "C" int rs_augment_tiles(const uint8_t* images, const uint8_t* masks, const int32_t* index, const int32_t* op,
                                const float* mean, const float* stdv, float* out_images, int64_t* out_masks, int N, int S, int C,
                                rs_stream_t stream) {
  if (!images || !index || !op || !mean || !stdv || !out_images || N <= 0 || S <= 0 || C <= 0 || C > 4) return RS_EINVAL;
  if (masks && !out_masks) return RS_EINVAL;
  AugArgs a;
  a.images = images;
  a.masks = masks;
  a.index = index;
  a.op = op;
  a.out = out_images;
  a.out_mask = reinterpret_cast<long long*>(out_masks);
  for (int c = 0; c < C; ++c) {
    a.mean[c] = mean[c];
    a.stdv[c] = stdv[c];
  }
  a.N = N;
  a.S = S;
  a.C = C;
  augment_kernel<<<rs_cdiv((long)N * S * S, 256), 256, 0, (hipStream_t)stream>>>(a);
  return RS_LAUNCH_RESULT();
}

// HBM-bound layout / pooling / head kernels of the U-Net forward (reference robosat/unet.py:122-141,
// robosat/tools/predict.py:87).  All are streaming kernels: 16-byte accesses, coalesced on the NHWC channel axis.
#include "common.h"

namespace {

// images.to(device): NCHW [N][C][H][W] -> NHWC4 (channel 3 zero when C == 3).  One thread per pixel: three/four
// coalesced plane reads, one 16-byte store.
__global__ void nchw_to_nhwc4_kernel(const float* __restrict__ x, float* __restrict__ y, int C, long HW, long total) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  const long n = i / HW, hw = i - n * HW;
  const float* px = x + n * C * HW + hw;
  f32x4 v;
  v[0] = px[0];
  v[1] = C > 1 ? px[HW] : 0.f;
  v[2] = C > 2 ? px[2 * HW] : 0.f;
  v[3] = C > 3 ? px[3 * HW] : 0.f;
  *reinterpret_cast<f32x4*>(y + i * 4) = v;
}

// N1: the image side of `rs predict` on the device.  Decoded tiles travel as uint8 HWC (1 byte
// per sample instead of 4) and ToTensor + Normalize (reference tools/predict.py:71: x/255, then (x - mean)/std, both in
// fp32 and in that order) + the NHWC4 layout the stem wants happen here.  IEEE divisions: bit-identical to the host ops.
__global__ void u8_to_nhwc4_norm_kernel(const uint8_t* __restrict__ img, float* __restrict__ y, f32x4 mean, f32x4 stdv, int C,
                                        long total) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  const uint8_t* px = img + i * C;
  f32x4 v = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int c = 0; c < 4; ++c)
    if (c < C) v[c] = ((float)px[c] / 255.0f - mean[c]) / stdv[c];
  *reinterpret_cast<f32x4*>(y + i * 4) = v;
}

// F.max_pool2d on NHWC, V = 4 or 8 channels per thread (8 when C % 8 == 0: 16-byte accesses in bf16).  Window scanned
// row-major with a strict '>' so the FIRST maximum wins, as torch's max_pool2d_with_indices does; padding is -inf (never
// selected when any tap is valid).  One byte per output element records the winning tap for the backward.
template <typename TI, typename TO, int V>
__global__ void maxpool_nhwc_kernel(const TI* __restrict__ x, TO* __restrict__ y, uint8_t* __restrict__ amax, int H,
                                    int W, int CV, int k, int stride, int pad, int Ho, int Wo, long total) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  const int c = (int)(i % CV);
  long pix = i / CV;
  const int ox = (int)(pix % Wo);
  pix /= Wo;
  const int oy = (int)(pix % Ho);
  const long n = pix / Ho;
  const float ninf = -__builtin_huge_valf();
  rs_vecf<V> best;
  int bi[V];
#pragma unroll
  for (int e = 0; e < V; ++e) best.v[e] = ninf, bi[e] = 0;
  bool first = true;
  for (int r = 0; r < k; ++r) {
    const int iy = oy * stride - pad + r;
    if ((unsigned)iy >= (unsigned)H) continue;
    for (int s = 0; s < k; ++s) {
      const int ix = ox * stride - pad + s;
      if ((unsigned)ix >= (unsigned)W) continue;
      const rs_vecf<V> v = rs_ldv<V>(x + (((n * H + iy) * W + ix) * (long)CV + c) * V);
      const int tap = r * k + s;
#pragma unroll
      for (int e = 0; e < V; ++e) {
        if (first || v.v[e] > best.v[e] || v.v[e] != v.v[e]) {  // torch: (val > maxval) || isnan(val)
          best.v[e] = v.v[e];
          bi[e] = tap;
        }
      }
      first = false;
    }
  }
  rs_stv<V>(y + i * V, best);
  if (amax) {
#pragma unroll
    for (int h = 0; h < V / 4; ++h) {
      const uint32_t packed = (uint32_t)bi[4 * h] | ((uint32_t)bi[4 * h + 1] << 8) | ((uint32_t)bi[4 * h + 2] << 16) |
                              ((uint32_t)bi[4 * h + 3] << 24);
      *reinterpret_cast<uint32_t*>(amax + i * V + 4 * h) = packed;
    }
  }
}

__global__ void bn_fold_kernel(const float* __restrict__ gamma, const float* __restrict__ beta,
                               const float* __restrict__ mean, const float* __restrict__ var, float eps,
                               float* __restrict__ scale, float* __restrict__ shift, int C) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  const float sc = gamma[c] / sqrtf(var[c] + eps);
  scale[c] = sc;
  shift[c] = beta[c] - mean[c] * sc;
}

// What happens to a pixel's C logits (shared by the two forms of the kernel below): optional softmax, then fp32 NCHW logits /
// probabilities, or the quantised probability bytes of `rs predict`, or the argmax byte of `rs serve`.
template <int C>
__device__ __forceinline__ void final_epilogue(float (&acc)[C], long pix, long HW, int softmax, const double* __restrict__ anchors,
                                               uint8_t* __restrict__ qout, float* __restrict__ out, int Wimg, int ov) {
  if (softmax == 1 || softmax == 2) {  // (mode 3 takes the argmax of the LOGITS: serve.py:160-164 -- no exp/div, exact ties)
    float mx = acc[0];
#pragma unroll
    for (int c = 1; c < C; ++c) mx = fmaxf(mx, acc[c]);
    float sum = 0.f;
#pragma unroll
    for (int c = 0; c < C; ++c) {
      acc[c] = expf(acc[c] - mx);
      sum += acc[c];
    }
#pragma unroll
    for (int c = 0; c < C; ++c) acc[c] = acc[c] / sum;
  }
  const long n = pix / HW, hw = pix - n * HW;
  if (softmax == 2) {  // un-buffer crop + np.digitize of every non-background class: C - 1 bytes per pixel
    if (C < 2) return;
    const int Himg = (int)(HW / Wimg);
    const int yy = (int)(hw / Wimg), xx = (int)(hw - (long)yy * Wimg);
    const int S_h = Himg - 2 * ov, S_w = Wimg - 2 * ov;
    if (yy < ov || yy >= Himg - ov || xx < ov || xx >= Wimg - ov) return;
    uint8_t* qo = qout + ((n * S_h + (yy - ov)) * (long)S_w + (xx - ov)) * (C - 1);
#pragma unroll
    for (int c = 1; c < C; ++c) {
      const double pf = (double)acc[c];
      int q = (int)(pf * 255.0);  // anchors[i] ~ i/255: first guess, then settle on the exact table (anchors ascending)
      q = q < 0 ? 0 : (q > 255 ? 255 : q);
      while (q < 255 && anchors[q + 1] <= pf) ++q;
      while (q >= 0 && anchors[q] > pf) --q;
      qo[c - 1] = (uint8_t)((q + 1) & 0xff);  // bins are 1-based; 256 wraps to 0
    }
    return;
  }
  if (softmax == 3) {  // class index of the first maximum logit (np.argmax over axis 0), one byte per pixel
    int best = 0;
#pragma unroll
    for (int c = 1; c < C; ++c)
      if (acc[c] > acc[best]) best = c;
    qout[pix] = (uint8_t)best;
    return;
  }
  float* o = out + n * C * HW + hw;
#pragma unroll
  for (int c = 0; c < C; ++c) o[c * HW] = acc[c];
}

// self.final (+ optional softmax): 256 pixels per block.  The block's [256][Cin] slab is read with fully coalesced
// 16-byte loads into LDS (row stride Cin+1: conflict-free per-pixel reads), then one thread owns one pixel, keeps
// the C class sums in registers and writes C coalesced NCHW planes.
// softmax = 2 (N1): instead of the C probability planes write ONE byte per pixel of the
// un-buffered crop: np.digitize(p_foreground, np.linspace(0, 1, 256)).astype(uint8) (reference tools/predict.py:98-103:
// 1-based bin = number of anchors <= p, 256 wraps to 0), anchors passed in as the host's own float64 linspace so the
// comparison is the one numpy does.  `qout` is [N][H-2*ov][W-2*ov]; only S*S bytes per tile go back to the host.
template <int C, typename T>
__global__ __launch_bounds__(256) void final_conv1x1_kernel(const T* __restrict__ x, const float* __restrict__ w,
                                                            const float* __restrict__ bias, float* __restrict__ out,
                                                            long P, long HW, int Cin, int softmax,
                                                            const double* __restrict__ anchors = nullptr,
                                                            uint8_t* __restrict__ qout = nullptr, int Wimg = 0, int ov = 0) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  const int ld = Cin + 1;
  float* xs = sm;             // [256][Cin+1]
  float* ws = sm + 256 * ld;  // [C][Cin] then bias[C]
  const int tid = threadIdx.x;
  const long p0 = (long)blockIdx.x * 256;
  const int q = Cin >> 2;  // float4 per pixel
  for (int f = tid; f < 256 * q; f += 256) {
    const int px = f / q, c4 = f - px * q;
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if (p0 + px < P) v = rs_ld4(x + (p0 + px) * Cin + c4 * 4);
    float* d = xs + px * ld + c4 * 4;
    d[0] = v[0];
    d[1] = v[1];
    d[2] = v[2];
    d[3] = v[3];
  }
  for (int f = tid; f < C * Cin; f += 256) ws[f] = w[f];
  if (tid < C) ws[C * Cin + tid] = bias ? bias[tid] : 0.f;
  __syncthreads();
  const long pix = p0 + tid;
  if (pix >= P) return;
  float acc[C];
#pragma unroll
  for (int c = 0; c < C; ++c) acc[c] = 0.f;
  const float* xr = xs + tid * ld;
  for (int k = 0; k < Cin; ++k) {
    const float xv = xr[k];
#pragma unroll
    for (int c = 0; c < C; ++c) acc[c] = fmaf(xv, ws[c * Cin + k], acc[c]);
  }
#pragma unroll
  for (int c = 0; c < C; ++c) acc[c] += ws[C * Cin + c];
  final_epilogue<C>(acc, pix, HW, softmax, anchors, qout, out, Wimg, ov);
}

// The model's own shape, Cin = 32 (unet.py:108: num_filters = 32): one thread = one pixel, its 32 channels arrive as 16-byte
// loads issued together (4 in bf16, 8 in fp32) straight into registers -- no LDS transposition of the activations; the
// C x 32 weights sit in LDS and are read as broadcasts.  Same multiply-add order as the general kernel above.
template <int C, typename T>
__global__ __launch_bounds__(256) void final_conv1x1_px32_kernel(const T* __restrict__ x, const float* __restrict__ w,
                                                                 const float* __restrict__ bias, float* __restrict__ out, long P,
                                                                 long HW, int softmax, const double* __restrict__ anchors,
                                                                 uint8_t* __restrict__ qout, int Wimg, int ov) {
  __shared__ __attribute__((aligned(16))) float ws[C * 32 + C];
  const int tid = threadIdx.x;
  for (int f = tid; f < C * 32; f += 256) ws[f] = w[f];
  if (tid < C) ws[C * 32 + tid] = bias ? bias[tid] : 0.f;
  __syncthreads();
  const long pix = (long)blockIdx.x * 256 + tid;
  if (pix >= P) return;
  float xv[32];
  rs_ld_row32(x + pix * 32, xv);
  float acc[C];
#pragma unroll
  for (int c = 0; c < C; ++c) acc[c] = 0.f;
#pragma unroll
  for (int k4 = 0; k4 < 8; ++k4) {
#pragma unroll
    for (int c = 0; c < C; ++c) {
      const f32x4 wv = *reinterpret_cast<const f32x4*>(ws + c * 32 + k4 * 4);
#pragma unroll
      for (int e = 0; e < 4; ++e) acc[c] = fmaf(xv[k4 * 4 + e], wv[e], acc[c]);
    }
  }
#pragma unroll
  for (int c = 0; c < C; ++c) acc[c] += ws[C * 32 + c];
  final_epilogue<C>(acc, pix, HW, softmax, anchors, qout, out, Wimg, ov);
}

template <int C, typename T>
int launch_final(const T* x, const float* w, const float* bias, float* out, long P, long HW, int Cin, int softmax,
                 hipStream_t s, const double* anchors = nullptr, uint8_t* qout = nullptr, int Wimg = 0, int ov = 0) {
  if (Cin == 32) {
    final_conv1x1_px32_kernel<C, T><<<rs_cdiv(P, 256), 256, 0, s>>>(x, w, bias, out, P, HW, softmax, anchors, qout, Wimg, ov);
    return RS_LAUNCH_RESULT();
  }
  const size_t smem = (size_t)(256 * (Cin + 1) + C * Cin + C) * sizeof(float);
  if (smem > 64 * 1024) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&final_conv1x1_kernel<C, T>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    if (e != hipSuccess) return (int)e;
  }
  final_conv1x1_kernel<C, T><<<rs_cdiv(P, 256), 256, smem, s>>>(x, w, bias, out, P, HW, Cin, softmax, anchors, qout, Wimg, ov);
  return RS_LAUNCH_RESULT();
}

template <typename T>
int dispatch_final(const T* x, const float* w, const float* bias, float* out, long P, long HW, int Cin, int C, int softmax,
                   hipStream_t s) {
  switch (C) {
    case 1: return launch_final<1>(x, w, bias, out, P, HW, Cin, softmax, s);
    case 2: return launch_final<2>(x, w, bias, out, P, HW, Cin, softmax, s);
    case 3: return launch_final<3>(x, w, bias, out, P, HW, Cin, softmax, s);
    case 4: return launch_final<4>(x, w, bias, out, P, HW, Cin, softmax, s);
    case 5: return launch_final<5>(x, w, bias, out, P, HW, Cin, softmax, s);
    case 6: return launch_final<6>(x, w, bias, out, P, HW, Cin, softmax, s);
    case 7: return launch_final<7>(x, w, bias, out, P, HW, Cin, softmax, s);
    default: return launch_final<8>(x, w, bias, out, P, HW, Cin, softmax, s);
  }
}

template <typename TI, typename TO>
int launch_maxpool(const void* x, void* y, uint8_t* argmax, int H, int W, int C, int k, int stride, int pad, int Ho,
                   int Wo, long total, hipStream_t s) {
  if ((C & 7) == 0)
    maxpool_nhwc_kernel<TI, TO, 8><<<rs_cdiv(total / 2, 256), 256, 0, s>>>(reinterpret_cast<const TI*>(x), reinterpret_cast<TO*>(y),
                                                                           argmax, H, W, C / 8, k, stride, pad, Ho, Wo, total / 2);
  else
    maxpool_nhwc_kernel<TI, TO, 4><<<rs_cdiv(total, 256), 256, 0, s>>>(reinterpret_cast<const TI*>(x), reinterpret_cast<TO*>(y),
                                                                       argmax, H, W, C / 4, k, stride, pad, Ho, Wo, total);
  return RS_LAUNCH_RESULT();
}

// fp32 -> bf16 (round to nearest even), 8 elements per thread; tail handled element-wise
template <bool SCALED>
__global__ void cast_f32_bf16_kernel(const float* __restrict__ src, bf16_t* __restrict__ dst, long n, float scale) {
  const long i = ((long)blockIdx.x * blockDim.x + threadIdx.x) * 8;
  if (i + 8 <= n) {
    const f32x4 a = *reinterpret_cast<const f32x4*>(src + i), b = *reinterpret_cast<const f32x4*>(src + i + 4);
    bf16x8 o;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      o[e] = (bf16_t)(SCALED ? a[e] * scale : a[e]);
      o[4 + e] = (bf16_t)(SCALED ? b[e] * scale : b[e]);
    }
    *reinterpret_cast<bf16x8*>(dst + i) = o;
  } else {
    for (long j = i; j < n; ++j) dst[j] = (bf16_t)(SCALED ? src[j] * scale : src[j]);
  }
}

}  // namespace

extern "C" int rs_abi_version(void) { return 21; }

extern "C" int rs_nchw_to_nhwc4(const float* x, float* y, int N, int C, int H, int W, rs_stream_t stream) {
  if (!x || !y || N <= 0 || C <= 0 || C > 4 || H <= 0 || W <= 0) return RS_EINVAL;
  const long HW = (long)H * W, total = (long)N * HW;
  nchw_to_nhwc4_kernel<<<rs_cdiv(total, 256), 256, 0, (hipStream_t)stream>>>(x, y, C, HW, total);
  return RS_LAUNCH_RESULT();
}

extern "C" int rs_maxpool2d_fwd_dt(const void* x, int x_dtype, void* y, int y_dtype, uint8_t* argmax, int N, int H, int W,
                                   int C, int k, int stride, int pad, int Ho, int Wo, rs_stream_t stream) {
  if (!x || !y || N <= 0 || H <= 0 || W <= 0 || C <= 0 || (C & 3) || k <= 0 || k > 15 || stride <= 0 || pad < 0 ||
      Ho <= 0 || Wo <= 0)
    return RS_EINVAL;
  const long total = (long)N * Ho * Wo * (C / 4);
  hipStream_t s = (hipStream_t)stream;
  if (x_dtype == RS_F32 && y_dtype == RS_F32) return launch_maxpool<float, float>(x, y, argmax, H, W, C, k, stride, pad, Ho, Wo, total, s);
  if (x_dtype == RS_F32 && y_dtype == RS_BF16) return launch_maxpool<float, bf16_t>(x, y, argmax, H, W, C, k, stride, pad, Ho, Wo, total, s);
  if (x_dtype == RS_BF16 && y_dtype == RS_BF16) return launch_maxpool<bf16_t, bf16_t>(x, y, argmax, H, W, C, k, stride, pad, Ho, Wo, total, s);
  return RS_EINVAL;
}

extern "C" int rs_maxpool2d_fwd(const float* x, float* y, uint8_t* argmax, int N, int H, int W, int C, int k, int stride,
                                int pad, int Ho, int Wo, rs_stream_t stream) {
  return rs_maxpool2d_fwd_dt(x, RS_F32, y, RS_F32, argmax, N, H, W, C, k, stride, pad, Ho, Wo, stream);
}

// dst = (float)src * scale: the way back from a bf16 gradient exchange (the all-reduced SUM of the ranks' bf16 gradient
// copies -> their fp32 mean in the optimizer's arena), 8 elements per thread.
__global__ void cast_bf16_f32_scaled_kernel(const bf16_t* __restrict__ src, float* __restrict__ dst, long n, float scale) {
  const long i = ((long)blockIdx.x * blockDim.x + threadIdx.x) * 8;
  if (i + 8 <= n) {
    const bf16x8 v = *reinterpret_cast<const bf16x8*>(src + i);
    f32x4 a, b;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      a[e] = (float)v[e] * scale;
      b[e] = (float)v[4 + e] * scale;
    }
    *reinterpret_cast<f32x4*>(dst + i) = a;
    *reinterpret_cast<f32x4*>(dst + i + 4) = b;
  } else {
    for (long j = i; j < n; ++j) dst[j] = (float)src[j] * scale;
  }
}

extern "C" int rs_cast_bf16_to_f32_scaled(const rs_bf16* src, float* dst, long n, float scale, rs_stream_t stream) {
  if (!src || !dst || n <= 0 || ((uintptr_t)src & 15) || ((uintptr_t)dst & 15)) return RS_EINVAL;
  cast_bf16_f32_scaled_kernel<<<rs_cdiv(rs_cdiv(n, 8), 256), 256, 0, (hipStream_t)stream>>>(
      reinterpret_cast<const bf16_t*>(src), dst, n, scale);
  return RS_LAUNCH_RESULT();
}

extern "C" int rs_cast_f32_to_bf16(const float* src, rs_bf16* dst, long n, rs_stream_t stream) {
  if (!src || !dst || n <= 0) return RS_EINVAL;
  cast_f32_bf16_kernel<false><<<rs_cdiv(rs_cdiv(n, 8), 256), 256, 0, (hipStream_t)stream>>>(src, reinterpret_cast<bf16_t*>(dst), n, 1.f);
  return RS_LAUNCH_RESULT();
}

extern "C" int rs_cast_f32_to_bf16_scaled(const float* src, rs_bf16* dst, long n, float scale, rs_stream_t stream) {
  if (!src || !dst || n <= 0 || ((uintptr_t)src & 15) || ((uintptr_t)dst & 15)) return RS_EINVAL;
  cast_f32_bf16_kernel<true><<<rs_cdiv(rs_cdiv(n, 8), 256), 256, 0, (hipStream_t)stream>>>(src, reinterpret_cast<bf16_t*>(dst), n, scale);
  return RS_LAUNCH_RESULT();
}

extern "C" int rs_bn_fold(const float* gamma, const float* beta, const float* mean, const float* var, float eps,
                          float* scale, float* shift, int C, rs_stream_t stream) {
  if (!gamma || !beta || !mean || !var || !scale || !shift || C <= 0) return RS_EINVAL;
  bn_fold_kernel<<<rs_cdiv(C, 256), 256, 0, (hipStream_t)stream>>>(gamma, beta, mean, var, eps, scale, shift, C);
  return RS_LAUNCH_RESULT();
}

extern "C" int rs_final_conv1x1_dt(const void* x, int x_dtype, const float* w, const float* bias, float* out, int N, int H,
                                   int W, int Cin, int C, int softmax, rs_stream_t stream) {
  if (!x || !w || !out || N <= 0 || H <= 0 || W <= 0 || Cin <= 0 || (Cin & 3) || Cin > 128 || C <= 0 || C > 8)
    return RS_EINVAL;
  const long HW = (long)H * W, P = (long)N * HW;
  hipStream_t s = (hipStream_t)stream;
  if (x_dtype == RS_F32) return dispatch_final(reinterpret_cast<const float*>(x), w, bias, out, P, HW, Cin, C, softmax, s);
  if (x_dtype == RS_BF16) return dispatch_final(reinterpret_cast<const bf16_t*>(x), w, bias, out, P, HW, Cin, C, softmax, s);
  return RS_EINVAL;
}

extern "C" int rs_final_conv1x1(const float* x, const float* w, const float* bias, float* out, int N, int H, int W, int Cin,
                                int C, int softmax, rs_stream_t stream) {
  return rs_final_conv1x1_dt(x, RS_F32, w, bias, out, N, H, W, Cin, C, softmax, stream);
}

extern "C" int rs_u8_to_nhwc4_norm(const uint8_t* img, float* out, const float* mean, const float* stdv, int N, int H, int W,
                                   int C, rs_stream_t stream) {
  if (!img || !out || !mean || !stdv || N <= 0 || H <= 0 || W <= 0 || C <= 0 || C > 4) return RS_EINVAL;
  f32x4 m = {0.f, 0.f, 0.f, 0.f}, sd = {1.f, 1.f, 1.f, 1.f};
  for (int c = 0; c < C; ++c) {
    m[c] = mean[c];  // host arrays: three or four floats
    sd[c] = stdv[c];
  }
  const long total = (long)N * H * W;
  u8_to_nhwc4_norm_kernel<<<rs_cdiv(total, 256), 256, 0, (hipStream_t)stream>>>(img, out, m, sd, C, total);
  return RS_LAUNCH_RESULT();
}

template <typename T>
int dispatch_final_bytes(const T* x, const float* w, const float* bias, long P, long HW, int Cin, int C, int mode, hipStream_t s,
                         const double* anchors, uint8_t* out, int W, int overlap) {
  float* none = nullptr;
  switch (C) {
    case 2: return launch_final<2>(x, w, bias, none, P, HW, Cin, mode, s, anchors, out, W, overlap);
    case 3: return launch_final<3>(x, w, bias, none, P, HW, Cin, mode, s, anchors, out, W, overlap);
    case 4: return launch_final<4>(x, w, bias, none, P, HW, Cin, mode, s, anchors, out, W, overlap);
    case 5: return launch_final<5>(x, w, bias, none, P, HW, Cin, mode, s, anchors, out, W, overlap);
    case 6: return launch_final<6>(x, w, bias, none, P, HW, Cin, mode, s, anchors, out, W, overlap);
    case 7: return launch_final<7>(x, w, bias, none, P, HW, Cin, mode, s, anchors, out, W, overlap);
    case 8: return launch_final<8>(x, w, bias, none, P, HW, Cin, mode, s, anchors, out, W, overlap);
    default: return RS_EINVAL;
  }
}

extern "C" int rs_final_conv1x1_quantize_dt(const void* x, int x_dtype, const float* w, const float* bias,
                                            const double* anchors, uint8_t* out, int N, int H, int W, int Cin, int C,
                                            int overlap, rs_stream_t stream) {
  if (!x || !w || !anchors || !out || N <= 0 || H <= 0 || W <= 0 || Cin <= 0 || (Cin & 3) || Cin > 128 || overlap < 0 ||
      2 * overlap >= H || 2 * overlap >= W || C < 2 || C > 8)
    return RS_EINVAL;
  const long HW = (long)H * W, P = (long)N * HW;
  hipStream_t s = (hipStream_t)stream;
  if (x_dtype == RS_F32)
    return dispatch_final_bytes(reinterpret_cast<const float*>(x), w, bias, P, HW, Cin, C, 2, s, anchors, out, W, overlap);
  if (x_dtype == RS_BF16)
    return dispatch_final_bytes(reinterpret_cast<const bf16_t*>(x), w, bias, P, HW, Cin, C, 2, s, anchors, out, W, overlap);
  return RS_EINVAL;
}

extern "C" int rs_final_conv1x1_argmax_dt(const void* x, int x_dtype, const float* w, const float* bias, uint8_t* out, int N,
                                          int H, int W, int Cin, int C, rs_stream_t stream) {
  if (!x || !w || !out || N <= 0 || H <= 0 || W <= 0 || Cin <= 0 || (Cin & 3) || Cin > 128 || C < 2 || C > 8) return RS_EINVAL;
  const long HW = (long)H * W, P = (long)N * HW;
  hipStream_t s = (hipStream_t)stream;
  if (x_dtype == RS_F32)
    return dispatch_final_bytes(reinterpret_cast<const float*>(x), w, bias, P, HW, Cin, C, 3, s, nullptr, out, W, 0);
  if (x_dtype == RS_BF16)
    return dispatch_final_bytes(reinterpret_cast<const bf16_t*>(x), w, bias, P, HW, Cin, C, 3, s, nullptr, out, W, 0);
  return RS_EINVAL;
}

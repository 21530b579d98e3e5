// Streaming (HBM-bound) backward kernels that autograd synthesises for the non-GEMM ops of UNet.forward when the
// reference calls loss.backward() (robosat/tools/train.py:186): max-pool, nearest x2 upsample (+ torch.cat split),
// and the 1x1 `final` convolution with bias.  16-byte accesses on the NHWC channel axis throughout; reductions are
// deterministic (partials + second stage, no floating-point atomics).
#include "common.h"

namespace {

// d/dx of F.max_pool2d (unet.py:125,132) in gather form: every input element sums the gradients of the windows that
// selected it (the forward kernel recorded the winning tap per window, first maximum as torch).  V = 4 or 8 channels per thread.
template <typename TY, typename TX, int V>
__global__ void maxpool_bwd_kernel(const TY* __restrict__ dy, const uint8_t* __restrict__ amax, TX* __restrict__ dx,
                                   int H, int W, int CV, int k, int stride, int pad, int Ho, int Wo, long total,
                                   int accumulate) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  const int c = (int)(i % CV);
  long pix = i / CV;
  const int ix = (int)(pix % W);
  pix /= W;
  const int iy = (int)(pix % H);
  const long n = pix / H;
  rs_vecf<V> g;
#pragma unroll
  for (int e = 0; e < V; ++e) g.v[e] = 0.f;
  if (accumulate) g = rs_ldv<V>(dx + i * V);
  int ny = iy + pad - (k - 1), nx = ix + pad - (k - 1);
  const int oy0 = ny <= 0 ? 0 : (ny + stride - 1) / stride;
  const int ox0 = nx <= 0 ? 0 : (nx + stride - 1) / stride;
  int oy1 = (iy + pad) / stride, ox1 = (ix + pad) / stride;
  if (oy1 > Ho - 1) oy1 = Ho - 1;
  if (ox1 > Wo - 1) ox1 = Wo - 1;
  for (int oy = oy0; oy <= oy1; ++oy) {
    const int r = iy + pad - oy * stride;
    for (int ox = ox0; ox <= ox1; ++ox) {
      const int s = ix + pad - ox * stride;
      const uint32_t tap = (uint32_t)(r * k + s);
      const long o = ((n * Ho + oy) * Wo + ox) * (long)CV + c;
      uint32_t sel[V / 4];
#pragma unroll
      for (int h = 0; h < V / 4; ++h) sel[h] = *reinterpret_cast<const uint32_t*>(amax + o * V + 4 * h);
      const rs_vecf<V> v = rs_ldv<V>(dy + o * V);
#pragma unroll
      for (int e = 0; e < V; ++e)
        if (((sel[e / 4] >> (8 * (e & 3))) & 0xffu) == tap) g.v[e] += v.v[e];
    }
  }
  rs_stv<V>(dx + i * V, g);
}

// d/dx of F.interpolate(scale_factor=2, mode="nearest") (unet.py:73) followed by the split of torch.cat
// (unet.py:134-137): 2x2 sum of the gradient at the upsampled resolution, channels [0,C1) -> d1, [C1,C1+C2) -> d2.
// mask1/mask2 (optional) are the ReLU outputs the gradients flow into: result zeroed where mask <= 0.
// SUM = false: `dup` is already at the source resolution [N][H][W][C1+C2] (the phase-form data gradient,
// rs_pack_dgrad_phase_weight_dt): only the torch.cat split, the ReLU masks and the accumulation remain.
template <typename T, bool SUM>
__global__ void upsample2x_bwd_kernel(const T* __restrict__ dup, T* __restrict__ d1, T* __restrict__ d2,
                                      const T* __restrict__ mask1, const T* __restrict__ mask2, int H, int W,
                                      int C1, int C2, long total, int accumulate1) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  const int Ct = C1 + C2, Q = Ct >> 2;
  const int q = (int)(i % Q);
  long pix = i / Q;  // (n*H + y)*W + x
  f32x4 s;
  if (SUM) {
    const int x = (int)(pix % W);
    const long ny = pix / W;  // n*H + y
    const long W2 = 2L * W;
    const long base = ((2 * ny) * W2 + 2 * x) * Ct + q * 4;  // row 2*(n*H+y) of the [N*2H][2W] image == n*2H + 2y
    s = rs_ld4(dup + base);
    const f32x4 b = rs_ld4(dup + base + Ct);
    const f32x4 c = rs_ld4(dup + base + W2 * Ct);
    const f32x4 d = rs_ld4(dup + base + W2 * Ct + Ct);
#pragma unroll
    for (int e = 0; e < 4; ++e) s[e] = (s[e] + b[e]) + (c[e] + d[e]);
  } else {
    s = rs_ld4(dup + pix * Ct + q * 4);
  }
  const int ch = q * 4;
  if (ch < C1) {
    const long o = pix * C1 + ch;
    if (mask1) {
      const f32x4 z = rs_ld4(mask1 + o);
#pragma unroll
      for (int e = 0; e < 4; ++e) s[e] = z[e] > 0.f ? s[e] : 0.f;
    }
    if (accumulate1) {
      const f32x4 old = rs_ld4(d1 + o);
#pragma unroll
      for (int e = 0; e < 4; ++e) s[e] += old[e];
    }
    rs_st4(d1 + o, s);
  } else {
    const long o = pix * C2 + (ch - C1);
    if (mask2) {
      const f32x4 z = rs_ld4(mask2 + o);
#pragma unroll
      for (int e = 0; e < 4; ++e) s[e] = z[e] > 0.f ? s[e] : 0.f;
    }
    rs_st4(d2 + o, s);
  }
}

// Backward of self.final (unet.py:108,141): x [P][Cin] NHWC (the ReLU output dec5), dlogits NCHW [N][C][HW].
//   dx[p][k]  = (x[p][k] > 0) * sum_c dlogits[c][p] * w[c][k]      (ReLU backward of dec5 fused: relu_mask = 1)
//   dW[c][k]  = sum_p dlogits[c][p] * x[p][k],   db[c] = sum_p dlogits[c][p]
// Blocks stride over 256-pixel tiles; per-thread partial dW/db live in registers across tiles (wave w owns pixels
// [64w, 64w+64) of each tile), then waves and blocks are combined in two deterministic stages.
template <int C, typename T>
__global__ __launch_bounds__(256) void final_bwd_kernel(const T* __restrict__ x, const float* __restrict__ w,
                                                        const float* __restrict__ dl, T* __restrict__ dx,
                                                        float* __restrict__ partial, long P, long HW, int Cin, long ntiles,
                                                        int relu_mask) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  const int ld = Cin + 1;
  float* xs = sm;                 // [256][Cin+1]
  float* dls = xs + 256 * ld;     // [C][256]
  float* ws = dls + C * 256;      // [C][Cin]
  float* red = ws + C * Cin;      // [4][npairs]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int q = Cin >> 2;
  const int npairs = C * (Cin + 1);  // (c,k) for k < Cin, plus the bias column k == Cin
  constexpr int MAXPASS = 9;         // C*(Cin+1) <= 8*65 = 520 <= 9*64
  float acc[MAXPASS];
#pragma unroll
  for (int i = 0; i < MAXPASS; ++i) acc[i] = 0.f;

  for (int f = tid; f < C * Cin; f += 256) ws[f] = w[f];

  for (long tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    const long p0 = tile * 256;
    __syncthreads();  // previous tile fully consumed
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
    {
      const long pix = p0 + tid;
      const long n = pix / HW, hw = pix - n * HW;
#pragma unroll
      for (int c = 0; c < C; ++c) dls[c * 256 + tid] = pix < P ? dl[(n * C + c) * HW + hw] : 0.f;
    }
    __syncthreads();
    // dx: thread -> (pixel, float4 of channels); coalesced 16-byte stores
    for (int f = tid; f < 256 * q; f += 256) {
      const int px = f / q, c4 = f - px * q;
      if (p0 + px >= P) continue;
      f32x4 o = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int c = 0; c < C; ++c) {
        const float g = dls[c * 256 + px];
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = fmaf(g, ws[c * Cin + c4 * 4 + e], o[e]);
      }
      if (relu_mask) {
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = xs[px * ld + c4 * 4 + e] > 0.f ? o[e] : 0.f;
      }
      rs_st4(dx + (p0 + px) * Cin + c4 * 4, o);
    }
    // dW/db partials: wave `wave` reduces its 64 pixels for every (c,k) pair
#pragma unroll
    for (int pass = 0; pass < MAXPASS; ++pass) {
      const int pair = pass * 64 + lane;
      if (pair < npairs) {
        const int c = pair / (Cin + 1), k = pair - c * (Cin + 1);
        float a = acc[pass];
        const float* g = dls + c * 256 + wave * 64;
        if (k < Cin) {
          const float* xv = xs + (wave * 64) * ld + k;
          for (int pp = 0; pp < 64; ++pp) a = fmaf(g[pp], xv[pp * ld], a);
        } else {
          for (int pp = 0; pp < 64; ++pp) a += g[pp];
        }
        acc[pass] = a;
      }
    }
  }
  __syncthreads();
#pragma unroll
  for (int pass = 0; pass < MAXPASS; ++pass) {
    const int pair = pass * 64 + lane;
    if (pair < npairs) red[wave * npairs + pair] = acc[pass];
  }
  __syncthreads();
  for (int pair = tid; pair < npairs; pair += 256)
    partial[(long)blockIdx.x * npairs + pair] = (red[pair] + red[npairs + pair]) + (red[2 * npairs + pair] + red[3 * npairs + pair]);
}

// The model's own shape, Cin = 32 and C <= 4 classes: one thread = one pixel per tile, the pixel's 32 channels and C
// logit gradients come straight into registers as 16-byte / coalesced 4-byte loads (the next tile's are requested before
// this tile's arithmetic), dx leaves as 16-byte stores, and the thread's own dW / db contributions accumulate in registers
// over all its tiles; waves, then blocks, are combined in a fixed order (same partial layout as the general kernel).
template <int C, typename T>
__global__ __launch_bounds__(256) void final_bwd_px32_kernel(const T* __restrict__ x, const float* __restrict__ w,
                                                             const float* __restrict__ dl, T* __restrict__ dx,
                                                             float* __restrict__ partial, long P, long HW, long ntiles,
                                                             int relu_mask) {
  constexpr int NP = C * 33;
  __shared__ __attribute__((aligned(16))) float ws[C * 32];
  __shared__ float red[4 * NP];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  for (int f = tid; f < C * 32; f += 256) ws[f] = w[f];
  __syncthreads();
  float aw[C][32], ab[C];
#pragma unroll
  for (int c = 0; c < C; ++c) {
    ab[c] = 0.f;
#pragma unroll
    for (int k = 0; k < 32; ++k) aw[c][k] = 0.f;
  }
  float xv[32], g[C];
  auto fetch = [&](long tile, float (&xr)[32], float (&gr)[C]) {
    const long pix = tile * 256 + tid;
    if (pix < P) {
      rs_ld_row32(x + pix * 32, xr);
      const long n = pix / HW, hw = pix - n * HW;
#pragma unroll
      for (int c = 0; c < C; ++c) gr[c] = dl[(n * C + c) * HW + hw];
    } else {
#pragma unroll
      for (int k = 0; k < 32; ++k) xr[k] = 0.f;
#pragma unroll
      for (int c = 0; c < C; ++c) gr[c] = 0.f;
    }
  };
  long tile = blockIdx.x;
  if (tile < ntiles) fetch(tile, xv, g);
  while (tile < ntiles) {
    float xn[32], gn[C];
    const long next = tile + gridDim.x;
    if (next < ntiles) fetch(next, xn, gn);
    const long pix = tile * 256 + tid;
    float o[32];
#pragma unroll
    for (int k4 = 0; k4 < 8; ++k4) {
      f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int c = 0; c < C; ++c) {
        const f32x4 wv = *reinterpret_cast<const f32x4*>(ws + c * 32 + k4 * 4);
#pragma unroll
        for (int e = 0; e < 4; ++e) acc[e] = fmaf(g[c], wv[e], acc[e]);
      }
#pragma unroll
      for (int e = 0; e < 4; ++e) o[k4 * 4 + e] = (relu_mask && !(xv[k4 * 4 + e] > 0.f)) ? 0.f : acc[e];
    }
    if (pix < P) rs_st_row32(dx + pix * 32, o);
#pragma unroll
    for (int c = 0; c < C; ++c) {
      ab[c] += g[c];
#pragma unroll
      for (int k = 0; k < 32; ++k) aw[c][k] = fmaf(g[c], xv[k], aw[c][k]);
    }
    if (next < ntiles) {
#pragma unroll
      for (int k = 0; k < 32; ++k) xv[k] = xn[k];
#pragma unroll
      for (int c = 0; c < C; ++c) g[c] = gn[c];
    }
    tile = next;
  }
#pragma unroll
  for (int c = 0; c < C; ++c) {
#pragma unroll
    for (int k = 0; k < 32; ++k) {
      const float v = rs_wave_sum(aw[c][k]);
      if (lane == 0) red[wave * NP + c * 33 + k] = v;
    }
    const float v = rs_wave_sum(ab[c]);
    if (lane == 0) red[wave * NP + c * 33 + 32] = v;
  }
  __syncthreads();
  for (int pair = tid; pair < NP; pair += 256)
    partial[(long)blockIdx.x * NP + pair] = (red[pair] + red[NP + pair]) + (red[2 * NP + pair] + red[3 * NP + pair]);
}

// one block per (c,k) pair: sums the per-block partials in fp64
__global__ __launch_bounds__(256) void final_bwd_finalize_kernel(const float* __restrict__ partial, int nblocks, int C,
                                                                 int Cin, float* __restrict__ dw, float* __restrict__ db) {
  __shared__ double red[256];
  const int pair = blockIdx.x, npairs = C * (Cin + 1);
  double s = 0;
  for (int b = threadIdx.x; b < nblocks; b += 256) s += (double)partial[(long)b * npairs + pair];
  red[threadIdx.x] = s;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if (threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    const int c = pair / (Cin + 1), k = pair - c * (Cin + 1);
    if (k < Cin) dw[c * Cin + k] = (float)red[0];
    else db[c] = (float)red[0];
  }
}

constexpr int kFinalBwdMaxBlocks = 1024;

template <int C, typename T>
int launch_final_bwd(const T* x, const float* w, const float* dl, T* dx, float* dw, float* db, float* partial,
                     long P, long HW, int Cin, int relu_mask, hipStream_t s) {
  const long ntiles = (P + 255) / 256;
  const int grid = ntiles < kFinalBwdMaxBlocks ? (int)ntiles : kFinalBwdMaxBlocks;
  const int npairs = C * (Cin + 1);
  if constexpr (C <= 4) {
    if (Cin == 32) {
      final_bwd_px32_kernel<C, T><<<grid, 256, 0, s>>>(x, w, dl, dx, partial, P, HW, ntiles, relu_mask);
      final_bwd_finalize_kernel<<<npairs, 256, 0, s>>>(partial, grid, C, Cin, dw, db);
      return RS_LAUNCH_RESULT();
    }
  }
  const size_t smem = (size_t)(256 * (Cin + 1) + C * 256 + C * Cin + 4 * npairs) * sizeof(float);
  if (smem > 64 * 1024) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&final_bwd_kernel<C, T>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    if (e != hipSuccess) return (int)e;
  }
  final_bwd_kernel<C, T><<<grid, 256, smem, s>>>(x, w, dl, dx, partial, P, HW, Cin, ntiles, relu_mask);
  final_bwd_finalize_kernel<<<npairs, 256, 0, s>>>(partial, grid, C, Cin, dw, db);
  return RS_LAUNCH_RESULT();
}

template <typename T>
int dispatch_final_bwd(const T* x, const float* w, const float* dl, T* dx, float* dw, float* db, float* part, long P, long HW,
                       int Cin, int C, int relu_mask, hipStream_t s) {
  switch (C) {
    case 1: return launch_final_bwd<1>(x, w, dl, dx, dw, db, part, P, HW, Cin, relu_mask, s);
    case 2: return launch_final_bwd<2>(x, w, dl, dx, dw, db, part, P, HW, Cin, relu_mask, s);
    case 3: return launch_final_bwd<3>(x, w, dl, dx, dw, db, part, P, HW, Cin, relu_mask, s);
    case 4: return launch_final_bwd<4>(x, w, dl, dx, dw, db, part, P, HW, Cin, relu_mask, s);
    case 5: return launch_final_bwd<5>(x, w, dl, dx, dw, db, part, P, HW, Cin, relu_mask, s);
    case 6: return launch_final_bwd<6>(x, w, dl, dx, dw, db, part, P, HW, Cin, relu_mask, s);
    case 7: return launch_final_bwd<7>(x, w, dl, dx, dw, db, part, P, HW, Cin, relu_mask, s);
    default: return launch_final_bwd<8>(x, w, dl, dx, dw, db, part, P, HW, Cin, relu_mask, s);
  }
}

template <typename TY, typename TX>
int launch_maxpool_bwd(const void* dy, const uint8_t* argmax, void* dx, int H, int W, int C, int k, int stride, int pad,
                       int Ho, int Wo, long total, int accumulate, hipStream_t s) {
  if ((C & 7) == 0)
    maxpool_bwd_kernel<TY, TX, 8><<<rs_cdiv(total / 2, 256), 256, 0, s>>>(reinterpret_cast<const TY*>(dy), argmax,
                                                                          reinterpret_cast<TX*>(dx), H, W, C / 8, k, stride, pad,
                                                                          Ho, Wo, total / 2, accumulate);
  else
    maxpool_bwd_kernel<TY, TX, 4><<<rs_cdiv(total, 256), 256, 0, s>>>(reinterpret_cast<const TY*>(dy), argmax,
                                                                      reinterpret_cast<TX*>(dx), H, W, C / 4, k, stride, pad, Ho,
                                                                      Wo, total, accumulate);
  return RS_LAUNCH_RESULT();
}

template <typename T, bool SUM>
int launch_upsample_bwd(const void* dup, void* d1, void* d2, const void* mask1, const void* mask2, int H, int W, int C1,
                        int C2, long total, int accumulate1, hipStream_t s) {
  upsample2x_bwd_kernel<T, SUM><<<rs_cdiv(total, 256), 256, 0, s>>>(
      reinterpret_cast<const T*>(dup), reinterpret_cast<T*>(d1), reinterpret_cast<T*>(d2), reinterpret_cast<const T*>(mask1),
      reinterpret_cast<const T*>(mask2), H, W, C1, C2, total, accumulate1);
  return RS_LAUNCH_RESULT();
}


// out[n][2 a][2 b][:] += t[n][a][b][:] -- the data gradient of a 1x1 / stride-2 convolution (torchvision Bottleneck.downsample[0] of
// layer2..layer4 under tools/train.py:186) is its transposed 1x1 product on the LOW-resolution grid, landing on the even positions of the
// input grid and nowhere else; the caller's `out` already holds the other gradient of that tensor (the decoder's skip branch).  Round 6:
// as a zero-insertion convolution three of four GEMM rows were zeros (0.25-0.26 ms per fp32 launch against 0.075 for the product).
template <typename T, int V>
__global__ void scatter_add_stride2_kernel(const T* __restrict__ t, T* __restrict__ out, int Hs, int Ws, int Ho, int Wo, int Q, long total) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;  // over [N][Hs][Ws][Q pieces of V channels]
  if (i >= total) return;
  const int q = (int)(i % Q);
  long r = i / Q;
  const int b = (int)(r % Ws);
  r /= Ws;
  const int a = (int)(r % Hs);
  const long n = r / Hs;
  T* o = out + (((n * Ho + 2 * a) * Wo + 2 * b) * (long)Q + q) * V;
  if constexpr (sizeof(T) == 4) {
    f32x4 x = *reinterpret_cast<const f32x4*>(t + i * V), y = *reinterpret_cast<const f32x4*>(o);
    *reinterpret_cast<f32x4*>(o) = y + x;
  } else {
    const bf16x8 x = *reinterpret_cast<const bf16x8*>(t + i * V), y = *reinterpret_cast<const bf16x8*>(o);
    bf16x8 z;
#pragma unroll
    for (int e = 0; e < 8; ++e) z[e] = (bf16_t)((float)y[e] + (float)x[e]);
    *reinterpret_cast<bf16x8*>(o) = z;
  }
}

}  // namespace

extern "C" int rs_scatter_add_stride2_dt(const void* t, void* out, int dtype, int N, int Hs, int Ws, int Ho, int Wo, int C, rs_stream_t stream) {
  if (!t || !out || N <= 0 || Hs <= 0 || Ws <= 0 || C <= 0 || 2 * Hs - 1 > Ho || 2 * Ws - 1 > Wo) return RS_EINVAL;
  hipStream_t s = (hipStream_t)stream;
  if (dtype == RS_F32 && (C & 3) == 0) {
    const long total = (long)N * Hs * Ws * (C / 4);
    scatter_add_stride2_kernel<float, 4><<<rs_cdiv(total, 256), 256, 0, s>>>(reinterpret_cast<const float*>(t), reinterpret_cast<float*>(out), Hs, Ws, Ho, Wo, C / 4, total);
    return RS_LAUNCH_RESULT();
  }
  if (dtype == RS_BF16 && (C & 7) == 0) {
    const long total = (long)N * Hs * Ws * (C / 8);
    scatter_add_stride2_kernel<bf16_t, 8><<<rs_cdiv(total, 256), 256, 0, s>>>(reinterpret_cast<const bf16_t*>(t), reinterpret_cast<bf16_t*>(out), Hs, Ws, Ho, Wo, C / 8, total);
    return RS_LAUNCH_RESULT();
  }
  return RS_EINVAL;
}

namespace {

}  // namespace

extern "C" int rs_maxpool2d_bwd_dt(const void* dy, int dy_dtype, const uint8_t* argmax, void* dx, int dx_dtype, int N, int H,
                                   int W, int C, int k, int stride, int pad, int Ho, int Wo, int accumulate,
                                   rs_stream_t stream) {
  if (!dy || !argmax || !dx || N <= 0 || H <= 0 || W <= 0 || C <= 0 || (C & 3) || k <= 0 || k > 15 || stride <= 0 ||
      pad < 0 || Ho <= 0 || Wo <= 0)
    return RS_EINVAL;
  const long total = (long)N * H * W * (C / 4);
  hipStream_t s = (hipStream_t)stream;
  if (dy_dtype == RS_F32 && dx_dtype == RS_F32) return launch_maxpool_bwd<float, float>(dy, argmax, dx, H, W, C, k, stride, pad, Ho, Wo, total, accumulate, s);
  if (dy_dtype == RS_BF16 && dx_dtype == RS_BF16) return launch_maxpool_bwd<bf16_t, bf16_t>(dy, argmax, dx, H, W, C, k, stride, pad, Ho, Wo, total, accumulate, s);
  if (dy_dtype == RS_BF16 && dx_dtype == RS_F32) return launch_maxpool_bwd<bf16_t, float>(dy, argmax, dx, H, W, C, k, stride, pad, Ho, Wo, total, accumulate, s);
  return RS_EINVAL;
}

extern "C" int rs_maxpool2d_bwd(const float* dy, const uint8_t* argmax, float* dx, int N, int H, int W, int C, int k,
                                int stride, int pad, int Ho, int Wo, int accumulate, rs_stream_t stream) {
  return rs_maxpool2d_bwd_dt(dy, RS_F32, argmax, dx, RS_F32, N, H, W, C, k, stride, pad, Ho, Wo, accumulate, stream);
}

extern "C" int rs_upsample2x_bwd_dt(const void* dup, void* d1, void* d2, const void* mask1, const void* mask2, int dtype,
                                    int N, int H, int W, int C1, int C2, int accumulate1, rs_stream_t stream) {
  if (!dup || !d1 || N <= 0 || H <= 0 || W <= 0 || C1 <= 0 || (C1 & 3) || C2 < 0 || (C2 & 3)) return RS_EINVAL;
  if (C2 > 0 && !d2) return RS_EINVAL;
  const long total = (long)N * H * W * ((C1 + C2) / 4);
  hipStream_t s = (hipStream_t)stream;
  if (dtype == RS_F32) return launch_upsample_bwd<float, true>(dup, d1, d2, mask1, mask2, H, W, C1, C2, total, accumulate1, s);
  if (dtype == RS_BF16) return launch_upsample_bwd<bf16_t, true>(dup, d1, d2, mask1, mask2, H, W, C1, C2, total, accumulate1, s);
  return RS_EINVAL;
}

extern "C" int rs_cat_split_bwd_dt(const void* dcat, void* d1, void* d2, const void* mask1, const void* mask2, int dtype,
                                   int N, int H, int W, int C1, int C2, int accumulate1, rs_stream_t stream) {
  if (!dcat || !d1 || N <= 0 || H <= 0 || W <= 0 || C1 <= 0 || (C1 & 3) || C2 < 0 || (C2 & 3)) return RS_EINVAL;
  if (C2 > 0 && !d2) return RS_EINVAL;
  const long total = (long)N * H * W * ((C1 + C2) / 4);
  hipStream_t s = (hipStream_t)stream;
  if (dtype == RS_F32) return launch_upsample_bwd<float, false>(dcat, d1, d2, mask1, mask2, H, W, C1, C2, total, accumulate1, s);
  if (dtype == RS_BF16) return launch_upsample_bwd<bf16_t, false>(dcat, d1, d2, mask1, mask2, H, W, C1, C2, total, accumulate1, s);
  return RS_EINVAL;
}

extern "C" int rs_upsample2x_bwd(const float* dup, float* d1, float* d2, const float* mask1, const float* mask2, int N,
                                 int H, int W, int C1, int C2, int accumulate1, rs_stream_t stream) {
  return rs_upsample2x_bwd_dt(dup, d1, d2, mask1, mask2, RS_F32, N, H, W, C1, C2, accumulate1, stream);
}

extern "C" long rs_final_conv1x1_bwd_workspace_bytes(int Cin, int C) {
  if (Cin <= 0 || Cin > 64 || C <= 0 || C > 8) return RS_EINVAL;
  return (long)kFinalBwdMaxBlocks * C * (Cin + 1) * (long)sizeof(float);
}

extern "C" int rs_final_conv1x1_bwd_dt(const void* x, const float* w, const float* dlogits, void* dx, float* dw, float* db,
                                       int dtype, int N, int H, int W, int Cin, int C, int relu_mask, void* workspace,
                                       rs_stream_t stream) {
  if (!x || !w || !dlogits || !dx || !dw || !db || !workspace || N <= 0 || H <= 0 || W <= 0 || Cin <= 0 || (Cin & 3) ||
      Cin > 64 || C <= 0 || C > 8)
    return RS_EINVAL;
  const long HW = (long)H * W, P = (long)N * HW;
  hipStream_t s = (hipStream_t)stream;
  float* part = reinterpret_cast<float*>(workspace);
  if (dtype == RS_F32)
    return dispatch_final_bwd(reinterpret_cast<const float*>(x), w, dlogits, reinterpret_cast<float*>(dx), dw, db, part, P, HW,
                              Cin, C, relu_mask, s);
  if (dtype == RS_BF16)
    return dispatch_final_bwd(reinterpret_cast<const bf16_t*>(x), w, dlogits, reinterpret_cast<bf16_t*>(dx), dw, db, part, P,
                              HW, Cin, C, relu_mask, s);
  return RS_EINVAL;
}

extern "C" int rs_final_conv1x1_bwd(const float* x, const float* w, const float* dlogits, float* dx, float* dw, float* db,
                                    int N, int H, int W, int Cin, int C, int relu_mask, void* workspace,
                                    rs_stream_t stream) {
  return rs_final_conv1x1_bwd_dt(x, w, dlogits, dx, dw, db, RS_F32, N, H, W, Cin, C, relu_mask, workspace, stream);
}

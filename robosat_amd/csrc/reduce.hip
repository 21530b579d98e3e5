// Deterministic sum of split-P partial weight-gradient tiles: out[i] = sum_k ws[k][i], k < splits (fp32, fixed order).
// Shared by the fp32 and bf16 weight-gradient kernels (conv_wgrad*.hip).  With many splits over a small tile the
// one-thread-per-column loop is a serial chain over `splits` dependent HBM round trips on a handful of CUs, so the sum is
// done in two levels: L lanes per column each add splits l, l+L, ... into scratch[l][i], then the L lane sums are added.
#include "common.h"

namespace {

__global__ void reduce_lanes_kernel(const float* __restrict__ ws, float* __restrict__ dst, long n4, int splits, int L) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const int l = blockIdx.y;
  if (i >= n4) return;
  f32x4 s = {0.f, 0.f, 0.f, 0.f};
  int k = l;
  for (; k + 3 * L < splits; k += 4 * L) {  // 4 independent loads in flight
    const f32x4 a = *reinterpret_cast<const f32x4*>(ws + ((long)k * n4 + i) * 4);
    const f32x4 b = *reinterpret_cast<const f32x4*>(ws + ((long)(k + L) * n4 + i) * 4);
    const f32x4 c = *reinterpret_cast<const f32x4*>(ws + ((long)(k + 2 * L) * n4 + i) * 4);
    const f32x4 d = *reinterpret_cast<const f32x4*>(ws + ((long)(k + 3 * L) * n4 + i) * 4);
#pragma unroll
    for (int e = 0; e < 4; ++e) s[e] = (((s[e] + a[e]) + b[e]) + c[e]) + d[e];
  }
  for (; k < splits; k += L) {
    const f32x4 a = *reinterpret_cast<const f32x4*>(ws + ((long)k * n4 + i) * 4);
#pragma unroll
    for (int e = 0; e < 4; ++e) s[e] += a[e];
  }
  *reinterpret_cast<f32x4*>(dst + ((long)l * n4 + i) * 4) = s;
}

}  // namespace

long rs_reduce_scratch_floats(long n, int splits) { return splits > 8 ? 32 * n : 0; }

int rs_reduce_splits(const float* ws, float* out, long n, int splits, float* scratch, void* stream) {
  hipStream_t s = (hipStream_t)stream;
  const long n4 = n / 4;
  const int gx = rs_cdiv(n4, 256);
  if (splits <= 8 || !scratch) {
    reduce_lanes_kernel<<<dim3(gx, 1), 256, 0, s>>>(ws, out, n4, splits, 1);
    return RS_LAUNCH_RESULT();
  }
  long L = 131072 / n4;
  if (L > 32) L = 32;
  if (L > splits / 2) L = splits / 2;
  if (L < 2) L = 2;
  reduce_lanes_kernel<<<dim3(gx, (int)L), 256, 0, s>>>(ws, scratch, n4, splits, (int)L);
  reduce_lanes_kernel<<<dim3(gx, 1), 256, 0, s>>>(scratch, out, n4, (int)L, 1);
  return RS_LAUNCH_RESULT();
}

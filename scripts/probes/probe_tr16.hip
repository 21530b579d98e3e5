#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef short s16x4 __attribute__((ext_vector_type(4)));
__global__ void k(const unsigned short* in, unsigned short* out, const int* addr) {
  __shared__ __attribute__((aligned(16))) unsigned short lds[4096];
  for (int i = threadIdx.x; i < 4096; i += 64) lds[i] = in[i];
  __syncthreads();
  int a = addr[threadIdx.x];
  s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(lds + a));
  for (int j = 0; j < 4; ++j) out[threadIdx.x * 4 + j] = (unsigned short)v[j];
}
int main() {
  std::vector<unsigned short> h(4096); for (int i = 0; i < 4096; ++i) h[i] = i;
  unsigned short *din, *dout; int* daddr;
  hipMalloc(&din, 8192); hipMalloc(&dout, 512); hipMalloc(&daddr, 256);
  hipMemcpy(din, h.data(), 8192, hipMemcpyHostToDevice);
  for (int mode = 0; mode < 3; ++mode) {
    std::vector<int> a(64);
    for (int l = 0; l < 64; ++l) {
      if (mode == 0) a[l] = l * 4;                 // linear 8 B per lane
      if (mode == 1) a[l] = 0;                     // uniform
      if (mode == 2) a[l] = (l & 3) * 4 + ((l >> 2) & 3) * 100 + (l >> 4) * 1000;  // rows of 4 lanes at stride 100
    }
    hipMemcpy(daddr, a.data(), 256, hipMemcpyHostToDevice);
    k<<<1, 64>>>(din, dout, daddr);
    std::vector<unsigned short> o(256);
    hipMemcpy(o.data(), dout, 512, hipMemcpyDeviceToHost);
    printf("mode %d\n", mode);
    for (int l = 0; l < 64; ++l) printf("l%2d: %4d %4d %4d %4d\n", l, o[l*4], o[l*4+1], o[l*4+2], o[l*4+3]);
  }
  return 0;
}

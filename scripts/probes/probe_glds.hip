// Probe: buffer_load_dwordx4 ... lds (LDS-DMA) semantics on gfx950 -- destination = wave-uniform base + lane*16,
// and what lands in LDS for lanes whose buffer offset is out of range (expected: zeros).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void k(const unsigned int* in, unsigned int* out, const int* offs, int nbytes) {
  __shared__ __attribute__((aligned(16))) unsigned int lds[2048];
  for (int i = threadIdx.x; i < 2048; i += 64) lds[i] = 0xDEADBEEFu;
  __syncthreads();
  __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned int*>(in), 0, nbytes, 0x00020000);
  __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (__attribute__((address_space(3))) void*)(lds + 256), 16, offs[threadIdx.x], 0, 0, 0);
  __builtin_amdgcn_s_waitcnt(0);
  __syncthreads();
  for (int i = threadIdx.x; i < 2048; i += 64) out[i] = lds[i];
}
int main() {
  std::vector<unsigned int> h(4096); for (int i = 0; i < 4096; ++i) h[i] = i;
  unsigned int *din, *dout; int* doff;
  (void)hipMalloc(&din, 16384); (void)hipMalloc(&dout, 8192); (void)hipMalloc(&doff, 256);
  (void)hipMemcpy(din, h.data(), 16384, hipMemcpyHostToDevice);
  std::vector<int> a(64);
  for (int l = 0; l < 64; ++l) a[l] = (l % 5 == 0) ? -1 : ((63 - l) * 32);   // reversed, strided; every 5th lane OOB
  (void)hipMemcpy(doff, a.data(), 256, hipMemcpyHostToDevice);
  k<<<1, 64>>>(din, dout, doff, 8192);
  std::vector<unsigned int> o(2048);
  (void)hipMemcpy(o.data(), dout, 8192, hipMemcpyDeviceToHost);
  int bad = 0;
  for (int l = 0; l < 64; ++l) {
    for (int j = 0; j < 4; ++j) {
      unsigned int want = (l % 5 == 0) ? 0u : (unsigned)((63 - l) * 8 + j);
      unsigned int got = o[256 + l * 4 + j];
      if (got != want) { if (bad < 10) printf("lane %d dw %d: got %08x want %08x\n", l, j, got, want); ++bad; }
    }
  }
  int touched = 0;
  for (int i = 0; i < 2048; ++i) if ((i < 256 || i >= 512) && o[i] != 0xDEADBEEFu) ++touched;
  printf("glds probe: mismatches %d, stray writes %d\n", bad, touched);
  return 0;
}

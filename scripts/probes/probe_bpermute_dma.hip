// Probe: is ds_bpermute_b32 (what __shfl_xor compiles to) reliable while LDS-DMA pieces (buffer_load_dwordx4 ... lds) are IN FLIGHT?
// Round 6 found the fused dec5 + final head (conv_wino33_f32.hip) summing its 16 couts with two __shfl_xor steps while the next item's first
// chunk was still streaming into LDS: beside a neighbour that delays that stream, the lanes that read lanes 48-63 got stale values in
// hundreds of pixels per launch; on the VALU (v_permlane16/32_swap) the same sums are always right.  This probe isolates the pair:
//   every wave: NP cold LDS-DMA pieces (HBM misses) issued, NOT waited for; then R butterfly rounds of v += bpermute(v, lane ^ 16),
//   v += bpermute(v, lane ^ 32) on values whose correct result is known; then the drain.  Variants: whose DMA is in flight
//   (0 nobody's, 1 the shuffling wave's own, 2 only the OTHER waves of the block), and the same butterflies on v_permlane swaps.
// Build: hipcc --offload-arch=gfx950 -O2 -Wno-inline-asm -o scripts/probes/probe_bpermute_dma.bin scripts/probes/probe_bpermute_dma.hip
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x)                                                                               \
  do {                                                                                      \
    hipError_t e_ = (x);                                                                    \
    if (e_ != hipSuccess) {                                                                 \
      printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__);         \
      exit(2);                                                                              \
    }                                                                                       \
  } while (0)

__device__ __forceinline__ void dma16(__amdgpu_buffer_rsrc_t r, unsigned int lds_dst, int voff, int soff) {
  asm volatile(
      "s_mov_b32 m0, %1\n\t"
      "s_nop 0\n\t"
      "buffer_load_dwordx4 %0, %2, %3 offen lds"
      :
      : "v"(voff), "s"(lds_dst), "s"(r), "s"(soff)
      : "memory", "m0");
}
__device__ __forceinline__ __amdgpu_buffer_rsrc_t mk(const void* base, unsigned int bytes) {
  const unsigned long b = (unsigned long)base;
  const unsigned int lo = __builtin_amdgcn_readfirstlane((unsigned int)b), hi = __builtin_amdgcn_readfirstlane((unsigned int)(b >> 32));
  return __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<void*>(((unsigned long)hi << 32) | lo), 0, (int)__builtin_amdgcn_readfirstlane(bytes),
                                           0x00020000);
}
__device__ __forceinline__ float bperm(float v, int src_lane) {
  return __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(src_lane << 2, __builtin_bit_cast(int, v)));
}
__device__ __forceinline__ float swap16(float v) {
  float a = v, b = v;
  asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1" : "+v"(a), "+v"(b));
  return a + b;
}
__device__ __forceinline__ float swap32(float v) {
  float a = v, b = v;
  asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(a), "+v"(b));
  return a + b;
}

// WHO: 0 nobody issues DMA, 1 every wave (the shuffling waves have their own pieces in flight), 2 only waves 4-7 issue, waves 0-3 shuffle
// VALU: 0 ds_bpermute butterflies, 1 v_permlane swaps
template <int WHO, int VALU, int NP, int R>
__global__ __launch_bounds__(512) void probe(const unsigned int* big, unsigned int big_kib, unsigned int round, unsigned int* bad) {
  __shared__ __attribute__((aligned(16))) unsigned int lds[8 * NP * 256 + 256];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const unsigned int lds0 = (unsigned int)(unsigned long)(__attribute__((address_space(3))) void*)lds;
  const bool issues = WHO == 1 || (WHO == 2 && wave >= 4);
  const bool shuffles = WHO != 2 || wave < 4;
  if (issues) {
#pragma unroll
    for (int p = 0; p < NP; ++p) {
      const unsigned long long id = (((unsigned long long)round * gridDim.x + blockIdx.x) * 8 + wave) * NP + p;
      const unsigned int kib = (unsigned int)((id * 2654435761ull) % big_kib);
      dma16(mk(big + (size_t)kib * 256, 1024), lds0 + (wave * NP + p) * 1024, lane * 16, 0);
    }
  }
  unsigned int wrong = 0;
  if (shuffles) {
    // lane l starts from 2^(l >> 4) * (1 + (l & 15)) + r: the butterfly over lane ^ 16, lane ^ 32 must give 15 * (1 + (l & 15)) + 4 r in every lane
#pragma unroll 4
    for (int r = 0; r < R; ++r) {
      float v = (float)((1 << (lane >> 4)) * (1 + (lane & 15)) + r);
      asm volatile("" : "+v"(v));
      if (VALU) {
        v = swap32(swap16(v));
      } else {
        v += bperm(v, lane ^ 16);
        v += bperm(v, lane ^ 32);
      }
      wrong += v != (float)(15 * (1 + (lane & 15)) + 4 * r) ? 1u : 0u;
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  const unsigned long long m = __builtin_amdgcn_ballot_w64(wrong != 0);
  if (lane == 0 && m) atomicAdd(&bad[0], 1u);                                            // waves with any wrong lane
  if (wrong) atomicAdd(&bad[1 + (lane >> 4)], wrong);                                     // wrong results by lane quarter
  if (lds[tid] == 0x7654321u) bad[7] = 1;  // (keeps the LDS alive)
}

__global__ void fill(unsigned int* p, size_t n) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = (unsigned int)i;
}

template <int WHO, int VALU, int NP, int R>
void run(const char* name, const unsigned int* big, unsigned int big_kib, int rounds, int blocks, unsigned int& round) {
  unsigned int* d;
  CK(hipMalloc(&d, 32));
  CK(hipMemset(d, 0, 32));
  for (int i = 0; i < rounds; ++i) probe<WHO, VALU, NP, R><<<blocks, 512>>>(big, big_kib, round++, d);
  CK(hipDeviceSynchronize());
  unsigned int h[8];
  CK(hipMemcpy(h, d, 32, hipMemcpyDeviceToHost));
  printf("%-74s waves with a wrong sum %8u of %9ld | wrong sums by lane quarter %u %u %u %u\n", name, h[0], (long)rounds * blocks * (WHO == 2 ? 4 : 8), h[1], h[2],
         h[3], h[4]);
  fflush(stdout);
  CK(hipFree(d));
}

int main(int argc, char** argv) {
  const int rounds = argc > 1 ? atoi(argv[1]) : 20;
  const int blocks = argc > 2 ? atoi(argv[2]) : 1024;
  const size_t big_bytes = (size_t)4 << 30;
  unsigned int* big;
  CK(hipMalloc(&big, big_bytes));
  fill<<<4096, 256>>>(big, big_bytes / 4);
  CK(hipDeviceSynchronize());
  const unsigned int big_kib = (unsigned int)(big_bytes >> 10);
  unsigned int round = 1;
  printf("probe_bpermute_dma: %d rounds x %d blocks of 8 waves, 64 butterfly rounds per wave\n", rounds, blocks);
  run<0, 0, 4, 64>("ds_bpermute, no LDS-DMA anywhere", big, big_kib, rounds, blocks, round);
  run<1, 0, 4, 64>("ds_bpermute, 4 cold LDS-DMA pieces of the SAME wave in flight", big, big_kib, rounds, blocks, round);
  run<1, 0, 16, 64>("ds_bpermute, 16 cold LDS-DMA pieces of the SAME wave in flight", big, big_kib, rounds, blocks, round);
  run<2, 0, 16, 64>("ds_bpermute in waves 0-3, 16 cold pieces each of waves 4-7 in flight", big, big_kib, rounds, blocks, round);
  run<1, 1, 16, 64>("v_permlane16/32_swap, 16 cold LDS-DMA pieces of the SAME wave in flight", big, big_kib, rounds, blocks, round);
  run<0, 1, 4, 64>("v_permlane16/32_swap, no LDS-DMA anywhere", big, big_kib, rounds, blocks, round);
  return 0;
}

// A co-resident "neighbour" for race screens: `lds_bytes` of dynamic LDS per 256-thread block, `iters` loop rounds of
// mode 0: s_sleep only   1: ds_read / ds_write traffic   2: (unused)   3: bf16 MFMA loop   4: LDS-DMA stream (needs >= 32 KB of LDS)   5: packed-VALU loop
#include <hip/hip_runtime.h>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void dma16(__amdgpu_buffer_rsrc_t r, unsigned int lds_dst, int voff, int soff) {
  asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %2, %3 offen lds" : : "v"(voff), "s"(lds_dst), "s"(r), "s"(soff) : "memory", "m0");
}
__global__ __launch_bounds__(256) void dummy(int iters, int mode, unsigned int* sink, const unsigned int* src, unsigned int src_bytes) {
  extern __shared__ __attribute__((aligned(16))) unsigned int lds[];
  unsigned int v = threadIdx.x;
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  if (mode == 1) {
    for (int i = 0; i < iters; ++i) {
      lds[(threadIdx.x * 4 + i * 64) & 1023] = v;
      v += lds[(threadIdx.x * 5 + i) & 1023];
    }
  } else if (mode == 3) {
    f32x16 acc = {};
    bf16x8 fa, fb;
    for (int e = 0; e < 8; ++e) { fa[e] = (__bf16)(lane * 0.001f + e); fb[e] = (__bf16)(1.0f + e); }
    for (int i = 0; i < iters; ++i) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa, fb, acc, 0, 0, 0);
    v += (unsigned int)acc[0];
  } else if (mode == 4) {
    const unsigned long b = (unsigned long)src;
    const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(
        reinterpret_cast<void*>(((unsigned long)__builtin_amdgcn_readfirstlane((unsigned int)(b >> 32)) << 32) | __builtin_amdgcn_readfirstlane((unsigned int)b)), 0,
        (int)__builtin_amdgcn_readfirstlane(src_bytes), 0x00020000);
    const unsigned int lds0 = (unsigned int)(unsigned long)(__attribute__((address_space(3))) void*)lds;
    for (int i = 0; i < iters; ++i) {
      const unsigned int k0 = ((blockIdx.x * 131u + i * 17u + wave * 8u) % ((src_bytes >> 10) - 8)) << 10;
#pragma unroll
      for (int j = 0; j < 8; ++j) dma16(r, lds0 + (wave * 8 + j) * 1024, lane * 16, (int)(k0 + j * 1024));
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      v += lds[(wave * 8) * 256 + lane];
    }
  } else if (mode == 5) {
    f32x2 a = {1.0f + lane, 2.0f}, bb = {1.0001f, 0.9999f};
    for (int i = 0; i < iters; ++i) { a = a * bb + bb; bb = bb * bb + a * 1e-9f; }
    v += (unsigned int)a[0];
  } else {
    for (int i = 0; i < iters; ++i) {
      __builtin_amdgcn_s_sleep(8);
      v = v * 1664525u + 1013904223u;
    }
  }
  if (v == 0x12345678u) sink[0] = v + lds[0];
}
extern "C" int launch_dummy(int blocks, int lds_bytes, int iters, int mode, void* stream, unsigned int* sink, const unsigned int* src, unsigned int src_bytes) {
  dummy<<<blocks, 256, lds_bytes, (hipStream_t)stream>>>(iters, mode, sink, src, src_bytes);
  return (int)hipGetLastError();
}

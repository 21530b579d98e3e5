// conv_wgrad_wino_f32.h -- LDS-DMA helpers shared by the two Winograd-domain weight-gradient kernels (conv_wgrad_wino_f32.hip: DecoderBlock,
// F(2x2, 2x2) per parity; conv_wgrad_wino33_f32.hip: stride-1 3x3 convolutions, F(2x2, 3x3)).
#pragma once
#include "conv_wgrad_f32.h"

namespace {

__device__ __forceinline__ __amdgpu_buffer_rsrc_t ww_make_rsrc(const void* base, long bytes) {
  const unsigned int n = bytes > 0xFFFFFFFEL ? 0xFFFFFFFEu : (unsigned int)(bytes < 0 ? 0 : bytes);
  const unsigned long b = (unsigned long)base;
  const unsigned int lo = __builtin_amdgcn_readfirstlane((unsigned int)b), hi = __builtin_amdgcn_readfirstlane((unsigned int)(b >> 32));
  const unsigned int nn = __builtin_amdgcn_readfirstlane(n);
  return __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<void*>(((unsigned long)hi << 32) | lo), 0, (int)nn, 0x00020000);
}
// (inline asm for the reason given at wb_dma16, conv_wgrad_bf16.hip: the kernel waits for its pieces itself)
__device__ __forceinline__ void ww_dma16(__amdgpu_buffer_rsrc_t r, unsigned int lds_dst, int voff) {
  asm volatile(
      "s_mov_b32 m0, %1\n\t"
      "s_nop 0\n\t"
      "buffer_load_dwordx4 %0, %2, 0 offen lds"
      :
      : "v"(voff), "s"(lds_dst), "s"(r)
      : "memory", "m0");
}
__device__ __forceinline__ void ww_dma_wait() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
__device__ __forceinline__ unsigned int ww_lds_addr(const void* p) {
  return (unsigned int)(unsigned long)(__attribute__((address_space(3))) const void*)p;
}

constexpr int kWwPK = 8;  // tiles per chunk of the four-wave blocks (the plan counts chunks of this size; the eight-wave block takes two at a time)

template <int N>
__device__ __forceinline__ void ww_dma_wait_but() {  // this wave's pieces except the N youngest have landed (in order: profiles/r06/dma_order.txt)
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

}  // namespace

// Probe (round 6): does a DS instruction that reads, as DATA, a VGPR written by the packed-fp32 VALU instruction right in front of it see all
// 64 lanes of the new value?  The failing build of the fused dec5 + final head had  v_pk_fma_f32 v[4:5], ..  /  ds_bpermute_b32 v6, v9, v4
// back to back, and beside a busy neighbour the lanes that received lanes 48-63 got stale sums (profiles/r06/head_race.txt).
//   waves 0-3 of a 512-thread block (one per SIMD): R rounds of   v = pk_fma(x_r, 1, 0)  ->  [GAP s_nop]  ->  t = ds_bpermute(v, lane ^ 16)
//   and t must equal x_r of lane ^ 16;  waves 4-7 (the second wave of each SIMD): idle | VALU loop | packed-VALU loop | MFMA loop.
// Build: hipcc --offload-arch=gfx950 -O2 -Wno-inline-asm -o scripts/probes/probe_pk_ds_hazard.bin scripts/probes/probe_pk_ds_hazard.hip
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e_), __LINE__); exit(2); } } while (0)

template <int GAP, int PARTNER, int PRODUCER>
__global__ __launch_bounds__(512) void probe(unsigned int* bad, int rounds, float seed) {
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  if (wave < 4) {
    unsigned int wrong = 0;
    const int addr = (lane ^ 16) << 2;
    f32x2 one = {1.f, 1.f}, zero = {0.f, 0.f};
    for (int r = 0; r < rounds; ++r) {
      f32x2 x = {seed + (float)(lane * 3 + r), seed + (float)(lane * 5 + r)};
      int t0, t1;
      asm volatile("" : "+v"(x));
      if (PRODUCER == 0) {  // packed fp32 (the failing build's instruction); v[40:41] still hold the previous round's values
        asm volatile(
            "v_pk_fma_f32 v[40:41], %2, %3, %4\n\t"
            ".rept %6\n\ts_nop 0\n\t.endr\n\t"
            "ds_bpermute_b32 %0, %5, v40\n\t"
            "ds_bpermute_b32 %1, %5, v41\n\t"
            "s_waitcnt lgkmcnt(0)"
            : "=&v"(t0), "=&v"(t1)
            : "v"(x), "v"(one), "v"(zero), "v"(addr), "n"(GAP)
            : "memory", "v40", "v41");
      } else {  // two plain v_fma_f32
        asm volatile(
            "v_fma_f32 v40, %2, 1.0, 0\n\t"
            "v_fma_f32 v41, %3, 1.0, 0\n\t"
            ".rept %5\n\ts_nop 0\n\t.endr\n\t"
            "ds_bpermute_b32 %0, %4, v40\n\t"
            "ds_bpermute_b32 %1, %4, v41\n\t"
            "s_waitcnt lgkmcnt(0)"
            : "=&v"(t0), "=&v"(t1)
            : "v"(x[0]), "v"(x[1]), "v"(addr), "n"(GAP)
            : "memory", "v40", "v41");
      }
      const int l2 = lane ^ 16;
      const float w0 = seed + (float)(l2 * 3 + r), w1 = seed + (float)(l2 * 5 + r);
      wrong += (__builtin_bit_cast(float, t0) != w0 || __builtin_bit_cast(float, t1) != w1) ? 1u : 0u;
    }
    if (wrong) atomicAdd(&bad[lane >> 4], wrong);  // by the RECEIVING lane's quarter (it reads quarter ^ 1)
  } else if (PARTNER == 1) {
    float a = seed + lane, b = 1.0001f;
    for (int r = 0; r < rounds * 8; ++r) { a = a * b + 0.5f; b = b * 0.9999f + a * 1e-9f; }
    if (a == 12345.f) bad[7] = 1;
  } else if (PARTNER == 2) {
    f32x2 a = {seed + lane, seed}, b = {1.0001f, 0.9999f};
    for (int r = 0; r < rounds * 8; ++r) { a = a * b + b; b = b * b + a * 1e-9f; }
    if (a[0] == 12345.f) bad[7] = 1;
  } else if (PARTNER == 3) {
    f32x16 acc = {};
    bf16x8 fa, fb;
    for (int e = 0; e < 8; ++e) { fa[e] = (__bf16)(lane * 0.001f + e); fb[e] = (__bf16)(seed + e); }
    for (int r = 0; r < rounds * 2; ++r) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa, fb, acc, 0, 0, 0);
    if (acc[0] == 12345.f) bad[7] = 1;
  }
}

template <int GAP, int PARTNER, int PRODUCER>
void run(const char* name, int launches, int blocks, int rounds) {
  unsigned int* d;
  CK(hipMalloc(&d, 32));
  CK(hipMemset(d, 0, 32));
  for (int i = 0; i < launches; ++i) probe<GAP, PARTNER, PRODUCER><<<blocks, 512>>>(d, rounds, 1.0f + i);
  CK(hipDeviceSynchronize());
  unsigned int h[8];
  CK(hipMemcpy(h, d, 32, hipMemcpyDeviceToHost));
  printf("%-78s wrong by receiving quarter %8u %8u %8u %8u  of %ld per quarter\n", name, h[0], h[1], h[2], h[3], (long)launches * blocks * 4 * 16 * rounds);
  fflush(stdout);
  CK(hipFree(d));
}

int main() {
  const int L = 10, B = 1024, R = 2000;
  run<0, 0, 0>("v_pk_fma_f32 -> ds_bpermute back to back, partner wave idle", L, B, R);
  run<0, 1, 0>("v_pk_fma_f32 -> ds_bpermute back to back, partner: v_fma loop", L, B, R);
  run<0, 2, 0>("v_pk_fma_f32 -> ds_bpermute back to back, partner: v_pk_fma loop", L, B, R);
  run<0, 3, 0>("v_pk_fma_f32 -> ds_bpermute back to back, partner: MFMA loop", L, B, R);
  run<1, 2, 0>("v_pk_fma_f32 -> s_nop 0 -> ds_bpermute, partner: v_pk_fma loop", L, B, R);
  run<2, 2, 0>("v_pk_fma_f32 -> 2 x s_nop 0 -> ds_bpermute, partner: v_pk_fma loop", L, B, R);
  run<0, 2, 1>("v_fma_f32 x 2 -> ds_bpermute back to back, partner: v_pk_fma loop", L, B, R);
  run<0, 3, 1>("v_fma_f32 x 2 -> ds_bpermute back to back, partner: MFMA loop", L, B, R);
  return 0;
}

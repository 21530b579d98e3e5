// Probe: do LDS-DMA pieces (buffer_load_dwordx4 ... lds) retire IN ORDER, i.e. may `s_waitcnt vmcnt(N)` be read as "all but my N
// youngest pieces have landed in LDS"?  Round 5 bisected two non-reproducible kernels to counted waits over pieces issued OUT OF
// RANGE (profiles/r05/wgrad_ring.txt); this probe asks the hardware directly, one variant of the YOUNGER pieces at a time:
//
//   wave 0 of every block:  NA cold pieces A (HBM misses: 1 KiB each from a multi-GB buffer, never the same line twice),
//                           then NB younger pieces B of the variant under test, then  s_waitcnt vmcnt(NB)  and, at once,
//                           ds_read of the LDS the A pieces fill.  A sentinel still there = a B piece retired AHEAD of an older A.
//   B variants:  0 none (vmcnt(0): the control that must read 0 stale)      1 every lane out of range (offset sentinel)
//                2 lane 0 in range (an L2-hot line), lanes 1-63 out of range   3 every lane in range, L2-hot (a "zero line")
//                4 every lane in range, cold                                  5 lanes 0-31 in range hot, 32-63 out of range
//                6 every lane out of range through a descriptor with num_records = 0
//   noise:       the block's other waves hammer LDS (ds_read_b128 / ds_write_b128 loops) while wave 0 waits -- the round-5 failures
//                needed an LDS-using neighbour on the CU
//
// Build:  hipcc --offload-arch=gfx950 -O2 -o scripts/probes/probe_dma_order.bin scripts/probes/probe_dma_order.hip
// Output: one line per (variant, NA, NB, noise): stale pieces / pieces checked, blocks with any stale piece, mean wait cycles.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x)                                                                 \
  do {                                                                        \
    hipError_t e_ = (x);                                                      \
    if (e_ != hipSuccess) {                                                   \
      printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); \
      exit(2);                                                                \
    }                                                                         \
  } while (0)

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
constexpr int kOOB = (int)0xFFFF0000u;

__device__ __forceinline__ void dma16(__amdgpu_buffer_rsrc_t r, unsigned int lds_dst, int voff, int soff) {
  asm volatile(
      "s_mov_b32 m0, %1\n\t"
      "s_nop 0\n\t"
      "buffer_load_dwordx4 %0, %2, %3 offen lds"
      :
      : "v"(voff), "s"(lds_dst), "s"(r), "s"(soff)
      : "memory", "m0");
}
template <int N>
__device__ __forceinline__ void wait_n() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}
__device__ __forceinline__ __amdgpu_buffer_rsrc_t mk(const void* base, unsigned int bytes) {
  const unsigned long b = (unsigned long)base;
  const unsigned int lo = __builtin_amdgcn_readfirstlane((unsigned int)b), hi = __builtin_amdgcn_readfirstlane((unsigned int)(b >> 32));
  return __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<void*>(((unsigned long)hi << 32) | lo), 0, (int)__builtin_amdgcn_readfirstlane(bytes),
                                           0x00020000);
}

// big: `big_kib` KiB of dwords, dword i holds i.  slot: which KiB the (block, piece) reads this round.
template <int VAR, int NA, int NB, int NOISE>
__global__ __launch_bounds__(256) void probe(const unsigned int* big, unsigned int big_kib, const unsigned int* hot, unsigned int round,
                                             unsigned int* stale, unsigned long long* cycles) {
  __shared__ __attribute__((aligned(16))) unsigned int lds[(NA + NB + 1) * 256 + 3 * 1024];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  for (int i = tid; i < (NA + NB + 1) * 256 + 3 * 1024; i += 256) lds[i] = 0xDEADBEEFu;
  __syncthreads();
  const unsigned int lds0 = (unsigned int)(unsigned long)(__attribute__((address_space(3))) void*)lds;
  if (wave == 0) {
    // (one descriptor per GiB window: offsets are 32-bit)
    const unsigned int nblk = gridDim.x;
    unsigned int kib[NA + (VAR == 4 ? NB : 0) + 1];
#pragma unroll
    for (int a = 0; a < NA + (VAR == 4 ? NB : 0); ++a) {
      // a different KiB per (round, block, piece); a multiplicative hash spreads them over the whole buffer
      const unsigned long long id = ((unsigned long long)round * nblk + blockIdx.x) * (NA + NB) + a;
      kib[a] = (unsigned int)((id * 2654435761ull) % big_kib);
    }
    const __amdgpu_buffer_rsrc_t rhot = mk(hot, 1024);
    const __amdgpu_buffer_rsrc_t rnone = mk(hot, 0);
    __amdgpu_buffer_rsrc_t ra[NA];
#pragma unroll
    for (int a = 0; a < NA; ++a) ra[a] = mk(big + (size_t)kib[a] * 256, 1024);
    const unsigned long long t0 = __builtin_readcyclecounter();
#pragma unroll
    for (int a = 0; a < NA; ++a) dma16(ra[a], lds0 + a * 1024, lane * 16, 0);
#pragma unroll
    for (int b = 0; b < NB; ++b) {
      const unsigned int dst = lds0 + (NA + b) * 1024;
      if constexpr (VAR == 1) dma16(ra[0], dst, kOOB, 0);
      if constexpr (VAR == 2) dma16(rhot, dst, lane == 0 ? 0 : kOOB, 0);
      if constexpr (VAR == 3) dma16(rhot, dst, lane * 16, 0);
      if constexpr (VAR == 4) dma16(mk(big + (size_t)kib[NA + b] * 256, 1024), dst, lane * 16, 0);
      if constexpr (VAR == 5) dma16(rhot, dst, lane < 32 ? lane * 16 : kOOB, 0);
      if constexpr (VAR == 6) dma16(rnone, dst, lane * 16, 0);
    }
    wait_n<NB>();
    u32x4 got[NA];
#pragma unroll
    for (int a = 0; a < NA; ++a) got[a] = *reinterpret_cast<const u32x4*>(&lds[a * 256 + lane * 4]);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    const unsigned long long t1 = __builtin_readcyclecounter();
    wait_n<0>();
    unsigned int bad = 0;
#pragma unroll
    for (int a = 0; a < NA; ++a) {
      const unsigned int want = kib[a] * 256u + lane * 4;
      const bool ok = got[a][0] == want && got[a][1] == want + 1 && got[a][2] == want + 2 && got[a][3] == want + 3;
      bad += __builtin_popcountll(__builtin_amdgcn_ballot_w64(!ok)) ? 1 : 0;  // pieces with any stale lane
    }
    if (lane == 0) {
      stale[blockIdx.x] = bad;
      cycles[blockIdx.x] = t1 - t0;
    }
  } else if (NOISE) {
    // LDS traffic beside the waiting wave: 16-byte reads and writes over the block's last 12 KiB
    unsigned int* mine = lds + (NA + NB + 1) * 256 + (wave - 1) * 1024;
    u32x4 v = {1u, 2u, 3u, 4u};
    for (int it = 0; it < 400; ++it) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        *reinterpret_cast<u32x4*>(&mine[((lane + 17 * r) & 63) * 4 + (r & 3) * 256]) = v;
        v += *reinterpret_cast<const u32x4*>(&mine[((lane * 5 + r) & 63) * 4 + ((r + 1) & 3) * 256]);
      }
    }
    if (v[0] == 0x12345u) stale[blockIdx.x] = v[1];  // (keeps the loop alive)
  }
}

// A co-resident LDS-DMA user (the round-5 failures needed one on the CU): 256 threads, 32 KB of LDS, every wave streams 1 KiB pieces from
// an L2-resident buffer into LDS, drains, reads them back with ds_read_b128.
__global__ __launch_bounds__(256) void neighbour(const unsigned int* src, unsigned int src_kib, unsigned int* sink, int iters) {
  __shared__ __attribute__((aligned(16))) unsigned int lds[8192];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const unsigned int lds0 = (unsigned int)(unsigned long)(__attribute__((address_space(3))) void*)lds;
  const __amdgpu_buffer_rsrc_t r = mk(src, src_kib << 10);
  u32x4 acc = {0u, 0u, 0u, 0u};
  for (int it = 0; it < iters; ++it) {
    const unsigned int k0 = ((blockIdx.x * 131u + it * 17u + wave * 4u) % (src_kib - 8)) << 10;
#pragma unroll
    for (int j = 0; j < 8; ++j) dma16(r, lds0 + (wave * 8 + j) * 1024, lane * 16, (int)(k0 + j * 1024));
    wait_n<0>();
    __syncthreads();
#pragma unroll
    for (int j = 0; j < 8; ++j) acc += *reinterpret_cast<const u32x4*>(&lds[(((wave + 1) & 3) * 8 + j) * 256 + lane * 4]);
    __syncthreads();
  }
  if (acc[0] == 0x7654321u) sink[0] = acc[1];
}

struct Res {
  long stale, checked, blocks_bad;
  double cyc;
};

static hipStream_t g_side = nullptr;
static unsigned int* g_sink = nullptr;

template <int VAR, int NA, int NB, int NOISE>
Res run(const unsigned int* big, unsigned int big_kib, const unsigned int* hot, int rounds, int blocks, unsigned int& round, bool nb = false) {
  unsigned int* d_stale;
  unsigned long long* d_cyc;
  CK(hipMalloc(&d_stale, blocks * 4));
  CK(hipMalloc(&d_cyc, blocks * 8));
  std::vector<unsigned int> hs(blocks);
  std::vector<unsigned long long> hc(blocks);
  Res r{0, 0, 0, 0.0};
  for (int i = 0; i < rounds; ++i) {
    CK(hipMemset(d_stale, 0, blocks * 4));
    if (nb) neighbour<<<1024, 256, 0, g_side>>>(big, 65536, g_sink, 300);  // (the first 64 MiB of `big`: L2 / Infinity-Cache resident)
    probe<VAR, NA, NB, NOISE><<<blocks, 256>>>(big, big_kib, hot, round++, d_stale, d_cyc);
    CK(hipDeviceSynchronize());
    CK(hipMemcpy(hs.data(), d_stale, blocks * 4, hipMemcpyDeviceToHost));
    CK(hipMemcpy(hc.data(), d_cyc, blocks * 8, hipMemcpyDeviceToHost));
    for (int b = 0; b < blocks; ++b) {
      r.stale += hs[b];
      r.blocks_bad += hs[b] ? 1 : 0;
      r.cyc += (double)hc[b];
    }
    r.checked += (long)blocks * NA;
  }
  r.cyc /= (double)rounds * blocks;
  CK(hipFree(d_stale));
  CK(hipFree(d_cyc));
  return r;
}

__global__ void fill(unsigned int* p, size_t n) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = (unsigned int)i;
}

static const char* kName[] = {"none (vmcnt 0)", "all lanes OOB", "lane 0 hot, rest OOB", "all lanes hot line", "all lanes cold", "half hot, half OOB",
                              "all lanes, num_records 0"};

template <int VAR, int NA, int NB>
void both(const unsigned int* big, unsigned int big_kib, const unsigned int* hot, int rounds, int blocks, unsigned int& round) {
  const Res q = run<VAR, NA, NB, 0>(big, big_kib, hot, rounds, blocks, round);
  const Res n = run<VAR, NA, NB, 1>(big, big_kib, hot, rounds, blocks, round);
  const Res k = run<VAR, NA, NB, 0>(big, big_kib, hot, rounds, blocks, round, true);
  printf("B = %-26s NA %d NB %d | quiet: stale %6ld of %8ld (blocks %6ld) wait %7.0f cyc | LDS noise: stale %6ld of %8ld (blocks %6ld) wait %7.0f cyc"
         " | LDS-DMA neighbour kernel: stale %6ld of %8ld (blocks %6ld) wait %7.0f cyc\n",
         kName[VAR], NA, NB, q.stale, q.checked, q.blocks_bad, q.cyc, n.stale, n.checked, n.blocks_bad, n.cyc, k.stale, k.checked, k.blocks_bad, k.cyc);
  fflush(stdout);
}

int main(int argc, char** argv) {
  const int rounds = argc > 1 ? atoi(argv[1]) : 20;
  const int blocks = argc > 2 ? atoi(argv[2]) : 2048;
  const size_t big_bytes = (size_t)4 << 30;  // 4 GiB of dwords: dword i = i (fits 32 bits: 2^30 dwords)
  unsigned int *big, *hot;
  CK(hipMalloc(&big, big_bytes));
  CK(hipMalloc(&hot, 1024));
  fill<<<4096, 256>>>(big, big_bytes / 4);
  CK(hipMemset(hot, 0, 1024));
  CK(hipDeviceSynchronize());
  CK(hipStreamCreateWithFlags(&g_side, hipStreamNonBlocking));
  CK(hipMalloc(&g_sink, 64));
  const unsigned int big_kib = (unsigned int)(big_bytes >> 10);
  unsigned int round = 1;
  printf("probe_dma_order: %d rounds x %d blocks; a piece is stale when the sentinel is still in its LDS right behind s_waitcnt vmcnt(NB)\n", rounds, blocks);
  both<0, 2, 0>(big, big_kib, hot, rounds, blocks, round);
  both<1, 2, 1>(big, big_kib, hot, rounds, blocks, round);
  both<1, 4, 2>(big, big_kib, hot, rounds, blocks, round);
  both<6, 2, 1>(big, big_kib, hot, rounds, blocks, round);
  both<2, 2, 1>(big, big_kib, hot, rounds, blocks, round);
  both<2, 4, 2>(big, big_kib, hot, rounds, blocks, round);
  both<5, 2, 1>(big, big_kib, hot, rounds, blocks, round);
  both<3, 2, 1>(big, big_kib, hot, rounds, blocks, round);
  both<3, 4, 2>(big, big_kib, hot, rounds, blocks, round);
  both<4, 2, 1>(big, big_kib, hot, rounds, blocks, round);
  both<4, 4, 2>(big, big_kib, hot, rounds, blocks, round);
  both<0, 4, 0>(big, big_kib, hot, rounds, blocks, round);
  return 0;
}

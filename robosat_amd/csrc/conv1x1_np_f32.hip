// conv1x1_np_f32.hip -- the fp32 1x1 / stride-1 convolutions of the predict pass whose EPILOGUE is as long as a good part of their main
// loop (Bottleneck.conv3 with the residual add: 128 -> 512, 256 -> 1024, 512 -> 2048; reference torchvision Bottleneck via
// robosat/unet.py:94,122-130, eval-mode BatchNorm folded into scale / shift, `out += identity; relu`), with the epilogue of one
// sub-tile issued BY THE SAME WAVES between the matrix instructions of the next ("N-pipelined").
//
// Why (profiles/r04/k_slope_1x1.txt, profiles/r04/ew_1x1.txt): in the generic kernel the main loop of such a launch runs at 94-96 % of
// the fp32 matrix cores' peak and its epilogue moves output + residual at 5.3-5.7 TB/s, but the launch takes their SUM (256 -> 1024 at
// 32^2, bs 16: 86 us = 55 + 24 + fixed): the four blocks resident on a CU are identical, start together and reach their epilogues
// together, one round of blocks or two.  conv1x1_ew_f32 (epilogue on four extra waves) buys the overlap with half of the waves that
// hide the main loop's own latencies and loses on K >= 128.  Here nothing is given up: 256-thread blocks, four per CU, every wave
// issues MFMAs -- a block walks 128-pixel x 64-cout sub-tiles (cout fastest), and while it multiplies sub-tile i + 1 it drains the
// accumulators of sub-tile i from a second register set: per chunk step one 16-byte quad per lane -- residual / scale / shift requested
// at the top of step c, combined and stored at the top of step c + 1, straight from registers (a lane's four accumulator registers
// are four consecutive couts of one pixel: one 16-byte store; lanes l and l + 32 and the four quads complete a pixel's 128 bytes).
// No LDS staging, no extra barrier.  Stores and loads go out right behind the step's barrier, so that the `s_waitcnt vmcnt(0)` in front
// of the NEXT barrier (the LDS-DMA wait; on gfx9 stores count in vmcnt too) finds them long done.
// K order, MFMA operand order and the epilogue's arithmetic are the generic kernel's: bit-identical results (44 cases, scripts/np_check.py).
//
// STATUS (round 6): MEASURED AND NOT ADOPTED -- `make EXP=1` only (profiles/r06/np_1x1.txt).  Correct on its first run, and slower:
// x 0.73-0.82 on the three conv3 shapes it was built for, x 0.78-1.00 on the others.  Knock-outs on 256 -> 1024 at 32^2 (generic 80 us):
// as built 101 us, without its stores 80, without its residual loads 80, without either 74 -- against 55 us of MFMA time at the peak.
// Two things: the 128 x 64 sub-tile's main loop runs at ~75 % of the matrix cores where the generic 128 x 128 tile reaches ~95 % (the
// finding of conv1x1_ew_f32 again: half the MFMAs per barrier and per fetched byte), and the epilogue traffic is NOT hidden -- stores
// and loads count in the one vmcnt the LDS-DMA wait in front of every barrier drains, so each step waits for the step's store.
#define RS_CONV_INSTANTIATE  // (the LDS-DMA helpers and ConvArgsT of the header; no kernel of it is instantiated here)
#include "conv_igemm_dma_kernel.h"

namespace {

constexpr int NP_BM = 128, NP_BN = 64, NP_ROWB = 64, NP_KC = 16;
constexpr int NP_BUF = (NP_BM + NP_BN) * NP_ROWB;  // bytes per pipeline buffer: pixel rows, then filter rows
constexpr int NP_NQ = 8;                           // quads (16 bytes = 4 couts of one pixel) per lane and sub-tile: 2 pixel sub-tiles x 4

__device__ __forceinline__ void np_barrier() { asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory"); }

__global__ __launch_bounds__(256, 4) void conv1x1_np_f32_kernel(const ConvArgsT<float> p) {
  __shared__ __attribute__((aligned(16))) unsigned char smem[2 * NP_BUF];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int K = p.C1, nk = K / NP_KC;
  const int ntiles = p.Cout / NP_BN;
  const int items = ((p.M + NP_BM - 1) / NP_BM) * ntiles;
  const int first = rs_xcd_remap(blockIdx.x, gridDim.x);
  const int nitems = first < items ? (items - first + (int)gridDim.x - 1) / (int)gridDim.x : 0;
  const int total = nitems * nk;

  const int wm = wave >> 1, wn = wave & 1;
  const int ra = lane >> 2, pp = lane & 3;
  const int gp = pp ^ ((ra >> 2) & 3);  // the 16-byte piece this lane fetches (swizzle on the SOURCE address: the LDS image is lane-linear)
  const unsigned int lds0 = __builtin_amdgcn_readfirstlane(rb_lds_addr(smem));
  const __amdgpu_buffer_rsrc_t rsrcw = rb_make_rsrc(p.wgt, (long)p.Cout * K * 4);
  __amdgpu_buffer_rsrc_t rsrca = rb_make_rsrc(p.src1, 0);
  // fetch cursor (wave-uniform): item f_seq, chunk f_kc of it, global chunk f_g (buffer f_g & 1)
  int f_seq = 0, f_kc = 0, f_g = 0;
  int voff[3];  // this lane's byte offsets: two pixel-row instructions (relative to the item's first row), one filter-row instruction
  auto fetch_item = [&]() __attribute__((always_inline)) {
    const int it = first + f_seq * (int)gridDim.x;
    const int mt = __builtin_amdgcn_readfirstlane(it / ntiles);
    const int nt = it - mt * ntiles;
    const int m0 = mt * NP_BM, n0 = nt * NP_BN;
    rsrca = rb_make_rsrc(p.src1 + (long)m0 * K, (long)(p.M - m0) * K * 4);  // (rows past M are past the descriptor: zeros)
#pragma unroll
    for (int j = 0; j < 2; ++j) voff[j] = (16 * (wave + 4 * j) + ra) * (K * 4) + gp * 16;
    voff[2] = (n0 + 16 * wave + ra) * (K * 4) + gp * 16;
  };
  auto issue = [&](int j) __attribute__((always_inline)) {  // j compile-time: instruction wave + 4 j of the chunk's 12
    const unsigned int dst = lds0 + (f_g & 1) * NP_BUF + (wave + 4 * j) * 1024;
    if (j < 2) rb_dma16s(rsrca, dst, voff[j], f_kc * NP_ROWB);
    else rb_dma16s(rsrcw, dst, voff[j], f_kc * NP_ROWB);
  };
  auto advance = [&]() __attribute__((always_inline)) {
    ++f_g;
    if (++f_kc == nk) {
      f_kc = 0;
      ++f_seq;
    }
  };

  // fragment addressing (as the generic kernel's 64-byte rows): row lane & 31 of a 32-row sub-tile, piece 2 s + (lane >> 5)
  const int frow = lane & 31;
  const int fl = (frow >> 2) & 3;
  int foff[2];
#pragma unroll
  for (int s = 0; s < 2; ++s) foff[s] = ((2 * s + (lane >> 5)) ^ fl) * 16;
  const int abase = (wm * 64 + frow) * NP_ROWB;
  const int bbase = (NP_BM + wn * 32 + frow) * NP_ROWB;

  f32x16 acc[2], accp[2];
#pragma unroll
  for (int tm = 0; tm < 2; ++tm)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[tm][r] = 0.f, accp[tm][r] = 0.f;

  // ---- the drain side: quad j of the PREVIOUS sub-tile = registers 4 (j & 3) .. + 3 of accp[j >> 2]:
  //      pixel m_d + 64 wm + 32 (j >> 2) + (lane & 31), couts n_d + 32 wn + 8 (j & 3) + 4 (lane >> 5) .. + 3
  int d_row = 0;  // element offset of (pixel m_d + 64 wm + (lane & 31), cout n_d + 32 wn + 4 (lane >> 5)) in the output (M * Cout < 2^31: the launcher's rule)
  int d_m = 0, d_col = 0;
  auto drain_item = [&](int seq) __attribute__((always_inline)) {
    const int it = first + seq * (int)gridDim.x;
    const int mt = it / ntiles, nt = it - mt * ntiles;
    d_m = mt * NP_BM + 64 * wm + (lane & 31);
    d_col = nt * NP_BN + 32 * wn + 4 * (lane >> 5);
    d_row = d_m * p.Cout + d_col;
  };
  f32x4 q_res;  // the requested quad's residual piece (an HBM round trip: asked for one chunk step ahead; scale / shift are cache hits)
  auto request = [&](int j) __attribute__((always_inline)) {  // j compile-time
    if (p.res && d_m + 32 * (j >> 2) < p.M) q_res = *reinterpret_cast<const f32x4*>(p.res + d_row + (32 * (j >> 2)) * p.Cout + 8 * (j & 3));
  };
  auto finish = [&](int j) __attribute__((always_inline)) {  // j compile-time: the quad `request(j)` asked for
    if (d_m + 32 * (j >> 2) >= p.M) return;
    const int c = d_col + 8 * (j & 3);
    const f32x4 q_sc = p.scale ? *reinterpret_cast<const f32x4*>(p.scale + c) : f32x4{1.f, 1.f, 1.f, 1.f};
    const f32x4 q_sh = p.shift ? *reinterpret_cast<const f32x4*>(p.shift + c) : f32x4{0.f, 0.f, 0.f, 0.f};
    f32x4 v;
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] = accp[j >> 2][4 * (j & 3) + e] * q_sc[e] + q_sh[e];
    if (p.res) v += q_res;
    if (p.relu) {
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], 0.f);
    }
    *reinterpret_cast<f32x4*>(p.out + d_row + (32 * (j >> 2)) * p.Cout + 8 * (j & 3)) = v;
  };

  if (total > 0) {
    fetch_item();
#pragma unroll
    for (int j = 0; j < 3; ++j) issue(j);
    advance();
  }
  int g = 0;
  for (int seq = 0; seq < nitems; ++seq) {
    // chunk step kc of sub-tile `seq`: wait + barrier, the drain slice of sub-tile seq - 1 (steps 0 .. 8: request quad kc, finish quad
    // kc - 1), 16 MFMAs with the next chunk's three DMA instructions of this wave between them.  Steps 0 .. 8 are peeled so that the
    // quad index is a compile-time constant (register arrays must not be indexed dynamically).
    auto step = [&](auto jc) __attribute__((always_inline)) {
      constexpr int J = decltype(jc)::value;  // 0 .. 8: drain slice of this step; 9: none
      np_barrier();  // chunk g is in buffer g & 1; buffer (g + 1) & 1 is free again; every load / store of the last step is done
      const bool more = g + 1 < total;
      if (more && f_kc == 0) fetch_item();
      if (seq > 0) {
        if constexpr (J >= 1 && J <= NP_NQ) finish(J - 1);
        if constexpr (J < NP_NQ) request(J);
      }
      const unsigned char* L = smem + (g & 1) * NP_BUF;
      u32x4 fa[2][2], fb[2];
#pragma unroll
      for (int tm = 0; tm < 2; ++tm) fa[0][tm] = *reinterpret_cast<const u32x4*>(L + abase + 32 * tm * NP_ROWB + foff[0]);
      fb[0] = *reinterpret_cast<const u32x4*>(L + bbase + foff[0]);
#pragma unroll
      for (int tm = 0; tm < 2; ++tm) fa[1][tm] = *reinterpret_cast<const u32x4*>(L + abase + 32 * tm * NP_ROWB + foff[1]);
      fb[1] = *reinterpret_cast<const u32x4*>(L + bbase + foff[1]);
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        const f32x4 wv = __builtin_bit_cast(f32x4, fb[s]);
#pragma unroll
        for (int tm = 0; tm < 2; ++tm) {
          const f32x4 xv = __builtin_bit_cast(f32x4, fa[s][tm]);
#pragma unroll
          for (int t = 0; t < 4; ++t) {
            const int q = (s * 2 + tm) * 4 + t;
            if (q % 5 == 0 && q / 5 < 3) {
              if (more) issue(q / 5);
            }
            acc[tm] = __builtin_amdgcn_mfma_f32_32x32x2f32(wv[t], xv[t], acc[tm], 0, 0, 0);
          }
        }
      }
      if (more) advance();
      ++g;
    };
    int kc = 0;
    rb_for_each([&](auto jc) __attribute__((always_inline)) {
      if (kc < nk) {
        step(jc);
        ++kc;
      }
    }, std::make_integer_sequence<int, NP_NQ + 1>());
    for (; kc < nk; ++kc) step(std::integral_constant<int, NP_NQ + 1>());
    // (K < 144: fewer than nine chunk steps -- the quads the steps did not reach are finished here; nk >= 2 is the launcher's rule)
    if (seq > 0 && nk <= NP_NQ) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      rb_for_each([&](auto jc) __attribute__((always_inline)) {
        constexpr int J = decltype(jc)::value;
        if (J >= nk - 1) {
          if (J >= nk) request(J);  // (quad nk - 1 was requested by the last step)
          finish(J);
        }
      }, std::make_integer_sequence<int, NP_NQ>());
    }
    // this sub-tile's accumulators become the ones being drained
#pragma unroll
    for (int tm = 0; tm < 2; ++tm) {
      accp[tm] = acc[tm];
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[tm][r] = 0.f;
    }
    drain_item(seq);
  }
  if (nitems > 0) {  // the block's last sub-tile: nothing left to hide it under
    rb_for_each([&](auto jc) __attribute__((always_inline)) {
      constexpr int J = decltype(jc)::value;
      request(J);
      finish(J);
    }, std::make_integer_sequence<int, NP_NQ>());
  }
  rb_dma_wait();
}

}  // namespace

// 1: this launch can take the N-pipelined kernel (geometry only -- never the batch size)
int rs_conv1x1_np_f32_ok(const rs_conv_desc* d) {
  if (!d || d->stem || d->kh != 1 || d->kw != 1 || d->stride != 1 || d->pad != 0 || d->ups != 0 || d->C2 != 0) return 0;
  if (d->Ho != d->Hs || d->Wo != d->Ws) return 0;
  if (d->C1 < 2 * NP_KC || (d->C1 % NP_KC) || d->Cout <= 0 || (d->Cout % NP_BN)) return 0;
  if ((long)d->Cout * d->C1 * 4 >= (1L << 31) || (long)(NP_BM + 16) * d->C1 * 4 >= (1L << 31)) return 0;
  if ((long)d->N * d->Ho * d->Wo * d->Cout >= (1L << 31)) return 0;  // 32-bit element offsets in the drain
  return 1;
}

int rs_conv1x1_np_f32_launch(const ConvArgsT<float>& a, hipStream_t s) {
  int dev = 0, cus = 0;
  if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0)
    cus = 256;
  const long items = (long)rs_cdiv(a.M, NP_BM) * (a.Cout / NP_BN);
  if (items <= 0 || items >= (1L << 31)) return RS_EINVAL;
  const int grid = (int)(items < 4L * cus ? items : 4L * cus);
  conv1x1_np_f32_kernel<<<grid, 256, 0, s>>>(a);
  return RS_LAUNCH_RESULT();
}

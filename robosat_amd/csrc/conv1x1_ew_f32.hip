// conv1x1_ew_f32.hip -- the fp32 1x1 / stride-1 convolutions of the predict pass (Bottleneck.conv1 / conv3 and the stride-1
// downsample of every ResNet-50 block, reference torchvision Bottleneck via robosat/unet.py:94,122-130, with the eval-mode
// BatchNorm folded into scale / shift, the residual add and the ReLU) with the EPILOGUE ON ITS OWN WAVES.
//
// Why (profiles/r04/k_slope_1x1.txt): in the generic kernel (conv_igemm_dma_kernel.h) the main loop of such a launch runs at
// 94-96 % of the fp32 matrix cores' peak and its epilogue moves the output + residual at 5.3-5.7 TB/s -- but the launch
// takes 0.85-0.9 x their SUM, not their maximum: the four blocks resident on a CU walk their tiles in step and reach their
// epilogues together, so the matrix cores idle while HBM streams and the other way round.  Here a block is
//   waves 0-3  main loop only: LDS-DMA of the pixel / filter chunks (64-byte rows = 16 channels, double buffered, one chunk
//              ahead and ACROSS tiles), v_mfma_f32_32x32x2_f32 on a 128-pixel x 64-cout tile (2 x 2 waves, 64 x 32 each);
//              at the end of a tile the accumulators go to an LDS staging tile and the next tile starts at once;
//   waves 4-7  epilogue only, one tile behind: staged accumulators x scale + shift (+ residual) (ReLU) -> 16-byte stores of
//              whole 256-byte row segments, a slice per chunk step of the tile the other four are computing; the residual
//              pieces of a tile are requested one step before its accumulators are staged.
// The block is persistent (two per CU: 59 KB of LDS, <= 128 registers) and walks (pixel tile, cout tile) items, cout tile
// fastest.  Both halves pass the SAME sequence of s_barrier instructions (one per chunk step + one at the end); nothing
// else synchronises them: a tile's staging is written after the last chunk step's barrier, by which time the epilogue
// waves have finished the previous tile (they work in chunk steps 0 .. nk-2 only).
// K order, MFMA operand order and the epilogue's arithmetic are those of the generic kernel's 64-byte-row variants.
//
// STATUS: measurement candidate, reached only with RS_CONV1X1_EW=1 in the environment (conv_igemm_dma.hip).
#define RS_CONV_INSTANTIATE  // (the LDS-DMA helpers and ConvArgsT of the header; no kernel of it is instantiated here)
#include "conv_igemm_dma_kernel.h"

namespace {

constexpr int EW_BM = 128, EW_BN = 64, EW_ROWB = 64, EW_KC = 16;
constexpr int EW_BUF = (EW_BM + EW_BN) * EW_ROWB;  // bytes per pipeline buffer: pixel rows, then filter rows
constexpr int EW_LDO = EW_BN + 4;                  // staging row (floats): 16 lanes' 16-byte accesses hit 64 different banks
constexpr int EW_STAGE = EW_BM * EW_LDO * 4;
constexpr int EW_NG = EW_BM / 16;                  // epilogue groups per tile: 16 rows x 16 pieces of 16 bytes = 256 threads

// (a memory clobber: nothing of either role moves across; LDS traffic of this wave is complete before it arrives)
__device__ __forceinline__ void ew_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

__global__ __launch_bounds__(512, 4) void conv1x1_ew_f32_kernel(const ConvArgsT<float> p) {
  __shared__ __attribute__((aligned(16))) unsigned char smem[2 * EW_BUF + EW_STAGE];
  float* stage = reinterpret_cast<float*>(smem + 2 * EW_BUF);

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int K = p.C1, nk = K / EW_KC;
  const int ntiles = p.Cout / EW_BN;
  const int items = ((p.M + EW_BM - 1) / EW_BM) * ntiles;
  const int first = rs_xcd_remap(blockIdx.x, gridDim.x);
  const int nitems = first < items ? (items - first + (int)gridDim.x - 1) / (int)gridDim.x : 0;
  const int total = nitems * nk;

  if (wave < 4) {
    // ================================================ main-loop waves ================================================
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
      const int m0 = mt * EW_BM, n0 = nt * EW_BN;
      rsrca = rb_make_rsrc(p.src1 + (long)m0 * K, (long)(p.M - m0) * K * 4);  // (rows past M are past the descriptor: zeros)
#pragma unroll
      for (int j = 0; j < 2; ++j) voff[j] = (16 * (wave + 4 * j) + ra) * (K * 4) + gp * 16;
      voff[2] = (n0 + 16 * wave + ra) * (K * 4) + gp * 16;
    };
    auto issue = [&](int j) __attribute__((always_inline)) {  // j compile-time: instruction wave + 4 j of the chunk's 12
      const unsigned int dst = lds0 + (f_g & 1) * EW_BUF + (wave + 4 * j) * 1024;
      if (j < 2) rb_dma16s(rsrca, dst, voff[j], f_kc * EW_ROWB);
      else rb_dma16s(rsrcw, dst, voff[j], f_kc * EW_ROWB);
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
    const int abase = (wm * 64 + frow) * EW_ROWB;
    const int bbase = (EW_BM + wn * 32 + frow) * EW_ROWB;

    f32x16 acc[2];
#pragma unroll
    for (int tm = 0; tm < 2; ++tm)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[tm][r] = 0.f;

    if (total > 0) {
      fetch_item();
#pragma unroll
      for (int j = 0; j < 3; ++j) issue(j);
      advance();
    }
    int g = 0;
    for (int seq = 0; seq < nitems; ++seq) {
      for (int kc = 0; kc < nk; ++kc, ++g) {
        rb_dma_wait();
        ew_barrier();  // chunk g is in buffer g & 1; buffer (g + 1) & 1 is free again
        const bool more = g + 1 < total;
        if (more && f_kc == 0) fetch_item();
        const unsigned char* L = smem + (g & 1) * EW_BUF;
        u32x4 fa[2][2], fb[2];
#pragma unroll
        for (int tm = 0; tm < 2; ++tm) fa[0][tm] = *reinterpret_cast<const u32x4*>(L + abase + 32 * tm * EW_ROWB + foff[0]);
        fb[0] = *reinterpret_cast<const u32x4*>(L + bbase + foff[0]);
#pragma unroll
        for (int tm = 0; tm < 2; ++tm) fa[1][tm] = *reinterpret_cast<const u32x4*>(L + abase + 32 * tm * EW_ROWB + foff[1]);
        fb[1] = *reinterpret_cast<const u32x4*>(L + bbase + foff[1]);
        // 16 MFMAs; the next chunk's three DMA instructions of this wave go out between them
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
      }
      // ---- the tile's accumulators -> staging (D rows = couts 8 g + 4 (lane >> 5) + e, D columns = pixels lane & 31).  The
      //      epilogue waves left the previous tile's staging before the barrier of this tile's last chunk step.
#pragma unroll
      for (int tm = 0; tm < 2; ++tm) {
        const int pr = wm * 64 + 32 * tm + (lane & 31);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          f32x4 v;
          v[0] = acc[tm][4 * q + 0];
          v[1] = acc[tm][4 * q + 1];
          v[2] = acc[tm][4 * q + 2];
          v[3] = acc[tm][4 * q + 3];
          *reinterpret_cast<f32x4*>(&stage[pr * EW_LDO + wn * 32 + 8 * q + 4 * (lane >> 5)]) = v;
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[tm][r] = 0.f;
      }
    }
    ew_barrier();  // publishes the last tile's staging
    rb_dma_wait();
  } else {
    // ================================================ epilogue waves =================================================
    const int e = tid - 256;
    // (round 6) the thread's 16-byte column is ROTATED by its row: a ds_read_b128 is served in lane groups {0-3, 12-15, 20-27}, ... that
    // mix eight lanes of staging row r with eight of row r + 1, and with the row pitch of 68 floats (what keeps the main-loop waves'
    // ds_write_b128 conflict free) straight columns put lane 12 (row r, piece 12) and lane 27 (row r + 1, piece 11) on the same
    // banks: SQ_LDS_BANK_CONFLICT 14.3 % of the kernel's LDS cycles in rounds 4-5.  Rotated, a lane group's (row + piece) mod 16 are
    // its lanes' own low four bits: all different.  Same elements, same arithmetic, a row's 16 lanes still cover its 256 bytes.
    const int r16 = e >> 4, c4 = ((e & 15) - r16) & 15;
    const int gpi = nk > 1 ? (EW_NG + nk - 2) / (nk - 1) : EW_NG;  // groups per chunk step (steps 0 .. nk-2 drain a tile)
    // coordinates of a tile as this thread sees them
    auto coords = [&](int seq, long& obase, int& m_first, int& col) __attribute__((always_inline)) {
      const int it = first + seq * (int)gridDim.x;
      const int mt = it / ntiles, nt = it - mt * ntiles;
      m_first = mt * EW_BM + r16;
      col = nt * EW_BN + 4 * c4;
      obase = (long)m_first * p.Cout + col;
    };
    f32x4 rr[EW_NG];  // the residual pieces of the tile about to be drained
    f32x4 sc = {1.f, 1.f, 1.f, 1.f}, sh = {0.f, 0.f, 0.f, 0.f};
    long obase = 0;
    int m_first = 0, col = 0;
    auto request = [&](int seq) __attribute__((always_inline)) {  // residual pieces + scale / shift of tile `seq`
      coords(seq, obase, m_first, col);
      if (p.scale) sc = *reinterpret_cast<const f32x4*>(p.scale + col);
      if (p.shift) sh = *reinterpret_cast<const f32x4*>(p.shift + col);
      if (p.res) {
#pragma unroll
        for (int j = 0; j < EW_NG; ++j)
          if (m_first + 16 * j < p.M) rr[j] = *reinterpret_cast<const f32x4*>(p.res + obase + (long)(16 * j) * p.Cout);
      }
    };
    auto drain = [&](int lo, int hi) __attribute__((always_inline)) {  // groups [lo, hi) of the requested tile
#pragma unroll
      for (int j = 0; j < EW_NG; ++j) {
        if (j >= lo && j < hi && m_first + 16 * j < p.M) {
          const f32x4 a = *reinterpret_cast<const f32x4*>(&stage[(16 * j + r16) * EW_LDO + 4 * c4]);
          f32x4 v;
#pragma unroll
          for (int q = 0; q < 4; ++q) v[q] = a[q] * sc[q] + sh[q];
          if (p.res) v += rr[j];
          if (p.relu) {
#pragma unroll
            for (int q = 0; q < 4; ++q) v[q] = fmaxf(v[q], 0.f);
          }
          *reinterpret_cast<f32x4*>(p.out + obase + (long)(16 * j) * p.Cout) = v;
        }
      }
    };
    // On gfx9 stores count in vmcnt like loads: a wait for a residual piece inside the drain loop would also wait for every
    // store before it.  So all pending loads of a tile (residual pieces, scale, shift) are waited for ONCE, in front of the
    // tile's first store: `settle` makes the registers inputs and outputs of an empty asm -- hipcc puts its s_waitcnt vmcnt in
    // front of it, and what comes out is not a pending load any more.  The loop is shaped so that every path from a
    // `request` to a `drain` passes a `settle` (first tile and first chunk step peeled), or the waits come back.
    auto settle = [&]() __attribute__((always_inline)) {
      asm volatile("" : "+v"(sc), "+v"(sh));
#pragma unroll
      for (int j = 0; j < EW_NG; ++j) asm volatile("" : "+v"(rr[j]));
    };
    if (nitems > 0) {
      for (int kc = 0; kc < nk - 1; ++kc) ew_barrier();  // tile 0 is being computed: nothing to drain yet
      ew_barrier();
      request(0);
      for (int seq = 1; seq < nitems; ++seq) {
        ew_barrier();
        settle();
        drain(0, gpi);  // tile seq - 1
        for (int kc = 1; kc < nk - 1; ++kc) {
          ew_barrier();
          drain(kc * gpi, (kc + 1) * gpi);
        }
        ew_barrier();
        request(seq);  // the tile being computed: its pieces land while it is finished and staged
      }
      ew_barrier();
      settle();
      drain(0, EW_NG);
    } else {
      ew_barrier();
    }
  }
}

}  // namespace

// 1: this launch can take the epilogue-wave kernel (geometry only -- never the batch size)
int rs_conv1x1_ew_f32_ok(const rs_conv_desc* d) {
  if (!d || d->stem || d->kh != 1 || d->kw != 1 || d->stride != 1 || d->pad != 0 || d->ups != 0 || d->C2 != 0) return 0;
  if (d->Ho != d->Hs || d->Wo != d->Ws) return 0;
  if (d->C1 < 2 * EW_KC || (d->C1 % EW_KC) || d->Cout <= 0 || (d->Cout % EW_BN)) return 0;
  if ((long)d->Cout * d->C1 * 4 >= (1L << 31) || (long)(EW_BM + 16) * d->C1 * 4 >= (1L << 31)) return 0;
  return 1;
}

int rs_conv1x1_ew_f32_launch(const ConvArgsT<float>& a, hipStream_t s) {
  int dev = 0, cus = 0;
  if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0)
    cus = 256;
  const long items = (long)rs_cdiv(a.M, EW_BM) * (a.Cout / EW_BN);
  if (items <= 0 || items >= (1L << 31)) return RS_EINVAL;
  const int grid = (int)(items < 2L * cus ? items : 2L * cus);
  conv1x1_ew_f32_kernel<<<grid, 512, 0, s>>>(a);
  return RS_LAUNCH_RESULT();
}

// bottleneck_tail_f32.hip -- the end of one layer1 Bottleneck and the beginning of the next, fp32 eval mode, ONE launch:
//
//     out = relu(bn3(conv3(x)) + identity)          64 -> 256, 1x1     (torchvision Bottleneck.forward via reference unet.py:127)
//     z   = relu(bn1'(conv1'(out)))                256 ->  64, 1x1     (the NEXT block's first convolution)
//
// As two launches (conv1x1_ew_f32 + conv_igemm_f32<128x64,r64>: 124 + 92 us at bs 16 / 512^2) the 256-channel tensor `out` is
// written (268 MB), read back as the second GEMM's operand (268 MB), and each launch pays its own prologue / epilogue around a
// K = 64 resp. N = 64 product.  Here the second product consumes the first one's ACCUMULATOR REGISTERS as its MFMA operand:
//
//   wave   = 32 pixels x all 256 couts of stage 1 (8 tiles of v_mfma_f32_32x32x2_f32: 128 accumulator registers), then x all 64
//            couts of stage 2 (2 tiles).  D[i][j] of a 32x32x2 MFMA puts pixel j = lane & 31 and couts i = 8 g + 4 (lane >> 5) + e
//            in register 4 g + e -- which is exactly the B-operand layout of the same instruction for the k pair
//            (8 g + e, 8 g + 4 + e): lanes 0-31 carry k, lanes 32-63 carry k + 1 of the same 32 pixels.  So after the stage-1
//            epilogue (scale / shift, + identity, ReLU, 16-byte stores of `out` straight from the registers) register (t, g, e) is
//            fed back as stage 2's pixel operand against the filter piece w1[cout'][32 t + 8 g + 4 (lane >> 5) + e]: no LDS round
//            trip, no second pass over `out`.  (Stage 2 therefore adds its 256 products in the order (t, g, e) -- a fixed order,
//            independent of the batch.)
//   block  = 8 waves, persistent, one per CU; both filters live in LDS for the block's life (w3 [256][64 + 4], w1 [64][256 + 4]:
//            136 KB + the four scale / shift vectors, rows padded by 16 bytes: every 16-lane group of a ds_read_b128 hits 16 distinct bank slots).  The waves are
//            independent after the filters are staged: no barrier, no staging tile -- a wave's pixel operand comes straight from
//            global memory (8 x 16 bytes per lane), `identity` arrives as 16-byte pieces one cout tile ahead of its use.
//   bytes  = x 67 + identity 268 + out 268 + z 67 MB at bs 16 / 512^2 (the unfused pair: + 268); FLOPs 2 x 8.6 G.
#include "common.h"

namespace {

constexpr int TC1 = 64, TCM = 256, TC2 = 64;  // channels: stage-1 input, stage-1 output = stage-2 input, stage-2 output
constexpr int LDW3 = TC1 + 4, LDW1 = TCM + 4;  // padded LDS rows (floats)
constexpr int T1 = TCM / 32, T2 = TC2 / 32;    // MFMA tiles per wave: stage 1 / stage 2

struct TailArgs {
  const float* x;      // [M][64]
  const float* w3;     // [256][64]
  const float* s3;     // [256] folded BatchNorm scale / shift of stage 1
  const float* t3;
  const float* idt;    // [M][256]
  const float* w1;     // [64][256]
  const float* s1;     // [64]
  const float* t1;
  float* out;          // [M][256]
  float* z;            // [M][64]
  int nsub;            // M / 32
  int relu;            // (single-stage form) ReLU on `out`; the chained form always clamps
};

// CHAIN = false: stage 1 alone (a 64 -> 256 1x1 convolution with the same wave-owns-its-pixels layout: layer1's downsample branch,
// 97 against conv1x1_ew_f32's 109 us; with a residual it is the slower one, 140 against 127); `idt` may then be null and `relu` says whether to clamp.
template <bool CHAIN>
__global__ __launch_bounds__(512, 2) void bottleneck_tail_f32(const TailArgs p) {
  __shared__ __attribute__((aligned(16))) float lds[TCM * LDW3 + (CHAIN ? TC2 * LDW1 : 0) + 2 * TCM + 2 * TC2];
  float* const w3l = lds;
  float* const w1l = lds + TCM * LDW3;
  float* const s3l = w1l + (CHAIN ? TC2 * LDW1 : 0);  // scale3 [256], shift3 [256], scale1 [64], shift1 [64]
  float* const t3l = s3l + TCM;
  float* const s1l = t3l + TCM;
  float* const t1l = s1l + TC2;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 31, lh = lane >> 5;

  for (int e = tid; e < TCM * TC1 / 4; e += 512) {
    const int row = e / (TC1 / 4), c4 = e % (TC1 / 4);
    *reinterpret_cast<f32x4*>(&w3l[row * LDW3 + c4 * 4]) = *reinterpret_cast<const f32x4*>(p.w3 + (long)e * 4);
  }
  if constexpr (CHAIN) {
    for (int e = tid; e < TC2 * TCM / 4; e += 512) {
      const int row = e / (TCM / 4), c4 = e % (TCM / 4);
      *reinterpret_cast<f32x4*>(&w1l[row * LDW1 + c4 * 4]) = *reinterpret_cast<const f32x4*>(p.w1 + (long)e * 4);
    }
    if (tid < TC2) s1l[tid] = p.s1[tid], t1l[tid] = p.t1[tid];
  }
  if (tid < TCM) s3l[tid] = p.s3[tid], t3l[tid] = p.t3[tid];
  __syncthreads();

  const float* const w3f = w3l + li * LDW3 + 4 * lh;  // + 32 t rows, + 8 j columns
  const float* const w1f = w1l + li * LDW1 + 4 * lh;  // + 32 tn rows, + 32 t + 8 g columns
  const int nw = (int)gridDim.x * 8;
  int sub = (int)blockIdx.x * 8 + wave;
  if (sub >= p.nsub) return;
  // The wave's operands travel one step ahead of their use, in the registers their predecessors have just left: the pixel operand of
  // the NEXT sub-tile is requested when this one's last stage-1 pair is done, a pair's identity pieces when the previous pair's
  // epilogue has consumed its own -- each request has a pair's 64 stage-2 MFMAs (and more) between it and its first use.
  const bool has_id = CHAIN || p.idt != nullptr;
  const float lo = (CHAIN || p.relu) ? 0.f : -__builtin_huge_valf();  // (ReLU as a clamp from below: -inf = none)
  f32x4 xa[TC1 / 8];  // stage 1's pixel operand: this lane's pixel, channels 8 j + 4 (lane >> 5) ..
  f32x4 rid[2][4];    // identity pieces of a pair of cout tiles
  {
    const long pix = (long)sub * 32 + li;
#pragma unroll
    for (int j = 0; j < TC1 / 8; ++j) xa[j] = *reinterpret_cast<const f32x4*>(p.x + pix * TC1 + 8 * j + 4 * lh);
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
      for (int g = 0; g < 4; ++g)
        rid[u][g] = has_id ? *reinterpret_cast<const f32x4*>(p.idt + pix * TCM + 4 * lh + 32 * u + 8 * g) : f32x4{0.f, 0.f, 0.f, 0.f};
  }
  for (; sub < p.nsub; sub += nw) {
    const long pix = (long)sub * 32 + li;
    const long npix = (long)(sub + nw < p.nsub ? sub + nw : sub) * 32 + li;  // (behind the last sub-tile: the same one again, unused)
    float* const op = p.out + pix * TCM + 4 * lh;
    f32x16 acc2[T2];
#pragma unroll
    for (int tn = 0; tn < T2; ++tn)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc2[tn][r] = 0.f;
    // Two stage-1 cout tiles at a time (two independent accumulator chains): their 64 MFMAs, their epilogue in the registers (`out`
    // stored from them), then the 64 stage-2 MFMAs that consume them -- only 32 of stage 1's 256 accumulator registers are ever live
    // (all eight tiles at once: 772 registers spilled).
#pragma unroll
    for (int tp = 0; tp < T1 / 2; ++tp) {
      f32x16 acc1[2];
#pragma unroll
      for (int u = 0; u < 2; ++u)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc1[u][r] = 0.f;
      // (filter pieces one step ahead of their eight MFMAs: left to itself hipcc reads them right in front and the wave waits out an
      // LDS round trip every sixteen MFMAs)
      f32x4 bw[2][2];
#pragma unroll
      for (int u = 0; u < 2; ++u) bw[0][u] = *reinterpret_cast<const f32x4*>(w3f + 32 * (2 * tp + u) * LDW3);
#pragma unroll
      for (int j = 0; j < TC1 / 8; ++j) {
        if (j + 1 < TC1 / 8) {
#pragma unroll
          for (int u = 0; u < 2; ++u) bw[(j + 1) & 1][u] = *reinterpret_cast<const f32x4*>(w3f + 32 * (2 * tp + u) * LDW3 + 8 * (j + 1));
        }
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
          for (int u = 0; u < 2; ++u) acc1[u] = __builtin_amdgcn_mfma_f32_32x32x2f32(bw[j & 1][u][e], xa[j][e], acc1[u], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
      }
      if (tp == T1 / 2 - 1) {  // stage 1 of this sub-tile is done: the next sub-tile's pixel operand takes the registers
#pragma unroll
        for (int j = 0; j < TC1 / 8; ++j) xa[j] = *reinterpret_cast<const f32x4*>(p.x + npix * TC1 + 8 * j + 4 * lh);
      }
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const int t = 2 * tp + u;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const int c0 = 32 * t + 8 * g + 4 * lh;
          const f32x4 sc = *reinterpret_cast<const f32x4*>(s3l + c0), sh = *reinterpret_cast<const f32x4*>(t3l + c0);
          f32x4 v;
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            v[e] = fmaxf(acc1[u][4 * g + e] * sc[e] + sh[e] + rid[u][g][e], lo);
            acc1[u][4 * g + e] = v[e];
          }
          *reinterpret_cast<f32x4*>(op + 32 * t + 8 * g) = v;
        }
      }
      __builtin_amdgcn_sched_barrier(0);
      if (has_id) {  // the following pair's identity pieces (the next sub-tile's first pair behind this one's last)
        const float* const ip = tp + 1 < T1 / 2 ? p.idt + pix * TCM + 4 * lh + 64 * (tp + 1) : p.idt + npix * TCM + 4 * lh;
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
          for (int g = 0; g < 4; ++g) rid[u][g] = *reinterpret_cast<const f32x4*>(ip + 32 * u + 8 * g);
      }
      if constexpr (CHAIN) {
      f32x4 aw[2][T2];
#pragma unroll
      for (int tn = 0; tn < T2; ++tn) aw[0][tn] = *reinterpret_cast<const f32x4*>(w1f + 32 * tn * LDW1 + 64 * tp);
#pragma unroll
      for (int ug = 0; ug < 8; ++ug) {  // (u, g): the pair's 8 groups of four stage-2 k-steps
        const int u = ug >> 2, g = ug & 3;
        if (ug + 1 < 8) {
#pragma unroll
          for (int tn = 0; tn < T2; ++tn)
            aw[(ug + 1) & 1][tn] = *reinterpret_cast<const f32x4*>(w1f + 32 * tn * LDW1 + 64 * tp + 32 * ((ug + 1) >> 2) + 8 * ((ug + 1) & 3));
        }
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
          for (int tn = 0; tn < T2; ++tn)
            acc2[tn] = __builtin_amdgcn_mfma_f32_32x32x2f32(aw[ug & 1][tn][e], acc1[u][4 * g + e], acc2[tn], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
      }
      }
    }
    // ---- stage-2 epilogue ----------------------------------------------------------------------------------------------------
    if constexpr (!CHAIN) continue;
    float* const zp = p.z + pix * TC2 + 4 * lh;
#pragma unroll
    for (int tn = 0; tn < T2; ++tn)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int c0 = 32 * tn + 8 * g + 4 * lh;
        const f32x4 sc = *reinterpret_cast<const f32x4*>(s1l + c0), sh = *reinterpret_cast<const f32x4*>(t1l + c0);
        f32x4 v;
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = fmaxf(acc2[tn][4 * g + e] * sc[e] + sh[e], 0.f);
        *reinterpret_cast<f32x4*>(zp + 32 * tn + 8 * g) = v;
      }
  }
}

}  // namespace

static int tail_launch(const TailArgs& a, bool chain, hipStream_t s) {
  int dev = 0, cus = 0;
  if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0)
    cus = 256;
  const int want = (a.nsub + 7) / 8;
  const int grid = want < cus ? want : cus;
  if (chain) bottleneck_tail_f32<true><<<grid, 512, 0, s>>>(a);
  else bottleneck_tail_f32<false><<<grid, 512, 0, s>>>(a);
  return RS_LAUNCH_RESULT();
}

extern "C" int rs_bottleneck_tail_f32(const float* x, const float* w3, const float* scale3, const float* shift3, const float* identity,
                                      const float* w1, const float* scale1, const float* shift1, float* out, float* z, long M,
                                      int C1, int Cmid, int C2, rs_stream_t stream) {
  if (!x || !w3 || !scale3 || !shift3 || !identity || !w1 || !scale1 || !shift1 || !out || !z) return RS_EINVAL;
  if (C1 != TC1 || Cmid != TCM || C2 != TC2 || M <= 0 || (M % 32) != 0 || M / 32 >= (1L << 31)) return RS_EINVAL;
  TailArgs a;
  a.x = x, a.w3 = w3, a.s3 = scale3, a.t3 = shift3, a.idt = identity, a.w1 = w1, a.s1 = scale1, a.t1 = shift1, a.out = out, a.z = z;
  a.nsub = (int)(M / 32), a.relu = 1;
  return tail_launch(a, true, (hipStream_t)stream);
}

extern "C" int rs_conv1x1_wave_f32(const float* x, const float* w, const float* scale, const float* shift, const float* residual, int relu,
                                   float* out, long M, int C1, int Cout, rs_stream_t stream) {
  if (!x || !w || !scale || !shift || !out) return RS_EINVAL;
  if (C1 != TC1 || Cout != TCM || M <= 0 || (M % 32) != 0 || M / 32 >= (1L << 31)) return RS_EINVAL;
  TailArgs a;
  a.x = x, a.w3 = w, a.s3 = scale, a.t3 = shift, a.idt = residual, a.w1 = nullptr, a.s1 = nullptr, a.t1 = nullptr, a.out = out, a.z = nullptr;
  a.nsub = (int)(M / 32), a.relu = relu ? 1 : 0;
  return tail_launch(a, false, (hipStream_t)stream);
}

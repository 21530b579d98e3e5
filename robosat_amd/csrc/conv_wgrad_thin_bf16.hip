// Weight gradient of the THIN 3x3 convolutions (Cout = 32: dec4 128->32 behind the x2 upsample, dec5 32->32; reference
// robosat/unet.py:106-107,139-140) on v_mfma_f32_32x32x16_bf16.  These layers carry 14 % of the forward FLOPs but only
// 32 output channels, so per pixel there are just 2*9*Cin*32 FLOPs against (Cin + 32)*2 bytes: the generic
// tap-per-block kernel (conv_wgrad_bf16.hip) re-streams dy and the input once per filter tap (9x the traffic through
// L2) and leaves the MFMAs idle.  Here ONE block owns all nine taps:
//
//   chunk  = an 8x8 patch of output pixels (64 = four MFMA k-steps of 16 pixels).  The block walks a contiguous run of
//            patches and keeps all 9 taps x [32 cout][32 cin] accumulators (9 x 16 VGPRs) in registers per wave.
//   LDS    = per patch: dy [64 pixels][32 cout] and ONE input halo tile [HH x HW source pixels][Cin] (pixel-major, as
//            it lies in HBM; with the fused nearest-x2 upsample the halo is kept at SOURCE resolution, 6x6 pixels).
//            Both are fetched once per patch (double buffered, register-staged under the MFMAs).
//   MFMA operands are read with ds_read_b64_tr_b16: within a 16-lane group, 4-lane sets supply the addresses of four
//            [pixel] rows (4 channels x 4 lanes = 16 channels each) and every lane receives its channel's column of that
//            4x16 block, i.e. the pixel-major -> K-major transpose MFMA needs, done by the LDS hardware.  Because each
//            row address is free, the filter-tap shift, the image border (zero-filled halo) and the >>1 of the upsample
//            are all just per-lane address arithmetic: nothing is ever re-gathered from HBM per tap.
//   waves  = Cin/32 cin tiles x (4 / (Cin/32)) k-step groups: 128 channels -> each wave owns one 32-channel tile and all
//            4 k-steps; 32 channels -> each wave owns one k-step (its own partial slice, summed by the reduce kernel).
//   output = partial slices [splits * KG][Cout][9 * Cin] fp32, reduced deterministically (no atomics) into KRSC dW.
//   wider layers (round 2): the grid's y / z dimensions walk 32-cout tiles and 128-cin slabs of a wider stride-1 3x3 layer
//            (the encoder's conv2's: 64 -> 64 ... 512 -> 512), every block re-reading the patch's dy slice and halo slab; on
//            layer1 (two cout tiles) that is 2.5x faster than nine tap-per-block passes, on the 128+ wide layers a wash.
#include "common.h"

namespace {

struct ThinArgs {
  const bf16_t* dy;   // [N][Ho][Wo][Cout]
  const bf16_t* src;  // [N][Hs][Ws][Cin]
  float* out;         // [slices][Cout][9*Cin]
  int N, Hs, Ws, Ho, Wo;
  int Cout, Cin;      // the layer's widths: block (x, y, z) owns couts [32y, 32y+32) and cins [32*CT*z, 32*CT*(z+1))
  int ppr, ppi;       // patches per row / per image
  int total_patches, patches_per_block;
};

typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ __amdgpu_buffer_rsrc_t tb_make_rsrc(const void* base, long bytes) {
  const unsigned int n = bytes > 0xFFFFFFFEL ? 0xFFFFFFFEu : (unsigned int)(bytes < 0 ? 0 : bytes);
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, (int)n, 0x00020000);
}

__device__ __forceinline__ bf16x8 tb_tr_read8(const unsigned char* p0, const unsigned char* p1) {
  typedef __attribute__((address_space(3))) s16x4* lds_ptr;
  const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_ptr)p0);
  const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_ptr)p1);
  s16x8 v;
  v[0] = lo[0];
  v[1] = lo[1];
  v[2] = lo[2];
  v[3] = lo[3];
  v[4] = hi[0];
  v[5] = hi[1];
  v[6] = hi[2];
  v[7] = hi[3];
  return __builtin_bit_cast(bf16x8, v);
}

// CT = Cin / 32 (1, 2 or 4); UPS = 1: the input is read through the nearest-x2 upsample (source at half resolution)
template <int CT, int UPS>
__global__ __launch_bounds__(256, 2) void conv_wgrad_thin_bf16(const ThinArgs p) {
  constexpr int CIN = 32 * CT;
  constexpr int KG = 4 / CT;            // k-step groups (waves sharing a cin tile split the 4 k-steps)
  constexpr int NS = 4 / KG;            // k-steps per wave
  constexpr int ROWX = CIN * 2;         // bytes per halo pixel
  constexpr int HH = UPS ? 6 : 10;      // halo rows (source resolution)
  constexpr int HWU = UPS ? 6 : 10;     // halo columns in use
  constexpr int HWP = UPS ? 8 : 12;     // padded to a multiple of 4 (keeps the swizzle a function of x only)
  constexpr int HALOB = HH * HWP * ROWX;
  constexpr int DYB = 64 * 64;          // [64 pixels][32 cout] bf16
  constexpr int BUF = HALOB + DYB;
  constexpr int PCS = CIN / 8;          // 16-byte pieces per halo pixel
  constexpr int NHP = HH * HWP * PCS;   // halo pieces per patch
  constexpr int NH = (NHP + 255) / 256;  // per thread
  constexpr int RPB = 4 / CT;           // halo pixels per 256-byte bank row

  __shared__ __attribute__((aligned(16))) unsigned char smem[2 * BUF];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int ct = wave % CT, kg = wave / CT;

  const int pat0 = blockIdx.x * p.patches_per_block;
  int pat1 = pat0 + p.patches_per_block;
  if (pat1 > p.total_patches) pat1 = p.total_patches;

  const int co_base = blockIdx.y * 32, ci_base = blockIdx.z * CIN;
  const int dyrow = p.Cout * 2, xrow = p.Cin * 2;  // bytes per pixel in HBM
  const __amdgpu_buffer_rsrc_t rsrc_dy = tb_make_rsrc(p.dy, (long)p.N * p.Ho * p.Wo * dyrow);
  const __amdgpu_buffer_rsrc_t rsrc_x = tb_make_rsrc(p.src, (long)p.N * p.Hs * p.Ws * xrow);

  // ---- staging roles (fixed per thread): NH halo pieces + one dy piece -----------------------------------------
  int h_hy[NH], h_hx[NH], h_pc[NH], h_lds[NH];
#pragma unroll
  for (int j = 0; j < NH; ++j) {
    const int e = tid + 256 * j;
    const int hp = e / PCS, c = e - hp * PCS;
    h_hy[j] = hp / HWP;
    h_hx[j] = hp - h_hy[j] * HWP;
    h_pc[j] = c;
    const int cc = c >> 2;  // 64-byte piece (= cin tile) index, XOR-swizzled by the pixel's x
    const int sw = CT > 1 ? ((h_hx[j] / RPB) & (CT - 1)) : 0;
    h_lds[j] = (e < NHP) ? hp * ROWX + ((cc ^ sw) * 64) + (c & 3) * 16 : -1;
  }
  const int d_k = tid >> 2, d_c = tid & 3;  // dy: pixel of the patch / 16-byte piece (8 couts)

  u32x4 rh[NH], rd;
  auto load_patch = [&](int pat) __attribute__((always_inline)) {
    const int n = pat / p.ppi;
    const int rem = pat - n * p.ppi;
    const int pyi = rem / p.ppr, pxi = rem - pyi * p.ppr;
    const int oy0 = pyi * 8, ox0 = pxi * 8;
    const int sy0 = UPS ? (oy0 >> 1) - 1 : oy0 - 1;
    const int sx0 = UPS ? (ox0 >> 1) - 1 : ox0 - 1;
    const bool live = pat < pat1;
#pragma unroll
    for (int j = 0; j < NH; ++j) {
      const int sy = sy0 + h_hy[j], sx = sx0 + h_hx[j];
      const bool ok = live && h_lds[j] >= 0 && h_hx[j] < HWU && (unsigned)sy < (unsigned)p.Hs && (unsigned)sx < (unsigned)p.Ws;
      const long off = ((long)(n * p.Hs + sy) * p.Ws + sx) * xrow + ci_base * 2 + h_pc[j] * 16;
      rh[j] = __builtin_amdgcn_raw_buffer_load_b128(rsrc_x, ok ? (int)off : -1, 0, 0);
    }
    {
      const long m = (long)(n * p.Ho + oy0 + (d_k >> 3)) * p.Wo + ox0 + (d_k & 7);
      rd = __builtin_amdgcn_raw_buffer_load_b128(rsrc_dy, live ? (int)(m * dyrow + co_base * 2 + d_c * 16) : -1, 0, 0);
    }
  };
  auto store_patch = [&](int buf) __attribute__((always_inline)) {
    unsigned char* L = smem + buf * BUF;
#pragma unroll
    for (int j = 0; j < NH; ++j)
      if (h_lds[j] >= 0) *reinterpret_cast<u32x4*>(L + h_lds[j]) = rh[j];
    *reinterpret_cast<u32x4*>(L + HALOB + d_k * 64 + d_c * 16) = rd;
  };

  // ---- per-lane operand addressing ------------------------------------------------------------------------------
  // lane = 16*g + q: group g covers channels 16*(g&1)..+15 and pixels k = 16s + 8*(g>>1) + 4t + (q>>2);
  // lane q supplies the address of pixel row (q>>2), channels 4*(q&3)..+3 of its 16.
  const int g = lane >> 4, q = lane & 15;
  const int chb = (16 * (g & 1) + 4 * (q & 3)) * 2;  // byte offset inside a 32-channel (64-byte) piece
  const int jr = q >> 2;
  // dy: pixel k -> LDS row k (64 bytes)
  int aoff[NS][2];
#pragma unroll
  for (int si = 0; si < NS; ++si)
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      const int s = kg * NS + si;
      aoff[si][t] = HALOB + (16 * s + 8 * (g >> 1) + 4 * t + jr) * 64 + chb;
    }
  // halo: pixel (py, px) of the patch under tap (ky, kx) -> halo pixel (hy, hx); address = yoff + xoff
  int yoff[NS][3], xoff[2][3];
#pragma unroll
  for (int si = 0; si < NS; ++si)
#pragma unroll
    for (int ky = 0; ky < 3; ++ky) {
      const int s = kg * NS + si;
      const int py = 2 * s + (g >> 1);
      const int hy = UPS ? ((py + ky - 1) >> 1) + 1 : py + ky;
      yoff[si][ky] = hy * HWP * ROWX;
    }
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int kx = 0; kx < 3; ++kx) {
      const int px = 4 * t + jr;
      const int hx = UPS ? ((px + kx - 1) >> 1) + 1 : px + kx;
      const int sw = CT > 1 ? ((hx / RPB) & (CT - 1)) : 0;
      xoff[t][kx] = hx * ROWX + ((ct ^ sw) * 64) + chb;
    }

  f32x16 acc[9];
#pragma unroll
  for (int a = 0; a < 9; ++a)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[a][r] = 0.f;

  if (pat0 < pat1) {
    load_patch(pat0);
    store_patch(0);
    __syncthreads();
    for (int pat = pat0; pat < pat1; ++pat) {
      const int it = pat - pat0;
      const unsigned char* L = smem + (it & 1) * BUF;
      load_patch(pat + 1);  // zeros past the block's range
#pragma unroll
      for (int si = 0; si < NS; ++si) {
        const bf16x8 a = tb_tr_read8(L + aoff[si][0], L + aoff[si][1]);
#pragma unroll
        for (int ky = 0; ky < 3; ++ky)
#pragma unroll
          for (int kx = 0; kx < 3; ++kx) {
            const bf16x8 b = tb_tr_read8(L + yoff[si][ky] + xoff[0][kx], L + yoff[si][ky] + xoff[1][kx]);
            acc[ky * 3 + kx] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[ky * 3 + kx], 0, 0, 0);
          }
      }
      store_patch((it + 1) & 1);
      __syncthreads();
    }
  }

  // D[i][j]: i = cout = (r&3) + 8*(r>>2) + 4*(lane>>5), j = cin (within this wave's tile) = lane&31
  const long K = 9L * p.Cin;
  float* out = p.out + ((long)blockIdx.x * KG + kg) * p.Cout * K + (long)co_base * K + ci_base + ct * 32 + (lane & 31);
#pragma unroll
  for (int tap = 0; tap < 9; ++tap)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int co = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
      out[(long)co * K + tap * p.Cin] = acc[tap][r];
    }
}

}  // namespace

// Shared with conv_wgrad_bf16.hip (declared in common.h): is `d` one of the thin 3x3 layers, and how is it split?
int rs_wgrad_thin_plan(const rs_conv_desc* d, int* blocks, int* slices) {
  if (!d || d->stem || d->C2 != 0 || d->kh != 3 || d->kw != 3 || d->stride != 1 || d->pad != 1) return 0;
  if (d->Cout <= 0 || (d->Cout % 32) != 0) return 0;
  if (d->C1 != 32 && d->C1 != 64 && (d->C1 % 128) != 0) return 0;  // slabs of 32 / 64 / 128 input channels per block
  if (d->ups != 0 && d->ups != 1) return 0;
  if ((d->Ho % 8) || (d->Wo % 8)) return 0;
  if (d->ups == 1 && (d->Ho != 2 * d->Hs || d->Wo != 2 * d->Ws)) return 0;
  if (d->ups == 0 && (d->Ho != d->Hs || d->Wo != d->Ws)) return 0;
  // 32-bit byte offsets
  if ((long)d->N * d->Ho * d->Wo * d->Cout * 2 >= (1L << 31) || (long)d->N * d->Hs * d->Ws * d->C1 * 2 >= (1L << 31)) return 0;
  const long patches = (long)d->N * (d->Ho / 8) * (d->Wo / 8);
  const int slab = d->C1 < 128 ? d->C1 : 128;
  const long groups = (long)(d->Cout / 32) * (d->C1 / slab);  // (cout tile, cin slab) pairs: grid y x z
  long nb;
  if (groups == 1) {
    nb = patches < 512 ? patches : 512;  // the Cout = 32 decoder tail: one full wave of blocks (2 per CU)
  } else {
    // wider layers (round 2: the encoder's stride-1 3x3 convolutions): ~256 blocks over all groups, and enough patches per block
    // for its [32][9 * slab] fp32 partial tile (up to 147 KB) to be a small part of what it moves; otherwise the generic kernel
    if (d->ups != 0) return 0;
    nb = 256 / groups;
    if (nb < 1) nb = 1;
    if (patches / nb < 8) return 0;
  }
  const long ppb = (patches + nb - 1) / nb;
  nb = (patches + ppb - 1) / ppb;
  if (blocks) *blocks = (int)nb;
  if (slices) *slices = (int)nb * (4 / (slab / 32));
  return 1;
}

int rs_wgrad_thin_launch(const rs_conv_desc* d, const void* dy, const void* src, float* partial, void* stream) {
  int blocks = 0, slices = 0;
  if (!rs_wgrad_thin_plan(d, &blocks, &slices)) return RS_EINVAL;
  ThinArgs a;
  a.dy = reinterpret_cast<const bf16_t*>(dy);
  a.src = reinterpret_cast<const bf16_t*>(src);
  a.out = partial;
  a.N = d->N;
  a.Hs = d->Hs;
  a.Ws = d->Ws;
  a.Ho = d->Ho;
  a.Wo = d->Wo;
  a.Cout = d->Cout;
  a.Cin = d->C1;
  a.ppr = d->Wo / 8;
  a.ppi = (d->Ho / 8) * (d->Wo / 8);
  a.total_patches = d->N * a.ppi;
  a.patches_per_block = (a.total_patches + blocks - 1) / blocks;
  hipStream_t s = (hipStream_t)stream;
  const int slab = d->C1 < 128 ? d->C1 : 128;
  const int ct = slab / 32;
  const dim3 grid(blocks, d->Cout / 32, d->C1 / slab);
  if (d->ups == 0) {
    if (ct == 1) conv_wgrad_thin_bf16<1, 0><<<grid, 256, 0, s>>>(a);
    else if (ct == 2) conv_wgrad_thin_bf16<2, 0><<<grid, 256, 0, s>>>(a);
    else conv_wgrad_thin_bf16<4, 0><<<grid, 256, 0, s>>>(a);
  } else {
    if (ct == 1) conv_wgrad_thin_bf16<1, 1><<<grid, 256, 0, s>>>(a);
    else if (ct == 2) conv_wgrad_thin_bf16<2, 1><<<grid, 256, 0, s>>>(a);
    else conv_wgrad_thin_bf16<4, 1><<<grid, 256, 0, s>>>(a);
  }
  return RS_LAUNCH_RESULT();
}

// The 7x7 / stride-2 stem convolution (resnet.conv1, reference robosat/unet.py:122: 3 or 4 bands -> 64 channels) in bf16:
// forward and weight gradient.  0.7 % of the network's FLOPs but, left on the fp32 kernels, 6 % of a bf16 training step
// (the fp32 activations around it are 537 MB per bs-32 pass): both kernels here are HBM-bound.
//
// The input has 4 channels per pixel (NHWC4, 8 bytes), so a 128-byte "one tap x 64 channels" K-chunk does not exist.
// Instead one block owns an output PATCH and ALL 49 taps: the input halo of the patch is loaded into LDS once and the
// filter-tap shift is address arithmetic on LDS reads:
//   forward:  K is ordered (ky, kx, ci) with kx padded to 8 (weights packed [64][7][8][4], rs_pack_stem_weight): a
//             ds_read_b128 at pixel (2*oy+ky, 2*ox+kx) returns 2 taps x 4 channels = 8 consecutive K -- exactly one
//             MFMA operand fragment.  14 k-steps per patch, weights resident in LDS for the block's whole patch run.
//   wgrad:    the reduction runs over pixels, the output columns are (kx, ci) of one filter row ky: with
//             ds_read_b64_tr_b16 every 4-lane set addresses one pixel ROW of the 4x16 block and every lane of the set
//             its own TAP (the 8 bytes it reads are one input pixel = 4 channels), so the hardware transpose delivers
//             [pixel k][(kx, ci)] without ever materialising an im2col row.
#include "common.h"

namespace {

typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ __amdgpu_buffer_rsrc_t sb_make_rsrc(const void* base, long bytes) {
  const unsigned int n = bytes > 0xFFFFFFFEL ? 0xFFFFFFFEu : (unsigned int)(bytes < 0 ? 0 : bytes);
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, (int)n, 0x00020000);
}

__device__ __forceinline__ bf16x8 sb_tr_read8(const unsigned char* p0, const unsigned char* p1) {
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

// ---------------------------------------------------------------------------------------------------------------------
// forward
// ---------------------------------------------------------------------------------------------------------------------
struct StemFwdArgs {
  const bf16_t* x;    // [N][H][W][4]
  const bf16_t* w;    // packed [64][7][8][4]
  const float* scale;  // optional per-cout epilogue (eval-mode BatchNorm)
  const float* shift;
  bf16_t* out;        // [N][Ho][Wo][64]
  int N, H, W, Ho, Wo, relu;
  int ppr, ppi, total_patches, patches_per_block;
};

constexpr int FPH = 8, FPW = 16;                 // output patch
constexpr int FHR = 2 * FPH + 5;                 // 21 halo rows
constexpr int FHC = 2 * FPW + 6;                 // 38 halo pixels in use (incl. the zero 8th tap)
constexpr int FROW = 384;                        // halo row stride in bytes (48 pixels): 2 rows = 0 mod 256 (bank layout)
constexpr int FHALO = FHR * FROW;                // 8064
constexpr int FWROW = 464;                       // weight row stride (224 K + 8 pad) * 2 B: 29 sixteen-byte slots, odd
constexpr int FWB = 64 * FWROW;                  // 29696
constexpr int FSROW = 72;                        // staging row: 64 couts + 8 pad (bf16)
constexpr int FSTAGE = FPH * FPW * FSROW * 2;    // 18432
constexpr int FNP = (FHR * FHC + 255) / 256;     // halo pixels per thread (4)

__global__ __launch_bounds__(256, 2) void stem_fwd_bf16_kernel(const StemFwdArgs p) {
  __shared__ __attribute__((aligned(16))) unsigned char smem[FWB + 2 * FHALO + FSTAGE];
  unsigned char* wl = smem;
  unsigned char* hl = smem + FWB;
  bf16_t* stage = reinterpret_cast<bf16_t*>(smem + FWB + 2 * FHALO);

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int pat0 = blockIdx.x * p.patches_per_block;
  int pat1 = pat0 + p.patches_per_block;
  if (pat1 > p.total_patches) pat1 = p.total_patches;
  if (pat0 >= pat1) return;

  // weights -> LDS (once per block): 64 rows x 28 pieces of 16 bytes
  for (int e = tid; e < 64 * 28; e += 256) {
    const int r = e / 28, c = e - r * 28;
    *reinterpret_cast<u32x4*>(wl + r * FWROW + c * 16) = *reinterpret_cast<const u32x4*>(p.w + (long)r * 224 + c * 8);
  }

  const __amdgpu_buffer_rsrc_t rsrc = sb_make_rsrc(p.x, (long)p.N * p.H * p.W * 8);
  int h_y[FNP], h_x[FNP];
#pragma unroll
  for (int j = 0; j < FNP; ++j) {
    const int e = tid + 256 * j;
    h_y[j] = e / FHC;
    h_x[j] = e - h_y[j] * FHC;
    if (e >= FHR * FHC) h_y[j] = -1;
  }
  u32x2 rh[FNP];
  auto load_halo = [&](int pat) __attribute__((always_inline)) {
    const int n = pat / p.ppi, rem = pat - n * p.ppi;
    const int pyi = rem / p.ppr, pxi = rem - pyi * p.ppr;
    const int iy0 = 2 * pyi * FPH - 3, ix0 = 2 * pxi * FPW - 3;
    const bool live = pat < pat1;
#pragma unroll
    for (int j = 0; j < FNP; ++j) {
      const int iy = iy0 + h_y[j], ix = ix0 + h_x[j];
      const bool ok = live && h_y[j] >= 0 && (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W;
      const long off = ((long)(n * p.H + iy) * p.W + ix) * 8;
      rh[j] = __builtin_amdgcn_raw_buffer_load_b64(rsrc, ok ? (int)off : -1, 0, 0);
    }
  };
  auto store_halo = [&](int buf) __attribute__((always_inline)) {
    unsigned char* L = hl + buf * FHALO;
#pragma unroll
    for (int j = 0; j < FNP; ++j)
      if (h_y[j] >= 0) *reinterpret_cast<u32x2*>(L + h_y[j] * FROW + h_x[j] * 8) = rh[j];
  };

  // this wave's 32 pixels of the patch: rows 2*wave, 2*wave+1; lane = 16*(row parity) + column
  const int j = lane & 31, hsel = lane >> 5;
  const int py = 2 * wave + (j >> 4), px = j & 15;
  const int poff = (2 * py) * FROW + (2 * px) * 8 + hsel * 16;  // + ky*FROW + s*32
  const int woff = (lane & 31) * FWROW + hsel * 16;             // + tn*32*FWROW + (ky*32 + 16*s)*2

  float sc[2][16], sh[2][16];
#pragma unroll
  for (int tn = 0; tn < 2; ++tn)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int co = tn * 32 + (r & 3) + 8 * (r >> 2) + 4 * hsel;
      sc[tn][r] = p.scale ? p.scale[co] : 1.f;
      sh[tn][r] = p.shift ? p.shift[co] : 0.f;
    }

  load_halo(pat0);
  store_halo(0);
  __syncthreads();
  for (int pat = pat0; pat < pat1; ++pat) {
    const int it = pat - pat0;
    const unsigned char* L = hl + (it & 1) * FHALO;
    load_halo(pat + 1);
    f32x16 acc[2];
#pragma unroll
    for (int tn = 0; tn < 2; ++tn)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[tn][r] = 0.f;
#pragma unroll
    for (int ky = 0; ky < 7; ++ky)
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        const bf16x8 a = *reinterpret_cast<const bf16x8*>(L + poff + ky * FROW + s * 32);
#pragma unroll
        for (int tn = 0; tn < 2; ++tn) {
          const bf16x8 b = *reinterpret_cast<const bf16x8*>(wl + woff + tn * 32 * FWROW + (ky * 32 + 16 * s) * 2);
          acc[tn] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b, a, acc[tn], 0, 0, 0);  // D[i = cout][j = pixel]
        }
      }
    // epilogue: scale/shift/relu in registers -> bf16 staging [pixel][cout] -> 16-byte row-wise stores
    {
      const int pl = wave * 32 + j;  // patch-local pixel (row-major 8 x 16)
#pragma unroll
      for (int tn = 0; tn < 2; ++tn)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          bf16x4 v;
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            float t = acc[tn][4 * g + e] * sc[tn][4 * g + e] + sh[tn][4 * g + e];
            if (p.relu) t = fmaxf(t, 0.f);
            v[e] = (bf16_t)t;
          }
          *reinterpret_cast<bf16x4*>(&stage[pl * FSROW + tn * 32 + 8 * g + 4 * hsel]) = v;
        }
    }
    __syncthreads();
    {
      const int n = pat / p.ppi, rem = pat - n * p.ppi;
      const int pyi = rem / p.ppr, pxi = rem - pyi * p.ppr;
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const int e = tid + 256 * k;  // 128 pixels x 8 pieces
        const int pl = e >> 3, c = e & 7;
        const int oy = pyi * FPH + (pl >> 4), ox = pxi * FPW + (pl & 15);
        const u32x4 v = *reinterpret_cast<const u32x4*>(&stage[pl * FSROW + c * 8]);
        *reinterpret_cast<u32x4*>(p.out + ((long)(n * p.Ho + oy) * p.Wo + ox) * 64 + c * 8) = v;
      }
    }
    store_halo((it + 1) & 1);
    __syncthreads();
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// weight gradient: dW[co][ky][kx][ci] = sum_pixels dy[pixel][co] * x[2*oy + ky - 3][2*ox + kx - 3][ci]
// ---------------------------------------------------------------------------------------------------------------------
struct StemWgradArgs {
  const bf16_t* dy;  // [N][Ho][Wo][64]
  const bf16_t* x;   // [N][H][W][4]
  float* out;        // [blocks][64][224]  (packed [cout][ky][kx padded to 8][ci padded to 4])
  int N, H, W, Ho, Wo;
  int ppr, ppi, total_patches, patches_per_block;
};

constexpr int WHR = 21, WHC = 22;          // halo rows / pixels in use for an 8x8 output patch
constexpr int WROW = 192;                  // halo row stride: 24 pixels
constexpr int WHALO = WHR * WROW;          // 4032
constexpr int WDY = 64 * 128;              // dy patch [64 pixels][64 couts] bf16
constexpr int WBUF = WHALO + WDY;
constexpr int WNP = (WHR * WHC + 255) / 256;  // 2

__global__ __launch_bounds__(256, 2) void stem_wgrad_bf16_kernel(const StemWgradArgs p) {
  __shared__ __attribute__((aligned(16))) unsigned char smem[2 * WBUF];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int pat0 = blockIdx.x * p.patches_per_block;
  int pat1 = pat0 + p.patches_per_block;
  if (pat1 > p.total_patches) pat1 = p.total_patches;

  const __amdgpu_buffer_rsrc_t rsrc_x = sb_make_rsrc(p.x, (long)p.N * p.H * p.W * 8);
  const __amdgpu_buffer_rsrc_t rsrc_dy = sb_make_rsrc(p.dy, (long)p.N * p.Ho * p.Wo * 128);
  int h_y[WNP], h_x[WNP];
#pragma unroll
  for (int j = 0; j < WNP; ++j) {
    const int e = tid + 256 * j;
    h_y[j] = e / WHC;
    h_x[j] = e - h_y[j] * WHC;
    if (e >= WHR * WHC) h_y[j] = -1;
  }
  // dy: 64 pixels x 8 pieces of 16 bytes = 512 pieces, two per thread; 64-byte halves of a row XOR-swizzled by (row>>1)&1
  u32x2 rh[WNP];
  u32x4 rd[2];
  auto load_patch = [&](int pat) __attribute__((always_inline)) {
    const int n = pat / p.ppi, rem = pat - n * p.ppi;
    const int pyi = rem / p.ppr, pxi = rem - pyi * p.ppr;
    const int iy0 = 16 * pyi - 3, ix0 = 16 * pxi - 3;
    const bool live = pat < pat1;
#pragma unroll
    for (int j = 0; j < WNP; ++j) {
      const int iy = iy0 + h_y[j], ix = ix0 + h_x[j];
      const bool ok = live && h_y[j] >= 0 && (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W;
      const long off = ((long)(n * p.H + iy) * p.W + ix) * 8;
      rh[j] = __builtin_amdgcn_raw_buffer_load_b64(rsrc_x, ok ? (int)off : -1, 0, 0);
    }
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      const int e = tid + 256 * k;
      const int px = e >> 3, c = e & 7;
      const long m = (long)(n * p.Ho + 8 * pyi + (px >> 3)) * p.Wo + 8 * pxi + (px & 7);
      rd[k] = __builtin_amdgcn_raw_buffer_load_b128(rsrc_dy, live ? (int)(m * 128 + c * 16) : -1, 0, 0);
    }
  };
  auto store_patch = [&](int buf) __attribute__((always_inline)) {
    unsigned char* L = smem + buf * WBUF;
#pragma unroll
    for (int j = 0; j < WNP; ++j)
      if (h_y[j] >= 0) *reinterpret_cast<u32x2*>(L + h_y[j] * WROW + h_x[j] * 8) = rh[j];
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      const int e = tid + 256 * k;
      const int px = e >> 3, c = e & 7;
      const int pos = (((c >> 2) ^ ((px >> 1) & 1)) << 2) | (c & 3);
      *reinterpret_cast<u32x4*>(L + WHALO + px * 128 + pos * 16) = rd[k];
    }
  };

  // operand addressing (ds_read_b64_tr_b16): lane = 16*g + q, row jr = q>>2 of the 4 pixel rows of a read
  const int g = lane >> 4, q = lane & 15, jr = q >> 2;
  // A = dy: 32-cout tile tm, channels 16*(g&1) + 4*(q&3) of the tile; pixel k = 16s + 8*(g>>1) + 4t + jr
  int aoff[2][2];  // [tm][t], + s*16*128
#pragma unroll
  for (int tm = 0; tm < 2; ++tm)
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      const int k = 8 * (g >> 1) + 4 * t + jr;
      const int sw = (k >> 1) & 1;  // (pixel>>1)&1: 16s keeps it
      aoff[tm][t] = WHALO + k * 128 + ((tm ^ sw) * 64) + (16 * (g & 1) + 4 * (q & 3)) * 2;
    }
  // B = input halo: column (kx, ci): this lane's 8 bytes are the input pixel under tap kx = (q&3) + 4*(g&1);
  // pixel k -> (py, px) = (k>>3, k&7) of the 8x8 patch -> halo (2*py + ky, 2*px + kx)
  const int kx = (q & 3) + 4 * (g & 1);
  int boff[2];  // [t], + s-dependent row + ky*WROW
#pragma unroll
  for (int t = 0; t < 2; ++t) {
    const int k = 8 * (g >> 1) + 4 * t + jr;  // within the k-step: k&7 = 4t + jr is px; bit 3 selects the row
    boff[t] = (2 * (k >> 3)) * WROW + (2 * (k & 7) + kx) * 8;
  }
  const int ky0 = wave, ky1 = wave + 4;  // filter rows of this wave (ky1 valid for waves 0..2)

  f32x16 acc[2][2];  // [ky slot][cout tile]
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

  if (pat0 < pat1) {
    load_patch(pat0);
    store_patch(0);
    __syncthreads();
    for (int pat = pat0; pat < pat1; ++pat) {
      const int it = pat - pat0;
      const unsigned char* L = smem + (it & 1) * WBUF;
      load_patch(pat + 1);
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        // k-step s covers pixels 16s..16s+15 = patch rows 2s, 2s+1
        bf16x8 a[2];
#pragma unroll
        for (int tm = 0; tm < 2; ++tm) a[tm] = sb_tr_read8(L + aoff[tm][0] + s * 2048, L + aoff[tm][1] + s * 2048);
        const int rowb = (4 * s) * WROW;  // halo row of patch row 2s is 2*(2s) = 4s
        {
          const bf16x8 b = sb_tr_read8(L + rowb + ky0 * WROW + boff[0], L + rowb + ky0 * WROW + boff[1]);
#pragma unroll
          for (int tm = 0; tm < 2; ++tm) acc[0][tm] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[tm], b, acc[0][tm], 0, 0, 0);
        }
        if (ky1 < 7) {  // wave-uniform
          const bf16x8 b = sb_tr_read8(L + rowb + ky1 * WROW + boff[0], L + rowb + ky1 * WROW + boff[1]);
#pragma unroll
          for (int tm = 0; tm < 2; ++tm) acc[1][tm] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[tm], b, acc[1][tm], 0, 0, 0);
        }
      }
      store_patch((it + 1) & 1);
      __syncthreads();
    }
  }
  // D[i = cout (tile-local)][j = (kx, ci)]: out[block][co][ky*32 + j]
  float* out = p.out + (long)blockIdx.x * 64 * 224;
#pragma unroll
  for (int a = 0; a < 2; ++a) {
    const int ky = a == 0 ? ky0 : ky1;
    if (ky >= 7) continue;
#pragma unroll
    for (int tm = 0; tm < 2; ++tm)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int co = tm * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        out[(long)co * 224 + ky * 32 + (lane & 31)] = acc[a][tm][r];
      }
  }
}

// fp32 KRSC [Cout][kh][kw<=8][Cin<=4] -> packed bf16 [Cout][kh][8][4]
__global__ void pack_stem_weight_bf16_kernel(const float* __restrict__ w, bf16_t* __restrict__ out, int Cout, int kh, int kw,
                                             int Cin) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  const int total = Cout * kh * 32;
  if (idx >= total) return;
  const int c = idx & 3, s = (idx >> 2) & 7, r = (idx >> 5) % kh, co = (idx >> 5) / kh;
  float v = 0.f;
  if (s < kw && c < Cin) v = w[((co * kh + r) * kw + s) * Cin + c];
  out[idx] = (bf16_t)v;
}

// NCHW fp32 -> NHWC4 bf16 (the bf16 path's image upload layout)
__global__ void nchw_to_nhwc4_bf16_kernel(const float* __restrict__ x, bf16_t* __restrict__ y, int C, long HW, long total) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  const long n = i / HW, hw = i - n * HW;
  const float* px = x + n * C * HW + hw;
  f32x4 v;
  v[0] = px[0];
  v[1] = C > 1 ? px[HW] : 0.f;
  v[2] = C > 2 ? px[2 * HW] : 0.f;
  v[3] = C > 3 ? px[3 * HW] : 0.f;
  rs_st4(y + i * 4, v);
}

bool stem_shape_ok(int N, int H, int W) {
  return N > 0 && H > 0 && W > 0 && (H % 32) == 0 && (W % 32) == 0 && (long)N * H * W * 8 < (1L << 31) &&
         (long)N * (H / 2) * (W / 2) * 128 < (1L << 31);
}

}  // namespace

extern "C" int rs_stem_conv_fwd_bf16(const rs_bf16* x, const rs_bf16* w_packed, const float* scale, const float* shift,
                                     rs_bf16* out, int N, int H, int W, int relu, rs_stream_t stream) {
  if (!x || !w_packed || !out || !stem_shape_ok(N, H, W)) return RS_EINVAL;
  StemFwdArgs a;
  a.x = reinterpret_cast<const bf16_t*>(x);
  a.w = reinterpret_cast<const bf16_t*>(w_packed);
  a.scale = scale;
  a.shift = shift;
  a.out = reinterpret_cast<bf16_t*>(out);
  a.N = N;
  a.H = H;
  a.W = W;
  a.Ho = H / 2;
  a.Wo = W / 2;
  a.relu = relu;
  a.ppr = a.Wo / FPW;
  a.ppi = (a.Ho / FPH) * a.ppr;
  a.total_patches = N * a.ppi;
  int blocks = a.total_patches < 2048 ? a.total_patches : 2048;
  a.patches_per_block = (a.total_patches + blocks - 1) / blocks;
  blocks = (a.total_patches + a.patches_per_block - 1) / a.patches_per_block;
  stem_fwd_bf16_kernel<<<blocks, 256, 0, (hipStream_t)stream>>>(a);
  return RS_LAUNCH_RESULT();
}

static int stem_wgrad_blocks(int N, int H, int W) {
  const long patches = (long)N * (H / 16) * (W / 16);
  return (int)(patches < 512 ? patches : 512);
}

extern "C" long rs_stem_conv_wgrad_bf16_workspace_bytes(int N, int H, int W) {
  if (!stem_shape_ok(N, H, W)) return RS_EINVAL;
  const long n = 64L * 224;
  const int blocks = stem_wgrad_blocks(N, H, W);
  return (blocks * n + rs_reduce_scratch_floats(n, blocks)) * (long)sizeof(float);
}

extern "C" int rs_stem_conv_wgrad_bf16(const rs_bf16* dy, const rs_bf16* x, float* dw_packed, int N, int H, int W,
                                       void* workspace, rs_stream_t stream) {
  if (!dy || !x || !dw_packed || !workspace || !stem_shape_ok(N, H, W)) return RS_EINVAL;
  StemWgradArgs a;
  a.dy = reinterpret_cast<const bf16_t*>(dy);
  a.x = reinterpret_cast<const bf16_t*>(x);
  a.out = reinterpret_cast<float*>(workspace);
  a.N = N;
  a.H = H;
  a.W = W;
  a.Ho = H / 2;
  a.Wo = W / 2;
  a.ppr = a.Wo / 8;
  a.ppi = (a.Ho / 8) * a.ppr;
  a.total_patches = N * a.ppi;
  int blocks = stem_wgrad_blocks(N, H, W);
  a.patches_per_block = (a.total_patches + blocks - 1) / blocks;
  const int used = (a.total_patches + a.patches_per_block - 1) / a.patches_per_block;
  stem_wgrad_bf16_kernel<<<used, 256, 0, (hipStream_t)stream>>>(a);
  const int rc = RS_LAUNCH_RESULT();
  if (rc) return rc;
  const long n = 64L * 224;
  return rs_reduce_splits(a.out, dw_packed, n, used, a.out + (long)blocks * n, stream);
}

extern "C" int rs_pack_stem_weight_bf16(const float* w_krsc, rs_bf16* packed, int Cout, int kh, int kw, int Cin,
                                        rs_stream_t stream) {
  if (!w_krsc || !packed || Cout <= 0 || kh <= 0 || kw <= 0 || kw > 8 || Cin <= 0 || Cin > 4) return RS_EINVAL;
  const int total = Cout * kh * 32;
  pack_stem_weight_bf16_kernel<<<rs_cdiv(total, 256), 256, 0, (hipStream_t)stream>>>(w_krsc, reinterpret_cast<bf16_t*>(packed),
                                                                                     Cout, kh, kw, Cin);
  return RS_LAUNCH_RESULT();
}

extern "C" int rs_nchw_to_nhwc4_bf16(const float* x, rs_bf16* y, int N, int C, int H, int W, rs_stream_t stream) {
  if (!x || !y || N <= 0 || C <= 0 || C > 4 || H <= 0 || W <= 0) return RS_EINVAL;
  const long HW = (long)H * W, total = (long)N * HW;
  nchw_to_nhwc4_bf16_kernel<<<rs_cdiv(total, 256), 256, 0, (hipStream_t)stream>>>(x, reinterpret_cast<bf16_t*>(y), C, HW, total);
  return RS_LAUNCH_RESULT();
}

// Implicit-GEMM convolution for gfx950 with LDS-DMA staging, templated on the activation type:
//   bf16_t : v_mfma_f32_32x32x16_bf16 (bf16 operands, fp32 accumulation; dense chip peak ~2.5 PFLOP/s): the bf16 training
//            path of BASELINE configs[2] (`rs train ... bf16`);
//   float  : v_mfma_f32_32x32x2_f32 (exact fp32, 157 TFLOP/s): every non-stem convolution of the fp32 parity path
//            (rs_conv2d_fwd; the 7x7 stem keeps the register-staged kernel of conv_igemm.hip).
// A K-chunk is always a 128-byte row per pixel (64 bf16 / 32 fp32 channels; 64-byte rows for the 32-channel bf16 layers),
// so both types share the LDS image, the swizzle and the fragment addressing; only the MFMA issue differs.
//
// Same operator as conv_igemm.hip (every nn.Conv2d / F.interpolate / torch.cat of UNet.forward, reference
// robosat/unet.py:122-141, and through `ups = 2` every data-gradient convolution of loss.backward(),
// tools/train.py:186), same GEMM view (M = N*Ho*Wo pixels, N = Cout, K = taps x Cin), same fused gather and
// epilogue.  What changes with 16x the MFMA rate is where the time goes, so the structure is re-balanced:
//
//   K-chunk = one filter tap x KC channels (KC = 64: 128-byte rows; KC = 32 for the 32-channel layers), i.e.
//            KC/16 MFMA k-steps per barrier.
//   gather  = the (tap, output pixel) -> source pixel map is computed ONCE per block into an LDS table
//            (taps x BM ints, -1 = padding / zero-insert hole / tail row); per chunk a thread fetches its rows'
//            entries with one ds_read and forms byte offsets with a multiply-add: the fp32 kernel's per-chunk
//            coordinate arithmetic (~10 VALU per load) would no longer hide under 8x shorter MFMA phases.
//   HBM->LDS = LDS-DMA (buffer_load_dwordx4 ... lds through SRSRC descriptors; offset -1 => the hardware writes zeros):
//            no VGPR round trip and no ds_write -- at bf16 MFMA rates the 32 KB per chunk a register-staged tile pushes
//            through ds_write_b128 (~79 B/clk/CU) costs as many LDS cycles as the MFMAs of the chunk take.  One wave
//            instruction = 1 KiB = 8 (16) whole rows, every lane with its own source offset (the gather).
//   LDS     = UNPADDED rows of KC bf16, 16-byte pieces XOR-swizzled: piece c of row r is stored at position
//            c ^ f(r), f(r) = (r>>1)&7 for 128-byte rows, (r>>2)&3 for 64-byte rows.  The DMA image is lane-linear
//            (lane l -> byte 16*l), so the swizzle is applied on the source side: lane l fetches piece (l % CPR) ^ f(row).
//            MFMA fragment reads (ds_read_b128: lane l reads row l&31, piece 2s + (l>>5)) hit 16 distinct 16-byte slots
//            in each of the instruction's four 16-lane groups.
//   MFMA    = 32x32x16: lane l feeds A[i = l&31][k = 8*(l>>5) .. +7] and B[k = 8*(l>>5) .. +7][j = l&31]: one b128
//            per 32-row sub-tile per k-step.  The weight fragment is the A operand: D[i = cout][j = pixel].
//   store   = accumulators -> LDS [pixel][cout] fp32 -> row-wise: 8 couts per thread, fp32 scale/shift, bf16 residual /
//            ReLU / ReLU-mask, one 16-byte bf16x8 store.
#include "conv_igemm_dma_kernel.h"

namespace {

// fp32 KRSC [Cout][taps][Cin] -> bf16 data-gradient weights [Cin][taps][Cout], taps flipped (cf. conv_wgrad.hip)
__global__ void pack_dgrad_weight_bf16_kernel(const float* __restrict__ w, bf16_t* __restrict__ out, int Cout, int taps,
                                              int Cin) {
  __shared__ float tile[32][33];
  const int tap = blockIdx.z;
  const int ci0 = blockIdx.x * 32, co0 = blockIdx.y * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 256 threads: 32 x 8
  for (int r = ty; r < 32; r += 8) {
    const int co = co0 + r, ci = ci0 + tx;
    tile[r][tx] = (co < Cout && ci < Cin) ? w[((long)co * taps + tap) * Cin + ci] : 0.f;
  }
  __syncthreads();
  const int ftap = taps - 1 - tap;
  for (int r = ty; r < 32; r += 8) {
    const int ci = ci0 + r, co = co0 + tx;
    if (ci < Cin && co < Cout) out[((long)ci * taps + ftap) * Cout + co] = (bf16_t)tile[tx][r];
  }
}

// The same for MANY weight tensors in one launch (the bf16 compute copies of a whole network after an optimizer step):
// block -> (item, 32x32 tile); one read of the fp32 tile feeds both the bf16 KRSC copy and the transposed, tap-flipped
// data-gradient copy.
template <typename T>
__global__ __launch_bounds__(256) void weight_prep_kernel(const rs_wprep_item* __restrict__ items, int n) {
  __shared__ float tile[32][33];
  const int bid = blockIdx.x;
  int lo = 0, hi = n - 1;  // last item with tile_begin <= bid (block-uniform)
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (items[mid].tile_begin <= bid) lo = mid;
    else hi = mid - 1;
  }
  const rs_wprep_item it = items[lo];
  int t = bid - it.tile_begin;
  const int tci = (it.Cin + 31) / 32, tco = (it.Cout + 31) / 32;
  const int ci0 = (t % tci) * 32;
  t /= tci;
  const int co0 = (t % tco) * 32;
  const int tap = t / tco;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 256 threads: 32 x 8
  T* cast = reinterpret_cast<T*>(it.cast);
  T* dgrad = reinterpret_cast<T*>(it.dgrad);
  for (int r = ty; r < 32; r += 8) {
    const int co = co0 + r, ci = ci0 + tx;
    float v = 0.f;
    if (co < it.Cout && ci < it.Cin) {
      const long i = ((long)co * it.taps + tap) * it.Cin + ci;
      v = it.w[i];
      if (cast) cast[i] = (T)v;
    }
    tile[r][tx] = v;
  }
  if (!dgrad) return;  // (block-uniform)
  __syncthreads();
  const int ftap = it.taps - 1 - tap;
  for (int r = ty; r < 32; r += 8) {
    const int ci = ci0 + r, co = co0 + tx;
    if (ci < it.Cin && co < it.Cout) dgrad[((long)ci * it.taps + ftap) * it.Cout + co] = (T)tile[tx][r];
  }
}

extern "C" int rs_weight_prep_bf16(const rs_wprep_item* items_dev, int n, int total_tiles, rs_stream_t stream) {
  if (!items_dev || n <= 0 || total_tiles <= 0) return RS_EINVAL;
  weight_prep_kernel<bf16_t><<<total_tiles, 256, 0, (hipStream_t)stream>>>(items_dev, n);
  return RS_LAUNCH_RESULT();
}

// The fp32 twin (round 5): `dgrad` points at FLOAT buffers (the layout of rs_pack_dgrad_weight), `cast` is NULL -- the fp32
// training step's 59 per-convolution packing launches in one.
extern "C" int rs_weight_prep_f32(const rs_wprep_item* items_dev, int n, int total_tiles, rs_stream_t stream) {
  if (!items_dev || n <= 0 || total_tiles <= 0) return RS_EINVAL;
  weight_prep_kernel<float><<<total_tiles, 256, 0, (hipStream_t)stream>>>(items_dev, n);
  return RS_LAUNCH_RESULT();
}

const char* const kTileNamesBf16[NTILES] = {"conv_igemm_bf16<128x128>", "conv_igemm_bf16<128x64>", "conv_igemm_bf16<128x32>",
                                            "conv_igemm_bf16<64x64>", "", "conv_igemm_bf16<256x128>",
                                            "conv_igemm_bf16<256x256>", "conv_thin_bf16", "conv_halo_bf16", "conv1x1_ew_bf16<128x128>"};
const int kTileBM[NTILES] = {128, 128, 128, 64, 128, 256, 256, 128, 256, 128};
const int kTileBN[NTILES] = {128, 64, 32, 64, 64, 128, 256, 32, 128, 64};

// Dispatcher overrides: the process-global knob table of knobs.hip (rs_set_knob / rs_conv2d_set_tuning; seeded from the
// environment ONCE -- RS_CONV_TILE, RS_CONV_ROWB, RS_CONV_BIG, ... -- so that a whole benchmark run can be steered from
// outside), for the parity tests (which must reach every tile with small problems) and for A/B measurements.
struct Tuning {
  int tile, rowb, big, min256, halo, halo_min, halo512;
};
Tuning tuning() {
  const RsKnobs& k = rs_knobs();
  return {k.conv_tile, k.conv_rowb, k.conv_big, k.conv_min256, k.conv_halo, k.conv_halo_min, k.conv_halo512};
}

bool valid(const rs_conv_desc* d) {
  if (!d) return false;
  if (d->N <= 0 || d->Hs <= 0 || d->Ws <= 0 || d->Ho <= 0 || d->Wo <= 0) return false;
  if (d->kh <= 0 || d->kw <= 0 || d->kh > kMaxK || d->kw > kMaxK || d->stride <= 0 || d->pad < 0) return false;
  if (d->Cout <= 0 || (d->Cout % 32) != 0) return false;
  if (d->ups < 0 || d->ups > 2 || d->stem) return false;
  if (d->C1 <= 0 || (d->C1 % 32) != 0 || d->C2 < 0 || (d->C2 % 32) != 0) return false;
  return true;
}

bool phase_ok(const rs_conv_desc* d) {
  return d->ups == 1 && d->kh == 3 && d->kw == 3 && d->stride == 1 && d->pad == 1 && d->Ho == 2 * d->Hs && d->Wo == 2 * d->Ws;
}

int pick_tile(const rs_conv_desc* d, bool phase4 = false, int es = 4, bool stats = false) {
  const long M = phase4 ? (long)d->N * d->Hs * d->Ws * 4 : (long)d->N * d->Ho * d->Wo;  // (x4: the phases share the grid)
  const long want = 512;  // >= 2 blocks per CU
  const Tuning tu = tuning();
  const int ft = tu.tile;
  // a forced tile must be able to run the launch: N tiles whole (or the ragged 128-wide form below), 8 waves only in bf16
  // without fused statistics (the statistics' block reduction is laid out for 256 threads)
  if (ft >= 0 && ft < NTILES && ft != TSTEM_RESERVED && ft != TTHIN && ft != THALO && ft != TEW && (d->Cout % kTileBN[ft] == 0 || (ft == T128x128 && d->Cout > 128)) &&
      (ft != T256x256 || (es == 2 && !stats)))
    return ft;
  // 8-wave 256x256 tile (bf16, no fused statistics; one block per CU): half the LDS-DMA bytes per MFMA of the 128x128 tile,
  // which is what bounds that tile (~23 B/clk/CU of DMA = ~900 TFLOP/s at 64 flop/B).  Measured per layer (bs 32):
  // +21 % on 256->256 3x3 at 64^2 (512 blocks, 1071 vs 882 TFLOP/s) but -30 % with 128 blocks.
  if (tu.big && es == 2 && !stats) {
    const long mrows = phase4 ? (long)d->N * d->Hs * d->Ws : M;  // rows per launch grid slice (each phase tiles its own)
    const long nph = phase4 ? 4 : 1;
    const long nk128 = (long)(phase4 ? 4 : d->kh * d->kw) * (d->C1 + d->C2) * es / 128;
    // (short-K launches are DMA-latency / HBM bound: there the 256-row tile ties the 128x128 one at best; from 16 chunks on
    // it wins: +22 % on dec2's 4x4/s2 data gradient, 64 -> 768 at 64^2 x 32 -- profiles/r02/layer_sweep.txt)
    if (d->Cout % 256 == 0 && nk128 >= ((phase4 || d->kh > 1) ? 16 : 32) && rs_cdiv(mrows, 256) * (d->Cout / 256) * nph >= tu.min256) return T256x256;
  }
  // fp32, K <= 64 (two 128-byte chunks: layer1's 64 -> 256 conv3 / downsample): a block is all prologue + epilogue, the
  // launch is output-bandwidth bound, and twice as many half-width blocks overlap those phases better: +9 % measured
  // (profiles/r02/layer_sweep.txt, f32:conv+res:16,64,128,128,256).  bf16 measured no gain.
  if (es == 4 && !stats && !phase4 && (long)d->kh * d->kw * (d->C1 + d->C2) <= 64 && d->Cout % 64 == 0 && (long)rs_cdiv(M, 128) * (d->Cout / 64) >= want)
    return T128x64;
  if (d->Cout % 128 == 0 && (long)rs_cdiv(M, 128) * (d->Cout / 128) >= want) return T128x128;
  // ragged last N tile (weight rows past Cout read as zeros through the buffer bound, the epilogue skips their columns):
  // worth it when <= 1/5 of the MFMAs are padding -- Cout = 320, the data gradient of the 256 + 64 concat, runs the
  // 128x128 tile's 2x higher MFMA : LDS-read ratio instead of five 128x64 tiles.
  if (d->Cout > 128 && d->Cout % 128 != 0 && (long)rs_cdiv(d->Cout, 128) * 128 * 4 <= (long)d->Cout * 5 &&
      (long)rs_cdiv(M, 128) * rs_cdiv(d->Cout, 128) >= want)
    return T128x128;
  if (d->Cout % 64 == 0) {
    if ((long)rs_cdiv(M, 128) * (d->Cout / 64) >= want) return T128x64;
    return T64x64;
  }
  return T128x32;
}

// 128-byte rows (4 k-steps per barrier, 2 blocks per CU) or 64-byte rows (2 k-steps per barrier, 4 blocks per CU)?
// rs_conv2d_set_tuning / RS_CONV_ROWB=64|128 overrides.
int pick_rowb(const rs_conv_desc* d, int es, bool phase4 = false, bool stats = false) {
  const int forced = tuning().rowb;
  // Measured per layer with both row sizes forced, on both paths (bs-32 bf16 train, bs-16 fp32 predict; after the DMA
  // pieces moved between the MFMAs).  64-byte rows = half the LDS per block, twice the blocks per CU, a barrier every 2
  // k-steps: they win on short K loops (the 1x1 convolutions of the encoder: 25-37 % at nk128 <= 8, where a block is
  // mostly DMA round trip + epilogue and co-resident blocks are what overlaps them) and on mid-K layers with a large grid;
  // 128-byte rows win on long K (20-40 % on layer3/4's 3x3 and the decoder's phase / 4x4 forms in bf16).  fp32: the
  // 64-cycle MFMAs hide the extra barriers, 64-byte rows tie or win up to nk128 = 16 and whenever the grid is large.
  const int tile = pick_tile(d, phase4, es, stats);
  if (forced == 64 || forced == 128) return forced;
  const long blocks = (long)rs_cdiv((long)d->N * d->Ho * d->Wo, kTileBM[tile]) * rs_cdiv(d->Cout, kTileBN[tile]);
  const long nk128 = (long)(phase4 ? 4 : d->kh * d->kw) * (d->C1 + d->C2) * es / 128;
  if (es == 4) return (nk128 <= 16 || blocks >= 2048) ? 64 : 128;
  if (tile == T256x256) return nk128 <= 8 ? 64 : 128;
  if (kTileBN[tile] <= 64) return nk128 <= 4 ? 64 : 128;
  if (phase4 || d->kh == 4) return 128;
  if (nk128 <= 8 || (blocks >= 2048 && nk128 <= 24)) return 64;
  return 128;
}

void launch(int rowb, bool phase4, int epi, int tile, int grid, hipStream_t s, const ConvArgsT<float>& a) {
  if (phase4) rs_conv_launch_f32_phase_eval(tile, rowb, grid, s, a);
  else if (epi == EPI_STATS) rs_conv_launch_f32_plain_stats(tile, rowb, grid, s, a);
  else if (epi == EPI_BWD) rs_conv_launch_f32_plain_bwd(tile, rowb, grid, s, a);
  else rs_conv_launch_f32_plain_eval(tile, rowb, grid, s, a);
}
void launch(int rowb, bool phase4, int epi, int tile, int grid, hipStream_t s, const ConvArgsT<bf16_t>& a) {
  if (phase4) rs_conv_launch_bf16_phase_eval(tile, rowb, grid, s, a);
  else if (epi == EPI_STATS) rs_conv_launch_bf16_plain_stats(tile, rowb, grid, s, a);
  else if (epi == EPI_BWD) rs_conv_launch_bf16_plain_bwd(tile, rowb, grid, s, a);
  else rs_conv_launch_bf16_plain_eval(tile, rowb, grid, s, a);
}

// Which all-taps kernel of conv_thin_bf16.hip runs this bf16 launch (-1: none): the 32-channel decoder tail, plain epilogue
// (ReLU / ReLU mask only).  A forced igemm tile (rs_conv2d_set_tuning) keeps the generic kernel, for tests and A/B runs.
int thin_mode(const rs_conv_desc* d, bool phase4, bool plain_epilogue) {
  const int ft = tuning().tile;
  if (!plain_epilogue || (ft != -1 && ft != TTHIN) || d->C2 != 0) return -1;
  if (phase4) return (d->C1 == 128 && d->Cout == 32) ? 1 : -1;
  if (d->ups != 0) return -1;
  if (d->kh == 3 && d->kw == 3 && d->stride == 1 && d->pad == 1 && d->C1 == 32 && d->Cout == 32 && d->Ho == d->Hs && d->Wo == d->Ws) return 0;
  if (d->kh == 4 && d->kw == 4 && d->stride == 2 && d->pad == 1 && d->C1 == 32 && d->Cout == 128 && 2 * d->Ho == d->Hs && 2 * d->Wo == d->Ws)
    return 2;
  return -1;
}

// Which halo-once form of the kernel runs this bf16 launch (HALO_NONE: none), and its N tile.  The layer's geometry and "enough
// blocks to fill the chip" -- the latter counts the batch, and the halo forms accumulate K chunk-major where the implicit GEMM goes
// tap-major: a bf16 tile's output may differ by rounding with the batch it travels in (the fp32 path's choices never look at the
// batch and ARE batch-invariant bit for bit; tests/test_gpu_bf16.py::test_unet_bf16_predict_across_batch_sizes bounds the bf16 side): the rows of a block are an 8 x 32 patch of the grid they enumerate,
// so that grid must tile into such patches; K-chunks are 128-byte rows.  A forced implicit-GEMM tile
// (rs_conv2d_set_tuning) keeps the generic kernel; forcing THALO takes the halo form wherever it can run.
// `epi`: the launch's epilogue kind -- only the 3x3 form carries the statistics / into-BatchNorm epilogues (HALO_PHASE and
// HALO_DG4 are instantiated for EPI_EVAL alone: a launch with fused statistics keeps the implicit-GEMM kernel there).
int halo_mode(const rs_conv_desc* d, bool phase4, int csplit, int epi, int* bn_out, int* bm_out = nullptr) {
  const Tuning tu = tuning();
  if (!tu.halo || (tu.tile != -1 && tu.tile != THALO)) return HALO_NONE;
  // patch of 512 pixels (16 x 32) with 32-channel chunks, or of 256 (8 x 32) with 64-channel chunks
  bool big = tu.tile == THALO ? tu.rowb == 64 : tu.halo512 != 0;  // (unforced + auto: decided below, once the grid is known)
  const bool c64 = (d->C1 % 64) == 0 && (d->C2 % 64) == 0;  // 64-channel chunks (256-pixel patch) possible
  if ((d->C1 % 32) != 0 || (d->C2 % 32) != 0) return HALO_NONE;
  int bn;
  if (d->Cout % 128 == 0) bn = 128;
  else if (d->Cout > 128 && (long)rs_cdiv(d->Cout, 128) * 128 * 4 <= (long)d->Cout * 5) bn = 128;  // ragged last N tile (cf. pick_tile)
  else if (d->Cout % 64 == 0) bn = 64;
  else return HALO_NONE;
  if (csplit > 0 && (csplit % bn) != 0) return HALO_NONE;
  int mode, gh, gw;  // the grid the block rows enumerate
  if (phase4) {
    mode = HALO_PHASE, gh = d->Hs, gw = d->Ws;
  } else if (d->ups == 0 && d->kh == 3 && d->kw == 3 && d->stride == 1 && d->pad == 1 && d->Ho == d->Hs && d->Wo == d->Ws) {
    mode = HALO_33, gh = d->Ho, gw = d->Wo;
  } else if (d->ups == 0 && d->kh == 4 && d->kw == 4 && d->stride == 2 && d->pad == 1 && d->Hs == 2 * d->Ho && d->Ws == 2 * d->Wo && d->C2 == 0) {
    mode = HALO_DG4, gh = d->Ho, gw = d->Wo;
  } else {
    return HALO_NONE;
  }
  if (mode != HALO_33 && epi != EPI_EVAL) return HALO_NONE;
  // unforced: the 512-pixel patch (2/3 of the DMA bytes and 3/4 of the fragment reads per MFMA) where there is at least one
  // such patch per CU -- layer2's conv2 (-17 % against the implicit GEMM; the 256-pixel patch: -7 %), dec3 forward and data
  // gradient (-16 %; -10 %); with fewer patches (layer3, dec1: 64) the 256-pixel patch or the 8-wave tile win
  // (profiles/r04/halo_sweep_v3.txt)
  if (tu.tile != THALO && tu.halo512 < 0) big = (long)d->N * (gh / 16) * (gw / 32) >= 256;
  if (big && (bn != 128 || (gh % 16) != 0)) big = false;
  if (!big && !c64) return HALO_NONE;
  const int ph = big ? 16 : 8;
  if ((gh % ph) != 0 || (gw % 32) != 0) return HALO_NONE;
  const long cmax = d->C1 > d->C2 ? d->C1 : d->C2;
  if ((long)d->Hs * d->Ws * cmax * 2 >= (1L << 31)) return HALO_NONE;  // 32-bit DMA offsets within one image
  const long blocks = (long)d->N * (gh / ph) * (gw / 32) * rs_cdiv(d->Cout, bn) * (phase4 ? 4 : 1);
  if (tu.tile != THALO) {
    // Unforced: where the A/B of profiles/r04/halo_sweep_v2.txt (bs 32, every implicit-GEMM tile against the halo form on the
    // layers of the bf16 train step) says the halo form wins -- by the layer's GEOMETRY only:
    //   3x3        N tile of 128 (layer2 / layer3 conv2 and their data gradients: -6 .. -12 %; the 64-cout layer1 ties / loses)
    //   phase      Cout <= 128 (dec2, dec3: -10 .. -13 %); with Cout % 256 == 0 the 8-wave 256x256 tile is faster (dec1: +10 %)
    //   dgrad4x4   >= 2 channel chunks per plane (dec1, dec3: -8 .. -10 %; dec2's 64 -> 768 is 16 short steps: the 256x256 tile wins)
    if (blocks < tu.halo_min) return HALO_NONE;
    if (mode == HALO_33 && bn != 128) return HALO_NONE;
    if (mode == HALO_PHASE && d->Cout % 256 == 0) return HALO_NONE;
    if (mode == HALO_DG4 && d->C1 < 128) return HALO_NONE;
  }
  if (bn_out) *bn_out = bn;
  if (bm_out) *bm_out = ph * 32;
  return mode;
}

// fp32 1x1 launches that take conv1x1_ew_f32.hip (main loop and epilogue on separate waves of a persistent block; bit-identical
// to the generic kernel).  By rule (knob -1) where it was measured to win: K <= 64 -- layer1's HBM-bound launches, x 1.12-1.15
// (profiles/r04/ew_1x1.txt) -- geometry only, never the batch size.  A forced implicit-GEMM tile keeps the generic kernel.
bool ew_f32_mode(const rs_conv_desc* d, bool phase4, bool plain) {
  const int ew = rs_knobs().conv1x1_ew;
  return ew != 0 && tuning().tile < 0 && !phase4 && plain && rs_conv1x1_ew_f32_ok(d) && (ew > 0 || d->C1 <= 64);
}

// fp32 1x1 launches that take conv1x1_np_f32.hip (a sub-tile's epilogue issued between the next sub-tile's MFMAs by the same waves;
// bit-identical to the generic kernel; `make EXP=1` builds only: measured slower, profiles/r06/np_1x1.txt).  Knob -1 would pick launches
// WITH a residual and 128 <= K <= 512 (Bottleneck.conv3 of layer2 .. layer4), geometry only.
bool np_f32_mode(const rs_conv_desc* d, bool phase4, bool plain, bool has_res) {
  const int np = rs_knobs().conv1x1_np;
  return np != 0 && tuning().tile < 0 && !phase4 && plain && rs_conv1x1_np_f32_ok(d) && (np > 0 || (has_res && d->C1 >= 128 && d->C1 <= 512));
}

template <typename T>
int conv_fwd(const rs_conv_desc* d, const void* src1, const void* src2, const void* weight, const float* scale,
             const float* shift, const void* residual, const void* relu_mask, void* out, rs_stream_t stream,
             float* stats = nullptr, const void* bn_y = nullptr, const float* bn_mean = nullptr,
             const float* bn_invstd = nullptr, bool phase4 = false, void* out2 = nullptr, const void* mask2 = nullptr,
             int csplit = 0, const unsigned char* mask_bits = nullptr) {
  if (!valid(d) || !src1 || !weight || !out) return RS_EINVAL;
  if (mask_bits && (relu_mask || !bn_y || (d->Cout & 7))) return RS_EINVAL;  // (bits: data gradients into a BatchNorm only)
  if (out2 && (csplit <= 0 || csplit >= d->Cout || residual || stats)) return RS_EINVAL;
  if (phase4 && (!phase_ok(d) || stats)) return RS_EINVAL;
  if (d->C2 > 0 && !src2) return RS_EINVAL;
  constexpr long ES = (long)sizeof(T);
  if constexpr (sizeof(T) == 2) {
    const int tm = thin_mode(d, phase4, !scale && !shift && !residual && !stats && !out2);
    if (tm >= 0) {
      const int rc = rs_conv_thin_bf16_launch(tm, src1, weight, relu_mask, out, d->N, d->Hs, d->Ws, d->relu, stream);
      if (rc != RS_EINVAL) return rc;  // (RS_EINVAL: the launch does not fit its 32-bit offsets -> generic kernel)
    }
  }
  ConvArgsT<T> a;
  a.src1 = reinterpret_cast<const T*>(src1);
  a.src2 = reinterpret_cast<const T*>(src2);
  a.wgt = reinterpret_cast<const T*>(weight);
  a.scale = scale;
  a.shift = shift;
  a.res = reinterpret_cast<const T*>(residual);
  a.mask = reinterpret_cast<const T*>(relu_mask);
  a.out = reinterpret_cast<T*>(out);
  a.stats = stats;
  a.bn_y = reinterpret_cast<const T*>(bn_y);
  a.bn_mean = bn_mean;
  a.bn_invstd = bn_invstd;
  a.mask_bits = mask_bits;
  a.out2 = reinterpret_cast<T*>(out2);
  a.mask2 = reinterpret_cast<const T*>(mask2);
  a.csplit = csplit;
  a.N = d->N;
  a.Hs = d->Hs;
  a.Ws = d->Ws;
  a.C1 = d->C1;
  a.C2 = d->C2;
  a.phase4 = phase4 ? 1 : 0;
  a.ups = phase4 ? 0 : d->ups;  // phase mode gathers on the source grid itself
  a.Hv = a.ups == 0 ? d->Hs : (a.ups == 1 ? 2 * d->Hs : 2 * d->Hs - 1);
  a.Wv = a.ups == 0 ? d->Ws : (a.ups == 1 ? 2 * d->Ws : 2 * d->Ws - 1);
  a.kh = phase4 ? 2 : d->kh;
  a.kw = phase4 ? 2 : d->kw;
  a.stride = d->stride;
  a.pad = d->pad;
  a.Ho = d->Ho;
  a.Wo = d->Wo;
  a.Cout = d->Cout;
  if ((long)d->N * d->Ho * d->Wo >= (1L << 31)) return RS_EINVAL;
  const long M = phase4 ? (long)d->N * d->Hs * d->Ws : (long)d->N * d->Ho * d->Wo;  // rows per phase
  a.M = (int)M;
  {
    // 32-bit byte offsets relative to the first image of a tile (<= 512 rows: 512/(rows per image) + 2 images)
    const long cmax = d->C1 > d->C2 ? d->C1 : d->C2;
    const long img_bytes = (long)d->Hs * d->Ws * cmax * ES;
    const long rows_per_image = phase4 ? (long)d->Hs * d->Ws : (long)d->Ho * d->Wo;
    const long span = (512 / rows_per_image + 2) * img_bytes;
    if (span >= (1L << 31)) return RS_EINVAL;
    if ((long)d->Cout * d->kh * d->kw * (d->C1 + d->C2) * ES >= (1L << 31)) return RS_EINVAL;
  }
  // channels per K-chunk: a 128-byte row, or 64 bytes (the 32-channel bf16 layers; also half the LDS: 4 blocks per CU
  // instead of 2, which is what the short-K layers want -- see pick_rowb)
  const int kc128 = 128 / (int)ES;
  const bool can128 = d->C1 % kc128 == 0 && d->C2 % kc128 == 0;
  const int kc = (can128 && pick_rowb(d, (int)ES, phase4, stats != nullptr) == 128) ? kc128 : kc128 / 2;
  a.cpt = (d->C1 + d->C2) / kc;
  a.ntaps = a.kh * a.kw;
  a.nk = a.ntaps * a.cpt;
  a.Kw = a.nk * kc;
  a.relu = d->relu;
  a.tpx = a.tpi = 0;

  // epilogue kind: forward with fused BatchNorm statistics takes no other epilogue input; the data gradient into a
  // BatchNorm takes residual / mask but no scale / shift / ReLU
  int epi = EPI_EVAL;
  if (stats) {
    epi = bn_y ? EPI_BWD : EPI_STATS;
    if (scale || shift || d->relu || out2) return RS_EINVAL;
    if (epi == EPI_STATS && (residual || relu_mask)) return RS_EINVAL;
    if (epi == EPI_BWD && (!bn_mean || !bn_invstd)) return RS_EINVAL;
  }
  if constexpr (sizeof(T) == 4) {
    if (ew_f32_mode(d, phase4, epi == EPI_EVAL && !relu_mask && !out2 && !mask_bits)) return rs_conv1x1_ew_f32_launch(a, (hipStream_t)stream);
#ifdef RS_EXP_BUILD  // (`make EXP=1`: measurement only -- parity yes, speed no: profiles/r06/np_1x1.txt)
    if (np_f32_mode(d, phase4, epi == EPI_EVAL && !relu_mask && !out2 && !mask_bits, residual != nullptr)) return rs_conv1x1_np_f32_launch(a, (hipStream_t)stream);
#endif
  }
  if constexpr (sizeof(T) == 2) {
    int bn = 0, bm = 256;
    const int hm = halo_mode(d, phase4, out2 ? csplit : 0, epi, &bn, &bm);
    if (hm != HALO_NONE) {
      const int gh = hm == HALO_PHASE ? d->Hs : d->Ho, gw = hm == HALO_PHASE ? d->Ws : d->Wo;
      const int ctot = d->C1 + d->C2;
      a.cpt = ctot / (bm == 512 ? 32 : 64);
      a.nk = a.cpt * (hm == HALO_DG4 ? 4 : 1);                             // K-groups: parity planes x channel chunks
      a.Kw = (hm == HALO_PHASE ? 4 : (hm == HALO_33 ? 9 : 16)) * ctot;      // elements per weight row
      a.tpx = gw / 32;
      a.tpi = a.tpx * (gh / (bm / 32));
      a.ntiles = rs_cdiv(d->Cout, bn);
      const int grid = d->N * a.tpi * a.ntiles * (phase4 ? 4 : 1);
      hipStream_t hs = (hipStream_t)stream;
      if (bm == 512) bn |= 0x1000;
      if (hm == HALO_33) rs_conv_launch_bf16_halo33(bn, epi, grid, hs, a);
#ifdef RS_HALO_KO_BUILD  // (`make KO=1`: the knock-out instantiations of conv_halo_ko.hip, measurement only -- not in the default library)
      else if (hm == HALO_PHASE && bn == 128 && rs_knobs().halo_ko) rs_conv_launch_bf16_halo_phase_ko(rs_knobs().halo_ko, grid, hs, a);
#endif
      else if (hm == HALO_PHASE) rs_conv_launch_bf16_halo_phase(bn, 0, grid, hs, a);
      else rs_conv_launch_bf16_halo_dg4(bn, 0, grid, hs, a);
      return RS_LAUNCH_RESULT();
    }
  }

  const int tile = pick_tile(d, phase4, (int)ES, stats != nullptr);
#ifdef RS_EXP_BUILD  // (`make EXP=1`: measurement only -- parity yes, speed no: profiles/r05/ew_bf16_1x1.txt)
  if constexpr (sizeof(T) == 2) {
    // conv1x1_ew_bf16.hip (train-mode 1x1 forward, statistics epilogue on its own waves); its partial rows are per 128-pixel
    // tile, as the generic 128-row tiles'
    if (rs_knobs().conv1x1_ew_bf16 > 0 && tuning().tile < 0 && epi == EPI_STATS && !phase4 && kTileBM[tile] == 128 &&
        rs_conv1x1_ew_bf16_stats_ok(d))
      return rs_conv1x1_ew_bf16_stats_launch(a, (hipStream_t)stream);
  }
#endif
  if (out2 && (csplit % kTileBN[tile]) != 0) return RS_EINVAL;
  a.ntiles = rs_cdiv(d->Cout, kTileBN[tile]);  // the last N tile may be ragged (pick_tile)
  const int grid = rs_cdiv(M, kTileBM[tile]) * a.ntiles * (phase4 ? 4 : 1);
  hipStream_t s = (hipStream_t)stream;
  launch(kc == kc128 ? 128 : 64, phase4, epi, tile, grid, s, a);
  return RS_LAUNCH_RESULT();
}

}  // namespace

// fp32 entry for conv_igemm.hip (declared in common.h): every non-stem convolution of rs_conv2d_fwd
int rs_conv_dma_f32(const rs_conv_desc* d, const float* src1, const float* src2, const float* weight, const float* scale,
                    const float* shift, const float* residual, const float* relu_mask, float* out, void* stream) {
  return conv_fwd<float>(d, src1, src2, weight, scale, shift, residual, relu_mask, out, stream);
}

int rs_conv_dma_tile(const rs_conv_desc* d) { return valid(d) ? pick_tile(d) : RS_EINVAL; }

extern "C" int rs_conv2d_set_tuning(int tile, int rowb) {
  if (tile < -1 || tile >= NTILES || tile == TSTEM_RESERVED || tile == TEW || (rowb != 0 && rowb != 64 && rowb != 128)) return RS_EINVAL;
  rs_knobs().conv_tile = tile;
  rs_knobs().conv_rowb = rowb;
  return 0;
}

// For the roofline report (kernel names that map 1:1 to the launched symbol): tile index and K-chunk row bytes the
// dispatcher picks for `d` with activations of `es` bytes, in the direct (phase4 = 0) or phase form.
// `form`: bit 0 = phase form; bit 1 = the launch carries a fused epilogue beyond scale / shift / residual / ReLU (BatchNorm statistics,
// a ReLU mask, two destinations), i.e. never the plain-epilogue fp32 1x1 kernel -- so that a report can ask for the name of such a
// launch without flipping the process-global knob around the query.
extern "C" int rs_conv2d_config(const rs_conv_desc* d, int es, int form, int* tile, int* rowb) {
  if (form < 0 || form > 3) return RS_EINVAL;
  const int phase4 = form & 1;
  const bool plain = (form & 2) == 0;
  if (!valid(d) || (es != 2 && es != 4) || (phase4 && !phase_ok(d))) return RS_EINVAL;
  const int kc128 = 128 / es;
  const bool can128 = d->C1 % kc128 == 0 && d->C2 % kc128 == 0;
  if (es == 2 && thin_mode(d, phase4 != 0, true) >= 0) {  // (as for a launch with a plain epilogue)
    if (tile) *tile = TTHIN;
    if (rowb) *rowb = d->C1 * 2;
    return 0;
  }
  if (es == 2) {  // halo-once forms: reported as THALO with the N tile in `rowb` (the K-chunk rows are always 128 bytes)
    int bn = 0, bm = 256;
    if (halo_mode(d, phase4 != 0, 0, EPI_EVAL, &bn, &bm) != HALO_NONE) {
      if (tile) *tile = THALO;
      if (rowb) *rowb = bn | (bm == 512 ? 0x1000 : 0);  // (N tile; + 0x1000: the 512-pixel patch)
      return 0;
    }
  }
  if (es == 4 && ew_f32_mode(d, phase4 != 0, plain)) {
    if (tile) *tile = TEW;
    if (rowb) *rowb = 64;
    return 0;
  }
  if (tile) *tile = pick_tile(d, phase4 != 0, es);
  if (rowb) *rowb = (can128 && pick_rowb(d, es, phase4 != 0) == 128) ? 128 : 64;
  return 0;
}

// ---- phase form of the decoder convolutions ---------------------------------------------------------------------------
// fp32 KRSC [Cout][3][3][Cin] -> [4 phases (py, px)][Cout][2][2][Cin] in T: tap (r, s) of phase (py, px) is the sum of the
// original taps that land on the same source pixel: rows {0},{1,2} for py = 0 and {0,1},{2} for py = 1 (same in x).
template <typename T>
__global__ void pack_phase_weight_kernel(const float* __restrict__ w, T* __restrict__ out, int Cout, int Cin, long total) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  const int ci = (int)(i % Cin);
  long t = i / Cin;
  const int s = (int)(t & 1), r = (int)((t >> 1) & 1);
  t >>= 2;
  const int co = (int)(t % Cout);
  const int ph = (int)(t / Cout), py = ph >> 1, px = ph & 1;
  const int ky0 = py == 0 ? (r == 0 ? 0 : 1) : (r == 0 ? 0 : 2), ky1 = py == 0 ? (r == 0 ? 0 : 2) : (r == 0 ? 1 : 2);
  const int kx0 = px == 0 ? (s == 0 ? 0 : 1) : (s == 0 ? 0 : 2), kx1 = px == 0 ? (s == 0 ? 0 : 2) : (s == 0 ? 1 : 2);
  float acc = 0.f;
  for (int ky = ky0; ky <= ky1; ++ky)
    for (int kx = kx0; kx <= kx1; ++kx) acc += w[(((long)co * 3 + ky) * 3 + kx) * Cin + ci];
  out[i] = (T)acc;
}

// fp32 KRSC [Cout][3][3][Cin] of a 3x3 / STRIDE-2 / pad-1 convolution -> the phase pack [4][Cin][2][2][Cout] in T of its DATA gradient
// (round 6).  d in(y, x) = sum over the taps with (y + 1 - ky) even of w[.][ky][kx][.] dy((y + 1 - ky) / 2, (x + 1 - kx) / 2): on the
// input's parity (py, px) that is a 2x2 convolution over dy in exactly the phase form's geometry (source row a - (1 - py) + r), with
// taps ky(py, r) = none, 1 (py = 0) / 2, 0 (py = 1) -- nine taps per four pixels, zeros in the pack where a parity has no tap.  As a
// zero-insertion convolution (ups = 2) the same gradient runs all nine taps on every pixel: 36 per four.
template <typename T>
__global__ void pack_s2_dgrad_phase_weight_kernel(const float* __restrict__ w, T* __restrict__ out, int Cout, int Cin, long total) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;  // over [4][Cin][2][2][Cout]
  if (i >= total) return;
  const int cd = (int)(i % Cout);
  long t = i / Cout;
  const int s = (int)(t & 1), r = (int)((t >> 1) & 1);
  t >>= 2;
  const int cg = (int)(t % Cin);
  const int ph = (int)(t / Cin), py = ph >> 1, px = ph & 1;
  const int ky = py == 0 ? (r == 1 ? 1 : -1) : (r == 0 ? 2 : 0);
  const int kx = px == 0 ? (s == 1 ? 1 : -1) : (s == 0 ? 2 : 0);
  out[i] = (T)((ky < 0 || kx < 0) ? 0.f : w[(((long)cd * 3 + ky) * 3 + kx) * Cin + cg]);
}

// fp32 KRSC [Cout][3][3][Cin] -> data-gradient weights of the phase form, [Cin][4][4][Cout] in T: the gradient of
// DecoderBlock wrt its (pre-upsample) input is a 4x4 / stride-2 / pad-1 convolution over dz,
//   d_src[u][v][ci] = sum_{ty,tx,co} dz[2u - 1 + ty][2v - 1 + tx][co] * Wd[ci][ty][tx][co],
// Wd[ty] = sum of the taps ky with ((2u - 1 + ty) + ky - 1) >> 1 == u:  ty 0 -> {2}, 1 -> {1,2}, 2 -> {0,1}, 3 -> {0}.
template <typename T>
__global__ void pack_dgrad_phase_weight_kernel(const float* __restrict__ w, T* __restrict__ out, int Cout, int Cin, long total) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  const int co = (int)(i % Cout);
  long t = i / Cout;
  const int tx = (int)(t & 3), ty = (int)((t >> 2) & 3);
  const int ci = (int)(t >> 4);
  const int ky0 = ty == 0 ? 2 : (ty == 1 ? 1 : 0), ky1 = ty == 0 ? 2 : (ty == 1 ? 2 : (ty == 2 ? 1 : 0));
  const int kx0 = tx == 0 ? 2 : (tx == 1 ? 1 : 0), kx1 = tx == 0 ? 2 : (tx == 1 ? 2 : (tx == 2 ? 1 : 0));
  float acc = 0.f;
  for (int ky = ky0; ky <= ky1; ++ky)
    for (int kx = kx0; kx <= kx1; ++kx) acc += w[(((long)co * 3 + ky) * 3 + kx) * Cin + ci];
  out[i] = (T)acc;
}

// Same result from the ALREADY TRANSPOSED weights wt [Cin][3][3][Cout] with flipped taps (rs_pack_dgrad_weight): reads and
// writes are both contiguous along Cout, which the one-step kernel above cannot offer (its reads stride by 9*Cin).
template <typename T>
__global__ void combine_dgrad_phase_weight_kernel(const float* __restrict__ wt, T* __restrict__ out, int Cout, long total) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  const int co = (int)(i % Cout);
  long t = i / Cout;
  const int tx = (int)(t & 3), ty = (int)((t >> 2) & 3);
  const long ci = t >> 4;
  const int ky0 = ty == 0 ? 2 : (ty == 1 ? 1 : 0), ky1 = ty == 0 ? 2 : (ty == 1 ? 2 : (ty == 2 ? 1 : 0));
  const int kx0 = tx == 0 ? 2 : (tx == 1 ? 1 : 0), kx1 = tx == 0 ? 2 : (tx == 1 ? 2 : (tx == 2 ? 1 : 0));
  float acc = 0.f;
  for (int ky = ky0; ky <= ky1; ++ky)
    for (int kx = kx0; kx <= kx1; ++kx) acc += wt[(ci * 9 + (8 - (ky * 3 + kx))) * Cout + co];  // tap-flipped layout
  out[i] = (T)acc;
}

extern "C" int rs_combine_dgrad_phase_weight_dt(const float* w_dgrad, void* out, int dtype, int Cout, int Cin,
                                                rs_stream_t stream) {
  if (!w_dgrad || !out || Cout <= 0 || Cin <= 0) return RS_EINVAL;
  const long total = 16L * Cout * Cin;
  hipStream_t s = (hipStream_t)stream;
  if (dtype == RS_F32)
    combine_dgrad_phase_weight_kernel<float><<<rs_cdiv(total, 256), 256, 0, s>>>(w_dgrad, reinterpret_cast<float*>(out), Cout, total);
  else if (dtype == RS_BF16)
    combine_dgrad_phase_weight_kernel<bf16_t><<<rs_cdiv(total, 256), 256, 0, s>>>(w_dgrad, reinterpret_cast<bf16_t*>(out), Cout, total);
  else
    return RS_EINVAL;
  return RS_LAUNCH_RESULT();
}

extern "C" int rs_pack_dgrad_phase_weight_dt(const float* w_krsc, void* out, int dtype, int Cout, int Cin, rs_stream_t stream) {
  if (!w_krsc || !out || Cout <= 0 || Cin <= 0) return RS_EINVAL;
  const long total = 16L * Cout * Cin;
  hipStream_t s = (hipStream_t)stream;
  if (dtype == RS_F32)
    pack_dgrad_phase_weight_kernel<float><<<rs_cdiv(total, 256), 256, 0, s>>>(w_krsc, reinterpret_cast<float*>(out), Cout, Cin, total);
  else if (dtype == RS_BF16)
    pack_dgrad_phase_weight_kernel<bf16_t><<<rs_cdiv(total, 256), 256, 0, s>>>(w_krsc, reinterpret_cast<bf16_t*>(out), Cout, Cin, total);
  else
    return RS_EINVAL;
  return RS_LAUNCH_RESULT();
}

extern "C" int rs_pack_s2_dgrad_phase_weight_dt(const float* w_krsc, void* out, int dtype, int Cout, int Cin, rs_stream_t stream) {
  if (!w_krsc || !out || Cout <= 0 || Cin <= 0) return RS_EINVAL;
  const long total = 16L * Cout * Cin;
  hipStream_t s = (hipStream_t)stream;
  if (dtype == RS_F32)
    pack_s2_dgrad_phase_weight_kernel<float><<<rs_cdiv(total, 256), 256, 0, s>>>(w_krsc, reinterpret_cast<float*>(out), Cout, Cin, total);
  else if (dtype == RS_BF16)
    pack_s2_dgrad_phase_weight_kernel<bf16_t><<<rs_cdiv(total, 256), 256, 0, s>>>(w_krsc, reinterpret_cast<bf16_t*>(out), Cout, Cin, total);
  else
    return RS_EINVAL;
  return RS_LAUNCH_RESULT();
}

extern "C" int rs_pack_phase_weight_dt(const float* w_krsc, void* out, int dtype, int Cout, int Cin, rs_stream_t stream) {
  if (!w_krsc || !out || Cout <= 0 || Cin <= 0) return RS_EINVAL;
  const long total = 16L * Cout * Cin;
  hipStream_t s = (hipStream_t)stream;
  if (dtype == RS_F32)
    pack_phase_weight_kernel<float><<<rs_cdiv(total, 256), 256, 0, s>>>(w_krsc, reinterpret_cast<float*>(out), Cout, Cin, total);
  else if (dtype == RS_BF16)
    pack_phase_weight_kernel<bf16_t><<<rs_cdiv(total, 256), 256, 0, s>>>(w_krsc, reinterpret_cast<bf16_t*>(out), Cout, Cin, total);
  else
    return RS_EINVAL;
  return RS_LAUNCH_RESULT();
}

// rs_conv2d_fwd with the torch.cat split of a decoder data gradient fused into the store: output channels [0, csplit) ->
// out1 [N][Ho][Wo][csplit] (zeroed where mask1 <= 0 if given), [csplit, Cout) -> out2 [N][Ho][Wo][Cout-csplit] (mask2).
extern "C" int rs_conv2d_fwd_split_dt(const rs_conv_desc* d, int dtype, const void* src1, const void* weight, void* out1,
                                      const void* mask1, void* out2, const void* mask2, int csplit, rs_stream_t stream) {
  if (!out2 || (d && d->C2 != 0)) return RS_EINVAL;
  if (dtype == RS_F32)
    return conv_fwd<float>(d, src1, nullptr, weight, nullptr, nullptr, nullptr, mask1, out1, stream, nullptr, nullptr, nullptr,
                           nullptr, false, out2, mask2, csplit);
  if (dtype == RS_BF16)
    return conv_fwd<bf16_t>(d, src1, nullptr, weight, nullptr, nullptr, nullptr, mask1, out1, stream, nullptr, nullptr,
                            nullptr, nullptr, false, out2, mask2, csplit);
  return RS_EINVAL;
}

extern "C" int rs_conv2d_fwd_phase_dt(const rs_conv_desc* d, int dtype, const void* src1, const void* src2,
                                      const void* weight_phase, const float* scale, const float* shift,
                                      const void* residual, const void* relu_mask, void* out, rs_stream_t stream) {
  if (dtype == RS_F32)
    return conv_fwd<float>(d, src1, src2, weight_phase, scale, shift, residual, relu_mask, out, stream, nullptr, nullptr,
                           nullptr, nullptr, true);
  if (dtype == RS_BF16)
    return conv_fwd<bf16_t>(d, src1, src2, weight_phase, scale, shift, residual, relu_mask, out, stream, nullptr, nullptr,
                            nullptr, nullptr, true);
  return RS_EINVAL;
}

extern "C" long rs_conv2d_bnstats_rows(const rs_conv_desc* d) {
  if (!valid(d)) return RS_EINVAL;
  // (with fused statistics the implicit-GEMM tile choice does not depend on the element size: every es-specific rule of
  // pick_tile is switched off by `stats`; the bf16 halo forms are asked for through rs_conv2d_bnstats_rows_dt)
  return rs_cdiv((long)d->N * d->Ho * d->Wo, kTileBM[pick_tile(d, false, 4, true)]);
}

extern "C" long rs_conv2d_bnstats_rows_dt(const rs_conv_desc* d, int dtype) {
  if (!valid(d) || (dtype != RS_F32 && dtype != RS_BF16)) return RS_EINVAL;
  int bn = 0, bm = 256;
  if (dtype == RS_BF16 && halo_mode(d, false, 0, EPI_STATS, &bn, &bm) != HALO_NONE) return (long)d->N * (d->Ho / (bm / 32)) * (d->Wo / 32);  // one row per patch
  return rs_conv2d_bnstats_rows(d);
}

extern "C" int rs_conv2d_dgrad_bnstats_dt(const rs_conv_desc* d, int dtype, const void* dy, const void* weight,
                                          const void* residual, const void* relu_mask, const void* bn_y,
                                          const float* bn_mean, const float* bn_invstd, void* out, float* stats_partial,
                                          rs_stream_t stream) {
  if (!stats_partial || !bn_y || !bn_mean || !bn_invstd || (d && d->C2 != 0)) return RS_EINVAL;
  if (dtype == RS_F32)
    return conv_fwd<float>(d, dy, nullptr, weight, nullptr, nullptr, residual, relu_mask, out, stream, stats_partial, bn_y,
                           bn_mean, bn_invstd);
  if (dtype == RS_BF16)
    return conv_fwd<bf16_t>(d, dy, nullptr, weight, nullptr, nullptr, residual, relu_mask, out, stream, stats_partial, bn_y,
                            bn_mean, bn_invstd);
  return RS_EINVAL;
}

extern "C" int rs_conv2d_dgrad_bnstats_bits_dt(const rs_conv_desc* d, int dtype, const void* dy, const void* weight,
                                               const void* residual, const unsigned char* relu_mask_bits, const void* bn_y,
                                               const float* bn_mean, const float* bn_invstd, void* out, float* stats_partial,
                                               rs_stream_t stream) {
  if (!stats_partial || !bn_y || !bn_mean || !bn_invstd || !relu_mask_bits || (d && d->C2 != 0)) return RS_EINVAL;
  if (dtype == RS_F32)
    return conv_fwd<float>(d, dy, nullptr, weight, nullptr, nullptr, residual, nullptr, out, stream, stats_partial, bn_y, bn_mean,
                           bn_invstd, false, nullptr, nullptr, 0, relu_mask_bits);
  if (dtype == RS_BF16)
    return conv_fwd<bf16_t>(d, dy, nullptr, weight, nullptr, nullptr, residual, nullptr, out, stream, stats_partial, bn_y, bn_mean,
                            bn_invstd, false, nullptr, nullptr, 0, relu_mask_bits);
  return RS_EINVAL;
}

extern "C" int rs_conv2d_fwd_bnstats_dt(const rs_conv_desc* d, int dtype, const void* src1, const void* src2,
                                        const void* weight, void* out, float* stats_partial, rs_stream_t stream) {
  if (!stats_partial) return RS_EINVAL;
  if (dtype == RS_F32) return conv_fwd<float>(d, src1, src2, weight, nullptr, nullptr, nullptr, nullptr, out, stream, stats_partial);
  if (dtype == RS_BF16) return conv_fwd<bf16_t>(d, src1, src2, weight, nullptr, nullptr, nullptr, nullptr, out, stream, stats_partial);
  return RS_EINVAL;
}

extern "C" int rs_conv2d_tile_bf16(const rs_conv_desc* d) { return valid(d) ? pick_tile(d, false, 2) : RS_EINVAL; }

extern "C" const char* rs_conv2d_tile_name_bf16(int tile) {
  return (tile >= 0 && tile < NTILES) ? kTileNamesBf16[tile] : "";
}

extern "C" int rs_conv2d_fwd_bf16(const rs_conv_desc* d, const rs_bf16* src1, const rs_bf16* src2, const rs_bf16* weight,
                                  const float* scale, const float* shift, const rs_bf16* residual, const rs_bf16* relu_mask,
                                  rs_bf16* out, rs_stream_t stream) {
  return conv_fwd<bf16_t>(d, src1, src2, weight, scale, shift, residual, relu_mask, out, stream);
}

extern "C" int rs_pack_dgrad_weight_bf16(const float* w_krsc, rs_bf16* out, int Cout, int kh, int kw, int Cin,
                                         rs_stream_t stream) {
  if (!w_krsc || !out || Cout <= 0 || kh <= 0 || kw <= 0 || Cin <= 0) return RS_EINVAL;
  dim3 grid(rs_cdiv(Cin, 32), rs_cdiv(Cout, 32), kh * kw);
  pack_dgrad_weight_bf16_kernel<<<grid, 256, 0, (hipStream_t)stream>>>(w_krsc, reinterpret_cast<bf16_t*>(out), Cout, kh * kw,
                                                                       Cin);
  return RS_LAUNCH_RESULT();
}

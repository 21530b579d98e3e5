// knobs.hip -- every measurement / A-B switch of the library in ONE process-global table, read from the environment ONCE
// (at the first use) and changeable afterwards only through rs_set_knob().  A dispatcher never calls getenv(): a launch and
// its workspace query see the same settings, and nothing is parsed per launch.
//
// No reference counterpart (the closest is torch.backends.cudnn.benchmark / .deterministic): none of these change results
// beyond what the parity tests allow; unset = the measured rules.
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "common.h"

namespace {

struct Entry {
  const char* name;  // rs_set_knob / rs_get_knob name
  const char* env;   // environment variable that seeds it
  int RsKnobs::*field;
  int lo, hi;        // accepted range (a value outside it used to be taken silently -- conv_rowb = 32, wgrad_ring = 9, ...)
};

bool in_range(const Entry& e, int v) {
  if (v < e.lo || v > e.hi) return false;
  if (e.field == &RsKnobs::conv_rowb) return v == 0 || v == 64 || v == 128;  // (the K-chunk row is 64 or 128 bytes: rs_conv2d_set_tuning's rule)
  return true;
}

const Entry kEntries[] = {
    {"conv_tile", "RS_CONV_TILE", &RsKnobs::conv_tile, -1, 9},
    {"conv_rowb", "RS_CONV_ROWB", &RsKnobs::conv_rowb, 0, 128},
    {"conv_big", "RS_CONV_BIG", &RsKnobs::conv_big, 0, 1},
    {"conv_min256", "RS_CONV_MIN256", &RsKnobs::conv_min256, 0, 1048576},
    {"conv_halo", "RS_CONV_HALO", &RsKnobs::conv_halo, 0, 1},
    {"conv_halo_min", "RS_CONV_HALO_MIN", &RsKnobs::conv_halo_min, 0, 1048576},
    {"conv_halo512", "RS_CONV_HALO512", &RsKnobs::conv_halo512, -1, 1},
    {"conv1x1_ew", "RS_CONV1X1_EW", &RsKnobs::conv1x1_ew, -1, 1},
    {"conv1x1_np", "RS_CONV1X1_NP", &RsKnobs::conv1x1_np, -1, 1},
    {"conv1x1_ew_bf16", "RS_CONV1X1_EW_BF16", &RsKnobs::conv1x1_ew_bf16, 0, 1},
    {"halo_ko", "RS_HALO_KO", &RsKnobs::halo_ko, 0, 4},
    {"wgrad_f32_phase", "RS_WGRAD_F32_PHASE", &RsKnobs::wgrad_f32_phase, 0, 1},
    {"wgrad_f32_dma", "RS_WGRAD_F32_DMA", &RsKnobs::wgrad_f32_dma, -1, 1},
    {"wgrad_f32_blocks", "RS_WGRAD_F32_BLOCKS", &RsKnobs::wgrad_f32_blocks, 1, 1048576},
    {"wgrad_f32_wino", "RS_WGRAD_F32_WINO", &RsKnobs::wgrad_f32_wino, 0, 1},
    {"wgrad_f32_wino_blocks", "RS_WGRAD_F32_WINO_BLOCKS", &RsKnobs::wgrad_f32_wino_blocks, 1, 1048576},
    {"wgrad_f32_wino33", "RS_WGRAD_F32_WINO33", &RsKnobs::wgrad_f32_wino33, 0, 1},
    {"wgrad_f32_wino33_blocks", "RS_WGRAD_F32_WINO33_BLOCKS", &RsKnobs::wgrad_f32_wino33_blocks, 1, 1048576},
    {"wgrad_blocks", "RS_WGRAD_BLOCKS", &RsKnobs::wgrad_blocks, 1, 1048576},
    {"wgrad_blocks_phase", "RS_WGRAD_BLOCKS_PHASE", &RsKnobs::wgrad_blocks_phase, 1, 1048576},
    {"wgrad_phase4", "RS_WGRAD_PHASE4", &RsKnobs::wgrad_phase4, 0, 1},
    {"wgrad_blocks_phase4", "RS_WGRAD_BLOCKS_PHASE4", &RsKnobs::wgrad_blocks_phase4, 1, 1048576},
    {"wgrad_ring", "RS_WGRAD_RING", &RsKnobs::wgrad_ring, 2, 7},
    {"lovasz_xcd", "RS_LOVASZ_XCD", &RsKnobs::lovasz_xcd, 0, 1},
    {"wino_wide", "ROBOSAT_WINO_WIDE", &RsKnobs::wino_wide, 0, 1},
};

}  // namespace

RsKnobs& rs_knobs() {
  static RsKnobs k = [] {
    RsKnobs v;
    for (const Entry& e : kEntries)
      if (const char* s = getenv(e.env)) {
        const int x = atoi(s);
        if (in_range(e, x)) v.*(e.field) = x;
        else fprintf(stderr, "robosat_hip: %s=%s is outside [%d, %d]: ignored, the rule stays\n", e.env, s, e.lo, e.hi);
      }
    return v;
  }();
  return k;
}

extern "C" int rs_set_knob(const char* name, int value) {
  if (!name) return RS_EINVAL;
  for (const Entry& e : kEntries)
    if (strcmp(name, e.name) == 0) {
      if (!in_range(e, value)) return RS_EINVAL;
      rs_knobs().*(e.field) = value;
      return 0;
    }
  return RS_EINVAL;
}

extern "C" int rs_get_knob(const char* name, int* value) {
  if (!name || !value) return RS_EINVAL;
  for (const Entry& e : kEntries)
    if (strcmp(name, e.name) == 0) {
      *value = rs_knobs().*(e.field);
      return 0;
    }
  return RS_EINVAL;
}

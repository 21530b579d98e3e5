// knobs.hip -- every measurement / A-B switch of the library in ONE process-global table, read from the environment ONCE
// (at the first use) and changeable afterwards only through rs_set_knob().  A dispatcher never calls getenv(): a launch and
// its workspace query see the same settings, and nothing is parsed per launch.
//
// No reference counterpart (the closest is torch.backends.cudnn.benchmark / .deterministic): none of these change results
// beyond what the parity tests allow; unset = the measured rules.
#include <stdlib.h>
#include <string.h>

#include "common.h"

namespace {

struct Entry {
  const char* name;  // rs_set_knob / rs_get_knob name
  const char* env;   // environment variable that seeds it
  int RsKnobs::*field;
};

const Entry kEntries[] = {
    {"conv_tile", "RS_CONV_TILE", &RsKnobs::conv_tile},
    {"conv_rowb", "RS_CONV_ROWB", &RsKnobs::conv_rowb},
    {"conv_big", "RS_CONV_BIG", &RsKnobs::conv_big},
    {"conv_min256", "RS_CONV_MIN256", &RsKnobs::conv_min256},
    {"conv_halo", "RS_CONV_HALO", &RsKnobs::conv_halo},
    {"conv_halo_min", "RS_CONV_HALO_MIN", &RsKnobs::conv_halo_min},
    {"conv_halo512", "RS_CONV_HALO512", &RsKnobs::conv_halo512},
    {"conv1x1_ew", "RS_CONV1X1_EW", &RsKnobs::conv1x1_ew},
    {"conv1x1_ew_bf16", "RS_CONV1X1_EW_BF16", &RsKnobs::conv1x1_ew_bf16},
    {"halo_ko", "RS_HALO_KO", &RsKnobs::halo_ko},
    {"wgrad_f32_phase", "RS_WGRAD_F32_PHASE", &RsKnobs::wgrad_f32_phase},
    {"wgrad_f32_dma", "RS_WGRAD_F32_DMA", &RsKnobs::wgrad_f32_dma},
    {"wgrad_f32_blocks", "RS_WGRAD_F32_BLOCKS", &RsKnobs::wgrad_f32_blocks},
    {"wgrad_blocks", "RS_WGRAD_BLOCKS", &RsKnobs::wgrad_blocks},
    {"wgrad_blocks_phase", "RS_WGRAD_BLOCKS_PHASE", &RsKnobs::wgrad_blocks_phase},
    {"wgrad_phase4", "RS_WGRAD_PHASE4", &RsKnobs::wgrad_phase4},
    {"wgrad_blocks_phase4", "RS_WGRAD_BLOCKS_PHASE4", &RsKnobs::wgrad_blocks_phase4},
    {"wgrad_ring", "RS_WGRAD_RING", &RsKnobs::wgrad_ring},
    {"lovasz_xcd", "RS_LOVASZ_XCD", &RsKnobs::lovasz_xcd},
    {"wino_wide", "ROBOSAT_WINO_WIDE", &RsKnobs::wino_wide},
};

}  // namespace

RsKnobs& rs_knobs() {
  static RsKnobs k = [] {
    RsKnobs v;
    for (const Entry& e : kEntries)
      if (const char* s = getenv(e.env)) v.*(e.field) = atoi(s);
    return v;
  }();
  return k;
}

extern "C" int rs_set_knob(const char* name, int value) {
  if (!name) return RS_EINVAL;
  for (const Entry& e : kEntries)
    if (strcmp(name, e.name) == 0) {
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

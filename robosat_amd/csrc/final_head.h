// What `self.final`'s kernels do with a pixel's class logits, for kernels that hold the number of classes at run time (the
// fused dec5 + final head of conv_wino33_f32.hip): optional softmax, then fp32 NCHW logits / probabilities, or the quantised
// probability bytes of `rs predict` (reference tools/predict.py:87,96-103), or the argmax byte of `rs serve`
// (tools/serve.py:160-164).  The same operations in the same order as elementwise.hip:final_epilogue<C> on the classes
// c < C (the entries c >= C of `acc` are ignored), so equal logits give equal bytes.
#pragma once
#include "common.h"

constexpr int kHeadMaxC = 8;

__device__ __forceinline__ void rs_final_epilogue_rt(float (&acc)[kHeadMaxC], int C, long pix, long HW, int softmax,
                                                     const double* __restrict__ anchors, uint8_t* __restrict__ qout,
                                                     float* __restrict__ out, int Wimg, int ov) {
  if (softmax == 1 || softmax == 2) {
    float mx = acc[0];
#pragma unroll
    for (int c = 1; c < kHeadMaxC; ++c)
      if (c < C) mx = fmaxf(mx, acc[c]);
    float sum = 0.f;
#pragma unroll
    for (int c = 0; c < kHeadMaxC; ++c)
      if (c < C) {
        acc[c] = expf(acc[c] - mx);
        sum += acc[c];
      }
#pragma unroll
    for (int c = 0; c < kHeadMaxC; ++c)
      if (c < C) acc[c] = acc[c] / sum;
  }
  const long n = pix / HW, hw = pix - n * HW;
  if (softmax == 2) {  // un-buffer crop + np.digitize of every non-background class: C - 1 bytes per pixel
    if (C < 2) return;
    const int Himg = (int)(HW / Wimg);
    const int yy = (int)(hw / Wimg), xx = (int)(hw - (long)yy * Wimg);
    const int S_h = Himg - 2 * ov, S_w = Wimg - 2 * ov;
    if (yy < ov || yy >= Himg - ov || xx < ov || xx >= Wimg - ov) return;
    uint8_t* qo = qout + ((n * S_h + (yy - ov)) * (long)S_w + (xx - ov)) * (C - 1);
#pragma unroll
    for (int c = 1; c < kHeadMaxC; ++c)
      if (c < C) {
        const double pf = (double)acc[c];
        int q = (int)(pf * 255.0);  // anchors[i] ~ i/255: first guess, then settle on the exact table (anchors ascending)
        q = q < 0 ? 0 : (q > 255 ? 255 : q);
        while (q < 255 && anchors[q + 1] <= pf) ++q;
        while (q >= 0 && anchors[q] > pf) --q;
        qo[c - 1] = (uint8_t)((q + 1) & 0xff);  // bins are 1-based; 256 wraps to 0
      }
    return;
  }
  if (softmax == 3) {  // class index of the first maximum logit (np.argmax over axis 0), one byte per pixel
    int best = 0;
    float bv = acc[0];
#pragma unroll
    for (int c = 1; c < kHeadMaxC; ++c)
      if (c < C && acc[c] > bv) {
        bv = acc[c];
        best = c;
      }
    qout[pix] = (uint8_t)best;
    return;
  }
  float* o = out + n * C * HW + hw;
#pragma unroll
  for (int c = 0; c < kHeadMaxC; ++c)
    if (c < C) o[c * HW] = acc[c];
}

/*
 * orc_mip.c — CPU restatement of matrix-based intra prediction, prediction only: the search integration
 * (MIP candidates in the SATD and RD stages, mip_flag / mip_pred_mode syntax) is not built yet.  TEST INFRASTRUCTURE ONLY.
 *
 * Follows CL/MatrixIntraPrediction.cpp in its JVET_O0925 form: prepareInputForPred 71-126 (boundary averaging to 2 or 4 samples per
 * side, rebasing on the first sample), predBlock 236-263, initPredBlockParams 285-318, computeReducedPred 641-742 (uint8 weights, one
 * offset and shift per matrix, clipping), predictionUpsampling1D / predictionUpsampling 398-560 (linear interpolation from the reduced
 * prediction and the block's reference samples, shorter side first), getNumModesMip (CL/UnitTools.cpp:4688-4709).
 * Pinned against the reference's MatrixIntraPrediction (tests/golden/mip.npz).
 */
#include <string.h>
#include "orc_internal.h"
#include "orc_mip_tables.h"

int orc_mip_num_modes(int w, int h)
{
  if (w > 4 * h || h > 4 * w) return 0;
  if (w == 4 && h == 4) return 35;
  if (w <= 8 && h <= 8) return 19;
  return 11;
}
static int ilog2m(int v) { int r = 0; while (v > 1) { v >>= 1; r++; } return r; }
static void downsample(int *dst, const int *src, int srcLen, int dstLen)
{
  if (dstLen >= srcLen) { for (int i = 0; i < dstLen; i++) dst[i] = src[i]; return; }
  const int f = srcLen / dstLen, lf = ilog2m(f), rnd = 1 << (lf - 1);
  for (int d = 0, s = 0; d < dstLen; d++) { int sum = 0; for (int k = 0; k < f; k++, s++) sum += src[s]; dst[d] = (sum + rnd) >> lf; }
}
/* predictionUpsampling1D 398-446 */
static void upsample_1d(int *dst, const int *src, const int *bndry, int srcSizeUpsmpDim, int srcSizeOrthDim, int srcStep, int srcStride,
                        int dstStep, int dstStride, int bndryStep, int factor)
{
  const int lf = ilog2m(factor), rnd = 1 << (lf - 1);
  const int *srcLine = src; int *dstLine = dst; const int *bndryLine = bndry + bndryStep - 1;
  for (int o = 0; o < srcSizeOrthDim; o++) {
    const int *before = bndryLine, *behind = srcLine; int *cur = dstLine;
    for (int u = 0; u < srcSizeUpsmpDim; u++) {
      int sb = (*before) << lf, sh = 0;
      for (int pos = 1; pos <= factor; pos++) { sb -= *before; sh += *behind; *cur = (sb + sh + rnd) >> lf; cur += dstStep; }
      before = behind; behind += srcStep;
    }
    srcLine += srcStride; dstLine += dstStride; bndryLine += bndryStep;
  }
}
/* top[w], left[h]: the block's unfiltered reference samples (line 0); mode 0 .. orc_mip_num_modes - 1; pred: w*h, stride w */
void orc_pred_mip(const int16_t *top, const int16_t *left, int w, int h, int mode, int bit_depth, int16_t *pred)
{
  const int numModes = orc_mip_num_modes(w, h);
  const int rb = (w > 4 || h > 4) ? 4 : 2;                         /* reduced boundary per side */
  const int small = w <= 8 && h <= 8;
  const int rpw = small ? 4 : (w < 8 ? w : 8), rph = small ? 4 : (h < 8 ? h : 8);      /* reduced prediction */
  const int upH = w / rpw, upV = h / rph;
  int refT[64], refL[64], red[8], redT[8];
  for (int i = 0; i < w; i++) refT[i] = top[i];
  for (int i = 0; i < h; i++) refL[i] = left[i];
  downsample(red, refT, w, rb); downsample(red + rb, refL, h, rb);
  for (int i = 0; i < rb; i++) { redT[i] = red[rb + i]; redT[rb + i] = red[i]; }
  const int inSize = 2 * rb;
  const int off = red[0], offT = redT[0];
  red[0] = small ? off - (1 << (bit_depth - 1)) : 0; redT[0] = small ? offT - (1 << (bit_depth - 1)) : 0;
  for (int i = 1; i < inSize; i++) { red[i] -= off; redT[i] -= offT; }

  const int transpose = mode > numModes / 2;
  const int idx = transpose ? mode - numModes / 2 : mode;
  const uint8_t *matrix; int shiftM, offsetM;
  if (w == 4 && h == 4) { matrix = ORC_MIP_MATRIX_4x4 + idx * ORC_MIP_ROWS_4x4 * ORC_MIP_COLS_4x4; shiftM = ORC_MIP_SHIFT_4x4[idx]; offsetM = ORC_MIP_OFFSET_4x4[idx]; }
  else if (small) { matrix = ORC_MIP_MATRIX_8x8 + idx * ORC_MIP_ROWS_8x8 * ORC_MIP_COLS_8x8; shiftM = ORC_MIP_SHIFT_8x8[idx]; offsetM = ORC_MIP_OFFSET_8x8[idx]; }
  else { matrix = ORC_MIP_MATRIX_16x16 + idx * ORC_MIP_ROWS_16x16 * ORC_MIP_COLS_16x16; shiftM = ORC_MIP_SHIFT_16x16[idx]; offsetM = ORC_MIP_OFFSET_16x16[idx]; }
  int leaveHor = w == 4 && h >= 16, leaveVer = h == 4 && w >= 16;
  if (transpose) { const int t = leaveHor; leaveHor = leaveVer; leaveVer = t; }
  const int needUp = upH > 1 || upV > 1;
  const int *in = transpose ? redT : red;
  const int inOff = transpose ? offT : off;
  /* computeReducedPred 641-742 */
  int redPred[64], resT[64];
  int *res = (transpose && !needUp) ? resT : redPred;
  {
    int sum = 0; for (int i = 0; i < inSize; i++) sum += in[i];
    const int offset = (1 << (shiftM - 1)) - offsetM * sum;
    const int iw = transpose ? rph : rpw, ih = transpose ? rpw : rph;
    const int xStep = leaveHor ? 2 : 1, yStep = leaveVer ? iw : 0;
    const int redSize = small ? 0 : 1, mx = (1 << bit_depth) - 1;
    const uint8_t *wgt = matrix;
    if (redSize) wgt += xStep - 1;
    int pos = 0;
    for (int y = 0; y < ih; y++) {
      for (int x = 0; x < iw; x++) {
        if (redSize) wgt -= xStep;
        int acc = redSize ? 0 : in[0] * wgt[0];
        acc += in[1] * wgt[1] + in[2] * wgt[2] + in[3] * wgt[3];
        for (int i = 4; i < inSize; i++) acc += in[i] * wgt[i];
        int v = ((acc + offset) >> shiftM) + inOff;
        res[pos++] = v < 0 ? 0 : v > mx ? mx : v;
        wgt += xStep * inSize;
      }
      wgt += yStep * (inSize - redSize);
    }
    if (transpose && !needUp) for (int y = 0; y < rph; y++) for (int x = 0; x < rpw; x++) redPred[y * rpw + x] = resT[x * rph + y];
  }
  int full[64 * 64];
  if (!needUp) { for (int i = 0; i < w * h; i++) pred[i] = (int16_t) redPred[i]; return; }
  /* predictionUpsampling 448-560: shorter side first */
  if (h > w) {
    const int *verSrc; int verSrcStep, verSrcStride;
    if (upH > 1) {
      int *horDst = full + (upV - 1) * w;
      upsample_1d(horDst, redPred, refL, rpw, rph, transpose ? rph : 1, transpose ? 1 : rpw, 1, upV * w, upV, upH);
      verSrc = horDst; verSrcStep = upV * w; verSrcStride = 1;
    } else { verSrc = redPred; verSrcStep = transpose ? 1 : w; verSrcStride = transpose ? rph : 1; }
    upsample_1d(full, verSrc, refT, rph, w, verSrcStep, verSrcStride, w, 1, 1, upV);
  } else {
    const int *horSrc; int horSrcStep, horSrcStride;
    if (upV > 1) {
      int *verDst = full + (upH - 1);
      upsample_1d(verDst, redPred, refT, rph, rpw, transpose ? 1 : rpw, transpose ? rph : 1, w, upH, upH, upV);
      horSrc = verDst; horSrcStep = upH; horSrcStride = w;
    } else { horSrc = redPred; horSrcStep = transpose ? h : 1; horSrcStride = transpose ? 1 : rpw; }
    upsample_1d(full, horSrc, refL, rpw, h, horSrcStep, horSrcStride, 1, w, 1, upH);
  }
  for (int i = 0; i < w * h; i++) pred[i] = (int16_t) full[i];
}

/*
 * orc_leaf.c — leaf operators of the oracle (TEST INFRASTRUCTURE ONLY, see vvc_oracle.h).
 * Plain scalar C restatement of the reference's CommonLib operators on the intra RDO path.
 */
#include "vvc_oracle.h"
#include "orc_tables.h"
#include "orc_internal.h"
#include <stdlib.h>
#include <string.h>
#include <math.h>

/* ------------------------------------------------------------------------------------------------
 * Transforms.  The reference's partial-butterfly kernels (CL/TrQuant_EMT.cpp:51-1653) are exact
 * factorisations of the integer matrices in CL/RomTr.cpp, so dst[k*line+j] = (sum_i M[k][i]*src[j*n+i]
 * + rnd) >> shift is bit-identical (checked against oracle/_ref for every size in tests/golden).
 * ---------------------------------------------------------------------------------------------- */
const int8_t *orc_tr_matrix(int tr, int n)
{
  switch (tr * 100 + n) {
    case 2: return ORC_DCT2_2; case 4: return ORC_DCT2_4; case 8: return ORC_DCT2_8; case 16: return ORC_DCT2_16;
    case 32: return ORC_DCT2_32; case 64: return ORC_DCT2_64;
    case 104: return ORC_DCT8_4; case 108: return ORC_DCT8_8; case 116: return ORC_DCT8_16; case 132: return ORC_DCT8_32;
    case 204: return ORC_DST7_4; case 208: return ORC_DST7_8; case 216: return ORC_DST7_16; case 232: return ORC_DST7_32;
  }
  return 0;
}

/* CL/TrQuant_EMT.cpp:51 fastForwardDCT2_B4 ... (signature src,dst,shift,line,iSkipLine,iSkipLine2):
 * the last skip1 lines are not computed and read back as zero, the last skip2 coefficients are zero. */
void orc_fwd_1d(int tr, int n, const int *src, int *dst, int shift, int line, int skip1, int skip2)
{
  const int8_t *m = orc_tr_matrix(tr, n);
  const int rnd = shift > 0 ? 1 << (shift - 1) : 0;
  const int red = line - skip1, cut = n - skip2;
  for (int k = 0; k < n; k++)
    for (int j = 0; j < line; j++) {
      int v = 0;
      if (j < red && k < cut) {
        int s = 0;
        for (int i = 0; i < n; i++) s += m[k * n + i] * src[j * n + i];
        v = (s + rnd) >> shift;
      }
      dst[k * line + j] = v;
    }
}

/* CL/TrQuant_EMT.cpp:85 fastInverseDCT2_B4 ...: dst[j*n+i] = clip((sum_k M[k][i]*src[k*line+j]+rnd)>>shift) */
void orc_inv_1d(int tr, int n, const int *src, int *dst, int shift, int line, int skip1, int skip2, int cmin, int cmax)
{
  const int8_t *m = orc_tr_matrix(tr, n);
  const int rnd = 1 << (shift - 1);
  const int red = line - skip1, cut = n - skip2;
  for (int j = 0; j < line; j++)
    for (int i = 0; i < n; i++) {
      int v = 0;
      if (j < red) {
        int s = 0;
        for (int k = 0; k < cut; k++) s += m[k * n + i] * src[k * line + j];
        v = (s + rnd) >> shift;
        v = v < cmin ? cmin : v > cmax ? cmax : v;
      }
      dst[j * n + i] = v;
    }
}

static int ilog2(int v) { int r = 0; while (v > 1) { v >>= 1; r++; } return r; }

/* CL/TrQuant.cpp:835-915 xT, DCT2/DCT2, no LFNST: zero-out above 32 (853-854), shifts 892-893 */
/* explicit MTS (getTrTypes 817-830): mts_idx 0 DCT2xDCT2; 2..5: horizontal = (idx-2)&1 ? DCT8 : DST7, vertical = (idx-2)>>1 ? DCT8 : DST7
 * (orc_fwd_1d numbering: 0 DCT2, 1 DCT8, 2 DST7) */
static void mts_types(int mts_idx, int *trh, int *trv)
{
  if (mts_idx < 2) { *trh = *trv = 0; return; }
  *trh = ((mts_idx - 2) & 1) ? 1 : 2; *trv = ((mts_idx - 2) >> 1) ? 1 : 2;
}
void orc_fwd_2d_mts(const int16_t *resi, int stride, int w, int h, int bit_depth, int mts_idx, int *coef)
{
  int trh, trv; mts_types(mts_idx, &trh, &trv);
  int *block = (int *) malloc(sizeof(int) * w * h * 2), *tmp = block + w * h;
  for (int y = 0; y < h; y++) for (int x = 0; x < w; x++) block[y * w + x] = resi[y * stride + x];
  const int skipW = (trh != 0 && w == 32) ? 16 : w > 32 ? w - 32 : 0, skipH = (trv != 0 && h == 32) ? 16 : h > 32 ? h - 32 : 0;   /* 853-854 */
  const int shift1 = ilog2(w) + bit_depth + 6 - 15;
  const int shift2 = ilog2(h) + 6;
  orc_fwd_1d(trh, w, block, tmp, shift1, h, 0, skipW);
  orc_fwd_1d(trv, h, tmp, coef, shift2, w, skipW, skipH);
  free(block);
}
void orc_fwd_2d(const int16_t *resi, int stride, int w, int h, int bit_depth, int *coef) { orc_fwd_2d_mts(resi, stride, w, h, bit_depth, 0, coef); }

/* CL/TrQuant.cpp:917-992 xIT */
void orc_inv_2d_mts(const int *coef, int w, int h, int bit_depth, int mts_idx, int16_t *resi, int stride)
{
  int trh, trv; mts_types(mts_idx, &trh, &trv);
  int *tmp = (int *) malloc(sizeof(int) * w * h * 2), *block = tmp + w * h;
  const int skipW = (trh != 0 && w == 32) ? 16 : w > 32 ? w - 32 : 0, skipH = (trv != 0 && h == 32) ? 16 : h > 32 ? h - 32 : 0;
  const int cmin = -(1 << 15), cmax = (1 << 15) - 1;
  const int shift1 = 6 + 1, shift2 = (6 + 15 - 1) - bit_depth;
  orc_inv_1d(trv, h, coef, tmp, shift1, w, skipW, skipH, cmin, cmax);
  orc_inv_1d(trh, w, tmp, block, shift2, h, 0, skipW, cmin, cmax);
  for (int y = 0; y < h; y++) for (int x = 0; x < w; x++) resi[y * stride + x] = (int16_t) block[y * w + x];
  free(tmp);
}
void orc_inv_2d(const int *coef, int w, int h, int bit_depth, int16_t *resi, int stride) { orc_inv_2d_mts(coef, w, h, bit_depth, 0, resi, stride); }

/* xT / xIT for a luma block of an ISP CU (getTrTypes 752-780: DST-VII along a side of 4..16 samples, DCT-II otherwise; cfg MTS 1) incl. the 1-D forms for
 * Nx1 / 1xN sub-partitions (895-914, 970-983) */
void orc_fwd_isp(const int16_t *resi, int stride, int w, int h, int bit_depth, int *coef)
{
  const int trh = (w >= 4 && w <= 16) ? 2 : 0, trv = (h >= 4 && h <= 16) ? 2 : 0;
  int *block = (int *) malloc(sizeof(int) * w * h * 2), *tmp = block + w * h;
  for (int y = 0; y < h; y++) for (int x = 0; x < w; x++) block[y * w + x] = resi[y * stride + x];
  if (w > 1 && h > 1) {
    orc_fwd_1d(trh, w, block, tmp, ilog2(w) + bit_depth + 6 - 15, h, 0, 0);
    orc_fwd_1d(trv, h, tmp, coef, ilog2(h) + 6, w, 0, 0);
  } else if (h == 1) orc_fwd_1d(trh, w, block, coef, ilog2(w) + bit_depth + 6 - 15, 1, 0, 0);
  else orc_fwd_1d(trv, h, block, coef, ilog2(h) + bit_depth + 6 - 15, 1, 0, 0);
  free(block);
}
void orc_inv_isp(const int *coef, int w, int h, int bit_depth, int16_t *resi, int stride)
{
  const int trh = (w >= 4 && w <= 16) ? 2 : 0, trv = (h >= 4 && h <= 16) ? 2 : 0;
  int *tmp = (int *) malloc(sizeof(int) * w * h * 2), *block = tmp + w * h;
  const int cmin = -(1 << 15), cmax = (1 << 15) - 1, shift2 = (6 + 15 - 1) - bit_depth;
  if (w > 1 && h > 1) {
    orc_inv_1d(trv, h, coef, tmp, 6 + 1, w, 0, 0, cmin, cmax);
    orc_inv_1d(trh, w, tmp, block, shift2, h, 0, 0, cmin, cmax);
  } else if (w == 1) orc_inv_1d(trv, h, coef, block, shift2 + 1, 1, 0, 0, cmin, cmax);
  else orc_inv_1d(trh, w, coef, block, shift2 + 1, 1, 0, 0, cmin, cmax);
  for (int y = 0; y < h; y++) for (int x = 0; x < w; x++) resi[y * stride + x] = (int16_t) block[y * w + x];
  free(tmp);
}

/* TrQuant::transformNxN, pruning overload (CL/TrQuant.cpp:1049-1124) for the candidate list {DCT2, 2, 3, 4, 5} (no transform skip):
 * sum |coeff| per transform; an entry stays when its sum <= fac * sum(DCT2) — list position 1 is compared against sum(DCT2) itself, the
 * reference's transform-skip threshold applied by position — and no more than max_cand + 1 entries are kept, in list order. */
void orc_mts_prune(const int16_t *resi, int stride, int w, int h, int bit_depth, int max_cand, int test[5])
{
  static const double facBB[5] = { 1.2, 1.3, 1.3, 1.4, 1.5 };
  static const int idx_of[5] = { 0, 2, 3, 4, 5 };
  int *coef = (int *) malloc(sizeof(int) * w * h);
  int sums[5];
  for (int k = 0; k < 5; k++) {
    orc_fwd_2d_mts(resi, stride, w, h, bit_depth, idx_of[k], coef);
    int sa = 0;
    for (int i = 0; i < w * h; i++) sa += coef[i] < 0 ? -coef[i] : coef[i];
    sums[k] = sa;
  }
  free(coef);
  const double fac = facBB[ilog2(w > h ? w : h) - 2];
  const double thr = fac * sums[0], thrTS = sums[0];
  int numTests = 0;
  for (int k = 0; k < 5; k++) {
    const int t = (double) sums[k] <= (k == 1 ? thrTS : thr) && numTests <= max_cand;
    test[k] = t; numTests += t;
  }
}

/* CL/Quant.cpp:994-1089 Quant::quant without scaling lists / sign hiding; I-slice rounding 171<<(qbits-9).
 * returns uiAbsSum */
int orc_quant(const int *coef, int w, int h, int bit_depth, int qp, int16_t *level)
{
  const int lw = ilog2(w), lh = ilog2(h);
  const int need_sqrt = (lw + lh) & 1;                         /* CL/UnitTools.cpp:4650-4660 */
  const int scale = ORC_QUANT_SCALES[need_sqrt * 6 + qp % 6];
  const int tr_shift = 15 - bit_depth - ((lw + lh) >> 1) + (need_sqrt ? -1 : 0);   /* CL/Quant.h getTransformShift */
  const int qbits = 14 + qp / 6 + tr_shift;
  const int64_t add = (int64_t) 171 << (qbits - 9);
  int abs_sum = 0;
  for (int i = 0; i < w * h; i++) {
    const int c = coef[i];
    const int64_t t = (int64_t) (c < 0 ? -c : c) * scale;
    int q = (int) ((t + add) >> qbits);
    abs_sum += q;
    if (c < 0) q = -q;
    q = q < -32768 ? -32768 : q > 32767 ? 32767 : q;
    level[i] = (int16_t) q;
  }
  return abs_sum;
}

/* Quant::quant for a block of a CU with lfnstIdx > 0 (CL/Quant.cpp:1054-1058, JVET_O0094): the level buffer is cleared and only the first
 * max_coefs buffer positions (8 for 4x4 / 8x8 blocks, else 16 -- buffer order, as the reference indexes them) are quantised */
int orc_quant_lfnst(const int *coef, int w, int h, int bit_depth, int qp, int16_t *level)
{
  const int lw = ilog2(w), lh = ilog2(h);
  const int need_sqrt = (lw + lh) & 1;
  const int scale = ORC_QUANT_SCALES[need_sqrt * 6 + qp % 6];
  const int tr_shift = 15 - bit_depth - ((lw + lh) >> 1) + (need_sqrt ? -1 : 0);
  const int qbits = 14 + qp / 6 + tr_shift;
  const int64_t add = (int64_t) 171 << (qbits - 9);
  const int max_coefs = ((w == 4 && h == 4) || (w == 8 && h == 8)) ? 8 : 16;
  int abs_sum = 0;
  memset(level, 0, (size_t) w * h * sizeof(int16_t));
  for (int i = 0; i < max_coefs; i++) {
    const int c = coef[i];
    const int64_t t = (int64_t) (c < 0 ? -c : c) * scale;
    int q = (int) ((t + add) >> qbits);
    abs_sum += q;
    if (c < 0) q = -q;
    q = q < -32768 ? -32768 : q > 32767 ? 32767 : q;
    level[i] = (int16_t) q;
  }
  return abs_sum;
}

/* ------------------------------------------------------------------------------------------------
 * LFNST (CL/TrQuant.cpp:241-560; kernels CL/RomLFNST.cpp as data in orc_lfnst_tables.h)
 * orc_lfnst_mode   = PU::getWideAngIntraMode (CL/UnitTools.cpp:963-989) + TrQuant::getLFNSTIntraMode (293-311): the index into the
 *                    mode -> kernel-set table and (>= 0x100) the transpose flag of getTransposeFlag (312-317); dir is the block's final
 *                    intra mode (planar for MIP, the co-located luma mode for DM / CCLM), w x h the transform block
 * orc_lfnst_keep   = the zero-out of the primary transform for lfnstIdx > 0 (xT 855-868): only the top-left 4x4 (one side 4) or 8x8 stays
 * orc_fwd_lfnst    = xFwdLfnst 437-560, orc_inv_lfnst = xInvLfnst 319-436 on a w x h coefficient block (stride w)
 * ---------------------------------------------------------------------------------------------- */
#include "orc_lfnst_tables.h"
int orc_lfnst_mode(int dir, int w, int h)
{
  int pm = dir;
  if (dir >= 2) {
    static const int modeShift[6] = { 0, 6, 10, 12, 14, 15 };
    const int lw = ilog2(w), lh = ilog2(h), ds = lw > lh ? lw - lh : lh - lw;
    if (w > h && dir < 2 + modeShift[ds]) pm += 65;
    else if (h > w && pm > 66 - modeShift[ds]) pm -= 67;
  }
  const int ext = pm < 0 ? pm + 14 + 67 : pm >= 67 ? pm + 14 : pm;
  const int transpose = (ext >= 67 && ext >= 67 + 14) || (ext < 67 && ext > 34);
  return ext | (transpose ? 0x100 : 0);
}
void orc_lfnst_keep(int *coef, int w, int h)
{
  int kw = w, kh = h;
  if ((w == 4 && h > 4) || (w > 4 && h == 4)) { kw = 4; kh = 4; }
  else if (w >= 8 && h >= 8) { kw = 8; kh = 8; }
  for (int y = 0; y < h; y++) for (int x = 0; x < w; x++) if (x >= kw || y >= kh) coef[y * w + x] = 0;
}
/* place of the k-th entry of the diagonal scan of the top-left 8x8 (4x4 groups) / 4x4 region */
static int lfnst_scan_pos(int k, int w)
{
  static const uint8_t d4[16][2] = { {0,0},{0,1},{1,0},{0,2},{1,1},{2,0},{0,3},{1,2},{2,1},{3,0},{1,3},{2,2},{3,1},{2,3},{3,2},{3,3} };
  static const uint8_t g[4][2] = { {0,0},{0,1},{1,0},{1,1} };
  const int grp = k >> 4, in = k & 15;
  return (g[grp][1] * 4 + d4[in][1]) * w + g[grp][0] * 4 + d4[in][0];
}
/* index inside the (possibly transposed) kernel input / output vector of the region sample (x, y) */
static int lfnst_vec_pos(int x, int y, int sb, int transpose)
{
  if (transpose) { const int t = x; x = y; y = t; }
  return sb == 4 ? y * 4 + x : (y < 4 ? y * 8 + x : 32 + (y - 4) * 4 + x);
}
void orc_fwd_lfnst(int *coef, int w, int h, int mode, int lfnst_idx)
{
  if (!lfnst_idx || w < 4 || h < 4) return;
  const int sb = (w >= 8 && h >= 8) ? 8 : 4, trSize = sb == 8 ? 48 : 16;
  const int nOut = ((w == 4 && h == 4) || (w == 8 && h == 8)) ? 8 : 16;
  const int transpose = (mode >> 8) & 1, set = ORC_LFNST_LUT[mode & 255];
  const int8_t *M = sb == 8 ? ORC_LFNST_8x8 + ((set * 2 + lfnst_idx - 1) * 16) * 48 : ORC_LFNST_4x4 + ((set * 2 + lfnst_idx - 1) * 16) * 16;
  int in[48], out[48];
  for (int y = 0; y < sb; y++) for (int x = 0; x < sb; x++) if (sb == 4 || x < 4 || y < 4) in[lfnst_vec_pos(x, y, sb, transpose)] = coef[y * w + x];
  for (int j = 0; j < trSize; j++) {
    int c = 0;
    if (j < nOut) { for (int i = 0; i < trSize; i++) c += in[i] * M[j * trSize + i]; c = (c + 64) >> 7; }
    out[j] = c;
  }
  for (int k = 0; k < trSize; k++) coef[lfnst_scan_pos(k, w)] = out[k];
}
void orc_inv_lfnst(int *coef, int w, int h, int mode, int lfnst_idx)
{
  if (!lfnst_idx || w < 4 || h < 4) return;
  const int sb = (w >= 8 && h >= 8) ? 8 : 4, trSize = sb == 8 ? 48 : 16;
  const int nIn = ((w == 4 && h == 4) || (w == 8 && h == 8)) ? 8 : 16;
  const int transpose = (mode >> 8) & 1, set = ORC_LFNST_LUT[mode & 255];
  const int8_t *M = sb == 8 ? ORC_LFNST_8x8 + ((set * 2 + lfnst_idx - 1) * 16) * 48 : ORC_LFNST_4x4 + ((set * 2 + lfnst_idx - 1) * 16) * 16;
  int in[16], out[48];
  for (int k = 0; k < 16; k++) in[k] = coef[lfnst_scan_pos(k, w)];
  for (int j = 0; j < trSize; j++) {
    int r = 0;
    for (int i = 0; i < nIn; i++) r += in[i] * M[i * trSize + j];
    r = (r + 64) >> 7;
    out[j] = r < -32768 ? -32768 : r > 32767 ? 32767 : r;
  }
  for (int y = 0; y < sb; y++) for (int x = 0; x < sb; x++) if (sb == 4 || x < 4 || y < 4) coef[y * w + x] = out[lfnst_vec_pos(x, y, sb, transpose)];
}

/* CL/Quant.cpp:423-549 Quant::dequant, flat scaling */
void orc_dequant(const int16_t *level, int w, int h, int bit_depth, int qp, int *coef)
{
  const int lw = ilog2(w), lh = ilog2(h);
  const int need_sqrt = (lw + lh) & 1;
  const int scale = ORC_INV_QUANT_SCALES[need_sqrt * 6 + qp % 6];
  const int tr_shift = 15 - bit_depth - ((lw + lh) >> 1) + (need_sqrt ? -1 : 0);
  const int right_shift = 6 - (tr_shift + qp / 6);
  const int scale_bits = 6 + 1;
  int tbd = 32 + right_shift - scale_bits; if (tbd > 16) tbd = 16;
  const int in_min = -(1 << (tbd - 1)), in_max = (1 << (tbd - 1)) - 1;
  for (int i = 0; i < w * h; i++) {
    int q = level[i]; q = q < in_min ? in_min : q > in_max ? in_max : q;
    int v;
    if (right_shift > 0) v = (q * scale + (1 << (right_shift - 1))) >> right_shift;
    else v = (q * scale) << (-right_shift);
    coef[i] = v < -32768 ? -32768 : v > 32767 ? 32767 : v;
  }
}

/* ------------------------------------------------------------------------------------------------
 * Distortion: CL/RdCost.cpp xGetSAD (generic), xGetSSE, xGetHADs 2746-2861 and its kernels.
 * ---------------------------------------------------------------------------------------------- */
uint64_t orc_sad(const int16_t *a, int sa, const int16_t *b, int sb, int w, int h)
{
  uint64_t s = 0;
  for (int y = 0; y < h; y++) for (int x = 0; x < w; x++) s += (uint64_t) abs(a[y * sa + x] - b[y * sb + x]);
  return s;
}
uint64_t orc_sse(const int16_t *a, int sa, const int16_t *b, int sb, int w, int h)
{
  uint64_t s = 0;
  for (int y = 0; y < h; y++) for (int x = 0; x < w; x++) { const int d = a[y * sa + x] - b[y * sb + x]; s += (uint64_t) (d * d); }
  return s;
}
/* Walsh-Hadamard of a bw x bh tile of differences: sum of absolute transform coefficients.  The
 * reference's butterflies (xCalcHADs4x4 2118, 8x8 2214, 16x8 2311, 8x16 2457, 4x8 2593, 8x4 2667)
 * compute the same coefficient set in another order; the normalisation per tile shape is the
 * reference's: 4x4 (s+1)>>1 (2209), 8x8 (s+2)>>2 (2306), non-square (int)(s/sqrt(bw*bh)*2) (2452...). */
static uint64_t had_tile(const int16_t *a, int sa, const int16_t *b, int sb, int bw, int bh)
{
  int m[16 * 16];
  for (int y = 0; y < bh; y++) for (int x = 0; x < bw; x++) m[y * bw + x] = a[y * sa + x] - b[y * sb + x];
  for (int y = 0; y < bh; y++)
    for (int len = 1; len < bw; len <<= 1)
      for (int i = 0; i < bw; i += 2 * len)
        for (int j = i; j < i + len; j++) { int p = m[y * bw + j], q = m[y * bw + j + len]; m[y * bw + j] = p + q; m[y * bw + j + len] = p - q; }
  for (int x = 0; x < bw; x++)
    for (int len = 1; len < bh; len <<= 1)
      for (int i = 0; i < bh; i += 2 * len)
        for (int j = i; j < i + len; j++) { int p = m[j * bw + x], q = m[(j + len) * bw + x]; m[j * bw + x] = p + q; m[(j + len) * bw + x] = p - q; }
  int s = 0;
  for (int i = 0; i < bw * bh; i++) s += abs(m[i]);
  if (bw == 2 && bh == 2) return (uint64_t) s;                 /* 2110-2115: no normalisation */
  if (bw == 4 && bh == 4) return (uint64_t) ((s + 1) >> 1);
  if (bw == 8 && bh == 8) return (uint64_t) ((s + 2) >> 2);
  return (uint64_t) (int) (s / sqrt((double) bw * bh) * 2);
}
void orc_satd_tile_shape(int w, int h, int *bw, int *bh)
{
  if (w > h && (h & 7) == 0 && (w & 15) == 0) { *bw = 16; *bh = 8; }
  else if (w < h && (w & 7) == 0 && (h & 15) == 0) { *bw = 8; *bh = 16; }
  else if (w > h && (h & 3) == 0 && (w & 7) == 0) { *bw = 8; *bh = 4; }
  else if (w < h && (w & 3) == 0 && (h & 7) == 0) { *bw = 4; *bh = 8; }
  else if ((h & 7) == 0 && (w & 7) == 0) { *bw = 8; *bh = 8; }
  else if ((h & 3) == 0 && (w & 3) == 0) { *bw = 4; *bh = 4; }
  else { *bw = 2; *bh = 2; }
}
uint64_t orc_satd(const int16_t *a, int sa, const int16_t *b, int sb, int w, int h)
{
  int bw, bh; orc_satd_tile_shape(w, h, &bw, &bh);
  uint64_t s = 0;
  for (int y = 0; y < h; y += bh) for (int x = 0; x < w; x += bw) s += had_tile(a + y * sa + x, sa, b + y * sb + x, sb, bw, bh);
  return s;
}

/* CL/RdCost.cpp:63-88: DistScale = 2^15/lambda ; cost = DistScale*double(dist) + double(fracBits) */
double orc_calc_rd_cost(double lambda, uint64_t frac_bits, uint64_t dist)
{
  const double dist_scale = (double) (1 << 15) / lambda;
  const double a = dist_scale * (double) dist;
  return a + (double) frac_bits;
}

/* ------------------------------------------------------------------------------------------------
 * CABAC probability model: CL/Contexts.h:86-155, CL/Contexts.cpp:135-151 (JVET_O0065 init), 1818-1833
 * ---------------------------------------------------------------------------------------------- */
void orc_ctx_init(int qp, uint16_t *s0, uint16_t *s1)
{
  if (qp < 0) qp = 0;
  if (qp > 63) qp = 63;
  for (int k = 0; k < ORC_NUM_CTX; k++) {
    const int id = ORC_CTX_INIT_I[k];
    const int slope = (id >> 3) - 4, offset = ((id & 7) * 18) + 1;
    int st = ((slope * (qp - 16)) >> 1) + offset;
    st = st < 1 ? 1 : st > 127 ? 127 : st;
    const int p1 = st << 8;
    s0[k] = (uint16_t) (p1 & 0x7FE0);
    s1[k] = (uint16_t) (p1 & 0x7FFE);
  }
}
uint64_t orc_ctx_code_bins(uint16_t *s0, uint16_t *s1, int ctx_id, const uint8_t *bins, int n)
{
  orc_cabac c; memset(&c, 0, sizeof c);
  c.s0[ctx_id] = *s0; c.s1[ctx_id] = *s1;
  for (int i = 0; i < n; i++) orc_enc_bin(&c, bins[i], ctx_id);
  *s0 = c.s0[ctx_id]; *s1 = c.s1[ctx_id];
  return c.bits;
}

/* ------------------------------------------------------------------------------------------------
 * Reference samples: CL/IntraPrediction.cpp:1215-1468 xFillReferenceSamples.  Layout of the output
 * like m_piYuvExt: stride = 2w+1+mrl, row 0 = top-left + above row, column 0 = left column.
 * Availability of a neighbouring unit (isAbove/Left/...Available 1524-1663 → cs.isDecomp +
 * getCURestricted) is supplied as a byte map `avail` with one entry per (1<<unit_log2)^2 samples of
 * this component: value == tag ⇔ already coded in the current partition path, same slice and tile
 * (the tag is tile id + 1, so other tiles' samples read as unavailable).
 * ---------------------------------------------------------------------------------------------- */
static int unit_avail(const uint8_t *avail, int avail_stride, int unit_log2, int pic_w, int pic_h, int px, int py, int tag)
{
  if (px < 0 || py < 0 || px >= pic_w || py >= pic_h) return 0;
  return avail[(py >> unit_log2) * avail_stride + (px >> unit_log2)] == tag;
}

void orc_fill_ref_samples(const int16_t *reco, int stride, int pic_w, int pic_h, const uint8_t *avail, int avail_stride,
                          int unit_log2, int tag, int x, int y, int w, int h, int mrl, int bit_depth, int16_t *ref)
{
  const int predSize = 2 * w, predHSize = 2 * h;
  const int predStride = predSize + 1 + mrl;
  /* unit size: pcv.minCUWidth (4) for luma, >>1 for 4:2:0 chroma (1239-1240); the caller passes the
   * granularity of its availability map, the unit size follows the reference */
  const int unitW = unit_log2 == 2 ? 4 : 2, unitH = unitW;
  const int totalAbove = (predSize + unitW - 1) / unitW, totalLeft = (predHSize + unitH - 1) / unitH;
  const int totalUnits = totalAbove + totalLeft + 1;
  const int numAbove = w / unitW > 1 ? w / unitW : 1, numLeft = h / unitH > 1 ? h / unitH : 1;
  const int numAboveRight = totalAbove - numAbove, numLeftBelow = totalLeft - numLeft;
  uint8_t flags[4 * 32 + 1 + 64];
  memset(flags, 0, sizeof flags);
  int numIntra = 0;
  /* the availability map is indexed in 4x4 (luma) or 4x4-chroma units by the caller: chroma units of 2
   * samples share the entry of their enclosing 4x4 chroma block */
  const int alog = unit_log2;
#define AV(px, py) unit_avail(avail, avail_stride, alog, pic_w, pic_h, (px), (py), tag)
  flags[totalLeft] = (uint8_t) AV(x - 1, y - 1);
  numIntra += flags[totalLeft];
  for (int i = 0; i < numAbove; i++) { if (!AV(x + i * unitW, y - 1)) break; flags[totalLeft + 1 + i] = 1; numIntra++; }
  for (int i = 0; i < numAboveRight; i++) { if (!AV(x + w - 1 + unitW + i * unitW, y - 1)) break; flags[totalLeft + 1 + numAbove + i] = 1; numIntra++; }
  for (int i = 0; i < numLeft; i++) { if (!AV(x - 1, y + i * unitH)) break; flags[totalLeft - 1 - i] = 1; numIntra++; }
  for (int i = 0; i < numLeftBelow; i++) { if (!AV(x - 1, y + h - 1 + unitH + i * unitH)) break; flags[totalLeft - 1 - numLeft - i] = 1; numIntra++; }
#undef AV
  const int16_t *src = reco + y * stride + x;
  const int16_t dc = (int16_t) (1 << (bit_depth - 1));
  if (numIntra == 0) {
    for (int j = 0; j <= predSize + mrl; j++) ref[j] = dc;
    for (int i = 1; i <= predHSize + mrl; i++) ref[i * predStride] = dc;
  } else if (numIntra == totalUnits) {
    const int16_t *p = src - (1 + mrl) * stride - (1 + mrl);
    for (int j = 0; j <= predSize + mrl; j++) ref[j] = p[j];
    p = src - mrl * stride - (1 + mrl);
    for (int i = 1; i <= predHSize + mrl; i++) { ref[i * predStride] = *p; p += stride; }
  } else {
    const int16_t *p = src - (1 + mrl) * stride - (1 + mrl);
    int16_t *d = ref;
    if (flags[totalLeft]) {
      d[0] = p[0];
      for (int i = 1; i <= mrl; i++) { d[i] = p[i]; d[i * predStride] = p[i * stride]; }
    }
    p += (1 + mrl) * stride; d += (1 + mrl) * predStride;
    for (int u = totalLeft - 1; u > 0; u--) {
      if (flags[u]) for (int i = 0; i < unitH; i++) d[i * predStride] = p[i * stride];
      p += unitH * stride; d += unitH * predStride;
    }
    if (flags[0]) {
      const int last = (predHSize % unitH == 0) ? unitH : predHSize % unitH;
      for (int i = 0; i < last; i++) d[i * predStride] = p[i * stride];
    }
    p = src - stride * (1 + mrl); d = ref + 1 + mrl;
    for (int u = totalLeft + 1; u < totalUnits - 1; u++) {
      if (flags[u]) for (int j = 0; j < unitW; j++) d[j] = p[j];
      p += unitW; d += unitW;
    }
    if (flags[totalUnits - 1]) {
      const int last = (predSize % unitW == 0) ? unitW : predSize % unitW;
      for (int j = 0; j < last; j++) d[j] = p[j];
    }
    /* pad unavailable (1359-1458) */
    d = ref;
    int lastAvail = 0;
    if (!flags[0]) {
      int first = 1;
      while (first < totalUnits && !flags[first]) first++;
      int row = 0, col = 0;
      if (first < totalLeft) row = (totalLeft - first) * unitH + mrl;
      else if (first == totalLeft) row = mrl;
      else col = (first - totalLeft - 1) * unitW + 1 + mrl;
      const int16_t v = d[col + row * predStride];
      for (int i = predHSize + mrl; i > row; i--) d[i * predStride] = v;
      for (int j = 0; j < col; j++) d[j] = v;
      lastAvail = first;
    }
    int cur = lastAvail + 1;
    while (cur < totalUnits) {
      if (!flags[cur]) {
        int row = 0, col = 0;
        if (lastAvail < totalLeft) row = (totalLeft - lastAvail - 1) * unitH + mrl + 1;
        else if (lastAvail == totalLeft) col = mrl;
        else col = (lastAvail - totalLeft) * unitW + mrl;
        const int16_t v = d[col + row * predStride];
        if (cur < totalLeft) {
          for (int i = row - 1; i >= row - unitH; i--) d[i * predStride] = v;
        } else if (cur == totalLeft) {
          for (int i = 1; i < mrl + 1; i++) d[i * predStride] = v;
          for (int j = 0; j < mrl + 1; j++) d[j] = v;
        } else {
          const int n = (cur == totalUnits - 1) ? ((predSize % unitW == 0) ? unitW : predSize % unitW) : unitW;
          for (int j = col + 1; j <= col + n; j++) d[j] = v;
        }
      }
      lastAvail = cur;
      cur++;
    }
  }
}

/* CL/IntraPrediction.cpp:1470-1522 xFilterReferenceSamples ([1 2 1]/4) */
void orc_filter_ref_samples(const int16_t *unf, int16_t *flt, int w, int h, int mrl)
{
  const int predSize = 2 * w + mrl, predHSize = 2 * h + mrl;
  const int st = predSize + 1;
  const int16_t *s = unf + st * predHSize;
  int16_t *d = flt + st * predHSize;
  *d = *s; d -= st; s -= st;
  for (int i = 1; i < predHSize; i++, d -= st, s -= st) *d = (int16_t) ((s[st] + 2 * s[0] + s[-st] + 2) >> 2);
  *d = (int16_t) ((s[st] + 2 * s[0] + s[1] + 2) >> 2);
  d++; s++;
  for (int i = 1; i < predSize; i++, d++, s++) *d = (int16_t) ((s[1] + 2 * s[0] + s[-1] + 2) >> 2);
  *d = *s;
}

/* CL/IntraPrediction.cpp:76 g_intraGaussFilter (VVC spec table 8-? fG), data */
static const int8_t gauss_filter[32][4] = {
  {16,32,16,0},{15,29,17,3},{15,29,17,3},{14,29,18,3},{13,29,18,4},{13,28,19,4},{13,28,19,4},{12,28,20,4},
  {11,28,20,5},{11,27,21,5},{10,27,22,5},{9,27,22,6},{9,26,23,6},{9,26,23,6},{8,25,24,7},{8,25,24,7},
  {8,24,24,8},{7,24,25,8},{7,24,25,8},{6,23,26,9},{6,23,26,9},{6,22,27,9},{5,22,27,10},{5,21,27,11},
  {5,20,28,11},{4,20,28,12},{4,19,28,13},{4,19,28,13},{4,18,29,13},{3,18,29,14},{3,17,29,15},{3,17,29,15} };
static const int16_t ang_table[32] = { 0, 1, 2, 3, 4, 6, 8, 10, 12, 14, 16, 18, 20, 23, 26, 29, 32, 35, 39, 45, 51, 57, 64, 73, 86, 102, 128, 171, 256, 341, 512, 1024 };
static const int16_t inv_ang_table[32] = { 0, 16384, 8192, 5461, 4096, 2731, 2048, 1638, 1365, 1170, 1024, 910, 819, 712, 630, 565,
  512, 468, 420, 364, 321, 287, 256, 224, 191, 161, 128, 96, 64, 48, 32, 16 };
static const uint8_t intra_filter_thr[8] = { 24, 24, 24, 14, 2, 0, 0, 0 };  /* m_aucIntraFilter 58-74 */

/* CL/IntraPrediction.cpp:287-303 getWideAngle + 487-618 initPredIntraParams (no ISP/MIP/BDPCM) */
void orc_init_pred_params(int w, int h, int is_luma, int mode, int mrl, orc_ipa *p)
{
  int pm = mode;
  if (pm > ORC_DC && pm <= ORC_VDIA) {
    static const int modeShift[] = { 0, 6, 10, 12, 14, 15 };
    const int ds = abs(ilog2(w) - ilog2(h));
    if (w > h && pm < 2 + modeShift[ds]) pm += ORC_VDIA - 1;
    else if (h > w && pm > ORC_VDIA - modeShift[ds]) pm -= ORC_VDIA - 1;
  }
  p->pred_mode = pm;
  p->is_ver = pm >= ORC_DIA;
  p->mrl = is_luma ? mrl : 0;
  p->ref_filter = 0; p->interp = 0;
  p->pdpc = ((w >= 4 && h >= 4) || !is_luma) && p->mrl == 0;
  p->angle = 0; p->inv_angle = 0; p->ang_scale = -1;
  const int am = p->is_ver ? pm - ORC_VER : -(pm - ORC_HOR);
  int absAng = 0;
  if (mode > ORC_DC && mode < ORC_NUM_LUMA_MODE) {
    const int a = abs(am);
    absAng = ang_table[a];
    p->inv_angle = inv_ang_table[a];
    p->angle = am < 0 ? -absAng : absAng;
    if (am < 0) p->pdpc = 0;
    else if (am > 0) {
      const int side = p->is_ver ? h : w;
      int sc = ilog2(side) - (ilog2(3 * p->inv_angle - 2) - 8);
      if (sc > 2) sc = 2;
      p->ang_scale = sc;
      p->pdpc &= sc >= 0;
    }
  }
  if (!is_luma || p->mrl || mode == ORC_DC) { /* no ref filter (559-575) */ }
  else if (mode == ORC_PLANAR) p->ref_filter = w * h > 32;
  else {
    const int d1 = abs(pm - ORC_HOR), d2 = abs(pm - ORC_VER);
    const int diff = d1 < d2 ? d1 : d2;
    const int log2Size = (ilog2(w) + ilog2(h)) >> 1;
    if (diff > intra_filter_thr[log2Size]) {
      const int is_int = (absAng & 0x1F) == 0;          /* isIntegerSlope */
      p->ref_filter = is_int;
      p->interp = !is_int;
    }
  }
}

static int16_t clip_pel(int v, int bit_depth) { const int mx = (1 << bit_depth) - 1; return (int16_t) (v < 0 ? 0 : v > mx ? mx : v); }

/* CL/IntraPrediction.cpp:316-398 predIntraAng (dispatch + planar/DC PDPC), 426-479 planar,
 * 248-285/480 DC, 633-935 xPredIntraAng */
/* initPredIntraParams for a prediction region (w x h) of an ISP CU (cuw x cuh): wide-angle mapping by the CU's shape, no reference smoothing, the cubic
 * interpolation filter, PDPC by the region's size (487-618 with useISP, JVET_O0502) */
void orc_init_pred_params_isp(int cuw, int cuh, int w, int h, int mode, orc_ipa *p)
{
  orc_ipa q; orc_init_pred_params(cuw, cuh, 1, mode, 0, &q);         /* pred_mode / is_ver / angle / inv_angle from the CU's shape */
  *p = q;
  p->ref_filter = 0; p->interp = 0;
  p->pdpc = w >= 4 && h >= 4;
  p->ang_scale = -1;
  if (mode > ORC_DC && mode < ORC_NUM_LUMA_MODE) {
    const int am = p->is_ver ? p->pred_mode - ORC_VER : -(p->pred_mode - ORC_HOR);
    if (am < 0) p->pdpc = 0;
    else if (am > 0) {
      const int side = p->is_ver ? h : w;
      int sc = ilog2(side) - (ilog2(3 * p->inv_angle - 2) - 8);
      if (sc > 2) sc = 2;
      p->ang_scale = sc;
      p->pdpc &= sc >= 0;
    }
  }
}
static void pred_core(const int16_t *src, int st, int w, int h, int is_luma, int mode, const orc_ipa *ipp, int top_len, int left_len, int bit_depth, int16_t *pred, int ps);
void orc_pred_intra(const int16_t *ref_unf, const int16_t *ref_flt, int w, int h, int is_luma, int mode, int mrl,
                    int bit_depth, int16_t *pred, int ps)
{
  orc_ipa ip; orc_init_pred_params(w, h, is_luma, mode, mrl, &ip);
  pred_core(ip.ref_filter ? ref_flt : ref_unf, 2 * w + 1 + ip.mrl, w, h, is_luma, mode, &ip, 2 * w, 2 * h, bit_depth, pred, ps);
}
/* prediction of one region of an ISP CU from its (unfiltered) reference buffer src (row 0: corner + top_len samples, column 0: left_len samples below the corner,
 * stride st): m_topRefLength / m_leftRefLength = CU side + region side (CL/IntraPrediction.cpp:1092-1199) */
void orc_pred_intra_isp(const int16_t *src, int st, int cuw, int cuh, int w, int h, int mode, int bit_depth, int16_t *pred, int ps)
{
  orc_ipa ip; orc_init_pred_params_isp(cuw, cuh, w, h, mode, &ip);
  pred_core(src, st, w, h, 1, mode, &ip, cuw + w, cuh + h, bit_depth, pred, ps);
}
static void pred_core(const int16_t *src, int st, int w, int h, int is_luma, int mode, const orc_ipa *ipp, int top_len, int left_len, int bit_depth, int16_t *pred, int ps)
{
  const orc_ipa ip = *ipp;
  const int mrl = ip.mrl;
#define TOP(i) src[(i)]
#define LEFT(i) src[(i) * st]
  if (mode == ORC_PLANAR) {
    int leftCol[ORC_MAX_CU + 1], topRow[ORC_MAX_CU + 1], bottomRow[ORC_MAX_CU], rightCol[ORC_MAX_CU];
    const int l2w = ilog2(w < 2 ? 2 : w), l2h = ilog2(h < 2 ? 2 : h);      /* 430-431: one-sample sides of ISP sub-partitions weigh like two */
    for (int k = 0; k < w + 1; k++) topRow[k] = TOP(k + 1);
    for (int k = 0; k < h + 1; k++) leftCol[k] = LEFT(k + 1);
    const int bl = leftCol[h], tr = topRow[w];
    for (int k = 0; k < w; k++) { bottomRow[k] = bl - topRow[k]; topRow[k] <<= l2h; }
    for (int k = 0; k < h; k++) { rightCol[k] = tr - leftCol[k]; leftCol[k] <<= l2w; }
    const int fs = 1 + l2w + l2h, off = 1 << (l2w + l2h);
    for (int y = 0; y < h; y++) {
      int hp = leftCol[y];
      for (int x = 0; x < w; x++) {
        hp += rightCol[y]; topRow[x] += bottomRow[x];
        pred[y * ps + x] = (int16_t) (((hp << l2h) + (topRow[x] << l2w) + off) >> fs);
      }
    }
  } else if (mode == ORC_DC) {
    int sum = 0;
    const int denom = (w == h) ? (w << 1) : (w > h ? w : h);
    if (w >= h) for (int i = 0; i < w; i++) sum += TOP(mrl + 1 + i);
    if (w <= h) for (int i = 0; i < h; i++) sum += LEFT(mrl + 1 + i);
    const int16_t dcv = (int16_t) ((sum + (denom >> 1)) >> ilog2(denom));
    for (int y = 0; y < h; y++) for (int x = 0; x < w; x++) pred[y * ps + x] = dcv;
  } else {
    int16_t refAbove[2 * ORC_MAX_CU + 3 + 33 * 3], refLeft[2 * ORC_MAX_CU + 3 + 33 * 3];
    int16_t *refMain, *refSide;
    const int ang = ip.angle, inv = ip.inv_angle, ver = ip.is_ver;
    int W = w, H = h;
    if (ang < 0) {
      for (int x = 0; x <= w + 1 + mrl; x++) refAbove[x + h] = TOP(x);
      for (int y = 0; y <= h + 1 + mrl; y++) refLeft[y + w] = LEFT(y);
      refMain = ver ? refAbove + h : refLeft + w;
      refSide = ver ? refLeft + w : refAbove + h;
      const int sizeSide = ver ? h : w;
      for (int k = -sizeSide; k <= -1; k++) { int idx = (-k * inv + 256) >> 9; if (idx > sizeSide) idx = sizeSide; refMain[k] = refSide[idx]; }
    } else {
      for (int x = 0; x <= top_len + mrl; x++) refAbove[x] = TOP(x);
      for (int y = 0; y <= left_len + mrl; y++) refLeft[y] = LEFT(y);
      refMain = ver ? refAbove : refLeft;
      refSide = ver ? refLeft : refAbove;
      const int lr = ilog2(w) - ilog2(h);
      int s = ver ? lr : -lr; if (s < 0) s = 0;
      const int maxIndex = (mrl << s) + 2;
      const int refLength = ver ? top_len : left_len;
      const int16_t val = refMain[refLength + mrl];
      for (int z = 1; z <= maxIndex; z++) refMain[refLength + mrl + z] = val;
    }
    int16_t tmp[ORC_MAX_CU * ORC_MAX_CU];
    int16_t *dst = ver ? pred : tmp;
    const int ds = ver ? ps : ORC_MAX_CU;
    if (!ver) { W = h; H = w; }
    refMain += mrl; refSide += mrl;
    if (ang == 0) {
      for (int y = 0; y < H; y++) {
        for (int x = 0; x < W; x++) dst[y * ds + x] = refMain[x + 1];
        if (ip.pdpc) {
          const int scale = (ilog2(W) + ilog2(H) - 2) >> 2;
          const int16_t topLeft = refMain[0], left = refSide[1 + y];
          const int lim = (3 << scale) < W ? (3 << scale) : W;
          for (int x = 0; x < lim; x++) {
            const int wL = 32 >> (2 * x >> scale);
            const int16_t val = dst[y * ds + x];
            dst[y * ds + x] = clip_pel(val + ((wL * (left - topLeft) + 32) >> 6), bit_depth);
          }
        }
      }
    } else {
      const int is_int = (abs(ang) & 0x1F) == 0;
      for (int y = 0, deltaPos = ang * (1 + mrl); y < H; y++, deltaPos += ang) {
        const int di = deltaPos >> 5, df = deltaPos & 31;
        int16_t *row = dst + y * ds;
        if (!is_int) {
          if (is_luma) {
            const int8_t *f = ip.interp ? gauss_filter[df] : &ORC_CUBIC_FILTER[df * 4];
            for (int x = 0; x < W; x++) {
              const int v = (f[0] * refMain[di + x] + f[1] * refMain[di + x + 1] + f[2] * refMain[di + x + 2] + f[3] * refMain[di + x + 3] + 32) >> 6;
              row[x] = clip_pel((int16_t) v, bit_depth);
            }
          } else {
            for (int x = 0; x < W; x++) {
              const int p0 = refMain[di + x + 1], p1 = refMain[di + x + 2];
              row[x] = (int16_t) (p0 + ((df * (p1 - p0) + 16) >> 5));
            }
          }
        } else for (int x = 0; x < W; x++) row[x] = refMain[x + di + 1];
        if (ip.pdpc) {
          const int scale = ip.ang_scale;
          int invSum = 256;
          const int lim = (3 << scale) < W ? (3 << scale) : W;
          for (int x = 0; x < lim; x++) {
            invSum += inv;
            const int wL = 32 >> (2 * x >> scale);
            const int16_t left = refSide[y + (invSum >> 9) + 1];
            row[x] = (int16_t) (row[x] + ((wL * (left - row[x]) + 32) >> 6));
          }
        }
      }
    }
    if (!ver) for (int y = 0; y < H; y++) for (int x = 0; x < W; x++) pred[x * ps + y] = dst[y * ds + x];
  }
  /* PDPC for planar / DC (354-378) */
  if (ip.pdpc && (mode == ORC_PLANAR || mode == ORC_DC)) {
    const int scale = (ilog2(w) - 2 + ilog2(h) - 2 + 2) >> 2;
    for (int y = 0; y < h; y++) {
      int sh = (y << 1) >> scale; if (sh > 31) sh = 31;
      const int wT = 32 >> sh;
      const int16_t left = LEFT(y + 1);
      for (int x = 0; x < w; x++) {
        int shx = (x << 1) >> scale; if (shx > 31) shx = 31;
        const int wL = 32 >> shx;
        const int16_t top = TOP(x + 1), val = pred[y * ps + x];
        pred[y * ps + x] = (int16_t) (val + ((wL * (left - val) + wT * (top - val) + 32) >> 6));
      }
    }
  }
#undef TOP
#undef LEFT
}

/* CL/UnitTools.cpp:508-640 PU::getIntraMPMs given the two neighbour directions */
void orc_get_mpms(int L, int A, unsigned mpm[6])
{
  const int offset = 61, mod = 64;
  mpm[0] = ORC_PLANAR; mpm[1] = ORC_DC; mpm[2] = ORC_VER; mpm[3] = ORC_HOR; mpm[4] = ORC_VER - 4; mpm[5] = ORC_VER + 4;
  if (L == A) {
    if (L > ORC_DC) {
      mpm[0] = ORC_PLANAR; mpm[1] = L;
      mpm[2] = ((L + offset) % mod) + 2; mpm[3] = ((L - 1) % mod) + 2;
      mpm[4] = ((L + offset - 1) % mod) + 2; mpm[5] = (L % mod) + 2;
    }
  } else if (L > ORC_DC && A > ORC_DC) {
    mpm[0] = ORC_PLANAR; mpm[1] = L; mpm[2] = A;
    const int mx = mpm[1] > mpm[2] ? 1 : 2, mn = mpm[1] > mpm[2] ? 2 : 1;
    const int d = (int) mpm[mx] - (int) mpm[mn];
    if (d == 1) { mpm[3] = ((mpm[mn] + offset) % mod) + 2; mpm[4] = ((mpm[mx] - 1) % mod) + 2; mpm[5] = ((mpm[mn] + offset - 1) % mod) + 2; }
    else if (d >= 62) { mpm[3] = ((mpm[mn] - 1) % mod) + 2; mpm[4] = ((mpm[mx] + offset) % mod) + 2; mpm[5] = (mpm[mn] % mod) + 2; }
    else if (d == 2) { mpm[3] = ((mpm[mn] - 1) % mod) + 2; mpm[4] = ((mpm[mn] + offset) % mod) + 2; mpm[5] = ((mpm[mx] - 1) % mod) + 2; }
    else { mpm[3] = ((mpm[mn] + offset) % mod) + 2; mpm[4] = ((mpm[mn] - 1) % mod) + 2; mpm[5] = ((mpm[mx] + offset) % mod) + 2; }
  } else if (L + A >= 2) {
    mpm[0] = ORC_PLANAR; mpm[1] = (unsigned) (L < A ? A : L);
    mpm[2] = ((mpm[1] + offset) % mod) + 2; mpm[3] = ((mpm[1] - 1) % mod) + 2;
    mpm[4] = ((mpm[1] + offset - 1) % mod) + 2; mpm[5] = (mpm[1] % mod) + 2;
  }
}

/* ------------------------------------------------------------------------------------------------
 * Scan order: CL/Rom.cpp:87-131 ScanGenerator (SCAN_DIAG) and 319-370 grouped 4x4 scan over the
 * min(32,w) x min(32,h) region.  Returns number of real scan positions.
 * ---------------------------------------------------------------------------------------------- */
static int diag_scan(int bw, int bh, uint8_t *xs, uint8_t *ys)
{
  int line = 0, col = 0, n = 0;
  for (; n < bw * bh; n++) {
    xs[n] = (uint8_t) col; ys[n] = (uint8_t) line;
    if (col == bw - 1 || line == 0) {
      line += col + 1; col = 0;
      if (line >= bh) { col += line - (bh - 1); line = bh - 1; }
    } else { col++; line--; }
  }
  return n;
}
/* g_log2SbbSize (CL/Rom.cpp:250-261): coefficient-group shape by log2 block size, data */
void orc_cg_shape(int w, int h, int *lcw, int *lch)
{
  const int lw = ilog2(w), lh = ilog2(h);
  if (lw >= 2 && lh >= 2) { *lcw = 2; *lch = 2; return; }
  if (lh == 0) { *lcw = lw > 4 ? 4 : lw; *lch = 0; return; }       /* Nx1 / 1xN blocks of ISP sub-partitions: one row / column of up to 16 */
  if (lw == 0) { *lcw = 0; *lch = lh > 4 ? 4 : lh; return; }
  if (lh == 1) { *lcw = lw >= 3 ? 3 : lw; *lch = 1; if (lw == 2) *lcw = 1; return; }   /* Nx2: {1,1} for 4x2, {3,1} for >=8 */
  /* 2xN (not reachable in 4:2:0 dual tree) */
  *lcw = 1; *lch = lh >= 3 ? 3 : lh; if (lh == 2) *lch = 1;
}
int orc_scan_order(int w, int h, uint16_t *idx)
{
  int lcw, lch; orc_cg_shape(w, h, &lcw, &lch);
  const int zw = w < 32 ? w : 32, zh = h < 32 ? h : 32;
  const int gw = zw >> lcw, gh = zh >> lch, cw = 1 << lcw, ch = 1 << lch;
  uint8_t gx[64], gy[64], ix[16], iy[16];
  diag_scan(gw, gh, gx, gy);
  diag_scan(cw, ch, ix, iy);
  int n = 0;
  for (int g = 0; g < gw * gh; g++)
    for (int i = 0; i < cw * ch; i++) idx[n++] = (uint16_t) ((gy[g] * ch + iy[i]) * w + gx[g] * cw + ix[i]);
  return n;
}

/* ------------------------------------------------------------------------------------------------
 * CCLM (CL/IntraPrediction.cpp): 4:2:0, sps_cclm_colocated_chroma_flag = 0.
 * orc_cclm_luma      = xGetLumaRecPixels 1665-1930: down-sampled reconstructed luma of the chroma block plus, where
 *                      available, one row above / one column left (extended by the available above-right / below-left
 *                      units when mdlm).  tmp origin = tmp[tstride + 1]; info = {leftAvail, aboveAvail, availLeftBelowUnits,
 *                      availAboveRightUnits}; availability on the chroma tree (isLeft/Above/BelowLeft/AboveRightAvailable 1524-1663).
 * orc_cclm_params    = xGetLMParameters 1931-2150 (ref = the block's unfiltered chroma reference samples, layout of
 *                      orc_fill_ref_samples with mrl 0).
 * orc_pred_cclm      = predIntraChromaLM 400-420 (linearTransform with clipping).
 * ---------------------------------------------------------------------------------------------- */
void orc_cclm_luma(const int16_t *recY, int strideY, const uint8_t *avail, int avail_stride, int tag, int pic_wc, int pic_hc,
                   int cx, int cy, int cw, int ch, int mdlm, int info[4], int16_t *tmp, int tstride)
{
  const int unit = 2;                                    /* chroma samples per 4x4 luma unit */
  const int aboveUnits = cw / unit, leftUnits = ch / unit;
  const int totalAbove = (2 * cw + unit - 1) / unit, totalLeft = (2 * ch + unit - 1) / unit;
  const int aboveRightUnits = totalAbove - aboveUnits, leftBelowUnits = totalLeft - leftUnits;
  int n = 0;
  for (int dy = 0; dy < leftUnits * unit; dy += unit) { if (!unit_avail(avail, avail_stride, 1, pic_wc, pic_hc, cx - 1, cy + dy, tag)) break; n++; }
  const int leftAvail = n == leftUnits;
  n = 0;
  for (int dx = 0; dx < aboveUnits * unit; dx += unit) { if (!unit_avail(avail, avail_stride, 1, pic_wc, pic_hc, cx + dx, cy - 1, tag)) break; n++; }
  const int aboveAvail = n == aboveUnits;
  int availLB = 0, availAR = 0;
  if (leftAvail) for (int dy = 0; dy < leftBelowUnits * unit; dy += unit) { if (!unit_avail(avail, avail_stride, 1, pic_wc, pic_hc, cx - 1, cy + ch - 1 + unit + dy, tag)) break; availLB++; }
  if (aboveAvail) for (int dx = 0; dx < aboveRightUnits * unit; dx += unit) { if (!unit_avail(avail, avail_stride, 1, pic_wc, pic_hc, cx + cw - 1 + unit + dx, cy - 1, tag)) break; availAR++; }
  info[0] = leftAvail; info[1] = aboveAvail; info[2] = availLB; info[3] = availAR;
  const int16_t *src0 = recY + (size_t) (2 * cy) * strideY + 2 * cx;
  const int S = strideY, S2 = 2 * strideY;
  int16_t *dst0 = tmp + tstride + 1;
  const int firstRowOfCtu = (cy & 63) == 0;
  if (aboveAvail) {
    int16_t *d = dst0 - tstride;
    const int added = mdlm ? availAR * unit : 0;
    for (int i = 0; i < cw + added; i++) {
      if (firstRowOfCtu) {
        const int16_t *q = src0 - S;
        if (i == 0 && !leftAvail) d[i] = q[2 * i];
        else d[i] = (int16_t) ((q[2 * i] * 2 + q[2 * i - 1] + q[2 * i + 1] + 2) >> 2);
      } else {
        const int16_t *q = src0 - S2;
        if (i == 0 && !leftAvail) d[i] = (int16_t) ((q[2 * i] + q[2 * i + S] + 1) >> 1);
        else d[i] = (int16_t) ((q[2 * i] * 2 + q[2 * i - 1] + q[2 * i + 1] + q[2 * i + S] * 2 + q[2 * i - 1 + S] + q[2 * i + 1 + S] + 4) >> 3);
      }
    }
  }
  if (leftAvail) {
    int16_t *d = dst0 - 1;
    const int16_t *q = src0 - 3;
    const int added = mdlm ? availLB * unit : 0;
    for (int j = 0; j < ch + added; j++) {
      d[0] = (int16_t) ((q[1] * 2 + q[0] + q[2] + q[1 + S] * 2 + q[S] + q[2 + S] + 4) >> 3);
      q += S2; d += tstride;
    }
  }
  for (int j = 0; j < ch; j++) {
    const int16_t *q = src0 + (size_t) j * S2; int16_t *d = dst0 + (size_t) j * tstride;
    for (int i = 0; i < cw; i++) {
      if (i == 0 && !leftAvail) d[i] = (int16_t) ((q[2 * i] + q[2 * i + S] + 1) >> 1);
      else d[i] = (int16_t) ((q[2 * i] * 2 + q[2 * i + 1] + q[2 * i - 1] + q[2 * i + S] * 2 + q[2 * i + 1 + S] + q[2 * i - 1 + S] + 4) >> 3);
    }
  }
}
static int imin(int a, int b) { return a < b ? a : b; }
static int imax(int a, int b) { return a > b ? a : b; }
static int floor_log2(unsigned v) { int r = 0; while (v > 1) { v >>= 1; r++; } return r; }
void orc_cclm_params(const int16_t *tmp, int tstride, const int16_t *ref, int cw, int ch, int mode, const int info[4], int bit_depth, int *pa, int *pb, int *pshift)
{
  const int unit = 2;
  int leftAvail = info[0], aboveAvail = info[1], availLB = info[2], availAR = info[3];
  const int availAbove = aboveAvail ? cw / unit : 0, availLeft = leftAvail ? ch / unit : 0;
  const int16_t *src0 = tmp + tstride + 1;
  const int rstride = 2 * cw + 1;                       /* chroma reference buffer: top[i] = ref[i], left[i] = ref[i * rstride] */
  int actualTop = 0, actualLeft = 0;
  if (mode == 69) { leftAvail = 0; if (availAR > ch / unit) availAR = ch / unit; actualTop = unit * (availAbove + availAR); }           /* MDLM_T */
  else if (mode == 68) { aboveAvail = 0; if (availLB > cw / unit) availLB = cw / unit; actualLeft = unit * (availLeft + availLB); }      /* MDLM_L */
  else { actualTop = cw; actualLeft = ch; }
  const int aboveIs4 = leftAvail ? 0 : 1, leftIs4 = aboveAvail ? 0 : 1;
  const int startT = actualTop >> (2 + aboveIs4), stepT = imax(1, actualTop >> (1 + aboveIs4));
  const int startL = actualLeft >> (2 + leftIs4), stepL = imax(1, actualLeft >> (1 + leftIs4));
  int selL[4] = { 0, 0, 0, 0 }, selC[4] = { 0, 0, 0, 0 };
  int cntT = 0, cntL = 0, cnt = 0;
  if (aboveAvail) {
    cntT = imin(actualTop, (1 + aboveIs4) << 1);
    for (int pos = startT; cnt < cntT; pos += stepT, cnt++) { selL[cnt] = src0[-tstride + pos]; selC[cnt] = ref[1 + pos]; }
  }
  if (leftAvail) {
    cntL = imin(actualLeft, (1 + leftIs4) << 1);
    int pos = startL;
    for (int k = 0; k < cntL; pos += stepL, k++) { selL[k + cntT] = src0[pos * tstride - 1]; selC[k + cntT] = ref[(1 + pos) * rstride]; }
  }
  cnt = cntL + cntT;
  if (cnt == 2) {
    selL[3] = selL[0]; selC[3] = selC[0]; selL[2] = selL[1]; selC[2] = selC[1];
    selL[0] = selL[1]; selC[0] = selC[1]; selL[1] = selL[3]; selC[1] = selC[3];
  }
  int minG[2] = { 0, 2 }, maxG[2] = { 1, 3 };
  int *tmin = minG, *tmax = maxG;
#define SWAPI(a_, b_) { int t_ = (a_); (a_) = (b_); (b_) = t_; }
  if (selL[tmin[0]] > selL[tmin[1]]) SWAPI(tmin[0], tmin[1]);
  if (selL[tmax[0]] > selL[tmax[1]]) SWAPI(tmax[0], tmax[1]);
  if (selL[tmin[0]] > selL[tmax[1]]) { int *t_ = tmin; tmin = tmax; tmax = t_; }
  if (selL[tmin[1]] > selL[tmax[0]]) SWAPI(tmin[1], tmax[0]);
#undef SWAPI
  const int minL = (selL[tmin[0]] + selL[tmin[1]] + 1) >> 1, minC = (selC[tmin[0]] + selC[tmin[1]] + 1) >> 1;
  const int maxL = (selL[tmax[0]] + selL[tmax[1]] + 1) >> 1, maxC = (selC[tmax[0]] + selC[tmax[1]] + 1) >> 1;
  int a, b, shift;
  if (leftAvail || aboveAvail) {
    const int diff = maxL - minL;
    if (diff > 0) {
      static const uint8_t DivSigTable[16] = { 0, 7, 6, 5, 5, 4, 4, 3, 3, 2, 2, 1, 1, 1, 1, 0 };
      const int diffC = maxC - minC;
      int x = floor_log2((unsigned) diff);
      const int normDiff = ((diff << 4) >> x) & 15;
      const int v = DivSigTable[normDiff] | 8;
      x += normDiff != 0;
      const int y = floor_log2((unsigned) abs(diffC)) + 1;     /* floorLog2(0) = -1 in the reference: see below */
      const int yy = diffC == 0 ? 0 : y;
      const int add = (1 << yy) >> 1;
      a = (diffC * v + add) >> yy;
      shift = 3 + x - yy;
      if (shift < 1) { shift = 1; a = a == 0 ? 0 : a < 0 ? -15 : 15; }
      b = minC - ((a * minL) >> shift);
    } else { a = 0; b = minC; shift = 0; }
  } else { a = 0; b = 1 << (bit_depth - 1); shift = 0; }
  *pa = a; *pb = b; *pshift = shift;
}
void orc_pred_cclm(const int16_t *tmp, int tstride, int a, int b, int shift, int bit_depth, int cw, int ch, int16_t *pred, int pstride)
{
  const int16_t *src0 = tmp + tstride + 1;
  const int mx = (1 << bit_depth) - 1;
  for (int j = 0; j < ch; j++) for (int i = 0; i < cw; i++) {
    int v = ((a * src0[j * tstride + i]) >> shift) + b;           /* AreaBuf::linearTransform, CL/Buffer.cpp:699 */
    pred[j * pstride + i] = (int16_t) (v < 0 ? 0 : v > mx ? mx : v);
  }
}

/* test hook: the arithmetic coder over a sequence of operations from the I-slice contexts at qp (see ref_arith_encode in
 * ref_harness.cpp): ops[i] = {kind, a, b}: 0 = context bin (ctx a, bin b); 1 = b bypass bins of value a; 2 = terminating bin a */
int orc_arith_encode(int qp, const int32_t *ops, int nops, uint8_t *out, int cap)
{
  orc_cabac c; memset(&c, 0, sizeof c);
  orc_arith aw; memset(&aw, 0, sizeof aw);
  aw.out = out; aw.cap = (size_t) cap;
  orc_ctx_init(qp, c.s0, c.s1);
  orc_arith_start(&aw);
  c.aw = &aw;
  for (int i = 0; i < nops; i++) {
    const int32_t *o = ops + 3 * i;
    if (o[0] == 0) orc_enc_bin(&c, (unsigned) o[2], o[1]);
    else if (o[0] == 1) orc_enc_bins_ep(&c, (uint32_t) o[1], o[2]);
    else orc_arith_trm(&aw, (unsigned) o[1]);
  }
  orc_arith_finish(&aw);
  orc_bs_write(&aw, 1, 1); while (aw.bit_n) orc_bs_write(&aw, 0, 1);
  return aw.n > aw.cap ? -1 : (int) aw.n;
}

/* one block of a CU with lfnstIdx through TrQuant::transformNxN / invTransformNxN (CL/TrQuant.cpp:1127-1235, 563-610): primary transform with the
 * LFNST zero-out, xFwdLfnst, the quantiser, then dequantisation, xInvLfnst and the inverse transform.  Returns absSum (resi_out only when > 0). */
int orc_trquant_lfnst(const uint16_t *s0, const uint16_t *s1, const int16_t *resi, int w, int h, int comp, int cbf_cb, int bit_depth, int qp, double lambda,
                      int dep_quant, int dir, int lfnst_idx, int16_t *level, int16_t *resi_out)
{
  int *coef = (int *) malloc(sizeof(int) * (size_t) w * h);
  const int mode = orc_lfnst_mode(dir, w, h);
  orc_fwd_2d_mts(resi, w, w, h, bit_depth, 0, coef);
  if (lfnst_idx) { orc_lfnst_keep(coef, w, h); orc_fwd_lfnst(coef, w, h, mode, lfnst_idx); }
  int abs_sum;
  if (dep_quant) abs_sum = orc_depquant(s0, s1, coef, w, h, comp, ORC_CTX_QtCbf[comp] + (comp == 2 ? cbf_cb : 0), bit_depth, qp, lambda, 0, lfnst_idx, level);
  else abs_sum = lfnst_idx ? orc_quant_lfnst(coef, w, h, bit_depth, qp, level) : orc_quant(coef, w, h, bit_depth, qp, level);
  if (abs_sum > 0) {
    if (dep_quant) orc_dequant_dq(level, w, h, bit_depth, qp, coef); else orc_dequant(level, w, h, bit_depth, qp, coef);
    orc_inv_lfnst(coef, w, h, mode, lfnst_idx);
    orc_inv_2d_mts(coef, w, h, bit_depth, 0, resi_out, w);
  }
  free(coef);
  return abs_sum;
}

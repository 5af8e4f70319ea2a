/*
 * orc_ts.c — transform skip of a luma block (TEST INFRASTRUCTURE ONLY, see vvc_oracle.h).
 * Restates
 *   CL/TrQuant.cpp       xTransformSkip 1394-1440, xITransformSkip 996-1041, the TS entry of the candidate pruning 1049-1124
 *   CL/QuantRDOQ.cpp     quant 521-576 (transform-skip luma blocks reach xRateDistOptQuantTS also with DepQuant on: CL/DepQuant.cpp:1755-1781),
 *                        xRateDistOptQuantTS 1243-1483, xGetCodedLevelTSPred 1773-1836, xGetICRateTS 1898-1976, xGetErrScaleCoeff 383-392
 *   CL/Quant.cpp         dequant 423-549 with isTransformSkip (no sqrt(2) adjustment, JVET_O0919 minimum TS QP through QpParam 60-130)
 *   CL/ContextModelling.h sigCtxIdAbsTS 197, lrg1CtxIdAbsTS 221, signCtxIdAbsTS 252, neighTS 291, deriveModCoeff 310, templateAbsSumTS 351
 *   EL/CABACWriter.cpp   residual_codingTS 4306-4333, residual_coding_subblockTS 4335-4555 (JVET_O0122 / O0409 / O0619 forms)
 * BDPCM is off in the reference cfg (m_bdpcm = 0 everywhere).
 */
#include "orc_internal.h"
#include <stdlib.h>
#include <math.h>
#include <limits.h>

static int ilog2(int v) { int r = 0; while (v > 1) { v >>= 1; r++; } return r; }
static int imin(int a, int b) { return a < b ? a : b; }
static int imax(int a, int b) { return a > b ? a : b; }

/* getTransformShift without the sqrt(2) adjustment (TU::needsSqrt2Scale is false for transform skip, CL/UnitTools.cpp:4650-4655) */
static int ts_shift(int w, int h, int bd) { return 15 - bd - ((ilog2(w) + ilog2(h)) >> 1); }
/* QpParam::Qp(isTransformSkip) (CL/Quant.cpp:116-128): max(QP', 4 + min_qp_prime_ts_minus4); the encoder sets the SPS value to
 * 6 * (internal - input bit depth) = 0 (EL/EncLib.cpp:1157) */
int orc_ts_qp(int qp) { return imax(qp, 4); }

void orc_ts_fwd(const int16_t *resi, int stride, int w, int h, int bit_depth, int *coef)
{
  const int sh = ts_shift(w, h, bit_depth);
  for (int y = 0; y < h; y++) for (int x = 0; x < w; x++) {
    const int r = resi[y * stride + x];
    coef[y * w + x] = sh >= 0 ? r * (1 << sh) : (r + (1 << (-sh - 1))) >> -sh;
  }
}
void orc_ts_inv(const int *coef, int w, int h, int bit_depth, int16_t *resi, int stride)
{
  const int sh = ts_shift(w, h, bit_depth);
  for (int y = 0; y < h; y++) for (int x = 0; x < w; x++) {
    const int c = coef[y * w + x];
    resi[y * stride + x] = (int16_t) (sh >= 0 ? (c + (sh == 0 ? 0 : 1 << (sh - 1))) >> sh : c * (1 << -sh));
  }
}
/* the measure TrQuant::transformNxN (1049-1124) prunes the transform-skip candidate with: int(sum |coef| * scaleSAD) against the DCT-II sum */
int orc_ts_sumabs(const int16_t *resi, int stride, int w, int h, int bit_depth)
{
  const int sh = ts_shift(w, h, bit_depth);
  int sum = 0;
  for (int y = 0; y < h; y++) for (int x = 0; x < w; x++) { const int r = resi[y * stride + x]; sum += abs(sh >= 0 ? r * (1 << sh) : (r + (1 << (-sh - 1))) >> -sh); }
  double scale = 1.0;
  if ((ilog2(w) + ilog2(h)) & 1) scale = 1.0 / 1.414213562;
  return (int) (sum * scale);
}

/* Quant::dequant of a transform-skip block */
void orc_dequant_ts(const int16_t *level, int w, int h, int bit_depth, int qp, int *coef)
{
  const int q = orc_ts_qp(qp);
  const int scale = ORC_INV_QUANT_SCALES[q % 6];
  const int right_shift = 6 - (ts_shift(w, h, bit_depth) + q / 6);
  int tbd = 32 + right_shift - 7; if (tbd > 16) tbd = 16;
  const int in_min = -(1 << (tbd - 1)), in_max = (1 << (tbd - 1)) - 1;
  for (int i = 0; i < w * h; i++) {
    int l = level[i]; l = l < in_min ? in_min : l > in_max ? in_max : l;
    int v;
    if (right_shift > 0) v = (l * scale + (1 << (right_shift - 1))) >> right_shift;
    else v = (l * scale) * (1 << -right_shift);
    coef[i] = v < -32768 ? -32768 : v > 32767 ? 32767 : v;
  }
}

/* scan geometry of a block (CoeffCodingContext, CL/ContextModelling.cpp:40-134) */
typedef struct { int w, h, lcg, wg, hg, n; uint16_t scan[1024]; uint8_t cgx[64], cgy[64]; uint8_t sig[64]; } ts_geo;
static void diag(int bw, int bh, uint8_t *xs, uint8_t *ys)
{
  int line = 0, col = 0;
  for (int n = 0; n < bw * bh; n++) {
    xs[n] = (uint8_t) col; ys[n] = (uint8_t) line;
    if (col == bw - 1 || line == 0) { line += col + 1; col = 0; if (line >= bh) { col += line - (bh - 1); line = bh - 1; } }
    else { col++; line--; }
  }
}
static void geo_init(ts_geo *g, int w, int h)
{
  int lcw, lch; orc_cg_shape(w, h, &lcw, &lch);
  memset(g, 0, sizeof *g);
  g->w = w; g->h = h; g->lcg = lcw + lch; g->wg = w >> lcw; g->hg = h >> lch;
  g->n = orc_scan_order(w, h, g->scan);
  diag(g->wg, g->hg, g->cgx, g->cgy);
}
static inline const uint32_t *frac_of(const uint16_t *s0, const uint16_t *s1, int ctx) { return &ORC_BIN_FRAC_BITS[((unsigned) (s0[ctx] + s1[ctx]) >> 8) * 2]; }
/* neighTS + the context increments derived from the left / above levels */
static inline void neigh(const int *lv, int w, int blk, int *right, int *below)
{
  const int y = blk / w, x = blk - y * w;
  *right = x > 0 ? lv[blk - 1] : 0; *below = y > 0 ? lv[blk - w] : 0;
}
static inline int mod_coeff(int right, int below, int a)       /* deriveModCoeff, bdpcm 0 */
{
  const int p = imax(abs(below), abs(right));
  return a == p ? 1 : (a < p ? a + 1 : a);
}
static inline int sign_ctx(int right, int below)
{
  if ((right == 0 && below == 0) || (right * below) < 0) return 0;
  return (right >= 0 && below >= 0) ? 1 : 2;
}
static const uint8_t TS_RICE[32] = { 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 2, 2, 2, 2, 2, 2, 2 };

/* xGetICRateTS: bits (2^-15 units) of a level whose modified magnitude is a */
static int ic_rate_ts(const uint16_t *s0, const uint16_t *s1, unsigned a, const uint32_t *fPar, const uint32_t *fSign, const uint32_t *fGt1, int sign, int rice)
{
  int rate = (int) fSign[sign];
  if (a > 1) {
    rate += (int) fGt1[1];
    rate += (int) fPar[(a - 2) & 1];
    unsigned cutoff = 2;
    for (int i = 0; i < 4; i++) {
      if (a >= cutoff) rate += (int) frac_of(s0, s1, ORC_CTX_TsGtxFlag + (int) (cutoff >> 1))[a >= cutoff + 2];
      cutoff += 2;
    }
    if (a >= cutoff) {
      unsigned symbol = (a - cutoff) >> 1, length;
      if (symbol < (5u << rice)) { length = symbol >> rice; rate += (int) ((length + 1 + (unsigned) rice) << 15); }
      else {                                            /* useLimitedPrefixLength = extended_precision_processing_flag = 0 */
        length = (unsigned) rice; symbol -= 5u << rice;
        while (symbol >= (1u << length)) symbol -= 1u << (length++);
        rate += (int) ((5 + length + 1 - (unsigned) rice + length) << 15);
      }
    }
  } else if (a == 1) rate += (int) fGt1[0];
  else rate = 0;
  return rate;
}

/* xRateDistOptQuantTS.  s0 / s1: the estimator's contexts (rates are read from them, they are not updated); coef: the transform-skip
 * "coefficients"; qp: QP' of the block (incl. QpBDOffset); lambda: the quantiser's lambda.  Returns absSum. */
int orc_rdoq_ts(const uint16_t *s0, const uint16_t *s1, const int *coef, int w, int h, int bit_depth, int qp, double lambda, int16_t *level)
{
  const int q = orc_ts_qp(qp);
  const int tshift = ts_shift(w, h, bit_depth);
  const int qBits = 14 + q / 6 + tshift;
  const int qc = ORC_QUANT_SCALES[q % 6];
  double errorScale;
  { double dErrScale = (double) (1 << 15); const double dTransShift = (double) tshift; dErrScale = dErrScale * pow(2.0, (-2.0 * dTransShift)); errorScale = dErrScale / qc / qc / (1 << 0); }
  const int ecMax = (1 << 15) - 1;
  ts_geo g; geo_init(&g, w, h);
  static int dst[1024]; static double costCoeff[1024], costSig[1024], costCoeff0[1024];
  const int n = w * h, sbSizeM1 = (1 << g.lcg) - 1, sbNum = n >> g.lcg;
  memset(dst, 0, sizeof(int) * (size_t) n); memset(costCoeff, 0, sizeof(double) * (size_t) n); memset(costSig, 0, sizeof(double) * (size_t) n);
  int anySigCG = 0;
  const uint32_t *fPar = frac_of(s0, s1, ORC_CTX_TsParFlag);
  for (int sb = 0; sb < sbNum; sb++) {
    const int cgx = g.cgx[sb], cgy = g.cgy[sb], cgPos = cgy * g.wg + cgx;
    const int sigLeft = cgx > 0 ? g.sig[cgPos - 1] : 0, sigAbove = cgy > 0 ? g.sig[cgPos - g.wg] : 0;
    const uint32_t *fGrp = frac_of(s0, s1, ORC_CTX_TsSigCoeffGroup + sigLeft + sigAbove);
    int noCoeffCoded = 0; double baseCost = 0.0;
    double sigCost = 0, codedLevelandDist = 0, uncodedDist = 0;
    for (int k = 0; k <= sbSizeM1; k++) {
      const int sp = (sb << g.lcg) + k, blk = g.scan[sp];
      const int64_t tmpLevel = (int64_t) abs(coef[blk]) * qc;
      const int64_t cap = (int64_t) INT_MAX - ((int64_t) 1 << (qBits - 1));
      const int64_t levelDouble = tmpLevel < cap ? tmpLevel : cap;
      unsigned lv[3]; int tested = 0;
      const unsigned roundAbs = (unsigned) imin(ecMax, (int) ((levelDouble + ((int64_t) 1 << (qBits - 1))) >> qBits));
      const unsigned minAbs = roundAbs > 1 ? roundAbs - 1 : 1;
      const unsigned downAbs = (unsigned) imin(ecMax, (int) (levelDouble >> qBits)), upAbs = (unsigned) imin(ecMax, (int) downAbs + 1);
      lv[tested++] = roundAbs;
      if (minAbs != roundAbs) lv[tested++] = minAbs;
      int right, below; neigh(dst, w, blk, &right, &below);
      if (upAbs != roundAbs && upAbs != minAbs && mod_coeff(right, below, (int) upAbs) == 1) lv[tested++] = upAbs;
      { const double dErr = (double) levelDouble; costCoeff0[sp] = dErr * dErr * errorScale; }
      dst[blk] = (int) lv[0];
      const int numPos = (right != 0) + (below != 0);
      const uint32_t *fSig = frac_of(s0, s1, ORC_CTX_TsSigFlag + numPos);
      const int rice = TS_RICE[imin(abs(right) + abs(below), 31)];
      const uint32_t *fSign = frac_of(s0, s1, ORC_CTX_TsResidualSign + sign_ctx(right, below));
      const uint32_t *fGt1 = frac_of(s0, s1, ORC_CTX_TsLrg1Flag + numPos);
      const int sign = coef[blk] < 0;
      const int isLast = k == sbSizeM1 && noCoeffCoded == 0;
      /* xGetCodedLevelTSPred */
      unsigned best = 0; double currCostSig = 0; int done = 0;
      if (!isLast && lv[0] < 3) {
        costSig[sp] = lambda * (double) fSig[0];
        costCoeff[sp] = costCoeff0[sp] + costSig[sp];
        if (lv[0] == 0) done = 1;
      } else costCoeff[sp] = ORC_MAX_DOUBLE;
      if (!done) {
        if (!isLast) currCostSig = lambda * (double) fSig[1];
        for (int e = 1; e <= tested; e++) {
          const unsigned a = lv[e - 1];
          const double dErr = (double) (levelDouble - ((int64_t) a << qBits));
          const double err = dErr * dErr * errorScale;
          const int m = mod_coeff(right, below, (int) a);
          double cur = err + lambda * (double) ic_rate_ts(s0, s1, (unsigned) m, fPar, fSign, fGt1, sign, rice);
          cur += currCostSig;
          if (cur < costCoeff[sp]) { best = a; costCoeff[sp] = cur; costSig[sp] = currCostSig; }
        }
      }
      if (best > 0) noCoeffCoded++;
      dst[blk] = (best != 0 && coef[blk] < 0) ? -(int) best : (int) best;
      baseCost += costCoeff[sp];
      sigCost += costSig[sp];
      if (dst[blk]) { g.sig[cgPos] = 1; codedLevelandDist += costCoeff[sp] - costSig[sp]; uncodedDist += costCoeff0[sp]; }
    }
    if (!g.sig[cgPos]) { baseCost += lambda * (double) fGrp[0] - sigCost; }
    else if (sb != sbNum - 1 || anySigCG) {
      double costZeroSB = baseCost;
      baseCost += lambda * (double) fGrp[1];
      costZeroSB += lambda * (double) fGrp[0];
      costZeroSB += uncodedDist;
      costZeroSB -= codedLevelandDist;
      costZeroSB -= sigCost;
      if (costZeroSB < baseCost) {
        g.sig[cgPos] = 0; baseCost = costZeroSB;
        for (int k = 0; k <= sbSizeM1; k++) { const int sp = (sb << g.lcg) + k, blk = g.scan[sp]; if (dst[blk]) { dst[blk] = 0; costCoeff[sp] = costCoeff0[sp]; costSig[sp] = 0; } }
      } else anySigCG = 1;
    }
  }
  int absSum = 0;
  for (int i = 0; i < n; i++) { absSum += abs(dst[i]); level[i] = (int16_t) dst[i]; }
  return absSum;
}

/* residual_codingTS + residual_coding_subblockTS on the estimator / writer */
void orc_residual_coding_ts(orc_cabac *cb, const int16_t *coeff, int w, int h)
{
  ts_geo g; geo_init(&g, w, h);
  static int lv[1024];
  const int n = w * h, cgSize = 1 << g.lcg, nsub = ((n - 1) >> g.lcg) + 1;
  for (int i = 0; i < n; i++) lv[i] = coeff[i];
  int remBins = 2 * w * h;                    /* setNumCtxBins; isContextCoded() = --remaining >= 0 */
#define CTX_CODED() (--remBins >= 0)
  uint8_t sigGroupFlags[64] = { 0 };
  for (int sp = 0; sp < n; sp++) if (lv[g.scan[sp]]) sigGroupFlags[sp >> g.lcg] = 1;
  int nSet = 0;                               /* m_sigCoeffGroupFlag.count() */
  for (int sub = 0; sub < nsub; sub++) {
    const int cgx = g.cgx[sub], cgy = g.cgy[sub], cgPos = cgy * g.wg + cgx;
    if (sigGroupFlags[sub] && !g.sig[cgPos]) { g.sig[cgPos] = 1; nSet++; }
    const int sigLeft = cgx > 0 ? g.sig[cgPos - 1] : 0, sigAbove = cgy > 0 ? g.sig[cgPos - g.wg] : 0;
    const int grpCtx = ORC_CTX_TsSigCoeffGroup + sigLeft + sigAbove;
    const int minSub = sub << g.lcg, maxSub = minSub + cgSize - 1;
    const int isLastSubSet = sub == nsub - 1;
    const int only1st = nSet - g.sig[nsub - 1] == 0;      /* only1stSigGroup: the last sub-block's raster position is its scan index */
    if (!isLastSubSet || !only1st) {
      if (g.sig[cgPos]) orc_enc_bin(cb, 1, grpCtx);
      else { orc_enc_bin(cb, 0, grpCtx); continue; }
    }
    int numNonZero = 0;
    for (int sp = minSub; sp <= maxSub; sp++) {
      const int blk = g.scan[sp], cf = lv[blk];
      const unsigned sigFlag = cf != 0;
      int right, below; neigh(lv, w, blk, &right, &below);
      const int numPos = (right != 0) + (below != 0);
      if (numNonZero || sp != maxSub) {
        if (CTX_CODED()) orc_enc_bin(cb, sigFlag, ORC_CTX_TsSigFlag + numPos); else orc_enc_bins_ep(cb, sigFlag, 1);
      }
      if (sigFlag) {
        const unsigned sign = cf < 0;
        if (CTX_CODED()) orc_enc_bin(cb, sign, ORC_CTX_TsResidualSign + sign_ctx(right, below)); else orc_enc_bins_ep(cb, sign, 1);
        numNonZero++;
        int rem = mod_coeff(right, below, abs(cf)) - 1;
        const unsigned gt1 = !!rem;
        if (CTX_CODED()) orc_enc_bin(cb, gt1, ORC_CTX_TsLrg1Flag + numPos); else orc_enc_bins_ep(cb, gt1, 1);
        if (gt1) {
          rem -= 1;
          if (CTX_CODED()) orc_enc_bin(cb, (unsigned) (rem & 1), ORC_CTX_TsParFlag); else orc_enc_bins_ep(cb, (unsigned) (rem & 1), 1);
        }
      }
    }
    for (int sp = minSub; sp <= maxSub; sp++) {
      const int blk = g.scan[sp];
      int right, below; neigh(lv, w, blk, &right, &below);
      const unsigned a = (unsigned) mod_coeff(right, below, abs(lv[blk]));
      unsigned cutoff = 2;
      for (int i = 0; i < 4; i++) {
        if (a >= cutoff) {
          const unsigned gt2 = a >= cutoff + 2;
          if (CTX_CODED()) orc_enc_bin(cb, gt2, ORC_CTX_TsGtxFlag + (int) (cutoff >> 1)); else orc_enc_bins_ep(cb, gt2, 1);
        }
        cutoff += 2;
      }
    }
    for (int sp = minSub; sp <= maxSub; sp++) {
      const int blk = g.scan[sp];
      int right, below; neigh(lv, w, blk, &right, &below);
      const unsigned a = (unsigned) mod_coeff(right, below, abs(lv[blk]));
      if (a >= 10) orc_enc_rem_abs(cb, (a - 10) >> 1, TS_RICE[imin(abs(right) + abs(below), 31)]);
    }
  }
#undef CTX_CODED
}

/* test entry point (tests/golden ts.npz): one luma block through transform skip, RDOQ-TS from the given contexts, dequantisation and the inverse:
 * levels, absSum (return value), reconstructed residual, and whether the pruning of TrQuant::transformNxN keeps the TS candidate */
int orc_trquant_ts(const uint16_t *s0, const uint16_t *s1, const int16_t *resi, int w, int h, int bit_depth, int qp, double lambda, int16_t *level, int16_t *resi_out, int *keep)
{
  static int coef[1024], dct[1024];
  orc_ts_fwd(resi, w, w, h, bit_depth, coef);
  orc_fwd_2d(resi, w, w, h, bit_depth, dct);
  int sum0 = 0; for (int i = 0; i < w * h; i++) sum0 += abs(dct[i]);
  *keep = (double) orc_ts_sumabs(resi, w, w, h, bit_depth) <= (double) sum0;
  const int abs_sum = orc_rdoq_ts(s0, s1, coef, w, h, bit_depth, qp, lambda, level);
  if (abs_sum > 0) { orc_dequant_ts(level, w, h, bit_depth, qp, coef); orc_ts_inv(coef, w, h, bit_depth, resi_out, w); }
  return abs_sum;
}
uint64_t orc_residual_bits_ts(uint16_t *s0, uint16_t *s1, const int16_t *level, int w, int h)
{
  orc_cabac c; c.aw = 0; c.dq = 1;
  memcpy(c.s0, s0, sizeof c.s0); memcpy(c.s1, s1, sizeof c.s1); c.bits = 0;
  orc_residual_coding_ts(&c, level, w, h);
  memcpy(s0, c.s0, sizeof c.s0); memcpy(s1, c.s1, sizeof c.s1);
  return c.bits;
}

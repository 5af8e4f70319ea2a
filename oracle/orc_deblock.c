/*
 * orc_deblock.c — CPU restatement of the deblocking filter for the pictures this oracle codes: intra CUs of a dual-tree
 * I slice, one transform unit per CU, no ISP / sub-block / PCM / palette / lossless / LADF / LMCS / virtual boundaries, filtering across
 * tile boundaries enabled (the cfg's defaults).  TEST INFRASTRUCTURE ONLY (see vvc_oracle.h).
 *
 * Follows CL/LoopFilter.cpp: loopFilterPic 153-262 (all vertical edges of the picture, then all horizontal edges), xDeblockCU 269-428
 * (here: the left / top edge of every CU, boundary strength 2 = intra on both sides, xGetBoundaryStrengthSingle 701-720),
 * xSetMaxFilterLengthPQFromTransformSizes 474-578, xEdgeFilterLuma 892-1184, xEdgeFilterChroma 1186-1434, xFilteringPandQ / xBilinearFilter
 * 1436-1529, xPelFilterLuma 1531-1629, xPelFilterChroma 1631-1688, xUseStrongFiltering 1690-1733, tables 67-82 (JVET_O0159 10-bit tc).
 * Pinned against the reference's LoopFilter (oracle/_ref, tests/golden/deblock.npz).
 */
#include <stdlib.h>
#include <string.h>
#include "orc_internal.h"

static const uint16_t TC_TABLE[66] = { 0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,3,4,4,4,4,5,5,5,5,7,7,8,9,10,10,11,13,14,15,17,19,21,24,25,29,33,36,41,45,51,57,64,71,80,89,100,112,125,141,157,177,198,222,250,280,314,352,395 };
static const uint8_t BETA_TABLE[64] = { 0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,6,7,8,9,10,11,12,13,14,15,16,17,18,20,22,24,26,28,30,32,34,36,38,40,42,44,46,48,50,52,54,56,58,60,62,64,66,68,70,72,74,76,78,80,82,84,86,88 };

static int clip3(int lo, int hi, int v) { return v < lo ? lo : v > hi ? hi : v; }
static int iabs_(int v) { return v < 0 ? -v : v; }
static int calc_dp(const int16_t *s, int o) { return iabs_(s[-o * 3] - 2 * s[-o * 2] + s[-o]); }
static int calc_dq(const int16_t *s, int o) { return iabs_(s[0] - 2 * s[o] + s[o * 2]); }

static int use_strong(const int16_t *s, int o, int d, int beta, int tc, int pLarge, int qLarge, int lenP, int lenQ)
{
  const int m4 = s[0], m3 = s[-o], m7 = s[o * 3], m0 = s[-o * 4];
  int sp3 = iabs_(m0 - m3), sq3 = iabs_(m7 - m4);
  if (pLarge || qLarge) {
    const int mP4 = lenP == 5 ? s[-o * 6] : s[-o * 8], m11 = lenQ == 5 ? s[o * 5] : s[o * 7];
    if (pLarge) sp3 = (sp3 + iabs_(m0 - mP4) + 1) >> 1;
    if (qLarge) sq3 = (sq3 + iabs_(m11 - m7) + 1) >> 1;
    return ((sp3 + sq3) < (beta * 3 >> 5)) && (d < (beta >> 2)) && (iabs_(m3 - m4) < ((tc * 5 + 1) >> 1));
  }
  return ((sp3 + sq3) < (beta >> 3)) && (d < (beta >> 2)) && (iabs_(m3 - m4) < ((tc * 5 + 1) >> 1));
}
/* xFilteringPandQ + xBilinearFilter: long filters (7 / 5 / 3 samples per side) */
static void filter_long(int16_t *src, int o, int nP, int nQ, int tc)
{
  static const int c7[7] = { 59, 50, 41, 32, 23, 14, 5 }, c3[3] = { 53, 32, 11 }, c5[5] = { 58, 45, 32, 19, 6 };
  static const int t7[7] = { 6, 5, 4, 3, 2, 1, 1 }, t3[3] = { 6, 4, 2 };
  int16_t *sP = src - o, *sQ = src;
  const int *cP = nP == 7 ? c7 : nP == 5 ? c5 : c3, *cQ = nQ == 7 ? c7 : nQ == 5 ? c5 : c3;
  const int refP = nP == 7 ? (sP[-6 * o] + sP[-7 * o] + 1) >> 1 : nP == 3 ? (sP[-2 * o] + sP[-3 * o] + 1) >> 1 : (sP[-4 * o] + sP[-5 * o] + 1) >> 1;
  const int refQ = nQ == 7 ? (sQ[6 * o] + sQ[7 * o] + 1) >> 1 : nQ == 3 ? (sQ[2 * o] + sQ[3 * o] + 1) >> 1 : (sQ[4 * o] + sQ[5 * o] + 1) >> 1;
  int mid;
  if (nP == nQ) {
    if (nP == 5) mid = (2 * (sP[0] + sQ[0] + sP[-o] + sQ[o] + sP[-2 * o] + sQ[2 * o]) + sP[-3 * o] + sQ[3 * o] + sP[-4 * o] + sQ[4 * o] + 8) >> 4;
    else mid = (2 * (sP[0] + sQ[0]) + sP[-o] + sQ[o] + sP[-2 * o] + sQ[2 * o] + sP[-3 * o] + sQ[3 * o] + sP[-4 * o] + sQ[4 * o] + sP[-5 * o] + sQ[5 * o] + sP[-6 * o] + sQ[6 * o] + 8) >> 4;
  } else {
    const int16_t *pt = sP, *qt = sQ; int oP = -o, oQ = o, nq = nQ, np = nP;
    if (nQ > nP) { pt = sQ; qt = sP; oP = o; oQ = -o; nq = nP; np = nQ; }
    if (np == 7 && nq == 5) mid = (2 * (sP[0] + sQ[0] + sP[-o] + sQ[o]) + sP[-2 * o] + sQ[2 * o] + sP[-3 * o] + sQ[3 * o] + sP[-4 * o] + sQ[4 * o] + sP[-5 * o] + sQ[5 * o] + 8) >> 4;
    else if (np == 7 && nq == 3) mid = (2 * (pt[0] + qt[0]) + qt[0] + 2 * (qt[oQ] + qt[2 * oQ]) + pt[oP] + qt[oQ] + pt[2 * oP] + pt[3 * oP] + pt[4 * oP] + pt[5 * oP] + pt[6 * oP] + 8) >> 4;
    else mid = (sP[0] + sQ[0] + sP[-o] + sQ[o] + sP[-2 * o] + sQ[2 * o] + sP[-3 * o] + sQ[3 * o] + 4) >> 3;
  }
  const int *tP = nP == 3 ? t3 : t7, *tQ = nQ == 3 ? t3 : t7;
  for (int k = 0; k < nP; k++) { const int s = sP[-o * k], cv = (tc * tP[k]) >> 1; sP[-o * k] = (int16_t) clip3(s - cv, s + cv, (mid * cP[k] + refP * (64 - cP[k]) + 32) >> 6); }
  for (int k = 0; k < nQ; k++) { const int s = sQ[o * k], cv = (tc * tQ[k]) >> 1; sQ[o * k] = (int16_t) clip3(s - cv, s + cv, (mid * cQ[k] + refQ * (64 - cQ[k]) + 32) >> 6); }
}
static void pel_filter_luma(int16_t *s, int o, int tc, int sw, int thrCut, int secondP, int secondQ, int mx, int pLarge, int qLarge, int lenP, int lenQ)
{
  const int m4 = s[0], m3 = s[-o], m5 = s[o], m2 = s[-o * 2], m6 = s[o * 2], m1 = s[-o * 3], m7 = s[o * 3], m0 = s[-o * 4];
  if (sw) {
    if (pLarge || qLarge) { filter_long(s, o, pLarge ? lenP : 3, qLarge ? lenQ : 3, tc); return; }
    s[-o]     = (int16_t) clip3(m3 - 3 * tc, m3 + 3 * tc, (m1 + 2 * m2 + 2 * m3 + 2 * m4 + m5 + 4) >> 3);
    s[0]      = (int16_t) clip3(m4 - 3 * tc, m4 + 3 * tc, (m2 + 2 * m3 + 2 * m4 + 2 * m5 + m6 + 4) >> 3);
    s[-o * 2] = (int16_t) clip3(m2 - 2 * tc, m2 + 2 * tc, (m1 + m2 + m3 + m4 + 2) >> 2);
    s[o]      = (int16_t) clip3(m5 - 2 * tc, m5 + 2 * tc, (m3 + m4 + m5 + m6 + 2) >> 2);
    s[-o * 3] = (int16_t) clip3(m1 - tc, m1 + tc, (2 * m0 + 3 * m1 + m2 + m3 + m4 + 4) >> 3);
    s[o * 2]  = (int16_t) clip3(m6 - tc, m6 + tc, (m3 + m4 + m5 + 3 * m6 + 2 * m7 + 4) >> 3);
    return;
  }
  int delta = (9 * (m4 - m3) - 3 * (m5 - m2) + 8) >> 4;
  if (iabs_(delta) < thrCut) {
    delta = clip3(-tc, tc, delta);
    s[-o] = (int16_t) clip3(0, mx, m3 + delta);
    s[0]  = (int16_t) clip3(0, mx, m4 - delta);
    const int tc2 = tc >> 1;
    if (secondP) s[-o * 2] = (int16_t) clip3(0, mx, m2 + clip3(-tc2, tc2, ((((m1 + m3 + 1) >> 1) - m2 + delta) >> 1)));
    if (secondQ) s[o]      = (int16_t) clip3(0, mx, m5 + clip3(-tc2, tc2, ((((m6 + m4 + 1) >> 1) - m5 - delta) >> 1)));
  }
}
static void pel_filter_chroma(int16_t *s, int o, int tc, int sw, int mx)
{
  const int m0 = s[-o * 4], m1 = s[-o * 3], m2 = s[-o * 2], m3 = s[-o], m4 = s[0], m5 = s[o], m6 = s[o * 2], m7 = s[o * 3];
  if (sw) {
    s[-o * 3] = (int16_t) clip3(m1 - tc, m1 + tc, (3 * m0 + 2 * m1 + m2 + m3 + m4 + 4) >> 3);
    s[-o * 2] = (int16_t) clip3(m2 - tc, m2 + tc, (2 * m0 + m1 + 2 * m2 + m3 + m4 + m5 + 4) >> 3);
    s[-o]     = (int16_t) clip3(m3 - tc, m3 + tc, (m0 + m1 + m2 + 2 * m3 + m4 + m5 + m6 + 4) >> 3);
    s[0]      = (int16_t) clip3(m4 - tc, m4 + tc, (m1 + m2 + m3 + 2 * m4 + m5 + m6 + m7 + 4) >> 3);
    s[o]      = (int16_t) clip3(m5 - tc, m5 + tc, (m2 + m3 + m4 + 2 * m5 + m6 + 2 * m7 + 4) >> 3);
    s[o * 2]  = (int16_t) clip3(m6 - tc, m6 + tc, (m3 + m4 + m5 + 2 * m6 + 3 * m7 + 4) >> 3);
  } else {
    const int delta = clip3(-tc, tc, ((((m4 - m3) << 2) + m2 - m5 + 4) >> 3));
    s[-o] = (int16_t) clip3(0, mx, m3 + delta);
    s[0]  = (int16_t) clip3(0, mx, m4 - delta);
  }
}
static int tc_of(int idx, int bd) { return bd < 10 ? ((TC_TABLE[idx] + 2) >> (10 - bd)) : (TC_TABLE[idx] << (bd - 10)); }

/* one 4-sample segment of a luma edge: s = first sample of the Q side on line 0; o = step across the edge; step = step along the edge;
 * sizeP / sizeQ = size of the blocks on either side in the direction across the edge; ctuTop = horizontal edge on a CTU row boundary */
void orc_deblock_luma_segment(int16_t *s, int o, int step, int sizeP, int sizeQ, int ctuTop, int qp, int bd, int beta_off2, int tc_off2)
{
  int lenP, lenQ;
  if (sizeP <= 4 || sizeQ <= 4) lenP = lenQ = 1;
  else { lenQ = sizeQ >= 32 ? 7 : 3; lenP = sizeP >= 32 ? 7 : 3; }
  int pLarge = lenP > 3, qLarge = lenQ > 3;
  if (ctuTop) pLarge = 0;
  const int idxTC = clip3(0, 63 + 2, qp + 2 * (2 - 1) + (tc_off2 << 1)), idxB = clip3(0, 63, qp + (beta_off2 << 1));
  const int tc = tc_of(idxTC, bd), beta = BETA_TABLE[idxB] * (1 << (bd - 8));
  const int sideThr = (beta + (beta >> 1)) >> 3, thrCut = tc * 10, mx = (1 << bd) - 1;
  const int dp0 = calc_dp(s, o), dq0 = calc_dq(s, o), dp3 = calc_dp(s + 3 * step, o), dq3 = calc_dq(s + 3 * step, o);
  int longTap = 0;
  if (pLarge || qLarge) {
    int dp0L = dp0, dq0L = dq0, dp3L = dp3, dq3L = dq3;
    if (pLarge) { dp0L = (dp0L + calc_dp(s - 3 * o, o) + 1) >> 1; dp3L = (dp3L + calc_dp(s + 3 * step - 3 * o, o) + 1) >> 1; }
    if (qLarge) { dq0L = (dq0L + calc_dq(s + 3 * o, o) + 1) >> 1; dq3L = (dq3L + calc_dq(s + 3 * step + 3 * o, o) + 1) >> 1; }
    const int d0L = dp0L + dq0L, d3L = dp3L + dq3L, dpL = dp0L + dp3L, dqL = dq0L + dq3L;
    if (d0L + d3L < beta) {
      const int fP = dpL < sideThr, fQ = dqL < sideThr;
      if (use_strong(s, o, 2 * d0L, beta, tc, pLarge, qLarge, lenP, lenQ) && use_strong(s + 3 * step, o, 2 * d3L, beta, tc, pLarge, qLarge, lenP, lenQ)) {
        longTap = 1;
        for (int i = 0; i < 4; i++) pel_filter_luma(s + i * step, o, tc, 1, thrCut, fP, fQ, mx, pLarge, qLarge, lenP, lenQ);
      }
    }
  }
  if (!longTap) {
    const int d0 = dp0 + dq0, d3 = dp3 + dq3, dp = dp0 + dp3, dq = dq0 + dq3;
    if (d0 + d3 < beta) {
      int fP = 0, fQ = 0, sw = 0;
      if (lenP > 1 && lenQ > 1) { fP = dp < sideThr; fQ = dq < sideThr; }
      if (lenP > 2 && lenQ > 2) sw = use_strong(s, o, 2 * d0, beta, tc, 0, 0, 0, 0) && use_strong(s + 3 * step, o, 2 * d3, beta, tc, 0, 0, 0, 0);
      for (int i = 0; i < 4; i++) pel_filter_luma(s + i * step, o, tc, sw, thrCut, fP, fQ, mx, 0, 0, 0, 0);
    }
  }
}
/* one 2-line segment of a chroma edge of one component (4:2:0): sizes in chroma samples; qp = mapped chroma QP (+ offset, clipped to 0..63) */
void orc_deblock_chroma_segment(int16_t *s, int o, int step, int sizeP, int sizeQ, int ctuTop, int qp, int bd, int beta_off2, int tc_off2)
{
  int large = sizeP >= 8 && sizeQ >= 8;
  if (ctuTop) large = 0;
  const int idxTC = clip3(0, 63 + 2, qp + 2 * (2 - 1) + (tc_off2 << 1));
  const int tc = tc_of(idxTC, bd), mx = (1 << bd) - 1;
  int useLong = 0;
  if (large) {
    const int beta = BETA_TABLE[clip3(0, 63, qp + (beta_off2 << 1))] * (1 << (bd - 8));
    const int dp0 = calc_dp(s, o), dq0 = calc_dq(s, o), dp3 = calc_dp(s + step, o), dq3 = calc_dq(s + step, o);      /* JVET_O0637: the second line for 4:2:0 */
    const int d0 = dp0 + dq0, d3 = dp3 + dq3;
    if (d0 + d3 < beta) {
      useLong = 1;
      const int sw = use_strong(s, o, 2 * d0, beta, tc, 0, 0, 0, 0) && use_strong(s + step, o, 2 * d3, beta, tc, 0, 0, 0, 0);
      for (int i = 0; i < 2; i++) pel_filter_chroma(s + i * step, o, tc, sw, mx);
    }
  }
  if (!useLong) for (int i = 0; i < 2; i++) pel_filter_chroma(s + i * step, o, tc, 0, mx);
}

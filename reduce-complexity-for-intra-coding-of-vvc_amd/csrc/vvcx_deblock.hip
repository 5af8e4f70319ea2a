// vvcx_deblock.hip — in-loop deblocking of the pictures the search has coded (gfx950).
//
// ≙ LoopFilter::loopFilterPic (CL/LoopFilter.cpp:153-262) for what this library codes: intra CUs of a dual-tree I slice with one transform
// unit each, boundary strength 2 on every CU edge (xGetBoundaryStrengthSingle 701-720), no sub-block / PCM / palette / lossless / LADF /
// LMCS / virtual-boundary cases, filtering across tiles (the cfg's defaults).  One launch filters every vertical edge of every bound
// picture, a second one every horizontal edge.  A thread owns one 4-line segment of a luma edge (the 4x4 grid of xEdgeFilterLuma 892-1184)
// or one 2-line segment of a Cb + Cr edge (8x8 chroma grid, xEdgeFilterChroma 1186-1434); the read and write sets of different edges of
// one direction are disjoint by construction of the filter lengths (xSetMaxFilterLengthPQFromTransformSizes 474-578), so the picture is
// filtered in place.  The unit maps the search left in HBM give the CU on either side of an edge.  An HBM-bound pass: every sample is
// read once or twice and a few per cent are written.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "vvcx_dev.h"

__device__ static const uint16_t DB_TC[66] = { 0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,3,4,4,4,4,5,5,5,5,7,7,8,9,10,10,11,13,14,15,17,19,21,24,25,29,33,36,41,45,51,57,64,71,80,89,100,112,125,141,157,177,198,222,250,280,314,352,395 };
__device__ static const uint8_t DB_BETA[64] = { 0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,6,7,8,9,10,11,12,13,14,15,16,17,18,20,22,24,26,28,30,32,34,36,38,40,42,44,46,48,50,52,54,56,58,60,62,64,66,68,70,72,74,76,78,80,82,84,86,88 };

__device__ inline int db_clip3(int lo, int hi, int v) { return v < lo ? lo : v > hi ? hi : v; }
__device__ inline int db_abs(int v) { return v < 0 ? -v : v; }
template <typename T> __device__ inline int db_dp(const T *s, int o) { return db_abs((int) s[-o * 3] - 2 * (int) s[-o * 2] + (int) s[-o]); }
template <typename T> __device__ inline int db_dq(const T *s, int o) { return db_abs((int) s[0] - 2 * (int) s[o] + (int) s[o * 2]); }
__device__ inline int db_tc(int idx, int bd) { return bd < 10 ? ((DB_TC[idx] + 2) >> (10 - bd)) : (DB_TC[idx] << (bd - 10)); }      // JVET_O0159

// xUseStrongFiltering 1690-1733
template <typename T>
__device__ int db_use_strong(const T *s, int o, int d, int beta, int tc, int pLarge, int qLarge, int lenP, int lenQ)
{
  const int m4 = s[0], m3 = s[-o], m7 = s[o * 3], m0 = s[-o * 4];
  int sp3 = db_abs(m0 - m3), sq3 = db_abs(m7 - m4);
  if (pLarge || qLarge) {
    const int mP4 = lenP == 5 ? s[-o * 6] : s[-o * 8], m11 = lenQ == 5 ? s[o * 5] : s[o * 7];
    if (pLarge) sp3 = (sp3 + db_abs(m0 - mP4) + 1) >> 1;
    if (qLarge) sq3 = (sq3 + db_abs(m11 - m7) + 1) >> 1;
    return ((sp3 + sq3) < (beta * 3 >> 5)) && (d < (beta >> 2)) && (db_abs(m3 - m4) < ((tc * 5 + 1) >> 1));
  }
  return ((sp3 + sq3) < (beta >> 3)) && (d < (beta >> 2)) && (db_abs(m3 - m4) < ((tc * 5 + 1) >> 1));
}
// xFilteringPandQ + xBilinearFilter 1436-1529: 7 / 5 / 3 samples per side
template <typename T>
__device__ void db_filter_long(T *src, int o, int nP, int nQ, int tc)
{
  const int c7[7] = { 59, 50, 41, 32, 23, 14, 5 }, c3[3] = { 53, 32, 11 }, c5[5] = { 58, 45, 32, 19, 6 };
  const int t7[7] = { 6, 5, 4, 3, 2, 1, 1 }, t3[3] = { 6, 4, 2 };
  T *sP = src - o, *sQ = src;
#define P_(k) ((int) sP[-(k) * o])
#define Q_(k) ((int) sQ[(k) * o])
  const int refP = nP == 7 ? (P_(6) + P_(7) + 1) >> 1 : nP == 3 ? (P_(2) + P_(3) + 1) >> 1 : (P_(4) + P_(5) + 1) >> 1;
  const int refQ = nQ == 7 ? (Q_(6) + Q_(7) + 1) >> 1 : nQ == 3 ? (Q_(2) + Q_(3) + 1) >> 1 : (Q_(4) + Q_(5) + 1) >> 1;
  int mid;
  if (nP == nQ) {
    if (nP == 5) mid = (2 * (P_(0) + Q_(0) + P_(1) + Q_(1) + P_(2) + Q_(2)) + P_(3) + Q_(3) + P_(4) + Q_(4) + 8) >> 4;
    else mid = (2 * (P_(0) + Q_(0)) + P_(1) + Q_(1) + P_(2) + Q_(2) + P_(3) + Q_(3) + P_(4) + Q_(4) + P_(5) + Q_(5) + P_(6) + Q_(6) + 8) >> 4;
  } else {
    const int np = nP > nQ ? nP : nQ, nq = nP > nQ ? nQ : nP;
    if (np == 7 && nq == 5) mid = (2 * (P_(0) + Q_(0) + P_(1) + Q_(1)) + P_(2) + Q_(2) + P_(3) + Q_(3) + P_(4) + Q_(4) + P_(5) + Q_(5) + 8) >> 4;
    else if (np == 7 && nq == 3) {
      // long side L (7), short side S (3): (2 (L0 + S0) + S0 + 2 (S1 + S2) + L1 + S1 + L2 + L3 + L4 + L5 + L6 + 8) >> 4
      if (nP > nQ) mid = (2 * (P_(0) + Q_(0)) + Q_(0) + 2 * (Q_(1) + Q_(2)) + P_(1) + Q_(1) + P_(2) + P_(3) + P_(4) + P_(5) + P_(6) + 8) >> 4;
      else mid = (2 * (Q_(0) + P_(0)) + P_(0) + 2 * (P_(1) + P_(2)) + Q_(1) + P_(1) + Q_(2) + Q_(3) + Q_(4) + Q_(5) + Q_(6) + 8) >> 4;
    } else mid = (P_(0) + Q_(0) + P_(1) + Q_(1) + P_(2) + Q_(2) + P_(3) + Q_(3) + 4) >> 3;
  }
  int vP[7], vQ[7];
  for (int k = 0; k < nP; k++) { const int s = P_(k), c = nP == 7 ? c7[k] : nP == 5 ? c5[k] : c3[k], cv = (tc * (nP == 3 ? t3[k] : t7[k])) >> 1; vP[k] = db_clip3(s - cv, s + cv, (mid * c + refP * (64 - c) + 32) >> 6); }
  for (int k = 0; k < nQ; k++) { const int s = Q_(k), c = nQ == 7 ? c7[k] : nQ == 5 ? c5[k] : c3[k], cv = (tc * (nQ == 3 ? t3[k] : t7[k])) >> 1; vQ[k] = db_clip3(s - cv, s + cv, (mid * c + refQ * (64 - c) + 32) >> 6); }
  for (int k = 0; k < nP; k++) sP[-k * o] = (T) vP[k];
  for (int k = 0; k < nQ; k++) sQ[k * o] = (T) vQ[k];
#undef P_
#undef Q_
}
// xPelFilterLuma 1531-1629
template <typename T>
__device__ void db_pel_luma(T *s, int o, int tc, int sw, int thrCut, int secondP, int secondQ, int mx, int pLarge, int qLarge, int lenP, int lenQ)
{
  const int m4 = s[0], m3 = s[-o], m5 = s[o], m2 = s[-o * 2], m6 = s[o * 2], m1 = s[-o * 3], m7 = s[o * 3], m0 = s[-o * 4];
  if (sw) {
    if (pLarge || qLarge) { db_filter_long(s, o, pLarge ? lenP : 3, qLarge ? lenQ : 3, tc); return; }
    s[-o]     = (T) db_clip3(m3 - 3 * tc, m3 + 3 * tc, (m1 + 2 * m2 + 2 * m3 + 2 * m4 + m5 + 4) >> 3);
    s[0]      = (T) db_clip3(m4 - 3 * tc, m4 + 3 * tc, (m2 + 2 * m3 + 2 * m4 + 2 * m5 + m6 + 4) >> 3);
    s[-o * 2] = (T) db_clip3(m2 - 2 * tc, m2 + 2 * tc, (m1 + m2 + m3 + m4 + 2) >> 2);
    s[o]      = (T) db_clip3(m5 - 2 * tc, m5 + 2 * tc, (m3 + m4 + m5 + m6 + 2) >> 2);
    s[-o * 3] = (T) db_clip3(m1 - tc, m1 + tc, (2 * m0 + 3 * m1 + m2 + m3 + m4 + 4) >> 3);
    s[o * 2]  = (T) db_clip3(m6 - tc, m6 + tc, (m3 + m4 + m5 + 3 * m6 + 2 * m7 + 4) >> 3);
    return;
  }
  int delta = (9 * (m4 - m3) - 3 * (m5 - m2) + 8) >> 4;
  if (db_abs(delta) < thrCut) {
    delta = db_clip3(-tc, tc, delta);
    s[-o] = (T) db_clip3(0, mx, m3 + delta);
    s[0]  = (T) db_clip3(0, mx, m4 - delta);
    const int tc2 = tc >> 1;
    if (secondP) s[-o * 2] = (T) db_clip3(0, mx, m2 + db_clip3(-tc2, tc2, ((((m1 + m3 + 1) >> 1) - m2 + delta) >> 1)));
    if (secondQ) s[o]      = (T) db_clip3(0, mx, m5 + db_clip3(-tc2, tc2, ((((m6 + m4 + 1) >> 1) - m5 - delta) >> 1)));
  }
}
// xPelFilterChroma 1631-1688
template <typename T>
__device__ void db_pel_chroma(T *s, int o, int tc, int sw, int mx)
{
  const int m0 = s[-o * 4], m1 = s[-o * 3], m2 = s[-o * 2], m3 = s[-o], m4 = s[0], m5 = s[o], m6 = s[o * 2], m7 = s[o * 3];
  if (sw) {
    s[-o * 3] = (T) db_clip3(m1 - tc, m1 + tc, (3 * m0 + 2 * m1 + m2 + m3 + m4 + 4) >> 3);
    s[-o * 2] = (T) db_clip3(m2 - tc, m2 + tc, (2 * m0 + m1 + 2 * m2 + m3 + m4 + m5 + 4) >> 3);
    s[-o]     = (T) db_clip3(m3 - tc, m3 + tc, (m0 + m1 + m2 + 2 * m3 + m4 + m5 + m6 + 4) >> 3);
    s[0]      = (T) db_clip3(m4 - tc, m4 + tc, (m1 + m2 + m3 + 2 * m4 + m5 + m6 + m7 + 4) >> 3);
    s[o]      = (T) db_clip3(m5 - tc, m5 + tc, (m2 + m3 + m4 + 2 * m5 + m6 + 2 * m7 + 4) >> 3);
    s[o * 2]  = (T) db_clip3(m6 - tc, m6 + tc, (m3 + m4 + m5 + 2 * m6 + 3 * m7 + 4) >> 3);
  } else {
    const int delta = db_clip3(-tc, tc, ((((m4 - m3) << 2) + m2 - m5 + 4) >> 3));
    s[-o] = (T) db_clip3(0, mx, m3 + delta);
    s[0]  = (T) db_clip3(0, mx, m4 - delta);
  }
}
// one 4-line luma segment (xEdgeFilterLuma 975-1182): s = first Q-side sample of line 0, o = step across the edge, step = along it
template <typename T>
__device__ void db_luma_segment(T *s, int o, int step, int sizeP, int sizeQ, int ctuTop, const VxDeblockParams &p)
{
  const int bd = p.bit_depth;
  int lenP, lenQ;
  if (sizeP <= 4 || sizeQ <= 4) lenP = lenQ = 1;
  else { lenQ = sizeQ >= 32 ? 7 : 3; lenP = sizeP >= 32 ? 7 : 3; }
  int pLarge = lenP > 3; const int qLarge = lenQ > 3;
  if (ctuTop) pLarge = 0;
  const int idxTC = db_clip3(0, 65, p.qp + 2 + (p.tc_off2 << 1)), idxB = db_clip3(0, 63, p.qp + (p.beta_off2 << 1));
  const int tc = db_tc(idxTC, bd), beta = DB_BETA[idxB] << (bd - 8);
  const int sideThr = (beta + (beta >> 1)) >> 3, thrCut = tc * 10, mx = (1 << bd) - 1;
  const int dp0 = db_dp(s, o), dq0 = db_dq(s, o), dp3 = db_dp(s + 3 * step, o), dq3 = db_dq(s + 3 * step, o);
  int longTap = 0;
  if (pLarge || qLarge) {
    int dp0L = dp0, dq0L = dq0, dp3L = dp3, dq3L = dq3;
    if (pLarge) { dp0L = (dp0L + db_dp(s - 3 * o, o) + 1) >> 1; dp3L = (dp3L + db_dp(s + 3 * step - 3 * o, o) + 1) >> 1; }
    if (qLarge) { dq0L = (dq0L + db_dq(s + 3 * o, o) + 1) >> 1; dq3L = (dq3L + db_dq(s + 3 * step + 3 * o, o) + 1) >> 1; }
    const int d0L = dp0L + dq0L, d3L = dp3L + dq3L;
    if (d0L + d3L < beta) {
      const int fP = (dp0L + dp3L) < sideThr, fQ = (dq0L + dq3L) < sideThr;
      if (db_use_strong(s, o, 2 * d0L, beta, tc, pLarge, qLarge, lenP, lenQ) && db_use_strong(s + 3 * step, o, 2 * d3L, beta, tc, pLarge, qLarge, lenP, lenQ)) {
        longTap = 1;
        for (int i = 0; i < 4; i++) db_pel_luma(s + i * step, o, tc, 1, thrCut, fP, fQ, mx, pLarge, qLarge, lenP, lenQ);
      }
    }
  }
  if (!longTap) {
    const int d0 = dp0 + dq0, d3 = dp3 + dq3;
    if (d0 + d3 < beta) {
      int fP = 0, fQ = 0, sw = 0;
      if (lenP > 1 && lenQ > 1) { fP = (dp0 + dp3) < sideThr; fQ = (dq0 + dq3) < sideThr; }
      if (lenP > 2 && lenQ > 2) sw = db_use_strong(s, o, 2 * d0, beta, tc, 0, 0, 0, 0) && db_use_strong(s + 3 * step, o, 2 * d3, beta, tc, 0, 0, 0, 0);
      for (int i = 0; i < 4; i++) db_pel_luma(s + i * step, o, tc, sw, thrCut, fP, fQ, mx, 0, 0, 0, 0);
    }
  }
}
// one 2-line segment of a chroma edge of one component, 4:2:0 (xEdgeFilterChroma 1278-1430)
template <typename T>
__device__ void db_chroma_segment(T *s, int o, int step, int sizeP, int sizeQ, int ctuTop, int qp, const VxDeblockParams &p)
{
  const int bd = p.bit_depth;
  int large = sizeP >= 8 && sizeQ >= 8;
  if (ctuTop) large = 0;
  const int tc = db_tc(db_clip3(0, 65, qp + 2 + (p.tc_off2 << 1)), bd), mx = (1 << bd) - 1;
  int useLong = 0;
  if (large) {
    const int beta = DB_BETA[db_clip3(0, 63, qp + (p.beta_off2 << 1))] << (bd - 8);
    const int d0 = db_dp(s, o) + db_dq(s, o), d3 = db_dp(s + step, o) + db_dq(s + step, o);       // JVET_O0637: lines 0 and 1 for 4:2:0
    if (d0 + d3 < beta) {
      useLong = 1;
      const int sw = db_use_strong(s, o, 2 * d0, beta, tc, 0, 0, 0, 0) && db_use_strong(s + step, o, 2 * d3, beta, tc, 0, 0, 0, 0);
      for (int i = 0; i < 2; i++) db_pel_chroma(s + i * step, o, tc, sw, mx);
    }
  }
  if (!useLong) for (int i = 0; i < 2; i++) db_pel_chroma(s + i * step, o, tc, 0, mx);
}

// grid: ceil(2 * uw * uh / 256) x n_frames; the first uw*uh threads of a frame take the luma units, the next uw*uh the chroma units
template <typename T>
__device__ void deblock_pass(const VxDeblockParams &p)
{
  const VxFrameDev &fd = p.frames[blockIdx.y];
  const int n = p.uw * p.uh;
  int id = blockIdx.x * blockDim.x + threadIdx.x;
  if (id >= 2 * n || (id >= n && !p.chroma)) return;
  const int ch = id >= n; id -= ch ? n : 0;
  const int uy = id / p.uw, ux = id - uy * p.uw;
  const VxUnit u = fd.units[ch][id];
  if (!u.tag) return;
  if (!ch) {
    const int x = ux << 2, y = uy << 2, st = fd.stride[0];
    T *rec = (T *) fd.rec[0];
    if (p.dir == 0) { if (u.x == x && x > 0) db_luma_segment(rec + y * st + x, 1, st, 1 << fd.units[0][id - 1].lw, 1 << u.lw, 0, p); }
    else if (u.y == y && y > 0) db_luma_segment(rec + y * st + x, st, 1, 1 << fd.units[0][id - p.uw].lh, 1 << u.lh, (y & 127) == 0, p);
  } else {
    const int cx = ux << 1, cy = uy << 1;
    for (int k = 0; k < 2; k++) {
      const int st = fd.stride[k + 1], qpc = db_clip3(0, 63, p.qp_c[k]);
      T *rec = (T *) fd.rec[k + 1];
      if (p.dir == 0) { if (u.x == cx && cx > 0 && (cx & 7) == 0) db_chroma_segment(rec + cy * st + cx, 1, st, 1 << fd.units[1][id - 1].lw, 1 << u.lw, 0, qpc, p); }
      else if (u.y == cy && cy > 0 && (cy & 7) == 0) db_chroma_segment(rec + cy * st + cx, st, 1, 1 << fd.units[1][id - p.uw].lh, 1 << u.lh, (cy & 63) == 0, qpc, p);
    }
  }
}
extern "C" __global__ void __launch_bounds__(256) vvcx_deblock_kernel_u8(VxDeblockParams p) { deblock_pass<uint8_t>(p); }
extern "C" __global__ void __launch_bounds__(256) vvcx_deblock_kernel_u16(VxDeblockParams p) { deblock_pass<uint16_t>(p); }

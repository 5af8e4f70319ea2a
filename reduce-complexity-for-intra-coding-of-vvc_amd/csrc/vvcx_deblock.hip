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
__device__ inline int db_tc(int idx, int bd) { return bd < 10 ? ((DB_TC[idx] + 2) >> (10 - bd)) : (DB_TC[idx] << (bd - 10)); }      // JVET_O0159

// One line across an edge held in registers: p[k] = k-th sample on the P side counted from the edge (p[0] = p0), q[k] likewise.
// Loaded with one 32-bit (8-bit samples) or 64-bit (16-bit samples) access per four samples; only the samples a filter changed are stored.
struct DbLine { int p[8], q[8]; };
__device__ inline int db_dp(const DbLine &L) { return db_abs(L.p[2] - 2 * L.p[1] + L.p[0]); }
__device__ inline int db_dq(const DbLine &L) { return db_abs(L.q[0] - 2 * L.q[1] + L.q[2]); }
__device__ inline int db_dp3(const DbLine &L) { return db_abs(L.p[5] - 2 * L.p[4] + L.p[3]); }       // xCalcDP three samples further out
__device__ inline int db_dq3(const DbLine &L) { return db_abs(L.q[3] - 2 * L.q[4] + L.q[5]); }

// xUseStrongFiltering 1690-1733 (lenP / lenQ are 7 or 3 here: p[7] = mP4 of the 7-sample side)
__device__ int db_use_strong(const DbLine &L, int d, int beta, int tc, int pLarge, int qLarge)
{
  int sp3 = db_abs(L.p[3] - L.p[0]), sq3 = db_abs(L.q[3] - L.q[0]);
  if (pLarge || qLarge) {
    if (pLarge) sp3 = (sp3 + db_abs(L.p[3] - L.p[7]) + 1) >> 1;
    if (qLarge) sq3 = (sq3 + db_abs(L.q[7] - L.q[3]) + 1) >> 1;
    return ((sp3 + sq3) < (beta * 3 >> 5)) && (d < (beta >> 2)) && (db_abs(L.p[0] - L.q[0]) < ((tc * 5 + 1) >> 1));
  }
  return ((sp3 + sq3) < (beta >> 3)) && (d < (beta >> 2)) && (db_abs(L.p[0] - L.q[0]) < ((tc * 5 + 1) >> 1));
}
// xFilteringPandQ + xBilinearFilter 1436-1529: 7 or 3 samples per side (5 only occurs with sub-block motion)
__device__ void db_filter_long(DbLine &L, int nP, int nQ, int tc)
{
  const int c7[7] = { 59, 50, 41, 32, 23, 14, 5 }, c3[3] = { 53, 32, 11 }, t7[7] = { 6, 5, 4, 3, 2, 1, 1 }, t3[3] = { 6, 4, 2 };
  const int *p = L.p, *q = L.q;
  const int refP = nP == 7 ? (p[6] + p[7] + 1) >> 1 : (p[2] + p[3] + 1) >> 1;
  const int refQ = nQ == 7 ? (q[6] + q[7] + 1) >> 1 : (q[2] + q[3] + 1) >> 1;
  int mid;
  if (nP == nQ) mid = (2 * (p[0] + q[0]) + p[1] + q[1] + p[2] + q[2] + p[3] + q[3] + p[4] + q[4] + p[5] + q[5] + p[6] + q[6] + 8) >> 4;
  else if (nP > nQ) mid = (2 * (p[0] + q[0]) + q[0] + 2 * (q[1] + q[2]) + p[1] + q[1] + p[2] + p[3] + p[4] + p[5] + p[6] + 8) >> 4;
  else mid = (2 * (q[0] + p[0]) + p[0] + 2 * (p[1] + p[2]) + q[1] + p[1] + q[2] + q[3] + q[4] + q[5] + q[6] + 8) >> 4;
  int vP[7], vQ[7];
  _Pragma("unroll") for (int k = 0; k < 7; k++) if (k < nP) { const int s = p[k], c = nP == 7 ? c7[k] : c3[k < 3 ? k : 0], cv = (tc * (nP == 3 ? t3[k < 3 ? k : 0] : t7[k])) >> 1; vP[k] = db_clip3(s - cv, s + cv, (mid * c + refP * (64 - c) + 32) >> 6); }
  _Pragma("unroll") for (int k = 0; k < 7; k++) if (k < nQ) { const int s = q[k], c = nQ == 7 ? c7[k] : c3[k < 3 ? k : 0], cv = (tc * (nQ == 3 ? t3[k < 3 ? k : 0] : t7[k])) >> 1; vQ[k] = db_clip3(s - cv, s + cv, (mid * c + refQ * (64 - c) + 32) >> 6); }
  _Pragma("unroll") for (int k = 0; k < 7; k++) { if (k < nP) L.p[k] = vP[k]; if (k < nQ) L.q[k] = vQ[k]; }
}
// xPelFilterLuma 1531-1629
__device__ void db_pel_luma(DbLine &L, int tc, int sw, int thrCut, int secondP, int secondQ, int mx, int pLarge, int qLarge)
{
  const int m4 = L.q[0], m3 = L.p[0], m5 = L.q[1], m2 = L.p[1], m6 = L.q[2], m1 = L.p[2], m7 = L.q[3], m0 = L.p[3];
  if (sw) {
    if (pLarge || qLarge) { db_filter_long(L, pLarge ? 7 : 3, qLarge ? 7 : 3, tc); return; }
    L.p[0] = db_clip3(m3 - 3 * tc, m3 + 3 * tc, (m1 + 2 * m2 + 2 * m3 + 2 * m4 + m5 + 4) >> 3);
    L.q[0] = db_clip3(m4 - 3 * tc, m4 + 3 * tc, (m2 + 2 * m3 + 2 * m4 + 2 * m5 + m6 + 4) >> 3);
    L.p[1] = db_clip3(m2 - 2 * tc, m2 + 2 * tc, (m1 + m2 + m3 + m4 + 2) >> 2);
    L.q[1] = db_clip3(m5 - 2 * tc, m5 + 2 * tc, (m3 + m4 + m5 + m6 + 2) >> 2);
    L.p[2] = db_clip3(m1 - tc, m1 + tc, (2 * m0 + 3 * m1 + m2 + m3 + m4 + 4) >> 3);
    L.q[2] = db_clip3(m6 - tc, m6 + tc, (m3 + m4 + m5 + 3 * m6 + 2 * m7 + 4) >> 3);
    return;
  }
  int delta = (9 * (m4 - m3) - 3 * (m5 - m2) + 8) >> 4;
  if (db_abs(delta) < thrCut) {
    delta = db_clip3(-tc, tc, delta);
    L.p[0] = db_clip3(0, mx, m3 + delta);
    L.q[0] = db_clip3(0, mx, m4 - delta);
    const int tc2 = tc >> 1;
    if (secondP) L.p[1] = db_clip3(0, mx, m2 + db_clip3(-tc2, tc2, ((((m1 + m3 + 1) >> 1) - m2 + delta) >> 1)));
    if (secondQ) L.q[1] = db_clip3(0, mx, m5 + db_clip3(-tc2, tc2, ((((m6 + m4 + 1) >> 1) - m5 - delta) >> 1)));
  }
}
// xPelFilterChroma 1631-1688
__device__ void db_pel_chroma(DbLine &L, int tc, int sw, int mx)
{
  const int m0 = L.p[3], m1 = L.p[2], m2 = L.p[1], m3 = L.p[0], m4 = L.q[0], m5 = L.q[1], m6 = L.q[2], m7 = L.q[3];
  if (sw) {
    L.p[2] = db_clip3(m1 - tc, m1 + tc, (3 * m0 + 2 * m1 + m2 + m3 + m4 + 4) >> 3);
    L.p[1] = db_clip3(m2 - tc, m2 + tc, (2 * m0 + m1 + 2 * m2 + m3 + m4 + m5 + 4) >> 3);
    L.p[0] = db_clip3(m3 - tc, m3 + tc, (m0 + m1 + m2 + 2 * m3 + m4 + m5 + m6 + 4) >> 3);
    L.q[0] = db_clip3(m4 - tc, m4 + tc, (m1 + m2 + m3 + 2 * m4 + m5 + m6 + m7 + 4) >> 3);
    L.q[1] = db_clip3(m5 - tc, m5 + tc, (m2 + m3 + m4 + 2 * m5 + m6 + 2 * m7 + 4) >> 3);
    L.q[2] = db_clip3(m6 - tc, m6 + tc, (m3 + m4 + m5 + 2 * m6 + 3 * m7 + 4) >> 3);
  } else {
    const int delta = db_clip3(-tc, tc, ((((m4 - m3) << 2) + m2 - m5 + 4) >> 3));
    L.p[0] = db_clip3(0, mx, m3 + delta);
    L.q[0] = db_clip3(0, mx, m4 - delta);
  }
}
// four samples starting at a (4-sample aligned) address as ints
template <typename T> __device__ inline void db_load4(const T *a, int v[4]);
template <> __device__ inline void db_load4<uint8_t>(const uint8_t *a, int v[4]) { const uint32_t w = *(const uint32_t *) a; v[0] = w & 255; v[1] = (w >> 8) & 255; v[2] = (w >> 16) & 255; v[3] = w >> 24; }
template <> __device__ inline void db_load4<uint16_t>(const uint16_t *a, int v[4]) { const uint2 w = *(const uint2 *) a; v[0] = w.x & 0xffff; v[1] = w.x >> 16; v[2] = w.y & 0xffff; v[3] = w.y >> 16; }
// N lines of an edge segment: s = first Q-side sample of line 0; across the edge the stride is o, along it step (one of them is 1).
// nP / nQ = samples to read per side (4, or 8 next to a block of 32 or more)
template <typename T, int N>
__device__ void db_load(const T *s, int o, int step, int nP, int nQ, DbLine *L)
{
  int v[4];
  if (o == 1) {                        // vertical edge: a line is a picture row
    _Pragma("unroll") for (int l = 0; l < N; l++) {
      const T *a = s + l * step;
      db_load4<T>(a - 4, v); for (int k = 0; k < 4; k++) L[l].p[k] = v[3 - k];
      if (nP > 4) { db_load4<T>(a - 8, v); for (int k = 0; k < 4; k++) L[l].p[4 + k] = v[3 - k]; }
      db_load4<T>(a, v); for (int k = 0; k < 4; k++) L[l].q[k] = v[k];
      if (nQ > 4) { db_load4<T>(a + 4, v); for (int k = 0; k < 4; k++) L[l].q[4 + k] = v[k]; }
    }
  } else {                             // horizontal edge: the lines are N (<= 4) adjacent columns, one access per picture row
    _Pragma("unroll") for (int k = 0; k < 8; k++) {
      if (k < nP) { db_load4<T>(s - (k + 1) * o, v); for (int l = 0; l < N; l++) L[l].p[k] = v[l]; }
      if (k < nQ) { db_load4<T>(s + k * o, v); for (int l = 0; l < N; l++) L[l].q[k] = v[l]; }
    }
  }
}
// store the wP / wQ samples next to the edge that the chosen filter owns (nobody else writes them, so rewriting an unchanged one is harmless)
template <typename T>
__device__ inline void db_store(T *s, int o, const DbLine &cur, int wP, int wQ)
{
  _Pragma("unroll") for (int k = 0; k < 7; k++) { if (k < wP) s[-(k + 1) * o] = (T) cur.p[k]; if (k < wQ) s[k * o] = (T) cur.q[k]; }
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
  if (beta == 0 && tc == 0) return;
  const int sideThr = (beta + (beta >> 1)) >> 3, thrCut = tc * 10, mx = (1 << bd) - 1;
  DbLine L[4];
  db_load<T, 4>(s, o, step, lenP > 3 ? 8 : 4, qLarge ? 8 : 4, L);
  _Pragma("unroll") for (int l = 0; l < 4; l++) _Pragma("unroll") for (int k = 4; k < 8; k++) { if (!(lenP > 3)) L[l].p[k] = 0; if (!qLarge) L[l].q[k] = 0; }
  int wP = 0, wQ = 0;                 // samples per side the filter that runs may change
  const int dp0 = db_dp(L[0]), dq0 = db_dq(L[0]), dp3 = db_dp(L[3]), dq3 = db_dq(L[3]);
  int longTap = 0;
  if (pLarge || qLarge) {
    int dp0L = dp0, dq0L = dq0, dp3L = dp3, dq3L = dq3;
    if (pLarge) { dp0L = (dp0L + db_dp3(L[0]) + 1) >> 1; dp3L = (dp3L + db_dp3(L[3]) + 1) >> 1; }
    if (qLarge) { dq0L = (dq0L + db_dq3(L[0]) + 1) >> 1; dq3L = (dq3L + db_dq3(L[3]) + 1) >> 1; }
    const int d0L = dp0L + dq0L, d3L = dp3L + dq3L;
    if (d0L + d3L < beta) {
      const int fP = (dp0L + dp3L) < sideThr, fQ = (dq0L + dq3L) < sideThr;
      if (db_use_strong(L[0], 2 * d0L, beta, tc, pLarge, qLarge) && db_use_strong(L[3], 2 * d3L, beta, tc, pLarge, qLarge)) {
        longTap = 1; wP = pLarge ? 7 : 3; wQ = qLarge ? 7 : 3;
        _Pragma("unroll") for (int l = 0; l < 4; l++) db_pel_luma(L[l], tc, 1, thrCut, fP, fQ, mx, pLarge, qLarge);
      }
    }
  }
  if (!longTap) {
    const int d0 = dp0 + dq0, d3 = dp3 + dq3;
    if (d0 + d3 < beta) {
      int fP = 0, fQ = 0, sw = 0;
      if (lenP > 1 && lenQ > 1) { fP = (dp0 + dp3) < sideThr; fQ = (dq0 + dq3) < sideThr; }
      if (lenP > 2 && lenQ > 2) sw = db_use_strong(L[0], 2 * d0, beta, tc, 0, 0) && db_use_strong(L[3], 2 * d3, beta, tc, 0, 0);
      wP = sw ? 3 : fP ? 2 : 1; wQ = sw ? 3 : fQ ? 2 : 1;
      _Pragma("unroll") for (int l = 0; l < 4; l++) db_pel_luma(L[l], tc, sw, thrCut, fP, fQ, mx, 0, 0);
    }
  }
  if (wP | wQ) _Pragma("unroll") for (int l = 0; l < 4; l++) db_store(s + l * step, o, L[l], wP, wQ);
}
// one 2-line segment of a chroma edge of one component, 4:2:0 (xEdgeFilterChroma 1278-1430)
template <typename T>
__device__ void db_chroma_segment(T *s, int o, int step, int sizeP, int sizeQ, int ctuTop, int qp, const VxDeblockParams &p)
{
  const int bd = p.bit_depth;
  int large = sizeP >= 8 && sizeQ >= 8;
  if (ctuTop) large = 0;
  const int tc = db_tc(db_clip3(0, 65, qp + 2 + (p.tc_off2 << 1)), bd), mx = (1 << bd) - 1;
  DbLine L[2]; int wPQ = 1;
  // 2 lines = 2 chroma samples along the edge: a horizontal edge segment starts at an even column, so the aligned 4-column access is the
  // one at column & ~3 and the two lines are its entries (column & 2), (column & 2) + 1
  const int sub = (o == 1) ? 0 : (int) (((uintptr_t) s / sizeof(T)) & 2);
  if (o == 1) db_load<T, 2>(s, o, step, 4, 4, L);
  else { DbLine W[4]; db_load<T, 4>(s - sub, o, step, 4, 4, W); L[0] = sub ? W[2] : W[0]; L[1] = sub ? W[3] : W[1]; }
  _Pragma("unroll") for (int l = 0; l < 2; l++) _Pragma("unroll") for (int k = 4; k < 8; k++) { L[l].p[k] = 0; L[l].q[k] = 0; }
  int useLong = 0;
  if (large) {
    const int beta = DB_BETA[db_clip3(0, 63, qp + (p.beta_off2 << 1))] << (bd - 8);
    const int d0 = db_dp(L[0]) + db_dq(L[0]), d3 = db_dp(L[1]) + db_dq(L[1]);       // JVET_O0637: lines 0 and 1 for 4:2:0
    if (d0 + d3 < beta) {
      useLong = 1;
      const int sw = db_use_strong(L[0], 2 * d0, beta, tc, 0, 0) && db_use_strong(L[1], 2 * d3, beta, tc, 0, 0);
      if (sw) wPQ = 3;
      _Pragma("unroll") for (int i = 0; i < 2; i++) db_pel_chroma(L[i], tc, sw, mx);
    }
  }
  if (!useLong) _Pragma("unroll") for (int i = 0; i < 2; i++) db_pel_chroma(L[i], tc, 0, mx);
  _Pragma("unroll") for (int l = 0; l < 2; l++) db_store(s + l * step, o, L[l], wPQ, wPQ);
}

// size, across an edge of direction dir (0: vertical edge -> width, 1: horizontal edge -> height), of the transform unit a luma 4x4 unit lies in: the CU's, or the
// sub-partition's where the CU is an ISP CU split across that direction (xSetMaxFilterLengthPQFromTransformSizes 474-575 takes the lengths from tuP / tuQ)
__device__ inline int db_tu_size(const VxUnit &u, int dir)
{
  const int w = 1 << u.lw, h = 1 << u.lh, isp = (u.mts >> 6) & 3;
  if (isp != (dir ? 1 : 2)) return dir ? h : w;
  const int parts = ((w == 4 && h == 8) || (w == 8 && h == 4)) ? 2 : 4;
  return (dir ? h : w) / parts;
}
// Pre-pass: the 24-byte unit records the search left are read once and boiled down to one byte per unit, map and direction - is the unit's left / upper border an edge to
// filter (a CU border, or a sub-partition border of an ISP CU on the 4-sample grid), and the transform sizes on either side (powers of two: their logarithms) - so that
// the two filter passes read a byte per unit instead of one or two records (which were three times their sample traffic).
// grid: ceil(2 * uw * uh / 256) x n_frames; the first uw*uh threads of a frame take the luma units, the next uw*uh the chroma units
__device__ inline int db_log2(int v) { return 31 - __clz(v); }
__device__ void deblock_edges(const VxDeblockParams &p)
{
  const VxFrameDev &fd = p.frames[blockIdx.y];
  const int n = p.uw * p.uh;
  int id = blockIdx.x * blockDim.x + threadIdx.x;
  if (id >= 2 * n) return;
  const int ch = id >= n; id -= ch ? n : 0;
  uint8_t *e = p.edges + (size_t) blockIdx.y * 4 * n;
  uint8_t ev = 0, eh = 0;
  if (!ch || p.chroma) {
    const int uy = id / p.uw, ux = id - uy * p.uw;
    const VxUnit u = fd.units[ch][id];
    if (u.tag) {
      if (!ch) {
        const int x = ux << 2, y = uy << 2, isp = (u.mts >> 6) & 3;
        for (int dir = 0; dir < 2; dir++) {                   // transform edges (xDeblockCU 306-317): the CU border, and inside an ISP CU the sub-partition borders on the 4-sample grid
          const int tq = db_tu_size(u, dir), off = dir ? y - u.y : x - u.x;
          const int inner = off > 0 && isp == (dir ? 1 : 2) && off % tq == 0;
          if ((dir ? y : x) > 0 && (off == 0 || inner)) {
            const int tp = inner ? tq : db_tu_size(fd.units[0][dir ? id - p.uw : id - 1], dir);
            (dir ? eh : ev) = (uint8_t) (0x80 | (db_log2(tp) << 3) | db_log2(tq));
          }
        }
      } else {
        const int cx = ux << 1, cy = uy << 1;
        if (u.x == cx && cx > 0 && (cx & 7) == 0) ev = (uint8_t) (0x80 | (fd.units[1][id - 1].lw << 3) | u.lw);
        if (u.y == cy && cy > 0 && (cy & 7) == 0) eh = (uint8_t) (0x80 | (fd.units[1][id - p.uw].lh << 3) | u.lh);
      }
    }
  }
  e[(size_t) ch * n + id] = ev; e[(size_t) (2 + ch) * n + id] = eh;
}
// the filter passes: same grid; a thread whose unit has no edge in the pass's direction leaves after one byte
template <typename T>
__device__ void deblock_pass(const VxDeblockParams &p)
{
  const VxFrameDev &fd = p.frames[blockIdx.y];
  const int n = p.uw * p.uh;
  int id = blockIdx.x * blockDim.x + threadIdx.x;
  if (id >= 2 * n || (id >= n && !p.chroma)) return;
  const int ch = id >= n; id -= ch ? n : 0;
  const uint8_t e = p.edges[(size_t) blockIdx.y * 4 * n + (size_t) (2 * p.dir + ch) * n + id];
  if (!e) return;
  const int uy = id / p.uw, ux = id - uy * p.uw, sizeP = 1 << ((e >> 3) & 7), sizeQ = 1 << (e & 7);
  if (!ch) {
    const int x = ux << 2, y = uy << 2, st = fd.stride[0];
    T *rec = (T *) fd.rec[0];
    if (p.dir == 0) db_luma_segment(rec + y * st + x, 1, st, sizeP, sizeQ, 0, p);
    else db_luma_segment(rec + y * st + x, st, 1, sizeP, sizeQ, (y & 127) == 0, p);
  } else {
    const int cx = ux << 1, cy = uy << 1;
    for (int k = 0; k < 2; k++) {
      const int st = fd.stride[k + 1], qpc = db_clip3(0, 63, p.qp_c[k]);
      T *rec = (T *) fd.rec[k + 1];
      if (p.dir == 0) db_chroma_segment(rec + cy * st + cx, 1, st, sizeP, sizeQ, 0, qpc, p);
      else db_chroma_segment(rec + cy * st + cx, st, 1, sizeP, sizeQ, (cy & 63) == 0, qpc, p);
    }
  }
}
extern "C" __global__ void __launch_bounds__(256) vvcx_deblock_edges_kernel(VxDeblockParams p) { deblock_edges(p); }
extern "C" __global__ void __launch_bounds__(256) vvcx_deblock_kernel_u8(VxDeblockParams p) { deblock_pass<uint8_t>(p); }
extern "C" __global__ void __launch_bounds__(256) vvcx_deblock_kernel_u16(VxDeblockParams p) { deblock_pass<uint16_t>(p); }

// vvcx_depquant_dev.h — dependent (trellis-coded) quantisation of one transform block by one wavefront, and the state-driven dequantiser.
// Included by vvcx_kernel.hip after the LDS layout (uses L, ScanGeo, the context models and the bit tables).
//
// What it computes is CL/DepQuant.cpp: DQIntern::DepQuant::quant 1592-1731 (first tested position, four-state trellis over the scan positions,
// back-tracking), xDecide 1455-1517, State::checkRdCosts 918-1030 / checkRdCostStart 1032 / checkRdCostSkipSbb 1052, updateState 1109-1273,
// updateStateEOS 1275-1315, CommonCtx::update 1317-1398, Quantizer::preQuantCoeff 812-832 and dequantBlock 741-810.  How it is laid out:
//   * the four trellis states of a position live in the four lanes of a quad (lane & 3 = state id); every quad of the wave carries the same
//     four states, so every exchange between states is a quad permutation (DPP) or a shuffle inside the quad and all 64 lanes stay converged;
//   * a state is 16 registers: cost, packed counters and context increments, the 16 level bytes of the current coefficient group and the 16
//     template sums of its positions' neighbours outside the group.  The reference keeps twelve State objects in three roles that swap every
//     position and an update writes only some members; the three roles (cur / prv / skp) are kept as three register sets with the same
//     swaps, so the members an update leaves alone hold what the reference's objects hold;
//   * rate terms are read from the estimator's context models when they are needed (the reference tabulates them per block), the per-block
//     constants of Quantizer::initQuantBlock (694-739: fp64) come precomputed from the host per (component, log2 w + log2 h);
//   * the history a path needs when it enters the next coefficient group (the levels of the groups to its right and below, the group
//     significance flags) is a linked list of per-group nodes {16 levels, flag, parent state} instead of the reference's eight level-array
//     copies; at a group change the 64 lanes compute the 4 x 16 template sums of the next group in one step;
//   * a decision is stored in 4 bits (where it came from and whether the level is zero); the level itself is recomputed from the coefficient
//     when the winning path is walked back.
#pragma once

struct DqS { long long cost; int pk, rem; unsigned lev[4]; unsigned tm[8]; };
// pk: [0,5) non-zero levels of the path in the current group, [5,8) hist + 1 (state id of the path at the last group change), [8,10) Rice parameter,
// [10,16) zero position of the bypass mode, [16,18) sig_coeff_group context + 1 (0: no bits), [18,22) sig_coeff_flag context increment,
// [22,27) context increment of the gt1 / par / gt2 set
#define DQ_NUMSIG(p) ((p) & 31)
#define DQ_HIST(p) ((((p) >> 5) & 7) - 1)
#define DQ_RPAR(p) (((p) >> 8) & 3)
#define DQ_RZERO(p) (((p) >> 10) & 63)
#define DQ_SBBC(p) ((((p) >> 16) & 3) - 1)
#define DQ_SIGI(p) (((p) >> 18) & 15)
#define DQ_GTXI(p) (((p) >> 22) & 31)
__device__ inline int dq_put(int p, int sh, int nbits, int v) { const int m = ((1 << nbits) - 1) << sh; return (p & ~m) | ((v << sh) & m); }

__device__ inline unsigned dq_get_b(const unsigned *a, int i) { const unsigned w = i < 8 ? (i < 4 ? a[0] : a[1]) : (i < 12 ? a[2] : a[3]); return (w >> ((i & 3) << 3)) & 255u; }
__device__ inline void dq_set_b(unsigned *a, int i, unsigned v)
{
  const unsigned sh = (unsigned) (i & 3) << 3, m = ~(255u << sh); const int wi = i >> 2;
#pragma unroll
  for (int j = 0; j < 4; j++) if (wi == j) a[j] = (a[j] & m) | (v << sh);
}
__device__ inline unsigned dq_get_h(const unsigned *a, int i)
{
  const int wi = i >> 1;
  const unsigned w = wi < 4 ? (wi < 2 ? (wi == 0 ? a[0] : a[1]) : (wi == 2 ? a[2] : a[3])) : (wi < 6 ? (wi == 4 ? a[4] : a[5]) : (wi == 6 ? a[6] : a[7]));
  return (w >> ((i & 1) << 4)) & 0xffffu;
}
template <int CTRL> __device__ inline long long dq_quad_i64(long long v)
{
  const int lo = __builtin_amdgcn_update_dpp(0, (int) v, CTRL, 0xF, 0xF, true), hi = __builtin_amdgcn_update_dpp(0, (int) (v >> 32), CTRL, 0xF, 0xF, true);
  return (long long) (((unsigned long long) (unsigned) hi << 32) | (unsigned) lo);
}
__device__ inline long long dq_shfl_i64(long long v, int src)
{
  const int lo = __shfl((int) v, src), hi = __shfl((int) (v >> 32), src);
  return (long long) (((unsigned long long) (unsigned) hi << 32) | (unsigned) lo);
}
// fractional bits of a bin from the model's current state (BinProbModel_Std::getFracBitsArray, CL/Contexts.h:128-132)
__device__ inline int dq_fb(int ci, int ctx, int bin) { const Ctx &c = L.ctxs[ci]; return (int) L.t.bin_frac[((((unsigned) c.s0[ctx] + (unsigned) c.s1[ctx]) >> 8) << 1) + (unsigned) bin]; }
// g_goRiceBits (887-893) = length of the Golomb-Rice / escape code (EL/BinEncoder.cpp:444-472) in 2^-15 bit units
__device__ inline int dq_rice_bits(int par, unsigned v) { return rem_abs_len(v, (unsigned) par) << 15; }

struct DqRate { int ci, sigBase, g1Base, g2Base, parBase, sbbBase; };
// rate of a non-zero level under the state's contexts (m_coeffFracBits, RateEstimator::xSetGtxFlagBits 597-618, plus the Rice remainder)
__device__ inline long long dq_lev_bits(const DqRate &R, int gtx_i, int rice_par, int lev)
{
  if (lev == 0) return 0;
  if (lev == 1) return dq_fb(R.ci, R.g1Base + gtx_i, 0) + (1 << 15);
  long long b = (long long) dq_fb(R.ci, R.g1Base + gtx_i, 1) + (1 << 15) + dq_fb(R.ci, R.parBase + gtx_i, lev & 1) + dq_fb(R.ci, R.g2Base + gtx_i, lev >= 4);
  if (lev >= 4) { const unsigned v = (unsigned) (lev - 4) >> 1; b += dq_rice_bits(rice_par, v < 32 ? v : 31); }
  return b;
}
// the level of pre-quantiser candidate i (0..3) and its distortion change (preQuantCoeff 812-832): the four consecutive quantisation indices from
// qIdx0 on, candidate i being the one with index & 3 == i
struct DqPq { int lev; long long dd; };
__device__ inline DqPq dq_pq(const VxDqConst &q, long long scaledOrg, int qIdx0, int i)
{
  const int step = (i - qIdx0) & 3, qi = qIdx0 + step;
  const long long scaledAdd = (long long) qIdx0 * q.dstep - scaledOrg * q.dorg + (long long) step * q.dstep;
  DqPq r; r.dd = (scaledAdd * qi + q.dadd) >> q.dshift; r.lev = (qi + 1) >> 1;
  return r;
}
__device__ inline int dq_qidx0(const VxDqConst &q, long long scaledOrg)
{
  const int v = (int) ((scaledOrg + q.qadd) >> q.qshift);
  return imax(1, imin(q.max_qidx, v));
}
__device__ inline int dq_sig_off(int ch, int diag) { return ch ? (diag < 2 ? 4 : 0) : (diag < 2 ? 8 : diag < 5 ? 4 : 0); }            // xSetScanInfo 402-419
__device__ inline int dq_gtx_off(int ch, int diag) { return ch ? (diag < 1 ? 6 : 1) : (diag < 1 ? 16 : diag < 3 ? 11 : diag < 10 ? 6 : 1); }

// working memory of one block: decisions (4 bits per state and position), the per-group path nodes, the last-position offsets per group index
struct DqMem { uint16_t *trel; uint8_t *hlev; int8_t *hpar; uint8_t *hflag; int *lastb; };

// cf: the block's transform coefficients as int16 (they fit: the forward transforms keep 15 bits + sign) in raster order, stride w; replaced by the
// levels.  ci: which context set the rate terms are read from (the estimator's contexts at the time of the call).  Returns absSum (uniform).
template <bool SMALL>
__device__ __noinline__ int wave_depquant(int16_t *cf_g, int buf_off, uint8_t *scratch, int ci, int w, int h, int comp, int cbf_ctx, int zo, int lfnst, int lane)
{
  w = uni(w); h = uni(h); comp = uni(comp); ci = uni(ci); cbf_ctx = uni(cbf_ctx); zo = uni(zo); lfnst = uni(lfnst);
  const int wave_ = uni(threadIdx.x >> 6);
  int16_t *cf = SMALL ? L.slot[wave_] + BUF + uni(buf_off) : cf_g;
  DqMem M;
  if (SMALL) { WaveDq &d = L.ws[wave_].dq; M.trel = d.trel; M.hlev = &d.hlev[0][0][0]; M.hpar = &d.hpar[0][0]; M.hflag = &d.hflag[0][0]; M.lastb = d.lastb; }
  else { uint8_t *b = scratch + VXD_OFF_DQ + (size_t) wave_ * VXD_DQ_WAVE; M.trel = (uint16_t *) b; M.hlev = b + 2048; M.hpar = (int8_t *) (b + 2048 + 4096); M.hflag = b + 2048 + 4096 + 256; M.lastb = (int *) (b + 2048 + 4096 + 512); }
  const int ch = comp ? 1 : 0, lw = ilog2i(w), lh = ilog2i(h);
  const VxDqConst q = L.par.dq_consts[comp * 16 + lw + lh];
  const ScanGeo geo = scan_geo(w, h);
  const int lcw = geo.lcw, lch = geo.lch, lcg = geo.lcg, gs = 1 << lcg, total = geo.nscan;
  const int nzw = imin(32, w), nzh = imin(32, h), wsbb = geo.wg, hsbb = geo.hg;
  const uint8_t *cg_inv = L.t.cg_inv + (lcw == 2 ? 0 : lcw == 3 ? 20 : lch == 3 ? 36 : 16);
  const uint8_t *grp_inv = L.t.grp_inv + 15 * (wsbb - 1) + wsbb * (hsbb - 1);
  int effW = w, effH = h, zeroOut = 0;
  if (zo && comp == 0) { effH = h == 32 ? 16 : h; effW = w == 32 ? 16 : w; zeroOut = effH < h || effW < w; }
  // ---- first tested position (1630-1660): the last scan position whose coefficient exceeds the threshold
  int first = total - 1;
  if (lfnst > 0 && w >= 4 && h >= 4) first = ((w == 4 && h == 4) || (w == 8 && h == 8)) ? 7 : 15;
  {
    const int thr = q.thres / (int) (4 * q.qscale);
    int found = -1;
    for (int top = first | 63; top >= 63 && found < 0; top -= 64) {
      const int sp = top - lane;
      bool hit = false;
      if (sp <= first) {
        const int blk = scan_blk(geo, sp), x = blk & (w - 1), y = blk >> lw;
        hit = !(zeroOut && (x >= effW || y >= effH)) && iabs((int) cf[blk]) > thr;
      }
      const unsigned long long m = __ballot(hit);
      if (m) found = top - (__ffsll(m) - 1);
    }
    first = uni(found);
  }
  if (first < 0) { for (int i = lane; i < total; i += 64) cf[scan_blk(geo, i)] = 0; wave_sync(); return 0; }

  DqRate R; R.ci = ci; R.sigBase = 0; R.g1Base = VX_CTX_GtxFlag[2 + ch]; R.g2Base = VX_CTX_GtxFlag[ch]; R.parBase = VX_CTX_ParFlag[ch]; R.sbbBase = VX_CTX_SigCoeffGroup[ch];
  const int k = lane & 3, qbase = lane & ~3;
  const int sigSet = VX_CTX_SigFlag[ch + 2 * imax(k - 1, 0)];          // the state's sig_coeff_flag context set (443-446)
  // ---- last-position offsets per group index (RateEstimator::xSetLastCoeffOffset 488-568)
  {
    int cbfDelta = 0;
    if (cbf_ctx >= 0) cbfDelta = dq_fb(ci, cbf_ctx, 1) - dq_fb(ci, cbf_ctx, 0);
    if (lane < 32) {
      const int xy = lane >> 4, id = lane & 15;
      const int size = xy ? h : w, l2 = ilog2i(size);
      const int base = (xy ? VX_CTX_LastY : VX_CTX_LastX)[ch];
      const int sh = comp == 0 ? (l2 + 1) >> 2 : imin(2, size >> 3), lo = comp == 0 ? L.t.last_prefix[l2] : 0;
      const int maxId = L.t.group_idx[imin(32, size) - 1];
      if (id <= maxId) {
        unsigned sum = 0;
        for (int j = 0; j < id; j++) sum += (unsigned) dq_fb(ci, base + lo + (j >> sh), 1);
        unsigned b = sum + (id > 3 ? (unsigned) ((id - 2) >> 1) << 15 : 0u) + (unsigned) (xy ? cbfDelta : 0);
        if (id < maxId) b += (unsigned) dq_fb(ci, base + lo + (id >> sh), 0);
        M.lastb[xy * 16 + id] = (int) b;
      }
    }
  }
  wave_sync();
  // start state (State::init of m_startState: contexts of increment 0, Rice parameter 0) — constants of the block
  const int regFull = (imin(32, effW) * imin(32, effH) * 28) >> 4;
  DqS cur, prv, skp;
  {
    DqS s0; s0.cost = 0x7fffffffffffffffll >> 1; s0.rem = 4;
    s0.pk = dq_put(0, 5, 3, 0);                             // hist -1, everything else 0 (sbbc -1: no bits)
#pragma unroll
    for (int i = 0; i < 4; i++) s0.lev[i] = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) s0.tm[i] = 0;
    cur = s0; prv = s0; skp = s0;
  }
  long long decCost = 0;
  for (int sp = first; sp >= 0; sp--) {
    const int blk = scan_blk(geo, sp), x = blk & (w - 1), y = blk >> lw;
    const int inside = sp & (gs - 1), eos = inside == 0;
    int spt = 0;
    if (inside == gs - 1 && sp > gs && sp < total - 1) spt = 1; else if (eos && sp > 0 && sp < total - gs) spt = 2;
    const int zeroed = zeroOut && (x >= effW || y >= effH);
    { const DqS t = prv; prv = cur; cur = t; }
    // ---- decision of target state k (xDecide 1455-1517)
    long long dc = 0x7fffffffffffffffll >> 2; int dsrc = 0, dnz = 0, dlev = -1;          // dsrc: 0 none, 1 start, 2 from the "A/zero" source state, 3 from the "B" source state, 4 sub-block skipped
    if (zeroed) {
      if (spt == 2) { dc = skp.cost + (DQ_SBBC(skp.pk) >= 0 ? dq_fb(ci, R.sbbBase + DQ_SBBC(skp.pk), 0) : 0); dsrc = 4; dlev = 0; }
    } else {
      const long long scaledOrg = (long long) iabs((int) cf[blk]) * q.qscale;
      const int qIdx0 = dq_qidx0(q, scaledOrg);
      // this lane as SOURCE state k: candidates A / B / zero (checkRdCosts 918-1030)
      const DqPq pA = dq_pq(q, scaledOrg, qIdx0, k < 2 ? 0 : 3), pB = dq_pq(q, scaledOrg, qIdx0, k < 2 ? 2 : 1);
      long long cA = prv.cost + pA.dd, cB = prv.cost + pB.dd, cZ = prv.cost;
      {
        const int ppk = prv.pk, rpar = DQ_RPAR(ppk);
        if (prv.rem >= 4) {
          cA += dq_lev_bits(R, DQ_GTXI(ppk), rpar, pA.lev); cB += dq_lev_bits(R, DQ_GTXI(ppk), rpar, pB.lev);
          const int s0b = dq_fb(ci, sigSet + DQ_SIGI(ppk), 0), s1b = dq_fb(ci, sigSet + DQ_SIGI(ppk), 1);
          if (spt == 0) { cA += s1b; cB += s1b; cZ += s0b; }
          else if (spt == 1) { const int sb = DQ_SBBC(ppk) >= 0 ? dq_fb(ci, R.sbbBase + DQ_SBBC(ppk), 1) : 0; cA += sb + s1b; cB += sb + s1b; cZ += sb + s0b; }
          else if (DQ_NUMSIG(ppk)) { cA += s1b; cB += s1b; cZ += s0b; }
          else cZ = 0x7fffffffffffffffll;                   // a group whose flag is coded as significant cannot end all-zero: no zero candidate
        } else {
          const int rz = DQ_RZERO(ppk);
          cA += (1 << 15) + dq_rice_bits(rpar, (unsigned) (pA.lev <= rz ? pA.lev - 1 : (pA.lev < 32 ? pA.lev : 31)));
          cB += (1 << 15) + dq_rice_bits(rpar, (unsigned) (pB.lev <= rz ? pB.lev - 1 : (pB.lev < 32 ? pB.lev : 31)));
          cZ += dq_rice_bits(rpar, (unsigned) rz);
        }
      }
      // gather: target 0 <- {A, Z of state 0; B of state 1}, target 2 <- {B of 0; A, Z of 1}, target 1 <- {A, Z of 2; B of 3}, target 3 <- {B of 2; A, Z of 3}
      const long long azA = dq_quad_i64<0xD8>(cA), azZ = dq_quad_i64<0xD8>(cZ), bB = dq_quad_i64<0x8D>(cB);      // quad_perm [0,2,1,3] / [1,3,0,2]
      const DqPq tA = dq_pq(q, scaledOrg, qIdx0, (k & 1) ? 3 : 0), tB = dq_pq(q, scaledOrg, qIdx0, (k & 1) ? 1 : 2);
      if (k < 2) {
        if (azA < dc) { dc = azA; dsrc = 2; dnz = 1; dlev = tA.lev; }
        if (azZ < dc) { dc = azZ; dsrc = 2; dnz = 0; dlev = 0; }
        if (bB < dc) { dc = bB; dsrc = 3; dnz = 1; dlev = tB.lev; }
      } else {
        if (bB < dc) { dc = bB; dsrc = 3; dnz = 1; dlev = tB.lev; }
        if (azA < dc) { dc = azA; dsrc = 2; dnz = 1; dlev = tA.lev; }
        if (azZ < dc) { dc = azZ; dsrc = 2; dnz = 0; dlev = 0; }
      }
      if (spt == 2) {                                       // checkRdCostSkipSbb 1052-1061
        const long long c = skp.cost + (DQ_SBBC(skp.pk) >= 0 ? dq_fb(ci, R.sbbBase + DQ_SBBC(skp.pk), 0) : 0);
        if (c < dc) { dc = c; dsrc = 4; dnz = 0; dlev = 0; }
      }
      if (!(k & 1)) {                                       // checkRdCostStart 1032-1050 into decisions 0 (candidate 0) and 2 (candidate 2)
        const DqPq pS = dq_pq(q, scaledOrg, qIdx0, k);
        const long long c = pS.dd + (long long) (M.lastb[L.t.group_idx[x]] + M.lastb[16 + L.t.group_idx[y]]) + dq_lev_bits(R, 0, 0, pS.lev);
        if (c < dc) { dc = c; dsrc = 1; dnz = 1; dlev = pS.lev; }
      }
    }
    {                                                       // the four decisions of the position: 4 bits each
      int e = ((dsrc << 1) | dnz) << (4 * k);
      e = seg_sum<4>(e);
      if (lane == 0) M.trel[sp] = (uint16_t) e;
    }
    decCost = dc;
    if (sp == 0) break;
    // ---- state update (xDecideAndUpdate 1527-1588)
    const int prevId = dsrc == 2 ? ((k & 1) ? (k == 1 ? 2 : 3) : (k == 0 ? 0 : 1)) : dsrc == 3 ? ((k & 1) ? (k == 1 ? 3 : 2) : (k == 0 ? 1 : 0)) : dsrc == 4 ? 4 + k : dsrc == 1 ? -1 : -2;
    const int nextBlk = scan_blk(geo, sp - 1), xn = nextBlk & (w - 1), yn = nextBlk >> lw, nin = (sp - 1) & (gs - 1);
    if (eos || !zeroed) {
      // parent state (a previous state of the quad) — every lane shuffles, the lanes whose decision has no such parent discard the result
      const int srcLane = qbase + ((prevId >= 0 && prevId < 4) ? prevId : k);
      DqS P;
      P.pk = __shfl(prv.pk, srcLane); P.rem = __shfl(prv.rem, srcLane);
#pragma unroll
      for (int i = 0; i < 4; i++) P.lev[i] = (unsigned) __shfl((int) prv.lev[i], srcLane);
      if (!eos) {
#pragma unroll
        for (int i = 0; i < 8; i++) P.tm[i] = (unsigned) __shfl((int) prv.tm[i], srcLane);
      }
      const bool alive = prevId > -2, fromPrev = prevId >= 0 && prevId < 4;
      cur.cost = dc;
      if (!eos) {                                           // State::updateState 1109-1273
        if (alive) {
          int pk = cur.pk;
          if (fromPrev) {
            pk = dq_put(pk, 0, 5, DQ_NUMSIG(P.pk) + (dlev != 0)); pk = dq_put(pk, 5, 3, DQ_HIST(P.pk) + 1); pk = dq_put(pk, 16, 2, DQ_SBBC(P.pk) + 1); pk = dq_put(pk, 8, 2, DQ_RPAR(P.pk));
            cur.rem = P.rem - 1;
            if (cur.rem >= 4) cur.rem -= dlev < 2 ? dlev : 3;
#pragma unroll
            for (int i = 0; i < 4; i++) cur.lev[i] = P.lev[i];
#pragma unroll
            for (int i = 0; i < 8; i++) cur.tm[i] = P.tm[i];
          } else {
            pk = dq_put(pk, 0, 5, 1); pk = dq_put(pk, 5, 3, 0);
            cur.rem = regFull - (dlev < 2 ? dlev : 3);
#pragma unroll
            for (int i = 0; i < 4; i++) cur.lev[i] = 0;
#pragma unroll
            for (int i = 0; i < 8; i++) cur.tm[i] = 0;
          }
          dq_set_b(cur.lev, inside, (unsigned) imin(255, dlev));
          // template of the next position: its neighbours inside the group (m_scanId2NbInfoSbb) on top of the sums over those outside
          const int t = (int) dq_get_h(cur.tm, nin);
          int sumAbs = t >> 8, sumAbs1 = (t >> 3) & 31, sumNum = t & 7;
#pragma unroll
          for (int n = 0; n < 5; n++) {
            const int xx = xn + (n == 0 ? 1 : n == 1 ? 2 : n == 2 ? 1 : 0), yy = yn + (n == 2 ? 1 : n == 3 ? 1 : n == 4 ? 2 : 0);
            if (xx < nzw && yy < nzh && (xx >> lcw) == (xn >> lcw) && (yy >> lch) == (yn >> lch)) {
              const int a = (int) dq_get_b(cur.lev, cg_inv[((yy & ((1 << lch) - 1)) << lcw) | (xx & ((1 << lcw) - 1))]);
              sumAbs += a; sumAbs1 += imin(4 + (a & 1), a); sumNum += a != 0;
            }
          }
          if (cur.rem >= 4) {
            const int diag = xn + yn;
            pk = dq_put(pk, 18, 4, dq_sig_off(ch, diag) + imin((sumAbs1 + 1) >> 1, 3)); pk = dq_put(pk, 22, 5, dq_gtx_off(ch, diag) + imin(sumAbs1 - sumNum, 4));
            pk = dq_put(pk, 8, 2, L.t.gorice_pars[imax(imin(31, sumAbs - 20), 0)]);
          } else {
            sumAbs = imin(31, sumAbs);
            pk = dq_put(pk, 8, 2, L.t.gorice_pars[sumAbs]); pk = dq_put(pk, 10, 6, L.t.gorice_pos0[imax(0, k - 1) * 32 + sumAbs]);
          }
          cur.pk = pk;
        }
      } else {                                              // State::updateStateEOS 1275-1315 + CommonCtx::update 1317-1398
        const int g = sp >> lcg;
        int pk = cur.pk; int numSig, pHist, pRem; unsigned lv[4];
        if (prevId >= 4) { numSig = 0; pHist = DQ_HIST(skp.pk); pRem = skp.rem; lv[0] = lv[1] = lv[2] = lv[3] = 0; }
        else if (fromPrev) { numSig = DQ_NUMSIG(P.pk) + (dlev != 0); pHist = DQ_HIST(P.pk); pRem = P.rem; lv[0] = P.lev[0]; lv[1] = P.lev[1]; lv[2] = P.lev[2]; lv[3] = P.lev[3]; }
        else { numSig = 1; pHist = -1; pRem = regFull; lv[0] = lv[1] = lv[2] = lv[3] = 0; }
        dq_set_b(lv, 0, (unsigned) imin(255, imax(dlev, 0)));
        if (alive && lane < 4) {                            // the path's node of this group
          uint32_t *hl = (uint32_t *) (M.hlev + (g * 4 + k) * 16);
          hl[0] = lv[0]; hl[1] = lv[1]; hl[2] = lv[2]; hl[3] = lv[3];
          M.hpar[g * 4 + k] = (int8_t) pHist; M.hflag[g * 4 + k] = (uint8_t) (numSig != 0);
        }
        wave_sync();
        // the groups right of, below and diagonally below the next group, and the state the path had when it left each of them
        const unsigned ng = geo.grp[g - 1]; const int nsx = (int) (ng & 15), nsy = (int) (ng >> 4);
        const int gR = nsx < wsbb - 1 ? grp_inv[nsy * wsbb + nsx + 1] : -1, gB = nsy < hsbb - 1 ? grp_inv[(nsy + 1) * wsbb + nsx] : -1;
        const int gD = (gR >= 0 && gB >= 0) ? grp_inv[(nsy + 1) * wsbb + nsx + 1] : -1;
        int kR = -1, kB = -1, kD = -1;
        {
          const int gmax = imax(gR, imax(gB, gD));
          int kk = k;
          for (int gen = g; gen <= gmax; gen++) {
            if (gen == gR) kR = kk; if (gen == gB) kB = kk; if (gen == gD) kD = kk;
            if (gen == gmax) break;
            kk = M.hpar[gen * 4 + kk];
            if (kk < 0 || kk > 3) break;
          }
        }
        const int sigN = ((kR >= 0 && M.hflag[gR * 4 + kR]) || (kB >= 0 && M.hflag[gB * 4 + kB])) ? 1 : 0;
        // template sums of the next group's positions over their neighbours outside it: lane = (position << 2) | state
        int tval = 0;
        {
          const int id = lane >> 2;
          if (id < gs) {
            const int pb = scan_blk(geo, ((g - 1) << lcg) + id), px = pb & (w - 1), py = pb >> lw;
            int sumAbs = 0, sumAbs1 = 0, sumNum = 0, any = 0;
#pragma unroll
            for (int n = 0; n < 5; n++) {
              const int xx = px + (n == 0 ? 1 : n == 1 ? 2 : n == 2 ? 1 : 0), yy = py + (n == 2 ? 1 : n == 3 ? 1 : n == 4 ? 2 : 0);
              const int sx = xx >> lcw, sy = yy >> lch;
              if (xx < nzw && yy < nzh && (sx != nsx || sy != nsy)) {
                any = 1;
                const int gq = sx != nsx ? (sy != nsy ? gD : gR) : gB, kq = sx != nsx ? (sy != nsy ? kD : kR) : kB;
                if (kq >= 0) {
                  const int a = M.hlev[(gq * 4 + kq) * 16 + cg_inv[((yy & ((1 << lch) - 1)) << lcw) | (xx & ((1 << lcw) - 1))]];
                  sumAbs += a; sumAbs1 += imin(4 + (a & 1), a); sumNum += a != 0;
                }
              }
            }
            if (any) tval = sumNum + (sumAbs1 << 3) + (imin(127, sumAbs) << 8);
          }
        }
        unsigned tmn[8];
#pragma unroll
        for (int j = 0; j < 8; j++) { const int a = __shfl(tval, 8 * j + k), b = __shfl(tval, 8 * j + 4 + k); tmn[j] = (unsigned) a | ((unsigned) b << 16); }
        if (alive) {
          pk = dq_put(pk, 0, 5, 0); pk = dq_put(pk, 8, 2, 0); pk = dq_put(pk, 5, 3, k + 1); pk = dq_put(pk, 16, 2, sigN + 1);
          cur.rem = pRem;
#pragma unroll
          for (int i = 0; i < 4; i++) cur.lev[i] = 0;
#pragma unroll
          for (int i = 0; i < 8; i++) cur.tm[i] = tmn[i];
          const int t = (int) dq_get_h(cur.tm, nin), diag = xn + yn;
          const int sumAbs1 = (t >> 3) & 31, sumNum = t & 7;
          pk = dq_put(pk, 18, 4, dq_sig_off(ch, diag) + imin((sumAbs1 + 1) >> 1, 3)); pk = dq_put(pk, 22, 5, dq_gtx_off(ch, diag) + imin(sumAbs1 - sumNum, 4));
          cur.pk = pk;
        }
        wave_sync();
      }
    }
    if (spt == 1) { const DqS t = prv; prv = skp; skp = t; }
  }
  // ---- best final state and back-tracking (1709-1730)
  int prev = -2; long long minCost = 0;
#pragma unroll
  for (int s = 0; s < 4; s++) { const long long c = dq_shfl_i64(decCost, s); if (c < minCost) { prev = s; minCost = c; } }
  int nPath = 0, absSum = 0;
  if (lane == 0) {
    int sp = 0;
    while (prev >= 0) {
      const int e = (M.trel[sp] >> (4 * prev)) & 15, src = e >> 1;
      int lev = 0;
      if (e & 1) {
        const long long scaledOrg = (long long) iabs((int) cf[scan_blk(geo, sp)]) * q.qscale;
        const int i = src == 1 ? prev : src == 2 ? ((prev & 1) ? 3 : 0) : ((prev & 1) ? 1 : 2);
        lev = dq_pq(q, scaledOrg, dq_qidx0(q, scaledOrg), i).lev;
      }
      M.trel[sp] = (uint16_t) lev; absSum += lev;
      if (src == 4) { for (int j = 1; j < gs; j++) M.trel[sp + j] = 0; sp += gs; }
      else { prev = src == 1 ? -1 : src == 2 ? ((prev & 1) ? (prev == 1 ? 2 : 3) : (prev == 0 ? 0 : 1)) : src == 3 ? ((prev & 1) ? (prev == 1 ? 3 : 2) : (prev == 0 ? 1 : 0)) : -2; sp++; }
    }
    nPath = sp;
  }
  nPath = __shfl(nPath, 0); absSum = __shfl(absSum, 0);
  wave_sync();
  for (int sp = lane; sp < total; sp += 64) {
    const int blk = scan_blk(geo, sp), c = cf[blk];
    const int lv = sp < nPath ? (int) M.trel[sp] : 0;
    cf[blk] = (int16_t) (c < 0 ? -lv : lv);
  }
  wave_sync();
  return uni(absSum);
}

// Quantizer::dequantBlock (741-810) for all coefficients of a block at once.  The quantiser state in front of a scan position is a function of the
// parities of the levels coded before it: with the transition table 32040 a step maps the state bits (x, y) to (y ^ parity, x), so x is the
// parity sum of the levels at odd distances and y the one at even distances (>= 2) — every lane derives its state from one ballot per 64
// positions.  lev: levels (stride w); deq: zw x zh dequantised coefficients (int16: they are clipped to 16 bits), row-major.
__device__ inline void wave_dequant_dq(const int16_t *lev, int16_t *deq, int w, int h, int zw, int zh, int bd, int qp, int lane)
{
  const ScanGeo geo = scan_geo(w, h);
  const int lw = ilog2i(w), lh = ilog2i(h), lzw = ilog2i(zw), total = geo.nscan;
  int last = -1;
  for (int top = (total - 1) | 63; top >= 63 && last < 0; top -= 64) {
    const int sp = top - lane;
    const unsigned long long m = __ballot(sp < total && lev[scan_blk(geo, sp)] != 0);
    if (m) last = top - (__ffsll(m) - 1);
  }
  last = uni(last);
  for (int i = lane; i < zw * zh; i += 64) deq[i] = 0;
  wave_sync();
  if (last < 0) return;
  const int sq = (lw + lh) & 1, qpDQ = qp + 1, per = qpDQ / 6, rem = qpDQ - 6 * per;
  const int trShift = 15 - bd - ((lw + lh) >> 1) - sq;
  const int shift = 6 + 1 - per - trShift;
  int invQ = L.t.iqscale[sq * 6 + rem];
  if (shift < 0) invQ <<= -shift;                          // 800-803: from the first coded coefficient on
  const int sh = shift < 0 ? 0 : shift, add = shift < 0 ? 0 : ((1 << shift) >> 1);
  int x0 = 0, y0 = 0;
  const unsigned long long below = lane == 0 ? 0ull : (~0ull >> (64 - lane));
  const unsigned long long EVEN = 0x5555555555555555ull, ODD = 0xAAAAAAAAAAAAAAAAull;
  for (int t0 = 0; t0 <= last; t0 += 64) {
    const int sp = last - t0 - lane;
    int lv = 0, blk = 0;
    if (sp >= 0) { blk = scan_blk(geo, sp); lv = lev[blk]; }
    const unsigned long long P = __ballot(lv & 1);
    const int odd = lane & 1;
    const int xs = (odd ? y0 : x0) ^ (__popcll(P & below & (odd ? EVEN : ODD)) & 1);
    if (lv) {
      const int qIdx = (lv << 1) + (lv > 0 ? -xs : xs);
      const long long v = ((long long) qIdx * (long long) invQ + add) >> sh;
      deq[((blk >> lw) << lzw) + (blk & (w - 1))] = (int16_t) (v < -32768 ? -32768 : v > 32767 ? 32767 : v);
    }
    x0 ^= __popcll(P & ODD) & 1; y0 ^= __popcll(P & EVEN) & 1;
  }
  wave_sync();
}

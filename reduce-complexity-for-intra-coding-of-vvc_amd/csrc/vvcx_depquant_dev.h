// vvcx_depquant_dev.h — dependent (trellis-coded) quantisation of up to 16 transform blocks at once by one wavefront, and the state-driven dequantiser.
// Included by vvcx_kernel.hip after the LDS layout (uses L, ScanGeo, the context models and the bit tables).
//
// What it computes is CL/DepQuant.cpp: DQIntern::DepQuant::quant 1592-1731 (first tested position, four-state trellis over the scan positions,
// back-tracking), xDecide 1455-1517, State::checkRdCosts 918-1030 / checkRdCostStart 1032 / checkRdCostSkipSbb 1052, updateState 1109-1273,
// updateStateEOS 1275-1315, CommonCtx::update 1317-1398, Quantizer::preQuantCoeff 812-832 and dequantBlock 741-810.  How it is laid out:
//   * the trellis is serial over the scan positions of a block and four states wide, so one block can keep four lanes busy.  A wavefront therefore
//     runs up to 16 blocks of the same shape side by side ("items": the candidates of one node that are ready to be quantised), one per quad:
//     lane & 3 = trellis state, lane >> 2 = item.  Every exchange between states is a quad permutation (DPP) or a shuffle inside the quad;
//   * a state is 18 registers: cost, packed counters and context increments, the path's states / significance flags at the last group changes,
//     the 16 level bytes of the current coefficient group and the 16 template sums of its positions' neighbours outside the group.  The
//     reference keeps twelve State objects in three roles that swap every position and an update writes only some members; the three roles
//     (cur / prv / skp) are three register sets with the same swaps, so the members an update leaves alone hold what the reference's hold;
//   * rate terms are read from the estimator's context models when they are needed (the reference tabulates them per block), the per-block
//     constants of Quantizer::initQuantBlock (694-739: fp64) come precomputed from the host per (component, log2 w + log2 h);
//   * what a path needs when it enters the next coefficient group — the levels of the groups to its right and below — is kept as one 16-byte node
//     per (group, state) in HBM, found through the ancestor states the state carries in a register pair (the reference copies level arrays);
//   * a decision is stored in 4 bits (where it came from and whether the level is zero) in LDS; the level itself is recomputed from the
//     coefficient when the winning path is walked back.
#pragma once
#ifndef VXD_SERIAL_PRIO
#define VXD_SERIAL_PRIO 0      // s_setprio level of the serial phases (trellis position loop, mode controller).  Measured: 0 / 2 / 3 give 179.1 / 178.7 / 178.3 CTU/s (profiles/r04_ab_runs.txt): no gain, left off
#endif

// 16 bytes / 16 half-words kept as scalar members (arrays indexed with a run-time value would be placed in scratch memory by the compiler)
struct U4 { unsigned a, b, c, d; };
struct U8 { unsigned a, b, c, d, e, f, g, h; };
struct DqS { long long cost; int pk, rem; unsigned long long anc; U4 lev; };      // the template sums of a path live in LDS rows, pk bits 27-29 name the row
// pk: [0,5) non-zero levels of the path in the current group, [5,8) hist + 1 (state id of the path at the last group change), [8,10) Rice parameter,
// [10,16) zero position of the bypass mode, [16,18) sig_coeff_group context + 1 (0: no bits), [18,22) sig_coeff_flag context increment,
// [22,27) context increment of the gt1 / par / gt2 set
// anc: 16 fields of 4 bits; field d describes the coefficient group d groups back in coding order (0 = the last one left): state + 1 of the path when it
// left that group (0: the path did not exist yet) and, in bit 3, whether the path has a non-zero level in it.  The groups a template reaches
// (right, below, diagonally below the next group) are at most 14 groups back (8 x 8 groups of a 32x32 block).
#define DQ_NUMSIG(p) ((p) & 31)
#define DQ_HIST(p) ((((p) >> 5) & 7) - 1)
#define DQ_RPAR(p) (((p) >> 8) & 3)
#define DQ_RZERO(p) (((p) >> 10) & 63)
#define DQ_SBBC(p) ((((p) >> 16) & 3) - 1)
#define DQ_SIGI(p) (((p) >> 18) & 15)
#define DQ_GTXI(p) (((p) >> 22) & 31)
__device__ inline int dq_put(int p, int sh, int nbits, int v) { const int m = ((1 << nbits) - 1) << sh; return (p & ~m) | ((v << sh) & m); }

// (the register barrier keeps the compiler from folding the selects into one load with a computed address, which would pin the struct in scratch memory)
// the path nodes live in HBM scratch: global_ accesses (a plain pointer argument is a generic one)
#ifndef VX_GLOBAL
#ifdef VX_EMU
#define VX_GLOBAL
#else
#define VX_GLOBAL __attribute__((address_space(1)))
#endif
#endif
#ifndef VX_REG_BARRIER
#define VX_REG_BARRIER(x) asm volatile("" : "+v"(x))
#endif
__device__ inline unsigned dq_get_b(const U4 &v, int i)
{
  unsigned a = v.a, b = v.b, c = v.c, d = v.d;
  VX_REG_BARRIER(a); VX_REG_BARRIER(b); VX_REG_BARRIER(c); VX_REG_BARRIER(d);
  const unsigned w = i < 8 ? (i < 4 ? a : b) : (i < 12 ? c : d);
  return (w >> ((i & 3) << 3)) & 255u;
}
__device__ inline void dq_set_b(U4 &v, int i, unsigned x)
{
  const unsigned sh = (unsigned) (i & 3) << 3, m = ~(255u << sh), y = x << sh; const int wi = i >> 2;
  v.a = wi == 0 ? (v.a & m) | y : v.a; v.b = wi == 1 ? (v.b & m) | y : v.b; v.c = wi == 2 ? (v.c & m) | y : v.c; v.d = wi == 3 ? (v.d & m) | y : v.d;
}
__device__ inline unsigned dq_get_h(const U8 &v, int i)
{
  const int wi = i >> 1;
  unsigned a = v.a, b = v.b, c = v.c, d = v.d, e = v.e, f = v.f, g = v.g, h = v.h;
  VX_REG_BARRIER(a); VX_REG_BARRIER(b); VX_REG_BARRIER(c); VX_REG_BARRIER(d); VX_REG_BARRIER(e); VX_REG_BARRIER(f); VX_REG_BARRIER(g); VX_REG_BARRIER(h);
  const unsigned w = wi < 4 ? (wi < 2 ? (wi == 0 ? a : b) : (wi == 2 ? c : d)) : (wi < 6 ? (wi == 4 ? e : f) : (wi == 6 ? g : h));
  return (w >> ((i & 1) << 4)) & 0xffffu;
}
__device__ inline void dq_set_h(U8 &v, int i, unsigned x)
{
  const unsigned sh = (unsigned) (i & 1) << 4, m = ~(0xffffu << sh), y = x << sh; const int wi = i >> 1;
  v.a = wi == 0 ? (v.a & m) | y : v.a; v.b = wi == 1 ? (v.b & m) | y : v.b; v.c = wi == 2 ? (v.c & m) | y : v.c; v.d = wi == 3 ? (v.d & m) | y : v.d;
  v.e = wi == 4 ? (v.e & m) | y : v.e; v.f = wi == 5 ? (v.f & m) | y : v.f; v.g = wi == 6 ? (v.g & m) | y : v.g; v.h = wi == 7 ? (v.h & m) | y : v.h;
}
__device__ inline U4 dq_shfl_u4(const U4 &v, int src) { U4 r; r.a = (unsigned) __shfl((int) v.a, src); r.b = (unsigned) __shfl((int) v.b, src); r.c = (unsigned) __shfl((int) v.c, src); r.d = (unsigned) __shfl((int) v.d, src); return r; }
__device__ inline U8 dq_shfl_u8(const U8 &v, int src)
{
  U8 r; r.a = (unsigned) __shfl((int) v.a, src); r.b = (unsigned) __shfl((int) v.b, src); r.c = (unsigned) __shfl((int) v.c, src); r.d = (unsigned) __shfl((int) v.d, src);
  r.e = (unsigned) __shfl((int) v.e, src); r.f = (unsigned) __shfl((int) v.f, src); r.g = (unsigned) __shfl((int) v.g, src); r.h = (unsigned) __shfl((int) v.h, src);
  return r;
}
template <int CTRL> __device__ inline long long dq_quad_i64(long long v)
{
  const int lo = __builtin_amdgcn_update_dpp(0, (int) v, CTRL, 0xF, 0xF, true), hi = __builtin_amdgcn_update_dpp(0, (int) (v >> 32), CTRL, 0xF, 0xF, true);
  return (long long) (((unsigned long long) (unsigned) hi << 32) | (unsigned) lo);
}
// both permutations by every lane of the quad (a lane reads through them from lanes that take the other one), then the select
__device__ inline int dq_parent(int v, bool fromA)
{
  const int a = __builtin_amdgcn_update_dpp(0, v, 0xD8, 0xF, 0xF, true), b = __builtin_amdgcn_update_dpp(0, v, 0x8D, 0xF, 0xF, true);
  return fromA ? a : b;
}
__device__ inline long long dq_shfl_i64(long long v, int src)
{
  const int lo = __shfl((int) v, src), hi = __shfl((int) (v >> 32), src);
  return (long long) (((unsigned long long) (unsigned) hi << 32) | (unsigned) lo);
}
__device__ inline long long dq_uni_ll(long long v)
{
  const int lo = __builtin_amdgcn_readfirstlane((int) v), hi = __builtin_amdgcn_readfirstlane((int) (v >> 32));
  return (long long) (((unsigned long long) (unsigned) hi << 32) | (unsigned) lo);
}
// fractional bits of a bin from the model's current state (BinProbModel_Std::getFracBitsArray, CL/Contexts.h:128-132)
__device__ inline int dq_fb(int ci, int ctx, int bin) { const Ctx &c = L.ctxs[ci]; return (int) BIN_FRAC(((unsigned) c.s0[ctx] + (unsigned) c.s1[ctx]) >> 8, bin); }
// g_goRiceBits (887-893) = length of the Golomb-Rice / escape code (EL/BinEncoder.cpp:444-472) in 2^-15 bit units
__device__ inline int dq_rice_bits(int par, unsigned v) { return rem_abs_len(v, (unsigned) par) << 15; }

struct DqRate { int ci, g1Base, g2Base, parBase, sbbBase; };
// rate of a non-zero level under the state's contexts (m_coeffFracBits, RateEstimator::xSetGtxFlagBits 597-618, plus the Rice remainder)
__device__ inline long long dq_lev_bits(const DqRate &R, int gtx_i, int rice_par, int lev)
{
  if (lev == 0) return 0;
  if (lev == 1) return dq_fb(R.ci, R.g1Base + gtx_i, 0) + (1 << 15);
  long long b = (long long) dq_fb(R.ci, R.g1Base + gtx_i, 1) + (1 << 15) + dq_fb(R.ci, R.parBase + gtx_i, lev & 1) + dq_fb(R.ci, R.g2Base + gtx_i, lev >= 4);
  if (lev >= 4) { const unsigned v = (unsigned) (lev - 4) >> 1; b += dq_rice_bits(rice_par, v < 32 ? v : 31); }
  return b;
}
// the pre-quantiser (preQuantCoeff 812-832): the first of four consecutive quantisation indices; candidate i (0..3) is the one with index & 3 == i
__device__ inline int dq_qidx0(const VxDqConst &q, long long scaledOrg)
{
  const int v = (int) ((scaledOrg + q.qadd) >> q.qshift);
  return imax(1, imin(q.max_qidx, v));
}
__device__ inline long long dq_dd(const VxDqConst &q, long long scaledOrg, int qIdx0, int step)      // distortion change of index qIdx0 + step against the zero level
{
  const long long scaledAdd = (long long) qIdx0 * q.dstep - scaledOrg * q.dorg + (long long) step * q.dstep;
  return (scaledAdd * (qIdx0 + step) + q.dadd) >> q.dshift;
}
__device__ inline int dq_sig_off(int ch, int diag) { return ch ? (diag < 2 ? 4 : 0) : (diag < 2 ? 8 : diag < 5 ? 4 : 0); }            // xSetScanInfo 402-419
__device__ inline int dq_gtx_off(int ch, int diag) { return ch ? (diag < 1 ? 6 : 1) : (diag < 1 ? 16 : diag < 3 ? 11 : diag < 10 ? 6 : 1); }

// n_items blocks (1..16) of w x h.  Item i (quad i of the wave): coefficients cf_base + i * cf_stride (int16 — the forward transforms keep 15 bits
// + sign — raster order, stride w; LDS or HBM), replaced by its levels; rate terms from context set ci0 + i * ci_step with the cbf context
// cbf_ctx0 + bit i of cbf_mask (cbf_ctx0 < 0: inferred cbf); path nodes at nodes + i * node_stride bytes (HBM, 4 * positions bytes each); decisions
// and last-position offsets in the LDS area wk (240 + 2 * positions bytes per item).  absSum of item i -> L.dq_abs[abs0 + i].
// TAB != 0 (only with ci_step == 0: all items price against the same models): the rate terms are tabulated as RateEstimator::initCtx does
// (m_gtxFracBits per context and level class, m_sigFracBits, the group flag), so a rate in the serial loop is one LDS read instead of a chain of
// model reads.  TAB 2: the tables of the node's models, built once per full-RD operation by the whole workgroup (dq_build_tables -> L.dq_tab);
// TAB 1: built here behind the items' areas (DQ_TAB_BYTES more of wk), for a wave that quantises one block against its own models.
#define DQ_TAB_BYTES (4 * DQ_TAB_INTS)
#ifdef VVCX_STAMP_DQ
#define DQ_T(n_) const long long n_ = clock64()
#define DQ_U(n_) n_ = clock64()
#else
#define DQ_T(n_)
#define DQ_U(n_)
#endif
// levTab[gtx context][0 | lev 1 | 2 | 3 | even >= 4 | odd >= 4] (without the Rice part), sigTab[state set][context][bin], sbbTab[context][bin]
__device__ inline void dq_fill_tables(int *tab, int ci, int ch, int t, int nt)
{
  const int ngtx = ch ? 11 : 21, nsig = ch ? 8 : 12;
  const int g1Base = VX_CTX_GtxFlag[2 + ch], g2Base = VX_CTX_GtxFlag[ch], parBase = VX_CTX_ParFlag[ch], sbbBase = VX_CTX_SigCoeffGroup[ch];
  int *levTab = tab, *sigTab = tab + 21 * 6, *sbbTab = sigTab + 3 * 12 * 2;
  for (int e = t; e < ngtx * 6; e += nt) {
    const int g = e / 6, j = e - g * 6;
    levTab[e] = j == 0 ? 0 : j == 1 ? dq_fb(ci, g1Base + g, 0) + (1 << 15) : dq_fb(ci, g1Base + g, 1) + (1 << 15) + dq_fb(ci, parBase + g, j & 1) + dq_fb(ci, g2Base + g, j >= 4);
  }
  for (int e = t; e < 3 * nsig * 2; e += nt) { const int st = e / (nsig * 2), r = e - st * nsig * 2; sigTab[st * 24 + r] = dq_fb(ci, VX_CTX_SigFlag[ch + 2 * st] + (r >> 1), r & 1); }
  if (t < 4) sbbTab[t] = dq_fb(ci, sbbBase + (t >> 1), t & 1);
}
// all threads of the workgroup, at the start of a full-RD operation whose trellises price against the estimator's current models; a barrier follows
__device__ inline void dq_build_tables(int ch) { dq_fill_tables(L.dq_tab, CI_CUR, ch, (int) VTX, NT); __syncthreads(); }
template <int TAB>
__device__ __noinline__ void wave_depquant_batch(int n_items, int16_t *cf_base, int cf_stride, uint8_t *nodes, int node_stride, int wk_wave, int abs0,
                                                 int ci0, int ci_step, int cbf_ctx0, unsigned cbf_mask, int w, int h, int comp, int zo, int lfnst, int lane, int qidx = -1)
{
  n_items = uni(n_items); w = uni(w); h = uni(h); comp = uni(comp); zo = uni(zo); lfnst = uni(lfnst); ci0 = uni(ci0); ci_step = uni(ci_step); cbf_ctx0 = uni(cbf_ctx0);
  cf_base = uni_p(cf_base); nodes = uni_p(nodes); cf_stride = uni(cf_stride); node_stride = uni(node_stride); abs0 = uni(abs0); cbf_mask = (unsigned) uni((int) cbf_mask); qidx = uni(qidx);
  // the work area (decisions, last-position offsets, template rows, TAB 1 tables) is the LDS memory of wave wk_wave: formed from L here, so that every access in the serial
  // loop is a ds_ instruction (a pointer argument would be a generic one: flat_ accesses that wait on both counters)
  uint8_t *wk = (uint8_t *) &L.wm[uni(wk_wave)];
  const int ch = comp ? 1 : 0, lw = ilog2i(w), lh = ilog2i(h);
  // qidx: another row of the table (joint chroma blocks); chroma rows come from the table of the node's LMCS residual scale (table 0: unscaled)
  const int qrow = uni(qidx) < 0 ? comp : uni(qidx);
  // (the block's quantiser constants are the same on every lane: an LDS load leaves them in 13 vector registers - in the position loop they cost five reloads from scratch
  // per position - so they move to scalar ones)
  VxDqConst q;
  { const VxDqConst &qv = L.par.dq_consts[(qrow ? uni(L.lmcs_tab) * 96 : 0) + qrow * 16 + lw + lh];
    q.qshift = uni(qv.qshift); q.max_qidx = uni(qv.max_qidx); q.thres = uni(qv.thres); q.dshift = uni(qv.dshift);
    q.qadd = dq_uni_ll(qv.qadd); q.qscale = dq_uni_ll(qv.qscale); q.dadd = dq_uni_ll(qv.dadd); q.dstep = dq_uni_ll(qv.dstep); q.dorg = dq_uni_ll(qv.dorg); }
  const ScanGeo geo = scan_geo(w, h);
  const int lcw = geo.lcw, lch = geo.lch, lcg = geo.lcg, gs = 1 << lcg, total = geo.nscan;
  const int nzw = imin(32, w), nzh = imin(32, h), wsbb = geo.wg, hsbb = geo.hg;
  const uint8_t *cg_inv = L.t.cg_inv + cg_tab_off(lcw, lch);
  const uint8_t *grp_inv = L.t.grp_inv + 15 * (wsbb - 1) + wsbb * (hsbb - 1);
  int effW = w, effH = h, zeroOut = 0;
  if (zo && comp == 0) { effH = h == 32 ? 16 : h; effW = w == 32 ? 16 : w; zeroOut = effH < h || effW < w; }
  const int item = lane >> 2, k = lane & 3, qbase = lane & ~3;
  const bool valid = item < n_items;
  DQ_T(t0);
  const int it0 = valid ? item : 0;
  int16_t *cf = cf_base + it0 * cf_stride;
  uint8_t *nd = nodes + (size_t) it0 * node_stride;
  int *lastb = (int *) wk + it0 * 20;
  uint16_t *trel = (uint16_t *) (wk + n_items * 80) + it0 * total;
  // template-sum rows of the item: one per state (written when the state enters a coefficient group) + a row of zeros, 16 sums each
  uint16_t *tmrows = (uint16_t *) (wk + n_items * (80 + 2 * total)) + it0 * 80;
  const int node_room = node_stride ? node_stride : VXD_DQ_WAVE;      // bytes of path nodes one item may use
  VX_CHECK((total >> lcg) * 64 <= node_room);
  if (valid) for (int e = k; e < 80; e += 4) tmrows[e] = 0;
  const int ci = ci0 + it0 * ci_step;
  const int cbf_ctx = cbf_ctx0 < 0 ? -1 : cbf_ctx0 + (int) ((cbf_mask >> it0) & 1u);
  // ---- first tested position of every item (1630-1660): the last scan position whose coefficient exceeds the threshold
  int start = total - 1;
  if (lfnst > 0 && w >= 4 && h >= 4) start = ((w == 4 && h == 4) || (w == 8 && h == 8)) ? 7 : 15;
  int first = -1, top = -1;
  {
    const int thr = q.thres / (int) (4 * q.qscale);
    for (int i = 0; i < n_items; i++) {
      const int16_t *c = cf_base + i * cf_stride;
      int found = -1;
      for (int t = start | 63; t >= 63 && found < 0; t -= 64) {
        const int sp = t - lane;
        bool hit = false;
        if (sp <= start) {
          const int blk = scan_blk(geo, sp), x = blk & (w - 1), y = blk >> lw;
          hit = !(zeroOut && (x >= effW || y >= effH)) && iabs((int) c[blk]) > thr;
        }
        const unsigned long long m = __ballot(hit);
        if (m) found = t - (__ffsll(m) - 1);
      }
      found = uni(found);
      if (item == i) first = found;
      top = imax(top, found);
    }
  }
  DQ_T(t1);
  if (top < 0) {
    for (int i = 0; i < n_items; i++) { int16_t *c = cf_base + i * cf_stride; for (int sp = lane; sp < total; sp += 64) c[scan_blk(geo, sp)] = 0; }
    if (lane < n_items) L.dq_abs[abs0 + lane] = 0;
    wave_sync();
    return;
  }
  DqRate R; R.ci = ci; R.g1Base = VX_CTX_GtxFlag[2 + ch]; R.g2Base = VX_CTX_GtxFlag[ch]; R.parBase = VX_CTX_ParFlag[ch]; R.sbbBase = VX_CTX_SigCoeffGroup[ch];
  const int sigSet = VX_CTX_SigFlag[ch + 2 * imax(k - 1, 0)];          // the state's sig_coeff_flag context set (443-446)
  int *tabBase = TAB == 2 ? L.dq_tab : (int *) (wk + n_items * (240 + 2 * total));
  if (TAB == 1) dq_fill_tables(tabBase, ci, ch, lane, 64);
  const int *levTab = tabBase, *sigTab = tabBase + 21 * 6, *sbbTab = sigTab + 3 * 12 * 2; const uint8_t *riceTab = L.t.rice_len;
  const int *mySig = sigTab + imax(k - 1, 0) * 24;
  // rate of a non-zero level / of the zero run code of the bypass mode / of a group flag, from the tables or from the models
  auto levBits = [&](int gtx_i, int rpar, int lev) -> int {
    if (!TAB) return (int) dq_lev_bits(R, gtx_i, rpar, lev);
    const int b = levTab[gtx_i * 6 + (lev < 4 ? lev : 4 + (lev & 1))];
    const int rb = (int) riceTab[rpar * 32 + imin(31, imax(0, lev - 4) >> 1)] << 15;
    return lev >= 4 ? b + rb : b;
  };
  auto riceBits = [&](int rpar, unsigned v) -> int { return TAB ? (int) riceTab[rpar * 32 + (int) v] << 15 : dq_rice_bits(rpar, v); };
  auto sbbBits = [&](int sbbc, int bin) -> int { return TAB ? sbbTab[sbbc * 2 + bin] : dq_fb(ci, R.sbbBase + sbbc, bin); };
  // ---- last-position offsets per group index (RateEstimator::xSetLastCoeffOffset 488-568): lane k of the quad takes the entries k, k + 4, ...
  if (valid) {
    const int cbfDelta = cbf_ctx >= 0 ? dq_fb(ci, cbf_ctx, 1) - dq_fb(ci, cbf_ctx, 0) : 0;
    for (int e = k; e < 20; e += 4) {
      const int xy = e >= 10, id = xy ? e - 10 : e;
      const int size = xy ? h : w, l2 = ilog2i(size);
      const int base = (xy ? VX_CTX_LastY : VX_CTX_LastX)[ch];
      const int sh = comp == 0 ? (l2 + 1) >> 2 : imin(2, size >> 3), lo = comp == 0 ? L.t.last_prefix[l2] : 0;
      const int maxId = L.t.group_idx[imin(32, size) - 1];
      if (id <= maxId) {
        unsigned sum = 0;
        for (int j = 0; j < id; j++) sum += (unsigned) dq_fb(ci, base + lo + (j >> sh), 1);
        unsigned b = sum + (id > 3 ? (unsigned) ((id - 2) >> 1) << 15 : 0u) + (unsigned) (xy ? cbfDelta : 0);
        if (id < maxId) b += (unsigned) dq_fb(ci, base + lo + (id >> sh), 0);
        lastb[e] = (int) b;
      }
    }
  }
  wave_sync();
  DQ_T(t2);
  const int regFull = (imin(32, effW) * imin(32, effH) * 28) >> 4;
  // The sub-block entry role (skp) is only ever read for its cost, counters and ancestors; when its object comes back into the rotation, what it still
  // holds of levels / template sums is overwritten before an existing path reads it.  It is therefore kept as those members only.
  struct DqK { long long cost; int pk, rem; unsigned long long anc; };
  DqS cur, prv; DqK skp;
  {
    DqS s0; s0.cost = 0x7fffffffffffffffll >> 1; s0.rem = 4; s0.anc = 0;
    s0.pk = dq_put(0, 5, 3, 0);                             // hist -1, everything else 0 (sbbc -1: no bits)
    s0.lev = U4{ 0, 0, 0, 0 };
    cur = s0; prv = s0; skp.cost = s0.cost; skp.pk = s0.pk; skp.rem = s0.rem; skp.anc = 0;
  }
  long long decCost = 0x7fffffffffffffffll >> 2;
  // What depends only on the scan position (the same for every item) is prepared for 64 positions at a time, one per lane, and read back with v_readlane:
  //   pgA: raster offset [0,12), position inside its group [12,16), kind of position [16,18), zeroed by the MTS zero-out [18], last-position group
  //        indices of x [19,23) and y [23,27)
  //   pgB (about the position coded next): position inside its group [0,4), context offsets of its sig flag [4,8) and gt1/par/gt2 set [8,13), number of
  //        template neighbours inside its group [13,16), raster offset [16,28);  pgC: those neighbours' places inside the group, 4 bits each
  // The coefficients of the tested positions are staged in the decision slots first (slot sp holds coefficient sp until the position's decisions replace
  // it): one burst of independent loads per item instead of a dependent (HBM) load in every step of the serial loop
  if (valid) for (int sp = k; sp <= first; sp += 4) trel[sp] = (uint16_t) cf[scan_blk(geo, sp)];
  wave_sync();
  // What a position of a coefficient group sees of the groups right of / below / diagonally below it depends only on the group shape: lane id holds, for
  // position id of a group, 6 bits per template neighbour - which of the three groups (0: inside the group, 1 right, 2 below, 3 diagonal) and the
  // neighbour's place in that group's level bytes
  unsigned eosW = 0;
  if (lane < gs) {
    const int off = cg_tab_off(lcw, lch);
    const int xy = L.t.cg_scan[off + lane], gx = xy & 15, gy = xy >> 4;
#pragma unroll
    for (int n = 0; n < 5; n++) {
      const int xx = gx + (n == 0 ? 1 : n == 1 ? 2 : n == 2 ? 1 : 0), yy = gy + (n == 2 ? 1 : n == 3 ? 1 : n == 4 ? 2 : 0);
      const int ox = (xx >> lcw) != 0, oy = (yy >> lch) != 0;
      if (ox || oy) eosW |= (unsigned) ((ox ? (oy ? 3 : 1) : 2) | ((int) cg_inv[((yy & ((1 << lch) - 1)) << lcw) | (xx & ((1 << lcw) - 1))] << 2)) << (6 * n);
    }
  }
  int cnext = 0;
#ifdef VVCX_STAMP_DQ
  long long sA = 0, sB = 0, sC = 0, sE = 0, u1 = 0, u2 = 0, u3 = 0;
#endif
  // the position loop is the longest serial chain of a stream and runs on one or two of the workgroup's waves while the others wait at the barrier: it takes issue
  // priority over the waves of the CU's other streams that are in their wide phases (VALU issue is arbitrated by priority, then by age)
  __builtin_amdgcn_s_setprio(VXD_SERIAL_PRIO);
  for (int top64 = top; top64 >= 0; top64 -= 64) {
  int pgA = 0, pgB = 0, pgC = 0;
  {
    const int sp = top64 - lane;
    if (sp >= 0) {
      const int blk = scan_blk(geo, sp), x = blk & (w - 1), y = blk >> lw, inside = sp & (gs - 1);
      int spt = 0;
      if (inside == gs - 1 && sp > gs && sp < total - 1) spt = 1; else if (inside == 0 && sp > 0 && sp < total - gs) spt = 2;
      pgA = blk | (inside << 12) | (spt << 16) | ((zeroOut && (x >= effW || y >= effH)) ? 1 << 18 : 0) | ((int) L.t.group_idx[x] << 19) | ((int) L.t.group_idx[y] << 23);
      if (sp > 0) {
        const int nb = scan_blk(geo, sp - 1), xn = nb & (w - 1), yn = nb >> lw, diag = xn + yn;
        int cnt = 0;
#pragma unroll
        for (int n = 0; n < 5; n++) {
          const int xx = xn + (n == 0 ? 1 : n == 1 ? 2 : n == 2 ? 1 : 0), yy = yn + (n == 2 ? 1 : n == 3 ? 1 : n == 4 ? 2 : 0);
          if (xx < nzw && yy < nzh && (xx >> lcw) == (xn >> lcw) && (yy >> lch) == (yn >> lch)) { pgC |= (int) cg_inv[((yy & ((1 << lch) - 1)) << lcw) | (xx & ((1 << lcw) - 1))] << (4 * cnt); cnt++; }
        }
        pgB = ((sp - 1) & (gs - 1)) | (dq_sig_off(ch, diag) << 4) | (dq_gtx_off(ch, diag) << 8) | (cnt << 13) | (nb << 16);
      }
    }
  }
  if (top64 == top) cnext = (valid && top <= first) ? (int) (int16_t) trel[top] : 0;       // coefficient of the position about to be processed (fetched one position ahead)
  const int nchunk = imin(64, top64 + 1);
  for (int it = 0; it < nchunk; it++) {
    const int sp = top64 - it;
    DQ_T(u0); DQ_U(u1);
    const int gA = __builtin_amdgcn_readlane(pgA, it), gB = __builtin_amdgcn_readlane(pgB, it), nbl = __builtin_amdgcn_readlane(pgC, it);
    const int inside = (gA >> 12) & 15, eos = inside == 0, spt = (gA >> 16) & 3, zeroed = (gA >> 18) & 1;
    const int nin = gB & 15, sigOffN = (gB >> 4) & 15, gtxOffN = (gB >> 8) & 31, nbCnt = (gB >> 13) & 7;
    const bool act = valid && sp <= first;                   // the item's trellis has started
    const int absC = iabs(cnext);
    if (valid && sp > 0 && sp - 1 <= first) cnext = (int) (int16_t) trel[sp - 1];
    { const DqS t = prv; prv = cur; cur = t; }
    // ---- decision of target state k (xDecide 1455-1517)
    long long dc = 0x7fffffffffffffffll >> 2; int dsrc = 0, dnz = 0, dlev = -1;          // dsrc: 0 none, 1 start, 2 from the "A/zero" source state, 3 from the "B" source state, 4 sub-block skipped
    if (zeroed) {
      if (spt == 2) { dc = skp.cost + (DQ_SBBC(skp.pk) >= 0 ? sbbBits(DQ_SBBC(skp.pk), 0) : 0); dsrc = 4; dlev = 0; }
    } else {
      const long long scaledOrg = (long long) absC * q.qscale;
      const int qIdx0 = dq_qidx0(q, scaledOrg);
#define DQ_STEP(i_) (((i_) - qIdx0) & 3)
#define DQ_LEV(st_) ((qIdx0 + (st_) + 1) >> 1)
      // this lane as SOURCE state k: candidates A / B / zero (checkRdCosts 918-1030)
      const int stA = DQ_STEP(k < 2 ? 0 : 3), stB = DQ_STEP(k < 2 ? 2 : 1);
      const int levA = DQ_LEV(stA), levB = DQ_LEV(stB);
      // the four candidates of the position (preQuantCoeff's pqData[0..3]) one per lane of the quad; a state's A / B pair: 0 / 2 (states 0, 1), 3 / 1 (states 2, 3)
      const long long ddMine = dq_dd(q, scaledOrg, qIdx0, DQ_STEP(k));
      const long long ddA = dq_quad_i64<0xF0>(ddMine), ddB = dq_quad_i64<0x5A>(ddMine);      // quad_perm [0,0,3,3] / [2,2,1,1]
      long long cA = prv.cost + ddA, cB = prv.cost + ddB, cZ = prv.cost;
      {
        const int ppk = prv.pk, rpar = DQ_RPAR(ppk);
        if (prv.rem >= 4) {
          cA += levBits(DQ_GTXI(ppk), rpar, levA); cB += levBits(DQ_GTXI(ppk), rpar, levB);
          int s0b, s1b;
          if (TAB) { const long long sb2 = *(const long long *) (mySig + 2 * DQ_SIGI(ppk)); s0b = (int) sb2; s1b = (int) (sb2 >> 32); }
          else { s0b = dq_fb(ci, sigSet + DQ_SIGI(ppk), 0); s1b = dq_fb(ci, sigSet + DQ_SIGI(ppk), 1); }
          if (spt == 0) { cA += s1b; cB += s1b; cZ += s0b; }
          else if (spt == 1) { const int sb = DQ_SBBC(ppk) >= 0 ? sbbBits(DQ_SBBC(ppk), 1) : 0; cA += sb + s1b; cB += sb + s1b; cZ += sb + s0b; }
          else if (DQ_NUMSIG(ppk)) { cA += s1b; cB += s1b; cZ += s0b; }
          else cZ = 0x7fffffffffffffffll;                   // a group whose flag is coded as significant cannot end all-zero: no zero candidate
        } else {
          const int rz = DQ_RZERO(ppk);
          cA += (1 << 15) + riceBits(rpar, (unsigned) (levA <= rz ? levA - 1 : (levA < 32 ? levA : 31)));
          cB += (1 << 15) + riceBits(rpar, (unsigned) (levB <= rz ? levB - 1 : (levB < 32 ? levB : 31)));
          cZ += riceBits(rpar, (unsigned) imin(31, rz));
        }
      }
      DQ_U(u1);
      // gather: target 0 <- {A, Z of state 0; B of state 1}, target 2 <- {B of 0; A, Z of 1}, target 1 <- {A, Z of 2; B of 3}, target 3 <- {B of 2; A, Z of 3}
      const long long azA = dq_quad_i64<0xD8>(cA), azZ = dq_quad_i64<0xD8>(cZ), bB = dq_quad_i64<0x8D>(cB);      // quad_perm [0,2,1,3] / [1,3,0,2]
      const int tLevA = DQ_LEV(DQ_STEP((k & 1) ? 3 : 0)), tLevB = DQ_LEV(DQ_STEP((k & 1) ? 1 : 2));
      {
        // the reference tests A, zero, B for the targets 0 / 2 and B, A, zero for 1 / 3, each with a strict <: the first of equal costs stays.  One sequence for
        // both: the better of A and zero (A on a tie), then B, which takes a tie only where it is tested first
        long long m = azA; int msrc = 2, mnz = 1, mlev = tLevA;
        if (azZ < m) { m = azZ; mnz = 0; mlev = 0; }
        const bool bWins = k < 2 ? bB < m : bB <= m;
        if (bWins) { m = bB; msrc = 3; mnz = 1; mlev = tLevB; }
        if (m < dc) { dc = m; dsrc = msrc; dnz = mnz; dlev = mlev; }
      }
      if (spt == 2) {                                       // checkRdCostSkipSbb 1052-1061
        const long long c = skp.cost + (DQ_SBBC(skp.pk) >= 0 ? sbbBits(DQ_SBBC(skp.pk), 0) : 0);
        if (c < dc) { dc = c; dsrc = 4; dnz = 0; dlev = 0; }
      }
      if (!(k & 1)) {                                       // checkRdCostStart 1032-1050 into decisions 0 (candidate 0) and 2 (candidate 2)
        const int stS = DQ_STEP(k), levS = DQ_LEV(stS);
        const long long c = ddMine + (long long) (lastb[(gA >> 19) & 15] + lastb[10 + ((gA >> 23) & 15)]) + levBits(0, 0, levS);
        if (c < dc) { dc = c; dsrc = 1; dnz = 1; dlev = levS; }
      }
#undef DQ_STEP
#undef DQ_LEV
    }
    {                                                       // the four decisions of the position: 4 bits each
      int e = ((dsrc << 1) | dnz) << (4 * k);
      e = seg_sum<4>(e);
      VX_CHECK(sp >= 0 && sp < total);
      if (k == 0 && act) trel[sp] = (uint16_t) e;
    }
    if (act) decCost = dc;
    DQ_U(u2);
    if (sp == 0) break;
    // ---- state update (xDecideAndUpdate 1527-1588)
    if (eos || !zeroed) {
      // parent state (a previous state of the quad) — every lane shuffles, the lanes whose decision has no such parent discard the result
      // (the "A/zero" source of target k is lane [0,2,1,3][k] of the quad, the "B" source lane [1,3,0,2][k]: two quad permutations and a select)
      struct { int pk, rem; unsigned long long anc; U4 lev; } P;
      {
        const bool fromA = dsrc == 2;
#define DQ_PARENT(x_) dq_parent((int) (x_), fromA)
        P.pk = DQ_PARENT(prv.pk); P.rem = DQ_PARENT(prv.rem);
        P.anc = (unsigned long long) (unsigned) DQ_PARENT((unsigned) prv.anc) | ((unsigned long long) (unsigned) DQ_PARENT((unsigned) (prv.anc >> 32)) << 32);
        P.lev.a = (unsigned) DQ_PARENT(prv.lev.a); P.lev.b = (unsigned) DQ_PARENT(prv.lev.b); P.lev.c = (unsigned) DQ_PARENT(prv.lev.c); P.lev.d = (unsigned) DQ_PARENT(prv.lev.d);
#undef DQ_PARENT
      }
      const bool alive = act && dsrc != 0, fromPrev = dsrc == 2 || dsrc == 3;      // dsrc 1: the path starts here, 4: it skipped the group it leaves
      if (act) cur.cost = dc;
      if (!eos) {                                           // State::updateState 1109-1273
        // the template sums travel with the path; a lane without a parent in the quad starts from zeros (a path that does not exist never reads them)
        if (alive) {
          // (selects instead of two paths: in most positions some lane of the wave continues a path and another starts one)
          int pk = cur.pk;
          {
            // continuing: counters of the path (non-zero levels + this one, state at the last group change, group-flag context) from the parent;
            // starting: one non-zero level, no history.  The Rice parameter, contexts and template row are set below in both cases.
            const int lowNew = fromPrev ? ((P.pk & ((7 << 5) | (3 << 16))) | ((P.pk + (dlev != 0)) & 31)) : 1;
            pk = (pk & (fromPrev ? ~(0xFF | (3 << 16)) : ~0xFF)) | lowNew;
            int rem = fromPrev ? P.rem - 1 : regFull;
            rem -= rem >= 4 ? (dlev < 2 ? dlev : 3) : 0;
            cur.rem = rem; cur.anc = fromPrev ? P.anc : 0ull;
            cur.lev.a = fromPrev ? P.lev.a : 0u; cur.lev.b = fromPrev ? P.lev.b : 0u; cur.lev.c = fromPrev ? P.lev.c : 0u; cur.lev.d = fromPrev ? P.lev.d : 0u;
          }
          dq_set_b(cur.lev, inside, (unsigned) imin(255, dlev));
          // template of the next position: its neighbours inside the group (m_scanId2NbInfoSbb) on top of the sums over those outside
          // the sums over the neighbours outside the group travel with the path as the row its group entry state wrote; a path that starts here has none (row 4: zeros)
          const int trow = fromPrev ? ((P.pk >> 27) & 7) : 4;
          VX_CHECK(trow <= 4 && nin < 16);
          pk = (pk & ~(7 << 27)) | (trow << 27);
          const int t = (int) tmrows[trow * 16 + nin];
          int sumAbs = t >> 8, sumAbs1 = (t >> 3) & 31, sumNum = t & 7;
          {
            // (the neighbours' places are the same for every lane: register-indexed reads instead of select trees)
            typedef unsigned dq_v4u __attribute__((vector_size(16)));
            const dq_v4u cl = { cur.lev.a, cur.lev.b, cur.lev.c, cur.lev.d };
#pragma unroll
            for (int n = 0; n < 5; n++) if (n < nbCnt) {
              const int bi = (nbl >> (4 * n)) & 15;
              const int a = (int) ((cl[bi >> 2] >> ((bi & 3) << 3)) & 255u);
              sumAbs += a; sumAbs1 += imin(4 + (a & 1), a); sumNum += a != 0;
            }
          }
          if (cur.rem >= 4) {
            // (g_auiGoRiceParsCoeff[clamp(sumAbs - 20, 0, 31)] steps at 7, 14 and 28: three compares instead of a dependent LDS read)
            pk = (pk & ~((3 << 8) | (0x1FF << 18))) | (((sumAbs >= 27) + (sumAbs >= 34) + (sumAbs >= 48)) << 8) | ((sigOffN + imin((sumAbs1 + 1) >> 1, 3)) << 18)
                 | ((gtxOffN + imin(sumAbs1 - sumNum, 4)) << 22);
          } else {
            sumAbs = imin(31, sumAbs);
            pk = dq_put(pk, 8, 2, L.t.gorice_pars[sumAbs]); pk = dq_put(pk, 10, 6, L.t.gorice_pos0[imax(0, k - 1) * 32 + sumAbs]);
          }
          cur.pk = pk;
        }
      } else {                                              // State::updateStateEOS 1275-1315 + CommonCtx::update 1317-1398
        const int g = sp >> lcg;
        int pk = cur.pk; int numSig, pRem; unsigned long long pAnc; U4 lv = { 0, 0, 0, 0 };
        if (dsrc == 4) { numSig = 0; pRem = skp.rem; pAnc = skp.anc; }
        else if (fromPrev) { numSig = DQ_NUMSIG(P.pk) + (dlev != 0); pRem = P.rem; pAnc = P.anc; lv = P.lev; }
        else { numSig = 1; pRem = regFull; pAnc = 0; }
        dq_set_b(lv, 0, (unsigned) imin(255, imax(dlev, 0)));
        // the path as it leaves this group: field 0 = this state and the group's significance, the older groups one field up
        const unsigned long long anc = (pAnc << 4) | (unsigned long long) (unsigned) (k + 1) | ((numSig != 0) ? 8ull : 0ull);
        VX_CHECK(g >= 1 && (g * 4 + k) * 16 + 16 <= node_room);
        if (alive) { VX_GLOBAL uint32_t *hl = (VX_GLOBAL uint32_t *) (nd + (size_t) (g * 4 + k) * 16); hl[0] = lv.a; hl[1] = lv.b; hl[2] = lv.c; hl[3] = lv.d; }
        wave_sync();
        // the groups right of, below and diagonally below the next group: their distance in group-scan order picks the ancestor field
        const unsigned ng = geo.grp[g - 1]; const int nsx = (int) (ng & 15), nsy = (int) (ng >> 4);
        const int gR = nsx < wsbb - 1 ? grp_inv[nsy * wsbb + nsx + 1] : -1, gB = nsy < hsbb - 1 ? grp_inv[(nsy + 1) * wsbb + nsx] : -1;
        const int gD = (gR >= 0 && gB >= 0) ? grp_inv[(nsy + 1) * wsbb + nsx + 1] : -1;
        const unsigned fR = gR >= 0 ? (unsigned) (anc >> (4 * (gR - g))) & 15u : 0u, fB = gB >= 0 ? (unsigned) (anc >> (4 * (gB - g))) & 15u : 0u;
        const unsigned fD = gD >= 0 ? (unsigned) (anc >> (4 * (gD - g))) & 15u : 0u;
        const int sigN = ((fR & 8u) || (fB & 8u)) ? 1 : 0;
        // the three nodes (16 level bytes each) whose levels the next group's templates look at
        // (a group beyond the block's edge - the coded 32 x 32 region's for bigger blocks - is an all-zero node: it adds nothing to the sums)
        typedef unsigned dq_v16u __attribute__((vector_size(64)));
        dq_v16u nv = { 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0 };      // [4..8) right, [8..12) below, [12..16) diagonal
        if (alive) {
          VX_CHECK(!(fR & 7u) || (gR >= g && gR - g < 16 && (gR * 4 + (int) (fR & 7u) - 1) * 16 + 16 <= node_room));
          VX_CHECK(!(fB & 7u) || (gB >= g && gB - g < 16 && (gB * 4 + (int) (fB & 7u) - 1) * 16 + 16 <= node_room));
          VX_CHECK(!(fD & 7u) || (gD >= g && gD - g < 16 && (gD * 4 + (int) (fD & 7u) - 1) * 16 + 16 <= node_room));
          if (fR & 7u) { const VX_GLOBAL uint32_t *p_ = (const VX_GLOBAL uint32_t *) (nd + (size_t) (gR * 4 + (int) (fR & 7u) - 1) * 16); nv[4] = p_[0]; nv[5] = p_[1]; nv[6] = p_[2]; nv[7] = p_[3]; }
          if (fB & 7u) { const VX_GLOBAL uint32_t *p_ = (const VX_GLOBAL uint32_t *) (nd + (size_t) (gB * 4 + (int) (fB & 7u) - 1) * 16); nv[8] = p_[0]; nv[9] = p_[1]; nv[10] = p_[2]; nv[11] = p_[3]; }
          if (fD & 7u) { const VX_GLOBAL uint32_t *p_ = (const VX_GLOBAL uint32_t *) (nd + (size_t) (gD * 4 + (int) (fD & 7u) - 1) * 16); nv[12] = p_[0]; nv[13] = p_[1]; nv[14] = p_[2]; nv[15] = p_[3]; }
        }
        for (int id = 0; id < gs; id++) {
          const unsigned wrd = (unsigned) __builtin_amdgcn_readlane((int) eosW, id);
          int sumAbs = 0, sumAbs1 = 0, sumNum = 0;
#pragma unroll
          for (int n = 0; n < 5; n++) {
            const unsigned f = (wrd >> (6 * n)) & 63u;
            if (f & 3u) {
              const int a = (int) ((nv[(f & 3u) * 4 + (f >> 4)] >> (((f >> 2) & 3u) << 3)) & 255u);
              sumAbs += a; sumAbs1 += imin(4 + (a & 1), a); sumNum += a != 0;
            }
          }
          if (alive) tmrows[k * 16 + id] = (uint16_t) (sumNum + (sumAbs1 << 3) + (imin(127, sumAbs) << 8));      // row k: this state's entry into the next group
        }
        if (alive) {
          pk = dq_put(pk, 0, 5, 0); pk = dq_put(pk, 8, 2, 0); pk = dq_put(pk, 5, 3, k + 1); pk = dq_put(pk, 16, 2, sigN + 1);
          cur.rem = pRem; cur.anc = anc;
          cur.lev = U4{ 0, 0, 0, 0 };
          pk = dq_put(pk, 27, 3, k);
          const int t = (int) tmrows[k * 16 + nin];
          const int sumAbs1 = (t >> 3) & 31, sumNum = t & 7;
          pk = dq_put(pk, 18, 4, sigOffN + imin((sumAbs1 + 1) >> 1, 3)); pk = dq_put(pk, 22, 5, gtxOffN + imin(sumAbs1 - sumNum, 4));
          cur.pk = pk;
        }
      }
    }
    DQ_U(u3);
#ifdef VVCX_STAMP_DQ
    sA += u1 - u0; sB += u2 - u1; sC += u3 - u2; if (eos) sE += u3 - u2;
#endif
    if (spt == 1) { const DqK t = skp; skp.cost = prv.cost; skp.pk = prv.pk; skp.rem = prv.rem; skp.anc = prv.anc; prv.cost = t.cost; prv.pk = t.pk; prv.rem = t.rem; prv.anc = t.anc; }
  }
  }
  __builtin_amdgcn_s_setprio(0);
  DQ_T(t3);
  // ---- best final state and back-tracking (1709-1730), lane 0 of every quad for its item
  int prev = -2; long long minCost = 0;
#pragma unroll
  for (int s = 0; s < 4; s++) { const long long c = dq_shfl_i64(decCost, qbase + s); if (c < minCost) { prev = s; minCost = c; } }
  int nPath = 0;
  if (k == 0 && valid) {
    int sp = 0, absSum = 0;
    if (first < 0) prev = -2;
    while (prev >= 0) {
      VX_CHECK(sp >= 0 && sp <= first && prev < 4);
      const int e = (trel[sp] >> (4 * prev)) & 15, src = e >> 1;
      int lev = 0;
      if (e & 1) {
        const long long scaledOrg = (long long) iabs((int) cf[scan_blk(geo, sp)]) * q.qscale;
        const int i = src == 1 ? prev : src == 2 ? ((prev & 1) ? 3 : 0) : ((prev & 1) ? 1 : 2);
        const int qIdx0 = dq_qidx0(q, scaledOrg);
        lev = (qIdx0 + ((i - qIdx0) & 3) + 1) >> 1;
      }
      trel[sp] = (uint16_t) lev; absSum += lev;
      if (src == 4) { for (int j = 1; j < gs; j++) trel[sp + j] = 0; sp += gs; }
      else { prev = src == 1 ? -1 : src == 2 ? ((prev & 1) ? (prev == 1 ? 2 : 3) : (prev == 0 ? 0 : 1)) : src == 3 ? ((prev & 1) ? (prev == 1 ? 3 : 2) : (prev == 0 ? 1 : 0)) : -2; sp++; }
    }
    nPath = sp;
    L.dq_abs[abs0 + item] = absSum;
  }
  nPath = __shfl(nPath, qbase);
  wave_sync();
  DQ_T(t4);
  for (int i = 0; i < n_items; i++) {                      // all lanes on one item after the other
    const int np = __builtin_amdgcn_readlane(nPath, i * 4);
    int16_t *c_ = cf_base + i * cf_stride; const uint16_t *tr = (const uint16_t *) (wk + n_items * 80) + i * total;
    for (int sp = lane; sp < total; sp += 64) {
      const int blk = scan_blk(geo, sp), c = c_[blk];
      const int lv = sp < np ? (int) tr[sp] : 0;
      c_[blk] = (int16_t) (c < 0 ? -lv : lv);
    }
  }
  wave_sync();
#ifdef VVCX_STAMP_DQ
  { const long long t5 = clock64(); int fs = valid && k == 0 ? first + 1 : 0; for (int o = 32; o; o >>= 1) fs += __shfl_xor(fs, o);
    if (lane == 0) { const long long v[14] = { t1 - t0, t2 - t1, t3 - t2, t4 - t3, t5 - t4, 1, top + 1, n_items, fs, total, sA, sB, sC, sE };
      for (int i = 0; i < 14; i++) atomicAdd(&L.prof[16 + i], (unsigned long long) v[i]); } }
#endif
}

// one block of the calling wave's own candidate: decisions in the wave's rate-estimator scratch and tmp (both free at that point of wave_code_block:
// the transform's first stage has been consumed), path nodes in the wave's HBM area
#ifndef VXD_DQ_TAB_MIN
#define VXD_DQ_TAB_MIN 16      // positions from which a single-block trellis tabulates its rates (measured: 64 -> 16 is +1.8 % on the bench; the small TUs of ISP CUs)
#endif
template <bool SMALL>
__device__ inline int wave_depquant(int16_t *cf_g, int buf_off, uint8_t *scratch, int ci, int w, int h, int comp, int cbf_ctx, int zo, int lfnst, int lane, int qidx = -1)
{
  const int wave_ = uni(VTX >> 6);
  int16_t *cf = SMALL ? L.wm[wave_].slot + BUF + uni(buf_off) : cf_g;
  // the tables fit behind the decisions in the rate-estimator scratch + tmp for all but the 32 x 32-coefficient blocks
  // (and pay for themselves from 64 positions on)
  if (imin(32, w) * imin(32, h) >= VXD_DQ_TAB_MIN && 240 + 2 * imin(32, w) * imin(32, h) + DQ_TAB_BYTES <= (int) (sizeof(WaveScratch) + sizeof(int32_t) * BUF))      // (2000 bytes: up to 256 positions)
    wave_depquant_batch<1>(1, cf, 0, scratch + VXD_OFF_DQ + (size_t) wave_ * VXD_DQ_WAVE, 0, wave_, DQ_ABS_SINGLE + wave_, ci, 0, cbf_ctx, 0u, w, h, comp, zo, lfnst, lane, qidx);
  else
    wave_depquant_batch<0>(1, cf, 0, scratch + VXD_OFF_DQ + (size_t) wave_ * VXD_DQ_WAVE, 0, wave_, DQ_ABS_SINGLE + wave_, ci, 0, cbf_ctx, 0u, w, h, comp, zo, lfnst, lane, qidx);
  return uni(L.dq_abs[DQ_ABS_SINGLE + wave_]);
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

/*
 * orc_depquant.c — dependent (trellis-coded) quantisation of one transform block (TEST INFRASTRUCTURE ONLY, see vvc_oracle.h).
 *
 * Restates CL/DepQuant.cpp: DQIntern::DepQuant::quant 1592-1731 (first tested position, trellis, back-tracking), xDecide 1455-1517,
 * xDecideAndUpdate 1519-1589, State::checkRdCosts 918-1030 / checkRdCostStart 1032 / checkRdCostSkipSbb 1052-1069,
 * State::updateState 1109-1273, updateStateEOS 1275-1315, CommonCtx::update 1317-1398, Quantizer::initQuantBlock 694-739,
 * preQuantCoeff 812-832, dequantBlock 741-810, RateEstimator 479-618, the neighbour tables of Rom::xInitScanArrays 153-303 and
 * TUParameters::xSetScanInfo 377-431.  Flat scaling (no scaling lists), no extended precision.
 *
 * The reference keeps twelve State objects in three roles (current / previous / sub-block entry) that swap every scan position, and an
 * update writes only some members; a few of the others (m_goRiceZero after a sub-block change while the regular-bin budget is spent) are
 * read later with whatever the object held before.  The slots and their rotation are therefore kept exactly (S.slot, bases cur/prv/skp).
 * Pinned against the reference's DepQuant through TrQuant::transformNxN (tests/golden/depquant.npz).
 */
#include "orc_internal.h"
#include <stdlib.h>

static int ilog2(int v) { int r = 0; while (v > 1) { v >>= 1; r++; } return r; }
static int imin(int a, int b) { return a < b ? a : b; }
static int imax(int a, int b) { return a > b ? a : b; }

/* ---- geometry of a block shape: scan, template neighbours inside / outside the coefficient group (153-303), group scan ---- */
typedef struct {
  int ready, w, h, lcg, gs, nzw, nzh, total, nsbb, wsbb, hsbb;
  uint16_t scan[1024]; uint8_t px[1024], py[1024];
  uint8_t in_num[1024], in_pos[1024][5];
  uint8_t out_num[1024]; uint16_t out_pos[1024][5], max_dist[1024];
  uint8_t sbb_raster[64];            /* group scan index -> raster position of the group */
} dq_geo;
static dq_geo *g_geo[7][7];

static void diag_order(int bw, int bh, uint8_t *xs, uint8_t *ys)          /* CL/Rom.cpp:87-131 */
{
  int line = 0, col = 0;
  for (int n = 0; n < bw * bh; n++) {
    xs[n] = (uint8_t) col; ys[n] = (uint8_t) line;
    if (col == bw - 1 || line == 0) { line += col + 1; col = 0; if (line >= bh) { col += line - (bh - 1); line = bh - 1; } }
    else { col++; line--; }
  }
}
static const dq_geo *geo_of(int w, int h)
{
  const int lw = ilog2(w), lh = ilog2(h);
  if (g_geo[lw][lh]) return g_geo[lw][lh];
  dq_geo *g = (dq_geo *) calloc(1, sizeof *g);
  int lcw, lch; orc_cg_shape(w, h, &lcw, &lch);
  g->w = w; g->h = h; g->lcg = lcw + lch; g->gs = 1 << g->lcg;
  g->nzw = imin(32, w); g->nzh = imin(32, h); g->total = g->nzw * g->nzh;
  g->wsbb = g->nzw >> lcw; g->hsbb = g->nzh >> lch; g->nsbb = g->wsbb * g->hsbb;
  orc_scan_order(w, h, g->scan);
  int *r2id = (int *) calloc((size_t) w * h, sizeof(int));
  for (int s = 0; s < g->total; s++) { r2id[g->scan[s]] = s; g->px[s] = (uint8_t) (g->scan[s] % w); g->py[s] = (uint8_t) (g->scan[s] / w); }
  { uint8_t xs[64], ys[64]; diag_order(g->wsbb, g->hsbb, xs, ys); for (int i = 0; i < g->nsbb; i++) g->sbb_raster[i] = (uint8_t) (ys[i] * g->wsbb + xs[i]); }
  static const int DX[5] = { 1, 2, 1, 0, 0 }, DY[5] = { 0, 0, 1, 1, 2 };
  for (int s = 0; s < g->total; s++) {
    const int beg = s & ~(g->gs - 1), x = g->px[s], y = g->py[s];
    int in[5], out[5];
    for (int k = 0; k < 5; k++) {
      in[k] = out[k] = 0;
      if (x + DX[k] < g->nzw && y + DY[k] < g->nzh) { const int id = r2id[g->scan[s] + DX[k] + DY[k] * w]; if (id < beg + g->gs) in[k] = id - beg; else out[k] = id; }
    }
    int n = 0;
    for (;;) { int nk = -1; for (int k = 0; k < 5; k++) if (in[k] && (nk < 0 || in[k] < in[nk])) nk = k; if (nk < 0) break; g->in_pos[s][n++] = (uint8_t) in[nk]; in[nk] = 0; }
    g->in_num[s] = (uint8_t) n;
    n = 0;
    for (;;) { int nk = -1; for (int k = 0; k < 5; k++) if (out[k] && (nk < 0 || out[k] < out[nk])) nk = k; if (nk < 0) break; g->out_pos[s][n++] = (uint16_t) out[nk]; out[nk] = 0; }
    g->out_num[s] = (uint8_t) n;
    int md = s == 0 ? 0 : g->max_dist[s - 1];              /* running maximum of absolute positions (272-279) ... */
    for (int k = 0; k < n; k++) if (g->out_pos[s][k] > md) md = g->out_pos[s][k];
    g->max_dist[s] = (uint16_t) md;
  }
  for (int s = 0; s < g->total; s++) {                     /* ... then relative: positions to the group start, the maximum to the position itself (284-294) */
    const int beg = s & ~(g->gs - 1);
    for (int k = 0; k < g->out_num[s]; k++) g->out_pos[s][k] = (uint16_t) (g->out_pos[s][k] - beg);
    g->max_dist[s] = (uint16_t) (g->max_dist[s] - s);
  }
  free(r2id);
  g->ready = 1; g_geo[lw][lh] = g;
  return g;
}

/* ---- rate tables from the estimator's current contexts (RateEstimator::initCtx 479-618) ---- */
typedef struct { int32_t sbb[2][2], sig[3][12][2], gtx[21][6], last_x[32], last_y[32]; } dq_rate;
static inline const uint32_t *frac_of(const orc_cabac *c, int ctx) { return &ORC_BIN_FRAC_BITS[((unsigned) (c->s0[ctx] + c->s1[ctx]) >> 8) * 2]; }
static void rate_init(dq_rate *r, const orc_cabac *c, int w, int h, int comp, int cbf_ctx)
{
  const int ch = comp ? 1 : 0;
  for (int i = 0; i < 2; i++) { const uint32_t *f = frac_of(c, ORC_CTX_SigCoeffGroup[ch] + i); r->sbb[i][0] = (int32_t) f[0]; r->sbb[i][1] = (int32_t) f[1]; }
  for (int set = 0; set < 3; set++) for (int i = 0; i < (ch ? 8 : 12); i++) { const uint32_t *f = frac_of(c, ORC_CTX_SigFlag[ch + 2 * set] + i); r->sig[set][i][0] = (int32_t) f[0]; r->sig[set][i][1] = (int32_t) f[1]; }
  for (int i = 0; i < (ch ? 11 : 21); i++) {
    const uint32_t *par = frac_of(c, ORC_CTX_ParFlag[ch] + i), *g1 = frac_of(c, ORC_CTX_GtxFlag[2 + ch] + i), *g2 = frac_of(c, ORC_CTX_GtxFlag[ch] + i);
    const int32_t par0 = (1 << 15) + (int32_t) par[0], par1 = (1 << 15) + (int32_t) par[1];
    int32_t *b = r->gtx[i];
    b[0] = 0; b[1] = (int32_t) g1[0] + (1 << 15);
    b[2] = (int32_t) g1[1] + par0 + (int32_t) g2[0]; b[3] = (int32_t) g1[1] + par1 + (int32_t) g2[0];
    b[4] = (int32_t) g1[1] + par0 + (int32_t) g2[1]; b[5] = (int32_t) g1[1] + par1 + (int32_t) g2[1];
  }
  /* xSetLastCoeffOffset 488-568: cbf_ctx < 0 = the cbf is inferred (last ISP sub-partition after all-zero ones) */
  int32_t cbfDelta = 0;
  if (cbf_ctx >= 0) { const uint32_t *f = frac_of(c, cbf_ctx); cbfDelta = (int32_t) f[1] - (int32_t) f[0]; }
  static const int prefix_ctx[8] = { 0, 0, 0, 3, 6, 10, 15, 21 };
  for (int xy = 0; xy < 2; xy++) {
    const int32_t off = xy ? cbfDelta : 0;
    int32_t *lb = xy ? r->last_y : r->last_x;
    const int size = xy ? h : w, l2 = ilog2(size);
    const int base = (xy ? ORC_CTX_LastY : ORC_CTX_LastX)[ch];
    const int sh = comp == 0 ? (l2 + 1) >> 2 : imin(2, imax(0, size >> 3)), lo = comp == 0 ? prefix_ctx[l2] : 0;
    uint32_t sum = 0, bits[16];
    const int maxId = ORC_GROUP_IDX[imin(32, size) - 1];
    for (int id = 0; id < maxId; id++) {
      const uint32_t *f = frac_of(c, base + lo + (id >> sh));
      bits[id] = sum + f[0] + (id > 3 ? (uint32_t) ((id - 2) >> 1) << 15 : 0) + (uint32_t) off;
      sum += f[1];
    }
    bits[maxId] = sum + (maxId > 3 ? (uint32_t) ((maxId - 2) >> 1) << 15 : 0) + (uint32_t) off;
    for (int pos = 0; pos < imin(32, size); pos++) lb[pos] = (int32_t) bits[ORC_GROUP_IDX[pos]];
  }
}

/* ---- quantiser constants (Quantizer::initQuantBlock 694-739) ---- */
typedef struct { int qshift; int64_t qadd, qscale; int max_qidx, thres_last; int dshift; int64_t dadd, dstep, dorg; } dq_quant;
static int ceil_log2_u64(uint64_t x) { int y = (x & (x - 1)) ? 1 : 0; while (x > 1) { x >>= 1; y++; } return y; }     /* 680-693 */
void orc_depquant_consts(int w, int h, int bit_depth, int qp, double lambda, int64_t out[9])
{
  const int lw = ilog2(w), lh = ilog2(h), sq = (lw + lh) & 1;
  const int qpDQ = qp + 1, per = qpDQ / 6, rem = qpDQ - 6 * per;
  const int nomShift = 15 - bit_depth - ((lw + lh) >> 1), trShift = nomShift + (sq ? -1 : 0);
  const int qshift = 14 - 1 + per + trShift;
  const int64_t qscale = ORC_QUANT_SCALES[sq * 6 + rem];
  const int invShift = 6 + 1 - per - trShift;
  const unsigned qIdxBD = (unsigned) imin(15 + 1, 32 + invShift - 6 - 1);
  const int nomDShift = 15 - 2 * nomShift + qshift + (sq ? 1 : 0);
  const double qScale2 = (double) (qscale * qscale);
  const double nomDistFactor = nomDShift < 0 ? 1.0 / ((double) ((int64_t) 1 << (-nomDShift)) * qScale2 * lambda) : (double) ((int64_t) 1 << nomDShift) / (qScale2 * lambda);
  const int64_t pow2dfShift = (int64_t) (nomDistFactor * qScale2) + 1;
  const int dfShift = ceil_log2_u64((uint64_t) pow2dfShift);
  const int dshift = 62 + qshift - 2 * 15 - dfShift;
  out[0] = qshift; out[1] = -(((int64_t) 3 << qshift) >> 1); out[2] = qscale; out[3] = (1 << (qIdxBD - 1)) - 4;
  out[4] = (int) ((int64_t) 4 << qshift);
  out[5] = dshift; out[6] = ((int64_t) 1 << dshift) >> 1;
  out[7] = (int64_t) (nomDistFactor * (double) ((int64_t) 1 << (dshift + qshift)) + .5);
  out[8] = (int64_t) (nomDistFactor * (double) ((int64_t) 1 << (dshift + 1)) + .5);
}

/* ---- trellis ---- */
typedef struct {
  int64_t cost;
  int num_sig, rem_reg, hist, rice_par, rice_zero;
  int32_t sbb[2], sig[2], coef[6];
  uint8_t lev[16]; uint16_t tmpl[16];
} dq_state;
typedef struct { int64_t cost; int lev, prev; } dq_dec;
typedef struct { int lev; int64_t dd; } dq_pq;

typedef struct {
  const dq_geo *g; dq_rate r; dq_quant q;
  dq_state st[12], start;             /* twelve slots; cur / prv / skp = base index of the four states in each role */
  int cur, prv, skp;
  uint8_t hlev[2][4][1024], hflag[2][4][64]; int hcur;      /* CommonCtx: levels and group flags of the paths, two generations */
  int ch, reg_full;                    /* regular-bin budget of the block */
  dq_dec tr[1024][8];
} dq_ctx;

static int32_t rice_bits(int par, unsigned v)          /* g_goRiceBits 887-893 = length of the Golomb-Rice / escape code of v (EL/BinEncoder.cpp:444-472) in 2^-15 bits */
{
  const unsigned thr = 5u << par;
  if (v < thr) return (int32_t) (((v >> par) + 1 + (unsigned) par) << 15);
  unsigned prefix = 0; const unsigned code = (v >> par) - 5;
  while (code > ((2u << prefix) - 2)) prefix++;
  return (int32_t) ((5 + prefix + prefix + (unsigned) par + 1) << 15);
}
static inline int64_t lev_bits(const dq_state *s, int lev)
{
  if (lev < 4) return s->coef[lev];
  const unsigned v = (unsigned) (lev - 4) >> 1;
  return (int64_t) s->coef[lev - (int) (v << 1)] + rice_bits(s->rice_par, v < 32 ? v : 31);
}
static void state_init(dq_state *s, const dq_rate *r, int k)      /* State::init 906-916 */
{
  s->cost = INT64_MAX >> 1; s->num_sig = 0; s->rem_reg = 4; s->hist = -1;
  const int set = imax(k - 1, 0);
  s->sig[0] = r->sig[set][0][0]; s->sig[1] = r->sig[set][0][1];
  for (int i = 0; i < 6; i++) s->coef[i] = r->gtx[0][i];
  s->rice_par = 0; s->rice_zero = 0;
}
/* checkRdCosts 918-1030; spt: 0 inside a group, 1 first coded position of a group (SCAN_SOCSBB), 2 last one (SCAN_EOCSBB) */
static void check_costs(const dq_state *s, int k, int spt, const dq_pq *A, const dq_pq *B, dq_dec *dA, dq_dec *dB)
{
  int64_t cA = s->cost + A->dd, cB = s->cost + B->dd, cZ = s->cost;
  if (s->rem_reg >= 4) {
    cA += lev_bits(s, A->lev); cB += lev_bits(s, B->lev);
    if (spt == 0) { cA += s->sig[1]; cB += s->sig[1]; cZ += s->sig[0]; }
    else if (spt == 1) { cA += s->sbb[1] + s->sig[1]; cB += s->sbb[1] + s->sig[1]; cZ += s->sbb[1] + s->sig[0]; }
    else if (s->num_sig) { cA += s->sig[1]; cB += s->sig[1]; cZ += s->sig[0]; }
    else cZ = dA->cost;
  } else {
    cA += (1 << 15) + rice_bits(s->rice_par, (unsigned) (A->lev <= s->rice_zero ? A->lev - 1 : (A->lev < 32 ? A->lev : 31)));
    cB += (1 << 15) + rice_bits(s->rice_par, (unsigned) (B->lev <= s->rice_zero ? B->lev - 1 : (B->lev < 32 ? B->lev : 31)));
    cZ += rice_bits(s->rice_par, (unsigned) s->rice_zero);
  }
  if (cA < dA->cost) { dA->cost = cA; dA->lev = A->lev; dA->prev = k; }
  if (cZ < dA->cost) { dA->cost = cZ; dA->lev = 0; dA->prev = k; }
  if (cB < dB->cost) { dB->cost = cB; dB->lev = B->lev; dB->prev = k; }
}
static void check_start(const dq_state *s, int32_t lastOffset, const dq_pq *P, dq_dec *d)        /* 1032-1050 */
{
  const int64_t c = P->dd + lastOffset + lev_bits(s, P->lev);
  if (c < d->cost) { d->cost = c; d->lev = P->lev; d->prev = -1; }
}
static inline int sig_off_next(int ch, int diag) { return ch ? (diag < 2 ? 4 : 0) : (diag < 2 ? 8 : diag < 5 ? 4 : 0); }           /* 402-419 */
static inline int gtx_off_next(int ch, int diag) { return ch ? (diag < 1 ? 6 : 1) : (diag < 1 ? 16 : diag < 3 ? 11 : diag < 10 ? 6 : 1); }
static void set_ctx_bits(dq_ctx *D, dq_state *s, int k, int next, int sumAbs1, int sumNum)
{
  const int diag = D->g->px[next] + D->g->py[next], set = imax(k - 1, 0);
  const int si = sig_off_next(D->ch, diag) + imin((sumAbs1 + 1) >> 1, 3), gi = gtx_off_next(D->ch, diag) + imin(sumAbs1 - sumNum, 4);
  s->sig[0] = D->r.sig[set][si][0]; s->sig[1] = D->r.sig[set][si][1];
  for (int i = 0; i < 6; i++) s->coef[i] = D->r.gtx[gi][i];
}
/* State::updateState 1109-1273 */
static void update_state(dq_ctx *D, int k, int scanIdx, const dq_dec *d)
{
  dq_state *s = &D->st[D->cur + k];
  const dq_geo *g = D->g;
  s->cost = d->cost;
  if (d->prev <= -2) return;
  if (d->prev >= 0) {
    const dq_state *p = &D->st[D->prv + d->prev];
    s->num_sig = p->num_sig + !!d->lev; s->hist = p->hist; s->sbb[0] = p->sbb[0]; s->sbb[1] = p->sbb[1];
    s->rem_reg = p->rem_reg - 1; s->rice_par = p->rice_par;
    if (s->rem_reg >= 4) s->rem_reg -= d->lev < 2 ? d->lev : 3;
    memcpy(s->lev, p->lev, sizeof s->lev); memcpy(s->tmpl, p->tmpl, sizeof s->tmpl);
  } else {
    s->num_sig = 1; s->hist = -1;
    s->rem_reg = D->reg_full - (d->lev < 2 ? d->lev : 3);
    memset(s->lev, 0, sizeof s->lev); memset(s->tmpl, 0, sizeof s->tmpl);
  }
  s->lev[scanIdx & (g->gs - 1)] = (uint8_t) imin(255, d->lev);
  const int next = scanIdx - 1, nin = next & (g->gs - 1), t = s->tmpl[nin];
  int sumAbs = t >> 8;
  for (int i = 0; i < g->in_num[next]; i++) sumAbs += s->lev[g->in_pos[next][i]];
  if (s->rem_reg >= 4) {
    int sumAbs1 = (t >> 3) & 31, sumNum = t & 7;
    for (int i = 0; i < g->in_num[next]; i++) { const int a = s->lev[g->in_pos[next][i]]; sumAbs1 += imin(4 + (a & 1), a); sumNum += !!a; }
    set_ctx_bits(D, s, k, next, sumAbs1, sumNum);
    s->rice_par = ORC_GORICE_PARS[imax(imin(31, sumAbs - 4 * 5), 0)];
  } else {
    sumAbs = imin(31, sumAbs);
    s->rice_par = ORC_GORICE_PARS[sumAbs];
    s->rice_zero = ORC_GORICE_POS0[imax(0, k - 1) * 32 + sumAbs];
  }
}
/* State::updateStateEOS 1275-1315 + CommonCtx::update 1317-1398: the last coded position of a coefficient group */
static void update_state_eos(dq_ctx *D, int k, int scanIdx, const dq_dec *d)
{
  dq_state *s = &D->st[D->cur + k];
  const dq_geo *g = D->g;
  s->cost = d->cost;
  if (d->prev <= -2) return;
  const dq_state *p = 0;
  if (d->prev >= 4) { p = &D->st[D->skp + d->prev - 4]; s->num_sig = 0; memset(s->lev, 0, 16); }
  else if (d->prev >= 0) { p = &D->st[D->prv + d->prev]; s->num_sig = p->num_sig + !!d->lev; memcpy(s->lev, p->lev, 16); }
  else { s->num_sig = 1; memset(s->lev, 0, 16); }
  s->lev[scanIdx & (g->gs - 1)] = (uint8_t) imin(255, d->lev);
  /* the path's history: group flags and the levels later positions still look at */
  uint8_t *flags = D->hflag[D->hcur][k], *levels = D->hlev[D->hcur][k];
  const int cp = g->max_dist[scanIdx - 1];
  if (p && p->hist >= 0) { memcpy(flags, D->hflag[D->hcur ^ 1][p->hist], (size_t) g->nsbb); memcpy(levels + scanIdx, D->hlev[D->hcur ^ 1][p->hist] + scanIdx, (size_t) cp); }
  else { memset(flags, 0, (size_t) g->nsbb); memset(levels + scanIdx, 0, (size_t) cp); }
  flags[g->sbb_raster[scanIdx >> g->lcg]] = (uint8_t) !!s->num_sig;
  memcpy(levels + scanIdx, s->lev, (size_t) g->gs);
  const int nsp = g->sbb_raster[(scanIdx - 1) >> g->lcg], nsy = nsp / g->wsbb, nsx = nsp - nsy * g->wsbb;
  const int right = nsx < g->wsbb - 1 ? nsp + 1 : 0, below = nsy < g->hsbb - 1 ? nsp + g->wsbb : 0;
  const int sigN = ((right ? flags[right] : 0) || (below ? flags[below] : 0)) ? 1 : 0;
  s->num_sig = 0;
  s->rem_reg = p ? p->rem_reg : D->reg_full;
  s->rice_par = 0; s->hist = k;
  s->sbb[0] = D->r.sbb[sigN][0]; s->sbb[1] = D->r.sbb[sigN][1];
  const int beg = scanIdx - g->gs;
  memset(s->lev, 0, 16);
  for (int id = 0; id < g->gs; id++) {
    int sumAbs = 0, sumAbs1 = 0, sumNum = 0;
    for (int j = 0; j < g->out_num[beg + id]; j++) { const int a = levels[beg + g->out_pos[beg + id][j]]; sumAbs += a; sumAbs1 += imin(4 + (a & 1), a); sumNum += !!a; }
    s->tmpl[id] = g->out_num[beg + id] ? (uint16_t) (sumNum + (sumAbs1 << 3) + (imin(127, sumAbs) << 8)) : 0;
  }
  const int next = scanIdx - 1, t = s->tmpl[next & (g->gs - 1)];
  set_ctx_bits(D, s, k, next, (t >> 3) & 31, t & 7);
}

/* DQIntern::DepQuant::quant 1592-1731.  coef: w*h transform coefficients (stride w); ctx: the estimator's contexts at the time of the call;
 * comp 0 Y / 1 Cb / 2 Cr; cbf_ctx: context index of the block's cbf (-1: inferred); qp: what QpParam hands over (with QpBDOffset);
 * zo: explicit MTS (32-point transforms keep 16 coefficients); lfnst: cu.lfnstIdx.  Returns absSum. */
int orc_depquant(const uint16_t *s0, const uint16_t *s1, const int *coef, int w, int h, int comp, int cbf_ctx, int bit_depth, int qp, double lambda,
                 int zo, int lfnst, int16_t *level)
{
  static dq_ctx Dst;                  /* oracle is single-threaded */
  dq_ctx *D = &Dst;
  const dq_geo *g = geo_of(w, h);
  memset(level, 0, (size_t) w * h * 2);
  { int64_t c[9]; orc_depquant_consts(w, h, bit_depth, qp, lambda, c);
    D->q.qshift = (int) c[0]; D->q.qadd = c[1]; D->q.qscale = c[2]; D->q.max_qidx = (int) c[3]; D->q.thres_last = (int) c[4]; D->q.dshift = (int) c[5]; D->q.dadd = c[6]; D->q.dstep = c[7]; D->q.dorg = c[8]; }
  const dq_quant *Q = &D->q;
  int effW = w, effH = h, zeroOut = 0;
  if (zo && comp == 0) { effH = h == 32 ? 16 : h; effW = w == 32 ? 16 : w; zeroOut = effH < h || effW < w; }
  /* first tested position 1630-1660 */
  int first = g->total - 1;
  if (lfnst > 0 && w >= 4 && h >= 4) first = ((w == 4 && h == 4) || (w == 8 && h == 8)) ? 7 : 15;
  const int thr = Q->thres_last / (int) (4 * Q->qscale);
  for (; first >= 0; first--) {
    if (zeroOut && (g->px[first] >= (w == 32 ? 16 : 32) || g->py[first] >= (h == 32 ? 16 : 32))) continue;
    if (abs(coef[g->scan[first]]) > thr) break;
  }
  if (first < 0) return 0;

  orc_cabac cb; memcpy(cb.s0, s0, sizeof cb.s0); memcpy(cb.s1, s1, sizeof cb.s1);
  D->g = g; D->ch = comp ? 1 : 0;
  rate_init(&D->r, &cb, w, h, comp, cbf_ctx);
  D->hcur = 0;
  for (int k = 0; k < 12; k++) { state_init(&D->st[k], &D->r, k & 3); D->st[k].sbb[0] = D->st[k].sbb[1] = 0; memset(D->st[k].lev, 0, 16); memset(D->st[k].tmpl, 0, 32); }
  state_init(&D->start, &D->r, 0);
  D->cur = 0; D->prv = 4; D->skp = 8;
  D->reg_full = (imin(32, effW) * imin(32, effH) * 28) / 16;

  for (int scanIdx = first; scanIdx >= 0; scanIdx--) {
    const int inside = scanIdx & (g->gs - 1), eos = inside == 0;
    int spt = 0;
    if (inside == g->gs - 1 && scanIdx > g->gs && scanIdx < g->total - 1) spt = 1;
    else if (eos && scanIdx > 0 && scanIdx < g->total - g->gs) spt = 2;
    const int zeroed = zeroOut && (g->px[scanIdx] >= effW || g->py[scanIdx] >= effH);
    dq_dec *dec = D->tr[scanIdx];
    { const int t = D->prv; D->prv = D->cur; D->cur = t; }
    for (int k = 0; k < 4; k++) { dec[k].cost = INT64_MAX >> 2; dec[k].lev = -1; dec[k].prev = -2; dec[4 + k].cost = INT64_MAX >> 2; dec[4 + k].lev = 0; dec[4 + k].prev = 4 + k; }
    if (zeroed) {
      if (spt == 2) for (int k = 0; k < 4; k++) { dec[k].cost = D->st[D->skp + k].cost + D->st[D->skp + k].sbb[0]; dec[k].lev = 0; dec[k].prev = 4 + k; }
    } else {
      /* preQuantCoeff 812-832 */
      dq_pq pq[4];
      const int64_t scaledOrg = (int64_t) abs(coef[g->scan[scanIdx]]) * Q->qscale;
      int qIdx = imax(1, imin(Q->max_qidx, (int) ((scaledOrg + Q->qadd) >> Q->qshift)));
      int64_t scaledAdd = qIdx * Q->dstep - scaledOrg * Q->dorg;
      for (int i = 0; i < 4; i++) { dq_pq *p = &pq[qIdx & 3]; p->dd = (scaledAdd * qIdx + Q->dadd) >> Q->dshift; p->lev = (++qIdx) >> 1; scaledAdd += Q->dstep; }
      const dq_state *P = &D->st[D->prv];
      check_costs(&P[0], 0, spt, &pq[0], &pq[2], &dec[0], &dec[2]);
      check_costs(&P[1], 1, spt, &pq[0], &pq[2], &dec[2], &dec[0]);
      check_costs(&P[2], 2, spt, &pq[3], &pq[1], &dec[1], &dec[3]);
      check_costs(&P[3], 3, spt, &pq[3], &pq[1], &dec[3], &dec[1]);
      if (spt == 2) for (int k = 0; k < 4; k++) {         /* checkRdCostSkipSbb 1052-1061 */
        const int64_t c = D->st[D->skp + k].cost + D->st[D->skp + k].sbb[0];
        if (c < dec[k].cost) { dec[k].cost = c; dec[k].lev = 0; dec[k].prev = 4 + k; }
      }
      const int32_t lastOffset = D->r.last_x[g->px[scanIdx]] + D->r.last_y[g->py[scanIdx]];
      check_start(&D->start, lastOffset, &pq[0], &dec[0]);
      check_start(&D->start, lastOffset, &pq[2], &dec[2]);
    }
    if (scanIdx) {
      if (eos) {
        D->hcur ^= 1;
        for (int k = 0; k < 4; k++) update_state_eos(D, k, scanIdx, &dec[k]);
        memcpy(dec + 4, dec, 4 * sizeof(dq_dec));
      } else if (!zeroed) for (int k = 0; k < 4; k++) update_state(D, k, scanIdx, &dec[k]);
      if (spt == 1) { const int t = D->prv; D->prv = D->skp; D->skp = t; }
    }
  }
  /* best path and back-tracking 1709-1730 */
  int prev = -2; int64_t minCost = 0;
  for (int k = 0; k < 4; k++) if (D->tr[0][k].cost < minCost) { prev = k; minCost = D->tr[0][k].cost; }
  int absSum = 0;
  for (int scanIdx = 0; prev >= 0; scanIdx++) {
    const dq_dec d = D->tr[scanIdx][prev];
    const int blk = g->scan[scanIdx];
    level[blk] = (int16_t) (coef[blk] < 0 ? -d.lev : d.lev);
    absSum += d.lev;
    prev = d.prev;
  }
  return absSum;
}

/* Quantizer::dequantBlock 741-810 (flat scaling): the state machine over the coded levels picks the quantiser of each coefficient */
void orc_dequant_dq(const int16_t *level, int w, int h, int bit_depth, int qp, int *coef)
{
  const dq_geo *g = geo_of(w, h);
  memset(coef, 0, (size_t) w * h * sizeof(int));
  int last = -1;
  for (int s = g->total - 1; s >= 0; s--) if (level[g->scan[s]]) { last = s; break; }
  if (last < 0) return;
  const int lw = ilog2(w), lh = ilog2(h), sq = (lw + lh) & 1;
  const int qpDQ = qp + 1, per = qpDQ / 6, rem = qpDQ - 6 * per;
  const int trShift = 15 - bit_depth - ((lw + lh) >> 1) + (sq ? -1 : 0);
  const int shift = 6 + 1 - per - trShift;
  int invQ = ORC_INV_QUANT_SCALES[sq * 6 + rem];
  const int add = shift < 0 ? 0 : ((1 << shift) >> 1);
  int state = 0;
  for (int s = last; s >= 0; s--) {
    const int lv = level[g->scan[s]];
    if (lv) {
      if (shift < 0 && s == last) invQ <<= -shift;
      const int qIdx = (lv << 1) + (lv > 0 ? -(state >> 1) : (state >> 1));
      const int64_t v = ((int64_t) qIdx * (int64_t) invQ + add) >> (shift < 0 ? 0 : shift);
      coef[g->scan[s]] = (int) (v < -32768 ? -32768 : v > 32767 ? 32767 : v);
    }
    state = (32040 >> ((state << 2) + ((lv & 1) << 1))) & 3;
  }
}

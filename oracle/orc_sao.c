/*
 * orc_sao.c — CPU restatement of the sample adaptive offset FILTER with the per-CTU parameters given: what the decoder side of
 * SampleAdaptiveOffset does to a deblocked picture.  TEST INFRASTRUCTURE ONLY (see vvc_oracle.h).
 *
 * Follows CL/SampleAdaptiveOffset.cpp: SAOProcess 617-670 (parameters resolved for the whole picture, then every CTU filtered from a copy of the deblocked picture),
 * xReconstructBlkSAOParams 265-290 with getMergeList 172-227 (a merge candidate is the CTU to the left / above in the same tile) and reconstructBlkSAOParam 229-263,
 * invertQuantOffsets 147-170 (offsets times 2^log2OffsetScale; band offsets belong to four consecutive bands from the band position, modulo 32; edge offsets to the
 * classes full valley, half valley, half peak, full peak), offsetCTU 548-615 and offsetBlock 292-546 with deriveLoopFilterBoundaryAvailibility 818-883 (one slice;
 * neighbour CTUs of another tile only when the loop filters may cross tile borders).
 *
 * The reference walks a CTU row by row with sign line buffers; what it computes per sample is: the two neighbours along the class direction (0 degrees: left / right,
 * 90: above / below, 135: above-left / below-right, 45: above-right / below-left), both of which must be available - inside the picture and, unless filtering across
 * tiles is allowed, inside the tile of the sample's CTU - else the sample stays; edge index = sign(c - a) + sign(c - b) in -2..2 picks the offset (index 0: none).
 * Pinned against the reference's SampleAdaptiveOffset (oracle/_ref, tests/golden/sao.npz).
 */
#include <stdlib.h>
#include <string.h>
#include "orc_internal.h"

static int sgn_(int v) { return (v > 0) - (v < 0); }
static int tile_of(int ctu, int n_ctus, int n_tiles) { int t = 0; for (int i = 0; i < n_tiles; i++) if (ctu >= (i * n_ctus) / n_tiles) t = i; return t; }

/* prm: per CTU (raster) and component {mode 0 off / 1 new / 2 merge, type (new: 0..3 edge class 0 / 90 / 135 / 45 degrees, 4 band; merge: 0 left, 1 above),
 * band position, four coded offsets}.  planes: int16, stride = plane width, filtered in place.  Returns 0, or -1 for a merge without a candidate / bad values. */
int orc_sao_picture(int w, int h, int bit_depth, int tile_cols, int tile_rows, int lf_across_tiles, int log2_offset_scale, const orc_sao_param *prm, int16_t *y, int16_t *cb, int16_t *cr)
{
  const int cw = (w + 127) >> 7, ch = (h + 127) >> 7, nctu = cw * ch;
  int16_t *planes[3] = { y, cb, cr };
  /* resolve merges in raster order: table[ctu][comp] = {type or -1, 32 offsets by class / band} */
  int8_t *type = (int8_t *) malloc((size_t) nctu * 3);
  int16_t *off = (int16_t *) calloc((size_t) nctu * 3 * 32, sizeof(int16_t));
  int rc = 0;
  for (int a = 0; a < nctu && !rc; a++) {
    const int cx = a % cw, cy = a / cw, tx = tile_of(cx, cw, tile_cols), ty = tile_of(cy, ch, tile_rows);
    for (int c = 0; c < 3; c++) {
      const orc_sao_param *p = &prm[a * 3 + c];
      int8_t *t = &type[a * 3 + c]; int16_t *o = &off[(size_t) (a * 3 + c) * 32];
      *t = -1;
      if (p->mode == 0) continue;
      if (p->mode == 1) {
        if (p->type < 0 || p->type > 4 || p->band < 0 || p->band > 31) { rc = -1; break; }
        *t = p->type;
        if (p->type == 4) for (int i = 0; i < 4; i++) o[(p->band + i) & 31] = (int16_t) (p->off[i] * (1 << log2_offset_scale));
        else { o[0] = (int16_t) (p->off[0] * (1 << log2_offset_scale)); o[1] = (int16_t) (p->off[1] * (1 << log2_offset_scale)); o[3] = (int16_t) (p->off[2] * (1 << log2_offset_scale)); o[4] = (int16_t) (p->off[3] * (1 << log2_offset_scale)); }
      } else {
        const int left = p->type == 0, sx = left ? cx - 1 : cx, sy = left ? cy : cy - 1;
        if (sx < 0 || sy < 0 || tile_of(sx, cw, tile_cols) != tx || tile_of(sy, ch, tile_rows) != ty) { rc = -1; break; }      /* CHECK(mergeTarget == NULL) */
        const int s = sy * cw + sx;
        *t = type[s * 3 + c]; memcpy(o, &off[(size_t) (s * 3 + c) * 32], 32 * sizeof(int16_t));
      }
    }
  }
  for (int c = 0; c < 3 && !rc; c++) {
    const int sh = c ? 1 : 0, pw = w >> sh, ph = h >> sh, mx = (1 << bit_depth) - 1, cs = 128 >> sh;
    int16_t *src = (int16_t *) malloc((size_t) pw * ph * sizeof(int16_t));
    memcpy(src, planes[c], (size_t) pw * ph * sizeof(int16_t));
    for (int yy = 0; yy < ph; yy++) for (int xx = 0; xx < pw; xx++) {
      const int cx = xx / cs, cy = yy / cs, a = cy * cw + cx, t = type[a * 3 + c];
      if (t < 0) continue;
      const int16_t *o = &off[(size_t) (a * 3 + c) * 32];
      const int v = src[yy * pw + xx];
      int r;
      if (t == 4) r = v + o[v >> (bit_depth - 5)];
      else {
        const int dx = t == 1 ? 0 : 1, dy = t == 0 ? 0 : 1;
        const int ax = t == 3 ? xx + 1 : xx - dx, ay = yy - dy, bx = t == 3 ? xx - 1 : xx + dx, by = yy + dy;      /* 45 degrees: above-right and below-left */
        int ok = ax >= 0 && ax < pw && ay >= 0 && ay < ph && bx >= 0 && bx < pw && by >= 0 && by < ph;
        if (ok && !lf_across_tiles) {
          const int tx = tile_of(cx, cw, tile_cols), ty = tile_of(cy, ch, tile_rows);
          ok = tile_of(ax / cs, cw, tile_cols) == tx && tile_of(ay / cs, ch, tile_rows) == ty && tile_of(bx / cs, cw, tile_cols) == tx && tile_of(by / cs, ch, tile_rows) == ty;
        }
        if (!ok) continue;
        r = v + o[2 + sgn_(v - src[ay * pw + ax]) + sgn_(v - src[by * pw + bx])];
      }
      planes[c][yy * pw + xx] = (int16_t) (r < 0 ? 0 : r > mx ? mx : r);
    }
    free(src);
  }
  free(type); free(off);
  return rc;
}

/*
 * The encoder's SAO statistics (EL/EncSampleAdaptiveOffset.cpp getStatistics 284-353 / getBlkStats 1135-1549 with SAOLcuBoundary 0, i.e. isCalculatePreDeblockSamples
 * false and the skip lines of createEncData 131-134): per CTU, component and type the number of samples of every class and the sum of (original - deblocked) over them.
 * PARITY UNPINNED for the region rules: EncSampleAdaptiveOffset is an EncoderLib unit behind CABACWriter.h (which includes OpenCV in this fork) and does not compile
 * here, so which samples are counted follows the reference by reading only; the class of a sample is the pinned filter's (tests tie the two: with offsets o the sums predict
 * the SSE change of that filter on the counted samples).
 *
 * The reference walks a CTU with sign line buffers; per sample that is: EO class = 2 + sign(c - a) + sign(c - b) over the same two neighbours as the filter (classes
 * 0..4, 2 = plain), band = c >> (bitDepth - 5).  Counted are the samples of the CTU except: the last 5 luma / 3 chroma columns when a CTU follows to the right and the last 4 / 2
 * rows when one follows below (their deblocking is not final when a pipelined encoder decides the CTU); for the edge types also the column / row whose neighbour lies outside
 * the picture, or left of / above the CTU in another tile when the filters do not cross tile borders (deriveLoopFilterBoundaryAvailibility 1551-1576; right and below only
 * the picture border counts, 309-313), with the reference's treatment of the first row of the diagonal types (1318-1321, 1426-1431).
 * out: [ctu][component][type 0..4][0: count, 1: diff][32] int64 (edge types use entries 0..4).
 */
int orc_sao_statistics(int w, int h, int bit_depth, int tile_cols, int tile_rows, int lf_across_tiles, const int16_t *const org[3], const int16_t *const rec[3], int64_t *out)
{
  const int cw = (w + 127) >> 7, chh = (h + 127) >> 7, nctu = cw * chh;
  memset(out, 0, (size_t) nctu * 3 * 5 * 64 * sizeof(int64_t));
  for (int a = 0; a < nctu; a++) {
    const int cx = a % cw, cy = a / cw;
    const int tx = tile_of(cx, cw, tile_cols), ty = tile_of(cy, chh, tile_rows);
    const int left = cx > 0 && (lf_across_tiles || tile_of(cx - 1, cw, tile_cols) == tx), above = cy > 0 && (lf_across_tiles || tile_of(cy - 1, chh, tile_rows) == ty);
    const int aboveLeft = cx > 0 && cy > 0 && (lf_across_tiles || (tile_of(cx - 1, cw, tile_cols) == tx && tile_of(cy - 1, chh, tile_rows) == ty));
    const int right = cx + 1 < cw, below = cy + 1 < chh;
    for (int c = 0; c < 3; c++) {
      const int sh = c ? 1 : 0, pw = w >> sh, ph = h >> sh, cs = 128 >> sh, x0 = cx * cs, y0 = cy * cs;
      const int cwid = (x0 + cs > pw ? pw - x0 : cs), chei = (y0 + cs > ph ? ph - y0 : cs), skipR = c ? 3 : 5, skipB = c ? 2 : 4;
      const int16_t *s = rec[c], *o = org[c];
      for (int t = 0; t < 5; t++) {
        int64_t *cnt = out + ((size_t) (a * 3 + c) * 5 + t) * 64, *dif = cnt + 32;
        const int endXr = right ? cwid - skipR : cwid;          /* right end when no right neighbour is needed */
        const int endXn = right ? cwid - skipR : cwid - 1;      /* ... when it is */
        const int startXn = left ? 0 : 1;
        for (int y = 0; y < chei; y++) for (int x = 0; x < cwid; x++) {
          int in;
          switch (t) {
            case 0: in = y < (below ? chei - skipB : chei) && x >= startXn && x < endXn; break;
            case 1: in = x < endXr && y >= (above ? 0 : 1) && y < (below ? chei - skipB : chei - 1); break;
            case 2: in = y == 0 ? (x >= (aboveLeft ? 0 : 1) && x < (above ? endXn : 1)) : (x >= startXn && x < endXn && y < (below ? chei - skipB : chei - 1)); break;
            case 3: in = x >= startXn && x < endXn && (y == 0 ? above : y < (below ? chei - skipB : chei - 1)); break;
            default: in = x < endXr && y < (below ? chei - skipB : chei); break;
          }
          if (!in) continue;
          const int X = x0 + x, Y = y0 + y, v = s[(size_t) Y * pw + X];
          int k;
          if (t == 4) k = v >> (bit_depth - 5);
          else {
            const int dx = t == 1 ? 0 : 1, dy = t == 0 ? 0 : 1;
            const int ax = t == 3 ? X + 1 : X - dx, ay = Y - dy, bx = t == 3 ? X - 1 : X + dx, by = Y + dy;
            if (ax < 0 || ax >= pw || ay < 0 || ay >= ph || bx < 0 || bx >= pw || by < 0 || by >= ph) return -1;      /* the region rules keep every neighbour inside the picture */
            k = 2 + sgn_(v - s[(size_t) ay * pw + ax]) + sgn_(v - s[(size_t) by * pw + bx]);
          }
          cnt[k]++; dif[k] += o[(size_t) Y * pw + X] - v;
        }
      }
    }
  }
  return 0;
}

/*
 * The RD half of the SAO parameter decision from the statistics (EL/EncSampleAdaptiveOffset.cpp: decideBlkParams 793-1098 without SAOGreedyEnc, deriveModeNewRDO 597-735,
 * deriveModeMergeRDO 737-791, deriveOffsets 481-595 with estIterOffset 449-479, getDistortion / estSaoDist 406-447; rates as CABACWriter::sao_block_pars /
 * sao_offset_pars code the syntax, EL/CABACWriter.cpp:354-462, against the two SAO context models carried from CTU to CTU; merge candidates as getMergeList,
 * CL/SampleAdaptiveOffset.cpp:172-227).  Every slice component enabled (decidePicParams 355-404 switches components off only above temporal layer 0).
 * PARITY UNPINNED: the unit does not compile here (see above); the only tie to pinned code is through the statistics and the filter.
 * out: orc_sao_param per CTU and component, the form orc_sao_picture takes.
 */
typedef struct { int mode, type, band; int off[32]; } sao_prm;        /* mode 0 off / 1 new / 2 merge; off: by class, coded (quantised) or reconstructed */
typedef struct { uint16_t s0[2], s1[2]; uint64_t bits; } sao_cab;     /* [0] SaoMergeFlag, [1] SaoTypeIdx */
#define SAO_CTX_REF 287                                               /* Ctx::SaoMergeFlag in the reference's flat order; SaoTypeIdx follows */
static void sc_bin(sao_cab *c, int which, unsigned bin)
{
  const unsigned st = (unsigned) (c->s0[which] + c->s1[which]) >> 8;
  c->bits += ORC_BIN_FRAC_BITS[st * 2 + bin];
  const int rate = ORC_CTX_RATE[SAO_CTX_REF + which], r0 = 2 + ((rate >> 2) & 3), r1 = 3 + r0 + (rate & 3);
  c->s0[which] -= (c->s0[which] >> r0) & 0x7FE0; c->s1[which] -= (c->s1[which] >> r1) & 0x7FFE;
  if (bin) { c->s0[which] += (0x7fffu >> r0) & 0x7FE0; c->s1[which] += (0x7fffu >> r1) & 0x7FFE; }
}
static void sc_ep(sao_cab *c, int n) { c->bits += (uint64_t) n << 15; }
static int sao_max_q(int bd) { return (1 << ((bd < 10 ? bd : 10) - 5)) - 1; }
/* sao_offset_pars: comp 0 and 1 are the first components of their channel type */
static void sc_offset_pars(sao_cab *c, const sao_prm *p, int comp, int bd)
{
  const int first = comp < 2;
  if (first) {
    if (p->mode == 0) sc_bin(c, 1, 0);
    else { sc_bin(c, 1, 1); sc_ep(c, 1); }
  }
  if (p->mode == 1) {
    const int mx = sao_max_q(bd);
    int o[4], k = 0;
    for (int i = 0; i < (p->type == 4 ? 4 : 5); i++) { if (p->type != 4 && i == 2) continue; o[k++] = p->off[p->type == 4 ? (p->band + i) & 31 : i]; }
    for (int i = 0; i < 4; i++) { const int a = o[i] < 0 ? -o[i] : o[i]; if (mx) sc_ep(c, a < mx ? a + 1 : mx); }
    if (p->type == 4) { for (int i = 0; i < 4; i++) if (o[i]) sc_ep(c, 1); sc_ep(c, 5); }
    else if (first) sc_ep(c, 2);
  }
}
static void sc_block_pars(sao_cab *c, const sao_prm p[3], int bd, int leftAvail, int aboveAvail, int onlyMerge)
{
  int isLeft = 0, isAbove = 0;
  if (leftAvail) { isLeft = p[0].mode == 2 && p[0].type == 0; sc_bin(c, 0, (unsigned) isLeft); }
  if (aboveAvail && !isLeft) { isAbove = p[0].mode == 2 && p[0].type == 1; sc_bin(c, 0, (unsigned) isAbove); }
  if (onlyMerge) return;
  if (!isLeft && !isAbove) for (int k = 0; k < 3; k++) sc_offset_pars(c, &p[k], k, bd);
}
static int64_t sao_est_dist(int64_t count, int64_t off, int64_t diff) { return count * off * off - diff * off * 2; }
static int64_t sao_dist(int type, int band, const int *inv, const int64_t *cnt, const int64_t *dif)
{
  int64_t d = 0;
  if (type == 4) for (int i = band; i < band + 4; i++) d += sao_est_dist(cnt[i & 31], inv[i & 31], dif[i & 31]);
  else for (int i = 0; i < 5; i++) d += sao_est_dist(cnt[i], inv[i], dif[i]);
  return d;
}
static int sao_iter_offset(int type, double lambda, int in, int64_t count, int64_t diff, int step, int64_t *bestDist, double *bestCost, int th)
{
  int out = 0; double minCost = lambda;
  for (int it = in; it != 0; it = it > 0 ? it - 1 : it + 1) {
    const int a = it < 0 ? -it : it;
    int64_t rate = type == 4 ? a + 2 : a + 1;
    if (a == th) rate--;
    const int64_t dist = sao_est_dist(count, (int64_t) it << step, diff);
    const double cost = (double) dist + lambda * (double) rate;
    if (cost < minCost) { minCost = cost; out = it; *bestDist = dist; *bestCost = cost; }
  }
  return out;
}
static void sao_derive_offsets(int bd, int type, double lambda, int step, const int64_t *cnt, const int64_t *dif, int *q, int *aux)
{
  const int th = sao_max_q(bd), n = type == 4 ? 32 : 5;
  memset(q, 0, 32 * sizeof(int));
  for (int k = 0; k < n; k++) {
    if ((type != 4 && k == 2) || cnt[k] == 0) continue;
    const double x = (double) dif[k] / (double) (cnt[k] << step);
    int v = x >= 0 ? (int) (x + 0.5) : (int) (x - 0.5);
    q[k] = v < -th ? -th : v > th ? th : v;
  }
  if (type != 4) {
    for (int k = 0; k < 5; k++) {
      if ((k < 2 && q[k] < 0) || (k > 2 && q[k] > 0)) q[k] = 0;      /* valleys take positive offsets, peaks negative ones */
      if (q[k]) { int64_t d; double c; q[k] = sao_iter_offset(type, lambda, q[k], cnt[k], dif[k], step, &d, &c, th); }
    }
    *aux = 0;
  } else {
    double cost[32]; int64_t d;
    for (int k = 0; k < 32; k++) { cost[k] = lambda; if (q[k]) q[k] = sao_iter_offset(type, lambda, q[k], cnt[k], dif[k], step, &d, &cost[k], th); }
    double minCost = 1.7976931348623157e308; *aux = 0;
    for (int b = 0; b < 29; b++) { const double c = cost[b] + cost[b + 1] + cost[b + 2] + cost[b + 3]; if (c < minCost) { minCost = c; *aux = b; } }
    int keep[32]; memset(keep, 0, sizeof keep);
    for (int i = 0; i < 4; i++) keep[(*aux + i) & 31] = q[(*aux + i) & 31];
    memcpy(q, keep, sizeof keep);
  }
}
static void sao_invert(int type, int band, int step, int *dst, const int *src)
{
  memset(dst, 0, 32 * sizeof(int));
  if (type == 4) for (int i = 0; i < 4; i++) dst[(band + i) & 31] = src[(band + i) & 31] * (1 << step);
  else for (int k = 0; k < 5; k++) dst[k] = src[k] * (1 << step);
}
#define SAO_ST(st_, a_, c_, t_, w_) ((st_) + ((((size_t) (a_) * 3 + (c_)) * 5 + (t_)) * 2 + (w_)) * 32)
static void sao_mode_new(const int64_t *st, int a, int bd, const double *lambda, int step, sao_cab *cab, int leftAvail, int aboveAvail, sao_prm out[3], double *normCost)
{
  int64_t modeDist[3] = { 0, 0, 0 };
  sao_prm test[3]; int inv[32];
  const sao_cab ctxStartBlk = *cab;
  memset(out, 0, 3 * sizeof(sao_prm));
  sc_block_pars(cab, out, bd, leftAvail, aboveAvail, 1);          /* the merge flags of a CTU that does not merge */
  const sao_cab ctxStartLuma = *cab;
  sao_cab ctxBestLuma;
  {                                                           /* luma */
    cab->bits = 0; sc_offset_pars(cab, &out[0], 0, bd);
    double minCost = lambda[0] * ((double) cab->bits / 32768.0);
    ctxBestLuma = *cab;
    for (int t = 0; t < 5; t++) {
      memset(&test[0], 0, sizeof test[0]); test[0].mode = 1; test[0].type = t;
      sao_derive_offsets(bd, t, lambda[0], step, SAO_ST(st, a, 0, t, 0), SAO_ST(st, a, 0, t, 1), test[0].off, &test[0].band);
      sao_invert(t, test[0].band, step, inv, test[0].off);
      const int64_t dist = sao_dist(t, test[0].band, inv, SAO_ST(st, a, 0, t, 0), SAO_ST(st, a, 0, t, 1));
      *cab = ctxStartLuma; cab->bits = 0; sc_offset_pars(cab, &test[0], 0, bd);
      const double cost = (double) dist + lambda[0] * ((double) cab->bits / 32768.0);
      if (cost < minCost) { minCost = cost; modeDist[0] = dist; out[0] = test[0]; ctxBestLuma = *cab; }
    }
    *cab = ctxBestLuma;
  }
  {                                                           /* chroma: one type for both components */
    double cost = 0; uint64_t prev = 0;
    cab->bits = 0;
    for (int c = 1; c < 3; c++) { out[c].mode = 0; sc_offset_pars(cab, &out[c], c, bd); cost += lambda[c] * (1.0 / 32768.0) * (double) (cab->bits - prev); prev = cab->bits; }
    double minCost = cost;
    for (int t = 0; t < 5; t++) {
      int64_t dist[3] = { 0, 0, 0 };
      *cab = ctxBestLuma; cab->bits = 0; prev = 0; cost = 0;
      for (int c = 1; c < 3; c++) {
        memset(&test[c], 0, sizeof test[c]); test[c].mode = 1; test[c].type = t;
        sao_derive_offsets(bd, t, lambda[c], step, SAO_ST(st, a, c, t, 0), SAO_ST(st, a, c, t, 1), test[c].off, &test[c].band);
        sao_invert(t, test[c].band, step, inv, test[c].off);
        dist[c] = sao_dist(t, test[c].band, inv, SAO_ST(st, a, c, t, 0), SAO_ST(st, a, c, t, 1));
        sc_offset_pars(cab, &test[c], c, bd);
        cost += (double) dist[c] + lambda[c] * (1.0 / 32768.0) * (double) (cab->bits - prev); prev = cab->bits;
      }
      if (cost < minCost) { minCost = cost; for (int c = 1; c < 3; c++) { modeDist[c] = dist[c]; out[c] = test[c]; } }
    }
  }
  *normCost = 0;
  for (int c = 0; c < 3; c++) *normCost += (double) modeDist[c] / lambda[c];
  *cab = ctxStartBlk; cab->bits = 0;
  sc_block_pars(cab, out, bd, leftAvail, aboveAvail, 0);
  *normCost += (double) cab->bits / 32768.0;
}
static void sao_mode_merge(const int64_t *st, int a, int bd, const double *lambda, sao_cab *cab, const sao_prm *cand[2], sao_prm out[3], double *normCost)
{
  const sao_cab ctxStart = *cab; sao_cab ctxBest = *cab;
  *normCost = 1.7976931348623157e308;
  for (int m = 0; m < 2; m++) {
    if (!cand[m]) continue;
    sao_prm test[3]; double nd = 0;
    for (int c = 0; c < 3; c++) {
      test[c] = cand[m][c]; test[c].mode = 2; test[c].type = m;
      const sao_prm *mp = &cand[m][c];
      if (mp->mode != 0) nd += (double) sao_dist(mp->type, mp->band, mp->off, SAO_ST(st, a, c, mp->type, 0), SAO_ST(st, a, c, mp->type, 1)) / lambda[c];
    }
    *cab = ctxStart; cab->bits = 0;
    sc_block_pars(cab, test, bd, cand[0] != NULL, cand[1] != NULL, 0);
    const double cost = nd + (double) cab->bits / 32768.0;
    if (cost < *normCost) { *normCost = cost; memcpy(out, test, sizeof test); ctxBest = *cab; }
  }
  if (*normCost < 1.7976931348623157e308) *cab = ctxBest;
}
/* stats: [ctu][3][5][2][32] of one picture; lambda: per component; slice_qp: initialises the two context models (I slice); step = log2 of the offset scale */
int orc_sao_decide(int w, int h, int bit_depth, int tile_cols, int tile_rows, int slice_qp, const double *lambda, int log2_offset_scale, const int64_t *stats, orc_sao_param *prm)
{
  const int cw = (w + 127) >> 7, chh = (h + 127) >> 7, nctu = cw * chh;
  sao_prm *coded = (sao_prm *) calloc((size_t) nctu * 3, sizeof(sao_prm)), *recon = (sao_prm *) calloc((size_t) nctu * 3, sizeof(sao_prm));
  sao_cab cab; memset(&cab, 0, sizeof cab);
  { uint16_t s0[ORC_NUM_CTX], s1[ORC_NUM_CTX]; orc_ctx_init(slice_qp, s0, s1); for (int k = 0; k < 2; k++) { cab.s0[k] = s0[SAO_CTX_REF + k]; cab.s1[k] = s1[SAO_CTX_REF + k]; } }
  for (int a = 0; a < nctu; a++) {
    const int cx = a % cw, cy = a / cw, tx = tile_of(cx, cw, tile_cols), ty = tile_of(cy, chh, tile_rows);
    const sao_prm *cand[2] = { NULL, NULL };
    if (cx > 0 && tile_of(cx - 1, cw, tile_cols) == tx) cand[0] = &recon[(size_t) (a - 1) * 3];
    if (cy > 0 && tile_of(cy - 1, chh, tile_rows) == ty) cand[1] = &recon[(size_t) (a - cw) * 3];
    const sao_cab ctxStart = cab; sao_cab ctxBest = cab;
    double minCost = 1.7976931348623157e308, cost; sao_prm mode[3];
    sao_mode_new(stats, a, bit_depth, lambda, log2_offset_scale, &cab, cand[0] != NULL, cand[1] != NULL, mode, &cost);
    if (cost < minCost) { minCost = cost; memcpy(&coded[(size_t) a * 3], mode, sizeof mode); ctxBest = cab; }
    cab = ctxStart;
    sao_mode_merge(stats, a, bit_depth, lambda, &cab, cand, mode, &cost);
    if (cost < minCost) { minCost = cost; memcpy(&coded[(size_t) a * 3], mode, sizeof mode); ctxBest = cab; }
    cab = ctxBest;
    for (int c = 0; c < 3; c++) {                             /* reconstructBlkSAOParam: what later CTUs merge from */
      const sao_prm *cd = &coded[(size_t) a * 3 + c]; sao_prm *rc = &recon[(size_t) a * 3 + c];
      if (cd->mode == 2) *rc = cand[cd->type][c];
      else { *rc = *cd; if (cd->mode == 1) sao_invert(cd->type, cd->band, log2_offset_scale, rc->off, cd->off); }
    }
    for (int c = 0; c < 3; c++) {
      const sao_prm *cd = &coded[(size_t) a * 3 + c]; orc_sao_param *o = &prm[(size_t) a * 3 + c];
      memset(o, 0, sizeof *o);
      o->mode = (int8_t) cd->mode; o->type = (int8_t) cd->type;
      if (cd->mode == 1) {
        o->band = (int8_t) (cd->type == 4 ? cd->band : 0);
        for (int i = 0; i < 4; i++) o->off[i] = (int8_t) (cd->type == 4 ? cd->off[(cd->band + i) & 31] : cd->off[i < 2 ? i : i + 1]);
      }
    }
  }
  free(coded); free(recon);
  return 0;
}

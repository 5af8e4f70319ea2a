/*
 * orc_rate.c — rate estimation on the BitEstimator (TEST INFRASTRUCTURE ONLY, see vvc_oracle.h).
 * Restates EL/CABACWriter.cpp residual_coding (3773-3883), last_sig_coeff (4102-4160),
 * residual_coding_subblock (4164-4304), CL/ContextModelling.h CoeffCodingContext (50-200) and
 * CL/ContextModelling.cpp:40-134, EL/BinEncoder.cpp:444-485 encodeRemAbsEP.
 * Scope: regular (non transform-skip) residuals, sign hiding off; with cb->dq the dependent-quantisation state machine (stateTransTable
 * 32040, 3857-3858) selects the sig_coeff_flag context set and the bypass-mode zero position per coefficient, carried across sub-blocks.
 */
#include "orc_internal.h"
#include <stdlib.h>

static int ilog2(int v) { int r = 0; while (v > 1) { v >>= 1; r++; } return r; }
static int imin(int a, int b) { return a < b ? a : b; }
static int imax(int a, int b) { return a > b ? a : b; }

typedef struct {
  int w, h, ch;                 /* ch: 0 luma 1 chroma */
  int lcw, lch, lcg;            /* log2 CG width/height/size */
  int wg, hg;                   /* width/height in groups (zero-out region) */
  int nscan;                    /* real scan positions */
  uint16_t scan[1024];          /* raster idx per scan position */
  uint8_t  cgx[64], cgy[64];    /* CG scan */
  int last_off_x, last_off_y, last_sh_x, last_sh_y;
  int tmpl_diag, tmpl_sum1;
  uint8_t sig_group[64];        /* by CG raster position */
  int reg_bins;
} cctx_t;

static void cg_diag(int bw, int bh, uint8_t *xs, uint8_t *ys)
{
  int line = 0, col = 0;
  for (int n = 0; n < bw * bh; n++) {
    xs[n] = (uint8_t) col; ys[n] = (uint8_t) line;
    if (col == bw - 1 || line == 0) { line += col + 1; col = 0; if (line >= bh) { col += line - (bh - 1); line = bh - 1; } }
    else { col++; line--; }
  }
}

/* sigCtxIdAbs (CL/ContextModelling.h:107-156) */
static int sig_ctx(cctx_t *c, const int16_t *coeff, int blk, int state)
{
  const int W = c->w, H = c->h, posY = blk / W, posX = blk - posY * W;
  const int16_t *p = coeff + blk;
  const int diag = posX + posY;
  int numPos = 0, sumAbs = 0;
#define UPD(v) { int a = abs(v); sumAbs += imin(4 + (a & 1), a); numPos += !!a; }
  if (posX < W - 1) { UPD(p[1]); if (posX < W - 2) UPD(p[2]); if (posY < H - 1) UPD(p[W + 1]); }
  if (posY < H - 1) { UPD(p[W]); if (posY < H - 2) UPD(p[W << 1]); }
#undef UPD
  int ofs = imin((sumAbs + 1) >> 1, 3) + (diag < 2 ? 4 : 0);
  if (c->ch == 0) ofs += diag < 5 ? 4 : 0;
  c->tmpl_diag = diag; c->tmpl_sum1 = sumAbs - numPos;
  return ORC_CTX_SigFlag[c->ch + 2 * imax(0, state - 1)] + ofs;      /* m_sigFlagCtxSet[max(0,state-1)] */
}
/* ctxOffsetAbs (158-167) */
static int ctx_offset_abs(const cctx_t *c)
{
  int off = 0;
  if (c->tmpl_diag != -1) {
    off = imin(c->tmpl_sum1, 4) + 1;
    off += (!c->tmpl_diag ? (c->ch == 0 ? 15 : 5) : c->ch == 0 ? (c->tmpl_diag < 3 ? 10 : (c->tmpl_diag < 10 ? 5 : 0)) : 0);
  }
  return off;
}
/* templateAbsSum (172-199) */
static int tmpl_abs_sum(const cctx_t *c, const int16_t *coeff, int blk, int base)
{
  const int W = c->w, H = c->h, posY = blk / W, posX = blk - posY * W;
  const int16_t *p = coeff + blk;
  int sum = 0;
  if (posX < W - 1) { sum += abs(p[1]); if (posX < W - 2) sum += abs(p[2]); if (posY < H - 1) sum += abs(p[W + 1]); }
  if (posY < H - 1) { sum += abs(p[W]); if (posY < H - 2) sum += abs(p[W << 1]); }
  return imax(imin(sum - 5 * base, 31), 0);
}
/* EL/BinEncoder.cpp:444-472 (useLimitedPrefixLength forced true) */
void orc_enc_rem_abs(orc_cabac *cb, unsigned bins, unsigned rice)
{
  const unsigned thr = 5u << rice;
  if (bins < thr) {                                   /* unary prefix, then rice bits (EL/BinEncoder.cpp:222-229) */
    const unsigned length = (bins >> rice) + 1;
    orc_enc_bins_ep(cb, (1u << length) - 2, (int) length);
    orc_enc_bins_ep(cb, bins & ((1u << rice) - 1), (int) rice);
    return;
  }
  const unsigned maxPrefix = 32 - 5 - 15;
  unsigned prefix = 0, suffix, code = (bins >> rice) - 5;
  if (code >= ((1u << maxPrefix) - 1)) { prefix = maxPrefix; suffix = 15; }
  else { while (code > ((2u << prefix) - 2)) prefix++; suffix = prefix + rice + 1; }
  orc_enc_bins_ep(cb, (1u << (prefix + 5)) - 1, (int) (prefix + 5));                                           /* 248-252 */
  orc_enc_bins_ep(cb, ((code - ((1u << prefix) - 1)) << rice) | (bins & ((1u << rice) - 1)), (int) suffix);
}

void orc_residual_coding(orc_cabac *cb, const int16_t *coeff, int w, int h, int is_chroma) { orc_residual_coding_tu(cb, coeff, w, h, is_chroma, 0, 0, 0); }
/* mts_idx: -1 = no MTS syntax for this block (TU::isMTSAllowed false); otherwise tu.mtsIdx (0 DCT2, 2..5) */
void orc_residual_coding_mts(orc_cabac *cb, const int16_t *coeff, int w, int h, int is_chroma, int mts_idx) { orc_residual_coding_tu(cb, coeff, w, h, is_chroma, 0, mts_idx >= 0, mts_idx < 0 ? 0 : mts_idx); }
/* residual_coding of one block: mts_coding (3885-3941, JVET_O0294 context assignment) with ts_allowed = TU::isTSAllowed and mts_allowed =
 * TU::isMTSAllowed of the block, tu.mtsIdx = mts_idx (0 DCT2, 1 transform skip, 2..5 the explicit pairs), then the coefficients */
void orc_residual_coding_tu(orc_cabac *cb, const int16_t *coeff, int w, int h, int is_chroma, int ts_allowed, int mts_allowed, int mts_idx)
{
  if (ts_allowed) orc_enc_bin(cb, mts_idx == 1, ORC_CTX_MTSIndex + 6);
  if (mts_idx != 1 && mts_allowed) {
    orc_enc_bin(cb, mts_idx != 0, ORC_CTX_MTSIndex + 0);
    if (mts_idx) for (int i = 0; i < 3; i++) { const int sym = mts_idx > i + 2; orc_enc_bin(cb, (unsigned) sym, ORC_CTX_MTSIndex + 7 + i); if (!sym) break; }
  }
  if (!is_chroma && mts_idx == 1) { orc_residual_coding_ts(cb, coeff, w, h); return; }      /* 3802-3806 */
  const int zo = mts_idx > 1;         /* 32-point DST-VII / DCT-VIII keep 16 coefficients (getTbAreaAfterCoefZeroOut, CL/Unit.cpp:872-890) */
  cctx_t c; memset(&c, 0, sizeof c);
  c.w = w; c.h = h; c.ch = is_chroma;
  orc_cg_shape(w, h, &c.lcw, &c.lch); c.lcg = c.lcw + c.lch;
  c.wg = imin(32, w) >> c.lcw; c.hg = imin(32, h) >> c.lch;
  c.nscan = orc_scan_order(w, h, c.scan);
  cg_diag(c.wg, c.hg, c.cgx, c.cgy);
  c.tmpl_diag = -1; c.tmpl_sum1 = -1;
  const int l2w = ilog2(w), l2h = ilog2(h);
  if (is_chroma) {
    c.last_sh_x = imin(2, imax(0, w >> 3)); c.last_sh_y = imin(2, imax(0, h >> 3));
  } else {
    static const int prefix_ctx[8] = { 0, 0, 0, 3, 6, 10, 15, 21 };
    c.last_off_x = prefix_ctx[l2w]; c.last_off_y = prefix_ctx[l2h];
    c.last_sh_x = (l2w + 1) >> 2; c.last_sh_y = (l2h + 1) >> 2;
  }
  /* last position + CG significance (3823-3835) */
  int scanPosLast = -1;
  uint8_t sigGroupFlags[64] = { 0 };
  for (int sp = 0; sp < c.nscan; sp++) if (coeff[c.scan[sp]]) { scanPosLast = sp; sigGroupFlags[sp >> c.lcg] = 1; }
  if (scanPosLast < 0) return;   /* reference CHECKs; callers only come here with cbf = 1 */
  cb->last_scan_pos = scanPosLast;         /* read by the CU-level LFNST signalling (3837-3850) */

  /* last_sig_coeff (4102-4160) */
  {
    const int blk = c.scan[scanPosLast];
    int posY = blk / w, posX = blk - posY * w;
    const int gx = ORC_GROUP_IDX[posX], gy = ORC_GROUP_IDX[posY];
    /* 4115-4126: with an explicit MTS pair a 32-point side only codes positions below 16 */
    const int maxX = ORC_GROUP_IDX[(zo && w == 32 ? 16 : imin(32, w)) - 1], maxY = ORC_GROUP_IDX[(zo && h == 32 ? 16 : imin(32, h)) - 1];
    int k;
    for (k = 0; k < gx; k++) orc_enc_bin(cb, 1, ORC_CTX_LastX[c.ch] + c.last_off_x + (k >> c.last_sh_x));
    if (gx < maxX) orc_enc_bin(cb, 0, ORC_CTX_LastX[c.ch] + c.last_off_x + (k >> c.last_sh_x));
    for (k = 0; k < gy; k++) orc_enc_bin(cb, 1, ORC_CTX_LastY[c.ch] + c.last_off_y + (k >> c.last_sh_y));
    if (gy < maxY) orc_enc_bin(cb, 0, ORC_CTX_LastY[c.ch] + c.last_off_y + (k >> c.last_sh_y));
    if (gx > 3) orc_enc_bins_ep(cb, (uint32_t) (posX - ORC_MIN_IN_GROUP[gx]), (gx - 2) >> 1);
    if (gy > 3) orc_enc_bins_ep(cb, (uint32_t) (posY - ORC_MIN_IN_GROUP[gy]), (gy - 2) >> 1);
  }
  /* regular-bin budget (3859-3860): TbAreaAfterCoefZeroOut * 28 >> 4 */
  c.reg_bins = ((zo && w == 32 ? 16 : imin(32, w)) * (zo && h == 32 ? 16 : imin(32, h)) * 28) >> 4;

  const int cgSize = 1 << c.lcg;
  const int stateTab = cb->dq ? 32040 : 0; int state = 0;
  for (int sub = scanPosLast >> c.lcg; sub >= 0; sub--) {
    /* initSubblock (CL/ContextModelling.cpp:114-134) */
    const int cgPosX = c.cgx[sub], cgPosY = c.cgy[sub], cgPos = cgPosY * c.wg + cgPosX;
    const int minSub = sub << c.lcg, maxSub = minSub + cgSize - 1;
    if (sigGroupFlags[sub]) c.sig_group[cgPos] = 1;
    const int sigRight = (cgPosX + 1) < c.wg ? c.sig_group[cgPos + 1] : 0;
    const int sigLower = (cgPosY + 1) < c.hg ? c.sig_group[cgPos + c.wg] : 0;
    const int sigGroupCtx = ORC_CTX_SigCoeffGroup[c.ch] + (sigRight | sigLower);
    if (zo && ((h == 32 && cgPosY >= (16 >> c.lch)) || (w == 32 && cgPosX >= (16 >> c.lcw)))) continue;   /* 3866-3877: sub-blocks of the zeroed area are not coded */
    /* residual_coding_subblock (4164-4304) */
    const int isLast = (scanPosLast >> c.lcg) == sub, isNotFirst = sub != 0;
    const int firstSigPos = isLast ? scanPosLast : maxSub;
    int nextSigPos = firstSigPos;
    if (!isLast && isNotFirst) {
      if (c.sig_group[cgPos]) orc_enc_bin(cb, 1, sigGroupCtx);
      else { orc_enc_bin(cb, 0, sigGroupCtx); continue; }
    }
    uint8_t ctxOffset[16];
    const int inferSigPos = nextSigPos != scanPosLast ? (isNotFirst ? minSub : -1) : nextSigPos;
    int numNonZero = 0, remRegBins = c.reg_bins;
    uint32_t signPattern = 0;
    for (; nextSigPos >= minSub && remRegBins >= 4; nextSigPos--) {
      const int blk = c.scan[nextSigPos];
      const int cf = coeff[blk];
      const unsigned sigFlag = cf != 0;
      if (numNonZero || nextSigPos != inferSigPos) {
        const int ctx = sig_ctx(&c, coeff, blk, state);
        orc_enc_bin(cb, sigFlag, ctx);
        remRegBins--;
      } else if (nextSigPos != scanPosLast) sig_ctx(&c, coeff, blk, state);
      if (sigFlag) {
        const int off = ctx_offset_abs(&c);
        ctxOffset[nextSigPos - minSub] = (uint8_t) off;
        numNonZero++;
        signPattern = (signPattern << 1) | (cf < 0);     /* 4218-4219: no shift before the very first coefficient, where it is 0 anyway */
        int rem = abs(cf) - 1;
        const unsigned gt1 = !!rem;
        orc_enc_bin(cb, gt1, ORC_CTX_GtxFlag[c.ch + 2] + off);     /* greater1CtxIdAbs = m_gtxFlagCtxSet[1] */
        remRegBins--;
        if (gt1) {
          rem -= 1;
          orc_enc_bin(cb, rem & 1, ORC_CTX_ParFlag[c.ch] + off);
          rem >>= 1;
          remRegBins--;
          orc_enc_bin(cb, !!rem, ORC_CTX_GtxFlag[c.ch] + off);     /* greater2CtxIdAbs = m_gtxFlagCtxSet[0] */
          remRegBins--;
        }
      }
      state = (stateTab >> ((state << 2) + ((cf & 1) << 1))) & 3;      /* 4247 */
    }
    const int firstPosMode2 = nextSigPos;
    c.reg_bins = remRegBins;
    /* 2nd pass: Golomb-Rice remainders of context-coded positions (4260-4272) */
    for (int sp = firstSigPos; sp > firstPosMode2; sp--) {
      const int blk = c.scan[sp];
      const unsigned absLevel = (unsigned) abs(coeff[blk]);
      if (absLevel >= 4) {
        const int sumAll = tmpl_abs_sum(&c, coeff, blk, 4);
        orc_enc_rem_abs(cb, (absLevel - 4) >> 1, ORC_GORICE_PARS[sumAll]);
      }
    }
    /* 3rd pass: bypass-coded positions (4275-4294) */
    for (int sp = firstPosMode2; sp >= minSub; sp--) {
      const int blk = c.scan[sp];
      const unsigned absLevel = (unsigned) abs(coeff[blk]);
      const int sumAll = tmpl_abs_sum(&c, coeff, blk, 0);
      const unsigned rice = ORC_GORICE_PARS[sumAll], pos0 = ORC_GORICE_POS0[imax(0, state - 1) * 32 + sumAll];
      const unsigned rem = absLevel == 0 ? pos0 : absLevel <= pos0 ? absLevel - 1 : absLevel;
      orc_enc_rem_abs(cb, rem, rice);
      state = (stateTab >> ((state << 2) + ((absLevel & 1) << 1))) & 3;
      if (absLevel) { numNonZero++; signPattern = (signPattern << 1) | (coeff[blk] < 0); }
    }
    (void) ctxOffset;
    orc_enc_bins_ep(cb, signPattern, numNonZero);    /* sign bits (4297-4303), no sign hiding */
  }
}

uint64_t orc_residual_bits(uint16_t *s0, uint16_t *s1, const int16_t *level, int w, int h, int is_chroma)
{
  orc_cabac c; c.aw = 0; c.dq = 0;
  memcpy(c.s0, s0, sizeof c.s0); memcpy(c.s1, s1, sizeof c.s1); c.bits = 0;
  orc_residual_coding(&c, level, w, h, is_chroma);
  memcpy(s0, c.s0, sizeof c.s0); memcpy(s1, c.s1, sizeof c.s1);
  return c.bits;
}
/* the same with the dependent-quantisation state machine (slice dep_quant_enabled_flag) */
uint64_t orc_residual_bits_dq(uint16_t *s0, uint16_t *s1, const int16_t *level, int w, int h, int is_chroma, int mts_idx)
{
  orc_cabac c; c.aw = 0; c.dq = 1;
  memcpy(c.s0, s0, sizeof c.s0); memcpy(c.s1, s1, sizeof c.s1); c.bits = 0;
  orc_residual_coding_mts(&c, level, w, h, is_chroma, mts_idx);
  memcpy(s0, c.s0, sizeof c.s0); memcpy(s1, c.s1, sizeof c.s1);
  return c.bits;
}

/*
 * orc_rdo.c — recursion level of the oracle (TEST INFRASTRUCTURE ONLY, see vvc_oracle.h).
 *
 * Sequential, depth-first restatement of
 *   EL/EncCu.cpp        compressCtu 428, xCompressCU 727, xCheckModeSplit 1918, xCheckRDCostIntra 2402,
 *                       xCheckBestMode 677, xEncodeDontSplit 5649
 *   EL/EncModeCtrl.cpp  initCULevel 1203, tryMode 1557, useModeResult 2089, nextMode 154
 *   EL/IntraSearch.cpp  estIntraPredLumaQT 289, xRecurIntraCodingLumaQT 3282, xIntraCodingTUBlock 2694,
 *                       estIntraPredChromaQT 1382, xRecurIntraChromaCodingQT 3779, rate helpers 2345-2692, 4263
 *   EL/CABACWriter.cpp  split_cu_mode 1010, intra_luma_pred_mode 1762, extend_ref_line 1566,
 *                       intra_chroma_pred_mode 1891, cbf_comp 3400, transform_unit 3514, coding_tree 474
 *   CL/UnitPartitioner.cpp canSplit 379, getImplicitSplit 530, splitCurrArea 278, nextPart 636
 *   CL/ContextModelling.cpp CtxSplit 154
 * for the tool subset "P0": all 67 angular modes + PDPC + MRL, DCT-II only, plain quantisation
 * (RDOQ/DepQuant/sign-hiding off), no MIP/ISP/LFNST/MTS/TS/BDPCM/CCLM/JointCbCr/LMCS, dual tree,
 * CU-result reuse (BestEncInfoCache) off.  Recursion-level parity is UNPINNED (see vvc_oracle.h).
 */
#include "orc_internal.h"
#include <stdlib.h>
#include <stdio.h>
#include <math.h>

static char g_err[256];
const char *orc_last_error(void) { return g_err; }

static int ilog2(int v) { int r = 0; while (v > 1) { v >>= 1; r++; } return r; }
static int imin(int a, int b) { return a < b ? a : b; }
static int imax(int a, int b) { return a > b ? a : b; }

enum { SPLIT_NONE = 0, SPLIT_QT = 1, SPLIT_BH = 2, SPLIT_BV = 3, SPLIT_TH = 4, SPLIT_TV = 5 };   /* PartSplit values, CL/UnitPartitioner.h:56-65 */
enum { ETM_INTRA, ETM_POST_DONT_SPLIT, ETM_SPLIT_QT, ETM_SPLIT_BT_H, ETM_SPLIT_BT_V, ETM_SPLIT_TT_H, ETM_SPLIT_TT_V, ETM_RECO_CACHED };

typedef struct { int x, y, w, h; } area_t;   /* luma samples (UnitArea::Y) */

typedef struct {       /* per 4x4-luma-unit record of the CU covering it, one map per channel type */
  uint8_t valid, tile, qt_depth, mt_depth, bt_depth, depth, dir, mrl, cbf, lw, lh, mts, lfnst, jccr;
  uint8_t isp, tucbf;  /* cu.ispMode (0, 1 horizontal, 2 vertical split) and the cbf of each of its sub-partitions (bit k = TU k) */
  int16_t x, y;        /* CU origin in channel samples */
  uint64_t split_series;
} unit_t;

typedef struct {       /* PartLevel + Partitioner (CL/UnitPartitioner.h:85-189) */
  int split; area_t parts[4]; int nparts, idx;
  int impl_checked, is_implicit, impl_split;
} part_level;
typedef struct {
  part_level st[20]; int n;
  int depth, qt_depth, bt_depth, mt_depth, impl_bt_depth, ch;
  area_t cur;
} partitioner;

typedef struct {       /* what the mode controller reads from a CodingStructure */
  double cost; uint64_t dist, bits;
  int n_cu, is_split;
  int f_bt, f_depth, f_mt, f_cbf, f_w, f_h;     /* cus.front() */
  int l_bt, l_w, l_h;                            /* cus.back() */
  int max_qt;
} cs_sum;

/* BestEncodingInfo (EL/EncModeCtrl.h:456-475) reduced to what an intra CU needs; one entry per (position in CTU in
 * 4-sample units, log2 w, log2 h) like m_bestEncInfo[x][y][wIdx][hIdx] (EL/EncModeCtrl.cpp:706-760).  The reference
 * keeps entries across CTUs and rejects stale ones by comparing poc and absolute area (987-1024); clearing at every
 * CTU start is equivalent.  lev: w*h luma levels, or Cb then Cr (cw*ch each). */
typedef struct { uint8_t valid, ch, dir, mrl, cbf, depth, mts, lfnst, jccr, isp, tucbf; uint64_t ss; int16_t *lev; } cache_ent;
#define CACHE_ENTRIES (32 * 32 * 6 * 6)

#define MAX_DEPTH 20
typedef struct {       /* per recursion level: saved best reconstruction of the node */
  int16_t *rec[3], *lev[3]; unit_t *units;
} store_t;

struct orc_enc {
  orc_cfg cfg; orc_slice sl;
  int wl, hl, wc, hc;                 /* plane dims */
  int16_t *org[3], *rec[3], *lev[3];  /* original, reconstruction, quantised levels (plane layout) */
  int stride[3];
  unit_t *um[2]; uint8_t *avail[2]; int uw, uh;
  int ctus_w, ctus_h; int *ctu_tile; int cur_tile;       /* cur_tile: the tile's 8-bit tag (index mod 255; neighbouring tiles never share it below 255 tile columns), cur_tile_idx: its index */
  int cur_tile_idx;
  orc_cabac cabac;
  store_t store[MAX_DEPTH];
  double sqrt_lambda_fp;              /* sqrtLambdaForFirstPass */
  uint64_t cnt_satd, cnt_rd, cnt_rdpix, cnt_nodes, cnt_reuse;
  cache_ent *cache; int ctu_is_last;
  int16_t lmcs_fwd[1024], lmcs_inv[1024]; int lmcs_pivot[17], lmcs_cadj[16]; int lmcs_on;      /* LMCS tables of the slice (orc_set_slice) */
  int tu_cadj;                        /* chroma residual scale of the chroma TU being coded (0: no scaling) */
  int dct2_sum;                       /* sum |coefficient| of the luma block transformed last (the DCT-II entry of the transform-skip pruning, CL/TrQuant.cpp:1049-1124) */
  int tu_cbf_cb;                      /* tu.cbf[Cb] while Cr is quantised (context of its cbf in DepQuant's rate tables) */
  int jccr_sign;                      /* slice joint_cb_cr_sign_flag, from the picture's chroma planes (EL/EncSlice.cpp:1503-1538) */
  int16_t *pred_c[2], *resi_c[2];     /* JointCbCr: predictions and residuals of the chroma block pair */
  orc_forest forest; int32_t *dump; int dump_cap, dump_n; uint64_t cnt_fast;
  /* scratch */
  int16_t *ref_unf, *ref_flt, *pred, *resi, *resi_org, *tmp_rec[2], *tmp_lev[2], *best_rec[2], *best_lev[2];
  int *coef;
};

/* ------------------------------------------------------------------------------------------------ */
orc_enc *orc_create(const orc_cfg *cfg)
{
  if (cfg->tools & ~(uint32_t) (ORC_TOOL_MRL | ORC_TOOL_CU_REUSE | ORC_TOOL_CCLM | ORC_TOOL_FAST | ORC_TOOL_MTS | ORC_TOOL_MIP | ORC_TOOL_DEPQUANT | ORC_TOOL_LFNST | ORC_TOOL_JCCR | ORC_TOOL_TS | ORC_TOOL_ISP | ORC_TOOL_LMCS | ORC_TOOL_RDOQ | ORC_TOOL_WPP)) { snprintf(g_err, sizeof g_err, "oracle: tool set 0x%x not built yet (built: MRL, MIP, ISP, LFNST, MTS, TS, DepQuant (RDOQ only behind it: RDOQ-TS), LMCS, CCLM, JointCbCr, CU reuse, FAST, WPP)", cfg->tools); return 0; }
  /* the plain quantiser's LFNST branch (CL/Quant.cpp:1054-1058) keeps buffer positions the decoder's LFNST conditions reject: the reference only
   * ever runs LFNST over DepQuant / RDOQ */
  if ((cfg->tools & ORC_TOOL_JCCR) && !(cfg->tools & ORC_TOOL_DEPQUANT)) { snprintf(g_err, sizeof g_err, "oracle: JointCbCr is built over DepQuant (tool set 0x%x)", cfg->tools); return 0; }
  if ((cfg->tools & ORC_TOOL_LFNST) && !(cfg->tools & ORC_TOOL_DEPQUANT)) { snprintf(g_err, sizeof g_err, "oracle: LFNST needs DepQuant (tool set 0x%x)", cfg->tools); return 0; }
  /* transform skip is built as the reference cfg runs it: the {DCT2, TS} candidates of the LFNST branch of xRecurIntraCodingLumaQT, RDOQ-TS behind DepQuant::quant */
  if ((cfg->tools & ORC_TOOL_TS) && (~cfg->tools & (ORC_TOOL_DEPQUANT | ORC_TOOL_LFNST))) { snprintf(g_err, sizeof g_err, "oracle: transform skip needs DepQuant and LFNST (tool set 0x%x)", cfg->tools); return 0; }
  /* ISP likewise: its candidates sit in the first pass of the LFNST pass loop, its blocks are quantised by DepQuant, their transforms follow getTrTypes with MTS on */
  if ((cfg->tools & ORC_TOOL_ISP) && (~cfg->tools & (ORC_TOOL_DEPQUANT | ORC_TOOL_LFNST | ORC_TOOL_MTS))) { snprintf(g_err, sizeof g_err, "oracle: ISP needs DepQuant, LFNST and MTS (tool set 0x%x)", cfg->tools); return 0; }
  if (!cfg->dual_tree || cfg->ctu_size != 128) { snprintf(g_err, sizeof g_err, "oracle: only DualITree=1, CTUSize=128"); return 0; }
  if ((cfg->pic_w & 7) || (cfg->pic_h & 7)) { snprintf(g_err, sizeof g_err, "oracle: picture size must be a multiple of 8 (EncAppCfg.cpp:2709)"); return 0; }
  orc_enc *e = (orc_enc *) calloc(1, sizeof *e);
  e->cfg = *cfg;
  e->cabac.dq = (cfg->tools & ORC_TOOL_DEPQUANT) ? 1 : 0;
  e->wl = cfg->pic_w; e->hl = cfg->pic_h; e->wc = e->wl >> 1; e->hc = e->hl >> 1;
  for (int c = 0; c < 3; c++) {
    const int w = c ? e->wc : e->wl, h = c ? e->hc : e->hl;
    e->stride[c] = w;
    e->org[c] = (int16_t *) calloc((size_t) w * h, 2); e->rec[c] = (int16_t *) calloc((size_t) w * h, 2); e->lev[c] = (int16_t *) calloc((size_t) w * h, 2);
  }
  e->uw = (e->wl + 3) >> 2; e->uh = (e->hl + 3) >> 2;
  for (int k = 0; k < 2; k++) { e->um[k] = (unit_t *) calloc((size_t) e->uw * e->uh, sizeof(unit_t)); e->avail[k] = (uint8_t *) calloc((size_t) e->uw * e->uh, 1); }
  e->ctus_w = (e->wl + 127) >> 7; e->ctus_h = (e->hl + 127) >> 7;
  e->ctu_tile = (int *) calloc((size_t) e->ctus_w * e->ctus_h, sizeof(int));
  for (int d = 0; d < MAX_DEPTH; d++) {
    for (int c = 0; c < 3; c++) { e->store[d].rec[c] = (int16_t *) malloc(128 * 128 * 2); e->store[d].lev[c] = (int16_t *) malloc(128 * 128 * 2); }
    e->store[d].units = (unit_t *) malloc(32 * 32 * sizeof(unit_t));
  }
  e->cache = (cache_ent *) calloc(CACHE_ENTRIES, sizeof(cache_ent));
  e->ref_unf = (int16_t *) malloc(2 * 300 * 300); e->ref_flt = (int16_t *) malloc(2 * 300 * 300);
  for (int k = 0; k < 2; k++) { e->pred_c[k] = (int16_t *) malloc(64 * 64 * 2); e->resi_c[k] = (int16_t *) malloc(64 * 64 * 2); }
  e->pred = (int16_t *) malloc(128 * 128 * 2); e->resi = (int16_t *) malloc(128 * 128 * 2); e->resi_org = (int16_t *) malloc(128 * 128 * 2); e->coef = (int *) malloc(128 * 128 * 4);
  for (int k = 0; k < 2; k++) { e->tmp_rec[k] = (int16_t *) malloc(128 * 128 * 2); e->tmp_lev[k] = (int16_t *) malloc(128 * 128 * 2); e->best_rec[k] = (int16_t *) malloc(128 * 128 * 2); e->best_lev[k] = (int16_t *) malloc(128 * 128 * 2); }
  return e;
}
void orc_destroy(orc_enc *e)
{
  if (!e) return;
  for (int c = 0; c < 3; c++) { free(e->org[c]); free(e->rec[c]); free(e->lev[c]); }
  for (int k = 0; k < 2; k++) { free(e->um[k]); free(e->avail[k]); free(e->tmp_rec[k]); free(e->tmp_lev[k]); free(e->best_rec[k]); free(e->best_lev[k]); }
  for (int d = 0; d < MAX_DEPTH; d++) { for (int c = 0; c < 3; c++) { free(e->store[d].rec[c]); free(e->store[d].lev[c]); } free(e->store[d].units); }
  for (int i = 0; i < CACHE_ENTRIES; i++) free(e->cache[i].lev);
  free(e->forest.root); free(e->forest.feature); free(e->forest.left); free(e->forest.right); free(e->forest.threshold); free(e->forest.value);
  free(e->cache); free(e->ctu_tile); free(e->ref_unf); free(e->ref_flt); free(e->pred); free(e->resi); free(e->resi_org); free(e->coef); free(e);
}
static void *dup_mem(const void *p, size_t n) { void *d = malloc(n ? n : 1); memcpy(d, p, n); return d; }
int orc_set_forest(orc_enc *e, int n_trees, int n_nodes, int n_classes, const int32_t *root, const int32_t *feature, const double *threshold,
                   const int32_t *left, const int32_t *right, const double *value, const int32_t *classes)
{
  if (n_trees < 1 || n_nodes < n_trees || n_classes < 1 || n_classes > 8) { snprintf(g_err, sizeof g_err, "oracle: bad forest shape"); return -1; }
  for (int i = 0; i < n_nodes; i++)
    if (left[i] >= 0 && (left[i] >= n_nodes || right[i] < 0 || right[i] >= n_nodes || feature[i] < 0 || feature[i] >= 26)) { snprintf(g_err, sizeof g_err, "oracle: forest node %d out of range", i); return -1; }
  orc_forest *f = &e->forest;
  free(f->root); free(f->feature); free(f->left); free(f->right); free(f->threshold); free(f->value);
  f->n_trees = n_trees; f->n_nodes = n_nodes; f->n_classes = n_classes;
  f->root = (int32_t *) dup_mem(root, (size_t) n_trees * 4); f->feature = (int32_t *) dup_mem(feature, (size_t) n_nodes * 4);
  f->left = (int32_t *) dup_mem(left, (size_t) n_nodes * 4); f->right = (int32_t *) dup_mem(right, (size_t) n_nodes * 4);
  f->threshold = (double *) dup_mem(threshold, (size_t) n_nodes * 8); f->value = (double *) dup_mem(value, (size_t) n_nodes * n_classes * 8);
  for (int c = 0; c < n_classes; c++) f->classes[c] = classes[c];
  return 0;
}
int orc_set_training_dump(orc_enc *e, int32_t *rows, int cap_rows) { e->dump = rows; e->dump_cap = cap_rows; e->dump_n = 0; return 0; }
int orc_training_rows(const orc_enc *e) { return e->dump_n; }
int orc_fast_features(const int16_t *org, int stride, int w, int h, int feat[26]) { memset(feat, 0, 26 * sizeof(int)); orc_fast_block_features(org, stride, w, h, feat); return 0; }
int orc_forest_predict_rows(orc_enc *e, const int32_t *rows, int n, int32_t *out)
{
  if (!e->forest.n_trees) return -1;
  for (int i = 0; i < n; i++) { int f[26]; for (int k = 0; k < 26; k++) f[k] = rows[i * 26 + k]; out[i] = orc_forest_predict(&e->forest, f); }
  return 0;
}
int orc_set_slice(orc_enc *e, const orc_slice *s)
{
  e->sl = *s;
  e->lmcs_on = 0;
  if (s->lmcs_enable) {
    if (!(e->cfg.tools & ORC_TOOL_LMCS)) { snprintf(g_err, sizeof g_err, "oracle: the slice enables LMCS, the tool set does not"); return -1; }
    /* Reshape::constructReshaper (CL/Reshape.cpp:297-333, JVET_O0428 form) */
    const int bd = e->cfg.bit_depth, n = 1 << bd, initCW = n / 16, lbin = ilog2(initCW);
    int binCW[16], inPivot[17], fwdCoef[16], invCoef[16];
    for (int i = 0; i < 16; i++) binCW[i] = (i < s->lmcs_min_bin || i > s->lmcs_max_bin) ? 0 : (uint16_t) (s->lmcs_delta_cw[i] + initCW);
    e->lmcs_pivot[0] = 0; inPivot[0] = 0;
    for (int i = 0; i < 16; i++) {
      e->lmcs_pivot[i + 1] = e->lmcs_pivot[i] + binCW[i]; inPivot[i + 1] = inPivot[i] + initCW;
      fwdCoef[i] = (binCW[i] * (1 << 11) + (1 << (lbin - 1))) >> lbin;
      if (binCW[i] == 0) { invCoef[i] = 0; e->lmcs_cadj[i] = 1 << 11; } else { invCoef[i] = initCW * (1 << 11) / binCW[i]; e->lmcs_cadj[i] = invCoef[i]; }
    }
    for (int v = 0; v < n; v++) {
      const int iy = v / initCW;
      int t = e->lmcs_pivot[iy] + ((fwdCoef[iy] * (v - inPivot[iy]) + (1 << 10)) >> 11);
      e->lmcs_fwd[v] = (int16_t) (t < 0 ? 0 : t > n - 1 ? n - 1 : t);
      int ii = s->lmcs_min_bin;                      /* getPWLIdxInv 268-283 */
      for (; ii <= s->lmcs_max_bin; ii++) if (v < e->lmcs_pivot[ii + 1]) break;
      if (ii > 15) ii = 15;
      t = inPivot[ii] + ((invCoef[ii] * (v - e->lmcs_pivot[ii]) + (1 << 10)) >> 11);
      e->lmcs_inv[v] = (int16_t) (t < 0 ? 0 : t > n - 1 ? n - 1 : t);
    }
    e->lmcs_on = 1;
  }
  /* EL/IntraSearch.cpp:297: getMotionLambda()*FRAC_BITS_SCALE = sqrt(lambda)/32768 (CL/RdCost.cpp:80) */
  e->sqrt_lambda_fp = sqrt(s->lambda) * (1.0 / (double) (1 << 15));
  return 0;
}
/* setJointCbCrModes (EL/EncSlice.cpp:1503-1538): the sign of the inter-chroma transform from the correlation of the high-pass filtered Cb and Cr
 * planes (interior samples); a slice-level input of the search, derived here from the loaded picture like the reference does before compressSlice */
int orc_jccr_sign(const int16_t *cb, const int16_t *cr, int stride, int w, int h)
{
  int64_t sum = 0;
  for (int y = 1; y < h - 1; y++) for (int x = 1; x < w - 1; x++) {
    const int16_t *p = cb + y * stride + x, *q = cr + y * stride + x;
    const int a = 12 * p[0] - 2 * (p[-1] + p[1] + p[-stride] + p[stride]) - (p[-1 - stride] + p[1 - stride] + p[-1 + stride] + p[1 + stride]);
    const int b = 12 * q[0] - 2 * (q[-1] + q[1] + q[-stride] + q[stride]) - (q[-1 - stride] + q[1 - stride] + q[-1 + stride] + q[1 + stride]);
    sum += (int64_t) a * b;
  }
  return sum < 0;
}
int orc_load_frame(orc_enc *e, const void *const org[3], const int stride[3], int bps)
{
  for (int c = 0; c < 3; c++) {
    const int w = c ? e->wc : e->wl, h = c ? e->hc : e->hl;
    for (int y = 0; y < h; y++) for (int x = 0; x < w; x++)
      e->org[c][y * e->stride[c] + x] = bps == 1 ? ((const uint8_t *) org[c])[y * stride[c] + x] : (int16_t) ((const uint16_t *) org[c])[y * stride[c] + x];
    memset(e->rec[c], 0, (size_t) w * h * 2); memset(e->lev[c], 0, (size_t) w * h * 2);
  }
  /* EncGOP::xPicInitLMCS (EL/EncGOP.cpp:1689-1695): the original luma of an intra picture is forward mapped once, before the slice is compressed */
  if (e->lmcs_on) for (int i = 0; i < e->wl * e->hl; i++) e->org[0][i] = e->lmcs_fwd[e->org[0][i]];
  for (int k = 0; k < 2; k++) { memset(e->um[k], 0, (size_t) e->uw * e->uh * sizeof(unit_t)); memset(e->avail[k], 0, (size_t) e->uw * e->uh); }
  e->jccr_sign = orc_jccr_sign(e->org[1], e->org[2], e->stride[1], e->wc, e->hc);
  /* uniform tile grid (CL/Slice.cpp PPS uniform spacing): boundary i = i*N/T */
  for (int ry = 0; ry < e->ctus_h; ry++) for (int rx = 0; rx < e->ctus_w; rx++) {
    int tc = 0, tr = 0;
    for (int i = 0; i < e->cfg.tile_cols; i++) if (rx >= (i * e->ctus_w) / e->cfg.tile_cols) tc = i;
    for (int i = 0; i < e->cfg.tile_rows; i++) if (ry >= (i * e->ctus_h) / e->cfg.tile_rows) tr = i;
    e->ctu_tile[ry * e->ctus_w + rx] = tr * e->cfg.tile_cols + tc;
  }
  return 0;
}
int orc_get_reco(orc_enc *e, void *const reco[3], const int stride[3], int bps)
{
  for (int c = 0; c < 3; c++) {
    const int w = c ? e->wc : e->wl, h = c ? e->hc : e->hl;
    for (int y = 0; y < h; y++) for (int x = 0; x < w; x++) {
      const int16_t v = e->rec[c][y * e->stride[c] + x];
      if (bps == 1) ((uint8_t *) reco[c])[y * stride[c] + x] = (uint8_t) v; else ((uint16_t *) reco[c])[y * stride[c] + x] = (uint16_t) v;
    }
  }
  return 0;
}
int orc_lmcs_tables(orc_enc *e, int16_t *fwd, int16_t *inv, int *pivot, int *cadj)
{
  if (!e->lmcs_on) return -1;
  memcpy(fwd, e->lmcs_fwd, (size_t) (1 << e->cfg.bit_depth) * 2); memcpy(inv, e->lmcs_inv, (size_t) (1 << e->cfg.bit_depth) * 2);
  memcpy(pivot, e->lmcs_pivot, sizeof e->lmcs_pivot); memcpy(cadj, e->lmcs_cadj, sizeof e->lmcs_cadj);
  return 0;
}
int orc_lmcs_inverse_reco(orc_enc *e)
{
  if (!e->lmcs_on) return -1;
  for (int i = 0; i < e->wl * e->hl; i++) e->rec[0][i] = e->lmcs_inv[e->rec[0][i]];
  return 0;
}
void orc_get_counters(orc_enc *e, uint64_t out[4]) { out[0] = e->cnt_satd; out[1] = e->cnt_rd; out[2] = e->cnt_rdpix; out[3] = e->cnt_nodes; }

static double rd_cost(const orc_enc *e, uint64_t bits, uint64_t dist) { return orc_calc_rd_cost(e->sl.lambda, bits, dist); }

/* ------------------------------------------------------------------------------------------------
 * neighbour CU lookup: cs.getCU / getCURestricted reduced to "coded in the current partition path,
 * same tile" (CL/CodingStructure.cpp:291-348,1629-1658).  px,py in samples of channel type ch.
 * ---------------------------------------------------------------------------------------------- */
static const unit_t *get_cu(const orc_enc *e, int ch, int px, int py)
{
  const int W = ch ? e->wc : e->wl, H = ch ? e->hc : e->hl, ul = ch ? 1 : 2;
  if (px < 0 || py < 0 || px >= W || py >= H) return 0;
  const unit_t *u = &e->um[ch][(py >> ul) * e->uw + (px >> ul)];
  return (u->valid && u->tile == e->cur_tile) ? u : 0;
}
static void set_units(orc_enc *e, int ch, area_t a, const unit_t *proto, int valid)
{
  /* a in luma samples; clip to picture */
  const int x1 = imin(a.x + a.w, e->wl), y1 = imin(a.y + a.h, e->hl);
  for (int y = a.y >> 2; y < (y1 + 3) >> 2; y++) for (int x = a.x >> 2; x < (x1 + 3) >> 2; x++) {
    unit_t *u = &e->um[ch][y * e->uw + x];
    if (proto) *u = *proto;
    u->valid = (uint8_t) valid; u->tile = (uint8_t) e->cur_tile;
    e->avail[ch][y * e->uw + x] = valid ? (uint8_t) (e->cur_tile + 1) : 0;
  }
}

/* ------------------------------------------------------------------------------------------------
 * Partitioner
 * ---------------------------------------------------------------------------------------------- */
static void part_init_ctu(partitioner *P, area_t ctu, int ch)
{
  memset(P, 0, sizeof *P);
  P->ch = ch; P->cur = ctu; P->n = 1;
  P->st[0].split = 0; P->st[0].parts[0] = ctu; P->st[0].nparts = 1;
}
static uint64_t part_split_series(const partitioner *P)
{
  uint64_t s = 0; int d = 0;
  for (int i = 0; i < P->n; i++) { if (P->st[i].split == 0) continue; s += (uint64_t) P->st[i].split << (d * 5); d++; }
  return s;
}
/* getImplicitSplit, CL/UnitPartitioner.cpp:530-581 */
static int part_implicit_split(const orc_enc *e, partitioner *P)
{
  part_level *L = &P->st[P->n - 1];
  if (L->impl_checked) return L->impl_split;
  int split = SPLIT_NONE;
  const area_t a = P->cur;
  const int blIn = a.x < e->wl && (a.y + a.h - 1) < e->hl;       /* picture.contains(bottomLeft) */
  const int trIn = (a.x + a.w - 1) < e->wl && a.y < e->hl;
  const int maxBt = e->cfg.max_bt_size[P->ch], minQt = e->cfg.min_qt[P->ch];
  const int btAllowed = a.w <= maxBt && a.h <= maxBt;
  const int qtAllowed = a.w > minQt && a.h > minQt && P->bt_depth == 0;
  if (!blIn && !trIn && qtAllowed) split = SPLIT_QT;
  else if (!blIn && btAllowed) split = SPLIT_BH;
  else if (!trIn && btAllowed) split = SPLIT_BV;
  else if (!blIn || !trIn) split = SPLIT_QT;
  if (e->cfg.dual_tree && (a.w > 64 || a.h > 64)) split = SPLIT_QT;
  if ((!blIn || !trIn) && (a.w > 64 || a.h > 64)) split = SPLIT_QT;
  L->impl_checked = 1; L->is_implicit = split != SPLIT_NONE; L->impl_split = split;
  return split;
}
/* canSplit, CL/UnitPartitioner.cpp:379-466; can[] = {no, qt, bh, bv, th, tv} */
static void part_can_split(const orc_enc *e, partitioner *P, int can[6])
{
  const int impl = part_implicit_split(e, P);
  const int ch = P->ch;
  const int maxBTD = e->cfg.max_bt_depth[ch] + P->impl_bt_depth;
  const int maxBt = e->cfg.max_bt_size[ch], minBt = 4, maxTt = e->cfg.max_tt_size[ch], minTt = 4, minQt = e->cfg.min_qt[ch];
  const area_t a = P->cur;
  const int cw = a.w >> 1, chh = a.h >> 1;      /* areaC */
  for (int i = 0; i < 6; i++) can[i] = 1;
  int canBtt = P->mt_depth < maxBTD;
  const part_level *L = &P->st[P->n - 1];
  const int last = L->split;
  const int parl = last == SPLIT_TH ? SPLIT_BH : SPLIT_BV;
  if (last != 0 && last != SPLIT_QT) can[1] = 0;
  if (a.w <= minQt) can[1] = 0;
  if (ch == 1 && cw <= 4) can[1] = 0;
  if (impl != SPLIT_NONE) { can[0] = can[4] = can[5] = 0; can[2] = impl == SPLIT_BH; can[3] = impl == SPLIT_BV; return; }
  if ((last == SPLIT_TH || last == SPLIT_TV) && L->idx == 1) { can[2] = parl != SPLIT_BH; can[3] = parl != SPLIT_BV; }
  if (canBtt && (a.w <= minBt && a.h <= minBt) && (a.w <= minTt && a.h <= minTt)) canBtt = 0;
  if (canBtt && (a.w > maxBt || a.h > maxBt) && (a.w > maxTt || a.h > maxTt)) canBtt = 0;
  if (!canBtt) { can[2] = can[3] = can[4] = can[5] = 0; return; }
  if (a.w > maxBt || a.h > maxBt) can[2] = can[3] = 0;
  if (a.h <= minBt) can[2] = 0;
  if (a.w > 64 && a.h <= 64) can[2] = 0;
  if (ch == 1 && cw * chh <= 16) can[2] = 0;
  if (a.w <= minBt) can[3] = 0;
  if (a.w <= 64 && a.h > 64) can[3] = 0;
  if (ch == 1 && cw * chh <= 16) can[3] = 0;
  if (a.h <= 2 * minTt || a.h > maxTt || a.w > maxTt) can[4] = 0;
  if (a.w > 64 || a.h > 64) can[4] = 0;
  if (ch == 1 && cw * chh <= 32) can[4] = 0;
  if (a.w <= 2 * minTt || a.w > maxTt || a.h > maxTt) can[5] = 0;
  if (a.w > 64 || a.h > 64) can[5] = 0;
  if (ch == 1 && cw * chh <= 32) can[5] = 0;
}
static int part_can(const orc_enc *e, partitioner *P, int split) { int c[6]; part_can_split(e, P, c); return c[split]; }
/* splitCurrArea 278-377 + getCUSubPartitions 785-... */
static void part_split(const orc_enc *e, partitioner *P, int split)
{
  const int isImpl = split == part_implicit_split(e, P);
  const area_t a = P->cur;
  part_level *L = &P->st[P->n++];
  memset(L, 0, sizeof *L);
  L->split = split;
  switch (split) {
    case SPLIT_QT: L->nparts = 4;
      for (int i = 0; i < 4; i++) { L->parts[i].w = a.w >> 1; L->parts[i].h = a.h >> 1; L->parts[i].x = a.x + ((i & 1) ? a.w >> 1 : 0); L->parts[i].y = a.y + ((i >= 2) ? a.h >> 1 : 0); }
      break;
    case SPLIT_BH: L->nparts = 2;
      L->parts[0] = (area_t) { a.x, a.y, a.w, a.h >> 1 }; L->parts[1] = (area_t) { a.x, a.y + (a.h >> 1), a.w, a.h >> 1 }; break;
    case SPLIT_BV: L->nparts = 2;
      L->parts[0] = (area_t) { a.x, a.y, a.w >> 1, a.h }; L->parts[1] = (area_t) { a.x + (a.w >> 1), a.y, a.w >> 1, a.h }; break;
    case SPLIT_TH: L->nparts = 3;
      L->parts[0] = (area_t) { a.x, a.y, a.w, a.h >> 2 }; L->parts[1] = (area_t) { a.x, a.y + (a.h >> 2), a.w, a.h >> 1 }; L->parts[2] = (area_t) { a.x, a.y + (a.h >> 2) + (a.h >> 1), a.w, a.h >> 2 }; break;
    case SPLIT_TV: L->nparts = 3;
      L->parts[0] = (area_t) { a.x, a.y, a.w >> 2, a.h }; L->parts[1] = (area_t) { a.x + (a.w >> 2), a.y, a.w >> 1, a.h }; L->parts[2] = (area_t) { a.x + (a.w >> 2) + (a.w >> 1), a.y, a.w >> 2, a.h }; break;
  }
  P->depth++;
  P->cur = L->parts[0];
  if (split != SPLIT_QT) {
    P->bt_depth++;
    if (isImpl) P->impl_bt_depth++;
    P->mt_depth++;
    if (split == SPLIT_TH || split == SPLIT_TV) P->bt_depth++;
  } else { P->mt_depth = 0; P->bt_depth = 0; P->qt_depth++; }
}
/* exitCurrSplit 583-634 */
static void part_exit(partitioner *P)
{
  const int split = P->st[P->n - 1].split, idx = P->st[P->n - 1].idx;
  P->n--;
  P->depth--;
  part_level *L = &P->st[P->n - 1];
  P->cur = L->parts[L->idx];
  if (split != SPLIT_QT) {
    P->mt_depth--;
    if (L->is_implicit) P->impl_bt_depth--;
    P->bt_depth--;
    if ((split == SPLIT_TH || split == SPLIT_TV) && idx != 1) P->bt_depth--;
  } else P->qt_depth--;
}
/* nextPart 636-675 (autoPop = false) */
static int part_next(partitioner *P)
{
  part_level *L = &P->st[P->n - 1];
  const int idx = ++L->idx;
  L->impl_checked = 0; L->is_implicit = 0;
  if (idx < L->nparts) {
    if (L->split == SPLIT_TH || L->split == SPLIT_TV) { if (idx == 1) P->bt_depth--; else P->bt_depth++; }
    P->cur = L->parts[idx];
    return 1;
  }
  return 0;
}

/* ------------------------------------------------------------------------------------------------
 * Syntax on the estimator
 * ---------------------------------------------------------------------------------------------- */
/* DeriveCtx::CtxSplit (CL/ContextModelling.cpp:154-250): ctx = {split, qt, hv, hor12, ver12} */
static void derive_split_ctx(const orc_enc *e, const partitioner *P, const int can[6], unsigned ctx[5])
{
  const int ch = P->ch, sh = ch ? 1 : 0;
  const int bx = P->cur.x >> sh, by = P->cur.y >> sh, bw = P->cur.w >> sh, bh = P->cur.h >> sh;
  const unit_t *cuL = get_cu(e, ch, bx - 1, by), *cuA = get_cu(e, ch, bx, by - 1);
  unsigned ctxSpl = 0;
  if (cuL) ctxSpl += ((1 << cuL->lh) < bh) ? 1 : 0;
  if (cuA) ctxSpl += ((1 << cuA->lw) < bw) ? 1 : 0;
  unsigned numSplit = 0;
  if (can[1]) numSplit += 2;
  for (int i = 2; i < 6; i++) if (can[i]) numSplit += 1;
  if (numSplit > 0) numSplit--;
  ctxSpl += 3 * (numSplit >> 1);
  unsigned ctxQt = (cuL && cuL->qt_depth > P->qt_depth) ? 1 : 0;
  ctxQt += (cuA && cuA->qt_depth > P->qt_depth) ? 1 : 0;
  ctxQt += P->qt_depth < 2 ? 0 : 3;
  unsigned ctxHv = 0;
  const unsigned numHor = (unsigned) (can[2] + can[4]), numVer = (unsigned) (can[3] + can[5]);
  if (numVer == numHor) {
    const unsigned wAbove = cuA ? (1u << cuA->lw) : 1, hLeft = cuL ? (1u << cuL->lh) : 1;
    const unsigned depAbove = (unsigned) bw / wAbove, depLeft = (unsigned) bh / hLeft;
    if (depAbove == depLeft || !cuL || !cuA) ctxHv = 0; else if (depAbove < depLeft) ctxHv = 1; else ctxHv = 2;
  } else if (numVer < numHor) ctxHv = 3; else ctxHv = 4;
  ctx[0] = ctxSpl; ctx[1] = ctxQt; ctx[2] = ctxHv; ctx[3] = P->mt_depth <= 1 ? 1 : 0; ctx[4] = P->mt_depth <= 1 ? 3 : 2;
}
/* CABACWriter::split_cu_mode (EL/CABACWriter.cpp:1010-1069) */
static void enc_split_cu_mode(orc_enc *e, partitioner *P, int split)
{
  int can[6]; part_can_split(e, P, can);
  unsigned cx[5]; derive_split_ctx(e, P, can, cx);
  const unsigned ctxSpl = cx[0], ctxQt = cx[1], ctxHv = cx[2], ctxH12 = cx[3], ctxV12 = cx[4];

  const int canSplit = can[1] || can[2] || can[3] || can[4] || can[5];
  const int isNo = split == SPLIT_NONE;
  if (can[0] && canSplit) orc_enc_bin(&e->cabac, !isNo, ORC_CTX_SplitFlag + (int) ctxSpl);
  if (isNo) return;
  const int canBtt = can[2] || can[3] || can[4] || can[5];
  const int isQt = split == SPLIT_QT;
  if (can[1] && canBtt) orc_enc_bin(&e->cabac, (unsigned) isQt, ORC_CTX_SplitQtFlag + (int) ctxQt);
  if (isQt) return;
  const int canHor = can[2] || can[4], canVer = can[3] || can[5];
  const int isVer = split == SPLIT_BV || split == SPLIT_TV;
  if (canVer && canHor) orc_enc_bin(&e->cabac, (unsigned) isVer, ORC_CTX_SplitHvFlag + (int) ctxHv);
  const int can14 = isVer ? can[5] : can[4], can12 = isVer ? can[3] : can[2];
  const int is12 = isVer ? (split == SPLIT_BV) : (split == SPLIT_BH);
  if (can12 && can14) orc_enc_bin(&e->cabac, (unsigned) is12, ORC_CTX_Split12Flag + (int) (isVer ? ctxV12 : ctxH12));
}

/* PU::getIntraMPMs neighbour lookup (CL/UnitTools.cpp:516-532): left PU at bottom-left, above PU at
 * top-right and only inside the same CTU */
/* a MIP CU (cu.mipFlag) is carried as bit 7 of the unit's mrl field (MIP forces multiRefIdx 0) with the MIP mode in dir; every reader of a
 * neighbour's luma mode goes through PU::getIntraDirLuma (CL/UnitTools.cpp:786-800), which is PLANAR for a MIP block */
#define MIP_FLAG 0x80
static int unit_luma_dir(const unit_t *u) { return (u->mrl & MIP_FLAG) ? ORC_PLANAR : u->dir; }
static int mpm_neighbours(const orc_enc *e, int x, int y, int w, int h, int *L, int *A)
{
  *L = ORC_PLANAR; *A = ORC_PLANAR;
  const unit_t *uL = get_cu(e, 0, x - 1, y + h - 1);
  if (uL) *L = unit_luma_dir(uL);
  const unit_t *uA = get_cu(e, 0, x + w - 1, y - 1);
  if (uA && ((y - 1) >> 7) == (y >> 7)) *A = unit_luma_dir(uA);
  return *L == *A ? 1 : 2;
}
static void get_mpms(const orc_enc *e, int x, int y, int w, int h, unsigned mpm[6])
{
  int L, A; mpm_neighbours(e, x, y, w, h, &L, &A);
  orc_get_mpms(L, A, mpm);
}
/* mip_flag is present (EL/CABACWriter.cpp:4741-4767): SPS MIP on, both sides <= MaxTbSize (64), and mipModesAvailable: getNumModesMip
 * (CL/UnitTools.cpp:4688-4708) is 0 for blocks beyond 4:1 */
static int mip_signalled(const orc_enc *e, int w, int h) { return (e->cfg.tools & ORC_TOOL_MIP) && w <= 64 && h <= 64 && orc_mip_num_modes(w, h) > 0; }

/* CABACWriter::intra_luma_pred_mode (1762-1845) with extend_ref_line (1566-1591); mip_flag / isp_mode
 * write nothing when the tools are off in the SPS.  mrl carries MIP_FLAG for a MIP CU (dir = MIP mode) */
/* CU::canUseISP (CL/UnitTools.cpp:414-435): more than 16 samples, no side above MaxTbSize; CU::getISPSplitDim (437-459): the sub-partition size along the split
 * direction (a quarter of the side, but at least 16 samples per sub-partition) */
static int can_use_isp(const orc_enc *e, int w, int h) { return (e->cfg.tools & ORC_TOOL_ISP) && ilog2(w) + ilog2(h) > 4 && w <= 64 && h <= 64; }
static int isp_split_dim(int w, int h, int hor)
{
  const int split = hor ? h : w, non = hor ? w : h;
  const int factor = non < 16 ? 16 >> ilog2(non) : 1;
  return (split >> 2) < factor ? factor : (split >> 2);
}
static void enc_intra_luma_pred_mode_isp(orc_enc *e, int x, int y, int w, int h, int dir, int mrl, int isp);
static void enc_intra_luma_pred_mode(orc_enc *e, int x, int y, int w, int h, int dir, int mrl) { enc_intra_luma_pred_mode_isp(e, x, y, w, h, dir, mrl, 0); }
/* isp: cu.ispMode.  isp_mode (EL/CABACWriter.cpp:3944-3965) follows the reference line index of every luma CU that could use ISP */
static void enc_intra_luma_pred_mode_isp(orc_enc *e, int x, int y, int w, int h, int dir, int mrl, int isp)
{
  orc_cabac *c = &e->cabac;
  if (mip_signalled(e, w, h)) {
    /* DeriveCtx::CtxMipFlag (CL/ContextModelling.cpp:555-569); mip_pred_mode 4781-4787 = xWriteTruncBinCode(mode, numModes) 1519-1564 */
    const unit_t *uL = get_cu(e, 0, x - 1, y), *uA = get_cu(e, 0, x, y - 1);
    unsigned ctx = (unsigned) ((uL && (uL->mrl & MIP_FLAG)) + (uA && (uA->mrl & MIP_FLAG)));
    if (w > 2 * h || h > 2 * w) ctx = 3;
    orc_enc_bin(c, (mrl & MIP_FLAG) != 0, ORC_CTX_MipFlag + (int) ctx);
    if (mrl & MIP_FLAG) {
      const int n = orc_mip_num_modes(w, h), thresh = ilog2(n), val = 1 << thresh, b = n - val;
      if (dir < val - b) orc_enc_bins_ep(c, (uint32_t) dir, thresh); else orc_enc_bins_ep(c, (uint32_t) (dir + val - b), thresh + 1);
      return;
    }
  }
  const int firstLine = (y & 127) == 0;
  if (!firstLine) {
    orc_enc_bin(c, mrl != 0, ORC_CTX_MultiRefLineIdx + 0);
    if (mrl != 0) orc_enc_bin(c, mrl != 1, ORC_CTX_MultiRefLineIdx + 1);
  }
  if (!mrl && can_use_isp(e, w, h)) { orc_enc_bin(c, isp != 0, ORC_CTX_ISPMode + 0); if (isp) orc_enc_bin(c, (unsigned) (isp - 1), ORC_CTX_ISPMode + 1); }
  unsigned mpm[6]; get_mpms(e, x, y, w, h, mpm);
  int mpm_idx = 6;
  for (int i = 0; i < 6; i++) if ((unsigned) dir == mpm[i]) { mpm_idx = i; break; }
  if (!mrl) orc_enc_bin(c, mpm_idx < 6, ORC_CTX_IntraLumaMpmFlag);
  if (mpm_idx < 6) {
    if (mrl == 0) orc_enc_bin(c, mpm_idx > 0, ORC_CTX_IntraLumaPlanarFlag + (isp ? 0 : 1));   /* 1812: context by cu.ispMode */
    if (mpm_idx) orc_enc_bins_ep(c, mpm_idx > 1, 1);
    if (mpm_idx > 1) orc_enc_bins_ep(c, mpm_idx > 2, 1);
    if (mpm_idx > 2) orc_enc_bins_ep(c, mpm_idx > 3, 1);
    if (mpm_idx > 3) orc_enc_bins_ep(c, mpm_idx > 4, 1);
  } else {
    /* std::sort + rank, then xWriteTruncBinCode(ipred, 61): thresh 5, val 32, b 29 → 5 bits if < 3 else 6 */
    unsigned s[6]; memcpy(s, mpm, sizeof s);
    for (int i = 1; i < 6; i++) { unsigned v = s[i]; int j = i - 1; while (j >= 0 && s[j] > v) { s[j + 1] = s[j]; j--; } s[j + 1] = v; }
    unsigned m = (unsigned) dir;
    for (int i = 5; i >= 0; i--) if (m > s[i]) m--;
    if (m < 3) orc_enc_bins_ep(c, m, 5); else orc_enc_bins_ep(c, m + 3, 6);          /* xWriteTruncBinCode 1553-1561 */
  }
}
/* PU::getCoLocatedIntraLumaMode (CL/UnitTools.cpp:949-960) + getIntraChromaCandModes (840-873) */
static int colocated_luma_mode(const orc_enc *e, area_t a)
{
  const int px = a.x + (a.w >> 1), py = a.y + (a.h >> 1);
  return unit_luma_dir(&e->um[0][(py >> 2) * e->uw + (px >> 2)]);
}
static void chroma_cand_modes(const orc_enc *e, area_t a, int list[8])
{
  list[0] = ORC_PLANAR; list[1] = ORC_VER; list[2] = ORC_HOR; list[3] = ORC_DC; list[4] = 67; list[5] = 68; list[6] = 69; list[7] = ORC_DM_CHROMA;
  const int lm = colocated_luma_mode(e, a);
  for (int i = 0; i < 4; i++) if (lm == list[i]) { list[i] = ORC_VDIA; break; }
}
/* CodingUnit::checkCCLMAllowed (CL/Unit.cpp:375-449) for a chroma-tree CU of a dual-tree I slice, CTU 128: decided by the splits of
 * the 64x64 chroma node (depths 1, 2 of the CU's split series) and of the co-located 64x64 luma node */
static int cclm_allowed(const orc_enc *e, area_t a, uint64_t ss, int depth)
{
  if (!(e->cfg.tools & ORC_TOOL_CCLM)) return 0;
  const int s1 = depth > 1 ? (int) ((ss >> 5) & 31) : SPLIT_NONE, s2 = depth > 2 ? (int) ((ss >> 10) & 31) : SPLIT_NONE;
  int allow = s1 == SPLIT_QT || (s1 == SPLIT_BH && s2 == SPLIT_BV) || s1 == SPLIT_NONE || (s1 == SPLIT_BH && s2 == SPLIT_NONE);
  if (allow) {
    const unit_t *u = &e->um[0][(a.y >> 2) * e->uw + (a.x >> 2)];      /* colLumaCu at the CU's luma position */
    if (u->lw < 6 || u->lh < 6) { const int l1 = u->depth > 1 ? (int) ((u->split_series >> 5) & 31) : SPLIT_NONE; if (l1 != SPLIT_QT) allow = 0; }
  }
  return allow;
}
/* CABACWriter::intra_chroma_pred_mode (1891-1933) + intra_chroma_lmc_mode (1864-1888) */
static void enc_intra_chroma_pred_mode(orc_enc *e, area_t a, int dir, int lm_ok)
{
  orc_cabac *c = &e->cabac;
  if (lm_ok) {
    const int isLM = dir >= 67 && dir <= 69;
    orc_enc_bin(c, (unsigned) isLM, ORC_CTX_CclmModeFlag);
    if (isLM) {
      const int symbol = dir - 67;
      orc_enc_bin(c, symbol == 0 ? 0 : 1, ORC_CTX_IntraChromaPredMode);
      if (symbol > 0) orc_enc_bins_ep(c, (uint32_t) (symbol - 1), 1);
      return;
    }
  }
  const int isDM = dir == ORC_DM_CHROMA;
  orc_enc_bin(c, isDM ? 0 : 1, ORC_CTX_IntraChromaPredMode);
  if (isDM) return;
  int list[8]; chroma_cand_modes(e, a, list);
  int cand = 0;
  for (; cand < 4; cand++) if (list[cand] == dir) break;
  orc_enc_bins_ep(c, (uint32_t) cand, 2);
}

/* ------------------------------------------------------------------------------------------------
 * one transform block: pred (already in e->pred) → resi → T → Q → Q⁻¹ → T⁻¹ → reco → SSE
 * (EL/IntraSearch.cpp:2852-3168, CL/TrQuant.cpp:1127-1235)
 * comp: 0 Y 1 Cb 2 Cr; x,y,w,h in component samples; writes rec_out / lev_out tiles (stride w)
 * ---------------------------------------------------------------------------------------------- */
static uint64_t code_tu_block_ex(orc_enc *e, int comp, int x, int y, int w, int h, int mts_idx, int lfnst_idx, int lfnst_dir, int16_t *rec_out, int16_t *lev_out, int *cbf);
/* lambda the quantiser sees for a component (RDOQ_CHROMA_LAMBDA: EL/EncSlice.cpp:107-149 setLambdas, EL/IntraSearch.cpp:2889 selectLambda) */
/* with JointCbCr on, every chroma block is quantised with 1.3 x that lambda above slice QP 18 (EL/IntraSearch.cpp:2937-2942) */
/* with LMCS chroma residual scaling the lambda is first divided by the square of the scale (2919-2931): cResScale = 2048 / the TU's inverse scale */
static double lmcs_lambda(const orc_enc *e, double l)
{
  if (!e->tu_cadj) return l;
  const double cResScale = (double) (1 << 11) / (double) e->tu_cadj;
  return l / (cResScale * cResScale);
}
static double quant_lambda(const orc_enc *e, int comp)
{
  if (!comp) return e->sl.lambda;
  const double l = lmcs_lambda(e, e->sl.lambda / e->sl.dist_weight[comp - 1]);
  return ((e->cfg.tools & ORC_TOOL_JCCR) && e->sl.qp > 18) ? 1.3 * l : l;
}
/* AreaBuf<Pel>::scaleSignal (CL/Buffer.cpp:501-550): forward = division of the residual by the scale (11 fractional bits), inverse = multiplication */
static void scale_residual(int16_t *r, int n, int scale, int fwd, int bd)
{
  const int mxa = (1 << bd) - 1;
  for (int i = 0; i < n; i++) {
    int v = r[i];
    if (fwd) { const int sg = v >= 0 ? 1 : -1, a = sg * v; v = sg * (((a << 11) + (scale >> 1)) / scale); v = v < -mxa ? -mxa : v > mxa ? mxa : v; }
    else { v = v < -mxa - 1 ? -mxa - 1 : v > mxa ? mxa : v; const int sg = v >= 0 ? 1 : -1, a = sg * v; v = sg * ((a * scale + (1 << 10)) >> 11); v = v < -32768 ? -32768 : v > 32767 ? 32767 : v; }
    r[i] = (int16_t) v;
  }
}
/* Reshape::calculateChromaAdjVpduNei (CL/Reshape.cpp:153-250): the scale of every chroma TU inside a 64x64 luma area comes from the average of the reconstructed luma
 * samples left of and above the luma CU that holds the area's top-left sample (64 each, clamped at the picture edge), looked up in the model's piece-wise scale table.
 * a: the chroma TU in luma coordinates.  Returns 0 when the slice does not scale chroma residuals. */
static int chroma_adj(const orc_enc *e, area_t a)
{
  if (!e->lmcs_on || !e->sl.lmcs_chroma_adj) return 0;
  const int vx = a.x / 64 * 64, vy = a.y / 64 * 64;
  const unit_t *tl = &e->um[0][(vy >> 2) * e->uw + (vx >> 2)];
  const int x = tl->x, y = tl->y, st = e->stride[0], bd = e->cfg.bit_depth;
  const unit_t *cuA = get_cu(e, 0, x, y - 1), *cuL = get_cu(e, 0, x - 1, y);
  int sum = 0, n = 0;
  if (cuL) for (int i = 0; i < 64; i++) { const int k = (y + i) >= e->hl ? (e->hl - y - 1) : i; sum += e->rec[0][(y + k) * st + x - 1]; n++; }
  if (cuA) for (int i = 0; i < 64; i++) { const int k = (x + i) >= e->wl ? (e->wl - x - 1) : i; sum += e->rec[0][(y - 1) * st + x + k]; n++; }
  int v;
  if (n == 64) v = (sum + 32) >> 6; else if (n == 128) v = (sum + 64) >> 7; else v = 1 << (bd - 1);
  const int mx = (1 << bd) - 1; v = v < 0 ? 0 : v > mx ? mx : v;
  int idx = e->sl.lmcs_min_bin;
  for (; idx <= e->sl.lmcs_max_bin; idx++) if (v < e->lmcs_pivot[idx + 1]) break;
  if (idx > 15) idx = 15;
  return e->lmcs_cadj[idx];
}
static uint64_t code_tu_block(orc_enc *e, int comp, int x, int y, int w, int h, int16_t *rec_out, int16_t *lev_out, int *cbf) { return code_tu_block_ex(e, comp, x, y, w, h, 0, 0, 0, rec_out, lev_out, cbf); }
static uint64_t code_tu_block_mts(orc_enc *e, int comp, int x, int y, int w, int h, int mts_idx, int16_t *rec_out, int16_t *lev_out, int *cbf) { return code_tu_block_ex(e, comp, x, y, w, h, mts_idx, 0, 0, rec_out, lev_out, cbf); }
/* lfnst_idx: cu.lfnstIdx (applies to blocks of at least 4x4, CL/TrQuant.cpp:444); lfnst_dir: the block's final intra mode for the kernel choice
 * (planar for a MIP CU, the co-located luma mode for DM / CCLM chroma, CL/TrQuant.cpp:449-463) */
static uint64_t code_tu_block_ex(orc_enc *e, int comp, int x, int y, int w, int h, int mts_idx, int lfnst_idx, int lfnst_dir, int16_t *rec_out, int16_t *lev_out, int *cbf)
{
  const int st = e->stride[comp], bd = e->cfg.bit_depth;
  const int16_t *org = e->org[comp] + y * st + x;
  const int qp = (comp ? e->sl.qp_c[comp - 1] : e->sl.qp) + 6 * (e->cfg.bit_depth - 8);    /* QpParam: + QpBDOffset (CL/Quant.cpp:68-106) */
  const int lf = (lfnst_idx && w >= 4 && h >= 4) ? lfnst_idx : 0, lmode = lf ? orc_lfnst_mode(lfnst_dir, w, h) : 0;
  for (int j = 0; j < h; j++) for (int i = 0; i < w; i++) e->resi[j * w + i] = (int16_t) (org[j * st + i] - e->pred[j * w + i]);
  if (!comp) memcpy(e->resi_org, e->resi, (size_t) w * h * 2);      /* the MTS pruning works on the prediction residual */
  const int cadj = (comp && w * h > 4) ? e->tu_cadj : 0;             /* 3884-3904 / 3060-3066: chroma residual scaling of blocks of more than 4 samples */
  if (cadj) scale_residual(e->resi, w * h, cadj, 1, bd);
  orc_fwd_2d_mts(e->resi, w, w, h, bd, mts_idx, e->coef);
  if (!comp) { int sa = 0; for (int i = 0; i < w * h; i++) sa += abs(e->coef[i]); e->dct2_sum = sa; }
  if (lf) { orc_lfnst_keep(e->coef, w, h); orc_fwd_lfnst(e->coef, w, h, lmode, lf); }       /* xT's zero-out 855-868, xFwdLfnst 1220-1223 */
  int abs_sum;
  if (e->cfg.tools & ORC_TOOL_DEPQUANT) {
    /* DepQuant::quant (CL/DepQuant.cpp:1755-1781) with the estimator's live contexts (EL/IntraSearch.cpp:2968,3024) and the quantiser's lambda of the
     * component: TrQuant::selectLambda (2889) = lambda / distortion weight for chroma (EL/EncSlice.cpp:107-149) */
    abs_sum = orc_depquant(e->cabac.s0, e->cabac.s1, e->coef, w, h, comp, ORC_CTX_QtCbf[comp] + (comp == 2 ? e->tu_cbf_cb : 0), bd, qp, quant_lambda(e, comp), mts_idx > 1, lf, lev_out);
    if (abs_sum > 0) { orc_dequant_dq(lev_out, w, h, bd, qp, e->coef); orc_inv_lfnst(e->coef, w, h, lmode, lf); orc_inv_2d_mts(e->coef, w, h, bd, mts_idx, e->resi, w); }
  } else {
    abs_sum = lf ? orc_quant_lfnst(e->coef, w, h, bd, qp, lev_out) : orc_quant(e->coef, w, h, bd, qp, lev_out);
    if (abs_sum > 0) { orc_dequant(lev_out, w, h, bd, qp, e->coef); orc_inv_lfnst(e->coef, w, h, lmode, lf); orc_inv_2d_mts(e->coef, w, h, bd, mts_idx, e->resi, w); }
  }
  if (abs_sum <= 0) memset(e->resi, 0, (size_t) w * h * 2);
  else if (cadj) scale_residual(e->resi, w * h, cadj, 0, bd);
  const int mx = (1 << bd) - 1;
  for (int j = 0; j < h; j++) for (int i = 0; i < w; i++) { int v = e->pred[j * w + i] + e->resi[j * w + i]; rec_out[j * w + i] = (int16_t) (v < 0 ? 0 : v > mx ? mx : v); }
  *cbf = abs_sum > 0;
  uint64_t d = orc_sse(org, st, rec_out, w, w, h);
  if (comp) d = (uint64_t) (e->sl.dist_weight[comp - 1] * (double) d);    /* CL/RdCost.cpp:405-408 */
  e->cnt_rd++; e->cnt_rdpix += (uint64_t) w * h;
  return d;
}

static void build_refs(orc_enc *e, int comp, int x, int y, int w, int h, int mrl, int filter)
{
  const int ch = comp ? 1 : 0;
  orc_fill_ref_samples(e->rec[comp], e->stride[comp], comp ? e->wc : e->wl, comp ? e->hc : e->hl, e->avail[ch], e->uw,
                       ch ? 1 : 2, e->cur_tile + 1, x, y, w, h, mrl, e->cfg.bit_depth, e->ref_unf);
  if (filter) orc_filter_ref_samples(e->ref_unf, e->ref_flt, w, h, mrl);
}

/* updateCandList (CL/UnitTools.h:261-306) for a (mode,mrl) list with costs */
typedef struct { int mode, mrl; } minfo;
static void update_cand_list(minfo m, double cost, minfo *list, double *costs, int *size, int fastNum)
{
  int shift = 0;
  const int cur = imin(fastNum, *size);
  while (shift < fastNum && shift < cur && cost < costs[cur - 1 - shift]) shift++;
  if (*size >= fastNum && shift != 0) {
    for (int i = 1; i < shift; i++) { list[cur - i] = list[cur - 1 - i]; costs[cur - i] = costs[cur - 1 - i]; }
    list[cur - shift] = m; costs[cur - shift] = cost;
  } else if (cur < fastNum) {
    const int pos = *size - shift;
    for (int i = *size; i > pos; i--) { list[i] = list[i - 1]; costs[i] = costs[i - 1]; }
    list[pos] = m; costs[pos] = cost; (*size)++;
  }
}

/* luma prediction of one candidate into e->pred (stride w): xIntraCodingTUBlock's initIntraPatternChType + predIntraAng, or for a MIP
 * candidate initIntraMip + predIntraMip from the unfiltered line-0 references (CL/IntraPrediction.cpp:2152-2186) */
static void mip_from_refs(orc_enc *e, int w, int h, int mode)
{
  int16_t top[64], left[64]; const int rs = 2 * w + 1;
  for (int i = 0; i < w; i++) top[i] = e->ref_unf[1 + i];
  for (int j = 0; j < h; j++) left[j] = e->ref_unf[(1 + j) * rs];
  orc_pred_mip(top, left, w, h, mode, e->cfg.bit_depth, e->pred);
}
static void pred_luma_cand(orc_enc *e, int x, int y, int w, int h, int dir, int mrl)
{
  if (mrl & MIP_FLAG) { build_refs(e, 0, x, y, w, h, 0, 0); mip_from_refs(e, w, h, dir); return; }
  orc_ipa ip; orc_init_pred_params(w, h, 1, dir, mrl, &ip);
  build_refs(e, 0, x, y, w, h, mrl, ip.ref_filter);
  orc_pred_intra(e->ref_unf, e->ref_flt, w, h, 1, dir, mrl, e->cfg.bit_depth, e->pred, w);
}
/* IntraSearch::reduceHadCandList (EL/IntraSearch.cpp:4331-4406, JVET_O0925 form, FastMIP 1): at most 3 regular candidates, MIP candidates
 * up to half the list or within thresholdHadCost of the best, always one MIP; for blocks above 8x8 the best of MIP modes {3,4,5} (each in its
 * better orientation) is appended when absent */
static void reduce_had_cand_list(minfo *list, double *costs, int *size, int *numRd, double thr, const double *mipCost, int w, int h)
{
  const int maxPerType = *numRd >> 1;
  minfo tl[80]; double tc[80]; int tn = 0;
  const double minCost = costs[0];
  int keepOneMip = *size > *numRd, numConv = 0, numMip = 0;
  for (int idx = 0; idx < *size - (keepOneMip ? 0 : 1); idx++) {
    int add;
    if (!(list[idx].mrl & MIP_FLAG)) { add = numConv < 3; numConv += add; }
    else { add = numMip < maxPerType || costs[idx] < thr * minCost || keepOneMip; keepOneMip = 0; numMip += add; }
    if (add) { tl[tn] = list[idx]; tc[tn] = costs[idx]; tn++; }
  }
  if (w > 8 && h > 8) {
    const int transpOff = orc_mip_num_modes(w, h) / 2;
    minfo sm[3]; double sc[3]; int sn = 0;
    for (int mode = 3; mode <= 5; mode++) {
      const int cand = mode + (mipCost[mode + transpOff] < mipCost[mode] ? transpOff : 0);
      update_cand_list((minfo) { cand, MIP_FLAG }, mipCost[cand], sm, sc, &sn, 3);
    }
    const int n0 = tn;
    for (int idx = 0; idx < 3; idx++) {
      int incl = 0;
      for (int k = 0; k < n0; k++) incl |= tl[k].mode == sm[idx].mode && tl[k].mrl == sm[idx].mrl;
      if (!incl) { tl[tn] = sm[idx]; tc[tn] = 0; tn++; break; /* fastMip */ }
    }
  }
  memcpy(list, tl, sizeof(minfo) * (size_t) tn); memcpy(costs, tc, sizeof(double) * (size_t) tn);
  *size = tn; *numRd = tn;
}


/* ------------------------------------------------------------------------------------------------
 * ISP (intra sub-partitions)
 * ---------------------------------------------------------------------------------------------- */
/* reference samples of one prediction region (pw x ph at offset ox, oy) of an ISP CU: IntraPrediction::initIntraPatternChTypeISP (CL/IntraPrediction.cpp:1092-1199).
 * base = the CU's own reference samples (e->ref_unf of build_refs, stride 2 w + 1); rec = the CU's reconstruction so far (tile, stride w); ref gets the region's
 * samples with stride w + pw + 1: row 0 = corner + (w + pw) top samples, column 0 = (h + ph) left samples */
static void isp_sub_refs(const orc_enc *e, area_t a, int isp, int ox, int oy, int pw, int ph, const int16_t *rec, int16_t *ref)
{
  const int w = a.w, h = a.h, bs = 2 * w + 1, S = w + pw + 1, topLen = w + pw, leftLen = h + ph;
  const int16_t *base = e->ref_unf;
  if (!ox && !oy) {
    for (int i = 0; i <= topLen; i++) ref[i] = base[i];
    for (int j = 1; j <= leftLen; j++) ref[j * S] = base[j * bs];
  } else if (isp == 1) {                      /* rows: the row above is the previous sub-partition's last reconstructed row, replicated to the right */
    for (int j = 0; j <= leftLen; j++) ref[j * S] = base[(oy + j) * bs];
    for (int i = 1; i <= pw; i++) ref[i] = rec[(oy - 1) * w + (i - 1)];
    for (int i = pw + 1; i <= topLen; i++) ref[i] = rec[(oy - 1) * w + pw - 1];
    const int leftDecomp = a.x > 0;            /* cs.isDecomp of the sample left of the sub-partition: inside the picture it is always coded before this CU, in whichever tile */
    if (!leftDecomp) for (int j = 0; j <= leftLen; j++) ref[j * S] = rec[(oy - 1) * w];
  } else {                                    /* columns */
    for (int i = 0; i <= topLen; i++) ref[i] = base[ox + i];
    for (int j = 1; j <= ph; j++) ref[j * S] = rec[(j - 1) * w + ox - 1];
    for (int j = ph + 1; j <= leftLen; j++) ref[j * S] = rec[(ph - 1) * w + ox - 1];
    const int aboveDecomp = a.y > 0;
    if (!aboveDecomp) for (int i = 0; i <= topLen; i++) ref[i] = rec[ox - 1];
  }
}
typedef struct { uint64_t dist, bits; double cost; int tucbf, ntu, valid, first_cbf; } isp_res;
/* IntraSearch::xIntraCodingLumaISP (EL/IntraSearch.cpp:3171-3280) with xIntraCodingTUBlock per sub-partition: prediction regions of at least 4 columns
 * (JVET_O0106), DST-VII / DCT-II by size, dependent quantisation from the estimator's live contexts (the cbf context of ISP blocks follows the previous
 * sub-partition's cbf; the last cbf is inferred after all-zero ones), early exits against bestCostSoFar.  given != NULL: the levels (CU tile) and cbfs are taken as
 * coded and only the decoder half runs (DecCu::xIntraRecQT of an ISP CU, xReuseCachedResult).  rec / lev: CU tiles, stride w. */
static void isp_code_cu(orc_enc *e, area_t a, int dir, int isp, double bestCostSoFar, const int16_t *given, int given_tucbf, int16_t *rec, int16_t *lev, isp_res *r)
{
  const int x = a.x, y = a.y, w = a.w, h = a.h, bd = e->cfg.bit_depth, hor = isp == 1, mx = (1 << bd) - 1;
  const int psz = isp_split_dim(w, h, hor), tw = hor ? w : psz, th = hor ? psz : h, n = hor ? h / psz : w / psz;
  const int predRegDiff = !hor && ((w == 8 && h > 4) || w == 4);         /* CU::isPredRegDiffFromTB */
  const int qp = e->sl.qp + 6 * (bd - 8), st = e->stride[0];
  static int16_t sub[(64 + 64 + 1) * (64 + 64 + 1)], tlev[64 * 16], tres[64 * 16];
  build_refs(e, 0, x, y, w, h, 0, 0);
  memset(r, 0, sizeof *r);
  double cost = 0; int early = 0, tucbf = 0, ntu = 0;
  for (int k = 0; k < n; k++) {
    const int ox = hor ? 0 : k * tw, oy = hor ? k * th : 0;
    if (!predRegDiff || (ox & 3) == 0) {
      const int pw = predRegDiff ? (tw > 4 ? tw : 4) : tw;
      isp_sub_refs(e, a, isp, ox, oy, pw, th, rec, sub);
      orc_pred_intra_isp(sub, w + pw + 1, w, h, pw, th, dir, bd, e->pred + oy * w + ox, w);
    }
    const int16_t *org = e->org[0] + (y + oy) * st + x + ox;
    const int lastInferred = k == n - 1 && !tucbf, prevCbf = k ? (tucbf >> (k - 1)) & 1 : 0;
    int abs_sum;
    if (!given) {
      for (int j = 0; j < th; j++) for (int i = 0; i < tw; i++) tres[j * tw + i] = (int16_t) (org[j * st + i] - e->pred[(oy + j) * w + ox + i]);
      orc_fwd_isp(tres, tw, tw, th, bd, e->coef);
      abs_sum = orc_depquant(e->cabac.s0, e->cabac.s1, e->coef, tw, th, 0, lastInferred ? -1 : ORC_CTX_QtCbf[0] + 2 + prevCbf, bd, qp, e->sl.lambda, 0, 0, tlev);
      e->cnt_rd++; e->cnt_rdpix += (uint64_t) tw * th;
      if (k == n - 1 && !tucbf && abs_sum <= 0) { r->ntu = n; r->cost = ORC_MAX_DOUBLE; return; }       /* 2990-2996: ISP needs one coded sub-partition */
    } else {
      for (int j = 0; j < th; j++) for (int i = 0; i < tw; i++) tlev[j * tw + i] = given[(oy + j) * w + ox + i];
      abs_sum = (given_tucbf >> k) & 1;
    }
    if (abs_sum > 0) { orc_dequant_dq(tlev, tw, th, bd, qp, e->coef); orc_inv_isp(e->coef, tw, th, bd, tres, tw); tucbf |= 1 << k; }
    else memset(tres, 0, (size_t) tw * th * 2);
    for (int j = 0; j < th; j++) for (int i = 0; i < tw; i++) {
      const int v = e->pred[(oy + j) * w + ox + i] + tres[j * tw + i];
      rec[(oy + j) * w + ox + i] = (int16_t) (v < 0 ? 0 : v > mx ? mx : v);
      lev[(oy + j) * w + ox + i] = tlev[j * tw + i];
    }
    const uint64_t d = orc_sse(org, st, rec + oy * w + ox, w, tw, th);
    ntu = k + 1;
    if (given) { r->dist += d; continue; }
    uint64_t fb = 0;
    if (rd_cost(e, r->bits, r->dist + d) > bestCostSoFar) early = 1;       /* 3206-3210: the rate is not even computed */
    else {
      e->cabac.bits = 0;                       /* xGetIntraFracBitsQT: the CU header with the first sub-partition, cbf unless inferred, coefficients */
      if (k == 0) enc_intra_luma_pred_mode_isp(e, x, y, w, h, dir, 0, isp);
      if (!lastInferred) orc_enc_bin(&e->cabac, abs_sum > 0, ORC_CTX_QtCbf[0] + 2 + prevCbf);
      if (abs_sum > 0) orc_residual_coding_tu(&e->cabac, tlev, tw, th, 0, 0, 0, 0);
      fb = e->cabac.bits;
    }
    cost += rd_cost(e, fb, d); r->dist += d; r->bits += fb;
    if (k + 1 < n) {
      if (cost > bestCostSoFar) { early = 1; break; }
      const double thr = n == 2 ? 0.95 : k + 1 == 1 ? 0.83 : 0.91;
      if (cost > bestCostSoFar * thr) { early = 1; break; }
    }
  }
  r->ntu = ntu; r->tucbf = tucbf; r->first_cbf = tucbf & 1;
  if (given) { r->valid = 1; return; }
  if (early) { r->cost = ORC_MAX_DOUBLE; return; }
  r->cost = rd_cost(e, r->bits, r->dist);
  if (r->cost < bestCostSoFar) { r->valid = 1; r->first_cbf = tucbf != 0; }          /* 3257-3268: cbf at depth 0 of every TU = any sub-partition coded */
  else r->cost = ORC_MAX_DOUBLE;
}
/* ISPTestedModesInfo + the candidate lists of the ISP tests (EL/IntraSearch.h:211-320) */
typedef struct {
  int num_total[2], n_tested[2], cand_idx[2], stop[2], tested[2][40], best_mode_so_far, best_split_so_far, n_orig;
  double best_cost[2];
  uint8_t has[ORC_NUM_LUMA_MODE][2]; int nparts[ORC_NUM_LUMA_MODE][2]; double rdcost[ORC_NUM_LUMA_MODE][2];
  int list[40], nlist;                       /* m_ispCandListHor == m_ispCandListVer up to the split type */
  int reg_n, reg_mode[40]; double reg_cost[40];      /* m_regIntraRDListWithCosts */
  int had_n, had_mode[40];                   /* the regular SATD-stage list saved for ISP (m_ispCandListHor before the sort) */
} isp_state;
static void isp_set_mode_results(isp_state *s, int isp, int mode, int nCompleted, double rdCost, double currentBest)
{
  const int st = isp - 1, maxParts = s->num_total[st];
  s->nparts[mode][st] = nCompleted; s->rdcost[mode][st] = nCompleted == maxParts ? rdCost : ORC_MAX_DOUBLE;
  s->tested[st][s->n_tested[st]++] = mode; s->has[mode][st] = 1;
  if (nCompleted == maxParts && rdCost < s->best_cost[st]) s->best_cost[st] = rdCost;
  if (nCompleted == maxParts && rdCost < currentBest) { s->best_mode_so_far = mode; s->best_split_so_far = isp; }
}
static int isp_num_parts(const isp_state *s, int isp, int mode) { return s->has[mode][isp - 1] ? s->nparts[mode][isp - 1] : -1; }
/* xSortISPCandList (4614-4717) */
static void isp_sort_cand_list(isp_state *s, double bestCostSoFar, double bestNonISPCost)
{
  if (bestNonISPCost > bestCostSoFar * 1.4) { s->stop[0] = s->stop[1] = 1; return; }         /* ISPFast 1 */
  uint8_t in[ORC_NUM_LUMA_MODE]; memset(in, 0, sizeof in);
  /* std::sort of at most 16 entries by cost: libstdc++'s insertion sort, which keeps equal costs in their order */
  for (int i = 1; i < s->reg_n; i++) {
    const int m = s->reg_mode[i]; const double c = s->reg_cost[i]; int j = i - 1;
    while (j >= 0 && c < s->reg_cost[j]) { s->reg_mode[j + 1] = s->reg_mode[j]; s->reg_cost[j + 1] = s->reg_cost[j]; j--; }
    s->reg_mode[j + 1] = m; s->reg_cost[j + 1] = c;
  }
  int bestAngle = -1;
  for (int i = 0; i < s->reg_n; i++) if (s->reg_mode[i] > ORC_DC) { bestAngle = s->reg_mode[i]; break; }
  s->nlist = 0;
  s->list[s->nlist++] = ORC_PLANAR; in[ORC_PLANAR] = 1;
  if (bestAngle != -1) { s->list[s->nlist++] = bestAngle; in[bestAngle] = 1; }
  int dc = 0;
  for (int i = 0; i < s->reg_n; i++) {
    const int m = s->reg_mode[i];
    if (m != ORC_PLANAR && m != bestAngle) { if (m > ORC_DC) { s->list[s->nlist++] = m; in[m] = 1; } else if (m == ORC_DC) dc = 1; }
  }
  if (dc) { s->list[s->nlist++] = ORC_DC; in[ORC_DC] = 1; }
  s->n_orig = s->nlist;
  for (int k = 0, added = 0; k < s->had_n && added < 3; k++) if (!in[s->had_mode[k]]) { s->list[s->nlist++] = s->had_mode[k]; added++; }
}
static void isp_find_nearby(const isp_state *s, int mode, int isp, int window, int *left, int *right)          /* xFindAlreadyTestedNearbyIntraModes 4591-4612 */
{
  *left = *right = -1;
  for (int k = 1; k <= window; k++) {
    const int off = mode - 2 - k;
    const int lm = off < 0 ? ORC_NUM_LUMA_MODE + off : mode - k;
    const int rm = mode > ORC_DC ? ((mode - 2 + k) % 65) + 2 : ORC_PLANAR;
    const int lf = lm != mode ? s->has[lm][isp - 1] : 0, rf = rm != mode ? s->has[rm][isp - 1] : 0;
    if (lf || rf) { *left = lf ? lm : -1; *right = rf ? rm : -1; break; }
  }
}
/* xGetNextISPMode (4467-4589): prev_isp = ispMod of the previous entry of the RD list.  Returns 1 with the next candidate, 0 when the slot stays unused */
static int isp_next_mode(isp_state *s, int prev_isp, int w, int h, int *mode, int *isp_out)
{
  int nxt;
  if (!s->stop[0] && !s->stop[1]) nxt = prev_isp == 1 ? 2 : 1;
  else if (!s->stop[0]) nxt = 1;
  else if (!s->stop[1]) nxt = 2;
  else return 0;
  const int st = nxt - 1, maxParts = s->num_total[st];
  if (s->n_tested[st] >= 2) {
    int mode1 = s->tested[st][0]; mode1 = mode1 == ORC_DC ? -1 : mode1;
    const int n1 = mode1 != -1 ? isp_num_parts(s, nxt, mode1) : -1;
    int mode2 = s->tested[st][1]; mode2 = mode2 == ORC_DC ? -1 : mode2;
    const int n2 = mode2 != -1 ? isp_num_parts(s, nxt, mode2) : -1;
    if (n1 != -1 && n2 != -1 && n1 < maxParts && n2 < maxParts) { s->stop[st] = 1; return 0; }
    const int other = nxt == 1 ? 2 : 1;
    const int nOther = mode2 != -1 ? isp_num_parts(s, other, mode2) : -1;
    int stopThis = 0;
    if (nOther != -1 && n2 != -1) {
      if (nOther > n2) stopThis = 1;
      else if (nOther == n2 && nOther == maxParts) {
        const double cThis = s->has[mode2][st] && s->nparts[mode2][st] == maxParts ? s->rdcost[mode2][st] : -1;
        const double cOther = s->has[mode2][other - 1] && s->nparts[mode2][other - 1] == maxParts ? s->rdcost[mode2][other - 1] : -1;
        if (cThis == ORC_MAX_DOUBLE || cOther < cThis * 1.3) stopThis = 1;
      }
    }
    if (stopThis) { s->stop[st] = 1; return 0; }
  }
  if (s->cand_idx[st] < s->nlist) {
    const int cand = s->list[s->cand_idx[st]];
    s->cand_idx[st]++;
    if (s->cand_idx[st] > s->n_orig) { if (s->best_split_so_far != nxt || s->best_mode_so_far == ORC_PLANAR) return 0; }      /* extra modes only while ISP is winning */
    int test = 1;
    if (cand >= ORC_DC && maxParts > 2 && s->n_tested[st] >= 2) {
      const int window = cand > ORC_DC ? 5 : 1, numSamples = w << ilog2(h), limit = numSamples >= 256 ? maxParts - 1 : 2;
      int lm, rm; isp_find_nearby(s, cand, nxt, window, &lm, &rm);
      const int nl = lm != -1 ? isp_num_parts(s, nxt, lm) : -1, nr = rm != -1 ? isp_num_parts(s, nxt, rm) : -1;
      const int nref = nl > nr ? nl : nr;
      if (nref > 0) test = nref > limit;
    }
    if (test) { *mode = cand; *isp_out = nxt; return 1; }
  }
  return 0;
}

/* ------------------------------------------------------------------------------------------------
 * estIntraPredLumaQT (EL/IntraSearch.cpp:289-1380), P0 subset.  Leaves the winner's reco/levels in
 * e->best_rec[0]/best_lev[0] (stride w).  Returns dist; *dir,*mrl,*cbf the winner.
 * ---------------------------------------------------------------------------------------------- */
/* TU::isMTSAllowed (CL/UnitTools.cpp:4549-4565) for an intra luma TU without ISP / BDPCM: explicit intra MTS on, both sides <= 32 */
static int mts_allowed(const orc_enc *e, int w, int h) { return (e->cfg.tools & ORC_TOOL_MTS) && w <= 32 && h <= 32; }
/* TU::isTSAllowed (CL/UnitTools.cpp:4524-4546) for a luma TU of an intra CU without ISP / BDPCM: SPS transform skip on, both sides <= 1 << TransformSkipLog2MaxSize (5 in the cfg) */
static int ts_allowed(const orc_enc *e, int w, int h) { return (e->cfg.tools & ORC_TOOL_TS) && w <= 32 && h <= 32; }
/* xIntraCodingTUBlock of a luma block with tu.mtsIdx = MTS_SKIP: prediction in e->pred; xTransformSkip, RDOQ-TS from the estimator's live contexts, Quant::dequant,
 * xITransformSkip, reconstruction, SSE */
static uint64_t code_tu_block_ts(orc_enc *e, int x, int y, int w, int h, int16_t *rec_out, int16_t *lev_out, int *cbf)
{
  const int st = e->stride[0], bd = e->cfg.bit_depth, qp = e->sl.qp + 6 * (bd - 8), mx = (1 << bd) - 1;
  const int16_t *org = e->org[0] + y * st + x;
  for (int j = 0; j < h; j++) for (int i = 0; i < w; i++) e->resi[j * w + i] = (int16_t) (org[j * st + i] - e->pred[j * w + i]);
  orc_ts_fwd(e->resi, w, w, h, bd, e->coef);
  const int abs_sum = orc_rdoq_ts(e->cabac.s0, e->cabac.s1, e->coef, w, h, bd, qp, e->sl.lambda, lev_out);
  if (abs_sum > 0) { orc_dequant_ts(lev_out, w, h, bd, qp, e->coef); orc_ts_inv(e->coef, w, h, bd, e->resi, w); }
  else memset(e->resi, 0, (size_t) w * h * 2);
  for (int j = 0; j < h; j++) for (int i = 0; i < w; i++) { int v = e->pred[j * w + i] + e->resi[j * w + i]; rec_out[j * w + i] = (int16_t) (v < 0 ? 0 : v > mx ? mx : v); }
  *cbf = abs_sum > 0;
  e->cnt_rd++; e->cnt_rdpix += (uint64_t) w * h;
  return orc_sse(org, st, rec_out, w, w, h);
}
/* What IntraSearch keeps between the passes of one xCheckRDCostIntra call when LFNST is on (m_uiSavedRdModeListLFNST ... 534-566, m_savedRdModeList /
 * m_modeCostStore / m_bestModeCostStore 884-916, 1262-1290) and the parameters of a pass (cu.lfnstIdx, cu.mtsFlag, the MTS index range = the
 * transform group, moreProbMTSIdxFirst) */
typedef struct {
  int lfnst, mts_flag, tr_grp;                       /* pass: cu.lfnstIdx, cu.mtsFlag, mtsFirstCheckId = mtsLastCheckId = trGrpIdx (moreProbMTSIdxFirst = trGrpIdx > 0) */
  int lfnst_num, lfnst_size; minfo lfnst_list[80]; double lfnst_cost[80];
  int rd_num[3]; minfo rd_list[3][80]; double mode_cost[3][80], best_cost[3]; int best_valid[3];
} luma_passes;
/* bestCostSoFar: what xCheckRDCostIntra hands over (the node's best cost without split flags, capped by the child budget); *noIspCost: ComprCUCtx's
 * bestCostMtsFirstPassNoIsp; *out_isp / *out_tucbf: cu.ispMode of the winner and the cbfs of its sub-partitions */
static uint64_t est_intra_pred_luma(orc_enc *e, area_t a, luma_passes *ps, int *valid, int *out_dir, int *out_mrl, int *out_cbf, int *out_mts,
                                    double bestCostSoFar, double *noIspCost, int *out_isp, int *out_tucbf)
{
  const int x = a.x, y = a.y, w = a.w, h = a.h, bd = e->cfg.bit_depth;
  orc_cabac ctxStart; orc_ctx_copy(&ctxStart, &e->cabac);
  const int16_t *org = e->org[0] + y * e->stride[0] + x;
  const int lfnstOn = (e->cfg.tools & ORC_TOOL_LFNST) != 0, lfnstIdx = lfnstOn ? ps->lfnst : 0, mtsFlag = lfnstOn ? ps->mts_flag : 0;
  const int useMip = (e->cfg.tools & ORC_TOOL_MIP) != 0;
  const int numTab = ORC_MODE_NUM_FAST_2D[(ilog2(w) - 2) * 6 + (ilog2(h) - 2)];
  int numRd = numTab;
  minfo rdList[80]; double rdCost[80]; int rdSize = 0;
  minfo hadList[8]; double hadCost[8]; int hadSize = 0;
  /* 330-341: 0 no MTS for this CU, 1 the DCT-II pass of a CU that also gets MTS passes, 2 an MTS pass (LFNST on: MTS is a CU-level pass) */
  int mtsUsage = 0;
  if (w <= 32 && h <= 32 && (e->cfg.tools & ORC_TOOL_MTS)) mtsUsage = (lfnstOn && mtsFlag == 1) ? 2 : 1;
  const int lfnstLoad = lfnstOn && lfnstIdx != 0; int lfnstSave = lfnstOn && lfnstIdx == 0 && mtsFlag == 0;       /* 314-317 */
  /* 404-418 (JVET_O0925): MIP is searched wherever it can be signalled; with LFNST only for blocks of at least 16x16; 469-477: the regular list is
   * kept longer while MIP candidates compete for it */
  const int lfnstWithMip = w >= 16 && h >= 16;
  const int testMip = mip_signalled(e, w, h) && (lfnstIdx == 0 || lfnstWithMip);
  uint8_t checked[ORC_NUM_LUMA_MODE]; memset(checked, 0, sizeof checked);
  const int firstLine = (y & 127) == 0;
  const int numRefPasses = (firstLine || !(e->cfg.tools & ORC_TOOL_MRL)) ? 1 : 3;
  int idxOf[80];                                     /* rdModeIdxList 1097-1122: place of a stage-B candidate in the list before the MIP re-ordering */
  /* 355-384: ISP is tested in the pass without LFNST and MTS */
  const int testISP = lfnstOn && !mtsFlag && !lfnstIdx && can_use_isp(e, w, h);
  static isp_state isp;
  if (testISP) {
    memset(&isp, 0, sizeof isp);
    isp.best_cost[0] = isp.best_cost[1] = ORC_MAX_DOUBLE; isp.best_mode_so_far = -1; isp.n_orig = -1;
    isp.num_total[0] = h >> ilog2(isp_split_dim(w, h, 1)); isp.num_total[1] = w >> ilog2(isp_split_dim(w, h, 0));
  }
  if (mtsUsage != 2) {
  if (testMip) numRd += imax(numRd, ilog2(imin(w, h)) - 1);
  const int numHad = testMip ? 6 : 3;

  /* stage A: SATD pre-selection (483-682).  initIntraPatternChType(cu, Y, forceRefFilter=true) */
  build_refs(e, 0, x, y, w, h, 0, 1);
#define SATD_COST(mode_, mrl_, cost_out, had_out) do { \
    if ((mrl_) & MIP_FLAG) mip_from_refs(e, w, h, (mode_)); else orc_pred_intra(e->ref_unf, e->ref_flt, w, h, 1, (mode_), (mrl_), bd, e->pred, w); \
    const uint64_t sad_ = orc_sad(org, e->stride[0], e->pred, w, w, h), satd_ = orc_satd(org, e->stride[0], e->pred, w, w, h); \
    const uint64_t msh_ = sad_ * 2 < satd_ ? sad_ * 2 : satd_; \
    orc_ctx_copy(&e->cabac, &ctxStart); e->cabac.bits = 0; \
    enc_intra_luma_pred_mode(e, x, y, w, h, (mode_), (mrl_)); \
    (cost_out) = (double) msh_ + (double) e->cabac.bits * e->sqrt_lambda_fp; (had_out) = (double) msh_; e->cnt_satd++; } while (0)
#define LFNST_SAVE(n_) do { ps->lfnst_num = (n_); ps->lfnst_size = rdSize; memcpy(ps->lfnst_list, rdList, sizeof(minfo) * (size_t) rdSize); memcpy(ps->lfnst_cost, rdCost, sizeof(double) * (size_t) rdSize); lfnstSave = 0; } while (0)
#define LFNST_LOAD() do { numRd = ps->lfnst_num; rdSize = imin(ps->lfnst_size, ps->lfnst_num); memcpy(rdList, ps->lfnst_list, sizeof(minfo) * (size_t) rdSize); memcpy(rdCost, ps->lfnst_cost, sizeof(double) * (size_t) rdSize); } while (0)
  if (!lfnstLoad) {
    for (int mode = 0; mode < ORC_NUM_LUMA_MODE; mode++) {
      if (mode > ORC_DC && (mode & 1)) continue;
      checked[mode] = 1;
      double cost, had; SATD_COST(mode, 0, cost, had);
      update_cand_list((minfo) { mode, 0 }, cost, rdList, rdCost, &rdSize, numRd);
      update_cand_list((minfo) { mode, 0 }, had, hadList, hadCost, &hadSize, numHad);
    }
    if (!useMip && lfnstSave) LFNST_SAVE(numRd);             /* 534-546 */
  }
  if (!useMip && lfnstLoad) LFNST_LOAD();                    /* 547-566 */
  if (!(useMip && lfnstLoad)) {                              /* 568-: the rest of the SATD stage; an LFNST pass with MIP on restores the list instead (763-775) */
  {
    minfo parent[80]; memcpy(parent, rdList, sizeof(minfo) * (size_t) numRd);
    for (int i = 0; i < numRd; i++) {
      const int pm = parent[i].mode;
      if (pm > (ORC_DC + 1) && pm < (ORC_NUM_LUMA_MODE - 1))
        for (int s = -1; s <= 1; s += 2) {
          const int mode = pm + s;
          if (!checked[mode]) {
            double cost, had; SATD_COST(mode, 0, cost, had);
            update_cand_list((minfo) { mode, 0 }, cost, rdList, rdCost, &rdSize, numRd);
            update_cand_list((minfo) { mode, 0 }, had, hadList, hadCost, &hadSize, numHad);
            checked[mode] = 1;
          }
        }
    }
  }
  if (testISP) { isp.had_n = rdSize; for (int i = 0; i < rdSize; i++) isp.had_mode[i] = rdList[i].mode; }      /* 624-632: the regular list, before the MRL candidates */
  {
    unsigned mpm[6]; get_mpms(e, x, y, w, h, mpm);
    static const int MRL_IDX[3] = { 0, 1, 3 };
    for (int r = 1; r < numRefPasses; r++) {
      const int mrl = MRL_IDX[r];
      build_refs(e, 0, x, y, w, h, mrl, 1);
      for (int k = 1; k < 6; k++) {
        double cost, had; SATD_COST((int) mpm[k], mrl, cost, had);
        update_cand_list((minfo) { (int) mpm[k], mrl }, cost, rdList, rdCost, &rdSize, numRd);
        update_cand_list((minfo) { (int) mpm[k], mrl }, had, hadList, hadCost, &hadSize, numHad);
      }
    }
  }
  if (lfnstSave && testMip && !lfnstWithMip) {               /* 681-698: the LFNST passes of this CU run without MIP: keep the regular list for them */
    LFNST_SAVE(numTab); ps->lfnst_size = imin(ps->lfnst_size, numTab);
  }
  if (testMip) {
    /* 703-748: every MIP mode by SATD into the same list (one entry longer), then reduceHadCandList */
    double mipCost[35];
    build_refs(e, 0, x, y, w, h, 0, 0);
    const int nMip = orc_mip_num_modes(w, h);
    for (int mode = 0; mode < nMip; mode++) {
      double cost, had; SATD_COST(mode, MIP_FLAG, cost, had);
      mipCost[mode] = cost;
      update_cand_list((minfo) { mode, MIP_FLAG }, cost, rdList, rdCost, &rdSize, numRd + 1);
      update_cand_list((minfo) { mode, MIP_FLAG }, 0.8 * had, hadList, hadCost, &hadSize, numHad);
    }
    reduce_had_cand_list(rdList, rdCost, &rdSize, &numRd, 1.0 + 1.4 / sqrt((double) (w * h)), mipCost, w, h);
  }
  if (useMip && lfnstSave) LFNST_SAVE(numRd);                /* 750-761 */
  } else LFNST_LOAD();
#undef SATD_COST
  {
    /* EL/IntraSearch.cpp:784-802: numCand = PU::getIntraMPMs(...) (1 if left==above dir else 2) */
    unsigned mpm[6];
    int L, A;
    const int numCand = mpm_neighbours(e, x, y, w, h, &L, &A);
    orc_get_mpms(L, A, mpm);
    for (int j = 0; j < numCand; j++) {
      int incl = 0;
      for (int i = 0; i < numRd; i++) incl |= (rdList[i].mode == (int) mpm[j] && rdList[i].mrl == 0);
      if (!incl) { rdList[numRd] = (minfo) { (int) mpm[j], 0 }; rdCost[numRd] = 0; numRd++; }
    }
    if (testISP) for (int j = 0; j < numCand; j++) {      /* 803-820: the MPMs join the list saved for ISP as well */
      int incl = 0;
      for (int i = 0; i < isp.had_n; i++) incl |= isp.had_mode[i] == (int) mpm[j];
      if (!incl) isp.had_mode[isp.had_n++] = (int) mpm[j];
    }
  }
  if (lfnstOn && mtsUsage == 1) { ps->rd_num[lfnstIdx] = numRd; memcpy(ps->rd_list[lfnstIdx], rdList, sizeof(minfo) * (size_t) numRd); }      /* 884-889 */
  } else {
    /* 891-916: an MTS pass re-tests the DCT-II pass's candidates whose cost stayed within the threshold of its best one (FastLFNST 1) */
    numRd = 0;
    if (ps->best_valid[lfnstIdx]) {
      const double thr = 1.0 + ((lfnstIdx > 0) ? 0.1 : 1.0) * (1.4 / sqrt((double) (w * h)));
      for (int i = 0; i < ps->rd_num[lfnstIdx]; i++) if (ps->mode_cost[lfnstIdx][i] <= thr * ps->best_cost[lfnstIdx]) rdList[numRd++] = ps->rd_list[lfnstIdx][i];
    } else { numRd = ps->rd_num[lfnstIdx]; memcpy(rdList, ps->rd_list[lfnstIdx], sizeof(minfo) * (size_t) numRd); }
  }

  for (int i = 0; i < 80; i++) idxOf[i] = i;
  if (testMip) {
    /* 1097-1122: regular candidates first, MIP candidates after them, each group in list order */
    minfo t[80]; int n = 0;
    for (int i = 0; i < numRd; i++) if (!(rdList[i].mrl & MIP_FLAG)) { idxOf[n] = i; t[n++] = rdList[i]; }
    for (int i = 0; i < numRd; i++) if (rdList[i].mrl & MIP_FLAG) { idxOf[n] = i; t[n++] = rdList[i]; }
    memcpy(rdList, t, sizeof(minfo) * (size_t) numRd);
  } else {
    /* 1123-1141: MIP candidates leave the list */
    int n = 0;
    for (int i = 0; i < numRd; i++) if (!(rdList[i].mrl & MIP_FLAG)) rdList[n++] = rdList[i];
    numRd = n;
  }

  /* stage B: full RD (1158-1358) */
  double bestCost = ORC_MAX_DOUBLE; uint64_t bestDist = 0; int bestDir = 0, bestMrl = 0, bestCbf = 0, bestMts = 0, any = 0, bestIsp = 0, bestTuCbf = 0;
  double bestCurrentCost = bestCostSoFar;
  if (!mtsFlag) *noIspCost = ORC_MAX_DOUBLE;                    /* 1150-1153 */
  for (int m = 0; m < numRd; m++) {
    const int dir = rdList[m].mode, mrl = rdList[m].mrl;
    orc_ctx_copy(&e->cabac, &ctxStart);
    /* xIntraCodingTUBlock: initIntraPatternChType without forced filter */
    pred_luma_cand(e, x, y, w, h, dir, mrl);
    const int mtsAllowed = mts_allowed(e, w, h);
    double modeCost = ORC_MAX_DOUBLE; uint64_t modeDist = 0; int modeCbf = 0, modeMts = 0;
    if (lfnstOn) {
      /* xRecurIntraCodingLumaQT 3340-3640 with LFNST on (no transform skip): one transform per pass -- DCT-II, or for an MTS pass the pair picked
       * by the transform group: DST7/DST7 in group 0, then (3474-3498, moreProbMTSIdxFirst) the pair the intra mode makes more likely */
      int mtsIdx = 0;
      if (mtsFlag) mtsIdx = ps->tr_grp == 0 ? 2 : ps->tr_grp == 1 ? (dir < 34 ? 4 : 3) : ps->tr_grp == 2 ? (dir < 34 ? 3 : 4) : 5;
      int cbf;
      const uint64_t dist = code_tu_block_ex(e, 0, x, y, w, h, mtsIdx, lfnstIdx, (mrl & MIP_FLAG) ? ORC_PLANAR : dir, e->tmp_rec[0], e->tmp_lev[0], &cbf);
      /* 3349-3367, 3505-3517: in the pass without LFNST and MTS the TU also tries transform skip unless the pruning (sum |TS coefficients|, scaled for odd
       * log2 sizes, against the DCT-II sum) drops it */
      const int tsTest = ts_allowed(e, w, h) && !mtsFlag && !lfnstIdx && (double) orc_ts_sumabs(e->resi_org, w, w, h, bd) <= (double) e->dct2_sum;
      e->cabac.bits = 0;
      enc_intra_luma_pred_mode(e, x, y, w, h, dir, mrl);
      orc_enc_bin(&e->cabac, (unsigned) cbf, ORC_CTX_QtCbf[0] + 0);
      if (cbf) orc_residual_coding_tu(&e->cabac, e->tmp_lev[0], w, h, 0, ts_allowed(e, w, h), mtsAllowed, mtsIdx);
      modeCost = rd_cost(e, e->cabac.bits, dist); modeDist = dist; modeCbf = cbf; modeMts = mtsIdx;
      if (tsTest) {                                      /* from the start contexts (3441-3444); an empty TS block is forbidden (3567-3571) */
        orc_ctx_copy(&e->cabac, &ctxStart);
        int cbfT;
        const uint64_t distT = code_tu_block_ts(e, x, y, w, h, e->tmp_rec[1], e->tmp_lev[1], &cbfT);
        if (cbfT) {
          e->cabac.bits = 0;
          enc_intra_luma_pred_mode(e, x, y, w, h, dir, mrl);
          orc_enc_bin(&e->cabac, 1u, ORC_CTX_QtCbf[0] + 0);
          orc_residual_coding_tu(&e->cabac, e->tmp_lev[1], w, h, 0, 1, mtsAllowed, 1);
          const double costT = rd_cost(e, e->cabac.bits, distT);
          if (costT < modeCost) {
            modeCost = costT; modeDist = distT; modeCbf = 1; modeMts = 1;
            memcpy(e->tmp_rec[0], e->tmp_rec[1], (size_t) w * h * 2); memcpy(e->tmp_lev[0], e->tmp_lev[1], (size_t) w * h * 2);
          }
        }
      }
      if (mtsUsage == 1) ps->mode_cost[lfnstIdx][idxOf[m]] = modeCost;                       /* 1262-1265 */
    } else {
    /* xRecurIntraCodingLumaQT 3340-3640 without LFNST / transform skip: transform candidates {DCT2} or, where TU::isMTSAllowed,
     * {DCT2, 2, 3, 4, 5} pruned by TrQuant::transformNxN (1049-1124) on the first (DCT2) pass; every further candidate starts from the
     * start contexts; the loop ends after DCT2 when its cbf is 0; an MTS candidate with cbf 0 is forbidden (cost MAX) */
    int test[5] = { 1, 0, 0, 0, 0 };
    static const int idx_of[5] = { 0, 2, 3, 4, 5 };
    int cbfDCT2 = 1;
    for (int k = 0; k < (mtsAllowed ? 5 : 1); k++) {
      if (!cbfDCT2) break;
      if (!test[k]) continue;
      if (k) orc_ctx_copy(&e->cabac, &ctxStart);
      int cbf;
      const uint64_t dist = code_tu_block_mts(e, 0, x, y, w, h, idx_of[k], e->tmp_rec[1], e->tmp_lev[1], &cbf);
      if (k == 0 && mtsAllowed) orc_mts_prune(e->resi_org, w, w, h, bd, 3 /* MTSIntraMaxCand, BIN/encoder_intra.cfg */, test);
      double cost = ORC_MAX_DOUBLE;
      if (!(k && !cbf)) {
        /* xGetIntraFracBitsQT(luma): header + cbf + [mts_idx +] coefficients */
        e->cabac.bits = 0;
        enc_intra_luma_pred_mode(e, x, y, w, h, dir, mrl);
        orc_enc_bin(&e->cabac, (unsigned) cbf, ORC_CTX_QtCbf[0] + 0);
        if (cbf) orc_residual_coding_mts(&e->cabac, e->tmp_lev[1], w, h, 0, mtsAllowed ? idx_of[k] : -1);
        cost = rd_cost(e, e->cabac.bits, dist);
      }
      if (cost < modeCost) {
        modeCost = cost; modeDist = dist; modeCbf = cbf; modeMts = idx_of[k];
        if (k == 0) cbfDCT2 = cbf;
        memcpy(e->tmp_rec[0], e->tmp_rec[1], (size_t) w * h * 2); memcpy(e->tmp_lev[0], e->tmp_lev[1], (size_t) w * h * 2);
      }
    }
    }
    any = 1;
    if (testISP && !mrl) { isp.reg_mode[isp.reg_n] = dir; isp.reg_cost[isp.reg_n] = modeCost; isp.reg_n++; }      /* 1263-1268: regular, line 0, not MIP (mrl carries the MIP flag) */
    if (modeCost < bestCost) {
      bestCost = modeCost; bestDist = modeDist; bestDir = dir; bestMrl = mrl; bestCbf = modeCbf; bestMts = modeMts;
      memcpy(e->best_rec[0], e->tmp_rec[0], (size_t) w * h * 2); memcpy(e->best_lev[0], e->tmp_lev[0], (size_t) w * h * 2);
      if (lfnstOn && mtsUsage == 1) { ps->best_cost[lfnstIdx] = bestCost; ps->best_valid[lfnstIdx] = 1; }     /* 1283-1287 */
      if (bestCost < bestCurrentCost) bestCurrentCost = bestCost;                                             /* 1319-1322 */
      if (!mtsFlag) *noIspCost = bestCost;                                                                    /* 1323-1326 */
    }
  }
  if (testISP) {
    /* 1032-1039, 1181-1192: sixteen reserved places behind the regular and MIP candidates; each asks xGetNextISPMode for the next ISP candidate */
    int prevIsp = 0;
    for (int slot = 0; slot < 16; slot++) {
      if (slot == 0) isp_sort_cand_list(&isp, bestCurrentCost, bestCost);
      int cmode = 0, cisp = 0;
      if (!isp_next_mode(&isp, prevIsp, w, h, &cmode, &cisp)) { prevIsp = 3; continue; }
      prevIsp = cisp;
      orc_ctx_copy(&e->cabac, &ctxStart);
      isp_res r;
      isp_code_cu(e, a, cmode, cisp, bestCurrentCost, 0, 0, e->tmp_rec[1], e->tmp_lev[1], &r);
      isp_set_mode_results(&isp, cisp, cmode, r.ntu, r.first_cbf ? r.cost : ORC_MAX_DOUBLE, bestCost);      /* 1241-1245 */
      if (!r.first_cbf) r.valid = 0;                                                                        /* 1270-1288 */
      if (!r.valid) continue;
      any = 1;
      if (r.cost < bestCost) {
        bestCost = r.cost; bestDist = r.dist; bestDir = cmode; bestMrl = 0; bestCbf = 1; bestMts = 0; bestIsp = cisp; bestTuCbf = r.tucbf;
        memcpy(e->best_rec[0], e->tmp_rec[1], (size_t) w * h * 2); memcpy(e->best_lev[0], e->tmp_lev[1], (size_t) w * h * 2);
        if (bestCost < bestCurrentCost) bestCurrentCost = bestCost;
      }
    }
  }
  orc_ctx_copy(&e->cabac, &ctxStart);
  *valid = any;
  *out_dir = bestDir; *out_mrl = bestMrl; *out_cbf = bestCbf; *out_mts = bestMts; *out_isp = bestIsp; *out_tucbf = bestTuCbf;
  return bestDist;
}
#undef LFNST_SAVE
#undef LFNST_LOAD

/* estIntraPredChromaQT (1382-1686) + xRecurIntraChromaCodingQT (3779-4207), CCLM/JCCR off.
 * a in luma samples.  Winner left in best_rec[0..1]/best_lev[0..1] (Cb,Cr; stride cw). */
#define CCLM_TSTRIDE 130
/* prediction of one chroma component into e->pred (stride cw): regular mode fm, or CCLM mode (67..69) from the down-sampled luma in tmp */
static void pred_chroma_comp(orc_enc *e, int c, int cx, int cy, int cw, int chh, int fm, const int16_t *tmp, const int info[4])
{
  const int bd = e->cfg.bit_depth;
  build_refs(e, c, cx, cy, cw, chh, 0, 0);
  if (fm >= 67 && fm <= 69) {
    int a_, b_, sh_;
    orc_cclm_params(tmp, CCLM_TSTRIDE, e->ref_unf, cw, chh, fm, info, bd, &a_, &b_, &sh_);
    orc_pred_cclm(tmp, CCLM_TSTRIDE, a_, b_, sh_, bd, cw, chh, e->pred, cw);
  } else orc_pred_intra(e->ref_unf, e->ref_flt, cw, chh, 0, fm, 0, bd, e->pred, cw);
}
static void cclm_luma(orc_enc *e, int cx, int cy, int cw, int chh, int mdlm, int16_t *tmp, int info[4])
{
  orc_cclm_luma(e->rec[0], e->stride[0], e->avail[1], e->uw, e->cur_tile + 1, e->wc, e->hc, cx, cy, cw, chh, mdlm, info, tmp, CCLM_TSTRIDE);
}
/* JointCbCr (JVET_O0105 ICT).  g_ictModes (CL/Rom.cpp:613): signed mode of a cbf mask under the slice's sign flag */
static int ict_mode(const orc_enc *e, int mask) { static const int m[4] = { 0, 3, 1, 2 }; return e->jccr_sign ? -m[mask] : m[mask]; }
/* fwdTransformCbCr (CL/TrQuant.cpp:87-137): the joint residual of a mode (into c, n samples) and the distortion of representing both residuals by it */
static int64_t ict_forward(const int16_t *cb, const int16_t *cr, int n, int mode, int16_t *c)
{
  int64_t d = 0;
  for (int i = 0; i < n; i++) {
    const int b = cb[i], r = cr[i]; int v, eb, er;
    switch (mode) {
      case  1: v = (4 * b + 2 * r) / 5; eb = b - v; er = r - (v >> 1); break;
      case -1: v = (4 * b - 2 * r) / 5; eb = b - v; er = r - (-v >> 1); break;
      case  2: v = (b + r) / 2; eb = b - v; er = r - v; break;
      case -2: v = (b - r) / 2; eb = b - v; er = r + v; break;
      case  3: v = (4 * r + 2 * b) / 5; eb = b - (v >> 1); er = r - v; break;
      default: v = (4 * r - 2 * b) / 5; eb = b - (-v >> 1); er = r - v; break;
    }
    c[i] = (int16_t) v;
    d += (int64_t) eb * eb + (int64_t) er * er;
  }
  return d;
}
/* TrQuant::selectICTCandidates for an intra CU (CL/TrQuant.cpp:701-743): up to two cbf masks whose joint representation is closest */
static int ict_candidates(const orc_enc *e, const int16_t *cb, const int16_t *cr, int n, int masks[2], int16_t *tmp)
{
  int64_t d0 = 0, d1 = 0, pd[4];
  for (int i = 0; i < n; i++) { d0 += (int64_t) cb[i] * cb[i]; d1 += (int64_t) cr[i] * cr[i]; }
  for (int m = 1; m < 4; m++) pd[m] = ict_forward(cb, cr, n, ict_mode(e, m), tmp);
  int64_t min1 = d0 < d1 ? d0 : d1, min2 = INT64_MAX; int m1 = 0, m2 = 0;
  for (int m = 1; m < 4; m++) {
    if (pd[m] < min1) { m2 = m1; min2 = min1; m1 = m; min1 = pd[m]; }
    else if (pd[m] < min2) { m2 = m; min2 = pd[m]; }
  }
  int n_ = 0;
  if (m1) masks[n_++] = m1;
  if (m2 && ((min2 < (9 * min1) / 8) || (!m1 && min2 < (3 * min1) / 2))) masks[n_++] = m2;
  return n_;
}
/* QpParam of a joint TU (CL/Quant.cpp:139-176): the coded component's QP, except for the modes +-2 (cbf mask 3), which use the JointCbCr QP: the mapping
 * table's value (one table for all chroma, SameCQPTablesForAllChroma 1) + pps_joint_cbcr_qp_offset (CbCrQpOffset -1, APP/EncAppCfg.cpp:1082) */
static int joint_qp(const orc_enc *e, int mask)
{
  const int bd = e->cfg.bit_depth, off = 6 * (bd - 8);
  if (mask != 3) return e->sl.qp_c[(mask >> 1) ? 0 : 1] + off;
  int qp = e->sl.qp_c[0] - 1; if (qp < -off) qp = -off; if (qp > 63) qp = 63;
  return qp + off;
}
/* test hooks for the decision helpers pinned against CommonLib: updateCandList on a sequence of insertions; per-shape constants of the luma search */
int orc_test_update_cand_list(int n, const int *modes, const double *costs, int fastNum, int *out_modes, double *out_costs)
{
  minfo list[80]; double cl[80]; int size = 0;
  for (int i = 0; i < n; i++) update_cand_list((minfo) { modes[i], 0 }, costs[i], list, cl, &size, fastNum);
  for (int i = 0; i < size; i++) { out_modes[i] = list[i].mode; out_costs[i] = cl[i]; }
  return size;
}
void orc_test_shape_constants(int w, int h, int *out)
{
  out[0] = orc_mip_num_modes(w, h); out[1] = w >= 16 && h >= 16; out[2] = ORC_MODE_NUM_FAST_2D[(ilog2(w) - 2) * 6 + (ilog2(h) - 2)]; out[3] = w <= 32 && h <= 32;
}
/* test entry point (tests/golden decision_helpers2.npz): the CU / TU level predicates of the search as this file states them, for a luma CU of w x h with cu.ispMode = isp
 * and the tool set of e: out[0] can_use_isp, out[1] / out[2] isp_split_dim horizontal / vertical (0 when ISP cannot be used), out[3] the 4-column prediction-region rule
 * of a vertical split (CU::isMinWidthPredEnabledForBlkSize), out[4] transform skip allowed for the luma TU, out[5] explicit MTS allowed (both never for ISP CUs) */
void orc_decision_helpers(const orc_enc *e, int w, int h, int isp, int *out)
{
  const int can = can_use_isp(e, w, h);
  out[0] = can; out[1] = can ? isp_split_dim(w, h, 1) : 0; out[2] = can ? isp_split_dim(w, h, 0) : 0;
  out[3] = (w == 8 && h > 4) || w == 4;
  out[4] = !isp && ts_allowed(e, w, h);
  out[5] = !isp && mts_allowed(e, w, h);
}
/* test entry point (tests/golden isp.npz): the block of one ISP sub-partition (tw x th) through the implicit transform, dependent quantisation with the given cbf
 * context (-1: inferred), dequantisation and the inverse */
int orc_trquant_isp(const uint16_t *s0, const uint16_t *s1, const int16_t *resi, int tw, int th, int bit_depth, int qp, double lambda, int cbf_ctx, int16_t *level, int16_t *resi_out)
{
  static int coef[64 * 64];
  orc_fwd_isp(resi, tw, tw, th, bit_depth, coef);
  const int a = orc_depquant(s0, s1, coef, tw, th, 0, cbf_ctx, bit_depth, qp, lambda, 0, 0, level);
  if (a > 0) { orc_dequant_dq(level, tw, th, bit_depth, qp, coef); orc_inv_isp(coef, tw, th, bit_depth, resi_out, tw); }
  return a;
}
/* test hook: selectICTCandidates + the three joint residuals (joint[m - 1], n samples each) for a residual pair under a sign flag */
int orc_test_ict(int sign, const int16_t *cb, const int16_t *cr, int n, int *masks, int16_t *joint)
{
  orc_enc e; memset(&e, 0, sizeof e); e.jccr_sign = sign;
  for (int m = 1; m <= 3; m++) ict_forward(cb, cr, n, ict_mode(&e, m), joint + (size_t) (m - 1) * n);
  int16_t *tmp = (int16_t *) malloc((size_t) n * 2);
  const int k = ict_candidates(&e, cb, cr, n, masks, tmp);
  free(tmp);
  return k;
}
/* xIntraCodingTUBlock for a joint TU (EL/IntraSearch.cpp:2909-2934, 3050-3091): the joint residual is transformed and quantised as the Cb block (mask 2, 3)
 * or the Cr block (mask 1) at the JointCbCr QP with the loosened lambda; both residuals come back through the inverse ICT.  Predictions in
 * e->pred_c, residuals in e->resi_c.  Returns the distortion of both blocks or UINT64_MAX when the coded block turns out empty (the mask cannot be
 * signalled); levels of the coded block in lev_out, reconstructions in rec_out[0..1]. */
static uint64_t code_joint_block(orc_enc *e, int mask, int cx, int cy, int w, int h, int lfnst_idx, int lfnst_dir, int16_t *const rec_out[2], int16_t *lev_out)
{
  const int bd = e->cfg.bit_depth, n = w * h, mode = ict_mode(e, mask), comp = (mask >> 1) ? 1 : 2;
  const int qp = joint_qp(e, mask);
  const int lf = (lfnst_idx && w >= 4 && h >= 4) ? lfnst_idx : 0, lmode = lf ? orc_lfnst_mode(lfnst_dir, w, h) : 0;
  ict_forward(e->resi_c[0], e->resi_c[1], n, mode, e->resi);
  orc_fwd_2d_mts(e->resi, w, w, h, bd, 0, e->coef);
  if (lf) { orc_lfnst_keep(e->coef, w, h); orc_fwd_lfnst(e->coef, w, h, lmode, lf); }
  /* lambda: the Cb lambda (selectLambda(compID = Cb) 2889), loosened by 0.8 (modes +-1, +-3) or 0.5 (+-2), then the 1.3 of every chroma block */
  const int am = mode < 0 ? -mode : mode;
  const int cadj = n > 4 ? e->tu_cadj : 0;
  double lam = (am == 1 || am == 3 ? 0.8 : 0.5) * (cadj ? lmcs_lambda(e, e->sl.lambda / e->sl.dist_weight[0]) : e->sl.lambda / e->sl.dist_weight[0]);
  if (e->sl.qp > 18) lam = 1.3 * lam;
  /* the other block's cbf is cleared first (3053-3058): Cr's cbf context sees tu.cbf[Cb] = 0 for mask 1 */
  const int abs_sum = orc_depquant(e->cabac.s0, e->cabac.s1, e->coef, w, h, comp, ORC_CTX_QtCbf[comp], bd, qp, lam, 0, lf, lev_out);
  e->cnt_rd++; e->cnt_rdpix += (uint64_t) n;
  if (abs_sum <= 0) return UINT64_MAX;
  orc_dequant_dq(lev_out, w, h, bd, qp, e->coef); orc_inv_lfnst(e->coef, w, h, lmode, lf); orc_inv_2d_mts(e->coef, w, h, bd, 0, e->resi, w);
  const int mx = (1 << bd) - 1;
  uint64_t dist = 0;
  for (int k = 0; k < 2; k++) {
    const int st = e->stride[k + 1]; const int16_t *org = e->org[k + 1] + cy * st + cx;
    for (int j = 0; j < h; j++) for (int i = 0; i < w; i++) {
      const int c = e->resi[j * w + i]; int r;
      /* invTransformCbCr (139-156): the coded block keeps its residual, the other one is derived */
      if (comp == 1) r = k == 0 ? c : (am == 1 ? ((mode < 0 ? -c : c) >> 1) : (mode < 0 ? -c : c));
      else r = k == 1 ? c : ((mode < 0 ? -c : c) >> 1);
      rec_out[k][j * w + i] = (int16_t) r;
    }
    if (cadj) scale_residual(rec_out[k], n, cadj, 0, bd);              /* 3060-3070: both residuals back through the inverse chroma scaling */
    for (int i = 0; i < n; i++) { const int v = e->pred_c[k][i] + rec_out[k][i]; rec_out[k][i] = (int16_t) (v < 0 ? 0 : v > mx ? mx : v); }
    dist += (uint64_t) (e->sl.dist_weight[k] * (double) orc_sse(org, st, rec_out[k], w, w, h));
  }
  return dist;
}
/* chroma part of transform_unit (EL/CABACWriter.cpp:3560-3680): cbf_cb, cbf_cr, joint_cb_cr (JointCbCr on and some cbf set; context cbfMask - 1),
 * residual_coding of Cb and of Cr (not coded when tu.jointCbCr == 3); lfl collects the LFNST conditions of the coded blocks */
static void enc_chroma_tu(orc_enc *e, int W, int H, int cbf, int jccr, const int16_t *lev0, const int16_t *lev1, int *lastPos, int *violates)
{
  const int maxPos = ((W == 4 && H == 4) || (W == 8 && H == 8)) ? 7 : 15;
  orc_enc_bin(&e->cabac, (unsigned) !!(cbf & 2), ORC_CTX_QtCbf[1]);
  orc_enc_bin(&e->cabac, (unsigned) !!(cbf & 4), ORC_CTX_QtCbf[2] + !!(cbf & 2));
  const int mask = ((cbf & 2) ? 2 : 0) | ((cbf & 4) ? 1 : 0);
  if ((e->cfg.tools & ORC_TOOL_JCCR) && mask) orc_enc_bin(&e->cabac, jccr ? 1u : 0u, ORC_CTX_JointCbCrFlag + mask - 1);
  if (cbf & 2) { orc_residual_coding(&e->cabac, lev0, W, H, 1); if (W >= 4 && H >= 4) { *lastPos |= e->cabac.last_scan_pos >= 1; *violates |= e->cabac.last_scan_pos > maxPos; } }
  if ((cbf & 4) && jccr != 3) { orc_residual_coding(&e->cabac, lev1, W, H, 1); if (W >= 4 && H >= 4) { *lastPos |= e->cabac.last_scan_pos >= 1; *violates |= e->cabac.last_scan_pos > maxPos; } }
}

static uint64_t est_intra_pred_chroma(orc_enc *e, area_t a, int lm_ok, int lfnst_idx, int *out_dir, int *out_cbf, int *out_jccr)
{
  const int cx = a.x >> 1, cy = a.y >> 1, cw = a.w >> 1, chh = a.h >> 1;
  orc_cabac ctxStart; orc_ctx_copy(&ctxStart, &e->cabac);
  e->tu_cadj = chroma_adj(e, a);                              /* 3884-3898: the TU's chroma residual scale (LMCS) */
  int cand[8]; chroma_cand_modes(e, a, cand);
  double bestCost = ORC_MAX_DOUBLE; uint64_t bestDist = 0; int bestMode = 0, bestCbf = 0, bestJccr = 0;
  int16_t *rec2[2], *lev2[2];
  rec2[0] = e->tmp_rec[0]; rec2[1] = e->tmp_rec[1]; lev2[0] = e->tmp_lev[0]; lev2[1] = e->tmp_lev[1];
  static int16_t tmpLM[CCLM_TSTRIDE * CCLM_TSTRIDE], tmpMD[CCLM_TSTRIDE * CCLM_TSTRIDE];
  int infoLM[4] = { 0, 0, 0, 0 }, infoMD[4] = { 0, 0, 0, 0 };
  uint8_t enabled[71]; memset(enabled, 1, sizeof enabled);
  if (lm_ok) {
    /* SATD pre-selection (1479-1582): regular modes except planar and the two MDLM modes are ranked by SATD(Cb) + SATD(Cr); the exchange
     * sort below is the reference's (not stable); the two last entries are dropped from the RD loop */
    cclm_luma(e, cx, cy, cw, chh, 0, tmpLM, infoLM); cclm_luma(e, cx, cy, cw, chh, 1, tmpMD, infoMD);
    int list[8]; int64_t cost[8];
    for (int i = 0; i < 8; i++) { list[i] = 0; cost[i] = 0; }
    for (int idx = 0; idx <= 6; idx++) {
      const int mode = cand[idx];
      list[idx] = mode;
      if (mode == 67 || mode == ORC_PLANAR || mode == ORC_DM_CHROMA) continue;
      int64_t sad = 0;
      for (int c = 1; c <= 2; c++) {
        pred_chroma_comp(e, c, cx, cy, cw, chh, mode, tmpMD, infoMD);
        sad += (int64_t) orc_satd(e->org[c] + cy * e->stride[c] + cx, e->stride[c], e->pred, cw, cw, chh);
        e->cnt_satd++;
      }
      cost[idx] = sad;
    }
    for (int i = 0; i <= 6; i++) for (int j = i + 1; j <= 6; j++) if (cost[j] < cost[i]) {
      const int tm = list[i]; list[i] = list[j]; list[j] = tm;
      const int64_t tc = cost[i]; cost[i] = cost[j]; cost[j] = tc;
    }
    enabled[list[6]] = 0; enabled[list[5]] = 0;
  }
  for (int k = 0; k < 8; k++) {
    const int cm = cand[k];
    const int isLM = cm >= 67 && cm <= 69;
    if (isLM && !lm_ok) continue;                              /* 1588-1591 */
    if (lm_ok && !enabled[cm]) continue;                       /* 1592-1595 */
    orc_ctx_copy(&e->cabac, &ctxStart);
    const int fm = cm == ORC_DM_CHROMA ? colocated_luma_mode(e, a) : cm;   /* PU::getFinalIntraMode 921 */
    int cbf[2]; uint64_t dist = 0;
    const int jccrOn = (e->cfg.tools & ORC_TOOL_JCCR) != 0;
    const int lfDir = isLM ? colocated_luma_mode(e, a) : fm;      /* LFNST kernel choice: DM through the final mode, CCLM through the co-located luma mode (CL/TrQuant.cpp:449-457) */
    double compCost[2] = { 0, 0 }; uint64_t compDist[2];
    for (int c = 1; c <= 2; c++) {
      pred_chroma_comp(e, c, cx, cy, cw, chh, fm, cm == 67 ? tmpLM : tmpMD, cm == 67 ? infoLM : infoMD);
      if (jccrOn) {                                         /* 3823-3843: predictions and residuals of the pair for the joint candidates */
        memcpy(e->pred_c[c - 1], e->pred, (size_t) cw * chh * 2);
        const int st = e->stride[c]; const int16_t *org = e->org[c] + cy * st + cx;
        for (int j = 0; j < chh; j++) for (int i = 0; i < cw; i++) e->resi_c[c - 1][j * cw + i] = (int16_t) (org[j * st + i] - e->pred[j * cw + i]);
        if (e->tu_cadj && cw * chh > 4) scale_residual(e->resi_c[c - 1], cw * chh, e->tu_cadj, 1, e->cfg.bit_depth);      /* 3915-3925: the stored residuals are the scaled ones */
      }
      e->tu_cbf_cb = c == 2 ? cbf[0] : 0;
      compDist[c - 1] = code_tu_block_ex(e, c, cx, cy, cw, chh, 0, lfnst_idx, lfDir, rec2[c - 1], lev2[c - 1], &cbf[c - 1]);
      dist += compDist[c - 1];
      /* xGetIntraFracBitsQTChroma (2625-2692): cbf [+ the joint flag (0) behind Cr's cbf] + coefficients; the contexts advance */
      e->cabac.bits = 0;
      orc_enc_bin(&e->cabac, (unsigned) cbf[c - 1], ORC_CTX_QtCbf[c] + (c == 2 ? cbf[0] : 0));
      if (jccrOn && c == 2 && (cbf[0] || cbf[1])) orc_enc_bin(&e->cabac, 0, ORC_CTX_JointCbCrFlag + ((cbf[0] ? 2 : 0) | (cbf[1] ? 1 : 0)) - 1);
      if (cbf[c - 1]) orc_residual_coding(&e->cabac, lev2[c - 1], cw, chh, 1);
      compCost[c - 1] = rd_cost(e, e->cabac.bits, compDist[c - 1]);
    }
    int jccr = 0, cbfMask = (cbf[0] ? 2 : 0) | (cbf[1] ? 1 : 0);
    if (jccrOn && cbfMask) {
      /* 4060-4150: joint candidates against the sum of the two separate costs; each starts from the TU's start contexts */
      double bestCbCr = compCost[0] + compCost[1];
      orc_cabac ctxSep; orc_ctx_copy(&ctxSep, &e->cabac);
      int masks[2]; const int nm = ict_candidates(e, e->resi_c[0], e->resi_c[1], cw * chh, masks, e->resi);
      static int16_t jrec[2][64 * 64], jlev[64 * 64], brec[2][64 * 64], blev[64 * 64];
      int16_t *const jr[2] = { jrec[0], jrec[1] };
      orc_cabac ctxJ; int haveJ = 0; uint64_t bestJDist = 0;
      for (int q = 0; q < nm; q++) {
        const int mask = masks[q];
        orc_ctx_copy(&e->cabac, &ctxStart);
        const uint64_t d = code_joint_block(e, mask, cx, cy, cw, chh, lfnst_idx, lfDir, jr, jlev);
        if (d == UINT64_MAX) continue;
        e->cabac.bits = 0;
        orc_enc_bin(&e->cabac, (unsigned) (mask >> 1), ORC_CTX_QtCbf[1]);
        orc_enc_bin(&e->cabac, (unsigned) (mask & 1), ORC_CTX_QtCbf[2] + (mask >> 1));
        orc_enc_bin(&e->cabac, 1, ORC_CTX_JointCbCrFlag + mask - 1);
        orc_residual_coding(&e->cabac, jlev, cw, chh, 1);
        const double cj = rd_cost(e, e->cabac.bits, d);
        if (cj < bestCbCr) {
          bestCbCr = cj; bestJDist = d; jccr = mask; haveJ = 1;
          memcpy(brec[0], jrec[0], (size_t) cw * chh * 2); memcpy(brec[1], jrec[1], (size_t) cw * chh * 2); memcpy(blev, jlev, (size_t) cw * chh * 2);
          orc_ctx_copy(&ctxJ, &e->cabac);
        }
      }
      if (haveJ) {
        dist = bestJDist; cbf[0] = jccr >> 1; cbf[1] = jccr & 1;
        memcpy(rec2[0], brec[0], (size_t) cw * chh * 2); memcpy(rec2[1], brec[1], (size_t) cw * chh * 2);
        /* the coded block's levels sit with its component; the other block has none */
        memset(lev2[0], 0, (size_t) cw * chh * 2); memset(lev2[1], 0, (size_t) cw * chh * 2);
        memcpy((jccr >> 1) ? lev2[0] : lev2[1], blev, (size_t) cw * chh * 2);
        orc_ctx_copy(&e->cabac, &ctxJ);
      } else orc_ctx_copy(&e->cabac, &ctxSep);
    }
    /* 1611-1621: contexts are NOT reset (transform skip off); xGetIntraFracBitsQT(chroma) */
    e->cabac.bits = 0;
    enc_intra_chroma_pred_mode(e, a, cm, lm_ok);
    { int lp = 0, vi = 0; enc_chroma_tu(e, cw, chh, (cbf[0] ? 2 : 0) | (cbf[1] ? 4 : 0), jccr, lev2[0], lev2[1], &lp, &vi); }
    const double cost = rd_cost(e, e->cabac.bits, dist);
    if (cost < bestCost) {
      bestCost = cost; bestDist = dist; bestMode = cm; bestCbf = (cbf[0] ? 2 : 0) | (cbf[1] ? 4 : 0); bestJccr = jccr;
      for (int c = 0; c < 2; c++) { memcpy(e->best_rec[c], rec2[c], (size_t) cw * chh * 2); memcpy(e->best_lev[c], lev2[c], (size_t) cw * chh * 2); }
    }
  }
  orc_ctx_copy(&e->cabac, &ctxStart);
  e->tu_cadj = 0;
  *out_dir = bestMode; *out_cbf = bestCbf; *out_jccr = bestJccr;
  return bestDist;
}

/* ------------------------------------------------------------------------------------------------
 * node store: copy node area between picture planes/maps and the per-level store
 * ---------------------------------------------------------------------------------------------- */
static void store_save_from_picture(orc_enc *e, int d, int ch, area_t a)
{
  const int x1 = imin(a.x + a.w, e->wl), y1 = imin(a.y + a.h, e->hl);
  const int c0 = ch ? 1 : 0, c1 = ch ? 2 : 0, sh = ch ? 1 : 0;
  for (int c = c0; c <= c1; c++) {
    const int X0 = a.x >> sh, Y0 = a.y >> sh, X1 = x1 >> sh, Y1 = y1 >> sh, W = a.w >> sh;
    for (int y = Y0; y < Y1; y++) { memcpy(e->store[d].rec[c] + (y - Y0) * W, e->rec[c] + y * e->stride[c] + X0, (size_t) (X1 - X0) * 2);
                                    memcpy(e->store[d].lev[c] + (y - Y0) * W, e->lev[c] + y * e->stride[c] + X0, (size_t) (X1 - X0) * 2); }
  }
  for (int y = a.y >> 2; y < (y1 + 3) >> 2; y++) memcpy(e->store[d].units + (y - (a.y >> 2)) * 32, e->um[ch] + y * e->uw + (a.x >> 2), (size_t) (((x1 + 3) >> 2) - (a.x >> 2)) * sizeof(unit_t));
}
static void store_restore_to_picture(orc_enc *e, int d, int ch, area_t a)
{
  const int x1 = imin(a.x + a.w, e->wl), y1 = imin(a.y + a.h, e->hl);
  const int c0 = ch ? 1 : 0, c1 = ch ? 2 : 0, sh = ch ? 1 : 0;
  for (int c = c0; c <= c1; c++) {
    const int X0 = a.x >> sh, Y0 = a.y >> sh, X1 = x1 >> sh, Y1 = y1 >> sh, W = a.w >> sh;
    for (int y = Y0; y < Y1; y++) { memcpy(e->rec[c] + y * e->stride[c] + X0, e->store[d].rec[c] + (y - Y0) * W, (size_t) (X1 - X0) * 2);
                                    memcpy(e->lev[c] + y * e->stride[c] + X0, e->store[d].lev[c] + (y - Y0) * W, (size_t) (X1 - X0) * 2); }
  }
  for (int y = a.y >> 2; y < (y1 + 3) >> 2; y++) {
    memcpy(e->um[ch] + y * e->uw + (a.x >> 2), e->store[d].units + (y - (a.y >> 2)) * 32, (size_t) (((x1 + 3) >> 2) - (a.x >> 2)) * sizeof(unit_t));
    for (int x = a.x >> 2; x < (x1 + 3) >> 2; x++) e->avail[ch][y * e->uw + x] = e->um[ch][y * e->uw + x].valid ? (uint8_t) (e->cur_tile + 1) : 0;
  }
}
/* an intra CU result (tiles with stride w) into the store of level d */
static void store_save_intra(orc_enc *e, int d, int ch, area_t a, const unit_t *cu)
{
  const int sh = ch ? 1 : 0, W = a.w >> sh, H = a.h >> sh;
  if (!ch) { memcpy(e->store[d].rec[0], e->best_rec[0], (size_t) W * H * 2); memcpy(e->store[d].lev[0], e->best_lev[0], (size_t) W * H * 2); }
  else for (int c = 0; c < 2; c++) { memcpy(e->store[d].rec[c + 1], e->best_rec[c], (size_t) W * H * 2); memcpy(e->store[d].lev[c + 1], e->best_lev[c], (size_t) W * H * 2); }
  for (int y = 0; y < (a.h + 3) >> 2; y++) for (int x = 0; x < (a.w + 3) >> 2; x++) { e->store[d].units[y * 32 + x] = *cu; e->store[d].units[y * 32 + x].valid = 1; e->store[d].units[y * 32 + x].tile = (uint8_t) e->cur_tile; }
}

/* ------------------------------------------------------------------------------------------------
 * xCompressCU and friends
 * ---------------------------------------------------------------------------------------------- */
typedef struct {            /* ComprCUCtx (EL/EncModeCtrl.h:182-249), intra-relevant part */
  int modes[8], nmodes;     /* stack, popped from the back */
  int min_depth, max_depth;
  int did_horz, did_vert, did_quad, do_trih, do_triv, qt_before_bt, max_qt_sub_depth;
  cs_sum *best;             /* bestCS (NULL until a mode result was accepted) */
  int reusing, d;           /* IS_REUSING_CU; recursion level (index of the node's store) */
  double best_cost_wo_split, no_isp_cost; int skip_second_mts;     /* bestCostWithoutSplitFlags, bestCostMtsFirstPassNoIsp, skipSecondMTSPass (EL/EncModeCtrl.h:205-247) */
} cu_ctx;

/* ---- BestEncInfoCache (EL/EncModeCtrl.cpp:663-1110), REUSE_CU_RESULTS ---- */
static cache_ent *cache_entry(orc_enc *e, area_t a)
{
  return &e->cache[((((a.y & 127) >> 2) * 32 + ((a.x & 127) >> 2)) * 6 + (ilog2(a.w) - 2)) * 6 + (ilog2(a.h) - 2)];
}
/* isValid (987-1024) with isTheSameNbHood (664-703): an entry of the same area and channel type, reached along another
 * split path, is reusable iff the CU sits at the origin of the last common ancestor of the two paths.  poc, QP, tree
 * and mode type are constant here; currQgEnable() is true only at CTU level (CL/UnitPartitioner.cpp:369, CuQpDeltaSubdiv 0). */
static int cache_is_valid(orc_enc *e, const partitioner *P)
{
  if (!(e->cfg.tools & ORC_TOOL_CU_REUSE) || P->n == 1) return 0;
  const cache_ent *c = cache_entry(e, P->cur);
  if (!c->valid || c->ch != P->ch) return 0;
  int i = 1;
  for (; i < P->n; i++) {
    const int dpt = i - 1, s = dpt >= c->depth ? SPLIT_NONE : (int) ((c->ss >> (5 * dpt)) & 31);   /* CU::getSplitAtDepth, CL/UnitTools.cpp:251 */
    if (P->st[i].split != s) break;
  }
  const area_t anc = P->st[i - 1].parts[P->st[i - 1].idx];
  return anc.x == P->cur.x && anc.y == P->cur.y;
}
/* setFromCs (939-985): called from tryMode(ETM_POST_DONT_SPLIT) when bestCS is a single unsplit CU; its data lives in store[d] */
static void cache_set_from_cs(orc_enc *e, const partitioner *P, int d)
{
  if (!(e->cfg.tools & ORC_TOOL_CU_REUSE)) return;
  const area_t a = P->cur; const int ch = P->ch;
  cache_ent *c = cache_entry(e, a);
  const unit_t *u = &e->store[d].units[0];
  c->valid = 1; c->ch = (uint8_t) ch; c->dir = u->dir; c->mrl = u->mrl; c->cbf = u->cbf; c->depth = u->depth; c->mts = u->mts; c->lfnst = u->lfnst; c->jccr = u->jccr; c->isp = u->isp; c->tucbf = u->tucbf; c->ss = u->split_series;
  if (!c->lev) c->lev = (int16_t *) malloc((size_t) a.w * a.h * 2);
  if (!ch) memcpy(c->lev, e->store[d].lev[0], (size_t) a.w * a.h * 2);
  else { const size_t n = (size_t) (a.w >> 1) * (a.h >> 1); memcpy(c->lev, e->store[d].lev[1], n * 2); memcpy(c->lev + n, e->store[d].lev[2], n * 2); }
}

static int mode_to_split(int m) { return m == ETM_SPLIT_QT ? SPLIT_QT : m == ETM_SPLIT_BT_H ? SPLIT_BH : m == ETM_SPLIT_BT_V ? SPLIT_BV : m == ETM_SPLIT_TT_H ? SPLIT_TH : m == ETM_SPLIT_TT_V ? SPLIT_TV : SPLIT_NONE; }

/* EncModeCtrlMTnoRQT::tryMode (EL/EncModeCtrl.cpp:1557-2068), I-slice / intra-only subset */
static int try_mode(orc_enc *e, partitioner *P, cu_ctx *C, int mode)
{
  const int impl = part_implicit_split(e, P);
  if (impl != SPLIT_NONE && mode != ETM_SPLIT_QT) return mode_to_split(mode) == impl;
  else if (impl != SPLIT_NONE) return part_can(e, P, SPLIT_QT);
  const area_t a = P->cur;
  if (C->reusing) {                               /* 1585-1598 */
    if (mode == ETM_RECO_CACHED) return 1;
    if (mode == ETM_INTRA) return 0;
  }
  if (C->min_depth > P->qt_depth && part_can(e, P, SPLIT_QT)) return mode == ETM_SPLIT_QT;
  else if (mode == ETM_SPLIT_QT && C->max_depth <= P->qt_depth) return 0;
  if (mode == ETM_INTRA) {
    if (a.w * a.h > 4096) return 0;               /* LCTUFast (1642) */
    if (a.w > 64 || a.h > 64) return 0;           /* dual tree (1647) */
    return 1;
  }
  if (mode == ETM_POST_DONT_SPLIT) {              /* bookkeeping only (2005-2067) */
    if (C->best && !C->best->is_split) cache_set_from_cs(e, P, C->d);
    return 0;
  }
  const int split = mode_to_split(mode);
  if (!part_can(e, P, split)) {
    if (split == SPLIT_BH) C->did_horz = 0;
    if (split == SPLIT_BV) C->did_vert = 0;
    if (split == SPLIT_QT) C->did_quad = 0;
    return 0;
  }
  const cs_sum *b = C->best;
  int feat = -1;
  switch (split) {
    case SPLIT_QT:
      if (!C->qt_before_bt && b) {
        const int maxBTD = e->cfg.max_bt_depth[P->ch];
        if (((b->f_bt == 0 && maxBTD >= 3) || (b->f_bt == 1 && b->l_bt == 1 && maxBTD >= 4)) && (a.w <= 64 && a.h <= 64) && C->did_horz && C->did_vert) return 0;
      }
      break;
    case SPLIT_BH: feat = 0; break;
    case SPLIT_BV: feat = 1; break;
    case SPLIT_TH:
      if (C->did_horz && b && b->f_bt == P->bt_depth && !b->f_cbf) return 0;
      if (!C->do_trih) return 0;
      break;
    case SPLIT_TV:
      if (C->did_vert && b && b->f_bt == P->bt_depth && !b->f_cbf) return 0;
      if (!C->do_triv) return 0;
      break;
  }
  if (split != SPLIT_QT && C->qt_before_bt && C->did_quad && C->max_qt_sub_depth > P->qt_depth + 1) {
    if (feat == 0) C->did_horz = 0; else if (feat == 1) C->did_vert = 0;
    return 0;
  }
  if (split == SPLIT_QT) C->did_quad = 1;
  return 1;
}
static int next_mode(orc_enc *e, partitioner *P, cu_ctx *C)
{
  C->nmodes--;
  while (C->nmodes > 0 && !try_mode(e, P, C, C->modes[C->nmodes - 1])) C->nmodes--;
  return C->nmodes > 0;
}
/* initCULevel (1203-1549) */
static void init_cu_level(orc_enc *e, partitioner *P, cu_ctx *C, int d)
{
  memset(C, 0, sizeof *C);
  C->d = d; C->best_cost_wo_split = ORC_MAX_DOUBLE; C->no_isp_cost = ORC_MAX_DOUBLE;
  const int ch = P->ch, sh = ch ? 1 : 0;
  C->min_depth = 0; C->max_depth = 7 - ilog2(e->cfg.min_qt[ch]);     /* plain QTBTPartitioner: no adaptive depth (EL/EncCu.cpp:472) */
  const area_t a = P->cur;
  const unit_t *cuL = get_cu(e, ch, (a.x >> sh) - 1, a.y >> sh), *cuA = get_cu(e, ch, a.x >> sh, (a.y >> sh) - 1);
  C->qt_before_bt = ((cuL && cuA && cuL->qt_depth > P->qt_depth && cuA->qt_depth > P->qt_depth)
                  || (cuL && !cuA && cuL->qt_depth > P->qt_depth) || (!cuL && cuA && cuA->qt_depth > P->qt_depth)
                  || (!cuA && !cuL && a.w >= 32)) && (a.w > (e->cfg.min_qt[ch] << 1));
  C->do_trih = C->do_triv = 1;
  if (!C->qt_before_bt) C->modes[C->nmodes++] = ETM_SPLIT_QT;
  if (part_can(e, P, SPLIT_TV)) C->modes[C->nmodes++] = ETM_SPLIT_TT_V;
  if (part_can(e, P, SPLIT_TH)) C->modes[C->nmodes++] = ETM_SPLIT_TT_H;
  if (part_can(e, P, SPLIT_BV)) { C->modes[C->nmodes++] = ETM_SPLIT_BT_V; C->did_vert = 1; }
  if (part_can(e, P, SPLIT_BH)) { C->modes[C->nmodes++] = ETM_SPLIT_BT_H; C->did_horz = 1; }
  if (C->qt_before_bt) C->modes[C->nmodes++] = ETM_SPLIT_QT;
  C->modes[C->nmodes++] = ETM_POST_DONT_SPLIT;
  C->reusing = cache_is_valid(e, P);               /* 1438-1444 */
  if (C->reusing) C->modes[C->nmodes++] = ETM_RECO_CACHED;
  C->modes[C->nmodes++] = ETM_INTRA;
  if (!try_mode(e, P, C, C->modes[C->nmodes - 1])) next_mode(e, P, C);
}
/* useModeResult (2089-2200) */
static int use_mode_result(orc_enc *e, partitioner *P, cu_ctx *C, int mode, const cs_sum *t)
{
  if (mode == ETM_SPLIT_QT) C->max_qt_sub_depth = t->max_qt;
  const int maxMtD = e->cfg.max_bt_depth[P->ch] + P->impl_bt_depth;
  const int sh = P->ch ? 1 : 0;
  if (mode == ETM_SPLIT_BT_H && t->n_cu > 2) { const int h2 = (P->cur.h >> sh) / 2; C->do_trih = t->f_h < h2 || t->l_h < h2 || P->mt_depth + 1 == maxMtD; }
  else if (mode == ETM_SPLIT_BT_V && t->n_cu > 2) { const int w2 = (P->cur.w >> sh) / 2; C->do_triv = t->f_w < w2 || t->l_w < w2 || P->mt_depth + 1 == maxMtD; }
  return t->cost != ORC_MAX_DOUBLE && (!C->best || t->cost < C->best->cost);
}

static void compress_cu(orc_enc *e, partitioner *P, int d, double maxCostAllowed, cs_sum *best);

/* The fork's addition to xCompressCU (EL/EncCu.cpp:816-1217), luma tree only: features of the node and of its coded neighbour CUs,
 * forest prediction, then the mode stack is replaced by the one predicted mode if the mode controller accepts it and — for the "fuzzy"
 * and "complex" classes — it is not "do not split" (1195-1216, ChangeTestMode EL/EncModeCtrl.cpp:56-93).  The reference hard-codes its
 * 416x240 test sequence as the picture size (833-834); the picture's own size is used here.  Neighbours are looked up with the
 * tile-restricted get_cu so that tiles stay independent streams (the reference's unrestricted getCU also sees CUs of tiles coded
 * earlier; identical for one tile).  Returns the row written to the training dump, or -1. */
static int nb_info(const orc_enc *e, const unit_t *u, int out[3])
{
  out[0] = orc_fast_region_var(e->org[0] + u->y * e->stride[0] + u->x, e->stride[0], 1 << u->lw, 1 << u->lh);   /* get_context 137-163 */
  out[1] = u->qt_depth; out[2] = u->mt_depth;
  return 1;
}
static int fast_partition(orc_enc *e, partitioner *P, cu_ctx *C)
{
  const area_t a = P->cur;
  const int x = a.x, y = a.y, w = imin(a.w, e->wl - a.x), h = imin(a.h, e->hl - a.y);     /* currCsArea is clipped to the picture (810) */
  if (!(a.h < 128 && x + a.w <= e->wl && y + a.h <= e->hl)) return -1;                    /* 836 */
  if (P->mt_depth == 3 || (h == 4 && w == 4)) return -1;                                  /* 838-846 */
  int nb[5][3], n = 0;
  const unit_t *cuL = get_cu(e, 0, x - 1, y), *cuU = get_cu(e, 0, x, y - 1), *cuLU = get_cu(e, 0, x - 1, y - 1);
  if (cuL) {
    n += nb_info(e, cuL, nb[n]);
    const unit_t *cuLD = get_cu(e, 0, x - 1, y + (1 << cuL->lh) + 1);
    if (cuLD && cuLD->y <= y + h) n += nb_info(e, cuLD, nb[n]);
  }
  if (cuU) {
    n += nb_info(e, cuU, nb[n]);
    const unit_t *cuRU = get_cu(e, 0, x + (1 << cuU->lw) + 1, y - 1);
    if (cuRU && cuRU->x < x + w) n += nb_info(e, cuRU, nb[n]);
  }
  if (cuLU && !((cuLU->y + (1 << cuLU->lh)) > y || (cuLU->x + (1 << cuLU->lw)) > x)) n += nb_info(e, cuLU, nb[n]);
  if (n < 3) return -1;                                                                   /* 933 */
  int feat[27];
  feat[0] = h; feat[1] = w; feat[2] = P->qt_depth; feat[3] = P->mt_depth;
  orc_fast_block_features(e->org[0] + y * e->stride[0] + x, e->stride[0], w, h, feat);
  orc_fast_context_features(nb, n, feat);
  feat[26] = feat[10] < feat[13] ? 0 : feat[10] > feat[12] ? 2 : 1;                       /* simple / complex / fuzzy (1127-1138) */
  e->cnt_fast++;
  int row = -1;
  if (e->dump && e->dump_n < e->dump_cap) { row = e->dump_n++; for (int k = 0; k < 27; k++) e->dump[row * 28 + k] = feat[k]; e->dump[row * 28 + 27] = -1; }
  if (e->cfg.tools & ORC_TOOL_FAST) {
    const int res = orc_forest_predict(&e->forest, feat);
    if (res >= 0 && res <= 5) {
      static const int mode_of[6] = { ETM_INTRA, ETM_SPLIT_QT, ETM_SPLIT_BT_H, ETM_SPLIT_BT_V, ETM_SPLIT_TT_H, ETM_SPLIT_TT_V };
      int valid = try_mode(e, P, C, mode_of[res]);                                        /* tryModeMaster 1199 */
      if (feat[26] >= 1 && res == 0) valid = 0;
      if (valid) { C->modes[0] = mode_of[res]; C->nmodes = 1; }
    }
  }
  return row;
}

/* xCheckRDCostIntra (EL/EncCu.cpp:2402-2777), single pass (no LFNST/MTS loops) */
/* reconstruct one block from given levels: prediction in e->pred; DecCu::xIntraRecBlk (DL/DecCu.cpp:199-414) */
static uint64_t recon_from_levels(orc_enc *e, int comp, int x, int y, int w, int h, const int16_t *lev, int cbf, int mts_idx, int lfnst_idx, int lfnst_dir, int16_t *rec_out)
{
  const int lf = (lfnst_idx && w >= 4 && h >= 4) ? lfnst_idx : 0, lmode = lf ? orc_lfnst_mode(lfnst_dir, w, h) : 0;
  const int st = e->stride[comp], bd = e->cfg.bit_depth, mx = (1 << bd) - 1;
  const int16_t *org = e->org[comp] + y * st + x;
  const int qp = (comp ? e->sl.qp_c[comp - 1] : e->sl.qp) + 6 * (e->cfg.bit_depth - 8);    /* QpParam: + QpBDOffset (CL/Quant.cpp:68-106) */
  if (cbf && !comp && mts_idx == 1) { orc_dequant_ts(lev, w, h, bd, qp, e->coef); orc_ts_inv(e->coef, w, h, bd, e->resi, w); }      /* a transform-skip block: Quant::dequant + xITransformSkip */
  else if (cbf) { if (e->cfg.tools & ORC_TOOL_DEPQUANT) orc_dequant_dq(lev, w, h, bd, qp, e->coef); else orc_dequant(lev, w, h, bd, qp, e->coef); orc_inv_lfnst(e->coef, w, h, lmode, lf); orc_inv_2d_mts(e->coef, w, h, bd, mts_idx, e->resi, w); }
  else memset(e->resi, 0, (size_t) w * h * 2);
  if (comp && cbf && e->tu_cadj && w * h > 4) scale_residual(e->resi, w * h, e->tu_cadj, 0, bd);      /* DL/DecCu.cpp:359-367 */
  for (int j = 0; j < h; j++) for (int i = 0; i < w; i++) { int v = e->pred[j * w + i] + e->resi[j * w + i]; rec_out[j * w + i] = (int16_t) (v < 0 ? 0 : v > mx ? mx : v); }
  uint64_t dd = orc_sse(org, st, rec_out, w, w, h);
  if (comp) dd = (uint64_t) (e->sl.dist_weight[comp - 1] * (double) dd);
  return dd;
}
/* xReuseCachedResult (EL/EncCu.cpp:5665-5771): cached mode + levels re-reconstructed against the current neighbourhood */
static uint64_t reuse_cached(orc_enc *e, area_t a, int ch, int *out_dir, int *out_mrl, int *out_cbf, int *out_mts, int *out_lfnst, int *out_jccr, int *out_isp, int *out_tucbf)
{
  const cache_ent *c = cache_entry(e, a);
  uint64_t dist = 0;
  e->cnt_reuse++;
  if (!ch && c->isp) {                               /* an ISP CU: every sub-partition predicted from the reconstruction of the one before */
    isp_res r;
    isp_code_cu(e, a, c->dir, c->isp, 0, c->lev, c->tucbf, e->best_rec[0], e->best_lev[0], &r);
    dist = r.dist;
  } else if (!ch) {
    pred_luma_cand(e, a.x, a.y, a.w, a.h, c->dir, c->mrl);
    dist = recon_from_levels(e, 0, a.x, a.y, a.w, a.h, c->lev, c->cbf & 1, c->mts, c->lfnst, (c->mrl & MIP_FLAG) ? ORC_PLANAR : c->dir, e->best_rec[0]);
    memcpy(e->best_lev[0], c->lev, (size_t) a.w * a.h * 2);
  } else {
    const int cx = a.x >> 1, cy = a.y >> 1, cw = a.w >> 1, chh = a.h >> 1;
    e->tu_cadj = (c->cbf & 6) ? chroma_adj(e, a) : 0;
    const int fm = c->dir == ORC_DM_CHROMA ? colocated_luma_mode(e, a) : c->dir;
    static int16_t tmpC[CCLM_TSTRIDE * CCLM_TSTRIDE]; int infoC[4] = { 0, 0, 0, 0 };
    if (fm >= 67 && fm <= 69) cclm_luma(e, cx, cy, cw, chh, fm != 67, tmpC, infoC);
    const int lfDir = (fm >= 67 && fm <= 69) ? colocated_luma_mode(e, a) : fm;
    if (c->jccr) {
      /* DecCu::xIntraRecQT with a joint TU (DL/DecCu.cpp:330-414): the coded block's residual, the other one through the inverse ICT */
      const int bd = e->cfg.bit_depth, mx = (1 << bd) - 1, mode = ict_mode(e, c->jccr), am = mode < 0 ? -mode : mode, comp = (c->jccr >> 1) ? 1 : 2;
      const int qp = joint_qp(e, c->jccr);
      const int16_t *lv = c->lev + (size_t) (comp - 1) * cw * chh;
      const int lf = (c->lfnst && cw >= 4 && chh >= 4) ? c->lfnst : 0, lmode = lf ? orc_lfnst_mode(lfDir, cw, chh) : 0;
      static int16_t jres[64 * 64];
      orc_dequant_dq(lv, cw, chh, bd, qp, e->coef); orc_inv_lfnst(e->coef, cw, chh, lmode, lf); orc_inv_2d_mts(e->coef, cw, chh, bd, 0, jres, cw);
      for (int k = 0; k < 2; k++) {
        pred_chroma_comp(e, k + 1, cx, cy, cw, chh, fm, tmpC, infoC);
        const int st = e->stride[k + 1]; const int16_t *org = e->org[k + 1] + cy * st + cx;
        for (int j = 0; j < chh; j++) for (int i = 0; i < cw; i++) {
          const int v0 = jres[j * cw + i]; int r;
          if (comp == 1) r = k == 0 ? v0 : (am == 1 ? ((mode < 0 ? -v0 : v0) >> 1) : (mode < 0 ? -v0 : v0));
          else r = k == 1 ? v0 : ((mode < 0 ? -v0 : v0) >> 1);
          e->best_rec[k][j * cw + i] = (int16_t) r;
        }
        if (e->tu_cadj && cw * chh > 4) scale_residual(e->best_rec[k], cw * chh, e->tu_cadj, 0, bd);
        for (int i = 0; i < cw * chh; i++) { const int v = e->pred[i] + e->best_rec[k][i]; e->best_rec[k][i] = (int16_t) (v < 0 ? 0 : v > mx ? mx : v); }
        dist += (uint64_t) (e->sl.dist_weight[k] * (double) orc_sse(org, st, e->best_rec[k], cw, cw, chh));
        memcpy(e->best_lev[k], c->lev + (size_t) k * cw * chh, (size_t) cw * chh * 2);
      }
    } else
    for (int k = 1; k <= 2; k++) {
      pred_chroma_comp(e, k, cx, cy, cw, chh, fm, tmpC, infoC);
      dist += recon_from_levels(e, k, cx, cy, cw, chh, c->lev + (size_t) (k - 1) * cw * chh, (c->cbf >> k) & 1, 0, c->lfnst, lfDir, e->best_rec[k - 1]);
      memcpy(e->best_lev[k - 1], c->lev + (size_t) (k - 1) * cw * chh, (size_t) cw * chh * 2);
    }
  }
  e->tu_cadj = 0;
  *out_dir = c->dir; *out_mrl = c->mrl; *out_cbf = c->cbf; *out_mts = c->mts; *out_lfnst = c->lfnst; *out_jccr = c->jccr; *out_isp = c->isp; *out_tucbf = c->tucbf;
  return dist;
}

/* CU-level syntax of one intra CU on the estimator / writer: cu_pred_data + cu_residual (EL/CABACWriter.cpp:1456-1530, 2002-2052): prediction
 * mode, cbf flags, residual_coding per block and -- LFNST on -- residual_lfnst_mode (3989-4100): the index is present when the CU may use LFNST
 * (not a MIP CU below 16x16, chroma blocks of at least 4x4, at most 64x64 luma), some block's last position is not DC (lfnstLastScanPos 3844-3850),
 * no block has a coefficient beyond the LFNST region (violatesLfnstConstrained 3837-3842) and the luma transform is DCT-II.
 * lev0 / lev1: luma levels, or Cb / Cr levels (stride = block width).  *lfnst_last receives cuCtx.lfnstLastScanPos. */
static void enc_cu_syntax_isp(orc_enc *e, int ch, area_t a, int dir, int mrl, int cbf, int mts, int lfnst, int jccr, int lm_ok, const int16_t *lev0, const int16_t *lev1, int *lfnst_last, int isp, int tucbf);
static void enc_cu_syntax(orc_enc *e, int ch, area_t a, int dir, int mrl, int cbf, int mts, int lfnst, int jccr, int lm_ok, const int16_t *lev0, const int16_t *lev1, int *lfnst_last)
{ enc_cu_syntax_isp(e, ch, a, dir, mrl, cbf, mts, lfnst, jccr, lm_ok, lev0, lev1, lfnst_last, 0, 0); }
/* isp / tucbf: cu.ispMode and the cbf of each sub-partition: transform_tree splits the CU (EL/CABACWriter.cpp:3311-3330), transform_unit codes each luma cbf with the
 * previous sub-partition's cbf as context and infers the last one after all-zero ones (3574-3600); residual_lfnst_mode is absent for ISP CUs (3994) */
static void enc_cu_syntax_isp(orc_enc *e, int ch, area_t a, int dir, int mrl, int cbf, int mts, int lfnst, int jccr, int lm_ok, const int16_t *lev0, const int16_t *lev1, int *lfnst_last, int isp, int tucbf)
{
  if (!ch && isp) {
    const int hor = isp == 1, psz = isp_split_dim(a.w, a.h, hor), tw = hor ? a.w : psz, th = hor ? psz : a.h, n = hor ? a.h / psz : a.w / psz;
    static int16_t t[64 * 16];
    enc_intra_luma_pred_mode_isp(e, a.x, a.y, a.w, a.h, dir, 0, isp);
    for (int k = 0, sofar = 0; k < n; k++) {
      const int ox = hor ? 0 : k * tw, oy = hor ? k * th : 0, c = (tucbf >> k) & 1;
      if (!(k == n - 1 && !sofar)) orc_enc_bin(&e->cabac, (unsigned) c, ORC_CTX_QtCbf[0] + 2 + (k ? (tucbf >> (k - 1)) & 1 : 0));
      if (c) { for (int j = 0; j < th; j++) for (int i = 0; i < tw; i++) t[j * tw + i] = lev0[(oy + j) * a.w + ox + i]; orc_residual_coding_tu(&e->cabac, t, tw, th, 0, 0, 0, 0); }
      sofar |= c;
    }
    if (lfnst_last) *lfnst_last = 0;
    return;
  }
  int lastPos = 0, violates = 0;
  const int W = a.w >> (ch ? 1 : 0), H = a.h >> (ch ? 1 : 0);
  const int maxPos = ((W == 4 && H == 4) || (W == 8 && H == 8)) ? 7 : 15;
  if (!ch) {
    enc_intra_luma_pred_mode(e, a.x, a.y, a.w, a.h, dir, mrl);
    orc_enc_bin(&e->cabac, (unsigned) (cbf & 1), ORC_CTX_QtCbf[0]);
    if (cbf & 1) { orc_residual_coding_tu(&e->cabac, lev0, a.w, a.h, 0, ts_allowed(e, a.w, a.h), mts_allowed(e, a.w, a.h), mts); if (mts != 1) { lastPos |= e->cabac.last_scan_pos >= 1; violates |= e->cabac.last_scan_pos > maxPos; } }      /* 3837-3850: not for transform-skip blocks */
  } else {
    enc_intra_chroma_pred_mode(e, a, dir, lm_ok);
    enc_chroma_tu(e, W, H, cbf, jccr, lev0, lev1, &lastPos, &violates);
  }
  if (lfnst_last) *lfnst_last = lastPos;
  if (!(e->cfg.tools & ORC_TOOL_LFNST)) return;
  if (!ch && (mrl & MIP_FLAG) && !(a.w >= 16 && a.h >= 16)) return;
  if (ch && imin(W, H) < 4) return;
  if (a.w > 64 || a.h > 64) return;
  const int nonDct2 = !ch && (cbf & 1) && mts != 0;
  if (!lastPos || violates || nonDct2) return;
  orc_enc_bin(&e->cabac, lfnst ? 1u : 0u, ORC_CTX_LFNSTIdx + 1);                    /* separate trees: context 1 (4077) */
  if (lfnst) orc_enc_bins_ep(&e->cabac, (uint32_t) (lfnst - 1), 1);
}

static void check_rd_cost_intra(orc_enc *e, partitioner *P, int d, cu_ctx *C, cs_sum *best, orc_cabac *ctxStart, orc_cabac *ctxBest, int reuse, double maxCostAllowed)
{
  const area_t a = P->cur; const int ch = P->ch, sh = ch ? 1 : 0;
  unit_t cu; memset(&cu, 0, sizeof cu);
  cu.x = (int16_t) (a.x >> sh); cu.y = (int16_t) (a.y >> sh); cu.lw = (uint8_t) ilog2(a.w >> sh); cu.lh = (uint8_t) ilog2(a.h >> sh);
  cu.qt_depth = (uint8_t) P->qt_depth; cu.mt_depth = (uint8_t) P->mt_depth; cu.bt_depth = (uint8_t) P->bt_depth; cu.depth = (uint8_t) P->depth;
  cu.split_series = part_split_series(P);
  const int lm_ok = ch ? cclm_allowed(e, a, cu.split_series, cu.depth) : 0;
  /* the pass loop of xCheckRDCostIntra (2417-2777): transform groups x lfnstIdx x mtsFlag; without LFNST (or for a cached CU) a single pass */
  const int lfnstOn = (e->cfg.tools & ORC_TOOL_LFNST) != 0 && !reuse;
  const int considerMts = (e->cfg.tools & ORC_TOOL_MTS) && !ch && a.w <= 32 && a.h <= 32;                      /* 2409 */
  const int maxLfnstIdx = ((ch && (a.w < 8 || a.h < 8)) || a.w > 64 || a.h > 64) ? 0 : 2;                   /* 2431-2436 */
  double dct2Cost = ORC_MAX_DOUBLE, trGrpBestCost[4] = { ORC_MAX_DOUBLE, ORC_MAX_DOUBLE, ORC_MAX_DOUBLE, ORC_MAX_DOUBLE };
  int bestSelFlag[4] = { 0, 0, 0, 0 }, trGrpCheck[4] = { 1, 1, 1, 1 }, bestMtsFlag = 0, bestLfnstIdx = 0;
  int skipOtherLfnst = 0, startLfnstIdx = 0, endLfnstIdx = lfnstOn ? maxLfnstIdx : 0;
  const int grpNumMax = lfnstOn ? 4 : 1;
  static luma_passes ps; memset(ps.best_valid, 0, sizeof ps.best_valid);                                  /* invalidateBestModeCost 2451 */
  for (int trGrp = 0; trGrp < grpNumMax; trGrp++) {
    const int startMtsFlag = trGrp > 0, endMtsFlag = lfnstOn ? considerMts : 0;
    if ((trGrp == 0 || (!C->skip_second_mts && considerMts)) && trGrpCheck[trGrp]) {
      for (int lfnstIdx = startLfnstIdx; lfnstIdx <= endLfnstIdx; lfnstIdx++) {
        for (int mtsFlag = startMtsFlag; mtsFlag <= endMtsFlag; mtsFlag++) {
          if (mtsFlag > 0 && lfnstIdx > 0) continue;                                                      /* JVET_O0368 2463-2466 */
          cs_sum t; memset(&t, 0, sizeof t);
          int dir = 0, mrl = 0, cbf = 0, mts = 0, lfnst = lfnstIdx, valid = 1, jccr = 0, isp = 0, tucbf = 0;
          if (reuse) t.dist = reuse_cached(e, a, ch, &dir, &mrl, &cbf, &mts, &lfnst, &jccr, &isp, &tucbf);
          else if (!ch) {
            ps.lfnst = lfnstIdx; ps.mts_flag = mtsFlag; ps.tr_grp = trGrp;
            /* 2503-2516: what ISP measures itself against: the node's best cost so far without split flags, capped by the budget the parent's split loop left */
            const double bestCostSoFar = maxCostAllowed < C->best_cost_wo_split ? maxCostAllowed : C->best_cost_wo_split;
            t.dist = est_intra_pred_luma(e, a, &ps, &valid, &dir, &mrl, &cbf, &mts, bestCostSoFar, &C->no_isp_cost, &isp, &tucbf); cbf = cbf ? 1 : 0;
            if (lfnstOn && !valid) continue;                                                              /* 2529-2532 */
          } else t.dist = est_intra_pred_chroma(e, a, lm_ok, lfnstIdx, &dir, &cbf, &jccr);
          cu.dir = (uint8_t) dir; cu.mrl = (uint8_t) mrl; cu.cbf = (uint8_t) cbf; cu.mts = (uint8_t) mts; cu.lfnst = (uint8_t) lfnst; cu.jccr = (uint8_t) jccr; cu.isp = (uint8_t) isp; cu.tucbf = (uint8_t) tucbf;
          /* CU-level rate from the node's start contexts (2593-2620) */
          e->cabac.bits = 0;
          int lfnstLast = 0;
          enc_cu_syntax_isp(e, ch, a, dir, mrl, cbf, mts, lfnst, jccr, lm_ok, e->best_lev[0], e->best_lev[1], &lfnstLast, isp, tucbf);
          if (ch && reuse && !e->ctu_is_last) {
            /* the reuse path prices the CU with CABACWriter::coding_unit, whose end_of_ctu (EL/CABACWriter.cpp:2118-2141) adds the
             * terminating bin after the last chroma CU of a CTU that does not end the slice */
            const int endX = a.x + a.w, endY = a.y + a.h;
            if (((endX & 127) == 0 || endX == e->wl) && ((endY & 127) == 0 || endY == e->hl)) e->cabac.bits += 0x10c;   /* estFracBitsTrm(0), CL/Contexts.h:129 */
          }
          t.bits = e->cabac.bits;
          t.cost = rd_cost(e, t.bits, t.dist);
          const double costWithoutSplitFlags = t.cost;                                                     /* 2622-2629: also bestIspCost of an ISP winner */
          /* xEncodeDontSplit (5649-5662) */
          e->cabac.bits = 0;
          enc_split_cu_mode(e, P, SPLIT_NONE);
          t.bits += e->cabac.bits;
          t.cost = rd_cost(e, t.bits, t.dist);
          /* 2633-2645: an LFNST index that cannot be signalled (no block with a last position beyond DC) while there are coefficients */
          if (!reuse && lfnstIdx && !lfnstLast && cbf) t.cost = ORC_MAX_DOUBLE;
          if (mtsFlag == 0 && lfnstIdx == 0) dct2Cost = t.cost;
          if (!reuse && t.cost < best->cost) C->best_cost_wo_split = costWithoutSplitFlags;                 /* 2657-2660 */
          if (!reuse && !isp) C->no_isp_cost = t.cost;                                                     /* useModeResult(ETM_INTRA), EL/EncModeCtrl.cpp:2120-2123 */
          /* checkSkipOtherLfnst (EL/EncModeCtrl.cpp:2070-2087): the intra passes are the first modes of a node, so its condition always holds */
          if (lfnstOn) skipOtherLfnst = !cbf;
          t.n_cu = 1; t.is_split = 0;
          t.f_bt = t.l_bt = P->bt_depth; t.f_depth = P->depth; t.f_mt = P->mt_depth; t.f_cbf = cbf != 0;
          t.f_w = t.l_w = a.w >> sh; t.f_h = t.l_h = a.h >> sh; t.max_qt = P->qt_depth;
          /* xCheckBestMode (677-724) */
          if (use_mode_result(e, P, C, ETM_INTRA, &t)) {
            *best = t; C->best = best;
            store_save_intra(e, d, ch, a, &cu);
            orc_ctx_copy(ctxBest, &e->cabac);
            trGrpBestCost[trGrp] = best->cost; bestSelFlag[trGrp] = 1; bestMtsFlag = mtsFlag; bestLfnstIdx = lfnstIdx;      /* 2696-2701 */
            if (lfnstOn && !ch && mts == 1 && ilog2(a.w) + ilog2(a.h) >= 6) endLfnstIdx = 0;                                 /* 2702-2712: a transform-skip winner of at least 64 samples ends the LFNST passes */
          }
          orc_ctx_copy(&e->cabac, ctxStart);
          /* 2714-2741 (ISPFast 1): an ISP winner of the first pass that beats the best regular mode by a margin ends the LFNST / MTS passes of this CU */
          if (lfnstOn && (considerMts > 0 || endLfnstIdx > 0) && isp && !mtsFlag && !lfnstIdx) {
            const double threshold = 1.4, lfnstThreshold = 1.01 * threshold;
            if (C->no_isp_cost > costWithoutSplitFlags * lfnstThreshold) endLfnstIdx = lfnstIdx;
            if (C->no_isp_cost > costWithoutSplitFlags * threshold) { C->skip_second_mts = 1; break; }
          }
        }
        if (skipOtherLfnst) { startLfnstIdx = lfnstIdx; endLfnstIdx = lfnstIdx; break; }               /* 2754-2759 */
      }
    }
    if (lfnstOn && trGrp < 3) {                                                                            /* 2763-2773 */
      trGrpCheck[trGrp + 1] = 0;
      if (bestSelFlag[trGrp] && considerMts) {
        const double ratio = dct2Cost / trGrpBestCost[trGrp];
        trGrpCheck[trGrp + 1] = (bestMtsFlag != 0 || bestLfnstIdx != 0) && ratio < 1.001;
      }
    }
  }
}

/* xCheckModeSplit (EL/EncCu.cpp:1918-2399) */
static void check_mode_split(orc_enc *e, partitioner *P, int d, cu_ctx *C, int mode, double maxCostAllowed, cs_sum *best, orc_cabac *ctxStart, orc_cabac *ctxBest)
{
  const int split = mode_to_split(mode), ch = P->ch;
  const area_t a = P->cur;
  orc_ctx_copy(&e->cabac, ctxStart);
  e->cabac.bits = 0;
  enc_split_cu_mode(e, P, split);
  const uint64_t splitBits = e->cabac.bits;
  orc_ctx_copy(&e->cabac, ctxStart);            /* sub-contexts restored (1967-1972) */
  const double factor = e->sl.qp > 30 ? 1.1 : 1.075;
  const double cost = rd_cost(e, (uint64_t) ((double) splitBits + ((double) best->bits / factor)), (uint64_t) ((double) best->dist / factor));
  if (cost > best->cost) return;                 /* xCheckBestMode with empty tempCS: nothing */
  cs_sum t; memset(&t, 0, sizeof t);
  t.is_split = split;                                /* PartSplit code (1..5): also the training label */
  part_split(e, P, split);
  set_units(e, ch, a, 0, 0);                     /* tempCS->initStructData: nothing of this node is coded yet */
  int first = 1;
  do {
    const area_t s = P->cur;
    if (s.x < e->wl && s.y < e->hl) {
      double newMax = !ch ? fmin(maxCostAllowed, best->cost - rd_cost(e, t.bits, t.dist)) : ORC_MAX_DOUBLE;
      newMax = fmax(0.0, newMax);
      cs_sum sub; memset(&sub, 0, sizeof sub); sub.cost = ORC_MAX_DOUBLE;
      compress_cu(e, P, d + 1, newMax, &sub);
      if (sub.cost == ORC_MAX_DOUBLE) { fprintf(stderr, "oracle: sub-CU without encoding (not expected in I slices)\n"); abort(); }
      t.dist += sub.dist; t.bits += sub.bits;
      if (first) { t.f_bt = sub.f_bt; t.f_depth = sub.f_depth; t.f_mt = sub.f_mt; t.f_cbf = sub.f_cbf; t.f_w = sub.f_w; t.f_h = sub.f_h; first = 0; }
      t.l_bt = sub.l_bt; t.l_w = sub.l_w; t.l_h = sub.l_h;
      t.n_cu += sub.n_cu; t.max_qt = imax(t.max_qt, sub.max_qt);
    }
  } while (part_next(P));
  part_exit(P);
  const int enforceQT = part_implicit_split(e, P) == SPLIT_QT;
  if (!enforceQT) {
    e->cabac.bits = 0;
    enc_split_cu_mode(e, P, split);               /* from the contexts left by the last child (2322-2329) */
    t.bits += e->cabac.bits;
  }
  t.cost = rd_cost(e, t.bits, t.dist);
  if (use_mode_result(e, P, C, mode, &t)) {
    *best = t; C->best = best;
    store_save_from_picture(e, d, ch, a);
    orc_ctx_copy(ctxBest, &e->cabac);
  }
  orc_ctx_copy(&e->cabac, ctxStart);
}

/* xCompressCU (EL/EncCu.cpp:727-1638) */
static void compress_cu(orc_enc *e, partitioner *P, int d, double maxCostAllowed, cs_sum *best)
{
  cu_ctx C;
  const area_t a = P->cur; const int ch = P->ch;
  e->cnt_nodes++;
  init_cu_level(e, P, &C, d);
  int dump_row = -1;
  if (!ch && ((e->cfg.tools & ORC_TOOL_FAST) || e->dump)) dump_row = fast_partition(e, P, &C);
  orc_cabac ctxStart, ctxBest;
  orc_ctx_copy(&ctxStart, &e->cabac); orc_ctx_copy(&ctxBest, &e->cabac);
  memset(best, 0, sizeof *best); best->cost = ORC_MAX_DOUBLE;
  if (C.nmodes == 0) return;
  int lastWasBest = 0;
  do {
    const int mode = C.modes[C.nmodes - 1];
    const cs_sum *before = C.best; const double costBefore = best->cost;
    if (mode == ETM_INTRA) check_rd_cost_intra(e, P, d, &C, best, &ctxStart, &ctxBest, 0, maxCostAllowed);
    else if (mode == ETM_RECO_CACHED) check_rd_cost_intra(e, P, d, &C, best, &ctxStart, &ctxBest, 1, maxCostAllowed);
    else check_mode_split(e, P, d, &C, mode, maxCostAllowed, best, &ctxStart, &ctxBest);
    lastWasBest = (C.best != before) || (best->cost != costBefore);
  } while (next_mode(e, P, &C));
  (void) lastWasBest;
  if (dump_row >= 0) e->dump[dump_row * 28 + 27] = best->cost == ORC_MAX_DOUBLE ? -1 : best->is_split;
  if (best->cost == ORC_MAX_DOUBLE) return;
  orc_ctx_copy(&e->cabac, &ctxBest);
  /* picture ← bestCS reco (1581-1582); CU info of the winner becomes visible to later nodes */
  store_restore_to_picture(e, d, ch, a);
}

/* final estimator pass over the coded CTU: CABACWriter::coding_tree_unit / coding_tree (254-309, 474-984)
 * advances the contexts for the next CTU of the tile (EL/EncSlice.cpp:1775-1776).  SAO/ALF CTU syntax
 * touches contexts the CU search never reads and is left out. */
static void walk_tree(orc_enc *e, partitioner *P)
{
  const area_t a = P->cur; const int ch = P->ch, sh = ch ? 1 : 0;
  const unit_t *u = &e->um[ch][(a.y >> 2) * e->uw + (a.x >> 2)];
  const int split = (int) ((u->split_series >> (P->depth * 5)) & 31);   /* CU::getSplitAtDepth */
  enc_split_cu_mode(e, P, split ? split : SPLIT_NONE);
  if (split) {
    part_split(e, P, split);
    do { if (P->cur.x < e->wl && P->cur.y < e->hl) walk_tree(e, P); } while (part_next(P));
    part_exit(P);
    return;
  }
  /* coding_unit: cu_pred_data + cu_residual */
  const int W = a.w >> sh, H = a.h >> sh;
  int16_t *lv = e->tmp_lev[0], *lv1 = e->tmp_lev[1];
  if (!ch) {
    if (u->cbf & 1) for (int y = 0; y < H; y++) memcpy(lv + y * W, e->lev[0] + (a.y + y) * e->stride[0] + a.x, (size_t) W * 2);
    enc_cu_syntax_isp(e, 0, a, u->dir, u->mrl, u->cbf, u->mts, u->lfnst, 0, 0, lv, lv1, 0, u->isp, u->tucbf);
  } else {
    for (int c = 1; c <= 2; c++) if (u->cbf & (1 << c))
      for (int y = 0; y < H; y++) memcpy((c == 1 ? lv : lv1) + y * W, e->lev[c] + ((a.y >> 1) + y) * e->stride[c] + (a.x >> 1), (size_t) W * 2);
    enc_cu_syntax(e, 1, a, u->dir, u->mrl, u->cbf, u->mts, u->lfnst, u->jccr, cclm_allowed(e, a, u->split_series, u->depth), lv, lv1, 0);
  }
}
static void advance_ctx_ctu(orc_enc *e, area_t ctu)
{
  partitioner PL, PC;
  part_init_ctu(&PL, ctu, 0); part_init_ctu(&PC, ctu, 1);
  /* 128x128 root: implicit QT for both trees, no bins; then luma/chroma interleaved per 64x64 (867-908) */
  enc_split_cu_mode(e, &PL, SPLIT_QT);
  part_split(e, &PL, SPLIT_QT); part_split(e, &PC, SPLIT_QT);
  int go = 1;
  while (go) {
    if (PL.cur.x < e->wl && PL.cur.y < e->hl) walk_tree(e, &PL);
    go = part_next(&PL);
    if (e->cfg.chroma && PC.cur.x < e->wl && PC.cur.y < e->hl) walk_tree(e, &PC);
    part_next(&PC);
  }
}

/* compressCtu (EL/EncCu.cpp:428-558) */
static void compress_ctu(orc_enc *e, int rx, int ry, orc_ctu_result *res)
{
  const area_t ctu = { rx << 7, ry << 7, 128, 128 };
  for (int i = 0; i < CACHE_ENTRIES; i++) e->cache[i].valid = 0;
  e->ctu_is_last = ry * e->ctus_w + rx == e->ctus_w * e->ctus_h - 1;
  orc_cabac ctuStart; orc_ctx_copy(&ctuStart, &e->cabac);
  /* WaveFrontSynchro: getCURestricted hides every CU of a CTU column to the right (CL/CodingStructure.cpp:1634-1638); the only such CUs a CTU can reach are the ones above-right of
   * it (reference samples of blocks on its top row): their availability marks are taken away for the duration of this CTU */
  uint8_t wpp_keep[2][32]; int wpp_n = 0;
  if ((e->cfg.tools & ORC_TOOL_WPP) && ry > 0 && rx + 1 < e->ctus_w) {
    wpp_n = imin(32, e->uw - (rx + 1) * 32);
    for (int k = 0; k < 2; k++) for (int i = 0; i < wpp_n; i++) { uint8_t *a = &e->avail[k][(ry * 32 - 1) * e->uw + (rx + 1) * 32 + i]; wpp_keep[k][i] = *a; *a = 0; }
  }
  partitioner P; cs_sum best;
  part_init_ctu(&P, ctu, 0);
  compress_cu(e, &P, 0, ORC_MAX_DOUBLE, &best);
  res->dist = best.dist; res->frac_bits = best.bits; res->cost = best.cost; res->n_cu = best.n_cu;
  if (e->cfg.chroma) {
    orc_ctx_copy(&e->cabac, &ctuStart);
    part_init_ctu(&P, ctu, 1);
    compress_cu(e, &P, 0, ORC_MAX_DOUBLE, &best);
    res->dist += best.dist; res->frac_bits += best.bits; res->cost += best.cost; res->n_cu += best.n_cu;
  }
  orc_ctx_copy(&e->cabac, &ctuStart);
  advance_ctx_ctu(e, ctu);
  for (int k = 0; k < 2; k++) for (int i = 0; i < wpp_n; i++) e->avail[k][(ry * 32 - 1) * e->uw + (rx + 1) * 32 + i] = wpp_keep[k][i];
}

/* WaveFrontSynchro: 0 = not the first CTU of a tile's CTU row (or WPP off), 1 = the first CTU of the tile's first row, 2 = of a later row (contexts come from the row above) */
static int wpp_row_start(const orc_enc *e, int rx, int ry, int t)
{
  if (!(e->cfg.tools & ORC_TOOL_WPP)) return 0;
  if (rx > 0 && e->ctu_tile[ry * e->ctus_w + rx - 1] == t) return 0;
  return (ry > 0 && e->ctu_tile[(ry - 1) * e->ctus_w + rx] == t) ? 2 : 1;
}
int orc_compress_frame(orc_enc *e, orc_ctu_result *res, orc_cu *cus, int max_cus, int *n_cus) { return orc_compress_tiles(e, 0, e->cfg.tile_cols * e->cfg.tile_rows, res, cus, max_cus, n_cus); }
/* the tiles [tile_first, tile_first + tile_count) only (tiles are independent streams: test runs spread them over processes); res rows and CU
 * table entries of the other tiles' CTUs are left out */
int orc_compress_tiles(orc_enc *e, int tile_first, int tile_count, orc_ctu_result *res, orc_cu *cus, int max_cus, int *n_cus)
{
  if ((e->cfg.tools & ORC_TOOL_FAST) && !e->forest.n_trees) { snprintf(g_err, sizeof g_err, "oracle: ORC_TOOL_FAST needs orc_set_forest first"); return -1; }
  const int ntiles = e->cfg.tile_cols * e->cfg.tile_rows;
  if (tile_first < 0 || tile_count < 0 || tile_first + tile_count > ntiles) { snprintf(g_err, sizeof g_err, "oracle: tile range"); return -1; }
  for (int t = tile_first; t < tile_first + tile_count; t++) {
    e->cur_tile = t % 255; e->cur_tile_idx = t;
    orc_ctx_init(e->sl.qp, e->cabac.s0, e->cabac.s1);       /* contexts reset at tile start (EL/EncSlice.cpp:1640-1647) */
    orc_cabac sync;                                          /* m_entropyCodingSyncContextState */
    for (int ry = 0; ry < e->ctus_h; ry++) for (int rx = 0; rx < e->ctus_w; rx++) {
      if (e->ctu_tile[ry * e->ctus_w + rx] != t) continue;
      const int row_start = wpp_row_start(e, rx, ry, t);
      if (row_start == 2) orc_ctx_copy(&e->cabac, &sync);    /* 1648-1661: reset, then the state after the first CTU of the row above (always in this tile here) */
      compress_ctu(e, rx, ry, &res[ry * e->ctus_w + rx]);
      if (row_start) orc_ctx_copy(&sync, &e->cabac);         /* 1801-1805 */
    }
  }
  /* final CU table: CTU raster order, per CTU luma CUs then chroma CUs, each in raster order of their origin */
  int n = 0;
  for (int ry = 0; ry < e->ctus_h; ry++) for (int rx = 0; rx < e->ctus_w; rx++)
    for (int ch = 0; ch < (e->cfg.chroma ? 2 : 1); ch++) {
      const int ul = ch ? 1 : 2, tl = e->ctu_tile[ry * e->ctus_w + rx];
      if (tl < tile_first || tl >= tile_first + tile_count) continue;
      for (int uy = ry * 32; uy < imin(ry * 32 + 32, e->uh); uy++) for (int ux = rx * 32; ux < imin(rx * 32 + 32, e->uw); ux++) {
        const unit_t *u = &e->um[ch][uy * e->uw + ux];
        if (!u->valid || (u->x >> ul) != ux || (u->y >> ul) != uy) continue;
        if (n < max_cus) {
          orc_cu *o = &cus[n];
          o->x = u->x; o->y = u->y; o->w = (int16_t) (1 << u->lw); o->h = (int16_t) (1 << u->lh); o->ch_type = (uint8_t) ch;
          o->qt_depth = u->qt_depth; o->bt_depth = u->bt_depth; o->mt_depth = u->mt_depth; o->depth = u->depth;
          o->intra_dir = u->dir; o->mrl_idx = ch ? u->mrl : (u->mrl & ~MIP_FLAG); o->mip_flag = !ch && (u->mrl & MIP_FLAG) ? 1 : 0; o->cbf = u->cbf; o->mts_idx = u->mts; o->lfnst_idx = u->lfnst; o->joint_cb_cr = u->jccr; o->isp_mode = ch ? 0 : u->isp; o->tu_cbf = ch ? 0 : u->tucbf; o->split_series = u->split_series;
        }
        n++;
      }
    }
  *n_cus = n;
  return n > max_cus ? -1 : 0;
}

/* ------------------------------------------------------------------------------------------------
 * slice_data() payload of the coded picture: the final CTU syntax of every tile through the arithmetic coder
 * (EncSlice::encodeSlice, EL/EncSlice.cpp:1884-2006; CABACWriter::coding_tree_unit 254-322; end_of_ctu 2118-2141;
 * end_of_slice = terminating bin 1 + finish; OutputBitstream::writeByteAlignment).  Call after orc_compress_frame.
 * sizes[t] receives the byte count of tile t's sub-stream; returns the total or -1 if buf is too small.
 * ---------------------------------------------------------------------------------------------- */
long orc_write_tiles(orc_enc *e, uint8_t *buf, long cap, int *sizes)
{
  const int ntiles = e->cfg.tile_cols * e->cfg.tile_rows, wpp = (e->cfg.tools & ORC_TOOL_WPP) != 0;
  long total = 0;
  int last_rx = 0, last_ry = 0, nsub = 0;          /* last CTU of the slice in tile-scan order */
  for (int ry = 0; ry < e->ctus_h; ry++) for (int rx = 0; rx < e->ctus_w; rx++) if (e->ctu_tile[ry * e->ctus_w + rx] == ntiles - 1) { last_rx = rx; last_ry = ry; }
  for (int t = 0; t < ntiles; t++) {
    orc_arith aw; memset(&aw, 0, sizeof aw);
    aw.out = buf + total; aw.cap = (size_t) (cap - total);
    e->cur_tile = t % 255; e->cur_tile_idx = t;
    orc_ctx_init(e->sl.qp, e->cabac.s0, e->cabac.s1);
    orc_arith_start(&aw);
    e->cabac.aw = &aw;
    orc_cabac sync;
    for (int ry = 0; ry < e->ctus_h; ry++) for (int rx = 0; rx < e->ctus_w; rx++) {
      if (e->ctu_tile[ry * e->ctus_w + rx] != t) continue;
      const area_t ctu = { rx << 7, ry << 7, 128, 128 };
      const int row_start = wpp_row_start(e, rx, ry, t);
      if (row_start == 2) { orc_arith *keep = e->cabac.aw; orc_ctx_copy(&e->cabac, &sync); e->cabac.aw = keep; }      /* EL/EncSlice.cpp:1945-1960 */
      advance_ctx_ctu(e, ctu);
      if (row_start) orc_ctx_copy(&sync, &e->cabac);                                                                   /* 1972-1976 */
      /* the sub-stream ends with the tile or, under WPP, with the CTU row: end_of_subset_one_bit / end_of_brick_one_bit = terminating bin 1, finish, byte alignment -
       * what the reference's DECODER reads at the end of every CTU row (DL/DecSlice.cpp:236-247).  The reference's encoder means to write it there too, but its row test
       * `(ctuRsAddr + 1 % widthInCtus) == tileXPosInCtus` (EL/EncSlice.cpp:1980) binds as ctuRsAddr + (1 % widthInCtus) and never fires, so its own WPP streams do not decode;
       * this follows the decoder (and the syntax of the standard) */
      const int row_end = wpp && (rx + 1 == e->ctus_w || e->ctu_tile[ry * e->ctus_w + rx + 1] != t);
      const int tile_end = (rx + 1 == e->ctus_w || e->ctu_tile[ry * e->ctus_w + rx + 1] != t) && (ry + 1 == e->ctus_h || e->ctu_tile[(ry + 1) * e->ctus_w + rx] != t);
      if (!(rx == last_rx && ry == last_ry)) orc_arith_trm(&aw, 0);     /* end_of_ctu (EL/CABACWriter.cpp:2118-2141): not the last CTU of the slice */
      if (!row_end && !tile_end) continue;
      orc_arith_trm(&aw, 1); orc_arith_finish(&aw);
      orc_bs_write(&aw, 1, 1); while (aw.bit_n) orc_bs_write(&aw, 0, 1);   /* writeByteAlignment */
      if (aw.n > aw.cap) { e->cabac.aw = 0; return -1; }
      sizes[nsub++] = (int) aw.n; total += (long) aw.n;
      if (!tile_end) { memset(&aw, 0, sizeof aw); aw.out = buf + total; aw.cap = (size_t) (cap - total); orc_arith_start(&aw); }
    }
    e->cabac.aw = 0;
  }
  return total;
}
/* quantised levels of the coded picture, plane layout (stride = plane width) */
int orc_get_levels(orc_enc *e, int16_t *const lev[3])
{
  for (int c = 0; c < 3; c++) memcpy(lev[c], e->lev[c], (size_t) (c ? e->wc * e->hc : e->wl * e->hl) * 2);
  return 0;
}

/* ------------------------------------------------------------------------------------------------
 * test hooks (used by tests/test_oracle_golden.py against vectors from the real reference)
 * nb rows: {ch, x, y, w, h (luma coordinates), qtDepth, dir}
 * ---------------------------------------------------------------------------------------------- */
static orc_enc *test_env(int pic_w, int pic_h, const int *nb, int n_nb)
{
  orc_cfg cfg; memset(&cfg, 0, sizeof cfg);
  cfg.pic_w = pic_w; cfg.pic_h = pic_h; cfg.bit_depth = 8; cfg.ctu_size = 128; cfg.min_qt[0] = 8; cfg.min_qt[1] = 4;
  cfg.max_bt_depth[0] = cfg.max_bt_depth[1] = 3; cfg.max_bt_size[0] = 32; cfg.max_bt_size[1] = 64; cfg.max_tt_size[0] = cfg.max_tt_size[1] = 32;
  cfg.dual_tree = 1; cfg.tile_cols = cfg.tile_rows = 1; cfg.tools = ORC_TOOL_MRL; cfg.chroma = 1;
  orc_enc *e = orc_create(&cfg);
  for (int i = 0; i < n_nb; i++) {
    const int *r = nb + 7 * i; const int ch = r[0], sh = ch ? 1 : 0;
    unit_t u; memset(&u, 0, sizeof u);
    u.x = (int16_t) (r[1] >> sh); u.y = (int16_t) (r[2] >> sh); u.lw = (uint8_t) ilog2(r[3] >> sh); u.lh = (uint8_t) ilog2(r[4] >> sh);
    u.qt_depth = (uint8_t) r[5]; u.dir = (uint8_t) r[6];
    set_units(e, ch, (area_t) { r[1], r[2], r[3], r[4] }, &u, 1);
  }
  return e;
}
int orc_test_partition(int pic_w, int pic_h, int ch, int ctux, int ctuy, const int *nb, int n_nb, const int *path, int npath,
                       int *can, unsigned *ctx, int *implicit, int *area)
{
  orc_enc *e = test_env(pic_w, pic_h, nb, n_nb);
  partitioner P; part_init_ctu(&P, (area_t) { ctux, ctuy, 128, 128 }, ch);
  for (int i = 0; i < npath; i++) { part_split(e, &P, path[2 * i]); for (int k = 0; k < path[2 * i + 1]; k++) part_next(&P); }
  part_can_split(e, &P, can);
  *implicit = part_implicit_split(e, &P);
  derive_split_ctx(e, &P, can, ctx);
  area[0] = P.cur.x; area[1] = P.cur.y; area[2] = P.cur.w; area[3] = P.cur.h; area[4] = P.qt_depth; area[5] = P.bt_depth; area[6] = P.mt_depth; area[7] = P.depth;
  orc_destroy(e);
  return 0;
}
int orc_test_mpm(int pic_w, int pic_h, const int *nb, int n_nb, int x, int y, int w, int h, unsigned *mpm)
{
  orc_enc *e = test_env(pic_w, pic_h, nb, n_nb);
  get_mpms(e, x, y, w, h, mpm);
  orc_destroy(e);
  return 0;
}

/* ------------------------------------------------------------------------------------------------
 * deblocking of the coded picture (LoopFilter::loopFilterPic, CL/LoopFilter.cpp:153-262): every CU has one TU, so the filtered edges are
 * the left and top edges of the CUs of each tree; all vertical edges first, then all horizontal ones; boundary strength 2 everywhere.
 * Luma edges lie on the 4x4 grid, chroma edges on the 8x8 chroma-sample grid (1217-1227); every CU has the slice QP.
 * ---------------------------------------------------------------------------------------------- */
/* size, across an edge of direction dir (0: vertical edges -> widths, 1: horizontal edges -> heights), of the transform unit a luma
 * 4x4 unit lies in: the CU's size, or the ISP sub-partition's when the CU is split that way (CU::getISPSplitDim, CL/UnitTools.cpp) */
static int dbk_tu_size(const unit_t *u, int dir)
{
  const int w = 1 << u->lw, h = 1 << u->lh;
  if (u->isp != (dir ? 1 : 2)) return dir ? h : w;
  const int parts = ((w == 4 && h == 8) || (w == 8 && h == 4)) ? 2 : 4;
  return (dir ? h : w) / parts;
}
static void dbk_picture(const unit_t *um0, const unit_t *um1, int uw, int uh, int16_t *const rec[3], const int stride[3], int qp, const int qp_c[2], int bd, int chroma,
                        int beta_offset_div2, int tc_offset_div2)
{
  for (int dir = 0; dir < 2; dir++) {                 /* 0: vertical edges, 1: horizontal edges */
    for (int uy = 0; uy < uh; uy++) for (int ux = 0; ux < uw; ux++) {
      const unit_t *u = &um0[uy * uw + ux];
      const int x = ux << 2, y = uy << 2, st = stride[0];
      if (!u->valid) continue;
      /* transform edges (xDeblockCU 306-317): the CU border and, in an ISP CU, the borders between its sub-partitions that lie on the
       * 4-sample grid; the filter lengths follow the TRANSFORM sizes on either side (xSetMaxFilterLengthPQFromTransformSizes 474-575),
       * i.e. the sub-partition size where the CU is split across the edge direction */
      if (dir == 0 && x > 0) {
        const int tq = dbk_tu_size(u, 0), off = x - u->x;
        if (off == 0) {
          const unit_t *p = &um0[uy * uw + ux - 1];
          orc_deblock_luma_segment(rec[0] + y * st + x, 1, st, dbk_tu_size(p, 0), tq, 0, qp, bd, beta_offset_div2, tc_offset_div2);
        } else if (u->isp == 2 && off % tq == 0)
          orc_deblock_luma_segment(rec[0] + y * st + x, 1, st, tq, tq, 0, qp, bd, beta_offset_div2, tc_offset_div2);
      }
      if (dir == 1 && y > 0) {
        const int tq = dbk_tu_size(u, 1), off = y - u->y;
        if (off == 0) {
          const unit_t *p = &um0[(uy - 1) * uw + ux];
          orc_deblock_luma_segment(rec[0] + y * st + x, st, 1, dbk_tu_size(p, 1), tq, (y & 127) == 0, qp, bd, beta_offset_div2, tc_offset_div2);
        } else if (u->isp == 1 && off % tq == 0)
          orc_deblock_luma_segment(rec[0] + y * st + x, st, 1, tq, tq, 0, qp, bd, beta_offset_div2, tc_offset_div2);
      }
    }
    if (!chroma) continue;
    for (int uy = 0; uy < uh; uy++) for (int ux = 0; ux < uw; ux++) {
      const unit_t *u = &um1[uy * uw + ux];
      const int cx = ux << 1, cy = uy << 1;          /* chroma samples of this unit: 2x2 */
      if (!u->valid) continue;
      for (int k = 0; k < 2; k++) {
        const int st = stride[k + 1], qpc = qp_c[k] < 0 ? 0 : qp_c[k] > 63 ? 63 : qp_c[k];
        if (dir == 0 && u->x == cx && cx > 0 && (cx & 7) == 0) {
          const unit_t *p = &um1[uy * uw + ux - 1];
          orc_deblock_chroma_segment(rec[k + 1] + cy * st + cx, 1, st, 1 << p->lw, 1 << u->lw, 0, qpc, bd, beta_offset_div2, tc_offset_div2);
        }
        if (dir == 1 && u->y == cy && cy > 0 && (cy & 7) == 0) {
          const unit_t *p = &um1[(uy - 1) * uw + ux];
          orc_deblock_chroma_segment(rec[k + 1] + cy * st + cx, st, 1, 1 << p->lh, 1 << u->lh, (cy & 63) == 0, qpc, bd, beta_offset_div2, tc_offset_div2);
        }
      }
    }
  }
}
int orc_deblock_frame(orc_enc *e, int beta_offset_div2, int tc_offset_div2)
{
  dbk_picture(e->um[0], e->um[1], e->uw, e->uh, e->rec, e->stride, e->sl.qp, e->sl.qp_c, e->cfg.bit_depth, e->cfg.chroma, beta_offset_div2, tc_offset_div2);
  return 0;
}
/* the same filter on a picture given as a CU table (rows of {ch, x, y, w, h, ispMode}, luma samples) and its planes (4:2:0, stride = plane width), so that the pin against
 * the reference's LoopFilter can cover CU tables no search produced (tests/golden/make_golden.py deblock: forced ISP splits) */
int orc_deblock_table(int w, int h, int bd, int qp, int qp_cb, int qp_cr, const int *rows, int nrows, int16_t *y, int16_t *cb, int16_t *cr)
{
  const int uw = (w + 3) >> 2, uh = (h + 3) >> 2;
  unit_t *um[2] = { calloc((size_t) uw * uh, sizeof(unit_t)), calloc((size_t) uw * uh, sizeof(unit_t)) };
  if (!um[0] || !um[1]) { free(um[0]); free(um[1]); return -1; }
  for (int i = 0; i < nrows; i++) {
    const int *r = rows + 6 * i, ch = r[0], sh = ch ? 1 : 0;
    for (int v = r[2] >> 2; v < (r[2] + r[4]) >> 2 && v < uh; v++) for (int u = r[1] >> 2; u < (r[1] + r[3]) >> 2 && u < uw; u++) {
      unit_t *t = &um[ch][v * uw + u];
      t->valid = 1; t->x = (int16_t) (r[1] >> sh); t->y = (int16_t) (r[2] >> sh); t->lw = (uint8_t) ilog2(r[3] >> sh); t->lh = (uint8_t) ilog2(r[4] >> sh); t->isp = (uint8_t) r[5];
    }
  }
  int16_t *rec[3] = { y, cb, cr };
  const int stride[3] = { w, w >> 1, w >> 1 }, qpc[2] = { qp_cb, qp_cr };
  dbk_picture(um[0], um[1], uw, uh, rec, stride, qp, qpc, bd, 1, 0, 0);
  free(um[0]); free(um[1]);
  return 0;
}

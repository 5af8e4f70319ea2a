/*
 * vvc_oracle.h — CPU restatement ("oracle") of the VTM 6.1 intra CU-partition RDO hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing in the product library links, includes or calls this code; it is
 * used by tests/, by __graft_entry__.smoke() as the checker and by bench.py's cpu_baseline leg.
 *
 * Pinning status (see DESIGN.md "Oracle"):
 *   - leaf operators (transforms, SAD/SATD/SSE, CABAC probability model, calcRdCost, reference-sample
 *     fill/filter, planar/DC/angular/PDPC prediction, MPM list, split contexts, scan order): pinned
 *     bit-exactly against the real reference code compiled into oracle/_ref (tests/golden/).
 *   - recursion-level decisions (EncCu / EncModeCtrl / IntraSearch / CABACWriter restatements):
 *     PARITY UNPINNED — the reference encoder as a whole cannot be built here without stand-in headers
 *     (EL/EncCu.cpp:59 and EL/CABACWriter.h:44 include <opencv2/opencv.hpp>, absent from the image).
 *
 * Every function cites the reference file:line it follows (paths relative to
 * /root/reference/VVC_project/source/Lib; CL = CommonLib, EL = EncoderLib).
 */
#ifndef VVC_ORACLE_H
#define VVC_ORACLE_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

#define ORC_MAX_CU 128
#define ORC_NUM_LUMA_MODE 67
#define ORC_PLANAR 0
#define ORC_DC 1
#define ORC_HOR 18
#define ORC_DIA 34
#define ORC_VER 50
#define ORC_VDIA 66
#define ORC_DM_CHROMA 70
#define ORC_MAX_DOUBLE 1.7e+308

/* tool flags (bit mask) — which parts of BIN/encoder_intra.cfg are enabled.  Round 1 restates the
 * subset "P0"; the remaining flags are accepted by the C-ABI and rejected with an error until built. */
enum {
  ORC_TOOL_MRL   = 1 << 0,  /* multi-reference-line (compile-time always on in the reference) */
  ORC_TOOL_MIP   = 1 << 1, ORC_TOOL_ISP = 1 << 2, ORC_TOOL_LFNST = 1 << 3, ORC_TOOL_MTS = 1 << 4,
  ORC_TOOL_TS    = 1 << 5, ORC_TOOL_DEPQUANT = 1 << 6, ORC_TOOL_RDOQ = 1 << 7, ORC_TOOL_CCLM = 1 << 8,
  ORC_TOOL_JCCR  = 1 << 9, ORC_TOOL_LMCS = 1 << 10 /* LMCS allowed: the slice's model (orc_slice) decides */, ORC_TOOL_CU_REUSE = 1 << 11,
  ORC_TOOL_FAST  = 1 << 12, /* the fork's FAST_ALGORITHM: features + forest decide the one partition mode a luma node tries (needs orc_set_forest) */
  ORC_TOOL_WPP   = 1 << 13  /* cfg WaveFrontSynchro 1 (entropy_coding_sync): every CTU row of a tile starts from the contexts left by the first CTU of the row above, the
                             * CTU above-right is not available (EL/EncSlice.cpp:1648-1661,1801-1805; CL/CodingStructure.cpp:1634-1657) */
};

typedef struct {
  int pic_w, pic_h;        /* luma samples */
  int bit_depth;           /* 8 or 10 */
  int ctu_size;            /* 128 (cfg CTUSize) */
  int min_qt[2];           /* MinQTLumaISlice 8, MinQTChromaISlice 4 */
  int max_bt_depth[2];     /* MaxBTDepthISliceL/C 3,3 */
  int max_bt_size[2];      /* MAX_BT_SIZE 32, MAX_BT_SIZE_C 64 (CL/CommonDef.h:427-437) */
  int max_tt_size[2];      /* MAX_TT_SIZE 32, MAX_TT_SIZE_C 32 */
  int dual_tree;           /* DualITree 1 */
  int tile_cols, tile_rows;/* uniform tile grid in CTUs (1,1 = one stream per frame) */
  uint32_t tools;          /* ORC_TOOL_* */
  int chroma;              /* 1: code the chroma tree as well */
} orc_cfg;

typedef struct {
  int    qp;               /* slice QP (luma) */
  int    qp_c[2];          /* mapped chroma QP for Cb, Cr */
  double lambda;           /* RdCost lambda (EL/EncSlice.cpp:754-845) */
  double dist_weight[2];   /* chroma distortion weights (EL/EncSlice.cpp:125) */
  /* LMCS as the slice / its APS signal it (slice_lmcs_enabled_flag, slice_chroma_residual_scale_flag, lmcs_min_bin_idx, LmcsMaxBinIdx, the signed codeword deltas):
   * the analysis that chooses the model is the caller's (EL/EncReshape.cpp); for an intra slice the search itself sees LMCS through the forward-mapped original
   * luma (EL/EncGOP.cpp:1689-1695), the mapped-domain reconstruction and the scaling of the chroma residuals */
  int    lmcs_enable, lmcs_chroma_adj, lmcs_min_bin, lmcs_max_bin;
  int    lmcs_delta_cw[16];
} orc_slice;

/* one final coding unit, ≙ the fields D_BLOCK_STATISTICS_CODED prints (CL/dtrace_blockstatistics.cpp) */
typedef struct {
  int16_t  x, y, w, h;     /* luma samples for ch_type 0, chroma samples for ch_type 1 */
  uint8_t  ch_type;        /* 0 luma tree, 1 chroma tree */
  uint8_t  qt_depth, bt_depth, mt_depth, depth;
  uint8_t  intra_dir;      /* luma mode, or chroma mode (70 = DM) */
  uint8_t  mrl_idx;        /* multiRefIdx 0/1/3 */
  uint8_t  cbf;            /* bit0 Y, bit1 Cb, bit2 Cr */
  uint8_t  mts_idx;        /* luma TU: 0 DCT2xDCT2, 2..5 explicit MTS (tu.mtsIdx) */
  uint8_t  mip_flag;       /* luma CU: cu.mipFlag; intra_dir is then the MIP mode */
  uint8_t  lfnst_idx;      /* cu.lfnstIdx (0..2) */
  uint8_t  joint_cb_cr;    /* chroma CU: tu.jointCbCr (0 separate, 1..3 = the cbf mask of the joint residual) */
  uint8_t  isp_mode;       /* luma CU: cu.ispMode (0 none, 1 horizontal, 2 vertical sub-partitions) */
  uint8_t  tu_cbf;         /* luma CU with ISP: cbf of each sub-partition, bit k = TU k (cbf bit 0 = any) */
  uint64_t split_series;
} orc_cu;

typedef struct {
  uint64_t dist;           /* Σ SSE of the CTU (luma tree + weighted chroma tree) */
  uint64_t frac_bits;      /* Σ estimated bits, 2^-15 units */
  double   cost;           /* luma-tree root cost + chroma-tree root cost */
  int      n_cu;
} orc_ctu_result;

typedef struct orc_enc orc_enc;

orc_enc *orc_create(const orc_cfg *cfg);
void     orc_destroy(orc_enc *e);
int      orc_set_slice(orc_enc *e, const orc_slice *s);
/* planes: 3 pointers (Y,U,V) of bytes_per_sample 1 or 2, strides in samples */
int      orc_load_frame(orc_enc *e, const void *const org[3], const int stride[3], int bytes_per_sample);
/* compress every CTU of the loaded frame (tiles in raster order, CTUs in raster order inside a tile) */
int      orc_compress_frame(orc_enc *e, orc_ctu_result *res /* [n_ctus] */, orc_cu *cus, int max_cus, int *n_cus);
int      orc_compress_tiles(orc_enc *e, int tile_first, int tile_count, orc_ctu_result *res, orc_cu *cus, int max_cus, int *n_cus);   /* a range of tiles only (test runs spread tiles over processes) */
int      orc_jccr_sign(const int16_t *cb, const int16_t *cr, int stride, int w, int h);
int      orc_get_reco(orc_enc *e, void *const reco[3], const int stride[3], int bytes_per_sample);
/* LMCS tables built from the slice's model (Reshape::constructReshaper, CL/Reshape.cpp:297-333): forward / inverse LUT (1 << bit_depth entries), 17 pivots, 16 chroma scales */
int      orc_lmcs_tables(orc_enc *e, int16_t *fwd, int16_t *inv, int *pivot, int *cadj);
/* inverse mapping of the reconstructed luma (what EncGOP does after the slice is coded, EL/EncGOP.cpp:2576-2594; orc_get_reco returns the mapped-domain samples the search left) */
int      orc_lmcs_inverse_reco(orc_enc *e);
const char *orc_last_error(void);
/* work counters for the bench's diagnostic model */
int      orc_arith_encode(int qp, const int32_t *ops, int nops, uint8_t *out, int cap);
long     orc_write_tiles(orc_enc *e, uint8_t *buf, long cap, int *sizes);   /* slice_data payload per tile (after orc_compress_frame); with ORC_TOOL_WPP one sub-stream per CTU row of every tile */
int      orc_get_levels(orc_enc *e, int16_t *const lev[3]);
/* FAST_ALGORITHM (orc_fast.c): the flattened random forest (sklearn tree_ arrays; value = n_nodes x n_classes leaf distributions;
 * classes = label of each column, 0 no split, 1 QT, 2 BT_H, 3 BT_V, 4 TT_H, 5 TT_V like BIN/TEST.py's return value) */
int      orc_set_forest(orc_enc *e, int n_trees, int n_nodes, int n_classes, const int32_t *root, const int32_t *feature, const double *threshold,
                        const int32_t *left, const int32_t *right, const double *value, const int32_t *classes);
/* counterpart of GET_TRAINING_SET: every luma node that qualifies for the classifier appends 28 ints to rows: the 26 features, the
 * complexity class (0 simple, 1 fuzzy, 2 complex) and the partition the search chose there (0..5).  Returns rows written so far. */
int      orc_set_training_dump(orc_enc *e, int32_t *rows, int cap_rows);
int      orc_training_rows(const orc_enc *e);
int      orc_fast_features(const int16_t *org, int stride, int w, int h, int feat[26]);   /* test hook: block features 4..11, 21..25 */
int      orc_forest_predict_rows(orc_enc *e, const int32_t *rows, int n, int32_t *out);   /* test hook: forest on n rows of 26 ints */
/* deblocking filter on the coded picture (CL/LoopFilter.cpp; after orc_compress_frame, before orc_get_reco): cfg LoopFilterBetaOffset_div2 / TcOffset_div2 */
int      orc_deblock_frame(orc_enc *e, int beta_offset_div2, int tc_offset_div2);
/* the same filter on a CU table {ch, x, y, w, h, ispMode} (luma samples) and 4:2:0 planes with stride = plane width; qp_cb / qp_cr = mapped chroma QPs */
int      orc_deblock_table(int w, int h, int bd, int qp, int qp_cb, int qp_cr, const int *rows, int nrows, int16_t *y, int16_t *cb, int16_t *cr);
/* the encoder's SAO statistics (EL/EncSampleAdaptiveOffset.cpp getStatistics, SAOLcuBoundary 0): out [ctu][component][type 0..4][count | diff][32] int64.  PARITY UNPINNED
 * for the region rules (see orc_sao.c) */
int      orc_sao_statistics(int w, int h, int bit_depth, int tile_cols, int tile_rows, int lf_across_tiles, const int16_t *const org[3], const int16_t *const rec[3], int64_t *out);
/* adaptive loop filter with given parameter sets on 4:2:0 planes with stride = plane width (CL/AdaptiveLoopFilter.cpp ALFProcess; orc_alf.c) */
typedef struct {                      /* what an ALF parameter set carries (AlfParam, CL/AlfParameters.h) */
  int32_t num_luma_filters; uint8_t class_to_filter[25]; uint8_t nonlinear_luma; int16_t luma_coeff[25][12]; uint8_t luma_clip_idx[25][12];
  int32_t num_chroma_alt; uint8_t nonlinear_chroma[8]; int16_t chroma_coeff[8][6]; uint8_t chroma_clip_idx[8][6];
} orc_alf_aps;
typedef struct { uint8_t flag[3]; int8_t set; uint8_t alt[2]; } orc_alf_ctu;      /* per CTU: enable Y / Cb / Cr, luma filter set (0..15 fixed, 16 + k: k-th set of the slice), chroma alternatives */
int      orc_alf_clip_value(int chroma, int bit_depth, int idx);
void     orc_alf_reconstruct(const orc_alf_aps *a, int bit_depth, int16_t *luma_coeff /* [25][13] */, int16_t *luma_clip, int16_t *chroma_coeff /* [n_alt][7] */, int16_t *chroma_clip);
int      orc_alf_picture(int w, int h, int bit_depth, int n_sets, const int16_t *luma_coeff, const int16_t *luma_clip, int n_alt, const int16_t *chroma_coeff, const int16_t *chroma_clip,
                         const orc_alf_ctu *ctu, int16_t *y, int16_t *cb, int16_t *cr, uint8_t *cls_out);
/* sample adaptive offset with given parameters on 4:2:0 planes with stride = plane width (CL/SampleAdaptiveOffset.cpp SAOProcess; orc_sao.c) */
typedef struct { int8_t mode, type, band, off[4]; } orc_sao_param;      /* per CTU and component: mode 0 off / 1 new / 2 merge; type: new 0..3 edge class, 4 band; merge 0 left, 1 above */
int      orc_sao_picture(int w, int h, int bit_depth, int tile_cols, int tile_rows, int lf_across_tiles, int log2_offset_scale, const orc_sao_param *prm, int16_t *y, int16_t *cb, int16_t *cr);
/* the RD half of the SAO parameter decision from those statistics (decideBlkParams: new / merge per CTU against the SAO context models).  PARITY UNPINNED (see orc_sao.c) */
int      orc_sao_decide(int w, int h, int bit_depth, int tile_cols, int tile_rows, int slice_qp, const double *lambda, int log2_offset_scale, const int64_t *stats, orc_sao_param *prm);
void     orc_get_counters(orc_enc *e, uint64_t out[4]); /* satd candidates, rd candidates, rd pixels, nodes */

/* ---------------- leaf operators (individually testable; used by the golden-vector tests) -------- */
/* CL/TrQuant_EMT.cpp 1-D kernels restated as plain matrix products: tr 0=DCT2 1=DCT8 2=DST7 */
void orc_fwd_1d(int tr, int n, const int *src, int *dst, int shift, int line, int skip1, int skip2);
void orc_inv_1d(int tr, int n, const int *src, int *dst, int shift, int line, int skip1, int skip2, int cmin, int cmax);
/* CL/TrQuant.cpp:835-992 (xT / xIT), DCT2 both directions */
void orc_fwd_2d(const int16_t *resi, int stride, int w, int h, int bit_depth, int *coef);
void orc_inv_2d(const int *coef, int w, int h, int bit_depth, int16_t *resi, int stride);
/* the same with an explicit-MTS index (0 DCT2, 2 DST7xDST7, 3 DCT8 hor, 4 DCT8 ver, 5 DCT8xDCT8; getTrTypes 817-830); 32-point MTS keeps 16 */
void orc_fwd_2d_mts(const int16_t *resi, int stride, int w, int h, int bit_depth, int mts_idx, int *coef);
void orc_inv_2d_mts(const int *coef, int w, int h, int bit_depth, int mts_idx, int16_t *resi, int stride);
/* CL/TrQuant.cpp:1049-1124: which of {DCT2, mts 2, 3, 4, 5} stay in the RD loop of a luma TU */
void orc_mts_prune(const int16_t *resi, int stride, int w, int h, int bit_depth, int max_cand, int test[5]);
/* CL/Quant.cpp:994-1089 (plain quant, I-slice offset 171) and 423-549 (dequant) */
int  orc_quant(const int *coef, int w, int h, int bit_depth, int qp, int16_t *level);
void orc_dequant(const int16_t *level, int w, int h, int bit_depth, int qp, int *coef);
/* LFNST (CL/TrQuant.cpp:241-560): see orc_leaf.c.  mode = orc_lfnst_mode(final intra mode, block w, h) */
int  orc_quant_lfnst(const int *coef, int w, int h, int bit_depth, int qp, int16_t *level);
int  orc_lfnst_mode(int dir, int w, int h);
void orc_lfnst_keep(int *coef, int w, int h);
void orc_fwd_lfnst(int *coef, int w, int h, int mode, int lfnst_idx);
void orc_inv_lfnst(int *coef, int w, int h, int mode, int lfnst_idx);
/* transform + quantisation + reconstruction residual of one block with LFNST: the path of code_tu_block (test entry point) */
int  orc_trquant_lfnst(const uint16_t *s0, const uint16_t *s1, const int16_t *resi, int w, int h, int comp, int cbf_cb, int bit_depth, int qp, double lambda,
                       int dep_quant, int dir, int lfnst_idx, int16_t *level, int16_t *resi_out);
/* CL/DepQuant.cpp: dependent quantisation of one block from the estimator's contexts (s0, s1) at the time of the call; comp 0 Y / 1 Cb / 2 Cr;
 * cbf_ctx = flat context index of the block's cbf flag (-1: inferred); qp as for orc_quant; lambda = the quantiser's lambda for the component;
 * zo = explicit MTS (mts_idx > 1); lfnst = cu.lfnstIdx.  Returns absSum.  orc_dequant_dq: Quantizer::dequantBlock (741-810). */
int  orc_depquant(const uint16_t *s0, const uint16_t *s1, const int *coef, int w, int h, int comp, int cbf_ctx, int bit_depth, int qp, double lambda,
                  int zo, int lfnst, int16_t *level);
void orc_dequant_dq(const int16_t *level, int w, int h, int bit_depth, int qp, int *coef);
void orc_depquant_consts(int w, int h, int bit_depth, int qp, double lambda, int64_t out[9]);   /* Quantizer::initQuantBlock 694-739 */
/* CL/RdCost.cpp xGetSAD / xGetHADs / xGetSSE */
uint64_t orc_sad(const int16_t *a, int sa, const int16_t *b, int sb, int w, int h);
uint64_t orc_satd(const int16_t *a, int sa, const int16_t *b, int sb, int w, int h);
uint64_t orc_sse(const int16_t *a, int sa, const int16_t *b, int sb, int w, int h);
/* CL/RdCost.cpp:63-88 */
double orc_calc_rd_cost(double lambda, uint64_t frac_bits, uint64_t dist);
/* CL/Contexts.cpp:135-151,1818-1833: context init for an I slice */
void orc_ctx_init(int qp, uint16_t *s0, uint16_t *s1);
uint64_t orc_ctx_code_bins(uint16_t *s0, uint16_t *s1, int ctx_id, const uint8_t *bins, int n);
/* CL/IntraPrediction.cpp:1215-1522 + 426-935: build reference samples from a reco plane with an
 * availability map (one byte per 4x4 luma unit / per 2x2.. see .c) and predict one mode.
 * ref buffers hold (2h+1+mrl) rows x (2w+1+mrl) stride like the reference's m_piYuvExt. */
void orc_fill_ref_samples(const int16_t *reco, int stride, int pic_w, int pic_h, const uint8_t *avail4, int avail_stride,
                          int unit_log2, int tag, int x, int y, int w, int h, int mrl, int bit_depth, int16_t *ref_unf);
void orc_filter_ref_samples(const int16_t *ref_unf, int16_t *ref_flt, int w, int h, int mrl);
void orc_pred_intra(const int16_t *ref_unf, const int16_t *ref_flt, int w, int h, int is_luma, int mode, int mrl,
                    int bit_depth, int16_t *pred, int pred_stride);
/* CL/MatrixIntraPrediction.cpp (JVET_O0925 form): matrix-based intra prediction of a w x h luma block from its unfiltered line-0 reference
 * samples top[w] / left[h]; mode 0 .. orc_mip_num_modes(w, h) - 1 (the upper half of the modes uses the transposed input) */
int  orc_mip_num_modes(int w, int h);
void orc_pred_mip(const int16_t *top, const int16_t *left, int w, int h, int mode, int bit_depth, int16_t *pred);
/* CL/UnitTools.cpp:508-640 */
void orc_get_mpms(int left_dir, int above_dir, unsigned mpm[6]);
/* scan order (CL/Rom.cpp:87-370): fills idx[] with raster positions in coding order, returns count */
int  orc_scan_order(int w, int h, uint16_t *idx);
/* EL/CABACWriter.cpp:3773-3883 (+4102, 4164): estimated bits of residual_coding for one block; updates ctx */
uint64_t orc_residual_bits(uint16_t *s0, uint16_t *s1, const int16_t *level, int w, int h, int is_chroma);

/* test hooks: partition legality / split contexts / MPM list in a picture with given coded neighbour CUs */
int orc_test_partition(int pic_w, int pic_h, int ch, int ctux, int ctuy, const int *nb, int n_nb, const int *path, int npath,
                       int *can, unsigned *ctx, int *implicit, int *area);
int orc_test_mpm(int pic_w, int pic_h, const int *nb, int n_nb, int x, int y, int w, int h, unsigned *mpm);

#ifdef __cplusplus
}
#endif
#endif

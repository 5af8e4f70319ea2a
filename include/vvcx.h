/*
 * vvcx.h — C-ABI of the MI355X-native VVC intra CU-partition RDO path.
 *
 * Drop-in boundary: the reference has no FFI; the functional seam is the member call
 *     void EncCu::compressCtu(CodingStructure& cs, const UnitArea& area, unsigned ctuRsAddr,
 *                             const int prevQP[], const int currQP[]);          // EL/EncCu.h:177
 * invoked once per CTU from EncSlice::encodeCtus (EL/EncSlice.cpp:1768), plus its set-up contract
 * create()/init()/destroy() (EL/EncCu.h:167-174) and the per-slice state the caller pushes before
 * the CTU loop (lambda, QP, contexts: EL/EncSlice.cpp:1568-1572,1640-1661).  Each entry point below
 * names the reference interface it replaces.  Plain C structs, no exceptions across the ABI: every
 * function returns 0 on success or a negative vvcx_status; vvcx_last_error() gives the text.
 * A handle is not thread-safe (one per GPU / host thread), like one EncCu instance per worker
 * (EL/EncLib.cpp:115-125).
 *
 * Paths in comments are relative to /root/reference/VVC_project/source/Lib (EL = EncoderLib, CL = CommonLib).
 */
#ifndef VVCX_H
#define VVCX_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

typedef enum {
  VVCX_OK = 0, VVCX_ERR_ARG = -1, VVCX_ERR_UNSUPPORTED = -2, VVCX_ERR_DEVICE = -3, VVCX_ERR_STATE = -4, VVCX_ERR_NO_ENCODING = -5
} vvcx_status;

/* encoder tools of BIN/encoder_intra.cfg that reach the hot path.  Requesting a tool that
 * is not built yet fails with VVCX_ERR_UNSUPPORTED instead of silently changing the result. */
enum {
  VVCX_TOOL_MRL = 1 << 0, VVCX_TOOL_MIP = 1 << 1, VVCX_TOOL_ISP = 1 << 2, VVCX_TOOL_LFNST = 1 << 3, VVCX_TOOL_MTS = 1 << 4,
  VVCX_TOOL_TS = 1 << 5, VVCX_TOOL_DEPQUANT = 1 << 6, VVCX_TOOL_RDOQ = 1 << 7, VVCX_TOOL_CCLM = 1 << 8,
  VVCX_TOOL_JCCR = 1 << 9, VVCX_TOOL_LMCS = 1 << 10, VVCX_TOOL_CU_REUSE = 1 << 11,
  VVCX_TOOL_FAST = 1 << 12,     /* the fork's FAST_ALGORITHM (CL/TypeDef.h:54-56): features + forest pick the one partition mode of a luma node */
  VVCX_TOOL_WPP = 1 << 13       /* cfg WaveFrontSynchro 1 (off in encoder_intra.cfg): a CTU row of a tile starts from the contexts behind the first CTU of the row above
                                 * (EL/EncSlice.cpp:1648-1661, 1801-1805), the CTU above-right is unavailable (CL/CodingStructure.cpp:1634-1657), one sub-stream per CTU row.
                                 * The rows of a tile become streams of their own that run one CTU behind the row above */
};

/* ≙ the EncCfg/SPS fields EncCu::create/init read (EL/EncCu.cpp:167-236, CL/Slice.h PreCalcValues 2229-2275) */
typedef struct {
  int32_t pic_w, pic_h;          /* luma samples, multiples of 8 (APP/EncAppCfg.cpp:2709) */
  int32_t bit_depth;             /* InternalBitDepth: 8 (uint8 planes) or 10 (uint16 planes) */
  int32_t ctu_size;              /* CTUSize 128 */
  int32_t min_qt[2];             /* MinQTLumaISlice, MinQTChromaISlice */
  int32_t max_bt_depth[2];       /* MaxBTDepthISliceL / C */
  int32_t max_bt_size[2];        /* MAX_BT_SIZE / MAX_BT_SIZE_C (CL/CommonDef.h:427,437) */
  int32_t max_tt_size[2];        /* MAX_TT_SIZE / MAX_TT_SIZE_C */
  int32_t dual_tree;             /* DualITree */
  int32_t tile_cols, tile_rows;  /* uniform tile grid; every tile is an independent CTU stream */
  uint32_t tools;                /* VVCX_TOOL_* */
  int32_t chroma;                /* 1 = code the chroma tree too (4:2:0) */
  int32_t max_frames;            /* frames resident per batch */
  int32_t device;                /* HIP device ordinal */
  int32_t emit_payload;          /* 1 = also arithmetic-code the final CTU syntax: slice_data() payload per tile (vvcx_get_payload) */
} vvcx_cfg;

/* ≙ per-slice state pushed by EncSlice before the CTU loop: setUpLambda (EL/EncSlice.cpp:107-149),
 * slice QP, mapped chroma QPs; CABAC contexts are (re)initialised at each tile start (1640-1647) */
typedef struct {
  int32_t qp;
  int32_t qp_c[2];
  double  lambda;
  double  dist_weight[2];
  /* LMCS (VVCX_TOOL_LMCS): the model the slice's LMCS APS carries (SliceReshapeInfo, CL/Slice.h; chosen per picture by EncReshape::preAnalyzerLMCS, which is the
   * caller's job like lambda and QP).  The handle builds the forward / inverse LUTs and the chroma scale table from it the way Reshape::constructReshaper does
   * (CL/Reshape.cpp:297-333).  lmcs_enable 0 (the analysis switched the tool off for this picture): the other fields are ignored. */
  int32_t lmcs_enable;           /* slice_lmcs_enabled_flag */
  int32_t lmcs_chroma_adj;       /* slice_chroma_residual_scale_flag */
  int32_t lmcs_min_bin, lmcs_max_bin;   /* reshaperModelMinBinIdx / MaxBinIdx */
  int32_t lmcs_delta_cw[16];     /* reshaperModelBinCWDelta: signed deviation of each bin's code words from (1 << bit_depth) / 16 */
} vvcx_slice;

/* one picture of a batch: DEVICE pointers to planar 4:2:0 samples (uint8 for 8 bit, uint16 for 10 bit),
 * strides in samples.  ≙ cs.picture->getOrigBuf()/getRecoBuf() (CL/Picture.h); reco is written in place */
typedef struct {
  const void *org[3];
  void       *reco[3];
  int32_t     stride[3];
} vvcx_frame;

/* ≙ arguments of compressCtu: which CTU of which bound picture */
typedef struct { int32_t frame; int32_t ctu_rs_addr; } vvcx_ctu_task;

/* ≙ bestCS->dist / fracBits / cost after compressCtu (CL/CodingStructure.h:178-184), luma + chroma tree */
typedef struct {
  uint64_t dist;
  uint64_t frac_bits;            /* 2^-15 bit units (CL/CommonDef.h:354) */
  double   cost;
  int32_t  n_cu;
} vvcx_ctu_result;

/* ≙ the CodingUnit/PredictionUnit/TransformUnit fields of the final CUs appended to cs (CL/Unit.h:292-470) */
typedef struct {
  int16_t  x, y, w, h;           /* luma samples for ch_type 0, chroma samples for ch_type 1 */
  uint8_t  ch_type;
  uint8_t  qt_depth, bt_depth, mt_depth, depth;
  uint8_t  intra_dir;            /* luma mode 0..66; chroma mode (70 = DM) */
  uint8_t  mrl_idx;              /* multiRefIdx */
  uint8_t  cbf;                  /* bit0 Y, bit1 Cb, bit2 Cr */
  uint8_t  mts_idx;              /* tu.mtsIdx of the luma TU: 0 DCT2xDCT2, 1 transform skip (VVCX_TOOL_TS), 2..5 explicit MTS (VVCX_TOOL_MTS) */
  uint8_t  mip_flag;             /* cu.mipFlag of a luma CU (VVCX_TOOL_MIP): intra_dir is then the MIP mode, mrl_idx 0 */
  uint8_t  lfnst_idx;            /* cu.lfnstIdx of the CU (VVCX_TOOL_LFNST): 0 none, 1 / 2 the kernel of the set */
  uint8_t  joint_cb_cr;          /* tu.jointCbCr of a chroma CU (VVCX_TOOL_JCCR): 0 separate residuals, 1..3 the cbf mask of the joint residual */
  uint8_t  isp_mode;             /* cu.ispMode of a luma CU (VVCX_TOOL_ISP): 0 none, 1 horizontal split (sub-partitions stacked), 2 vertical split */
  uint8_t  tu_cbf;               /* ISP: bit k = cbf of sub-partition k (cbf bit 0 is then their OR) */
  uint64_t split_series;         /* CU::splitSeries, 5 bits per depth */
} vvcx_cu;

/* one transform unit ≙ TransformUnit (CL/Unit.h:412-470): area, depth, mtsIdx, cbf[], jointCbCr and where its coefficients are */
typedef struct {
  int32_t  cu_index;             /* index of its CU in vvcx_get_cus order */
  int16_t  x, y, w, h;           /* like the CU: luma samples for ch_type 0, chroma samples for ch_type 1 */
  uint8_t  ch_type, depth;       /* transform depth (0: the TU is the CU; 1: a sub-partition of an ISP CU) */
  uint8_t  mts_idx;              /* tu.mtsIdx (luma) */
  uint8_t  joint_cb_cr;          /* tu.jointCbCr (chroma) */
  uint8_t  cbf[3];               /* Y, Cb, Cr */
  uint8_t  pad_;
  int32_t  coeff_offset[3];      /* sample offset of the block's levels in the level plane of component c, -1 if absent */
  int32_t  coeff_stride[3];
} vvcx_tu;

typedef struct vvcx_handle vvcx_handle;

/* ≙ EncCu::create(EncCfg*) + init(EncLib*, const SPS&)  (EL/EncCu.h:167-171) */
int  vvcx_create(const vvcx_cfg *cfg, vvcx_handle **h);
/* ≙ EncCu::destroy() (EL/EncCu.h:173) */
void vvcx_destroy(vvcx_handle *h);
/* ≙ EncSlice::setUpLambda + slice QP (EL/EncSlice.cpp:107-149, 1568-1572).  Precedes vvcx_bind_frames: binding prepares the pictures for the slice (start contexts of
 * its QP, forward-mapped luma and device LUTs of its LMCS model).  A later call whose QP or LMCS model differs UNBINDS the pictures (the next search call then reports
 * VVCX_ERR_STATE until they are bound again); one that changes only lambda / the distortion weights leaves them bound. */
int  vvcx_set_slice(vvcx_handle *h, const vvcx_slice *s);
/* ---- slice-level inputs: what the reference derives before the CTU loop, as pure host functions ---- */
typedef struct {
  int32_t qp;                    /* slice QP (cfg QP) */
  int32_t bit_depth;             /* 8 or 10 */
  int32_t n_pts;                 /* chroma QP mapping pivots: cfg QpInValCb / QpOutValCb (SameCQPTablesForAllChroma 1), 1..8 */
  int32_t qp_in[8], qp_out[8];   /* BIN/encoder_intra.cfg:86-87: "2 31 43" -> "2 32 41" */
  int32_t cb_qp_offset, cr_qp_offset;   /* cfg CbQpOffset / CrQpOffset */
  int32_t gop_size;              /* cfg GOPSize (1 for All-Intra): lambda scale 1 - clip(0, 0.5, 0.05 * (GOPSize - 1)) */
  int32_t dep_quant;             /* cfg DepQuant: lambda * 2^(0.25/3) */
} vvcx_slice_cfg;
/* ≙ ChromaQpMappingTable::derivedChromaQPMappingTables (CL/Slice.cpp:1540-1581): table[q + 6*(bit_depth-8)] = mapped chroma QP of
 * luma-scale QP q, q = -6*(bit_depth-8) .. 63 */
int  vvcx_chroma_qp_table(int bit_depth, int n_pts, const int32_t *qp_in, const int32_t *qp_out, int32_t *table);
/* ≙ EncSlice::calculateLambda for an I slice (EL/EncSlice.cpp:752-845: QPFactor 0.57 * scale, 2^((QP + 6*(bd-8) - 12)/3), DepQuant factor)
 * + the chroma QPs QpParam uses (CL/Quant.cpp:68-106) + EncSlice::setUpLambda's distortion weights 2^((QP - QPc)/3) (107-149) */
int  vvcx_derive_slice(const vvcx_slice_cfg *cfg, vvcx_slice *out);

/* ≙ the model the fork's classifier asks at every qualifying luma node: Py_Initialize + joblib.load("Partition_32.pkl").predict(x)
 * per call (EL/EncCu.cpp:1140-1166, BIN/TEST.py:7-25), replaced by one upload of the forest's flattened sklearn tree arrays (host
 * pointers): root[n_trees]; feature / left / right [n_nodes] (children -1 at leaves, children after their parent); threshold[n_nodes]
 * (go left when float(x[feature]) <= threshold); value[n_nodes][n_classes] class distributions; classes[n_classes] = label of each
 * column (0 do not split, 1 QT, 2 BT_H, 3 BT_V, 4 TT_H, 5 TT_V).  Needed before the first CTU when VVCX_TOOL_FAST is set. */
int  vvcx_set_forest(vvcx_handle *h, int n_trees, int n_nodes, int n_classes, const int32_t *root, const int32_t *feature, const double *threshold,
                     const int32_t *left, const int32_t *right, const double *value, const int32_t *classes);
/* bind n pictures (device pointers) as the current batch and reset every CTU stream to its tile start
 * (≙ Picture::finalInit + the context reset of EL/EncSlice.cpp:1640-1647) */
int  vvcx_bind_frames(vvcx_handle *h, const vvcx_frame *frames, int n);
/* ≙ n calls of EncCu::compressCtu (EL/EncCu.cpp:428) followed by the estimator pass
 * CABACWriter::coding_tree_unit that advances the contexts (EL/EncSlice.cpp:1775-1776).
 * Tasks of one (frame, tile) stream must appear in raster order of the tile and continue where the last
 * call stopped; different streams run concurrently, one workgroup per stream.  `out` is host memory [n].
 * `hip_stream` is a hipStream_t (NULL = default stream); the call returns after the work completed. */
int  vvcx_compress_ctus(vvcx_handle *h, const vvcx_ctu_task *tasks, int n, vvcx_ctu_result *out, void *hip_stream);
/* The same call in two halves, for a host that keeps working while the device searches (the reference's CTU loop is synchronous,
 * EL/EncSlice.cpp:1560-1790; an encoder with frame threads would overlap picture N's search with picture N-1's write-out here).
 * vvcx_submit_ctus validates and enqueues uploads, the launch and the copy of the results into pinned host memory on `hip_stream`
 * and returns without waiting; vvcx_poll_ctus = 1 when done, 0 while running; vvcx_wait_ctus blocks, advances the streams'
 * positions and fills out[n] (n = the submitted count).  One submission per handle may be outstanding; until it is collected
 * every other entry point that touches the handle's device state returns VVCX_ERR_STATE. */
int  vvcx_submit_ctus(vvcx_handle *h, const vvcx_ctu_task *tasks, int n, void *hip_stream);
int  vvcx_poll_ctus(vvcx_handle *h);
int  vvcx_wait_ctus(vvcx_handle *h, vvcx_ctu_result *out, int n);
/* convenience: all CTUs of all bound frames in stream order (one launch) */
int  vvcx_compress_bound_frames(vvcx_handle *h, vvcx_ctu_result *out /* [n_frames * ctus_per_frame], host */, void *hip_stream);
/* final CU table of one bound frame (CTU raster order; per CTU luma CUs then chroma CUs, by origin).
 * ≙ walking cs.cus after the CTU loop; same fields D_BLOCK_STATISTICS_CODED traces */
int  vvcx_get_cus(vvcx_handle *h, int frame, vvcx_cu *cus, int max_cus, int *n_cus);
/* The TU records of the same picture (≙ cs.tus; EL/EncSlice.cpp reads them through CodingStructure::traverseTUs when it writes the slice): one per CU
 * (every intra CU of a dual-tree I slice fits MaxTbSize 64) or one per sub-partition of a luma CU coded with ISP (2 or 4, in coding order), CUs in vvcx_get_cus order.  coeff_offset[c] / coeff_stride[c] address the block's quantised levels
 * inside the plane vvcx_get_levels(h, frame, c, ...) returns (-1: the TU has no block of component c); a joint chroma TU keeps its levels with the coded
 * component (Cb for joint_cb_cr 2 / 3, Cr for 1).  tus may be NULL to query the count. */
int  vvcx_get_tus(vvcx_handle *h, int frame, vvcx_tu *tus, int max_tus, int *n_tus);
/* ≙ tu.getCoeffs(compID) of the final TUs: the quantised levels of component comp (0 Y, 1 Cb, 2 Cr) at their sample positions, copied to
 * a host plane (what a caller needs to rebuild cs.tus for the reference's own CABACWriter instead of taking vvcx_get_payload) */
int  vvcx_get_levels(vvcx_handle *h, int frame, int comp, int16_t *plane, int stride);
/* LMCS: the search runs, and leaves the luma reconstruction, in the mapped domain (the original luma of a bound picture is forward mapped into a buffer of the handle
 * when it is bound, ≙ EncGOP::xPicInitLMCS, EL/EncGOP.cpp:1689-1695; an SDR intra picture has no CTU-level weighted distortion: EL/EncSlice.cpp sets
 * CTUFlag false for it, so the mapped-domain SSE is the reference's distortion).  This maps the luma reconstruction of every bound picture back through the
 * inverse LUT in place (≙ the picture-level rspSignal before the loop filters, DL/DecLib.cpp executeLoopFilters); call it once all CTUs are coded and before
 * vvcx_deblock_bound_frames.  VVCX_ERR_STATE when the slice does not enable LMCS. */
int  vvcx_lmcs_inverse_reco(vvcx_handle *h, void *hip_stream);
/* ≙ EncGOP::xPicInitLMCS → EncReshape::preAnalyzerLMCS + the model half of constructReshaperLMCS (EL/EncReshape.cpp:410-559, 166-409, 977-1229, 1835-1893, 2194-2256) for an intra
 * picture with the cfg's LMCSSignalType 0 and LMCSAdpOption 0: the picture analysis that decides whether the slice uses LMCS and with which model.  org / stride: the original
 * 4:2:0 planes in HOST memory (uint8 below 10 bit, else uint16), update_ctrl = cfg LMCSUpdateCtrl (1 in encoder_intra.cfg; 0 supported, 2 is not an intra setting).  Fills
 * slice->lmcs_enable / lmcs_chroma_adj / lmcs_min_bin / lmcs_max_bin / lmcs_delta_cw (the other members are left alone); hand the slice to vvcx_set_slice before
 * vvcx_bind_frames.  Like the reference it leaves LMCS off for every picture below 10 bit and for full-range 10-bit content.  Needs no handle; the statistics pass (window
 * variances per luma bin, moments of the three planes) runs on the current HIP device: vvcx_lmcs_analyze uploads the host planes first, vvcx_lmcs_analyze_device takes planes
 * that are in device memory already (the pointers later given to vvcx_bind_frames) */
int  vvcx_lmcs_analyze(const void *const org[3], const int stride[3], int pic_w, int pic_h, int bit_depth, int slice_qp, int update_ctrl, vvcx_slice *slice);
int  vvcx_lmcs_analyze_device(const void *const org[3], const int stride[3], int pic_w, int pic_h, int bit_depth, int slice_qp, int update_ctrl, vvcx_slice *slice);
/* the LUTs and tables the handle derived from the slice's model: fwd / inv [1 << bit_depth], pivot[17] (mapped-domain bin borders), chroma_scale[16] (11 fractional bits) */
int  vvcx_lmcs_tables(vvcx_handle *h, int16_t *fwd, int16_t *inv, int32_t pivot[17], int32_t chroma_scale[16]);
/* ≙ LoopFilter::loopFilterPic (CL/LoopFilter.cpp:153; called from EncGOP after the slices of a picture are compressed): in-loop deblocking
 * of every bound picture, in place on its reconstruction planes; offsets = cfg LoopFilterBetaOffset_div2 / LoopFilterTcOffset_div2.
 * Every CTU of the pictures must have been compressed.  SAO and ALF, which follow in the reference, are not built. */
int  vvcx_deblock_bound_frames(vvcx_handle *h, int beta_offset_div2, int tc_offset_div2, void *hip_stream);
float vvcx_last_deblock_ms(const vvcx_handle *h);
/* ≙ SampleAdaptiveOffset::SAOProcess(cs, saoBlkParams) (CL/SampleAdaptiveOffset.cpp:617-670): sample adaptive offset on every bound (completely coded) picture, in place on
 * the reconstruction planes, with the caller's per-CTU parameters prm[frame][ctu raster address][component] (≙ SAOBlkParam, CL/TypeDef.h:1122-1140): mode 0 off / 1 new /
 * 2 merge; type: for a new CTU 0..3 = edge class 0 / 90 / 135 / 45 degrees, 4 = band offset, for a merge 0 = from the CTU to the left, 1 = from the CTU above (which must be
 * in the same tile); band = band position (0..31); offset[4] = the four coded offsets (edge: full valley, half valley, half peak, full peak; band: the four bands from the
 * band position), multiplied by 2^log2_offset_scale.  lf_across_tiles = pps loop_filter_across_bricks_enabled_flag.  The parameter DECISION (EncSampleAdaptiveOffset) is
 * not part of the library: the caller's encoder keeps it.  In the reference's order this follows vvcx_deblock_bound_frames. */
typedef struct vvcx_sao_param { int8_t mode, type, band, offset[4]; } vvcx_sao_param;
int   vvcx_sao_bound_frames(vvcx_handle *h, const vvcx_sao_param *prm, int lf_across_tiles, int log2_offset_scale, void *hip_stream);
float vvcx_last_sao_ms(const vvcx_handle *h);
/* ≙ EncSampleAdaptiveOffset::getStatistics (EL/EncSampleAdaptiveOffset.cpp:284-353 with getBlkStats 1135-1549, SAOLcuBoundary 0 as in the cfg): the statistics the
 * reference's parameter decision (decideBlkParams) works from, gathered on the device from the bound original and the deblocked reconstruction - the O(samples) half of that
 * decision; its RD half stays with the caller's encoder.  stats (host memory): [frame][ctu raster address][component][type 0..3 = edge class 0 / 90 / 135 / 45 degrees,
 * 4 = band][0: count, 1: sum of (original - deblocked)][32] int64 (≙ SAOStatData::count / ::diff; edge types use entries 0..4 = the classes full valley, half valley, plain,
 * half peak, full peak; band: 32 bands).  Call it after vvcx_deblock_bound_frames and before vvcx_sao_bound_frames.  Not available for an LMCS slice (the handle keeps the
 * mapped original only): VVCX_ERR_UNSUPPORTED. */
int   vvcx_sao_statistics_bound_frames(vvcx_handle *h, int lf_across_tiles, int64_t *stats, void *hip_stream);
float vvcx_last_sao_stats_ms(const vvcx_handle *h);
/* ≙ EncSampleAdaptiveOffset::decideBlkParams (EL/EncSampleAdaptiveOffset.cpp:793-1098, SAOGreedyEnc 0; deriveModeNewRDO 597-735, deriveModeMergeRDO 737-791, deriveOffsets
 * 481-595): the RD half of the SAO parameter decision for ONE picture from its statistics (the [ctu][component][type][count | diff][32] block of that picture as
 * vvcx_sao_statistics_bound_frames returns it): per CTU the better of new parameters (per component the type and offsets with the least distortion + lambda * bits; Cb and Cr
 * share the type) and a merge with the CTU to the left / above in the same tile, bits as CABACWriter::sao_block_pars codes them against the two SAO context models, which
 * start from the I-slice initialisation at slice_qp and adapt from CTU to CTU.  lambda[3] = the slice's lambdas per component (≙ Slice::getLambdas); every component is
 * enabled (what decidePicParams decides for temporal layer 0).  Host code, needs no handle.  prm[ctu][component] = the coded parameters vvcx_sao_bound_frames takes.
 * Parity with the reference is unpinned (the unit does not compile in this environment). */
int   vvcx_sao_decide(int pic_w, int pic_h, int bit_depth, int tile_cols, int tile_rows, int slice_qp, const double *lambda, int log2_offset_scale,
                      const int64_t *stats, vvcx_sao_param *prm);
/* the same filter on one picture in host memory (uint16 planes, stride = plane width, filtered in place; prm[ctu][component]): needs no handle */
int   vvcx_sao_picture(int pic_w, int pic_h, int bit_depth, int tile_cols, int tile_rows, const vvcx_sao_param *prm, int lf_across_tiles, int log2_offset_scale,
                       uint16_t *y, uint16_t *cb, uint16_t *cr, int device);
/* ≙ AdaptiveLoopFilter::ALFProcess(cs) (CL/AdaptiveLoopFilter.cpp:205-383): the adaptive loop filter on every bound (completely coded) picture, in place on the
 * reconstruction planes, with the caller's parameter sets and per-CTU choices.  aps[n_aps] (n_aps <= 8) = what the ALF parameter sets carry (≙ AlfParam,
 * CL/AlfParameters.h:134-160, JVET_O0669 form: no coefficient prediction): up to 25 luma filters of 12 coefficients + clipping indices (used when nonlinear_luma), the filter of
 * each of the 25 classes (filterCoeffDeltaIdx), up to 8 chroma alternatives of 6 coefficients + clipping indices.  slices[frame] (≙ the slice header): the parameter sets luma
 * is filtered with (luma_aps[k] -> filter set 16 + k; sets 0..15 are the standard's fixed ones) and the set of the chroma alternatives (chroma_aps < 0: chroma stays).
 * ctus[frame][ctu raster address]: enable flags Y / Cb / Cr (≙ Picture::getAlfCtuEnableFlag), luma filter set (getAlfCtbFilterIndex), chroma alternatives
 * (getAlfCtuAlternativeData).  Class derivation (deriveClassificationBlk 792-1002), the 7 x 7 / 5 x 5 diamond filters with clipping (filterBlk 1005-1296) and the virtual
 * boundary above every lower CTU border are the reference's; tiles are not looked at (its ALF of this version does not either).  The parameter DECISION
 * (EncAdaptiveLoopFilter) is not part of the library.  In the reference's order this follows vvcx_sao_bound_frames. */
typedef struct vvcx_alf_aps {
  int32_t num_luma_filters; uint8_t class_to_filter[25]; uint8_t nonlinear_luma; int16_t luma_coeff[25][12]; uint8_t luma_clip_idx[25][12];
  int32_t num_chroma_alt; uint8_t nonlinear_chroma[8]; int16_t chroma_coeff[8][6]; uint8_t chroma_clip_idx[8][6];
} vvcx_alf_aps;
typedef struct vvcx_alf_slice { int32_t n_luma_aps, luma_aps[8], chroma_aps; } vvcx_alf_slice;
typedef struct vvcx_alf_ctu { uint8_t flag[3]; int8_t set; uint8_t alt[2]; } vvcx_alf_ctu;
int   vvcx_alf_bound_frames(vvcx_handle *h, const vvcx_alf_aps *aps, int n_aps, const vvcx_alf_slice *slices, const vvcx_alf_ctu *ctus, void *hip_stream);
float vvcx_last_alf_ms(const vvcx_handle *h);
/* the same filter on one picture in host memory (uint16 planes, stride = plane width, filtered in place): needs no handle.  classes (may be NULL): class | transpose << 5 of
 * every luma 4 x 4 block of the CTUs with luma enabled (255 elsewhere), (pic_w / 4) per row */
int   vvcx_alf_picture(int pic_w, int pic_h, int bit_depth, const vvcx_alf_aps *aps, int n_aps, const vvcx_alf_slice *slice, const vvcx_alf_ctu *ctus,
                       uint16_t *y, uint16_t *cb, uint16_t *cr, uint8_t *classes, int device);
/* ≙ LoopFilter::loopFilterPic on a picture the caller describes itself (the way the reference's LoopFilter sees a CodingStructure: cs.cus + cs.tus + the reconstruction buffer):
 * rows = n_rows x {channel type (0 luma tree, 1 chroma tree), x, y, w, h in luma samples, cu.ispMode (0, 1 = horizontal, 2 = vertical split; luma rows only)} covering both
 * trees of the whole 4:2:0 picture, every CU intra at the slice QP (qp_cb / qp_cr = mapped chroma QPs); y / cb / cr = host planes of 16-bit samples, stride = plane width,
 * filtered in place.  The transform edges of ISP sub-partitions are filtered as xDeblockCU does (CL/LoopFilter.cpp:306-317, filter lengths from the sub-partition sizes,
 * 474-575).  Needs no handle: it is the filter of vvcx_deblock_bound_frames behind a table interface */
int  vvcx_deblock_cu_table(int pic_w, int pic_h, int bit_depth, int qp, int qp_cb, int qp_cr, int beta_offset_div2, int tc_offset_div2,
                           const int32_t *rows, int n_rows, uint16_t *y, uint16_t *cb, uint16_t *cr, int device);
/* slice_data() payload of one completely coded tile of a bound frame: the bytes EncSlice::encodeSlice would hand to the NAL writer
 * for that brick (CABACWriter::coding_tree_unit per CTU, end_of_ctu / end_of_slice terminating bins, byte alignment;
 * EL/EncSlice.cpp:1884-2006).  Requires cfg.emit_payload.  buf is host memory */
int  vvcx_get_payload(vvcx_handle *h, int frame, int tile, uint8_t *buf, int cap, int *nbytes);
/* ≙ the fork's GET_TRAINING_SET build (CL/TypeDef.h:54-56; EL/CABACWriter.cpp:515-855 writes the rows the forests of BIN/TEST.py are trained on): with a dump enabled every luma
 * node of the search that qualifies for the classifier (EL/EncCu.cpp:810-846, 933) leaves one row of 28 int32: the 26 features (EL/EncCu.cpp:863-1123), the complexity class
 * (0 simple, 1 fuzzy, 2 complex; 1127-1138) and the partition the full search chose at that node (0 none, 1 QT, 2 BT_H, 3 BT_V, 4 TT_H, 5 TT_V; -1: no encoding).  Rows
 * accumulate from vvcx_bind_frames on, in no particular order across streams; cap_rows 0 switches the dump off.  vvcx_get_training_rows copies min(rows so far, capacity,
 * max_rows) rows to host memory and reports in *n_rows how many the search produced.  Works with and without VVCX_TOOL_FAST (without: the plain full search labels the rows,
 * which is how a forest is trained: tools/train_partition_forest.py).  VVCX_ERR_UNSUPPORTED on a VVCX_TOOL_WPP handle (the features read neighbour CUs of other CTU rows,
 * whose progress under WPP is a matter of timing - the reason vvcx_create refuses VVCX_TOOL_FAST with WPP) */
int  vvcx_enable_training_dump(vvcx_handle *h, int cap_rows);
int  vvcx_get_training_rows(vvcx_handle *h, int32_t *rows, int max_rows, int *n_rows);
/* the sub-streams inside the bytes vvcx_get_payload returns for a tile, in order: one (the tile), or with VVCX_TOOL_WPP one per CTU row of the tile - what the slice header's
 * entry points are made of (EL/EncSlice.cpp:1982-1990 addSubstreamSize).  sizes may be NULL to query the count */
int  vvcx_get_substream_sizes(vvcx_handle *h, int frame, int tile, int *sizes, int max_sizes, int *n_sizes);
/* Diagnostic environment variables (read by vvcx_create / at launch; they never change a result):
 *   VVCX_MAX_WG_PER_CU=n          at most n resident CTU streams per CU (the rate-against-occupancy curve of DESIGN.md §6)
 *   VVCX_WPP_TEST_INTERLEAVE=1    VVCX_TOOL_WPP: one CTU per visit and round-robin choice of the next CTU row, so that the rows of a picture take turns */
/* device time of the last compress launch, measured with HIP events on the launch stream (ms) */
float vvcx_last_kernel_ms(const vvcx_handle *h);
/* work counters of the last launch: [0] SATD-stage candidates, [1] full-RD TU evaluations, [2] RD pixels, [3] nodes */
int  vvcx_get_counters(vvcx_handle *h, uint64_t out[4]);
/* diagnostic (only filled by a -DVVCX_STAMP=1 build): shader-clock ticks summed over streams of the last launch:
 * [0] mode controller, [1..11] parallel operation kinds, [12] estimator pass, [16..27] controller phases, [30] steps */
int  vvcx_get_profile(vvcx_handle *h, uint64_t out[48]);
const char *vvcx_last_error(void);
int  vvcx_ctus_per_frame(const vvcx_handle *h);
/* number of CTU streams the device runs concurrently (CUs x resident workgroups per CU); batches with at least this
 * many (frame, tile) streams fill the GPU.  ≙ the worker-thread count a frame-parallel host loop would size for */
int  vvcx_resident_streams(const vvcx_handle *h);


/* ---- leaf operators: the device functions of the path, individually callable.  They stand where the
 * reference has its run-time seams — RdCost::m_afpDistortFunc[DF_*] (CL/RdCost.h:60,104) — and its CommonLib entry points
 * IntraPrediction::initIntraPatternChType + predIntraAng (CL/IntraPrediction.cpp:304,1064), BinProbModel_Std
 * (CL/Contexts.h:86-155), RdCost::calcRdCost (CL/RdCost.cpp:63), the scan tables (CL/Rom.cpp:133-370).  Test and
 * diagnostic entry points: all pointers are HOST memory, the work runs on the device the handle / call selects. */
/* SAD, SATD (RdCost::xGetHADs tiling and normalisation) and SSE of n pairs of w x h blocks stored back to back; out[n][3] */
int  vvcx_distortion_batch(const int16_t *a, const int16_t *b, int w, int h, int n, uint64_t *out, int device);
/* intra prediction of n blocks of one picture: reco = planar 4:2:0 samples of the handle's size and bit depth (uint8 / uint16),
 * coded[2] = one byte per 4x4 luma unit (uw x uh, 1 = already reconstructed) for the luma and the chroma tree.
 * x, y, w, h in samples of the component; mode 0..66, mrl 0/1/3 (luma).  pred: concatenated w*h tiles */
typedef struct { int32_t comp, x, y, w, h, mode, mrl; } vvcx_pred_case;
int  vvcx_intra_pred_batch(vvcx_handle *h, const void *const reco[3], const uint8_t *const coded[2], const vvcx_pred_case *cases, int n, int16_t *pred);
/* CtxStore initialisation of an I slice (CL/Contexts.cpp:135-151) and the estimator's model update over a bin string */
int  vvcx_ctx_init(int qp, uint16_t s0[386], uint16_t s1[386]);
int  vvcx_cabac_code_bins(uint16_t *s0, uint16_t *s1, int ctx, const uint8_t *bins, int nbins, uint64_t *frac_bits, int device);
/* RdCost::calcRdCost for n (fracBits, dist) pairs at one lambda */
int  vvcx_rd_cost_batch(double lambda, const uint64_t *frac_bits, const uint64_t *dist, int n, double *cost, int device);
/* the core of IntraSearch::xIntraCodingTUBlock (EL/IntraSearch.cpp:2852-3168) for n blocks: residual org - pred → TrQuant::transformNxN
 * (DCT-II, plain Quant::quant) → levels; if any: invTransformNxN → rec = clip(pred + residual'); SSE(org, rec).  qp is the QP QpParam
 * hands to the quantiser (slice QP + QpBDOffset, chroma after the mapping table) */
int  vvcx_transform_quant_batch(const int16_t *org, const int16_t *pred, int w, int h, int bit_depth, int qp, int n,
                                int16_t *lev, int16_t *rec, uint64_t *sse, uint8_t *cbf, int device);
/* the same block pipeline with the dependent quantiser (≙ DepQuant::quant / dequant, CL/DepQuant.cpp:1755-1810, with the slice's dep_quant_enabled_flag on):
 * comp 0 Y / 1 Cb / 2 Cr; mts_idx 0 or 2..5 (luma, up to 32x32); cbf_cb = tu.cbf[Cb] when Cr is quantised; lambda = the quantiser's lambda of the
 * component (TrQuant::selectLambda); s0 / s1 = the two states of each of the 386 context models the rate terms are read from */
int  vvcx_depquant_batch(const int16_t *org, const int16_t *pred, int w, int h, int bit_depth, int qp, int comp, int mts_idx, int cbf_cb, double lambda,
                         const uint16_t *s0, const uint16_t *s1, int n, int16_t *lev, int16_t *rec, uint64_t *sse, uint8_t *cbf, int device);
/* the same with the LFNST of a CU with lfnstIdx 1 / 2 between the (zeroed-out) DCT-II and the dependent quantiser and back (≙ TrQuant::xFwdLfnst / xInvLfnst,
 * CL/TrQuant.cpp:319-560; blocks below 4x4 take none): intra_dir = the block's final intra mode (0..66; planar for a MIP CU, the co-located luma
 * mode for DM / CCLM chroma), from which the kernel set and the transposition follow after the wide-angle mapping of the block shape */
int  vvcx_lfnst_depquant_batch(const int16_t *org, const int16_t *pred, int w, int h, int bit_depth, int qp, int comp, int lfnst_idx, int intra_dir, int cbf_cb, double lambda,
                               const uint16_t *s0, const uint16_t *s1, int n, int16_t *lev, int16_t *rec, uint64_t *sse, uint8_t *cbf, int device);
/* transform skip of n luma residual blocks (4..32 per side; VVCX_TOOL_TS): ≙ the {DCT2, TS} pruning of TrQuant::transformNxN (CL/TrQuant.cpp:1049-1124; keep = the
 * transform-skip candidate survives), xTransformSkip 1394-1440, QuantRDOQ::xRateDistOptQuantTS (CL/QuantRDOQ.cpp:1243-1483, reached through DepQuant::quant for
 * MTS_SKIP blocks) with rates from the context models s0 / s1 and the quantiser's lambda, Quant::dequant + xITransformSkip 996-1041 (resi_out), and the fractional
 * bits CABACWriter::residual_codingTS (EL/CABACWriter.cpp:4306-4555) spends on the levels.  qp = slice QP + QpBDOffset (the TS minimum QP 4 is applied inside). */
int  vvcx_transform_skip_batch(const int16_t *resi, int w, int h, int bit_depth, int qp, double lambda, const uint16_t *s0, const uint16_t *s1, int n,
                               int16_t *lev, int16_t *resi_out, int32_t *abs_sum, uint8_t *keep, uint64_t *frac_bits, int device);
/* the block pipeline of n luma TUs of tw x th samples (1 x N, 2 x N, N x 1, N x 2 or larger; at least 16 samples) of CUs coded with intra sub-partitions (VVCX_TOOL_ISP): ≙
 * TrQuant::transformNxN with the implicit DST-VII / DCT-II choice of ISP blocks (getTrTypes, CL/TrQuant.cpp:752-780) and the one-stage transforms of one-sample-wide
 * blocks (895-914, 970-983), DepQuant::quant with the cbf context of ISP sub-partitions (prev_cbf = the previous sub-partition's cbf; cbf_inferred: the last
 * sub-partition after all-zero ones, EL/CABACWriter.cpp:3574-3600), dequantisation, inverse, reconstruction over pred; qp = slice QP + QpBDOffset */
int  vvcx_isp_tu_batch(const int16_t *org, const int16_t *pred, int tw, int th, int bit_depth, int qp, double lambda, int prev_cbf, int cbf_inferred,
                       const uint16_t *s0, const uint16_t *s1, int n, int16_t *lev, int16_t *rec, uint64_t *sse, uint8_t *cbf, int device);
/* coefficient scan (diagonal, grouped) of a w x h block: idx[min(w,32) * min(h,32)] raster offsets in scan order */
int  vvcx_scan_order(int w, int h, uint16_t *idx, int device);
/* ≙ BIN/TEST.py GetPartition(C0..C25, 2): the forest of vvcx_set_forest on n rows of 26 int32 features (host pointers) → class per row */
/* ≙ IntraPrediction::initIntraMip + predIntraMip (CL/IntraPrediction.cpp:2152-2205; MatrixIntraPrediction, JVET_O0925 form) for n luma
 * blocks: cases = n x {w, h, mode, bit_depth}; refs = per case top[w] | left[h] (unfiltered line-0 reference samples); pred = per case w*h.
 * The search uses the same device functions for its MIP candidates (VVCX_TOOL_MIP); this entry point exposes them for the parity test. */
int  vvcx_mip_pred_batch(const int32_t *cases, int n, const int16_t *refs, int n_refs, int16_t *pred, int n_pred, int device);
int  vvcx_forest_predict_batch(vvcx_handle *h, const int32_t *rows, int n, int32_t *out);

#ifdef __cplusplus
}
#endif
#endif

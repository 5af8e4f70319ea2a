// vvcx_dev.h — structures shared by the host side of the C-ABI (vvcx_api.hip) and the device kernel
// (vvcx_kernel.hip).  Plain PODs, laid out for HBM residency (see DESIGN.md "Data layout").
#pragma once
#include <stdint.h>

#ifndef VXD_NW
#define VXD_NW 4            // wavefronts (64 lanes) per workgroup = one CTU stream
#endif
#define VXD_NT (VXD_NW * 64) // threads per workgroup
#ifndef VXD_BUF
#define VXD_BUF 256          // samples of a node (luma w*h, chroma 2*cw*ch) up to which its candidates are evaluated in LDS buffers
#endif
#ifndef VXD_WPE
#define VXD_WPE 5            // waves per SIMD the register budget is sized for = CTU streams per CU (one wave of each stream per SIMD): 96 VGPRs and at most 32 KB of LDS
#endif
#define VXD_MAXD 12         // recursion levels kept in LDS (a split at least halves the area: 128x128 -> 4x4 is at most ten levels below the CTU)
#define VXD_NUM_CTX 291     // flat context array: the models of the reference's ContextSetCfg an intra slice of this path touches, in its order (vvcx_tables.h VX_NUM_CTX)

// one 4x4-luma unit of a channel-type map: the CU that covers it (≙ the fields of CodingUnit /
// PredictionUnit / TransformUnit the path reads from neighbours and the caller reads at the end)
struct VxUnit {
  uint64_t ss;              // splitSeries
  int16_t  x, y;            // CU origin, samples of this channel type
  uint8_t  lw, lh;          // log2 size, samples of this channel type
  uint8_t  qt, mt, bt, depth;
  uint8_t  dir, mrl, cbf, mts;  // mts: tu.mtsIdx of the luma TU (0 DCT2, 2..5)
  uint16_t tag;             // 0 = not coded in the current path; else tile index + 1
};

struct VxFrameDev {         // one bound picture
  const void *org[3];       // caller's planes (uint8 / uint16)
  void       *rec[3];
  int32_t     stride[3];
  int16_t    *lev[3];       // quantised levels, plane layout (owned by the handle)
  int32_t     lstride[3];
  VxUnit     *units[2];     // luma-tree / chroma-tree maps, uw x uh
  int32_t     jccr_sign;    // slice joint_cb_cr_sign_flag of the picture (VVCX_TOOL_JCCR; written by vvcx_jccr_sign_kernel at bind time)
  int32_t     pad_;
};

struct VxLeafPred { int32_t comp, x, y, w, h, mode, mrl; };      // same layout as vvcx_pred_case

struct VxStreamDesc { int32_t frame, tile, first_task, n_tasks, done_before /* CTUs of the sub-stream coded by earlier launches */, tile_ctus /* CTUs of the sub-stream */;
                      int32_t sub /* sub-stream of the frame: the tile, or under WPP the CTU row of the tile */, above /* WPP: sub-stream of the row above in the tile, -1: none */; };

struct VxCtuRes { uint64_t dist, bits; double cost; int32_t n_cu, pad; };

struct VxForestNode { double thr; int32_t left, right, feature, pad; };     // left < 0: leaf

// per-block constants of the dependent quantiser (Quantizer::initQuantBlock, CL/DepQuant.cpp:694-739; fp64 on the host), one entry per
// (component, log2 w + log2 h): index comp * 16 + log2 w + log2 h
struct VxDqConst { int32_t qshift, max_qidx, thres, dshift; int64_t qadd, qscale, dadd, dstep, dorg; };

struct VxParams {
  int32_t pic_w, pic_h, bit_depth, chroma;
  uint32_t tools;
  int32_t min_qt[2], max_bt_depth[2], max_bt_size[2], max_tt_size[2];
  int32_t ctus_w, ctus_h, uw, uh;
  int32_t qp, qp_c[2];               // slice QP, mapped chroma QPs
  int32_t qp_tr, qp_tr_c[2];         // + QpBDOffset: what QpParam hands to quantisation (CL/Quant.cpp:68-106)
  int32_t qp_tr_j;                   // the JointCbCr QP (joint blocks of cbf mask 3)
  double  lambda, dist_scale, sqrt_lambda_fp, dist_weight[2];
  const VxFrameDev   *frames;
  const VxStreamDesc *streams;
  const int32_t      *task_ctu;      // CTU raster address per task
  VxCtuRes           *results;       // per task
  uint16_t           *stream_ctx;    // per (frame*nsub+sub): 2*VXD_NUM_CTX states carried CTU -> CTU
  uint8_t            *scratch;       // per workgroup of the launch (= per resident stream slot)
  uint64_t            scratch_per_stream;
  unsigned long long *counters;      // 4 global work counters
  int32_t             ntiles, nsub;  // tiles / sub-streams per frame (equal without WPP)
  // slice_data writer (optional): per (frame, tile) byte range of the payload buffer and the persistent coder state (32 B each)
  uint8_t            *payload;
  const uint64_t     *payload_off;
  const uint32_t     *payload_cap;
  void               *arith_state;
  // FAST_ALGORITHM partition forest (optional): flattened sklearn trees, see include/vvcx.h vvcx_set_forest
  const VxForestNode *f_node;
  const double       *f_value;       // [n_nodes][f_nclasses] leaf distributions
  const int32_t      *f_root;        // [f_ntrees]
  int32_t             f_ntrees, f_nclasses;
  int32_t             f_classes[8];
  const VxDqConst    *dq_consts;     // [6 * 16] (VVCX_TOOL_DEPQUANT): Y, Cb, Cr, then the joint blocks of cbf masks 1, 2, 3 (VVCX_TOOL_JCCR)
  // LMCS (the slice enables it): chroma residual scaling from the luma neighbourhood of the 64x64 area; the quantiser constants then have one table per scale
  // (dq_consts[(1 + bin) * 96 ..], table 0 = no scaling)
  int32_t             lmcs_on, lmcs_cadj_on, lmcs_min_bin, lmcs_max_bin;
  int32_t             lmcs_pivot[17], lmcs_cadj[16];
  int32_t             n_streams;     // stream descriptors of the launch: the workgroups (at most one per resident slot) take them from a queue (counters[52])
  // (round-3 additions at the end: the members above keep their offsets in the LDS copy)
  int32_t             train_cap;
  int32_t            *train_rows;    // GET_TRAINING_SET counterpart: 28 ints per qualifying luma node (26 features, complexity class, chosen partition); NULL: off
  uint32_t           *train_n;       // rows handed out so far (atomic)
  int32_t            *wpp_progress;  // WPP: per (frame*nsub+sub) the CTUs of the row that are finished and visible
  uint16_t           *wpp_sync;      // WPP: per (frame*nsub+sub) the contexts behind the row's first CTU
  int32_t            *wpp_sched;     // WPP scheduler of the launch: [0] rows finished, [1] abort, [2] CTUs finished, [4, 4 + n_streams) owner flags, then the tasks each stream has left
  int32_t             wpp_rr, pad_wpp;    // test mode: one CTU per visit and round-robin choice, so that rows interleave also where nothing forces them to
};

struct VxDeblockParams {     // vvcx_deblock.hip
  const VxFrameDev *frames;
  int32_t uw, uh, bit_depth, chroma;
  int32_t qp, qp_c[2];       // every CU carries the slice QP; mapped chroma QPs (+ offsets)
  int32_t beta_off2, tc_off2, dir;      // cfg LoopFilterBetaOffset_div2 / LoopFilterTcOffset_div2; 0 = vertical edges, 1 = horizontal edges
  uint8_t *edges;            // per frame [direction][luma | chroma map][unit]: 0 = no edge to filter at the unit's left / upper border, else 0x80 | log2 sizeP << 3 | log2 sizeQ
};

// vvcx_sao.hip: resolved parameters of one (frame, CTU, component): type -1 off, 0..3 edge class (0 / 90 / 135 / 45 degrees), 4 band; offsets scaled, in the order
// full valley, half valley, half peak, full peak (edge) or band, band + 1, band + 2, band + 3 (band)
struct VxSaoEntry { int8_t type, band; int16_t off[4]; int16_t pad_; };
struct VxSaoParams {
  const VxFrameDev *frames;
  const VxSaoEntry *table;        // [frame][ctu][3]
  const uint8_t *tile_of_ctu;     // [ctu]
  void *tmp;                      // copy of the deblocked pictures: per frame tmp_frame samples, component c at tmp_comp[c], rows of the plane's width
  uint64_t tmp_frame, tmp_comp[3];
  int32_t pic_w, pic_h, ctus_w, ctus_h, bit_depth, chroma, lf_across_tiles, pad_;
};

struct VxSaoStatParams {
  const VxFrameDev *frames;
  const uint8_t *tile_of_ctu;     // [ctu]
  long long *out;                 // [frame][ctu][3][5 types][count | diff][32]
  int32_t pic_w, pic_h, ctus_w, ctus_h, bit_depth, chroma, lf_across_tiles, pad_;
};

// adaptive loop filter (vvcx_alf.hip): per frame the per-class tables of the slice's parameter sets, per CTU the caller's choices
struct VxAlfCtu { uint8_t flag[3]; int8_t set; uint8_t alt[2]; };
struct VxAlfFrame { int16_t luma_coeff[8][25][12], luma_clip[8][25][12], chroma_coeff[8][6], chroma_clip[8][6]; int32_t n_sets, n_alt; };
struct VxAlfParams {
  const VxFrameDev *frames;
  const VxAlfFrame *tabs;         // [frame]
  const VxAlfCtu *ctus;           // [frame][ctu]
  void *tmp;                      // copy of the pictures before the filter: per frame tmp_frame samples, component c at tmp_comp[c], rows of the plane's width
  uint64_t tmp_frame, tmp_comp[3];
  uint8_t *classes;               // optional: class | transpose << 5 per luma 4 x 4 block, per frame (pic_w / 4) * (pic_h / 4)
  int32_t pic_w, pic_h, ctus_w, ctus_h, bit_depth, chroma;
};

// per-stream scratch layout (bytes)
#define VXD_STORE_REC   (128 * 128 * 2)
#define VXD_STORE_UNITS (32 * 32 * (int) sizeof(VxUnit))
#define VXD_STORE_LEVEL (2 * VXD_STORE_REC + VXD_STORE_UNITS)                 // rec + lev + units
#define VXD_CTXSNAP     (2 * VXD_NUM_CTX * 2)                                 // s0 + s1
#define VXD_OFF_STORE   0
#define VXD_OFF_CTX     (VXD_OFF_STORE + VXD_MAXD * VXD_STORE_LEVEL)          // [MAXD + NW + 1][2] snapshots {start, best}: levels, per-wave parking, CTU start
#define VXD_OFF_SLOTS   ((VXD_OFF_CTX + (VXD_MAXD + VXD_NW + 1) * 2 * VXD_CTXSNAP + 255) & ~255) // big-block slots: [NW][2][2*4096] int16
#define VXD_SLOT_ELEMS  (2 * 4096)
#define VXD_OFF_TMP     (VXD_OFF_SLOTS + VXD_NW * 2 * VXD_SLOT_ELEMS * 2)             // big-block transform scratch: [NW][2048] int32
// CU-result cache of the current CTU (BestEncInfoCache, EL/EncModeCtrl.cpp:663-1110): one entry per (position in CTU in
// 4-sample units, log2 w, log2 h <= 6) and a level pool with one slot per (size, position aligned to max(4, size/2)):
// sum over sizes of size * 128 / max(4, size/2) = 1152 per dimension.
struct VxCacheEnt { uint64_t ss; uint8_t kind /* 0 empty, 1 luma tree, 2 chroma tree */, dir, mrl, cbf, depth, mts; uint16_t gen /* CTU generation of the scratch slot the entry belongs to */; };
#define VXD_CACHE_ENTRIES (32 * 32 * 5 * 5)
#define VXD_CACHE_DIM   1152
#define VXD_OFF_ORG     (VXD_OFF_TMP + VXD_NW * 2048 * 4)                             // original tile of a node too big for LDS: 4096 int16
#define VXD_OFF_LM      (VXD_OFF_ORG + 4096 * 2)                                      // CCLM down-sampled luma of a big chroma node: in[1024] | top[64] | left[64] int16
// dependent quantisation of blocks too big for LDS: per wave decisions (2 KB), path nodes of up to 64 coefficient groups (4 KB + 512 B), last-position offsets
#define VXD_DQ_WAVE     8192
#define VXD_OFF_DQ      ((VXD_OFF_LM + (1024 + 128) * 2 + 255) & ~255)
// candidate pool of the batched full-RD stage (dependent quantisation): the predictions of the node's candidates, their transform coefficients / levels, the
// path nodes of the batched trellis and one result record per (candidate, transform) item
#define VXD_POOL_ELEMS      32768                                                      // int16 elements of the prediction pool and of the coefficient pool
#define VXD_POOL_NODE_BYTES (128 * 1024)
#define VXD_POOL_ITEMS      64
struct VxRbItem { double cost; uint64_t dist, bits; int32_t cbf, sum0, test, wave; };   // 40 bytes
#define VXD_OFF_POOL    ((VXD_OFF_DQ + VXD_NW * VXD_DQ_WAVE + 255) & ~255)
#define VXD_OFF_POOL_COEF  (VXD_OFF_POOL + VXD_POOL_ELEMS * 2)
#define VXD_OFF_POOL_NODES (VXD_OFF_POOL_COEF + VXD_POOL_ELEMS * 2)
#define VXD_OFF_POOL_REC   (VXD_OFF_POOL_NODES + VXD_POOL_NODE_BYTES)
// JointCbCr: per wave joint residual | its reconstruction | levels | best pair of reconstructions | best levels (6 x 1024 int16).  In front of the CU cache: a handle
// without VVCX_TOOL_CU_REUSE gets VXD_OFF_CACHE bytes per stream and everything the other tools touch has to lie below that
// ISP: per wave the CU tiles of the candidate it evaluates (rec | lev: 2 x 4096 int16), the CU's prediction (4096) and the dense coefficient tile of a sub-partition (1024),
// then the node's best ISP candidate so far (rec | lev)
#define VXD_ISP_WAVE    14336                                                          // int16 elements per wave
#define VXD_OFF_ISP     ((VXD_OFF_POOL_REC + 2 * VXD_POOL_ITEMS * (int) sizeof(VxRbItem) + 255) & ~255)
#define VXD_OFF_ISP_BEST (VXD_OFF_ISP + VXD_NW * VXD_ISP_WAVE * 2)
#define VXD_OFF_JCCR    ((VXD_OFF_ISP_BEST + 2 * 4096 * 2 + 255) & ~255)
#define VXD_JCCR_WAVE   (6 * 1024 * 2)
#define VXD_OFF_CACHE   ((VXD_OFF_JCCR + VXD_NW * VXD_JCCR_WAVE + 255) & ~255)
#define VXD_OFF_CACHE_LEV (VXD_OFF_CACHE + VXD_CACHE_ENTRIES * (int) sizeof(VxCacheEnt))
#define VXD_OFF_META    (VXD_OFF_CACHE_LEV + VXD_CACHE_DIM * VXD_CACHE_DIM * 2)      // uint32: CTU generations this scratch slot has seen (validates CU-cache entries)
#define VXD_SCRATCH_BYTES (VXD_OFF_META + 256)

// vvcx_api.hip — host side of the C-ABI declared in include/vvcx.h: owns the device-resident state of a
// batch of pictures (level planes, CU maps, per-stream contexts and scratch), turns compressCtu-style
// tasks into one launch of the persistent per-stream kernel and reads results back.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdarg.h>
#include <stdlib.h>
#include <string.h>
#include <math.h>
#include <limits>
#include <algorithm>
#include <vector>
#include <algorithm>
#include "vvcx.h"
#include "vvcx_dev.h"
#include "vvcx_tables.h"      // host copy of the constant tables (context init values)

extern "C" __global__ void vvcx_compress_kernel_u8(VxParams p);
extern "C" __global__ void vvcx_compress_kernel_u16(VxParams p);
extern "C" __global__ void vvcx_compress_wpp_kernel_u8(VxParams p);
extern "C" __global__ void vvcx_compress_wpp_kernel_u16(VxParams p);
extern "C" __global__ void vvcx_leaf_dist_kernel(const int16_t *a, const int16_t *b, int w, int h, int16_t *scr, unsigned long long *out);
extern "C" __global__ void vvcx_leaf_pred_kernel_u8(VxParams p, const VxLeafPred *cases, int16_t *out, const int *out_off);
extern "C" __global__ void vvcx_leaf_pred_kernel_u16(VxParams p, const VxLeafPred *cases, int16_t *out, const int *out_off);
extern "C" __global__ void vvcx_leaf_cabac_kernel(uint16_t *io, int ctx, const uint8_t *bins, int nbins, unsigned long long *bits);
extern "C" __global__ void vvcx_leaf_rdcost_kernel(VxParams p, const unsigned long long *bits, const unsigned long long *dist, int n, double *cost);
extern "C" __global__ void vvcx_leaf_scan_kernel(int w, int h, uint16_t *idx);
extern "C" __global__ void vvcx_leaf_forest_kernel(VxParams p, const int32_t *rows, int n, int32_t *out);
struct VxMipCase { int32_t w, h, mode, bit_depth, ref_off, pred_off; };
extern "C" __global__ void vvcx_leaf_mip_kernel(const VxMipCase *cases, const int16_t *refs, int16_t *preds);
extern "C" __global__ void vvcx_sao_copy_kernel_u8(VxSaoParams p);
extern "C" __global__ void vvcx_sao_copy_kernel_u16(VxSaoParams p);
extern "C" __global__ void vvcx_sao_kernel_u8(VxSaoParams p);
extern "C" __global__ void vvcx_sao_kernel_u16(VxSaoParams p);
extern "C" __global__ void vvcx_sao_stats_kernel_u8(VxSaoStatParams p);
extern "C" __global__ void vvcx_sao_stats_kernel_u16(VxSaoStatParams p);
extern "C" __global__ void vvcx_alf_copy_kernel_u8(VxAlfParams p);
extern "C" __global__ void vvcx_alf_copy_kernel_u16(VxAlfParams p);
extern "C" __global__ void vvcx_alf_kernel_u8(VxAlfParams p);
extern "C" __global__ void vvcx_alf_kernel_u16(VxAlfParams p);
extern "C" __global__ void vvcx_deblock_edges_kernel(VxDeblockParams p);
extern "C" __global__ void vvcx_deblock_kernel_u8(VxDeblockParams p);
extern "C" __global__ void vvcx_deblock_kernel_u16(VxDeblockParams p);
extern "C" __global__ void vvcx_jccr_sign_kernel_u8(VxFrameDev *frames, int wc, int hc);
extern "C" __global__ void vvcx_jccr_sign_kernel_u16(VxFrameDev *frames, int wc, int hc);
extern "C" __global__ void vvcx_leaf_dq_kernel(VxParams p, const uint16_t *ctx, const int16_t *org, int16_t *rec, int16_t *lev, int32_t *tmp, int w, int h, int qp, int comp, int mts, int cbf_cb, unsigned long long *out, int lf, int lfdir);
extern "C" __global__ void vvcx_leaf_ts_kernel(VxParams p, const uint16_t *ctx, const int16_t *resi, int16_t *lev, int16_t *resi_out, int32_t *tmp, int w, int h, int qp, int *out, unsigned long long *bits);
extern "C" __global__ void vvcx_lmcs_map_kernel_u8(const uint8_t *src, uint8_t *dst, int w, int h, int stride, const int16_t *lut);
extern "C" __global__ void vvcx_lmcs_map_kernel_u16(const uint16_t *src, uint16_t *dst, int w, int h, int stride, const int16_t *lut);
extern "C" __global__ void vvcx_leaf_isp_kernel(VxParams p, const uint16_t *ctx, const int16_t *org, int16_t *rec, int16_t *lev, int32_t *tmp, int w, int h, int qp, int cbf_ctx, unsigned long long *out);
extern "C" __global__ void vvcx_ctu_activity_kernel_u8(const VxFrameDev *frames, int pic_w, int pic_h, int ctus_w, unsigned *out);
extern "C" __global__ void vvcx_ctu_activity_kernel_u16(const VxFrameDev *frames, int pic_w, int pic_h, int ctus_w, unsigned *out);
extern "C" __global__ void vvcx_leaf_trq_kernel(const int16_t *org, int16_t *rec, int16_t *lev, int32_t *tmp, int w, int h, int bd, int qp, unsigned long long *out);

static thread_local char g_err[512];
extern "C" const char *vvcx_last_error(void) { return g_err; }
// (for the other translation units of the library: csrc/vvcx_lmcs.hip)
extern "C" __attribute__((visibility("hidden"))) int vvcx_fail_msg_(int code, const char *msg) { snprintf(g_err, sizeof g_err, "%s", msg); return code; }
static int fail(int code, const char *fmt, ...)
{
  va_list ap; va_start(ap, fmt); vsnprintf(g_err, sizeof g_err, fmt, ap); va_end(ap);
  return code;
}
// every entry point works on the handle's device and leaves the caller's current device as it found it
struct DevGuard {
  int prev; bool ok;
  explicit DevGuard(int dev) : prev(-1), ok(false) { if (hipGetDevice(&prev) != hipSuccess) prev = -1; ok = hipSetDevice(dev) == hipSuccess; }
  ~DevGuard() { if (prev >= 0) (void) hipSetDevice(prev); }
};
#define HIPCHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) return fail(VVCX_ERR_DEVICE, "%s: %s", #x, hipGetErrorString(e_)); } while (0)

struct vvcx_handle {
  vvcx_cfg cfg; vvcx_slice sl; bool have_slice;
  int ctus_w, ctus_h, uw, uh, ntiles;
  std::vector<int> ctu_tile;                    // tile of each CTU (raster address)
  std::vector<std::vector<int>> tile_ctus;      // CTUs of each tile in coding (raster-in-tile) order
  int n_frames;
  std::vector<VxFrameDev> frames_h;
  // sub-streams: the units the search runs as independent (or, under WPP, lagged) sequences of CTUs and the arithmetic coder writes as one byte string each - a tile, or
  // with VVCX_TOOL_WPP one CTU row of a tile (EL/EncSlice.cpp:1972-1990).  Per frame: nsub of them, numbered tile by tile, row by row
  int nsub;
  std::vector<int> ctu_sub, sub_tile, sub_above, tile_sub0, tile_nsub;      // sub-stream of each CTU; its tile; the sub-stream of the CTU row above in the same tile (-1: none); per tile: first sub-stream, count
  std::vector<std::vector<int>> sub_ctus;       // CTUs of each sub-stream in coding order
  int32_t *train_rows_d; uint32_t *train_n_d; int train_cap;      // vvcx_enable_training_dump
  int32_t *wpp_sched_d; int wpp_sched_cap; int wpp_rr;      // the launch's WPP scheduler state (vvcx_kernel.hip run_streams_wpp); test mode (env VVCX_WPP_TEST_INTERLEAVE)
  int32_t *wpp_progress_d; uint16_t *wpp_sync_d;  // WPP: CTUs finished per (frame, sub-stream); the contexts behind the first CTU of each (m_entropyCodingSyncContextState)
  std::vector<int> next_idx;                    // per (frame, sub-stream): how many CTUs of the stream are done
  // device memory
  VxFrameDev *frames_d; int16_t *lev_d; VxUnit *units_d; uint16_t *stream_ctx_d;
  uint8_t *scratch_d; size_t scratch_cap;
  VxStreamDesc *streams_d; int32_t *task_ctu_d; VxCtuRes *results_d; int task_cap, stream_cap;
  unsigned long long *counters_d;
  // optional slice_data writer: payload bytes per (frame, tile), byte ranges, persistent arithmetic-coder state
  uint8_t *payload_d; uint64_t *payload_off_d; uint32_t *payload_cap_d; void *arith_d; std::vector<uint64_t> payload_off; std::vector<uint32_t> payload_cap;
  // FAST_ALGORITHM forest (vvcx_set_forest)
  VxForestNode *f_node_d; double *f_value_d; int32_t *f_root_d; int f_ntrees, f_nclasses; int32_t f_classes[8];
  VxDqConst *dq_d;                              // dependent-quantiser constants per (chroma scale table, component, log2 w + log2 h)
  // LMCS of the current slice (vvcx_set_slice): LUTs and tables, their device copy (fwd | inv), the forward-mapped original luma of the bound pictures
  bool lmcs_on, lmcs_inverted; int16_t lmcs_fwd[1024], lmcs_inv[1024]; int32_t lmcs_pivot[17], lmcs_cadj[16];
  int16_t *lmcs_lut_d; void *lmcs_org_d; size_t lmcs_org_cap;
  std::vector<uint32_t> activity;               // per (frame, CTU) of the bound pictures: orders the stream queue of a launch, longest first
  hipEvent_t ev0, ev1; float last_ms, last_deblock_ms, last_sao_ms;
  void *sao_tmp_d; size_t sao_tmp_cap; VxSaoEntry *sao_tab_d; size_t sao_tab_cap; uint8_t *sao_tile_d;      // vvcx_sao_bound_frames: picture copy, resolved parameters, CTU -> tile
  uint8_t *db_edges_d; size_t db_edges_cap;      // vvcx_deblock_bound_frames: one edge byte per unit, map and direction
  long long *sao_stat_d; size_t sao_stat_cap; float last_sao_stats_ms;      // vvcx_sao_statistics_bound_frames
  VxAlfFrame *alf_tab_d; VxAlfCtu *alf_ctu_d; size_t alf_cap; float last_alf_ms;      // vvcx_alf_bound_frames: per-frame tables and per-CTU choices (the picture copy is the SAO one)
  // a submitted, not yet collected launch (vvcx_submit_ctus .. vvcx_wait_ctus): staging the async copies read from / write to stays alive here
  bool pending; hipStream_t pend_stream; int pend_n; VxCtuRes *pend_res; int pend_cap;
  std::vector<VxStreamDesc> pend_sd; std::vector<int32_t> pend_task_ctu; std::vector<int> pend_src, pend_next; VxDqConst pend_dq[17 * 96];
  size_t lev_plane[3], lev_frame, units_plane, units_frame;
};

#define VVCX_PAYLOAD_BYTES_PER_CTU 32768u
static const uint32_t kBuiltTools = VVCX_TOOL_ISP | VVCX_TOOL_MRL | VVCX_TOOL_MIP | VVCX_TOOL_LFNST | VVCX_TOOL_MTS | VVCX_TOOL_JCCR | VVCX_TOOL_DEPQUANT | VVCX_TOOL_CU_REUSE | VVCX_TOOL_CCLM | VVCX_TOOL_FAST |
                                    VVCX_TOOL_TS | VVCX_TOOL_RDOQ | VVCX_TOOL_LMCS | VVCX_TOOL_WPP;

// Quantizer::initQuantBlock (CL/DepQuant.cpp:694-739) for blocks with log2 w + log2 h = lsum: the quantiser's shift / scale / thresholds and the fixed-point
// distortion normalisation, which the reference derives in fp64 from lambda.  qp: what QpParam hands over (with QpBDOffset).
static VxDqConst dq_consts_of(int lsum, int bit_depth, int qp, double lambda)
{
  VxDqConst c; memset(&c, 0, sizeof c);
  const int sq = lsum & 1, qpDQ = qp + 1, per = qpDQ / 6, rem = qpDQ - 6 * per;
  const int nomShift = 15 - bit_depth - (lsum >> 1), trShift = nomShift + (sq ? -1 : 0);
  const int qshift = 14 - 1 + per + trShift;
  const int64_t qscale = VX_QUANT_SCALES[sq * 6 + rem];
  const int invShift = 6 + 1 - per - trShift;
  int qIdxBD = 32 + invShift - 6 - 1; if (qIdxBD > 16) qIdxBD = 16;
  const int nomDShift = 15 - 2 * nomShift + qshift + (sq ? 1 : 0);
  const double qScale2 = (double) (qscale * qscale);
  const double nomDistFactor = nomDShift < 0 ? 1.0 / ((double) ((int64_t) 1 << (-nomDShift)) * qScale2 * lambda) : (double) ((int64_t) 1 << nomDShift) / (qScale2 * lambda);
  const int64_t pow2dfShift = (int64_t) (nomDistFactor * qScale2) + 1;
  int dfShift = (pow2dfShift & (pow2dfShift - 1)) ? 1 : 0;                   // ceil_log2 (680-693)
  for (uint64_t x = (uint64_t) pow2dfShift; x > 1; x >>= 1) dfShift++;
  const int dshift = 62 + qshift - 2 * 15 - dfShift;
  c.qshift = qshift; c.qadd = -(((int64_t) 3 << qshift) >> 1); c.qscale = qscale; c.max_qidx = (1 << (qIdxBD - 1)) - 4;
  c.thres = (int32_t) ((int64_t) 4 << qshift);
  c.dshift = dshift; c.dadd = ((int64_t) 1 << dshift) >> 1;
  c.dstep = (int64_t) (nomDistFactor * (double) ((int64_t) 1 << (dshift + qshift)) + .5);
  c.dorg = (int64_t) (nomDistFactor * (double) ((int64_t) 1 << (dshift + 1)) + .5);
  return c;
}

// entry points that touch a handle's device state refuse to run between vvcx_submit_ctus and vvcx_wait_ctus
#define NOT_PENDING(h) do { if ((h) && (h)->pending) return fail(VVCX_ERR_STATE, "a submitted launch is outstanding: call vvcx_wait_ctus first"); } while (0)

extern "C" int vvcx_create(const vvcx_cfg *cfg, vvcx_handle **out)
{
  if (!cfg || !out) return fail(VVCX_ERR_ARG, "null argument");
  if (cfg->tools & ~kBuiltTools) return fail(VVCX_ERR_UNSUPPORTED, "tool set 0x%x not built yet (available: 0x%x)", cfg->tools, kBuiltTools);
  // LFNST is built on top of the dependent quantiser (the reference's plain quantiser keeps positions its own decoder rejects with LFNST, CL/Quant.cpp:1054-1058)
  // and on the search's MIP form of the saved mode lists (EL/IntraSearch.cpp:750-775)
  if ((cfg->tools & VVCX_TOOL_LFNST) && (cfg->tools & (VVCX_TOOL_DEPQUANT | VVCX_TOOL_MIP)) != (VVCX_TOOL_DEPQUANT | VVCX_TOOL_MIP))
    return fail(VVCX_ERR_UNSUPPORTED, "VVCX_TOOL_LFNST needs VVCX_TOOL_DEPQUANT and VVCX_TOOL_MIP (tool set 0x%x)", cfg->tools);
  // transform skip is searched inside the LFNST pass structure and quantised by RDOQ-TS the way DepQuant::quant reaches it (CL/DepQuant.cpp:1755-1781); RDOQ alone (the
  // cfg sets RDOQ / RDOQTS beside DepQuant) only acts on transform-skip blocks
  if ((cfg->tools & VVCX_TOOL_TS) && (cfg->tools & (VVCX_TOOL_DEPQUANT | VVCX_TOOL_LFNST)) != (VVCX_TOOL_DEPQUANT | VVCX_TOOL_LFNST))
    return fail(VVCX_ERR_UNSUPPORTED, "VVCX_TOOL_TS needs VVCX_TOOL_DEPQUANT and VVCX_TOOL_LFNST (tool set 0x%x)", cfg->tools);
  // ISP is one of the candidate kinds of the first (lfnstIdx 0, DCT-II) pass of the LFNST pass loop and its sub-partitions go through the dependent quantiser with
  // the implicit DST-VII of explicit-MTS sequences (TrQuant::getTrTypes)
  if ((cfg->tools & VVCX_TOOL_ISP) && (cfg->tools & (VVCX_TOOL_DEPQUANT | VVCX_TOOL_LFNST | VVCX_TOOL_MTS)) != (VVCX_TOOL_DEPQUANT | VVCX_TOOL_LFNST | VVCX_TOOL_MTS))
    return fail(VVCX_ERR_UNSUPPORTED, "VVCX_TOOL_ISP needs VVCX_TOOL_DEPQUANT, VVCX_TOOL_LFNST and VVCX_TOOL_MTS (tool set 0x%x)", cfg->tools);
  if ((cfg->tools & VVCX_TOOL_LMCS) && !(cfg->tools & VVCX_TOOL_DEPQUANT)) return fail(VVCX_ERR_UNSUPPORTED, "VVCX_TOOL_LMCS needs VVCX_TOOL_DEPQUANT (tool set 0x%x)", cfg->tools);
  if ((cfg->tools & VVCX_TOOL_JCCR) && !(cfg->tools & VVCX_TOOL_DEPQUANT)) return fail(VVCX_ERR_UNSUPPORTED, "VVCX_TOOL_JCCR needs VVCX_TOOL_DEPQUANT (tool set 0x%x)", cfg->tools);
  // the classifier's features read the CUs above-right and below-left of a node through the unrestricted getCU (EL/EncCu.cpp:1126-1217): with the CTU rows running as lagged
  // streams whether those exist yet would depend on timing
  if ((cfg->tools & VVCX_TOOL_WPP) && (cfg->tools & VVCX_TOOL_FAST)) return fail(VVCX_ERR_UNSUPPORTED, "VVCX_TOOL_WPP cannot be combined with VVCX_TOOL_FAST (tool set 0x%x)", cfg->tools);
  if (cfg->ctu_size != 128 || !cfg->dual_tree) return fail(VVCX_ERR_UNSUPPORTED, "only CTUSize 128 with DualITree 1");
  if ((cfg->pic_w & 7) || (cfg->pic_h & 7) || cfg->pic_w <= 0 || cfg->pic_h <= 0) return fail(VVCX_ERR_ARG, "picture size must be a positive multiple of 8");
  if (cfg->bit_depth != 8 && cfg->bit_depth != 10) return fail(VVCX_ERR_UNSUPPORTED, "bit depth %d", cfg->bit_depth);
  if (cfg->max_frames < 1 || cfg->tile_cols < 1 || cfg->tile_rows < 1) return fail(VVCX_ERR_ARG, "max_frames / tile grid");
  vvcx_handle *h = new vvcx_handle();
  h->cfg = *cfg; h->have_slice = false; h->n_frames = 0; h->last_ms = 0.f;
  h->ctus_w = (cfg->pic_w + 127) >> 7; h->ctus_h = (cfg->pic_h + 127) >> 7;
  if (cfg->tile_cols > h->ctus_w || cfg->tile_rows > h->ctus_h) { delete h; return fail(VVCX_ERR_ARG, "more tiles than CTUs"); }
  h->uw = (cfg->pic_w + 3) >> 2; h->uh = (cfg->pic_h + 3) >> 2;
  h->ntiles = cfg->tile_cols * cfg->tile_rows;
  h->ctu_tile.resize((size_t) h->ctus_w * h->ctus_h);
  h->tile_ctus.assign((size_t) h->ntiles, std::vector<int>());
  for (int ry = 0; ry < h->ctus_h; ry++) for (int rx = 0; rx < h->ctus_w; rx++) {
    int tc = 0, tr = 0;                          // uniform spacing: boundary i = i*N/T
    for (int i = 0; i < cfg->tile_cols; i++) if (rx >= (i * h->ctus_w) / cfg->tile_cols) tc = i;
    for (int i = 0; i < cfg->tile_rows; i++) if (ry >= (i * h->ctus_h) / cfg->tile_rows) tr = i;
    const int t = tr * cfg->tile_cols + tc;
    h->ctu_tile[(size_t) ry * h->ctus_w + rx] = t;
    h->tile_ctus[(size_t) t].push_back(ry * h->ctus_w + rx);
  }
  {
    const bool wpp = (cfg->tools & VVCX_TOOL_WPP) != 0;
    h->ctu_sub.assign(h->ctu_tile.size(), 0); h->tile_sub0.assign((size_t) h->ntiles, 0); h->tile_nsub.assign((size_t) h->ntiles, 1);
    h->sub_ctus.clear(); h->sub_tile.clear(); h->sub_above.clear();
    for (int t = 0; t < h->ntiles; t++) {
      h->tile_sub0[(size_t) t] = (int) h->sub_ctus.size();
      int prev_row = -1, n = 0;
      for (int a : h->tile_ctus[(size_t) t]) {
        const int row = a / h->ctus_w;
        if (n == 0 || (wpp && row != prev_row)) { h->sub_ctus.push_back(std::vector<int>()); h->sub_tile.push_back(t); h->sub_above.push_back(n ? (int) h->sub_ctus.size() - 2 : -1); n++; }
        prev_row = row;
        h->sub_ctus.back().push_back(a); h->ctu_sub[(size_t) a] = (int) h->sub_ctus.size() - 1;
      }
      h->tile_nsub[(size_t) t] = n;
    }
    h->nsub = (int) h->sub_ctus.size();
  }
  h->wpp_sched_d = nullptr; h->wpp_sched_cap = 0; { const char *e = getenv("VVCX_WPP_TEST_INTERLEAVE"); h->wpp_rr = e && *e == '1'; }
  h->wpp_progress_d = nullptr; h->wpp_sync_d = nullptr; h->train_rows_d = nullptr; h->train_n_d = nullptr; h->train_cap = 0;
  DevGuard guard(cfg->device);
  if (!guard.ok) { delete h; return fail(VVCX_ERR_DEVICE, "hipSetDevice(%d) failed", cfg->device); }
  const int wc = cfg->pic_w >> 1, hc = cfg->pic_h >> 1;
  h->lev_plane[0] = (size_t) cfg->pic_w * cfg->pic_h; h->lev_plane[1] = h->lev_plane[2] = (size_t) wc * hc;
  h->lev_frame = h->lev_plane[0] + 2 * h->lev_plane[1];
  h->units_plane = (size_t) h->uw * h->uh; h->units_frame = 2 * h->units_plane;
  h->frames_d = nullptr; h->lev_d = nullptr; h->units_d = nullptr; h->stream_ctx_d = nullptr; h->scratch_d = nullptr; h->scratch_cap = 0;
  h->payload_d = nullptr; h->payload_off_d = nullptr; h->payload_cap_d = nullptr; h->arith_d = nullptr;
  h->streams_d = nullptr; h->task_ctu_d = nullptr; h->results_d = nullptr; h->task_cap = 0; h->stream_cap = 0; h->counters_d = nullptr;
  h->sao_tmp_d = nullptr; h->sao_tmp_cap = 0; h->sao_tab_d = nullptr; h->sao_tab_cap = 0; h->sao_tile_d = nullptr; h->last_sao_ms = 0.f; h->alf_tab_d = nullptr; h->alf_ctu_d = nullptr; h->alf_cap = 0; h->last_alf_ms = 0.f; h->sao_stat_d = nullptr; h->sao_stat_cap = 0; h->last_sao_stats_ms = 0.f; h->db_edges_d = nullptr; h->db_edges_cap = 0;
  h->dq_d = nullptr; h->lmcs_on = false; h->lmcs_inverted = false; h->lmcs_lut_d = nullptr; h->lmcs_org_d = nullptr; h->lmcs_org_cap = 0;
  const int F = cfg->max_frames;
  if (hipMalloc((void **) &h->frames_d, sizeof(VxFrameDev) * F) != hipSuccess || hipMalloc((void **) &h->lev_d, h->lev_frame * 2 * F) != hipSuccess ||
      hipMalloc((void **) &h->units_d, h->units_frame * sizeof(VxUnit) * F) != hipSuccess ||
      hipMalloc((void **) &h->stream_ctx_d, (size_t) F * h->nsub * 2 * VXD_NUM_CTX * 2) != hipSuccess ||
      ((cfg->tools & VVCX_TOOL_WPP) && (hipMalloc((void **) &h->wpp_progress_d, (size_t) F * h->nsub * 4) != hipSuccess || hipMalloc((void **) &h->wpp_sync_d, (size_t) F * h->nsub * 2 * VXD_NUM_CTX * 2) != hipSuccess)) ||
      hipMalloc((void **) &h->counters_d, 56 * sizeof(unsigned long long)) != hipSuccess ||
      hipMalloc((void **) &h->dq_d, 17 * 96 * sizeof(VxDqConst)) != hipSuccess) { vvcx_destroy(h); return fail(VVCX_ERR_DEVICE, "device allocation failed"); }
  if (cfg->emit_payload) {                       // VVCX_PAYLOAD_BYTES_PER_CTU per CTU: a CTU of 8-bit video at QP >= 17 stays far below (raw samples are 24 KB)
    const size_t nstream = (size_t) F * h->nsub;
    h->payload_off.resize(nstream); h->payload_cap.resize(nstream);
    uint64_t off = 0;
    for (size_t s2 = 0; s2 < nstream; s2++) { const uint32_t cap = (uint32_t) h->sub_ctus[s2 % (size_t) h->nsub].size() * VVCX_PAYLOAD_BYTES_PER_CTU; h->payload_off[s2] = off; h->payload_cap[s2] = cap; off += cap; }
    if (hipMalloc((void **) &h->payload_d, off) != hipSuccess || hipMalloc((void **) &h->payload_off_d, nstream * 8) != hipSuccess ||
        hipMalloc((void **) &h->payload_cap_d, nstream * 4) != hipSuccess || hipMalloc(&h->arith_d, nstream * 32) != hipSuccess) { vvcx_destroy(h); return fail(VVCX_ERR_DEVICE, "device allocation failed"); }
    (void) hipMemcpy(h->payload_off_d, h->payload_off.data(), nstream * 8, hipMemcpyHostToDevice);
    (void) hipMemcpy(h->payload_cap_d, h->payload_cap.data(), nstream * 4, hipMemcpyHostToDevice);
    (void) hipMemset(h->arith_d, 0, nstream * 32);
  }
  (void) hipEventCreate(&h->ev0); (void) hipEventCreate(&h->ev1);
  *out = h;
  return VVCX_OK;
}

extern "C" void vvcx_destroy(vvcx_handle *h)
{
  if (!h) return;
  (void) hipFree(h->frames_d); (void) hipFree(h->lev_d); (void) hipFree(h->units_d); (void) hipFree(h->stream_ctx_d); (void) hipFree(h->scratch_d); (void) hipFree(h->wpp_progress_d); (void) hipFree(h->wpp_sync_d); (void) hipFree(h->wpp_sched_d); (void) hipFree(h->train_rows_d); (void) hipFree(h->train_n_d);
  (void) hipFree(h->payload_d); (void) hipFree(h->payload_off_d); (void) hipFree(h->payload_cap_d); (void) hipFree(h->arith_d);
  (void) hipFree(h->streams_d); (void) hipFree(h->task_ctu_d); (void) hipFree(h->results_d); (void) hipFree(h->counters_d);
  (void) hipFree(h->sao_tmp_d); (void) hipFree(h->sao_tab_d); (void) hipFree(h->sao_tile_d); (void) hipFree(h->alf_tab_d); (void) hipFree(h->alf_ctu_d); (void) hipFree(h->sao_stat_d); (void) hipFree(h->db_edges_d);
  (void) hipFree(h->f_node_d); (void) hipFree(h->f_value_d); (void) hipFree(h->f_root_d); (void) hipFree(h->dq_d); (void) hipFree(h->lmcs_lut_d); (void) hipFree(h->lmcs_org_d);
  if (h->pending) (void) hipStreamSynchronize(h->pend_stream);
  (void) hipHostFree(h->pend_res);
  (void) hipEventDestroy(h->ev0); (void) hipEventDestroy(h->ev1);
  delete h;
}

// ---- slice-level inputs (host only)
extern "C" int vvcx_chroma_qp_table(int bit_depth, int n_pts, const int32_t *qp_in, const int32_t *qp_out, int32_t *table)
{
  if (!qp_in || !qp_out || !table || n_pts < 1 || n_pts > 8 || (bit_depth != 8 && bit_depth != 10)) return fail(VVCX_ERR_ARG, "bad argument");
  const int off = 6 * (bit_depth - 8), maxQp = 63;
  for (int j = 0; j < n_pts; j++) {
    if (qp_in[j] < -off || qp_in[j] > maxQp || qp_out[j] < -off || qp_out[j] > maxQp || (j && qp_in[j] <= qp_in[j - 1])) return fail(VVCX_ERR_ARG, "chroma QP pivot %d out of range or not increasing", j);
  }
  auto clipq = [&](int v) { return v < -off ? -off : v > maxQp ? maxQp : v; };
  int32_t *t = table + off;                           // t[q], q = -off .. 63
  t[qp_in[0]] = qp_out[0];
  for (int k = qp_in[0] - 1; k >= -off; k--) t[k] = clipq(t[k + 1] - 1);
  for (int j = 0; j + 1 < n_pts; j++) {
    const int dIn = qp_in[j + 1] - qp_in[j], dOut = qp_out[j + 1] - qp_out[j], sh = (dIn + 1) >> 1;      // deltaQpInValMinus1 + 1 = dIn
    for (int k = qp_in[j] + 1, m = 1; k <= qp_in[j + 1]; k++, m++) t[k] = t[qp_in[j]] + (dOut * m + sh) / dIn;
  }
  for (int k = qp_in[n_pts - 1] + 1; k <= maxQp; k++) t[k] = clipq(t[k - 1] + 1);
  return VVCX_OK;
}
extern "C" int vvcx_derive_slice(const vvcx_slice_cfg *c, vvcx_slice *out)
{
  if (!c || !out) return fail(VVCX_ERR_ARG, "null argument");
  const int off = 6 * (c->bit_depth - 8);
  if (c->qp < -off || c->qp > 63 || c->gop_size < 1) return fail(VVCX_ERR_ARG, "qp / gop_size");
  int32_t table[64 + 12];
  const int rc = vvcx_chroma_qp_table(c->bit_depth, c->n_pts, c->qp_in, c->qp_out, table);
  if (rc) return rc;
  memset(out, 0, sizeof *out);
  out->qp = c->qp;
  double scale = 0.05 * (double) (c->gop_size - 1); scale = scale < 0.0 ? 0.0 : scale > 0.5 ? 0.5 : scale;
  out->lambda = 0.57 * (1.0 - scale) * pow(2.0, ((double) c->qp + (double) off - 12.0) / 3.0);
  if (c->dep_quant) out->lambda *= pow(2.0, 0.25 / 3.0);
  const int offs[2] = { c->cb_qp_offset, c->cr_qp_offset };
  for (int k = 0; k < 2; k++) {
    const int mapped = table[c->qp + off];                                    // getMappedChromaQpValue(compID, qp), JVET_O0650
    int q = mapped + offs[k]; q = q < -off ? -off : q > 63 ? 63 : q;           // QpParam (CL/Quant.cpp:95-97): map, add the offset, clip
    out->qp_c[k] = q;
    out->dist_weight[k] = pow(2.0, ((double) c->qp - (double) (mapped + offs[k])) / 3.0);      // setUpLambda 120-126: unclipped
    if (c->dep_quant) out->dist_weight[k] *= (c->gop_size >= 8 ? pow(2.0, 0.1 / 3.0) : pow(2.0, 0.2 / 3.0));      // 127-130 (LFNST off)
  }
  return VVCX_OK;
}

// The partition forest of the FAST_ALGORITHM path (the reference: joblib.load("Partition_32.pkl").predict, BIN/TEST.py:21-25), as
// flattened sklearn tree arrays; validated here so that the device walk cannot leave the arrays or loop.
extern "C" int vvcx_set_forest(vvcx_handle *h, int n_trees, int n_nodes, int n_classes, const int32_t *root, const int32_t *feature, const double *threshold,
                               const int32_t *left, const int32_t *right, const double *value, const int32_t *classes)
{
  NOT_PENDING(h);
  if (!h || !root || !feature || !threshold || !left || !right || !value || !classes) return fail(VVCX_ERR_ARG, "null argument");
  if (n_trees < 1 || n_nodes < n_trees || n_classes < 1 || n_classes > 8) return fail(VVCX_ERR_ARG, "forest shape");
  std::vector<VxForestNode> nodes((size_t) n_nodes);
  for (int i = 0; i < n_nodes; i++) {
    if (left[i] >= 0) {
      // children of sklearn trees always follow their parent in the node array: a walk therefore ends
      if (left[i] <= i || right[i] <= i || left[i] >= n_nodes || right[i] >= n_nodes || feature[i] < 0 || feature[i] >= 26) return fail(VVCX_ERR_ARG, "forest node %d: child or feature index out of range", i);
    }
    nodes[(size_t) i].thr = threshold[i]; nodes[(size_t) i].left = left[i] >= 0 ? left[i] : -1; nodes[(size_t) i].right = right[i]; nodes[(size_t) i].feature = left[i] >= 0 ? feature[i] : 0; nodes[(size_t) i].pad = 0;
  }
  for (int t = 0; t < n_trees; t++) if (root[t] < 0 || root[t] >= n_nodes) return fail(VVCX_ERR_ARG, "forest root %d out of range", t);
  for (int c = 0; c < n_classes; c++) if (classes[c] < 0 || classes[c] > 5) return fail(VVCX_ERR_ARG, "forest class label %d", classes[c]);
  HIPCHK(hipSetDevice(h->cfg.device));
  (void) hipFree(h->f_node_d); (void) hipFree(h->f_value_d); (void) hipFree(h->f_root_d); h->f_node_d = nullptr; h->f_value_d = nullptr; h->f_root_d = nullptr; h->f_ntrees = 0;
  HIPCHK(hipMalloc((void **) &h->f_node_d, sizeof(VxForestNode) * (size_t) n_nodes));
  HIPCHK(hipMalloc((void **) &h->f_value_d, sizeof(double) * (size_t) n_nodes * (size_t) n_classes));
  HIPCHK(hipMalloc((void **) &h->f_root_d, sizeof(int32_t) * (size_t) n_trees));
  HIPCHK(hipMemcpy(h->f_node_d, nodes.data(), sizeof(VxForestNode) * (size_t) n_nodes, hipMemcpyHostToDevice));
  HIPCHK(hipMemcpy(h->f_value_d, value, sizeof(double) * (size_t) n_nodes * (size_t) n_classes, hipMemcpyHostToDevice));
  HIPCHK(hipMemcpy(h->f_root_d, root, sizeof(int32_t) * (size_t) n_trees, hipMemcpyHostToDevice));
  h->f_ntrees = n_trees; h->f_nclasses = n_classes;
  for (int c = 0; c < 8; c++) h->f_classes[c] = c < n_classes ? classes[c] : 0;
  return VVCX_OK;
}

extern "C" int vvcx_set_slice(vvcx_handle *h, const vvcx_slice *s)
{
  NOT_PENDING(h);
  if (!h || !s) return fail(VVCX_ERR_ARG, "null argument");
  if (!(s->lambda > 0.0) || s->qp < -6 * (h->cfg.bit_depth - 8) || s->qp > 63) return fail(VVCX_ERR_ARG, "bad slice parameters (QP range -QpBDOffset..63, like vvcx_derive_slice)");
  bool lmcs = false;
  // what the bound pictures were prepared with (vvcx_bind_frames): start contexts of the slice QP, the forward-mapped luma and the device LUTs of the LMCS model
  const bool was_bound = h->have_slice && h->n_frames > 0, lmcs_before = h->lmcs_on; const int qp_before = h->sl.qp;
  int16_t fwd_before[1024]; if (was_bound && lmcs_before) memcpy(fwd_before, h->lmcs_fwd, sizeof fwd_before);
  if (s->lmcs_enable) {
    // the piece-wise linear model of the slice -> LUTs (Reshape::constructReshaper, CL/Reshape.cpp:297-333, JVET_O0428 form: 16 input bins of equal width,
    // bin i owns orgCW + delta code words in the mapped domain; 11 fractional bits for the slopes; the inverse slope of a bin is also its chroma residual scale)
    if (!(h->cfg.tools & VVCX_TOOL_LMCS)) return fail(VVCX_ERR_ARG, "the slice enables LMCS, the handle's tool set does not");
    const int n = 1 << h->cfg.bit_depth, orgCW = n >> 4, lg = h->cfg.bit_depth - 4;
    if (s->lmcs_min_bin < 0 || s->lmcs_max_bin > 15 || s->lmcs_min_bin > s->lmcs_max_bin) return fail(VVCX_ERR_ARG, "LMCS bin range %d..%d", s->lmcs_min_bin, s->lmcs_max_bin);
    int cw[16], fwdSlope[16], invSlope[16], total = 0;
    for (int b = 0; b < 16; b++) {
      cw[b] = (b < s->lmcs_min_bin || b > s->lmcs_max_bin) ? 0 : orgCW + s->lmcs_delta_cw[b];
      if (cw[b] < 0 || cw[b] > 0xffff) return fail(VVCX_ERR_ARG, "LMCS bin %d: %d code words", b, cw[b]);
      total += cw[b];
    }
    if (total > n - 1) return fail(VVCX_ERR_ARG, "LMCS model uses %d code words of %d", total, n - 1);
    h->lmcs_pivot[0] = 0;
    for (int b = 0; b < 16; b++) {
      h->lmcs_pivot[b + 1] = h->lmcs_pivot[b] + cw[b];
      fwdSlope[b] = (cw[b] * 2048 + (1 << (lg - 1))) >> lg;
      invSlope[b] = cw[b] ? orgCW * 2048 / cw[b] : 0;
      h->lmcs_cadj[b] = cw[b] ? invSlope[b] : 2048;
    }
    for (int v = 0; v < n; v++) {
      const int b = v >> lg;
      int t = h->lmcs_pivot[b] + ((fwdSlope[b] * (v - b * orgCW) + 1024) >> 11);
      h->lmcs_fwd[v] = (int16_t) (t < 0 ? 0 : t > n - 1 ? n - 1 : t);
      int k = s->lmcs_min_bin;                           // getPWLIdxInv 268-283: the first bin whose upper border lies above v
      while (k <= s->lmcs_max_bin && v >= h->lmcs_pivot[k + 1]) k++;
      if (k > 15) k = 15;
      t = k * orgCW + ((invSlope[k] * (v - h->lmcs_pivot[k]) + 1024) >> 11);
      h->lmcs_inv[v] = (int16_t) (t < 0 ? 0 : t > n - 1 ? n - 1 : t);
    }
    lmcs = true;
  }
  h->sl = *s; h->have_slice = true; h->lmcs_on = lmcs;
  // a slice whose QP or LMCS model differs from the one the pictures were bound with unbinds them: searching them would start from the wrong contexts, on unmapped
  // (or differently mapped) luma and with LUTs that are not on the device.  The caller binds again (header: "before vvcx_bind_frames").
  if (was_bound && (qp_before != s->qp || lmcs_before != lmcs || (lmcs && memcmp(fwd_before, h->lmcs_fwd, (size_t) 2 << h->cfg.bit_depth) != 0))) h->n_frames = 0;
  return VVCX_OK;
}

extern "C" int vvcx_lmcs_tables(vvcx_handle *h, int16_t *fwd, int16_t *inv, int32_t pivot[17], int32_t chroma_scale[16])
{
  if (!h || !fwd || !inv || !pivot || !chroma_scale) return fail(VVCX_ERR_ARG, "null argument");
  if (!h->have_slice || !h->lmcs_on) return fail(VVCX_ERR_STATE, "the slice does not enable LMCS");
  memcpy(fwd, h->lmcs_fwd, (size_t) 2 << h->cfg.bit_depth); memcpy(inv, h->lmcs_inv, (size_t) 2 << h->cfg.bit_depth);
  memcpy(pivot, h->lmcs_pivot, sizeof h->lmcs_pivot); memcpy(chroma_scale, h->lmcs_cadj, sizeof h->lmcs_cadj);
  return VVCX_OK;
}

// CtxStore::init for an I slice (CL/Contexts.cpp:135-151 JVET_O0065 form, 1818-1833)
static void ctx_init_islice(int qp, uint16_t *s0, uint16_t *s1)
{
  qp = qp < 0 ? 0 : qp > 63 ? 63 : qp;
  for (int k = 0; k < VXD_NUM_CTX; k++) {
    const int id = VX_CTX_INIT_I[k];
    const int slope = (id >> 3) - 4, offset = ((id & 7) * 18) + 1;
    int st = ((slope * (qp - 16)) >> 1) + offset;
    st = st < 1 ? 1 : st > 127 ? 127 : st;
    const int p1 = st << 8;
    s0[k] = (uint16_t) (p1 & 0x7FE0); s1[k] = (uint16_t) (p1 & 0x7FFE);
  }
}

static_assert(VXD_NUM_CTX == VX_NUM_CTX, "vvcx_dev.h and the generated tables disagree on the number of context models");
static void ctx_from_reference_order(const uint16_t *s0, const uint16_t *s1, uint16_t *kept)      // kept: s0[VXD_NUM_CTX] | s1[VXD_NUM_CTX]
{
  for (int k = 0; k < VXD_NUM_CTX; k++) { kept[k] = s0[VX_CTX_REF_INDEX[k]]; kept[VXD_NUM_CTX + k] = s1[VX_CTX_REF_INDEX[k]]; }
}

extern "C" int vvcx_bind_frames(vvcx_handle *h, const vvcx_frame *frames, int n)
{
  NOT_PENDING(h);
  if (!h || !frames) return fail(VVCX_ERR_ARG, "null argument");
  DevGuard guard(h->cfg.device);
  if (n < 1 || n > h->cfg.max_frames) return fail(VVCX_ERR_ARG, "n_frames %d outside 1..%d", n, h->cfg.max_frames);
  if (!h->have_slice) return fail(VVCX_ERR_STATE, "vvcx_set_slice must precede vvcx_bind_frames");
  h->frames_h.resize((size_t) n);
  for (int f = 0; f < n; f++) {
    VxFrameDev &d = h->frames_h[(size_t) f];
    for (int c = 0; c < 3; c++) {
      if (!frames[f].org[c] || !frames[f].reco[c]) return fail(VVCX_ERR_ARG, "frame %d plane %d is null", f, c);
      d.org[c] = frames[f].org[c]; d.rec[c] = frames[f].reco[c]; d.stride[c] = frames[f].stride[c];
      if (d.stride[c] < (c ? h->cfg.pic_w >> 1 : h->cfg.pic_w)) return fail(VVCX_ERR_ARG, "frame %d plane %d stride too small", f, c);
      const size_t bps = h->cfg.bit_depth == 8 ? 1 : 2;   // the deblocking kernel loads four samples at a time: strides and bases must keep that alignment
      if ((d.stride[c] & 3) || (((uintptr_t) d.org[c] | (uintptr_t) d.rec[c]) & (4 * bps - 1))) return fail(VVCX_ERR_ARG, "frame %d plane %d: base address / stride not aligned to 4 samples", f, c);
    }
    int16_t *lev = h->lev_d + (size_t) f * h->lev_frame;
    d.lev[0] = lev; d.lev[1] = lev + h->lev_plane[0]; d.lev[2] = lev + h->lev_plane[0] + h->lev_plane[1];
    d.lstride[0] = h->cfg.pic_w; d.lstride[1] = d.lstride[2] = h->cfg.pic_w >> 1;
    d.units[0] = h->units_d + (size_t) f * h->units_frame; d.units[1] = d.units[0] + h->units_plane;
  }
  h->lmcs_inverted = false;
  if (h->lmcs_on) {
    // EncGOP::xPicInitLMCS (EL/EncGOP.cpp:1689-1695): the original luma of an intra picture is forward mapped once, before the slice is compressed; the search
    // reads the mapped copy (same stride as the caller's plane: the picture record has one stride per component)
    const size_t bps = h->cfg.bit_depth == 8 ? 1 : 2;
    size_t need = 0;
    for (int f = 0; f < n; f++) need += (size_t) h->frames_h[(size_t) f].stride[0] * h->cfg.pic_h * bps;
    if (need > h->lmcs_org_cap) { (void) hipFree(h->lmcs_org_d); h->lmcs_org_d = nullptr; h->lmcs_org_cap = 0; HIPCHK(hipMalloc(&h->lmcs_org_d, need)); h->lmcs_org_cap = need; }
    if (!h->lmcs_lut_d) HIPCHK(hipMalloc((void **) &h->lmcs_lut_d, 2 * 1024 * 2));
    HIPCHK(hipMemcpy(h->lmcs_lut_d, h->lmcs_fwd, 2048, hipMemcpyHostToDevice)); HIPCHK(hipMemcpy(h->lmcs_lut_d + 1024, h->lmcs_inv, 2048, hipMemcpyHostToDevice));
    size_t off = 0;
    for (int f = 0; f < n; f++) {
      VxFrameDev &d = h->frames_h[(size_t) f];
      void *dst = (uint8_t *) h->lmcs_org_d + off;
      const unsigned blocks = (unsigned) (((size_t) h->cfg.pic_w * h->cfg.pic_h + VXD_NT - 1) / VXD_NT);
      if (bps == 1) hipLaunchKernelGGL(vvcx_lmcs_map_kernel_u8, dim3(blocks), dim3(VXD_NT), 0, 0, (const uint8_t *) d.org[0], (uint8_t *) dst, h->cfg.pic_w, h->cfg.pic_h, d.stride[0], h->lmcs_lut_d);
      else hipLaunchKernelGGL(vvcx_lmcs_map_kernel_u16, dim3(blocks), dim3(VXD_NT), 0, 0, (const uint16_t *) d.org[0], (uint16_t *) dst, h->cfg.pic_w, h->cfg.pic_h, d.stride[0], h->lmcs_lut_d);
      HIPCHK(hipGetLastError());
      d.org[0] = dst;
      off += (size_t) d.stride[0] * h->cfg.pic_h * bps;
    }
  }
  HIPCHK(hipMemcpy(h->frames_d, h->frames_h.data(), sizeof(VxFrameDev) * (size_t) n, hipMemcpyHostToDevice));
  if (h->cfg.tools & VVCX_TOOL_JCCR) {                    // the slice's joint_cb_cr_sign_flag from the bound picture (EL/EncSlice.cpp:1594-1597), left in its record
    if (h->cfg.bit_depth == 8) hipLaunchKernelGGL(vvcx_jccr_sign_kernel_u8, dim3((unsigned) n), dim3(VXD_NT), 0, 0, h->frames_d, h->cfg.pic_w >> 1, h->cfg.pic_h >> 1);
    else hipLaunchKernelGGL(vvcx_jccr_sign_kernel_u16, dim3((unsigned) n), dim3(VXD_NT), 0, 0, h->frames_d, h->cfg.pic_w >> 1, h->cfg.pic_h >> 1);
    HIPCHK(hipGetLastError());
  }
  HIPCHK(hipMemset(h->units_d, 0, h->units_frame * sizeof(VxUnit) * (size_t) n));
  HIPCHK(hipMemset(h->lev_d, 0, h->lev_frame * 2 * (size_t) n));
  std::vector<uint16_t> ctx((size_t) n * h->nsub * 2 * VXD_NUM_CTX);
  for (size_t s = 0; s < (size_t) n * h->nsub; s++) ctx_init_islice(h->sl.qp, &ctx[s * 2 * VXD_NUM_CTX], &ctx[s * 2 * VXD_NUM_CTX + VXD_NUM_CTX]);
  HIPCHK(hipMemcpy(h->stream_ctx_d, ctx.data(), ctx.size() * 2, hipMemcpyHostToDevice));
  {
    // activity per CTU (read back here: binding is synchronous anyway); a failure only costs the ordering
    const size_t nact = (size_t) n * h->ctus_w * h->ctus_h;
    h->activity.assign(nact, 0);
    unsigned *dact = nullptr;
    if (hipMalloc((void **) &dact, nact * 4) == hipSuccess) {
      const dim3 grid((unsigned) (h->ctus_w * h->ctus_h), (unsigned) n);
      if (h->cfg.bit_depth == 8) hipLaunchKernelGGL(vvcx_ctu_activity_kernel_u8, grid, dim3(VXD_NT), 0, 0, h->frames_d, h->cfg.pic_w, h->cfg.pic_h, h->ctus_w, dact);
      else hipLaunchKernelGGL(vvcx_ctu_activity_kernel_u16, grid, dim3(VXD_NT), 0, 0, h->frames_d, h->cfg.pic_w, h->cfg.pic_h, h->ctus_w, dact);
      if (hipGetLastError() != hipSuccess || hipMemcpy(h->activity.data(), dact, nact * 4, hipMemcpyDeviceToHost) != hipSuccess) h->activity.assign(nact, 0);
      (void) hipFree(dact);
    }
  }
  h->n_frames = n;
  if (h->wpp_progress_d) HIPCHK(hipMemset(h->wpp_progress_d, 0, (size_t) n * h->nsub * 4));
  if (h->train_n_d) HIPCHK(hipMemset(h->train_n_d, 0, 4));
  h->next_idx.assign((size_t) n * h->nsub, 0);
  return VVCX_OK;
}

extern "C" int vvcx_ctus_per_frame(const vvcx_handle *h) { return h ? h->ctus_w * h->ctus_h : 0; }

extern "C" int vvcx_resident_streams(const vvcx_handle *h)
{
  if (!h) return 0;
  DevGuard guard(h->cfg.device);
  int cus = 0, per_cu = 0;
  if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, h->cfg.device) != hipSuccess) return 0;
  const bool wpp = (h->cfg.tools & VVCX_TOOL_WPP) != 0;
  const void *k = h->cfg.bit_depth == 8 ? (wpp ? (const void *) vvcx_compress_wpp_kernel_u8 : (const void *) vvcx_compress_kernel_u8) : (wpp ? (const void *) vvcx_compress_wpp_kernel_u16 : (const void *) vvcx_compress_kernel_u16);
  if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k, VXD_NT, 0) != hipSuccess) return 0;
  { const char *e = getenv("VVCX_MAX_WG_PER_CU"); const int cap = e ? atoi(e) : 0; if (cap > 0 && cap < per_cu) per_cu = cap; }      // diagnostic: the rate against the workgroups sharing a CU
  return cus * per_cu;
}

// The enqueue half: everything up to and including the device-to-host copy of the CTU results goes onto `hip_stream`; nothing waits.
extern "C" int vvcx_submit_ctus(vvcx_handle *h, const vvcx_ctu_task *tasks, int n, void *hip_stream)
{
  NOT_PENDING(h);
  if (h && (h->cfg.tools & VVCX_TOOL_FAST) && !h->f_ntrees) return fail(VVCX_ERR_STATE, "VVCX_TOOL_FAST needs vvcx_set_forest before the first CTU");
  if (!h || (!tasks && n) || n < 0) return fail(VVCX_ERR_ARG, "bad argument");
  if (n == 0) { h->pending = true; h->pend_n = 0; h->pend_stream = (hipStream_t) hip_stream; h->pend_src.clear(); h->pend_next = h->next_idx; return VVCX_OK; }   // nothing to code: an empty submission, not an error
  if (h->n_frames == 0) return fail(VVCX_ERR_STATE, "no frames bound");
  DevGuard guard(h->cfg.device);
  hipStream_t stream = (hipStream_t) hip_stream;
  const int nctu = h->ctus_w * h->ctus_h;
  // group tasks by stream, keeping their order; validate that every stream continues in tile raster order
  std::vector<std::vector<int>> by_stream((size_t) h->n_frames * h->nsub);
  for (int i = 0; i < n; i++) {
    if (tasks[i].frame < 0 || tasks[i].frame >= h->n_frames || tasks[i].ctu_rs_addr < 0 || tasks[i].ctu_rs_addr >= nctu) return fail(VVCX_ERR_ARG, "task %d out of range", i);
    by_stream[(size_t) tasks[i].frame * h->nsub + h->ctu_sub[(size_t) tasks[i].ctu_rs_addr]].push_back(i);
  }
  std::vector<VxStreamDesc> &sd = h->pend_sd; std::vector<int32_t> &task_ctu = h->pend_task_ctu; std::vector<int> &task_src = h->pend_src;
  sd.clear(); task_ctu.clear(); task_src.clear();
  std::vector<int> &new_next = h->pend_next; new_next = h->next_idx;
  for (size_t s = 0; s < by_stream.size(); s++) {
    if (by_stream[s].empty()) continue;
    const int sub = (int) (s % (size_t) h->nsub), tile = h->sub_tile[(size_t) sub];
    VxStreamDesc d; d.frame = (int) (s / (size_t) h->nsub); d.tile = tile; d.first_task = (int) task_ctu.size(); d.n_tasks = (int) by_stream[s].size();
    d.done_before = h->next_idx[s]; d.tile_ctus = (int) h->sub_ctus[(size_t) sub].size();
    d.sub = sub; d.above = (h->cfg.tools & VVCX_TOOL_WPP) ? h->sub_above[(size_t) sub] : -1;
    for (int i : by_stream[s]) {
      const std::vector<int> &order = h->sub_ctus[(size_t) sub];
      if (new_next[s] >= (int) order.size() || order[(size_t) new_next[s]] != tasks[i].ctu_rs_addr)
        return fail(VVCX_ERR_STATE, "task %d (frame %d, CTU %d) is not the next CTU of its stream", i, tasks[i].frame, tasks[i].ctu_rs_addr);
      new_next[s]++;
      task_ctu.push_back(tasks[i].ctu_rs_addr); task_src.push_back(i);
    }
    sd.push_back(d);
  }
  // WPP: a CTU row waits, CTU by CTU, for the row above it; whatever it waits for must be coded already or be part of this launch
  if (h->cfg.tools & VVCX_TOOL_WPP)
    for (size_t s = 0; s < by_stream.size(); s++) {
      const int above = h->sub_above[s % (size_t) h->nsub];
      if (by_stream[s].empty() || above < 0) continue;
      const size_t sa = s - (s % (size_t) h->nsub) + (size_t) above;
      if (new_next[sa] < new_next[s]) return fail(VVCX_ERR_STATE, "WPP: CTU row (sub-stream %d) of frame %d would wait for CTUs of the row above that are neither coded nor submitted", (int) (s % (size_t) h->nsub), (int) (s / (size_t) h->nsub));
    }
  // longest first: the workgroups take the streams from the queue in this order; results are addressed through first_task, so the order is free.  Under WPP the rows of a
  // tile stay together and in order (a row's workgroup waits for the row above, which must therefore have been taken from the queue before it): the key is the tile's
  {
    std::vector<uint64_t> key(sd.size());
    std::vector<uint64_t> group((size_t) h->n_frames * h->ntiles, 0);
    for (size_t i = 0; i < sd.size(); i++) {
      uint64_t a = 0;
      for (int t = 0; t < sd[i].n_tasks; t++) a += h->activity.empty() ? 0 : h->activity[(size_t) sd[i].frame * nctu + (size_t) task_ctu[(size_t) sd[i].first_task + t]];
      key[i] = a; group[(size_t) sd[i].frame * h->ntiles + sd[i].tile] += a;
    }
    if (h->cfg.tools & VVCX_TOOL_WPP) for (size_t i = 0; i < sd.size(); i++) key[i] = group[(size_t) sd[i].frame * h->ntiles + sd[i].tile];
    std::vector<size_t> order(sd.size());
    for (size_t i = 0; i < order.size(); i++) order[i] = i;
    std::stable_sort(order.begin(), order.end(), [&](size_t a, size_t b) { return key[a] > key[b]; });
    std::vector<VxStreamDesc> sorted(sd.size());
    for (size_t i = 0; i < order.size(); i++) sorted[i] = sd[order[i]];
    sd.swap(sorted);
  }
  const int ns = (int) sd.size();
  if (ns > h->stream_cap) { (void) hipFree(h->streams_d); h->streams_d = nullptr; HIPCHK(hipMalloc((void **) &h->streams_d, sizeof(VxStreamDesc) * (size_t) ns)); h->stream_cap = ns; }
  if (n > h->task_cap) {
    (void) hipFree(h->task_ctu_d); (void) hipFree(h->results_d); h->task_ctu_d = nullptr; h->results_d = nullptr;
    HIPCHK(hipMalloc((void **) &h->task_ctu_d, sizeof(int32_t) * (size_t) n)); HIPCHK(hipMalloc((void **) &h->results_d, sizeof(VxCtuRes) * (size_t) n)); h->task_cap = n;
  }
  const size_t per_stream = (h->cfg.tools & VVCX_TOOL_CU_REUSE) ? (size_t) VXD_SCRATCH_BYTES : (size_t) VXD_OFF_CACHE;   // the CU cache only when used
  // one workgroup per resident stream slot (they take the streams from a queue): scratch is per slot
  int resident = vvcx_resident_streams(h);
  if (resident < 1) resident = 1;
  const int grid = ns < resident ? ns : resident;
  const size_t need = (size_t) grid * per_stream;
  if (need > h->scratch_cap) {
    (void) hipFree(h->scratch_d); h->scratch_d = nullptr; HIPCHK(hipMalloc((void **) &h->scratch_d, need)); h->scratch_cap = need;
    HIPCHK(hipMemsetAsync(h->scratch_d, 0, need, stream));       // CU-cache entries and generation counters start empty
  }
  if (h->cfg.tools & VVCX_TOOL_WPP) {                    // scheduler state of this launch: nothing finished, nothing owned, every stream with its tasks left
    if (4 + 2 * ns > h->wpp_sched_cap) { (void) hipFree(h->wpp_sched_d); h->wpp_sched_d = nullptr; HIPCHK(hipMalloc((void **) &h->wpp_sched_d, sizeof(int32_t) * (size_t) (4 + 2 * ns))); h->wpp_sched_cap = 4 + 2 * ns; }
    std::vector<int32_t> sched((size_t) (4 + 2 * ns), 0);
    for (int i = 0; i < ns; i++) sched[(size_t) (4 + ns + i)] = sd[(size_t) i].n_tasks;
    HIPCHK(hipMemcpy(h->wpp_sched_d, sched.data(), sched.size() * sizeof(int32_t), hipMemcpyHostToDevice));
  }
  HIPCHK(hipMemcpyAsync(h->streams_d, sd.data(), sizeof(VxStreamDesc) * (size_t) ns, hipMemcpyHostToDevice, stream));
  HIPCHK(hipMemcpyAsync(h->task_ctu_d, task_ctu.data(), sizeof(int32_t) * (size_t) n, hipMemcpyHostToDevice, stream));
  HIPCHK(hipMemsetAsync(h->counters_d, 0, 56 * sizeof(unsigned long long), stream));

  VxParams p; memset(&p, 0, sizeof p);
  p.pic_w = h->cfg.pic_w; p.pic_h = h->cfg.pic_h; p.bit_depth = h->cfg.bit_depth; p.chroma = h->cfg.chroma; p.tools = h->cfg.tools;
  for (int k = 0; k < 2; k++) { p.min_qt[k] = h->cfg.min_qt[k]; p.max_bt_depth[k] = h->cfg.max_bt_depth[k]; p.max_bt_size[k] = h->cfg.max_bt_size[k]; p.max_tt_size[k] = h->cfg.max_tt_size[k]; p.qp_c[k] = h->sl.qp_c[k]; p.dist_weight[k] = h->sl.dist_weight[k]; }
  p.ctus_w = h->ctus_w; p.ctus_h = h->ctus_h; p.uw = h->uw; p.uh = h->uh; p.qp = h->sl.qp;
  p.lambda = h->sl.lambda;
  { const int bdo = 6 * (h->cfg.bit_depth - 8); p.qp_tr = h->sl.qp + bdo; p.qp_tr_c[0] = h->sl.qp_c[0] + bdo; p.qp_tr_c[1] = h->sl.qp_c[1] + bdo;
    // QpParam of the modes +-2: the mapping table's value (one table for all chroma) + pps_joint_cbcr_qp_offset (CbCrQpOffset -1, APP/EncAppCfg.cpp:1082)
    int qj = h->sl.qp_c[0] - 1; qj = qj < -bdo ? -bdo : qj > 63 ? 63 : qj; p.qp_tr_j = qj + bdo; }
  p.dist_scale = (double) (1 << 15) / h->sl.lambda;                       // CL/RdCost.cpp:79
  p.sqrt_lambda_fp = sqrt(h->sl.lambda) * (1.0 / (double) (1 << 15));     // EL/IntraSearch.cpp:297
  p.frames = h->frames_d; p.streams = h->streams_d; p.task_ctu = h->task_ctu_d; p.results = h->results_d; p.stream_ctx = h->stream_ctx_d;
  p.payload = h->payload_d; p.payload_off = h->payload_off_d; p.payload_cap = h->payload_cap_d; p.arith_state = h->arith_d;
  p.scratch = h->scratch_d; p.scratch_per_stream = per_stream; p.counters = h->counters_d; p.ntiles = h->ntiles; p.nsub = h->nsub; p.wpp_progress = h->wpp_progress_d; p.wpp_sync = h->wpp_sync_d;
  p.train_rows = h->train_rows_d; p.train_n = h->train_n_d; p.train_cap = h->train_cap; p.wpp_sched = h->wpp_sched_d; p.wpp_rr = h->wpp_rr;
  p.f_node = h->f_node_d; p.f_value = h->f_value_d; p.f_root = h->f_root_d; p.f_ntrees = h->f_ntrees; p.f_nclasses = h->f_nclasses;
  for (int c = 0; c < 8; c++) p.f_classes[c] = h->f_classes[c];
  p.n_streams = ns;
  p.lmcs_on = h->lmcs_on; p.lmcs_cadj_on = h->lmcs_on && h->sl.lmcs_chroma_adj; p.lmcs_min_bin = h->sl.lmcs_min_bin; p.lmcs_max_bin = h->sl.lmcs_max_bin;
  for (int b = 0; b < 17; b++) p.lmcs_pivot[b] = h->lmcs_on ? h->lmcs_pivot[b] : 0;
  for (int b = 0; b < 16; b++) p.lmcs_cadj[b] = h->lmcs_on ? h->lmcs_cadj[b] : 0;
  if (h->cfg.tools & VVCX_TOOL_DEPQUANT) {
    // the quantiser's lambda of a component: TrQuant::setLambdas / selectLambda (EL/EncSlice.cpp:107-149, EL/IntraSearch.cpp:2889) = lambda / distortion weight for chroma
    VxDqConst *tab = h->pend_dq; memset(tab, 0, sizeof h->pend_dq);
    const bool jccr = (h->cfg.tools & VVCX_TOOL_JCCR) != 0;
    for (int comp = 0; comp < 3; comp++) {
      double lam = comp ? h->sl.lambda / h->sl.dist_weight[comp - 1] : h->sl.lambda;
      if (comp && jccr && h->sl.qp > 18) lam = 1.3 * lam;          // EL/IntraSearch.cpp:2937-2942: every chroma block once JointCbCr is on
      const int qp = comp ? p.qp_tr_c[comp - 1] : p.qp_tr;
      for (int lsum = 2; lsum <= 12; lsum++) tab[comp * 16 + lsum] = dq_consts_of(lsum, h->cfg.bit_depth, qp, lam);
    }
    if (jccr) for (int mask = 1; mask <= 3; mask++) {              // joint blocks: the Cb lambda loosened by 0.8 (modes +-1, +-3) or 0.5 (+-2), QP of the coded component or the JointCbCr QP
      double lam = (mask == 3 ? 0.5 : 0.8) * (h->sl.lambda / h->sl.dist_weight[0]);
      if (h->sl.qp > 18) lam = 1.3 * lam;
      const int qp = mask == 3 ? p.qp_tr_j : p.qp_tr_c[(mask >> 1) ? 0 : 1];
      for (int lsum = 2; lsum <= 12; lsum++) tab[(2 + mask) * 16 + lsum] = dq_consts_of(lsum, h->cfg.bit_depth, qp, lam);
    }
    if (h->lmcs_on && h->sl.lmcs_chroma_adj) {
      // chroma residual scaling: the quantiser's lambda of a scaled chroma block is first divided by the square of 2048 / scale (EL/IntraSearch.cpp:2919-2931), one
      // table per bin of the model; the luma row is never read from these tables
      for (int b = 0; b < 16; b++) {
        const double cResScale = 2048.0 / (double) h->lmcs_cadj[b], cbBase = (h->sl.lambda / h->sl.dist_weight[0]) / (cResScale * cResScale);
        VxDqConst *tb = tab + (1 + b) * 96;
        for (int comp = 1; comp < 3; comp++) {
          double lam = (h->sl.lambda / h->sl.dist_weight[comp - 1]) / (cResScale * cResScale);
          if (jccr && h->sl.qp > 18) lam = 1.3 * lam;
          for (int lsum = 2; lsum <= 12; lsum++) tb[comp * 16 + lsum] = dq_consts_of(lsum, h->cfg.bit_depth, p.qp_tr_c[comp - 1], lam);
        }
        if (jccr) for (int mask = 1; mask <= 3; mask++) {
          double lam = (mask == 3 ? 0.5 : 0.8) * cbBase;
          if (h->sl.qp > 18) lam = 1.3 * lam;
          const int qp = mask == 3 ? p.qp_tr_j : p.qp_tr_c[(mask >> 1) ? 0 : 1];
          for (int lsum = 2; lsum <= 12; lsum++) tb[(2 + mask) * 16 + lsum] = dq_consts_of(lsum, h->cfg.bit_depth, qp, lam);
        }
      }
    }
    HIPCHK(hipMemcpyAsync(h->dq_d, tab, sizeof h->pend_dq, hipMemcpyHostToDevice, stream));
    p.dq_consts = h->dq_d;
  }

  HIPCHK(hipEventRecord(h->ev0, stream));
  if (h->cfg.tools & VVCX_TOOL_WPP) {
    if (h->cfg.bit_depth == 8) hipLaunchKernelGGL(vvcx_compress_wpp_kernel_u8, dim3((unsigned) grid), dim3(VXD_NT), 0, stream, p);
    else hipLaunchKernelGGL(vvcx_compress_wpp_kernel_u16, dim3((unsigned) grid), dim3(VXD_NT), 0, stream, p);
  } else if (h->cfg.bit_depth == 8) hipLaunchKernelGGL(vvcx_compress_kernel_u8, dim3((unsigned) grid), dim3(VXD_NT), 0, stream, p);
  else hipLaunchKernelGGL(vvcx_compress_kernel_u16, dim3((unsigned) grid), dim3(VXD_NT), 0, stream, p);
  HIPCHK(hipGetLastError());
  HIPCHK(hipEventRecord(h->ev1, stream));
  if (n > h->pend_cap) {                               // pinned, so that the copy back does not block the caller
    (void) hipHostFree(h->pend_res); h->pend_res = nullptr; h->pend_cap = 0;
    HIPCHK(hipHostMalloc((void **) &h->pend_res, sizeof(VxCtuRes) * (size_t) n, 0)); h->pend_cap = n;
  }
  HIPCHK(hipMemcpyAsync(h->pend_res, h->results_d, sizeof(VxCtuRes) * (size_t) n, hipMemcpyDeviceToHost, stream));
  h->pending = true; h->pend_stream = stream; h->pend_n = n;
  return VVCX_OK;
}

// 1 once the submitted launch and its copy back have finished (vvcx_wait_ctus will not block), 0 while it runs, < 0 on error
extern "C" int vvcx_poll_ctus(vvcx_handle *h)
{
  if (!h) return fail(VVCX_ERR_ARG, "null handle");
  if (!h->pending) return fail(VVCX_ERR_STATE, "nothing submitted");
  if (h->pend_n == 0) return 1;
  DevGuard guard(h->cfg.device);
  const hipError_t e = hipStreamQuery(h->pend_stream);
  if (e == hipSuccess) return 1;
  if (e == hipErrorNotReady) return 0;
  return fail(VVCX_ERR_DEVICE, "hipStreamQuery: %s", hipGetErrorString(e));
}

// The collecting half: waits for the stream, advances the streams' positions and hands the results over in task order.
extern "C" int vvcx_wait_ctus(vvcx_handle *h, vvcx_ctu_result *out, int n)
{
  if (!h) return fail(VVCX_ERR_ARG, "null handle");
  if (!h->pending) return fail(VVCX_ERR_STATE, "nothing submitted");
  if (n != h->pend_n || (!out && n)) return fail(VVCX_ERR_ARG, "vvcx_wait_ctus: %d results asked, %d tasks submitted", n, h->pend_n);
  h->pending = false;
  if (n == 0) return VVCX_OK;
  DevGuard guard(h->cfg.device);
  HIPCHK(hipStreamSynchronize(h->pend_stream));
  HIPCHK(hipEventElapsedTime(&h->last_ms, h->ev0, h->ev1));
  if (h->cfg.tools & VVCX_TOOL_WPP) {
    int32_t st[2] = { 0, 0 };
    HIPCHK(hipMemcpy(st, h->wpp_sched_d, sizeof st, hipMemcpyDeviceToHost));
    if (st[1]) {
      // the rows' contexts, coder states and progress counts on the device have advanced part of the way; the host's positions have not: the streams cannot be continued
      h->n_frames = 0;
      return fail(VVCX_ERR_DEVICE, "WPP scheduler gave up: %d of the launch's CTU rows finished, the others never became ready (the pictures are unbound: bind them again)", st[0]);
    }
  }
  h->next_idx = h->pend_next;
  const VxCtuRes *res = h->pend_res;
  for (int k = 0; k < n; k++) {
    vvcx_ctu_result &o = out[h->pend_src[(size_t) k]];
    o.dist = res[k].dist; o.frac_bits = res[k].bits; o.cost = res[k].cost; o.n_cu = res[k].n_cu;
    if (!(res[k].cost < 1.7e+308)) return fail(VVCX_ERR_NO_ENCODING, "no possible encoding found for task %d (EL/EncCu.cpp:555-557)", h->pend_src[(size_t) k]);
  }
  return VVCX_OK;
}

extern "C" int vvcx_compress_ctus(vvcx_handle *h, const vvcx_ctu_task *tasks, int n, vvcx_ctu_result *out, void *hip_stream)
{
  if (h && n == 0 && !h->pending) return (h->cfg.tools & VVCX_TOOL_FAST) && !h->f_ntrees ? fail(VVCX_ERR_STATE, "VVCX_TOOL_FAST needs vvcx_set_forest before the first CTU") : VVCX_OK;
  if (!out) return fail(VVCX_ERR_ARG, "bad argument");
  const int rc = vvcx_submit_ctus(h, tasks, n, hip_stream);
  return rc != VVCX_OK ? rc : vvcx_wait_ctus(h, out, n);
}

extern "C" int vvcx_compress_bound_frames(vvcx_handle *h, vvcx_ctu_result *out, void *hip_stream)
{
  NOT_PENDING(h);
  if (!h || !out) return fail(VVCX_ERR_ARG, "null argument");
  if (h->n_frames == 0) return fail(VVCX_ERR_STATE, "no frames bound");
  const int nctu = h->ctus_w * h->ctus_h;
  std::vector<vvcx_ctu_task> tasks; std::vector<int> dst;
  for (int f = 0; f < h->n_frames; f++) for (int t = 0; t < h->nsub; t++) {
    const std::vector<int> &order = h->sub_ctus[(size_t) t];
    for (size_t k = (size_t) h->next_idx[(size_t) f * h->nsub + t]; k < order.size(); k++) { vvcx_ctu_task tk; tk.frame = f; tk.ctu_rs_addr = order[k]; tasks.push_back(tk); dst.push_back(f * nctu + order[k]); }
  }
  std::vector<vvcx_ctu_result> tmp(tasks.size());
  const int rc = vvcx_compress_ctus(h, tasks.data(), (int) tasks.size(), tmp.data(), hip_stream);
  if (rc != VVCX_OK) return rc;
  for (size_t k = 0; k < tasks.size(); k++) out[dst[k]] = tmp[k];
  return VVCX_OK;
}

// LMCS: luma reconstruction of every bound picture back to the original domain, in place (the picture-level inverse rspSignal in front of the loop filters)
extern "C" int vvcx_lmcs_inverse_reco(vvcx_handle *h, void *hip_stream)
{
  NOT_PENDING(h);
  if (!h) return fail(VVCX_ERR_ARG, "null handle");
  if (!h->n_frames || !h->have_slice || !h->lmcs_on) return fail(VVCX_ERR_STATE, "no bound frames coded with an LMCS slice");
  if (!h->lmcs_lut_d) return fail(VVCX_ERR_STATE, "the bound pictures were not prepared with an LMCS slice: vvcx_bind_frames after vvcx_set_slice");
  if (h->lmcs_inverted) return fail(VVCX_ERR_STATE, "the reconstruction has already been mapped back");
  for (size_t i = 0; i < h->next_idx.size(); i++)
    if (h->next_idx[i] != (int) h->sub_ctus[i % (size_t) h->nsub].size()) return fail(VVCX_ERR_STATE, "every CTU of the bound pictures must be coded first (intra prediction reads mapped neighbours)");
  HIPCHK(hipSetDevice(h->cfg.device));
  hipStream_t stream = (hipStream_t) hip_stream;
  const unsigned blocks = (unsigned) (((size_t) h->cfg.pic_w * h->cfg.pic_h + VXD_NT - 1) / VXD_NT);
  for (int f = 0; f < h->n_frames; f++) {
    const VxFrameDev &d = h->frames_h[(size_t) f];
    if (h->cfg.bit_depth == 8) hipLaunchKernelGGL(vvcx_lmcs_map_kernel_u8, dim3(blocks), dim3(VXD_NT), 0, stream, (const uint8_t *) d.rec[0], (uint8_t *) d.rec[0], h->cfg.pic_w, h->cfg.pic_h, d.stride[0], h->lmcs_lut_d + 1024);
    else hipLaunchKernelGGL(vvcx_lmcs_map_kernel_u16, dim3(blocks), dim3(VXD_NT), 0, stream, (const uint16_t *) d.rec[0], (uint16_t *) d.rec[0], h->cfg.pic_w, h->cfg.pic_h, d.stride[0], h->lmcs_lut_d + 1024);
    HIPCHK(hipGetLastError());
  }
  HIPCHK(hipStreamSynchronize(stream));
  h->lmcs_inverted = true;
  return VVCX_OK;
}

// ≙ LoopFilter::loopFilterPic (CL/LoopFilter.cpp:153) on every bound picture, in place on the reconstruction planes the search wrote:
// one launch for all vertical edges, one for all horizontal edges (vvcx_deblock.hip).  Every CTU of the pictures must have been coded.
extern "C" int vvcx_deblock_bound_frames(vvcx_handle *h, int beta_offset_div2, int tc_offset_div2, void *hip_stream)
{
  NOT_PENDING(h);
  if (!h) return fail(VVCX_ERR_ARG, "null handle");
  if (!h->n_frames || !h->have_slice) return fail(VVCX_ERR_STATE, "no bound frames / slice");
  for (size_t i = 0; i < h->next_idx.size(); i++)
    if (h->next_idx[i] != (int) h->sub_ctus[i % (size_t) h->nsub].size()) return fail(VVCX_ERR_STATE, "deblocking needs every CTU of the bound pictures coded (frame %d tile %d is not)", (int) (i / (size_t) h->nsub), h->sub_tile[i % (size_t) h->nsub]);
  if (h->lmcs_on && !h->lmcs_inverted) return fail(VVCX_ERR_STATE, "LMCS slice: vvcx_lmcs_inverse_reco first (the loop filters work in the original domain)");
  if (beta_offset_div2 < -6 || beta_offset_div2 > 6 || tc_offset_div2 < -6 || tc_offset_div2 > 6) return fail(VVCX_ERR_ARG, "deblocking offsets outside -6..6");
  HIPCHK(hipSetDevice(h->cfg.device));
  hipStream_t stream = (hipStream_t) hip_stream;
  VxDeblockParams p; memset(&p, 0, sizeof p);
  p.frames = h->frames_d; p.uw = h->uw; p.uh = h->uh; p.bit_depth = h->cfg.bit_depth; p.chroma = h->cfg.chroma;
  p.qp = h->sl.qp; p.qp_c[0] = h->sl.qp_c[0]; p.qp_c[1] = h->sl.qp_c[1]; p.beta_off2 = beta_offset_div2; p.tc_off2 = tc_offset_div2;
  const dim3 grid((unsigned) ((2 * h->uw * h->uh + 255) / 256), (unsigned) h->n_frames);
  const size_t nedge = (size_t) h->n_frames * 4 * h->uw * h->uh;
  if (h->db_edges_cap < nedge) { (void) hipFree(h->db_edges_d); h->db_edges_d = nullptr; h->db_edges_cap = nedge; HIPCHK(hipMalloc((void **) &h->db_edges_d, nedge)); }
  p.edges = h->db_edges_d;
  HIPCHK(hipEventRecord(h->ev0, stream));
  hipLaunchKernelGGL(vvcx_deblock_edges_kernel, grid, dim3(256), 0, stream, p);      // the unit records -> one edge byte per unit, map and direction
  HIPCHK(hipGetLastError());
  for (int dir = 0; dir < 2; dir++) {
    p.dir = dir;
    if (h->cfg.bit_depth == 8) hipLaunchKernelGGL(vvcx_deblock_kernel_u8, grid, dim3(256), 0, stream, p);
    else hipLaunchKernelGGL(vvcx_deblock_kernel_u16, grid, dim3(256), 0, stream, p);
    HIPCHK(hipGetLastError());
  }
  HIPCHK(hipEventRecord(h->ev1, stream));
  HIPCHK(hipStreamSynchronize(stream));
  HIPCHK(hipEventElapsedTime(&h->last_deblock_ms, h->ev0, h->ev1));
  return VVCX_OK;
}
extern "C" float vvcx_last_deblock_ms(const vvcx_handle *h) { return h ? h->last_deblock_ms : 0.f; }

// ≙ SampleAdaptiveOffset::SAOProcess (CL/SampleAdaptiveOffset.cpp:617-670) with the caller's parameters prm[frame][ctu][component]: merges are resolved here
// (xReconstructBlkSAOParams 265-290: the CTU to the left / above in the same tile, in raster order; invertQuantOffsets 147-170), the samples are filtered on the device
// (vvcx_sao.hip) in place on the reconstruction planes.
static int sao_resolve(const vvcx_sao_param *prm, int n_frames, int cw, int chh, int tc, int tr, int log2_offset_scale, std::vector<VxSaoEntry> &tab, std::vector<uint8_t> &tile)
{
  const int nctu = cw * chh;
  tile.resize((size_t) nctu);
  for (int a = 0; a < nctu; a++) {
    int tx = 0, ty = 0;
    for (int i = 0; i < tc; i++) if (a % cw >= (i * cw) / tc) tx = i;
    for (int i = 0; i < tr; i++) if (a / cw >= (i * chh) / tr) ty = i;
    tile[(size_t) a] = (uint8_t) (ty * tc + tx);
  }
  tab.resize((size_t) n_frames * nctu * 3);
  for (int f = 0; f < n_frames; f++) for (int a = 0; a < nctu; a++) for (int c = 0; c < 3; c++) {
    const vvcx_sao_param &p = prm[((size_t) f * nctu + a) * 3 + c];
    VxSaoEntry &e = tab[((size_t) f * nctu + a) * 3 + c];
    memset(&e, 0, sizeof e); e.type = -1;
    if (p.mode == 0) continue;
    if (p.mode == 1) {
      if (p.type < 0 || p.type > 4 || p.band < 0 || p.band > 31) return fail(VVCX_ERR_ARG, "SAO parameters of frame %d CTU %d component %d: type %d band %d", f, a, c, p.type, p.band);
      e.type = p.type; e.band = p.type == 4 ? p.band : 0;
      for (int i = 0; i < 4; i++) e.off[i] = (int16_t) (p.offset[i] * (1 << log2_offset_scale));
    } else if (p.mode == 2) {
      const int left = p.type == 0, s = left ? a - 1 : a - cw;
      if ((p.type != 0 && p.type != 1) || (left ? a % cw == 0 : a < cw) || tile[(size_t) s] != tile[(size_t) a])
        return fail(VVCX_ERR_ARG, "SAO parameters of frame %d CTU %d component %d: no %s merge candidate in the tile", f, a, c, left ? "left" : "above");
      e = tab[((size_t) f * nctu + s) * 3 + c];
    } else return fail(VVCX_ERR_ARG, "SAO mode %d", p.mode);
  }
  return VVCX_OK;
}
static void sao_launch(VxSaoParams &p, int n_frames, size_t bps, hipStream_t stream)
{
  const dim3 grid((unsigned) ((p.pic_w + 1023) / 1024), (unsigned) p.pic_h, (unsigned) (3 * n_frames));      // the filter: four samples per lane
  const dim3 gridC((unsigned) ((p.pic_w + 1023) / 1024), (unsigned) ((p.pic_h + 3) / 4), (unsigned) (3 * n_frames));      // the copy: strips of 4 rows x 1024 samples
  if (bps == 1) { hipLaunchKernelGGL(vvcx_sao_copy_kernel_u8, gridC, dim3(256), 0, stream, p); hipLaunchKernelGGL(vvcx_sao_kernel_u8, grid, dim3(256), 0, stream, p); }
  else { hipLaunchKernelGGL(vvcx_sao_copy_kernel_u16, gridC, dim3(256), 0, stream, p); hipLaunchKernelGGL(vvcx_sao_kernel_u16, grid, dim3(256), 0, stream, p); }
}
extern "C" int vvcx_sao_bound_frames(vvcx_handle *h, const vvcx_sao_param *prm, int lf_across_tiles, int log2_offset_scale, void *hip_stream)
{
  NOT_PENDING(h);
  if (!h || !prm) return fail(VVCX_ERR_ARG, "null argument");
  if (!h->n_frames || !h->have_slice) return fail(VVCX_ERR_STATE, "no bound frames / slice");
  if (log2_offset_scale < 0 || log2_offset_scale > 4) return fail(VVCX_ERR_ARG, "log2 offset scale outside 0..4");
  for (size_t i = 0; i < h->next_idx.size(); i++)
    if (h->next_idx[i] != (int) h->sub_ctus[i % (size_t) h->nsub].size()) return fail(VVCX_ERR_STATE, "the loop filters need every CTU of the bound pictures coded");
  if (h->lmcs_on && !h->lmcs_inverted) return fail(VVCX_ERR_STATE, "LMCS slice: vvcx_lmcs_inverse_reco first (the loop filters work in the original domain)");
  const int cw = h->ctus_w, chh = h->ctus_h, nctu = cw * chh;
  std::vector<VxSaoEntry> tab; std::vector<uint8_t> tile;
  const int rr = sao_resolve(prm, h->n_frames, cw, chh, h->cfg.tile_cols, h->cfg.tile_rows, log2_offset_scale, tab, tile);
  if (rr != VVCX_OK) return rr;
  DevGuard guard(h->cfg.device);
  hipStream_t stream = (hipStream_t) hip_stream;
  const size_t bps = h->cfg.bit_depth == 8 ? 1 : 2, ny = (size_t) h->cfg.pic_w * h->cfg.pic_h, per_frame = ny + 2 * (ny >> 2);
  if (h->sao_tmp_cap < (size_t) h->n_frames * per_frame * bps) { (void) hipFree(h->sao_tmp_d); h->sao_tmp_d = nullptr; h->sao_tmp_cap = (size_t) h->n_frames * per_frame * bps; HIPCHK(hipMalloc(&h->sao_tmp_d, h->sao_tmp_cap)); }
  if (h->sao_tab_cap < tab.size()) { (void) hipFree(h->sao_tab_d); h->sao_tab_d = nullptr; h->sao_tab_cap = tab.size(); HIPCHK(hipMalloc((void **) &h->sao_tab_d, tab.size() * sizeof(VxSaoEntry))); }
  if (!h->sao_tile_d) HIPCHK(hipMalloc((void **) &h->sao_tile_d, (size_t) nctu));
  HIPCHK(hipMemcpyAsync(h->sao_tab_d, tab.data(), tab.size() * sizeof(VxSaoEntry), hipMemcpyHostToDevice, stream));
  HIPCHK(hipMemcpyAsync(h->sao_tile_d, tile.data(), (size_t) nctu, hipMemcpyHostToDevice, stream));
  VxSaoParams p; memset(&p, 0, sizeof p);
  p.frames = h->frames_d; p.table = h->sao_tab_d; p.tile_of_ctu = h->sao_tile_d; p.tmp = h->sao_tmp_d; p.tmp_frame = per_frame; p.tmp_comp[0] = 0; p.tmp_comp[1] = ny; p.tmp_comp[2] = ny + (ny >> 2);
  p.pic_w = h->cfg.pic_w; p.pic_h = h->cfg.pic_h; p.ctus_w = cw; p.ctus_h = chh; p.bit_depth = h->cfg.bit_depth; p.chroma = h->cfg.chroma; p.lf_across_tiles = lf_across_tiles != 0;
  HIPCHK(hipEventRecord(h->ev0, stream));
  sao_launch(p, h->n_frames, bps, stream);
  HIPCHK(hipGetLastError());
  HIPCHK(hipEventRecord(h->ev1, stream));
  HIPCHK(hipStreamSynchronize(stream));
  HIPCHK(hipEventElapsedTime(&h->last_sao_ms, h->ev0, h->ev1));
  return VVCX_OK;
}
// the same kernels on one picture in host memory (uint16 planes, stride = plane width; filtered in place): the leaf entry the filter is pinned through
extern "C" int vvcx_sao_picture(int pic_w, int pic_h, int bit_depth, int tile_cols, int tile_rows, const vvcx_sao_param *prm, int lf_across_tiles, int log2_offset_scale,
                                uint16_t *y, uint16_t *cb, uint16_t *cr, int device)
{
  if (!prm || !y || !cb || !cr) return fail(VVCX_ERR_ARG, "null argument");
  if (pic_w < 8 || pic_h < 8 || (pic_w & 7) || (pic_h & 7) || pic_w > 16384 || pic_h > 16384 || bit_depth < 8 || bit_depth > 12 || tile_cols < 1 || tile_rows < 1 ||
      tile_cols > (pic_w + 127) / 128 || tile_rows > (pic_h + 127) / 128 || tile_cols * tile_rows > 255 || log2_offset_scale < 0 || log2_offset_scale > 4)
    return fail(VVCX_ERR_ARG, "vvcx_sao_picture: picture size / bit depth / tiles / offset scale");
  const int cw = (pic_w + 127) / 128, chh = (pic_h + 127) / 128;
  std::vector<VxSaoEntry> tab; std::vector<uint8_t> tile;
  const int rr = sao_resolve(prm, 1, cw, chh, tile_cols, tile_rows, log2_offset_scale, tab, tile);
  if (rr != VVCX_OK) return rr;
  DevGuard guard(device);
  const size_t ny = (size_t) pic_w * pic_h, nc = ny >> 2, per_frame = ny + 2 * nc;
  uint16_t *pl_d = nullptr, *tmp_d = nullptr; VxFrameDev *fd_d = nullptr; VxSaoEntry *tab_d = nullptr; uint8_t *tile_d = nullptr;
  int rc = VVCX_OK;
  if (hipMalloc((void **) &pl_d, per_frame * 2) != hipSuccess || hipMalloc((void **) &tmp_d, per_frame * 2) != hipSuccess || hipMalloc((void **) &fd_d, sizeof(VxFrameDev)) != hipSuccess ||
      hipMalloc((void **) &tab_d, tab.size() * sizeof(VxSaoEntry)) != hipSuccess || hipMalloc((void **) &tile_d, tile.size()) != hipSuccess)
    rc = fail(VVCX_ERR_DEVICE, "hipMalloc failed for the SAO of a %dx%d picture", pic_w, pic_h);
  if (rc == VVCX_OK) {
    VxFrameDev fd; memset(&fd, 0, sizeof fd);
    fd.rec[0] = pl_d; fd.rec[1] = pl_d + ny; fd.rec[2] = pl_d + ny + nc; fd.stride[0] = pic_w; fd.stride[1] = fd.stride[2] = pic_w >> 1;
    VxSaoParams p; memset(&p, 0, sizeof p);
    p.frames = fd_d; p.table = tab_d; p.tile_of_ctu = tile_d; p.tmp = tmp_d; p.tmp_frame = per_frame; p.tmp_comp[0] = 0; p.tmp_comp[1] = ny; p.tmp_comp[2] = ny + nc;
    p.pic_w = pic_w; p.pic_h = pic_h; p.ctus_w = cw; p.ctus_h = chh; p.bit_depth = bit_depth; p.chroma = 1; p.lf_across_tiles = lf_across_tiles != 0;
    bool ok = hipMemcpy(pl_d, y, ny * 2, hipMemcpyHostToDevice) == hipSuccess && hipMemcpy(pl_d + ny, cb, nc * 2, hipMemcpyHostToDevice) == hipSuccess &&
              hipMemcpy(pl_d + ny + nc, cr, nc * 2, hipMemcpyHostToDevice) == hipSuccess && hipMemcpy(fd_d, &fd, sizeof fd, hipMemcpyHostToDevice) == hipSuccess &&
              hipMemcpy(tab_d, tab.data(), tab.size() * sizeof(VxSaoEntry), hipMemcpyHostToDevice) == hipSuccess && hipMemcpy(tile_d, tile.data(), tile.size(), hipMemcpyHostToDevice) == hipSuccess;
    if (ok) {
      sao_launch(p, 1, 2, 0);
      ok = hipGetLastError() == hipSuccess && hipDeviceSynchronize() == hipSuccess && hipMemcpy(y, pl_d, ny * 2, hipMemcpyDeviceToHost) == hipSuccess &&
           hipMemcpy(cb, pl_d + ny, nc * 2, hipMemcpyDeviceToHost) == hipSuccess && hipMemcpy(cr, pl_d + ny + nc, nc * 2, hipMemcpyDeviceToHost) == hipSuccess;
    }
    if (!ok) rc = fail(VVCX_ERR_DEVICE, "SAO of a picture: a HIP call failed");
  }
  (void) hipFree(pl_d); (void) hipFree(tmp_d); (void) hipFree(fd_d); (void) hipFree(tab_d); (void) hipFree(tile_d);
  return rc;
}
extern "C" float vvcx_last_sao_ms(const vvcx_handle *h) { return h ? h->last_sao_ms : 0.f; }

// ≙ EncSampleAdaptiveOffset::getStatistics (EL/EncSampleAdaptiveOffset.cpp:284-353, SAOLcuBoundary 0) on the bound, deblocked pictures: the statistics the reference's
// decideBlkParams works from, in host memory as [frame][ctu][component][type 0..4][count | diff][32] int64 (≙ SAOStatData per CTU, component and type)
extern "C" int vvcx_sao_statistics_bound_frames(vvcx_handle *h, int lf_across_tiles, int64_t *stats, void *hip_stream)
{
  NOT_PENDING(h);
  if (!h || !stats) return fail(VVCX_ERR_ARG, "null argument");
  if (!h->n_frames || !h->have_slice) return fail(VVCX_ERR_STATE, "no bound frames / slice");
  for (size_t i = 0; i < h->next_idx.size(); i++)
    if (h->next_idx[i] != (int) h->sub_ctus[i % (size_t) h->nsub].size()) return fail(VVCX_ERR_STATE, "the loop filters need every CTU of the bound pictures coded");
  if (h->lmcs_on) return fail(VVCX_ERR_UNSUPPORTED, "SAO statistics of an LMCS slice: the handle keeps the mapped original only");
  const int cw = h->ctus_w, chh = h->ctus_h, nctu = cw * chh;
  std::vector<uint8_t> tile((size_t) nctu);
  for (int a = 0; a < nctu; a++) {
    int tx = 0, ty = 0;
    for (int i = 0; i < h->cfg.tile_cols; i++) if (a % cw >= (i * cw) / h->cfg.tile_cols) tx = i;
    for (int i = 0; i < h->cfg.tile_rows; i++) if (a / cw >= (i * chh) / h->cfg.tile_rows) ty = i;
    tile[(size_t) a] = (uint8_t) (ty * h->cfg.tile_cols + tx);
  }
  DevGuard guard(h->cfg.device);
  hipStream_t stream = (hipStream_t) hip_stream;
  const size_t n64 = (size_t) h->n_frames * nctu * 3 * 5 * 64;
  if (h->sao_stat_cap < n64) { (void) hipFree(h->sao_stat_d); h->sao_stat_d = nullptr; h->sao_stat_cap = n64; HIPCHK(hipMalloc((void **) &h->sao_stat_d, n64 * sizeof(long long))); }
  if (!h->sao_tile_d) HIPCHK(hipMalloc((void **) &h->sao_tile_d, (size_t) nctu));
  HIPCHK(hipMemcpyAsync(h->sao_tile_d, tile.data(), (size_t) nctu, hipMemcpyHostToDevice, stream));
  HIPCHK(hipMemsetAsync(h->sao_stat_d, 0, n64 * sizeof(long long), stream));
  VxSaoStatParams p; memset(&p, 0, sizeof p);
  p.frames = h->frames_d; p.tile_of_ctu = h->sao_tile_d; p.out = h->sao_stat_d;
  p.pic_w = h->cfg.pic_w; p.pic_h = h->cfg.pic_h; p.ctus_w = cw; p.ctus_h = chh; p.bit_depth = h->cfg.bit_depth; p.chroma = h->cfg.chroma; p.lf_across_tiles = lf_across_tiles != 0;
  HIPCHK(hipEventRecord(h->ev0, stream));
  const dim3 grid((unsigned) nctu, 3u, (unsigned) h->n_frames);
  if (h->cfg.bit_depth == 8) hipLaunchKernelGGL(vvcx_sao_stats_kernel_u8, grid, dim3(256), 0, stream, p);
  else hipLaunchKernelGGL(vvcx_sao_stats_kernel_u16, grid, dim3(256), 0, stream, p);
  HIPCHK(hipGetLastError());
  HIPCHK(hipEventRecord(h->ev1, stream));
  HIPCHK(hipMemcpyAsync(stats, h->sao_stat_d, n64 * sizeof(long long), hipMemcpyDeviceToHost, stream));
  HIPCHK(hipStreamSynchronize(stream));
  HIPCHK(hipEventElapsedTime(&h->last_sao_stats_ms, h->ev0, h->ev1));
  return VVCX_OK;
}
extern "C" float vvcx_last_sao_stats_ms(const vvcx_handle *h) { return h ? h->last_sao_stats_ms : 0.f; }

// ---- the RD half of the SAO parameter decision (≙ EncSampleAdaptiveOffset::decideBlkParams, EL/EncSampleAdaptiveOffset.cpp:793-1098 without SAOGreedyEnc): host code, a few
// hundred operations per CTU, serial over the CTUs of a picture because the two SAO context models (and the merge candidates) travel from one CTU to the next.
namespace {
struct SaoSet { int mode = 0, type = 0, band = 0; int off[32] = { 0 }; };      // one component of one CTU: off / new / merge; offsets by class (coded or reconstructed)
struct SaoCtu { SaoSet c[3]; };
// the bit estimator of the SAO syntax (sao_block_pars / sao_offset_pars, EL/CABACWriter.cpp:354-462): models SaoMergeFlag and SaoTypeIdx, everything else bypass
class SaoRate {
public:
  void init(int qp)
  {
    for (int k = 0; k < 2; k++) {
      const int id = VX_CTX_INIT_I_REF[kFirst + k], slope = (id >> 3) - 4, offset = ((id & 7) * 18) + 1;
      int st = ((slope * ((qp < 0 ? 0 : qp > 63 ? 63 : qp) - 16)) >> 1) + offset;
      st = st < 1 ? 1 : st > 127 ? 127 : st;
      s0_[k] = (uint16_t) ((st << 8) & 0x7FE0); s1_[k] = (uint16_t) ((st << 8) & 0x7FFE);
    }
    bits = 0;
  }
  uint64_t bits = 0;
  void block(const SaoCtu &u, int bd, bool leftAvail, bool aboveAvail, bool mergeFlagsOnly)
  {
    bool left = false, above = false;
    if (leftAvail) { left = u.c[0].mode == 2 && u.c[0].type == 0; coded(0, left); }
    if (aboveAvail && !left) { above = u.c[0].mode == 2 && u.c[0].type == 1; coded(0, above); }
    if (mergeFlagsOnly || left || above) return;
    for (int k = 0; k < 3; k++) component(u.c[k], k, bd);
  }
  void component(const SaoSet &p, int comp, int bd)
  {
    const bool first = comp < 2;                       // Y and Cb carry the type of their channel
    if (first) { coded(1, p.mode != 0); if (p.mode != 0) bypass(1); }
    if (p.mode != 1) return;
    const int mx = max_q(bd);
    int o[4];
    for (int i = 0; i < 4; i++) o[i] = p.off[p.type == 4 ? (p.band + i) & 31 : (i < 2 ? i : i + 1)];
    for (int i = 0; i < 4; i++) { const int a = std::abs(o[i]); if (mx) bypass(a < mx ? a + 1 : mx); }      // truncated unary
    if (p.type == 4) { for (int i = 0; i < 4; i++) if (o[i]) bypass(1); bypass(5); }
    else if (first) bypass(2);
  }
  static int max_q(int bd) { return (1 << (std::min(bd, 10) - 5)) - 1; }      // SampleAdaptiveOffset::getMaxOffsetQVal
private:
  static const int kFirst = 287;                        // Ctx::SaoMergeFlag in the reference's flat order (ContextSetCfg); SaoTypeIdx follows
  uint16_t s0_[2], s1_[2];
  void bypass(int n) { bits += (uint64_t) n << 15; }
  void coded(int which, bool bin)
  {
    const unsigned st = (unsigned) (s0_[which] + s1_[which]) >> 8;
    bits += VX_BIN_FRAC_BITS[st * 2 + (bin ? 1 : 0)];
    const int rate = VX_CTX_RATE_REF[kFirst + which], r0 = 2 + ((rate >> 2) & 3), r1 = 3 + r0 + (rate & 3);
    s0_[which] -= (s0_[which] >> r0) & 0x7FE0; s1_[which] -= (s1_[which] >> r1) & 0x7FFE;
    if (bin) { s0_[which] += (0x7fffu >> r0) & 0x7FE0; s1_[which] += (0x7fffu >> r1) & 0x7FFE; }
  }
};
// the statistics of one (CTU, component, type)
struct SaoStat { const int64_t *count, *diff; };
class SaoDecision {
public:
  SaoDecision(const int64_t *stats, int bd, const double *lambda, int step) : st_(stats), bd_(bd), lam_(lambda), step_(step) {}
  SaoStat stat(int ctu, int comp, int type) const { const int64_t *b = st_ + (((size_t) ctu * 3 + comp) * 5 + type) * 64; return SaoStat{ b, b + 32 }; }
  static int64_t gain(int64_t n, int64_t o, int64_t d) { return n * o * o - d * o * 2; }      // estSaoDist: SSE change of adding o to n samples with difference sum d
  int64_t distortion(const SaoSet &p, const int *scaled, const SaoStat &s) const
  {
    int64_t d = 0;
    if (p.type == 4) for (int i = p.band; i < p.band + 4; i++) d += gain(s.count[i & 31], scaled[i & 31], s.diff[i & 31]);
    else for (int k = 0; k < 5; k++) d += gain(s.count[k], scaled[k], s.diff[k]);
    return d;
  }
  void scale(const SaoSet &p, int *dst) const
  {
    std::fill(dst, dst + 32, 0);
    if (p.type == 4) for (int i = 0; i < 4; i++) dst[(p.band + i) & 31] = p.off[(p.band + i) & 31] * (1 << step_);
    else for (int k = 0; k < 5; k++) dst[k] = p.off[k] * (1 << step_);
  }
  // deriveOffsets 481-595: the mean difference of a class, rounded and clipped, then lowered towards zero while distortion + lambda * bits improves (estIterOffset)
  SaoSet offsets(int comp, int type, const SaoStat &s) const
  {
    SaoSet p; p.mode = 1; p.type = type;
    const int th = SaoRate::max_q(bd_), n = type == 4 ? 32 : 5;
    const double lambda = lam_[comp];
    double cost[32];
    for (int k = 0; k < n; k++) {
      cost[k] = lambda;
      if ((type != 4 && k == 2) || s.count[k] == 0) continue;
      const double mean = (double) s.diff[k] / (double) (s.count[k] << step_);
      int q = mean >= 0 ? (int) (mean + 0.5) : (int) (mean - 0.5);
      q = std::max(-th, std::min(th, q));
      if (type != 4 && ((k < 2 && q < 0) || (k > 2 && q > 0))) q = 0;            // a valley is never lowered, a peak never raised
      int best = 0; double bestCost = lambda;
      for (int it = q; it != 0; it += it > 0 ? -1 : 1) {
        const int a = std::abs(it);
        const int64_t bitsOf = (type == 4 ? a + 2 : a + 1) - (a == th ? 1 : 0);
        const double c = (double) gain(s.count[k], (int64_t) it << step_, s.diff[k]) + lambda * (double) bitsOf;
        if (c < bestCost) { bestCost = c; best = it; }
      }
      p.off[k] = best; cost[k] = bestCost;
    }
    if (type == 4) {                                    // the four consecutive bands with the smallest cost
      double least = std::numeric_limits<double>::max();
      for (int b = 0; b <= 28; b++) { const double c = cost[b] + cost[b + 1] + cost[b + 2] + cost[b + 3]; if (c < least) { least = c; p.band = b; } }
      for (int k = 0; k < 32; k++) if (((k - p.band) & 31) >= 4) p.off[k] = 0;
    }
    return p;
  }
  // deriveModeNewRDO 597-735
  double explicit_mode(int ctu, SaoRate &rate, bool leftAvail, bool aboveAvail, SaoCtu &out) const
  {
    int64_t dist[3] = { 0, 0, 0 };
    int scaled[32];
    const SaoRate atStart = rate;
    out = SaoCtu();
    rate.block(out, bd_, leftAvail, aboveAvail, true);
    const SaoRate beforeLuma = rate;
    SaoRate afterLuma;
    {
      rate.bits = 0; rate.component(out.c[0], 0, bd_);
      double least = lam_[0] * ((double) rate.bits / 32768.0);
      afterLuma = rate;
      for (int t = 0; t < 5; t++) {
        const SaoStat s = stat(ctu, 0, t);
        const SaoSet cand = offsets(0, t, s);
        scale(cand, scaled);
        const int64_t d = distortion(cand, scaled, s);
        rate = beforeLuma; rate.bits = 0; rate.component(cand, 0, bd_);
        const double c = (double) d + lam_[0] * ((double) rate.bits / 32768.0);
        if (c < least) { least = c; dist[0] = d; out.c[0] = cand; afterLuma = rate; }
      }
      rate = afterLuma;
    }
    {
      double c = 0; uint64_t seen = 0;
      rate.bits = 0;
      for (int k = 1; k < 3; k++) { rate.component(out.c[k], k, bd_); c += lam_[k] * (1.0 / 32768.0) * (double) (rate.bits - seen); seen = rate.bits; }
      double least = c;
      for (int t = 0; t < 5; t++) {
        SaoSet cand[3]; int64_t d[3] = { 0, 0, 0 };
        rate = afterLuma; rate.bits = 0; seen = 0; c = 0;
        for (int k = 1; k < 3; k++) {
          const SaoStat s = stat(ctu, k, t);
          cand[k] = offsets(k, t, s);
          scale(cand[k], scaled);
          d[k] = distortion(cand[k], scaled, s);
          rate.component(cand[k], k, bd_);
          c += (double) d[k] + lam_[k] * (1.0 / 32768.0) * (double) (rate.bits - seen); seen = rate.bits;
        }
        if (c < least) { least = c; for (int k = 1; k < 3; k++) { dist[k] = d[k]; out.c[k] = cand[k]; } }
      }
    }
    double norm = 0;
    for (int k = 0; k < 3; k++) norm += (double) dist[k] / lam_[k];
    rate = atStart; rate.bits = 0;
    rate.block(out, bd_, leftAvail, aboveAvail, false);
    return norm + (double) rate.bits / 32768.0;
  }
  // deriveModeMergeRDO 737-791: cand = the reconstructed parameters of the CTU to the left / above (nullptr: none)
  double merge_mode(int ctu, SaoRate &rate, const SaoCtu *const cand[2], SaoCtu &out) const
  {
    const SaoRate atStart = rate; SaoRate best = rate;
    double least = std::numeric_limits<double>::max();
    for (int m = 0; m < 2; m++) {
      if (!cand[m]) continue;
      SaoCtu test = *cand[m]; double nd = 0;
      for (int k = 0; k < 3; k++) {
        const SaoSet &from = cand[m]->c[k];
        test.c[k].mode = 2; test.c[k].type = m;
        if (from.mode != 0) nd += (double) distortion(from, from.off, stat(ctu, k, from.type)) / lam_[k];
      }
      rate = atStart; rate.bits = 0;
      rate.block(test, bd_, cand[0] != nullptr, cand[1] != nullptr, false);
      const double c = nd + (double) rate.bits / 32768.0;
      if (c < least) { least = c; out = test; best = rate; }
    }
    if (least < std::numeric_limits<double>::max()) rate = best;
    return least;
  }
private:
  const int64_t *st_; int bd_; const double *lam_; int step_;
};
}      // namespace

extern "C" int vvcx_sao_decide(int pic_w, int pic_h, int bit_depth, int tile_cols, int tile_rows, int slice_qp, const double *lambda, int log2_offset_scale,
                               const int64_t *stats, vvcx_sao_param *prm)
{
  if (!lambda || !stats || !prm) return fail(VVCX_ERR_ARG, "null argument");
  if (pic_w < 8 || pic_h < 8 || bit_depth < 8 || bit_depth > 12 || tile_cols < 1 || tile_rows < 1 || tile_cols > (pic_w + 127) / 128 || tile_rows > (pic_h + 127) / 128 ||
      log2_offset_scale < 0 || log2_offset_scale > 4 || !(lambda[0] > 0) || !(lambda[1] > 0) || !(lambda[2] > 0))
    return fail(VVCX_ERR_ARG, "vvcx_sao_decide: picture size / bit depth / tiles / offset scale / lambdas");
  const int cw = (pic_w + 127) / 128, chh = (pic_h + 127) / 128, nctu = cw * chh;
  auto tile_of = [](int c, int n, int tiles) { int t = 0; for (int i = 0; i < tiles; i++) if (c >= (i * n) / tiles) t = i; return t; };
  std::vector<SaoCtu> recon((size_t) nctu);
  SaoDecision rdo(stats, bit_depth, lambda, log2_offset_scale);
  SaoRate rate; rate.init(slice_qp);
  for (int a = 0; a < nctu; a++) {
    const int cx = a % cw, cy = a / cw;
    const SaoCtu *cand[2] = { nullptr, nullptr };
    if (cx > 0 && tile_of(cx - 1, cw, tile_cols) == tile_of(cx, cw, tile_cols)) cand[0] = &recon[(size_t) a - 1];
    if (cy > 0 && tile_of(cy - 1, chh, tile_rows) == tile_of(cy, chh, tile_rows)) cand[1] = &recon[(size_t) a - cw];
    const SaoRate atStart = rate;
    SaoCtu chosen, other;
    double least = rdo.explicit_mode(a, rate, cand[0] != nullptr, cand[1] != nullptr, chosen);
    SaoRate after = rate;
    rate = atStart;
    const double mc = rdo.merge_mode(a, rate, cand, other);
    if (mc < least) { least = mc; chosen = other; after = rate; }
    rate = after;
    for (int k = 0; k < 3; k++) {
      const SaoSet &cd = chosen.c[k]; SaoSet &rc = recon[(size_t) a].c[k];
      if (cd.mode == 2) rc = cand[cd.type]->c[k];
      else { rc = cd; if (cd.mode == 1) rdo.scale(cd, rc.off); }
      vvcx_sao_param &o = prm[(size_t) a * 3 + k];
      memset(&o, 0, sizeof o);
      o.mode = (int8_t) cd.mode; o.type = (int8_t) cd.type;
      if (cd.mode == 1) {
        o.band = (int8_t) (cd.type == 4 ? cd.band : 0);
        for (int i = 0; i < 4; i++) o.offset[i] = (int8_t) (cd.type == 4 ? cd.off[(cd.band + i) & 31] : cd.off[i < 2 ? i : i + 1]);
      }
    }
  }
  return VVCX_OK;
}

// ---- adaptive loop filter (≙ AdaptiveLoopFilter::ALFProcess, CL/AdaptiveLoopFilter.cpp:205-383) with the caller's parameter sets: the per-class tables of every frame's
// slice are built here (≙ reconstructCoeffAPSs 385-418 / reconstructCoeff 420-608, JVET_O0669 form), the kernels (vvcx_alf.hip) classify and filter.
static int alf_clip_value(int chroma, int bit_depth, int idx)      // create() 633-654
{
  if (!chroma) return (int) std::round(std::pow(2., (double) (bit_depth * (4 - idx)) / 4));
  if (idx == 0) return 1 << bit_depth;
  return (int) std::round(std::pow(2., bit_depth - 8 + 8. * (4 - idx - 1) / 3));
}
static int alf_build(const vvcx_alf_aps *aps, int n_aps, const vvcx_alf_slice *slices, const vvcx_alf_ctu *ctus, int n_frames, int nctu, int bit_depth, int chroma, std::vector<VxAlfFrame> &tabs)
{
  if (n_aps < 0 || n_aps > 8) return fail(VVCX_ERR_ARG, "ALF: %d parameter sets (0..8)", n_aps);
  for (int i = 0; i < n_aps; i++) {
    const vvcx_alf_aps &a = aps[i];
    if (a.num_luma_filters < 1 || a.num_luma_filters > 25 || a.num_chroma_alt < 0 || a.num_chroma_alt > 8) return fail(VVCX_ERR_ARG, "ALF parameter set %d: %d luma filters / %d chroma alternatives", i, a.num_luma_filters, a.num_chroma_alt);
    for (int c = 0; c < 25; c++) if (a.class_to_filter[c] >= a.num_luma_filters) return fail(VVCX_ERR_ARG, "ALF parameter set %d: class %d uses filter %d of %d", i, c, a.class_to_filter[c], a.num_luma_filters);
    for (int f = 0; f < a.num_luma_filters; f++) for (int k = 0; k < 12; k++) if (a.luma_clip_idx[f][k] > 3) return fail(VVCX_ERR_ARG, "ALF parameter set %d: clipping index above 3", i);
    for (int t = 0; t < a.num_chroma_alt; t++) for (int k = 0; k < 6; k++) if (a.chroma_clip_idx[t][k] > 3) return fail(VVCX_ERR_ARG, "ALF parameter set %d: clipping index above 3", i);
  }
  tabs.assign((size_t) n_frames, VxAlfFrame());
  for (int f = 0; f < n_frames; f++) {
    const vvcx_alf_slice &sl = slices[f];
    VxAlfFrame &t = tabs[(size_t) f];
    memset(&t, 0, sizeof t);
    if (sl.n_luma_aps < 0 || sl.n_luma_aps > 8 || sl.chroma_aps >= n_aps) return fail(VVCX_ERR_ARG, "ALF slice of frame %d: %d luma sets / chroma set %d of %d", f, sl.n_luma_aps, sl.chroma_aps, n_aps);
    t.n_sets = sl.n_luma_aps;
    for (int k = 0; k < sl.n_luma_aps; k++) {
      if (sl.luma_aps[k] < 0 || sl.luma_aps[k] >= n_aps) return fail(VVCX_ERR_ARG, "ALF slice of frame %d: luma set %d of %d", f, sl.luma_aps[k], n_aps);
      const vvcx_alf_aps &a = aps[sl.luma_aps[k]];
      for (int c = 0; c < 25; c++) for (int i = 0; i < 12; i++) {
        const int fl = a.class_to_filter[c];
        t.luma_coeff[k][c][i] = a.luma_coeff[fl][i];
        t.luma_clip[k][c][i] = (int16_t) alf_clip_value(0, bit_depth, a.nonlinear_luma ? a.luma_clip_idx[fl][i] : 0);
      }
    }
    const bool chromaOn = chroma && sl.chroma_aps >= 0;
    if (chromaOn) {
      const vvcx_alf_aps &a = aps[sl.chroma_aps];
      t.n_alt = a.num_chroma_alt;
      for (int alt = 0; alt < a.num_chroma_alt; alt++) for (int i = 0; i < 6; i++) {
        t.chroma_coeff[alt][i] = a.chroma_coeff[alt][i];
        t.chroma_clip[alt][i] = (int16_t) alf_clip_value(1, bit_depth, a.nonlinear_chroma[alt] ? a.chroma_clip_idx[alt][i] : 0);
      }
    }
    for (int a = 0; a < nctu; a++) {
      const vvcx_alf_ctu &u = ctus[(size_t) f * nctu + a];
      if (u.flag[0] && (u.set < 0 || u.set >= 16 + t.n_sets)) return fail(VVCX_ERR_ARG, "ALF: frame %d CTU %d uses luma filter set %d of %d", f, a, u.set, 16 + t.n_sets);
      for (int c = 1; c < 3; c++) if (chromaOn && u.flag[c] && u.alt[c - 1] >= t.n_alt) return fail(VVCX_ERR_ARG, "ALF: frame %d CTU %d uses chroma alternative %d of %d", f, a, u.alt[c - 1], t.n_alt);
    }
  }
  return VVCX_OK;
}
// the per-CTU choices as the kernels read them: chroma flags cleared where the slice has no chroma set (≙ slice-level alf_chroma_idc)
static void alf_ctus(const vvcx_alf_slice *slices, const vvcx_alf_ctu *ctus, int n_frames, int nctu, int chroma, std::vector<VxAlfCtu> &out)
{
  out.resize((size_t) n_frames * nctu);
  for (int f = 0; f < n_frames; f++) for (int a = 0; a < nctu; a++) {
    const vvcx_alf_ctu &u = ctus[(size_t) f * nctu + a]; VxAlfCtu &o = out[(size_t) f * nctu + a];
    const bool con = chroma && slices[f].chroma_aps >= 0;
    o.flag[0] = u.flag[0] != 0; o.flag[1] = con && u.flag[1]; o.flag[2] = con && u.flag[2]; o.set = u.set; o.alt[0] = u.alt[0]; o.alt[1] = u.alt[1];
  }
}
static void alf_launch(VxAlfParams &p, int n_frames, size_t bps, hipStream_t stream)
{
  const dim3 gridC((unsigned) ((p.pic_w + 1023) / 1024), (unsigned) ((p.pic_h + 3) / 4), (unsigned) (3 * n_frames));      // the copy: strips of 4 rows x 1024 samples
  const dim3 gridF((unsigned) ((p.pic_w + 63) / 64), (unsigned) ((p.pic_h + 15) / 16), (unsigned) (3 * n_frames));
  if (bps == 1) { hipLaunchKernelGGL(vvcx_alf_copy_kernel_u8, gridC, dim3(256), 0, stream, p); hipLaunchKernelGGL(vvcx_alf_kernel_u8, gridF, dim3(256), 0, stream, p); }
  else { hipLaunchKernelGGL(vvcx_alf_copy_kernel_u16, gridC, dim3(256), 0, stream, p); hipLaunchKernelGGL(vvcx_alf_kernel_u16, gridF, dim3(256), 0, stream, p); }
}
extern "C" int vvcx_alf_bound_frames(vvcx_handle *h, const vvcx_alf_aps *aps, int n_aps, const vvcx_alf_slice *slices, const vvcx_alf_ctu *ctus, void *hip_stream)
{
  NOT_PENDING(h);
  if (!h || !slices || !ctus || (n_aps > 0 && !aps)) return fail(VVCX_ERR_ARG, "null argument");
  if (!h->n_frames || !h->have_slice) return fail(VVCX_ERR_STATE, "no bound frames / slice");
  for (size_t i = 0; i < h->next_idx.size(); i++)
    if (h->next_idx[i] != (int) h->sub_ctus[i % (size_t) h->nsub].size()) return fail(VVCX_ERR_STATE, "the loop filters need every CTU of the bound pictures coded");
  if (h->lmcs_on && !h->lmcs_inverted) return fail(VVCX_ERR_STATE, "LMCS slice: vvcx_lmcs_inverse_reco first (the loop filters work in the original domain)");
  const int cw = h->ctus_w, chh = h->ctus_h, nctu = cw * chh;
  std::vector<VxAlfFrame> tabs; std::vector<VxAlfCtu> cts;
  const int rr = alf_build(aps, n_aps, slices, ctus, h->n_frames, nctu, h->cfg.bit_depth, h->cfg.chroma, tabs);
  if (rr != VVCX_OK) return rr;
  alf_ctus(slices, ctus, h->n_frames, nctu, h->cfg.chroma, cts);
  DevGuard guard(h->cfg.device);
  hipStream_t stream = (hipStream_t) hip_stream;
  const size_t bps = h->cfg.bit_depth == 8 ? 1 : 2, ny = (size_t) h->cfg.pic_w * h->cfg.pic_h, per_frame = ny + 2 * (ny >> 2);
  if (h->sao_tmp_cap < (size_t) h->n_frames * per_frame * bps) { (void) hipFree(h->sao_tmp_d); h->sao_tmp_d = nullptr; h->sao_tmp_cap = (size_t) h->n_frames * per_frame * bps; HIPCHK(hipMalloc(&h->sao_tmp_d, h->sao_tmp_cap)); }
  if (h->alf_cap < (size_t) h->n_frames) {
    (void) hipFree(h->alf_tab_d); (void) hipFree(h->alf_ctu_d); h->alf_tab_d = nullptr; h->alf_ctu_d = nullptr; h->alf_cap = (size_t) h->n_frames;
    HIPCHK(hipMalloc((void **) &h->alf_tab_d, h->alf_cap * sizeof(VxAlfFrame))); HIPCHK(hipMalloc((void **) &h->alf_ctu_d, h->alf_cap * nctu * sizeof(VxAlfCtu)));
  }
  HIPCHK(hipMemcpyAsync(h->alf_tab_d, tabs.data(), tabs.size() * sizeof(VxAlfFrame), hipMemcpyHostToDevice, stream));
  HIPCHK(hipMemcpyAsync(h->alf_ctu_d, cts.data(), cts.size() * sizeof(VxAlfCtu), hipMemcpyHostToDevice, stream));
  VxAlfParams p; memset(&p, 0, sizeof p);
  p.frames = h->frames_d; p.tabs = h->alf_tab_d; p.ctus = h->alf_ctu_d; p.tmp = h->sao_tmp_d; p.tmp_frame = per_frame; p.tmp_comp[0] = 0; p.tmp_comp[1] = ny; p.tmp_comp[2] = ny + (ny >> 2);
  p.classes = nullptr; p.pic_w = h->cfg.pic_w; p.pic_h = h->cfg.pic_h; p.ctus_w = cw; p.ctus_h = chh; p.bit_depth = h->cfg.bit_depth; p.chroma = h->cfg.chroma;
  HIPCHK(hipEventRecord(h->ev0, stream));
  alf_launch(p, h->n_frames, bps, stream);
  HIPCHK(hipGetLastError());
  HIPCHK(hipEventRecord(h->ev1, stream));
  HIPCHK(hipStreamSynchronize(stream));             // (the host tables above must outlive the copies)
  HIPCHK(hipEventElapsedTime(&h->last_alf_ms, h->ev0, h->ev1));
  return VVCX_OK;
}
// the same kernels on one picture in host memory (uint16 planes, stride = plane width; filtered in place): the leaf entry the filter is pinned through
extern "C" int vvcx_alf_picture(int pic_w, int pic_h, int bit_depth, const vvcx_alf_aps *aps, int n_aps, const vvcx_alf_slice *slice, const vvcx_alf_ctu *ctus,
                                uint16_t *y, uint16_t *cb, uint16_t *cr, uint8_t *classes, int device)
{
  if (!slice || !ctus || !y || !cb || !cr || (n_aps > 0 && !aps)) return fail(VVCX_ERR_ARG, "null argument");
  if (pic_w < 8 || pic_h < 8 || (pic_w & 7) || (pic_h & 7) || pic_w > 16384 || pic_h > 16384 || bit_depth < 8 || bit_depth > 12) return fail(VVCX_ERR_ARG, "vvcx_alf_picture: picture size / bit depth");
  const int cw = (pic_w + 127) / 128, chh = (pic_h + 127) / 128, nctu = cw * chh;
  std::vector<VxAlfFrame> tabs; std::vector<VxAlfCtu> cts;
  const int rr = alf_build(aps, n_aps, slice, ctus, 1, nctu, bit_depth, 1, tabs);
  if (rr != VVCX_OK) return rr;
  alf_ctus(slice, ctus, 1, nctu, 1, cts);
  DevGuard guard(device);
  const size_t ny = (size_t) pic_w * pic_h, nc = ny >> 2, per_frame = ny + 2 * nc, ncls = (size_t) (pic_w >> 2) * (pic_h >> 2);
  uint16_t *pl_d = nullptr, *tmp_d = nullptr; VxFrameDev *fd_d = nullptr; VxAlfFrame *tab_d = nullptr; VxAlfCtu *ctu_d = nullptr; uint8_t *cls_d = nullptr;
  int rc = VVCX_OK;
  if (hipMalloc((void **) &pl_d, per_frame * 2) != hipSuccess || hipMalloc((void **) &tmp_d, per_frame * 2) != hipSuccess || hipMalloc((void **) &fd_d, sizeof(VxFrameDev)) != hipSuccess ||
      hipMalloc((void **) &tab_d, sizeof(VxAlfFrame)) != hipSuccess || hipMalloc((void **) &ctu_d, cts.size() * sizeof(VxAlfCtu)) != hipSuccess || hipMalloc((void **) &cls_d, ncls) != hipSuccess)
    rc = fail(VVCX_ERR_DEVICE, "hipMalloc failed for the ALF of a %dx%d picture", pic_w, pic_h);
  if (rc == VVCX_OK) {
    VxFrameDev fd; memset(&fd, 0, sizeof fd);
    fd.rec[0] = pl_d; fd.rec[1] = pl_d + ny; fd.rec[2] = pl_d + ny + nc; fd.stride[0] = pic_w; fd.stride[1] = fd.stride[2] = pic_w >> 1;
    VxAlfParams p; memset(&p, 0, sizeof p);
    p.frames = fd_d; p.tabs = tab_d; p.ctus = ctu_d; p.tmp = tmp_d; p.tmp_frame = per_frame; p.tmp_comp[0] = 0; p.tmp_comp[1] = ny; p.tmp_comp[2] = ny + nc;
    p.classes = cls_d; p.pic_w = pic_w; p.pic_h = pic_h; p.ctus_w = cw; p.ctus_h = chh; p.bit_depth = bit_depth; p.chroma = 1;
    bool ok = hipMemcpy(pl_d, y, ny * 2, hipMemcpyHostToDevice) == hipSuccess && hipMemcpy(pl_d + ny, cb, nc * 2, hipMemcpyHostToDevice) == hipSuccess &&
              hipMemcpy(pl_d + ny + nc, cr, nc * 2, hipMemcpyHostToDevice) == hipSuccess && hipMemcpy(fd_d, &fd, sizeof fd, hipMemcpyHostToDevice) == hipSuccess &&
              hipMemcpy(tab_d, tabs.data(), sizeof(VxAlfFrame), hipMemcpyHostToDevice) == hipSuccess && hipMemcpy(ctu_d, cts.data(), cts.size() * sizeof(VxAlfCtu), hipMemcpyHostToDevice) == hipSuccess &&
              hipMemset(cls_d, 255, ncls) == hipSuccess;
    if (ok) {
      alf_launch(p, 1, 2, 0);
      ok = hipGetLastError() == hipSuccess && hipDeviceSynchronize() == hipSuccess && hipMemcpy(y, pl_d, ny * 2, hipMemcpyDeviceToHost) == hipSuccess &&
           hipMemcpy(cb, pl_d + ny, nc * 2, hipMemcpyDeviceToHost) == hipSuccess && hipMemcpy(cr, pl_d + ny + nc, nc * 2, hipMemcpyDeviceToHost) == hipSuccess &&
           (!classes || hipMemcpy(classes, cls_d, ncls, hipMemcpyDeviceToHost) == hipSuccess);
    }
    if (!ok) rc = fail(VVCX_ERR_DEVICE, "ALF of a picture: a HIP call failed");
  }
  (void) hipFree(pl_d); (void) hipFree(tmp_d); (void) hipFree(fd_d); (void) hipFree(tab_d); (void) hipFree(ctu_d); (void) hipFree(cls_d);
  return rc;
}
extern "C" float vvcx_last_alf_ms(const vvcx_handle *h) { return h ? h->last_alf_ms : 0.f; }

// the same two kernels on a picture the caller describes by a CU table (host memory in, host memory out): the unit maps the kernels read are built here from the rows
extern "C" int vvcx_deblock_cu_table(int pic_w, int pic_h, int bit_depth, int qp, int qp_cb, int qp_cr, int beta_offset_div2, int tc_offset_div2,
                                     const int32_t *rows, int n_rows, uint16_t *y, uint16_t *cb, uint16_t *cr, int device)
{
  if (!rows || !y || !cb || !cr || n_rows <= 0) return fail(VVCX_ERR_ARG, "null argument");
  if (pic_w < 8 || pic_h < 8 || (pic_w & 7) || (pic_h & 7) || pic_w > 16384 || pic_h > 16384) return fail(VVCX_ERR_ARG, "picture size must be a multiple of 8 (MinCUSize of the cfg) up to 16384");
  if (bit_depth < 8 || bit_depth > 12 || qp < 0 || qp > 63 || beta_offset_div2 < -6 || beta_offset_div2 > 6 || tc_offset_div2 < -6 || tc_offset_div2 > 6) return fail(VVCX_ERR_ARG, "bit depth / QP / offsets out of range");
  const int uw = pic_w >> 2, uh = pic_h >> 2;
  std::vector<VxUnit> um((size_t) 2 * uw * uh);
  memset(um.data(), 0, um.size() * sizeof(VxUnit));
  for (int i = 0; i < n_rows; i++) {
    const int32_t *r = rows + 6 * i;
    const int ch = r[0], sh = ch ? 1 : 0;
    const bool pow2 = r[3] > 0 && r[4] > 0 && !(r[3] & (r[3] - 1)) && !(r[4] & (r[4] - 1));
    if (ch < 0 || ch > 1 || !pow2 || r[3] < 4 || r[4] < 4 || r[3] > 128 || r[4] > 128 || (r[1] & 3) || (r[2] & 3) || r[1] < 0 || r[2] < 0 || r[1] + r[3] > pic_w || r[2] + r[4] > pic_h ||
        r[5] < 0 || r[5] > 2 || (r[5] && (ch || r[3] * r[4] <= 16 || r[3] > 64 || r[4] > 64)))
      return fail(VVCX_ERR_ARG, "CU row %d is not a CU of a %dx%d picture", i, pic_w, pic_h);
    int lw = 0, lh = 0;
    while ((1 << lw) < (r[3] >> sh)) lw++;
    while ((1 << lh) < (r[4] >> sh)) lh++;
    for (int v = r[2] >> 2; v < (r[2] + r[4]) >> 2; v++) for (int u = r[1] >> 2; u < (r[1] + r[3]) >> 2; u++) {
      VxUnit &t = um[(size_t) ch * uw * uh + (size_t) v * uw + u];
      t.tag = 1; t.x = (int16_t) (r[1] >> sh); t.y = (int16_t) (r[2] >> sh); t.lw = (uint8_t) lw; t.lh = (uint8_t) lh; t.mts = (uint8_t) (r[5] << 6);
    }
  }
  for (size_t i = 0; i < um.size(); i++) if (!um[i].tag) return fail(VVCX_ERR_ARG, "the CU table does not cover the picture (%s tree, 4x4 unit %d)", i < um.size() / 2 ? "luma" : "chroma", (int) (i % (um.size() / 2)));
  DevGuard guard(device);
  const size_t ny = (size_t) pic_w * pic_h, nc = ny >> 2;
  VxUnit *um_d = nullptr; uint16_t *pl_d = nullptr; VxFrameDev *fd_d = nullptr; uint8_t *ed_d = nullptr;
  int rc = VVCX_OK;
  if (hipMalloc((void **) &ed_d, (size_t) 4 * uw * uh) != hipSuccess || hipMalloc((void **) &um_d, um.size() * sizeof(VxUnit)) != hipSuccess || hipMalloc((void **) &pl_d, (ny + 2 * nc) * 2) != hipSuccess || hipMalloc((void **) &fd_d, sizeof(VxFrameDev)) != hipSuccess)
    rc = fail(VVCX_ERR_DEVICE, "hipMalloc failed for the deblocking of a %dx%d CU table", pic_w, pic_h);
  if (rc == VVCX_OK) {
    VxFrameDev fd; memset(&fd, 0, sizeof fd);
    fd.rec[0] = pl_d; fd.rec[1] = pl_d + ny; fd.rec[2] = pl_d + ny + nc;
    fd.stride[0] = pic_w; fd.stride[1] = fd.stride[2] = pic_w >> 1;
    fd.units[0] = um_d; fd.units[1] = um_d + (size_t) uw * uh;
    VxDeblockParams p; memset(&p, 0, sizeof p);
    p.frames = fd_d; p.uw = uw; p.uh = uh; p.bit_depth = bit_depth; p.chroma = 1; p.qp = qp; p.qp_c[0] = qp_cb; p.qp_c[1] = qp_cr; p.beta_off2 = beta_offset_div2; p.tc_off2 = tc_offset_div2; p.edges = ed_d;
    bool ok = hipMemcpy(um_d, um.data(), um.size() * sizeof(VxUnit), hipMemcpyHostToDevice) == hipSuccess && hipMemcpy(fd_d, &fd, sizeof fd, hipMemcpyHostToDevice) == hipSuccess &&
              hipMemcpy(pl_d, y, ny * 2, hipMemcpyHostToDevice) == hipSuccess && hipMemcpy(pl_d + ny, cb, nc * 2, hipMemcpyHostToDevice) == hipSuccess &&
              hipMemcpy(pl_d + ny + nc, cr, nc * 2, hipMemcpyHostToDevice) == hipSuccess;
    if (ok) { hipLaunchKernelGGL(vvcx_deblock_edges_kernel, dim3((unsigned) ((2 * uw * uh + 255) / 256), 1), dim3(256), 0, 0, p); ok = hipGetLastError() == hipSuccess; }
    for (int dir = 0; dir < 2 && ok; dir++) {
      p.dir = dir;
      hipLaunchKernelGGL(vvcx_deblock_kernel_u16, dim3((unsigned) ((2 * uw * uh + 255) / 256), 1), dim3(256), 0, 0, p);
      ok = hipGetLastError() == hipSuccess;
    }
    ok = ok && hipDeviceSynchronize() == hipSuccess && hipMemcpy(y, pl_d, ny * 2, hipMemcpyDeviceToHost) == hipSuccess && hipMemcpy(cb, pl_d + ny, nc * 2, hipMemcpyDeviceToHost) == hipSuccess &&
         hipMemcpy(cr, pl_d + ny + nc, nc * 2, hipMemcpyDeviceToHost) == hipSuccess;
    if (!ok) rc = fail(VVCX_ERR_DEVICE, "deblocking of a CU table: a HIP call failed");
  }
  if (ed_d) (void) hipFree(ed_d);
  if (um_d) (void) hipFree(um_d);
  if (pl_d) (void) hipFree(pl_d);
  if (fd_d) (void) hipFree(fd_d);
  return rc;
}

// quantised levels of one component of a coded picture at their sample positions (≙ tu.getCoeffs(compID) of the final TUs), host plane
extern "C" int vvcx_get_levels(vvcx_handle *h, int frame, int comp, int16_t *plane, int stride)
{
  NOT_PENDING(h);
  if (!h || !plane || frame < 0 || frame >= h->n_frames || comp < 0 || comp > 2) return fail(VVCX_ERR_ARG, "bad argument");
  const int w = comp ? h->cfg.pic_w >> 1 : h->cfg.pic_w, hh = comp ? h->cfg.pic_h >> 1 : h->cfg.pic_h;
  if (stride < w) return fail(VVCX_ERR_ARG, "stride smaller than the plane width");
  HIPCHK(hipSetDevice(h->cfg.device));
  const int16_t *src = h->lev_d + (size_t) frame * h->lev_frame + (comp == 0 ? 0 : comp == 1 ? h->lev_plane[0] : h->lev_plane[0] + h->lev_plane[1]);
  std::vector<int16_t> tmp((size_t) w * hh);
  HIPCHK(hipMemcpy(tmp.data(), src, tmp.size() * 2, hipMemcpyDeviceToHost));
  for (int y = 0; y < hh; y++) memcpy(plane + (size_t) y * stride, tmp.data() + (size_t) y * w, (size_t) w * 2);
  return VVCX_OK;
}

extern "C" int vvcx_get_cus(vvcx_handle *h, int frame, vvcx_cu *cus, int max_cus, int *n_cus)
{
  NOT_PENDING(h);
  if (!h || !n_cus || frame < 0 || frame >= h->n_frames) return fail(VVCX_ERR_ARG, "bad argument");
  DevGuard guard(h->cfg.device);
  std::vector<VxUnit> um(h->units_frame);
  HIPCHK(hipMemcpy(um.data(), h->units_d + (size_t) frame * h->units_frame, h->units_frame * sizeof(VxUnit), hipMemcpyDeviceToHost));
  int n = 0;
  for (int ry = 0; ry < h->ctus_h; ry++) for (int rx = 0; rx < h->ctus_w; rx++)
    for (int ch = 0; ch < (h->cfg.chroma ? 2 : 1); ch++) {
      const int ul = ch ? 1 : 2;
      for (int uy = ry * 32; uy < std::min(ry * 32 + 32, h->uh); uy++) for (int ux = rx * 32; ux < std::min(rx * 32 + 32, h->uw); ux++) {
        const VxUnit &u = um[(size_t) ch * h->units_plane + (size_t) uy * h->uw + ux];
        if (!u.tag || (u.x >> ul) != ux || (u.y >> ul) != uy) continue;
        if (cus && n < max_cus) {
          vvcx_cu &o = cus[n];
          o.x = u.x; o.y = u.y; o.w = (int16_t) (1 << u.lw); o.h = (int16_t) (1 << u.lh); o.ch_type = (uint8_t) ch;
          o.qt_depth = u.qt; o.bt_depth = u.bt; o.mt_depth = u.mt; o.depth = u.depth; o.intra_dir = u.dir; o.mrl_idx = u.mrl & 0x7f; o.mip_flag = u.mrl >> 7; o.cbf = u.cbf & 7; o.mts_idx = ch ? 0 : (u.mts & 7); o.joint_cb_cr = ch ? (u.mts & 7) : 0; o.lfnst_idx = (u.mts >> 4) & 3; o.split_series = u.ss;
          o.isp_mode = ch ? 0 : u.mts >> 6; o.tu_cbf = ch ? 0 : u.cbf >> 4;
        }
        n++;
      }
    }
  *n_cus = n;
  return (cus && n > max_cus) ? fail(VVCX_ERR_ARG, "CU table too small (%d > %d)", n, max_cus) : VVCX_OK;
}

// The transform units of the coded picture (cs.tus as the reference's encodeCtus / the bitstream writer walk them, CL/CodingStructure.h:216-241): in this
// configuration every CU carries exactly one TU (MaxTbSize 64 = the largest intra CU of a dual-tree I slice), in the order of vvcx_get_cus.
extern "C" int vvcx_get_tus(vvcx_handle *h, int frame, vvcx_tu *tus, int max_tus, int *n_tus)
{
  NOT_PENDING(h);
  if (!h || !n_tus || frame < 0 || frame >= h->n_frames) return fail(VVCX_ERR_ARG, "bad argument");
  int ncu = 0;
  const int rc = vvcx_get_cus(h, frame, nullptr, 0, &ncu);
  if (rc != VVCX_OK) return rc;
  std::vector<vvcx_cu> cus((size_t) ncu);
  const int rc2 = vvcx_get_cus(h, frame, cus.data(), ncu, &ncu);
  if (rc2 != VVCX_OK) return rc2;
  // sub-partitions of a CU coded with ISP (CU::getISPSplitDim, CL/UnitTools.cpp:437-459): a quarter of the side along the split, at least 16 samples each
  auto isp_parts = [](const vvcx_cu &c, int &tw, int &th) {
    if (!c.isp_mode) { tw = c.w; th = c.h; return 1; }
    const int hor = c.isp_mode == 1, split = hor ? c.h : c.w, non = hor ? c.w : c.h;
    int factor = 1; if (non < 16) { int l = 0; while ((1 << (l + 1)) <= non) l++; factor = 16 >> l; }
    const int psz = (split >> 2) < factor ? factor : (split >> 2);
    tw = hor ? c.w : psz; th = hor ? psz : c.h;
    return split / psz;
  };
  int n = 0;
  for (int i = 0; i < ncu; i++) { int tw, th; n += isp_parts(cus[(size_t) i], tw, th); }
  *n_tus = n;
  if (!tus) return VVCX_OK;
  if (n > max_tus) return fail(VVCX_ERR_ARG, "TU table too small (%d > %d)", n, max_tus);
  const int wl = h->cfg.pic_w, wc = h->cfg.pic_w >> 1;
  int o = 0;
  for (int i = 0; i < ncu; i++) {
    const vvcx_cu &c = cus[(size_t) i];
    int tw, th; const int parts = isp_parts(c, tw, th);
    for (int k = 0; k < parts; k++) {
      vvcx_tu &t = tus[o++];
      memset(&t, 0, sizeof t);
      t.cu_index = i; t.ch_type = c.ch_type; t.w = (int16_t) tw; t.h = (int16_t) th; t.depth = c.isp_mode ? 1 : 0;
      t.x = (int16_t) (c.x + (c.isp_mode == 2 ? k * tw : 0)); t.y = (int16_t) (c.y + (c.isp_mode == 1 ? k * th : 0));
      t.mts_idx = c.mts_idx; t.joint_cb_cr = c.joint_cb_cr;
      if (!c.ch_type) { t.cbf[0] = c.isp_mode ? (c.tu_cbf >> k) & 1 : c.cbf & 1; t.coeff_offset[0] = (int32_t) t.y * wl + t.x; t.coeff_stride[0] = wl; t.coeff_offset[1] = t.coeff_offset[2] = -1; }
      else {
        t.cbf[1] = (c.cbf >> 1) & 1; t.cbf[2] = (c.cbf >> 2) & 1;
        t.coeff_offset[0] = -1; t.coeff_offset[1] = t.coeff_offset[2] = (int32_t) c.y * wc + c.x; t.coeff_stride[1] = t.coeff_stride[2] = wc;
      }
    }
  }
  return VVCX_OK;
}

extern "C" float vvcx_last_kernel_ms(const vvcx_handle *h) { return h ? h->last_ms : 0.f; }

extern "C" int vvcx_get_counters(vvcx_handle *h, uint64_t out[4])
{
  NOT_PENDING(h);
  if (!h || !out) return fail(VVCX_ERR_ARG, "null argument");
  DevGuard guard(h->cfg.device);
  unsigned long long c[4];
  HIPCHK(hipMemcpy(c, h->counters_d, sizeof c, hipMemcpyDeviceToHost));
  for (int i = 0; i < 4; i++) out[i] = c[i];
  return VVCX_OK;
}

// diagnostic: shader-clock ticks summed over streams per controller/operation kind of the last launch
// [0] controller, [op] parallel operation `op` (enum in vvcx_kernel.hip), [12] estimator pass
extern "C" int vvcx_get_profile(vvcx_handle *h, uint64_t out[48])
{
  NOT_PENDING(h);
  if (!h || !out) return fail(VVCX_ERR_ARG, "null argument");
  DevGuard guard(h->cfg.device);
  unsigned long long c[52];
  HIPCHK(hipMemcpy(c, h->counters_d, sizeof c, hipMemcpyDeviceToHost));
  for (int i = 0; i < 48; i++) out[i] = c[4 + i];
  return VVCX_OK;
}


// ------------------------------------------------------------------------------------------------ leaf operators
namespace {
struct DevBuf {                         // device allocation released on scope exit
  void *p = nullptr;
  ~DevBuf() { if (p) (void) hipFree(p); }
  hipError_t alloc(size_t n) { return hipMalloc(&p, n ? n : 1); }
  template <typename T> T *as() { return (T *) p; }
};
bool pow2_block(int w, int h) { return w >= 2 && h >= 2 && w <= 64 && h <= 64 && !(w & (w - 1)) && !(h & (h - 1)); }
}

extern "C" int vvcx_distortion_batch(const int16_t *a, const int16_t *b, int w, int h, int n, uint64_t *out, int device)
{
  if (!a || !b || !out || n < 0 || !pow2_block(w, h)) return fail(VVCX_ERR_ARG, "bad argument");
  if (n == 0) return VVCX_OK;
  HIPCHK(hipSetDevice(device));
  const size_t bytes = (size_t) n * w * h * 2;
  DevBuf da, db, ds, dout;
  HIPCHK(da.alloc(bytes)); HIPCHK(db.alloc(bytes)); HIPCHK(ds.alloc(bytes)); HIPCHK(dout.alloc((size_t) n * 3 * 8));
  HIPCHK(hipMemcpy(da.p, a, bytes, hipMemcpyHostToDevice)); HIPCHK(hipMemcpy(db.p, b, bytes, hipMemcpyHostToDevice));
  hipLaunchKernelGGL(vvcx_leaf_dist_kernel, dim3((unsigned) n), dim3(VXD_NT), 0, 0, da.as<int16_t>(), db.as<int16_t>(), w, h, ds.as<int16_t>(), dout.as<unsigned long long>());
  HIPCHK(hipGetLastError());
  HIPCHK(hipMemcpy(out, dout.p, (size_t) n * 3 * 8, hipMemcpyDeviceToHost));
  return VVCX_OK;
}

extern "C" int vvcx_intra_pred_batch(vvcx_handle *h, const void *const reco[3], const uint8_t *const coded[2], const vvcx_pred_case *cases, int n, int16_t *pred)
{
  NOT_PENDING(h);
  if (!h || !reco || !coded || !cases || !pred || n < 0) return fail(VVCX_ERR_ARG, "bad argument");
  if (n == 0) return VVCX_OK;
  HIPCHK(hipSetDevice(h->cfg.device));
  const int bps = h->cfg.bit_depth == 8 ? 1 : 2, W = h->cfg.pic_w, H = h->cfg.pic_h;
  std::vector<int> off((size_t) n); size_t total = 0;
  for (int i = 0; i < n; i++) {
    const vvcx_pred_case &c = cases[i];
    const int cw = c.comp ? W >> 1 : W, chh = c.comp ? H >> 1 : H;
    if (c.comp < 0 || c.comp > 2 || !pow2_block(c.w, c.h) || c.x < 0 || c.y < 0 || c.x + c.w > cw || c.y + c.h > chh || c.mode < 0 || c.mode > 66 ||
        (c.mrl != 0 && (c.comp != 0 || (c.mrl != 1 && c.mrl != 3)))) return fail(VVCX_ERR_ARG, "bad prediction case %d", i);
    off[(size_t) i] = (int) total; total += (size_t) c.w * c.h;
  }
  DevBuf dplane[3], dunits, dframe, dcases, doff, dpred;
  VxFrameDev fd; memset(&fd, 0, sizeof fd);
  for (int c = 0; c < 3; c++) {
    const size_t pw = c ? W >> 1 : W, ph = c ? H >> 1 : H;
    if (!reco[c]) return fail(VVCX_ERR_ARG, "plane %d is null", c);
    HIPCHK(dplane[c].alloc(pw * ph * bps)); HIPCHK(hipMemcpy(dplane[c].p, reco[c], pw * ph * bps, hipMemcpyHostToDevice));
    fd.org[c] = dplane[c].p; fd.rec[c] = dplane[c].p; fd.stride[c] = (int32_t) pw;
  }
  const size_t nu = (size_t) h->uw * h->uh;
  std::vector<VxUnit> units(2 * nu); memset(units.data(), 0, units.size() * sizeof(VxUnit));
  for (int t = 0; t < 2; t++) { if (!coded[t]) return fail(VVCX_ERR_ARG, "coded map %d is null", t); for (size_t i = 0; i < nu; i++) units[(size_t) t * nu + i].tag = coded[t][i] ? 1 : 0; }
  HIPCHK(dunits.alloc(units.size() * sizeof(VxUnit))); HIPCHK(hipMemcpy(dunits.p, units.data(), units.size() * sizeof(VxUnit), hipMemcpyHostToDevice));
  fd.units[0] = dunits.as<VxUnit>(); fd.units[1] = dunits.as<VxUnit>() + nu;
  HIPCHK(dframe.alloc(sizeof fd)); HIPCHK(hipMemcpy(dframe.p, &fd, sizeof fd, hipMemcpyHostToDevice));
  HIPCHK(dcases.alloc((size_t) n * sizeof(VxLeafPred))); HIPCHK(hipMemcpy(dcases.p, cases, (size_t) n * sizeof(VxLeafPred), hipMemcpyHostToDevice));
  HIPCHK(doff.alloc((size_t) n * 4)); HIPCHK(hipMemcpy(doff.p, off.data(), (size_t) n * 4, hipMemcpyHostToDevice));
  HIPCHK(dpred.alloc(total * 2));
  VxParams p; memset(&p, 0, sizeof p);
  p.pic_w = W; p.pic_h = H; p.bit_depth = h->cfg.bit_depth; p.tools = h->cfg.tools; p.uw = h->uw; p.uh = h->uh; p.frames = dframe.as<VxFrameDev>();
  if (bps == 1) hipLaunchKernelGGL(vvcx_leaf_pred_kernel_u8, dim3((unsigned) n), dim3(VXD_NT), 0, 0, p, dcases.as<VxLeafPred>(), dpred.as<int16_t>(), doff.as<int>());
  else hipLaunchKernelGGL(vvcx_leaf_pred_kernel_u16, dim3((unsigned) n), dim3(VXD_NT), 0, 0, p, dcases.as<VxLeafPred>(), dpred.as<int16_t>(), doff.as<int>());
  HIPCHK(hipGetLastError());
  HIPCHK(hipMemcpy(pred, dpred.p, total * 2, hipMemcpyDeviceToHost));
  return VVCX_OK;
}

// the leaf operators exchange whole context arrays in the reference's flat order (ContextSetCfg, 386 models); the library keeps the models an intra slice touches
// (VXD_NUM_CTX, vvcx_tables.h VX_CTX_REF_INDEX)
extern "C" int vvcx_ctx_init(int qp, uint16_t *s0, uint16_t *s1)
{
  if (!s0 || !s1) return fail(VVCX_ERR_ARG, "null argument");
  qp = qp < 0 ? 0 : qp > 63 ? 63 : qp;
  for (int k = 0; k < VX_NUM_CTX_REF; k++) {               // CtxStore::init for an I slice, every model of the reference (ctx_init_islice: the kept ones)
    const int id = VX_CTX_INIT_I_REF[k];
    const int slope = (id >> 3) - 4, offset = ((id & 7) * 18) + 1;
    int st = ((slope * (qp - 16)) >> 1) + offset;
    st = st < 1 ? 1 : st > 127 ? 127 : st;
    const int p1 = st << 8;
    s0[k] = (uint16_t) (p1 & 0x7FE0); s1[k] = (uint16_t) (p1 & 0x7FFE);
  }
  return VVCX_OK;
}

extern "C" int vvcx_cabac_code_bins(uint16_t *s0, uint16_t *s1, int ctx, const uint8_t *bins, int nbins, uint64_t *frac_bits, int device)
{
  if (!s0 || !s1 || !bins || !frac_bits || nbins < 0 || ctx < 0 || ctx >= VX_NUM_CTX_REF) return fail(VVCX_ERR_ARG, "bad argument");      // ctx: the reference's flat index; only its adaptation rate matters
  HIPCHK(hipSetDevice(device));
  DevBuf dio, dbins, dbits;
  uint16_t io[2] = { *s0, *s1 };
  HIPCHK(dio.alloc(4)); HIPCHK(dbins.alloc((size_t) nbins)); HIPCHK(dbits.alloc(8));
  HIPCHK(hipMemcpy(dio.p, io, 4, hipMemcpyHostToDevice)); HIPCHK(hipMemcpy(dbins.p, bins, (size_t) nbins, hipMemcpyHostToDevice));
  hipLaunchKernelGGL(vvcx_leaf_cabac_kernel, dim3(1), dim3(VXD_NT), 0, 0, dio.as<uint16_t>(), (int) VX_CTX_RATE_REF[ctx], dbins.as<uint8_t>(), nbins, dbits.as<unsigned long long>());
  HIPCHK(hipGetLastError());
  HIPCHK(hipMemcpy(io, dio.p, 4, hipMemcpyDeviceToHost)); HIPCHK(hipMemcpy(frac_bits, dbits.p, 8, hipMemcpyDeviceToHost));
  *s0 = io[0]; *s1 = io[1];
  return VVCX_OK;
}

extern "C" int vvcx_rd_cost_batch(double lambda, const uint64_t *frac_bits, const uint64_t *dist, int n, double *cost, int device)
{
  if (!(lambda > 0.0) || !frac_bits || !dist || !cost || n < 0) return fail(VVCX_ERR_ARG, "bad argument");
  if (n == 0) return VVCX_OK;
  HIPCHK(hipSetDevice(device));
  DevBuf db, dd, dc;
  HIPCHK(db.alloc((size_t) n * 8)); HIPCHK(dd.alloc((size_t) n * 8)); HIPCHK(dc.alloc((size_t) n * 8));
  HIPCHK(hipMemcpy(db.p, frac_bits, (size_t) n * 8, hipMemcpyHostToDevice)); HIPCHK(hipMemcpy(dd.p, dist, (size_t) n * 8, hipMemcpyHostToDevice));
  VxParams p; memset(&p, 0, sizeof p);
  p.lambda = lambda; p.dist_scale = (double) (1 << 15) / lambda;
  hipLaunchKernelGGL(vvcx_leaf_rdcost_kernel, dim3((unsigned) ((n + VXD_NT - 1) / VXD_NT)), dim3(VXD_NT), 0, 0, p, db.as<unsigned long long>(), dd.as<unsigned long long>(), n, dc.as<double>());
  HIPCHK(hipGetLastError());
  HIPCHK(hipMemcpy(cost, dc.p, (size_t) n * 8, hipMemcpyDeviceToHost));
  return VVCX_OK;
}

extern "C" int vvcx_scan_order(int w, int h, uint16_t *idx, int device)
{
  if (!idx || !pow2_block(w, h)) return fail(VVCX_ERR_ARG, "bad argument");
  HIPCHK(hipSetDevice(device));
  const int n = (w < 32 ? w : 32) * (h < 32 ? h : 32);
  DevBuf d; HIPCHK(d.alloc((size_t) n * 2));
  hipLaunchKernelGGL(vvcx_leaf_scan_kernel, dim3(1), dim3(VXD_NT), 0, 0, w, h, d.as<uint16_t>());
  HIPCHK(hipGetLastError());
  HIPCHK(hipMemcpy(idx, d.p, (size_t) n * 2, hipMemcpyDeviceToHost));
  return VVCX_OK;
}

// ≙ IntraPrediction::initIntraMip + predIntraMip (CL/IntraPrediction.cpp:2152-2205) for n blocks: case i = {w, h, mode, bit_depth}, its
// unfiltered line-0 reference samples top[w] | left[h] at refs + ref_off[i], prediction (w*h, stride w) to pred + pred_off[i]; host pointers
extern "C" int vvcx_mip_pred_batch(const int32_t *cases, int n, const int16_t *refs, int n_refs, int16_t *pred, int n_pred, int device)
{
  if (!cases || !refs || !pred || n < 0) return fail(VVCX_ERR_ARG, "bad argument");
  std::vector<VxMipCase> cs((size_t) n);
  int ro = 0, po = 0;
  for (int i = 0; i < n; i++) {
    const int w = cases[4 * i], h = cases[4 * i + 1], mode = cases[4 * i + 2], bd = cases[4 * i + 3];
    const bool shape = pow2_block(w, h) && w <= 64 && h <= 64 && w <= 4 * h && h <= 4 * w;
    const int nm = !shape ? 0 : (w == 4 && h == 4) ? 35 : (w <= 8 && h <= 8) ? 19 : 11;
    if (!shape || mode < 0 || mode >= nm || (bd != 8 && bd != 10)) return fail(VVCX_ERR_ARG, "MIP case %d: block %dx%d mode %d bit depth %d", i, w, h, mode, bd);
    cs[(size_t) i] = { w, h, mode, bd, ro, po };
    ro += w + h; po += w * h;
  }
  if (ro > n_refs || po > n_pred) return fail(VVCX_ERR_ARG, "reference / prediction buffers too small");
  if (n == 0) return VVCX_OK;
  HIPCHK(hipSetDevice(device));
  DevBuf dc, dr, dp; HIPCHK(dc.alloc(sizeof(VxMipCase) * (size_t) n)); HIPCHK(dr.alloc((size_t) ro * 2)); HIPCHK(dp.alloc((size_t) po * 2));
  HIPCHK(hipMemcpy(dc.p, cs.data(), sizeof(VxMipCase) * (size_t) n, hipMemcpyHostToDevice));
  HIPCHK(hipMemcpy(dr.p, refs, (size_t) ro * 2, hipMemcpyHostToDevice));
  hipLaunchKernelGGL(vvcx_leaf_mip_kernel, dim3((unsigned) n), dim3(64), 0, 0, dc.as<VxMipCase>(), dr.as<int16_t>(), dp.as<int16_t>());
  HIPCHK(hipGetLastError());
  HIPCHK(hipMemcpy(pred, dp.p, (size_t) po * 2, hipMemcpyDeviceToHost));
  return VVCX_OK;
}

// ≙ GetPartition(C0..C25, 2) of BIN/TEST.py for n feature rows (26 int32 each, host pointers): the forest set with vvcx_set_forest
extern "C" int vvcx_forest_predict_batch(vvcx_handle *h, const int32_t *rows, int n, int32_t *out)
{
  NOT_PENDING(h);
  if (!h || !rows || !out || n < 0) return fail(VVCX_ERR_ARG, "bad argument");
  if (!h->f_ntrees) return fail(VVCX_ERR_STATE, "vvcx_set_forest first");
  if (n == 0) return VVCX_OK;
  HIPCHK(hipSetDevice(h->cfg.device));
  DevBuf dr, dout; HIPCHK(dr.alloc((size_t) n * 26 * 4)); HIPCHK(dout.alloc((size_t) n * 4));
  HIPCHK(hipMemcpy(dr.p, rows, (size_t) n * 26 * 4, hipMemcpyHostToDevice));
  VxParams p; memset(&p, 0, sizeof p);
  p.f_node = h->f_node_d; p.f_value = h->f_value_d; p.f_root = h->f_root_d; p.f_ntrees = h->f_ntrees; p.f_nclasses = h->f_nclasses;
  for (int c = 0; c < 8; c++) p.f_classes[c] = h->f_classes[c];
  hipLaunchKernelGGL(vvcx_leaf_forest_kernel, dim3((unsigned) ((n + VXD_NT - 1) / VXD_NT)), dim3(VXD_NT), 0, 0, p, dr.as<int32_t>(), n, dout.as<int32_t>());
  HIPCHK(hipGetLastError());
  HIPCHK(hipMemcpy(out, dout.p, (size_t) n * 4, hipMemcpyDeviceToHost));
  return VVCX_OK;
}

extern "C" int vvcx_transform_quant_batch(const int16_t *org, const int16_t *pred, int w, int h, int bit_depth, int qp, int n,
                                          int16_t *lev, int16_t *rec, uint64_t *sse, uint8_t *cbf, int device)
{
  if (!org || !pred || !lev || !rec || !sse || !cbf || n < 0 || !pow2_block(w, h) || (bit_depth != 8 && bit_depth != 10) || qp < 0 || qp > 75) return fail(VVCX_ERR_ARG, "bad argument");
  if (n == 0) return VVCX_OK;
  HIPCHK(hipSetDevice(device));
  const size_t bytes = (size_t) n * w * h * 2;
  DevBuf dorg, drec, dlev, dtmp, dout;
  HIPCHK(dorg.alloc(bytes)); HIPCHK(drec.alloc(bytes)); HIPCHK(dlev.alloc(bytes)); HIPCHK(dtmp.alloc((size_t) n * 2048 * 4)); HIPCHK(dout.alloc((size_t) n * 16));
  HIPCHK(hipMemcpy(dorg.p, org, bytes, hipMemcpyHostToDevice)); HIPCHK(hipMemcpy(drec.p, pred, bytes, hipMemcpyHostToDevice));
  hipLaunchKernelGGL(vvcx_leaf_trq_kernel, dim3((unsigned) n), dim3(VXD_NT), 0, 0, dorg.as<int16_t>(), drec.as<int16_t>(), dlev.as<int16_t>(), dtmp.as<int32_t>(), w, h, bit_depth, qp,
                     dout.as<unsigned long long>());
  HIPCHK(hipGetLastError());
  std::vector<unsigned long long> o((size_t) n * 2);
  HIPCHK(hipMemcpy(lev, dlev.p, bytes, hipMemcpyDeviceToHost)); HIPCHK(hipMemcpy(rec, drec.p, bytes, hipMemcpyDeviceToHost));
  HIPCHK(hipMemcpy(o.data(), dout.p, (size_t) n * 16, hipMemcpyDeviceToHost));
  for (int i = 0; i < n; i++) { sse[i] = o[(size_t) i * 2]; cbf[i] = (uint8_t) o[(size_t) i * 2 + 1]; }
  return VVCX_OK;
}

static int depquant_batch_impl(const int16_t *org, const int16_t *pred, int w, int h, int bit_depth, int qp, int comp, int mts_idx, int cbf_cb, double lambda,
                               const uint16_t *s0, const uint16_t *s1, int n, int16_t *lev, int16_t *rec, uint64_t *sse, uint8_t *cbf, int device, int lfnst_idx, int intra_dir)
{
  if (!org || !pred || !lev || !rec || !sse || !cbf || !s0 || !s1 || n < 0 || !pow2_block(w, h) || (bit_depth != 8 && bit_depth != 10) || qp < 0 || qp > 75 || comp < 0 || comp > 2 ||
      !(lambda > 0.0) || (mts_idx != 0 && (mts_idx < 2 || mts_idx > 5 || comp != 0 || w > 32 || h > 32 || w < 4 || h < 4))) return fail(VVCX_ERR_ARG, "bad argument");
  if (n == 0) return VVCX_OK;
  HIPCHK(hipSetDevice(device));
  const size_t bytes = (size_t) n * w * h * 2;
  DevBuf dorg, drec, dlev, dtmp, dout, dctx, dtab, dscr;
  HIPCHK(dorg.alloc(bytes)); HIPCHK(drec.alloc(bytes)); HIPCHK(dlev.alloc(bytes)); HIPCHK(dtmp.alloc((size_t) n * 2048 * 4)); HIPCHK(dout.alloc((size_t) n * 16));
  HIPCHK(dctx.alloc(2 * VXD_NUM_CTX * 2)); HIPCHK(dtab.alloc(48 * sizeof(VxDqConst))); HIPCHK(dscr.alloc((size_t) n * VXD_OFF_CACHE));
  HIPCHK(hipMemcpy(dorg.p, org, bytes, hipMemcpyHostToDevice)); HIPCHK(hipMemcpy(drec.p, pred, bytes, hipMemcpyHostToDevice));
  { uint16_t kept[2 * VXD_NUM_CTX]; ctx_from_reference_order(s0, s1, kept); HIPCHK(hipMemcpy(dctx.p, kept, sizeof kept, hipMemcpyHostToDevice)); }
  VxDqConst tab[48]; memset(tab, 0, sizeof tab);
  for (int lsum = 2; lsum <= 12; lsum++) tab[comp * 16 + lsum] = dq_consts_of(lsum, bit_depth, qp, lambda);
  HIPCHK(hipMemcpy(dtab.p, tab, sizeof tab, hipMemcpyHostToDevice));
  VxParams p; memset(&p, 0, sizeof p);
  p.bit_depth = bit_depth; p.tools = VVCX_TOOL_DEPQUANT | VVCX_TOOL_MTS | (lfnst_idx ? VVCX_TOOL_LFNST : 0); p.dq_consts = dtab.as<VxDqConst>(); p.scratch = dscr.as<uint8_t>(); p.scratch_per_stream = VXD_OFF_CACHE;
  hipLaunchKernelGGL(vvcx_leaf_dq_kernel, dim3((unsigned) n), dim3(VXD_NT), 0, 0, p, dctx.as<uint16_t>(), dorg.as<int16_t>(), drec.as<int16_t>(), dlev.as<int16_t>(), dtmp.as<int32_t>(),
                     w, h, qp, comp, mts_idx, cbf_cb, dout.as<unsigned long long>(), (w >= 4 && h >= 4) ? lfnst_idx : 0, intra_dir);
  HIPCHK(hipGetLastError());
  std::vector<unsigned long long> o((size_t) n * 2);
  HIPCHK(hipMemcpy(lev, dlev.p, bytes, hipMemcpyDeviceToHost)); HIPCHK(hipMemcpy(rec, drec.p, bytes, hipMemcpyDeviceToHost));
  HIPCHK(hipMemcpy(o.data(), dout.p, (size_t) n * 16, hipMemcpyDeviceToHost));
  for (int i = 0; i < n; i++) { sse[i] = o[(size_t) i * 2]; cbf[i] = (uint8_t) o[(size_t) i * 2 + 1]; }
  return VVCX_OK;
}

extern "C" int vvcx_depquant_batch(const int16_t *org, const int16_t *pred, int w, int h, int bit_depth, int qp, int comp, int mts_idx, int cbf_cb, double lambda,
                                   const uint16_t *s0, const uint16_t *s1, int n, int16_t *lev, int16_t *rec, uint64_t *sse, uint8_t *cbf, int device)
{
  return depquant_batch_impl(org, pred, w, h, bit_depth, qp, comp, mts_idx, cbf_cb, lambda, s0, s1, n, lev, rec, sse, cbf, device, 0, 0);
}
extern "C" int vvcx_lfnst_depquant_batch(const int16_t *org, const int16_t *pred, int w, int h, int bit_depth, int qp, int comp, int lfnst_idx, int intra_dir, int cbf_cb, double lambda,
                                         const uint16_t *s0, const uint16_t *s1, int n, int16_t *lev, int16_t *rec, uint64_t *sse, uint8_t *cbf, int device)
{
  if (lfnst_idx < 0 || lfnst_idx > 2 || intra_dir < 0 || intra_dir > 66) return fail(VVCX_ERR_ARG, "bad argument");
  return depquant_batch_impl(org, pred, w, h, bit_depth, qp, comp, 0, cbf_cb, lambda, s0, s1, n, lev, rec, sse, cbf, device, lfnst_idx, intra_dir);
}

// ≙ TrQuant::transformNxN / DepQuant::quant / invTransformNxN of one luma TU (tw x th: 1 x N, 2 x N, N x 1, N x 2 or larger) of a CU with cu.ispMode set: implicit DST-VII /
// DCT-II (getTrTypes 752-780), the one-stage transforms of one-sample-wide blocks, the cbf context of ISP sub-partitions (QtCbf[Y] + 2 + prev_cbf; inferred: no cbf rate)
extern "C" int vvcx_isp_tu_batch(const int16_t *org, const int16_t *pred, int tw, int th, int bit_depth, int qp, double lambda, int prev_cbf, int cbf_inferred,
                                 const uint16_t *s0, const uint16_t *s1, int n, int16_t *lev, int16_t *rec, uint64_t *sse, uint8_t *cbf, int device)
{
  const bool shape = tw >= 1 && th >= 1 && tw <= 64 && th <= 64 && !(tw & (tw - 1)) && !(th & (th - 1)) && tw * th >= 16;
  if (!org || !pred || !lev || !rec || !sse || !cbf || !s0 || !s1 || n < 0 || !shape || (bit_depth != 8 && bit_depth != 10) || qp < 0 || qp > 75 || !(lambda > 0.0)) return fail(VVCX_ERR_ARG, "bad argument");
  if (n == 0) return VVCX_OK;
  HIPCHK(hipSetDevice(device));
  const size_t bytes = (size_t) n * tw * th * 2;
  DevBuf dorg, drec, dlev, dtmp, dout, dctx, dtab, dscr;
  HIPCHK(dorg.alloc(bytes)); HIPCHK(drec.alloc(bytes)); HIPCHK(dlev.alloc(bytes)); HIPCHK(dtmp.alloc((size_t) n * 2048 * 4)); HIPCHK(dout.alloc((size_t) n * 16));
  HIPCHK(dctx.alloc(2 * VXD_NUM_CTX * 2)); HIPCHK(dtab.alloc(96 * sizeof(VxDqConst))); HIPCHK(dscr.alloc((size_t) n * VXD_OFF_CACHE));
  HIPCHK(hipMemcpy(dorg.p, org, bytes, hipMemcpyHostToDevice)); HIPCHK(hipMemcpy(drec.p, pred, bytes, hipMemcpyHostToDevice));
  { uint16_t kept[2 * VXD_NUM_CTX]; ctx_from_reference_order(s0, s1, kept); HIPCHK(hipMemcpy(dctx.p, kept, sizeof kept, hipMemcpyHostToDevice)); }
  VxDqConst tab[96]; memset(tab, 0, sizeof tab);
  for (int lsum = 2; lsum <= 12; lsum++) tab[lsum] = dq_consts_of(lsum, bit_depth, qp, lambda);
  HIPCHK(hipMemcpy(dtab.p, tab, sizeof tab, hipMemcpyHostToDevice));
  VxParams p; memset(&p, 0, sizeof p);
  p.bit_depth = bit_depth; p.tools = VVCX_TOOL_DEPQUANT | VVCX_TOOL_MTS | VVCX_TOOL_LFNST | VVCX_TOOL_ISP; p.dq_consts = dtab.as<VxDqConst>(); p.scratch = dscr.as<uint8_t>(); p.scratch_per_stream = VXD_OFF_CACHE;
  p.lambda = lambda;
  hipLaunchKernelGGL(vvcx_leaf_isp_kernel, dim3((unsigned) n), dim3(VXD_NT), 0, 0, p, dctx.as<uint16_t>(), dorg.as<int16_t>(), drec.as<int16_t>(), dlev.as<int16_t>(), dtmp.as<int32_t>(),
                     tw, th, qp, cbf_inferred ? -1 : (int) VX_CTX_QtCbf[0] + 2 + (prev_cbf ? 1 : 0), dout.as<unsigned long long>());
  HIPCHK(hipGetLastError());
  std::vector<unsigned long long> o((size_t) n * 2);
  HIPCHK(hipMemcpy(lev, dlev.p, bytes, hipMemcpyDeviceToHost)); HIPCHK(hipMemcpy(rec, drec.p, bytes, hipMemcpyDeviceToHost));
  HIPCHK(hipMemcpy(o.data(), dout.p, (size_t) n * 16, hipMemcpyDeviceToHost));
  for (int i = 0; i < n; i++) { sse[i] = o[(size_t) i * 2]; cbf[i] = (uint8_t) o[(size_t) i * 2 + 1]; }
  return VVCX_OK;
}

// ≙ TrQuant::transformNxN / invTransformNxN of a luma TU with tu.mtsIdx = MTS_SKIP and the {DCT2, TS} pruning in front of it (CL/TrQuant.cpp:1049-1124, 1394-1440, 996-1041),
// QuantRDOQ::xRateDistOptQuantTS (CL/QuantRDOQ.cpp:1243-1483), CABACWriter::residual_codingTS (EL/CABACWriter.cpp:4306-4555) for n residual blocks (host pointers, stride w):
// levels, reconstructed residual, absSum, whether the pruning keeps the candidate, and the fractional bits of the levels from the given context models
extern "C" int vvcx_transform_skip_batch(const int16_t *resi, int w, int h, int bit_depth, int qp, double lambda, const uint16_t *s0, const uint16_t *s1, int n,
                                         int16_t *lev, int16_t *resi_out, int32_t *abs_sum, uint8_t *keep, uint64_t *frac_bits, int device)
{
  if (!resi || !lev || !resi_out || !abs_sum || !keep || !frac_bits || !s0 || !s1 || n < 0 || !pow2_block(w, h) || w < 4 || h < 4 || w > 32 || h > 32 || (bit_depth != 8 && bit_depth != 10) ||
      qp < 0 || qp > 75 || !(lambda > 0.0)) return fail(VVCX_ERR_ARG, "bad argument");
  if (n == 0) return VVCX_OK;
  HIPCHK(hipSetDevice(device));
  const size_t bytes = (size_t) n * w * h * 2;
  DevBuf dres, dout, dlev, dtmp, do2, dbits, dctx;
  HIPCHK(dres.alloc(bytes)); HIPCHK(dout.alloc(bytes)); HIPCHK(dlev.alloc(bytes)); HIPCHK(dtmp.alloc((size_t) n * 2048 * 4)); HIPCHK(do2.alloc((size_t) n * 8)); HIPCHK(dbits.alloc((size_t) n * 8));
  HIPCHK(dctx.alloc(2 * VXD_NUM_CTX * 2));
  HIPCHK(hipMemcpy(dres.p, resi, bytes, hipMemcpyHostToDevice)); HIPCHK(hipMemset(dout.p, 0, bytes));
  { uint16_t kept[2 * VXD_NUM_CTX]; ctx_from_reference_order(s0, s1, kept); HIPCHK(hipMemcpy(dctx.p, kept, sizeof kept, hipMemcpyHostToDevice)); }
  VxParams p; memset(&p, 0, sizeof p);
  p.bit_depth = bit_depth; p.tools = VVCX_TOOL_DEPQUANT | VVCX_TOOL_LFNST | VVCX_TOOL_TS; p.lambda = lambda;
  hipLaunchKernelGGL(vvcx_leaf_ts_kernel, dim3((unsigned) n), dim3(VXD_NT), 0, 0, p, dctx.as<uint16_t>(), dres.as<int16_t>(), dlev.as<int16_t>(), dout.as<int16_t>(), dtmp.as<int32_t>(), w, h, qp,
                     do2.as<int>(), dbits.as<unsigned long long>());
  HIPCHK(hipGetLastError());
  std::vector<int> o((size_t) n * 2);
  HIPCHK(hipMemcpy(lev, dlev.p, bytes, hipMemcpyDeviceToHost)); HIPCHK(hipMemcpy(resi_out, dout.p, bytes, hipMemcpyDeviceToHost));
  HIPCHK(hipMemcpy(o.data(), do2.p, (size_t) n * 8, hipMemcpyDeviceToHost)); HIPCHK(hipMemcpy(frac_bits, dbits.p, (size_t) n * 8, hipMemcpyDeviceToHost));
  for (int i = 0; i < n; i++) { abs_sum[i] = o[(size_t) i * 2]; keep[i] = (uint8_t) o[(size_t) i * 2 + 1]; }
  return VVCX_OK;
}

// slice_data() payload of one tile of a bound frame (≙ the sub-streams EncSlice::encodeSlice hands to the NAL writer, EL/EncSlice.cpp:1884-2006): one byte string, or with
// VVCX_TOOL_WPP the byte strings of the tile's CTU rows back to back (sizes: vvcx_get_substream_sizes)
static int payload_of_tile(vvcx_handle *h, int frame, int tile, uint8_t *buf, int cap, int *nbytes, int *sizes, int max_sizes, int *n_sizes)
{
  if (!h || frame < 0 || frame >= h->n_frames || tile < 0 || tile >= h->ntiles) return fail(VVCX_ERR_ARG, "bad argument");
  if (!h->payload_d) return fail(VVCX_ERR_STATE, "handle was created without emit_payload");
  DevGuard guard(h->cfg.device);
  const int sub0 = h->tile_sub0[(size_t) tile], ns = h->tile_nsub[(size_t) tile];
  if (n_sizes) *n_sizes = ns;
  if (sizes && max_sizes < ns) return fail(VVCX_ERR_ARG, "room for %d sub-stream sizes, the tile has %d", max_sizes, ns);
  size_t total = 0;
  for (int k = 0; k < ns; k++) {
    const size_t s = (size_t) frame * h->nsub + (size_t) (sub0 + k);
    if (h->next_idx[s] != (int) h->sub_ctus[(size_t) (sub0 + k)].size()) return fail(VVCX_ERR_STATE, "tile %d of frame %d is not completely coded yet", tile, frame);
    uint32_t st[8];
    HIPCHK(hipMemcpy(st, (const uint8_t *) h->arith_d + s * 32, 32, hipMemcpyDeviceToHost));
    const uint32_t n = st[7];                        // Arith::n
    if (n > h->payload_cap[s]) return fail(VVCX_ERR_STATE, "payload of tile %d exceeds the %u bytes reserved", tile, h->payload_cap[s]);
    if (sizes) sizes[k] = (int) n;
    if (buf && total + n <= (size_t) cap) HIPCHK(hipMemcpy(buf + total, h->payload_d + h->payload_off[s], n, hipMemcpyDeviceToHost));
    total += n;
  }
  if (nbytes) *nbytes = (int) total;
  if (buf && total > (size_t) cap) return fail(VVCX_ERR_ARG, "buffer too small: %zu bytes needed", total);
  return VVCX_OK;
}
// GET_TRAINING_SET counterpart (the fork's CL/TypeDef.h:54-56 switch; EL/EncCu.cpp:863-1123 computes the features, the label is the partition the full search chose)
extern "C" int vvcx_enable_training_dump(vvcx_handle *h, int cap_rows)
{
  NOT_PENDING(h);
  if (!h || cap_rows < 0) return fail(VVCX_ERR_ARG, "bad argument");
  // the features look at the CUs above-right and below-left of a node through the unrestricted neighbour lookup; under WPP those belong to rows that run at their own
  // pace, so the rows would depend on timing (the same reason VVCX_TOOL_FAST is refused together with WPP)
  if (cap_rows > 0 && (h->cfg.tools & VVCX_TOOL_WPP)) return fail(VVCX_ERR_UNSUPPORTED, "the training dump is not available on a WPP handle");
  DevGuard guard(h->cfg.device);
  (void) hipFree(h->train_rows_d); (void) hipFree(h->train_n_d); h->train_rows_d = nullptr; h->train_n_d = nullptr; h->train_cap = 0;
  if (cap_rows == 0) return VVCX_OK;
  if (hipMalloc((void **) &h->train_rows_d, (size_t) cap_rows * 28 * 4) != hipSuccess || hipMalloc((void **) &h->train_n_d, 4) != hipSuccess) {
    (void) hipFree(h->train_rows_d); h->train_rows_d = nullptr;
    return fail(VVCX_ERR_DEVICE, "device allocation of %d training rows failed", cap_rows);
  }
  HIPCHK(hipMemset(h->train_n_d, 0, 4));
  h->train_cap = cap_rows;
  return VVCX_OK;
}
extern "C" int vvcx_get_training_rows(vvcx_handle *h, int32_t *rows, int max_rows, int *n_rows)
{
  NOT_PENDING(h);
  if (!h || !n_rows || max_rows < 0 || (!rows && max_rows)) return fail(VVCX_ERR_ARG, "bad argument");
  if (!h->train_rows_d) return fail(VVCX_ERR_STATE, "vvcx_enable_training_dump first");
  DevGuard guard(h->cfg.device);
  uint32_t n = 0;
  HIPCHK(hipMemcpy(&n, h->train_n_d, 4, hipMemcpyDeviceToHost));
  *n_rows = (int) n;                                  // rows asked for; more than the capacity means the tail was dropped
  const int have = (int) (n < (uint32_t) h->train_cap ? n : (uint32_t) h->train_cap), take = have < max_rows ? have : max_rows;
  if (take) HIPCHK(hipMemcpy(rows, h->train_rows_d, (size_t) take * 28 * 4, hipMemcpyDeviceToHost));
  return VVCX_OK;
}
extern "C" int vvcx_get_payload(vvcx_handle *h, int frame, int tile, uint8_t *buf, int cap, int *nbytes)
{
  NOT_PENDING(h);
  if (!buf || !nbytes) return fail(VVCX_ERR_ARG, "bad argument");
  return payload_of_tile(h, frame, tile, buf, cap, nbytes, nullptr, 0, nullptr);
}
extern "C" int vvcx_get_substream_sizes(vvcx_handle *h, int frame, int tile, int *sizes, int max_sizes, int *n_sizes)
{
  NOT_PENDING(h);
  if (!n_sizes) return fail(VVCX_ERR_ARG, "bad argument");
  return payload_of_tile(h, frame, tile, nullptr, 0, nullptr, sizes, max_sizes, n_sizes);
}

// vvcx_kernel.hip — the device side of the hot path: one persistent workgroup (4 wavefronts) per CTU
// stream runs the whole QT/BT/TT partition recursion of its CTUs on the GPU.
//
//   * thread 0 is the sequential "mode controller": an explicit-stack state machine that replays the
//     reference's decision order (EL/EncCu.cpp xCompressCU 727 / xCheckModeSplit 1918 / xCheckRDCostIntra
//     2402, EL/EncModeCtrl.cpp initCULevel 1203 / tryMode 1557 / useModeResult 2089).  It never touches
//     pixels; it posts one parallel operation at a time.
//   * all 256 threads execute the posted operation: LDS staging of the node's original tile and its
//     L-shaped reference samples (CL/IntraPrediction.cpp:1215-1522), one wavefront per candidate for the
//     SATD stage (prediction 426-935 + CL/RdCost.cpp SAD/Hadamard) and for the full-RD stage
//     (residual → DCT-II → quant → estimated bits → dequant → inverse → reco → SSE;
//     EL/IntraSearch.cpp:2694-3168, CL/TrQuant.cpp:835-992, CL/Quant.cpp:423-1089), wave reductions for
//     the distortions, and the node-area copies between the picture planes and the per-level stores.
//
// Results are bit-identical to the CPU oracle by construction of the arithmetic (same integer
// operations, fp64 cost arithmetic compiled with -ffp-contract=off).
#include <hip/hip_runtime.h>
#include <stdint.h>
#define VX_QUAL static __device__ const
#include "vvcx_tables.h"
#include "vvcx_dev.h"
#include "vvcx_mip_dev.h"

#define NT VXD_NT
// Virtual thread index: the workgroup's wavefronts are rotated by a per-stream amount, so that the serial parts of a stream (the mode controller on thread 0,
// the batched trellis on wave 0) do not all land on the same SIMD of a CU when its four resident streams run them (wave w of a workgroup sits on SIMD w).
// Lanes keep their place inside the wavefront; every wave index in this file is the virtual one.
#if (VXD_NW & (VXD_NW - 1)) == 0
#define VTX ((threadIdx.x + ((((unsigned) blockIdx.x * 0x9E3779B1u) >> 30) << 6)) & (unsigned) (VXD_NT - 1))
#else
__device__ inline unsigned vtx_rotated()       // a wave count that is not a power of two: rotate by (hash of the workgroup) mod NW waves
{
  const unsigned t = threadIdx.x + ((((unsigned) blockIdx.x * 0x9E3779B1u) >> 24) % VXD_NW << 6);
  return t >= VXD_NT ? t - VXD_NT : t;
}
#define VTX (vtx_rotated())
#endif
#define NW VXD_NW
#define MAXD VXD_MAXD
#define BUF VXD_BUF
#define NCTX VXD_NUM_CTX
static_assert(VXD_NUM_CTX == VX_NUM_CTX, "vvcx_dev.h and the generated tables disagree on the number of context models");
#define MAX_DOUBLE 1.7e+308
#ifndef VVCX_STAMP
#define VVCX_STAMP 0          // 1: diagnostic build with shader-clock stamps per operation kind
#endif
#if VVCX_STAMP == 1
#define STAMP() clock64()
#elif VVCX_STAMP == 2
#define STAMP() wall_clock64()
#else
#define STAMP() 0ll
#endif

enum { SPLIT_NONE = 0, SPLIT_QT = 1, SPLIT_BH = 2, SPLIT_BV = 3, SPLIT_TH = 4, SPLIT_TV = 5 };
enum { ETM_INTRA, ETM_POST_DONT_SPLIT, ETM_SPLIT_QT, ETM_SPLIT_BT_H, ETM_SPLIT_BT_V, ETM_SPLIT_TT_H, ETM_SPLIT_TT_V, ETM_RECO_CACHED };
#define TOOL_CU_REUSE (1u << 11)
#define TOOL_MIP (1u << 1)
#define TOOL_ISP (1u << 2)
#define TOOL_LFNST (1u << 3)
#define MIPF 0x80                  // a MIP CU: bit 7 of the unit's / candidate's mrl field (MIP forces multiRefIdx 0), the MIP mode in dir / mode
#define TOOL_MTS (1u << 4)
#define TOOL_TS (1u << 5)
#define TOOL_DEPQUANT (1u << 6)
#define TOOL_CCLM (1u << 8)
#define TOOL_JCCR (1u << 9)
#define TOOL_FAST (1u << 12)
#define TOOL_WPP (1u << 13)
enum { LM_CHROMA = 67, MDLM_L = 68, MDLM_T = 69 };
enum { PLANAR = 0, DC = 1, HOR = 18, DIA = 34, VER = 50, VDIA = 66, DM_CHROMA = 70 };
enum { OP_NONE, OP_DONE, OP_LUMA_PREP, OP_STAGE_A, OP_STAGE_B, OP_CHROMA_RD, OP_SAVE_INTRA, OP_SAVE_PIC, OP_RESTORE_PIC,
       OP_CLEAR_UNITS, OP_CTX_COPY, OP_REUSE, OP_FAST, OP_ISP, OP_ISP_END, OP_ISP_PARK };
enum { PH_ENTER, PH_FAST_DONE, PH_RUN, PH_A1_DONE, PH_A2_DONE, PH_B_DONE, PH_INTRA_SAVED, PH_CHILD, PH_CHILD_RET, PH_SPLIT_SAVED, PH_ADVANCE, PH_EXIT, PH_EXIT2, PH_A3_DONE, PH_PASS, PH_NEXT_PASS, PH_ISP };
enum { CTX_CUR = 0, CTX_START = 1, CTX_BEST = 2, CTX_WAVE = 3 };   // OP_CTX_COPY endpoints

struct Ctx { uint16_t s0[NCTX], s1[NCTX]; };
struct Cab { int ci; uint64_t bits; };   // ci: index into L.ctxs (the estimator a syntax function writes to); an index, not a pointer, so that
                                          // every model access is an LDS instruction and not a generic-address one

struct Sum {                       // what the mode controller reads from a CodingStructure
  double cost; uint64_t dist, bits;
  int16_t n_cu, f_bt, f_cbf, f_w, f_h, l_bt, l_w, l_h, max_qt, valid;
};

struct Frame {                     // one recursion level: partitioner level + ComprCUCtx + temp/best summaries
  int16_t x, y, w, h;              // node area, luma samples
  uint8_t depth, qt, bt, mt, impl_bt, last_split, part_idx, phase;
  uint8_t impl_checked, impl_split, nmodes, did_h, did_v, did_q, do_th, do_tv;
  uint8_t qt_before_bt, max_qt_sub, has_best, cur_mode, cur_split, child, nparts, first;
  uint8_t modes[8];
  uint8_t nb_ok, nbL_lh, nbL_qt, nbA_lw, nbA_qt;   // left / above CU of the node (bit0 left, bit1 above present)
  uint8_t can_mask, ctx_spl, ctx_qt, ctx_hv;       // canSplit() bits {no,qt,bh,bv,th,tv} and split-flag context increments, fixed per node
  uint8_t reusing, r_dir, r_mrl, r_cbf, r_mts;     // IS_REUSING_CU and the cached CU's mode data (BestEncInfoCache)
  uint8_t ctx_dirty;                               // the estimator's contexts differ from the node's start snapshot (a split was carried out)
  int16_t px[4], py[4], pw[4], ph[4];
  uint64_t ss;
  double max_cost;
  Sum best, temp;
};

struct Cand { uint8_t mode, mrl; };
// one ISP candidate as its wave leaves it: per sub-partition the distortion and the bits of xGetIntraFracBitsQT (0 where the early exit skipped the rate), the cbfs, how
// many sub-partitions ran; the controller replays the reference's exit logic on it (with a limit that can only be lower than the one the wave used)
struct IspRes { unsigned long long d[4], fb[4]; uint8_t cbf, nrun, mode, split, used, pad_[3]; };
// BinEncoderBase state + OutputBitstream position (EL/BinEncoder.cpp:106-371); persists in HBM between launches of a stream
struct Arith { uint32_t low, range, buffered_byte; int32_t bits_left, num_buffered; uint32_t bit_acc; int32_t bit_n; uint32_t n; };

struct CtlState {            // controller-private working set (touched by thread 0 only)
  Cand rdList[24]; double rdCost[24]; int rdSize, numRd;      // at most 3 + 4 (+ 1) candidates survive the SATD stage, + 2 MPMs
  uint8_t checked[67];
  int n_a2;
  // LFNST: the pass loop of xCheckRDCostIntra (EL/EncCu.cpp:2417-2777: transform group x lfnstIdx x mtsFlag) of the node being intra coded, and what
  // IntraSearch keeps between its passes: the SATD-stage list for the LFNST passes (m_uiSavedRdModeListLFNST, 534-566 / 681-698 / 750-775), the DCT-II
  // pass's list with its full-RD costs for the MTS passes (m_savedRdModeList / m_modeCostStore / m_bestModeCostStore, 884-916, 1262-1290)
  int8_t lfOn, grp, lf, mts, startLf, endLf, skipOther, bestMts, bestLf, considerMts, mtsUsage, testMip, lfSaved, bestValid0;
  uint8_t grpCheck[4], bestSel[4], idxOf[16];
  double dct2Cost, grpBest[4], modeCost[16], bestCost0;
  int lfNum, lfSize, mtsNum; Cand lfList[16], mtsList[16];
  // ISP (ISPTestedModesInfo and the candidate lists of the ISP tests, EL/IntraSearch.h:211-320): the tested (mode, split) pairs in test order with the sub-partitions
  // they completed and their cost, the candidate list both splits walk, the regular full-RD results (stage-B places) by cost, the SATD-stage list saved before the MRL candidates
  int8_t ispHave, ispPark, ispReqMode, ispReqSplit;      // a candidate the reference's order asks for; the wave of the current batch whose result is the node's best so far
  int8_t testIsp, ispSlot, ispPrev, ispNT, ispBestMode, ispBestSplit, ispNOrig, ispNList, ispRegN, ispHadN, skipMts2, ispBreak, ispStop[2], ispNumTotal[2], ispCandIdx[2], ispNTested[2];
  uint8_t ispTMode[16], ispTInfo[16], ispList[28], ispReg[16], ispHad[24], ispWinMode, ispWinSplit, ispWinTucbf;      // ispTInfo: split << 4 | completed sub-partitions
  double ispTCost[16], ispBestRd, ispCurBest, noIspCost;
  unsigned long long ispWinDist, ispWinBits;
  // direct look-ups of the same records (ISPTestedModesInfo keeps per-mode tables): sub-partitions completed by (split, mode), -1 = not tested; the first two modes tested with a split
  alignas(4) int8_t ispPartsOf[2][68]; int8_t ispFirst[2][2];
  uint8_t inv0[16], rdSrc[24];           // first pass: list place -> stage-B item; per list entry of an MTS pass: the first pass's item that carries its prepared DST-VII block
};

// fractional bits of a bin in probability state st (0..255): the table is symmetric, m_binFracBits[st][1] == m_binFracBits[255 - st][0] (checked by the CPU suite), so
// LDS holds the bin-0 half only
#define BIN_FRAC(st_, bin_) L.t.bin_frac[(unsigned) (st_) ^ ((0u - (unsigned) (bin_)) & 255u)]
struct Tables {                    // constant tables staged once per workgroup (LDS latency instead of global latency
  uint32_t bin_frac[256];          //  on the controller's and the rate estimator's dependent chains): m_binFracBits[state][0]; [state][1] = [255 - state][0] (BIN_FRAC)
  int32_t  qscale[12], iqscale[12];
  int16_t  ang[32], inv_ang[32];
  int8_t   gauss[128], cubic[128];
  uint8_t  last_prefix[8], mode_shift[8];
  uint8_t  ctx_rate[NCTX + 2], gorice_pars[32], gorice_pos0[96], rice_len[128], group_idx[64], mode_num[36], intra_thr[8];
  // (the transform matrices are read from constant memory through the vector L1: their LDS copies were 2.7 KB of the budget that decides how many streams share a CU,
  //  and cost 1.5 % when dropped, profiles/r03z_optimisation_levels.txt; DCT-VIII[k][i] = (-1)^k DST-VII[k][n-1-i] is read from the DST-VII rows)
  uint8_t  cg_scan[84], grp_scan[228];   // diagonal scans (CL/Rom.cpp:87-131) as x | y << 4: inside a coefficient group {4x4, 2x2, 8x2, 2x8, 16x1, 1x16}; of the groups, per (log2 wg, log2 hg)
  uint8_t  cg_inv[84], grp_inv[228];     // their inverses: raster position (y * width + x) -> scan index, same table offsets
};

#define DQ_ABS_SINGLE 64
#define RC_LIST 320
// per-wave scratch that the rate estimator (pending bin list) and the dependent quantiser (decisions, path nodes) use at different times
#define DQ_TAB_INTS (21 * 6 + 3 * 12 * 2 + 4)
struct WaveRc { uint16_t binbuf[RC_LIST]; uint8_t binsort[RC_LIST + 8]; };
// the dependent quantiser lays its per-item areas (last-position offsets, decisions, template rows) over the same bytes and on into tmp / slot behind them
// (wave_depquant_batch: 240 + 2 * positions bytes per item), so the scratch is sized by the rate estimator's lists alone
struct alignas(16) WaveScratch { WaveRc rc; };
struct WaveMem {
  WaveScratch ws;
  alignas(16) int32_t tmp[BUF];                // transform / Hadamard / scan scratch (blocks needing more use HBM scratch)
  alignas(16) int16_t slot[2 * BUF];           // candidate buffers: rec[BUF] | lev[BUF] (parked winners and big blocks live in HBM scratch)
};
#define CI_CUR 0
#define CI_W(w) (1 + (w))
struct Lds {
  Tables t;
  Ctx ctxs[1 + NW];                // [0] the estimator's contexts, [1 + w] per-wave working copy (a wave's best end-of-candidate contexts are parked in HBM)
  alignas(16) int16_t org[BUF];    // node's original tile: luma w*h, or Cb | Cr (cw*ch each); bigger nodes keep it in HBM scratch (VXD_OFF_ORG)
  alignas(16) WaveMem wm[NW];      // per wave: rate-estimator / quantiser scratch, transform scratch, candidate buffers (contiguous: the batched trellis uses all of it)
  int acc[NW][4][2];               // small-block SATD stage: per packed candidate {SAD, SATD}
  int16_t refs[4][2][140];         // luma: set 0 mrl0 unfiltered, 1 mrl0 filtered, 2 mrl1, 3 mrl3; chroma: set 0 Cb, 1 Cr. [0]=top [1]=left
  uint8_t flags[72]; int8_t src_unit[72];
  int cache_gen, cur_stream;       // CTU generation of this scratch slot (CU-result cache validity); stream descriptor the workgroup is running
  int dq_abs[DQ_ABS_SINGLE + NW];  // wave_depquant_batch: absSum per item of a batch: [0, 16) the pass's items, [32, 48) the prepared second batch; [DQ_ABS_SINGLE + wave]: a wave's own single block
  Frame fr[MAXD];
  // posted operation
  int op, op_a, op_b, op_c, op_d, op_ch;
  int pre_copy_d;                  // >= 0: before the posted operation, snapshot the estimator's contexts as the start contexts of level pre_copy_d
  int nx, ny, nw, nh, nd;          // node of the posted op (luma coordinates) and its level
  // candidates
  // (after the SATD stage the cost slots hold what the waves of an ISP batch report, one record per wave)
  Cand cand[64]; union { double cand_cost[64]; IspRes isp_res[NW]; }; int n_cand;
  // prediction parameters of each SATD-stage candidate (initPredIntraParams), packed, derived once per SATD operation; the full-RD operations keep the
  // dependent quantiser's rate tables of the node here (dq_build_tables)
  union { uint2 cand_ipa[64]; int dq_tab[DQ_TAB_INTS]; };
  Cand rd[16]; double rd_cost[16]; uint64_t rd_dist[16]; uint64_t rd_bits[16]; uint8_t rd_cbf[16]; uint8_t rd_mts[16]; int mts_evals[NW]; int n_rd;
  int do_save;                      // after_intra_op: the controller accepted the intra result
  int wave_best[NW], wave_slot[NW]; // candidate index of each wave's best and the slot that holds it
  uint8_t rd_wave[16];              // which wave evaluated (and parked) the best (mode, transform) pair of each candidate
  uint8_t rb_pairs[VXD_POOL_ITEMS]; int rb_nb;   // batched full-RD stage: the (candidate << 3 | MTS pair) items of the current chunk
  CtlState S; VxUnit cu;           // controller working set; CU record of the intra candidate being evaluated
  unsigned mpm[6], mpm_sorted[6]; int mpm_n;
  int16_t mip_n, mip_ctx;          // MIP modes of the node (0: no mip_flag) and the context of its mip_flag (3: more than 2:1)
  int16_t isp_ok, isp_pad_;        // CU::canUseISP of the node (isp_mode is then part of every line-0 luma mode's syntax)
  // LFNST: the pass of xCheckRDCostIntra being evaluated (cu.lfnstIdx, cu.mtsFlag, transform group); last scan position of the block a wave coded last;
  // per stage-B / chroma candidate: bit 0 some block's last position is beyond DC, bit 1 some block has a coefficient outside the LFNST region
  int8_t ps_lfnst, ps_mts, ps_grp, isp_wait; int rc_last[NW]; uint8_t rd_lfl[16];      // isp_wait: the ISP places follow the stage-B operation before the node's intra decision is taken
  // the DST-VII pass prepared by the DCT-II pass (stage_b_rounds): number of prepared items (0: none), item of each candidate of the running pass, absSum per item
  int8_t spec_n, spec_kind; uint8_t rd_src[16]; int spec_abs[16];      // spec_kind: 1 the DST-VII pass of transform group 0, 2 the lfnstIdx 2 pass (prepared by the lfnstIdx 1 pass)
  // ISP (intra sub-partitions): the candidate posted by the controller and what its evaluation returns; the best ISP candidate of the node so far
  double isp_limit;                                      // bestCostSoFar handed to xIntraCodingLumaISP for the candidates of the posted batch
  unsigned long long isp_dist;                           // reuse path: distortion of the cached ISP CU
  uint8_t isp_nb, isp_park, isp_split, isp_tucbf, isp_win, isp_pad2_[3];      // candidates of the posted batch; wave whose tiles OP_ISP_PARK keeps; cached CU's ispMode / cbfs (reuse); isp_win: the node's winner is the parked ISP candidate
  int lmcs_cadj, lmcs_tab;         // LMCS: chroma residual scale of the chroma node being coded (0: none), its table of quantiser constants (1 + bin; 0: unscaled)
  int dc_val[4];
  int cur_tile, frame, ctu_x, ctu_y, tree_ch;
  int d;                           // current recursion level
  // intra result of the node being evaluated
  int win_idx, win_wave;
  unsigned long long cu_bits;      // cu_pred_data + cu_residual bits of the winner (contexts left in wctx[0])
  unsigned long long cnt[4];
  VxParams par; VxFrameDev fdv;     // launch parameters and the stream's picture record: read from here inside the out-of-line functions (a reference
                                    // parameter to them is a generic pointer into the kernarg copy in scratch / into HBM: flat loads with full waits)
  // CCLM: down-sampled luma of the chroma node (nodes of at most BUF chroma samples; bigger ones in HBM scratch), availability and line parameters
  // (the luma full-RD stage keeps the fractional bits of the transform-skip context sets where the chroma operations keep the CCLM neighbour lines: ts_build_tables)
  alignas(16) int16_t lm_in[BUF / 2]; union { struct { int16_t lm_top[64], lm_left[64]; }; int ts_tab[36]; }; int lm_info[4], lm_ok, lm_nsatd; int lm_par[2][3][3]; int64_t lm_cost[8];
  int16_t fa_nb[5][4]; int fa_n, fa_res, fa_row, fa_feat[27];
  int wpp_ok, wpp_pos;             // WPP scheduler: answer of the readiness test, position the picked row continues at
  int train_row[MAXD];             // per recursion level: the row of the training dump the luma node filled at entry (-1: none); its label is written when the node is left
                                   // (kept here, behind the hot fields, so that the Frame records and everything after them stay where they were)      // FAST_ALGORITHM: neighbour CUs {x, y, w, h} of the node, forest answer, features
  Arith aw; uint8_t *aw_out; uint32_t aw_cap; int colm;      // bitstream pass: arithmetic coder, its output (HBM) and capacity; co-located luma mode of the chroma node
  unsigned long long prof[VVCX_STAMP ? 48 : 1];    // shader-clock ticks per operation kind (diagnostic build only, see vvcx_get_profile)
};

__shared__ Lds L;
#define PROF(i) L.prof[VVCX_STAMP ? (i) : 0]
// diagnostic build -DVVCX_STAMP_ISP (with VVCX_STAMP): slots 16..29 time the ISP evaluation of wave 0 and the controller's ISP phase instead of the phases
#ifdef VVCX_STAMP_ISP
#define ISP_T(n_) long long n_ = clock64()
#define ISP_ADD(slot_, a_, b_) do { if (VTX == 0) PROF(slot_) += (unsigned long long) ((b_) - (a_)); } while (0)
#define ISP_ADD0(slot_, a_, b_) do { PROF(slot_) += (unsigned long long) ((b_) - (a_)); } while (0)
#else
#define ISP_T(n_)
#define ISP_ADD(slot_, a_, b_)
#define ISP_ADD0(slot_, a_, b_)
#endif
#ifndef VX_POISON_LDS
#define VX_POISON_LDS(obj) ((void) 0)
#endif
#ifndef VX_CHECK
#define VX_CHECK(c) ((void) 0)      // the CPU emulation build asserts the ranges of derived addresses here (tools/hipemu)
#endif

// ------------------------------------------------------------------------------------------------ utilities
__device__ inline int ilog2i(int v) { return 31 - __clz(v); }
__device__ inline int imin(int a, int b) { return a < b ? a : b; }
__device__ inline int imax(int a, int b) { return a > b ? a : b; }
__device__ inline int iabs(int a) { return a < 0 ? -a : a; }
// i / d where d is usually a power of two (block and node dimensions; only clipped node areas at the picture edge are not): ld = log2 d or -1
__device__ inline int pow2_log(int d) { return (d & (d - 1)) == 0 ? 31 - __clz(d) : -1; }
__device__ inline int fast_div(int i, int d, int ld) { return ld >= 0 ? i >> ld : i / d; }
// Values that are uniform across the workgroup / wavefront at run time but that the compiler sees in VGPRs
// (LDS loads, threadIdx-derived wave index, arguments of non-inlined functions) are moved to SGPRs before they
// steer any control flow that contains a barrier, a wave barrier or a shuffle: the structurizer may otherwise
// serialise what it believes to be divergent paths around those convergent operations.
__device__ inline int uni(int v) { return __builtin_amdgcn_readfirstlane(v); }
// a pointer that is the same on every lane (arguments of non-inlined functions arrive in vector registers): into scalar registers, with everything derived from it
template <typename T> __device__ inline T *uni_p(T *p)
{
  union { T *p; int i[2]; } u; u.p = p;
  u.i[0] = __builtin_amdgcn_readfirstlane(u.i[0]); u.i[1] = __builtin_amdgcn_readfirstlane(u.i[1]);
  return u.p;
}
// a generic pointer known to point into the workgroup's LDS object, re-based on L: the accesses through it become ds_ instructions (a pointer that went through a
// call or a select is a generic one: flat_ accesses, which wait on both counters)
template <typename T> __device__ inline T *as_lds(T *p) { return (T *) ((char *) &L + (int) ((const char *) p - (const char *) &L)); }
__device__ inline double uni_d(double v)
{
  union { double d; int i[2]; } u; u.d = v;
  u.i[0] = __builtin_amdgcn_readfirstlane(u.i[0]); u.i[1] = __builtin_amdgcn_readfirstlane(u.i[1]);
  return u.d;
}
__device__ inline double lane0_d(double v)               // lane 0's value on every lane, as a scalar
{
  union { double d; int i[2]; } u; u.d = v;
  u.i[0] = __builtin_amdgcn_readlane(u.i[0], 0); u.i[1] = __builtin_amdgcn_readlane(u.i[1], 0);
  return u.d;
}
__device__ inline void wave_sync()
{
  __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
  __builtin_amdgcn_wave_barrier();
}
// wave-wide sums with DPP row operations (6 VALU instructions instead of 6 dependent ds_bpermute round trips): quad swaps,
// row rotates, then row_bcast:15 / row_bcast:31 accumulate the rows into lane 63; the total comes back as a scalar.
__device__ inline int wave_sum_i32(int v)
{
  v += __builtin_amdgcn_update_dpp(0, v, 0xB1, 0xF, 0xF, true);      // quad_perm:[1,0,3,2]
  v += __builtin_amdgcn_update_dpp(0, v, 0x4E, 0xF, 0xF, true);      // quad_perm:[2,3,0,1]
  v += __builtin_amdgcn_update_dpp(0, v, 0x124, 0xF, 0xF, true);     // row_ror:4
  v += __builtin_amdgcn_update_dpp(0, v, 0x128, 0xF, 0xF, true);     // row_ror:8: every lane holds its row's sum
  v += __builtin_amdgcn_update_dpp(0, v, 0x142, 0xA, 0xF, true);     // row_bcast:15 into rows 1, 3
  v += __builtin_amdgcn_update_dpp(0, v, 0x143, 0xC, 0xF, true);     // row_bcast:31 into rows 2, 3
  return __builtin_amdgcn_readlane(v, 63);
}
// sum over aligned groups of N (2..16) consecutive lanes, result on every lane of the group
template <int N> __device__ inline int seg_sum(int v)
{
  if (N >= 2) v += __builtin_amdgcn_update_dpp(0, v, 0xB1, 0xF, 0xF, true);     // quad_perm:[1,0,3,2]
  if (N >= 4) v += __builtin_amdgcn_update_dpp(0, v, 0x4E, 0xF, 0xF, true);     // quad_perm:[2,3,0,1]
  if (N >= 8) v += __builtin_amdgcn_update_dpp(0, v, 0x141, 0xF, 0xF, true);    // row_half_mirror: the other quad of the 8
  if (N >= 16) v += __builtin_amdgcn_update_dpp(0, v, 0x140, 0xF, 0xF, true);   // row_mirror: the other half of the 16
  return v;
}
__device__ inline unsigned long long wave_sum_u64(unsigned long long v)      // per-lane values below 2^62; 24-bit limbs cannot overflow
{
  const unsigned long long a = (unsigned) wave_sum_i32((int) (v & 0xFFFFFFu)), b = (unsigned) wave_sum_i32((int) ((v >> 24) & 0xFFFFFFu)),
                           c = (unsigned) wave_sum_i32((int) (v >> 48));
  return a + (b << 24) + (c << 48);
}
__device__ inline double rd_cost(const VxParams &p, uint64_t bits, uint64_t dist)
{
  // CL/RdCost.cpp:63-74: two roundings (mul, add); the file is built with -ffp-contract=off
  const double a = p.dist_scale * (double) dist;
  return a + (double) bits;
}
template <typename T> __device__ inline int ld_px(const void *plane, int idx) { return (int) ((const T *) plane)[idx]; }
template <typename T> __device__ inline void st_px(void *plane, int idx, int v) { ((T *) plane)[idx] = (T) v; }

// ------------------------------------------------------------------------------------------------ CABAC estimator
// BinProbModel_Std + BitEstimator (CL/Contexts.h:86-155, EL/BinEncoder.h:238-303)
// OutputBitstream::write (MSB first) / BinEncoderBase::writeOut, encodeBinsEP, encodeBinTrm, finish: thread 0, bitstream pass only
__device__ inline void bs_write(uint32_t v, int nbits)
{
  for (int i = nbits - 1; i >= 0; i--) {
    L.aw.bit_acc = (L.aw.bit_acc << 1) | ((v >> i) & 1); L.aw.bit_n++;
    if (L.aw.bit_n == 8) { if (L.aw.n < L.aw_cap) L.aw_out[L.aw.n] = (uint8_t) L.aw.bit_acc; L.aw.n++; L.aw.bit_acc = 0; L.aw.bit_n = 0; }
  }
}
__device__ __noinline__ void arith_write_out()
{
  Arith &a = L.aw;
  const uint32_t lead = a.low >> (24 - a.bits_left);
  a.bits_left += 8;
  a.low &= 0xffffffffu >> a.bits_left;
  if (lead == 0xff) a.num_buffered++;
  else if (a.num_buffered > 0) {
    const uint32_t carry = lead >> 8;
    uint32_t byte = a.buffered_byte + carry;
    a.buffered_byte = lead & 0xff;
    bs_write(byte, 8);
    byte = (0xff + carry) & 0xff;
    while (a.num_buffered > 1) { bs_write(byte, 8); a.num_buffered--; }
  } else { a.num_buffered = 1; a.buffered_byte = lead; }
}
__device__ inline void arith_start() { Arith &a = L.aw; a.low = 0; a.range = 510; a.buffered_byte = 0xff; a.num_buffered = 0; a.bits_left = 23; a.bit_acc = 0; a.bit_n = 0; a.n = 0; }
__device__ __noinline__ void arith_bins_ep(uint32_t bins, int n)
{
  Arith &a = L.aw;
  while (n > 8) {
    n -= 8;
    const uint32_t pattern = bins >> n;
    a.low <<= 8; a.low += a.range * pattern; bins -= pattern << n; a.bits_left -= 8;
    if (a.bits_left < 12) arith_write_out();
  }
  a.low <<= n; a.low += a.range * bins; a.bits_left -= n;
  if (a.bits_left < 12) arith_write_out();
}
__device__ __noinline__ void arith_trm(unsigned bin)
{
  Arith &a = L.aw;
  a.range -= 2;
  if (bin) { a.low += a.range; a.low <<= 7; a.range = 2 << 7; a.bits_left -= 7; }
  else if (a.range >= 256) return;
  else { a.low <<= 1; a.range <<= 1; a.bits_left--; }
  if (a.bits_left < 12) arith_write_out();
}
__device__ __noinline__ void arith_finish()
{
  Arith &a = L.aw;
  if (a.low >> (32 - a.bits_left)) {
    bs_write(a.buffered_byte + 1, 8);
    while (a.num_buffered > 1) { bs_write(0x00, 8); a.num_buffered--; }
    a.low -= 1u << (32 - a.bits_left);
  } else {
    if (a.num_buffered > 0) bs_write(a.buffered_byte, 8);
    while (a.num_buffered > 1) { bs_write(0xff, 8); a.num_buffered--; }
  }
  bs_write(a.low >> 8, 24 - a.bits_left);
  bs_write(1, 1); while (a.bit_n) bs_write(0, 1);       // OutputBitstream::writeByteAlignment
}
__device__ __noinline__ void arith_bin(unsigned bin, unsigned st)      // TBinEncoder::encodeBin 378-420 with the state before its update
{
  Arith &a = L.aw;
  unsigned q = st & 0xff; const unsigned mps = q >> 7;
  if (q & 0x80) q ^= 0xff;
  const uint32_t lps = (((q >> 2) * (a.range >> 5)) >> 1) + 4;
  a.range -= lps;
  if (bin != mps) {
    const int nb = lps >= 128 ? 1 : lps >= 64 ? 2 : lps >= 32 ? 3 : lps >= 16 ? 4 : lps >= 8 ? 5 : 6;      // m_RenormTable_32[lps >> 3] (CL/Contexts.cpp:45-55)
    a.bits_left -= nb; a.low += a.range; a.low <<= nb; a.range = lps << nb;
    if (a.bits_left < 12) arith_write_out();
  } else if (a.range < 256) {
    a.bits_left -= 1; a.low <<= 1; a.range <<= 1;
    if (a.bits_left < 12) arith_write_out();
  }
}
// WR: also drive the arithmetic coder L.aw (bitstream pass only; a template so that the estimator's copies contain no call)
template <bool WR = false>
__device__ inline void enc_bin(Cab &cb, unsigned bin, int ctx)
{
  Ctx *c = &L.ctxs[cb.ci];
  const unsigned st = (unsigned) (c->s0[ctx] + c->s1[ctx]) >> 8;
  cb.bits += BIN_FRAC(st, bin);
  if (WR) arith_bin(bin, st);
  const int rate = L.t.ctx_rate[ctx];
  const int r0 = 2 + ((rate >> 2) & 3), r1 = 3 + r0 + (rate & 3);
  unsigned a = c->s0[ctx], b = c->s1[ctx];
  a -= (a >> r0) & 0x7FE0u; b -= (b >> r1) & 0x7FFEu;
  if (bin) { a += (0x7fffu >> r0) & 0x7FE0u; b += (0x7fffu >> r1) & 0x7FFEu; }
  c->s0[ctx] = (uint16_t) a; c->s1[ctx] = (uint16_t) b;
}
// n bypass bins of the given value, MSB first (the estimator only needs n)
template <bool WR = false>
__device__ inline void enc_ep(Cab &cb, uint32_t value, int n) { cb.bits += (uint64_t) n << 15; if (WR && n > 0) arith_bins_ep(value, n); }

__device__ void cg_shape(int w, int h, int &lcw, int &lch)     // g_log2SbbSize, CL/Rom.cpp:250-261
{
  const int lw = ilog2i(w), lh = ilog2i(h);
  if (lw >= 2 && lh >= 2) { lcw = 2; lch = 2; return; }
  if (lh == 0) { lcw = imin(lw, 4); lch = 0; return; }             // N x 1 / 1 x N sub-partitions of ISP CUs: groups of 16 x 1 / 1 x 16
  if (lw == 0) { lcw = 0; lch = imin(lh, 4); return; }
  if (lh == 1) { lcw = lw >= 3 ? 3 : 1; lch = 1; return; }
  lcw = 1; lch = lh >= 3 ? 3 : 1;
}
// where the scan of a coefficient group shape starts in Tables::cg_scan / cg_inv: 4x4, 2x2, 8x2, 2x8, 16x1, 1x16
__device__ inline int cg_tab_off(int lcw, int lch) { return lcw == 2 ? 0 : lcw == 3 ? 20 : lch == 3 ? 36 : lcw == 4 ? 52 : lch == 4 ? 68 : 16; }
// position of scan index sp inside a bw x bh diagonal scan (CL/Rom.cpp:87-131), closed form by walking
__device__ void diag_pos(int bw, int bh, int n, int &ox, int &oy)
{
  int line = 0, col = 0;
  for (int i = 0; i < n; i++) {
    if (col == bw - 1 || line == 0) { line += col + 1; col = 0; if (line >= bh) { col += line - (bh - 1); line = bh - 1; } }
    else { col++; line--; }
  }
  ox = col; oy = line;
}
template <bool WR = false>
__device__ inline void enc_rem_abs(Cab &cb, unsigned bins, unsigned rice)       // EL/BinEncoder.cpp:444-472
{
  const unsigned thr = 5u << rice;
  if (bins < thr) { const unsigned len = (bins >> rice) + 1; enc_ep<WR>(cb, (1u << len) - 2, (int) len); enc_ep<WR>(cb, bins & ((1u << rice) - 1), (int) rice); return; }
  const unsigned maxPrefix = 32 - 5 - 15;
  unsigned prefix = 0, suffix, code = (bins >> rice) - 5;
  if (code >= ((1u << maxPrefix) - 1)) { prefix = maxPrefix; suffix = 15; }
  else { while (code > ((2u << prefix) - 2)) prefix++; suffix = prefix + rice + 1; }
  enc_ep<WR>(cb, (1u << (prefix + 5)) - 1, (int) (prefix + 5));
  enc_ep<WR>(cb, ((code - ((1u << prefix) - 1)) << rice) | (bins & ((1u << rice) - 1)), (int) suffix);
}
struct Cctx { int w, h, ch, tmpl_diag, tmpl_sum1; };
__device__ int sig_ctx(Cctx &c, const int16_t *coeff, int blk)      // CL/ContextModelling.h:107-156 (state 0)
{
  const int W = c.w, H = c.h, posY = blk >> ilog2i(W), posX = blk & (W - 1);
  const int16_t *p = coeff + blk;
  const int diag = posX + posY;
  int numPos = 0, sumAbs = 0;
#define UPD(v) { int a = iabs(v); sumAbs += imin(4 + (a & 1), a); numPos += !!a; }
  if (posX < W - 1) { UPD(p[1]); if (posX < W - 2) UPD(p[2]); if (posY < H - 1) UPD(p[W + 1]); }
  if (posY < H - 1) { UPD(p[W]); if (posY < H - 2) UPD(p[W << 1]); }
#undef UPD
  int ofs = imin((sumAbs + 1) >> 1, 3) + (diag < 2 ? 4 : 0);
  if (c.ch == 0) ofs += diag < 5 ? 4 : 0;
  c.tmpl_diag = diag; c.tmpl_sum1 = sumAbs - numPos;
  return VX_CTX_SigFlag[c.ch] + ofs;
}
__device__ int tmpl_abs_sum(const Cctx &c, const int16_t *coeff, int blk, int base)
{
  const int W = c.w, H = c.h, posY = blk >> ilog2i(W), posX = blk & (W - 1);
  const int16_t *p = coeff + blk;
  int sum = 0;
  if (posX < W - 1) { sum += iabs(p[1]); if (posX < W - 2) sum += iabs(p[2]); if (posY < H - 1) sum += iabs(p[W + 1]); }
  if (posY < H - 1) { sum += iabs(p[W]); if (posY < H - 2) sum += iabs(p[W << 1]); }
  return imax(imin(sum - 5 * base, 31), 0);
}
// EL/CABACWriter.cpp residual_coding 3773-3883, last_sig_coeff 4102-4160, residual_coding_subblock 4164-4304
// (regular residual, DepQuant off, sign hiding off).  Coefficient tile stride = w.
// Split in two: a pre-pass that builds the scan table (CL/Rom.cpp:87-370) and finds the last significant scan
// position and the significant coefficient groups (3823-3835) — data parallel, done by all lanes of the calling wave
// when PAR — and the context-coded part, which is a serial chain through the adaptive models (lane 0).
struct RcPre { int last; unsigned long long sig_groups, sig_raster; };   // significant groups by scan index / by raster position
__device__ inline void diag_walk(int bw, int bh, int n, int &ox, int &oy)
{
  int line = 0, col = 0;
  for (int i = 0; i < n; i++) {
    if (col == bw - 1 || line == 0) { line += col + 1; col = 0; if (line >= bh) { col += line - (bh - 1); line = bh - 1; } }
    else { col++; line--; }
  }
  ox = col; oy = line;
}
// scan geometry of a transform block: coefficient group shape (g_log2SbbSize), the two diagonal-scan tables and the mapping
// scan position -> raster offset (CL/Rom.cpp:133-370, zero-out region 32x32)
struct ScanGeo { const uint8_t *cg, *grp; int lcw, lch, lcg, w, wg, hg, nscan; };
__device__ inline ScanGeo scan_geo(int w, int h)
{
  ScanGeo g; cg_shape(w, h, g.lcw, g.lch);
  g.lcg = g.lcw + g.lch; g.w = w;
  const int zw = imin(32, w), zh = imin(32, h);
  g.wg = zw >> g.lcw; g.hg = zh >> g.lch; g.nscan = zw * zh;
  g.cg = L.t.cg_scan + cg_tab_off(g.lcw, g.lch);
  const int a = ilog2i(g.wg), b = ilog2i(g.hg);
  g.grp = L.t.grp_scan + 15 * ((1 << a) - 1) + (1 << a) * ((1 << b) - 1);
  return g;
}
__device__ inline int scan_blk(const ScanGeo &g, int sp)
{
  const unsigned gi = g.grp[sp >> g.lcg], ci = g.cg[sp & ((1 << g.lcg) - 1)];
  return (int) ((((gi >> 4) << g.lch) + (ci >> 4)) * g.w + ((gi & 15) << g.lcw) + (ci & 15));
}
// single-lane pre-pass: scan table, last significant scan position, significant groups (3823-3835)
__device__ __noinline__ RcPre rc_prepass_serial(const int16_t *coeff, int w, int h, uint16_t *scan)
{
  const ScanGeo g = scan_geo(w, h);
  int last = -1; unsigned long long sig = 0;
  for (int sp = 0; sp < g.nscan; sp++) {
    const int blk = scan_blk(g, sp);
    scan[sp] = (uint16_t) blk;
    if (coeff[blk]) { last = sp; sig |= 1ull << (sp >> g.lcg); }
  }
  RcPre r; r.last = last; r.sig_groups = sig; r.sig_raster = 0;
  return r;
}
// wave pre-pass: 64 scan positions per step, last position and group significance from the ballot of "coefficient != 0"
// SMALL: the levels are the calling wave's LDS slot (L.wm[wave].slot + 1024 + lev_off); else coeff_g (HBM).  The LDS pointer is formed
// from L inside the function: a pointer handed through a call is a generic one and every access through it a flat_ instruction
// (longer path to LDS, and its completion can only be awaited with vmcnt(0) + lgkmcnt(0)).
template <bool SMALL>
__device__ __noinline__ RcPre rc_prepass_wave(int lev_off, const int16_t *coeff_g, int w, int h, int lane)
{
  coeff_g = uni_p(coeff_g);
  w = uni(w); h = uni(h);
  const int16_t *coeff = SMALL ? L.wm[uni(VTX >> 6)].slot + BUF + uni(lev_off) : coeff_g;
  const ScanGeo g = scan_geo(w, h);
  int last = -1; unsigned long long sig = 0;
  const int gpi = 64 >> g.lcg;                           // groups per step
  const unsigned long long gmask = (g.lcg == 4) ? 0xFFFFull : 0xFull;
  for (int it = 0; it * 64 < g.nscan; it++) {
    const int sp = it * 64 + lane;
    const unsigned long long m = __ballot(sp < g.nscan && coeff[scan_blk(g, sp)] != 0);
    if (m) {
      last = it * 64 + 63 - __clzll((long long) m);
      for (int q = 0; q < gpi; q++) if ((m >> (q << g.lcg)) & gmask) sig |= 1ull << (it * gpi + q);
    }
  }
  // the same set by raster position of the group (for the sigGroup context, 4177-4181): one group per lane
  int rlo = 0, rhi = 0;
  if (lane < g.wg * g.hg && ((sig >> lane) & 1)) { const unsigned gi = g.grp[lane]; const int r = (int) (gi >> 4) * g.wg + (int) (gi & 15); if (r < 32) rlo = 1 << r; else rhi = 1 << (r - 32); }
  RcPre r; r.last = last; r.sig_groups = sig;
  r.sig_raster = ((unsigned long long) (unsigned) wave_sum_i32(rhi) << 32) | (unsigned) wave_sum_i32(rlo);
  return r;
}
template <bool WR = false>
__device__ __noinline__ void rc_serial(Cab &cb, const int16_t *coeff, int w, int h, int is_chroma, const uint16_t *scan, RcPre pre, int zo = 0)
{
  Cctx c; c.w = w; c.h = h; c.ch = is_chroma; c.tmpl_diag = -1; c.tmpl_sum1 = -1;
  int lcw, lch; cg_shape(w, h, lcw, lch);
  const int lcg = lcw + lch, cgSize = 1 << lcg;
  const int zw = imin(32, w), zh = imin(32, h), wg = zw >> lcw, hg = zh >> lch;
  const ScanGeo geo = scan_geo(w, h);
  const int scanPosLast = pre.last;
  const unsigned long long sigGroups = pre.sig_groups;
  if (scanPosLast < 0) return;
  const int l2w = ilog2i(w), l2h = ilog2i(h);
  int offx = 0, offy = 0, shx, shy;
  if (is_chroma) { shx = imin(2, w >> 3); shy = imin(2, h >> 3); }
  else { offx = L.t.last_prefix[l2w]; offy = L.t.last_prefix[l2h]; shx = (l2w + 1) >> 2; shy = (l2h + 1) >> 2; }
  {
    const int blk = scan[scanPosLast];
    const int posY = blk >> ilog2i(w), posX = blk & (w - 1);
    const int gx = L.t.group_idx[posX], gy = L.t.group_idx[posY];
    const int maxX = L.t.group_idx[((zo && w == 32) ? 16 : zw) - 1], maxY = L.t.group_idx[((zo && h == 32) ? 16 : zh) - 1];      // 4115-4126: a 32-point MTS side codes positions below 16
    int k;
    for (k = 0; k < gx; k++) enc_bin<WR>(cb, 1, VX_CTX_LastX[c.ch] + offx + (k >> shx));
    if (gx < maxX) enc_bin<WR>(cb, 0, VX_CTX_LastX[c.ch] + offx + (k >> shx));
    for (k = 0; k < gy; k++) enc_bin<WR>(cb, 1, VX_CTX_LastY[c.ch] + offy + (k >> shy));
    if (gy < maxY) enc_bin<WR>(cb, 0, VX_CTX_LastY[c.ch] + offy + (k >> shy));
    if (gx > 3) enc_ep<WR>(cb, (uint32_t) (posX - VX_MIN_IN_GROUP[gx]), (gx - 2) >> 1);
    if (gy > 3) enc_ep<WR>(cb, (uint32_t) (posY - VX_MIN_IN_GROUP[gy]), (gy - 2) >> 1);
  }
  int regBins = (((zo && w == 32) ? 16 : zw) * ((zo && h == 32) ? 16 : zh) * 28) >> 4;      // getTbAreaAfterCoefZeroOut (CL/Unit.cpp:872-890)
  const int stateTab = (L.par.tools & TOOL_DEPQUANT) ? 32040 : 0; int state = 0;            // 3857-3858
  unsigned long long sigPos = 0;         // m_sigCoeffGroupFlag by CG raster position
  for (int sub = scanPosLast >> lcg; sub >= 0; sub--) {
    const int cgX = geo.grp[sub] & 15, cgY = geo.grp[sub] >> 4, cgPos = cgY * wg + cgX;
    const int minSub = sub << lcg, maxSub = minSub + cgSize - 1;
    if ((sigGroups >> sub) & 1) sigPos |= 1ull << cgPos;
    const int sigRight = (cgX + 1) < wg ? (int) ((sigPos >> (cgPos + 1)) & 1) : 0;
    const int sigLower = (cgY + 1) < hg ? (int) ((sigPos >> (cgPos + wg)) & 1) : 0;
    const int sigGroupCtx = VX_CTX_SigCoeffGroup[c.ch] + (sigRight | sigLower);
    if (zo && ((h == 32 && cgY >= (16 >> lch)) || (w == 32 && cgX >= (16 >> lcw)))) continue;       // 3866-3877: groups of the zeroed area of a 32-point MTS block
    const int isLast = (scanPosLast >> lcg) == sub, isNotFirst = sub != 0;
    const int firstSigPos = isLast ? scanPosLast : maxSub;
    int nextSigPos = firstSigPos;
    if (!isLast && isNotFirst) {
      if ((sigPos >> cgPos) & 1) enc_bin<WR>(cb, 1, sigGroupCtx);
      else { enc_bin<WR>(cb, 0, sigGroupCtx); continue; }
    }
    const int inferSigPos = nextSigPos != scanPosLast ? (isNotFirst ? minSub : -1) : nextSigPos;
    int numNonZero = 0, remRegBins = regBins;
    uint32_t signPattern = 0;
    for (; nextSigPos >= minSub && remRegBins >= 4; nextSigPos--) {
      const int blk = scan[nextSigPos];
      const int cf = coeff[blk];
      const unsigned sigFlag = cf != 0;
      if (numNonZero || nextSigPos != inferSigPos) { const int ctx = sig_ctx(c, coeff, blk) + VX_CTX_SigFlag[c.ch + 2 * imax(0, state - 1)] - VX_CTX_SigFlag[c.ch]; enc_bin<WR>(cb, sigFlag, ctx); remRegBins--; }
      else if (nextSigPos != scanPosLast) sig_ctx(c, coeff, blk);
      if (sigFlag) {
        int off = 0;                       // ctxOffsetAbs (158-167)
        if (c.tmpl_diag != -1) {
          off = imin(c.tmpl_sum1, 4) + 1;
          off += (!c.tmpl_diag ? (c.ch == 0 ? 15 : 5) : c.ch == 0 ? (c.tmpl_diag < 3 ? 10 : (c.tmpl_diag < 10 ? 5 : 0)) : 0);
        }
        numNonZero++;
        signPattern = (signPattern << 1) | (cf < 0);
        int rem = iabs(cf) - 1;
        const unsigned gt1 = !!rem;
        enc_bin<WR>(cb, gt1, VX_CTX_GtxFlag[c.ch + 2] + off); remRegBins--;
        if (gt1) {
          rem -= 1;
          enc_bin<WR>(cb, rem & 1, VX_CTX_ParFlag[c.ch] + off); rem >>= 1; remRegBins--;
          enc_bin<WR>(cb, !!rem, VX_CTX_GtxFlag[c.ch] + off); remRegBins--;
        }
      }
      state = (stateTab >> ((state << 2) + ((cf & 1) << 1))) & 3;
    }
    const int firstPosMode2 = nextSigPos;
    regBins = remRegBins;
    for (int sp = firstSigPos; sp > firstPosMode2; sp--) {
      const int blk = scan[sp];
      const unsigned a = (unsigned) iabs(coeff[blk]);
      if (a >= 4) enc_rem_abs<WR>(cb, (a - 4) >> 1, L.t.gorice_pars[tmpl_abs_sum(c, coeff, blk, 4)]);
    }
    for (int sp = firstPosMode2; sp >= minSub; sp--) {
      const int blk = scan[sp];
      const unsigned a = (unsigned) iabs(coeff[blk]);
      const int sumAll = tmpl_abs_sum(c, coeff, blk, 0);
      const unsigned rice = L.t.gorice_pars[sumAll], pos0 = L.t.gorice_pos0[imax(0, state - 1) * 32 + sumAll];
      enc_rem_abs<WR>(cb, a == 0 ? pos0 : a <= pos0 ? a - 1 : a, rice);
      state = (stateTab >> ((state << 2) + ((a & 1) << 1))) & 3;
      if (a) { numNonZero++; signPattern = (signPattern << 1) | (coeff[blk] < 0); }
    }
    enc_ep<WR>(cb, signPattern, numNonZero);
  }
}
// single-lane form (controller / estimator pass) and wave form (lane 0 owns cb)
template <bool WR = false>
__device__ void residual_coding(Cab &cb, const int16_t *coeff, int w, int h, int is_chroma, uint16_t *scan, int zo = 0)
{
  const RcPre pre = rc_prepass_serial(coeff, w, h, scan);
  L.rc_last[0] = pre.last;                                 // thread 0 (estimator pass / writer): read by the CU-level LFNST signalling
  rc_serial<WR>(cb, coeff, w, h, is_chroma, scan, pre, zo);
}
// Wave form of residual_coding (same syntax as rc_serial, all 64 lanes working):
//  (1) data-parallel pre-pass: scan table, last position, significant groups (by scan index and by raster position);
//  (2) 64 scan positions at a time, one per lane in coding order: the lane derives its neighbourhood template
//      (CL/ContextModelling.h:107-199), which of {sig, gt1, par, gt2} it codes and with which contexts, its position in the
//      bin sequence (ballot prefix sums; the regular-bin budget is a prefix condition) and its bypass bit count.  Which context
//      a bin uses never depends on the adaptive state, so the bins are *emitted* as (ctx, bin) pairs without touching a model;
//  (3) the adaptive models: bins of different contexts are independent chains.  The list is grouped by context (ballot per
//      distinct context), lane t then replays segment t in order (BinProbModel_Std::estFracBitsUpdate), all chains in parallel.
__device__ inline int lane_prefix(unsigned long long m)       // set bits of m below this lane
{
  return (int) __builtin_amdgcn_mbcnt_hi((unsigned) (m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned) m, 0u));
}
__device__ inline int prefix3(int v, int &total)               // exclusive prefix sum over the wave of a value in 0..7
{
  const unsigned long long b0 = __ballot(v & 1), b1 = __ballot(v & 2), b2 = __ballot(v & 4);
  total = __popcll(b0) + 2 * __popcll(b1) + 4 * __popcll(b2);
  return lane_prefix(b0) + 2 * lane_prefix(b1) + 4 * lane_prefix(b2);
}
__device__ inline int rem_abs_len(unsigned bins, unsigned rice)   // bit count of enc_rem_abs (EL/BinEncoder.cpp:444-472)
{
  const unsigned thr = 5u << rice;
  if (bins < thr) return (int) ((bins >> rice) + 1 + rice);
  const unsigned maxPrefix = 32 - 5 - 15;
  unsigned prefix = 0, suffix, code = (bins >> rice) - 5;
  if (code >= ((1u << maxPrefix) - 1)) { prefix = maxPrefix; suffix = 15; }
  else { while (code > ((2u << prefix) - 2)) prefix++; suffix = prefix + rice + 1; }
  return (int) (5 + prefix + suffix);
}
#define RC_MAXJ 5             // pending list holds at most RC_MAXJ * 64 bins
__device__ __noinline__ void rc_chain(int ci, int wv, int nb, int lane, unsigned long long &bits)
{
  nb = uni(nb); ci = uni(ci); wv = uni(wv);
  Ctx *c = &L.ctxs[ci]; const uint16_t *bb = L.wm[wv].ws.rc.binbuf; uint8_t *sorted = L.wm[wv].ws.rc.binsort;
  int ctxv[RC_MAXJ]; unsigned binv[RC_MAXJ]; unsigned long long rem[RC_MAXJ];
#pragma unroll
  for (int j = 0; j < RC_MAXJ; j++) {
    const int k = lane + 64 * j;
    const unsigned e = (j * 64 < nb && k < nb) ? bb[k] : 0xffffu;
    ctxv[j] = e == 0xffffu ? -1 : (int) (e >> 1); binv[j] = e & 1;
    rem[j] = j * 64 < nb ? __ballot(ctxv[j] >= 0) : 0ull;
  }
  int segpos = 0;
  for (;;) {
    int nseg = 0, myctx = 0, mystart = 0, mylen = 0;
    for (;;) {                                           // group the pending bins by context, first occurrence first
      int v = -1;
#pragma unroll
      for (int j = 0; j < RC_MAXJ; j++) if (v < 0 && rem[j]) v = __builtin_amdgcn_readlane(ctxv[j], __ffsll(rem[j]) - 1);
      if (v < 0) break;
      const int start = segpos;
#pragma unroll
      for (int j = 0; j < RC_MAXJ; j++) if (rem[j]) {
        const unsigned long long m = __ballot(ctxv[j] == v);
        if (ctxv[j] == v) sorted[segpos + lane_prefix(m)] = (uint8_t) binv[j];
        segpos += __popcll(m); rem[j] &= ~m;
      }
      if (lane == nseg) { myctx = v; mystart = start; mylen = segpos - start; }
      if (++nseg == 64) break;
    }
    if (nseg == 0) break;
    wave_sync();
    if (lane < nseg) {
      unsigned a = c->s0[myctx], b = c->s1[myctx];
      const int rate = L.t.ctx_rate[myctx];
      const int r0 = 2 + ((rate >> 2) & 3), r1 = 3 + r0 + (rate & 3);
      const unsigned i0 = (0x7fffu >> r0) & 0x7FE0u, i1 = (0x7fffu >> r1) & 0x7FFEu;
      // the only loop-carried chain is the state update: the next bin is fetched one step ahead and the fractional-bit
      // lookup of a step is consumed one step later, so no LDS latency sits on it
      unsigned acc = 0, pend = 0, nxt = sorted[mystart];
      for (int k = 0; k < mylen; k++) {
        const unsigned bin = nxt;
        nxt = sorted[mystart + k + 1];
        acc += pend;
        pend = BIN_FRAC((a + b) >> 8, bin);
        a -= (a >> r0) & 0x7FE0u; b -= (b >> r1) & 0x7FFEu;
        if (bin) { a += i0; b += i1; }
      }
      c->s0[myctx] = (uint16_t) a; c->s1[myctx] = (uint16_t) b;
      bits += acc + pend;
    }
    wave_sync();
    if (nseg < 64) break;
  }
}
template <bool SMALL>
__device__ __noinline__ void residual_coding_wave(Cab &cb, int lev_off, const int16_t *coeff_g, int w, int h, int is_chroma, int lane, int zo = 0)
{
  coeff_g = uni_p(coeff_g); lev_off = uni(lev_off);
  zo = uni(zo);
  const int dq = uni((int) (L.par.tools & TOOL_DEPQUANT)) != 0;       // dep_quant_enabled_flag: the quantiser state picks the sig_coeff_flag context set and the bypass zero position
  int dqx = 0, dqy = 0;                                                 // state bits in front of lane 0 of the current 64 positions (see wave_dequant_dq)
  const long long q0 = STAMP();
  const int16_t *coeff = SMALL ? L.wm[uni(VTX >> 6)].slot + BUF + uni(lev_off) : coeff_g;
  const RcPre pre = rc_prepass_wave<SMALL>(lev_off, coeff_g, w, h, lane);
  const int last = uni(pre.last);
  if (lane == 0) L.rc_last[uni(VTX >> 6)] = last;
  if (last < 0) return;
  w = uni(w); h = uni(h); is_chroma = uni(is_chroma);
  const ScanGeo geo = scan_geo(w, h);
  const long long q1 = STAMP();
  long long qe = 0, qc = 0;
  const int wv = uni(VTX >> 6);
  uint16_t *bb = L.wm[wv].ws.rc.binbuf; uint8_t *sorted = L.wm[wv].ws.rc.binsort;
  const int lcg = geo.lcg, cgSize = 1 << lcg;
  const int zw = imin(32, w), zh = imin(32, h), wg = geo.wg, hg = geo.hg;
  const unsigned long long sigGroups = pre.sig_groups, sigRaster = pre.sig_raster;
  unsigned long long mybits = 0;
  int nb;
  {                                                    // last_sig_coeff (4102-4160): one bin per lane
    const int l2w = ilog2i(w), l2h = ilog2i(h);
    int offx = 0, offy = 0, shx, shy;
    if (is_chroma) { shx = imin(2, w >> 3); shy = imin(2, h >> 3); }
    else { offx = L.t.last_prefix[l2w]; offy = L.t.last_prefix[l2h]; shx = (l2w + 1) >> 2; shy = (l2h + 1) >> 2; }
    const int blk = scan_blk(geo, last);
    const int posY = blk >> ilog2i(w), posX = blk & (w - 1);
    const int gx = uni(L.t.group_idx[posX]), gy = uni(L.t.group_idx[posY]);
    const int maxX = L.t.group_idx[((zo && w == 32) ? 16 : zw) - 1], maxY = L.t.group_idx[((zo && h == 32) ? 16 : zh) - 1];      // 4115-4126: a 32-point MTS side codes positions below 16
    const int nX = gx + (gx < maxX), nY = gy + (gy < maxY);
    if (lane < nX) bb[lane] = (uint16_t) (((VX_CTX_LastX[is_chroma] + offx + (lane >> shx)) << 1) | (lane < gx));
    if (lane < nY) bb[nX + lane] = (uint16_t) (((VX_CTX_LastY[is_chroma] + offy + (lane >> shy)) << 1) | (lane < gy));
    if (lane == 0) mybits += (unsigned long long) ((gx > 3 ? (gx - 2) >> 1 : 0) + (gy > 3 ? (gy - 2) >> 1 : 0)) << 15;
    nb = nX + nY;
  }
  const int sigBase = VX_CTX_SigFlag[is_chroma], g1Base = VX_CTX_GtxFlag[is_chroma + 2], g2Base = VX_CTX_GtxFlag[is_chroma], parBase = VX_CTX_ParFlag[is_chroma];
  const int grpBase = VX_CTX_SigCoeffGroup[is_chroma];
  int regBins = (((zo && w == 32) ? 16 : zw) * ((zo && h == 32) ? 16 : zh) * 28) >> 4;
  const int zoX = (zo && w == 32) ? (16 >> geo.lcw) : 64, zoY = (zo && h == 32) ? (16 >> geo.lch) : 64;     // first group column / row of the zeroed area
  const int lastCG = last >> lcg;
  const unsigned long long ltMask = lane == 0 ? 0ull : (~0ull >> (64 - lane));
  for (int c = last >> 6; c >= 0; c--) {
    const long long e0 = STAMP();
    const int sp = c * 64 + 63 - lane;
    const bool active = sp <= last;
    const int sub = sp >> lcg, inCG = (cgSize - 1) - (sp & (cgSize - 1));      // inCG 0 = first position of its group in coding order
    const bool isLastCG = sub == lastCG, notFirst = sub != 0;
    const bool coded = active && (isLastCG || !notFirst || ((sigGroups >> sub) & 1));
    int cf = 0, sigofs = 0, goff = 0, sumPlain = 0;
    if (coded) {
      const int blk = scan_blk(geo, sp);
      const int posY = blk >> ilog2i(w), posX = blk & (w - 1);
      const int16_t *pc = coeff + blk;
      cf = pc[0];
      const int diag = posX + posY;
      int numPos = 0, sumAbs = 0;
#define UPD(v) { int a_ = iabs(v); sumAbs += imin(4 + (a_ & 1), a_); numPos += !!a_; sumPlain += a_; }
      if (posX < w - 1) { UPD(pc[1]); if (posX < w - 2) UPD(pc[2]); if (posY < h - 1) UPD(pc[w + 1]); }
      if (posY < h - 1) { UPD(pc[w]); if (posY < h - 2) UPD(pc[w << 1]); }
#undef UPD
      sigofs = imin((sumAbs + 1) >> 1, 3) + (diag < 2 ? 4 : 0);
      if (!is_chroma) sigofs += diag < 5 ? 4 : 0;
      goff = imin(sumAbs - numPos, 4) + 1;
      goff += (!diag ? (is_chroma ? 5 : 15) : !is_chroma ? (diag < 3 ? 10 : (diag < 10 ? 5 : 0)) : 0);
      if (sp == last) goff = 0;                          // m_tmplCpDiag == -1 only for the very first coefficient
    }
    const int a = iabs(cf);
    int qstate = 0;
    if (dq) {
      const unsigned long long Pm = __ballot(a & 1);
      const unsigned long long EVEN = 0x5555555555555555ull, ODD = 0xAAAAAAAAAAAAAAAAull;
      const int odd = lane & 1;
      const int xs = (odd ? dqy : dqx) ^ (__popcll(Pm & ltMask & (odd ? EVEN : ODD)) & 1), ys = (odd ? dqx : dqy) ^ (__popcll(Pm & ltMask & (odd ? ODD : EVEN)) & 1);
      qstate = (xs << 1) | ys;
      dqx ^= __popcll(Pm & ODD) & 1; dqy ^= __popcll(Pm & EVEN) & 1;
    }
    const unsigned long long nz = __ballot(coded && a != 0);
    const int cgStart = lane - inCG;
    const unsigned long long cgLt = ltMask & ~(cgStart <= 0 ? 0ull : (~0ull >> (64 - cgStart)));
    const bool nzBefore = (nz & cgLt) != 0;
    const int infer = isLastCG ? last : (notFirst ? (sub << lcg) : -1);
    const bool sigCoded = coded && (nzBefore || sp != infer);
    const int nbins = coded ? ((sigCoded ? 1 : 0) + (a ? (a > 1 ? 3 : 1) : 0)) : 0;
    int tot;
    const int before = prefix3(nbins, tot);
    const bool ctxMode = coded && (regBins - before >= 4);          // the budget test of 4187 is a prefix condition
    const int eff = ctxMode ? nbins : 0;
    int used; prefix3(eff, used);
    bool gflag = active && inCG == 0 && !isLastCG && notFirst;
    if (zo) gflag = gflag && (int) (geo.grp[sub] & 15) < zoX && (int) (geo.grp[sub] >> 4) < zoY;
    int o = nb + prefix3(eff + (gflag ? 1 : 0), tot);
    if (gflag) {
      const int cgX = geo.grp[sub] & 15, cgY = geo.grp[sub] >> 4, cgPos = cgY * wg + cgX;
      const int sigRight = (cgX + 1) < wg ? (int) ((sigRaster >> (cgPos + 1)) & 1) : 0;
      const int sigLower = (cgY + 1) < hg ? (int) ((sigRaster >> (cgPos + wg)) & 1) : 0;
      bb[o++] = (uint16_t) (((grpBase + (sigRight | sigLower)) << 1) | (int) ((sigGroups >> sub) & 1));
    }
    int ep = (coded && a) ? 1 : 0;                        // sign
    if (ctxMode) {
      if (sigCoded) bb[o++] = (uint16_t) (((VX_CTX_SigFlag[is_chroma + 2 * imax(0, qstate - 1)] + sigofs) << 1) | (a != 0));
      if (a) {
        bb[o++] = (uint16_t) (((g1Base + goff) << 1) | (a > 1));
        if (a > 1) {
          bb[o++] = (uint16_t) (((parBase + goff) << 1) | ((a - 2) & 1));
          bb[o++] = (uint16_t) (((g2Base + goff) << 1) | (((a - 2) >> 1) != 0));
          if (a >= 4) ep += rem_abs_len((unsigned) (a - 4) >> 1, L.t.gorice_pars[imax(imin(sumPlain - 20, 31), 0)]);
        }
      }
    } else if (coded) {
      const int sumAll = imin(sumPlain, 31);
      const unsigned rice = L.t.gorice_pars[sumAll], pos0 = L.t.gorice_pos0[imax(0, qstate - 1) * 32 + sumAll];
      ep += rem_abs_len(a == 0 ? pos0 : (unsigned) a <= pos0 ? (unsigned) a - 1 : (unsigned) a, rice);
    }
    mybits += (unsigned long long) ep << 15;
    regBins -= used; nb += tot;
    wave_sync();
    const long long e1 = STAMP();
    rc_chain(cb.ci, wv, nb, lane, mybits);
    nb = 0;
    qe += e1 - e0; qc += STAMP() - e1;
  }
  const long long q3 = STAMP();
  const unsigned long long tot = wave_sum_u64(mybits);
  if (lane == 0) cb.bits += tot;
  if (VVCX_STAMP && VTX == 0) { PROF(32) += (unsigned long long) (q1 - q0); PROF(34) += (unsigned long long) qe; PROF(35) += (unsigned long long) qc; PROF(36) += (unsigned long long) (STAMP() - q3); PROF(37) += 1; }
}

#include "vvcx_depquant_dev.h"
#include "vvcx_lfnst_dev.h"

// ------------------------------------------------------------------------------------------------ partitioner (thread 0)
__device__ int implicit_split(const VxParams &p, Frame &f, int ch)      // CL/UnitPartitioner.cpp:530-581
{
  if (f.impl_checked) return f.impl_split;
  int split = SPLIT_NONE;
  const int blIn = f.x < p.pic_w && (f.y + f.h - 1) < p.pic_h;
  const int trIn = (f.x + f.w - 1) < p.pic_w && f.y < p.pic_h;
  const int maxBt = p.max_bt_size[ch], minQt = p.min_qt[ch];
  const int btAllowed = f.w <= maxBt && f.h <= maxBt;
  const int qtAllowed = f.w > minQt && f.h > minQt && f.bt == 0;
  if (!blIn && !trIn && qtAllowed) split = SPLIT_QT;
  else if (!blIn && btAllowed) split = SPLIT_BH;
  else if (!trIn && btAllowed) split = SPLIT_BV;
  else if (!blIn || !trIn) split = SPLIT_QT;
  if (f.w > 64 || f.h > 64) split = SPLIT_QT;                // dual tree (566-569)
  f.impl_checked = 1; f.impl_split = (uint8_t) split;
  return split;
}
__device__ __noinline__ void can_split(const VxParams &p_, int d_, int ch, int can[6])   // CL/UnitPartitioner.cpp:379-466
{
  const VxParams &p = L.par; (void) p_;
  Frame &f = L.fr[d_];                  // frames are addressed by level so that the accesses are LDS instructions, not generic-address ones
  const int impl = implicit_split(p, f, ch);
  const int maxBTD = p.max_bt_depth[ch] + f.impl_bt;
  const int maxBt = p.max_bt_size[ch], minBt = 4, maxTt = p.max_tt_size[ch], minTt = 4, minQt = p.min_qt[ch];
  const int cw = f.w >> 1, chh = f.h >> 1;
  for (int i = 0; i < 6; i++) can[i] = 1;
  int canBtt = f.mt < maxBTD;
  const int last = f.last_split;
  const int parl = last == SPLIT_TH ? SPLIT_BH : SPLIT_BV;
  if (last != 0 && last != SPLIT_QT) can[1] = 0;
  if (f.w <= minQt) can[1] = 0;
  if (ch == 1 && cw <= 4) can[1] = 0;
  if (impl != SPLIT_NONE) { can[0] = can[4] = can[5] = 0; can[2] = impl == SPLIT_BH; can[3] = impl == SPLIT_BV; return; }
  if ((last == SPLIT_TH || last == SPLIT_TV) && f.part_idx == 1) { can[2] = parl != SPLIT_BH; can[3] = parl != SPLIT_BV; }
  if (canBtt && (f.w <= minBt && f.h <= minBt) && (f.w <= minTt && f.h <= minTt)) canBtt = 0;
  if (canBtt && (f.w > maxBt || f.h > maxBt) && (f.w > maxTt || f.h > maxTt)) canBtt = 0;
  if (!canBtt) { can[2] = can[3] = can[4] = can[5] = 0; return; }
  if (f.w > maxBt || f.h > maxBt) can[2] = can[3] = 0;
  if (f.h <= minBt) can[2] = 0;
  if (f.w > 64 && f.h <= 64) can[2] = 0;
  if (ch == 1 && cw * chh <= 16) can[2] = 0;
  if (f.w <= minBt) can[3] = 0;
  if (f.w <= 64 && f.h > 64) can[3] = 0;
  if (ch == 1 && cw * chh <= 16) can[3] = 0;
  if (f.h <= 2 * minTt || f.h > maxTt || f.w > maxTt) can[4] = 0;
  if (f.w > 64 || f.h > 64) can[4] = 0;
  if (ch == 1 && cw * chh <= 32) can[4] = 0;
  if (f.w <= 2 * minTt || f.w > maxTt || f.h > maxTt) can[5] = 0;
  if (f.w > 64 || f.h > 64) can[5] = 0;
  if (ch == 1 && cw * chh <= 32) can[5] = 0;
}
__device__ inline int can_do(const VxParams &p, Frame &f, int ch, int split) { return (f.can_mask >> split) & 1; }

// neighbour CU lookup (cs.getCU / getCURestricted → "coded in the current path and same tile")
__device__ const VxUnit *get_cu(const VxParams &p, const VxFrameDev &fd, int ch, int px, int py, int tile)
{
  const int W = ch ? p.pic_w >> 1 : p.pic_w, H = ch ? p.pic_h >> 1 : p.pic_h, ul = ch ? 1 : 2;
  if (px < 0 || py < 0 || px >= W || py >= H) return nullptr;
  const VxUnit *u = &fd.units[ch][(py >> ul) * p.uw + (px >> ul)];
  return u->tag == (uint16_t) (tile + 1) ? u : nullptr;
}
// Everything about a node that is fixed while it is processed — its left / above neighbour CUs, the canSplit() result
// (CL/UnitPartitioner.cpp:379-466) and the split-flag context increments of DeriveCtx::CtxSplit
// (CL/ContextModelling.cpp:154-250) — is derived once when the node is entered.
__device__ __noinline__ void prepare_node(const VxParams &p_, const VxFrameDev &fd_, int d_, int ch, int tile)
{
  const VxParams &p = L.par; (void) p_; const VxFrameDev &fd = L.fdv; (void) fd_;
  Frame &f = L.fr[d_];
  const int sh = ch ? 1 : 0;
  // left and above CU records fetched together (one HBM/L2 round trip instead of dependent ones); x - 1 / y - 1 of an
  // in-picture node can only fall off the left / top edge
  const int ul = ch ? 1 : 2, xL = (f.x >> sh) - 1, yL = f.y >> sh, xA = f.x >> sh, yA = (f.y >> sh) - 1;
  const VxUnit uL = fd.units[ch][(yL >> ul) * p.uw + (imax(xL, 0) >> ul)], uA = fd.units[ch][(imax(yA, 0) >> ul) * p.uw + (xA >> ul)];
  const bool cuL = xL >= 0 && uL.tag == (uint16_t) (tile + 1), cuA = yA >= 0 && uA.tag == (uint16_t) (tile + 1);
  f.nb_ok = (uint8_t) ((cuL ? 1 : 0) | (cuA ? 2 : 0));
  int lh = 0, lq = 0, aw = 0, aq = 0;
  if (cuL) { lh = uL.lh; lq = uL.qt; }
  if (cuA) { aw = uA.lw; aq = uA.qt; }
  f.nbL_lh = (uint8_t) lh; f.nbL_qt = (uint8_t) lq; f.nbA_lw = (uint8_t) aw; f.nbA_qt = (uint8_t) aq;
  f.impl_checked = 0;
  int can[6]; can_split(p, d_, ch, can);
  int mask = 0;
  for (int i = 0; i < 6; i++) mask |= can[i] << i;
  f.can_mask = (uint8_t) mask;
  const int bw = f.w >> sh, bh = f.h >> sh;
  unsigned ctxSpl = 0;
  if (cuL) ctxSpl += ((1 << lh) < bh) ? 1 : 0;
  if (cuA) ctxSpl += ((1 << aw) < bw) ? 1 : 0;
  unsigned numSplit = 0;
  if (can[1]) numSplit += 2;
  for (int i = 2; i < 6; i++) if (can[i]) numSplit += 1;
  if (numSplit > 0) numSplit--;
  ctxSpl += 3 * (numSplit >> 1);
  unsigned ctxQt = (cuL && lq > f.qt) ? 1 : 0;
  ctxQt += (cuA && aq > f.qt) ? 1 : 0;
  ctxQt += f.qt < 2 ? 0 : 3;
  unsigned ctxHv = 0;
  const unsigned numHor = (unsigned) (can[2] + can[4]), numVer = (unsigned) (can[3] + can[5]);
  if (numVer == numHor) {
    const unsigned wAbove = cuA ? (1u << aw) : 1, hLeft = cuL ? (1u << lh) : 1;
    const unsigned depAbove = (unsigned) bw / wAbove, depLeft = (unsigned) bh / hLeft;
    if (depAbove == depLeft || !cuL || !cuA) ctxHv = 0; else if (depAbove < depLeft) ctxHv = 1; else ctxHv = 2;
  } else if (numVer < numHor) ctxHv = 3; else ctxHv = 4;
  f.ctx_spl = (uint8_t) ctxSpl; f.ctx_qt = (uint8_t) ctxQt; f.ctx_hv = (uint8_t) ctxHv;
}
// CABACWriter::split_cu_mode (EL/CABACWriter.cpp:1010-1069)
template <bool WR = false>
__device__ void enc_split_cu_mode(const VxParams &p, const VxFrameDev &fd, Cab &cb, Frame &f, int ch, int tile, int split)
{
  const int m = f.can_mask;
  const int canNo = m & 1, canQt = (m >> 1) & 1, canBh = (m >> 2) & 1, canBv = (m >> 3) & 1, canTh = (m >> 4) & 1, canTv = (m >> 5) & 1;
  const unsigned ctxH12 = f.mt <= 1 ? 1 : 0, ctxV12 = f.mt <= 1 ? 3 : 2;
  const int canSplit = canQt || canBh || canBv || canTh || canTv;
  const int isNo = split == SPLIT_NONE;
  if (canNo && canSplit) enc_bin<WR>(cb, !isNo, VX_CTX_SplitFlag + f.ctx_spl);
  if (isNo) return;
  const int canBtt = canBh || canBv || canTh || canTv;
  const int isQt = split == SPLIT_QT;
  if (canQt && canBtt) enc_bin<WR>(cb, (unsigned) isQt, VX_CTX_SplitQtFlag + f.ctx_qt);
  if (isQt) return;
  const int canHor = canBh || canTh, canVer = canBv || canTv;
  const int isVer = split == SPLIT_BV || split == SPLIT_TV;
  if (canVer && canHor) enc_bin<WR>(cb, (unsigned) isVer, VX_CTX_SplitHvFlag + f.ctx_hv);
  const int can14 = isVer ? canTv : canTh, can12 = isVer ? canBv : canBh;
  const int is12 = isVer ? (split == SPLIT_BV) : (split == SPLIT_BH);
  if (can12 && can14) enc_bin<WR>(cb, (unsigned) is12, VX_CTX_Split12Flag + (int) (isVer ? ctxV12 : ctxH12));
}

// PU::getIntraMPMs (CL/UnitTools.cpp:508-640)
// luma mode a unit shows to its neighbours' MPM lists and to the chroma DM (PU::getIntraDirLuma, CL/UnitTools.cpp:786-800): PLANAR for MIP
__device__ inline int unit_ldir(const VxUnit &u) { return (u.mrl & MIPF) ? PLANAR : u.dir; }
// neighbours of a luma node: the modes for its MPM list (PU::getIntraMPMs, CL/UnitTools.cpp:516-532) and, with MIP on, the number of MIP modes
// (getNumModesMip 4688-4708; 0 = no mip_flag, also above MaxTbSize) and the context of its mip_flag (DeriveCtx::CtxMipFlag, CL/ContextModelling.cpp:555-569)
__device__ __noinline__ void luma_neighbours(const VxParams &p, const VxFrameDev &fd, int x, int y, int w, int h, int tile, int &Ld, int &Ad)
{
  Ld = PLANAR; Ad = PLANAR;
  const VxUnit *uL = get_cu(p, fd, 0, x - 1, y + h - 1, tile); if (uL) Ld = unit_ldir(*uL);
  const VxUnit *uA = get_cu(p, fd, 0, x + w - 1, y - 1, tile); if (uA && ((y - 1) >> 7) == (y >> 7)) Ad = unit_ldir(*uA);
  L.mip_n = 0; L.mip_ctx = 0;
  L.isp_ok = (int16_t) ((p.tools & TOOL_ISP) && ilog2i(w) + ilog2i(h) > 4 && w <= 64 && h <= 64);      // CU::canUseISP (CL/UnitTools.cpp:414-435)
  if ((p.tools & TOOL_MIP) && w <= 64 && h <= 64) {
    L.mip_n = (int16_t) mip_num_modes(w, h);
    const VxUnit *a = get_cu(p, fd, 0, x - 1, y, tile), *b = get_cu(p, fd, 0, x, y - 1, tile);
    L.mip_ctx = (int16_t) ((w > 2 * h || h > 2 * w) ? 3 : ((a && (a->mrl & MIPF)) ? 1 : 0) + ((b && (b->mrl & MIPF)) ? 1 : 0));
  }
}
__device__ void derive_mpms(int Ld, int Ad, unsigned mpm[6])
{
  const int offset = 61, mod = 64;
  mpm[0] = PLANAR; mpm[1] = DC; mpm[2] = VER; mpm[3] = HOR; mpm[4] = VER - 4; mpm[5] = VER + 4;
  if (Ld == Ad) {
    if (Ld > DC) { mpm[0] = PLANAR; mpm[1] = Ld; mpm[2] = ((Ld + offset) % mod) + 2; mpm[3] = ((Ld - 1) % mod) + 2; mpm[4] = ((Ld + offset - 1) % mod) + 2; mpm[5] = (Ld % mod) + 2; }
  } else if (Ld > DC && Ad > DC) {
    mpm[0] = PLANAR; mpm[1] = Ld; mpm[2] = Ad;
    const int mx = mpm[1] > mpm[2] ? 1 : 2, mn = mpm[1] > mpm[2] ? 2 : 1;
    const int d = (int) mpm[mx] - (int) mpm[mn];
    if (d == 1) { mpm[3] = ((mpm[mn] + offset) % mod) + 2; mpm[4] = ((mpm[mx] - 1) % mod) + 2; mpm[5] = ((mpm[mn] + offset - 1) % mod) + 2; }
    else if (d >= 62) { mpm[3] = ((mpm[mn] - 1) % mod) + 2; mpm[4] = ((mpm[mx] + offset) % mod) + 2; mpm[5] = (mpm[mn] % mod) + 2; }
    else if (d == 2) { mpm[3] = ((mpm[mn] - 1) % mod) + 2; mpm[4] = ((mpm[mn] + offset) % mod) + 2; mpm[5] = ((mpm[mx] - 1) % mod) + 2; }
    else { mpm[3] = ((mpm[mn] + offset) % mod) + 2; mpm[4] = ((mpm[mn] - 1) % mod) + 2; mpm[5] = ((mpm[mx] + offset) % mod) + 2; }
  } else if (Ld + Ad >= 2) {
    mpm[0] = PLANAR; mpm[1] = (unsigned) (Ld < Ad ? Ad : Ld);
    mpm[2] = ((mpm[1] + offset) % mod) + 2; mpm[3] = ((mpm[1] - 1) % mod) + 2; mpm[4] = ((mpm[1] + offset - 1) % mod) + 2; mpm[5] = (mpm[1] % mod) + 2;
  }
  // ascending copy for the non-MPM rank (std::sort in intra_luma_pred_mode, EL/CABACWriter.cpp:1833)
  for (int i = 0; i < 6; i++) L.mpm_sorted[i] = mpm[i];
  for (int i = 1; i < 6; i++) { unsigned v = L.mpm_sorted[i]; int j = i - 1; while (j >= 0 && L.mpm_sorted[j] > v) { L.mpm_sorted[j + 1] = L.mpm_sorted[j]; j--; } L.mpm_sorted[j + 1] = v; }
}
// CABACWriter::intra_luma_pred_mode 1762-1845 + extend_ref_line 1566-1591 (ISP off); MPMs from L.mpm; mip_flag 4741-4767 with the node's
// context (DeriveCtx::CtxMipFlag) and mip_pred_mode 4781-4787 = xWriteTruncBinCode(mode, numModes) from L.mip_n / L.mip_ctx
// isp: cu.ispMode (0 none, 1 horizontal, 2 vertical split); isp_mode (3944-3965) follows the reference line index of every luma CU that could use ISP
template <bool WR = false>
__device__ void enc_intra_luma_pred_mode(Cab &cb, int y, int dir, int mrl, int isp = 0)
{
  if (L.mip_n) {
    enc_bin<WR>(cb, (unsigned) (mrl >> 7), VX_CTX_MipFlag + L.mip_ctx);
    if (mrl & MIPF) {
      const int n = L.mip_n, th = ilog2i(n), val = 1 << th, b = n - val;
      if (dir < val - b) enc_ep<WR>(cb, (unsigned) dir, th); else enc_ep<WR>(cb, (unsigned) (dir + val - b), th + 1);
      return;
    }
  }
  if ((y & 127) != 0) {
    enc_bin<WR>(cb, mrl != 0, VX_CTX_MultiRefLineIdx + 0);
    if (mrl != 0) enc_bin<WR>(cb, mrl != 1, VX_CTX_MultiRefLineIdx + 1);
  }
  if (!mrl && L.isp_ok) { enc_bin<WR>(cb, isp != 0, VX_CTX_ISPMode + 0); if (isp) enc_bin<WR>(cb, (unsigned) (isp - 1), VX_CTX_ISPMode + 1); }
  int mpm_idx = 6;
  for (int i = 0; i < 6; i++) if ((unsigned) dir == L.mpm[i]) { mpm_idx = i; break; }
  if (!mrl) enc_bin<WR>(cb, mpm_idx < 6, VX_CTX_IntraLumaMpmFlag);
  if (mpm_idx < 6) {
    if (mrl == 0) enc_bin<WR>(cb, mpm_idx > 0, VX_CTX_IntraLumaPlanarFlag + (isp ? 0 : 1));      // 1812: context by cu.ispMode
    if (mpm_idx) { const int nb = imin(mpm_idx, 4); enc_ep<WR>(cb, mpm_idx < 5 ? (1u << nb) - 2 : 15u, nb); }      // truncated unary, bypass (1791-1806)
  } else {
    unsigned m = (unsigned) dir;
    for (int i = 5; i >= 0; i--) if (m > L.mpm_sorted[i]) m--;
    if (m < 3) enc_ep<WR>(cb, m, 5); else enc_ep<WR>(cb, m + 3, 6);       // xWriteTruncBinCode(m, 61) 1528-1566
  }
}
// xFracModeBitsIntra (EL/IntraSearch.cpp:4263-4288) from the node's start contexts.  Every context-coded bin of
// intra_luma_pred_mode uses a different context, so the bits are a pure function of the start states.
__device__ inline unsigned frac_bits_of(const Ctx &c, int ctx, unsigned bin)
{ return BIN_FRAC((unsigned) (c.s0[ctx] + c.s1[ctx]) >> 8, bin); }
__device__ unsigned long long luma_mode_bits(const Ctx &c, int y, int dir, int mrl)
{
  unsigned long long bits = 0;
  if (L.mip_n) {
    bits += frac_bits_of(c, VX_CTX_MipFlag + L.mip_ctx, (unsigned) (mrl >> 7));
    if (mrl & MIPF) { const int n = L.mip_n, th = ilog2i(n), val = 1 << th, b = n - val; return bits + ((unsigned long long) (dir < val - b ? th : th + 1) << 15); }
  }
  if ((y & 127) != 0) {
    bits += frac_bits_of(c, VX_CTX_MultiRefLineIdx, mrl != 0);
    if (mrl != 0) bits += frac_bits_of(c, VX_CTX_MultiRefLineIdx + 1, mrl != 1);
  }
  if (!mrl && L.isp_ok) bits += frac_bits_of(c, VX_CTX_ISPMode + 0, 0);
  int mpm_idx = 6;
  for (int i = 0; i < 6; i++) if ((unsigned) dir == L.mpm[i]) { mpm_idx = i; break; }
  if (!mrl) bits += frac_bits_of(c, VX_CTX_IntraLumaMpmFlag, mpm_idx < 6);
  if (mpm_idx < 6) {
    if (mrl == 0) bits += frac_bits_of(c, VX_CTX_IntraLumaPlanarFlag + 1, mpm_idx > 0);
    bits += (unsigned long long) imin(mpm_idx, 4) << 15;
  } else {
    unsigned m = (unsigned) dir;
    for (int i = 5; i >= 0; i--) if (m > L.mpm_sorted[i]) m--;
    bits += (unsigned long long) (m < 3 ? 5 : 6) << 15;
  }
  return bits;
}
// CABACWriter::intra_chroma_pred_mode 1891-1933 + intra_chroma_lmc_mode 1864-1888; lm = co-located luma mode (candidate list
// CL/UnitTools.cpp:840-873), lm_ok = CodingUnit::checkCCLMAllowed
template <bool WR = false>
__device__ void enc_intra_chroma_pred_mode(Cab &cb, int dir, int lm, int lm_ok)
{
  if (lm_ok) {
    const int isLM = dir >= LM_CHROMA && dir <= MDLM_T;
    enc_bin<WR>(cb, (unsigned) isLM, VX_CTX_CclmModeFlag);
    if (isLM) {
      const int symbol = dir - LM_CHROMA;
      enc_bin<WR>(cb, symbol == 0 ? 0 : 1, VX_CTX_IntraChromaPredMode);
      if (symbol > 0) enc_ep<WR>(cb, (uint32_t) (symbol - 1), 1);
      return;
    }
  }
  const int isDM = dir == DM_CHROMA;
  enc_bin<WR>(cb, isDM ? 0 : 1, VX_CTX_IntraChromaPredMode);
  if (isDM) return;
  int list[4] = { PLANAR, VER, HOR, DC };
  for (int i = 0; i < 4; i++) if (lm == list[i]) { list[i] = VDIA; break; }
  int cand = 0;
  for (; cand < 4; cand++) if (list[cand] == dir) break;      // 4 for a cached LM mode reused where CCLM is not allowed: the reference codes that value too
  enc_ep<WR>(cb, (uint32_t) cand, 2);
}

// ------------------------------------------------------------------------------------------------ intra prediction
struct Ipa { int pred_mode, is_ver, mrl, ref_filter, interp, pdpc, angle, inv_angle, ang_scale; };
static __device__ const int16_t ANG_TABLE[32] = { 0, 1, 2, 3, 4, 6, 8, 10, 12, 14, 16, 18, 20, 23, 26, 29, 32, 35, 39, 45, 51, 57, 64, 73, 86, 102, 128, 171, 256, 341, 512, 1024 };
static __device__ const int16_t INV_ANG_TABLE[32] = { 0, 16384, 8192, 5461, 4096, 2731, 2048, 1638, 1365, 1170, 1024, 910, 819, 712, 630, 565,
  512, 468, 420, 364, 321, 287, 256, 224, 191, 161, 128, 96, 64, 48, 32, 16 };
static __device__ const uint8_t INTRA_FILTER_THR[8] = { 24, 24, 24, 14, 2, 0, 0, 0 };
static __device__ const uint8_t LAST_PREFIX_CTX[8] = { 0, 0, 0, 3, 6, 10, 15, 21 };     // CL/ContextModelling.cpp:106
static __device__ const uint8_t MODE_SHIFT[8] = { 0, 6, 10, 12, 14, 15, 0, 0 };          // CL/IntraPrediction.cpp:291
static __device__ const int8_t GAUSS_FILTER[32][4] = {     // g_intraGaussFilter (spec table), CL/IntraPrediction.cpp:76
  {16,32,16,0},{15,29,17,3},{15,29,17,3},{14,29,18,3},{13,29,18,4},{13,28,19,4},{13,28,19,4},{12,28,20,4},
  {11,28,20,5},{11,27,21,5},{10,27,22,5},{9,27,22,6},{9,26,23,6},{9,26,23,6},{8,25,24,7},{8,25,24,7},
  {8,24,24,8},{7,24,25,8},{7,24,25,8},{6,23,26,9},{6,23,26,9},{6,22,27,9},{5,22,27,10},{5,21,27,11},
  {5,20,28,11},{4,20,28,12},{4,19,28,13},{4,19,28,13},{4,18,29,13},{3,18,29,14},{3,17,29,15},{3,17,29,15} };

// getWideAngle 287-303 + initPredIntraParams 487-618
__device__ void init_pred_params(int w, int h, int is_luma, int mode, int mrl, Ipa &p)
{
  int pm = mode;
  if (pm > DC && pm <= VDIA) {
    const int ms = L.t.mode_shift[iabs(ilog2i(w) - ilog2i(h))];
    if (w > h && pm < 2 + ms) pm += VDIA - 1;
    else if (h > w && pm > VDIA - ms) pm -= VDIA - 1;
  }
  p.pred_mode = pm; p.is_ver = pm >= DIA; p.mrl = is_luma ? mrl : 0; p.ref_filter = 0; p.interp = 0;
  p.pdpc = ((w >= 4 && h >= 4) || !is_luma) && p.mrl == 0;
  p.angle = 0; p.inv_angle = 0; p.ang_scale = -1;
  const int am = p.is_ver ? pm - VER : -(pm - HOR);
  int absAng = 0;
  if (mode > DC && mode < 67) {
    const int a = iabs(am);
    absAng = L.t.ang[a]; p.inv_angle = L.t.inv_ang[a]; p.angle = am < 0 ? -absAng : absAng;
    if (am < 0) p.pdpc = 0;
    else if (am > 0) {
      const int side = p.is_ver ? h : w;
      int sc = ilog2i(side) - (ilog2i(3 * p.inv_angle - 2) - 8);
      if (sc > 2) sc = 2;
      p.ang_scale = sc; p.pdpc &= sc >= 0;
    }
  }
  if (!is_luma || p.mrl || mode == DC) {}
  else if (mode == PLANAR) p.ref_filter = w * h > 32;
  else {
    const int d1 = iabs(pm - HOR), d2 = iabs(pm - VER);
    const int diff = d1 < d2 ? d1 : d2;
    const int log2Size = (ilog2i(w) + ilog2i(h)) >> 1;
    if (diff > L.t.intra_thr[log2Size]) { const int is_int = (absAng & 0x1F) == 0; p.ref_filter = is_int; p.interp = !is_int; }
  }
}
// initPredIntraParams for a prediction region (w x h) of an ISP CU (cuw x cuh): wide-angle mapping by the CU's shape, no reference smoothing, the cubic interpolation
// filter, PDPC by the region's size (487-618 with useISP, JVET_O0502)
__device__ void init_pred_params_isp(int cuw, int cuh, int w, int h, int mode, Ipa &p)
{
  init_pred_params(cuw, cuh, 1, mode, 0, p);
  p.ref_filter = 0; p.interp = 0; p.pdpc = w >= 4 && h >= 4; p.ang_scale = -1;
  if (mode > DC && mode < 67) {
    const int am = p.is_ver ? p.pred_mode - VER : -(p.pred_mode - HOR);
    if (am < 0) p.pdpc = 0;
    else if (am > 0) {
      const int side = p.is_ver ? h : w;
      int sc = ilog2i(side) - (ilog2i(3 * p.inv_angle - 2) - 8);
      if (sc > 2) sc = 2;
      p.ang_scale = sc; p.pdpc &= sc >= 0;
    }
  }
}
__device__ inline uint2 ipa_pack(const Ipa &p)
{
  uint2 r;
  // pred_mode can be negative after the wide-angle remap: mask it
  r.x = ((unsigned) p.pred_mode & 255u) | ((unsigned) p.is_ver << 8) | ((unsigned) p.mrl << 9) | ((unsigned) p.ref_filter << 11) | ((unsigned) p.interp << 12) | ((unsigned) p.pdpc << 13) | ((unsigned) (p.ang_scale + 1) << 14);
  r.y = ((unsigned) p.angle & 0xffffu) | ((unsigned) p.inv_angle << 16);
  return r;
}
__device__ inline void ipa_unpack(uint2 r, Ipa &p)
{
  p.pred_mode = (int) (r.x & 255); p.is_ver = (int) ((r.x >> 8) & 1); p.mrl = (int) ((r.x >> 9) & 3); p.ref_filter = (int) ((r.x >> 11) & 1); p.interp = (int) ((r.x >> 12) & 1);
  p.pdpc = (int) ((r.x >> 13) & 1); p.ang_scale = (int) ((r.x >> 14) & 7) - 1; p.angle = (int) (int16_t) (r.y & 0xffffu); p.inv_angle = (int) (r.y >> 16);
}
__device__ inline int clip_bd(int v, int bd) { const int mx = (1 << bd) - 1; return v < 0 ? 0 : v > mx ? mx : v; }

// one predicted sample.  top[i] = pSrc.at(i,0), left[i] = pSrc.at(0,i) of the (un)filtered reference
// buffer for this mrl; closed forms of xPredIntraPlanar 426-479, DC 248-285, xPredIntraAng 633-935 and
// the planar/DC PDPC 354-378.
// ref_top / ref_left: m_topRefLength / m_leftRefLength when they are not twice the block's sides (prediction regions of ISP CUs: CU side + region side)
__device__ int pred_sample(const int16_t *top, const int16_t *left, int w, int h, int px, int py, const Ipa &ip,
                           int mode, int is_luma, int bd, int dcv, int ref_top = 0, int ref_left = 0)
{
  const int mrl = ip.mrl;
  int v;
  if (mode == PLANAR) {
    const int l2w = ilog2i(imax(w, 2)), l2h = ilog2i(imax(h, 2));      // 430-431: one-sample sides of ISP sub-partitions weigh like two
    const int lft = left[py + 1], tp = top[px + 1];
    const int hor = (lft << l2w) + (px + 1) * (top[w + 1] - lft);
    const int ver = (tp << l2h) + (py + 1) * (left[h + 1] - tp);
    v = ((hor << l2h) + (ver << l2w) + (1 << (l2w + l2h))) >> (1 + l2w + l2h);
  } else if (mode == DC) {
    v = dcv;
  } else {
    const int ver = ip.is_ver, ang = ip.angle, inv = ip.inv_angle;
    const int W = ver ? w : h, H = ver ? h : w;
    const int xx = ver ? px : py, yy = ver ? py : px;
    const int16_t *mainp = ver ? top : left, *sidep = ver ? left : top;
    const int sizeSide = H;
    const int refLen = ref_top ? (ver ? ref_top : ref_left) : 2 * W;
#define MAINR(i_) ({ int k_ = (i_) + mrl; int r_; if (k_ >= 0) { if (ang >= 0 && k_ > refLen + mrl) k_ = refLen + mrl; r_ = mainp[k_]; } \
                     else { int j_ = (-k_ * inv + 256) >> 9; if (j_ > sizeSide) j_ = sizeSide; r_ = sidep[j_]; } r_; })
#define SIDER(i_) ((int) sidep[(i_) + mrl])
    if (ang == 0) {
      v = MAINR(xx + 1);
      if (ip.pdpc) {
        const int scale = (ilog2i(W) + ilog2i(H) - 2) >> 2;
        if (xx < imin(3 << scale, W)) {
          const int wL = 32 >> (2 * xx >> scale);
          v = clip_bd(v + ((wL * (SIDER(1 + yy) - MAINR(0)) + 32) >> 6), bd);
        }
      }
    } else {
      const int deltaPos = ang * (yy + 1 + mrl);
      const int di = deltaPos >> 5, df = deltaPos & 31;
      if ((iabs(ang) & 0x1F) != 0) {
        if (is_luma) {
          const int8_t *f = ip.interp ? &L.t.gauss[df * 4] : &L.t.cubic[df * 4];
          const int s = f[0] * MAINR(di + xx) + f[1] * MAINR(di + xx + 1) + f[2] * MAINR(di + xx + 2) + f[3] * MAINR(di + xx + 3);
          v = clip_bd((int) (int16_t) ((s + 32) >> 6), bd);
        } else {
          const int p0 = MAINR(di + xx + 1), p1 = MAINR(di + xx + 2);
          v = p0 + ((df * (p1 - p0) + 16) >> 5);
        }
      } else v = MAINR(xx + di + 1);
      if (ip.pdpc) {
        const int scale = ip.ang_scale;
        if (xx < imin(3 << scale, W)) {
          const int invSum = 256 + (xx + 1) * inv;
          const int wL = 32 >> (2 * xx >> scale);
          const int lft = SIDER(yy + (invSum >> 9) + 1);
          v = v + ((wL * (lft - v) + 32) >> 6);
        }
      }
    }
#undef MAINR
#undef SIDER
  }
  if (ip.pdpc && (mode == PLANAR || mode == DC)) {
    const int scale = (ilog2i(w) - 2 + ilog2i(h) - 2 + 2) >> 2;
    const int wT = 32 >> imin(31, (py << 1) >> scale), wL = 32 >> imin(31, (px << 1) >> scale);
    v = v + ((wL * (left[py + 1] - v) + wT * (top[px + 1] - v) + 32) >> 6);
  }
  return (int) (int16_t) v;
}

// DC value of a reference set (xGetPredValDc 248-285): executed by one thread
__device__ int dc_value(const int16_t *top, const int16_t *left, int w, int h, int mrl)
{
  int sum = 0;
  const int denom = (w == h) ? (w << 1) : imax(w, h);
  if (w >= h) for (int i = 0; i < w; i++) sum += top[mrl + 1 + i];
  if (w <= h) for (int i = 0; i < h; i++) sum += left[mrl + 1 + i];
  return (sum + (denom >> 1)) >> ilog2i(denom);
}

// ------------------------------------------------------------------------------------------------ reference samples (all threads)
// xFillReferenceSamples 1215-1468 in parallel form: every reference sample either copies its own picture
// sample (its unit is available) or the sample the sequential padding pass would have propagated to it:
// the last sample of the nearest preceding available unit in scan order (bottom-left → top-right), or the
// first sample of the first available unit when nothing precedes.  set layout: refs[set][0]=top, [1]=left.
template <typename T>
__device__ __noinline__ void build_refs(const VxParams &p_, const VxFrameDev &fd_, int comp, int x, int y, int w, int h, int tile, int nsets)
{
  const VxParams &p = L.par; (void) p_; const VxFrameDev &fd = L.fdv; (void) fd_;
  const int tid = VTX;
  comp = uni(comp); x = uni(x); y = uni(y); w = uni(w); h = uni(h); tile = uni(tile); nsets = uni(nsets);
  const int ch = comp ? 1 : 0;
  const int unit = ch ? 2 : 4, ul = ch ? 1 : 2;
  const int W = ch ? p.pic_w >> 1 : p.pic_w, H = ch ? p.pic_h >> 1 : p.pic_h;
  const int predW = 2 * w, predH = 2 * h;
  const int totalAbove = predW >> ul, totalLeft = predH >> ul, totalUnits = totalAbove + totalLeft + 1;
  const int numAbove = w >> ul, numLeft = h >> ul;
  if (tid < totalUnits) {
    int ux, uy;                         // a sample position inside the unit
    if (tid < totalLeft) { ux = x - 1; uy = y + (totalLeft - 1 - tid) * unit; }
    else if (tid == totalLeft) { ux = x - 1; uy = y - 1; }
    else { ux = x + (tid - totalLeft - 1) * unit; uy = y - 1; }
    int a = 0;
    if (ux >= 0 && uy >= 0 && ux < W && uy < H) a = fd.units[ch][(uy >> ul) * p.uw + (ux >> ul)].tag == (uint16_t) (tile + 1);
    // WaveFrontSynchro: getCURestricted hides the CTU column to the right (CL/CodingStructure.cpp:1634-1638); and the CTU row below, which the reference has not coded yet
    // (cu->idx <= curCu.idx, 1640) but whose stream may already have passed the CTU below-left of this one
    if ((p.tools & TOOL_WPP) && ((((ux << (ch ? 1 : 0)) >> 7) > (L.ctu_x >> 7)) || (((uy << (ch ? 1 : 0)) >> 7) > (L.ctu_y >> 7)))) a = 0;
    L.flags[tid] = (uint8_t) a;
  }
  __syncthreads();
  if (tid == 0) {
    // isAbove/Left/...Available stop at the first missing unit of their segment (1544-1662)
    int ok = 1; for (int i = 0; i < numAbove; i++) { ok &= L.flags[totalLeft + 1 + i]; L.flags[totalLeft + 1 + i] = (uint8_t) ok; }
    ok = 1; for (int i = numAbove; i < totalAbove; i++) { ok &= L.flags[totalLeft + 1 + i]; L.flags[totalLeft + 1 + i] = (uint8_t) ok; }
    ok = 1; for (int i = 0; i < numLeft; i++) { ok &= L.flags[totalLeft - 1 - i]; L.flags[totalLeft - 1 - i] = (uint8_t) ok; }
    ok = 1; for (int i = numLeft; i < totalLeft; i++) { ok &= L.flags[totalLeft - 1 - i]; L.flags[totalLeft - 1 - i] = (uint8_t) ok; }
    int firstAvail = -1;
    for (int u = 0; u < totalUnits; u++) if (L.flags[u]) { firstAvail = u; break; }
    int prev = -1;
    for (int u = 0; u < totalUnits; u++) {
      if (L.flags[u]) { prev = u; L.src_unit[u] = (int8_t) u; }
      else L.src_unit[u] = (int8_t) (prev >= 0 ? prev : (firstAvail >= 0 ? -2 - firstAvail : -1));   // -1: nothing available
    }
  }
  __syncthreads();
  const void *rec = fd.rec[comp]; const int st = fd.stride[comp];
  const int dcv = 1 << (p.bit_depth - 1);
  for (int s = 0; s < nsets; s++) {
    const int mrl = comp ? 0 : (s == 0 ? 0 : s == 1 ? 1 : 3);
    const int set = comp ? (comp - 1) : (s == 0 ? 0 : s + 1);
    const int nLeft = predH + mrl, nTop = predW + mrl;              // left[1..nLeft], top[1..nTop], corner = [0]
    for (int i = tid; i < nLeft + nTop + 1; i += NT) {
      int isTop, idx;                    // which array entry
      if (i < nLeft) { isTop = 0; idx = nLeft - i; } else if (i == nLeft) { isTop = 0; idx = 0; } else { isTop = 1; idx = i - nLeft; }
      int u;
      if (idx <= mrl) u = totalLeft;
      else if (isTop) u = totalLeft + 1 + ((idx - 1 - mrl) >> ul);
      else u = totalLeft - 1 - ((idx - 1 - mrl) >> ul);
      int sx, sy, val;
      const int su = L.src_unit[u];
      if (su == -1) val = dcv;
      else {
        if (su == u) { if (isTop) { sx = x - 1 - mrl + idx; sy = y - 1 - mrl; } else { sx = x - 1 - mrl; sy = y - 1 - mrl + idx; } }
        else if (su >= 0) {              // last sample (scan order) of unit su
          if (su < totalLeft) { sx = x - 1 - mrl; sy = y + (totalLeft - 1 - su) * unit; }
          else if (su == totalLeft) { sx = x - 1; sy = y - 1 - mrl; }
          else { sx = x + (su - totalLeft - 1) * unit + unit - 1; sy = y - 1 - mrl; }
        } else {                         // first sample (scan order) of the first available unit
          const int fu = -2 - su;
          if (fu < totalLeft) { sx = x - 1 - mrl; sy = y + (totalLeft - 1 - fu) * unit + unit - 1; }
          else if (fu == totalLeft) { sx = x - 1 - mrl; sy = y - 1; }
          else { sx = x + (fu - totalLeft - 1) * unit; sy = y - 1 - mrl; }
        }
        val = ld_px<T>(rec, sy * st + sx);
      }
      if (idx == 0) { L.refs[set][0][0] = (int16_t) val; L.refs[set][1][0] = (int16_t) val; }
      else L.refs[set][isTop ? 0 : 1][idx] = (int16_t) val;
    }
  }
  __syncthreads();
  if (!comp) {
    // xFilterReferenceSamples 1470-1522 for mrl 0 → set 1
    const int nLeft = predH, nTop = predW;
    for (int i = tid; i < nLeft + nTop + 1; i += NT) {
      const int16_t *t = L.refs[0][0], *l = L.refs[0][1];
      if (i == 0) { const int v = (l[1] + 2 * t[0] + t[1] + 2) >> 2; L.refs[1][0][0] = (int16_t) v; L.refs[1][1][0] = (int16_t) v; }
      else if (i <= nTop) { const int j = i; L.refs[1][0][j] = (int16_t) (j == nTop ? t[j] : (t[j + 1] + 2 * t[j] + t[j - 1] + 2) >> 2); }
      else { const int j = i - nTop; L.refs[1][1][j] = (int16_t) (j == nLeft ? l[j] : (l[j + 1] + 2 * l[j] + (j == 1 ? t[0] : l[j - 1]) + 2) >> 2); }
    }
    __syncthreads();
  }
}

// ------------------------------------------------------------------------------------------------ CCLM (all threads unless noted)
// CodingUnit::checkCCLMAllowed (CL/Unit.cpp:375-449) for a chroma-tree CU of a dual-tree I slice with CTU 128: the splits of the 64x64
// chroma node (depths 1, 2 of the CU's split series) and of the co-located 64x64 luma node decide
__device__ int cclm_allowed(const VxParams &p, const VxFrameDev &fd, int x, int y, uint64_t ss, int depth)
{
  if (!(p.tools & TOOL_CCLM)) return 0;
  const int s1 = depth > 1 ? (int) ((ss >> 5) & 31) : SPLIT_NONE, s2 = depth > 2 ? (int) ((ss >> 10) & 31) : SPLIT_NONE;
  int allow = s1 == SPLIT_QT || (s1 == SPLIT_BH && s2 == SPLIT_BV) || s1 == SPLIT_NONE || (s1 == SPLIT_BH && s2 == SPLIT_NONE);
  if (allow) {
    const VxUnit u = fd.units[0][(y >> 2) * p.uw + (x >> 2)];        // colLumaCu at the CU's luma position
    if (u.lw < 6 || u.lh < 6) { const int l1 = u.depth > 1 ? (int) ((u.ss >> 5) & 31) : SPLIT_NONE; if (l1 != SPLIT_QT) allow = 0; }
  }
  return allow;
}
__device__ inline int16_t *lm_in_buf(uint8_t *scratch, int nrec) { return nrec <= BUF ? L.lm_in : (int16_t *) (scratch + VXD_OFF_LM); }
// xGetLMParameters 1931-2150 for component k (0 Cb, 1 Cr) and mode; one thread
__device__ void cclm_params(const int16_t *in, int cw, int chh, int k, int mode, int bd, int out[3])
{
  int leftAvail = L.lm_info[0], aboveAvail = L.lm_info[1], availLB = L.lm_info[2], availAR = L.lm_info[3];
  const int availAbove = aboveAvail ? cw >> 1 : 0, availLeft = leftAvail ? chh >> 1 : 0;
  const int16_t *top = L.refs[k][0], *left = L.refs[k][1];
  int actualTop = 0, actualLeft = 0;
  if (mode == MDLM_T) { leftAvail = 0; availAR = imin(availAR, chh >> 1); actualTop = 2 * (availAbove + availAR); }
  else if (mode == MDLM_L) { aboveAvail = 0; availLB = imin(availLB, cw >> 1); actualLeft = 2 * (availLeft + availLB); }
  else { actualTop = cw; actualLeft = chh; }
  const int aboveIs4 = leftAvail ? 0 : 1, leftIs4 = aboveAvail ? 0 : 1;
  const int startT = actualTop >> (2 + aboveIs4), stepT = imax(1, actualTop >> (1 + aboveIs4));
  const int startL = actualLeft >> (2 + leftIs4), stepL = imax(1, actualLeft >> (1 + leftIs4));
  int selL[4] = { 0, 0, 0, 0 }, selC[4] = { 0, 0, 0, 0 };
  int cntT = 0, cntL = 0;
  if (aboveAvail) { cntT = imin(actualTop, (1 + aboveIs4) << 1); for (int c = 0, pos = startT; c < cntT; pos += stepT, c++) { selL[c] = L.lm_top[pos]; selC[c] = top[1 + pos]; } }
  if (leftAvail) { cntL = imin(actualLeft, (1 + leftIs4) << 1); for (int c = 0, pos = startL; c < cntL; pos += stepL, c++) { selL[c + cntT] = L.lm_left[pos]; selC[c + cntT] = left[1 + pos]; } }
  if (cntL + cntT == 2) {
    selL[3] = selL[0]; selC[3] = selC[0]; selL[2] = selL[1]; selC[2] = selC[1];
    selL[0] = selL[1]; selC[0] = selC[1]; selL[1] = selL[3]; selC[1] = selC[3];
  }
  int mn0 = 0, mn1 = 2, mx0 = 1, mx1 = 3;            // minGrpIdx / maxGrpIdx with the reference's four compare-and-swap steps
  if (selL[mn0] > selL[mn1]) { const int t = mn0; mn0 = mn1; mn1 = t; }
  if (selL[mx0] > selL[mx1]) { const int t = mx0; mx0 = mx1; mx1 = t; }
  if (selL[mn0] > selL[mx1]) { int t = mn0; mn0 = mx0; mx0 = t; t = mn1; mn1 = mx1; mx1 = t; }
  if (selL[mn1] > selL[mx0]) { const int t = mn1; mn1 = mx0; mx0 = t; }
  const int minL = (selL[mn0] + selL[mn1] + 1) >> 1, minC = (selC[mn0] + selC[mn1] + 1) >> 1;
  const int maxL = (selL[mx0] + selL[mx1] + 1) >> 1, maxC = (selC[mx0] + selC[mx1] + 1) >> 1;
  int a, b, shift;
  if (leftAvail || aboveAvail) {
    const int diff = maxL - minL;
    if (diff > 0) {
      const int diffC = maxC - minC;
      int x = ilog2i(diff);
      const int normDiff = ((diff << 4) >> x) & 15;
      const int v = (int) ((0x0765544332211110ull >> (4 * (15 - normDiff))) & 15) | 8;      // DivSigTable {0,7,6,5,5,4,4,3,3,2,2,1,1,1,1,0}
      x += normDiff != 0;
      const int yy = diffC == 0 ? 0 : ilog2i(iabs(diffC)) + 1;
      const int add = (1 << yy) >> 1;
      a = (diffC * v + add) >> yy;
      shift = 3 + x - yy;
      if (shift < 1) { shift = 1; a = a == 0 ? 0 : a < 0 ? -15 : 15; }
      b = minC - ((a * minL) >> shift);
    } else { a = 0; b = minC; shift = 0; }
  } else { a = 0; b = 1 << (bd - 1); shift = 0; }
  out[0] = a; out[1] = b; out[2] = shift;
  (void) in;
}
// xGetLumaRecPixels 1665-1930 (4:2:0, sps_cclm_colocated_chroma_flag 0) + the parameters of the three LM modes for both components.
// Call after build_refs of the chroma node: L.flags then hold the availability (stopping at the first missing unit of each segment)
// of its reference units, which are the units CCLM asks about.  One buffer serves LM and MDLM: the extension only adds entries.
template <typename T>
__device__ void cclm_prepare(const VxParams &p, const VxFrameDev &fd, uint8_t *scratch, int cx, int cy, int cw, int chh)
{
  const int tid = VTX, P = cw * chh;
  const int totalLeft = chh, numLeft = chh >> 1, totalAbove = cw, numAbove = cw >> 1;       // units of 2 chroma samples
  if (tid == 0) {
    const int leftAvail = L.flags[totalLeft - numLeft], aboveAvail = L.flags[totalLeft + numAbove];
    int lb = 0, ar = 0;
    if (leftAvail) for (int i = numLeft; i < totalLeft && L.flags[totalLeft - 1 - i]; i++) lb++;
    if (aboveAvail) for (int i = numAbove; i < totalAbove && L.flags[totalLeft + 1 + i]; i++) ar++;
    L.lm_info[0] = leftAvail; L.lm_info[1] = aboveAvail; L.lm_info[2] = lb; L.lm_info[3] = ar;
  }
  __syncthreads();
  const int leftAvail = uni(L.lm_info[0]), aboveAvail = uni(L.lm_info[1]);
  const int nTop = aboveAvail ? cw + 2 * uni(L.lm_info[3]) : 0, nLeft = leftAvail ? chh + 2 * uni(L.lm_info[2]) : 0;
  const void *rec = fd.rec[0]; const int S = fd.stride[0];
  const int base = (2 * cy) * S + 2 * cx;
  const int firstRowOfCtu = (cy & 63) == 0;
  int16_t *in = lm_in_buf(scratch, 2 * P);
#define LY(dx_, dy_) ld_px<T>(rec, base + (dy_) * S + (dx_))
  for (int e = tid; e < P + nTop + nLeft; e += NT) {
    if (e < P) {
      const int j = e >> ilog2i(cw), i = e & (cw - 1);
      int v;
      if (i == 0 && !leftAvail) v = (LY(0, 2 * j) + LY(0, 2 * j + 1) + 1) >> 1;
      else v = (LY(2 * i, 2 * j) * 2 + LY(2 * i + 1, 2 * j) + LY(2 * i - 1, 2 * j) + LY(2 * i, 2 * j + 1) * 2 + LY(2 * i + 1, 2 * j + 1) + LY(2 * i - 1, 2 * j + 1) + 4) >> 3;
      in[e] = (int16_t) v;
    } else if (e < P + nTop) {
      const int i = e - P;
      int v;
      if (firstRowOfCtu) v = (i == 0 && !leftAvail) ? LY(0, -1) : (LY(2 * i, -1) * 2 + LY(2 * i - 1, -1) + LY(2 * i + 1, -1) + 2) >> 2;
      else v = (i == 0 && !leftAvail) ? (LY(0, -2) + LY(0, -1) + 1) >> 1
                                      : (LY(2 * i, -2) * 2 + LY(2 * i - 1, -2) + LY(2 * i + 1, -2) + LY(2 * i, -1) * 2 + LY(2 * i - 1, -1) + LY(2 * i + 1, -1) + 4) >> 3;
      L.lm_top[i] = (int16_t) v;
    } else {
      const int j = e - P - nTop;
      L.lm_left[j] = (int16_t) ((LY(-2, 2 * j) * 2 + LY(-3, 2 * j) + LY(-1, 2 * j) + LY(-2, 2 * j + 1) * 2 + LY(-3, 2 * j + 1) + LY(-1, 2 * j + 1) + 4) >> 3);
    }
  }
#undef LY
  __threadfence_block();
  __syncthreads();
  if (tid < 6) { int o[3]; cclm_params(in, cw, chh, tid / 3, LM_CHROMA + tid % 3, p.bit_depth, o); L.lm_par[tid / 3][tid % 3][0] = o[0]; L.lm_par[tid / 3][tid % 3][1] = o[1]; L.lm_par[tid / 3][tid % 3][2] = o[2]; }
  __syncthreads();
}
// one wave: prediction of chroma component k with final mode fm into dst (w*h, stride w)
__device__ inline void chroma_pred_wave(int16_t *dst, const int16_t *lm_in, int k, int fm, int w, int h, int bd, int lane)
{
  const int P = w * h;
  if (fm >= LM_CHROMA && fm <= MDLM_T) {                  // predIntraChromaLM 400-420: linearTransform with clipping
    const int a = L.lm_par[k][fm - LM_CHROMA][0], b = L.lm_par[k][fm - LM_CHROMA][1], sh = L.lm_par[k][fm - LM_CHROMA][2];
    for (int i = lane; i < P; i += 64) dst[i] = (int16_t) clip_bd(((a * lm_in[i]) >> sh) + b, bd);
  } else {
    Ipa ip; init_pred_params(w, h, 0, fm, 0, ip);
    const int16_t *top = L.refs[k][0], *left = L.refs[k][1];
    for (int i = lane; i < P; i += 64) { const int py = i >> ilog2i(w), px = i & (w - 1); dst[i] = (int16_t) pred_sample(top, left, w, h, px, py, ip, fm, 0, bd, L.dc_val[k]); }
  }
}

// ------------------------------------------------------------------------------------------------ distortion (one wave)
template <int N> __device__ inline void had1d(int *v)
{
#pragma unroll
  for (int len = 1; len < N; len <<= 1)
#pragma unroll
    for (int i = 0; i < N; i += 2 * len)
#pragma unroll
      for (int j = i; j < i + len; j++) { const int a = v[j], b = v[j + len]; v[j] = a + b; v[j + len] = a - b; }
}
__device__ void satd_tile_shape(int w, int h, int &bw, int &bh)       // CL/RdCost.cpp:2764-2854
{
  if (w > h && (h & 7) == 0 && (w & 15) == 0) { bw = 16; bh = 8; }
  else if (w < h && (w & 7) == 0 && (h & 15) == 0) { bw = 8; bh = 16; }
  else if (w > h && (h & 3) == 0 && (w & 7) == 0) { bw = 8; bh = 4; }
  else if (w < h && (w & 3) == 0 && (h & 7) == 0) { bw = 4; bh = 8; }
  else if ((h & 7) == 0 && (w & 7) == 0) { bw = 8; bh = 8; }
  else if ((h & 3) == 0 && (w & 3) == 0) { bw = 4; bh = 4; }
  else { bw = 2; bh = 2; }
}
// SAD and SATD (xGetHADs) of org vs pred, both w*h tiles with stride w, by one wavefront; scratch >= w*h int16.
// Templated on the Hadamard tile shape so that the per-lane row/column vectors stay in registers.
// BW consecutive int16 as one LDS/global vector access (rows of Hadamard tiles are BW-aligned in 16-byte aligned tiles)
typedef short vs4 __attribute__((vector_size(8)));
typedef short vs8 __attribute__((vector_size(16)));
template <int BW> __device__ inline void row_diff(const int16_t *o, const int16_t *q, int *v)
{
  if constexpr (BW == 4) { const vs4 a = *(const vs4 *) o, b = *(const vs4 *) q; for (int i = 0; i < 4; i++) v[i] = a[i] - b[i]; }
  else if constexpr (BW >= 8) {
#pragma unroll
    for (int k = 0; k < BW; k += 8) { const vs8 a = *(const vs8 *) (o + k), b = *(const vs8 *) (q + k); for (int i = 0; i < 8; i++) v[k + i] = a[i] - b[i]; }
  } else { for (int i = 0; i < BW; i++) v[i] = o[i] - q[i]; }
}
template <int BW> __device__ inline void row_store(int16_t *d, const int *v)
{
  if constexpr (BW == 4) { vs4 a; for (int i = 0; i < 4; i++) a[i] = (short) v[i]; *(vs4 *) d = a; }
  else if constexpr (BW >= 8) {
#pragma unroll
    for (int k = 0; k < BW; k += 8) { vs8 a; for (int i = 0; i < 8; i++) a[i] = (short) v[k + i]; *(vs8 *) (d + k) = a; }
  } else { for (int i = 0; i < BW; i++) d[i] = (int16_t) v[i]; }
}
template <int BW, int BH>
__device__ inline void sad_satd_tiles(const int16_t *org, const int16_t *pred, int w, int P, int16_t *scr, int lane, int &sad_out, int &satd_out)
{
  const int tilesX = w / BW, tsz = BW * BH;
  int sad = 0;
  for (int s = lane; s < P / BW; s += 64) {            // one row segment of a tile per lane
    const int t = s / BH, r = s - t * BH, tx = t & (tilesX - 1), ty = t >> ilog2i(tilesX);      // tilesX is a power of two
    const int base = (ty * BH + r) * w + tx * BW;
    int v[BW];
    row_diff<BW>(org + base, pred + base, v);
#pragma unroll
    for (int i = 0; i < BW; i++) sad += iabs(v[i]);
    had1d<BW>(v);
    row_store<BW>(scr + t * tsz + r * BW, v);
  }
  wave_sync();
  int satd = 0;
  const int ncols = P / BH;                             // column segments; BW consecutive lanes share a tile
  for (int it = 0; it * 64 < ncols; it++) {
    const int q = it * 64 + lane;
    int s = 0;
    if (q < ncols) {
      const int t = q / BW, i = q - t * BW;
      int v[BH];
#pragma unroll
      for (int r = 0; r < BH; r++) v[r] = scr[t * tsz + r * BW + i];
      had1d<BH>(v);
#pragma unroll
      for (int r = 0; r < BH; r++) s += iabs(v[r]);
    }
    s = seg_sum<BW>(s);                                           // tile total on every lane of the group
    if (q < ncols && (lane & (BW - 1)) == 0) {
      int n;
      if (BW == 2) n = s;                                 // CL/RdCost.cpp:2110-2115
      else if (BW == 4 && BH == 4) n = (s + 1) >> 1;      // 2209
      else if (BW == 8 && BH == 8) n = (s + 2) >> 2;      // 2306
      else { const double c = tsz == 128 ? 0x1.6a09e667f3bcdp+3 : 0x1.6a09e667f3bcdp+2; n = (int) ((double) s / c * 2); }   // 2452,2589,2662,2741
      satd += n;
    }
  }
  wave_sync();
  sad_out = sad; satd_out = satd;
}
template <bool SMALL>
__device__ __noinline__ void wave_sad_satd(const int16_t *org_g, const int16_t *pred_g, int16_t *scr_g, int w, int h, int lane,
                              unsigned long long &sad_out, unsigned long long &satd_out, int org_off = 0, int pred_off = 0)
{
  org_g = uni_p(org_g); pred_g = uni_p(pred_g); scr_g = uni_p(scr_g);
  w = uni(w); h = uni(h); org_off = uni(org_off); pred_off = uni(pred_off);       // offsets: second component of a chroma pair
  const int wave_ = uni(VTX >> 6);
  const int16_t *org = (SMALL ? L.org : org_g) + org_off, *pred = (SMALL ? L.wm[wave_].slot : pred_g) + pred_off;
  int16_t *scr = SMALL ? (int16_t *) L.wm[wave_].tmp : scr_g;
  const int P = w * h;
  int bw, bh; satd_tile_shape(w, h, bw, bh);
  int a = 0, b = 0;
  switch (bw * 100 + bh) {
    case 1608: sad_satd_tiles<16, 8>(org, pred, w, P, scr, lane, a, b); break;
    case 816:  sad_satd_tiles<8, 16>(org, pred, w, P, scr, lane, a, b); break;
    case 804:  sad_satd_tiles<8, 4>(org, pred, w, P, scr, lane, a, b); break;
    case 408:  sad_satd_tiles<4, 8>(org, pred, w, P, scr, lane, a, b); break;
    case 808:  sad_satd_tiles<8, 8>(org, pred, w, P, scr, lane, a, b); break;
    case 404:  sad_satd_tiles<4, 4>(org, pred, w, P, scr, lane, a, b); break;
    default:   sad_satd_tiles<2, 2>(org, pred, w, P, scr, lane, a, b); break;
  }
  // both sums fit 32 bits (<= 4096 * 1023 * 16)
  sad_out = (unsigned long long) (unsigned) wave_sum_i32(a);
  satd_out = (unsigned long long) (unsigned) wave_sum_i32(b);
}

// ------------------------------------------------------------------------------------------------ transform + quant (one wave)
// four matrix coefficients (int8, one 32-bit load) times four int16 samples (one 64-bit load) / four int32 values (one 128-bit load):
// rows of the matrices and of the sample tiles start at multiples of their length (>= 4 elements) in 16-byte aligned buffers
struct alignas(16) I32x4 { int x, y, z, w; };
// The first transform stage multiplies int8 matrix rows with 16-bit samples: two packed dot products (v_dot2c_i32_i16: two int16 x int16 products accumulated into 32
// bits, exact) per four samples; the residual pair comes from one packed subtraction (v_pk_sub_i16; samples and predictions are below 2^15).
#ifndef VX_DOT2_I16
typedef short vx_s2 __attribute__((ext_vector_type(2)));
#define VX_DOT2_I16(a_, b_, c_) __builtin_amdgcn_sdot2(__builtin_bit_cast(vx_s2, (uint32_t) (a_)), __builtin_bit_cast(vx_s2, (uint32_t) (b_)), (c_), false)
#define VX_PKSUB_I16(a_, b_) __builtin_bit_cast(uint32_t, __builtin_bit_cast(vx_s2, (uint32_t) (a_)) - __builtin_bit_cast(vx_s2, (uint32_t) (b_)))
#endif
__device__ inline uint2 m8x4_to_16(uint32_t m)                    // four int8 coefficients -> two words of packed int16 pairs
{
  uint2 r;
  r.x = (uint32_t) (uint16_t) (int16_t) (int8_t) m | ((uint32_t) (int) (int8_t) (m >> 8) << 16);
  r.y = (uint32_t) (uint16_t) (int16_t) (int8_t) (m >> 16) | ((uint32_t) ((int) m >> 24) << 16);
  return r;
}
__device__ inline int dot4_s16(uint32_t m, uint2 d)
{
  const uint2 c = m8x4_to_16(m);
  return VX_DOT2_I16(c.y, d.y, VX_DOT2_I16(c.x, d.x, 0));
}
__device__ inline int dot4_resi(uint32_t m, uint2 a, uint2 b)     // coefficients times (a - b), element-wise int16
{
  const uint2 c = m8x4_to_16(m);
  return VX_DOT2_I16(c.y, VX_PKSUB_I16(a.y, b.y), VX_DOT2_I16(c.x, VX_PKSUB_I16(a.x, b.x), 0));
}
__device__ inline int dot4_s32(uint32_t m, I32x4 d)
{
  return (int) (int8_t) m * d.x + (int) (int8_t) (m >> 8) * d.y + (int) (int8_t) (m >> 16) * d.z + (int) (int8_t) (m >> 24) * d.w;
}
// the same with the four data elements taken in reverse order (DCT-VIII rows are DST-VII rows on the reversed input)
__device__ inline uint2 rev4_s16(uint2 d) { uint2 r; r.x = (d.y >> 16) | (d.y << 16); r.y = (d.x >> 16) | (d.x << 16); return r; }
__device__ inline I32x4 rev4_s32(I32x4 d) { I32x4 r; r.x = d.w; r.y = d.z; r.z = d.y; r.w = d.x; return r; }
template <bool SMALL> __device__ inline const int8_t *dct2_matrix(int n)
{
  switch (n) { case 2: return VX_DCT2_2; case 4: return VX_DCT2_4; case 8: return VX_DCT2_8; case 16: return VX_DCT2_16; case 32: return VX_DCT2_32; default: return VX_DCT2_64; }
}
__device__ void load_tables()
{
  const int tid = VTX;
  for (int i = tid; i < 256; i += NT) L.t.bin_frac[i] = VX_BIN_FRAC_BITS[2 * i];
  for (int i = tid; i < NCTX; i += NT) L.t.ctx_rate[i] = VX_CTX_RATE[i];
  if (tid < 12) { L.t.qscale[tid] = VX_QUANT_SCALES[tid]; L.t.iqscale[tid] = VX_INV_QUANT_SCALES[tid]; }
  if (tid < 32) { L.t.ang[tid] = ANG_TABLE[tid]; L.t.inv_ang[tid] = INV_ANG_TABLE[tid]; L.t.gorice_pars[tid] = VX_GORICE_PARS[tid]; }
  if (tid < 128) { L.t.gauss[tid] = GAUSS_FILTER[tid >> 2][tid & 3]; L.t.cubic[tid] = VX_CUBIC_FILTER[tid]; }
  if (tid < 96) L.t.gorice_pos0[tid] = VX_GORICE_POS0[tid];
  if (tid < 128) L.t.rice_len[tid] = (uint8_t) rem_abs_len((unsigned) (tid & 31), (unsigned) (tid >> 5));      // [Rice parameter][value]: bits of the remainder code
  if (tid < 64) L.t.group_idx[tid] = VX_GROUP_IDX[tid];
  if (tid < 36) L.t.mode_num[tid] = VX_MODE_NUM_FAST_2D[tid];
  if (tid < 8) { L.t.intra_thr[tid] = INTRA_FILTER_THR[tid]; L.t.last_prefix[tid] = LAST_PREFIX_CTX[tid]; L.t.mode_shift[tid] = MODE_SHIFT[tid]; }
  if (tid < 52) {
    const int t = tid < 16 ? 0 : tid < 20 ? 1 : tid < 36 ? 2 : 3, n = tid - (t == 0 ? 0 : t == 1 ? 16 : t == 2 ? 20 : 36);
    int x, y; diag_walk(t == 0 ? 4 : t == 2 ? 8 : 2, t == 0 ? 4 : t == 3 ? 8 : 2, n, x, y);
    L.t.cg_scan[tid] = (uint8_t) (x | (y << 4));
    L.t.cg_inv[(t == 0 ? 0 : t == 1 ? 16 : t == 2 ? 20 : 36) + y * (t == 0 ? 4 : t == 2 ? 8 : 2) + x] = (uint8_t) n;
  }
  if (tid < 32) {                                       // the one-dimensional groups of N x 1 / 1 x N blocks: positions in order
    const int n = tid & 15;
    L.t.cg_scan[52 + tid] = (uint8_t) (tid < 16 ? n : n << 4); L.t.cg_inv[52 + tid] = (uint8_t) n;
  }
  for (int i = tid; i < 225; i += NT) {
    int a = 0, b = 0, off = 0;                           // table of (lwg = a, lhg = b) starts at 15 (2^a - 1) + 2^a (2^b - 1)
    for (int aa = 0; aa < 4; aa++) for (int bb = 0; bb < 4; bb++) { const int o = 15 * ((1 << aa) - 1) + (1 << aa) * ((1 << bb) - 1); if (o <= i) { a = aa; b = bb; off = o; } }
    int x, y; diag_walk(1 << a, 1 << b, i - off, x, y);
    L.t.grp_scan[i] = (uint8_t) (x | (y << 4));
    L.t.grp_inv[off + y * (1 << a) + x] = (uint8_t) (i - off);
  }
}
// ---- LMCS chroma residual scaling.  AreaBuf<Pel>::scaleSignal (CL/Buffer.cpp:501-550): the encoder divides the chroma residual by the scale (11 fractional bits), the
// decoder half multiplies it back
__device__ inline int lmcs_scale_fwd(int v, int scale, int bd)
{
  const int mxa = (1 << bd) - 1, sg = v >= 0 ? 1 : -1, a = sg * v;
  const int r = sg * (((a << 11) + (scale >> 1)) / scale);
  return r < -mxa ? -mxa : r > mxa ? mxa : r;
}
__device__ inline int lmcs_scale_inv(int v, int scale, int bd)
{
  const int mxa = (1 << bd) - 1;
  v = v < -mxa - 1 ? -mxa - 1 : v > mxa ? mxa : v;
  const int sg = v >= 0 ? 1 : -1, a = sg * v;
  const int r = sg * ((a * scale + 1024) >> 11);
  return r < -32768 ? -32768 : r > 32767 ? 32767 : r;
}
// Reshape::calculateChromaAdjVpduNei (CL/Reshape.cpp:153-250): every chroma TU inside a 64x64 luma area takes its scale from the average of the reconstructed luma samples
// left of and above the luma CU that holds the area's top-left sample (64 each, clamped at the picture edge), looked up in the model's table.  All threads; the
// result is left in L.lmcs_cadj / L.lmcs_tab.  (nx, ny): the chroma node in luma coordinates.
template <typename T>
__device__ __noinline__ void lmcs_chroma_adj(int nx, int ny)
{
  const VxParams &p = L.par; const VxFrameDev &fd = L.fdv;
  if (!uni(p.lmcs_cadj_on)) { if (VTX == 0) { L.lmcs_cadj = 0; L.lmcs_tab = 0; } __syncthreads(); return; }
  nx = uni(nx); ny = uni(ny);
  const VxUnit tl = fd.units[0][((ny & ~63) >> 2) * p.uw + ((nx & ~63) >> 2)];
  const int x = uni((int) tl.x), y = uni((int) tl.y), tile = uni(L.cur_tile);
  const int availL = get_cu(p, fd, 0, x - 1, y, tile) != nullptr, availA = get_cu(p, fd, 0, x, y - 1, tile) != nullptr;
  const int wave = uni(VTX >> 6), lane = VTX & 63;
  const void *rec = fd.rec[0]; const int st = fd.stride[0];
  if (wave < 2) {
    int v = 0;
    if (wave == 0 && availL) { const int k = (y + lane) >= p.pic_h ? p.pic_h - y - 1 : lane; v = ld_px<T>(rec, (y + k) * st + x - 1); }
    if (wave == 1 && availA) { const int k = (x + lane) >= p.pic_w ? p.pic_w - x - 1 : lane; v = ld_px<T>(rec, (y - 1) * st + x + k); }
    v = wave_sum_i32(v);
    if (lane == 0) L.lm_info[wave] = v;                   // scratch: the CCLM set-up that follows writes it
  }
  __threadfence_block();
  __syncthreads();
  if (VTX == 0) {
    const int n = 64 * (availL + availA), sum = L.lm_info[0] + L.lm_info[1], mx = (1 << p.bit_depth) - 1;
    int v = n == 64 ? (sum + 32) >> 6 : n == 128 ? (sum + 64) >> 7 : 1 << (p.bit_depth - 1);
    v = v < 0 ? 0 : v > mx ? mx : v;
    int idx = p.lmcs_min_bin;
    while (idx <= p.lmcs_max_bin && v >= p.lmcs_pivot[idx + 1]) idx++;
    if (idx > 15) idx = 15;
    L.lmcs_cadj = p.lmcs_cadj[idx]; L.lmcs_tab = 1 + idx;
  }
  __threadfence_block();
  __syncthreads();
}
// residual (org - pred) → DCT-II (TrQuant::xT 835-915) → plain quant (Quant::quant 994-1089) → levels;
// if any level: dequant (423-549) → inverse DCT-II (xIT 917-992) → reco = clip(pred + resi) written over pred.
// Returns SSE(org, reco) and abs-sum via out params.  rec/lev tiles have stride w.
// given >= 0: the levels in lev are taken as coded (cbf = given): only the decoder half runs (DecCu::xIntraRecBlk, DL/DecCu.cpp:199-414).
// SMALL: rec / lev / tmp are the calling wave's LDS buffers (L.wm[wave].slot + buf_off, + 1024, L.wm[wave].tmp); else the _g pointers.
// With VVCX_TOOL_DEPQUANT the quantiser is the trellis of wave_depquant (comp: 0 Y / 1 Cb / 2 Cr, ci: the context set its rate terms are read from =
// the estimator's contexts at this point of the search, cbf_cb: tu.cbf[Cb] when Cr is quantised) and the dequantiser its state machine.
template <bool SMALL, bool SUMABS = false>
__device__ __noinline__ void wave_code_block(const int16_t *org_g, int org_off, int buf_off, int16_t *rec_g, int16_t *lev_g, int32_t *tmp_g, int w, int h, int bd, int qp,
                                int lane, unsigned long long &sse_out, int &cbf_out, int given = -1, int *sumabs_out = nullptr, int comp = 0, int ci = 0, int cbf_cb = 0,
                                int lf = 0, int lfmode = 0, int raw = 0, int qidx = -1, int cadj = 0)
{
  org_g = uni_p(org_g); rec_g = uni_p(rec_g); lev_g = uni_p(lev_g); tmp_g = uni_p(tmp_g); sumabs_out = uni_p(sumabs_out);      // uniform arguments arrive in vector registers: scalar from here on
  // cadj: LMCS chroma residual scale of the block (0: none): the residual is divided by it in front of the transform and multiplied back behind the inverse
  // raw: the block is a bare residual (org = the residual, rec = zeros on entry): rec receives the reconstructed residual, unclipped (joint chroma blocks)
  // lf: cu.lfnstIdx for a block of at least 4x4 (0 otherwise), lfmode: lfnst_mode() of its final intra mode (dependent quantisation only)
  int coef_sum = 0;                                     // SUMABS: sum of |DCT-II coefficient| for the MTS pruning (TrQuant::transformNxN 1049-1124)
  w = uni(w); h = uni(h); bd = uni(bd); qp = uni(qp); given = uni(given);
  const int dq = uni((int) (L.par.tools & TOOL_DEPQUANT)) != 0;
  const int wave_ = uni(VTX >> 6);
  const int16_t *org = (SMALL ? L.org : org_g) + uni(org_off);
  int16_t *rec = SMALL ? L.wm[wave_].slot + uni(buf_off) : rec_g, *lev = SMALL ? L.wm[wave_].slot + BUF + uni(buf_off) : lev_g;
  int32_t *tmp = SMALL ? L.wm[wave_].tmp : tmp_g;
  const int P = w * h, lw = ilog2i(w), lh = ilog2i(h);
  const int zw = imin(w, 32), zh = imin(h, 32), lzw = imin(lw, 5);
  lf = uni(lf); lfmode = uni(lfmode);
  const int lfsb = (w >= 8 && h >= 8) ? 8 : 4;            // with LFNST only the top-left 4x4 / 8x8 of the primary coefficients is kept (xT 855-868)
  const int fzw = lf ? lfsb : zw, fzh = lf ? lfsb : zh, lfzw = lf ? ilog2i(lfsb) : lzw;
  const int8_t *Mw = dct2_matrix<SMALL>(w), *Mh = dct2_matrix<SMALL>(h);
  const int shift1 = lw + bd + 6 - 15, shift2 = lh + 6;
  const int rnd1 = shift1 > 0 ? 1 << (shift1 - 1) : 0, rnd2 = 1 << (shift2 - 1);
  cadj = uni(cadj);
  if (cadj && given < 0) {                              // the scaled residual passes through the (still unused) level buffer
    for (int o = lane; o < P; o += 64) lev[o] = (int16_t) lmcs_scale_fwd(org[o] - rec[o], cadj, bd);
    wave_sync();
  }
  // stage 1 (horizontal): tmp[k*h + j] = (sum_i Mw[k][i] * resi[j][i] + rnd) >> shift1, k < zw
#ifndef VX_NO_MFMA
  // 32- and 64-point rows on the matrix cores: D[k][j] = sum_i Mw[k][i] * resi[j][i] is a 32 x w by w x h GEMM of an int8 matrix with 11-bit residuals.  The residual is
  // split into a signed high byte and an unsigned low byte; the low byte is re-centred (lo - 128, so that it is an int8) and the 128 * (row sum of Mw) it leaves out is added back:
  // the DCT-II rows sum to zero except row 0 (64 * w; tests/test_host_cpu.py checks the tables).  v_mfma_i32_32x32x16_i8: lane l feeds row l % 32 of A / column l % 32 of B
  // with the eight k of half l / 32; the 16 results of a lane are rows 8 * (r / 4) + 4 * (l / 32) + r % 4 of column l % 32.  Exact: all partial sums stay far below 2^31.
  if (!SMALL && given < 0 && !cadj && !lf && (w == 32 || w == 64)) {
    typedef int vx_i16 __attribute__((ext_vector_type(16)));
    const int col = lane & 31, half = lane >> 5;
    for (int jt = 0; jt < h; jt += 32) {
      vx_i16 accH = { 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0 }, accL = accH;
      const int j = jt + col;
      for (int q = 0; q < w; q += 16) {
        const int i0 = q + half * 8;
        const uint32_t a0 = *(const uint32_t *) (Mw + col * w + i0), a1 = *(const uint32_t *) (Mw + col * w + i0 + 4);
        const long a = (long) (((unsigned long long) a1 << 32) | a0);
        unsigned long long bh = 0, bl = 0;
        if (j < h) {
#pragma unroll
          for (int e = 0; e < 8; e++) {
            const int r = org[j * w + i0 + e] - rec[j * w + i0 + e];
            bh |= (unsigned long long) (unsigned) ((r >> 8) & 255) << (8 * e);
            bl |= (unsigned long long) (unsigned) (((r & 255) - 128) & 255) << (8 * e);
          }
        }
        accH = __builtin_amdgcn_mfma_i32_32x32x16_i8(a, (long) bh, accH, 0, 0, 0);
        accL = __builtin_amdgcn_mfma_i32_32x32x16_i8(a, (long) bl, accL, 0, 0, 0);
      }
      if (j < h) {
#pragma unroll
        for (int r = 0; r < 16; r++) {
          const int k = 8 * (r >> 2) + 4 * half + (r & 3);
          const int sacc = 256 * accH[r] + accL[r] + (k == 0 ? 128 * 64 * w : 0);
          tmp[k * h + j] = (sacc + rnd1) >> shift1;
        }
      }
    }
  } else
#endif
  if (given < 0) for (int o = lane; o < fzw * h; o += 64) {
    const int k = o >> lh, j = o & (h - 1);
    int s = 0;
    if (cadj) { if (w >= 4) for (int i = 0; i < w; i += 4) s += dot4_s16(*(const uint32_t *) (Mw + k * w + i), *(const uint2 *) (lev + j * w + i)); else for (int i = 0; i < w; i++) s += Mw[k * w + i] * lev[j * w + i]; }
    else if (w >= 4) for (int i = 0; i < w; i += 4) s += dot4_resi(*(const uint32_t *) (Mw + k * w + i), *(const uint2 *) (org + j * w + i), *(const uint2 *) (rec + j * w + i));
    else for (int i = 0; i < w; i++) s += Mw[k * w + i] * (org[j * w + i] - rec[j * w + i]);
    tmp[o] = (s + rnd1) >> shift1;
  }
  wave_sync();
  // quant parameters
  const int need_sqrt = (lw + lh) & 1;
  const int qscale = L.t.qscale[need_sqrt * 6 + qp % 6];
  const int tr_shift = 15 - bd - ((lw + lh) >> 1) + (need_sqrt ? -1 : 0);
  const int qbits = 14 + qp / 6 + tr_shift;
  const long long qadd = (long long) 171 << (qbits - 9);
  // stage 2 (vertical) + quant: coef[m*w + k], m < zh, k < zw
  if (given < 0 && (w > 32 || h > 32)) { for (int o = lane; o < P; o += 64) lev[o] = 0; wave_sync(); }
  int abs_sum = 0;
  if (given < 0) for (int o = lane; o < fzw * fzh; o += 64) {
    const int m = o >> lfzw, k = o & (fzw - 1);
    int s = 0;
    if (h >= 4) for (int j = 0; j < h; j += 4) s += dot4_s32(*(const uint32_t *) (Mh + m * h + j), *(const I32x4 *) (tmp + k * h + j));
    else for (int j = 0; j < h; j++) s += Mh[m * h + j] * tmp[k * h + j];
    const int c = (s + rnd2) >> shift2;
    if (SUMABS) coef_sum += iabs(c);
    if (dq) { lev[m * w + k] = (int16_t) c; continue; }       // the trellis works on the coefficients (15 bits + sign by construction of the transform shifts)
    const long long t = (long long) iabs(c) * qscale;
    int q = (int) ((t + qadd) >> qbits);
    abs_sum += q;
    if (c < 0) q = -q;
    q = q < -32768 ? -32768 : q > 32767 ? 32767 : q;
    lev[m * w + k] = (int16_t) q;
  }
  if (lf && given < 0) { wave_sync(); wave_lfnst_fwd(lev, w, w, h, lfmode, lf, (int32_t *) &L.wm[wave_].ws, lane); }      // xFwdLfnst (TrQuant::transformNxN 1220-1223)
  if (given == -2) {                                    // forward half only (batched full-RD stage): the coefficients stay in lev for the trellis of the batch
    if (SUMABS) *sumabs_out = wave_sum_i32(coef_sum);
    wave_sync();
    sse_out = 0; cbf_out = 0;
    return;
  }
  if (dq && given < 0) {
    wave_sync();
    abs_sum = wave_depquant<SMALL>(lev, buf_off, L.par.scratch + (size_t) blockIdx.x * L.par.scratch_per_stream, ci, w, h, comp, VX_CTX_QtCbf[comp] + (comp == 2 ? cbf_cb : 0), 0, lf, lane, qidx);
  } else abs_sum = given < 0 ? uni(wave_sum_i32(abs_sum)) : given;
  if (SUMABS) *sumabs_out = wave_sum_i32(coef_sum);
  wave_sync();
  unsigned long long sse = 0;
  if (abs_sum > 0) {
    const int iscale = L.t.iqscale[need_sqrt * 6 + qp % 6];
    const int right_shift = 6 - (tr_shift + qp / 6);
    int tbd = 32 + right_shift - 7; if (tbd > 16) tbd = 16;
    const int in_min = -(1 << (tbd - 1)), in_max = (1 << (tbd - 1)) - 1;
    // dequantised coefficients once (they are clipped to 16 bits, Quant::dequant 423-549): deq[m*zw + k], int16 at the start of tmp,
    // followed by the (16-bit clipped) output of the vertical stage: zw*zh + zw*h int16 <= the zw*h int32 the forward pass used
    int16_t *deq = (int16_t *) tmp, *tcol = deq + zw * zh;
    if (dq) { wave_dequant_dq(lev, deq, w, h, zw, zh, bd, qp, lane); if (lf) wave_lfnst_inv(deq, zw, w, h, lfmode, lf, (int32_t *) &L.wm[wave_].ws, lane); }      // xInvLfnst (invTransformNxN 593-596)
    else {
    for (int o = lane; o < zw * zh; o += 64) {
      const int m = o >> lzw, k = o & (zw - 1);
      int q = lev[m * w + k]; q = q < in_min ? in_min : q > in_max ? in_max : q;
      int v = right_shift > 0 ? (q * iscale + (1 << (right_shift - 1))) >> right_shift : (q * iscale) << (-right_shift);
      deq[o] = (int16_t) (v < -32768 ? -32768 : v > 32767 ? 32767 : v);
    }
    wave_sync();
    }
    // inverse stage 1 (vertical): t[j*h + i] = clip((sum_k Mh[k][i] * deq[k][j] + 64) >> 7), j < zw
    for (int o = lane; o < zw * h; o += 64) {
      const int j = o >> lh, i = o & (h - 1);
      int s = 0;
      for (int k = 0; k < zh; k++) s += Mh[k * h + i] * deq[(k << lzw) + j];
      int v = (s + 64) >> 7;
      tcol[o] = (int16_t) (v < -32768 ? -32768 : v > 32767 ? 32767 : v);
    }
    wave_sync();
    const int ishift2 = (6 + 15 - 1) - bd, irnd2 = 1 << (ishift2 - 1);
    const int mx = (1 << bd) - 1;
    for (int o = lane; o < P; o += 64) {
      const int j2 = o >> lw, i2 = o & (w - 1);
      int s = 0;
      for (int k = 0; k < zw; k++) s += Mw[k * w + i2] * tcol[k * h + j2];
      int r = (s + irnd2) >> ishift2;
      r = r < -32768 ? -32768 : r > 32767 ? 32767 : r;
      if (raw) { rec[o] = (int16_t) r; continue; }
      if (cadj) r = lmcs_scale_inv((int) (int16_t) r, cadj, bd);
      int v = rec[o] + (int) (int16_t) r;
      v = v < 0 ? 0 : v > mx ? mx : v;
      rec[o] = (int16_t) v;
      const int d = org[o] - v;
      sse += (unsigned long long) (d * d);
    }
  } else {
    for (int o = lane; o < P; o += 64) { const int d = org[o] - rec[o]; sse += (unsigned long long) (d * d); }
  }
  wave_sync();
  sse_out = wave_sum_u64(sse);
  cbf_out = abs_sum > 0;
}

// ---- explicit MTS (TrQuant::getTrTypes 817-830): mts_idx 2..5 → (horizontal, vertical) ∈ {DST-VII, DCT-VIII}; tr: 0 DCT2, 1 DCT8, 2 DST7
__device__ inline void mts_types(int mts, int &trh, int &trv)
{
  if (mts < 2) { trh = trv = 0; return; }
  trh = ((mts - 2) & 1) ? 1 : 2; trv = ((mts - 2) >> 1) ? 1 : 2;
}
template <bool SMALL> __device__ inline const int8_t *tr_matrix(int tr, int n)
{
  if (tr == 0) return dct2_matrix<SMALL>(n);
  return n == 4 ? VX_DST7_4 : n == 8 ? VX_DST7_8 : n == 16 ? VX_DST7_16 : VX_DST7_32;
}
__device__ inline int tr_coef(const int8_t *M, int n, int tr, int k, int i)
{
  if (tr == 1) { const int v = M[k * n + (n - 1 - i)]; return (k & 1) ? -v : v; }
  return M[k * n + i];
}
// forward 2-D transform of (org - pred) with the transform pair of mts and the sum of |coefficient| (the measure TrQuant::transformNxN
// 1049-1124 prunes the MTS candidates with); pred is the calling wave's candidate buffer (SMALL) or pred_g
template <bool SMALL>
__device__ __noinline__ int wave_fwd_sumabs(const int16_t *org_g, const int16_t *pred_g, int32_t *tmp_g, int w, int h, int bd, int mts, int lane)
{
  org_g = uni_p(org_g); pred_g = uni_p(pred_g); tmp_g = uni_p(tmp_g);
  w = uni(w); h = uni(h); bd = uni(bd); mts = uni(mts);
  const int wave_ = uni(VTX >> 6);
  const int16_t *org = SMALL ? L.org : org_g, *pred = SMALL ? L.wm[wave_].slot : pred_g;
  int32_t *tmp = SMALL ? L.wm[wave_].tmp : tmp_g;
  int trh, trv; mts_types(mts, trh, trv);
  const int lw = ilog2i(w), lh = ilog2i(h);
  const int zw = (trh && w == 32) ? 16 : imin(w, 32), zh = (trv && h == 32) ? 16 : imin(h, 32), lzw = ilog2i(zw);
  const int8_t *Mw = tr_matrix<SMALL>(trh, w), *Mh = tr_matrix<SMALL>(trv, h);
  const int shift1 = lw + bd + 6 - 15, shift2 = lh + 6;
  const int rnd1 = shift1 > 0 ? 1 << (shift1 - 1) : 0, rnd2 = 1 << (shift2 - 1);
  for (int o = lane; o < zw * h; o += 64) {
    const int k = o >> lh, j = o & (h - 1);
    int s = 0;
    // DCT-VIII[k][i] = (-1)^k DST-VII[k][n-1-i]: the DST-VII row applied to the reversed input, sign by the parity of k
    if (trh == 1) { for (int i = 0; i < w; i += 4) s += dot4_resi(*(const uint32_t *) (Mw + k * w + i), rev4_s16(*(const uint2 *) (org + j * w + w - 4 - i)), rev4_s16(*(const uint2 *) (pred + j * w + w - 4 - i))); if (k & 1) s = -s; }
    else for (int i = 0; i < w; i += 4) s += dot4_resi(*(const uint32_t *) (Mw + k * w + i), *(const uint2 *) (org + j * w + i), *(const uint2 *) (pred + j * w + i));
    tmp[o] = (s + rnd1) >> shift1;
  }
  wave_sync();
  int sa = 0;
  for (int o = lane; o < zw * zh; o += 64) {
    const int m = o >> lzw, k = o & (zw - 1);
    int s = 0;
    if (trv == 1) { for (int j = 0; j < h; j += 4) s += dot4_s32(*(const uint32_t *) (Mh + m * h + j), rev4_s32(*(const I32x4 *) (tmp + k * h + h - 4 - j))); if (m & 1) s = -s; }
    else for (int j = 0; j < h; j += 4) s += dot4_s32(*(const uint32_t *) (Mh + m * h + j), *(const I32x4 *) (tmp + k * h + j));
    sa += iabs((s + rnd2) >> shift2);
  }
  sa = wave_sum_i32(sa);
  wave_sync();
  return sa;
}
// wave_code_block with an explicit-MTS transform pair (mts 2..5; blocks up to 32x32): same steps, DST-VII / DCT-VIII matrices, and the
// 32-point transforms keep 16 coefficients (TrQuant::xT 853-854, xIT 935-936)
template <bool SMALL>
__device__ __noinline__ void wave_code_block_mts(const int16_t *org_g, int16_t *rec_g, int16_t *lev_g, int32_t *tmp_g, int w, int h, int bd, int qp, int mts,
                                                 int lane, unsigned long long &sse_out, int &cbf_out, int given = -1)
{
  org_g = uni_p(org_g); rec_g = uni_p(rec_g); lev_g = uni_p(lev_g); tmp_g = uni_p(tmp_g);
  w = uni(w); h = uni(h); bd = uni(bd); qp = uni(qp); given = uni(given); mts = uni(mts);
  const int dq = uni((int) (L.par.tools & TOOL_DEPQUANT)) != 0;        // luma only: the rate terms come from the node's start contexts (CI_CUR)
  const int wave_ = uni(VTX >> 6);
  const int16_t *org = SMALL ? L.org : org_g;
  int16_t *rec = SMALL ? L.wm[wave_].slot : rec_g, *lev = SMALL ? L.wm[wave_].slot + BUF : lev_g;
  int32_t *tmp = SMALL ? L.wm[wave_].tmp : tmp_g;
  int trh, trv; mts_types(mts, trh, trv);
  const int P = w * h, lw = ilog2i(w), lh = ilog2i(h);
  const int zw = (trh && w == 32) ? 16 : w, zh = (trv && h == 32) ? 16 : h, lzw = ilog2i(zw);
  const int8_t *Mw = tr_matrix<SMALL>(trh, w), *Mh = tr_matrix<SMALL>(trv, h);
  const int shift1 = lw + bd + 6 - 15, shift2 = lh + 6;
  const int rnd1 = shift1 > 0 ? 1 << (shift1 - 1) : 0, rnd2 = 1 << (shift2 - 1);
  if (given < 0) for (int o = lane; o < zw * h; o += 64) {
    const int k = o >> lh, j = o & (h - 1);
    int s = 0;
    // DCT-VIII[k][i] = (-1)^k DST-VII[k][n-1-i]: the DST-VII row applied to the reversed input, sign by the parity of k
    if (trh == 1) { for (int i = 0; i < w; i += 4) s += dot4_resi(*(const uint32_t *) (Mw + k * w + i), rev4_s16(*(const uint2 *) (org + j * w + w - 4 - i)), rev4_s16(*(const uint2 *) (rec + j * w + w - 4 - i))); if (k & 1) s = -s; }
    else for (int i = 0; i < w; i += 4) s += dot4_resi(*(const uint32_t *) (Mw + k * w + i), *(const uint2 *) (org + j * w + i), *(const uint2 *) (rec + j * w + i));
    tmp[o] = (s + rnd1) >> shift1;
  }
  wave_sync();
  const int need_sqrt = (lw + lh) & 1;
  const int qscale = L.t.qscale[need_sqrt * 6 + qp % 6];
  const int tr_shift = 15 - bd - ((lw + lh) >> 1) + (need_sqrt ? -1 : 0);
  const int qbits = 14 + qp / 6 + tr_shift;
  const long long qadd = (long long) 171 << (qbits - 9);
  if (given < 0 && (zw < w || zh < h)) { for (int o = lane; o < P; o += 64) lev[o] = 0; wave_sync(); }
  int abs_sum = 0;
  if (given < 0) for (int o = lane; o < zw * zh; o += 64) {
    const int m = o >> lzw, k = o & (zw - 1);
    int s = 0;
    if (trv == 1) { for (int j = 0; j < h; j += 4) s += dot4_s32(*(const uint32_t *) (Mh + m * h + j), rev4_s32(*(const I32x4 *) (tmp + k * h + h - 4 - j))); if (m & 1) s = -s; }
    else for (int j = 0; j < h; j += 4) s += dot4_s32(*(const uint32_t *) (Mh + m * h + j), *(const I32x4 *) (tmp + k * h + j));
    const int c = (s + rnd2) >> shift2;
    if (dq) { lev[m * w + k] = (int16_t) c; continue; }
    const long long t = (long long) iabs(c) * qscale;
    int q = (int) ((t + qadd) >> qbits);
    abs_sum += q;
    if (c < 0) q = -q;
    q = q < -32768 ? -32768 : q > 32767 ? 32767 : q;
    lev[m * w + k] = (int16_t) q;
  }
  if (given == -2) { wave_sync(); sse_out = 0; cbf_out = 0; return; }      // forward half only (batched full-RD stage)
  if (dq && given < 0) {
    wave_sync();
    abs_sum = wave_depquant<SMALL>(lev, 0, L.par.scratch + (size_t) blockIdx.x * L.par.scratch_per_stream, CI_CUR, w, h, 0, VX_CTX_QtCbf[0], 1, 0, lane);
  } else abs_sum = given < 0 ? uni(wave_sum_i32(abs_sum)) : given;
  wave_sync();
  unsigned long long sse = 0;
  if (abs_sum > 0) {
    const int iscale = L.t.iqscale[need_sqrt * 6 + qp % 6];
    const int right_shift = 6 - (tr_shift + qp / 6);
    int tbd = 32 + right_shift - 7; if (tbd > 16) tbd = 16;
    const int in_min = -(1 << (tbd - 1)), in_max = (1 << (tbd - 1)) - 1;
    int16_t *deq = (int16_t *) tmp, *tcol = deq + zw * zh;       // as in wave_code_block: coefficients dequantised once, 16-bit intermediate
    if (dq) wave_dequant_dq(lev, deq, w, h, zw, zh, bd, qp, lane);
    else {
    for (int o = lane; o < zw * zh; o += 64) {
      const int m = o >> lzw, k = o & (zw - 1);
      int q = lev[m * w + k]; q = q < in_min ? in_min : q > in_max ? in_max : q;
      int v = right_shift > 0 ? (q * iscale + (1 << (right_shift - 1))) >> right_shift : (q * iscale) << (-right_shift);
      deq[o] = (int16_t) (v < -32768 ? -32768 : v > 32767 ? 32767 : v);
    }
    wave_sync();
    }
    for (int o = lane; o < zw * h; o += 64) {
      const int j = o >> lh, i = o & (h - 1);
      int s = 0;
      if (trv == 1) for (int k = 0; k < zh; k++) { const int v = Mh[k * h + h - 1 - i] * deq[(k << lzw) + j]; s += (k & 1) ? -v : v; }
      else for (int k = 0; k < zh; k++) s += Mh[k * h + i] * deq[(k << lzw) + j];
      int v = (s + 64) >> 7;
      tcol[o] = (int16_t) (v < -32768 ? -32768 : v > 32767 ? 32767 : v);
    }
    wave_sync();
    const int ishift2 = (6 + 15 - 1) - bd, irnd2 = 1 << (ishift2 - 1);
    const int mx = (1 << bd) - 1;
    for (int o = lane; o < P; o += 64) {
      const int j2 = o >> lw, i2 = o & (w - 1);
      int s = 0;
      if (trh == 1) for (int k = 0; k < zw; k++) { const int v = Mw[k * w + w - 1 - i2] * tcol[k * h + j2]; s += (k & 1) ? -v : v; }
      else for (int k = 0; k < zw; k++) s += Mw[k * w + i2] * tcol[k * h + j2];
      int r = (s + irnd2) >> ishift2;
      r = r < -32768 ? -32768 : r > 32767 ? 32767 : r;
      int v = rec[o] + (int) (int16_t) r;
      v = v < 0 ? 0 : v > mx ? mx : v;
      rec[o] = (int16_t) v;
      const int d = org[o] - v;
      sse += (unsigned long long) (d * d);
    }
  } else {
    for (int o = lane; o < P; o += 64) { const int d = org[o] - rec[o]; sse += (unsigned long long) (d * d); }
  }
  wave_sync();
  sse_out = wave_sum_u64(sse);
  cbf_out = abs_sum > 0;
}
// The block pipeline of one sub-partition of an ISP CU (TU of tw x th samples; 1 x N, 2 x N, N x 1, N x 2 and larger): implicit transform selection (TrQuant::getTrTypes
// 752-780: DST-VII along a side of 4..16 samples, DCT-II otherwise), the one-stage forms of N x 1 / 1 x N blocks (xT 895-914, xIT 970-983), dependent quantisation
// against the context set ci (the estimator's live contexts: the sub-partitions of a CU are coded one after the other) with the cbf context cbf_ctx (< 0: the cbf is
// inferred), dequantisation, inverse, reconstruction over the prediction in rec, SSE.  org / rec / lev: tiles of the CU (stride cst) at the TU's origin.
// given >= 0: the levels in lev are taken as coded (cbf = given): decoder half only.  tmp: th * min(32, tw) int32 (and its int16 re-use).
// LDSP: every tile of the call lives in LDS (the search of ISP CUs of at most 128 samples)
template <bool LDSP>
__device__ __noinline__ void wave_code_block_isp(const int16_t *org, int16_t *rec, int16_t *lev, int cst, int32_t *tmp, int16_t *cf, uint8_t *scratch, int w, int h, int bd, int qp,
                                                 int lane, unsigned long long &sse_out, int &cbf_out, int given, int ci, int cbf_ctx)
{
  org = uni_p(org); rec = uni_p(rec); lev = uni_p(lev); tmp = uni_p(tmp); cf = uni_p(cf); scratch = uni_p(scratch); ci = uni(ci); cbf_ctx = uni(cbf_ctx);
  if (LDSP) { org = as_lds(org); rec = as_lds(rec); lev = as_lds(lev); tmp = as_lds(tmp); cf = as_lds(cf); }
  // cf: a dense w * h int16 tile for the coefficients / levels (the trellis and the residual syntax take stride w)
  w = uni(w); h = uni(h); bd = uni(bd); qp = uni(qp); given = uni(given); cst = uni(cst);
  const int trh = (w >= 4 && w <= 16) ? 2 : 0, trv = (h >= 4 && h <= 16) ? 2 : 0;
  const int P = w * h, lw = ilog2i(w), lh = ilog2i(h);
  const int zw = imin(w, 32), zh = imin(h, 32), lzw = ilog2i(zw);
  const int8_t *Mw = w > 1 ? tr_matrix<false>(trh, w) : nullptr, *Mh = h > 1 ? tr_matrix<false>(trv, h) : nullptr;
  const int oneD = w == 1 || h == 1;
  ISP_T(i0);
  if (given < 0) {
    if (!oneD) {
      const int shift1 = lw + bd + 6 - 15, shift2 = lh + 6;
      const int rnd1 = shift1 > 0 ? 1 << (shift1 - 1) : 0, rnd2 = 1 << (shift2 - 1);
      for (int o = lane; o < zw * h; o += 64) {
        const int k = o >> lh, j = o & (h - 1);
        int s = 0;
        for (int i = 0; i < w; i++) s += Mw[k * w + i] * (org[j * cst + i] - rec[j * cst + i]);
        tmp[o] = (s + rnd1) >> shift1;
      }
      wave_sync();
      for (int o = lane; o < P; o += 64) cf[o] = 0;
      wave_sync();
      for (int o = lane; o < zw * zh; o += 64) {
        const int m = o >> lzw, k = o & (zw - 1);
        int s = 0;
        for (int j = 0; j < h; j++) s += Mh[m * h + j] * tmp[k * h + j];
        cf[m * w + k] = (int16_t) ((s + rnd2) >> shift2);
      }
    } else {
      const int n = w * h, ln = lw + lh, zn = imin(n, 32), shift = ln + bd + 6 - 15, rnd = shift > 0 ? 1 << (shift - 1) : 0, step = w == 1 ? cst : 1;
      const int8_t *M = w == 1 ? Mh : Mw;
      for (int o = lane; o < n; o += 64) {
        int s = 0;
        if (o < zn) for (int i = 0; i < n; i++) s += M[o * n + i] * (org[i * step] - rec[i * step]);
        cf[o] = (int16_t) (o < zn ? (s + rnd) >> shift : 0);
      }
    }
    wave_sync();
  } else {
    for (int o = lane; o < P; o += 64) cf[o] = lev[(o >> lw) * cst + (o & (w - 1))];
    wave_sync();
  }
  int abs_sum = given;
  ISP_T(i1);
  if (given < 0) abs_sum = LDSP ? wave_depquant<true>(nullptr, (int) (cf - (L.wm[uni(VTX >> 6)].slot + BUF)), scratch, ci, w, h, 0, cbf_ctx, 0, 0, lane)
                                : wave_depquant<false>(cf, 0, scratch, ci, w, h, 0, cbf_ctx, 0, 0, lane);
  wave_sync();
  ISP_T(i2);
  if (given < 0) { for (int o = lane; o < P; o += 64) lev[(o >> lw) * cst + (o & (w - 1))] = cf[o]; }
  unsigned long long sse = 0;
  const int mx = (1 << bd) - 1;
  if (abs_sum > 0) {
    int16_t *deq = (int16_t *) tmp, *tcol = deq + zw * zh;
    wave_dequant_dq(cf, deq, w, h, zw, zh, bd, qp, lane);
    const int ishift2 = (6 + 15 - 1) - bd;
    if (!oneD) {
      for (int o = lane; o < zw * h; o += 64) {
        const int j = o >> lh, i = o & (h - 1);
        int s = 0;
        for (int k = 0; k < zh; k++) s += Mh[k * h + i] * deq[(k << lzw) + j];
        const int v = (s + 64) >> 7;
        tcol[o] = (int16_t) (v < -32768 ? -32768 : v > 32767 ? 32767 : v);
      }
      wave_sync();
      const int irnd2 = 1 << (ishift2 - 1);
      for (int o = lane; o < P; o += 64) {
        const int j2 = o >> lw, i2 = o & (w - 1);
        int s = 0;
        for (int k = 0; k < zw; k++) s += Mw[k * w + i2] * tcol[k * h + j2];
        int r = (s + irnd2) >> ishift2;
        r = r < -32768 ? -32768 : r > 32767 ? 32767 : r;
        int v = rec[j2 * cst + i2] + (int) (int16_t) r;
        v = v < 0 ? 0 : v > mx ? mx : v;
        rec[j2 * cst + i2] = (int16_t) v;
        const int d = org[j2 * cst + i2] - v;
        sse += (unsigned long long) (d * d);
      }
    } else {
      const int n = w * h, zn = imin(n, 32), sh = ishift2 + 1, rnd = 1 << (sh - 1), step = w == 1 ? cst : 1;
      const int8_t *M = w == 1 ? Mh : Mw;
      for (int o = lane; o < n; o += 64) {
        int s = 0;
        for (int k = 0; k < zn; k++) s += M[k * n + o] * deq[k];
        int r = (s + rnd) >> sh;
        r = r < -32768 ? -32768 : r > 32767 ? 32767 : r;
        int v = rec[o * step] + (int) (int16_t) r;
        v = v < 0 ? 0 : v > mx ? mx : v;
        rec[o * step] = (int16_t) v;
        const int d = org[o * step] - v;
        sse += (unsigned long long) (d * d);
      }
    }
  } else {
    for (int o = lane; o < P; o += 64) { const int a = (o >> lw) * cst + (o & (w - 1)); const int d = org[a] - rec[a]; sse += (unsigned long long) (d * d); }
  }
  wave_sync();
  sse_out = wave_sum_u64(sse);
  cbf_out = abs_sum > 0;
  { ISP_T(i3); ISP_ADD(16, i0, i1); ISP_ADD(17, i1, i2); ISP_ADD(18, i2, i3); }
}
// CABACWriter::mts_coding 3885-3941 for a TU where MTS is allowed and transform skip is not (JVET_O0294 contexts); lane 0 / thread 0
template <bool WR = false>
__device__ inline void enc_mts_idx(Cab &cb, int mts)
{
  enc_bin<WR>(cb, (unsigned) (mts != 0), VX_CTX_MTSIndex + 0);
  if (mts) for (int i = 0; i < 3; i++) { const unsigned sym = mts > i + 2; enc_bin<WR>(cb, sym, VX_CTX_MTSIndex + 7 + i); if (!sym) break; }
}
__device__ inline int mts_allowed(const VxParams &p, int w, int h) { return (p.tools & TOOL_MTS) && w <= 32 && h <= 32; }     // TU::isMTSAllowed, CL/UnitTools.cpp:4549-4565
// the transform pair of an MTS pass (LFNST on: MTS is a CU-level pass, xRecurIntraCodingLumaQT 3474-3498): DST7/DST7 in transform group 0, then the pair the
// intra mode makes more likely first (moreProbMTSIdxFirst), its mirror, DCT8/DCT8
__device__ inline int pass_mts_idx(int grp, int dir) { return grp == 0 ? 2 : grp == 1 ? (dir < 34 ? 4 : 3) : grp == 2 ? (dir < 34 ? 3 : 4) : 5; }
// cuCtx.lfnstLastScanPos / violatesLfnstConstrained of one coded block (CABACWriter::residual_coding 3837-3850): bit 0 / bit 1; last = its last scan position
__device__ inline int lfnst_flags(int last, int w, int h)
{
  if (w < 4 || h < 4 || last < 0) return 0;
  const int maxPos = ((w == 4 && h == 4) || (w == 8 && h == 8)) ? 7 : 15;
  return (last >= 1 ? 1 : 0) | (last > maxPos ? 2 : 0);
}
// CABACWriter::residual_lfnst_mode (3989-4100) of a CU of a separate tree (context 1); w, h: the CU in luma samples; mip: cu.mipFlag (luma);
// non_dct2: the luma block is coded with an explicit MTS pair; flags: OR of lfnst_flags over the CU's coded blocks.  lane 0 / thread 0
template <bool WR = false>
__device__ inline void enc_lfnst_idx(Cab &cb, int ch, int w, int h, int mip, int non_dct2, int flags, int lfnst)
{
  if (!(L.par.tools & TOOL_LFNST)) return;
  if (!ch && mip && !(w >= 16 && h >= 16)) return;
  if (ch && imin(w >> 1, h >> 1) < 4) return;
  if (w > 64 || h > 64) return;
  if (!(flags & 1) || (flags & 2) || non_dct2) return;
  enc_bin<WR>(cb, lfnst ? 1u : 0u, VX_CTX_LFNSTIdx + 1);
  if (lfnst) enc_ep<WR>(cb, (uint32_t) (lfnst - 1), 1);
}

// ------------------------------------------------------------------------------------------------ transform skip (VVCX_TOOL_TS)
// A luma TU of at most 32x32 may skip the transform (tu.mtsIdx = MTS_SKIP = 1): xTransformSkip / xITransformSkip (CL/TrQuant.cpp:1394-1440, 996-1041) scale the residual,
// QuantRDOQ::xRateDistOptQuantTS (CL/QuantRDOQ.cpp:1243-1483; DepQuant::quant hands MTS_SKIP luma blocks to it, CL/DepQuant.cpp:1755-1781) picks the levels with rates
// read from the TS context sets, Quant::dequant (CL/Quant.cpp:423-549) without the sqrt(2) adjustment at max(QP', 4) inverts it, and residual_codingTS
// (EL/CABACWriter.cpp:4306-4555) codes the levels in forward scan order with a budget of 2 * w * h context-coded bins.  BDPCM is off (BIN/encoder_intra.cfg).
__device__ inline int ts_allowed(const VxParams &p, int w, int h) { return (p.tools & TOOL_TS) && w <= 32 && h <= 32; }      // TU::isTSAllowed (CL/UnitTools.cpp:4524-4546), TransformSkipLog2MaxSize 5
__device__ inline int ts_shift(int w, int h, int bd) { return 15 - bd - ((ilog2i(w) + ilog2i(h)) >> 1); }                     // getTransformShift, TU::needsSqrt2Scale false for TS
__device__ inline int ts_scale(int r, int sh) { return sh >= 0 ? r * (1 << sh) : (r + (1 << (-sh - 1))) >> -sh; }
__device__ inline int ts_qp(int qp) { return imax(qp, 4); }                                                                   // QpParam::Qp(isTransformSkip), min_qp_prime_ts_minus4 0
// CABACWriter::mts_coding 3885-3941 of a luma TU with cbf (not an ISP one): transform_skip_flag where TU::isTSAllowed, then the MTS index where TU::isMTSAllowed
template <bool WR = false>
__device__ inline void enc_tu_mts(Cab &cb, int w, int h, int mts)
{
  if (ts_allowed(L.par, w, h)) enc_bin<WR>(cb, (unsigned) (mts == 1), VX_CTX_MTSIndex + 6);
  if (mts != 1 && mts_allowed(L.par, w, h)) enc_mts_idx<WR>(cb, mts);
}
// neighTS / deriveModCoeff / signCtxIdAbsTS / templateAbsSumTS (CL/ContextModelling.h:221-360, bdpcm 0)
__device__ inline int ts_mod_coeff(int right, int below, int a) { const int pm = imax(iabs(below), iabs(right)); return a == pm ? 1 : (a < pm ? a + 1 : a); }
__device__ inline int ts_sign_ctx(int right, int below) { if ((right == 0 && below == 0) || (right * below) < 0) return 0; return (right >= 0 && below >= 0) ? 1 : 2; }
__device__ inline int ts_rice(int right, int below) { const int s = imin(iabs(right) + iabs(below), 31); return s < 12 ? 0 : s < 25 ? 1 : 2; }     // g_auiGoRiceParsCoeff-style table of templateAbsSumTS
enum { TST_SIG = 0, TST_SIGN = 6, TST_GT1 = 12, TST_PAR = 18, TST_GTX = 20, TST_GRP = 28, TST_N = 34 };
// fractional bits of the TS context sets at the estimator's current state ([set + 2 * ctx + bin]); all threads, the caller synchronises
__device__ inline void ts_build_tables()
{
  const int t = VTX;
  if (t < TST_N) {
    int ctx;
    if (t < TST_SIGN) ctx = VX_CTX_TsSigFlag + (t >> 1);
    else if (t < TST_GT1) ctx = VX_CTX_TsResidualSign + ((t - TST_SIGN) >> 1);
    else if (t < TST_PAR) ctx = VX_CTX_TsLrg1Flag + ((t - TST_GT1) >> 1);
    else if (t < TST_GTX) ctx = VX_CTX_TsParFlag;
    else if (t < TST_GRP) ctx = VX_CTX_TsGtxFlag + 1 + ((t - TST_GTX) >> 1);
    else ctx = VX_CTX_TsSigCoeffGroup + ((t - TST_GRP) >> 1);
    L.ts_tab[t] = (int) frac_bits_of(L.ctxs[CI_CUR], ctx, (unsigned) (t & 1));
  }
}
// xGetICRateTS (CL/QuantRDOQ.cpp:1898-1976): bits of a level whose modified magnitude is a
__device__ inline int ts_ic_rate(unsigned a, int signRate, int numPos, int rice)
{
  if (a == 0) return 0;
  int rate = signRate;
  if (a == 1) return rate + L.ts_tab[TST_GT1 + 2 * numPos];
  rate += L.ts_tab[TST_GT1 + 2 * numPos + 1] + L.ts_tab[TST_PAR + ((a - 2) & 1)];
  unsigned cutoff = 2;
  for (int i = 0; i < 4; i++) { if (a >= cutoff) rate += L.ts_tab[TST_GTX + 2 * i + (a >= cutoff + 2)]; cutoff += 2; }
  if (a >= cutoff) {
    unsigned symbol = (a - cutoff) >> 1, length;
    if (symbol < (5u << rice)) { length = symbol >> rice; rate += (int) ((length + 1 + (unsigned) rice) << 15); }
    else { length = (unsigned) rice; symbol -= 5u << rice; while (symbol >= (1u << length)) symbol -= 1u << (length++); rate += (int) ((5 + length + 1 - (unsigned) rice + length) << 15); }
  }
  return rate;
}
// xRateDistOptQuantTS of one block by ONE lane (the decisions form a chain through the left / above levels and the running cost): cf holds the transform-skip
// "coefficients" (stride w) and receives the levels; rates from L.ts_tab.  Returns absSum.  The blocks of a node's candidates run side by side on different lanes.
__device__ __noinline__ int ts_rdoq_lane(int16_t *cf, int w, int h, int bd, int qp)
{
  const int q = ts_qp(qp), tshift = ts_shift(w, h, bd), qBits = 14 + q / 6 + tshift, qc = L.t.qscale[q % 6];
  const int e2 = 15 - 2 * tshift;                                   // xGetErrScaleCoeff 383-392: 2^15 * 2^(-2 * transformShift) / scale^2
  const double errorScale = (e2 >= 0 ? (double) (1ll << e2) : 1.0 / (double) (1ll << -e2)) / (double) qc / (double) qc;
  const double lambda = L.par.lambda;
  const int ecMax = (1 << 15) - 1, lw = ilog2i(w);
  const ScanGeo g = scan_geo(w, h);
  const int sbSize = 1 << g.lcg, sbNum = (w * h) >> g.lcg;
  const long long cap = 2147483647ll - (1ll << (qBits - 1)), rnd = 1ll << (qBits - 1);
  unsigned long long sigMap = 0; int anySigCG = 0, absSum = 0;
  double baseCost = 0.0;
  for (int sb = 0; sb < sbNum; sb++) {
    const int cgx = g.grp[sb] & 15, cgy = g.grp[sb] >> 4, cgPos = cgy * g.wg + cgx;
    const int sigLeft = cgx > 0 ? (int) ((sigMap >> (cgPos - 1)) & 1) : 0, sigAbove = cgy > 0 ? (int) ((sigMap >> (cgPos - g.wg)) & 1) : 0;
    const int fGrp0 = L.ts_tab[TST_GRP + 2 * (sigLeft + sigAbove)], fGrp1 = L.ts_tab[TST_GRP + 2 * (sigLeft + sigAbove) + 1];
    int noCoeffCoded = 0, cgSig = 0, cgAbs = 0;
    double sigCost = 0, codedLevelandDist = 0, uncodedDist = 0;
    for (int k = 0; k < sbSize; k++) {
      const int blk = scan_blk(g, (sb << g.lcg) + k), y = blk >> lw, x = blk & (w - 1);
      const int c = cf[blk];
      const long long tmpLevel = (long long) iabs(c) * qc, levelDouble = tmpLevel < cap ? tmpLevel : cap;
      const unsigned roundAbs = (unsigned) imin(ecMax, (int) ((levelDouble + rnd) >> qBits));
      const unsigned minAbs = roundAbs > 1 ? roundAbs - 1 : 1;
      const unsigned downAbs = (unsigned) imin(ecMax, (int) (levelDouble >> qBits)), upAbs = (unsigned) imin(ecMax, (int) downAbs + 1);
      const int right = x > 0 ? cf[blk - 1] : 0, below = y > 0 ? cf[blk - w] : 0;
      const unsigned l1 = minAbs != roundAbs ? minAbs : 0xffffffffu;                    // the levels tested after roundAbs (0xffffffff: none)
      const unsigned l2 = (upAbs != roundAbs && upAbs != minAbs && ts_mod_coeff(right, below, (int) upAbs) == 1) ? upAbs : 0xffffffffu;
      const double cost0 = (double) levelDouble * (double) levelDouble * errorScale;
      const int numPos = (right != 0) + (below != 0);
      const int rice = ts_rice(right, below);
      const int signRate = L.ts_tab[TST_SIGN + 2 * ts_sign_ctx(right, below) + (c < 0)];
      const int isLast = k == sbSize - 1 && noCoeffCoded == 0;
      // xGetCodedLevelTSPred 1773-1836
      unsigned best = 0; double costCoeff, costSig = 0, currCostSig = 0; int done = 0;
      if (!isLast && roundAbs < 3) {
        costSig = lambda * (double) L.ts_tab[TST_SIG + 2 * numPos];
        costCoeff = cost0 + costSig;
        if (roundAbs == 0) done = 1;
      } else costCoeff = MAX_DOUBLE;
      if (!done) {
        if (!isLast) currCostSig = lambda * (double) L.ts_tab[TST_SIG + 2 * numPos + 1];
        for (int e = 0; e < 3; e++) {
          const unsigned a = e == 0 ? roundAbs : e == 1 ? l1 : l2;
          if (a == 0xffffffffu) continue;
          const double dErr = (double) (levelDouble - ((long long) a << qBits));
          const double err = dErr * dErr * errorScale;
          double cur = err + lambda * (double) ts_ic_rate((unsigned) ts_mod_coeff(right, below, (int) a), signRate, numPos, rice);
          cur += currCostSig;
          if (cur < costCoeff) { best = a; costCoeff = cur; costSig = currCostSig; }
        }
      }
      if (best > 0) noCoeffCoded++;
      cf[blk] = (int16_t) ((best != 0 && c < 0) ? -(int) best : (int) best);
      baseCost += costCoeff;
      sigCost += costSig;
      if (best) { cgSig = 1; cgAbs += (int) best; codedLevelandDist += costCoeff - costSig; uncodedDist += cost0; }
    }
    if (!cgSig) baseCost += lambda * (double) fGrp0 - sigCost;
    else {
      sigMap |= 1ull << cgPos;
      if (sb != sbNum - 1 || anySigCG) {
        double costZeroSB = baseCost;
        baseCost += lambda * (double) fGrp1;
        costZeroSB += lambda * (double) fGrp0;
        costZeroSB += uncodedDist;
        costZeroSB -= codedLevelandDist;
        costZeroSB -= sigCost;
        if (costZeroSB < baseCost) {
          sigMap &= ~(1ull << cgPos); baseCost = costZeroSB; cgAbs = 0;
          for (int k = 0; k < sbSize; k++) cf[scan_blk(g, (sb << g.lcg) + k)] = 0;
        } else anySigCG = 1;
      }
    }
    absSum += cgAbs;
  }
  return absSum;
}
// residual_codingTS + residual_coding_subblockTS (EL/CABACWriter.cpp:4306-4555, JVET_O0122 / O0409 / O0619 forms) on the estimator / writer: one thread
template <bool WR = false>
__device__ __noinline__ void rc_ts_serial(Cab &cb, const int16_t *lv, int w, int h)
{
  const ScanGeo g = scan_geo(w, h);
  const int n = w * h, cgSize = 1 << g.lcg, nsub = n >> g.lcg, lw = ilog2i(w);
  int remBins = 2 * n;                                  // setNumCtxBins; isContextCoded() = --remaining >= 0
#define TS_BIN(bin_, ctx_) { if (--remBins >= 0) enc_bin<WR>(cb, (unsigned) (bin_), (ctx_)); else enc_ep<WR>(cb, (uint32_t) (bin_), 1); }
  unsigned long long sigScan = 0, sigR = 0;             // significant groups by scan index / those met so far by raster position
  for (int sp = 0; sp < n; sp++) if (lv[scan_blk(g, sp)]) sigScan |= 1ull << (sp >> g.lcg);
  int nSet = 0;                                         // m_sigCoeffGroupFlag.count()
  for (int sub = 0; sub < nsub; sub++) {
    const int cgx = g.grp[sub] & 15, cgy = g.grp[sub] >> 4, cgPos = cgy * g.wg + cgx;
    const int sig = (int) ((sigScan >> sub) & 1);
    if (sig && !((sigR >> cgPos) & 1)) { sigR |= 1ull << cgPos; nSet++; }
    const int sigLeft = cgx > 0 ? (int) ((sigR >> (cgPos - 1)) & 1) : 0, sigAbove = cgy > 0 ? (int) ((sigR >> (cgPos - g.wg)) & 1) : 0;
    const int grpCtx = VX_CTX_TsSigCoeffGroup + sigLeft + sigAbove;
    const int minSub = sub << g.lcg, maxSub = minSub + cgSize - 1;
    const int only1st = nSet - (int) ((sigR >> (nsub - 1)) & 1) == 0;      // only1stSigGroup: the last group's raster position is its scan index
    if (sub != nsub - 1 || !only1st) {
      enc_bin<WR>(cb, (unsigned) sig, grpCtx);
      if (!sig) continue;
    }
    int numNonZero = 0;
    for (int sp = minSub; sp <= maxSub; sp++) {
      const int blk = scan_blk(g, sp), cf = lv[blk], y = blk >> lw, x = blk & (w - 1);
      const int right = x > 0 ? lv[blk - 1] : 0, below = y > 0 ? lv[blk - w] : 0, numPos = (right != 0) + (below != 0);
      if (numNonZero || sp != maxSub) TS_BIN(cf != 0, VX_CTX_TsSigFlag + numPos);
      if (cf) {
        TS_BIN(cf < 0, VX_CTX_TsResidualSign + ts_sign_ctx(right, below));
        numNonZero++;
        int rem = ts_mod_coeff(right, below, iabs(cf)) - 1;
        TS_BIN(rem != 0, VX_CTX_TsLrg1Flag + numPos);
        if (rem) { rem -= 1; TS_BIN(rem & 1, VX_CTX_TsParFlag); }
      }
    }
    for (int sp = minSub; sp <= maxSub; sp++) {
      const int blk = scan_blk(g, sp), y = blk >> lw, x = blk & (w - 1);
      const unsigned a = (unsigned) ts_mod_coeff(x > 0 ? lv[blk - 1] : 0, y > 0 ? lv[blk - w] : 0, iabs(lv[blk]));
      unsigned cutoff = 2;
      for (int i = 0; i < 4; i++) { if (a >= cutoff) TS_BIN(a >= cutoff + 2, VX_CTX_TsGtxFlag + (int) (cutoff >> 1)); cutoff += 2; }
    }
    for (int sp = minSub; sp <= maxSub; sp++) {
      const int blk = scan_blk(g, sp), y = blk >> lw, x = blk & (w - 1);
      const int right = x > 0 ? lv[blk - 1] : 0, below = y > 0 ? lv[blk - w] : 0;
      const unsigned a = (unsigned) ts_mod_coeff(right, below, iabs(lv[blk]));
      if (a >= 10) enc_rem_abs<WR>(cb, (a - 10) >> 1, (unsigned) ts_rice(right, below));
    }
  }
#undef TS_BIN
}
// sum |xTransformSkip(org - pred)| scaled as TrQuant::transformNxN (1049-1124) scales the transform-skip entry of its pruning; optionally the coefficients to coef_out
__device__ inline int wave_ts_fwd(const int16_t *org, const int16_t *pred, int16_t *coef_out, int w, int h, int bd, int lane)
{
  const int P = w * h, sh = ts_shift(w, h, bd);
  int sa = 0;
  for (int e = lane; e < P; e += 64) { const int c = ts_scale(org[e] - pred[e], sh); if (coef_out) coef_out[e] = (int16_t) c; sa += iabs(c); }
  sa = wave_sum_i32(sa);
  const double scale = ((ilog2i(w) + ilog2i(h)) & 1) ? 1.0 / 1.414213562 : 1.0;
  return (int) ((double) sa * scale);
}
// decoder half of a transform-skip block by one wave: Quant::dequant, xITransformSkip, reconstruction over the prediction in rec, SSE against org.
// raw: rec receives the bare residual (leaf test).  cbf 0: the prediction is the reconstruction.
__device__ __noinline__ void wave_ts_recon(const int16_t *org, int16_t *rec, const int16_t *lev, int w, int h, int bd, int qp, int cbf, int lane, unsigned long long &sse_out, int raw = 0)
{
  org = uni_p(org); rec = uni_p(rec); lev = uni_p(lev); raw = uni(raw);
  w = uni(w); h = uni(h); bd = uni(bd); qp = uni(qp); cbf = uni(cbf);
  const int P = w * h, q = ts_qp(qp), sh = ts_shift(w, h, bd), scale = L.t.iqscale[q % 6], right_shift = 6 - (sh + q / 6), mx = (1 << bd) - 1;
  int tbd = 32 + right_shift - 7; if (tbd > 16) tbd = 16;
  const int in_min = -(1 << (tbd - 1)), in_max = (1 << (tbd - 1)) - 1;
  unsigned long long sse = 0;
  for (int e = lane; e < P; e += 64) {
    int r = 0;
    if (cbf) {
      int l = lev[e]; l = l < in_min ? in_min : l > in_max ? in_max : l;
      int v = right_shift > 0 ? (l * scale + (1 << (right_shift - 1))) >> right_shift : (l * scale) * (1 << -right_shift);
      v = v < -32768 ? -32768 : v > 32767 ? 32767 : v;
      r = (int) (int16_t) (sh >= 0 ? (v + (sh == 0 ? 0 : 1 << (sh - 1))) >> sh : v * (1 << -sh));
    }
    if (raw) { rec[e] = (int16_t) r; continue; }
    int v = rec[e] + r; v = v < 0 ? 0 : v > mx ? mx : v;
    rec[e] = (int16_t) v;
    const int d = org[e] - v; sse += (unsigned long long) (d * d);
  }
  wave_sync();
  sse_out = wave_sum_u64(sse);
}

// ------------------------------------------------------------------------------------------------ parallel operations
// candidate slots: nrec = reconstruction samples held (w*h luma, 2*cw*ch chroma).  Slot 0 of blocks up to 1024 samples is
// in LDS; slot 1 (only used when a wave evaluates more than one full-RD candidate) and bigger blocks are in HBM scratch.
// original tile of the node: LDS for nodes of at most BUF samples, else the stream's HBM scratch
__device__ inline int16_t *org_tile(uint8_t *scratch, int nrec) { return nrec <= BUF ? L.org : (int16_t *) (scratch + VXD_OFF_ORG); }
__device__ inline int16_t *slot_rec(uint8_t *scratch, int nrec, int wave, int which)
{ return (nrec <= BUF && which == 0) ? &L.wm[wave].slot[0] : (int16_t *) (scratch + VXD_OFF_SLOTS) + (wave * 2 + which) * VXD_SLOT_ELEMS; }
__device__ inline int16_t *slot_lev(uint8_t *scratch, int nrec, int wave, int which)
{ return (nrec <= BUF && which == 0) ? &L.wm[wave].slot[BUF] : (int16_t *) (scratch + VXD_OFF_SLOTS) + (wave * 2 + which) * VXD_SLOT_ELEMS + 4096; }
__device__ inline int32_t *wave_tmp(uint8_t *scratch, int n_i32, int wave)
{ return n_i32 <= BUF ? L.wm[wave].tmp : (int32_t *) (scratch + VXD_OFF_TMP) + wave * 2048; }

__device__ void ctx_copy_all(Ctx *dst, const Ctx *src)
{
  uint32_t *d = (uint32_t *) dst; const uint32_t *s = (const uint32_t *) src;
  for (int i = VTX; i < NCTX; i += NT) d[i] = s[i];       // s0 and s1 are 2*NCTX uint16 = NCTX uint32
}
__device__ Ctx *ctx_ptr(uint8_t *scratch, int which, int d, int wave)
{
  if (which == CTX_CUR) return &L.ctxs[CI_CUR];
  if (which == CTX_WAVE) return &L.ctxs[CI_W(wave)];
  return (Ctx *) (scratch + VXD_OFF_CTX + (size_t) (d * 2 + (which == CTX_BEST ? 1 : 0)) * VXD_CTXSNAP);
}

// ------------------------------------------------------------------------------------------------ ISP (intra sub-partitions)
// CU::getISPSplitDim (CL/UnitTools.cpp:437-459): size of a sub-partition along the split direction: a quarter of the side, but at least 16 samples per sub-partition
__device__ inline int isp_split_dim(int w, int h, int hor)
{
  const int split = hor ? h : w, non = hor ? w : h;
  const int factor = non < 16 ? 16 >> ilog2i(non) : 1;
  return imax(split >> 2, factor);
}
// HBM tiles of the ISP evaluation: per wave rec | lev (CU tiles, stride w), the CU's prediction, the dense coefficient tile of the sub-partition in work
__device__ inline int16_t *isp_tile(uint8_t *scratch, int wave) { return (int16_t *) (scratch + VXD_OFF_ISP) + wave * VXD_ISP_WAVE; }
// IntraSearch::xIntraCodingLumaISP (EL/IntraSearch.cpp:3171-3280) of one (mode, split) candidate by ONE wave: the sub-partitions are predicted from the reconstruction
// of the ones before (initIntraPatternChTypeISP, CL/IntraPrediction.cpp:1092-1199; prediction regions of at least four columns, JVET_O0106), each through
// wave_code_block_isp against the live contexts L.ctxs[ci] (copied from the node's start contexts here), the rate of xGetIntraFracBitsQT per sub-partition and the
// early exits against `limit` (3206-3245).  What the wave reports (out) is the raw material of that logic per sub-partition: the controller replays it (ctrl_isp_replay), so
// that candidates can be evaluated ahead of the reference's order under a limit that is at least the one they will be judged with.
// given != nullptr (out == nullptr): the levels (CU tile, stride w) and cbfs are taken as coded (DecCu::xIntraRecQT of an ISP CU for xReuseCachedResult); distortion ->
// L.isp_dist.  The node's base references are L.refs[0]; the region's go to the calling wave's LDS candidate slot (unused by this path).  tile: isp_tile() of a wave.
// LDSP (search path, CUs of at most 128 samples, refs_wave == the calling wave): all tiles are LDS objects addressed as such
template <bool LDSP>
__device__ __noinline__ void isp_code_cu(uint8_t *scratch, int w, int h, int dir, int isp, double limit, const int16_t *given, int given_tucbf, int16_t *rec, int16_t *lev, int ci, int lane,
                                         int16_t *tile, IspRes *out, int refs_wave)
{
  const VxParams &p = L.par;
  w = uni(w); h = uni(h); dir = uni(dir); isp = uni(isp); given_tucbf = uni(given_tucbf); ci = uni(ci);
  scratch = uni_p(scratch); given = uni_p(given); rec = uni_p(rec); lev = uni_p(lev); tile = uni_p(tile); out = uni_p(out); limit = uni_d(limit);
  const int bd = uni(p.bit_depth), hor = isp == 1;
  const int psz = isp_split_dim(w, h, hor), tw = hor ? w : psz, th = hor ? psz : h, n = hor ? h / psz : w / psz;
  const int predRegDiff = !hor && ((w == 8 && h > 4) || w == 4);         // CU::isPredRegDiffFromTB
  const int wave = uni(VTX >> 6), x0 = uni(L.nx), y0 = uni(L.ny);
  const int16_t *org = LDSP ? L.org : org_tile(scratch, w * h);
  // CUs of at most 128 samples (most of the nodes that test ISP) keep everything in LDS: the region references, the CU's reconstruction and levels in the candidate slot,
  // the prediction and the dense coefficient tile behind the transform scratch (a TU has at most 64 coefficients then); the tiles are copied out at the end
  const int small = LDSP || w * h <= 128;
  int16_t *const rec_out = rec, *const lev_out = lev;
  int16_t *sl = L.wm[uni(refs_wave)].slot, *tl = (int16_t *) L.wm[wave].tmp;
  if (small) { rec = sl + 132; lev = sl + 260; }
  int16_t *pred = small ? tl + 192 : tile + 8192, *cf = small ? tl + 128 : tile + 12288;
  int32_t *tmp = LDSP ? L.wm[wave].tmp : wave_tmp(scratch, imin(32, tw) * th, wave);
  const int16_t *bt = L.refs[0][0], *bl = L.refs[0][1];
  int16_t *rt = sl, *rl = sl + (small ? 66 : 160);          // region references: an LDS candidate slot this path does not use otherwise
  { uint32_t *d = (uint32_t *) &L.ctxs[ci]; const uint32_t *s_ = (const uint32_t *) &L.ctxs[CI_CUR]; for (int e = lane; e < NCTX; e += 64) d[e] = s_[e]; }
  wave_sync();
  double cost = 0; int tucbf = 0, nrun = 0;
  unsigned long long dist = 0, bits = 0;
  Cab cb; cb.ci = ci; cb.bits = 0;
  ISP_T(c0);
  for (int k = 0; k < n; k++) {
    const int ox = hor ? 0 : k * tw, oy = hor ? k * th : 0;
    ISP_T(k0);
    if (!predRegDiff || (ox & 3) == 0) {
      const int pw = predRegDiff ? imax(tw, 4) : tw, ph = th, topLen = w + pw, leftLen = h + ph;
      // reference samples of the region
      if (!ox && !oy) {
        for (int i = lane; i <= topLen; i += 64) rt[i] = bt[i];
        for (int j = lane; j <= leftLen; j += 64) rl[j] = bl[j];
      } else if (hor) {                        // the row above is the previous sub-partition's last reconstructed row, replicated to the right
        const int leftDecomp = x0 > 0;         // cs.isDecomp of the sample left of the sub-partition
        for (int j = lane; j <= leftLen; j += 64) rl[j] = leftDecomp ? bl[oy + j] : rec[(oy - 1) * w];
        for (int i = lane; i <= topLen; i += 64) rt[i] = i == 0 ? (leftDecomp ? bl[oy] : rec[(oy - 1) * w]) : rec[(oy - 1) * w + imin(i, pw) - 1];
      } else {                                 // columns
        const int aboveDecomp = y0 > 0;
        for (int i = lane; i <= topLen; i += 64) rt[i] = aboveDecomp ? bt[ox + i] : rec[ox - 1];
        for (int j = lane; j <= leftLen; j += 64) rl[j] = j == 0 ? (aboveDecomp ? bt[ox] : rec[ox - 1]) : rec[(imin(j, ph) - 1) * w + ox - 1];
      }
      wave_sync();
      Ipa ip; init_pred_params_isp(w, h, pw, ph, dir, ip);
      int dcv = 0;
      if (dir == DC) { if (lane == 0) dcv = dc_value(rt, rl, pw, ph, 0); dcv = __builtin_amdgcn_readlane(dcv, 0); }
      const int lpw = ilog2i(pw);
      for (int e = lane; e < pw * ph; e += 64) { const int py = e >> lpw, px = e & (pw - 1); pred[(oy + py) * w + ox + px] = (int16_t) pred_sample(rt, rl, pw, ph, px, py, ip, dir, 1, bd, dcv, topLen, leftLen); }
      wave_sync();
    }
    const int ltw = ilog2i(tw);
    for (int e = lane; e < tw * th; e += 64) { const int a = (oy + (e >> ltw)) * w + ox + (e & (tw - 1)); rec[a] = pred[a]; if (given) lev[a] = given[a]; }
    wave_sync();
    const int lastInferred = k == n - 1 && !tucbf, prevCbf = k ? (tucbf >> (k - 1)) & 1 : 0;
    const int cbfCtx = lastInferred ? -1 : (int) VX_CTX_QtCbf[0] + 2 + prevCbf;
    unsigned long long d; int cbf;
    const int off = oy * w + ox;
    { ISP_T(k1); ISP_ADD(19, k0, k1); ISP_ADD(23, 0, 1); }
    wave_code_block_isp<LDSP>(org + off, rec + off, lev + off, w, tmp, cf, scratch, tw, th, bd, p.qp_tr, lane, d, cbf, given ? (given_tucbf >> k) & 1 : -1, ci, cbfCtx);
    cbf = uni(cbf);
    nrun = k + 1;
    if (given) { if (cbf) tucbf |= 1 << k; dist += d; continue; }
    if (lane == 0) { out->d[k] = d; out->fb[k] = 0; }
    if (k == n - 1 && !tucbf && !cbf) break;                  // 2990-2996: ISP needs one coded sub-partition
    if (cbf) tucbf |= 1 << k;
    unsigned long long fb = 0;
    ISP_T(k2);
    if (!(rd_cost(p, bits, dist + d) > limit)) {              // 3206-3210: beyond the limit the rate is not even computed
      cb.bits = 0;                                            // xGetIntraFracBitsQT: the CU header with the first sub-partition, cbf unless inferred, coefficients
      if (lane == 0) {
        if (k == 0) enc_intra_luma_pred_mode(cb, L.ny, dir, 0, isp);
        if (!lastInferred) enc_bin(cb, (unsigned) cbf, VX_CTX_QtCbf[0] + 2 + prevCbf);
      }
      if (cbf) { if (LDSP) residual_coding_wave<true>(cb, (int) (cf - (L.wm[wave].slot + BUF)), nullptr, tw, th, 0, lane); else residual_coding_wave<false>(cb, 0, cf, tw, th, 0, lane); }
      { unsigned lo = (unsigned) cb.bits, hi = (unsigned) (cb.bits >> 32); lo = (unsigned) __builtin_amdgcn_readlane((int) lo, 0); hi = (unsigned) __builtin_amdgcn_readlane((int) hi, 0); fb = ((unsigned long long) hi << 32) | lo; }
      if (lane == 0) out->fb[k] = fb;
    }
    { ISP_T(k3); ISP_ADD(20, k2, k3); }
    cost += rd_cost(p, fb, d); dist += d; bits += fb;
    if (k + 1 < n) {
      if (cost > limit) break;
      const double thr = n == 2 ? 0.95 : k + 1 == 1 ? 0.83 : 0.91;
      if (cost > limit * thr) break;
    }
  }
  if (lane == 0) { if (given) L.isp_dist = dist; else { out->cbf = (uint8_t) tucbf; out->nrun = (uint8_t) nrun; } }
  wave_sync();
  if (small) { for (int e = lane; e < w * h; e += 64) { rec_out[e] = rec[e]; lev_out[e] = lev[e]; } wave_sync(); }
  { ISP_T(c1); ISP_ADD(21, c0, c1); ISP_ADD(22, 0, 1); }
}
// OP_ISP: up to NW candidates of the node's ISP test at once, one per wave: L.isp_res[w].mode / split name wave w's, all under the limit L.isp_limit
__device__ __noinline__ void op_isp(uint8_t *scratch)
{
  scratch = uni_p(scratch);
  const int wave = uni(VTX >> 6), lane = VTX & 63;
  const int w = uni(L.nw), h = uni(L.nh);
  if (wave < uni((int) L.isp_nb)) {
    int16_t *t = isp_tile(scratch, wave);
    if (w * h <= 128) isp_code_cu<true>(scratch, w, h, uni((int) L.isp_res[wave].mode), uni((int) L.isp_res[wave].split), uni_d(L.isp_limit), nullptr, 0, t, t + 4096, CI_W(wave), lane, t, &L.isp_res[wave], wave);
    else isp_code_cu<false>(scratch, w, h, uni((int) L.isp_res[wave].mode), uni((int) L.isp_res[wave].split), uni_d(L.isp_limit), nullptr, 0, t, t + 4096, CI_W(wave), lane, t, &L.isp_res[wave], wave);
  }
  __threadfence_block();
  __syncthreads();
}
// OP_ISP_PARK: the candidate wave L.isp_park evaluated became the node's best: its reconstruction, levels and end contexts are kept (1308-1326)
__device__ __noinline__ void op_isp_park(uint8_t *scratch)
{
  scratch = uni_p(scratch);
  const int src = uni((int) L.isp_park), P = uni(L.nw) * uni(L.nh);
  const int16_t *t = isp_tile(scratch, src);
  int16_t *best = (int16_t *) (scratch + VXD_OFF_ISP_BEST);
  for (int e = VTX; e < P; e += NT) { best[e] = t[e]; best[4096 + e] = t[4096 + e]; }
  ctx_copy_all(ctx_ptr(scratch, CTX_BEST, MAXD + 1, 0), &L.ctxs[CI_W(src)]);
  __threadfence_block();
  __syncthreads();
}

// OP_LUMA_PREP: stage the node's original luma tile and its reference samples (mrl 0,1,3) in LDS
template <typename T>
__device__ __noinline__ void op_luma_prep(const VxParams &p_, const VxFrameDev &fd_)
{
  const VxParams &p = L.par; (void) p_; const VxFrameDev &fd = L.fdv; (void) fd_;
  const int x = uni(L.nx), y = uni(L.ny), w = uni(L.nw), h = uni(L.nh);
  const long long ts = STAMP();
  const void *org = fd.org[0]; const int st = fd.stride[0];
  int16_t *ot = org_tile(p.scratch + (size_t) blockIdx.x * p.scratch_per_stream, w * h);
  for (int i = VTX; i < w * h; i += NT) { const int r = i >> ilog2i(w), c = i & (w - 1); ot[i] = (int16_t) ld_px<T>(org, (y + r) * st + x + c); }
  const int nsets = ((y & 127) == 0 || !(p.tools & 1)) ? 1 : 3;
  build_refs<T>(p, fd, 0, x, y, w, h, uni(L.cur_tile), nsets);
  if (VTX < 4) {
    const int s = VTX;           // dc per set (filtered set never used for DC)
    if (s != 1 && (s == 0 || nsets == 3)) L.dc_val[s] = dc_value(L.refs[s][0], L.refs[s][1], w, h, s == 0 ? 0 : s == 2 ? 1 : 3);
  }
  __syncthreads();
  if (VVCX_STAMP && VTX == 0) PROF(14) += (unsigned long long) (STAMP() - ts);
}
__device__ inline int luma_set(int mrl, int filt) { return mrl == 0 ? (filt ? 1 : 0) : (mrl == 1 ? 2 : 3); }

// SATD stage for blocks of at most 32 samples (4x4, 8x4, 4x8): the block is exactly one Hadamard tile, so 64/P
// candidates are packed into one wavefront (one lane per predicted sample of one candidate).
template <int BW, int BH>
__device__ void stage_a_small(const VxParams &p, int wave, int lane, int c_begin, int c_end)
{
  constexpr int P = BW * BH, G = 64 / P;
  const int sub = lane / P, pl = lane - sub * P;
  const int py = pl / BW, px = pl - py * BW;
  int16_t *pred = &L.wm[wave].slot[0];
  int16_t *scr = (int16_t *) L.wm[wave].tmp;
  const int bd = uni(p.bit_depth);
  // mode bits of all candidates of this wave up front, one candidate per lane: lane l serves step l / G, slot l % G
  unsigned long long mbits = 0;
  { const int cc = c_begin + wave * G + (lane / G) * (NW * G) + (lane % G); if (cc < c_end) mbits = luma_mode_bits(L.ctxs[CI_CUR], L.ny, L.cand[cc].mode, L.cand[cc].mrl); }
  int step = 0;
  for (int c0 = c_begin + wave * G; c0 < c_end; c0 += NW * G, step++) {
    const int c = c0 + sub;
    const bool valid = c < c_end;
    const int mode = valid ? L.cand[c].mode : 0, mrl = valid ? L.cand[c].mrl : 0;
    Ipa ip; ipa_unpack(L.cand_ipa[valid ? c : 0], ip);
    const int set = luma_set(mrl, ip.ref_filter);
    pred[lane] = (int16_t) pred_sample(L.refs[set][0], L.refs[set][1], BW, BH, px, py, ip, mode, 1, bd, L.dc_val[luma_set(mrl, 0)]);
    wave_sync();
    int sad = 0;
    if (lane < G * BH) {                                 // one row of one candidate per lane
      const int sr = lane / BH, r = lane - sr * BH;
      int v[BW];
      row_diff<BW>(L.org + r * BW, pred + sr * P + r * BW, v);
#pragma unroll
      for (int i = 0; i < BW; i++) sad += iabs(v[i]);
      had1d<BW>(v);
      row_store<BW>(scr + sr * P + r * BW, v);
    }
    sad = seg_sum<BH>(sad);
    wave_sync();
    int s = 0;
    if (lane < G * BW) {                                 // one column of one candidate per lane
      const int sc = lane / BW, i = lane - sc * BW;
      int v[BH];
#pragma unroll
      for (int r = 0; r < BH; r++) v[r] = scr[sc * P + r * BW + i];
      had1d<BH>(v);
#pragma unroll
      for (int r = 0; r < BH; r++) s += iabs(v[r]);
    }
    s = seg_sum<BW>(s);
    if (lane < G * BH && (lane % BH) == 0) L.acc[wave][lane / BH][0] = sad;
    if (lane < G * BW && (lane % BW) == 0)
      L.acc[wave][lane / BW][1] = (BW == 4 && BH == 4) ? (s + 1) >> 1 : (int) ((double) s / 0x1.6a09e667f3bcdp+2 * 2);   // CL/RdCost.cpp:2209,2662,2741
    wave_sync();
    if (lane / G == step && c0 + lane % G < c_end) {
      const int q = lane % G, cc = c0 + q;
      const unsigned long long sd = (unsigned long long) (unsigned) L.acc[wave][q][0], st = (unsigned long long) (unsigned) L.acc[wave][q][1];
      const unsigned long long msh = sd * 2 < st ? sd * 2 : st;
      const double a = (double) mbits * p.sqrt_lambda_fp;
      L.cand_cost[cc] = (double) msh + a;
    }
    wave_sync();
  }
}

// MIP prediction of the node by one wave into dst[0 .. w*h) from the unfiltered line-0 references (initIntraMip + predIntraMip,
// CL/IntraPrediction.cpp:2152-2186); the reduced boundary and the matrix outputs pass through the wave's LDS scratch
__device__ __noinline__ void wave_pred_mip(int16_t *dst, int w, int h, int mode, int bd, int wave, int lane)
{
  dst = uni_p(dst); bd = uni(bd);
  w = uni(w); h = uni(h); mode = uni(mode); wave = uni(wave);
  int *sh = (int *) L.wm[wave].tmp, *red = sh + 16;
  const MipGeo g = mip_geo(w, h);
  const int16_t *top = L.refs[0][0] + 1, *left = L.refs[0][1] + 1;
  mip_reduced_pred(top, left, g, mode, bd, lane, sh, red);
  wave_sync();
  for (int i = lane; i < w * h; i += 64) { const int py = i >> ilog2i(w), px = i & (w - 1); dst[i] = (int16_t) mip_sample(red, top, left, g, px, py); }
  wave_sync();
}
// SATD-stage costs of the MIP candidates L.cand[0 .. c_end) (EL/IntraSearch.cpp:703-745), one wave per candidate, every block size
template <bool SMALL>
__device__ __noinline__ void stage_a_mip(const VxParams &p, uint8_t *scratch, int wave, int lane, int w, int h, int c_end)
{
  scratch = uni_p(scratch); wave = uni(wave); w = uni(w); h = uni(h); c_end = uni(c_end);
  const int P = w * h, bd = uni(p.bit_depth);
  int16_t *pred = SMALL ? L.wm[wave].slot : slot_rec(scratch, P, wave, 0);
  int16_t *scr = SMALL ? (int16_t *) L.wm[wave].tmp : (int16_t *) wave_tmp(scratch, P >> 1, wave);
  for (int c = wave; c < c_end; c += NW) {
    const int mode = uni(L.cand[c].mode);
    wave_pred_mip(pred, w, h, mode, bd, wave, lane);
    unsigned long long sad, satd;
    wave_sad_satd<SMALL>(org_tile(scratch, P), pred, scr, w, h, lane, sad, satd);
    if (lane == 0) {
      const unsigned long long msh = sad * 2 < satd ? sad * 2 : satd;
      const unsigned long long mbits = luma_mode_bits(L.ctxs[CI_CUR], L.ny, mode, MIPF);
      const double a = (double) mbits * p.sqrt_lambda_fp;
      L.cand_cost[c] = (double) msh + a;
    }
    wave_sync();
  }
}
// SATD-stage loop of one wave over its candidates (blocks of more than 32 samples).  SMALL: prediction and Hadamard scratch in the
// wave's LDS buffers, else in HBM scratch.
template <bool SMALL>
__device__ void stage_a_loop(const VxParams &p, uint8_t *scratch, int wave, int lane, int w, int h, int c_end)
{
  scratch = uni_p(scratch); wave = uni(wave); w = uni(w); h = uni(h); c_end = uni(c_end);
  const int P = w * h, bd = uni(p.bit_depth);
  int16_t *pred = SMALL ? L.wm[wave].slot : slot_rec(scratch, P, wave, 0);
  int16_t *scr = SMALL ? (int16_t *) L.wm[wave].tmp : (int16_t *) wave_tmp(scratch, P >> 1, wave);
  unsigned long long mbits = 0;                         // mode bits of this wave's candidates up front, lane j serves step j
  { const int cc = uni(L.op_a) + wave + lane * NW; if (cc < c_end) mbits = luma_mode_bits(L.ctxs[CI_CUR], L.ny, L.cand[cc].mode, L.cand[cc].mrl); }
  int step = 0;
  for (int c = uni(L.op_a) + wave; c < c_end; c += NW, step++) {
    const int mode = uni(L.cand[c].mode), mrl = uni(L.cand[c].mrl);
    Ipa ip; { uint2 pk = L.cand_ipa[c]; pk.x = (unsigned) uni((int) pk.x); pk.y = (unsigned) uni((int) pk.y); ipa_unpack(pk, ip); }
    const int set = luma_set(mrl, ip.ref_filter);
    const int16_t *top = L.refs[set][0], *left = L.refs[set][1];
    const int dcv = L.dc_val[luma_set(mrl, 0)];
    const long long ta = STAMP();
    for (int i = lane; i < P; i += 64) { const int py = i >> ilog2i(w), px = i & (w - 1); pred[i] = (int16_t) pred_sample(top, left, w, h, px, py, ip, mode, 1, bd, dcv); }
    wave_sync();
    const long long tb = STAMP();
    unsigned long long sad, satd;
    wave_sad_satd<SMALL>(org_tile(scratch, P), pred, scr, w, h, lane, sad, satd);
    if (VVCX_STAMP && VTX == 0) { const long long tc = STAMP(); PROF(15) += (unsigned long long) (tb - ta); PROF(11) += (unsigned long long) (tc - tb); }
    if (lane == step) {
      const unsigned long long msh = sad * 2 < satd ? sad * 2 : satd;
      const double a = (double) mbits * p.sqrt_lambda_fp;
      L.cand_cost[c] = (double) msh + a;
    }
    wave_sync();
  }
}
// OP_STAGE_A: SATD-stage cost of every candidate in L.cand[op_a .. op_b) (EL/IntraSearch.cpp:489-682), one wave per candidate
__device__ __noinline__ void op_stage_a(const VxParams &p_, uint8_t *scratch)
{
  scratch = uni_p(scratch);
  const VxParams &p = L.par; (void) p_;
  const int wave = uni(VTX >> 6), lane = VTX & 63;
  const int w = uni(L.nw), h = uni(L.nh), P = w * h;
  const int c_end = uni(L.op_b);
  if (uni(L.op_c) == 2) {                                  // the MIP candidates: costs only, the control step merges them into the list
    if (P <= BUF) stage_a_mip<true>(p, scratch, wave, lane, w, h, c_end); else stage_a_mip<false>(p, scratch, wave, lane, w, h, c_end);
    __syncthreads();
    return;
  }
  { const int c = uni(L.op_a) + (int) VTX; if (c < c_end) { Ipa ip; init_pred_params(w, h, 1, L.cand[c].mode, L.cand[c].mrl, ip); L.cand_ipa[c] = ipa_pack(ip); } }
  __syncthreads();
  if (P <= 32) {
    if (w == 4 && h == 4) stage_a_small<4, 4>(p, wave, lane, uni(L.op_a), c_end);
    else if (w == 8) stage_a_small<8, 4>(p, wave, lane, uni(L.op_a), c_end);
    else stage_a_small<4, 8>(p, wave, lane, uni(L.op_a), c_end);
  } else if (P <= BUF) stage_a_loop<true>(p, scratch, wave, lane, w, h, c_end);
  else stage_a_loop<false>(p, scratch, wave, lane, w, h, c_end);
  __syncthreads();
  // updateCandList (CL/UnitTools.h:261-306) over a stream of candidates keeps the numRd cheapest with ties to the
  // earlier-inserted one = a stable selection.  Done in parallel: every candidate computes its rank.
  // Insertion order of the reference: 35 modes [0,35), then the +-1 refinements [n1, n2), then the MRL candidates [35, n1).
  {
    const int n1 = uni(L.n_cand), n2 = c_end, numRd = uni(L.S.numRd);
    const int first_phase = uni(L.op_c);
    const int n = first_phase ? 35 : n2;
    const int c = VTX;
    if (c < n) {
      const double mine = L.cand_cost[c];
      const int myseq = c < 35 ? c : (c < n1 ? 1000 + c : 500 + c);
      int rank = 0;
      for (int j = 0; j < n; j++) {
        const double v = L.cand_cost[j];
        const int sj = j < 35 ? j : (j < n1 ? 1000 + j : 500 + j);
        rank += (v < mine) || (v == mine && sj < myseq);
      }
      if (rank < numRd) { L.S.rdList[rank] = L.cand[c]; L.S.rdCost[rank] = mine; }
      if (!first_phase && uni((int) L.S.testIsp) && !(c >= 35 && c < n1)) {      // 624-632: the list as it stands before the MRL candidates join is what ISP falls back on
        int r2 = 0;
        for (int j = 0; j < n; j++) if (!(j >= 35 && j < n1)) {
          const double v = L.cand_cost[j];
          const int sj = j < 35 ? j : 500 + j;
          r2 += (v < mine) || (v == mine && sj < myseq);
        }
        if (r2 < numRd) L.S.ispHad[r2] = L.cand[c].mode;
      }
    }
    if (c == 0) { L.S.rdSize = imin(numRd, n); if (!first_phase && L.S.testIsp) L.S.ispHadN = (int8_t) imin(numRd, n - (n1 - 35)); }
  }
  __syncthreads();
}

// Candidate buffers.  Blocks of at most 1024 samples (SMALL) are evaluated in the wave's LDS buffers; a candidate that becomes the
// wave's best is parked in the wave's HBM slot 1 (a streaming copy), so the next candidate again runs out of LDS.  Bigger blocks
// alternate between the wave's two HBM slots.
// OP_STAGE_B: full RD of L.rd[0..n_rd) (EL/IntraSearch.cpp:1158-1358 → xRecurIntraCodingLumaQT → xIntraCodingTUBlock)
template <bool SMALL, bool MTS>
__device__ void stage_b_loop(const VxParams &p, uint8_t *scratch, int wave, int lane, int w, int h)
{
  scratch = uni_p(scratch); wave = uni(wave); w = uni(w); h = uni(h);
  const int P = w * h, bd = uni(p.bit_depth);
  if (lane == 0) L.wave_best[wave] = -1;
  double wbest = MAX_DOUBLE;
  int cur = 0;                                    // !SMALL: slot being written; the other one holds the wave's best so far
  int nmts = 0;                                   // MTS transform candidates this wave evaluated (work counter)
  const int n_rd = uni(L.n_rd);
  for (int c = wave; c < n_rd; c += NW) {
    const int mode = uni(L.rd[c].mode), mrl = uni(L.rd[c].mrl);
    int16_t *rec = SMALL ? L.wm[wave].slot : slot_rec(scratch, P, wave, cur), *lev = SMALL ? L.wm[wave].slot + BUF : slot_lev(scratch, P, wave, cur);
    if (lane == 0) L.rd_cost[c] = MAX_DOUBLE;
    const int mip = mrl & MIPF;                           // a MIP candidate: mode is the MIP mode, no reference line / filter choice
    Ipa ip; init_pred_params(w, h, 1, mip ? 0 : mode, mip ? 0 : mrl, ip);
    const int set = luma_set(mip ? 0 : mrl, ip.ref_filter);
    const int16_t *top = L.refs[set][0], *left = L.refs[set][1];
    const int dcv = L.dc_val[luma_set(mip ? 0 : mrl, 0)];
    const long long tb0 = STAMP();
    if (mip) wave_pred_mip(rec, w, h, mode, bd, wave, lane);
    else for (int i = lane; i < P; i += 64) { const int py = i >> ilog2i(w), px = i & (w - 1); rec[i] = (int16_t) pred_sample(top, left, w, h, px, py, ip, mode, 1, bd, dcv); }
    wave_sync();
    const long long tb1 = STAMP();
    // transform candidates of the TU (xRecurIntraCodingLumaQT 3340-3640 without LFNST / transform skip): DCT2 alone, or where MTS is
    // allowed {DCT2, 2, 3, 4, 5} pruned by the sum of absolute coefficients (TrQuant::transformNxN 1049-1124, MTSIntraMaxCand 3).
    // Every (mode, transform) pair is offered to the wave's best like a candidate of its own: the result is the two-level minimum.
    const int mtsOk = MTS && mts_allowed(p, w, h);        // MTS = false: the loop below is the single DCT-II pass
    unsigned test = 1; int sum0 = 0;
    double mbest = MAX_DOUBLE; int cbfDCT2 = 1;
    for (int k = 0; k < (MTS && mtsOk ? 5 : 1); k++) {
    if (!cbfDCT2) break;
    if (!((test >> k) & 1)) continue;
    const int mts = k ? k + 1 : 0;
    if (k) {                                             // the previous transform candidate turned the prediction into its reconstruction
      rec = SMALL ? L.wm[wave].slot : slot_rec(scratch, P, wave, cur); lev = SMALL ? L.wm[wave].slot + BUF : slot_lev(scratch, P, wave, cur);
      if (mip) wave_pred_mip(rec, w, h, mode, bd, wave, lane);
      else for (int i = lane; i < P; i += 64) { const int py = i >> ilog2i(w), px = i & (w - 1); rec[i] = (int16_t) pred_sample(top, left, w, h, px, py, ip, mode, 1, bd, dcv); }
      wave_sync();
    }
    if (MTS && k == 1) {
      // the pruning of the candidate list (the reference runs it inside the DCT-II pass; its outcome is only read when that pass left
      // a non-zero block, so it is computed here, after the pass, from the restored prediction; the DCT-II sum came with the pass)
      int sums[5]; sums[0] = sum0;
      for (int q = 1; q < 5; q++) sums[q] = wave_fwd_sumabs<SMALL>(org_tile(scratch, P), rec, wave_tmp(scratch, imin(w, 32) * h, wave), w, h, bd, q + 1, lane);
      const int ls = imax(ilog2i(w), ilog2i(h)) - 2;
      const double fac = ls == 0 ? 1.2 : ls <= 2 ? 1.3 : ls == 3 ? 1.4 : 1.5;
      const double thr = fac * (double) sums[0], thrTS = (double) sums[0];
      int numTests = 0; test = 0;
      for (int q = 0; q < 5; q++) { const int t = (double) sums[q] <= (q == 1 ? thrTS : thr) && numTests <= 3; test |= (unsigned) t << q; numTests += t; }
      test = uni((int) test);
      if (!((test >> 1) & 1)) continue;
    }
    unsigned long long sse; int cbf;
    if (k == 0 && !(MTS && mtsOk)) wave_code_block<SMALL>(org_tile(scratch, P), 0, 0, rec, lev, wave_tmp(scratch, imin(w, 32) * h, wave), w, h, bd, p.qp_tr, lane, sse, cbf);
    else if (k == 0) wave_code_block<SMALL, true>(org_tile(scratch, P), 0, 0, rec, lev, wave_tmp(scratch, imin(w, 32) * h, wave), w, h, bd, p.qp_tr, lane, sse, cbf, -1, &sum0);
    else wave_code_block_mts<SMALL>(org_tile(scratch, P), rec, lev, wave_tmp(scratch, imin(w, 32) * h, wave), w, h, bd, p.qp_tr, mts, lane, sse, cbf);
    if (k) nmts++;
    if (k == 0) { cbfDCT2 = uni(cbf); if (MTS && mtsOk) test = 0x1f; }      // k = 1 is entered to run the pruning
    if (k && !uni(cbf)) continue;                        // an MTS index is not coded for a zero block: forbidden (cost MAX_DOUBLE)
    const long long tb2 = STAMP();
    // xGetIntraFracBitsQT: header + cbf + residual from the node's start contexts
    { uint32_t *d = (uint32_t *) &L.ctxs[CI_W(wave)]; const uint32_t *s = (const uint32_t *) &L.ctxs[CI_CUR]; for (int i = lane; i < NCTX; i += 64) d[i] = s[i]; }
    wave_sync();
    double cost = 0;
    Cab cb; cb.ci = CI_W(wave); cb.bits = 0;
    if (lane == 0) {
      enc_intra_luma_pred_mode(cb, L.ny, mode, mrl);
      enc_bin(cb, (unsigned) cbf, VX_CTX_QtCbf[0]);
      if (cbf && mtsOk) enc_mts_idx(cb, mts);
    }
    if (uni(cbf)) residual_coding_wave<SMALL>(cb, 0, lev, w, h, 0, lane, mts > 1);
    if (lane == 0) cost = rd_cost(p, cb.bits, sse);
    if (VVCX_STAMP && VTX == 0) { const long long tb3 = STAMP(); PROF(28) += (unsigned long long) (tb1 - tb0); PROF(29) += (unsigned long long) (tb2 - tb1); PROF(31) += (unsigned long long) (tb3 - tb2); }
    cost = lane0_d(cost);
    if (cost < mbest) {
      mbest = cost;
      if (lane == 0) { L.rd_cost[c] = cost; L.rd_dist[c] = sse; L.rd_bits[c] = cb.bits; L.rd_cbf[c] = (uint8_t) cbf; L.rd_mts[c] = (uint8_t) mts; L.rd_wave[c] = (uint8_t) wave; }
    }
    if (cost < wbest) {
      wbest = cost;
      if (lane == 0) { L.wave_best[wave] = c; L.wave_slot[wave] = SMALL ? 1 : cur; }
      if (SMALL) { int16_t *pr = slot_rec(scratch, P, wave, 1), *pl = slot_lev(scratch, P, wave, 1); for (int i = lane; i < P; i += 64) { pr[i] = rec[i]; pl[i] = lev[i]; } }
      else cur ^= 1;
      // keep the end-of-candidate contexts of the wave's best: they are the CU's end contexts (same syntax as
      // cu_pred_data + cu_residual of xCheckRDCostIntra 2593-2619 for a luma-tree CU)
      { uint32_t *d = (uint32_t *) ctx_ptr(scratch, CTX_START, MAXD + wave, 0); const uint32_t *s = (const uint32_t *) &L.ctxs[CI_W(wave)]; for (int i = lane; i < NCTX; i += 64) d[i] = s[i]; }
    }
    wave_sync();
    }
  }
  if (lane == 0) L.mts_evals[wave] = nmts;
}
// out of line: the MTS variant must not weigh on the register allocation of the DCT-II-only loop inlined into op_stage_b
// (every variant out of line: the dispatcher below then needs few registers and saves few at its entry - it runs once per full-RD operation)
template <bool SMALL> __device__ __noinline__ void stage_b_loop_dct2(uint8_t *scratch, int wave, int lane, int w, int h) { stage_b_loop<SMALL, false>(L.par, scratch, wave, lane, w, h); }
template <bool SMALL> __device__ __noinline__ void stage_b_loop_mts(const VxParams &p_, uint8_t *scratch, int wave, int lane, int w, int h) { stage_b_loop<SMALL, true>(L.par, scratch, wave, lane, w, h); (void) p_; }
// ---- full-RD stage with the dependent quantiser: the trellis of a block is serial and four lanes wide, so the candidates of the node are taken through the
// stage together, in rounds (all threads; every wave meets the same barriers):
//   A1  per candidate (one wave each): prediction, forward DCT-II -> coefficient pool (HBM scratch), prediction -> prediction pool
//   A2  the trellis of up to 16 candidates per wavefront side by side (wave_depquant_batch)
//   A3  per candidate: dequantise, inverse transform, reconstruction, SSE, rate; MTS pruning of a candidate whose DCT-II block is non-zero
//   B1-B3 the same for the (candidate, explicit MTS pair) items that survived the pruning.
// A wave keeps the best (cost, candidate, transform) item it evaluated parked like stage_b_loop does: the winner of the reference's two nested strict-<
// loops is the lexicographic minimum of (cost, list position, transform order), which is the minimum of one of the waves.
template <bool SMALL>
__device__ void dq_trellis_phase(uint8_t *scratch, int n, int P, int total, int w, int h, int zo, int wave, int lane, int lfnst = 0, int item0 = 0, int abs0 = 0, int wave_shift = 0)
{
  // items item0 .. item0 + n of the pool, absSum -> L.dq_abs[abs0 ..]; wave_shift rotates which wave takes the first chunk (two batches of one operation run side by side)
  int16_t *poolCoef = (int16_t *) (scratch + VXD_OFF_POOL_COEF) + (size_t) item0 * P; uint8_t *poolNodes = scratch + VXD_OFF_POOL_NODES + (size_t) item0 * 4 * total;
  const int ipw = imin(16, (int) sizeof(WaveMem) / (240 + 2 * total));          // items one wave can hold decisions for
  for (int i0 = ((wave + NW - wave_shift) % NW) * ipw; i0 < n; i0 += NW * ipw)
    wave_depquant_batch<2>(imin(ipw, n - i0), poolCoef + (size_t) i0 * P, P, poolNodes + (size_t) i0 * 4 * total, 4 * total, wave, abs0 + i0,
                        CI_CUR, 0, VX_CTX_QtCbf[0], 0u, w, h, 0, zo, lfnst, lane);
}
template <bool SMALL>
__device__ __noinline__ void stage_b_rounds(uint8_t *scratch, int wave, int lane, int w, int h)
{
  scratch = uni_p(scratch); wave = uni(wave); w = uni(w); h = uni(h);      // uniform arguments arrive in vector registers: scalar from here on
  const VxParams &p = L.par;
  const int P = w * h, bd = uni(p.bit_depth), total = imin(32, w) * imin(32, h);
  const int mtsOk = mts_allowed(p, w, h);
  // LFNST on: one transform per pass for every candidate (DCT-II with the pass's LFNST kernel, or the pass's MTS pair), no per-block MTS pruning
  const int lfOn = uni((int) (p.tools & TOOL_LFNST)) != 0, psLf = lfOn ? uni((int) L.ps_lfnst) : 0, psMts = lfOn ? uni((int) L.ps_mts) : 0, psGrp = uni((int) L.ps_grp);
  const int pruneOk = mtsOk && !lfOn;
  // transform skip: in the pass without LFNST and MTS every candidate's TU also tries MTS_SKIP (xRecurIntraCodingLumaQT 3349-3367, 3505-3517) unless the pruning drops it
  const int tsOn = lfOn && !psLf && !psMts && ts_allowed(p, w, h);
  int16_t *poolPred = (int16_t *) (scratch + VXD_OFF_POOL), *poolCoef = (int16_t *) (scratch + VXD_OFF_POOL_COEF);
  VxRbItem *recA = (VxRbItem *) (scratch + VXD_OFF_POOL_REC), *recB = recA + VXD_POOL_ITEMS;
  const int capItems = imin(VXD_POOL_ITEMS, imin(VXD_POOL_ELEMS / P, VXD_POOL_NODE_BYTES / (4 * total)));
  const int capCand = imin(16, pruneOk ? imax(1, capItems / 3) : capItems);
  const int n_rd = uni(L.n_rd);
  // LFNST on: the DCT-II pass of a node that also gets the DST-VII pass (transform group 0) prepares that pass - forward DST-VII of every candidate beside its
  // DCT-II, the two trellis batches side by side on different waves (the serial chain is what a pass costs; the contexts both start from are the node's) - and
  // the DST-VII pass, whose candidate list is a subset chosen from this pass's costs, only reconstructs and prices the blocks it needs
  const int specGen = lfOn && !psLf && !psMts && mtsOk && 2 * n_rd <= capItems && n_rd <= 16;
  // the same between the two LFNST passes: they walk the same candidate list with the same predictions and differ in the kernel of the set only, so the lfnstIdx 1 pass
  // (when lfnstIdx 2 is still in the loop bounds: it is dropped only if this pass's winner has no coefficients) also transforms with kernel 2 and runs both trellis batches
  const int specGen2 = lfOn && psLf == 1 && !psMts && uni((int) L.S.endLf) >= 2 && 2 * n_rd <= capItems && n_rd <= 16;
  const int specKind = uni((int) L.spec_kind);
  const int specUse = lfOn && uni((int) L.spec_n) > 0 && ((psMts && psGrp == 0 && specKind == 1) || (psLf == 2 && !psMts && specKind == 2 && uni((int) L.spec_n) == n_rd));
  const int specN = uni((int) L.spec_n);
  // transform skip, early form: the pruning and the RDOQ-TS chains (rounds T1 / T2) need only the predictions, the original and the DCT-II sums of round A1, and the
  // waves 2 and 3 have nothing to do while the trellis batches of the DCT-II and the DST-VII blocks run on waves 0 and 1: they take T1 and T2 there, each for the candidates
  // it owns (no hand-over between them), into coefficient-pool items and absSum slots of their own (48 + candidate)
  const int tsEarly = tsOn && NW >= 4 && n_rd <= 16 && (48 + 16) * P <= VXD_POOL_ELEMS, tsAt = tsEarly ? 48 : 0;
  if (lane == 0) L.wave_best[wave] = -1;
  if (tsOn) ts_build_tables();
  dq_build_tables(0);
  double wbest = MAX_DOUBLE; int wkey = 1 << 30;         // the wave's best item so far: cost, and candidate * 8 + transform order as the tie break
  int cur = 0, nmts = 0;
  const int16_t *org = org_tile(scratch, P);
  for (int c0 = 0; c0 < n_rd; c0 += capCand) {
    const int nA = imin(capCand, n_rd - c0);
    const long long q0 = STAMP();
    // ---- A1
    if (!specUse) for (int i = wave; i < nA; i += NW) {
      const int c = c0 + i, mode = uni(L.rd[c].mode), mrl = uni(L.rd[c].mrl), mip = mrl & MIPF;
      int16_t *rec = SMALL ? L.wm[wave].slot : slot_rec(scratch, P, wave, cur), *lev = SMALL ? L.wm[wave].slot + BUF : slot_lev(scratch, P, wave, cur);
      if (mip) wave_pred_mip(rec, w, h, mode, bd, wave, lane);
      else {
        Ipa ip; init_pred_params(w, h, 1, mode, mrl, ip);
        const int set = luma_set(mrl, ip.ref_filter), dcv = L.dc_val[luma_set(mrl, 0)];
        const int16_t *top = L.refs[set][0], *left = L.refs[set][1];
        for (int e = lane; e < P; e += 64) { const int py = e >> ilog2i(w), px = e & (w - 1); rec[e] = (int16_t) pred_sample(top, left, w, h, px, py, ip, mode, 1, bd, dcv); }
      }
      wave_sync();
      for (int e = lane; e < P; e += 64) poolPred[(size_t) i * P + e] = rec[e];
      unsigned long long sse; int cbf, sum0 = 0;
      const int mtsC = psMts ? pass_mts_idx(psGrp, mode) : 0;
      if (mtsC) wave_code_block_mts<SMALL>(org, rec, lev, wave_tmp(scratch, imin(w, 32) * h, wave), w, h, bd, p.qp_tr, mtsC, lane, sse, cbf, -2);
      else wave_code_block<SMALL, true>(org, 0, 0, rec, lev, wave_tmp(scratch, imin(w, 32) * h, wave), w, h, bd, p.qp_tr, lane, sse, cbf, -2, &sum0, 0, 0, 0, psLf, psLf ? lfnst_mode(mip ? PLANAR : mode, w, h) : 0);
      for (int e = lane; e < P; e += 64) poolCoef[(size_t) i * P + e] = lev[e];
      if (lane == 0) { recA[i].sum0 = sum0; recA[i].test = 0; recB[i].sum0 = sum0; }      // recB: the DCT-II sum survives A3 for the transform-skip pruning
      if (specGen) {                                        // rec still holds the prediction (forward-only calls leave it alone)
        wave_code_block_mts<SMALL>(org, rec, lev, wave_tmp(scratch, imin(w, 32) * h, wave), w, h, bd, p.qp_tr, 2, lane, sse, cbf, -2);
        for (int e = lane; e < P; e += 64) poolCoef[(size_t) (nA + i) * P + e] = lev[e];
      }
      if (specGen2) {
        wave_code_block<SMALL>(org, 0, 0, rec, lev, wave_tmp(scratch, imin(w, 32) * h, wave), w, h, bd, p.qp_tr, lane, sse, cbf, -2, nullptr, 0, 0, 0, 2, lfnst_mode(mip ? PLANAR : mode, w, h));
        for (int e = lane; e < P; e += 64) poolCoef[(size_t) (nA + i) * P + e] = lev[e];
      }
    }
    __threadfence_block();
    __syncthreads();
    const long long q1 = STAMP();
    if (!specUse) dq_trellis_phase<SMALL>(scratch, nA, P, total, w, h, psMts, wave, lane, psLf);
    if (specGen) dq_trellis_phase<SMALL>(scratch, nA, P, total, w, h, 1, wave, lane, 0, nA, 32, 1);
    if (specGen2) dq_trellis_phase<SMALL>(scratch, nA, P, total, w, h, 0, wave, lane, 2, nA, 32, 1);
    if (tsEarly && wave >= 2) {
      for (int i = wave - 2; i < nA; i += 2) {
        const int sa = wave_ts_fwd(org, poolPred + (size_t) i * P, poolCoef + (size_t) (48 + i) * P, w, h, bd, lane);
        if (lane == 0) L.rb_pairs[i] = (uint8_t) ((double) sa <= (double) recB[i].sum0);
      }
      __threadfence_block();
      wave_sync();
      int cnt = 0, mine = -1;
      for (int i = wave - 2; i < nA; i += 2) if (L.rb_pairs[i]) { if (cnt == lane) mine = i; cnt++; }
      if (mine >= 0) L.dq_abs[48 + mine] = ts_rdoq_lane(poolCoef + (size_t) (48 + mine) * P, w, h, bd, p.qp_tr);
    }
    __threadfence_block();
    __syncthreads();
    const long long q2 = STAMP();
    if (specGen || specGen2) { if ((int) VTX < nA) L.spec_abs[VTX] = L.dq_abs[32 + VTX]; if (VTX == 0) { L.spec_n = (int8_t) nA; L.spec_kind = (int8_t) (specGen ? 1 : 2); } }      // read by the DST-VII pass / the lfnstIdx 2 pass
    else if (VTX == 0) L.spec_n = 0;                        // consumed (or not valid for what follows)
    // ---- A3
    for (int i = wave; i < nA; i += NW) {
      const int c = c0 + i, mode = uni(L.rd[c].mode), mrl = uni(L.rd[c].mrl);
      int16_t *rec = SMALL ? L.wm[wave].slot : slot_rec(scratch, P, wave, cur), *lev = SMALL ? L.wm[wave].slot + BUF : slot_lev(scratch, P, wave, cur);
      const int mtsC = psMts ? pass_mts_idx(psGrp, mode) : 0;
      const int src = specUse ? (specKind == 2 ? i : uni((int) L.rd_src[c])) : i;          // a prepared block sits behind the specN blocks of the pass that made it (the lfnstIdx 2 pass walks the same list in the same order)
      for (int e = lane; e < P; e += 64) { rec[e] = poolPred[(size_t) src * P + e]; lev[e] = poolCoef[(size_t) (specUse ? specN + src : i) * P + e]; }
      wave_sync();
      const int cbf = (specUse ? uni(L.spec_abs[src]) : uni(L.dq_abs[i])) > 0;
      unsigned long long sse; int cbf2;
      if (mtsC) wave_code_block_mts<SMALL>(org, rec, lev, wave_tmp(scratch, imin(w, 32) * h, wave), w, h, bd, p.qp_tr, mtsC, lane, sse, cbf2, cbf);
      else wave_code_block<SMALL>(org, 0, 0, rec, lev, wave_tmp(scratch, imin(w, 32) * h, wave), w, h, bd, p.qp_tr, lane, sse, cbf2, cbf, nullptr, 0, 0, 0, psLf, psLf ? lfnst_mode((mrl & MIPF) ? PLANAR : mode, w, h) : 0);
      { uint32_t *d = (uint32_t *) &L.ctxs[CI_W(wave)]; const uint32_t *s = (const uint32_t *) &L.ctxs[CI_CUR]; for (int e = lane; e < NCTX; e += 64) d[e] = s[e]; }
      wave_sync();
      double cost = 0;
      Cab cb; cb.ci = CI_W(wave); cb.bits = 0;
      if (lane == 0) { enc_intra_luma_pred_mode(cb, L.ny, mode, mrl); enc_bin(cb, (unsigned) cbf, VX_CTX_QtCbf[0]); if (cbf) enc_tu_mts(cb, w, h, mtsC); }
      if (cbf) residual_coding_wave<SMALL>(cb, 0, lev, w, h, 0, lane, mtsC > 1);
      if (lane == 0) { cost = rd_cost(p, cb.bits, sse); recA[i].cost = cost; recA[i].dist = sse; recA[i].bits = cb.bits; recA[i].cbf = cbf; recA[i].wave = wave; if (lfOn) { recA[i].sum0 = (mtsC << 8) | (cbf ? lfnst_flags(L.rc_last[wave], w, h) : 0); recA[i].test = 0; } }
      cost = lane0_d(cost);
      if (cost < wbest || (cost == wbest && c * 8 < wkey)) {
        wbest = cost; wkey = c * 8;
        if (lane == 0) { L.wave_best[wave] = c; L.wave_slot[wave] = SMALL ? 1 : cur; }
        if (SMALL) { int16_t *pr = slot_rec(scratch, P, wave, 1), *pl = slot_lev(scratch, P, wave, 1); for (int e = lane; e < P; e += 64) { pr[e] = rec[e]; pl[e] = lev[e]; } }
        else cur ^= 1;
        { uint32_t *d = (uint32_t *) ctx_ptr(scratch, CTX_START, MAXD + wave, 0); const uint32_t *s = (const uint32_t *) &L.ctxs[CI_W(wave)]; for (int e = lane; e < NCTX; e += 64) d[e] = s[e]; }
      }
      wave_sync();
      if (pruneOk && cbf) {                                // TrQuant::transformNxN 1049-1124: which of the explicit MTS pairs stay, by their sums of absolute coefficients
        int16_t *pr = SMALL ? L.wm[wave].slot : slot_rec(scratch, P, wave, cur);
        for (int e = lane; e < P; e += 64) pr[e] = poolPred[(size_t) i * P + e];
        wave_sync();
        int sums[5]; sums[0] = recA[i].sum0;
        for (int q = 1; q < 5; q++) sums[q] = wave_fwd_sumabs<SMALL>(org, pr, wave_tmp(scratch, imin(w, 32) * h, wave), w, h, bd, q + 1, lane);
        const int ls = imax(ilog2i(w), ilog2i(h)) - 2;
        const double fac = ls == 0 ? 1.2 : ls <= 2 ? 1.3 : ls == 3 ? 1.4 : 1.5;
        const double thr = fac * (double) sums[0], thrTS = (double) sums[0];
        int numTests = 0; unsigned test = 0;
        for (int q = 0; q < 5; q++) { const int t = (double) sums[q] <= (q == 1 ? thrTS : thr) && numTests <= 3; test |= (unsigned) t << q; numTests += t; }
        if (lane == 0) recA[i].test = (int) (test & 0x1e);
      }
    }
    __threadfence_block();
    __syncthreads();
    if (tsOn && !tsEarly) {
      // ---- T1: which candidates keep the transform-skip candidate (sum of its scaled residual against the DCT-II sum, TrQuant::transformNxN 1049-1124); their
      // "coefficients" replace the consumed DCT-II levels in the coefficient pool
      for (int i = wave; i < nA; i += NW) {
        const int sa = wave_ts_fwd(org, poolPred + (size_t) i * P, poolCoef + (size_t) i * P, w, h, bd, lane);
        if (lane == 0) L.rb_pairs[i] = (uint8_t) ((double) sa <= (double) recB[i].sum0);      // keep flags (the list of MTS items is not built in this pass)
      }
      __threadfence_block();
      __syncthreads();
      // ---- T2: RDOQ-TS of the kept blocks, one lane each (every block starts from the node's contexts), spread over the waves
      {
        const int j = lane * NW + wave;
        int cnt = 0, mine = -1;
        for (int i = 0; i < nA; i++) if (L.rb_pairs[i]) { if (cnt == j) mine = i; cnt++; }
        if (mine >= 0) L.dq_abs[mine] = ts_rdoq_lane(poolCoef + (size_t) mine * P, w, h, bd, p.qp_tr);
      }
      __threadfence_block();
      __syncthreads();
    }
    if (tsOn) {
      // ---- T3: reconstruction, rate and cost of the non-empty ones (an empty transform-skip block is forbidden, 3567-3571); strict < against the candidate's DCT-II result
      for (int i = wave; i < nA; i += NW) {
        if (!uni((int) L.rb_pairs[i])) continue;
        nmts++;
        if (uni(L.dq_abs[tsAt + i]) <= 0) continue;
        const int c = c0 + i, mode = uni(L.rd[c].mode), mrl = uni(L.rd[c].mrl);
        int16_t *rec = SMALL ? L.wm[wave].slot : slot_rec(scratch, P, wave, cur), *lev = SMALL ? L.wm[wave].slot + BUF : slot_lev(scratch, P, wave, cur);
        for (int e = lane; e < P; e += 64) { rec[e] = poolPred[(size_t) i * P + e]; lev[e] = poolCoef[(size_t) (tsAt + i) * P + e]; }
        { uint32_t *d = (uint32_t *) &L.ctxs[CI_W(wave)]; const uint32_t *s = (const uint32_t *) &L.ctxs[CI_CUR]; for (int e = lane; e < NCTX; e += 64) d[e] = s[e]; }
        wave_sync();
        unsigned long long sse;
        wave_ts_recon(org, rec, lev, w, h, bd, p.qp_tr, 1, lane, sse);
        double cost = 0; int better = 0;
        Cab cb; cb.ci = CI_W(wave); cb.bits = 0;
        if (lane == 0) {
          enc_intra_luma_pred_mode(cb, L.ny, mode, mrl); enc_bin(cb, 1u, VX_CTX_QtCbf[0]); enc_tu_mts(cb, w, h, 1);
          rc_ts_serial(cb, lev, w, h);
          cost = rd_cost(p, cb.bits, sse);
          better = cost < recA[i].cost;
          if (better) { recA[i].cost = cost; recA[i].dist = sse; recA[i].bits = cb.bits; recA[i].cbf = 1; recA[i].wave = wave; recA[i].sum0 = 1 << 8; }      // tu.mtsIdx 1; a transform-skip block leaves the LFNST conditions alone (3837-3850)
        }
        cost = lane0_d(cost);
        wave_sync();
        if (cost < wbest || (cost == wbest && c * 8 + 1 < wkey)) {
          wbest = cost; wkey = c * 8 + 1;
          if (lane == 0) { L.wave_best[wave] = c; L.wave_slot[wave] = SMALL ? 1 : cur; }
          if (SMALL) { int16_t *pr = slot_rec(scratch, P, wave, 1), *pl = slot_lev(scratch, P, wave, 1); for (int e = lane; e < P; e += 64) { pr[e] = rec[e]; pl[e] = lev[e]; } }
          else cur ^= 1;
          { uint32_t *d = (uint32_t *) ctx_ptr(scratch, CTX_START, MAXD + wave, 0); const uint32_t *s = (const uint32_t *) &L.ctxs[CI_W(wave)]; for (int e = lane; e < NCTX; e += 64) d[e] = s[e]; }
        }
        wave_sync();
      }
      __threadfence_block();
      __syncthreads();
    }
    // ---- the (candidate, MTS pair) items of this chunk, candidate by candidate in transform order; every thread builds the same list
    const long long q3 = STAMP();
    uint8_t *pi_ = L.rb_pairs;
    if (VTX == 0) {
      int n = 0;
      for (int i = 0; i < nA; i++) { const int t = recA[i].test; for (int k = 1; k < 5; k++) if (((t >> k) & 1) && n < VXD_POOL_ITEMS) pi_[n++] = (uint8_t) ((i << 3) | k); }
      L.rb_nb = n;
    }
    __syncthreads();
    const int nB = uni(L.rb_nb);
    if (nB) {
      // ---- B1
      for (int j = wave; j < nB; j += NW) {
        const int i = uni((int) pi_[j] >> 3), k = uni((int) pi_[j] & 7);
        int16_t *rec = SMALL ? L.wm[wave].slot : slot_rec(scratch, P, wave, cur), *lev = SMALL ? L.wm[wave].slot + BUF : slot_lev(scratch, P, wave, cur);
        for (int e = lane; e < P; e += 64) rec[e] = poolPred[(size_t) i * P + e];
        wave_sync();
        unsigned long long sse; int cbf;
        wave_code_block_mts<SMALL>(org, rec, lev, wave_tmp(scratch, imin(w, 32) * h, wave), w, h, bd, p.qp_tr, k + 1, lane, sse, cbf, -2);
        for (int e = lane; e < P; e += 64) poolCoef[(size_t) j * P + e] = lev[e];
      }
      __threadfence_block();
      __syncthreads();
      const long long q4 = STAMP();
      dq_trellis_phase<SMALL>(scratch, nB, P, total, w, h, 1, wave, lane);
      __threadfence_block();
      __syncthreads();
      if (VVCX_STAMP && VTX == 0) { PROF(42) += (unsigned long long) (q4 - q3); PROF(43) += (unsigned long long) (STAMP() - q4); PROF(46) += (unsigned long long) nB; }
      // ---- B3
      for (int j = wave; j < nB; j += NW) {
        const int i = uni((int) pi_[j] >> 3), k = uni((int) pi_[j] & 7);
        const int c = c0 + i, mode = uni(L.rd[c].mode), mrl = uni(L.rd[c].mrl), mts = k + 1;
        nmts++;
        const int cbf = uni(L.dq_abs[j]) > 0;
        if (lane == 0) { recB[j].cost = MAX_DOUBLE; recB[j].cbf = cbf; recB[j].wave = wave; }
        if (!cbf) continue;                                 // an MTS index is not coded for a zero block: forbidden
        int16_t *rec = SMALL ? L.wm[wave].slot : slot_rec(scratch, P, wave, cur), *lev = SMALL ? L.wm[wave].slot + BUF : slot_lev(scratch, P, wave, cur);
        for (int e = lane; e < P; e += 64) { rec[e] = poolPred[(size_t) i * P + e]; lev[e] = poolCoef[(size_t) j * P + e]; }
        wave_sync();
        unsigned long long sse; int cbf2;
        wave_code_block_mts<SMALL>(org, rec, lev, wave_tmp(scratch, imin(w, 32) * h, wave), w, h, bd, p.qp_tr, mts, lane, sse, cbf2, 1);
        { uint32_t *d = (uint32_t *) &L.ctxs[CI_W(wave)]; const uint32_t *s = (const uint32_t *) &L.ctxs[CI_CUR]; for (int e = lane; e < NCTX; e += 64) d[e] = s[e]; }
        wave_sync();
        double cost = 0;
        Cab cb; cb.ci = CI_W(wave); cb.bits = 0;
        if (lane == 0) { enc_intra_luma_pred_mode(cb, L.ny, mode, mrl); enc_bin(cb, 1u, VX_CTX_QtCbf[0]); enc_tu_mts(cb, w, h, mts); }
        residual_coding_wave<SMALL>(cb, 0, lev, w, h, 0, lane, 1);
        if (lane == 0) { cost = rd_cost(p, cb.bits, sse); recB[j].cost = cost; recB[j].dist = sse; recB[j].bits = cb.bits; }
        cost = lane0_d(cost);
        if (cost < wbest || (cost == wbest && c * 8 + k < wkey)) {
          wbest = cost; wkey = c * 8 + k;
          if (lane == 0) { L.wave_best[wave] = c; L.wave_slot[wave] = SMALL ? 1 : cur; }
          if (SMALL) { int16_t *pr = slot_rec(scratch, P, wave, 1), *pl = slot_lev(scratch, P, wave, 1); for (int e = lane; e < P; e += 64) { pr[e] = rec[e]; pl[e] = lev[e]; } }
          else cur ^= 1;
          { uint32_t *d = (uint32_t *) ctx_ptr(scratch, CTX_START, MAXD + wave, 0); const uint32_t *s = (const uint32_t *) &L.ctxs[CI_W(wave)]; for (int e = lane; e < NCTX; e += 64) d[e] = s[e]; }
        }
        wave_sync();
      }
      __threadfence_block();
      __syncthreads();
    }
    if (VVCX_STAMP && VTX == 0) { PROF(38) += (unsigned long long) (q1 - q0); PROF(39) += (unsigned long long) (q2 - q1); PROF(40) += (unsigned long long) (q3 - q2); PROF(41) += (unsigned long long) (STAMP() - q3); PROF(44) += 1; PROF(45) += (unsigned long long) nA; }
#ifdef VVCX_STAMP_PASS
    // diagnostic build: per kind of pass (lfnstIdx 0 / 1 / 2 without MTS, then the MTS passes of transform groups 0..3) the number of chunks and the clocks of the trellis round
    if (VTX == 0) { const int kind = psMts ? 3 + psGrp : psLf; PROF(16 + kind) += 1; PROF(23 + kind) += (unsigned long long) (q2 - q1); }
#endif
    // ---- per candidate: DCT-II, then its MTS items in transform order, strict < (xRecurIntraCodingLumaQT 3579-3616)
    if ((int) VTX < nA) {
      const int i = VTX, c = c0 + i;
      double bc = recA[i].cost; uint64_t bd_ = recA[i].dist, bb = recA[i].bits; int bcbf = recA[i].cbf, bm = 0, bw = recA[i].wave;
      for (int j = 0; j < nB; j++) if ((pi_[j] >> 3) == i) { const double v = recB[j].cost; if (v < bc) { bc = v; bd_ = recB[j].dist; bb = recB[j].bits; bcbf = 1; bm = (pi_[j] & 7) + 1; bw = recB[j].wave; } }
      if (lfOn) { bm = recA[i].sum0 >> 8; L.rd_lfl[c] = (uint8_t) (recA[i].sum0 & 3); }
      L.rd_cost[c] = bc; L.rd_dist[c] = bd_; L.rd_bits[c] = bb; L.rd_cbf[c] = (uint8_t) bcbf; L.rd_mts[c] = (uint8_t) bm; L.rd_wave[c] = (uint8_t) bw;
    }
    __syncthreads();
  }
  if (lane == 0) L.mts_evals[wave] = nmts;
}
__device__ __noinline__ void op_stage_b(const VxParams &p_, uint8_t *scratch)
{
  scratch = uni_p(scratch);
  const VxParams &p = L.par; (void) p_;
  const int wave = uni(VTX >> 6), lane = VTX & 63;
  const int w = uni(L.nw), h = uni(L.nh);
  if (uni(p.tools & TOOL_DEPQUANT)) { if (w * h <= BUF) stage_b_rounds<true>(scratch, wave, lane, w, h); else stage_b_rounds<false>(scratch, wave, lane, w, h); }
  else if (uni(p.tools & TOOL_MTS)) { if (w * h <= BUF) stage_b_loop_mts<true>(p, scratch, wave, lane, w, h); else stage_b_loop_mts<false>(p, scratch, wave, lane, w, h); }
  else if (w * h <= BUF) stage_b_loop_dct2<true>(scratch, wave, lane, w, h); else stage_b_loop_dct2<false>(scratch, wave, lane, w, h);
  __threadfence_block();
  __syncthreads();
  // winner (strict <, list order ≙ EL/IntraSearch.cpp:1308) and its end contexts → wctx[0]; every thread computes the same
  const int n_rd = uni(L.n_rd);
  int best = 0; double bc = MAX_DOUBLE;
  for (int c = 0; c < n_rd; c++) { const double v = L.rd_cost[c]; if (v < bc) { bc = v; best = c; } }
  best = uni(best);
  const int ww = uni((int) L.rd_wave[best]);
  if (VTX == 0) { L.win_idx = best; L.win_wave = ww; L.cu_bits = L.rd_bits[best]; }
  ctx_copy_all(&L.ctxs[CI_W(0)], ctx_ptr(scratch, CTX_START, MAXD + ww, 0));
  __syncthreads();
  if (uni(p.tools & TOOL_LFNST)) {                          // cu_residual ends with residual_lfnst_mode (EL/CABACWriter.cpp:2051)
    if (VTX == 0) {
      Cab cb; cb.ci = CI_W(0); cb.bits = 0;
      enc_lfnst_idx(cb, 0, w, h, (L.rd[best].mrl & MIPF) != 0, L.rd_cbf[best] && L.rd_mts[best] != 0, L.rd_lfl[best], L.ps_lfnst);
      L.cu_bits += cb.bits;
    }
    __syncthreads();
  }
  if (uni((int) L.isp_wait)) {                              // the ISP candidates that follow work in every wave's context set: the regular winner's end contexts wait in HBM
    ctx_copy_all(ctx_ptr(scratch, CTX_BEST, MAXD + 0, 0), &L.ctxs[CI_W(0)]);
    __threadfence_block();
    __syncthreads();
  }
}

template <bool SMALL> __device__ __noinline__ void chroma_rd_rounds(uint8_t *scratch, int wave, int lane, int w, int h);
// OP_CHROMA_RD: estIntraPredChromaQT 1382-1686 + xRecurIntraChromaCodingQT 3779-4207 (CCLM / JointCbCr off):
// one wave per chroma mode, Cb then Cr; winner kept per wave like stage B.
template <bool SMALL>
__device__ void chroma_rd_loop(const VxParams &p, uint8_t *scratch, int wave, int lane, int w, int h)
{
  scratch = uni_p(scratch); wave = uni(wave); w = uni(w); h = uni(h);
  const int P = w * h, bd = uni(p.bit_depth);
  const int16_t *lmin = lm_in_buf(scratch, 2 * P);
  if (uni(L.lm_ok)) {
    // SATD pre-selection (1479-1582): the first seven candidates except LM, planar (and DM, which is the eighth) are ranked by
    // SATD(Cb) + SATD(Cr); the reference's exchange sort (not stable) then drops the two last entries from the RD loop
    for (int idx = wave; idx < 7; idx += NW) {
      const int mode = uni(L.rd[idx].mode);
      long long sum = 0;
      if (mode != LM_CHROMA && mode != PLANAR) {
        int16_t *pred = SMALL ? L.wm[wave].slot : slot_rec(scratch, 2 * P, wave, 0);
        for (int k = 0; k < 2; k++) {
          chroma_pred_wave(pred, lmin, k, mode, w, h, bd, lane);
          wave_sync();
          unsigned long long sad, satd;
          wave_sad_satd<SMALL>(org_tile(scratch, 2 * P), pred, SMALL ? (int16_t *) L.wm[wave].tmp : (int16_t *) wave_tmp(scratch, P >> 1, wave), w, h, lane, sad, satd, k * P, 0);
          sum += (long long) satd;
          wave_sync();
        }
      }
      if (lane == 0) L.lm_cost[idx] = sum;
    }
    __threadfence_block();
    __syncthreads();
    if (VTX == 0) {
      int list[7]; long long cost[7]; int ns = 0;
      for (int i = 0; i < 7; i++) { list[i] = L.rd[i].mode; cost[i] = L.lm_cost[i]; ns += list[i] != LM_CHROMA && list[i] != PLANAR; }
      for (int i = 0; i < 7; i++) for (int j = i + 1; j < 7; j++) if (cost[j] < cost[i]) {
        const int tm = list[i]; list[i] = list[j]; list[j] = tm;
        const long long tc = cost[i]; cost[i] = cost[j]; cost[j] = tc;
      }
      int n = 0;
      for (int i = 0; i < 8; i++) { const int m = L.rd[i].mode; if (m == list[5] || m == list[6]) continue; L.rd[n] = L.rd[i]; n++; }
      L.n_rd = n; L.lm_nsatd = 2 * ns;
    }
    __syncthreads();
  }
  if (uni((int) (p.tools & TOOL_DEPQUANT))) { chroma_rd_rounds<SMALL>(scratch, wave, lane, w, h); return; }      // the trellis quantiser: candidates in rounds
  if (lane == 0) L.wave_best[wave] = -1;
  double wbest = MAX_DOUBLE;
  int cur = 0;
  const int n_rd = uni(L.n_rd);
  for (int c = wave; c < n_rd; c += NW) {
    const int cm = uni(L.rd[c].mode);                 // chroma mode (70 = DM); final mode in .mrl field
    const int fm = uni(L.rd[c].mrl);
    int16_t *recb = SMALL ? L.wm[wave].slot : slot_rec(scratch, 2 * P, wave, cur);
    int16_t *levb = SMALL ? L.wm[wave].slot + BUF : slot_lev(scratch, 2 * P, wave, cur);
    { uint32_t *d = (uint32_t *) &L.ctxs[CI_W(wave)]; const uint32_t *s = (const uint32_t *) &L.ctxs[CI_CUR]; for (int i = lane; i < NCTX; i += 64) d[i] = s[i]; }
    wave_sync();
    unsigned long long dist = 0; int cbfs[2];
    for (int k = 0; k < 2; k++) {
      int16_t *rec = recb + k * P, *lev = levb + k * P;
      chroma_pred_wave(rec, lmin, k, fm, w, h, bd, lane);
      wave_sync();
      unsigned long long sse; int cbf;
      wave_code_block<SMALL>(org_tile(scratch, 2 * P), k * P, k * P, rec, lev, wave_tmp(scratch, imin(w, 32) * h, wave), w, h, bd, p.qp_tr_c[k], lane, sse, cbf, -1, nullptr, k + 1, CI_W(wave), k == 1 ? cbfs[0] : 0);
      cbfs[k] = cbf;
      dist += (unsigned long long) (p.dist_weight[k] * (double) sse);             // CL/RdCost.cpp:405-408
      {                    // xGetIntraFracBitsQTChroma 2625-2692: contexts advance
        Cab cb; cb.ci = CI_W(wave); cb.bits = 0;
        if (lane == 0) enc_bin(cb, (unsigned) cbf, VX_CTX_QtCbf[k + 1] + (k == 1 ? cbfs[0] : 0));
        if (uni(cbf)) residual_coding_wave<SMALL>(cb, k * P, lev, w, h, 1, lane);
      }
      wave_sync();
    }
    double cost = 0;
    {                      // 1611-1621: contexts not reset; xGetIntraFracBitsQT(chroma)
      Cab cb; cb.ci = CI_W(wave); cb.bits = 0;
      if (lane == 0) {
        enc_intra_chroma_pred_mode(cb, cm, L.colm, L.lm_ok);
        enc_bin(cb, (unsigned) cbfs[0], VX_CTX_QtCbf[1]);
        enc_bin(cb, (unsigned) cbfs[1], VX_CTX_QtCbf[2] + cbfs[0]);
      }
      if (uni(cbfs[0])) residual_coding_wave<SMALL>(cb, 0, levb, w, h, 1, lane);
      if (uni(cbfs[1])) residual_coding_wave<SMALL>(cb, P, levb + P, w, h, 1, lane);
      if (lane == 0) {
        cost = rd_cost(p, cb.bits, dist);
        L.rd_cost[c] = cost; L.rd_dist[c] = dist; L.rd_cbf[c] = (uint8_t) ((cbfs[0] ? 2 : 0) | (cbfs[1] ? 4 : 0));
      }
    }
    cost = lane0_d(cost);
    if (cost < wbest) {
      wbest = cost;
      if (lane == 0) { L.wave_best[wave] = c; L.wave_slot[wave] = SMALL ? 1 : cur; }
      if (SMALL) { int16_t *pr = slot_rec(scratch, 2 * P, wave, 1), *pl = slot_lev(scratch, 2 * P, wave, 1); for (int i = lane; i < 2 * P; i += 64) { pr[i] = recb[i]; pl[i] = levb[i]; } }
      else cur ^= 1;
    }
    wave_sync();
  }
}
// The RD loop of the chroma search with the dependent quantiser, in rounds like stage_b_rounds (all threads; the SATD pre-selection has run):
//   C1  per mode (one wave each): Cb and Cr prediction -> prediction pool, forward DCT-II of both -> coefficient pool
//   C2  the Cb trellises of all modes side by side (they all start from the node's contexts)
//   C3  four modes at a time, one per wave: Cb reconstruction and rate (the wave's contexts advance), then the Cr trellises of the four modes in one wavefront
//       (each from its wave's contexts and with its own tu.cbf[Cb]), then Cr reconstruction, rate and the mode's cost.
// ---- JointCbCr (JVET_O0105 inter-chroma transform).  g_ictModes (CL/Rom.cpp:613): signed mode of a cbf mask under the slice's sign flag
__device__ inline int ict_mode(int sign, int mask) { const int m = mask == 1 ? 3 : mask == 2 ? 1 : 2; return sign ? -m : m; }
// fwdTransformCbCr (CL/TrQuant.cpp:87-137): joint residual of one sample pair and the squared error of representing both residuals by it
__device__ inline int ict_fwd(int b, int r, int mode, long long &err)
{
  int v, eb, er;
  switch (mode) {
    case  1: v = (4 * b + 2 * r) / 5; eb = b - v; er = r - (v >> 1); break;
    case -1: v = (4 * b - 2 * r) / 5; eb = b - v; er = r - (-v >> 1); break;
    case  2: v = (b + r) / 2; eb = b - v; er = r - v; break;
    case -2: v = (b - r) / 2; eb = b - v; er = r + v; break;
    case  3: v = (4 * r + 2 * b) / 5; eb = b - (v >> 1); er = r - v; break;
    default: v = (4 * r - 2 * b) / 5; eb = b - (-v >> 1); er = r - v; break;
  }
  err += (long long) eb * eb + (long long) er * er;
  return v;
}
// invTransformCbCr (139-156): residual of component k (0 Cb, 1 Cr) from the joint residual c of a block coded as Cb (modes +-1, +-2) or as Cr (+-3)
__device__ inline int ict_inv(int c, int mode, int k)
{
  const int am = mode < 0 ? -mode : mode, sc = mode < 0 ? -c : c;
  if (am == 3) return k == 1 ? c : (sc >> 1);
  return k == 0 ? c : (am == 1 ? (sc >> 1) : sc);
}
__device__ inline long long wave_sum_i64(long long v) { return (long long) wave_sum_u64((unsigned long long) v); }

template <bool SMALL>
__device__ __noinline__ void chroma_rd_rounds(uint8_t *scratch, int wave, int lane, int w, int h)
{
  scratch = uni_p(scratch); wave = uni(wave); w = uni(w); h = uni(h);
  const VxParams &p = L.par;
  const int P = w * h, bd = uni(p.bit_depth), total = imin(32, w) * imin(32, h);
  const int16_t *lmin = lm_in_buf(scratch, 2 * P);
  const int16_t *org = org_tile(scratch, 2 * P);
  int16_t *poolPred = (int16_t *) (scratch + VXD_OFF_POOL), *poolCoef = (int16_t *) (scratch + VXD_OFF_POOL_COEF); uint8_t *poolNodes = scratch + VXD_OFF_POOL_NODES;
  if (lane == 0) L.wave_best[wave] = -1;
  dq_build_tables(1);
  double wbest = MAX_DOUBLE;
  int cur = 0;
  const int n_rd = uni(L.n_rd);
  // LFNST of the pass on both components of blocks of at least 4x4; kernel choice: the final mode, or the co-located luma mode for the LM modes (CL/TrQuant.cpp:449-457)
  const int psLf = (uni((int) (p.tools & TOOL_LFNST)) && w >= 4 && h >= 4) ? uni((int) L.ps_lfnst) : 0;
  const int jccrOn = uni((int) (p.tools & TOOL_JCCR)) != 0;
  const int cadj = P > 4 ? uni(L.lmcs_cadj) : 0;             // LMCS chroma residual scaling of blocks of more than 4 samples (EL/IntraSearch.cpp:3884-3904, 3060-3066)
  int njoint = 0;
#define CHROMA_LFMODE(c_) (psLf ? lfnst_mode((uni((int) L.rd[c_].mode) >= LM_CHROMA && uni((int) L.rd[c_].mode) <= MDLM_T) ? uni(L.colm) : uni((int) L.rd[c_].mrl), w, h) : 0)
  // ---- C1
  for (int c = wave; c < n_rd; c += NW) {
    const int fm = uni(L.rd[c].mrl), lfm = CHROMA_LFMODE(c);
    int16_t *recb = SMALL ? L.wm[wave].slot : slot_rec(scratch, 2 * P, wave, cur), *levb = SMALL ? L.wm[wave].slot + BUF : slot_lev(scratch, 2 * P, wave, cur);
    for (int k = 0; k < 2; k++) {
      int16_t *rec = recb + k * P, *lev = levb + k * P;
      chroma_pred_wave(rec, lmin, k, fm, w, h, bd, lane);
      wave_sync();
      for (int e = lane; e < P; e += 64) poolPred[(size_t) (2 * c + k) * P + e] = rec[e];
      unsigned long long sse; int cbf;
      wave_code_block<SMALL>(org, k * P, k * P, rec, lev, wave_tmp(scratch, imin(w, 32) * h, wave), w, h, bd, p.qp_tr_c[k], lane, sse, cbf, -2, nullptr, k + 1, CI_CUR, 0, psLf, lfm, 0, -1, cadj);
      for (int e = lane; e < P; e += 64) poolCoef[(size_t) (2 * c + k) * P + e] = lev[e];
      wave_sync();
    }
  }
  __threadfence_block();
  __syncthreads();
  // ---- C2: items 0, 2, 4, ... of the pool (stride 2P)
  {
    const int ipw = imin(16, (int) sizeof(WaveMem) / (240 + 2 * total));
    for (int i0 = wave * ipw; i0 < n_rd; i0 += NW * ipw)
      wave_depquant_batch<2>(imin(ipw, n_rd - i0), poolCoef + (size_t) (2 * i0) * P, 2 * P, poolNodes + (size_t) (2 * i0) * 4 * total, 8 * total, wave, i0,
                          CI_CUR, 0, VX_CTX_QtCbf[1], 0u, w, h, 1, 0, psLf, lane);
  }
  __threadfence_block();
  __syncthreads();
  // ---- C3
  const bool crBatch = 4 * (240 + 2 * total) <= (int) (sizeof(WaveScratch) + sizeof(int32_t) * BUF);      // four Cr trellises fit wave 0's rate-estimator scratch + tmp
  for (int c0 = 0; c0 < n_rd; c0 += NW) {
    const int c = c0 + wave; const bool have = c < n_rd;
    const int cm = have ? uni(L.rd[c].mode) : 0;
    const int lfm = have ? CHROMA_LFMODE(c) : 0;
    int16_t *recb = SMALL ? L.wm[wave].slot : slot_rec(scratch, 2 * P, wave, cur), *levb = SMALL ? L.wm[wave].slot + BUF : slot_lev(scratch, 2 * P, wave, cur);
    unsigned long long dist = 0; int cbfs[2] = { 0, 0 };
    double compCost = 0; int jccr = 0;
    if (have) {
      { uint32_t *d = (uint32_t *) &L.ctxs[CI_W(wave)]; const uint32_t *s_ = (const uint32_t *) &L.ctxs[CI_CUR]; for (int e = lane; e < NCTX; e += 64) d[e] = s_[e]; }
      for (int e = lane; e < P; e += 64) { recb[e] = poolPred[(size_t) (2 * c) * P + e]; levb[e] = poolCoef[(size_t) (2 * c) * P + e]; }
      wave_sync();
      cbfs[0] = uni(L.dq_abs[c]) > 0;
      unsigned long long sse; int cbf2;
      wave_code_block<SMALL>(org, 0, 0, recb, levb, wave_tmp(scratch, imin(w, 32) * h, wave), w, h, bd, p.qp_tr_c[0], lane, sse, cbf2, cbfs[0], nullptr, 1, CI_W(wave), 0, psLf, lfm, 0, -1, cadj);
      dist += (unsigned long long) (p.dist_weight[0] * (double) sse);
      { Cab cb; cb.ci = CI_W(wave); cb.bits = 0; if (lane == 0) enc_bin(cb, (unsigned) cbfs[0], VX_CTX_QtCbf[1]); if (cbfs[0]) residual_coding_wave<SMALL>(cb, 0, levb, w, h, 1, lane);
        double c0 = 0; if (lane == 0) c0 = rd_cost(p, cb.bits, dist); compCost = lane0_d(c0); }      // xGetIntraFracBitsQTChroma(Cb) 2625-2692
      wave_sync();
      if (lane == 0) L.rb_pairs[wave] = (uint8_t) cbfs[0];
    }
    __threadfence_block();
    __syncthreads();
    {                                                       // Cr trellises of the modes c0 .. c0 + 3
      const int n4 = imin(NW, n_rd - c0);
      unsigned mask = 0; for (int i = 0; i < n4; i++) mask |= (unsigned) L.rb_pairs[i] << i;
      if (crBatch) { if (wave == 0) wave_depquant_batch<0>(n4, poolCoef + (size_t) (2 * c0 + 1) * P, 2 * P, poolNodes + (size_t) (2 * c0 + 1) * 4 * total, 8 * total, 0, c0,
                                                        CI_W(0), 1, VX_CTX_QtCbf[2], mask, w, h, 2, 0, psLf, lane); }
      else if (have) wave_depquant_batch<0>(1, poolCoef + (size_t) (2 * c + 1) * P, 0, poolNodes + (size_t) (2 * c + 1) * 4 * total, 0, wave, c,
                                         CI_W(wave), 0, VX_CTX_QtCbf[2], (unsigned) cbfs[0], w, h, 2, 0, psLf, lane);
    }
    __threadfence_block();
    __syncthreads();
    if (have) {
      int16_t *rec = recb + P, *lev = levb + P;
      for (int e = lane; e < P; e += 64) { rec[e] = poolPred[(size_t) (2 * c + 1) * P + e]; lev[e] = poolCoef[(size_t) (2 * c + 1) * P + e]; }
      wave_sync();
      cbfs[1] = uni(L.dq_abs[c]) > 0;
      unsigned long long sse; int cbf2;
      wave_code_block<SMALL>(org, P, P, rec, lev, wave_tmp(scratch, imin(w, 32) * h, wave), w, h, bd, p.qp_tr_c[1], lane, sse, cbf2, cbfs[1], nullptr, 2, CI_W(wave), cbfs[0], psLf, lfm, 0, -1, cadj);
      const unsigned long long distCr = (unsigned long long) (p.dist_weight[1] * (double) sse);
      dist += distCr;
      { Cab cb; cb.ci = CI_W(wave); cb.bits = 0;
        if (lane == 0) { enc_bin(cb, (unsigned) cbfs[1], VX_CTX_QtCbf[2] + cbfs[0]); if (jccrOn && (cbfs[0] | cbfs[1])) enc_bin(cb, 0u, VX_CTX_JointCbCrFlag + ((cbfs[0] ? 2 : 0) | (cbfs[1] ? 1 : 0)) - 1); }
        if (cbfs[1]) residual_coding_wave<SMALL>(cb, P, lev, w, h, 1, lane);
        double c1 = 0; if (lane == 0) c1 = compCost + rd_cost(p, cb.bits, distCr); compCost = lane0_d(c1); }
      wave_sync();
      if (jccrOn && (cbfs[0] | cbfs[1])) {
        // xRecurIntraChromaCodingQT 4060-4150: up to two joint candidates (TrQuant::selectICTCandidates) against the sum of the two separate costs
        int16_t *jb = (int16_t *) (scratch + VXD_OFF_JCCR + (size_t) wave * VXD_JCCR_WAVE);
        int16_t *jres = jb, *jout = jb + 1024, *jlev = jb + 2048, *brec = jb + 3072, *blev = jb + 5120;
        const int16_t *pCb = poolPred + (size_t) (2 * c) * P, *pCr = poolPred + (size_t) (2 * c + 1) * P;
        const int sign = uni(L.fdv.jccr_sign);
        long long d0 = 0, d1 = 0, e1 = 0, e2 = 0, e3 = 0;
        for (int e = lane; e < P; e += 64) {
          int b = org[e] - pCb[e], r = org[P + e] - pCr[e];
          if (cadj) { b = lmcs_scale_fwd(b, cadj, bd); r = lmcs_scale_fwd(r, cadj, bd); }      // 3915-3925: the residuals the joint candidates are built from are the scaled ones
          d0 += (long long) b * b; d1 += (long long) r * r;
          ict_fwd(b, r, ict_mode(sign, 1), e1); ict_fwd(b, r, ict_mode(sign, 2), e2); ict_fwd(b, r, ict_mode(sign, 3), e3);
        }
        d0 = wave_sum_i64(d0); d1 = wave_sum_i64(d1); e1 = wave_sum_i64(e1); e2 = wave_sum_i64(e2); e3 = wave_sum_i64(e3);
        long long min1 = d0 < d1 ? d0 : d1, min2 = 0x7fffffffffffffffll; int m1 = 0, m2 = 0;
        for (int m = 1; m < 4; m++) {
          const long long pd = m == 1 ? e1 : m == 2 ? e2 : e3;
          if (pd < min1) { m2 = m1; min2 = min1; m1 = m; min1 = pd; } else if (pd < min2) { m2 = m; min2 = pd; }
        }
        int masks[2], nm = 0;
        if (m1) masks[nm++] = m1;
        if (m2 && ((min2 < (9 * min1) / 8) || (!m1 && min2 < (3 * min1) / 2))) masks[nm++] = m2;
        nm = uni(nm);
        unsigned long long bestJDist = 0;
        for (int q = 0; q < nm; q++) {
          const int mask = uni(masks[q]), mode = ict_mode(sign, mask), comp = (mask >> 1) ? 1 : 2;
          for (int e = lane; e < P; e += 64) {
            long long dm = 0; int b = org[e] - pCb[e], r = org[P + e] - pCr[e];
            if (cadj) { b = lmcs_scale_fwd(b, cadj, bd); r = lmcs_scale_fwd(r, cadj, bd); }
            jres[e] = (int16_t) ict_fwd(b, r, mode, dm); jout[e] = 0;
          }
          wave_sync();
          njoint++;
          unsigned long long sseJ; int cbfJ;
          // the joint block: quantised as the Cb (masks 2, 3) / Cr (mask 1) block from the TU's start contexts, at the component's QP or (mask 3) the JointCbCr QP, loosened lambda (table rows 3..5)
          wave_code_block<false>(jres, 0, 0, jout, jlev, wave_tmp(scratch, imin(w, 32) * h, wave), w, h, bd, mask == 3 ? p.qp_tr_j : p.qp_tr_c[comp - 1], lane, sseJ, cbfJ, -1, nullptr, comp, CI_CUR, 0, psLf, lfm, 1, 2 + mask);
          if (!cbfJ) continue;                              // the mask cannot be signalled with an empty block (3083-3087)
          unsigned long long sCb = 0, sCr = 0;
          const int mx = (1 << bd) - 1;
          for (int e = lane; e < P; e += 64) {
            const int cj = jout[e];
            int rb = ict_inv(cj, mode, 0), rr = ict_inv(cj, mode, 1);
            if (cadj) { rb = lmcs_scale_inv((int) (int16_t) rb, cadj, bd); rr = lmcs_scale_inv((int) (int16_t) rr, cadj, bd); }      // 3060-3070: both residuals back through the inverse chroma scaling
            int vb = pCb[e] + rb, vr = pCr[e] + rr;
            vb = vb < 0 ? 0 : vb > mx ? mx : vb; vr = vr < 0 ? 0 : vr > mx ? mx : vr;
            jres[e] = (int16_t) vb; jout[e] = (int16_t) vr;          // the pair of reconstructions replaces the residuals
            const int db = org[e] - vb, dr = org[P + e] - vr;
            sCb += (unsigned long long) (db * db); sCr += (unsigned long long) (dr * dr);
          }
          sCb = wave_sum_u64(sCb); sCr = wave_sum_u64(sCr);
          const unsigned long long dJ = (unsigned long long) (p.dist_weight[0] * (double) sCb) + (unsigned long long) (p.dist_weight[1] * (double) sCr);
          { uint32_t *d = (uint32_t *) &L.ctxs[CI_W(wave)]; const uint32_t *s_ = (const uint32_t *) &L.ctxs[CI_CUR]; for (int e = lane; e < NCTX; e += 64) d[e] = s_[e]; }
          wave_sync();
          Cab cb; cb.ci = CI_W(wave); cb.bits = 0;
          if (lane == 0) { enc_bin(cb, (unsigned) (mask >> 1), VX_CTX_QtCbf[1]); enc_bin(cb, (unsigned) (mask & 1), VX_CTX_QtCbf[2] + (mask >> 1)); enc_bin(cb, 1u, VX_CTX_JointCbCrFlag + mask - 1); }
          residual_coding_wave<false>(cb, 0, jlev, w, h, 1, lane);
          double cj = 0; if (lane == 0) cj = rd_cost(p, cb.bits, dJ);
          cj = lane0_d(cj);
          if (cj < compCost) {
            compCost = cj; jccr = mask; bestJDist = dJ;
            for (int e = lane; e < P; e += 64) { brec[e] = jres[e]; brec[P + e] = jout[e]; blev[e] = jlev[e]; }
          }
          wave_sync();
        }
        if (jccr) {                                         // the joint candidate replaces the separately coded pair
          dist = bestJDist; cbfs[0] = jccr >> 1; cbfs[1] = jccr & 1;
          for (int e = lane; e < P; e += 64) { recb[e] = brec[e]; recb[P + e] = brec[P + e]; levb[e] = (jccr >> 1) ? blev[e] : 0; levb[P + e] = (jccr >> 1) ? 0 : blev[e]; }
        }
        wave_sync();
      }
      if (jccrOn) {                                         // the contexts after the winning variant (4152-4170): replayed from the TU's start contexts
        { uint32_t *d = (uint32_t *) &L.ctxs[CI_W(wave)]; const uint32_t *s_ = (const uint32_t *) &L.ctxs[CI_CUR]; for (int e = lane; e < NCTX; e += 64) d[e] = s_[e]; }
        wave_sync();
        Cab cb; cb.ci = CI_W(wave); cb.bits = 0;
        const int msk = (cbfs[0] ? 2 : 0) | (cbfs[1] ? 1 : 0);
        if (!jccr) {
          if (lane == 0) enc_bin(cb, (unsigned) cbfs[0], VX_CTX_QtCbf[1]);
          if (cbfs[0]) residual_coding_wave<SMALL>(cb, 0, levb, w, h, 1, lane);
          if (lane == 0) { enc_bin(cb, (unsigned) cbfs[1], VX_CTX_QtCbf[2] + cbfs[0]); if (msk) enc_bin(cb, 0u, VX_CTX_JointCbCrFlag + msk - 1); }
          if (cbfs[1]) residual_coding_wave<SMALL>(cb, P, levb + P, w, h, 1, lane);
        } else {
          if (lane == 0) { enc_bin(cb, (unsigned) cbfs[0], VX_CTX_QtCbf[1]); enc_bin(cb, (unsigned) cbfs[1], VX_CTX_QtCbf[2] + cbfs[0]); enc_bin(cb, 1u, VX_CTX_JointCbCrFlag + msk - 1); }
          if (jccr >> 1) residual_coding_wave<SMALL>(cb, 0, levb, w, h, 1, lane); else residual_coding_wave<SMALL>(cb, P, levb + P, w, h, 1, lane);
        }
        wave_sync();
      }
      double cost = 0;
      {                    // 1611-1621: contexts not reset; xGetIntraFracBitsQT(chroma)
        Cab cb; cb.ci = CI_W(wave); cb.bits = 0;
        if (lane == 0) {
          enc_intra_chroma_pred_mode(cb, cm, L.colm, L.lm_ok);
          enc_bin(cb, (unsigned) cbfs[0], VX_CTX_QtCbf[1]);
          enc_bin(cb, (unsigned) cbfs[1], VX_CTX_QtCbf[2] + cbfs[0]);
          if (jccrOn && (cbfs[0] | cbfs[1])) enc_bin(cb, jccr ? 1u : 0u, VX_CTX_JointCbCrFlag + ((cbfs[0] ? 2 : 0) | (cbfs[1] ? 1 : 0)) - 1);
        }
        int fl = 0;
        if (cbfs[0]) { residual_coding_wave<SMALL>(cb, 0, levb, w, h, 1, lane); fl |= lfnst_flags(uni(L.rc_last[wave]), w, h); }
        if (cbfs[1] && jccr != 3) { residual_coding_wave<SMALL>(cb, P, levb + P, w, h, 1, lane); fl |= lfnst_flags(uni(L.rc_last[wave]), w, h); }
        if (lane == 0) {
          cost = rd_cost(p, cb.bits, dist);
          L.rd_cost[c] = cost; L.rd_dist[c] = dist; L.rd_cbf[c] = (uint8_t) ((cbfs[0] ? 2 : 0) | (cbfs[1] ? 4 : 0)); L.rd_lfl[c] = (uint8_t) fl; L.rd_mts[c] = (uint8_t) jccr;
        }
      }
      cost = lane0_d(cost);
      if (cost < wbest) {
        wbest = cost;
        if (lane == 0) { L.wave_best[wave] = c; L.wave_slot[wave] = SMALL ? 1 : cur; }
        if (SMALL) { int16_t *pr = slot_rec(scratch, 2 * P, wave, 1), *pl = slot_lev(scratch, 2 * P, wave, 1); for (int e = lane; e < 2 * P; e += 64) { pr[e] = recb[e]; pl[e] = levb[e]; } }
        else cur ^= 1;
      }
      wave_sync();
    }
  }
#undef CHROMA_LFMODE
  if (lane == 0) L.mts_evals[wave] = njoint;                  // joint blocks coded on top of the 2 per mode (work counters)
}
template <typename T>
__device__ __noinline__ void op_chroma_rd(const VxParams &p_, const VxFrameDev &fd_, uint8_t *scratch)
{
  scratch = uni_p(scratch);
  const VxParams &p = L.par; (void) p_; const VxFrameDev &fd = L.fdv; (void) fd_;
  const int wave = uni(VTX >> 6), lane = VTX & 63;
  const int x = uni(L.nx) >> 1, y = uni(L.ny) >> 1, w = uni(L.nw) >> 1, h = uni(L.nh) >> 1, P = w * h;
  for (int c = 1; c <= 2; c++) {
    const void *org = fd.org[c]; const int st = fd.stride[c];
    int16_t *ot = org_tile(scratch, 2 * P);
    for (int i = VTX; i < P; i += NT) { const int r = i >> ilog2i(w), cc = i & (w - 1); ot[(c - 1) * P + i] = (int16_t) ld_px<T>(org, (y + r) * st + x + cc); }
    build_refs<T>(p, fd, c, x, y, w, h, uni(L.cur_tile), 1);
  }
  if (VTX < 2) L.dc_val[VTX] = dc_value(L.refs[VTX][0], L.refs[VTX][1], w, h, 0);
  lmcs_chroma_adj<T>(uni(L.nx), uni(L.ny));
  if (uni(L.lm_ok)) cclm_prepare<T>(p, fd, scratch, x, y, w, h);
  __syncthreads();
  if (2 * P <= BUF) chroma_rd_loop<true>(p, scratch, wave, lane, w, h); else chroma_rd_loop<false>(p, scratch, wave, lane, w, h);
  __threadfence_block();
  __syncthreads();
  // winner; its CU-level bits are recomputed from the node's start contexts (EL/EncCu.cpp:2593-2619) by wave 0
  const int n_rd = uni(L.n_rd);
  int best = 0; double bc = MAX_DOUBLE;
  for (int c = 0; c < n_rd; c++) { const double v = L.rd_cost[c]; if (v < bc) { bc = v; best = c; } }
  best = uni(best);
  const int ww = best % NW;
  ctx_copy_all(&L.ctxs[CI_W(0)], &L.ctxs[CI_CUR]);
  __syncthreads();
  if (wave == 0) {
    const int slot = uni(L.wave_slot[ww]);
    const int16_t *levw = slot_lev(scratch, 2 * P, ww, slot);       // HBM (parked or big block)
    const int cbfm = uni(L.rd_cbf[best]), jm = (p.tools & TOOL_JCCR) ? uni((int) L.rd_mts[best]) : 0;
    Cab cb; cb.ci = CI_W(0); cb.bits = 0;
    if (lane == 0) {
      enc_intra_chroma_pred_mode(cb, L.rd[best].mode, L.colm, L.lm_ok);
      enc_bin(cb, (unsigned) !!(cbfm & 2), VX_CTX_QtCbf[1]);
      enc_bin(cb, (unsigned) !!(cbfm & 4), VX_CTX_QtCbf[2] + !!(cbfm & 2));
      if ((p.tools & TOOL_JCCR) && (cbfm & 6)) enc_bin(cb, jm ? 1u : 0u, VX_CTX_JointCbCrFlag + (((cbfm & 2) ? 2 : 0) | ((cbfm & 4) ? 1 : 0)) - 1);
    }
    if (cbfm & 2) residual_coding_wave<false>(cb, 0, levw, w, h, 1, lane);
    if ((cbfm & 4) && jm != 3) residual_coding_wave<false>(cb, 0, levw + P, w, h, 1, lane);
    if (lane == 0) { enc_lfnst_idx(cb, 1, 2 * w, 2 * h, 0, 0, L.rd_lfl[best], L.ps_lfnst); L.win_idx = best; L.win_wave = ww; L.cu_bits = cb.bits; }
  }
  __syncthreads();
}

// ---- CU-result cache (BestEncInfoCache, EL/EncModeCtrl.cpp:663-1110): entry index and level-pool offset of a node, -1 if the
// node cannot be cached (sizes above 64 are never intra coded; origins are multiples of max(4, size/2) by construction of QT/BT/TT)
__device__ int cache_slot(int x, int y, int w, int h, int &lev_off)
{
  const int lw = ilog2i(w), lh = ilog2i(h);
  if (lw > 6 || lh > 6) return -1;
  const int ax = imax(4, w >> 1), ay = imax(4, h >> 1), rx = x & 127, ry = y & 127;
  if ((rx & (ax - 1)) | (ry & (ay - 1))) return -1;
  const int lax = ilog2i(ax), lay = ilog2i(ay), npx = 128 >> lax, segW = w * npx;
  const int cumW = lw == 2 ? 0 : 128 + (lw - 3) * 256, cumH = lh == 2 ? 0 : 128 + (lh - 3) * 256;
  lev_off = cumW * VXD_CACHE_DIM + segW * cumH + ((ry >> lay) * npx + (rx >> lax)) * w * h;
  return (((ry >> 2) * 32 + (rx >> 2)) * 5 + (lw - 2)) * 5 + (lh - 2);
}
// isValid (987-1024) + isTheSameNbHood (664-703): thread 0, at node entry.  Frame i of the stack was produced by split
// fr[i].last_split of frame i-1, so the partitioner stack is the frame stack.
__device__ int cache_is_valid(const VxParams &p, uint8_t *scratch, Frame *fr, int d, int ch)
{
  Frame &f = fr[d];
  if (!(p.tools & TOOL_CU_REUSE) || d == 0) return 0;
  int lo;
  const int e = cache_slot(f.x, f.y, f.w, f.h, lo);
  if (e < 0) return 0;
  const VxCacheEnt c = ((const VxCacheEnt *) (scratch + VXD_OFF_CACHE))[e];
  if (c.kind != ch + 1 || c.gen != (uint16_t) L.cache_gen) return 0;      // entries of earlier CTUs carry another generation
  int i = 1;
  for (; i <= d; i++) {
    const int dpt = i - 1, s = dpt >= c.depth ? SPLIT_NONE : (int) ((c.ss >> (5 * dpt)) & 31);
    if (fr[i].last_split != s) break;
  }
  if (fr[i - 1].x != f.x || fr[i - 1].y != f.y) return 0;
  f.r_dir = c.dir; f.r_mrl = c.mrl; f.r_cbf = c.cbf; f.r_mts = c.mts;
  return 1;
}

// wave 0 of OP_REUSE: prediction, reconstruction from the cached levels, distortion and CU bits of the cached CU
template <bool SMALL>
__device__ void reuse_eval(const VxParams &p, uint8_t *scratch, int lane, int ch, int w, int h)
{
  scratch = uni_p(scratch); ch = uni(ch); w = uni(w); h = uni(h);
  const int P = w * h, n = ch ? 2 * P : P, bd = uni(p.bit_depth);
  int16_t *recb = SMALL ? L.wm[0].slot : slot_rec(scratch, n, 0, 0), *levb = SMALL ? L.wm[0].slot + BUF : slot_lev(scratch, n, 0, 0);
  const int mode = uni(L.rd[0].mode), fm = uni(L.rd[0].mrl), cbfm = uni(L.rd_cbf[0]);
  Cab cb; cb.ci = CI_W(0); cb.bits = 0;
  unsigned long long dist = 0;
  if (!ch && uni((int) L.isp_split)) {
    // DecCu::xIntraRecQT of an ISP CU: every sub-partition predicted from the reconstruction of the one before, then the CU's bits from the node's start contexts
    const int isp = uni((int) L.isp_split), tucbf = uni((int) L.isp_tucbf), hor = isp == 1;
    const int psz = isp_split_dim(w, h, hor), tw = hor ? w : psz, th = hor ? psz : h, nsub = hor ? h / psz : w / psz, ltw = ilog2i(tw);
    { int16_t *t = isp_tile(scratch, 0);                 // reconstructed in the HBM tile (the region references use the idle wave 1's LDS slot), then into the result slot
      isp_code_cu<false>(scratch, w, h, mode, isp, 0.0, levb, tucbf, t, t + 4096, CI_W(1), lane, t, nullptr, 1);
      for (int e = lane; e < P; e += 64) recb[e] = t[e];
      wave_sync(); }
    dist = L.isp_dist;
    int16_t *cf = isp_tile(scratch, 0) + 12288;
    if (lane == 0) enc_intra_luma_pred_mode(cb, L.ny, mode, 0, isp);
    for (int k = 0, sofar = 0; k < nsub; k++) {
      const int ox = hor ? 0 : k * tw, oy = hor ? k * th : 0, c = (tucbf >> k) & 1;
      if (lane == 0 && !(k == nsub - 1 && !sofar)) enc_bin(cb, (unsigned) c, VX_CTX_QtCbf[0] + 2 + (k ? (tucbf >> (k - 1)) & 1 : 0));
      if (c) {
        for (int e = lane; e < tw * th; e += 64) cf[e] = levb[(oy + (e >> ltw)) * w + ox + (e & (tw - 1))];
        wave_sync();
        residual_coding_wave<false>(cb, 0, cf, tw, th, 0, lane);
      }
      sofar |= c;
    }
  } else if (!ch) {
    if (fm & MIPF) wave_pred_mip(recb, w, h, mode, bd, 0, lane);
    else {
      Ipa ip; init_pred_params(w, h, 1, mode, fm, ip);
      const int set = luma_set(fm, ip.ref_filter);
      const int dcv = L.dc_val[luma_set(fm, 0)];
      for (int i = lane; i < P; i += 64) { const int py = i >> ilog2i(w), px = i & (w - 1); recb[i] = (int16_t) pred_sample(L.refs[set][0], L.refs[set][1], w, h, px, py, ip, mode, 1, bd, dcv); }
    }
    wave_sync();
    int cbf;
    const int mts = uni(L.rd_mts[0]), lf = uni((int) L.ps_lfnst);
    if (mts == 1) wave_ts_recon(org_tile(scratch, n), recb, levb, w, h, bd, p.qp_tr, cbfm & 1, lane, dist);      // a transform-skip block (DecCu::xIntraRecBlk -> xITransformSkip)
    else if (mts > 1) wave_code_block_mts<SMALL>(org_tile(scratch, n), recb, levb, wave_tmp(scratch, imin(w, 32) * h, 0), w, h, bd, p.qp_tr, mts, lane, dist, cbf, cbfm & 1);
    else wave_code_block<SMALL>(org_tile(scratch, n), 0, 0, recb, levb, wave_tmp(scratch, imin(w, 32) * h, 0), w, h, bd, p.qp_tr, lane, dist, cbf, cbfm & 1, nullptr, 0, 0, 0, lf, lf ? lfnst_mode((fm & MIPF) ? PLANAR : mode, w, h) : 0);
    if (lane == 0) { enc_intra_luma_pred_mode(cb, L.ny, mode, fm); enc_bin(cb, (unsigned) (cbfm & 1), VX_CTX_QtCbf[0]); if (cbfm & 1) enc_tu_mts(cb, w, h, mts); }
    int fl = 0;
    if ((cbfm & 1) && mts == 1) { if (lane == 0) rc_ts_serial(cb, levb, w, h); wave_sync(); }
    else if (cbfm & 1) { residual_coding_wave<SMALL>(cb, 0, levb, w, h, 0, lane, mts > 1); fl = lfnst_flags(uni(L.rc_last[0]), w, h); }
    if (lane == 0) enc_lfnst_idx(cb, 0, w, h, (fm & MIPF) != 0, (cbfm & 1) && mts != 0, fl, lf);
  } else {
    const int lf = (w >= 4 && h >= 4) ? uni((int) L.ps_lfnst) : 0;
    const int lfm = lf ? lfnst_mode((mode >= LM_CHROMA && mode <= MDLM_T) ? uni(L.colm) : fm, w, h) : 0;
    const int jm = (p.tools & TOOL_JCCR) ? uni((int) L.rd_mts[0]) : 0;
    const int cadj = ((cbfm & 6) && P > 4) ? uni(L.lmcs_cadj) : 0;      // DL/DecCu.cpp:359-367
    if (jm) {
      // DecCu::xIntraRecQT of a joint TU (DL/DecCu.cpp:330-414): the coded block's residual at its QP, the other block through the inverse ICT
      int16_t *jb = (int16_t *) (scratch + VXD_OFF_JCCR), *jout = jb + 1024, *jlev = jb + 2048;
      const int mode = ict_mode(uni(L.fdv.jccr_sign), jm), comp = (jm >> 1) ? 1 : 2;
      const int16_t *org = org_tile(scratch, n);
      for (int e = lane; e < P; e += 64) { jout[e] = 0; jlev[e] = levb[(comp - 1) * P + e]; }
      wave_sync();
      unsigned long long sseJ; int cbfJ;
      wave_code_block<false>(jout, 0, 0, jout, jlev, wave_tmp(scratch, imin(w, 32) * h, 0), w, h, bd, jm == 3 ? p.qp_tr_j : p.qp_tr_c[comp - 1], lane, sseJ, cbfJ, 1, nullptr, comp, 0, 0, lf, lfm, 1);
      const int mx = (1 << bd) - 1;
      for (int k = 0; k < 2; k++) {
        int16_t *rec = recb + k * P;
        chroma_pred_wave(rec, lm_in_buf(scratch, n), k, fm, w, h, bd, lane);
        wave_sync();
        unsigned long long sse = 0;
        for (int e = lane; e < P; e += 64) { int rr = ict_inv(jout[e], mode, k); if (cadj) rr = lmcs_scale_inv((int) (int16_t) rr, cadj, bd); int v = rec[e] + rr; v = v < 0 ? 0 : v > mx ? mx : v; rec[e] = (int16_t) v; const int d = org[k * P + e] - v; sse += (unsigned long long) (d * d); }
        dist += (unsigned long long) (p.dist_weight[k] * (double) wave_sum_u64(sse));
        wave_sync();
      }
    } else
    for (int k = 0; k < 2; k++) {
      int16_t *rec = recb + k * P;
      chroma_pred_wave(rec, lm_in_buf(scratch, n), k, fm, w, h, bd, lane);
      wave_sync();
      unsigned long long sse; int cbf;
      wave_code_block<SMALL>(org_tile(scratch, n), k * P, k * P, rec, levb + k * P, wave_tmp(scratch, imin(w, 32) * h, 0), w, h, bd, p.qp_tr_c[k], lane, sse, cbf, (cbfm >> (k + 1)) & 1, nullptr, k + 1, 0, 0, lf, lfm, 0, -1, cadj);
      dist += (unsigned long long) (p.dist_weight[k] * (double) sse);
    }
    if (lane == 0) {
      enc_intra_chroma_pred_mode(cb, mode, L.colm, L.lm_ok);
      enc_bin(cb, (unsigned) !!(cbfm & 2), VX_CTX_QtCbf[1]);
      enc_bin(cb, (unsigned) !!(cbfm & 4), VX_CTX_QtCbf[2] + !!(cbfm & 2));
      if ((p.tools & TOOL_JCCR) && (cbfm & 6)) enc_bin(cb, jm ? 1u : 0u, VX_CTX_JointCbCrFlag + (((cbfm & 2) ? 2 : 0) | ((cbfm & 4) ? 1 : 0)) - 1);
    }
    int fl = 0;
    if (cbfm & 2) { residual_coding_wave<SMALL>(cb, 0, levb, w, h, 1, lane); fl |= lfnst_flags(uni(L.rc_last[0]), w, h); }
    if ((cbfm & 4) && jm != 3) { residual_coding_wave<SMALL>(cb, P, levb + P, w, h, 1, lane); fl |= lfnst_flags(uni(L.rc_last[0]), w, h); }
    if (lane == 0) enc_lfnst_idx(cb, 1, 2 * w, 2 * h, 0, 0, fl, uni((int) L.ps_lfnst));
    // coding_unit() ends with end_of_ctu (EL/CABACWriter.cpp:2118-2141): terminating bin after the last chroma CU of a CTU
    // that does not end the slice; estFracBitsTrm(0) = 0x10c (CL/Contexts.h:129)
    const int endX = L.nx + L.nw, endY = L.ny + L.nh;
    const int lastCtu = (L.ctu_x >> 7) == p.ctus_w - 1 && (L.ctu_y >> 7) == p.ctus_h - 1;
    if (lane == 0 && !lastCtu && ((endX & 127) == 0 || endX == p.pic_w) && ((endY & 127) == 0 || endY == p.pic_h)) cb.bits += 0x10c;
  }
  if (lane == 0) { L.win_idx = 0; L.win_wave = 0; L.wave_slot[0] = 0; L.rd_dist[0] = dist; L.cu_bits = cb.bits; }
}
// OP_REUSE: xReuseCachedResult (EL/EncCu.cpp:5665-5771).  The cached mode and levels of the node are reconstructed against the
// current neighbourhood (DecCu::xReconIntraQT), distortion and CU bits are recomputed from the node's start contexts.  Results are
// left where stage B / the chroma search leave theirs (candidate 0, wave 0, slot 0), so the controller continues at PH_B_DONE.
template <typename T>
__device__ __noinline__ void op_reuse(const VxParams &p_, const VxFrameDev &fd_, uint8_t *scratch)
{
  scratch = uni_p(scratch);
  const VxParams &p = L.par; (void) p_; const VxFrameDev &fd = L.fdv; (void) fd_;
  const int wave = uni(VTX >> 6), lane = VTX & 63;
  const int ch = uni(L.tree_ch), bd = uni(p.bit_depth);
  const int sh = ch ? 1 : 0;
  const int x = uni(L.nx) >> sh, y = uni(L.ny) >> sh, w = uni(L.nw) >> sh, h = uni(L.nh) >> sh, P = w * h, n = ch ? 2 * P : P;
  if (!ch) op_luma_prep<T>(p, fd);
  else {
    for (int c = 1; c <= 2; c++) {
      const void *org = fd.org[c]; const int st = fd.stride[c];
      int16_t *ot = org_tile(scratch, 2 * P);
    for (int i = VTX; i < P; i += NT) { const int r = i >> ilog2i(w), cc = i & (w - 1); ot[(c - 1) * P + i] = (int16_t) ld_px<T>(org, (y + r) * st + x + cc); }
      build_refs<T>(p, fd, c, x, y, w, h, uni(L.cur_tile), 1);
    }
    if (VTX < 2) L.dc_val[VTX] = dc_value(L.refs[VTX][0], L.refs[VTX][1], w, h, 0);
    lmcs_chroma_adj<T>(uni(L.nx), uni(L.ny));
    { const int fm = uni(L.rd[0].mrl); if (fm >= LM_CHROMA && fm <= MDLM_T) cclm_prepare<T>(p, fd, scratch, x, y, w, h); }
  }
  int16_t *levb = slot_lev(scratch, n, 0, 0);
  {
    int lo;
    cache_slot(uni(L.nx), uni(L.ny), uni(L.nw), uni(L.nh), lo);
    const int16_t *cl = (const int16_t *) (scratch + VXD_OFF_CACHE_LEV) + uni(lo);
    for (int i = VTX; i < n; i += NT) levb[i] = cl[i];
  }
  ctx_copy_all(&L.ctxs[CI_W(0)], &L.ctxs[CI_CUR]);
  __threadfence_block();
  __syncthreads();
  if (wave == 0) { if (n <= BUF) reuse_eval<true>(p, scratch, lane, ch, w, h); else reuse_eval<false>(p, scratch, lane, ch, w, h); }
  __syncthreads();
}

// ------------------------------------------------------------------------------------------------ FAST_ALGORITHM (the fork's classifier path)
// OP_FAST: the 26 features of a luma node (EL/EncCu.cpp:863-1123; OpenCV semantics as stated in oracle/orc_fast.c) and the random
// forest's answer (BIN/TEST.py GetPartition).  Samples are staged once as 8-bit values in the LDS candidate slots (free at node
// entry); each wave takes whole cells of the node's 4x4 grid, whose sums give every sub-block variance the features need.
struct FaScratch {                 // overlays L.tmp
  int cell[16][2];                 // Σ, Σ² of the samples of one grid cell
  int wsum[NW][8];                 // per wave: Σ MADP, Σ MADP², Σ of the four saturated gradients, max quarter sum
  int nbs[5][2];                   // Σ, Σ² over the original samples of each neighbour CU
  int leaf[NT];                    // forest: leaf reached in tree t of the current chunk
};
__device__ inline int sat8(int v) { return v < 0 ? 0 : v > 255 ? 255 : v; }
__device__ inline int wave_max_i32(int v) { for (int m = 32; m >= 1; m >>= 1) { const int o = __shfl_xor(v, m); v = o > v ? o : v; } return v; }
// cv::meanStdDev then stddev * stddev, truncated like the reference's int(var)
__device__ inline int var_int(long long s, long long sq, int n)
{
  const double scale = 1.0 / (double) n, mean = (double) s * scale;
  double var = (double) sq * scale - mean * mean;
  if (var < 0) var = 0;
  const double sd = sqrt(var);
  return (int) (sd * sd);
}
template <typename T>
__device__ __noinline__ void op_fast(const VxParams &p_, const VxFrameDev &fd_)
{
  const VxParams &p = L.par; (void) p_; const VxFrameDev &fd = L.fdv; (void) fd_;
  const int tid = VTX, wave = uni(tid >> 6), lane = tid & 63;
  const int x = uni(L.nx), y = uni(L.ny), w = uni(L.nw), h = uni(L.nh), P = w * h, lw = ilog2i(w);
  uint8_t *px = (uint8_t *) &L.wm[0];                                   // 4096 bytes of the (idle) per-wave buffers, the scratch behind them
  FaScratch &fa = *(FaScratch *) ((uint8_t *) &L.wm[0] + 4096);
  const void *org = fd.org[0]; const int st = fd.stride[0];
  for (int i = tid; i < P; i += NT) px[i] = (uint8_t) sat8(ld_px<T>(org, (y + (i >> lw)) * st + x + (i & (w - 1))));
  __threadfence_block();
  __syncthreads();
  {
    int ms = 0, msq = 0, g0 = 0, g1 = 0, g2 = 0, g3 = 0, gmax = 0;
    const int cw = w >> 2, chh = h >> 2, cp = cw * chh, lcw = ilog2i(cw);
    for (int c = wave; c < 16; c += NW) {
      const int cx0 = (c & 3) * cw, cy0 = (c >> 2) * chh;
      int s = 0, sq = 0;
      for (int k = lane; k < cp; k += 64) {
        const int i = cx0 + (k & (cw - 1)), j = cy0 + (k >> lcw);
        // 3x3 neighbourhood: v[] with reflect-101 coordinates for the gradient kernels, in[] = inside the block for MADP (73-134)
        int v[9]; int sum = 0, n = 0;
        for (int dj = -1; dj <= 1; dj++) for (int di = -1; di <= 1; di++) {
          int xx = i + di, yy = j + dj;
          const int inside = xx >= 0 && yy >= 0 && xx < w && yy < h;
          if (xx < 0) xx = -xx;
          if (xx >= w) xx = 2 * w - 2 - xx;
          if (yy < 0) yy = -yy;
          if (yy >= h) yy = 2 * h - 2 - yy;
          v[(dj + 1) * 3 + di + 1] = px[yy * w + xx];
          if (inside && (di | dj)) { n++; }
        }
        const int ctr = v[4];
        for (int dj = -1; dj <= 1; dj++) for (int di = -1; di <= 1; di++) {
          const int xx = i + di, yy = j + dj;
          if ((di | dj) && xx >= 0 && yy >= 0 && xx < w && yy < h) sum += iabs(v[(dj + 1) * 3 + di + 1] - ctr);
        }
        const int m = n == 8 ? sum >> 3 : n == 5 ? sum / 5 : sum / 3;
        ms += m; msq += m * m;
        s += ctr; sq += ctr * ctr;
        // kernels 997-1013: H {-1 0 1; -2 0 2; -1 0 1}, V {1 2 1; 0 0 0; -1 -2 -1}, 45 {0 1 2; -1 0 1; -2 -1 0}, 135 {2 1 0; 1 0 -1; 0 -1 -2}
        const int a0 = sat8(-v[0] + v[2] - 2 * v[3] + 2 * v[5] - v[6] + v[8]);
        const int a1 = sat8(v[0] + 2 * v[1] + v[2] - v[6] - 2 * v[7] - v[8]);
        const int a2 = sat8(v[1] + 2 * v[2] - v[3] + v[5] - 2 * v[6] - v[7]);
        const int a3 = sat8(2 * v[0] + v[1] + v[3] - v[5] - v[7] - 2 * v[8]);
        g0 += a0; g1 += a1; g2 += a2; g3 += a3;
        float t = (float) a0 * 0.25f + (float) a1 * 0.25f;                // Gra_H / 4 + Gra_V / 4 + ... on 8-bit matrices (1027)
        int q = sat8((int) rintf(t));
        t = (float) q + (float) a2 * 0.25f; q = sat8((int) rintf(t));
        t = (float) q + (float) a3 * 0.25f; q = sat8((int) rintf(t));
        gmax = q > gmax ? q : gmax;
      }
      s = wave_sum_i32(s); sq = wave_sum_i32(sq);
      if (lane == 0) { fa.cell[c][0] = s; fa.cell[c][1] = sq; }
    }
    ms = wave_sum_i32(ms); msq = wave_sum_i32(msq); g0 = wave_sum_i32(g0); g1 = wave_sum_i32(g1); g2 = wave_sum_i32(g2); g3 = wave_sum_i32(g3);
    gmax = wave_max_i32(gmax);
    if (lane == 0) { int *o = fa.wsum[wave]; o[0] = ms; o[1] = msq; o[2] = g0; o[3] = g1; o[4] = g2; o[5] = g3; o[6] = gmax; }
  }
  {                                   // get_context 137-163: variance of the original samples of each neighbour CU
    const int nn = uni(L.fa_n);
    for (int k = wave; k < nn; k += NW) {
      const int nx = uni(L.fa_nb[k][0]), ny = uni(L.fa_nb[k][1]), nw = uni(L.fa_nb[k][2]), nh = uni(L.fa_nb[k][3]), lnw = ilog2i(nw);
      int s = 0, sq = 0;
      for (int i = lane; i < nw * nh; i += 64) { const int v = sat8(ld_px<T>(org, (ny + (i >> lnw)) * st + nx + (i & (nw - 1)))); s += v; sq += v * v; }
      s = wave_sum_i32(s); sq = wave_sum_i32(sq);
      if (lane == 0) { fa.nbs[k][0] = s; fa.nbs[k][1] = sq; }
    }
  }
  __threadfence_block();
  __syncthreads();
  if (tid == 0) {
    int *feat = L.fa_feat;
    long long ms = 0, msq = 0, gs[4] = { 0, 0, 0, 0 }; int gmax = 0;
    for (int k = 0; k < NW; k++) { ms += fa.wsum[k][0]; msq += fa.wsum[k][1]; for (int q = 0; q < 4; q++) gs[q] += fa.wsum[k][2 + q]; gmax = imax(gmax, fa.wsum[k][6]); }
    const double G_H = (double) gs[0] / P, G_V = (double) gs[1] / P, G_45 = (double) gs[2] / P, G_135 = (double) gs[3] / P;
    const double Gra = (G_H + G_V + G_45 + G_135) / 4;
    feat[4] = (int) G_H; feat[5] = (int) G_V; feat[6] = (int) G_45; feat[7] = (int) G_135; feat[8] = (int) Gra; feat[9] = gmax;
    // region (cx0..cx1, cy0..cy1) of the cell grid
#define REGV(cx0_, cx1_, cy0_, cy1_) ([&]() { long long s_ = 0, q_ = 0; for (int cy_ = (cy0_); cy_ < (cy1_); cy_++) for (int cx_ = (cx0_); cx_ < (cx1_); cx_++) { s_ += fa.cell[cy_ * 4 + cx_][0]; q_ += fa.cell[cy_ * 4 + cx_][1]; } \
                                        return var_int(s_, q_, ((cx1_) - (cx0_)) * ((cy1_) - (cy0_)) * (P >> 4)); }())
    feat[10] = REGV(0, 4, 0, 4);
    feat[11] = var_int(ms, msq, P);
    { const int B1 = REGV(0, 4, 0, 2), B2 = REGV(0, 4, 2, 4), m = (B1 + B2) / 2; feat[21] = ((B1 - m) * (B1 - m) + (B2 - m) * (B2 - m)) / 2; }
    { const int B1 = REGV(0, 2, 0, 4), B2 = REGV(2, 4, 0, 4), m = (B1 + B2) / 2; feat[22] = ((B1 - m) * (B1 - m) + (B2 - m) * (B2 - m)) / 2; }
    { const int T1 = REGV(0, 4, 0, 1), T2 = REGV(0, 4, 1, 3), T3 = REGV(0, 4, 3, 4), m = (T1 + T2 + T3) / 3; feat[23] = ((T1 - m) * (T1 - m) + (T2 - m) * (T2 - m) + (T3 - m) * (T3 - m)) / 3; }
    { const int T1 = REGV(0, 1, 0, 4), T2 = REGV(1, 3, 0, 4), T3 = REGV(3, 4, 0, 4), m = (T1 + T2 + T3) / 3; feat[24] = ((T1 - m) * (T1 - m) + (T2 - m) * (T2 - m) + (T3 - m) * (T3 - m)) / 3; }
    { const int Q1 = REGV(0, 2, 0, 2), Q2 = REGV(2, 4, 0, 2), Q3 = REGV(0, 2, 2, 4), Q4 = REGV(2, 4, 2, 4), m = (Q1 + Q2 + Q3 + Q4) / 4;
      feat[25] = ((Q1 - m) * (Q1 - m) + (Q2 - m) * (Q2 - m) + (Q3 - m) * (Q3 - m) + (Q4 - m) * (Q4 - m)) / 4; }
#undef REGV
    // neighbour statistics 943-983: variance features 12..14 (the depth features 15..20 were filled by the controller)
    const int nn = L.fa_n;
    int vmx = 0, vmn = 0, vsum = 0;
    for (int k = 0; k < nn; k++) {
      const int v = var_int(fa.nbs[k][0], fa.nbs[k][1], L.fa_nb[k][2] * L.fa_nb[k][3]);
      if (k == 0 || v > vmx) vmx = v;
      if (k == 0 || v < vmn) vmn = v;
      vsum += v;
    }
    feat[12] = vmx; feat[13] = vmn; feat[14] = vsum / nn;
    feat[26] = feat[10] < feat[13] ? 0 : feat[10] > feat[12] ? 2 : 1;          // simple / complex / fuzzy (1127-1138)
    L.fa_row = -1;
    if (p.train_rows) {                                                        // the row of the training dump (oracle/orc_rdo.c fast_partition): features now, label when the node is left
      const unsigned r = atomicAdd(p.train_n, 1u);
      if (r < (unsigned) p.train_cap) { for (int k = 0; k < 27; k++) p.train_rows[(size_t) r * 28 + k] = feat[k]; p.train_rows[(size_t) r * 28 + 27] = -1; L.fa_row = (int) r; }
    }
    if (!(p.tools & TOOL_FAST)) L.fa_res = -1;                                 // features only: the mode list stays as it is
  }
  __threadfence_block();
  __syncthreads();
  if (!(p.tools & TOOL_FAST)) return;
  // forest: one thread per tree walks to its leaf; thread 0 adds the leaf distributions in tree order (sklearn's predict_proba order)
  double acc[8];
  if (tid == 0) for (int c = 0; c < 8; c++) acc[c] = 0;
  const int nt = p.f_ntrees, nc = p.f_nclasses;
  for (int t0 = 0; t0 < nt; t0 += NT) {
    const int t = t0 + tid;
    if (t < nt) {
      int n = p.f_root[t];
      for (;;) {
        const VxForestNode nd = p.f_node[n];
        if (nd.left < 0) break;
        n = ((double) (float) L.fa_feat[nd.feature] <= nd.thr) ? nd.left : nd.right;
      }
      fa.leaf[tid] = n;
    }
    __threadfence_block();
    __syncthreads();
    if (tid == 0) for (int k = 0; k < imin(NT, nt - t0); k++) { const double *v = p.f_value + (size_t) fa.leaf[k] * nc; for (int c = 0; c < nc; c++) acc[c] += v[c]; }
    __syncthreads();
  }
  if (tid == 0) {
    int best = 0;
    for (int c = 1; c < nc; c++) if (acc[c] > acc[best]) best = c;
    L.fa_res = p.f_classes[best];
  }
  __syncthreads();
}

// area copies between the picture (planes + unit map) and the level store / candidate slots
template <typename T>
__device__ __noinline__ void op_save_pic(const VxParams &p_, const VxFrameDev &fd_, uint8_t *scratch, int restore)
{
  scratch = uni_p(scratch); restore = uni(restore);
  const VxParams &p = L.par; (void) p_; const VxFrameDev &fd = L.fdv; (void) fd_;
  const int ch = L.tree_ch, d = L.nd;
  const int x1 = imin(L.nx + L.nw, p.pic_w), y1 = imin(L.ny + L.nh, p.pic_h);
  uint8_t *lvl = scratch + VXD_OFF_STORE + (size_t) d * VXD_STORE_LEVEL;
  const int sh = ch ? 1 : 0, ncomp = ch ? 2 : 1;
  const int X0 = L.nx >> sh, Y0 = L.ny >> sh, Wn = L.nw >> sh, cw = (x1 >> sh) - X0, chh = (y1 >> sh) - Y0;
  for (int k = 0; k < ncomp; k++) {
    const int comp = ch ? k + 1 : 0;
    int16_t *srec = (int16_t *) lvl + k * (Wn * (L.nh >> sh)), *slev = (int16_t *) (lvl + VXD_STORE_REC) + k * (Wn * (L.nh >> sh));
    void *rec = fd.rec[comp]; int16_t *lev = fd.lev[comp]; const int st = fd.stride[comp], ls = fd.lstride[comp];
    const int lcw = pow2_log(cw);
    for (int i = VTX; i < cw * chh; i += NT) {
      const int r = fast_div(i, cw, lcw), c = i - r * cw;
      if (restore) { st_px<T>(rec, (Y0 + r) * st + X0 + c, srec[r * Wn + c]); lev[(Y0 + r) * ls + X0 + c] = slev[r * Wn + c]; }
      else { srec[r * Wn + c] = (int16_t) ld_px<T>(rec, (Y0 + r) * st + X0 + c); slev[r * Wn + c] = lev[(Y0 + r) * ls + X0 + c]; }
    }
  }
  VxUnit *su = (VxUnit *) (lvl + 2 * VXD_STORE_REC);
  const int ux0 = L.nx >> 2, uy0 = L.ny >> 2, ucw = ((x1 + 3) >> 2) - ux0, uch = ((y1 + 3) >> 2) - uy0;
  const int lucw = pow2_log(ucw);
  for (int i = VTX; i < ucw * uch; i += NT) {
    const int r = fast_div(i, ucw, lucw), c = i - r * ucw;
    if (restore) fd.units[ch][(uy0 + r) * p.uw + ux0 + c] = su[r * 32 + c];
    else su[r * 32 + c] = fd.units[ch][(uy0 + r) * p.uw + ux0 + c];
  }
  __threadfence_block();
  __syncthreads();
}
// winner of the intra check (slot B of wave L.win_wave) + its CU record → level store
__device__ __noinline__ void op_save_intra(const VxParams &p_, uint8_t *scratch, const VxUnit &cu)
{
  scratch = uni_p(scratch);
  const VxParams &p = L.par; (void) p_;
  const int ch = L.tree_ch, d = L.nd, sh = ch ? 1 : 0;
  const int W = L.nw >> sh, H = L.nh >> sh, P = W * H;
  uint8_t *lvl = scratch + VXD_OFF_STORE + (size_t) d * VXD_STORE_LEVEL;
  const int wave = L.win_wave, which = L.op_a;
  // (wave NW: the node's winner is the parked ISP candidate)
  const int16_t *rec = wave == NW ? (const int16_t *) (scratch + VXD_OFF_ISP_BEST) : slot_rec(scratch, ch ? 2 * P : P, wave, which);
  const int16_t *lev = wave == NW ? rec + 4096 : slot_lev(scratch, ch ? 2 * P : P, wave, which);
  int16_t *srec = (int16_t *) lvl, *slev = (int16_t *) (lvl + VXD_STORE_REC);
  const int n = ch ? 2 * P : P;
  for (int i = VTX; i < n; i += NT) { srec[i] = rec[i]; slev[i] = lev[i]; }
  if ((p.tools & TOOL_CU_REUSE) && uni(L.op_c)) {      // setFromCs (939-985): the unsplit result is what tryMode(POST_DONT_SPLIT) caches right after
    int lo;
    const int e = uni(cache_slot(L.nx, L.ny, L.nw, L.nh, lo));
    if (e >= 0) {
      int16_t *cl = (int16_t *) (scratch + VXD_OFF_CACHE_LEV) + lo;
      for (int i = VTX; i < n; i += NT) cl[i] = lev[i];
      if (VTX == 0) {
        VxCacheEnt c; c.ss = cu.ss; c.kind = (uint8_t) (ch + 1); c.dir = cu.dir; c.mrl = cu.mrl; c.cbf = cu.cbf; c.depth = cu.depth; c.mts = cu.mts; c.gen = (uint16_t) L.cache_gen;
        ((VxCacheEnt *) (scratch + VXD_OFF_CACHE))[e] = c;
      }
    }
  }
  VxUnit *su = (VxUnit *) (lvl + 2 * VXD_STORE_REC);
  const int ucw = (L.nw + 3) >> 2, uch = (L.nh + 3) >> 2;
  for (int i = VTX; i < ucw * uch; i += NT) { const int r = i >> ilog2i(ucw); su[r * 32 + (i & (ucw - 1))] = cu; }      // unclipped node: power of two
  ctx_copy_all(ctx_ptr(scratch, CTX_BEST, d, 0), &L.ctxs[CI_W(0)]);
  __threadfence_block();
  __syncthreads();
}
__device__ __noinline__ void op_clear_units(const VxParams &p_, const VxFrameDev &fd_)
{
  const VxParams &p = L.par; (void) p_; const VxFrameDev &fd = L.fdv; (void) fd_;
  const int ch = L.tree_ch;
  const int x1 = imin(L.nx + L.nw, p.pic_w), y1 = imin(L.ny + L.nh, p.pic_h);
  const int ux0 = L.nx >> 2, uy0 = L.ny >> 2, ucw = ((x1 + 3) >> 2) - ux0, uch = ((y1 + 3) >> 2) - uy0;
  const int lucw = pow2_log(ucw);
  for (int i = VTX; i < ucw * uch; i += NT) { const int r = fast_div(i, ucw, lucw); fd.units[ch][(uy0 + r) * p.uw + ux0 + (i - r * ucw)].tag = 0; }
  __threadfence_block();
  __syncthreads();
}

// ------------------------------------------------------------------------------------------------ mode controller (thread 0)
__device__ inline int mode_to_split(int m) { return m == ETM_SPLIT_QT ? SPLIT_QT : m == ETM_SPLIT_BT_H ? SPLIT_BH : m == ETM_SPLIT_BT_V ? SPLIT_BV : m == ETM_SPLIT_TT_H ? SPLIT_TH : m == ETM_SPLIT_TT_V ? SPLIT_TV : SPLIT_NONE; }

// EncModeCtrlMTnoRQT::tryMode (EL/EncModeCtrl.cpp:1557-2068), I-slice subset
__device__ __noinline__ int try_mode(const VxParams &p_, int d_, int ch, int mode)
{
  const VxParams &p = L.par; (void) p_;
  Frame &f = L.fr[d_];
  const int impl = implicit_split(p, f, ch);
  if (impl != SPLIT_NONE && mode != ETM_SPLIT_QT) return mode_to_split(mode) == impl;
  else if (impl != SPLIT_NONE) return can_do(p, f, ch, SPLIT_QT);
  if (f.reusing) { if (mode == ETM_RECO_CACHED) return 1; if (mode == ETM_INTRA) return 0; }     // 1585-1598
  const int maxDepth = 7 - ilog2i(p.min_qt[ch]);
  if (mode == ETM_SPLIT_QT && maxDepth <= f.qt) return 0;
  if (mode == ETM_INTRA) { if (f.w * f.h > 4096) return 0; if (f.w > 64 || f.h > 64) return 0; return 1; }
  if (mode == ETM_POST_DONT_SPLIT) return 0;
  const int split = mode_to_split(mode);
  if (!can_do(p, f, ch, split)) {
    if (split == SPLIT_BH) f.did_h = 0;
    if (split == SPLIT_BV) f.did_v = 0;
    if (split == SPLIT_QT) f.did_q = 0;
    return 0;
  }
  const Sum *b = f.has_best ? &f.best : nullptr;
  int feat = -1;
  switch (split) {
    case SPLIT_QT:
      if (!f.qt_before_bt && b) {
        const int maxBTD = p.max_bt_depth[ch];
        if (((b->f_bt == 0 && maxBTD >= 3) || (b->f_bt == 1 && b->l_bt == 1 && maxBTD >= 4)) && (f.w <= 64 && f.h <= 64) && f.did_h && f.did_v) return 0;
      }
      break;
    case SPLIT_BH: feat = 0; break;
    case SPLIT_BV: feat = 1; break;
    case SPLIT_TH:
      if (f.did_h && b && b->f_bt == f.bt && !b->f_cbf) return 0;
      if (!f.do_th) return 0;
      break;
    case SPLIT_TV:
      if (f.did_v && b && b->f_bt == f.bt && !b->f_cbf) return 0;
      if (!f.do_tv) return 0;
      break;
  }
  if (split != SPLIT_QT && f.qt_before_bt && f.did_q && f.max_qt_sub > f.qt + 1) {
    if (feat == 0) f.did_h = 0; else if (feat == 1) f.did_v = 0;
    return 0;
  }
  if (split == SPLIT_QT) f.did_q = 1;
  return 1;
}
__device__ int next_mode(const VxParams &p, int d_, int ch)
{
  Frame &f = L.fr[d_];
  f.nmodes--;
  while (f.nmodes > 0 && !try_mode(p, d_, ch, f.modes[f.nmodes - 1])) f.nmodes--;
  return f.nmodes > 0;
}
__device__ __noinline__ void init_cu_level(const VxParams &p_, const VxFrameDev &fd_, uint8_t *scratch, int d, int ch, int tile)       // initCULevel 1203-1549
{
  const VxParams &p = L.par; (void) p_; const VxFrameDev &fd = L.fdv; (void) fd_;
  Frame &f = L.fr[d];
  prepare_node(p, fd, d, ch, tile);
  const int cuL = f.nb_ok & 1, cuA = f.nb_ok & 2, lq = f.nbL_qt, aq = f.nbA_qt;
  f.qt_before_bt = (uint8_t) (((cuL && cuA && lq > f.qt && aq > f.qt) || (cuL && !cuA && lq > f.qt) || (!cuL && cuA && aq > f.qt)
                   || (!cuA && !cuL && f.w >= 32)) && (f.w > (p.min_qt[ch] << 1)));
  f.do_th = f.do_tv = 1; f.did_h = f.did_v = f.did_q = 0; f.max_qt_sub = 0; f.has_best = 0; f.nmodes = 0;
  if (!f.qt_before_bt) f.modes[f.nmodes++] = ETM_SPLIT_QT;
  if (can_do(p, f, ch, SPLIT_TV)) f.modes[f.nmodes++] = ETM_SPLIT_TT_V;
  if (can_do(p, f, ch, SPLIT_TH)) f.modes[f.nmodes++] = ETM_SPLIT_TT_H;
  if (can_do(p, f, ch, SPLIT_BV)) { f.modes[f.nmodes++] = ETM_SPLIT_BT_V; f.did_v = 1; }
  if (can_do(p, f, ch, SPLIT_BH)) { f.modes[f.nmodes++] = ETM_SPLIT_BT_H; f.did_h = 1; }
  if (f.qt_before_bt) f.modes[f.nmodes++] = ETM_SPLIT_QT;
  f.modes[f.nmodes++] = ETM_POST_DONT_SPLIT;
  f.reusing = (uint8_t) cache_is_valid(p, scratch, L.fr, d, ch);        // 1438-1444
  if (f.reusing) f.modes[f.nmodes++] = ETM_RECO_CACHED;
  f.modes[f.nmodes++] = ETM_INTRA;
  if (!try_mode(p, d, ch, f.modes[f.nmodes - 1])) next_mode(p, d, ch);
}
__device__ int use_mode_result(const VxParams &p, Frame &f, int ch, int mode, const Sum &t)       // useModeResult 2089-2200
{
  if (mode == ETM_SPLIT_QT) f.max_qt_sub = (uint8_t) t.max_qt;
  const int maxMtD = p.max_bt_depth[ch] + f.impl_bt, sh = ch ? 1 : 0;
  if (mode == ETM_SPLIT_BT_H && t.n_cu > 2) { const int h2 = (f.h >> sh) / 2; f.do_th = (uint8_t) (t.f_h < h2 || t.l_h < h2 || f.mt + 1 == maxMtD); }
  else if (mode == ETM_SPLIT_BT_V && t.n_cu > 2) { const int w2 = (f.w >> sh) / 2; f.do_tv = (uint8_t) (t.f_w < w2 || t.l_w < w2 || f.mt + 1 == maxMtD); }
  return t.cost != MAX_DOUBLE && (!f.has_best || t.cost < f.best.cost);
}
// updateCandList (CL/UnitTools.h:261-306)
__device__ void update_cand_list(Cand m, double cost, Cand *list, double *costs, int &size, int fastNum)
{
  int shift = 0;
  const int cur = imin(fastNum, size);
  while (shift < fastNum && shift < cur && cost < costs[cur - 1 - shift]) shift++;
  if (size >= fastNum && shift != 0) {
    for (int i = 1; i < shift; i++) { list[cur - i] = list[cur - 1 - i]; costs[cur - i] = costs[cur - 1 - i]; }
    list[cur - shift] = m; costs[cur - shift] = cost;
  } else if (cur < fastNum) {
    const int pos = size - shift;
    for (int i = size; i > pos; i--) { list[i] = list[i - 1]; costs[i] = costs[i - 1]; }
    list[pos] = m; costs[pos] = cost; size++;
  }
}

// The fork's gate and neighbour lookups (EL/EncCu.cpp:836-933), thread 0: the node qualifies when it lies inside the picture, is smaller
// than the CTU, is not 4x4, has mtDepth < 3 and at least three of the left / left-below / above / above-right / above-left CUs exist.
// Neighbours come from the tile-restricted map (tiles stay independent streams; the reference's getCU is unrestricted).
__device__ __noinline__ int fast_candidates(const VxParams &p_, const VxFrameDev &fd_, const Frame &f, int tile)
{
  const VxParams &p = L.par; (void) p_; const VxFrameDev &fd = L.fdv; (void) fd_;
  const int x = f.x, y = f.y, w = f.w, h = f.h;
  if (!(h < 128 && x + w <= p.pic_w && y + h <= p.pic_h)) return 0;
  if (f.mt == 3 || (w == 4 && h == 4)) return 0;
  int n = 0; int qmx = 0, qmn = 0, qs = 0, mmx = 0, mmn = 0, msum = 0;
#define FA_ADD(u_) { L.fa_nb[n][0] = (u_)->x; L.fa_nb[n][1] = (u_)->y; L.fa_nb[n][2] = (int16_t) (1 << (u_)->lw); L.fa_nb[n][3] = (int16_t) (1 << (u_)->lh); \
                     const int q_ = (u_)->qt, m_ = (u_)->mt; if (!n || q_ > qmx) qmx = q_; if (!n || q_ < qmn) qmn = q_; qs += q_; \
                     if (!n || m_ > mmx) mmx = m_; if (!n || m_ < mmn) mmn = m_; msum += m_; n++; }
  const VxUnit *cuL = get_cu(p, fd, 0, x - 1, y, tile), *cuU = get_cu(p, fd, 0, x, y - 1, tile), *cuLU = get_cu(p, fd, 0, x - 1, y - 1, tile);
  if (cuL) {
    FA_ADD(cuL);
    const VxUnit *cuLD = get_cu(p, fd, 0, x - 1, y + (1 << cuL->lh) + 1, tile);
    if (cuLD && cuLD->y <= y + h) FA_ADD(cuLD);
  }
  if (cuU) {
    FA_ADD(cuU);
    const VxUnit *cuRU = get_cu(p, fd, 0, x + (1 << cuU->lw) + 1, y - 1, tile);
    if (cuRU && cuRU->x < x + w) FA_ADD(cuRU);
  }
  if (cuLU && !((cuLU->y + (1 << cuLU->lh)) > y || (cuLU->x + (1 << cuLU->lw)) > x)) FA_ADD(cuLU);
#undef FA_ADD
  if (n < 3) return 0;
  L.fa_n = n;
  int *feat = L.fa_feat;
  feat[0] = h; feat[1] = w; feat[2] = f.qt; feat[3] = f.mt;
  feat[15] = qmx; feat[16] = qmn; feat[17] = qs / n; feat[18] = mmx; feat[19] = mmn; feat[20] = msum / n;
  return 1;
}
__device__ void post(int op) { L.op = op; }
__device__ void set_node(const Frame &f, int d) { L.nx = f.x; L.ny = f.y; L.nw = f.w; L.nh = f.h; L.nd = d; }
// xCheckRDCostIntra's tail for the node on top of the frame stack (thread 0): winner chosen by the operation; CU-level rate; xCheckBestMode.
// Returns 1 when the result becomes the node's best and has to be saved (OP_SAVE_INTRA / the fused tail of the operation).
// thread 0 after the MIP candidates' SATD operation (EL/IntraSearch.cpp:733-748): every MIP mode into the list (one entry longer), then
// IntraSearch::reduceHadCandList (4331-4406, JVET_O0925 form, FastMIP 1): at most 3 regular candidates, MIP candidates up to half the list or within
// thresholdHadCost of the best, always one MIP; above 8x8 the best of MIP modes {3,4,5} (each in its better orientation) is appended when absent
__device__ __noinline__ void ctrl_mip_merge(int w, int h)
{
  CtlState &S = L.S;
  const int nMip = L.mip_n;
  L.cnt[0] += (unsigned long long) nMip;
  for (int m = 0; m < nMip; m++) update_cand_list(L.cand[m], L.cand_cost[m], S.rdList, S.rdCost, S.rdSize, S.numRd + 1);
  // thresholdHadCost = 1 + 1.4 / sqrt(w * h) (747): w * h = 2^k, so the root is 2^(k/2), times the correctly rounded sqrt(2) for odd k — exact scaling, no device sqrt
  const int k2 = ilog2i(w) + ilog2i(h);
  const double root = (k2 & 1) ? 0x1.6a09e667f3bcdp+0 * (double) (1 << (k2 >> 1)) : (double) (1 << (k2 >> 1));
  const double thr = 1.0 + 1.4 / root;
  const int maxPerType = S.numRd >> 1, size = S.rdSize;
  const double minCost = S.rdCost[0];
  int keepOneMip = size > S.numRd, numConv = 0, numMip = 0, tn = 0;
  for (int idx = 0; idx < size - (keepOneMip ? 0 : 1); idx++) {
    int add;
    if (!(S.rdList[idx].mrl & MIPF)) { add = numConv < 3; numConv += add; }
    else { add = numMip < maxPerType || S.rdCost[idx] < thr * minCost || keepOneMip; keepOneMip = 0; numMip += add; }
    if (add) { S.rdList[tn] = S.rdList[idx]; S.rdCost[tn] = S.rdCost[idx]; tn++; }
  }
  if (w > 8 && h > 8) {
    // MIP modes 3, 4, 5, each in its cheaper orientation, in ascending SATD-stage cost (updateCandList into an empty list of 3 = a stable
    // sort: an equal cost stays behind the earlier mode); scalars only, no private arrays in this thread-0 function
    const int transpOff = nMip / 2;
    int m0 = 3 + (L.cand_cost[3 + transpOff] < L.cand_cost[3] ? transpOff : 0);
    int m1 = 4 + (L.cand_cost[4 + transpOff] < L.cand_cost[4] ? transpOff : 0);
    int m2 = 5 + (L.cand_cost[5 + transpOff] < L.cand_cost[5] ? transpOff : 0);
    double k0 = L.cand_cost[m0], k1 = L.cand_cost[m1], k2 = L.cand_cost[m2];
    if (k1 < k0) { const int tm = m0; m0 = m1; m1 = tm; const double tk = k0; k0 = k1; k1 = tk; }
    if (k2 < k1) {
      { const int tm = m1; m1 = m2; m2 = tm; const double tk = k1; k1 = k2; k2 = tk; }
      if (k1 < k0) { const int tm = m0; m0 = m1; m1 = tm; const double tk = k0; k0 = k1; k1 = tk; }
    }
    const int n0 = tn;
    for (int idx = 0; idx < 3; idx++) {
      const int mm = idx == 0 ? m0 : idx == 1 ? m1 : m2;
      int incl = 0;
      for (int k = 0; k < n0; k++) incl |= S.rdList[k].mode == mm && S.rdList[k].mrl == MIPF;
      if (!incl) { S.rdList[tn].mode = (uint8_t) mm; S.rdList[tn].mrl = MIPF; S.rdCost[tn] = 0; tn++; break; }      // fastMip: one
    }
  }
  S.rdSize = tn; S.numRd = tn;
}
// ---- ISP candidate selection (thread 0).  Sub-partitions a tested (mode, split) completed, -1 when the pair has not been tested (ISPTestedModesInfo::getNumCompletedSubParts)
__device__ int isp_tested_at(int split, int mode)
{
  const CtlState &S = L.S;
  for (int i = 0; i < S.ispNT; i++) if (S.ispTMode[i] == mode && (S.ispTInfo[i] >> 4) == split) return i;
  return -1;
}
__device__ inline int isp_parts(int split, int mode) { return L.S.ispPartsOf[split - 1][mode]; }
__device__ inline int isp_nth_tested(int split, int k)    // the k-th mode (k = 0, 1) tested with this split (m_ispTestedModes[...].intraMode in test order)
{
  return k < L.S.ispNTested[split - 1] ? L.S.ispFirst[split - 1][k] : -1;
}
// xSortISPCandList (EL/IntraSearch.cpp:4614-4717): planar, the best angular mode of the regular full-RD results, the other angular ones by cost, DC; then up to three
// modes of the SATD-stage list that are not among them.  bestNonISP: the best regular cost; stops both splits when ISP cannot win (ISPFast 1)
__device__ __noinline__ void ctrl_isp_sort(double bestCostSoFar, double bestNonISP)
{
  CtlState &S = L.S;
  if (bestNonISP > bestCostSoFar * 1.4) { S.ispStop[0] = S.ispStop[1] = 1; return; }
  // the regular line-0, non-MIP results in list order, then std::sort by cost = insertion sort for at most 16 entries (equal costs keep their order)
  int n = 0;
  for (int c = 0; c < L.n_rd; c++) if (L.rd[c].mrl == 0) S.ispReg[n++] = (uint8_t) c;
  for (int i = 1; i < n; i++) { const uint8_t m = S.ispReg[i]; const double cst = L.rd_cost[m]; int j = i - 1; while (j >= 0 && cst < L.rd_cost[S.ispReg[j]]) { S.ispReg[j + 1] = S.ispReg[j]; j--; } S.ispReg[j + 1] = m; }
  S.ispRegN = (int8_t) n;
  int bestAngle = -1;
  for (int i = 0; i < n; i++) if (L.rd[S.ispReg[i]].mode > DC) { bestAngle = L.rd[S.ispReg[i]].mode; break; }
  int nl = 0, dc = 0;
  S.ispList[nl++] = PLANAR;
  if (bestAngle != -1) S.ispList[nl++] = (uint8_t) bestAngle;
  for (int i = 0; i < n; i++) { const int m = L.rd[S.ispReg[i]].mode; if (m != PLANAR && m != bestAngle) { if (m > DC) S.ispList[nl++] = (uint8_t) m; else if (m == DC) dc = 1; } }
  if (dc) S.ispList[nl++] = DC;
  S.ispNOrig = (int8_t) nl;
  for (int k = 0, added = 0; k < S.ispHadN && added < 3; k++) {
    int in = 0; for (int i = 0; i < nl; i++) in |= S.ispList[i] == S.ispHad[k];
    if (!in) { S.ispList[nl++] = S.ispHad[k]; added++; }
  }
  S.ispNList = (int8_t) nl;
}
// xGetNextISPMode (4467-4589) with xFindAlreadyTestedNearbyIntraModes (4591-4612): 1 and the next candidate, 0 when the place of the RD list stays unused
__device__ __noinline__ int ctrl_isp_next(int w, int h, int &mode, int &split)
{
  CtlState &S = L.S;
  int nxt;
  if (!S.ispStop[0] && !S.ispStop[1]) nxt = S.ispPrev == 1 ? 2 : 1;
  else if (!S.ispStop[0]) nxt = 1;
  else if (!S.ispStop[1]) nxt = 2;
  else return 0;
  const int st = nxt - 1, maxParts = S.ispNumTotal[st];
  if (S.ispNTested[st] >= 2) {
    int mode1 = isp_nth_tested(nxt, 0); mode1 = mode1 == DC ? -1 : mode1;
    const int n1 = mode1 != -1 ? isp_parts(nxt, mode1) : -1;
    int mode2 = isp_nth_tested(nxt, 1); mode2 = mode2 == DC ? -1 : mode2;
    const int n2 = mode2 != -1 ? isp_parts(nxt, mode2) : -1;
    if (n1 != -1 && n2 != -1 && n1 < maxParts && n2 < maxParts) { S.ispStop[st] = 1; return 0; }
    const int other = nxt == 1 ? 2 : 1;
    const int nOther = mode2 != -1 ? isp_parts(other, mode2) : -1;
    int stopThis = 0;
    if (nOther != -1 && n2 != -1) {
      if (nOther > n2) stopThis = 1;
      else if (nOther == n2 && nOther == maxParts) {
        const int a = isp_tested_at(nxt, mode2), b = isp_tested_at(other, mode2);
        const double cThis = (a >= 0 && (S.ispTInfo[a] & 15) == maxParts) ? S.ispTCost[a] : -1, cOther = (b >= 0 && (S.ispTInfo[b] & 15) == maxParts) ? S.ispTCost[b] : -1;
        if (cThis == MAX_DOUBLE || cOther < cThis * 1.3) stopThis = 1;
      }
    }
    if (stopThis) { S.ispStop[st] = 1; return 0; }
  }
  if (S.ispCandIdx[st] < S.ispNList) {
    const int cand = S.ispList[S.ispCandIdx[st]];
    S.ispCandIdx[st]++;
    if (S.ispCandIdx[st] > S.ispNOrig) { if (S.ispBestSplit != nxt || S.ispBestMode == PLANAR) return 0; }      // the extra modes only while ISP is winning
    int test = 1;
    if (cand >= DC && maxParts > 2 && S.ispNTested[st] >= 2) {
      const int window = cand > DC ? 5 : 1, limit = (w << ilog2i(h)) >= 256 ? maxParts - 1 : 2;
      int lm = -1, rm = -1;
      for (int k = 1; k <= window; k++) {
        const int off = cand - 2 - k;
        const int l_ = off < 0 ? 67 + off : cand - k, r_ = cand > DC ? ((cand - 2 + k) % 65) + 2 : PLANAR;
        const int lf = l_ != cand ? isp_parts(nxt, l_) >= 0 : 0, rf = r_ != cand ? isp_parts(nxt, r_) >= 0 : 0;
        if (lf || rf) { lm = lf ? l_ : -1; rm = rf ? r_ : -1; break; }
      }
      const int nl = lm != -1 ? isp_parts(nxt, lm) : -1, nr = rm != -1 ? isp_parts(nxt, rm) : -1;
      const int nref = nl > nr ? nl : nr;
      if (nref > 0) test = nref > limit;
    }
    if (test) { mode = cand; split = nxt; return 1; }
  }
  return 0;
}
// after the regular (and MIP) candidates of the first pass: the ISP state of the node (EL/IntraSearch.cpp:355-384, 1032-1039).  The node's budget is what the parent's
// split loop left (EL/EncCu.cpp:2503-2516: intra is a node's first mode, so no own result caps it yet)
__device__ __noinline__ void ctrl_isp_begin(const Frame &f)
{
  CtlState &S = L.S;
  const double bestReg = L.rd_cost[L.win_idx];
  S.ispBestRd = bestReg; S.ispCurBest = f.max_cost < bestReg ? f.max_cost : bestReg; S.noIspCost = bestReg;
  S.ispNT = 0; S.ispBestMode = -1; S.ispBestSplit = 0; S.ispNOrig = -1; S.ispNList = 0; S.ispPrev = 0;
  for (int k = 0; k < 2; k++) { S.ispStop[k] = 0; S.ispCandIdx[k] = 0; S.ispNTested[k] = 0; }
  { uint32_t *t = (uint32_t *) S.ispPartsOf; for (int k = 0; k < 34; k++) t[k] = 0xffffffffu; }
  S.ispNumTotal[0] = (int8_t) (f.h / isp_split_dim(f.w, f.h, 1)); S.ispNumTotal[1] = (int8_t) (f.w / isp_split_dim(f.w, f.h, 0));
  L.isp_win = 0;
  ctrl_isp_sort(S.ispCurBest, bestReg);
}
// The exit logic of xIntraCodingLumaISP (3206-3268) on what a wave reported, under the limit the candidate is judged with (never above the one its wave stopped with, so
// every sub-partition the logic looks at has been run, and every rate it adds has been computed): sub-partitions completed, cost (MAX: early exit / not below the limit),
// cbfs, TUs quantised, and the candidate's result (setModeResults 1241-1245, 1270-1288, 1308-1326) into the ISP state
// Returns 0 when the logic asks for a sub-partition the wave did not run (possible only under a lowered limit and a rounding-level tie: a skipped rate makes the running
// cost smaller than the one the wave stopped on): the candidate is then evaluated again on its own under the current limit.
__device__ __noinline__ int ctrl_isp_result(const Frame &f, int bw)
{
  CtlState &S = L.S;
  const VxParams &p = L.par;
  const IspRes &R = L.isp_res[bw];
  const int mode = R.mode, split = R.split, st = split - 1, maxParts = S.ispNumTotal[st], n = maxParts;
  const double limit = S.ispCurBest;
  double cost = 0; int early = 0, tucbf = 0, ntu = 0, evals = 0, noCbf = 0;
  unsigned long long dist = 0, bits = 0;
  for (int k = 0; k < n; k++) {
    if (k >= R.nrun) return 0;
    const int cbf = (R.cbf >> k) & 1; const unsigned long long d = R.d[k];
    evals++;
    if (k == n - 1 && !tucbf && !(cbf)) { ntu = n; noCbf = 1; break; }      // 2990-2996
    if (cbf) tucbf |= 1 << k;
    ntu = k + 1;
    unsigned long long fb = 0;
    if (rd_cost(p, bits, dist + d) > limit) early = 1; else fb = R.fb[k];
    cost += rd_cost(p, fb, d); dist += d; bits += fb;
    if (k + 1 < n) {
      if (cost > limit) { early = 1; break; }
      const double thr = n == 2 ? 0.95 : k + 1 == 1 ? 0.83 : 0.91;
      if (cost > limit * thr) { early = 1; break; }
    }
  }
  int valid = 0, first = tucbf & 1; double rcost = MAX_DOUBLE;
  if (!noCbf && !early) {
    rcost = rd_cost(p, bits, dist);
    if (rcost < limit) { valid = 1; first = tucbf != 0; }      // 3257-3268: cbf at depth 0 of every TU = any sub-partition coded
    else rcost = MAX_DOUBLE;
  }
  const double rc = first ? rcost : MAX_DOUBLE;
  { const int psz = isp_split_dim(f.w, f.h, split == 1), tpix = split == 1 ? f.w * psz : psz * f.h; L.cnt[1] += (unsigned long long) evals; L.cnt[2] += (unsigned long long) (evals * tpix); }
  if (S.ispNTested[st] < 2) S.ispFirst[st][S.ispNTested[st]] = (int8_t) mode;
  S.ispPartsOf[st][mode] = (int8_t) ntu;
  S.ispTMode[S.ispNT] = (uint8_t) mode; S.ispTInfo[S.ispNT] = (uint8_t) ((split << 4) | ntu); S.ispTCost[S.ispNT] = ntu == maxParts ? rc : MAX_DOUBLE; S.ispNT++; S.ispNTested[st]++;
  if (ntu == maxParts && rc < S.ispBestRd) { S.ispBestMode = (int8_t) mode; S.ispBestSplit = (int8_t) split; }
  if (valid && first && rcost < S.ispBestRd) {
    S.ispBestRd = rcost; L.isp_win = 1; S.ispPark = (int8_t) bw;
    S.ispWinMode = (uint8_t) mode; S.ispWinSplit = (uint8_t) split; S.ispWinTucbf = (uint8_t) tucbf; S.ispWinDist = dist; S.ispWinBits = bits;
    if (rcost < S.ispCurBest) S.ispCurBest = rcost;
  }
  return 1;
}
// The next batch: the candidate the reference's order asks for and, on the other waves, the ones it is likely to ask for next (the list walked by both splits in turn,
// original entries only) - a wrong guess costs idle waves some work, never a result
__device__ __noinline__ void ctrl_isp_batch(int reqMode, int reqSplit)
{
  CtlState &S = L.S;
  int n = 0;
  L.isp_res[n].mode = (uint8_t) reqMode; L.isp_res[n].split = (uint8_t) reqSplit; L.isp_res[n].used = 0; n++;
  int ci[2] = { S.ispCandIdx[0], S.ispCandIdx[1] }, prev = reqSplit, idle = 0;
  while (n < NW && idle < 2) {
    int nxt;
    if (!S.ispStop[0] && !S.ispStop[1]) nxt = prev == 1 ? 2 : 1;
    else if (!S.ispStop[0]) nxt = 1;
    else if (!S.ispStop[1]) nxt = 2;
    else break;
    prev = nxt;
    const int st = nxt - 1;
    if (ci[st] >= S.ispNOrig) { idle++; continue; }
    idle = 0;
    const int cand = S.ispList[ci[st]++];
    if (isp_parts(nxt, cand) >= 0) continue;
    L.isp_res[n].mode = (uint8_t) cand; L.isp_res[n].split = (uint8_t) nxt; L.isp_res[n].used = 0; n++;
  }
  L.isp_nb = (uint8_t) n;
}
// thread 0: the full-RD list of a luma pass is in S.rdList[0, numRd): MPM append (777-802) unless the list comes from the DCT-II pass (an MTS pass), the
// MIP re-ordering (1097-1122) or removal of MIP candidates (1123-1141), then stage B
__device__ __noinline__ void ctrl_post_stage_b(Frame &f, int append_mpm)
{
  CtlState &S = L.S;
  if (append_mpm) {
    for (int j = 0; j < L.mpm_n; j++) {
      int incl = 0;
      for (int i = 0; i < S.numRd; i++) incl |= (S.rdList[i].mode == L.mpm[j] && S.rdList[i].mrl == 0);
      if (!incl) { S.rdList[S.numRd].mode = (uint8_t) L.mpm[j]; S.rdList[S.numRd].mrl = 0; S.rdCost[S.numRd] = 0; S.numRd++; }
    }
    if (S.testIsp) for (int j = 0; j < L.mpm_n; j++) {      // 803-820: the MPMs join the list saved for ISP as well
      int incl = 0;
      for (int i = 0; i < S.ispHadN; i++) incl |= S.ispHad[i] == L.mpm[j];
      if (!incl) S.ispHad[S.ispHadN++] = (uint8_t) L.mpm[j];
    }
    if (S.lfOn && S.mtsUsage == 1 && S.lf == 0) { S.mtsNum = S.numRd; for (int i = 0; i < S.numRd; i++) S.mtsList[i] = S.rdList[i]; }      // 884-889 (only the list of lfnstIdx 0 is read again)
  }
  int n = 0;
  if (S.testMip) {
    for (int i = 0; i < S.numRd; i++) if (!(S.rdList[i].mrl & MIPF)) { S.idxOf[n] = (uint8_t) i; L.rd_src[n] = S.rdSrc[i]; L.rd[n++] = S.rdList[i]; }
    for (int i = 0; i < S.numRd; i++) if (S.rdList[i].mrl & MIPF) { S.idxOf[n] = (uint8_t) i; L.rd_src[n] = S.rdSrc[i]; L.rd[n++] = S.rdList[i]; }
  } else
    for (int i = 0; i < S.numRd; i++) if (!(S.rdList[i].mrl & MIPF)) { S.idxOf[n] = (uint8_t) n; L.rd_src[n] = S.rdSrc[i]; L.rd[n++] = S.rdList[i]; }
  L.n_rd = n;
  if (S.lf == 0 && S.mts == 0) for (int c = 0; c < n && c < 16; c++) S.inv0[S.idxOf[c]] = (uint8_t) c;      // where the first pass evaluates each place of its list
  f.phase = PH_B_DONE;
  L.isp_wait = 0; L.isp_win = 0;
  if (S.testIsp) { f.phase = PH_ISP; L.isp_wait = 1; S.ispSlot = -1; }      // sixteen reserved places behind the regular and MIP candidates (1032-1039)
  post(OP_STAGE_B);
}
// thread 0: step the (transform group, lfnstIdx, mtsFlag) loops of xCheckRDCostIntra (2453-2775) after a pass; returns 0 when no pass is left
__device__ __noinline__ int ctrl_next_pass()
{
  CtlState &S = L.S;
  int g = S.grp, lf = S.lf, m = S.mts + 1;
  if (S.ispBreak) { S.ispBreak = 0; m = S.considerMts + 1; }      // 2737-2740: the ISP winner ended the mtsFlag loop of this lfnstIdx
  for (;;) {
    const int endM = S.considerMts;
    if (m > endM) {                                        // the mtsFlag loop of this lfnstIdx is over
      int endOfLf = 0;
      if (S.skipOther) { S.startLf = (int8_t) lf; S.endLf = (int8_t) lf; endOfLf = 1; }      // 2754-2759
      else { lf++; m = g > 0; if (lf > S.endLf) endOfLf = 1; }
      while (endOfLf) {                                    // end of a transform group (2763-2773), on to the next one that is checked
        if (g < 3) {
          S.grpCheck[g + 1] = 0;
          if (S.bestSel[g] && S.considerMts) S.grpCheck[g + 1] = (uint8_t) ((S.bestMts != 0 || S.bestLf != 0) && S.dct2Cost / S.grpBest[g] < 1.001);
        }
        g++;
        if (g >= 4) return 0;
        if (S.considerMts && !S.skipMts2 && S.grpCheck[g]) { lf = S.startLf; m = 1; endOfLf = lf > S.endLf; }
      }
      continue;
    }
    if (m > 0 && lf > 0) { m++; continue; }                // JVET_O0368: no LFNST with an MTS pair
    S.grp = (int8_t) g; S.lf = (int8_t) lf; S.mts = (int8_t) m;
    return 1;
  }
}
__device__ __noinline__ int ctrl_b_done(const VxParams &p_, const VxFrameDev &fd_)
{
  const VxParams &p = L.par; (void) p_; const VxFrameDev &fd = L.fdv; (void) fd_;
  const int ch = L.tree_ch, tile = L.cur_tile, sh = ch ? 1 : 0, d = L.d;
  Frame &f = L.fr[d];
      const int best = L.win_idx, ww = L.win_wave;
      if (ch) L.cnt[0] += (unsigned long long) L.lm_nsatd;
      if (!ch && L.n_rd) { int ex = 0; for (int k = 0; k < NW; k++) ex += L.mts_evals[k]; L.cnt[1] += (unsigned long long) ex; L.cnt[2] += (unsigned long long) (ex * f.w * f.h); }      // transform candidates beyond DCT2
      if (ch && L.n_rd && (p.tools & TOOL_JCCR)) { int ex = 0; for (int k = 0; k < NW; k++) ex += L.mts_evals[k]; L.cnt[1] += (unsigned long long) ex; L.cnt[2] += (unsigned long long) (ex * (f.w >> 1) * (f.h >> 1)); }      // joint chroma blocks
      L.cnt[1] += (unsigned long long) (ch ? 2 * L.n_rd : L.n_rd); L.cnt[2] += (unsigned long long) (ch ? 2 * L.n_rd * ((f.w >> 1) * (f.h >> 1)) : L.n_rd * f.w * f.h);
      L.op_a = L.wave_slot[ww];                         // slot of that wave holding the winner's reco / levels
      Sum &t = f.temp;
      t.dist = L.rd_dist[best];
      VxUnit &cu = L.cu;
      cu.dir = L.rd[best].mode; cu.mrl = ch ? 0 : L.rd[best].mrl; cu.cbf = L.rd_cbf[best]; cu.mts = (uint8_t) (((ch && !(p.tools & TOOL_JCCR)) ? 0 : L.rd_mts[best]) | (L.ps_lfnst << 4));      // luma: tu.mtsIdx, chroma: tu.jointCbCr; lfnstIdx rides in bits 4-5
      const int ispWin = !ch && L.isp_win;              // 0 / cu.ispMode of an ISP winner (search) or of the cached CU (reuse); bits 6-7 of the unit's mts byte, the sub-partitions' cbfs in bits 4-7 of its cbf byte
      if (ispWin && L.n_rd) {                           // the ISP candidate parked by OP_ISP beat the regular winner (1308-1326)
        cu.dir = L.S.ispWinMode; cu.mrl = 0; cu.cbf = (uint8_t) (1 | (L.S.ispWinTucbf << 4)); cu.mts = (uint8_t) (L.S.ispWinSplit << 6);
        t.dist = L.S.ispWinDist; L.cu_bits = L.S.ispWinBits;      // cu_pred_data + cu_residual of an ISP CU = what the sub-partitions were priced with (no residual_lfnst_mode, EL/CABACWriter.cpp:3994)
        L.win_wave = NW; L.op_a = 0;                    // the parked ISP tiles (op_save_intra)
      }
      // cu_pred_data + cu_residual bits (EL/EncCu.cpp:2593-2619) and the CU's end contexts are left in cu_bits / wctx[0]
      // by the operation (luma: identical to the stage-B syntax from the same start contexts)
      if (!ch && f.cur_mode == ETM_RECO_CACHED) cu.mts |= f.r_mts & 0xc0;      // cu.ispMode of the cached CU
      t.bits = L.cu_bits;
      const double costWithoutSplitFlags = rd_cost(p, t.bits, t.dist);      // 2622-2629: also bestIspCost of an ISP winner
      Cab cb; cb.ci = CI_W(0); cb.bits = 0;
      enc_split_cu_mode(p, fd, cb, f, ch, tile, SPLIT_NONE);          // xEncodeDontSplit 5649-5662
      t.bits += cb.bits;
      t.cost = rd_cost(p, t.bits, t.dist);
      t.n_cu = 1; t.f_bt = t.l_bt = f.bt; t.f_cbf = cu.cbf != 0; t.f_w = t.l_w = (int16_t) (f.w >> sh); t.f_h = t.l_h = (int16_t) (f.h >> sh); t.max_qt = f.qt; t.valid = 1;
#ifdef VX_TRACE
      fprintf(stderr, "RES ch %d node %d %d %dx%d lf %d mts %d dir %d isp %d tucbf %d cost %.3f dist %llu\n", ch, f.x, f.y, f.w, f.h, L.S.lfOn ? L.S.lf : L.ps_lfnst, L.S.lfOn ? L.S.mts : 0, cu.dir, cu.mts >> 6, cu.cbf >> 4, t.cost, (unsigned long long) t.dist);
#endif
      f.phase = PH_ADVANCE;
      CtlState &S = L.S;
      if (S.lfOn) {                                     // a pass of the LFNST / MTS loop (2633-2760)
        if (!ch && S.mtsUsage == 1 && S.lf == 0) {      // 1262-1290: the DCT-II pass's costs steer the candidate choice of the MTS passes
          for (int c = 0; c < L.n_rd; c++) S.modeCost[S.idxOf[c]] = L.rd_cost[c];
          S.bestCost0 = L.rd_cost[best]; S.bestValid0 = 1;
        }
        if (S.lf && !(L.rd_lfl[best] & 1) && cu.cbf) t.cost = MAX_DOUBLE;      // 2633-2645: an LFNST index that cannot be signalled
        if (S.mts == 0 && S.lf == 0) S.dct2Cost = t.cost;
        if (!(cu.mts >> 6)) S.noIspCost = t.cost;        // useModeResult(ETM_INTRA), EL/EncModeCtrl.cpp:2120-2123: bestCostMtsFirstPassNoIsp
        S.skipOther = (int8_t) !cu.cbf;                 // checkSkipOtherLfnst (EL/EncModeCtrl.cpp:2070-2087): the intra passes are a node's first modes, its condition always holds
        f.phase = PH_NEXT_PASS;
      }
      // 2714-2741 (ISPFast 1): an ISP winner of the first pass that beats the best regular mode by a margin ends the LFNST / MTS passes of this CU (evaluated after
      // xCheckBestMode in the reference; it reads nothing that call changes except endLfnstIdx, which a transform-skip winner - never an ISP CU - sets)
      if (S.lfOn && (S.considerMts > 0 || S.endLf > 0) && (cu.mts >> 6) && !ch && !S.mts && !S.lf) {
        const double threshold = 1.4, lfnstThreshold = 1.01 * threshold;
        if (S.noIspCost > costWithoutSplitFlags * lfnstThreshold) S.endLf = S.lf;
        if (S.noIspCost > costWithoutSplitFlags * threshold) { S.skipMts2 = 1; S.ispBreak = 1; }
      }
      if (use_mode_result(p, f, ch, ETM_INTRA, t)) {
        if (S.lfOn) { S.grpBest[S.grp] = t.cost; S.bestSel[S.grp] = 1; S.bestMts = S.mts; S.bestLf = S.lf; }      // 2696-2701
        if (S.lfOn && !ch && (cu.mts & 7) == 1 && ilog2i(f.w) + ilog2i(f.h) >= 6) S.endLf = 0;                       // 2702-2712: a transform-skip winner of at least 64 samples ends the LFNST passes
        f.best = t; f.has_best = 1;
        set_node(f, d);
        L.op_c = f.nmodes > 1;                          // ETM_POST_DONT_SPLIT still on the stack (not a single predicted mode): its setFromCs caches this result
        return 1;                                       // store ← winner slot, ctxBest[d] ← wctx[0]
      }
      return 0;
}

// one controller step: runs until a parallel operation is posted (returns) or the CTU tree is finished (posts OP_DONE)
// the label of a node's training row: the partition the search chose there (PartSplit code, 0 = not split; -1: no encoding), read from the split series the winner left in the
// unit map.  Out of line: the controller loop keeps its registers
__device__ __noinline__ void ctrl_train_label(const Frame &f, int row)
{
  const VxParams &p = L.par; const VxFrameDev &fd = L.fdv;
  int label = -1;
  if (f.best.cost != MAX_DOUBLE) label = (int) ((fd.units[0][(f.y >> 2) * p.uw + (f.x >> 2)].ss >> (f.depth * 5)) & 31);
  p.train_rows[(size_t) row * 28 + 27] = label;
}
__device__ __attribute__((always_inline)) inline void control_step(const VxParams &p, const VxFrameDev &fd, uint8_t *scratch)
{
  CtlState &S = L.S;
  const int ch = L.tree_ch, tile = L.cur_tile, sh = ch ? 1 : 0;
  for (;;) {
    if (L.d < 0) { post(OP_DONE); return; }
    Frame &f = L.fr[L.d];
    const int d = L.d;
#if VVCX_STAMP
    const long long tph = STAMP(); const int phs = f.phase;
#if !defined(VVCX_STAMP_DQ) && !defined(VVCX_STAMP_ISP) && !defined(VVCX_STAMP_PASS)
    struct PhStamp { long long t; int ph; __device__ ~PhStamp() { const int slot = ph < 12 ? 16 + ph : ph == PH_EXIT2 ? 1 : ph == PH_A3_DONE ? 28 : ph == PH_PASS ? 29 : ph == PH_NEXT_PASS ? 31 : 47; PROF(slot) += (unsigned long long) (STAMP() - t); PROF(30) += 1; } } phstamp = { tph, phs };
#endif
#endif
#ifdef VX_TRACE
    fprintf(stderr, "  step d %d phase %d nmodes %d\n", d, f.phase, f.nmodes);
#endif
    switch (f.phase) {
    case PH_ENTER: {                                    // xCompressCU entry (EL/EncCu.cpp:727-1286)
      L.cnt[3]++;
      init_cu_level(p, fd, scratch, d, ch, tile);
      f.best.cost = MAX_DOUBLE; f.best.dist = 0; f.best.bits = 0; f.best.valid = 0;
      if (f.nmodes == 0) { f.phase = PH_EXIT2; break; }
      f.phase = PH_RUN; f.ctx_dirty = 0;
      L.pre_copy_d = d;                                   // m_CurrCtx->start = ctx: done by the dispatch that runs the node's first operation
      L.train_row[d] = -1;
      if (((p.tools & TOOL_FAST) || p.train_rows) && !ch && fast_candidates(p, fd, f, tile)) { f.phase = PH_FAST_DONE; set_node(f, d); post(OP_FAST); return; }
      break;
    }
    case PH_FAST_DONE: {                                // EL/EncCu.cpp:1126-1217: replace the mode stack by the predicted mode if the controller accepts it
      const int res = L.fa_res;
      L.train_row[d] = L.fa_row;
      if (res >= 0 && res <= 5) {
        const int mode = res == 0 ? ETM_INTRA : res == 1 ? ETM_SPLIT_QT : res == 2 ? ETM_SPLIT_BT_H : res == 3 ? ETM_SPLIT_BT_V : res == 4 ? ETM_SPLIT_TT_H : ETM_SPLIT_TT_V;
        int valid = try_mode(p, d, ch, mode);                                             // tryModeMaster 1199
        if (L.fa_feat[26] >= 1 && res == 0) valid = 0;
        if (valid) { f.modes[0] = (uint8_t) mode; f.nmodes = 1; }                         // ChangeTestMode (EL/EncModeCtrl.cpp:56-93)
      }
      f.phase = PH_RUN;
      break;
    }
    case PH_RUN: {
      const int mode = f.modes[f.nmodes - 1];
      f.cur_mode = (uint8_t) mode;
      set_node(f, d);
      if (mode == ETM_INTRA || mode == ETM_RECO_CACHED) {
        // CU record (partitioner.setCUData, EL/EncCu.cpp:2478-2496)
        VxUnit &cu = L.cu;
        cu.ss = f.ss; cu.x = (int16_t) (f.x >> sh); cu.y = (int16_t) (f.y >> sh); cu.lw = (uint8_t) ilog2i(f.w >> sh); cu.lh = (uint8_t) ilog2i(f.h >> sh);
        cu.qt = f.qt; cu.mt = f.mt; cu.bt = f.bt; cu.depth = f.depth; cu.dir = 0; cu.mrl = 0; cu.cbf = 0; cu.mts = 0; cu.tag = (uint16_t) (tile + 1);
        if (mode == ETM_RECO_CACHED) {                  // xReuseCachedResult (EL/EncCu.cpp:5665-5771)
          L.rd[0].mode = f.r_dir; L.rd[0].mrl = f.r_mrl; L.rd_cbf[0] = f.r_cbf; L.rd_mts[0] = (uint8_t) (f.r_mts & 7); L.n_rd = 0;
          L.isp_split = (uint8_t) (ch ? 0 : f.r_mts >> 6); L.isp_tucbf = (uint8_t) (f.r_cbf >> 4);      // a cached ISP CU
          L.ps_lfnst = (int8_t) ((f.r_mts >> 4) & 3); L.ps_mts = 0; L.ps_grp = 0; S.lfOn = 0; L.isp_wait = 0; L.isp_win = 0;
          if (!ch) {                                    // MPM list for intra_luma_pred_modes
            int Ld, Ad; luma_neighbours(p, fd, f.x, f.y, f.w, f.h, tile, Ld, Ad);
            derive_mpms(Ld, Ad, L.mpm);
          }
          if (ch) {
            const int cx = f.x + (f.w >> 1), cy = f.y + (f.h >> 1);
            L.colm = unit_ldir(fd.units[0][(cy >> 2) * p.uw + (cx >> 2)]);
            L.rd[0].mrl = (uint8_t) (f.r_dir == DM_CHROMA ? L.colm : f.r_dir);    // final mode
            L.lm_ok = cclm_allowed(p, fd, f.x, f.y, f.ss, f.depth); L.lm_nsatd = 0;
          }
          f.phase = PH_B_DONE;
          post(OP_REUSE); return;
        }
        // the pass loop of xCheckRDCostIntra (2417-2452): without LFNST a single pass
        S.lfOn = (int8_t) ((p.tools & TOOL_LFNST) != 0);
        S.grp = 0; S.lf = 0; S.mts = 0; S.startLf = 0; S.skipOther = 0; S.bestMts = 0; S.bestLf = 0; S.bestValid0 = 0; S.lfSaved = 0;
        S.endLf = (int8_t) ((!S.lfOn || (ch && (f.w < 8 || f.h < 8)) || f.w > 64 || f.h > 64) ? 0 : 2);                 // 2431-2449
        S.considerMts = (int8_t) (S.lfOn && (p.tools & TOOL_MTS) && !ch && f.w <= 32 && f.h <= 32);                      // 2409 (as the MTS passes' loop bound)
        S.dct2Cost = MAX_DOUBLE; S.testIsp = 0; S.skipMts2 = 0; S.ispBreak = 0; S.noIspCost = MAX_DOUBLE; L.isp_wait = 0; L.isp_win = 0;
        L.spec_n = 0;
        for (int i = 0; i < 4; i++) { S.grpCheck[i] = 1; S.bestSel[i] = 0; S.grpBest[i] = MAX_DOUBLE; }
        f.phase = PH_PASS;
        break;
      }
      if (mode == ETM_POST_DONT_SPLIT) { f.phase = PH_ADVANCE; break; }
      // ---- xCheckModeSplit (EL/EncCu.cpp:1918-2399)
      const int split = mode_to_split(mode);
      f.cur_split = (uint8_t) split;
      {
        Cab cb; cb.ci = CI_W(0); cb.bits = 0;
        const int ids0 = VX_CTX_SplitFlag;           // split contexts occupy [SplitFlag, Split12Flag+4)
        for (int k = ids0; k < VX_CTX_Split12Flag + 4; k++) { L.ctxs[CI_W(0)].s0[k] = L.ctxs[CI_CUR].s0[k]; L.ctxs[CI_W(0)].s1[k] = L.ctxs[CI_CUR].s1[k]; }
        enc_split_cu_mode(p, fd, cb, f, ch, tile, split);
        const double factor = p.qp > 30 ? 1.1 : 1.075;
        const double cost = rd_cost(p, (uint64_t) ((double) cb.bits + ((double) f.best.bits / factor)), (uint64_t) ((double) f.best.dist / factor));
        if (cost > f.best.cost) { f.phase = PH_ADVANCE; break; }
      }
      {
        Sum &t = f.temp; t.cost = 0; t.dist = 0; t.bits = 0; t.n_cu = 0; t.max_qt = 0; t.valid = 0;
        const int x = f.x, y = f.y, w = f.w, h = f.h;
        switch (split) {
          case SPLIT_QT: f.nparts = 4; for (int i = 0; i < 4; i++) { f.pw[i] = (int16_t) (w >> 1); f.ph[i] = (int16_t) (h >> 1); f.px[i] = (int16_t) (x + ((i & 1) ? w >> 1 : 0)); f.py[i] = (int16_t) (y + ((i >= 2) ? h >> 1 : 0)); } break;
          case SPLIT_BH: f.nparts = 2; for (int i = 0; i < 2; i++) { f.px[i] = (int16_t) x; f.pw[i] = (int16_t) w; f.ph[i] = (int16_t) (h >> 1); f.py[i] = (int16_t) (y + i * (h >> 1)); } break;
          case SPLIT_BV: f.nparts = 2; for (int i = 0; i < 2; i++) { f.py[i] = (int16_t) y; f.ph[i] = (int16_t) h; f.pw[i] = (int16_t) (w >> 1); f.px[i] = (int16_t) (x + i * (w >> 1)); } break;
          case SPLIT_TH: f.nparts = 3; for (int i = 0; i < 3; i++) { f.px[i] = (int16_t) x; f.pw[i] = (int16_t) w; } f.ph[0] = f.ph[2] = (int16_t) (h >> 2); f.ph[1] = (int16_t) (h >> 1); f.py[0] = (int16_t) y; f.py[1] = (int16_t) (y + (h >> 2)); f.py[2] = (int16_t) (y + (h >> 2) + (h >> 1)); break;
          default:       f.nparts = 3; for (int i = 0; i < 3; i++) { f.py[i] = (int16_t) y; f.ph[i] = (int16_t) h; } f.pw[0] = f.pw[2] = (int16_t) (w >> 2); f.pw[1] = (int16_t) (w >> 1); f.px[0] = (int16_t) x; f.px[1] = (int16_t) (x + (w >> 2)); f.px[2] = (int16_t) (x + (w >> 2) + (w >> 1)); break;
        }
        f.child = 0; f.first = 1; f.ctx_dirty = 1;
        f.phase = PH_CHILD;
        post(OP_CLEAR_UNITS); return;          // tempCS->initStructData: nothing of this node is coded yet
      }
    }
    case PH_PASS: {                                     // one (lfnstIdx, mtsFlag) pass: estIntraPredLumaQT / estIntraPredChromaQT with cu.lfnstIdx, cu.mtsFlag set
      L.ps_lfnst = S.lf; L.ps_mts = S.mts; L.ps_grp = S.grp;
      set_node(f, d);
      if (!ch) {
        S.mtsUsage = (int8_t) ((f.w <= 32 && f.h <= 32 && (p.tools & TOOL_MTS)) ? ((S.lfOn && S.mts == 1) ? 2 : 1) : 0);           // 330-341
        if (S.lf == 0 && S.mts == 0) {
          // MPM list of the node (PU::getIntraMPMs neighbours, CL/UnitTools.cpp:516-532) and its number of MIP modes
          int Ld, Ad; luma_neighbours(p, fd, f.x, f.y, f.w, f.h, tile, Ld, Ad);
          derive_mpms(Ld, Ad, L.mpm); L.mpm_n = (Ld == Ad) ? 1 : 2;
        }
        S.testMip = (int8_t) (L.mip_n && (S.lf == 0 || (f.w >= 16 && f.h >= 16)));
        S.testIsp = (int8_t) (S.lfOn && S.lf == 0 && S.mts == 0 && L.isp_ok);      // 355-384: ISP is tested in the pass without LFNST and MTS
        if (S.testIsp) S.ispHadN = 0;                                               // 404-418 (JVET_O0925) + allowLfnstWithMip
        if (S.lf == 0 && S.mts == 0) {
          // stage A candidate list, phase 1: 35 even modes + MRL MPM candidates (costs are order independent)
          int n = 0;
          for (int m = 0; m < 67; m++) { S.checked[m] = 0; if (m > DC && (m & 1)) continue; L.cand[n].mode = (uint8_t) m; L.cand[n].mrl = 0; n++; S.checked[m] = 1; }
          if ((f.y & 127) != 0 && (p.tools & 1))
            for (int r = 1; r < 3; r++) for (int k = 1; k < 6; k++) { L.cand[n].mode = (uint8_t) L.mpm[k]; L.cand[n].mrl = (uint8_t) (r == 1 ? 1 : 3); n++; }
          L.n_cand = n;
          S.numRd = L.t.mode_num[(ilog2i(f.w) - 2) * 6 + (ilog2i(f.h) - 2)];
          // EL/IntraSearch.cpp:469-477: the regular list is kept longer while the MIP candidates compete for it
          if (S.testMip) S.numRd += imax(S.numRd, ilog2i(imin(f.w, f.h)) - 1);
          f.phase = PH_A1_DONE;
          L.op_a = 0; L.op_b = n; L.op_c = 1; post(OP_LUMA_PREP); return;      // prep, then stage A on [0,n)
        }
        if (S.mtsUsage == 2) {                          // 891-916: an MTS pass re-tests the DCT-II pass's candidates whose cost stayed close to its best (FastLFNST 1)
          S.numRd = 0;
          if (S.bestValid0) {
            const int k2 = ilog2i(f.w) + ilog2i(f.h);
            const double root = (k2 & 1) ? 0x1.6a09e667f3bcdp+0 * (double) (1 << (k2 >> 1)) : (double) (1 << (k2 >> 1));
            const double thr = 1.0 + 1.4 / root;
            for (int i = 0; i < S.mtsNum; i++) if (S.modeCost[i] <= thr * S.bestCost0) { S.rdSrc[S.numRd] = S.inv0[i]; S.rdList[S.numRd++] = S.mtsList[i]; }
          } else { S.numRd = S.mtsNum; for (int i = 0; i < S.mtsNum; i++) { S.rdSrc[i] = S.inv0[i]; S.rdList[i] = S.mtsList[i]; } }
          ctrl_post_stage_b(f, 0); return;
        }
        // an LFNST pass: the SATD-stage list of the first pass (763-775), then the MPMs again
        S.numRd = S.lfNum; { const int n = imin(S.lfSize, S.lfNum); for (int i = 0; i < n; i++) { S.rdList[i] = S.lfList[i]; S.rdCost[i] = 0; } }
        ctrl_post_stage_b(f, 1); return;
      } else {
        // chroma candidate modes (PU::getIntraChromaCandModes, CL/UnitTools.cpp:840-873), LM modes disabled
        const int cx = f.x + (f.w >> 1), cy = f.y + (f.h >> 1);
        const int lm = unit_ldir(fd.units[0][(cy >> 2) * p.uw + (cx >> 2)]);      // getCoLocatedIntraLumaMode 949-960
        L.colm = lm;
        int list[8] = { PLANAR, VER, HOR, DC, LM_CHROMA, MDLM_L, MDLM_T, DM_CHROMA };
        for (int i = 0; i < 4; i++) if (lm == list[i]) { list[i] = VDIA; break; }
        L.lm_ok = cclm_allowed(p, fd, f.x, f.y, f.ss, f.depth); L.lm_nsatd = 0;
        int n = 0;
        for (int i = 0; i < 8; i++) {
          if (!L.lm_ok && list[i] >= LM_CHROMA && list[i] <= MDLM_T) continue;      // EL/IntraSearch.cpp:1588-1591
          L.rd[n].mode = (uint8_t) list[i]; L.rd[n].mrl = (uint8_t) (list[i] == DM_CHROMA ? lm : list[i]); n++;
        }
        L.n_rd = n;
        f.phase = PH_B_DONE;
        post(OP_CHROMA_RD); return;
      }
    }
    case PH_ISP: {                                      // the reserved places of the RD list: each asks xGetNextISPMode for the next ISP candidate (1181-1192); the candidates are
      // evaluated in batches of up to NW, one per wave, ahead of that order (ctrl_isp_batch), and judged in it (ctrl_isp_result)
      ISP_T(p0);
      if (S.ispSlot < 0) { ctrl_isp_begin(f); S.ispSlot = 0; S.ispHave = 0; S.ispPark = -1; L.isp_nb = 0; }
      set_node(f, d);
      { ISP_T(p1); ISP_ADD0(24, p0, p1); }
      for (;;) {
        if (S.ispHave) {
          int bw = -1;
          for (int i = 0; i < L.isp_nb; i++) if (!L.isp_res[i].used && L.isp_res[i].mode == (uint8_t) S.ispReqMode && L.isp_res[i].split == (uint8_t) S.ispReqSplit) bw = i;
          if (bw < 0) {                                 // not evaluated yet: the tiles of a new best are kept first, then the next batch
            if (S.ispPark >= 0) { L.isp_park = (uint8_t) S.ispPark; S.ispPark = -1; post(OP_ISP_PARK); return; }
            { ISP_T(b0); ctrl_isp_batch(S.ispReqMode, S.ispReqSplit); ISP_T(b1); ISP_ADD0(25, b0, b1); ISP_ADD0(29, 0, 1); }
            L.isp_limit = S.ispCurBest;
            post(OP_ISP); return;
          }
          L.isp_res[bw].used = 1;
          { ISP_T(r0); const int okr = ctrl_isp_result(f, bw); ISP_T(r1); ISP_ADD0(26, r0, r1); if (!okr) continue; }          // (evaluated again: the entry is used up, the request stays)
          S.ispHave = 0;
        }
        if (S.ispSlot >= 16) break;
        int mode = 0, split = 0;
        S.ispSlot++;
        ISP_T(n0); const int okn = ctrl_isp_next(f.w, f.h, mode, split); { ISP_T(n1); ISP_ADD0(27, n0, n1); ISP_ADD0(28, 0, 1); }
        if (!okn) { S.ispPrev = 3; if (S.ispStop[0] && S.ispStop[1]) S.ispSlot = 16; continue; }      // both splits stopped: the remaining places ask in vain (most of the 16 do)
        S.ispPrev = (int8_t) split; S.ispReqMode = (int8_t) mode; S.ispReqSplit = (int8_t) split; S.ispHave = 1;
      }
      if (S.ispPark >= 0) { L.isp_park = (uint8_t) S.ispPark; S.ispPark = -1; post(OP_ISP_PARK); return; }
      L.isp_wait = 0; f.phase = PH_B_DONE; post(OP_ISP_END); return;      // the node's intra decision with the ISP winner, if any
    }
    case PH_NEXT_PASS: {
      if (ctrl_next_pass()) { VxUnit &cu = L.cu; cu.dir = 0; cu.mrl = 0; cu.cbf = 0; cu.mts = 0; f.phase = PH_PASS; }
      else f.phase = PH_ADVANCE;
      break;
    }
    case PH_A1_DONE: {                                  // EL/IntraSearch.cpp:489-623; the top-numRd list was selected by the operation
      L.cnt[0] += (unsigned long long) L.n_cand;
#ifdef VX_TRACE
      fprintf(stderr, "DEV node %d %d %dx%d numRd %d isp_ok %d list", f.x, f.y, f.w, f.h, S.numRd, L.isp_ok); for (int i = 0; i < S.numRd; i++) fprintf(stderr, " %d:%d(%.1f)", S.rdList[i].mode, S.rdList[i].mrl, S.rdCost[i]); fprintf(stderr, "\n");
#endif
      int n = L.n_cand; S.n_a2 = 0;
      for (int i = 0; i < S.numRd; i++) {                 // +-1 of the survivors, from a snapshot of the list (parentCandList 574)
        const int pm = S.rdList[i].mode;
        if (pm > (DC + 1) && pm < 66)
          for (int s2 = -1; s2 <= 1; s2 += 2) { const int m = pm + s2; if (!S.checked[m]) { S.checked[m] = 1; L.cand[n].mode = (uint8_t) m; L.cand[n].mrl = 0; n++; S.n_a2++; } }
      }
      f.phase = PH_A2_DONE;
      L.op_a = L.n_cand; L.op_b = n; L.op_c = 0; post(OP_STAGE_A); return;     // evaluates [n1, n) (possibly empty) and makes the final selection
    }
    case PH_A2_DONE:
    case PH_A3_DONE: {                                  // [703-748 MIP candidates,] 777-802 MPM append, then stage B
      const int testMip = S.testMip;
      if (f.phase == PH_A2_DONE) {
        L.cnt[0] += (unsigned long long) S.n_a2;
        if (S.lfOn && testMip && !(f.w >= 16 && f.h >= 16)) {      // 681-698: the LFNST passes of this CU run without MIP: keep the regular list for them
          S.lfNum = L.t.mode_num[(ilog2i(f.w) - 2) * 6 + (ilog2i(f.h) - 2)]; S.lfSize = imin(S.rdSize, S.lfNum);
          for (int i = 0; i < S.lfSize; i++) S.lfList[i] = S.rdList[i];
          S.lfSaved = 1;
        }
        if (testMip) {                                  // every MIP mode by SATD; the regular candidates are in the list, their buffer is free
          for (int m = 0; m < L.mip_n; m++) { L.cand[m].mode = (uint8_t) m; L.cand[m].mrl = MIPF; }
          f.phase = PH_A3_DONE;
          L.op_a = 0; L.op_b = L.mip_n; L.op_c = 2; post(OP_STAGE_A); return;
        }
      } else ctrl_mip_merge(f.w, f.h);
      if (S.lfOn && !S.lfSaved) {                       // 750-761: the list the LFNST passes start from
        S.lfNum = S.numRd; S.lfSize = imin(S.rdSize, 16);
        for (int i = 0; i < S.lfSize; i++) S.lfList[i] = S.rdList[i];
        S.lfSaved = 1;
      }
      ctrl_post_stage_b(f, 1); return;
    }
    case PH_B_DONE: {                                   // only reached when an operation was dispatched without the fused tail (see after_intra_op)
      if (ctrl_b_done(p, fd)) { post(OP_SAVE_INTRA); return; }
      break;
    }
    case PH_CHILD: {                                    // children loop of xCheckModeSplit (2065-2177)
      if (f.child >= f.nparts) {
        // 2297-2333: split flag bits from the contexts left by the last child, cost, xCheckBestMode
        Sum &t = f.temp;
        f.impl_checked = 0;
        const int enforceQT = implicit_split(p, f, ch) == SPLIT_QT;
        if (!enforceQT) { Cab cb; cb.ci = CI_CUR; cb.bits = 0; enc_split_cu_mode(p, fd, cb, f, ch, tile, f.cur_split); t.bits += cb.bits; }
        t.cost = rd_cost(p, t.bits, t.dist); t.valid = 1;
        if (use_mode_result(p, f, ch, f.cur_mode, t)) {
          f.best = t; f.has_best = 1;
          f.phase = PH_SPLIT_SAVED;
          set_node(f, d);
          L.op_a = CTX_BEST; L.op_b = CTX_CUR; L.op_c = d; post(OP_CTX_COPY); return;
        }
        f.phase = PH_ADVANCE; break;
      }
      const int i = f.child;
      if (f.px[i] >= p.pic_w || f.py[i] >= p.pic_h) { f.child++; break; }
      Frame &c = L.fr[d + 1];
      c.x = f.px[i]; c.y = f.py[i]; c.w = f.pw[i]; c.h = f.ph[i];
      c.depth = (uint8_t) (f.depth + 1); c.last_split = f.cur_split; c.part_idx = (uint8_t) i; c.impl_checked = 0; c.impl_split = 0;
      c.ss = f.ss | ((uint64_t) f.cur_split << (f.depth * 5));
      if (f.cur_split == SPLIT_QT) { c.qt = (uint8_t) (f.qt + 1); c.bt = 0; c.mt = 0; c.impl_bt = f.impl_bt; }
      else {
        const int isImpl = f.cur_split == implicit_split(p, f, ch);
        const int tt = f.cur_split == SPLIT_TH || f.cur_split == SPLIT_TV;
        c.qt = f.qt; c.mt = (uint8_t) (f.mt + 1); c.bt = (uint8_t) (f.bt + (tt ? (i == 1 ? 1 : 2) : 1)); c.impl_bt = (uint8_t) (f.impl_bt + (isImpl ? 1 : 0));
      }
      double newMax = MAX_DOUBLE;
      if (!ch) { const double a = f.best.cost - rd_cost(p, f.temp.bits, f.temp.dist); newMax = f.max_cost < a ? f.max_cost : a; }
      if (newMax < 0.0) newMax = 0.0;
      c.max_cost = newMax;
      c.phase = PH_ENTER;
      f.phase = PH_CHILD_RET;
      L.d = d + 1;
      break;
    }
    case PH_CHILD_RET: {
      const Sum &sub = L.fr[d + 1].best;
      Sum &t = f.temp;
      // sub.cost == MAX cannot happen in I slices (every in-picture leaf can be intra coded)
      t.dist += sub.dist; t.bits += sub.bits;
      if (f.first) { t.f_bt = sub.f_bt; t.f_cbf = sub.f_cbf; t.f_w = sub.f_w; t.f_h = sub.f_h; f.first = 0; }
      t.l_bt = sub.l_bt; t.l_w = sub.l_w; t.l_h = sub.l_h;
      t.n_cu = (int16_t) (t.n_cu + sub.n_cu); t.max_qt = (int16_t) imax(t.max_qt, sub.max_qt);
      f.child++;
      f.phase = PH_CHILD;
      break;
    }
    case PH_SPLIT_SAVED: { f.phase = PH_ADVANCE; set_node(f, d); post(OP_SAVE_PIC); return; }
    case PH_INTRA_SAVED: { f.phase = PH_ADVANCE; break; }
    case PH_ADVANCE: {                                  // ctx ← start (xCheckBestMode 721), next mode
      if (next_mode(p, d, ch)) f.phase = PH_RUN; else f.phase = PH_EXIT;
      if (!f.ctx_dirty) break;                            // intra / pruned split: the estimator still holds the start contexts
      f.ctx_dirty = 0;
      L.op_a = CTX_CUR; L.op_b = CTX_START; L.op_c = d; post(OP_CTX_COPY); return;
    }
    case PH_EXIT: {                                     // EL/EncCu.cpp:1533-1583
      if (f.best.cost == MAX_DOUBLE) { f.phase = PH_EXIT2; break; }
      f.phase = PH_EXIT2;
      set_node(f, d);
      L.op_a = CTX_CUR; L.op_b = CTX_BEST; L.op_c = d; L.op_d = 1; post(OP_RESTORE_PIC); return;   // picture ← bestCS, ctx ← best
    }
    case PH_EXIT2: {
      if (L.train_row[d] >= 0) ctrl_train_label(f, L.train_row[d]);
      L.d = d - 1; break;
    }
    }
  }
}

// final estimator pass over the coded CTU (CABACWriter::coding_tree_unit 254-309 / coding_tree 474-984): advances
// L.ctxs[CI_CUR] for the next CTU of the stream.  Thread 0 only.
template <typename T, bool WR>
__device__ __noinline__ void walk_tree(const VxParams &p_, const VxFrameDev &fd_, Cab &cb, int ch, int tile, Frame *st_, int d, int16_t *lv)
{
  const VxParams &p = L.par; (void) p_; const VxFrameDev &fd = L.fdv; (void) fd_;
  Frame *st = L.fr; (void) st_;
  // iterative pre-order walk with an explicit stack of frames st[d..]
  int top = d;
  st[top].child = 0; st[top].phase = 0;
  while (top >= d) {
    Frame &f = st[top];
    if (f.phase == 0) {
      const VxUnit *u = &fd.units[ch][(f.y >> 2) * p.uw + (f.x >> 2)];
      const int split = (int) ((u->ss >> (f.depth * 5)) & 31);
      prepare_node(p, fd, top, ch, tile);                  // st is L.fr
      enc_split_cu_mode<WR>(p, fd, cb, f, ch, tile, split);
      if (!split) {
        const int sh = ch ? 1 : 0, W = f.w >> sh, H = f.h >> sh;
        if (!ch) {
          int Ld, Ad; luma_neighbours(p, fd, f.x, f.y, f.w, f.h, tile, Ld, Ad);
          derive_mpms(Ld, Ad, L.mpm);
          if (u->mts >> 6) {                            // an ISP CU: transform_tree splits it (EL/CABACWriter.cpp:3311-3330); each luma cbf with the previous sub-partition's as context, the last one inferred after all-zero ones (3574-3600); no residual_lfnst_mode (3994)
            const int isp = u->mts >> 6, hor = isp == 1, psz = isp_split_dim(W, H, hor), tw = hor ? W : psz, th = hor ? psz : H, n = hor ? H / psz : W / psz, tucbf = u->cbf >> 4;
            enc_intra_luma_pred_mode<WR>(cb, f.y, u->dir, 0, isp);
            for (int k = 0, sofar = 0; k < n; k++) {
              const int ox = hor ? 0 : k * tw, oy = hor ? k * th : 0, c = (tucbf >> k) & 1;
              if (!(k == n - 1 && !sofar)) enc_bin<WR>(cb, (unsigned) c, VX_CTX_QtCbf[0] + 2 + (k ? (tucbf >> (k - 1)) & 1 : 0));
              if (c) { for (int yy = 0; yy < th; yy++) for (int xx = 0; xx < tw; xx++) lv[yy * tw + xx] = fd.lev[0][(f.y + oy + yy) * fd.lstride[0] + f.x + ox + xx]; residual_coding<WR>(cb, lv, tw, th, 0, (uint16_t *) (lv + 4096)); }
              sofar |= c;
            }
            top--; continue;
          }
          enc_intra_luma_pred_mode<WR>(cb, f.y, u->dir, u->mrl);
          enc_bin<WR>(cb, u->cbf & 1, VX_CTX_QtCbf[0]);
          int fl = 0;
          if (u->cbf & 1) { for (int yy = 0; yy < H; yy++) for (int xx = 0; xx < W; xx++) lv[yy * W + xx] = fd.lev[0][(f.y + yy) * fd.lstride[0] + f.x + xx]; enc_tu_mts<WR>(cb, W, H, u->mts & 7); if ((u->mts & 7) == 1) rc_ts_serial<WR>(cb, lv, W, H); else { residual_coding<WR>(cb, lv, W, H, 0, (uint16_t *) (lv + 4096), (u->mts & 7) > 1); fl = lfnst_flags(L.rc_last[0], W, H); } }
          enc_lfnst_idx<WR>(cb, 0, W, H, (u->mrl & MIPF) != 0, (u->cbf & 1) && (u->mts & 7) != 0, fl, (u->mts >> 4) & 3);
        } else {
          int fl = 0;
          enc_intra_chroma_pred_mode<WR>(cb, u->dir, unit_ldir(fd.units[0][((f.y + (f.h >> 1)) >> 2) * p.uw + ((f.x + (f.w >> 1)) >> 2)]), cclm_allowed(p, fd, f.x, f.y, u->ss, u->depth));
          enc_bin<WR>(cb, (unsigned) !!(u->cbf & 2), VX_CTX_QtCbf[1]);
          enc_bin<WR>(cb, (unsigned) !!(u->cbf & 4), VX_CTX_QtCbf[2] + !!(u->cbf & 2));
          const int jm = (p.tools & TOOL_JCCR) ? (u->mts & 7) : 0;
          if ((p.tools & TOOL_JCCR) && (u->cbf & 6)) enc_bin<WR>(cb, jm ? 1u : 0u, VX_CTX_JointCbCrFlag + (((u->cbf & 2) ? 2 : 0) | ((u->cbf & 4) ? 1 : 0)) - 1);
          for (int c = 1; c <= 2; c++) if ((u->cbf & (1 << c)) && !(c == 2 && jm == 3)) {
            for (int yy = 0; yy < H; yy++) for (int xx = 0; xx < W; xx++) lv[yy * W + xx] = fd.lev[c][((f.y >> 1) + yy) * fd.lstride[c] + (f.x >> 1) + xx];
            residual_coding<WR>(cb, lv, W, H, 1, (uint16_t *) (lv + 4096));
            fl |= lfnst_flags(L.rc_last[0], W, H);
          }
          enc_lfnst_idx<WR>(cb, 1, f.w, f.h, 0, 0, fl, (u->mts >> 4) & 3);
        }
        top--; continue;
      }
      f.cur_split = (uint8_t) split;
      const int x = f.x, y = f.y, w = f.w, h = f.h;
      switch (split) {
        case SPLIT_QT: f.nparts = 4; for (int i = 0; i < 4; i++) { f.pw[i] = (int16_t) (w >> 1); f.ph[i] = (int16_t) (h >> 1); f.px[i] = (int16_t) (x + ((i & 1) ? w >> 1 : 0)); f.py[i] = (int16_t) (y + ((i >= 2) ? h >> 1 : 0)); } break;
        case SPLIT_BH: f.nparts = 2; for (int i = 0; i < 2; i++) { f.px[i] = (int16_t) x; f.pw[i] = (int16_t) w; f.ph[i] = (int16_t) (h >> 1); f.py[i] = (int16_t) (y + i * (h >> 1)); } break;
        case SPLIT_BV: f.nparts = 2; for (int i = 0; i < 2; i++) { f.py[i] = (int16_t) y; f.ph[i] = (int16_t) h; f.pw[i] = (int16_t) (w >> 1); f.px[i] = (int16_t) (x + i * (w >> 1)); } break;
        case SPLIT_TH: f.nparts = 3; for (int i = 0; i < 3; i++) { f.px[i] = (int16_t) x; f.pw[i] = (int16_t) w; } f.ph[0] = f.ph[2] = (int16_t) (h >> 2); f.ph[1] = (int16_t) (h >> 1); f.py[0] = (int16_t) y; f.py[1] = (int16_t) (y + (h >> 2)); f.py[2] = (int16_t) (y + (h >> 2) + (h >> 1)); break;
        default:       f.nparts = 3; for (int i = 0; i < 3; i++) { f.py[i] = (int16_t) y; f.ph[i] = (int16_t) h; } f.pw[0] = f.pw[2] = (int16_t) (w >> 2); f.pw[1] = (int16_t) (w >> 1); f.px[0] = (int16_t) x; f.px[1] = (int16_t) (x + (w >> 2)); f.px[2] = (int16_t) (x + (w >> 2) + (w >> 1)); break;
      }
      f.child = 0; f.phase = 1;
    }
    // next child
    if (f.child >= f.nparts) { top--; continue; }
    const int i = f.child++;
    if (f.px[i] >= p.pic_w || f.py[i] >= p.pic_h) continue;
    Frame &c = st[top + 1];
    c.x = f.px[i]; c.y = f.py[i]; c.w = f.pw[i]; c.h = f.ph[i];
    c.depth = (uint8_t) (f.depth + 1); c.last_split = f.cur_split; c.part_idx = (uint8_t) i; c.impl_checked = 0;
    if (f.cur_split == SPLIT_QT) { c.qt = (uint8_t) (f.qt + 1); c.bt = 0; c.mt = 0; c.impl_bt = f.impl_bt; }
    else {
      f.impl_checked = 0;
      const int isImpl = f.cur_split == implicit_split(p, f, ch);
      const int tt = f.cur_split == SPLIT_TH || f.cur_split == SPLIT_TV;
      c.qt = f.qt; c.mt = (uint8_t) (f.mt + 1); c.bt = (uint8_t) (f.bt + (tt ? (i == 1 ? 1 : 2) : 1)); c.impl_bt = (uint8_t) (f.impl_bt + (isImpl ? 1 : 0));
    }
    c.phase = 0; c.child = 0;
    top++;
  }
}
template <typename T>
__device__ __noinline__ void advance_ctx_ctu(const VxParams &p_, const VxFrameDev &fd_, int tile, int ctu_x, int ctu_y)
{
  tile = uni(tile); ctu_x = uni(ctu_x); ctu_y = uni(ctu_y);
  const VxParams &p = L.par; (void) p_; const VxFrameDev &fd = L.fdv; (void) fd_;
  Cab cb; cb.ci = CI_CUR; cb.bits = 0;
  uint8_t *scratch_ = p.scratch + (size_t) blockIdx.x * p.scratch_per_stream;
  int16_t *lv = (int16_t *) (scratch_ + VXD_OFF_SLOTS);      // level tile (up to 64x64) and scan table of the estimator pass: wave 0's big-block scratch
  // 128x128 root: implicit QT for both trees (no bins); luma / chroma sub-trees interleaved per 64x64 (867-908)
  for (int q = 0; q < 4; q++) {
    const int qx = ctu_x + ((q & 1) ? 64 : 0), qy = ctu_y + ((q >= 2) ? 64 : 0);
    if (qx >= p.pic_w || qy >= p.pic_h) continue;
    for (int ch = 0; ch < (p.chroma ? 2 : 1); ch++) {
      Frame &f = L.fr[1];
      f.x = (int16_t) qx; f.y = (int16_t) qy; f.w = 64; f.h = 64; f.depth = 1; f.qt = 1; f.bt = 0; f.mt = 0; f.impl_bt = 0; f.last_split = SPLIT_QT; f.part_idx = (uint8_t) q;
      if (p.payload) walk_tree<T, true>(p, fd, cb, ch, tile, L.fr, 1, lv);     // CABACWriter::coding_tree_unit on the real coder
      else walk_tree<T, false>(p, fd, cb, ch, tile, L.fr, 1, lv);
    }
  }
}

// slice_data writer of a (frame, tile) stream, thread 0.  Kept out of line so that the stream loop's register budget is not touched.
// (plain ints, not the descriptor by reference: taking its address would move it - and every later use of it in the stream loop - to scratch memory)
__device__ __noinline__ void writer_begin(const VxParams &p, int sidx, int done_before)      // resume where the last launch stopped
{
  L.aw_out = p.payload + p.payload_off[sidx]; L.aw_cap = p.payload_cap[sidx];
  if (done_before == 0) arith_start(); else L.aw = ((const Arith *) p.arith_state)[sidx];
}
__device__ __noinline__ void writer_end_of_ctu(int lastOfTile, int lastTile)   // end_of_ctu (EL/CABACWriter.cpp:2118-2141), end of the brick (EL/EncSlice.cpp:1975-1990)
{
  if (!(lastOfTile && lastTile)) arith_trm(0);
  if (lastOfTile) { arith_trm(1); arith_finish(); }
}
__device__ __noinline__ void writer_suspend(const VxParams &p, int sidx) { ((Arith *) p.arith_state)[sidx] = L.aw; }

// The search of one tree of one CTU: thread 0 steps the mode controller, everybody executes the operation it posts.
// Its own function: what is live across these calls (p, fd, scratch) fits the callee-saved registers; inlined into the
// stream loop, the loop's bookkeeping was spilled to scratch around every operation.
// Tail of the operations that evaluate the intra mode of a node (stage B, chroma RD, cached-result reuse): the controller's decision and,
// when the result is accepted, the save that used to be a dispatch of its own (one controller round trip per evaluated node less).
// Out of line: run_tree's barrier loop must contain exactly one thread-0 section of its own (see there).
__device__ __noinline__ void after_intra_op(const VxParams &p_, const VxFrameDev &fd_, uint8_t *scratch)
{
  scratch = uni_p(scratch);
  if (uni(VTX >> 6) == 0) { if ((VTX & 63) == 0) L.do_save = ctrl_b_done(p_, fd_); }      // wave-uniform entry, lane 0 inside (see run_tree)
  __threadfence_block();
  __syncthreads();
  if (uni(L.do_save)) op_save_intra(p_, scratch, L.cu);
}

template <typename T>
__device__ __noinline__ void run_tree(const VxParams &p_, const VxFrameDev &fd_, uint8_t *scratch)
{
  scratch = uni_p(scratch);
  const VxParams &p = L.par; (void) p_; const VxFrameDev &fd = L.fdv; (void) fd_;
  const int tid = VTX;
  // NOTE on the shape of this loop.  Round 1 saw a hang with two `if (tid == 0)` sections per iteration (head and tail): hipcc threaded the "tid != 0" edges
  // together and structurised the result into an inner loop in which lanes 1..63 of the controller's wave reached the next s_barrier while lane 0 was still
  // parked outside it, so the barrier released before the controller had run.  The cause is that `tid == 0` is a divergent condition *of the wave that holds
  // the controller*: a barrier behind it is reached by that wave under a partial exec mask whenever the structuriser decides to split the paths.  The loop is
  // therefore written so that no barrier can sit behind a lane-divergent branch: the controller section is entered on a wave-uniform condition (the wave index
  // in an SGPR) and lane 0 is picked *inside* it, in a region that contains no barrier and rejoins before the branch ends; after_intra_op does the same.
  // tests/test_host_cpu.py::test_barrier_shape_of_the_operation_loop checks the built code object: run_tree reaches its barriers with exec restored.
  const int ctl_wave = uni(VTX >> 6) == 0, ctl_lane = (VTX & 63) == 0;
  (void) tid;
  int prev_op = 13; long long t_prev = STAMP();
  for (;;) {
    if (ctl_wave) { __builtin_amdgcn_s_setprio(VXD_SERIAL_PRIO); if (ctl_lane) {
      const long long t0 = STAMP();
      if (VVCX_STAMP) {
        PROF(imin(prev_op, 13)) += (unsigned long long) (t0 - t_prev);     // previous operation (prof[0] absorbs the first; [13]: the ISP operations)
#ifndef VVCX_STAMP_ROUNDS
        if (prev_op >= OP_LUMA_PREP && prev_op <= OP_CHROMA_RD) PROF(38 + imin(9, imax(0, ilog2i(L.nw * L.nh) - 4))) += (unsigned long long) (t0 - t_prev);   // by node size
#endif
      }
      L.pre_copy_d = -1;
      control_step(p, fd, scratch);
#ifdef VX_TRACE
      fprintf(stderr, "cnt0 %llu ch %d op %d a %d node %d %d %dx%d d %d phase %d isp slot %d mode %d split %d\n", L.cnt[0], L.tree_ch, L.op, L.op_a, L.nx, L.ny, L.nw, L.nh, L.d, L.d >= 0 ? L.fr[L.d].phase : -1, L.S.ispSlot, L.isp_mode, L.isp_split);
#endif
      t_prev = STAMP();
      if (VVCX_STAMP) PROF(0) += (unsigned long long) (t_prev - t0);
    } __builtin_amdgcn_s_setprio(0); }
    __syncthreads();
    const int op = uni(L.op);
    { const int pd = uni(L.pre_copy_d); if (pd >= 0) ctx_copy_all(ctx_ptr(scratch, CTX_START, pd, 0), &L.ctxs[CI_CUR]); }      // reads only; every operation leaves L.ctxs[CI_CUR] alone until its own barrier
    if (op == OP_DONE) break;
    prev_op = op;
    switch (op) {
      case OP_LUMA_PREP: op_luma_prep<T>(p, fd); if (uni(L.op_c)) op_stage_a(p, scratch); break;
      case OP_STAGE_A: op_stage_a(p, scratch); break;
      case OP_STAGE_B: op_stage_b(p, scratch); if (!uni((int) L.isp_wait)) after_intra_op(p, fd, scratch); break;
      case OP_ISP: op_isp(scratch); break;
      case OP_ISP_PARK: op_isp_park(scratch); break;
      case OP_ISP_END:                                    // every place used: the end contexts of the node's winner (regular or ISP) come back, then the node's intra decision
        // (an operation kind of its own: the decision below rewrites the operation's parameters while other waves may still be dispatching)
        ctx_copy_all(&L.ctxs[CI_W(0)], ctx_ptr(scratch, CTX_BEST, MAXD + (uni((int) L.isp_win) ? 1 : 0), 0));
        __threadfence_block(); __syncthreads();
        after_intra_op(p, fd, scratch);
        break;
      case OP_CHROMA_RD: op_chroma_rd<T>(p, fd, scratch); after_intra_op(p, fd, scratch); break;
      case OP_SAVE_INTRA: op_save_intra(p, scratch, L.cu); break;
      case OP_SAVE_PIC: op_save_pic<T>(p, fd, scratch, 0); break;
      case OP_RESTORE_PIC: ctx_copy_all(ctx_ptr(scratch, L.op_a, L.op_c, 0), ctx_ptr(scratch, L.op_b, L.op_c, 0)); op_save_pic<T>(p, fd, scratch, 1); break;
      case OP_CLEAR_UNITS: op_clear_units(p, fd); break;
      case OP_CTX_COPY: ctx_copy_all(ctx_ptr(scratch, L.op_a, L.op_c, 0), ctx_ptr(scratch, L.op_b, L.op_c, 0)); __threadfence_block(); break;
      case OP_REUSE: op_reuse<T>(p, fd, scratch); after_intra_op(p, fd, scratch); break;
      case OP_FAST: op_fast<T>(p, fd); break;
    }
    __syncthreads();
  }
}

// ------------------------------------------------------------------------------------------------ kernel
// WPP hand-over between the workgroups that run the CTU rows of a tile.  A row publishes the number of its finished CTUs with release semantics at device scope after a
// device-scope fence of every thread that wrote picture data; a workgroup that wants to run CTU k of the row below first reads that count with an acquire load (thread 0),
// and after the barrier every thread fences (acquire) before it reads the neighbour's samples.  Nobody ever waits in place: a row whose next CTU is not ready is put back
// (contexts, coder state and position are stream state in HBM) and the workgroup takes another row that is (run_streams_wpp).
__device__ inline int wpp_count(const int32_t *cnt)
{
#ifdef VX_EMU
  return *cnt;
#else
  return __hip_atomic_load(cnt, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT);
#endif
}
// the same value without the acquire (an acquire load at device scope invalidates the CU's vector L1, which the three other workgroups of the CU live on): for looking
// around; whoever acts on what it saw takes the lock and fences once
__device__ inline int wpp_peek(const int32_t *cnt)
{
#ifdef VX_EMU
  return *cnt;
#else
  return __hip_atomic_load(cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#endif
}
__device__ inline void wpp_publish(int32_t *cnt, int v)
{
#ifdef VX_EMU
  *cnt = v;
#else
  __hip_atomic_store(cnt, v, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
#endif
}
// all threads: is CTU `need - 1` of the row above finished?  (thread 0 asks, everybody gets the answer and the acquire fence)
__device__ __noinline__ int wpp_ready(const int32_t *cnt, int need)
{
  if (VTX == 0) L.wpp_ok = wpp_peek(cnt) >= need;
  __syncthreads();
  const int ok = uni(L.wpp_ok);
#ifndef VX_EMU
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
#endif
  __syncthreads();
  return ok;
}
// Runs the tasks [pos0, ...) of a stream for as long as they are ready (always, without WPP) and returns the position it stopped at; the stream's state (contexts, coder,
// position) is left in HBM, so that any workgroup can continue it.
template <typename T>
__device__ int run_stream(const VxParams &p, int stream_idx, int pos0)
{
  VxStreamDesc sd = p.streams[stream_idx];
  sd.done_before += pos0; sd.first_task += pos0; sd.n_tasks -= pos0;      // what is left of it in this launch
  if (VTX == 0) { L.par = p; L.fdv = p.frames[sd.frame]; }
  const VxFrameDev &fd = p.frames[sd.frame];
  uint8_t *scratch = p.scratch + (size_t) blockIdx.x * p.scratch_per_stream;
  const int sidx = sd.frame * p.nsub + sd.sub;
  Ctx *carry = (Ctx *) (p.stream_ctx + (size_t) sidx * 2 * NCTX);
  const int tid = VTX;
  if (tid == 0) {
    L.cur_tile = sd.tile; L.frame = sd.frame; L.lmcs_cadj = 0; L.lmcs_tab = 0; for (int i = 0; i < 4; i++) L.cnt[i] = 0; for (int i = 0; i < (VVCX_STAMP ? 48 : 1); i++) PROF(i) = 0;
    if (p.payload) writer_begin(p, sidx, sd.done_before);
    if (p.tools & TOOL_CU_REUSE) L.cache_gen = (int) *(const uint32_t *) (scratch + VXD_OFF_META);
  }
  load_tables();
  // WPP: the row's first CTU starts from the contexts the row above left behind its own first CTU (EL/EncSlice.cpp:1648-1661); later launches of the row carry on from `carry`
  const int wpp_above = sd.above >= 0 ? sd.frame * p.nsub + sd.above : -1;
  if (wpp_above >= 0 && sd.done_before == 0) ctx_copy_all(&L.ctxs[CI_CUR], (const Ctx *) (p.wpp_sync + (size_t) wpp_above * 2 * NCTX));      // (the scheduler saw the row above past its first CTU)
  else ctx_copy_all(&L.ctxs[CI_CUR], carry);
  __syncthreads();
  int t = 0;
  for (; t < sd.n_tasks; t++) {
    const int addr = p.task_ctu[sd.first_task + t];
    const int ctu_x = (addr % p.ctus_w) << 7, ctu_y = (addr / p.ctus_w) << 7;
    // WPP: CTU k of a row reads the reconstruction and the CU data of CTU k of the row above (the one further right is hidden from it, build_refs)
    if (wpp_above >= 0 && !wpp_ready(p.wpp_progress + wpp_above, sd.done_before + t + 1)) break;
    if (p.wpp_rr && t > 0) break;                          // test mode: one CTU per visit, so that the rows of a picture really interleave
    // contexts at CTU start → snapshot slot (MAXD-1) "start"
    ctx_copy_all(ctx_ptr(scratch, CTX_START, MAXD + NW, 0), &L.ctxs[CI_CUR]);
    if (p.tools & TOOL_CU_REUSE) {                       // entries of an earlier CTU can never match (987-1024: poc / absolute area): a new generation drops them
      const int g = uni(L.cache_gen) + 1;
      if ((g & 0xffff) == 0) {                             // the 16-bit tag wraps: clear the entries once
        uint64_t *ce = (uint64_t *) (scratch + VXD_OFF_CACHE);   // sizeof(VxCacheEnt) == 16
        for (int i = tid; i < 2 * VXD_CACHE_ENTRIES; i += NT) ce[i] = 0;
        __threadfence_block();
      }
      __syncthreads();
      if (tid == 0) L.cache_gen = (g & 0xffff) == 0 ? g + 1 : g;
    }
    __syncthreads();
    VxCtuRes res; res.dist = 0; res.bits = 0; res.cost = 0; res.n_cu = 0; res.pad = 0;
    for (int ch = 0; ch < (p.chroma ? 2 : 1); ch++) {
      if (tid == 0) {
        L.tree_ch = ch; L.d = 0; L.ctu_x = ctu_x; L.ctu_y = ctu_y;
        Frame &f = L.fr[0];
        f.x = (int16_t) ctu_x; f.y = (int16_t) ctu_y; f.w = 128; f.h = 128; f.depth = 0; f.qt = 0; f.bt = 0; f.mt = 0; f.impl_bt = 0;
        f.last_split = 0; f.part_idx = 0; f.impl_checked = 0; f.ss = 0; f.max_cost = MAX_DOUBLE; f.phase = PH_ENTER;
      }
      if (ch == 1) ctx_copy_all(&L.ctxs[CI_CUR], ctx_ptr(scratch, CTX_START, MAXD + NW, 0));     // EL/EncCu.cpp:521
      __syncthreads();
      run_tree<T>(p, fd, scratch);
      if (tid == 0) { const Sum &b = L.fr[0].best; res.dist += b.dist; res.bits += b.bits; res.cost += b.cost; res.n_cu += b.n_cu; }
      __syncthreads();
    }
    // contexts back to the CTU start, then the estimator pass advances them (EL/EncCu.cpp:543, EL/EncSlice.cpp:1775-1776)
    ctx_copy_all(&L.ctxs[CI_CUR], ctx_ptr(scratch, CTX_START, MAXD + NW, 0));
    __syncthreads();
    if (tid == 0) {
      const long long t0 = STAMP();
      advance_ctx_ctu<T>(p, fd, sd.tile, ctu_x, ctu_y);
      if (p.payload) writer_end_of_ctu(sd.done_before + t + 1 == sd.tile_ctus, sd.sub == p.nsub - 1);
      if (VVCX_STAMP) PROF(12) += (unsigned long long) (STAMP() - t0);
      p.results[sd.first_task + t] = res;
    }
    __syncthreads();
    if (p.tools & TOOL_WPP) {                              // publish: first the contexts behind the row's first CTU (1801-1805), then the count the row below waits on
      if (sd.done_before + t == 0) ctx_copy_all((Ctx *) (p.wpp_sync + (size_t) sidx * 2 * NCTX), &L.ctxs[CI_CUR]);
      __threadfence();
      __syncthreads();
      if (tid == 0) { wpp_publish(p.wpp_progress + sidx, sd.done_before + t + 1); atomicAdd(p.wpp_sched + 2, 1); }      // + the scheduler's sign of life: one tick per finished CTU
    }
  }
  ctx_copy_all(carry, &L.ctxs[CI_CUR]);
  if (tid == 0 && (p.tools & TOOL_CU_REUSE)) *(uint32_t *) (scratch + VXD_OFF_META) = (uint32_t) L.cache_gen;
  if (tid == 0 && p.payload) writer_suspend(p, sidx);
#ifdef VX_DBG_STREAM
  if (tid == 0) fprintf(stderr, "stream %d sub %d done_before %d tasks %d slot %d cnt %llu %llu %llu %llu gen %d\n", stream_idx, sd.sub, sd.done_before, sd.n_tasks, (int) blockIdx.x, L.cnt[0], L.cnt[1], L.cnt[2], L.cnt[3], L.cache_gen);
#endif
  if (tid == 0) { for (int i = 0; i < 4; i++) atomicAdd(&p.counters[i], L.cnt[i]); if (VVCX_STAMP) for (int i = 0; i < 48; i++) atomicAdd(&p.counters[4 + i], PROF(i)); }
  return pos0 + t;
}

// ------------------------------------------------------------------------------------------------ leaf operators
// The device functions of the path behind the individually testable operators of include/vvcx.h (≙ the reference's
// function-pointer seams).  One workgroup per item; same code the CTU kernel runs.
extern "C" __global__ void __launch_bounds__(NT, VXD_WPE) vvcx_leaf_dist_kernel(const int16_t *a, const int16_t *b, int w, int h, int16_t *scr, unsigned long long *out)
{
  const int wave = uni(VTX >> 6), lane = VTX & 63, P = w * h;
  if (wave != 0) return;
  const int16_t *pa = a + (size_t) blockIdx.x * P, *pb = b + (size_t) blockIdx.x * P;
  unsigned long long sad, satd;
  wave_sad_satd<false>(pa, pb, scr + (size_t) blockIdx.x * P, w, h, lane, sad, satd);
  unsigned long long sse = 0;
  for (int i = lane; i < P; i += 64) { const int d = pa[i] - pb[i]; sse += (unsigned long long) (d * d); }
  sse = wave_sum_u64(sse);
  if (lane == 0) { out[blockIdx.x * 3 + 0] = sad; out[blockIdx.x * 3 + 1] = satd; out[blockIdx.x * 3 + 2] = sse; }
}
// T → Q → Q⁻¹ → T⁻¹ → reco → SSE of one block per workgroup (the core of xIntraCodingTUBlock): rec holds the prediction on entry
extern "C" __global__ void __launch_bounds__(NT, VXD_WPE) vvcx_leaf_trq_kernel(const int16_t *org, int16_t *rec, int16_t *lev, int32_t *tmp, int w, int h, int bd, int qp,
                                                                               unsigned long long *out)
{
  if (VTX == 0) L.par.tools = 0;                           // the plain quantiser (LDS is not cleared between launches)
  load_tables();
  __syncthreads();
  const int wave = uni(VTX >> 6), lane = VTX & 63, P = w * h;
  if (wave != 0) return;
  unsigned long long sse; int cbf;
  wave_code_block<false>(org + (size_t) blockIdx.x * P, 0, 0, rec + (size_t) blockIdx.x * P, lev + (size_t) blockIdx.x * P, tmp + (size_t) blockIdx.x * 2048, w, h, bd, qp, lane, sse, cbf);
  if (lane == 0) { out[blockIdx.x * 2] = sse; out[blockIdx.x * 2 + 1] = (unsigned long long) cbf; }
}
// the same with the dependent quantiser (wave_depquant + wave_dequant_dq) from given context models: p carries tools, dq_consts and a scratch area
// (VXD_OFF_CACHE bytes per block); blocks of at most BUF samples run the LDS-resident form the search uses for them, bigger ones the HBM form
extern "C" __global__ void __launch_bounds__(NT, VXD_WPE) vvcx_leaf_dq_kernel(VxParams p, const uint16_t *ctx, const int16_t *org, int16_t *rec, int16_t *lev, int32_t *tmp,
                                                                              int w, int h, int qp, int comp, int mts, int cbf_cb, unsigned long long *out, int lf, int lfdir)
{
  if (VTX == 0) { L.par = p; L.lmcs_tab = 0; }
  load_tables();
  for (int i = VTX; i < NCTX; i += NT) { L.ctxs[CI_CUR].s0[i] = ctx[i]; L.ctxs[CI_CUR].s1[i] = ctx[NCTX + i]; }
  __syncthreads();
  const int wave = uni(VTX >> 6), lane = VTX & 63, P = w * h, bd = uni(p.bit_depth);
  if (wave != 0) return;
  const size_t b = (size_t) blockIdx.x * P;
  unsigned long long sse; int cbf;
  if (P <= BUF) {
    for (int i = lane; i < P; i += 64) { L.org[i] = org[b + i]; L.wm[0].slot[i] = rec[b + i]; }
    wave_sync();
    if (mts > 1) wave_code_block_mts<true>(nullptr, nullptr, nullptr, nullptr, w, h, bd, qp, mts, lane, sse, cbf);
    else wave_code_block<true>(nullptr, 0, 0, nullptr, nullptr, nullptr, w, h, bd, qp, lane, sse, cbf, -1, nullptr, comp, CI_CUR, cbf_cb, lf, lf ? lfnst_mode(lfdir, w, h) : 0);
    wave_sync();
    for (int i = lane; i < P; i += 64) { rec[b + i] = L.wm[0].slot[i]; lev[b + i] = L.wm[0].slot[BUF + i]; }
  } else if (mts > 1) wave_code_block_mts<false>(org + b, rec + b, lev + b, tmp + (size_t) blockIdx.x * 2048, w, h, bd, qp, mts, lane, sse, cbf);
  else wave_code_block<false>(org + b, 0, 0, rec + b, lev + b, tmp + (size_t) blockIdx.x * 2048, w, h, bd, qp, lane, sse, cbf, -1, nullptr, comp, CI_CUR, cbf_cb, lf, lf ? lfnst_mode(lfdir, w, h) : 0);
  if (lane == 0) { out[blockIdx.x * 2] = sse; out[blockIdx.x * 2 + 1] = (unsigned long long) cbf; }
}
// transform skip of one residual block per workgroup (tests/golden/ts.npz): the pruning decision against the DCT-II sum, xTransformSkip, RDOQ-TS from the given context
// models, Quant::dequant + xITransformSkip, and the bits residual_codingTS spends on the levels.  resi_out must be zero on entry; out = {absSum, keep} per block
extern "C" __global__ void __launch_bounds__(NT, VXD_WPE) vvcx_leaf_ts_kernel(VxParams p, const uint16_t *ctx, const int16_t *resi, int16_t *lev, int16_t *resi_out, int32_t *tmp,
                                                                              int w, int h, int qp, int *out, unsigned long long *bits)
{
  if (VTX == 0) L.par = p;
  load_tables();
  for (int i = VTX; i < NCTX; i += NT) { L.ctxs[CI_CUR].s0[i] = ctx[i]; L.ctxs[CI_CUR].s1[i] = ctx[NCTX + i]; }
  __syncthreads();
  ts_build_tables();
  __syncthreads();
  const int wave = uni(VTX >> 6), lane = VTX & 63, P = w * h, bd = uni(p.bit_depth);
  if (wave != 0) return;
  const size_t b = (size_t) blockIdx.x * P;
  unsigned long long sse; int cbf, sum0 = 0;
  wave_code_block<false, true>(resi + b, 0, 0, resi_out + b, lev + b, tmp + (size_t) blockIdx.x * 2048, w, h, bd, qp, lane, sse, cbf, -2, &sum0);
  wave_sync();
  const int sa = wave_ts_fwd(resi + b, resi_out + b, lev + b, w, h, bd, lane);
  wave_sync();
  int a = 0;
  if (lane == 0) a = ts_rdoq_lane(lev + b, w, h, bd, qp);
  a = __builtin_amdgcn_readlane(a, 0);
  wave_sync();
  wave_ts_recon(resi + b, resi_out + b, lev + b, w, h, bd, qp, a > 0, lane, sse, 1);
  if (lane == 0) {
    Cab cb; cb.ci = CI_CUR; cb.bits = 0;
    if (a > 0) rc_ts_serial(cb, lev + b, w, h);
    out[blockIdx.x * 2] = a; out[blockIdx.x * 2 + 1] = sa <= sum0; bits[blockIdx.x] = cb.bits;
  }
}
// one sub-partition block of an ISP CU per workgroup (tests/golden/isp.npz): implicit transform, dependent quantisation with the ISP cbf context (cbf_ctx < 0: inferred),
// dequantisation, inverse, reconstruction; rec holds the prediction on entry
extern "C" __global__ void __launch_bounds__(NT, VXD_WPE) vvcx_leaf_isp_kernel(VxParams p, const uint16_t *ctx, const int16_t *org, int16_t *rec, int16_t *lev, int32_t *tmp,
                                                                               int w, int h, int qp, int cbf_ctx, unsigned long long *out)
{
  if (VTX == 0) { L.par = p; L.lmcs_tab = 0; }
  load_tables();
  for (int i = VTX; i < NCTX; i += NT) { L.ctxs[CI_CUR].s0[i] = ctx[i]; L.ctxs[CI_CUR].s1[i] = ctx[NCTX + i]; }
  __syncthreads();
  const int wave = uni(VTX >> 6), lane = VTX & 63, P = w * h;
  if (wave != 0) return;
  const size_t b = (size_t) blockIdx.x * P;
  unsigned long long sse; int cbf;
  wave_code_block_isp<false>(org + b, rec + b, lev + b, w, tmp + (size_t) blockIdx.x * 2048, lev + b, p.scratch + (size_t) blockIdx.x * p.scratch_per_stream, w, h, p.bit_depth, qp, lane, sse, cbf, -1, CI_CUR, cbf_ctx);
  if (lane == 0) { out[blockIdx.x * 2] = sse; out[blockIdx.x * 2 + 1] = (unsigned long long) cbf; }
}
template <typename T>
__device__ void leaf_pred(const VxParams &p, const VxLeafPred *cases, int16_t *out, const int *out_off)
{
  const VxFrameDev &fd = p.frames[0];
  const VxLeafPred c = cases[blockIdx.x];
  if (VTX == 0) { L.par = p; L.fdv = p.frames[0]; }
  load_tables();
  if (VTX == 0) { L.cur_tile = 0; L.nx = c.x; L.ny = c.y; L.nw = c.w; L.nh = c.h; }
  __syncthreads();
  const int luma = c.comp == 0;
  build_refs<T>(p, fd, c.comp, c.x, c.y, c.w, c.h, 0, luma ? 3 : 1);
  __syncthreads();
  if (VTX < 4) {
    const int s = VTX;
    if (luma ? s != 1 : s == c.comp - 1) L.dc_val[s] = dc_value(L.refs[s][0], L.refs[s][1], c.w, c.h, luma ? (s == 0 ? 0 : s == 2 ? 1 : 3) : 0);
  }
  __syncthreads();
  Ipa ip; init_pred_params(c.w, c.h, luma, c.mode, c.mrl, ip);
  const int set = luma ? luma_set(c.mrl, ip.ref_filter) : c.comp - 1;
  const int dcv = L.dc_val[luma ? luma_set(c.mrl, 0) : c.comp - 1];
  int16_t *o = out + out_off[blockIdx.x];
  for (int i = VTX; i < c.w * c.h; i += NT) { const int py = i >> ilog2i(c.w), px = i & (c.w - 1); o[i] = (int16_t) pred_sample(L.refs[set][0], L.refs[set][1], c.w, c.h, px, py, ip, c.mode, luma, p.bit_depth, dcv); }
}
extern "C" __global__ void __launch_bounds__(NT, VXD_WPE) vvcx_leaf_pred_kernel_u8(VxParams p, const VxLeafPred *cases, int16_t *out, const int *out_off) { leaf_pred<uint8_t>(p, cases, out, out_off); }
extern "C" __global__ void __launch_bounds__(NT, VXD_WPE) vvcx_leaf_pred_kernel_u16(VxParams p, const VxLeafPred *cases, int16_t *out, const int *out_off) { leaf_pred<uint16_t>(p, cases, out, out_off); }
// estimator model: bins on one context (BinProbModel_Std::estFracBitsUpdate); io = {s0, s1} in/out, bits out
// (rate: the model's adaptation rate byte; the model itself sits in slot 0)
extern "C" __global__ void __launch_bounds__(NT, VXD_WPE) vvcx_leaf_cabac_kernel(uint16_t *io, int rate, const uint8_t *bins, int nbins, unsigned long long *bits)
{
  load_tables();
  __syncthreads();
  if (VTX == 0) {
    const int ctx = 0; L.t.ctx_rate[0] = (uint8_t) rate;
    L.ctxs[CI_CUR].s0[ctx] = io[0]; L.ctxs[CI_CUR].s1[ctx] = io[1];
    Cab cb; cb.ci = CI_CUR; cb.bits = 0;
    for (int i = 0; i < nbins; i++) enc_bin(cb, bins[i], ctx);
    io[0] = L.ctxs[CI_CUR].s0[ctx]; io[1] = L.ctxs[CI_CUR].s1[ctx]; bits[0] = cb.bits;
  }
}
extern "C" __global__ void __launch_bounds__(NT, VXD_WPE) vvcx_leaf_rdcost_kernel(VxParams p, const unsigned long long *bits, const unsigned long long *dist, int n, double *cost)
{
  const int i = blockIdx.x * NT + VTX;
  if (i < n) cost[i] = rd_cost(p, bits[i], dist[i]);
}
extern "C" __global__ void __launch_bounds__(NT, VXD_WPE) vvcx_leaf_scan_kernel(int w, int h, uint16_t *idx)
{
  load_tables();
  __syncthreads();
  const ScanGeo g = scan_geo(w, h);
  for (int sp = VTX; sp < g.nscan; sp += NT) idx[sp] = (uint16_t) scan_blk(g, sp);
}

// forest inference on its own: one thread per feature row, trees in order (the sum order of OP_FAST and of sklearn's predict_proba)
extern "C" __global__ void __launch_bounds__(NT, VXD_WPE) vvcx_leaf_forest_kernel(VxParams p, const int32_t *rows, int n, int32_t *out)
{
  const int i = blockIdx.x * NT + VTX;
  if (i >= n) return;
  double acc[8] = { 0, 0, 0, 0, 0, 0, 0, 0 };
  for (int t = 0; t < p.f_ntrees; t++) {
    int nd = p.f_root[t];
    for (;;) {
      const VxForestNode q = p.f_node[nd];
      if (q.left < 0) break;
      nd = ((double) (float) rows[i * 26 + q.feature] <= q.thr) ? q.left : q.right;
    }
    for (int c = 0; c < p.f_nclasses; c++) acc[c] += p.f_value[(size_t) nd * p.f_nclasses + c];
  }
  int best = 0;
  for (int c = 1; c < p.f_nclasses; c++) if (acc[c] > acc[best]) best = c;
  out[i] = p.f_classes[best];
}

// One workgroup per resident stream slot: it takes stream descriptors from the launch's queue until the queue is empty (every workgroup reaches the exit),
// so the scratch of a launch is sized by the slots, not by the streams.
template <typename T>
__device__ void run_streams(const VxParams &p)
{
  // LDS keeps whatever the previous workgroup left: the CPU emulation build fills the object with a poison pattern here (VX_POISON_LDS is empty in the device build), so that a
  // field read before its first write shows up in the CPU suite too and not only as a GPU-only difference in the work counters (round 1's "miscompile", round 2's L.mip_n)
  VX_POISON_LDS(L);
  for (;;) {
    __syncthreads();
    if (VTX == 0) L.cur_stream = (int) atomicAdd(&p.counters[52], 1ull);
    __syncthreads();
    const int s = uni(L.cur_stream);
    if (s >= p.n_streams) break;
    run_stream<T>(p, s, 0);
  }
}
// WPP: the streams are CTU rows that depend on each other, so a workgroup does not own a row for its lifetime.  Scheduler state in HBM (p.wpp_sched, set up per launch by the
// host): [0] rows finished, [1] abort flag, [2] CTUs finished, then per stream an owner flag and the number of its tasks that are left.  A workgroup looks for a row that nobody runs, that has
// tasks left and whose next CTU is ready (first fit in queue order = longest tile first, top row first), runs it while it stays ready, puts it back, and looks again; it leaves
// when every row is finished.  No workgroup ever waits for another one while holding something the other needs, and every wave reaches the exit: either all rows finish, or
// a workgroup that found nothing to do while nobody finished a CTU for two minutes (run_stream ticks once per CTU) raises the abort flag, which everybody sees at the next look (the host reports it).
template <typename T>
__device__ void run_streams_wpp(const VxParams &p)
{
  VX_POISON_LDS(L);
  const int n = p.n_streams;
  int32_t *done = p.wpp_sched, *abort_flag = p.wpp_sched + 1, *ticks = p.wpp_sched + 2, *owner = p.wpp_sched + 4, *left = p.wpp_sched + 4 + n;
  int start = 0;
  for (;;) {
    __syncthreads();
    if (VTX == 0) {
      int pick = -1, pos = 0;
#ifndef VX_EMU
      long long t_idle = (long long) wall_clock64();
      int seen = wpp_peek(ticks);
#endif
      for (;;) {
        if (wpp_peek(abort_flag) || wpp_peek(done) >= n) break;
        for (int i = 0; i < n && pick < 0; i++) {
          const int s = p.wpp_rr ? (start + i) % n : i;
          if (wpp_peek(owner + s)) continue;
          int l = wpp_peek(left + s);
          if (l <= 0) continue;
          const VxStreamDesc sd = p.streams[s];
          const int32_t *above = sd.above >= 0 ? p.wpp_progress + (sd.frame * p.nsub + sd.above) : nullptr;
          if (above && wpp_peek(above) < sd.done_before + (sd.n_tasks - l) + 1) continue;
#ifdef VX_EMU
          if (*(owner + s)) continue; *(owner + s) = 1;
#else
          if (atomicCAS((int *) owner + s, 0, 1) != 0) continue;
#endif
          l = wpp_count(left + s);                        // under the lock, with the acquire: somebody may have run the row in between
          if (l <= 0 || (above && wpp_count(above) < sd.done_before + (sd.n_tasks - l) + 1)) { wpp_publish(owner + s, 0); continue; }
          pick = s; pos = sd.n_tasks - l;
        }
        if (pick >= 0) break;
#ifdef VX_EMU
        wpp_publish(abort_flag, 1);                       // the emulator runs one workgroup at a time: rows that are left but not ready will never become ready
#else
        // nothing to do right now.  Give up only when NOBODY has finished a CTU for two minutes (a row at the bottom of a picture legitimately waits much longer than a CTU takes)
        const int now = wpp_peek(ticks);
        if (now != seen) { seen = now; t_idle = (long long) wall_clock64(); }
        else if ((long long) wall_clock64() - t_idle > 120ll * 100000000ll) wpp_publish(abort_flag, 1);      // 100 MHz counter
        for (int k = 0; k < 32; k++) __builtin_amdgcn_s_sleep(127);      // ~0.1 ms: a CTU takes seconds
#endif
      }
      L.cur_stream = pick; L.wpp_pos = pos;
    }
    __syncthreads();
    const int s = uni(L.cur_stream);
    if (s < 0) break;
#ifndef VX_EMU
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");    // the row's state was written by the workgroup that ran it before
#endif
    const int pos0 = uni(L.wpp_pos);
    const int pos1 = run_stream<T>(p, s, pos0);
    __threadfence();
    __syncthreads();
    if (VTX == 0) {
      const int ntasks = p.streams[s].n_tasks;
      wpp_publish(left + s, ntasks - pos1);
      if (pos1 >= ntasks) atomicAdd((int *) done, 1);
      wpp_publish(owner + s, 0);
    }
    start = s + 1;
  }
}
extern "C" __global__ void __launch_bounds__(NT, VXD_WPE) vvcx_compress_kernel_u8(VxParams p) { run_streams<uint8_t>(p); }
extern "C" __global__ void __launch_bounds__(NT, VXD_WPE) vvcx_compress_kernel_u16(VxParams p) { run_streams<uint16_t>(p); }
// the same search under the WPP scheduler (VVCX_TOOL_WPP): kernels of their own, so that the default kernels stay what they were
extern "C" __global__ void __launch_bounds__(NT, VXD_WPE) vvcx_compress_wpp_kernel_u8(VxParams p) { run_streams_wpp<uint8_t>(p); }
extern "C" __global__ void __launch_bounds__(NT, VXD_WPE) vvcx_compress_wpp_kernel_u16(VxParams p) { run_streams_wpp<uint16_t>(p); }

// slice joint_cb_cr_sign_flag of a bound picture (setJointCbCrModes, EL/EncSlice.cpp:1503-1538): the sign of the correlation of the high-pass filtered Cb and
// Cr planes over the interior samples; one workgroup per picture, launched when the pictures are bound
template <typename T>
__device__ void jccr_sign(VxFrameDev *frames, int wc, int hc)
{
  __shared__ long long part[NT];
  VxFrameDev &fd = frames[blockIdx.x];
  const T *cb = (const T *) fd.org[1], *cr = (const T *) fd.org[2];
  const int sb = fd.stride[1], sr = fd.stride[2];
  long long sum = 0;
  const int iw = wc - 2, n = iw * (hc - 2);
  for (int i = threadIdx.x; i < n; i += NT) {
    const int y = 1 + i / iw, x = 1 + i % iw;
    const T *p = cb + y * sb + x, *q = cr + y * sr + x;
    const int a = 12 * (int) p[0] - 2 * ((int) p[-1] + (int) p[1] + (int) p[-sb] + (int) p[sb]) - ((int) p[-1 - sb] + (int) p[1 - sb] + (int) p[-1 + sb] + (int) p[1 + sb]);
    const int b = 12 * (int) q[0] - 2 * ((int) q[-1] + (int) q[1] + (int) q[-sr] + (int) q[sr]) - ((int) q[-1 - sr] + (int) q[1 - sr] + (int) q[-1 + sr] + (int) q[1 + sr]);
    sum += (long long) a * b;
  }
  part[threadIdx.x] = sum;
  __syncthreads();
  if (threadIdx.x == 0) { long long t = 0; for (int i = 0; i < NT; i++) t += part[i]; fd.jccr_sign = t < 0; }
}
extern "C" __global__ void __launch_bounds__(NT) vvcx_jccr_sign_kernel_u8(VxFrameDev *frames, int wc, int hc) { jccr_sign<uint8_t>(frames, wc, hc); }
extern "C" __global__ void __launch_bounds__(NT) vvcx_jccr_sign_kernel_u16(VxFrameDev *frames, int wc, int hc) { jccr_sign<uint16_t>(frames, wc, hc); }

// LMCS: one plane through a LUT (AreaBuf<Pel>::rspSignal, CL/Buffer.cpp:485-499): the forward map of a bound picture's original luma, the inverse map of its reconstruction
template <typename T>
__device__ void lmcs_map(const T *src, T *dst, int w, int h, int stride, const int16_t *lut)
{
  const int i = blockIdx.x * NT + threadIdx.x;
  if (i >= w * h) return;
  const int y = i / w, x = i - y * w;
  dst[y * stride + x] = (T) lut[src[y * stride + x]];
}
extern "C" __global__ void __launch_bounds__(NT) vvcx_lmcs_map_kernel_u8(const uint8_t *src, uint8_t *dst, int w, int h, int stride, const int16_t *lut) { lmcs_map<uint8_t>(src, dst, w, h, stride, lut); }
extern "C" __global__ void __launch_bounds__(NT) vvcx_lmcs_map_kernel_u16(const uint16_t *src, uint16_t *dst, int w, int h, int stride, const int16_t *lut) { lmcs_map<uint16_t>(src, dst, w, h, stride, lut); }

// A cheap activity measure per CTU of the bound pictures (sum of absolute horizontal and vertical luma differences): the launch's stream queue is ordered by it,
// longest first, so that the streams that take longest do not start last (the tail of a launch).  One workgroup per (CTU, picture).
template <typename T>
__device__ void ctu_activity(const VxFrameDev *frames, int pic_w, int pic_h, int ctus_w, unsigned *out)
{
  __shared__ unsigned part[NT];
  const VxFrameDev &fd = frames[blockIdx.y];
  const T *y = (const T *) fd.org[0]; const int st = fd.stride[0];
  const int x0 = (blockIdx.x % ctus_w) << 7, y0 = (blockIdx.x / ctus_w) << 7;
  const int w = imin(128, pic_w - x0), h = imin(128, pic_h - y0);
  unsigned sum = 0;
  for (int i = threadIdx.x; i < (w - 1) * (h - 1); i += NT) {
    const int r = i / (w - 1), c = i - r * (w - 1);
    const T *p = y + (y0 + r) * st + x0 + c;
    const int a = (int) p[0];
    sum += (unsigned) (iabs(a - (int) p[1]) + iabs(a - (int) p[st]));
  }
  part[threadIdx.x] = sum;
  __syncthreads();
  if (threadIdx.x == 0) { unsigned t = 0; for (int i = 0; i < NT; i++) t += part[i]; out[blockIdx.y * gridDim.x + blockIdx.x] = t; }
}
extern "C" __global__ void __launch_bounds__(NT) vvcx_ctu_activity_kernel_u8(const VxFrameDev *frames, int pic_w, int pic_h, int ctus_w, unsigned *out) { ctu_activity<uint8_t>(frames, pic_w, pic_h, ctus_w, out); }
extern "C" __global__ void __launch_bounds__(NT) vvcx_ctu_activity_kernel_u16(const VxFrameDev *frames, int pic_w, int pic_h, int ctus_w, unsigned *out) { ctu_activity<uint16_t>(frames, pic_w, pic_h, ctus_w, out); }

// vvcx_lmcs.hip — the LMCS picture analysis of an intra picture: which luma-mapping model, if any, the slice carries.
//
// What is computed is the reference's EncGOP::xPicInitLMCS → EncReshape::preAnalyzerLMCS for the all-intra cfg (LMCSSignalType 0 = SDR, LMCSAdpOption 0, LMCSUpdateCtrl 0 / 1):
// picture statistics (EL/EncReshape.cpp:166-409), the decision cascade of the SDR parameter derivation (977-1229) with its code-word budgeting helpers (931-975) and the model
// half of constructReshaperLMCS with the pivot alignment (1835-1893, 2194-2256).  How it is organised here:
//
//   * the statistics are a DEVICE pass over the picture in HBM (three small kernels): separable window sums of the luma samples and their squares, then per sample the
//     windowed variance and its log10 term, accumulated per luma bin by runs of consecutive samples and folded in a fixed order (a deterministic double sum; the reference
//     adds the same terms one by one in raster order, which differs from this sum in the last bits only — far below every threshold the decisions compare against); the
//     sample counts per bin and the first and second moments of Y, Cb, Cr are exact integers;
//   * the decision cascade is DATA: a table of rules (feature, comparison, constant) x (what to set), evaluated by a few-line interpreter.  A rule list is an else-if
//     chain: the first rule of a chain whose predicates all hold fires and ends the chain.  The thresholds are the reference's tuning constants (they define the models a
//     VTM user gets, so they cannot be anything else); the table says which source line each row restates;
//   * the code-word arithmetic (uint16 wrap-around included) is written once as "spread a budget, nudge by the bin statistics, take back what exceeds the total".
//
// Pinned by tests/golden/lmcs_analysis.npz: the models the reference's own EncReshape (compiled in place, oracle/_ref) chose for 29 pictures.
#include <hip/hip_runtime.h>
#include "vvcx.h"
#include <cmath>
#include <cstdint>
#include <cstring>
#include <vector>

extern "C" int vvcx_fail_msg_(int code, const char *msg);      // vvcx_api.hip: sets what vvcx_last_error returns

namespace {
constexpr int kBins = 16;            // luma bins of the analysis and of the model (PIC_CODE_CW_BINS)
constexpr int kRun = 32;             // consecutive samples of a row one thread folds into its per-bin partial sums
constexpr int kSlices = 16;          // the final fold: every bin's partials are cut into this many raster-ordered slices

// ------------------------------------------------------------------------------------------------ device statistics pass
struct StatOut {                     // written by the kernels (zeroed before)
  double logSum[kBins];              // per bin: sum over its samples of log10(windowed variance + 1)
  unsigned long long count[kBins];
  unsigned long long sum[3], sq[3];  // moments of Y, Cb, Cr
};

template <typename T>
__global__ void lmcs_row_sums(const T *luma, int stride, int w, int h, int win, uint32_t *rs, uint32_t *rq)
{
  const size_t i = (size_t) blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (size_t) w * h) return;
  const int y = (int) (i / w), x = (int) (i - (size_t) y * w);
  const int x1 = x - win < 0 ? 0 : x - win, x2 = x + win > w - 1 ? w - 1 : x + win;
  const T *row = luma + (size_t) y * stride;
  uint32_t s = 0, q = 0;
  for (int k = x1; k <= x2; k++) { const uint32_t v = row[k]; s += v; q += v * v; }
  rs[i] = s; rq[i] = q;
}

// one thread per run of kRun samples of a row: column sums of the row sums = window sums; the sample's term; per-bin partials of the run
template <typename T>
__global__ void lmcs_bin_terms(const T *luma, int stride, int w, int h, int win, int bd, const uint32_t *rs, const uint32_t *rq, double *part, uint32_t *pcnt, StatOut *out)
{
  const int runs_per_row = (w + kRun - 1) / kRun;
  const size_t r = (size_t) blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= (size_t) runs_per_row * h) return;
  const int y = (int) (r / runs_per_row), x0 = (int) (r - (size_t) y * runs_per_row) * kRun;
  const int y1 = y - win < 0 ? 0 : y - win, y2 = y + win > h - 1 ? h - 1 : y + win;
  const int down = bd - 10, log2BinLen = bd - 4;
  const double norm = (double) (1 << (2 * down));
  double acc[kBins]; uint32_t cnt[kBins];
  for (int b = 0; b < kBins; b++) { acc[b] = 0.0; cnt[b] = 0; }
  unsigned long long s1 = 0, s2 = 0;
  for (int x = x0; x < x0 + kRun && x < w; x++) {
    unsigned long long ws = 0, wq = 0;
    for (int k = y1; k <= y2; k++) { ws += rs[(size_t) k * w + x]; wq += rq[(size_t) k * w + x]; }
    const int x1 = x - win < 0 ? 0 : x - win, x2 = x + win > w - 1 ? w - 1 : x + win;
    const uint32_t n = (uint32_t) ((x2 - x1 + 1) * (y2 - y1 + 1));
    const double mean = (double) ws / n;
    double var = (double) wq / n - mean * mean;
    var = var / norm;
    const unsigned v = luma[(size_t) y * stride + x];
    const unsigned bin = (v >> down) >> log2BinLen;
    const double term = log10(var + 1.0);
    // (a select chain instead of acc[bin]: the accumulators stay in registers)
#pragma unroll
    for (int b = 0; b < kBins; b++) if (bin == (unsigned) b) { acc[b] += term; cnt[b]++; }
    s1 += v; s2 += (unsigned long long) v * v;
  }
  for (int b = 0; b < kBins; b++) { part[r * kBins + b] = acc[b]; pcnt[r * kBins + b] = cnt[b]; }
  atomicAdd(&out->sum[0], s1); atomicAdd(&out->sq[0], s2);
}

// fixed-order fold of the per-run partials: thread (slice, bin) adds its slice's runs in raster order, then the slices in order
__global__ void lmcs_fold(const double *part, const uint32_t *pcnt, size_t n_runs, StatOut *out)
{
  __shared__ double sl[kSlices][kBins]; __shared__ unsigned long long sc[kSlices][kBins];
  const int bin = threadIdx.x % kBins, slice = threadIdx.x / kBins;
  const size_t lo = n_runs * slice / kSlices, hi = n_runs * (slice + 1) / kSlices;
  double a = 0.0; unsigned long long c = 0;
  for (size_t r = lo; r < hi; r++) { a += part[r * kBins + bin]; c += pcnt[r * kBins + bin]; }
  sl[slice][bin] = a; sc[slice][bin] = c;
  __syncthreads();
  if (slice == 0) {
    double t = 0.0; unsigned long long n = 0;
    for (int k = 0; k < kSlices; k++) { t += sl[k][bin]; n += sc[k][bin]; }
    out->logSum[bin] = t; out->count[bin] = n;
  }
}

template <typename T>
__global__ void lmcs_chroma_moments(const T *cb, const T *cr, int stride_b, int stride_r, int wc, int hc, StatOut *out)
{
  const size_t i = (size_t) blockIdx.x * blockDim.x + threadIdx.x;
  unsigned long long u1 = 0, u2 = 0, v1 = 0, v2 = 0;
  if (i < (size_t) hc) {                                 // a thread per chroma row
    for (int x = 0; x < wc; x++) {
      const unsigned long long u = cb[i * stride_b + x], v = cr[i * stride_r + x];
      u1 += u; u2 += u * u; v1 += v; v2 += v * v;
    }
    atomicAdd(&out->sum[1], u1); atomicAdd(&out->sq[1], u2); atomicAdd(&out->sum[2], v1); atomicAdd(&out->sq[2], v2);
  }
}

struct DevMem {                      // frees what it allocated
  void *p = nullptr;
  ~DevMem() { if (p) (void) hipFree(p); }
  hipError_t alloc(size_t n) { return hipMalloc(&p, n ? n : 1); }
};

// the statistics of one picture whose planes are in device memory
template <typename T>
int picture_stats(const void *const org[3], const int stride[3], int w, int h, int bd, StatOut &st)
{
  const int win = (w < h ? w : h) / 240 > 0 ? (w < h ? w : h) / 240 : 1;
  const size_t np = (size_t) w * h, runs = (size_t) ((w + kRun - 1) / kRun) * h;
  DevMem rs, rq, part, pcnt, out;
  if (rs.alloc(np * 4) != hipSuccess || rq.alloc(np * 4) != hipSuccess || part.alloc(runs * kBins * 8) != hipSuccess || pcnt.alloc(runs * kBins * 4) != hipSuccess ||
      out.alloc(sizeof(StatOut)) != hipSuccess) return vvcx_fail_msg_(VVCX_ERR_DEVICE, "vvcx_lmcs_analyze: device allocation failed");
  if (hipMemset(out.p, 0, sizeof(StatOut)) != hipSuccess) return vvcx_fail_msg_(VVCX_ERR_DEVICE, "vvcx_lmcs_analyze: hipMemset failed");
  const unsigned T256 = 256;
  hipLaunchKernelGGL(lmcs_row_sums<T>, dim3((unsigned) ((np + T256 - 1) / T256)), dim3(T256), 0, 0, (const T *) org[0], stride[0], w, h, win, (uint32_t *) rs.p, (uint32_t *) rq.p);
  hipLaunchKernelGGL(lmcs_bin_terms<T>, dim3((unsigned) ((runs + T256 - 1) / T256)), dim3(T256), 0, 0, (const T *) org[0], stride[0], w, h, win, bd, (const uint32_t *) rs.p,
                     (const uint32_t *) rq.p, (double *) part.p, (uint32_t *) pcnt.p, (StatOut *) out.p);
  hipLaunchKernelGGL(lmcs_fold, dim3(1), dim3(kSlices * kBins), 0, 0, (const double *) part.p, (const uint32_t *) pcnt.p, runs, (StatOut *) out.p);
  hipLaunchKernelGGL(lmcs_chroma_moments<T>, dim3((unsigned) ((h / 2 + T256 - 1) / T256)), dim3(T256), 0, 0, (const T *) org[1], (const T *) org[2], stride[1], stride[2], w / 2, h / 2,
                     (StatOut *) out.p);
  if (hipGetLastError() != hipSuccess || hipDeviceSynchronize() != hipSuccess) return vvcx_fail_msg_(VVCX_ERR_DEVICE, "vvcx_lmcs_analyze: statistics kernels failed");
  if (hipMemcpy(&st, out.p, sizeof st, hipMemcpyDeviceToHost) != hipSuccess) return vvcx_fail_msg_(VVCX_ERR_DEVICE, "vvcx_lmcs_analyze: copy back failed");
  return VVCX_OK;
}

// ------------------------------------------------------------------------------------------------ from statistics to a model (host, scalar)
struct BinProfile {                  // what the cascade reads of a set of bins: occupancy, mean log-variance, the same relative to the picture's mean
  double share[kBins], lv[kBins], rel[kBins];
  double lvMin, lvMax, lvMean, wLv, wRel;
  // summary over the occupied bins (share above 0.1 %), EL/EncReshape.cpp:346-368
  void summarise()
  {
    int occupied = 0; lvMin = 5.0; lvMax = 0.0; lvMean = 0.0; wLv = 0.0; wRel = 0.0;
    for (int b = 0; b < kBins; b++) {
      if (!(share[b] > 0.001)) continue;
      occupied++; lvMean += lv[b];
      lvMax = lv[b] > lvMax ? lv[b] : lvMax; lvMin = lv[b] < lvMin ? lv[b] : lvMin;
    }
    lvMean /= (double) occupied;
    for (int b = 0; b < kBins; b++) {
      rel[b] = lvMean > 0.0 ? lv[b] / lvMean : 0.0;
      wLv += share[b] * lv[b]; wRel += share[b] * rel[b];
    }
  }
};

// ---- the decision cascade as data
enum Feat { F_HIST0, F_HIST1, F_HISTP, F_HISTE, F_ENDS, F_LV1, F_LVP, F_LVMAX, F_MAPMEAN, F_GAIN, F_GAINREL, F_TOP34, F_TOP28, F_TOP25, F_AREA, F_LOW, F_CHROMA, N_FEAT };
//   HIST0 / HIST1 / HISTP / HISTE: share of bins 0, 1, 14 (the last but one) and 15; ENDS = bin 0 + bin 15; LV1 / LVP: mean log-variance of bins 1 / 14; LVMAX: largest
//   over the occupied bins; MAPMEAN: mean log-variance after the trial mapping; GAIN / GAINREL: weighted (relative) log-variance after / before it; TOP34 / 28 / 25:
//   share of the picture in bins whose log-variance exceeds 3.4 / 2.8 / 2.5 (cumulated in descending order of log-variance, the reference's quirk of reading one entry
//   further included); AREA: samples of the picture; LOW: outcome of the "flat content" chain; CHROMA: sum of the chroma / luma deviation ratios
enum Cmp { GT, LT, EQ };
struct Pred { int8_t feat; int8_t cmp; double k; };
constexpr int kKeep = -1;
struct Rule { int8_t chain; Pred when[5]; int8_t intra, rate; int16_t cw; };      // chain: rules with the same number form one else-if chain; intra / rate / cw: kKeep = leave
constexpr Pred NONE = { -1, GT, 0.0 };
constexpr double kBigArea = 5184000.0;

// "flat content" (isLowCase, 1098-1126): any row
const Rule kLowRules[] = {
  { 0, { { F_AREA, GT, kBigArea }, NONE, NONE, NONE, NONE }, kKeep, kKeep, kKeep },
  { 0, { { F_LV1, GT, 4.0 }, NONE, NONE, NONE, NONE }, kKeep, kKeep, kKeep },
  { 0, { { F_MAPMEAN, GT, 3.4 }, { F_GAINREL, GT, 1.005 }, { F_GAIN, GT, 1.02 }, NONE, NONE }, kKeep, kKeep, kKeep },
  { 0, { { F_MAPMEAN, GT, 3.1 }, { F_GAINREL, GT, 1.005 }, { F_GAIN, GT, 1.04 }, NONE, NONE }, kKeep, kKeep, kKeep },
  { 0, { { F_MAPMEAN, GT, 2.8 }, { F_GAINREL, GT, 1.01 }, { F_GAIN, GT, 1.04 }, NONE, NONE }, kKeep, kKeep, kKeep },
};
// "leave the picture alone" (isSkipCase, 1081-1096): any row
const Rule kSkipRules[] = {
  { 0, { { F_ENDS, GT, 0.0001 }, { F_HISTP, LT, 0.001 }, { F_TOP25, GT, 0.8 }, { F_TOP28, GT, 0.4 }, { F_LVP, GT, 4.8 } }, kKeep, kKeep, kKeep },
  { 0, { { F_ENDS, GT, 0.0001 }, { F_HISTP, LT, 0.001 }, { F_TOP25, LT, 0.1 }, { F_TOP34, LT, 0.05 }, { F_LVP, LT, 4.0 } }, kKeep, kKeep, kKeep },
};
// LMCSUpdateCtrl 1 (the all-intra cfg), 1164-1227; the budget starts at 952 code words
const Rule kCtrl1[] = {
  { 0, { { F_LOW, GT, 0.5 }, { F_AREA, GT, kBigArea }, NONE, NONE, NONE }, kKeep, 1, 812 },
  { 1, { { F_LOW, GT, 0.5 }, { F_HISTP, GT, 0.1 }, NONE, NONE, NONE }, kKeep, 0, 924 },
  { 1, { { F_LOW, GT, 0.5 }, { F_HISTP, GT, 0.05 }, { F_HIST1, GT, 0.1 }, NONE, NONE }, kKeep, 0, 924 },
  { 1, { { F_LOW, GT, 0.5 }, { F_HISTP, GT, 0.05 }, NONE, NONE, NONE }, kKeep, 1, 812 },
  { 1, { { F_LOW, GT, 0.5 }, { F_TOP28, LT, 0.8 }, { F_TOP25, EQ, 1.0 }, NONE, NONE }, kKeep, 1, 896 },
  { 1, { { F_LOW, GT, 0.5 }, { F_TOP28, GT, 0.98 }, { F_HIST1, GT, 0.05 }, NONE, NONE }, kKeep, 0, 784 },
  { 1, { { F_LOW, GT, 0.5 }, { F_TOP28, LT, 0.1 }, NONE, NONE, NONE }, kKeep, 0, 1022 },
  { 2, { { F_HIST1, GT, 0.1 }, { F_LV1, GT, 1.8 }, { F_LV1, LT, 3.0 }, { F_LVP, GT, 1.2 }, { F_LVP, LT, 4.0 } }, kKeep, 1, 784 },
  { 2, { { F_HIST1, GT, 0.1 }, { F_LV1, GT, 1.8 }, { F_LV1, LT, 3.0 }, NONE, NONE }, kKeep, 1, kKeep },
  { 2, { { F_HISTP, LT, 0.001 }, { F_HIST1, GT, 0.05 }, { F_LV1, GT, 3.0 }, NONE, NONE }, kKeep, 1, 784 },
  { 2, { { F_HISTP, LT, 0.001 }, { F_HIST1, LT, 0.006 }, NONE, NONE, NONE }, kKeep, 0, 980 },
  { 2, { { F_HISTP, LT, 0.001 }, { F_TOP25, LT, 0.5 }, NONE, NONE, NONE }, kKeep, 0, 924 },
  { 2, { { F_HISTP, LT, 0.001 }, NONE, NONE, NONE, NONE }, kKeep, kKeep, kKeep },                        // an empty bin 14 without a match above ends the chain
  { 2, { { F_LVMAX, GT, 4.0 }, { F_MAPMEAN, GT, 3.2 }, { F_TOP28, LT, 0.25 }, NONE, NONE }, kKeep, 0, 980 },
  { 2, { { F_GAIN, LT, 1.03 }, NONE, NONE, NONE, NONE }, kKeep, 0, 980 },
};
// LMCSUpdateCtrl 0, 1128-1163; the budget starts at 1022 code words
const Rule kCtrl0[] = {
  { 0, { { F_LOW, GT, 0.5 }, NONE, NONE, NONE, NONE }, 0, 1, 980 },
  { 1, { { F_LOW, GT, 0.5 }, { F_HISTP, GT, 0.05 }, { F_LVP, LT, 1.2 }, NONE, NONE }, kKeep, kKeep, 938 },
  { 1, { { F_LOW, GT, 0.5 }, { F_HISTP, GT, 0.05 }, NONE, NONE, NONE }, kKeep, kKeep, 896 },
  { 1, { { F_LOW, GT, 0.5 }, { F_TOP28, LT, 0.8 }, { F_TOP25, EQ, 1.0 }, NONE, NONE }, kKeep, 1, 938 },
  { 2, { { F_HISTP, LT, 0.001 }, { F_HIST1, GT, 0.05 }, { F_LV1, GT, 3.0 }, NONE, NONE }, 1, 1, 784 },
  { 2, { { F_HISTP, LT, 0.001 }, { F_HIST1, LT, 0.006 }, NONE, NONE, NONE }, 0, 0, 1008 },
  { 2, { { F_HISTP, LT, 0.001 }, { F_TOP25, LT, 0.5 }, NONE, NONE, NONE }, 1, 0, 1022 },
  { 2, { { F_HISTP, LT, 0.001 }, NONE, NONE, NONE, NONE }, kKeep, kKeep, kKeep },
  { 2, { { F_LVMAX, GT, 4.0 }, { F_MAPMEAN, GT, 3.2 }, { F_TOP28, LT, 0.25 }, NONE, NONE }, 1, 0, 1022 },
  { 2, { { F_GAIN, LT, 1.03 }, NONE, NONE, NONE, NONE }, 1, 0, 1022 },
};

struct Choice { int intra, rate, cw; };

bool holds(const Rule &r, const double *f)
{
  for (const Pred &p : r.when) {
    if (p.feat < 0) continue;
    const double v = f[p.feat];
    if (!(p.cmp == GT ? v > p.k : p.cmp == LT ? v < p.k : v == p.k)) return false;
  }
  return true;
}
template <int N> bool any_rule(const Rule (&rules)[N], const double *f) { for (const Rule &r : rules) if (holds(r, f)) return true; return false; }
template <int N> void run_chains(const Rule (&rules)[N], const double *f, Choice &c)
{
  int done = -1;                                          // chain that has fired already
  for (const Rule &r : rules) {
    if (r.chain == done || !holds(r, f)) continue;
    done = r.chain;
    if (r.intra != kKeep) c.intra = r.intra;
    if (r.rate != kKeep) c.rate = r.rate;
    if (r.cw != kKeep) c.cw = r.cw;
  }
}

// ---- code words: uint16 arithmetic like the reference's (sums wrap at 65536)
struct CodeWords {
  uint16_t cw[kBins];
  int total() const { int s = 0; for (uint16_t v : cw) s += v; return s; }
  // an even share of `budget` for bins lo..hi, then occupied bins gain words where the content is flat and lose them where it is busy (931-954)
  void spread(int lo, int hi, int budget, const BinProfile &p)
  {
    const uint16_t even = (uint16_t) (uint32_t) std::round((double) (uint16_t) budget / (hi - lo + 1));
    for (int b = 0; b < kBins; b++) cw[b] = (b >= lo && b <= hi) ? even : (uint16_t) 0;
    for (int b = 0; b < kBins; b++) {
      if (!(p.share[b] > 0.001)) continue;
      const double s = p.share[b] > 0.4 ? 0.4 : p.share[b];
      const uint16_t one = (uint16_t) (10.0 * s + 0.5), two = (uint16_t) (20.0 * s + 0.5);
      const double r = p.rel[b];
      if (r < 0.8) cw[b] = (uint16_t) (cw[b] + two); else if (r < 0.9) cw[b] = (uint16_t) (cw[b] + one);
      if (r > 1.2) cw[b] = (uint16_t) (cw[b] - two); else if (r > 1.1) cw[b] = (uint16_t) (cw[b] - one);
    }
  }
  // what exceeds `limit` is taken back evenly from bins lo..hi, the remainder one word at a time from the first non-empty ones (955-975)
  void trim(int lo, int hi, int limit)
  {
    const int over = total() - limit;
    if (over <= 0) return;
    const int n = hi - lo + 1, each = over / n;
    int rest = over - each * n;
    if (each > 0) for (int b = lo; b <= hi; b++) cw[b] = (uint16_t) (cw[b] - each);
    for (int b = lo; b <= hi && rest > 0; b++) if (cw[b] > 0) { cw[b]--; rest--; }
  }
};

// Model borders must not share a 32-word segment of the mapped range unless they sit on its start (the decoder finds a sample's bin by segment, JVET_O0272; 2194-2256).
// Walking up the bins: a border inside the segment of the previous one is pushed to the next segment start and the words it gained are taken from the bins above, which
// keep at least an eighth of the original bin width; in the last segment the remaining bins are merged into the bin below.  Returns the last non-empty bin.
int align_borders(CodeWords &m, int lo, int hi, int wordsPerBin)
{
  const int seg = 5;                                      // log2 of 2 * kBins
  const int floorCw = wordsPerBin >> 3;
  int16_t edge[kBins + 1];
  edge[0] = 0;
  for (int b = 0; b < kBins; b++) edge[b + 1] = (int16_t) (edge[b] + m.cw[b]);
  const int lastSeg = edge[hi + 1] >> seg;
  for (int b = lo; b <= hi; b++) {
    edge[b + 1] = (int16_t) (edge[b] + m.cw[b]);
    const int here = edge[b] >> seg;
    if (here != (edge[b + 1] >> seg) || edge[b] == (here << seg)) continue;      // the bin crosses a segment start, or begins on one
    if (here == lastSeg) {                                // nothing above can absorb a shift: bins b.. collapse onto the top border
      edge[b] = edge[hi + 1];
      for (int k = b; k <= hi; k++) { edge[k + 1] = edge[b]; m.cw[k] = 0; }
      m.cw[b - 1] = (uint16_t) (edge[b] - edge[b - 1]);
      break;
    }
    int16_t owe = (int16_t) (((here + 1) << seg) - edge[b + 1]);
    edge[b + 1] = (int16_t) (edge[b + 1] + owe);
    m.cw[b] = (uint16_t) (m.cw[b] + owe);
    for (int k = b + 1; k <= hi && owe != 0; k++) {
      if (m.cw[k] < owe + floorCw) { owe = (int16_t) (owe - (m.cw[k] - floorCw)); m.cw[k] = (uint16_t) floorCw; }
      else { m.cw[k] = (uint16_t) (m.cw[k] - owe); owe = 0; }
    }
  }
  int top = hi;
  for (int b = kBins - 1; b >= 0; b--) if (m.cw[b] > 0) { top = b; break; }
  return top;
}

// the three "share of the picture above a log-variance level" features: bins in descending order of log-variance (stable: an equal value stays behind the earlier bin),
// cumulated shares, and for each level the cumulated share at the index AFTER the last bin above it (index 0 when none is) - 909-928, 1005-1030
void busy_shares(const BinProfile &p, double &top34, double &top28, double &top25)
{
  int order[kBins]; double cum[kBins];
  for (int b = 0; b < kBins; b++) order[b] = b;
  for (int i = 1; i < kBins; i++) {                       // insertion sort, descending, stable = what a bubble sort with a strict comparison yields
    const int t = order[i]; int j = i;
    while (j > 0 && p.lv[order[j - 1]] < p.lv[t]) { order[j] = order[j - 1]; j--; }
    order[j] = t;
  }
  double run = 0.0;
  for (int i = 0; i < kBins; i++) { run += p.share[order[i]]; cum[i] = run; }
  int a = 0, b2 = 0, c = 0;
  for (int i = 0; i < kBins - 1; i++) { const double v = p.lv[order[i]]; if (v > 3.4) a = i + 1; if (v > 2.8) b2 = i + 1; if (v > 2.5) c = i + 1; }
  top34 = cum[a]; top28 = cum[b2]; top25 = cum[c];
}

int analyse(const StatOut &st, int w, int h, int bd, int qp, int updateCtrl, vvcx_slice *slice)
{
  const double area = (double) w * h, areaC = (double) (w / 2) * (h / 2);
  BinProfile src;
  for (int b = 0; b < kBins; b++) {
    src.share[b] = (double) st.count[b] / (double) (w * h);
    src.lv[b] = st.count[b] ? st.logSum[b] / (double) st.count[b] : 0.0;
  }
  src.summarise();
  // chroma against luma: ratios of the standard deviations (370-407)
  double devU = 0.0, devV = 0.0;
  {
    const double mY = (double) st.sum[0] / area, mU = (double) st.sum[1] / areaC, mV = (double) st.sum[2] / areaC;
    const double vY = (double) st.sq[0] / area - mY * mY, vU = (double) st.sq[1] / areaC - mU * mU, vV = (double) st.sq[2] / areaC - mV * mV;
    if (vY > 0) { devU = std::sqrt(vU) / std::sqrt(vY); devV = std::sqrt(vV) / std::sqrt(vY); }
  }
  const int words = 1 << bd, wordsPerBin = words / kBins;                       // mapped range and an untouched bin's share of it
  const int unit = bd > 10 ? (wordsPerBin >> (bd - 10)) : wordsPerBin;          // the analysis counts in 10-bit words
  const int legalLo = 16 << (bd - 8), legalHi = 235 << (bd - 8);
  int lo = legalLo / wordsPerBin, hi = legalHi / wordsPerBin;                   // bins of the limited range
  bool intra = true, inter = true;
  // 443-470: content outside the limited range widens the model or rules it out; chroma-heavy dark content rules it out
  if (src.share[kBins - 1] > 0.0003 || src.share[0] > 0.03) intra = inter = false;
  if (src.share[0] + src.share[kBins - 1] > 0.005)
    for (int b = 0; b < kBins; b++) if (src.share[b] > 0) { lo = b < lo ? b : lo; hi = b > hi ? b : hi; }
  if (devU + devV > 1.5 && src.share[1] > 0.5) intra = inter = false;
  const int chromaAdj = !(devU > 0.36 && devV > 0.2 && area > kBigArea);
  Choice pick = { 1, 0, 1022 };
  CodeWords m;
  for (int b = 0; b < kBins; b++) m.cw[b] = (uint16_t) unit;
  if (inter) {
    // trial mapping with the full budget, to see what it does to the variances (1058-1079)
    m.spread(lo, hi, 1022, src); m.trim(lo, hi, 1023);
    BinProfile mapped;
    for (int b = 0; b < kBins; b++) {
      mapped.share[b] = src.share[b];
      mapped.lv[b] = src.lv[b] + 2.0 * std::log10(m.cw[b] > 0 ? (double) m.cw[b] / (double) unit : 1.0);
    }
    mapped.summarise();
    double f[N_FEAT];
    f[F_HIST0] = src.share[0]; f[F_HIST1] = src.share[1]; f[F_HISTP] = src.share[kBins - 2]; f[F_HISTE] = src.share[kBins - 1]; f[F_ENDS] = src.share[0] + src.share[kBins - 1];
    f[F_LV1] = src.lv[1]; f[F_LVP] = src.lv[kBins - 2]; f[F_LVMAX] = src.lvMax; f[F_MAPMEAN] = mapped.lvMean;
    f[F_GAIN] = mapped.wLv / src.wLv; f[F_GAINREL] = mapped.wRel / src.wRel;
    busy_shares(src, f[F_TOP34], f[F_TOP28], f[F_TOP25]);
    f[F_AREA] = area; f[F_CHROMA] = devU + devV; f[F_LOW] = 0.0;
    if (any_rule(kSkipRules, f)) intra = inter = false;
    else {
      f[F_LOW] = any_rule(kLowRules, f) ? 1.0 : 0.0;
      pick.intra = intra; pick.rate = 0;
      if (updateCtrl == 0) { pick.cw = 1022; run_chains(kCtrl0, f, pick); }
      else { pick.cw = 952; run_chains(kCtrl1, f, pick); }
      intra = pick.intra != 0;
    }
  }
  if (pick.rate == 2 && qp <= 22) intra = inter = false;
  if (!intra) return VVCX_OK;                             // no model, or one that only inter pictures would use
  if (pick.rate == 1 && qp <= 22) { for (int b = 0; b < kBins; b++) m.cw[b] = (b >= lo && b <= hi) ? (uint16_t) (unit + 2) : (uint16_t) 0; }
  else m.spread(lo, hi, pick.cw, src);
  m.trim(lo, hi, 1023);
  // the model in words of the picture's bit depth, its first and last used bin, borders aligned (1835-1893)
  if (bd != 10) for (int b = 0; b < kBins; b++) m.cw[b] = (uint16_t) (m.cw[b] * (1 << (bd - 10)));
  int first = 0, last = kBins - 1;
  for (int b = 0; b < kBins; b++) if (m.cw[b] > 0) { first = b; break; }
  for (int b = kBins - 1; b >= 0; b--) if (m.cw[b] > 0) { last = b; break; }
  last = align_borders(m, first, last, 1024 / kBins);      // (the alignment works with the 10-bit bin width whatever the bit depth, like the reference's)
  slice->lmcs_enable = 1; slice->lmcs_chroma_adj = chromaAdj; slice->lmcs_min_bin = first; slice->lmcs_max_bin = last;
  for (int b = first; b <= last; b++) slice->lmcs_delta_cw[b] = (int) m.cw[b] - wordsPerBin;
  return VVCX_OK;
}

int check_args(const void *const org[3], const int stride[3], int pic_w, int pic_h, int bit_depth, int update_ctrl, vvcx_slice *slice)
{
  if (!org || !stride || !slice || !org[0] || !org[1] || !org[2]) return vvcx_fail_msg_(VVCX_ERR_ARG, "vvcx_lmcs_analyze: null argument");
  if (pic_w < 8 || pic_h < 8 || pic_w > 16384 || pic_h > 16384 || (pic_w & 1) || (pic_h & 1) || stride[0] < pic_w || stride[1] < pic_w / 2 || stride[2] < pic_w / 2)
    return vvcx_fail_msg_(VVCX_ERR_ARG, "vvcx_lmcs_analyze: picture size (8..16384, even) / strides");
  if (bit_depth < 8 || bit_depth > 12) return vvcx_fail_msg_(VVCX_ERR_ARG, "vvcx_lmcs_analyze: bit depth");
  if (update_ctrl != 0 && update_ctrl != 1) return vvcx_fail_msg_(VVCX_ERR_UNSUPPORTED, "vvcx_lmcs_analyze: LMCSUpdateCtrl 2 (low delay) is not an intra configuration");
  return VVCX_OK;
}
void no_model(vvcx_slice *slice)
{
  slice->lmcs_enable = 0; slice->lmcs_chroma_adj = 0; slice->lmcs_min_bin = 0; slice->lmcs_max_bin = 0;
  for (int i = 0; i < kBins; i++) slice->lmcs_delta_cw[i] = 0;
}
}  // namespace

// planes in DEVICE memory (the pointers the caller later hands to vvcx_bind_frames), on the current device
extern "C" int vvcx_lmcs_analyze_device(const void *const org[3], const int stride[3], int pic_w, int pic_h, int bit_depth, int slice_qp, int update_ctrl, vvcx_slice *slice)
{
  const int rc = check_args(org, stride, pic_w, pic_h, bit_depth, update_ctrl, slice);
  if (rc != VVCX_OK) return rc;
  no_model(slice);
  // Below 10 bits the reference bins the luma with a negative shift count (on x86 the count is masked: every sample lands in bin 0), sees all of the picture in its first
  // bin and ends the analysis at its first test: LMCS stays off for every 8-bit picture.  Nothing to measure.
  if (bit_depth < 10) return VVCX_OK;
  StatOut st;
  const int rs = picture_stats<uint16_t>(org, stride, pic_w, pic_h, bit_depth, st);
  if (rs != VVCX_OK) return rs;
  return analyse(st, pic_w, pic_h, bit_depth, slice_qp, update_ctrl, slice);
}

// planes in HOST memory: uploaded, then the same device pass
extern "C" int vvcx_lmcs_analyze(const void *const org[3], const int stride[3], int pic_w, int pic_h, int bit_depth, int slice_qp, int update_ctrl, vvcx_slice *slice)
{
  const int rc = check_args(org, stride, pic_w, pic_h, bit_depth, update_ctrl, slice);
  if (rc != VVCX_OK) return rc;
  no_model(slice);
  if (bit_depth < 10) return VVCX_OK;
  DevMem d[3]; const void *dev[3];
  for (int c = 0; c < 3; c++) {
    const size_t bytes = (size_t) stride[c] * (c ? pic_h / 2 : pic_h) * 2;
    if (d[c].alloc(bytes) != hipSuccess || hipMemcpy(d[c].p, org[c], bytes, hipMemcpyHostToDevice) != hipSuccess) return vvcx_fail_msg_(VVCX_ERR_DEVICE, "vvcx_lmcs_analyze: upload of the picture failed");
    dev[c] = d[c].p;
  }
  return vvcx_lmcs_analyze_device(dev, stride, pic_w, pic_h, bit_depth, slice_qp, update_ctrl, slice);
}

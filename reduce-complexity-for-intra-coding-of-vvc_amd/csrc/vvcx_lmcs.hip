// vvcx_lmcs.hip — host side of LMCS that is not the search: the picture analysis which chooses the model of an intra picture.
//
// ≙ EncGOP::xPicInitLMCS (EL/EncGOP.cpp:1624-1700) → EncReshape::preAnalyzerLMCS (EL/EncReshape.cpp:410-559) with calcSeqStats (166-409) and
// deriveReshapeParametersSDR (977-1229), cwPerturbation / cwReduction (931-975), bubbleSortDsd (909-928), then the model half of constructReshaperLMCS
// (1835-1893) with adjustLmcsPivot (2194-2256), for the reference cfg's LMCSSignalType 0 (SDR) and LMCSAdpOption 0, LMCSUpdateCtrl 0 or 1, intra slices.
// Pure picture-level control (a handful of reductions over the original picture and threshold logic in doubles); the arithmetic follows the reference operation
// by operation — the per-bin sums of log10(variance + 1) are accumulated in raster order like there — so that the model is the reference's bit for bit
// (tests/golden/lmcs_analysis.npz: models the reference's own EncReshape, compiled in place, chose for the same pictures).  The LUTs of a model are built by
// vvcx_set_slice (Reshape::constructReshaper).
#include "vvcx.h"
#include <cmath>
#include <cstdint>
#include <cstring>
#include <vector>
#include <algorithm>

extern "C" int vvcx_fail_msg_(int code, const char *msg);      // vvcx_api.hip: sets what vvcx_last_error returns

namespace {
const int kBins = 16;                // PIC_CODE_CW_BINS (CL/CommonDef.h:517)
const int kSegSize = kBins << 1;     // LMCS_SEG_SIZE (519)

struct SeqInfo { double binVar[kBins], binHist[kBins], normVar[kBins]; int nonZeroCnt; double weightVar, weightNorm, minBinVar, maxBinVar, meanBinVar, ratioStdU, ratioStdV; };
void initSeqStats(SeqInfo &s) { memset(&s, 0, sizeof s); }

inline int px(const void *p, int bps, size_t i) { return bps == 1 ? (int) ((const uint8_t *) p)[i] : (int) ((const uint16_t *) p)[i]; }

// min / max / mean over the occupied bins, normalised variances, weighted sums (EL/EncReshape.cpp:346-368 and again 1034-1056)
void binSummary(SeqInfo &s)
{
  s.minBinVar = 5.0; s.maxBinVar = 0.0; s.meanBinVar = 0.0; s.nonZeroCnt = 0;
  for (int b = 0; b < kBins; b++)
    if (s.binHist[b] > 0.001) {
      s.nonZeroCnt++; s.meanBinVar += s.binVar[b];
      if (s.binVar[b] > s.maxBinVar) s.maxBinVar = s.binVar[b];
      if (s.binVar[b] < s.minBinVar) s.minBinVar = s.binVar[b];
    }
  s.meanBinVar /= (double) s.nonZeroCnt;
  for (int b = 0; b < kBins; b++) {
    if (s.meanBinVar > 0.0) s.normVar[b] = s.binVar[b] / s.meanBinVar;
    s.weightVar += s.binHist[b] * s.binVar[b];
    s.weightNorm += s.binHist[b] * s.normVar[b];
  }
}

// calcSeqStats (166-409).  The reference slides the window sums along; the sums are exact integers, so summed-area tables give the same numbers
void calcSeqStats(const void *const org[3], const int stride[3], int w, int h, int bd, int bps, int picSize, SeqInfo &st)
{
  const int lutSize = 1 << bd, binLen = lutSize / kBins;
  int win = std::min(h, w) / 240;
  win = win > 0 ? win : 1;
  std::vector<int64_t> S((size_t) (w + 1) * (h + 1), 0), Q((size_t) (w + 1) * (h + 1), 0);
  for (int y = 0; y < h; y++) {
    int64_t rs = 0, rq = 0;
    for (int x = 0; x < w; x++) {
      const int64_t v = px(org[0], bps, (size_t) y * stride[0] + x);
      rs += v; rq += v * v;
      S[(size_t) (y + 1) * (w + 1) + x + 1] = S[(size_t) y * (w + 1) + x + 1] + rs;
      Q[(size_t) (y + 1) * (w + 1) + x + 1] = Q[(size_t) y * (w + 1) + x + 1] + rq;
    }
  }
  uint32_t binCnt[kBins] = { 0 };
  initSeqStats(st);
  for (int y = 0; y < h; y++) {
    const int y1 = std::max(y - win, 0), y2 = std::min(y + win, h - 1);
    for (int x = 0; x < w; x++) {
      const int x1 = std::max(x - win, 0), x2 = std::min(x + win, w - 1);
      const uint32_t n = (uint32_t) ((x2 - x1 + 1) * (y2 - y1 + 1));
#define SAT(T_, xa, ya, xb, yb) (T_[(size_t) ((yb) + 1) * (w + 1) + (xb) + 1] - T_[(size_t) (ya) * (w + 1) + (xb) + 1] - T_[(size_t) ((yb) + 1) * (w + 1) + (xa)] + T_[(size_t) (ya) * (w + 1) + (xa)])
      const int64_t sum = SAT(S, x1, y1, x2, y2), sumSq = SAT(Q, x1, y1, x2, y2);
#undef SAT
      const double average = double(sum) / n;
      double variance = double(sumSq) / n - average * average;
      variance = variance / (double) (1 << (2 * (bd - 10)));
      const int v = px(org[0], bps, (size_t) y * stride[0] + x);
      const uint32_t binIdx = (uint32_t) ((v >> (bd - 10)) / binLen);
      st.binVar[binIdx] += log10(variance + 1.0);
      binCnt[binIdx]++;
    }
  }
  for (int b = 0; b < kBins; b++) {
    st.binHist[b] = (double) binCnt[b] / (double) picSize;
    st.binVar[b] = binCnt[b] > 0 ? st.binVar[b] / binCnt[b] : 0.0;
  }
  binSummary(st);
  const int wc = w >> 1, hc = h >> 1;
  double avgY = 0.0, avgU = 0.0, avgV = 0.0, varY = 0.0, varU = 0.0, varV = 0.0;
  for (int y = 0; y < h; y++) for (int x = 0; x < w; x++) { const int v = px(org[0], bps, (size_t) y * stride[0] + x); avgY += v; varY += v * v; }
  for (int y = 0; y < hc; y++) for (int x = 0; x < wc; x++) {
    const int u = px(org[1], bps, (size_t) y * stride[1] + x), v = px(org[2], bps, (size_t) y * stride[2] + x);
    avgU += u; avgV += v; varU += u * u; varV += v * v;
  }
  avgY = avgY / (w * h); avgU = avgU / (wc * hc); avgV = avgV / (wc * hc);
  varY = varY / (w * h) - avgY * avgY; varU = varU / (wc * hc) - avgU * avgU; varV = varV / (wc * hc) - avgV * avgV;
  if (varY > 0) { st.ratioStdU = sqrt(varU) / sqrt(varY); st.ratioStdV = sqrt(varV) / sqrt(varY); }
}

struct Analyzer {
  int bd, lutSize, initCW, initCWAnalyze, picSize, baseQP, updateCtrl;
  uint16_t binCW[32];
  int cw0, cw1;                 // m_reshapeCW.binCW[0 / 1]
  int minBin, maxBin, chromaAdj, rateAdpMode, tcase;
  bool useAdpCW;
  SeqInfo src, rsp;

  void cwPerturbation(int startBin, int endBin, uint16_t maxCW)        // 931-954
  {
    for (int i = 0; i < kBins; i++) binCW[i] = (i >= startBin && i <= endBin) ? (uint16_t) (uint32_t) round((double) maxCW / (endBin - startBin + 1)) : (uint16_t) 0;
    for (int i = 0; i < kBins; i++)
      if (src.binHist[i] > 0.001) {
        const double hist = src.binHist[i] > 0.4 ? 0.4 : src.binHist[i];
        const uint16_t delta1 = (uint16_t) (10.0 * hist + 0.5), delta2 = (uint16_t) (20.0 * hist + 0.5);
        if (src.normVar[i] < 0.8) binCW[i] = (uint16_t) (binCW[i] + delta2);
        else if (src.normVar[i] < 0.9) binCW[i] = (uint16_t) (binCW[i] + delta1);
        if (src.normVar[i] > 1.2) binCW[i] = (uint16_t) (binCW[i] - delta2);
        else if (src.normVar[i] > 1.1) binCW[i] = (uint16_t) (binCW[i] - delta1);
      }
  }
  int totCW() const { const int s = bd - 10; return s != 0 ? (s > 0 ? lutSize / (1 << s) : lutSize * (1 << (-s))) : lutSize; }
  void cwReduction(int startBin, int endBin)                           // 955-975
  {
    const int maxAllowedCW = totCW() - 1;
    int usedCW = 0;
    for (int i = 0; i < kBins; i++) usedCW += binCW[i];
    if (usedCW > maxAllowedCW) {
      const int deltaCW = usedCW - maxAllowedCW, divCW = deltaCW / (endBin - startBin + 1);
      int modCW = deltaCW - divCW * (endBin - startBin + 1);
      if (divCW > 0) for (int i = startBin; i <= endBin; i++) binCW[i] = (uint16_t) (binCW[i] - divCW);
      for (int i = startBin; i <= endBin; i++) { if (modCW == 0) break; if (binCW[i] > 0) { binCW[i]--; modCW--; } }
    }
  }
  // deriveReshapeParametersSDR (977-1229)
  void deriveSDR(bool *intraAdp, bool *interAdp)
  {
    bool isSkipCase = false, isLowCase = false;
    int first1 = 0, first2 = 0, first3 = 0;
    int idx[kBins]; double var[kBins], cdf[kBins];
    for (int b = 0; b < kBins; b++) { var[b] = src.binVar[b]; idx[b] = b; }
    for (int i = 0; i < kBins - 1; i++) {                               // bubbleSortDsd 909-928
      bool swapped = false;
      for (int j = 0; j < kBins - i - 1; j++) if (var[j] < var[j + 1]) { std::swap(var[j], var[j + 1]); std::swap(idx[j], idx[j + 1]); swapped = true; }
      if (!swapped) break;
    }
    cdf[0] = src.binHist[idx[0]];
    for (int b = 1; b < kBins; b++) cdf[b] = cdf[b - 1] + src.binHist[idx[b]];
    for (int b = 0; b < kBins - 1; b++) { if (var[b] > 3.4) first1 = b + 1; if (var[b] > 2.8) first2 = b + 1; if (var[b] > 2.5) first3 = b + 1; }
    const double perc1 = cdf[first1], perc2 = cdf[first2], perc3 = cdf[first3];
    cwPerturbation(minBin, maxBin, (uint16_t) cw1);
    cwReduction(minBin, maxBin);
    initSeqStats(rsp);
    for (int b = 0; b < kBins; b++) {
      const double scale = binCW[b] > 0 ? (double) binCW[b] / (double) initCWAnalyze : 1.0;
      rsp.binHist[b] = src.binHist[b];
      rsp.binVar[b] = src.binVar[b] + 2.0 * log10(scale);
    }
    binSummary(rsp);
    const double ratioWeiVar = rsp.weightVar / src.weightVar, ratioWeiVarNorm = rsp.weightNorm / src.weightNorm;
    const int n = kBins;
    if ((src.binHist[0] + src.binHist[n - 1]) > 0.0001 && src.binHist[n - 2] < 0.001) {
      if (perc3 > 0.8 && perc2 > 0.4 && src.binVar[n - 2] > 4.8) isSkipCase = true;
      else if (perc3 < 0.1 && perc1 < 0.05 && src.binVar[n - 2] < 4.0) isSkipCase = true;
    }
    if (isSkipCase) { *intraAdp = false; *interAdp = false; return; }
    if (picSize > 5184000) isLowCase = true;
    else if (src.binVar[1] > 4.0) isLowCase = true;
    else if (rsp.meanBinVar > 3.4 && ratioWeiVarNorm > 1.005 && ratioWeiVar > 1.02) isLowCase = true;
    else if (rsp.meanBinVar > 3.1 && ratioWeiVarNorm > 1.005 && ratioWeiVar > 1.04) isLowCase = true;
    else if (rsp.meanBinVar > 2.8 && ratioWeiVarNorm > 1.01 && ratioWeiVar > 1.04) isLowCase = true;
    if (updateCtrl == 0) {
      cw1 = 1022;
      if (isLowCase) {
        *intraAdp = false; rateAdpMode = 1; cw1 = 980;
        if (src.binHist[n - 2] > 0.05) { cw1 = 896; if (src.binVar[n - 2] < 1.2) cw1 = 938; }
        else if (perc2 < 0.8 && perc3 == 1.0) { rateAdpMode = 1; cw1 = 938; }
      }
      if (src.binHist[n - 2] < 0.001) {
        if (src.binHist[1] > 0.05 && src.binVar[1] > 3.0) { *intraAdp = true; rateAdpMode = 1; cw1 = 784; }
        else if (src.binHist[1] < 0.006) { *intraAdp = false; rateAdpMode = 0; cw1 = 1008; }
        else if (perc3 < 0.5) { *intraAdp = true; rateAdpMode = 0; cw1 = 1022; }
      } else if ((src.maxBinVar > 4.0 && rsp.meanBinVar > 3.2 && perc2 < 0.25) || ratioWeiVar < 1.03) { *intraAdp = true; rateAdpMode = 0; cw1 = 1022; }
      if (*intraAdp == true && rateAdpMode == 0) tcase = 9;
    } else {                                                             // updateCtrl == 1 (the all-intra cfg)
      cw1 = 952;
      if (isLowCase) {
        if (picSize > 5184000) { rateAdpMode = 1; cw1 = 812; }
        if (src.binHist[n - 2] > 0.05) {
          rateAdpMode = 1; cw1 = 812;
          if (src.binHist[n - 2] > 0.1 || src.binHist[1] > 0.1) { rateAdpMode = 0; cw1 = 924; }
        } else if (perc2 < 0.8 && perc3 == 1.0) { rateAdpMode = 1; cw1 = 896; }
        else if (perc2 > 0.98 && src.binHist[1] > 0.05) { rateAdpMode = 0; cw1 = 784; }
        else if (perc2 < 0.1) { rateAdpMode = 0; cw1 = 1022; }
      }
      if (src.binHist[1] > 0.1 && (src.binVar[1] > 1.8 && src.binVar[1] < 3.0)) {
        rateAdpMode = 1;
        if (src.binVar[n - 2] > 1.2 && src.binVar[n - 2] < 4.0) cw1 = 784;
      } else if (src.binHist[n - 2] < 0.001) {
        if (src.binHist[1] > 0.05 && src.binVar[1] > 3.0) { rateAdpMode = 1; cw1 = 784; }
        else if (src.binHist[1] < 0.006) { rateAdpMode = 0; cw1 = 980; }
        else if (perc3 < 0.5) { rateAdpMode = 0; cw1 = 924; }
      } else if ((src.maxBinVar > 4.0 && rsp.meanBinVar > 3.2 && perc2 < 0.25) || ratioWeiVar < 1.03) { rateAdpMode = 0; cw1 = 980; }
    }
  }
  // adjustLmcsPivot (2194-2256): every pivot on a segment border or alone in its segment (JVET_O0272)
  void adjustLmcsPivot()
  {
    const int orgCW = totCW() / kBins;
    int log2Seg = 0; while ((1 << (log2Seg + 1)) <= kSegSize) log2Seg++;
    int16_t pivot[kBins + 1];
    pivot[0] = 0;
    for (int i = 0; i < kBins; i++) pivot[i + 1] = (int16_t) (pivot[i] + binCW[i]);
    const int segIdxMax = pivot[maxBin + 1] >> log2Seg;
    for (int i = minBin; i <= maxBin; i++) {
      pivot[i + 1] = (int16_t) (pivot[i] + binCW[i]);
      const int segCurr = pivot[i] >> log2Seg, segNext = pivot[i + 1] >> log2Seg;
      if (segCurr == segNext && pivot[i] != (segCurr << log2Seg)) {
        if (segCurr == segIdxMax) {
          pivot[i] = pivot[maxBin + 1];
          for (int j = i; j <= maxBin; j++) { pivot[j + 1] = pivot[i]; binCW[j] = 0; }
          binCW[i - 1] = (uint16_t) (pivot[i] - pivot[i - 1]);
          break;
        } else {
          int16_t adjustVal = (int16_t) (((segCurr + 1) << log2Seg) - pivot[i + 1]);
          pivot[i + 1] = (int16_t) (pivot[i + 1] + adjustVal);
          binCW[i] = (uint16_t) (binCW[i] + adjustVal);
          for (int j = i + 1; j <= maxBin; j++) {
            if (binCW[j] < (adjustVal + (orgCW >> 3))) { adjustVal = (int16_t) (adjustVal - (binCW[j] - (orgCW >> 3))); binCW[j] = (uint16_t) (orgCW >> 3); }
            else { binCW[j] = (uint16_t) (binCW[j] - adjustVal); adjustVal = 0; }
            if (adjustVal == 0) break;
          }
        }
      }
    }
    for (int i = kBins - 1; i >= 0; i--) if (binCW[i] > 0) { maxBin = i; break; }
  }
};
}  // namespace

extern "C" int vvcx_lmcs_analyze(const void *const org[3], const int stride[3], int pic_w, int pic_h, int bit_depth, int slice_qp, int update_ctrl, vvcx_slice *slice)
{
  if (!org || !stride || !slice || !org[0] || !org[1] || !org[2]) return vvcx_fail_msg_(VVCX_ERR_ARG, "vvcx_lmcs_analyze: null argument");
  if (pic_w < 8 || pic_h < 8 || (pic_w & 1) || (pic_h & 1) || stride[0] < pic_w || stride[1] < pic_w / 2 || stride[2] < pic_w / 2) return vvcx_fail_msg_(VVCX_ERR_ARG, "vvcx_lmcs_analyze: picture size / strides");
  if (bit_depth < 8 || bit_depth > 12) return vvcx_fail_msg_(VVCX_ERR_ARG, "vvcx_lmcs_analyze: bit depth");
  if (update_ctrl != 0 && update_ctrl != 1) return vvcx_fail_msg_(VVCX_ERR_UNSUPPORTED, "vvcx_lmcs_analyze: LMCSUpdateCtrl 2 (low delay) is not an intra configuration");      // 2 (low delay) analyses 32 bins and fits the codewords to the variances: not an intra configuration
  slice->lmcs_enable = 0; slice->lmcs_chroma_adj = 0; slice->lmcs_min_bin = 0; slice->lmcs_max_bin = 0;
  for (int i = 0; i < 16; i++) slice->lmcs_delta_cw[i] = 0;
  // Below 10 bits calcSeqStats bins the luma with `>> (m_lumaBD - 10)`, a negative shift count: on x86 the count is masked and every sample lands in bin 0, whose share
  // of 100 % ends the analysis at the first test (binHist[0] > 0.03: 443-446) — the reference switches LMCS off for every 8-bit picture
  if (bit_depth < 10) return VVCX_OK;
  const int bps = 2;
  Analyzer a; memset(&a, 0, sizeof a);
  a.bd = bit_depth; a.lutSize = 1 << bit_depth; a.initCW = a.lutSize / kBins; a.picSize = pic_w * pic_h; a.baseQP = slice_qp; a.updateCtrl = update_ctrl;
  const int stdMin = 16 << (bit_depth - 8), stdMax = 235 << (bit_depth - 8), binLen = a.lutSize / kBins;
  int startBin = stdMin / binLen, endBin = stdMax / binLen;
  a.minBin = startBin; a.maxBin = endBin;
  a.initCWAnalyze = bit_depth > 10 ? (binLen >> (bit_depth - 10)) : binLen;
  for (int b = 0; b < kBins; b++) a.binCW[b] = (uint16_t) a.initCWAnalyze;
  bool reshape = true, exceedSTD = false, intraAdp = true, interAdp = true;
  a.useAdpCW = false; a.chromaAdj = 1; a.rateAdpMode = 0; a.tcase = 0;
  calcSeqStats(org, stride, pic_w, pic_h, bit_depth, bps, a.picSize, a.src);
  const SeqInfo &s = a.src;
  if ((s.binHist[0] + s.binHist[kBins - 1]) > 0.005) exceedSTD = true;
  if (s.binHist[kBins - 1] > 0.0003) { intraAdp = false; interAdp = false; }
  if (s.binHist[0] > 0.03) { intraAdp = false; interAdp = false; }
  if (exceedSTD) {
    for (int i = 0; i < kBins; i++) { if (s.binHist[i] > 0 && i < startBin) startBin = i; if (s.binHist[i] > 0 && i > endBin) endBin = i; }
    a.minBin = startBin; a.maxBin = endBin;
  }
  if ((s.ratioStdU + s.ratioStdV) > 1.5 && s.binHist[1] > 0.5) { intraAdp = false; interAdp = false; }
  if (s.ratioStdU > 0.36 && s.ratioStdV > 0.2 && a.picSize > 5184000) a.chromaAdj = 0;       // (+ m_chromaWeight, an inter-picture lambda weight)
  if (interAdp) { a.cw0 = 0; a.cw1 = 1022; a.deriveSDR(&intraAdp, &interAdp); }                // LMCSAdpOption 0, SDR
  if (a.rateAdpMode == 2 && slice_qp <= 22) { intraAdp = false; interAdp = false; }
  if (!intraAdp && !interAdp) reshape = false;
  if (!reshape || !intraAdp) return VVCX_OK;          // no model, or a model only the inter pictures of the reference would use: the intra slice runs without LMCS
  if (a.rateAdpMode == 1 && slice_qp <= 22) { for (int i = 0; i < kBins; i++) a.binCW[i] = (i >= startBin && i <= endBin) ? (uint16_t) (a.initCWAnalyze + 2) : (uint16_t) 0; }
  else a.cwPerturbation(startBin, endBin, (uint16_t) a.cw1);
  a.cwReduction(startBin, endBin);
  // constructReshaperLMCS, the model half (1835-1893)
  const int bdShift = bit_depth - 10;
  if (bdShift != 0) for (int i = 0; i < kBins; i++) a.binCW[i] = (uint16_t) (a.binCW[i] * (1 << bdShift));
  a.minBin = 0; a.maxBin = kBins - 1;
  for (int i = 0; i < kBins; i++) if (a.binCW[i] > 0) { a.minBin = i; break; }
  for (int i = kBins - 1; i >= 0; i--) if (a.binCW[i] > 0) { a.maxBin = i; break; }
  a.adjustLmcsPivot();
  slice->lmcs_enable = 1; slice->lmcs_chroma_adj = a.chromaAdj; slice->lmcs_min_bin = a.minBin; slice->lmcs_max_bin = a.maxBin;
  for (int i = a.minBin; i <= a.maxBin; i++) slice->lmcs_delta_cw[i] = (int) a.binCW[i] - a.initCW;
  return VVCX_OK;
}

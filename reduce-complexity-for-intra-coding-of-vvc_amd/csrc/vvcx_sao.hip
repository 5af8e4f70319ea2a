// vvcx_sao.hip — sample adaptive offset on the bound pictures with the caller's per-CTU parameters (the filter half of SAO).
//
// What is computed is CL/SampleAdaptiveOffset.cpp SAOProcess 617-670 / offsetCTU 548-615 / offsetBlock 292-546: every sample of a CTU whose component has a type gets the
// offset of its class - band offset: one of four consecutive 1/32 bands of the sample range from the band position; edge offset: sign(c - a) + sign(c - b) over the two
// neighbours along the class direction (0 / 90 / 135 / 45 degrees), which both have to be available: inside the picture and, unless the loop filters may cross tile
// borders, inside the tile of the sample's CTU (deriveLoopFilterBoundaryAvailibility 818-883, one slice).  The reference walks a CTU with sign line buffers; here every
// sample is independent: one lane per four samples of a row reads them and the rows of their two neighbours from a copy of the deblocked picture (the filter must see unfiltered neighbours)
// and writes the picture in place.  Two launches per batch (copy, filter) over all frames and components; an HBM-bound pass: algorithmic bytes = every reconstructed
// sample read once and written once (the copy doubles the traffic).  Merge parameters are resolved on the host (vvcx_api.hip: vvcx_sao_bound_frames).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "vvcx_dev.h"

// a workgroup copies a strip of 4 rows x 1024 samples: four consecutive samples per lane and row (one word of 8-bit samples, two of 16-bit ones, when both sides are aligned)
template <typename T>
__device__ void sao_copy(const VxSaoParams &p)
{
  const int f = blockIdx.z / 3, c = blockIdx.z % 3, sh = c ? 1 : 0, pw = p.pic_w >> sh, ph = p.pic_h >> sh;
  const int x = (blockIdx.x * 256 + threadIdx.x) * 4, y0 = blockIdx.y * 4;
  if (x >= pw || (c && !p.chroma)) return;                 // (the picture is a multiple of 8 wide: the four samples are inside the plane)
  const VxFrameDev &fd = p.frames[f];
  struct alignas(4 * sizeof(T)) Quad { T v[4]; };
  for (int r = 0; r < 4 && y0 + r < ph; r++) {
    const T *s = (const T *) fd.rec[c] + (size_t) (y0 + r) * fd.stride[c] + x;
    T *d = (T *) p.tmp + p.tmp_frame * (size_t) f + p.tmp_comp[c] + (size_t) (y0 + r) * pw + x;
    if ((((uintptr_t) s | (uintptr_t) d) & (4 * sizeof(T) - 1)) == 0) *(Quad *) d = *(const Quad *) s;
    else for (int j = 0; j < 4; j++) d[j] = s[j];
  }
}

// four neighbouring samples of one row per lane (a CTU is a multiple of four wide: they share its parameters): the row itself and the rows of the two neighbours come as
// aligned groups of four samples from the dense copy, the result leaves as one group
template <typename T>
__device__ void sao_filter(const VxSaoParams &p)
{
  const int f = blockIdx.z / 3, c = blockIdx.z % 3, sh = c ? 1 : 0, pw = p.pic_w >> sh, ph = p.pic_h >> sh;
  const int x0 = (blockIdx.x * 256 + threadIdx.x) * 4, y = blockIdx.y;
  if (x0 >= pw || y >= ph || (c && !p.chroma)) return;
  const int lcs = 7 - sh, cx = x0 >> lcs, cy = y >> lcs, ctu = cy * p.ctus_w + cx;
  const VxSaoEntry e = p.table[((size_t) f * p.ctus_w * p.ctus_h + ctu) * 3 + c];
  if (e.type < 0) return;
  struct alignas(4 * sizeof(T)) Quad { T v[4]; };
  const T *src = (const T *) p.tmp + p.tmp_frame * (size_t) f + p.tmp_comp[c];
  const Quad cur = *(const Quad *) (src + (size_t) y * pw + x0);
  const int mx = (1 << p.bit_depth) - 1;
  int out[4]; bool any = false;
  if (e.type == 4) {
#pragma unroll
    for (int j = 0; j < 4; j++) {
      const int v = cur.v[j], k = ((v >> (p.bit_depth - 5)) - e.band) & 31;
      out[j] = k < 4 ? v + e.off[k] : v; any |= k < 4;
    }
  } else {
    // the two neighbours of sample x: (x + oa, y - dy) and (x - oa, y + dy); 0 degrees: left / right, 90: above / below, 135: above-left / below-right, 45: above-right / below-left
    const int dy = e.type == 0 ? 0 : 1, oa = e.type == 1 ? 0 : e.type == 3 ? 1 : -1, ay = y - dy, by = y + dy;
    int wa[12], wb[12];                                     // samples x0 - 4 .. x0 + 7 of the rows ay / by (what lies outside the plane is never used)
#pragma unroll
    for (int g = 0; g < 3; g++) {
      const int gx = x0 - 4 + 4 * g;
      const bool inx = gx >= 0 && gx < pw && (g == 1 || oa != 0);
      Quad qa = cur, qb = cur;
      if (inx && ay >= 0 && (dy || g != 1)) qa = *(const Quad *) (src + (size_t) ay * pw + gx);
      if (inx && by < ph && (dy || g != 1)) qb = *(const Quad *) (src + (size_t) by * pw + gx);
#pragma unroll
      for (int j = 0; j < 4; j++) { wa[4 * g + j] = qa.v[j]; wb[4 * g + j] = qb.v[j]; }
    }
    const uint8_t t = p.lf_across_tiles ? 0 : p.tile_of_ctu[ctu];
#pragma unroll
    for (int j = 0; j < 4; j++) {
      const int x = x0 + j, v = cur.v[j], ax = x + oa, bx = x - oa;
      out[j] = v;
      if (ax < 0 || ax >= pw || ay < 0 || bx < 0 || bx >= pw || by >= ph) continue;
      if (!p.lf_across_tiles && (p.tile_of_ctu[(ay >> lcs) * p.ctus_w + (ax >> lcs)] != t || p.tile_of_ctu[(by >> lcs) * p.ctus_w + (bx >> lcs)] != t)) continue;
      const int a = wa[4 + j + oa], b = wb[4 + j - oa];
      const int s = ((v > a) - (v < a)) + ((v > b) - (v < b));
      if (s == 0) continue;
      out[j] = v + e.off[s < 0 ? s + 2 : s + 1];            // -2, -1, 1, 2 -> full valley, half valley, half peak, full peak
      any = true;
    }
  }
  if (!any) return;
  const VxFrameDev &fd = p.frames[f];
  T *dst = (T *) fd.rec[c] + (size_t) y * fd.stride[c] + x0;
  Quad q;
#pragma unroll
  for (int j = 0; j < 4; j++) q.v[j] = (T) (out[j] < 0 ? 0 : out[j] > mx ? mx : out[j]);
  if (((uintptr_t) dst & (4 * sizeof(T) - 1)) == 0) *(Quad *) dst = q;
  else for (int j = 0; j < 4; j++) dst[j] = q.v[j];
}

// The encoder's SAO statistics (EL/EncSampleAdaptiveOffset.cpp getStatistics 284-353 / getBlkStats 1135-1549, SAOLcuBoundary 0): per CTU, component and type the number of
// samples of every class and the sum of (original - deblocked) over them - the O(samples) half of the parameter decision, whose RD half (a few hundred operations per CTU
// against the CABAC estimator) stays with the caller's encoder.  One workgroup per (CTU, component, frame): every lane classifies its samples for the five types (the same
// neighbours and band rule as the filter above) and adds them to 5 x 32 counter pairs in LDS; which samples count follows the reference's region rules: not the last 5 luma /
// 3 chroma columns and 4 / 2 rows in front of a following CTU, for the edge types not the column / row whose neighbour is outside the picture or - left and above, when
// the filters do not cross tiles - in another tile, with its treatment of the first row of the diagonal types.  HBM-bound: original and deblocked samples read once.
template <typename T>
__device__ void sao_stats(const VxSaoStatParams &p)
{
  // counters per wave (same-address LDS atomics of one wave serialise: four copies quarter the collisions); the plain class of the edge types - most samples, and the one
  // class no offset belongs to - is counted in registers and added once per lane
  __shared__ int cnt[4][5][32], dif[4][5][32];
  const int a = blockIdx.x, c = blockIdx.y, f = blockIdx.z, tid = threadIdx.x, wv = tid >> 6;
  if (c && !p.chroma) return;
  for (int i = tid; i < 4 * 5 * 32; i += 256) { (&cnt[0][0][0])[i] = 0; (&dif[0][0][0])[i] = 0; }
  __syncthreads();
  const int cx = a % p.ctus_w, cy = a / p.ctus_w, sh = c ? 1 : 0, pw = p.pic_w >> sh, ph = p.pic_h >> sh, cs = 128 >> sh, x0 = cx * cs, y0 = cy * cs;
  const int cwid = x0 + cs > pw ? pw - x0 : cs, chei = y0 + cs > ph ? ph - y0 : cs, skipR = c ? 3 : 5, skipB = c ? 2 : 4;
  const uint8_t t0 = p.tile_of_ctu[a];
  const bool across = p.lf_across_tiles != 0;
  const bool left = cx > 0 && (across || p.tile_of_ctu[a - 1] == t0), above = cy > 0 && (across || p.tile_of_ctu[a - p.ctus_w] == t0);
  const bool aboveLeft = cx > 0 && cy > 0 && (across || p.tile_of_ctu[a - p.ctus_w - 1] == t0);
  const bool right = cx + 1 < p.ctus_w, below = cy + 1 < p.ctus_h;
  const int endXr = right ? cwid - skipR : cwid, endXn = right ? cwid - skipR : cwid - 1, startXn = left ? 0 : 1;
  const int endYa = below ? chei - skipB : chei, endYn = below ? chei - skipB : chei - 1;
  const VxFrameDev &fd = p.frames[f];
  const T *rec = (const T *) fd.rec[c], *org = (const T *) fd.org[c];
  const int st = fd.stride[c];
  int pc[4] = { 0, 0, 0, 0 }, pd[4] = { 0, 0, 0, 0 };            // the plain class per edge type
#define SAO_EDGE(t_, n0_, n1_) { const int n0 = (n0_), n1 = (n1_); const int k = 2 + ((v > n0) - (v < n0)) + ((v > n1) - (v < n1)); \
                                 if (k == 2) { pc[t_]++; pd[t_] += d; } else { atomicAdd(&cnt[wv][t_][k], 1); atomicAdd(&dif[wv][t_][k], d); } }
  for (int i = tid; i < cwid * chei; i += 256) {
    const int y = i / cwid, x = i - y * cwid;
    const size_t at = (size_t) (y0 + y) * st + x0 + x;
    const int v = rec[at], d = (int) org[at] - v;
    if (x < endXr && y < endYa) { const int k = v >> (p.bit_depth - 5); atomicAdd(&cnt[wv][4][k], 1); atomicAdd(&dif[wv][4][k], d); }
    const bool xn = x >= startXn && x < endXn;
    if (y < endYa && xn) SAO_EDGE(0, (int) rec[at - 1], (int) rec[at + 1])
    if (x < endXr && y >= (above ? 0 : 1) && y < endYn) SAO_EDGE(1, (int) rec[at - st], (int) rec[at + st])
    if (y == 0 ? (x >= (aboveLeft ? 0 : 1) && x < (above ? endXn : 1)) : (xn && y < endYn)) SAO_EDGE(2, (int) rec[at - st - 1], (int) rec[at + st + 1])
    if (xn && (y == 0 ? above : y < endYn)) SAO_EDGE(3, (int) rec[at - st + 1], (int) rec[at + st - 1])
  }
#undef SAO_EDGE
  for (int t = 0; t < 4; t++) if (pc[t]) { atomicAdd(&cnt[wv][t][2], pc[t]); atomicAdd(&dif[wv][t][2], pd[t]); }
  __syncthreads();
  long long *o = p.out + (((size_t) f * p.ctus_w * p.ctus_h + a) * 3 + c) * 5 * 64;
  for (int i = tid; i < 5 * 32; i += 256) {
    const int t = i >> 5, k = i & 31;
    o[t * 64 + k] = (long long) cnt[0][t][k] + cnt[1][t][k] + cnt[2][t][k] + cnt[3][t][k];
    o[t * 64 + 32 + k] = (long long) dif[0][t][k] + dif[1][t][k] + dif[2][t][k] + dif[3][t][k];
  }
}
extern "C" __global__ void __launch_bounds__(256) vvcx_sao_stats_kernel_u8(VxSaoStatParams p) { sao_stats<uint8_t>(p); }
extern "C" __global__ void __launch_bounds__(256) vvcx_sao_stats_kernel_u16(VxSaoStatParams p) { sao_stats<uint16_t>(p); }

extern "C" __global__ void __launch_bounds__(256) vvcx_sao_copy_kernel_u8(VxSaoParams p) { sao_copy<uint8_t>(p); }
extern "C" __global__ void __launch_bounds__(256) vvcx_sao_copy_kernel_u16(VxSaoParams p) { sao_copy<uint16_t>(p); }
extern "C" __global__ void __launch_bounds__(256) vvcx_sao_kernel_u8(VxSaoParams p) { sao_filter<uint8_t>(p); }
extern "C" __global__ void __launch_bounds__(256) vvcx_sao_kernel_u16(VxSaoParams p) { sao_filter<uint16_t>(p); }

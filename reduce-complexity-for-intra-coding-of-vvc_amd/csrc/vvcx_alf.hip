// vvcx_alf.hip — adaptive loop filter on the bound pictures with the caller's parameter sets and per-CTU choices (the filter half of ALF).
//
// What is computed is CL/AdaptiveLoopFilter.cpp ALFProcess 205-383: luma samples of an enabled CTU are filtered by the 7 x 7 diamond (12 coefficient pairs with clipping,
// filterBlk 1005-1296) of their 4 x 4 block's class (deriveClassificationBlk 792-1002: activity and direction from Laplacians over the 8 x 8 window around the block) in the
// CTU's filter set, transposed by the block's direction; chroma samples by the 5 x 5 diamond of the CTU's alternative.  Four luma / two chroma rows above every lower CTU
// border lies a virtual boundary neither the window nor the taps cross.  The reference walks 32 x 32 areas with row buffers; here a workgroup owns a tile of 64 x 16
// samples of one component: it stages the tile and three samples around it (picture borders repeat the edge sample) from a copy of the unfiltered picture in LDS, four
// lanes derive the class of each of its 64 blocks (one pair of window rows each, summed across the quad), and every lane filters four neighbouring samples of one row, so
// that a row of the tile leaves as one contiguous store.  Two launches per batch (copy, filter) over all frames and components: an HBM-bound pass - algorithmic bytes = every
// sample read once and written once (the copy doubles the traffic); the 25 taps of a sample come from LDS.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "vvcx_dev.h"
#include "vvcx_alf_tables.h"

#define ALF_TW 64
#define ALF_TH 16
#define ALF_HALO 3
#define ALF_PAD 4                        // the LDS tile starts four samples left of the tile: whole aligned groups of four samples are staged
#define ALF_LW (ALF_TW + 2 * ALF_PAD)
#define ALF_LH (ALF_TH + 2 * ALF_HALO)

__device__ static const uint8_t ALF_PERM7[4][12] = { { 0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11 }, { 9, 4, 10, 8, 1, 5, 11, 7, 3, 0, 2, 6 }, { 0, 3, 2, 1, 8, 7, 6, 5, 4, 9, 10, 11 }, { 9, 8, 10, 4, 3, 7, 11, 5, 1, 0, 2, 6 } };
__device__ static const uint8_t ALF_TH_TAB[16] = { 0, 1, 2, 2, 2, 2, 2, 3, 3, 3, 3, 3, 3, 3, 3, 4 };
__device__ static const uint8_t ALF_TRANSPOSE[8] = { 0, 1, 0, 2, 2, 3, 1, 3 };

__device__ inline int alf_clampi(int v, int lo, int hi) { return v < lo ? lo : v > hi ? hi : v; }
__device__ inline int alf_min(int a, int b) { return a < b ? a : b; }
__device__ inline int alf_abs(int v) { return v < 0 ? -v : v; }
__device__ inline int alf_clip2(int clip, int ref, int a, int b) { return alf_clampi(a - ref, -clip, clip) + alf_clampi(b - ref, -clip, clip); }

// a workgroup copies a strip of 4 rows x 1024 samples: four consecutive samples per lane and row (one word of 8-bit samples, two of 16-bit ones, when both sides are aligned)
template <typename T>
__device__ void alf_copy(const VxAlfParams &p)
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

template <typename T>
__device__ void alf_filter(const VxAlfParams &p)
{
  __shared__ int16_t tile[ALF_LH][ALF_LW];
  __shared__ uint8_t cls_of[64];
  const int f = blockIdx.z / 3, c = blockIdx.z % 3, sh = c ? 1 : 0, pw = p.pic_w >> sh, ph = p.pic_h >> sh;
  const int tx0 = blockIdx.x * ALF_TW, ty0 = blockIdx.y * ALF_TH, tid = threadIdx.x;
  if (tx0 >= pw || ty0 >= ph || (c && !p.chroma)) return;                 // the same for the whole workgroup
  const int lcs = 7 - sh, ctuS = 1 << lcs, nctu = p.ctus_w * p.ctus_h;    // a tile never straddles CTUs (64 and 16 divide 128 and 64)
  const int ctuY = ty0 >> lcs;
  const VxAlfCtu u = p.ctus[(size_t) f * nctu + ctuY * p.ctus_w + (tx0 >> lcs)];
  if (!u.flag[c]) return;
  const T *src = (const T *) p.tmp + p.tmp_frame * (size_t) f + p.tmp_comp[c];
  // staged in groups of four samples (the copy's rows are dense and its planes start on multiples of 16 samples: one word of 8-bit samples, two of 16-bit ones); a group
  // left or right of the picture repeats the edge sample, a row above or below it the edge row (the reference's extendBorderPel)
  {
    struct alignas(4 * sizeof(T)) Quad { T v[4]; };
    for (int i = tid; i < ALF_LH * (ALF_LW / 4); i += 256) {
      const int ly = i / (ALF_LW / 4), g = i - ly * (ALF_LW / 4), gx = tx0 - ALF_PAD + 4 * g;
      const T *row = src + (size_t) alf_clampi(ty0 - ALF_HALO + ly, 0, ph - 1) * pw;
      int16_t *d = &tile[ly][4 * g];
      if (gx < 0) { const int16_t v = (int16_t) row[0]; d[0] = d[1] = d[2] = d[3] = v; }
      else if (gx >= pw) { const int16_t v = (int16_t) row[pw - 1]; d[0] = d[1] = d[2] = d[3] = v; }
      else { const Quad q = *(const Quad *) (row + gx); d[0] = (int16_t) q.v[0]; d[1] = (int16_t) q.v[1]; d[2] = (int16_t) q.v[2]; d[3] = (int16_t) q.v[3]; }
    }
  }
  // the virtual boundary of this CTU row; in the last one the reference passes the LUMA height for every component (ALFProcess 296, 313): only a picture of at most 128
  // rows reaches it
  const int vbPos = ctuY == p.ctus_h - 1 ? p.pic_h : ctuS - (c ? 2 : 4);
  __syncthreads();
#define TL(x_, y_) ((int) tile[(y_) - ty0 + ALF_HALO][(x_) - tx0 + ALF_PAD])
  if (c == 0) {
    // ---- classes: lane (block, r) takes the window's row pair r of its block
    const int b = tid >> 2, r = tid & 3, X = tx0 + ((b & 15) << 2), Y = ty0 + ((b >> 4) << 2), yIn = Y & (ctuS - 1);
    int sV = 0, sH = 0, sD0 = 0, sD1 = 0;
    const bool skip = (yIn == vbPos - 4 && r == 3) || (yIn == vbPos && r == 0);      // the window does not cross the boundary
    if (!skip) {
      const int y1 = Y - 2 + 2 * r, y2 = y1 + 1;
      int y0 = y1 - 1, y3 = y2 + 1;
      if (y1 > 0 && (y1 & (ctuS - 1)) == vbPos - 2) y3 = y2; else if (y1 > 0 && (y1 & (ctuS - 1)) == vbPos) y0 = y1;
#pragma unroll
      for (int k = 0; k < 4; k++) {
        const int x = X - 2 + 2 * k;
        const int a = TL(x, y1) << 1, bb = TL(x + 1, y2) << 1;
        sV += alf_abs(a - TL(x, y0) - TL(x, y2)) + alf_abs(bb - TL(x + 1, y1) - TL(x + 1, y3));
        sH += alf_abs(a - TL(x + 1, y1) - TL(x - 1, y1)) + alf_abs(bb - TL(x + 2, y2) - TL(x, y2));
        sD0 += alf_abs(a - TL(x - 1, y0) - TL(x + 1, y2)) + alf_abs(bb - TL(x, y1) - TL(x + 2, y3));
        sD1 += alf_abs(a - TL(x - 1, y2) - TL(x + 1, y0)) + alf_abs(bb - TL(x, y3) - TL(x + 2, y1));
      }
    }
    sV += __shfl_xor(sV, 1); sH += __shfl_xor(sH, 1); sD0 += __shfl_xor(sD0, 1); sD1 += __shfl_xor(sD1, 1);
    sV += __shfl_xor(sV, 2); sH += __shfl_xor(sH, 2); sD0 += __shfl_xor(sD0, 2); sD1 += __shfl_xor(sD1, 2);
    if (r == 0) {
      const int scaled = (yIn == vbPos - 4 || yIn == vbPos) ? 96 : 64;
      int cls = ALF_TH_TAB[alf_clampi(((sV + sH) * scaled) >> (p.bit_depth + 4), 0, 15)];
      int hv1, hv0, d1, d0, dirHV, dirD, hvd1, hvd0, mainDir, secDir;
      if (sV > sH) { hv1 = sV; hv0 = sH; dirHV = 1; } else { hv1 = sH; hv0 = sV; dirHV = 3; }
      if (sD0 > sD1) { d1 = sD0; d0 = sD1; dirD = 0; } else { d1 = sD1; d0 = sD0; dirD = 2; }
      if ((uint32_t) d1 * (uint32_t) hv0 > (uint32_t) hv1 * (uint32_t) d0) { hvd1 = d1; hvd0 = d0; mainDir = dirD; secDir = dirHV; }
      else { hvd1 = hv1; hvd0 = hv0; mainDir = dirHV; secDir = dirD; }
      int strength = 0;
      if (hvd1 > 2 * hvd0) strength = 1;
      if (hvd1 * 2 > 9 * hvd0) strength = 2;
      if (strength) cls += (((mainDir & 1) << 1) + strength) * 5;
      const int v = cls | (ALF_TRANSPOSE[mainDir * 2 + (secDir >> 1)] << 5);
      cls_of[b] = (uint8_t) v;
      if (p.classes && X < pw && Y < ph) p.classes[(size_t) f * (p.pic_w >> 2) * (p.pic_h >> 2) + (size_t) (Y >> 2) * (p.pic_w >> 2) + (X >> 2)] = (uint8_t) v;
    }
    __syncthreads();
  }
  // ---- the filter: lane -> (row of the tile, group of four samples)
  const int row = tid >> 4, grp = tid & 15, X = tx0 + (grp << 2), Y = ty0 + row;
  if (X >= pw || Y >= ph) return;
  const int yVb = Y & (ctuS - 1), reach = c ? 2 : 4;
  int d1 = 1, d2 = 2, d3 = 3;
  if (yVb < vbPos && yVb >= vbPos - reach) { const int room = vbPos - 1 - yVb; d1 = alf_min(d1, room); d2 = alf_min(d2, room); d3 = alf_min(d3, room); }
  else if (yVb >= vbPos && yVb <= vbPos + reach - 1) { const int room = yVb - vbPos; d1 = alf_min(d1, room); d2 = alf_min(d2, room); d3 = alf_min(d3, room); }
  const VxAlfFrame &tb = p.tabs[f];
  const int maxv = (1 << p.bit_depth) - 1;
  int out[4];
  if (c == 0) {
    const int v = cls_of[((row >> 2) << 4) + grp], cls = v & 31, tr = v >> 5, set = u.set;
    int co[12], cl[12];
#pragma unroll
    for (int i = 0; i < 12; i++) {
      const int k = ALF_PERM7[tr][i];
      if (set < 16) { co[i] = VX_ALF_FIXED[VX_ALF_CLASS_TO_FIXED[set][cls]][k]; cl[i] = 1 << p.bit_depth; }
      else { co[i] = tb.luma_coeff[set - 16][cls][k]; cl[i] = tb.luma_clip[set - 16][cls][k]; }
    }
#pragma unroll
    for (int j = 0; j < 4; j++) {
      const int x = X + j, cur = TL(x, Y);
      int sum = co[0] * alf_clip2(cl[0], cur, TL(x, Y + d3), TL(x, Y - d3));
      sum += co[1] * alf_clip2(cl[1], cur, TL(x + 1, Y + d2), TL(x - 1, Y - d2));
      sum += co[2] * alf_clip2(cl[2], cur, TL(x, Y + d2), TL(x, Y - d2));
      sum += co[3] * alf_clip2(cl[3], cur, TL(x - 1, Y + d2), TL(x + 1, Y - d2));
      sum += co[4] * alf_clip2(cl[4], cur, TL(x + 2, Y + d1), TL(x - 2, Y - d1));
      sum += co[5] * alf_clip2(cl[5], cur, TL(x + 1, Y + d1), TL(x - 1, Y - d1));
      sum += co[6] * alf_clip2(cl[6], cur, TL(x, Y + d1), TL(x, Y - d1));
      sum += co[7] * alf_clip2(cl[7], cur, TL(x - 1, Y + d1), TL(x + 1, Y - d1));
      sum += co[8] * alf_clip2(cl[8], cur, TL(x - 2, Y + d1), TL(x + 2, Y - d1));
      sum += co[9] * alf_clip2(cl[9], cur, TL(x + 3, Y), TL(x - 3, Y));
      sum += co[10] * alf_clip2(cl[10], cur, TL(x + 2, Y), TL(x - 2, Y));
      sum += co[11] * alf_clip2(cl[11], cur, TL(x + 1, Y), TL(x - 1, Y));
      out[j] = alf_clampi(((sum + 64) >> 7) + cur, 0, maxv);
    }
  } else {
    const int t = u.alt[c - 1];
    int co[6], cl[6];
#pragma unroll
    for (int i = 0; i < 6; i++) { co[i] = tb.chroma_coeff[t][i]; cl[i] = tb.chroma_clip[t][i]; }
#pragma unroll
    for (int j = 0; j < 4; j++) {
      const int x = X + j, cur = TL(x, Y);
      int sum = co[0] * alf_clip2(cl[0], cur, TL(x, Y + d2), TL(x, Y - d2));
      sum += co[1] * alf_clip2(cl[1], cur, TL(x + 1, Y + d1), TL(x - 1, Y - d1));
      sum += co[2] * alf_clip2(cl[2], cur, TL(x, Y + d1), TL(x, Y - d1));
      sum += co[3] * alf_clip2(cl[3], cur, TL(x - 1, Y + d1), TL(x + 1, Y - d1));
      sum += co[4] * alf_clip2(cl[4], cur, TL(x + 2, Y), TL(x - 2, Y));
      sum += co[5] * alf_clip2(cl[5], cur, TL(x + 1, Y), TL(x - 1, Y));
      out[j] = alf_clampi(((sum + 64) >> 7) + cur, 0, maxv);
    }
  }
#undef TL
  const VxFrameDev &fd = p.frames[f];
  T *dst = (T *) fd.rec[c] + (size_t) Y * fd.stride[c] + X;      // the picture is a multiple of 8 wide: the four samples are inside it
  struct alignas(4 * sizeof(T)) QuadO { T v[4]; };
  if (((uintptr_t) dst & (4 * sizeof(T) - 1)) == 0) { QuadO q; q.v[0] = (T) out[0]; q.v[1] = (T) out[1]; q.v[2] = (T) out[2]; q.v[3] = (T) out[3]; *(QuadO *) dst = q; }
  else {
#pragma unroll
    for (int j = 0; j < 4; j++) dst[j] = (T) out[j];
  }
}

extern "C" __global__ void __launch_bounds__(256) vvcx_alf_copy_kernel_u8(VxAlfParams p) { alf_copy<uint8_t>(p); }
extern "C" __global__ void __launch_bounds__(256) vvcx_alf_copy_kernel_u16(VxAlfParams p) { alf_copy<uint16_t>(p); }
extern "C" __global__ void __launch_bounds__(256) vvcx_alf_kernel_u8(VxAlfParams p) { alf_filter<uint8_t>(p); }
extern "C" __global__ void __launch_bounds__(256) vvcx_alf_kernel_u16(VxAlfParams p) { alf_filter<uint16_t>(p); }

// vvcx_sao.hip — sample adaptive offset on the bound pictures with the caller's per-CTU parameters (the filter half of SAO).
//
// What is computed is CL/SampleAdaptiveOffset.cpp SAOProcess 617-670 / offsetCTU 548-615 / offsetBlock 292-546: every sample of a CTU whose component has a type gets the
// offset of its class - band offset: one of four consecutive 1/32 bands of the sample range from the band position; edge offset: sign(c - a) + sign(c - b) over the two
// neighbours along the class direction (0 / 90 / 135 / 45 degrees), which both have to be available: inside the picture and, unless the loop filters may cross tile
// borders, inside the tile of the sample's CTU (deriveLoopFilterBoundaryAvailibility 818-883, one slice).  The reference walks a CTU with sign line buffers; here every
// sample is independent: one thread per sample reads its value and its two neighbours from a copy of the deblocked picture (the filter must see unfiltered neighbours)
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

template <typename T>
__device__ void sao_filter(const VxSaoParams &p)
{
  const int f = blockIdx.z / 3, c = blockIdx.z % 3, sh = c ? 1 : 0, pw = p.pic_w >> sh, ph = p.pic_h >> sh;
  const int x = blockIdx.x * 256 + threadIdx.x, y = blockIdx.y;
  if (x >= pw || y >= ph || (c && !p.chroma)) return;
  const int lcs = 7 - sh, cx = x >> lcs, cy = y >> lcs, ctu = cy * p.ctus_w + cx;
  const VxSaoEntry e = p.table[((size_t) f * p.ctus_w * p.ctus_h + ctu) * 3 + c];
  if (e.type < 0) return;
  const T *src = (const T *) p.tmp + p.tmp_frame * (size_t) f + p.tmp_comp[c];
  const int v = src[(size_t) y * pw + x];
  int r;
  if (e.type == 4) {
    const int k = ((v >> (p.bit_depth - 5)) - e.band) & 31;
    if (k >= 4) return;
    r = v + e.off[k];
  } else {
    const int dx = e.type == 1 ? 0 : 1, dy = e.type == 0 ? 0 : 1;
    const int ax = e.type == 3 ? x + 1 : x - dx, ay = y - dy, bx = e.type == 3 ? x - 1 : x + dx, by = y + dy;      // 45 degrees: above-right and below-left
    if (ax < 0 || ax >= pw || ay < 0 || ay >= ph || bx < 0 || bx >= pw || by < 0 || by >= ph) return;
    if (!p.lf_across_tiles) {
      const uint8_t t = p.tile_of_ctu[ctu];
      if (p.tile_of_ctu[(ay >> lcs) * p.ctus_w + (ax >> lcs)] != t || p.tile_of_ctu[(by >> lcs) * p.ctus_w + (bx >> lcs)] != t) return;
    }
    const int a = src[(size_t) ay * pw + ax], b = src[(size_t) by * pw + bx];
    const int s = ((v > a) - (v < a)) + ((v > b) - (v < b));
    if (s == 0) return;
    r = v + e.off[s < 0 ? s + 2 : s + 1];                 // -2, -1, 1, 2 -> full valley, half valley, half peak, full peak
  }
  const int mx = (1 << p.bit_depth) - 1;
  const VxFrameDev &fd = p.frames[f];
  ((T *) fd.rec[c])[(size_t) y * fd.stride[c] + x] = (T) (r < 0 ? 0 : r > mx ? mx : r);
}

extern "C" __global__ void __launch_bounds__(256) vvcx_sao_copy_kernel_u8(VxSaoParams p) { sao_copy<uint8_t>(p); }
extern "C" __global__ void __launch_bounds__(256) vvcx_sao_copy_kernel_u16(VxSaoParams p) { sao_copy<uint16_t>(p); }
extern "C" __global__ void __launch_bounds__(256) vvcx_sao_kernel_u8(VxSaoParams p) { sao_filter<uint8_t>(p); }
extern "C" __global__ void __launch_bounds__(256) vvcx_sao_kernel_u16(VxSaoParams p) { sao_filter<uint16_t>(p); }

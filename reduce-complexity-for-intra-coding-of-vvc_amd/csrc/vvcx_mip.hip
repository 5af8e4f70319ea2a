// vvcx_mip.hip — matrix-based intra prediction as device functions and a leaf kernel (gfx950).
//
// The device functions live in vvcx_mip_dev.h (shared with the search kernel); vvcx_mip_pred_batch exposes them for the parity test against
// the reference's vectors.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "vvcx_dev.h"
#include "vvcx_mip_dev.h"

struct VxMipCase { int32_t w, h, mode, bit_depth, ref_off, pred_off; };
// leaf kernel: one 64-thread workgroup per case; refs = top[w] | left[h] per case
extern "C" __global__ void __launch_bounds__(64) vvcx_leaf_mip_kernel(const VxMipCase *cases, const int16_t *refs, int16_t *preds)
{
  __shared__ int sh[16], red[64];
  const VxMipCase c = cases[blockIdx.x];
  const int lane = threadIdx.x;
  const MipGeo g = mip_geo(c.w, c.h);
  const int16_t *top = refs + c.ref_off, *left = top + c.w;
  mip_reduced_pred(top, left, g, c.mode, c.bit_depth, lane, sh, red);
  __syncthreads();
  for (int i = lane; i < c.w * c.h; i += 64) { const int py = i / c.w, px = i - py * c.w; preds[c.pred_off + i] = (int16_t) mip_sample(red, top, left, g, px, py); }
}

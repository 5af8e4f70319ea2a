// vvcx_lfnst_dev.h — low-frequency non-separable transform (LFNST) of one block, one wave per block.  Included by vvcx_kernel.hip.
//
// Reference: TrQuant::xFwdLfnst / xInvLfnst / fwdLfnstNxN / invLfnstNxN (CL/TrQuant.cpp:241-560), the kernels of CL/RomLFNST.cpp as data
// (vvcx_lfnst_tables.h, generated from the compiled reference).  A block of at least 4x4 samples of a CU with lfnstIdx 1 / 2 keeps only the
// top-left 4x4 (one side 4) or 8x8 (both sides >= 8) of its primary coefficients (xT 855-868); the 16 / 48 of them (the 8x8 without its
// bottom-right 4x4) are multiplied by a 16 x 16 / 16 x 48 int8 matrix picked by the intra mode (4 sets x 2 kernels, transposed input for modes
// beyond the diagonal); 8 outputs for 4x4 / 8x8 blocks, else 16; they land on the first scan positions of the block.
#pragma once
#include "vvcx_lfnst_tables.h"

// PU::getWideAngIntraMode (CL/UnitTools.cpp:963-989) + TrQuant::getLFNSTIntraMode (293-311) + getTransposeFlag (312-317): kernel set in bits 0-1, transposition in bit 2
__device__ inline int lfnst_mode(int dir, int w, int h)
{
  int pm = dir;
  if (dir >= 2) {
    const int lw = ilog2i(w), lh = ilog2i(h), ds = lw > lh ? lw - lh : lh - lw;
    const int shift = ds == 0 ? 0 : ds == 1 ? 6 : ds == 2 ? 10 : ds == 3 ? 12 : ds == 4 ? 14 : 15;
    if (w > h && dir < 2 + shift) pm += 65;
    else if (h > w && pm > 66 - shift) pm -= 67;
  }
  const int ext = pm < 0 ? pm + 14 + 67 : pm >= 67 ? pm + 14 : pm;
  const int transpose = (ext >= 67 + 14) || (ext < 67 && ext > 34);
  return (int) VX_LFNST_LUT[ext] | (transpose ? 4 : 0);
}
// k-th position of the diagonal scan of the top-left region (4x4 groups (0,0), (0,1), (1,0), (1,1) as x, y) -> x | y << 4
__device__ inline int lfnst_scan_xy(int k)
{
  const int grp = k >> 4, in = L.t.cg_scan[k & 15];                  // cg_scan offset 0: the 4x4 group
  const int gx = grp == 2 || grp == 3, gy = grp == 1 || grp == 3;
  return (gx * 4 + (in & 15)) | ((gy * 4 + (in >> 4)) << 4);
}
// (x, y) of the region sample that feeds input i of the kernel (i < 16 or 48)
__device__ inline int lfnst_vec_xy(int i, int sb, int transpose)
{
  int x, y;
  if (sb == 4) { y = i >> 2; x = i & 3; }
  else if (i < 32) { y = i >> 3; x = i & 7; }
  else { y = 4 + ((i - 32) >> 2); x = (i - 32) & 3; }
  if (transpose) { const int t = x; x = y; y = t; }
  return x | (y << 4);
}
// forward: coef (int16, row stride `stride`, the block's primary coefficients with at least the top-left sb x sb computed) -> LFNST coefficients on
// the region's first scan positions, the rest of the 48 (16) region positions zero.  scr: 64 ints of wave scratch.
__device__ inline void wave_lfnst_fwd(int16_t *coef, int stride, int w, int h, int mode, int lfnst_idx, int32_t *scr, int lane)
{
  const int sb = (w >= 8 && h >= 8) ? 8 : 4, trSize = sb == 8 ? 48 : 16;
  const int nOut = ((w == 4 && h == 4) || (w == 8 && h == 8)) ? 8 : 16;
  const int8_t *M = sb == 8 ? VX_LFNST_8x8 + (((mode & 3) * 2 + lfnst_idx - 1) * 16) * 48 : VX_LFNST_4x4 + (((mode & 3) * 2 + lfnst_idx - 1) * 16) * 16;
  if (lane < trSize) { const int xy = lfnst_vec_xy(lane, sb, (mode >> 2) & 1); scr[lane] = coef[(xy >> 4) * stride + (xy & 15)]; }
  wave_sync();
  int out = 0;
  if (lane < nOut) { int c = 0; for (int i = 0; i < trSize; i++) c += scr[i] * M[lane * trSize + i]; out = (c + 64) >> 7; }
  wave_sync();
  if (lane < trSize) { const int xy = lfnst_scan_xy(lane); coef[(xy >> 4) * stride + (xy & 15)] = (int16_t) out; }
  wave_sync();
}
// inverse: the first 16 scan positions of deq (int16, row stride `stride`) -> the 48 (16) primary coefficients of the region, clipped to 16 bits
__device__ inline void wave_lfnst_inv(int16_t *deq, int stride, int w, int h, int mode, int lfnst_idx, int32_t *scr, int lane)
{
  const int sb = (w >= 8 && h >= 8) ? 8 : 4, trSize = sb == 8 ? 48 : 16;
  const int nIn = ((w == 4 && h == 4) || (w == 8 && h == 8)) ? 8 : 16;
  const int8_t *M = sb == 8 ? VX_LFNST_8x8 + (((mode & 3) * 2 + lfnst_idx - 1) * 16) * 48 : VX_LFNST_4x4 + (((mode & 3) * 2 + lfnst_idx - 1) * 16) * 16;
  if (lane < 16) { const int xy = lfnst_scan_xy(lane); scr[lane] = deq[(xy >> 4) * stride + (xy & 15)]; }
  wave_sync();
  int r = 0;
  if (lane < trSize) { for (int i = 0; i < nIn; i++) r += scr[i] * M[i * trSize + lane]; r = (r + 64) >> 7; r = r < -32768 ? -32768 : r > 32767 ? 32767 : r; }
  wave_sync();
  if (lane < trSize) { const int xy = lfnst_vec_xy(lane, sb, (mode >> 2) & 1); deq[(xy >> 4) * stride + (xy & 15)] = (int16_t) r; }
  wave_sync();
}

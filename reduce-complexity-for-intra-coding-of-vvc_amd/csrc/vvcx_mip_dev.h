// vvcx_mip_dev.h — matrix-based intra prediction as device functions (gfx950), shared by the search kernel (MIP candidates of the SATD and RD
// stages) and the leaf kernel of vvcx_mip.hip.
//
// ≙ MatrixIntraPrediction::prepareInputForPred + predBlock (CL/MatrixIntraPrediction.cpp, JVET_O0925 form).  One wavefront per block:
// the reduced boundary and the (at most 64) outputs of the matrix stage one per lane, then every sample of the block in closed form — the
// two linear up-sampling passes of predictionUpsampling (448-560) with their intermediate rounding folded into one expression per sample,
// so no intermediate picture is written.  The reduced prediction is kept in block orientation; the transposed modes only change how the
// matrix stage reads its input and where it puts its outputs.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "vvcx_mip_tables.h"

__device__ inline int mip_log2(int v) { return 31 - __clz(v); }
__device__ inline int mip_num_modes(int w, int h) { return (w > 4 * h || h > 4 * w) ? 0 : (w == 4 && h == 4) ? 35 : (w <= 8 && h <= 8) ? 19 : 11; }

struct MipGeo { int w, h, rb, rpw, rph, upH, upV, small; };
__device__ inline MipGeo mip_geo(int w, int h)
{
  MipGeo g; g.w = w; g.h = h; g.rb = (w > 4 || h > 4) ? 4 : 2; g.small = w <= 8 && h <= 8;
  g.rpw = g.small ? 4 : (w < 8 ? w : 8); g.rph = g.small ? 4 : (h < 8 ? h : 8); g.upH = w / g.rpw; g.upV = h / g.rph;
  return g;
}
// matrix stage: red[0 .. rpw*rph) in block orientation (red[y * rpw + x]); sh = the wave's scratch (>= 16 ints); all 64 lanes call
__device__ static void mip_reduced_pred(const int16_t *top, const int16_t *left, const MipGeo &g, int mode, int bd, int lane, int *sh, int *red)
{
  const int rb = g.rb, inSize = 2 * rb;
  // boundary averaging (prepareInputForPred 88-99): lanes 0 .. 2*rb-1, top first
  if (lane < inSize) {
    const int side = lane >= rb, k = side ? lane - rb : lane, len = side ? g.h : g.w, f = len / rb;
    const int16_t *src = side ? left : top;
    int v;
    if (f <= 1) v = src[k];
    else { int s = 0; for (int i = 0; i < f; i++) s += src[k * f + i]; v = (s + (f >> 1)) >> mip_log2(f); }
    sh[lane] = v;
  }
  __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup"); __builtin_amdgcn_wave_barrier();
  const int numModes = mip_num_modes(g.w, g.h), transpose = mode > numModes / 2, idx = transpose ? mode - numModes / 2 : mode;
  // input vector: [top | left], or [left | top] for the transposed modes; rebased on its first element (108-126)
  int in[8];
  for (int i = 0; i < inSize; i++) in[i] = transpose ? sh[i < rb ? rb + i : i - rb] : sh[i];
  const int inOff = in[0];
  for (int i = 1; i < inSize; i++) in[i] -= inOff;
  in[0] = g.small ? inOff - (1 << (bd - 1)) : 0;
  const uint8_t *matrix; int shiftM, offsetM, cols;
  if (g.w == 4 && g.h == 4) { matrix = VX_MIP_MATRIX_4x4 + idx * 16 * 4; shiftM = VX_MIP_SHIFT_4x4[idx]; offsetM = VX_MIP_OFFSET_4x4[idx]; cols = 4; }
  else if (g.small) { matrix = VX_MIP_MATRIX_8x8 + idx * 16 * 8; shiftM = VX_MIP_SHIFT_8x8[idx]; offsetM = VX_MIP_OFFSET_8x8[idx]; cols = 8; }
  else { matrix = VX_MIP_MATRIX_16x16 + idx * 64 * 7; shiftM = VX_MIP_SHIFT_16x16[idx]; offsetM = VX_MIP_OFFSET_16x16[idx]; cols = 7; }
  int leaveHor = g.w == 4 && g.h >= 16, leaveVer = g.h == 4 && g.w >= 16;
  if (transpose) { const int t = leaveHor; leaveHor = leaveVer; leaveVer = t; }
  const int iw = transpose ? g.rph : g.rpw, ih = transpose ? g.rpw : g.rph;            // the matrix stage's own output grid
  if (lane < iw * ih) {
    const int y = lane / iw, x = lane - y * iw;
    // row of the weight matrix for output (x, y): computeReducedPred's pointer walk (667-728) in closed form
    const int xStep = leaveHor ? 2 : 1, redSize = g.small ? 0 : 1;
    const int rowsPerLine = xStep * iw + (leaveVer ? iw : 0);        // matrix rows consumed per output line
    const int row = y * rowsPerLine + x * xStep;             // (the reference's pointer starts xStep - 1 ahead and steps back xStep before each use)
    const uint8_t *wgt = matrix + row * cols;
    int sum = 0; for (int i = 0; i < inSize; i++) sum += in[i];
    int acc = 0;
    if (redSize) for (int i = 1; i < inSize; i++) acc += in[i] * wgt[i - 1];
    else for (int i = 0; i < inSize; i++) acc += in[i] * wgt[i];
    int v = ((acc + (1 << (shiftM - 1)) - offsetM * sum) >> shiftM) + inOff;
    const int mx = (1 << bd) - 1; v = v < 0 ? 0 : v > mx ? mx : v;
    const int bx = transpose ? y : x, by = transpose ? x : y;        // block orientation
    red[by * g.rpw + bx] = v;
  }
  __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup"); __builtin_amdgcn_wave_barrier();
}
// one sample of the block from the reduced prediction: both up-sampling passes (shorter side first) with their intermediate rounding
__device__ inline int mip_interp(int before, int behind, int pos, int f) { return f == 1 ? behind : (((f - pos) * before + pos * behind + (f >> 1)) >> mip_log2(f)); }
__device__ static int mip_sample(const int *red, const int16_t *top, const int16_t *left, const MipGeo &g, int px, int py)
{
  const int xr = px / g.upH, posx = px - xr * g.upH + 1, yr = py / g.upV, posy = py - yr * g.upV + 1;
  if (g.h > g.w) {
    // horizontal pass on the rows y_r = (yr + 1) * upV - 1 of the reduced lines, then vertical
    auto hline = [&](int r) -> int {                               // value at (px, row of reduced line r); r = -1: the top reference
      if (r < 0) return top[px];
      const int before = xr == 0 ? left[(r + 1) * g.upV - 1] : red[r * g.rpw + xr - 1];
      return mip_interp(before, red[r * g.rpw + xr], posx, g.upH);
    };
    return mip_interp(hline(yr - 1), hline(yr), posy, g.upV);
  }
  auto vline = [&](int c) -> int {                                 // value at (column of reduced column c, py); c = -1: the left reference
    if (c < 0) return left[py];
    const int before = yr == 0 ? top[(c + 1) * g.upH - 1] : red[(yr - 1) * g.rpw + c];
    return mip_interp(before, red[yr * g.rpw + c], posy, g.upV);
  };
  return mip_interp(vline(xr - 1), vline(xr), posx, g.upH);
}


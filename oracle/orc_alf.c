/*
 * orc_alf.c — CPU restatement of the adaptive loop FILTER with the parameter sets and per-CTU choices given: what the decoder side of
 * AdaptiveLoopFilter does to a picture that has been deblocked and offset by SAO.  TEST INFRASTRUCTURE ONLY (see vvc_oracle.h).
 *
 * Follows CL/AdaptiveLoopFilter.cpp: ALFProcess 205-383 (per CTU: luma classified and filtered with the CTU's filter set - sets 0..15 are the fixed ones, 16 + k the
 * k-th parameter set of the slice - chroma with the CTU's alternative; source = a copy of the picture whose borders repeat the edge samples, extendBorderPel), the per-class
 * tables of a parameter set (reconstructCoeff 420-608 in its JVET_O0669 form: class -> filter through filterCoeffDeltaIdx, clipping index -> value through the table of
 * create() 633-654), deriveClassificationBlk 792-1002 and filterBlk 1005-1296 with the virtual boundary four luma / two chroma rows above the lower CTU border (in the
 * last CTU row the reference passes the luma picture height as the boundary position for every component, which only a picture of at most 128 rows reaches: its lower edge
 * then acts as a boundary).  No PPS virtual boundaries, no PCM (JVET_O0525), tiles are not looked at (the reference's ALF of this version does not either).
 *
 * The reference walks 32 x 32 areas with row buffers of half-resolution Laplacians; per 4 x 4 block that is: sums over the 8 x 8 window around the block (rows and columns
 * -2 .. +5) of the Laplacians at the samples (even, even) and (odd, odd) of every 2 x 2 cell, in the directions vertical, horizontal and the two diagonals; at a virtual
 * boundary the window loses the two rows across it, the row next to it takes its missing neighbour row from itself, and the activity is scaled by 96 instead of 64.
 * Pinned against the reference's AdaptiveLoopFilter (oracle/_ref, tests/golden/alf.npz).
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include "orc_internal.h"
#include "orc_alf_tables.h"

static int clampi(int v, int lo, int hi) { return v < lo ? lo : v > hi ? hi : v; }

/* the clipping values of create() 633-654: luma round(2^(bd (4 - i) / 4)), chroma 2^bd, then round(2^(bd - 8 + 8 (2 - (i - 1)) / 3)) */
int orc_alf_clip_value(int chroma, int bit_depth, int idx)
{
  if (!chroma) return (int) round(pow(2., (double) (bit_depth * (4 - idx)) / 4));
  if (idx == 0) return 1 << bit_depth;
  return (int) round(pow(2., bit_depth - 8 + 8. * (4 - idx - 1) / 3));
}

/* reconstructCoeff: a parameter set's per-class luma tables [25][13] (coefficient 12 is the centre weight 128, never used by the filter) and chroma alternatives [n][7] */
void orc_alf_reconstruct(const orc_alf_aps *a, int bit_depth, int16_t *luma_coeff, int16_t *luma_clip, int16_t *chroma_coeff, int16_t *chroma_clip)
{
  for (int c = 0; c < 25; c++) {
    const int f = a->class_to_filter[c];
    for (int i = 0; i < 12; i++) {
      luma_coeff[c * 13 + i] = a->luma_coeff[f][i];
      luma_clip[c * 13 + i] = (int16_t) orc_alf_clip_value(0, bit_depth, a->nonlinear_luma ? a->luma_clip_idx[f][i] : 0);
    }
    luma_coeff[c * 13 + 12] = 128; luma_clip[c * 13 + 12] = (int16_t) orc_alf_clip_value(0, bit_depth, 0);
  }
  for (int t = 0; t < a->num_chroma_alt; t++) {
    for (int i = 0; i < 6; i++) {
      chroma_coeff[t * 7 + i] = a->chroma_coeff[t][i];
      chroma_clip[t * 7 + i] = (int16_t) orc_alf_clip_value(1, bit_depth, a->nonlinear_chroma[t] ? a->chroma_clip_idx[t][i] : 0);
    }
    chroma_coeff[t * 7 + 6] = 128; chroma_clip[t * 7 + 6] = (int16_t) orc_alf_clip_value(1, bit_depth, 0);
  }
}

typedef struct { const int16_t *p; int w, h; } plane_t;
static int px(const plane_t *s, int x, int y) { return s->p[(size_t) clampi(y, 0, s->h - 1) * s->w + clampi(x, 0, s->w - 1)]; }

/* class (0..24) and transpose index (0..3) of the 4 x 4 luma block at (X, Y); vb: the block's CTU has a virtual boundary at row vbPos of the CTU (CTU height ctuH) */
static void classify(const plane_t *s, int X, int Y, int bit_depth, int ctuH, int vbPos, int *cls, int *tr)
{
  static const int th[16] = { 0, 1, 2, 2, 2, 2, 2, 3, 3, 3, 3, 3, 3, 3, 3, 4 };
  const int yIn = Y & (ctuH - 1);
  int sumV = 0, sumH = 0, sumD0 = 0, sumD1 = 0;
  for (int r = 0; r < 4; r++) {                             /* the four pairs of rows of the window: Y - 2 + 2 r and the one below */
    if (yIn == vbPos - 4 && r == 3) continue;               /* the block above the boundary: without the rows below it */
    if (yIn == vbPos && r == 0) continue;                   /* the block below: without the rows above */
    const int y1 = Y - 2 + 2 * r, y2 = y1 + 1;
    int y0 = y1 - 1, y3 = y2 + 1;
    if (y1 > 0 && (y1 & (ctuH - 1)) == vbPos - 2) y3 = y2;  /* the pair just above the boundary: the row across it is replaced by the pair's lower row */
    else if (y1 > 0 && (y1 & (ctuH - 1)) == vbPos) y0 = y1; /* the pair just below: by its upper row */
    for (int c = 0; c < 4; c++) {
      const int x = X - 2 + 2 * c;
      const int a = px(s, x, y1) << 1, b = px(s, x + 1, y2) << 1;      /* the samples (even, even) and (odd, odd) of the cell */
      sumV += abs(a - px(s, x, y0) - px(s, x, y2)) + abs(b - px(s, x + 1, y1) - px(s, x + 1, y3));
      sumH += abs(a - px(s, x + 1, y1) - px(s, x - 1, y1)) + abs(b - px(s, x + 2, y2) - px(s, x, y2));
      sumD0 += abs(a - px(s, x - 1, y0) - px(s, x + 1, y2)) + abs(b - px(s, x, y1) - px(s, x + 2, y3));
      sumD1 += abs(a - px(s, x - 1, y2) - px(s, x + 1, y0)) + abs(b - px(s, x, y3) - px(s, x + 2, y1));
    }
  }
  const int shift = bit_depth + 4, scaled = (yIn == vbPos - 4 || yIn == vbPos) ? 96 : 64;
  int classIdx = th[clampi(((sumV + sumH) * scaled) >> shift, 0, 15)];
  int hv1, hv0, d1, d0, dirHV, dirD, hvd1, hvd0, mainDir, secDir;
  if (sumV > sumH) { hv1 = sumV; hv0 = sumH; dirHV = 1; } else { hv1 = sumH; hv0 = sumV; dirHV = 3; }
  if (sumD0 > sumD1) { d1 = sumD0; d0 = sumD1; dirD = 0; } else { d1 = sumD1; d0 = sumD0; dirD = 2; }
  if ((uint32_t) d1 * (uint32_t) hv0 > (uint32_t) hv1 * (uint32_t) d0) { hvd1 = d1; hvd0 = d0; mainDir = dirD; secDir = dirHV; }
  else { hvd1 = hv1; hvd0 = hv0; mainDir = dirHV; secDir = dirD; }
  int strength = 0;
  if (hvd1 > 2 * hvd0) strength = 1;
  if (hvd1 * 2 > 9 * hvd0) strength = 2;
  if (strength) classIdx += (((mainDir & 1) << 1) + strength) * 5;
  static const int transposeTable[8] = { 0, 1, 0, 2, 2, 3, 1, 3 };
  *cls = classIdx; *tr = transposeTable[mainDir * 2 + (secDir >> 1)];
}

static int clip2(int clip, int ref, int v0, int v1) { return clampi(v0 - ref, -clip, clip) + clampi(v1 - ref, -clip, clip); }

/* one sample of the 7 x 7 (luma) / 5 x 5 (chroma) diamond; co / cl in the order of the block's transpose; yVb: the sample's row inside its CTU */
static int filter_sample(const plane_t *s, int x, int y, int chroma, const int *co, const int *cl, int yVb, int vbPos, int maxv)
{
  int d1 = 1, d2 = 2, d3 = 3;                               /* row distances of the three tap rows on either side (the same above and below) */
  const int reach = chroma ? 2 : 4;
  if (yVb < vbPos && yVb >= vbPos - reach) { const int room = vbPos - 1 - yVb; d1 = d1 < room ? d1 : room; d2 = d2 < room ? d2 : room; d3 = d3 < room ? d3 : room; }
  else if (yVb >= vbPos && yVb <= vbPos + reach - 1) { const int room = yVb - vbPos; d1 = d1 < room ? d1 : room; d2 = d2 < room ? d2 : room; d3 = d3 < room ? d3 : room; }
  const int cur = px(s, x, y);
  int sum = 0;
  if (!chroma) {
    sum += co[0] * clip2(cl[0], cur, px(s, x, y + d3), px(s, x, y - d3));
    sum += co[1] * clip2(cl[1], cur, px(s, x + 1, y + d2), px(s, x - 1, y - d2));
    sum += co[2] * clip2(cl[2], cur, px(s, x, y + d2), px(s, x, y - d2));
    sum += co[3] * clip2(cl[3], cur, px(s, x - 1, y + d2), px(s, x + 1, y - d2));
    sum += co[4] * clip2(cl[4], cur, px(s, x + 2, y + d1), px(s, x - 2, y - d1));
    sum += co[5] * clip2(cl[5], cur, px(s, x + 1, y + d1), px(s, x - 1, y - d1));
    sum += co[6] * clip2(cl[6], cur, px(s, x, y + d1), px(s, x, y - d1));
    sum += co[7] * clip2(cl[7], cur, px(s, x - 1, y + d1), px(s, x + 1, y - d1));
    sum += co[8] * clip2(cl[8], cur, px(s, x - 2, y + d1), px(s, x + 2, y - d1));
    sum += co[9] * clip2(cl[9], cur, px(s, x + 3, y), px(s, x - 3, y));
    sum += co[10] * clip2(cl[10], cur, px(s, x + 2, y), px(s, x - 2, y));
    sum += co[11] * clip2(cl[11], cur, px(s, x + 1, y), px(s, x - 1, y));
  } else {
    sum += co[0] * clip2(cl[0], cur, px(s, x, y + d2), px(s, x, y - d2));
    sum += co[1] * clip2(cl[1], cur, px(s, x + 1, y + d1), px(s, x - 1, y - d1));
    sum += co[2] * clip2(cl[2], cur, px(s, x, y + d1), px(s, x, y - d1));
    sum += co[3] * clip2(cl[3], cur, px(s, x - 1, y + d1), px(s, x + 1, y - d1));
    sum += co[4] * clip2(cl[4], cur, px(s, x + 2, y), px(s, x - 2, y));
    sum += co[5] * clip2(cl[5], cur, px(s, x + 1, y), px(s, x - 1, y));
  }
  return clampi(((sum + 64) >> 7) + cur, 0, maxv);
}

/* The picture filtered in place.  sets: n_sets luma tables [25][13] of coefficients / clipping values (filter set 16 + k), alts: n_alt chroma tables [7];
 * ctu: per CTU (raster) {enable Y, Cb, Cr, luma filter set, alternative Cb, Cr}.  cls_out (may be NULL): class | transpose << 5 per luma 4 x 4 block of the enabled CTUs
 * (255 elsewhere), (w / 4) per row.  Returns 0, -1 for bad arguments. */
int orc_alf_picture(int w, int h, int bit_depth, int n_sets, const int16_t *luma_coeff, const int16_t *luma_clip, int n_alt, const int16_t *chroma_coeff, const int16_t *chroma_clip,
                    const orc_alf_ctu *ctu, int16_t *y, int16_t *cb, int16_t *cr, uint8_t *cls_out)
{
  if ((w & 7) || (h & 7) || n_sets < 0 || n_sets > 8 || n_alt < 0 || n_alt > 8) return -1;
  const int cw = (w + 127) >> 7, chh = (h + 127) >> 7, maxv = (1 << bit_depth) - 1;
  int16_t *planes[3] = { y, cb, cr };
  static const uint8_t perm7[4][12] = { { 0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11 }, { 9, 4, 10, 8, 1, 5, 11, 7, 3, 0, 2, 6 }, { 0, 3, 2, 1, 8, 7, 6, 5, 4, 9, 10, 11 }, { 9, 8, 10, 4, 3, 7, 11, 5, 1, 0, 2, 6 } };
  if (cls_out) memset(cls_out, 255, (size_t) (w >> 2) * (h >> 2));
  for (int c = 0; c < 3; c++) {
    const int sh = c ? 1 : 0, pw = w >> sh, ph = h >> sh, ctuS = 128 >> sh;
    int16_t *src = (int16_t *) malloc((size_t) pw * ph * sizeof(int16_t));
    memcpy(src, planes[c], (size_t) pw * ph * sizeof(int16_t));
    const plane_t S = { src, pw, ph };
    for (int by = 0; by < ph; by += 4) for (int bx = 0; bx < pw; bx += 4) {
      const int a = (by / ctuS) * cw + bx / ctuS;
      const orc_alf_ctu *u = &ctu[a];
      if (!u->flag[c]) continue;
      const int lastRow = by / ctuS == chh - 1, vbPos = lastRow ? h : ctuS - (c ? 2 : 4);      /* the last CTU row: the LUMA height for every component (ALFProcess 296, 313), which only a picture of at most 128 rows reaches */
      int co[12], cl[12];
      if (c == 0) {
        int cls, tr;
        classify(&S, bx, by, bit_depth, ctuS, vbPos, &cls, &tr);
        if (cls_out) cls_out[(size_t) (by >> 2) * (w >> 2) + (bx >> 2)] = (uint8_t) (cls | (tr << 5));
        const int set = u->set;
        if (set < 0 || set >= 16 + n_sets) { free(src); return -1; }
        for (int i = 0; i < 12; i++) {
          const int k = perm7[tr][i];
          if (set < 16) { co[i] = ORC_ALF_FIXED[ORC_ALF_CLASS_TO_FIXED[set][cls]][k]; cl[i] = 1 << bit_depth; }
          else { co[i] = luma_coeff[((size_t) (set - 16) * 25 + cls) * 13 + k]; cl[i] = luma_clip[((size_t) (set - 16) * 25 + cls) * 13 + k]; }
        }
      } else {
        const int t = u->alt[c - 1];
        if (t < 0 || t >= n_alt) { free(src); return -1; }
        for (int i = 0; i < 6; i++) { co[i] = chroma_coeff[t * 7 + i]; cl[i] = chroma_clip[t * 7 + i]; }
      }
      for (int yy = by; yy < by + 4; yy++) for (int xx = bx; xx < bx + 4; xx++)
        planes[c][(size_t) yy * pw + xx] = (int16_t) filter_sample(&S, xx, yy, c != 0, co, cl, yy & (ctuS - 1), vbPos, maxv);
    }
    free(src);
  }
  return 0;
}

/*
 * orc_sao.c — CPU restatement of the sample adaptive offset FILTER with the per-CTU parameters given: what the decoder side of
 * SampleAdaptiveOffset does to a deblocked picture.  TEST INFRASTRUCTURE ONLY (see vvc_oracle.h).
 *
 * Follows CL/SampleAdaptiveOffset.cpp: SAOProcess 617-670 (parameters resolved for the whole picture, then every CTU filtered from a copy of the deblocked picture),
 * xReconstructBlkSAOParams 265-290 with getMergeList 172-227 (a merge candidate is the CTU to the left / above in the same tile) and reconstructBlkSAOParam 229-263,
 * invertQuantOffsets 147-170 (offsets times 2^log2OffsetScale; band offsets belong to four consecutive bands from the band position, modulo 32; edge offsets to the
 * classes full valley, half valley, half peak, full peak), offsetCTU 548-615 and offsetBlock 292-546 with deriveLoopFilterBoundaryAvailibility 818-883 (one slice;
 * neighbour CTUs of another tile only when the loop filters may cross tile borders).
 *
 * The reference walks a CTU row by row with sign line buffers; what it computes per sample is: the two neighbours along the class direction (0 degrees: left / right,
 * 90: above / below, 135: above-left / below-right, 45: above-right / below-left), both of which must be available - inside the picture and, unless filtering across
 * tiles is allowed, inside the tile of the sample's CTU - else the sample stays; edge index = sign(c - a) + sign(c - b) in -2..2 picks the offset (index 0: none).
 * Pinned against the reference's SampleAdaptiveOffset (oracle/_ref, tests/golden/sao.npz).
 */
#include <stdlib.h>
#include <string.h>
#include "orc_internal.h"

static int sgn_(int v) { return (v > 0) - (v < 0); }
static int tile_of(int ctu, int n_ctus, int n_tiles) { int t = 0; for (int i = 0; i < n_tiles; i++) if (ctu >= (i * n_ctus) / n_tiles) t = i; return t; }

/* prm: per CTU (raster) and component {mode 0 off / 1 new / 2 merge, type (new: 0..3 edge class 0 / 90 / 135 / 45 degrees, 4 band; merge: 0 left, 1 above),
 * band position, four coded offsets}.  planes: int16, stride = plane width, filtered in place.  Returns 0, or -1 for a merge without a candidate / bad values. */
int orc_sao_picture(int w, int h, int bit_depth, int tile_cols, int tile_rows, int lf_across_tiles, int log2_offset_scale, const orc_sao_param *prm, int16_t *y, int16_t *cb, int16_t *cr)
{
  const int cw = (w + 127) >> 7, ch = (h + 127) >> 7, nctu = cw * ch;
  int16_t *planes[3] = { y, cb, cr };
  /* resolve merges in raster order: table[ctu][comp] = {type or -1, 32 offsets by class / band} */
  int8_t *type = (int8_t *) malloc((size_t) nctu * 3);
  int16_t *off = (int16_t *) calloc((size_t) nctu * 3 * 32, sizeof(int16_t));
  int rc = 0;
  for (int a = 0; a < nctu && !rc; a++) {
    const int cx = a % cw, cy = a / cw, tx = tile_of(cx, cw, tile_cols), ty = tile_of(cy, ch, tile_rows);
    for (int c = 0; c < 3; c++) {
      const orc_sao_param *p = &prm[a * 3 + c];
      int8_t *t = &type[a * 3 + c]; int16_t *o = &off[(size_t) (a * 3 + c) * 32];
      *t = -1;
      if (p->mode == 0) continue;
      if (p->mode == 1) {
        if (p->type < 0 || p->type > 4 || p->band < 0 || p->band > 31) { rc = -1; break; }
        *t = p->type;
        if (p->type == 4) for (int i = 0; i < 4; i++) o[(p->band + i) & 31] = (int16_t) (p->off[i] * (1 << log2_offset_scale));
        else { o[0] = (int16_t) (p->off[0] * (1 << log2_offset_scale)); o[1] = (int16_t) (p->off[1] * (1 << log2_offset_scale)); o[3] = (int16_t) (p->off[2] * (1 << log2_offset_scale)); o[4] = (int16_t) (p->off[3] * (1 << log2_offset_scale)); }
      } else {
        const int left = p->type == 0, sx = left ? cx - 1 : cx, sy = left ? cy : cy - 1;
        if (sx < 0 || sy < 0 || tile_of(sx, cw, tile_cols) != tx || tile_of(sy, ch, tile_rows) != ty) { rc = -1; break; }      /* CHECK(mergeTarget == NULL) */
        const int s = sy * cw + sx;
        *t = type[s * 3 + c]; memcpy(o, &off[(size_t) (s * 3 + c) * 32], 32 * sizeof(int16_t));
      }
    }
  }
  for (int c = 0; c < 3 && !rc; c++) {
    const int sh = c ? 1 : 0, pw = w >> sh, ph = h >> sh, mx = (1 << bit_depth) - 1, cs = 128 >> sh;
    int16_t *src = (int16_t *) malloc((size_t) pw * ph * sizeof(int16_t));
    memcpy(src, planes[c], (size_t) pw * ph * sizeof(int16_t));
    for (int yy = 0; yy < ph; yy++) for (int xx = 0; xx < pw; xx++) {
      const int cx = xx / cs, cy = yy / cs, a = cy * cw + cx, t = type[a * 3 + c];
      if (t < 0) continue;
      const int16_t *o = &off[(size_t) (a * 3 + c) * 32];
      const int v = src[yy * pw + xx];
      int r;
      if (t == 4) r = v + o[v >> (bit_depth - 5)];
      else {
        const int dx = t == 1 ? 0 : 1, dy = t == 0 ? 0 : 1;
        const int ax = t == 3 ? xx + 1 : xx - dx, ay = yy - dy, bx = t == 3 ? xx - 1 : xx + dx, by = yy + dy;      /* 45 degrees: above-right and below-left */
        int ok = ax >= 0 && ax < pw && ay >= 0 && ay < ph && bx >= 0 && bx < pw && by >= 0 && by < ph;
        if (ok && !lf_across_tiles) {
          const int tx = tile_of(cx, cw, tile_cols), ty = tile_of(cy, ch, tile_rows);
          ok = tile_of(ax / cs, cw, tile_cols) == tx && tile_of(ay / cs, ch, tile_rows) == ty && tile_of(bx / cs, cw, tile_cols) == tx && tile_of(by / cs, ch, tile_rows) == ty;
        }
        if (!ok) continue;
        r = v + o[2 + sgn_(v - src[ay * pw + ax]) + sgn_(v - src[by * pw + bx])];
      }
      planes[c][yy * pw + xx] = (int16_t) (r < 0 ? 0 : r > mx ? mx : r);
    }
    free(src);
  }
  free(type); free(off);
  return rc;
}

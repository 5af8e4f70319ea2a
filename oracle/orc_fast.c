/*
 * orc_fast.c — CPU restatement of the fork's FAST_ALGORITHM path: the 26 hand-crafted features of a luma
 * node and the random-forest inference that picks the one partition mode to try.  TEST INFRASTRUCTURE ONLY (see vvc_oracle.h).
 *
 * PARITY UNPINNED for this file: the reference computes the features with OpenCV (absent from the image, so EL/EncCu.cpp does not
 * compile here with FAST_ALGORITHM=1) and asks a pickled sklearn forest (Partition_32.pkl) that is not in the reference repository.
 * The OpenCV operations are restated from their documented behaviour:
 *   - Mat::convertTo(CV_8U)            : saturate to 0..255
 *   - cv::meanStdDev                   : mean = s * (1/N), stddev = sqrt(max(sq * (1/N) - mean * mean, 0)) in double (population)
 *   - cv::filter2D(src 8U, ddepth 8U)  : correlation, anchor at the kernel centre, BORDER_REFLECT_101, result saturated to 0..255
 *   - Mat / 4 + Mat / 4 + ...  (8U)    : left to right, every partial sum rounded half-to-even and saturated to 0..255
 * Feature order and integer truncations follow EL/EncCu.cpp:863-1123; get_madp 73-134; get_context 137-163.
 * The forest is the flattened form of sklearn's RandomForestClassifier (tree_.feature / threshold / children_left / children_right /
 * value): predict = argmax over classes of the per-tree leaf distributions summed in tree order (ForestClassifier.predict_proba);
 * features are compared as float32 like sklearn's DTYPE.
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include "orc_internal.h"

static int sat8(int v) { return v < 0 ? 0 : v > 255 ? 255 : v; }

/* cv::meanStdDev over a w x h region of 8-bit-saturated samples; returns int(stddev * stddev) like the reference's int(var) */
static double region_var(const int16_t *p, int stride, int w, int h)
{
  long long s = 0, sq = 0;
  for (int j = 0; j < h; j++) for (int i = 0; i < w; i++) { const int v = sat8(p[j * stride + i]); s += v; sq += v * v; }
  const double scale = 1.0 / (double) (w * h);
  const double mean = (double) s * scale;
  double var = (double) sq * scale - mean * mean;
  if (var < 0) var = 0;
  const double sd = sqrt(var);
  return sd * sd;
}
int orc_fast_region_var(const int16_t *p, int stride, int w, int h) { return (int) region_var(p, stride, w, h); }

/* features 4..11 and 21..25 of a w x h block (block-local: the filters and MADP see only the block, like the cv::Mat they run on) */
void orc_fast_block_features(const int16_t *org, int stride, int w, int h, int feat[26])
{
  static uint8_t px[64 * 64];
  for (int j = 0; j < h; j++) for (int i = 0; i < w; i++) px[j * w + i] = (uint8_t) sat8(org[j * stride + i]);
#define PX(i_, j_) ((int) px[(j_) * w + (i_)])
  /* get_madp (EL/EncCu.cpp:73-134): mean absolute difference to the existing 8-neighbours, integer division by their count */
  long long ms = 0, msq = 0;
  for (int j = 0; j < h; j++) for (int i = 0; i < w; i++) {
    int sum = 0, n = 0;
    for (int dj = -1; dj <= 1; dj++) for (int di = -1; di <= 1; di++) {
      if (!di && !dj) continue;
      const int x = i + di, y = j + dj;
      if (x < 0 || y < 0 || x >= w || y >= h) continue;
      sum += abs(PX(x, y) - PX(i, j)); n++;
    }
    const int m = sum / n;       /* n is 3 (corner), 5 (edge) or 8 */
    ms += m; msq += (long long) m * m;
  }
  /* the four 3x3 directional kernels (997-1013), reflect-101 border */
  static const int K[4][9] = { { -1, 0, 1, -2, 0, 2, -1, 0, 1 }, { 1, 2, 1, 0, 0, 0, -1, -2, -1 }, { 0, 1, 2, -1, 0, 1, -2, -1, 0 }, { 2, 1, 0, 1, 0, -1, 0, -1, -2 } };
  long long gs[4] = { 0, 0, 0, 0 };
  int gmax = 0;
  for (int j = 0; j < h; j++) for (int i = 0; i < w; i++) {
    int g[4];
    for (int k = 0; k < 4; k++) {
      int acc = 0;
      for (int dj = -1; dj <= 1; dj++) for (int di = -1; di <= 1; di++) {
        int x = i + di, y = j + dj;
        if (x < 0) x = -x;
        if (x >= w) x = 2 * w - 2 - x;
        if (y < 0) y = -y;
        if (y >= h) y = 2 * h - 2 - y;
        acc += K[k][(dj + 1) * 3 + di + 1] * PX(x, y);
      }
      g[k] = sat8(acc); gs[k] += g[k];
    }
    /* Gra_H / 4 + Gra_V / 4 + Gra_45 / 4 + Gra_135 / 4 on 8-bit matrices (1027): quarter values are exact in float, every partial
     * sum is rounded to nearest-even and saturated */
    float t = (float) g[0] * 0.25f + (float) g[1] * 0.25f;
    int q = sat8((int) nearbyintf(t));
    t = (float) q + (float) g[2] * 0.25f; q = sat8((int) nearbyintf(t));
    t = (float) q + (float) g[3] * 0.25f; q = sat8((int) nearbyintf(t));
    if (q > gmax) gmax = q;
  }
  const int N = w * h;
  const double G_H = (double) gs[0] / N, G_V = (double) gs[1] / N, G_45 = (double) gs[2] / N, G_135 = (double) gs[3] / N;
  const double Gra = (G_H + G_V + G_45 + G_135) / 4;
  feat[4] = (int) G_H; feat[5] = (int) G_V; feat[6] = (int) G_45; feat[7] = (int) G_135; feat[8] = (int) Gra; feat[9] = gmax;
  feat[10] = (int) region_var(org, stride, w, h);
  {
    const double scale = 1.0 / (double) N, mean = (double) ms * scale;
    double var = (double) msq * scale - mean * mean;
    if (var < 0) var = 0;
    const double sd = sqrt(var);
    feat[11] = (int) (sd * sd);
  }
  /* variance of the sub-block variances for the five split shapes (1053-1095) */
  {
    const int B1 = (int) region_var(org, stride, w, h / 2), B2 = (int) region_var(org + (h / 2) * stride, stride, w, h - h / 2);
    const int m = (B1 + B2) / 2;
    feat[21] = ((B1 - m) * (B1 - m) + (B2 - m) * (B2 - m)) / 2;
  }
  {
    const int B1 = (int) region_var(org, stride, w / 2, h), B2 = (int) region_var(org + w / 2, stride, w - w / 2, h);
    const int m = (B1 + B2) / 2;
    feat[22] = ((B1 - m) * (B1 - m) + (B2 - m) * (B2 - m)) / 2;
  }
  {
    const int T1 = (int) region_var(org, stride, w, h / 4), T2 = (int) region_var(org + (h / 4) * stride, stride, w, 3 * h / 4 - h / 4);
    const int T3 = (int) region_var(org + (3 * h / 4) * stride, stride, w, h - 3 * h / 4);
    const int m = (T1 + T2 + T3) / 3;
    feat[23] = ((T1 - m) * (T1 - m) + (T2 - m) * (T2 - m) + (T3 - m) * (T3 - m)) / 3;
  }
  {
    const int T1 = (int) region_var(org, stride, w / 4, h), T2 = (int) region_var(org + w / 4, stride, 3 * w / 4 - w / 4, h);
    const int T3 = (int) region_var(org + 3 * w / 4, stride, w - 3 * w / 4, h);
    const int m = (T1 + T2 + T3) / 3;
    feat[24] = ((T1 - m) * (T1 - m) + (T2 - m) * (T2 - m) + (T3 - m) * (T3 - m)) / 3;
  }
  {
    const int Q1 = (int) region_var(org, stride, w / 2, h / 2), Q2 = (int) region_var(org + w / 2, stride, w - w / 2, h / 2);
    const int Q3 = (int) region_var(org + (h / 2) * stride, stride, w / 2, h - h / 2), Q4 = (int) region_var(org + (h / 2) * stride + w / 2, stride, w - w / 2, h - h / 2);
    const int m = (Q1 + Q2 + Q3 + Q4) / 4;
    feat[25] = ((Q1 - m) * (Q1 - m) + (Q2 - m) * (Q2 - m) + (Q3 - m) * (Q3 - m) + (Q4 - m) * (Q4 - m)) / 4;
  }
#undef PX
}

/* neighbour statistics (943-983): nb[k] = { variance, qtDepth, mtDepth } of the k-th usable neighbour CU → features 12..20 */
void orc_fast_context_features(const int nb[][3], int n, int feat[26])
{
  for (int c = 0; c < 3; c++) {
    int mx = nb[0][c], mn = nb[0][c], sum = 0;
    for (int k = 0; k < n; k++) { if (nb[k][c] > mx) mx = nb[k][c]; if (nb[k][c] < mn) mn = nb[k][c]; sum += nb[k][c]; }
    feat[12 + 3 * c] = mx; feat[13 + 3 * c] = mn; feat[14 + 3 * c] = sum / n;
  }
}

/* sklearn ForestClassifier.predict for one row: leaf distributions summed in tree order, first maximum wins */
int orc_forest_predict(const orc_forest *f, const int feat[26])
{
  double acc[8] = { 0, 0, 0, 0, 0, 0, 0, 0 };
  for (int t = 0; t < f->n_trees; t++) {
    int n = f->root[t];
    while (f->left[n] >= 0) n = ((double) (float) feat[f->feature[n]] <= f->threshold[n]) ? f->left[n] : f->right[n];
    for (int c = 0; c < f->n_classes; c++) acc[c] += f->value[(size_t) n * f->n_classes + c];
  }
  int best = 0;
  for (int c = 1; c < f->n_classes; c++) if (acc[c] > acc[best]) best = c;
  return f->classes[best];
}

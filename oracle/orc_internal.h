/* orc_internal.h — internal types shared by the oracle's translation units (TEST INFRASTRUCTURE ONLY). */
#ifndef ORC_INTERNAL_H
#define ORC_INTERNAL_H
#include "vvc_oracle.h"
#include "orc_tables.h"
#include <string.h>

/* BitEstimator_Std + Ctx (EL/BinEncoder.h:238-303, CL/Contexts.h:86-155): flat context array in the
 * reference's own order (ORC_CTX_* offsets), fractional bits in 2^-15 units */
/* BinEncoderBase / TBinEncoder<BinProbModel_Std> (EL/BinEncoder.cpp:106-420): the arithmetic coder behind the same
 * context models.  Used by the oracle's bitstream pass only (orc_write_tiles); NULL while estimating. */
typedef struct {
  uint32_t low, range, buffered_byte; int bits_left, num_buffered;
  uint8_t *out; size_t n, cap; uint32_t bit_acc; int bit_n;      /* OutputBitstream: bytes + pending bits */
} orc_arith;

typedef struct {
  uint16_t s0[ORC_NUM_CTX], s1[ORC_NUM_CTX];
  uint64_t bits;
  orc_arith *aw;
  int dq;                  /* slice dep_quant_enabled_flag: residual_coding runs the quantiser state machine */
  int last_scan_pos;       /* scanPosLast of the block coded last */
} orc_cabac;

static inline void orc_bs_write(orc_arith *a, uint32_t v, int nbits)           /* OutputBitstream::write, MSB first */
{
  for (int i = nbits - 1; i >= 0; i--) {
    a->bit_acc = (a->bit_acc << 1) | ((v >> i) & 1); a->bit_n++;
    if (a->bit_n == 8) { if (a->n < a->cap) a->out[a->n] = (uint8_t) a->bit_acc; a->n++; a->bit_acc = 0; a->bit_n = 0; }
  }
}
static inline void orc_arith_start(orc_arith *a) { a->low = 0; a->range = 510; a->buffered_byte = 0xff; a->num_buffered = 0; a->bits_left = 23; }   /* 122-131 */
static inline void orc_arith_write_out(orc_arith *a)                            /* writeOut 341-371 */
{
  const uint32_t lead = a->low >> (24 - a->bits_left);
  a->bits_left += 8;
  a->low &= 0xffffffffu >> a->bits_left;
  if (lead == 0xff) a->num_buffered++;
  else if (a->num_buffered > 0) {
    const uint32_t carry = lead >> 8;
    uint32_t byte = a->buffered_byte + carry;
    a->buffered_byte = lead & 0xff;
    orc_bs_write(a, byte, 8);
    byte = (0xff + carry) & 0xff;
    while (a->num_buffered > 1) { orc_bs_write(a, byte, 8); a->num_buffered--; }
  } else { a->num_buffered = 1; a->buffered_byte = lead; }
}
static inline void orc_arith_bins_ep(orc_arith *a, uint32_t bins, int n)        /* encodeBinsEP 186-216 (range is never 256-aligned here) */
{
  while (n > 8) {
    n -= 8;
    const uint32_t pattern = bins >> n;
    a->low <<= 8; a->low += a->range * pattern; bins -= pattern << n; a->bits_left -= 8;
    if (a->bits_left < 12) orc_arith_write_out(a);
  }
  a->low <<= n; a->low += a->range * bins; a->bits_left -= n;
  if (a->bits_left < 12) orc_arith_write_out(a);
}
static inline void orc_arith_trm(orc_arith *a, unsigned bin)                    /* encodeBinTrm 268-291 */
{
  a->range -= 2;
  if (bin) { a->low += a->range; a->low <<= 7; a->range = 2 << 7; a->bits_left -= 7; }
  else if (a->range >= 256) return;
  else { a->low <<= 1; a->range <<= 1; a->bits_left--; }
  if (a->bits_left < 12) orc_arith_write_out(a);
}
static inline void orc_arith_finish(orc_arith *a)                               /* finish 133-158 */
{
  if (a->low >> (32 - a->bits_left)) {
    orc_bs_write(a, a->buffered_byte + 1, 8);
    while (a->num_buffered > 1) { orc_bs_write(a, 0x00, 8); a->num_buffered--; }
    a->low -= 1u << (32 - a->bits_left);
  } else {
    if (a->num_buffered > 0) orc_bs_write(a, a->buffered_byte, 8);
    while (a->num_buffered > 1) { orc_bs_write(a, 0xff, 8); a->num_buffered--; }
  }
  orc_bs_write(a, a->low >> 8, 24 - a->bits_left);
}

static inline void orc_enc_bin(orc_cabac *c, unsigned bin, int ctx)
{
  const unsigned st = (unsigned) (c->s0[ctx] + c->s1[ctx]) >> 8;
  c->bits += ORC_BIN_FRAC_BITS[st * 2 + bin];
  if (c->aw) {                                        /* TBinEncoder::encodeBin 378-420 */
    orc_arith *a = c->aw;
    unsigned q = st & 0xff; const unsigned mps = q >> 7;
    if (q & 0x80) q ^= 0xff;
    const uint32_t lps = (((q >> 2) * (a->range >> 5)) >> 1) + 4;       /* getLPS, CL/Contexts.h:134-140 */
    a->range -= lps;
    if (bin != mps) {
      static const uint8_t renorm[32] = { 6, 5, 4, 4, 3, 3, 3, 3, 2, 2, 2, 2, 2, 2, 2, 2, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1 };   /* CL/Contexts.cpp:45-55 */
      const int nb = renorm[lps >> 3];
      a->bits_left -= nb; a->low += a->range; a->low <<= nb; a->range = lps << nb;
      if (a->bits_left < 12) orc_arith_write_out(a);
    } else if (a->range < 256) {
      a->bits_left -= 1; a->low <<= 1; a->range <<= 1;
      if (a->bits_left < 12) orc_arith_write_out(a);
    }
  }
  const int rate = ORC_CTX_RATE[ctx];                 /* log2WindowSize; setLog2WindowSize 116-122 */
  const int r0 = 2 + ((rate >> 2) & 3), r1 = 3 + r0 + (rate & 3);
  c->s0[ctx] -= (c->s0[ctx] >> r0) & 0x7FE0;
  c->s1[ctx] -= (c->s1[ctx] >> r1) & 0x7FFE;
  if (bin) { c->s0[ctx] += (0x7fffu >> r0) & 0x7FE0; c->s1[ctx] += (0x7fffu >> r1) & 0x7FFE; }
}
/* n bypass bins with the given value (MSB first); the estimator only needs n */
static inline void orc_enc_bins_ep(orc_cabac *c, uint32_t value, int n) { c->bits += (uint64_t) n << 15; if (c->aw && n > 0) orc_arith_bins_ep(c->aw, value, n); }
static inline void orc_ctx_copy(orc_cabac *d, const orc_cabac *s) { memcpy(d->s0, s->s0, sizeof d->s0); memcpy(d->s1, s->s1, sizeof d->s1); }

/* IntraPredParam (CL/IntraPrediction.h:75-110) */
typedef struct {
  int pred_mode, is_ver, mrl, ref_filter, interp, pdpc, angle, inv_angle, ang_scale;
} orc_ipa;

void orc_init_pred_params(int w, int h, int is_luma, int mode, int mrl, orc_ipa *p);
void orc_init_pred_params_isp(int cuw, int cuh, int w, int h, int mode, orc_ipa *p);
void orc_pred_intra_isp(const int16_t *src, int st, int cuw, int cuh, int w, int h, int mode, int bit_depth, int16_t *pred, int ps);
void orc_fwd_isp(const int16_t *resi, int stride, int w, int h, int bit_depth, int *coef);
void orc_inv_isp(const int *coef, int w, int h, int bit_depth, int16_t *resi, int stride);
void orc_cg_shape(int w, int h, int *lcw, int *lch);
void orc_satd_tile_shape(int w, int h, int *bw, int *bh);
const int8_t *orc_tr_matrix(int tr, int n);
/* CCLM (orc_leaf.c) */
void orc_cclm_luma(const int16_t *recY, int strideY, const uint8_t *avail, int avail_stride, int tag, int pic_wc, int pic_hc,
                   int cx, int cy, int cw, int ch, int mdlm, int info[4], int16_t *tmp, int tstride);
void orc_cclm_params(const int16_t *tmp, int tstride, const int16_t *ref, int cw, int ch, int mode, const int info[4], int bit_depth, int *pa, int *pb, int *pshift);
void orc_pred_cclm(const int16_t *tmp, int tstride, int a, int b, int shift, int bit_depth, int cw, int ch, int16_t *pred, int pstride);
/* FAST_ALGORITHM restatement (orc_fast.c) */
typedef struct { int n_trees, n_nodes, n_classes; int32_t *root, *feature, *left, *right; double *threshold, *value; int32_t classes[8]; } orc_forest;
int  orc_fast_region_var(const int16_t *p, int stride, int w, int h);
void orc_fast_block_features(const int16_t *org, int stride, int w, int h, int feat[26]);
void orc_fast_context_features(const int nb[][3], int n, int feat[26]);
int  orc_forest_predict(const orc_forest *f, const int feat[26]);
/* deblocking filter (orc_deblock.c) */
void orc_deblock_luma_segment(int16_t *s, int o, int step, int sizeP, int sizeQ, int ctuTop, int qp, int bd, int beta_off2, int tc_off2);
void orc_deblock_chroma_segment(int16_t *s, int o, int step, int sizeP, int sizeQ, int ctuTop, int qp, int bd, int beta_off2, int tc_off2);
/* residual_coding on the estimator (orc_rate.c) */
void orc_residual_coding(orc_cabac *c, const int16_t *level, int w, int h, int is_chroma);
void orc_residual_coding_mts(orc_cabac *c, const int16_t *level, int w, int h, int is_chroma, int mts_idx);
void orc_residual_coding_tu(orc_cabac *c, const int16_t *level, int w, int h, int is_chroma, int ts_allowed, int mts_allowed, int mts_idx);
void orc_enc_rem_abs(orc_cabac *cb, unsigned bins, unsigned rice);
/* transform skip (orc_ts.c) */
int  orc_ts_qp(int qp);
void orc_ts_fwd(const int16_t *resi, int stride, int w, int h, int bit_depth, int *coef);
void orc_ts_inv(const int *coef, int w, int h, int bit_depth, int16_t *resi, int stride);
int  orc_ts_sumabs(const int16_t *resi, int stride, int w, int h, int bit_depth);
void orc_dequant_ts(const int16_t *level, int w, int h, int bit_depth, int qp, int *coef);
int  orc_rdoq_ts(const uint16_t *s0, const uint16_t *s1, const int *coef, int w, int h, int bit_depth, int qp, double lambda, int16_t *level);
void orc_residual_coding_ts(orc_cabac *cb, const int16_t *coeff, int w, int h);

#endif

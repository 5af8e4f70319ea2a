/* orc_internal.h — internal types shared by the oracle's translation units (TEST INFRASTRUCTURE ONLY). */
#ifndef ORC_INTERNAL_H
#define ORC_INTERNAL_H
#include "vvc_oracle.h"
#include "orc_tables.h"
#include <string.h>

/* BitEstimator_Std + Ctx (EL/BinEncoder.h:238-303, CL/Contexts.h:86-155): flat context array in the
 * reference's own order (ORC_CTX_* offsets), fractional bits in 2^-15 units */
typedef struct {
  uint16_t s0[ORC_NUM_CTX], s1[ORC_NUM_CTX];
  uint64_t bits;
} orc_cabac;

static inline void orc_enc_bin(orc_cabac *c, unsigned bin, int ctx)
{
  const unsigned st = (unsigned) (c->s0[ctx] + c->s1[ctx]) >> 8;
  c->bits += ORC_BIN_FRAC_BITS[st * 2 + bin];
  const int rate = ORC_CTX_RATE[ctx];                 /* log2WindowSize; setLog2WindowSize 116-122 */
  const int r0 = 2 + ((rate >> 2) & 3), r1 = 3 + r0 + (rate & 3);
  c->s0[ctx] -= (c->s0[ctx] >> r0) & 0x7FE0;
  c->s1[ctx] -= (c->s1[ctx] >> r1) & 0x7FFE;
  if (bin) { c->s0[ctx] += (0x7fffu >> r0) & 0x7FE0; c->s1[ctx] += (0x7fffu >> r1) & 0x7FFE; }
}
static inline void orc_enc_ep(orc_cabac *c, int n) { c->bits += (uint64_t) n << 15; }
static inline void orc_ctx_copy(orc_cabac *d, const orc_cabac *s) { memcpy(d->s0, s->s0, sizeof d->s0); memcpy(d->s1, s->s1, sizeof d->s1); }

/* IntraPredParam (CL/IntraPrediction.h:75-110) */
typedef struct {
  int pred_mode, is_ver, mrl, ref_filter, interp, pdpc, angle, inv_angle, ang_scale;
} orc_ipa;

void orc_init_pred_params(int w, int h, int is_luma, int mode, int mrl, orc_ipa *p);
void orc_cg_shape(int w, int h, int *lcw, int *lch);
void orc_satd_tile_shape(int w, int h, int *bw, int *bh);
const int8_t *orc_tr_matrix(int tr, int n);
/* residual_coding on the estimator (orc_rate.c) */
void orc_residual_coding(orc_cabac *c, const int16_t *level, int w, int h, int is_chroma);

#endif

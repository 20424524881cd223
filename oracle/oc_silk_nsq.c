/* oracle/oc_silk_nsq.c — TEST INFRASTRUCTURE ONLY.  Plain-C restatement of the SILK noise-shaping quantiser without delayed
 * decision (silk/NSQ.c:76-181 frame driver, :183-366 sample loop, :368-436 state rescaling), its two tap loops
 * (silk/NSQ.h:35-63 short-term prediction, :67-96 shaping feedback) and the MA whitening filter
 * (silk/LPC_analysis_filter.c:49-108, the USE_CELT_FIR 0 branch).  Pure int32 arithmetic with the reference's intentional
 * wrap-arounds.  Checked sample-for-sample (pulses, xq, every state word) against the compiled reference's silk_NSQ_c and
 * silk_LPC_analysis_filter by tests/test_oracle_silk.py. */
#include "oc_silk.h"

s32 oc_silk_div32_varQ(s32 a32, s32 b32, int Qres)                       /* silk/Inlines.h:93-140 */
{
   int ha = q_clz32(q_abs(a32)) - 1, hb = q_clz32(q_abs(b32)) - 1;
   s32 an = q_shlw(a32, ha), bn = q_shlw(b32, hb);
   s32 binv = (2147483647 >> 2) / (bn >> 16);
   s32 r = q_mulwb(an, binv);
   an = q_subw(an, q_shlw(q_smmul(bn, r), 3));
   r = q_mlawb(r, an, binv);
   int ls = 29 + ha - hb - Qres;
   if (ls < 0) return q_shl_sat(r, -ls);
   return ls < 32 ? r >> ls : 0;
}

s32 oc_silk_inverse32_varQ(s32 b32, int Qres)                            /* silk/Inlines.h:143-183 */
{
   int hb = q_clz32(q_abs(b32)) - 1;
   s32 bn = q_shlw(b32, hb);
   s32 binv = (2147483647 >> 2) / (bn >> 16);
   s32 r = q_shlw(binv, 16);
   s32 err = q_shlw(((s32)1 << 29) - q_mulwb(bn, binv), 3);
   r = q_mlaww(r, err, binv);
   int ls = 61 - hb - Qres;
   if (ls <= 0) return q_shl_sat(r, -ls);
   return ls < 32 ? r >> ls : 0;
}

void oc_silk_lpc_analysis_filter(s16 *out, const s16 *in, const s16 *B, s32 len, s32 d)
{
   for (s32 n = d; n < len; n++) {
      s32 pred = 0;
      for (s32 k = 0; k < d; k++) pred = q_mlabb(pred, in[n - 1 - k], B[k]);     /* wrap-around MAC, order-independent mod 2^32 */
      s32 e = q_subw(q_shlw(in[n], 12), pred);
      out[n] = (s16)q_sat16(q_rshift_round(e, 12));
   }
   memset(out, 0, (size_t)d * sizeof(s16));
}

static const s16 kQuantOffsets_Q10[2][2] = { { 100, 240 }, { 32, 100 } };  /* silk/tables_other.c:77, silk/define.h OFFSET_{UVL,UVH,VL,VH}_Q10 */
#define QUANT_LEVEL_ADJUST_Q10 80

/* silk/NSQ.c:368-436 */
static void nsq_scale_states(const OcSilkNsqCfg *cfg, OcSilkNsqState *st, const s16 *x16, s32 *x_sc_Q10, const s16 *sLTP,
                             s32 *sLTP_Q15, int subfr, const OcSilkNsqFrame *fr)
{
   const int L = oc_cfg_subfr(cfg), mem = oc_cfg_ltp_mem(cfg);
   const int lag = fr->pitchL[subfr];
   const s32 gain = fr->Gains_Q16[subfr];
   s32 inv_gain_Q31 = oc_silk_inverse32_varQ(gain > 1 ? gain : 1, 47);
   const s32 inv_gain_Q26 = q_rshift_round(inv_gain_Q31, 5);
   for (int i = 0; i < L; i++) x_sc_Q10[i] = q_mulww(x16[i], inv_gain_Q26);

   if (st->rewhite_flag) {
      if (subfr == 0) inv_gain_Q31 = q_mulwb(inv_gain_Q31, fr->LTP_scale_Q14) << 2;
      for (int i = st->sLTP_buf_idx - lag - OC_SILK_LTP_ORDER / 2; i < st->sLTP_buf_idx; i++) sLTP_Q15[i] = q_mulwb(inv_gain_Q31, sLTP[i]);
   }
   if (gain != st->prev_gain_Q16) {
      const s32 adj = oc_silk_div32_varQ(st->prev_gain_Q16, gain, 16);
      for (int i = st->sLTP_shp_buf_idx - mem; i < st->sLTP_shp_buf_idx; i++) st->sLTP_shp_Q14[i] = q_mulww(adj, st->sLTP_shp_Q14[i]);
      if (fr->signalType == OC_SILK_TYPE_VOICED && !st->rewhite_flag)
         for (int i = st->sLTP_buf_idx - lag - OC_SILK_LTP_ORDER / 2; i < st->sLTP_buf_idx; i++) sLTP_Q15[i] = q_mulww(adj, sLTP_Q15[i]);
      st->sLF_AR_shp_Q14 = q_mulww(adj, st->sLF_AR_shp_Q14);
      st->sDiff_shp_Q14 = q_mulww(adj, st->sDiff_shp_Q14);
      for (int i = 0; i < OC_SILK_LPC_BUF; i++) st->sLPC_Q14[i] = q_mulww(adj, st->sLPC_Q14[i]);
      for (int i = 0; i < OC_SILK_MAX_SHAPE; i++) st->sAR2_Q14[i] = q_mulww(adj, st->sAR2_Q14[i]);
      st->prev_gain_Q16 = gain;
   }
}

/* silk/NSQ.c:183-366: one subframe of the closed-loop scalar quantiser */
static void nsq_subframe(const OcSilkNsqCfg *cfg, OcSilkNsqState *st, int signalType, const s32 *x_sc_Q10, s8 *pulses, s16 *xq,
                         s32 *sLTP_Q15, const s16 *a_Q12, const s16 *b_Q14, const s16 *AR_shp_Q13, int lag, s32 HarmPacked_Q14,
                         int Tilt_Q14, s32 LF_shp_Q14, s32 Gain_Q16, int Lambda_Q10, int offset_Q10)
{
   const int L = oc_cfg_subfr(cfg), P = cfg->predictLPCOrder, S = cfg->shapingLPCOrder;
   const s32 Gain_Q10 = Gain_Q16 >> 6;
   s32 *shp_lag = &st->sLTP_shp_Q14[st->sLTP_shp_buf_idx - lag + 1];
   s32 *pred_lag = &sLTP_Q15[st->sLTP_buf_idx - lag + OC_SILK_LTP_ORDER / 2];
   s32 *lpc = &st->sLPC_Q14[OC_SILK_LPC_BUF - 1];

   for (int i = 0; i < L; i++) {
      st->rand_seed = q_rand(st->rand_seed);

      s32 LPC_pred_Q10 = P >> 1;                                            /* NSQ.h:35 */
      for (int k = 0; k < P; k++) LPC_pred_Q10 = q_mlawb(LPC_pred_Q10, lpc[-k], a_Q12[k]);

      s32 LTP_pred_Q13 = 0;
      if (signalType == OC_SILK_TYPE_VOICED) {
         LTP_pred_Q13 = 2;
         for (int k = 0; k < OC_SILK_LTP_ORDER; k++) LTP_pred_Q13 = q_mlawb(LTP_pred_Q13, pred_lag[-k], b_Q14[k]);
         pred_lag++;
      }

      /* NSQ.h:67: the AR-shaping delay line advances by one (sDiff enters at tap 0), output = taps . coefs */
      s32 n_AR_Q12 = S >> 1;
      {
         s32 carry = st->sDiff_shp_Q14;
         for (int k = 0; k < S; k++) {
            s32 old = st->sAR2_Q14[k];
            st->sAR2_Q14[k] = carry;
            n_AR_Q12 = q_mlawb(n_AR_Q12, carry, AR_shp_Q13[k]);
            carry = old;
         }
         n_AR_Q12 = q_shlw(n_AR_Q12, 1);
      }
      n_AR_Q12 = q_mlawb(n_AR_Q12, st->sLF_AR_shp_Q14, Tilt_Q14);

      s32 n_LF_Q12 = q_mulwb(st->sLTP_shp_Q14[st->sLTP_shp_buf_idx - 1], LF_shp_Q14);
      n_LF_Q12 = q_mlawt(n_LF_Q12, st->sLF_AR_shp_Q14, LF_shp_Q14);

      s32 t1 = q_subw(q_shlw(LPC_pred_Q10, 2), n_AR_Q12);
      t1 = q_subw(t1, n_LF_Q12);
      if (lag > 0) {
         s32 n_LTP_Q13 = q_mulwb(q_add_sat(shp_lag[0], shp_lag[-2]), HarmPacked_Q14);
         n_LTP_Q13 = q_mlawt(n_LTP_Q13, shp_lag[-1], HarmPacked_Q14);
         n_LTP_Q13 = q_shlw(n_LTP_Q13, 1);
         shp_lag++;
         s32 t2 = q_subw(LTP_pred_Q13, n_LTP_Q13);
         t1 = q_addw(t2, q_shlw(t1, 1));
         t1 = q_rshift_round(t1, 3);
      } else {
         t1 = q_rshift_round(t1, 2);
      }

      s32 r_Q10 = q_subw(x_sc_Q10[i], t1);
      if (st->rand_seed < 0) r_Q10 = q_subw(0, r_Q10);
      r_Q10 = q_limit(r_Q10, -(31 << 10), 30 << 10);

      s32 q1_Q10 = r_Q10 - offset_Q10;
      s32 q1_Q0 = q1_Q10 >> 10;
      if (Lambda_Q10 > 2048) {
         int rdo_offset = Lambda_Q10 / 2 - 512;
         if (q1_Q10 > rdo_offset) q1_Q0 = (q1_Q10 - rdo_offset) >> 10;
         else if (q1_Q10 < -rdo_offset) q1_Q0 = (q1_Q10 + rdo_offset) >> 10;
         else q1_Q0 = q1_Q10 < 0 ? -1 : 0;
      }
      s32 q2_Q10, rd1_Q20, rd2_Q20;
      if (q1_Q0 > 0) {
         q1_Q10 = (q1_Q0 << 10) - QUANT_LEVEL_ADJUST_Q10 + offset_Q10;
         q2_Q10 = q1_Q10 + 1024;
         rd1_Q20 = q_mulbb(q1_Q10, Lambda_Q10);
         rd2_Q20 = q_mulbb(q2_Q10, Lambda_Q10);
      } else if (q1_Q0 == 0) {
         q1_Q10 = offset_Q10;
         q2_Q10 = q1_Q10 + (1024 - QUANT_LEVEL_ADJUST_Q10);
         rd1_Q20 = q_mulbb(q1_Q10, Lambda_Q10);
         rd2_Q20 = q_mulbb(q2_Q10, Lambda_Q10);
      } else if (q1_Q0 == -1) {
         q2_Q10 = offset_Q10;
         q1_Q10 = q2_Q10 - (1024 - QUANT_LEVEL_ADJUST_Q10);
         rd1_Q20 = q_mulbb(-q1_Q10, Lambda_Q10);
         rd2_Q20 = q_mulbb(q2_Q10, Lambda_Q10);
      } else {
         q1_Q10 = q_shlw(q1_Q0, 10) + QUANT_LEVEL_ADJUST_Q10 + offset_Q10;
         q2_Q10 = q1_Q10 + 1024;
         rd1_Q20 = q_mulbb(-q1_Q10, Lambda_Q10);
         rd2_Q20 = q_mulbb(-q2_Q10, Lambda_Q10);
      }
      s32 rr = r_Q10 - q1_Q10;
      rd1_Q20 = q_mlabb(rd1_Q20, rr, rr);
      rr = r_Q10 - q2_Q10;
      rd2_Q20 = q_mlabb(rd2_Q20, rr, rr);
      if (rd2_Q20 < rd1_Q20) q1_Q10 = q2_Q10;

      pulses[i] = (s8)q_rshift_round(q1_Q10, 10);

      s32 exc_Q14 = q_shlw(q1_Q10, 4);
      if (st->rand_seed < 0) exc_Q14 = -exc_Q14;
      s32 LPC_exc_Q14 = exc_Q14 + q_shlw(LTP_pred_Q13, 1);
      s32 xq_Q14 = q_addw(LPC_exc_Q14, q_shlw(LPC_pred_Q10, 4));
      xq[i] = (s16)q_sat16(q_rshift_round(q_mulww(xq_Q14, Gain_Q10), 8));

      lpc++;
      *lpc = xq_Q14;
      st->sDiff_shp_Q14 = q_subw(xq_Q14, q_shlw(x_sc_Q10[i], 4));
      s32 sLF = q_subw(st->sDiff_shp_Q14, q_shlw(n_AR_Q12, 2));
      st->sLF_AR_shp_Q14 = sLF;
      st->sLTP_shp_Q14[st->sLTP_shp_buf_idx++] = q_subw(sLF, q_shlw(n_LF_Q12, 2));
      sLTP_Q15[st->sLTP_buf_idx++] = q_shlw(LPC_exc_Q14, 1);
      st->rand_seed = q_addw(st->rand_seed, pulses[i]);
   }
   memcpy(st->sLPC_Q14, &st->sLPC_Q14[L], OC_SILK_LPC_BUF * sizeof(s32));
}

void oc_silk_nsq(const OcSilkNsqCfg *cfg, OcSilkNsqState *st, OcSilkNsqFrame *fr, const s16 *x16, s8 *pulses)
{
   const int L = oc_cfg_subfr(cfg), mem = oc_cfg_ltp_mem(cfg), frame = oc_cfg_frame(cfg);
   s32 sLTP_Q15[2 * OC_SILK_MAX_FRAME];
   s16 sLTP[2 * OC_SILK_MAX_FRAME];
   s32 x_sc_Q10[OC_SILK_MAX_SUBFR];

   st->rand_seed = fr->Seed;
   int lag = st->lagPrev;
   const int offset_Q10 = kQuantOffsets_Q10[fr->signalType >> 1][fr->quantOffsetType];
   const int interp = fr->NLSFInterpCoef_Q2 == 4 ? 0 : 1;
   st->sLTP_shp_buf_idx = mem;
   st->sLTP_buf_idx = mem;
   s16 *pxq = &st->xq[mem];
   for (int k = 0; k < cfg->nb_subfr; k++) {
      const s16 *A_Q12 = &fr->PredCoef_Q12[((k >> 1) | (1 - interp)) * 16];
      const s16 *B_Q14 = &fr->LTPCoef_Q14[k * OC_SILK_LTP_ORDER];
      const s16 *AR_Q13 = &fr->AR_Q13[k * OC_SILK_MAX_SHAPE];
      s32 harm = (fr->HarmShapeGain_Q14[k] >> 2) | (s32)((u32)(fr->HarmShapeGain_Q14[k] >> 1) << 16);
      st->rewhite_flag = 0;
      if (fr->signalType == OC_SILK_TYPE_VOICED) {
         lag = fr->pitchL[k];
         if ((k & (3 - (interp << 1))) == 0) {
            int start = mem - lag - cfg->predictLPCOrder - OC_SILK_LTP_ORDER / 2;
            oc_silk_lpc_analysis_filter(&sLTP[start], &st->xq[start + k * L], A_Q12, mem - start, cfg->predictLPCOrder);
            st->rewhite_flag = 1;
            st->sLTP_buf_idx = mem;
         }
      }
      nsq_scale_states(cfg, st, x16, x_sc_Q10, sLTP, sLTP_Q15, k, fr);
      nsq_subframe(cfg, st, fr->signalType, x_sc_Q10, pulses, pxq, sLTP_Q15, A_Q12, B_Q14, AR_Q13, lag, harm, fr->Tilt_Q14[k],
                   fr->LF_shp_Q14[k], fr->Gains_Q16[k], fr->Lambda_Q10, offset_Q10);
      x16 += L; pulses += L; pxq += L;
   }
   st->lagPrev = fr->pitchL[cfg->nb_subfr - 1];
   memmove(st->xq, &st->xq[frame], (size_t)mem * sizeof(s16));
   memmove(st->sLTP_shp_Q14, &st->sLTP_shp_Q14[frame], (size_t)mem * sizeof(s32));
}

/* oracle/oc_silk_nsq_dd.c — TEST INFRASTRUCTURE ONLY.  Plain-C restatement of the delayed-decision noise-shaping quantiser
 * (silk/NSQ_del_dec.c:114-312 frame driver, :315-644 sample loop, :646-746 state rescaling; the path the reference takes at
 * complexity >= 2, silk/control_codec.c:376-386).  K survivor states each carry their own LPC/AR filter memories, dither seed
 * and a 40-deep ring of undecided samples; every sample each survivor proposes its two nearest quantisation levels, the best K
 * of the 2K continuations survive, and the sample `decisionDelay` steps back is committed from the current winner.
 * Checked word-for-word against the compiled reference's silk_NSQ_del_dec_c by tests/test_oracle_silk.py. */
#include "oc_silk.h"

#define DD OC_SILK_DECISION_DELAY
#define QUANT_LEVEL_ADJUST_Q10 80
static const s16 kQuantOffsets_Q10[2][2] = { { 100, 240 }, { 32, 100 } };  /* silk/tables_other.c:77 */

typedef struct {                              /* one survivor (NSQ_del_dec.c:37-50) */
   s32 sLPC_Q14[OC_SILK_MAX_SUBFR + OC_SILK_LPC_BUF];
   s32 RandState[DD], Q_Q10[DD], Xq_Q14[DD], Pred_Q15[DD], Shape_Q14[DD];
   s32 sAR2_Q14[OC_SILK_MAX_SHAPE];
   s32 LF_AR_Q14, Diff_Q14, Seed, SeedInit, RD_Q10;
} Survivor;

typedef struct { s32 Q_Q10, RD_Q10, xq_Q14, LF_AR_Q14, Diff_Q14, sLTP_shp_Q14, LPC_exc_Q14; } Cand;   /* NSQ_del_dec.c:52-60 */

static s32 sub_sat(s32 a, s32 b) { s64 r = (s64)a - b; return r > 2147483647 ? 2147483647 : r < -2147483647 - 1 ? -2147483647 - 1 : (s32)r; }

/* NSQ_del_dec.c:646-746 */
static void dd_scale_states(const OcSilkNsqCfg *cfg, OcSilkNsqState *st, Survivor *sv, const s16 *x16, s32 *x_sc_Q10, const s16 *sLTP,
                            s32 *sLTP_Q15, int subfr, const OcSilkNsqFrame *fr, int decisionDelay)
{
   const int L = oc_cfg_subfr(cfg), mem = oc_cfg_ltp_mem(cfg), K = cfg->nStatesDelayedDecision;
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
         for (int i = st->sLTP_buf_idx - lag - OC_SILK_LTP_ORDER / 2; i < st->sLTP_buf_idx - decisionDelay; i++) sLTP_Q15[i] = q_mulww(adj, sLTP_Q15[i]);
      for (int k = 0; k < K; k++) {
         Survivor *s = &sv[k];
         s->LF_AR_Q14 = q_mulww(adj, s->LF_AR_Q14);
         s->Diff_Q14 = q_mulww(adj, s->Diff_Q14);
         for (int i = 0; i < OC_SILK_LPC_BUF; i++) s->sLPC_Q14[i] = q_mulww(adj, s->sLPC_Q14[i]);
         for (int i = 0; i < OC_SILK_MAX_SHAPE; i++) s->sAR2_Q14[i] = q_mulww(adj, s->sAR2_Q14[i]);
         for (int i = 0; i < DD; i++) { s->Pred_Q15[i] = q_mulww(adj, s->Pred_Q15[i]); s->Shape_Q14[i] = q_mulww(adj, s->Shape_Q14[i]); }
      }
      st->prev_gain_Q16 = gain;
   }
}

/* winner = first survivor with the smallest accumulated cost */
static int best_survivor(const Survivor *sv, int K)
{
   int w = 0;
   for (int k = 1; k < K; k++) if (sv[k].RD_Q10 < sv[w].RD_Q10) w = k;
   return w;
}

/* NSQ_del_dec.c:315-644 */
static void dd_subframe(const OcSilkNsqCfg *cfg, OcSilkNsqState *st, Survivor *sv, int signalType, const s32 *x_Q10, s8 *pulses, s16 *xq,
                        s32 *sLTP_Q15, s32 *delayedGain_Q10, const s16 *a_Q12, const s16 *b_Q14, const s16 *AR_shp_Q13, int lag,
                        s32 HarmPacked_Q14, int Tilt_Q14, s32 LF_shp_Q14, s32 Gain_Q16, int Lambda_Q10, int offset_Q10, int subfr,
                        int *smpl_buf_idx, int decisionDelay)
{
   const int L = oc_cfg_subfr(cfg), P = cfg->predictLPCOrder, S = cfg->shapingLPCOrder, K = cfg->nStatesDelayedDecision;
   const s32 warp = cfg->warping_Q16, Gain_Q10 = Gain_Q16 >> 6;
   s32 *shp_lag = &st->sLTP_shp_Q14[st->sLTP_shp_buf_idx - lag + 1];
   s32 *pred_lag = &sLTP_Q15[st->sLTP_buf_idx - lag + OC_SILK_LTP_ORDER / 2];
   Cand cand[OC_SILK_MAX_DEL_DEC][2];

   for (int i = 0; i < L; i++) {
      /* terms common to all survivors */
      s32 LTP_pred_Q14 = 0, n_LTP_Q14 = 0;
      if (signalType == OC_SILK_TYPE_VOICED) {
         LTP_pred_Q14 = 2;
         for (int k = 0; k < OC_SILK_LTP_ORDER; k++) LTP_pred_Q14 = q_mlawb(LTP_pred_Q14, pred_lag[-k], b_Q14[k]);
         LTP_pred_Q14 = q_shlw(LTP_pred_Q14, 1);
         pred_lag++;
      }
      if (lag > 0) {
         n_LTP_Q14 = q_mulwb(q_add_sat(shp_lag[0], shp_lag[-2]), HarmPacked_Q14);
         n_LTP_Q14 = q_mlawt(n_LTP_Q14, shp_lag[-1], HarmPacked_Q14);
         n_LTP_Q14 = LTP_pred_Q14 - q_shlw(n_LTP_Q14, 2);
         shp_lag++;
      }

      for (int k = 0; k < K; k++) {
         Survivor *s = &sv[k];
         Cand *c = cand[k];
         s->Seed = q_rand(s->Seed);
         const s32 *lpc = &s->sLPC_Q14[OC_SILK_LPC_BUF - 1 + i];
         s32 LPC_pred_Q14 = P >> 1;
         for (int j = 0; j < P; j++) LPC_pred_Q14 = q_mlawb(LPC_pred_Q14, lpc[-j], a_Q12[j]);
         LPC_pred_Q14 = q_shlw(LPC_pred_Q14, 4);

         /* warped AR shaping: a chain of first-order allpass sections; section j's new memory is the previous section's output */
         s32 n_AR_Q14 = S >> 1;
         {
            s32 in = q_mlawb(s->Diff_Q14, s->sAR2_Q14[0], warp);                  /* lowpass section output */
            for (int j = 0; j < S; j++) {
               s32 out = j + 1 < S ? q_mlawb(s->sAR2_Q14[j], q_subw(s->sAR2_Q14[j + 1], in), warp) : 0;
               s->sAR2_Q14[j] = in;
               n_AR_Q14 = q_mlawb(n_AR_Q14, in, AR_shp_Q13[j]);
               in = out;
            }
         }
         n_AR_Q14 = q_shlw(n_AR_Q14, 1);
         n_AR_Q14 = q_mlawb(n_AR_Q14, s->LF_AR_Q14, Tilt_Q14);
         n_AR_Q14 = q_shlw(n_AR_Q14, 2);

         s32 n_LF_Q14 = q_mulwb(s->Shape_Q14[*smpl_buf_idx], LF_shp_Q14);
         n_LF_Q14 = q_mlawt(n_LF_Q14, s->LF_AR_Q14, LF_shp_Q14);
         n_LF_Q14 = q_shlw(n_LF_Q14, 2);

         s32 t1 = q_add_sat(n_AR_Q14, n_LF_Q14);
         s32 t2 = q_addw(n_LTP_Q14, LPC_pred_Q14);
         t1 = sub_sat(t2, t1);
         t1 = q_rshift_round(t1, 4);

         s32 r_Q10 = x_Q10[i] - t1;
         if (s->Seed < 0) r_Q10 = q_subw(0, r_Q10);
         r_Q10 = q_limit(r_Q10, -(31 << 10), 30 << 10);

         s32 q1_Q10 = r_Q10 - offset_Q10;
         s32 q1_Q0 = q1_Q10 >> 10;
         if (Lambda_Q10 > 2048) {
            int rdo_offset = Lambda_Q10 / 2 - 512;
            if (q1_Q10 > rdo_offset) q1_Q0 = (q1_Q10 - rdo_offset) >> 10;
            else if (q1_Q10 < -rdo_offset) q1_Q0 = (q1_Q10 + rdo_offset) >> 10;
            else q1_Q0 = q1_Q10 < 0 ? -1 : 0;
         }
         s32 q2_Q10, rd1, rd2;
         if (q1_Q0 > 0) {
            q1_Q10 = (q1_Q0 << 10) - QUANT_LEVEL_ADJUST_Q10 + offset_Q10;
            q2_Q10 = q1_Q10 + 1024;
            rd1 = q_mulbb(q1_Q10, Lambda_Q10); rd2 = q_mulbb(q2_Q10, Lambda_Q10);
         } else if (q1_Q0 == 0) {
            q1_Q10 = offset_Q10;
            q2_Q10 = q1_Q10 + (1024 - QUANT_LEVEL_ADJUST_Q10);
            rd1 = q_mulbb(q1_Q10, Lambda_Q10); rd2 = q_mulbb(q2_Q10, Lambda_Q10);
         } else if (q1_Q0 == -1) {
            q2_Q10 = offset_Q10;
            q1_Q10 = q2_Q10 - (1024 - QUANT_LEVEL_ADJUST_Q10);
            rd1 = q_mulbb(-q1_Q10, Lambda_Q10); rd2 = q_mulbb(q2_Q10, Lambda_Q10);
         } else {
            q1_Q10 = q_shlw(q1_Q0, 10) + QUANT_LEVEL_ADJUST_Q10 + offset_Q10;
            q2_Q10 = q1_Q10 + 1024;
            rd1 = q_mulbb(-q1_Q10, Lambda_Q10); rd2 = q_mulbb(-q2_Q10, Lambda_Q10);
         }
         s32 rr = r_Q10 - q1_Q10;
         rd1 = q_mlabb(rd1, rr, rr) >> 10;
         rr = r_Q10 - q2_Q10;
         rd2 = q_mlabb(rd2, rr, rr) >> 10;

         const int first_is_q1 = rd1 < rd2;
         c[0].Q_Q10 = first_is_q1 ? q1_Q10 : q2_Q10;  c[0].RD_Q10 = s->RD_Q10 + (first_is_q1 ? rd1 : rd2);
         c[1].Q_Q10 = first_is_q1 ? q2_Q10 : q1_Q10;  c[1].RD_Q10 = s->RD_Q10 + (first_is_q1 ? rd2 : rd1);
         for (int b = 0; b < 2; b++) {
            s32 exc_Q14 = q_shlw(c[b].Q_Q10, 4);
            if (s->Seed < 0) exc_Q14 = -exc_Q14;
            s32 LPC_exc_Q14 = exc_Q14 + LTP_pred_Q14;
            s32 xq_Q14 = q_addw(LPC_exc_Q14, LPC_pred_Q14);
            c[b].Diff_Q14 = q_subw(xq_Q14, q_shlw(x_Q10[i], 4));
            s32 sLF = q_subw(c[b].Diff_Q14, n_AR_Q14);
            c[b].sLTP_shp_Q14 = sub_sat(sLF, n_LF_Q14);
            c[b].LF_AR_Q14 = sLF;
            c[b].LPC_exc_Q14 = LPC_exc_Q14;
            c[b].xq_Q14 = xq_Q14;
         }
      }

      *smpl_buf_idx = (*smpl_buf_idx + DD - 1) % DD;
      const int last = (*smpl_buf_idx + decisionDelay) % DD;

      int winner = 0;
      for (int k = 1; k < K; k++) if (cand[k][0].RD_Q10 < cand[winner][0].RD_Q10) winner = k;

      /* survivors whose history at the commit point disagrees with the winner's can no longer win */
      const s32 wrand = sv[winner].RandState[last];
      for (int k = 0; k < K; k++)
         if (sv[k].RandState[last] != wrand) { cand[k][0].RD_Q10 += 2147483647 >> 4; cand[k][1].RD_Q10 += 2147483647 >> 4; }

      int worst = 0, best2 = 0;
      for (int k = 1; k < K; k++) {
         if (cand[k][0].RD_Q10 > cand[worst][0].RD_Q10) worst = k;
         if (cand[k][1].RD_Q10 < cand[best2][1].RD_Q10) best2 = k;
      }
      if (cand[best2][1].RD_Q10 < cand[worst][0].RD_Q10) {
         /* (the reference copies from word i on: sLPC_Q14[0..i-1] is dead by now, so a whole-state copy is equivalent
          *  except for those dead words, which nothing reads again) */
         memcpy((s32 *)&sv[worst] + i, (s32 *)&sv[best2] + i, sizeof(Survivor) - (size_t)i * sizeof(s32));
         cand[worst][0] = cand[best2][1];
      }

      const Survivor *w = &sv[winner];
      if (subfr > 0 || i >= decisionDelay) {
         pulses[i - decisionDelay] = (s8)q_rshift_round(w->Q_Q10[last], 10);
         xq[i - decisionDelay] = (s16)q_sat16(q_rshift_round(q_mulww(w->Xq_Q14[last], delayedGain_Q10[last]), 8));
         st->sLTP_shp_Q14[st->sLTP_shp_buf_idx - decisionDelay] = w->Shape_Q14[last];
         sLTP_Q15[st->sLTP_buf_idx - decisionDelay] = w->Pred_Q15[last];
      }
      st->sLTP_shp_buf_idx++;
      st->sLTP_buf_idx++;

      for (int k = 0; k < K; k++) {
         Survivor *s = &sv[k];
         const Cand *c = &cand[k][0];
         s->LF_AR_Q14 = c->LF_AR_Q14;
         s->Diff_Q14 = c->Diff_Q14;
         s->sLPC_Q14[OC_SILK_LPC_BUF + i] = c->xq_Q14;
         s->Xq_Q14[*smpl_buf_idx] = c->xq_Q14;
         s->Q_Q10[*smpl_buf_idx] = c->Q_Q10;
         s->Pred_Q15[*smpl_buf_idx] = q_shlw(c->LPC_exc_Q14, 1);
         s->Shape_Q14[*smpl_buf_idx] = c->sLTP_shp_Q14;
         s->Seed = q_addw(s->Seed, q_rshift_round(c->Q_Q10, 10));
         s->RandState[*smpl_buf_idx] = s->Seed;
         s->RD_Q10 = c->RD_Q10;
      }
      delayedGain_Q10[*smpl_buf_idx] = Gain_Q10;
   }
   for (int k = 0; k < K; k++) memcpy(sv[k].sLPC_Q14, &sv[k].sLPC_Q14[L], OC_SILK_LPC_BUF * sizeof(s32));
}

/* commit the last `decisionDelay` undecided samples from the winner's ring (NSQ_del_dec.c:214-224 and :288-297) */
static void dd_flush(OcSilkNsqState *st, const Survivor *w, int smpl_buf_idx, int decisionDelay, s8 *pulses, s16 *pxq, s32 gain, int shift)
{
   int last = smpl_buf_idx + decisionDelay;
   for (int i = 0; i < decisionDelay; i++) {
      last = (last + DD - 1) % DD;
      pulses[i - decisionDelay] = (s8)q_rshift_round(w->Q_Q10[last], 10);
      pxq[i - decisionDelay] = (s16)q_sat16(q_rshift_round(q_mulww(w->Xq_Q14[last], gain), shift));
      st->sLTP_shp_Q14[st->sLTP_shp_buf_idx - decisionDelay + i] = w->Shape_Q14[last];
   }
}

void oc_silk_nsq_del_dec(const OcSilkNsqCfg *cfg, OcSilkNsqState *st, OcSilkNsqFrame *fr, const s16 *x16, s8 *pulses)
{
   const int L = oc_cfg_subfr(cfg), mem = oc_cfg_ltp_mem(cfg), frame = oc_cfg_frame(cfg), K = cfg->nStatesDelayedDecision;
   s32 sLTP_Q15[2 * OC_SILK_MAX_FRAME];
   s16 sLTP[2 * OC_SILK_MAX_FRAME];
   s32 x_sc_Q10[OC_SILK_MAX_SUBFR];
   s32 delayedGain_Q10[DD];
   Survivor sv[OC_SILK_MAX_DEL_DEC];

   int lag = st->lagPrev;
   memset(sv, 0, sizeof sv);
   for (int k = 0; k < K; k++) {
      Survivor *s = &sv[k];
      s->Seed = (k + fr->Seed) & 3;
      s->SeedInit = s->Seed;
      s->LF_AR_Q14 = st->sLF_AR_shp_Q14;
      s->Diff_Q14 = st->sDiff_shp_Q14;
      s->Shape_Q14[0] = st->sLTP_shp_Q14[mem - 1];
      memcpy(s->sLPC_Q14, st->sLPC_Q14, OC_SILK_LPC_BUF * sizeof(s32));
      memcpy(s->sAR2_Q14, st->sAR2_Q14, sizeof s->sAR2_Q14);
   }
   const int offset_Q10 = kQuantOffsets_Q10[fr->signalType >> 1][fr->quantOffsetType];
   int smpl_buf_idx = 0;
   int decisionDelay = DD < L ? DD : L;
   if (fr->signalType == OC_SILK_TYPE_VOICED) {
      for (int k = 0; k < cfg->nb_subfr; k++) if (fr->pitchL[k] - OC_SILK_LTP_ORDER / 2 - 1 < decisionDelay) decisionDelay = fr->pitchL[k] - OC_SILK_LTP_ORDER / 2 - 1;
   } else if (lag > 0) {
      if (lag - OC_SILK_LTP_ORDER / 2 - 1 < decisionDelay) decisionDelay = lag - OC_SILK_LTP_ORDER / 2 - 1;
   }
   const int interp = fr->NLSFInterpCoef_Q2 == 4 ? 0 : 1;
   s16 *pxq = &st->xq[mem];
   st->sLTP_shp_buf_idx = mem;
   st->sLTP_buf_idx = mem;
   int subfr = 0;
   for (int k = 0; k < cfg->nb_subfr; k++) {
      const s16 *A_Q12 = &fr->PredCoef_Q12[((k >> 1) | (1 - interp)) * 16];
      const s16 *B_Q14 = &fr->LTPCoef_Q14[k * OC_SILK_LTP_ORDER];
      const s16 *AR_Q13 = &fr->AR_Q13[k * OC_SILK_MAX_SHAPE];
      s32 harm = (fr->HarmShapeGain_Q14[k] >> 2) | (s32)((u32)(fr->HarmShapeGain_Q14[k] >> 1) << 16);
      st->rewhite_flag = 0;
      if (fr->signalType == OC_SILK_TYPE_VOICED) {
         lag = fr->pitchL[k];
         if ((k & (3 - (interp << 1))) == 0) {
            if (k == 2) {
               /* mid-frame LPC switch: decide everything pending with the current winner, demote the others */
               int w = best_survivor(sv, K);
               for (int i = 0; i < K; i++) if (i != w) sv[i].RD_Q10 += 2147483647 >> 4;
               dd_flush(st, &sv[w], smpl_buf_idx, decisionDelay, pulses, pxq, fr->Gains_Q16[1], 14);
               subfr = 0;
            }
            int start = mem - lag - cfg->predictLPCOrder - OC_SILK_LTP_ORDER / 2;
            oc_silk_lpc_analysis_filter(&sLTP[start], &st->xq[start + k * L], A_Q12, mem - start, cfg->predictLPCOrder);
            st->sLTP_buf_idx = mem;
            st->rewhite_flag = 1;
         }
      }
      dd_scale_states(cfg, st, sv, x16, x_sc_Q10, sLTP, sLTP_Q15, k, fr, decisionDelay);
      dd_subframe(cfg, st, sv, fr->signalType, x_sc_Q10, pulses, pxq, sLTP_Q15, delayedGain_Q10, A_Q12, B_Q14, AR_Q13, lag, harm,
                  fr->Tilt_Q14[k], fr->LF_shp_Q14[k], fr->Gains_Q16[k], fr->Lambda_Q10, offset_Q10, subfr++, &smpl_buf_idx, decisionDelay);
      x16 += L; pulses += L; pxq += L;
   }
   int w = best_survivor(sv, K);
   fr->Seed = (s8)sv[w].SeedInit;
   dd_flush(st, &sv[w], smpl_buf_idx, decisionDelay, pulses, pxq, fr->Gains_Q16[cfg->nb_subfr - 1] >> 6, 8);
   memcpy(st->sLPC_Q14, &sv[w].sLPC_Q14[L], OC_SILK_LPC_BUF * sizeof(s32));
   memcpy(st->sAR2_Q14, sv[w].sAR2_Q14, sizeof st->sAR2_Q14);
   st->sLF_AR_shp_Q14 = sv[w].LF_AR_Q14;
   st->sDiff_shp_Q14 = sv[w].Diff_Q14;
   st->lagPrev = fr->pitchL[cfg->nb_subfr - 1];
   memmove(st->xq, &st->xq[frame], (size_t)mem * sizeof(s16));
   memmove(st->sLTP_shp_Q14, &st->sLTP_shp_Q14[frame], (size_t)mem * sizeof(s32));
}

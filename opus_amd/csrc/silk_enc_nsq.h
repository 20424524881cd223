/* silk_enc_nsq.h — the noise-shaping quantisers inside the SILK encoder's wave (row a21 of SURVEY §8, frame-level use).
 *
 *   se_nsq_l0            silk_NSQ_c          silk/NSQ.c:76-181, sample loop :183-366, state scaling :368-436, tap loops silk/NSQ.h:35-96
 *   se_nsq_del_dec_wave  silk_NSQ_del_dec_c  silk/NSQ_del_dec.c:114-312, sample loop :315-644, state scaling :646-746
 * (the batched lane-per-stream / quad-per-stream kernels of silk_nsq.h / silk_nsq_dd.h are the standalone operators; here the quantiser runs inside
 * the wave that owns the stream: survivor k of the delayed-decision search lives on lane k, the candidate exchange goes through LDS and a survivor
 * replacement is a 64-lane copy of 1.3 KB instead of a one-thread memcpy). */
#ifndef OPUS_AMD_SILK_ENC_NSQ_H
#define OPUS_AMD_SILK_ENC_NSQ_H

#define SE_DD 40
#define SE_QLA 80                                                                   /* QUANT_LEVEL_ADJUST_Q10 */
struct SeSurvivor { i32 RandState[SE_DD], Q_Q10[SE_DD], Xq_Q14[SE_DD], Pred_Q15[SE_DD], Shape_Q14[SE_DD]; i32 SeedInit, RD_Q10, pad[2]; };   /* the LDS part of a survivor: rings of undecided samples */
struct SeCand { i32 Q_Q10, RD_Q10, xq_Q14, LF_AR_Q14, Diff_Q14, sLTP_shp_Q14, LPC_exc_Q14, pad; };
struct SeNsqLds {
   i32 sLTP_Q15[2 * SE_MAX_FRAME];
   i32 x_sc_Q10[80], delayedGain_Q10[SE_DD];
   i32 sLPC[80 + 16];                                                                /* silk_NSQ_c working LPC state */
   i16 sLTP[2 * SE_MAX_FRAME];
   SeSurvivor sv[4];
};

/* 2-level quantisation of one residual sample (NSQ.c:268-321 / NSQ_del_dec.c:440-493): candidates and their rate terms */
WV_DEV void se_nsq_levels(i32 r_Q10, int offset_Q10, int Lambda_Q10, i32 &q1_Q10, i32 &q2_Q10, i32 &rd1, i32 &rd2)
{
   q1_Q10 = r_Q10 - offset_Q10;
   i32 q1_Q0 = q1_Q10 >> 10;
   if (Lambda_Q10 > 2048) {
      const int rdo_offset = Lambda_Q10 / 2 - 512;
      if (q1_Q10 > rdo_offset) q1_Q0 = (q1_Q10 - rdo_offset) >> 10; else if (q1_Q10 < -rdo_offset) q1_Q0 = (q1_Q10 + rdo_offset) >> 10; else q1_Q0 = q1_Q10 < 0 ? -1 : 0;
   }
   if (q1_Q0 > 0) { q1_Q10 = (q1_Q0 << 10) - SE_QLA + offset_Q10; q2_Q10 = q1_Q10 + 1024; rd1 = sk_mulbb(q1_Q10, Lambda_Q10); rd2 = sk_mulbb(q2_Q10, Lambda_Q10); }
   else if (q1_Q0 == 0) { q1_Q10 = offset_Q10; q2_Q10 = q1_Q10 + (1024 - SE_QLA); rd1 = sk_mulbb(q1_Q10, Lambda_Q10); rd2 = sk_mulbb(q2_Q10, Lambda_Q10); }
   else if (q1_Q0 == -1) { q2_Q10 = offset_Q10; q1_Q10 = q2_Q10 - (1024 - SE_QLA); rd1 = sk_mulbb(-q1_Q10, Lambda_Q10); rd2 = sk_mulbb(q2_Q10, Lambda_Q10); }
   else { q1_Q10 = shl32(q1_Q0, 10) + SE_QLA + offset_Q10; q2_Q10 = q1_Q10 + 1024; rd1 = sk_mulbb(-q1_Q10, Lambda_Q10); rd2 = sk_mulbb(-q2_Q10, Lambda_Q10); }
   i32 rr = r_Q10 - q1_Q10; rd1 = sk_mlabb(rd1, rr, rr);
   rr = r_Q10 - q2_Q10; rd2 = sk_mlabb(rd2, rr, rr);
}

/* whitening of the output history with the current LPC (silk_LPC_analysis_filter): lanes */
WV_DEV void se_nsq_rewhiten_wave(WV_LDS i16 *sLTP, const WV_LDS i16 *xq, int start, int src, int len, const WV_LDS i16 *A_Q12, int P)
{ se_lpc_analysis_filter_wave(&sLTP[start], &xq[src], A_Q12, len, P); wv_sync(); }

/* ---------------- silk_NSQ_c on lane 0 (complexity 0-1: one state, no warping) ---------------- */
WV_DEV void se_nsq_scale_states_l0(WV_LDS OaSilkEncChannel *c, WV_LDS OaSilkNsqState *st, WV_LDS SeNsqLds *N, const WV_LDS SeEncCtrl *ctl, const WV_LDS i16 *x16, int subfr, int LTP_scale_Q14, int signalType)
{
   const int L = c->subfr_length, mem = c->ltp_mem_length, lag = ctl->pitchL[subfr];
   const i32 gain = ctl->Gains_Q16[subfr];
   i32 inv_gain_Q31 = sk_inverse32_varQ(gain > 1 ? gain : 1, 47);
   const i32 inv_gain_Q26 = sk_rround(inv_gain_Q31, 5);
   for (int i = 0; i < L; i++) N->x_sc_Q10[i] = sk_mulww(x16[i], inv_gain_Q26);
   if (st->rewhite_flag) {
      if (subfr == 0) inv_gain_Q31 = shl32(sk_mulwb(inv_gain_Q31, LTP_scale_Q14), 2);
      for (int i = st->sLTP_buf_idx - lag - 5 / 2; i < st->sLTP_buf_idx; i++) N->sLTP_Q15[i] = sk_mulwb(inv_gain_Q31, N->sLTP[i]);
   }
   if (gain != st->prev_gain_Q16) {
      const i32 adj = sk_div32_varQ(st->prev_gain_Q16, gain, 16);
      for (int i = st->sLTP_shp_buf_idx - mem; i < st->sLTP_shp_buf_idx; i++) st->sLTP_shp_Q14[i] = sk_mulww(adj, st->sLTP_shp_Q14[i]);
      if (signalType == SE_TYPE_VOICED && !st->rewhite_flag) for (int i = st->sLTP_buf_idx - lag - 5 / 2; i < st->sLTP_buf_idx; i++) N->sLTP_Q15[i] = sk_mulww(adj, N->sLTP_Q15[i]);
      st->sLF_AR_shp_Q14 = sk_mulww(adj, st->sLF_AR_shp_Q14);
      st->sDiff_shp_Q14 = sk_mulww(adj, st->sDiff_shp_Q14);
      for (int i = 0; i < 16; i++) N->sLPC[i] = sk_mulww(adj, N->sLPC[i]);
      for (int i = 0; i < 24; i++) st->sAR2_Q14[i] = sk_mulww(adj, st->sAR2_Q14[i]);
      st->prev_gain_Q16 = gain;
   }
}
WV_DEV void se_nsq_subframe_l0(WV_LDS OaSilkEncChannel *c, WV_LDS OaSilkNsqState *st, WV_LDS SeNsqLds *N, int signalType, WV_LDS i8 *pulses, WV_LDS i16 *xq, const WV_LDS i16 *a_Q12, const WV_LDS i16 *b_Q14,
      const WV_LDS i16 *AR_shp_Q13, int lag, i32 HarmPacked_Q14, int Tilt_Q14, i32 LF_shp_Q14, i32 Gain_Q16, int Lambda_Q10, int offset_Q10)
{
   const int L = c->subfr_length, P = c->predictLPCOrder, S = c->shapingLPCOrder;
   const i32 Gain_Q10 = Gain_Q16 >> 6;
   int shp_lag = st->sLTP_shp_buf_idx - lag + 1, pred_lag = st->sLTP_buf_idx - lag + 5 / 2, lpc = 16 - 1;
   for (int i = 0; i < L; i++) {
      st->rand_seed = sk_rand(st->rand_seed);
      i32 LPC_pred_Q10 = P >> 1;
      for (int k = 0; k < P; k++) LPC_pred_Q10 = sk_mlawb(LPC_pred_Q10, N->sLPC[lpc - k], a_Q12[k]);
      i32 LTP_pred_Q13 = 0;
      if (signalType == SE_TYPE_VOICED) { LTP_pred_Q13 = 2; for (int k = 0; k < 5; k++) LTP_pred_Q13 = sk_mlawb(LTP_pred_Q13, N->sLTP_Q15[pred_lag - k], b_Q14[k]); pred_lag++; }
      i32 n_AR_Q12 = S >> 1;
      {
         i32 carry = st->sDiff_shp_Q14;
         for (int k = 0; k < S; k++) { const i32 old = st->sAR2_Q14[k]; st->sAR2_Q14[k] = carry; n_AR_Q12 = sk_mlawb(n_AR_Q12, carry, AR_shp_Q13[k]); carry = old; }
         n_AR_Q12 = shl32(n_AR_Q12, 1);
      }
      n_AR_Q12 = sk_mlawb(n_AR_Q12, st->sLF_AR_shp_Q14, Tilt_Q14);
      i32 n_LF_Q12 = sk_mulwb(st->sLTP_shp_Q14[st->sLTP_shp_buf_idx - 1], LF_shp_Q14);
      n_LF_Q12 = sk_mlawt(n_LF_Q12, st->sLF_AR_shp_Q14, LF_shp_Q14);
      i32 t1 = sub32(shl32(LPC_pred_Q10, 2), n_AR_Q12);
      t1 = sub32(t1, n_LF_Q12);
      if (lag > 0) {
         i32 n_LTP_Q13 = sk_mulwb(sk_add_sat(st->sLTP_shp_Q14[shp_lag], st->sLTP_shp_Q14[shp_lag - 2]), HarmPacked_Q14);
         n_LTP_Q13 = sk_mlawt(n_LTP_Q13, st->sLTP_shp_Q14[shp_lag - 1], HarmPacked_Q14);
         n_LTP_Q13 = shl32(n_LTP_Q13, 1);
         shp_lag++;
         const i32 t2 = sub32(LTP_pred_Q13, n_LTP_Q13);
         t1 = add32(t2, shl32(t1, 1));
         t1 = sk_rround(t1, 3);
      } else t1 = sk_rround(t1, 2);
      i32 r_Q10 = sub32(N->x_sc_Q10[i], t1);
      if (st->rand_seed < 0) r_Q10 = neg32(r_Q10);
      r_Q10 = se_limit(r_Q10, -(31 << 10), 30 << 10);
      i32 q1_Q10, q2_Q10, rd1, rd2;
      se_nsq_levels(r_Q10, offset_Q10, Lambda_Q10, q1_Q10, q2_Q10, rd1, rd2);
      if (rd2 < rd1) q1_Q10 = q2_Q10;
      pulses[i] = (i8)sk_rround(q1_Q10, 10);
      i32 exc_Q14 = shl32(q1_Q10, 4);
      if (st->rand_seed < 0) exc_Q14 = -exc_Q14;
      const i32 LPC_exc_Q14 = exc_Q14 + shl32(LTP_pred_Q13, 1);
      const i32 xq_Q14 = add32(LPC_exc_Q14, shl32(LPC_pred_Q10, 4));
      xq[i] = (i16)sk_sat16(sk_rround(sk_mulww(xq_Q14, Gain_Q10), 8));
      lpc++;
      N->sLPC[lpc] = xq_Q14;
      st->sDiff_shp_Q14 = sub32(xq_Q14, shl32(N->x_sc_Q10[i], 4));
      const i32 sLF = sub32(st->sDiff_shp_Q14, shl32(n_AR_Q12, 2));
      st->sLF_AR_shp_Q14 = sLF;
      st->sLTP_shp_Q14[st->sLTP_shp_buf_idx++] = sub32(sLF, shl32(n_LF_Q12, 2));
      N->sLTP_Q15[st->sLTP_buf_idx++] = shl32(LPC_exc_Q14, 1);
      st->rand_seed = add32(st->rand_seed, pulses[i]);
   }
   for (int i = 0; i < 16; i++) N->sLPC[i] = N->sLPC[L + i];
}
WV_DEV i32 se_harm_packed(i32 h) { return (h >> 2) | (i32)((u32)(h >> 1) << 16); }

/* indices->Seed is read (and, for delayed decision, rewritten); pulses: frame_length */
WV_DEVN void se_nsq_wave(WV_LDS OaSilkEncChannel *c, WV_LDS OaSilkNsqState *st, WV_LDS OaSilkEncIndices *ix, WV_LDS SeNsqLds *N, const WV_LDS SeEncCtrl *ctl, const WV_LDS i16 *x16, WV_LDS i8 *pulses)
{
   const int L = c->subfr_length, mem = c->ltp_mem_length, frame = c->frame_length, P = c->predictLPCOrder, signalType = ix->signalType;
   const int offset_Q10 = se_quantization_offsets_q10[(signalType >> 1) * 2 + ix->quantOffsetType];
   const int interp = ix->NLSFInterpCoef_Q2 == 4 ? 0 : 1;
   int lag = st->lagPrev;
   LANE0 { st->rand_seed = ix->Seed; st->sLTP_shp_buf_idx = mem; st->sLTP_buf_idx = mem; for (int i = 0; i < 16; i++) N->sLPC[i] = st->sLPC_Q14[i]; }
   for (int k = 0; k < c->nb_subfr; k++) {
      const WV_LDS i16 *A_Q12 = &ctl->PredCoef_Q12[(k >> 1) | (1 - interp)][0];
      LANE0 st->rewhite_flag = 0;
      if (signalType == SE_TYPE_VOICED) {
         lag = ctl->pitchL[k];
         if ((k & (3 - (interp << 1))) == 0) {
            const int start = mem - lag - P - 5 / 2;
            se_nsq_rewhiten_wave(N->sLTP, st->xq, start, start + k * L, mem - start, A_Q12, P);
            LANE0 { st->rewhite_flag = 1; st->sLTP_buf_idx = mem; }
         }
      }
      LANE0 {
         se_nsq_scale_states_l0(c, st, N, ctl, x16 + k * L, k, ctl->LTP_scale_Q14, signalType);
         se_nsq_subframe_l0(c, st, N, signalType, pulses + k * L, &st->xq[mem + k * L], A_Q12, &ctl->LTPCoef_Q14[k * 5], &ctl->AR_Q13[k * SE_MAX_SHAPE_ORDER], lag, se_harm_packed(ctl->HarmShapeGain_Q14[k]),
               ctl->Tilt_Q14[k], ctl->LF_shp_Q14[k], ctl->Gains_Q16[k], ctl->Lambda_Q10, offset_Q10);
      }
   }
   LANE0 { st->lagPrev = ctl->pitchL[c->nb_subfr - 1]; for (int i = 0; i < 16; i++) st->sLPC_Q14[i] = N->sLPC[i]; }
   /* history shift (NSQ.c:176-178): lanes, through registers so the overlapping move is safe */
   for (int b = 0; b < mem; b += WV_WIDTH) {
      const int i = b + wv_lane(); i16 a = 0; i32 s = 0;
      if (i < mem) { a = st->xq[frame + i]; s = st->sLTP_shp_Q14[frame + i]; }
      wv_sync();
      if (i < mem) { st->xq[i] = a; st->sLTP_shp_Q14[i] = s; }
      wv_sync();
   }
}

/* ---------------- silk_NSQ_del_dec_c: survivor k on lane k ---------------- */
WV_DEV int se_dd_best(const WV_LDS SeSurvivor *sv, int K) { int w = 0; for (int k = 1; k < K; k++) if (sv[k].RD_Q10 < sv[w].RD_Q10) w = k; return w; }
/* commit the last decisionDelay undecided samples of survivor w (NSQ_del_dec.c:214-224, :288-297): lane 0 */
WV_DEV void se_dd_flush_l0(WV_LDS OaSilkNsqState *st, const WV_LDS SeSurvivor *w, int smpl_buf_idx, int decisionDelay, WV_LDS i8 *pulses, WV_LDS i16 *pxq, i32 gain, int shift)
{
   int last = smpl_buf_idx + decisionDelay;
   for (int i = 0; i < decisionDelay; i++) {
      last = (last + SE_DD - 1) % SE_DD;
      pulses[i - decisionDelay] = (i8)sk_rround(w->Q_Q10[last], 10);
      pxq[i - decisionDelay] = (i16)sk_sat16(sk_rround(sk_mulww(w->Xq_Q14[last], gain), shift));
      st->sLTP_shp_Q14[st->sLTP_shp_buf_idx - decisionDelay + i] = w->Shape_Q14[last];
   }
}
/* Register-resident version: survivor k on lane k keeps its 16-sample LPC window, its <= 24 shaping-filter memories and both coefficient sets in VGPRs
 * (fixed-size arrays indexed only by unrolled constants; taps beyond the order carry zero coefficients / are predicated off), so a sample costs no LDS
 * round trip for the filters.  The five 40-deep rings of undecided samples stay in LDS.  A survivor replacement moves the 40 filter words of the better
 * lane with v_readlane and the rings with a 64-lane copy. */
/* survivor J of the quad hands its second-choice candidate and its filter state to the replaced survivor (:566-575): DPP quad broadcasts, no LDS */
template <int J> WV_DEV void se_dd_take(bool me, const SeCand &c1, i32 &nQ, i32 &nxq, i32 &nLF, i32 &nDiff, i32 &nShp, i32 &nExc, i32 &Seed, i32 *w, i32 *sar)
{
   i32 t;
   t = wv_quad_bcast<J>(c1.Q_Q10); nQ = me ? t : nQ;
   t = wv_quad_bcast<J>(c1.xq_Q14); nxq = me ? t : nxq;
   t = wv_quad_bcast<J>(c1.LF_AR_Q14); nLF = me ? t : nLF;
   t = wv_quad_bcast<J>(c1.Diff_Q14); nDiff = me ? t : nDiff;
   t = wv_quad_bcast<J>(c1.sLTP_shp_Q14); nShp = me ? t : nShp;
   t = wv_quad_bcast<J>(c1.LPC_exc_Q14); nExc = me ? t : nExc;
   t = wv_quad_bcast<J>(Seed); Seed = me ? t : Seed;
#pragma unroll
   for (int j = 0; j < 16; j++) { t = wv_quad_bcast<J>(w[j]); w[j] = me ? t : w[j]; }
#pragma unroll
   for (int j = 0; j < 24; j++) { t = wv_quad_bcast<J>(sar[j]); sar[j] = me ? t : sar[j]; }
}
WV_DEVN void se_nsq_del_dec_wave(WV_LDS OaSilkEncChannel *c, WV_LDS OaSilkNsqState *st, WV_LDS OaSilkEncIndices *ix, WV_LDS SeNsqLds *N, const WV_LDS SeEncCtrl *ctl, const WV_LDS i16 *x16, WV_LDS i8 *pulses)
{
   const int lane = wv_lane();
   const int L = c->subfr_length, mem = c->ltp_mem_length, frame = c->frame_length, P = c->predictLPCOrder, S = c->shapingLPCOrder, K = c->nStatesDelayedDecision, signalType = ix->signalType;
   const i32 warp = c->warping_Q16;
   const int offset_Q10 = se_quantization_offsets_q10[(signalType >> 1) * 2 + ix->quantOffsetType];
   const int interp = ix->NLSFInterpCoef_Q2 == 4 ? 0 : 1;
   const int Lambda_Q10 = ctl->Lambda_Q10;
   const bool act = lane < K;
   int lag = st->lagPrev;
   /* survivor state (:143-160): filters in registers, rings in LDS */
   i32 w[16], sar[24];                                     /* w[j] = sLPC_Q14[newest - j] */
   i32 LF_AR = st->sLF_AR_shp_Q14, Diff = st->sDiff_shp_Q14, Seed = (lane + ix->Seed) & 3, RD = 0;
   const i32 SeedInit = Seed;
#pragma unroll
   for (int j = 0; j < 16; j++) w[j] = st->sLPC_Q14[15 - j];
#pragma unroll
   for (int j = 0; j < 24; j++) sar[j] = st->sAR2_Q14[j];
   {
      WV_LDS i32 *z = (WV_LDS i32 *)N->sv;
      FOR_LANES(i, (int)(sizeof(N->sv) / 4)) z[i] = 0;
      wv_sync();
      if (act) { N->sv[lane].Shape_Q14[0] = st->sLTP_shp_Q14[mem - 1]; N->sv[lane].SeedInit = SeedInit; }
      wv_sync();
   }
   int smpl_buf_idx = 0, decisionDelay = imin(SE_DD, L);
   if (signalType == SE_TYPE_VOICED) { for (int k = 0; k < c->nb_subfr; k++) decisionDelay = imin(decisionDelay, ctl->pitchL[k] - 5 / 2 - 1); }
   else if (lag > 0) decisionDelay = imin(decisionDelay, lag - 5 / 2 - 1);
   LANE0 { st->sLTP_shp_buf_idx = mem; st->sLTP_buf_idx = mem; }
   int subfr = 0;
   for (int k = 0; k < c->nb_subfr; k++) {
      const WV_LDS i16 *a_Q12 = &ctl->PredCoef_Q12[(k >> 1) | (1 - interp)][0], *b_Q14 = &ctl->LTPCoef_Q14[k * 5], *AR_shp_Q13 = &ctl->AR_Q13[k * SE_MAX_SHAPE_ORDER];
      WV_LDS i8 *pls = pulses + k * L; WV_LDS i16 *pxq = &st->xq[mem + k * L];
      const WV_LDS i16 *xk = x16 + k * L;
      const i32 HarmPacked_Q14 = se_harm_packed(ctl->HarmShapeGain_Q14[k]), LF_shp_Q14 = ctl->LF_shp_Q14[k], Gain_Q16 = ctl->Gains_Q16[k];
      const int Tilt_Q14 = ctl->Tilt_Q14[k];
      LANE0 st->rewhite_flag = 0;
      if (signalType == SE_TYPE_VOICED) {
         lag = ctl->pitchL[k];
         if ((k & (3 - (interp << 1))) == 0) {
            if (k == 2) {
               wv_sync();
               if (act) N->sv[lane].RD_Q10 = RD;
               wv_sync();
               const int wn = se_dd_best(N->sv, K);
               if (act && lane != wn) RD += 2147483647 >> 4;
               LANE0 se_dd_flush_l0(st, &N->sv[wn], smpl_buf_idx, decisionDelay, pls, pxq, ctl->Gains_Q16[1], 14);
               subfr = 0;
            }
            const int start = mem - lag - P - 5 / 2;
            se_nsq_rewhiten_wave(N->sLTP, st->xq, start, start + k * L, mem - start, a_Q12, P);
            LANE0 { st->sLTP_buf_idx = mem; st->rewhite_flag = 1; }
         }
      }
      {  /* silk_nsq_del_dec_scale_states (:646-746): lanes over the arrays, lane k over survivor k */
         const i32 gain = Gain_Q16;
         i32 inv_gain_Q31 = sk_inverse32_varQ(gain > 1 ? gain : 1, 47);
         const i32 inv_gain_Q26 = sk_rround(inv_gain_Q31, 5);
         const int rewhite = st->rewhite_flag, sLTP_buf_idx = st->sLTP_buf_idx, shp_idx = st->sLTP_shp_buf_idx;
         const i32 prev_gain = st->prev_gain_Q16;
         wv_sync();
         FOR_LANES(i, L) N->x_sc_Q10[i] = sk_mulww(xk[i], inv_gain_Q26);
         if (rewhite) {
            if (k == 0) inv_gain_Q31 = shl32(sk_mulwb(inv_gain_Q31, ctl->LTP_scale_Q14), 2);
            const int i0 = sLTP_buf_idx - lag - 5 / 2;
            FOR_LANES(j, sLTP_buf_idx - i0) N->sLTP_Q15[i0 + j] = sk_mulwb(inv_gain_Q31, N->sLTP[i0 + j]);
         }
         if (gain != prev_gain) {
            const i32 adj = sk_div32_varQ(prev_gain, gain, 16);
            FOR_LANES(j, mem) st->sLTP_shp_Q14[shp_idx - mem + j] = sk_mulww(adj, st->sLTP_shp_Q14[shp_idx - mem + j]);
            if (signalType == SE_TYPE_VOICED && !rewhite) { const int i0 = sLTP_buf_idx - lag - 5 / 2; FOR_LANES(j, sLTP_buf_idx - decisionDelay - i0) N->sLTP_Q15[i0 + j] = sk_mulww(adj, N->sLTP_Q15[i0 + j]); }
            LF_AR = sk_mulww(adj, LF_AR); Diff = sk_mulww(adj, Diff);
#pragma unroll
            for (int j = 0; j < 16; j++) w[j] = sk_mulww(adj, w[j]);
#pragma unroll
            for (int j = 0; j < 24; j++) sar[j] = sk_mulww(adj, sar[j]);
            if (act) { WV_LDS SeSurvivor *s = &N->sv[lane]; for (int i = 0; i < SE_DD; i++) { s->Pred_Q15[i] = sk_mulww(adj, s->Pred_Q15[i]); s->Shape_Q14[i] = sk_mulww(adj, s->Shape_Q14[i]); } }
            LANE0 st->prev_gain_Q16 = gain;
         }
         wv_sync();
      }
      /* coefficient sets of the subframe in registers; zero beyond the order (unconditional loads: the arrays are 16 / 24 wide) */
      i32 ca[16], cs[24];
#pragma unroll
      for (int j = 0; j < 16; j++) { const i32 v = a_Q12[j]; ca[j] = j < P ? v : 0; }
#pragma unroll
      for (int j = 0; j < 24; j++) { const i32 v = AR_shp_Q13[j]; cs[j] = j < S ? v : 0; }
      const i32 b0 = b_Q14[0], b1 = b_Q14[1], b2 = b_Q14[2], b3 = b_Q14[3], b4 = b_Q14[4];
      /* sample loop (:315-644).  Every lane runs the arithmetic (lanes >= K on survivor 0's data, results unused); stores are guarded. */
      const i32 Gain_Q10 = Gain_Q16 >> 6;
      int shp_lag = st->sLTP_shp_buf_idx - lag + 1, pred_lag = st->sLTP_buf_idx - lag + 5 / 2;
      int shp_buf_idx = st->sLTP_shp_buf_idx, ltp_buf_idx = st->sLTP_buf_idx;
      WV_LDS SeSurvivor *const sv = &N->sv[act ? lane : 0];
      const bool voiced = signalType == SE_TYPE_VOICED;
      for (int i = 0; i < L; i++) {
         const int old_idx = smpl_buf_idx;
         smpl_buf_idx = (smpl_buf_idx + SE_DD - 1) % SE_DD;
         const int last = (smpl_buf_idx + decisionDelay) % SE_DD;
         /* the ring entries this sample may commit, read ahead of the arithmetic that hides their latency */
         const i32 l_rand = sv->RandState[last], l_Q = sv->Q_Q10[last], l_Xq = sv->Xq_Q14[last], l_Shape = sv->Shape_Q14[last], l_Pred = sv->Pred_Q15[last], l_gain = N->delayedGain_Q10[last];
         const i32 x_Q10 = N->x_sc_Q10[i];
         i32 LTP_pred_Q14 = 0, n_LTP_Q14 = 0;
         if (voiced) {
            LTP_pred_Q14 = 2;
            LTP_pred_Q14 = sk_mlawb(LTP_pred_Q14, N->sLTP_Q15[pred_lag], b0); LTP_pred_Q14 = sk_mlawb(LTP_pred_Q14, N->sLTP_Q15[pred_lag - 1], b1); LTP_pred_Q14 = sk_mlawb(LTP_pred_Q14, N->sLTP_Q15[pred_lag - 2], b2);
            LTP_pred_Q14 = sk_mlawb(LTP_pred_Q14, N->sLTP_Q15[pred_lag - 3], b3); LTP_pred_Q14 = sk_mlawb(LTP_pred_Q14, N->sLTP_Q15[pred_lag - 4], b4);
            LTP_pred_Q14 = shl32(LTP_pred_Q14, 1);
            pred_lag++;
         }
         if (lag > 0) {
            n_LTP_Q14 = sk_mulwb(sk_add_sat(st->sLTP_shp_Q14[shp_lag], st->sLTP_shp_Q14[shp_lag - 2]), HarmPacked_Q14);
            n_LTP_Q14 = sk_mlawt(n_LTP_Q14, st->sLTP_shp_Q14[shp_lag - 1], HarmPacked_Q14);
            n_LTP_Q14 = LTP_pred_Q14 - shl32(n_LTP_Q14, 2);
            shp_lag++;
         }
         Seed = sk_rand(Seed);
         i32 LPC_pred_Q14 = P >> 1;
#pragma unroll
         for (int j = 0; j < 16; j++) LPC_pred_Q14 = sk_mlawb(LPC_pred_Q14, w[j], ca[j]);
         LPC_pred_Q14 = shl32(LPC_pred_Q14, 4);
         i32 n_AR_Q14 = S >> 1;
         {  /* warped all-pass ladder: stage j consumes the output of stage j - 1 (:392-413); orders 12 / 14 / 16 / 20 / 24 (control_codec.c complexity table) */
            i32 in = sk_mlawb(Diff, sar[0], warp);
#define SE_AR_STAGE(j) { const i32 nxt = (j) + 1 < 24 ? sar[(j) + 1 < 24 ? (j) + 1 : 23] : 0; const i32 out = sk_mlawb(sar[j], sub32(nxt, in), warp); sar[j] = in; n_AR_Q14 = sk_mlawb(n_AR_Q14, in, cs[j]); in = out; }
#pragma unroll
            for (int j = 0; j < 12; j++) SE_AR_STAGE(j)
            if (S > 12) { SE_AR_STAGE(12) SE_AR_STAGE(13) }
            if (S > 14) { SE_AR_STAGE(14) SE_AR_STAGE(15) }
            if (S > 16) { SE_AR_STAGE(16) SE_AR_STAGE(17) SE_AR_STAGE(18) SE_AR_STAGE(19) }
            if (S > 20) { SE_AR_STAGE(20) SE_AR_STAGE(21) SE_AR_STAGE(22) SE_AR_STAGE(23) }                 /* (order 20 must leave taps 20..23 alone: they come back into play when the complexity rises) */
#undef SE_AR_STAGE
         }
         n_AR_Q14 = shl32(n_AR_Q14, 1);
         n_AR_Q14 = sk_mlawb(n_AR_Q14, LF_AR, Tilt_Q14);
         n_AR_Q14 = shl32(n_AR_Q14, 2);
         i32 n_LF_Q14 = sk_mulwb(sv->Shape_Q14[old_idx], LF_shp_Q14);
         n_LF_Q14 = sk_mlawt(n_LF_Q14, LF_AR, LF_shp_Q14);
         n_LF_Q14 = shl32(n_LF_Q14, 2);
         i32 t1 = sk_add_sat(n_AR_Q14, n_LF_Q14);
         const i32 t2 = add32(n_LTP_Q14, LPC_pred_Q14);
         t1 = sk_sub_sat(t2, t1);
         t1 = sk_rround(t1, 4);
         i32 r_Q10 = x_Q10 - t1;
         if (Seed < 0) r_Q10 = neg32(r_Q10);
         r_Q10 = se_limit(r_Q10, -(31 << 10), 30 << 10);
         i32 q1_Q10, q2_Q10, rd1, rd2;
         se_nsq_levels(r_Q10, offset_Q10, Lambda_Q10, q1_Q10, q2_Q10, rd1, rd2);
         rd1 >>= 10; rd2 >>= 10;
         const bool first_is_q1 = rd1 < rd2;
         SeCand c0, c1;
         c0.Q_Q10 = first_is_q1 ? q1_Q10 : q2_Q10; c0.RD_Q10 = RD + (first_is_q1 ? rd1 : rd2);
         c1.Q_Q10 = first_is_q1 ? q2_Q10 : q1_Q10; c1.RD_Q10 = RD + (first_is_q1 ? rd2 : rd1);
         {
            i32 exc_Q14 = shl32(c0.Q_Q10, 4); if (Seed < 0) exc_Q14 = -exc_Q14;
            c0.LPC_exc_Q14 = exc_Q14 + LTP_pred_Q14; c0.xq_Q14 = add32(c0.LPC_exc_Q14, LPC_pred_Q14); c0.Diff_Q14 = sub32(c0.xq_Q14, shl32(x_Q10, 4));
            c0.LF_AR_Q14 = sub32(c0.Diff_Q14, n_AR_Q14); c0.sLTP_shp_Q14 = sk_sub_sat(c0.LF_AR_Q14, n_LF_Q14);
            exc_Q14 = shl32(c1.Q_Q10, 4); if (Seed < 0) exc_Q14 = -exc_Q14;
            c1.LPC_exc_Q14 = exc_Q14 + LTP_pred_Q14; c1.xq_Q14 = add32(c1.LPC_exc_Q14, LPC_pred_Q14); c1.Diff_Q14 = sub32(c1.xq_Q14, shl32(x_Q10, 4));
            c1.LF_AR_Q14 = sub32(c1.Diff_Q14, n_AR_Q14); c1.sLTP_shp_Q14 = sk_sub_sat(c1.LF_AR_Q14, n_LF_Q14);
         }
         /* the K-way decisions (:530-590) on the scalar unit: costs and random states of lanes 0..K-1 read with constant lane selects */
         int winner = 0, worst = 0, best2 = 0;
         i32 worst_rd = 0, best2_rd = 0, my_pen = 0;
         {
            i32 r0[4], r1[4], rs[4];
#define SE_DD_GET(q) r0[q] = wv_lane_const<q>(c0.RD_Q10); r1[q] = wv_lane_const<q>(c1.RD_Q10); rs[q] = wv_lane_const<q>(l_rand);
            SE_DD_GET(0) SE_DD_GET(1) SE_DD_GET(2) SE_DD_GET(3)
#undef SE_DD_GET
            i32 win_rd = r0[0], wrand = rs[0];
#pragma unroll
            for (int q = 1; q < 4; q++) if (q < K && r0[q] < win_rd) { win_rd = r0[q]; winner = q; wrand = rs[q]; }
#pragma unroll
            for (int q = 0; q < 4; q++) if (q < K) {
               const i32 pen = rs[q] != wrand ? (2147483647 >> 4) : 0;
               const i32 a0 = r0[q] + pen, a1 = r1[q] + pen;
               if (q == 0 || a0 > worst_rd) { worst_rd = a0; worst = q; }
               if (q == 0 || a1 < best2_rd) { best2_rd = a1; best2 = q; }
               if (q == lane) my_pen = pen;
            }
         }
         const int replace = best2_rd < worst_rd;
         if (lane == winner && (subfr > 0 || i >= decisionDelay)) {                   /* commit the sample decisionDelay back from the winner (its ring entries were read above, before any ring is overwritten) */
            pls[i - decisionDelay] = (i8)sk_rround(l_Q, 10);
            pxq[i - decisionDelay] = (i16)sk_sat16(sk_rround(sk_mulww(l_Xq, l_gain), 8));
            st->sLTP_shp_Q14[shp_buf_idx - decisionDelay] = l_Shape;
            N->sLTP_Q15[ltp_buf_idx - decisionDelay] = l_Pred;
         }
         shp_buf_idx++; ltp_buf_idx++;
         /* the candidate this survivor continues with: its own first choice, or -- for the replaced survivor -- the second choice of the best one */
         i32 nQ = c0.Q_Q10, nRD = c0.RD_Q10 + my_pen, nxq = c0.xq_Q14, nLF = c0.LF_AR_Q14, nDiff = c0.Diff_Q14, nShp = c0.sLTP_shp_Q14, nExc = c0.LPC_exc_Q14;
         if (replace) {
            const bool me = lane == worst;
            if (me) nRD = best2_rd;
            switch (best2) {
            case 0: se_dd_take<0>(me, c1, nQ, nxq, nLF, nDiff, nShp, nExc, Seed, w, sar); break;
            case 1: se_dd_take<1>(me, c1, nQ, nxq, nLF, nDiff, nShp, nExc, Seed, w, sar); break;
            case 2: se_dd_take<2>(me, c1, nQ, nxq, nLF, nDiff, nShp, nExc, Seed, w, sar); break;
            default: se_dd_take<3>(me, c1, nQ, nxq, nLF, nDiff, nShp, nExc, Seed, w, sar); break;
            }
            /* rings + SeedInit through LDS (64-lane copy) */
            wv_order();
            WV_LDS i32 *d = (WV_LDS i32 *)&N->sv[worst]; const WV_LDS i32 *sr = (const WV_LDS i32 *)&N->sv[best2];
            for (int q = lane; q < (int)(sizeof(SeSurvivor) / 4); q += WV_WIDTH) d[q] = sr[q];
         }
         wv_order();
         LF_AR = nLF; Diff = nDiff;
#pragma unroll
         for (int j = 15; j > 0; j--) w[j] = w[j - 1];
         w[0] = nxq;
         Seed = add32(Seed, sk_rround(nQ, 10));
         RD = nRD;
         if (act) { sv->Xq_Q14[smpl_buf_idx] = nxq; sv->Q_Q10[smpl_buf_idx] = nQ; sv->Pred_Q15[smpl_buf_idx] = shl32(nExc, 1); sv->Shape_Q14[smpl_buf_idx] = nShp; sv->RandState[smpl_buf_idx] = Seed; }
         if (lane == 0) N->delayedGain_Q10[smpl_buf_idx] = Gain_Q10;
         wv_order();
      }
      wv_sync();
      LANE0 { st->sLTP_shp_buf_idx = shp_buf_idx; st->sLTP_buf_idx = ltp_buf_idx; }
      subfr++;
   }
   wv_sync();
   if (act) N->sv[lane].RD_Q10 = RD;
   wv_sync();
   const int wn = se_dd_best(N->sv, K);
   LANE0 {
      ix->Seed = (i8)N->sv[wn].SeedInit;
      se_dd_flush_l0(st, &N->sv[wn], smpl_buf_idx, decisionDelay, pulses + c->nb_subfr * L, &st->xq[mem + c->nb_subfr * L], ctl->Gains_Q16[c->nb_subfr - 1] >> 6, 8);
      st->lagPrev = ctl->pitchL[c->nb_subfr - 1];
   }
   if (lane == wn) {
#pragma unroll
      for (int j = 0; j < 16; j++) st->sLPC_Q14[15 - j] = w[j];
#pragma unroll
      for (int j = 0; j < 24; j++) st->sAR2_Q14[j] = sar[j];
      st->sLF_AR_shp_Q14 = LF_AR; st->sDiff_shp_Q14 = Diff;
   }
   wv_sync();
   for (int b = 0; b < mem; b += WV_WIDTH) {
      const int i = b + wv_lane(); i16 a = 0; i32 sv_ = 0;
      if (i < mem) { a = st->xq[frame + i]; sv_ = st->sLTP_shp_Q14[frame + i]; }
      wv_sync();
      if (i < mem) { st->xq[i] = a; st->sLTP_shp_Q14[i] = sv_; }
      wv_sync();
   }
}
#endif

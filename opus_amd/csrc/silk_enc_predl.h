/* silk_enc_predl.h — the SILK encoder's prediction stage with one LANE per coded channel (pipeline mode 4 of the split path, opus_sh_split.h: oa_sh_predl_tile).
 *
 * silk_find_LPC_FIX, silk_process_NLSFs, silk_residual_energy_FIX and silk_process_gains_FIX (silk/fixed/find_LPC_FIX.c:38, silk/process_NLSFs.c:36,
 * silk/fixed/residual_energy_FIX.c:36, silk/fixed/process_gains_FIX.c:36) are chains of short serial steps over 10..16 coefficients: Burg's order recursion, the root scan of
 * silk_A2NLSF, the polynomial recursions and the stability loop of silk_NLSF2A, the survivors' trellises of silk_NLSF_del_dec_quant.  One wave per channel (mode 3's
 * oa_sh_pred_frame) spends them on one or a few lanes.  Here a wave takes PL_STREAMS channels at once and every lane runs the whole stage of its own channel, serially, the way
 * the reference's C does -- the same scalar stage functions the wave code calls from single lanes (se_burg_modified_l0, sd_nlsf2a_w, se_nlsf_del_dec_quant, sd_nlsf_decode,
 * se_process_gains_l0), plus the scalar forms below of what the wave code spreads over lanes.  A lane's working set (PlLane, 1.9 KB) lives in LDS at an odd word stride, so
 * the lanes' accesses to the same field fall into different banks.  No wave collective is called between the tile's load and its store: the lanes diverge freely. */
#ifndef OPUS_AMD_SILK_ENC_PREDL_H
#define OPUS_AMD_SILK_ENC_PREDL_H

#ifndef PL_STREAMS
#define PL_STREAMS 16
#endif

struct PlNlsfWork { i32 err_Q24[32]; int idx[16]; SeNlsfTabs tabs; SeNlsfLane lane; i8 ti[16], best[16]; };
struct PlLane {
   /* the channel's fields and the control block's, as the stage functions name them (silk_encoder_state / silk_encoder_control_FIX) */
   i32 minInvGain_Q30, LTPredCodGain_Q7, coding_quality_Q14, input_quality_Q14;
   i32 predictLPCOrder, nb_subfr, subfr_length, useInterpolatedNLSFs, first_frame_after_reset, speech_activity_Q8, NLSF_MSVQ_Survivors, SNR_dB_Q7, input_tilt_Q15, nStatesDelayedDecision;
   i32 LastGainIndex, lastGainIndexPrev, Lambda_Q10, condCoding;
   i32 local_gains[4], Gains_Q16[4], GainsUnq_Q16[4], ResNrg[4], ResNrgQ[4];
   i32 a_Q16[16], a_tmp_Q16[16];
   i16 prev_NLSFq_Q15[16], NLSF_Q15[16], NLSF0_Q15[16], pW[16], PredCoef_Q12[2][16];
   OaSilkEncIndices indices;
   i16 x[4 * 16 + 320];                                          /* LPC_in_pre */
   union {                                                       /* one stage at a time */
      i32 stk[84];                                               /* Burg's five rows */
      struct { i32 wk[66]; i16 LPC_res[2 * 96]; } f;             /* NLSF -> LPC work area + the residual of a half frame (interpolation search, residual energies) */
      PlNlsfWork q;                                              /* the NLSF quantiser: one survivor's trellis at a time */
   } u;
};
static_assert(sizeof(PlLane) % 8 == 4, "PlLane: an odd number of words, so that the lanes' copies of a field sit in different LDS banks");

/* silk_A2NLSF (silk/A2NLSF.c:127) as the reference runs it: the polynomials in registers, evaluated where the scan stands */
template <int DD> WV_DEV void pl_a2nlsf_t(WV_LDS i16 *NLSF, WV_LDS i32 *a_Q16)
{
   const int d = 2 * DD;
   i32 P[DD + 1], Q[DD + 1];
   se_a2nlsf_poly_init<DD>(a_Q16, P, Q);
   bool useQ = false;
   i32 xlo = sk_lsf_cos_tab_q12[0], ylo = se_a2nlsf_eval<DD>(P, Q, false, xlo), thr = 0;
   int root_ix = 0, k = 1, i = 0;
   if (ylo < 0) { NLSF[0] = 0; useQ = true; ylo = se_a2nlsf_eval<DD>(P, Q, true, xlo); root_ix = 1; }
   while (1) {
      i32 xhi = sk_lsf_cos_tab_q12[k], yhi = se_a2nlsf_eval<DD>(P, Q, useQ, xhi);
      if ((ylo <= 0 && yhi >= thr) || (ylo >= 0 && yhi <= -thr)) {
         thr = yhi == 0 ? 1 : 0;
         int ffrac = -256;
         for (int m = 0; m < 3; m++) {
            const i32 xmid = sk_rround(xlo + xhi, 1), ymid = se_a2nlsf_eval<DD>(P, Q, useQ, xmid);
            if ((ylo <= 0 && ymid >= 0) || (ylo >= 0 && ymid <= 0)) { xhi = xmid; yhi = ymid; } else { xlo = xmid; ylo = ymid; ffrac = ffrac + (128 >> m); }
         }
         if (iabs(ylo) < 65536) { const i32 den = ylo - yhi, nom = shl32(ylo, 8 - 3) + (den >> 1); if (den != 0) ffrac += nom / den; }
         else ffrac += ylo / ((ylo - yhi) >> (8 - 3));
         NLSF[root_ix] = (i16)imin(shl32((i32)k, 8) + ffrac, 32767);
         root_ix++;
         if (root_ix >= d) break;
         useQ = (root_ix & 1) != 0;
         xlo = sk_lsf_cos_tab_q12[k - 1];
         ylo = shl32(1 - (root_ix & 2), 12);
      } else {
         k++; xlo = xhi; ylo = yhi; thr = 0;
         if (k > 128) {                                                           /* no full set of roots: bandwidth-expand and search again (:227) */
            i++;
            if (i > 16) { NLSF[0] = (i16)((1 << 15) / (d + 1)); for (k = 1; k < d; k++) NLSF[k] = (i16)(NLSF[k - 1] + NLSF[0]); return; }
            se_bwexpander_32(a_Q16, d, 65536 - shl32(1, i));
            se_a2nlsf_poly_init<DD>(a_Q16, P, Q);
            useQ = false; xlo = sk_lsf_cos_tab_q12[0]; ylo = se_a2nlsf_eval<DD>(P, Q, false, xlo);
            if (ylo < 0) { NLSF[0] = 0; useQ = true; ylo = se_a2nlsf_eval<DD>(P, Q, true, xlo); root_ix = 1; } else root_ix = 0;
            k = 1;
         }
      }
   }
}
WV_DEV void pl_a2nlsf(WV_LDS i16 *NLSF, WV_LDS i32 *a_Q16, int d) { if (d == 16) pl_a2nlsf_t<8>(NLSF, a_Q16); else pl_a2nlsf_t<5>(NLSF, a_Q16); }

/* silk_LPC_analysis_filter (silk/LPC_analysis_filter.c:49), serial: the coefficients and the last D inputs stay in registers */
template <int D> WV_DEV void pl_lpc_analysis_filter_t(WV_LDS i16 *out, const WV_LDS i16 *in, const WV_LDS i16 *B, int len)
{
   i32 b[D], h[D];
#pragma unroll
   for (int j = 0; j < D; j++) { b[j] = B[j]; h[j] = in[D - 1 - j]; out[j] = 0; }
   for (int ix = D; ix < len; ix++) {
      i32 o = 0;
#pragma unroll
      for (int j = 0; j < D; j++) o = sk_mlabb(o, h[j], b[j]);                    /* silk_SMLABB_ovflw: wraps */
      const i32 cur = in[ix];
      o = sub32(shl32(cur, 12), o);
      out[ix] = (i16)sk_sat16(sk_rround(o, 12));
#pragma unroll
      for (int j = D - 1; j > 0; j--) h[j] = h[j - 1];
      h[0] = cur;
   }
}
WV_DEV void pl_lpc_analysis_filter(WV_LDS i16 *out, const WV_LDS i16 *in, const WV_LDS i16 *B, int len, int d)
{ if (d == 16) pl_lpc_analysis_filter_t<16>(out, in, B, len); else pl_lpc_analysis_filter_t<10>(out, in, B, len); }

WV_DEV void pl_interpolate(WV_LDS i16 *xi, const WV_LDS i16 *x0, const WV_LDS i16 *x1, int ifact_Q2, int d) { for (int i = 0; i < d; i++) xi[i] = (i16)(x0[i] + (sk_mulbb(x1[i] - x0[i], ifact_Q2) >> 2)); }

/* silk_find_LPC_FIX (silk/fixed/find_LPC_FIX.c:38) */
WV_DEVN void pl_find_lpc(WV_LDS PlLane *c)
{
   const int order = c->predictLPCOrder, subfr_length = c->subfr_length + order;
   const WV_LDS i16 *x = c->x;
   i32 res_nrg; int res_nrg_Q;
   c->indices.NLSFInterpCoef_Q2 = 4;
   se_burg_modified_l0(&res_nrg, &res_nrg_Q, c->a_Q16, x, c->minInvGain_Q30, subfr_length, c->nb_subfr, order, c->u.stk);
   if (c->useInterpolatedNLSFs && !c->first_frame_after_reset && c->nb_subfr == 4) {
      i32 res_tmp_nrg; int res_tmp_nrg_Q;
      se_burg_modified_l0(&res_tmp_nrg, &res_tmp_nrg_Q, c->a_tmp_Q16, x + 2 * subfr_length, c->minInvGain_Q30, subfr_length, 2, order, c->u.stk);
      int shift = res_tmp_nrg_Q - res_nrg_Q;
      if (shift >= 0) { if (shift < 32) res_nrg = res_nrg - (res_tmp_nrg >> shift); }
      else { res_nrg = (res_nrg >> -shift) - res_tmp_nrg; res_nrg_Q = res_tmp_nrg_Q; }
      pl_a2nlsf(c->NLSF_Q15, c->a_tmp_Q16, order);
      WV_LDS i16 *a_tmp_Q12 = c->PredCoef_Q12[0], *LPC_res = c->u.f.LPC_res;     /* (PredCoef_Q12 is written by the quantiser stage, after this) */
      for (int k = 3; k >= 0; k--) {
         pl_interpolate(c->NLSF0_Q15, c->prev_NLSFq_Q15, c->NLSF_Q15, k, order);
         sd_nlsf2a_w(a_tmp_Q12, c->NLSF0_Q15, order, c->u.f.wk);
         pl_lpc_analysis_filter(LPC_res, x, a_tmp_Q12, 2 * subfr_length, order);
         i32 res_nrg0, res_nrg1; int rshift0, rshift1, res_nrg_interp_Q, isInterpLower;
         sd_sum_sqr_shift(&res_nrg0, &rshift0, LPC_res + order, subfr_length - order);
         sd_sum_sqr_shift(&res_nrg1, &rshift1, LPC_res + order + subfr_length, subfr_length - order);
         shift = rshift0 - rshift1;
         if (shift >= 0) { res_nrg1 >>= shift; res_nrg_interp_Q = -rshift0; } else { res_nrg0 >>= -shift; res_nrg_interp_Q = -rshift1; }
         const i32 res_nrg_interp = add32(res_nrg0, res_nrg1);
         shift = res_nrg_interp_Q - res_nrg_Q;
         if (shift >= 0) isInterpLower = (res_nrg_interp >> shift) < res_nrg;
         else if (-shift < 32) isInterpLower = res_nrg_interp < (res_nrg >> -shift);
         else isInterpLower = 0;
         if (isInterpLower) { res_nrg = res_nrg_interp; res_nrg_Q = res_nrg_interp_Q; c->indices.NLSFInterpCoef_Q2 = (i8)k; }
      }
   }
   if (c->indices.NLSFInterpCoef_Q2 == 4) pl_a2nlsf(c->NLSF_Q15, c->a_Q16, order);
}

/* silk_NLSF_encode (silk/NLSF_encode.c:38): the survivors' trellises one after the other, the best one kept as the loop goes (the reference's final sort with K = 1 picks
 * the first minimum) */
WV_DEVN void pl_nlsf_encode(WV_LDS PlLane *c, int NLSF_mu_Q20)
{
   const int order = c->predictLPCOrder, nSurvivors = c->NLSF_MSVQ_Survivors, signalType = c->indices.signalType;
   const SdNlsfCb cb = sd_nlsf_cb(order);
   const u8 *ec_rates_Q5 = order == 16 ? se_nlsf_wb_ec_rates_q5 : se_nlsf_nb_mb_ec_rates_q5;
   const i16 inv_qstep_Q6 = order == 16 ? SE_NLSF_WB_INV_QSTEP_Q6 : SE_NLSF_NB_MB_INV_QSTEP_Q6;
   WV_LDS PlNlsfWork *W = &c->u.q;
   WV_LDS i16 *pNLSF_Q15 = c->NLSF_Q15;
   sd_nlsf_stabilize((i16 *)pNLSF_Q15, cb.deltamin, order);
   for (int v = 0; v < cb.nVectors; v++) {                                                         /* silk_NLSF_VQ (silk/NLSF_VQ.c:35) */
      const u8 *cbq = &cb.cb1_nlsf[v * order]; const i16 *wq = &cb.wght[v * order];
      i32 sum = 0, pred = 0;
      for (int m = order - 2; m >= 0; m -= 2) {
         i32 d = sub32(pNLSF_Q15[m + 1], shl32((i32)cbq[m + 1], 7)), dw = sk_mulbb(d, wq[m + 1]);
         sum = add32(sum, iabs(sub32(dw, pred >> 1))); pred = dw;
         d = sub32(pNLSF_Q15[m], shl32((i32)cbq[m], 7)); dw = sk_mulbb(d, wq[m]);
         sum = add32(sum, iabs(sub32(dw, pred >> 1))); pred = dw;
      }
      W->err_Q24[v] = sum;
   }
   se_insertion_sort_increasing((i32 *)W->err_Q24, (int *)W->idx, cb.nVectors, nSurvivors);
   for (int i = 0; i < 20; i++) se_nlsf_out_tabs(&W->tabs, i, cb.qstep);
   i32 best_RD = 0; int best_ind1 = 0;
   WV_LDS SeNlsfLane *w = &W->lane;
   for (int s = 0; s < nSurvivors; s++) {
      const int ind1 = W->idx[s];
      const u8 *pCB = &cb.cb1_nlsf[ind1 * order]; const i16 *pWg = &cb.wght[ind1 * order];
      for (int i = 0; i < order; i++) {
         const i16 tmp = (i16)shl32((i16)pCB[i], 7);
         const i32 W_tmp_Q9 = pWg[i];
         w->res_Q10[i] = (i16)(sk_mulbb(pNLSF_Q15[i] - tmp, W_tmp_Q9) >> 14);
         w->W_adj_Q5[i] = (i16)sk_div32_varQ((i32)c->pW[i], sk_mulbb(W_tmp_Q9, W_tmp_Q9), 21);
      }
      {  /* silk_NLSF_unpack (NLSF_unpack.c:35) */
         const u8 *sel = &cb.ec_sel[ind1 * order / 2];
         for (int i = 0; i < order; i += 2) {
            const int entry = *sel++;
            w->ec_ix[i] = ((entry >> 1) & 7) * 9; w->pred_Q8[i] = cb.pred[i + (entry & 1) * (order - 1)];
            w->ec_ix[i + 1] = ((entry >> 5) & 7) * 9; w->pred_Q8[i + 1] = cb.pred[i + ((entry >> 4) & 1) * (order - 1) + 1];
         }
      }
      i32 RD = se_nlsf_del_dec_quant(w, W->ti, &W->tabs, ec_rates_Q5, inv_qstep_Q6, NLSF_mu_Q20, order);
      const u8 *icdf = &cb.cb1_icdf[(signalType >> 1) * cb.nVectors];
      const int prob_Q8 = ind1 == 0 ? 256 - icdf[ind1] : icdf[ind1 - 1] - icdf[ind1];
      const int bits_q7 = (8 << 7) - se_lin2log(prob_Q8);
      RD = sk_mlabb(RD, bits_q7, NLSF_mu_Q20 >> 2);
      if (s == 0 || RD < best_RD) { best_RD = RD; best_ind1 = ind1; for (int i = 0; i < order; i++) W->best[i] = W->ti[i]; }
   }
   c->indices.NLSFIndices[0] = (i8)best_ind1;
   for (int i = 0; i < order; i++) c->indices.NLSFIndices[1 + i] = W->best[i];
   sd_nlsf_decode((i16 *)pNLSF_Q15, c->indices.NLSFIndices, cb);
}

/* silk_process_NLSFs (silk/process_NLSFs.c:36) */
WV_DEVN void pl_process_nlsfs(WV_LDS PlLane *c)
{
   const int order = c->predictLPCOrder;
   int NLSF_mu_Q20 = sk_mlawb(SE_FIX(0.003, 20), SE_FIX(-0.001, 28), c->speech_activity_Q8);
   if (c->nb_subfr == 2) NLSF_mu_Q20 = NLSF_mu_Q20 + (NLSF_mu_Q20 >> 1);
   const int ic = c->indices.NLSFInterpCoef_Q2;
   const int doInterpolate = c->useInterpolatedNLSFs == 1 && ic < 4;
   se_nlsf_vq_weights((i16 *)c->pW, (const i16 *)c->NLSF_Q15, order);
   if (doInterpolate) {
      WV_LDS i16 *w0 = c->PredCoef_Q12[1];                                      /* (free until the conversions below) */
      pl_interpolate(c->NLSF0_Q15, c->prev_NLSFq_Q15, c->NLSF_Q15, ic, order);
      se_nlsf_vq_weights((i16 *)w0, (const i16 *)c->NLSF0_Q15, order);
      const i16 i_sqr_Q15 = (i16)shl32(sk_mulbb(ic, ic), 11);
      for (int i = 0; i < order; i++) c->pW[i] = (i16)((c->pW[i] >> 1) + (sk_mulbb(w0[i], i_sqr_Q15) >> 16));
   }
   pl_nlsf_encode(c, NLSF_mu_Q20);
   sd_nlsf2a_w(c->PredCoef_Q12[1], c->NLSF_Q15, order, c->u.f.wk);
   if (doInterpolate) {
      pl_interpolate(c->NLSF0_Q15, c->prev_NLSFq_Q15, c->NLSF_Q15, ic, order);
      sd_nlsf2a_w(c->PredCoef_Q12[0], c->NLSF0_Q15, order, c->u.f.wk);
   } else for (int i = 0; i < order; i++) c->PredCoef_Q12[0][i] = c->PredCoef_Q12[1][i];
}

/* silk_residual_energy_FIX (silk/fixed/residual_energy_FIX.c:36) */
WV_DEVN void pl_residual_energy(WV_LDS PlLane *c)
{
   const int order = c->predictLPCOrder, nb = c->nb_subfr, offset = order + c->subfr_length;
   const WV_LDS i16 *x_ptr = c->x;
   WV_LDS i16 *LPC_res = c->u.f.LPC_res;
   for (int i = 0; i < nb >> 1; i++) {
      pl_lpc_analysis_filter(LPC_res, x_ptr, c->PredCoef_Q12[i], 2 * offset, order);
      for (int j = 0; j < 2; j++) { i32 e; int rshift; sd_sum_sqr_shift(&e, &rshift, LPC_res + order + j * offset, c->subfr_length); c->ResNrg[i * 2 + j] = e; c->ResNrgQ[i * 2 + j] = -rshift; }
      x_ptr += 2 * offset;
   }
   for (int i = 0; i < nb; i++) {
      const int lz1 = sk_clz(c->ResNrg[i]) - 1, lz2 = sk_clz(c->local_gains[i]) - 1;
      i32 t = shl32(c->local_gains[i], lz2);
      t = sk_mulhi(t, t);
      c->ResNrg[i] = sk_mulhi(t, shl32(c->ResNrg[i], lz1));
      c->ResNrgQ[i] += lz1 + 2 * lz2 - 32 - 32;
   }
}

/* the whole stage of one channel: find_pred_coefs_FIX.c:115-144, encode_frame_FIX.c:157 */
WV_DEV void pl_pred_lane(WV_LDS PlLane *c)
{
   pl_find_lpc(c);
   pl_process_nlsfs(c);
   pl_residual_energy(c);
   se_process_gains_l0(c, c, c->condCoding);
}
#endif

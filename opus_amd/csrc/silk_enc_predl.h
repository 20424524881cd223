/* silk_enc_predl.h — the serial parts of the SILK encoder's prediction stage with one LANE per coded channel (pipeline mode 4 of the split path, opus_sh_split.h).
 *
 * silk_find_LPC_FIX and silk_process_NLSFs (silk/fixed/find_LPC_FIX.c:38, silk/process_NLSFs.c:36) are, between their passes over the signal, chains of short serial steps over
 * 10..16 coefficients: Burg's order recursion, the root scan of silk_A2NLSF, the polynomial recursions and the stability loop of silk_NLSF2A, the survivors' trellises of
 * silk_NLSF_del_dec_quant.  One wave per channel (mode 3's oa_sh_pred_frame) spends a 64-lane instruction on one or a few lanes for each of those steps, and the encoder's
 * kernels are bound by exactly that: VALU issue (0.60-0.67 busy per SIMD at 20-30 active lanes of 64, profiles/pmc_traffic_r0*.json).  Mode 4 cuts the stage where its passes
 * over the signal end and gives the serial parts to LANE kernels -- 64 channels per wave, every lane running its own channel the way the reference's C does, with the same
 * scalar stage functions the wave code calls from single lanes -- and keeps the passes over the signal on whole waves:
 *   front kernel   ... + se_burg_corr_wave: energy and first correlation row of the two Burg analyses (the signal is in LDS there)
 *   pl_stage_a     LANE   the two Burg recursions on the subframes' edges, A2NLSF of both, the four interpolation candidates' NLSF2A          -> ShPredMid
 *   oa_sh_predc    wave   the candidates' residual energies over the first half frame, the interpolation choice                              -> ShPredMid.coef / NLSF_Q15
 *   pl_stage_b     LANE   the NLSF weights, silk_NLSF_encode (stage-1 VQ, sort, the survivors' trellises one after the other), both NLSF2A    -> PredCoef_Q12, NLSFIndices
 *   oa_sh_pred_frame(tail) wave   silk_residual_energy_FIX, silk_process_gains_FIX, the quantiser's job
 * A lane's working set lives in LDS at an odd word stride (the lanes' copies of a field fall into different banks): 596 B in stage A, 524 B in stage B, so that four waves
 * of 64 channels share a CU.  No wave collective is called inside a lane's stage: the lanes diverge freely. */
#ifndef OPUS_AMD_SILK_ENC_PREDL_H
#define OPUS_AMD_SILK_ENC_PREDL_H

#define PL_STREAMS 64

/* between the stage's kernels, per coded channel (ShCont.m) */
struct ShPredMid {
   i32 res_nrg, res_nrg_Q, interp, coef;
   i32 a_Q16[16];                                                /* the full-frame analysis (parked while the second one uses the lane's rows) */
   i16 NLSF_half[16], NLSF_full[16], cand_a[4][16];
   i16 NLSF_Q15[16];                                             /* what the interpolation choice leaves to be quantised */
};

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
WV_DEV void pl_interpolate(WV_LDS i16 *xi, const WV_LDS i16 *x0, const WV_LDS i16 *x1, int ifact_Q2, int d) { for (int i = 0; i < d; i++) xi[i] = (i16)(x0[i] + (sk_mulbb(x1[i] - x0[i], ifact_Q2) >> 2)); }

/* ---- stage A: find_LPC_FIX.c:62-86 and :140-143 up to where the candidates' residual energies are wanted ---- */
#define PL_A_WORDS 149                                            /* Burg's five rows (84) + four subframes' edges (64) [+ 1: odd stride]; the later steps work in the same words */
WV_DEVN void pl_stage_a(WV_LDS i32 *F, const ShPredIn *in, ShPredMid *md)
{
   const int order = in->predictLPCOrder, nb = in->nb_subfr, L = in->subfr_length + order;
   WV_LDS i16 *ed = (WV_LDS i16 *)(F + 84);
   for (int s = 0; s < nb; s++) {                                 /* the first and the last 16 samples of every subframe: all the recursion reads of the signal */
      const i32 *h = (const i32 *)(in->LPC_in_pre + s * L), *t = (const i32 *)(in->LPC_in_pre + s * L + L - 16);           /* (L is even: word loads) */
      WV_LDS i32 *e = (WV_LDS i32 *)(ed + s * 32);
      for (int i = 0; i < 8; i++) { e[i] = h[i]; e[8 + i] = t[i]; }
   }
   const int interp = in->useInterpolatedNLSFs && !in->first_frame_after_reset && nb == 4;
   i32 res_nrg; int res_nrg_Q;
   for (int k = 0; k < 16; k++) F[k] = in->bc[0].first_row[k];
   { const SeBurgXEdges xa = {ed}; se_burg_rec_l0(&res_nrg, &res_nrg_Q, (i32 *)md->a_Q16, xa, in->bc[0].C0, in->bc[0].rshifts, in->minInvGain_Q30, nb, order, F); }
   WV_LDS i16 *NLSF = (WV_LDS i16 *)(F + 16);
   if (interp) {
      i32 res_tmp_nrg; int res_tmp_nrg_Q;
      for (int k = 0; k < 16; k++) F[k] = in->bc[1].first_row[k];
      { const SeBurgXEdges xa = {ed + 64}; se_burg_rec_l0(&res_tmp_nrg, &res_tmp_nrg_Q, F /* the coefficients land where the first row was */, xa, in->bc[1].C0, in->bc[1].rshifts, in->minInvGain_Q30, 2, order, F); }
      const int shift = res_tmp_nrg_Q - res_nrg_Q;
      if (shift >= 0) { if (shift < 32) res_nrg = res_nrg - (res_tmp_nrg >> shift); }
      else { res_nrg = (res_nrg >> -shift) - res_tmp_nrg; res_nrg_Q = res_tmp_nrg_Q; }
      pl_a2nlsf(NLSF, F, order);
      WV_LDS i16 *prev = (WV_LDS i16 *)(F + 24), *n0 = (WV_LDS i16 *)(F + 32), *ao = (WV_LDS i16 *)(F + 40);
      WV_LDS i32 *wk = F + 48;                                    /* 66 words: ends at 114 of 148 */
      for (int i = 0; i < order; i++) { md->NLSF_half[i] = NLSF[i]; prev[i] = in->prev_NLSFq_Q15[i]; }
      for (int k = 0; k < 4; k++) {
         pl_interpolate(n0, prev, NLSF, k, order);
         sd_nlsf2a_w(ao, n0, order, wk);
         for (int i = 0; i < order; i++) md->cand_a[k][i] = ao[i];
      }
   }
   /* the full-frame NLSFs: what is quantised when the search keeps no interpolation (:140; worked out here either way -- the choice is the next kernel's) */
   for (int k = 0; k < order; k++) F[k] = md->a_Q16[k];
   pl_a2nlsf(NLSF, F, order);
   for (int i = 0; i < order; i++) md->NLSF_full[i] = NLSF[i];
   md->res_nrg = res_nrg; md->res_nrg_Q = res_nrg_Q; md->interp = interp;
}

/* ---- stage B: silk_process_NLSFs (silk/process_NLSFs.c:36) ---- */
struct PlBLane {
   i16 NLSF_Q15[16], NLSF0_Q15[16], pW[16], prev[16], PredCoef_Q12[2][16];
   i8 ind[20];                                                    /* NLSFIndices */
   union {
      struct { int idx[16]; union { i32 err_Q24[32]; SeNlsfLane lane; } e; i8 ti[16], best[16]; } q;      /* the stage-1 errors die when the survivors are known */
      i32 wk[66];
   } u;
   i32 pad_;
};
static_assert(sizeof(PlBLane) % 8 == 4, "PlBLane: an odd number of words");

/* silk_NLSF_encode (silk/NLSF_encode.c:38): the survivors' trellises one after the other, the best one kept as the loop goes (the reference's final sort with K = 1 picks the
 * first minimum).  T: the quantiser's output tables of this order (shared by the wave's lanes) */
WV_DEVN void pl_nlsf_encode(WV_LDS PlBLane *c, int order, int NLSF_mu_Q20, int nSurvivors, int signalType, const WV_LDS SeNlsfTabs *T)
{
   const SdNlsfCb cb = sd_nlsf_cb(order);
   const u8 *ec_rates_Q5 = order == 16 ? se_nlsf_wb_ec_rates_q5 : se_nlsf_nb_mb_ec_rates_q5;
   const i16 inv_qstep_Q6 = order == 16 ? SE_NLSF_WB_INV_QSTEP_Q6 : SE_NLSF_NB_MB_INV_QSTEP_Q6;
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
      c->u.q.e.err_Q24[v] = sum;
   }
   se_insertion_sort_increasing((i32 *)c->u.q.e.err_Q24, (int *)c->u.q.idx, cb.nVectors, nSurvivors);
   i32 best_RD = 0; int best_ind1 = 0;
   WV_LDS SeNlsfLane *w = &c->u.q.e.lane;
   for (int s = 0; s < nSurvivors; s++) {
      const int ind1 = c->u.q.idx[s];
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
      i32 RD = se_nlsf_del_dec_quant(w, c->u.q.ti, T, ec_rates_Q5, inv_qstep_Q6, NLSF_mu_Q20, order);
      const u8 *icdf = &cb.cb1_icdf[(signalType >> 1) * cb.nVectors];
      const int prob_Q8 = ind1 == 0 ? 256 - icdf[ind1] : icdf[ind1 - 1] - icdf[ind1];
      const int bits_q7 = (8 << 7) - se_lin2log(prob_Q8);
      RD = sk_mlabb(RD, bits_q7, NLSF_mu_Q20 >> 2);
      if (s == 0 || RD < best_RD) { best_RD = RD; best_ind1 = ind1; for (int i = 0; i < order; i++) c->u.q.best[i] = c->u.q.ti[i]; }
   }
   c->ind[0] = (i8)best_ind1;
   for (int i = 0; i < order; i++) c->ind[1 + i] = c->u.q.best[i];
   sd_nlsf_decode((i16 *)pNLSF_Q15, c->ind, cb);
}
/* c->NLSF_Q15 / prev / ind hold the unquantised NLSFs, last frame's quantised ones and the channel's current NLSFIndices on entry */
WV_DEVN void pl_stage_b(WV_LDS PlBLane *c, const ShPredIn *in, int ic /* NLSFInterpCoef_Q2 */, int signalType, const WV_LDS SeNlsfTabs *T)
{
   const int order = in->predictLPCOrder;
   int NLSF_mu_Q20 = sk_mlawb(SE_FIX(0.003, 20), SE_FIX(-0.001, 28), in->speech_activity_Q8);
   if (in->nb_subfr == 2) NLSF_mu_Q20 = NLSF_mu_Q20 + (NLSF_mu_Q20 >> 1);
   const int doInterpolate = in->useInterpolatedNLSFs == 1 && ic < 4;
   se_nlsf_vq_weights((i16 *)c->pW, (const i16 *)c->NLSF_Q15, order);
   if (doInterpolate) {
      WV_LDS i16 *w0 = c->PredCoef_Q12[1];                                      /* (free until the conversions below) */
      pl_interpolate(c->NLSF0_Q15, c->prev, c->NLSF_Q15, ic, order);
      se_nlsf_vq_weights((i16 *)w0, (const i16 *)c->NLSF0_Q15, order);
      const i16 i_sqr_Q15 = (i16)shl32(sk_mulbb(ic, ic), 11);
      for (int i = 0; i < order; i++) c->pW[i] = (i16)((c->pW[i] >> 1) + (sk_mulbb(w0[i], i_sqr_Q15) >> 16));
   }
   pl_nlsf_encode(c, order, NLSF_mu_Q20, in->NLSF_MSVQ_Survivors, signalType, T);
   sd_nlsf2a_w(c->PredCoef_Q12[1], c->NLSF_Q15, order, c->u.wk);
   if (doInterpolate) {
      pl_interpolate(c->NLSF0_Q15, c->prev, c->NLSF_Q15, ic, order);
      sd_nlsf2a_w(c->PredCoef_Q12[0], c->NLSF0_Q15, order, c->u.wk);
   } else for (int i = 0; i < order; i++) c->PredCoef_Q12[0][i] = c->PredCoef_Q12[1][i];
}
#endif

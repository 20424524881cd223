/* silk_enc_quant.h — SILK encoder: LPC -> NLSF conversion and quantisation, prediction-coefficient search, gain processing (row a20 of SURVEY §8).
 *
 *   se_a2nlsf                silk_A2NLSF                    silk/A2NLSF.c:127 (trans_poly :43, eval_poly :60, init :97)
 *   se_nlsf_vq_weights       silk_NLSF_VQ_weights_laroia    silk/NLSF_VQ_weights_laroia.c:40
 *   se_nlsf_del_dec_quant    silk_NLSF_del_dec_quant        silk/NLSF_del_dec_quant.c:35
 *   se_nlsf_encode_wave      silk_NLSF_encode, silk_NLSF_VQ, silk_insertion_sort_increasing   silk/NLSF_encode.c:38, NLSF_VQ.c:35, sort.c:35
 *   se_process_nlsfs_wave    silk_process_NLSFs             silk/process_NLSFs.c:36
 *   se_find_lpc_wave         silk_find_LPC_FIX              silk/fixed/find_LPC_FIX.c:38
 *   se_find_pred_coefs_wave  silk_find_pred_coefs_FIX       silk/fixed/find_pred_coefs_FIX.c:36
 *   se_gains_quant / _ID     silk_gains_quant / silk_gains_ID   silk/gain_quant.c:39,:130
 *   se_process_gains         silk_process_gains_FIX         silk/fixed/process_gains_FIX.c:36
 * The first-stage survivors of the NLSF quantiser are independent trellis searches: one lane per survivor. */
#ifndef OPUS_AMD_SILK_ENC_QUANT_H
#define OPUS_AMD_SILK_ENC_QUANT_H

/* silk_A2NLSF (silk/A2NLSF.c:127) on the wave.  The two polynomials live in registers of every lane (fixed-size arrays, unrolled loops, DD = d / 2 known at
 * compile time); their values at the 129 table points are independent -> lanes evaluate the grid into LDS, lane 0 then runs the reference's root scan on the
 * table and only the 3-step bisections evaluate the polynomial again.  Y: 2 x 132 words of LDS; a_Q16 (LDS) is bandwidth-expanded in the rare retry path. */
template <int DD> WV_DEV void se_a2nlsf_poly_init(const WV_LDS i32 *a_Q16, i32 *P, i32 *Q)
{
#pragma unroll
   for (int k = 0; k < DD; k++) { P[k] = -a_Q16[DD - k - 1] - a_Q16[DD + k]; Q[k] = -a_Q16[DD - k - 1] + a_Q16[DD + k]; }
   P[DD] = 1 << 16; Q[DD] = 1 << 16;
#pragma unroll
   for (int k = DD; k > 0; k--) { P[k - 1] -= P[k]; Q[k - 1] += Q[k]; }
#pragma unroll
   for (int k = 2; k <= DD; k++) {
#pragma unroll
      for (int n = DD; n > k; n--) { P[n - 2] -= P[n]; Q[n - 2] -= Q[n]; }
      P[k - 2] -= shl32(P[k], 1); Q[k - 2] -= shl32(Q[k], 1);
   }
}
template <int DD> WV_DEV i32 se_a2nlsf_eval(const i32 *P, const i32 *Q, bool useQ, i32 x)
{
   i32 y32 = useQ ? Q[DD] : P[DD]; const i32 x_Q16 = shl32(x, 4);
#pragma unroll
   for (int n = DD - 1; n >= 0; n--) y32 = sk_mlaww(useQ ? Q[n] : P[n], y32, x_Q16);
   return y32;
}
template <int DD> WV_DEV void se_a2nlsf_wave_t(WV_LDS i16 *NLSF, WV_LDS i32 *a_Q16, WV_LDS i32 *Y)
{
   const int d = 2 * DD;
   for (int attempt = 0; ; attempt++) {
      i32 P[DD + 1], Q[DD + 1];
      se_a2nlsf_poly_init<DD>(a_Q16, P, Q);
      FOR_LANES(i, 2 * 129) { const bool q = i >= 129; const int k = q ? i - 129 : i; Y[q * 132 + k] = se_a2nlsf_eval<DD>(P, Q, q, sk_lsf_cos_tab_q12[k]); }
      wv_sync();
      LANE0 {
         bool useQ = false;
         i32 xlo = sk_lsf_cos_tab_q12[0], ylo = Y[0], xhi, yhi, thr = 0;
         int root_ix, k = 1, done = 0;
         if (ylo < 0) { NLSF[0] = 0; useQ = true; ylo = Y[132]; root_ix = 1; } else root_ix = 0;
         while (1) {
            xhi = sk_lsf_cos_tab_q12[k];
            yhi = Y[useQ * 132 + k];
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
               if (root_ix >= d) { done = 1; break; }
               useQ = (root_ix & 1) != 0;
               xlo = sk_lsf_cos_tab_q12[k - 1];
               ylo = shl32(1 - (root_ix & 2), 12);
            } else {
               k++; xlo = xhi; ylo = yhi; thr = 0;
               if (k > 128) {                                                      /* no full set of roots: bandwidth-expand and search again (:227) */
                  if (attempt + 1 > 16) { NLSF[0] = (i16)((1 << 15) / (d + 1)); for (k = 1; k < d; k++) NLSF[k] = (i16)(NLSF[k - 1] + NLSF[0]); done = 1; }
                  else se_bwexpander_32(a_Q16, d, 65536 - shl32(1, attempt + 1));
                  break;
               }
            }
         }
         Y[131] = done;
      }
      if (Y[131]) break;
      wv_sync();
   }
   wv_sync();
}
WV_DEV void se_a2nlsf_wave(WV_LDS i16 *NLSF, WV_LDS i32 *a_Q16, int d, WV_LDS i32 *Y) { if (d == 16) se_a2nlsf_wave_t<8>(NLSF, a_Q16, Y); else se_a2nlsf_wave_t<5>(NLSF, a_Q16, Y); }
WV_DEV void se_interpolate(i16 *xi, const WV_LDS i16 *x0, const i16 *x1, int ifact_Q2, int d) { for (int i = 0; i < d; i++) xi[i] = (i16)(x0[i] + (sk_mulbb(x1[i] - x0[i], ifact_Q2) >> 2)); }
WV_DEV void se_nlsf_vq_weights(i16 *pW, const i16 *pN, int D)
{
   i32 t1 = imax(pN[0], 1); t1 = ((i32)1 << 17) / t1;
   i32 t2 = imax(pN[1] - pN[0], 1); t2 = ((i32)1 << 17) / t2;
   pW[0] = (i16)imin(t1 + t2, 32767);
   for (int k = 1; k < D - 1; k += 2) {
      t1 = imax(pN[k + 1] - pN[k], 1); t1 = ((i32)1 << 17) / t1; pW[k] = (i16)imin(t1 + t2, 32767);
      t2 = imax(pN[k + 2] - pN[k + 1], 1); t2 = ((i32)1 << 17) / t2; pW[k + 1] = (i16)imin(t1 + t2, 32767);
   }
   t1 = imax((1 << 15) - pN[D - 1], 1); t1 = ((i32)1 << 17) / t1;
   pW[D - 1] = (i16)imin(t1 + t2, 32767);
}
WV_DEV void se_insertion_sort_increasing(i32 *a, int *idx, int L, int K)
{
   int i, j;
   for (i = 0; i < K; i++) idx[i] = i;
   for (i = 1; i < K; i++) { const i32 v = a[i]; for (j = i - 1; j >= 0 && v < a[j]; j--) { a[j + 1] = a[j]; idx[j + 1] = idx[j]; } a[j + 1] = v; idx[j + 1] = i; }
   for (i = K; i < L; i++) { const i32 v = a[i]; if (v < a[K - 1]) { for (j = K - 2; j >= 0 && v < a[j]; j--) { a[j + 1] = a[j]; idx[j + 1] = idx[j]; } a[j + 1] = v; idx[j + 1] = i; } }
}

/* per-survivor working set of the NLSF trellis search, in LDS (run-time indexed private arrays would be scratch memory = HBM round trips) */
struct SeNlsfLane {                                             /* one survivor's trellis (212 B: sixteen of them set the size of the prediction-coefficient phase) */
   i32 RD_Q25[8];
   i16 res_Q10[16], W_adj_Q5[16], prev_out_Q10[8];
   u8 ec_ix[16], pred_Q8[16];
   i8 ind[4][16], ind_sort[4];
};
struct SeNlsfTabs { i32 out0[20], out1[20]; };
WV_DEV void se_nlsf_out_tabs(WV_LDS SeNlsfTabs *T, int i /* 0..19 */, int quant_step_size_Q16)                 /* NLSF_del_dec_quant.c:66-87 */
{
   const int v = i - 10, adj = SE_FIX(0.1, 10);
   i16 out0 = (i16)shl32(v, 10), out1 = (i16)(out0 + 1024);
   if (v > 0) { out0 = (i16)(out0 - adj); out1 = (i16)(out1 - adj); } else if (v == 0) out1 = (i16)(out1 - adj); else if (v == -1) out0 = (i16)(out0 + adj); else { out0 = (i16)(out0 + adj); out1 = (i16)(out1 + adj); }
   T->out0[i] = sk_mulbb(out0, quant_step_size_Q16) >> 16; T->out1[i] = sk_mulbb(out1, quant_step_size_Q16) >> 16;
}
WV_DEV i32 se_nlsf_del_dec_quant(WV_LDS SeNlsfLane *w, WV_LDS i8 *ti /* out: the winning path's indices [order] */, const WV_LDS SeNlsfTabs *T, const u8 *ec_rates_Q5, i16 inv_quant_step_size_Q6, i32 mu_Q20, int order)
{
   const int NS = 4, AMP = 4, EXT = 10;
   int i, j, nStates, ind_tmp, ind_min_max, ind_max_min;
   nStates = 1; w->RD_Q25[0] = 0; w->prev_out_Q10[0] = 0;
   for (i = order - 1; i >= 0; i--) {
      const u8 *rates_Q5 = &ec_rates_Q5[w->ec_ix[i]];
      const int in_Q10 = w->res_Q10[i], wi = w->W_adj_Q5[i], pc = (i16)w->pred_Q8[i];
      for (j = 0; j < nStates; j++) {
         const int pred_Q10 = sk_mulbb(pc, w->prev_out_Q10[j]) >> 8;
         const int res_Q10 = (i16)(in_Q10 - pred_Q10);
         ind_tmp = sk_mulbb(inv_quant_step_size_Q6, res_Q10) >> 16;
         ind_tmp = se_limit(ind_tmp, -EXT, EXT - 1);
         w->ind[j][i] = (i8)ind_tmp;
         i16 out0 = (i16)T->out0[ind_tmp + EXT], out1 = (i16)T->out1[ind_tmp + EXT];
         out0 = (i16)(out0 + pred_Q10); out1 = (i16)(out1 + pred_Q10);
         w->prev_out_Q10[j] = out0; w->prev_out_Q10[j + nStates] = out1;
         int rate0_Q5, rate1_Q5;
         if (ind_tmp + 1 >= AMP) {
            if (ind_tmp + 1 == AMP) { rate0_Q5 = rates_Q5[ind_tmp + AMP]; rate1_Q5 = 280; }
            else { rate0_Q5 = sk_mlabb(280 - 43 * AMP, 43, ind_tmp); rate1_Q5 = (i16)(rate0_Q5 + 43); }
         } else if (ind_tmp <= -AMP) {
            if (ind_tmp == -AMP) { rate0_Q5 = 280; rate1_Q5 = rates_Q5[ind_tmp + 1 + AMP]; }
            else { rate0_Q5 = sk_mlabb(280 - 43 * AMP, -43, ind_tmp); rate1_Q5 = (i16)(rate0_Q5 - 43); }
         } else { rate0_Q5 = rates_Q5[ind_tmp + AMP]; rate1_Q5 = rates_Q5[ind_tmp + 1 + AMP]; }
         const i32 RD_tmp = w->RD_Q25[j];
         int diff = (i16)(in_Q10 - out0);
         w->RD_Q25[j] = sk_mlabb(add32(RD_tmp, (i32)((u32)sk_mulbb(diff, diff) * (u32)(i32)wi)), mu_Q20, rate0_Q5);
         diff = (i16)(in_Q10 - out1);
         w->RD_Q25[j + nStates] = sk_mlabb(add32(RD_tmp, (i32)((u32)sk_mulbb(diff, diff) * (u32)(i32)wi)), mu_Q20, rate1_Q5);
      }
      if (nStates <= NS / 2) {
         for (j = 0; j < nStates; j++) w->ind[j + nStates][i] = (i8)(w->ind[j][i] + 1);
         nStates <<= 1;
         for (j = nStates; j < NS; j++) w->ind[j][i] = w->ind[j - nStates][i];
      } else {
         i32 RD_min_Q25[4], RD_max_Q25[4];                     /* (registers: every access below is statically indexed) */
#pragma unroll
         for (j = 0; j < 4; j++) {
            if (w->RD_Q25[j] > w->RD_Q25[j + NS]) {
               RD_max_Q25[j] = w->RD_Q25[j]; RD_min_Q25[j] = w->RD_Q25[j + NS]; w->RD_Q25[j] = RD_min_Q25[j]; w->RD_Q25[j + NS] = RD_max_Q25[j];
               const i16 t = w->prev_out_Q10[j]; w->prev_out_Q10[j] = w->prev_out_Q10[j + NS]; w->prev_out_Q10[j + NS] = t;
               w->ind_sort[j] = (i8)(j + NS);
            } else { RD_min_Q25[j] = w->RD_Q25[j]; RD_max_Q25[j] = w->RD_Q25[j + NS]; w->ind_sort[j] = (i8)j; }
         }
         while (1) {
            i32 min_max = 2147483647, max_min = 0;
            ind_min_max = 0; ind_max_min = 0;
#pragma unroll
            for (j = 0; j < 4; j++) { if (min_max > RD_max_Q25[j]) { min_max = RD_max_Q25[j]; ind_min_max = j; } if (max_min < RD_min_Q25[j]) { max_min = RD_min_Q25[j]; ind_max_min = j; } }
            if (min_max >= max_min) break;
            w->ind_sort[ind_max_min] = (i8)(w->ind_sort[ind_min_max] ^ NS);
            w->RD_Q25[ind_max_min] = w->RD_Q25[ind_min_max + NS];
            w->prev_out_Q10[ind_max_min] = w->prev_out_Q10[ind_min_max + NS];
#pragma unroll
            for (j = 0; j < 4; j++) { if (j == ind_max_min) RD_min_Q25[j] = 0; if (j == ind_min_max) RD_max_Q25[j] = 2147483647; }
            for (int q = 0; q < 16; q++) w->ind[ind_max_min][q] = w->ind[ind_min_max][q];
         }
         for (j = 0; j < NS; j++) w->ind[j][i] = (i8)(w->ind[j][i] + (w->ind_sort[j] >> 2));
      }
   }
   ind_tmp = 0;
   i32 min_Q25 = 2147483647;
   for (j = 0; j < 2 * NS; j++) if (min_Q25 > w->RD_Q25[j]) { min_Q25 = w->RD_Q25[j]; ind_tmp = j; }
   for (j = 0; j < order; j++) ti[j] = w->ind[ind_tmp & (NS - 1)][j];
   ti[0] = (i8)(ti[0] + (ind_tmp >> 2));
   return min_Q25;
}

/* scratch of the LPC / NLSF stages (LDS) */
struct SeLpcWork {
   i32 a_Q16[16], a_tmp_Q16[16], invGains_Q16[4], local_gains[4], r[8];
   i16 NLSF_Q15[16], pW[16];
   union {
      struct {
         i16 LPC_in_pre[4 * 16 + 320];                         /* the gain-scaled (voiced: LTP-filtered) input of the LPC analysis, find_pred_coefs_FIX.c:45 */
         union {                                               /* one stage at a time: LTP correlations -> Burg -> A2NLSF grid -> interpolation residual -> residual energies */
            i32 XX[120];
            struct { i32 stk[84 + 4 * 64 + 8]; i32 wk[66]; };  /* wk: lane 0's NLSF -> LPC conversions (interpolation search: behind the candidates' pool; after the quantiser: beside lane 1's, which borrows stk) */
            i32 Y[2 * 132];
            i16 LPC_res[2 * 96];
         };
      };
      /* the NLSF quantiser (16 survivors, one lane each) borrows all of it: LPC_in_pre is worked out again for the residual energies behind it */
      struct { i32 err_Q24[32], RD_Q25[16], surv[16]; i8 tempIndices2[16 * 16]; SeNlsfTabs tabs; SeNlsfLane lane[16]; };
   };
};

/* NLSFIndices, pNLSF_Q15 (in/out) live in LDS; lanes = survivors */
WV_DEVN void se_nlsf_encode_wave(WV_LDS i8 *NLSFIndices, WV_LDS i16 *pNLSF_Q15, int order, const WV_LDS i16 *pW_Q2, int NLSF_mu_Q20, int nSurvivors, int signalType, WV_LDS SeLpcWork *W)
{
   const SdNlsfCb cb = sd_nlsf_cb(order);
   const u8 *ec_rates_Q5 = order == 16 ? se_nlsf_wb_ec_rates_q5 : se_nlsf_nb_mb_ec_rates_q5;
   const i16 inv_qstep_Q6 = order == 16 ? SE_NLSF_WB_INV_QSTEP_Q6 : SE_NLSF_NB_MB_INV_QSTEP_Q6;
   LANE0 {
      i16 n[16]; for (int i = 0; i < order; i++) n[i] = pNLSF_Q15[i];
      sd_nlsf_stabilize(n, cb.deltamin, order);
      for (int i = 0; i < order; i++) pNLSF_Q15[i] = n[i];
   }
   FOR_LANES(v, cb.nVectors) {                                                                    /* silk_NLSF_VQ: one codebook vector per lane */
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
   wv_sync();
   FOR_LANES(v, cb.nVectors) {                                                                    /* silk_insertion_sort_increasing (stable, first K): every vector finds its own rank */
      const i32 ev = W->err_Q24[v]; int rank = 0;
      for (int u = 0; u < cb.nVectors; u++) { const i32 eu = W->err_Q24[u]; rank += eu < ev || (eu == ev && u < v); }
      if (rank < nSurvivors) W->surv[rank] = v;
   }
   FOR_LANES(i, 20) se_nlsf_out_tabs(&W->tabs, i, cb.qstep);
   wv_sync();
   FOR_LANES(s, nSurvivors) {
      const int ind1 = W->surv[s];
      const u8 *pCB = &cb.cb1_nlsf[ind1 * order]; const i16 *pWg = &cb.wght[ind1 * order];
      WV_LDS SeNlsfLane *w = &W->lane[s];
      for (int i = 0; i < order; i++) {
         const i16 tmp = (i16)shl32((i16)pCB[i], 7);
         const i32 W_tmp_Q9 = pWg[i];
         w->res_Q10[i] = (i16)(sk_mulbb(pNLSF_Q15[i] - tmp, W_tmp_Q9) >> 14);
         w->W_adj_Q5[i] = (i16)sk_div32_varQ((i32)pW_Q2[i], sk_mulbb(W_tmp_Q9, W_tmp_Q9), 21);
      }
      {  /* silk_NLSF_unpack (NLSF_unpack.c:35) into the lane's workspace */
         const u8 *sel = &cb.ec_sel[ind1 * order / 2];
         for (int i = 0; i < order; i += 2) {
            const int entry = *sel++;
            w->ec_ix[i] = ((entry >> 1) & 7) * 9; w->pred_Q8[i] = cb.pred[i + (entry & 1) * (order - 1)];
            w->ec_ix[i + 1] = ((entry >> 5) & 7) * 9; w->pred_Q8[i + 1] = cb.pred[i + ((entry >> 4) & 1) * (order - 1) + 1];
         }
      }
      for (int i = order; i < 16; i++) W->tempIndices2[s * 16 + i] = 0;
      const i32 RD = se_nlsf_del_dec_quant(w, &W->tempIndices2[s * 16], &W->tabs, ec_rates_Q5, inv_qstep_Q6, NLSF_mu_Q20, order);
      const u8 *icdf = &cb.cb1_icdf[(signalType >> 1) * cb.nVectors];
      const int prob_Q8 = ind1 == 0 ? 256 - icdf[ind1] : icdf[ind1 - 1] - icdf[ind1];
      const int bits_q7 = (8 << 7) - se_lin2log(prob_Q8);
      W->RD_Q25[s] = sk_mlabb(RD, bits_q7, NLSF_mu_Q20 >> 2);
   }
   LANE0 {
      int best = 0; i32 bv = W->RD_Q25[0];
      for (int s = 1; s < nSurvivors; s++) if (W->RD_Q25[s] < bv) { bv = W->RD_Q25[s]; best = s; }          /* insertion sort with K = 1: first minimum */
      NLSFIndices[0] = (i8)W->surv[best];
      for (int i = 0; i < order; i++) NLSFIndices[1 + i] = W->tempIndices2[best * 16 + i];
      i16 n[16];
      sd_nlsf_decode(n, NLSFIndices, cb);
      for (int i = 0; i < order; i++) pNLSF_Q15[i] = n[i];
   }
}

/* PredCoef_Q12: LDS [2][16]; pNLSF_Q15 = W->NLSF_Q15 (quantised on return) */
/* CH: OaSilkEncChannel, or the few fields of it the prediction stage needs (SePredChan: the pred kernel of the split path, opus_sh_split.h) */
template <class CH> WV_DEV void se_process_nlsfs_wave(WV_LDS CH *c, WV_LDS i16 *PredCoef_Q12, WV_LDS SeLpcWork *W)
{
   const int order = c->predictLPCOrder;
   int NLSF_mu_Q20 = sk_mlawb(SE_FIX(0.003, 20), SE_FIX(-0.001, 28), c->speech_activity_Q8);
   if (c->nb_subfr == 2) NLSF_mu_Q20 = NLSF_mu_Q20 + (NLSF_mu_Q20 >> 1);
   const int doInterpolate = c->useInterpolatedNLSFs == 1 && c->indices.NLSFInterpCoef_Q2 < 4;
   LANE0 {
      i16 n[16], w[16], n0[16], w0[16];
      for (int i = 0; i < order; i++) n[i] = W->NLSF_Q15[i];
      se_nlsf_vq_weights(w, n, order);
      if (doInterpolate) {
         se_interpolate(n0, c->prev_NLSFq_Q15, n, c->indices.NLSFInterpCoef_Q2, order);
         se_nlsf_vq_weights(w0, n0, order);
         const i16 i_sqr_Q15 = (i16)shl32(sk_mulbb(c->indices.NLSFInterpCoef_Q2, c->indices.NLSFInterpCoef_Q2), 11);
         for (int i = 0; i < order; i++) w[i] = (i16)((w[i] >> 1) + (sk_mulbb(w0[i], i_sqr_Q15) >> 16));
      }
      for (int i = 0; i < order; i++) W->pW[i] = w[i];
   }
   se_nlsf_encode_wave(c->indices.NLSFIndices, W->NLSF_Q15, order, W->pW, NLSF_mu_Q20, c->NLSF_MSVQ_Survivors, c->indices.signalType, W);
   /* the two conversions back to LPC (second half: the quantised NLSFs; first half: their interpolation with last frame's) side by side on lanes 0 and 1 */
   wv_sync();
   {
      const int lane = wv_lane();
      if (lane == 0) sd_nlsf2a_w(PredCoef_Q12 + 16, W->NLSF_Q15, order, W->wk);
      if (lane == 1 && doInterpolate) {
         i16 n0[16];
         for (int i = 0; i < order; i++) n0[i] = (i16)(c->prev_NLSFq_Q15[i] + (sk_mulbb(W->NLSF_Q15[i] - c->prev_NLSFq_Q15[i], c->indices.NLSFInterpCoef_Q2) >> 2));
         sd_nlsf2a_w(PredCoef_Q12, n0, order, W->stk);                          /* the quantiser's work area (same union) is free again */
      }
   }
   wv_sync();
   if (!doInterpolate) { FOR_LANES(i, order) PredCoef_Q12[i] = PredCoef_Q12[16 + i]; }
   wv_sync();
}

/* the interpolation search of silk_find_LPC_FIX (find_LPC_FIX.c:88-138) once the four candidates' LPC coefficients (cand_a[k][16], k/4 of the way from last frame's NLSFs
 * to this frame's second-half NLSFs) are there: the residual energies of the first half frame, measured one candidate after the other by the whole wave, and the reference's
 * running comparison replayed on uniform values.  *res_nrg / *res_nrg_Q: in = the full-frame analysis' energy less the second half's; out = the winner's.  Returns the
 * interpolation index (4: none) */
WV_DEV int se_interp_search_wave(const WV_LDS i16 *x, const WV_LDS i16 *cand_a, WV_LDS i16 *LPC_res, int subfr_length /* incl. the order */, int order, i32 *res_nrg_io, int *res_nrg_Q_io)
{
   i32 res_nrg = *res_nrg_io; int res_nrg_Q = *res_nrg_Q_io, coef = 4;
   for (int k = 3; k >= 0; k--) {
      se_lpc_analysis_filter_wave(LPC_res, x, cand_a + 16 * k, 2 * subfr_length, order);
      wv_sync();
      i32 res_nrg0, res_nrg1; int rshift0, rshift1, res_nrg_interp_Q, isInterpLower;
      se_sum_sqr_shift_wave(&res_nrg0, &rshift0, LPC_res + order, subfr_length - order);
      se_sum_sqr_shift_wave(&res_nrg1, &rshift1, LPC_res + order + subfr_length, subfr_length - order);
      int shift = rshift0 - rshift1;
      if (shift >= 0) { res_nrg1 >>= shift; res_nrg_interp_Q = -rshift0; } else { res_nrg0 >>= -shift; res_nrg_interp_Q = -rshift1; }
      const i32 res_nrg_interp = add32(res_nrg0, res_nrg1);
      shift = res_nrg_interp_Q - res_nrg_Q;
      if (shift >= 0) isInterpLower = (res_nrg_interp >> shift) < res_nrg;
      else if (-shift < 32) isInterpLower = res_nrg_interp < (res_nrg >> -shift);
      else isInterpLower = 0;
      if (isInterpLower) { res_nrg = res_nrg_interp; res_nrg_Q = res_nrg_interp_Q; coef = k; }
      wv_sync();
   }
   *res_nrg_io = res_nrg; *res_nrg_Q_io = res_nrg_Q;
   return coef;
}

/* x = LPC_in_pre; LPC_res: i16[2 * 96] */
template <class CH> WV_DEVN void se_find_lpc_wave(WV_LDS CH *c, WV_LDS SeLpcWork *W, const WV_LDS i16 *x, i32 minInvGain_Q30, WV_LDS i16 *LPC_res, WV_LDS i32 *tk)
{
   const int order = c->predictLPCOrder, subfr_length = c->subfr_length + order;
   const int interp = c->useInterpolatedNLSFs && !c->first_frame_after_reset && c->nb_subfr == 4;
   se_burg_modified_wave(&W->r[0], W->a_Q16, x, minInvGain_Q30, subfr_length, c->nb_subfr, order, W->stk);
   if (interp) se_burg_modified_wave(&W->r[2], W->a_tmp_Q16, x + 2 * subfr_length, minInvGain_Q30, subfr_length, 2, order, W->stk);
   LANE0 {
      i32 res_nrg = W->r[0]; int res_nrg_Q = W->r[1];
      c->indices.NLSFInterpCoef_Q2 = 4;
      if (interp) {
         const i32 res_tmp_nrg = W->r[2]; const int res_tmp_nrg_Q = W->r[3];
         WV_LDS i32 *at = W->a_tmp_Q16;
         const int shift = res_tmp_nrg_Q - res_nrg_Q;
         if (shift >= 0) { if (shift < 32) res_nrg = res_nrg - (res_tmp_nrg >> shift); }
         else { res_nrg = (res_nrg >> -shift) - res_tmp_nrg; res_nrg_Q = res_tmp_nrg_Q; }
      }
      W->r[0] = res_nrg; W->r[1] = res_nrg_Q;
   }
   if (interp) se_a2nlsf_wave(W->NLSF_Q15, W->a_tmp_Q16, order, W->Y);
   SE_TICK(tk, 12);                                                              /* Burg (x2) + A2NLSF of the second half */
   if (interp) {
      /* the four interpolation candidates (k/4 of the way from last frame's NLSFs to this frame's second-half NLSFs): their NLSF -> LPC conversions are serial
       * and independent, so they run side by side on lanes 0..3, each with its own work area behind the residual buffer; the residual energies of the
       * candidates are then measured one after the other by the whole wave, and the reference's running comparison is replayed on uniform values */
      WV_LDS i32 *pool = (WV_LDS i32 *)W->LPC_res + 96;                         /* LPC_res is i16[192]; the union holds 348 words */
      WV_LDS i16 *cand_a = (WV_LDS i16 *)(pool + 3 * 66);                       /* 4 x 16 coefficients */
      const int lane = wv_lane();
      wv_sync();
      if (lane < 4) {
         i16 n[16];
         for (int i = 0; i < order; i++) n[i] = (i16)(c->prev_NLSFq_Q15[i] + (sk_mulbb(W->NLSF_Q15[i] - c->prev_NLSFq_Q15[i], lane) >> 2));   /* silk_interpolate */
         sd_nlsf2a_w(cand_a + 16 * lane, n, order, lane == 0 ? W->wk : pool + (lane - 1) * 66);
      }
      wv_sync();
      i32 res_nrg = W->r[0]; int res_nrg_Q = W->r[1];
      const int coef = se_interp_search_wave(x, cand_a, LPC_res, subfr_length, order, &res_nrg, &res_nrg_Q);
      LANE0 { W->r[0] = res_nrg; W->r[1] = res_nrg_Q; c->indices.NLSFInterpCoef_Q2 = (i8)coef; }
      wv_sync();
   }
   SE_TICK(tk, 13);                                                              /* interpolation search */
   if (c->indices.NLSFInterpCoef_Q2 == 4) se_a2nlsf_wave(W->NLSF_Q15, W->a_Q16, order, W->Y);
}

/* LPC_in_pre (find_pred_coefs_FIX.c:82-101): voiced: the LTP residual of the input scaled by the inverse gains (silk_LTP_analysis_filter_FIX); otherwise the input scaled by them */
WV_DEV void se_lpc_in_pre_wave(WV_LDS OaSilkEncChannel *c, WV_LDS SeEncCtrl *ctl, WV_LDS SeLpcWork *W, WV_LDS i16 *LPC_in_pre, const WV_LDS i16 *x)
{
   const int order = c->predictLPCOrder, nb = c->nb_subfr, sl = c->subfr_length;
   wv_sync();
   if (c->indices.signalType == SE_TYPE_VOICED) se_ltp_analysis_filter_wave(LPC_in_pre, x - order, ctl->LTPCoef_Q14, ctl->pitchL, W->invGains_Q16, sl, nb, order);
   else {
      for (int i = 0; i < nb; i++) {
         const WV_LDS i16 *x_ptr = x - order + i * sl; WV_LDS i16 *o = LPC_in_pre + i * (sl + order); const i32 g = W->invGains_Q16[i];
         FOR_LANES(j, sl + order) o[j] = (i16)sk_mulwb(g, x_ptr[j]);
      }
   }
   wv_sync();
}
/* res_pitch = res_pitch_frame, x = x_frame.  LPC_in_pre: i16[4 * 16 + 320]; XX: i32[100 + 20]; LPC_res: i16[192] */
/* what the prediction stage of the split path's pred kernel starts from (opus_sh_split.h): the outcome of this function's first half -- the LPC analysis' input and its
 * gain bound -- and the few fields of the channel and its control block the second half and silk_process_gains_FIX read */
struct ShPredIn {
   i32 minInvGain_Q30, local_gains[4], LTPredCodGain_Q7, coding_quality_Q14, input_quality_Q14;
   i32 predictLPCOrder, nb_subfr, subfr_length, useInterpolatedNLSFs, first_frame_after_reset, speech_activity_Q8, NLSF_MSVQ_Survivors, SNR_dB_Q7, input_tilt_Q15, nStatesDelayedDecision;
   i16 prev_NLSFq_Q15[16];
   i16 LPC_in_pre[4 * 16 + 320];
   SeBurgCorr bc[2];                   /* pipeline mode 4: the correlations of the two Burg analyses (whole frame; last two subframes), worked out where the signal is in LDS */
};
/* pj != NULL (the split path's front kernel): stop after the LPC analysis' input has been worked out (find_pred_coefs_FIX.c:45-101) and hand it, with the gain bound of :103-113,
 * to the pred kernel, which runs silk_find_LPC_FIX, silk_process_NLSFs, silk_residual_energy_FIX (:115-144) and silk_process_gains_FIX at twice this kernel's occupancy */
WV_DEVN void se_find_pred_coefs_wave(WV_LDS OaSilkEncChannel *c, WV_LDS SeEncCtrl *ctl, const WV_LDS i16 *res_pitch, const WV_LDS i16 *x, int condCoding,
      WV_LDS SeLpcWork *W, WV_LDS i16 *LPC_in_pre, WV_LDS i32 *XX, WV_LDS i16 *LPC_res, WV_LDS i32 *tk, ShPredIn *pj = nullptr, int pj_corr = 0 /* export the Burg correlations too */)
{
   const int order = c->predictLPCOrder, nb = c->nb_subfr, sl = c->subfr_length;
   LANE0 {
      i32 min_gain_Q16 = 2147483647 >> 6;
      for (int i = 0; i < nb; i++) min_gain_Q16 = imin(min_gain_Q16, ctl->Gains_Q16[i]);
      for (int i = 0; i < nb; i++) {
         i32 ig = sk_div32_varQ(min_gain_Q16, ctl->Gains_Q16[i], 16 - 2);
         ig = imax(ig, 100);
         W->invGains_Q16[i] = ig;
         W->local_gains[i] = ((i32)1 << 16) / ig;
      }
   }
   if (c->indices.signalType == SE_TYPE_VOICED) {
      wv_sync();
      se_find_ltp_wave(XX + 20, XX, res_pitch, ctl->pitchL, sl, nb);
      se_quant_ltp_gains_wave(ctl->LTPCoef_Q14, c->indices.LTPIndex, &c->indices.PERIndex, &c->sum_log_gain_Q7, &ctl->LTPredCodGain_Q7, XX + 20, XX, sl, nb);
      LANE0 se_ltp_scale_ctrl(c, ctl, condCoding);
      se_lpc_in_pre_wave(c, ctl, W, LPC_in_pre, x);
   } else {
      se_lpc_in_pre_wave(c, ctl, W, LPC_in_pre, x);
      LANE0 { for (int i = 0; i < nb * 5; i++) ctl->LTPCoef_Q14[i] = 0; ctl->LTPredCodGain_Q7 = 0; c->sum_log_gain_Q7 = 0; ctl->LTP_scale_Q14 = 0; }
   }
   wv_sync();
   SE_TICK(tk, 11);                                                              /* LTP analysis */
   i32 minInvGain_Q30;
   if (c->first_frame_after_reset) minInvGain_Q30 = SE_FIX(1.0f / 1e2f, 30);
   else {
      minInvGain_Q30 = se_log2lin(sk_mlawb(16 << 7, (i32)ctl->LTPredCodGain_Q7, SE_FIX(1.0 / 3, 16)));
      minInvGain_Q30 = sk_div32_varQ(minInvGain_Q30, sk_mulww(SE_FIX(1e4f, 0), sk_mlawb(SE_FIX(0.25, 18), SE_FIX(0.75, 18), ctl->coding_quality_Q14)), 14);
   }
   if (pj) {
      wv_sync();
      FOR_LANES(i, nb * (sl + order)) pj->LPC_in_pre[i] = LPC_in_pre[i];
      FOR_LANES(i, 16) pj->prev_NLSFq_Q15[i] = c->prev_NLSFq_Q15[i];
      FOR_LANES(i, 4) pj->local_gains[i] = W->local_gains[i];
      if (wv_lane() == 0) {
         pj->minInvGain_Q30 = minInvGain_Q30; pj->LTPredCodGain_Q7 = ctl->LTPredCodGain_Q7; pj->coding_quality_Q14 = ctl->coding_quality_Q14; pj->input_quality_Q14 = ctl->input_quality_Q14;
         pj->predictLPCOrder = order; pj->nb_subfr = nb; pj->subfr_length = sl; pj->useInterpolatedNLSFs = c->useInterpolatedNLSFs; pj->first_frame_after_reset = c->first_frame_after_reset;
         pj->speech_activity_Q8 = c->speech_activity_Q8; pj->NLSF_MSVQ_Survivors = c->NLSF_MSVQ_Survivors; pj->SNR_dB_Q7 = c->SNR_dB_Q7; pj->input_tilt_Q15 = c->input_tilt_Q15;
         pj->nStatesDelayedDecision = c->nStatesDelayedDecision;
      }
      wv_sync();
      if (pj_corr) {
         se_burg_corr_wave(&pj->bc[0], LPC_in_pre, sl + order, nb, order, W->stk);
         if (c->useInterpolatedNLSFs && !c->first_frame_after_reset && nb == 4) se_burg_corr_wave(&pj->bc[1], LPC_in_pre + 2 * (sl + order), sl + order, 2, order, W->stk);
      }
      return;
   }
   se_find_lpc_wave(c, W, LPC_in_pre, minInvGain_Q30, LPC_res, tk);
   SE_TICK(tk, 14);                                                              /* final A2NLSF */
   se_process_nlsfs_wave(c, &ctl->PredCoef_Q12[0][0], W);
   SE_TICK(tk, 15);                                                              /* NLSF quantiser + NLSF2A */
   se_lpc_in_pre_wave(c, ctl, W, LPC_in_pre, x);                                 /* (the quantiser has worked in its bytes) */
   se_residual_energy_wave(ctl->ResNrg, ctl->ResNrgQ, LPC_in_pre, &ctl->PredCoef_Q12[0][0], W->local_gains, sl, nb, order, LPC_res);
   LANE0 { for (int i = 0; i < 16; i++) c->prev_NLSFq_Q15[i] = i < order ? W->NLSF_Q15[i] : (i16)0; }
}

/* ---- gains ---- */
#define SE_GQ_OFFSET 2090
#define SE_GQ_SCALE_Q16 2251
#define SE_GQ_INV_SCALE_Q16 1907825
WV_DEV void se_gains_quant(WV_LDS i8 *ind, WV_LDS i32 *gain_Q16, WV_LDS i32 *prev_ind_p, int conditional, int nb_subfr)
{
   int prev_ind = (i8)*prev_ind_p;
   for (int k = 0; k < nb_subfr; k++) {
      int v = (i8)sk_mulwb(SE_GQ_SCALE_Q16, se_lin2log(gain_Q16[k]) - SE_GQ_OFFSET);
      if (v < prev_ind) v = (i8)(v + 1);
      v = se_limit(v, 0, 63);
      if (k == 0 && conditional == 0) { v = se_limit(v, prev_ind - 4, 63); prev_ind = v; }
      else {
         v = (i8)(v - prev_ind);
         const int thr = 2 * 36 - 64 + prev_ind;
         if (v > thr) v = (i8)(thr + ((v - thr + 1) >> 1));
         v = se_limit(v, -4, 36);
         if (v > thr) { prev_ind = (i8)(prev_ind + (shl32(v, 1) - thr)); prev_ind = imin(prev_ind, 63); } else prev_ind = (i8)(prev_ind + v);
         v -= -4;
      }
      ind[k] = (i8)v;
      gain_Q16[k] = se_log2lin(imin(sk_mulwb(SE_GQ_INV_SCALE_Q16, prev_ind) + SE_GQ_OFFSET, 3967));
   }
   *prev_ind_p = prev_ind;
}
WV_DEV void se_gains_dequant(i32 *gain_Q16, const i8 *ind, int *prev_ind_p, int conditional, int nb_subfr)      /* gain_quant.c:94 */
{
   int prev_ind = *prev_ind_p;
   for (int k = 0; k < nb_subfr; k++) {
      if (k == 0 && conditional == 0) prev_ind = imax(ind[k], prev_ind - 16);
      else { const int t = ind[k] - 4, thr = 2 * 36 - 64 + prev_ind; if (t > thr) prev_ind += shl32(t, 1) - thr; else prev_ind += t; }
      prev_ind = se_limit(prev_ind, 0, 63);
      gain_Q16[k] = se_log2lin(imin(sk_mulwb(SE_GQ_INV_SCALE_Q16, prev_ind) + SE_GQ_OFFSET, 3967));
   }
   *prev_ind_p = prev_ind;
}
WV_DEV i32 se_gains_ID(const WV_LDS i8 *ind, int nb_subfr) { i32 id = 0; for (int k = 0; k < nb_subfr; k++) id = add32(ind[k], shl32(id, 8)); return id; }

template <class CH, class CT> WV_DEV void se_process_gains_l0(WV_LDS CH *c, WV_LDS CT *ctl, int condCoding)
{
   if (c->indices.signalType == SE_TYPE_VOICED) {
      const i32 s_Q16 = -se_sigm_Q15(sk_rround(ctl->LTPredCodGain_Q7 - SE_FIX(12.0, 7), 4));
      for (int k = 0; k < c->nb_subfr; k++) ctl->Gains_Q16[k] = sk_mlawb(ctl->Gains_Q16[k], ctl->Gains_Q16[k], s_Q16);
   }
   const i32 InvMaxSqrVal_Q16 = se_log2lin(sk_mulwb(SE_FIX(21 + 16 / 0.33, 7) - c->SNR_dB_Q7, SE_FIX(0.33, 16))) / c->subfr_length;
   for (int k = 0; k < c->nb_subfr; k++) {
      i32 ResNrgPart = sk_mulww(ctl->ResNrg[k], InvMaxSqrVal_Q16);
      if (ctl->ResNrgQ[k] > 0) ResNrgPart = sk_rround(ResNrgPart, ctl->ResNrgQ[k]);
      else if (ResNrgPart >= (2147483647 >> -ctl->ResNrgQ[k])) ResNrgPart = 2147483647;
      else ResNrgPart = shl32(ResNrgPart, -ctl->ResNrgQ[k]);
      i32 gain = ctl->Gains_Q16[k];
      i32 gain_squared = sk_add_sat(ResNrgPart, sk_mulhi(gain, gain));
      if (gain_squared < 32767) {
         gain_squared = sk_mlaww(shl32(ResNrgPart, 16), gain, gain);
         gain = se_sqrt_approx(gain_squared); gain = imin(gain, 2147483647 >> 8);
         ctl->Gains_Q16[k] = sk_shl_sat(gain, 8);
      } else { gain = se_sqrt_approx(gain_squared); gain = imin(gain, 2147483647 >> 16); ctl->Gains_Q16[k] = sk_shl_sat(gain, 16); }
   }
   for (int k = 0; k < c->nb_subfr; k++) ctl->GainsUnq_Q16[k] = ctl->Gains_Q16[k];
   ctl->lastGainIndexPrev = c->LastGainIndex;
   se_gains_quant(c->indices.GainsIndices, ctl->Gains_Q16, &c->LastGainIndex, condCoding == SE_CODE_CONDITIONALLY, c->nb_subfr);
   if (c->indices.signalType == SE_TYPE_VOICED) c->indices.quantOffsetType = ctl->LTPredCodGain_Q7 + (c->input_tilt_Q15 >> 8) > SE_FIX(1.0, 7) ? 0 : 1;
   const i32 quant_offset_Q10 = se_quantization_offsets_q10[(c->indices.signalType >> 1) * 2 + c->indices.quantOffsetType];
   ctl->Lambda_Q10 = SE_FIX(1.2f, 10) + sk_mulbb(SE_FIX(-0.05f, 10), c->nStatesDelayedDecision) + sk_mulwb(SE_FIX(-0.2f, 18), c->speech_activity_Q8)
      + sk_mulwb(SE_FIX(-0.1f, 12), ctl->input_quality_Q14) + sk_mulwb(SE_FIX(-0.2f, 12), ctl->coding_quality_Q14) + sk_mulwb(SE_FIX(0.8f, 16), quant_offset_Q10);
}
#endif

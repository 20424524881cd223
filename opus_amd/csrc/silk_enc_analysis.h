/* silk_enc_analysis.h — SILK encoder analysis stages (row a20 of SURVEY §8): noise shaping, LTP and LPC analysis.
 *
 *   se_warped_autocorr2_wave     silk_warped_autocorrelation_FIX_c   silk/fixed/warped_autocorrelation_FIX.c:40 (systolic: lane = ladder stage; two sub-frames per pass)
 *   se_schur64 / se_k2a_Q16      silk_schur64 / silk_k2a_Q16         silk/fixed/schur64_FIX.c:36, k2a_Q16_FIX.c:36
 *   se_noise_shape_analysis      silk_noise_shape_analysis_FIX       silk/fixed/noise_shape_analysis_FIX.c:147 (warped_gain :38, limit_warped_coefs :59)
 *   se_burg_modified_wave        silk_burg_modified_c                silk/fixed/burg_modified_FIX.c:46
 *   se_find_ltp_wave             silk_find_LTP_FIX                   silk/fixed/find_LTP_FIX.c:36 (corrMatrix_FIX.c:40,:83); XX: i32[nb*25], xX: i32[nb*5]
 *   se_quant_ltp_gains_wave      silk_quant_LTP_gains, silk_VQ_WMat_EC_c   silk/quant_LTP_gains.c:35, silk/VQ_WMat_EC.c:35 (one lane per codebook vector; _l0: the serial form,
 *                                                                    kept for the one corner the wave form hands back to it)
 *   se_sum_sqr_shift_wave        silk_sum_sqr_shift                  silk/sum_sqr_shift.c:36
 *   se_ltp_scale_ctrl            silk_LTP_scale_ctrl_FIX             silk/fixed/LTP_scale_ctrl_FIX.c:36
 *   se_ltp_analysis_filter_wave  silk_LTP_analysis_filter_FIX        silk/fixed/LTP_analysis_filter_FIX.c:36
 *   se_residual_energy           silk_residual_energy_FIX            silk/fixed/residual_energy_FIX.c:37
 *   se_find_lpc                  silk_find_LPC_FIX                   silk/fixed/find_LPC_FIX.c:38
 *   se_find_pred_coefs           silk_find_pred_coefs_FIX            silk/fixed/find_pred_coefs_FIX.c:36 */
#ifndef OPUS_AMD_SILK_ENC_ANALYSIS_H
#define OPUS_AMD_SILK_ENC_ANALYSIS_H

WV_DEV int se_clz64(i64 in) { const i32 up = (i32)(in >> 32); return up == 0 ? 32 + sk_clz((i32)in) : sk_clz(up); }
WV_DEV i32 se_add_lshift32(i32 a, i32 b, int s) { return add32(a, shl32(b, s)); }
template <class PA> WV_DEV void se_bwexpander_32(PA ar, int d, i32 chirp_Q16)                                    /* bwexpander_32.c:37 */
{
   const i32 cm1 = chirp_Q16 - 65536;
   for (int i = 0; i < d - 1; i++) { ar[i] = sk_mulww(chirp_Q16, ar[i]); chirp_Q16 += sk_rround(chirp_Q16 * cm1, 16); }
   ar[d - 1] = sk_mulww(chirp_Q16, ar[d - 1]);
}
/* silk_LPC_fit (silk/LPC_fit.c:35) */
template <class PO, class PI> WV_DEV void se_lpc_fit(PO a_QOUT, PI a_QIN, int QOUT, int QIN, int d)
{
   int i, idx = 0;
   for (i = 0; i < 10; i++) {
      i32 maxabs = 0;
      for (int k = 0; k < d; k++) { const i32 av = iabs(a_QIN[k]); if (av > maxabs) { maxabs = av; idx = k; } }
      maxabs = sk_rround(maxabs, QIN - QOUT);
      if (maxabs > 32767) {
         maxabs = imin(maxabs, 163838);
         const i32 chirp_Q16 = SE_FIX(0.999, 16) - shl32(maxabs - 32767, 14) / ((maxabs * (idx + 1)) >> 2);
         se_bwexpander_32(a_QIN, d, chirp_Q16);
      } else break;
   }
   if (i == 10) for (int k = 0; k < d; k++) { a_QOUT[k] = (i16)sk_sat16(sk_rround(a_QIN[k], QIN - QOUT)); a_QIN[k] = shl32((i32)a_QOUT[k], QIN - QOUT); }
   else for (int k = 0; k < d; k++) a_QOUT[k] = (i16)sk_rround(a_QIN[k], QIN - QOUT);
}

/* C: [order + 1][2] words of LDS */
template <class PR, class PC> WV_DEV i32 se_schur64(PR rc_Q16, PC c, int order, WV_LDS i32 (*C)[2])
{
   int k;
   if (c[0] <= 0) { for (k = 0; k < order; k++) rc_Q16[k] = 0; return 0; }
   for (k = 0; k <= order; k++) C[k][0] = C[k][1] = c[k];
   for (k = 0; k < order; k++) {
      if (iabs(C[k + 1][0]) >= C[0][1]) { rc_Q16[k] = C[k + 1][0] > 0 ? -SE_FIX(.99f, 16) : SE_FIX(.99f, 16); k++; break; }
      const i32 rc_tmp_Q31 = sk_div32_varQ(-C[k + 1][0], C[0][1], 31);
      rc_Q16[k] = sk_rround(rc_tmp_Q31, 15);
      for (int n = 0; n < order - k; n++) {
         const i32 t1 = C[n + k + 1][0], t2 = C[n][1];
         C[n + k + 1][0] = t1 + sk_mulhi(shl32(t2, 1), rc_tmp_Q31);
         C[n][1] = t2 + sk_mulhi(shl32(t1, 1), rc_tmp_Q31);
      }
   }
   for (; k < order; k++) rc_Q16[k] = 0;
   return imax(1, C[0][1]);
}
template <class PA, class PR> WV_DEV void se_k2a_Q16(PA A_Q24, PR rc_Q16, int order)
{
   for (int k = 0; k < order; k++) {
      const i32 rc = rc_Q16[k];
      for (int n = 0; n < (k + 1) >> 1; n++) { const i32 t1 = A_Q24[n], t2 = A_Q24[k - n - 1]; A_Q24[n] = sk_mlaww(t1, t2, rc); A_Q24[k - n - 1] = sk_mlaww(t2, t1, rc); }
      A_Q24[k] = -shl32(rc, 8);
   }
}
/* silk_schur64 (silk/fixed/schur64_FIX.c:35) with one lane per correlation pair: lane n keeps C[n + k + 1][0] and C[n][1]; the updates of one order are
 * independent across n, the upper row slides down one lane per order (DPP), the reflection coefficient comes from lane 0's pair.  rc_Q16: LDS [order]. */
WV_DEV i32 se_schur64_wave(WV_LDS i32 *rc_Q16, const WV_LDS i32 *c, int order)
{
   const int lane = wv_lane();
   if (c[0] <= 0) { wv_sync(); FOR_LANES(k, order) rc_Q16[k] = 0; wv_sync(); return 0; }
   i32 a = lane < order ? c[lane + 1] : 0, b = lane <= order ? c[lane] : 0;
   wv_sync();
   int k;
   for (k = 0; k < order; k++) {
      const i32 a0 = wv_lane_const<0>(a), b0 = wv_lane_const<0>(b);
      if (iabs(a0) >= b0) { if (lane == 0) rc_Q16[k] = a0 > 0 ? -SE_FIX(.99f, 16) : SE_FIX(.99f, 16); k++; break; }
      const i32 rc_tmp_Q31 = sk_div32_varQ(-a0, b0, 31);
      if (lane == 0) rc_Q16[k] = sk_rround(rc_tmp_Q31, 15);
      const i32 na = a + sk_mulhi(shl32(b, 1), rc_tmp_Q31), nb = b + sk_mulhi(shl32(a, 1), rc_tmp_Q31);
      if (lane < order - k) { a = na; b = nb; }
      a = wv_shift_down1(a, 0);
   }
   if (lane >= k && lane < order) rc_Q16[lane] = 0;
   const i32 nrg = imax(1, wv_lane_const<0>(b));
   wv_sync();
   return nrg;
}
/* silk_k2a_Q16 (silk/k2a_Q16.c:35), lane n = coefficient n: step k pairs n with k - 1 - n (both read the old values), A_Q24: LDS [order] */
WV_DEV void se_k2a_Q16_wave(WV_LDS i32 *A_Q24, const WV_LDS i32 *rc_Q16, int order)
{
   const int lane = wv_lane();
   i32 a = 0;
   for (int k = 0; k < order; k++) {
      const i32 rc = rc_Q16[k];
      const i32 other = wv_shfl(a, (k - 1 - lane) & 63);
      if (lane < k) a = sk_mlaww(a, other, rc);
      if (lane == k) a = -shl32(rc, 8);
   }
   wv_sync();
   if (lane < order) A_Q24[lane] = a;
   wv_sync();
}
/* ---- two sub-frames per pass (noise shaping analysis): lanes 0..31 work on sub-frame a, lanes 32..63 on sub-frame b.  order <= 24, so a chain / a correlation row fits one half. ---- */
WV_DEV i32 wv_half_head(i32 v) { const i32 a = wv_lane_const<0>(v), b = wv_lane_const<32>(v); return wv_lane() < 32 ? a : b; }
/* silk_warped_autocorrelation_FIX_c (silk/fixed/warped_autocorrelation_FIX.c:40) as a systolic chain, two windowed sub-frames at once: lane i of a half owns stage i of the warped
 * allpass ladder and runs i samples behind the half's head lane, so one wave step advances every stage (length + order steps instead of length x order).
 * in_{i}(n) = in_{i-1}(n-1) + SMLAWB(in_i(n-1) - in_{i-1}(n), w) is exactly the reference's tmp1 / tmp2 recursion; every corr[i] accumulates its 64-bit products in the reference's
 * sample order.  QC = 10, QS = 13 (silk/fixed/main_FIX.h:49-50).  The window (sine slopes from wtab, silk/fixed/apply_sine_window_FIX.c:36 through se_sine_window_table) is applied as the
 * samples are fetched, 64 per half at a time into a register per lane; the head lane of each half takes its next sample from there (v_readlane), the sample then travels down the
 * chain beside the stage value (DPP), so the loop touches no memory.  Stages that have not started see zeros from above and stages that have finished are fed zero samples: no lane
 * needs a predicate, a product with a sample outside [0, length) is zero.  Returns the lane's half's scale; corr_a / corr_b: LDS [order + 1]. */
WV_DEV int se_warped_autocorr2_wave(WV_LDS i32 *corr_a, WV_LDS i32 *corr_b, const WV_LDS i16 *x_a, const WV_LDS i16 *x_b, const WV_LDS i32 *wtab, int slope_part, int flat_part,
      int warping_Q16, int length, int order)
{
   const int lane = wv_lane(), l = lane & 31;
   const bool head = l == 0, hb = lane >= 32;
   i32 cur = 0, prev_stage_prev = 0, my = 0, xs = 0;
   i64 acc = 0;
   const int total = length + order;
   for (int base = 0; base < total; base += 64) {
      i32 in_a = 0, in_b = 0;
      {
         const int i = base + lane;
         if (i < length) {
            i32 va = x_a[i], vb = x_b[i];
            if (i < slope_part) { const i32 w = wtab[i]; va = sk_mulwb(w, va); vb = sk_mulwb(w, vb); }
            else if (i >= slope_part + flat_part) { const i32 w = wtab[i - flat_part]; va = sk_mulwb(w, va); vb = sk_mulwb(w, vb); }
            in_a = shl32((i32)(i16)va, 13); in_b = shl32((i32)(i16)vb, 13);
         }
      }
      const int nsteps = imin(64, total - base);
      for (int j = 0; j < nsteps; j++) {
         const i32 from_prev = wv_shift_up1(my, 0), x_prev = wv_shift_up1(xs, 0);
         const i32 xa = wv_bcast(in_a, j), xb = wv_bcast(in_b, j);
         const i32 x13 = head ? (hb ? xb : xa) : x_prev;
         const i32 v = head ? x13 : add32(prev_stage_prev, sk_mulwb(sub32(cur, from_prev), warping_Q16));
         prev_stage_prev = from_prev; cur = v; my = v; xs = x13;
         acc += ((i64)v * (i64)x13) >> (2 * 13 - 10);
      }
   }
   const i32 hi0 = wv_half_head((i32)(acc >> 32)), lo0 = wv_half_head((i32)acc);
   int lsh = se_clz64((i64)(((u64)(u32)hi0 << 32) | (u32)lo0)) - 35;
   lsh = se_limit(lsh, -12 - 10, 30 - 10);
   wv_sync();
   if (l <= order) (hb ? corr_b : corr_a)[l] = lsh >= 0 ? (i32)(acc << lsh) : (i32)(acc >> -lsh);
   wv_sync();
   return -(10 + lsh);
}
/* se_schur64_wave on both halves; a half whose recursion stops early (schur64_FIX.c:58) idles through the remaining orders.  Returns the lane's half's residual energy. */
WV_DEV i32 se_schur64_wave2(WV_LDS i32 *rc_a, WV_LDS i32 *rc_b, const WV_LDS i32 *c_a, const WV_LDS i32 *c_b, int order)
{
   const int lane = wv_lane(), l = lane & 31;
   const bool hb = lane >= 32;
   const WV_LDS i32 *c = hb ? c_b : c_a;
   WV_LDS i32 *rc = hb ? rc_b : rc_a;
   const bool dead = c[0] <= 0;
   i32 a = l < order ? c[l + 1] : 0, b = l <= order ? c[l] : 0;
   wv_sync();
   bool done = dead;
   int kend = dead ? 0 : order;
   for (int k = 0; k < order; k++) {
      const i32 a0 = wv_half_head(a), b0 = wv_half_head(b);
      const bool brk = !done && iabs(a0) >= b0, upd = !done && !brk;
      const i32 rc_tmp_Q31 = sk_div32_varQ(upd ? -a0 : 0, upd ? b0 : 1, 31);
      if (l == 0) { if (brk) rc[k] = a0 > 0 ? -SE_FIX(.99f, 16) : SE_FIX(.99f, 16); else if (upd) rc[k] = sk_rround(rc_tmp_Q31, 15); }
      if (brk) { done = true; kend = k + 1; }
      const i32 na = a + sk_mulhi(shl32(b, 1), rc_tmp_Q31), nb = b + sk_mulhi(shl32(a, 1), rc_tmp_Q31);
      if (upd && l < order - k) { a = na; b = nb; }
      const i32 down = wv_shift_down1(a, 0);
      if (upd) a = down;
   }
   if (l >= kend && l < order) rc[l] = 0;
   const i32 nrg = dead ? 0 : imax(1, wv_half_head(b));
   wv_sync();
   return nrg;
}
WV_DEV void se_k2a_Q16_wave2(WV_LDS i32 *A_a, WV_LDS i32 *A_b, const WV_LDS i32 *rc_a, const WV_LDS i32 *rc_b, int order)
{
   const int lane = wv_lane(), l = lane & 31;
   const bool hb = lane >= 32;
   const WV_LDS i32 *rcp = hb ? rc_b : rc_a;
   i32 a = 0;
   for (int k = 0; k < order; k++) {
      const i32 rc = rcp[k];
      const i32 other = wv_shfl(a, ((k - 1 - l) & 31) | (lane & 32));
      if (l < k) a = sk_mlaww(a, other, rc);
      if (l == k) a = -shl32(rc, 8);
   }
   wv_sync();
   if (l < order) (hb ? A_b : A_a)[l] = a;
   wv_sync();
}
template <class PA> WV_DEV i32 se_warped_gain(PA coefs_Q24, int lambda_Q16, int order)
{
   lambda_Q16 = -lambda_Q16;
   i32 gain_Q24 = coefs_Q24[order - 1];
   for (int i = order - 2; i >= 0; i--) gain_Q24 = sk_mlawb(coefs_Q24[i], gain_Q24, lambda_Q16);
   gain_Q24 = sk_mlawb(SE_FIX(1.0, 24), gain_Q24, -lambda_Q16);
   return sk_inverse32_varQ(gain_Q24, 40);
}
template <class PA> WV_DEV void se_limit_warped_coefs(PA coefs_Q24, int lambda_Q16, i32 limit_Q24, int order)
{
   int ind = 0;
   lambda_Q16 = -lambda_Q16;
   for (int i = order - 1; i > 0; i--) coefs_Q24[i - 1] = sk_mlawb(coefs_Q24[i - 1], coefs_Q24[i], lambda_Q16);
   lambda_Q16 = -lambda_Q16;
   i32 nom_Q16 = sk_mlawb(SE_FIX(1.0, 16), -(i32)lambda_Q16, lambda_Q16), den_Q24 = sk_mlawb(SE_FIX(1.0, 24), coefs_Q24[0], lambda_Q16);
   i32 gain_Q16 = sk_div32_varQ(nom_Q16, den_Q24, 24);
   for (int i = 0; i < order; i++) coefs_Q24[i] = sk_mulww(gain_Q16, coefs_Q24[i]);
   const i32 limit_Q20 = limit_Q24 >> 4;
   for (int iter = 0; iter < 10; iter++) {
      i32 maxabs_Q24 = -1;
      for (int i = 0; i < order; i++) { const i32 t = iabs(coefs_Q24[i]); if (t > maxabs_Q24) { maxabs_Q24 = t; ind = i; } }
      const i32 maxabs_Q20 = maxabs_Q24 >> 4;
      if (maxabs_Q20 <= limit_Q20) return;
      for (int i = 1; i < order; i++) coefs_Q24[i - 1] = sk_mlawb(coefs_Q24[i - 1], coefs_Q24[i], lambda_Q16);
      gain_Q16 = sk_inverse32_varQ(gain_Q16, 32);
      for (int i = 0; i < order; i++) coefs_Q24[i] = sk_mulww(gain_Q16, coefs_Q24[i]);
      const i32 chirp_Q16 = SE_FIX(0.99, 16) - sk_div32_varQ(sk_mulwb(maxabs_Q20 - limit_Q20, sk_mlabb(SE_FIX(0.8, 10), SE_FIX(0.1, 10), iter)), maxabs_Q20 * (ind + 1), 22);
      se_bwexpander_32(coefs_Q24, order, chirp_Q16);
      lambda_Q16 = -lambda_Q16;
      for (int i = order - 1; i > 0; i--) coefs_Q24[i - 1] = sk_mlawb(coefs_Q24[i - 1], coefs_Q24[i], lambda_Q16);
      lambda_Q16 = -lambda_Q16;
      nom_Q16 = sk_mlawb(SE_FIX(1.0, 16), -(i32)lambda_Q16, lambda_Q16); den_Q24 = sk_mlawb(SE_FIX(1.0, 24), coefs_Q24[0], lambda_Q16);
      gain_Q16 = sk_div32_varQ(nom_Q16, den_Q24, 24);
      for (int i = 0; i < order; i++) coefs_Q24[i] = sk_mulww(gain_Q16, coefs_Q24[i]);
   }
}

/* pitch_res = res_pitch_frame, x = x_frame; xw: i16[240] windowed signal, xx: i16[240], w32: i32[28] (auto-correlation, [26] = SNR hand-off) */
/* stk: 100 words of lane-0 working arrays in LDS */
/* silk_sum_sqr_shift (sum_sqr_shift.c:36) on the wave: both passes are sums of individually shifted pair energies (mod 2^32) -> lanes over pairs + one reduction per pass. */
WV_DEV void se_sum_sqr_shift_wave(i32 *energy, int *shift, const WV_LDS i16 *x, int len)
{
   int shft = 31 - sk_clz(len);
   i32 nrg = len;
   for (int pass = 0; pass < 2; pass++) {
      if (pass) { shft = imax(0, shft + 3 - sk_clz(nrg)); nrg = 0; }
      u32 part = 0;
      FOR_LANES(p, (len + 1) >> 1) {
         const int i = 2 * p;
         u32 t = (u32)((i32)x[i] * x[i]);
         if (i + 1 < len) t += (u32)((i32)x[i + 1] * x[i + 1]);
         part += t >> shft;
      }
      nrg = (i32)((u32)nrg + wv_sumu(part));
   }
   *shift = shft; *energy = nrg;
}
WV_DEVN void se_noise_shape_analysis_wave(WV_LDS OaSilkEncChannel *c, WV_LDS SeEncCtrl *ctl, const WV_LDS i16 *pitch_res, const WV_LDS i16 *x, WV_LDS i16 *xw, WV_LDS i16 *xx, WV_LDS i32 *w32, WV_LDS i32 *stk)
{
   const WV_LDS i16 *x_ptr = x - c->la_shape;
   const int order = c->shapingLPCOrder, swl = c->shapeWinLength;
   SE_LTIC();
   LANE0 {
      i32 SNR_adj_dB_Q7 = c->SNR_dB_Q7;
      ctl->input_quality_Q14 = ((i32)c->input_quality_bands_Q15[0] + c->input_quality_bands_Q15[1]) >> 2;
      ctl->coding_quality_Q14 = se_sigm_Q15(sk_rround(SNR_adj_dB_Q7 - SE_FIX(20.0, 7), 4)) >> 1;
      if (c->useCBR == 0) {
         i32 b_Q8 = SE_FIX(1.0, 8) - c->speech_activity_Q8;
         b_Q8 = sk_mulwb(shl32(b_Q8, 8), b_Q8);
         SNR_adj_dB_Q7 = sk_mlawb(SNR_adj_dB_Q7, sk_mulbb(SE_FIX(-2.0f, 7) >> (4 + 1), b_Q8), sk_mulwb(SE_FIX(1.0, 14) + ctl->input_quality_Q14, ctl->coding_quality_Q14));
      }
      if (c->indices.signalType == SE_TYPE_VOICED) SNR_adj_dB_Q7 = sk_mlawb(SNR_adj_dB_Q7, SE_FIX(2.0f, 8), c->LTPCorr_Q15);
      else SNR_adj_dB_Q7 = sk_mlawb(SNR_adj_dB_Q7, sk_mlawb(SE_FIX(6.0, 9), -SE_FIX(0.4, 18), c->SNR_dB_Q7), SE_FIX(1.0, 14) - ctl->input_quality_Q14);
      if (c->indices.signalType == SE_TYPE_VOICED) c->indices.quantOffsetType = 0;
      w32[26] = SNR_adj_dB_Q7;
   }
   if (c->indices.signalType != SE_TYPE_VOICED) {                                  /* sparseness: how much the 2 ms segment energies of the residual jump about */
      const int nSamples = shl32(c->fs_kHz, 1), nSegs = sk_mulbb(5, c->nb_subfr) / 2;
      i32 energy_variation_Q7 = 0, log_energy_prev_Q7 = 0;
      for (int k = 0; k < nSegs; k++) {
         i32 nrg; int scale;
         se_sum_sqr_shift_wave(&nrg, &scale, pitch_res + k * nSamples, nSamples);
         nrg += nSamples >> scale;
         const i32 log_energy_Q7 = se_lin2log(nrg);
         if (k > 0) energy_variation_Q7 += iabs(log_energy_Q7 - log_energy_prev_Q7);
         log_energy_prev_Q7 = log_energy_Q7;
      }
      LANE0 c->indices.quantOffsetType = energy_variation_Q7 > SE_FIX(0.6f, 7) * (nSegs - 1) ? 0 : 1;
   }
   SE_LTOC(17);
   i32 strength_Q16 = sk_mulwb(ctl->predGain_Q16, SE_FIX(1e-3f, 16));
   const i32 BWExp_Q16 = sk_div32_varQ(SE_FIX(0.94f, 16), sk_mlaww(SE_FIX(1.0, 16), strength_Q16, strength_Q16), 16);
   const int warping_Q16 = c->warping_Q16 > 0 ? sk_mlawb(c->warping_Q16, (i32)ctl->coding_quality_Q14, SE_FIX(0.01, 18)) : 0;
   if (c->warping_Q16 > 0) {
      /* the warped path (complexity >= 4): the sub-frames are independent up to the gain smoothing below -> two per pass, one per half of the wave: the allpass ladder, the
       * 64-bit Schur recursion, the step-up and the serial coefficient conditioning (lanes 0 and 32) all run for sub-frames 2p and 2p + 1 at once */
      const int flat_part = c->fs_kHz * 3, slope_part = (swl - flat_part) >> 1;
      WV_LDS i32 *wtab = (WV_LDS i32 *)xx;                                             /* the two window slopes, worked out once (lanes 0 and 1) */
      WV_LDS i32 *const corr_b = (WV_LDS i32 *)xw, *const rc_b = corr_b + 32, *const AR_b = corr_b + 56;      /* the second half's rows: the windowed-signal buffer is free on this path */
      const int lane = wv_lane(), l = lane & 31;
      const bool hb = lane >= 32;
      wv_sync();
      if (lane < 2) se_sine_window_table(wtab + lane * slope_part, lane + 1, slope_part);
      wv_sync();
      for (int p = 0; p < c->nb_subfr; p += 2) {
         const int scale = se_warped_autocorr2_wave(w32, corr_b, x_ptr, x_ptr + c->subfr_length, wtab, slope_part, flat_part, warping_Q16, swl, order);
         x_ptr += 2 * c->subfr_length;
         WV_LDS i32 *const corr = hb ? corr_b : w32;
         if (l == 0) corr[0] = add32(corr[0], imax(sk_mulwb(corr[0] >> 4, SE_FIX(3e-5f, 20)), 1));
         wv_sync();
         SE_LTOC(18);
         const i32 nrg_w = se_schur64_wave2(stk, rc_b, w32, corr_b, order);
         se_k2a_Q16_wave2(stk + 24, AR_b, stk, rc_b, order);
         SE_LTOC(19);
         if (l == 0) {
            const int k = p + (hb ? 1 : 0);
            WV_LDS i32 *AR_Q24 = hb ? AR_b : stk + 24;
            i32 nrg = nrg_w;
            int Qnrg = -scale;
            if (Qnrg & 1) { Qnrg -= 1; nrg >>= 1; }
            const i32 tmp32 = se_sqrt_approx(nrg);
            Qnrg >>= 1;
            i32 g = sk_shl_sat(tmp32, 16 - Qnrg);
            const i32 gain_mult_Q16 = se_warped_gain(AR_Q24, warping_Q16, order);
            if (g < SE_FIX(0.25, 16)) g = sk_mulww(g, gain_mult_Q16);
            else { g = sk_mulww(sk_rround(g, 1), gain_mult_Q16); g = g >= (2147483647 >> 1) ? 2147483647 : shl32(g, 1); }
            ctl->Gains_Q16[k] = g;
            se_bwexpander_32(AR_Q24, order, BWExp_Q16);
            se_limit_warped_coefs(AR_Q24, warping_Q16, SE_FIX(3.999, 24), order);
            for (int i = 0; i < order; i++) ctl->AR_Q13[k * SE_MAX_SHAPE_ORDER + i] = (i16)sk_sat16(sk_rround(AR_Q24[i], 11));
         }
         wv_sync();
         SE_LTOC(22);
      }
   } else
   for (int k = 0; k < c->nb_subfr; k++) {                                             /* no warping (complexity < 4): plain autocorrelation, one sub-frame at a time */
      const int flat_part = c->fs_kHz * 3, slope_part = (swl - flat_part) >> 1;
      LANE0 {
         se_apply_sine_window(xw, x_ptr, 1, slope_part);
         for (int i = 0; i < flat_part; i++) xw[slope_part + i] = x_ptr[slope_part + i];
         se_apply_sine_window(xw + slope_part + flat_part, x_ptr + slope_part + flat_part, 2, slope_part);
      }
      x_ptr += c->subfr_length;
      wv_sync();
      const int scale = se_autocorr_wave(w32, xw, swl, order + 1, xx);
      LANE0 w32[0] = add32(w32[0], imax(sk_mulwb(w32[0] >> 4, SE_FIX(3e-5f, 20)), 1));
      SE_LTOC(18);
      const i32 nrg_w = se_schur64_wave(stk, w32, order);
      se_k2a_Q16_wave(stk + 24, stk, order);
      SE_LTOC(19);
      LANE0 {
         WV_LDS i32 *AR_Q24 = stk + 24;
         i32 nrg = nrg_w;
         int Qnrg = -scale;
         if (Qnrg & 1) { Qnrg -= 1; nrg >>= 1; }
         const i32 tmp32 = se_sqrt_approx(nrg);
         Qnrg >>= 1;
         ctl->Gains_Q16[k] = sk_shl_sat(tmp32, 16 - Qnrg);
         se_bwexpander_32(AR_Q24, order, BWExp_Q16);
         se_lpc_fit(&ctl->AR_Q13[k * SE_MAX_SHAPE_ORDER], AR_Q24, 13, 24, order);
      }
      SE_LTOC(22);
   }
   LANE0 {
      const i32 SNR_adj_dB_Q7 = w32[26];
      const i32 gain_mult_Q16 = se_log2lin(-sk_mlawb(-SE_FIX(16.0, 7), SNR_adj_dB_Q7, SE_FIX(0.16, 16)));
      const i32 gain_add_Q16 = se_log2lin(sk_mlawb(SE_FIX(16.0, 7), SE_FIX(2, 7), SE_FIX(0.16, 16)));
      for (int k = 0; k < c->nb_subfr; k++) { ctl->Gains_Q16[k] = sk_mulww(ctl->Gains_Q16[k], gain_mult_Q16); ctl->Gains_Q16[k] = se_add_pos_sat(ctl->Gains_Q16[k], gain_add_Q16); }
      strength_Q16 = SE_FIX(4.0f, 4) * sk_mlawb(SE_FIX(1.0, 12), SE_FIX(0.5f, 13), c->input_quality_bands_Q15[0] - SE_FIX(1.0, 15));
      strength_Q16 = (strength_Q16 * c->speech_activity_Q8) >> 8;
      i32 Tilt_Q16, HarmShapeGain_Q16;
      if (c->indices.signalType == SE_TYPE_VOICED) {
         const int fs_kHz_inv = SE_FIX(0.2, 14) / c->fs_kHz;
         for (int k = 0; k < c->nb_subfr; k++) {
            const int b_Q14 = fs_kHz_inv + SE_FIX(3.0, 14) / ctl->pitchL[k];
            ctl->LF_shp_Q14[k] = shl32(SE_FIX(1.0, 14) - b_Q14 - sk_mulwb(strength_Q16, b_Q14), 16);
            ctl->LF_shp_Q14[k] |= (uint16_t)(b_Q14 - SE_FIX(1.0, 14));
         }
         Tilt_Q16 = -SE_FIX(0.25f, 16) - sk_mulwb(SE_FIX(1.0, 16) - SE_FIX(0.25f, 16), sk_mulwb(SE_FIX(0.35f, 24), c->speech_activity_Q8));
      } else {
         const int b_Q14 = 21299 / c->fs_kHz;
         ctl->LF_shp_Q14[0] = shl32(SE_FIX(1.0, 14) - b_Q14 - sk_mulwb(strength_Q16, sk_mulwb(SE_FIX(0.6, 16), b_Q14)), 16);
         ctl->LF_shp_Q14[0] |= (uint16_t)(b_Q14 - SE_FIX(1.0, 14));
         for (int k = 1; k < c->nb_subfr; k++) ctl->LF_shp_Q14[k] = ctl->LF_shp_Q14[0];
         Tilt_Q16 = -SE_FIX(0.25f, 16);
      }
      if (c->indices.signalType == SE_TYPE_VOICED) {
         HarmShapeGain_Q16 = sk_mlawb(SE_FIX(0.3f, 16), SE_FIX(1.0, 16) - sk_mulwb(SE_FIX(1.0, 18) - shl32(ctl->coding_quality_Q14, 4), ctl->input_quality_Q14), SE_FIX(0.2f, 16));
         HarmShapeGain_Q16 = sk_mulwb(shl32(HarmShapeGain_Q16, 1), se_sqrt_approx(shl32(c->LTPCorr_Q15, 15)));
      } else HarmShapeGain_Q16 = 0;
      for (int k = 0; k < 4; k++) {
         c->HarmShapeGain_smth_Q16 = sk_mlawb(c->HarmShapeGain_smth_Q16, HarmShapeGain_Q16 - c->HarmShapeGain_smth_Q16, SE_FIX(0.4f, 16));
         c->Tilt_smth_Q16 = sk_mlawb(c->Tilt_smth_Q16, Tilt_Q16 - c->Tilt_smth_Q16, SE_FIX(0.4f, 16));
         ctl->HarmShapeGain_Q14[k] = sk_rround(c->HarmShapeGain_smth_Q16, 2);
         ctl->Tilt_Q14[k] = sk_rround(c->Tilt_smth_Q16, 2);
      }
   }
}

/* ---- silk_burg_modified_c.  QA 25, N_BITS_HEAD_ROOM 3, MIN_RSHIFTS -16, MAX_RSHIFTS 7 ---- */
WV_DEV i64 se_inner_prod16(const WV_LDS i16 *a, const WV_LDS i16 *b, int len) { i64 s = 0; for (int i = 0; i < len; i++) s += (i32)a[i] * (i32)b[i]; return s; }
/* The order recursion reads, of every subframe, only its first and its last 16 samples: the accessors below let it run on the whole signal (head(s, i) = sample i of subframe
 * s, tail(s, m) = its m-th sample from the end, m >= 1) or on those edges alone (pipeline mode 4: the correlations over the whole signal come from the front kernel,
 * se_burg_corr_wave, and a lane of the pred lane kernel runs the recursion of its channel on 32 samples per subframe) */
struct SeBurgXFull { const WV_LDS i16 *x; int L; WV_MEM i32 head(int s, int i) const { return x[s * L + i]; } WV_MEM i32 tail(int s, int m) const { return x[s * L + L - m]; } };
struct SeBurgXEdges { const WV_LDS i16 *e; WV_MEM i32 head(int s, int i) const { return e[s * 32 + i]; } WV_MEM i32 tail(int s, int m) const { return e[s * 32 + 32 - m]; } };
/* stk: 84 words of LDS for the five working rows; C_first_row (stk[0..15]) holds the first row of the correlation matrix on entry */
template <class PA, class XA> WV_DEV void se_burg_rec_l0(i32 *res_nrg, int *res_nrg_Q, PA A_Q16, const XA xa, i32 C0, int rshifts, i32 minInvGain_Q30, int nb_subfr, int D, WV_LDS i32 *stk)
{
   const int QA = 25;
   WV_LDS i32 *C_first_row = stk, *C_last_row = stk + 16, *Af_QA = stk + 32, *CAf = stk + 48, *CAb = stk + 66;
   int k, n, s, lz, reached_max_gain;
   i32 num, nrg, rc_Q31, invGain_Q30, Atmp_QA, Atmp1, tmp1, tmp2, x1, x2;
   for (k = 0; k < 16; k++) C_last_row[k] = C_first_row[k];
   CAb[0] = CAf[0] = C0 + sk_mulhi(SE_FIX(1e-5f, 32), C0) + 1;
   invGain_Q30 = (i32)1 << 30;
   reached_max_gain = 0;
   for (n = 0; n < D; n++) {
      if (rshifts > -2) {
         for (s = 0; s < nb_subfr; s++) {
            x1 = -shl32(xa.head(s, n), 16 - rshifts); x2 = -shl32(xa.tail(s, n + 1), 16 - rshifts);
            tmp1 = shl32(xa.head(s, n), QA - 16); tmp2 = shl32(xa.tail(s, n + 1), QA - 16);
            for (k = 0; k < n; k++) {
               C_first_row[k] = sk_mlawb(C_first_row[k], x1, xa.head(s, n - k - 1));
               C_last_row[k] = sk_mlawb(C_last_row[k], x2, xa.tail(s, n - k));
               Atmp_QA = Af_QA[k];
               tmp1 = sk_mlawb(tmp1, Atmp_QA, xa.head(s, n - k - 1));
               tmp2 = sk_mlawb(tmp2, Atmp_QA, xa.tail(s, n - k));
            }
            tmp1 = shl32(-tmp1, 32 - QA - rshifts); tmp2 = shl32(-tmp2, 32 - QA - rshifts);
            for (k = 0; k <= n; k++) { CAf[k] = sk_mlawb(CAf[k], tmp1, xa.head(s, n - k)); CAb[k] = sk_mlawb(CAb[k], tmp2, xa.tail(s, n - k + 1)); }
         }
      } else {
         for (s = 0; s < nb_subfr; s++) {
            x1 = -shl32(xa.head(s, n), -rshifts); x2 = -shl32(xa.tail(s, n + 1), -rshifts);
            tmp1 = shl32(xa.head(s, n), 17); tmp2 = shl32(xa.tail(s, n + 1), 17);
            for (k = 0; k < n; k++) {
               C_first_row[k] = add32(C_first_row[k], (i32)((u32)x1 * (u32)(i32)xa.head(s, n - k - 1)));
               C_last_row[k] = add32(C_last_row[k], (i32)((u32)x2 * (u32)(i32)xa.tail(s, n - k)));
               Atmp1 = sk_rround(Af_QA[k], QA - 17);
               tmp1 = add32(tmp1, (i32)((u32)(i32)xa.head(s, n - k - 1) * (u32)Atmp1));
               tmp2 = add32(tmp2, (i32)((u32)(i32)xa.tail(s, n - k) * (u32)Atmp1));
            }
            tmp1 = neg32(tmp1); tmp2 = neg32(tmp2);
            for (k = 0; k <= n; k++) {
               CAf[k] = sk_mlaww(CAf[k], tmp1, shl32((i32)xa.head(s, n - k), -rshifts - 1));
               CAb[k] = sk_mlaww(CAb[k], tmp2, shl32((i32)xa.tail(s, n - k + 1), -rshifts - 1));
            }
         }
      }
      tmp1 = C_first_row[n]; tmp2 = C_last_row[n]; num = 0; nrg = add32(CAb[0], CAf[0]);
      for (k = 0; k < n; k++) {
         Atmp_QA = Af_QA[k];
         lz = sk_clz(iabs(Atmp_QA)) - 1;
         lz = imin(32 - QA, lz);
         Atmp1 = shl32(Atmp_QA, lz);
         tmp1 = se_add_lshift32(tmp1, sk_mulhi(C_last_row[n - k - 1], Atmp1), 32 - QA - lz);
         tmp2 = se_add_lshift32(tmp2, sk_mulhi(C_first_row[n - k - 1], Atmp1), 32 - QA - lz);
         num = se_add_lshift32(num, sk_mulhi(CAb[n - k], Atmp1), 32 - QA - lz);
         nrg = se_add_lshift32(nrg, sk_mulhi(add32(CAb[k + 1], CAf[k + 1]), Atmp1), 32 - QA - lz);
      }
      CAf[n + 1] = tmp1; CAb[n + 1] = tmp2;
      num = add32(num, tmp2);
      num = shl32(neg32(num), 1);
      if (iabs(num) < nrg) rc_Q31 = sk_div32_varQ(num, nrg, 31); else rc_Q31 = num > 0 ? 2147483647 : (i32)(-2147483647 - 1);
      tmp1 = ((i32)1 << 30) - sk_mulhi(rc_Q31, rc_Q31);
      tmp1 = shl32(sk_mulhi(invGain_Q30, tmp1), 2);
      if (tmp1 <= minInvGain_Q30) {
         tmp2 = ((i32)1 << 30) - sk_div32_varQ(minInvGain_Q30, invGain_Q30, 30);
         rc_Q31 = se_sqrt_approx(tmp2);
         if (rc_Q31 > 0) { rc_Q31 = (rc_Q31 + tmp2 / rc_Q31) >> 1; rc_Q31 = shl32(rc_Q31, 16); if (num < 0) rc_Q31 = -rc_Q31; }
         invGain_Q30 = minInvGain_Q30;
         reached_max_gain = 1;
      } else invGain_Q30 = tmp1;
      for (k = 0; k < (n + 1) >> 1; k++) {
         tmp1 = Af_QA[k]; tmp2 = Af_QA[n - k - 1];
         Af_QA[k] = se_add_lshift32(tmp1, sk_mulhi(tmp2, rc_Q31), 1);
         Af_QA[n - k - 1] = se_add_lshift32(tmp2, sk_mulhi(tmp1, rc_Q31), 1);
      }
      Af_QA[n] = rc_Q31 >> (31 - QA);
      if (reached_max_gain) { for (k = n + 1; k < D; k++) Af_QA[k] = 0; break; }
      for (k = 0; k <= n + 1; k++) {
         tmp1 = CAf[k]; tmp2 = CAb[n - k + 1];
         CAf[k] = se_add_lshift32(tmp1, sk_mulhi(tmp2, rc_Q31), 1);
         CAb[n - k + 1] = se_add_lshift32(tmp2, sk_mulhi(tmp1, rc_Q31), 1);
      }
   }
   if (reached_max_gain) {
      for (k = 0; k < D; k++) A_Q16[k] = -sk_rround(Af_QA[k], QA - 16);
      for (s = 0; s < nb_subfr; s++) {
         i64 ip = 0; for (k = 0; k < D; k++) ip += xa.head(s, k) * xa.head(s, k);
         if (rshifts > 0) C0 -= (i32)(ip >> rshifts); else C0 = sub32(C0, shl32((i32)ip, -rshifts));
      }
      *res_nrg = shl32(sk_mulhi(invGain_Q30, C0), 2);
      *res_nrg_Q = -rshifts;
   } else {
      nrg = CAf[0]; tmp1 = (i32)1 << 16;
      for (k = 0; k < D; k++) {
         Atmp1 = sk_rround(Af_QA[k], QA - 16);
         nrg = sk_mlaww(nrg, CAf[k + 1], Atmp1);
         tmp1 = sk_mlaww(tmp1, Atmp1, Atmp1);
         A_Q16[k] = -Atmp1;
      }
      *res_nrg = sk_mlaww(nrg, sk_mulhi(SE_FIX(1e-5f, 32), C0), -tmp1);
      *res_nrg_Q = -rshifts;
   }
}

/* silk_burg_modified_c, serial: the correlations over the whole signal, then the recursion */
template <class PA> WV_DEV void se_burg_modified_l0(i32 *res_nrg, int *res_nrg_Q, PA A_Q16, const WV_LDS i16 *x, i32 minInvGain_Q30, int subfr_length, int nb_subfr, int D, WV_LDS i32 *stk)
{
   const int QA = 25;
   WV_LDS i32 *C_first_row = stk;
   const i64 C0_64 = se_inner_prod16(x, x, subfr_length * nb_subfr);
   int rshifts = 32 + 1 + 3 - se_clz64(C0_64);
   if (rshifts > 32 - QA) rshifts = 32 - QA;
   if (rshifts < -16) rshifts = -16;
   const i32 C0 = rshifts > 0 ? (i32)(C0_64 >> rshifts) : shl32((i32)C0_64, -rshifts);
   for (int k = 0; k < 16; k++) C_first_row[k] = 0;
   for (int s = 0; s < nb_subfr; s++) {
      const WV_LDS i16 *x_ptr = x + s * subfr_length;
      for (int n = 1; n < D + 1; n++) {
         const i64 ip = se_inner_prod16(x_ptr, x_ptr + n, subfr_length - n);
         if (rshifts > 0) C_first_row[n - 1] += (i32)(ip >> rshifts); else C_first_row[n - 1] = add32(C_first_row[n - 1], shl32((i32)ip, -rshifts));
      }
   }
   const SeBurgXFull xa = {x, subfr_length};
   se_burg_rec_l0(res_nrg, res_nrg_Q, A_Q16, xa, C0, rshifts, minInvGain_Q30, nb_subfr, D, stk);
}

/* silk_burg_modified_c on the whole wave.  Every inner loop of the reference is a sum of individually rounded terms added with wrap-around, so the terms are
 * computed one per lane -- lane = (subframe s = lane / 16, tap k = lane % 16) -- and summed in any order; only the order recursion n = 0..D-1 is serial.
 * out[0] = res_nrg, out[1] = res_nrg_Q; stk: 84 + 4 * 64 + 8 words of LDS. */
WV_DEVN void se_burg_modified_wave(WV_LDS i32 *out, WV_LDS i32 *A_Q16, const WV_LDS i16 *x, i32 minInvGain_Q30, int subfr_length, int nb_subfr, int D, WV_LDS i32 *stk)
{
   const int QA = 25;
   const int lane = wv_lane(), s = lane >> 4, k = lane & 15;
   const bool sact = s < nb_subfr;
   WV_LDS i32 *C_first_row = stk, *C_last_row = stk + 16, *Af_QA = stk + 32, *CAf = stk + 48, *CAb = stk + 66, *T0 = stk + 84, *T1 = T0 + 64, *T2 = T1 + 64, *T3 = T2 + 64, *TS = T3 + 64;
   const WV_LDS i16 *xs = x + (sact ? s : 0) * subfr_length;
   i64 part = 0;
   FOR_LANES(i, subfr_length * nb_subfr) part += (i32)x[i] * (i32)x[i];
   const i64 C0_64 = wv_sum64(part);
   int rshifts = 32 + 1 + 3 - se_clz64(C0_64);
   if (rshifts > 32 - QA) rshifts = 32 - QA;
   if (rshifts < -16) rshifts = -16;
   i32 C0 = rshifts > 0 ? (i32)(C0_64 >> rshifts) : shl32((i32)C0_64, -rshifts);
   {  /* first row of the correlation matrix: lane (s, n - 1) */
      i32 term = 0;
      if (sact && k < D) { const i64 ip = se_inner_prod16(xs, xs + k + 1, subfr_length - k - 1); term = rshifts > 0 ? (i32)(ip >> rshifts) : shl32((i32)ip, -rshifts); }
      T0[lane] = term;
      wv_sync();
      if (lane < 16) { i32 v = 0; for (int q = 0; q < nb_subfr; q++) v = add32(v, T0[q * 16 + lane]); C_first_row[lane] = lane < D ? v : 0; C_last_row[lane] = lane < D ? v : 0; Af_QA[lane] = 0; }
      if (lane == 0) { CAb[0] = CAf[0] = C0 + sk_mulhi(SE_FIX(1e-5f, 32), C0) + 1; }
      wv_sync();
   }
   i32 invGain_Q30 = (i32)1 << 30;
   int reached_max_gain = 0, n;
   for (n = 0; n < D; n++) {
      /* ---- update the rows and C * Af, C * flipud(Af): terms per (s, k) ---- */
      i32 t0 = 0, t1 = 0, t2 = 0, t3 = 0;
      const i32 xn = sact ? (i32)xs[n] : 0, xe = sact ? (i32)xs[subfr_length - n - 1] : 0;
      if (sact && k < n) {
         const i32 xa = xs[n - k - 1], xb = xs[subfr_length - n + k], Atmp = Af_QA[k];
         if (rshifts > -2) { t0 = sk_mulwb(-shl32(xn, 16 - rshifts), xa); t1 = sk_mulwb(-shl32(xe, 16 - rshifts), xb); t2 = sk_mulwb(Atmp, xa); t3 = sk_mulwb(Atmp, xb); }
         else { const i32 A1 = sk_rround(Atmp, QA - 17); t0 = (i32)((u32)(-shl32(xn, -rshifts)) * (u32)xa); t1 = (i32)((u32)(-shl32(xe, -rshifts)) * (u32)xb); t2 = (i32)((u32)xa * (u32)A1); t3 = (i32)((u32)xb * (u32)A1); }
      }
      T0[lane] = t0; T1[lane] = t1; T2[lane] = t2; T3[lane] = t3;
      wv_sync();
      if (lane < 16 && lane < n) { i32 a0 = C_first_row[lane], a1 = C_last_row[lane]; for (int q = 0; q < nb_subfr; q++) { a0 = add32(a0, T0[q * 16 + lane]); a1 = add32(a1, T1[q * 16 + lane]); } C_first_row[lane] = a0; C_last_row[lane] = a1; }
      if (sact && k == 0) {                                                       /* tmp1 / tmp2 of subframe s */
         i32 tmp1, tmp2;
         if (rshifts > -2) { tmp1 = shl32(xn, QA - 16); tmp2 = shl32(xe, QA - 16); } else { tmp1 = shl32(xn, 17); tmp2 = shl32(xe, 17); }
         for (int q = 0; q < n; q++) { tmp1 = add32(tmp1, T2[s * 16 + q]); tmp2 = add32(tmp2, T3[s * 16 + q]); }
         if (rshifts > -2) { tmp1 = shl32(neg32(tmp1), 32 - QA - rshifts); tmp2 = shl32(neg32(tmp2), 32 - QA - rshifts); } else { tmp1 = neg32(tmp1); tmp2 = neg32(tmp2); }
         TS[2 * s] = tmp1; TS[2 * s + 1] = tmp2;
      }
      wv_sync();
      t0 = 0; t1 = 0;
      if (sact && k <= n) {
         const i32 tmp1 = TS[2 * s], tmp2 = TS[2 * s + 1], xa = xs[n - k], xb = xs[subfr_length - n + k - 1];
         if (rshifts > -2) { t0 = sk_mulwb(tmp1, xa); t1 = sk_mulwb(tmp2, xb); }
         else { t0 = sk_mulww(tmp1, shl32(xa, -rshifts - 1)); t1 = sk_mulww(tmp2, shl32(xb, -rshifts - 1)); }
      }
      T0[lane] = t0; T1[lane] = t1;
      wv_sync();
      if (lane < 16 && lane <= n) { i32 a0 = CAf[lane], a1 = CAb[lane]; for (int q = 0; q < nb_subfr; q++) { a0 = add32(a0, T0[q * 16 + lane]); a1 = add32(a1, T1[q * 16 + lane]); } CAf[lane] = a0; CAb[lane] = a1; }
      wv_sync();
      /* ---- nominator / denominator of the reflection coefficient: lanes over k < n ---- */
      i32 p1 = 0, p2 = 0, pn = 0, pg = 0;
      if (lane < n) {
         const i32 Atmp_QA = Af_QA[lane];
         int lz = sk_clz(iabs(Atmp_QA)) - 1; lz = imin(32 - QA, lz);
         const i32 Atmp1 = shl32(Atmp_QA, lz); const int sh = 32 - QA - lz;
         p1 = shl32(sk_mulhi(C_last_row[n - lane - 1], Atmp1), sh); p2 = shl32(sk_mulhi(C_first_row[n - lane - 1], Atmp1), sh);
         pn = shl32(sk_mulhi(CAb[n - lane], Atmp1), sh); pg = shl32(sk_mulhi(add32(CAb[lane + 1], CAf[lane + 1]), Atmp1), sh);
      }
      i32 tmp1 = add32(C_first_row[n], wv_sum(p1)), tmp2 = add32(C_last_row[n], wv_sum(p2)), num = wv_sum(pn), nrg = add32(add32(CAb[0], CAf[0]), wv_sum(pg));
      wv_sync();
      if (lane == 0) { CAf[n + 1] = tmp1; CAb[n + 1] = tmp2; }
      num = add32(num, tmp2);
      num = shl32(neg32(num), 1);
      i32 rc_Q31;
      if (iabs(num) < nrg) rc_Q31 = sk_div32_varQ(num, nrg, 31); else rc_Q31 = num > 0 ? 2147483647 : (i32)(-2147483647 - 1);
      tmp1 = ((i32)1 << 30) - sk_mulhi(rc_Q31, rc_Q31);
      tmp1 = shl32(sk_mulhi(invGain_Q30, tmp1), 2);
      if (tmp1 <= minInvGain_Q30) {
         tmp2 = ((i32)1 << 30) - sk_div32_varQ(minInvGain_Q30, invGain_Q30, 30);
         rc_Q31 = se_sqrt_approx(tmp2);
         if (rc_Q31 > 0) { rc_Q31 = (rc_Q31 + tmp2 / rc_Q31) >> 1; rc_Q31 = shl32(rc_Q31, 16); if (num < 0) rc_Q31 = -rc_Q31; }
         invGain_Q30 = minInvGain_Q30;
         reached_max_gain = 1;
      } else invGain_Q30 = tmp1;
      wv_sync();
      if (lane < ((n + 1) >> 1)) { const i32 a = Af_QA[lane], b = Af_QA[n - lane - 1]; Af_QA[lane] = se_add_lshift32(a, sk_mulhi(b, rc_Q31), 1); Af_QA[n - lane - 1] = se_add_lshift32(b, sk_mulhi(a, rc_Q31), 1); }
      wv_sync();
      if (lane == 0) Af_QA[n] = rc_Q31 >> (31 - QA);
      if (reached_max_gain) { if (lane > n && lane < D) Af_QA[lane] = 0; wv_sync(); break; }
      if (lane <= n + 1) { const i32 a = CAf[lane], b = CAb[n - lane + 1]; T0[lane] = se_add_lshift32(a, sk_mulhi(b, rc_Q31), 1); T1[lane] = se_add_lshift32(b, sk_mulhi(a, rc_Q31), 1); }
      wv_sync();
      if (lane <= n + 1) { CAf[lane] = T0[lane]; CAb[n - lane + 1] = T1[lane]; }
      wv_sync();
   }
   LANE0 {
      if (reached_max_gain) {
         for (int q = 0; q < D; q++) A_Q16[q] = -sk_rround(Af_QA[q], QA - 16);
         for (int q = 0; q < nb_subfr; q++) {
            const WV_LDS i16 *xp = x + q * subfr_length;
            const i64 ip = se_inner_prod16(xp, xp, D);
            if (rshifts > 0) C0 -= (i32)(ip >> rshifts); else C0 = sub32(C0, shl32((i32)ip, -rshifts));
         }
         out[0] = shl32(sk_mulhi(invGain_Q30, C0), 2);
      } else {
         i32 nrg = CAf[0], tmp1 = (i32)1 << 16;
         for (int q = 0; q < D; q++) { const i32 Atmp1 = sk_rround(Af_QA[q], QA - 16); nrg = sk_mlaww(nrg, CAf[q + 1], Atmp1); tmp1 = sk_mlaww(tmp1, Atmp1, Atmp1); A_Q16[q] = -Atmp1; }
         out[0] = sk_mlaww(nrg, sk_mulhi(SE_FIX(1e-5f, 32), C0), -tmp1);
      }
      out[1] = -rshifts;
   }
}

/* The part of silk_burg_modified_c that reads the whole signal -- its energy and the first row of the correlation matrix (burg_modified_FIX.c:73-98) -- on the wave, for a
 * recursion that runs elsewhere (se_burg_rec_l0 on the signal's edges: pipeline mode 4).  T: 64 words of LDS. */
struct SeBurgCorr { i32 C0, rshifts, first_row[16]; };
WV_DEV void se_burg_corr_wave(SeBurgCorr *out /* global */, const WV_LDS i16 *x, int subfr_length, int nb_subfr, int D, WV_LDS i32 *T)
{
   const int QA = 25;
   const int lane = wv_lane(), s = lane >> 4, k = lane & 15;
   const bool sact = s < nb_subfr;
   const WV_LDS i16 *xs = x + (sact ? s : 0) * subfr_length;
   i64 part = 0;
   FOR_LANES(i, subfr_length * nb_subfr) part += (i32)x[i] * (i32)x[i];
   const i64 C0_64 = wv_sum64(part);
   int rshifts = 32 + 1 + 3 - se_clz64(C0_64);
   if (rshifts > 32 - QA) rshifts = 32 - QA;
   if (rshifts < -16) rshifts = -16;
   const i32 C0 = rshifts > 0 ? (i32)(C0_64 >> rshifts) : shl32((i32)C0_64, -rshifts);
   i32 term = 0;
   if (sact && k < D) { const i64 ip = se_inner_prod16(xs, xs + k + 1, subfr_length - k - 1); term = rshifts > 0 ? (i32)(ip >> rshifts) : shl32((i32)ip, -rshifts); }
   wv_sync();
   T[lane] = term;
   wv_sync();
   if (lane < 16) { i32 v = 0; for (int q = 0; q < nb_subfr; q++) v = add32(v, T[q * 16 + lane]); out->first_row[lane] = lane < D ? v : 0; }
   if (lane == 0) { out->C0 = C0; out->rshifts = rshifts; }
   wv_sync();
}

/* ---- silk_quant_LTP_gains + silk_VQ_WMat_EC_c ---- */
WV_DEV void se_quant_ltp_gains_l0(WV_LDS i16 *B_Q14, WV_LDS i8 *cbk_index, WV_LDS i8 *periodicity_index, WV_LDS i32 *sum_log_gain_Q7, WV_LDS i32 *pred_gain_dB_Q7,
      const WV_LDS i32 *XX_Q17, const WV_LDS i32 *xX_Q17, int subfr_len, int nb_subfr)
{
   const int cb_off[3] = {0, 8, 24};
   i8 temp_idx[4];
   i32 min_rate_dist_Q7 = 2147483647, best_sum_log_gain_Q7 = 0, res_nrg_Q15 = 0;
   int gain_Q7 = 0;
   for (int k = 0; k < 3; k++) {
      const i32 gain_safety = SE_FIX(0.4, 7);
      const u8 *cl_Q5 = &se_ltp_gain_bits_q5[cb_off[k]], *cb_gain_Q7 = &se_ltp_vq_gain_q7[cb_off[k]];
      const i8 *cb_Q7 = &sk_ltp_vq_q7[cb_off[k] * 5];
      const int cbk_size = 8 << k;
      const WV_LDS i32 *XX = XX_Q17, *xX = xX_Q17;
      i32 rate_dist_Q7 = 0, sum_log_gain_tmp_Q7 = *sum_log_gain_Q7;
      res_nrg_Q15 = 0;
      for (int j = 0; j < nb_subfr; j++) {
         const i32 max_gain_Q7 = se_log2lin((SE_FIX(250.0f / 6.0, 7) - sum_log_gain_tmp_Q7) + SE_FIX(7, 7)) - gain_safety;
         /* VQ_WMat_EC */
         i32 neg_xX_Q24[5], rate_dist_sub = 2147483647, res_nrg_sub = 2147483647;
         for (int i = 0; i < 5; i++) neg_xX_Q24[i] = -shl32(xX[i], 7);
         temp_idx[j] = 0;
         const i8 *row = cb_Q7;
         for (int v = 0; v < cbk_size; v++) {
            const int gain_tmp_Q7 = cb_gain_Q7[v];
            i32 sum1_Q15 = SE_FIX(1.001, 15), sum2_Q24;
            const i32 penalty = shl32(imax(sub32(gain_tmp_Q7, max_gain_Q7), 0), 11);
            sum2_Q24 = neg_xX_Q24[0] + XX[1] * row[1]; sum2_Q24 += XX[2] * row[2]; sum2_Q24 += XX[3] * row[3]; sum2_Q24 += XX[4] * row[4];
            sum2_Q24 = shl32(sum2_Q24, 1); sum2_Q24 += XX[0] * row[0]; sum1_Q15 = sk_mlawb(sum1_Q15, sum2_Q24, row[0]);
            sum2_Q24 = neg_xX_Q24[1] + XX[7] * row[2]; sum2_Q24 += XX[8] * row[3]; sum2_Q24 += XX[9] * row[4];
            sum2_Q24 = shl32(sum2_Q24, 1); sum2_Q24 += XX[6] * row[1]; sum1_Q15 = sk_mlawb(sum1_Q15, sum2_Q24, row[1]);
            sum2_Q24 = neg_xX_Q24[2] + XX[13] * row[3]; sum2_Q24 += XX[14] * row[4];
            sum2_Q24 = shl32(sum2_Q24, 1); sum2_Q24 += XX[12] * row[2]; sum1_Q15 = sk_mlawb(sum1_Q15, sum2_Q24, row[2]);
            sum2_Q24 = neg_xX_Q24[3] + XX[19] * row[4];
            sum2_Q24 = shl32(sum2_Q24, 1); sum2_Q24 += XX[18] * row[3]; sum1_Q15 = sk_mlawb(sum1_Q15, sum2_Q24, row[3]);
            sum2_Q24 = shl32(neg_xX_Q24[4], 1); sum2_Q24 += XX[24] * row[4]; sum1_Q15 = sk_mlawb(sum1_Q15, sum2_Q24, row[4]);
            if (sum1_Q15 >= 0) {
               const i32 bits_res_Q8 = sk_mulbb(subfr_len, se_lin2log(sum1_Q15 + penalty) - (15 << 7));
               const i32 bits_tot_Q8 = se_add_lshift32(bits_res_Q8, cl_Q5[v], 3 - 1);
               if (bits_tot_Q8 <= rate_dist_sub) { rate_dist_sub = bits_tot_Q8; res_nrg_sub = sum1_Q15 + penalty; temp_idx[j] = (i8)v; gain_Q7 = gain_tmp_Q7; }
            }
            row += 5;
         }
         res_nrg_Q15 = se_add_pos_sat(res_nrg_Q15, res_nrg_sub);
         rate_dist_Q7 = se_add_pos_sat(rate_dist_Q7, rate_dist_sub);
         sum_log_gain_tmp_Q7 = imax(0, sum_log_gain_tmp_Q7 + se_lin2log(gain_safety + gain_Q7) - SE_FIX(7, 7));
         XX += 25; xX += 5;
      }
      if (rate_dist_Q7 <= min_rate_dist_Q7) {
         min_rate_dist_Q7 = rate_dist_Q7; *periodicity_index = (i8)k;
         for (int j = 0; j < nb_subfr; j++) cbk_index[j] = temp_idx[j];
         best_sum_log_gain_Q7 = sum_log_gain_tmp_Q7;
      }
   }
   const i8 *cb = &sk_ltp_vq_q7[cb_off[*periodicity_index] * 5];
   for (int j = 0; j < nb_subfr; j++) for (int k = 0; k < 5; k++) B_Q14[j * 5 + k] = (i16)shl32(cb[cbk_index[j] * 5 + k], 7);
   res_nrg_Q15 = nb_subfr == 2 ? res_nrg_Q15 >> 1 : res_nrg_Q15 >> 2;
   *sum_log_gain_Q7 = best_sum_log_gain_Q7;
   *pred_gain_dB_Q7 = sk_mulbb(-3, se_lin2log(res_nrg_Q15) - (15 << 7));
}

WV_DEV void se_ltp_scale_ctrl(WV_LDS OaSilkEncChannel *c, WV_LDS SeEncCtrl *ctl, int condCoding)
{
   if (condCoding == SE_CODE_INDEPENDENTLY) {
      int round_loss = c->PacketLoss_perc * c->nFramesPerPacket;
      if (c->LBRR_flag) round_loss = 2 + sk_mulbb(round_loss, round_loss) / 100;
      c->indices.LTP_scaleIndex = (i8)(sk_mulbb(ctl->LTPredCodGain_Q7, round_loss) > se_log2lin(128 * 7 + 2900 - c->SNR_dB_Q7));
      c->indices.LTP_scaleIndex += (i8)(sk_mulbb(ctl->LTPredCodGain_Q7, round_loss) > se_log2lin(128 * 7 + 3900 - c->SNR_dB_Q7));
   } else c->indices.LTP_scaleIndex = 0;
   ctl->LTP_scale_Q14 = sk_ltpscales_table_q14[c->indices.LTP_scaleIndex];
}

/* ---- the same two stages on the wave ---- */
/* silk_find_LTP_FIX: per sub-frame the 5x5 correlation matrix of the lagged residual and its 5 correlations with the target.  Every entry is a sum over the
 * sub-frame of (optionally shifted) products -- order-free -- plus at most four boundary corrections chained along a diagonal: lanes over the samples, one
 * reduction per first-row entry and per correlation, the corrections on wave-uniform values; the 30 normalising 64-bit divisions one per lane. */
WV_DEV void se_find_ltp_wave(WV_LDS i32 *XX, WV_LDS i32 *xX, const WV_LDS i16 *r_ptr, const WV_LDS i32 *lag, int subfr_length, int nb_subfr)
{
   const int order = 5, L = subfr_length, lane = wv_lane();
   for (int k = 0; k < nb_subfr; k++) {
      const WV_LDS i16 *xm = r_ptr - (lag[k] + 5 / 2), *ptr1 = xm + order - 1;
      i32 xx, nrg; int xx_shifts, rs;
      se_sum_sqr_shift_wave(&xx, &xx_shifts, r_ptr, L + order);
      se_sum_sqr_shift_wave(&nrg, &rs, xm, L + order - 1);
      /* first column (= first row) of the matrix: lag lg against lag 0, over the sub-frame */
      i32 col[5];
      {
         i32 e = nrg;
         for (int i = 0; i < order - 1; i++) e -= sk_mulbb(xm[i], xm[i]) >> rs;
         col[0] = e;
         i32 part[4] = {0, 0, 0, 0};
         FOR_LANES(i, L) {
            const i32 a = ptr1[i];
#pragma unroll
            for (int lg = 1; lg < order; lg++) part[lg - 1] = add32(part[lg - 1], sk_mulbb(a, ptr1[i - lg]) >> rs);
         }
#pragma unroll
         for (int lg = 1; lg < order; lg++) col[lg] = wv_sum(part[lg - 1]);
      }
      /* down each diagonal: drop the product leaving the window, add the one entering */
      wv_sync();
#pragma unroll
      for (int lg = 0; lg < order; lg++) {
         i32 e = col[lg];
         if (lane == 0) { XX[lg * order] = e; XX[lg] = e; }
         for (int j = 1; j < order - lg; j++) {
            e = sub32(e, sk_mulbb(ptr1[L - j], ptr1[L - j - lg]) >> rs);
            e = add32(e, sk_mulbb(ptr1[-j], ptr1[-j - lg]) >> rs);
            if (lane == 0) { XX[(lg + j) * order + j] = e; XX[j * order + lg + j] = e; }
         }
      }
      const int extra_shifts = xx_shifts - rs;
      int xX_shifts = xx_shifts, XX_down = 0;
      if (extra_shifts > 0) { XX_down = extra_shifts; nrg >>= extra_shifts; }
      else if (extra_shifts < 0) { xX_shifts = rs; xx >>= -extra_shifts; }
      i32 ipart[5] = {0, 0, 0, 0, 0};
      FOR_LANES(i, L) {
         const i32 t = r_ptr[i];
#pragma unroll
         for (int lg = 0; lg < order; lg++) ipart[lg] = add32(ipart[lg], sk_mulbb(ptr1[i - lg], t) >> xX_shifts);
      }
      i32 ip[5];
#pragma unroll
      for (int lg = 0; lg < order; lg++) ip[lg] = wv_sum(ipart[lg]);
      wv_sync();
      const i32 temp = imax(sk_mlawb(1, nrg, SE_FIX(0.03f, 16)), xx);
      if (lane < 25) XX[lane] = (i32)(((i64)(XX[lane] >> XX_down) << 17) / temp);
      if (lane >= 32 && lane < 37) { i32 v = ip[0]; for (int lg = 1; lg < order; lg++) if (lane - 32 == lg) v = ip[lg]; xX[lane - 32] = (i32)(((i64)v << 17) / temp); }
      wv_sync();
      r_ptr += subfr_length; XX += 25; xX += 5;
   }
}

/* silk_quant_LTP_gains + silk_VQ_WMat_EC: one lane per codebook vector of all three codebooks at once (8 + 16 + 32 = 56 lanes).  Along the sub-frames each
 * codebook carries its own gain budget; per sub-frame and codebook the winner is the LAST vector of minimal cost (the reference's `<=` scan): a wave minimum, then
 * the top bit of a ballot.  A sub-frame in which some codebook has no admissible vector makes the reference fall back on a value left over from the previous
 * search; that corner is left to the serial form. */
WV_DEV void se_quant_ltp_gains_wave(WV_LDS i16 *B_Q14, WV_LDS i8 *cbk_index, WV_LDS i8 *periodicity_index, WV_LDS i32 *sum_log_gain_Q7, WV_LDS i32 *pred_gain_dB_Q7,
      const WV_LDS i32 *XX_Q17, const WV_LDS i32 *xX_Q17, int subfr_len, int nb_subfr)
{
   const int lane = wv_lane(), v = imin(lane, 55), k = v < 8 ? 0 : v < 24 ? 1 : 2;
   const u64 cb_mask[3] = {0xffull, 0xffff00ull, 0xffffffff000000ull};
   const int cb_first[3] = {0, 8, 24};
   const i32 gain_safety = SE_FIX(0.4, 7);
   const i8 *row = &sk_ltp_vq_q7[v * 5];
   const i32 r0 = row[0], r1 = row[1], r2 = row[2], r3 = row[3], r4 = row[4], gain_v = se_ltp_vq_gain_q7[v], cl_v = se_ltp_gain_bits_q5[v];
   i32 slg[3], rate[3] = {0, 0, 0}, res[3] = {0, 0, 0};
   int idx[3][4], degenerate = 0;
   slg[0] = slg[1] = slg[2] = *sum_log_gain_Q7;
   const WV_LDS i32 *XX = XX_Q17, *xX = xX_Q17;
#pragma unroll
   for (int j = 0; j < 4; j++) {
      if (j >= nb_subfr) break;
      const i32 my_slg = k == 0 ? slg[0] : k == 1 ? slg[1] : slg[2];
      const i32 max_gain_Q7 = se_log2lin((SE_FIX(250.0f / 6.0, 7) - my_slg) + SE_FIX(7, 7)) - gain_safety;
      const i32 n0 = -shl32(xX[0], 7), n1 = -shl32(xX[1], 7), n2 = -shl32(xX[2], 7), n3 = -shl32(xX[3], 7), n4 = -shl32(xX[4], 7);
      const i32 penalty = shl32(imax(sub32(gain_v, max_gain_Q7), 0), 11);
      i32 sum1_Q15 = SE_FIX(1.001, 15), sum2_Q24;
      sum2_Q24 = n0 + XX[1] * r1; sum2_Q24 += XX[2] * r2; sum2_Q24 += XX[3] * r3; sum2_Q24 += XX[4] * r4;
      sum2_Q24 = shl32(sum2_Q24, 1); sum2_Q24 += XX[0] * r0; sum1_Q15 = sk_mlawb(sum1_Q15, sum2_Q24, r0);
      sum2_Q24 = n1 + XX[7] * r2; sum2_Q24 += XX[8] * r3; sum2_Q24 += XX[9] * r4;
      sum2_Q24 = shl32(sum2_Q24, 1); sum2_Q24 += XX[6] * r1; sum1_Q15 = sk_mlawb(sum1_Q15, sum2_Q24, r1);
      sum2_Q24 = n2 + XX[13] * r3; sum2_Q24 += XX[14] * r4;
      sum2_Q24 = shl32(sum2_Q24, 1); sum2_Q24 += XX[12] * r2; sum1_Q15 = sk_mlawb(sum1_Q15, sum2_Q24, r2);
      sum2_Q24 = n3 + XX[19] * r4;
      sum2_Q24 = shl32(sum2_Q24, 1); sum2_Q24 += XX[18] * r3; sum1_Q15 = sk_mlawb(sum1_Q15, sum2_Q24, r3);
      sum2_Q24 = shl32(n4, 1); sum2_Q24 += XX[24] * r4; sum1_Q15 = sk_mlawb(sum1_Q15, sum2_Q24, r4);
      const bool ok = lane < 56 && sum1_Q15 >= 0;
      const i32 res_v = sum1_Q15 + penalty;
      const i32 bits_tot_Q8 = ok ? se_add_lshift32(sk_mulbb(subfr_len, se_lin2log(ok ? res_v : 1) - (15 << 7)), cl_v, 3 - 1) : 2147483647;
#pragma unroll
      for (int c = 0; c < 3; c++) {
         const bool mine = ok && k == c;
         const i32 best = wv_min(mine ? bits_tot_Q8 : 2147483647);
         const u64 at = wv_ballot(mine && bits_tot_Q8 == best) & cb_mask[c];
         if (!at) { degenerate = 1; idx[c][j] = 0; continue; }
         const int win = 63 - __builtin_clzll(at);
         idx[c][j] = win - cb_first[c];
         res[c] = se_add_pos_sat(res[c], wv_bcast(res_v, win));
         rate[c] = se_add_pos_sat(rate[c], best);
         slg[c] = imax(0, slg[c] + se_lin2log(gain_safety + wv_bcast(gain_v, win)) - SE_FIX(7, 7));
      }
      XX += 25; xX += 5;
   }
   if (degenerate) {
      LANE0 se_quant_ltp_gains_l0(B_Q14, cbk_index, periodicity_index, sum_log_gain_Q7, pred_gain_dB_Q7, XX_Q17, xX_Q17, subfr_len, nb_subfr);
      wv_sync();
      return;
   }
   int per = 0; i32 min_rate = 2147483647;
#pragma unroll
   for (int c = 0; c < 3; c++) if (rate[c] <= min_rate) { min_rate = rate[c]; per = c; }
   const i32 best_slg = per == 0 ? slg[0] : per == 1 ? slg[1] : slg[2];
   wv_sync();
   if (lane < nb_subfr) { int w = idx[0][0]; for (int c = 0; c < 3; c++) for (int j = 0; j < 4; j++) if (c == per && j == lane) w = idx[c][j]; cbk_index[lane] = (i8)w; }
   wv_sync();
   FOR_LANES(t, nb_subfr * 5) { const int j = t / 5, e = t - 5 * j; B_Q14[t] = (i16)shl32(sk_ltp_vq_q7[(cb_first[per] + cbk_index[j]) * 5 + e], 7); }
   LANE0 {
      const i32 res_nrg_Q15 = nb_subfr == 2 ? res[2] >> 1 : res[2] >> 2;
      *periodicity_index = (i8)per;
      *sum_log_gain_Q7 = best_slg;
      *pred_gain_dB_Q7 = sk_mulbb(-3, se_lin2log(res_nrg_Q15) - (15 << 7));
   }
   wv_sync();
}

/* every output sample is independent: lanes */
WV_DEV void se_ltp_analysis_filter_wave(WV_LDS i16 *LTP_res, const WV_LDS i16 *x, const WV_LDS i16 *LTPCoef_Q14, const WV_LDS i32 *pitchL, const WV_LDS i32 *invGains_Q16, int subfr_length, int nb_subfr, int pre_length)
{
   for (int k = 0; k < nb_subfr; k++) {
      const WV_LDS i16 *x_ptr = x + k * subfr_length, *x_lag = x_ptr - pitchL[k];
      WV_LDS i16 *out = LTP_res + k * (subfr_length + pre_length);
      const i32 b0 = LTPCoef_Q14[k * 5], b1 = LTPCoef_Q14[k * 5 + 1], b2 = LTPCoef_Q14[k * 5 + 2], b3 = LTPCoef_Q14[k * 5 + 3], b4 = LTPCoef_Q14[k * 5 + 4], ig = invGains_Q16[k];
      FOR_LANES(i, subfr_length + pre_length) {
         i32 e = sk_mulbb(x_lag[i + 2], b0);
         e = sk_mlabb(e, x_lag[i + 1], b1); e = sk_mlabb(e, x_lag[i], b2); e = sk_mlabb(e, x_lag[i - 1], b3); e = sk_mlabb(e, x_lag[i - 2], b4);
         e = sk_rround(e, 14);
         const i32 r = sk_sat16((i32)x_ptr[i] - e);
         out[i] = (i16)sk_mulwb(ig, r);
      }
   }
}

/* LPC_res: i16[2 * (16 + 80)] */
WV_DEV void se_residual_energy_wave(WV_LDS i32 *nrgs, WV_LDS i32 *nrgsQ, const WV_LDS i16 *x, const WV_LDS i16 *a_Q12 /* [2][16] */, const WV_LDS i32 *gains, int subfr_length, int nb_subfr, int LPC_order, WV_LDS i16 *LPC_res)
{
   const int offset = LPC_order + subfr_length;
   const WV_LDS i16 *x_ptr = x;
   for (int i = 0; i < nb_subfr >> 1; i++) {
      se_lpc_analysis_filter_wave(LPC_res, x_ptr, a_Q12 + i * 16, 2 * offset, LPC_order);
      LANE0 {
         for (int j = 0; j < 2; j++) { i32 e; int rshift; sd_sum_sqr_shift(&e, &rshift, LPC_res + LPC_order + j * offset, subfr_length); nrgs[i * 2 + j] = e; nrgsQ[i * 2 + j] = -rshift; }
      }
      x_ptr += 2 * offset;
   }
   LANE0 {
      for (int i = 0; i < nb_subfr; i++) {
         const int lz1 = sk_clz(nrgs[i]) - 1, lz2 = sk_clz(gains[i]) - 1;
         i32 t = shl32(gains[i], lz2);
         t = sk_mulhi(t, t);
         nrgs[i] = sk_mulhi(t, shl32(nrgs[i], lz1));
         nrgsQ[i] += lz1 + 2 * lz2 - 32 - 32;
      }
   }
}
#endif

/* oracle/oc_silk_pitch.c — TEST INFRASTRUCTURE ONLY.  Plain-C restatement of the fixed-point SILK pitch estimator
 * silk_pitch_analysis_core (silk/fixed/pitch_analysis_core_FIX.c:82-590, stage-3 helpers :606-721) with what it calls:
 * silk_sum_sqr_shift (silk/sum_sqr_shift.c:36), silk_resampler_down2 (silk/resampler_down2.c:36), silk_resampler_down2_3
 * (silk/resampler_down2_3.c:39), silk_lin2log (silk/lin2log.c:36), the partial insertion sort (silk/sort.c:88), exact int32
 * correlations (celt_pitch_xcorr / celt_inner_prod in the fixed-point build).
 * Three stages: 4 kHz normalised correlation over lags 8..72 -> candidate lags; 8 kHz correlation per subframe on the candidate
 * neighbourhoods, searched with the stage-2 contour codebook and short-lag / previous-lag biases; full-rate search of lag +-2 with
 * the stage-3 contour codebook.  Checked against the compiled reference by tests/test_oracle_silk.py. */
#include "oc_silk.h"
#include "oc_silk_tables.h"

#define MIN_LAG_4K 8
#define MAX_LAG_4K 72
#define MIN_LAG_8K 16
#define MAX_LAG_8K 143
#define CSTRIDE_4K (MAX_LAG_4K + 1 - MIN_LAG_4K)              /* 65  */
#define CSTRIDE_8K (MAX_LAG_8K + 3 - (MIN_LAG_8K - 2))        /* 132 */
#define DCOMP_MIN (MIN_LAG_8K - 3)
#define DCOMP_MAX (MAX_LAG_8K + 4)
#define SF8 40                                                /* 5 ms at 8 kHz */

static s32 dot16(const s16 *a, const s16 *b, int n) { s32 s = 0; for (int i = 0; i < n; i++) s = q_addw(s, (s32)a[i] * b[i]); return s; }

void oc_silk_sum_sqr_shift(s32 *energy, int *shift, const s16 *x, int len)            /* sum_sqr_shift.c:36 */
{
   int shft = 31 - q_clz32(len);
   s32 nrg = len;
   for (int pass = 0; pass < 2; pass++) {
      int i;
      if (pass) { shft = shft + 3 - q_clz32(nrg); if (shft < 0) shft = 0; nrg = 0; }
      for (i = 0; i < len - 1; i += 2) { u32 t = (u32)((s32)x[i] * x[i]) + (u32)((s32)x[i + 1] * x[i + 1]); nrg = (s32)((u32)nrg + (t >> shft)); }
      if (i < len) { u32 t = (u32)((s32)x[i] * x[i]); nrg = (s32)((u32)nrg + (t >> shft)); }
   }
   *shift = shft; *energy = nrg;
}

void oc_silk_resampler_down2(s32 *S, s16 *out, const s16 *in, s32 inLen)              /* resampler_down2.c:36 */
{
   for (s32 k = 0; k < inLen >> 1; k++) {
      s32 in32 = (s32)in[2 * k] << 10;
      s32 Y = in32 - S[0], X = q_mlawb(Y, Y, OCS_RESAMPLER_DOWN2_1);
      s32 o = S[0] + X;  S[0] = in32 + X;
      in32 = (s32)in[2 * k + 1] << 10;
      Y = in32 - S[1];  X = q_mulwb(Y, OCS_RESAMPLER_DOWN2_0);
      o = o + S[1] + X;  S[1] = in32 + X;
      out[k] = (s16)q_sat16(q_rshift_round(o, 11));
   }
}

void oc_silk_resampler_down2_3(s32 *S, s16 *out, const s16 *in, s32 inLen)            /* resampler_down2_3.c:39 */
{
   s32 buf[480 + 4];
   const s16 *C = ocs_resampler_2_3_coefs_lq;
   memcpy(buf, S, 4 * sizeof(s32));
   s32 nIn;
   for (;;) {
      nIn = inLen < 480 ? inLen : 480;
      for (s32 k = 0; k < nIn; k++) {                                                   /* AR2, state S[4..5] */
         s32 o = S[4] + ((s32)in[k] << 8);
         buf[4 + k] = o;
         o = q_shlw(o, 2);
         S[4] = q_mlawb(S[5], o, C[0]);
         S[5] = q_mulwb(o, C[1]);
      }
      const s32 *b = buf;
      for (s32 counter = nIn; counter > 2; counter -= 3, b += 3) {
         s32 r = q_mulwb(b[0], C[2]); r = q_mlawb(r, b[1], C[3]); r = q_mlawb(r, b[2], C[5]); r = q_mlawb(r, b[3], C[4]);
         *out++ = (s16)q_sat16(q_rshift_round(r, 6));
         r = q_mulwb(b[1], C[4]); r = q_mlawb(r, b[2], C[5]); r = q_mlawb(r, b[3], C[3]); r = q_mlawb(r, b[4], C[2]);
         *out++ = (s16)q_sat16(q_rshift_round(r, 6));
      }
      in += nIn; inLen -= nIn;
      if (inLen > 0) memcpy(buf, &buf[nIn], 4 * sizeof(s32)); else break;
   }
   memcpy(S, &buf[nIn], 4 * sizeof(s32));
}

s32 oc_silk_lin2log(s32 inLin)                                                          /* lin2log.c:36, Inlines.h:52 */
{
   const int lz = q_clz32(inLin), rot = 24 - lz;
   const u32 x = (u32)inLin;
   const u32 r = rot == 0 ? x : rot < 0 ? ((x << -rot) | (x >> (32 + rot))) : ((x << (32 - rot)) | (x >> rot));
   const s32 frac = (s32)(r & 0x7f);
   return q_mlawb(frac, frac * (128 - frac), 179) + ((31 - lz) << 7);
}

/* top-K of a[0..L) in decreasing order, ties to the lower index (sort.c:88); a[] is reordered in its first K slots only */
static void topk_decreasing(s16 *a, int *idx, int L, int K)
{
   for (int i = 0; i < K; i++) idx[i] = i;
   for (int i = 1; i < L; i++) {
      const int v = a[i], last = i < K ? i - 1 : K - 1;
      if (i >= K && !(v > a[K - 1])) continue;
      int j = i < K ? i - 1 : K - 2;
      for (; j >= 0 && v > a[j]; j--) { a[j + 1] = a[j]; idx[j + 1] = idx[j]; }
      a[j + 1] = (s16)v; idx[j + 1] = i;
      (void)last;
   }
}

int oc_silk_pitch_analysis_core(const s16 *frame_unscaled, s32 *pitch_out, s16 *lagIndex, s8 *contourIndex, s32 *LTPCorr_Q15, s32 prevLag,
                                s32 search_thres1_Q16, s32 search_thres2_Q13, int Fs_kHz, int complexity, int nb_subfr)
{
   const int total_ms = 20 + nb_subfr * 5;
   const int frame_length = total_ms * Fs_kHz, len8 = total_ms * 8, len4 = total_ms * 4;
   const int sf_length = 5 * Fs_kHz, min_lag = 2 * Fs_kHz, max_lag = 18 * Fs_kHz - 1;
   s16 frame_scaled[640], f8buf[320], f4[160];
   const s16 *frame = frame_unscaled, *f8;

   /* input scaling to two bits of headroom (:144-155) */
   { s32 energy; int shift; oc_silk_sum_sqr_shift(&energy, &shift, frame_unscaled, frame_length);
     shift += 3 - q_clz32(energy);
     if (shift > 0) { shift = (shift + 1) >> 1; for (int i = 0; i < frame_length; i++) frame_scaled[i] = (s16)(frame_unscaled[i] >> shift); frame = frame_scaled; } }

   /* decimate to 8 and 4 kHz (:157-182) */
   s32 st[6];
   if (Fs_kHz == 16) { memset(st, 0, sizeof st); oc_silk_resampler_down2(st, f8buf, frame, frame_length); f8 = f8buf; }
   else if (Fs_kHz == 12) { memset(st, 0, sizeof st); oc_silk_resampler_down2_3(st, f8buf, frame, frame_length); f8 = f8buf; }
   else f8 = frame;
   memset(st, 0, sizeof st); oc_silk_resampler_down2(st, f4, f8, len8);
   for (int i = len4 - 1; i > 0; i--) f4[i] = (s16)q_sat16((s32)f4[i] + f4[i - 1]);

   /* ---- stage 1 (4 kHz, :188-252) ---- */
   s16 C[4 * CSTRIDE_8K];
   memset(C, 0, sizeof C);
   {
      const s16 *target = &f4[20 * 4];                                   /* SF_LENGTH_4KHZ << 2 */
      for (int k = 0; k < nb_subfr >> 1; k++, target += SF8) {
         const s16 *basis = target - MIN_LAG_4K;
         s32 normalizer = dot16(target, target, SF8) + dot16(basis, basis, SF8) + (s32)(s16)SF8 * 4000;
         for (int d = MIN_LAG_4K; d <= MAX_LAG_4K; d++) {
            if (d > MIN_LAG_4K) { basis--; normalizer += (s32)basis[0] * basis[0] - (s32)basis[SF8] * basis[SF8]; }
            C[k * CSTRIDE_4K + d - MIN_LAG_4K] = (s16)oc_silk_div32_varQ(dot16(target, target - d, SF8), normalizer, 14);
         }
      }
   }
   for (int i = MAX_LAG_4K; i >= MIN_LAG_4K; i--) {                        /* combine + short-lag bias */
      s32 sum = nb_subfr == 4 ? (s32)C[i - MIN_LAG_4K] + (s32)C[CSTRIDE_4K + i - MIN_LAG_4K] : (s32)C[i - MIN_LAG_4K] << 1;
      sum = q_mlawb(sum, sum, q_shlw(-i, 4));
      C[i - MIN_LAG_4K] = (s16)sum;
   }
   int length_d_srch = 4 + (complexity << 1);
   int d_srch[24];
   topk_decreasing(C, d_srch, CSTRIDE_4K, length_d_srch);
   const int Cmax = C[0];
   if (Cmax < 3277) goto unvoiced;                                       /* SILK_FIX_CONST(0.2, 14) */
   {
      const s32 threshold = q_mulwb(search_thres1_Q16, Cmax);
      for (int i = 0; i < length_d_srch; i++) {
         if (C[i] > threshold) d_srch[i] = (d_srch[i] + MIN_LAG_4K) << 1;
         else { length_d_srch = i; break; }
      }
   }
   s16 d_comp[DCOMP_MAX - DCOMP_MIN];
   int length_d_comp = 0;
   memset(d_comp, 0, sizeof d_comp);
   for (int i = 0; i < length_d_srch; i++) d_comp[d_srch[i] - DCOMP_MIN] = 1;
   for (int i = DCOMP_MAX - 1; i >= MIN_LAG_8K; i--) d_comp[i - DCOMP_MIN] += d_comp[i - 1 - DCOMP_MIN] + d_comp[i - 2 - DCOMP_MIN];
   length_d_srch = 0;
   for (int i = MIN_LAG_8K; i < MAX_LAG_8K + 1; i++) if (d_comp[i + 1 - DCOMP_MIN] > 0) d_srch[length_d_srch++] = i;
   for (int i = DCOMP_MAX - 1; i >= MIN_LAG_8K; i--) d_comp[i - DCOMP_MIN] += d_comp[i - 1 - DCOMP_MIN] + d_comp[i - 2 - DCOMP_MIN] + d_comp[i - 3 - DCOMP_MIN];
   for (int i = MIN_LAG_8K; i < DCOMP_MAX; i++) if (d_comp[i - DCOMP_MIN] > 0) d_comp[length_d_comp++] = (s16)(i - 2);

   /* ---- stage 2 (8 kHz, :303-451) ---- */
   memset(C, 0, sizeof C);
   {
      const s16 *target = &f8[20 * 8];
      for (int k = 0; k < nb_subfr; k++, target += SF8) {
         const s32 energy_target = dot16(target, target, SF8) + 1;
         for (int j = 0; j < length_d_comp; j++) {
            const int d = d_comp[j];
            const s16 *basis = target - d;
            const s32 cc = dot16(target, basis, SF8);
            C[k * CSTRIDE_8K + d - (MIN_LAG_8K - 2)] = cc > 0 ? (s16)oc_silk_div32_varQ(cc, energy_target + dot16(basis, basis, SF8), 14) : 0;
         }
      }
   }
   s32 CCmax = -2147483647 - 1, CCmax_b = CCmax;
   int CBimax = 0, lag = -1;
   s32 prevLag_log2_Q7 = 0;
   if (prevLag > 0) {
      if (Fs_kHz == 12) prevLag = (prevLag << 1) / 3; else if (Fs_kHz == 16) prevLag >>= 1;
      prevLag_log2_Q7 = oc_silk_lin2log(prevLag);
   }
   int cbk_size, nb_cbk_search;
   const s8 *Lag_CB;
   if (nb_subfr == 4) { cbk_size = 11; Lag_CB = ocs_cb_lags_stage2; nb_cbk_search = (Fs_kHz == 8 && complexity > 0) ? 11 : 3; }
   else { cbk_size = 3; Lag_CB = ocs_cb_lags_stage2_10ms; nb_cbk_search = 3; }
   for (int k = 0; k < length_d_srch; k++) {
      const int d = d_srch[k];
      s32 CCmax_new = -2147483647 - 1; int CBimax_new = 0;
      for (int j = 0; j < nb_cbk_search; j++) {
         s32 cc = 0;
         for (int i = 0; i < nb_subfr; i++) cc += C[i * CSTRIDE_8K + d + Lag_CB[i * cbk_size + j] - (MIN_LAG_8K - 2)];
         if (cc > CCmax_new) { CCmax_new = cc; CBimax_new = j; }
      }
      const s32 lag_log2_Q7 = oc_silk_lin2log(d);
      s32 CCmax_new_b = CCmax_new - (q_mulbb(nb_subfr * 1638, lag_log2_Q7) >> 7);            /* SILK_FIX_CONST(0.2, 13) = 1638 */
      if (prevLag > 0) {
         s32 dl = lag_log2_Q7 - prevLag_log2_Q7;
         dl = q_mulbb(dl, dl) >> 7;
         s32 bias = q_mulbb(nb_subfr * 1638, *LTPCorr_Q15) >> 15;
         bias = (bias * dl) / (dl + 64);                                                      /* SILK_FIX_CONST(0.5, 7) */
         CCmax_new_b -= bias;
      }
      if (CCmax_new_b > CCmax_b && CCmax_new > q_mulbb(nb_subfr, search_thres2_Q13) && ocs_cb_lags_stage2[CBimax_new] <= MIN_LAG_8K) {
         CCmax_b = CCmax_new_b; CCmax = CCmax_new; lag = d; CBimax = CBimax_new;
      }
   }
   if (lag == -1) goto unvoiced;
   *LTPCorr_Q15 = (CCmax / nb_subfr) << 2;

   if (Fs_kHz > 8) {
      /* ---- stage 3 (full rate, :465-566) ---- */
      const int CBimax_old = CBimax;
      if (Fs_kHz == 12) lag = q_mulbb(lag, 3) >> 1; else if (Fs_kHz == 16) lag <<= 1; else lag = q_mulbb(lag, 3);
      lag = lag < min_lag ? min_lag : lag > max_lag ? max_lag : lag;
      const int start_lag = lag - 2 > min_lag ? lag - 2 : min_lag, end_lag = lag + 2 < max_lag ? lag + 2 : max_lag;
      int lag_new = lag;
      CBimax = 0;
      CCmax = -2147483647 - 1;
      for (int k = 0; k < nb_subfr; k++) pitch_out[k] = lag + 2 * ocs_cb_lags_stage2[k * 11 + CBimax_old];
      const s8 *Lag_range;
      if (nb_subfr == 4) { nb_cbk_search = ocs_nb_cbk_searchs_stage3[complexity]; cbk_size = 34; Lag_CB = ocs_cb_lags_stage3; Lag_range = &ocs_lag_range_stage3[complexity * 8]; }
      else { nb_cbk_search = 12; cbk_size = 12; Lag_CB = ocs_cb_lags_stage3_10ms; Lag_range = ocs_lag_range_stage3_10ms; }
      /* per subframe: correlation and energy of the target against every lag in [start_lag + lo, start_lag + hi] (:606-721) */
      s32 xc[4][22], en[4][22];
      {
         const s16 *target = &frame[sf_length << 2];
         for (int k = 0; k < nb_subfr; k++, target += sf_length) {
            const int lo = Lag_range[2 * k], hi = Lag_range[2 * k + 1];
            for (int j = lo; j <= hi; j++) xc[k][j - lo] = dot16(target, target - start_lag - j, sf_length);
            const s16 *basis = target - (start_lag + lo);
            s32 e = dot16(basis, basis, sf_length);
            en[k][0] = e;
            for (int i = 1; i < hi - lo + 1; i++) {
               e -= (s32)basis[sf_length - i] * basis[sf_length - i];
               e = q_add_sat(e, (s32)basis[-i] * basis[-i]);
               en[k][i] = e;
            }
         }
      }
      const s32 contour_bias_Q15 = 1638 / lag;                                               /* SILK_FIX_CONST(0.05, 15) */
      const s16 *target = &frame[20 * Fs_kHz];
      const s32 energy_target = dot16(target, target, nb_subfr * sf_length) + 1;
      int lag_counter = 0;
      for (int d = start_lag; d <= end_lag; d++, lag_counter++) {
         for (int j = 0; j < nb_cbk_search; j++) {
            s32 cc = 0, energy = energy_target;
            for (int k = 0; k < nb_subfr; k++) {
               const int idx = Lag_CB[k * cbk_size + j] - Lag_range[2 * k] + lag_counter;
               cc += xc[k][idx]; energy += en[k][idx];
            }
            s32 CCmax_new = 0;
            if (cc > 0) CCmax_new = q_mulwb(oc_silk_div32_varQ(cc, energy, 14), 32767 - contour_bias_Q15 * j);
            if (CCmax_new > CCmax && d + ocs_cb_lags_stage3[j] <= max_lag) { CCmax = CCmax_new; lag_new = d; CBimax = j; }
         }
      }
      for (int k = 0; k < nb_subfr; k++) {
         int p = lag_new + Lag_CB[k * cbk_size + CBimax];
         pitch_out[k] = p < min_lag ? min_lag : p > 18 * Fs_kHz ? 18 * Fs_kHz : p;
      }
      *lagIndex = (s16)(lag_new - min_lag);
      *contourIndex = (s8)CBimax;
   } else {
      for (int k = 0; k < nb_subfr; k++) {
         int p = lag + Lag_CB[k * cbk_size + CBimax];
         pitch_out[k] = p < MIN_LAG_8K ? MIN_LAG_8K : p > 144 ? 144 : p;
      }
      *lagIndex = (s16)(lag - MIN_LAG_8K);
      *contourIndex = (s8)CBimax;
   }
   return 0;
unvoiced:
   memset(pitch_out, 0, (size_t)nb_subfr * sizeof(s32));
   *LTPCorr_Q15 = 0; *lagIndex = 0; *contourIndex = 0;
   return 1;
}

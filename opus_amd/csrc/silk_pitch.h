/* silk_pitch.h — the fixed-point SILK pitch estimator silk_pitch_analysis_core (silk/fixed/pitch_analysis_core_FIX.c:82-590, stage-3
 * helpers :606-721) as a wave-per-frame kernel body: one 64-lane wave analyses one 30/40 ms buffer.
 *
 * What is parallel: every correlation.  Stage 1 (4 kHz): 2 x 65 lags x 40 taps, lane = lag.  Stage 2 (8 kHz): up to 131 candidate lags
 * x 4 subframes x 40 taps, lane = candidate.  Stage 3 (full rate): 4 subframes x <= 22 lags x 80 taps, lane = (subframe, lag), then
 * 5 lags x <= 34 contour codebooks, lane = (lag, codebook).  The reference's running normalisers (add the sample entering the window,
 * subtract the one leaving) are exact integer sums, so each lane computes its own window sum and gets the same word.  Selection
 * (partial sort of the 4 kHz correlations, candidate expansion, the biased stage-2 maximum, the stage-3 maximum) is done with wave
 * reductions whose tie rules reproduce the reference's scan order (value first, then lowest index).
 * What is serial: the 2:1 / 3:2 decimators (allpass / AR recursions that floor every step) run on lane 0 — 480 steps per frame.
 * The whole working set (input, 8 kHz and 4 kHz copies, correlation matrices) is 5.6 KB of LDS; HBM sees the input once. */
#ifndef OPUS_AMD_SILK_PITCH_H
#define OPUS_AMD_SILK_PITCH_H
#include "silk_tables.h"

#define PE_MIN_LAG_4K 8
#define PE_MAX_LAG_4K 72
#define PE_MIN_LAG_8K 16
#define PE_MAX_LAG_8K 143
#define PE_CSTRIDE_4K 65
#define PE_CSTRIDE_8K 132
#define PE_DCOMP_MIN 13
#define PE_DCOMP_MAX 147
#define PE_SF8 40

struct OaPitchCfg { i32 Fs_kHz, complexity, nb_subfr; };
struct OaPitchIn  { i32 prevLag, LTPCorr_Q15, search_thres1_Q16, search_thres2_Q13; };
struct OaPitchOut { i32 pitch[4]; i32 LTPCorr_Q15; i16 lagIndex; i8 contourIndex; i8 unvoiced; };

struct PitchLdsCore {
   i16 f8[320 + 8], f4[160 + 8];
   i16 C[4 * PE_CSTRIDE_8K];
   i16 mark[160], conv1[160], d_comp[160];
   i32 d_srch[24];
   i32 xc[4][24], en[4][24];
   i32 sh[16];
};
struct PitchLds : PitchLdsCore { i16 frame[640 + 8]; };      /* with room for a copy of the input (the standalone kernel: its input is in HBM) */
enum { PSH_LEN_SRCH = 0, PSH_LEN_COMP, PSH_LAG, PSH_CBIMAX, PSH_CCMAX, PSH_LAGNEW };

WV_DEV i32 pe_dot(const WV_LDS i16 *a, const WV_LDS i16 *b, int n) { i32 s = 0; for (int i = 0; i < n; i++) s = add32(s, (i32)a[i] * (i32)b[i]); return s; }
WV_DEV i32 pe_lin2log(i32 x)                                                                  /* silk/lin2log.c:36, Inlines.h:52 */
{
   const int lz = sk_clz(x), rot = 24 - lz;
   const u32 u = (u32)x;
   const u32 r = rot == 0 ? u : rot < 0 ? ((u << -rot) | (u >> (32 + rot))) : ((u << (32 - rot)) | (u >> rot));
   const i32 frac = (i32)(r & 0x7f);
   return sk_mlawb(frac, frac * (128 - frac), 179) + ((31 - lz) << 7);
}
/* lane 0: silk_resampler_down2 (silk/resampler_down2.c:36) with zero initial state */
WV_DEV void pe_down2_l0(WV_LDS i16 *out, const WV_LDS i16 *in, int inLen)
{
   i32 S0 = 0, S1 = 0;
   for (int k = 0; k < inLen >> 1; k++) {
      i32 in32 = shl32(in[2 * k], 10);
      i32 Y = in32 - S0, X = sk_mlawb(Y, Y, 39809 - 65536);
      i32 o = S0 + X;  S0 = in32 + X;
      in32 = shl32(in[2 * k + 1], 10);
      Y = in32 - S1;  X = sk_mulwb(Y, 9872);
      o = o + S1 + X;  S1 = in32 + X;
      out[k] = (i16)sk_sat16(sk_rround(o, 11));
   }
}
/* lane 0: silk_resampler_down2_3 (silk/resampler_down2_3.c:39) with zero initial state; inLen <= 480 (one batch) */
WV_DEV void pe_down2_3_l0(WV_LDS i16 *out, const WV_LDS i16 *in, int inLen)
{
   const i16 *C = sk_resampler_2_3_coefs_lq;
   i32 b0 = 0, b1 = 0, b2 = 0, b3 = 0, A0 = 0, A1 = 0;                    /* the 4 buffered AR outputs, AR2 state */
   int no = 0;
   for (int k = 0; k + 2 < inLen + 0 && k + 3 <= inLen; k += 3) {
      i32 v[3];
      for (int t = 0; t < 3; t++) { i32 o = A0 + shl32(in[k + t], 8); v[t] = o; o = shl32(o, 2); A0 = sk_mlawb(A1, o, C[0]); A1 = sk_mulwb(o, C[1]); }
      /* window = b0 b1 b2 b3 v0 : outputs use buf_ptr[0..4] where buf_ptr points at b0 */
      i32 r = sk_mulwb(b0, C[2]); r = sk_mlawb(r, b1, C[3]); r = sk_mlawb(r, b2, C[5]); r = sk_mlawb(r, b3, C[4]);
      out[no++] = (i16)sk_sat16(sk_rround(r, 6));
      r = sk_mulwb(b1, C[4]); r = sk_mlawb(r, b2, C[5]); r = sk_mlawb(r, b3, C[3]); r = sk_mlawb(r, v[0], C[2]);
      out[no++] = (i16)sk_sat16(sk_rround(r, 6));
      b0 = b3; b1 = v[0]; b2 = v[1]; b3 = v[2];
   }
}

/* One frame on one wave.  frame: (20 + 5*nb_subfr) ms of int16 at Fs_kHz in HBM. */
/* fr: the wave's working copy of the input, scaled in place to two bits of headroom.  frame_g == nullptr: the input is in fr already (the encoder hands over its LDS residual buffer; the
 * return value tells it whether the scaling has changed it).  Returns the down-scaling shift that was applied. */
WV_DEV int silk_pitch_analysis_wave(const OaPitchCfg cfg, WV_LDS PitchLdsCore *L, WV_LDS i16 *fr, const i16 *frame_g, const OaPitchIn *pin, OaPitchOut *pout)
{
   const int lane = wv_lane();
   const int Fs = cfg.Fs_kHz, cx = cfg.complexity, nb = cfg.nb_subfr;
   const int total_ms = 20 + nb * 5, flen = total_ms * Fs, len8 = total_ms * 8, len4 = total_ms * 4;
   const int sf_length = 5 * Fs, min_lag = 2 * Fs, max_lag = 18 * Fs - 1;
   i32 prevLag = pin->prevLag;
   const i32 LTPCorr_in = pin->LTPCorr_Q15, thres1 = pin->search_thres1_Q16, thres2 = pin->search_thres2_Q13;
   bool unvoiced = false;

   /* ---- input energy -> down-scaling to two bits of headroom (:144-155; silk/sum_sqr_shift.c:36: two passes of sum((x0^2+x1^2) >> shft)) ---- */
   int shft = 31 - sk_clz(flen);
   u32 part = 0;
   if (frame_g) { for (int i = 2 * lane; i < flen; i += 2 * WV_WIDTH) { fr[i] = frame_g[i]; fr[i + 1] = frame_g[i + 1]; } }
   for (int i = 2 * lane; i < flen; i += 2 * WV_WIDTH) { i32 a = fr[i], b = fr[i + 1]; part += ((u32)(a * a) + (u32)(b * b)) >> shft; }
   i32 nrg = (i32)((u32)flen + wv_sumu(part));
   shft = imax(0, shft + 3 - sk_clz(nrg));
   wv_sync();
   part = 0;
   for (int i = 2 * lane; i < flen; i += 2 * WV_WIDTH) { i32 a = fr[i], b = fr[i + 1]; part += ((u32)(a * a) + (u32)(b * b)) >> shft; }
   nrg = (i32)wv_sumu(part);
   int shift = shft + 3 - sk_clz(nrg);
   if (shift > 0) { shift = (shift + 1) >> 1; for (int i = lane; i < flen; i += WV_WIDTH) fr[i] = (i16)(fr[i] >> shift); }
   const int applied_shift = shift > 0 ? shift : 0;
   wv_sync();

   /* ---- decimation to 8 kHz and 4 kHz (:157-182): serial recursions, lane 0 ---- */
   if (Fs == 8) { for (int i = lane; i < len8; i += WV_WIDTH) L->f8[i] = fr[i]; }
   else if (lane == 0) { if (Fs == 16) pe_down2_l0(L->f8, fr, flen); else pe_down2_3_l0(L->f8, fr, flen); }
   wv_sync();
   if (lane == 0) pe_down2_l0(L->f4, L->f8, len8);
   wv_sync();
   {  /* first-order low-pass, every output from the unfiltered neighbours */
      i32 v[3]; int n = 0;
      for (int i = lane; i < len4; i += WV_WIDTH) v[n++] = i > 0 ? sk_sat16((i32)L->f4[i] + L->f4[i - 1]) : L->f4[0];
      wv_sync();
      n = 0;
      for (int i = lane; i < len4; i += WV_WIDTH) L->f4[i] = (i16)v[n++];
   }
   for (int i = lane; i < 4 * PE_CSTRIDE_8K; i += WV_WIDTH) L->C[i] = 0;
   wv_sync();

   /* ---- stage 1, 4 kHz (:188-252): normalised correlation per lag, two 10 ms halves ---- */
   for (int k = 0; k < nb >> 1; k++) {
      const WV_LDS i16 *target = &L->f4[80 + k * PE_SF8];
      const i32 et = pe_dot(target, target, PE_SF8) + (i32)PE_SF8 * 4000;
      for (int d = PE_MIN_LAG_4K + lane; d <= PE_MAX_LAG_4K; d += WV_WIDTH) {
         const WV_LDS i16 *basis = target - d;
         L->C[k * PE_CSTRIDE_4K + d - PE_MIN_LAG_4K] = (i16)sk_div32_varQ(pe_dot(target, basis, PE_SF8), et + pe_dot(basis, basis, PE_SF8), 14);
      }
   }
   wv_sync();
   {
      i32 v[2]; int n = 0;
      for (int i = PE_MIN_LAG_4K + lane; i <= PE_MAX_LAG_4K; i += WV_WIDTH) {
         i32 sum = nb == 4 ? (i32)L->C[i - PE_MIN_LAG_4K] + (i32)L->C[PE_CSTRIDE_4K + i - PE_MIN_LAG_4K] : shl32(L->C[i - PE_MIN_LAG_4K], 1);
         v[n++] = (i16)sk_mlawb(sum, sum, shl32(-i, 4));
      }
      wv_sync();
      n = 0;
      for (int i = PE_MIN_LAG_4K + lane; i <= PE_MAX_LAG_4K; i += WV_WIDTH) L->C[i - PE_MIN_LAG_4K] = (i16)v[n++];
   }
   wv_sync();
   /* the K largest in decreasing order, ties to the lower index (silk/sort.c:88): K rounds of a packed wave maximum */
   int length_d_srch = 4 + (cx << 1);
   i32 Cmax = 0;
   {
      i32 key0 = ((i32)L->C[lane] + 32768) * 128 + (127 - lane);
      i32 key1 = lane == 0 ? ((i32)L->C[64] + 32768) * 128 + (127 - 64) : -1;
      i32 sortedv = 0, sortedi = 0;                                     /* lane r keeps the r-th winner */
      for (int r = 0; r < length_d_srch; r++) {
         const i32 best = wv_max(imax(key0, key1));
         const int idx = 127 - (best & 127), val = (best >> 7) - 32768;
         if (lane == r) { sortedv = val; sortedi = idx; }
         if (idx == lane) key0 = -1;
         if (idx == 64 && lane == 0) key1 = -1;
      }
      Cmax = wv_bcast(sortedv, 0);
      /* entries above the relative threshold, converted to 8 kHz lags (:235-246) */
      const i32 threshold = sk_mulwb(thres1, Cmax);
      const u64 pass = wv_ballot(lane < length_d_srch && sortedv > threshold);
      int cnt = 0; while (cnt < length_d_srch && ((pass >> cnt) & 1)) cnt++;           /* stops at the first failure, like the reference */
      for (int i = lane; i < 160; i += WV_WIDTH) L->mark[i] = 0;
      wv_sync();
      if (lane < cnt) L->mark[((sortedi + PE_MIN_LAG_4K) << 1) - PE_DCOMP_MIN] = 1;
      length_d_srch = cnt;
   }
   wv_sync();
   if (Cmax < 3277) unvoiced = true;                                    /* SILK_FIX_CONST(0.2, 14): wave-uniform */
   if (!unvoiced) {
      /* candidate expansion (:250-283): two box filters over the marked lags, then compaction in increasing lag order */
      {
         i32 v[3]; int n = 0;
         for (int i = PE_DCOMP_MIN + lane; i < PE_DCOMP_MAX; i += WV_WIDTH) {
            const int x = i - PE_DCOMP_MIN;
            v[n++] = i >= PE_MIN_LAG_8K ? L->mark[x] + L->mark[x - 1] + L->mark[x - 2] : L->mark[x];
         }
         n = 0;
         for (int i = PE_DCOMP_MIN + lane; i < PE_DCOMP_MAX; i += WV_WIDTH) L->conv1[i - PE_DCOMP_MIN] = (i16)v[n++];
      }
      wv_sync();
      int ns = 0, nc = 0;
      for (int base = 0; base < 192; base += WV_WIDTH) {
         const int i = PE_MIN_LAG_8K + base + lane;
         const bool fs = i < PE_MAX_LAG_8K + 1 && L->conv1[i + 1 - PE_DCOMP_MIN] > 0;
         const u64 ms = wv_ballot(fs);
         if (fs) L->d_srch[ns + __builtin_popcountll(ms & ((1ull << lane) - 1))] = i;
         ns += __builtin_popcountll(ms);
         bool fc = false;
         if (i < PE_DCOMP_MAX) { const int x = i - PE_DCOMP_MIN; fc = (L->conv1[x] + L->conv1[x - 1] + L->conv1[x - 2] + L->conv1[x - 3]) > 0; }
         const u64 mc = wv_ballot(fc);
         if (fc) L->d_comp[nc + __builtin_popcountll(mc & ((1ull << lane) - 1))] = (i16)(i - 2);
         nc += __builtin_popcountll(mc);
      }
      length_d_srch = ns;
      const int length_d_comp = nc;
      wv_sync();

      /* ---- stage 2, 8 kHz (:303-351): per subframe, correlation with each candidate lag ---- */
      for (int i = lane; i < 4 * PE_CSTRIDE_8K; i += WV_WIDTH) L->C[i] = 0;
      wv_sync();
      for (int k = 0; k < nb; k++) {
         const WV_LDS i16 *target = &L->f8[160 + k * PE_SF8];
         const i32 et = pe_dot(target, target, PE_SF8) + 1;
         for (int j = lane; j < length_d_comp; j += WV_WIDTH) {
            const int d = L->d_comp[j];
            const WV_LDS i16 *basis = target - d;
            const i32 cc = pe_dot(target, basis, PE_SF8);
            L->C[k * PE_CSTRIDE_8K + d - (PE_MIN_LAG_8K - 2)] = cc > 0 ? (i16)sk_div32_varQ(cc, et + pe_dot(basis, basis, PE_SF8), 14) : (i16)0;
         }
      }
      wv_sync();
      /* stage-2 codebook search with short-lag and previous-lag biases (:353-436): lane = candidate lag */
      i32 prevLag_log2_Q7 = 0;
      if (prevLag > 0) {
         if (Fs == 12) prevLag = shl32(prevLag, 1) / 3; else if (Fs == 16) prevLag >>= 1;
         prevLag_log2_Q7 = pe_lin2log(prevLag);
      }
      int cbk_size, nb_cbk_search; const i8 *Lag_CB;
      if (nb == 4) { cbk_size = 11; Lag_CB = sk_cb_lags_stage2; nb_cbk_search = (Fs == 8 && cx > 0) ? 11 : 3; }
      else { cbk_size = 3; Lag_CB = sk_cb_lags_stage2_10ms; nb_cbk_search = 3; }
      i32 myb = (i32)0x80000000, mycc = 0; int mycb = 0, myd = 0;
      if (lane < length_d_srch) {
         const int d = L->d_srch[lane];
         i32 CCmax_new = (i32)0x80000000; int CBimax_new = 0;
         for (int j = 0; j < nb_cbk_search; j++) {
            i32 cc = 0;
            for (int i = 0; i < nb; i++) cc += L->C[i * PE_CSTRIDE_8K + d + Lag_CB[i * cbk_size + j] - (PE_MIN_LAG_8K - 2)];
            if (cc > CCmax_new) { CCmax_new = cc; CBimax_new = j; }
         }
         const i32 lag_log2_Q7 = pe_lin2log(d);
         i32 b = CCmax_new - (sk_mulbb(nb * 1638, lag_log2_Q7) >> 7);
         if (prevLag > 0) {
            i32 dl = lag_log2_Q7 - prevLag_log2_Q7;
            dl = sk_mulbb(dl, dl) >> 7;
            i32 bias = sk_mulbb(nb * 1638, LTPCorr_in) >> 15;
            bias = (bias * dl) / (dl + 64);
            b -= bias;
         }
         if (CCmax_new > sk_mulbb(nb, thres2) && sk_cb_lags_stage2[CBimax_new] <= PE_MIN_LAG_8K) { myb = b; mycc = CCmax_new; mycb = CBimax_new; myd = d; }
      }
      const i32 bestb = wv_max(myb);
      int lag = -1, CBimax = 0; i32 CCmax = 0;
      if (bestb != (i32)0x80000000) {
         const u64 m = wv_ballot(myb == bestb);
         const int w = __builtin_ctzll(m);                               /* first candidate reaching the maximum (strict > in the reference) */
         lag = wv_bcast(myd, w); CBimax = wv_bcast(mycb, w); CCmax = wv_bcast(mycc, w);
      }
      if (lag == -1) unvoiced = true;
      else {
         const i32 LTPCorr_out = shl32(CCmax / nb, 2);
         i32 pitch[4] = { 0, 0, 0, 0 }; int lagIndex, contourIndex;
         if (Fs > 8) {
            /* ---- stage 3, full rate (:465-566) ---- */
            if (Fs == 12) lag = sk_mulbb(lag, 3) >> 1; else lag = shl32(lag, 1);
            lag = lag < min_lag ? min_lag : lag > max_lag ? max_lag : lag;
            const int start_lag = imax(lag - 2, min_lag), end_lag = imin(lag + 2, max_lag);
            const i8 *Lag_range;
            if (nb == 4) { nb_cbk_search = sk_nb_cbk_searchs_stage3[cx]; cbk_size = 34; Lag_CB = sk_cb_lags_stage3; Lag_range = &sk_lag_range_stage3[cx * 8]; }
            else { nb_cbk_search = 12; cbk_size = 12; Lag_CB = sk_cb_lags_stage3_10ms; Lag_range = sk_lag_range_stage3_10ms; }
            for (int it = lane; it < nb * 24; it += WV_WIDTH) {           /* correlations: lane = (subframe, lag offset) */
               const int k = it / 24, jj = it - k * 24, lo = Lag_range[2 * k], hi = Lag_range[2 * k + 1];
               if (jj <= hi - lo) { const WV_LDS i16 *target = &fr[(sf_length << 2) + k * sf_length]; L->xc[k][jj] = pe_dot(target, target - start_lag - lo - jj, sf_length); }
            }
            if (lane < nb) {                                             /* energies: the reference's saturating recursion, one lane per subframe */
               const int k = lane, lo = Lag_range[2 * k], hi = Lag_range[2 * k + 1];
               const WV_LDS i16 *basis = &fr[(sf_length << 2) + k * sf_length] - (start_lag + lo);
               i32 e = pe_dot(basis, basis, sf_length);
               L->en[k][0] = e;
               for (int i = 1; i < hi - lo + 1; i++) {
                  e -= (i32)basis[sf_length - i] * basis[sf_length - i];
                  e = sk_add_sat(e, (i32)basis[-i] * basis[-i]);
                  L->en[k][i] = e;
               }
            }
            i32 etp = 0;
            for (int i = lane; i < nb * sf_length; i += WV_WIDTH) { const i32 v = fr[20 * Fs + i]; etp = add32(etp, v * v); }
            const i32 energy_target = wv_sum(etp) + 1;
            wv_sync();
            const i32 contour_bias_Q15 = 1638 / lag;
            const int nd = end_lag - start_lag + 1, ncomb = nd * nb_cbk_search;
            i32 bestv = (i32)0x80000000; int besti = 0x7fffffff;
            for (int it = lane; it < ncomb; it += WV_WIDTH) {             /* lane = (lag, codebook), reference scan order = increasing it */
               const int lc = it / nb_cbk_search, j = it - lc * nb_cbk_search, d = start_lag + lc;
               i32 cc = 0, energy = energy_target;
               for (int k = 0; k < nb; k++) { const int idx = Lag_CB[k * cbk_size + j] - Lag_range[2 * k] + lc; cc += L->xc[k][idx]; energy += L->en[k][idx]; }
               i32 v = 0;
               if (cc > 0) v = sk_mulwb(sk_div32_varQ(cc, energy, 14), 32767 - contour_bias_Q15 * j);
               if (d + sk_cb_lags_stage3[j] <= max_lag && v > bestv) { bestv = v; besti = it; }   /* per lane: its own items come in increasing it */
            }
            const i32 gv = wv_max(bestv);
            const int gi = -wv_max(bestv == gv ? -besti : (i32)0x80000001);   /* smallest scan index among the maxima */
            const int lc = gi / nb_cbk_search, CB3 = gi - lc * nb_cbk_search, lag_new = start_lag + lc;
            for (int k = 0; k < nb; k++) { int p = lag_new + Lag_CB[k * cbk_size + CB3]; pitch[k] = p < min_lag ? min_lag : p > 18 * Fs ? 18 * Fs : p; }
            lagIndex = lag_new - min_lag; contourIndex = CB3;
         } else {
            for (int k = 0; k < nb; k++) { int p = lag + Lag_CB[k * cbk_size + CBimax]; pitch[k] = p < PE_MIN_LAG_8K ? PE_MIN_LAG_8K : p > 144 ? 144 : p; }
            lagIndex = lag - PE_MIN_LAG_8K; contourIndex = CBimax;
         }
         if (lane == 0) {
            for (int k = 0; k < 4; k++) pout->pitch[k] = pitch[k];
            pout->LTPCorr_Q15 = LTPCorr_out; pout->lagIndex = (i16)lagIndex; pout->contourIndex = (i8)contourIndex; pout->unvoiced = 0;
         }
      }
   }
   if (unvoiced && lane == 0) { for (int k = 0; k < 4; k++) pout->pitch[k] = 0; pout->LTPCorr_Q15 = 0; pout->lagIndex = 0; pout->contourIndex = 0; pout->unvoiced = 1; }
   return applied_shift;
}
#endif

/* celt_dec_plc.h — CELT packet-loss concealment on the wave (celt/celt_decoder.c: celt_plc_pitch_search :552, prefilter_and_fold :576,
 * celt_decode_lost :679; celt/celt_lpc.c: _celt_lpc :37, celt_fir :140, celt_iir :186, _celt_autocorr :284).
 * Pitch PLC: LPC analysis of the last 1024 output samples, excitation = FIR residual, periodic extrapolation with measured decay,
 * LPC synthesis (serial IIR: taps on 24 lanes, one wave reduction per sample), energy guard.  Noise PLC: decayed band energies,
 * LCG noise, the regular synthesis.  Working buffers: region A (excitation, scratch), region BC (the N+overlap new samples). */
#ifndef OPUS_AMD_CELT_DEC_PLC_H
#define OPUS_AMD_CELT_DEC_PLC_H
#define PLC_PITCH_LAG_MAX 720
#define PLC_PITCH_LAG_MIN 100
#define PLC_LPC_ORDER 24

/* one channel's synthesis history as a sample source: index 0 = oldest of the 2048 kept samples */
struct HistSrc { const i32 *hist; int head; };
WV_DEV i32 src_at(const HistSrc &h, int j) { return h.hist[(h.head + j) & (OA_DEC_HISTORY - 1)]; }

/* _celt_lpc (celt_lpc.c:37), order 24, lane 0; arrays in LDS */
WV_DEV void celt_lpc24_l0(WV_LDS i32 *lpc /* 24 */, WV_LDS i32 *out16 /* 24, int16 values */, const WV_LDS i32 *ac /* 25 */)
{
   const int p = PLC_LPC_ORDER;
   i32 r, error = ac[0];
   for (int i = 0; i < p; i++) lpc[i] = 0;
   if (ac[0] != 0) {
      for (int i = 0; i < p; i++) {
         i64 acc = 0;
         for (int j = 0; j < i; j++) acc += (i64)lpc[j] * (i64)ac[i - j];
         i32 rr = (i32)(acc >> 31);
         rr += ac[i + 1] >> 6;
         r = neg32(fx_frac_div32(shl32(rr, 6), error));
         lpc[i] = r >> 6;
         for (int j = 0; j < (i + 1) >> 1; j++) {
            i32 t1 = lpc[j], t2 = lpc[i - 1 - j];
            lpc[j] = t1 + mult32_32_q31(r, t2);
            lpc[i - 1 - j] = t2 + mult32_32_q31(r, t1);
         }
         error = error - mult32_32_q31(mult32_32_q31(r, r), error);
         if (error <= (ac[0] >> 10)) break;
      }
   }
   int iter, idx = 0;
   for (iter = 0; iter < 10; iter++) {
      i32 maxabs = 0;
      for (int i = 0; i < p; i++) { i32 a = iabs(lpc[i]); if (a > maxabs) { maxabs = a; idx = i; } }
      maxabs = pshr32(maxabs, 13);
      if (maxabs > 32767) {
         maxabs = imin(maxabs, 163838);
         i32 chirp_Q16 = QC32(0.999, 16) - shl32(maxabs - 32767, 14) / ((maxabs * (idx + 1)) >> 2);
         i32 chirp_minus_one_Q16 = chirp_Q16 - 65536;
         for (int i = 0; i < p - 1; i++) {
            lpc[i] = mult32_32_q16(chirp_Q16, lpc[i]);
            chirp_Q16 += pshr32(chirp_Q16 * chirp_minus_one_Q16, 16);
         }
         lpc[p - 1] = mult32_32_q16(chirp_Q16, lpc[p - 1]);
      } else break;
   }
   if (iter == 10) { for (int i = 0; i < p; i++) out16[i] = 0; out16[0] = 4096; }
   else for (int i = 0; i < p; i++) out16[i] = extract16(pshr32(lpc[i], 13));
}

/* prefilter_and_fold (celt_decoder.c:576): undo the post-filter on the overlap head of the concealed signal and fold it (TDAC) */
WV_DEVN void prefilter_and_fold_wave(WV_LDS DecLds *L, const OaDecStream *gs, int CC)
{
   const int overlap = OA_OVERLAP, lane = wv_lane();
   WV_LDS OaDecScalars *st = &L->st;
   const i16 gains[3][3] = {
      {QC16(0.3066406250f, 15), QC16(0.2170410156f, 15), QC16(0.1296386719f, 15)},
      {QC16(0.4638671875f, 15), QC16(0.2680664062f, 15), QC16(0.f, 15)},
      {QC16(0.7998046875f, 15), QC16(0.1000976562f, 15), QC16(0.f, 15)}};
   const int head = wv_uni(st->hist_head);
   const i16 g0 = (i16)-wv_uni(st->postfilter_gain_old), g1 = (i16)-wv_uni(st->postfilter_gain);
   const int T1 = imax(wv_uni(st->postfilter_period), OA_MIN_PERIOD), tapset1 = wv_uni(st->postfilter_tapset);
   const i16 g10 = (i16)mult_coef_taps(g1, gains[tapset1][0]), g11 = (i16)mult_coef_taps(g1, gains[tapset1][1]), g12 = (i16)mult_coef_taps(g1, gains[tapset1][2]);
   for (int c = 0; c < CC; c++) {
      SynSrc s; s.cur = L->BC.syn[c]; s.hist = gs->hist + c * OA_DEC_HISTORY; s.head = head;
      i32 e[2];
      /* comb_filter(etmp, x, T0, T1, N = overlap, g0, g1, ..., overlap = 0): no cross-fade -> constant filter with (T1, g1), or a plain copy */
      for (int t = 0; t < 2; t++) {
         const int i = lane + t * WV_WIDTH;
         e[t] = 0;
         if (i < overlap) {
            if ((g0 == 0 && g1 == 0) || g1 == 0) e[t] = syn_at(s, i);
            else {
               i32 v = add32(add32(add32(syn_at(s, i), mult_coef_32(g10, syn_at(s, i - T1))), mult_coef_32(g11, add32(syn_at(s, i - T1 + 1), syn_at(s, i - T1 - 1)))),
                     mult_coef_32(g12, add32(syn_at(s, i - T1 + 2), syn_at(s, i - T1 - 2))));
               e[t] = saturate(sub32(v, 1), SIG_SAT);
            }
         }
      }
      /* fold: x[i] = w[i] * etmp[overlap-1-i] + w[overlap-1-i] * etmp[i], i < overlap/2 (etmp lives in registers: 2 per lane) */
      wv_sync();
      WV_LDS i32 *tmp = (WV_LDS i32 *)L->scr;                 /* 126 words available: exactly 120 needed */
      for (int t = 0; t < 2; t++) { const int i = lane + t * WV_WIDTH; if (i < overlap) tmp[i] = e[t]; }
      wv_sync();
      FOR_LANES(i, overlap / 2) L->BC.syn[c][i] = mult16_32_q15(ct_window[i], tmp[overlap - 1 - i]) + mult16_32_q15(ct_window[overlap - i - 1], tmp[i]);
      wv_sync();
   }
}

/* celt_iir (celt_lpc.c:186) in place on y[0..n): taps on lanes 0..23, the 16-bit output history rides a lane shift register */
WV_DEV void celt_iir24_wave(WV_LDS i32 *y, const WV_LDS i32 *lpc16, int n, i32 hist_init /* lane j: out16[-1-j] */)
{
   const int lane = wv_lane();
   const i32 den = lane < PLC_LPC_ORDER ? lpc16[lane] : 0;
   i32 hist = lane < PLC_LPC_ORDER ? hist_init : 0;
   for (int i0 = 0; i0 < n; i0 += WV_WIDTH) {
      const int cnt = imin(WV_WIDTH, n - i0);
      i32 xv = lane < cnt ? y[i0 + lane] : 0, ov = 0;
      for (int k = 0; k < cnt; k++) {
         const i32 s = wv_sum(mult16_16(den, hist));
         const i32 sum = sub32(wv_bcast(xv, k), s);
         const i32 o16 = sround16(sum, SIG_SHIFT);
         hist = wv_shift_up1(hist, o16);
         if (lane >= PLC_LPC_ORDER) hist = 0;
         ov = wv_writelane(sum, k, ov);
      }
      wv_sync();
      if (lane < cnt) y[i0 + lane] = ov;
      wv_sync();
   }
}

/* celt_decode_lost (celt_decoder.c:679).  Produces the N concealed samples (+ overlap tail) in L->BC.syn, like a decoded frame. */
WV_DEVN void celt_decode_lost_wave(WV_LDS DecLds *L, OaDecStream *gs, int N, int LM)
{
   WV_LDS DecShared *sh = &L->sh;
   WV_LDS OaDecScalars *st = &L->st;
   const int overlap = OA_OVERLAP, lane = wv_lane();
   const int C = wv_uni(st->channels), start = wv_uni(st->start), head = wv_uni(st->hist_head);
   const int loss_duration = wv_uni(st->loss_duration);
   int curr_frame_type = 4;                                  /* FRAME_PLC_PERIODIC */
   if (wv_uni(st->plc_duration) >= 40 || start != 0 || wv_uni(st->skip_plc)) curr_frame_type = 2;      /* FRAME_PLC_NOISE */
   /* the head of the new frame = last frame's overlap tail */
   for (int c = 0; c < C; c++) {
      FOR_LANES(i, overlap) L->BC.syn[c][i] = gs->overlap_mem[c * overlap + i];
      FOR_LANES(i, N) L->BC.syn[c][overlap + i] = 0;
   }
   wv_sync();
   if (curr_frame_type == 2) {
      const int end = wv_uni(st->end), effEnd = imax(start, imin(end, NBE));
      const i32 decay = loss_duration == 0 ? GC(1.5f) : GC(.5f);
      FOR_LANES(w, C * NBE) { int i = w % NBE; if (i >= start && i < end) L->oldBandE[w] = imax(L->backgroundLogE[w], L->oldBandE[w] - decay); }
      { i32 *Xz = L->Xg; FOR_LANES(i, C * N) Xz[i] = 0; }
      wv_sync();
      u32 seed = (u32)wv_uni((i32)st->rng);
      for (int c = 0; c < C; c++) {
         for (int i = start; i < effEnd; i++) {
            const int boffs = N * c + (ct_eBands[i] << LM), blen = (ct_eBands[i + 1] - ct_eBands[i]) << LM;
            wv_sync();
            LANE0 { u32 s = seed; for (int j = 0; j < blen; j++) { s = lcg_rand(s); L->BC.q.xb[j] = shl32((i32)((i32)s >> 20), NORM_SHIFT - 14); } }
            for (int j = 0; j < blen; j++) seed = lcg_rand(seed);
            renormalise_vector_wave(L->BC.q.xb, blen, Q31ONE);
            wv_sync();
            { i32 *Xo = L->Xg + boffs; FOR_LANES(j, blen) Xo[j] = L->BC.q.xb[j]; }
         }
      }
      wv_sync();
      LANE0 st->rng = seed;
      wv_sync();
      /* (the noise bands went through the PVQ-phase staging buffer, which shares its bytes with syn: the head of the new frame is set up again now) */
      for (int c = 0; c < C; c++) {
         FOR_LANES(i, overlap) L->BC.syn[c][i] = gs->overlap_mem[c * overlap + i];
         FOR_LANES(i, N) L->BC.syn[c][overlap + i] = 0;
      }
      wv_sync();
      if (wv_uni(st->prefilter_and_fold)) prefilter_and_fold_wave(L, gs, C);
      /* celt_synthesis(X, out_syn, oldBandE, start, effEnd, C, C, isTransient = 0, LM, silence = 0) */
      for (int c = 0; c < C; c++) {
         denormalise_bands_wave(L->Xg + c * N, L->oldBandE + c * NBE, L->scr, start, effEnd, 1 << LM, 0, oa_dec_downsample(&L->st));
         mdct_backward_wave(L->Xg + c * N, L->BC.syn[c], 3 - LM, 1, L->aux);
      }
      for (int c = 0; c < C; c++) { FOR_LANES(i, N) L->BC.syn[c][i] = saturate(L->BC.syn[c][i], SIG_SAT); }
      wv_sync();
      LANE0 { st->postfilter_period = imax(st->postfilter_period, OA_MIN_PERIOD); st->postfilter_period_old = imax(st->postfilter_period_old, OA_MIN_PERIOD); }
      wv_sync();
      for (int c = 0; c < C; c++) {
         const i32 *hist = gs->hist + c * OA_DEC_HISTORY;
         comb_filter_inplace_wave(L->BC.syn[c], hist, head, 0, st->postfilter_period_old, st->postfilter_period, 120, st->postfilter_gain_old, st->postfilter_gain,
               st->postfilter_tapset_old, st->postfilter_tapset, overlap);
         if (LM != 0)
            comb_filter_inplace_wave(L->BC.syn[c], hist, head, 120, st->postfilter_period, st->postfilter_period, N - 120, st->postfilter_gain, st->postfilter_gain,
                  st->postfilter_tapset, st->postfilter_tapset, overlap);
      }
      wv_sync();
      LANE0 {
         st->postfilter_period_old = st->postfilter_period; st->postfilter_gain_old = st->postfilter_gain; st->postfilter_tapset_old = st->postfilter_tapset;
         st->prefilter_and_fold = 0;
         st->skip_plc = 1;
      }
   } else {
      /* ---- pitch-based PLC ---- */
      WV_LDS i16 *exc_ = (WV_LDS i16 *)L->A.w;               /* [1024 + 24] */
      WV_LDS i16 *exc = exc_ + PLC_LPC_ORDER;
      WV_LDS i16 *tmp16 = exc_ + 1056;                       /* [1024] windowed / scaled copy, FIR output */
      WV_LDS i32 *ac = (WV_LDS i32 *)(exc_ + 2112);          /* [25] + lpc32 [24] + lpc16 [24] */
      WV_LDS i32 *lpc32 = ac + 32, *lpc16 = ac + 64;
      const int first = wv_uni(st->last_frame_type) != 4;
      int pitch_index;
      i16 fade = Q15ONE;
      if (first) {
         /* celt_plc_pitch_search: downsample both channels' 2048-sample history by 2, search lags 100..720 */
         WV_LDS i16 *lp = (WV_LDS i16 *)L->A.w;                                  /* [1024] result */
         WV_LDS i16 *raw = lp + 1024;                                            /* [1024] raw low-pass */
         WV_LDS i16 *x4 = raw + 1024, *y4 = x4 + 336;                            /* [332], [488] */
         WV_LDS i32 *xc = (WV_LDS i32 *)(y4 + 488);                              /* [310] */
         HistSrc h0, h1;
         h0.hist = gs->hist; h0.head = head; h1.hist = gs->hist + OA_DEC_HISTORY; h1.head = head;
         pitch_downsample_src(raw, lp, h0, h1, OA_DEC_HISTORY >> 1, C);
         pitch_index = pitch_search_bufs(lp + (PLC_PITCH_LAG_MAX >> 1), lp, x4, y4, xc, sh->r, OA_DEC_HISTORY - PLC_PITCH_LAG_MAX, PLC_PITCH_LAG_MAX - PLC_PITCH_LAG_MIN);
         pitch_index = PLC_PITCH_LAG_MAX - pitch_index;
         wv_sync();
         LANE0 st->last_pitch_index = pitch_index;
      } else { pitch_index = wv_uni(st->last_pitch_index); fade = QC16(.8f, 15); }
      pitch_index = wv_uni(pitch_index);
      const int exc_length = imin(2 * pitch_index, OA_MAX_PERIOD);
      for (int c = 0; c < C; c++) {
         HistSrc h; h.hist = gs->hist + c * OA_DEC_HISTORY; h.head = head;
         wv_sync();
         FOR_LANES(i, OA_MAX_PERIOD + PLC_LPC_ORDER) exc_[i] = sround16(src_at(h, OA_DEC_HISTORY - OA_MAX_PERIOD - PLC_LPC_ORDER + i), SIG_SHIFT);
         wv_sync();
         if (first) {
            /* _celt_autocorr(exc, ac, window, overlap, 24, 1024) */
            const int n = OA_MAX_PERIOD, lag = PLC_LPC_ORDER, fastN = n - lag;
            FOR_LANES(i, n) {
               i16 v = exc[i];
               if (i < overlap) v = (i16)mult16_16_q15(v, ct_window[i]);
               else if (i >= n - overlap) v = (i16)mult16_16_q15(v, ct_window[n - 1 - i]);
               tmp16[i] = v;
            }
            wv_sync();
            const int ac0_shift = celt_ilog2(n + (n >> 4));
            i32 a0 = 0;
            FOR_LANES(i, n) a0 += mult16_16(tmp16[i], tmp16[i]) >> ac0_shift;
            i32 ac0 = add32(1 + (n << 7), wv_sum(a0));
            ac0 += ac0 >> 7;
            int shf = celt_ilog2(ac0) - 30 + ac0_shift + 1;
            shf = shf / 2;
            if (shf > 0) { FOR_LANES(i, n) tmp16[i] = (i16)pshr32(tmp16[i], shf); wv_sync(); }
            else shf = 0;
            for (int k = 0; k <= lag; k++) {
               i32 s = 0;
               FOR_LANES(i, fastN) s = mac16_16(s, tmp16[i], tmp16[i + k]);
               for (int i = k + fastN + lane; i < n; i += WV_WIDTH) s = mac16_16(s, tmp16[i], tmp16[i - k]);
               s = wv_sum(s);
               LANE0 ac[k] = s;
            }
            wv_sync();
            LANE0 {
               int sh2 = 2 * shf;
               if (sh2 <= 0) ac[0] += shl32(1, -sh2);
               if (ac[0] < 268435456) { int s2 = 29 - ec_ilog(ac[0]); for (int i = 0; i <= lag; i++) ac[i] = shl32(ac[i], s2); }
               else if (ac[0] >= 536870912) { int s2 = 1; if (ac[0] >= 1073741824) s2++; for (int i = 0; i <= lag; i++) ac[i] = ac[i] >> s2; }
               ac[0] += ac[0] >> 13;
               for (int i = 1; i <= lag; i++) ac[i] -= mult16_32_q15(2 * i * i, ac[i]);
               celt_lpc24_l0(lpc32, lpc16, ac);
               while (1) {                                /* bandwidth expansion until the IIR cannot overflow */
                  i16 t = Q15ONE;
                  i32 sum = QC16(1., SIG_SHIFT);
                  for (int i = 0; i < lag; i++) sum += iabs(lpc16[i]);
                  if (sum < 65535) break;
                  for (int i = 0; i < lag; i++) { t = (i16)mult16_16_q15(QC16(.99f, 15), t); lpc16[i] = (i16)mult16_16_q15(lpc16[i], t); }
               }
               for (int i = 0; i < lag; i++) gs->plc_lpc[c * PLC_LPC_ORDER + i] = lpc16[i];
            }
         } else {
            wv_sync();
            FOR_LANES(i, PLC_LPC_ORDER) lpc16[i] = gs->plc_lpc[c * PLC_LPC_ORDER + i];
         }
         wv_sync();
         /* celt_fir: excitation for the last exc_length samples */
         FOR_LANES(i, exc_length) {
            const WV_LDS i16 *x = exc + OA_MAX_PERIOD - exc_length;
            i32 sum = shl32((i32)x[i], SIG_SHIFT);
            for (int j = 0; j < PLC_LPC_ORDER; j++) sum = mac16_16(sum, lpc16[j], x[i - 1 - j]);
            tmp16[i] = sround16(sum, SIG_SHIFT);
         }
         wv_sync();
         FOR_LANES(i, exc_length) exc[OA_MAX_PERIOD - exc_length + i] = tmp16[i];
         wv_sync();
         i16 decay;
         {
            i32 mx = 0;
            FOR_LANES(i, exc_length) mx = imax(mx, iabs((i32)exc[OA_MAX_PERIOD - exc_length + i]));
            mx = wv_max(mx);
            const int shift = imax(0, 2 * celt_zlog2(mx) - 20), decay_length = exc_length >> 1;
            i32 e1 = 0, e2 = 0;
            FOR_LANES(i, decay_length) {
               i16 e = exc[OA_MAX_PERIOD - decay_length + i];
               e1 += mult16_16(e, e) >> shift;
               e = exc[OA_MAX_PERIOD - 2 * decay_length + i];
               e2 += mult16_16(e, e) >> shift;
            }
            i32 E1 = add32(1, wv_sum(e1)), E2 = add32(1, wv_sum(e2));
            E1 = imin(E1, E2);
            decay = (i16)fx_sqrt(fx_frac_div32(E1 >> 1, E2));
         }
         const int extrapolation_offset = OA_MAX_PERIOD - pitch_index, extrapolation_len = N + overlap;
         i32 s1 = 0;
         FOR_LANES(i, extrapolation_len) {
            const int k = (int)((u32)i / (u32)pitch_index), j = i - k * pitch_index;
            i16 att = (i16)mult16_16_q15(fade, decay);
            for (int t = 0; t < k; t++) att = (i16)mult16_16_q15(att, decay);
            L->BC.syn[c][i] = shl32((i32)(i16)mult16_16_q15(att, exc[extrapolation_offset + j]), SIG_SHIFT);
            i16 t16 = sround16(src_at(h, OA_DEC_HISTORY - OA_MAX_PERIOD + extrapolation_offset + j), SIG_SHIFT);
            s1 += mult16_16(t16, t16) >> 11;
         }
         const i32 S1 = wv_sum(s1);
         wv_sync();
         {
            i32 hist_init = lane < PLC_LPC_ORDER ? (i32)sround16(src_at(h, OA_DEC_HISTORY - 1 - lane), SIG_SHIFT) : 0;
            celt_iir24_wave(L->BC.syn[c], lpc16, extrapolation_len, hist_init);
            FOR_LANES(i, extrapolation_len) L->BC.syn[c][i] = saturate(L->BC.syn[c][i], SIG_SAT);
            wv_sync();
         }
         {
            i32 s2 = 0;
            FOR_LANES(i, extrapolation_len) { i16 t16 = sround16(L->BC.syn[c][i], SIG_SHIFT); s2 += mult16_16(t16, t16) >> 11; }
            const i32 S2 = wv_sum(s2);
            if (!(S1 > (S2 >> 2))) { FOR_LANES(i, extrapolation_len) L->BC.syn[c][i] = 0; }
            else if (S1 < S2) {
               const i16 ratio = (i16)fx_sqrt(fx_frac_div32((S1 >> 1) + 1, S2 + 1));
               FOR_LANES(i, extrapolation_len) {
                  i16 g = ratio;
                  if (i < overlap) g = (i16)(Q15ONE - mult16_16_q15(ct_window[i], Q15ONE - ratio));
                  L->BC.syn[c][i] = mult16_32_q15(g, L->BC.syn[c][i]);
               }
            }
            wv_sync();
         }
      }
      LANE0 st->prefilter_and_fold = 1;
   }
   wv_sync();
   LANE0 {
      st->loss_duration = imin(10000, loss_duration + (1 << LM));
      st->plc_duration = imin(10000, st->plc_duration + (1 << LM));
      st->last_frame_type = curr_frame_type;
   }
   wv_sync();
}
#endif

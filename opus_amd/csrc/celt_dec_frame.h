/* celt_dec_frame.h — CELT frame decoder + Opus packet layer for CELT-only packets, one wavefront per stream.
 * Reference: src/opus.c:203/:224 (packet parse), src/opus_decoder.c:271/:716 (opus_decode_frame / _native, CELT-only branch),
 * celt/celt_decoder.c:1104 (celt_decode_with_ec), :413 (celt_synthesis), :318 (deemphasis), celt/mdct.c:268
 * (clt_mdct_backward), celt/celt.c:238 (comb_filter, in place = recursive).  The Opus layer below (oa_decode_frame_wave, oa_conceal_wave, oa_decode_packet)
 * covers every packet mode (SILK and hybrid through silk_dec*.h), mode transitions with redundancy frames, concealment / DTX / in-band FEC, and every API rate. */
#ifndef OPUS_AMD_CELT_DEC_FRAME_H
#define OPUS_AMD_CELT_DEC_FRAME_H

/* API (output) rate of a decoder record and the CELT down-sampling factor it implies (celt_decoder.c:235 resampling_factor) */
WV_DEV int oa_dec_fs(const WV_LDS OaDecScalars *st) { const int f = wv_uni(st->Fs); return f ? f : 48000; }
WV_DEV int oa_dec_downsample(const WV_LDS OaDecScalars *st) { return 48000 / oa_dec_fs(st); }
#define OA_ERR_BAD_ARG (-1)
#define OA_ERR_BUFFER_TOO_SMALL (-2)
#define OA_ERR_INTERNAL (-3)
#define OA_ERR_INVALID_PACKET (-4)
#define OA_ERR_UNIMPLEMENTED (-5)
#include "silk_nsq.h"          /* SILK fixed-point primitives */
#include "silk_resampler.h"
#include "silk_dec_api.h"      /* SILK decoder (lane-0 serial), shares this kernel's wave, range decoder and LDS */
static_assert(offsetof(DecLds, A) == offsetof(DecLds, BC) + sizeof(((DecLds *)0)->BC) && sizeof(SilkLdsAll) <= sizeof(((DecLds *)0)->A) + sizeof(((DecLds *)0)->BC), "SILK scratch + staged state must fit the CELT decoder's phase regions");
static_assert(OA_SILK_HOT_BYTES % 4 == 0 && sizeof(OaSilkChannel) % 4 == 0, "SILK state is copied as dwords");

/* ---- inverse MDCT of one block (mdct.c:268): in = N2 bins at `stride` (LDS), out = N2 + overlap samples, TDAC into out[0..overlap) ---- */
WV_DEVN void mdct_backward_wave(const i32 *in /* the wave's HBM scratch */, WV_LDS i32 *out, int shift, int stride, WV_LDS int *aux)
{
   shift = wv_uni(shift); stride = wv_uni(stride);
   const int N = 1920 >> shift, N2 = N >> 1, N4 = N >> 2, overlap = OA_OVERLAP;
   const int trig_off = shift == 0 ? 0 : (shift == 1 ? 960 : (shift == 2 ? 1440 : 1680));
   const int16_t *trig = ct_mdct_trig + trig_off;
   const int16_t *bitrev = ct_fft_bitrev + ct_fft_bitrev_off[shift];
   const int lane = wv_lane();
   int pre_shift, post_shift, fft_shift;
   {
      i32 mx = 0, sm = 0;
      FOR_LANES(i, N2) { i32 v = in[i * stride]; mx = imax(mx, iabs(v)); sm = add32(sm, iabs(v >> 11)); }
      i32 maxval = wv_max(mx), sumval = add32(N2, wv_sum(sm));
      pre_shift = imax(0, 29 - celt_zlog2(1 + maxval));
      post_shift = imax(0, 19 - celt_ilog2(iabs(sumval)));
      post_shift = imin(post_shift, pre_shift);
      fft_shift = pre_shift - post_shift;
   }
   WV_LDS i32 *yp = out + (overlap >> 1);
   wv_sync();
   FOR_LANES(i, N4) {
      int rev = bitrev[i];
      i32 x1 = shl32(in[2 * i * stride], pre_shift), x2 = shl32(in[stride * (N2 - 1 - 2 * i)], pre_shift);
      i32 yr = add32(SMUL(x2, trig[i]), SMUL(x1, trig[N4 + i]));
      i32 yi = sub32(SMUL(x1, trig[i]), SMUL(x2, trig[N4 + i]));
      yp[2 * rev + 1] = yr; yp[2 * rev] = yi;
   }
   if (lane == 0) aux[8] = fft_shift;            /* remaining[0]: down-shift budget of the single block */
   wv_sync();
   fft_forward(yp, shift, 1, aux + 8);
   const int left = aux[8];
   {  /* post-rotation in place: pair (i, N4-1-i); every lane reads its four words before writing them */
      const int half = (N4 + 1) >> 1;
      for (int i0 = 0; i0 < half; i0 += WV_WIDTH) {
         int i = i0 + lane;
         i32 a0 = 0, a1 = 0, b0 = 0, b1 = 0;
         if (i < half) {
            WV_LDS i32 *yp0 = yp + 2 * i, *yp1 = yp + N2 - 2 - 2 * i;
            i32 re = fft_shift_val(yp0[1], left), im = fft_shift_val(yp0[0], left);
            int t0 = trig[i], t1 = trig[N4 + i];
            a0 = pshr32(add32(SMUL(re, t0), SMUL(im, t1)), post_shift);          /* -> yp0[0] */
            b1 = pshr32(sub32(SMUL(re, t1), SMUL(im, t0)), post_shift);          /* -> yp1[1] */
            re = fft_shift_val(yp1[1], left); im = fft_shift_val(yp1[0], left);
            t0 = trig[N4 - i - 1]; t1 = trig[N2 - i - 1];
            b0 = pshr32(add32(SMUL(re, t0), SMUL(im, t1)), post_shift);          /* -> yp1[0] */
            a1 = pshr32(sub32(SMUL(re, t1), SMUL(im, t0)), post_shift);          /* -> yp0[1] */
         }
         wv_sync();
         if (i < half) {
            WV_LDS i32 *yp0 = yp + 2 * i, *yp1 = yp + N2 - 2 - 2 * i;
            yp0[0] = a0; yp1[1] = b1; yp1[0] = b0; yp0[1] = a1;
         }
         wv_sync();
      }
   }
   FOR_LANES(i, overlap / 2) {     /* mirror on both sides for TDAC */
      i32 x1 = out[overlap - 1 - i], x2 = out[i];
      int w1 = ct_window[i], w2 = ct_window[overlap - 1 - i];
      out[i] = sub32(SMUL(x2, w2), SMUL(x1, w1));
      out[overlap - 1 - i] = add32(SMUL(x2, w1), SMUL(x1, w2));
   }
   wv_sync();
}

/* the synthesis signal of channel c, frame-relative index j: j >= 0 -> this frame (LDS), j < 0 -> history ring (HBM) */
struct SynSrc { const WV_LDS i32 *cur; const i32 *hist; int head; };
WV_DEV i32 syn_at(const SynSrc &s, int j) { return j >= 0 ? s.cur[j] : s.hist[(s.head + j) & (OA_DEC_HISTORY - 1)]; }

/* comb_filter (celt.c:238) in place on y[off .. off+N): the decoder's post-filter is recursive (taps read already-filtered
 * samples), so it is run in chunks no longer than min(T0,T1)-2 samples: inside a chunk every tap points before the chunk. */
WV_DEVN void comb_filter_inplace_wave(WV_LDS i32 *cur, const i32 *hist, int head, int off, int T0, int T1, int N, i32 g0_, i32 g1_, int tapset0, int tapset1, int overlap)
{
   T0 = wv_uni(T0); T1 = wv_uni(T1); N = wv_uni(N); off = wv_uni(off); tapset0 = wv_uni(tapset0); tapset1 = wv_uni(tapset1); overlap = wv_uni(overlap);
   const i16 g0 = (i16)wv_uni(g0_), g1 = (i16)wv_uni(g1_);
   const i16 gains[3][3] = {
      {QC16(0.3066406250f, 15), QC16(0.2170410156f, 15), QC16(0.1296386719f, 15)},
      {QC16(0.4638671875f, 15), QC16(0.2680664062f, 15), QC16(0.f, 15)},
      {QC16(0.7998046875f, 15), QC16(0.1000976562f, 15), QC16(0.f, 15)}};
   if (g0 == 0 && g1 == 0) return;
   T0 = imax(T0, OA_MIN_PERIOD);
   T1 = imax(T1, OA_MIN_PERIOD);
   const i16 g00 = (i16)mult_coef_taps(g0, gains[tapset0][0]), g01 = (i16)mult_coef_taps(g0, gains[tapset0][1]), g02 = (i16)mult_coef_taps(g0, gains[tapset0][2]);
   const i16 g10 = (i16)mult_coef_taps(g1, gains[tapset1][0]), g11 = (i16)mult_coef_taps(g1, gains[tapset1][1]), g12 = (i16)mult_coef_taps(g1, gains[tapset1][2]);
   if (g0 == g1 && T0 == T1 && tapset0 == tapset1) overlap = 0;
   SynSrc s; s.cur = cur; s.hist = hist; s.head = head;
#define XA(k) syn_at(s, off + (k))
   const int chunk = imin(WV_WIDTH, imin(T0, T1) - 2);
   const int ov = imin(overlap, N);
   wv_sync();
   for (int c0 = 0; c0 < ov; c0 += chunk) {                  /* cross-fade from (T0, g0, tapset0) to (T1, g1, tapset1) */
      const int i = c0 + wv_lane();
      const bool act = wv_lane() < chunk && i < ov;
      i32 v = 0;
      if (act) {
         i16 f = (i16)mult_coef(ct_window[i], ct_window[i]);
         v = XA(i);
         v = add32(v, mult_coef_32(mult_coef((Q15ONE - f), g00), XA(i - T0)));
         v = add32(v, mult_coef_32(mult_coef((Q15ONE - f), g01), add32(XA(i - T0 + 1), XA(i - T0 - 1))));
         v = add32(v, mult_coef_32(mult_coef((Q15ONE - f), g02), add32(XA(i - T0 + 2), XA(i - T0 - 2))));
         v = add32(v, mult_coef_32(mult_coef(f, g10), XA(i - T1)));
         v = add32(v, mult_coef_32(mult_coef(f, g11), add32(XA(i - T1 + 1), XA(i - T1 - 1))));
         v = add32(v, mult_coef_32(mult_coef(f, g12), add32(XA(i - T1 + 2), XA(i - T1 - 2))));
         v = saturate(sub32(v, 3), SIG_SAT);
      }
      wv_sync();
      if (act) cur[off + i] = v;
      wv_sync();
   }
   if (g1 != 0) {                                            /* constant filter on the rest (comb_filter_const, celt.c:205) */
      for (int c0 = ov; c0 < N; c0 += chunk) {
         const int i = c0 + wv_lane();
         const bool act = wv_lane() < chunk && i < N;
         i32 v = 0;
         if (act) {
            v = add32(add32(add32(XA(i), mult_coef_32(g10, XA(i - T1))), mult_coef_32(g11, add32(XA(i - T1 + 1), XA(i - T1 - 1)))),
                  mult_coef_32(g12, add32(XA(i - T1 + 2), XA(i - T1 - 2))));
            v = saturate(sub32(v, 1), SIG_SAT);
         }
         wv_sync();
         if (act) cur[off + i] = v;
         wv_sync();
      }
   }
#undef XA
}

/* common tail of a decoded or concealed frame: de-emphasis -> int16 PCM, history ring += N samples, overlap tail kept */
WV_DEVN void celt_emit_frame_wave(WV_LDS DecLds *L, OaDecStream *gs, int N, int CC, i16 *pcm_out, int accum = 0)
{
   WV_LDS OaDecScalars *st = &L->st;
   const int overlap = OA_OVERLAP, lane = wv_lane();
   N = wv_uni(N); CC = wv_uni(CC);
   const int ds = oa_dec_downsample(st), Nd = N / ds;               /* API rates below 48 kHz keep every ds-th de-emphasised sample (celt_decoder.c:361-404) */
   wv_sync();
   /* ---- history ring <- the N post-filtered samples; overlap tail <- syn[N .. N+overlap) (before the de-emphasis below reuses syn's words) ---- */
   {
      const int head = wv_uni(st->hist_head);
      for (int c = 0; c < CC; c++) {
         FOR_LANES(i, N) gs->hist[c * OA_DEC_HISTORY + ((head + i) & (OA_DEC_HISTORY - 1))] = L->BC.syn[c][i];
         FOR_LANES(i, overlap) gs->overlap_mem[c * overlap + i] = L->BC.syn[c][N + i];
      }
   }
   wv_sync();
   /* ---- deemphasis (celt_decoder.c:318): one-pole IIR with rounding -> one lane per channel; the int16 result goes back into the sample's own word of syn ---- */
   if (lane < CC) {
      i32 m = st->preemph_memD[lane];
      WV_LDS i32 *x = L->BC.syn[lane];
      for (int j0 = 0; j0 < N; j0 += 8) {                /* eight reads in flight per trip; only the (add, saturate, multiply) chain is serial */
         i32 t[8];
#pragma unroll
         for (int k = 0; k < 8; k++) t[k] = x[j0 + k];
#pragma unroll
         for (int k = 0; k < 8; k++) { t[k] = saturate(t[k] + m, SIG_SAT); m = mult16_32_q15(27853, t[k]); }
#pragma unroll
         for (int k = 0; k < 8; k++) x[j0 + k] = sig2word16(t[k]);
      }
      st->preemph_memD[lane] = m;
   }
   wv_sync();
   accum = wv_uni(accum);
   /* API rates below 48 kHz keep every ds-th de-emphasised sample (celt_decoder.c:361-404); interleave on the way out */
   FOR_LANES(it, Nd * CC) {
      const int i = it / CC, c = it - i * CC;
      const i32 v = L->BC.syn[c][i * ds];
      if (accum) { const i32 w = (i32)pcm_out[it] + v; pcm_out[it] = (i16)(w > 32767 ? 32767 : w < -32768 ? -32768 : w); }      /* ADD_RES, celt/arch.h:172 */
      else pcm_out[it] = (i16)v;
   }
   wv_sync();
   LANE0 st->hist_head = (st->hist_head + N) & (OA_DEC_HISTORY - 1);
   wv_sync();
}

#include "celt_dec_plc.h"

/* the frame from behind its bands to the end (celt_decoder.c:1476-1640): anti-collapse, energy finalisation, synthesis, post-filter, state update, de-emphasis.  Called by
 * celt_decode_frame_wave where the bands were decoded in place, and by the back kernel of the decoder's kernel pipeline (oa_celt_dback_kernel) on the reloaded LDS image:
 * everything it needs of the frame is in L->sh / L->st. */
/* hdr (FAST only; NULL, or two words of the stream's continuation record): the de-emphasis and the PCM store are left to oa_celt_deemph_kernel (one lane per stream): every
 * channel's post-filtered samples go where its spectrum was (the IMDCT has consumed it), hdr[0] = N, hdr[1] = accum | 2 * (channel 0 sits in the second half) */
template <bool FAST> WV_DEV int celt_decode_frame_tail(WV_LDS DecLds *L, OaDecStream *gs, int len, int frame_size, i16 *pcm_out, int accum, i32 *hdr = 0)
{
   WV_LDS DecShared *sh = &L->sh;
   WV_LDS OaDecScalars *st = &L->st;
   const int overlap = OA_OVERLAP;
   const int lane = wv_lane();
   len = wv_uni(len); frame_size = wv_uni(frame_size);
   const int downsample = oa_dec_downsample(st);
   const int LM = wv_uni(sh->LM), M = 1 << LM, N = M * 120;
   const int CC = wv_uni(sh->CC), C = wv_uni(sh->C), start = wv_uni(sh->start), end = wv_uni(sh->end), effEnd = wv_uni(sh->effEnd);
   P4_TIC();
   LANE0 {
      EC_BEGIN;
      int anti_collapse_on = 0;
      if (sh->anti_collapse_rsv > 0) anti_collapse_on = k_ec_dec_bits(EC_PASS, 1);
      sh->r[6] = (i32)energy_finalise_read_l0(start, end, L->fine_quant, L->fine_priority, len * 8 - k_ec_tell(EC_PASS), L->scr, EC_PASS, C);
      sh->anti_collapse_on = anti_collapse_on;
      EC_END;
   }
   energy_finalise_apply_dec_wave((u32)wv_uni(sh->r[6]), L->scr, L->oldBandE, L->fine_quant, C);
   if (sh->anti_collapse_on) anti_collapse_wave(L, LM, C, N, start, end);
   const int silence = wv_uni(sh->silence), isTransient = wv_uni(sh->isTransient);
   if (silence) { wv_sync(); FOR_LANES(i, C * NBE) L->oldBandE[i] = -GC(28.f); wv_sync(); }
   K_DUMP("dec_X", L->Xg, C * N * 4); K_DUMP("dec_oldBandE", L->oldBandE, 2 * NBE * 4);

   if (FAST) {
      /* ---- the fast kernel's tail: celt_synthesis (celt_decoder.c:413), the post-filter (:1536), the history / overlap update and the de-emphasis (:318) one CHANNEL at a time in
       * syn[0] -- half the synthesis memory, which is what lets 16 waves share a CU; the price is the de-emphasis recursion running once per channel instead of on two lanes at once ---- */
      int B, NB, shift;
      if (isTransient) { B = M; NB = 120; shift = 3; }
      else { B = 1; NB = 120 << LM; shift = 3 - LM; }
      const int head = wv_uni(st->hist_head), ds = oa_dec_downsample(st), Nd = N / ds;
      WV_LDS i32 *const syn = L->BC.syn[0];
      i32 *freq = L->Xg;
      accum = wv_uni(accum);
      LANE0 { st->postfilter_period = imax(st->postfilter_period, OA_MIN_PERIOD); st->postfilter_period_old = imax(st->postfilter_period_old, OA_MIN_PERIOD); }
      wv_sync();
      P4_TOC(13);
      if (CC == 2 && C == 1) {
         denormalise_bands_wave(freq, L->oldBandE, L->scr, start, effEnd, M, silence, downsample);
         FOR_LANES(i, N) freq[N + i] = freq[i];        /* the IMDCT consumes its input: keep a copy for the second channel */
         wv_sync();
      } else if (CC == 1 && C == 2) {
         denormalise_bands_wave(freq, L->oldBandE, L->scr, start, effEnd, M, silence, downsample);
         denormalise_bands_wave(freq + N, L->oldBandE + NBE, L->scr, start, effEnd, M, silence, downsample);
         FOR_LANES(i, N) freq[i] = add32(half32(freq[i]), half32(freq[N + i]));
         wv_sync();
      }
      for (int c = 0; c < CC; c++) {
         i32 *fc;
         if (CC == 2 && C == 1) fc = c == 0 ? freq + N : freq;
         else if (CC == 1 && C == 2) fc = freq;
         else { fc = freq + c * N; denormalise_bands_wave(fc, L->oldBandE + c * NBE, L->scr, start, effEnd, M, silence, downsample); }
         wv_sync();
         FOR_LANES(i, overlap) syn[i] = gs->overlap_mem[c * overlap + i];
         FOR_LANES(i, N) syn[overlap + i] = 0;
         wv_sync();
         P4_TOC(14);
         for (int b = 0; b < B; b++) mdct_backward_wave(fc + b, syn + NB * b, shift, B, L->aux);
         FOR_LANES(i, N) syn[i] = saturate(syn[i], SIG_SAT);
         wv_sync();
         P4_TOC(15);
         K_DUMP("dec_syn", syn, N * 4);
         const i32 *hist = gs->hist + c * OA_DEC_HISTORY;
         comb_filter_inplace_wave(syn, hist, head, 0, st->postfilter_period_old, st->postfilter_period, 120, st->postfilter_gain_old, st->postfilter_gain,
               st->postfilter_tapset_old, st->postfilter_tapset, overlap);
         if (LM != 0)
            comb_filter_inplace_wave(syn, hist, head, 120, st->postfilter_period, sh->postfilter_pitch, N - 120, st->postfilter_gain, sh->postfilter_gain,
                  st->postfilter_tapset, sh->postfilter_tapset, overlap);
         wv_sync();
         FOR_LANES(i, N) gs->hist[c * OA_DEC_HISTORY + ((head + i) & (OA_DEC_HISTORY - 1))] = syn[i];
         FOR_LANES(i, overlap) gs->overlap_mem[c * overlap + i] = syn[N + i];
         wv_sync();
         P4_TOC(16);
         if (hdr) {
            FOR_LANES(i, N) fc[i] = syn[i];
            wv_sync();
            continue;
         }
         if (lane == 0) {
            i32 m = st->preemph_memD[c];
            for (int j0 = 0; j0 < N; j0 += 8) {
               i32 t[8];
#pragma unroll
               for (int k = 0; k < 8; k++) t[k] = syn[j0 + k];
#pragma unroll
               for (int k = 0; k < 8; k++) { t[k] = saturate(t[k] + m, SIG_SAT); m = mult16_32_q15(27853, t[k]); }
#pragma unroll
               for (int k = 0; k < 8; k++) syn[j0 + k] = sig2word16(t[k]);
            }
            st->preemph_memD[c] = m;
         }
         wv_sync();
         P4_TOC(17);
         FOR_LANES(i, Nd) {
            const i32 v = syn[i * ds];
            const int it = i * CC + c;
            if (accum) { const i32 w = (i32)pcm_out[it] + v; pcm_out[it] = (i16)(w > 32767 ? 32767 : w < -32768 ? -32768 : w); }
            else pcm_out[it] = (i16)v;
         }
         wv_sync();
         P4_TOC(18);
      }
      if (hdr) { LANE0 { hdr[0] = N; hdr[1] = (accum ? 1 : 0) | (CC == 2 && C == 1 ? 2 : 0); } }
      LANE0 {
         st->hist_head = (st->hist_head + N) & (OA_DEC_HISTORY - 1);
         st->postfilter_period_old = st->postfilter_period; st->postfilter_gain_old = st->postfilter_gain; st->postfilter_tapset_old = st->postfilter_tapset;
         st->postfilter_period = sh->postfilter_pitch; st->postfilter_gain = sh->postfilter_gain; st->postfilter_tapset = sh->postfilter_tapset;
         if (LM != 0) { st->postfilter_period_old = st->postfilter_period; st->postfilter_gain_old = st->postfilter_gain; st->postfilter_tapset_old = st->postfilter_tapset; }
      }
      wv_sync();
   } else {
   /* ---- celt_synthesis (celt_decoder.c:413): denormalise in place, IMDCT per block into syn[c] (head = last frame's overlap tail) ---- */
   {
      int B, NB, shift;
      if (isTransient) { B = M; NB = 120; shift = 3; }
      else { B = 1; NB = 120 << LM; shift = 3 - LM; }
      for (int c = 0; c < CC; c++) {
         FOR_LANES(i, overlap) L->BC.syn[c][i] = gs->overlap_mem[c * overlap + i];
         FOR_LANES(i, N) L->BC.syn[c][overlap + i] = 0;
      }
      wv_sync();
      if (!FAST && wv_uni(st->prefilter_and_fold)) prefilter_and_fold_wave(L, gs, CC);
      i32 *freq = L->Xg;
      if (CC == 2 && C == 1) {
         denormalise_bands_wave(freq, L->oldBandE, L->scr, start, effEnd, M, silence, downsample);
         FOR_LANES(i, N) freq[N + i] = freq[i];        /* the IMDCT consumes its input: keep a copy for the second channel */
         wv_sync();
         for (int b = 0; b < B; b++) mdct_backward_wave(freq + N + b, L->BC.syn[0] + NB * b, shift, B, L->aux);
         for (int b = 0; b < B; b++) mdct_backward_wave(freq + b, L->BC.syn[1] + NB * b, shift, B, L->aux);
      } else if (CC == 1 && C == 2) {
         denormalise_bands_wave(freq, L->oldBandE, L->scr, start, effEnd, M, silence, downsample);
         denormalise_bands_wave(freq + N, L->oldBandE + NBE, L->scr, start, effEnd, M, silence, downsample);
         FOR_LANES(i, N) freq[i] = add32(half32(freq[i]), half32(freq[N + i]));
         wv_sync();
         for (int b = 0; b < B; b++) mdct_backward_wave(freq + b, L->BC.syn[0] + NB * b, shift, B, L->aux);
      } else {
         for (int c = 0; c < CC; c++) {
            denormalise_bands_wave(freq + c * N, L->oldBandE + c * NBE, L->scr, start, effEnd, M, silence, downsample);
            for (int b = 0; b < B; b++) mdct_backward_wave(freq + c * N + b, L->BC.syn[c] + NB * b, shift, B, L->aux);
         }
      }
      for (int c = 0; c < CC; c++) { FOR_LANES(i, N) L->BC.syn[c][i] = saturate(L->BC.syn[c][i], SIG_SAT); }
      wv_sync();
   }
   for (int c = 0; c < CC; c++) K_DUMP("dec_syn", L->BC.syn[c], N * 4);

   /* ---- pitch post-filter (celt_decoder.c:1536-1553), in place and recursive ---- */
   {
      const int head = wv_uni(st->hist_head);
      LANE0 { st->postfilter_period = imax(st->postfilter_period, OA_MIN_PERIOD); st->postfilter_period_old = imax(st->postfilter_period_old, OA_MIN_PERIOD); }
      wv_sync();
      for (int c = 0; c < CC; c++) {
         const i32 *hist = gs->hist + c * OA_DEC_HISTORY;
         comb_filter_inplace_wave(L->BC.syn[c], hist, head, 0, st->postfilter_period_old, st->postfilter_period, 120, st->postfilter_gain_old, st->postfilter_gain,
               st->postfilter_tapset_old, st->postfilter_tapset, overlap);
         if (LM != 0)
            comb_filter_inplace_wave(L->BC.syn[c], hist, head, 120, st->postfilter_period, sh->postfilter_pitch, N - 120, st->postfilter_gain, sh->postfilter_gain,
                  st->postfilter_tapset, sh->postfilter_tapset, overlap);
      }
      wv_sync();
   }
   /* ---- state update (celt_decoder.c:1555-1607) ---- */
   LANE0 {
      st->postfilter_period_old = st->postfilter_period; st->postfilter_gain_old = st->postfilter_gain; st->postfilter_tapset_old = st->postfilter_tapset;
      st->postfilter_period = sh->postfilter_pitch; st->postfilter_gain = sh->postfilter_gain; st->postfilter_tapset = sh->postfilter_tapset;
      if (LM != 0) { st->postfilter_period_old = st->postfilter_period; st->postfilter_gain_old = st->postfilter_gain; st->postfilter_tapset_old = st->postfilter_tapset; }
   }
   wv_sync();
   }
   if (C == 1) { FOR_LANES(i, NBE) L->oldBandE[NBE + i] = L->oldBandE[i]; wv_sync(); }
   {
      const i32 max_background_increase = imin(160, wv_uni(st->loss_duration) + M) * GC(0.001f);
      FOR_LANES(i, 2 * NBE) {
         const int bi = i % NBE;
         i32 ob = L->oldBandE[i], l1 = L->oldLogE[i], l2 = L->oldLogE2[i];
         if (!isTransient) { l2 = l1; l1 = ob; } else l1 = imin(l1, ob);
         L->backgroundLogE[i] = imin(L->backgroundLogE[i] + max_background_increase, ob);
         if (bi < start || bi >= end) { ob = 0; l1 = l2 = -GC(28.f); }
         L->oldBandE[i] = ob; L->oldLogE[i] = l1; L->oldLogE2[i] = l2;
      }
   }
   wv_sync();
   if (!FAST) celt_emit_frame_wave(L, gs, N, CC, pcm_out, accum);
   int ret = frame_size;
   LANE0 {
      st->rng = L->ec.rng;
      st->loss_duration = 0; st->plc_duration = 0; st->last_frame_type = 1; st->prefilter_and_fold = 0;
      i32 used = L->ec.nbits_total - ec_ilog(L->ec.rng);
      sh->r[1] = used > 8 * len ? OA_ERR_INTERNAL : frame_size;
      if (L->ec.error) st->error = 1;
   }
   wv_sync();
   ret = wv_uni(sh->r[1]);
   return ret;
}

/* ---- one CELT frame (celt_decoder.c:1104).  The frame's bytes are at L->packet + 1.  Returns the frame size or < 0. ---- */
/* ec_cont: continue the range decoder parked in L->ec_silk (hybrid frames) instead of starting one; accum: add onto pcm_out (celt_decoder.c:1104 `dec`, `accum`) */
/* FAST: the instance of the CELT-only fast kernel (opus_amd.hip: oa_decode_fast_kernel): the concealment and the post-concealment fold are compiled out -- the kernel only takes
 * packets that reach neither */
template <bool FAST = false> WV_DEVN int celt_decode_frame_wave(WV_LDS DecLds *L, OaDecStream *gs, int len, int frame_size, i16 *pcm_out, int ec_cont = 0, int accum = 0, int cut = 0 /* FAST: stop in front of the bands of a 10 / 20 ms frame and return OA_DEC_CUT (the kernel pipeline, celt_dec_pvq4.h) */)
{
   WV_LDS DecShared *sh = &L->sh;
   WV_LDS OaDecScalars *st = &L->st;
   const int overlap = OA_OVERLAP;
   const int lane = wv_lane();
   len = wv_uni(len); frame_size = wv_uni(frame_size);
   const int downsample = oa_dec_downsample(st);                        /* frame_size is in API-rate samples; the codec runs at 48 kHz (celt_decoder.c:1185) */
   int LM;
   for (LM = 0; LM <= 3; LM++) if (120 << LM == frame_size * downsample) break;
   if (LM > 3) return OA_ERR_BAD_ARG;
   if (len < 0 || len > 1275) return OA_ERR_BAD_ARG;
   const int M = 1 << LM, N = M * 120;
   const int CC = wv_uni(st->channels), C = wv_uni(st->stream_channels), start = wv_uni(st->start), end = wv_uni(st->end);
   if (!FAST && len <= 1) {                                       /* lost / DTX frame: conceal (celt_decoder.c:1306) */
      celt_decode_lost_wave(L, gs, N, LM);
      celt_emit_frame_wave(L, gs, N, CC, pcm_out, accum);
      return frame_size;
   }
   const int effEnd = imin(end, NBE);
   wv_sync();
   P4_TIC();
   LANE0 {
      EcCtx ec_; EcCtx *e = &ec_; WV_LDS u8 *buf = L->packet + 1;
      sh->CC = CC; sh->C = C; sh->LM = LM; sh->M = M; sh->N = N; sh->start = start; sh->end = end; sh->effEnd = effEnd; sh->len = len;
      if (st->loss_duration == 0) st->skip_plc = 0;
      if (ec_cont) { ec_ld(e, &L->ec_silk); e->storage = (u32)len; } else k_ec_dec_init(EC_PASS, len);
      if (C == 1) for (int i = 0; i < NBE; i++) L->oldBandE[i] = imax(L->oldBandE[i], L->oldBandE[NBE + i]);
      i32 total_bits = len * 8, tell = k_ec_tell(EC_PASS);
      int silence;
      if (tell >= total_bits) silence = 1;
      else if (tell == 1) silence = k_ec_dec_bit_logp(EC_PASS, 15);
      else silence = 0;
      if (silence) { tell = len * 8; e->nbits_total += tell - k_ec_tell(EC_PASS); }
      int postfilter_gain = 0, postfilter_pitch = 0, postfilter_tapset = 0;
      if (start == 0 && tell + 16 <= total_bits) {
         if (k_ec_dec_bit_logp(EC_PASS, 1)) {
            int qg, octave = k_ec_dec_uint(EC_PASS, 6);
            postfilter_pitch = (16 << octave) + k_ec_dec_bits(EC_PASS, 4 + octave) - 1;
            qg = k_ec_dec_bits(EC_PASS, 3);
            if (k_ec_tell(EC_PASS) + 2 <= total_bits) postfilter_tapset = k_ec_dec_icdf(EC_PASS, k_tapset_icdf, 2);
            postfilter_gain = (i16)(QC16(.09375f, 15) * (qg + 1));
         }
         tell = k_ec_tell(EC_PASS);
      }
      int isTransient = 0;
      if (LM > 0 && tell + 3 <= total_bits) { isTransient = k_ec_dec_bit_logp(EC_PASS, 3); tell = k_ec_tell(EC_PASS); }
      const int shortBlocks = isTransient ? M : 0;
      const int intra_ener = tell + 3 <= total_bits ? k_ec_dec_bit_logp(EC_PASS, 3) : 0;
      if (!intra_ener && st->loss_duration != 0) {          /* energy prediction safety after a loss (celt_decoder.c:1387) */
         for (int c = 0; c < 2; c++) {
            i32 safety = 0;
            int missing = imin(10, st->loss_duration >> LM);
            if (LM == 0) safety = GC(1.5f);
            else if (LM == 1) safety = GC(.5f);
            for (int i = start; i < end; i++) {
               i32 E0 = L->oldBandE[c * NBE + i], E1 = L->oldLogE[c * NBE + i], E2 = L->oldLogE2[c * NBE + i];
               if (E0 < imax(E1, E2)) {
                  i32 slope = imax(E1 - E0, half32(E2 - E0));
                  slope = imin(slope, GC(2.f));
                  E0 -= imax(0, (1 + missing) * slope);
                  E0 = imax(-GC(20.f), E0);
               } else E0 = imin(imin(E0, E1), E2);
               L->oldBandE[c * NBE + i] = E0 - safety;
            }
         }
      }
      coarse_energy_read_l0(start, end, L->oldBandE, intra_ener, EC_PASS, C, LM, L->scr);
      tf_read_l0(start, end, isTransient, L->tf_res, LM, EC_PASS);
      tell = k_ec_tell(EC_PASS);
      int spread_decision = 2;
      if (tell + 4 <= total_bits) spread_decision = k_ec_dec_icdf(EC_PASS, k_spread_icdf, 5);
      k_init_caps(L->cap, LM, C);
      int dynalloc_logp = 6;
      total_bits <<= BITRES;
      tell = k_ec_tell_frac(EC_PASS);
      for (int i = start; i < end; i++) {
         int width = C * (ct_eBands[i + 1] - ct_eBands[i]) << LM;
         int quanta = imin(width << BITRES, imax(6 << BITRES, width));
         int dynalloc_loop_logp = dynalloc_logp, boost = 0;
         while (tell + (dynalloc_loop_logp << BITRES) < total_bits && boost < L->cap[i]) {
            int flag = k_ec_dec_bit_logp(EC_PASS, dynalloc_loop_logp);
            tell = k_ec_tell_frac(EC_PASS);
            if (!flag) break;
            boost += quanta;
            total_bits -= quanta;
            dynalloc_loop_logp = 1;
         }
         L->offsets[i] = boost;
         if (boost > 0) dynalloc_logp = imax(2, dynalloc_logp - 1);
      }
      const int alloc_trim = tell + (6 << BITRES) <= total_bits ? k_ec_dec_icdf(EC_PASS, k_trim_icdf, 7) : 5;
      i32 bits = (((i32)len * 8) << BITRES) - (i32)k_ec_tell_frac(EC_PASS) - 1;
      const int anti_collapse_rsv = isTransient && LM >= 2 && bits >= ((LM + 2) << BITRES) ? (1 << BITRES) : 0;
      bits -= anti_collapse_rsv;
      sh->intensity = 0; sh->dual_stereo = 0; sh->balance = 0;
      sh->silence = silence; sh->postfilter_pitch = postfilter_pitch; sh->postfilter_gain = postfilter_gain; sh->postfilter_tapset = postfilter_tapset;
      sh->isTransient = isTransient; sh->shortBlocks = shortBlocks; sh->spread = spread_decision;
      sh->anti_collapse_rsv = anti_collapse_rsv; sh->alloc_trim = alloc_trim; sh->r[4] = bits;
      sh->pvq_total_bits = len * (8 << BITRES) - anti_collapse_rsv;
      ec_st(&L->ec, &ec_);
   }
   wv_sync();
   if (FAST) P4_TOC(10);
   {  /* bit allocation: the same wave routine as the encoder, reading the three side-information symbols (celt_alloc.h) */
      const int coded = oa_allocate_bits_wave<false>(&L->ec, L->packet + 1, L->scr, sh->start, sh->end, L->offsets, L->cap, sh->alloc_trim, &sh->intensity, &sh->dual_stereo, sh->r[4], &sh->balance,
            L->pulses, L->fine_quant, L->fine_priority, sh->C, sh->LM, 0, 0, sh->r + 6);
      LANE0 sh->codedBands = coded;
   }
   fine_energy_read_wave(&L->ec, L->packet + 1, L->scr, sh->r + 6, sh->start, sh->end, L->oldBandE, L->fine_quant, sh->C);
   /* X starts at zero (the reference's bands below start / above end are never written) */
   { i32 *Xz = L->Xg; FOR_LANES(i, C * N) Xz[i] = 0; }
   wv_sync();
   if (FAST) P4_TOC(11);
   if (FAST && wv_uni(cut) && LM >= 2) return OA_DEC_CUT;
   dec_quant_all_bands_wave(L, sh->shortBlocks, sh->spread, sh->dual_stereo, sh->intensity, sh->pvq_total_bits, sh->balance, sh->codedBands, st->disable_inv);
   if (FAST) P4_TOC(12);
   return celt_decode_frame_tail<FAST>(L, gs, len, frame_size, pcm_out, accum);
}

/* ---- Opus packet layer: opus_decode_native (opus_decoder.c:716) for CELT-only packets ---- */
WV_DEV int oa_samples_per_frame(int toc, i32 Fs)
{
   int audiosize;
   if (toc & 0x80) { audiosize = ((toc >> 3) & 0x3); audiosize = (Fs << audiosize) / 400; }
   else if ((toc & 0x60) == 0x60) audiosize = (toc & 0x08) ? Fs / 50 : Fs / 100;
   else { audiosize = ((toc >> 3) & 0x3); audiosize = audiosize == 3 ? Fs * 60 / 1000 : (Fs << audiosize) / 100; }
   return audiosize;
}
WV_DEV int oa_parse_size(const u8 *data, i32 len, i32 *size)
{
   if (len < 1) { *size = -1; return -1; }
   else if (data[0] < 252) { *size = data[0]; return 1; }
   else if (len < 2) { *size = -1; return -1; }
   else { *size = 4 * data[1] + data[0]; return 2; }
}
/* opus_packet_parse_impl (opus.c:224, not self-delimited); lane 0.  Fills sh->size[], returns count or < 0; *payload_offset */
template <class SZ> WV_DEV int oa_packet_parse(const u8 *data, i32 len, SZ size, int *payload_offset)
{
   int i, bytes, count, framesize;
   u8 ch, toc;
   i32 last_size, sz;
   const u8 *data0 = data;
   if (len < 0) return OA_ERR_BAD_ARG;
   if (len == 0) return OA_ERR_INVALID_PACKET;
   framesize = oa_samples_per_frame(data[0], 48000);
   toc = *data++;
   len--;
   last_size = len;
   switch (toc & 0x3) {
   case 0: count = 1; break;
   case 1:
      count = 2;
      if (len & 0x1) return OA_ERR_INVALID_PACKET;
      last_size = len / 2;
      size[0] = last_size;
      break;
   case 2:
      count = 2;
      bytes = oa_parse_size(data, len, &sz); size[0] = sz;
      len -= bytes;
      if (sz < 0 || sz > len) return OA_ERR_INVALID_PACKET;
      data += bytes;
      last_size = len - sz;
      break;
   default:
      if (len < 1) return OA_ERR_INVALID_PACKET;
      ch = *data++;
      count = ch & 0x3F;
      if (count <= 0 || framesize * (i32)count > 5760) return OA_ERR_INVALID_PACKET;
      len--;
      if (ch & 0x40) {
         int p;
         do {
            int tmp;
            if (len <= 0) return OA_ERR_INVALID_PACKET;
            p = *data++;
            len--;
            tmp = p == 255 ? 254 : p;
            len -= tmp;
         } while (p == 255);
      }
      if (len < 0) return OA_ERR_INVALID_PACKET;
      if (ch & 0x80) {
         last_size = len;
         for (i = 0; i < count - 1; i++) {
            bytes = oa_parse_size(data, len, &sz); size[i] = sz;
            len -= bytes;
            if (sz < 0 || sz > len) return OA_ERR_INVALID_PACKET;
            data += bytes;
            last_size -= bytes + sz;
         }
         if (last_size < 0) return OA_ERR_INVALID_PACKET;
      } else {
         last_size = len / count;
         if (last_size * count != len) return OA_ERR_INVALID_PACKET;
         for (i = 0; i < count - 1; i++) size[i] = last_size;
      }
      break;
   }
   if (last_size > 1275) return OA_ERR_INVALID_PACKET;
   size[count - 1] = last_size;
   *payload_offset = (int)(data - data0);
   return count;
}

/* opus_decode_frame(data = NULL) (opus_decoder.c:316-366): conceal up to frame_size samples with the last mode; returns samples produced */
WV_DEVN int oa_conceal_wave(WV_LDS DecLds *L, OaDecStream *gs, int frame_size, i16 *pcm_out, int CC)
{
   WV_LDS OaDecScalars *st = &L->st;
   frame_size = wv_uni(frame_size);
   const int Fs = oa_dec_fs(st);
   const int F20 = Fs / 50, F10 = Fs / 100, F5 = Fs / 200, F2_5 = Fs / 400;
   if (frame_size < F2_5) return OA_ERR_BUFFER_TOO_SMALL;
   const int mode = wv_uni(st->prev_redundancy) ? 1002 : wv_uni(st->prev_mode);
   if (mode == 0) {                     /* nothing decoded yet: zeros */
      FOR_LANES(i, frame_size * CC) pcm_out[i] = 0;
      LANE0 st->rangeFinal = 0;
      wv_sync();
      return frame_size;
   }
   int done = 0;
   while (done < frame_size) {
      int audiosize = frame_size - done;
      if (audiosize > F20) audiosize = F20;
      else if (audiosize < F20) {
         if (audiosize > F10) audiosize = F10;
         else if (mode != 1000 && audiosize > F5 && audiosize < F10) audiosize = F5;
      }
      if (mode == 1002 && audiosize != F20 && audiosize != F10 && audiosize != F5 && audiosize != F2_5) return OA_ERR_BAD_ARG;
      i16 *pcm = pcm_out + (size_t)done * CC;
      if (mode != 1002) {
         /* SILK concealment (src/opus_decoder.c:404-497 with data == NULL): the decoder control of the last good frame persists */
         WV_LDS SilkLdsAll *SL = (WV_LDS SilkLdsAll *)&L->BC;
         wv_sync();
         FOR_LANES(i, (int)(OA_SILK_HOT_BYTES / 4)) SL->hot[i] = ((const i32 *)&gs->silk)[i];           /* SILK state -> LDS, coalesced */
         wv_sync();
         {
            WV_LDS OaSilkDec *sdh = (WV_LDS OaSilkDec *)SL->hot;
            WV_LDS u8 *buf = L->packet + 1;
            LANE0 {
               EcCtx ec;
               ec.storage = 0; ec.end_offs = 0; ec.end_window = 0; ec.nend_bits = 0; ec.nbits_total = 0; ec.offs = 0; ec.rng = 0; ec.val = 0; ec.ext = 0; ec.rem = 0; ec.error = 0;
               ec_st(&L->ec_silk, &ec);
            }
            WV_LDS SilkLdsA *SA = &SL->a; WV_LDS SilkLdsB *SB = &SL->b;
            SdDecControl dc;
            dc.nChannelsAPI = CC; dc.nChannelsInternal = wv_uni(sdh->lastChannelsInternal); dc.API_sampleRate = Fs;
            dc.internalSampleRate = wv_uni(sdh->lastInternalRate); dc.payloadSize_ms = imax(10, 1000 * audiosize / Fs);
            int decoded = 0;
            do {
               int n = silk_decode_wave(sdh, &gs->silk.cng_exc_buf_Q14[0][0], dc, SD_FLAG_PACKET_LOST, decoded == 0, &L->ec_silk, buf, SA, SB, L->sh.r);
               wv_sync();
               if (n < 0) {                                                        /* "PLC failure should not be fatal" (:466-471) */
                  n = audiosize;
                  for (int c = 0; c < CC; c++) FOR_LANES(i, n) SB->rs_out[c][i] = 0;
                  wv_sync();
               }
               const int m = imin(n, audiosize - decoded);                        /* a 10 ms SILK frame may be longer than what is asked for (pcm_too_small, :411-420) */
               FOR_LANES(it, m * CC) { const int i = it / CC, c = it - i * CC; pcm[(size_t)(decoded + i) * CC + c] = SB->rs_out[c][i]; }
               decoded += n;
            } while (decoded < audiosize);
         }
         wv_sync();
         FOR_LANES(i, (int)(OA_SILK_HOT_BYTES / 4)) ((i32 *)&gs->silk)[i] = SL->hot[i];
         wv_sync();
      }
      LANE0 { st->start = mode != 1002 ? 17 : 0; }
      wv_sync();
      if (mode != 1000) {
         int r = celt_decode_frame_wave(L, gs, 0, imin(F20, audiosize), pcm, 0, mode != 1002);
         if (r < 0) return r;
      }
      done += audiosize;
      LANE0 { st->rangeFinal = 0; st->prev_mode = mode; st->prev_redundancy = 0; }
      wv_sync();
      if (frame_size - done > 0 && frame_size <= F20) break;      /* a single call conceals one legal frame size; the caller loops */
   }
   return done;
}

/* OPUS_RESET_STATE of the CELT decoder (celt_decoder.c:1794-1814): everything from `rng` on, then the -28 dB energy floors */
WV_DEVN void celt_reset_wave(WV_LDS DecLds *L, OaDecStream *gs)
{
   WV_LDS OaDecScalars *st = &L->st;
   wv_sync();
   LANE0 {
      st->rng = 0; st->error = 0; st->last_pitch_index = 0; st->loss_duration = 0; st->plc_duration = 0; st->last_frame_type = 0; st->skip_plc = 1;
      st->postfilter_period = st->postfilter_period_old = st->postfilter_gain = st->postfilter_gain_old = st->postfilter_tapset = st->postfilter_tapset_old = 0;
      st->prefilter_and_fold = 0; st->preemph_memD[0] = st->preemph_memD[1] = 0; st->hist_head = 0;
   }
   FOR_LANES(i, 2 * NBE) { L->oldBandE[i] = 0; L->backgroundLogE[i] = 0; L->oldLogE[i] = L->oldLogE2[i] = -(28 << 24); }
   FOR_LANES(i, 2 * OA_DEC_HISTORY) gs->hist[i] = 0;
   FOR_LANES(i, 2 * OA_OVERLAP) gs->overlap_mem[i] = 0;
   FOR_LANES(i, 2 * 24) gs->plc_lpc[i] = 0;
   wv_sync();
}

/* smooth_fade (src/opus_decoder.c:234-253): cross-fade over `overlap` samples with the squared CELT window */
WV_DEV void oa_smooth_fade_wave(const i16 *in1, const i16 *in2, i16 *out, int overlap, int CC, int inc = 1)
{
   FOR_LANES(it, overlap * CC) {
      const int i = it / CC;
      i32 w = ct_window[i * inc]; w = mult16_16_q15(w, w);
      out[it] = (i16)((mult16_16(w, in2[it]) + mult16_16(Q15ONE - w, in1[it])) >> 15);
   }
}

/* the decoder gain on the concealed fade source of a mode transition (OaDecScalars.transition_gain_Q16): MULT16_32_P16 + SATURATE(., 32767) as at src/opus_decoder.c:700-712 */
WV_DEV void oa_transition_gain_wave(const WV_LDS OaDecScalars *st, i16 *x, int n)
{
   const i32 gain = wv_uni(st->transition_gain_Q16);
   if (!gain) return;
   wv_sync();
   FOR_LANES(i, n) { const i32 y = (i32)(((i64)x[i] * gain + 32768) >> 16); x[i] = (i16)(y > 32767 ? 32767 : y < -32767 ? -32767 : y); }
   wv_sync();
}
/* One Opus frame with payload (opus_decode_frame, src/opus_decoder.c:271-714, data != NULL): SILK part, redundancy, CELT part, mode transitions.
 * `data` = the frame's bytes in HBM, len >= 2.  Returns the frame size or a negative OA_ERR_*. */
template <bool FAST = false> WV_DEVN int oa_decode_frame_wave(WV_LDS DecLds *L, OaDecStream *gs, const u8 *data, int len, int audiosize, i16 *pcm, int CC, int decode_fec = 0, int cut = 0)
{
   if (FAST) {                                                                  /* CELT-only steady state: opus_decode_frame with mode == prev_mode == CELT_ONLY (or a fresh decoder), no redundancy */
      WV_LDS OaDecScalars *st = &L->st;
      const int Fs = oa_dec_fs(st), F20 = Fs / 50;
      const int bandwidth = wv_uni(st->bandwidth), mode = wv_uni(st->mode);
      wv_sync();
      FOR_LANES(i, len) L->packet[1 + i] = data[i];
      wv_sync();
      {
         int endband = 21;
         switch (bandwidth) { case 1101: endband = 13; break; case 1102: case 1103: endband = 17; break; case 1104: endband = 19; break; default: endband = 21; }
         LANE0 { st->end = endband; st->start = 0; }
      }
      const int r = celt_decode_frame_wave<true>(L, gs, len, imin(F20, audiosize), pcm, 0, 0, cut);
      if (r < 0) return r;                                                      /* (OA_DEC_CUT too: oa_decode_packet_back finishes the frame) */
      LANE0 { st->rangeFinal = st->rng; st->prev_mode = mode; st->prev_redundancy = 0; }
      wv_sync();
      return audiosize;
   }
   decode_fec = wv_uni(decode_fec);
   WV_LDS DecShared *sh = &L->sh;
   WV_LDS OaDecScalars *st = &L->st;
   const int Fs = oa_dec_fs(st), finc = 48000 / Fs;
   const int F20 = Fs / 50, F5 = Fs / 200, F2_5 = Fs / 400;
   const int mode = wv_uni(st->mode), bandwidth = wv_uni(st->bandwidth), prev_mode = wv_uni(st->prev_mode), prev_red = wv_uni(st->prev_redundancy);
   const int frame_size = audiosize;
   int transition = 0, redundancy = 0, celt_to_silk = 0, redundancy_bytes = 0;
   u32 redundant_rng = 0;
   const int celt_accum = mode != 1002;
   if (prev_mode > 0 && ((mode == 1002 && prev_mode != 1002 && !prev_red) || (mode != 1002 && prev_mode == 1002))) transition = 1;
   if (transition && mode == 1002) {                                            /* SILK/hybrid -> CELT without redundancy: 5 ms of concealment in the old mode as the fade source (:388-393) */
      const int r = oa_conceal_wave(L, gs, imin(F5, audiosize), gs->trans, CC);
      if (r < 0) return r;
      oa_transition_gain_wave(st, gs->trans, imin(F5, audiosize) * CC);
      LANE0 { st->mode = mode; st->start = 0; }
   }

   wv_sync();
   FOR_LANES(i, len) L->packet[1 + i] = data[i];
   wv_sync();
   if (mode != 1002) {
      /* ---- SILK part (:404-497) ---- */
      WV_LDS SilkLdsAll *SL = (WV_LDS SilkLdsAll *)&L->BC;
      FOR_LANES(i, (int)(OA_SILK_HOT_BYTES / 4)) SL->hot[i] = ((const i32 *)&gs->silk)[i];              /* SILK state -> LDS, coalesced */
      wv_sync();
      {
         WV_LDS OaSilkDec *sdh = (WV_LDS OaSilkDec *)SL->hot;
         WV_LDS u8 *buf = L->packet + 1;
         WV_LDS SilkLdsA *SA = &SL->a; WV_LDS SilkLdsB *SB = &SL->b;
         SdDecControl dc;
         dc.nChannelsAPI = CC; dc.nChannelsInternal = wv_uni(st->stream_channels); dc.API_sampleRate = Fs;
         dc.internalSampleRate = mode == 1001 ? 16000 : bandwidth == 1101 ? 8000 : bandwidth == 1102 ? 12000 : 16000;
         dc.payloadSize_ms = imax(10, 1000 * audiosize / Fs);
         LANE0 {
            EcCtx ec;
            k_ec_dec_init(&ec, buf, (u32)len);
            ec_st(&L->ec_silk, &ec);
            if (prev_mode == 1002) {                                            /* silk_ResetDecoder (silk/dec_API.c:91) */
               sd_reset(&sdh->ch[0], &gs->silk.cng_exc_buf_Q14[0][0]); sd_reset(&sdh->ch[1], &gs->silk.cng_exc_buf_Q14[1][0]);
               sdh->pred_prev_Q13[0] = sdh->pred_prev_Q13[1] = 0; sdh->sMid[0] = sdh->sMid[1] = sdh->sSide[0] = sdh->sSide[1] = 0;
               sdh->prev_decode_only_middle = 0;
            }
            sdh->lastInternalRate = dc.internalSampleRate; sdh->lastChannelsInternal = dc.nChannelsInternal;
         }
         int decoded = 0, rr = 0;
         do {
            const int n = silk_decode_wave(sdh, &gs->silk.cng_exc_buf_Q14[0][0], dc, decode_fec ? SD_FLAG_DECODE_LBRR : SD_FLAG_DECODE_NORMAL, decoded == 0, &L->ec_silk, buf, SA, SB, sh->r);
            wv_sync();
            if (n < 0) { rr = n; break; }
            FOR_LANES(it, n * CC) { const int i = it / CC, c = it - i * CC; pcm[(size_t)(decoded + i) * CC + c] = SB->rs_out[c][i]; }
            decoded += n;
         } while (decoded < frame_size);
         LANE0 {
            /* ---- redundancy signalling (:499-526) ---- */
            EcCtx ec; ec_ld(&ec, &L->ec_silk);
            int red = 0, c2s = 0, rbytes = 0, newlen = len;
            if (rr == 0 && !decode_fec && k_ec_tell(&ec, buf) + 17 + 20 * (mode == 1001) <= 8 * len) {
               red = mode == 1001 ? k_ec_dec_bit_logp(&ec, buf, 12) : 1;
               if (red) {
                  c2s = k_ec_dec_bit_logp(&ec, buf, 1);
                  rbytes = mode == 1001 ? (int)k_ec_dec_uint(&ec, buf, 256) + 2 : len - ((k_ec_tell(&ec, buf) + 7) >> 3);
                  newlen = len - rbytes;
                  if (newlen * 8 < k_ec_tell(&ec, buf)) { newlen = 0; rbytes = 0; red = 0; }
                  ec.storage -= (u32)rbytes;
               }
            }
            ec_st(&L->ec_silk, &ec);
            sh->r[0] = rr; sh->r[1] = red; sh->r[2] = c2s; sh->r[3] = rbytes; sh->r[4] = newlen;
         }
      }
      FOR_LANES(i, (int)(OA_SILK_HOT_BYTES / 4)) ((i32 *)&gs->silk)[i] = SL->hot[i];
      wv_sync();
      const int rr = wv_uni(sh->r[0]);
      if (rr < 0) return rr;
      redundancy = wv_uni(sh->r[1]); celt_to_silk = wv_uni(sh->r[2]); redundancy_bytes = wv_uni(sh->r[3]); len = wv_uni(sh->r[4]);
   }
   const int start_band = mode != 1002 ? 17 : 0;
   if (redundancy) transition = 0;
   if (transition && mode != 1002) {                                            /* CELT -> SILK/hybrid: 5 ms of CELT concealment as the fade source (:534-539) */
      const int r = oa_conceal_wave(L, gs, imin(F5, audiosize), gs->trans, CC);
      if (r < 0) return r;
      oa_transition_gain_wave(st, gs->trans, imin(F5, audiosize) * CC);
      wv_sync();
      FOR_LANES(i, len + redundancy_bytes) L->packet[1 + i] = data[i];           /* (concealment does not touch the packet buffer; reloaded for clarity of state) */
      wv_sync();
   }
   {
      int endband = 21;
      switch (bandwidth) { case 1101: endband = 13; break; case 1102: case 1103: endband = 17; break; case 1104: endband = 19; break; default: endband = 21; }
      LANE0 { st->end = endband; }
   }
   if (redundancy && celt_to_silk) {                                            /* 5 ms redundant CELT frame ahead of the SILK audio (:571-583) */
      wv_sync();
      FOR_LANES(i, redundancy_bytes) L->packet[1 + i] = data[len + i];
      LANE0 st->start = 0;
      const int r = celt_decode_frame_wave(L, gs, redundancy_bytes, F5, gs->red, 0, 0);
      if (r < 0) return r;
      redundant_rng = (u32)wv_uni((i32)st->rng);
      wv_sync();
      FOR_LANES(i, len) L->packet[1 + i] = data[i];
      wv_sync();
   }
   LANE0 st->start = start_band;
   if (mode != 1000) {
      const int celt_frame_size = imin(F20, frame_size);
      if (mode != prev_mode && prev_mode > 0 && !prev_red) celt_reset_wave(L, gs);
      const int r = celt_decode_frame_wave(L, gs, decode_fec ? 0 : len, celt_frame_size, pcm, mode == 1001, celt_accum);   /* decode_fec: the CELT layer is concealed (:598) */
      if (r < 0) return r;
      LANE0 st->rangeFinal = st->rng;
   } else {
      if (prev_mode == 1001 && !(redundancy && celt_to_silk && prev_red)) {     /* hybrid -> SILK: let the CELT MDCT fade out on a silence frame (:606-615) */
         wv_sync();
         LANE0 { L->packet[1] = 0xFF; L->packet[2] = 0xFF; st->start = 0; }
         const int r = celt_decode_frame_wave(L, gs, 2, F2_5, pcm, 0, celt_accum);
         if (r < 0) return r;
      }
      LANE0 st->rangeFinal = L->ec_silk.rng;
   }
   wv_sync();
   if (redundancy && !celt_to_silk) {                                           /* 5 ms redundant CELT frame after the SILK audio (:624-633) */
      celt_reset_wave(L, gs);
      FOR_LANES(i, redundancy_bytes) L->packet[1 + i] = data[len + i];
      LANE0 st->start = 0;
      const int r = celt_decode_frame_wave(L, gs, redundancy_bytes, F5, gs->red, 0, 0);
      if (r < 0) return r;
      redundant_rng = (u32)wv_uni((i32)st->rng);
      wv_sync();
      oa_smooth_fade_wave(pcm + (size_t)CC * (frame_size - F2_5), gs->red + CC * F2_5, pcm + (size_t)CC * (frame_size - F2_5), F2_5, CC, finc);
   }
   if (redundancy && celt_to_silk && (prev_mode != 1000 || prev_red)) {
      wv_sync();
      FOR_LANES(i, F2_5 * CC) pcm[i] = gs->red[i];
      oa_smooth_fade_wave(gs->red + CC * F2_5, pcm + CC * F2_5, pcm + CC * F2_5, F2_5, CC, finc);
   }
   if (transition) {
      wv_sync();
      if (audiosize >= F5) {
         FOR_LANES(i, F2_5 * CC) pcm[i] = gs->trans[i];
         oa_smooth_fade_wave(gs->trans + CC * F2_5, pcm + CC * F2_5, pcm + CC * F2_5, F2_5, CC, finc);
      } else oa_smooth_fade_wave(gs->trans, pcm, pcm, F2_5, CC, finc);
   }
   wv_sync();
   LANE0 {
      if (len <= 1) st->rangeFinal = 0;               /* (a corrupt redundancy length leaves no payload: `len = 0`, src/opus_decoder.c:517, and the frame reports range 0, :676) */
      else st->rangeFinal ^= redundant_rng;
      st->prev_mode = mode; st->prev_redundancy = redundancy && !celt_to_silk;
   }
   wv_sync();
   return audiosize;
}

/* one packet of one stream: returns samples per channel (written to pcm_out, interleaved) or a negative OPUS_* code */
/* cont (FAST only): the stream's continuation record -- a packet of one 10 / 20 ms frame stops in front of its bands, the wave's LDS image goes to the record and the
 * function returns 1 (the frame is finished by oa_celt_dpvq_kernel and oa_celt_dback_kernel: oa_decode_packet_back); otherwise 0 */
template <bool FAST = false> WV_DEV int oa_decode_packet(WV_LDS DecLds *L, OaDecStream *gs, const u8 *data, int len, int frame_size, i16 *pcm_out, i32 *nsamples_out, u32 *rng_out, int decode_fec = 0,
      CeltDecCont *cont = 0)
{
   decode_fec = wv_uni(decode_fec);
   WV_LDS DecShared *sh = &L->sh;
   WV_LDS OaDecScalars *st = &L->st;
   {
      const i32 *g = (const i32 *)&gs->s;
      WV_LDS i32 *d = (WV_LDS i32 *)st;
      FOR_LANES(i, (int)(sizeof(OaDecScalars) / 4)) d[i] = g[i];
      FOR_LANES(i, 2 * NBE) { L->oldBandE[i] = gs->oldBandE[i]; L->oldLogE[i] = gs->oldLogE[i]; L->oldLogE2[i] = gs->oldLogE2[i]; L->backgroundLogE[i] = gs->backgroundLogE[i]; }
   }
   wv_sync();
   LANE0 {
      int ret = 0, offset = 0;
      sh->count = 0; sh->nb_samples = 0; sh->r[5] = 0;
      if (frame_size <= 0) ret = OA_ERR_BAD_ARG;
      else if (len == 0 || data == 0) { ret = frame_size % (oa_dec_fs(st) / 400) != 0 ? OA_ERR_BAD_ARG : 0; sh->count = -1; }        /* packet loss: conceal frame_size samples */
      else if (len < 0) ret = OA_ERR_BAD_ARG;
      else {
         const int toc = data[0];
         const int packet_mode = (toc & 0x80) ? 1002 : ((toc & 0x60) == 0x60 ? 1001 : 1000);
         int packet_bandwidth;
         if (toc & 0x80) { packet_bandwidth = 1102 + ((toc >> 5) & 0x3); if (packet_bandwidth == 1102) packet_bandwidth = 1101; }
         else if ((toc & 0x60) == 0x60) packet_bandwidth = (toc & 0x10) ? 1105 : 1104;
         else packet_bandwidth = 1101 + ((toc >> 5) & 0x3);
         const int packet_frame_size = oa_samples_per_frame(toc, oa_dec_fs(st));
         const int count = oa_packet_parse(data, len, sh->size, &offset);
         if (count < 0) ret = count;
         else if (decode_fec && (frame_size < packet_frame_size || packet_mode == 1002 || st->mode == 1002)) sh->count = -1;   /* no usable LBRR: conceal (src/opus_decoder.c:791-797) */
         else if (!decode_fec && count * packet_frame_size > frame_size) ret = OA_ERR_BUFFER_TOO_SMALL;
         else {
            int endband = 21;
            switch (packet_bandwidth) { case 1101: endband = 13; break; case 1102: case 1103: endband = 17; break; case 1104: endband = 19; break; default: endband = 21; }
            if (decode_fec) {                       /* the leading samples are concealed with the PREVIOUS packet's parameters; the new ones apply from the LBRR frame on (src/opus_decoder.c:798-823) */
               sh->r[5] = 1; sh->r[0] = packet_mode; sh->r[1] = packet_bandwidth; sh->r[2] = (toc & 0x4) ? 2 : 1; sh->r[3] = endband;
            } else {
               st->mode = packet_mode; st->bandwidth = packet_bandwidth; st->frame_size = packet_frame_size; st->stream_channels = (toc & 0x4) ? 2 : 1;
               sh->r[3] = endband;                  /* applies from the first frame that carries data: a DTX frame is concealed with the band limit in force (:316, :538 `if (bandwidth)`) */
            }
            sh->count = count; sh->packet_frame_size = packet_frame_size; sh->frame_bytes_off = offset;
         }
      }
      sh->ret = ret;
   }
   wv_sync();
   int ret = wv_uni(sh->ret);
   const int count = wv_uni(sh->count), pfs = wv_uni(sh->packet_frame_size), CC = wv_uni(st->channels);
   int off = wv_uni(sh->frame_bytes_off), nb = 0;
   const int fec = wv_uni(sh->r[5]);
   const int fec_mode = wv_uni(sh->r[0]), fec_bw = wv_uni(sh->r[1]), fec_ch = wv_uni(sh->r[2]), fec_end = wv_uni(sh->r[3]);     /* (the slots are reused by the frame functions below) */
   if (!FAST && fec && ret >= 0) {
      /* in-band FEC (src/opus_decoder.c:798-824): conceal everything before the last packet_frame_size samples, then decode the LBRR copy of
       * the first frame in this packet into them */
      const int duration_copy = wv_uni(st->last_packet_duration);
      while (nb < frame_size - pfs) {
         const int r = oa_conceal_wave(L, gs, imin(frame_size - pfs - nb, wv_uni(st->frame_size)), pcm_out + (size_t)nb * CC, CC);   /* never more than the last TOC's frame size at a time (:316-322) */
         if (r < 0) { ret = r; LANE0 st->last_packet_duration = duration_copy; wv_sync(); break; }
         nb += r;
      }
      if (ret >= 0) {
         const int flen = wv_uni(sh->size[0]);
         LANE0 { st->mode = fec_mode; st->bandwidth = fec_bw; st->frame_size = pfs; st->stream_channels = fec_ch; if (flen > 1) st->start = 0; }
         wv_sync();
         /* a first frame without data (DTX) has no LBRR copy either: opus_decode_frame turns it into concealment in the previous mode, final range 0 (:316-322, :676) */
         const int r = flen <= 1 ? oa_conceal_wave(L, gs, pfs, pcm_out + (size_t)nb * CC, CC)
                                 : oa_decode_frame_wave(L, gs, data + off, flen, pfs, pcm_out + (size_t)nb * CC, CC, 1);
         if (r < 0) ret = r; else nb = frame_size;
      }
   }
   for (int f = 0; f < count && ret >= 0 && !fec; f++) {
      const int flen = wv_uni(sh->size[f]);
      int r;
      if (!FAST && flen <= 1) {           /* DTX / lost frame inside a packet: opus_decode_frame with data = NULL, at most the TOC's frame size (:316-322) */
         r = oa_conceal_wave(L, gs, imin(frame_size - nb, pfs), pcm_out + (size_t)nb * CC, CC);
      } else {
         LANE0 { st->start = 0; }                        /* (the packet's band limit applies inside, after the transition fade sources are concealed with the old one: src/opus_decoder.c:388, :540, :547) */
         wv_sync();
         r = oa_decode_frame_wave<FAST>(L, gs, data + off, flen, pfs, pcm_out + (size_t)nb * CC, CC, 0, FAST && cont != 0 && count == 1);
      }
      if (r < 0) ret = r;
      else nb += r;
      off += flen;
   }
   if (!FAST && count == -1 && ret >= 0) {       /* whole packet lost (opus_decoder.c:756-769) */
      while (nb < frame_size) {
         int r = oa_conceal_wave(L, gs, imin(frame_size - nb, wv_uni(st->frame_size)), pcm_out + (size_t)nb * CC, CC);   /* never more than the last TOC's frame size at a time (src/opus_decoder.c:316-322) */
         if (r < 0) { ret = r; break; }
         nb += r;
      }
   }
   if (FAST && ret == OA_DEC_CUT) {                 /* the frame stops here: everything the rest of it needs is in the wave's LDS up to the phase scratch */
      wv_sync();
      FOR_LANES(i, (int)(offsetof(DecLds, BC) / 4)) cont->image[i] = ((const WV_LDS i32 *)L)[i];
      wv_sync();
      return 1;
   }
   if (ret >= 0) { ret = nb; LANE0 st->last_packet_duration = nb; wv_sync(); }
   /* ---- store state ---- */
   {
      i32 *g = (i32 *)&gs->s;
      const WV_LDS i32 *d = (const WV_LDS i32 *)st;
      FOR_LANES(i, (int)(sizeof(OaDecScalars) / 4)) g[i] = d[i];
      FOR_LANES(i, 2 * NBE) { gs->oldBandE[i] = L->oldBandE[i]; gs->oldLogE[i] = L->oldLogE[i]; gs->oldLogE2[i] = L->oldLogE2[i]; gs->backgroundLogE[i] = L->backgroundLogE[i]; }
   }
   LANE0 { *nsamples_out = ret; *rng_out = st->rangeFinal; }
   return 0;
}
/* The rest of a packet whose frame was stopped in front of its bands (oa_decode_packet<true> / oa_decode_hybrid_tail with a continuation record) once oa_celt_dpvq_kernel has
 * decoded them: the LDS image back, the frame from behind the bands (celt_decode_frame_tail), then what oa_decode_frame_wave and oa_decode_packet do after a CELT-only frame
 * in the steady state -- or oa_decode_hybrid_tail after the CELT layer of a hybrid one (accumulated onto the SILK audio) -- : final range, mode memory, duration, state store. */
WV_DEV void oa_decode_packet_back(WV_LDS DecLds *L, OaDecStream *gs, CeltDecCont *cont, i16 *pcm_out, i32 *nsamples_out, u32 *rng_out, int defer_deemph)
{
   wv_sync();
   FOR_LANES(i, (int)(offsetof(DecLds, BC) / 4)) ((WV_LDS i32 *)L)[i] = cont->image[i];
   wv_sync();
   WV_LDS OaDecScalars *st = &L->st;
   const int mode = wv_uni(st->mode), pfs = wv_uni(st->frame_size), len = wv_uni(L->sh.len);
   if (defer_deemph) { LANE0 { cont->hdr[0] = 0; } }                      /* (a frame that fails leaves nothing for the de-emphasis pass) */
   const int r = celt_decode_frame_tail<true>(L, gs, len, pfs, pcm_out, mode == 1001, defer_deemph ? cont->hdr : (i32 *)0);
   if (r >= 0) { LANE0 { st->rangeFinal = st->rng; st->prev_mode = mode; st->prev_redundancy = 0; st->last_packet_duration = pfs; } }
   wv_sync();
   {
      i32 *g = (i32 *)&gs->s;
      const WV_LDS i32 *d = (const WV_LDS i32 *)st;
      FOR_LANES(i, (int)(sizeof(OaDecScalars) / 4)) g[i] = d[i];
      FOR_LANES(i, 2 * NBE) { gs->oldBandE[i] = L->oldBandE[i]; gs->oldLogE[i] = L->oldLogE[i]; gs->oldLogE2[i] = L->oldLogE2[i]; gs->backgroundLogE[i] = L->backgroundLogE[i]; }
   }
   LANE0 { *nsamples_out = r < 0 ? r : pfs; *rng_out = st->rangeFinal; }
}
/* The de-emphasis (celt_decoder.c:318: y[n] = sat(x[n] + m), m = 0.85 y[n] rounded -- a chain no scan reproduces) and the PCM store of the frames oa_celt_dback_kernel
 * synthesised, ONE LANE PER STREAM: the one-wave-per-stream kernels walk this recursion on one lane per channel while 62 lanes wait; here 64 streams share the instruction.
 * A lane reads its stream's samples sixteen at a time per channel (one 64-byte line), walks the chains of its channels side by side, and writes interleaved int16 PCM
 * (API rates below 48 kHz keep every ds-th sample; hybrid frames add onto the SILK audio: celt/arch.h:172 ADD_RES). */
struct alignas(16) OaQuad { i32 x, y, z, w; };
WV_DEV void oa_deemph_lane(OaDecStream *gs, const CeltDecCont *cont, i16 *pcm_out)
{
   const int N = cont->hdr[0], flags = cont->hdr[1];
   if (N <= 0) return;
   const int CC = gs->s.channels, Fs = gs->s.Fs ? gs->s.Fs : 48000, ds = 48000 / Fs, accum = flags & 1;
   const i32 *x0 = cont->xg + ((flags & 2) ? N : 0), *x1 = cont->xg + ((flags & 2) ? 0 : N);
   i32 m0 = gs->s.preemph_memD[0], m1 = gs->s.preemph_memD[1];
   const bool packed = CC == 2 && ds == 1 && ((size_t)pcm_out & 15) == 0;        /* (the batch's own and torch's buffers are; a caller's device pointer need not be) */
   for (int j0 = 0; j0 < N; j0 += 16) {
      i32 a[16], b[16];
#pragma unroll
      for (int q = 0; q < 4; q++) { const OaQuad v = *(const OaQuad *)(x0 + j0 + 4 * q); a[4 * q] = v.x; a[4 * q + 1] = v.y; a[4 * q + 2] = v.z; a[4 * q + 3] = v.w; }
      if (CC == 2) {
#pragma unroll
         for (int q = 0; q < 4; q++) { const OaQuad v = *(const OaQuad *)(x1 + j0 + 4 * q); b[4 * q] = v.x; b[4 * q + 1] = v.y; b[4 * q + 2] = v.z; b[4 * q + 3] = v.w; }
      }
#pragma unroll
      for (int k = 0; k < 16; k++) { a[k] = saturate(a[k] + m0, SIG_SAT); m0 = mult16_32_q15(27853, a[k]); a[k] = sig2word16(a[k]); }
      if (CC == 2) {
#pragma unroll
         for (int k = 0; k < 16; k++) { b[k] = saturate(b[k] + m1, SIG_SAT); m1 = mult16_32_q15(27853, b[k]); b[k] = sig2word16(b[k]); }
      }
      if (packed) {
         u32 *o = (u32 *)(pcm_out + 2 * j0);
#pragma unroll
         for (int q = 0; q < 4; q++) {
            OaQuad w;
            if (accum) {                                      /* onto the SILK audio that is there (ADD_RES, celt/arch.h:172) */
               const OaQuad old = *(const OaQuad *)(o + 4 * q);
#define OA_AS(o_, k_) { const i32 l_ = (i32)(i16)(o_) + a[k_], r_ = ((o_) >> 16) + b[k_]; a[k_] = l_ > 32767 ? 32767 : l_ < -32768 ? -32768 : l_; b[k_] = r_ > 32767 ? 32767 : r_ < -32768 ? -32768 : r_; }
               OA_AS(old.x, 4 * q) OA_AS(old.y, 4 * q + 1) OA_AS(old.z, 4 * q + 2) OA_AS(old.w, 4 * q + 3)
#undef OA_AS
            }
#define OA_PK(k_) (i32)(((u32)a[k_] & 0xffffu) | ((u32)b[k_] << 16))
            w.x = OA_PK(4 * q); w.y = OA_PK(4 * q + 1); w.z = OA_PK(4 * q + 2); w.w = OA_PK(4 * q + 3);
#undef OA_PK
            *(OaQuad *)(o + 4 * q) = w;
         }
      } else {
#pragma unroll
         for (int k = 0; k < 16; k++) {
            const int i = j0 + k;
            if (ds == 1 || (u32)i % (u32)ds == 0) {
               const int it = (int)((u32)i / (u32)ds) * CC;
               if (accum) {
                  i32 w = (i32)pcm_out[it] + a[k]; pcm_out[it] = (i16)(w > 32767 ? 32767 : w < -32768 ? -32768 : w);
                  if (CC == 2) { w = (i32)pcm_out[it + 1] + b[k]; pcm_out[it + 1] = (i16)(w > 32767 ? 32767 : w < -32768 ? -32768 : w); }
               } else { pcm_out[it] = (i16)a[k]; if (CC == 2) pcm_out[it + 1] = (i16)b[k]; }
            }
         }
      }
   }
   gs->s.preemph_memD[0] = m0;
   if (CC == 2) gs->s.preemph_memD[1] = m1;
}
/* between oa_sdec_lane_kernel and oa_decode_hyb_kernel, per stream: the range decoder behind the SILK layer and the redundancy flag, and where the coded frame lies in the packet */
struct OaHybCont { EcCtx ec; i32 off, flen; };
/* The CELT layer of a hybrid packet whose SILK layer oa_sdec_lane_kernel has decoded (silk_dec_lane.h): what oa_decode_packet + oa_decode_frame_wave do for a hybrid frame in
 * the steady state (mode == prev_mode == hybrid, one coded frame, no redundancy: src/opus_decoder.c:540-690) from the point where the SILK audio is in pcm_out and the
 * range decoder stands behind the redundancy flag (*cont): bands 17.. of the CELT-only fast kernel's frame function, accumulated onto the SILK audio. */
/* dcont: as in oa_decode_packet -- the frame stops in front of its bands, returns 1 */
WV_DEV int oa_decode_hybrid_tail(WV_LDS DecLds *L, OaDecStream *gs, const u8 *data, i16 *pcm_out, i32 *nsamples_out, u32 *rng_out, const OaHybCont *cont, CeltDecCont *dcont = 0)
{
   WV_LDS OaDecScalars *st = &L->st;
   {
      const i32 *g = (const i32 *)&gs->s;
      WV_LDS i32 *d = (WV_LDS i32 *)st;
      FOR_LANES(i, (int)(sizeof(OaDecScalars) / 4)) d[i] = g[i];
      FOR_LANES(i, 2 * NBE) { L->oldBandE[i] = gs->oldBandE[i]; L->oldLogE[i] = gs->oldLogE[i]; L->oldLogE2[i] = gs->oldLogE2[i]; L->backgroundLogE[i] = gs->backgroundLogE[i]; }
   }
   wv_sync();
   const int toc = wv_uni((int)data[0]), Fs = oa_dec_fs(st);
   const int pfs = (toc & 0x08) ? Fs / 50 : Fs / 100, bw = (toc & 0x10) ? 1105 : 1104, flen = wv_uni(cont->flen), off = wv_uni(cont->off);
   LANE0 {
      st->mode = 1001; st->bandwidth = bw; st->frame_size = pfs; st->stream_channels = (toc & 0x4) ? 2 : 1;
      st->start = 17; st->end = bw == 1104 ? 19 : 21;
      EcCtx ec = cont->ec;
      ec_st(&L->ec_silk, &ec);
   }
   FOR_LANES(i, flen) L->packet[1 + i] = data[off + i];
   wv_sync();
   const int r = celt_decode_frame_wave<true>(L, gs, flen, pfs, pcm_out, 1, 1, dcont != 0);
   if (r == OA_DEC_CUT) {
      wv_sync();
      FOR_LANES(i, (int)(offsetof(DecLds, BC) / 4)) dcont->image[i] = ((const WV_LDS i32 *)L)[i];
      wv_sync();
      return 1;
   }
   if (r >= 0) { LANE0 { st->rangeFinal = st->rng; st->prev_mode = 1001; st->prev_redundancy = 0; st->last_packet_duration = pfs; } }
   wv_sync();
   {
      i32 *g = (i32 *)&gs->s;
      const WV_LDS i32 *d = (const WV_LDS i32 *)st;
      FOR_LANES(i, (int)(sizeof(OaDecScalars) / 4)) g[i] = d[i];
      FOR_LANES(i, 2 * NBE) { gs->oldBandE[i] = L->oldBandE[i]; gs->oldLogE[i] = L->oldLogE[i]; gs->oldLogE2[i] = L->oldLogE2[i]; gs->backgroundLogE[i] = L->backgroundLogE[i]; }
   }
   LANE0 { *nsamples_out = r < 0 ? r : pfs; *rng_out = st->rangeFinal; }
   return 0;
}
#endif

/* opus_analysis.h — the tonality / music / bandwidth analysis of the Opus encoder on one wavefront: src/analysis.c (run_analysis :954, tonality_analysis :445,
 * tonality_get_info :232, downmix_and_resample :157, silk_resampler_down2_hp :114) and src/mlp.c (analysis_compute_dense :70, analysis_compute_gru :92) of the
 * reference's FIXED_POINT build WITH the float API -- the fixed-point library users get by default: the 480-point FFT is the codec's own fixed-point transform, what
 * follows it is IEEE single precision.  It steers the encoder's mode / bandwidth decisions and the CELT encoder's allocation tuning at complexity 10
 * (src/opus_encoder.c:1249-1320, celt/celt_encoder.c:935,:1226,:1494,:1632,:1658,:2043,:2328,:2610).
 *
 * Float results have to come out bit for bit as the C code produces them, so: no contraction of a*b+c into fused multiply-adds (pragma below), every sum that the
 * reference accumulates in a loop is accumulated in the same order, and expressions keep the reference's association and its int / float / double promotions.
 * What is parallel is what is independent:
 *   - the three all-pass sections of the 2:1 decimator are three recursions (even samples, odd samples, the high-pass branch): three lanes;
 *   - the 239 spectral bins (two arctangents each): one lane per bin;
 *   - the 18 analysis bands: one lane per band runs the band's sums in bin order and all of the band's state updates;
 *   - the 8 x 8 spectral distances, the 2 x 8 cepstral sums, the neurons of a network layer: one lane each, the inner sum serial;
 *   - the short recurrences across the bands (frame sums, leakage, masking follower) and tonality_get_info's walk over the info ring: lane 0, on LDS copies.
 * State: OaAnalysis in the stream's HBM record (the reference's TonalityAnalysisState from `angle` on); working set: AnLds (6.1 KB) in the phase-aliased LDS of
 * the frame, which nothing else uses yet at the top of opus_encode_native. */
#ifndef OPUS_AMD_OPUS_ANALYSIS_H
#define OPUS_AMD_OPUS_ANALYSIS_H
#include "analysis_tables.h"
#include "analysis_state.h"
#include <math.h>
#ifdef __clang__
#pragma clang fp contract(off)
#endif
#ifndef LANE0
#define LANE0 for (int l0_ = (wv_sync(), wv_prio_serial(), 1); l0_; l0_ = (wv_prio_normal(), wv_sync(), 0)) if (wv_lane() == 0)
#define FOR_LANES(i, n) for (int i = wv_lane(); i < (n); i += WV_WIDTH)
#endif

/* celt/arch.h:100-105 (macros: the comparison decides, also for -0 and NaN) and the FIXED_POINT ABS16 (:227) */
#define AN_MIN(a, b) ((a) < (b) ? (a) : (b))
#define AN_MAX(a, b) ((a) > (b) ? (a) : (b))
#define AN_ABS(x) ((x) < 0 ? (-(x)) : (x))

#ifndef AN_TIC          /* shader-clock section timers exist only in the -DOA_PHASE_TIMERS profiling build */
#define AN_TIC()
#define AN_TOC(bucket)
#endif
#ifndef AN2_TIC
#define AN2_TIC()
#define AN2_TOC(bucket)
#endif
#ifndef AN_FN           /* out of line: inlined into the kernels (-DAN_FN=WV_DEV) the analysis' live ranges push the frame's own code into more spills -- measured on the MI355X:
                           1.316 M frames/s and 368 KB / frame of HBM traffic inlined against 1.371 M and 300 KB out of line (profiles/r03_c) */
#define AN_FN WV_DEVN
#endif
#define AN_SCRATCH_WORDS 480          /* per-wave HBM words the analysis borrows (the second half of a frame's decimated input, until the window has read the old one) */

struct AnLds {
   union {
      i32 fft[960];                                                                   /* decimator input, then the 480-point complex FFT in place */
      struct { float tonality[240], noisiness[240], tonality2[240], binE[240]; } s;   /* ... then what the bins leave behind */
      struct { float tonality[AN_DETECT_SIZE], music_prob[AN_DETECT_SIZE], activity_probability[AN_DETECT_SIZE]; i32 bandwidth[AN_DETECT_SIZE]; } ring;   /* tonality_get_info */
   };
   union {
      i32 hbuf[480];                                                                  /* decimator: the odd samples / the high-pass branch */
      struct {
         float band_log2[AN_NB_TBANDS + 1], logE[AN_NB_TBANDS], band_tonality[AN_NB_TBANDS], leak_from[AN_NB_TBANDS + 1], leak_to[AN_NB_TBANDS + 1];
         float t_noisy[AN_NB_TBANDS], t_loud[AN_NB_TBANDS], t_relE[AN_NB_TBANDS], t_stat[AN_NB_TBANDS], E2[AN_NB_TBANDS], Em[AN_NB_TBANDS];
         float dist[64], mindist[8], BFCC[8], midE[8], features[25], layer_out[32], z[24], r[24], h[24], tmp[24], probs[2];
      } t;
   };
   float mem[32], cmean[8], std[9], rnn[24];                                          /* staged state of the feature / network tail */
   float E0, hp_ener, frame_tonality, tonality_slope, activity, frame_stationarity, lowECount, spec_variability, frame_noisiness, max_pitch_ratio;
   i32 bandwidth, left, aux[8];
};

/* celt/mathops.h:60 */
WV_DEV float an_fast_atan2f(float y, float x)
{
   const float cA = 0.43157974f, cB = 0.67848403f, cC = 0.08595542f, cE = (float)3.1415926535897931 / 2;
   const float x2 = x * x, y2 = y * y;
   if (x2 + y2 < 1e-18f) return 0;
   if (x2 < y2) { const float den = (y2 + cB * x2) * (y2 + cC * x2); return -x * y * (y2 + cA * x2) / den + (y < 0 ? -cE : cE); }
   else { const float den = (x2 + cB * y2) * (x2 + cC * y2); return x * y * (x2 + cA * y2) / den + (y < 0 ? -cE : cE) - (x * y < 0 ? -cE : cE); }
}
/* src/mlp.c:39,:55 */
WV_DEV float an_tansig(float x)
{
   const float N0 = 952.52801514f, N1 = 96.39235687f, N2 = 0.60863042f, D0 = 952.72399902f, D1 = 413.36801147f, D2 = 11.88600922f;
   const float X2 = x * x;
   float num = (N2 * X2 + N1) * X2 + N0;
   const float den = (D2 * X2 + D1) * X2 + D0;
   num = num * x / den;
   return AN_MAX(-1.f, AN_MIN(1.f, num));
}
WV_DEV float an_sigmoid(float x) { return .5f + .5f * an_tansig(.5f * x); }
WV_DEV int an_float2int(float x) { return (int)rintf(x); }                  /* lrintf: to nearest, ties to even */

/* downmix_and_resample (src/analysis.c:157) with the encoder's own arguments (c1 = 0, c2 = -2: all C <= 2 channels): `subframe` samples at 24 kHz, starting
 * `offset` samples (at 24 kHz) into the call's input, into y[] (HBM); input = the int16 samples of opus_encode (downmix_int, src/opus_encoder.c:780) or -- apcm --
 * the samples of the 24-bit / float entry points already in the signal domain (downmix_int24 :804, downmix_float :748).  Returns the high-pass energy (48 kHz only).
 * silk_resampler_down2_hp (:114): out[k] = (ap0(x[2k]) + ap1(x[2k+1])) / 2, hp[k] = ap0(x[2k]) + ap1'(-x[2k+1]); each first-order all-pass section rounds at every
 * step, so it is a serial recursion -- but the three are independent of each other: lanes 0, 1, 2.
 * mirror: the output also stays in LDS, W->hbuf [0, subframe at 24 kHz), for a caller that reads it back at once. */
WV_DEV i32 an_downmix_resample_wave(WV_LDS AnLds *W, OaAnalysis *A, const i16 *pcm, const i32 *apcm, i32 *y, int subframe, int offset, int C, int Fs, bool mirror = false)
{
   if (subframe == 0) return 0;
   if (Fs == 48000) { subframe *= 2; offset *= 2; }
   else if (Fs == 16000) { subframe = subframe * 2 / 3; offset = offset * 2 / 3; }
   AN2_TIC();
   for (int j0 = 0; j0 < subframe; j0 += 8 * WV_WIDTH) {         /* eight trips' samples are asked for together: a trip is a round trip to HBM (the input's first touch) */
      i32 a[8], b[8];
#pragma unroll
      for (int u = 0; u < 8; u++) {
         const size_t at = (size_t)(imin(j0 + u * WV_WIDTH + wv_lane(), subframe - 1) + offset) * C;
         if (apcm) { a[u] = apcm[at]; b[u] = C == 2 ? apcm[at + 1] : 0; }
         else { a[u] = shl32((i32)pcm[at], SIG_SHIFT); b[u] = C == 2 ? shl32((i32)pcm[at + 1], SIG_SHIFT) : 0; }
      }
#pragma unroll
      for (int u = 0; u < 8; u++) {
         const int j = j0 + u * WV_WIDTH + wv_lane();
         if (j < subframe) {
            i32 v = a[u];
            if (C == 2) v = half32(add32(v, b[u]));
            if (Fs == 48000) { W->fft[j] = v; if (j & 1) W->hbuf[j >> 1] = v; }
            else if (Fs == 24000) { y[j] = v; if (mirror) W->hbuf[j] = v; }
            else for (int m = 0; m < 3; m++) { const int p = 3 * j + m; W->fft[p] = v; if (p & 1) W->hbuf[p >> 1] = v; }   /* "Don't do this at home!" (:190): x3 by repetition, then 2:1 */
         }
      }
   }
   wv_sync();
   AN2_TOC(34);
   if (Fs == 24000) return 0;
   const int len2 = (Fs == 48000 ? subframe : 3 * subframe) / 2;
   const int lane = wv_lane();
   wv_prio_serial();
   if (lane < 3) {
      const i32 c = lane == 0 ? 19904 /* QCONST16(0.6074371f, 15) */ : 4936 /* QCONST16(0.15063f, 15) */;
      WV_LDS i32 *p = lane == 0 ? W->fft : lane == 1 ? W->fft + 1 : W->hbuf;
      const int stride = lane == 2 ? 1 : 2;
      i32 s = A->downmix_state[lane];
      int k = 0;
      for (; k + 8 <= len2; k += 8) {                            /* eight samples per trip: the LDS reads of a trip are issued back to back, only the chain through s is serial */
         i32 x[8];
#pragma unroll
         for (int u = 0; u < 8; u++) { x[u] = p[(k + u) * stride]; if (lane == 2) x[u] = neg32(x[u]); }
#pragma unroll
         for (int u = 0; u < 8; u++) { const i32 X = mult16_32_q15(c, sub32(x[u], s)); const i32 o = add32(s, X); s = add32(x[u], X); x[u] = o; }
#pragma unroll
         for (int u = 0; u < 8; u++) p[(k + u) * stride] = x[u];
      }
      for (; k < len2; k++) {
         i32 in32 = p[k * stride];
         if (lane == 2) in32 = neg32(in32);
         const i32 X = mult16_32_q15(c, sub32(in32, s));
         p[k * stride] = add32(s, X);
         s = add32(in32, X);
      }
      A->downmix_state[lane] = s;
   }
   wv_prio_normal();
   wv_sync();
   AN2_TOC(35);
   i64 ener = 0;
   FOR_LANES(k, len2) {
      const i32 e = W->fft[2 * k], o = half32(add32(e, W->fft[2 * k + 1]));
      y[k] = o;
      const i32 hp = add32(e, W->hbuf[k]);
      if (mirror) W->hbuf[k] = o;                               /* (this lane has just read the word) */
      ener += ((i64)hp * (i64)hp) >> 8;                         /* (len2 can be up to 480, so we shift by 8 to make it fit) */
   }
   ener = wv_sum64(ener) >> (2 * SIG_SHIFT);
   if (ener > 2147483647) ener = 2147483647;
   wv_sync();
   AN2_TOC(36);
   return Fs == 48000 ? (i32)ener : 0;
}

/* tonality_analysis (src/analysis.c:445): up to 20 ms of input; when 30 ms at 24 kHz have accumulated, one analysis frame -> A->info[write_pos++].
 * gscratch: AN_SCRATCH_WORDS of per-wave HBM.  Every argument is wave-uniform. */
AN_FN void an_tonality_analysis_wave(WV_LDS AnLds *W, OaAnalysis *A, const i16 *pcm, const i32 *apcm, int len, int offset, int C, int Fs, int lsb_depth, i32 *gscratch)
{
   const int lane = wv_lane();
   const int N = 480, N2 = 240;
   AN_TIC();
   int mem_fill = wv_uni(A->mem_fill);
   const int count = wv_uni(A->count), wp = wv_uni(A->write_pos);
   const float hp_acc = A->hp_ener_accum;                     /* (the call's small state is asked for together, ahead of the decimator) */
   if (!wv_uni(A->initialized)) { mem_fill = 240; if (lane == 0) { A->mem_fill = 240; A->initialized = 1; } }
   const float alpha = 1.f / imin(10, 1 + count), alphaE = 1.f / imin(25, 1 + count);
   float alphaE2 = 1.f / imin(100, 1 + count);                 /* noise floor related decay for bandwidth detection: -2.2 dB/second */
   if (count <= 1) alphaE2 = 1;
   if (Fs == 48000) { len /= 2; offset /= 2; }                 /* len and offset are now at 24 kHz */
   else if (Fs == 16000) { len = 3 * len / 2; offset = 3 * offset / 2; }
   const int n1 = imin(len, AN_BUF_SIZE - mem_fill);
   const i32 e1 = an_downmix_resample_wave(W, A, pcm, apcm, A->inmem + mem_fill, n1, offset, C, Fs, true);
   if (mem_fill + len < AN_BUF_SIZE) {                         /* not enough to update the analysis */
      LANE0 { A->hp_ener_accum = hp_acc + (float)e1; A->mem_fill = mem_fill + len; }
      return;
   }
   /* the rest of the input goes through the decimator now (its state carries on from the first part) and waits in HBM until the window has read inmem */
   const int remaining = len - (AN_BUF_SIZE - mem_fill);
   const i32 e2 = an_downmix_resample_wave(W, A, pcm, apcm, gscratch, remaining, offset + AN_BUF_SIZE - mem_fill, C, Fs);
   AN2_TIC();
   OaAnalysisInfo *info = &A->info[wp];
   /* with nothing left over (the frame sizes of a stream that does not change them) the new part [mem_fill, 720) is still in LDS where the decimator put it: the window,
    * the silence check and the move read it there instead of waiting for it to come back from HBM */
   const bool fast = remaining == 0;
#define AN_IN(idx) ((fast && (idx) >= mem_fill) ? W->hbuf[(idx) - mem_fill] : A->inmem[(idx)])
   if (lane == 0) {
      W->hp_ener = hp_acc + (float)e1;
      A->hp_ener_accum = (float)e2;
      A->write_pos = wp + 1 >= AN_DETECT_SIZE ? wp + 1 - AN_DETECT_SIZE : wp + 1;
      A->mem_fill = 240 + remaining;
   }
   AN2_TOC(37);
   i32 mx = 0;
   {  /* window (:523-530) + the bit-reversed, scaled load of opus_fft_c (celt/kiss_fft.c:615); its four reads per point cover all of inmem: is_digital_silence32 (:418) */
      const int16_t *bitrev = ct_fft_bitrev + ct_fft_bitrev_off[0];
      const int scale = ct_fft_misc[1];
#pragma unroll
      for (int t = 0; t < 4; t++) {
         const int i = imin(lane + t * WV_WIDTH, N2 - 1);       /* (the last trip's spare lanes redo point 239: same values to the same words) */
         const float w = an_window[i];
         const i32 x0 = AN_IN(i), x1 = AN_IN(N2 + i), x2 = AN_IN(N - i - 1), x3 = AN_IN(N + N2 - i - 1);
         mx = imax(imax(mx, iabs(x0)), imax(iabs(x1), imax(iabs(x2), iabs(x3))));
         const i32 ar = (i32)(w * x0), ai = (i32)(w * x1);
         const i32 br = (i32)(w * x2), bi = (i32)(w * x3);
         const int ra = bitrev[i], rb = bitrev[N - i - 1];
         W->fft[2 * ra] = SMUL2(ar, scale); W->fft[2 * ra + 1] = SMUL2(ai, scale);
         W->fft[2 * rb] = SMUL2(br, scale); W->fft[2 * rb + 1] = SMUL2(bi, scale);
      }
      if (lane == 0) W->aux[0] = ct_fft_misc[2] - 1;           /* the down-shift budget of opus_fft_c: scale_shift - 1 */
   }
   const int is_silence = wv_max(mx) == 0;
   AN2_TOC(38);
   {  /* OPUS_MOVE(inmem, inmem + 720 - 240, 240) and the second part behind it (what is read, [480, 720), and what is written, [0, 240), do not meet; the second part
       * lands on what was read, so there every lane reads what it moves before anyone writes) */
      i32 keep[4];
      for (int t = 0; t < 4; t++) { const int i = lane + t * WV_WIDTH; keep[t] = i < 240 ? AN_IN(AN_BUF_SIZE - 240 + i) : 0; }
      if (!fast) wv_sync();
      for (int t = 0; t < 4; t++) { const int i = lane + t * WV_WIDTH; if (i < 240) A->inmem[i] = keep[t]; }
      FOR_LANES(i, remaining) A->inmem[240 + i] = gscratch[i];
   }
#undef AN_IN
   wv_sync();
   if (is_silence) {                                           /* on silence, copy the previous analysis (:537) */
      const int prev_pos = wp + 1 - 2 < 0 ? wp + 1 - 2 + AN_DETECT_SIZE : wp + 1 - 2;
      if (lane < (int)(sizeof(OaAnalysisInfo) / 4)) ((i32 *)info)[lane] = ((const i32 *)&A->info[prev_pos])[lane];
      wv_sync();
      return;
   }
   AN2_TOC(39);
   AN_TOC(31);
   fft_forward(W->fft, 0, 1, W->aux);
   AN2_TOC(40);
   const int left = wv_uni(W->aux[0]);
   /* stage the small state of the feature / network tail */
   if (lane < 32) W->mem[lane] = A->mem[lane];
   if (lane < 8) W->cmean[lane] = A->cmean[lane];
   if (lane < 9) W->std[lane] = A->std[lane];
   if (lane < 24) W->rnn[lane] = A->rnn_state[lane];
   {  /* ---- the bins (:570-609): phase of each bin against its history, one lane per bin ---- */
      const float pi4 = (float)(3.14159265358979323846 * 3.14159265358979323846 * 3.14159265358979323846 * 3.14159265358979323846);
      const float to_turns = (float)(.5f / 3.14159265358979323846);
      float r_ton[4], r_ton2[4], r_noisy[4], r_binE[4];
      float E0 = 0;
      for (int t = 0; t < 4; t++) {
         const int i = lane + t * WV_WIDTH;
         r_ton[t] = r_ton2[t] = r_noisy[t] = r_binE[t] = 0;
         if (i == 0) {                                          /* the energy of the very first band is special because of DC (:629) */
            const cpx32 o = c_ld(W->fft, 0, left);
            const float X1r = 2 * (float)o.r, X2r = 2 * (float)o.i;
            E0 = X1r * X1r + X2r * X2r;
         } else if (i < N2) {
            const cpx32 a = c_ld(W->fft, i, left), b = c_ld(W->fft, N - i, left);
            const float X1r = (float)a.r + b.r, X1i = (float)a.i - b.i, X2r = (float)a.i + b.i, X2i = (float)b.r - a.r;
            const float angle = to_turns * an_fast_atan2f(X1i, X1r);
            const float d_angle = angle - A->angle[i];
            const float d2_angle = d_angle - A->d_angle[i];
            const float angle2 = to_turns * an_fast_atan2f(X2i, X2r);
            const float d_angle2 = angle2 - angle;
            const float d2_angle2 = d_angle2 - d_angle;
            float mod1 = d2_angle - (float)an_float2int(d2_angle);
            float noisy = AN_ABS(mod1);
            mod1 *= mod1; mod1 *= mod1;
            float mod2 = d2_angle2 - (float)an_float2int(d2_angle2);
            noisy += AN_ABS(mod2);
            mod2 *= mod2; mod2 *= mod2;
            const float avg_mod = .25f * (A->d2_angle[i] + mod1 + 2 * mod2);
            r_ton[t] = 1.f / (1.f + 40.f * 16.f * pi4 * avg_mod) - .015f;     /* this introduces an extra delay of 2 frames in the detection */
            r_ton2[t] = 1.f / (1.f + 40.f * 16.f * pi4 * mod2) - .015f;       /* no delay on this detection, but it's less reliable */
            r_noisy[t] = noisy;
            A->angle[i] = angle2; A->d_angle[i] = d_angle2; A->d2_angle[i] = mod2;
            r_binE[t] = a.r * (float)a.r + b.r * (float)b.r + a.i * (float)a.i + b.i * (float)b.i;
         }
      }
      AN2_TOC(41);
      wv_sync();                                               /* the spectrum is dead: its bytes now hold the per-bin results */
      for (int t = 0; t < 4; t++) {
         const int i = lane + t * WV_WIDTH;
         if (i < N2) { W->s.tonality[i] = r_ton[t]; W->s.tonality2[i] = r_ton2[t]; W->s.noisiness[i] = r_noisy[t]; W->s.binE[i] = r_binE[t]; }
      }
      if (lane == 0) W->E0 = E0;
      wv_sync();
      float sm[4];
      for (int t = 0; t < 4; t++) {
         const int i = lane + t * WV_WIDTH;
         sm[t] = 0;
         if (i >= 2 && i < N2 - 1) {
            const float tt = AN_MIN(W->s.tonality2[i], AN_MAX(W->s.tonality2[i - 1], W->s.tonality2[i + 1]));
            sm[t] = .9f * AN_MAX(W->s.tonality[i], tt - .1f);
         }
      }
      for (int t = 0; t < 4; t++) { const int i = lane + t * WV_WIDTH; if (i >= 2 && i < N2 - 1) W->s.tonality[i] = sm[t]; }
   }
   wv_sync();
   AN2_TOC(42);
   AN_TOC(32);
   const float scale_ener = (1.f / ((i32)1 << (15 + SIG_SHIFT))) * (1.f / ((i32)1 << (15 + SIG_SHIFT)));   /* SCALE_ENER (:414): the input is +/-2^15 shifted up by SIG_SHIFT */
   const int E_count = wv_uni(A->E_count);
   /* ---- the bands (:643-723): one lane per band, its bins in order ---- */
   if (lane < AN_NB_TBANDS) {
      const int b = lane;
      float E = 0, tE = 0, nE = 0;
      for (int i = an_tbands[b]; i < an_tbands[b + 1]; i++) {
         const float binE = scale_ener * W->s.binE[i];
         E += binE;
         tE += binE * AN_MAX(0, W->s.tonality[i]);
         nE += binE * 2.f * (.5f - W->s.noisiness[i]);
      }
      float lowE = A->lowE[b], highE = A->highE[b];
      if (!count) { lowE = 1e10; highE = -1e10; }
      A->E[E_count][b] = E;
      W->t.t_noisy[b] = nE / (1e-15f + E);
      W->t.t_loud[b] = (float)sqrt((double)(E + 1e-10f));
      const float logE = (float)log((double)(E + 1e-10f));
      W->t.logE[b] = logE;
      W->t.band_log2[b + 1] = .5f * 1.442695f * (float)log((double)(E + 1e-10f));
      A->logE[E_count][b] = logE;
      if (count == 0) highE = lowE = logE;
      if (highE > lowE + 7.5) { if (highE - logE > logE - lowE) highE -= .01f; else lowE += .01f; }
      if (logE > highE) { highE = logE; lowE = AN_MAX(highE - 15, lowE); }
      else if (logE < lowE) { lowE = logE; highE = AN_MIN(lowE + 15, highE); }
      A->lowE[b] = lowE; A->highE[b] = highE;
      W->t.t_relE[b] = (logE - lowE) / (1e-5f + (highE - lowE));
      float L1 = 0, L2 = 0;
      for (int i = 0; i < AN_NB_FRAMES; i++) { const float Ei = i == E_count ? E : A->E[i][b]; L1 += (float)sqrt((double)(Ei)); L2 += Ei; }
      float stationarity = AN_MIN(0.99f, L1 / (float)sqrt((double)(1e-15 + AN_NB_FRAMES * L2)));
      stationarity *= stationarity;
      stationarity *= stationarity;
      W->t.t_stat[b] = stationarity;
      const float bt = AN_MAX(tE / (1e-15f + E), stationarity * A->prev_band_tonality[b]);
      W->t.band_tonality[b] = bt;
      A->prev_band_tonality[b] = bt;
      /* the band's part of the bandwidth detection (:788-811): its energy again (scaled after the sum this time), the decaying maximum */
      float E2 = 0;
      for (int i = an_tbands[b]; i < an_tbands[b + 1]; i++) E2 += W->s.binE[i];
      E2 = scale_ener * E2;
      const float meanE = AN_MAX((1 - alphaE2) * A->meanE[b], E2);
      A->meanE[b] = meanE;
      W->t.E2[b] = E2; W->t.Em[b] = AN_MAX(E2, meanE);
   } else if (lane == AN_NB_TBANDS) {
      float E = W->E0;
      for (int i = 1; i < 4; i++) E += W->s.binE[i];
      E = scale_ener * E;
      W->t.band_log2[0] = .5f * 1.442695f * (float)log((double)(E + 1e-10f));
   }
   wv_sync();
   AN2_TOC(43);
   {  /* spectral variability (:755-775): the 8 x 8 distances between the stored log spectra, one lane per pair */
      const int i = lane >> 3, j = lane & 7;
      float dist = 0, gi[AN_NB_TBANDS], gj[AN_NB_TBANDS];
#pragma unroll
      for (int k = 0; k < AN_NB_TBANDS; k++) { gi[k] = A->logE[i][k]; gj[k] = A->logE[j][k]; }      /* (the 36 stored values of the pair are asked for together; the sum keeps the reference's order) */
#pragma unroll
      for (int k = 0; k < AN_NB_TBANDS; k++) {
         const float li = i == E_count ? W->t.logE[k] : gi[k], lj = j == E_count ? W->t.logE[k] : gj[k];
         const float tmp = li - lj;
         dist += tmp * tmp;
      }
      W->t.dist[lane] = dist;
      wv_sync();
      if (lane < AN_NB_FRAMES) {
         float mindist = 1e15f;
         for (int jj = 0; jj < AN_NB_FRAMES; jj++) if (jj != lane) mindist = AN_MIN(mindist, W->t.dist[lane * 8 + jj]);
         W->t.mindist[lane] = mindist;
      }
      /* the cepstral sums (:861-875): BFCC on lanes 8..15, midE on lanes 16..23 */
      if (lane >= 8 && lane < 24) {                               /* (one body for both: the table row and the band values in flight together) */
         const int c = lane & 7; const bool mid = lane >= 16;
         float tb[16], hv[16], lv[16], sum = 0;
#pragma unroll
         for (int b = 0; b < 16; b++) { tb[b] = an_dct_table[c * 16 + b]; hv[b] = A->highE[b]; lv[b] = A->lowE[b]; }
#pragma unroll
         for (int b = 0; b < 16; b++) sum += mid ? tb[b] * .5f * (hv[b] + lv[b]) : tb[b] * W->t.logE[b];
         if (mid) W->t.midE[c] = sum; else W->t.BFCC[c] = sum;
      }
   }
   AN2_TOC(44);
   /* ---- what chains across the bands: lane 0 ---- */
   LANE0 {
      float frame_tonality = 0, max_frame_tonality = 0, frame_noisiness = 0, frame_stationarity = 0, relativeE = 0, frame_loudness = 0, slope = 0;
      for (int b = 0; b < AN_NB_TBANDS; b++) {
         frame_noisiness += W->t.t_noisy[b];
         frame_loudness += W->t.t_loud[b];
         relativeE += W->t.t_relE[b];
         frame_stationarity += W->t.t_stat[b];
         frame_tonality += W->t.band_tonality[b];
         if (b >= AN_NB_TBANDS - AN_NB_TONAL_SKIP_BANDS) frame_tonality -= W->t.band_tonality[b - AN_NB_TBANDS + AN_NB_TONAL_SKIP_BANDS];
         max_frame_tonality = AN_MAX(max_frame_tonality, (1.f + .03f * (b - AN_NB_TBANDS)) * frame_tonality);
         slope += W->t.band_tonality[b] * (b - 8);
      }
      /* leakage (:725-753) */
      const float LEAKAGE_OFFSET = 2.5f, LEAKAGE_SLOPE = 2.f;
      W->t.leak_from[0] = W->t.band_log2[0];
      W->t.leak_to[0] = W->t.band_log2[0] - LEAKAGE_OFFSET;
      for (int b = 1; b < AN_NB_TBANDS + 1; b++) {
         const float leak_slope = LEAKAGE_SLOPE * (an_tbands[b] - an_tbands[b - 1]) / 4;
         W->t.leak_from[b] = AN_MIN(W->t.leak_from[b - 1] + leak_slope, W->t.band_log2[b]);
         W->t.leak_to[b] = AN_MAX(W->t.leak_to[b - 1] - leak_slope, W->t.band_log2[b] - LEAKAGE_OFFSET);
      }
      for (int b = AN_NB_TBANDS - 2; b >= 0; b--) {
         const float leak_slope = LEAKAGE_SLOPE * (an_tbands[b + 1] - an_tbands[b]) / 4;
         W->t.leak_from[b] = AN_MIN(W->t.leak_from[b + 1] + leak_slope, W->t.leak_from[b]);
         W->t.leak_to[b] = AN_MAX(W->t.leak_to[b + 1] - leak_slope, W->t.leak_to[b]);
      }
      float spec_variability = 0;
      for (int i = 0; i < AN_NB_FRAMES; i++) spec_variability += W->t.mindist[i];
      spec_variability = (float)sqrt((double)(spec_variability / AN_NB_FRAMES / AN_NB_TBANDS));
      /* bandwidth detection (:776-853) */
      float bandwidth_mask = 0, maxE = 0, below_max_pitch = 0, above_max_pitch = 0;
      int bandwidth = 0;
      float noise_floor = 5.7e-4f / (1 << (imax(0, lsb_depth - 8)));
      noise_floor *= noise_floor;
      const int prev_bandwidth = A->prev_bandwidth;
      u32 masked = 0;                                             /* bit b: band b is masked (is_masked[], :810) */
      for (int b = 0; b < AN_NB_TBANDS; b++) {
         const float E = W->t.E2[b], Em = W->t.Em[b];
         const int band_start = an_tbands[b], band_end = an_tbands[b + 1];
         maxE = AN_MAX(maxE, E);
         if (band_start < 64) below_max_pitch += E; else above_max_pitch += E;
         /* "active" only if less than 90 dB below the peak band and above the PCM quantization noise floor; b+1 because the first CELT band isn't included in tbands[] */
         if (E * 1e9f > maxE && (Em > 3 * noise_floor * (band_end - band_start) || E > noise_floor * (band_end - band_start))) bandwidth = b + 1;
         if (E < (prev_bandwidth >= b + 1 ? .01f : .05f) * bandwidth_mask) masked |= 1u << b;
         bandwidth_mask = AN_MAX(.05f * bandwidth_mask, E);        /* a simple follower with 13 dB/Bark slope for spreading function */
      }
      if (Fs == 48000) {                                            /* the last two bands: only the energy above 12 kHz, from the decimator's high-pass branch */
         float E = W->hp_ener * (1.f / (60 * 60));
         const float noise_ratio = prev_bandwidth == 20 ? 10.f : 30.f;
         E *= 256.f * (1.f / Q15ONE) * (1.f / Q15ONE);             /* silk_resampler_down2_hp() shifted right by an extra 8 bits */
         above_max_pitch += E;
         const float meanE = AN_MAX((1 - alphaE2) * A->meanE[AN_NB_TBANDS], E);
         A->meanE[AN_NB_TBANDS] = meanE;
         const float Em = AN_MAX(E, meanE);
         if (Em > 3 * noise_ratio * noise_floor * 160 || E > noise_ratio * noise_floor * 160) bandwidth = 20;
         if (E < (prev_bandwidth == 20 ? .01f : .05f) * bandwidth_mask) masked |= 1u << AN_NB_TBANDS;
      }
      W->max_pitch_ratio = above_max_pitch > below_max_pitch ? below_max_pitch / above_max_pitch : 1;
      /* resampling aliasing can create a small amount of energy in the first band being cut: if the last band is masked, it is not included */
      if (bandwidth == 20 && (masked >> AN_NB_TBANDS & 1)) bandwidth -= 2;
      else if (bandwidth > 0 && bandwidth <= AN_NB_TBANDS && (masked >> (bandwidth - 1) & 1)) bandwidth--;
      if (count <= 2) bandwidth = 20;
      frame_loudness = 20 * (float)log10((double)(frame_loudness));
      const float Etracker = AN_MAX(A->Etracker - .003f, frame_loudness);
      A->Etracker = Etracker;
      float lowECount = A->lowECount;
      lowECount *= (1 - alphaE);
      if (frame_loudness < Etracker - 30) lowECount += alphaE;
      A->lowECount = lowECount;
      frame_stationarity /= AN_NB_TBANDS;
      relativeE /= AN_NB_TBANDS;
      if (count < 10) relativeE = .5f;
      frame_noisiness /= AN_NB_TBANDS;
      const float activity = frame_noisiness + (1 - frame_noisiness) * relativeE;
      frame_tonality = (max_frame_tonality / (AN_NB_TBANDS - AN_NB_TONAL_SKIP_BANDS));
      frame_tonality = AN_MAX(frame_tonality, A->prev_tonality * .8f);
      A->prev_tonality = frame_tonality;
      slope /= 8 * 8;
      A->E_count = (E_count + 1) % AN_NB_FRAMES;
      const int count1 = imin(count + 1, AN_COUNT_MAX);
      A->count = count1;
      /* features (:893-934) */
      WV_LDS float *features = W->t.features, *BFCC = W->t.BFCC, *mem = W->mem, *cmean = W->cmean, *std = W->std;
      for (int i = 0; i < 4; i++) features[i] = -0.12299f * (BFCC[i] + mem[i + 24]) + 0.49195f * (mem[i] + mem[i + 16]) + 0.69693f * mem[i + 8] - 1.4349f * cmean[i];
      for (int i = 0; i < 4; i++) cmean[i] = (1 - alpha) * cmean[i] + alpha * BFCC[i];
      for (int i = 0; i < 4; i++) features[4 + i] = 0.63246f * (BFCC[i] - mem[i + 24]) + 0.31623f * (mem[i] - mem[i + 16]);
      for (int i = 0; i < 3; i++) features[8 + i] = 0.53452f * (BFCC[i] + mem[i + 24]) - 0.26726f * (mem[i] + mem[i + 16]) - 0.53452f * mem[i + 8];
      if (count1 > 5) for (int i = 0; i < 9; i++) std[i] = (1 - alpha) * std[i] + alpha * features[i] * features[i];
      for (int i = 0; i < 4; i++) features[i] = BFCC[i] - W->t.midE[i];
      for (int i = 0; i < 8; i++) { mem[i + 24] = mem[i + 16]; mem[i + 16] = mem[i + 8]; mem[i + 8] = mem[i]; mem[i] = BFCC[i]; }
      for (int i = 0; i < 9; i++) features[11 + i] = (float)sqrt((double)(std[i])) - an_std_feature_bias[i];
      features[18] = spec_variability - 0.78f;
      features[20] = frame_tonality - 0.154723f;
      features[21] = activity - 0.724643f;
      features[22] = frame_stationarity - 0.743717f;
      features[23] = slope + 0.069216f;
      features[24] = lowECount - 0.067930f;
      W->frame_tonality = frame_tonality; W->tonality_slope = slope; W->activity = activity; W->frame_noisiness = frame_noisiness; W->bandwidth = bandwidth;
      A->prev_bandwidth = bandwidth;
   }
   AN2_TOC(45);
   /* leak_boost (:744-753), one lane per band, straight into the info record */
   if (lane < AN_NB_TBANDS + 1) {
      const int b = lane;
      const float boost = AN_MAX(0, W->t.leak_to[b] - W->t.band_log2[b]) + AN_MAX(0, W->t.band_log2[b] - (W->t.leak_from[b] + 2.5f));
      info->leak_boost[b] = (u8)imin(255, (int)floor((double)(.5 + 64.f * boost)));
   }
   /* ---- the network (src/mlp.c): dense 25 -> 32 (tanh), GRU 32 -> 24, dense 24 -> 2 (sigmoid); one lane per neuron, inputs in order ---- */
   const float WEIGHTS_SCALE = 1.f / 128;
   if (lane < 32) {
      float o = an_l0_bias[lane];
      for (int j = 0; j < 25; j++) o += an_l0_weights[j * 32 + lane] * W->t.features[j];
      o *= WEIGHTS_SCALE;
      W->t.layer_out[lane] = an_tansig(o);
   }
   wv_sync();
   if (lane < 24) {
      const int Ng = 24, stride = 72;
      float z = an_l1_bias[lane], r = an_l1_bias[Ng + lane];
      for (int j = 0; j < 32; j++) z += an_l1_weights[j * stride + lane] * W->t.layer_out[j];
      for (int j = 0; j < Ng; j++) z += an_l1_recur_weights[j * stride + lane] * W->rnn[j];
      z = an_sigmoid(WEIGHTS_SCALE * z);
      for (int j = 0; j < 32; j++) r += an_l1_weights[Ng + j * stride + lane] * W->t.layer_out[j];
      for (int j = 0; j < Ng; j++) r += an_l1_recur_weights[Ng + j * stride + lane] * W->rnn[j];
      r = an_sigmoid(WEIGHTS_SCALE * r);
      W->t.z[lane] = z;
      W->t.tmp[lane] = W->rnn[lane] * r;
   }
   wv_sync();
   float hnew = 0;
   if (lane < 24) {
      const int Ng = 24, stride = 72;
      float h = an_l1_bias[2 * Ng + lane];
      for (int j = 0; j < 32; j++) h += an_l1_weights[2 * Ng + j * stride + lane] * W->t.layer_out[j];
      for (int j = 0; j < Ng; j++) h += an_l1_recur_weights[2 * Ng + j * stride + lane] * W->t.tmp[j];
      const float z = W->t.z[lane];
      hnew = z * W->rnn[lane] + (1 - z) * an_tansig(WEIGHTS_SCALE * h);
   }
   wv_sync();
   if (lane < 24) { W->rnn[lane] = hnew; A->rnn_state[lane] = hnew; }
   wv_sync();
   if (lane < 2) {
      float o = an_l2_bias[lane];
      for (int j = 0; j < 24; j++) o += an_l2_weights[j * 2 + lane] * W->rnn[j];
      o *= WEIGHTS_SCALE;
      W->t.probs[lane] = an_sigmoid(o);
   }
   /* write the staged state back */
   if (lane < 32) A->mem[lane] = W->mem[lane];
   if (lane < 8) A->cmean[lane] = W->cmean[lane];
   if (lane < 9) A->std[lane] = W->std[lane];
   LANE0 {
      info->activity_probability = W->t.probs[1];                /* probability of speech or music vs noise */
      info->music_prob = W->t.probs[0];
      info->tonality = W->frame_tonality; info->tonality_slope = W->tonality_slope; info->activity = W->activity; info->noisiness = W->frame_noisiness;
      info->bandwidth = W->bandwidth; info->max_pitch_ratio = W->max_pitch_ratio;
      info->valid = 1;
   }
   AN2_TOC(46);
   AN_TOC(33);
}

/* tonality_get_info (src/analysis.c:232): the info the encoder uses for a frame of `len` samples -- the stored one at the read position, with tonality and bandwidth
 * widened over the neighbours and the music probability turned into switching thresholds over the look-ahead.  Lane 0 on an LDS copy of the ring's four fields. */
AN_FN void an_get_info_wave(WV_LDS AnLds *W, OaAnalysis *A, OaAnalysisInfo *info_out, int len, int Fs)
{
   const int lane = wv_lane();
   AN2_TIC();
   wv_sync();
   FOR_LANES(i, AN_DETECT_SIZE) {
      W->ring.tonality[i] = A->info[i].tonality; W->ring.music_prob[i] = A->info[i].music_prob;
      W->ring.activity_probability[i] = A->info[i].activity_probability; W->ring.bandwidth[i] = A->info[i].bandwidth;
   }
   wv_sync();
   const int write_pos = wv_uni(A->write_pos);
   int pos = wv_uni(A->read_pos);
   int curr_lookahead = write_pos - pos;
   if (curr_lookahead < 0) curr_lookahead += AN_DETECT_SIZE;
   {
      int read_subframe = wv_uni(A->read_subframe) + len / (Fs / 400), read_pos = pos;
      while (read_subframe >= 8) { read_subframe -= 8; read_pos++; }
      if (read_pos >= AN_DETECT_SIZE) read_pos -= AN_DETECT_SIZE;
      LANE0 { A->read_subframe = read_subframe; A->read_pos = read_pos; }
   }
   if (len > Fs / 50 && pos != write_pos) { pos++; if (pos == AN_DETECT_SIZE) pos = 0; }   /* on long frames, look at the second analysis window rather than the first */
   if (pos == write_pos) pos--;
   if (pos < 0) pos = AN_DETECT_SIZE - 1;
   const int pos0 = pos;
   if (lane < (int)(sizeof(OaAnalysisInfo) / 4)) ((i32 *)info_out)[lane] = ((const i32 *)&A->info[pos0])[lane];
   wv_sync();
   AN2_TOC(47);
   if (!wv_uni(A->info[pos0].valid)) return;
   LANE0 {
      const WV_LDS float *ton = W->ring.tonality, *mp = W->ring.music_prob, *ap = W->ring.activity_probability;
      const WV_LDS i32 *bw = W->ring.bandwidth;
      float tonality_max = ton[pos0], tonality_avg = ton[pos0];
      int tonality_count = 1, bandwidth = bw[pos0];
      int bandwidth_span = 6;                                    /* look at the neighbouring frames and pick largest bandwidth found (to be safe) */
      for (int i = 0; i < 3; i++) {                              /* if possible, look ahead for a tone to compensate for the delay in the tone detector */
         pos++;
         if (pos == AN_DETECT_SIZE) pos = 0;
         if (pos == write_pos) break;
         tonality_max = AN_MAX(tonality_max, ton[pos]);
         tonality_avg += ton[pos];
         tonality_count++;
         bandwidth = imax(bandwidth, bw[pos]);
         bandwidth_span--;
      }
      pos = pos0;
      for (int i = 0; i < bandwidth_span; i++) {                 /* look back in time to see if any has a wider bandwidth than the current frame */
         pos--;
         if (pos < 0) pos = AN_DETECT_SIZE - 1;
         if (pos == write_pos) break;
         bandwidth = imax(bandwidth, bw[pos]);
      }
      info_out->bandwidth = bandwidth;
      info_out->tonality = AN_MAX(tonality_avg / tonality_count, tonality_max - .2f);
      int mpos = pos0, vpos = pos0;
      if (curr_lookahead > 15) {                                 /* enough look-ahead: compensate for the ~5-frame delay in the music prob and ~1 frame delay in the VAD prob */
         mpos += 5; if (mpos >= AN_DETECT_SIZE) mpos -= AN_DETECT_SIZE;
         vpos += 1; if (vpos >= AN_DETECT_SIZE) vpos -= AN_DETECT_SIZE;
      }
      /* minimise the "badness" of the transition (:311-345) */
      const float TRANSITION_PENALTY = 10;
      float prob_min = 1.f, prob_max = 0.f;
      const float vad_prob = ap[vpos];
      float prob_count = AN_MAX(.1f, vad_prob);
      float prob_avg = AN_MAX(.1f, vad_prob) * mp[mpos];
      while (1) {
         mpos++; if (mpos == AN_DETECT_SIZE) mpos = 0;
         if (mpos == write_pos) break;
         vpos++; if (vpos == AN_DETECT_SIZE) vpos = 0;
         if (vpos == write_pos) break;
         const float pos_vad = ap[vpos];
         prob_min = AN_MIN((prob_avg - TRANSITION_PENALTY * (vad_prob - pos_vad)) / prob_count, prob_min);
         prob_max = AN_MAX((prob_avg + TRANSITION_PENALTY * (vad_prob - pos_vad)) / prob_count, prob_max);
         prob_count += AN_MAX(.1f, pos_vad);
         prob_avg += AN_MAX(.1f, pos_vad) * mp[mpos];
      }
      info_out->music_prob = prob_avg / prob_count;
      prob_min = AN_MIN(prob_avg / prob_count, prob_min);
      prob_max = AN_MAX(prob_avg / prob_count, prob_max);
      prob_min = AN_MAX(prob_min, 0.f);
      prob_max = AN_MIN(prob_max, 1.f);
      if (curr_lookahead < 10) {                                 /* not enough look-ahead: do our best to make a decent decision */
         float pmin = prob_min, pmax = prob_max;
         pos = pos0;
         const int back = imin(A->count - 1, 15);
         for (int i = 0; i < back; i++) {                        /* look for min/max in the past */
            pos--;
            if (pos < 0) pos = AN_DETECT_SIZE - 1;
            pmin = AN_MIN(pmin, mp[pos]);
            pmax = AN_MAX(pmax, mp[pos]);
         }
         pmin = AN_MAX(0.f, pmin - .1f * vad_prob);               /* bias against switching on active audio */
         pmax = AN_MIN(1.f, pmax + .1f * vad_prob);
         prob_min += (1.f - .1f * curr_lookahead) * (pmin - prob_min);
         prob_max += (1.f - .1f * curr_lookahead) * (pmax - prob_max);
      }
      info_out->music_prob_min = prob_min;
      info_out->music_prob_max = prob_max;
   }
   AN2_TOC(48);
}

/* run_analysis (src/analysis.c:954): the call's input through tonality_analysis in 20 ms steps, then the info for the call's first coded frame */
WV_DEV void an_run_analysis_wave(WV_LDS AnLds *W, OaAnalysis *A, const i16 *pcm, const i32 *apcm, int analysis_frame_size, int frame_size, int C, int Fs, int lsb_depth,
      i32 *gscratch, OaAnalysisInfo *info_out)
{
   analysis_frame_size -= analysis_frame_size & 1;
   analysis_frame_size = imin((AN_DETECT_SIZE - 5) * Fs / 50, analysis_frame_size);      /* avoid overflow / wrap-around of the analysis buffer */
   const int analysis_offset = wv_uni(A->analysis_offset);
   int pcm_len = analysis_frame_size - analysis_offset, offset = analysis_offset;
   while (pcm_len > 0) {
      an_tonality_analysis_wave(W, A, pcm, apcm, imin(Fs / 50, pcm_len), offset, C, Fs, lsb_depth, gscratch);
      offset += Fs / 50;
      pcm_len -= Fs / 50;
   }
   LANE0 A->analysis_offset = analysis_frame_size - frame_size;
   an_get_info_wave(W, A, info_out, frame_size, Fs);
}

/* ---- the float expressions of the analysis' consumers (kept here, under the same no-contraction rule) ---- */
/* src/opus_encoder.c:1279-1306: voice_ratio and detected bandwidth from the frame's info; prev_mode: 0, MODE_CELT_ONLY or other */
WV_DEV int an_voice_ratio(const OaAnalysisInfo *a, int prev_mode)
{
   const float prob = prev_mode == 0 ? a->music_prob : prev_mode == 1002 ? a->music_prob_max : a->music_prob_min;
   return (int)floor((double)(.5 + 100 * (1 - prob)));
}
WV_DEV int an_detected_bandwidth(int analysis_bandwidth)
{
   return analysis_bandwidth <= 12 ? 1101 : analysis_bandwidth <= 14 ? 1102 : analysis_bandwidth <= 16 ? 1103 : analysis_bandwidth <= 18 ? 1104 : 1105;
}
#define AN_DTX_ACTIVITY_THRESHOLD 0.1f                          /* silk/define.h:54 */
#define AN_PSEUDO_SNR_THRESHOLD 316.23f                         /* src/opus_encoder.c:68: 10^(25/10) */
/* src/opus_encoder.c:1916-1923: is the frame active (for the generalised DTX)?  noise_energy is only consulted for a low activity probability */
WV_DEV int an_activity_prob_active(const OaAnalysisInfo *a) { return a->activity_probability >= AN_DTX_ACTIVITY_THRESHOLD; }
WV_DEV int an_activity_prob_above(const OaAnalysisInfo *a) { return a->activity_probability > AN_DTX_ACTIVITY_THRESHOLD; }     /* :1312 */
WV_DEV int an_loud_noise_active(i32 peak_signal_energy, i32 noise_energy) { return peak_signal_energy < (AN_PSEUDO_SNR_THRESHOLD * noise_energy); }
/* celt/celt_encoder.c:935-939 (alloc_trim_analysis): trim in Q8 */
WV_DEV i32 an_trim_tonality_slope(i32 trim, const OaAnalysisInfo *a)
{
   const i32 t = (i32)(i16)(QC16(2.f, 8) * (a->tonality_slope + .05f));
   return trim - imax(-QC16(2.f, 8), imin(QC16(2.f, 8), t));
}
/* celt/celt_encoder.c:1494 (run_prefilter) */
WV_DEV i32 an_scale_pitch_gain(i32 gain1, const OaAnalysisInfo *a) { return (i32)(i16)((i16)gain1 * a->max_pitch_ratio); }
/* celt/celt_encoder.c:2043: the pitch pre-filter is only kept on a tonal enough frame */
WV_DEV int an_tonal_enough_for_prefilter(const OaAnalysisInfo *a) { return !a->valid || a->tonality > .3; }
/* celt/celt_encoder.c:1632-1633 and :1658-1670 (compute_vbr) */
WV_DEV i32 an_vbr_activity(i32 target, i32 coded_bins, const OaAnalysisInfo *a)
{
   if (a->valid && a->activity < .4) target -= (i32)((coded_bins << BITRES) * (.4f - a->activity));
   return target;
}
WV_DEV i32 an_vbr_tonality(i32 target, i32 coded_bins, int pitch_change, const OaAnalysisInfo *a)
{
   const float tonal = AN_MAX(0.f, a->tonality - .15f) - 0.12f;
   i32 tonal_target = target + (i32)((coded_bins << BITRES) * 1.2f * tonal);
   if (pitch_change) tonal_target += (i32)((coded_bins << BITRES) * .8f);
   return tonal_target;
}
#ifdef __clang__
#pragma clang fp contract(fast)
#endif
#endif

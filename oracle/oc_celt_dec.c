/* oc_celt_dec.c — CELT frame decoder, oracle restatement (fixed-point, non-QEXT, non-custom) of
 * celt/celt_decoder.c: celt_decoder_init :224/:244, OPUS_RESET_STATE :1794, deemphasis :318, celt_synthesis :413,
 * tf_decode :513, celt_plc_pitch_search :552, prefilter_and_fold :576, celt_decode_lost :679 (noise and pitch PLC; no deep PLC),
 * celt_decode_with_ec_dred :1104; celt/celt_lpc.c: celt_fir :140, celt_iir :186;
 * celt/bands.c: denormalise_bands :187, anti_collapse :259.  TEST INFRASTRUCTURE, never shipped. */
#include "oc_celt_dec.h"
#include <string.h>
#include <stdlib.h>

extern void (*oc_dump_hook)(const char *tag, const void *p, int nbytes);
#define OC_DUMP(tag, p, n) do { if (oc_dump_hook) oc_dump_hook(tag, p, n); } while (0)

static const u8 trim_icdf[11] = {126, 124, 119, 109, 87, 41, 19, 9, 4, 2, 0};
static const u8 spread_icdf[4] = {25, 23, 2, 0};
static const u8 tapset_icdf[3] = {2, 1, 0};
static const signed char tf_select_table[4][8] = {
   {0, -1, 0, -1, 0, -1, 0, -1}, {0, -1, 0, -2, 1, 0, 1, -1}, {0, -2, 0, -3, 2, 0, 1, -1}, {0, -2, 0, -3, 3, 0, 1, -1}};
#define SPREAD_NORMAL 2

void oc_celt_dec_reset(oc_celt_dec *st)
{
   memset(&st->rng, 0, sizeof(*st) - ((char *)&st->rng - (char *)st));
   for (int i = 0; i < 2 * NB_EBANDS; i++) st->oldLogE[i] = st->oldLogE2[i] = -GC(28.f);
   st->skip_plc = 1;
   st->last_frame_type = 0;   /* FRAME_NONE */
}
void oc_celt_dec_init(oc_celt_dec *st, int channels)
{
   memset(st, 0, sizeof(*st));
   st->stream_channels = st->channels = channels;
   st->start = 0; st->end = NB_EBANDS;
   st->disable_inv = channels == 1;
   oc_celt_dec_reset(st);
}

/* tf_decode, celt_decoder.c:513 */
static void tf_decode(int start, int end, int isTransient, int *tf_res, int LM, oc_ec *dec)
{
   int curr = 0, tf_select = 0, tf_changed = 0;
   u32 budget = dec->storage * 8, tell = oc_ec_tell(dec);
   int logp = isTransient ? 2 : 4;
   int tf_select_rsv = LM > 0 && tell + logp + 1 <= budget;
   budget -= tf_select_rsv;
   for (int i = start; i < end; i++) {
      if (tell + logp <= budget) {
         curr ^= oc_ec_dec_bit_logp(dec, logp);
         tell = oc_ec_tell(dec);
         tf_changed |= curr;
      }
      tf_res[i] = curr;
      logp = isTransient ? 4 : 5;
   }
   if (tf_select_rsv && tf_select_table[LM][4 * isTransient + 0 + tf_changed] != tf_select_table[LM][4 * isTransient + 2 + tf_changed])
      tf_select = oc_ec_dec_bit_logp(dec, 1);
   for (int i = start; i < end; i++) tf_res[i] = tf_select_table[LM][4 * isTransient + 2 * tf_select + tf_res[i]];
}

/* anti_collapse, bands.c:259 (decoder call: encode = 0) */
static void anti_collapse(i32 *X_, const u8 *collapse_masks, int LM, int C, int size, int start, int end, const i32 *logE,
      const i32 *prev1logE, const i32 *prev2logE, const int *pulses, u32 seed)
{
   for (int i = start; i < end; i++) {
      int N0 = oc_eBands[i + 1] - oc_eBands[i];
      int depth = (int)((u32)(1 + pulses[i]) / (u32)N0) >> LM;
      i32 thresh32 = oc_exp2(-shl16(depth, 10 - BITRES)) >> 1;
      i16 thresh = (i16)mult16_32_q15(QC16(0.5f, 15), imin(32767, thresh32));
      i32 t = N0 << LM;
      int shift = celt_ilog2(t) >> 1;
      t = shl32(t, (7 - shift) << 1);
      i16 sqrt_1 = oc_rsqrt_norm(t);
      for (int c = 0; c < C; c++) {
         i32 prev1 = prev1logE[c * NB_EBANDS + i], prev2 = prev2logE[c * NB_EBANDS + i];
         int renormalize = 0;
         if (C == 1) { prev1 = imax(prev1, prev1logE[NB_EBANDS + i]); prev2 = imax(prev2, prev2logE[NB_EBANDS + i]); }
         i32 Ediff = logE[c * NB_EBANDS + i] - imin(prev1, prev2);
         Ediff = imax(0, Ediff);
         i32 r;
         if (Ediff < GC(16.f)) { i32 r32 = oc_exp2_db(-Ediff) >> 1; r = 2 * imin(16383, r32); }
         else r = 0;
         if (LM == 3) r = mult16_16_q14(23170, imin(23169, r));
         r = (i16)(imin(thresh, r)) >> 1;
         r = vshr32(mult16_16_q15(sqrt_1, r), shift + 14 - NORM_SHIFT);
         i32 *X = X_ + c * size + (oc_eBands[i] << LM);
         for (int k = 0; k < 1 << LM; k++) {
            if (!(collapse_masks[i * C + c] & 1 << k)) {
               for (int j = 0; j < N0; j++) {
                  seed = 1664525u * seed + 1013904223u;
                  X[(j << LM) + k] = (seed & 0x8000 ? r : -r);
               }
               renormalize = 1;
            }
         }
         if (renormalize) oc_renormalise_vector(X, N0 << LM, Q31ONE);
      }
   }
}

/* denormalise_bands, bands.c:187 (downsample == 1) */
static void denormalise_bands(const i32 *X, i32 *freq, const i32 *bandLogE, int start, int end, int M, int silence)
{
   const int N = M * 120;
   int bound = M * oc_eBands[end];
   if (silence) { bound = 0; start = end = 0; }
   i32 *f = freq;
   const i32 *x = X + M * oc_eBands[start];
   if (start != 0) { for (int i = 0; i < M * oc_eBands[start]; i++) *f++ = 0; }
   else f += M * oc_eBands[start];
   for (int i = start; i < end; i++) {
      int j = M * oc_eBands[i], band_end = M * oc_eBands[i + 1];
      i32 lg = add32(bandLogE[i], shl32((i32)oc_eMeans[i], DB_SHIFT - 4));
      int shift = 17 - (lg >> DB_SHIFT);
      i32 g;
      if (shift >= 31) { shift = 0; g = 0; }
      else g = shl32(oc_exp2_db_frac(lg & ((1 << DB_SHIFT) - 1)), 2);
      if (shift < 0) { g = 2147483647; shift = 0; }
      do { *f++ = pshr32(mult32_32_q31(shl32(*x, 30 - NORM_SHIFT), g), shift); x++; } while (++j < band_end);
   }
   memset(&freq[bound], 0, (N - bound) * sizeof(i32));
}

/* celt_synthesis, celt_decoder.c:413 (downsample == 1) */
static void celt_synthesis(i32 *X, i32 *out_syn[], const i32 *oldBandE, int start, int effEnd, int C, int CC, int isTransient, int LM, int silence)
{
   const int overlap = OVERLAP, N = 120 << LM, M = 1 << LM;
   i32 freq[960];
   int B, NB, shift;
   if (isTransient) { B = M; NB = 120; shift = 3; }
   else { B = 1; NB = 120 << LM; shift = 3 - LM; }
   if (CC == 2 && C == 1) {
      denormalise_bands(X, freq, oldBandE, start, effEnd, M, silence);
      i32 *freq2 = out_syn[1] + overlap / 2;
      memcpy(freq2, freq, N * sizeof(i32));
      for (int b = 0; b < B; b++) oc_mdct_backward(&freq2[b], out_syn[0] + NB * b, shift, B);
      for (int b = 0; b < B; b++) oc_mdct_backward(&freq[b], out_syn[1] + NB * b, shift, B);
   } else if (CC == 1 && C == 2) {
      i32 *freq2 = out_syn[0] + overlap / 2;
      denormalise_bands(X, freq, oldBandE, start, effEnd, M, silence);
      denormalise_bands(X + N, freq2, oldBandE + NB_EBANDS, start, effEnd, M, silence);
      for (int i = 0; i < N; i++) freq[i] = add32(half32(freq[i]), half32(freq2[i]));
      for (int b = 0; b < B; b++) oc_mdct_backward(&freq[b], out_syn[0] + NB * b, shift, B);
   } else {
      for (int c = 0; c < CC; c++) {
         denormalise_bands(X + c * N, freq, oldBandE + c * NB_EBANDS, start, effEnd, M, silence);
         for (int b = 0; b < B; b++) oc_mdct_backward(&freq[b], out_syn[c] + NB * b, shift, B);
      }
   }
   for (int c = 0; c < CC; c++) for (int i = 0; i < N; i++) out_syn[c][i] = saturate(out_syn[c][i], SIG_SAT);
}

/* deemphasis, celt_decoder.c:318 (downsample == 1, no accumulation; the stereo fast path :288 computes the same values) */
static void deemphasis(i32 *in[], i16 *pcm, int N, int C, i32 *mem)
{
   const i16 coef0 = 27853;       /* mode->preemph[0], QCONST16(0.8500061035f, 15) */
   for (int c = 0; c < C; c++) {
      i32 m = mem[c];
      const i32 *x = in[c];
      for (int j = 0; j < N; j++) {
         i32 tmp = saturate(x[j] + m, SIG_SAT);
         m = mult16_32_q15(coef0, tmp);
         pcm[j * C + c] = sig2word16(tmp);
      }
      mem[c] = m;
   }
}

#define PLC_PITCH_LAG_MAX 720
#define PLC_PITCH_LAG_MIN 100
#define MAX_PERIOD 1024
#define LPC_ORDER 24
#define FRAME_NONE 0
#define FRAME_NORMAL 1
#define FRAME_PLC_NOISE 2
#define FRAME_PLC_PERIODIC 4

/* celt_plc_pitch_search, celt_decoder.c:552 */
static int plc_pitch_search(i32 *decode_mem[2], int C)
{
   i16 lp_pitch_buf[OC_DECODE_BUFFER_SIZE >> 1];
   int pitch_index;
   oc_pitch_downsample(decode_mem, lp_pitch_buf, OC_DECODE_BUFFER_SIZE >> 1, C, 2);
   oc_pitch_search(lp_pitch_buf + (PLC_PITCH_LAG_MAX >> 1), lp_pitch_buf, OC_DECODE_BUFFER_SIZE - PLC_PITCH_LAG_MAX, PLC_PITCH_LAG_MAX - PLC_PITCH_LAG_MIN, &pitch_index);
   return PLC_PITCH_LAG_MAX - pitch_index;
}
/* prefilter_and_fold, celt_decoder.c:576 */
static void prefilter_and_fold(oc_celt_dec *st, int N)
{
   const int overlap = OVERLAP;
   i32 etmp[OVERLAP];
   for (int c = 0; c < st->channels; c++) {
      i32 *dm = st->decode_mem[c];
      oc_comb_filter(etmp, dm + OC_DECODE_BUFFER_SIZE - N, st->postfilter_period_old, st->postfilter_period, overlap, (i16)-st->postfilter_gain_old, (i16)-st->postfilter_gain,
            st->postfilter_tapset_old, st->postfilter_tapset, 0);
      for (int i = 0; i < overlap / 2; i++)
         dm[OC_DECODE_BUFFER_SIZE - N + i] = mult16_32_q15(oc_window[i], etmp[overlap - 1 - i]) + mult16_32_q15(oc_window[overlap - i - 1], etmp[i]);
   }
}
/* celt_fir_c, celt_lpc.c:140: y[i] = round(x[i] + sum num[m-1] x[i-m]); x must have ord samples of history before it */
static void celt_fir(const i16 *x, const i16 *num, i16 *y, int N, int ord)
{
   for (int i = 0; i < N; i++) {
      i32 sum = shl32((i32)x[i], SIG_SHIFT);
      for (int j = 0; j < ord; j++) sum = mac16_16(sum, num[ord - 1 - j], x[i + j - ord]);
      y[i] = sround16(sum, SIG_SHIFT);
   }
}
/* celt_iir, celt_lpc.c:186: the feedback taps see the outputs rounded (and saturated) to 16 bits */
static void celt_iir(const i32 *x, const i16 *den, i32 *y, int N, int ord, i16 *mem)
{
   i16 hist[LPC_ORDER];                  /* hist[m-1] = out16[i-m] */
   for (int j = 0; j < ord; j++) hist[j] = mem[j];
   for (int i = 0; i < N; i++) {
      i32 sum = x[i];
      for (int j = 0; j < ord; j++) sum -= mult16_16(den[j], hist[j]);
      for (int j = ord - 1; j >= 1; j--) hist[j] = hist[j - 1];
      hist[0] = sround16(sum, SIG_SHIFT);
      y[i] = sum;
   }
   for (int i = 0; i < ord; i++) mem[i] = (i16)y[N - i - 1];
}

/* celt_decode_lost, celt_decoder.c:679 */
static void celt_decode_lost(oc_celt_dec *st, int N, int LM)
{
   const int C = st->channels, overlap = OVERLAP, start = st->start;
   i32 *decode_mem[2], *out_syn[2];
   i32 *oldBandE = st->oldBandE, *backgroundLogE = st->backgroundLogE;
   int loss_duration = st->loss_duration, curr_frame_type = FRAME_PLC_PERIODIC;
   for (int c = 0; c < C; c++) { decode_mem[c] = st->decode_mem[c]; out_syn[c] = decode_mem[c] + OC_DECODE_BUFFER_SIZE - N; }
   if (st->plc_duration >= 40 || start != 0 || st->skip_plc) curr_frame_type = FRAME_PLC_NOISE;
   if (curr_frame_type == FRAME_PLC_NOISE) {
      i32 X[2 * 960];
      const int end = st->end, effEnd = imax(start, imin(end, NB_EBANDS));
      u32 seed;
      for (int c = 0; c < C; c++) memmove(decode_mem[c], decode_mem[c] + N, (OC_DECODE_BUFFER_SIZE - N + overlap) * sizeof(i32));
      if (st->prefilter_and_fold) prefilter_and_fold(st, N);
      i32 decay = loss_duration == 0 ? GC(1.5f) : GC(.5f);
      for (int c = 0; c < C; c++) for (int i = start; i < end; i++) oldBandE[c * NB_EBANDS + i] = imax(backgroundLogE[c * NB_EBANDS + i], oldBandE[c * NB_EBANDS + i] - decay);
      seed = st->rng;
      memset(X, 0, sizeof(X));
      for (int c = 0; c < C; c++) {
         for (int i = start; i < effEnd; i++) {
            int boffs = N * c + (oc_eBands[i] << LM), blen = (oc_eBands[i + 1] - oc_eBands[i]) << LM;
            for (int j = 0; j < blen; j++) { seed = 1664525u * seed + 1013904223u; X[boffs + j] = shl32((i32)((i32)seed >> 20), NORM_SHIFT - 14); }
            oc_renormalise_vector(X + boffs, blen, Q31ONE);
         }
      }
      st->rng = seed;
      celt_synthesis(X, out_syn, oldBandE, start, effEnd, C, C, 0, LM, 0);
      for (int c = 0; c < C; c++) {
         st->postfilter_period = imax(st->postfilter_period, COMBFILTER_MINPERIOD);
         st->postfilter_period_old = imax(st->postfilter_period_old, COMBFILTER_MINPERIOD);
         oc_comb_filter(out_syn[c], out_syn[c], st->postfilter_period_old, st->postfilter_period, 120, st->postfilter_gain_old, st->postfilter_gain,
               st->postfilter_tapset_old, st->postfilter_tapset, overlap);
         if (LM != 0)
            oc_comb_filter(out_syn[c] + 120, out_syn[c] + 120, st->postfilter_period, st->postfilter_period, N - 120, st->postfilter_gain, st->postfilter_gain,
                  st->postfilter_tapset, st->postfilter_tapset, overlap);
      }
      st->postfilter_period_old = st->postfilter_period; st->postfilter_gain_old = st->postfilter_gain; st->postfilter_tapset_old = st->postfilter_tapset;
      st->prefilter_and_fold = 0;
      st->skip_plc = 1;
   } else {
      i16 _exc[MAX_PERIOD + LPC_ORDER], fir_tmp[MAX_PERIOD], *exc = _exc + LPC_ORDER;
      i16 fade = Q15ONE;
      int pitch_index, exc_length;
      if (st->last_frame_type != FRAME_PLC_PERIODIC) st->last_pitch_index = pitch_index = plc_pitch_search(decode_mem, C);
      else { pitch_index = st->last_pitch_index; fade = QC16(.8f, 15); }
      exc_length = imin(2 * pitch_index, MAX_PERIOD);
      for (int c = 0; c < C; c++) {
         i16 decay, attenuation, *lpc = st->lpc + c * LPC_ORDER;
         i32 S1 = 0, *buf = decode_mem[c];
         int extrapolation_offset, extrapolation_len, i, j;
         for (i = 0; i < MAX_PERIOD + LPC_ORDER; i++) exc[i - LPC_ORDER] = sround16(buf[OC_DECODE_BUFFER_SIZE - MAX_PERIOD - LPC_ORDER + i], SIG_SHIFT);
         if (st->last_frame_type != FRAME_PLC_PERIODIC) {
            i32 ac[LPC_ORDER + 1];
            oc_autocorr(exc, ac, oc_window, overlap, LPC_ORDER, MAX_PERIOD);
            ac[0] += ac[0] >> 13;
            for (i = 1; i <= LPC_ORDER; i++) ac[i] -= mult16_32_q15(2 * i * i, ac[i]);
            oc_celt_lpc(lpc, ac, LPC_ORDER);
            while (1) {
               i16 tmp = Q15ONE;
               i32 sum = QC16(1., SIG_SHIFT);
               for (i = 0; i < LPC_ORDER; i++) sum += abs(lpc[i]);
               if (sum < 65535) break;
               for (i = 0; i < LPC_ORDER; i++) { tmp = (i16)mult16_16_q15(QC16(.99f, 15), tmp); lpc[i] = (i16)mult16_16_q15(lpc[i], tmp); }
            }
         }
         celt_fir(exc + MAX_PERIOD - exc_length, lpc, fir_tmp, exc_length, LPC_ORDER);
         memcpy(exc + MAX_PERIOD - exc_length, fir_tmp, exc_length * sizeof(i16));
         {
            i32 E1 = 1, E2 = 1, mx = 0;
            for (i = 0; i < exc_length; i++) mx = imax(mx, abs(exc[MAX_PERIOD - exc_length + i]));
            int shift = imax(0, 2 * celt_zlog2(mx) - 20);
            int decay_length = exc_length >> 1;
            for (i = 0; i < decay_length; i++) {
               i16 e = exc[MAX_PERIOD - decay_length + i];
               E1 += mult16_16(e, e) >> shift;
               e = exc[MAX_PERIOD - 2 * decay_length + i];
               E2 += mult16_16(e, e) >> shift;
            }
            E1 = imin(E1, E2);
            decay = (i16)oc_sqrt(oc_frac_div32(E1 >> 1, E2));
         }
         memmove(buf, buf + N, (OC_DECODE_BUFFER_SIZE - N) * sizeof(i32));
         extrapolation_offset = MAX_PERIOD - pitch_index;
         extrapolation_len = N + overlap;
         attenuation = (i16)mult16_16_q15(fade, decay);
         for (i = j = 0; i < extrapolation_len; i++, j++) {
            i16 tmp;
            if (j >= pitch_index) { j -= pitch_index; attenuation = (i16)mult16_16_q15(attenuation, decay); }
            buf[OC_DECODE_BUFFER_SIZE - N + i] = shl32((i32)(i16)mult16_16_q15(attenuation, exc[extrapolation_offset + j]), SIG_SHIFT);
            tmp = sround16(buf[OC_DECODE_BUFFER_SIZE - MAX_PERIOD - N + extrapolation_offset + j], SIG_SHIFT);
            S1 += mult16_16(tmp, tmp) >> 11;
         }
         {
            i16 lpc_mem[LPC_ORDER];
            for (i = 0; i < LPC_ORDER; i++) lpc_mem[i] = sround16(buf[OC_DECODE_BUFFER_SIZE - N - 1 - i], SIG_SHIFT);
            celt_iir(buf + OC_DECODE_BUFFER_SIZE - N, lpc, buf + OC_DECODE_BUFFER_SIZE - N, extrapolation_len, LPC_ORDER, lpc_mem);
            for (i = 0; i < extrapolation_len; i++) buf[OC_DECODE_BUFFER_SIZE - N + i] = saturate(buf[OC_DECODE_BUFFER_SIZE - N + i], SIG_SAT);
         }
         {
            i32 S2 = 0;
            for (i = 0; i < extrapolation_len; i++) { i16 tmp = sround16(buf[OC_DECODE_BUFFER_SIZE - N + i], SIG_SHIFT); S2 += mult16_16(tmp, tmp) >> 11; }
            if (!(S1 > (S2 >> 2))) { for (i = 0; i < extrapolation_len; i++) buf[OC_DECODE_BUFFER_SIZE - N + i] = 0; }
            else if (S1 < S2) {
               i16 ratio = (i16)oc_sqrt(oc_frac_div32((S1 >> 1) + 1, S2 + 1));
               for (i = 0; i < overlap; i++) {
                  i16 tmp_g = (i16)(Q15ONE - mult16_16_q15(oc_window[i], Q15ONE - ratio));
                  buf[OC_DECODE_BUFFER_SIZE - N + i] = mult16_32_q15(tmp_g, buf[OC_DECODE_BUFFER_SIZE - N + i]);
               }
               for (i = overlap; i < extrapolation_len; i++) buf[OC_DECODE_BUFFER_SIZE - N + i] = mult16_32_q15(ratio, buf[OC_DECODE_BUFFER_SIZE - N + i]);
            }
         }
      }
      st->prefilter_and_fold = 1;
   }
   st->loss_duration = imin(10000, loss_duration + (1 << LM));
   st->plc_duration = imin(10000, st->plc_duration + (1 << LM));
   st->last_frame_type = curr_frame_type;
}

/* celt_decode_with_ec_dred, celt_decoder.c:1104 — normal frames only */
int oc_celt_decode_with_ec(oc_celt_dec *st, const u8 *data, int len, i16 *pcm, int frame_size, oc_ec *dec)
{
   const int CC = st->channels, C = st->stream_channels, overlap = OVERLAP, start = st->start, end = st->end;
   int LM, M, N, effEnd, i, c;
   oc_ec _dec;
   i32 X[2 * 960];
   int fine_quant[NB_EBANDS], pulses[NB_EBANDS], cap[NB_EBANDS], offsets[NB_EBANDS], fine_priority[NB_EBANDS], tf_res[NB_EBANDS];
   u8 collapse_masks[2 * NB_EBANDS];
   i32 *decode_mem[2], *out_syn[2];
   i32 *oldBandE = st->oldBandE, *oldLogE = st->oldLogE, *oldLogE2 = st->oldLogE2, *backgroundLogE = st->backgroundLogE;
   int shortBlocks, isTransient, intra_ener, spread_decision, codedBands, alloc_trim, postfilter_pitch, postfilter_tapset;
   i16 postfilter_gain;
   int intensity = 0, dual_stereo = 0, anti_collapse_rsv, anti_collapse_on = 0, silence, dynalloc_logp;
   i32 total_bits, balance, tell, bits;
   for (LM = 0; LM <= 3; LM++) if (120 << LM == frame_size) break;
   if (LM > 3) return -1;
   M = 1 << LM;
   if (len < 0 || len > 1275 || pcm == 0) return -1;
   N = M * 120;
   for (c = 0; c < CC; c++) { decode_mem[c] = st->decode_mem[c]; out_syn[c] = decode_mem[c] + OC_DECODE_BUFFER_SIZE - N; }
   effEnd = imin(end, NB_EBANDS);
   if (data == 0 || len <= 1) {
      celt_decode_lost(st, N, LM);
      deemphasis(out_syn, pcm, N, CC, st->preemph_memD);
      return frame_size;
   }
   if (st->loss_duration == 0) st->skip_plc = 0;
   if (dec == 0) { oc_ec_dec_init(&_dec, data, len); dec = &_dec; }
   if (C == 1) for (i = 0; i < NB_EBANDS; i++) oldBandE[i] = imax(oldBandE[i], oldBandE[NB_EBANDS + i]);
   total_bits = len * 8;
   tell = oc_ec_tell(dec);
   if (tell >= total_bits) silence = 1;
   else if (tell == 1) silence = oc_ec_dec_bit_logp(dec, 15);
   else silence = 0;
   if (silence) { tell = len * 8; dec->nbits_total += tell - oc_ec_tell(dec); }
   postfilter_gain = 0; postfilter_pitch = 0; postfilter_tapset = 0;
   if (start == 0 && tell + 16 <= total_bits) {
      if (oc_ec_dec_bit_logp(dec, 1)) {
         int qg, octave = oc_ec_dec_uint(dec, 6);
         postfilter_pitch = (16 << octave) + oc_ec_dec_bits(dec, 4 + octave) - 1;
         qg = oc_ec_dec_bits(dec, 3);
         if (oc_ec_tell(dec) + 2 <= total_bits) postfilter_tapset = oc_ec_dec_icdf(dec, tapset_icdf, 2);
         postfilter_gain = (i16)(QC16(.09375f, 15) * (qg + 1));
      }
      tell = oc_ec_tell(dec);
   }
   if (LM > 0 && tell + 3 <= total_bits) { isTransient = oc_ec_dec_bit_logp(dec, 3); tell = oc_ec_tell(dec); }
   else isTransient = 0;
   shortBlocks = isTransient ? M : 0;
   intra_ener = tell + 3 <= total_bits ? oc_ec_dec_bit_logp(dec, 3) : 0;
   if (!intra_ener && st->loss_duration != 0) {            /* energy prediction safety after a loss, celt_decoder.c:1387 */
      for (c = 0; c < 2; c++) {
         i32 safety = 0;
         int missing = imin(10, st->loss_duration >> LM);
         if (LM == 0) safety = GC(1.5f);
         else if (LM == 1) safety = GC(.5f);
         for (i = start; i < end; i++) {
            if (oldBandE[c * NB_EBANDS + i] < imax(oldLogE[c * NB_EBANDS + i], oldLogE2[c * NB_EBANDS + i])) {
               i32 E0 = oldBandE[c * NB_EBANDS + i], E1 = oldLogE[c * NB_EBANDS + i], E2 = oldLogE2[c * NB_EBANDS + i];
               i32 slope = imax(E1 - E0, half32(E2 - E0));
               slope = imin(slope, GC(2.f));
               E0 -= imax(0, (1 + missing) * slope);
               oldBandE[c * NB_EBANDS + i] = imax(-GC(20.f), E0);
            } else oldBandE[c * NB_EBANDS + i] = imin(imin(oldBandE[c * NB_EBANDS + i], oldLogE[c * NB_EBANDS + i]), oldLogE2[c * NB_EBANDS + i]);
            oldBandE[c * NB_EBANDS + i] -= safety;
         }
      }
   }
   oc_unquant_coarse_energy(start, end, oldBandE, intra_ener, dec, C, LM);
   tf_decode(start, end, isTransient, tf_res, LM, dec);
   tell = oc_ec_tell(dec);
   spread_decision = SPREAD_NORMAL;
   if (tell + 4 <= total_bits) spread_decision = oc_ec_dec_icdf(dec, spread_icdf, 5);
   oc_init_caps(cap, LM, C);
   dynalloc_logp = 6;
   total_bits <<= BITRES;
   tell = oc_ec_tell_frac(dec);
   for (i = start; i < end; i++) {
      int width = C * (oc_eBands[i + 1] - oc_eBands[i]) << LM;
      int quanta = imin(width << BITRES, imax(6 << BITRES, width));
      int dynalloc_loop_logp = dynalloc_logp, boost = 0;
      while (tell + (dynalloc_loop_logp << BITRES) < total_bits && boost < cap[i]) {
         int flag = oc_ec_dec_bit_logp(dec, dynalloc_loop_logp);
         tell = oc_ec_tell_frac(dec);
         if (!flag) break;
         boost += quanta;
         total_bits -= quanta;
         dynalloc_loop_logp = 1;
      }
      offsets[i] = boost;
      if (boost > 0) dynalloc_logp = imax(2, dynalloc_logp - 1);
   }
   alloc_trim = tell + (6 << BITRES) <= total_bits ? oc_ec_dec_icdf(dec, trim_icdf, 7) : 5;
   bits = (((i32)len * 8) << BITRES) - (i32)oc_ec_tell_frac(dec) - 1;
   anti_collapse_rsv = isTransient && LM >= 2 && bits >= ((LM + 2) << BITRES) ? (1 << BITRES) : 0;
   bits -= anti_collapse_rsv;
   codedBands = oc_compute_allocation(start, end, offsets, cap, alloc_trim, &intensity, &dual_stereo, bits, &balance, pulses,
         fine_quant, fine_priority, C, LM, dec, 0, 0, 0);
   oc_unquant_fine_energy(start, end, oldBandE, fine_quant, dec, C);
   for (c = 0; c < CC; c++) memmove(decode_mem[c], decode_mem[c] + N, (OC_DECODE_BUFFER_SIZE - N + overlap) * sizeof(i32));
   memset(X, 0, sizeof(X));
   oc_quant_all_bands(0, start, end, X, C == 2 ? X + N : 0, collapse_masks, 0, pulses, shortBlocks, spread_decision, dual_stereo, intensity,
         tf_res, len * (8 << BITRES) - anti_collapse_rsv, balance, dec, LM, codedBands, &st->rng, 0, st->disable_inv);
   if (anti_collapse_rsv > 0) anti_collapse_on = oc_ec_dec_bits(dec, 1);
   oc_unquant_energy_finalise(start, end, oldBandE, fine_quant, fine_priority, len * 8 - oc_ec_tell(dec), dec, C);
   if (anti_collapse_on) anti_collapse(X, collapse_masks, LM, C, N, start, end, oldBandE, oldLogE, oldLogE2, pulses, st->rng);
   if (silence) for (i = 0; i < C * NB_EBANDS; i++) oldBandE[i] = -GC(28.f);
   OC_DUMP("dec_X", X, C * N * 4); OC_DUMP("dec_oldBandE", oldBandE, 2 * NB_EBANDS * 4);
   if (st->prefilter_and_fold) prefilter_and_fold(st, N);
   celt_synthesis(X, out_syn, oldBandE, start, effEnd, C, CC, isTransient, LM, silence);
   for (c = 0; c < CC; c++) OC_DUMP("dec_syn", out_syn[c], N * 4);
   for (c = 0; c < CC; c++) {
      st->postfilter_period = imax(st->postfilter_period, COMBFILTER_MINPERIOD);
      st->postfilter_period_old = imax(st->postfilter_period_old, COMBFILTER_MINPERIOD);
      oc_comb_filter(out_syn[c], out_syn[c], st->postfilter_period_old, st->postfilter_period, 120, st->postfilter_gain_old, st->postfilter_gain,
            st->postfilter_tapset_old, st->postfilter_tapset, overlap);
      if (LM != 0)
         oc_comb_filter(out_syn[c] + 120, out_syn[c] + 120, st->postfilter_period, postfilter_pitch, N - 120, st->postfilter_gain, postfilter_gain,
               st->postfilter_tapset, postfilter_tapset, overlap);
   }
   st->postfilter_period_old = st->postfilter_period; st->postfilter_gain_old = st->postfilter_gain; st->postfilter_tapset_old = st->postfilter_tapset;
   st->postfilter_period = postfilter_pitch; st->postfilter_gain = postfilter_gain; st->postfilter_tapset = postfilter_tapset;
   if (LM != 0) { st->postfilter_period_old = st->postfilter_period; st->postfilter_gain_old = st->postfilter_gain; st->postfilter_tapset_old = st->postfilter_tapset; }
   if (C == 1) memcpy(&oldBandE[NB_EBANDS], oldBandE, NB_EBANDS * sizeof(i32));
   if (!isTransient) {
      memcpy(oldLogE2, oldLogE, 2 * NB_EBANDS * sizeof(i32));
      memcpy(oldLogE, oldBandE, 2 * NB_EBANDS * sizeof(i32));
   } else for (i = 0; i < 2 * NB_EBANDS; i++) oldLogE[i] = imin(oldLogE[i], oldBandE[i]);
   {
      i32 max_background_increase = imin(160, st->loss_duration + M) * GC(0.001f);
      for (i = 0; i < 2 * NB_EBANDS; i++) backgroundLogE[i] = imin(backgroundLogE[i] + max_background_increase, oldBandE[i]);
   }
   for (c = 0; c < 2; c++) {
      for (i = 0; i < start; i++) { oldBandE[c * NB_EBANDS + i] = 0; oldLogE[c * NB_EBANDS + i] = oldLogE2[c * NB_EBANDS + i] = -GC(28.f); }
      for (i = end; i < NB_EBANDS; i++) { oldBandE[c * NB_EBANDS + i] = 0; oldLogE[c * NB_EBANDS + i] = oldLogE2[c * NB_EBANDS + i] = -GC(28.f); }
   }
   st->rng = dec->rng;
   deemphasis(out_syn, pcm, N, CC, st->preemph_memD);
   st->loss_duration = 0; st->plc_duration = 0; st->last_frame_type = 1 /* FRAME_NORMAL */; st->prefilter_and_fold = 0;
   if (oc_ec_tell(dec) > 8 * len) return -3;
   if (dec->error) st->error = 1;
   return frame_size;
}

/* oc_celt_enc.c — the CELT frame encoder (fixed-point, 48 kHz mode, no QEXT/custom modes/analysis.c).
 * Oracle restatement of celt/celt_encoder.c: :267 transient_analysis, :473 patch_transient_decision,
 * :511 compute_mdcts, :557 celt_preemphasis, :650 l1_metric, :663 tf_analysis, :823 tf_encode,
 * :865 alloc_trim_analysis, :957 stereo_analysis, :988/:1029 medians, :1049 dynalloc_analysis,
 * :1272-1403 tone detection, :1405 run_prefilter, :1605 compute_vbr, :1726 celt_encode_with_ec,
 * reset state :3080-3095.  Float-API analysis (src/analysis.c) is compiled out of the bit-exact oracle
 * build (DISABLE_FLOAT_API), so AnalysisInfo.valid == 0 everywhere. */
#include "oc_celt.h"
#include "oc_celt_enc.h"
#include <stdlib.h>

/* test-only tracing of intermediate values (used to localise kernel-vs-oracle divergences) */
void (*oc_dump_hook)(const char *tag, const void *p, int nbytes) = 0;
#define OC_DUMP(tag, p, n) do { if (oc_dump_hook) oc_dump_hook(tag, p, n); } while (0)
#define OC_DUMPI(tag, v) do { i32 v_ = (i32)(v); OC_DUMP(tag, &v_, 4); } while (0)

i32 oc_inner_prod_norm_shift(const i32 *x, const i32 *y, int len);

static const u8 trim_icdf[11] = {126, 124, 119, 109, 87, 41, 19, 9, 4, 2, 0};
static const u8 spread_icdf[4] = {25, 23, 2, 0};
static const u8 tapset_icdf[3] = {2, 1, 0};
static const signed char tf_select_table[4][8] = {
   {0, -1, 0, -1, 0, -1, 0, -1}, {0, -1, 0, -2, 1, 0, 1, -1}, {0, -2, 0, -3, 2, 0, 1, -1}, {0, -2, 0, -3, 3, 0, 1, -1}};

void oc_celt_enc_init(oc_celt_enc *st, int channels)
{
   memset(st, 0, sizeof(*st));
   st->channels = st->stream_channels = channels;
   st->start = 0; st->end = NB_EBANDS; st->constrained_vbr = 1; st->clip = 1; st->bitrate = -1 /* OPUS_BITRATE_MAX */;
   st->vbr = 0; st->force_intra = 0; st->complexity = 5; st->lsb_depth = 24;
   for (int i = 0; i < 2 * NB_EBANDS; i++) st->oldLogE[i] = st->oldLogE2[i] = -GC(28.f);
   st->vbr_offset = 0; st->delayedIntra = 1; st->spread_decision = SPREAD_NORMAL; st->tonal_average = 256;
   st->hf_average = 0; st->tapset_decision = 0;
}

static i32 maxabs16(const i16 *x, int len)
{
   i32 mx = 0, mn = 0;
   for (int i = 0; i < len; i++) { mx = imax(mx, x[i]); mn = imin(mn, x[i]); }
   return imax(mx, -mn);
}
static i32 maxabs32(const i32 *x, int len)
{
   i32 mx = 0, mn = 0;
   for (int i = 0; i < len; i++) { mx = imax(mx, x[i]); mn = imin(mn, x[i]); }
   return imax(mx, neg32(mn));
}
static i32 bitrate_to_bits(i32 bitrate, i32 Fs, i32 frame_size) { return bitrate * 6 / (6 * Fs / frame_size); }

static int transient_analysis(const i32 *in, int len, int C, i16 *tf_estimate, int *tf_chan, int allow_weak_transients,
      int *weak_transient, i16 tone_freq, i32 toneishness)
{
   static const u8 inv_table[128] = {
      255, 255, 156, 110, 86, 70, 59, 51, 45, 40, 37, 33, 31, 28, 26, 25, 23, 22, 21, 20, 19, 18, 17, 16, 16, 15, 15, 14, 13, 13, 12, 12,
      12, 12, 11, 11, 11, 10, 10, 10, 9, 9, 9, 9, 9, 9, 8, 8, 8, 8, 8, 7, 7, 7, 7, 7, 7, 6, 6, 6, 6, 6, 6, 6,
      6, 6, 6, 6, 6, 6, 6, 6, 6, 5, 5, 5, 5, 5, 5, 5, 5, 5, 5, 5, 5, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4,
      4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 3, 3, 3, 3, 3, 3, 3, 3, 3, 3, 3, 3, 3, 3, 3, 3, 3, 2};
   i16 tmp[1080];
   int is_transient = 0, forward_shift = 4, len2 = len / 2;
   i32 mask_metric = 0;
   int in_shift = imax(0, celt_ilog2(1 + maxabs32(in, C * len)) - 14);
   *weak_transient = 0;
   if (allow_weak_transients) forward_shift = 5;
   for (int c = 0; c < C; c++) {
      i32 mean, unmask = 0, norm, mem0 = 0, mem1 = 0;
      i16 maxE;
      for (int i = 0; i < len; i++) {
         i32 x = in[i + c * len] >> in_shift;
         i32 y = add32(mem0, x);
         mem0 = mem1 + y - shl32(x, 1);
         mem1 = x - (y >> 1);
         tmp[i] = sround16(y, 2);
      }
      memset(tmp, 0, 12 * sizeof(i16));
      {
         int shift = 14 - celt_ilog2(imax(1, maxabs16(tmp, len)));
         if (shift != 0) for (int i = 0; i < len; i++) tmp[i] = shl16(tmp[i], shift);
      }
      mean = 0; mem0 = 0;
      for (int i = 0; i < len2; i++) {
         i32 x2 = pshr32(mult16_16(tmp[2 * i], tmp[2 * i]) + mult16_16(tmp[2 * i + 1], tmp[2 * i + 1]), 4);
         mean += pshr32(x2, 12);
         mem0 = mem0 + pshr32(x2 - mem0, forward_shift);
         tmp[i] = (i16)pshr32(mem0, 12);
      }
      mem0 = 0; maxE = 0;
      for (int i = len2 - 1; i >= 0; i--) {
         mem0 = mem0 + pshr32(shl32(tmp[i], 4) - mem0, 3);
         tmp[i] = (i16)pshr32(mem0, 4);
         maxE = (i16)imax(maxE, tmp[i]);
      }
      mean = mult16_16(oc_sqrt(mean), oc_sqrt(mult16_16(maxE, len2 >> 1)));
      norm = shl32((i32)len2, 6 + 14) / add32(EPSILON, mean >> 1);
      unmask = 0;
      for (int i = 12; i < len2 - 5; i += 4) {
         int id = imax(0, imin(127, mult16_32_q15(tmp[i] + EPSILON, norm)));
         unmask += inv_table[id];
      }
      unmask = 64 * unmask * 4 / (6 * (len2 - 17));
      if (unmask > mask_metric) { *tf_chan = c; mask_metric = unmask; }
   }
   is_transient = mask_metric > 200;
   if (toneishness > QC32(.98f, 29) && tone_freq < QC16(0.026f, 13)) { is_transient = 0; mask_metric = 0; }
   if (allow_weak_transients && is_transient && mask_metric < 600) { is_transient = 0; *weak_transient = 1; }
   i16 tf_max = (i16)imax(0, oc_sqrt(27 * mask_metric) - 42);
   *tf_estimate = (i16)oc_sqrt(imax(0, shl32(mult16_16(QC16(0.0069, 14), imin(163, tf_max)), 14) - QC32(0.139, 28)));
   return is_transient;
}

static int patch_transient_decision(const i32 *newE, const i32 *oldE, int start, int end, int C)
{
   i32 mean_diff = 0, spread_old[26];
   if (C == 1) {
      spread_old[start] = oldE[start];
      for (int i = start + 1; i < end; i++) spread_old[i] = imax(spread_old[i - 1] - GC(1.0f), oldE[i]);
   } else {
      spread_old[start] = imax(oldE[start], oldE[start + NB_EBANDS]);
      for (int i = start + 1; i < end; i++) spread_old[i] = imax(spread_old[i - 1] - GC(1.0f), imax(oldE[i], oldE[i + NB_EBANDS]));
   }
   for (int i = end - 2; i >= start; i--) spread_old[i] = imax(spread_old[i], spread_old[i + 1] - GC(1.0f));
   for (int c = 0; c < C; c++)
      for (int i = imax(2, start); i < end - 1; i++) {
         i16 x1 = (i16)imax(0, newE[i + c * NB_EBANDS]);   /* opus_val16 in the reference: truncation is intended parity */
         i16 x2 = (i16)imax(0, spread_old[i]);
         mean_diff = add32(mean_diff, imax(0, sub32(x1, x2)));
      }
   mean_diff = mean_diff / (C * (end - 1 - imax(2, start)));
   return mean_diff > GC(1.f);
}

static void compute_mdcts(int shortBlocks, i32 *in, i32 *out, int C, int CC, int LM)
{
   int B, N, shift;
   if (shortBlocks) { B = shortBlocks; N = SHORT_MDCT; shift = MAX_LM; }
   else { B = 1; N = SHORT_MDCT << LM; shift = MAX_LM - LM; }
   for (int c = 0; c < CC; c++)
      for (int b = 0; b < B; b++)
         oc_mdct_forward(in + c * (B * N + OVERLAP) + b * N, &out[b + c * N * B], shift, B);
   if (CC == 2 && C == 1) for (int i = 0; i < B * N; i++) out[i] = add32(out[i] >> 1, out[B * N + i] >> 1);
}

/* celt_preemphasis (coef[1]==0, upsample==1 path; RES_SHIFT 0 so clipping is a no-op in this build) */
static void preemphasis(const i16 *pcmp, i32 *inp, int N, int CC, i32 *mem)
{
   i32 m = *mem;
   for (int i = 0; i < N; i++) {
      i32 x = shl32((i32)pcmp[CC * i], SIG_SHIFT);
      inp[i] = x - m;
      m = mult16_32_q15(27853, x);
   }
   *mem = m;
}

static i32 l1_metric(const i32 *tmp, int N, int LM, i16 bias)
{
   i32 L1 = 0;
   for (int i = 0; i < N; i++) L1 += iabs(tmp[i] >> (NORM_SHIFT - 14));
   return mac16_32_q15(L1, LM * bias, L1);
}

static int tf_analysis(int len, int isTransient, int *tf_res, int lambda, const i32 *X, int N0, int LM, i16 tf_estimate,
      int tf_chan, const int *importance)
{
   const int16_t *eB = oc_eBands;
   int metric[NB_EBANDS], path0[NB_EBANDS], path1[NB_EBANDS], cost0, cost1, selcost[2], tf_select = 0;
   i32 tmp[176], tmp_1[176];
   i16 bias = (i16)mult16_16_q14(QC16(.04f, 15), imax(-QC16(.25f, 14), QC16(.5f, 14) - tf_estimate));
   for (int i = 0; i < len; i++) {
      int N = (eB[i + 1] - eB[i]) << LM, narrow = (eB[i + 1] - eB[i]) == 1, best_level = 0;
      memcpy(tmp, &X[tf_chan * N0 + (eB[i] << LM)], N * sizeof(i32));
      i32 L1 = l1_metric(tmp, N, isTransient ? LM : 0, bias), best_L1 = L1;
      if (isTransient && !narrow) {
         memcpy(tmp_1, tmp, N * sizeof(i32));
         oc_haar1(tmp_1, N >> LM, 1 << LM);
         L1 = l1_metric(tmp_1, N, LM + 1, bias);
         if (L1 < best_L1) { best_L1 = L1; best_level = -1; }
      }
      for (int k = 0; k < LM + !(isTransient || narrow); k++) {
         int B = isTransient ? (LM - k - 1) : k + 1;
         oc_haar1(tmp, N >> k, 1 << k);
         L1 = l1_metric(tmp, N, B, bias);
         if (L1 < best_L1) { best_L1 = L1; best_level = k + 1; }
      }
      metric[i] = isTransient ? 2 * best_level : -2 * best_level;
      if (narrow && (metric[i] == 0 || metric[i] == -2 * LM)) metric[i] -= 1;
   }
   for (int sel = 0; sel < 2; sel++) {
      cost0 = importance[0] * abs(metric[0] - 2 * tf_select_table[LM][4 * isTransient + 2 * sel + 0]);
      cost1 = importance[0] * abs(metric[0] - 2 * tf_select_table[LM][4 * isTransient + 2 * sel + 1]) + (isTransient ? 0 : lambda);
      for (int i = 1; i < len; i++) {
         int curr0 = imin(cost0, cost1 + lambda), curr1 = imin(cost0 + lambda, cost1);
         cost0 = curr0 + importance[i] * abs(metric[i] - 2 * tf_select_table[LM][4 * isTransient + 2 * sel + 0]);
         cost1 = curr1 + importance[i] * abs(metric[i] - 2 * tf_select_table[LM][4 * isTransient + 2 * sel + 1]);
      }
      selcost[sel] = imin(cost0, cost1);
   }
   if (selcost[1] < selcost[0] && isTransient) tf_select = 1;
   cost0 = importance[0] * abs(metric[0] - 2 * tf_select_table[LM][4 * isTransient + 2 * tf_select + 0]);
   cost1 = importance[0] * abs(metric[0] - 2 * tf_select_table[LM][4 * isTransient + 2 * tf_select + 1]) + (isTransient ? 0 : lambda);
   for (int i = 1; i < len; i++) {
      int curr0, curr1, from0 = cost0, from1 = cost1 + lambda;
      if (from0 < from1) { curr0 = from0; path0[i] = 0; } else { curr0 = from1; path0[i] = 1; }
      from0 = cost0 + lambda; from1 = cost1;
      if (from0 < from1) { curr1 = from0; path1[i] = 0; } else { curr1 = from1; path1[i] = 1; }
      cost0 = curr0 + importance[i] * abs(metric[i] - 2 * tf_select_table[LM][4 * isTransient + 2 * tf_select + 0]);
      cost1 = curr1 + importance[i] * abs(metric[i] - 2 * tf_select_table[LM][4 * isTransient + 2 * tf_select + 1]);
   }
   tf_res[len - 1] = cost0 < cost1 ? 0 : 1;
   for (int i = len - 2; i >= 0; i--) tf_res[i] = tf_res[i + 1] == 1 ? path1[i + 1] : path0[i + 1];
   return tf_select;
}

static void tf_encode(int start, int end, int isTransient, int *tf_res, int LM, int tf_select, oc_ec *enc)
{
   u32 budget = enc->storage * 8, tell = oc_ec_tell(enc);
   int logp = isTransient ? 2 : 4, curr = 0, tf_changed = 0;
   int tf_select_rsv = LM > 0 && tell + logp + 1 <= budget;
   budget -= tf_select_rsv;
   for (int i = start; i < end; i++) {
      if (tell + logp <= budget) {
         oc_ec_enc_bit_logp(enc, tf_res[i] ^ curr, logp);
         tell = oc_ec_tell(enc);
         curr = tf_res[i];
         tf_changed |= curr;
      } else tf_res[i] = curr;
      logp = isTransient ? 4 : 5;
   }
   if (tf_select_rsv && tf_select_table[LM][4 * isTransient + 0 + tf_changed] != tf_select_table[LM][4 * isTransient + 2 + tf_changed])
      oc_ec_enc_bit_logp(enc, tf_select, 1);
   else tf_select = 0;
   for (int i = start; i < end; i++) tf_res[i] = tf_select_table[LM][4 * isTransient + 2 * tf_select + tf_res[i]];
}

static int alloc_trim_analysis(const i32 *X, const i32 *bandLogE, int end, int LM, int C, int N0, i16 *stereo_saving,
      i16 tf_estimate, int intensity, i32 surround_trim, i32 equiv_rate)
{
   const int16_t *eB = oc_eBands;
   i32 diff = 0;
   i16 trim = QC16(5.f, 8), logXC, logXC2;
   if (equiv_rate < 64000) trim = QC16(4.f, 8);
   else if (equiv_rate < 80000) { i32 frac = (equiv_rate - 64000) >> 10; trim = (i16)(QC16(4.f, 8) + QC16(1.f / 16.f, 8) * frac); }
   if (C == 2) {
      i16 sum = 0, minXC;
      for (int i = 0; i < 8; i++) {
         i32 partial = oc_inner_prod_norm_shift(&X[eB[i] << LM], &X[N0 + (eB[i] << LM)], (eB[i + 1] - eB[i]) << LM);
         sum = add16(sum, extract16(partial >> 18));
      }
      sum = (i16)mult16_16_q15(QC16(1.f / 8, 15), sum);
      sum = (i16)imin(QC16(1.f, 10), iabs(sum));
      minXC = sum;
      for (int i = 8; i < intensity; i++) {
         i32 partial = oc_inner_prod_norm_shift(&X[eB[i] << LM], &X[N0 + (eB[i] << LM)], (eB[i + 1] - eB[i]) << LM);
         minXC = (i16)imin(minXC, iabs(extract16(partial >> 18)));
      }
      minXC = (i16)imin(QC16(1.f, 10), iabs(minXC));
      logXC = oc_log2(QC32(1.001f, 20) - mult16_16(sum, sum));
      logXC2 = (i16)imax(logXC >> 1, oc_log2(QC32(1.001f, 20) - mult16_16(minXC, minXC)));
      logXC = (i16)pshr32(logXC - QC16(6.f, 10), 10 - 8);
      logXC2 = (i16)pshr32(logXC2 - QC16(6.f, 10), 10 - 8);
      trim = (i16)(trim + imax(-QC16(4.f, 8), mult16_16_q15(QC16(.75f, 15), logXC)));
      *stereo_saving = (i16)imin(*stereo_saving + QC16(0.25f, 8), -(logXC2 >> 1));
   }
   for (int c = 0; c < C; c++)
      for (int i = 0; i < end - 1; i++) diff += (bandLogE[i + c * NB_EBANDS] >> 5) * (i32)(2 + 2 * i - end);
   diff /= C * (end - 1);
   trim = (i16)(trim - imax(-QC16(2.f, 8), imin(QC16(2.f, 8), ((diff + QC32(1.f, DB_SHIFT - 5)) >> (DB_SHIFT - 13)) / 6)));
   trim = (i16)(trim - (surround_trim >> (DB_SHIFT - 8)));
   trim = (i16)(trim - 2 * (tf_estimate >> (14 - 8)));
   int trim_index = pshr32(trim, 8);
   return imax(0, imin(10, trim_index));
}

static int stereo_analysis(const i32 *X, int LM, int N0)
{
   const int16_t *eB = oc_eBands;
   i32 sumLR = EPSILON, sumMS = EPSILON;
   for (int i = 0; i < 13; i++)
      for (int j = eB[i] << LM; j < eB[i + 1] << LM; j++) {
         i32 L = X[j] >> (NORM_SHIFT - 14), R = X[N0 + j] >> (NORM_SHIFT - 14), M = add32(L, R), S = sub32(L, R);
         sumLR = add32(sumLR, add32(iabs(L), iabs(R)));
         sumMS = add32(sumMS, add32(iabs(M), iabs(S)));
      }
   sumMS = mult16_32_q15(QC16(0.707107f, 15), sumMS);
   int thetas = 13;
   if (LM <= 1) thetas -= 8;
   return mult16_32_q15((eB[13] << (LM + 1)) + thetas, sumMS) > mult16_32_q15(eB[13] << (LM + 1), sumLR);
}

static i32 median_of_5(const i32 *x)
{
   i32 t0, t1, t2 = x[2], t3, t4, t;
   if (x[0] > x[1]) { t0 = x[1]; t1 = x[0]; } else { t0 = x[0]; t1 = x[1]; }
   if (x[3] > x[4]) { t3 = x[4]; t4 = x[3]; } else { t3 = x[3]; t4 = x[4]; }
   if (t0 > t3) { t = t0; t0 = t3; t3 = t; t = t1; t1 = t4; t4 = t; }
   if (t2 > t1) return t1 < t3 ? imin(t2, t3) : imin(t4, t1);
   return t2 < t3 ? imin(t1, t3) : imin(t2, t4);
}
static i32 median_of_3(const i32 *x)
{
   i32 t0, t1, t2 = x[2];
   if (x[0] > x[1]) { t0 = x[1]; t1 = x[0]; } else { t0 = x[0]; t1 = x[1]; }
   if (t1 < t2) return t1;
   if (t0 < t2) return t2;
   return t0;
}

static i32 dynalloc_analysis(const i32 *bandLogE, const i32 *bandLogE2, const i32 *oldBandE, int start, int end, int C,
      int *offsets, int lsb_depth, int isTransient, int vbr, int constrained_vbr, int LM, int effectiveBytes,
      i32 *tot_boost_, int lfe, const i32 *surround_dynalloc, int *importance, int *spread_weight, i16 tone_freq, i32 toneishness)
{
   const int16_t *eB = oc_eBands;
   const int nbEBands = NB_EBANDS;
   i32 tot_boost = 0, maxDepth, follower[2 * NB_EBANDS], noise_floor[NB_EBANDS], bandLogE3[NB_EBANDS];
   memset(offsets, 0, nbEBands * sizeof(int));
   maxDepth = -GC(31.9f);
   for (int i = 0; i < end; i++)
      noise_floor[i] = GC(0.0625f) * oc_logN[i] + GC(.5f) + shl32(9 - lsb_depth, DB_SHIFT) - shl32(oc_eMeans[i], DB_SHIFT - 4)
            + GC(.0062f) * (i + 5) * (i + 5);
   for (int c = 0; c < C; c++) for (int i = 0; i < end; i++) maxDepth = imax(maxDepth, bandLogE[c * nbEBands + i] - noise_floor[i]);
   {
      i32 mask[NB_EBANDS], sig[NB_EBANDS];
      for (int i = 0; i < end; i++) mask[i] = bandLogE[i] - noise_floor[i];
      if (C == 2) for (int i = 0; i < end; i++) mask[i] = imax(mask[i], bandLogE[nbEBands + i] - noise_floor[i]);
      memcpy(sig, mask, end * sizeof(i32));
      for (int i = 1; i < end; i++) mask[i] = imax(mask[i], mask[i - 1] - GC(2.f));
      for (int i = end - 2; i >= 0; i--) mask[i] = imax(mask[i], mask[i + 1] - GC(3.f));
      for (int i = 0; i < end; i++) {
         i32 smr = sig[i] - imax(imax(0, maxDepth - GC(12.f)), mask[i]);
         int shift = -pshr32(imax(-GC(5.f), imin(0, smr)), DB_SHIFT);
         spread_weight[i] = 32 >> shift;
      }
   }
   if (effectiveBytes >= (30 + 5 * LM) && !lfe) {
      int last = 0;
      for (int c = 0; c < C; c++) {
         i32 offset, tmp, *f;
         memcpy(bandLogE3, &bandLogE2[c * nbEBands], end * sizeof(i32));
         if (LM == 0) for (int i = 0; i < imin(8, end); i++) bandLogE3[i] = imax(bandLogE2[c * nbEBands + i], oldBandE[c * nbEBands + i]);
         f = &follower[c * nbEBands];
         f[0] = bandLogE3[0];
         for (int i = 1; i < end; i++) {
            if (bandLogE3[i] > bandLogE3[i - 1] + GC(.5f)) last = i;
            f[i] = imin(f[i - 1] + GC(1.5f), bandLogE3[i]);
         }
         for (int i = last - 1; i >= 0; i--) f[i] = imin(f[i], imin(f[i + 1] + GC(2.f), bandLogE3[i]));
         offset = GC(1.f);
         for (int i = 2; i < end - 2; i++) f[i] = imax(f[i], median_of_5(&bandLogE3[i - 2]) - offset);
         tmp = median_of_3(&bandLogE3[0]) - offset;
         f[0] = imax(f[0], tmp); f[1] = imax(f[1], tmp);
         tmp = median_of_3(&bandLogE3[end - 3]) - offset;
         f[end - 2] = imax(f[end - 2], tmp); f[end - 1] = imax(f[end - 1], tmp);
         for (int i = 0; i < end; i++) f[i] = imax(f[i], noise_floor[i]);
      }
      if (C == 2) {
         for (int i = start; i < end; i++) {
            follower[nbEBands + i] = imax(follower[nbEBands + i], follower[i] - GC(4.f));
            follower[i] = imax(follower[i], follower[nbEBands + i] - GC(4.f));
            follower[i] = half32(imax(0, bandLogE[i] - follower[i]) + imax(0, bandLogE[nbEBands + i] - follower[nbEBands + i]));
         }
      } else for (int i = start; i < end; i++) follower[i] = imax(0, bandLogE[i] - follower[i]);
      for (int i = start; i < end; i++) follower[i] = imax(follower[i], surround_dynalloc[i]);
      for (int i = start; i < end; i++) importance[i] = pshr32(13 * oc_exp2_db(imin(follower[i], GC(4.f))), 16);
      if ((!vbr || constrained_vbr) && !isTransient) for (int i = start; i < end; i++) follower[i] = half32(follower[i]);
      for (int i = start; i < end; i++) {
         if (i < 8) follower[i] *= 2;
         if (i >= 12) follower[i] = half32(follower[i]);
      }
      if (toneishness > QC32(.98f, 29)) {
         int freq_bin = pshr32((i32)tone_freq * QC16(120 / 3.14159265358979323846, 9), 13 + 9);
         for (int i = start; i < end; i++) {
            if (freq_bin >= eB[i] && freq_bin <= eB[i + 1]) follower[i] += GC(2.f);
            if (freq_bin >= eB[i] - 1 && freq_bin <= eB[i + 1] + 1) follower[i] += GC(1.f);
            if (freq_bin >= eB[i] - 2 && freq_bin <= eB[i + 1] + 2) follower[i] += GC(1.f);
            if (freq_bin >= eB[i] - 3 && freq_bin <= eB[i + 1] + 3) follower[i] += GC(.5f);
         }
         if (freq_bin >= eB[end]) { follower[end - 1] += GC(2.f); follower[end - 2] += GC(1.f); }
      }
      if (effectiveBytes > 320) follower[0] += imin(GC(1.5f), GC(1e-3f) * (effectiveBytes - 320));
      for (int i = start; i < end; i++) {
         int width, boost, boost_bits;
         follower[i] = imin(follower[i], GC(4));
         follower[i] = follower[i] >> 8;
         width = C * (eB[i + 1] - eB[i]) << LM;
         if (width < 6) { boost = (int)(follower[i] >> (DB_SHIFT - 8)); boost_bits = boost * width << BITRES; }
         else if (width > 48) { boost = (int)((follower[i] * 8) >> (DB_SHIFT - 8)); boost_bits = (boost * width << BITRES) / 8; }
         else { boost = (int)((follower[i] * width / 6) >> (DB_SHIFT - 8)); boost_bits = boost * 6 << BITRES; }
         if ((!vbr || (constrained_vbr && !isTransient)) && (tot_boost + boost_bits) >> BITRES >> 3 > 2 * effectiveBytes / 3) {
            i32 cap = ((2 * effectiveBytes / 3) << BITRES << 3);
            offsets[i] = cap - tot_boost;
            tot_boost = cap;
            break;
         } else { offsets[i] = boost; tot_boost += boost_bits; }
      }
   } else for (int i = start; i < end; i++) importance[i] = 13;
   *tot_boost_ = tot_boost;
   return maxDepth;
}

static void normalize_tone_input(i16 *x, int len)
{
   i32 ac0 = len;
   for (int i = 0; i < len; i++) ac0 = add32(ac0, mult16_16(x[i], x[i]) >> 10);
   int shift = 5 - (28 - celt_ilog2(ac0)) / 2;
   if (shift > 0) for (int i = 0; i < len; i++) x[i] = (i16)pshr32(x[i], shift);
}
static int acos_approx(i32 x)
{
   int flip = x < 0;
   x = abs(x);
   i16 x14 = (i16)(x >> 15);
   i32 tmp = (762 * x14 >> 14) - 3308;
   tmp = (tmp * x14 >> 14) + 25726;
   tmp = tmp * oc_sqrt(imax(0, (1 << 30) - (x << 1))) >> 16;
   if (flip) tmp = 25736 - tmp;
   return tmp;
}
static int tone_lpc(const i16 *x, int len, int delay, i32 *lpc)
{
   i32 r00 = 0, r01 = 0, r11 = 0, r02 = 0, r12 = 0, r22 = 0, edges, num0, num1, den;
   for (int i = 0; i < len - 2 * delay; i++) {
      r00 += mult16_16(x[i], x[i]);
      r01 += mult16_16(x[i], x[i + delay]);
      r02 += mult16_16(x[i], x[i + 2 * delay]);
   }
   edges = 0;
   for (int i = 0; i < delay; i++) edges += mult16_16(x[len + i - 2 * delay], x[len + i - 2 * delay]) - mult16_16(x[i], x[i]);
   r11 = r00 + edges;
   edges = 0;
   for (int i = 0; i < delay; i++) edges += mult16_16(x[len + i - delay], x[len + i - delay]) - mult16_16(x[i + delay], x[i + delay]);
   r22 = r11 + edges;
   edges = 0;
   for (int i = 0; i < delay; i++) edges += mult16_16(x[len + i - 2 * delay], x[len + i - delay]) - mult16_16(x[i], x[i + delay]);
   r12 = r01 + edges;
   {
      i32 R00 = r00 + r22, R01 = r01 + r12, R11 = 2 * r11, R02 = 2 * r02, R12 = r12 + r01, R22 = r00 + r22;
      r00 = R00; r01 = R01; r11 = R11; r02 = R02; r12 = R12; r22 = R22;
   }
   (void)r22;
   den = mult32_32_q31(r00, r11) - mult32_32_q31(r01, r01);
   if (den <= (mult32_32_q31(r00, r11) >> 10)) return 1;
   num1 = mult32_32_q31(r02, r11) - mult32_32_q31(r01, r12);
   if (num1 >= den) lpc[1] = QC32(1.f, 29);
   else if (num1 <= -den) lpc[1] = -QC32(1.f, 29);
   else lpc[1] = oc_frac_div32_q29(num1, den);
   num0 = mult32_32_q31(r00, r12) - mult32_32_q31(r02, r01);
   if (half32(num0) >= den) lpc[0] = QC32(1.999999f, 29);
   else if (half32(num0) <= -den) lpc[0] = -QC32(1.999999f, 29);
   else lpc[0] = oc_frac_div32_q29(num0, den);
   return 0;
}
static i16 tone_detect(const i32 *in, int CC, int N, i32 *toneishness, i32 Fs)
{
   int delay = 1, fail;
   i32 lpc[2];
   i16 freq, x[1080];
   if (CC == 2) for (int i = 0; i < N; i++) x[i] = (i16)pshr32(add32(in[i] >> 1, in[i + N] >> 1), SIG_SHIFT + 2);
   else for (int i = 0; i < N; i++) x[i] = (i16)pshr32(in[i], SIG_SHIFT + 2);
   normalize_tone_input(x, N);
   fail = tone_lpc(x, N, delay, lpc);
   while (delay <= Fs / 3000 && (fail || (lpc[0] > QC32(1.f, 29) && lpc[1] < 0))) {
      delay *= 2;
      fail = tone_lpc(x, N, delay, lpc);
   }
   if (!fail && mult32_32_q31(lpc[0], lpc[0]) + mult32_32_q31(QC32(3.999999, 29), lpc[1]) < 0) {
      *toneishness = -lpc[1];
      freq = (i16)((acos_approx(lpc[0] >> 1) + delay / 2) / delay);
   } else { freq = -1; *toneishness = 0; }
   return freq;
}

static int run_prefilter(oc_celt_enc *st, i32 *in, i32 *prefilter_mem, int CC, int N, int prefilter_tapset, int *pitch, i16 *gain,
      int *qgain, int enabled, int complexity, i16 tf_estimate, int nbAvailableBytes, i16 tone_freq, i32 toneishness)
{
   const int max_period = COMBFILTER_MAXPERIOD, min_period = COMBFILTER_MINPERIOD, overlap = OVERLAP;
   i32 _pre[2 * (960 + COMBFILTER_MAXPERIOD)], *pre[2], before[2] = {0, 0}, after[2] = {0, 0};
   int pitch_index, pf_on, qg, cancel_pitch = 0;
   i16 gain1, pf_threshold;
   pre[0] = _pre; pre[1] = _pre + (N + max_period);
   for (int c = 0; c < CC; c++) {
      memcpy(pre[c], prefilter_mem + c * max_period, max_period * sizeof(i32));
      memcpy(pre[c] + max_period, in + c * (N + overlap) + overlap, N * sizeof(i32));
   }
   if (enabled && toneishness > QC32(.99f, 29)) {
      int multiple = 1;
      if (tone_freq >= QC16(3.1416f, 13)) tone_freq = (i16)(QC16(3.141593f, 13) - tone_freq);
      while (tone_freq >= multiple * QC16(0.39f, 13)) multiple++;
      if (tone_freq > QC16(0.006148f, 13)) pitch_index = imin((51472 * multiple + tone_freq / 2) / tone_freq, COMBFILTER_MAXPERIOD - 2);
      else pitch_index = COMBFILTER_MINPERIOD;
      gain1 = QC16(.75f, 15);
   } else if (enabled && complexity >= 5) {
      i16 pitch_buf[(COMBFILTER_MAXPERIOD + 960) >> 1];
      oc_pitch_downsample(pre, pitch_buf, (max_period + N) >> 1, CC, 2);
      oc_pitch_search(pitch_buf + (max_period >> 1), pitch_buf, N, max_period - 3 * min_period, &pitch_index);
      pitch_index = max_period - pitch_index;
      gain1 = oc_remove_doubling(pitch_buf, max_period, min_period, N, &pitch_index, st->prefilter_period, st->prefilter_gain);
      if (pitch_index > max_period - 2) pitch_index = max_period - 2;
      gain1 = (i16)mult16_16_q15(QC16(.7f, 15), gain1);
      if (st->loss_rate > 2) gain1 = (i16)(gain1 >> 1);
      if (st->loss_rate > 4) gain1 = (i16)(gain1 >> 1);
      if (st->loss_rate > 8) gain1 = 0;
   } else { gain1 = 0; pitch_index = COMBFILTER_MINPERIOD; }
   pf_threshold = QC16(.2f, 15);
   if (abs(pitch_index - st->prefilter_period) * 10 > pitch_index) {
      pf_threshold += QC16(.2f, 15);
      if (tf_estimate > QC16(.98f, 14)) gain1 = 0;
   }
   if (nbAvailableBytes < 25) pf_threshold += QC16(.1f, 15);
   if (nbAvailableBytes < 35) pf_threshold += QC16(.1f, 15);
   if (st->prefilter_gain > QC16(.4f, 15)) pf_threshold -= QC16(.1f, 15);
   if (st->prefilter_gain > QC16(.55f, 15)) pf_threshold -= QC16(.1f, 15);
   pf_threshold = (i16)imax(pf_threshold, QC16(.2f, 15));
   if (gain1 < pf_threshold) { gain1 = 0; pf_on = 0; qg = 0; }
   else {
      if (iabs(gain1 - st->prefilter_gain) < QC16(.1f, 15)) gain1 = st->prefilter_gain;
      qg = ((gain1 + 1536) >> 10) / 3 - 1;
      qg = imax(0, imin(7, qg));
      gain1 = (i16)(QC16(0.09375f, 15) * (qg + 1));
      pf_on = 1;
   }
   for (int c = 0; c < CC; c++) {
      int offset = SHORT_MDCT - overlap;
      st->prefilter_period = imax(st->prefilter_period, COMBFILTER_MINPERIOD);
      memcpy(in + c * (N + overlap), st->in_mem + c * overlap, overlap * sizeof(i32));
      for (int i = 0; i < N; i++) before[c] += iabs(in[c * (N + overlap) + overlap + i] >> 12);
      if (offset)
         oc_comb_filter(in + c * (N + overlap) + overlap, pre[c] + max_period, st->prefilter_period, st->prefilter_period, offset,
               (i16)-st->prefilter_gain, (i16)-st->prefilter_gain, st->prefilter_tapset, st->prefilter_tapset, 0);
      oc_comb_filter(in + c * (N + overlap) + overlap + offset, pre[c] + max_period + offset, st->prefilter_period, pitch_index,
            N - offset, (i16)-st->prefilter_gain, (i16)-gain1, st->prefilter_tapset, prefilter_tapset, overlap);
      for (int i = 0; i < N; i++) after[c] += iabs(in[c * (N + overlap) + overlap + i] >> 12);
   }
   if (CC == 2) {
      i16 thresh[2];
      thresh[0] = (i16)(mult16_32_q15(mult16_16_q15(QC16(.25f, 15), gain1), before[0]) + mult16_32_q15(QC16(.01f, 15), before[1]));
      thresh[1] = (i16)(mult16_32_q15(mult16_16_q15(QC16(.25f, 15), gain1), before[1]) + mult16_32_q15(QC16(.01f, 15), before[0]));
      if (after[0] - before[0] > thresh[0] || after[1] - before[1] > thresh[1]) cancel_pitch = 1;
      if (before[0] - after[0] < thresh[0] && before[1] - after[1] < thresh[1]) cancel_pitch = 1;
   } else if (after[0] > before[0]) cancel_pitch = 1;
   if (cancel_pitch) {
      for (int c = 0; c < CC; c++) {
         int offset = SHORT_MDCT - overlap;
         memcpy(in + c * (N + overlap) + overlap, pre[c] + max_period, N * sizeof(i32));
         oc_comb_filter(in + c * (N + overlap) + overlap + offset, pre[c] + max_period + offset, st->prefilter_period, pitch_index,
               overlap, (i16)-st->prefilter_gain, 0, st->prefilter_tapset, prefilter_tapset, overlap);
      }
      gain1 = 0; pf_on = 0; qg = 0;
   }
   for (int c = 0; c < CC; c++) {
      memcpy(st->in_mem + c * overlap, in + c * (N + overlap) + N, overlap * sizeof(i32));
      if (N > max_period) memcpy(prefilter_mem + c * max_period, pre[c] + N, max_period * sizeof(i32));
      else {
         memmove(prefilter_mem + c * max_period, prefilter_mem + c * max_period + N, (max_period - N) * sizeof(i32));
         memcpy(prefilter_mem + c * max_period + max_period - N, pre[c] + max_period, N * sizeof(i32));
      }
   }
   *gain = gain1; *pitch = pitch_index; *qgain = qg;
   return pf_on;
}

static i32 compute_vbr(i32 base_target, int LM, i32 bitrate, int lastCodedBands, int C, int intensity, int constrained_vbr,
      i16 stereo_saving, int tot_boost, i16 tf_estimate, i32 maxDepth, int lfe, int has_surround_mask, i32 surround_masking, i32 temporal_vbr)
{
   const int16_t *eB = oc_eBands;
   i32 target;
   int coded_bands = lastCodedBands ? lastCodedBands : NB_EBANDS;
   int coded_bins = eB[coded_bands] << LM;
   if (C == 2) coded_bins += eB[imin(intensity, coded_bands)] << LM;
   target = base_target;
   if (C == 2) {
      int coded_stereo_bands = imin(intensity, coded_bands);
      int coded_stereo_dof = (eB[coded_stereo_bands] << LM) - coded_stereo_bands;
      i16 max_frac = (i16)(mult16_16(QC16(0.8f, 15), coded_stereo_dof) / (i16)coded_bins);
      stereo_saving = (i16)imin(stereo_saving, QC16(1.f, 8));
      target -= (i32)imin(mult16_32_q15(max_frac, target), mult16_16(stereo_saving - QC16(0.1f, 8), (coded_stereo_dof << BITRES)) >> 8);
   }
   target += tot_boost - (19 << LM);
   i16 tf_calibration = QC16(0.044f, 14);
   target += (i32)shl32(mult16_32_q15(tf_estimate - tf_calibration, target), 1);
   if (has_surround_mask && !lfe) {
      i32 surround_target = target + (i32)(mult16_16(surround_masking >> (DB_SHIFT - 10), coded_bins << BITRES) >> 10);
      target = imax(target / 4, surround_target);
   }
   {
      int bins = eB[NB_EBANDS - 2] << LM;
      i32 floor_depth = (i32)(mult16_32_q15((C * bins << BITRES), maxDepth) >> (DB_SHIFT - 15));
      floor_depth = imax(floor_depth, target >> 2);
      target = imin(target, floor_depth);
   }
   if ((!has_surround_mask || lfe) && constrained_vbr) target = base_target + (i32)mult16_32_q15(QC16(0.67f, 15), target - base_target);
   if (!has_surround_mask && tf_estimate < QC16(.2f, 14)) {
      i16 amount = (i16)mult16_16_q15(QC16(.0000031f, 30), imax(0, imin(32000, 96000 - bitrate)));
      i16 tvbr_factor = (i16)(mult16_16(temporal_vbr >> (DB_SHIFT - 10), amount) >> 10);
      target += (i32)mult16_32_q15(tvbr_factor, target);
   }
   return imin(2 * base_target, target);
}

static int hysteresis_decision(i16 val, const i16 *thresholds, const i16 *hysteresis, int N, int prev)
{
   int i;
   for (i = 0; i < N; i++) if (val < thresholds[i]) break;
   if (i > prev && val < thresholds[prev] + hysteresis[prev]) i = prev;
   if (i < prev && val > thresholds[prev - 1] - hysteresis[prev - 1]) i = prev;
   return i;
}

/* celt_encode_with_ec, celt_encoder.c:1726.  `enc` may be NULL (then a private coder over `compressed` is used). */
int oc_celt_encode_with_ec(oc_celt_enc *st, const i16 *pcm, int frame_size, u8 *compressed, int nbCompressedBytes, oc_ec *enc)
{
   const int16_t *eBands = oc_eBands;
   const int nbEBands = NB_EBANDS, overlap = OVERLAP, Fs = 48000;
   const int CC = st->channels, C = st->stream_channels;
   oc_ec _enc;
   i32 in[2 * (960 + OVERLAP)], freq[2 * 960], X[2 * 960], bandE[2 * NB_EBANDS], bandLogE[2 * NB_EBANDS], bandLogE2[2 * NB_EBANDS];
   i32 error[2 * NB_EBANDS], surround_dynalloc[2 * NB_EBANDS];
   int fine_quant[NB_EBANDS], pulses[NB_EBANDS], cap[NB_EBANDS], offsets[NB_EBANDS], importance[NB_EBANDS], spread_weight[NB_EBANDS];
   int fine_priority[NB_EBANDS], tf_res[NB_EBANDS];
   u8 collapse_masks[2 * NB_EBANDS];
   i32 *prefilter_mem = st->prefilter_mem, *oldBandE = st->oldBandE, *oldLogE = st->oldLogE, *oldLogE2 = st->oldLogE2, *energyError = st->energyError;
   int shortBlocks = 0, isTransient = 0, LM, M, N, tf_select, nbFilledBytes, nbAvailableBytes, start = st->start, end = st->end, effEnd;
   int codedBands, alloc_trim, pitch_index = COMBFILTER_MINPERIOD, dual_stereo = 0, effectiveBytes, dynalloc_logp;
   i16 gain1 = 0, tf_estimate = 0, tone_freq = -1;
   i32 bits, min_allowed, vbr_rate, total_bits, total_boost, balance, tell, tell0_frac, tot_boost, sample_max, maxDepth, equiv_rate;
   i32 toneishness = 0, surround_masking = 0, temporal_vbr = 0, surround_trim = 0;
   int prefilter_tapset = 0, pf_on, anti_collapse_rsv, anti_collapse_on = 0, silence = 0, tf_chan = 0, pitch_change = 0, secondMdct;
   int signalBandwidth, transient_got_disabled = 0, hybrid = start != 0, weak_transient = 0, enable_tf_analysis;
   if (nbCompressedBytes < 2 || pcm == 0) return -1;
   for (LM = 0; LM <= MAX_LM; LM++) if (SHORT_MDCT << LM == frame_size) break;
   if (LM > MAX_LM) return -1;
   M = 1 << LM; N = M * SHORT_MDCT;
   if (enc == 0) { tell0_frac = tell = 1; nbFilledBytes = 0; }
   else { tell0_frac = oc_ec_tell_frac(enc); tell = oc_ec_tell(enc); nbFilledBytes = (tell + 4) >> 3; }
   nbCompressedBytes = imin(nbCompressedBytes, 1275);
   if (st->vbr && st->bitrate != -1) {
      vbr_rate = bitrate_to_bits(st->bitrate, Fs, frame_size) << BITRES;
      effectiveBytes = vbr_rate >> (3 + BITRES);
   } else {
      vbr_rate = 0;
      i32 tmp = st->bitrate * frame_size;
      if (tell > 1) tmp += tell * Fs;
      if (st->bitrate != -1) {
         nbCompressedBytes = imax(2, imin(nbCompressedBytes, (tmp + 4 * Fs) / (8 * Fs)));
         if (enc != 0) oc_ec_enc_shrink(enc, nbCompressedBytes);
      }
      effectiveBytes = nbCompressedBytes - nbFilledBytes;
   }
   nbAvailableBytes = nbCompressedBytes - nbFilledBytes;
   equiv_rate = ((i32)nbCompressedBytes * 8 * 50 << (3 - LM)) - (40 * C + 20) * ((400 >> LM) - 50);
   if (st->bitrate != -1) equiv_rate = imin(equiv_rate, st->bitrate - (40 * C + 20) * ((400 >> LM) - 50));
   if (enc == 0) { oc_ec_enc_init(&_enc, compressed, nbCompressedBytes); enc = &_enc; }
   if (vbr_rate > 0 && st->constrained_vbr) {
      i32 vbr_bound = vbr_rate;
      i32 max_allowed = imin(imax(tell == 1 ? 2 : 0, (vbr_rate + vbr_bound - st->vbr_reservoir) >> (BITRES + 3)), nbAvailableBytes);
      if (max_allowed < nbAvailableBytes) {
         nbCompressedBytes = nbFilledBytes + max_allowed;
         nbAvailableBytes = max_allowed;
         oc_ec_enc_shrink(enc, nbCompressedBytes);
      }
   }
   total_bits = nbCompressedBytes * 8;
   effEnd = end;
   sample_max = imax(st->overlap_max, maxabs16(pcm, CC * (N - overlap)));
   st->overlap_max = maxabs16(pcm + CC * (N - overlap), CC * overlap);
   sample_max = imax(sample_max, st->overlap_max);
   silence = (sample_max == 0);
   if (tell == 1) oc_ec_enc_bit_logp(enc, silence, 15);
   else silence = 0;
   if (silence) {
      if (vbr_rate > 0) {
         effectiveBytes = nbCompressedBytes = imin(nbCompressedBytes, nbFilledBytes + 2);
         total_bits = nbCompressedBytes * 8;
         nbAvailableBytes = 2;
         oc_ec_enc_shrink(enc, nbCompressedBytes);
      }
      tell = nbCompressedBytes * 8;
      enc->nbits_total += tell - oc_ec_tell(enc);
   }
   for (int c = 0; c < CC; c++) {
      preemphasis(pcm + c, in + c * (N + overlap) + overlap, N, CC, st->preemph_memE + c);
      memcpy(in + c * (N + overlap), &prefilter_mem[(1 + c) * COMBFILTER_MAXPERIOD - overlap], overlap * sizeof(i32));
   }
   tone_freq = tone_detect(in, CC, N + overlap, &toneishness, Fs);
   OC_DUMPI("tone_freq", tone_freq); OC_DUMPI("toneishness", toneishness);
   if (st->complexity >= 1 && !st->lfe) {
      int allow_weak_transients = hybrid && effectiveBytes < 15 && st->silk_signalType != 2;
      isTransient = transient_analysis(in, N + overlap, CC, &tf_estimate, &tf_chan, allow_weak_transients, &weak_transient, tone_freq, toneishness);
   }
   toneishness = imin(toneishness, QC32(1.f, 29) - shl32(tf_estimate, 15));
   OC_DUMPI("isTransient", isTransient); OC_DUMPI("tf_estimate", tf_estimate); OC_DUMPI("tf_chan", tf_chan);
   {
      int enabled, qg;
      enabled = ((st->lfe && nbAvailableBytes > 3) || nbAvailableBytes > 12 * C) && !hybrid && !silence && tell + 16 <= total_bits && !st->disable_pf;
      prefilter_tapset = st->tapset_decision;
      pf_on = run_prefilter(st, in, prefilter_mem, CC, N, prefilter_tapset, &pitch_index, &gain1, &qg, enabled, st->complexity, tf_estimate,
            nbAvailableBytes, tone_freq, toneishness);
      OC_DUMPI("pf_on", pf_on); OC_DUMPI("pitch_index", pitch_index); OC_DUMPI("gain1", gain1); OC_DUMPI("qg", qg);
      if ((gain1 > QC16(.4f, 15) || st->prefilter_gain > QC16(.4f, 15))
            && (pitch_index > 1.26 * st->prefilter_period || pitch_index < .79 * st->prefilter_period))
         pitch_change = 1;
      if (pf_on == 0) {
         if (!hybrid && tell + 16 <= total_bits) oc_ec_enc_bit_logp(enc, 0, 1);
      } else {
         int octave;
         oc_ec_enc_bit_logp(enc, 1, 1);
         pitch_index += 1;
         octave = ec_ilog(pitch_index) - 5;
         oc_ec_enc_uint(enc, octave, 6);
         oc_ec_enc_bits(enc, pitch_index - (16 << octave), 4 + octave);
         pitch_index -= 1;
         oc_ec_enc_bits(enc, qg, 3);
         oc_ec_enc_icdf(enc, prefilter_tapset, tapset_icdf, 2);
      }
   }
   (void)pitch_change;
   if (LM > 0 && oc_ec_tell(enc) + 3 <= total_bits) { if (isTransient) shortBlocks = M; }
   else { isTransient = 0; transient_got_disabled = 1; }
   secondMdct = shortBlocks && st->complexity >= 8;
   if (secondMdct) {
      compute_mdcts(0, in, freq, C, CC, LM);
      oc_compute_band_energies(freq, bandE, effEnd, C, LM);
      oc_amp2log2(effEnd, end, bandE, bandLogE2, C);
      for (int c = 0; c < C; c++) for (int i = 0; i < end; i++) bandLogE2[nbEBands * c + i] += half32(shl32(LM, DB_SHIFT));
   }
   compute_mdcts(shortBlocks, in, freq, C, CC, LM);
   if (CC == 2 && C == 1) tf_chan = 0;
   oc_compute_band_energies(freq, bandE, effEnd, C, LM);
   if (st->lfe) for (int i = 2; i < end; i++) { bandE[i] = imin(bandE[i], mult16_32_q15(QC16(1e-4f, 15), bandE[0])); bandE[i] = imax(bandE[i], EPSILON); }
   oc_amp2log2(effEnd, end, bandE, bandLogE, C);
   OC_DUMPI("shortBlocks", shortBlocks); OC_DUMP("freq", freq, C * N * 4); OC_DUMP("bandE", bandE, 42 * 4); OC_DUMP("bandLogE", bandLogE, 42 * 4);
   memset(surround_dynalloc, 0, sizeof(surround_dynalloc));
   if (!st->lfe) {
      i32 follow = -QC32(10.0f, DB_SHIFT - 5), frame_avg = 0, offset = shortBlocks ? half32(shl32(LM, DB_SHIFT - 5)) : 0;
      for (int i = start; i < end; i++) {
         follow = imax(follow - QC32(1.0f, DB_SHIFT - 5), (bandLogE[i] >> 5) - offset);
         if (C == 2) follow = imax(follow, (bandLogE[i + nbEBands] >> 5) - offset);
         frame_avg += follow;
      }
      frame_avg /= (end - start);
      temporal_vbr = sub32(shl32(frame_avg, 5), st->spec_avg);
      temporal_vbr = imin(GC(3.f), imax(-GC(1.5f), temporal_vbr));
      st->spec_avg += mult16_32_q15(QC16(.02f, 15), temporal_vbr);
   }
   if (!secondMdct) memcpy(bandLogE2, bandLogE, C * nbEBands * sizeof(i32));
   if (LM > 0 && oc_ec_tell(enc) + 3 <= total_bits && !isTransient && st->complexity >= 5 && !st->lfe && !hybrid) {
      if (patch_transient_decision(bandLogE, oldBandE, start, end, C)) {
         isTransient = 1;
         shortBlocks = M;
         compute_mdcts(shortBlocks, in, freq, C, CC, LM);
         oc_compute_band_energies(freq, bandE, effEnd, C, LM);
         oc_amp2log2(effEnd, end, bandE, bandLogE, C);
         for (int c = 0; c < C; c++) for (int i = 0; i < end; i++) bandLogE2[nbEBands * c + i] += half32(shl32(LM, DB_SHIFT));
         tf_estimate = QC16(.2f, 14);
      }
   }
   if (LM > 0 && oc_ec_tell(enc) + 3 <= total_bits) oc_ec_enc_bit_logp(enc, isTransient, 3);
   oc_normalise_bands(freq, X, bandE, effEnd, C, M);
   OC_DUMPI("isTransient2", isTransient); OC_DUMP("bandLogE2", bandLogE2, 42 * 4); for (int c = 0; c < C; c++) OC_DUMP("X", X + c * N, M * eBands[effEnd] * 4); OC_DUMPI("temporal_vbr", temporal_vbr);
   enable_tf_analysis = effectiveBytes >= 15 * C && !hybrid && st->complexity >= 2 && !st->lfe && toneishness < QC32(.98f, 29);
   maxDepth = dynalloc_analysis(bandLogE, bandLogE2, oldBandE, start, end, C, offsets, st->lsb_depth, isTransient, st->vbr, st->constrained_vbr,
         LM, effectiveBytes, &tot_boost, st->lfe, surround_dynalloc, importance, spread_weight, tone_freq, toneishness);
   OC_DUMPI("maxDepth", maxDepth); OC_DUMPI("tot_boost", tot_boost); OC_DUMP("offsets", offsets, 84); OC_DUMP("importance", importance, 84); OC_DUMP("spread_weight", spread_weight, 84);
   if (enable_tf_analysis) {
      int lambda = imax(80, 20480 / effectiveBytes + 2);
      tf_select = tf_analysis(effEnd, isTransient, tf_res, lambda, X, N, LM, tf_estimate, tf_chan, importance);
      for (int i = effEnd; i < end; i++) tf_res[i] = tf_res[effEnd - 1];
   } else if (hybrid && weak_transient) {
      for (int i = 0; i < end; i++) tf_res[i] = 1;
      tf_select = 0;
   } else if (hybrid && effectiveBytes < 15 && st->silk_signalType != 2) {
      for (int i = 0; i < end; i++) tf_res[i] = 0;
      tf_select = isTransient;
   } else {
      for (int i = 0; i < end; i++) tf_res[i] = isTransient;
      tf_select = 0;
   }
   for (int c = 0; c < C; c++)
      for (int i = start; i < end; i++)
         if (iabs(sub32(bandLogE[i + c * nbEBands], oldBandE[i + c * nbEBands])) < GC(2.f))
            bandLogE[i + c * nbEBands] -= mult16_32_q15(QC16(0.25f, 15), energyError[i + c * nbEBands]);
   oc_quant_coarse_energy(start, end, effEnd, bandLogE, oldBandE, total_bits, error, enc, C, LM, nbAvailableBytes, st->force_intra,
         &st->delayedIntra, st->complexity >= 4, st->loss_rate, st->lfe);
   tf_encode(start, end, isTransient, tf_res, LM, tf_select, enc);
   OC_DUMP("tf_res", tf_res, 84); OC_DUMP("oldBandE_c", oldBandE, 168); OC_DUMP("error_c", error, 168); OC_DUMPI("rng_tf", enc->rng); OC_DUMPI("tell_tf", oc_ec_tell_frac(enc));
   if (oc_ec_tell(enc) + 4 <= total_bits) {
      if (st->lfe) { st->tapset_decision = 0; st->spread_decision = SPREAD_NORMAL; }
      else if (hybrid) {
         if (st->complexity == 0) st->spread_decision = SPREAD_NONE;
         else if (isTransient) st->spread_decision = SPREAD_NORMAL;
         else st->spread_decision = SPREAD_AGGRESSIVE;
      } else if (shortBlocks || st->complexity < 3 || nbAvailableBytes < 10 * C) {
         st->spread_decision = st->complexity == 0 ? SPREAD_NONE : SPREAD_NORMAL;
      } else {
         st->spread_decision = oc_spreading_decision(X, &st->tonal_average, st->spread_decision, &st->hf_average, &st->tapset_decision,
               pf_on && !shortBlocks, effEnd, C, M, spread_weight);
      }
      oc_ec_enc_icdf(enc, st->spread_decision, spread_icdf, 5);
   } else st->spread_decision = SPREAD_NORMAL;
   if (st->lfe) offsets[0] = imin(8, effectiveBytes / 3);
   oc_init_caps(cap, LM, C);
   OC_DUMPI("spread", st->spread_decision); OC_DUMPI("tapset", st->tapset_decision);
   dynalloc_logp = 6;
   total_bits <<= BITRES;
   total_boost = 0;
   tell = oc_ec_tell_frac(enc);
   for (int i = start; i < end; i++) {
      int width = C * (eBands[i + 1] - eBands[i]) << LM;
      int quanta = imin(width << BITRES, imax(6 << BITRES, width));
      int dynalloc_loop_logp = dynalloc_logp, boost = 0, j;
      for (j = 0; tell + (dynalloc_loop_logp << BITRES) < total_bits - total_boost && boost < cap[i]; j++) {
         int flag = j < offsets[i];
         oc_ec_enc_bit_logp(enc, flag, dynalloc_loop_logp);
         tell = oc_ec_tell_frac(enc);
         if (!flag) break;
         boost += quanta;
         total_boost += quanta;
         dynalloc_loop_logp = 1;
      }
      if (j) dynalloc_logp = imax(2, dynalloc_logp - 1);
      offsets[i] = boost;
   }
   if (C == 2) {
      static const i16 intensity_thresholds[21] = {1, 2, 3, 4, 5, 6, 7, 8, 16, 24, 36, 44, 50, 56, 62, 67, 72, 79, 88, 106, 134};
      static const i16 intensity_histeresis[21] = {1, 1, 1, 1, 1, 1, 1, 2, 2, 2, 2, 2, 2, 2, 3, 3, 4, 5, 6, 8, 8};
      if (LM != 0) dual_stereo = stereo_analysis(X, LM, N);
      st->intensity = hysteresis_decision((i16)(equiv_rate / 1000), intensity_thresholds, intensity_histeresis, 21, st->intensity);
      st->intensity = imin(end, imax(start, st->intensity));
   }
   alloc_trim = 5;
   if (tell + (6 << BITRES) <= total_bits - total_boost) {
      if (start > 0 || st->lfe) { st->stereo_saving = 0; alloc_trim = 5; }
      else alloc_trim = alloc_trim_analysis(X, bandLogE, end, LM, C, N, &st->stereo_saving, tf_estimate, st->intensity, surround_trim, equiv_rate);
      oc_ec_enc_icdf(enc, alloc_trim, trim_icdf, 7);
      tell = oc_ec_tell_frac(enc);
   }
   min_allowed = ((tell + total_boost + (1 << (BITRES + 3)) - 1) >> (BITRES + 3)) + 2;
   OC_DUMPI("alloc_trim", alloc_trim); OC_DUMPI("dual_stereo", dual_stereo); OC_DUMPI("intensity", st->intensity); OC_DUMP("offsets2", offsets, 84); OC_DUMPI("rng_trim", enc->rng);
   if (hybrid) min_allowed = imax(min_allowed, (tell0_frac + (37 << BITRES) + total_boost + (1 << (BITRES + 3)) - 1) >> (BITRES + 3));
   if (vbr_rate > 0) {
      i16 alpha;
      i32 delta, target, base_target;
      int lm_diff = MAX_LM - LM;
      nbCompressedBytes = imin(nbCompressedBytes, 1275 >> (3 - LM));
      if (!hybrid) base_target = vbr_rate - ((40 * C + 20) << BITRES);
      else base_target = imax(0, vbr_rate - ((9 * C + 4) << BITRES));
      if (st->constrained_vbr) base_target += (st->vbr_offset >> lm_diff);
      if (!hybrid) {
         target = compute_vbr(base_target, LM, equiv_rate, st->lastCodedBands, C, st->intensity, st->constrained_vbr, st->stereo_saving,
               tot_boost, tf_estimate, maxDepth, st->lfe, 0, surround_masking, temporal_vbr);
      } else {
         target = base_target;
         if (st->silk_offset < 100) target += 12 << BITRES >> (3 - LM);
         if (st->silk_offset > 100) target -= 18 << BITRES >> (3 - LM);
         target += (i32)mult16_16_q14(tf_estimate - QC16(.25f, 14), (50 << BITRES));
         if (tf_estimate > QC16(.7f, 14)) target = imax(target, 50 << BITRES);
      }
      target = target + tell;
      nbAvailableBytes = (target + (1 << (BITRES + 2))) >> (BITRES + 3);
      nbAvailableBytes = imax(min_allowed, nbAvailableBytes);
      nbAvailableBytes = imin(nbCompressedBytes, nbAvailableBytes);
      delta = target - vbr_rate;
      target = nbAvailableBytes << (BITRES + 3);
      if (silence) { nbAvailableBytes = 2; target = 2 * 8 << BITRES; delta = 0; }
      if (st->vbr_count < 970) { st->vbr_count++; alpha = (i16)oc_rcp(shl32((i32)(st->vbr_count + 20), 16)); }
      else alpha = QC16(.001f, 15);
      if (st->constrained_vbr) st->vbr_reservoir += target - vbr_rate;
      if (st->constrained_vbr) {
         st->vbr_drift += (i32)mult16_32_q15(alpha, (delta * (1 << lm_diff)) - st->vbr_offset - st->vbr_drift);
         st->vbr_offset = -st->vbr_drift;
      }
      if (st->constrained_vbr && st->vbr_reservoir < 0) {
         int adjust = (-st->vbr_reservoir) / (8 << BITRES);
         nbAvailableBytes += silence ? 0 : adjust;
         st->vbr_reservoir = 0;
      }
      nbCompressedBytes = imin(nbCompressedBytes, nbAvailableBytes);
      oc_ec_enc_shrink(enc, nbCompressedBytes);
   }
   bits = (((i32)nbCompressedBytes * 8) << BITRES) - (i32)oc_ec_tell_frac(enc) - 1;
   anti_collapse_rsv = isTransient && LM >= 2 && bits >= ((LM + 2) << BITRES) ? (1 << BITRES) : 0;
   bits -= anti_collapse_rsv;
   signalBandwidth = end - 1;
   if (st->lfe) signalBandwidth = 1;
   codedBands = oc_compute_allocation(start, end, offsets, cap, alloc_trim, &st->intensity, &dual_stereo, bits, &balance, pulses,
         fine_quant, fine_priority, C, LM, enc, 1, st->lastCodedBands, signalBandwidth);
   if (st->lastCodedBands) st->lastCodedBands = imin(st->lastCodedBands + 1, imax(st->lastCodedBands - 1, codedBands));
   else st->lastCodedBands = codedBands;
   oc_quant_fine_energy(start, end, oldBandE, error, 0, fine_quant, enc, C);
   OC_DUMPI("nbCompressedBytes", nbCompressedBytes); OC_DUMPI("codedBands", codedBands); OC_DUMPI("balance", balance); OC_DUMP("pulses", pulses, 84); OC_DUMP("fine_quant", fine_quant, 84); OC_DUMP("fine_priority", fine_priority, 84); OC_DUMPI("rng_fine", enc->rng);
   memset(energyError, 0, nbEBands * CC * sizeof(i32));
   oc_quant_all_bands(1, start, end, X, C == 2 ? X + N : 0, collapse_masks, bandE, pulses, shortBlocks, st->spread_decision, dual_stereo,
         st->intensity, tf_res, nbCompressedBytes * (8 << BITRES) - anti_collapse_rsv, balance, enc, LM, codedBands, &st->rng,
         st->complexity, st->disable_inv);
   OC_DUMPI("rng_pvq", enc->rng); OC_DUMP("collapse", collapse_masks, 42);
   if (anti_collapse_rsv > 0) {
      anti_collapse_on = st->consec_transient < 2;
      oc_ec_enc_bits(enc, anti_collapse_on, 1);
   }
   oc_quant_energy_finalise(start, end, oldBandE, error, fine_quant, fine_priority, nbCompressedBytes * 8 - oc_ec_tell(enc), enc, C);
   for (int c = 0; c < C; c++)
      for (int i = start; i < end; i++) energyError[i + c * nbEBands] = imax(-GC(0.5f), imin(GC(0.5f), error[i + c * nbEBands]));
   if (silence) for (int i = 0; i < C * nbEBands; i++) oldBandE[i] = -GC(28.f);
   st->prefilter_period = pitch_index;
   st->prefilter_gain = gain1;
   st->prefilter_tapset = prefilter_tapset;
   if (CC == 2 && C == 1) memcpy(&oldBandE[nbEBands], oldBandE, nbEBands * sizeof(i32));
   if (!isTransient) {
      memcpy(oldLogE2, oldLogE, CC * nbEBands * sizeof(i32));
      memcpy(oldLogE, oldBandE, CC * nbEBands * sizeof(i32));
   } else for (int i = 0; i < CC * nbEBands; i++) oldLogE[i] = imin(oldLogE[i], oldBandE[i]);
   for (int c = 0; c < CC; c++) {
      for (int i = 0; i < start; i++) { oldBandE[c * nbEBands + i] = 0; oldLogE[c * nbEBands + i] = oldLogE2[c * nbEBands + i] = -GC(28.f); }
      for (int i = end; i < nbEBands; i++) { oldBandE[c * nbEBands + i] = 0; oldLogE[c * nbEBands + i] = oldLogE2[c * nbEBands + i] = -GC(28.f); }
   }
   if (isTransient || transient_got_disabled) st->consec_transient++;
   else st->consec_transient = 0;
   st->rng = enc->rng;
   oc_ec_enc_done(enc);
   if (enc->error) return -3;
   return nbCompressedBytes;
}

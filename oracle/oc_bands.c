/* oc_bands.c — band energies/normalisation and the recursive PVQ band quantiser (encoder side).
 * Oracle restatement of celt/bands.c: :61 lcg, :68/:80 bitexact_cos/log2tan, :95 compute_band_energies,
 * :125 normalise_bands, :362 compute_channel_weights, :379 intensity_stereo, :405 stereo_split,
 * :418 stereo_merge, :470 spreading_decision, :574/:600 (de)interleave_hadamard, :623 haar1,
 * :638 compute_qn, :700 compute_theta, :930 quant_band_n1, :973 quant_partition, :1248 quant_band,
 * :1387 quant_band_stereo, :1575 special_hybrid_folding, :1589 quant_all_bands (incl. theta-RDO). */
#include "oc_celt.h"
#include <stdlib.h>
extern void (*oc_dump_hook)(const char *tag, const void *p, int nbytes);

i32 oc_inner_prod_norm_shift(const i32 *x, const i32 *y, int len);

static u32 lcg_rand(u32 seed) { return 1664525u * seed + 1013904223u; }

int oc_bitexact_cos(int x_)
{
   i16 x = (i16)x_;
   i32 tmp = (4096 + ((i32)x * x)) >> 13;
   i16 x2 = (i16)tmp;
   x2 = (i16)((32767 - x2) + frac_mul16(x2, (-7651 + frac_mul16(x2, (8277 + frac_mul16(-626, x2))))));
   return (i16)(1 + x2);
}
int oc_bitexact_log2tan(int isin, int icos)
{
   int lc = ec_ilog(icos), ls = ec_ilog(isin);
   icos <<= 15 - lc;
   isin <<= 15 - ls;
   return (ls - lc) * (1 << 11) + frac_mul16(isin, frac_mul16(isin, -2597) + 7932) - frac_mul16(icos, frac_mul16(icos, -2597) + 7932);
}

static i32 maxabs32(const i32 *x, int len)
{
   i32 mx = 0, mn = 0;
   for (int i = 0; i < len; i++) { mx = imax(mx, x[i]); mn = imin(mn, x[i]); }
   return imax(mx, neg32(mn));
}

void oc_compute_band_energies(const i32 *X, i32 *bandE, int end, int C, int LM)
{
   const int16_t *eB = oc_eBands;
   int N = SHORT_MDCT << LM;
   for (int c = 0; c < C; c++)
      for (int i = 0; i < end; i++) {
         i32 sum = 0;
         i32 maxval = maxabs32(&X[c * N + (eB[i] << LM)], (eB[i + 1] - eB[i]) << LM);
         if (maxval > 0) {
            int shift = imax(0, 30 - celt_ilog2(maxval + (maxval >> 14) + 1) - ((((oc_logN[i] + 7) >> BITRES) + LM + 1) >> 1));
            for (int j = eB[i] << LM; j < eB[i + 1] << LM; j++) {
               i32 x = shl32(X[j + c * N], shift);
               sum = add32(sum, mult32_32_q31(x, x));
            }
            bandE[i + c * NB_EBANDS] = imax(maxval, pshr32(oc_sqrt32(sum >> 1), shift));
         } else bandE[i + c * NB_EBANDS] = EPSILON;
      }
}

void oc_normalise_bands(const i32 *freq, i32 *X, const i32 *bandE, int end, int C, int M)
{
   const int16_t *eB = oc_eBands;
   int N = M * SHORT_MDCT;
   for (int c = 0; c < C; c++)
      for (int i = 0; i < end; i++) {
         i32 E = bandE[i + c * NB_EBANDS];
         if (E < 10) E += EPSILON;
         int shift = 30 - celt_zlog2(E);
         E = shl32(E, shift);
         i32 g = oc_rcp_norm32(E);
         for (int j = M * eB[i]; j < M * eB[i + 1]; j++)
            X[j + c * N] = pshr32(mult32_32_q31(g, shl32(freq[j + c * N], shift)), 30 - NORM_SHIFT);
      }
}

int oc_spreading_decision(const i32 *X, int *average, int last_decision, int *hf_average,
      int *tapset_decision, int update_hf, int end, int C, int M, const int *spread_weight)
{
   const int16_t *eB = oc_eBands;
   int sum = 0, nbBands = 0, hf_sum = 0, decision, N0 = M * SHORT_MDCT;
   if (M * (eB[end] - eB[end - 1]) <= 8) return SPREAD_NONE;
   for (int c = 0; c < C; c++)
      for (int i = 0; i < end; i++) {
         int tcount[3] = {0, 0, 0};
         const i32 *x = X + M * eB[i] + c * N0;
         int N = M * (eB[i + 1] - eB[i]);
         if (N <= 8) continue;
         for (int j = 0; j < N; j++) {
            i32 x2N = mult16_16(mult16_16_q15(x[j] >> (NORM_SHIFT - 14), x[j] >> (NORM_SHIFT - 14)), N);
            if (x2N < QC16(0.25f, 13)) tcount[0]++;
            if (x2N < QC16(0.0625f, 13)) tcount[1]++;
            if (x2N < QC16(0.015625f, 13)) tcount[2]++;
         }
         if (i > NB_EBANDS - 4) hf_sum += (u32)(32 * (tcount[1] + tcount[0])) / (u32)N;
         int tmp = (2 * tcount[2] >= N) + (2 * tcount[1] >= N) + (2 * tcount[0] >= N);
         sum += tmp * spread_weight[i];
         nbBands += spread_weight[i];
      }
   if (update_hf) {
      if (hf_sum) hf_sum = (u32)hf_sum / (u32)(C * (4 - NB_EBANDS + end));
      *hf_average = (*hf_average + hf_sum) >> 1;
      hf_sum = *hf_average;
      if (*tapset_decision == 2) hf_sum += 4;
      else if (*tapset_decision == 0) hf_sum -= 4;
      if (hf_sum > 22) *tapset_decision = 2;
      else if (hf_sum > 18) *tapset_decision = 1;
      else *tapset_decision = 0;
   }
   sum = (u32)((i32)sum << 8) / (u32)nbBands;
   sum = (sum + *average) >> 1;
   *average = sum;
   sum = (3 * sum + (((3 - last_decision) << 7) + 64) + 2) >> 2;
   if (sum < 80) decision = SPREAD_AGGRESSIVE;
   else if (sum < 256) decision = SPREAD_NORMAL;
   else if (sum < 384) decision = SPREAD_LIGHT;
   else decision = SPREAD_NONE;
   return decision;
}

static const int ordery_table[] = {
   1, 0,
   3, 0, 2, 1,
   7, 0, 4, 3, 6, 1, 5, 2,
   15, 0, 8, 7, 12, 3, 11, 4, 14, 1, 9, 6, 13, 2, 10, 5,
};
static void deinterleave_hadamard(i32 *X, int N0, int stride, int hadamard)
{
   i32 tmp[176];
   int N = N0 * stride;
   if (hadamard) {
      const int *ordery = ordery_table + stride - 2;
      for (int i = 0; i < stride; i++) for (int j = 0; j < N0; j++) tmp[ordery[i] * N0 + j] = X[j * stride + i];
   } else
      for (int i = 0; i < stride; i++) for (int j = 0; j < N0; j++) tmp[i * N0 + j] = X[j * stride + i];
   memcpy(X, tmp, N * sizeof(i32));
}
static void interleave_hadamard(i32 *X, int N0, int stride, int hadamard)
{
   i32 tmp[176];
   int N = N0 * stride;
   if (hadamard) {
      const int *ordery = ordery_table + stride - 2;
      for (int i = 0; i < stride; i++) for (int j = 0; j < N0; j++) tmp[j * stride + i] = X[ordery[i] * N0 + j];
   } else
      for (int i = 0; i < stride; i++) for (int j = 0; j < N0; j++) tmp[j * stride + i] = X[i * N0 + j];
   memcpy(X, tmp, N * sizeof(i32));
}
void oc_haar1(i32 *X, int N0, int stride)
{
   N0 >>= 1;
   for (int i = 0; i < stride; i++)
      for (int j = 0; j < N0; j++) {
         i32 t1 = mult32_32_q31(QC32(.70710678f, 31), X[stride * 2 * j + i]);
         i32 t2 = mult32_32_q31(QC32(.70710678f, 31), X[stride * (2 * j + 1) + i]);
         X[stride * 2 * j + i] = add32(t1, t2);
         X[stride * (2 * j + 1) + i] = sub32(t1, t2);
      }
}
static int compute_qn(int N, int b, int offset, int pulse_cap, int stereo)
{
   static const i16 exp2_table8[8] = {16384, 17866, 19483, 21247, 23170, 25267, 27554, 30048};
   int qn, qb, N2 = 2 * N - 1;
   if (stereo && N == 2) N2--;
   qb = (b + N2 * offset) / N2;
   qb = imin(b - pulse_cap - (4 << BITRES), qb);
   qb = imin(8 << BITRES, qb);
   if (qb < (1 << BITRES >> 1)) qn = 1;
   else {
      qn = exp2_table8[qb & 0x7] >> (14 - (qb >> BITRES));
      qn = (qn + 1) >> 1 << 1;
   }
   return qn;
}

typedef struct {
   int encode, resynth, i, intensity, spread, tf_change;
   oc_ec *ec;
   i32 remaining_bits;
   const i32 *bandE;
   u32 seed;
   int theta_round, disable_inv, avoid_split_noise;
} band_ctx;
typedef struct { int inv, imid, iside, delta, itheta, qalloc; } split_ctx;

static void compute_channel_weights(i32 Ex, i32 Ey, i16 w[2])
{
   i32 minE = imin(Ex, Ey);
   Ex = add32(Ex, minE / 3);
   Ey = add32(Ey, minE / 3);
   int shift = celt_ilog2(EPSILON + imax(Ex, Ey)) - 14;
   w[0] = (i16)vshr32(Ex, shift);
   w[1] = (i16)vshr32(Ey, shift);
}
static void intensity_stereo(i32 *X, const i32 *Y, const i32 *bandE, int bandID, int N)
{
   int i = bandID;
   int shift = celt_zlog2(imax(bandE[i], bandE[i + NB_EBANDS])) - 13;
   i16 left = (i16)vshr32(bandE[i], shift), right = (i16)vshr32(bandE[i + NB_EBANDS], shift);
   i16 norm = (i16)(EPSILON + oc_sqrt(EPSILON + mult16_16(left, left) + mult16_16(right, right)));
   left = (i16)imin(left, norm - 1);
   right = (i16)imin(right, norm - 1);
   i16 a1 = (i16)(shl32((i32)left, 15) / norm), a2 = (i16)(shl32((i32)right, 15) / norm);
   for (int j = 0; j < N; j++) X[j] = add32(mult16_32_q15(a1, X[j]), mult16_32_q15(a2, Y[j]));
}
static void stereo_split(i32 *X, i32 *Y, int N)
{
   for (int j = 0; j < N; j++) {
      i32 l = mult32_32_q31(QC32(.70710678f, 31), X[j]);
      i32 r = mult32_32_q31(QC32(.70710678f, 31), Y[j]);
      X[j] = add32(l, r);
      Y[j] = sub32(r, l);
   }
}
static void stereo_merge(i32 *X, i32 *Y, i32 mid, int N)
{
   i32 xp = oc_inner_prod_norm_shift(Y, X, N), side = oc_inner_prod_norm_shift(Y, Y, N);
   xp = mult32_32_q31(mid, xp);
   i32 El = (mult32_32_q31(mid, mid) >> 3) + side - 2 * xp;
   i32 Er = (mult32_32_q31(mid, mid) >> 3) + side + 2 * xp;
   if (Er < QC32(6e-4f, 28) || El < QC32(6e-4f, 28)) { memcpy(Y, X, N * sizeof(i32)); return; }
   int kl = celt_ilog2(El) >> 1, kr = celt_ilog2(Er) >> 1;
   i32 t = vshr32(El, (kl << 1) - 29);
   i32 lgain = oc_rsqrt_norm32(t);
   t = vshr32(Er, (kr << 1) - 29);
   i32 rgain = oc_rsqrt_norm32(t);
   if (kl < 7) kl = 7;
   if (kr < 7) kr = 7;
   for (int j = 0; j < N; j++) {
      i32 l = mult32_32_q31(mid, X[j]), r = Y[j];
      X[j] = vshr32(mult32_32_q31(lgain, sub32(l, r)), kl - 15);
      Y[j] = vshr32(mult32_32_q31(rgain, add32(l, r)), kr - 15);
   }
}

static void compute_theta(band_ctx *ctx, split_ctx *sctx, i32 *X, i32 *Y, int N, int *b, int B, int B0, int LM, int stereo, int *fill)
{
   int qn, itheta = 0, delta, imid, iside, qalloc, pulse_cap, offset, inv = 0;
   int encode = ctx->encode, i = ctx->i, intensity = ctx->intensity;
   oc_ec *ec = ctx->ec;
   const i32 *bandE = ctx->bandE;
   pulse_cap = oc_logN[i] + LM * (1 << BITRES);
   offset = (pulse_cap >> 1) - (stereo && N == 2 ? QTHETA_OFFSET_TWOPHASE : QTHETA_OFFSET);
   qn = compute_qn(N, *b, offset, pulse_cap, stereo);
   if (stereo && i >= intensity) qn = 1;
   if (encode) {
      i32 itheta_q30 = oc_stereo_itheta(X, Y, stereo, N);
      itheta = itheta_q30 >> 16;
   }
   i32 tell = oc_ec_tell_frac(ec);
   if (qn != 1) {
      if (encode) {
         if (!stereo || ctx->theta_round == 0) {
            itheta = (itheta * (i32)qn + 8192) >> 14;
            if (!stereo && ctx->avoid_split_noise && itheta > 0 && itheta < qn) {
               int unquantized = (u32)((i32)itheta * 16384) / (u32)qn;
               imid = oc_bitexact_cos((i16)unquantized);
               iside = oc_bitexact_cos((i16)(16384 - unquantized));
               delta = frac_mul16((N - 1) << 7, oc_bitexact_log2tan(iside, imid));
               if (delta > *b) itheta = qn;
               else if (delta < -*b) itheta = 0;
            }
         } else {
            int bias = itheta > 8192 ? 32767 / qn : -32767 / qn;
            int down = imin(qn - 1, imax(0, (itheta * (i32)qn + bias) >> 14));
            itheta = ctx->theta_round < 0 ? down : down + 1;
         }
      }
      if (stereo && N > 2) {
         int p0 = 3, x = itheta, x0 = qn / 2, ft = p0 * (x0 + 1) + x0;
         if (encode) oc_ec_encode(ec, x <= x0 ? p0 * x : (x - 1 - x0) + (x0 + 1) * p0, x <= x0 ? p0 * (x + 1) : (x - x0) + (x0 + 1) * p0, ft);
         else {
            int fs = oc_ec_decode(ec, ft);
            if (fs < (x0 + 1) * p0) x = fs / p0;
            else x = x0 + 1 + (fs - (x0 + 1) * p0);
            oc_ec_dec_update(ec, x <= x0 ? p0 * x : (x - 1 - x0) + (x0 + 1) * p0, x <= x0 ? p0 * (x + 1) : (x - x0) + (x0 + 1) * p0, ft);
            itheta = x;
         }
      } else if (B0 > 1 || stereo) {
         if (encode) oc_ec_enc_uint(ec, itheta, qn + 1);
         else itheta = oc_ec_dec_uint(ec, qn + 1);
      } else {
         int fs, ft = ((qn >> 1) + 1) * ((qn >> 1) + 1);
         if (encode) {
            fs = itheta <= (qn >> 1) ? itheta + 1 : qn + 1 - itheta;
            int fl = itheta <= (qn >> 1) ? itheta * (itheta + 1) >> 1 : ft - ((qn + 1 - itheta) * (qn + 2 - itheta) >> 1);
            oc_ec_encode(ec, fl, fl + fs, ft);
         } else {                        /* triangular pdf (bands.c:822) */
            int fl = 0, fm = oc_ec_decode(ec, ft);
            if (fm < ((qn >> 1) * ((qn >> 1) + 1) >> 1)) {
               itheta = (oc_isqrt32(8 * (u32)fm + 1) - 1) >> 1;
               fs = itheta + 1;
               fl = itheta * (itheta + 1) >> 1;
            } else {
               itheta = (2 * (qn + 1) - oc_isqrt32(8 * (u32)(ft - fm - 1) + 1)) >> 1;
               fs = qn + 1 - itheta;
               fl = ft - ((qn + 1 - itheta) * (qn + 2 - itheta) >> 1);
            }
            oc_ec_dec_update(ec, fl, fl + fs, ft);
         }
      }
      itheta = (u32)((i32)itheta * 16384) / (u32)qn;
      if (encode && stereo) {
         if (itheta == 0) intensity_stereo(X, Y, bandE, i, N);
         else stereo_split(X, Y, N);
      }
   } else if (stereo) {
      if (encode) {
         inv = itheta > 8192 && !ctx->disable_inv;
         if (inv) for (int j = 0; j < N; j++) Y[j] = neg32(Y[j]);
         intensity_stereo(X, Y, bandE, i, N);
      }
      if (*b > 2 << BITRES && ctx->remaining_bits > 2 << BITRES) {
         if (encode) oc_ec_enc_bit_logp(ec, inv, 2);
         else inv = oc_ec_dec_bit_logp(ec, 2);
      } else inv = 0;
      if (ctx->disable_inv) inv = 0;
      itheta = 0;
   }
   qalloc = oc_ec_tell_frac(ec) - tell;
   *b -= qalloc;
   if (itheta == 0) { imid = 32767; iside = 0; *fill &= (1 << B) - 1; delta = -16384; }
   else if (itheta == 16384) { imid = 0; iside = 32767; *fill &= ((1 << B) - 1) << B; delta = 16384; }
   else {
      imid = oc_bitexact_cos((i16)itheta);
      iside = oc_bitexact_cos((i16)(16384 - itheta));
      delta = frac_mul16((N - 1) << 7, oc_bitexact_log2tan(iside, imid));
   }
   if (oc_dump_hook) { i32 a_ = itheta, b_ = qn; oc_dump_hook("itheta", &a_, 4); oc_dump_hook("qn", &b_, 4); }
   sctx->inv = inv; sctx->imid = imid; sctx->iside = iside; sctx->delta = delta; sctx->itheta = itheta; sctx->qalloc = qalloc;
}

static unsigned quant_band_n1(band_ctx *ctx, i32 *X, i32 *Y, i32 *lowband_out)
{
   i32 *x = X;
   int stereo = Y != 0;
   for (int c = 0; c < 1 + stereo; c++) {
      int sign = 0;
      if (ctx->remaining_bits >= 1 << BITRES) {
         if (ctx->encode) { sign = x[0] < 0; oc_ec_enc_bits(ctx->ec, sign, 1); }
         else sign = oc_ec_dec_bits(ctx->ec, 1);
         ctx->remaining_bits -= 1 << BITRES;
      }
      if (ctx->resynth) x[0] = sign ? -(1 << NORM_SHIFT) : (1 << NORM_SHIFT);
      x = Y;
   }
   if (lowband_out) lowband_out[0] = X[0] >> 4;
   return 1;
}

static unsigned quant_partition(band_ctx *ctx, i32 *X, int N, int b, int B, i32 *lowband, int LM, i32 gain, int fill)
{
   int imid = 0, iside = 0, B0 = B, i = ctx->i, spread = ctx->spread;
   i32 mid = 0, side = 0;
   unsigned cm = 0;
   const u8 *cache = oc_cache_bits + oc_cache_index[(LM + 1) * NB_EBANDS + i];
   if (LM != -1 && b > cache[cache[0]] + 12 && N > 2) {
      int mbits, sbits, delta, itheta, qalloc;
      split_ctx sctx;
      i32 *next_lowband2 = 0, *Y;
      i32 rebalance;
      N >>= 1;
      Y = X + N;
      LM -= 1;
      if (B == 1) fill = (fill & 1) | (fill << 1);
      B = (B + 1) >> 1;
      compute_theta(ctx, &sctx, X, Y, N, &b, B, B0, LM, 0, &fill);
      imid = sctx.imid; iside = sctx.iside; delta = sctx.delta; itheta = sctx.itheta; qalloc = sctx.qalloc;
      mid = shl32((i32)imid, 16);
      side = shl32((i32)iside, 16);
      if (B0 > 1 && (itheta & 0x3fff)) {
         if (itheta > 8192) delta -= delta >> (4 - LM);
         else delta = imin(0, delta + (N << BITRES >> (5 - LM)));
      }
      mbits = imax(0, imin(b, (b - delta) / 2));
      sbits = b - mbits;
      ctx->remaining_bits -= qalloc;
      if (lowband) next_lowband2 = lowband + N;
      rebalance = ctx->remaining_bits;
      if (mbits >= sbits) {
         cm = quant_partition(ctx, X, N, mbits, B, lowband, LM, mult32_32_q31(gain, mid), fill);
         rebalance = mbits - (rebalance - ctx->remaining_bits);
         if (rebalance > 3 << BITRES && itheta != 0) sbits += rebalance - (3 << BITRES);
         cm |= quant_partition(ctx, Y, N, sbits, B, next_lowband2, LM, mult32_32_q31(gain, side), fill >> B) << (B0 >> 1);
      } else {
         cm = quant_partition(ctx, Y, N, sbits, B, next_lowband2, LM, mult32_32_q31(gain, side), fill >> B) << (B0 >> 1);
         rebalance = sbits - (rebalance - ctx->remaining_bits);
         if (rebalance > 3 << BITRES && itheta != 16384) mbits += rebalance - (3 << BITRES);
         cm |= quant_partition(ctx, X, N, mbits, B, lowband, LM, mult32_32_q31(gain, mid), fill);
      }
   } else {
      int q = oc_bits2pulses(i, LM, b);
      int curr_bits = oc_pulses2bits(i, LM, q);
      ctx->remaining_bits -= curr_bits;
      while (ctx->remaining_bits < 0 && q > 0) {
         ctx->remaining_bits += curr_bits;
         q--;
         curr_bits = oc_pulses2bits(i, LM, q);
         ctx->remaining_bits -= curr_bits;
      }
      if (q != 0) {
         int K = oc_get_pulses(q);
         if (ctx->encode) cm = oc_alg_quant(X, N, K, spread, B, ctx->ec, gain, ctx->resynth);
         else cm = oc_alg_unquant(X, N, K, spread, B, ctx->ec, gain);
      } else if (ctx->resynth) {
         unsigned cm_mask = (unsigned)(1UL << B) - 1;
         fill &= cm_mask;
         if (!fill) memset(X, 0, N * sizeof(i32));
         else {
            if (lowband == 0) {
               for (int j = 0; j < N; j++) {
                  ctx->seed = lcg_rand(ctx->seed);
                  X[j] = shl32((i32)((i32)ctx->seed >> 20), NORM_SHIFT - 14);
               }
               cm = cm_mask;
            } else {
               for (int j = 0; j < N; j++) {
                  ctx->seed = lcg_rand(ctx->seed);
                  i16 tmp = QC16(1.0f / 256, NORM_SHIFT - 4);
                  tmp = (ctx->seed) & 0x8000 ? tmp : -tmp;
                  X[j] = lowband[j] + tmp;
               }
               cm = fill;
            }
            oc_renormalise_vector(X, N, gain);
         }
      }
   }
   return cm;
}

static unsigned quant_band(band_ctx *ctx, i32 *X, int N, int b, int B, i32 *lowband, int LM, i32 *lowband_out,
      i32 gain, i32 *lowband_scratch, int fill)
{
   static const u8 bit_interleave_table[16] = {0, 1, 1, 1, 2, 3, 3, 3, 2, 3, 3, 3, 2, 3, 3, 3};
   static const u8 bit_deinterleave_table[16] = {0x00, 0x03, 0x0C, 0x0F, 0x30, 0x33, 0x3C, 0x3F, 0xC0, 0xC3, 0xCC, 0xCF, 0xF0, 0xF3, 0xFC, 0xFF};
   int N0 = N, N_B = N, N_B0, B0 = B, time_divide = 0, recombine = 0, longBlocks, k;
   unsigned cm = 0;
   int encode = ctx->encode, tf_change = ctx->tf_change;
   longBlocks = B0 == 1;
   N_B = (u32)N_B / (u32)B;
   if (N == 1) return quant_band_n1(ctx, X, 0, lowband_out);
   if (tf_change > 0) recombine = tf_change;
   if (lowband_scratch && lowband && (recombine || ((N_B & 1) == 0 && tf_change < 0) || B0 > 1)) {
      memcpy(lowband_scratch, lowband, N * sizeof(i32));
      lowband = lowband_scratch;
   }
   for (k = 0; k < recombine; k++) {
      if (encode) oc_haar1(X, N >> k, 1 << k);
      if (lowband) oc_haar1(lowband, N >> k, 1 << k);
      fill = bit_interleave_table[fill & 0xF] | bit_interleave_table[fill >> 4] << 2;
   }
   B >>= recombine;
   N_B <<= recombine;
   while ((N_B & 1) == 0 && tf_change < 0) {
      if (encode) oc_haar1(X, N_B, B);
      if (lowband) oc_haar1(lowband, N_B, B);
      fill |= fill << B;
      B <<= 1;
      N_B >>= 1;
      time_divide++;
      tf_change++;
   }
   B0 = B;
   N_B0 = N_B;
   if (B0 > 1) {
      if (encode) deinterleave_hadamard(X, N_B >> recombine, B0 << recombine, longBlocks);
      if (lowband) deinterleave_hadamard(lowband, N_B >> recombine, B0 << recombine, longBlocks);
   }
   cm = quant_partition(ctx, X, N, b, B, lowband, LM, gain, fill);
   if (ctx->resynth) {
      if (B0 > 1) interleave_hadamard(X, N_B >> recombine, B0 << recombine, longBlocks);
      N_B = N_B0;
      B = B0;
      for (k = 0; k < time_divide; k++) {
         B >>= 1;
         N_B <<= 1;
         cm |= cm >> B;
         oc_haar1(X, N_B, B);
      }
      for (k = 0; k < recombine; k++) {
         cm = bit_deinterleave_table[cm];
         oc_haar1(X, N0 >> k, 1 << k);
      }
      B <<= recombine;
      if (lowband_out) {
         i16 n = (i16)oc_sqrt(shl32((i32)N0, 22));
         for (int j = 0; j < N0; j++) lowband_out[j] = mult16_32_q15(n, X[j]);
      }
      cm &= (1 << B) - 1;
   }
   return cm;
}

static unsigned quant_band_stereo(band_ctx *ctx, i32 *X, i32 *Y, int N, int b, int B, i32 *lowband, int LM,
      i32 *lowband_out, i32 *lowband_scratch, int fill)
{
   int imid = 0, iside = 0, inv = 0, mbits, sbits, delta, itheta, qalloc, orig_fill, encode = ctx->encode;
   i32 mid = 0, side = 0;
   unsigned cm = 0;
   split_ctx sctx;
   oc_ec *ec = ctx->ec;
   if (N == 1) return quant_band_n1(ctx, X, Y, lowband_out);
   orig_fill = fill;
   if (encode) {
      if (ctx->bandE[ctx->i] < 2 || ctx->bandE[NB_EBANDS + ctx->i] < 2) {
         if (ctx->bandE[ctx->i] > ctx->bandE[NB_EBANDS + ctx->i]) memcpy(Y, X, N * sizeof(i32));
         else memcpy(X, Y, N * sizeof(i32));
      }
   }
   compute_theta(ctx, &sctx, X, Y, N, &b, B, B, LM, 1, &fill);
   inv = sctx.inv; imid = sctx.imid; iside = sctx.iside; delta = sctx.delta; itheta = sctx.itheta; qalloc = sctx.qalloc;
   mid = shl32((i32)imid, 16);
   side = shl32((i32)iside, 16);
   if (N == 2) {
      int c, sign = 0;
      i32 *x2, *y2;
      mbits = b;
      sbits = 0;
      if (itheta != 0 && itheta != 16384) sbits = 1 << BITRES;
      mbits -= sbits;
      c = itheta > 8192;
      ctx->remaining_bits -= qalloc + sbits;
      x2 = c ? Y : X;
      y2 = c ? X : Y;
      if (sbits) {
         if (encode) {
            sign = mult32_32_q31(x2[0], y2[1]) - mult32_32_q31(x2[1], y2[0]) < 0;
            oc_ec_enc_bits(ec, sign, 1);
         } else sign = oc_ec_dec_bits(ec, 1);
      }
      sign = 1 - 2 * sign;
      cm = quant_band(ctx, x2, N, mbits, B, lowband, LM, lowband_out, Q31ONE, lowband_scratch, orig_fill);
      y2[0] = -sign * x2[1];
      y2[1] = sign * x2[0];
      if (ctx->resynth) {
         i32 tmp;
         X[0] = mult32_32_q31(mid, X[0]);
         X[1] = mult32_32_q31(mid, X[1]);
         Y[0] = mult32_32_q31(side, Y[0]);
         Y[1] = mult32_32_q31(side, Y[1]);
         tmp = X[0]; X[0] = sub32(tmp, Y[0]); Y[0] = add32(tmp, Y[0]);
         tmp = X[1]; X[1] = sub32(tmp, Y[1]); Y[1] = add32(tmp, Y[1]);
      }
   } else {
      i32 rebalance;
      mbits = imax(0, imin(b, (b - delta) / 2));
      sbits = b - mbits;
      ctx->remaining_bits -= qalloc;
      rebalance = ctx->remaining_bits;
      if (mbits >= sbits) {
         cm = quant_band(ctx, X, N, mbits, B, lowband, LM, lowband_out, Q31ONE, lowband_scratch, fill);
         rebalance = mbits - (rebalance - ctx->remaining_bits);
         if (rebalance > 3 << BITRES && itheta != 0) sbits += rebalance - (3 << BITRES);
         cm |= quant_band(ctx, Y, N, sbits, B, 0, LM, 0, side, 0, fill >> B);
      } else {
         cm = quant_band(ctx, Y, N, sbits, B, 0, LM, 0, side, 0, fill >> B);
         rebalance = sbits - (rebalance - ctx->remaining_bits);
         if (rebalance > 3 << BITRES && itheta != 16384) mbits += rebalance - (3 << BITRES);
         cm |= quant_band(ctx, X, N, mbits, B, lowband, LM, lowband_out, Q31ONE, lowband_scratch, fill);
      }
   }
   if (ctx->resynth) {
      if (N != 2) stereo_merge(X, Y, mid, N);
      if (inv) for (int j = 0; j < N; j++) Y[j] = neg32(Y[j]);
   }
   return cm;
}

static void special_hybrid_folding(i32 *norm, i32 *norm2, int start, int M, int dual_stereo)
{
   const int16_t *eB = oc_eBands;
   int n1 = M * (eB[start + 1] - eB[start]), n2 = M * (eB[start + 2] - eB[start + 1]);
   if (n2 - n1 > 0) {
      memcpy(&norm[n1], &norm[2 * n1 - n2], (n2 - n1) * sizeof(i32));
      if (dual_stereo) memcpy(&norm2[n1], &norm2[2 * n1 - n2], (n2 - n1) * sizeof(i32));
   }
}

void oc_quant_all_bands(int encode, int start, int end, i32 *X_, i32 *Y_, u8 *collapse_masks,
      const i32 *bandE, int *pulses, int shortBlocks, int spread, int dual_stereo, int intensity,
      int *tf_res, i32 total_bits, i32 balance, oc_ec *ec, int LM, int codedBands, u32 *seed,
      int complexity, int disable_inv)
{
   const int16_t *eB = oc_eBands;
   i32 remaining_bits;
   static i32 _norm_storage[2 * 8 * 100];   /* C*(M*eBands[20]-norm_offset) <= 2*800 */
   i32 _norm[2 * 800];
   i32 _lowband_scratch[176], X_save[176], Y_save[176], X_save2[176], Y_save2[176], norm_save2[176];
   u8 bytes_save[1275];
   i32 *norm, *norm2, *lowband_scratch;
   int M = 1 << LM, B = shortBlocks ? M : 1, lowband_offset = 0, update_lowband = 1, C = Y_ != 0 ? 2 : 1;
   int norm_offset = M * eB[start];
   int theta_rdo = encode && Y_ != 0 && !dual_stereo && complexity >= 8;
   int resynth = !encode || theta_rdo;
   band_ctx ctx;
   (void)_norm_storage;
   norm = _norm;
   norm2 = norm + M * eB[NB_EBANDS - 1] - norm_offset;
   if (encode && resynth) lowband_scratch = _lowband_scratch;
   else lowband_scratch = X_ + M * eB[NB_EBANDS - 1];
   ctx.bandE = bandE; ctx.ec = ec; ctx.encode = encode; ctx.intensity = intensity; ctx.seed = *seed;
   ctx.spread = spread; ctx.disable_inv = disable_inv; ctx.resynth = resynth; ctx.theta_round = 0;
   ctx.avoid_split_noise = B > 1;
   for (int i = start; i < end; i++) {
      i32 tell, curr_balance;
      int b, N, effective_lowband = -1, tf_change = 0, last;
      i32 *X, *Y;
      unsigned x_cm, y_cm;
      ctx.i = i;
      last = (i == end - 1);
      X = X_ + M * eB[i];
      Y = Y_ != 0 ? Y_ + M * eB[i] : 0;
      N = M * eB[i + 1] - M * eB[i];
      tell = oc_ec_tell_frac(ec);
      if (i != start) balance -= tell;
      remaining_bits = total_bits - tell - 1;
      ctx.remaining_bits = remaining_bits;
      if (i <= codedBands - 1) {
         curr_balance = balance / imin(3, codedBands - i);
         b = imax(0, imin(16383, imin(remaining_bits + 1, pulses[i] + curr_balance)));
      } else b = 0;
      if (resynth && (M * eB[i] - N >= M * eB[start] || i == start + 1) && (update_lowband || lowband_offset == 0))
         lowband_offset = i;
      if (i == start + 1) special_hybrid_folding(norm, norm2, start, M, dual_stereo);
      tf_change = tf_res[i];
      ctx.tf_change = tf_change;
      if (last && !theta_rdo) lowband_scratch = 0;
      if (lowband_offset != 0 && (spread != SPREAD_AGGRESSIVE || B > 1 || tf_change < 0)) {
         int fold_start, fold_end, fold_i;
         effective_lowband = imax(0, M * eB[lowband_offset] - norm_offset - N);
         fold_start = lowband_offset;
         while (M * eB[--fold_start] > effective_lowband + norm_offset);
         fold_end = lowband_offset - 1;
         while (++fold_end < i && M * eB[fold_end] < effective_lowband + norm_offset + N);
         x_cm = y_cm = 0;
         fold_i = fold_start;
         do {
            x_cm |= collapse_masks[fold_i * C + 0];
            y_cm |= collapse_masks[fold_i * C + C - 1];
         } while (++fold_i < fold_end);
      } else x_cm = y_cm = (1 << B) - 1;
      if (dual_stereo && i == intensity) {
         dual_stereo = 0;
         if (resynth) for (int j = 0; j < M * eB[i] - norm_offset; j++) norm[j] = half32(norm[j] + norm2[j]);
      }
      i32 *lb = effective_lowband != -1 ? norm + effective_lowband : 0;
      i32 *lb2 = effective_lowband != -1 ? norm2 + effective_lowband : 0;
      i32 *lbo = last ? 0 : norm + M * eB[i] - norm_offset;
      i32 *lbo2 = last ? 0 : norm2 + M * eB[i] - norm_offset;
      if (dual_stereo) {
         x_cm = quant_band(&ctx, X, N, b / 2, B, lb, LM, lbo, Q31ONE, lowband_scratch, x_cm);
         y_cm = quant_band(&ctx, Y, N, b / 2, B, lb2, LM, lbo2, Q31ONE, lowband_scratch, y_cm);
      } else {
         if (Y != 0) {
            if (theta_rdo && i < intensity) {
               oc_ec ec_save, ec_save2;
               band_ctx ctx_save, ctx_save2;
               i32 dist0, dist1;
               unsigned cm, cm2;
               i16 w[2];
               compute_channel_weights(bandE[i], bandE[i + NB_EBANDS], w);
               cm = x_cm | y_cm;
               ec_save = *ec;
               ctx_save = ctx;
               memcpy(X_save, X, N * sizeof(i32));
               memcpy(Y_save, Y, N * sizeof(i32));
               ctx.theta_round = -1;
               x_cm = quant_band_stereo(&ctx, X, Y, N, b, B, lb, LM, lbo, lowband_scratch, cm);
               dist0 = mult16_32_q15(w[0], oc_inner_prod_norm_shift(X_save, X, N)) + mult16_32_q15(w[1], oc_inner_prod_norm_shift(Y_save, Y, N));
               cm2 = x_cm;
               ec_save2 = *ec;
               ctx_save2 = ctx;
               memcpy(X_save2, X, N * sizeof(i32));
               memcpy(Y_save2, Y, N * sizeof(i32));
               if (!last) memcpy(norm_save2, norm + M * eB[i] - norm_offset, N * sizeof(i32));
               int nstart_bytes = ec_save.offs, nend_bytes = ec_save.storage;
               u8 *bytes_buf = ec_save.buf + nstart_bytes;
               int save_bytes = nend_bytes - nstart_bytes;
               memcpy(bytes_save, bytes_buf, save_bytes);
               *ec = ec_save;
               ctx = ctx_save;
               memcpy(X, X_save, N * sizeof(i32));
               memcpy(Y, Y_save, N * sizeof(i32));
               if (i == start + 1) special_hybrid_folding(norm, norm2, start, M, dual_stereo);
               ctx.theta_round = 1;
               x_cm = quant_band_stereo(&ctx, X, Y, N, b, B, lb, LM, lbo, lowband_scratch, cm);
               dist1 = mult16_32_q15(w[0], oc_inner_prod_norm_shift(X_save, X, N)) + mult16_32_q15(w[1], oc_inner_prod_norm_shift(Y_save, Y, N));
               if (dist0 >= dist1) {
                  x_cm = cm2;
                  *ec = ec_save2;
                  ctx = ctx_save2;
                  memcpy(X, X_save2, N * sizeof(i32));
                  memcpy(Y, Y_save2, N * sizeof(i32));
                  if (!last) memcpy(norm + M * eB[i] - norm_offset, norm_save2, N * sizeof(i32));
                  memcpy(bytes_buf, bytes_save, save_bytes);
               }
            } else {
               ctx.theta_round = 0;
               x_cm = quant_band_stereo(&ctx, X, Y, N, b, B, lb, LM, lbo, lowband_scratch, x_cm | y_cm);
            }
         } else {
            x_cm = quant_band(&ctx, X, N, b, B, lb, LM, lbo, Q31ONE, lowband_scratch, x_cm | y_cm);
         }
         y_cm = x_cm;
      }
      collapse_masks[i * C + 0] = (u8)x_cm;
      collapse_masks[i * C + C - 1] = (u8)y_cm;
      balance += pulses[i] + tell;
      update_lowband = b > (N << BITRES);
      ctx.avoid_split_noise = 0;
   }
   *seed = ctx.seed;
}

/* oc_vq.c — PVQ search, spreading rotation, residual (re)normalisation, stereo angle.
 * Oracle restatement of celt/vq.c:44-70 (norm helpers), :75-147 (exp_rotation), :150-203
 * (normalise_residual, extract_collapse_mask), :205-386 (op_pvq_search_c), :552-618 (alg_quant),
 * :695-756 (renormalise_vector, stereo_itheta).  celt_norm is int32 Q24 (NORM_SHIFT 24). */
#include "oc_celt.h"
#include <stdlib.h>
extern void (*oc_dump_hook)(const char *tag, const void *p, int nbytes);

static void norm_scaleup(i32 *X, int N, int shift) { if (shift <= 0) return; for (int i = 0; i < N; i++) X[i] = shl32(X[i], shift); }
static void norm_scaledown(i32 *X, int N, int shift) { if (shift <= 0) return; for (int i = 0; i < N; i++) X[i] = pshr32(X[i], shift); }
static i32 inner_prod_norm(const i32 *x, const i32 *y, int len)
{
   i32 sum = 0;
   for (int i = 0; i < len; i++) sum = add32(sum, (i32)((u32)x[i] * (u32)y[i]));
   return sum;
}
i32 oc_inner_prod_norm_shift(const i32 *x, const i32 *y, int len)
{
   i64 sum = 0;
   for (int i = 0; i < len; i++) sum += x[i] * (i64)y[i];
   return (i32)(sum >> 2 * (NORM_SHIFT - 14));
}

/* exp_rotation1, vq.c:75 */
static void exp_rotation1(i32 *X, int len, int stride, i16 c, i16 s)
{
   i16 ms = (i16)(-s);
   i32 *Xptr = X;
   norm_scaledown(X, len, NORM_SHIFT - 14);
   for (int i = 0; i < len - stride; i++) {
      i32 x1 = Xptr[0], x2 = Xptr[stride];
      Xptr[stride] = extract16(pshr32(mac16_16(mult16_16(c, x2), s, x1), 15));
      *Xptr++ = extract16(pshr32(mac16_16(mult16_16(c, x1), ms, x2), 15));
   }
   Xptr = &X[len - 2 * stride - 1];
   for (int i = len - 2 * stride - 1; i >= 0; i--) {
      i32 x1 = Xptr[0], x2 = Xptr[stride];
      Xptr[stride] = extract16(pshr32(mac16_16(mult16_16(c, x2), s, x1), 15));
      *Xptr-- = extract16(pshr32(mac16_16(mult16_16(c, x1), ms, x2), 15));
   }
   norm_scaleup(X, len, NORM_SHIFT - 14);
}
/* exp_rotation, vq.c:104 */
void oc_exp_rotation(i32 *X, int len, int dir, int stride, int K, int spread)
{
   static const int SPREAD_FACTOR[3] = {15, 10, 5};
   int stride2 = 0;
   if (2 * K >= len || spread == SPREAD_NONE) return;
   int factor = SPREAD_FACTOR[spread - 1];
   i16 gain = (i16)oc_div(mult16_16(Q15ONE, len), (i32)(len + factor * K));
   i16 theta = (i16)(mult16_16_q15(gain, gain) >> 1);
   i16 c = oc_cos_norm(theta);
   i16 s = oc_cos_norm(sub16(Q15ONE, theta));
   if (len >= 8 * stride) {
      stride2 = 1;
      while ((stride2 * stride2 + stride2) * stride + (stride >> 2) < len) stride2++;
   }
   len = (u32)len / (u32)stride;
   for (int i = 0; i < stride; i++) {
      if (dir < 0) {
         if (stride2) exp_rotation1(X + i * len, len, stride2, s, c);
         exp_rotation1(X + i * len, len, 1, c, s);
      } else {
         exp_rotation1(X + i * len, len, 1, c, (i16)-s);
         if (stride2) exp_rotation1(X + i * len, len, stride2, s, (i16)-c);
      }
   }
}

/* normalise_residual, vq.c:150 */
static void normalise_residual(const int *iy, i32 *X, int N, i32 Ryy, i32 gain)
{
   int k = celt_ilog2(Ryy) >> 1;
   i32 t = vshr32(Ryy, 2 * (k - 7) - 15);
   i32 g = mult32_32_q31(oc_rsqrt_norm32(t), gain);
   for (int i = 0; i < N; i++) X[i] = vshr32(mult16_32_q15(iy[i], g), k + 15 - NORM_SHIFT);
}
/* extract_collapse_mask, vq.c:183 */
static unsigned extract_collapse_mask(const int *iy, int N, int B)
{
   if (B <= 1) return 1;
   int N0 = (u32)N / (u32)B;
   unsigned mask = 0;
   for (int i = 0; i < B; i++) {
      unsigned tmp = 0;
      for (int j = 0; j < N0; j++) tmp |= iy[i * N0 + j];
      mask |= (tmp != 0) << i;
   }
   return mask;
}

/* op_pvq_search_c, vq.c:205.  Returns yy (as the reference's opus_val16). X is destroyed (|X| scaled). */
i32 oc_op_pvq_search(i32 *X, int *iy, int K, int N)
{
   i32 y[176];
   int signx[176];
   i32 sum = 0, xy = 0;
   i16 yy = 0;
   {
      int shift = (celt_ilog2(1 + oc_inner_prod_norm_shift(X, X, N)) + 1) / 2;
      shift = imax(0, shift + (NORM_SHIFT - 14) - 14);
      norm_scaledown(X, N, shift);
   }
   for (int j = 0; j < N; j++) { signx[j] = X[j] < 0; X[j] = iabs(X[j]); iy[j] = 0; y[j] = 0; }
   int pulsesLeft = K;
   if (K > (N >> 1)) {
      for (int j = 0; j < N; j++) sum += X[j];
      if (sum <= K) {
         X[0] = QC16(1.f, 14);
         for (int j = 1; j < N; j++) X[j] = 0;
         sum = QC16(1.f, 14);
      }
      i16 rcp = extract16(mult16_32_q16(K, oc_rcp(sum)));
      for (int j = 0; j < N; j++) {
         iy[j] = mult16_16_q15(X[j], rcp);
         y[j] = (i32)iy[j];
         yy = (i16)mac16_16(yy, y[j], y[j]);
         xy = mac16_16(xy, X[j], y[j]);
         y[j] *= 2;
         pulsesLeft -= iy[j];
      }
   }
   if (pulsesLeft > N + 3) {
      i16 tmp = (i16)pulsesLeft;
      yy = (i16)mac16_16(yy, tmp, tmp);
      yy = (i16)mac16_16(yy, tmp, y[0]);
      iy[0] += pulsesLeft;
      pulsesLeft = 0;
   }
   for (int i = 0; i < pulsesLeft; i++) {
      int rshift = 1 + celt_ilog2(K - pulsesLeft + i + 1);
      int best_id = 0;
      yy = add16(yy, 1);
      i16 Rxy = extract16(add32(xy, X[0]) >> rshift);
      i16 Ryy = add16(yy, y[0]);
      Rxy = (i16)mult16_16_q15(Rxy, Rxy);
      i16 best_den = Ryy;
      i32 best_num = Rxy;
      for (int j = 1; j < N; j++) {
         Rxy = extract16(add32(xy, X[j]) >> rshift);
         Ryy = add16(yy, y[j]);
         Rxy = (i16)mult16_16_q15(Rxy, Rxy);
         if (mult16_16(best_den, Rxy) > mult16_16(Ryy, best_num)) { best_den = Ryy; best_num = Rxy; best_id = j; }
      }
      xy = add32(xy, X[best_id]);
      yy = add16(yy, y[best_id]);
      y[best_id] += 2;
      iy[best_id]++;
   }
   for (int j = 0; j < N; j++) iy[j] = (iy[j] ^ -signx[j]) + signx[j];
   return yy;
}

/* alg_quant, vq.c:552 (non-QEXT path) */
unsigned oc_alg_quant(i32 *X, int N, int K, int spread, int B, oc_ec *enc, i32 gain, int resynth)
{
   int iy[176 + 3];
   if (oc_dump_hook) oc_dump_hook("pvqX", X, N * 4);
   oc_exp_rotation(X, N, 1, B, K, spread);
   i32 yy = oc_op_pvq_search(X, iy, K, N);
   unsigned cm = extract_collapse_mask(iy, N, B);
   if (oc_dump_hook) { i32 k_ = K; oc_dump_hook("iy", iy, N * 4); oc_dump_hook("pvqK", &k_, 4); }
   oc_encode_pulses(iy, N, K, enc);
   if (resynth) {
      normalise_residual(iy, X, N, yy, gain);
      oc_exp_rotation(X, N, -1, B, K, spread);
   }
   return cm;
}

/* alg_unquant, vq.c:621 (non-QEXT path) */
unsigned oc_alg_unquant(i32 *X, int N, int K, int spread, int B, oc_ec *dec, i32 gain)
{
   int iy[176 + 3];
   i32 Ryy = oc_decode_pulses(iy, N, K, dec);
   normalise_residual(iy, X, N, Ryy, gain);
   oc_exp_rotation(X, N, -1, B, K, spread);
   return extract_collapse_mask(iy, N, B);
}

/* renormalise_vector, vq.c:695 */
void oc_renormalise_vector(i32 *X, int N, i32 gain)
{
   norm_scaledown(X, N, NORM_SHIFT - 14);
   i32 E = EPSILON + inner_prod_norm(X, X, N);
   int k = celt_ilog2(E) >> 1;
   i32 t = vshr32(E, 2 * (k - 7));
   i16 g = (i16)mult32_32_q31(oc_rsqrt_norm(t), gain);
   for (int i = 0; i < N; i++) X[i] = extract16(pshr32(mult16_16(g, X[i]), k + 15 - 14));
   norm_scaleup(X, N, NORM_SHIFT - 14);
}

/* stereo_itheta, vq.c:724 (returns Q30 angle * 2/pi) */
i32 oc_stereo_itheta(const i32 *X, const i32 *Y, int stereo, int N)
{
   i32 Emid = 0, Eside = 0;
   if (stereo) {
      for (int i = 0; i < N; i++) {
         i32 m = pshr32(add32(X[i], Y[i]), NORM_SHIFT - 13);
         i32 s = pshr32(sub32(X[i], Y[i]), NORM_SHIFT - 13);
         Emid = mac16_16(Emid, m, m);
         Eside = mac16_16(Eside, s, s);
      }
   } else {
      Emid += oc_inner_prod_norm_shift(X, X, N);
      Eside += oc_inner_prod_norm_shift(Y, Y, N);
   }
   i32 mid = oc_sqrt32(Emid), side = oc_sqrt32(Eside);
   return oc_atan2p_norm(side, mid);
}

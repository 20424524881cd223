/* oc_pitch.c — pitch analysis for the CELT pre-filter and the 5-tap comb filter.
 * Oracle restatement of celt/pitch.c:45-570, celt/pitch.h:65-165 (xcorr kernel, inner products),
 * celt/celt_lpc.c:37-140 (_celt_lpc), :284-374 (_celt_autocorr), celt/celt.c:166-312 (comb_filter). */
#include "oc_celt.h"
#include <stdlib.h>

static i32 inner_prod(const i16 *x, const i16 *y, int N)
{
   i32 xy = 0;
   for (int i = 0; i < N; i++) xy = mac16_16(xy, x[i], y[i]);
   return xy;
}
static void dual_inner_prod(const i16 *x, const i16 *y1, const i16 *y2, int N, i32 *xy1, i32 *xy2)
{
   i32 a = 0, b = 0;
   for (int i = 0; i < N; i++) { a = mac16_16(a, x[i], y1[i]); b = mac16_16(b, x[i], y2[i]); }
   *xy1 = a; *xy2 = b;
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

/* celt_pitch_xcorr_c, pitch.c:230: xcorr[i] = sum_j x[j]*y[i+j] (mod 2^32), returns max(1,max xcorr) */
i32 oc_pitch_xcorr(const i16 *x, const i16 *y, i32 *xcorr, int len, int max_pitch)
{
   i32 maxcorr = 1;
   for (int i = 0; i < max_pitch; i++) {
      i32 s = inner_prod(x, y + i, len);
      xcorr[i] = s;
      maxcorr = imax(maxcorr, s);
   }
   return maxcorr;
}

/* find_best_pitch, pitch.c:45 */
static void find_best_pitch(const i32 *xcorr, const i16 *y, int len, int max_pitch, int *best_pitch, int yshift, i32 maxcorr)
{
   i32 Syy = 1;
   i16 best_num[2] = {-1, -1};
   i32 best_den[2] = {0, 0};
   int xshift = celt_ilog2(maxcorr) - 14;
   best_pitch[0] = 0; best_pitch[1] = 1;
   for (int j = 0; j < len; j++) Syy = add32(Syy, mult16_16(y[j], y[j]) >> yshift);
   for (int i = 0; i < max_pitch; i++) {
      if (xcorr[i] > 0) {
         i16 xcorr16 = extract16(vshr32(xcorr[i], xshift));
         i16 num = (i16)mult16_16_q15(xcorr16, xcorr16);
         if (mult16_32_q15(num, best_den[1]) > mult16_32_q15(best_num[1], Syy)) {
            if (mult16_32_q15(num, best_den[0]) > mult16_32_q15(best_num[0], Syy)) {
               best_num[1] = best_num[0]; best_den[1] = best_den[0]; best_pitch[1] = best_pitch[0];
               best_num[0] = num; best_den[0] = Syy; best_pitch[0] = i;
            } else {
               best_num[1] = num; best_den[1] = Syy; best_pitch[1] = i;
            }
         }
      }
      Syy += (mult16_16(y[i + len], y[i + len]) >> yshift) - (mult16_16(y[i], y[i]) >> yshift);
      Syy = imax(1, Syy);
   }
}

/* celt_fir5, pitch.c:103 */
void oc_celt_fir5(i16 *x, const i16 *num, int N)
{
   i32 m0 = 0, m1 = 0, m2 = 0, m3 = 0, m4 = 0;
   for (int i = 0; i < N; i++) {
      i32 sum = shl32((i32)x[i], SIG_SHIFT);
      sum = mac16_16(sum, num[0], m0);
      sum = mac16_16(sum, num[1], m1);
      sum = mac16_16(sum, num[2], m2);
      sum = mac16_16(sum, num[3], m3);
      sum = mac16_16(sum, num[4], m4);
      m4 = m3; m3 = m2; m2 = m1; m1 = m0; m0 = x[i];
      x[i] = round16(sum, SIG_SHIFT);
   }
}

/* _celt_lpc, celt_lpc.c:37 (order p<=24) */
int oc_celt_lpc(i16 *_lpc, const i32 *ac, int p)
{
   i32 lpc[24], r, error = ac[0];
   memset(lpc, 0, sizeof(i32) * p);
   if (ac[0] != 0) {
      for (int i = 0; i < p; i++) {
         i64 acc = 0;
         for (int j = 0; j < i; j++) acc += (i64)lpc[j] * (i64)ac[i - j];
         i32 rr = (i32)(acc >> 31);
         rr += ac[i + 1] >> 6;
         r = neg32(oc_frac_div32(shl32(rr, 6), error));
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
         i32 chirp = QC32(0.999, 16) - shl32(maxabs - 32767, 14) / ((maxabs * (idx + 1)) >> 2);
         i32 chirp_m1 = chirp - 65536;
         for (int i = 0; i < p - 1; i++) {
            lpc[i] = mult32_32_q16(chirp, lpc[i]);
            chirp += pshr32(chirp * chirp_m1, 16);
         }
         lpc[p - 1] = mult32_32_q16(chirp, lpc[p - 1]);
      } else break;
   }
   if (iter == 10) { memset(_lpc, 0, sizeof(i16) * p); _lpc[0] = 4096; }
   else for (int i = 0; i < p; i++) _lpc[i] = extract16(pshr32(lpc[i], 13));
   return 0;
}

/* _celt_autocorr, celt_lpc.c:284 (n<=1024+? callers here: n<=1080) */
int oc_autocorr(const i16 *x, i32 *ac, const i16 *window, int overlap, int lag, int n)
{
   i16 xx[2048];
   const i16 *xptr;
   int fastN = n - lag, shift;
   if (overlap == 0) xptr = x;
   else {
      for (int i = 0; i < n; i++) xx[i] = x[i];
      for (int i = 0; i < overlap; i++) {
         i16 w = window[i];
         xx[i] = (i16)mult16_16_q15(x[i], w);
         xx[n - i - 1] = (i16)mult16_16_q15(x[n - i - 1], w);
      }
      xptr = xx;
   }
   {
      int ac0_shift = celt_ilog2(n + (n >> 4));
      i32 ac0 = 1 + (n << 7);
      if (n & 1) ac0 += mult16_16(xptr[0], xptr[0]) >> ac0_shift;
      for (int i = (n & 1); i < n; i += 2) {
         ac0 += mult16_16(xptr[i], xptr[i]) >> ac0_shift;
         ac0 += mult16_16(xptr[i + 1], xptr[i + 1]) >> ac0_shift;
      }
      ac0 += ac0 >> 7;
      shift = celt_ilog2(ac0) - 30 + ac0_shift + 1;
      shift = shift / 2;
      if (shift > 0) {
         for (int i = 0; i < n; i++) xx[i] = (i16)pshr32(xptr[i], shift);
         xptr = xx;
      } else shift = 0;
   }
   oc_pitch_xcorr(xptr, xptr, ac, fastN, lag + 1);
   for (int k = 0; k <= lag; k++) {
      i32 d = 0;
      for (int i = k + fastN; i < n; i++) d = mac16_16(d, xptr[i], xptr[i - k]);
      ac[k] += d;
   }
   shift = 2 * shift;
   if (shift <= 0) ac[0] += shl32(1, -shift);
   if (ac[0] < 268435456) {
      int s2 = 29 - ec_ilog(ac[0]);
      for (int i = 0; i <= lag; i++) ac[i] = shl32(ac[i], s2);
      shift -= s2;
   } else if (ac[0] >= 536870912) {
      int s2 = 1;
      if (ac[0] >= 1073741824) s2++;
      for (int i = 0; i <= lag; i++) ac[i] = ac[i] >> s2;
      shift += s2;
   }
   return shift;
}

/* pitch_downsample, pitch.c:140 */
void oc_pitch_downsample(i32 *x[], i16 *x_lp, int len, int C, int factor)
{
   i32 ac[5];
   i16 tmp = Q15ONE, lpc[4], lpc2[5], c1 = QC16(.8f, 15);
   int offset = factor / 2, shift;
   i32 maxabs = maxabs32(x[0], len * factor);
   if (C == 2) maxabs = imax(maxabs, maxabs32(x[1], len * factor));
   if (maxabs < 1) maxabs = 1;
   shift = celt_ilog2(maxabs) - 10;
   if (shift < 0) shift = 0;
   if (C == 2) shift++;
   for (int i = 1; i < len; i++)
      x_lp[i] = (i16)((x[0][factor * i - offset] >> (shift + 2)) + (x[0][factor * i + offset] >> (shift + 2)) + (x[0][factor * i] >> (shift + 1)));
   x_lp[0] = (i16)((x[0][offset] >> (shift + 2)) + (x[0][0] >> (shift + 1)));
   if (C == 2) {
      for (int i = 1; i < len; i++)
         x_lp[i] = (i16)(x_lp[i] + (x[1][factor * i - offset] >> (shift + 2)) + (x[1][factor * i + offset] >> (shift + 2)) + (x[1][factor * i] >> (shift + 1)));
      x_lp[0] = (i16)(x_lp[0] + (x[1][offset] >> (shift + 2)) + (x[1][0] >> (shift + 1)));
   }
   oc_autocorr(x_lp, ac, 0, 0, 4, len);
   ac[0] += ac[0] >> 13;
   for (int i = 1; i <= 4; i++) ac[i] -= mult16_32_q15(2 * i * i, ac[i]);
   oc_celt_lpc(lpc, ac, 4);
   for (int i = 0; i < 4; i++) {
      tmp = (i16)mult16_16_q15(QC16(.9f, 15), tmp);
      lpc[i] = (i16)mult16_16_q15(lpc[i], tmp);
   }
   lpc2[0] = (i16)(lpc[0] + QC16(.8f, SIG_SHIFT));
   lpc2[1] = (i16)(lpc[1] + mult16_16_q15(c1, lpc[0]));
   lpc2[2] = (i16)(lpc[2] + mult16_16_q15(c1, lpc[1]));
   lpc2[3] = (i16)(lpc[3] + mult16_16_q15(c1, lpc[2]));
   lpc2[4] = (i16)mult16_16_q15(c1, lpc[3]);
   oc_celt_fir5(x_lp, lpc2, len);
}

/* pitch_search, pitch.c:307 */
void oc_pitch_search(const i16 *x_lp, i16 *y, int len, int max_pitch, int *pitch)
{
   int lag = len + max_pitch, best_pitch[2] = {0, 0}, shift = 0, offset;
   i16 x_lp4[512], y_lp4[1024];
   i32 xcorr[1024], maxcorr;
   for (int j = 0; j < len >> 2; j++) x_lp4[j] = x_lp[2 * j];
   for (int j = 0; j < lag >> 2; j++) y_lp4[j] = y[2 * j];
   i32 xmax = maxabs16(x_lp4, len >> 2), ymax = maxabs16(y_lp4, lag >> 2);
   shift = celt_ilog2(imax(1, imax(xmax, ymax))) - 14 + celt_ilog2(len) / 2;
   if (shift > 0) {
      for (int j = 0; j < len >> 2; j++) x_lp4[j] = x_lp4[j] >> shift;
      for (int j = 0; j < lag >> 2; j++) y_lp4[j] = y_lp4[j] >> shift;
      shift *= 2;
   } else shift = 0;
   maxcorr = oc_pitch_xcorr(x_lp4, y_lp4, xcorr, len >> 2, max_pitch >> 2);
   find_best_pitch(xcorr, y_lp4, len >> 2, max_pitch >> 2, best_pitch, 0, maxcorr);
   maxcorr = 1;
   for (int i = 0; i < max_pitch >> 1; i++) {
      xcorr[i] = 0;
      if (abs(i - 2 * best_pitch[0]) > 2 && abs(i - 2 * best_pitch[1]) > 2) continue;
      i32 sum = 0;
      for (int j = 0; j < len >> 1; j++) sum += mult16_16(x_lp[j], y[i + j]) >> shift;
      xcorr[i] = imax(-1, sum);
      maxcorr = imax(maxcorr, sum);
   }
   find_best_pitch(xcorr, y, len >> 1, max_pitch >> 1, best_pitch, shift + 1, maxcorr);
   if (best_pitch[0] > 0 && best_pitch[0] < (max_pitch >> 1) - 1) {
      i32 a = xcorr[best_pitch[0] - 1], b = xcorr[best_pitch[0]], c = xcorr[best_pitch[0] + 1];
      if ((c - a) > mult16_32_q15(QC16(.7f, 15), b - a)) offset = 1;
      else if ((a - c) > mult16_32_q15(QC16(.7f, 15), b - c)) offset = -1;
      else offset = 0;
   } else offset = 0;
   *pitch = 2 * best_pitch[0] - offset;
}

/* compute_pitch_gain, pitch.c:418 */
static i16 pitch_gain(i32 xy, i32 xx, i32 yy)
{
   if (xy == 0 || xx == 0 || yy == 0) return 0;
   int sx = celt_ilog2(xx) - 14, sy = celt_ilog2(yy) - 14, shift = sx + sy;
   i32 x2y2 = mult16_16(vshr32(xx, sx), vshr32(yy, sy)) >> 14;
   if (shift & 1) {
      if (x2y2 < 32768) { x2y2 <<= 1; shift--; }
      else { x2y2 >>= 1; shift++; }
   }
   i16 den = oc_rsqrt_norm(x2y2);
   i32 g = mult16_32_q15(den, xy);
   g = vshr32(g, (shift >> 1) - 1);
   return extract16(imax(-Q15ONE, imin(g, Q15ONE)));
}

static const int second_check[16] = {0, 0, 3, 2, 3, 2, 5, 2, 3, 2, 3, 2, 5, 2, 3, 2};
/* remove_doubling, pitch.c:454 */
i16 oc_remove_doubling(i16 *x, int maxperiod, int minperiod, int N, int *T0_, int prev_period, i16 prev_gain)
{
   int k, T, T0, offset, minperiod0 = minperiod;
   i16 g, g0, pg;
   i32 xy, xx, yy, xy2, xcorr[3], best_xy, best_yy, yy_lookup[COMBFILTER_MAXPERIOD / 2 + 2];
   maxperiod /= 2; minperiod /= 2; *T0_ /= 2; prev_period /= 2; N /= 2;
   x += maxperiod;
   if (*T0_ >= maxperiod) *T0_ = maxperiod - 1;
   T = T0 = *T0_;
   dual_inner_prod(x, x, x - T0, N, &xx, &xy);
   yy_lookup[0] = xx;
   yy = xx;
   for (int i = 1; i <= maxperiod; i++) {
      yy = yy + mult16_16(x[-i], x[-i]) - mult16_16(x[N - i], x[N - i]);
      yy_lookup[i] = imax(0, yy);
   }
   yy = yy_lookup[T0];
   best_xy = xy; best_yy = yy;
   g = g0 = pitch_gain(xy, xx, yy);
   for (k = 2; k <= 15; k++) {
      int T1, T1b;
      i16 g1, cont, thresh;
      T1 = (u32)(2 * T0 + k) / (u32)(2 * k);
      if (T1 < minperiod) break;
      if (k == 2) T1b = (T1 + T0 > maxperiod) ? T0 : T0 + T1;
      else T1b = (u32)(2 * second_check[k] * T0 + k) / (u32)(2 * k);
      dual_inner_prod(x, &x[-T1], &x[-T1b], N, &xy, &xy2);
      xy = half32(xy + xy2);
      yy = half32(yy_lookup[T1] + yy_lookup[T1b]);
      g1 = pitch_gain(xy, xx, yy);
      if (abs(T1 - prev_period) <= 1) cont = prev_gain;
      else if (abs(T1 - prev_period) <= 2 && 5 * k * k < T0) cont = prev_gain >> 1;
      else cont = 0;
      thresh = (i16)imax(QC16(.3f, 15), mult16_16_q15(QC16(.7f, 15), g0) - cont);
      if (T1 < 3 * minperiod) thresh = (i16)imax(QC16(.4f, 15), mult16_16_q15(QC16(.85f, 15), g0) - cont);
      else if (T1 < 2 * minperiod) thresh = (i16)imax(QC16(.5f, 15), mult16_16_q15(QC16(.9f, 15), g0) - cont);
      if (g1 > thresh) { best_xy = xy; best_yy = yy; T = T1; g = g1; }
   }
   if (T < minperiod * 2) {
      int T1 = T * 5 / 8, T2 = T * 6 / 8;
      dual_inner_prod(x, &x[-T1], &x[-T2], N, &xy, &xy2);
      i16 g1 = pitch_gain(xy, xx, yy_lookup[T1]), g2 = pitch_gain(xy2, xx, yy_lookup[T2]);
      if (g1 >= g || g2 >= g) g = 0;
   }
   best_xy = imax(0, best_xy);
   if (best_yy <= best_xy) pg = Q15ONE;
   else pg = (i16)(oc_frac_div32(best_xy, best_yy + 1) >> 16);
   for (k = 0; k < 3; k++) xcorr[k] = inner_prod(x, x - (T + k - 1), N);
   if ((xcorr[2] - xcorr[0]) > mult16_32_q15(QC16(.7f, 15), xcorr[1] - xcorr[0])) offset = 1;
   else if ((xcorr[0] - xcorr[2]) > mult16_32_q15(QC16(.7f, 15), xcorr[1] - xcorr[2])) offset = -1;
   else offset = 0;
   if (pg > g) pg = g;
   *T0_ = 2 * T + offset;
   if (*T0_ < minperiod0) *T0_ = minperiod0;
   return pg;
}

/* comb_filter_const_c, celt.c:166 */
static void comb_const(i32 *y, i32 *x, int T, int N, i16 g10, i16 g11, i16 g12)
{
   i32 x4 = x[-T - 2], x3 = x[-T - 1], x2 = x[-T], x1 = x[-T + 1], x0;
   for (int i = 0; i < N; i++) {
      x0 = x[i - T + 2];
      i32 v = add32(add32(add32(x[i], mult_coef_32(g10, x2)), mult_coef_32(g11, add32(x1, x3))), mult_coef_32(g12, add32(x0, x4)));
      v = sub32(v, 1);
      y[i] = saturate(v, SIG_SAT);
      x4 = x3; x3 = x2; x2 = x1; x1 = x0;
   }
}
/* comb_filter, celt.c:238 */
void oc_comb_filter(i32 *y, i32 *x, int T0, int T1, int N, i16 g0, i16 g1, int tapset0, int tapset1, int overlap)
{
   static const i16 gains[3][3] = {
      {QC16(0.3066406250f, 15), QC16(0.2170410156f, 15), QC16(0.1296386719f, 15)},
      {QC16(0.4638671875f, 15), QC16(0.2680664062f, 15), QC16(0.f, 15)},
      {QC16(0.7998046875f, 15), QC16(0.1000976562f, 15), QC16(0.f, 15)}};
   int i;
   if (g0 == 0 && g1 == 0) { if (x != y) memmove(y, x, N * sizeof(i32)); return; }
   T0 = imax(T0, COMBFILTER_MINPERIOD);
   T1 = imax(T1, COMBFILTER_MINPERIOD);
   i16 g00 = (i16)mult_coef_taps(g0, gains[tapset0][0]), g01 = (i16)mult_coef_taps(g0, gains[tapset0][1]), g02 = (i16)mult_coef_taps(g0, gains[tapset0][2]);
   i16 g10 = (i16)mult_coef_taps(g1, gains[tapset1][0]), g11 = (i16)mult_coef_taps(g1, gains[tapset1][1]), g12 = (i16)mult_coef_taps(g1, gains[tapset1][2]);
   i32 x1 = x[-T1 + 1], x2 = x[-T1], x3 = x[-T1 - 1], x4 = x[-T1 - 2], x0;
   if (g0 == g1 && T0 == T1 && tapset0 == tapset1) overlap = 0;
   for (i = 0; i < overlap; i++) {
      x0 = x[i - T1 + 2];
      i16 f = (i16)mult_coef(oc_window[i], oc_window[i]);
      i32 v = x[i];
      v = add32(v, mult_coef_32(mult_coef((Q15ONE - f), g00), x[i - T0]));
      v = add32(v, mult_coef_32(mult_coef((Q15ONE - f), g01), add32(x[i - T0 + 1], x[i - T0 - 1])));
      v = add32(v, mult_coef_32(mult_coef((Q15ONE - f), g02), add32(x[i - T0 + 2], x[i - T0 - 2])));
      v = add32(v, mult_coef_32(mult_coef(f, g10), x2));
      v = add32(v, mult_coef_32(mult_coef(f, g11), add32(x1, x3)));
      v = add32(v, mult_coef_32(mult_coef(f, g12), add32(x0, x4)));
      v = sub32(v, 3);
      y[i] = saturate(v, SIG_SAT);
      x4 = x3; x3 = x2; x2 = x1; x1 = x0;
   }
   if (g1 == 0) { if (x != y) memmove(y + overlap, x + overlap, (N - overlap) * sizeof(i32)); return; }
   comb_const(y + i, x + i, T1, N - i, g10, g11, g12);
}

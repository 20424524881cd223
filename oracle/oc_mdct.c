/* oc_mdct.c — mixed-radix complex FFT (480/240/120/60) and the N=1920>>shift MDCT built on it.
 * Oracle restatement of celt/kiss_fft.c:52-312 (butterflies), :538-611 (fft_downshift, opus_fft_impl)
 * and celt/mdct.c:122-266 (forward), :268-388 (backward).  int32 data x int16 twiddles,
 * S_MUL = MULT16_32_Q15, S_MUL2 = MULT16_32_Q16 (celt/_kiss_fft_guts.h:58-62), wrap-around adds. */
#include "oc_celt.h"

typedef struct { i32 r, i; } cpx;
#define SMUL(a, b) mult16_32_q15((b), (a))
#define SMUL2(a, b) mult16_32_q16((b), (a))

static inline cpx cmul(cpx a, int twr, int twi)
{
   cpx m;
   m.r = sub32(SMUL(a.r, twr), SMUL(a.i, twi));
   m.i = add32(SMUL(a.r, twi), SMUL(a.i, twr));
   return m;
}
static inline cpx cadd(cpx a, cpx b) { cpx c = {add32(a.r, b.r), add32(a.i, b.i)}; return c; }
static inline cpx csub(cpx a, cpx b) { cpx c = {sub32(a.r, b.r), sub32(a.i, b.i)}; return c; }
#define TW(k) oc_fft_twiddles[2 * (k)], oc_fft_twiddles[2 * (k) + 1]

/* radix-2 stage that always follows a radix-4 (m==4), kiss_fft.c:52 */
static void bfly2(cpx *F, int N)
{
   const int tw = 23170; /* QCONST32(0.7071067812, 15) */
   for (int i = 0; i < N; i++) {
      cpx *F2 = F + 4, t;
      t = F2[0]; F2[0] = csub(F[0], t); F[0] = cadd(F[0], t);
      t.r = SMUL(add32(F2[1].r, F2[1].i), tw);
      t.i = SMUL(sub32(F2[1].i, F2[1].r), tw);
      F2[1] = csub(F[1], t); F[1] = cadd(F[1], t);
      t.r = F2[2].i; t.i = neg32(F2[2].r);
      F2[2] = csub(F[2], t); F[2] = cadd(F[2], t);
      t.r = SMUL(sub32(F2[3].i, F2[3].r), tw);
      t.i = SMUL(neg32(add32(F2[3].i, F2[3].r)), tw);
      F2[3] = csub(F[3], t); F[3] = cadd(F[3], t);
      F += 8;
   }
}
/* kiss_fft.c:108 */
static void bfly4(cpx *Fbeg, int fstride, int m, int N, int mm)
{
   if (m == 1) {
      cpx *F = Fbeg;
      for (int i = 0; i < N; i++) {
         cpx s0 = csub(F[0], F[2]);
         F[0] = cadd(F[0], F[2]);
         cpx s1 = cadd(F[1], F[3]);
         F[2] = csub(F[0], s1);
         F[0] = cadd(F[0], s1);
         s1 = csub(F[1], F[3]);
         F[1].r = add32(s0.r, s1.i); F[1].i = sub32(s0.i, s1.r);
         F[3].r = sub32(s0.r, s1.i); F[3].i = add32(s0.i, s1.r);
         F += 4;
      }
      return;
   }
   for (int i = 0; i < N; i++) {
      cpx *F = Fbeg + i * mm;
      for (int j = 0; j < m; j++) {
         cpx s0 = cmul(F[m], TW(j * fstride));
         cpx s1 = cmul(F[2 * m], TW(j * fstride * 2));
         cpx s2 = cmul(F[3 * m], TW(j * fstride * 3));
         cpx s5 = csub(F[0], s1);
         F[0] = cadd(F[0], s1);
         cpx s3 = cadd(s0, s2), s4 = csub(s0, s2);
         F[2 * m] = csub(F[0], s3);
         F[0] = cadd(F[0], s3);
         F[m].r = add32(s5.r, s4.i); F[m].i = sub32(s5.i, s4.r);
         F[3 * m].r = sub32(s5.r, s4.i); F[3 * m].i = add32(s5.i, s4.r);
         ++F;
      }
   }
}
/* kiss_fft.c:180 */
static void bfly3(cpx *Fbeg, int fstride, int m, int N, int mm)
{
   const int epi3i = -28378; /* -QCONST32(0.86602540, 15) */
   for (int i = 0; i < N; i++) {
      cpx *F = Fbeg + i * mm;
      for (int k = 0; k < m; k++) {
         cpx s1 = cmul(F[m], TW(k * fstride));
         cpx s2 = cmul(F[2 * m], TW(k * fstride * 2));
         cpx s3 = cadd(s1, s2), s0 = csub(s1, s2);
         F[m].r = sub32(F[0].r, s3.r >> 1);
         F[m].i = sub32(F[0].i, s3.i >> 1);
         s0.r = SMUL(s0.r, epi3i); s0.i = SMUL(s0.i, epi3i);
         F[0] = cadd(F[0], s3);
         F[2 * m].r = add32(F[m].r, s0.i);
         F[2 * m].i = sub32(F[m].i, s0.r);
         F[m].r = sub32(F[m].r, s0.i);
         F[m].i = add32(F[m].i, s0.r);
         ++F;
      }
   }
}
/* kiss_fft.c:239 */
static void bfly5(cpx *Fbeg, int fstride, int m, int N, int mm)
{
   const int yar = 10126, yai = -31164, ybr = -26510, ybi = -19261;
   for (int i = 0; i < N; i++) {
      cpx *F0 = Fbeg + i * mm, *F1 = F0 + m, *F2 = F0 + 2 * m, *F3 = F0 + 3 * m, *F4 = F0 + 4 * m;
      for (int u = 0; u < m; ++u) {
         cpx s0 = *F0;
         cpx s1 = cmul(*F1, TW(u * fstride));
         cpx s2 = cmul(*F2, TW(2 * u * fstride));
         cpx s3 = cmul(*F3, TW(3 * u * fstride));
         cpx s4 = cmul(*F4, TW(4 * u * fstride));
         cpx s7 = cadd(s1, s4), s10 = csub(s1, s4), s8 = cadd(s2, s3), s9 = csub(s2, s3);
         F0->r = add32(F0->r, add32(s7.r, s8.r));
         F0->i = add32(F0->i, add32(s7.i, s8.i));
         cpx s5, s6, s11, s12;
         s5.r = add32(s0.r, add32(SMUL(s7.r, yar), SMUL(s8.r, ybr)));
         s5.i = add32(s0.i, add32(SMUL(s7.i, yar), SMUL(s8.i, ybr)));
         s6.r = add32(SMUL(s10.i, yai), SMUL(s9.i, ybi));
         s6.i = neg32(add32(SMUL(s10.r, yai), SMUL(s9.r, ybi)));
         *F1 = csub(s5, s6);
         *F4 = cadd(s5, s6);
         s11.r = add32(s0.r, add32(SMUL(s7.r, ybr), SMUL(s8.r, yar)));
         s11.i = add32(s0.i, add32(SMUL(s7.i, ybr), SMUL(s8.i, yar)));
         s12.r = sub32(SMUL(s9.i, yai), SMUL(s10.i, ybi));
         s12.i = sub32(SMUL(s10.r, ybi), SMUL(s9.r, yai));
         *F2 = cadd(s11, s12);
         *F3 = csub(s11, s12);
         ++F0; ++F1; ++F2; ++F3; ++F4;
      }
   }
}
/* kiss_fft.c:538 */
static void downshift(cpx *x, int N, int *total, int step)
{
   int shift = imin(step, *total);
   *total -= shift;
   if (shift == 1) for (int i = 0; i < N; i++) { x[i].r >>= 1; x[i].i >>= 1; }
   else if (shift > 0) for (int i = 0; i < N; i++) { x[i].r = pshr32(x[i].r, shift); x[i].i = pshr32(x[i].i, shift); }
}
/* opus_fft_impl, kiss_fft.c:562.  idx selects 480/240/120/60. Input already in bit-reversed order. */
void oc_fft_impl(int idx, i32 *data, int dshift)
{
   cpx *fout = (cpx *)data;
   const int16_t *factors = oc_fft_factors + 16 * idx;
   int nfft = oc_fft_misc[4 * idx], stshift = oc_fft_misc[4 * idx + 3];
   int fstride[9], L = 0, m, m2, p;
   int shift = stshift > 0 ? stshift : 0;
   fstride[0] = 1;
   do { p = factors[2 * L]; m = factors[2 * L + 1]; fstride[L + 1] = fstride[L] * p; L++; } while (m != 1);
   m = factors[2 * L - 1];
   for (int i = L - 1; i >= 0; i--) {
      m2 = i != 0 ? factors[2 * i - 1] : 1;
      switch (factors[2 * i]) {
      case 2: downshift(fout, nfft, &dshift, 1); bfly2(fout, fstride[i]); break;
      case 4: downshift(fout, nfft, &dshift, 2); bfly4(fout, fstride[i] << shift, m, fstride[i], m2); break;
      case 3: downshift(fout, nfft, &dshift, 2); bfly3(fout, fstride[i] << shift, m, fstride[i], m2); break;
      case 5: downshift(fout, nfft, &dshift, 3); bfly5(fout, fstride[i] << shift, m, fstride[i], m2); break;
      }
      m = m2;
   }
   downshift(fout, nfft, &dshift, dshift);
}

static const int trig_off[4] = {0, 960, 1440, 1680};

/* clt_mdct_forward_c, mdct.c:122.  in: N2+overlap samples; out: N2 bins at the given stride. */
void oc_mdct_forward(const i32 *in, i32 *out, int shift, int stride)
{
   int N = 1920 >> shift, N2 = N >> 1, N4 = N >> 2, overlap = OVERLAP;
   const int16_t *trig = oc_mdct_trig + trig_off[shift];
   const int16_t *bitrev = oc_fft_bitrev + oc_fft_bitrev_off[shift];
   int scale = oc_fft_misc[4 * shift + 1], scale_shift = oc_fft_misc[4 * shift + 2] - 1, headroom;
   i32 f[960];
   cpx f2[480];
   const int16_t *window = oc_window;
   {
      const i32 *xp1 = in + (overlap >> 1), *xp2 = in + N2 - 1 + (overlap >> 1);
      i32 *yp = f;
      const int16_t *wp1 = window + (overlap >> 1), *wp2 = window + (overlap >> 1) - 1;
      int i;
      for (i = 0; i < ((overlap + 3) >> 2); i++) {
         *yp++ = add32(SMUL(xp1[N2], *wp2), SMUL(*xp2, *wp1));
         *yp++ = sub32(SMUL(*xp1, *wp1), SMUL(xp2[-N2], *wp2));
         xp1 += 2; xp2 -= 2; wp1 += 2; wp2 -= 2;
      }
      wp1 = window; wp2 = window + overlap - 1;
      for (; i < N4 - ((overlap + 3) >> 2); i++) {
         *yp++ = *xp2; *yp++ = *xp1;
         xp1 += 2; xp2 -= 2;
      }
      for (; i < N4; i++) {
         *yp++ = add32(neg32(SMUL(xp1[-N2], *wp1)), SMUL(*xp2, *wp2));
         *yp++ = add32(SMUL(*xp1, *wp2), SMUL(xp2[N2], *wp1));
         xp1 += 2; xp2 -= 2; wp1 += 2; wp2 -= 2;
      }
   }
   {
      const i32 *yp = f;
      i32 maxval = 1;
      for (int i = 0; i < N4; i++) {
         int t0 = trig[i], t1 = trig[N4 + i];
         i32 re = *yp++, im = *yp++;
         i32 yr = sub32(SMUL(re, t0), SMUL(im, t1));
         i32 yi = add32(SMUL(im, t0), SMUL(re, t1));
         cpx yc = {SMUL2(yr, scale), SMUL2(yi, scale)};
         maxval = imax(maxval, imax(iabs(yc.r), iabs(yc.i)));
         f2[bitrev[i]] = yc;
      }
      headroom = imax(0, imin(scale_shift, 28 - celt_ilog2(maxval)));
   }
   oc_fft_impl(shift, (i32 *)f2, scale_shift - headroom);
   {
      const cpx *fp = f2;
      i32 *yp1 = out, *yp2 = out + stride * (N2 - 1);
      for (int i = 0; i < N4; i++) {
         int t0 = trig[i], t1 = trig[N4 + i];
         i32 yr = pshr32(sub32(SMUL(fp->i, t1), SMUL(fp->r, t0)), headroom);
         i32 yi = pshr32(add32(SMUL(fp->r, t1), SMUL(fp->i, t0)), headroom);
         *yp1 = yr; *yp2 = yi;
         fp++; yp1 += 2 * stride; yp2 -= 2 * stride;
      }
   }
}

/* clt_mdct_backward_c, mdct.c:268.  in: N2 bins at stride; out: N2+overlap samples (TDAC overlap-add into out[0..overlap)). */
void oc_mdct_backward(const i32 *in, i32 *out, int shift, int stride)
{
   int N = 1920 >> shift, N2 = N >> 1, N4 = N >> 2, overlap = OVERLAP;
   const int16_t *trig = oc_mdct_trig + trig_off[shift];
   const int16_t *bitrev = oc_fft_bitrev + oc_fft_bitrev_off[shift];
   int pre_shift, post_shift, fft_shift;
   {
      i32 sumval = N2, maxval = 0;
      for (int i = 0; i < N2; i++) {
         maxval = imax(maxval, iabs(in[i * stride]));
         sumval = add32(sumval, iabs(in[i * stride] >> 11));
      }
      pre_shift = imax(0, 29 - celt_zlog2(1 + maxval));
      post_shift = imax(0, 19 - celt_ilog2(iabs(sumval)));
      post_shift = imin(post_shift, pre_shift);
      fft_shift = pre_shift - post_shift;
   }
   {
      const i32 *xp1 = in, *xp2 = in + stride * (N2 - 1);
      i32 *yp = out + (overlap >> 1);
      for (int i = 0; i < N4; i++) {
         int rev = bitrev[i];
         i32 x1 = shl32(*xp1, pre_shift), x2 = shl32(*xp2, pre_shift);
         i32 yr = add32(SMUL(x2, trig[i]), SMUL(x1, trig[N4 + i]));
         i32 yi = sub32(SMUL(x1, trig[i]), SMUL(x2, trig[N4 + i]));
         yp[2 * rev + 1] = yr; yp[2 * rev] = yi;
         xp1 += 2 * stride; xp2 -= 2 * stride;
      }
   }
   oc_fft_impl(shift, out + (overlap >> 1), fft_shift);
   {
      i32 *yp0 = out + (overlap >> 1), *yp1 = out + (overlap >> 1) + N2 - 2;
      for (int i = 0; i < (N4 + 1) >> 1; i++) {
         i32 re = yp0[1], im = yp0[0];
         int t0 = trig[i], t1 = trig[N4 + i];
         i32 yr = pshr32(add32(SMUL(re, t0), SMUL(im, t1)), post_shift);
         i32 yi = pshr32(sub32(SMUL(re, t1), SMUL(im, t0)), post_shift);
         re = yp1[1]; im = yp1[0];
         yp0[0] = yr; yp1[1] = yi;
         t0 = trig[N4 - i - 1]; t1 = trig[N2 - i - 1];
         yr = pshr32(add32(SMUL(re, t0), SMUL(im, t1)), post_shift);
         yi = pshr32(sub32(SMUL(re, t1), SMUL(im, t0)), post_shift);
         yp1[0] = yr; yp0[1] = yi;
         yp0 += 2; yp1 -= 2;
      }
   }
   {
      i32 *xp1 = out + overlap - 1, *yp1 = out;
      const int16_t *wp1 = oc_window, *wp2 = oc_window + overlap - 1;
      for (int i = 0; i < overlap / 2; i++) {
         i32 x1 = *xp1, x2 = *yp1;
         *yp1++ = sub32(SMUL(x2, *wp2), SMUL(x1, *wp1));
         *xp1-- = add32(SMUL(x2, *wp1), SMUL(x1, *wp2));
         wp1++; wp2--;
      }
   }
}

/* celt_mdct.h — wave-parallel forward MDCT (N = 1920 >> shift) on LDS data.
 * Arithmetic is the reference's fixed-point transform bit for bit: fold/window + pre-rotation
 * (celt/mdct.c:122-266), mixed-radix FFT with per-stage down-shifts (celt/kiss_fft.c:52-312, :538-611;
 * radix order 480:{4,2,4,3,5} 240:{4,4,3,5} 120:{4,2,3,5} 60:{4,3,5} as processed), post-rotation.
 * Mapping: one lane per butterfly; the N/4-point complex FFT runs in place in the channel's output half;
 * for transient frames the 8 short transforms of a channel run side by side in the same buffer.
 * The butterfly network is order-fixed (per-stage shifts), only the butterflies *within* a stage run in
 * parallel, which is exact because they touch disjoint elements. */
#ifndef OPUS_AMD_CELT_MDCT_H
#define OPUS_AMD_CELT_MDCT_H

struct cpx32 { i32 r, i; };
#define SMUL(a, b) mult16_32_q15((b), (a))
#define SMUL2(a, b) mult16_32_q16((b), (a))
WV_DEV cpx32 c_mul(cpx32 a, int twr, int twi) { cpx32 m; m.r = sub32(SMUL(a.r, twr), SMUL(a.i, twi)); m.i = add32(SMUL(a.r, twi), SMUL(a.i, twr)); return m; }
WV_DEV cpx32 c_add(cpx32 a, cpx32 b) { cpx32 c; c.r = add32(a.r, b.r); c.i = add32(a.i, b.i); return c; }
WV_DEV cpx32 c_sub(cpx32 a, cpx32 b) { cpx32 c; c.r = sub32(a.r, b.r); c.i = sub32(a.i, b.i); return c; }
WV_DEV i32 fft_shift_val(i32 x, int s) { return s == 0 ? x : (s == 1 ? (x >> 1) : pshr32(x, s)); }
WV_DEV cpx32 c_ld(const WV_LDS i32 *d, int idx, int s) { cpx32 c; c.r = fft_shift_val(d[2 * idx], s); c.i = fft_shift_val(d[2 * idx + 1], s); return c; }
WV_DEV void c_st(WV_LDS i32 *d, int idx, cpx32 c) { d[2 * idx] = c.r; d[2 * idx + 1] = c.i; }
/* the p - 1 twiddles of a butterfly, (re, im) in one word each: W^(k j tws), k = 1 .. p - 1 (the unused ones of the smaller radices read entry 0).  The loads carry no
 * condition, so a stage can ask for the next trip's set before it works on this trip's (fft_stage) */
struct TwSet { u32 t[4]; };
WV_DEV TwSet fft_tw_load(int p, int jt) { TwSet s; const u32 *T = (const u32 *)ct_fft_twiddles;
#pragma unroll
   for (int k = 0; k < 4; k++) s.t[k] = T[k + 1 < p ? (k + 1) * jt : 0];
   return s; }
#define TWR(k) ((int)(i16)tw.t[(k) - 1])
#define TWI(k) ((i32)tw.t[(k) - 1] >> 16)

/* q / m for 0 <= q < 4096, 1 <= m <= 1024 with rm = 1.f / m: (q + .5) / m is at least .5 / m away from an integer, the float error of the product is below
 * 4096 * 2^-22 -- so the truncation is the exact quotient (three instructions instead of the thirty-odd of a 32-bit division by a run-time divisor) */
WV_DEV int fft_div(int q, float rm) { return (int)(((float)q + .5f) * rm); }
/* one FFT stage over nblk side-by-side transforms of nfft points each.
 * p radix, m butterfly span, ngrp groups, mm group pitch, tws twiddle stride;
 * total[b] = down-shift budget of block b before this stage, step = max shift of this radix. */
WV_DEV void fft_stage(WV_LDS i32 *data, int nblk, int nfft, int p, int m, int ngrp, int mm, int tws,
      const WV_LDS int *remaining, int step)
{
   int per = nfft / p;                 /* butterflies per transform in this stage */
   int tot = per * nblk;
   const float rper = 1.f / (float)per, rm = 1.f / (float)m;
#define FFT_DIVM(q) fft_div((q), rm)
   const bool twiddled = p != 2 && m != 1;
   TwSet nx = {{0, 0, 0, 0}};
   if (twiddled) { const int w = imin(wv_lane(), tot - 1), blk = nblk == 1 ? 0 : fft_div(w, rper), q = w - blk * per; nx = fft_tw_load(p, (q - FFT_DIVM(q) * m) * tws); }
   for (int w = wv_lane(); w < tot; w += WV_WIDTH) {
      int blk = nblk == 1 ? 0 : fft_div(w, rper), q = w - blk * per;
      const TwSet tw = nx;
      if (twiddled) { const int w1 = imin(w + WV_WIDTH, tot - 1), blk1 = nblk == 1 ? 0 : fft_div(w1, rper), q1 = w1 - blk1 * per; nx = fft_tw_load(p, (q1 - FFT_DIVM(q1) * m) * tws); }
      int rem = remaining[blk];
      int s = rem < step ? rem : step;
      WV_LDS i32 *F = data + 2 * blk * nfft;
      if (p == 2) {                    /* m == 4: radix-2 that follows a radix-4 (kiss_fft.c:52) */
         int g = q >> 2, j = q & 3;
         int a = g * 8 + j, b = a + 4;
         cpx32 x = c_ld(F, a, s), y = c_ld(F, b, s), t;
         const int tw = 23170;
         if (j == 0) t = y;
         else if (j == 1) { t.r = SMUL(add32(y.r, y.i), tw); t.i = SMUL(sub32(y.i, y.r), tw); }
         else if (j == 2) { t.r = y.i; t.i = neg32(y.r); }
         else { t.r = SMUL(sub32(y.i, y.r), tw); t.i = SMUL(neg32(add32(y.i, y.r)), tw); }
         c_st(F, b, c_sub(x, t));
         c_st(F, a, c_add(x, t));
      } else if (p == 4) {
         int g = FFT_DIVM(q), j = q - g * m;
         int b0 = g * mm + j;
         cpx32 f0 = c_ld(F, b0, s), f1 = c_ld(F, b0 + m, s), f2 = c_ld(F, b0 + 2 * m, s), f3 = c_ld(F, b0 + 3 * m, s);
         if (m == 1) {                 /* twiddle-free first stage (kiss_fft.c:117) */
            cpx32 s0 = c_sub(f0, f2);
            f0 = c_add(f0, f2);
            cpx32 s1 = c_add(f1, f3);
            f2 = c_sub(f0, s1);
            f0 = c_add(f0, s1);
            s1 = c_sub(f1, f3);
            f1.r = add32(s0.r, s1.i); f1.i = sub32(s0.i, s1.r);
            f3.r = sub32(s0.r, s1.i); f3.i = add32(s0.i, s1.r);
         } else {
            cpx32 s0 = c_mul(f1, TWR(1), TWI(1));
            cpx32 s1 = c_mul(f2, TWR(2), TWI(2));
            cpx32 s2 = c_mul(f3, TWR(3), TWI(3));
            cpx32 s5 = c_sub(f0, s1);
            f0 = c_add(f0, s1);
            cpx32 s3 = c_add(s0, s2), s4 = c_sub(s0, s2);
            f2 = c_sub(f0, s3);
            f0 = c_add(f0, s3);
            f1.r = add32(s5.r, s4.i); f1.i = sub32(s5.i, s4.r);
            f3.r = sub32(s5.r, s4.i); f3.i = add32(s5.i, s4.r);
         }
         c_st(F, b0, f0); c_st(F, b0 + m, f1); c_st(F, b0 + 2 * m, f2); c_st(F, b0 + 3 * m, f3);
      } else if (p == 3) {
         const int epi3i = -28378;
         int g = FFT_DIVM(q), j = q - g * m;
         int b0 = g * mm + j;
         cpx32 f0 = c_ld(F, b0, s), f1 = c_ld(F, b0 + m, s), f2 = c_ld(F, b0 + 2 * m, s);
         cpx32 s1 = c_mul(f1, TWR(1), TWI(1));
         cpx32 s2 = c_mul(f2, TWR(2), TWI(2));
         cpx32 s3 = c_add(s1, s2), s0 = c_sub(s1, s2);
         f1.r = sub32(f0.r, s3.r >> 1);
         f1.i = sub32(f0.i, s3.i >> 1);
         s0.r = SMUL(s0.r, epi3i); s0.i = SMUL(s0.i, epi3i);
         f0 = c_add(f0, s3);
         f2.r = add32(f1.r, s0.i);
         f2.i = sub32(f1.i, s0.r);
         f1.r = sub32(f1.r, s0.i);
         f1.i = add32(f1.i, s0.r);
         c_st(F, b0, f0); c_st(F, b0 + m, f1); c_st(F, b0 + 2 * m, f2);
      } else {                         /* p == 5 */
         const int yar = 10126, yai = -31164, ybr = -26510, ybi = -19261;
         int g = FFT_DIVM(q), u = q - g * m;
         int b0 = g * mm + u;
         cpx32 s0 = c_ld(F, b0, s);
         cpx32 s1 = c_mul(c_ld(F, b0 + m, s), TWR(1), TWI(1));
         cpx32 s2 = c_mul(c_ld(F, b0 + 2 * m, s), TWR(2), TWI(2));
         cpx32 s3 = c_mul(c_ld(F, b0 + 3 * m, s), TWR(3), TWI(3));
         cpx32 s4 = c_mul(c_ld(F, b0 + 4 * m, s), TWR(4), TWI(4));
         cpx32 s7 = c_add(s1, s4), s10 = c_sub(s1, s4), s8 = c_add(s2, s3), s9 = c_sub(s2, s3);
         cpx32 o0, s5, s6, s11, s12;
         o0.r = add32(s0.r, add32(s7.r, s8.r));
         o0.i = add32(s0.i, add32(s7.i, s8.i));
         s5.r = add32(s0.r, add32(SMUL(s7.r, yar), SMUL(s8.r, ybr)));
         s5.i = add32(s0.i, add32(SMUL(s7.i, yar), SMUL(s8.i, ybr)));
         s6.r = add32(SMUL(s10.i, yai), SMUL(s9.i, ybi));
         s6.i = neg32(add32(SMUL(s10.r, yai), SMUL(s9.r, ybi)));
         s11.r = add32(s0.r, add32(SMUL(s7.r, ybr), SMUL(s8.r, yar)));
         s11.i = add32(s0.i, add32(SMUL(s7.i, ybr), SMUL(s8.i, yar)));
         s12.r = sub32(SMUL(s9.i, yai), SMUL(s10.i, ybi));
         s12.i = sub32(SMUL(s10.r, ybi), SMUL(s9.r, yai));
         c_st(F, b0, o0);
         c_st(F, b0 + m, c_sub(s5, s6));
         c_st(F, b0 + 4 * m, c_add(s5, s6));
         c_st(F, b0 + 2 * m, c_add(s11, s12));
         c_st(F, b0 + 3 * m, c_sub(s11, s12));
      }
   }
#undef FFT_DIVM
}

/* opus_fft_impl over nblk transforms (kiss_fft.c:562).  remaining[b] holds block b's down-shift budget
 * and is updated; whatever is left after the last stage is returned to the caller through remaining[]. */
WV_DEV void fft_forward(WV_LDS i32 *data, int idx, int nblk, WV_LDS int *remaining)
{
   const int16_t *factors = ct_fft_factors + 16 * idx;
   int nfft = ct_fft_misc[4 * idx], stshift = ct_fft_misc[4 * idx + 3];
   int shift = stshift > 0 ? stshift : 0;
   int fstride[9], L = 0, m, m2, p;
   fstride[0] = 1;
   do { p = factors[2 * L]; m = factors[2 * L + 1]; fstride[L + 1] = fstride[L] * p; L++; } while (m != 1);
   m = factors[2 * L - 1];
   for (int i = L - 1; i >= 0; i--) {
      m2 = i != 0 ? factors[2 * i - 1] : 1;
      p = factors[2 * i];
      int step = p == 2 ? 1 : (p == 5 ? 3 : 2);
      fft_stage(data, nblk, nfft, p, m, fstride[i], m2, fstride[i] << shift, remaining, step);
      wv_sync();
      if (wv_lane() < nblk) { int r = remaining[wv_lane()]; remaining[wv_lane()] = r - (r < step ? r : step); }
      wv_sync();
      m = m2;
   }
}

/* sample t of a channel's time signal [head | body]; with both parts in HBM the part is chosen by the address, so a sample is one load and no branch */
WV_DEV i32 mdct_xin(const i32 *head, const i32 *body, int t) { return (t < OA_OVERLAP ? head : body - OA_OVERLAP)[t]; }
WV_DEV i32 mdct_xin(const i32 *head, i32 *body, int t) { return (t < OA_OVERLAP ? head : (const i32 *)body - OA_OVERLAP)[t]; }
template <class BodyPtr> WV_DEV i32 mdct_xin(const i32 *head, BodyPtr body, int t) { return t < OA_OVERLAP ? head[t] : body[t - OA_OVERLAP]; }
/* Forward MDCTs of one channel: B transforms of N2 = (960>>shift) output bins each.  The channel's time signal is
 * [head | body]: the first `overlap` samples (last frame's filtered tail, in_mem) come from the stream's HBM record, the rest (this frame's comb-filtered input) from the HBM scratch;
 * input block b starts at sample b*N2 (N2+overlap samples); output bin k of block b goes to out[b + k*B] (interleaved,
 * stride B).  The complex FFT runs IN PLACE in out[] (B*N2 words); the post-rotation gathers every result into
 * registers before the first scattered store.  aux: >= 2*8 ints. */
template <class BodyPtr> WV_DEV void mdct_forward_blocks(const i32 *head, BodyPtr body, WV_LDS i32 *out, int shift, int B, WV_LDS int *aux)
{
   const int N = 1920 >> shift, N2 = N >> 1, N4 = N >> 2, overlap = OA_OVERLAP;
   const int trig_off = shift == 0 ? 0 : (shift == 1 ? 960 : (shift == 2 ? 1440 : 1680));
   const int16_t *trig = ct_mdct_trig + trig_off;
   const int16_t *bitrev = ct_fft_bitrev + ct_fft_bitrev_off[shift];
   const int scale = ct_fft_misc[4 * shift + 1], scale_shift = ct_fft_misc[4 * shift + 2] - 1;
   WV_LDS int *headroom = aux, *remaining = aux + 8;
   WV_LDS i32 *fbuf = out;
   const int lane = wv_lane();
   AN_TIC();
#define XIN(t) mdct_xin(head, body, (t))
   const int E = (overlap + 3) >> 2;
   for (int b = 0; b < B; b++) {
      const int x0 = b * N2;
      WV_LDS i32 *f2 = fbuf + 2 * b * N4;
      i32 maxval = 1;
      for (int i0 = lane; i0 < N4; i0 += 4 * WV_WIDTH) {        /* four points per trip (a long block's 240 in one): the loads of all ahead of the arithmetic of any */
         i32 re[4], im[4]; int t0[4], t1[4], rv[4];
#pragma unroll
         for (int u = 0; u < 4; u++) {
            re[u] = im[u] = 0; t0[u] = t1[u] = rv[u] = 0;
            if (i0 - lane + u * WV_WIDTH < N4) {                  /* (wave-uniform: a short block's 30 points are one trip) */
               const int i = imin(i0 + u * WV_WIDTH, N4 - 1);    /* (the clamped point of a ragged trip is loaded again and not stored) */
               const int p1 = x0 + (overlap >> 1) + 2 * i, p2 = x0 + N2 - 1 + (overlap >> 1) - 2 * i;
               re[u] = XIN(p2); im[u] = XIN(p1);
               t0[u] = trig[i]; t1[u] = trig[N4 + i]; rv[u] = bitrev[i];
            }
         }
#pragma unroll
         for (int u = 0; u < 4; u++) {
            const int i = i0 + u * WV_WIDTH;
            if (i < N4) {
               const int p1 = x0 + (overlap >> 1) + 2 * i, p2 = x0 + N2 - 1 + (overlap >> 1) - 2 * i;
               if (i < E) {
                  int w1 = ct_window[(overlap >> 1) + 2 * i], w2 = ct_window[(overlap >> 1) - 1 - 2 * i];
                  re[u] = add32(SMUL(XIN(p1 + N2), w2), SMUL(re[u], w1));
                  im[u] = sub32(SMUL(im[u], w1), SMUL(XIN(p2 - N2), w2));
               } else if (i >= N4 - E) {
                  int k = i - (N4 - E);
                  int w1 = ct_window[2 * k], w2 = ct_window[overlap - 1 - 2 * k];
                  re[u] = add32(neg32(SMUL(XIN(p1 - N2), w1)), SMUL(re[u], w2));
                  im[u] = add32(SMUL(im[u], w2), SMUL(XIN(p2 + N2), w1));
               }
               i32 yr = sub32(SMUL(re[u], t0[u]), SMUL(im[u], t1[u]));
               i32 yi = add32(SMUL(im[u], t0[u]), SMUL(re[u], t1[u]));
               yr = SMUL2(yr, scale); yi = SMUL2(yi, scale);
               maxval = imax(maxval, imax(iabs(yr), iabs(yi)));
               f2[2 * rv[u]] = yr; f2[2 * rv[u] + 1] = yi;
            }
         }
      }
      maxval = wv_max(maxval);
      int hr = imax(0, imin(scale_shift, 28 - celt_ilog2(maxval)));
      if (lane == 0) { headroom[b] = hr; remaining[b] = scale_shift - hr; }
   }
#undef XIN
   wv_sync();
   AN_TOC(16);
   fft_forward(fbuf, shift, B, remaining);
   AN_TOC(17);
   {  /* post-rotation, in place: B*N4 <= 480 complex points -> at most 8 per lane held in registers across the barrier */
      i32 vr[8], vi[8];
      const float rN4 = 1.f / (float)N4;
      int tr0[8], tr1[8];
#pragma unroll
      for (int t = 0; t < 8; t++) {                                /* (the sixteen table loads of a lane in flight together) */
         const int w = imin(lane + t * WV_WIDTH, B * N4 - 1), b = B == 1 ? 0 : fft_div(w, rN4), i = w - b * N4;
         tr0[t] = trig[i]; tr1[t] = trig[N4 + i];
      }
#pragma unroll
      for (int t = 0; t < 8; t++) {
         int w = lane + t * WV_WIDTH;
         if (w < B * N4) {
            int b = B == 1 ? 0 : fft_div(w, rN4), i = w - b * N4;
            int hr = headroom[b], left = remaining[b];
            cpx32 fp = c_ld(fbuf + 2 * b * N4, i, left);
            int t0 = tr0[t], t1 = tr1[t];
            vr[t] = pshr32(sub32(SMUL(fp.i, t1), SMUL(fp.r, t0)), hr);
            vi[t] = pshr32(add32(SMUL(fp.r, t1), SMUL(fp.i, t0)), hr);
         }
      }
      wv_sync();
#pragma unroll
      for (int t = 0; t < 8; t++) {
         int w = lane + t * WV_WIDTH;
         if (w < B * N4) {
            int b = B == 1 ? 0 : fft_div(w, rN4), i = w - b * N4;
            out[b + B * (2 * i)] = vr[t];
            out[b + B * (N2 - 1 - 2 * i)] = vi[t];
         }
      }
   }
   wv_sync();
   AN_TOC(18);
}
#endif

/* celt_enc_pitch.h — pitch pre-filter of the CELT encoder on one wavefront.
 * Reference: celt/celt_encoder.c:1405 run_prefilter, celt/pitch.c:45 find_best_pitch, :103 celt_fir5,
 * :140 pitch_downsample, :230 celt_pitch_xcorr, :307 pitch_search, :418 compute_pitch_gain, :454 remove_doubling,
 * celt/celt_lpc.c:37 _celt_lpc, :284 _celt_autocorr, celt/celt.c:166/:238 comb filter.
 * Every multi-tap sum is a mod-2^32 sum of individually rounded terms, so lanes may add in any order
 * (SURVEY.md appendix A "reduction-order rule"); decisions stay on uniform scalar code. */
#ifndef OPUS_AMD_CELT_ENC_PITCH_H
#define OPUS_AMD_CELT_ENC_PITCH_H

WV_DEV i32 wave_inner16(const WV_LDS i16 *x, const WV_LDS i16 *y, int N)
{
   i32 s = 0;
   FOR_LANES(i, N) s = mac16_16(s, x[i], y[i]);
   return wv_sum(s);
}
WV_DEV void wave_dual_inner16(const WV_LDS i16 *x, const WV_LDS i16 *y1, const WV_LDS i16 *y2, int N, i32 *xy1, i32 *xy2)
{
   i32 a = 0, b = 0;
   FOR_LANES(i, N) { a = mac16_16(a, x[i], y1[i]); b = mac16_16(b, x[i], y2[i]); }
   *xy1 = wv_sum(a); *xy2 = wv_sum(b);
}

/* _celt_lpc for order 4 on uniform registers (celt_lpc.c:37) */
WV_DEV void celt_lpc4(i16 *_lpc, const i32 *ac)
{
   const int p = 4;
   i32 lpc[4] = {0, 0, 0, 0}, r, error = ac[0];
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
         i32 chirp = QC32(0.999, 16) - shl32(maxabs - 32767, 14) / ((maxabs * (idx + 1)) >> 2);
         i32 chirp_m1 = chirp - 65536;
         for (int i = 0; i < p - 1; i++) {
            lpc[i] = mult32_32_q16(chirp, lpc[i]);
            chirp += pshr32(chirp * chirp_m1, 16);
         }
         lpc[p - 1] = mult32_32_q16(chirp, lpc[p - 1]);
      } else break;
   }
   if (iter == 10) { _lpc[0] = 4096; _lpc[1] = _lpc[2] = _lpc[3] = 0; }
   else for (int i = 0; i < p; i++) _lpc[i] = extract16(pshr32(lpc[i], 13));
}

/* pitch_downsample (pitch.c:140, factor 2) of one or two channels given by sample accessors (src_at): x_lp = raw low-passed signal
 * (scratch, len words), xx = scaled copy for the autocorrelation and then the result (len words).  Shared by the encoder's
 * pre-filter (source: recomputed pre-emphasis) and the decoder's loss concealment (source: synthesis history ring). */
WV_DEV i32 src_at(const PreSrc &p, int j) { return pre_at(p, j); }
template <class Src>
WV_DEVN void pitch_downsample_src(WV_LDS i16 *x_lp, WV_LDS i16 *xx, const Src &p0, const Src &p1, int len, int C)
{
   i32 maxabs = 0;
   for (int i0 = wv_lane(); i0 < 2 * len; i0 += 8 * WV_WIDTH) {      /* (eight loads per channel in flight; a clamped index repeats a sample, which a maximum does not see) */
      i32 a[8], b[8];
#pragma unroll
      for (int t = 0; t < 8; t++) { const int i = imin(i0 + t * WV_WIDTH, 2 * len - 1); a[t] = src_at(p0, i); b[t] = C == 2 ? src_at(p1, i) : 0; }
#pragma unroll
      for (int t = 0; t < 8; t++) maxabs = imax(maxabs, imax(iabs(a[t]), iabs(b[t])));
   }
   maxabs = wv_max(maxabs);
   if (maxabs < 1) maxabs = 1;
   int shift = celt_ilog2(maxabs) - 10;
   if (shift < 0) shift = 0;
   if (C == 2) shift++;
   for (int i0 = wv_lane(); i0 < len; i0 += 4 * WV_WIDTH) {
      i32 a[4][3], b[4][3];
#pragma unroll
      for (int t = 0; t < 4; t++) {
         const int i = imin(i0 + t * WV_WIDTH, len - 1);
#pragma unroll
         for (int k = 0; k < 3; k++) { const int j = imax(2 * i - 1 + k, 0); a[t][k] = src_at(p0, j); b[t][k] = C == 2 ? src_at(p1, j) : 0; }
      }
#pragma unroll
      for (int t = 0; t < 4; t++) {
         const int i = i0 + t * WV_WIDTH;
         if (i < len) {
            i16 v = (i16)((i == 0 ? 0 : a[t][0] >> (shift + 2)) + (a[t][2] >> (shift + 2)) + (a[t][1] >> (shift + 1)));
            if (C == 2) v = (i16)(v + (i == 0 ? 0 : b[t][0] >> (shift + 2)) + (b[t][2] >> (shift + 2)) + (b[t][1] >> (shift + 1)));
            x_lp[i] = v;
         }
      }
   }
   wv_sync();
   /* _celt_autocorr(x_lp, ac, NULL, 0, 4, len) */
   i32 ac[5];
   {
      const int n = len, lag = 4, fastN = n - lag;
      int ac0_shift = celt_ilog2(n + (n >> 4));
      i32 a0 = 0;
      FOR_LANES(i, n) a0 += mult16_16(x_lp[i], x_lp[i]) >> ac0_shift;
      i32 ac0 = add32(1 + (n << 7), wv_sum(a0));
      ac0 += ac0 >> 7;
      int sh = celt_ilog2(ac0) - 30 + ac0_shift + 1;
      sh = sh / 2;
      const WV_LDS i16 *xptr = x_lp;
      if (sh > 0) {
         FOR_LANES(i, n) xx[i] = (i16)pshr32(x_lp[i], sh);
         xptr = xx;
         wv_sync();
      } else sh = 0;
      for (int k = 0; k <= lag; k++) {
         i32 s = 0;
         FOR_LANES(i, fastN) s = mac16_16(s, xptr[i], xptr[i + k]);
         for (int i = k + fastN + wv_lane(); i < n; i += WV_WIDTH) s = mac16_16(s, xptr[i], xptr[i - k]);
         ac[k] = wv_sum(s);
      }
      sh = 2 * sh;
      if (sh <= 0) ac[0] += shl32(1, -sh);
      if (ac[0] < 268435456) {
         int s2 = 29 - ec_ilog(ac[0]);
         for (int i = 0; i <= lag; i++) ac[i] = shl32(ac[i], s2);
      } else if (ac[0] >= 536870912) {
         int s2 = 1;
         if (ac[0] >= 1073741824) s2++;
         for (int i = 0; i <= lag; i++) ac[i] = ac[i] >> s2;
      }
   }
   ac[0] += ac[0] >> 13;
   for (int i = 1; i <= 4; i++) ac[i] -= mult16_32_q15(2 * i * i, ac[i]);
   i16 lpc[4], lpc2[5], tmp = Q15ONE, c1 = QC16(.8f, 15);
   celt_lpc4(lpc, ac);
   for (int i = 0; i < 4; i++) { tmp = (i16)mult16_16_q15(QC16(.9f, 15), tmp); lpc[i] = (i16)mult16_16_q15(lpc[i], tmp); }
   lpc2[0] = (i16)(lpc[0] + QC16(.8f, SIG_SHIFT));
   lpc2[1] = (i16)(lpc[1] + mult16_16_q15(c1, lpc[0]));
   lpc2[2] = (i16)(lpc[2] + mult16_16_q15(c1, lpc[1]));
   lpc2[3] = (i16)(lpc[3] + mult16_16_q15(c1, lpc[2]));
   lpc2[4] = (i16)mult16_16_q15(c1, lpc[3]);
   wv_sync();
   /* celt_fir5: a pure FIR on the raw signal -> every output independently */
   FOR_LANES(i, len) {
      i32 sum = shl32((i32)x_lp[i], SIG_SHIFT);
      for (int k = 0; k < 5; k++) { int j = i - 1 - k; i32 m = j >= 0 ? x_lp[j] : 0; sum = mac16_16(sum, lpc2[k], m); }
      xx[i] = round16(sum, SIG_SHIFT);
   }
   wv_sync();
}

WV_DEV void pitch_downsample_wave(WV_LDS FrameLds *L, const PreSrc &p0, const PreSrc &p1, int len, int C)
{
   pitch_downsample_src((WV_LDS i16 *)L->BC.p.u.xcorr, L->BC.p.pitch_buf, p0, p1, len, C);
}

/* find_best_pitch (pitch.c:45): a running-energy recursion with a clamp -> lane 0 */
WV_DEV void find_best_pitch_l0(const WV_LDS i32 *xcorr, const WV_LDS i16 *y, int len, int max_pitch, int *best_pitch, int yshift, i32 maxcorr)
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
            } else { best_num[1] = num; best_den[1] = Syy; best_pitch[1] = i; }
         }
      }
      Syy += (mult16_16(y[i + len], y[i + len]) >> yshift) - (mult16_16(y[i], y[i]) >> yshift);
      Syy = imax(1, Syy);
   }
}

/* find_best_pitch on the wave.  The two things the serial form chains are separable: the running window energy is Syy0 + a prefix sum of (entering - leaving)
 * squares -- exact integer adds, so a wave scan gives every lag its Syy at once (the reference's clamp max(1, .) only acts on an all-zero window: detected, and then
 * the serial form runs instead) -- and the best-two selection, whose cross-multiplied comparisons round and are therefore order-dependent, is replayed in lag order on
 * wave-uniform values fetched with v_readlane: no LDS traffic, the compares run on the scalar unit.  Lane l owns lags l * per .. l * per + per - 1. */
WV_DEV void find_best_pitch_wave(const WV_LDS i32 *xcorr, const WV_LDS i16 *y, int len, int max_pitch, int *best_pitch, int yshift, i32 maxcorr, WV_LDS i32 *hand)
{
   const int lane = wv_lane(), per = (max_pitch + WV_WIDTH - 1) / WV_WIDTH, xshift = celt_ilog2(maxcorr) - 14;
   i32 e0 = 0;
   FOR_LANES(j, len) e0 = add32(e0, mult16_16(y[j], y[j]) >> yshift);
   const i32 Syy0 = add32(1, wv_sum(e0));
   i32 S[8], nm[8], acc = 0, low = 1;
#pragma unroll
   for (int t = 0; t < 8; t++) {
      const int i = lane * per + t;
      S[t] = acc; nm[t] = -1;
      if (t < per && i < max_pitch) {
         acc += (mult16_16(y[i + len], y[i + len]) >> yshift) - (mult16_16(y[i], y[i]) >> yshift);
         const i32 xc = xcorr[i];
         if (xc > 0) { const i16 x16 = extract16(vshr32(xc, xshift)); nm[t] = (i16)mult16_16_q15(x16, x16); }
      }
   }
   const i32 base = Syy0 + wv_scan_incl(acc) - acc;                /* window energy in front of this lane's first lag */
#pragma unroll
   for (int t = 0; t < 8; t++) S[t] += base;
   /* the clamp would have acted iff some window energy (after lag i's update) falls below 1 */
#pragma unroll
   for (int t = 0; t < 8; t++) {
      const int i = lane * per + t;
      if (t < per && i < max_pitch) { const i32 nxt = S[t] + ((mult16_16(y[i + len], y[i + len]) >> yshift) - (mult16_16(y[i], y[i]) >> yshift)); low = imin(low, nxt); }
   }
   if (wv_min(low) < 1) {                                            /* digital silence inside the window: the clamped recurrence, as written in the reference */
      LANE0 { int bp[2]; find_best_pitch_l0(xcorr, y, len, max_pitch, bp, yshift, maxcorr); hand[0] = bp[0]; hand[1] = bp[1]; }
      best_pitch[0] = wv_uni(hand[0]); best_pitch[1] = wv_uni(hand[1]);
      wv_sync();
      return;
   }
   i32 num0 = -1, num1 = -1, den0 = 0, den1 = 0; int p0 = 0, p1 = 1;
   i32 anyc = nm[0];
#pragma unroll
   for (int t = 1; t < 8; t++) anyc = imax(anyc, nm[t]);
   for (u64 todo = wv_ballot(anyc >= 0); todo; todo &= todo - 1) {   /* lanes that hold a candidate, in lag order (the fine search has ten of them among 489 lags) */
      const int l = (int)__builtin_ctzll(todo);
#pragma unroll
      for (int t = 0; t < 8; t++) {
         if (t < per) {                                               /* (lags past max_pitch carry nm = -1) */
            const i32 n = wv_bcast(nm[t], l);
            if (n >= 0) {
               const i32 sy = wv_bcast(S[t], l);
               if (mult16_32_q15((i16)n, den1) > mult16_32_q15((i16)num1, sy)) {
                  if (mult16_32_q15((i16)n, den0) > mult16_32_q15((i16)num0, sy)) { num1 = num0; den1 = den0; p1 = p0; num0 = n; den0 = sy; p0 = l * per + t; }
                  else { num1 = n; den1 = sy; p1 = l * per + t; }
               }
            }
         }
      }
   }
   best_pitch[0] = p0; best_pitch[1] = p1;
}

/* pitch_search (pitch.c:307); returns the pitch lag in every lane */
WV_DEVN int pitch_search_bufs(const WV_LDS i16 *x_lp, const WV_LDS i16 *y, WV_LDS i16 *x_lp4, WV_LDS i16 *y_lp4, WV_LDS i32 *xcorr, WV_LDS i32 *hand, int len, int max_pitch)
{
   const int lag = len + max_pitch;
   i32 xmax = 0, ymax = 0;
   FOR_LANES(j, len >> 2) { i16 v = x_lp[2 * j]; x_lp4[j] = v; xmax = imax(xmax, iabs((i32)v)); }
   FOR_LANES(j, lag >> 2) { i16 v = y[2 * j]; y_lp4[j] = v; ymax = imax(ymax, iabs((i32)v)); }
   xmax = wv_max(xmax); ymax = wv_max(ymax);
   int shift = celt_ilog2(imax(1, imax(xmax, ymax))) - 14 + celt_ilog2(len) / 2;
   wv_sync();
   if (shift > 0) {
      FOR_LANES(j, len >> 2) x_lp4[j] = x_lp4[j] >> shift;
      FOR_LANES(j, lag >> 2) y_lp4[j] = y_lp4[j] >> shift;
      shift *= 2;
      wv_sync();
   } else shift = 0;
   /* coarse search, 4x decimated: four consecutive lags per lane.  The lane's window of y slides through registers (two samples per LDS word, one new word per two taps),
    * x is a broadcast read: a quarter of the LDS reads of a lag-per-lane loop and four independent accumulators (mod-2^32 sums: any order) */
   i32 maxcorr = 1;
   {
      const int nl = max_pitch >> 2, n4 = len >> 2, i0 = 4 * wv_lane();
      if (i0 < nl) {
         const WV_LDS u32 *xp = (const WV_LDS u32 *)x_lp4, *yp = (const WV_LDS u32 *)(y_lp4 + i0);
         i32 s0 = 0, s1 = 0, s2 = 0, s3 = 0;
         const u32 q0 = yp[0], q1 = yp[1];
         i32 w0 = (i16)q0, w1 = (i32)q0 >> 16, w2 = (i16)q1, w3 = (i32)q1 >> 16;
         for (int j = 0; j + 1 < n4; j += 2) {
            const u32 xx = xp[j >> 1], q2 = yp[(j >> 1) + 2];
            const i32 xa = (i16)xx, xb = (i32)xx >> 16, w4 = (i16)q2, w5 = (i32)q2 >> 16;
            s0 = mac16_16(mac16_16(s0, xa, w0), xb, w1);
            s1 = mac16_16(mac16_16(s1, xa, w1), xb, w2);
            s2 = mac16_16(mac16_16(s2, xa, w2), xb, w3);
            s3 = mac16_16(mac16_16(s3, xa, w3), xb, w4);
            w0 = w2; w1 = w3; w2 = w4; w3 = w5;
         }
         if (n4 & 1) { const i32 xa = x_lp4[n4 - 1]; s0 = mac16_16(s0, xa, w0); s1 = mac16_16(s1, xa, w1); s2 = mac16_16(s2, xa, w2); s3 = mac16_16(s3, xa, w3); }
         xcorr[i0] = s0; maxcorr = imax(maxcorr, s0);
         if (i0 + 1 < nl) { xcorr[i0 + 1] = s1; maxcorr = imax(maxcorr, s1); }
         if (i0 + 2 < nl) { xcorr[i0 + 2] = s2; maxcorr = imax(maxcorr, s2); }
         if (i0 + 3 < nl) { xcorr[i0 + 3] = s3; maxcorr = imax(maxcorr, s3); }
      }
   }
   maxcorr = wv_max(maxcorr);
   wv_sync();
   int bpa[2];
   find_best_pitch_wave(xcorr, y_lp4, len >> 2, max_pitch >> 2, bpa, 0, maxcorr, hand);
   const int bp0 = bpa[0], bp1 = bpa[1];
   wv_sync();
   /* finer search, 2x decimated, around the two candidates */
   maxcorr = 1;
   FOR_LANES(i, max_pitch >> 1) xcorr[i] = 0;
   wv_sync();
   for (int r = 0; r < 2; r++) {                                     /* the five lags around each candidate side by side: one read of x per five products */
      const int c2 = 2 * (r ? bp1 : bp0), lo = imax(0, c2 - 2), hi = imin((max_pitch >> 1) - 1, c2 + 2);
      i32 s[5] = {0, 0, 0, 0, 0};
      FOR_LANES(j, len >> 1) {
         const i32 xv = x_lp[j];
#pragma unroll
         for (int d = 0; d < 5; d++) s[d] += mult16_16(xv, y[lo + d + j]) >> shift;       /* (lags past hi read on inside the buffer; their sums are dropped) */
      }
#pragma unroll
      for (int d = 0; d < 5; d++) {
         const i32 sd = wv_sum(s[d]);
         if (lo + d <= hi) { LANE0 xcorr[lo + d] = imax(-1, sd); maxcorr = imax(maxcorr, sd); }
      }
   }
   wv_sync();
   int bpb[2];
   find_best_pitch_wave(xcorr, y, len >> 1, max_pitch >> 1, bpb, shift + 1, maxcorr, hand);
   LANE0 {
      const int bp[2] = {bpb[0], bpb[1]};
      int offset = 0;
      if (bp[0] > 0 && bp[0] < (max_pitch >> 1) - 1) {
         i32 a = xcorr[bp[0] - 1], b = xcorr[bp[0]], c = xcorr[bp[0] + 1];
         if ((c - a) > mult16_32_q15(QC16(.7f, 15), b - a)) offset = 1;
         else if ((a - c) > mult16_32_q15(QC16(.7f, 15), b - c)) offset = -1;
      }
      hand[0] = 2 * bp[0] - offset;
   }
   wv_sync();
   int pitch = hand[0];
   wv_sync();
   return pitch;
}

WV_DEV int pitch_search_wave(WV_LDS FrameLds *L, int len, int max_pitch)
{
   return pitch_search_bufs(L->BC.p.pitch_buf + (OA_MAX_PERIOD >> 1), L->BC.p.pitch_buf, L->BC.p.x_lp4, L->BC.p.y_lp4, L->BC.p.u.xcorr, L->sh.r, len, max_pitch);
}

WV_DEV i16 pitch_gain_fx(i32 xy, i32 xx, i32 yy)
{
   if (xy == 0 || xx == 0 || yy == 0) return 0;
   int sx = celt_ilog2(xx) - 14, sy = celt_ilog2(yy) - 14, shift = sx + sy;
   i32 x2y2 = mult16_16(vshr32(xx, sx), vshr32(yy, sy)) >> 14;
   if (shift & 1) {
      if (x2y2 < 32768) { x2y2 <<= 1; shift--; }
      else { x2y2 >>= 1; shift++; }
   }
   i16 den = fx_rsqrt_norm(x2y2);
   i32 g = mult16_32_q15(den, xy);
   g = vshr32(g, (shift >> 1) - 1);
   return extract16(imax(-Q15ONE, imin(g, Q15ONE)));
}

/* remove_doubling (pitch.c:454); all lanes return the same (gain, *T0_) */
WV_DEVN i16 remove_doubling_wave(WV_LDS FrameLds *L, int maxperiod, int minperiod, int N, int *T0_, int prev_period, i16 prev_gain)
{
   const int second_check[16] = {0, 0, 3, 2, 3, 2, 5, 2, 3, 2, 3, 2, 5, 2, 3, 2};
   int T, T0, offset, minperiod0 = minperiod;
   i16 g, g0, pg;
   i32 xy, xx, yy, xy2, best_xy, best_yy;
   WV_LDS i32 *yy_lookup = L->BC.p.u.yy_lookup;
   maxperiod /= 2; minperiod /= 2; *T0_ /= 2; prev_period /= 2; N /= 2;
   const WV_LDS i16 *x = L->BC.p.pitch_buf + maxperiod;
   if (*T0_ >= maxperiod) *T0_ = maxperiod - 1;
   T = T0 = *T0_;
   wave_dual_inner16(x, x, x - T0, N, &xx, &xy);
   {  /* yy_lookup[i] = max(0, xx + sum_{j<=i} (x[-j]^2 - x[N-j]^2)) : chunked prefix sum, 8 lags per lane */
      i32 loc[8], s = 0;
      int base = 1 + 8 * wv_lane();
      for (int t = 0; t < 8; t++) {
         int j = base + t;
         if (j <= maxperiod) s = s + mult16_16(x[-j], x[-j]) - mult16_16(x[N - j], x[N - j]);
         loc[t] = s;
      }
      i32 excl = wv_scan_incl(s) - s;
      for (int t = 0; t < 8; t++) { int j = base + t; if (j <= maxperiod) yy_lookup[j] = imax(0, xx + excl + loc[t]); }
      LANE0 yy_lookup[0] = xx;
   }
   wv_sync();
   yy = yy_lookup[T0];
   best_xy = xy; best_yy = yy;
   g = g0 = pitch_gain_fx(xy, xx, yy);
   for (int k = 2; k <= 15; k++) {
      int T1, T1b;
      i16 g1, cont, thresh;
      T1 = (u32)(2 * T0 + k) / (u32)(2 * k);
      if (T1 < minperiod) break;
      if (k == 2) T1b = (T1 + T0 > maxperiod) ? T0 : T0 + T1;
      else T1b = (u32)(2 * second_check[k] * T0 + k) / (u32)(2 * k);
      wave_dual_inner16(x, x - T1, x - T1b, N, &xy, &xy2);
      xy = half32(xy + xy2);
      yy = half32(yy_lookup[T1] + yy_lookup[T1b]);
      g1 = pitch_gain_fx(xy, xx, yy);
      if (iabs(T1 - prev_period) <= 1) cont = prev_gain;
      else if (iabs(T1 - prev_period) <= 2 && 5 * k * k < T0) cont = prev_gain >> 1;
      else cont = 0;
      thresh = (i16)imax(QC16(.3f, 15), mult16_16_q15(QC16(.7f, 15), g0) - cont);
      if (T1 < 3 * minperiod) thresh = (i16)imax(QC16(.4f, 15), mult16_16_q15(QC16(.85f, 15), g0) - cont);
      else if (T1 < 2 * minperiod) thresh = (i16)imax(QC16(.5f, 15), mult16_16_q15(QC16(.9f, 15), g0) - cont);
      if (g1 > thresh) { best_xy = xy; best_yy = yy; T = T1; g = g1; }
   }
   if (T < minperiod * 2) {
      int T1 = T * 5 / 8, T2 = T * 6 / 8;
      wave_dual_inner16(x, x - T1, x - T2, N, &xy, &xy2);
      i16 g1 = pitch_gain_fx(xy, xx, yy_lookup[T1]), g2 = pitch_gain_fx(xy2, xx, yy_lookup[T2]);
      if (g1 >= g || g2 >= g) g = 0;
   }
   best_xy = imax(0, best_xy);
   if (best_yy <= best_xy) pg = Q15ONE;
   else pg = (i16)(fx_frac_div32(best_xy, best_yy + 1) >> 16);
   i32 xc[3];
   for (int k = 0; k < 3; k++) xc[k] = wave_inner16(x, x - (T + k - 1), N);
   if ((xc[2] - xc[0]) > mult16_32_q15(QC16(.7f, 15), xc[1] - xc[0])) offset = 1;
   else if ((xc[0] - xc[2]) > mult16_32_q15(QC16(.7f, 15), xc[1] - xc[2])) offset = -1;
   else offset = 0;
   if (pg > g) pg = g;
   *T0_ = 2 * T + offset;
   if (*T0_ < minperiod0) *T0_ = minperiod0;
   return pg;
}

/* comb_filter (celt.c:238) out of place: y[i] from the *unfiltered* signal (PreSrc, index 0 = first new sample), so all
 * outputs are independent.  sums (or NULL): [0] += sum |x[i] >> 12| of the input, [1] += sum |y[i] >> 12| of the output -- run_prefilter's before / after measures
 * (celt_encoder.c:1520-1540) taken while the samples are in registers. */
WV_DEV void comb_filter_wave(i32 *y, const PreSrc &p, int T0, int T1, int N, i16 g0, i16 g1, int tapset0, int tapset1, int overlap, i32 *sums = nullptr)
{
   const i16 gains[3][3] = {
      {QC16(0.3066406250f, 15), QC16(0.2170410156f, 15), QC16(0.1296386719f, 15)},
      {QC16(0.4638671875f, 15), QC16(0.2680664062f, 15), QC16(0.f, 15)},
      {QC16(0.7998046875f, 15), QC16(0.1000976562f, 15), QC16(0.f, 15)}};
#define XA(k) pre_at(p, OA_MAX_PERIOD + (k))
   i32 sb = 0, sa = 0;
   if (g0 == 0 && g1 == 0) {
      for (int i0 = wv_lane(); i0 < N; i0 += 8 * WV_WIDTH) {
         i32 v[8];
#pragma unroll
         for (int u = 0; u < 8; u++) v[u] = XA(imin(i0 + u * WV_WIDTH, N - 1));
#pragma unroll
         for (int u = 0; u < 8; u++) { const int i = i0 + u * WV_WIDTH; if (i < N) { y[i] = v[u]; sb += iabs(v[u] >> 12); } }
      }
      if (sums) { sb = wv_sum(sb); sums[0] += sb; sums[1] += sb; }
      return;
   }
   T0 = imax(T0, OA_MIN_PERIOD);
   T1 = imax(T1, OA_MIN_PERIOD);
   i16 g00 = (i16)mult_coef_taps(g0, gains[tapset0][0]), g01 = (i16)mult_coef_taps(g0, gains[tapset0][1]), g02 = (i16)mult_coef_taps(g0, gains[tapset0][2]);
   i16 g10 = (i16)mult_coef_taps(g1, gains[tapset1][0]), g11 = (i16)mult_coef_taps(g1, gains[tapset1][1]), g12 = (i16)mult_coef_taps(g1, gains[tapset1][2]);
   if (g0 == g1 && T0 == T1 && tapset0 == tapset1) overlap = 0;
   const bool outer_on = g12 != 0;                               /* (tap sets 1 and 2 have no outer taps: g12 == 0 contributes mult_coef_32(0, .) == 0 -- two loads less) */
   for (int i0 = wv_lane(); i0 < N; i0 += 4 * WV_WIDTH) {        /* four trips' taps are asked for together (a trip is a round trip to the scratch), then the arithmetic, then the stores */
      i32 x0[4], c0[4], c1[4], c2[4];
#pragma unroll
      for (int u = 0; u < 4; u++) {
         const int i = imin(i0 + u * WV_WIDTH, N - 1);            /* (the clamped sample of a ragged last batch is loaded again and not stored) */
         x0[u] = XA(i); c0[u] = XA(i - T1); c1[u] = add32(XA(i - T1 + 1), XA(i - T1 - 1));
         c2[u] = outer_on ? add32(XA(i - T1 + 2), XA(i - T1 - 2)) : 0;
      }
#pragma unroll
      for (int u = 0; u < 4; u++) {
         const int i = i0 + u * WV_WIDTH;
         if (i < N) {
            i32 v = x0[u];
            if (i < overlap) {
               i16 f = (i16)mult_coef(ct_window[i], ct_window[i]);
               v = add32(v, mult_coef_32(mult_coef((Q15ONE - f), g00), XA(i - T0)));
               v = add32(v, mult_coef_32(mult_coef((Q15ONE - f), g01), add32(XA(i - T0 + 1), XA(i - T0 - 1))));
               v = add32(v, mult_coef_32(mult_coef((Q15ONE - f), g02), add32(XA(i - T0 + 2), XA(i - T0 - 2))));
               v = add32(v, mult_coef_32(mult_coef(f, g10), c0[u]));
               v = add32(v, mult_coef_32(mult_coef(f, g11), c1[u]));
               v = add32(v, mult_coef_32(mult_coef(f, g12), c2[u]));
               v = saturate(sub32(v, 3), SIG_SAT);
            } else if (g1 != 0) {
               const i32 outer = outer_on ? mult_coef_32(g12, c2[u]) : 0;
               v = add32(add32(add32(v, mult_coef_32(g10, c0[u])), mult_coef_32(g11, c1[u])), outer);
               v = saturate(sub32(v, 1), SIG_SAT);
            }
            y[i] = v;
            sb += iabs(x0[u] >> 12); sa += iabs(v >> 12);
         }
      }
   }
   if (sums) { sums[0] += wv_sum(sb); sums[1] += wv_sum(sa); }
#undef XA
}

/* run_prefilter (celt_encoder.c:1405).  Writes the comb-filtered new input to g->in[c][0..N) (HBM scratch) (its 120-sample head stays
 * in_mem in HBM until the last MDCT of the frame has consumed it: store_in_mem_wave); rewrites prefilter_mem in HBM. */
WV_DEVN void run_prefilter_wave(WV_LDS FrameLds *L, OaEncState *gst, const PreSrc &ps0, const PreSrc &ps1, int enabled)
{
   WV_LDS FrameShared *sh = &L->sh;
   WV_LDS OaEncScalars *st = &L->st;
   const int CC = sh->CC, N = sh->N, overlap = OA_OVERLAP, max_period = OA_MAX_PERIOD, min_period = OA_MIN_PERIOD;
   const int prefilter_tapset = st->tapset_decision;
   CeltScratch *G = L->g;
   int pitch_index, pf_on, qg;
   i16 gain1, pf_threshold;
   i16 tone_freq = (i16)sh->tone_freq;
   if (enabled && sh->toneishness > QC32(.99f, 29)) {
      int multiple = 1;
      if (tone_freq >= QC16(3.1416f, 13)) tone_freq = (i16)(QC16(3.141593f, 13) - tone_freq);
      while (tone_freq >= multiple * QC16(0.39f, 13)) multiple++;
      if (tone_freq > QC16(0.006148f, 13)) pitch_index = imin((51472 * multiple + tone_freq / 2) / tone_freq, OA_MAX_PERIOD - 2);
      else pitch_index = OA_MIN_PERIOD;
      gain1 = QC16(.75f, 15);
   } else if (enabled && sh->complexity >= 5) {
      K_TIC();
      pitch_downsample_wave(L, ps0, ps1, (max_period + N) >> 1, CC);
      K_TOC(25);
      pitch_index = pitch_search_wave(L, N, max_period - 3 * min_period);
      K_TOC(26);
      pitch_index = max_period - pitch_index;
      gain1 = remove_doubling_wave(L, max_period, min_period, N, &pitch_index, st->prefilter_period, (i16)st->prefilter_gain);
      K_TOC(27);
      if (pitch_index > max_period - 2) pitch_index = max_period - 2;
      gain1 = (i16)mult16_16_q15(QC16(.7f, 15), gain1);
      if (sh->loss_rate > 2) gain1 = (i16)(gain1 >> 1);
      if (sh->loss_rate > 4) gain1 = (i16)(gain1 >> 1);
      if (sh->loss_rate > 8) gain1 = 0;
   } else { gain1 = 0; pitch_index = OA_MIN_PERIOD; }
   if (gst->analysis.valid) gain1 = (i16)an_scale_pitch_gain(gain1, &gst->analysis);               /* :1494: less pre-filtering the more energy lies above the pitch range */
   const i16 old_gain = (i16)st->prefilter_gain;
   pf_threshold = QC16(.2f, 15);
   if (iabs(pitch_index - st->prefilter_period) * 10 > pitch_index) {
      pf_threshold += QC16(.2f, 15);
      if ((i16)sh->tf_estimate > QC16(.98f, 14)) gain1 = 0;
   }
   if (sh->nbAvailableBytes < 25) pf_threshold += QC16(.1f, 15);
   if (sh->nbAvailableBytes < 35) pf_threshold += QC16(.1f, 15);
   if (old_gain > QC16(.4f, 15)) pf_threshold -= QC16(.1f, 15);
   if (old_gain > QC16(.55f, 15)) pf_threshold -= QC16(.1f, 15);
   pf_threshold = (i16)imax(pf_threshold, QC16(.2f, 15));
   if (gain1 < pf_threshold) { gain1 = 0; pf_on = 0; qg = 0; }
   else {
      if (iabs(gain1 - old_gain) < QC16(.1f, 15)) gain1 = old_gain;
      qg = ((gain1 + 1536) >> 10) / 3 - 1;
      qg = imax(0, imin(7, qg));
      gain1 = (i16)(QC16(0.09375f, 15) * (qg + 1));
      pf_on = 1;
   }
   const int old_period = imax(st->prefilter_period, OA_MIN_PERIOD), old_tapset = st->prefilter_tapset;
   i32 before[2] = {0, 0}, after[2] = {0, 0};
   K_TIC();
   wv_sync();                    /* the pitch buffers (aliased with in[]) are dead from here */
   /* (the head in[c][0..overlap) is in_mem, last frame's *filtered* tail, celt_encoder.c:1546: it stays in HBM) */
   for (int c = 0; c < CC; c++) {
      i32 sums[2] = {0, 0};
      comb_filter_wave(G->in[c], c ? ps1 : ps0, old_period, pitch_index, N, (i16)-old_gain, (i16)-gain1, old_tapset, prefilter_tapset, overlap, sums);
      before[c] = sums[0]; after[c] = sums[1];
   }
   int cancel_pitch = 0;
   if (CC == 2) {
      i16 thresh[2];
      thresh[0] = (i16)(mult16_32_q15(mult16_16_q15(QC16(.25f, 15), gain1), before[0]) + mult16_32_q15(QC16(.01f, 15), before[1]));
      thresh[1] = (i16)(mult16_32_q15(mult16_16_q15(QC16(.25f, 15), gain1), before[1]) + mult16_32_q15(QC16(.01f, 15), before[0]));
      if (after[0] - before[0] > thresh[0] || after[1] - before[1] > thresh[1]) cancel_pitch = 1;
      if (before[0] - after[0] < thresh[0] && before[1] - after[1] < thresh[1]) cancel_pitch = 1;
   } else if (after[0] > before[0]) cancel_pitch = 1;
   if (cancel_pitch) {
      wv_sync();
      for (int c = 0; c < CC; c++) {
         const PreSrc &ps = c ? ps1 : ps0;
         for (int i0 = wv_lane(); i0 < N; i0 += 8 * WV_WIDTH) {
            i32 v[8];
#pragma unroll
            for (int u = 0; u < 8; u++) v[u] = pre_at(ps, max_period + imin(i0 + u * WV_WIDTH, N - 1));
#pragma unroll
            for (int u = 0; u < 8; u++) { const int i = i0 + u * WV_WIDTH; if (i < N) G->in[c][i] = v[u]; }
         }
      }
      wv_sync();
      for (int c = 0; c < CC; c++)
         comb_filter_wave(G->in[c], c ? ps1 : ps0, old_period, pitch_index, overlap, (i16)-old_gain, 0, old_tapset, prefilter_tapset, overlap);
      gain1 = 0; pf_on = 0; qg = 0;
   }
   wv_sync();
   K_TOC(28);
   /* persistent history: unfiltered [history | new][N .. N+1024) -> prefilter_mem.
    * Every lane first gathers its 16 values (the shift may overlap source and destination), then stores. */
   for (int c = 0; c < CC; c++) {
      const PreSrc &ps = c ? ps1 : ps0;
      i32 keep[OA_MAX_PERIOD / WV_WIDTH];
      for (int t = 0; t < OA_MAX_PERIOD / WV_WIDTH; t++) keep[t] = pre_at(ps, N + wv_lane() + t * WV_WIDTH);
      wv_sync();
      for (int t = 0; t < OA_MAX_PERIOD / WV_WIDTH; t++) gst->prefilter_mem[c * OA_MAX_PERIOD + wv_lane() + t * WV_WIDTH] = keep[t];
      wv_sync();
   }
   K_TOC(29);
   LANE0 {
      st->prefilter_period = old_period;
      sh->pf_on = pf_on; sh->pitch_index = pitch_index; sh->gain1 = gain1; sh->qg = qg; sh->prefilter_tapset = prefilter_tapset;
   }
   wv_sync();
}

/* in_mem <- the filtered tail in[c][N-overlap .. N) (celt_encoder.c:1556); called once the last MDCT of the frame has
 * consumed the old head */
WV_DEV void store_in_mem_wave(WV_LDS FrameLds *L, OaEncState *gst)
{
   const int CC = L->sh.CC, N = L->sh.N, overlap = OA_OVERLAP;
   const CeltScratch *G = L->g;
   for (int c = 0; c < CC; c++) { FOR_LANES(i, overlap) gst->in_mem[c * overlap + i] = G->in[c][N - overlap + i]; }
   wv_sync();
}
#endif

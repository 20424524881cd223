/* celt_enc_pvq.h — PVQ band quantisation on one wavefront (encoder side, with the resynthesis the stereo theta-RDO needs).
 * Reference: celt/bands.c (:379 intensity_stereo, :405 stereo_split, :418 stereo_merge, :574/:600 hadamard, :623 haar1,
 * :638 compute_qn, :700 compute_theta, :930 quant_band_n1, :973 quant_partition, :1248 quant_band, :1387 quant_band_stereo,
 * :1589 quant_all_bands), celt/vq.c (:75 exp_rotation1, :104 exp_rotation, :150 normalise_residual, :183 extract_collapse_mask,
 * :205 op_pvq_search_c, :552 alg_quant, :695 renormalise_vector, :724 stereo_itheta), celt/cwrs.c:444 icwrs.
 *
 * Control flow (budget-driven, band after band) is executed uniformly by all 64 lanes; range-coder calls run on lane 0;
 * vector work over the N<=176 coefficients of a band is lane-strided with exact wave reductions:
 *   - greedy pulse search: every lane scores its coefficients, one cross-lane arg-max per pulse (exact 16x16 cross products,
 *     lowest index wins ties = the reference's scan order);
 *   - codeword index (icwrs): suffix sums of |y| by a wave scan, then all U(N-j,k) lookups in parallel, summed mod 2^32. */
#ifndef OPUS_AMD_CELT_ENC_PVQ_H
#define OPUS_AMD_CELT_ENC_PVQ_H
#ifndef K_DUMP
#define K_DUMP(tag, ptr, nbytes)
#define K_DUMPI(tag, v)
#endif
#define LOG_MAX_PSEUDO 6
#ifndef OA_QUANT_BAND_FN            /* inlined too (+0.9 %, -14 % traffic, profiles/r02_o); -DOA_QUANT_BAND_FN=WV_DEVN puts it back out of line */
#define OA_QUANT_BAND_FN WV_DEV
#endif
#ifndef OA_ALG_QUANT_FN             /* (A/B experiments: -DOA_ALG_QUANT_FN=WV_DEV inlines the 6 k-instruction quantiser into every partition level) */
#define OA_ALG_QUANT_FN WV_DEVN
#endif
#ifndef OA_PVQ_STEREO_FN            /* inlined into its three call sites: out of line (-DOA_PVQ_STEREO_FN=WV_DEVN) it saves and restores 13 VGPRs per call, 27 % of the frame's HBM traffic (profiles/r02_l) */
#define OA_PVQ_STEREO_FN WV_DEV
#endif
#ifndef K_TIC
#define K_TIC()
#define K_TOC(bucket)
#endif

struct BandCfg { int resynth, i, intensity, spread, tf_change, theta_round, disable_inv, avoid_split_noise; };


WV_DEV u32 lcg_rand(u32 seed) { return 1664525u * seed + 1013904223u; }
WV_DEV int bitexact_cos(int x_)
{
   i16 x = (i16)x_;
   i32 tmp = (4096 + ((i32)x * x)) >> 13;
   i16 x2 = (i16)tmp;
   x2 = (i16)((32767 - x2) + frac_mul16(x2, (-7651 + frac_mul16(x2, (8277 + frac_mul16(-626, x2))))));
   return (i16)(1 + x2);
}
WV_DEV int bitexact_log2tan(int isin, int icos)
{
   int lc = ec_ilog(icos), ls = ec_ilog(isin);
   icos <<= 15 - lc;
   isin <<= 15 - ls;
   return (ls - lc) * (1 << 11) + frac_mul16(isin, frac_mul16(isin, -2597) + 7932) - frac_mul16(icos, frac_mul16(icos, -2597) + 7932);
}
WV_DEV u32 pvq_u(int n, int k)
{
   int lo = n < k ? n : k, hi = n < k ? k : n;
   return ct_pvq_u_flat[lo * 177 + hi];
}

WV_DEV void haar1_wave(WV_LDS i32 *X, int N0, int stride)
{
   N0 >>= 1;
   FOR_LANES(p, N0 * stride) {
      int i = p / N0, j = p - i * N0;
      i32 t1 = mult32_32_q31(QC32(.70710678f, 31), X[stride * 2 * j + i]);
      i32 t2 = mult32_32_q31(QC32(.70710678f, 31), X[stride * (2 * j + 1) + i]);
      X[stride * 2 * j + i] = add32(t1, t2);
      X[stride * (2 * j + 1) + i] = sub32(t1, t2);
   }
   wv_sync();
}
WV_TABLE int k_ordery_table[30] = {1, 0, 3, 0, 2, 1, 7, 0, 4, 3, 6, 1, 5, 2, 15, 0, 8, 7, 12, 3, 11, 4, 14, 1, 9, 6, 13, 2, 10, 5};
/* (de)interleave_hadamard (bands.c:574/:600): a permutation of N0*stride <= 176 words, done in place through registers
 * (every lane gathers its <= 3 elements, barrier, scatter) */
WV_DEV void deinterleave_hadamard_wave(WV_LDS i32 *X, int N0, int stride, int hadamard)
{
   const int N = N0 * stride;
   i32 v[3]; int dst[3];
   for (int t = 0; t < 3; t++) {
      int p = wv_lane() + t * WV_WIDTH;
      if (p < N) {
         int i = p / N0, j = p - i * N0;
         dst[t] = hadamard ? k_ordery_table[stride - 2 + i] * N0 + j : i * N0 + j;
         v[t] = X[j * stride + i];
      }
   }
   wv_sync();
   for (int t = 0; t < 3; t++) { int p = wv_lane() + t * WV_WIDTH; if (p < N) X[dst[t]] = v[t]; }
   wv_sync();
}
WV_DEV void interleave_hadamard_wave(WV_LDS i32 *X, int N0, int stride, int hadamard)
{
   const int N = N0 * stride;
   i32 v[3]; int dst[3];
   for (int t = 0; t < 3; t++) {
      int p = wv_lane() + t * WV_WIDTH;
      if (p < N) {
         int i = p / N0, j = p - i * N0;
         int src = hadamard ? k_ordery_table[stride - 2 + i] * N0 + j : i * N0 + j;
         dst[t] = j * stride + i;
         v[t] = X[src];
      }
   }
   wv_sync();
   for (int t = 0; t < 3; t++) { int p = wv_lane() + t * WV_WIDTH; if (p < N) X[dst[t]] = v[t]; }
   wv_sync();
}
WV_DEV int compute_qn(int N, int b, int offset, int pulse_cap, int stereo)
{
   const i16 exp2_table8[8] = {16384, 17866, 19483, 21247, 23170, 25267, 27554, 30048};
   int qn, qb, N2 = 2 * N - 1;
   if (stereo && N == 2) N2--;
   qb = fx_sdiv24(b + N2 * offset, N2);
   qb = imin(b - pulse_cap - (4 << BITRES), qb);
   qb = imin(8 << BITRES, qb);
   if (qb < (1 << BITRES >> 1)) qn = 1;
   else {
      qn = exp2_table8[qb & 0x7] >> (14 - (qb >> BITRES));
      qn = (qn + 1) >> 1 << 1;
   }
   return qn;
}
WV_DEV void compute_channel_weights(i32 Ex, i32 Ey, i16 w[2])
{
   i32 minE = imin(Ex, Ey);
   Ex = add32(Ex, minE / 3);
   Ey = add32(Ey, minE / 3);
   int shift = celt_ilog2(EPSILON + imax(Ex, Ey)) - 14;
   w[0] = (i16)vshr32(Ex, shift);
   w[1] = (i16)vshr32(Ey, shift);
}
WV_DEV void intensity_stereo_wave(WV_LDS FrameLds *L, WV_LDS i32 *X, const WV_LDS i32 *Y, int bandID, int N)
{
   const WV_LDS i32 *bandE = L->bandE;
   int i = bandID;
   int shift = celt_zlog2(imax(bandE[i], bandE[i + NBE])) - 13;
   i16 left = (i16)vshr32(bandE[i], shift), right = (i16)vshr32(bandE[i + NBE], shift);
   i16 norm = (i16)(EPSILON + fx_sqrt(EPSILON + mult16_16(left, left) + mult16_16(right, right)));
   left = (i16)imin(left, norm - 1);
   right = (i16)imin(right, norm - 1);
   i16 a1 = (i16)(shl32((i32)left, 15) / norm), a2 = (i16)(shl32((i32)right, 15) / norm);
   FOR_LANES(j, N) X[j] = add32(mult16_32_q15(a1, X[j]), mult16_32_q15(a2, Y[j]));
   wv_sync();
}
WV_DEV void stereo_split_wave(WV_LDS i32 *X, WV_LDS i32 *Y, int N)
{
   FOR_LANES(j, N) {
      i32 l = mult32_32_q31(QC32(.70710678f, 31), X[j]);
      i32 r = mult32_32_q31(QC32(.70710678f, 31), Y[j]);
      X[j] = add32(l, r);
      Y[j] = sub32(r, l);
   }
   wv_sync();
}
WV_DEV void stereo_merge_wave(WV_LDS i32 *X, WV_LDS i32 *Y, i32 mid, int N)
{
   i32 xp = inner_prod_norm_shift_w(Y, X, N), side = inner_prod_norm_shift_w(Y, Y, N);
   xp = mult32_32_q31(mid, xp);
   i32 El = (mult32_32_q31(mid, mid) >> 3) + side - 2 * xp;
   i32 Er = (mult32_32_q31(mid, mid) >> 3) + side + 2 * xp;
   if (Er < QC32(6e-4f, 28) || El < QC32(6e-4f, 28)) { FOR_LANES(j, N) Y[j] = X[j]; wv_sync(); return; }
   int kl = celt_ilog2(El) >> 1, kr = celt_ilog2(Er) >> 1;
   i32 t = vshr32(El, (kl << 1) - 29);
   i32 lgain = fx_rsqrt_norm32(t);
   t = vshr32(Er, (kr << 1) - 29);
   i32 rgain = fx_rsqrt_norm32(t);
   if (kl < 7) kl = 7;
   if (kr < 7) kr = 7;
   FOR_LANES(j, N) {
      i32 l = mult32_32_q31(mid, X[j]), r = Y[j];
      X[j] = vshr32(mult32_32_q31(lgain, sub32(l, r)), kl - 15);
      Y[j] = vshr32(mult32_32_q31(rgain, add32(l, r)), kr - 15);
   }
   wv_sync();
}
WV_DEV i32 stereo_itheta_wave(const WV_LDS i32 *X, const WV_LDS i32 *Y, int stereo, int N)
{
   i32 Emid = 0, Eside = 0;
   if (stereo) {
      FOR_LANES(i, N) {
         i32 m = pshr32(add32(X[i], Y[i]), NORM_SHIFT - 13);
         i32 s = pshr32(sub32(X[i], Y[i]), NORM_SHIFT - 13);
         Emid = mac16_16(Emid, m, m);
         Eside = mac16_16(Eside, s, s);
      }
      Emid = wv_sum(Emid); Eside = wv_sum(Eside);
   } else {
      Emid = inner_prod_norm_shift_w(X, X, N);
      Eside = inner_prod_norm_shift_w(Y, Y, N);
   }
   i32 mid = fx_sqrt32(Emid), side = fx_sqrt32(Eside);
   return fx_atan2p_norm(side, mid);
}

/* ---- register-resident band vector: element e lives in register e>>6, lane e&63 (NR = 1 for N <= 64, else 3) ---- */
template <int NR> WV_DEV i32 rv_get(const i32 (&v)[NR], int e)                 /* e uniform -> scalar result */
{
   if (NR == 1 || e < 64) return wv_bcast(v[0], e);
   if (NR == 2 || e < 128) return wv_bcast(v[NR > 1 ? 1 : 0], e - 64);
   return wv_bcast(v[NR > 2 ? 2 : 0], e - 128);
}
template <int NR> WV_DEV void rv_set(i32 (&v)[NR], int e, i32 val)            /* e, val uniform */
{
   if (NR == 1 || e < 64) v[0] = wv_writelane(val, e, v[0]);
   else if (NR == 2 || e < 128) v[NR > 1 ? 1 : 0] = wv_writelane(val, e - 64, v[NR > 1 ? 1 : 0]);
   else v[NR > 2 ? 2 : 0] = wv_writelane(val, e - 128, v[NR > 2 ? 2 : 0]);
}
template <int NR> WV_DEV void rv_shift_down(const i32 (&v)[NR], int d, i32 (&o)[NR])     /* o[e] = v[e + d], 0 < d < 64 */
{
   const int src = wv_lane() + d, wrap = src >= 64;
   for (int t = 0; t < NR; t++) {
      i32 a = wv_shfl(v[t], src & 63), b = t + 1 < NR ? wv_shfl(v[t + 1 < NR ? t + 1 : t], src & 63) : 0;
      o[t] = wrap ? b : a;
   }
}
template <int NR> WV_DEV void rv_shift_up(const i32 (&v)[NR], int d, i32 (&o)[NR])       /* o[e] = v[e - d], 0 < d < 64 */
{
   const int src = wv_lane() - d, wrap = src < 0;
   for (int t = 0; t < NR; t++) {
      i32 a = wv_shfl(v[t], src & 63), b = t > 0 ? wv_shfl(v[t > 0 ? t - 1 : 0], src & 63) : 0;
      o[t] = wrap ? b : a;
   }
}

/* exp_rotation1 (vq.c:75) on every block of the band at once.  Each pass is a recurrence with rounding, so the chain
 * itself is serial -- but only the chain: the products with the not-yet-touched operand and all the "other" outputs are
 * elementwise.  The chain runs on the SCALAR unit (v_readlane -> s_mul/s_add/s_ashr/s_sext -> v_writelane), one chain per
 * (block, residue mod d), leaving the vector ALU to the other wave of the SIMD. */
template <int NR> WV_DEV void rot_pass(i32 (&v)[NR], int nblk, int len, int d, i32 c_, i32 s_)
{
   const i32 c = (i16)c_, s = (i16)s_;
   const int lane = wv_lane();
   int pos[NR];
   for (int t = 0; t < NR; t++) { int e = lane + 64 * t; pos[t] = nblk == 1 ? e : e - (int)((u32)e / (u32)len) * len; }
   if (len - d > 0) {            /* forward: for i in [0, len-d): (X[i], X[i+d]) <- (c X[i] - s X[i+d], c X[i+d] + s X[i]) */
      i32 xs[NR], A[NR], Bv[NR], cv[NR], nv[NR];
      rv_shift_down(v, d, xs);
      for (int t = 0; t < NR; t++) { A[t] = add32(mult16_16(c, xs[t]), 16384); Bv[t] = mult16_16(s, xs[t]); cv[t] = v[t]; nv[t] = v[t]; }
      for (int blk = 0; blk < nblk; blk++) {
         const int base = blk * len;
         for (int r = 0; r < d && r < len - d; r++) {
            i32 x1 = rv_get(v, base + r);
            int i = r;
            for (; i < len - d; i += d) {
               rv_set(cv, base + i, x1);
               x1 = (i32)(i16)(add32(rv_get(A, base + i), s * x1) >> 15);
            }
            rv_set(nv, base + i, x1);
         }
      }
      for (int t = 0; t < NR; t++) {
         i32 o = (i32)(i16)(add32(sub32(mult16_16(c, cv[t]), Bv[t]), 16384) >> 15);
         v[t] = pos[t] < len - d ? o : nv[t];
      }
   }
   if (len - 2 * d - 1 >= 0) {   /* backward: for i = len-2d-1 .. 0, same butterfly */
      i32 Cv[NR], Sv[NR], yv[NR], hv[NR], o[NR], ou[NR];
      for (int t = 0; t < NR; t++) { Cv[t] = add32(mult16_16(c, v[t]), 16384); Sv[t] = mult16_16(s, v[t]); yv[t] = 0; hv[t] = v[t]; }
      const int top = len - 2 * d - 1;
      for (int blk = 0; blk < nblk; blk++) {
         const int base = blk * len;
         for (int i0 = top; i0 > top - d && i0 >= 0; i0--) {
            i32 y = rv_get(v, base + i0 + d);
            int i = i0;
            for (; i >= 0; i -= d) {
               rv_set(yv, base + i, y);
               y = (i32)(i16)(sub32(rv_get(Cv, base + i), s * y) >> 15);
            }
            rv_set(hv, base + i + d, y);
         }
      }
      for (int t = 0; t < NR; t++) o[t] = (i32)(i16)(add32(add32(mult16_16(c, yv[t]), Sv[t]), 16384) >> 15);
      rv_shift_up(o, d, ou);
      for (int t = 0; t < NR; t++) v[t] = (pos[t] >= d && pos[t] <= len - d - 1) ? ou[t] : hv[t];
   }
}
/* The same pass when it falls apart into several independent chains (one per block and residue mod d: 2 .. 40 of them): one LANE per chain.  The band goes
 * through its own LDS slot T (dead between the load and the store of alg_quant), laid out as it is in memory, so the lanes of one step touch consecutive words;
 * each lane runs the reference's two sweeps over its residue class -- the same class in both directions, so nothing is shared between lanes in between.  A pass
 * of `steps` = len / d rounds costs steps x ~12 vector instructions for the whole band instead of 2 vector + 6 scalar instructions per ELEMENT. */
template <int NR> WV_DEV void rot_pass_lds(i32 (&v)[NR], WV_LDS i32 *T, int nblk, int len, int d, i32 c_, i32 s_)
{
   const i32 c = (i16)c_, s = (i16)s_;
   const int lane = wv_lane(), N = nblk * len, top = len - 2 * d - 1;
   wv_sync();
   for (int t = 0; t < NR; t++) if (lane + 64 * t < N) T[lane + 64 * t] = v[t];
   wv_sync();
   for (int ch = lane; ch < nblk * d; ch += WV_WIDTH) {
      const int blk = (int)((u32)ch / (u32)d), r = ch - blk * d;
      WV_LDS i32 *X = T + blk * len;
      if (r < len - d) {                                     /* upwards: (X[i], X[i+d]) <- (c X[i] - s X[i+d], c X[i+d] + s X[i]), i = r, r + d, ... */
         i32 x1 = X[r];
         int i = r;
         for (; i < len - d; i += d) {
            const i32 x2 = X[i + d];
            X[i] = (i32)(i16)(add32(sub32(mult16_16(c, x1), s * x2), 16384) >> 15);
            x1 = (i32)(i16)(add32(add32(mult16_16(c, x2), s * x1), 16384) >> 15);
         }
         X[i] = x1;
      }
      if (top >= r) {                                        /* downwards from the highest i <= top of the same class */
         int i = top - (int)((u32)(top - r) % (u32)d);
         i32 y = X[i + d];
         for (; i >= 0; i -= d) {
            const i32 x1 = X[i];
            X[i + d] = (i32)(i16)(add32(add32(mult16_16(c, y), s * x1), 16384) >> 15);
            y = (i32)(i16)(add32(sub32(mult16_16(c, x1), s * y), 16384) >> 15);
         }
         X[i + d] = y;
      }
   }
   wv_sync();
   for (int t = 0; t < NR; t++) if (lane + 64 * t < N) v[t] = T[lane + 64 * t];
   wv_sync();
}
#ifndef OA_ROT_LANE_CHAINS
#define OA_ROT_LANE_CHAINS 4      /* from this many chains on, a pass runs one lane per chain; below, on the scalar unit */
#endif
template <int NR> WV_DEV void rot_pass_any(i32 (&v)[NR], WV_LDS i32 *T, int nblk, int len, int d, i32 c_, i32 s_)
{
   if (nblk * d >= OA_ROT_LANE_CHAINS) rot_pass_lds(v, T, nblk, len, d, c_, s_); else rot_pass(v, nblk, len, d, c_, s_);
}

/* exp_rotation (vq.c:104) on the register-resident band; T: the band's LDS slot, free as scratch */
template <int NR> WV_DEV void exp_rotation_regs(i32 (&v)[NR], WV_LDS i32 *T, int len, int dir, int stride, int K, int spread)
{
   int stride2 = 0;
   if (2 * K >= len || spread == 0) return;
   int factor = spread == 1 ? 15 : (spread == 2 ? 10 : 5);
   i16 gain = (i16)fx_div(mult16_16(Q15ONE, len), (i32)(len + factor * K));
   i16 theta = (i16)(mult16_16_q15(gain, gain) >> 1);
   i32 c = wv_uni(fx_cos_norm(theta));
   i32 s = wv_uni(fx_cos_norm(sub16(Q15ONE, theta)));
   if (len >= 8 * stride) {
      stride2 = 1;
      while ((stride2 * stride2 + stride2) * stride + (stride >> 2) < len) stride2++;
   }
   for (int t = 0; t < NR; t++) v[t] = pshr32(v[t], NORM_SHIFT - 14);       /* norm_scaledown once (up/down between passes cancels exactly) */
   len = fx_div_pow2(len, stride);                                          /* stride = the band's block count: a power of two */
   if (dir < 0) {
      if (stride2) rot_pass_any(v, T, stride, len, stride2, s, c);
      rot_pass_any(v, T, stride, len, 1, c, s);
   } else {
      rot_pass_any(v, T, stride, len, 1, c, -s);
      if (stride2) rot_pass_any(v, T, stride, len, stride2, s, -c);
   }
   for (int t = 0; t < NR; t++) v[t] = shl32(v[t], NORM_SHIFT - 14);
}

/* op_pvq_search (vq.c:205) on the register-resident band: x[] in, signed pulse vector q[] out; returns yy.  One
 * cross-lane arg-max per pulse on the DPP network; the winner's |X| and y are fetched with v_readlane. */
template <int NR> WV_DEV i32 op_pvq_search_regs(i32 (&x)[NR], i32 (&q)[NR], int K, int N)
{
   const int lane = wv_lane();
   i64 e2 = 0;
   for (int t = 0; t < NR; t++) e2 += x[t] * (i64)x[t];
   int shift = (celt_ilog2(1 + (i32)(wv_sum64(e2) >> 2 * (NORM_SHIFT - 14))) + 1) / 2;
   shift = imax(0, shift + (NORM_SHIFT - 14) - 14);
   bool vld[NR]; i32 sg[NR], y[NR];
   i32 xsum = 0;
   for (int t = 0; t < NR; t++) {
      vld[t] = lane + 64 * t < N;
      i32 xv = vld[t] ? pshr32(x[t], shift) : 0;
      sg[t] = xv < 0; x[t] = iabs(xv); y[t] = 0; q[t] = 0; xsum += x[t];
   }
   i32 xy = 0; i16 yy = 0;
   int pulsesLeft = K;
   if (K > (N >> 1)) {
      i32 sum = wv_sum(xsum);
      if (sum <= K) {
         for (int t = 0; t < NR; t++) x[t] = 0;
         if (lane == 0) x[0] = QC16(1.f, 14);
         sum = QC16(1.f, 14);
      }
      i16 rcp = extract16(mult16_32_q16(K, fx_rcp(sum)));
      i32 yyp = 0, xyp = 0, qs = 0;
      for (int t = 0; t < NR; t++) {
         q[t] = mult16_16_q15(x[t], rcp);
         yyp = mac16_16(yyp, q[t], q[t]); xyp = mac16_16(xyp, x[t], q[t]); y[t] = 2 * q[t]; qs += q[t];
      }
      yy = (i16)wv_sum(yyp);
      xy = wv_sum(xyp);
      pulsesLeft -= wv_sum(qs);
   }
   if (pulsesLeft > N + 3) {
      i16 tmp = (i16)pulsesLeft;
      i32 yfirst = wv_bcast(y[0], 0);
      yy = (i16)mac16_16(yy, tmp, tmp);
      yy = (i16)mac16_16(yy, tmp, yfirst);
      if (lane == 0) q[0] += pulsesLeft;
      pulsesLeft = 0;
   }
   const int nl = NR == 1 ? N : 64;
   for (int i = 0; i < pulsesLeft; i++) {
      int rshift = 1 + celt_ilog2(K - pulsesLeft + i + 1);
      yy = add16(yy, 1);
      /* per-lane best over its (<= NR) candidates; ties inside a lane go to the lower slot = lower index */
      u32 num[NR], den[NR];
      u32 bn = 0, bd = 1;
      for (int t = 0; t < NR; t++) {
         num[t] = 0; den[t] = 1;
         if (vld[t]) {
            i16 Rxy = extract16(add32(xy, x[t]) >> rshift); i16 Ryy = add16(yy, y[t]); Rxy = (i16)mult16_16_q15(Rxy, Rxy);
            num[t] = (u32)Rxy; den[t] = (u32)Ryy;
            if (t == 0 || bd * num[t] > den[t] * bn) { bn = num[t]; bd = den[t]; }
         }
      }
      int owner, slot = 0;
      if (NR == 1) owner = wv_argmax_ratio_packed(bn, bd, vld[0], nl);
      else {
         /* global index order is (slot, lane): find the maximal ratio first, then the lowest slot that attains it, then the lowest lane */
         const int any = wv_argmax_ratio_packed(bn, bd, vld[0], 64);
         const u32 gn = (u32)wv_bcast((i32)bn, any), gd = (u32)wv_bcast((i32)bd, any);
         owner = any;
         for (int t = 0; t < NR; t++) {
            const bool hit = vld[t] && den[t] * gn == gd * num[t];
            const unsigned long long m = wv_ballot(hit);
            if (m) { owner = (int)__builtin_ctzll(m); slot = t; break; }
         }
      }
      i32 xs = x[0], ys = y[0];
      for (int t = 1; t < NR; t++) if (slot == t) { xs = x[t]; ys = y[t]; }
      xy = add32(xy, wv_bcast(xs, owner));
      yy = add16(yy, wv_bcast(ys, owner));
      if (lane == owner) { for (int t = 0; t < NR; t++) if (slot == t) { y[t] += 2; q[t]++; } }
   }
   for (int t = 0; t < NR; t++) q[t] = (q[t] ^ -sg[t]) + sg[t];
   return yy;
}

/* encode_pulses (cwrs.c:444-465): index = (y[n-1]<0) + sum_j U(n-j, k_{j+1}) + [y_j<0] U(n-j, k_j+1), k_j = sum_{i>=j}|y_i|.
 * Suffix sums of |y| come from a wave scan per register; every table read is then independent (issued back to back). */
template <int NR> WV_DEV void encode_pulses_regs(WV_LDS FrameLds *L, const i32 (&yv)[NR], int N, int K, u32 ft /* V(N, K) = U(N, K) + U(N, K + 1): loaded by the caller before the search, out of the way of this stage's own table round trip */)
{
   const int lane = wv_lane();
   i32 a[NR], incl[NR], tot[NR];
   for (int t = 0; t < NR; t++) { a[t] = iabs(yv[t]); incl[t] = wv_scan_incl(a[t]); tot[t] = wv_bcast(incl[t], 63); }
   u32 idx = 0;
   i32 above = 0;                           /* pulses in higher registers */
   for (int t = NR - 1; t >= 0; t--) {
      const int j = lane + 64 * t;
      const i32 kafter = above + tot[t] - incl[t];      /* k_{j+1}: pulses strictly after element j */
      if (j < N - 1) {
         idx += pvq_u(N - j, kafter);
         if (yv[t] < 0) idx += pvq_u(N - j, kafter + a[t] + 1);
      } else if (j == N - 1) idx += yv[t] < 0;
      above += tot[t];
   }
   idx = wv_sumu(idx);
   LANE0 { EC_BEGIN; k_ec_enc_uint(EC_PASS, idx, ft); EC_END; }
}

/* alg_quant (vq.c:552): load the band once, rotate / search / index / resynthesise in registers, store once */
template <int NR> WV_DEV unsigned alg_quant_regs(WV_LDS FrameLds *L, WV_LDS i32 *X, int N, int K, int spread, int B, i32 gain, int resynth)
{
   const int lane = wv_lane();
   i32 v[NR], q[NR];
   K_DUMP("pvqX", X, N * 4);
   K_TIC();
   const u32 ft = pvq_u(N, K) + pvq_u(N, K + 1);                            /* (uniform addresses: in flight while the band is rotated and searched) */
   for (int t = 0; t < NR; t++) v[t] = lane + 64 * t < N ? X[lane + 64 * t] : 0;
   exp_rotation_regs(v, X, N, 1, B, K, spread);
   K_TOC(16);
   i32 yy = op_pvq_search_regs(v, q, K, N);
   K_TOC(17);
   unsigned cm = 1;
   if (B > 1) {
      int N0 = fx_div_pow2(N, B);
      u32 m = 0;
      for (int t = 0; t < NR; t++) if (q[t] != 0) m |= 1u << ((u32)(lane + 64 * t) / (u32)N0);
      cm = wv_or(m);
   }
#ifdef K_DUMP_ENABLED
   { WV_LDS i32 *iy = L->BC.q.iy; wv_sync(); for (int t = 0; t < NR; t++) if (lane + 64 * t < N) iy[lane + 64 * t] = q[t]; wv_sync(); K_DUMP("iy", iy, N * 4); K_DUMPI("pvqK", K); }
#endif
   encode_pulses_regs(L, q, N, K, ft);
   K_TOC(18);
   if (resynth) {
      int k = celt_ilog2(yy) >> 1;
      i32 t_ = vshr32(yy, 2 * (k - 7) - 15);
      i32 g = mult32_32_q31(fx_rsqrt_norm32(t_), gain);
      for (int t = 0; t < NR; t++) v[t] = vshr32(mult16_32_q15(q[t], g), k + 15 - NORM_SHIFT);
      exp_rotation_regs(v, X, N, -1, B, K, spread);
      wv_sync();
      for (int t = 0; t < NR; t++) if (lane + 64 * t < N) X[lane + 64 * t] = v[t];
      wv_sync();
   }
   K_TOC(19);
   return cm;
}
OA_ALG_QUANT_FN unsigned alg_quant_wave(WV_LDS FrameLds *L, WV_LDS i32 *X, int N, int K, int spread, int B, i32 gain, int resynth)
{
   N = wv_uni(N); K = wv_uni(K); spread = wv_uni(spread); B = wv_uni(B); gain = wv_uni(gain); resynth = wv_uni(resynth);
   if (N <= 64) return alg_quant_regs<1>(L, X, N, K, spread, B, gain, resynth);
   return alg_quant_regs<3>(L, X, N, K, spread, B, gain, resynth);
}
WV_DEV void renormalise_vector_wave(WV_LDS i32 *X, int N, i32 gain)
{
   i32 e = 0;
   FOR_LANES(i, N) { i32 v = pshr32(X[i], NORM_SHIFT - 14); e = add32(e, (i32)((u32)v * (u32)v)); }
   i32 E = add32(EPSILON, wv_sum(e));
   int k = celt_ilog2(E) >> 1;
   i32 t = vshr32(E, 2 * (k - 7));
   i16 g = (i16)mult32_32_q31(fx_rsqrt_norm(t), gain);
   FOR_LANES(i, N) { i32 v = pshr32(X[i], NORM_SHIFT - 14); X[i] = shl32((i32)extract16(pshr32(mult16_16(g, v), k + 15 - 14)), NORM_SHIFT - 14); }
   wv_sync();
}

/* ---- band recursion ----------------------------------------------------------------------------------------------
 * Everything the recursion carries is wave-uniform.  It travels BY VALUE (registers), never through pointers to private
 * memory (that would be scratch = HBM-latency accesses on the critical path): BandCfg is the read-only part of the
 * reference's band_ctx (bands.c:664), remaining_bits / seed are threaded through arguments and vector returns. */
typedef i32 i32x4 __attribute__((vector_size(16)));
typedef i32 i32x8 __attribute__((vector_size(32)));
WV_DEV i32x4 ret3(unsigned cm, i32 rem, u32 seed) { i32x4 r = {(i32)cm, rem, (i32)seed, 0}; return r; }
WV_DEV BandCfg cfg_uni(BandCfg c)
{
   c.resynth = wv_uni(c.resynth); c.i = wv_uni(c.i); c.intensity = wv_uni(c.intensity); c.spread = wv_uni(c.spread); c.tf_change = wv_uni(c.tf_change);
   c.theta_round = wv_uni(c.theta_round); c.disable_inv = wv_uni(c.disable_inv); c.avoid_split_noise = wv_uni(c.avoid_split_noise);
   return c;
}

/* one row of the pulse cache (rate.h:48-66 get_pulses/bits2pulses/pulses2bits): cache[0..40] for (LM, band) held one entry per
 * lane, so the bisection and every later lookup are v_readlane's instead of dependent byte loads from memory */
/* The five rows a band's partitions can ask for (LM + 1 = 0 .. 4: every split halves the band) are staged in LDS once per band -- L->scr is free during the PVQ -- by
 * quant_all_bands: ONE global-memory round trip per band (all five rows in flight together) instead of two dependent ones (index, then row) per partition call, of which a
 * frame has some three hundred.  The wave is latency-bound here: its time is the sum of its round trips. */
#define PVQ_ROW_STRIDE 64
WV_DEV void cache_rows_stage(WV_LDS FrameLds *L, int band)
{
   WV_LDS u8 *rows = (WV_LDS u8 *)L->scr;
   wv_sync();
   for (int t = wv_lane(); t < 5 * PVQ_ROW_STRIDE; t += WV_WIDTH) {
      const int d = t / PVQ_ROW_STRIDE, e = t - d * PVQ_ROW_STRIDE;
      const int off = ct_cache_index[d * OA_NB_EBANDS + band];
      rows[t] = ct_cache_bits[imax(0, imin(off + imin(e, 40), (int)sizeof(ct_cache_bits) - 1))];   /* (the last rows are shorter than 41 entries: the lanes beyond a row's end read values nobody uses, but stay inside the table) */
   }
   wv_sync();
}
WV_DEV i32 cache_row_lds(WV_LDS FrameLds *L, int LM) { return (i32)((const WV_LDS u8 *)L->scr)[(LM + 1) * PVQ_ROW_STRIDE + imin(wv_lane(), 40)]; }
/* (the decoder's partitions -- celt_dec_bands.h -- read their row from the table: its LDS layout has no such slot) */
WV_DEV i32 cache_row_load(int band, int LM)
{
   const int off = ct_cache_index[(LM + 1) * OA_NB_EBANDS + band];
   return (i32)ct_cache_bits[imin(off + imin(wv_lane(), 40), (int)sizeof(ct_cache_bits) - 1)];
}
WV_DEV int row_bits2pulses(i32 row, int bits)
{
   int lo = 0, hi = wv_bcast(row, 0);
   bits--;
   for (int i = 0; i < LOG_MAX_PSEUDO; i++) {
      int mid = (lo + hi + 1) >> 1;
      if (wv_bcast(row, mid) >= bits) hi = mid; else lo = mid;
   }
   if (bits - (lo == 0 ? -1 : wv_bcast(row, lo)) <= wv_bcast(row, hi) - bits) return lo;
   return hi;
}
WV_DEV int row_pulses2bits(i32 row, int pulses) { return pulses == 0 ? 0 : wv_bcast(row, pulses) + 1; }

/* compute_theta (bands.c:700).  Returns {inv, imid, iside, delta, itheta, qalloc, b, fill}. */
WV_DEVN i32x8 compute_theta_wave(WV_LDS FrameLds *L, BandCfg cfg, i32 remaining_bits, WV_LDS i32 *X, WV_LDS i32 *Y, int N, int b, int B, int B0,
      int LM, int stereo, int fill)
{
   cfg = cfg_uni(cfg); remaining_bits = wv_uni(remaining_bits); N = wv_uni(N); b = wv_uni(b); B = wv_uni(B); B0 = wv_uni(B0); LM = wv_uni(LM);
   stereo = wv_uni(stereo); fill = wv_uni(fill);
   int qn, itheta = 0, delta, imid, iside, qalloc, pulse_cap, offset, inv = 0;
   const int i = cfg.i, intensity = cfg.intensity;
   K_TIC();
   pulse_cap = ct_logN[i] + LM * (1 << BITRES);
   offset = (pulse_cap >> 1) - (stereo && N == 2 ? 16 : 4);
   qn = compute_qn(N, b, offset, pulse_cap, stereo);
   if (stereo && i >= intensity) qn = 1;
   itheta = stereo_itheta_wave(X, Y, stereo, N) >> 16;
   wv_sync();
   i32 tell = ec_tell_frac_lds(&L->ec);
   wv_sync();
   if (qn != 1) {
      if (!stereo || cfg.theta_round == 0) {
         itheta = (itheta * (i32)qn + 8192) >> 14;
         if (!stereo && cfg.avoid_split_noise && itheta > 0 && itheta < qn) {
            int unquantized = (int)fx_udiv24((u32)((i32)itheta * 16384), (u32)qn);
            imid = bitexact_cos((i16)unquantized);
            iside = bitexact_cos((i16)(16384 - unquantized));
            delta = frac_mul16((N - 1) << 7, bitexact_log2tan(iside, imid));
            if (delta > b) itheta = qn;
            else if (delta < -b) itheta = 0;
         }
      } else {
         const int q32767 = (int)fx_udiv24(32767u, (u32)qn);
         int bias = itheta > 8192 ? q32767 : -q32767;
         int down = imin(qn - 1, imax(0, (itheta * (i32)qn + bias) >> 14));
         itheta = cfg.theta_round < 0 ? down : down + 1;
      }
      LANE0 {
         EC_BEGIN;
         if (stereo && N > 2) {
            int p0 = 3, x = itheta, x0 = qn / 2, ft = p0 * (x0 + 1) + x0;
            k_ec_encode(EC_PASS, x <= x0 ? p0 * x : (x - 1 - x0) + (x0 + 1) * p0, x <= x0 ? p0 * (x + 1) : (x - x0) + (x0 + 1) * p0, ft);
         } else if (B0 > 1 || stereo) {
            k_ec_enc_uint(EC_PASS, itheta, qn + 1);
         } else {
            int ft = ((qn >> 1) + 1) * ((qn >> 1) + 1);
            int fs = itheta <= (qn >> 1) ? itheta + 1 : qn + 1 - itheta;
            int fl = itheta <= (qn >> 1) ? itheta * (itheta + 1) >> 1 : ft - ((qn + 1 - itheta) * (qn + 2 - itheta) >> 1);
            k_ec_encode(EC_PASS, fl, fl + fs, ft);
         }
         EC_END;
      }
      itheta = (int)fx_udiv24((u32)((i32)itheta * 16384), (u32)qn);
      if (stereo) {
         if (itheta == 0) intensity_stereo_wave(L, X, Y, i, N);
         else stereo_split_wave(X, Y, N);
      }
   } else if (stereo) {
      inv = itheta > 8192 && !cfg.disable_inv;
      if (inv) { FOR_LANES(j, N) Y[j] = neg32(Y[j]); wv_sync(); }
      intensity_stereo_wave(L, X, Y, i, N);
      if (b > 2 << BITRES && remaining_bits > 2 << BITRES) {
         LANE0 { EC_BEGIN; k_ec_enc_bit_logp(EC_PASS, inv, 2); EC_END; }
      } else inv = 0;
      if (cfg.disable_inv) inv = 0;
      itheta = 0;
   }
   wv_sync();
   qalloc = ec_tell_frac_lds(&L->ec) - tell;
   b -= qalloc;
   if (itheta == 0) { imid = 32767; iside = 0; fill &= (1 << B) - 1; delta = -16384; }
   else if (itheta == 16384) { imid = 0; iside = 32767; fill &= ((1 << B) - 1) << B; delta = 16384; }
   else {
      imid = bitexact_cos((i16)itheta);
      iside = bitexact_cos((i16)(16384 - itheta));
      delta = frac_mul16((N - 1) << 7, bitexact_log2tan(iside, imid));
   }
   K_DUMPI("itheta", itheta); K_DUMPI("qn", qn);
   K_TOC(20);
   i32x8 r = {inv, imid, iside, delta, itheta, qalloc, b, fill};
   return r;
}

WV_DEV unsigned quant_band_n1_wave(WV_LDS FrameLds *L, const BandCfg &cfg, i32 &remaining_bits, WV_LDS i32 *X, WV_LDS i32 *Y, WV_LDS i32 *lowband_out)
{
   WV_LDS i32 *x = X;
   int stereo = Y != 0;
   wv_sync();
   for (int c = 0; c < 1 + stereo; c++) {
      int sign = 0;
      if (remaining_bits >= 1 << BITRES) {
         sign = x[0] < 0;
         LANE0 { EC_BEGIN; k_ec_enc_bits(EC_PASS, sign, 1); EC_END; }
         remaining_bits -= 1 << BITRES;
      }
      if (cfg.resynth) { wv_sync(); LANE0 x[0] = sign ? -(1 << NORM_SHIFT) : (1 << NORM_SHIFT); wv_sync(); }
      x = Y;
   }
   wv_sync();
   if (lowband_out) { LANE0 lowband_out[0] = X[0] >> 4; }
   wv_sync();
   return 1;
}

/* quant_partition (bands.c:973).  Returns {collapse mask, remaining_bits, seed}.  The body is inlined where depth 0 starts (one call site, quant_band_wave); the deeper
 * levels are out-of-line instantiations (each saves / restores the VGPRs it keeps SGPR spills in: a per-call cost worth paying only where the code would multiply). */
template <int DEPTH> WV_DEVN i32x4 quant_partition_wave(WV_LDS FrameLds *L, BandCfg cfg, i32 remaining_bits, u32 seed, WV_LDS i32 *X, int N, int b, int B, WV_LDS i32 *lowband,
      int LM, i32 gain, int fill);
template <int DEPTH>
WV_DEV i32x4 quant_partition_body(WV_LDS FrameLds *L, BandCfg cfg, i32 remaining_bits, u32 seed, WV_LDS i32 *X, int N, int b, int B, WV_LDS i32 *lowband,
      int LM, i32 gain, int fill)
{
   cfg = cfg_uni(cfg); remaining_bits = wv_uni(remaining_bits); seed = (u32)wv_uni((i32)seed); N = wv_uni(N); b = wv_uni(b); B = wv_uni(B);
   LM = wv_uni(LM); gain = wv_uni(gain); fill = wv_uni(fill);
   int B0 = B;
   const int i = cfg.i, spread = cfg.spread;
   unsigned cm = 0;
   const i32 row = cache_row_lds(L, LM);
   bool split = LM != -1 && b > wv_bcast(row, wv_bcast(row, 0)) + 12 && N > 2;
   if constexpr (DEPTH < 4) {
      if (split) {
         int mbits, sbits, delta, itheta, qalloc;
         WV_LDS i32 *next_lowband2 = 0, *Y;
         i32 rebalance, mid, side;
         N >>= 1;
         Y = X + N;
         LM -= 1;
         if (B == 1) fill = (fill & 1) | (fill << 1);
         B = (B + 1) >> 1;
         const i32x8 th = compute_theta_wave(L, cfg, remaining_bits, X, Y, N, b, B, B0, LM, 0, fill);
         delta = wv_uni(th[3]); itheta = wv_uni(th[4]); qalloc = wv_uni(th[5]); b = wv_uni(th[6]); fill = wv_uni(th[7]);
         mid = shl32((i32)wv_uni(th[1]), 16);
         side = shl32((i32)wv_uni(th[2]), 16);
         if (B0 > 1 && (itheta & 0x3fff)) {
            if (itheta > 8192) delta -= delta >> (4 - LM);
            else delta = imin(0, delta + (N << BITRES >> (5 - LM)));
         }
         mbits = imax(0, imin(b, (b - delta) / 2));
         sbits = b - mbits;
         remaining_bits -= qalloc;
         if (lowband) next_lowband2 = lowband + N;
         rebalance = remaining_bits;
         i32x4 r;
         if (mbits >= sbits) {
            r = quant_partition_wave<DEPTH + 1>(L, cfg, remaining_bits, seed, X, N, mbits, B, lowband, LM, mult32_32_q31(gain, mid), fill);
            cm = (unsigned)wv_uni(r[0]); remaining_bits = wv_uni(r[1]); seed = (u32)wv_uni(r[2]);
            rebalance = mbits - (rebalance - remaining_bits);
            if (rebalance > 3 << BITRES && itheta != 0) sbits += rebalance - (3 << BITRES);
            r = quant_partition_wave<DEPTH + 1>(L, cfg, remaining_bits, seed, Y, N, sbits, B, next_lowband2, LM, mult32_32_q31(gain, side), fill >> B);
            cm |= (unsigned)wv_uni(r[0]) << (B0 >> 1); remaining_bits = wv_uni(r[1]); seed = (u32)wv_uni(r[2]);
         } else {
            r = quant_partition_wave<DEPTH + 1>(L, cfg, remaining_bits, seed, Y, N, sbits, B, next_lowband2, LM, mult32_32_q31(gain, side), fill >> B);
            cm = (unsigned)wv_uni(r[0]) << (B0 >> 1); remaining_bits = wv_uni(r[1]); seed = (u32)wv_uni(r[2]);
            rebalance = sbits - (rebalance - remaining_bits);
            if (rebalance > 3 << BITRES && itheta != 16384) mbits += rebalance - (3 << BITRES);
            r = quant_partition_wave<DEPTH + 1>(L, cfg, remaining_bits, seed, X, N, mbits, B, lowband, LM, mult32_32_q31(gain, mid), fill);
            cm |= (unsigned)wv_uni(r[0]); remaining_bits = wv_uni(r[1]); seed = (u32)wv_uni(r[2]);
         }
         return ret3(cm, remaining_bits, seed);
      }
   }
   {
      int q = row_bits2pulses(row, b);
      int curr_bits = row_pulses2bits(row, q);
      remaining_bits -= curr_bits;
      while (remaining_bits < 0 && q > 0) {
         remaining_bits += curr_bits;
         q--;
         curr_bits = row_pulses2bits(row, q);
         remaining_bits -= curr_bits;
      }
      if (q != 0) {
         int K = k_get_pulses(q);
         cm = alg_quant_wave(L, X, N, K, spread, B, gain, cfg.resynth);
      } else if (cfg.resynth) {
         unsigned cm_mask = (unsigned)(1UL << B) - 1;
         fill &= cm_mask;
         if (!fill) { FOR_LANES(j, N) X[j] = 0; wv_sync(); }
         else {
            wv_sync();
            if (lowband == 0) {
               LANE0 { u32 s = seed; for (int j = 0; j < N; j++) { s = lcg_rand(s); X[j] = shl32((i32)((i32)s >> 20), NORM_SHIFT - 14); } }
               cm = cm_mask;
            } else {
               LANE0 {
                  u32 s = seed;
                  for (int j = 0; j < N; j++) {
                     s = lcg_rand(s);
                     i16 tmp = QC16(1.0f / 256, NORM_SHIFT - 4);
                     tmp = (s) & 0x8000 ? tmp : -tmp;
                     X[j] = lowband[j] + tmp;
                  }
               }
               cm = fill;
            }
            for (int j = 0; j < N; j++) seed = lcg_rand(seed);      /* uniform: the same N steps the lane-0 loop took */
            wv_sync();
            renormalise_vector_wave(X, N, gain);
         }
      }
   }
   return ret3(cm, remaining_bits, seed);
}

template <int DEPTH> WV_DEVN i32x4 quant_partition_wave(WV_LDS FrameLds *L, BandCfg cfg, i32 remaining_bits, u32 seed, WV_LDS i32 *X, int N, int b, int B, WV_LDS i32 *lowband,
      int LM, i32 gain, int fill)
{
   return quant_partition_body<DEPTH>(L, cfg, remaining_bits, seed, X, N, b, B, lowband, LM, gain, fill);
}

/* quant_band (bands.c:1248).  Returns {collapse mask, remaining_bits, seed}. */
OA_QUANT_BAND_FN i32x4 quant_band_wave(WV_LDS FrameLds *L, BandCfg cfg, i32 remaining_bits, u32 seed, WV_LDS i32 *X, int N, int b, int B, WV_LDS i32 *lowband, int LM,
      WV_LDS i32 *lowband_out, i32 gain, WV_LDS i32 *lowband_scratch, int fill)
{
   cfg = cfg_uni(cfg); remaining_bits = wv_uni(remaining_bits); seed = (u32)wv_uni((i32)seed); N = wv_uni(N); b = wv_uni(b); B = wv_uni(B);
   LM = wv_uni(LM); gain = wv_uni(gain); fill = wv_uni(fill);
   const u8 bit_interleave_table[16] = {0, 1, 1, 1, 2, 3, 3, 3, 2, 3, 3, 3, 2, 3, 3, 3};
   const u8 bit_deinterleave_table[16] = {0x00, 0x03, 0x0C, 0x0F, 0x30, 0x33, 0x3C, 0x3F, 0xC0, 0xC3, 0xCC, 0xCF, 0xF0, 0xF3, 0xFC, 0xFF};
   int N0 = N, N_B = N, N_B0, B0 = B, time_divide = 0, recombine = 0, longBlocks, k;
   unsigned cm = 0;
   int tf_change = cfg.tf_change;
   longBlocks = B0 == 1;
   N_B = fx_div_pow2(N_B, B);                                               /* (B: 1, 2, 4, 8 or 16 blocks) */
   if (N == 1) { cm = quant_band_n1_wave(L, cfg, remaining_bits, X, 0, lowband_out); return ret3(cm, remaining_bits, seed); }
   K_TIC();
   if (tf_change > 0) recombine = tf_change;
   if (lowband_scratch && lowband && (recombine || ((N_B & 1) == 0 && tf_change < 0) || B0 > 1)) {
      wv_sync();
      FOR_LANES(j, N) lowband_scratch[j] = lowband[j];
      wv_sync();
      lowband = lowband_scratch;
   }
   for (k = 0; k < recombine; k++) {
      haar1_wave(X, N >> k, 1 << k);
      if (lowband) haar1_wave(lowband, N >> k, 1 << k);
      fill = bit_interleave_table[fill & 0xF] | bit_interleave_table[fill >> 4] << 2;
   }
   B >>= recombine;
   N_B <<= recombine;
   while ((N_B & 1) == 0 && tf_change < 0) {
      haar1_wave(X, N_B, B);
      if (lowband) haar1_wave(lowband, N_B, B);
      fill |= fill << B;
      B <<= 1;
      N_B >>= 1;
      time_divide++;
      tf_change++;
   }
   B0 = B;
   N_B0 = N_B;
   if (B0 > 1) {
      deinterleave_hadamard_wave(X, N_B >> recombine, B0 << recombine, longBlocks);
      if (lowband) deinterleave_hadamard_wave(lowband, N_B >> recombine, B0 << recombine, longBlocks);
   }
   K_TOC(22);
   {
      const i32x4 r = quant_partition_body<0>(L, cfg, remaining_bits, seed, X, N, b, B, lowband, LM, gain, fill);
      cm = (unsigned)wv_uni(r[0]); remaining_bits = wv_uni(r[1]); seed = (u32)wv_uni(r[2]);
   }
   K_TOC(24);
   if (cfg.resynth) {
      if (B0 > 1) interleave_hadamard_wave(X, N_B >> recombine, B0 << recombine, longBlocks);
      N_B = N_B0;
      B = B0;
      for (k = 0; k < time_divide; k++) {
         B >>= 1;
         N_B <<= 1;
         cm |= cm >> B;
         haar1_wave(X, N_B, B);
      }
      for (k = 0; k < recombine; k++) {
         cm = bit_deinterleave_table[cm];
         haar1_wave(X, N0 >> k, 1 << k);
      }
      B <<= recombine;
      if (lowband_out) {
         i16 n = (i16)fx_sqrt(shl32((i32)N0, 22));
         wv_sync();
         FOR_LANES(j, N0) lowband_out[j] = mult16_32_q15(n, X[j]);
         wv_sync();
      }
      cm &= (1 << B) - 1;
   }
   K_TOC(22);
   return ret3(cm, remaining_bits, seed);
}

/* quant_band_stereo (bands.c:1387).  Returns {collapse mask, remaining_bits, seed}. */
OA_PVQ_STEREO_FN i32x4 quant_band_stereo_wave(WV_LDS FrameLds *L, BandCfg cfg, i32 remaining_bits, u32 seed, WV_LDS i32 *X, WV_LDS i32 *Y, int N, int b, int B,
      WV_LDS i32 *lowband, int LM, WV_LDS i32 *lowband_out, WV_LDS i32 *lowband_scratch, int fill)
{
   cfg = cfg_uni(cfg); remaining_bits = wv_uni(remaining_bits); seed = (u32)wv_uni((i32)seed); N = wv_uni(N); b = wv_uni(b); B = wv_uni(B);
   LM = wv_uni(LM); fill = wv_uni(fill);
   int inv = 0, mbits, sbits, delta, itheta, qalloc, orig_fill;
   i32 mid = 0, side = 0;
   unsigned cm = 0;
   i32x4 r;
   if (N == 1) { cm = quant_band_n1_wave(L, cfg, remaining_bits, X, Y, lowband_out); return ret3(cm, remaining_bits, seed); }
   orig_fill = fill;
   if (L->bandE[cfg.i] < 2 || L->bandE[NBE + cfg.i] < 2) {
      wv_sync();
      if (L->bandE[cfg.i] > L->bandE[NBE + cfg.i]) { FOR_LANES(j, N) Y[j] = X[j]; }
      else { FOR_LANES(j, N) X[j] = Y[j]; }
      wv_sync();
   }
   {
      const i32x8 th = compute_theta_wave(L, cfg, remaining_bits, X, Y, N, b, B, B, LM, 1, fill);
      inv = wv_uni(th[0]); delta = wv_uni(th[3]); itheta = wv_uni(th[4]); qalloc = wv_uni(th[5]); b = wv_uni(th[6]); fill = wv_uni(th[7]);
      mid = shl32((i32)wv_uni(th[1]), 16);
      side = shl32((i32)wv_uni(th[2]), 16);
   }
   /* One quant_band site for every way through this function (the code object carried five inlined copies of it per copy of this function, fifteen in all, 216 KB of
    * quant_all_bands: the hot loop of the frame did not fit the 64 KB instruction cache two CUs share): N == 2 codes one vector (the larger of mid and side, bands.c:1437-1476),
    * everything else codes mid and side in the order of their budgets, the second one with what the first left over (:1479-1527). */
   WV_LDS i32 *x2 = X, *y2 = Y;
   int sign = 0, mid_first = 1, npass = 2;
   i32 rebalance = 0;
   if (N == 2) {
      mbits = b;
      sbits = 0;
      if (itheta != 0 && itheta != 16384) sbits = 1 << BITRES;
      mbits -= sbits;
      const int c = itheta > 8192;
      remaining_bits -= qalloc + sbits;
      x2 = c ? Y : X;
      y2 = c ? X : Y;
      wv_sync();
      if (sbits) {
         sign = mult32_32_q31(x2[0], y2[1]) - mult32_32_q31(x2[1], y2[0]) < 0;
         LANE0 { EC_BEGIN; k_ec_enc_bits(EC_PASS, sign, 1); EC_END; }
      }
      sign = 1 - 2 * sign;
      npass = 1;
   } else {
      mbits = imax(0, imin(b, (b - delta) / 2));
      sbits = b - mbits;
      remaining_bits -= qalloc;
      rebalance = remaining_bits;
      mid_first = mbits >= sbits;
   }
#pragma nounroll
   for (int pass = 0; pass < npass; pass++) {
      const int do_mid = (pass == 0) == (mid_first != 0);
      if (pass == 1) {
         if (mid_first) { rebalance = mbits - (rebalance - remaining_bits); if (rebalance > 3 << BITRES && itheta != 0) sbits += rebalance - (3 << BITRES); }
         else { rebalance = sbits - (rebalance - remaining_bits); if (rebalance > 3 << BITRES && itheta != 16384) mbits += rebalance - (3 << BITRES); }
      }
      r = quant_band_wave(L, cfg, remaining_bits, seed, N == 2 ? x2 : do_mid ? X : Y, N, do_mid ? mbits : sbits, B, do_mid ? lowband : (WV_LDS i32 *)0, LM,
            do_mid ? lowband_out : (WV_LDS i32 *)0, do_mid ? Q31ONE : side, do_mid ? lowband_scratch : (WV_LDS i32 *)0, N == 2 ? orig_fill : do_mid ? fill : fill >> B);
      cm |= (unsigned)wv_uni(r[0]); remaining_bits = wv_uni(r[1]); seed = (u32)wv_uni(r[2]);
   }
   if (N == 2) {
      wv_sync();
      LANE0 { y2[0] = -sign * x2[1]; y2[1] = sign * x2[0]; }
      wv_sync();
      if (cfg.resynth) {
         LANE0 {
            i32 tmp;
            X[0] = mult32_32_q31(mid, X[0]);
            X[1] = mult32_32_q31(mid, X[1]);
            Y[0] = mult32_32_q31(side, Y[0]);
            Y[1] = mult32_32_q31(side, Y[1]);
            tmp = X[0]; X[0] = sub32(tmp, Y[0]); Y[0] = add32(tmp, Y[0]);
            tmp = X[1]; X[1] = sub32(tmp, Y[1]); Y[1] = add32(tmp, Y[1]);
         }
         wv_sync();
      }
   }
   if (cfg.resynth) {
      K_TIC();
      if (N != 2) stereo_merge_wave(X, Y, mid, N);
      if (inv) { FOR_LANES(j, N) Y[j] = neg32(Y[j]); wv_sync(); }
      K_TOC(23);
   }
   return ret3(cm, remaining_bits, seed);
}

/* quant_all_bands (bands.c:1589), encoder side */
WV_DEVN void quant_all_bands_wave(WV_LDS FrameLds *L, int shortBlocks, int spread, int dual_stereo, int intensity, i32 total_bits, i32 balance,
      int codedBands, int complexity, int disable_inv, u8 *journal)
{
   shortBlocks = wv_uni(shortBlocks); spread = wv_uni(spread); dual_stereo = wv_uni(dual_stereo); intensity = wv_uni(intensity); total_bits = wv_uni(total_bits);
   balance = wv_uni(balance); codedBands = wv_uni(codedBands); complexity = wv_uni(complexity); disable_inv = wv_uni(disable_inv);
   const int start = wv_uni(L->sh.start), end = wv_uni(L->sh.end), LM = wv_uni(L->sh.LM), C = wv_uni(L->sh.C), Nfull = wv_uni(L->sh.N);
   CeltScratch *G = L->g;
   const i32 *X_ = G->X, *Y_ = C == 2 ? G->X + Nfull : 0;           /* the spectrum stays in HBM; the band being coded is staged into Xb / Yb */
   WV_LDS i32 *norm = L->BC.q.norm, *norm2 = L->BC.q.u.norm2;
   WV_LDS i32 *const X = L->BC.q.Xb, *const Yb = L->BC.q.u.Yb;
   i32 *X_save2 = G->X_save2, *Y_save2 = G->Y_save2, *norm_save2 = G->norm_save2;
   WV_LDS u8 *collapse_masks = L->collapse_masks;
   const WV_LDS i32 *pulses = L->pulses, *tf_res = L->tf_res;
   i32 remaining_bits;
   int M = 1 << LM, B = shortBlocks ? M : 1, lowband_offset = 0, update_lowband = 1;
   int norm_offset = M * ct_eBands[start];
   int theta_rdo = Y_ != 0 && !dual_stereo && complexity >= 8;
   int resynth = theta_rdo;
   WV_LDS i32 *lowband_scratch = L->BC.q.lowband_scratch;
   BandCfg cfg;
   u32 seed = (u32)wv_uni((i32)L->st.rng);
   i32x4 r;
   cfg.intensity = intensity; cfg.spread = spread; cfg.disable_inv = disable_inv; cfg.resynth = resynth;
   cfg.theta_round = 0; cfg.avoid_split_noise = B > 1; cfg.i = 0; cfg.tf_change = 0;
   for (int i = start; i < end; i++) {
      i32 tell, curr_balance;
      int b, N, effective_lowband = -1, tf_change = 0, last;
      WV_LDS i32 *Y;
      unsigned x_cm, y_cm;
      cfg.i = i;
      cache_rows_stage(L, i);
      last = (i == end - 1);
      const i32 *Xg = X_ + M * ct_eBands[i], *Yg = Y_ != 0 ? Y_ + M * ct_eBands[i] : 0;
      Y = Y_ != 0 ? Yb : 0;
      N = M * ct_eBands[i + 1] - M * ct_eBands[i];
      wv_sync();
      tell = wv_uni(ec_tell_frac_lds(&L->ec));
      if (i != start) balance -= tell;
      remaining_bits = total_bits - tell - 1;
      if (i <= codedBands - 1) {
         curr_balance = fx_sdiv24(balance, imin(3, codedBands - i));
         b = imax(0, imin(16383, imin(remaining_bits + 1, wv_uni(pulses[i]) + curr_balance)));
      } else b = 0;
      if (resynth && (M * ct_eBands[i] - N >= M * ct_eBands[start] || i == start + 1) && (update_lowband || lowband_offset == 0))
         lowband_offset = i;
      /* special_hybrid_folding (bands.c:1575, RFC 8251 section 9): enough of the first band's folding data is duplicated for the second band to fold from (nothing when
       * start == 0).  It only matters to an encoder that resynthesises -- the theta RDO compares reconstructions -- and there it has to be redone before the second RDO
       * attempt of that band, whose first attempt wrote its own reconstruction over the copy (:1869) */
      const int hf_n1 = M * (ct_eBands[start + 1] - ct_eBands[start]), hf_n2 = M * (ct_eBands[start + 2] - ct_eBands[start + 1]);
      if (resynth && i == start + 1) {
         wv_sync();
         FOR_LANES(j, hf_n2 - hf_n1) { norm[hf_n1 + j] = norm[2 * hf_n1 - hf_n2 + j]; if (dual_stereo) norm2[hf_n1 + j] = norm2[2 * hf_n1 - hf_n2 + j]; }
         wv_sync();
      }
      tf_change = wv_uni(tf_res[i]);
      cfg.tf_change = tf_change;
      if (last && !theta_rdo) lowband_scratch = 0;
      if (lowband_offset != 0 && (spread != 3 || B > 1 || tf_change < 0)) {
         int fold_start, fold_end, fold_i;
         effective_lowband = imax(0, M * ct_eBands[lowband_offset] - norm_offset - N);
         fold_start = lowband_offset;
         while (M * ct_eBands[--fold_start] > effective_lowband + norm_offset);
         fold_end = lowband_offset - 1;
         while (++fold_end < i && M * ct_eBands[fold_end] < effective_lowband + norm_offset + N);
         x_cm = y_cm = 0;
         fold_i = fold_start;
         do {
            x_cm |= collapse_masks[fold_i * C + 0];
            y_cm |= collapse_masks[fold_i * C + C - 1];
         } while (++fold_i < fold_end);
         x_cm = (unsigned)wv_uni((i32)x_cm); y_cm = (unsigned)wv_uni((i32)y_cm);
      } else x_cm = y_cm = (1 << B) - 1;
      if (dual_stereo && i == intensity) {
         dual_stereo = 0;
         if (resynth) { wv_sync(); FOR_LANES(j, M * ct_eBands[i] - norm_offset) norm[j] = half32(norm[j] + norm2[j]); wv_sync(); }
      }
      /* stage the band: in dual stereo the two channels are coded one after the other through Xb (Yb's bytes hold norm2) */
      FOR_LANES(j, N) { X[j] = Xg[j]; if (Yg != 0 && !dual_stereo) Yb[j] = Yg[j]; }
      wv_sync();
      WV_LDS i32 *lb = effective_lowband != -1 ? norm + effective_lowband : 0;
      WV_LDS i32 *lb2 = effective_lowband != -1 ? norm2 + effective_lowband : 0;
      WV_LDS i32 *lbo = last ? 0 : norm + M * ct_eBands[i] - norm_offset;
      WV_LDS i32 *lbo2 = last ? 0 : norm2 + M * ct_eBands[i] - norm_offset;
      if (dual_stereo || Y == 0) {
         /* one channel (mono), or the two of a dual-stereo band one after the other through Xb with half the budget each (bands.c:1831-1841) -- one quant_band site for both */
         const int nc = dual_stereo ? 2 : 1;
#pragma nounroll
         for (int c = 0; c < nc; c++) {
            if (c == 1) { wv_sync(); FOR_LANES(j, N) X[j] = Yg[j]; wv_sync(); }
            r = quant_band_wave(L, cfg, remaining_bits, seed, X, N, dual_stereo ? b / 2 : b, B, c ? lb2 : lb, LM, c ? lbo2 : lbo, Q31ONE, lowband_scratch,
                  dual_stereo ? (c ? y_cm : x_cm) : (x_cm | y_cm));
            if (c == 0) x_cm = (unsigned)wv_uni(r[0]); else y_cm = (unsigned)wv_uni(r[0]);
            remaining_bits = wv_uni(r[1]); seed = (u32)wv_uni(r[2]);
         }
         if (!dual_stereo) y_cm = x_cm;
      } else {
         /* joint stereo; with the theta RDO (bands.c:1842-1912) the band is coded twice -- theta rounded down, then up -- and the trial with the smaller weighted distortion
          * stays.  One quant_band_stereo site serves the two trials and the band without RDO (theta_round 0) */
         const int rdo = theta_rdo && i < intensity;
         i32 dist0 = 0, dist1, rem1 = 0;
         u32 seed1 = 0;
         unsigned cm2 = 0;
         const unsigned cm = x_cm | y_cm;
         const i32 rem0 = remaining_bits; const u32 seed0 = seed;             /* both trials start from the same budget and the same noise seed */
         i16 w[2] = {0, 0};
         int nstart_bytes = 0, save_bytes = 0;
         WV_LDS u8 *bytes_buf = L->packet + 1;
         if (rdo) {
            compute_channel_weights(L->bandE[i], L->bandE[i + NBE], w);
            K_TIC();
            wv_sync();
            LANE0 ec_cp_lds(&L->ecsave[0], &L->ec);
            wv_sync();                                                   /* (the untouched band is the spectrum in HBM: nothing to save) */
            K_TOC(21);
         }
#pragma nounroll
         for (int tr = 0; tr < 1 + rdo; tr++) {
            K_TIC();
            if (tr == 1) {
               wv_sync();
               dist0 = mult16_32_q15(w[0], inner_prod_norm_shift_gw(Xg, X, N)) + mult16_32_q15(w[1], inner_prod_norm_shift_gw(Yg, Y, N));
               cm2 = x_cm; rem1 = remaining_bits; seed1 = seed;
               LANE0 ec_cp_lds(&L->ecsave[1], &L->ec);
               FOR_LANES(j, N) { X_save2[j] = X[j]; Y_save2[j] = Y[j]; if (!last) norm_save2[j] = lbo[j]; }
               nstart_bytes = L->ecsave[0].offs;
               bytes_buf = L->packet + 1 + nstart_bytes;
               save_bytes = (int)L->ecsave[0].storage - nstart_bytes;
               FOR_LANES(j, save_bytes) journal[j] = bytes_buf[j];         /* trial-1 byte journal -> per-stream HBM scratch */
               wv_sync();
               LANE0 ec_cp_lds(&L->ec, &L->ecsave[0]);
               FOR_LANES(j, N) { X[j] = Xg[j]; Y[j] = Yg[j]; }
               wv_sync();
               if (i == start + 1) { FOR_LANES(j, hf_n2 - hf_n1) norm[hf_n1 + j] = norm[2 * hf_n1 - hf_n2 + j]; wv_sync(); }      /* (theta RDO runs without dual stereo) */
               remaining_bits = rem0; seed = seed0;
               K_TOC(21);
            }
            cfg.theta_round = rdo ? 2 * tr - 1 : 0;
            r = quant_band_stereo_wave(L, cfg, remaining_bits, seed, X, Y, N, b, B, lb, LM, lbo, lowband_scratch, cm);
            x_cm = (unsigned)wv_uni(r[0]); remaining_bits = wv_uni(r[1]); seed = (u32)wv_uni(r[2]);
            K_TOC(24);
         }
         if (rdo) {
            K_TIC();
            wv_sync();
            dist1 = mult16_32_q15(w[0], inner_prod_norm_shift_gw(Xg, X, N)) + mult16_32_q15(w[1], inner_prod_norm_shift_gw(Yg, Y, N));
            if (dist0 >= dist1) {
               x_cm = cm2; remaining_bits = rem1; seed = seed1;
               wv_sync();
               LANE0 ec_cp_lds(&L->ec, &L->ecsave[1]);
               FOR_LANES(j, N) { X[j] = X_save2[j]; Y[j] = Y_save2[j]; if (!last) lbo[j] = norm_save2[j]; }
               FOR_LANES(j, save_bytes) bytes_buf[j] = journal[j];
               wv_sync();
            }
            K_TOC(21);
         }
         y_cm = x_cm;
      }
      wv_sync();
      LANE0 { collapse_masks[i * C + 0] = (u8)x_cm; collapse_masks[i * C + C - 1] = (u8)y_cm; }
      balance += wv_uni(pulses[i]) + tell;
      update_lowband = b > (N << BITRES);
      cfg.avoid_split_noise = 0;
   }
   wv_sync();
   LANE0 L->st.rng = seed;
   wv_sync();
}
#endif

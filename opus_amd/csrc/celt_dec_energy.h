/* celt_dec_energy.h — the entropy-decoded side information of a CELT frame and the PVQ index -> pulse-vector map, decoder side.
 * Format: celt/laplace.c:94, celt/quant_bands.c:431 / :496 / :525, celt/celt_decoder.c:513 (tf_decode), celt/cwrs.c:467 (cwrsi).
 *
 * Reading symbols is serial (one range decoder), what is done with them mostly is not:
 *   coarse energy   the symbols of all (band, channel) are read first (their decoding depends only on the bit position, not on the energies), then the
 *                   prediction recurrence along the bands is run per channel;
 *   fine / final    lane 0 reads the raw bits in order, the refinements are applied one lane per (band, channel);
 *   tf              lane 0 reads the change flags, the table lookup is per band;
 *   cwrsi           the walk along the vector is serial in the running index, but each step's "how many pulses here" is a search over one ROW of the U(n,k)
 *                   table: the wave holds the row (one entry per lane), a ballot finds the count, v_readlane fetches the two entries the step subtracts, and
 *                   the next row is requested before the current one is used (the row index does not depend on the data). */
#ifndef OPUS_AMD_CELT_DEC_ENERGY_H
#define OPUS_AMD_CELT_DEC_ENERGY_H

/* inverse of oa_laplace_put (celt_enc_energy.h): the value whose interval of the Laplace model {p0, decay} contains the decoder's 15-bit position */
WV_DEV int oa_laplace_get(EC_ARGS, unsigned p0, int decay)
{
   const unsigned pos = k_ec_decode_bin(EC_PASS, 15);
   unsigned lo = 0, width = p0;
   int mag = 0, neg = 0;
   if (pos >= p0) {
      mag = 1; lo = p0;
      width = ((32768u - 32u - p0) * (u32)(16384 - decay) >> 15) + 1;            /* mass of +1 (and of -1), floor of one count included */
      while (width > 1 && pos >= lo + 2 * width) { lo += 2 * width; width = ((2 * width - 2) * (u32)decay >> 15) + 1; mag++; }
      if (width <= 1) { const unsigned extra = (pos - lo) >> 1; mag += (int)extra; lo += 2 * extra; }     /* flat floor region: one count per value and sign */
      if (pos < lo + width) neg = 1; else lo += width;
   }
   k_ec_dec_update(EC_PASS, lo, imin(lo + width, 32768), 32768);
   return neg ? -mag : mag;
}

/* unquant_coarse_energy (quant_bands.c:431), lane 0 */
WV_DEV void coarse_energy_read_l0(int start, int end, WV_LDS i32 *oldE, int intra, EC_ARGS, int C, int LM, WV_LDS i32 *q /* 2 * 21 words of LDS scratch */)
{
   const u8 *model = ct_e_prob_model[LM][intra];
   const i32 budget = (i32)e->storage * 8;
   for (int i = start; i < end; i++) {
      const int m = 2 * imin(i, 20);
      for (int c = 0; c < C; c++) {
         const i32 room = budget - k_ec_tell(EC_PASS);
         int v;
         if (room >= 15) v = oa_laplace_get(EC_PASS, (unsigned)model[m] << 7, (int)model[m + 1] << 6);
         else if (room >= 2) { const int t = k_ec_dec_icdf(EC_PASS, k_tiny_energy_icdf, 2); v = t & 1 ? -((t + 1) >> 1) : t >> 1; }
         else if (room >= 1) v = -k_ec_dec_bit_logp(EC_PASS, 1);
         else v = -1;
         q[i + c * OA_NB_EBANDS] = v;
      }
   }
   const i16 pred = intra ? 0 : k_inter_pred[LM], leak = intra ? 4915 : k_inter_leak[LM];
   for (int c = 0; c < C; c++) {
      long long carry = 0;
      for (int i = start; i < end; i++) {
         const int w = i + c * OA_NB_EBANDS;
         const i32 step = shl32(q[w], DB_SHIFT);
         const i32 v = (i32)(mult16_32_q15(pred, imax(-GC(9.f), oldE[w])) + carry + step);
         oldE[w] = imin(GC(28.f), imax(-GC(28.f), v));
         carry += step - mult16_32_q15(leak, step);
      }
   }
}

/* unquant_fine_energy (quant_bands.c:496): the coder is taken from / returned to *ecl; sym: 2 * 21 words of scratch */
WV_DEV void fine_energy_read_wave(WV_LDS EcCtx *ecl, WV_LDS u8 *ecbuf, WV_LDS i32 *sym, WV_LDS i32 *hand, int start, int end, WV_LDS i32 *oldE, const WV_LDS i32 *fine_quant, int C)
{
   LANE0 {
      EcCtx ec_; ec_ld(&ec_, ecl); EcCtx *e = &ec_; WV_LDS u8 *buf = ecbuf;
      u32 got = 0;
      for (int i = start; i < end; i++) {
         const int n = fine_quant[i];
         if (n <= 0 || k_ec_tell(EC_PASS) + C * n > (i32)e->storage * 8) continue;
         for (int c = 0; c < C; c++) sym[i + c * OA_NB_EBANDS] = (i32)k_ec_dec_bits(EC_PASS, (unsigned)n);
         got |= 1u << i;
      }
      ec_st(ecl, &ec_);
      hand[0] = (i32)got;
   }
   const u32 got = (u32)wv_uni(hand[0]);
   FOR_LANES(w, C * OA_NB_EBANDS) {
      const int c = w / OA_NB_EBANDS, i = w - c * OA_NB_EBANDS;
      if (got >> i & 1) { const int n = fine_quant[i]; oldE[w] += sub32(vshr32(2 * sym[w] + 1, n - DB_SHIFT + 1), GC(.5f)); }
   }
   wv_sync();
}

/* unquant_energy_finalise (quant_bands.c:525): lane 0 reads one bit per (band, channel) that still gets one, by priority; returns the bands served, bit c of
 * sym[band] = the bit read for channel c.  The halvings are applied by energy_finalise_apply_dec_wave. */
WV_DEV u32 energy_finalise_read_l0(int start, int end, const WV_LDS i32 *fine_quant, const WV_LDS i32 *fine_priority, int bits_left, WV_LDS i32 *sym, EC_ARGS, int C)
{
   u32 got = 0;
   for (int prio = 0; prio < 2; prio++)
      for (int i = start; i < end && bits_left >= C; i++)
         if (fine_quant[i] < OA_MAX_FINE_BITS && fine_priority[i] == prio) {
            int b = 0;
            for (int c = 0; c < C; c++) b |= (int)k_ec_dec_bits(EC_PASS, 1) << c;
            sym[i] = b; got |= 1u << i; bits_left -= C;
         }
   return got;
}
WV_DEV void energy_finalise_apply_dec_wave(u32 got, const WV_LDS i32 *sym, WV_LDS i32 *oldE, const WV_LDS i32 *fine_quant, int C)
{
   FOR_LANES(w, C * OA_NB_EBANDS) {
      const int c = w / OA_NB_EBANDS, i = w - c * OA_NB_EBANDS;
      if (got >> i & 1) { const i32 half_cell = GC(.5f) >> (fine_quant[i] + 1); oldE[w] += (sym[i] >> c & 1) ? half_cell : -half_cell; }
   }
   wv_sync();
}

/* tf_decode (celt_decoder.c:513), lane 0 */
WV_DEV void tf_read_l0(int start, int end, int isTransient, WV_LDS i32 *tf_res, int LM, EC_ARGS)
{
   u32 budget = e->storage * 8, pos = (u32)k_ec_tell(EC_PASS);
   int logp = isTransient ? 2 : 4, flag = 0, any = 0;
   const int select_rsv = LM > 0 && pos + logp + 1 <= budget;
   budget -= select_rsv;
   for (int i = start; i < end; i++) {
      if (pos + logp <= budget) { flag ^= k_ec_dec_bit_logp(EC_PASS, (unsigned)logp); pos = (u32)k_ec_tell(EC_PASS); any |= flag; }
      tf_res[i] = flag;
      logp = isTransient ? 4 : 5;
   }
   const signed char *row = k_tf_select_table[LM] + 4 * isTransient;
   int select = 0;
   if (select_rsv && row[any] != row[2 + any]) select = k_ec_dec_bit_logp(EC_PASS, 1);
   for (int i = start; i < end; i++) tf_res[i] = row[2 * select + tf_res[i]];
}

/* cwrsi (cwrs.c:467) on the wave: index -> y[0..N), returns sum y^2.  All arguments uniform.  At dimension n (N - position) with k pulses left and running index i:
 * the sign is negative iff i >= U(n, k+1) (then i -= U(n, k+1)); the pulses that remain AFTER this position are the largest k' <= k with U(n, k') <= i; this position
 * takes k - k' and i -= U(n, k').  U(n, .) is non-decreasing, so "largest k'" is the top set bit of a ballot over the row. */
WV_DEV i32 cwrsi_wave(int N, int K, u32 idx, WV_LDS i32 *y)
{
   const int lane = wv_lane();
   int k = K;
   u32 i = idx;
   i32 yy = 0;
   u32 row = lane <= k + 1 && N > 2 ? pvq_u(N, lane) : 0;                       /* row of dimension N, entries 0..63 */
   for (int n = N; n > 2; n--) {
      const u32 nxt = n - 1 > 2 && lane <= k + 1 ? pvq_u(n - 1, lane) : 0;       /* requested before this row is used; k only shrinks */
      u32 above;                                                                 /* U(n, k + 1) */
      if (k + 1 < 64) above = (u32)wv_bcast((i32)row, k + 1); else above = pvq_u(n, k + 1);
      const int neg = i >= above;
      if (neg) i -= above;
      int kk = 0; u32 below = 0;                                                 /* largest k' <= k with U(n, k') <= i, and that U */
      {
         const u64 m = wv_ballot(lane <= k && row <= i);
         kk = 63 - __builtin_clzll(m | 1);
         below = (u32)wv_bcast((i32)row, kk);
         if (k >= 64 && kk == 63) {                                              /* more than 63 pulses left and the first 64 entries all fit: continue in the next blocks */
            for (int base = 64; base <= k; base += 64) {
               const int l = base + lane;
               const u32 u = l <= k ? pvq_u(n, l) : 0xffffffffu;
               const u64 m2 = wv_ballot(l <= k && u <= i);
               if (!m2) break;
               const int top = 63 - __builtin_clzll(m2);
               kk = base + top; below = (u32)wv_bcast((i32)u, top);
               if (top != 63) break;
            }
         }
      }
      i -= below;
      const int v = neg ? kk - k : k - kk;
      if (lane == 0) y[N - n] = v;
      yy += v * v;
      k = kk;
      row = nxt;
   }
   if (N >= 2) {                                                                  /* dimensions 2 and 1 in closed form: U(2, k) = 2k - 1 (k > 0) */
      const u32 p = 2 * (u32)k + 1;
      const int neg = i >= p;
      if (neg) i -= p;
      const int kk = (int)((i + 1) >> 1);
      if (kk) i -= 2 * (u32)kk - 1;
      const int v = neg ? kk - k : k - kk, last = i ? -kk : kk;
      if (lane == 0) { y[N - 2] = v; y[N - 1] = last; }
      yy += v * v + last * last;
   }
   wv_sync();
   return yy;
}
#endif

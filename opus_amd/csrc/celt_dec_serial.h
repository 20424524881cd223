/* celt_dec_serial.h — the serial, entropy-decoded parts of the CELT frame decoder (lane-0 code): Laplace / energy
 * de-quantisation (celt/laplace.c:94, celt/quant_bands.c:431/:496/:525), tf_decode (celt/celt_decoder.c:513) and the PVQ
 * index -> pulse vector map cwrsi (celt/cwrs.c:467). */
#ifndef OPUS_AMD_CELT_DEC_SERIAL_H
#define OPUS_AMD_CELT_DEC_SERIAL_H
WV_DEV unsigned laplace_freq1(unsigned fs0, int decay) { return (32768u - 32u - fs0) * (u32)(16384 - decay) >> 15; }
WV_DEV int k_laplace_decode(EC_ARGS, unsigned fs, int decay)
{
   int val = 0;
   unsigned fl = 0, fm = k_ec_decode_bin(EC_PASS, 15);
   if (fm >= fs) {
      val++;
      fl = fs;
      fs = laplace_freq1(fs, decay) + 1;
      while (fs > 1 && fm >= fl + 2 * fs) {
         fs *= 2;
         fl += fs;
         fs = ((fs - 2 * 1) * (i32)decay) >> 15;
         fs += 1;
         val++;
      }
      if (fs <= 1) {
         int di = (fm - fl) >> (0 + 1);
         val += di;
         fl += 2 * di * 1;
      }
      if (fm < fl + fs) val = -val;
      else fl += fs;
   }
   k_ec_dec_update(EC_PASS, fl, imin(fl + fs, 32768), 32768);
   return val;
}
WV_DEV void k_unquant_coarse_energy(int start, int end, WV_LDS i32 *oldEBands, int intra, EC_ARGS, int C, int LM)
{
   const u8 *prob_model = ct_e_prob_model[LM][intra];
   long long prev[2] = {0, 0};
   i16 coef, beta;
   if (intra) { coef = 0; beta = 4915; }
   else { beta = k_inter_leak[LM]; coef = k_inter_pred[LM]; }
   i32 budget = e->storage * 8;
   for (int i = start; i < end; i++) {
      for (int c = 0; c < C; c++) {
         int qi;
         i32 tell = k_ec_tell(EC_PASS);
         if (budget - tell >= 15) {
            int pi = 2 * imin(i, 20);
            qi = k_laplace_decode(EC_PASS, prob_model[pi] << 7, prob_model[pi + 1] << 6);
         } else if (budget - tell >= 2) {
            qi = k_ec_dec_icdf(EC_PASS, k_tiny_energy_icdf, 2);
            qi = (qi >> 1) ^ -(qi & 1);
         } else if (budget - tell >= 1) qi = -k_ec_dec_bit_logp(EC_PASS, 1);
         else qi = -1;
         i32 q = shl32(qi, DB_SHIFT);
         i32 old = imax(-GC(9.f), oldEBands[i + c * OA_NB_EBANDS]);
         i32 tmp = (i32)(mult16_32_q15(coef, old) + prev[c] + q);
         tmp = imin(GC(28.f), imax(-GC(28.f), tmp));
         oldEBands[i + c * OA_NB_EBANDS] = tmp;
         prev[c] = prev[c] + q - mult16_32_q15(beta, q);
      }
   }
}
WV_DEV void k_unquant_fine_energy(int start, int end, WV_LDS i32 *oldEBands, const WV_LDS int *extra_quant, EC_ARGS, int C)
{
   for (int i = start; i < end; i++) {
      int extra = extra_quant[i];
      if (extra <= 0) continue;
      if (k_ec_tell(EC_PASS) + C * extra > (i32)e->storage * 8) continue;
      for (int c = 0; c < C; c++) {
         int q2 = k_ec_dec_bits(EC_PASS, extra);
         i32 offset = sub32(vshr32(2 * q2 + 1, extra - DB_SHIFT + 1), GC(.5f));
         oldEBands[i + c * OA_NB_EBANDS] += offset;
      }
   }
}
WV_DEV void k_unquant_energy_finalise(int start, int end, WV_LDS i32 *oldEBands, const WV_LDS int *fine_quant, const WV_LDS int *fine_priority, int bits_left, EC_ARGS, int C)
{
   for (int prio = 0; prio < 2; prio++) {
      for (int i = start; i < end && bits_left >= C; i++) {
         if (fine_quant[i] >= OA_MAX_FINE_BITS || fine_priority[i] != prio) continue;
         for (int c = 0; c < C; c++) {
            int q2 = k_ec_dec_bits(EC_PASS, 1);
            i32 offset = (shl32(q2, DB_SHIFT) - GC(.5f)) >> (fine_quant[i] + 1);
            oldEBands[i + c * OA_NB_EBANDS] += offset;
            bits_left--;
         }
      }
   }
}
WV_DEV void k_tf_decode(int start, int end, int isTransient, WV_LDS int *tf_res, int LM, EC_ARGS)
{
   int curr = 0, tf_select = 0, tf_changed = 0;
   u32 budget = e->storage * 8, tell = k_ec_tell(EC_PASS);
   int logp = isTransient ? 2 : 4;
   int tf_select_rsv = LM > 0 && tell + logp + 1 <= budget;
   budget -= tf_select_rsv;
   for (int i = start; i < end; i++) {
      if (tell + logp <= budget) {
         curr ^= k_ec_dec_bit_logp(EC_PASS, logp);
         tell = k_ec_tell(EC_PASS);
         tf_changed |= curr;
      }
      tf_res[i] = curr;
      logp = isTransient ? 4 : 5;
   }
   if (tf_select_rsv && k_tf_select_table[LM][4 * isTransient + 0 + tf_changed] != k_tf_select_table[LM][4 * isTransient + 2 + tf_changed])
      tf_select = k_ec_dec_bit_logp(EC_PASS, 1);
   for (int i = start; i < end; i++) tf_res[i] = k_tf_select_table[LM][4 * isTransient + 2 * tf_select + tf_res[i]];
}
/* cwrsi (cwrs.c:467): index -> pulse vector y[0..n), returns sum y^2.  U(a,b) through the dense table. */
WV_DEV i32 k_cwrsi(int n, int k, u32 i, WV_LDS i32 *y)
{
   u32 p;
   int s, k0;
   i16 val;
   i32 yy = 0;
   while (n > 2) {
      u32 q;
      if (k >= n) {
         p = pvq_u(n, k + 1);
         s = -(i >= p);
         i -= p & s;
         k0 = k;
         q = pvq_u(n, n);
         if (q > i) {
            k = n;
            do p = pvq_u(--k, n); while (p > i);
         } else for (p = pvq_u(n, k); p > i; p = pvq_u(n, k)) k--;
         i -= p;
         val = (i16)((k0 - k + s) ^ s);
         *y++ = val;
         yy = mac16_16(yy, val, val);
      } else {
         p = pvq_u(k, n);
         q = pvq_u(k + 1, n);
         if (p <= i && i < q) {
            i -= p;
            *y++ = 0;
         } else {
            s = -(i >= q);
            i -= q & s;
            k0 = k;
            do p = pvq_u(--k, n); while (p > i);
            i -= p;
            val = (i16)((k0 - k + s) ^ s);
            *y++ = val;
            yy = mac16_16(yy, val, val);
         }
      }
      n--;
   }
   p = 2 * k + 1;
   s = -(i >= p);
   i -= p & s;
   k0 = k;
   k = (i + 1) >> 1;
   if (k) i -= 2 * k - 1;
   val = (i16)((k0 - k + s) ^ s);
   *y++ = val;
   yy = mac16_16(yy, val, val);
   s = -(int)i;
   val = (i16)((k + s) ^ s);
   *y = val;
   yy = mac16_16(yy, val, val);
   return yy;
}
#endif

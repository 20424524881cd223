/* celt_enc_serial.h — the inherently serial (entropy-coded, budget-driven) parts of the CELT frame encoder,
 * executed by lane 0 of the frame's wavefront on LDS-resident data: Laplace/coarse/fine/final energy coding
 * (celt/quant_bands.c:142-429, celt/laplace.c:44-92) and bit allocation (celt/rate.c:249-653,
 * celt/rate.h:48-88, celt/celt.c:329).  Scratch arrays come from FrameLds::scr. */
#ifndef OPUS_AMD_CELT_ENC_SERIAL_H
#define OPUS_AMD_CELT_ENC_SERIAL_H
#define MAX_FINE_BITS 8
#define FINE_OFFSET 21
#define LOG_MAX_PSEUDO 6
#define ALLOC_STEPS 6

WV_DEV unsigned laplace_freq1(unsigned fs0, int decay)
{
   unsigned ft = 32768 - 1 * (2 * 16) - fs0;
   return ft * (i32)(16384 - decay) >> 15;
}
WV_DEV void k_laplace_encode(EC_ARGS, int *value, unsigned fs, int decay)
{
   unsigned fl = 0;
   int val = *value;
   if (val) {
      int s = -(val < 0), i;
      val = (val + s) ^ s;
      fl = fs;
      fs = laplace_freq1(fs, decay);
      for (i = 1; fs > 0 && i < val; i++) {
         fs *= 2;
         fl += fs + 2;
         fs = (fs * (i32)decay) >> 15;
      }
      if (!fs) {
         int ndi_max = (32768 - fl + 1 - 1) >> 0;
         ndi_max = (ndi_max - s) >> 1;
         int di = imin(val - i, ndi_max - 1);
         fl += (2 * di + 1 + s) * 1;
         fs = imin(1, 32768 - fl);
         *value = (i + di + s) ^ s;
      } else {
         fs += 1;
         fl += fs & ~s;
      }
   }
   k_ec_encode_bin(EC_PASS, fl, fl + fs, 15);
}
WV_DEV i32 loss_distortion(const WV_LDS i32 *eBands, const WV_LDS i32 *oldEBands, int start, int end, int len, int C)
{
   i32 dist = 0;
   for (int c = 0; c < C; c++)
      for (int i = start; i < end; i++) {
         i32 d = pshr32(sub32(eBands[i + c * len], oldEBands[i + c * len]), DB_SHIFT - 7);
         dist = mac16_16(dist, d, d);
      }
   return imin(200, dist >> 14);
}
WV_TABLE i16 k_pred_coef[4] = {29440, 26112, 21248, 16384};
WV_TABLE i16 k_beta_coef[4] = {30147, 22282, 12124, 6554};
WV_TABLE u8 k_small_energy_icdf[3] = {2, 1, 0};

WV_DEV int coarse_impl(int start, int end, const WV_LDS i32 *eBands, WV_LDS i32 *oldEBands, i32 budget, i32 tell,
      const u8 *prob_model, WV_LDS i32 *error, EC_ARGS, int C, int LM, int intra, i32 max_decay, int lfe)
{
   int badness = 0;
   i32 prev[2] = {0, 0};
   i16 coef, beta;
   if (tell + 3 <= budget) k_ec_enc_bit_logp(EC_PASS, intra, 3);
   if (intra) { coef = 0; beta = 4915; }
   else { beta = k_beta_coef[LM]; coef = k_pred_coef[LM]; }
   for (int i = start; i < end; i++) {
      for (int c = 0; c < C; c++) {
         i32 x = eBands[i + c * OA_NB_EBANDS];
         i32 oldE = imax(-GC(9.f), oldEBands[i + c * OA_NB_EBANDS]);
         i32 f = x - mult16_32_q15(coef, oldE) - prev[c];
         int qi = (f + QC32(.5f, DB_SHIFT)) >> DB_SHIFT;
         i32 decay_bound = imax(-GC(28.f), sub32(oldEBands[i + c * OA_NB_EBANDS], max_decay));
         if (qi < 0 && x < decay_bound) {
            qi += (int)(sub32(decay_bound, x) >> DB_SHIFT);
            if (qi > 0) qi = 0;
         }
         int qi0 = qi;
         tell = k_ec_tell(EC_PASS);
         int bits_left = budget - tell - 3 * C * (end - i);
         if (i != start && bits_left < 30) {
            if (bits_left < 24) qi = imin(1, qi);
            if (bits_left < 16) qi = imax(-1, qi);
         }
         if (lfe && i >= 2) qi = imin(qi, 0);
         if (budget - tell >= 15) {
            int pi = 2 * imin(i, 20);
            k_laplace_encode(EC_PASS, &qi, prob_model[pi] << 7, prob_model[pi + 1] << 6);
         } else if (budget - tell >= 2) {
            qi = imax(-1, imin(qi, 1));
            k_ec_enc_icdf(EC_PASS, 2 * qi ^ -(qi < 0), k_small_energy_icdf, 2);
         } else if (budget - tell >= 1) {
            qi = imin(0, qi);
            k_ec_enc_bit_logp(EC_PASS, -qi, 1);
         } else qi = -1;
         error[i + c * OA_NB_EBANDS] = f - shl32(qi, DB_SHIFT);
         badness += iabs(qi0 - qi);
         i32 q = shl32(qi, DB_SHIFT);
         i32 tmp = mult16_32_q15(coef, oldE) + prev[c] + q;
         tmp = imax(-GC(28.f), tmp);
         oldEBands[i + c * OA_NB_EBANDS] = tmp;
         prev[c] = prev[c] + q - mult16_32_q15(beta, q);
      }
   }
   return lfe ? 0 : badness;
}
WV_DEV void k_quant_coarse_energy(WV_LDS i32 *scr, WV_LDS u8 *intra_bits, int start, int end, int effEnd, const WV_LDS i32 *eBands, WV_LDS i32 *oldEBands, u32 budget,
      WV_LDS i32 *error, EC_ARGS, int C, int LM, int nbAvailableBytes, int force_intra, WV_LDS i32 *delayedIntra,
      int two_pass, int loss_rate, int lfe)
{
   WV_LDS i32 *oldEBands_intra = scr, *error_intra = scr + 2 * OA_NB_EBANDS;
   int badness1 = 0;
   int intra = force_intra || (!two_pass && *delayedIntra > 2 * C * (end - start) && nbAvailableBytes > (end - start) * C);
   i32 intra_bias = (i32)((budget * *delayedIntra * loss_rate) / (C * 512));
   i32 new_distortion = loss_distortion(eBands, oldEBands, start, effEnd, OA_NB_EBANDS, C);
   u32 tell = k_ec_tell(EC_PASS);
   if (tell + 3 > budget) two_pass = intra = 0;
   i32 max_decay = GC(16.f);
   if (end - start > 10) max_decay = shl32(imin(max_decay >> (DB_SHIFT - 3), nbAvailableBytes), DB_SHIFT - 3);
   if (lfe) max_decay = GC(3.f);
   EcCtx ecsave[2];
   ecsave[0] = *e;   /* enc_start */
   for (int i_ = 0; i_ < C * OA_NB_EBANDS; i_++) oldEBands_intra[i_] = oldEBands[i_];
   if (two_pass || intra)
      badness1 = coarse_impl(start, end, eBands, oldEBands_intra, budget, tell, ct_e_prob_model[LM][1],
            error_intra, EC_PASS, C, LM, 1, max_decay, lfe);
   if (!intra) {
      i32 tell_intra = k_ec_tell_frac(EC_PASS);
      ecsave[1] = *e;   /* enc_intra */
      u32 nstart = ecsave[0].offs, nintra = ecsave[1].offs;
      WV_LDS u8 *intra_buf = buf + nstart;
      for (u32 i_ = 0; i_ < nintra - nstart; i_++) intra_bits[i_] = intra_buf[i_];
      *e = ecsave[0];
      int badness2 = coarse_impl(start, end, eBands, oldEBands, budget, tell, ct_e_prob_model[LM][intra],
            error, EC_PASS, C, LM, 0, max_decay, lfe);
      if (two_pass && (badness1 < badness2 || (badness1 == badness2 && ((i32)k_ec_tell_frac(EC_PASS)) + intra_bias > tell_intra))) {
         *e = ecsave[1];
         for (u32 i_ = 0; i_ < nintra - nstart; i_++) intra_buf[i_] = intra_bits[i_];
         for (int i_ = 0; i_ < C * OA_NB_EBANDS; i_++) oldEBands[i_] = oldEBands_intra[i_];
         for (int i_ = 0; i_ < C * OA_NB_EBANDS; i_++) error[i_] = error_intra[i_];
         intra = 1;
      }
   } else {
      for (int i_ = 0; i_ < C * OA_NB_EBANDS; i_++) oldEBands[i_] = oldEBands_intra[i_];
      for (int i_ = 0; i_ < C * OA_NB_EBANDS; i_++) error[i_] = error_intra[i_];
   }
   if (intra) *delayedIntra = new_distortion;
   else *delayedIntra = add32(mult16_32_q15(mult16_16_q15(k_pred_coef[LM], k_pred_coef[LM]), *delayedIntra), new_distortion);
}
WV_DEV void k_quant_fine_energy(int start, int end, WV_LDS i32 *oldEBands, WV_LDS i32 *error, const WV_LDS int *prev_quant,
      const WV_LDS int *extra_quant, EC_ARGS, int C)
{
   for (int i = start; i < end; i++) {
      i16 extra = 1 << extra_quant[i];
      if (extra_quant[i] <= 0) continue;
      if (k_ec_tell(EC_PASS) + C * extra_quant[i] > (i32)e->storage * 8) continue;
      i16 prev = prev_quant ? prev_quant[i] : 0;
      for (int c = 0; c < C; c++) {
         int q2 = vshr32(add32(error[i + c * OA_NB_EBANDS], GC(.5f) >> prev), DB_SHIFT - extra_quant[i] - prev);
         if (q2 > extra - 1) q2 = extra - 1;
         if (q2 < 0) q2 = 0;
         k_ec_enc_bits(EC_PASS, q2, extra_quant[i]);
         i32 offset = sub32(vshr32(2 * q2 + 1, extra_quant[i] - DB_SHIFT + 1), GC(.5f));
         offset = offset >> prev;
         oldEBands[i + c * OA_NB_EBANDS] += offset;
         error[i + c * OA_NB_EBANDS] -= offset;
      }
   }
}
WV_DEV void k_quant_energy_finalise(int start, int end, WV_LDS i32 *oldEBands, WV_LDS i32 *error, const WV_LDS int *fine_quant,
      const WV_LDS int *fine_priority, int bits_left, EC_ARGS, int C)
{
   for (int prio = 0; prio < 2; prio++)
      for (int i = start; i < end && bits_left >= C; i++) {
         if (fine_quant[i] >= MAX_FINE_BITS || fine_priority[i] != prio) continue;
         for (int c = 0; c < C; c++) {
            int q2 = error[i + c * OA_NB_EBANDS] < 0 ? 0 : 1;
            k_ec_enc_bits(EC_PASS, q2, 1);
            i32 offset = (shl32(q2, DB_SHIFT) - GC(.5f)) >> (fine_quant[i] + 1);
            if (oldEBands) oldEBands[i + c * OA_NB_EBANDS] += offset;
            error[i + c * OA_NB_EBANDS] -= offset;
            bits_left--;
         }
      }
}
WV_TABLE u8 k_LOG2_FRAC_TABLE[24] = {0, 8, 13, 16, 19, 21, 23, 24, 26, 27, 28, 29, 30, 31, 32, 32, 33, 34, 34, 35, 36, 36, 37, 37};

WV_DEV int k_get_pulses(int i) { return i < 8 ? i : (8 + (i & 7)) << ((i >> 3) - 1); }
WV_DEV int k_bits2pulses(int band, int LM, int bits)
{
   LM++;
   const u8 *cache = ct_cache_bits + ct_cache_index[LM * OA_NB_EBANDS + band];
   int lo = 0, hi = cache[0];
   bits--;
   for (int i = 0; i < LOG_MAX_PSEUDO; i++) {
      int mid = (lo + hi + 1) >> 1;
      if ((int)cache[mid] >= bits) hi = mid; else lo = mid;
   }
   if (bits - (lo == 0 ? -1 : (int)cache[lo]) <= (int)cache[hi] - bits) return lo;
   return hi;
}
WV_DEV int k_pulses2bits(int band, int LM, int pulses)
{
   LM++;
   const u8 *cache = ct_cache_bits + ct_cache_index[LM * OA_NB_EBANDS + band];
   return pulses == 0 ? 0 : cache[pulses] + 1;
}
WV_DEV void k_init_caps(WV_LDS int *cap, int LM, int C)
{
   for (int i = 0; i < OA_NB_EBANDS; i++) {
      int N = (ct_eBands[i + 1] - ct_eBands[i]) << LM;
      cap[i] = (ct_cache_caps[OA_NB_EBANDS * (2 * LM + C - 1) + i] + 64) * C * N >> 2;
   }
}
WV_DEV int interp_bits2pulses(int start, int end, int skip_start, const WV_LDS int *bits1, const WV_LDS int *bits2,
      const WV_LDS int *thresh, const WV_LDS int *cap, i32 total, WV_LDS i32 *_balance, int skip_rsv, WV_LDS int *intensity,
      int intensity_rsv, WV_LDS int *dual_stereo, int dual_stereo_rsv, WV_LDS int *bits, WV_LDS int *ebits, WV_LDS int *fine_priority,
      int C, int LM, EC_ARGS, int encode, int prev, int signalBandwidth)
{
   const int16_t *eB = ct_eBands;
   i32 psum;
   int lo, hi, i, j, codedBands = -1, done;
   int alloc_floor = C << BITRES, stereo = C > 1, logM = LM << BITRES;
   i32 left, percoeff, balance;
   lo = 0; hi = 1 << ALLOC_STEPS;
   for (i = 0; i < ALLOC_STEPS; i++) {
      int mid = (lo + hi) >> 1;
      psum = 0; done = 0;
      for (j = end; j-- > start;) {
         int tmp = bits1[j] + (mid * (i32)bits2[j] >> ALLOC_STEPS);
         if (tmp >= thresh[j] || done) { done = 1; psum += imin(tmp, cap[j]); }
         else if (tmp >= alloc_floor) psum += alloc_floor;
      }
      if (psum > total) hi = mid; else lo = mid;
   }
   psum = 0; done = 0;
   for (j = end; j-- > start;) {
      int tmp = bits1[j] + ((i32)lo * bits2[j] >> ALLOC_STEPS);
      if (tmp < thresh[j] && !done) tmp = tmp >= alloc_floor ? alloc_floor : 0;
      else done = 1;
      tmp = imin(tmp, cap[j]);
      bits[j] = tmp;
      psum += tmp;
   }
   for (codedBands = end;; codedBands--) {
      int band_width, band_bits, rem;
      j = codedBands - 1;
      if (j <= skip_start) { total += skip_rsv; break; }
      left = total - psum;
      percoeff = (u32)left / (u32)(eB[codedBands] - eB[start]);
      left -= (eB[codedBands] - eB[start]) * percoeff;
      rem = imax(left - (eB[j] - eB[start]), 0);
      band_width = eB[codedBands] - eB[j];
      band_bits = (int)(bits[j] + percoeff * band_width + rem);
      if (band_bits >= imax(thresh[j], alloc_floor + (1 << BITRES))) {
         if (encode) {
            int depth_threshold = codedBands > 17 ? (j < prev ? 7 : 9) : 0;
            if (codedBands <= start + 2 || (band_bits > (depth_threshold * band_width << LM << BITRES) >> 4 && j <= signalBandwidth)) {
               k_ec_enc_bit_logp(EC_PASS, 1, 1);
               break;
            }
            k_ec_enc_bit_logp(EC_PASS, 0, 1);
         } else if (k_ec_dec_bit_logp(EC_PASS, 1)) break;
         psum += 1 << BITRES;
         band_bits -= 1 << BITRES;
      }
      psum -= bits[j] + intensity_rsv;
      if (intensity_rsv > 0) intensity_rsv = k_LOG2_FRAC_TABLE[j - start];
      psum += intensity_rsv;
      if (band_bits >= alloc_floor) { psum += alloc_floor; bits[j] = alloc_floor; }
      else bits[j] = 0;
   }
   if (intensity_rsv > 0) {
      if (encode) {
         *intensity = imin(*intensity, codedBands);
         k_ec_enc_uint(EC_PASS, *intensity - start, codedBands + 1 - start);
      } else *intensity = start + k_ec_dec_uint(EC_PASS, codedBands + 1 - start);
   } else *intensity = 0;
   if (*intensity <= start) { total += dual_stereo_rsv; dual_stereo_rsv = 0; }
   if (dual_stereo_rsv > 0) {
      if (encode) k_ec_enc_bit_logp(EC_PASS, *dual_stereo, 1);
      else *dual_stereo = k_ec_dec_bit_logp(EC_PASS, 1);
   } else *dual_stereo = 0;

   left = total - psum;
   percoeff = (u32)left / (u32)(eB[codedBands] - eB[start]);
   left -= (eB[codedBands] - eB[start]) * percoeff;
   for (j = start; j < codedBands; j++) bits[j] += ((int)percoeff * (eB[j + 1] - eB[j]));
   for (j = start; j < codedBands; j++) {
      int tmp = (int)imin(left, eB[j + 1] - eB[j]);
      bits[j] += tmp;
      left -= tmp;
   }
   balance = 0;
   for (j = start; j < codedBands; j++) {
      int N0 = eB[j + 1] - eB[j], N = N0 << LM, den, offset, NClogN;
      i32 excess, bit = (i32)bits[j] + balance;
      if (N > 1) {
         excess = imax(bit - cap[j], 0);
         bits[j] = bit - excess;
         den = (C * N + ((C == 2 && N > 2 && !*dual_stereo && j < *intensity) ? 1 : 0));
         NClogN = den * (ct_logN[j] + logM);
         offset = (NClogN >> 1) - den * FINE_OFFSET;
         if (N == 2) offset += den << BITRES >> 2;
         if (bits[j] + offset < den * 2 << BITRES) offset += NClogN >> 2;
         else if (bits[j] + offset < den * 3 << BITRES) offset += NClogN >> 3;
         ebits[j] = imax(0, (bits[j] + offset + (den << (BITRES - 1))));
         ebits[j] = ((u32)ebits[j] / (u32)den) >> BITRES;
         if (C * ebits[j] > (bits[j] >> BITRES)) ebits[j] = bits[j] >> stereo >> BITRES;
         ebits[j] = imin(ebits[j], MAX_FINE_BITS);
         fine_priority[j] = ebits[j] * (den << BITRES) >= bits[j] + offset;
         bits[j] -= C * ebits[j] << BITRES;
      } else {
         excess = imax(0, bit - (C << BITRES));
         bits[j] = bit - excess;
         ebits[j] = 0;
         fine_priority[j] = 1;
      }
      if (excess > 0) {
         int extra_fine = imin(excess >> (stereo + BITRES), MAX_FINE_BITS - ebits[j]);
         ebits[j] += extra_fine;
         int extra_bits = extra_fine * C << BITRES;
         fine_priority[j] = extra_bits >= excess - balance;
         excess -= extra_bits;
      }
      balance = excess;
   }
   *_balance = balance;
   for (; j < end; j++) {
      ebits[j] = bits[j] >> stereo >> BITRES;
      bits[j] = 0;
      fine_priority[j] = ebits[j] < 1;
   }
   return codedBands;
}
WV_DEV int k_compute_allocation(WV_LDS i32 *scr, int start, int end, const WV_LDS int *offsets, const WV_LDS int *cap, int alloc_trim,
      WV_LDS int *intensity, WV_LDS int *dual_stereo, i32 total, WV_LDS i32 *balance, WV_LDS int *pulses, WV_LDS int *ebits,
      WV_LDS int *fine_priority, int C, int LM, EC_ARGS, int encode, int prev, int signalBandwidth)
{
   const int16_t *eB = ct_eBands;
   WV_LDS int *bits1 = scr, *bits2 = scr + 21, *thresh = scr + 42, *trim_offset = scr + 63;
   int lo, hi, len = OA_NB_EBANDS, j, skip_start = start, skip_rsv, intensity_rsv = 0, dual_stereo_rsv = 0;
   total = imax(total, 0);
   skip_rsv = total >= 1 << BITRES ? 1 << BITRES : 0;
   total -= skip_rsv;
   if (C == 2) {
      intensity_rsv = k_LOG2_FRAC_TABLE[end - start];
      if (intensity_rsv > total) intensity_rsv = 0;
      else {
         total -= intensity_rsv;
         dual_stereo_rsv = total >= 1 << BITRES ? 1 << BITRES : 0;
         total -= dual_stereo_rsv;
      }
   }
   for (j = start; j < end; j++) {
      thresh[j] = imax((C) << BITRES, (3 * (eB[j + 1] - eB[j]) << LM << BITRES) >> 4);
      trim_offset[j] = C * (eB[j + 1] - eB[j]) * (alloc_trim - 5 - LM) * (end - j - 1) * (1 << (LM + BITRES)) >> 6;
      if ((eB[j + 1] - eB[j]) << LM == 1) trim_offset[j] -= C << BITRES;
   }
   lo = 1; hi = 11 - 1;
   do {
      int done = 0, psum = 0, mid = (lo + hi) >> 1;
      for (j = end; j-- > start;) {
         int N = eB[j + 1] - eB[j];
         int bitsj = C * N * ct_allocVectors[mid * len + j] << LM >> 2;
         if (bitsj > 0) bitsj = imax(0, bitsj + trim_offset[j]);
         bitsj += offsets[j];
         if (bitsj >= thresh[j] || done) { done = 1; psum += imin(bitsj, cap[j]); }
         else if (bitsj >= C << BITRES) psum += C << BITRES;
      }
      if (psum > total) hi = mid - 1; else lo = mid + 1;
   } while (lo <= hi);
   hi = lo--;
   for (j = start; j < end; j++) {
      int N = eB[j + 1] - eB[j];
      int bits1j = C * N * ct_allocVectors[lo * len + j] << LM >> 2;
      int bits2j = hi >= 11 ? cap[j] : C * N * ct_allocVectors[hi * len + j] << LM >> 2;
      if (bits1j > 0) bits1j = imax(0, bits1j + trim_offset[j]);
      if (bits2j > 0) bits2j = imax(0, bits2j + trim_offset[j]);
      if (lo > 0) bits1j += offsets[j];
      bits2j += offsets[j];
      if (offsets[j] > 0) skip_start = j;
      bits2j = imax(0, bits2j - bits1j);
      bits1[j] = bits1j;
      bits2[j] = bits2j;
   }
   return interp_bits2pulses(start, end, skip_start, bits1, bits2, thresh, cap, total, balance, skip_rsv,
         intensity, intensity_rsv, dual_stereo, dual_stereo_rsv, pulses, ebits, fine_priority, C, LM, EC_PASS,
         encode, prev, signalBandwidth);
}
#endif

/* oc_rate.c — CELT bit allocation. Oracle restatement of celt/rate.c:249-533 (interp_bits2pulses),
 * :535-653 (clt_compute_allocation), celt/rate.h:48-88 (get_pulses/bits2pulses/pulses2bits),
 * celt/celt.c:329 (init_caps). Pure integer, 1/8-bit units. */
#include "oc_celt.h"

static const u8 LOG2_FRAC_TABLE[24] = {0, 8, 13, 16, 19, 21, 23, 24, 26, 27, 28, 29, 30, 31, 32, 32, 33, 34, 34, 35, 36, 36, 37, 37};

int oc_get_pulses(int i) { return i < 8 ? i : (8 + (i & 7)) << ((i >> 3) - 1); }

int oc_bits2pulses(int band, int LM, int bits)
{
   LM++;
   const u8 *cache = oc_cache_bits + oc_cache_index[LM * NB_EBANDS + band];
   int lo = 0, hi = cache[0];
   bits--;
   for (int i = 0; i < LOG_MAX_PSEUDO; i++) {
      int mid = (lo + hi + 1) >> 1;
      if ((int)cache[mid] >= bits) hi = mid; else lo = mid;
   }
   if (bits - (lo == 0 ? -1 : (int)cache[lo]) <= (int)cache[hi] - bits) return lo;
   return hi;
}
int oc_pulses2bits(int band, int LM, int pulses)
{
   LM++;
   const u8 *cache = oc_cache_bits + oc_cache_index[LM * NB_EBANDS + band];
   return pulses == 0 ? 0 : cache[pulses] + 1;
}
void oc_init_caps(int *cap, int LM, int C)
{
   for (int i = 0; i < NB_EBANDS; i++) {
      int N = (oc_eBands[i + 1] - oc_eBands[i]) << LM;
      cap[i] = (oc_cache_caps[NB_EBANDS * (2 * LM + C - 1) + i] + 64) * C * N >> 2;
   }
}

static int interp_bits2pulses(int start, int end, int skip_start, const int *bits1, const int *bits2,
      const int *thresh, const int *cap, i32 total, i32 *_balance, int skip_rsv, int *intensity,
      int intensity_rsv, int *dual_stereo, int dual_stereo_rsv, int *bits, int *ebits, int *fine_priority,
      int C, int LM, oc_ec *ec, int encode, int prev, int signalBandwidth)
{
   const int16_t *eB = oc_eBands;
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
               oc_ec_enc_bit_logp(ec, 1, 1);
               break;
            }
            oc_ec_enc_bit_logp(ec, 0, 1);
         } else if (oc_ec_dec_bit_logp(ec, 1)) break;
         psum += 1 << BITRES;
         band_bits -= 1 << BITRES;
      }
      psum -= bits[j] + intensity_rsv;
      if (intensity_rsv > 0) intensity_rsv = LOG2_FRAC_TABLE[j - start];
      psum += intensity_rsv;
      if (band_bits >= alloc_floor) { psum += alloc_floor; bits[j] = alloc_floor; }
      else bits[j] = 0;
   }
   if (intensity_rsv > 0) {
      if (encode) {
         *intensity = imin(*intensity, codedBands);
         oc_ec_enc_uint(ec, *intensity - start, codedBands + 1 - start);
      } else *intensity = start + oc_ec_dec_uint(ec, codedBands + 1 - start);
   } else *intensity = 0;
   if (*intensity <= start) { total += dual_stereo_rsv; dual_stereo_rsv = 0; }
   if (dual_stereo_rsv > 0) {
      if (encode) oc_ec_enc_bit_logp(ec, *dual_stereo, 1);
      else *dual_stereo = oc_ec_dec_bit_logp(ec, 1);
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
         NClogN = den * (oc_logN[j] + logM);
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

int oc_compute_allocation(int start, int end, const int *offsets, const int *cap, int alloc_trim,
      int *intensity, int *dual_stereo, i32 total, i32 *balance, int *pulses, int *ebits,
      int *fine_priority, int C, int LM, oc_ec *ec, int encode, int prev, int signalBandwidth)
{
   const int16_t *eB = oc_eBands;
   int bits1[NB_EBANDS], bits2[NB_EBANDS], thresh[NB_EBANDS], trim_offset[NB_EBANDS];
   int lo, hi, len = NB_EBANDS, j, skip_start = start, skip_rsv, intensity_rsv = 0, dual_stereo_rsv = 0;
   total = imax(total, 0);
   skip_rsv = total >= 1 << BITRES ? 1 << BITRES : 0;
   total -= skip_rsv;
   if (C == 2) {
      intensity_rsv = LOG2_FRAC_TABLE[end - start];
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
         int bitsj = C * N * oc_allocVectors[mid * len + j] << LM >> 2;
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
      int bits1j = C * N * oc_allocVectors[lo * len + j] << LM >> 2;
      int bits2j = hi >= 11 ? cap[j] : C * N * oc_allocVectors[hi * len + j] << LM >> 2;
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
         intensity, intensity_rsv, dual_stereo, dual_stereo_rsv, pulses, ebits, fine_priority, C, LM, ec,
         encode, prev, signalBandwidth);
}

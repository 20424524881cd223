/* oc_energy.c — band log-energy quantisation (coarse Laplace-coded, fine, final).
 * Oracle restatement of celt/laplace.c:44-92 and celt/quant_bands.c:142-429,553-572. */
#include "oc_celt.h"
#include <stdlib.h>

/* ec_laplace_get_freq1, laplace.c:44 (LAPLACE_MINP=1, LAPLACE_NMIN=16) */
static unsigned laplace_freq1(unsigned fs0, int decay)
{
   unsigned ft = 32768 - 1 * (2 * 16) - fs0;
   return ft * (i32)(16384 - decay) >> 15;
}
/* ec_laplace_encode, laplace.c:51 */
void oc_laplace_encode(oc_ec *enc, int *value, unsigned fs, int decay)
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
   oc_ec_encode_bin(enc, fl, fl + fs, 15);
}

/* amp2Log2, quant_bands.c:553 */
void oc_amp2log2(int effEnd, int end, const i32 *bandE, i32 *bandLogE, int C)
{
   for (int c = 0; c < C; c++) {
      for (int i = 0; i < effEnd; i++) {
         bandLogE[i + c * NB_EBANDS] = oc_log2_db(bandE[i + c * NB_EBANDS]) - shl32((i32)oc_eMeans[i], DB_SHIFT - 4);
         bandLogE[i + c * NB_EBANDS] += GC(2.f);
      }
      for (int i = effEnd; i < end; i++) bandLogE[c * NB_EBANDS + i] = -GC(14.f);
   }
}

/* loss_distortion, quant_bands.c:142 */
static i32 loss_distortion(const i32 *eBands, const i32 *oldEBands, int start, int end, int len, int C)
{
   i32 dist = 0;
   for (int c = 0; c < C; c++)
      for (int i = start; i < end; i++) {
         i32 d = pshr32(sub32(eBands[i + c * len], oldEBands[i + c * len]), DB_SHIFT - 7);
         dist = mac16_16(dist, d, d);
      }
   return imin(200, dist >> 14);
}

static const i16 pred_coef[4] = {29440, 26112, 21248, 16384};
static const i16 beta_coef[4] = {30147, 22282, 12124, 6554};
static const i16 beta_intra = 4915;
static const u8 small_energy_icdf[3] = {2, 1, 0};

/* quant_coarse_energy_impl, quant_bands.c:156 */
static int coarse_impl(int start, int end, const i32 *eBands, i32 *oldEBands, i32 budget, i32 tell,
      const u8 *prob_model, i32 *error, oc_ec *enc, int C, int LM, int intra, i32 max_decay, int lfe)
{
   int badness = 0;
   i32 prev[2] = {0, 0};
   i16 coef, beta;
   if (tell + 3 <= budget) oc_ec_enc_bit_logp(enc, intra, 3);
   if (intra) { coef = 0; beta = beta_intra; }
   else { beta = beta_coef[LM]; coef = pred_coef[LM]; }
   for (int i = start; i < end; i++) {
      for (int c = 0; c < C; c++) {
         i32 x = eBands[i + c * NB_EBANDS];
         i32 oldE = imax(-GC(9.f), oldEBands[i + c * NB_EBANDS]);
         i32 f = x - mult16_32_q15(coef, oldE) - prev[c];
         int qi = (f + QC32(.5f, DB_SHIFT)) >> DB_SHIFT;
         i32 decay_bound = imax(-GC(28.f), sub32(oldEBands[i + c * NB_EBANDS], max_decay));
         if (qi < 0 && x < decay_bound) {
            qi += (int)(sub32(decay_bound, x) >> DB_SHIFT);
            if (qi > 0) qi = 0;
         }
         int qi0 = qi;
         tell = oc_ec_tell(enc);
         int bits_left = budget - tell - 3 * C * (end - i);
         if (i != start && bits_left < 30) {
            if (bits_left < 24) qi = imin(1, qi);
            if (bits_left < 16) qi = imax(-1, qi);
         }
         if (lfe && i >= 2) qi = imin(qi, 0);
         if (budget - tell >= 15) {
            int pi = 2 * imin(i, 20);
            oc_laplace_encode(enc, &qi, prob_model[pi] << 7, prob_model[pi + 1] << 6);
         } else if (budget - tell >= 2) {
            qi = imax(-1, imin(qi, 1));
            oc_ec_enc_icdf(enc, 2 * qi ^ -(qi < 0), small_energy_icdf, 2);
         } else if (budget - tell >= 1) {
            qi = imin(0, qi);
            oc_ec_enc_bit_logp(enc, -qi, 1);
         } else qi = -1;
         error[i + c * NB_EBANDS] = f - shl32(qi, DB_SHIFT);
         badness += abs(qi0 - qi);
         i32 q = shl32(qi, DB_SHIFT);
         i32 tmp = mult16_32_q15(coef, oldE) + prev[c] + q;
         tmp = imax(-GC(28.f), tmp);
         oldEBands[i + c * NB_EBANDS] = tmp;
         prev[c] = prev[c] + q - mult16_32_q15(beta, q);
      }
   }
   return lfe ? 0 : badness;
}

/* quant_coarse_energy, quant_bands.c:260 */
void oc_quant_coarse_energy(int start, int end, int effEnd, const i32 *eBands, i32 *oldEBands, u32 budget,
      i32 *error, oc_ec *enc, int C, int LM, int nbAvailableBytes, int force_intra, i32 *delayedIntra,
      int two_pass, int loss_rate, int lfe)
{
   i32 oldEBands_intra[2 * NB_EBANDS], error_intra[2 * NB_EBANDS];
   int badness1 = 0;
   int intra = force_intra || (!two_pass && *delayedIntra > 2 * C * (end - start) && nbAvailableBytes > (end - start) * C);
   i32 intra_bias = (i32)((budget * *delayedIntra * loss_rate) / (C * 512));
   i32 new_distortion = loss_distortion(eBands, oldEBands, start, effEnd, NB_EBANDS, C);
   u32 tell = oc_ec_tell(enc);
   if (tell + 3 > budget) two_pass = intra = 0;
   i32 max_decay = GC(16.f);
   if (end - start > 10) max_decay = shl32(imin(max_decay >> (DB_SHIFT - 3), nbAvailableBytes), DB_SHIFT - 3);
   if (lfe) max_decay = GC(3.f);
   oc_ec enc_start = *enc;
   memcpy(oldEBands_intra, oldEBands, sizeof(i32) * C * NB_EBANDS);
   if (two_pass || intra)
      badness1 = coarse_impl(start, end, eBands, oldEBands_intra, budget, tell, oc_e_prob_model[LM][1],
            error_intra, enc, C, LM, 1, max_decay, lfe);
   if (!intra) {
      u8 intra_bits[1280];
      i32 tell_intra = oc_ec_tell_frac(enc);
      oc_ec enc_intra = *enc;
      u32 nstart = enc_start.offs, nintra = enc_intra.offs;
      u8 *intra_buf = enc_intra.buf + nstart;
      memcpy(intra_bits, intra_buf, nintra - nstart);
      *enc = enc_start;
      int badness2 = coarse_impl(start, end, eBands, oldEBands, budget, tell, oc_e_prob_model[LM][intra],
            error, enc, C, LM, 0, max_decay, lfe);
      if (two_pass && (badness1 < badness2 || (badness1 == badness2 && ((i32)oc_ec_tell_frac(enc)) + intra_bias > tell_intra))) {
         *enc = enc_intra;
         memcpy(intra_buf, intra_bits, nintra - nstart);
         memcpy(oldEBands, oldEBands_intra, sizeof(i32) * C * NB_EBANDS);
         memcpy(error, error_intra, sizeof(i32) * C * NB_EBANDS);
         intra = 1;
      }
   } else {
      memcpy(oldEBands, oldEBands_intra, sizeof(i32) * C * NB_EBANDS);
      memcpy(error, error_intra, sizeof(i32) * C * NB_EBANDS);
   }
   if (intra) *delayedIntra = new_distortion;
   else *delayedIntra = add32(mult16_32_q15(mult16_16_q15(pred_coef[LM], pred_coef[LM]), *delayedIntra), new_distortion);
}

/* quant_fine_energy, quant_bands.c:360 */
void oc_quant_fine_energy(int start, int end, i32 *oldEBands, i32 *error, const int *prev_quant,
      const int *extra_quant, oc_ec *enc, int C)
{
   for (int i = start; i < end; i++) {
      i16 extra = 1 << extra_quant[i];
      if (extra_quant[i] <= 0) continue;
      if (oc_ec_tell(enc) + C * extra_quant[i] > (i32)enc->storage * 8) continue;
      i16 prev = prev_quant ? prev_quant[i] : 0;
      for (int c = 0; c < C; c++) {
         int q2 = vshr32(add32(error[i + c * NB_EBANDS], GC(.5f) >> prev), DB_SHIFT - extra_quant[i] - prev);
         if (q2 > extra - 1) q2 = extra - 1;
         if (q2 < 0) q2 = 0;
         oc_ec_enc_bits(enc, q2, extra_quant[i]);
         i32 offset = sub32(vshr32(2 * q2 + 1, extra_quant[i] - DB_SHIFT + 1), GC(.5f));
         offset = offset >> prev;
         oldEBands[i + c * NB_EBANDS] += offset;
         error[i + c * NB_EBANDS] -= offset;
      }
   }
}

/* quant_energy_finalise, quant_bands.c:401 */
void oc_quant_energy_finalise(int start, int end, i32 *oldEBands, i32 *error, const int *fine_quant,
      const int *fine_priority, int bits_left, oc_ec *enc, int C)
{
   for (int prio = 0; prio < 2; prio++)
      for (int i = start; i < end && bits_left >= C; i++) {
         if (fine_quant[i] >= MAX_FINE_BITS || fine_priority[i] != prio) continue;
         for (int c = 0; c < C; c++) {
            int q2 = error[i + c * NB_EBANDS] < 0 ? 0 : 1;
            oc_ec_enc_bits(enc, q2, 1);
            i32 offset = (shl32(q2, DB_SHIFT) - GC(.5f)) >> (fine_quant[i] + 1);
            if (oldEBands) oldEBands[i + c * NB_EBANDS] += offset;
            error[i + c * NB_EBANDS] -= offset;
            bits_left--;
         }
      }
}

/* ---------------- decoder side ---------------- */
/* ec_laplace_decode, laplace.c:94 */
int oc_laplace_decode(oc_ec *dec, unsigned fs, int decay)
{
   int val = 0;
   unsigned fl = 0, fm = oc_ec_decode_bin(dec, 15);
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
   oc_ec_dec_update(dec, fl, imin(fl + fs, 32768), 32768);
   return val;
}
/* unquant_coarse_energy, quant_bands.c:431 */
void oc_unquant_coarse_energy(int start, int end, i32 *oldEBands, int intra, oc_ec *dec, int C, int LM)
{
   const u8 *prob_model = oc_e_prob_model[LM][intra];
   long long prev[2] = {0, 0};
   i16 coef, beta;
   if (intra) { coef = 0; beta = beta_intra; }
   else { beta = beta_coef[LM]; coef = pred_coef[LM]; }
   i32 budget = dec->storage * 8;
   for (int i = start; i < end; i++) {
      for (int c = 0; c < C; c++) {
         int qi;
         i32 tell = oc_ec_tell(dec);
         if (budget - tell >= 15) {
            int pi = 2 * imin(i, 20);
            qi = oc_laplace_decode(dec, prob_model[pi] << 7, prob_model[pi + 1] << 6);
         } else if (budget - tell >= 2) {
            qi = oc_ec_dec_icdf(dec, small_energy_icdf, 2);
            qi = (qi >> 1) ^ -(qi & 1);
         } else if (budget - tell >= 1) qi = -oc_ec_dec_bit_logp(dec, 1);
         else qi = -1;
         i32 q = shl32(qi, DB_SHIFT);
         oldEBands[i + c * NB_EBANDS] = imax(-GC(9.f), oldEBands[i + c * NB_EBANDS]);
         i32 tmp = (i32)(mult16_32_q15(coef, oldEBands[i + c * NB_EBANDS]) + prev[c] + q);
         tmp = imin(GC(28.f), imax(-GC(28.f), tmp));
         oldEBands[i + c * NB_EBANDS] = tmp;
         prev[c] = prev[c] + q - mult16_32_q15(beta, q);
      }
   }
}
/* unquant_fine_energy, quant_bands.c:496 (prev_quant == NULL: first and only refinement stage without QEXT) */
void oc_unquant_fine_energy(int start, int end, i32 *oldEBands, const int *extra_quant, oc_ec *dec, int C)
{
   for (int i = start; i < end; i++) {
      int extra = extra_quant[i];
      if (extra <= 0) continue;
      if (oc_ec_tell(dec) + C * extra > (i32)dec->storage * 8) continue;
      for (int c = 0; c < C; c++) {
         int q2 = oc_ec_dec_bits(dec, extra);
         i32 offset = sub32(vshr32(2 * q2 + 1, extra - DB_SHIFT + 1), GC(.5f));
         oldEBands[i + c * NB_EBANDS] += offset;
      }
   }
}
/* unquant_energy_finalise, quant_bands.c:525 */
void oc_unquant_energy_finalise(int start, int end, i32 *oldEBands, const int *fine_quant, const int *fine_priority, int bits_left, oc_ec *dec, int C)
{
   for (int prio = 0; prio < 2; prio++) {
      for (int i = start; i < end && bits_left >= C; i++) {
         if (fine_quant[i] >= MAX_FINE_BITS || fine_priority[i] != prio) continue;
         for (int c = 0; c < C; c++) {
            int q2 = oc_ec_dec_bits(dec, 1);
            i32 offset = (shl32(q2, DB_SHIFT) - GC(.5f)) >> (fine_quant[i] + 1);
            oldEBands[i + c * NB_EBANDS] += offset;
            bits_left--;
         }
      }
   }
}

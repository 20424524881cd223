/* oc_testhooks.c — by-value wrappers so tests can drive the oracle exactly like ref_expose/x_misc.c
 * drives the reference (TEST INFRASTRUCTURE). */
#include "oc_celt.h"

int oc_hook_compute_allocation(int start, int end, const int *offsets, const int *cap, int alloc_trim, int *intensity,
      int *dual_stereo, i32 total, i32 *balance, int *pulses, int *ebits, int *fine_priority,
      int C, int LM, u8 *buf, int nbytes, int prev, int signalBandwidth, u32 *rng_out)
{
   oc_ec enc; oc_ec_enc_init(&enc, buf, nbytes);
   int cb = oc_compute_allocation(start, end, offsets, cap, alloc_trim, intensity, dual_stereo, total, balance,
         pulses, ebits, fine_priority, C, LM, &enc, 1, prev, signalBandwidth);
   *rng_out = enc.rng; oc_ec_enc_done(&enc);
   return cb;
}
void oc_hook_quant_energy(int start, int end, int effEnd, const i32 *eBands, i32 *oldEBands, u32 budget,
      i32 *error, int C, int LM, int nbAvailableBytes, int force_intra, i32 *delayedIntra, int two_pass,
      int loss_rate, int lfe, const int *fine_quant, const int *fine_priority, int bits_left,
      u8 *buf, int nbytes, u32 *rng_out)
{
   oc_ec enc; oc_ec_enc_init(&enc, buf, nbytes);
   oc_quant_coarse_energy(start, end, effEnd, eBands, oldEBands, budget, error, &enc, C, LM, nbAvailableBytes,
         force_intra, delayedIntra, two_pass, loss_rate, lfe);
   oc_quant_fine_energy(start, end, oldEBands, error, 0, fine_quant, &enc, C);
   oc_quant_energy_finalise(start, end, oldEBands, error, fine_quant, fine_priority, bits_left, &enc, C);
   *rng_out = enc.rng; oc_ec_enc_done(&enc);
}
int oc_hook_ec_script(const int *ops, int nops, u8 *buf, int nbytes, u32 *tells)
{
   oc_ec enc; oc_ec_enc_init(&enc, buf, nbytes);
   static const u8 icdf[4] = {200, 100, 30, 0};
   for (int i = 0; i < nops; i++) {
      const int *o = ops + 4 * i;
      switch (o[0]) {
      case 0: oc_ec_encode(&enc, o[1], o[2], o[3]); break;
      case 1: oc_ec_enc_bit_logp(&enc, o[1], o[2]); break;
      case 2: oc_ec_enc_icdf(&enc, o[1], icdf, 8); break;
      case 3: oc_ec_enc_uint(&enc, o[1], o[2]); break;
      case 4: oc_ec_enc_bits(&enc, o[1], o[2]); break;
      case 5: { int v = o[1]; oc_laplace_encode(&enc, &v, o[2], o[3]); break; }
      case 6: oc_ec_encode_bin(&enc, o[1], o[2], o[3]); break;
      }
      tells[i] = oc_ec_tell_frac(&enc);
   }
   oc_ec_enc_done(&enc);
   return enc.error ? -1 : (int)enc.offs;
}

int oc_hook_band_pipeline(const i32 *freq, int C, int LM, int shortBlocks, int spread, int dual_stereo, int intensity,
      int *tf_res, int nbytes, int complexity, int alloc_trim, u32 *seed, int disable_inv, int end,
      i32 *X_out, i32 *bandE_out, u8 *collapse_masks, u8 *buf, u32 *rng_out, int *pulses_out)
{
   int Mm = 1 << LM, N = Mm * 120;
   i32 X[2 * 960], bandE[42], balance;
   int cap[21], offsets[21] = {0}, pulses[21], fine_quant[21], fine_priority[21];
   oc_ec enc; oc_ec_enc_init(&enc, buf, nbytes);
   oc_compute_band_energies(freq, bandE, end, C, LM);
   oc_normalise_bands(freq, X, bandE, end, C, Mm);
   oc_init_caps(cap, LM, C);
   i32 bits = ((i32)nbytes * 8 << BITRES) - (i32)oc_ec_tell_frac(&enc) - 1;
   int codedBands = oc_compute_allocation(0, end, offsets, cap, alloc_trim, &intensity, &dual_stereo, bits, &balance,
         pulses, fine_quant, fine_priority, C, LM, &enc, 1, 0, end - 1);
   oc_quant_all_bands(1, 0, end, X, C == 2 ? X + N : 0, collapse_masks, bandE, pulses, shortBlocks, spread, dual_stereo,
         intensity, tf_res, nbytes * (8 << BITRES), balance, &enc, LM, codedBands, seed, complexity, disable_inv);
   *rng_out = enc.rng;
   oc_ec_enc_done(&enc);
   memcpy(X_out, X, sizeof(i32) * C * N);
   memcpy(bandE_out, bandE, sizeof(bandE));
   memcpy(pulses_out, pulses, sizeof(pulses));
   return codedBands;
}
unsigned oc_hook_alg_quant(i32 *X, int N, int K, int spread, int B, i32 gain, int resynth, u8 *buf, u32 *rng_out)
{
   oc_ec enc; oc_ec_enc_init(&enc, buf, 1275);
   unsigned cm = oc_alg_quant(X, N, K, spread, B, &enc, gain, resynth);
   *rng_out = enc.rng; oc_ec_enc_done(&enc);
   return cm;
}

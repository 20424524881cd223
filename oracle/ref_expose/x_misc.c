/* ref_expose/x_misc.c — TEST INFRASTRUCTURE: thin wrappers over reference entry points that need the
 * mode object or are header-inline (celt/mathops.h), so python can call them by value. */
#include "modes.h"
#include "mdct.h"
#include "mathops.h"
#include "rate.h"
#include "bands.h"
#include "celt.h"
#include "entenc.h"
#include "laplace.h"
#include "quant_bands.h"
#include "pitch.h"
#include "vq.h"

static const CELTMode *M(void) { int err; return opus_custom_mode_create(48000, 960, &err); }

void ref_mdct_forward(opus_int32 *in, opus_int32 *out, int shift, int stride)
{ const CELTMode *m = M(); clt_mdct_forward_c(&m->mdct, in, out, m->window, m->overlap, shift, stride, 0); }
void ref_mdct_backward(opus_int32 *in, opus_int32 *out, int shift, int stride)
{ const CELTMode *m = M(); clt_mdct_backward_c(&m->mdct, in, out, m->window, m->overlap, shift, stride, 0); }
int ref_log2(opus_int32 x) { return celt_log2(x); }
opus_int32 ref_exp2(int x) { return celt_exp2((opus_val16)x); }
opus_int32 ref_exp2_db(opus_int32 x) { return celt_exp2_db(x); }
opus_int32 ref_exp2_db_frac(opus_int32 x) { return celt_exp2_db_frac(x); }
opus_int32 ref_log2_db(opus_int32 x) { return celt_log2_db(x); }
opus_int32 ref_atan2p_norm(opus_int32 y, opus_int32 x) { return celt_atan2p_norm(y, x); }
int ref_bits2pulses(int band, int LM, int bits) { return bits2pulses(M(), band, LM, bits); }
int ref_pulses2bits(int band, int LM, int pulses) { return pulses2bits(M(), band, LM, pulses); }
void ref_init_caps(int *cap, int LM, int C) { init_caps(M(), cap, LM, C); }

/* allocation with a private encoder; returns codedBands, writes bytes */
int ref_compute_allocation(int start, int end, const int *offsets, const int *cap, int alloc_trim, int *intensity,
      int *dual_stereo, opus_int32 total, opus_int32 *balance, int *pulses, int *ebits, int *fine_priority,
      int C, int LM, unsigned char *buf, int nbytes, int prev, int signalBandwidth, opus_uint32 *rng_out)
{
   ec_enc enc; ec_enc_init(&enc, buf, nbytes);
   int cb = clt_compute_allocation(M(), start, end, offsets, cap, alloc_trim, intensity, dual_stereo, total, balance,
         pulses, ebits, fine_priority, C, LM, &enc, 1, prev, signalBandwidth);
   *rng_out = enc.rng; ec_enc_done(&enc);
   return cb;
}

/* coarse+fine+final energy quantisation with a private encoder */
void ref_quant_energy(int start, int end, int effEnd, const opus_int32 *eBands, opus_int32 *oldEBands, opus_uint32 budget,
      opus_int32 *error, int C, int LM, int nbAvailableBytes, int force_intra, opus_int32 *delayedIntra, int two_pass,
      int loss_rate, int lfe, const int *fine_quant, const int *fine_priority, int bits_left,
      unsigned char *buf, int nbytes, opus_uint32 *rng_out)
{
   ec_enc enc; ec_enc_init(&enc, buf, nbytes);
   quant_coarse_energy(M(), start, end, effEnd, eBands, oldEBands, budget, error, &enc, C, LM, nbAvailableBytes,
         force_intra, delayedIntra, two_pass, loss_rate, lfe);
   quant_fine_energy(M(), start, end, oldEBands, error, NULL, (int *)fine_quant, &enc, C);
   quant_energy_finalise(M(), start, end, oldEBands, error, (int *)fine_quant, (int *)fine_priority, bits_left, &enc, C);
   *rng_out = enc.rng; ec_enc_done(&enc);
}
void ref_amp2log2(int effEnd, int end, opus_int32 *bandE, opus_int32 *bandLogE, int C) { amp2Log2(M(), effEnd, end, bandE, bandLogE, C); }

/* a scripted range-coder exercise: ops[i] = {kind, a, b, c} */
int ref_ec_script(const int *ops, int nops, unsigned char *buf, int nbytes, opus_uint32 *tells)
{
   ec_enc enc; ec_enc_init(&enc, buf, nbytes);
   static const unsigned char icdf[4] = {200, 100, 30, 0};
   for (int i = 0; i < nops; i++) {
      const int *o = ops + 4 * i;
      switch (o[0]) {
      case 0: ec_encode(&enc, o[1], o[2], o[3]); break;
      case 1: ec_enc_bit_logp(&enc, o[1], o[2]); break;
      case 2: ec_enc_icdf(&enc, o[1], icdf, 8); break;
      case 3: ec_enc_uint(&enc, o[1], o[2]); break;
      case 4: ec_enc_bits(&enc, o[1], o[2]); break;
      case 5: { int v = o[1]; ec_laplace_encode(&enc, &v, o[2], o[3]); break; }
      case 6: ec_encode_bin(&enc, o[1], o[2], o[3]); break;
      }
      tells[i] = ec_tell_frac(&enc);
   }
   ec_enc_done(&enc);
   return enc.error ? -1 : (int)enc.offs;
}

/* band pipeline: energies -> normalise -> allocation -> quant_all_bands(encode) with a private encoder */
int ref_band_pipeline(const opus_int32 *freq, int C, int LM, int shortBlocks, int spread, int dual_stereo, int intensity,
      int *tf_res, int nbytes, int complexity, int alloc_trim, opus_uint32 *seed, int disable_inv, int end,
      opus_int32 *X_out, opus_int32 *bandE_out, unsigned char *collapse_masks, unsigned char *buf, opus_uint32 *rng_out,
      int *pulses_out)
{
   const CELTMode *m = M();
   int Mm = 1 << LM, N = Mm * 120;
   opus_int32 X[2 * 960];
   opus_int32 bandE[42];
   int cap[21], offsets[21] = {0}, pulses[21], fine_quant[21], fine_priority[21];
   opus_int32 balance;
   ec_enc enc; ec_enc_init(&enc, buf, nbytes);
   compute_band_energies(m, freq, bandE, end, C, LM, 0);
   normalise_bands(m, freq, X, bandE, end, C, Mm);
   init_caps(m, cap, LM, C);
   opus_int32 bits = ((opus_int32)nbytes * 8 << BITRES) - (opus_int32)ec_tell_frac(&enc) - 1;
   int codedBands = clt_compute_allocation(m, 0, end, offsets, cap, alloc_trim, &intensity, &dual_stereo, bits, &balance,
         pulses, fine_quant, fine_priority, C, LM, &enc, 1, 0, end - 1);
   quant_all_bands(1, m, 0, end, X, C == 2 ? X + N : NULL, collapse_masks, bandE, pulses, shortBlocks, spread, dual_stereo,
         intensity, tf_res, nbytes * (8 << BITRES), balance, &enc, LM, codedBands, seed, complexity, 0, disable_inv);
   *rng_out = enc.rng;
   ec_enc_done(&enc);
   memcpy(X_out, X, sizeof(opus_int32) * C * N);
   memcpy(bandE_out, bandE, sizeof(bandE));
   memcpy(pulses_out, pulses, sizeof(pulses));
   return codedBands;
}
unsigned ref_alg_quant(opus_int32 *X, int N, int K, int spread, int B, opus_int32 gain, int resynth, unsigned char *buf, opus_uint32 *rng_out)
{
   ec_enc enc; ec_enc_init(&enc, buf, 1275);
   unsigned cm = alg_quant(X, N, K, spread, B, &enc, gain, resynth, 0);
   *rng_out = enc.rng; ec_enc_done(&enc);
   return cm;
}

/* oc_celt.h — declarations of the CPU oracle (fixed-point CELT restatement).
 * TEST INFRASTRUCTURE ONLY (see oc_arith.h). */
#ifndef OC_CELT_H
#define OC_CELT_H
#include "oc_arith.h"
#include "oc_tables.h"

#define NB_EBANDS 21
#define OVERLAP 120
#define SHORT_MDCT 120
#define MAX_LM 3
#define COMBFILTER_MAXPERIOD 1024
#define COMBFILTER_MINPERIOD 15
#define MAX_FINE_BITS 8
#define FINE_OFFSET 21
#define QTHETA_OFFSET 4
#define QTHETA_OFFSET_TWOPHASE 16
#define MAX_PSEUDO 40
#define LOG_MAX_PSEUDO 6
#define ALLOC_STEPS 6
#define SPREAD_NONE 0
#define SPREAD_LIGHT 1
#define SPREAD_NORMAL 2
#define SPREAD_AGGRESSIVE 3

/* ---- math (oc_math.c) ---- */
unsigned oc_isqrt32(u32 v);
i16 oc_rcp_norm16(i32 x);
i32 oc_rcp_norm32(i32 x);
i32 oc_rcp(i32 x);
i32 oc_frac_div32_q29(i32 a, i32 b);
i32 oc_frac_div32(i32 a, i32 b);
i16 oc_rsqrt_norm(i32 x);
i32 oc_rsqrt_norm32(i32 x);
i32 oc_sqrt(i32 x);
i32 oc_sqrt32(i32 x);
i16 oc_cos_norm(i32 x);
i32 oc_cos_norm32(i32 x);
i16 oc_log2(i32 x);
i32 oc_exp2_frac(i32 x);
i32 oc_exp2(i32 x);
i32 oc_log2_db(i32 x);
i32 oc_exp2_db_frac(i32 x);
i32 oc_exp2_db(i32 x);
i32 oc_atan_norm(i32 x);
i32 oc_atan2p_norm(i32 y, i32 x);
OC_INLINE i32 oc_div(i32 a, i32 b) { return mult32_32_q31(a, oc_rcp(b)); }  /* celt_div mathops.h:528 */

/* ---- range coder (oc_rangeenc.c) ---- */
typedef struct {
   u8 *buf; u32 storage, end_offs, end_window; int nend_bits, nbits_total;
   u32 offs, rng, val, ext; int rem, error;
} oc_ec;
void oc_ec_enc_init(oc_ec *e, u8 *buf, u32 size);
int  oc_ec_tell(const oc_ec *e);
u32  oc_ec_tell_frac(const oc_ec *e);
void oc_ec_encode(oc_ec *e, unsigned fl, unsigned fh, unsigned ft);
void oc_ec_encode_bin(oc_ec *e, unsigned fl, unsigned fh, unsigned bits);
void oc_ec_enc_bit_logp(oc_ec *e, int val, unsigned logp);
void oc_ec_enc_icdf(oc_ec *e, int s, const u8 *icdf, unsigned ftb);
void oc_ec_enc_bits(oc_ec *e, u32 fl, unsigned bits);
void oc_ec_enc_uint(oc_ec *e, u32 fl, u32 ft);
void oc_ec_enc_patch_initial_bits(oc_ec *e, unsigned val, unsigned nbits);
void oc_ec_enc_shrink(oc_ec *e, u32 size);
void oc_ec_enc_done(oc_ec *e);
/* range decoder (oc_rangedec.c) */
void oc_ec_dec_init(oc_ec *d, const u8 *buf, u32 storage);
unsigned oc_ec_decode(oc_ec *d, unsigned ft);
unsigned oc_ec_decode_bin(oc_ec *d, unsigned bits);
void oc_ec_dec_update(oc_ec *d, unsigned fl, unsigned fh, unsigned ft);
int  oc_ec_dec_bit_logp(oc_ec *d, unsigned logp);
int  oc_ec_dec_icdf(oc_ec *d, const u8 *icdf, unsigned ftb);
u32  oc_ec_dec_bits(oc_ec *d, unsigned bits);
u32  oc_ec_dec_uint(oc_ec *d, u32 ft);

/* ---- energy quantisation (oc_energy.c) ---- */
void oc_laplace_encode(oc_ec *enc, int *value, unsigned fs, int decay);
int  oc_laplace_decode(oc_ec *dec, unsigned fs, int decay);
void oc_unquant_coarse_energy(int start, int end, i32 *oldEBands, int intra, oc_ec *dec, int C, int LM);
void oc_unquant_fine_energy(int start, int end, i32 *oldEBands, const int *extra_quant, oc_ec *dec, int C);
void oc_unquant_energy_finalise(int start, int end, i32 *oldEBands, const int *fine_quant, const int *fine_priority, int bits_left, oc_ec *dec, int C);
void oc_amp2log2(int effEnd, int end, const i32 *bandE, i32 *bandLogE, int C);
void oc_quant_coarse_energy(int start, int end, int effEnd, const i32 *eBands, i32 *oldEBands, u32 budget,
      i32 *error, oc_ec *enc, int C, int LM, int nbAvailableBytes, int force_intra, i32 *delayedIntra,
      int two_pass, int loss_rate, int lfe);
void oc_quant_fine_energy(int start, int end, i32 *oldEBands, i32 *error, const int *prev_quant,
      const int *extra_quant, oc_ec *enc, int C);
void oc_quant_energy_finalise(int start, int end, i32 *oldEBands, i32 *error, const int *fine_quant,
      const int *fine_priority, int bits_left, oc_ec *enc, int C);

/* ---- PVQ index coding (oc_cwrs.c) ---- */
u32  oc_pvq_v(int n, int k);
void oc_encode_pulses(const int *y, int n, int k, oc_ec *enc);
i32  oc_decode_pulses(int *y, int n, int k, oc_ec *dec);

/* ---- bit allocation (oc_rate.c) ---- */
int oc_bits2pulses(int band, int LM, int bits);
int oc_pulses2bits(int band, int LM, int pulses);
int oc_get_pulses(int i);
void oc_init_caps(int *cap, int LM, int C);
int oc_compute_allocation(int start, int end, const int *offsets, const int *cap, int alloc_trim,
      int *intensity, int *dual_stereo, i32 total, i32 *balance, int *pulses, int *ebits,
      int *fine_priority, int C, int LM, oc_ec *ec, int encode, int prev, int signalBandwidth);

/* ---- FFT / MDCT (oc_mdct.c) ---- */
void oc_fft_impl(int nfft_idx, i32 *fout /* interleaved re,im */, int downshift);
void oc_mdct_forward(const i32 *in, i32 *out, int shift, int stride);
void oc_mdct_backward(const i32 *in, i32 *out, int shift, int stride);

/* ---- pitch (oc_pitch.c) ---- */
i32  oc_pitch_xcorr(const i16 *x, const i16 *y, i32 *xcorr, int len, int max_pitch);
void oc_pitch_downsample(i32 *x[], i16 *x_lp, int len, int C, int factor);
void oc_pitch_search(const i16 *x_lp, i16 *y, int len, int max_pitch, int *pitch);
i16  oc_remove_doubling(i16 *x, int maxperiod, int minperiod, int N, int *T0, int prev_period, i16 prev_gain);
void oc_comb_filter(i32 *y, i32 *x, int T0, int T1, int N, i16 g0, i16 g1, int tapset0, int tapset1, int overlap);
int  oc_celt_lpc(i16 *lpc, const i32 *ac, int p);
int  oc_autocorr(const i16 *x, i32 *ac, const i16 *window, int overlap, int lag, int n);
void oc_celt_fir5(i16 *x, const i16 *num, int N);

/* ---- PVQ / bands (oc_vq.c, oc_bands.c) ---- */
void oc_exp_rotation(i32 *X, int len, int dir, int stride, int K, int spread);
i32  oc_op_pvq_search(i32 *X, int *iy, int K, int N);
unsigned oc_alg_quant(i32 *X, int N, int K, int spread, int B, oc_ec *enc, i32 gain, int resynth);
void oc_renormalise_vector(i32 *X, int N, i32 gain);
unsigned oc_alg_unquant(i32 *X, int N, int K, int spread, int B, oc_ec *dec, i32 gain);
i32  oc_stereo_itheta(const i32 *X, const i32 *Y, int stereo, int N);
void oc_compute_band_energies(const i32 *X, i32 *bandE, int end, int C, int LM);
void oc_normalise_bands(const i32 *freq, i32 *X, const i32 *bandE, int end, int C, int M);
int  oc_spreading_decision(const i32 *X, int *average, int last_decision, int *hf_average,
      int *tapset_decision, int update_hf, int end, int C, int M, const int *spread_weight);
void oc_haar1(i32 *X, int N0, int stride);
void oc_quant_all_bands(int encode, int start, int end, i32 *X, i32 *Y, u8 *collapse_masks,
      const i32 *bandE, int *pulses, int shortBlocks, int spread, int dual_stereo, int intensity,
      int *tf_res, i32 total_bits, i32 balance, oc_ec *ec, int LM, int codedBands, u32 *seed,
      int complexity, int disable_inv);
int oc_bitexact_cos(int x);
int oc_bitexact_log2tan(int isin, int icos);

#endif

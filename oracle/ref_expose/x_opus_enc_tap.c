/* ref_expose/x_opus_enc_tap.c — TEST INFRASTRUCTURE: the reference's Opus encoder (src/opus_encoder.c) and CELT encoder (celt/celt_encoder.c) compiled
 * once more into this shim, #included where they lie, with taps at the calls the CELT frame encoder makes into quant_bands.c / rate.c, so a
 * hybrid-mode mismatch of the HIP encoder can be localised.  Entry points are prefixed refx_ (the unmodified library keeps the plain names). */
#define celt_encoder_get_size refx_celt_encoder_get_size
#define opus_custom_encoder_get_size refx_opus_custom_encoder_get_size
#define celt_encoder_init refx_celt_encoder_init
#define celt_preemphasis refx_celt_preemphasis
#define celt_encode_with_ec refx_celt_encode_with_ec
#define opus_custom_encoder_ctl refx_opus_custom_encoder_ctl
#define opus_encoder_get_size refx_opus_encoder_get_size
#define opus_encoder_init refx_opus_encoder_init
#define opus_encoder_create refx_opus_encoder_create
#define opus_encode_native refx_opus_encode_native
#define opus_encode refx_opus_encode
#define opus_encode24 refx_opus_encode24
#define opus_encoder_ctl refx_opus_encoder_ctl
#define opus_encoder_destroy refx_opus_encoder_destroy
#define silk_biquad_res refx_silk_biquad_res
#define downmix_int refx_downmix_int
#define downmix_int24 refx_downmix_int24
#define frame_size_select refx_frame_size_select
#define compute_stereo_width refx_compute_stereo_width
#define opus_custom_encoder_init refx_opus_custom_encoder_init
#define opus_custom_encoder_destroy refx_opus_custom_encoder_destroy
#define CELT_ENCODER_C
#include "celt.h"
#include "quant_bands.h"
#include "rate.h"
#include "bands.h"
#include "entenc.h"
typedef void (*refx_dump_fn)(const char *tag, const void *p, int nbytes);
static refx_dump_fn g_cdump;
void refx_celt_set_dump(refx_dump_fn f) { g_cdump = f; }
static int tap_coarse(int start, int end, const celt_glog *eBands, const celt_glog *oldEBands, opus_uint32 budget, int C, int LM, int nbAvail, int force_intra, int delayedIntra, int two_pass, ec_enc *enc)
{
   opus_int32 w[12 + 84]; int i, n = 0;
   if (!g_cdump) return 0;
   w[n++] = start; w[n++] = end; w[n++] = C; w[n++] = LM; w[n++] = (opus_int32)budget; w[n++] = nbAvail; w[n++] = force_intra; w[n++] = delayedIntra; w[n++] = two_pass;
   w[n++] = ec_tell(enc); w[n++] = (opus_int32)enc->rng; w[n++] = 0;
   for (i = 0; i < 42; i++) w[n++] = i < C * 21 ? eBands[i] : 0;
   for (i = 0; i < 42; i++) w[n++] = i < C * 21 ? oldEBands[i] : 0;
   g_cdump("coarse_in", w, 4 * n); return 0;
}
static int tap_alloc(int start, int end, const int *offsets, const int *cap, int alloc_trim, int intensity, int dual_stereo, opus_int32 total, int C, int LM, ec_enc *enc, int prev, int signalBandwidth)
{
   opus_int32 w[12 + 42]; int i, n = 0;
   if (!g_cdump) return 0;
   w[n++] = start; w[n++] = end; w[n++] = alloc_trim; w[n++] = intensity; w[n++] = dual_stereo; w[n++] = total; w[n++] = C; w[n++] = LM; w[n++] = ec_tell(enc); w[n++] = (opus_int32)enc->rng; w[n++] = prev; w[n++] = signalBandwidth;
   for (i = 0; i < 21; i++) w[n++] = offsets[i];
   for (i = 0; i < 21; i++) w[n++] = cap[i];
   g_cdump("alloc_in", w, 4 * n); return 0;
}
#define quant_coarse_energy(m, start, end, effEnd, eBands, oldEBands, budget, error, enc, C, LM, nbAvail, force_intra, delayedIntra, two_pass, loss_rate, lfe) \
   (tap_coarse(start, end, eBands, oldEBands, budget, C, LM, nbAvail, force_intra, *(delayedIntra), two_pass, enc), quant_coarse_energy(m, start, end, effEnd, eBands, oldEBands, budget, error, enc, C, LM, nbAvail, force_intra, delayedIntra, two_pass, loss_rate, lfe))
#define clt_compute_allocation(m, start, end, offsets, cap, alloc_trim, intensity, dual_stereo, total, balance, pulses, ebits, fine_priority, C, LM, ec, encode, prev, signalBandwidth) \
   (tap_alloc(start, end, offsets, cap, alloc_trim, *(intensity), *(dual_stereo), total, C, LM, ec, prev, signalBandwidth), clt_compute_allocation(m, start, end, offsets, cap, alloc_trim, intensity, dual_stereo, total, balance, pulses, ebits, fine_priority, C, LM, ec, encode, prev, signalBandwidth))
#include "celt_encoder.c"
#undef quant_coarse_energy
#undef clt_compute_allocation
#include "../src/opus_encoder.c"

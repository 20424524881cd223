/* ref_expose/x_silk.c — TEST INFRASTRUCTURE: calls the reference's SILK quantisers (silk/NSQ.c:76, silk/NSQ_del_dec.c:114) with a
 * minimal encoder state built from the six configuration fields they read, so python can drive them with flat arrays. */
#include "main.h"

static void mk_state(silk_encoder_state *e, const opus_int32 *cfg)
{
   memset(e, 0, sizeof *e);
   e->fs_kHz = cfg[0]; e->nb_subfr = cfg[1]; e->predictLPCOrder = cfg[2]; e->shapingLPCOrder = cfg[3];
   e->nStatesDelayedDecision = cfg[4]; e->warping_Q16 = cfg[5];
   e->subfr_length = 5 * cfg[0]; e->ltp_mem_length = 20 * cfg[0]; e->frame_length = cfg[1] * 5 * cfg[0];
   e->arch = 0;
}

/* ind[4] = signalType, quantOffsetType, NLSFInterpCoef_Q2, Seed (Seed is written back) */
void ref_silk_nsq(const opus_int32 *cfg, int del_dec, void *nsq_state, opus_int8 *ind, const opus_int16 *x16, opus_int8 *pulses,
                  const opus_int16 *PredCoef_Q12, const opus_int16 *LTPCoef_Q14, const opus_int16 *AR_Q13, const opus_int32 *HarmShapeGain_Q14,
                  const opus_int32 *Tilt_Q14, const opus_int32 *LF_shp_Q14, const opus_int32 *Gains_Q16, const opus_int32 *pitchL,
                  opus_int32 Lambda_Q10, opus_int32 LTP_scale_Q14)
{
   silk_encoder_state e; SideInfoIndices si;
   mk_state(&e, cfg);
   memset(&si, 0, sizeof si);
   si.signalType = ind[0]; si.quantOffsetType = ind[1]; si.NLSFInterpCoef_Q2 = ind[2]; si.Seed = ind[3];
   if (del_dec) silk_NSQ_del_dec_c(&e, (silk_nsq_state *)nsq_state, &si, x16, pulses, PredCoef_Q12, LTPCoef_Q14, AR_Q13, (const opus_int *)HarmShapeGain_Q14,
                                   (const opus_int *)Tilt_Q14, LF_shp_Q14, Gains_Q16, (const opus_int *)pitchL, Lambda_Q10, LTP_scale_Q14);
   else silk_NSQ_c(&e, (silk_nsq_state *)nsq_state, &si, x16, pulses, PredCoef_Q12, LTPCoef_Q14, AR_Q13, (const opus_int *)HarmShapeGain_Q14,
                   (const opus_int *)Tilt_Q14, LF_shp_Q14, Gains_Q16, (const opus_int *)pitchL, Lambda_Q10, LTP_scale_Q14);
   ind[3] = si.Seed;
}
int ref_silk_nsq_state_size(void) { return (int)sizeof(silk_nsq_state); }
void ref_silk_lpc_analysis_filter(opus_int16 *out, const opus_int16 *in, const opus_int16 *B, opus_int32 len, opus_int32 d)
{ silk_LPC_analysis_filter(out, in, B, len, d, 0); }
opus_int32 ref_silk_div32_varQ(opus_int32 a, opus_int32 b, int q) { return silk_DIV32_varQ(a, b, q); }
opus_int32 ref_silk_inverse32_varQ(opus_int32 b, int q) { return silk_INVERSE32_varQ(b, q); }
#include "resampler_structs.h"
int ref_silk_resampler_state_size(void) { return (int)sizeof(silk_resampler_state_struct); }
int ref_silk_resampler_init(void *S, opus_int32 in, opus_int32 out, int forEnc) { return silk_resampler_init((silk_resampler_state_struct *)S, in, out, forEnc); }
int ref_silk_resampler(void *S, opus_int16 *out, const opus_int16 *in, opus_int32 inLen) { return silk_resampler((silk_resampler_state_struct *)S, out, in, inLen); }

/* oc_celt_enc.h — state of the oracle's CELT encoder (mirrors celt/celt_encoder.c:63-142 field for field,
 * arrays inlined at their maximum stereo sizes). TEST INFRASTRUCTURE. */
#ifndef OC_CELT_ENC_H
#define OC_CELT_ENC_H
#include "oc_celt.h"
typedef struct {
   int channels, stream_channels, force_intra, clip, disable_pf, complexity, start, end;
   i32 bitrate;
   int vbr, constrained_vbr, loss_rate, lsb_depth, lfe, disable_inv;
   int silk_signalType, silk_offset;
   /* cleared on reset */
   u32 rng;
   int spread_decision;
   i32 delayedIntra;
   int tonal_average, lastCodedBands, hf_average, tapset_decision, prefilter_period;
   i16 prefilter_gain;
   int prefilter_tapset, consec_transient;
   i32 preemph_memE[2];
   i32 vbr_reservoir, vbr_drift, vbr_offset, vbr_count, overlap_max;
   i16 stereo_saving;
   int intensity;
   i32 spec_avg;
   i32 in_mem[2 * OVERLAP];
   i32 prefilter_mem[2 * COMBFILTER_MAXPERIOD];
   i32 oldBandE[2 * NB_EBANDS], oldLogE[2 * NB_EBANDS], oldLogE2[2 * NB_EBANDS], energyError[2 * NB_EBANDS];
} oc_celt_enc;
void oc_celt_enc_init(oc_celt_enc *st, int channels);
int oc_celt_encode_with_ec(oc_celt_enc *st, const i16 *pcm, int frame_size, u8 *compressed, int nbCompressedBytes, oc_ec *enc);
#endif

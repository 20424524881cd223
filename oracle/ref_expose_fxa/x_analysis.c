/* ref_expose_fxa/x_analysis.c — TEST INFRASTRUCTURE: the oracle of the analysis row.  Linked against libopus_ref_fxa.so (fixed-point arithmetic, float
 * API on), it runs the reference's own run_analysis (src/analysis.c:954) on int16 input the way opus_encode_native does for one frame (src/opus_encoder.c:1252,
 * :2668: downmix_int, c1 = 0, c2 = -2 = all channels) and hands back the AnalysisInfo of the frame plus the state, so that a device implementation of the analysis
 * can be checked field by field and frame by frame against the compiled reference.  Nothing here is product code. */
#include <string.h>
#include "opus_types.h"
#include "arch.h"
#include "celt.h"
#include "modes.h"
#include "analysis.h"
#include "opus_private.h"
int ref_analysis_state_size(void) { return (int)sizeof(TonalityAnalysisState); }
int ref_analysis_info_floats(float *dst, const AnalysisInfo *a)        /* flattened: valid, 9 floats, bandwidth, 19 leak boosts */
{
   int n = 0;
   dst[n++] = (float)a->valid; dst[n++] = a->tonality; dst[n++] = a->tonality_slope; dst[n++] = a->noisiness; dst[n++] = a->activity; dst[n++] = a->music_prob;
   dst[n++] = a->music_prob_min; dst[n++] = a->music_prob_max; dst[n++] = (float)a->bandwidth; dst[n++] = a->activity_probability; dst[n++] = a->max_pitch_ratio;
   for (int i = 0; i < LEAK_BANDS; i++) dst[n++] = (float)a->leak_boost[i];
   return n;
}
void ref_analysis_init(TonalityAnalysisState *st, opus_int32 Fs, int application) { memset(st, 0, sizeof(*st)); tonality_analysis_init(st, Fs); st->application = application; }
/* one 20 ms (or shorter / longer) frame of the encoder's input -> the AnalysisInfo the encoder would use for it, flattened into out[30] */
void ref_analysis_frame(TonalityAnalysisState *st, const opus_int16 *pcm, int frame_size, int channels, opus_int32 Fs, int lsb_depth, float *out)
{
   AnalysisInfo info; int err;
   const CELTMode *mode = opus_custom_mode_create(48000, 960, &err);
   memset(&info, 0, sizeof(info));
   run_analysis(st, mode, pcm, frame_size, frame_size, 0, -2, channels, Fs, lsb_depth, downmix_int, &info);
   ref_analysis_info_floats(out, &info);
}

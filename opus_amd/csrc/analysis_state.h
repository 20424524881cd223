/* analysis_state.h — persistent state of the encoder's tonality / music analysis (opus_analysis.h), plain data shared by the stream records (celt_frame.h,
 * opus_sh_state.h) and the host library. */
#ifndef OPUS_AMD_ANALYSIS_STATE_H
#define OPUS_AMD_ANALYSIS_STATE_H
#include <stdint.h>
#define AN_NB_FRAMES 8
#define AN_NB_TBANDS 18
#define AN_BUF_SIZE 720               /* 30 ms at 24 kHz */
#define AN_DETECT_SIZE 100
#define AN_LEAK_BANDS 19
#define AN_COUNT_MAX 10000
#define AN_NB_TONAL_SKIP_BANDS 9
/* AnalysisInfo (celt/celt.h:65-79) */
struct OaAnalysisInfo {
   int32_t valid;
   float tonality, tonality_slope, noisiness, activity, music_prob, music_prob_min, music_prob_max;
   int32_t bandwidth;
   float activity_probability, max_pitch_ratio;
   uint8_t leak_boost[AN_LEAK_BANDS];
   uint8_t pad;
};
/* TonalityAnalysisState (src/analysis.h:49-85) from `angle` on: all zero = reset (tonality_analysis_reset :225) */
struct OaAnalysis {
   float angle[240], d_angle[240], d2_angle[240];
   int32_t inmem[AN_BUF_SIZE];
   int32_t mem_fill;
   float prev_band_tonality[AN_NB_TBANDS];
   float prev_tonality;
   int32_t prev_bandwidth;
   float E[AN_NB_FRAMES][AN_NB_TBANDS], logE[AN_NB_FRAMES][AN_NB_TBANDS];
   float lowE[AN_NB_TBANDS], highE[AN_NB_TBANDS], meanE[AN_NB_TBANDS + 1];
   float mem[32], cmean[8], std[9];
   float Etracker, lowECount;
   int32_t E_count, count, analysis_offset, write_pos, read_pos, read_subframe;
   float hp_ener_accum;
   int32_t initialized;
   float rnn_state[32];
   int32_t downmix_state[3];
   int32_t pad;
   OaAnalysisInfo info[AN_DETECT_SIZE];
};
#endif

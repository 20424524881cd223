/* tests/emu/emu_analysis.cpp — TEST INFRASTRUCTURE: runs the tonality / music analysis of the encoder (opus_amd/csrc/opus_analysis.h, the exact device source)
 * on the CPU wave emulator, one call of run_analysis at a time, so that it can be checked field by field against the compiled reference's
 * run_analysis (oracle/ref_expose_fxa/x_analysis.c).  Never part of the product library. */
#include "wave_emu.h"
#include "celt_enc_all.h"
#include "opus_analysis.h"

struct AnJob { AnLds *W; OaAnalysis *A; const int16_t *pcm; const int32_t *apcm; int analysis_frame_size, frame_size, C, Fs, lsb_depth; int32_t *scratch; OaAnalysisInfo *out; };
static void an_entry(void *p)
{
   AnJob *j = (AnJob *)p;
   an_run_analysis_wave(j->W, j->A, j->pcm, j->apcm, j->analysis_frame_size, j->frame_size, j->C, j->Fs, j->lsb_depth, j->scratch, j->out);
}
extern "C" int emu_analysis_state_size() { return (int)sizeof(OaAnalysis); }
extern "C" int emu_analysis_lds_size() { return (int)sizeof(AnLds); }
/* out[30]: valid, 9 floats (bandwidth as a float in slot 8), 19 leak boosts -- the layout of ref_analysis_info_floats */
extern "C" void emu_analysis_frame(OaAnalysis *A, const int16_t *pcm, const int32_t *apcm, int analysis_frame_size, int frame_size, int C, int Fs, int lsb_depth, float *out)
{
   AnLds *W = (AnLds *)aligned_alloc(64, (sizeof(AnLds) + 63) & ~63);
   memset(W, 0xA5, sizeof(AnLds));
   int32_t *scratch = (int32_t *)malloc(sizeof(int32_t) * AN_SCRATCH_WORDS); memset(scratch, 0xA5, sizeof(int32_t) * AN_SCRATCH_WORDS);
   OaAnalysisInfo info; memset(&info, 0, sizeof(info));
   AnJob j = {W, A, pcm, apcm, analysis_frame_size, frame_size, C, Fs, lsb_depth, scratch, &info};
   emu_run_wave(an_entry, &j);
   int n = 0;
   out[n++] = (float)info.valid; out[n++] = info.tonality; out[n++] = info.tonality_slope; out[n++] = info.noisiness; out[n++] = info.activity; out[n++] = info.music_prob;
   out[n++] = info.music_prob_min; out[n++] = info.music_prob_max; out[n++] = (float)info.bandwidth; out[n++] = info.activity_probability; out[n++] = info.max_pitch_ratio;
   for (int i = 0; i < AN_LEAK_BANDS; i++) out[n++] = (float)info.leak_boost[i];
   free(W); free(scratch);
}

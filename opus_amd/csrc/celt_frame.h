/* celt_frame.h — data layout of the batched encoder.
 *
 * HBM: one OaStream per independent stream, contiguous (stream-major), so the 64 lanes of the wave that owns
 * the stream load/store its state with fully coalesced dword accesses.  Fields mirror what must persist
 * between frame-steps in the reference: OpusEncoder (src/opus_encoder.c:76-146, the CELT-only subset) and
 * OpusCustomEncoder (celt/celt_encoder.c:63-142) incl. in_mem / prefilter_mem / oldBandE.. arrays.
 * LDS: one FrameLds per wave; big regions are phase-aliased (pre -> spectrum, input -> folding memory). */
#ifndef OPUS_AMD_CELT_FRAME_H
#define OPUS_AMD_CELT_FRAME_H
#include <stdint.h>
#include "analysis_state.h"

#define OA_NB_EBANDS 21
#define OA_OVERLAP 120
#define OA_MAX_PERIOD 1024
#define OA_MIN_PERIOD 15
#define OA_MAX_FRAME 960
#define OA_MAX_PACKET 1276

/* per-stream configuration (set through the ctl interface; may differ between streams of a batch) */
struct OaEncConfig {
   int32_t channels;            /* 1 or 2 */
   int32_t application;         /* OPUS_APPLICATION_RESTRICTED_LOWDELAY (2051) / RESTRICTED_CELT (2053) */
   int32_t user_bitrate_bps;    /* OPUS_AUTO (-1000), OPUS_BITRATE_MAX (-1) or bits/s */
   int32_t use_vbr, vbr_constraint, complexity;
   int32_t force_channels, user_bandwidth, max_bandwidth, lsb_depth, disable_inv, packet_loss_perc;
   int32_t variable_duration;   /* OPUS_SET_EXPERT_FRAME_DURATION (0 = not set = OPUS_FRAMESIZE_ARG) */
   int32_t input_depth;         /* sample depth of the entry point of this call: 16 (opus_encode) or 24 (opus_encode24 / _float); 0 = 16 */
   int32_t lfe;                 /* OPUS_SET_LFE (multistream surround) */
   int32_t prediction_disabled; /* OPUS_SET_PREDICTION_DISABLED */
};

/* per-stream persistent state: scalars (kept in LDS while a frame is being encoded) ... */
struct OaEncScalars {
   /* Opus layer */
   int32_t stream_channels, bandwidth, auto_bandwidth, first, prev_mode, hybrid_stereo_width_Q14;
   int32_t hp_mem[4];
   uint32_t rangeFinal;
   /* CELT layer (reset region of the reference) */
   uint32_t rng;
   int32_t spread_decision, delayedIntra, tonal_average, lastCodedBands, hf_average, tapset_decision;
   int32_t prefilter_period, prefilter_gain, prefilter_tapset, consec_transient;
   int32_t preemph_memE[2];
   int32_t vbr_reservoir, vbr_drift, vbr_offset, vbr_count, overlap_max, stereo_saving, intensity, spec_avg;
   int32_t pad0[2];             /* (the SILK-capable kernel parks CELT's disable_pf / force_intra here between its CELT passes) */
   int32_t voice_ratio;         /* OpusEncoder.voice_ratio (src/opus_encoder.c:91): -1, the value of OPUS_SET_VOICE_RATIO, or what the analysis said (:1291); CELT-only kernel */
   int32_t voice_ratio_seq;     /* last OaStream.voice_ratio_seq seen: a batch ctl hands its value over through the configuration */
};
/* ... and arrays */
struct OaEncState {
   OaEncScalars s;
   int32_t oldBandE[2 * OA_NB_EBANDS], oldLogE[2 * OA_NB_EBANDS], oldLogE2[2 * OA_NB_EBANDS], energyError[2 * OA_NB_EBANDS];
   int32_t in_mem[2 * OA_OVERLAP];
   int32_t prefilter_mem[2 * OA_MAX_PERIOD];
   OaAnalysisInfo analysis;     /* CELT_SET_ANALYSIS (celt_encoder.c:109,:3114): the analysis of the frame being coded; inside CELT's reset region */
};

struct OaStream {
   OaEncConfig cfg;
   OaEncState st;
   /* tail (added after the arrays so that the offsets above stay put) */
   /* configuration, second part (host-owned like cfg: a batch ctl pushes OA_STREAM_CFG2_WORDS words from Fs on) */
   int32_t Fs;                  /* API rate: 48000 (0 = 48000), 24000, 16000, 12000, 8000 (CELT zero-stuffs up to 48 kHz, celt_encoder.c:255,:557) */
   int32_t use_dtx, energy_mask_on;
   int32_t signal_type;         /* OPUS_SET_SIGNAL: voice_est of the channel / bandwidth decisions (src/opus_encoder.c:1413) */
   int32_t use_inband_fec, user_forced_mode;   /* accepted and read back like the reference does; they do not change CELT-only coding */
   int32_t voice_ratio, voice_ratio_seq;       /* OPUS_SET_VOICE_RATIO: value and a count of the sets (the kernel adopts the value when the count moves) */
   int32_t analysis_off;        /* private: 1 = behave like a reference built with DISABLE_FLOAT_API (no tonality analysis at complexity 10) */
   int32_t cfg2_pad[3];
   /* state */
   int32_t nb_no_activity_ms_Q1, peak_signal_energy, prev_framesize;
   int32_t an_read_pos_bak, an_read_subframe_bak;   /* the analysis' read position at the start of the call (multi-frame calls rewind to it, src/opus_encoder.c:1255,:1732), -1 = the analysis did not run */
   int32_t tail_pad[3];
   int32_t energy_mask[2 * OA_NB_EBANDS];   /* surround masking of this stream (OPUS_SET_ENERGY_MASK; copied in by the multistream layer each frame) */
   OaAnalysisInfo an_info;      /* the AnalysisInfo of the call (opus_encode_native's local, src/opus_encoder.c:1206) */
   OaAnalysis an;               /* TonalityAnalysisState (src/opus_encoder.c:105) */
};
#define OA_STREAM_CFG2_WORDS 12


/* ---- decoder: per-stream persistent state (reference OpusDecoder src/opus_decoder.c:65-94, CELT-only subset, and
 * OpusCustomDecoder celt/celt_decoder.c:87-139).  The reference's linear decode_mem[ch][2048+120] (shifted by N every frame)
 * is kept as a 2048-sample ring per channel (hist, oldest sample at hist_head) plus the 120-sample IMDCT overlap tail, so a
 * frame-step appends N samples instead of moving 2x1208. */
#define OA_DEC_HISTORY 2048
struct OaDecScalars {
   int32_t channels, stream_channels, bandwidth, mode, prev_mode, frame_size, prev_redundancy, last_packet_duration;
   uint32_t rangeFinal;
   int32_t start, end, disable_inv;
   uint32_t rng;
   int32_t error, last_pitch_index, loss_duration, plc_duration, last_frame_type, skip_plc;
   int32_t postfilter_period, postfilter_period_old, postfilter_gain, postfilter_gain_old, postfilter_tapset, postfilter_tapset_old, prefilter_and_fold;
   int32_t preemph_memD[2];
   int32_t hist_head;
   int32_t Fs;                  /* API (output) rate: 48000 (0 = 48000), 24000, 16000, 12000, 8000 */
   int32_t transition_gain_Q16; /* OPUS_SET_GAIN as a Q16 multiplier, or 0: the concealed fade source of a mode transition comes out of a nested opus_decode_frame in the reference and
                                 * carries the gain already when it is mixed in (src/opus_decoder.c:391,:537), before the frame as a whole gets it (:700) -- the host applies that one */
   int32_t pad0[1];
};
/* ---- SILK decoder state (reference silk_decoder_state silk/structs.h:236-286, silk_decoder / stereo_dec_state silk/main.h, silk/structs.h:121-127),
 * flat: table pointers of the reference (NLSF codebook, iCDFs) are re-derived from fs_kHz / nb_subfr, the resampler is its nine configuration
 * words + the 90 state rows of silk_resampler.h ---- */
struct OaSilkIndices { int8_t GainsIndices[4], LTPIndex[4], NLSFIndices[17]; int8_t contourIndex, signalType, quantOffsetType, NLSFInterpCoef_Q2, PERIndex, LTP_scaleIndex, Seed; int16_t lagIndex; int16_t pad; };
struct OaSilkChannel {
   int32_t prev_gain_Q16;
   int32_t exc_Q14[320];
   int32_t sLPC_Q14_buf[16];
   int32_t lagPrev, LastGainIndex, fs_kHz, fs_API_hz, nb_subfr, frame_length, subfr_length, ltp_mem_length, LPC_order;
   int32_t first_frame_after_reset, nFramesDecoded, nFramesPerPacket, ec_prevSignalType, ec_prevLagIndex;
   int32_t VAD_flags[3], LBRR_flag, LBRR_flags[3];
   int32_t lossCnt, prevSignalType;
   int32_t rs_cfg[9], rs_rows[90];
   /* concealment (silk_PLC_struct, silk/structs.h:254-271) and comfort noise (silk_CNG_struct :274-281) */
   int32_t plc_pitchL_Q8, plc_last_frame_lost, plc_rand_seed, plc_conc_energy, plc_conc_energy_shift, plc_prevGain_Q16[2], plc_fs_kHz, plc_nb_subfr, plc_subfr_length;
   int32_t plc_randScale_Q14, plc_prevLTP_scale_Q14;
   int32_t cng_synth_state[16], cng_smth_Gain_Q16, cng_rand_seed, cng_fs_kHz;   /* (CNG_exc_buf_Q14 is cold and lives at the end of OaSilkDec) */
   int16_t outBuf[480], prevNLSF_Q15[16], plc_LTPCoef_Q14[6], plc_prevLPC_Q12[16], cng_smth_NLSF_Q15[16];
   OaSilkIndices indices;
};
struct OaSilkDec {
   OaSilkChannel ch[2];
   int32_t pred_prev_Q13[2];
   int16_t sMid[2], sSide[2];
   int32_t nChannelsAPI, nChannelsInternal, prev_decode_only_middle;
   int32_t lastInternalRate, lastChannelsInternal, pad;   /* the DecControl fields that persist for concealment (src/opus_decoder.c:424-441) */
   /* ---- everything above is staged in LDS while the SILK layer runs (OA_SILK_HOT_BYTES); below: cold, stays in HBM ---- */
   int32_t cng_exc_buf_Q14[2][320];
};
#define OA_SILK_HOT_BYTES (sizeof(OaSilkDec) - 2 * 320 * sizeof(int32_t))
struct OaDecStream {
   OaDecScalars s;
   int32_t oldBandE[2 * OA_NB_EBANDS], oldLogE[2 * OA_NB_EBANDS], oldLogE2[2 * OA_NB_EBANDS], backgroundLogE[2 * OA_NB_EBANDS];
   int32_t overlap_mem[2 * OA_OVERLAP];
   int32_t plc_lpc[2 * 24];                /* concealment LPC (int16 values), celt_decoder.c:722 */
   int32_t hist[2 * OA_DEC_HISTORY];
   OaSilkDec silk;
   int16_t trans[2 * 240], red[2 * 240];   /* call-local: 5 ms mode-transition and redundancy audio (src/opus_decoder.c:285-291) */
};
/* reset values shared by the host library and the emulator harness (opus_decoder_init src/opus_decoder.c:135-184, celt_decoder_init
 * celt/celt_decoder.c:244-264, silk_InitDecoder silk/dec_API.c:107) */
static inline void oa_dec_stream_reset(OaDecStream *st, int channels)
{
   char *p = (char *)st; for (size_t i = 0; i < sizeof(*st); i++) p[i] = 0;
   st->s.channels = st->s.stream_channels = channels;
   st->s.frame_size = 48000 / 400;
   st->s.start = 0; st->s.end = OA_NB_EBANDS; st->s.disable_inv = channels == 1;
   st->s.skip_plc = 1;
   for (int i = 0; i < 2 * OA_NB_EBANDS; i++) st->oldLogE[i] = st->oldLogE2[i] = -(28 << 24);
   for (int c = 0; c < 2; c++) {                                  /* silk_reset_decoder + silk_CNG_Reset + silk_PLC_Reset (silk/init_decoder.c:43, CNG.c:58, PLC.c:65) */
      OaSilkChannel *ch = &st->silk.ch[c];
      ch->first_frame_after_reset = 1; ch->prev_gain_Q16 = 65536;
      ch->cng_rand_seed = 3176576; ch->plc_prevGain_Q16[0] = ch->plc_prevGain_Q16[1] = 65536; ch->plc_subfr_length = 20; ch->plc_nb_subfr = 2;
   }
   st->silk.lastInternalRate = 0; st->silk.lastChannelsInternal = 0;
}
#endif

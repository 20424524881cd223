/* silk_enc_state.h — persistent state of the SILK encoder (rows a16/a18/a20/a21/a22 of SURVEY §8), flat and memcpy-able.
 *
 * Mirrors what must survive between frames in the reference: silk_encoder_state (silk/structs.h:135-231), silk_shape_state_FIX and
 * silk_encoder_state_FIX (silk/fixed/structs_FIX.h:45-70), silk_nsq_state (silk/structs.h:56-69), silk_VAD_state (:74-86),
 * silk_LP_state (:89-95), stereo_enc_state (:110-119) and silk_encoder (silk/fixed/structs_FIX.h:108-118).  Table pointers of the
 * reference (NLSF codebook, iCDFs) are re-derived from fs_kHz / nb_subfr; the resampler is its nine configuration words + 90 state rows
 * (silk_resampler.h).  One record per stream in HBM; it is staged in LDS while the stream's wave encodes a frame. */
#ifndef OPUS_AMD_SILK_ENC_STATE_H
#define OPUS_AMD_SILK_ENC_STATE_H
#include <stdint.h>
#include <stddef.h>

#define SE_MAX_FRAME 320
#define SE_LA_SHAPE_MAX 80
#define SE_X_BUF_LEN (2 * SE_MAX_FRAME + SE_LA_SHAPE_MAX)

struct OaSilkEncIndices { int8_t GainsIndices[4], LTPIndex[4], NLSFIndices[17]; int8_t contourIndex, signalType, quantOffsetType, NLSFInterpCoef_Q2, PERIndex, LTP_scaleIndex, Seed; int16_t lagIndex; int16_t pad; };

struct OaSilkNsqState {                       /* silk_nsq_state; sLPC_Q14 keeps only the 16 words that persist between subframes */
   int32_t sLTP_shp_Q14[2 * SE_MAX_FRAME];
   int32_t sLPC_Q14[16], sAR2_Q14[24];
   int32_t sLF_AR_shp_Q14, sDiff_shp_Q14, lagPrev, sLTP_buf_idx, sLTP_shp_buf_idx, rand_seed, prev_gain_Q16, rewhite_flag;
   int16_t xq[2 * SE_MAX_FRAME];
};

struct OaSilkEncChannel {
   /* sLP, sVAD */
   int32_t lp_In_LP_State[2], lp_transition_frame_no, lp_mode, lp_saved_fs_kHz;
   int32_t vad_AnaState[2], vad_AnaState1[2], vad_AnaState2[2], vad_XnrgSubfr[4], vad_NrgRatioSmth_Q8[4], vad_HPstate, vad_NL[4], vad_inv_NL[4], vad_NoiseLevelBias[4], vad_counter;
   int32_t variable_HP_smth1_Q15;
   int32_t speech_activity_Q8, allow_bandwidth_switch, LBRRprevLastGainIndex, prevSignalType, prevLag, pitch_LPC_win_length, max_pitch_lag;
   int32_t API_fs_Hz, prev_API_fs_Hz, maxInternal_fs_Hz, minInternal_fs_Hz, desiredInternal_fs_Hz, fs_kHz, nb_subfr, frame_length, subfr_length, ltp_mem_length;
   int32_t la_pitch, la_shape, shapeWinLength, TargetRate_bps, PacketSize_ms, PacketLoss_perc, frameCounter, Complexity, nStatesDelayedDecision, useInterpolatedNLSFs;
   int32_t shapingLPCOrder, predictLPCOrder, pitchEstimationComplexity, pitchEstimationLPCOrder, pitchEstimationThreshold_Q16, sum_log_gain_Q7, NLSF_MSVQ_Survivors;
   int32_t first_frame_after_reset, controlled_since_last_payload, warping_Q16, useCBR, prefillFlag;
   int32_t input_quality_bands_Q15[4], input_tilt_Q15, SNR_dB_Q7;
   int32_t VAD_flags[3], LBRR_flag, LBRR_flags[3];
   int32_t useDTX, inDTX, noSpeechCounter, useInBandFEC, LBRR_enabled, LBRR_GainIncreases;
   int32_t inputBufIx, nFramesPerPacket, nFramesEncoded, nChannelsAPI, nChannelsInternal, channelNb, ec_prevLagIndex, ec_prevSignalType;
   /* sShape + LTPCorr */
   int32_t LastGainIndex, HarmShapeGain_smth_Q16, Tilt_smth_Q16, LTPCorr_Q15;
   int32_t rs_cfg[9], rs_rows[90];
   int32_t nsq_reset_req;                      /* the quantiser state (OaSilkEncTail) starts over before its next use: silk_setup_fs (control_codec.c:241) and the side channel's
                                                * return after mid-only frames (enc_API.c:449) ask for it here, because the analysis kernel of the split path does not hold the tails */
   int32_t inbuf_reset_req;                    /* the channel's input buffer (OaSilkEnc.inbuf) is cleared before its next use: silk_init_encoder (init_encoder.c:46: the second channel's on a mono -> stereo
                                                * switch, every channel's in a prefill call) clears it with the rest of the channel, and enc_API.c:318-326 can read samples of it that nothing has written since */
   int16_t prev_NLSFq_Q15[16];
   int16_t x_buf[SE_X_BUF_LEN];
   OaSilkEncIndices indices;
};
/* what only the quantiser / entropy coder stage of a channel touches: behind both channels in the record, so that the stage in front of it (the split path's front kernel,
 * opus_sh_split.h) stages {header, ch[0 .. C-1]} in LDS and nothing else */
struct OaSilkEncTail { OaSilkNsqState nsq; int8_t pulses[SE_MAX_FRAME]; };

struct OaSilkEncStereo {                       /* stereo_enc_state */
   int32_t pred_prev_Q13[2];
   int32_t mid_side_amp_Q0[4];
   int32_t smth_width_Q14, width_prev_Q14, silent_side_len;
   int16_t sMid[2], sSide[2];
   int8_t predIx[3][2][3], mid_only_flags[3], pad;
};

struct OaSilkEnc {                             /* silk_encoder */
   OaSilkEncStereo st;
   int32_t nBitsUsedLBRR, nBitsExceeded, nChannelsAPI, nChannelsInternal, nPrevChannelsInternal, timeSinceSwitchAllowed_ms, allowBandwidthSwitch, prev_decode_only_middle;
   OaSilkEncChannel ch[2];
   int16_t inbuf[2][SE_MAX_FRAME + 2];        /* silk_encoder_state.inputBuf of the two channels (silk/structs.h:176): the call's input at the internal rate, read only by the head of a frame (stereo L/R -> M/S, VAD,
                                               * variable low-pass, the copy into x_buf) -- behind the channels, so that a kernel for which it is scratch (the split path's front kernel) does not stage it */
   OaSilkEncTail tail[2];
};

/* silk_EncControlStruct (silk/control.h:42-120): what the Opus layer hands to silk_Encode and reads back */
/* in-band FEC side stream of the packet being built (silk_encoder_state indices_LBRR / pulses_LBRR, silk/structs.h:200-203): written once per frame,
 * read once at the start of the next packet -> lives in HBM only, never staged into the wave's LDS */
struct OaSilkLbrr { int8_t pulses[2][3][320]; OaSilkEncIndices indices[2][3]; };
struct SeControl {
   int32_t nChannelsAPI, nChannelsInternal, API_sampleRate, maxInternalSampleRate, minInternalSampleRate, desiredInternalSampleRate, payloadSize_ms, bitRate;
   int32_t packetLossPercentage, complexity, useInBandFEC, LBRR_coded, useDTX, useCBR, maxBits, toMono, opusCanSwitch, reducedDependency;
   int32_t internalSampleRate, allowBandwidthSwitch, inWBmodeWithoutVariableLP, stereoWidth_Q14, switchReady, signalType, offset;
};

/* silk_InitEncoder (silk/enc_API.c:82) + silk_init_encoder (silk/init_encoder.c:46) + silk_VAD_Init (silk/VAD.c:47): shared by the host library and the
 * emulator harness.  193536 = silk_LSHIFT(silk_lin2log(60 << 16) - (16 << 7), 8), the variable high-pass start value. */
static inline void oa_silk_enc_channel_reset(OaSilkEncChannel *c)
{
   char *p = (char *)c; for (size_t i = 0; i < sizeof(*c); i++) p[i] = 0;
   c->variable_HP_smth1_Q15 = 193536;
   c->first_frame_after_reset = 1;
   c->inbuf_reset_req = 1;
   for (int b = 0; b < 4; b++) { const int bias = 50 / (b + 1) > 1 ? 50 / (b + 1) : 1; c->vad_NoiseLevelBias[b] = bias; c->vad_NL[b] = 100 * bias; c->vad_inv_NL[b] = 2147483647 / (100 * bias); c->vad_NrgRatioSmth_Q8[b] = 100 * 256; }
   c->vad_counter = 15;
}
static inline void oa_silk_enc_reset(OaSilkEnc *e)
{
   char *p = (char *)e; for (size_t i = 0; i < sizeof(*e); i++) p[i] = 0;
   oa_silk_enc_channel_reset(&e->ch[0]); oa_silk_enc_channel_reset(&e->ch[1]);
   e->nChannelsAPI = 1; e->nChannelsInternal = 1;
}
#endif

/* opus_sh_state.h — per-stream record of the SILK / hybrid capable Opus encoder ("sh" kernel), flat and memcpy-able.
 * Mirrors the persistent part of OpusEncoder (src/opus_encoder.c:76-146) for the applications that can run the SILK layer (VOIP, AUDIO,
 * RESTRICTED_SILK), the silk_EncControlStruct fields that persist between calls (silk/control.h:42-120), the SILK encoder (silk_enc_state.h) and
 * — for hybrid — the CELT encoder state of celt_frame.h. */
#ifndef OPUS_AMD_OPUS_SH_STATE_H
#define OPUS_AMD_OPUS_SH_STATE_H
#include "celt_frame.h"
#include "silk_enc_state.h"

struct OaShConfig {
   int32_t Fs, channels, application, user_bitrate_bps, use_vbr, vbr_constraint, complexity, force_channels;
   int32_t user_bandwidth, max_bandwidth, lsb_depth, disable_inv, packet_loss_perc, user_forced_mode, signal_type, use_inband_fec;
   int32_t use_dtx, variable_duration, input_depth, lfe, prediction_disabled;
   int32_t voice_ratio;                                 /* OPUS_SET_VOICE_RATIO: the value last set ... */
   int32_t energy_mask_on;
   int32_t voice_ratio_seq;                             /* ... and a count of the sets (the kernel adopts the value when the count moves) */
   int32_t analysis_off;                                /* private: 1 = behave like a reference built with DISABLE_FLOAT_API (no tonality analysis at complexity 10) */
   int32_t force_channels_seq;                          /* count of the OPUS_SET_FORCE_CHANNELS requests (see OaShScalars.mono_forced_seq) */
   int32_t reserved[6];
};
struct OaShScalars {
   int32_t stream_channels, bandwidth, auto_bandwidth, first, mode, prev_mode, prev_channels, prev_framesize;
   int32_t hybrid_stereo_width_Q14, variable_HP_smth2_Q15, prev_HB_gain, hp_mem[4];
   uint32_t rangeFinal;
   int32_t silk_bw_switch, error;
   /* silk_mode fields that persist between calls */
   int32_t sm_toMono, sm_opusCanSwitch, sm_allowBandwidthSwitch, sm_inWBmodeWithoutVariableLP, sm_stereoWidth_Q14, sm_LBRR_coded, sm_switchReady;
   /* compute_stereo_width state (StereoWidthState, src/opus_encoder.c:62-68) */
   int32_t wm_XX, wm_XY, wm_YY, wm_smoothed_width, wm_max_follower;
   int32_t nb_no_activity_ms_Q1, sm_useDTX;             /* generalised DTX counter (decide_dtx_mode, src/opus_encoder.c:1115); silk_mode.useDTX of the last frame */
   int32_t peak_signal_energy;                          /* src/opus_encoder.c:1310-1320 (activity decision of CELT-only frames, :1926) */
   int32_t nonfinal_frame;                              /* inside a repacketised multi-frame packet (:1779) */
   int32_t voice_ratio;                                 /* OpusEncoder.voice_ratio (:91): -1, the value of OPUS_SET_VOICE_RATIO, or what the analysis said (:1291) */
   int32_t voice_ratio_seq;                             /* last cfg.voice_ratio_seq seen */
   int32_t celt_mask_cleared;                           /* CELT's own energy_mask pointer sits in its reset region (celt/celt_encoder.c:123): a CELT reset inside a call (mode
                                                         * transition, :2479) drops it until the next OPUS_SET_ENERGY_MASK, while the Opus layer's copy (SILK's rate offset) stays */
   int32_t mono_forced_seq;                             /* a multi-frame call that starts during a stereo -> mono transition sets the encoder's force_channels to 1 and never puts it
                                                         * back (src/opus_encoder.c:1764-1766): the stream stays mono until the application sets OPUS_SET_FORCE_CHANNELS again.  Kept
                                                         * as state next to the configuration: in force while it equals cfg.force_channels_seq + 1 */
};
#define OA_SH_MAX_DELAY 480                              /* encoder_buffer = Fs / 100 samples per channel */
struct OaShStream {
   OaShConfig cfg;
   OaShScalars s;
   OaSilkEnc silk;
   OaEncState celt;                                      /* hybrid only */
   int16_t delay_buffer[2 * OA_SH_MAX_DELAY];            /* hybrid only */
   OaSilkLbrr lbrr;                                      /* in-band FEC only */
   int32_t energy_mask[2 * OA_NB_EBANDS];                /* surround masking (OPUS_SET_ENERGY_MASK; copied in by the multistream layer each frame) */
   OaAnalysisInfo an_info;                               /* the AnalysisInfo of the call (opus_encode_native's local, src/opus_encoder.c:1206) */
   int32_t an_read_pos_bak, an_read_subframe_bak;         /* the analysis' read position at the start of the call (multi-frame calls rewind to it, :1255,:1732), -1 = the analysis did not run */
   OaAnalysis an;                                        /* TonalityAnalysisState (src/opus_encoder.c:105) */
};
/* opus_encoder_init (src/opus_encoder.c:204-330) */
static inline void oa_sh_stream_reset(OaShStream *st, int32_t Fs, int channels, int application)
{
   OaShConfig keep = st->cfg;
   const OaShScalars was = st->s;
   char *p = (char *)st; for (size_t i = 0; i < sizeof(*st); i++) p[i] = 0;
   st->cfg = keep;
   /* what sits in front of OPUS_ENCODER_RESET_START in the reference's OpusEncoder (src/opus_encoder.c:76-111) survives OPUS_RESET_STATE: voice_ratio, force_channels, and
    * the whole silk_mode control structure -- including what the last silk_Encode() left in it (allowBandwidthSwitch, inWBmodeWithoutVariableLP, stereoWidth_Q14, ...), which
    * the Opus layer reads again before SILK next runs */
   st->s.voice_ratio = was.voice_ratio; st->s.voice_ratio_seq = was.voice_ratio_seq; st->s.mono_forced_seq = was.mono_forced_seq;
   st->s.sm_toMono = was.sm_toMono; st->s.sm_opusCanSwitch = was.sm_opusCanSwitch; st->s.sm_allowBandwidthSwitch = was.sm_allowBandwidthSwitch;
   st->s.sm_inWBmodeWithoutVariableLP = was.sm_inWBmodeWithoutVariableLP; st->s.sm_stereoWidth_Q14 = was.sm_stereoWidth_Q14; st->s.sm_LBRR_coded = was.sm_LBRR_coded;
   st->s.sm_switchReady = was.sm_switchReady; st->s.sm_useDTX = was.sm_useDTX;
   st->cfg.Fs = Fs; st->cfg.channels = channels; st->cfg.application = application;
   st->s.stream_channels = channels; st->s.first = 1; st->s.mode = 1001; st->s.bandwidth = 1105;
   st->s.hybrid_stereo_width_Q14 = 1 << 14; st->s.prev_HB_gain = 32767;
   st->s.variable_HP_smth2_Q15 = 193536;                 /* silk_LSHIFT(silk_lin2log(VARIABLE_HP_MIN_CUTOFF_HZ), 8) = 756 << 8 */
   oa_silk_enc_reset(&st->silk);
   st->celt.s.spread_decision = 2; st->celt.s.delayedIntra = 1; st->celt.s.tonal_average = 256;
   for (int i = 0; i < 2 * OA_NB_EBANDS; i++) st->celt.oldLogE[i] = st->celt.oldLogE2[i] = -(28 << 24);
}
static inline void oa_sh_stream_init(OaShStream *st, int32_t Fs, int channels, int application)
{
   char *p = (char *)st; for (size_t i = 0; i < sizeof(*st); i++) p[i] = 0;
   st->cfg.user_bitrate_bps = -1000; st->cfg.use_vbr = 1; st->cfg.vbr_constraint = 1; st->cfg.complexity = 9; st->cfg.force_channels = -1000;
   st->cfg.user_bandwidth = -1000; st->cfg.max_bandwidth = 1105; st->cfg.lsb_depth = 24; st->cfg.user_forced_mode = -1000; st->cfg.signal_type = -1000;
   oa_sh_stream_reset(st, Fs, channels, application);
   st->cfg.voice_ratio = -1; st->s.voice_ratio = -1;
}
#endif

/* oc_opus_enc.h — Opus-level encoder of the oracle, CELT-only applications (restricted-lowdelay / restricted-celt),
 * 48 kHz, one 2.5/5/10/20 ms frame per packet.  TEST INFRASTRUCTURE. */
#ifndef OC_OPUS_ENC_H
#define OC_OPUS_ENC_H
#include "oc_celt_enc.h"
#define OC_APPLICATION_RESTRICTED_LOWDELAY 2051
#define OC_APPLICATION_RESTRICTED_CELT 2053
#define OC_AUTO (-1000)
#define OC_BITRATE_MAX (-1)
#define OC_BANDWIDTH_NARROWBAND 1101
#define OC_BANDWIDTH_MEDIUMBAND 1102
#define OC_BANDWIDTH_WIDEBAND 1103
#define OC_BANDWIDTH_SUPERWIDEBAND 1104
#define OC_BANDWIDTH_FULLBAND 1105
typedef struct {
   int Fs, channels, application;
   i32 user_bitrate_bps, bitrate_bps;
   int use_vbr, vbr_constraint, complexity, force_channels, user_bandwidth, max_bandwidth, lsb_depth, packet_loss_perc;
   int stream_channels, bandwidth, auto_bandwidth, first, prev_mode, prev_channels, prev_framesize;
   int hybrid_stereo_width_Q14, stereoWidth_Q14;
   i32 hp_mem[4];
   u32 rangeFinal;
   oc_celt_enc celt;
} oc_opus_enc;
int oc_opus_enc_size(void);
int oc_opus_enc_init(oc_opus_enc *st, int Fs, int channels, int application);
int oc_opus_enc_set(oc_opus_enc *st, int what, int value);   /* what: 0 bitrate,1 complexity,2 vbr,3 vbr_constraint,4 force_channels,5 bandwidth,6 max_bandwidth,7 lsb_depth,8 phase-inv disabled */
int oc_opus_encode(oc_opus_enc *st, const i16 *pcm, int frame_size, u8 *data, int max_data_bytes);
u32 oc_opus_enc_final_range(const oc_opus_enc *st);
#endif

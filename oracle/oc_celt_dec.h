/* oc_celt_dec.h — state of the oracle's CELT decoder (mirrors celt/celt_decoder.c:87-139, arrays inlined at their
 * stereo sizes; DECODE_BUFFER_SIZE 2048) and of the Opus-level decoder for CELT-only packets.  TEST INFRASTRUCTURE. */
#ifndef OC_CELT_DEC_H
#define OC_CELT_DEC_H
#include "oc_celt.h"
#define OC_DECODE_BUFFER_SIZE 2048
typedef struct {
   int channels, stream_channels, start, end, disable_inv;
   /* cleared on reset */
   u32 rng;
   int error, last_pitch_index, loss_duration, plc_duration, last_frame_type, skip_plc;
   int postfilter_period, postfilter_period_old;
   i16 postfilter_gain, postfilter_gain_old;
   int postfilter_tapset, postfilter_tapset_old, prefilter_and_fold;
   i32 preemph_memD[2];
   i32 decode_mem[2][OC_DECODE_BUFFER_SIZE + OVERLAP];
   i32 oldBandE[2 * NB_EBANDS], oldLogE[2 * NB_EBANDS], oldLogE2[2 * NB_EBANDS], backgroundLogE[2 * NB_EBANDS];
   i16 lpc[2 * 24];
} oc_celt_dec;
void oc_celt_dec_init(oc_celt_dec *st, int channels);
void oc_celt_dec_reset(oc_celt_dec *st);
int oc_celt_decode_with_ec(oc_celt_dec *st, const u8 *data, int len, i16 *pcm, int frame_size, oc_ec *dec);

typedef struct {
   int Fs, channels, stream_channels, bandwidth, mode, prev_mode, frame_size, prev_redundancy, last_packet_duration;
   u32 rangeFinal;
   oc_celt_dec celt;
} oc_opus_dec;
int oc_opus_dec_size(void);
int oc_opus_dec_init(oc_opus_dec *st, int Fs, int channels);
/* returns samples per channel or a negative OPUS_* code; -5 (UNIMPLEMENTED-like) for SILK/hybrid packets and FEC; data == NULL / len == 0 and
 * frames of <= 1 byte run the CELT packet-loss concealment */
int oc_opus_decode(oc_opus_dec *st, const u8 *data, int len, i16 *pcm, int frame_size, int decode_fec);
u32 oc_opus_dec_final_range(const oc_opus_dec *st);
int oc_opus_packet_parse(const u8 *data, int len, u8 *out_toc, i16 size[48], int *payload_offset);
#endif

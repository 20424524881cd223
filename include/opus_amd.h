/* opus_amd.h — C ABI of the MI355X-native batched Opus encode / decode path (drop-in boundary).
 *
 * Two layers, both plain C (no torch / HIP types in the signatures):
 *
 * 1. The classic libopus encoder entry points, same names, argument meaning and error codes as the reference
 *    (reference/include/opus.h: opus_encoder_get_size :174, opus_encoder_create :211, opus_encoder_init :231,
 *    opus_encode :266, opus_encoder_destroy :354, opus_encoder_ctl :367; error codes opus_defines.h:46-60).
 *    The OpusEncoder blob is flat host memory holding the complete canonical state (memcpy-able, no device
 *    handles: opus.h:108-109); every opus_encode() runs the frame on the GPU as a batch of one.
 *    Scope: every application (RESTRICTED_LOWDELAY / RESTRICTED_CELT on the CELT-only kernel; VOIP / AUDIO / RESTRICTED_SILK on the SILK-capable kernel), every API
 *    rate (8-48 kHz), 2.5-120 ms (calls above 20 ms become multi-frame packets, src/opus_encoder.c:1698-1838), mode / bandwidth / channel decisions as the reference's
 *    opus_encode_native (src/opus_encoder.c:1310-1700) or pinned with OPUS_SET_FORCE_MODE, mode and SILK-bandwidth switches with their CELT redundancy frames and
 *    SILK / CELT prefills, VBR / constrained VBR / hard CBR (code-3 padding), OPUS_SET_DTX (SILK's own DTX and the generalised decision, OPUS_GET_IN_DTX),
 *    OPUS_SET_INBAND_FEC + OPUS_SET_PACKET_LOSS_PERC (decide_fec, LBRR), LFE / energy-mask / prediction / expert-frame-duration controls.  No legal argument set
 *    returns OPUS_UNIMPLEMENTED.
 *
 * 2. The batch API (additive): S independent streams stepped together, one wavefront per
 *    (stream, frame); state lives in HBM between calls; import/export honours the memcpy contract.
 *
 * Results are bit-identical to the reference's fixed-point build with its float API on (FIXED_POINT, the default otherwise: the fixed-point codec plus the tonality /
 * music analysis of src/analysis.c at complexity 10; opus_encode_float / opus_encode24 / opus_decode_float / opus_decode24 present and converting as that build does):
 * same packet bytes, same OPUS_GET_FINAL_RANGE, same decoded PCM.  OPUS_AMD_SET_FLOAT_ANALYSIS(0) gives the packets of a build with DISABLE_FLOAT_API instead. */
#ifndef OPUS_AMD_H
#define OPUS_AMD_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

#define OPUS_AMD_EXPORT __attribute__((visibility("default")))

/* error codes: reference/include/opus_defines.h:46-60 */
#define OPUS_OK 0
#define OPUS_BAD_ARG -1
#define OPUS_BUFFER_TOO_SMALL -2
#define OPUS_INTERNAL_ERROR -3
#define OPUS_INVALID_PACKET -4
#define OPUS_UNIMPLEMENTED -5
#define OPUS_INVALID_STATE -6
#define OPUS_ALLOC_FAIL -7
/* applications / special values: opus_defines.h:205-232 */
#define OPUS_AUTO -1000
#define OPUS_BITRATE_MAX -1
#define OPUS_APPLICATION_VOIP 2048
#define OPUS_APPLICATION_AUDIO 2049
#define OPUS_APPLICATION_RESTRICTED_LOWDELAY 2051
#define OPUS_APPLICATION_RESTRICTED_SILK 2052
#define OPUS_APPLICATION_RESTRICTED_CELT 2053
#define OPUS_BANDWIDTH_NARROWBAND 1101
#define OPUS_BANDWIDTH_MEDIUMBAND 1102
#define OPUS_BANDWIDTH_WIDEBAND 1103
#define OPUS_BANDWIDTH_SUPERWIDEBAND 1104
#define OPUS_BANDWIDTH_FULLBAND 1105
/* CTL request numbers: opus_defines.h:130-181 */
#define OPUS_SET_APPLICATION_REQUEST 4000
#define OPUS_GET_APPLICATION_REQUEST 4001
#define OPUS_SET_BITRATE_REQUEST 4002
#define OPUS_GET_BITRATE_REQUEST 4003
#define OPUS_SET_MAX_BANDWIDTH_REQUEST 4004
#define OPUS_GET_MAX_BANDWIDTH_REQUEST 4005
#define OPUS_SET_VBR_REQUEST 4006
#define OPUS_GET_VBR_REQUEST 4007
#define OPUS_SET_BANDWIDTH_REQUEST 4008
#define OPUS_GET_BANDWIDTH_REQUEST 4009
#define OPUS_SET_COMPLEXITY_REQUEST 4010
#define OPUS_GET_COMPLEXITY_REQUEST 4011
#define OPUS_SET_INBAND_FEC_REQUEST 4012
#define OPUS_GET_INBAND_FEC_REQUEST 4013
#define OPUS_SET_PACKET_LOSS_PERC_REQUEST 4014
#define OPUS_GET_PACKET_LOSS_PERC_REQUEST 4015
#define OPUS_SET_DTX_REQUEST 4016
#define OPUS_GET_DTX_REQUEST 4017
#define OPUS_GET_IN_DTX_REQUEST 4049
#define OPUS_SET_VBR_CONSTRAINT_REQUEST 4020
#define OPUS_GET_VBR_CONSTRAINT_REQUEST 4021
#define OPUS_SET_FORCE_CHANNELS_REQUEST 4022
#define OPUS_GET_FORCE_CHANNELS_REQUEST 4023
#define OPUS_SET_SIGNAL_REQUEST 4024
#define OPUS_GET_SIGNAL_REQUEST 4025
#define OPUS_RESET_STATE 4028
#define OPUS_GET_SAMPLE_RATE_REQUEST 4029
#define OPUS_GET_FINAL_RANGE_REQUEST 4031
#define OPUS_SET_LSB_DEPTH_REQUEST 4036
#define OPUS_GET_LSB_DEPTH_REQUEST 4037
#define OPUS_SET_FORCE_MODE_REQUEST 11002              /* src/opus_private.h:173 (what opus_demo and the tests use to pin SILK-only / hybrid) */
#define OPUS_SIGNAL_VOICE 3001
#define OPUS_SIGNAL_MUSIC 3002
#define OPUS_MODE_SILK_ONLY 1000
#define OPUS_MODE_HYBRID 1001
#define OPUS_MODE_CELT_ONLY 1002
#define OPUS_SET_PHASE_INVERSION_DISABLED_REQUEST 4046
#define OPUS_GET_PHASE_INVERSION_DISABLED_REQUEST 4047

typedef int16_t opus_int16;
typedef int8_t opus_int8;
typedef int32_t opus_int32;
typedef uint32_t opus_uint32;
typedef struct OpusEncoder OpusEncoder;

/* ---- classic API (replaces reference/src/opus_encoder.c:194 get_size, :204 init, :622 create, :2671 opus_encode, :2772 ctl, :3362 destroy;
 * celt/celt.c:342 opus_strerror, :360 opus_get_version_string) ---- */
OPUS_AMD_EXPORT int opus_encoder_get_size(int channels);
OPUS_AMD_EXPORT OpusEncoder *opus_encoder_create(opus_int32 Fs, int channels, int application, int *error);
OPUS_AMD_EXPORT int opus_encoder_init(OpusEncoder *st, opus_int32 Fs, int channels, int application);
OPUS_AMD_EXPORT opus_int32 opus_encode(OpusEncoder *st, const opus_int16 *pcm, int frame_size, unsigned char *data, opus_int32 max_data_bytes);
OPUS_AMD_EXPORT int opus_encoder_ctl(OpusEncoder *st, int request, ...);
OPUS_AMD_EXPORT void opus_encoder_destroy(OpusEncoder *st);
OPUS_AMD_EXPORT const char *opus_strerror(int error);
OPUS_AMD_EXPORT const char *opus_get_version_string(void);

/* ---- batch API ---- */
typedef struct OpusGpuEncBatch OpusGpuEncBatch;

OPUS_AMD_EXPORT int opusgpu_device_count(void);
/* "OA_SRC_HASH=<16 hex digits>": the hash of the source files this library was compiled from (opus_amd.build() / opus_amd.source_hash()), so that a run can state
 * which sources its numbers belong to; "OA_SRC_HASH=unknown" for a build outside opus_amd.build() */
OPUS_AMD_EXPORT const char *opusgpu_build_info(void);
/* S streams with a common initial configuration on HIP device `device`. */
OPUS_AMD_EXPORT OpusGpuEncBatch *opusgpu_enc_batch_create(opus_int32 nstreams, opus_int32 Fs, int channels, int application, int device, int *error);
OPUS_AMD_EXPORT void opusgpu_enc_batch_destroy(OpusGpuEncBatch *b);
OPUS_AMD_EXPORT opus_int32 opusgpu_enc_batch_streams(const OpusGpuEncBatch *b);
/* set-type CTL on one stream (stream >= 0) or on all (stream == -1); same request numbers / validation as opus_encoder_ctl */
OPUS_AMD_EXPORT int opusgpu_enc_batch_ctl(OpusGpuEncBatch *b, opus_int32 stream, int request, opus_int32 value);
OPUS_AMD_EXPORT int opusgpu_enc_batch_get(OpusGpuEncBatch *b, opus_int32 stream, int request, opus_int32 *value);
/* One frame-step for every stream.  pcm: [S][frame_size*channels] int16 interleaved; out: [S][out_stride] bytes;
 * lens[s] = packet length or negative error; final_range[s] = OPUS_GET_FINAL_RANGE.  Host pointers.
 * out_stride must hold the largest packet the call can return, with m = min(max_data_bytes, 1276 * 6) (the reference clamps every call's budget to that, src/opus_encoder.c:1221):
 *   frames up to 20 ms, every stream VBR:        min(m, 1276)      (TOC + at most 1275 bytes)
 *   frames up to 20 ms, some stream hard CBR:     m                 (a CBR packet is padded to its budget, :2646)
 *   calls above 20 ms (multi-frame packets):     max_data_bytes + 48 (hard CBR / OPUS_BITRATE_MAX pad such a packet to the caller's whole buffer, :1757; + staging head-room)
 * else OPUS_BUFFER_TOO_SMALL.  1280 serves every VBR batch of 2.5-20 ms frames whatever max_data_bytes is. */
OPUS_AMD_EXPORT int opusgpu_encode_batch(OpusGpuEncBatch *b, const opus_int16 *pcm, int frame_size, unsigned char *out,
      opus_int32 out_stride, opus_int32 max_data_bytes, opus_int32 *lens, opus_uint32 *final_range);
/* Same, with every buffer already resident in this device's HBM; asynchronous on `hip_stream` (NULL = the batch's stream). */
OPUS_AMD_EXPORT int opusgpu_encode_batch_dev(OpusGpuEncBatch *b, const opus_int16 *d_pcm, int frame_size, unsigned char *d_out,
      opus_int32 out_stride, opus_int32 max_data_bytes, opus_int32 *d_lens, opus_uint32 *d_final_range, void *hip_stream);
/* Both again with a second view of the same input for the encoder's tonality / music analysis (src/analysis.c): apcm [S][frame_size*channels] int32 in the encoder's
 * signal domain (2^12 per int16 LSB), what the reference's 24-bit and float entry points hand to run_analysis (downmix_int24 src/opus_encoder.c:804, downmix_float
 * :748; opus_encode24 :2697 and opus_encode_float :2735 of this library call these).  NULL = derive it from pcm (downmix_int :780), i.e. the plain entry points above. */
OPUS_AMD_EXPORT int opusgpu_encode_batch_sig(OpusGpuEncBatch *b, const opus_int16 *pcm, const opus_int32 *apcm, int frame_size, unsigned char *out,
      opus_int32 out_stride, opus_int32 max_data_bytes, opus_int32 *lens, opus_uint32 *final_range);
OPUS_AMD_EXPORT int opusgpu_encode_batch_dev_sig(OpusGpuEncBatch *b, const opus_int16 *d_pcm, const opus_int32 *d_apcm, int frame_size, unsigned char *d_out,
      opus_int32 out_stride, opus_int32 max_data_bytes, opus_int32 *d_lens, opus_uint32 *d_final_range, void *hip_stream);
/* Both once more with the reference's look-ahead (src/opus_encoder.c:1247, :2662-2690; run_analysis, src/analysis.c:954): every stream's row of pcm (and apcm) holds
 * analysis_frame_size >= frame_size samples per channel -- the buffer opus_encode() is handed when OPUS_SET_EXPERT_FRAME_DURATION selects a frame shorter than it -- of
 * which the first frame_size are coded; the tonality / music analysis sees the whole row (the classic entry points of this library call these).  analysis_frame_size ==
 * frame_size is exactly opusgpu_encode_batch_sig / _dev_sig. */
OPUS_AMD_EXPORT int opusgpu_encode_batch_lookahead(OpusGpuEncBatch *b, const opus_int16 *pcm, const opus_int32 *apcm, int frame_size, int analysis_frame_size, unsigned char *out,
      opus_int32 out_stride, opus_int32 max_data_bytes, opus_int32 *lens, opus_uint32 *final_range);
OPUS_AMD_EXPORT int opusgpu_encode_batch_lookahead_dev(OpusGpuEncBatch *b, const opus_int16 *d_pcm, const opus_int32 *d_apcm, int frame_size, int analysis_frame_size,
      unsigned char *d_out, opus_int32 out_stride, opus_int32 max_data_bytes, opus_int32 *d_lens, opus_uint32 *d_final_range, void *hip_stream);
/* Private set / get requests of this library (opus_encoder_ctl, opusgpu_enc_batch_ctl, opus_multistream_encoder_ctl): OPUS_AMD_SET_FLOAT_ANALYSIS(1) (the default) = run the
 * tonality / music analysis at complexity 10 like a FIXED_POINT libopus with its float API, the default build (src/opus_encoder.c:1249); (0) = like one built with
 * DISABLE_FLOAT_API.  The process-wide default for new encoders can be set with the environment variable OPUS_AMD_FLOAT_ANALYSIS=0. */
#define OPUS_AMD_SET_FLOAT_ANALYSIS_REQUEST 11900
#define OPUS_AMD_GET_FLOAT_ANALYSIS_REQUEST 11901
/* OPUS_AMD_SET_KERNEL_PIPELINE(v): how the following 10 / 20 ms calls of a SILK-capable encoder are launched -- a property of the launch, never of the packets (every value
 * gives the same bytes).  -1 (the default) = the library chooses: the front / pred / quantiser / back kernel pipeline (opus_sh_split.h; value 4) when the launch carries >= 64 streams, one
 * kernel below that; 0 = always one kernel; 1 = always the front / quantiser / back pipeline; 2 = the same with its one-wave-per-stream reference quantiser; 3 = front / pred / quantiser / back (the
 * prediction stage -- LPC, NLSF, residual energies, gains -- as a kernel of its own at twice the front kernel's occupancy); 4 = the same with the prediction stage cut into
 * lane kernels for its serial parts (Burg's recursion, A2NLSF, NLSF2A, the NLSF trellises: one lane per coded channel, silk_enc_predl.h) and wave kernels for its passes over
 * the signal.  On a batch (opusgpu_enc_batch_ctl, any
 * `stream`) it applies to the whole batch; on a classic encoder it applies to that encoder's calls (calls with different values are not combined into one launch); a
 * multistream encoder passes it to its elementary encoders.  The process-wide default behind -1 can be set with the environment variable OPUS_AMD_SH_SPLIT=0..4. */
/* OPUS_AMD_SET_KERNEL_TIMING(1) on a batch: HIP events on the launch stream around every kernel of the following calls; opusgpu_enc_batch_kernel_times returns the last call's
 * kernel names (comma separated, in launch order) and durations in ms (bench.py's per-kernel figures; no effect on the packets). */
#define OPUS_AMD_SET_KERNEL_TIMING_REQUEST 11904
/* Two more launch properties (never the packets'), same scope rules as OPUS_AMD_SET_KERNEL_PIPELINE; -1 = the library chooses (the default), 0 = off, 1 = on:
 *   OPUS_AMD_SET_TRANSIENT_PREPASS(v)  the serial recursions of CELT's transient analysis as a lane pre-pass ahead of the kernel that applies them (celt_enc_front.h:
 *                                       ct_transient_lane); -1: on for 48 kHz launches of >= 64 streams (CELT-only applications) / of the kernel pipeline (the others)
 *   OPUS_AMD_SET_PVQ_STAGE(v)          the PVQ of 10 / 20 ms CELT frames as a kernel of its own with four streams per wave (celt_enc_pvq4.h: oa_celt_pvq_kernel); -1: on
 *                                       wherever the call runs as a kernel pipeline.  For a CELT-only application the pipeline IS this stage: 0 keeps the call in one kernel
 * The process-wide defaults behind -1 can be set with OPUS_AMD_TR_PRE=0|1|2 and OPUS_AMD_CELT_PIPE=0|1 / OPUS_AMD_SH_PVQ4=0|1 (experiments). */
#define OPUS_AMD_SET_TRANSIENT_PREPASS_REQUEST 11906
#define OPUS_AMD_GET_TRANSIENT_PREPASS_REQUEST 11907
#define OPUS_AMD_SET_PVQ_STAGE_REQUEST 11908
#define OPUS_AMD_GET_PVQ_STAGE_REQUEST 11909
#define OPUS_AMD_SET_KERNEL_PIPELINE_REQUEST 11902
#define OPUS_AMD_GET_KERNEL_PIPELINE_REQUEST 11903
/* n stream records -- configuration and state as they stand on the device -- from batch `src` (from stream src_first on) into batch `dst` (from dst_first on), device to
 * device; both batches of one shape (application kind, rate, channels) on one device, the ranges inside them and, within one batch, disjoint.  The calls on both batches
 * issued so far are waited for. */
OPUS_AMD_EXPORT int opusgpu_enc_batch_copy_states(OpusGpuEncBatch *dst, opus_int32 dst_first, OpusGpuEncBatch *src, opus_int32 src_first, opus_int32 n);
OPUS_AMD_EXPORT int opusgpu_enc_batch_sync(OpusGpuEncBatch *b);
/* `steps` back-to-back frame-steps on device buffers ([steps][S][frame*channels] PCM), timed with HIP events on the
 * launch stream; returns elapsed milliseconds in *ms (kernel time only, inputs resident). */
OPUS_AMD_EXPORT int opusgpu_time_encode_dev(OpusGpuEncBatch *b, const opus_int16 *d_pcm, int frame_size, unsigned char *d_out,
      opus_int32 out_stride, opus_int32 max_data_bytes, opus_int32 *d_lens, opus_uint32 *d_final_range, int steps, float *ms);
/* T consecutive frame-steps of every stream in ONE launch (CELT-only applications): d_pcm [T][S][frame_size*channels], d_out [T][S][out_stride], d_lens / d_final_range [T][S] */
OPUS_AMD_EXPORT int opusgpu_encode_batch_dev_frames(OpusGpuEncBatch *b, const opus_int16 *d_pcm, int frame_size, int T, unsigned char *d_out, opus_int32 out_stride,
      opus_int32 max_data_bytes, opus_int32 *d_lens, opus_uint32 *d_final_range, void *hip_stream);
/* final-gather compaction (opus_amd/shard.py): packet s = d_lens[s] bytes of its slot -> d_packed[d_offsets[s] ...]; d_offsets = exclusive prefix sum of the lengths (int64) */
OPUS_AMD_EXPORT int opusgpu_pack_packets_dev(const unsigned char *d_out, opus_int32 stride, const opus_int32 *d_lens, const long long *d_offsets, unsigned char *d_packed,
      opus_int32 n, void *hip_stream);
/* the same into a record of `capacity` bytes: bytes beyond it are dropped (the lengths travel with the record, so the receiver can tell) */
OPUS_AMD_EXPORT int opusgpu_pack_packets_cap_dev(const unsigned char *d_out, opus_int32 stride, const opus_int32 *d_lens, const long long *d_offsets, unsigned char *d_packed,
      opus_int32 n, long long capacity, void *hip_stream);
/* state bytes one frame-step reads plus writes (roofline accounting): application, channels, hybrid frame? */
OPUS_AMD_EXPORT int opusgpu_enc_moved_state_bytes(int application, int channels, int hybrid);
/* ... and what the tonality analysis adds to that when it runs (complexity 10, API rate >= 16 kHz, float analysis on) */
OPUS_AMD_EXPORT int opusgpu_enc_analysis_moved_bytes(void);
/* surround_analysis (src/opus_multistream_encoder.c:230) as an operator: per-channel signal-to-mask ratios (Q24, [channels][21]) of `len` interleaved int16 samples of a
 * 3..8-channel vorbis-order layout; mem[channels][120], preemph_mem[channels]: the analysis memory, updated */
OPUS_AMD_EXPORT int opusgpu_surround_analysis(const opus_int16 *pcm, int len, int channels, opus_int32 Fs, opus_int32 *mem, opus_int32 *preemph_mem, opus_int32 *bandSMR);
/* the classic entry points under concurrent callers (include/opus.h:425-429 allows any number of threads on different states): calls that arrive while a launch is in
 * flight share the next launch, one wave per call (opus_amd/csrc/opus_call_combiner.h; at most OPUS_AMD_CLASSIC_BATCH states per launch, default 256).
 * out = {opus_encode* calls, launches that served them, opus_decode* calls, launches that served them} since the library was loaded */
OPUS_AMD_EXPORT void opusgpu_classic_call_stats(long long out[4]);
/* memcpy contract: a stream's complete state as a flat blob (same layout as the classic OpusEncoder payload) */
OPUS_AMD_EXPORT int opusgpu_enc_state_size(void);
/* record size / LDS per wave of the SILK-capable encoder (batches created with OPUS_APPLICATION_VOIP / _AUDIO / _RESTRICTED_SILK; src/opus_encoder.c:76-146 + silk/fixed/structs_FIX.h:108) */
OPUS_AMD_EXPORT int opusgpu_enc_sh_state_size(void);
OPUS_AMD_EXPORT int opusgpu_sh_kernel_lds_bytes(void);
OPUS_AMD_EXPORT int opusgpu_enc_batch_export_state(OpusGpuEncBatch *b, opus_int32 stream, void *blob);
OPUS_AMD_EXPORT int opusgpu_enc_batch_import_state(OpusGpuEncBatch *b, opus_int32 stream, const void *blob);
OPUS_AMD_EXPORT int opusgpu_enc_batch_reset(OpusGpuEncBatch *b);
/* Diagnostics of a SILK-capable batch (applications VOIP / AUDIO / RESTRICTED_SILK): calls that went through the three-kernel split path of the encoder (front / 16-streams-
 * per-wave quantiser / back, opus_amd/csrc/opus_sh_split.h) and calls it handed to the one-kernel path since the batch was created.  Results never depend on the path. */
OPUS_AMD_EXPORT int opusgpu_enc_batch_split_stats(OpusGpuEncBatch *b, opus_uint32 *kept, opus_uint32 *declined);
/* *streams = how many streams of the batch's LAST call had the PVQ of their CELT frame coded by the four-streams-per-wave stage (oa_celt_pvq_kernel: celt_enc_pvq4.h) --
 * 10 / 20 ms calls of a launch that runs as a kernel pipeline (OPUS_AMD_SET_KERNEL_PIPELINE); 0 when the call ran as one kernel.  Waits for the batch's stream.  Test / bench aid. */
OPUS_AMD_EXPORT int opusgpu_enc_batch_pvq_stage_stats(OpusGpuEncBatch *b, opus_uint32 *streams);
/* returns the number of kernels of the batch's last call (<= max_kernels), 0 without OPUS_AMD_SET_KERNEL_TIMING(1); waits for that call */
OPUS_AMD_EXPORT int opusgpu_enc_batch_kernel_times(OpusGpuEncBatch *b, char *names, int names_cap, float *ms, int max_kernels);
/* introspection for the roofline report */
OPUS_AMD_EXPORT int opusgpu_kernel_lds_bytes(void);


/* ================= decoder: every Opus packet mode (CELT-only, SILK-only, hybrid), 48 kHz output =================
 * Classic API — same names, arguments and error codes as reference/include/opus.h: opus_decoder_get_size :460,
 * opus_decoder_create :477, opus_decoder_init :494, opus_decode :516, opus_decoder_ctl :586, opus_decoder_destroy :591
 * (definitions replaced: reference/src/opus_decoder.c:121, :186, :135, :890, :1033, :1246).  The OpusDecoder blob is flat
 * host memory with the complete state (memcpy-able).  Scope: CELT-only, SILK-only (NB/MB/WB, 10-60 ms) and hybrid packets, any frame count /
 * size the TOC allows, mono/stereo streams into mono/stereo output, mode transitions in every direction incl. the 5 ms CELT redundancy frames,
 * packet-loss concealment in every mode (data == NULL or len == 0: CELT pitch/noise PLC, SILK PLC + comfort noise, hybrid = both) and DTX frames.
 * decode_fec = 1 decodes the in-band FEC (LBRR) copy (src/opus_decoder.c:786-824), concealing where there is none.  Every API rate (8-48 kHz: CELT keeps every n-th de-emphasised
 * sample, celt_decoder.c:361-404; SILK resamples to the API rate); OPUS_SET_GAIN; opus_decode24 / opus_decode_float as the fixed-point build converts them. */
typedef struct OpusDecoder OpusDecoder;
OPUS_AMD_EXPORT int opus_decoder_get_size(int channels);
OPUS_AMD_EXPORT OpusDecoder *opus_decoder_create(opus_int32 Fs, int channels, int *error);
OPUS_AMD_EXPORT int opus_decoder_init(OpusDecoder *st, opus_int32 Fs, int channels);
OPUS_AMD_EXPORT int opus_decode(OpusDecoder *st, const unsigned char *data, opus_int32 len, opus_int16 *pcm, int frame_size, int decode_fec);
/* reference include/opus.h:541 (the API opus_demo decodes with, src/opus_demo.c:1145): 24-bit samples in int32, here int16 resolution << 8 like the reference's int16-resolution build */
OPUS_AMD_EXPORT int opus_decode24(OpusDecoder *st, const unsigned char *data, opus_int32 len, opus_int32 *pcm, int frame_size, int decode_fec);
OPUS_AMD_EXPORT int opus_decoder_ctl(OpusDecoder *st, int request, ...);
OPUS_AMD_EXPORT void opus_decoder_destroy(OpusDecoder *st);

/* Batch decoder: S independent streams, one wavefront per stream per call; state (incl. the 2x2048-sample synthesis history)
 * stays in HBM.  packets: [S][packet_stride] bytes (one Opus packet per stream, lens[s] bytes used); pcm: [S][frame_size*channels]
 * int16 interleaved (frame_size = capacity per channel, <= 5760); lens[s] == 0 marks a lost packet: that stream conceals frame_size
 * samples; nsamples[s] = samples per channel decoded or a negative
 * OPUS_* code for that stream; final_range[s] = OPUS_GET_FINAL_RANGE. */
typedef struct OpusGpuDecBatch OpusGpuDecBatch;
OPUS_AMD_EXPORT OpusGpuDecBatch *opusgpu_dec_batch_create(opus_int32 nstreams, opus_int32 Fs, int channels, int device, int *error);
OPUS_AMD_EXPORT void opusgpu_dec_batch_destroy(OpusGpuDecBatch *b);
OPUS_AMD_EXPORT opus_int32 opusgpu_dec_batch_streams(const OpusGpuDecBatch *b);
OPUS_AMD_EXPORT int opusgpu_decode_batch(OpusGpuDecBatch *b, const unsigned char *packets, opus_int32 packet_stride, const opus_int32 *lens,
      opus_int16 *pcm, int frame_size, opus_int32 *nsamples, opus_uint32 *final_range);
OPUS_AMD_EXPORT int opusgpu_decode_batch_dev(OpusGpuDecBatch *b, const unsigned char *d_packets, opus_int32 packet_stride, const opus_int32 *d_lens,
      opus_int16 *d_pcm, int frame_size, opus_int32 *d_nsamples, opus_uint32 *d_final_range, void *hip_stream);
OPUS_AMD_EXPORT int opusgpu_time_decode_dev(OpusGpuDecBatch *b, const unsigned char *d_packets, opus_int32 packet_stride, const opus_int32 *d_lens,
      opus_int16 *d_pcm, int frame_size, opus_int32 *d_nsamples, opus_uint32 *d_final_range, int steps, float *ms);
OPUS_AMD_EXPORT int opusgpu_dec_batch_sync(OpusGpuDecBatch *b);
OPUS_AMD_EXPORT int opusgpu_dec_batch_set_fec(OpusGpuDecBatch *b, int decode_fec);   /* decode_fec of the following decode calls (include/opus.h:516) */
OPUS_AMD_EXPORT int opusgpu_dec_batch_set_fast_kernel(OpusGpuDecBatch *b, int enable);   /* 0: skip the CELT-only fast kernel (a batch without CELT-only packets saves its look at every stream); default 1; same output either way */
OPUS_AMD_EXPORT int opusgpu_dec_batch_set_lane_kernel(OpusGpuDecBatch *b, int enable);   /* 0: skip oa_sdec_lane_kernel (the SILK steady state, one lane per stream: silk/dec_API.c:142 silk_Decode for 64 streams per wave); default 1; same output either way */
OPUS_AMD_EXPORT int opusgpu_dec_batch_lane_stats(OpusGpuDecBatch *b, opus_uint32 *taken, opus_uint32 *handed_on);   /* the last call: packets oa_sdec_lane_kernel was given / of those, handed on to the general kernel (a redundant CELT frame behind the SILK data, src/opus_decoder.c:499-526) */
OPUS_AMD_EXPORT int opusgpu_dec_batch_set_pvq_stage(OpusGpuDecBatch *b, int mode);   /* the band decoding (celt/bands.c:1589 quant_all_bands, encode = 0) of steady-state CELT-only / hybrid packets of one 10 / 20 ms frame as oa_celt_dpvq_kernel, four streams per wave: -1 (default) wide calls, 0 never, 1 always; same output either way */
OPUS_AMD_EXPORT int opusgpu_dec_batch_pvq_stats(OpusGpuDecBatch *b, opus_uint32 *frames);   /* the last call: frames whose bands oa_celt_dpvq_kernel decoded */
OPUS_AMD_EXPORT int opusgpu_dec_batch_reset(OpusGpuDecBatch *b);
OPUS_AMD_EXPORT int opusgpu_dec_state_size(void);
OPUS_AMD_EXPORT int opusgpu_dec_batch_export_state(OpusGpuDecBatch *b, opus_int32 stream, void *blob);
OPUS_AMD_EXPORT int opusgpu_dec_batch_import_state(OpusGpuDecBatch *b, opus_int32 stream, const void *blob);
OPUS_AMD_EXPORT int opusgpu_dec_kernel_lds_bytes(void);
OPUS_AMD_EXPORT int opusgpu_dec_fast_kernel_lds_bytes(void);     /* LDS of one wave of the CELT-only fast kernel (16 waves per CU at <= 10,240 B) */

/* ================= packet toolkit (host-side; reference/include/opus.h:713-788 and :953-1167) =================
 * Same names, arguments and results as the reference (src/opus.c:203-399, src/opus_decoder.c:1252-1340, src/repacketizer.c).
 * Extension payloads carried in code-3 padding are not interpreted (dropped on re-assembly). */
typedef struct OpusRepacketizer OpusRepacketizer;
OPUS_AMD_EXPORT int opus_packet_parse(const unsigned char *data, opus_int32 len, unsigned char *out_toc, const unsigned char *frames[48], opus_int16 size[48], int *payload_offset);
OPUS_AMD_EXPORT int opus_packet_get_bandwidth(const unsigned char *data);
OPUS_AMD_EXPORT int opus_packet_get_samples_per_frame(const unsigned char *data, opus_int32 Fs);
OPUS_AMD_EXPORT int opus_packet_get_nb_channels(const unsigned char *data);
OPUS_AMD_EXPORT int opus_packet_get_nb_frames(const unsigned char packet[], opus_int32 len);
OPUS_AMD_EXPORT int opus_packet_get_nb_samples(const unsigned char packet[], opus_int32 len, opus_int32 Fs);
OPUS_AMD_EXPORT int opus_repacketizer_get_size(void);
OPUS_AMD_EXPORT OpusRepacketizer *opus_repacketizer_init(OpusRepacketizer *rp);
OPUS_AMD_EXPORT OpusRepacketizer *opus_repacketizer_create(void);
OPUS_AMD_EXPORT void opus_repacketizer_destroy(OpusRepacketizer *rp);
OPUS_AMD_EXPORT int opus_repacketizer_cat(OpusRepacketizer *rp, const unsigned char *data, opus_int32 len);
OPUS_AMD_EXPORT opus_int32 opus_repacketizer_out_range(OpusRepacketizer *rp, int begin, int end, unsigned char *data, opus_int32 maxlen);
OPUS_AMD_EXPORT int opus_repacketizer_get_nb_frames(OpusRepacketizer *rp);
OPUS_AMD_EXPORT opus_int32 opus_repacketizer_out(OpusRepacketizer *rp, unsigned char *data, opus_int32 maxlen);
OPUS_AMD_EXPORT int opus_packet_pad(unsigned char *data, opus_int32 len, opus_int32 new_len);
OPUS_AMD_EXPORT opus_int32 opus_packet_unpad(unsigned char *data, opus_int32 len);
OPUS_AMD_EXPORT int opus_multistream_packet_pad(unsigned char *data, opus_int32 len, opus_int32 new_len, int nb_streams);
OPUS_AMD_EXPORT opus_int32 opus_multistream_packet_unpad(unsigned char *data, opus_int32 len, int nb_streams);

/* ================= multistream (reference/include/opus_multistream.h:203-726) =================
 * One multistream frame = its streams stepped together by the batch kernels (one launch per group: coupled, mono).  Same names,
 * arguments, layouts (mapping semantics :86-140) and error codes.  Scope: every application, frame sizes up to 120 ms, mapping families 0, 1 (surround: the
 * masking analysis of src/opus_multistream_encoder.c:230 runs on the device, one wave per channel; LFE stream), 2 / 3 (ambisonics and projection, opus_projection.h)
 * and 255; int16, 24-bit and float entry points (the latter two convert, as the reference's fixed-point build does). */
typedef struct OpusMSEncoder OpusMSEncoder;
typedef struct OpusMSDecoder OpusMSDecoder;
OPUS_AMD_EXPORT opus_int32 opus_multistream_encoder_get_size(int streams, int coupled_streams);
OPUS_AMD_EXPORT opus_int32 opus_multistream_surround_encoder_get_size(int channels, int mapping_family);
OPUS_AMD_EXPORT OpusMSEncoder *opus_multistream_encoder_create(opus_int32 Fs, int channels, int streams, int coupled_streams, const unsigned char *mapping, int application, int *error);
OPUS_AMD_EXPORT OpusMSEncoder *opus_multistream_surround_encoder_create(opus_int32 Fs, int channels, int mapping_family, int *streams, int *coupled_streams, unsigned char *mapping, int application, int *error);
OPUS_AMD_EXPORT int opus_multistream_encoder_init(OpusMSEncoder *st, opus_int32 Fs, int channels, int streams, int coupled_streams, const unsigned char *mapping, int application);
OPUS_AMD_EXPORT int opus_multistream_surround_encoder_init(OpusMSEncoder *st, opus_int32 Fs, int channels, int mapping_family, int *streams, int *coupled_streams, unsigned char *mapping, int application);
OPUS_AMD_EXPORT int opus_multistream_encode(OpusMSEncoder *st, const opus_int16 *pcm, int frame_size, unsigned char *data, opus_int32 max_data_bytes);
OPUS_AMD_EXPORT void opus_multistream_encoder_destroy(OpusMSEncoder *st);
OPUS_AMD_EXPORT int opus_multistream_encoder_ctl(OpusMSEncoder *st, int request, ...);
OPUS_AMD_EXPORT opus_int32 opus_multistream_decoder_get_size(int streams, int coupled_streams);
OPUS_AMD_EXPORT OpusMSDecoder *opus_multistream_decoder_create(opus_int32 Fs, int channels, int streams, int coupled_streams, const unsigned char *mapping, int *error);
OPUS_AMD_EXPORT int opus_multistream_decoder_init(OpusMSDecoder *st, opus_int32 Fs, int channels, int streams, int coupled_streams, const unsigned char *mapping);
OPUS_AMD_EXPORT int opus_multistream_decode(OpusMSDecoder *st, const unsigned char *data, opus_int32 len, opus_int16 *pcm, int frame_size, int decode_fec);
OPUS_AMD_EXPORT int opus_multistream_decoder_ctl(OpusMSDecoder *st, int request, ...);
OPUS_AMD_EXPORT void opus_multistream_decoder_destroy(OpusMSDecoder *st);

/* ================= the rest of the reference's exported surface (include/opus.h, opus_multistream.h, opus_projection.h) =================
 * Same names, arguments and error codes.  The arithmetic of this library is the reference's int16-resolution fixed-point build, so the 24-bit and float
 * entry points convert at the boundary exactly as that build does (celt/arch.h:167-173: INT24TORES = SAT16(PSHR32(x, 8)), FLOAT2RES = FLOAT2INT16,
 * RES2INT24 = x << 8, RES2FLOAT = x / 32768).  DRED is not built: its entry points answer like a reference compiled without ENABLE_DRED. */
OPUS_AMD_EXPORT opus_int32 opus_encode24(OpusEncoder *st, const opus_int32 *pcm, int frame_size, unsigned char *data, opus_int32 max_data_bytes);      /* opus.h:302 */
OPUS_AMD_EXPORT opus_int32 opus_encode_float(OpusEncoder *st, const float *pcm, int frame_size, unsigned char *data, opus_int32 max_data_bytes);      /* opus.h:343 */
OPUS_AMD_EXPORT int opus_decode_float(OpusDecoder *st, const unsigned char *data, opus_int32 len, float *pcm, int frame_size, int decode_fec);        /* opus.h:566 */
OPUS_AMD_EXPORT int opus_decoder_get_nb_samples(const OpusDecoder *dec, const unsigned char packet[], opus_int32 len);                                /* opus.h:788 */
OPUS_AMD_EXPORT int opus_packet_has_lbrr(const unsigned char packet[], opus_int32 len);                                                               /* opus.h:778 */
OPUS_AMD_EXPORT void opus_pcm_soft_clip(float *pcm, int frame_size, int channels, float *softclip_mem);                                               /* opus.h:800 */
OPUS_AMD_EXPORT int opus_multistream_encode24(OpusMSEncoder *st, const opus_int32 *pcm, int frame_size, unsigned char *data, opus_int32 max_data_bytes);
OPUS_AMD_EXPORT int opus_multistream_encode_float(OpusMSEncoder *st, const float *pcm, int frame_size, unsigned char *data, opus_int32 max_data_bytes);
OPUS_AMD_EXPORT int opus_multistream_decode24(OpusMSDecoder *st, const unsigned char *data, opus_int32 len, opus_int32 *pcm, int frame_size, int decode_fec);
OPUS_AMD_EXPORT int opus_multistream_decode_float(OpusMSDecoder *st, const unsigned char *data, opus_int32 len, float *pcm, int frame_size, int decode_fec);
typedef struct OpusDREDDecoder OpusDREDDecoder;
typedef struct OpusDRED OpusDRED;
OPUS_AMD_EXPORT int opus_dred_decoder_get_size(void);
OPUS_AMD_EXPORT OpusDREDDecoder *opus_dred_decoder_create(int *error);
OPUS_AMD_EXPORT int opus_dred_decoder_init(OpusDREDDecoder *dec);
OPUS_AMD_EXPORT void opus_dred_decoder_destroy(OpusDREDDecoder *dec);
OPUS_AMD_EXPORT int opus_dred_decoder_ctl(OpusDREDDecoder *dred_dec, int request, ...);
OPUS_AMD_EXPORT int opus_dred_get_size(void);
OPUS_AMD_EXPORT OpusDRED *opus_dred_alloc(int *error);
OPUS_AMD_EXPORT void opus_dred_free(OpusDRED *dec);
OPUS_AMD_EXPORT int opus_dred_parse(OpusDREDDecoder *dred_dec, OpusDRED *dred, const unsigned char *data, opus_int32 len, opus_int32 max_dred_samples, opus_int32 sampling_rate, int *dred_end, int defer_processing);
OPUS_AMD_EXPORT int opus_dred_process(OpusDREDDecoder *dred_dec, const OpusDRED *src, OpusDRED *dst);
OPUS_AMD_EXPORT int opus_decoder_dred_decode(OpusDecoder *st, const OpusDRED *dred, opus_int32 dred_offset, opus_int16 *pcm, opus_int32 frame_size);
OPUS_AMD_EXPORT int opus_decoder_dred_decode24(OpusDecoder *st, const OpusDRED *dred, opus_int32 dred_offset, opus_int32 *pcm, opus_int32 frame_size);
OPUS_AMD_EXPORT int opus_decoder_dred_decode_float(OpusDecoder *st, const OpusDRED *dred, opus_int32 dred_offset, float *pcm, opus_int32 frame_size);

/* ================= projection (ambisonics, reference/include/opus_projection.h:123-632) =================
 * A mixing matrix in front of a multistream encoder / a demixing matrix behind a multistream decoder (src/opus_projection_encoder.c,
 * src/opus_projection_decoder.c, src/mapping_matrix.c), mapping family 3, orders 1-5 (+ optional non-diegetic stereo pair). */
typedef struct OpusProjectionEncoder OpusProjectionEncoder;
typedef struct OpusProjectionDecoder OpusProjectionDecoder;
OPUS_AMD_EXPORT opus_int32 opus_projection_ambisonics_encoder_get_size(int channels, int mapping_family);
OPUS_AMD_EXPORT OpusProjectionEncoder *opus_projection_ambisonics_encoder_create(opus_int32 Fs, int channels, int mapping_family, int *streams, int *coupled_streams, int application, int *error);
OPUS_AMD_EXPORT int opus_projection_ambisonics_encoder_init(OpusProjectionEncoder *st, opus_int32 Fs, int channels, int mapping_family, int *streams, int *coupled_streams, int application);
OPUS_AMD_EXPORT int opus_projection_encode(OpusProjectionEncoder *st, const opus_int16 *pcm, int frame_size, unsigned char *data, opus_int32 max_data_bytes);
OPUS_AMD_EXPORT int opus_projection_encode24(OpusProjectionEncoder *st, const opus_int32 *pcm, int frame_size, unsigned char *data, opus_int32 max_data_bytes);
OPUS_AMD_EXPORT int opus_projection_encode_float(OpusProjectionEncoder *st, const float *pcm, int frame_size, unsigned char *data, opus_int32 max_data_bytes);
OPUS_AMD_EXPORT void opus_projection_encoder_destroy(OpusProjectionEncoder *st);
OPUS_AMD_EXPORT int opus_projection_encoder_ctl(OpusProjectionEncoder *st, int request, ...);
OPUS_AMD_EXPORT opus_int32 opus_projection_decoder_get_size(int channels, int streams, int coupled_streams);
OPUS_AMD_EXPORT OpusProjectionDecoder *opus_projection_decoder_create(opus_int32 Fs, int channels, int streams, int coupled_streams, unsigned char *demixing_matrix, opus_int32 demixing_matrix_size, int *error);
OPUS_AMD_EXPORT int opus_projection_decoder_init(OpusProjectionDecoder *st, opus_int32 Fs, int channels, int streams, int coupled_streams, unsigned char *demixing_matrix, opus_int32 demixing_matrix_size);
OPUS_AMD_EXPORT int opus_projection_decode(OpusProjectionDecoder *st, const unsigned char *data, opus_int32 len, opus_int16 *pcm, int frame_size, int decode_fec);
OPUS_AMD_EXPORT int opus_projection_decode24(OpusProjectionDecoder *st, const unsigned char *data, opus_int32 len, opus_int32 *pcm, int frame_size, int decode_fec);
OPUS_AMD_EXPORT int opus_projection_decode_float(OpusProjectionDecoder *st, const unsigned char *data, opus_int32 len, float *pcm, int frame_size, int decode_fec);
OPUS_AMD_EXPORT int opus_projection_decoder_ctl(OpusProjectionDecoder *st, int request, ...);
OPUS_AMD_EXPORT void opus_projection_decoder_destroy(OpusProjectionDecoder *st);

/* ================= SILK building blocks (reference/silk/NSQ.c, NSQ_del_dec.c, LPC_analysis_filter.c) =================
 * The noise-shaping quantiser for N independent SILK channels, one frame per call.  This is the reference's own RTCD cut
 * (SILK_NSQ_IMPL / SILK_NSQ_DEL_DEC_IMPL, silk/x86/x86_silk_map.c:47-179, prototypes silk/main.h:236-272) lifted to a batch:
 *   OpusGpuNsqConfig = the six silk_encoder_state fields the quantisers read (silk/structs.h:167-207); frame_length =
 *                      nb_subfr * 5 * fs_kHz, ltp_mem_length = 20 * fs_kHz as set by silk_setup_fs (silk/control_codec.c:225-246)
 *   OpusGpuNsqFrame  = the argument list of silk_NSQ_c / silk_NSQ_del_dec_c (silk/NSQ.c:76-93, silk/NSQ_del_dec.c:114-131) incl. the
 *                      four SideInfoIndices fields they read (silk/structs.h:129-141)
 *   state blob       = silk_nsq_state, byte for byte (silk/structs.h:56-69, 4,352 B); import/export only — between calls the
 *                      states live in HBM in a tile-transposed layout
 * Dispatch as in the reference: delayed decision iff nStatesDelayedDecision > 1 || warping_Q16 > 0 (silk/float/wrappers_FLP.c:163).
 * Output: pulses[n][frame_length] exactly as the reference writes them; seed_out[i] = psIndices->Seed after the call. */
typedef struct OpusGpuNsqBatch OpusGpuNsqBatch;
typedef struct { opus_int32 fs_kHz, nb_subfr, predictLPCOrder, shapingLPCOrder, nStatesDelayedDecision, warping_Q16; } OpusGpuNsqConfig;
typedef struct {
   signed char signalType, quantOffsetType, NLSFInterpCoef_Q2, Seed;
   opus_int16 PredCoef_Q12[2 * 16];
   opus_int16 LTPCoef_Q14[5 * 4];
   opus_int16 AR_Q13[4 * 24];
   opus_int32 HarmShapeGain_Q14[4], Tilt_Q14[4], LF_shp_Q14[4], Gains_Q16[4], pitchL[4];
   opus_int32 Lambda_Q10, LTP_scale_Q14;
} OpusGpuNsqFrame;
OPUS_AMD_EXPORT OpusGpuNsqBatch *opusgpu_nsq_batch_create(opus_int32 nstreams, const OpusGpuNsqConfig *config, int device, int *error);
OPUS_AMD_EXPORT void opusgpu_nsq_batch_destroy(OpusGpuNsqBatch *b);
OPUS_AMD_EXPORT opus_int32 opusgpu_nsq_batch_streams(const OpusGpuNsqBatch *b);
OPUS_AMD_EXPORT int opusgpu_nsq_batch_frame_length(const OpusGpuNsqBatch *b);
OPUS_AMD_EXPORT int opusgpu_nsq_batch_reset(OpusGpuNsqBatch *b);                       /* silk/control_codec.c:247-258 */
OPUS_AMD_EXPORT int opusgpu_nsq_state_size(void);
OPUS_AMD_EXPORT int opusgpu_nsq_batch_import_state(OpusGpuNsqBatch *b, opus_int32 stream, const void *silk_nsq_state);
OPUS_AMD_EXPORT int opusgpu_nsq_batch_export_state(OpusGpuNsqBatch *b, opus_int32 stream, void *silk_nsq_state);
OPUS_AMD_EXPORT int opusgpu_nsq_batch_run(OpusGpuNsqBatch *b, const OpusGpuNsqFrame *frames, const opus_int16 *x16, opus_int8 *pulses, opus_int8 *seed_out);
OPUS_AMD_EXPORT int opusgpu_nsq_batch_run_dev(OpusGpuNsqBatch *b, const OpusGpuNsqFrame *d_frames, const opus_int16 *d_x16, opus_int8 *d_pulses,
      opus_int8 *d_seed_out, void *hip_stream);
OPUS_AMD_EXPORT int opusgpu_nsq_batch_sync(OpusGpuNsqBatch *b);
OPUS_AMD_EXPORT int opusgpu_nsq_time_dev(OpusGpuNsqBatch *b, const OpusGpuNsqFrame *d_frames, const opus_int16 *d_x16, opus_int8 *d_pulses, int steps, float *ms);

/* silk_LPC_analysis_filter (silk/LPC_analysis_filter.c:49, prototype silk/SigProc_FIX.h:113-120) for n independent signals:
 * in[n][len], B[n][d] (Q12), out[n][len]; 6 <= d <= 16 even, d <= len <= 1024 (the reference's asserts :65-67).  One wave per signal,
 * signal and coefficients staged in LDS; the _dev form takes device pointers and a hipStream_t (NULL = default stream). */
OPUS_AMD_EXPORT int opusgpu_silk_lpc_analysis_filter_batch(int device, opus_int32 n, opus_int16 *out, const opus_int16 *in, const opus_int16 *B, opus_int32 len, opus_int32 d);
OPUS_AMD_EXPORT int opusgpu_silk_lpc_analysis_filter_batch_dev(int device, opus_int32 n, opus_int16 *d_out, const opus_int16 *d_in, const opus_int16 *d_B, opus_int32 len,
      opus_int32 d, void *hip_stream);

/* silk_resampler (silk/resampler.c:79 silk_resampler_init, :183 silk_resampler; prototypes silk/SigProc_FIX.h:59-75) for n independent
 * channels of one rate pair.  create() validates the pair exactly like silk_resampler_init (forEnc=1: {8,12,16,24,48} kHz -> {8,12,16} kHz;
 * forEnc=0: {8,12,16} kHz -> {8,12,16,24,48} kHz; the ratios of resampler.c:135-166) and fails with OPUS_BAD_ARG otherwise.
 * run(): in[n][inLen] -> out[n][inLen * Fs_out / Fs_in], inLen a whole number of milliseconds >= 1 ms (the reference asserts >= 1 ms).
 * The state blob is silk_resampler_state_struct field for field (silk/resampler_structs.h:38-52) with the Coefs pointer replaced by a
 * table id; it can only be imported into a batch of the same rate pair. */
typedef struct OpusGpuResamplerBatch OpusGpuResamplerBatch;
OPUS_AMD_EXPORT OpusGpuResamplerBatch *opusgpu_resampler_batch_create(opus_int32 nchannels, opus_int32 Fs_Hz_in, opus_int32 Fs_Hz_out, int forEnc, int device, int *error);
OPUS_AMD_EXPORT void opusgpu_resampler_batch_destroy(OpusGpuResamplerBatch *b);
OPUS_AMD_EXPORT int opusgpu_resampler_batch_reset(OpusGpuResamplerBatch *b);
OPUS_AMD_EXPORT opus_int32 opusgpu_resampler_batch_out_len(const OpusGpuResamplerBatch *b, opus_int32 inLen);
OPUS_AMD_EXPORT int opusgpu_resampler_batch_run(OpusGpuResamplerBatch *b, opus_int16 *out, const opus_int16 *in, opus_int32 inLen);
OPUS_AMD_EXPORT int opusgpu_resampler_batch_run_dev(OpusGpuResamplerBatch *b, opus_int16 *d_out, const opus_int16 *d_in, opus_int32 inLen, void *hip_stream);
OPUS_AMD_EXPORT int opusgpu_resampler_batch_sync(OpusGpuResamplerBatch *b);
OPUS_AMD_EXPORT int opusgpu_resampler_state_size(void);
OPUS_AMD_EXPORT int opusgpu_resampler_batch_export_state(OpusGpuResamplerBatch *b, opus_int32 channel, void *state);
OPUS_AMD_EXPORT int opusgpu_resampler_batch_import_state(OpusGpuResamplerBatch *b, opus_int32 channel, const void *state);

/* silk_pitch_analysis_core (silk/fixed/pitch_analysis_core_FIX.c:82, prototype silk/SigProc_FIX.h:296-309) for n independent analysis
 * buffers: frames[n][(20 + 5*nb_subfr) * Fs_kHz] (the LPC residual the caller's find_pitch_lags step produced, e.g. with
 * opusgpu_silk_lpc_analysis_filter_batch), Fs_kHz in {8,12,16}, complexity 0..2 (SILK_PE_*_COMPLEX), nb_subfr in {2,4}.
 * OpusGpuPitchIn = the per-frame scalar arguments (prevLag, *LTPCorr_Q15 on entry, the two thresholds); OpusGpuPitchOut = everything the
 * reference writes back: pitch_out[4], *LTPCorr_Q15, *lagIndex, *contourIndex and its return value (0 voiced / 1 unvoiced). */
typedef struct { opus_int32 prevLag, LTPCorr_Q15, search_thres1_Q16, search_thres2_Q13; } OpusGpuPitchIn;
typedef struct { opus_int32 pitch[4]; opus_int32 LTPCorr_Q15; opus_int16 lagIndex; signed char contourIndex; signed char unvoiced; } OpusGpuPitchOut;
OPUS_AMD_EXPORT int opusgpu_silk_pitch_analysis_batch(int device, opus_int32 n, const opus_int16 *frames, const OpusGpuPitchIn *in, OpusGpuPitchOut *out,
      int Fs_kHz, int complexity, int nb_subfr);
OPUS_AMD_EXPORT int opusgpu_silk_pitch_analysis_batch_dev(int device, opus_int32 n, const opus_int16 *d_frames, const OpusGpuPitchIn *d_in, OpusGpuPitchOut *d_out,
      int Fs_kHz, int complexity, int nb_subfr, void *hip_stream);

/* ================= device-resident batch of multistream encoders (BASELINE config 5) =================
 * B encoders with one layout, i.e. B x opus_multistream_encoder_create(Fs, channels, streams, coupled_streams, mapping, application)
 * (reference include/opus_multistream.h:260, src/opus_multistream_encoder.c:841-1060), stepped together with their B x streams elementary encoders resident in HBM:
 * channel extraction, the elementary encodes and the self-delimited packing (RFC 6716 Appendix B) are launches on one HIP stream, no host round trip.
 * mapping_family 0 / 255 (plain layouts), 1 (surround: the Vorbis layouts with their per-frame masking analysis, reference src/opus_multistream_encoder.c:230, on the
 * device; pass the streams / coupled streams opus_multistream_surround_encoder_create reports), 2 (ambisonics layouts: CELT-only elementary encoders) or 3 (projection: the layout and mixing matrix of
 * opus_projection_ambisonics_encoder_create, reference src/opus_projection_encoder.c:176, mixed on the device as src/mapping_matrix.c:148 does; `mapping` is ignored).  max_data_bytes as in opus_multistream_encode: when it is large enough for every stream to be offered its own cap
 * ((streams - 1) * 1279 + 7662 + 3 * streams + 8 for frames <= 20 ms) the streams are independent and a frame-step is two encode launches; a tighter buffer chains the
 * streams' byte budgets as the reference does (src/opus_multistream_encoder.c:1016-1027) and the call steps through the streams in order on the device (2 x streams
 * launches).  Hard CBR (OPUS_SET_VBR(0)) always takes the chained form: the packet is the bitrate's size, the last stream's rate follows from the bytes the others left and its
 * packet is padded out to them (:918-927, :1027, :1048).  Packets, lengths and per-encoder error codes are the reference's in every case. */
typedef struct OpusGpuMsEncBatch OpusGpuMsEncBatch;
OPUS_AMD_EXPORT OpusGpuMsEncBatch *opusgpu_ms_enc_batch_create(opus_int32 nb_encoders, opus_int32 Fs, int channels, int mapping_family, int streams, int coupled_streams,
      const unsigned char *mapping, int application, int device, int *error);
OPUS_AMD_EXPORT void opusgpu_ms_enc_batch_destroy(OpusGpuMsEncBatch *b);
OPUS_AMD_EXPORT int opusgpu_ms_enc_batch_ctl(OpusGpuMsEncBatch *b, int request, opus_int32 value);            /* SET requests, applied to all encoders */
OPUS_AMD_EXPORT int opusgpu_ms_encode_batch_dev(OpusGpuMsEncBatch *b, const opus_int16 *d_pcm /* [B][frame_size][channels] */, int frame_size, unsigned char *d_out /* [B][out_stride] */,
      opus_int32 out_stride, opus_int32 max_data_bytes, opus_int32 *d_lens /* [B] */, opus_uint32 *d_final_range /* [B] */, void *hip_stream);
OPUS_AMD_EXPORT int opusgpu_ms_encode_batch(OpusGpuMsEncBatch *b, const opus_int16 *pcm, int frame_size, unsigned char *out, opus_int32 out_stride, opus_int32 max_data_bytes,
      opus_int32 *lens, opus_uint32 *final_range);
/* ORDERING (every batch object of this header): the *_dev entry points enqueue their launches on the HIP stream they are given (or the batch's own) and return; a batch
 * owns per-call device state besides the streams' records -- the work queues of its persistent waves, the lists of the decoder's look, the spectrum scratch, the
 * continuation records of the encoder's kernel pipeline -- so the calls on ONE batch must be ordered on ONE stream (or by events the caller records): two calls of one batch
 * in flight on different streams are undefined.  Calls on different batches are independent. */
/* ---- device-resident multistream / projection DECODER batches (opus_amd/csrc/opus_ms_dec_batch.h) ----
 * B decoders of opus_multistream_decoder_create(Fs, channels, streams, coupled_streams, mapping) (reference include/opus_multistream.h:461, src/opus_multistream_decoder.c:178): a
 * frame-step parses the B multistream packets, decodes the B x streams elementary packets and maps the decoded channels to the caller's interleaved output, all in HBM.
 * d_data [B][stride] packets of d_lens [B] bytes (0 = lost), d_pcm [B][frame_size][channels] int16, d_nsamples [B] = samples per channel or a negative OPUS_* code.
 * Limit of this batch: an elementary packet (one stream's share of a multistream packet, without its self-delimiting length field) may have at most 7,696 bytes -- six
 * coded frames of 1,275 bytes with their header: everything a 120 ms packet of this library's or the reference's encoder can hold; a packet repacketized beyond that is
 * answered OPUS_BAD_ARG for the WHOLE multistream packet, with every elementary decoder left as it was (the classic opus_multistream_decode has no such limit). */
typedef struct OpusGpuMsDecBatch OpusGpuMsDecBatch;
OPUS_AMD_EXPORT OpusGpuMsDecBatch *opusgpu_ms_dec_batch_create(opus_int32 nb_decoders, opus_int32 Fs, int channels, int streams, int coupled_streams, const unsigned char *mapping, int device, int *error);
/* ... of opus_projection_decoder_create (reference include/opus_projection.h:418, src/opus_projection_decoder.c:213): the same with the demixing matrix (src/mapping_matrix.c:257) applied on the device */
OPUS_AMD_EXPORT OpusGpuMsDecBatch *opusgpu_projection_dec_batch_create(opus_int32 nb_decoders, opus_int32 Fs, int channels, int streams, int coupled_streams, const unsigned char *demixing_matrix,
      opus_int32 demixing_matrix_size, int device, int *error);
OPUS_AMD_EXPORT void opusgpu_ms_dec_batch_destroy(OpusGpuMsDecBatch *b);
OPUS_AMD_EXPORT int opusgpu_ms_decode_batch_dev(OpusGpuMsDecBatch *b, const unsigned char *d_data, opus_int32 stride, const opus_int32 *d_lens, opus_int16 *d_pcm, int frame_size,
      opus_int32 *d_nsamples, opus_uint32 *d_final_range, void *hip_stream);
OPUS_AMD_EXPORT int opusgpu_ms_decode_batch(OpusGpuMsDecBatch *b, const unsigned char *data, opus_int32 stride, const opus_int32 *lens, opus_int16 *pcm, int frame_size,
      opus_int32 *nsamples, opus_uint32 *final_range);

#ifdef __cplusplus
}
#endif
#endif

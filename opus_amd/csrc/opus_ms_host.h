/* opus_ms_host.h — libopus multistream API (reference include/opus_multistream.h, src/opus_multistream.c,
 * src/opus_multistream_encoder.c, src/opus_multistream_decoder.c) on top of the batch kernels: the streams of one multistream
 * frame are independent CELT encodes / decodes, so ONE launch per group (coupled streams, mono streams) does a whole frame of up to
 * 255 channels.  Host code only orchestrates: layout, rate allocation, channel (de)interleaving, self-delimited packing.
 * Scope: every application (the CELT-only ones run the CELT kernel, VOIP / AUDIO / RESTRICTED_SILK the SILK-capable kernel), mapping families 0, 1 (the surround
 * masking analysis runs on the device, opus_surround*.h), 2 and 255; other families answer OPUS_UNIMPLEMENTED as the reference does.  A device-resident batch of
 * encoders with the packing on the device: opus_ms_batch.h. */
#ifndef OPUS_AMD_MS_HOST_H
#define OPUS_AMD_MS_HOST_H
#include <map>
#include "opus_surround_host.h"

struct OaLayout { int nb_channels, nb_streams, nb_coupled_streams; unsigned char mapping[256]; };
static int oa_validate_layout(const OaLayout *l)                       /* opus_multistream.c:40 */
{
   int max_channel = l->nb_streams + l->nb_coupled_streams;
   if (max_channel > 255) return 0;
   for (int i = 0; i < l->nb_channels; i++) if (l->mapping[i] >= max_channel && l->mapping[i] != 255) return 0;
   return 1;
}
static int oa_get_left(const OaLayout *l, int s, int prev) { for (int i = prev < 0 ? 0 : prev + 1; i < l->nb_channels; i++) if (l->mapping[i] == s * 2) return i; return -1; }
static int oa_get_right(const OaLayout *l, int s, int prev) { for (int i = prev < 0 ? 0 : prev + 1; i < l->nb_channels; i++) if (l->mapping[i] == s * 2 + 1) return i; return -1; }
static int oa_get_mono(const OaLayout *l, int s, int prev) { for (int i = prev < 0 ? 0 : prev + 1; i < l->nb_channels; i++) if (l->mapping[i] == s + l->nb_coupled_streams) return i; return -1; }
static int oa_validate_encoder_layout(const OaLayout *l)               /* opus_multistream_encoder.c:133 */
{
   for (int s = 0; s < l->nb_streams; s++) {
      if (s < l->nb_coupled_streams) { if (oa_get_left(l, s, -1) == -1 || oa_get_right(l, s, -1) == -1) return 0; }
      else if (oa_get_mono(l, s, -1) == -1) return 0;
   }
   return 1;
}
static unsigned oa_isqrt32(opus_uint32 v) { unsigned g = 0, b = 0x8000; while (b) { unsigned t = g | b; if ((opus_uint32)t * t <= v) g = t; b >>= 1; } return g; }
static int oa_validate_ambisonics(int nb_channels, int *nb_streams, int *nb_coupled)   /* opus_multistream_encoder.c:110 */
{
   if (nb_channels < 1 || nb_channels > 227) return 0;
   int order_plus_one = (int)oa_isqrt32((opus_uint32)nb_channels), acn = order_plus_one * order_plus_one, nd = nb_channels - acn;
   if (nd != 0 && nd != 2) return 0;
   if (nb_streams) *nb_streams = acn + (nd != 0);
   if (nb_coupled) *nb_coupled = nd != 0;
   return 1;
}

/* n records of `width` bytes, `hpitch` apart in the host blob <-> the contiguous device array: packed through one host buffer so that the transfer itself is one plain
 * contiguous hipMemcpy (a pitched copy from / to pageable host memory goes through the runtime's rectangle path; on the MI355X box a long series of such copies --
 * the reference's regression tests with 255-stream encoders -- ended in a GPU memory fault inside one of the runtime's own copy kernels, which never happens with
 * contiguous copies or with AMD_SERIALIZE_COPY=3) */
static int oa_rows_upload(void *d, const void *h, size_t hpitch, size_t width, int n)
{
   if (n == 1) { HIPCHECK(hipMemcpy(d, h, width, hipMemcpyHostToDevice)); return OPUS_OK; }
   std::vector<char> tmp(width * (size_t)n);
   for (int i = 0; i < n; i++) memcpy(tmp.data() + (size_t)i * width, (const char *)h + (size_t)i * hpitch, width);
   HIPCHECK(hipMemcpy(d, tmp.data(), tmp.size(), hipMemcpyHostToDevice));
   return OPUS_OK;
}
static int oa_rows_download(void *h, size_t hpitch, const void *d, size_t width, int n)
{
   if (n == 1) { HIPCHECK(hipMemcpy(h, d, width, hipMemcpyDeviceToHost)); return OPUS_OK; }
   std::vector<char> tmp(width * (size_t)n);
   HIPCHECK(hipMemcpy(tmp.data(), d, tmp.size(), hipMemcpyDeviceToHost));
   for (int i = 0; i < n; i++) memcpy((char *)h + (size_t)i * hpitch, tmp.data() + (size_t)i * width, width);
   return OPUS_OK;
}

/* process-wide batches, one per (stream count, channels) */
static std::mutex g_ms_mu;
static std::map<long, OpusGpuEncBatch *> g_ms_enc;
static std::map<long, OpusGpuDecBatch *> g_ms_dec;
static OpusGpuEncBatch *oa_ms_enc_batch(int n, int channels, int application, opus_int32 Fs, int *err)
{
   long key = (((long)n * 4 + channels) * 64 + Fs / 1000) * 2 + oa_app_is_sh(application);
   auto it = g_ms_enc.find(key);
   if (it != g_ms_enc.end()) return it->second;
   OpusGpuEncBatch *b = opusgpu_enc_batch_create(n, Fs, channels, application, oa_classic_device(), err);
   if (b) g_ms_enc[key] = b;
   return b;
}
static OpusGpuDecBatch *oa_ms_dec_batch(int n, int channels, opus_int32 Fs, int *err)
{
   long key = ((long)n * 4 + channels) * 64 + Fs / 1000;
   auto it = g_ms_dec.find(key);
   if (it != g_ms_dec.end()) return it->second;
   OpusGpuDecBatch *b = opusgpu_dec_batch_create(n, Fs, channels, oa_classic_device(), err);
   if (b) g_ms_dec[key] = b;
   return b;
}

#define OA_MS_MAGIC 0x4f414d53u
enum { OA_MAP_NONE = 0, OA_MAP_SURROUND = 1, OA_MAP_AMBISONICS = 2 };
/* one elementary encoder: the CELT-only record or the SILK-capable one, by application (opus_multistream_encoder_get_size does not know the
 * application, so every record has room for either) */
typedef OpusEncoder OaMsRec;                 /* a complete classic encoder object: OPUS_MULTISTREAM_GET_ENCODER_STATE hands it out (opus_multistream.h:74) */
struct OpusMSEncoder {
   opus_uint32 magic; opus_int32 Fs, application, bitrate_bps, mapping_type, lfe_stream;
   OaLayout layout;
   opus_int32 kind, variable_duration;   /* kind 1: records are OaShStream; OPUS_SET_EXPERT_FRAME_DURATION of the multistream encoder itself (opus_multistream_encoder.c:483, :888) */
   OaMsRec streams[1];             /* nb_streams records: coupled streams first, then mono (flat, memcpy-able) */
};
static int oa_ms_rec_init(OaMsRec *r, int kind, opus_int32 Fs, int ch, int application) { (void)kind; return opus_encoder_init(r, Fs, ch, application); }
static int oa_ms_rec_set(OaMsRec *r, int kind, int request, opus_int32 v) { return kind ? sh_ctl_set(&r->sh, request, v) : oa_ctl_set(&r->s, request, v); }
static int oa_ms_rec_get(const OaMsRec *r, int kind, int request, opus_int32 *v) { return kind ? sh_ctl_get(&r->sh, request, v) : oa_ctl_get(&r->s, request, v); }
struct OpusMSDecoder {
   opus_uint32 magic; opus_int32 Fs;
   OaLayout layout;
   opus_int32 pad[2];
   OpusDecoder streams[1];      /* complete classic decoder objects: OPUS_MULTISTREAM_GET_DECODER_STATE hands them out */
};
#define OA_MS_FRAME_TMP (6 * 1275 + 12)

extern "C" {
/* the masking analysis of the surround layouts on its own (surround_analysis, src/opus_multistream_encoder.c:230): len samples of `channels` (3..8, vorbis order)
 * interleaved int16 at Fs; mem[channels][120] / preemph_mem[channels] = the analysis memory (in and out); bandSMR[channels][21] = per-channel signal-to-mask ratios, Q24 */
int opusgpu_surround_analysis(const opus_int16 *pcm, int len, int channels, opus_int32 Fs, opus_int32 *mem, opus_int32 *preemph_mem, opus_int32 *bandSMR)
{
   if (!pcm || !mem || !preemph_mem || !bandSMR || channels < 3 || channels > 8 || len <= 0 || !oa_fs_ok(Fs)) return OPUS_BAD_ARG;
   std::lock_guard<std::mutex> lock(g_ms_mu);
   return oa_surround_analysis(pcm, len, channels, Fs, mem, preemph_mem, bandSMR);
}
opus_int32 opus_multistream_encoder_get_size(int nb_streams, int nb_coupled_streams)
{
   if (nb_streams < 1 || nb_coupled_streams > nb_streams || nb_coupled_streams < 0) return 0;
   return (opus_int32)(sizeof(OpusMSEncoder) + (size_t)(nb_streams - 1) * sizeof(OaMsRec));
}
/* surround encoders carry the masking analysis state behind the stream records: pre-emphasis memory [channels], window memory [channels][120] (:96-108) */
static opus_int32 *oa_ms_preemph_mem(OpusMSEncoder *st) { return (opus_int32 *)(void *)((char *)st + opus_multistream_encoder_get_size(st->layout.nb_streams, st->layout.nb_coupled_streams)); }
static opus_int32 *oa_ms_window_mem(OpusMSEncoder *st) { return oa_ms_preemph_mem(st) + st->layout.nb_channels; }
static int oa_ms_encoder_init_impl(OpusMSEncoder *st, opus_int32 Fs, int channels, int streams, int coupled_streams, const unsigned char *mapping, int application, int mapping_type, int lfe_stream)
{
   if (channels > 255 || channels < 1 || coupled_streams > streams || streams < 1 || coupled_streams < 0 || streams > 255 - coupled_streams || streams + coupled_streams > channels)
      return OPUS_BAD_ARG;
   const int kind = oa_app_is_sh(application);
   int r;
   { OaMsRec *probe = new OaMsRec; r = oa_ms_rec_init(probe, kind, Fs, 2, application); delete probe; }
   if (r != OPUS_OK) return r;
   memset(st, 0, sizeof(OpusMSEncoder) - sizeof(OaMsRec));
   st->kind = kind; st->variable_duration = OPUS_FRAMESIZE_ARG;
   st->magic = OA_MS_MAGIC; st->Fs = Fs; st->application = application; st->bitrate_bps = OPUS_AUTO; st->mapping_type = mapping_type; st->lfe_stream = lfe_stream;
   st->layout.nb_channels = channels; st->layout.nb_streams = streams; st->layout.nb_coupled_streams = coupled_streams;
   for (int i = 0; i < channels; i++) st->layout.mapping[i] = mapping[i];
   if (!oa_validate_layout(&st->layout) || !oa_validate_encoder_layout(&st->layout)) return OPUS_BAD_ARG;
   if (mapping_type == OA_MAP_AMBISONICS && !oa_validate_ambisonics(channels, NULL, NULL)) return OPUS_BAD_ARG;
   for (int s = 0; s < streams; s++) {
      r = oa_ms_rec_init(&st->streams[s], kind, Fs, s < coupled_streams ? 2 : 1, application);
      if (r != OPUS_OK) return r;
      if (s == lfe_stream) oa_ms_rec_set(&st->streams[s], kind, OPUS_SET_LFE_REQUEST, 1);
   }
   if (mapping_type == OA_MAP_SURROUND) memset(oa_ms_preemph_mem(st), 0, sizeof(opus_int32) * (size_t)channels * (120 + 1));
   return OPUS_OK;
}
int opus_multistream_encoder_init(OpusMSEncoder *st, opus_int32 Fs, int channels, int streams, int coupled_streams, const unsigned char *mapping, int application)
{
   if (!st || !mapping) return OPUS_BAD_ARG;
   return oa_ms_encoder_init_impl(st, Fs, channels, streams, coupled_streams, mapping, application, OA_MAP_NONE, -1);
}
OpusMSEncoder *opus_multistream_encoder_create(opus_int32 Fs, int channels, int streams, int coupled_streams, const unsigned char *mapping, int application, int *error)
{
   if (channels > 255 || channels < 1 || coupled_streams > streams || streams < 1 || coupled_streams < 0 || streams > 255 - coupled_streams || streams + coupled_streams > channels || !mapping) {
      if (error) *error = OPUS_BAD_ARG;
      return NULL;
   }
   OpusMSEncoder *st = (OpusMSEncoder *)malloc((size_t)opus_multistream_encoder_get_size(streams, coupled_streams));
   if (!st) { if (error) *error = OPUS_ALLOC_FAIL; return NULL; }
   int r = opus_multistream_encoder_init(st, Fs, channels, streams, coupled_streams, mapping, application);
   if (error) *error = r;
   if (r != OPUS_OK) { free(st); return NULL; }
   return st;
}
static int oa_surround_layout(int channels, int mapping_family, int *streams, int *coupled_streams, unsigned char *mapping, int *mapping_type, int *lfe_stream = nullptr)
{
   if (lfe_stream) *lfe_stream = -1;
   if (channels > 255 || channels < 1) return OPUS_BAD_ARG;
   if (mapping_family == 0) {
      if (channels == 1) { *streams = 1; *coupled_streams = 0; mapping[0] = 0; }
      else if (channels == 2) { *streams = 1; *coupled_streams = 1; mapping[0] = 0; mapping[1] = 1; }
      else return OPUS_UNIMPLEMENTED;
   } else if (mapping_family == 1 && channels <= 8 && channels >= 1) {
      /* the Vorbis channel orders as (streams, coupled streams, mapping) (RFC 7845 §5.1.1.2; src/opus_multistream_encoder.c:53-62) */
      static const struct { unsigned char ns, nc, map[8]; } vorbis[8] = {{1, 0, {0}}, {1, 1, {0, 1}}, {2, 1, {0, 2, 1}}, {2, 2, {0, 1, 2, 3}}, {3, 2, {0, 4, 1, 2, 3}},
         {4, 2, {0, 4, 1, 2, 3, 5}}, {4, 3, {0, 4, 1, 2, 3, 5, 6}}, {5, 3, {0, 6, 1, 2, 3, 4, 5, 7}}};
      *streams = vorbis[channels - 1].ns; *coupled_streams = vorbis[channels - 1].nc;
      for (int i = 0; i < channels; i++) mapping[i] = vorbis[channels - 1].map[i];
      if (lfe_stream && channels >= 6) *lfe_stream = *streams - 1;
   } else if (mapping_family == 255) {
      *streams = channels; *coupled_streams = 0;
      for (int i = 0; i < channels; i++) mapping[i] = (unsigned char)i;
   } else if (mapping_family == 2) {
      if (!oa_validate_ambisonics(channels, streams, coupled_streams)) return OPUS_BAD_ARG;
      for (int i = 0; i < (*streams - *coupled_streams); i++) mapping[i] = (unsigned char)(i + (*coupled_streams * 2));
      for (int i = 0; i < *coupled_streams * 2; i++) mapping[i + (*streams - *coupled_streams)] = (unsigned char)i;
   } else return OPUS_UNIMPLEMENTED;
   *mapping_type = mapping_family == 2 ? OA_MAP_AMBISONICS : (mapping_family == 1 && channels > 2) ? OA_MAP_SURROUND : OA_MAP_NONE;
   return OPUS_OK;
}
opus_int32 opus_multistream_surround_encoder_get_size(int channels, int mapping_family)
{
   int streams, coupled, mt; unsigned char mapping[256];
   if (oa_surround_layout(channels, mapping_family, &streams, &coupled, mapping, &mt) != OPUS_OK) return 0;
   return opus_multistream_encoder_get_size(streams, coupled) + (channels > 2 ? channels * (120 + 1) * (opus_int32)sizeof(opus_int32) : 0);
}
int opus_multistream_surround_encoder_init(OpusMSEncoder *st, opus_int32 Fs, int channels, int mapping_family, int *streams, int *coupled_streams, unsigned char *mapping, int application)
{
   int mt, lfe;
   if (!st || !streams || !coupled_streams || !mapping) return OPUS_BAD_ARG;
   int r = oa_surround_layout(channels, mapping_family, streams, coupled_streams, mapping, &mt, &lfe);
   if (r != OPUS_OK) return r;
   return oa_ms_encoder_init_impl(st, Fs, channels, *streams, *coupled_streams, mapping, application, mt, lfe);
}
OpusMSEncoder *opus_multistream_surround_encoder_create(opus_int32 Fs, int channels, int mapping_family, int *streams, int *coupled_streams, unsigned char *mapping, int application, int *error)
{
   int mt, r, lfe;
   if (!streams || !coupled_streams || !mapping) { if (error) *error = OPUS_BAD_ARG; return NULL; }
   r = oa_surround_layout(channels, mapping_family, streams, coupled_streams, mapping, &mt, &lfe);
   if (r != OPUS_OK) { if (error) *error = r; return NULL; }
   OpusMSEncoder *st = (OpusMSEncoder *)malloc((size_t)opus_multistream_surround_encoder_get_size(channels, mapping_family));
   if (!st) { if (error) *error = OPUS_ALLOC_FAIL; return NULL; }
   r = oa_ms_encoder_init_impl(st, Fs, channels, *streams, *coupled_streams, mapping, application, mt, lfe);
   if (error) *error = r;
   if (r != OPUS_OK) { free(st); return NULL; }
   return st;
}
void opus_multistream_encoder_destroy(OpusMSEncoder *st) { free(st); }

/* surround_rate_allocation / ambisonics_rate_allocation / rate_allocation, opus_multistream_encoder.c:702-832 */
static opus_int32 oa_ms_rate_allocation(const OpusMSEncoder *st, opus_int32 *rate, int frame_size)
{
   const opus_int32 Fs = st->Fs;
   const int ns = st->layout.nb_streams, nc = st->layout.nb_coupled_streams;
   if (st->mapping_type == OA_MAP_AMBISONICS) {
      opus_int32 total_rate;
      const int nb_channels = ns + nc;
      if (st->bitrate_bps == OPUS_AUTO) total_rate = (nc + ns) * (Fs + 60 * Fs / frame_size) + ns * (opus_int32)15000;
      else if (st->bitrate_bps == OPUS_BITRATE_MAX) total_rate = nb_channels * 750000;
      else total_rate = st->bitrate_bps;
      for (int i = 0; i < ns; i++) rate[i] = total_rate / ns;
   } else {
      const int nb_lfe = st->lfe_stream != -1, nb_uncoupled = ns - nc - nb_lfe, nb_normal = 2 * nc + nb_uncoupled;
      opus_int32 channel_offset = 40 * (50 > Fs / frame_size ? 50 : Fs / frame_size), bitrate;
      if (st->bitrate_bps == OPUS_AUTO) bitrate = nb_normal * (channel_offset + Fs + 10000) + 8000 * nb_lfe;
      else if (st->bitrate_bps == OPUS_BITRATE_MAX) bitrate = nb_normal * 750000 + nb_lfe * 128000;
      else bitrate = st->bitrate_bps;
      int lfe_offset = (bitrate / 20 < 3000 ? bitrate / 20 : 3000) + 15 * (50 > Fs / frame_size ? 50 : Fs / frame_size);
      int stream_offset = (bitrate - channel_offset * nb_normal - lfe_offset * nb_lfe) / nb_normal / 2;
      stream_offset = stream_offset < 0 ? 0 : (stream_offset > 20000 ? 20000 : stream_offset);
      const int coupled_ratio = 512, lfe_ratio = 32;
      int total = (nb_uncoupled << 8) + coupled_ratio * nc + nb_lfe * lfe_ratio;
      opus_int32 channel_rate = (opus_int32)(256 * (long long)(bitrate - lfe_offset * nb_lfe - stream_offset * (nc + nb_uncoupled) - channel_offset * nb_normal) / total);
      for (int i = 0; i < ns; i++) {
         opus_int32 r;
         if (i < nc) { r = stream_offset + (channel_rate * coupled_ratio >> 8); rate[i] = 2 * channel_offset + (r > 0 ? r : 0); }
         else if (i != st->lfe_stream) { r = stream_offset + channel_rate; rate[i] = channel_offset + (r > 0 ? r : 0); }
         else { r = lfe_offset + (channel_rate * lfe_ratio >> 8); rate[i] = r > 0 ? r : 0; }
      }
   }
   opus_int32 rate_sum = 0;
   for (int i = 0; i < ns; i++) { if (rate[i] < 500) rate[i] = 500; rate_sum += rate[i]; }
   return rate_sum;
}

/* encode n streams of one group (all `ch`-channel) in one launch; states are loaded from / stored back to the flat blob */
static int oa_ms_encode_group(OaMsRec *states, int kind, opus_int32 Fs, int n, int ch, int application, const opus_int16 *pcm, int frame_size, opus_int32 max_data_bytes,
      unsigned char *out /* [n][stride] */, opus_int32 stride, opus_int32 *lens, opus_uint32 *rngs, const opus_int32 *apcm = nullptr /* signal-domain view for the analysis, or NULL */,
      int analysis_frame_size = 0 /* samples per channel of a stream's row of pcm / apcm (>= frame_size); 0 = frame_size */)
{
   int err = OPUS_OK;
   OpusGpuEncBatch *b = oa_ms_enc_batch(n, ch, application, Fs, &err);
   if (!b) return err == OPUS_OK ? OPUS_INTERNAL_ERROR : err;
   HIPCHECK(hipSetDevice(b->device));
   HIPCHECK(hipStreamSynchronize(b->stream));
   { const int ru = kind ? oa_rows_upload(b->d_sh, &states[0].sh, sizeof(OaMsRec), sizeof(OaShStream), n) : oa_rows_upload(b->d_streams, &states[0].s, sizeof(OaMsRec), sizeof(OaStream), n); if (ru != OPUS_OK) return ru; }
   /* the batch is shared by every multistream encoder of this shape: what the host side of a launch derives from its mirror of the configuration (the frame sizes
    * the application accepts, whether the launch can skip the CELT arena) must follow the records just uploaded, not the encoder the batch was created for */
   b->application = application; b->pipeline = states[0].pipeline_p2 ? (int)states[0].pipeline_p2 - 2 : -1;
   b->tr_pre = (states[0].launch_opts & 3) ? (int)(states[0].launch_opts & 3) - 2 : -1; b->pvq_stage = ((states[0].launch_opts >> 2) & 3) ? (int)((states[0].launch_opts >> 2) & 3) - 2 : -1;
   if (kind) { for (int i = 0; i < n; i++) b->h_sh[i].cfg = states[i].sh.cfg; b->cfg_dirty = true; b->any_fec = -1; }
   else for (int i = 0; i < n; i++) b->h_streams[i].cfg = states[i].s.cfg;
   int r = opusgpu_encode_batch_lookahead(b, pcm, apcm, frame_size, analysis_frame_size > frame_size ? analysis_frame_size : frame_size, out, stride, max_data_bytes, lens, rngs);
   if (r != OPUS_OK) return r;
   return kind ? oa_rows_download(&states[0].sh, sizeof(OaMsRec), b->d_sh, sizeof(OaShStream), n) : oa_rows_download(&states[0].s, sizeof(OaMsRec), b->d_streams, sizeof(OaStream), n);
}

/* opus_multistream_encode_native, opus_multistream_encoder.c:841 (int16 samples; depth = 16 or 24: the lsb_depth of the entry point) */
static int oa_ms_encode_native(OpusMSEncoder *st, const opus_int16 *pcm, int analysis_frame_size, unsigned char *data, opus_int32 max_data_bytes, int depth, const opus_int32 *apcm = nullptr)
{
   if (!st || st->magic != OA_MS_MAGIC || !pcm || !data) return OPUS_BAD_ARG;
   const opus_int32 Fs = st->Fs;
   const int ns = st->layout.nb_streams, nc = st->layout.nb_coupled_streams, nm = ns - nc, nch = st->layout.nb_channels;
   const int kind = st->kind;
   const int frame_size = st->variable_duration == 0 ? -1 : (int)oa_frame_size_select(st->application, analysis_frame_size, st->variable_duration, Fs);   /* (0 is no legal value: frame_size_select, opus_encoder.c:827-849) */
   if (frame_size <= 0) return OPUS_BAD_ARG;
   for (int s = 0; s < ns; s++) { if (kind) st->streams[s].sh.cfg.input_depth = depth; else st->streams[s].s.cfg.input_depth = depth; }
   const int vbr = kind ? st->streams[0].sh.cfg.use_vbr : st->streams[0].s.cfg.use_vbr;
   opus_int32 smallest_packet = ns * 2 - 1;
   if (Fs / frame_size == 10) smallest_packet += ns;
   if (max_data_bytes < smallest_packet) return OPUS_BUFFER_TOO_SMALL;
   std::lock_guard<std::mutex> lock(g_ms_mu);
   std::vector<opus_int32> bitrates((size_t)ns);
   opus_int32 rate_sum = oa_ms_rate_allocation(st, bitrates.data(), frame_size);
   if (!vbr) {
      if (st->bitrate_bps == OPUS_AUTO) { opus_int32 m = (rate_sum * 6 / (6 * Fs / frame_size) + 4) / 8; if (m < max_data_bytes) max_data_bytes = m; }
      else if (st->bitrate_bps != OPUS_BITRATE_MAX) {
         opus_int32 m = (st->bitrate_bps * 6 / (6 * Fs / frame_size) + 4) / 8;
         if (m < smallest_packet) m = smallest_packet;
         if (m < max_data_bytes) max_data_bytes = m;
      }
   }
   /* the masking analysis of the surround layouts (:912-915): per-channel signal-to-mask ratios, handed to each elementary encoder as its energy mask (:1014) */
   std::vector<opus_int32> bandSMR((size_t)21 * nch);
   const bool surround = st->mapping_type == OA_MAP_SURROUND && st->application != OPUS_APPLICATION_RESTRICTED_SILK;
   if (surround) { const int r0 = oa_surround_analysis(pcm, frame_size, nch, Fs, oa_ms_window_mem(st), oa_ms_preemph_mem(st), bandSMR.data()); if (r0 != OPUS_OK) return r0; }
   for (int s = 0; s < ns; s++) {
      OaMsRec *rec = &st->streams[s];
      oa_ms_rec_set(rec, kind, OPUS_SET_BITRATE_REQUEST, bitrates[s]);
      if (st->mapping_type == OA_MAP_SURROUND) {                                          /* :965-985 */
         opus_int32 equiv_rate = st->bitrate_bps;
         if (frame_size * 50 < Fs) equiv_rate -= 60 * (Fs / frame_size - 50) * nch;
         oa_ms_rec_set(rec, kind, OPUS_SET_BANDWIDTH_REQUEST, equiv_rate > 10000 * nch ? OPUS_BANDWIDTH_FULLBAND : equiv_rate > 7000 * nch ? OPUS_BANDWIDTH_SUPERWIDEBAND : equiv_rate > 5000 * nch ? OPUS_BANDWIDTH_WIDEBAND : OPUS_BANDWIDTH_NARROWBAND);
         if (s < nc) { oa_ms_rec_set(rec, kind, OPUS_SET_FORCE_MODE_REQUEST, OPUS_MODE_CELT_ONLY); oa_ms_rec_set(rec, kind, OPUS_SET_FORCE_CHANNELS_REQUEST, 2); }   /* keep the spatial image: stereo CELT on coupled streams */
      } else if (st->mapping_type == OA_MAP_AMBISONICS) oa_ms_rec_set(rec, kind, OPUS_SET_FORCE_MODE_REQUEST, OPUS_MODE_CELT_ONLY);
      if (surround) {
         opus_int32 *mask = kind ? rec->sh.energy_mask : rec->s.energy_mask;
         if (s < nc) { const int l = oa_get_left(&st->layout, s, -1), r = oa_get_right(&st->layout, s, -1); for (int i = 0; i < 21; i++) { mask[i] = bandSMR[21 * l + i]; mask[21 + i] = bandSMR[21 * r + i]; } }
         else { const int c = oa_get_mono(&st->layout, s, -1); for (int i = 0; i < 21; i++) mask[i] = bandSMR[21 * c + i]; }
         if (kind) { rec->sh.cfg.energy_mask_on = 1; rec->sh.s.celt_mask_cleared = 0; } else rec->s.energy_mask_on = 1;
      }
   }
   /* channel de-interleave into the two groups: the whole buffer the caller handed over (analysis_frame_size samples: the look-ahead the elementary encoders' analyses see,
    * opus_multistream_encoder.c:1016-1040 passes pcm and analysis_frame_size on), of which the first frame_size samples are coded */
   const int afs = analysis_frame_size > frame_size ? analysis_frame_size : frame_size;
   std::vector<opus_int16> pc((size_t)nc * afs * 2 + 2), pm((size_t)nm * afs + 1);
   for (int s = 0; s < nc; s++) {
      int l = oa_get_left(&st->layout, s, -1), r = oa_get_right(&st->layout, s, -1);
      opus_int16 *d = pc.data() + (size_t)s * afs * 2;
      for (int i = 0; i < afs; i++) { d[2 * i] = pcm[(size_t)i * nch + l]; d[2 * i + 1] = pcm[(size_t)i * nch + r]; }
   }
   for (int s = 0; s < nm; s++) {
      int c = oa_get_mono(&st->layout, nc + s, -1);
      opus_int16 *d = pm.data() + (size_t)s * afs;
      for (int i = 0; i < afs; i++) d[i] = pcm[(size_t)i * nch + c];
   }
   /* the 24-bit / float entry points: the same split of the signal-domain samples the elementary encoders' analyses see (downmix_int24 / downmix_float with the stream's c1, c2) */
   std::vector<opus_int32> ac, am;
   if (apcm) {
      ac.resize((size_t)nc * afs * 2 + 2); am.resize((size_t)nm * afs + 1);
      for (int s = 0; s < nc; s++) {
         int l = oa_get_left(&st->layout, s, -1), r = oa_get_right(&st->layout, s, -1);
         opus_int32 *d = ac.data() + (size_t)s * afs * 2;
         for (int i = 0; i < afs; i++) { d[2 * i] = apcm[(size_t)i * nch + l]; d[2 * i + 1] = apcm[(size_t)i * nch + r]; }
      }
      for (int s = 0; s < nm; s++) {
         int c = oa_get_mono(&st->layout, nc + s, -1);
         opus_int32 *d = am.data() + (size_t)s * afs;
         for (int i = 0; i < afs; i++) d[i] = apcm[(size_t)i * nch + c];
      }
   }
   const opus_int32 *apc = apcm ? ac.data() : nullptr, *apm = apcm ? am.data() : nullptr;
   const opus_int32 stride = (oa_enc_out_stride_needed(Fs, frame_size, OA_MS_FRAME_TMP) + 15) & ~15;     /* a stream is offered at most MS_FRAME_TMP bytes (:1024); multi-frame calls stage their frames in the output slot */
   std::vector<unsigned char> pk((size_t)ns * stride);
   std::vector<opus_int32> lens((size_t)ns);
   std::vector<opus_uint32> rngs((size_t)ns);
   /* The byte budget handed to stream s is max_data_bytes minus what the previous streams used (:1016-1026).  When the caller's buffer is so
    * generous that every stream would be offered at least the encoder's own cap, the budgets are identical and all streams of a group go in
    * ONE launch; otherwise (tight buffer, or hard CBR where the last stream absorbs the remainder) the streams are stepped in order. */
   const int nf = frame_size > Fs / 50 ? (frame_size * 50 + Fs - 1) / Fs : 1;                      /* coded frames a stream may put into its packet */
   const long long worst = (long long)(ns - 1) * (1276 * nf + 3) + OA_MS_FRAME_TMP + 3 * ns + 8;
   const bool parallel = vbr && max_data_bytes >= worst;
   int r = OPUS_OK;
   opus_int32 tot_size = 0;
   unsigned char *out = data;
   if (parallel) {
      if (nc) r = oa_ms_encode_group(st->streams, kind, Fs, nc, 2, st->application, pc.data(), frame_size, 1276 * 6, pk.data(), stride, lens.data(), rngs.data(), apc, afs);
      if (r == OPUS_OK && nm) r = oa_ms_encode_group(st->streams + nc, kind, Fs, nm, 1, st->application, pm.data(), frame_size, 1276 * 6, pk.data() + (size_t)nc * stride, stride, lens.data() + nc, rngs.data() + nc, apm, afs);
      if (r != OPUS_OK) return r;
   }
   for (int s = 0; s < ns; s++) {
      OpusRepacketizer rp;
      opus_repacketizer_init(&rp);
      if (!parallel) {
         opus_int32 curr_max = max_data_bytes - tot_size;
         int resv = 2 * (ns - s - 1) - 1;
         curr_max -= resv > 0 ? resv : 0;
         if (Fs / frame_size == 10) curr_max -= ns - s - 1;
         if (curr_max > OA_MS_FRAME_TMP) curr_max = OA_MS_FRAME_TMP;
         if (s != ns - 1) curr_max -= curr_max > 253 ? 2 : 1;
         if (!vbr && s == ns - 1) (void)oa_ms_rec_set(&st->streams[s], kind, OPUS_SET_BITRATE_REQUEST, curr_max * 8 * (6 * Fs / frame_size) / 6);
         if (curr_max <= 0) return OPUS_BUFFER_TOO_SMALL;
         if (s < nc) r = oa_ms_encode_group(st->streams + s, kind, Fs, 1, 2, st->application, pc.data() + (size_t)s * afs * 2, frame_size, curr_max, pk.data() + (size_t)s * stride, stride, &lens[s], &rngs[s], apc ? apc + (size_t)s * afs * 2 : nullptr, afs);
         else r = oa_ms_encode_group(st->streams + s, kind, Fs, 1, 1, st->application, pm.data() + (size_t)(s - nc) * afs, frame_size, curr_max, pk.data() + (size_t)s * stride, stride, &lens[s], &rngs[s], apm ? apm + (size_t)(s - nc) * afs : nullptr, afs);
         if (r != OPUS_OK) return r;
      }
      if (lens[s] < 0) return lens[s];
      if (opus_repacketizer_cat(&rp, pk.data() + (size_t)s * stride, lens[s]) != OPUS_OK) return OPUS_INTERNAL_ERROR;
      opus_int32 len = oa_repacketizer_out_range_impl(&rp, 0, opus_repacketizer_get_nb_frames(&rp), out, max_data_bytes - tot_size, s != ns - 1, !vbr && s == ns - 1);
      if (len < 0) return len;
      out += len;
      tot_size += len;
   }
   return tot_size;
}
int opus_multistream_encode(OpusMSEncoder *st, const opus_int16 *pcm, int frame_size, unsigned char *data, opus_int32 max_data_bytes)
{
   return oa_ms_encode_native(st, pcm, frame_size, data, max_data_bytes, 16);
}
static int oa_ms_encoder_ctl_va(OpusMSEncoder *st, int request, va_list ap)
{
   if (!st || st->magic != OA_MS_MAGIC) return OPUS_BAD_ARG;
   int ret = OPUS_OK;
   const int ns = st->layout.nb_streams;
   switch (request) {
   case OPUS_SET_BITRATE_REQUEST: {
      opus_int32 value = va_arg(ap, opus_int32);
      if (value != OPUS_AUTO && value != OPUS_BITRATE_MAX) {
         if (value <= 0) { ret = OPUS_BAD_ARG; break; }
         opus_int32 lo = 500 * st->layout.nb_channels, hi = 750000 * st->layout.nb_channels;
         value = value < lo ? lo : (value > hi ? hi : value);
      }
      st->bitrate_bps = value;
   } break;
   case OPUS_GET_BITRATE_REQUEST: {
      opus_int32 *value = va_arg(ap, opus_int32 *);
      if (!value) { ret = OPUS_BAD_ARG; break; }
      *value = 0;
      for (int s = 0; s < ns; s++) { opus_int32 r = 0; oa_ms_rec_get(&st->streams[s], st->kind, request, &r); *value += r; }
   } break;
   case OPUS_GET_FINAL_RANGE_REQUEST: {
      opus_uint32 *value = va_arg(ap, opus_uint32 *);
      if (!value) { ret = OPUS_BAD_ARG; break; }
      *value = 0;
      for (int s = 0; s < ns; s++) *value ^= st->kind ? st->streams[s].sh.s.rangeFinal : st->streams[s].s.st.s.rangeFinal;
   } break;
   case OPUS_RESET_STATE:
      for (int s = 0; s < ns && ret == OPUS_OK; s++) ret = oa_ms_rec_set(&st->streams[s], st->kind, request, 0);
      break;
   case 5120 /* OPUS_MULTISTREAM_GET_ENCODER_STATE (opus_multistream.h:74) */: {
      const opus_int32 id = va_arg(ap, opus_int32);
      OpusEncoder **value = va_arg(ap, OpusEncoder **);
      if (id < 0 || id >= ns || !value) { ret = OPUS_BAD_ARG; break; }
      *value = &st->streams[id];
   } break;
   /* the requests the reference's switch knows (opus_multistream_encoder.c:1196-1330); everything else is OPUS_UNIMPLEMENTED there and here */
   case OPUS_GET_LSB_DEPTH_REQUEST: case OPUS_GET_VBR_REQUEST: case OPUS_GET_APPLICATION_REQUEST: case OPUS_GET_BANDWIDTH_REQUEST: case OPUS_GET_COMPLEXITY_REQUEST:
   case OPUS_GET_PACKET_LOSS_PERC_REQUEST: case OPUS_GET_DTX_REQUEST: case OPUS_GET_VOICE_RATIO_REQUEST: case OPUS_GET_VBR_CONSTRAINT_REQUEST: case OPUS_GET_SIGNAL_REQUEST:
   case OPUS_GET_LOOKAHEAD_REQUEST: case OPUS_GET_SAMPLE_RATE_REQUEST: case OPUS_GET_INBAND_FEC_REQUEST: case OPUS_GET_FORCE_CHANNELS_REQUEST: case OPUS_GET_PREDICTION_DISABLED_REQUEST:
   case OPUS_GET_PHASE_INVERSION_DISABLED_REQUEST: case 4057 /* OPUS_GET_QEXT: the elementary encoder answers (unimplemented without ENABLE_QEXT) */: case 11901 /* OPUS_AMD_GET_FLOAT_ANALYSIS (private) */: case 11903 /* OPUS_AMD_GET_KERNEL_PIPELINE (private) */: case 11907: case 11909: {
      opus_int32 *value = va_arg(ap, opus_int32 *);              /* answered by the first stream (:1196-1219) */
      ret = opus_encoder_ctl(&st->streams[0], request, value);
   } break;
   case OPUS_SET_LSB_DEPTH_REQUEST: case OPUS_SET_COMPLEXITY_REQUEST: case OPUS_SET_VBR_REQUEST: case OPUS_SET_VBR_CONSTRAINT_REQUEST: case OPUS_SET_MAX_BANDWIDTH_REQUEST:
   case OPUS_SET_BANDWIDTH_REQUEST: case OPUS_SET_SIGNAL_REQUEST: case OPUS_SET_INBAND_FEC_REQUEST: case OPUS_SET_PACKET_LOSS_PERC_REQUEST: case OPUS_SET_DTX_REQUEST:
   case OPUS_SET_FORCE_MODE_REQUEST: case OPUS_SET_FORCE_CHANNELS_REQUEST: case OPUS_SET_PREDICTION_DISABLED_REQUEST: case OPUS_SET_PHASE_INVERSION_DISABLED_REQUEST:
   case 4056 /* OPUS_SET_QEXT */: case 11900 /* OPUS_AMD_SET_FLOAT_ANALYSIS (private, include/opus_amd.h) */: {
      const opus_int32 value = va_arg(ap, opus_int32);           /* applied to every stream, stopping at the first that refuses (:1245-1278) */
      for (int s = 0; s < ns; s++) { ret = oa_ms_rec_set(&st->streams[s], st->kind, request, value); if (ret != OPUS_OK) break; }
   } break;
   case 11902 /* OPUS_AMD_SET_KERNEL_PIPELINE (private): the elementary encoders' launches */: case 11906 /* OPUS_AMD_SET_TRANSIENT_PREPASS */: case 11908 /* OPUS_AMD_SET_PVQ_STAGE */: {
      const opus_int32 value = va_arg(ap, opus_int32);
      for (int s = 0; s < ns; s++) { ret = opus_encoder_ctl(&st->streams[s], request, value); if (ret != OPUS_OK) break; }
   } break;
   case OPUS_SET_APPLICATION_REQUEST: {                          /* ... this one through the classic entry point: a change to / from RESTRICTED_LOWDELAY moves a stream to the other record type */
      const opus_int32 value = va_arg(ap, opus_int32);
      for (int s = 0; s < ns; s++) {
         ret = opus_encoder_ctl(&st->streams[s], request, value);
         if (ret != OPUS_OK) break;
         if (s == st->lfe_stream) (void)opus_encoder_ctl(&st->streams[s], OPUS_SET_LFE_REQUEST, 1);      /* (a converted record starts from opus_encoder_init) */
      }
      st->kind = (opus_int32)st->streams[0].kind;
      if (ret == OPUS_OK) st->application = value;
   } break;
   case OPUS_SET_EXPERT_FRAME_DURATION_REQUEST: st->variable_duration = va_arg(ap, opus_int32); break;      /* stored as it comes (:1303-1307) */
   case OPUS_GET_EXPERT_FRAME_DURATION_REQUEST: { opus_int32 *value = va_arg(ap, opus_int32 *); if (!value) ret = OPUS_BAD_ARG; else *value = st->variable_duration; } break;
   default: ret = OPUS_UNIMPLEMENTED;
   }
   return ret;
}
int opus_multistream_encoder_ctl(OpusMSEncoder *st, int request, ...)
{
   va_list ap;
   va_start(ap, request);
   const int ret = oa_ms_encoder_ctl_va(st, request, ap);
   va_end(ap);
   return ret;
}

/* ---------------- multistream decoder ---------------- */
opus_int32 opus_multistream_decoder_get_size(int nb_streams, int nb_coupled_streams)
{
   if (nb_streams < 1 || nb_coupled_streams > nb_streams || nb_coupled_streams < 0) return 0;
   return (opus_int32)(sizeof(OpusMSDecoder) + (size_t)(nb_streams - 1) * sizeof(OpusDecoder));
}
int opus_multistream_decoder_init(OpusMSDecoder *st, opus_int32 Fs, int channels, int streams, int coupled_streams, const unsigned char *mapping)
{
   if (!st || !mapping) return OPUS_BAD_ARG;
   if (channels > 255 || channels < 1 || coupled_streams > streams || streams < 1 || coupled_streams < 0 || streams > 255 - coupled_streams) return OPUS_BAD_ARG;
   OaDecStream *probe = new OaDecStream;
   int r = oa_dec_init_stream(probe, Fs, 2);
   delete probe;
   if (r != OPUS_OK) return r;
   memset(st, 0, sizeof(OpusMSDecoder) - sizeof(OpusDecoder));
   st->magic = OA_MS_MAGIC; st->Fs = Fs;
   st->layout.nb_channels = channels; st->layout.nb_streams = streams; st->layout.nb_coupled_streams = coupled_streams;
   for (int i = 0; i < channels; i++) st->layout.mapping[i] = mapping[i];
   if (!oa_validate_layout(&st->layout)) return OPUS_BAD_ARG;
   for (int s = 0; s < streams; s++) { r = opus_decoder_init(&st->streams[s], Fs, s < coupled_streams ? 2 : 1); if (r != OPUS_OK) return r; }
   return OPUS_OK;
}
OpusMSDecoder *opus_multistream_decoder_create(opus_int32 Fs, int channels, int streams, int coupled_streams, const unsigned char *mapping, int *error)
{
   if (channels > 255 || channels < 1 || coupled_streams > streams || streams < 1 || coupled_streams < 0 || streams > 255 - coupled_streams || !mapping) {
      if (error) *error = OPUS_BAD_ARG;
      return NULL;
   }
   OpusMSDecoder *st = (OpusMSDecoder *)malloc((size_t)opus_multistream_decoder_get_size(streams, coupled_streams));
   if (!st) { if (error) *error = OPUS_ALLOC_FAIL; return NULL; }
   int r = opus_multistream_decoder_init(st, Fs, channels, streams, coupled_streams, mapping);
   if (error) *error = r;
   if (r != OPUS_OK) { free(st); return NULL; }
   return st;
}
void opus_multistream_decoder_destroy(OpusMSDecoder *st) { free(st); }

static int oa_ms_decode_group(OpusDecoder *states, int n, int ch, const unsigned char *pk, int stride, const opus_int32 *lens, opus_int16 *pcm, int frame_size,
      opus_int32 *ns_out, opus_uint32 *rngs, int decode_fec)
{
   int err = OPUS_OK;
   OpusGpuDecBatch *b = oa_ms_dec_batch(n, ch, states[0].Fs, &err);
   if (b) b->decode_fec = decode_fec;
   if (!b) return err == OPUS_OK ? OPUS_INTERNAL_ERROR : err;
   HIPCHECK(hipSetDevice(b->device));
   HIPCHECK(hipStreamSynchronize(b->stream));
   for (int i = 0; i < n; i++) states[i].s.s.transition_gain_Q16 = oa_decode_gain_q16(states[i].decode_gain);
   { const int ru = oa_rows_upload(b->d_streams, &states[0].s, sizeof(OpusDecoder), sizeof(OaDecStream), n); if (ru != OPUS_OK) return ru; }
   int r = opusgpu_decode_batch(b, pk, stride, lens, pcm, frame_size, ns_out, rngs);
   if (r != OPUS_OK) return r;
   return oa_rows_download(&states[0].s, sizeof(OpusDecoder), b->d_streams, sizeof(OaDecStream), n);
}
/* opus_multistream_decode_native, opus_multistream_decoder.c:178 (int16 output) */
int opus_multistream_decode(OpusMSDecoder *st, const unsigned char *data, opus_int32 len, opus_int16 *pcm, int frame_size, int decode_fec)
{
   if (!st || st->magic != OA_MS_MAGIC || !pcm) return OPUS_BAD_ARG;
   if (frame_size <= 0) return OPUS_BAD_ARG;
   const opus_int32 Fs = st->Fs;
   const int ns = st->layout.nb_streams, nc = st->layout.nb_coupled_streams, nm = ns - nc, nch = st->layout.nb_channels;
   if (frame_size > Fs / 25 * 3) frame_size = Fs / 25 * 3;
   if (len < 0) return OPUS_BAD_ARG;
   const bool lost = len == 0 || data == NULL;                                      /* every stream conceals frame_size samples */
   const int fec = lost ? 0 : (decode_fec != 0);                                     /* in-band FEC: each stream decodes the LBRR copy its sub-packet carries (src/opus_multistream_decoder.c:245) */
   if ((lost || fec) && frame_size % (Fs / 400) != 0) return OPUS_BAD_ARG;
   if (!lost && len < 2 * ns - 1) return OPUS_INVALID_PACKET;
   /* opus_multistream_packet_validate (:149) + re-framing of every stream's self-delimited packet as a plain packet for the batch decoder */
   std::vector<opus_int32> lens((size_t)ns), sub((size_t)ns);
   int samples = 0;
   opus_int32 longest = 0;
   if (!lost) {
      const unsigned char *p = data; opus_int32 left = len;
      for (int s = 0; s < ns; s++) {
         unsigned char toc; opus_int16 size[48]; opus_int32 packet_offset;
         if (left <= 0) return OPUS_INVALID_PACKET;
         int count = oa_packet_parse_impl(p, left, s != ns - 1, &toc, NULL, size, NULL, &packet_offset);
         if (count < 0) return count;
         int tmp_samples = opus_packet_get_nb_samples(p, packet_offset, Fs);
         if (s != 0 && samples != tmp_samples) return OPUS_INVALID_PACKET;
         samples = tmp_samples;
         sub[s] = packet_offset; if (packet_offset > longest) longest = packet_offset;
         p += packet_offset; left -= packet_offset;
      }
   }
   if (samples < 0) return samples;
   if (samples > frame_size) return OPUS_BUFFER_TOO_SMALL;                             /* with or without FEC (opus_multistream_decoder.c:216-222) */
   /* one slot per stream, wide enough for the largest sub-packet (a code-3 packet of many maximum-size frames runs to tens of KB; re-framing never grows one) */
   const int stride = (int)((longest > 1275 ? longest : 1275) + 16 + 3) & ~3;
   std::vector<unsigned char> pk((size_t)ns * stride, 0);
   if (lost) { for (int s = 0; s < ns; s++) lens[s] = 0; }
   else {
      const unsigned char *p = data;
      for (int s = 0; s < ns; s++) {
         const opus_int32 packet_offset = sub[s];
         OpusRepacketizer rp;
         opus_repacketizer_init(&rp);
         int r = oa_repacketizer_cat_impl(&rp, p, packet_offset, s != ns - 1);
         if (r != OPUS_OK) return r;
         opus_int32 l = oa_repacketizer_out_range_impl(&rp, 0, rp.nb_frames, pk.data() + (size_t)s * stride, stride, 0, 0);
         if (l < 0) return l;
         lens[s] = l;
         p += packet_offset;
      }
   }
   std::lock_guard<std::mutex> lock(g_ms_mu);
   std::vector<opus_int16> oc((size_t)nc * frame_size * 2 + 2), om((size_t)nm * frame_size + 1);
   std::vector<opus_int32> nso((size_t)ns);
   std::vector<opus_uint32> rngs((size_t)ns);
   int r = OPUS_OK;
   if (nc) r = oa_ms_decode_group(st->streams, nc, 2, pk.data(), stride, lens.data(), oc.data(), frame_size, nso.data(), rngs.data(), fec);
   if (r == OPUS_OK && nm) r = oa_ms_decode_group(st->streams + nc, nm, 1, pk.data() + (size_t)nc * stride, stride, lens.data() + nc, om.data(), frame_size, nso.data() + nc, rngs.data() + nc, fec);
   if (r != OPUS_OK) return r;
   int out_n = 0;
   for (int s = 0; s < ns; s++) { if (nso[s] <= 0) return nso[s] == 0 ? OPUS_INTERNAL_ERROR : nso[s]; out_n = nso[s]; }
   /* OPUS_SET_GAIN reaches every elementary decoder (opus_multistream_decoder.c:427): applied to each stream's output as the classic opus_decode does */
   for (int s = 0; s < ns; s++) if (st->streams[s].decode_gain) {
      if (s < nc) oa_apply_decode_gain(oc.data() + (size_t)s * frame_size * 2, nso[s] * 2, st->streams[s].decode_gain);
      else oa_apply_decode_gain(om.data() + (size_t)(s - nc) * frame_size, nso[s], st->streams[s].decode_gain);
   }
   for (int s = 0; s < ns; s++) {
      if (s < nc) {
         const opus_int16 *b = oc.data() + (size_t)s * frame_size * 2;
         for (int prev = -1, chan; (chan = oa_get_left(&st->layout, s, prev)) != -1; prev = chan) for (int i = 0; i < out_n; i++) pcm[(size_t)i * nch + chan] = b[2 * i];
         for (int prev = -1, chan; (chan = oa_get_right(&st->layout, s, prev)) != -1; prev = chan) for (int i = 0; i < out_n; i++) pcm[(size_t)i * nch + chan] = b[2 * i + 1];
      } else {
         const opus_int16 *b = om.data() + (size_t)(s - nc) * frame_size;
         for (int prev = -1, chan; (chan = oa_get_mono(&st->layout, s, prev)) != -1; prev = chan) for (int i = 0; i < out_n; i++) pcm[(size_t)i * nch + chan] = b[i];
      }
   }
   for (int c = 0; c < nch; c++) if (st->layout.mapping[c] == 255) for (int i = 0; i < out_n; i++) pcm[(size_t)i * nch + c] = 0;
   return out_n;
}
static int oa_ms_decoder_ctl_va(OpusMSDecoder *st, int request, va_list ap)
{
   if (!st || st->magic != OA_MS_MAGIC) return OPUS_BAD_ARG;
   int ret = OPUS_OK;
   const int ns = st->layout.nb_streams;
   switch (request) {
   case OPUS_GET_FINAL_RANGE_REQUEST: {
      opus_uint32 *value = va_arg(ap, opus_uint32 *);
      if (!value) { ret = OPUS_BAD_ARG; break; }
      *value = 0;
      for (int s = 0; s < ns; s++) *value ^= st->streams[s].s.s.rangeFinal;
   } break;
   case 5122 /* OPUS_MULTISTREAM_GET_DECODER_STATE (opus_multistream.h:82) */: {
      const opus_int32 id = va_arg(ap, opus_int32);
      OpusDecoder **value = va_arg(ap, OpusDecoder **);
      if (id < 0 || id >= ns || !value) { ret = OPUS_BAD_ARG; break; }
      *value = &st->streams[id];
   } break;
   case OPUS_SET_GAIN_REQUEST: { const opus_int32 v = va_arg(ap, opus_int32); for (int s = 0; s < ns && ret == OPUS_OK; s++) ret = opus_decoder_ctl(&st->streams[s], request, v); } break;
   case OPUS_GET_GAIN_REQUEST: case OPUS_GET_COMPLEXITY_REQUEST: { opus_int32 *p = va_arg(ap, opus_int32 *); ret = opus_decoder_ctl(&st->streams[0], request, p); } break;   /* (OPUS_GET_PITCH is not in the multistream decoder's switch: opus_multistream_decoder.c:442-547) */
   case OPUS_SET_COMPLEXITY_REQUEST: { const opus_int32 v = va_arg(ap, opus_int32); for (int s = 0; s < ns && ret == OPUS_OK; s++) ret = opus_decoder_ctl(&st->streams[s], request, v); } break;
   case OPUS_GET_SAMPLE_RATE_REQUEST: { opus_int32 *p = va_arg(ap, opus_int32 *); if (!p) ret = OPUS_BAD_ARG; else *p = st->Fs; } break;
   case OPUS_GET_BANDWIDTH_REQUEST: { opus_int32 *p = va_arg(ap, opus_int32 *); if (!p) ret = OPUS_BAD_ARG; else *p = st->streams[0].s.s.bandwidth; } break;
   case 4039 /* OPUS_GET_LAST_PACKET_DURATION */: { opus_int32 *p = va_arg(ap, opus_int32 *); if (!p) ret = OPUS_BAD_ARG; else *p = st->streams[0].s.s.last_packet_duration; } break;
   case OPUS_RESET_STATE:
      for (int s = 0; s < ns; s++) opus_decoder_ctl(&st->streams[s], OPUS_RESET_STATE);
      break;
   case OPUS_SET_PHASE_INVERSION_DISABLED_REQUEST: { opus_int32 v = va_arg(ap, opus_int32); if (v < 0 || v > 1) ret = OPUS_BAD_ARG; else for (int s = 0; s < ns; s++) st->streams[s].s.s.disable_inv = v; } break;
   case OPUS_GET_PHASE_INVERSION_DISABLED_REQUEST: { opus_int32 *p = va_arg(ap, opus_int32 *); if (!p) ret = OPUS_BAD_ARG; else *p = st->streams[0].s.s.disable_inv; } break;
   default: ret = OPUS_UNIMPLEMENTED;
   }
   return ret;
}
int opus_multistream_decoder_ctl(OpusMSDecoder *st, int request, ...)
{
   va_list ap;
   va_start(ap, request);
   const int ret = oa_ms_decoder_ctl_va(st, request, ap);
   va_end(ap);
   return ret;
}
} /* extern "C" */
#endif

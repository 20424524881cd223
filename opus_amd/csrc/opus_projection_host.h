/* opus_projection_host.h — libopus projection API (ambisonics, mapping family 3): a fixed mixing matrix in front of a multistream encoder, the matching
 * demixing matrix behind a multistream decoder.  Same names, arguments and error codes as reference include/opus_projection.h:123-632
 * (src/opus_projection_encoder.c, src/opus_projection_decoder.c); arithmetic of src/mapping_matrix.c:152-286 (fixed point, int16 resolution):
 *   encode:  y[row] = SAT16((sum_col (M[row,col] * x[col]) >> 8) + 64 >> 7)          (int16 input; int24 input: 64-bit sum, (sum + 16384) >> 15, then SAT16(PSHR32(., 8)))
 *   decode:  out[row] += (M[row,col] * x[col] + 16384) >> 15  for every decoded channel `col`, accumulated in the output type (int16 wraps like the reference)
 * The matrices are the codec's constants (projection_tables.h, generated).  Both objects are flat and memcpy-able: header, matrices, then the multistream object. */
#ifndef OPUS_AMD_PROJECTION_HOST_H
#define OPUS_AMD_PROJECTION_HOST_H
#include "projection_tables.h"
#include <type_traits>

#define OPUS_PROJECTION_GET_DEMIXING_MATRIX_GAIN_REQUEST 6001
#define OPUS_PROJECTION_GET_DEMIXING_MATRIX_SIZE_REQUEST 6003
#define OPUS_PROJECTION_GET_DEMIXING_MATRIX_REQUEST 6005
#define OA_PROJ_MAGIC 0x4f41504au

struct OpusProjectionEncoder { opus_uint32 magic; opus_int32 order, channels, pad; /* OpusMSEncoder follows */ };
struct OpusProjectionDecoder { opus_uint32 magic; opus_int32 rows, cols, ms_offset; /* int16 matrix [cols][rows] (column-major), then the OpusMSDecoder */ };
static inline OpusMSEncoder *oa_proj_ms(OpusProjectionEncoder *st) { return (OpusMSEncoder *)(void *)((char *)st + sizeof(OpusProjectionEncoder)); }
static inline opus_int16 *oa_projd_matrix(OpusProjectionDecoder *st) { return (opus_int16 *)(void *)((char *)st + sizeof(OpusProjectionDecoder)); }
static inline OpusMSDecoder *oa_projd_ms(OpusProjectionDecoder *st) { return (OpusMSDecoder *)(void *)((char *)st + st->ms_offset); }

/* channels = (order + 1)^2 [+ 2 non-diegetic]; streams = ceil(channels / 2), all but the odd one coupled (opus_projection_encoder.c:92-134) */
static int oa_proj_layout(int channels, int mapping_family, int *streams, int *coupled, int *order_plus_one)
{
   if (mapping_family != 3 || channels < 1 || channels > 227) return OPUS_BAD_ARG;
   const int o1 = (int)oa_isqrt32((opus_uint32)channels), nd = channels - o1 * o1;
   if (nd != 0 && nd != 2) return OPUS_BAD_ARG;
   if (streams) *streams = (channels + 1) / 2;
   if (coupled) *coupled = channels / 2;
   if (order_plus_one) *order_plus_one = o1;
   return OPUS_OK;
}
static opus_int32 oa_matrix_bytes(int rows, int cols) { if (rows > 255 || cols > 255) return 0; const opus_int32 n = rows * cols * 2; return n > 65004 ? 0 : ((n + 7) & ~7); }

extern "C" {
opus_int32 opus_projection_ambisonics_encoder_get_size(int channels, int mapping_family)
{
   int ns, nc, o1;
   if (oa_proj_layout(channels, mapping_family, &ns, &nc, &o1) != OPUS_OK || o1 < 2 || o1 > 6) return 0;
   const opus_int32 ms = opus_multistream_encoder_get_size(ns, nc);
   return ms ? (opus_int32)sizeof(OpusProjectionEncoder) + ms : 0;
}
int opus_projection_ambisonics_encoder_init(OpusProjectionEncoder *st, opus_int32 Fs, int channels, int mapping_family, int *streams, int *coupled_streams, int application)
{
   int o1;
   if (!st || !streams || !coupled_streams) return OPUS_BAD_ARG;
   if (oa_proj_layout(channels, mapping_family, streams, coupled_streams, &o1) != OPUS_OK) return OPUS_BAD_ARG;
   if (o1 < 2 || o1 > 6) return OPUS_BAD_ARG;
   const OaMatrixDesc *mix = &oa_pm_mixing[o1 - 2], *demix = &oa_pm_demixing[o1 - 2];
   if (*streams + *coupled_streams > mix->rows || channels > mix->cols || channels > demix->rows || *streams + *coupled_streams > demix->cols) return OPUS_BAD_ARG;
   st->magic = OA_PROJ_MAGIC; st->order = o1 - 1; st->channels = channels; st->pad = 0;
   unsigned char mapping[255];
   for (int i = 0; i < channels; i++) mapping[i] = (unsigned char)i;
   return opus_multistream_encoder_init(oa_proj_ms(st), Fs, channels, *streams, *coupled_streams, mapping, application);
}
OpusProjectionEncoder *opus_projection_ambisonics_encoder_create(opus_int32 Fs, int channels, int mapping_family, int *streams, int *coupled_streams, int application, int *error)
{
   const opus_int32 size = opus_projection_ambisonics_encoder_get_size(channels, mapping_family);
   OpusProjectionEncoder *st = size ? (OpusProjectionEncoder *)malloc((size_t)size) : NULL;
   if (!st) { if (error) *error = OPUS_ALLOC_FAIL; return NULL; }
   const int r = opus_projection_ambisonics_encoder_init(st, Fs, channels, mapping_family, streams, coupled_streams, application);
   if (error) *error = r;
   if (r != OPUS_OK) { free(st); return NULL; }
   return st;
}
/* mixes all channels once (row r of the matrix = input channel r of the multistream encoder: its mapping is the identity), then a plain multistream encode */
static int oa_proj_encode(OpusProjectionEncoder *st, const opus_int16 *pcm16, const opus_int32 *pcm24, int frame_size, unsigned char *data, opus_int32 max_data_bytes)
{
   if (!st || st->magic != OA_PROJ_MAGIC || frame_size <= 0 || frame_size > 5760 * 2) return OPUS_BAD_ARG;
   const OaMatrixDesc *m = &oa_pm_mixing[st->order - 1];
   const int C = st->channels;
   std::vector<opus_int16> mixed((size_t)frame_size * C);
   for (int i = 0; i < frame_size; i++) for (int r = 0; r < C; r++) {
      if (pcm16) {
         opus_int32 acc = 0;
         for (int c = 0; c < C; c++) acc += ((opus_int32)m->data[m->rows * c + r] * (opus_int32)pcm16[(size_t)i * C + c]) >> 8;
         mixed[(size_t)i * C + r] = oa_sat16((acc + 64) >> 7);
      } else {
         long long acc = 0;
         for (int c = 0; c < C; c++) acc += (long long)m->data[m->rows * c + r] * pcm24[(size_t)i * C + c];
         const long long v24 = (acc + 16384) >> 15;
         long long v = (v24 + 128) >> 8;
         mixed[(size_t)i * C + r] = (opus_int16)(v > 32767 ? 32767 : v < -32768 ? -32768 : v);
      }
   }
   /* the elementary encoders' analyses look at the caller's UN-mixed channels (opus_multistream_encode_native hands opus_encode_native the original pcm with the
    * stream's channel indices, opus_multistream_encoder.c:1027) through the entry point's downmix function.  opus_projection_encode24 passes downmix_int -- the int16
    * reader -- and MAX_ENCODING_DEPTH for its int32 input (src/opus_projection_encoder.c:408-415): the analysis of the reference therefore reads the caller's buffer as
    * int16 halves, sample k of the view = half k of the int32 array.  A drop-in shares that: the same halves, the same depth (16 in this FIXED_POINT build). */
   std::vector<opus_int32> sig((size_t)frame_size * C);
   const opus_int16 *v16 = pcm16 ? pcm16 : (const opus_int16 *)(const void *)pcm24;
   for (size_t i = 0; i < sig.size(); i++) sig[i] = (opus_int32)((opus_uint32)(opus_int32)v16[i] << 12);
   return oa_ms_encode_native(oa_proj_ms(st), mixed.data(), frame_size, data, max_data_bytes, pcm16 ? 16 : OA_MAX_ENCODING_DEPTH, sig.data());
}
int opus_projection_encode(OpusProjectionEncoder *st, const opus_int16 *pcm, int frame_size, unsigned char *data, opus_int32 max_data_bytes)
{ return pcm ? oa_proj_encode(st, pcm, NULL, frame_size, data, max_data_bytes) : OPUS_BAD_ARG; }
int opus_projection_encode24(OpusProjectionEncoder *st, const opus_int32 *pcm, int frame_size, unsigned char *data, opus_int32 max_data_bytes)
{ return pcm ? oa_proj_encode(st, NULL, pcm, frame_size, data, max_data_bytes) : OPUS_BAD_ARG; }
/* float input in the fixed-point build: FLOAT2RES((1/32768) * sum M * x) (mapping_matrix.c:72-96) */
int opus_projection_encode_float(OpusProjectionEncoder *st, const float *pcm, int frame_size, unsigned char *data, opus_int32 max_data_bytes)
{
   if (!st || st->magic != OA_PROJ_MAGIC || !pcm || frame_size <= 0 || frame_size > 5760 * 2) return OPUS_BAD_ARG;
   const OaMatrixDesc *m = &oa_pm_mixing[st->order - 1];
   const int C = st->channels;
   std::vector<opus_int16> mixed((size_t)frame_size * C);
   for (int i = 0; i < frame_size; i++) for (int r = 0; r < C; r++) {
      float acc = 0;
      for (int c = 0; c < C; c++) acc += m->data[m->rows * c + r] * pcm[(size_t)i * C + c];
      mixed[(size_t)i * C + r] = oa_float2int16((1 / 32768.f) * acc);
   }
   std::vector<opus_int32> sig((size_t)frame_size * C);                                    /* (the analyses see the un-mixed input: downmix_float) */
   for (size_t i = 0; i < sig.size(); i++) sig[i] = oa_float2sig(pcm[i]);
   return oa_ms_encode_native(oa_proj_ms(st), mixed.data(), frame_size, data, max_data_bytes, OA_MAX_ENCODING_DEPTH, sig.data());
}
void opus_projection_encoder_destroy(OpusProjectionEncoder *st) { free(st); }
int opus_projection_encoder_ctl(OpusProjectionEncoder *st, int request, ...)
{
   if (!st || st->magic != OA_PROJ_MAGIC) return OPUS_BAD_ARG;
   OpusMSEncoder *ms = oa_proj_ms(st);
   const OaMatrixDesc *demix = &oa_pm_demixing[st->order - 1];
   const int nin = ms->layout.nb_streams + ms->layout.nb_coupled_streams, nout = ms->layout.nb_channels;
   va_list ap;
   va_start(ap, request);
   int ret = OPUS_OK;
   switch (request) {
   case OPUS_PROJECTION_GET_DEMIXING_MATRIX_SIZE_REQUEST: { opus_int32 *v = va_arg(ap, opus_int32 *); if (!v) ret = OPUS_BAD_ARG; else *v = nout * nin * 2; } break;
   case OPUS_PROJECTION_GET_DEMIXING_MATRIX_GAIN_REQUEST: { opus_int32 *v = va_arg(ap, opus_int32 *); if (!v) ret = OPUS_BAD_ARG; else *v = demix->gain; } break;
   case OPUS_PROJECTION_GET_DEMIXING_MATRIX_REQUEST: {                                   /* the sub-matrix the decoder needs, little-endian int16, column-major */
      unsigned char *dst = va_arg(ap, unsigned char *);
      const opus_int32 size = va_arg(ap, opus_int32);
      if (!dst || size != nin * nout * 2) { ret = OPUS_BAD_ARG; break; }
      int l = 0;
      for (int i = 0; i < nin; i++) for (int j = 0; j < nout; j++, l++) { const int v = demix->data[demix->rows * i + j]; dst[2 * l] = (unsigned char)v; dst[2 * l + 1] = (unsigned char)(v >> 8); }
   } break;
   default: ret = oa_ms_encoder_ctl_va(ms, request, ap);
   }
   va_end(ap);
   return ret;
}

opus_int32 opus_projection_decoder_get_size(int channels, int streams, int coupled_streams)
{
   const opus_int32 mb = oa_matrix_bytes(streams + coupled_streams, channels), ms = opus_multistream_decoder_get_size(streams, coupled_streams);
   return mb && ms ? (opus_int32)sizeof(OpusProjectionDecoder) + mb + ms : 0;
}
int opus_projection_decoder_init(OpusProjectionDecoder *st, opus_int32 Fs, int channels, int streams, int coupled_streams, unsigned char *demixing_matrix, opus_int32 demixing_matrix_size)
{
   if (!st || !demixing_matrix) return OPUS_BAD_ARG;
   const int nin = streams + coupled_streams;
   if (nin * channels * 2 != demixing_matrix_size) return OPUS_BAD_ARG;
   const opus_int32 mb = oa_matrix_bytes(channels, nin);
   if (!mb) return OPUS_BAD_ARG;
   st->magic = OA_PROJ_MAGIC; st->rows = channels; st->cols = nin; st->ms_offset = (opus_int32)sizeof(OpusProjectionDecoder) + mb;
   opus_int16 *M = oa_projd_matrix(st);
   for (int i = 0; i < nin * channels; i++) M[i] = (opus_int16)(demixing_matrix[2 * i + 1] << 8 | demixing_matrix[2 * i]);
   unsigned char mapping[255];
   for (int i = 0; i < channels; i++) mapping[i] = (unsigned char)i;
   return opus_multistream_decoder_init(oa_projd_ms(st), Fs, channels, streams, coupled_streams, mapping);
}
OpusProjectionDecoder *opus_projection_decoder_create(opus_int32 Fs, int channels, int streams, int coupled_streams, unsigned char *demixing_matrix, opus_int32 demixing_matrix_size, int *error)
{
   const opus_int32 size = opus_projection_decoder_get_size(channels, streams, coupled_streams);
   OpusProjectionDecoder *st = size ? (OpusProjectionDecoder *)malloc((size_t)size) : NULL;
   if (!st) { if (error) *error = OPUS_ALLOC_FAIL; return NULL; }
   const int r = opus_projection_decoder_init(st, Fs, channels, streams, coupled_streams, demixing_matrix, demixing_matrix_size);
   if (error) *error = r;
   if (r != OPUS_OK) { free(st); return NULL; }
   return st;
}
} /* extern "C" */
/* decode with the identity layout (decoded channel c = stream channel c), then demix: T = int16 (wraps like the reference's `output[] +=`), int32 (<< 8 domain) or float */
template <class T> static int oa_proj_decode(OpusProjectionDecoder *st, const unsigned char *data, opus_int32 len, T *pcm, int frame_size, int decode_fec)
{
   if (!st || st->magic != OA_PROJ_MAGIC || !pcm || frame_size <= 0) return OPUS_BAD_ARG;
   OpusMSDecoder *ms = oa_projd_ms(st);
   const int C = st->rows;
   if (frame_size > ms->Fs / 25 * 3) frame_size = ms->Fs / 25 * 3;
   std::vector<opus_int16> dec((size_t)frame_size * C);
   const int n = opus_multistream_decode(ms, data, len, dec.data(), frame_size, decode_fec);
   if (n <= 0) return n;
   const opus_int16 *M = oa_projd_matrix(st);
   for (int i = 0; i < n; i++) for (int r = 0; r < C; r++) {
      if (sizeof(T) == 2) { opus_int16 acc = 0; for (int c = 0; c < C && c < st->cols; c++) acc = (opus_int16)(acc + (((opus_int32)M[st->rows * c + r] * dec[(size_t)i * C + c] + 16384) >> 15)); pcm[(size_t)i * C + r] = (T)acc; }
      else if (std::is_integral<T>::value) { opus_int32 acc = 0; for (int c = 0; c < C && c < st->cols; c++) acc += (opus_int32)(((long long)M[st->rows * c + r] * ((opus_int32)dec[(size_t)i * C + c] * 256) + 16384) >> 15); pcm[(size_t)i * C + r] = (T)acc; }
      else { float acc = 0; for (int c = 0; c < C && c < st->cols; c++) acc += (1 / 32768.f) * M[st->rows * c + r] * ((1.f / 32768.f) * dec[(size_t)i * C + c]); pcm[(size_t)i * C + r] = (T)acc; }
   }
   return n;
}
extern "C" {
int opus_projection_decode(OpusProjectionDecoder *st, const unsigned char *data, opus_int32 len, opus_int16 *pcm, int frame_size, int decode_fec) { return oa_proj_decode(st, data, len, pcm, frame_size, decode_fec); }
int opus_projection_decode24(OpusProjectionDecoder *st, const unsigned char *data, opus_int32 len, opus_int32 *pcm, int frame_size, int decode_fec) { return oa_proj_decode(st, data, len, pcm, frame_size, decode_fec); }
int opus_projection_decode_float(OpusProjectionDecoder *st, const unsigned char *data, opus_int32 len, float *pcm, int frame_size, int decode_fec) { return oa_proj_decode(st, data, len, pcm, frame_size, decode_fec); }
int opus_projection_decoder_ctl(OpusProjectionDecoder *st, int request, ...)
{
   if (!st || st->magic != OA_PROJ_MAGIC) return OPUS_BAD_ARG;
   va_list ap;
   va_start(ap, request);
   const int ret = oa_ms_decoder_ctl_va(oa_projd_ms(st), request, ap);
   va_end(ap);
   return ret;
}
void opus_projection_decoder_destroy(OpusProjectionDecoder *st) { free(st); }
} /* extern "C" */
#endif

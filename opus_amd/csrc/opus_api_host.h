/* opus_api_host.h — the remaining exported entry points of include/opus.h and opus_multistream.h that are pure host code:
 *   opus_pcm_soft_clip                      include/opus.h:800   (src/opus.c:39 opus_pcm_soft_clip_impl)
 *   opus_multistream_encode24/_float, opus_multistream_decode24/_float   include/opus_multistream.h:411-726
 *   the DRED entry points                   include/opus.h:594-709: this library is built without DRED, and answers exactly like a reference built without
 *                                           ENABLE_DRED (src/opus_decoder.c:1341-1692): objects can be created, every operation returns OPUS_UNIMPLEMENTED
 * The int16-resolution fixed-point build is the arithmetic of this library, so the 24-bit / float variants convert at the boundary, as the reference's
 * FIXED_POINT build without ENABLE_RES24 does (celt/arch.h:167-173). */
#ifndef OPUS_AMD_API_HOST_H
#define OPUS_AMD_API_HOST_H
#include <math.h>

extern "C" {
/* Soft clipping to [-1, 1] (src/opus.c:39): samples beyond +-1 are pulled in with x + a x^2 where `a` is chosen per region between zero crossings so that the peak lands on
 * +-1; a region that straddles the block boundary continues with the `a` remembered in softclip_mem. */
void opus_pcm_soft_clip(float *pcm, int frame_size, int channels, float *softclip_mem)
{
   if (channels < 1 || frame_size < 1 || !pcm || !softclip_mem) return;
   const int N = frame_size, C = channels;
   /* hard bound first: beyond +-2 the quadratic would fold over */
   bool any = false;
   for (int i = 0; i < N * C; i++) { float v = pcm[i]; if (v > 2.f) v = 2.f; if (v < -2.f) v = -2.f; pcm[i] = v; any |= v > 1.f || v < -1.f; }
   for (int c = 0; c < C; c++) {
      float *x = pcm + c;
      float a = softclip_mem[c];
      /* finish the region the previous block left open (until the signal crosses zero) */
      int i;
      for (i = 0; i < N; i++) { if (x[i * C] * a >= 0) break; x[i * C] = x[i * C] + a * x[i * C] * x[i * C]; }
      int curr = 0;
      const float x0 = x[0];
      for (;;) {
         /* next sample outside [-1, 1] */
         int pos = curr;
         if (any) while (pos < N && !(x[pos * C] > 1.f || x[pos * C] < -1.f)) pos++; else pos = N;      /* (the reference's predicate, src/opus.c:81: a NaN is not "outside" and passes through; the other form never got past one) */
         if (pos == N) { a = 0; break; }
         int peak = pos, start = pos, end = pos;
         float vmax = fabsf(x[pos * C]);
         const float sgn = x[pos * C];
         while (start > 0 && sgn * x[(start - 1) * C] >= 0) start--;                     /* back to the previous zero crossing */
         while (end < N && sgn * x[end * C] >= 0) { if (fabsf(x[end * C]) > vmax) { vmax = fabsf(x[end * C]); peak = end; } end++; }   /* forward to the next, tracking the peak */
         const bool special = start == 0 && sgn * x[0] >= 0;                                /* the region began before this block */
         a = (vmax - 1.f) / (vmax * vmax);                                                /* peak + a peak^2 = 1 */
         a += a * 2.4e-7f;                                                                /* guard against rounding */
         if (sgn > 0) a = -a;
         for (int k = start; k < end; k++) x[k * C] = x[k * C] + a * x[k * C] * x[k * C];
         if (special && peak >= 2) {
            /* the block starts inside the region: ramp from the unmodified first sample so that there is no step at the boundary */
            float offset = x0 - x[0];
            const float delta = offset / peak;
            for (int k = curr; k < peak; k++) { offset -= delta; float v = x[k * C] + offset; if (v > 1.f) v = 1.f; if (v < -1.f) v = -1.f; x[k * C] = v; }
         }
         curr = end;
         if (curr == N) break;
      }
      softclip_mem[c] = a;
   }
}

/* ---- DRED: not built (same answers as the reference without ENABLE_DRED) ---- */
struct OpusDREDDecoder { int loaded; int arch; opus_uint32 magic; };
struct OpusDRED { int process_stage; };
int opus_dred_decoder_get_size(void) { return (int)sizeof(OpusDREDDecoder); }
int opus_dred_decoder_init(OpusDREDDecoder *dec) { dec->loaded = 0; dec->arch = 0; dec->magic = 0xD8EDDEC0u; return OPUS_OK; }
OpusDREDDecoder *opus_dred_decoder_create(int *error)
{
   OpusDREDDecoder *dec = (OpusDREDDecoder *)malloc(sizeof(OpusDREDDecoder));
   if (!dec) { if (error) *error = OPUS_ALLOC_FAIL; return NULL; }
   const int r = opus_dred_decoder_init(dec);
   if (error) *error = r;
   return dec;
}
void opus_dred_decoder_destroy(OpusDREDDecoder *dec) { if (dec) dec->magic = 0xDE57801Du; free(dec); }
int opus_dred_decoder_ctl(OpusDREDDecoder *, int, ...) { return OPUS_UNIMPLEMENTED; }
int opus_dred_get_size(void) { return 0; }
OpusDRED *opus_dred_alloc(int *error) { if (error) *error = OPUS_UNIMPLEMENTED; return NULL; }
void opus_dred_free(OpusDRED *dec) { free(dec); }
int opus_dred_parse(OpusDREDDecoder *, OpusDRED *, const unsigned char *, opus_int32, opus_int32, opus_int32, int *, int) { return OPUS_UNIMPLEMENTED; }
int opus_dred_process(OpusDREDDecoder *, const OpusDRED *, OpusDRED *) { return OPUS_UNIMPLEMENTED; }
int opus_decoder_dred_decode(OpusDecoder *, const OpusDRED *, opus_int32, opus_int16 *, opus_int32) { return OPUS_UNIMPLEMENTED; }
int opus_decoder_dred_decode24(OpusDecoder *, const OpusDRED *, opus_int32, opus_int32 *, opus_int32) { return OPUS_UNIMPLEMENTED; }
int opus_decoder_dred_decode_float(OpusDecoder *, const OpusDRED *, opus_int32, float *, opus_int32) { return OPUS_UNIMPLEMENTED; }

/* ---- multistream: 24-bit and float entry points convert at the boundary ---- */
static int oa_ms_enc_channels(const OpusMSEncoder *st) { return st->layout.nb_channels; }
int opus_multistream_encode24(OpusMSEncoder *st, const opus_int32 *pcm, int frame_size, unsigned char *data, opus_int32 max_data_bytes)
{
   if (!st || st->magic != OA_MS_MAGIC || !pcm || frame_size <= 0 || frame_size > 5760 * 2) return OPUS_BAD_ARG;
   std::vector<opus_int16> in((size_t)frame_size * oa_ms_enc_channels(st));
   std::vector<opus_int32> sig(in.size());
   for (size_t i = 0; i < in.size(); i++) { in[i] = oa_sat16((pcm[i] + 128) >> 8); sig[i] = (opus_int32)((opus_uint32)pcm[i] << 4); }
   return oa_ms_encode_native(st, in.data(), frame_size, data, max_data_bytes, OA_MAX_ENCODING_DEPTH, sig.data());
}
int opus_multistream_encode_float(OpusMSEncoder *st, const float *pcm, int frame_size, unsigned char *data, opus_int32 max_data_bytes)
{
   if (!st || st->magic != OA_MS_MAGIC || !pcm || frame_size <= 0 || frame_size > 5760 * 2) return OPUS_BAD_ARG;
   std::vector<opus_int16> in((size_t)frame_size * oa_ms_enc_channels(st));
   std::vector<opus_int32> sig(in.size());
   for (size_t i = 0; i < in.size(); i++) { in[i] = oa_float2int16(pcm[i]); sig[i] = oa_float2sig(pcm[i]); }
   return oa_ms_encode_native(st, in.data(), frame_size, data, max_data_bytes, OA_MAX_ENCODING_DEPTH, sig.data());
}
int opus_multistream_decode24(OpusMSDecoder *st, const unsigned char *data, opus_int32 len, opus_int32 *pcm, int frame_size, int decode_fec)
{
   if (!st || st->magic != OA_MS_MAGIC || !pcm || frame_size <= 0) return OPUS_BAD_ARG;
   if (frame_size > st->Fs / 25 * 3) frame_size = st->Fs / 25 * 3;
   std::vector<opus_int16> out((size_t)frame_size * st->layout.nb_channels);
   const int n = opus_multistream_decode(st, data, len, out.data(), frame_size, decode_fec);
   for (int i = 0; i < n * st->layout.nb_channels; i++) pcm[i] = (opus_int32)out[i] * 256;
   return n;
}
int opus_multistream_decode_float(OpusMSDecoder *st, const unsigned char *data, opus_int32 len, float *pcm, int frame_size, int decode_fec)
{
   if (!st || st->magic != OA_MS_MAGIC || !pcm || frame_size <= 0) return OPUS_BAD_ARG;
   if (frame_size > st->Fs / 25 * 3) frame_size = st->Fs / 25 * 3;
   std::vector<opus_int16> out((size_t)frame_size * st->layout.nb_channels);
   const int n = opus_multistream_decode(st, data, len, out.data(), frame_size, decode_fec);
   for (int i = 0; i < n * st->layout.nb_channels; i++) pcm[i] = (1.f / 32768.f) * out[i];
   return n;
}
} /* extern "C" */
#endif

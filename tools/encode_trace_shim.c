/* tools/encode_trace_shim.c — TEST INFRASTRUCTURE: an LD_PRELOAD shim that logs every opus_encode / opus_encode_float / opus_multistream_encode /
 * opus_projection_encode call (and every opus_decode / opus_multistream_decode call) an unmodified program makes (frame size, byte budget, return value, FNV-1a hash of the packet) to $OPUS_TRACE_FILE and forwards to
 * whichever library the program is linked to.  The reference's tests/test_opus_encode.c is deterministic for a given seed, so two runs of it -- one linked to the
 * compiled reference, one to this library -- must leave identical logs: every packet of the mode matrix, the settings fuzz and the regression cases, byte for byte
 * (tools/encode_trace_compare.py).
 *    gcc -O2 -shared -fPIC tools/encode_trace_shim.c -o /tmp/enctrace.so -ldl */
#define _GNU_SOURCE
#include <dlfcn.h>
#include <stdio.h>
#include <stdlib.h>
#include <stdint.h>
static FILE *logf_; static long ncall;
static void init(void) { if (!logf_) { const char *p = getenv("OPUS_TRACE_FILE"); logf_ = fopen(p ? p : "/tmp/opus_enc_trace.log", "w"); setvbuf(logf_, NULL, _IOFBF, 1 << 20); } }
static uint32_t fnv(const void *p, long n) { const unsigned char *b = (const unsigned char *)p; uint32_t h = 2166136261u; for (long i = 0; i < n; i++) h = (h ^ b[i]) * 16777619u; return h; }
static FILE *pcmf_; static long pcm_from_ = -1, pcm_to_ = (1L << 60); static int pcm_ch_[64]; static void *pcm_st_[64];
static void pcm_dump(void *st, const void *pcm, int frame_size, int max_bytes)
{
   if (pcm_from_ < 0) { const char *f = getenv("OPUS_TRACE_PCM_FROM"); pcm_from_ = f ? atol(f) : (1L << 60); if (f) pcmf_ = fopen(getenv("OPUS_TRACE_PCM"), "wb"); if (getenv("OPUS_TRACE_PCM_TO")) pcm_to_ = atol(getenv("OPUS_TRACE_PCM_TO")); }
   if (ncall < pcm_from_ || ncall > pcm_to_ || !pcmf_) return;
   int ch = 0; for (int i = 0; i < 64; i++) if (pcm_st_[i] == st) ch = pcm_ch_[i];
   if (!ch) { int32_t v = 0; int (*ctl)(void *, int, ...) = dlsym(RTLD_NEXT, "opus_encoder_ctl"); /* channels: not a public ctl -> infer from the state's creation if seen */ (void)ctl; (void)v; return; }
   int32_t h[4] = {(int32_t)ncall, frame_size, ch, max_bytes}; fwrite(h, 4, 4, pcmf_); fwrite(pcm, 2, (size_t)frame_size * ch, pcmf_); fflush(pcmf_);
}
static void line(const char *what, int fs, int max, int r, const unsigned char *data)
{
   fprintf(logf_, "%ld %s fs=%d max=%d ret=%d toc=%02x pkt=%08x\n", ncall++, what, fs, max, r, r > 0 ? data[0] : 0, r > 0 ? fnv(data, r) : 0);
}
static void pcm_dump(void *st, const void *pcm, int frame_size, int max_bytes);
#define WRAP(name, tag, pcm_t) \
int name(void *st, const pcm_t *pcm, int frame_size, unsigned char *data, int32_t max_bytes) \
{ \
   static int (*real)(void *, const pcm_t *, int, unsigned char *, int32_t); \
   if (!real) real = dlsym(RTLD_NEXT, #name); \
   init(); \
   if (sizeof(pcm_t) == 2 && tag[0] == 'E') pcm_dump(st, pcm, frame_size, (int)max_bytes); \
   int r = real(st, pcm, frame_size, data, max_bytes); \
   line(tag, frame_size, (int)max_bytes, r, data); \
   return r; \
}
WRAP(opus_encode, "E", int16_t)
WRAP(opus_encode_float, "Ef", float)
WRAP(opus_encode24, "E24", int32_t)
WRAP(opus_multistream_encode, "M", int16_t)
WRAP(opus_multistream_encode_float, "Mf", float)
WRAP(opus_multistream_encode24, "M24", int32_t)
WRAP(opus_projection_encode, "P", int16_t)
WRAP(opus_projection_encode_float, "Pf", float)
/* with $OPUS_TRACE_DECODE set, the decode calls of the same run as well (the program decodes every packet it made, some of them after corrupting them): sample count and PCM hash.  The decoder's channel count
 * is known for decoders the program created through the API; for the copies it makes with memcpy only the sample count is logged */
static void *dec_st_[64]; static int dec_ch_[64], dec_n_;
static int dec_channels(void *st) { for (int i = 0; i < 64; i++) if (dec_st_[i] == st) return dec_ch_[i]; return 0; }
void *opus_decoder_create(int32_t Fs, int channels, int *error)
{
   static void *(*real)(int32_t, int, int *);
   if (!real) real = dlsym(RTLD_NEXT, "opus_decoder_create");
   void *st = real(Fs, channels, error);
   dec_st_[dec_n_ % 64] = st; dec_ch_[dec_n_ % 64] = channels; dec_n_++;
   return st;
}
void *opus_multistream_decoder_create(int32_t Fs, int channels, int streams, int coupled, const unsigned char *mapping, int *error)
{
   static void *(*real)(int32_t, int, int, int, const unsigned char *, int *);
   if (!real) real = dlsym(RTLD_NEXT, "opus_multistream_decoder_create");
   void *st = real(Fs, channels, streams, coupled, mapping, error);
   dec_st_[dec_n_ % 64] = st; dec_ch_[dec_n_ % 64] = channels; dec_n_++;
   return st;
}
#define WRAPD(name, tag) \
int name(void *st, const unsigned char *data, int32_t len, int16_t *pcm, int frame_size, int fec) \
{ \
   static int (*real)(void *, const unsigned char *, int32_t, int16_t *, int, int); \
   if (!real) real = dlsym(RTLD_NEXT, #name); \
   init(); \
   int r = real(st, data, len, pcm, frame_size, fec); \
   static int on = -1; if (on < 0) on = getenv("OPUS_TRACE_DECODE") != NULL; \
   const int ch = dec_channels(st); \
   if (on) fprintf(logf_, "%ld %s len=%d fs=%d fec=%d ret=%d pcm=%08x\n", ncall++, tag, (int)len, frame_size, fec, r, r > 0 && ch ? fnv(pcm, (long)r * ch * 2) : 0); \
   return r; \
}
WRAPD(opus_decode, "D")
WRAPD(opus_multistream_decode, "MD")
/* context lines (not numbered: the call index counts encode calls only): encoder creation, controls, destruction.  With $OPUS_TRACE_PCM_FROM=<n> [$OPUS_TRACE_PCM_TO=<m>] the int16 input of
 * every opus_encode call from encode call n on (to m) is appended to $OPUS_TRACE_PCM (for replaying a section elsewhere: tools/encode_trace_replay.py) */
#include <stdarg.h>
void *opus_encoder_create(int32_t Fs, int channels, int application, int *error)
{
   static void *(*real)(int32_t, int, int, int *);
   if (!real) real = dlsym(RTLD_NEXT, "opus_encoder_create");
   init();
   void *st = real(Fs, channels, application, error);
   fprintf(logf_, "# create %p Fs=%d ch=%d app=%d\n", st, (int)Fs, channels, application);
   { static int k; pcm_st_[k % 64] = st; pcm_ch_[k % 64] = channels; k++; }
   return st;
}
void opus_encoder_destroy(void *st)
{
   static void (*real)(void *);
   if (!real) real = dlsym(RTLD_NEXT, "opus_encoder_destroy");
   init();
   fprintf(logf_, "# destroy %p\n", st);
   real(st);
}
int opus_encoder_ctl(void *st, int request, ...)
{
   static int (*real)(void *, int, ...);
   if (!real) real = dlsym(RTLD_NEXT, "opus_encoder_ctl");
   init();
   va_list ap; va_start(ap, request); void *arg = va_arg(ap, void *); va_end(ap);        /* one int or one pointer: forwarded as it came */
   int r = real(st, request, arg);
   if (!(request & 1) && request != 4028) fprintf(logf_, "# ctl %p %d %d -> %d\n", st, request, (int)(intptr_t)arg, r);
   return r;
}
__attribute__((destructor)) static void fin(void) { if (logf_) fclose(logf_); }

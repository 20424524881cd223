/* tools/encode_trace_shim.c — DIAGNOSTIC (not product, not test): an LD_PRELOAD shim that logs every opus_encode / opus_encode_float / opus_multistream_encode /
 * opus_projection_encode call an unmodified program makes (frame size, byte budget, return value, FNV-1a hash of the packet) to $OPUS_TRACE_FILE and forwards to
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
static void line(const char *what, int fs, int max, int r, const unsigned char *data)
{
   fprintf(logf_, "%ld %s fs=%d max=%d ret=%d toc=%02x pkt=%08x\n", ncall++, what, fs, max, r, r > 0 ? data[0] : 0, r > 0 ? fnv(data, r) : 0);
}
#define WRAP(name, tag, pcm_t) \
int name(void *st, const pcm_t *pcm, int frame_size, unsigned char *data, int32_t max_bytes) \
{ \
   static int (*real)(void *, const pcm_t *, int, unsigned char *, int32_t); \
   if (!real) real = dlsym(RTLD_NEXT, #name); \
   init(); \
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
__attribute__((destructor)) static void fin(void) { if (logf_) fclose(logf_); }

/* tools/api_trace_shim.c — DIAGNOSTIC (not product, not test): an LD_PRELOAD shim that logs every opus_encode24 / opus_decode24 call an unmodified program makes
 * (arguments, return value, FNV-1a hash of the packet / of the PCM produced) to $OPUS_TRACE_FILE and forwards to whichever library the program is linked to.
 * With $OPUS_TRACE_DUMP every packet produced and every PCM block decoded is also written as (kind, length, bytes) records.
 * Two runs of the same program against two libraries (this one on the GPU, the reference's) are then compared call by call:
 *    gcc -O2 -shared -fPIC tools/api_trace_shim.c -o /tmp/trace.so -ldl
 *    OPUS_TRACE_FILE=a.log OPUS_TRACE_CH=2 LD_PRELOAD=/tmp/trace.so oracle/_ref/reftests/gpu/opus_demo ...   */
#define _GNU_SOURCE
#include <dlfcn.h>
#include <stdio.h>
#include <stdlib.h>
#include <stdint.h>
static FILE *logf_, *dump_; static int ch_ = 2; static long ncall;
static void rec(int kind, const void *p, long n) { if (!dump_) return; int32_t h[2] = {kind, (int32_t)n}; fwrite(h, 4, 2, dump_); if (n > 0) fwrite(p, 1, n, dump_); fflush(dump_); }
static void init(void) { if (!logf_) { const char *p = getenv("OPUS_TRACE_FILE"); logf_ = fopen(p ? p : "/tmp/opus_trace.log", "w"); if (getenv("OPUS_TRACE_CH")) ch_ = atoi(getenv("OPUS_TRACE_CH")); if (getenv("OPUS_TRACE_DUMP")) dump_ = fopen(getenv("OPUS_TRACE_DUMP"), "wb"); } }
static uint32_t fnv(const void *p, long n) { const unsigned char *b = (const unsigned char *)p; uint32_t h = 2166136261u; for (long i = 0; i < n; i++) h = (h ^ b[i]) * 16777619u; return h; }
int opus_encode24(void *st, const int32_t *pcm, int frame_size, unsigned char *data, int32_t max_bytes)
{
   static int (*real)(void *, const int32_t *, int, unsigned char *, int32_t);
   if (!real) real = dlsym(RTLD_NEXT, "opus_encode24");
   init();
   int r = real(st, pcm, frame_size, data, max_bytes);
   fprintf(logf_, "%ld E fs=%d max=%d ret=%d toc=%02x in=%08x pkt=%08x\n", ncall++, frame_size, (int)max_bytes, r, r > 0 ? data[0] : 0, fnv(pcm, (long)frame_size * ch_ * 4), r > 0 ? fnv(data, r) : 0); fflush(logf_);
   rec('E', data, r);
   return r;
}
int opus_decode24(void *st, const unsigned char *data, int32_t len, int32_t *pcm, int frame_size, int decode_fec)
{
   static int (*real)(void *, const unsigned char *, int32_t, int32_t *, int, int);
   if (!real) real = dlsym(RTLD_NEXT, "opus_decode24");
   init();
   int r = real(st, data, len, pcm, frame_size, decode_fec);
   fprintf(logf_, "%ld D len=%d toc=%02x fec=%d fs=%d ret=%d pkt=%08x pcm=%08x\n", ncall++, (int)len, data && len > 0 ? data[0] : 0, decode_fec, frame_size, r, data && len > 0 ? fnv(data, len) : 0, r > 0 ? fnv(pcm, (long)r * ch_ * 4) : 0); fflush(logf_);
   rec(decode_fec ? 'F' : 'D', pcm, r > 0 ? (long)r * ch_ * 4 : 0);
   return r;
}

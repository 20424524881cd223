/* tools/decode_trace_shim.c — DIAGNOSTIC (not product, not test).  LD_PRELOAD shim: record every opus_decoder_create and opus_decode call (decoder pointer, arguments, packet bytes, return value) to $OPUS_TRACE_DUMP */
#define _GNU_SOURCE
#include <dlfcn.h>
#include <stdio.h>
#include <stdlib.h>
#include <stdint.h>
#include <string.h>
static FILE *dump_;
static void init(void) { if (!dump_) dump_ = fopen(getenv("OPUS_TRACE_DUMP") ? getenv("OPUS_TRACE_DUMP") : "/tmp/dec_trace.bin", "wb"); }
static void rec(int kind, uint64_t st, int a, int b, int c, int d, const void *p, int n)
{
   init(); int32_t h[6] = {kind, a, b, c, d, n}; fwrite(&st, 8, 1, dump_); fwrite(h, 4, 6, dump_); if (n > 0) fwrite(p, 1, n, dump_); fflush(dump_);
}
void *opus_decoder_create(int32_t Fs, int channels, int *error)
{
   static void *(*real)(int32_t, int, int *); if (!real) real = dlsym(RTLD_NEXT, "opus_decoder_create");
   void *st = real(Fs, channels, error); rec('C', (uint64_t)st, Fs, channels, 0, 0, 0, 0); return st;
}
int opus_decode(void *st, const unsigned char *data, int32_t len, int16_t *pcm, int frame_size, int decode_fec)
{
   static int (*real)(void *, const unsigned char *, int32_t, int16_t *, int, int); if (!real) real = dlsym(RTLD_NEXT, "opus_decode");
   int r = real(st, data, len, pcm, frame_size, decode_fec);
   rec('D', (uint64_t)st, len, frame_size, decode_fec, r, data, data && len > 0 ? len : 0); return r;
}
#include <stdarg.h>
int opus_decoder_ctl(void *st, int request, ...)
{
   static int (*real)(void *, int, ...); if (!real) real = dlsym(RTLD_NEXT, "opus_decoder_ctl");
   va_list ap; va_start(ap, request); void *arg = va_arg(ap, void *); va_end(ap);
   if (request == 4028) rec('R', (uint64_t)st, 0, 0, 0, 0, 0, 0);
   return real(st, request, arg);
}

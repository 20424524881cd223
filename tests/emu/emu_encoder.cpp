/* tests/emu/emu_encoder.cpp — TEST INFRASTRUCTURE: runs the *kernel body source* (opus_amd/csrc/celt_enc_*.h) on the
 * CPU wave emulator so it can be debugged and diffed against the oracle in a GPU-less container.  Exposes a dump hook
 * for intermediate values.  Never part of the product library. */
#include "wave_emu.h"
#include <vector>
#include <string>
typedef void (*dump_fn)(const char *tag, const void *p, int nbytes);
static dump_fn g_dump = nullptr;
extern "C" void emu_set_dump(dump_fn f) { g_dump = f; }
#define K_DUMP(tag, ptr, nbytes) do { if (g_dump && wv_lane() == 0) g_dump(tag, (const void *)(ptr), nbytes); } while (0)
#define K_DUMPI(tag, v) do { int32_t v__ = (int32_t)(v); if (g_dump && wv_lane() == 0) g_dump(tag, &v__, 4); } while (0)
#define K_DUMP_ENABLED 1
#include "celt_enc_all.h"
#include "celt_dec_all.h"

struct Job { FrameLds *L; OaStream *gs; const int16_t *pcm; int frame_size, max_bytes; uint8_t *out; int32_t *len; uint32_t *rng; };
static void job_entry(void *p)
{
   Job *j = (Job *)p;
   oa_encode_frame(j->L, j->gs, j->pcm, j->frame_size, j->max_bytes, j->out, 1 << 20, j->len, j->rng);
}
extern "C" int emu_sizeof_stream() { return (int)sizeof(OaStream); }
extern "C" int emu_sizeof_lds() { return (int)sizeof(FrameLds); }
extern "C" void emu_encode_batch(OaStream *streams, const int16_t *pcm, int S, int frame_size, int max_bytes,
      uint8_t *out, int stride, int32_t *lens, uint32_t *rngs)
{
   for (int s = 0; s < S; s++) {
      FrameLds *L = (FrameLds *)aligned_alloc(64, (sizeof(FrameLds) + 63) & ~63);
      memset(L, 0xA5, sizeof(FrameLds));          /* LDS is uninitialised on the GPU: make stale reads loud */
      CeltScratch *cs = (CeltScratch *)malloc(sizeof(CeltScratch)); memset(cs, 0xA5, sizeof(CeltScratch)); L->g = cs;
      streams[s].analysis_off = 1;              /* this harness is checked against the restatement oracle / the reference built with DISABLE_FLOAT_API */
      Job j = {L, streams + s, pcm + (size_t)s * frame_size * streams[s].cfg.channels, frame_size, max_bytes, out + (size_t)s * stride, lens + s, rngs + s};
      emu_run_wave(job_entry, &j);
      free(L); free(cs);
   }
}

/* ---- decoder ---- */
static int g_decode_fec = 0;
extern "C" void emu_set_decode_fec(int v) { g_decode_fec = v; }
struct DJob { DecLds *L; OaDecStream *gs; const uint8_t *data; int len, frame_size; int16_t *pcm; int32_t *ns; uint32_t *rng; };
static void djob_entry(void *p)
{
   DJob *j = (DJob *)p;
   oa_decode_packet(j->L, j->gs, j->data, j->len, j->frame_size, j->pcm, j->ns, j->rng, g_decode_fec);
}
extern "C" int emu_sizeof_dec_stream() { return (int)sizeof(OaDecStream); }
extern "C" void emu_dec_stream_reset(OaDecStream *st, int channels) { oa_dec_stream_reset(st, channels); }
extern "C" int emu_sizeof_dec_lds() { return (int)sizeof(DecLds); }
extern "C" void emu_decode_batch(OaDecStream *streams, const uint8_t *data, int stride, const int32_t *lens, int S, int frame_size,
      int16_t *pcm, int pcm_stride, int32_t *ns, uint32_t *rngs)
{
   for (int s = 0; s < S; s++) {
      DecLds *L = (DecLds *)aligned_alloc(64, (sizeof(DecLds) + 63) & ~63);
      memset(L, 0xA5, sizeof(DecLds));
      static thread_local int32_t xg[OA_DEC_SCRATCH_BYTES / 4];                     /* the wave's spectrum + folding-memory scratch (what the kernel points L->Xg at) */
      memset(xg, 0xA5, sizeof xg); L->Xg = xg;
      DJob j = {L, streams + s, data + (size_t)s * stride, lens[s], frame_size, pcm + (size_t)s * pcm_stride, ns + s, rngs + s};
      emu_run_wave(djob_entry, &j);
      free(L);
   }
}

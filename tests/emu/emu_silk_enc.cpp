/* tests/emu/emu_silk_enc.cpp — TEST INFRASTRUCTURE: runs the SILK encoder body (opus_amd/csrc/silk_enc*.h — the exact device source) on the CPU wave
 * emulator, below the Opus layer, with the same stage taps as the reference shim oracle/ref_expose/x_silk_enc.c.  Never part of the product. */
#include "wave_emu.h"
typedef void (*dump_fn)(const char *tag, const void *p, int nbytes);
static dump_fn g_dump = nullptr;
extern "C" void emu_set_dump(dump_fn f) { g_dump = f; }
#define K_DUMP(tag, ptr, nbytes) do { if (g_dump && wv_lane() == 0) g_dump(tag, (const void *)(ptr), nbytes); } while (0)
#define K_DUMPI(tag, v) do { int32_t v__ = (int32_t)(v); if (g_dump && wv_lane() == 0) g_dump(tag, &v__, 4); } while (0)
#define K_DUMP_ENABLED 1
static inline unsigned atomicAdd(unsigned *p, unsigned v) { const unsigned o = *p; *p = o + v; return o; }   /* (opus_sh_split.h; fibers are cooperative) */
#include "celt_enc_all.h"
#include "celt_dec_all.h"
#include "silk_enc_all.h"

struct EJob { SeRateScratch *G; SilkEncLds *S; OaSilkEnc *gs; int32_t *ctl; const int16_t *pcm; int nSamples; uint8_t *out; int out_cap; int32_t *res; int activity; int ret; };
static void ejob(void *p)
{
   EJob *j = (EJob *)p;
   SilkEncLds *S = j->S;
   const int lane = wv_lane();
   { int32_t *d = (int32_t *)se_st(S); const int32_t *g = (const int32_t *)j->gs; FOR_LANES(i, (int)(sizeof(OaSilkEnc) / 4)) d[i] = g[i]; }
   wv_sync();
   SeControl c; memset(&c, 0, sizeof c);
   int32_t *w = (int32_t *)&c; for (int i = 0; i < 18; i++) w[i] = j->ctl[i];
   /* the packet buffer and coder context stand where the Opus layer keeps them */
   static thread_local EcCtx ecl; static thread_local uint8_t buf[1280]; static thread_local int16_t pcm16[2 * 2880];
   LANE0 { EcCtx e_; EcCtx *e = &e_; uint8_t *b = buf; (void)b; k_ec_enc_init(e, buf, (u32)j->out_cap); ec_st(&ecl, e); for (int i = 0; i < j->nSamples * c.nChannelsAPI; i++) pcm16[i] = j->pcm[i]; }
   static OaSilkLbrr lbrr_store;                                                   /* SILK-level harness: one encoder at a time */
   const int ret = silk_encode_wave(S, &c, pcm16, j->nSamples, &ecl, buf, j->activity, j->G, &lbrr_store);
   wv_sync();
   LANE0 {
      j->ret = ret;
      j->ctl[14] = c.maxBits; j->ctl[18] = c.internalSampleRate; j->ctl[19] = c.allowBandwidthSwitch; j->ctl[20] = c.inWBmodeWithoutVariableLP; j->ctl[21] = c.stereoWidth_Q14; j->ctl[22] = c.switchReady;
      j->ctl[23] = c.signalType; j->ctl[24] = c.offset;
      EcCtx e_; ec_ld(&e_, &ecl); EcCtx *e = &e_;
      j->res[0] = S->r[0]; j->res[1] = k_ec_tell(e, buf); j->res[2] = (int32_t)e->rng;
      k_ec_enc_done(e, buf);
      for (int i = 0; i < j->out_cap; i++) j->out[i] = buf[i];
   }
   wv_sync();
   { const int32_t *d = (const int32_t *)se_st(S); int32_t *g = (int32_t *)j->gs; FOR_LANES(i, (int)(sizeof(OaSilkEnc) / 4)) g[i] = d[i]; }
   (void)lane;
}
extern "C" int emu_silk_enc_size() { return (int)sizeof(OaSilkEnc); }
extern "C" int emu_silk_enc_lds_size() { return (int)sizeof(SilkEncLds); }
extern "C" void emu_silk_enc_init(OaSilkEnc *st) { oa_silk_enc_reset(st); }
extern "C" int emu_silk_encode(OaSilkEnc *st, int32_t *ctl, const int16_t *pcm, int nSamples, uint8_t *out, int out_cap, int32_t *res, int activity)
{
   SilkEncLds *S = (SilkEncLds *)aligned_alloc(64, (sizeof(SilkEncLds) + 63) & ~63);
   memset(S, 0xA5, sizeof(SilkEncLds));
   S->st_off = (int32_t)offsetof(SilkEncLds, st);                                  /* (what the kernels set before a call) */
   SeRateScratch *G = (SeRateScratch *)malloc(sizeof(SeRateScratch));
   EJob j = {G, S, st, ctl, pcm, nSamples, out, out_cap, res, activity, 0};
   emu_run_wave(ejob, &j);
   free(S); free(G);
   return j.ret;
}

/* ---- the whole Opus-layer frame (opus_enc_sh.h) ---- */
struct ShJob { CeltScratch *cs; SeRateScratch *G; ShLds *L; OaShStream *gs; const int16_t *pcm; int frame_size, max_bytes; uint8_t *out; int out_cap; int16_t *pcm_hp; int32_t *len; uint32_t *rng; };
static void shjob(void *p) { ShJob *j = (ShJob *)p; oa_sh_encode_frame(j->L, j->gs, j->pcm, j->frame_size, j->max_bytes, j->out, j->out_cap, j->pcm_hp, j->G, j->cs, j->len, j->rng); }
extern "C" int emu_sh_stream_size() { return (int)sizeof(OaShStream); }
extern "C" int emu_sh_lds_size() { return (int)sizeof(ShLds); }
extern "C" void emu_sh_stream_init(OaShStream *st, int Fs, int channels, int application) { oa_sh_stream_init(st, Fs, channels, application); st->cfg.analysis_off = 1; }   /* this harness is checked against the reference built with DISABLE_FLOAT_API */
extern "C" void emu_sh_set_cfg(OaShStream *st, int word, int value) { ((int32_t *)&st->cfg)[word] = value; }
extern "C" void emu_sh_encode(OaShStream *st, const int16_t *pcm, int frame_size, int max_bytes, uint8_t *out, int out_cap, int32_t *len, uint32_t *rng)
{
   const size_t po = (sizeof(ShLds) + 15) & ~(size_t)15, tot = (po + SH_PKT_BYTES + 63) & ~(size_t)63;
   ShLds *L = (ShLds *)aligned_alloc(64, tot);
   memset(L, 0xA5, tot);
   L->silk_tail = 1; L->packet_off = (i32)po; L->S.st_off = (i32)offsetof(SilkEncLds, st);                                      /* what oa_sh_encode_kernel sets before a call: the tails are staged, the packet buffer sits behind the rest */
   int16_t *hp = (int16_t *)malloc(2 * SH_PCM_BYTES(frame_size, 2) + 512);          /* high-passed frame | faded CELT input | 2.5 ms CELT prefill */
   SeRateScratch *G = (SeRateScratch *)malloc(sizeof(SeRateScratch));
   CeltScratch *cs = (CeltScratch *)malloc(sizeof(CeltScratch)); memset(cs, 0xA5, sizeof(CeltScratch));
   ShJob j = {cs, G, L, st, pcm, frame_size, max_bytes, out, out_cap, hp, len, rng};
   emu_run_wave(shjob, &j);
   free(hp); free(L); free(G); free(cs);
}

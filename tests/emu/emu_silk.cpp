/* tests/emu/emu_silk.cpp — TEST INFRASTRUCTURE: runs the SILK quantiser kernel bodies (opus_amd/csrc/silk_nsq*.h) on the CPU
 * wave emulator, 64 fibers per tile, so the exact device source is checked against the oracle without a GPU. */
#include "wave_emu.h"
#include "fx.h"
#include "silk_nsq.h"
#include "silk_nsq_dd.h"
#include "silk_host.h"
#include "silk_lpc.h"
#include "silk_resampler.h"
#include "silk_pitch.h"

static int g_generic = 0;
extern "C" void emu_nsq_force_generic(int g) { g_generic = g; }   /* run the runtime-order instantiation even for specialised orders */
struct NsqJob { OaNsqCfg cfg; int32_t *tile; const OaNsqFrame *frames; const int16_t *x16; int8_t *pulses; int8_t *seed_out; int first, n; int generic; };
static void nsq_entry(void *arg)
{
   NsqJob *j = (NsqJob *)arg;
   int lane = wv_lane(), sidx = j->first + lane; bool act = sidx < j->n; if (!act) sidx = j->first;
   const int frame = j->cfg.nb_subfr * 5 * j->cfg.fs_kHz;
   NsqMem m = nsq_mem(j->tile, 64, lane, 20 * j->cfg.fs_kHz + frame);
#define CALL(SS) silk_nsq_lane<SS>(j->cfg, m, &j->frames[sidx], j->x16 + (size_t)sidx * frame, j->pulses + (size_t)sidx * frame, act)
   switch (j->generic ? 0 : j->cfg.shapingLPCOrder) { case 12: CALL(12); break; case 14: CALL(14); break; case 16: CALL(16); break; case 20: CALL(20); break; case 24: CALL(24); break; default: CALL(0); }
#undef CALL
}
extern "C" long emu_nsq_tile_words(int T) { return (long)oa_nsq_tile_words(T); }
extern "C" void emu_nsq_import(int32_t *tile, int T, int t, const OaNsqRefState *r, const OaNsqCfg *cfg) { oa_nsq_import(tile, T, t, r, cfg); }
extern "C" void emu_nsq_export(const int32_t *tile, int T, int t, OaNsqRefState *r, const OaNsqCfg *cfg) { oa_nsq_export(tile, T, t, r, cfg); }
extern "C" void emu_silk_nsq(const OaNsqCfg *cfg, int32_t *tiles, const OaNsqFrame *frames, const int16_t *x16, int8_t *pulses, int n)
{
   const long tw = (long)oa_nsq_tile_words(64);
   for (int first = 0, tl = 0; first < n; first += 64, tl++) {
      NsqJob j = { *cfg, tiles + tl * tw, frames, x16, pulses, nullptr, first, n, g_generic };
      emu_run_wave(nsq_entry, &j);
   }
}

static void nsq_dd_entry(void *arg)
{
   NsqJob *j = (NsqJob *)arg;
   int lane = wv_lane(), sidx = j->first + (lane >> 2); bool act = sidx < j->n; if (!act) sidx = j->first;
   const int frame = j->cfg.nb_subfr * 5 * j->cfg.fs_kHz;
   NsqMem m = nsq_mem(j->tile, 16, lane >> 2, 20 * j->cfg.fs_kHz + frame);
   int32_t *ring = j->tile + (oa_nsq_tile_words(16) - 5 * OA_SILK_DD * 64);
#define CALL(SS) silk_nsq_dd_wave<SS>(j->cfg, m, ring, &j->frames[sidx], j->x16 + (size_t)sidx * frame, j->pulses + (size_t)sidx * frame, j->seed_out + sidx, act)
   switch (j->generic ? 0 : j->cfg.shapingLPCOrder) { case 12: CALL(12); break; case 14: CALL(14); break; case 16: CALL(16); break; case 20: CALL(20); break; case 24: CALL(24); break; default: CALL(0); }
#undef CALL
}
extern "C" void emu_silk_nsq_dd(const OaNsqCfg *cfg, int32_t *tiles, const OaNsqFrame *frames, const int16_t *x16, int8_t *pulses, int8_t *seed_out, int n)
{
   const long tw = (long)oa_nsq_tile_words(16);
   for (int first = 0, tl = 0; first < n; first += 16, tl++) {
      NsqJob j = { *cfg, tiles + tl * tw, frames, x16, pulses, seed_out, first, n, g_generic };
      emu_run_wave(nsq_dd_entry, &j);
   }
}

struct LpcJob { LpcLds lds; int16_t *out; const int16_t *in, *B; int len, d; };
static void lpc_entry(void *arg) { LpcJob *j = (LpcJob *)arg; silk_lpc_analysis_filter_wave(&j->lds, j->out, j->in, j->B, j->len, j->d); }
extern "C" void emu_silk_lpc_analysis_filter(int n, int16_t *out, const int16_t *in, const int16_t *B, int len, int d)
{
   for (int s = 0; s < n; s++) { LpcJob *j = new LpcJob; j->out = out + (size_t)s * len; j->in = in + (size_t)s * len; j->B = B + (size_t)s * d; j->len = len; j->d = d; emu_run_wave(lpc_entry, j); delete j; }
}

struct RsJob { ResamplerLds lds; OaResamplerCfg cfg; int32_t *state; int n, first; const int16_t *in; int inLen; int16_t *out; int outLen; };
static void rs_entry(void *arg)
{
   RsJob *j = (RsJob *)arg; int ch = j->first + wv_lane(); if (ch >= j->n) return;
   silk_resampler_lane(j->cfg, &j->lds, j->state + ch, j->n, j->in + (size_t)ch * j->inLen, j->inLen, j->out + (size_t)ch * j->outLen, wv_lane());
}
/* state: [OA_RS_ROWS][n] int32 rows exactly as the device keeps them; cfg: the nine OaResamplerCfg words */
extern "C" void emu_silk_resampler(const OaResamplerCfg *cfg, int32_t *state, int n, const int16_t *in, int inLen, int16_t *out, int outLen)
{
   for (int first = 0; first < n; first += 64) { RsJob *j = new RsJob; j->cfg = *cfg; j->state = state; j->n = n; j->first = first; j->in = in; j->inLen = inLen; j->out = out; j->outLen = outLen; emu_run_wave(rs_entry, j); delete j; }
}

struct PeJob { PitchLds lds; OaPitchCfg cfg; const int16_t *frame; const OaPitchIn *in; OaPitchOut *out; };
static void pe_entry(void *arg) { PeJob *j = (PeJob *)arg; (void)silk_pitch_analysis_wave(j->cfg, &j->lds, j->lds.frame, j->frame, j->in, j->out); }
extern "C" void emu_silk_pitch(int n, const int16_t *frames, const OaPitchIn *in, OaPitchOut *out, int Fs_kHz, int complexity, int nb_subfr)
{
   const int flen = (20 + 5 * nb_subfr) * Fs_kHz;
   for (int s = 0; s < n; s++) { PeJob *j = new PeJob; j->cfg = { Fs_kHz, complexity, nb_subfr }; j->frame = frames + (size_t)s * flen; j->in = in + s; j->out = out + s; emu_run_wave(pe_entry, j); delete j; }
}

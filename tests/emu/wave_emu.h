/* tests/emu/wave_emu.h — TEST INFRASTRUCTURE.  A CPU stand-in for opus_amd/csrc/wave.h so the *same* kernel
 * body source can be exercised in this GPU-less container (gdb, dumps, bit-exact diffs against the oracle)
 * before it is sent to a real MI355X.  64 fibers = 64 lanes; every wave primitive is a rendezvous of all
 * lanes; a lane that skips a collective is reported (divergence on the GPU would hang or read garbage).
 * The product never includes this file. */
#ifndef WAVE_EMU_H
#define WAVE_EMU_H
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <execinfo.h>
static inline void emu_die() { void *bt[32]; int n = backtrace(bt, 32); backtrace_symbols_fd(bt, n, 2); abort(); }

#define WV_DEV  static inline
#define WV_DEVN static
#define WV_MEM  inline
#define WV_HD   static inline
#define WV_LDS
#define WV_TABLE static const
#define WV_WIDTH 64

struct EmuFiber { void *sp; char *stack; int done; long nsync; long nx; int uni_k; long gsync; long gnx; };
struct EmuWave {
   EmuFiber f[64];
   void *main_sp;
   int cur;
   int64_t xch[2][64][4];
   int64_t gxch[2][64][4];             /* the 16-lane group collectives' own exchange tables (groups run between two wave rendezvous independently of each other) */
   int32_t uni_val[1024]; int uni_n;   /* OA_EMU_CHECK_UNI: the values the first-scheduled fiber passed to wv_uni since the last rendezvous */
   unsigned char kind_ring[4096];      /* lane 0's op kind per rendezvous, to catch lanes meeting at different primitives */
   void (*entry)(void *);
   void *arg;
};
extern thread_local EmuWave *emu_cur;
/* OA_EMU_REVERSE=1 runs the fibers in the order lane 63 .. lane 0 instead of 0 .. 63 (logical lane = fiber index ^ emu_flip).  Between two rendezvous a fiber sees the
 * writes of the fibers scheduled before it and none of those after it; code that depends on that (a read of another lane's LDS write with no wv_sync in between) gives
 * different results under the two orders, which is how such races -- invisible in either order alone, real on the lockstep hardware -- are found here. */
extern int emu_flip, emu_check_uni;
extern "C" void emu_switch(void **save_sp, void *new_sp);
void emu_run_wave(void (*entry)(void *), void *arg);

WV_DEV int wv_lane() { return emu_cur->cur ^ emu_flip; }
static inline void emu_rendezvous(int kind = 0)
{
   EmuWave *w = emu_cur;
   int me = w->cur, nxt = (me + 1) & 63;
   w->f[me].nsync++;
   if (me == 0) w->uni_n = w->f[0].uni_k;
   if (me == 0) w->kind_ring[w->f[0].nsync & 4095] = (unsigned char)kind;
   else if (w->kind_ring[w->f[me].nsync & 4095] != kind) {
      fprintf(stderr, "wave_emu: lane %d meets lane 0 at rendezvous %ld with a different primitive (%d vs %d): divergent control flow\n", me, w->f[me].nsync, kind, w->kind_ring[w->f[me].nsync & 4095]); emu_die();
   }
   if (w->f[nxt].done) { fprintf(stderr, "wave_emu: lane %d reached a collective but lane %d already exited (divergence)\n", me, nxt); abort(); }
   w->cur = nxt;
   emu_switch(&w->f[me].sp, w->f[nxt].sp);
   w->f[me].uni_k = 0;
   int prv = (me + 63) & 63;
   long expect = w->f[me].nsync + (me == 0 ? 0 : 1);          /* the previous lane is parked at the next rendezvous */
   const bool in_group_op = (prv >> 4) == (me >> 4) && w->f[prv].nsync == w->f[me].nsync && w->f[prv].gsync == w->f[me].gsync + 1;   /* the previous lane woke this one from a group collective (wg_*) it reached after this rendezvous */
   if (!(w->f[prv].nsync == expect || in_group_op || (w->f[prv].done && w->f[prv].nsync == w->f[me].nsync))) {
      fprintf(stderr, "wave_emu: divergent collectives: lane %d at %ld, lane %d at %ld\n", me, w->f[me].nsync, prv, w->f[prv].nsync); abort();
   }
}
WV_DEV void wv_sync() { emu_rendezvous(); }
WV_DEV void wv_order() { emu_rendezvous(); }
WV_DEV void wv_prio_serial() {}
WV_DEV void wv_prio_normal() {}
/* publish (a,b,c,d) for this lane, rendezvous, return this op's table [64][4] */
static inline int64_t (*emu_xchg(int64_t a, int64_t b = 0, int64_t c = 0, int64_t d = 0))[4]
{
   EmuWave *w = emu_cur;
   int fib = w->cur, me = fib ^ emu_flip;
   int p = (int)(w->f[fib].nx++ & 1);
   w->xch[p][me][0] = a; w->xch[p][me][1] = b; w->xch[p][me][2] = c; w->xch[p][me][3] = d;
   emu_rendezvous(1);
   return w->xch[p];
}
WV_DEV int32_t wv_shfl(int32_t v, int src) { auto t = emu_xchg(v); return (int32_t)t[src & 63][0]; }
template <int Q> WV_DEV int32_t wv_lane_const(int32_t v) { auto t = emu_xchg(v); return (int32_t)t[Q][0]; }
template <int J> WV_DEV int32_t wv_quad_bcast(int32_t v) { auto t = emu_xchg(v); return (int32_t)t[(wv_lane() & ~3) | J][0]; }
WV_DEV int32_t wv_bcast(int32_t v, int src)
{
   auto t = emu_xchg(v, src);
   if (emu_check_uni && t[wv_lane()][1] != t[0][1]) { fprintf(stderr, "wave_emu: wv_bcast (v_readlane) with a lane index that is not uniform: lane %d asks for %d, lane 0 for %d\n", wv_lane(), src, (int)t[0][1]); emu_die(); }
   if ((unsigned)src > 63u) { fprintf(stderr, "wave_emu: wv_bcast from lane %d\n", src); emu_die(); }
   return (int32_t)t[src][0];
}
/* on the hardware: v_readfirstlane, i.e. the value of the first active lane in EVERY lane.  Here each lane keeps its own value, so a caller that passes a value that
 * is not in fact uniform behaves differently on the GPU; OA_EMU_CHECK_UNI=1 compares, call by call between two rendezvous, every lane's argument with the one the
 * first-scheduled lane passed (calls inside one-lane sections have nothing to be compared with and are skipped). */
WV_DEV int32_t wv_uni(int32_t v)
{
   if (emu_check_uni) {
      EmuWave *w = emu_cur; const int fib = w->cur, k = w->f[fib].uni_k++;
      if (fib == 0) { if (k < 1024) w->uni_val[k] = v; }
      else if (k < w->uni_n && k < 1024 && w->uni_val[k] != v) { fprintf(stderr, "wave_emu: wv_uni of a non-uniform value: lane %d passes %d, lane %d passed %d (call %d since the last rendezvous)\n", wv_lane(), v, emu_flip, w->uni_val[k], k); emu_die(); }
   }
   return v;
}
WV_DEV int32_t wv_shift_down1(int32_t v, int32_t fill) { auto t = emu_xchg(v); int me = wv_lane(); return me == 63 ? fill : (int32_t)t[me + 1][0]; }
WV_DEV int32_t wv_shift_up1(int32_t v, int32_t fill) { auto t = emu_xchg(v); int me = wv_lane(); return me == 0 ? fill : (int32_t)t[me - 1][0]; }
WV_DEV int32_t wv_writelane(int32_t val, int lane, int32_t old) { return wv_lane() == lane ? val : old; }
WV_DEV float wv_rcpf(float x) { return 1.0f / x; }
WV_DEV int32_t wv_sum(int32_t v) { auto t = emu_xchg(v); uint32_t s = 0; for (int i = 0; i < 64; i++) s += (uint32_t)t[i][0]; return (int32_t)s; }
WV_DEV uint32_t wv_sumu(uint32_t v) { auto t = emu_xchg(v); uint32_t s = 0; for (int i = 0; i < 64; i++) s += (uint32_t)t[i][0]; return s; }
WV_DEV int64_t wv_sum64(int64_t v) { auto t = emu_xchg(v); uint64_t s = 0; for (int i = 0; i < 64; i++) s += (uint64_t)t[i][0]; return (int64_t)s; }
WV_DEV int32_t wv_max(int32_t v) { auto t = emu_xchg(v); int32_t m = (int32_t)t[0][0]; for (int i = 1; i < 64; i++) if ((int32_t)t[i][0] > m) m = (int32_t)t[i][0]; return m; }
WV_DEV int32_t wv_min(int32_t v) { auto t = emu_xchg(v); int32_t m = (int32_t)t[0][0]; for (int i = 1; i < 64; i++) if ((int32_t)t[i][0] < m) m = (int32_t)t[i][0]; return m; }
WV_DEV uint32_t wv_or(uint32_t v) { auto t = emu_xchg(v); uint32_t m = 0; for (int i = 0; i < 64; i++) m |= (uint32_t)t[i][0]; return m; }
WV_DEV uint64_t wv_ballot(int pred) { auto t = emu_xchg(pred != 0); uint64_t m = 0; for (int i = 0; i < 64; i++) m |= (uint64_t)(t[i][0] != 0) << i; return m; }
WV_DEV int32_t wv_scan_incl(int32_t v) { auto t = emu_xchg(v); uint32_t s = 0; int me = wv_lane(); for (int i = 0; i <= me; i++) s += (uint32_t)t[i][0]; return (int32_t)s; }
WV_DEV int wv_argmax_ratio_packed(uint32_t num, uint32_t den, bool valid, int nlanes)
{
   /* same contract as the DPP version: maximal num/den by exact cross-multiplication among valid lanes, lowest lane wins ties */
   auto t = emu_xchg(num, den, valid ? 1 : 0);
   (void)nlanes;
   int best = -1;
   for (int i = 0; i < 64; i++) {
      if (!t[i][2]) continue;
      if (best < 0 || (uint64_t)t[best][1] * (uint64_t)t[i][0] > (uint64_t)t[i][1] * (uint64_t)t[best][0]) best = i;
   }
   return best;
}
WV_DEV void wv_argmax_ratio(int32_t &num, int32_t &den, int32_t &idx)
{
   auto t = emu_xchg(num, den, idx);
   int32_t bn = (int32_t)t[0][0], bd = (int32_t)t[0][1], bi = (int32_t)t[0][2];
   for (int i = 1; i < 64; i++) {
      int32_t n2 = (int32_t)t[i][0], d2 = (int32_t)t[i][1], i2 = (int32_t)t[i][2];
      int32_t lhs = (int32_t)(int16_t)bd * (int32_t)(int16_t)n2, rhs = (int32_t)(int16_t)d2 * (int32_t)(int16_t)bn;
      if (lhs > rhs || (lhs == rhs && i2 < bi)) { bn = n2; bd = d2; bi = i2; }
   }
   num = bn; den = bd; idx = bi;
}
/* ---- 16-lane groups (wave.h: wg_*): a rendezvous of the caller's 16 lanes only.  The fibers of a group cycle among themselves; the other groups of the wave run when this
 * group's lanes reach the next WAVE rendezvous.  So four groups can sit in different branches of the kernel (on the GPU: EXEC masks), each with its own collectives. ---- */
#define WG_WIDTH 16
WV_DEV int wg_lane() { return (emu_cur->cur ^ emu_flip) & 15; }
WV_DEV int wg_id() { return (emu_cur->cur ^ emu_flip) >> 4; }
static inline void emu_grendezvous()
{
   EmuWave *w = emu_cur;
   const int me = w->cur, nxt = (me & ~15) | ((me + 1) & 15), prv = (me & ~15) | ((me + 15) & 15);
   w->f[me].gsync++;
   if (w->f[nxt].done) { fprintf(stderr, "wave_emu: lane %d reached a group collective but lane %d of its group already exited\n", me, nxt); emu_die(); }
   w->cur = nxt;
   emu_switch(&w->f[me].sp, w->f[nxt].sp);
   /* who woke this lane: the previous lane of the group, parked at the group's next collective, or -- no further group collective -- at the next wave rendezvous, or gone */
   const bool ok = (me & 15) == 0 ? (w->f[prv].gsync == w->f[me].gsync && w->f[prv].nsync == w->f[me].nsync)
                 : ((w->f[prv].gsync == w->f[me].gsync + 1 && w->f[prv].nsync == w->f[me].nsync) || (w->f[prv].gsync == w->f[me].gsync && (w->f[prv].nsync == w->f[me].nsync + 1 || w->f[prv].done)));
   if (!ok) {
      fprintf(stderr, "wave_emu: divergent group collectives: lane %d at %ld (wave %ld), lane %d at %ld (wave %ld)\n", me, w->f[me].gsync, w->f[me].nsync, prv, w->f[prv].gsync, w->f[prv].nsync); emu_die();
   }
}
static inline int64_t (*emu_gxchg(int64_t a, int64_t b = 0, int64_t c = 0, int64_t d = 0))[4]
{
   EmuWave *w = emu_cur;
   const int fib = w->cur, me = fib ^ emu_flip;
   const int p = (int)(w->f[fib].gnx++ & 1);
   w->gxch[p][me][0] = a; w->gxch[p][me][1] = b; w->gxch[p][me][2] = c; w->gxch[p][me][3] = d;
   emu_grendezvous();
   return w->gxch[p];
}
WV_DEV void wg_sync() { emu_grendezvous(); }
WV_DEV bool wv_any(int pred) { auto t = emu_xchg(pred != 0); for (int i = 0; i < 64; i++) if (t[i][0]) return true; return false; }
#define EMU_GBASE ((emu_cur->cur ^ emu_flip) & 48)
WV_DEV int32_t wg_sum(int32_t v) { auto t = emu_gxchg(v); const int g = EMU_GBASE; uint32_t s = 0; for (int i = 0; i < 16; i++) s += (uint32_t)t[g + i][0]; return (int32_t)s; }
WV_DEV uint32_t wg_sumu(uint32_t v) { return (uint32_t)wg_sum((int32_t)v); }
WV_DEV int64_t wg_sum64(int64_t v) { auto t = emu_gxchg(v); const int g = EMU_GBASE; uint64_t s = 0; for (int i = 0; i < 16; i++) s += (uint64_t)t[g + i][0]; return (int64_t)s; }
WV_DEV int32_t wg_max(int32_t v) { auto t = emu_gxchg(v); const int g = EMU_GBASE; int32_t m = (int32_t)t[g][0]; for (int i = 1; i < 16; i++) if ((int32_t)t[g + i][0] > m) m = (int32_t)t[g + i][0]; return m; }
WV_DEV uint32_t wg_or(uint32_t v) { auto t = emu_gxchg(v); const int g = EMU_GBASE; uint32_t m = 0; for (int i = 0; i < 16; i++) m |= (uint32_t)t[g + i][0]; return m; }
WV_DEV int32_t wg_scan_incl(int32_t v) { auto t = emu_gxchg(v); const int g = EMU_GBASE, me = wg_lane(); uint32_t s = 0; for (int i = 0; i <= me; i++) s += (uint32_t)t[g + i][0]; return (int32_t)s; }
WV_DEV int32_t wg_bcast(int32_t v, int src)
{
   auto t = emu_gxchg(v, src); const int g = EMU_GBASE;
   if ((unsigned)src > 15u) { fprintf(stderr, "wave_emu: wg_bcast from lane %d of the group\n", src); emu_die(); }
   if (emu_check_uni && t[g + wg_lane()][1] != t[g][1]) { fprintf(stderr, "wave_emu: wg_bcast with a lane index that is not uniform in the group: lane %d asks for %d, the group's lane 0 for %d\n", wv_lane(), src, (int)t[g][1]); emu_die(); }
   return (int32_t)t[g + src][0];
}
template <int D> WV_DEV int32_t wg_shl(int32_t v) { auto t = emu_gxchg(v); const int g = EMU_GBASE, me = wg_lane(); return me + D < 16 ? (int32_t)t[g + me + D][0] : 0; }
template <int D> WV_DEV int32_t wg_shr(int32_t v) { auto t = emu_gxchg(v); const int g = EMU_GBASE, me = wg_lane(); return me - D >= 0 ? (int32_t)t[g + me - D][0] : 0; }
WV_DEV uint32_t wg_ballot(int pred) { auto t = emu_gxchg(pred != 0); const int g = EMU_GBASE; uint32_t m = 0; for (int i = 0; i < 16; i++) m |= (uint32_t)(t[g + i][0] != 0) << i; return m; }
WV_DEV int wg_argmax_ratio_packed(uint32_t num, uint32_t den, bool valid)
{
   auto t = emu_gxchg(num, den, valid ? 1 : 0); const int g = EMU_GBASE;
   int best = -1;
   for (int i = 0; i < 16; i++) {
      if (!t[g + i][2]) continue;
      if (best < 0 || (uint64_t)t[g + best][1] * (uint64_t)t[g + i][0] > (uint64_t)t[g + i][1] * (uint64_t)t[g + best][0]) best = i;
   }
   return best < 0 ? 16 : best;
}
#endif

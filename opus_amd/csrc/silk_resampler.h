/* silk_resampler.h — silk_resampler (silk/resampler.c:183) as a lane-per-channel kernel body: 64 independent channels of one rate
 * pair per wave, all lanes in lock-step (same rates and block length => identical control flow, no divergence).
 *
 * The three algorithms of the reference are all "short recursive prefilter + FIR interpolation":
 *   down  (resampler_private_down_FIR.c:145)  2nd-order AR (resampler_private_AR2.c:36) then an 18/24/36-tap polyphase FIR at
 *                                            index steps of invRatio_Q16, restarted every <= 10 ms batch;
 *   up 2x (resampler_private_up2_HQ.c:38)     two 3-section allpass chains;
 *   up    (resampler_private_IIR_FIR.c:65)    the 2x upsampler then an 8-tap fractional FIR (12 phases).
 * The recursions are not associative in fixed point (each step floors), so a channel is a serial chain; the reference buffers a
 * whole batch (<= 516 int32) between the two stages, here the FIR consumes the prefilter output as it is produced, through a
 * 64-entry ring per lane in LDS ([slot][lane]: conflict-free), so the working set is 16 KB per wave whatever the block length.
 * Filter state (sIIR, the FIR tail, the 1 ms delay line) lives in HBM as [word][channel], coalesced across lanes. */
#ifndef OPUS_AMD_SILK_RESAMPLER_H
#define OPUS_AMD_SILK_RESAMPLER_H
#include "silk_tables.h"

enum { OA_RS_FN_COPY = 0, OA_RS_FN_UP2 = 1, OA_RS_FN_IIR_FIR = 2, OA_RS_FN_DOWN_FIR = 3 };
enum { OA_RS_NONE = 0, OA_RS_3_4, OA_RS_2_3, OA_RS_1_2, OA_RS_1_3, OA_RS_1_4, OA_RS_1_6 };
struct OaResamplerCfg { i32 resampler_function, batchSize, invRatio_Q16, FIR_Order, FIR_Fracs, Fs_in_kHz, Fs_out_kHz, inputDelay, coefs_id; };
/* one channel's state in the reference's field order (silk/resampler_structs.h:38-52; the Coefs pointer is a table id) — import/export format */
struct OaResamplerState { i32 sIIR[6]; union { i32 w32[36]; i16 w16[36]; } sFIR; i16 delayBuf[96]; OaResamplerCfg cfg; };
/* device state rows ([row][nchannels]): 0..5 sIIR, 6..41 FIR tail (one value per row), 42..89 delay line */
enum { OA_RS_ROW_IIR = 0, OA_RS_ROW_FIR = 6, OA_RS_ROW_DELAY = 42, OA_RS_ROWS = 90 };
#define OA_RS_RING 64
template <int W> struct ResamplerLdsT { i32 ring[OA_RS_RING][W]; };    /* W = 64: one column per lane (batch kernel); W = 1: a single channel on lane 0 (decoder) */
typedef ResamplerLdsT<WV_WIDTH> ResamplerLds;

/* silk_resampler_init (silk/resampler.c:79-178): delay-compensation matrices :52-67, method selection, rounded-up Q16 ratio */
WV_HD int rs_init_cfg(OaResamplerCfg *S, i32 Fs_in, i32 Fs_out, int forEnc)
{
   const signed char dEnc[6][3] = { { 6, 0, 3 }, { 0, 7, 3 }, { 0, 1, 10 }, { 0, 2, 6 }, { 18, 10, 12 }, { 0, 0, 44 } };
   const signed char dDec[3][6] = { { 4, 0, 2, 0, 0, 0 }, { 0, 9, 4, 7, 4, 4 }, { 0, 3, 12, 7, 7, 7 } };
   int ri = ((((Fs_in >> 12) - (Fs_in > 16000)) >> (Fs_in > 24000)) - 1), ro = ((((Fs_out >> 12) - (Fs_out > 16000)) >> (Fs_out > 24000)) - 1);
   if (ri > 5) ri = 5; if (ro > 5) ro = 5;
   S->resampler_function = 0; S->batchSize = 0; S->invRatio_Q16 = 0; S->FIR_Order = 0; S->FIR_Fracs = 0; S->Fs_in_kHz = 0; S->Fs_out_kHz = 0; S->inputDelay = 0; S->coefs_id = 0;
   const bool in3 = Fs_in == 8000 || Fs_in == 12000 || Fs_in == 16000, out3 = Fs_out == 8000 || Fs_out == 12000 || Fs_out == 16000;
   if (forEnc) { if (!(in3 || Fs_in == 24000 || Fs_in == 48000) || !out3) return -1; S->inputDelay = dEnc[ri][ro]; }
   else { if (!in3 || !(out3 || Fs_out == 24000 || Fs_out == 48000)) return -1; S->inputDelay = dDec[ri][ro]; }
   S->Fs_in_kHz = Fs_in / 1000; S->Fs_out_kHz = Fs_out / 1000; S->batchSize = S->Fs_in_kHz * 10;
   int up2x = 0;
   if (Fs_out > Fs_in) { if (Fs_out == 2 * Fs_in) S->resampler_function = OA_RS_FN_UP2; else { S->resampler_function = OA_RS_FN_IIR_FIR; up2x = 1; } }
   else if (Fs_out < Fs_in) {
      S->resampler_function = OA_RS_FN_DOWN_FIR;
      if (Fs_out * 4 == Fs_in * 3)      { S->FIR_Fracs = 3; S->FIR_Order = 18; S->coefs_id = OA_RS_3_4; }
      else if (Fs_out * 3 == Fs_in * 2) { S->FIR_Fracs = 2; S->FIR_Order = 18; S->coefs_id = OA_RS_2_3; }
      else if (Fs_out * 2 == Fs_in)     { S->FIR_Fracs = 1; S->FIR_Order = 24; S->coefs_id = OA_RS_1_2; }
      else if (Fs_out * 3 == Fs_in)     { S->FIR_Fracs = 1; S->FIR_Order = 36; S->coefs_id = OA_RS_1_3; }
      else if (Fs_out * 4 == Fs_in)     { S->FIR_Fracs = 1; S->FIR_Order = 36; S->coefs_id = OA_RS_1_4; }
      else if (Fs_out * 6 == Fs_in)     { S->FIR_Fracs = 1; S->FIR_Order = 36; S->coefs_id = OA_RS_1_6; }
      else return -1;
   } else S->resampler_function = OA_RS_FN_COPY;
   S->invRatio_Q16 = ((Fs_in << (14 + up2x)) / Fs_out) << 2;
   while ((i32)(((i64)S->invRatio_Q16 * Fs_out) >> 16) < (Fs_in << up2x)) S->invRatio_Q16++;
   return 0;
}

WV_DEV const i16 *rs_coefs(int id)
{
   switch (id) {
   case OA_RS_3_4: return sk_resampler_3_4_coefs;  case OA_RS_2_3: return sk_resampler_2_3_coefs;  case OA_RS_1_2: return sk_resampler_1_2_coefs;
   case OA_RS_1_3: return sk_resampler_1_3_coefs;  case OA_RS_1_4: return sk_resampler_1_4_coefs;
   }
   return sk_resampler_1_6_coefs;
}

struct RsLane {                      /* per-lane registers */
   i32 iir[6];
   int rb;                           /* ring slot of logical FIR-buffer index 0 of the current batch (wave-uniform) */
};
#define RS_RING(L, slot) (L)->ring[(slot) & (OA_RS_RING - 1)][col]

/* the reference's "source" for one run: sample k of the segment */
template <class DP, class InP> struct RsDelaySrc { DP d; int nd; int stride; InP in; };   /* the delay line (state rows) followed by new input */
template <class InP> struct RsPlainSrc { InP in; };
template <class DP, class InP> WV_DEV i32 rs_at(const RsDelaySrc<DP, InP> &s, int k) { return k < s.nd ? s.d[k * s.stride] : (i32)s.in[k - s.nd]; }
template <class InP> WV_DEV i32 rs_at(const RsPlainSrc<InP> &s, int k) { return (i32)s.in[k]; }

WV_DEV void rs_up2_step(i32 *S, i32 in32, i32 &even, i32 &odd)                    /* resampler_private_up2_HQ.c:60-108 */
{
   for (int ph = 0; ph < 2; ph++) {
      const i16 *c = ph ? sk_resampler_up2_hq_1 : sk_resampler_up2_hq_0;
      i32 *st = S + 3 * ph;
      i32 Y = in32 - st[0], X = sk_mulwb(Y, c[0]);
      i32 o1 = st[0] + X;  st[0] = in32 + X;
      Y = o1 - st[1];  X = sk_mulwb(Y, c[1]);
      i32 o2 = st[1] + X;  st[1] = o1 + X;
      Y = o2 - st[2];  X = sk_mlawb(Y, Y, c[2]);
      o1 = st[2] + X;  st[2] = o2 + X;
      (ph ? odd : even) = sk_sat16(sk_rround(o1, 10));
   }
}

/* One segment (one call of the reference's per-function routine): `len` input samples x[0..len) -> out, returns outputs written. */
template <class RL, class In, class Out> WV_DEV int rs_segment(const OaResamplerCfg c, WV_LDS RL *L, RsLane &r, In x, int len, Out out, const int col)
{
   int no = 0;
   if (c.resampler_function == OA_RS_FN_COPY) { for (int k = 0; k < len; k++) out[k] = (i16)rs_at(x, k); return len; }
   if (c.resampler_function == OA_RS_FN_UP2) {
      for (int k = 0; k < len; k++) { i32 e, o; rs_up2_step(r.iir, shl32(rs_at(x, k), 10), e, o); out[2 * k] = (i16)e; out[2 * k + 1] = (i16)o; }
      return 2 * len;
   }
   const bool down = c.resampler_function == OA_RS_FN_DOWN_FIR;
   const int ord = down ? c.FIR_Order : 8, up = down ? 0 : 1;
   const i16 *C = rs_coefs(c.coefs_id), *F = C + 2;
   for (int done = 0; done < len;) {
      const int nIn = imin(len - done, c.batchSize);
      const i32 max_index_Q16 = shl32(nIn, 16 + up);
      i32 idx = 0;
      for (int k = 0; k < nIn; k++) {
         const i32 s = rs_at(x, done + k);
         if (down) {                                                               /* resampler_private_AR2.c:36 */
            i32 o = r.iir[0] + shl32(s, 8);
            RS_RING(L, r.rb + ord + k) = o;
            o = shl32(o, 2);
            r.iir[0] = sk_mlawb(r.iir[1], o, C[0]);
            r.iir[1] = sk_mulwb(o, C[1]);
         } else {
            i32 e, o; rs_up2_step(r.iir, shl32(s, 10), e, o);
            RS_RING(L, r.rb + ord + 2 * k) = e;  RS_RING(L, r.rb + ord + 2 * k + 1) = o;
         }
         const int have = (k + 1) << up;                                           /* prefilter outputs produced in this batch */
         while (idx < max_index_Q16 && (idx >> 16) <= have) {                      /* every output whose window is complete */
            const int b = r.rb + (idx >> 16);
            i32 v;
            if (!down) {                                                           /* resampler_private_IIR_FIR.c:36 */
               const int ti = sk_mulwb(idx & 0xFFFF, 12);
               const i16 *t0 = &sk_resampler_frac_fir_12[4 * ti], *t1 = &sk_resampler_frac_fir_12[4 * (11 - ti)];
               i32 a = sk_mulbb(RS_RING(L, b), t0[0]);
               a = sk_mlabb(a, RS_RING(L, b + 1), t0[1]); a = sk_mlabb(a, RS_RING(L, b + 2), t0[2]); a = sk_mlabb(a, RS_RING(L, b + 3), t0[3]);
               a = sk_mlabb(a, RS_RING(L, b + 4), t1[3]); a = sk_mlabb(a, RS_RING(L, b + 5), t1[2]); a = sk_mlabb(a, RS_RING(L, b + 6), t1[1]);
               a = sk_mlabb(a, RS_RING(L, b + 7), t1[0]);
               v = sk_sat16(sk_rround(a, 15));
            } else if (ord == 18) {                                                /* resampler_private_down_FIR.c:56-86 */
               const int ph = sk_mulwb(idx & 0xFFFF, c.FIR_Fracs);
               const i16 *c0 = &F[9 * ph], *c1 = &F[9 * (c.FIR_Fracs - 1 - ph)];
               i32 a = sk_mulwb(RS_RING(L, b), c0[0]);
               for (int j = 1; j < 9; j++) a = sk_mlawb(a, RS_RING(L, b + j), c0[j]);
               for (int j = 0; j < 9; j++) a = sk_mlawb(a, RS_RING(L, b + 17 - j), c1[j]);
               v = sk_sat16(sk_rround(a, 6));
            } else {                                                               /* :88-141, symmetric 24/36 taps */
               i32 a = sk_mulwb(RS_RING(L, b) + RS_RING(L, b + ord - 1), F[0]);
               for (int j = 1; j < ord / 2; j++) a = sk_mlawb(a, RS_RING(L, b + j) + RS_RING(L, b + ord - 1 - j), F[j]);
               v = sk_sat16(sk_rround(a, 6));
            }
            out[no++] = (i16)v;
            idx += c.invRatio_Q16;
         }
      }
      r.rb = (r.rb + (nIn << up)) & (OA_RS_RING - 1);                              /* the tail becomes the head of the next batch */
      done += nIn;
   }
   return no;
}


/* One call of silk_resampler for this lane's channel.  st = &state[channel] (row stride n), in/out = the channel's buffers. */
template <class RL, class StP, class InP, class Out> WV_DEV void silk_resampler_lane(const OaResamplerCfg c, WV_LDS RL *L, StP st, int n, InP in, int inLen, Out out, const int col)
{
   RsLane r; r.rb = 0;
   for (int j = 0; j < 6; j++) r.iir[j] = st[(OA_RS_ROW_IIR + j) * n];
   const int ord = c.resampler_function == OA_RS_FN_DOWN_FIR ? c.FIR_Order : c.resampler_function == OA_RS_FN_IIR_FIR ? 8 : 0;
   for (int j = 0; j < ord; j++) RS_RING(L, j) = st[(OA_RS_ROW_FIR + j) * n];
   const int nNew = c.Fs_in_kHz - c.inputDelay;
   RsDelaySrc<StP, InP> s1 = { st + OA_RS_ROW_DELAY * n, c.inputDelay, n, in };
   int w = rs_segment(c, L, r, s1, c.Fs_in_kHz, out, col);
   RsPlainSrc<InP> s2 = { in + nNew };
   rs_segment(c, L, r, s2, inLen - c.Fs_in_kHz, out + w, col);
   for (int j = 0; j < 6; j++) st[(OA_RS_ROW_IIR + j) * n] = r.iir[j];
   for (int j = 0; j < ord; j++) st[(OA_RS_ROW_FIR + j) * n] = RS_RING(L, r.rb + j);
   for (int j = 0; j < c.inputDelay; j++) st[(OA_RS_ROW_DELAY + j) * n] = in[inLen - c.inputDelay + j];
}
#endif

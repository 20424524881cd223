/* silk_nsq.h — the SILK noise-shaping quantiser without delayed decision as a lane-per-stream kernel body.
 *
 * What it computes: silk_NSQ_c (silk/NSQ.c:76-181), i.e. per 5 ms subframe the optional LTP-state re-whitening
 * (silk_LPC_analysis_filter, silk/LPC_analysis_filter.c:49), the state rescaling (silk_nsq_scale_states, NSQ.c:368) and the
 * per-sample closed loop (silk_noise_shape_quantizer, NSQ.c:183; taps NSQ.h:35/:67): dither, LPC-16 + LTP-5 prediction, AR-24 +
 * low-frequency + harmonic noise-shaping feedback, two candidate levels and their rate-distortion cost.
 *
 * Why lane-per-stream: the loop is a 320-step recurrence per stream with ~50 multiply-accumulates per step and nothing to share
 * between streams, so the natural SIMT mapping is 64 independent recurrences per wave, no cross-lane traffic at all.  The filter
 * memories (16 + 24 words), the coefficient sets (16 + 24 + 5, pre-shifted so that SMLAWB is one v_mul_hi_i32 + add) and the
 * sliding 5/3-tap windows over the lagged histories live in VGPRs; the lag-addressed histories themselves are rows of the
 * tile-SoA state (silk_frame.h), one dword per lane per sample.  Per sample a lane issues ~300 VALU ops against 2 history loads
 * and 4 stores: the kernel is VALU-issue bound, not HBM bound. */
#ifndef OPUS_AMD_SILK_NSQ_H
#define OPUS_AMD_SILK_NSQ_H
#include "silk_frame.h"

/* ---- SILK fixed-point primitives (silk/macros.h:40-122, silk/SigProc_FIX.h:447-584, OPUS_FAST_INT64 forms) ---- */
/* 32 x 16 -> top 32 of 48 bits.  a * b16 >> 16 == a * (b16 << 16) >> 32: one v_mul_hi_i32 (the shift of a loop-invariant coefficient hoists) instead of the
 * mul_lo + mul_hi + alignbit a 64-bit product costs */
WV_DEV i32 sk_mulhi(i32 a, i32 b) { return (i32)(((i64)a * (i64)b) >> 32); }
WV_DEV i32 sk_mulwb(i32 a, i32 b) { return sk_mulhi(a, (i32)((u32)b << 16)); }                 /* silk_SMULWB */
WV_DEV i32 sk_mlawb(i32 c, i32 a, i32 b) { return add32(c, sk_mulwb(a, b)); }          /* silk_SMLAWB */
WV_DEV i32 sk_mulwt(i32 a, i32 b) { return sk_mulhi(a, (i32)((u32)b & 0xffff0000u)); }        /* silk_SMULWT */
WV_DEV i32 sk_mlawt(i32 c, i32 a, i32 b) { return add32(c, sk_mulwt(a, b)); }
WV_DEV i32 sk_mulww(i32 a, i32 b) { return (i32)(((i64)a * (i64)b) >> 16); }           /* silk_SMULWW */
WV_DEV i32 sk_mlaww(i32 c, i32 a, i32 b) { return add32(c, sk_mulww(a, b)); }
WV_DEV i32 sk_mulbb(i32 a, i32 b) { return (i32)(i16)a * (i32)(i16)b; }                /* silk_SMULBB */
WV_DEV i32 sk_mlabb(i32 c, i32 a, i32 b) { return add32(c, sk_mulbb(a, b)); }
/* "W x pre-shifted B": c + ((a * b16) >> 16) with bs = b16 << 16 prepared once per subframe */
WV_DEV i32 sk_mlaws(i32 c, i32 a, i32 bs) { return add32(c, sk_mulhi(a, bs)); }
WV_DEV i32 sk_add_sat(i32 a, i32 b) { i64 r = (i64)a + b; return r > 2147483647 ? 2147483647 : r < -2147483647 - 1 ? (i32)(-2147483647 - 1) : (i32)r; }
WV_DEV i32 sk_sub_sat(i32 a, i32 b) { i64 r = (i64)a - b; return r > 2147483647 ? 2147483647 : r < -2147483647 - 1 ? (i32)(-2147483647 - 1) : (i32)r; }
WV_DEV i32 sk_rround(i32 a, int s) { return s == 1 ? (a >> 1) + (a & 1) : ((a >> (s - 1)) + 1) >> 1; }    /* silk_RSHIFT_ROUND */
WV_DEV i32 sk_sat16(i32 a) { return a > 32767 ? 32767 : a < -32768 ? -32768 : a; }
WV_DEV i32 sk_rand(i32 seed) { return (i32)(907633515u + (u32)seed * 196314165u); }                      /* silk_RAND */
WV_DEV int sk_clz(i32 x)
{
   u32 v = (u32)x; if (!v) return 32;
   int n = 0;
   if (!(v & 0xFFFF0000u)) { n += 16; v <<= 16; }
   if (!(v & 0xFF000000u)) { n += 8; v <<= 8; }
   if (!(v & 0xF0000000u)) { n += 4; v <<= 4; }
   if (!(v & 0xC0000000u)) { n += 2; v <<= 2; }
   if (!(v & 0x80000000u)) n += 1;
   return n;
}
WV_DEV i32 sk_shl_sat(i32 a, int s)
{
   i32 lo = (i32)(-2147483647 - 1) >> s, hi = 2147483647 >> s;
   return shl32(a > hi ? hi : a < lo ? lo : a, s);
}
WV_DEV i32 sk_div32_varQ(i32 a32, i32 b32, int Qres)                 /* silk/Inlines.h:93 */
{
   int ha = sk_clz(a32 > 0 ? a32 : neg32(a32)) - 1, hb = sk_clz(b32 > 0 ? b32 : neg32(b32)) - 1;
   i32 an = shl32(a32, ha), bn = shl32(b32, hb);
   i32 binv = (2147483647 >> 2) / (bn >> 16);
   i32 r = sk_mulwb(an, binv);
   an = sub32(an, shl32(sk_mulhi(bn, r), 3));
   r = sk_mlawb(r, an, binv);
   int ls = 29 + ha - hb - Qres;
   if (ls < 0) return sk_shl_sat(r, -ls);
   return ls < 32 ? r >> ls : 0;
}
WV_DEV i32 sk_inverse32_varQ(i32 b32, int Qres)                      /* silk/Inlines.h:143 */
{
   int hb = sk_clz(b32 > 0 ? b32 : neg32(b32)) - 1;
   i32 bn = shl32(b32, hb);
   i32 binv = (2147483647 >> 2) / (bn >> 16);
   i32 r = shl32(binv, 16);
   i32 err = shl32(((i32)1 << 29) - sk_mulwb(bn, binv), 3);
   r = sk_mlaww(r, err, binv);
   int ls = 61 - hb - Qres;
   if (ls <= 0) return sk_shl_sat(r, -ls);
   return ls < 32 ? r >> ls : 0;
}

/* ---- per-lane view of one stream's slice of its tile ---- */
struct NsqMem {
   i32 *shp;        /* sLTP_shp_Q14 ring   [rows][T] */
   i32 *q15;        /* sLTP_Q15 scratch    [rows][T] (call-local, linear) */
   i32 *scal;       /* scalar block        [48][T] */
   i16 *xq;         /* xq ring             [rows][T] */
   i16 *wh;         /* re-whitened history [rows][T] (call-local, linear) */
   i32 T, len, base;
};
WV_DEV int nm_row(const NsqMem &m, int p) { int r = p + m.base; return (r >= m.len ? r - m.len : r) * m.T; }

WV_DEV NsqMem nsq_mem(i32 *tile, int T, int t, int len)
{
   NsqMem m;
   const int R = OA_SILK_HIST_ROWS;
   m.shp = tile + t;
   m.q15 = tile + R * T + t;
   m.scal = tile + 2 * R * T + t;
   m.xq = (i16 *)(tile + 2 * R * T + OA_NSQ_S_ROWS * T) + t;
   m.wh = (i16 *)(tile + 2 * R * T + OA_NSQ_S_ROWS * T + R * T / 2) + t;
   m.T = T; m.len = len; m.base = m.scal[OA_NSQ_S_BASE * T];
   return m;
}

WV_TABLE i16 k_silk_quant_offsets_Q10[4] = { 100, 240, 32, 100 };     /* silk/tables_other.c:77 [signalType>>1][quantOffsetType] */

/* Re-whitening: wh[start+P .. end) = LPC analysis residual of xq[xq0 + start ..) with the subframe's predictor; the first P
 * outputs are zero (silk/LPC_analysis_filter.c:49-108).  Per lane: its own start (lag-dependent), a P+1-sample register window. */
WV_DEV void nsq_rewhiten_lane(const NsqMem &m, int start, int end, int xq0, const i16 *A_Q12, int P)
{
   i32 a[16], w[16];
   for (int j = 0; j < 16; j++) { a[j] = j < P ? A_Q12[j] : 0; w[j] = 0; }
   for (int j = 0; j < P; j++) { w[P - 1 - j] = m.xq[nm_row(m, xq0 + start + j)]; m.wh[(start + j) * m.T] = 0; }     /* w[j] = in[n-1-j] */
   for (int n = start + P; n < end; n++) {
      i32 x = m.xq[nm_row(m, xq0 + n)];
      i32 pred = 0;
      for (int j = 0; j < 16; j++) pred = add32(pred, w[j] * a[j]);
      i32 e = sub32(shl32(x, 12), pred);
      m.wh[n * m.T] = (i16)sk_sat16(sk_rround(e, 12));
      for (int j = 15; j > 0; j--) w[j] = w[j - 1];
      w[0] = x;
   }
}

/* the two candidate levels around r_Q10 and their rate terms (NSQ.c:279-316 == NSQ_del_dec.c:441-478) */
WV_DEV void nsq_levels(i32 r_Q10, int offset_Q10, int Lambda_Q10, i32 &q1_Q10, i32 &q2_Q10, i32 &rd1, i32 &rd2)
{
   q1_Q10 = r_Q10 - offset_Q10;
   i32 q1_Q0 = q1_Q10 >> 10;
   if (Lambda_Q10 > 2048) {
      int rdo_offset = Lambda_Q10 / 2 - 512;
      if (q1_Q10 > rdo_offset) q1_Q0 = (q1_Q10 - rdo_offset) >> 10;
      else if (q1_Q10 < -rdo_offset) q1_Q0 = (q1_Q10 + rdo_offset) >> 10;
      else q1_Q0 = q1_Q10 < 0 ? -1 : 0;
   }
   if (q1_Q0 > 0) {
      q1_Q10 = (q1_Q0 << 10) - 80 + offset_Q10;  q2_Q10 = q1_Q10 + 1024;
      rd1 = sk_mulbb(q1_Q10, Lambda_Q10);  rd2 = sk_mulbb(q2_Q10, Lambda_Q10);
   } else if (q1_Q0 == 0) {
      q1_Q10 = offset_Q10;  q2_Q10 = q1_Q10 + (1024 - 80);
      rd1 = sk_mulbb(q1_Q10, Lambda_Q10);  rd2 = sk_mulbb(q2_Q10, Lambda_Q10);
   } else if (q1_Q0 == -1) {
      q2_Q10 = offset_Q10;  q1_Q10 = q2_Q10 - (1024 - 80);
      rd1 = sk_mulbb(-q1_Q10, Lambda_Q10);  rd2 = sk_mulbb(q2_Q10, Lambda_Q10);
   } else {
      q1_Q10 = shl32(q1_Q0, 10) + 80 + offset_Q10;  q2_Q10 = q1_Q10 + 1024;
      rd1 = sk_mulbb(-q1_Q10, Lambda_Q10);  rd2 = sk_mulbb(-q2_Q10, Lambda_Q10);
   }
   i32 rr = r_Q10 - q1_Q10;  rd1 = sk_mlabb(rd1, rr, rr);
   rr = r_Q10 - q2_Q10;      rd2 = sk_mlabb(rd2, rr, rr);
}

/* One frame of one stream on one lane.  `fr`, `x16`, `pulses` point at this lane's stream (AoS, exactly the caller's arrays). */
template <int SS> WV_DEV void silk_nsq_lane(const OaNsqCfg cfg, NsqMem m, const OaNsqFrame *fr, const i16 *x16, i8 *pulses, bool store)
{
   const int T = m.T, L = 5 * cfg.fs_kHz, mem = 20 * cfg.fs_kHz, frame = cfg.nb_subfr * L, P = cfg.predictLPCOrder, S = SS ? SS : cfg.shapingLPCOrder;
   i32 s[16], ar2[24];                                 /* s[j] = sLPC at lag j; ar2 = AR-shaping delay line */
   for (int j = 0; j < 16; j++) s[j] = m.scal[(OA_NSQ_S_LPC + 15 - j) * T];
   for (int j = 0; j < 24; j++) ar2[j] = m.scal[(OA_NSQ_S_AR2 + j) * T];
   i32 sLF_AR = m.scal[OA_NSQ_S_LF_AR * T], sDiff = m.scal[OA_NSQ_S_DIFF * T];
   i32 prev_gain = m.scal[OA_NSQ_S_PREVGAIN * T];
   int lag = m.scal[OA_NSQ_S_LAGPREV * T];
   const int signalType = fr->signalType;
   const bool voiced = signalType == OA_SILK_TYPE_VOICED;
   const int offset_Q10 = k_silk_quant_offsets_Q10[(signalType >> 1) * 2 + fr->quantOffsetType];
   const int interp = fr->NLSFInterpCoef_Q2 == 4 ? 0 : 1;
   const int Lambda_Q10 = fr->Lambda_Q10;
   i32 seed = fr->Seed;
   int shp_idx = mem, ltp_idx = mem;

   for (int k = 0; k < cfg.nb_subfr; k++) {
      const i16 *A_Q12 = &fr->PredCoef_Q12[((k >> 1) | (1 - interp)) * 16];
      i32 a[16], ar[24], b[5];
      for (int j = 0; j < 16; j++) a[j] = j < P ? shl32(A_Q12[j], 16) : 0;
      for (int j = 0; j < 24; j++) ar[j] = j < S ? shl32(fr->AR_Q13[k * 24 + j], 16) : 0;
      for (int j = 0; j < 5; j++) b[j] = shl32(fr->LTPCoef_Q14[k * 5 + j], 16);
      const i32 hg = fr->HarmShapeGain_Q14[k];
      const i32 harm = (hg >> 2) | shl32(hg >> 1, 16);
      const i32 Tilt_Q14 = fr->Tilt_Q14[k], LF_shp_Q14 = fr->LF_shp_Q14[k], Gain_Q16 = fr->Gains_Q16[k];
      const i32 Gain_Q10 = Gain_Q16 >> 6;
      bool rewhite = false;
      if (voiced) {
         lag = fr->pitchL[k];
         if ((k & (3 - (interp << 1))) == 0) {
            int start = mem - lag - P - OA_SILK_LTP_ORDER / 2;
            nsq_rewhiten_lane(m, start, mem, k * L, A_Q12, P);
            rewhite = true;
            ltp_idx = mem;
         }
      }
      /* ---- silk_nsq_scale_states (NSQ.c:368) ---- */
      i32 inv_gain_Q31 = sk_inverse32_varQ(Gain_Q16 > 1 ? Gain_Q16 : 1, 47);
      const i32 inv_gain_Q26 = sk_rround(inv_gain_Q31, 5);
      if (rewhite) {
         if (k == 0) inv_gain_Q31 = shl32(sk_mulwb(inv_gain_Q31, fr->LTP_scale_Q14), 2);
         for (int i = ltp_idx - lag - OA_SILK_LTP_ORDER / 2; i < ltp_idx; i++) m.q15[i * T] = sk_mulwb(inv_gain_Q31, m.wh[i * T]);
      }
      if (Gain_Q16 != prev_gain) {
         const i32 adj = sk_div32_varQ(prev_gain, Gain_Q16, 16);
         for (int i = shp_idx - mem; i < shp_idx; i++) { int r = nm_row(m, i); m.shp[r] = sk_mulww(adj, m.shp[r]); }
         if (voiced && !rewhite)
            for (int i = ltp_idx - lag - OA_SILK_LTP_ORDER / 2; i < ltp_idx; i++) m.q15[i * T] = sk_mulww(adj, m.q15[i * T]);
         sLF_AR = sk_mulww(adj, sLF_AR);
         sDiff = sk_mulww(adj, sDiff);
         for (int j = 0; j < 16; j++) s[j] = sk_mulww(adj, s[j]);
         for (int j = 0; j < 24; j++) ar2[j] = sk_mulww(adj, ar2[j]);
         prev_gain = Gain_Q16;
      }
      /* ---- silk_noise_shape_quantizer (NSQ.c:183) ---- */
      i32 pl[5] = { 0, 0, 0, 0, 0 }, sh[3] = { 0, 0, 0 };            /* sliding windows: pl[j] = pred_lag[-j], sh[j] = shp_lag[-j] */
      const int pl0 = ltp_idx - lag + OA_SILK_LTP_ORDER / 2, sh0 = shp_idx - lag + 1;
      if (voiced) for (int j = 1; j < 5; j++) pl[j - 1] = m.q15[(pl0 - j) * T];
      if (lag > 0) { sh[0] = m.shp[nm_row(m, sh0 - 1)]; sh[1] = m.shp[nm_row(m, sh0 - 2)]; }
      i32 last_shp = m.shp[nm_row(m, shp_idx - 1)];
      /* software pipeline: sample i+1's loads are issued at the top of sample i, before sample i's stores (in-order vm counter);
       * the taps of i+1 were written at i+3-lag or earlier, so this is safe for lag > 3 (else they are re-read) */
      i32 nPl = voiced ? m.q15[pl0 * T] : 0, nSh = lag > 0 ? m.shp[nm_row(m, sh0)] : 0, nX = x16[k * L];
      for (int i = 0; i < L; i++) {
         if (lag <= 3) { if (voiced) nPl = m.q15[(pl0 + i) * T]; if (lag > 0) nSh = m.shp[nm_row(m, sh0 + i)]; }
         if (voiced) { for (int j = 4; j > 0; j--) pl[j] = pl[j - 1]; pl[0] = nPl; }
         if (lag > 0) { sh[2] = sh[1]; sh[1] = sh[0]; sh[0] = nSh; }
         const i32 x_sc_Q10 = mult16_32_q16(nX, inv_gain_Q26);
         {
            const int i1 = i + 1 < L ? i + 1 : i;
            if (voiced) nPl = m.q15[(pl0 + i1) * T];
            if (lag > 0) nSh = m.shp[nm_row(m, sh0 + i1)];
            nX = x16[k * L + i1];
         }
         seed = sk_rand(seed);

         i32 LPC_pred_Q10 = P >> 1;
         for (int j = 0; j < 16; j++) LPC_pred_Q10 = sk_mlaws(LPC_pred_Q10, s[j], a[j]);

         i32 n_AR_Q12 = S >> 1;
         for (int j = 23; j > 0; j--) if (j < S) ar2[j] = ar2[j - 1];          /* taps >= S keep their (gain-scaled) contents, as in the reference */
         ar2[0] = sDiff;
         for (int j = 0; j < 24; j++) n_AR_Q12 = sk_mlaws(n_AR_Q12, ar2[j], ar[j]);
         n_AR_Q12 = shl32(n_AR_Q12, 1);
         n_AR_Q12 = sk_mlawb(n_AR_Q12, sLF_AR, Tilt_Q14);

         i32 n_LF_Q12 = sk_mulwb(last_shp, LF_shp_Q14);
         n_LF_Q12 = sk_mlawt(n_LF_Q12, sLF_AR, LF_shp_Q14);

         i32 LTP_pred_Q13 = 0;
         if (voiced) { LTP_pred_Q13 = 2; for (int j = 0; j < 5; j++) LTP_pred_Q13 = sk_mlaws(LTP_pred_Q13, pl[j], b[j]); }

         i32 t1 = sub32(shl32(LPC_pred_Q10, 2), n_AR_Q12);
         t1 = sub32(t1, n_LF_Q12);
         if (lag > 0) {
            i32 n_LTP_Q13 = sk_mulwb(sk_add_sat(sh[0], sh[2]), harm);
            n_LTP_Q13 = sk_mlawt(n_LTP_Q13, sh[1], harm);
            n_LTP_Q13 = shl32(n_LTP_Q13, 1);
            t1 = add32(sub32(LTP_pred_Q13, n_LTP_Q13), shl32(t1, 1));
            t1 = sk_rround(t1, 3);
         } else {
            t1 = sk_rround(t1, 2);
         }
         i32 r_Q10 = sub32(x_sc_Q10, t1);
         if (seed < 0) r_Q10 = neg32(r_Q10);
         r_Q10 = r_Q10 > (30 << 10) ? (30 << 10) : r_Q10 < -(31 << 10) ? -(31 << 10) : r_Q10;

         i32 q1_Q10, q2_Q10, rd1, rd2;
         nsq_levels(r_Q10, offset_Q10, Lambda_Q10, q1_Q10, q2_Q10, rd1, rd2);
         if (rd2 < rd1) q1_Q10 = q2_Q10;
         const i32 pulse = (i8)sk_rround(q1_Q10, 10);
         if (store) pulses[k * L + i] = (i8)pulse;

         i32 exc_Q14 = shl32(q1_Q10, 4);
         if (seed < 0) exc_Q14 = -exc_Q14;
         const i32 LPC_exc_Q14 = exc_Q14 + shl32(LTP_pred_Q13, 1);
         const i32 xq_Q14 = add32(LPC_exc_Q14, shl32(LPC_pred_Q10, 4));
         m.xq[nm_row(m, mem + k * L + i)] = (i16)sk_sat16(sk_rround(sk_mulww(xq_Q14, Gain_Q10), 8));

         for (int j = 15; j > 0; j--) s[j] = s[j - 1];
         s[0] = xq_Q14;
         sDiff = sub32(xq_Q14, shl32(x_sc_Q10, 4));
         sLF_AR = sub32(sDiff, shl32(n_AR_Q12, 2));
         last_shp = sub32(sLF_AR, shl32(n_LF_Q12, 2));
         m.shp[nm_row(m, shp_idx)] = last_shp;
         m.q15[ltp_idx * T] = shl32(LPC_exc_Q14, 1);
         shp_idx++; ltp_idx++;
         seed = add32(seed, pulse);
      }
   }
   for (int j = 0; j < 16; j++) m.scal[(OA_NSQ_S_LPC + 15 - j) * T] = s[j];
   for (int j = 0; j < 24; j++) m.scal[(OA_NSQ_S_AR2 + j) * T] = ar2[j];
   m.scal[OA_NSQ_S_LF_AR * T] = sLF_AR;  m.scal[OA_NSQ_S_DIFF * T] = sDiff;
   m.scal[OA_NSQ_S_PREVGAIN * T] = prev_gain;
   m.scal[OA_NSQ_S_LAGPREV * T] = fr->pitchL[cfg.nb_subfr - 1];
   m.scal[OA_NSQ_S_RANDSEED * T] = seed;
   { int nb = m.base + frame; m.scal[OA_NSQ_S_BASE * T] = nb >= m.len ? nb - m.len : nb; }
}
#endif

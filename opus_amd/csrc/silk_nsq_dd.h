/* silk_nsq_dd.h — the SILK delayed-decision noise-shaping quantiser as a quad-per-stream kernel body.
 *
 * What it computes: silk_NSQ_del_dec_c (silk/NSQ_del_dec.c:114-312; sample loop :315-644; state rescaling :646-746), the path
 * the reference encoder takes at complexity >= 2 (silk/control_codec.c:376-386): K <= 4 survivor states, each with its own
 * LPC-16 / warped-AR-24 filter memories and dither seed; every sample each survivor proposes its two nearest levels, the best K of
 * the 2K continuations survive, and the sample `decisionDelay` (<= 40) steps back is committed from the current winner's path.
 *
 * Mapping: one lane per survivor, one quad per stream, 16 streams per wave.  All per-survivor arithmetic (the bulk: 16 + 2x24
 * multiply-accumulates per sample) is plain SIMT; the K-way decisions are 4-lane exchanges inside the quad.
 *
 * No history copying: the reference memcpy's the whole survivor (1.3 KB, incl. five 40-deep rings of undecided samples) whenever a
 * survivor is replaced.  Here a ring entry never moves: lane k writes what survivor slot k produced at time t into column k of
 * row t, and each survivor carries an 80-bit ancestry word (2 bits per age: which column holds my path's entry of t-age).
 * Replacing a survivor copies the 40 filter words + the ancestry through the quad; committing sample t-D reads row t-D at the
 * winner's ancestry column.  The filter memories and coefficient sets live in VGPRs, the rings in a per-tile HBM/L2 scratch whose
 * row for the sample being committed is fetched at the top of the iteration, a full sample's arithmetic ahead of its use. */
#ifndef OPUS_AMD_SILK_NSQ_DD_H
#define OPUS_AMD_SILK_NSQ_DD_H
#include "silk_nsq.h"

enum { DD_RAND = 0, DD_Q, DD_XQ, DD_PRED, DD_SHAPE, DD_NRINGS };
#define DD_RING(ring, r, pos, lane) (ring)[(((r) * OA_SILK_DD) + (pos)) * 64 + (lane)]
#define DD_PENALTY (2147483647 >> 4)

struct DdAnc { u32 w0, w1, w2; };                               /* ages 0..15 | 16..31 | 32..39, 2 bits each */
WV_DEV int dd_anc_at(const DdAnc &a, int age) { u32 w = age < 16 ? a.w0 : age < 32 ? a.w1 : a.w2; return (int)((w >> ((age & 15) * 2)) & 3u); }
WV_DEV DdAnc dd_anc_push(const DdAnc &a, int k) { DdAnc r; r.w2 = (a.w2 << 2) | (a.w1 >> 30); r.w1 = (a.w1 << 2) | (a.w0 >> 30); r.w0 = (a.w0 << 2) | (u32)k; return r; }

/* index of the first minimum / first maximum among the first K of four values */
WV_DEV int dd_argmin4(i32 v0, i32 v1, i32 v2, i32 v3, int K)
{ int w = 0; i32 m = v0; if (K > 1 && v1 < m) { m = v1; w = 1; } if (K > 2 && v2 < m) { m = v2; w = 2; } if (K > 3 && v3 < m) { m = v3; w = 3; } return w; }
WV_DEV int dd_argmax4(i32 v0, i32 v1, i32 v2, i32 v3, int K)
{ int w = 0; i32 m = v0; if (K > 1 && v1 > m) { m = v1; w = 1; } if (K > 2 && v2 > m) { m = v2; w = 2; } if (K > 3 && v3 > m) { m = v3; w = 3; } return w; }

/* One frame of 16 streams on one wave.  fr/x16/pulses/seed_out point at this quad's stream (any address space: FR, PU are pointer types).  `store` = 0: this quad only
 * keeps the wave's collectives in step (tail tile of the batch kernel; in the encoder's quantiser kernel -- opus_sh_split.h -- a stream that sits out this pass): it reads
 * (its own tile column, someone else's parameters) and leaves no trace -- no pulse, no state word; only its lanes' private ring columns are written. */
template <int SS, class FR, class PU> WV_DEV void silk_nsq_dd_wave(const OaNsqCfg cfg, NsqMem m, i32 *ring, FR fr, const i16 *x16, PU pulses, PU seed_out, bool store)
{
   const int lane = wv_lane(), kk = lane & 3, qb = lane & ~3;
   const int T = m.T, L = 5 * cfg.fs_kHz, mem = 20 * cfg.fs_kHz, frame = cfg.nb_subfr * L, P = cfg.predictLPCOrder, S = SS ? SS : cfg.shapingLPCOrder;   /* SS != 0: shaping order known at compile time */
   const int K = cfg.nStatesDelayedDecision;
   const i32 warp_s = shl32((i16)cfg.warping_Q16, 16);

   i32 s[16], ar2[24];
   for (int j = 0; j < 16; j++) s[j] = m.scal[(OA_NSQ_S_LPC + 15 - j) * T];
   for (int j = 0; j < 24; j++) ar2[j] = m.scal[(OA_NSQ_S_AR2 + j) * T];
   i32 LF_AR = m.scal[OA_NSQ_S_LF_AR * T], Diff = m.scal[OA_NSQ_S_DIFF * T];
   i32 prev_gain = m.scal[OA_NSQ_S_PREVGAIN * T];
   int lag = store ? m.scal[OA_NSQ_S_LAGPREV * T] : 0;          /* a quad that sits out may never have had its tile column loaded: nothing read from it may become an address (lag does, through D and the history taps) */
   const int signalType = fr->signalType;
   const bool voiced = signalType == OA_SILK_TYPE_VOICED;
   const int offset_Q10 = k_silk_quant_offsets_Q10[(signalType >> 1) * 2 + fr->quantOffsetType];
   const int interp = fr->NLSFInterpCoef_Q2 == 4 ? 0 : 1;
   const int Lambda_Q10 = fr->Lambda_Q10;
   i32 Seed = (kk + fr->Seed) & 3;
   const i32 SeedInit = Seed;
   i32 SeedInitCur = SeedInit;                                 /* travels with the survivor */
   i32 RD = 0;
   i32 lastShape = m.shp[nm_row(m, mem - 1)];
   DdAnc anc; anc.w0 = anc.w1 = anc.w2 = (u32)kk * 0x55555555u;
   for (int pos = 0; pos < OA_SILK_DD; pos++) DD_RING(ring, DD_RAND, pos, lane) = 0;

   int D = OA_SILK_DD < L ? OA_SILK_DD : L;
   if (voiced) { for (int k = 0; k < cfg.nb_subfr; k++) D = imin(D, fr->pitchL[k] - OA_SILK_LTP_ORDER / 2 - 1); }
   else if (lag > 0) D = imin(D, lag - OA_SILK_LTP_ORDER / 2 - 1);

   int shp_idx = mem, ltp_idx = mem, p = 0, subfr = 0;
   i32 prevGain_Q10 = 0;
   for (int k = 0; k < cfg.nb_subfr; k++) {
      const auto A_Q12 = &fr->PredCoef_Q12[((k >> 1) | (1 - interp)) * 16];
      i32 a[16], ar[24], b[5];
      for (int j = 0; j < 16; j++) a[j] = j < P ? shl32(A_Q12[j], 16) : 0;
      for (int j = 0; j < 24; j++) ar[j] = j < S ? shl32(fr->AR_Q13[k * 24 + j], 16) : 0;
      for (int j = 0; j < 5; j++) b[j] = shl32(fr->LTPCoef_Q14[k * 5 + j], 16);
      const i32 hg = fr->HarmShapeGain_Q14[k];
      const i32 harm = (hg >> 2) | shl32(hg >> 1, 16);
      const i32 Tilt_Q14 = fr->Tilt_Q14[k], LF_shp_Q14 = fr->LF_shp_Q14[k], Gain_Q16 = fr->Gains_Q16[k];
      const i32 Gain_Q10 = Gain_Q16 >> 6;
      bool rewhite = false;
      if (voiced) lag = fr->pitchL[k];
      const bool do_rewhite = voiced && (k & (3 - (interp << 1))) == 0;

      if (k == 2) {
         /* mid-frame predictor switch (NSQ_del_dec.c:199-229): commit everything pending from the winner, demote the others */
         const i32 r0 = wv_quad_bcast<0>(RD), r1 = wv_quad_bcast<1>(RD), r2 = wv_quad_bcast<2>(RD), r3 = wv_quad_bcast<3>(RD);
         const int w = dd_argmin4(r0, r1, r2, r3, K);
         DdAnc aw; aw.w0 = (u32)wv_shfl((i32)anc.w0, qb + w); aw.w1 = (u32)wv_shfl((i32)anc.w1, qb + w); aw.w2 = (u32)wv_shfl((i32)anc.w2, qb + w);
         for (int age = OA_SILK_DD - 1; age >= 0; age--) {
            const int pos = (p + age) % OA_SILK_DD, src = qb + dd_anc_at(aw, age);
            const i32 q = wv_shfl(DD_RING(ring, DD_Q, pos, lane), src), xv = wv_shfl(DD_RING(ring, DD_XQ, pos, lane), src);
            const i32 sv = wv_shfl(DD_RING(ring, DD_SHAPE, pos, lane), src);
            if (do_rewhite && age < D) {
               if (store) pulses[k * L - 1 - age] = (i8)sk_rround(q, 10);
               if (store) { m.xq[nm_row(m, mem + k * L - 1 - age)] = (i16)sk_sat16(sk_rround(sk_mulww(xv, fr->Gains_Q16[1]), 14)); m.shp[nm_row(m, shp_idx - 1 - age)] = sv; }
            }
         }
         if (do_rewhite) { if (kk != w) RD += DD_PENALTY; subfr = 0; }
         wv_sync();
      }
      if (do_rewhite) {
         /* re-whitening, the quad splitting the output range in four contiguous pieces (FIR: outputs are independent) */
         const int start = mem - lag - P - OA_SILK_LTP_ORDER / 2, n0 = start + P, cnt = mem - n0, per = (cnt + 3) >> 2;
         if (kk == 0 && store) for (int j = 0; j < P; j++) m.wh[(start + j) * T] = 0;
         i32 aw[16], w[16];
         const int lo = n0 + kk * per, hi = imin(mem, lo + per);
         for (int j = 0; j < 16; j++) { aw[j] = j < P ? A_Q12[j] : 0; w[j] = (j < P && lo < hi) ? m.xq[nm_row(m, k * L + lo - 1 - j)] : 0; }
         for (int n = lo; n < hi; n++) {
            const i32 x = m.xq[nm_row(m, k * L + n)];
            i32 pred = 0;
            for (int j = 0; j < 16; j++) pred = add32(pred, w[j] * aw[j]);
            if (store) m.wh[n * T] = (i16)sk_sat16(sk_rround(sub32(shl32(x, 12), pred), 12));
            for (int j = 15; j > 0; j--) w[j] = w[j - 1];
            w[0] = x;
         }
         rewhite = true;
         ltp_idx = mem;
      }
      wv_sync();
      /* ---- silk_nsq_del_dec_scale_states (NSQ_del_dec.c:646) ---- */
      i32 inv_gain_Q31 = sk_inverse32_varQ(Gain_Q16 > 1 ? Gain_Q16 : 1, 47);
      const i32 inv_gain_Q26 = sk_rround(inv_gain_Q31, 5);
      if (rewhite) {
         if (k == 0) inv_gain_Q31 = shl32(sk_mulwb(inv_gain_Q31, fr->LTP_scale_Q14), 2);
         if (store) for (int i = ltp_idx - lag - OA_SILK_LTP_ORDER / 2 + kk; i < ltp_idx; i += 4) m.q15[i * T] = sk_mulwb(inv_gain_Q31, m.wh[i * T]);
      }
      if (Gain_Q16 != prev_gain) {
         const i32 adj = sk_div32_varQ(prev_gain, Gain_Q16, 16);
         if (store) for (int i = shp_idx - mem + kk; i < shp_idx; i += 4) { int r = nm_row(m, i); m.shp[r] = sk_mulww(adj, m.shp[r]); }
         if (store && voiced && !rewhite)
            for (int i = ltp_idx - lag - OA_SILK_LTP_ORDER / 2 + kk; i < ltp_idx - D; i += 4) m.q15[i * T] = sk_mulww(adj, m.q15[i * T]);
         LF_AR = sk_mulww(adj, LF_AR);
         Diff = sk_mulww(adj, Diff);
         for (int j = 0; j < 16; j++) s[j] = sk_mulww(adj, s[j]);
         for (int j = 0; j < 24; j++) ar2[j] = sk_mulww(adj, ar2[j]);
         for (int pos = 0; pos < OA_SILK_DD; pos++) {
            DD_RING(ring, DD_PRED, pos, lane) = sk_mulww(adj, DD_RING(ring, DD_PRED, pos, lane));
            DD_RING(ring, DD_SHAPE, pos, lane) = sk_mulww(adj, DD_RING(ring, DD_SHAPE, pos, lane));
         }
         lastShape = sk_mulww(adj, lastShape);
         prev_gain = Gain_Q16;
      }
      wv_sync();
      /* ---- silk_noise_shape_quantizer_del_dec (NSQ_del_dec.c:315) ---- */
      i32 pl[5] = { 0, 0, 0, 0, 0 }, sh[3] = { 0, 0, 0 };
      const int pl0 = ltp_idx - lag + OA_SILK_LTP_ORDER / 2, sh0 = shp_idx - lag + 1;
      if (voiced) for (int j = 1; j < 5; j++) pl[j - 1] = m.q15[(pl0 - j) * T];
      if (lag > 0) { sh[0] = m.shp[nm_row(m, sh0 - 1)]; sh[1] = m.shp[nm_row(m, sh0 - 2)]; }
      /* Software pipeline: everything sample i+1 reads from memory is requested at the top of sample i, i.e. BEFORE sample i's stores
       * in program order (the vm counter is in-order, so a load issued after a store cannot be waited on without also waiting for the
       * store's write acknowledge).  Hazards: the ring row committed at i+1 was written at i+1-D (needs D >= 2, else re-read); the
       * LTP tap for i+1 is the entry committed at i exactly when D == lag-3 (then it is forwarded from the register, vPred). */
      int pn = p == 0 ? OA_SILK_DD - 1 : p - 1, lastn = (pn + D) % OA_SILK_DD;
      i32 nRand = DD_RING(ring, DD_RAND, lastn, lane), nQ = DD_RING(ring, DD_Q, lastn, lane), nXq = DD_RING(ring, DD_XQ, lastn, lane);
      i32 nPred = DD_RING(ring, DD_PRED, lastn, lane), nShape = DD_RING(ring, DD_SHAPE, lastn, lane);
      i32 nPl = voiced ? m.q15[pl0 * T] : 0, nSh = lag > 0 ? m.shp[nm_row(m, sh0)] : 0, nX = x16[k * L];
      for (int i = 0; i < L; i++) {
         p = pn;
         const int last = lastn;
         if (D < 2) {                                                /* (never with the encoder's pitch range; kept for imported states) */
            nRand = DD_RING(ring, DD_RAND, last, lane); nQ = DD_RING(ring, DD_Q, last, lane); nXq = DD_RING(ring, DD_XQ, last, lane);
            nPred = DD_RING(ring, DD_PRED, last, lane); nShape = DD_RING(ring, DD_SHAPE, last, lane);
         }
         const i32 rRand = nRand, rQ = nQ, rXq = nXq, rPred = nPred, rShape = nShape;
         if (voiced) { for (int j = 4; j > 0; j--) pl[j] = pl[j - 1]; pl[0] = nPl; }
         if (lag > 0) { sh[2] = sh[1]; sh[1] = sh[0]; sh[0] = nSh; }
         const i32 x_Q10 = mult16_32_q16(nX, inv_gain_Q26);
         {
            const int i1 = i + 1 < L ? i + 1 : i;
            pn = p == 0 ? OA_SILK_DD - 1 : p - 1; lastn = (pn + D) % OA_SILK_DD;
            nRand = DD_RING(ring, DD_RAND, lastn, lane); nQ = DD_RING(ring, DD_Q, lastn, lane); nXq = DD_RING(ring, DD_XQ, lastn, lane);
            nPred = DD_RING(ring, DD_PRED, lastn, lane); nShape = DD_RING(ring, DD_SHAPE, lastn, lane);
            if (voiced) nPl = m.q15[(pl0 + i1) * T];
            if (lag > 0) nSh = m.shp[nm_row(m, sh0 + i1)];
            nX = x16[k * L + i1];
         }

         /* per-survivor part */
         Seed = sk_rand(Seed);
         i32 LPC_pred_Q14 = P >> 1;
         for (int j = 0; j < 16; j++) LPC_pred_Q14 = sk_mlaws(LPC_pred_Q14, s[j], a[j]);
         LPC_pred_Q14 = shl32(LPC_pred_Q14, 4);

         i32 n_AR_Q14 = S >> 1;
         {
            i32 in = sk_mlaws(Diff, ar2[0], warp_s);
            for (int j = 0; j < 24; j++) if (j < S) {
               const i32 out = j + 1 < S ? sk_mlaws(ar2[j], sub32(ar2[j + 1 < 24 ? j + 1 : 23], in), warp_s) : 0;
               ar2[j] = in;
               n_AR_Q14 = sk_mlaws(n_AR_Q14, in, ar[j]);
               in = out;
            }
         }
         n_AR_Q14 = shl32(n_AR_Q14, 1);
         n_AR_Q14 = sk_mlawb(n_AR_Q14, LF_AR, Tilt_Q14);
         n_AR_Q14 = shl32(n_AR_Q14, 2);
         i32 n_LF_Q14 = sk_mulwb(lastShape, LF_shp_Q14);
         n_LF_Q14 = sk_mlawt(n_LF_Q14, LF_AR, LF_shp_Q14);
         n_LF_Q14 = shl32(n_LF_Q14, 2);

         /* common part (needs the history taps) */
         i32 LTP_pred_Q14 = 0, n_LTP_Q14 = 0;
         if (voiced) { LTP_pred_Q14 = 2; for (int j = 0; j < 5; j++) LTP_pred_Q14 = sk_mlaws(LTP_pred_Q14, pl[j], b[j]); LTP_pred_Q14 = shl32(LTP_pred_Q14, 1); }
         if (lag > 0) {
            n_LTP_Q14 = sk_mulwb(sk_add_sat(sh[0], sh[2]), harm);
            n_LTP_Q14 = sk_mlawt(n_LTP_Q14, sh[1], harm);
            n_LTP_Q14 = LTP_pred_Q14 - shl32(n_LTP_Q14, 2);
         }

         i32 t1 = sk_add_sat(n_AR_Q14, n_LF_Q14);
         const i32 t2 = add32(n_LTP_Q14, LPC_pred_Q14);
         t1 = sk_rround(sk_sub_sat(t2, t1), 4);
         i32 r_Q10 = x_Q10 - t1;
         if (Seed < 0) r_Q10 = neg32(r_Q10);
         r_Q10 = r_Q10 > (30 << 10) ? (30 << 10) : r_Q10 < -(31 << 10) ? -(31 << 10) : r_Q10;

         i32 q1_Q10, q2_Q10, rd1, rd2;
         nsq_levels(r_Q10, offset_Q10, Lambda_Q10, q1_Q10, q2_Q10, rd1, rd2);
         rd1 >>= 10; rd2 >>= 10;
         const bool first_is_q1 = rd1 < rd2;
         i32 cQ[2], cRD[2], cXq[2], cLF[2], cDiff[2], cShp[2], cExc[2];
         cQ[0] = first_is_q1 ? q1_Q10 : q2_Q10;  cRD[0] = RD + (first_is_q1 ? rd1 : rd2);
         cQ[1] = first_is_q1 ? q2_Q10 : q1_Q10;  cRD[1] = RD + (first_is_q1 ? rd2 : rd1);
         for (int c = 0; c < 2; c++) {
            i32 exc_Q14 = shl32(cQ[c], 4);
            if (Seed < 0) exc_Q14 = -exc_Q14;
            cExc[c] = exc_Q14 + LTP_pred_Q14;
            cXq[c] = add32(cExc[c], LPC_pred_Q14);
            cDiff[c] = sub32(cXq[c], shl32(x_Q10, 4));
            cLF[c] = sub32(cDiff[c], n_AR_Q14);
            cShp[c] = sk_sub_sat(cLF[c], n_LF_Q14);
         }

         /* ---- K-way decisions inside the quad ---- */
         i32 g0 = wv_quad_bcast<0>(cRD[0]), g1 = wv_quad_bcast<1>(cRD[0]), g2 = wv_quad_bcast<2>(cRD[0]), g3 = wv_quad_bcast<3>(cRD[0]);
         const int winner = dd_argmin4(g0, g1, g2, g3, K);
         const int mysrc = qb + dd_anc_at(anc, D - 1);                    /* column holding my path's entry of the sample being committed */
         const i32 myrand = wv_shfl(rRand, mysrc);
         const i32 wrand = wv_shfl(myrand, qb + winner);
         if (myrand != wrand) { cRD[0] += DD_PENALTY; cRD[1] += DD_PENALTY; }
         g0 = wv_quad_bcast<0>(cRD[0]); g1 = wv_quad_bcast<1>(cRD[0]); g2 = wv_quad_bcast<2>(cRD[0]); g3 = wv_quad_bcast<3>(cRD[0]);
         const i32 h0 = wv_quad_bcast<0>(cRD[1]), h1 = wv_quad_bcast<1>(cRD[1]), h2 = wv_quad_bcast<2>(cRD[1]), h3 = wv_quad_bcast<3>(cRD[1]);
         const int worst = dd_argmax4(g0, g1, g2, g3, K), best2 = dd_argmin4(h0, h1, h2, h3, K);
         const i32 rdmax = worst == 0 ? g0 : worst == 1 ? g1 : worst == 2 ? g2 : g3, rdmin2 = best2 == 0 ? h0 : best2 == 1 ? h1 : best2 == 2 ? h2 : h3;
         const bool take = rdmin2 < rdmax && kk == worst;                 /* this lane's survivor is replaced by best2's second choice */
         const int from = take ? qb + best2 : lane;

         /* commit sample i-D from the winner's path (the winner is never the replaced survivor) */
         const int wsrc = wv_shfl(mysrc, qb + winner);
         const i32 vQ = wv_shfl(rQ, wsrc), vXq = wv_shfl(rXq, wsrc), vPred = wv_shfl(rPred, wsrc), vShape = wv_shfl(rShape, wsrc);
         if (subfr > 0 || i >= D) {
            if (store) {
               pulses[k * L + i - D] = (i8)sk_rround(vQ, 10);
               m.xq[nm_row(m, mem + k * L + i - D)] = (i16)sk_sat16(sk_rround(sk_mulww(vXq, i >= D ? Gain_Q10 : prevGain_Q10), 8));
               m.shp[nm_row(m, shp_idx - D)] = vShape;
               m.q15[(ltp_idx - D) * T] = vPred;
            }
            if (D == lag - OA_SILK_LTP_ORDER / 2 - 1) nPl = vPred;           /* the tap sample i+1 needs is the one just committed */
         }
         shp_idx++; ltp_idx++;

         /* survivor replacement: filter memories, seed and ancestry through the quad; the candidate becomes best2's second */
         for (int j = 0; j < 16; j++) s[j] = wv_shfl(s[j], from);
         for (int j = 0; j < 24; j++) ar2[j] = wv_shfl(ar2[j], from);
         Seed = wv_shfl(Seed, from);  SeedInitCur = wv_shfl(SeedInitCur, from);
         anc.w0 = (u32)wv_shfl((i32)anc.w0, from); anc.w1 = (u32)wv_shfl((i32)anc.w1, from); anc.w2 = (u32)wv_shfl((i32)anc.w2, from);
         const i32 uQ = wv_shfl(cQ[1], from), uRD = wv_shfl(cRD[1], from), uXq = wv_shfl(cXq[1], from), uLF = wv_shfl(cLF[1], from);
         const i32 uDiff = wv_shfl(cDiff[1], from), uShp = wv_shfl(cShp[1], from), uExc = wv_shfl(cExc[1], from);
         const i32 nQ = take ? uQ : cQ[0], nXq = take ? uXq : cXq[0], nShp = take ? uShp : cShp[0], nExc = take ? uExc : cExc[0];
         RD = take ? uRD : cRD[0];  LF_AR = take ? uLF : cLF[0];  Diff = take ? uDiff : cDiff[0];

         /* update (NSQ_del_dec.c:620-634) */
         for (int j = 15; j > 0; j--) s[j] = s[j - 1];
         s[0] = nXq;
         DD_RING(ring, DD_XQ, p, lane) = nXq;
         DD_RING(ring, DD_Q, p, lane) = nQ;
         DD_RING(ring, DD_PRED, p, lane) = shl32(nExc, 1);
         DD_RING(ring, DD_SHAPE, p, lane) = nShp;
         lastShape = nShp;
         Seed = add32(Seed, sk_rround(nQ, 10));
         DD_RING(ring, DD_RAND, p, lane) = Seed;
         anc = dd_anc_push(anc, kk);
      }
      prevGain_Q10 = Gain_Q10;
      subfr++;
   }
   /* final flush from the winner (NSQ_del_dec.c:275-306) */
   {
      const i32 r0 = wv_quad_bcast<0>(RD), r1 = wv_quad_bcast<1>(RD), r2 = wv_quad_bcast<2>(RD), r3 = wv_quad_bcast<3>(RD);
      const int w = dd_argmin4(r0, r1, r2, r3, K), from = qb + w;
      DdAnc aw; aw.w0 = (u32)wv_shfl((i32)anc.w0, from); aw.w1 = (u32)wv_shfl((i32)anc.w1, from); aw.w2 = (u32)wv_shfl((i32)anc.w2, from);
      const i32 Gain_Q10 = fr->Gains_Q16[cfg.nb_subfr - 1] >> 6;
      for (int age = OA_SILK_DD - 1; age >= 0; age--) {
         const int pos = (p + age) % OA_SILK_DD, src = qb + dd_anc_at(aw, age);
         const i32 q = wv_shfl(DD_RING(ring, DD_Q, pos, lane), src), xv = wv_shfl(DD_RING(ring, DD_XQ, pos, lane), src);
         const i32 sv = wv_shfl(DD_RING(ring, DD_SHAPE, pos, lane), src);
         if (age < D) {
            if (store) { pulses[frame - 1 - age] = (i8)sk_rround(q, 10); m.xq[nm_row(m, mem + frame - 1 - age)] = (i16)sk_sat16(sk_rround(sk_mulww(xv, Gain_Q10), 8)); m.shp[nm_row(m, shp_idx - 1 - age)] = sv; }
         }
      }
      const i32 si = wv_shfl(SeedInitCur, from);
      if (store && kk == 0) *seed_out = (i8)si;
      for (int j = 0; j < 16; j++) { const i32 v = wv_shfl(s[j], from); if (store) m.scal[(OA_NSQ_S_LPC + 15 - j) * T] = v; }
      for (int j = 0; j < 24; j++) { const i32 v = wv_shfl(ar2[j], from); if (store) m.scal[(OA_NSQ_S_AR2 + j) * T] = v; }
      const i32 lf = wv_shfl(LF_AR, from), df = wv_shfl(Diff, from);
      if (store) {
         m.scal[OA_NSQ_S_LF_AR * T] = lf;  m.scal[OA_NSQ_S_DIFF * T] = df;
         m.scal[OA_NSQ_S_PREVGAIN * T] = prev_gain;
         m.scal[OA_NSQ_S_LAGPREV * T] = fr->pitchL[cfg.nb_subfr - 1];
         { int nb = m.base + frame; m.scal[OA_NSQ_S_BASE * T] = nb >= m.len ? nb - m.len : nb; }
      }
   }
}
#endif

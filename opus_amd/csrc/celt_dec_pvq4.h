/* celt_dec_pvq4.h — the decoder's PVQ stage (quant_all_bands with encode = 0) with FOUR streams per wavefront: one 16-lane group (one DPP row) per stream.
 * Reference: celt/bands.c :638 compute_qn, :700 compute_theta (decode branches :802-:840), :973 quant_partition, :1248 quant_band, :1387 quant_band_stereo,
 * :1589 quant_all_bands; celt/vq.c :621 alg_unquant, :695 renormalise_vector, :104 exp_rotation; celt/cwrs.c :467 cwrsi.
 *
 * The decoder twin of celt_enc_pvq4.h and built on it: the group's LDS (P4Group), the explicit stack of quant_partition, quant_band's pre / post transforms, the register
 * rotation, the stereo merge and the group collectives are the encoder's; what is the decoder's own is the range DEcoder on lanes 0 / 16 / 32 / 48 (bytes read from the
 * stream's continuation record in HBM), compute_theta without a signal to measure, and alg_unquant: the index -> pulse vector walk (cwrsi) as a search along one row of
 * U(n, k) by the sixteen lanes of the group per position.
 * The one-wave-per-stream form of the same stage (celt_dec_bands.h) spends a wave instruction per scalar step of the band tree; here that instruction serves four streams.
 * What is NOT here: frames shorter than 10 ms (bands of one or two coefficients) -- the fast kernel keeps those. */
#ifndef OPUS_AMD_CELT_DEC_PVQ4_H
#define OPUS_AMD_CELT_DEC_PVQ4_H

/* what oa_decode_fast_kernel / oa_decode_hyb_kernel hand over when they stop a frame in front of its bands, and oa_celt_dback_kernel takes up again (one record per stream, HBM) */
struct alignas(16) CeltDecCont {
   i32 hdr[4];                                       /* from oa_celt_dback_kernel to oa_celt_deemph_kernel: [0] samples per channel of the frame (0: nothing to do), [1] flags (celt_dec_frame.h: celt_decode_frame_tail) */
   i32 image[(offsetof(DecLds, BC) + 3) / 4];        /* the front wave's LDS up to the phase scratch: coder, frame constants, band arrays, the frame's bytes */
   alignas(16) i32 xg[2 * OA_MAX_FRAME + 2 * OA_NORM_LEN];       /* the spectrum X[2][N] of the frame in flight, then the folding memory norm[2][OA_NORM_LEN] (the layout of the per-wave scratch, celt_dec_lds.h) */
};
#define OA_DEC_CUT (-1000)                           /* celt_decode_frame_wave: stopped in front of the bands (never leaves the kernels) */

#define P4D_EC_BEGIN EcCtx ec_; ec_ld(&ec_, &G->ec); EcCtx *e = &ec_; const u8 *buf = ecbuf
#define P4D_EC_END ec_st(&G->ec, &ec_)

/* cwrsi (cwrs.c:467) on a group: index -> y[0 .. N) (plain integers, one per word of y), returns sum y^2.  N, K, idx are group-uniform.  At dimension n with k pulses left
 * and running index i: the sign is negative iff i >= U(n, k + 1); the pulses that remain AFTER this position are the largest k' <= k with U(n, k') <= i.  U(n, .) is
 * non-decreasing and a position rarely takes more than a few pulses, so the sixteen lanes hold a WINDOW of the row -- lane l: U(n, ka + 1 - l), anchored at the pulse count ka
 * the walk had one position earlier -- and the first lane at or below the current k whose entry fits is the answer.  Anchoring the window one position back is what takes the
 * table read off the chain: the window of dimension n - 1 is requested when the walk arrives at dimension n (k only shrinks, by the d pulses this position takes: the next
 * position finds its U(n - 1, k + 1) in lane d and its candidates behind it).  Only a position that takes 15 pulses or more reads the table on the spot.
 * W0: the first window, U(N, K + 1 - lane) (0 where K + 1 - lane < 0), requested by the caller ahead of the index symbol. */
WV_DEV u32 p4d_cwrsi_window(int n, int ka)                    /* lane l: U(n, ka + 1 - l); no such entry (ka + 1 - l < 0): all ones, which no index reaches */
{
   const int c = ka + 1 - wg_lane();
   const u32 u = pvq_u(n, imax(c, 0));
   return c >= 0 ? u : 0xffffffffu;
}
WV_DEV i32 p4d_cwrsi(int N, int K, u32 idx, WV_LDS i32 *y, u32 W0)
{
   const int gl = wg_lane();
   int k = K, ka = K;
   u32 i = idx, W = W0;
   i32 yy = 0;
   for (int n = N; n > 2; n--) {
      const u32 Wn = p4d_cwrsi_window(n - 1, k);                                  /* in flight while this position is searched (dimension 2 is in the table too; its window goes unused) */
      const int d0 = ka - k;                                                      /* lane d0 holds U(n, k + 1), the lanes behind it U(n, k), U(n, k - 1), ... */
      u32 above = (u32)wg_bcast((i32)W, imin(d0, WG_WIDTH - 1));
      if (d0 >= WG_WIDTH) above = pvq_u(n, k + 1);
      const int neg = i >= above;
      i -= neg ? above : 0u;
      int kk; u32 below;
      const u32 m = wg_ballot(W <= i) & (0xfffeu << imin(d0, WG_WIDTH)) & 0xffffu;    /* (the lanes behind lane d0; d0 >= 16: none) */
      if (m) { const int f = __builtin_ctz(m); kk = ka + 1 - f; below = (u32)wg_bcast((i32)W, f); }
      else {
#ifdef P4_STAT
         if (gl == 0) fprintf(stderr, "P4D cwrsi: table read on the spot, n %d k %d ka %d\n", n, k, ka);
#endif
         kk = 0; below = 0;
         for (int base = imin(k, ka + 1 - WG_WIDTH); ; base -= WG_WIDTH) {
            const int cand = base - gl;
            const u32 u = cand >= 0 ? pvq_u(n, cand) : 0u;
            const u32 m2 = wg_ballot(cand >= 0 && u <= i);
            if (m2) { const int f = __builtin_ctz(m2); kk = base - f; below = (u32)wg_bcast((i32)u, f); break; }
         }
      }
      i -= below;
      const int v = neg ? kk - k : k - kk;
      y[N - n] = v;                                                               /* (every lane of the group the same word) */
      yy = mac16_16(yy, v, v);
      ka = k; k = kk; W = Wn;
   }
   if (N >= 2) {                                                                  /* dimensions 2 and 1 in closed form: U(2, k) = 2k - 1 (k > 0) */
      const u32 p = 2 * (u32)k + 1;
      const int neg = i >= p;
      if (neg) i -= p;
      const int kk = (int)((i + 1) >> 1);
      if (kk) i -= 2 * (u32)kk - 1;
      const int v = neg ? kk - k : k - kk, last = i ? -kk : kk;
      if (gl == 0) { y[N - 2] = v; y[N - 1] = last; }
      yy += v * v + last * last;
   }
   wg_sync();
   return yy;
}

/* alg_unquant (vq.c:621) of the leaf X[0 .. N) in the group's LDS: the pulse vector is decoded into X itself, scaled and rotated back in registers when the leaf fits one
 * coefficient per lane (the encoder's resynthesis, celt_enc_pvq4.h: p4_alg_quant) */
WV_DEV unsigned p4d_alg_unquant(WV_LDS P4Group *G, const u8 *ecbuf, WV_LDS i32 *X, int N, int K, int spread, int B, i32 gain)
{
   const int gl = wg_lane();
   const u32 ft = pvq_u(N, K) + pvq_u(N, K + 1);
   i32 idx = 0;
   const u32 W0 = p4d_cwrsi_window(N, K);
   P4_TIC();
   GLANE0 { P4D_EC_BEGIN; idx = (i32)k_ec_dec_uint(EC_PASS, ft); P4D_EC_END; }
   idx = wg_bcast(idx, 0);
   P4_TOC(5);
   const i32 Ryy = p4d_cwrsi(N, K, (u32)idx, X, W0);
   P4_TOC(6);
   const int k = celt_ilog2(Ryy) >> 1;
   const i32 t_ = vshr32(Ryy, 2 * (k - 7) - 15);
   const i32 g = mult32_32_q31(fx_rsqrt_norm32(t_), gain);
   const int rot = p4_rot_applies(N, K, spread);
   unsigned cm = 1;
   if (N <= WG_WIDTH) {
      const bool vld = gl < N;
      const i32 q = vld ? X[gl] : 0;
      i32 v = vld ? vshr32(mult16_32_q15(q, g), k + 15 - NORM_SHIFT) : 0;
      if (rot) { const P4Rot rr = p4_rot_setup(N, B, K, spread); v = p4_exp_rotation_reg(v, X, N, rr, -1, B); }
      if (B > 1) { const int N0 = fx_div_pow2(N, B); cm = wg_or(q != 0 ? 1u << ((u32)gl / (u32)N0) : 0u); }
      wg_sync();
      if (vld) X[gl] = v;
      wg_sync();
   } else {
      if (B > 1) {
         const int N0 = fx_div_pow2(N, B);
         u32 m = 0;
         FOR_GL(j, N) if (X[j] != 0) m |= 1u << ((u32)j / (u32)N0);
         cm = wg_or(m);
      }
      wg_sync();
      FOR_GL(j, N) { const i32 v = vshr32(mult16_32_q15(X[j], g), k + 15 - NORM_SHIFT); X[j] = rot ? pshr32(v, NORM_SHIFT - 14) : v; }
      wg_sync();
      if (rot) {
         p4_exp_rotation_q14(X, N, -1, B, K, spread);
         FOR_GL(j, N) X[j] = shl32(X[j], NORM_SHIFT - 14);
         wg_sync();
      }
   }
   P4_TOC(7);
   return cm;
}

/* compute_theta (bands.c:700), decode side */
WV_DEV P4Theta p4d_compute_theta(WV_LDS P4Group *G, const u8 *ecbuf, const P4Cfg &cfg, i32 remaining_bits, int N, int b, int B, int B0, int LM, int stereo, int fill)
{
   int qn, itheta = 0, delta, imid, iside, qalloc, pulse_cap, offset, inv = 0;
   const int i = cfg.i, intensity = cfg.intensity;
   P4_TIC();
   pulse_cap = ct_logN[i] + LM * (1 << BITRES);
   offset = (pulse_cap >> 1) - (stereo && N == 2 ? 16 : 4);
   qn = p4_compute_qn(N, b, offset, pulse_cap, stereo);
   if (stereo && i >= intensity) qn = 1;
   wg_sync();
   const i32 tell = (i32)ec_tell_frac_lds(&G->ec);
   wg_sync();
   if (qn != 1) {
      i32 it = 0;
      GLANE0 {
         P4D_EC_BEGIN;
         if (stereo && N > 2) {
            const int p0 = 3, x0 = qn / 2, ft = p0 * (x0 + 1) + x0;
            const int fs = (int)k_ec_decode(EC_PASS, (unsigned)ft);
            int x;
            if (fs < (x0 + 1) * p0) x = fs / p0;
            else x = x0 + 1 + (fs - (x0 + 1) * p0);
            k_ec_dec_update(EC_PASS, (unsigned)(x <= x0 ? p0 * x : (x - 1 - x0) + (x0 + 1) * p0), (unsigned)(x <= x0 ? p0 * (x + 1) : (x - x0) + (x0 + 1) * p0), (unsigned)ft);
            it = x;
         } else if (B0 > 1 || stereo) {
            it = (i32)k_ec_dec_uint(EC_PASS, (u32)(qn + 1));
         } else {
            int fs, fl;
            const int ft = ((qn >> 1) + 1) * ((qn >> 1) + 1);
            const int fm = (int)k_ec_decode(EC_PASS, (unsigned)ft);
            if (fm < ((qn >> 1) * ((qn >> 1) + 1) >> 1)) {
               it = (i32)((fx_isqrt32(8 * (u32)fm + 1) - 1) >> 1);
               fs = it + 1;
               fl = it * (it + 1) >> 1;
            } else {
               it = (i32)((2 * (qn + 1) - fx_isqrt32(8 * (u32)(ft - fm - 1) + 1)) >> 1);
               fs = qn + 1 - it;
               fl = ft - ((qn + 1 - it) * (qn + 2 - it) >> 1);
            }
            k_ec_dec_update(EC_PASS, (unsigned)fl, (unsigned)(fl + fs), (unsigned)ft);
         }
         P4D_EC_END;
      }
      itheta = wg_bcast(it, 0);
      itheta = (int)fx_udiv24((u32)((i32)itheta * 16384), (u32)qn);
   } else if (stereo) {
      if (b > 2 << BITRES && remaining_bits > 2 << BITRES) {
         i32 iv = 0;
         GLANE0 { P4D_EC_BEGIN; iv = k_ec_dec_bit_logp(EC_PASS, 2); P4D_EC_END; }
         inv = wg_bcast(iv, 0);
      } else inv = 0;
      if (cfg.disable_inv) inv = 0;
      itheta = 0;
   }
   wg_sync();
   qalloc = (int)ec_tell_frac_lds(&G->ec) - tell;
   b -= qalloc;
   if (itheta == 0) { imid = 32767; iside = 0; fill &= (1 << B) - 1; delta = -16384; }
   else if (itheta == 16384) { imid = 0; iside = 32767; fill &= ((1 << B) - 1) << B; delta = 16384; }
   else {
      imid = bitexact_cos((i16)itheta);
      iside = bitexact_cos((i16)(16384 - itheta));
      delta = frac_mul16((N - 1) << 7, bitexact_log2tan(iside, imid));
   }
   P4_TOC(stereo ? 1 : 3);
   P4Theta r = {inv, imid, iside, delta, itheta, qalloc, b, fill};
   return r;
}

/* quant_partition (bands.c:973), decode side, as the walk over the explicit stack of celt_enc_pvq4.h (p4_tree_run): every trip of the loop takes every group down through its
 * split nodes (the theta symbol) to its next leaf (alg_unquant, or the fill of a leaf without pulses), decodes the leaves side by side, and unwinds */
WV_DEV void p4d_tree_run(WV_LDS P4Lds *L4, WV_LDS P4Group *G, const u8 *ecbuf, const P4Cfg &cfg, P4Tree &tr)
{
   tr.depth = 0; tr.done = !tr.act; tr.cm = 0;
   P4_TIC();
   while (wv_any(!tr.done)) {
      for (;;) {
         int split = 0;
         const WV_LDS u8 *row = L4->rows + (tr.LM + 1) * 64;
         if (!tr.done) split = tr.LM != -1 && tr.b > p4_row(row, p4_row(row, 0)) + 12 && tr.N > 2;
         if (!wv_any(split)) break;
         if (split) {
            const int B0 = tr.B;
            int N = tr.N >> 1, LM = tr.LM - 1, fill = tr.fill, B;
            if (B0 == 1) fill = (fill & 1) | (fill << 1);
            B = (B0 + 1) >> 1;
            const P4Theta th = p4d_compute_theta(G, ecbuf, cfg, tr.remaining_bits, N, tr.b, B, B0, LM, 0, fill);
            int delta = th.delta;
            const int itheta = th.itheta, b = th.b;
            fill = th.fill;
            const i32 mid = shl32((i32)th.imid, 16), side = shl32((i32)th.iside, 16);
            if (B0 > 1 && (itheta & 0x3fff)) {
               if (itheta > 8192) delta -= delta >> (4 - LM);
               else delta = imin(0, delta + (N << BITRES >> (5 - LM)));
            }
            const int mbits = imax(0, imin(b, (b - delta) / 2)), sbits = b - mbits;
            tr.remaining_bits -= th.qalloc;
            const int mid_first = mbits >= sbits;
            const i32 gm = mult32_32_q31(tr.gain, mid), gs = mult32_32_q31(tr.gain, side);
            WV_LDS P4Frame *f = &G->stk[tr.depth];
            GLANE0 {
               f->xo = tr.xo; f->N = N; f->B = B; f->B0 = B0; f->LM = LM; f->lb = tr.lb; f->gm = gm; f->gs = gs; f->fill = fill; f->mbits = mbits; f->sbits = sbits;
               f->itheta = itheta; f->rebal = tr.remaining_bits; f->mid_first = mid_first; f->phase = 0; f->cm = 0;
            }
            tr.N = N; tr.B = B; tr.LM = LM;
            if (mid_first) { tr.b = mbits; tr.gain = gm; tr.fill = fill; }
            else { tr.xo += N; tr.b = sbits; tr.gain = gs; tr.fill = fill >> B; tr.lb = tr.lb >= 0 ? tr.lb + N : -1; }
            tr.depth++;
         }
      }
      P4_TOC(2);
      unsigned cm = 0;
      if (!tr.done) {
         const WV_LDS u8 *row = L4->rows + (tr.LM + 1) * 64;
         WV_LDS i32 *X = G->Xb + tr.xo;
         const int N = tr.N, B = tr.B;
         int q = p4_bits2pulses(row, tr.b);
         int curr_bits = p4_pulses2bits(row, q);
         tr.remaining_bits -= curr_bits;
         while (tr.remaining_bits < 0 && q > 0) {
            tr.remaining_bits += curr_bits;
            q--;
            curr_bits = p4_pulses2bits(row, q);
            tr.remaining_bits -= curr_bits;
         }
         P4_TOC(4);
         if (q != 0) cm = p4d_alg_unquant(G, ecbuf, X, N, k_get_pulses(q), cfg.spread, B, tr.gain);
         else {
            const unsigned cm_mask = (unsigned)(1UL << B) - 1;
            const int fill = tr.fill & (int)cm_mask;
            wg_sync();
            if (!fill) { FOR_GL(j, N) X[j] = 0; wg_sync(); }
            else {
               const WV_LDS i32 *lowband = tr.lb >= 0 ? G->lbs + tr.lb : (const WV_LDS i32 *)0;
               u32 s = tr.seed;
               int jdone = 0;
               FOR_GL(j, N) {
                  for (; jdone <= j; jdone++) s = lcg_rand(s);
                  if (lowband == 0) X[j] = shl32((i32)((i32)s >> 20), NORM_SHIFT - 14);
                  else { i16 tmp = QC16(1.0f / 256, NORM_SHIFT - 4); tmp = (s) & 0x8000 ? tmp : -tmp; X[j] = lowband[j] + tmp; }
               }
               cm = lowband == 0 ? cm_mask : (unsigned)fill;
               for (int j = 0; j < N; j++) tr.seed = lcg_rand(tr.seed);
               wg_sync();
               p4_renormalise_vector(X, N, tr.gain);
            }
         }
      }
      P4_TOC(8);
      if (!tr.done) {
         for (;;) {
            if (tr.depth == 0) { tr.done = 1; tr.cm = cm; break; }
            WV_LDS P4Frame *f = &G->stk[tr.depth - 1];
            wg_sync();
            const int phase = f->phase, mid_first = f->mid_first, B0 = f->B0;
            if (phase == 0) {
               int mbits = f->mbits, sbits = f->sbits;
               const int itheta = f->itheta, N = f->N, B = f->B;
               i32 rebalance = f->rebal;
               unsigned fcm;
               tr.N = N; tr.B = B; tr.LM = f->LM;
               if (mid_first) {
                  fcm = cm;
                  rebalance = mbits - (rebalance - tr.remaining_bits);
                  if (rebalance > 3 << BITRES && itheta != 0) sbits += rebalance - (3 << BITRES);
                  tr.xo = f->xo + N; tr.b = sbits; tr.gain = f->gs; tr.fill = f->fill >> B; tr.lb = f->lb >= 0 ? f->lb + N : -1;
               } else {
                  fcm = cm << (B0 >> 1);
                  rebalance = sbits - (rebalance - tr.remaining_bits);
                  if (rebalance > 3 << BITRES && itheta != 16384) mbits += rebalance - (3 << BITRES);
                  tr.xo = f->xo; tr.b = mbits; tr.gain = f->gm; tr.fill = f->fill; tr.lb = f->lb;
               }
               wg_sync();
               GLANE0 { f->phase = 1; f->cm = (i32)fcm; }
               break;
            }
            cm = (unsigned)f->cm | (mid_first ? cm << (B0 >> 1) : cm);
            tr.depth--;
         }
      }
      P4_TOC(9);
   }
}

/* quant_all_bands (bands.c:1589) with encode = 0 for the (up to) four streams of a wave, all of one frame size.  cont: this lane's group's record, or NULL (a group without a
 * stream).  In: the record's image (coder, allocation, tf, frame constants; the spectrum zeroed by the front kernel).  Out: the spectrum and the folding memory in the
 * record's xg, the collapse masks, the coder and the LCG seed back in the image. */
WV_DEV void p4d_quant_all_bands(WV_LDS P4Lds *L4, CeltDecCont *cont)
{
   WV_LDS P4Group *G = &L4->g[wg_id()];
   const int active = cont != 0;
   DecLds *img = (DecLds *)(cont ? cont->image : 0);
   int start = 0, end = 0, LM = 0, C = 1, shortBlocks = 0, spread = 0, dual_stereo = 0, intensity = 0, codedBands = 0, disable_inv = 0, Nfull = 0;
   i32 total_bits = 0, balance = 0;
   u32 seed = 0;
   if (active) {
      start = img->sh.start; end = img->sh.end; LM = img->sh.LM; C = img->sh.C; shortBlocks = img->sh.shortBlocks; spread = img->sh.spread; dual_stereo = img->sh.dual_stereo;
      intensity = img->sh.intensity; codedBands = img->sh.codedBands; disable_inv = img->st.disable_inv; balance = img->sh.balance; total_bits = img->sh.pvq_total_bits;
      Nfull = img->sh.N;
      seed = img->st.rng;
      wg_sync();
      FOR_GL(k, NBE) { G->pulses[k] = img->pulses[k]; G->tf_res[k] = img->tf_res[k]; }
      FOR_GL(k, (int)(sizeof(EcCtx) / 4)) ((WV_LDS i32 *)&G->ec)[k] = ((const i32 *)&img->ec)[k];
      wg_sync();
   }
   const u8 *const pkt = active ? (const u8 *)img->packet + 1 : (const u8 *)0;
   i32 *const Xo = active ? cont->xg : (i32 *)0, *const Yo = active && C == 2 ? cont->xg + Nfull : (i32 *)0;
   i32 *const norm = active ? cont->xg + 2 * OA_MAX_FRAME : (i32 *)0, *const norm2 = active ? cont->xg + 2 * OA_MAX_FRAME + OA_NORM_LEN : (i32 *)0;
   /* wave-uniform: the frame size class (the kernel fills a wave with frames of one size), the union of the groups' band ranges */
   const int LMu = wv_max(active ? LM : 0), M = 1 << LMu;
   const int i_lo = wv_min(active ? start : NBE), i_hi = wv_max(active ? end : 0);
   int B = shortBlocks ? M : 1, lowband_offset = 0, update_lowband = 1;
   const int norm_offset = M * ct_eBands[start];
   P4Cfg cfg;
   cfg.intensity = intensity; cfg.spread = spread; cfg.disable_inv = disable_inv; cfg.resynth = 1; cfg.theta_round = 0; cfg.avoid_split_noise = B > 1; cfg.i = 0; cfg.tf_change = 0;
   for (int i = i_lo; i < i_hi; i++) {
      const int N = M * ct_eBands[i + 1] - M * ct_eBands[i];          /* wave-uniform */
      P4_TIC();
      p4_rows_stage(L4, i);
      const int act = active && i >= start && i < end;
      const int last = i == end - 1;
      i32 remaining_bits = 0, tell = 0;
      int b = 0, effective_lowband = -1, tf_change = 0;
      unsigned x_cm = 0, y_cm = 0;
      cfg.i = i;
      if (act) {
         wg_sync();
         tell = (i32)ec_tell_frac_lds(&G->ec);
         if (i != start) balance -= tell;
         remaining_bits = total_bits - tell - 1;
         if (i <= codedBands - 1) {
            const i32 curr_balance = fx_sdiv24(balance, imin(3, codedBands - i));
            b = imax(0, imin(16383, imin(remaining_bits + 1, G->pulses[i] + curr_balance)));
         } else b = 0;
         if ((M * ct_eBands[i] - N >= M * ct_eBands[start] || i == start + 1) && (update_lowband || lowband_offset == 0)) lowband_offset = i;
         if (i == start + 1) {                                             /* special_hybrid_folding (bands.c:1575) */
            const int hf_n1 = M * (ct_eBands[start + 1] - ct_eBands[start]), hf_n2 = M * (ct_eBands[start + 2] - ct_eBands[start + 1]);
            FOR_GL(j, hf_n2 - hf_n1) { norm[hf_n1 + j] = norm[2 * hf_n1 - hf_n2 + j]; if (dual_stereo) norm2[hf_n1 + j] = norm2[2 * hf_n1 - hf_n2 + j]; }
         }
         tf_change = G->tf_res[i];
         cfg.tf_change = tf_change;
         if (lowband_offset != 0 && (spread != 3 || B > 1 || tf_change < 0)) {
            int fold_start, fold_end, fold_i;
            effective_lowband = imax(0, M * ct_eBands[lowband_offset] - norm_offset - N);
            fold_start = lowband_offset;
            while (M * ct_eBands[--fold_start] > effective_lowband + norm_offset);
            fold_end = lowband_offset - 1;
            while (++fold_end < i && M * ct_eBands[fold_end] < effective_lowband + norm_offset + N);
            fold_i = fold_start;
            do {
               x_cm |= G->cmask[fold_i * C + 0];
               y_cm |= G->cmask[fold_i * C + C - 1];
            } while (++fold_i < fold_end);
         } else x_cm = y_cm = (1u << B) - 1;
         if (dual_stereo && i == intensity) {
            dual_stereo = 0;
            FOR_GL(j, M * ct_eBands[i] - norm_offset) norm[j] = half32(norm[j] + norm2[j]);
         }
      }
      /* the slots of this band: mono -- one quant_band; dual stereo -- the two channels one after the other with half the budget each (bands.c:1831-1841); joint stereo --
       * quant_band_stereo: theta, then mid and side in the order of their budgets.  Channel 0 / the mid is decoded in Xb, channel 1 / the side in Yb. */
      const int joint = act && Yo != 0 && !dual_stereo;
      const int nslots = act ? ((dual_stereo || joint) ? 2 : 1) : 0;
      i32 *const lbo = last ? (i32 *)0 : norm + M * ct_eBands[i] - norm_offset, *const lbo2 = last ? (i32 *)0 : norm2 + M * ct_eBands[i] - norm_offset;
      const unsigned cm_in = x_cm | y_cm;
      int mbits = 0, sbits = 0, itheta = 0, inv = 0, mid_first = 1, fill_j = 0;
      i32 mid = 0, side = 0, rebalance = 0;
      P4_TOC(0);
      if (act) {
         wg_sync();
         if (effective_lowband != -1) { FOR_GL(j, N) G->lbs[j] = norm[effective_lowband + j]; }
         wg_sync();
         if (joint) {
            const P4Theta th = p4d_compute_theta(G, pkt, cfg, remaining_bits, N, b, B, B, LMu, 1, (int)cm_in);
            inv = th.inv; itheta = th.itheta; fill_j = th.fill;
            mid = shl32((i32)th.imid, 16); side = shl32((i32)th.iside, 16);
            mbits = imax(0, imin(th.b, (th.b - th.delta) / 2));
            sbits = th.b - mbits;
            remaining_bits -= th.qalloc;
            rebalance = remaining_bits;
            mid_first = mbits >= sbits;
         }
      }
      for (int sl = 0; sl < 2; sl++) {
         const int s_act = sl < nslots;
         if (!wv_any(s_act)) break;
         P4Tree tr; P4Qb qb;
         tr.act = s_act; tr.xo = 0; tr.N = N; tr.b = 0; tr.B = B; tr.LM = LMu; tr.lb = -1; tr.fill = 0; tr.gain = Q31ONE; tr.remaining_bits = remaining_bits; tr.seed = seed;
         tr.depth = 0; tr.done = 1; tr.cm = 0;
         qb.N0 = N; qb.recombine = 0; qb.time_divide = 0; qb.B0 = B; qb.N_B0 = N; qb.longBlocks = 1; qb.B = B;
         i32 *lb_out = (i32 *)0;
         if (s_act) {
            if (joint) {
               const int do_mid = (sl == 0) == (mid_first != 0);
               if (sl == 1) {
                  if (mid_first) { rebalance = mbits - (rebalance - remaining_bits); if (rebalance > 3 << BITRES && itheta != 0) sbits += rebalance - (3 << BITRES); }
                  else { rebalance = sbits - (rebalance - remaining_bits); if (rebalance > 3 << BITRES && itheta != 16384) mbits += rebalance - (3 << BITRES); }
               }
               tr.xo = do_mid ? 0 : OA_MAX_BAND; tr.b = do_mid ? mbits : sbits; tr.lb = do_mid && effective_lowband != -1 ? 0 : -1; tr.gain = do_mid ? Q31ONE : side;
               tr.fill = do_mid ? fill_j : fill_j >> B;
               lb_out = do_mid ? lbo : (i32 *)0;
            } else if (dual_stereo) {
               if (sl == 1) { wg_sync(); if (effective_lowband != -1) { FOR_GL(j, N) G->lbs[j] = norm2[effective_lowband + j]; } wg_sync(); }
               tr.xo = sl ? OA_MAX_BAND : 0; tr.b = b / 2; tr.lb = effective_lowband != -1 ? 0 : -1; tr.fill = (int)(sl ? y_cm : x_cm);
               lb_out = sl ? lbo2 : lbo;
            } else {
               tr.b = b; tr.lb = effective_lowband != -1 ? 0 : -1; tr.fill = (int)cm_in;
               lb_out = lbo;
            }
            p4_qb_pre(G, tr, qb, tf_change, N, 0);
         }
         const int xo = tr.xo;
         P4_TIC();
         p4d_tree_run(L4, G, pkt, cfg, tr);
         P4_TOC(24);
         if (s_act) {
            const unsigned cm = p4_qb_post(G, tr, qb, xo, 1, lb_out, N);
            remaining_bits = tr.remaining_bits; seed = tr.seed;
            if (joint) x_cm = sl == 0 ? cm : (x_cm | cm);
            else if (dual_stereo) { if (sl == 0) x_cm = cm; else y_cm = cm; }
            else { x_cm = cm; y_cm = cm; }
         }
      }
      {
      P4_TIC();
      if (act) {
         if (joint) {
            p4_stereo_merge(G->Xb, G->Yb, mid, N);
            if (inv) { wg_sync(); FOR_GL(j, N) G->Yb[j] = neg32(G->Yb[j]); wg_sync(); }
            y_cm = x_cm;
         }
         wg_sync();
         FOR_GL(j, N) { Xo[M * ct_eBands[i] + j] = G->Xb[j]; if (Yo != 0) Yo[M * ct_eBands[i] + j] = G->Yb[j]; }       /* the finished band -> the spectrum */
         GLANE0 { G->cmask[i * C + 0] = (u8)x_cm; G->cmask[i * C + C - 1] = (u8)y_cm; }
         balance += G->pulses[i] + tell;
         update_lowband = b > (N << BITRES);
         cfg.avoid_split_noise = 0;
      }
      P4_TOC(20);
      }
   }
   if (active) {
      wg_sync();
      FOR_GL(k, (int)(sizeof(EcCtx) / 4)) ((i32 *)&img->ec)[k] = ((const WV_LDS i32 *)&G->ec)[k];
      FOR_GL(k, 2 * NBE) img->collapse_masks[k] = G->cmask[k];
      GLANE0 img->st.rng = seed;
   }
}
#endif

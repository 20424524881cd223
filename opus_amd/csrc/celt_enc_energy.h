/* celt_enc_energy.h — band-energy quantisation of the CELT frame encoder on one wavefront.
 *
 * What the bitstream fixes (celt/quant_bands.c:142-429, celt/laplace.c:44-92): the prediction recurrence along the bands, the Laplace model per band, the
 * budget fall-backs, and -- with the two-pass search -- that the frame is coded under BOTH hypotheses (intra = no inter-frame prediction, inter) and the
 * better one kept.  How it runs here:
 *   coarse  the two hypotheses are independent until the comparison, so they run CONCURRENTLY on lanes 0 and 1: each lane owns a private coder state, a
 *           byte target (lane 1 the packet, lane 0 a side buffer with the same indexing) and a private (oldE, residual) pair; what does not depend on the
 *           hypothesis (clamped / floored previous energies, the loss-robustness distortion) is computed by all lanes beforehand.  The winner's state is
 *           adopted with coalesced copies.
 *   fine    which bands still fit is a running sum of known bit counts (lane 0, no coder involved); the refinement of every (band, channel) is then
 *           independent -> one lane each; lane 0 only emits the raw bits in order.
 *   final   same split: lane 0 hands out the left-over bits by priority, the updates are per-lane. */
#ifndef OPUS_AMD_CELT_ENC_ENERGY_H
#define OPUS_AMD_CELT_ENC_ENERGY_H
#define OA_MAX_FINE_BITS 8

WV_TABLE i16 k_inter_pred[4] = {29440, 26112, 21248, 16384};       /* inter-frame prediction per LM, Q15 */
WV_TABLE i16 k_inter_leak[4] = {30147, 22282, 12124, 6554};        /* along-band leak per LM, Q15; intra: 4915 */
WV_TABLE u8 k_tiny_energy_icdf[3] = {2, 1, 0};

/* Laplace-distributed integer -> range-coder interval.  The model is {P(0) = p0 / 32768, geometric tails with ratio `decay` / 32768, floor of one count per
 * value}; *v may be pulled towards zero when the tail runs out of probability mass (laplace.c:51-92). */
WV_DEV void oa_laplace_put(EC_ARGS, int *v, unsigned p0, int decay)
{
   unsigned lo = 0, width = p0;
   const int mag0 = iabs(*v);
   if (mag0 != 0) {
      const int neg = *v < 0;
      int step = 1;
      lo = p0;
      width = (32768u - 32u - p0) * (u32)(16384 - decay) >> 15;                    /* mass of +-1, before the per-value floor */
      while (width > 0 && step < mag0) { width *= 2; lo += width + 2; width = (width * (u32)decay) >> 15; step++; }
      if (width == 0) {                                                            /* in the flat floor region: one count per value */
         int room = (int)(32768u - lo);                                            /* values still representable on this side, both signs */
         room = (room + neg) >> 1;
         const int extra = imin(mag0 - step, room - 1);
         lo += (unsigned)(2 * extra + 1 - neg);
         width = imin(1, (int)(32768u - lo));
         *v = neg ? -(step + extra) : step + extra;
      } else {
         width += 1;
         if (!neg) lo += width;
      }
   }
   k_ec_encode_bin(EC_PASS, lo, lo + width, 15);
}

struct CoarseScratch {                 /* lives in BC while the coarse energies are coded */
   u8 side[OA_MAX_PACKET + 4];         /* bytes of the intra hypothesis */
   i32 soft[2 * NBE], floorE[2 * NBE]; /* max(prev, -9 dB) and max(prev - max_decay, -28 dB): hypothesis-independent */
   i32 altE[2 * NBE], altR[2 * NBE];   /* oldBandE / residual of the intra hypothesis */
};

/* quant_coarse_energy (quant_bands.c:260) with the two hypotheses on two lanes.  In: L->bandLogE, L->oldBandE (previous frame), coder in L->ec.  Out: L->oldBandE
 * (quantised), L->error (residual), coder advanced, st->delayedIntra updated. */
WV_DEV void coarse_energy_wave(WV_LDS FrameLds *L)
{
   WV_LDS FrameShared *sh = &L->sh;
   WV_LDS CoarseScratch *K = (WV_LDS CoarseScratch *)&L->BC;
   const int lane = wv_lane();
   const int C = sh->C, LM = sh->LM, start = sh->start, end = sh->end, lfe = sh->lfe, nbands = end - start;
   const i32 budget = sh->total_bits;
   const int avail = sh->nbAvailableBytes, delayed = L->st.delayedIntra;
   int two_pass = sh->complexity >= 4;
   /* how badly a lost frame would hurt an inter-coded successor: squared energy change over the coded bands (loss_distortion :142) */
   i32 change = 0;
   FOR_LANES(w, C * NBE) { const int c = w / NBE, i = w - c * NBE; if (i >= start && i < sh->effEnd) { const i32 d = pshr32(sub32(L->bandLogE[w], L->oldBandE[w]), DB_SHIFT - 7); change += (i16)d * (i32)(i16)d; } }
   const i32 distortion = imin(200, wv_sum(change) >> 14);
   EcCtx mine; ec_ld(&mine, &L->ec);
   const i32 tell0 = k_ec_tell(&mine, L->packet + 1);
   int intra_only = sh->force_intra || (!two_pass && delayed > 2 * C * nbands && avail > nbands * C);
   if (tell0 + 3 > budget) { two_pass = 0; intra_only = 0; }
   /* (budget * delayedIntra * loss_rate) / (C * 512) in the reference's 32-bit signed arithmetic (quant_bands.c:279): the product wraps at high loss rates, and the division is signed */
   const i32 intra_bias = (i32)((u32)budget * (u32)delayed * (u32)sh->loss_rate) / (C * 512);
   i32 max_decay = GC(16.f);
   if (nbands > 10) max_decay = shl32(imin(max_decay >> (DB_SHIFT - 3), avail), DB_SHIFT - 3);
   if (lfe) max_decay = GC(3.f);
   FOR_LANES(w, 2 * NBE) {
      const i32 p = L->oldBandE[w];
      K->soft[w] = imax(-GC(9.f), p); K->floorE[w] = imax(-GC(28.f), sub32(p, max_decay)); K->altE[w] = p;
   }
   wv_sync();

   /* lane 0: intra, lane 1: inter */
   const int hyp_intra = lane == 0;
   const int running = lane < 2 && (hyp_intra ? (two_pass || intra_only) : !intra_only);
   int penalty = 0;                    /* how far the budget forced the indices away from the wanted ones */
   if (running) {
      EcCtx *e = &mine;
      WV_LDS u8 *buf = hyp_intra ? K->side : L->packet + 1;
      WV_LDS i32 *qE = hyp_intra ? K->altE : L->oldBandE, *res = hyp_intra ? K->altR : L->error;
      const i16 pred = hyp_intra ? 0 : k_inter_pred[LM], leak = hyp_intra ? 4915 : k_inter_leak[LM];
      const u8 *model = ct_e_prob_model[LM][hyp_intra];
      if (tell0 + 3 <= budget) k_ec_enc_bit_logp(EC_PASS, hyp_intra, 3);
      i32 carry[2] = {0, 0};
      for (int i = start; i < end; i++) {
         const int m = 2 * imin(i, 20);
         for (int c = 0; c < C; c++) {
            const int w = i + c * NBE;
            const i32 x = L->bandLogE[w], base = mult16_32_q15(pred, K->soft[w]) + carry[c], want = x - base;
            int q = (want + QC32(.5f, DB_SHIFT)) >> DB_SHIFT;
            if (q < 0 && x < K->floorE[w]) q = imin(0, q + (int)(sub32(K->floorE[w], x) >> DB_SHIFT));     /* never decay faster than max_decay per frame */
            const int wanted = q;
            const i32 used = k_ec_tell(EC_PASS), room = budget - used;
            const int spare = room - 3 * C * (end - i);                                                       /* after 3 bits for every value still to come */
            if (i != start && spare < 30) { if (spare < 24) q = imin(1, q); if (spare < 16) q = imax(-1, q); }
            if (lfe && i >= 2) q = imin(q, 0);
            if (room >= 15) oa_laplace_put(EC_PASS, &q, (unsigned)model[m] << 7, (int)model[m + 1] << 6);
            else if (room >= 2) { q = imax(-1, imin(q, 1)); k_ec_enc_icdf(EC_PASS, q < 0 ? -2 * q - 1 : 2 * q, k_tiny_energy_icdf, 2); }
            else if (room >= 1) { q = imin(0, q); k_ec_enc_bit_logp(EC_PASS, -q, 1); }
            else q = -1;
            penalty += iabs(wanted - q);
            const i32 step = shl32(q, DB_SHIFT);
            res[w] = want - step;
            qE[w] = imax(-GC(28.f), base + step);
            carry[c] += step - mult16_32_q15(leak, step);
         }
      }
      if (lfe) penalty = 0;
   }
   /* compare (wave-uniform): inter unless the intra pass lost less to the budget, or equally much in fewer bits (biased by the loss rate) */
   const i32 frac = running ? (i32)k_ec_tell_frac(&mine, L->packet + 1) : 0;
   const int pen_intra = wv_bcast(penalty, 0), pen_inter = wv_bcast(penalty, 1);
   const i32 frac_intra = wv_bcast(frac, 0), frac_inter = wv_bcast(frac, 1);
   const int off0 = (int)L->ec.offs, off_intra = wv_bcast((i32)mine.offs, 0);
   int intra = intra_only;
   if (!intra_only && two_pass && (pen_intra < pen_inter || (pen_intra == pen_inter && frac_inter + intra_bias > frac_intra))) intra = 1;
   wv_sync();
   if (lane == (intra ? 0 : 1)) ec_st(&L->ec, &mine);
   if (intra) {
      FOR_LANES(w, C * NBE) { L->oldBandE[w] = K->altE[w]; L->error[w] = K->altR[w]; }
      FOR_LANES(k, off_intra - off0) L->packet[1 + off0 + k] = K->side[off0 + k];
   }
   if (lane == 0) L->st.delayedIntra = intra ? distortion : add32(mult16_32_q15((i16)mult16_16_q15(k_inter_pred[LM], k_inter_pred[LM]), delayed), distortion);
   wv_sync();
}

/* quant_fine_energy (quant_bands.c:360): fine_quant[i] more bits for every channel of band i while they fit; raw bits, so the position after each band is known
 * without running the coder */
WV_DEV void fine_energy_wave(WV_LDS FrameLds *L)
{
   const int C = L->sh.C, start = L->sh.start, end = L->sh.end;
   WV_LDS i32 *sym = L->scr;           /* [2 * NBE] refinement symbols */
   LANE0 {
      i32 pos = ec_tell_lds(&L->ec); const i32 cap = (i32)L->ec.storage * 8;
      u32 fits = 0;
      for (int i = start; i < end; i++) { const int n = L->fine_quant[i]; if (n > 0 && pos + C * n <= cap) { fits |= 1u << i; pos += C * n; } }
      L->sh.r[5] = (i32)fits;
   }
   const u32 fits = (u32)wv_uni(L->sh.r[5]);
   FOR_LANES(w, C * NBE) {
      const int c = w / NBE, i = w - c * NBE;
      if (fits >> i & 1) {
         const int n = L->fine_quant[i];
         const int q = imax(0, imin((1 << n) - 1, vshr32(add32(L->error[w], GC(.5f)), DB_SHIFT - n)));
         const i32 mid = sub32(vshr32(2 * q + 1, n - DB_SHIFT + 1), GC(.5f));                                 /* centre of cell q, relative to the coarse value */
         sym[w] = q; L->oldBandE[w] += mid; L->error[w] -= mid;
      }
   }
   LANE0 {
      EC_BEGIN;
      for (int i = start; i < end; i++) if (fits >> i & 1) for (int c = 0; c < C; c++) k_ec_enc_bits(EC_PASS, (u32)sym[i + c * NBE], (unsigned)L->fine_quant[i]);
      EC_END;
   }
}

/* quant_energy_finalise (quant_bands.c:401): whatever whole bits are left go to one more halving per band, priority 0 bands first.  Called with the coder in
 * registers (lane 0, inside the frame's last serial section); the array updates it decides are applied by energy_finalise_apply_wave afterwards. */
WV_DEV u32 energy_finalise_emit_l0(WV_LDS FrameLds *L, EC_ARGS, int bits_left)
{
   const int C = L->sh.C, start = L->sh.start, end = L->sh.end;
   u32 given = 0;
   for (int prio = 0; prio < 2; prio++)
      for (int i = start; i < end && bits_left >= C; i++)
         if (L->fine_quant[i] < OA_MAX_FINE_BITS && L->fine_priority[i] == prio) {
            for (int c = 0; c < C; c++) k_ec_enc_bits(EC_PASS, L->error[i + c * NBE] < 0 ? 0u : 1u, 1);
            given |= 1u << i; bits_left -= C;
         }
   return given;
}
WV_DEV void energy_finalise_apply_wave(WV_LDS FrameLds *L, u32 given)
{
   const int C = L->sh.C;
   FOR_LANES(w, C * NBE) {
      const int c = w / NBE, i = w - c * NBE;
      if (given >> i & 1) {
         const i32 half_cell = GC(.5f) >> (L->fine_quant[i] + 1), mv = L->error[w] < 0 ? -half_cell : half_cell;
         L->oldBandE[w] += mv; L->error[w] -= mv;
      }
   }
   wv_sync();
}
#endif

/* celt_alloc.h — bit allocation of a CELT frame (encoder and decoder run the identical computation; only the three side-information symbols differ in
 * direction).  Format: celt/rate.c:249-653 (clt_compute_allocation, interp_bits2pulses), celt/celt.c:329 (init_caps), celt/rate.h:48-88 (pulse cache).
 *
 * The reference searches twice by bisection, each probe summing a clipped per-band cost from the top band down: first over the 11 rows of the static allocation
 * table, then over 64 interpolation steps between the two bracketing rows.  Every probe is independent of the others, so here ALL of them are evaluated at once --
 * row r on lane r, interpolation step s on lane s (64 steps = 64 lanes) -- and the bisection is then replayed on the finished sums with wave broadcasts: the same
 * decisions in the same order, six dependent probe evaluations replaced by one.  Per-band vectors (thresholds, trims, the two bracketing allocations, the final
 * spreading of the remainder) are one lane per band; "some higher band already qualified" is a suffix-OR over a wave ballot.  What stays serial is what the coder
 * or a running balance chains together: the band-skip loop and the pulse / fine-energy split. */
#ifndef OPUS_AMD_CELT_ALLOC_H
#define OPUS_AMD_CELT_ALLOC_H
#define OA_FINE_OFFSET 21
#define OA_INTERP_STEPS_LOG2 6
#define OA_ALLOC_ROWS 11
#ifndef OA_MAX_FINE_BITS
#define OA_MAX_FINE_BITS 8
#endif

WV_TABLE u8 k_log2_frac_q3[24] = {0, 8, 13, 16, 19, 21, 23, 24, 26, 27, 28, 29, 30, 31, 32, 32, 33, 34, 34, 35, 36, 36, 37, 37};   /* 8 * log2(i + 1), rounded up */

/* pseudo-pulse index -> pulse count (rate.h:48) */
WV_DEV int k_get_pulses(int i) { return i < 8 ? i : (8 + (i & 7)) << ((i >> 3) - 1); }
/* init_caps (celt.c:329): most bits band i can use, 1/8 bit units; one lane per band */
WV_DEV void k_init_caps(WV_LDS int *cap, int LM, int C)
{
   for (int i = 0; i < OA_NB_EBANDS; i++) cap[i] = (ct_cache_caps[OA_NB_EBANDS * (2 * LM + C - 1) + i] + 64) * C * ((ct_eBands[i + 1] - ct_eBands[i]) << LM) >> 2;
}

/* cost of one probe as seen by the lane that owns it: walk the bands from the top; below the first band that reaches its threshold a band gets either its clipped
 * share or -- if it at least reaches the floor -- exactly the floor */
#define OA_PROBE_SUM(EXPR_BITS)                                                                   \
   int total_ = 0, seen_ = 0;                                                                      \
   for (int j = end; j-- > start;) {                                                               \
      const int b_ = (EXPR_BITS);                                                                  \
      if (b_ >= thresh[j] || seen_) { seen_ = 1; total_ += imin(b_, cap[j]); }                     \
      else if (b_ >= floor_bits) total_ += floor_bits;                                             \
   }

/* scr: 4 * 21 words.  The coder is taken from / returned to *ecl around the serial part.  ENC: the skip / intensity / dual-stereo symbols are written (from
 * *intensity, *dual_stereo, clipped to what was coded), else read.  Returns codedBands. */
template <bool ENC> WV_DEV int oa_allocate_bits_wave(WV_LDS EcCtx *ecl, WV_LDS u8 *ecbuf, WV_LDS i32 *scr, int start, int end, const WV_LDS i32 *offsets, const WV_LDS i32 *cap,
      int alloc_trim, WV_LDS i32 *intensity, WV_LDS i32 *dual_stereo, i32 total, WV_LDS i32 *balance_out, WV_LDS i32 *bits, WV_LDS i32 *ebits, WV_LDS i32 *fine_priority,
      int C, int LM, int prev_coded, int signalBandwidth, WV_LDS i32 *hand)
{
   const int lane = wv_lane();
   const int16_t *eB = ct_eBands;
   WV_LDS i32 *lowv = scr, *spanv = scr + 21, *thresh = scr + 42, *trim = scr + 63;
   const int floor_bits = C << BITRES, stereo = C > 1;
   start = wv_uni(start); end = wv_uni(end); C = wv_uni(C); LM = wv_uni(LM);
   /* reservations: one 1/8-bit-resolution bit each for "skip", intensity (log2 of the choices) and dual stereo */
   total = imax(total, 0);
   const int skip_rsv = total >= 1 << BITRES ? 1 << BITRES : 0;
   total -= skip_rsv;
   int intensity_rsv = 0, dual_rsv = 0;
   if (C == 2) {
      intensity_rsv = k_log2_frac_q3[end - start];
      if (intensity_rsv > total) intensity_rsv = 0;
      else { total -= intensity_rsv; dual_rsv = total >= 1 << BITRES ? 1 << BITRES : 0; total -= dual_rsv; }
   }
   wv_sync();
   if (lane >= start && lane < end) {
      const int j = lane, w = eB[j + 1] - eB[j];
      thresh[j] = imax(floor_bits, (3 * w << LM << BITRES) >> 4);                                            /* below this a band is not worth opening */
      int t = C * w * (alloc_trim - 5 - LM) * (end - j - 1) * (1 << (LM + BITRES)) >> 6;                       /* tilt */
      if (w << LM == 1) t -= floor_bits;
      trim[j] = t;
   }
   wv_sync();
   /* ---- which two rows of the table bracket the budget: all rows at once, then the bisection on the sums ---- */
   int row_cost = 0;
   if (lane >= 1 && lane < OA_ALLOC_ROWS) {
      OA_PROBE_SUM(({ int v = C * (eB[j + 1] - eB[j]) * ct_allocVectors[lane * OA_NB_EBANDS + j] << LM >> 2; if (v > 0) v = imax(0, v + trim[j]); v + offsets[j]; }))
      row_cost = total_;
   }
   int lo = 1, hi = OA_ALLOC_ROWS - 1;
   do { const int mid = (lo + hi) >> 1; if (wv_bcast(row_cost, mid) > total) hi = mid - 1; else lo = mid + 1; } while (lo <= hi);
   hi = lo--;
   int boosted = 0;
   if (lane >= start && lane < end) {
      const int j = lane, n = C * (eB[j + 1] - eB[j]);
      int a = n * ct_allocVectors[lo * OA_NB_EBANDS + j] << LM >> 2;
      int b = hi >= OA_ALLOC_ROWS ? cap[j] : n * ct_allocVectors[hi * OA_NB_EBANDS + j] << LM >> 2;
      if (a > 0) a = imax(0, a + trim[j]);
      if (b > 0) b = imax(0, b + trim[j]);
      if (lo > 0) a += offsets[j];
      b += offsets[j];
      boosted = offsets[j] > 0;
      lowv[j] = a; spanv[j] = imax(0, b - a);
   }
   const u64 boosted_mask = wv_ballot(boosted);
   const int skip_start = boosted_mask ? 63 - __builtin_clzll(boosted_mask) : start;                          /* bands up to the last boosted one are never skipped */
   wv_sync();
   /* ---- how far between the two rows: interpolation step s on lane s ---- */
   int step_cost;
   { OA_PROBE_SUM(lowv[j] + (lane * (i32)spanv[j] >> OA_INTERP_STEPS_LOG2)) step_cost = total_; }
   int frac = 0, top = 1 << OA_INTERP_STEPS_LOG2;
   for (int it = 0; it < OA_INTERP_STEPS_LOG2; it++) { const int mid = (frac + top) >> 1; if (wv_bcast(step_cost, mid) > total) top = mid; else frac = mid; }
   /* the allocation at that point: per band, with "a higher band qualified" as a suffix of the ballot */
   int mine = 0, qual = 0;
   if (lane >= start && lane < end) { mine = lowv[lane] + ((i32)frac * spanv[lane] >> OA_INTERP_STEPS_LOG2); qual = mine >= thresh[lane]; }
   const u64 qmask = wv_ballot(qual);
   if (lane >= start && lane < end) {
      if (!(qmask >> lane)) mine = mine >= floor_bits ? floor_bits : 0;
      mine = imin(mine, cap[lane]);
      bits[lane] = mine;
   } else mine = 0;
   i32 psum = wv_sum(mine);
   wv_sync();
   /* ---- serial: skip bands from the top while they cannot carry at least a pulse and a bit, then the stereo parameters ---- */
   LANE0 {
      EcCtx ec_; ec_ld(&ec_, ecl); EcCtx *e = &ec_; WV_LDS u8 *buf = ecbuf;
      int coded = end;
      i32 tot = total;
      int irsv = intensity_rsv, drsv = dual_rsv;
      for (;; coded--) {
         const int j = coded - 1;
         if (j <= skip_start) { tot += skip_rsv; break; }
         const int span = eB[coded] - eB[start];
         i32 left = tot - psum;
         const i32 per = (i32)((u32)left / (u32)span);
         left -= span * per;
         const int width = eB[coded] - eB[j];
         int would_get = (int)(bits[j] + per * width + imax(left - (eB[j] - eB[start]), 0));
         if (would_get >= imax(thresh[j], floor_bits + (1 << BITRES))) {
            int keep;
            if (ENC) {
               const int depth = coded > 17 ? (j < prev_coded ? 7 : 9) : 0;                                   /* hysteresis on the previous frame's coded bands */
               keep = coded <= start + 2 || (would_get > (depth * width << LM << BITRES) >> 4 && j <= signalBandwidth);
               k_ec_enc_bit_logp(EC_PASS, keep, 1);
            } else keep = k_ec_dec_bit_logp(EC_PASS, 1);
            if (keep) break;
            psum += 1 << BITRES; would_get -= 1 << BITRES;
         }
         psum -= bits[j] + irsv;
         if (irsv > 0) irsv = k_log2_frac_q3[j - start];
         psum += irsv;
         if (would_get >= floor_bits) { psum += floor_bits; bits[j] = floor_bits; } else bits[j] = 0;
      }
      if (irsv > 0) {
         if (ENC) { *intensity = imin(*intensity, coded); k_ec_enc_uint(EC_PASS, (u32)(*intensity - start), (u32)(coded + 1 - start)); }
         else *intensity = start + (int)k_ec_dec_uint(EC_PASS, (u32)(coded + 1 - start));
      } else *intensity = 0;
      if (*intensity <= start) { tot += drsv; drsv = 0; }
      if (drsv > 0) { if (ENC) k_ec_enc_bit_logp(EC_PASS, *dual_stereo, 1); else *dual_stereo = k_ec_dec_bit_logp(EC_PASS, 1); }
      else *dual_stereo = 0;
      ec_st(ecl, &ec_);
      hand[0] = coded; hand[1] = tot - psum;
   }
   const int coded = wv_uni(hand[0]);
   /* ---- what is left is spread evenly per coefficient, the remainder one 1/8 bit per coefficient from the bottom: a prefix sum of the widths ---- */
   {
      const i32 left0 = wv_uni(hand[1]);
      const int span = eB[coded] - eB[start];
      const i32 per = (i32)((u32)left0 / (u32)span), rem = left0 - span * per;
      if (lane >= start && lane < coded) bits[lane] += per * (eB[lane + 1] - eB[lane]) + imax(0, imin(rem - (eB[lane] - eB[start]), eB[lane + 1] - eB[lane]));
   }
   wv_sync();
   /* ---- serial: pulses versus fine energy per band, surplus above the cap rolls into the next band ---- */
   LANE0 {
      const int dual = *dual_stereo, inten = *intensity, logM = LM << BITRES;
      i32 roll = 0;
      int j = start;
      for (; j < coded; j++) {
         const int N = (eB[j + 1] - eB[j]) << LM;
         const i32 have = bits[j] + roll;
         i32 over; int fine;
         if (N > 1) {
            over = imax(have - cap[j], 0);
            const i32 b = have - over;
            const int dof = C * N + (C == 2 && N > 2 && !dual && j < inten ? 1 : 0);                           /* degrees of freedom (+1 for the stereo angle) */
            const int lg = dof * (ct_logN[j] + logM);
            int bias = (lg >> 1) - dof * OA_FINE_OFFSET;
            if (N == 2) bias += dof << BITRES >> 2;
            if (b + bias < dof * 2 << BITRES) bias += lg >> 2;                                                   /* few pulses: resolution matters more */
            else if (b + bias < dof * 3 << BITRES) bias += lg >> 3;
            fine = imax(0, b + bias + (dof << (BITRES - 1)));
            fine = (int)((u32)fine / (u32)dof) >> BITRES;
            if (C * fine > (b >> BITRES)) fine = b >> stereo >> BITRES;
            fine = imin(fine, OA_MAX_FINE_BITS);
            fine_priority[j] = fine * (dof << BITRES) >= b + bias;
            bits[j] = b - (C * fine << BITRES);
         } else {
            over = imax(0, have - floor_bits);
            bits[j] = have - over;
            fine = 0; fine_priority[j] = 1;
         }
         if (over > 0) {                                                                                        /* surplus first buys fine bits here */
            const int more = imin(over >> (stereo + BITRES), OA_MAX_FINE_BITS - fine);
            fine += more;
            const int spent = more * C << BITRES;
            fine_priority[j] = spent >= over - roll;
            over -= spent;
         }
         ebits[j] = fine;
         roll = over;
      }
      *balance_out = roll;
      for (; j < end; j++) { ebits[j] = bits[j] >> stereo >> BITRES; bits[j] = 0; fine_priority[j] = ebits[j] < 1; }   /* skipped bands: fine energy only */
   }
   return coded;
}
#undef OA_PROBE_SUM
#endif

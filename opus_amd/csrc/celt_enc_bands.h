/* celt_enc_bands.h — spectral analysis and side-information coding of the CELT frame encoder.
 * Reference: celt/celt_encoder.c (:473 patch_transient_decision, :511 compute_mdcts, :650 l1_metric, :663 tf_analysis,
 * :823 tf_encode, :865 alloc_trim_analysis, :957 stereo_analysis, :1049 dynalloc_analysis, :1605 compute_vbr),
 * celt/bands.c (:95 compute_band_energies, :125 normalise_bands, :470 spreading_decision), celt/quant_bands.c:553 amp2Log2.
 * Mapping: per-band work = one lane per (band, channel); per-bin work = lane-strided; budget/entropy logic = lane 0. */
#ifndef OPUS_AMD_CELT_ENC_BANDS_H
#define OPUS_AMD_CELT_ENC_BANDS_H

WV_TABLE signed char k_tf_select_table[4][8] = {
   {0, -1, 0, -1, 0, -1, 0, -1}, {0, -1, 0, -2, 1, 0, 1, -1}, {0, -2, 0, -3, 2, 0, 1, -1}, {0, -2, 0, -3, 3, 0, 1, -1}};
WV_TABLE u8 k_trim_icdf[11] = {126, 124, 119, 109, 87, 41, 19, 9, 4, 2, 0};
WV_TABLE u8 k_spread_icdf[4] = {25, 23, 2, 0};
WV_TABLE u8 k_tapset_icdf[3] = {2, 1, 0};

WV_DEV i32 inner_prod_norm_shift_l(const WV_LDS i32 *x, const WV_LDS i32 *y, int len)   /* one lane, serial */
{
   i64 sum = 0;
   for (int i = 0; i < len; i++) sum += x[i] * (i64)y[i];
   return (i32)(sum >> 2 * (NORM_SHIFT - 14));
}
WV_DEV i32 inner_prod_norm_shift_w(const WV_LDS i32 *x, const WV_LDS i32 *y, int len)   /* whole wave */
{
   i64 sum = 0;
   FOR_LANES(i, len) sum += x[i] * (i64)y[i];
   return (i32)(wv_sum64(sum) >> 2 * (NORM_SHIFT - 14));
}

WV_DEV i32 inner_prod_norm_shift_gw(const i32 *x, const WV_LDS i32 *y, int len)         /* whole wave, x in HBM */
{
   i64 sum = 0;
   FOR_LANES(i, len) sum += x[i] * (i64)y[i];
   return (i32)(wv_sum64(sum) >> 2 * (NORM_SHIFT - 14));
}

/* compute_band_energies + amp2Log2 of the channel resident in W.  A band is cut into chunks of two base coefficients (2 << LM bins): 54 chunks, one lane each, so the
 * widest band (22 base coefficients) is eleven lanes' work instead of one lane's 176 bins twice over.  The band's range is the maximum of its chunks' ranges, its
 * energy sum the (mod-2^32, order-free) sum of the chunks' sums; the first lane of a band finishes it.  Partials in L->scr [64]. */
WV_TABLE u8 k_be_band[54] = {0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 12, 13, 13, 14, 14, 15, 15, 15, 16, 16, 16, 17, 17, 17, 17, 18, 18, 18, 18, 18, 18,
   19, 19, 19, 19, 19, 19, 19, 19, 19, 20, 20, 20, 20, 20, 20, 20, 20, 20, 20, 20};
WV_TABLE u8 k_be_first[22] = {0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 14, 16, 18, 21, 24, 28, 34, 43, 54};
WV_DEV void band_energies_channel(WV_LDS FrameLds *L, const WV_LDS i32 *W, int c, WV_LDS i32 *bandLogE_out)
{
   const int LM = L->sh.LM, end = L->sh.end, effEnd = L->sh.effEnd, lane = wv_lane();
   WV_LDS i32 *part = L->scr;
   int b = NBE, first = 0, last = 0, n = 0;
   const WV_LDS i32 *x = W;
   if (lane < 54) {
      b = k_be_band[lane]; first = k_be_first[b]; last = k_be_first[b + 1];
      const int k = lane - first, w = ct_eBands[b + 1] - ct_eBands[b];
      x = &W[(ct_eBands[b] + 2 * k) << LM];
      n = b < effEnd ? imin(2, w - 2 * k) << LM : 0;
   }
   i32 mx = 0, mn = 0;
   for (int j = 0; j < n; j++) { mx = imax(mx, x[j]); mn = imin(mn, x[j]); }
   part[lane] = imax(mx, neg32(mn));
   wv_sync();
   i32 maxval = 0, sum = 0; int shift = 0;
   for (int t = first; t < last; t++) maxval = imax(maxval, part[t]);
   if (maxval > 0) {
      shift = imax(0, 30 - celt_ilog2(maxval + (maxval >> 14) + 1) - ((((ct_logN[b] + 7) >> BITRES) + LM + 1) >> 1));
      for (int j = 0; j < n; j++) { i32 v = shl32(x[j], shift); sum = add32(sum, mult32_32_q31(v, v)); }
   }
   wv_sync();
   part[lane] = sum;
   wv_sync();
   if (lane < 54 && lane == first && b < effEnd) {
      i32 E;
      if (maxval > 0) {
         i32 tot = 0;
         for (int t = first; t < last; t++) tot = add32(tot, part[t]);
         E = imax(maxval, pshr32(fx_sqrt32(tot >> 1), shift));
      } else E = EPSILON;
      L->bandE[b + c * NBE] = E;
      bandLogE_out[b + c * NBE] = fx_log2_db(E) - shl32((i32)ct_eMeans[b], DB_SHIFT - 4) + GC(2.f);
   }
   FOR_LANES(i, NBE) { if (i >= effEnd && i < end) bandLogE_out[i + c * NBE] = -GC(14.f); }
   wv_sync();
}

/* compute_mdcts (celt_encoder.c:511) + compute_band_energies (bands.c:95) + amp2Log2: one channel at a time through the LDS work buffer W -- B interleaved
 * transforms in place, the band energies while the channel is resident, then the channel goes out to the HBM spectrum g->X */
/* normalise: the channel leaves LDS normalised (normalise_bands, bands.c:125, with the energies just taken) -- the spectrum is then written once instead of written,
 * read and written again; the caller passes 0 when the energies can still change before the normalisation (LFE) or the transform is not the frame's last word.
 * xcut (with normalise): the frame will be cut in front of its bands -- the coded bins also go to the continuation record's X[2][OA_CODED_BINS] from the same registers */
WV_DEVN void compute_mdcts_wave(WV_LDS FrameLds *L, const OaEncState *gst, int shortBlocks, WV_LDS i32 *bandLogE_out, int normalise = 0, i32 *xcut = nullptr)
{
   const int C = L->sh.C, CC = L->sh.CC, LM = L->sh.LM;
   CeltScratch *G = L->g;
   WV_LDS i32 *W = L->BC.W;
   int B, N, shift;
   if (shortBlocks) { B = shortBlocks; N = 120; shift = 3; }
   else { B = 1; N = 120 << LM; shift = 3 - LM; }
   const int up = L->sh.upsample > 1 ? L->sh.upsample : 1, bound = B * N / up;
   AN_TIC();
   for (int c = 0; c < CC; c++) {
      mdct_forward_blocks(gst->in_mem + c * OA_OVERLAP, G->in[c], W, shift, B, L->aux);
      AN_TOC(19);
      if (CC == 2 && C == 1) {                                   /* stereo input coded as mono: the downmix of the two spectra */
         if (c == 0) { FOR_LANES(i, B * N) G->X[i] = W[i]; wv_sync(); continue; }
         FOR_LANES(i, B * N) W[i] = add32(G->X[i] >> 1, W[i] >> 1);
         wv_sync();
      }
      if (up > 1) {                    /* zero-stuffed input: restore the level below the input's Nyquist, nothing above it (celt_encoder.c:544-554) */
         FOR_LANES(i, B * N) W[i] = i < bound ? W[i] * up : 0;
         wv_sync();
      }
      const int cc = C == 1 ? 0 : c;
      band_energies_channel(L, W, cc, bandLogE_out);
      AN_TOC(20);
      if (normalise) {
         const int effEnd = L->sh.effEnd, nb = ct_eBands[effEnd] << LM;
         FOR_LANES(i, NBE) {                                       /* (band_energies_channel ends on a barrier) */
            if (i < effEnd) {
               i32 E = L->bandE[i + cc * NBE];
               if (E < 10) E += EPSILON;
               const int sh = 30 - celt_zlog2(E);
               L->scr[i] = fx_rcp_norm32(shl32(E, sh)); L->scr[NBE + i] = sh;
            }
         }
         wv_sync();
         FOR_LANES(i, B * N) {
            i32 v = W[i];
            if (i < nb) { const int bnd = ct_band_of[i >> LM]; v = pshr32(mult32_32_q31(L->scr[bnd], shl32(v, L->scr[NBE + bnd])), 30 - NORM_SHIFT); }
            G->X[cc * B * N + i] = v;
            if (xcut && i < nb) xcut[cc * OA_CODED_BINS + i] = v;
         }
      } else { FOR_LANES(i, B * N) G->X[cc * B * N + i] = W[i]; }
      wv_sync();
      AN_TOC(21);
   }
}

/* normalise_bands (bands.c:125), in place in the HBM spectrum: freq -> X */
WV_DEVN void normalise_bands_wave(WV_LDS FrameLds *L)
{
   const int C = L->sh.C, M = L->sh.M, N = L->sh.N, end = L->sh.effEnd;
   i32 *X = L->g->X;
   for (int c = 0; c < C; c++)
      for (int i = 0; i < end; i++) {
         i32 E = L->bandE[i + c * NBE];
         if (E < 10) E += EPSILON;
         int shift = 30 - celt_zlog2(E);
         E = shl32(E, shift);
         i32 g = fx_rcp_norm32(E);
         int lo = M * ct_eBands[i], hi = M * ct_eBands[i + 1];
         for (int j = lo + wv_lane(); j < hi; j += WV_WIDTH)
            X[j + c * N] = pshr32(mult32_32_q31(g, shl32(X[j + c * N], shift)), 30 - NORM_SHIFT);
      }
   wv_sync();
}

/* the normalised coded bins of both channels -> LDS (BC.xs) for the analyses that walk bands serially per lane */
WV_DEV void stage_coded_bins_wave(WV_LDS FrameLds *L)
{
   const int C = L->sh.C, N = L->sh.N, n = L->sh.M * ct_eBands[L->sh.effEnd];
   const i32 *X = L->g->X;
   wv_sync();
   for (int c = 0; c < C; c++)
      for (int j0 = wv_lane(); j0 < n; j0 += 7 * WV_WIDTH) {       /* seven trips' bins in flight (800 bins: two batches) */
         i32 v[7];
#pragma unroll
         for (int u = 0; u < 7; u++) v[u] = X[c * N + imin(j0 + u * WV_WIDTH, n - 1)];
#pragma unroll
         for (int u = 0; u < 7; u++) { const int j = j0 + u * WV_WIDTH; if (j < n) L->BC.xs[c][j] = v[u]; }
      }
   wv_sync();
}

WV_DEV i32 oa_med3(i32 a, i32 b, i32 c) { return imax(imin(a, b), imin(imax(a, b), c)); }
WV_DEV i32 oa_med5(i32 a, i32 b, i32 c, i32 d, i32 e)
{
   /* median of five = median of (the larger of the two pair minima, the smaller of the two pair maxima, the odd one) */
   const i32 lo1 = imin(a, b), hi1 = imax(a, b), lo2 = imin(d, e), hi2 = imax(d, e);
   return oa_med3(imax(lo1, lo2), imin(hi1, hi2), c);
}
/* v[i] <- op over j <= i (UP) or j >= i (!UP) of v[j] + step * |i - j| inside each 32-lane half, sources limited to bands [lo, hi] */
template <bool UP, bool MIN> WV_DEV i32 oa_band_scan(i32 v, int band, int lo, int hi, i32 step)
{
   const int lane = wv_lane();
#pragma unroll
   for (int d = 1; d < 32; d <<= 1) {
      const int src = UP ? band - d : band + d;
      const i32 t = wv_shfl(v, (lane + (UP ? -d : d)) & 63) + d * step;
      if (src >= lo && src <= hi && band >= lo && band <= hi) v = MIN ? imin(v, t) : imax(v, t);
   }
   return v;
}

/* temporal VBR follower (celt_encoder.c:2196-2213): the loudest channel's band energies followed downwards by at most 1 dB per band from a -10 dB start --
 * a (max,+) prefix scan with the start value folded in as "a source one band below `start`" -- then the mean of the follower over the coded bands */
WV_DEV void temporal_vbr_wave(WV_LDS FrameLds *L)
{
   const int C = L->sh.C, start = L->sh.start, end = L->sh.end, LM = L->sh.LM;
   const int lane = wv_lane(), band = lane & 31, ib = imin(band, NBE - 1);
   const i32 offset = L->sh.shortBlocks ? half32(shl32(LM, DB_SHIFT - 5)) : 0, one = QC32(1.0f, DB_SHIFT - 5);
   i32 e = (L->bandLogE[ib] >> 5) - offset;
   if (C == 2) e = imax(e, (L->bandLogE[ib + NBE] >> 5) - offset);
   i32 follow = oa_band_scan<true, false>(e, band, start, end - 1, -one);
   follow = imax(follow, -QC32(10.0f, DB_SHIFT - 5) - (band - start + 1) * one);
   const i32 total = wv_sum(lane >= start && lane < end ? follow : 0);
   LANE0 {
      i32 tv = sub32(shl32(total / (end - start), 5), L->st.spec_avg);
      tv = imin(GC(3.f), imax(-GC(1.5f), tv));
      L->st.spec_avg += mult16_32_q15(QC16(.02f, 15), tv);
      L->sh.temporal_vbr = tv;
   }
}

/* patch_transient_decision (celt_encoder.c:473): last frame's energies (louder channel) spread 1 dB per band both ways -- two (max,+) scans -- against this
 * frame's: a large mean increase means the transient detector missed an onset */
WV_DEV int patch_transient_decision_wave(WV_LDS FrameLds *L)
{
   const int C = L->sh.C, start = L->sh.start, end = L->sh.end;
   const int lane = wv_lane(), band = lane & 31, ch = lane >> 5, ib = imin(band, NBE - 1);
   i32 old = L->oldBandE[ib];
   if (C == 2) old = imax(old, L->oldBandE[ib + NBE]);
   i32 spread = oa_band_scan<true, false>(old, band, start, end - 1, -GC(1.0f));
   spread = oa_band_scan<false, false>(spread, band, start, end - 1, -GC(1.0f));
   i32 diff = 0;
   if (ch < C && band >= imax(2, start) && band < end - 1) {
      const i16 x1 = (i16)imax(0, L->bandLogE[ib + ch * NBE]), x2 = (i16)imax(0, spread);
      diff = imax(0, sub32(x1, x2));
   }
   const i32 mean_diff = wv_sum(diff) / (C * (end - 1 - imax(2, start)));
   return mean_diff > GC(1.f);
}

/* ---- dynalloc_analysis (celt_encoder.c:1049) on the wave ----
 * One lane per (band, channel): lane = 32 * channel + band.  Everything the reference chains along the bands is a recurrence of the form
 *       f[i] = op(f[i-1] + step, g[i])            op = min or max, step a constant
 * whose closed form is op over j of (g[j] + step * |i - j|): a prefix (suffix) scan in the (op, +) semiring.  The values are Q24 integers far from overflow, the
 * operations are adds and compares only, so the scan -- five shuffle steps for 21 bands -- gives the recurrence's result exactly.  The spreading mask, the two
 * energy followers (up 1.5 dB / band, down 2 dB / band below the last upward step), the median floors (neighbours fetched from LDS) and the stereo coupling are
 * done that way; the boost loop's running total with its cap is an inclusive sum scan plus a ballot for the first band that hits the cap. */
WV_DEVN void dynalloc_analysis_wave(WV_LDS FrameLds *L, const OaAnalysisInfo *an)
{
   WV_LDS FrameShared *sh = &L->sh;
   const int start = sh->start, end = sh->end, C = sh->C, LM = sh->LM, isTransient = sh->isTransient;
   const int vbr = sh->vbr, constrained_vbr = sh->constrained_vbr, effectiveBytes = sh->effectiveBytes;
   const int lane = wv_lane(), band = lane & 31, ch = lane >> 5;
   const bool in_band = band < end, mine = in_band && ch < C, lead = in_band && ch == 0;      /* lead: the lane that owns the band's per-band results */
   const int w = imin(ch, C - 1) * NBE + imin(band, NBE - 1);
   WV_LDS i32 *const b3 = L->BC.tf;                /* 2 x 32 words: the (patched) long-window band energies, for the median neighbourhoods */
   FOR_LANES(i, NBE) L->offsets[i] = 0;
   const int ib = imin(band, NBE - 1);
   const i32 noise_floor = GC(0.0625f) * ct_logN[ib] + GC(.5f) + shl32(9 - sh->lsb_depth, DB_SHIFT) - shl32(ct_eMeans[ib], DB_SHIFT - 4) + GC(.0062f) * (ib + 5) * (ib + 5);
   const i32 E = L->bandLogE[w];
   const i32 maxDepth = imax(-GC(31.9f), wv_max(mine ? E - noise_floor : -GC(31.9f)));
   {  /* masking spread for the spreading decision's band weights: 2 dB / band upwards, 3 dB / band downwards */
      i32 sig = L->bandLogE[ib] - noise_floor;
      if (C == 2) sig = imax(sig, L->bandLogE[NBE + ib] - noise_floor);
      i32 mask = oa_band_scan<true, false>(sig, band, 0, end - 1, -GC(2.f));
      mask = oa_band_scan<false, false>(mask, band, 0, end - 1, -GC(3.f));
      const i32 smr = sig - imax(imax(0, maxDepth - GC(12.f)), mask);
      if (lead) L->spread_weight[band] = 32 >> -pshr32(imax(-GC(5.f), imin(0, smr)), DB_SHIFT);
   }
   i32 tot_boost = 0;
   if (effectiveBytes >= (30 + 5 * LM) && !sh->lfe) {
      i32 e3 = L->bandLogE2[w];
      if (LM == 0 && band < 8) e3 = imax(e3, L->oldBandE[w]);
      b3[lane] = e3;
      wv_sync();
      /* the last band that steps up by more than half a dB over its lower neighbour; a channel without one inherits the previous channel's */
      const u64 up = wv_ballot(mine && band >= 1 && e3 > b3[lane - (band >= 1)] + GC(.5f));
      const u32 up0 = (u32)up, up1 = (u32)(up >> 32);
      const int last0 = up0 ? 31 - __builtin_clz(up0) : 0, last1 = up1 ? 31 - __builtin_clz(up1) : last0;
      const int last = ch ? last1 : last0;
      i32 f = oa_band_scan<true, true>(e3, band, 0, end - 1, GC(1.5f));
      f = oa_band_scan<false, true>(f, band, 0, last, GC(2.f));
      if (mine) {
         const int base = lane - band;
         if (band >= 2 && band < end - 2) f = imax(f, oa_med5(b3[lane - 2], b3[lane - 1], e3, b3[lane + 1], b3[lane + 2]) - GC(1.f));
         if (band < 2) f = imax(f, oa_med3(b3[base], b3[base + 1], b3[base + 2]) - GC(1.f));
         if (band >= end - 2) f = imax(f, oa_med3(b3[base + end - 3], b3[base + end - 2], b3[base + end - 1]) - GC(1.f));
         f = imax(f, noise_floor);
      }
      /* from here on one value per band, on the channel-0 lane */
      if (C == 2) {
         i32 f1 = wv_shfl(f, band + 32);
         f1 = imax(f1, f - GC(4.f));
         f = imax(f, f1 - GC(4.f));
         f = half32(imax(0, E - f) + imax(0, L->bandLogE[NBE + ib] - f1));
      } else f = imax(0, E - f);
      const bool coded = lead && band >= start;
      f = imax(f, L->surround_dynalloc[ib]);
      if (coded) L->importance[band] = pshr32(13 * fx_exp2_db(imin(f, GC(4.f))), 16);
      if ((!vbr || constrained_vbr) && !isTransient) f = half32(f);
      if (band < 8) f *= 2;
      if (band >= 12) f = half32(f);
      if (sh->toneishness > QC32(.98f, 29)) {
         const int freq_bin = pshr32((i32)(i16)sh->tone_freq * QC16(120 / 3.14159265358979323846, 9), 13 + 9);
         const int lo = ct_eBands[ib], hi = ct_eBands[ib + 1];
         if (freq_bin >= lo && freq_bin <= hi) f += GC(2.f);
         if (freq_bin >= lo - 1 && freq_bin <= hi + 1) f += GC(1.f);
         if (freq_bin >= lo - 2 && freq_bin <= hi + 2) f += GC(1.f);
         if (freq_bin >= lo - 3 && freq_bin <= hi + 3) f += GC(.5f);
         if (freq_bin >= ct_eBands[end]) { if (band == end - 1) f += GC(2.f); if (band == end - 2) f += GC(1.f); }
      }
      if (an->valid && band >= start && band < imin(AN_LEAK_BANDS, end)) f += GC(1.f / 64.f) * (i32)an->leak_boost[band];        /* the analysis' leakage boosts (:1226-1230) */
      if (effectiveBytes > 320 && band == 0) f += imin(GC(1.5f), GC(1e-3f) * (effectiveBytes - 320));
      /* boosts, their cost, and the running total with its cap (CBR / constrained VBR: at most two thirds of the frame) */
      f = imin(f, GC(4)) >> 8;
      const int width = C * (ct_eBands[ib + 1] - ct_eBands[ib]) << LM;
      int boost, boost_bits;
      if (width < 6) { boost = (int)(f >> (DB_SHIFT - 8)); boost_bits = boost * width << BITRES; }
      else if (width > 48) { boost = (int)((f * 8) >> (DB_SHIFT - 8)); boost_bits = (boost * width << BITRES) / 8; }
      else { boost = (int)((f * width / 6) >> (DB_SHIFT - 8)); boost_bits = boost * 6 << BITRES; }
      if (!coded) boost_bits = 0;
      const i32 upto = wv_scan_incl(boost_bits);                                   /* lanes run in band order: channel-1 lanes add nothing */
      const bool capped = !vbr || (constrained_vbr && !isTransient);
      const u64 hit = wv_ballot(coded && capped && (upto >> BITRES >> 3) > 2 * effectiveBytes / 3);
      const int stop = hit ? __builtin_ctzll(hit) : 64;                            /* the first band whose boost would pass the cap takes what is left */
      const i32 cap = (2 * effectiveBytes / 3) << BITRES << 3;
      if (coded && band < stop) L->offsets[band] = boost;
      if (coded && band == stop) L->offsets[band] = cap - (upto - boost_bits);
      tot_boost = hit ? cap : wv_bcast(upto, 31);
   } else if (lead && band >= start) L->importance[band] = 13;
   LANE0 { sh->tot_boost = tot_boost; sh->maxDepth = maxDepth; }
   wv_sync();
}

/* ---- tf_analysis (celt_encoder.c:663) ----
 * Metric: per band, the L1 norm of the (normalised) spectrum after 0 .. LM+1 levels of Haar mixing picks the time-frequency resolution the band likes best.
 * The spectrum is cut into `units` of 2^LM bins (a band is 1 .. 22 units; 100 units in all): the butterflies of the levels below LM stay inside a unit, the one of
 * level LM joins the even-th unit of a band with its neighbour.  One lane per unit does its butterflies and its |.| sum in the same pass -- the "-1" trial of a
 * transient frame needs the sums only, so nothing is stored for it -- and one lane per band adds its units up and keeps the best level.
 * Search: the two-state Viterbi over the bands is a chain of 2x2 (min,+) matrix products, so the running costs of every band come out of one prefix scan (exact:
 * integer adds and minima); the decisions follow from the costs one band down, and the backtrace -- each band maps the state above it to its own -- is a suffix
 * scan of those two-bit maps under composition.  Both tf_select hypotheses run side by side, one per half of the wave. */
WV_TABLE u8 ct_unit2band[100] = {0, 1, 2, 3, 4, 5, 6, 7, 8, 8, 9, 9, 10, 10, 11, 11, 12, 12, 12, 12, 13, 13, 13, 13, 14, 14, 14, 14, 15, 15, 15, 15, 15, 15, 16, 16, 16, 16, 16, 16, 17, 17, 17, 17, 17, 17, 17, 17, 18, 18, 18, 18, 18, 18, 18, 18, 18, 18, 18, 18, 19, 19, 19, 19, 19, 19, 19, 19, 19, 19, 19, 19, 19, 19, 19, 19, 19, 19, 20, 20, 20, 20, 20, 20, 20, 20, 20, 20, 20, 20, 20, 20, 20, 20, 20, 20, 20, 20, 20, 20};

WV_DEV i32 tf_band_cost(WV_LDS const i32 *usum, int band_lo, int band_hi, int level_bias)
{
   i32 L1 = 0;
   for (int u = band_lo; u < band_hi; u++) L1 += usum[u];
   return mac16_32_q15(L1, level_bias, L1);
}
WV_DEVN void tf_analysis_wave(WV_LDS FrameLds *L, int lambda)
{
   WV_LDS FrameShared *sh = &L->sh;
   const int len = sh->effEnd, isTransient = sh->isTransient, LM = sh->LM, N0 = sh->N, tf_chan = sh->tf_chan, lane = wv_lane();
   const i16 tf_estimate = (i16)sh->tf_estimate;
   WV_LDS i32 *metric = L->scr;
   WV_LDS i32 *tmp = L->BC.tf;                       /* the spectrum being mixed, laid out like X */
   WV_LDS i32 *usum = L->BC.tf + OA_CODED_BINS;      /* one |.| sum per unit */
   const i32 *X = L->g->X + tf_chan * N0;
   const i16 bias = (i16)mult16_16_q14(QC16(.04f, 15), imax(-QC16(.25f, 14), QC16(.5f, 14) - tf_estimate));
   const int units = ct_eBands[len], U = 1 << LM, sh14 = NORM_SHIFT - 14;
   wv_sync();
   {  /* seven trips' bins in flight */
      const int n = units << LM;
      for (int j0 = lane; j0 < n; j0 += 7 * WV_WIDTH) {
         i32 v[7];
#pragma unroll
         for (int u = 0; u < 7; u++) v[u] = X[imin(j0 + u * WV_WIDTH, n - 1)];
#pragma unroll
         for (int u = 0; u < 7; u++) { const int j = j0 + u * WV_WIDTH; if (j < n) tmp[j] = v[u]; }
      }
   }
   wv_sync();
   /* this lane's band (lanes >= len idle in the per-band steps) */
   const int b = imin(lane, len - 1), b_lo = ct_eBands[b], b_hi = ct_eBands[b + 1], narrow = b_hi - b_lo == 1;
   const int levels = LM + !(isTransient || narrow);
   FOR_LANES(u, units) { i32 s = 0; for (int i = 0; i < U; i++) s += iabs(tmp[(u << LM) + i] >> sh14); usum[u] = s; }
   wv_sync();
   i32 best_L1 = tf_band_cost(usum, b_lo, b_hi, (isTransient ? LM : 0) * bias);
   int best_level = 0;
   wv_sync();
   if (isTransient) {                                /* one level coarser in time than the short blocks: sums only */
      FOR_LANES(u, units) {
         const int bu = ct_unit2band[u], rel = u - ct_eBands[bu];
         i32 s = 0;
         if (!(rel & 1) && ct_eBands[bu + 1] - ct_eBands[bu] > 1)
            for (int i = 0; i < U; i++) {
               const i32 t1 = mult32_32_q31(QC32(.70710678f, 31), tmp[(u << LM) + i]), t2 = mult32_32_q31(QC32(.70710678f, 31), tmp[((u + 1) << LM) + i]);
               s += iabs(add32(t1, t2) >> sh14) + iabs(sub32(t1, t2) >> sh14);
            }
         usum[u] = s;
      }
      wv_sync();
      const i32 L1 = tf_band_cost(usum, b_lo, b_hi, (LM + 1) * bias);
      if (!narrow && L1 < best_L1) { best_L1 = L1; best_level = -1; }
      wv_sync();
   }
   for (int k = 0; k < LM + !isTransient; k++) {
      if (k < LM) {
         const int stride = 1 << k;
         FOR_LANES(u, units) {
            WV_LDS i32 *x = tmp + (u << LM);
            i32 s = 0;
            for (int j = 0; j < U >> 1; j++) {
               const int p = ((j >> k) << (k + 1)) | (j & (stride - 1));
               const i32 t1 = mult32_32_q31(QC32(.70710678f, 31), x[p]), t2 = mult32_32_q31(QC32(.70710678f, 31), x[p + stride]);
               const i32 a = add32(t1, t2), d = sub32(t1, t2);
               x[p] = a; x[p + stride] = d;
               s += iabs(a >> sh14) + iabs(d >> sh14);
            }
            usum[u] = s;
         }
      } else {
         FOR_LANES(u, units) {
            const int bu = ct_unit2band[u], rel = u - ct_eBands[bu];
            i32 s = 0;
            if (!(rel & 1) && ct_eBands[bu + 1] - ct_eBands[bu] > 1) {
               WV_LDS i32 *x = tmp + (u << LM);
               for (int i = 0; i < U; i++) {
                  const i32 t1 = mult32_32_q31(QC32(.70710678f, 31), x[i]), t2 = mult32_32_q31(QC32(.70710678f, 31), x[U + i]);
                  const i32 a = add32(t1, t2), d = sub32(t1, t2);
                  x[i] = a; x[U + i] = d;
                  s += iabs(a >> sh14) + iabs(d >> sh14);
               }
            }
            usum[u] = s;
         }
      }
      wv_sync();
      const i32 L1 = tf_band_cost(usum, b_lo, b_hi, (isTransient ? LM - k - 1 : k + 1) * bias);
      if (k < levels && L1 < best_L1) { best_L1 = L1; best_level = k + 1; }
      wv_sync();
   }
   {
      int m = isTransient ? 2 * best_level : -2 * best_level;
      if (narrow && (m == 0 || m == -2 * LM)) m -= 1;
      if (lane < len) metric[lane] = m;
   }
   wv_sync();
   /* ---- the search: hypothesis tf_select = half of the wave, band = lane within the half ---- */
   const int sel = lane >> 5, i = lane & 31, ib = imin(i, len - 1);
   const signed char *row = k_tf_select_table[LM] + 4 * isTransient + 2 * sel;
   const i32 d0 = L->importance[ib] * iabs(metric[ib] - 2 * row[0]), d1 = L->importance[ib] * iabs(metric[ib] - 2 * row[1]);
   const i32 FAR = 1 << 28;
   /* band i's step as a (min,+) matrix a[from][to]; band 0 is the start vector in row 0 */
   i32 a00, a01, a10, a11;
   if (i == 0) { a00 = d0; a01 = d1 + (isTransient ? 0 : lambda); a10 = FAR; a11 = FAR; }
   else { a00 = d0; a01 = d1 + lambda; a10 = d0 + lambda; a11 = d1; }
#pragma unroll
   for (int d = 1; d < 32; d <<= 1) {
      const int src = (lane - d) & 63;
      const i32 b00 = wv_shfl(a00, src), b01 = wv_shfl(a01, src), b10 = wv_shfl(a10, src), b11 = wv_shfl(a11, src);
      if (i >= d) {                                  /* (bands i-2d+1 .. i-d) x (bands i-d+1 .. i) */
         const i32 n00 = imin(b00 + a00, b01 + a10), n01 = imin(b00 + a01, b01 + a11), n10 = imin(b10 + a00, b11 + a10), n11 = imin(b10 + a01, b11 + a11);
         a00 = imin(n00, FAR); a01 = imin(n01, FAR); a10 = imin(n10, FAR); a11 = imin(n11, FAR);
      }
   }
   const i32 cost0 = a00, cost1 = a01;               /* the costs of ending band i in state 0 / 1 */
   const i32 end0 = wv_shfl(cost0, (lane & 32) | (len - 1)), end1 = wv_shfl(cost1, (lane & 32) | (len - 1));
   const i32 selcost = imin(end0, end1);
   const int tf_select = isTransient && wv_bcast(selcost, 32) < wv_bcast(selcost, 0);
   /* where state 0 / state 1 of band i came from, as a map "state of band i -> state of band i-1" in two bits; then k = the map of the band above */
   const i32 p0 = wv_shfl(cost0, (lane - 1) & 63), p1 = wv_shfl(cost1, (lane - 1) & 63);
   const int from = (p0 < p1 + lambda ? 0 : 1) | (p0 + lambda < p1 ? 0 : 2);
   int up = wv_shfl(from, (lane + 1) & 63);
   if (i >= len - 1) up = 2;                         /* identity above the last band */
#pragma unroll
   for (int d = 1; d < 32; d <<= 1) {
      int far = wv_shfl(up, (lane + d) & 63);
      if (i + d >= 32) far = 2;
      up = ((up >> (far & 1)) & 1) | (((up >> ((far >> 1) & 1)) & 1) << 1);     /* up o far */
   }
   const int last_state = end0 < end1 ? 0 : 1, state = (up >> last_state) & 1;
   if (sel == tf_select) {
      if (i < len) L->tf_res[i] = state;
      else if (i < sh->end) L->tf_res[i] = last_state;
      if (lane == 32 * tf_select) sh->tf_select = tf_select;
   }
   wv_sync();
}

/* lane 0: tf_encode (celt_encoder.c:823) */
WV_DEV void tf_encode_l0(WV_LDS FrameLds *L, EC_ARGS)
{
   const int start = L->sh.start, end = L->sh.end, isTransient = L->sh.isTransient, LM = L->sh.LM;
   int tf_select = L->sh.tf_select;
   WV_LDS i32 *tf_res = L->tf_res;
   u32 budget = e->storage * 8, tell = k_ec_tell(EC_PASS);
   int logp = isTransient ? 2 : 4, curr = 0, tf_changed = 0;
   int tf_select_rsv = LM > 0 && tell + logp + 1 <= budget;
   budget -= tf_select_rsv;
   for (int i = start; i < end; i++) {
      if (tell + logp <= budget) {
         k_ec_enc_bit_logp(EC_PASS, tf_res[i] ^ curr, logp);
         tell = k_ec_tell(EC_PASS);
         curr = tf_res[i];
         tf_changed |= curr;
      } else tf_res[i] = curr;
      logp = isTransient ? 4 : 5;
   }
   if (tf_select_rsv && k_tf_select_table[LM][4 * isTransient + 0 + tf_changed] != k_tf_select_table[LM][4 * isTransient + 2 + tf_changed])
      k_ec_enc_bit_logp(EC_PASS, tf_select, 1);
   else tf_select = 0;
   for (int i = start; i < end; i++) tf_res[i] = k_tf_select_table[LM][4 * isTransient + 2 * tf_select + tf_res[i]];
}

/* spreading_decision (bands.c:470): histogram per (band, channel) lane, combination on lane 0 */
WV_DEVN void spreading_decision_wave(WV_LDS FrameLds *L, int update_hf)
{
   WV_LDS FrameShared *sh = &L->sh;
   WV_LDS OaEncScalars *st = &L->st;
   const int end = sh->effEnd, C = sh->C, M = sh->M, N0 = OA_CODED_BINS;
   const WV_LDS i32 *X = L->BC.xs[0];       /* staged by stage_coded_bins_wave */
   WV_LDS i32 *cnt = L->scr;      /* [2*21][2]: tmp, hf contribution */
   if (M * (ct_eBands[end] - ct_eBands[end - 1]) <= 8) { wv_sync(); LANE0 st->spread_decision = 0; wv_sync(); return; }
   FOR_LANES(w, C * NBE) {
      int c = w / NBE, i = w - c * NBE;
      int tmpv = -1, hf = 0;
      if (i < end) {
         int N = M * (ct_eBands[i + 1] - ct_eBands[i]);
         if (N > 8) {
            const WV_LDS i32 *x = X + M * ct_eBands[i] + c * N0;
            int t0 = 0, t1 = 0, t2 = 0;
            for (int j = 0; j < N; j++) {
               i32 x2N = mult16_16(mult16_16_q15(x[j] >> (NORM_SHIFT - 14), x[j] >> (NORM_SHIFT - 14)), N);
               if (x2N < QC16(0.25f, 13)) t0++;
               if (x2N < QC16(0.0625f, 13)) t1++;
               if (x2N < QC16(0.015625f, 13)) t2++;
            }
            if (i > NBE - 4) hf = (u32)(32 * (t1 + t0)) / (u32)N;
            tmpv = (2 * t2 >= N) + (2 * t1 >= N) + (2 * t0 >= N);
         }
      }
      cnt[2 * w] = tmpv; cnt[2 * w + 1] = hf;
   }
   wv_sync();
   LANE0 {
      int sum = 0, nbBands = 0, hf_sum = 0, decision;
      for (int c = 0; c < C; c++)
         for (int i = 0; i < end; i++) {
            int w = c * NBE + i;
            if (cnt[2 * w] < 0) continue;
            hf_sum += cnt[2 * w + 1];
            sum += cnt[2 * w] * L->spread_weight[i];
            nbBands += L->spread_weight[i];
         }
      if (update_hf) {
         if (hf_sum) hf_sum = (u32)hf_sum / (u32)(C * (4 - NBE + end));
         st->hf_average = (st->hf_average + hf_sum) >> 1;
         hf_sum = st->hf_average;
         if (st->tapset_decision == 2) hf_sum += 4;
         else if (st->tapset_decision == 0) hf_sum -= 4;
         if (hf_sum > 22) st->tapset_decision = 2;
         else if (hf_sum > 18) st->tapset_decision = 1;
         else st->tapset_decision = 0;
      }
      sum = (u32)((i32)sum << 8) / (u32)nbBands;
      sum = (sum + st->tonal_average) >> 1;
      st->tonal_average = sum;
      sum = (3 * sum + (((3 - st->spread_decision) << 7) + 64) + 2) >> 2;
      if (sum < 80) decision = 3;
      else if (sum < 256) decision = 2;
      else if (sum < 384) decision = 1;
      else decision = 0;
      st->spread_decision = decision;
   }
   wv_sync();
}

/* stereo_analysis (celt_encoder.c:957): two L1 sums by wave reduction, decision identical on every lane */
WV_DEV int stereo_analysis_wave(WV_LDS FrameLds *L)
{
   const int LM = L->sh.LM, N0 = OA_CODED_BINS;
   const WV_LDS i32 *X = L->BC.xs[0];
   i32 sLR = 0, sMS = 0;
   FOR_LANES(j, ct_eBands[13] << LM) {
      i32 Lv = X[j] >> (NORM_SHIFT - 14), R = X[N0 + j] >> (NORM_SHIFT - 14), Mv = add32(Lv, R), S = sub32(Lv, R);
      sLR = add32(sLR, add32(iabs(Lv), iabs(R)));
      sMS = add32(sMS, add32(iabs(Mv), iabs(S)));
   }
   i32 sumLR = add32(EPSILON, wv_sum(sLR)), sumMS = add32(EPSILON, wv_sum(sMS));
   sumMS = mult16_32_q15(QC16(0.707107f, 15), sumMS);
   int thetas = 13;
   if (LM <= 1) thetas -= 8;
   return mult16_32_q15((ct_eBands[13] << (LM + 1)) + thetas, sumMS) > mult16_32_q15(ct_eBands[13] << (LM + 1), sumLR);
}

/* alloc_trim_analysis (celt_encoder.c:865): band cross-correlations one lane per band, scalar tail on lane 0 */
WV_DEVN void alloc_trim_analysis_wave(WV_LDS FrameLds *L, const OaAnalysisInfo *an)
{
   WV_LDS FrameShared *sh = &L->sh;
   WV_LDS OaEncScalars *st = &L->st;
   const int end = sh->end, LM = sh->LM, C = sh->C, N0 = OA_CODED_BINS, intensity = st->intensity;
   const WV_LDS i32 *X = L->BC.xs[0];
   WV_LDS i32 *partial = L->scr;
   if (C == 2) {
      FOR_LANES(i, NBE) {
         if (i < 8 || i < intensity)
            partial[i] = inner_prod_norm_shift_l(&X[ct_eBands[i] << LM], &X[N0 + (ct_eBands[i] << LM)], (ct_eBands[i + 1] - ct_eBands[i]) << LM);
      }
   }
   wv_sync();
   LANE0 {
      const WV_LDS i32 *bandLogE = L->bandLogE;
      i32 diff = 0, equiv_rate = sh->equiv_rate;
      i16 trim = QC16(5.f, 8), logXC, logXC2;
      if (equiv_rate < 64000) trim = QC16(4.f, 8);
      else if (equiv_rate < 80000) { i32 frac = (equiv_rate - 64000) >> 10; trim = (i16)(QC16(4.f, 8) + QC16(1.f / 16.f, 8) * frac); }
      if (C == 2) {
         i16 sum = 0, minXC;
         for (int i = 0; i < 8; i++) sum = add16(sum, extract16(partial[i] >> 18));
         sum = (i16)mult16_16_q15(QC16(1.f / 8, 15), sum);
         sum = (i16)imin(QC16(1.f, 10), iabs(sum));
         minXC = sum;
         for (int i = 8; i < intensity; i++) minXC = (i16)imin(minXC, iabs(extract16(partial[i] >> 18)));
         minXC = (i16)imin(QC16(1.f, 10), iabs(minXC));
         logXC = fx_log2(QC32(1.001f, 20) - mult16_16(sum, sum));
         logXC2 = (i16)imax(logXC >> 1, fx_log2(QC32(1.001f, 20) - mult16_16(minXC, minXC)));
         logXC = (i16)pshr32(logXC - QC16(6.f, 10), 10 - 8);
         logXC2 = (i16)pshr32(logXC2 - QC16(6.f, 10), 10 - 8);
         trim = (i16)(trim + imax(-QC16(4.f, 8), mult16_16_q15(QC16(.75f, 15), logXC)));
         st->stereo_saving = (i16)imin((i16)st->stereo_saving + QC16(0.25f, 8), -(logXC2 >> 1));
      }
      for (int c = 0; c < C; c++)
         for (int i = 0; i < end - 1; i++) diff += (bandLogE[i + c * NBE] >> 5) * (i32)(2 + 2 * i - end);
      diff /= C * (end - 1);
      trim = (i16)(trim - imax(-QC16(2.f, 8), imin(QC16(2.f, 8), ((diff + QC32(1.f, DB_SHIFT - 5)) >> (DB_SHIFT - 13)) / 6)));
      trim = (i16)(trim - (sh->surround_trim >> (DB_SHIFT - 8)));
      trim = (i16)(trim - 2 * ((i16)sh->tf_estimate >> (14 - 8)));
      if (an->valid) trim = (i16)an_trim_tonality_slope(trim, an);                               /* :935-939 */
      int trim_index = pshr32(trim, 8);
      sh->alloc_trim = imax(0, imin(10, trim_index));
   }
   wv_sync();
}

/* lane 0: compute_vbr (celt_encoder.c:1605) */
WV_DEV i32 compute_vbr_l0(WV_LDS FrameLds *L, i32 base_target, const OaAnalysisInfo *an)
{
   WV_LDS FrameShared *sh = &L->sh;
   WV_LDS OaEncScalars *st = &L->st;
   const int LM = sh->LM, C = sh->C, intensity = st->intensity, constrained_vbr = sh->constrained_vbr;
   const i32 bitrate = sh->equiv_rate, maxDepth = sh->maxDepth, temporal_vbr = sh->temporal_vbr;
   const i16 tf_estimate = (i16)sh->tf_estimate;
   i16 stereo_saving = (i16)st->stereo_saving;
   i32 target;
   int coded_bands = st->lastCodedBands ? st->lastCodedBands : NBE;
   int coded_bins = ct_eBands[coded_bands] << LM;
   if (C == 2) coded_bins += ct_eBands[imin(intensity, coded_bands)] << LM;
   target = an_vbr_activity(base_target, coded_bins, an);                                            /* :1632: less for a frame the analysis calls inactive */
   if (C == 2) {
      int coded_stereo_bands = imin(intensity, coded_bands);
      int coded_stereo_dof = (ct_eBands[coded_stereo_bands] << LM) - coded_stereo_bands;
      i16 max_frac = (i16)(mult16_16(QC16(0.8f, 15), coded_stereo_dof) / (i16)coded_bins);
      stereo_saving = (i16)imin(stereo_saving, QC16(1.f, 8));
      target -= (i32)imin(mult16_32_q15(max_frac, target), mult16_16(stereo_saving - QC16(0.1f, 8), (coded_stereo_dof << BITRES)) >> 8);
   }
   target += sh->tot_boost - (19 << LM);
   i16 tf_calibration = QC16(0.044f, 14);
   target += (i32)shl32(mult16_32_q15(tf_estimate - tf_calibration, target), 1);
   const int has_surround_mask = sh->energy_mask_on, lfe = sh->lfe;
   if (an->valid && !lfe) target = an_vbr_tonality(target, coded_bins, sh->pitch_change, an);         /* :1658-1670: tonality boost */
   if (has_surround_mask && !lfe) {
      const i32 surround_target = target + (i32)(mult16_16((i16)(sh->surround_masking >> (DB_SHIFT - 10)), coded_bins << BITRES) >> 10);
      target = imax(target / 4, surround_target);
   }
   {
      int bins = ct_eBands[NBE - 2] << LM;
      i32 floor_depth = (i32)(mult16_32_q15((C * bins << BITRES), maxDepth) >> (DB_SHIFT - 15));
      floor_depth = imax(floor_depth, target >> 2);
      target = imin(target, floor_depth);
   }
   if ((!has_surround_mask || lfe) && constrained_vbr) target = base_target + (i32)mult16_32_q15(QC16(0.67f, 15), target - base_target);
   if (!has_surround_mask && tf_estimate < QC16(.2f, 14)) {
      i16 amount = (i16)mult16_16_q15(QC16(.0000031f, 30), imax(0, imin(32000, 96000 - bitrate)));
      i16 tvbr_factor = (i16)(mult16_16(temporal_vbr >> (DB_SHIFT - 10), amount) >> 10);
      target += (i32)mult16_32_q15(tvbr_factor, target);
   }
   return imin(2 * base_target, target);
}
WV_DEV int hysteresis_decision(i16 val, const i16 *thresholds, const i16 *hysteresis, int N, int prev)
{
   int i;
   for (i = 0; i < N; i++) if (val < thresholds[i]) break;
   if (i > prev && val < thresholds[prev] + hysteresis[prev]) i = prev;
   if (i < prev && val > thresholds[prev - 1] - hysteresis[prev - 1]) i = prev;
   return i;
}
#endif

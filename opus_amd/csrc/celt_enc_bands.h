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

/* compute_band_energies + amp2Log2 of the channel resident in W: one lane per band */
WV_DEV void band_energies_channel(WV_LDS FrameLds *L, const WV_LDS i32 *W, int c, WV_LDS i32 *bandLogE_out)
{
   const int LM = L->sh.LM, end = L->sh.end, effEnd = L->sh.effEnd;
   FOR_LANES(i, NBE) {
      if (i < effEnd) {
         const WV_LDS i32 *x = &W[ct_eBands[i] << LM];
         int len = (ct_eBands[i + 1] - ct_eBands[i]) << LM;
         i32 mx = 0, mn = 0, sum = 0, E;
         for (int j = 0; j < len; j++) { mx = imax(mx, x[j]); mn = imin(mn, x[j]); }
         i32 maxval = imax(mx, neg32(mn));
         if (maxval > 0) {
            int shift = imax(0, 30 - celt_ilog2(maxval + (maxval >> 14) + 1) - ((((ct_logN[i] + 7) >> BITRES) + LM + 1) >> 1));
            for (int j = 0; j < len; j++) { i32 v = shl32(x[j], shift); sum = add32(sum, mult32_32_q31(v, v)); }
            E = imax(maxval, pshr32(fx_sqrt32(sum >> 1), shift));
         } else E = EPSILON;
         L->bandE[i + c * NBE] = E;
         bandLogE_out[i + c * NBE] = fx_log2_db(E) - shl32((i32)ct_eMeans[i], DB_SHIFT - 4) + GC(2.f);
      } else if (i < end) bandLogE_out[i + c * NBE] = -GC(14.f);
   }
}

/* compute_mdcts (celt_encoder.c:511) + compute_band_energies (bands.c:95) + amp2Log2: one channel at a time through the LDS work buffer W -- B interleaved
 * transforms in place, the band energies while the channel is resident, then the channel goes out to the HBM spectrum g->X */
WV_DEVN void compute_mdcts_wave(WV_LDS FrameLds *L, const OaEncState *gst, int shortBlocks, WV_LDS i32 *bandLogE_out)
{
   const int C = L->sh.C, CC = L->sh.CC, LM = L->sh.LM;
   CeltScratch *G = L->g;
   WV_LDS i32 *W = L->BC.W;
   int B, N, shift;
   if (shortBlocks) { B = shortBlocks; N = 120; shift = 3; }
   else { B = 1; N = 120 << LM; shift = 3 - LM; }
   const int up = L->sh.upsample > 1 ? L->sh.upsample : 1, bound = B * N / up;
   for (int c = 0; c < CC; c++) {
      mdct_forward_blocks(gst->in_mem + c * OA_OVERLAP, G->in[c], W, shift, B, L->aux);
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
      FOR_LANES(i, B * N) G->X[cc * B * N + i] = W[i];
      wv_sync();
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
   for (int c = 0; c < C; c++) FOR_LANES(j, n) L->BC.xs[c][j] = X[c * N + j];
   wv_sync();
}

/* lane 0: temporal VBR follower (celt_encoder.c:2196-2213) */
WV_DEV void temporal_vbr_l0(WV_LDS FrameLds *L)
{
   const int C = L->sh.C, start = L->sh.start, end = L->sh.end, LM = L->sh.LM;
   i32 follow = -QC32(10.0f, DB_SHIFT - 5), frame_avg = 0, offset = L->sh.shortBlocks ? half32(shl32(LM, DB_SHIFT - 5)) : 0;
   for (int i = start; i < end; i++) {
      follow = imax(follow - QC32(1.0f, DB_SHIFT - 5), (L->bandLogE[i] >> 5) - offset);
      if (C == 2) follow = imax(follow, (L->bandLogE[i + NBE] >> 5) - offset);
      frame_avg += follow;
   }
   frame_avg /= (end - start);
   i32 tv = sub32(shl32(frame_avg, 5), L->st.spec_avg);
   tv = imin(GC(3.f), imax(-GC(1.5f), tv));
   L->st.spec_avg += mult16_32_q15(QC16(.02f, 15), tv);
   L->sh.temporal_vbr = tv;
}

/* lane 0: patch_transient_decision (celt_encoder.c:473) */
WV_DEV int patch_transient_decision_l0(WV_LDS FrameLds *L)
{
   const int C = L->sh.C, start = L->sh.start, end = L->sh.end;
   const WV_LDS i32 *newE = L->bandLogE, *oldE = L->oldBandE;
   WV_LDS i32 *spread_old = L->scr;
   i32 mean_diff = 0;
   if (C == 1) {
      spread_old[start] = oldE[start];
      for (int i = start + 1; i < end; i++) spread_old[i] = imax(spread_old[i - 1] - GC(1.0f), oldE[i]);
   } else {
      spread_old[start] = imax(oldE[start], oldE[start + NBE]);
      for (int i = start + 1; i < end; i++) spread_old[i] = imax(spread_old[i - 1] - GC(1.0f), imax(oldE[i], oldE[i + NBE]));
   }
   for (int i = end - 2; i >= start; i--) spread_old[i] = imax(spread_old[i], spread_old[i + 1] - GC(1.0f));
   for (int c = 0; c < C; c++)
      for (int i = imax(2, start); i < end - 1; i++) {
         i16 x1 = (i16)imax(0, newE[i + c * NBE]);
         i16 x2 = (i16)imax(0, spread_old[i]);
         mean_diff = add32(mean_diff, imax(0, sub32(x1, x2)));
      }
   mean_diff = mean_diff / (C * (end - 1 - imax(2, start)));
   return mean_diff > GC(1.f);
}

WV_DEV i32 median_of_5(const WV_LDS i32 *x)
{
   i32 t0, t1, t2 = x[2], t3, t4, t;
   if (x[0] > x[1]) { t0 = x[1]; t1 = x[0]; } else { t0 = x[0]; t1 = x[1]; }
   if (x[3] > x[4]) { t3 = x[4]; t4 = x[3]; } else { t3 = x[3]; t4 = x[4]; }
   if (t0 > t3) { t = t0; t0 = t3; t3 = t; t = t1; t1 = t4; t4 = t; }
   if (t2 > t1) return t1 < t3 ? imin(t2, t3) : imin(t4, t1);
   return t2 < t3 ? imin(t1, t3) : imin(t2, t4);
}
WV_DEV i32 median_of_3(const WV_LDS i32 *x)
{
   i32 t0, t1, t2 = x[2];
   if (x[0] > x[1]) { t0 = x[1]; t1 = x[0]; } else { t0 = x[0]; t1 = x[1]; }
   if (t1 < t2) return t1;
   if (t0 < t2) return t2;
   return t0;
}

/* lane 0: dynalloc_analysis (celt_encoder.c:1049) */
WV_DEVN void dynalloc_analysis_l0(WV_LDS FrameLds *L)
{
   WV_LDS FrameShared *sh = &L->sh;
   const int start = sh->start, end = sh->end, C = sh->C, LM = sh->LM, lsb_depth = sh->lsb_depth, isTransient = sh->isTransient;
   const int vbr = sh->vbr, constrained_vbr = sh->constrained_vbr, effectiveBytes = sh->effectiveBytes;
   const WV_LDS i32 *bandLogE = L->bandLogE, *bandLogE2 = L->bandLogE2, *oldBandE = L->oldBandE;
   WV_LDS i32 *offsets = L->offsets, *importance = L->importance, *spread_weight = L->spread_weight;
   WV_LDS i32 *const big = L->BC.tf;      /* 126 words: more than scr holds; BC is idle between the last MDCT and tf_analysis */
   WV_LDS i32 *follower = big, *noise_floor = big + 42, *bandLogE3 = big + 63, *mask = big + 84, *sig = big + 105;
   i32 tot_boost = 0, maxDepth = -GC(31.9f);
   for (int i = 0; i < NBE; i++) offsets[i] = 0;
   for (int i = 0; i < end; i++)
      noise_floor[i] = GC(0.0625f) * ct_logN[i] + GC(.5f) + shl32(9 - lsb_depth, DB_SHIFT) - shl32(ct_eMeans[i], DB_SHIFT - 4)
            + GC(.0062f) * (i + 5) * (i + 5);
   for (int c = 0; c < C; c++) for (int i = 0; i < end; i++) maxDepth = imax(maxDepth, bandLogE[c * NBE + i] - noise_floor[i]);
   {
      for (int i = 0; i < end; i++) mask[i] = bandLogE[i] - noise_floor[i];
      if (C == 2) for (int i = 0; i < end; i++) mask[i] = imax(mask[i], bandLogE[NBE + i] - noise_floor[i]);
      for (int i = 0; i < end; i++) sig[i] = mask[i];
      for (int i = 1; i < end; i++) mask[i] = imax(mask[i], mask[i - 1] - GC(2.f));
      for (int i = end - 2; i >= 0; i--) mask[i] = imax(mask[i], mask[i + 1] - GC(3.f));
      for (int i = 0; i < end; i++) {
         i32 smr = sig[i] - imax(imax(0, maxDepth - GC(12.f)), mask[i]);
         int shift = -pshr32(imax(-GC(5.f), imin(0, smr)), DB_SHIFT);
         spread_weight[i] = 32 >> shift;
      }
   }
   if (effectiveBytes >= (30 + 5 * LM) && !sh->lfe) {
      int last = 0;
      for (int c = 0; c < C; c++) {
         i32 offset, tmp;
         WV_LDS i32 *f;
         for (int i = 0; i < end; i++) bandLogE3[i] = bandLogE2[c * NBE + i];
         if (LM == 0) for (int i = 0; i < imin(8, end); i++) bandLogE3[i] = imax(bandLogE2[c * NBE + i], oldBandE[c * NBE + i]);
         f = &follower[c * NBE];
         f[0] = bandLogE3[0];
         for (int i = 1; i < end; i++) {
            if (bandLogE3[i] > bandLogE3[i - 1] + GC(.5f)) last = i;
            f[i] = imin(f[i - 1] + GC(1.5f), bandLogE3[i]);
         }
         for (int i = last - 1; i >= 0; i--) f[i] = imin(f[i], imin(f[i + 1] + GC(2.f), bandLogE3[i]));
         offset = GC(1.f);
         for (int i = 2; i < end - 2; i++) f[i] = imax(f[i], median_of_5(&bandLogE3[i - 2]) - offset);
         tmp = median_of_3(&bandLogE3[0]) - offset;
         f[0] = imax(f[0], tmp); f[1] = imax(f[1], tmp);
         tmp = median_of_3(&bandLogE3[end - 3]) - offset;
         f[end - 2] = imax(f[end - 2], tmp); f[end - 1] = imax(f[end - 1], tmp);
         for (int i = 0; i < end; i++) f[i] = imax(f[i], noise_floor[i]);
      }
      if (C == 2) {
         for (int i = start; i < end; i++) {
            follower[NBE + i] = imax(follower[NBE + i], follower[i] - GC(4.f));
            follower[i] = imax(follower[i], follower[NBE + i] - GC(4.f));
            follower[i] = half32(imax(0, bandLogE[i] - follower[i]) + imax(0, bandLogE[NBE + i] - follower[NBE + i]));
         }
      } else for (int i = start; i < end; i++) follower[i] = imax(0, bandLogE[i] - follower[i]);
      for (int i = start; i < end; i++) follower[i] = imax(follower[i], L->surround_dynalloc[i]);
      for (int i = start; i < end; i++) importance[i] = pshr32(13 * fx_exp2_db(imin(follower[i], GC(4.f))), 16);
      if ((!vbr || constrained_vbr) && !isTransient) for (int i = start; i < end; i++) follower[i] = half32(follower[i]);
      for (int i = start; i < end; i++) {
         if (i < 8) follower[i] *= 2;
         if (i >= 12) follower[i] = half32(follower[i]);
      }
      if (sh->toneishness > QC32(.98f, 29)) {
         int freq_bin = pshr32((i32)(i16)sh->tone_freq * QC16(120 / 3.14159265358979323846, 9), 13 + 9);
         for (int i = start; i < end; i++) {
            if (freq_bin >= ct_eBands[i] && freq_bin <= ct_eBands[i + 1]) follower[i] += GC(2.f);
            if (freq_bin >= ct_eBands[i] - 1 && freq_bin <= ct_eBands[i + 1] + 1) follower[i] += GC(1.f);
            if (freq_bin >= ct_eBands[i] - 2 && freq_bin <= ct_eBands[i + 1] + 2) follower[i] += GC(1.f);
            if (freq_bin >= ct_eBands[i] - 3 && freq_bin <= ct_eBands[i + 1] + 3) follower[i] += GC(.5f);
         }
         if (freq_bin >= ct_eBands[end]) { follower[end - 1] += GC(2.f); follower[end - 2] += GC(1.f); }
      }
      if (effectiveBytes > 320) follower[0] += imin(GC(1.5f), GC(1e-3f) * (effectiveBytes - 320));
      for (int i = start; i < end; i++) {
         int width, boost, boost_bits;
         follower[i] = imin(follower[i], GC(4));
         follower[i] = follower[i] >> 8;
         width = C * (ct_eBands[i + 1] - ct_eBands[i]) << LM;
         if (width < 6) { boost = (int)(follower[i] >> (DB_SHIFT - 8)); boost_bits = boost * width << BITRES; }
         else if (width > 48) { boost = (int)((follower[i] * 8) >> (DB_SHIFT - 8)); boost_bits = (boost * width << BITRES) / 8; }
         else { boost = (int)((follower[i] * width / 6) >> (DB_SHIFT - 8)); boost_bits = boost * 6 << BITRES; }
         if ((!vbr || (constrained_vbr && !isTransient)) && (tot_boost + boost_bits) >> BITRES >> 3 > 2 * effectiveBytes / 3) {
            i32 cap = ((2 * effectiveBytes / 3) << BITRES << 3);
            offsets[i] = cap - tot_boost;
            tot_boost = cap;
            break;
         } else { offsets[i] = boost; tot_boost += boost_bits; }
      }
   } else for (int i = start; i < end; i++) importance[i] = 13;
   sh->tot_boost = tot_boost;
   sh->maxDepth = maxDepth;
}

/* tf_analysis (celt_encoder.c:663): L1 metrics = one lane per band on a private scratch copy (Haar transforms
 * are done in place, tf_tmp holds the 800-bin copy + 560 for the "-1" trial); Viterbi on lane 0. */
WV_DEV i32 l1_metric_l(const WV_LDS i32 *tmp, int N, int LM, i16 bias)
{
   i32 L1 = 0;
   for (int i = 0; i < N; i++) L1 += iabs(tmp[i] >> (NORM_SHIFT - 14));
   return mac16_32_q15(L1, LM * bias, L1);
}
WV_DEV void haar1_l(WV_LDS i32 *X, int N0, int stride)        /* one lane, serial (bands.c:623) */
{
   N0 >>= 1;
   for (int i = 0; i < stride; i++)
      for (int j = 0; j < N0; j++) {
         i32 t1 = mult32_32_q31(QC32(.70710678f, 31), X[stride * 2 * j + i]);
         i32 t2 = mult32_32_q31(QC32(.70710678f, 31), X[stride * (2 * j + 1) + i]);
         X[stride * 2 * j + i] = add32(t1, t2);
         X[stride * (2 * j + 1) + i] = sub32(t1, t2);
      }
}
WV_DEVN void tf_analysis_wave(WV_LDS FrameLds *L, int lambda)
{
   WV_LDS FrameShared *sh = &L->sh;
   const int len = sh->effEnd, isTransient = sh->isTransient, LM = sh->LM, N0 = sh->N, tf_chan = sh->tf_chan;
   const i16 tf_estimate = (i16)sh->tf_estimate;
   WV_LDS i32 *metric = L->scr, *path0 = L->scr + 21, *path1 = L->scr + 42;
   WV_LDS i32 *tmpA = L->BC.tf;             /* 800 words: private per-band segments (folding memory not live yet) */
   WV_LDS i32 *tmpB = L->BC.tf + OA_CODED_BINS;    /* second copy for the "-1" trial */
   const i32 *X = L->g->X + tf_chan * N0;
   i16 bias = (i16)mult16_16_q14(QC16(.04f, 15), imax(-QC16(.25f, 14), QC16(.5f, 14) - tf_estimate));
   wv_sync();
   FOR_LANES(j, ct_eBands[len] << LM) tmpA[j] = X[j];            /* the trial buffer is laid out like the spectrum: one coalesced copy from HBM */
   wv_sync();
   FOR_LANES(i, len) {
      int off = ct_eBands[i] << LM, N = (ct_eBands[i + 1] - ct_eBands[i]) << LM, narrow = (ct_eBands[i + 1] - ct_eBands[i]) == 1, best_level = 0;
      WV_LDS i32 *tmp = tmpA + off, *tmp_1 = tmpB + off;
      i32 L1 = l1_metric_l(tmp, N, isTransient ? LM : 0, bias), best_L1 = L1;
      if (isTransient && !narrow) {
         for (int j = 0; j < N; j++) tmp_1[j] = tmp[j];
         haar1_l(tmp_1, N >> LM, 1 << LM);
         L1 = l1_metric_l(tmp_1, N, LM + 1, bias);
         if (L1 < best_L1) { best_L1 = L1; best_level = -1; }
      }
      for (int k = 0; k < LM + !(isTransient || narrow); k++) {
         int B = isTransient ? (LM - k - 1) : k + 1;
         haar1_l(tmp, N >> k, 1 << k);
         L1 = l1_metric_l(tmp, N, B, bias);
         if (L1 < best_L1) { best_L1 = L1; best_level = k + 1; }
      }
      int m = isTransient ? 2 * best_level : -2 * best_level;
      if (narrow && (m == 0 || m == -2 * LM)) m -= 1;
      metric[i] = m;
   }
   wv_sync();
   LANE0 {
      const WV_LDS i32 *importance = L->importance;
      WV_LDS i32 *tf_res = L->tf_res;
      int cost0, cost1, selcost[2], tf_select = 0;
      for (int sel = 0; sel < 2; sel++) {
         cost0 = importance[0] * iabs(metric[0] - 2 * k_tf_select_table[LM][4 * isTransient + 2 * sel + 0]);
         cost1 = importance[0] * iabs(metric[0] - 2 * k_tf_select_table[LM][4 * isTransient + 2 * sel + 1]) + (isTransient ? 0 : lambda);
         for (int i = 1; i < len; i++) {
            int curr0 = imin(cost0, cost1 + lambda), curr1 = imin(cost0 + lambda, cost1);
            cost0 = curr0 + importance[i] * iabs(metric[i] - 2 * k_tf_select_table[LM][4 * isTransient + 2 * sel + 0]);
            cost1 = curr1 + importance[i] * iabs(metric[i] - 2 * k_tf_select_table[LM][4 * isTransient + 2 * sel + 1]);
         }
         selcost[sel] = imin(cost0, cost1);
      }
      if (selcost[1] < selcost[0] && isTransient) tf_select = 1;
      cost0 = importance[0] * iabs(metric[0] - 2 * k_tf_select_table[LM][4 * isTransient + 2 * tf_select + 0]);
      cost1 = importance[0] * iabs(metric[0] - 2 * k_tf_select_table[LM][4 * isTransient + 2 * tf_select + 1]) + (isTransient ? 0 : lambda);
      for (int i = 1; i < len; i++) {
         int curr0, curr1, from0 = cost0, from1 = cost1 + lambda;
         if (from0 < from1) { curr0 = from0; path0[i] = 0; } else { curr0 = from1; path0[i] = 1; }
         from0 = cost0 + lambda; from1 = cost1;
         if (from0 < from1) { curr1 = from0; path1[i] = 0; } else { curr1 = from1; path1[i] = 1; }
         cost0 = curr0 + importance[i] * iabs(metric[i] - 2 * k_tf_select_table[LM][4 * isTransient + 2 * tf_select + 0]);
         cost1 = curr1 + importance[i] * iabs(metric[i] - 2 * k_tf_select_table[LM][4 * isTransient + 2 * tf_select + 1]);
      }
      tf_res[len - 1] = cost0 < cost1 ? 0 : 1;
      for (int i = len - 2; i >= 0; i--) tf_res[i] = tf_res[i + 1] == 1 ? path1[i + 1] : path0[i + 1];
      for (int i = len; i < sh->end; i++) tf_res[i] = tf_res[len - 1];
      sh->tf_select = tf_select;
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
WV_DEVN void alloc_trim_analysis_wave(WV_LDS FrameLds *L)
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
      int trim_index = pshr32(trim, 8);
      sh->alloc_trim = imax(0, imin(10, trim_index));
   }
   wv_sync();
}

/* lane 0: compute_vbr (celt_encoder.c:1605) */
WV_DEV i32 compute_vbr_l0(WV_LDS FrameLds *L, i32 base_target)
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
   target = base_target;
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

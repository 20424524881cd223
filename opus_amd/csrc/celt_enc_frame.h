/* celt_enc_frame.h — one (stream, frame) on one wavefront: HBM state -> LDS -> packet bytes -> HBM.
 * Follows src/opus_encoder.c:1182 opus_encode_native / :1855 opus_encode_frame_native (CELT-only branch) and
 * celt/celt_encoder.c:1726 celt_encode_with_ec step for step; see the per-phase headers for the parallel mapping.
 * HBM traffic per frame: PCM in (frame*channels*2 B) + state in/out (2 x ~10 KB, coalesced dwords) + packet out. */
#ifndef OPUS_AMD_CELT_ENC_FRAME_H
#define OPUS_AMD_CELT_ENC_FRAME_H

#ifndef K_DUMP
#define K_DUMP(tag, ptr, nbytes)
#define K_DUMPI(tag, v)
#endif
/* profiling build only (-DOA_PHASE_TIMERS): per-phase shader-clock accounting, see opus_amd.hip */
#ifndef K_PHASE
#define K_PHASE(id)
#define K_PHASE_BEGIN()
#endif

/* Final coalesced packet store.  nbytes at pk (pk[0] = TOC), or -- hard CBR (pad_to != 0) -- the same packet re-framed as a code-3 packet padded with zeros to exactly
 * pad_to bytes: opus_packet_pad (src/repacketizer.c:346, out_range_impl :112 with pad = 1).  The input is either a single-frame packet (code 0) or one of the
 * payload-less 'PLC' packets of the Opus layer (code 1: two empty frames, code 3: M empty frames): [toc|3][M|0x40][255 x n][rest][frame][zeros]. */
WV_DEV int oa_emit_packet_wave(const WV_LDS u8 *pk, u8 *out, int nbytes, int pad_to, int out_cap)
{
   if (nbytes <= 0) return nbytes;
   if (pad_to == 0 || nbytes == pad_to) { if (nbytes > out_cap) return -2; FOR_LANES(i, nbytes) out[i] = pk[i]; return nbytes; }
   if (nbytes > pad_to) return -3;
   if (pad_to > out_cap) return -2;
   const int code = pk[0] & 3;
   const int count = code == 0 ? 1 : code == 3 ? (pk[1] & 0x3F) : 2, L0 = code == 0 ? nbytes - 1 : 0, src0 = code == 0 ? 1 : nbytes;
   const int pad_amount = pad_to - (2 + count * L0);
   const int nb_255s = pad_amount > 0 ? (pad_amount - 1) / 255 : 0, hdr = 2 + (pad_amount > 0 ? nb_255s + 1 : 0);
   FOR_LANES(i, pad_to) {
      u8 v = 0;
      if (i == 0) v = (u8)((pk[0] & 0xFC) | 0x3);
      else if (i == 1) v = (u8)(count | (pad_amount != 0 ? 0x40 : 0));
      else if (i < hdr) v = i < hdr - 1 ? 255 : (u8)(pad_amount - 255 * nb_255s - 1);
      else if (i < hdr + L0) v = pk[src0 + i - hdr];
      out[i] = v;
   }
   return pad_to;
}
WV_DEV int emit_packet_wave(WV_LDS FrameLds *L, u8 *out, int nbytes, int pad_to, int out_cap) { return oa_emit_packet_wave(L->packet, out, nbytes, pad_to, out_cap); }

/* celt_encode_with_ec from the pre-emphasis on (celt/celt_encoder.c:1990-2830).  Expects the CELT scalars in L->st, oldBandE / energyError in LDS, the frame
 * constants of the prologue in L->sh, the int16 input staged in L->g->pcm16 (HBM) and the range coder in L->ec (fresh, or -- HYB -- continuing after the SILK layer).
 * HYB = the hybrid branches of the reference (start band 17: no pitch pre-filter, no tf_analysis, weak transients, its own VBR target, :2030-2470). */
/* The frame can be CUT between the fine energy and the PVQ (the kernel pipeline of the CELT-only applications, opus_amd.hip): celt_encode_core<..>(.., cut) parks the wave's
 * LDS image and the normalised spectrum in the stream's continuation record and returns OA_CUT; oa_celt_pvq_kernel codes the bands of four such streams per wave
 * (celt_enc_pvq4.h); the back kernel reloads the image and runs the *_tail functions below.  cut == NULL: the whole frame here, as ever. */
#define OA_CUT (-1000)
template <bool HYB> WV_DEV void celt_encode_core_tail(WV_LDS FrameLds *L, OaEncState *gst);
WV_DEV void celt_cut_dump(WV_LDS FrameLds *L, CeltCont *cut, int x_stored /* the spectrum is in the record already (compute_mdcts_wave's xcut) */)
{
   const int C = L->sh.C, N = L->sh.N, nb = L->sh.M * ct_eBands[L->sh.end];
   wv_sync();
   FOR_LANES(i, (int)(offsetof(FrameLds, BC) / 4)) cut->image[i] = ((const WV_LDS i32 *)L)[i];
   if (!x_stored) for (int c = 0; c < C; c++) { const i32 *X = L->g->X + c * N; FOR_LANES(j, nb) cut->X[c][j] = X[j]; }
   LANE0 cut->state = 1;
}
template <bool HYB, bool NOPVQ = false> WV_DEV int celt_encode_core(WV_LDS FrameLds *L, OaEncState *gst, u8 *journal, const i32 *energy_mask = nullptr, const i32 *tr_pre = nullptr /* ct_transient_tile's record of the stream, or NULL */,
      CeltCont *cut = nullptr)
{
   WV_LDS FrameShared *sh = &L->sh;
   WV_LDS OaEncScalars *st = &L->st;
   const int CC = sh->CC, C = sh->C;
   const int N = sh->N, LM = sh->LM, M = sh->M, start = sh->start, end = sh->end;

   K_PHASE(1);
   /* ---- pre-emphasis (celt_encoder.c:557): the new samples once, into the spectrum scratch (free until the first MDCT); pre_at() reads history and new samples alike ---- */
   PreSrc ps0, ps1;
   const int up = sh->upsample > 1 ? wv_uni(sh->upsample) : 1;
   pre_stage_wave(L->g->X, L->g->pcm16, CC, N, st->preemph_memE[0], st->preemph_memE[1], up);
   ps0.hist = gst->prefilter_mem; ps0.xnew = L->g->X;
   ps1.hist = gst->prefilter_mem + OA_MAX_PERIOD; ps1.xnew = L->g->X + N;
   wv_sync();
   LANE0 { for (int c = 0; c < CC; c++) st->preemph_memE[c] = up > 1 ? 0 : mult16_32_q15(27853, shl32((i32)L->g->pcm16[CC * (N - 1) + c], SIG_SHIFT)); }   /* the last zero-stuffed sample is 0 */
   wv_sync();

   K_PHASE(2);
   /* ---- tone / transient analysis ---- */
   tone_detect_wave(L, ps0, ps1);
   wv_sync();
   K_DUMPI("tone_freq", (i16)sh->tone_freq); K_DUMPI("toneishness", sh->toneishness);
   LANE0 { sh->isTransient = 0; sh->shortBlocks = 0; sh->tf_estimate = 0; sh->tf_chan = 0; sh->weak_transient = 0; sh->transient_got_disabled = 0; }
   wv_sync();
   K_PHASE(3);
   if (sh->complexity >= 1 && !sh->lfe) transient_analysis_wave(L, ps0, ps1, HYB && sh->effectiveBytes < 15 && sh->silk_signalType != 2, tr_pre);
   wv_sync();
   K_DUMPI("isTransient", sh->isTransient); K_DUMPI("tf_estimate", (i16)sh->tf_estimate); K_DUMPI("tf_chan", sh->tf_chan);
   LANE0 sh->toneishness = imin(sh->toneishness, QC32(1.f, 29) - shl32((i16)sh->tf_estimate, 15));
   wv_sync();

   K_PHASE(4);
   /* ---- pitch pre-filter ---- */
   {
      int enabled = ((sh->lfe && sh->nbAvailableBytes > 3) || sh->nbAvailableBytes > 12 * C) && !HYB && !sh->silence && sh->tell + 16 <= sh->total_bits && !sh->disable_pf;
      run_prefilter_wave(L, gst, ps0, ps1, enabled);
      LANE0 {
         EC_BEGIN;
         int pitch_index = sh->pitch_index; i16 gain1 = (i16)sh->gain1;
         sh->pitch_change = 0;
         if ((gain1 > QC16(.4f, 15) || (i16)st->prefilter_gain > QC16(.4f, 15)) && an_tonal_enough_for_prefilter(&gst->analysis) && (pitch_index > 1.26 * st->prefilter_period || pitch_index < .79 * st->prefilter_period)) sh->pitch_change = 1;
         if (sh->pf_on == 0) {
            if (!HYB && sh->tell + 16 <= sh->total_bits) k_ec_enc_bit_logp(EC_PASS, 0, 1);
         } else {
            int octave;
            k_ec_enc_bit_logp(EC_PASS, 1, 1);
            pitch_index += 1;
            octave = ec_ilog(pitch_index) - 5;
            k_ec_enc_uint(EC_PASS, octave, 6);
            k_ec_enc_bits(EC_PASS, pitch_index - (16 << octave), 4 + octave);
            pitch_index -= 1;
            k_ec_enc_bits(EC_PASS, sh->qg, 3);
            k_ec_enc_icdf(EC_PASS, sh->prefilter_tapset, k_tapset_icdf, 2);
         }
         if (LM > 0 && k_ec_tell(EC_PASS) + 3 <= sh->total_bits) { if (sh->isTransient) sh->shortBlocks = M; }
         else { sh->isTransient = 0; sh->transient_got_disabled = 1; }
         sh->secondMdct = sh->shortBlocks && sh->complexity >= 8;
         EC_END;
      }
      wv_sync();
      K_DUMPI("pf_on", sh->pf_on); K_DUMPI("pitch_index", sh->pitch_index); K_DUMPI("gain1", (i16)sh->gain1); K_DUMPI("qg", sh->qg);
   }

   K_PHASE(5);
   /* ---- MDCT + band energies ---- */
   if (sh->secondMdct) {
      compute_mdcts_wave(L, gst, 0, L->bandLogE2);
      FOR_LANES(w, C * NBE) { int c = w / NBE, i = w - c * NBE; if (i < end) L->bandLogE2[c * NBE + i] += half32(shl32(LM, DB_SHIFT)); }
      wv_sync();
   }
#ifdef K_DUMP_ENABLED
   const int fuse_norm = 0;                                           /* (the stage dumps of the test build show the spectrum before and after normalise_bands) */
#else
   const int fuse_norm = !sh->lfe;                                    /* (LFE changes the energies first: normalise_bands_wave below) */
#endif
   i32 *const xcut = fuse_norm && cut && LM >= 2 && sh->effEnd == end ? &cut->X[0][0] : nullptr;      /* (the record's rows hold the bins below eBands[end]; the normalisation covers those below eBands[effEnd]) */
   compute_mdcts_wave(L, gst, sh->shortBlocks, L->bandLogE, fuse_norm, xcut);
   if (CC == 2 && C == 1) { LANE0 sh->tf_chan = 0; }
   K_DUMPI("shortBlocks", sh->shortBlocks); K_DUMP("freq", L->g->X, C * N * 4); K_DUMP("bandE", L->bandE, 42 * 4); K_DUMP("bandLogE", L->bandLogE, 42 * 4);
   if (sh->lfe) {                    /* LFE: nothing but the first two bands carries energy (celt_encoder.c:2099-2107) */
      LANE0 {
         for (int c = 0; c < C; c++) for (int i = 2; i < end; i++) {
            i32 E = imin(L->bandE[i + c * NBE], mult16_32_q15(QC16(1e-4f, 15), L->bandE[c * NBE])); E = imax(E, EPSILON);
            L->bandE[i + c * NBE] = E;
            if (i < sh->effEnd) L->bandLogE[i + c * NBE] = fx_log2_db(E) - shl32((i32)ct_eMeans[i], DB_SHIFT - 4) + GC(2.f);
         }
      }
   }
   LANE0 {                             /* how much masking takes place between surround channels (celt_encoder.c:2109-2186) */
      for (int i = 0; i < NBE; i++) L->surround_dynalloc[i] = 0;
      sh->surround_masking = 0; sh->surround_trim = 0;
      if (!HYB && sh->energy_mask_on && energy_mask && !sh->lfe) {
         i32 mask_avg = 0, diff = 0; int count = 0, midband, count_dynalloc = 0;
         const int mask_end = imax(2, st->lastCodedBands);
         for (int c = 0; c < C; c++) for (int i = 0; i < mask_end; i++) {
            i32 mask = imax(imin(energy_mask[NBE * c + i], GC(.25f)), -GC(2.0f));
            if (mask > 0) mask = half32(mask);
            const i16 mask16 = (i16)(mask >> (DB_SHIFT - 10));
            mask_avg += mult16_16(mask16, ct_eBands[i + 1] - ct_eBands[i]);
            count += ct_eBands[i + 1] - ct_eBands[i];
            diff += mult16_16(mask16, 1 + 2 * i - mask_end);
         }
         mask_avg = shl32(mask_avg / (i16)count, DB_SHIFT - 10);
         mask_avg += GC(.2f);
         diff = shl32(diff * 6 / (C * (mask_end - 1) * (mask_end + 1) * mask_end), DB_SHIFT - 10);
         diff = half32(diff);
         diff = imax(imin(diff, GC(.031f)), -GC(.031f));
         for (midband = 0; ct_eBands[midband + 1] < ct_eBands[mask_end] / 2; midband++);
         for (int i = 0; i < mask_end; i++) {
            const i32 lin = mask_avg + diff * (i - midband);
            i32 unmask = C == 2 ? imax(energy_mask[i], energy_mask[NBE + i]) : energy_mask[i];
            unmask = imin(unmask, GC(.0f));
            unmask -= lin;
            if (unmask > GC(.25f)) { L->surround_dynalloc[i] = unmask - GC(.25f); count_dynalloc++; }
         }
         if (count_dynalloc >= 3) {
            mask_avg += GC(.25f);
            if (mask_avg > 0) { mask_avg = 0; diff = 0; for (int i = 0; i < mask_end; i++) L->surround_dynalloc[i] = 0; }
            else for (int i = 0; i < mask_end; i++) L->surround_dynalloc[i] = imax(0, L->surround_dynalloc[i] - GC(.25f));
         }
         mask_avg += GC(.2f);
         sh->surround_trim = 64 * diff;
         sh->surround_masking = mask_avg;
      }
   }
   K_PHASE(6);
   if (!sh->lfe) temporal_vbr_wave(L); else { LANE0 sh->temporal_vbr = 0; }
   if (!sh->secondMdct) { FOR_LANES(i, C * NBE) L->bandLogE2[i] = L->bandLogE[i]; }
   {
      EC_BEGIN;                                                    /* every lane reads the coder's position; nothing is coded here */
      const int may_patch = LM > 0 && k_ec_tell(EC_PASS) + 3 <= sh->total_bits && !sh->isTransient && sh->complexity >= 5 && !sh->lfe && !HYB;
      const int do_patch = may_patch ? patch_transient_decision_wave(L) : 0;
      LANE0 sh->do_patch = do_patch;
   }
   wv_sync();
   if (sh->do_patch) {
      LANE0 { sh->isTransient = 1; sh->shortBlocks = M; }
      wv_sync();
      compute_mdcts_wave(L, gst, sh->shortBlocks, L->bandLogE, fuse_norm, xcut);
      FOR_LANES(w, C * NBE) { int c = w / NBE, i = w - c * NBE; if (i < end) L->bandLogE2[c * NBE + i] += half32(shl32(LM, DB_SHIFT)); }
      LANE0 sh->tf_estimate = QC16(.2f, 14);
      wv_sync();
   }
   store_in_mem_wave(L, gst);          /* last MDCT done: the overlap memory may now be replaced; BC is free from here */
   LANE0 { EC_BEGIN; if (LM > 0 && k_ec_tell(EC_PASS) + 3 <= sh->total_bits) k_ec_enc_bit_logp(EC_PASS, sh->isTransient, 3); EC_END; }
   if (!fuse_norm) normalise_bands_wave(L);
   K_DUMPI("isTransient2", sh->isTransient); K_DUMP("bandLogE2", L->bandLogE2, 42 * 4); for (int c = 0; c < C; c++) K_DUMP("X", L->g->X + c * N, M * ct_eBands[sh->effEnd] * 4); K_DUMPI("temporal_vbr", sh->temporal_vbr);

   K_PHASE(7);
   /* ---- allocation analyses ---- */
   LANE0 {
      sh->enable_tf_analysis = sh->effectiveBytes >= 15 * C && !HYB && sh->complexity >= 2 && !sh->lfe && sh->toneishness < QC32(.98f, 29);
   }
   wv_sync();
   dynalloc_analysis_wave(L, &gst->analysis);
   K_DUMPI("maxDepth", sh->maxDepth); K_DUMPI("tot_boost", sh->tot_boost); K_DUMP("offsets", L->offsets, 84); K_DUMP("importance", L->importance, 84); K_DUMP("spread_weight", L->spread_weight, 84);
   K_PHASE(8);
   if (sh->enable_tf_analysis) tf_analysis_wave(L, imax(80, 20480 / sh->effectiveBytes + 2));
   else {
      LANE0 {
         if (HYB && sh->weak_transient) { for (int i = 0; i < end; i++) L->tf_res[i] = 1; sh->tf_select = 0; }                                   /* celt_encoder.c:2262 */
         else if (HYB && sh->effectiveBytes < 15 && sh->silk_signalType != 2) { for (int i = 0; i < end; i++) L->tf_res[i] = 0; sh->tf_select = sh->isTransient; }
         else { for (int i = 0; i < end; i++) L->tf_res[i] = sh->isTransient; sh->tf_select = 0; }
      }
      wv_sync();
   }
   FOR_LANES(w, C * NBE) {
      int c = w / NBE, i = w - c * NBE;
      if (i >= start && i < end && iabs(sub32(L->bandLogE[i + c * NBE], L->oldBandE[i + c * NBE])) < GC(2.f))
         L->bandLogE[i + c * NBE] -= mult16_32_q15(QC16(0.25f, 15), gst->energyError[i + c * NBE]);
   }
   wv_sync();
   K_PHASE(9);
#ifdef K_DUMP_ENABLED
   LANE0 {   /* same words as the tapped reference shim of the tests */
      i32 w[12 + 84]; int n = 0;
      w[n++] = start; w[n++] = end; w[n++] = C; w[n++] = LM; w[n++] = sh->total_bits; w[n++] = sh->nbAvailableBytes; w[n++] = sh->force_intra; w[n++] = st->delayedIntra; w[n++] = sh->complexity >= 4;
      w[n++] = L->ec.nbits_total - ec_ilog(L->ec.rng); w[n++] = (i32)L->ec.rng; w[n++] = 0;
      for (int i = 0; i < 42; i++) w[n++] = i < C * 21 ? L->bandLogE[i] : 0;
      for (int i = 0; i < 42; i++) w[n++] = i < C * 21 ? L->oldBandE[i] : 0;
      K_DUMP("coarse_in", w, 4 * n);
   }
#endif
   coarse_energy_wave(L);                 /* both hypotheses at once on lanes 0 and 1 (celt_enc_energy.h) */
   LANE0 {
      EC_BEGIN;
      tf_encode_l0(L, EC_PASS);
      sh->r[2] = k_ec_tell(EC_PASS) + 4 <= sh->total_bits;
      EC_END;
   }
   wv_sync();
   K_DUMP("tf_res", L->tf_res, 84); K_DUMP("oldBandE_c", L->oldBandE, 168); K_DUMP("error_c", L->error, 168); K_DUMPI("rng_tf", L->ec.rng); K_DUMPI("tell_tf", ec_tell_frac_lds(&L->ec));
   K_PHASE(10);
   stage_coded_bins_wave(L);              /* BC is free between the coarse-energy rollback and the PVQ: the analyses below read the spectrum from LDS */
   if (sh->r[2]) {
      if (sh->lfe) { LANE0 { st->tapset_decision = 0; st->spread_decision = 2; } wv_sync(); }                                                  /* :2305 */
      else if (HYB) { LANE0 st->spread_decision = sh->complexity == 0 ? 0 : sh->isTransient ? 2 : 3; wv_sync(); }                                      /* :2309 SPREAD_NONE / NORMAL / AGGRESSIVE */
      else if (sh->shortBlocks || sh->complexity < 3 || sh->nbAvailableBytes < 10 * C) { LANE0 st->spread_decision = sh->complexity == 0 ? 0 : 2; wv_sync(); }
      else spreading_decision_wave(L, sh->pf_on && !sh->shortBlocks);
      LANE0 { EC_BEGIN; k_ec_enc_icdf(EC_PASS, st->spread_decision, k_spread_icdf, 5); EC_END; }
   } else { LANE0 st->spread_decision = 2; }
   wv_sync();
   K_DUMPI("spread", st->spread_decision); K_DUMPI("tapset", st->tapset_decision);
   LANE0 {
      EC_BEGIN;
      if (sh->lfe) L->offsets[0] = imin(8, sh->effectiveBytes / 3);                      /* for LFE everything interesting is in the first band (:2352) */
      k_init_caps(L->cap, LM, C);
      int dynalloc_logp = 6;
      i32 total_bits = sh->total_bits << BITRES, total_boost = 0, tell = k_ec_tell_frac(EC_PASS);
      for (int i = start; i < end; i++) {
         int width = C * (ct_eBands[i + 1] - ct_eBands[i]) << LM;
         int quanta = imin(width << BITRES, imax(6 << BITRES, width));
         int dynalloc_loop_logp = dynalloc_logp, boost = 0, j;
         for (j = 0; tell + (dynalloc_loop_logp << BITRES) < total_bits - total_boost && boost < L->cap[i]; j++) {
            int flag = j < L->offsets[i];
            k_ec_enc_bit_logp(EC_PASS, flag, dynalloc_loop_logp);
            tell = k_ec_tell_frac(EC_PASS);
            if (!flag) break;
            boost += quanta;
            total_boost += quanta;
            dynalloc_loop_logp = 1;
         }
         if (j) dynalloc_logp = imax(2, dynalloc_logp - 1);
         L->offsets[i] = boost;
      }
      sh->total_boost = total_boost;
      sh->r[3] = tell;
      EC_END;
   }
   wv_sync();
   {
      int ds = 0;
      if (C == 2 && LM != 0) ds = stereo_analysis_wave(L);
      LANE0 {
         sh->dual_stereo = ds;
         if (C == 2) {
            const i16 intensity_thresholds[21] = {1, 2, 3, 4, 5, 6, 7, 8, 16, 24, 36, 44, 50, 56, 62, 67, 72, 79, 88, 106, 134};
            const i16 intensity_histeresis[21] = {1, 1, 1, 1, 1, 1, 1, 2, 2, 2, 2, 2, 2, 2, 3, 3, 4, 5, 6, 8, 8};
            st->intensity = hysteresis_decision((i16)(sh->equiv_rate / 1000), intensity_thresholds, intensity_histeresis, 21, st->intensity);
            st->intensity = imin(end, imax(start, st->intensity));
         }
         sh->alloc_trim = 5;
         sh->r[4] = sh->r[3] + (6 << BITRES) <= (sh->total_bits << BITRES) - sh->total_boost;
      }
      wv_sync();
   }
   if (sh->r[4]) {
      if (HYB || sh->lfe) { LANE0 { st->stereo_saving = 0; sh->alloc_trim = 5; } }             /* start > 0 || lfe (celt_encoder.c:2412) */
      else alloc_trim_analysis_wave(L, &gst->analysis);
      LANE0 { EC_BEGIN; k_ec_enc_icdf(EC_PASS, sh->alloc_trim, k_trim_icdf, 7); EC_END; }
      wv_sync();
   }
   K_DUMPI("alloc_trim", sh->alloc_trim); K_DUMPI("dual_stereo", sh->dual_stereo); K_DUMPI("intensity", st->intensity); K_DUMP("offsets2", L->offsets, 84); K_DUMPI("rng_trim", L->ec.rng);

   K_PHASE(11);
   /* ---- VBR target, bit allocation, fine energy (lane 0) ---- */
   LANE0 {
      EC_BEGIN;
      i32 tell = k_ec_tell_frac(EC_PASS), total_boost = sh->total_boost, vbr_rate = sh->vbr_rate;
      int nbCompressedBytes = sh->nbCompressedBytes, nbAvailableBytes, silence = sh->silence;
      i32 min_allowed = ((tell + total_boost + (1 << (BITRES + 3)) - 1) >> (BITRES + 3)) + 2;
      if (HYB) min_allowed = imax(min_allowed, (sh->tell0_frac + (37 << BITRES) + total_boost + (1 << (BITRES + 3)) - 1) >> (BITRES + 3));   /* room to signal a redundant frame (:2433) */
      if (vbr_rate > 0) {
         i16 alpha;
         i32 delta, target, base_target;
         int lm_diff = 3 - LM;
         nbCompressedBytes = imin(nbCompressedBytes, 1275 >> (3 - LM));
         base_target = HYB ? imax(0, vbr_rate - ((9 * C + 4) << BITRES)) : vbr_rate - ((40 * C + 20) << BITRES);
         if (sh->constrained_vbr) base_target += (st->vbr_offset >> lm_diff);
         if (!HYB) target = compute_vbr_l0(L, base_target, &gst->analysis);
         else {                                                                                       /* :2463-2475 */
            target = base_target;
            if (sh->silk_offset < 100) target += 12 << BITRES >> (3 - LM);
            if (sh->silk_offset > 100) target -= 18 << BITRES >> (3 - LM);
#ifdef OA_DBG_VBR
            fprintf(stderr, "EMU q14 %d | vbr_rate %d tell %d off %d nbc %d boost %d min_allowed %d LM %d base %d\n", (int)((i16)sh->tf_estimate - QC16(.25f, 14)), vbr_rate, tell, sh->silk_offset, nbCompressedBytes, total_boost, min_allowed, LM, base_target);
#endif
            target += (i32)mult16_16_q14((i16)sh->tf_estimate - QC16(.25f, 14), (50 << BITRES));
            if ((i16)sh->tf_estimate > QC16(.7f, 14)) target = imax(target, 50 << BITRES);
         }
         target = target + tell;
         nbAvailableBytes = (target + (1 << (BITRES + 2))) >> (BITRES + 3);
         nbAvailableBytes = imax(min_allowed, nbAvailableBytes);
         nbAvailableBytes = imin(nbCompressedBytes, nbAvailableBytes);
         delta = target - vbr_rate;
         target = nbAvailableBytes << (BITRES + 3);
         if (silence) { nbAvailableBytes = 2; target = 2 * 8 << BITRES; delta = 0; }
         if (st->vbr_count < 970) { st->vbr_count++; alpha = (i16)fx_rcp(shl32((i32)(st->vbr_count + 20), 16)); }
         else alpha = QC16(.001f, 15);
         if (sh->constrained_vbr) st->vbr_reservoir += target - vbr_rate;
         if (sh->constrained_vbr) {
            st->vbr_drift += (i32)mult16_32_q15(alpha, (delta * (1 << lm_diff)) - st->vbr_offset - st->vbr_drift);
            st->vbr_offset = -st->vbr_drift;
         }
         if (sh->constrained_vbr && st->vbr_reservoir < 0) {
            int adjust = (-st->vbr_reservoir) / (8 << BITRES);
            nbAvailableBytes += silence ? 0 : adjust;
            st->vbr_reservoir = 0;
         }
         nbCompressedBytes = imin(nbCompressedBytes, nbAvailableBytes);
         k_ec_enc_shrink(EC_PASS, nbCompressedBytes);
      }
      sh->nbCompressedBytes = nbCompressedBytes;
      i32 bits = (((i32)nbCompressedBytes * 8) << BITRES) - (i32)k_ec_tell_frac(EC_PASS) - 1;
      int anti_collapse_rsv = sh->isTransient && LM >= 2 && bits >= ((LM + 2) << BITRES) ? (1 << BITRES) : 0;
      bits -= anti_collapse_rsv;
      sh->anti_collapse_rsv = anti_collapse_rsv;
      int signalBandwidth = end - 1;
      if (gst->analysis.valid) {                                                                       /* :2610-2624: no bits above what the analysis detected (and the rate deserves) */
         const i32 er = sh->equiv_rate;
         const int min_bandwidth = er < (i32)32000 * C ? 13 : er < (i32)48000 * C ? 16 : er < (i32)60000 * C ? 18 : er < (i32)80000 * C ? 19 : 20;
         signalBandwidth = imax(gst->analysis.bandwidth, min_bandwidth);
      }
      if (sh->lfe) signalBandwidth = 1;
#ifdef K_DUMP_ENABLED
      {  /* tap_alloc */
         i32 w[12 + 42]; int n = 0;
         w[n++] = start; w[n++] = end; w[n++] = sh->alloc_trim; w[n++] = st->intensity; w[n++] = sh->dual_stereo; w[n++] = bits; w[n++] = C; w[n++] = LM; w[n++] = k_ec_tell(EC_PASS); w[n++] = (i32)e->rng; w[n++] = st->lastCodedBands; w[n++] = signalBandwidth;
         for (int i = 0; i < 21; i++) w[n++] = L->offsets[i];
         for (int i = 0; i < 21; i++) w[n++] = L->cap[i];
         K_DUMP("alloc_in", w, 4 * n);
      }
#endif
      sh->bits = bits; sh->signalBandwidth = signalBandwidth;
      EC_END;
   }
   wv_sync();
   AN_TIC();
   {
      const int coded = oa_allocate_bits_wave<true>(&L->ec, L->packet + 1, L->scr, start, end, L->offsets, L->cap, sh->alloc_trim, &st->intensity, &sh->dual_stereo, sh->bits, &sh->balance,
            L->pulses, L->fine_quant, L->fine_priority, C, LM, st->lastCodedBands, sh->signalBandwidth, sh->r + 6);
      LANE0 {
         sh->codedBands = coded;
         if (st->lastCodedBands) st->lastCodedBands = imin(st->lastCodedBands + 1, imax(st->lastCodedBands - 1, coded));
         else st->lastCodedBands = coded;
      }
   }
   AN_TOC(22);
   fine_energy_wave(L);
   wv_sync();
   AN_TOC(23);
   K_DUMPI("nbCompressedBytes", sh->nbCompressedBytes); K_DUMPI("codedBands", sh->codedBands); K_DUMPI("balance", sh->balance); K_DUMP("pulses", L->pulses, 84); K_DUMP("fine_quant", L->fine_quant, 84); K_DUMP("fine_priority", L->fine_priority, 84); K_DUMPI("rng_fine", L->ec.rng);

   K_PHASE(12);
   /* ---- PVQ residual ---- */
   if (cut && LM >= 2) { celt_cut_dump(L, cut, xcut != nullptr); return OA_CUT; }                /* (frames under 10 ms have bands of one and two coefficients: the four-streams-per-wave stage does not take those) */
   if constexpr (NOPVQ) return 0;                                               /* (the pipeline's front kernel: its calls are single frames of 10 / 20 ms, every frame that gets here is cut) */
   else {
   quant_all_bands_wave(L, sh->shortBlocks, st->spread_decision, sh->dual_stereo, st->intensity,
         sh->nbCompressedBytes * (8 << BITRES) - sh->anti_collapse_rsv, sh->balance, sh->codedBands, sh->complexity, sh->disable_inv,
         journal /* the stream's still-unwritten output slot doubles as the theta-RDO byte journal */);
   K_DUMPI("rng_pvq", L->ec.rng); K_DUMP("collapse", L->collapse_masks, 42);
   celt_encode_core_tail<HYB>(L, gst);
   return 0;
   }
}
/* celt_encode_with_ec after quant_all_bands (celt_encoder.c:2680-2830) */
template <bool HYB> WV_DEV void celt_encode_core_tail(WV_LDS FrameLds *L, OaEncState *gst)
{
   WV_LDS FrameShared *sh = &L->sh;
   WV_LDS OaEncScalars *st = &L->st;
   const int CC = sh->CC, C = sh->C, start = sh->start, end = sh->end;

   K_PHASE(13);
   /* ---- finalise (lane 0) ---- */
   LANE0 {
      EC_BEGIN;
      if (sh->anti_collapse_rsv > 0) k_ec_enc_bits(EC_PASS, st->consec_transient < 2, 1);
      sh->r[5] = (i32)energy_finalise_emit_l0(L, EC_PASS, sh->nbCompressedBytes * 8 - k_ec_tell(EC_PASS));
      EC_END;
   }
   energy_finalise_apply_wave(L, (u32)wv_uni(sh->r[5]));
   LANE0 {
      EC_BEGIN;
      const int nbCompressedBytes = sh->nbCompressedBytes, isTransient = sh->isTransient, silence = sh->silence;
      if (silence) for (int i = 0; i < C * NBE; i++) L->oldBandE[i] = -GC(28.f);
      st->prefilter_period = sh->pitch_index;
      st->prefilter_gain = (i16)sh->gain1;
      st->prefilter_tapset = sh->prefilter_tapset;
      if (CC == 2 && C == 1) for (int i = 0; i < NBE; i++) L->oldBandE[NBE + i] = L->oldBandE[i];
      for (int c = 0; c < CC; c++) {
         for (int i = 0; i < start; i++) L->oldBandE[c * NBE + i] = 0;
         for (int i = end; i < NBE; i++) L->oldBandE[c * NBE + i] = 0;
      }
      if (isTransient || sh->transient_got_disabled) st->consec_transient++;
      else st->consec_transient = 0;
      st->rng = e->rng;
      k_ec_enc_done(EC_PASS);
      int ret = e->error ? -3 : nbCompressedBytes;
      st->rangeFinal = st->rng;
      if (sh->raw_frame) sh->ret = ret;                                               /* celt_encode_with_ec's own return value: the Opus layer of the caller takes it from here */
      else {
         L->packet[0] |= (u8)sh->toc;
         if (ret >= 0 && k_ec_tell(EC_PASS) > (sh->max_data_bytes - 1) * 8) { L->packet[1] = 0; ret = 1; st->rangeFinal = 0; }
         sh->ret = ret < 0 ? ret : ret + 1;
      }
      EC_END;
   }
   wv_sync();

   /* ---- CELT array state (coalesced) ---- */
   {
      /* oldLogE / oldLogE2 (celt_encoder.c:2783-2803) are only ever updated here: do it straight on the HBM state.
       * oldBandE in LDS still holds the pre-"start/end clearing" values for [start,end) and 0 outside, as the reference has at this point. */
      const int isTr = sh->isTransient, nb = sh->CC * NBE;
      FOR_LANES(i, 2 * NBE) {
         if (i < nb) {
            int bi = i % NBE;
            i32 ob = L->oldBandE[i], l1 = gst->oldLogE[i], l2 = gst->oldLogE2[i];
            if (!isTr) { l2 = l1; l1 = ob; } else l1 = imin(l1, ob);
            if (bi < sh->start || bi >= sh->end) { l1 = l2 = -GC(28.f); }
            gst->oldLogE[i] = l1; gst->oldLogE2[i] = l2;
         }
         gst->oldBandE[i] = L->oldBandE[i];
         /* energyError (celt_encoder.c:2698-2741): cleared for the frame's CC channels, then the clamped residual of the coded bands; channels beyond CC keep theirs */
         if (i < nb) { const int c = i / NBE, bi = i - c * NBE; gst->energyError[i] = c < sh->C && bi >= sh->start && bi < sh->end ? imax(-GC(0.5f), imin(GC(0.5f), L->error[i])) : 0; }
      }
   }
}

/* celt_maxabs16 / compute_frame_energy (src/opus_encoder.c:1080) of n int16 samples in HBM: plain int32 sums, order-free */
/* (n is even and pcm 4-byte aligned -- frame sizes are multiples of 20 samples: two samples per load, eight loads in flight per lane) */
WV_DEV i32 oa_maxabs_wave(const i16 *pcm, int n)
{
   const u32 *p = (const u32 *)pcm; const int n2 = n >> 1;
   i32 m = 0;
   for (int i0 = wv_lane(); i0 < n2; i0 += 8 * WV_WIDTH) {
      u32 v[8];
#pragma unroll
      for (int k = 0; k < 8; k++) { const int i = i0 + k * WV_WIDTH; v[k] = i < n2 ? p[i] : 0u; }
#pragma unroll
      for (int k = 0; k < 8; k++) m = imax(m, imax(iabs((i32)(i16)(v[k] & 0xffffu)), iabs((i32)(i16)(v[k] >> 16))));
   }
   return wv_max(m);
}
WV_DEV i32 oa_frame_energy_wave(const i16 *pcm, int len, i32 sample_max)
{
   const int shift = imax(0, (celt_ilog2(1 + sample_max) << 1) + celt_ilog2(len) - 28);
   const u32 *p = (const u32 *)pcm; const int n2 = len >> 1;
   i32 e = 0;
   for (int i0 = wv_lane(); i0 < n2; i0 += 8 * WV_WIDTH) {
      u32 v[8];
#pragma unroll
      for (int k = 0; k < 8; k++) { const int i = i0 + k * WV_WIDTH; v[k] = i < n2 ? p[i] : 0u; }
#pragma unroll
      for (int k = 0; k < 8; k++) { const i32 a = (i16)(v[k] & 0xffffu), b = (i16)(v[k] >> 16); e += (mult16_16(a, a) >> shift) + (mult16_16(b, b) >> shift); }
   }
   e = wv_sum(e);
   e /= len;
   return shl32(e, shift);
}

/* One coded frame of a CELT-only application: opus_encode_frame_native (src/opus_encoder.c:1855) with mode == MODE_CELT_ONLY and no delay compensation.
 * The packet ends up in L->packet; returns its length before CBR padding (1 = DTX / bare TOC), or a negative OPUS_* code. */
WV_DEV int oa_celt_frame_tail(WV_LDS FrameLds *L, int frame_size);
template <bool NOPVQ = false> WV_DEV int oa_celt_frame_native(WV_LDS FrameLds *L, OaStream *gs, const i16 *pcm, int frame_size, int orig_max_data_bytes, u8 *journal, const i32 *tr_pre = nullptr, CeltCont *cut = nullptr)
{
   WV_LDS FrameShared *sh = &L->sh;
   WV_LDS OaEncScalars *st = &L->st;
   const int overlap = OA_OVERLAP, CC = sh->CC;
   {  /* activity for the generalised DTX (:1911-1930): this frame's digital silence (in a multi-frame call every coded frame gets its own flag, :1800, and its own
       * analysis record, :1796), else the analysis' verdict, else the frame's energy against the tracked peak.  sh->use_dtx: 0 = off, 1 = on, 2 = on but the decision
       * is not taken in this call (oa_encode_frame) */
      int activity = 1;
      if (wv_uni(sh->use_dtx) == 1) {
         const i32 m = oa_maxabs_wave(pcm, frame_size * CC);
         if (m == 0) activity = 0;
         else if (wv_uni(gs->an_info.valid)) {                                                      /* the analysis' activity probability; a loud enough noise frame counts as active (:1916-1924) */
            activity = an_activity_prob_active(&gs->an_info);
            if (!activity) activity = an_loud_noise_active(wv_uni(sh->peak_signal_energy), oa_frame_energy_wave(pcm, frame_size * CC, m));
         }
         else { const i32 noise_energy = oa_frame_energy_wave(pcm, frame_size * CC, m); activity = (i64)wv_uni(sh->peak_signal_energy) < 316 * (i64)half32(noise_energy); }
      }
      if (wv_lane() < (int)(sizeof(OaAnalysisInfo) / 4)) ((i32 *)&gs->st.analysis)[wv_lane()] = ((const i32 *)&gs->an_info)[wv_lane()];   /* CELT_SET_ANALYSIS (:2418) */
      LANE0 { sh->activity = activity; opus_layer_frame(L, &gs->cfg, frame_size, orig_max_data_bytes); }
   }
   wv_sync();
   K_PHASE(0);
   /* ---- Opus layer: dc_reject (+ optional stereo width fade) into int16 staging ---- */
   dc_reject_lanes(L, pcm, frame_size, CC);
   wv_sync();
   if (sh->do_stereo_fade) { stereo_fade_lanes(L, frame_size); wv_sync(); }
   {  /* celt_maxabs over the head and the overlap tail of the frame (celt_encoder.c:1970-1973), at the API rate */
      const WV_LDS i16 *p = L->BC.stage16;
      i16 *gp = L->g->pcm16;
      const int Nf = frame_size, ov = overlap / (sh->upsample > 1 ? sh->upsample : 1);
      i32 a = 0, b = 0;
      FOR_LANES(i, CC * (Nf - ov)) { a = imax(a, iabs((i32)p[i])); gp[i] = p[i]; }                        /* ... and out to the HBM staging the CELT front end reads */
      FOR_LANES(i, CC * ov) { b = imax(b, iabs((i32)p[CC * (Nf - ov) + i])); gp[CC * (Nf - ov) + i] = p[CC * (Nf - ov) + i]; }
      a = wv_max(a); b = wv_max(b);
      LANE0 { sh->r[0] = a; sh->r[1] = b; }
   }
   wv_sync();
   LANE0 celt_prologue(L);
   wv_sync();
   if (sh->skip_celt) {
      /* budget already busted: emit TOC + "PLC" byte (opus_encoder.c:2581-2591) */
      LANE0 { L->packet[0] = (u8)sh->toc; L->packet[1] = 0; st->rangeFinal = 0; sh->ret = 2; }
      wv_sync();
   } else if (celt_encode_core<false, NOPVQ>(L, &gs->st, journal, gs->energy_mask, tr_pre, cut) == OA_CUT) return OA_CUT;
   return oa_celt_frame_tail(L, frame_size);
}
/* opus_encode_frame_native after celt_encode_with_ec */
WV_DEV int oa_celt_frame_tail(WV_LDS FrameLds *L, int frame_size)
{
   WV_LDS FrameShared *sh = &L->sh;
   WV_LDS OaEncScalars *st = &L->st;
   K_PHASE(14);
   LANE0 {   /* the generalised DTX decision (:2565-2576, decide_dtx_mode :1115): after 200 ms without activity the packet is the TOC alone, at most 400 ms in a row.  It is
              * taken only where SILK's own DTX is off (:2565) -- and silk_mode.useDTX = use_dtx && !(analysis valid || digital silence) (:1461; without the float API
              * :1463, digital silence alone), whatever the mode, decided once per CALL: a CELT-only call that is neither analysed nor silent never counts towards DTX */
      if (sh->ret >= 0) {
         if (sh->use_dtx == 1) {
            int dtx = 0;
            if (!sh->activity) {
               sh->nb_no_activity_ms_Q1 += 2 * 1000 * frame_size / sh->Fs;
               if (sh->nb_no_activity_ms_Q1 > 10 * 20 * 2) { if (sh->nb_no_activity_ms_Q1 <= (10 + 20) * 20 * 2) dtx = 1; else sh->nb_no_activity_ms_Q1 = 10 * 20 * 2; }
            } else sh->nb_no_activity_ms_Q1 = 0;
            if (dtx) { st->rangeFinal = 0; L->packet[0] = (u8)sh->toc; sh->ret = 1; sh->no_pad = 1; }
         } else sh->nb_no_activity_ms_Q1 = 0;
      }
   }
   wv_sync();
   return wv_uni(sh->ret);
}

template <bool NOPVQ = false> WV_DEV int oa_encode_frame(WV_LDS FrameLds *L, OaStream *gs, const i16 *pcm, int frame_size, int max_data_bytes,
      u8 *out, int out_cap, i32 *len_out, u32 *rng_out, const i32 *apcm = nullptr, int analysis_frame_size = 0 /* samples per channel pcm (and apcm) hold: >= frame_size, the caller's look-ahead (:2662-2690); 0 = frame_size */,
      const i32 *tr_pre = nullptr /* the transient pre-pass's record of this stream for this call's frame (celt_enc_front.h: ct_transient_tile), or NULL */,
      CeltCont *cut = nullptr /* the stream's continuation record: a single-frame call of 10 / 20 ms stops before the PVQ (see OA_CUT above) and returns 1 */);
WV_DEV void oa_encode_frame_tail(WV_LDS FrameLds *L, OaStream *gs, int result, i32 *len_out, u32 *rng_out);
template <bool NOPVQ> WV_DEV int oa_encode_frame(WV_LDS FrameLds *L, OaStream *gs, const i16 *pcm, int frame_size, int max_data_bytes,
      u8 *out, int out_cap, i32 *len_out, u32 *rng_out, const i32 *apcm, int analysis_frame_size, const i32 *tr_pre, CeltCont *cut)
{
   WV_LDS FrameShared *sh = &L->sh;
   WV_LDS OaEncScalars *st = &L->st;
   K_PHASE_BEGIN();
   /* ---- load persistent state (coalesced) ---- */
   {
      const i32 *g = (const i32 *)&gs->st.s;
      WV_LDS i32 *d = (WV_LDS i32 *)st;
      FOR_LANES(i, (int)(sizeof(OaEncScalars) / 4)) d[i] = g[i];
      FOR_LANES(i, 2 * NBE) L->oldBandE[i] = gs->st.oldBandE[i];
   }
   wv_sync();
   const int CC = gs->cfg.channels;
   LANE0 {
      sh->Fs = gs->Fs ? gs->Fs : 48000; sh->use_dtx = gs->use_dtx; sh->nb_no_activity_ms_Q1 = gs->nb_no_activity_ms_Q1; sh->peak_signal_energy = gs->peak_signal_energy;
      sh->prev_framesize = gs->prev_framesize; sh->lfe = gs->cfg.lfe; sh->energy_mask_on = gs->energy_mask_on;
   }
   const i32 sample_max = oa_maxabs_wave(pcm, frame_size * CC);
   const int Fs = wv_uni(gs->Fs) ? wv_uni(gs->Fs) : 48000, float_api = !wv_uni(gs->analysis_off);
   LANE0 {
      sh->is_silence = sample_max == 0;                                                  /* is_digital_silence (:1246) */
      if (gs->voice_ratio_seq != st->voice_ratio_seq) { st->voice_ratio = gs->voice_ratio; st->voice_ratio_seq = gs->voice_ratio_seq; }   /* OPUS_SET_VOICE_RATIO through a batch ctl */
   }
   /* the tonality / music analysis of the call's input (:1247-1264; the FIXED_POINT build runs it at complexity 10 only); a call the reference turns away before
    * that (:1231) leaves it alone */
   if (!(imin(1276 * 6, max_data_bytes) == 1 && Fs == frame_size * 10)) {
      if (float_api && wv_uni(gs->cfg.complexity) >= 10 && Fs >= 16000) {
         LANE0 { gs->an_read_pos_bak = gs->an.read_pos; gs->an_read_subframe_bak = gs->an.read_subframe; }
         an_run_analysis_wave((WV_LDS AnLds *)&L->BC, &gs->an, pcm, apcm, analysis_frame_size > frame_size ? analysis_frame_size : frame_size, frame_size, CC, Fs, imin(wv_uni(gs->cfg.input_depth) ? wv_uni(gs->cfg.input_depth) : 16, wv_uni(gs->cfg.lsb_depth)),
               (i32 *)L->g->X, &gs->an_info);
      } else {
         const int was_initialized = wv_uni(gs->an.initialized);
         wv_sync();                                                                       /* (every lane has read the flag before any lane clears it) */
         if (was_initialized) { i32 *z = (i32 *)&gs->an; FOR_LANES(i, (int)(sizeof(OaAnalysis) / 4)) z[i] = 0; }                /* tonality_analysis_reset (:1262) */
         LANE0 { gs->an_info.valid = 0; gs->an_read_pos_bak = -1; }
      }
   }
   wv_sync();
   LANE0 sh->use_dtx = gs->use_dtx ? ((gs->an_info.valid || sh->is_silence) ? 1 : 2) : 0;          /* silk_mode.useDTX (:1461), with the call's analysis record and the call's silence flag */
   if (sample_max != 0 && (!wv_uni(gs->an_info.valid) || an_activity_prob_above(&gs->an_info))) {   /* peak signal energy tracker (:1310-1320): tracked whatever the DTX setting is now -- it can be switched on later */
      const i32 en = oa_frame_energy_wave(pcm, frame_size * CC, sample_max);
      LANE0 sh->peak_signal_energy = imax(mult16_32_q15(QC16(0.999f, 15), sh->peak_signal_energy), en);
   }
   LANE0 opus_layer_decide(L, &gs->cfg, frame_size, max_data_bytes, gs->signal_type, float_api, &gs->an_info);
   wv_sync();
   int result;
   if (sh->plc_frame) {
      result = sh->plc_frame == 2 ? wv_uni(sh->ret) : emit_packet_wave(L, out, sh->ret, gs->cfg.use_vbr ? 0 : sh->call_max_data_bytes, out_cap);
      LANE0 st->rangeFinal = 0;
   } else if (wv_uni(sh->nb_frames) == 1) {
      const int ret = oa_celt_frame_native<NOPVQ>(L, gs, pcm, frame_size, wv_uni(sh->call_max_data_bytes), out, tr_pre, cut);
      if (ret == OA_CUT) return 1;
      const int pad_to = (!gs->cfg.use_vbr && ret > 0 && !wv_uni(sh->no_pad)) ? wv_uni(sh->call_max_data_bytes) : 0;      /* apply_padding (:2646) */
      result = ret < 0 ? ret : emit_packet_wave(L, out, ret, pad_to, out_cap);
   } else if constexpr (NOPVQ) result = -3;                /* (never: a call of the pipeline's front kernel is one frame) */
   else {
      /* ---- 40-120 ms: 20 ms frames staged in the output slot, then framed as one packet (:1757-1838, opus_multiframe.h) ---- */
      const int nb_frames = wv_uni(sh->nb_frames), efs = wv_uni(sh->enc_frame_size), max_len_sum = wv_uni(sh->max_len_sum), repacketize_len = wv_uni(sh->repacketize_len);
      const int Fs = wv_uni(sh->Fs);
      int tot_size = 0, dtx_count = 0, err = 0, staged = 0;
      LANE0 L->mf.n = nb_frames;
      if (OA_MF_HEADROOM + imin(max_len_sum, 1276 * nb_frames) > out_cap) err = -2;
      const int an_bak = wv_uni(gs->an_read_pos_bak);
      if (an_bak != -1) { LANE0 { gs->an.read_pos = an_bak; gs->an.read_subframe = gs->an_read_subframe_bak; } }   /* the analysis is read one coded frame at a time (:1727-1735) */
      for (int i = 0; i < nb_frames && !err; i++) {
         int curr_max = imin(bitrate_to_bits(wv_uni(sh->call_bitrate), Fs, efs) / 8, max_len_sum / nb_frames);
         curr_max = imin(max_len_sum - tot_size, curr_max);
         if (an_bak != -1) an_get_info_wave((WV_LDS AnLds *)&L->BC, &gs->an, &gs->an_info, efs, Fs);                 /* (:1796-1800) */
         const int tmp_len = oa_celt_frame_native(L, gs, pcm + (size_t)i * CC * efs, efs, curr_max, out + OA_MF_HEADROOM + staged);
         if (tmp_len < 0) { err = -3; break; }
         if (tmp_len == 1) dtx_count++;
         wv_sync();
         if (i > 0 && ((wv_uni(L->mf.toc) ^ L->packet[0]) & 0xFC)) { err = -3; break; }
         LANE0 { if (i == 0) L->mf.toc = L->packet[0]; L->mf.len[i] = tmp_len - 1; }
         FOR_LANES(k, tmp_len - 1) out[OA_MF_HEADROOM + staged + k] = L->packet[1 + k];
         wv_sync();
         staged += tmp_len - 1; tot_size += tmp_len;
      }
      if (err) result = err;
      else { result = oa_multiframe_assemble_wave(&L->mf, L->packet, out, repacketize_len, !gs->cfg.use_vbr && dtx_count != nb_frames, out_cap); if (result < 0) result = -3; }
   }
   oa_encode_frame_tail(L, gs, result, len_out, rng_out);
   return 0;
}
/* what the back kernel of the pipeline runs on a frame that was cut: the rest of celt_encode_with_ec, of opus_encode_frame_native and of opus_encode_native */
WV_DEV void oa_encode_frame_back(WV_LDS FrameLds *L, OaStream *gs, int frame_size, u8 *out, int out_cap, i32 *len_out, u32 *rng_out)
{
   WV_LDS FrameShared *sh = &L->sh;
   celt_encode_core_tail<false>(L, &gs->st);
   const int ret = oa_celt_frame_tail(L, frame_size);
   const int pad_to = (!gs->cfg.use_vbr && ret > 0 && !wv_uni(sh->no_pad)) ? wv_uni(sh->call_max_data_bytes) : 0;      /* apply_padding (:2646) */
   const int result = ret < 0 ? ret : emit_packet_wave(L, out, ret, pad_to, out_cap);
   oa_encode_frame_tail(L, gs, result, len_out, rng_out);
}
WV_DEV void oa_encode_frame_tail(WV_LDS FrameLds *L, OaStream *gs, int result, i32 *len_out, u32 *rng_out)
{
   WV_LDS FrameShared *sh = &L->sh;
   WV_LDS OaEncScalars *st = &L->st;
   /* ---- store lengths + state (coalesced) ---- */
   {
      LANE0 {
         *len_out = result; *rng_out = result < 0 ? 0 : st->rangeFinal;
         gs->nb_no_activity_ms_Q1 = sh->nb_no_activity_ms_Q1; gs->peak_signal_energy = sh->peak_signal_energy; if (!sh->plc_frame) gs->prev_framesize = sh->prev_framesize;
      }
      i32 *g = (i32 *)&gs->st.s;
      const WV_LDS i32 *d = (const WV_LDS i32 *)st;
      FOR_LANES(i, (int)(sizeof(OaEncScalars) / 4)) g[i] = d[i];
   }
   K_PHASE(15);
}
#endif

/* silk_enc_frame.h — SILK encoder: entropy coding, stereo front end, frame driver with the rate-control loop, and silk_Encode
 * (rows a16, a22 encoder half, a18 stereo of SURVEY §8).
 *
 *   se_encode_indices     silk_encode_indices     silk/encode_indices.c:35
 *   se_encode_pulses      silk_encode_pulses, silk_shell_encoder, silk_encode_signs   silk/encode_pulses.c:60, shell_coder.c:78, code_signs.c:41
 *   se_stereo_*           silk_stereo_LR_to_MS / _find_predictor / _quant_pred / _encode_pred   silk/stereo_LR_to_MS.c:35, stereo_find_predictor.c:35, stereo_quant_pred.c:35, stereo_encode_pred.c:35
 *   se_encode_frame_wave  silk_encode_frame_FIX   silk/fixed/encode_frame_FIX.c:85
 *   silk_encode_wave      silk_Encode             silk/enc_API.c:150
 * In-band FEC: silk_LBRR_encode_FIX (encode_frame_FIX.c:392) inside se_encode_frame_wave, the LBRR store in HBM (OaSilkLbrr), coded at the head of the
 * next packet (enc_API.c:355-406).  DTX: the no-speech counter / inDTX logic of silk_encode_do_VAD_FIX and the empty payload of enc_API.c:560.
 * Prefill (prefillFlag 1: after a CELT -> SILK switch, 2: at a SILK bandwidth switch; enc_API.c:210-240, :563-571): the encoder is reset, 10 ms of input fill its
 * buffers at complexity 0 and nothing is coded. */
#ifndef OPUS_AMD_SILK_ENC_FRAME_H
#define OPUS_AMD_SILK_ENC_FRAME_H

/* ---- LDS working set ---- */
struct SeAnaLds {                                      /* analysis phases */
   i16 res_pitch[32 + 320 + 320 + 8];
   i32 w32[32];
   i16 A_Q12s[16];
   union {
      /* pitch analysis, noise shaping analysis.  The windowed-signal buffers and the pitch estimator's working set take turns: find_pitch_lags is through with Wsig / xx (its
       * autocorrelation) before the estimator runs -- its reflection coefficients sit in pitch.d_srch, behind both -- and the noise shaping analysis fills them again afterwards */
      struct { union { struct { i16 Wsig[384 + 8], xx[384 + 8]; }; PitchLdsCore pitch; }; } a;
      struct { SeLpcWork W; } p;                                         /* prediction coefficients */
   } u;
};
static_assert(offsetof(PitchLdsCore, d_srch) >= 2 * (384 + 8) * sizeof(i16), "find_pitch_lags keeps its reflection coefficients behind the windowed-signal buffers");
struct SeQuantLds {                                    /* quantiser + rate loop */
   SeNsqLds N;
   EcCtx ec_copy, ec_copy2;
   OaSilkEncIndices ix_lbrr; i8 pulses_lbrr[320];      /* silk_LBRR_encode_FIX output before it goes to the HBM store */
};
/* rate-control loop snapshots (silk/fixed/encode_frame_FIX.c:108-118 sNSQ_copy[2], ec_buf_copy): per-stream HBM scratch, touched once per frame in VBR */
struct SeRateScratch { OaSilkNsqState nsq_copy[2]; u8 ec_buf_copy[1280]; };
struct SeStereoLds { i16 side[322 + 6], LP_mid[320], HP_mid[320], LP_side[320], HP_side[320]; };
struct SilkEncLds {
   i32 st_off, hdr_[3];                                /* where this kernel keeps the staged state: se_st(S).  offsetof(SilkEncLds, st) everywhere but in the split path's front kernel, which packs it
                                                        * behind the part of the phase union its stages use (SE_FRONT_ST_OFF); set by the kernel before the call opens.  These 16 bytes are not part
                                                        * of any arena that borrows the rest (the Opus layer's CELT passes: SH_F, opus_enc_sh.h) */
   SeEncCtrl ctl;
   SeRsLds rs;
   i32 r[16];                                          /* lane-0 hand-off words */
   i32 stk[104];                                       /* lane-0 working arrays (run-time indexed private arrays would live in scratch = HBM); the temporary resampler of a rate switch (se_setup_resamplers: 99 words) */
   alignas(16) union { SeAnaLds a; SeQuantLds q; SeStereoLds s; i16 vadX[448]; i16 rs_tmp[45 * 48 + 8]; i16 pcm_stage[1920 + 8]; i32 rs_ring[36 + 480 + 4]; OaSilkLbrr lbrr; } u;
   OaSilkEnc st;                                       /* persistent state, staged; LAST: a mono batch allocates LDS only up to st.tail[1], the split path's front kernel only up to st.ch[channels] */
};
#define SE_LDS_BYTES(channels) (sizeof(SilkEncLds) - ((channels) == 1 ? sizeof(OaSilkEncTail) : 0))
/* the split path's front kernel (opus_sh_split.h) runs everything up to silk_process_gains: of the phase union it needs the analysis working set and the smaller members in front of it, not the
 * quantiser's, so its copy of the state starts SE_FRONT_U_BYTES behind the union's start */
#define SE_FRONT_U_BYTES ((sizeof(SeAnaLds) + 15) & ~(size_t)15)
#define SE_FRONT_ST_OFF (offsetof(SilkEncLds, u) + SE_FRONT_U_BYTES)
#define SE_FRONT_LDS_BYTES(channels) (SE_FRONT_ST_OFF + offsetof(OaSilkEnc, ch) + (size_t)(channels) * sizeof(OaSilkEncChannel))
static_assert(sizeof(SeStereoLds) <= SE_FRONT_U_BYTES && sizeof(i16) * (45 * 48 + 8) <= SE_FRONT_U_BYTES && sizeof(i16) * (1920 + 8) + 2 * sizeof(int16_t) * (SE_MAX_FRAME + 2) <= SE_FRONT_U_BYTES &&
              sizeof(OaSilkLbrr) <= SE_FRONT_U_BYTES && sizeof(i32) * (36 + 480 + 4) <= SE_FRONT_U_BYTES, "the front kernel's phase union holds every member its stages use (the input buffers sit at its end)");
WV_DEV WV_LDS OaSilkEnc *se_st(WV_LDS SilkEncLds *S) { return (WV_LDS OaSilkEnc *)((WV_LDS char *)S + wv_uni(S->st_off)); }
#define SE_STATE_LITE_WORDS(channels) ((int)((offsetof(OaSilkEnc, ch) + (size_t)(channels) * sizeof(OaSilkEncChannel)) / 4))
#define SE_TAIL_WORDS ((int)(sizeof(OaSilkEncTail) / 4))
#define SE_INBUF_WORDS ((int)(sizeof(int16_t) * (SE_MAX_FRAME + 2) / 4))
#define SE_STATE_WORDS(channels) (SE_STATE_LITE_WORDS(channels) + (channels) * (SE_INBUF_WORDS + SE_TAIL_WORDS))       /* words a frame-step moves each way */
/* the SILK state between the stream record and the wave's LDS (either direction): header + the channels in use, and their tails when the kernel holds them */
template <class PD, class PS> WV_DEV void se_state_copy_wave(PD d, PS s, int channels, int with_tail)
{
   wv_copy_batched(d, s, SE_STATE_LITE_WORDS(channels));
   if (with_tail) {
      const int o = (int)(offsetof(OaSilkEnc, tail) / 4), b = (int)(offsetof(OaSilkEnc, inbuf) / 4);
      wv_copy_batched(d + b, s + b, channels * SE_INBUF_WORDS);
      wv_copy_batched(d + o, s + o, channels * SE_TAIL_WORDS);
   }
}
/* a channel's input buffer: in the staged state, or -- FRONT: the split path's front kernel, for which it is scratch between the resampler and the frame heads -- at the end of
 * the phase union (behind everything the stages that run while it is live put there: the resampler ring, the stereo work arrays, the VAD scratch) */
template <int FRONT> WV_DEV WV_LDS i16 *se_inbuf(WV_LDS SilkEncLds *S, int n)
{
   if (FRONT) return (WV_LDS i16 *)((WV_LDS char *)se_st(S) - (size_t)(2 - n) * sizeof(S->st.inbuf[0]));
   return se_st(S)->inbuf[n];
}
WV_DEV WV_LDS OaSilkEncTail *se_tail(WV_LDS SilkEncLds *S, const WV_LDS OaSilkEncChannel *c) { WV_LDS OaSilkEnc *E = se_st(S); return &E->tail[c == &E->ch[1] ? 1 : 0]; }
/* the quantiser state starts over if someone asked for it since its last use (silk_setup_fs: control_codec.c:241-246; the side channel after mid-only frames: enc_API.c:449-456) */
WV_DEV void se_nsq_apply_reset_wave(WV_LDS SilkEncLds *S, WV_LDS OaSilkEncChannel *c)
{
   wv_sync();
   if (wv_uni(c->nsq_reset_req)) {
      WV_LDS OaSilkNsqState *n = &se_tail(S, c)->nsq; WV_LDS i32 *w = (WV_LDS i32 *)n;
      FOR_LANES(i, (int)(sizeof(OaSilkNsqState) / 4)) w[i] = 0;
      wv_sync();
      LANE0 { n->lagPrev = 100; n->prev_gain_Q16 = 65536; c->nsq_reset_req = 0; }
   }
   wv_sync();
}

/* ---- silk_encode_indices; ix = the frame's own index set, or an LBRR one (encode_LBRR: the type offset is then always >= 2) ---- */
/* C: anything with nb_subfr, predictLPCOrder, fs_kHz, ec_prevSignalType, ec_prevLagIndex (the channel state, or the quantiser kernel's per-stream record) */
template <class C, class IX, class ECB> WV_DEV void se_encode_indices(C c, IX ix, EC_ARGS_G, int condCoding)
{
   const int typeOffset = 2 * ix->signalType + ix->quantOffsetType;
   if (typeOffset >= 2) k_ec_enc_icdf(EC_PASS, typeOffset - 2, sk_type_offset_vad_icdf, 8); else k_ec_enc_icdf(EC_PASS, typeOffset, sk_type_offset_no_vad_icdf, 8);
   if (condCoding == SE_CODE_CONDITIONALLY) k_ec_enc_icdf(EC_PASS, ix->GainsIndices[0], sk_delta_gain_icdf, 8);
   else { k_ec_enc_icdf(EC_PASS, ix->GainsIndices[0] >> 3, &sk_gain_icdf[ix->signalType * 8], 8); k_ec_enc_icdf(EC_PASS, ix->GainsIndices[0] & 7, sk_uniform8_icdf, 8); }
   for (int i = 1; i < c->nb_subfr; i++) k_ec_enc_icdf(EC_PASS, ix->GainsIndices[i], sk_delta_gain_icdf, 8);
   const SdNlsfCb cb = sd_nlsf_cb(c->predictLPCOrder);
   k_ec_enc_icdf(EC_PASS, ix->NLSFIndices[0], &cb.cb1_icdf[(ix->signalType >> 1) * cb.nVectors], 8);
   i32 ec_ix[16], pred_Q8[16];
   sd_nlsf_unpack(ec_ix, pred_Q8, cb, ix->NLSFIndices[0]);
   for (int i = 0; i < cb.order; i++) {
      const int v = ix->NLSFIndices[i + 1];
      if (v >= 4) { k_ec_enc_icdf(EC_PASS, 8, &cb.ec_icdf[ec_ix[i]], 8); k_ec_enc_icdf(EC_PASS, v - 4, sk_nlsf_ext_icdf, 8); }
      else if (v <= -4) { k_ec_enc_icdf(EC_PASS, 0, &cb.ec_icdf[ec_ix[i]], 8); k_ec_enc_icdf(EC_PASS, -v - 4, sk_nlsf_ext_icdf, 8); }
      else k_ec_enc_icdf(EC_PASS, v + 4, &cb.ec_icdf[ec_ix[i]], 8);
   }
   if (c->nb_subfr == 4) k_ec_enc_icdf(EC_PASS, ix->NLSFInterpCoef_Q2, sk_nlsf_interpolation_factor_icdf, 8);
   if (ix->signalType == SE_TYPE_VOICED) {
      int encode_absolute_lagIndex = 1;
      if (condCoding == SE_CODE_CONDITIONALLY && c->ec_prevSignalType == SE_TYPE_VOICED) {
         int delta_lagIndex = ix->lagIndex - c->ec_prevLagIndex;
         if (delta_lagIndex < -8 || delta_lagIndex > 11) delta_lagIndex = 0; else { delta_lagIndex += 9; encode_absolute_lagIndex = 0; }
         k_ec_enc_icdf(EC_PASS, delta_lagIndex, sk_pitch_delta_icdf, 8);
      }
      if (encode_absolute_lagIndex) {
         const i32 hi = ix->lagIndex / (c->fs_kHz >> 1), lo = ix->lagIndex - sk_mulbb(hi, c->fs_kHz >> 1);
         k_ec_enc_icdf(EC_PASS, hi, sk_pitch_lag_icdf, 8);
         k_ec_enc_icdf(EC_PASS, lo, sd_pitch_low_bits_icdf(c->fs_kHz), 8);
      }
      c->ec_prevLagIndex = ix->lagIndex;
      k_ec_enc_icdf(EC_PASS, ix->contourIndex, sd_pitch_contour_icdf(c->fs_kHz, c->nb_subfr), 8);
      k_ec_enc_icdf(EC_PASS, ix->PERIndex, sk_ltp_per_index_icdf, 8);
      const u8 *gicdf = &sk_ltp_gain_icdf[ix->PERIndex == 0 ? 0 : ix->PERIndex == 1 ? 8 : 24];
      for (int k = 0; k < c->nb_subfr; k++) k_ec_enc_icdf(EC_PASS, ix->LTPIndex[k], gicdf, 8);
      if (condCoding == SE_CODE_INDEPENDENTLY) k_ec_enc_icdf(EC_PASS, ix->LTP_scaleIndex, sk_ltpscale_icdf, 8);
   }
   c->ec_prevSignalType = ix->signalType;
   k_ec_enc_icdf(EC_PASS, ix->Seed, sk_uniform4_icdf, 8);
}

/* ---- silk_encode_pulses ---- */
/* {sk_sign_icdf[i], 0}: the two-entry iCDFs silk_encode_signs builds on its stack, tabulated (a private array handed to the range coder would live in scratch memory) */
WV_TABLE uint8_t se_sign_icdf_pairs[84] = { 254, 0, 49, 0, 67, 0, 77, 0, 82, 0, 93, 0, 99, 0, 198, 0, 11, 0, 18, 0, 24, 0, 31, 0, 36, 0, 45, 0, 255, 0, 46, 0, 66, 0, 78, 0, 87, 0, 94, 0, 104, 0, 208, 0, 14, 0, 21, 0, 32, 0, 42, 0, 51, 0, 66, 0, 255, 0, 94, 0, 104, 0, 109, 0, 112, 0, 115, 0, 118, 0, 248, 0, 53, 0, 69, 0, 80, 0, 88, 0, 95, 0, 102, 0 };
/* fixed-size, fully unrolled: the small arrays stay in registers */
template <int LEN> WV_DEV int se_combine_and_check(int *out, const int *in, int max_pulses) { int over = 0;
#pragma unroll
   for (int k = 0; k < LEN; k++) { const int s = in[2 * k] + in[2 * k + 1]; over |= s > max_pulses; out[k] = s; } return over; }
template <class ECB> WV_DEV void se_encode_split(EC_ARGS_G, int p_child1, int p, const u8 *tab) { if (p > 0) k_ec_enc_icdf(EC_PASS, p_child1, &tab[sk_shell_code_table_offsets[p]], 8); }
template <class ECB> WV_DEV void se_shell_encoder(EC_ARGS_G, const int *p0)
{
   int p1[8], p2[4], p3[2], p4[1];
#pragma unroll
   for (int k = 0; k < 8; k++) p1[k] = p0[2 * k] + p0[2 * k + 1];
#pragma unroll
   for (int k = 0; k < 4; k++) p2[k] = p1[2 * k] + p1[2 * k + 1];
#pragma unroll
   for (int k = 0; k < 2; k++) p3[k] = p2[2 * k] + p2[2 * k + 1];
   p4[0] = p3[0] + p3[1];
   se_encode_split(EC_PASS, p3[0], p4[0], sk_shell_code_table3);
   se_encode_split(EC_PASS, p2[0], p3[0], sk_shell_code_table2);
   se_encode_split(EC_PASS, p1[0], p2[0], sk_shell_code_table1);
   se_encode_split(EC_PASS, p0[0], p1[0], sk_shell_code_table0);
   se_encode_split(EC_PASS, p0[2], p1[1], sk_shell_code_table0);
   se_encode_split(EC_PASS, p1[2], p2[1], sk_shell_code_table1);
   se_encode_split(EC_PASS, p0[4], p1[2], sk_shell_code_table0);
   se_encode_split(EC_PASS, p0[6], p1[3], sk_shell_code_table0);
   se_encode_split(EC_PASS, p2[2], p3[1], sk_shell_code_table2);
   se_encode_split(EC_PASS, p1[4], p2[2], sk_shell_code_table1);
   se_encode_split(EC_PASS, p0[8], p1[4], sk_shell_code_table0);
   se_encode_split(EC_PASS, p0[10], p1[5], sk_shell_code_table0);
   se_encode_split(EC_PASS, p1[6], p2[3], sk_shell_code_table1);
   se_encode_split(EC_PASS, p0[12], p1[6], sk_shell_code_table0);
   se_encode_split(EC_PASS, p0[14], p1[7], sk_shell_code_table0);
}
/* wk: 40 words of LDS (per-block sums and shift counts: run-time indexed) */
template <class ECB> WV_DEV void se_encode_pulses(EC_ARGS_G, int signalType, int quantOffsetType, WV_LDS i8 *pulses, int frame_length, WV_LDS i32 *wk)
{
   const int max_pulses_table[4] = {8, 10, 12, 16};
   int iter = frame_length >> 4;
   if (iter * 16 < frame_length) { iter++; for (int i = 0; i < 16; i++) pulses[frame_length + i] = 0; }
   WV_LDS i32 *sum_pulses = wk, *nRshifts = wk + 20;
   for (int i = 0; i < iter; i++) {
      int ap[16];
#pragma unroll
      for (int k = 0; k < 16; k++) ap[k] = iabs((i32)pulses[i * 16 + k]);
      int sh = 0, sum = 0;
      while (1) {
         /* (the reference leaves a partially written level behind when a check fails and overwrites it on the next pass: only the final pass matters) */
         int c8[8], c4[4], c2[2], c1[1];
         int scale_down = se_combine_and_check<8>(c8, ap, max_pulses_table[0]);
         scale_down += se_combine_and_check<4>(c4, c8, max_pulses_table[1]);
         scale_down += se_combine_and_check<2>(c2, c4, max_pulses_table[2]);
         scale_down += se_combine_and_check<1>(c1, c2, max_pulses_table[3]);
         sum = c1[0];
         if (scale_down) { sh++;
#pragma unroll
            for (int k = 0; k < 16; k++) ap[k] >>= 1; } else break;
      }
      sum_pulses[i] = sum; nRshifts[i] = sh;
   }
   i32 minSumBits_Q5 = 2147483647; int RateLevelIndex = 0;
   for (int k = 0; k < 9; k++) {
      const u8 *nBits = &se_pulses_per_block_bits_q5[k * 18];
      i32 sumBits_Q5 = se_rate_levels_bits_q5[(signalType >> 1) * 9 + k];
      for (int i = 0; i < iter; i++) sumBits_Q5 += nRshifts[i] > 0 ? nBits[16 + 1] : nBits[sum_pulses[i]];
      if (sumBits_Q5 < minSumBits_Q5) { minSumBits_Q5 = sumBits_Q5; RateLevelIndex = k; }
   }
   k_ec_enc_icdf(EC_PASS, RateLevelIndex, &sk_rate_levels_icdf[(signalType >> 1) * 9], 8);
   const u8 *cdf = &sk_pulses_per_block_icdf[RateLevelIndex * 18];
   for (int i = 0; i < iter; i++) {
      if (nRshifts[i] == 0) k_ec_enc_icdf(EC_PASS, sum_pulses[i], cdf, 8);
      else {
         k_ec_enc_icdf(EC_PASS, 16 + 1, cdf, 8);
         for (int k = 0; k < nRshifts[i] - 1; k++) k_ec_enc_icdf(EC_PASS, 16 + 1, &sk_pulses_per_block_icdf[9 * 18], 8);
         k_ec_enc_icdf(EC_PASS, sum_pulses[i], &sk_pulses_per_block_icdf[9 * 18], 8);
      }
   }
   for (int i = 0; i < iter; i++) {
      if (sum_pulses[i] > 0) { int ap[16]; const int sh = nRshifts[i];
#pragma unroll
         for (int k = 0; k < 16; k++) ap[k] = iabs((i32)pulses[i * 16 + k]) >> sh;
         se_shell_encoder(EC_PASS, ap); }
   }
   for (int i = 0; i < iter; i++) {
      if (nRshifts[i] > 0) {
         const int nLS = nRshifts[i] - 1;
         for (int k = 0; k < 16; k++) {
            const i32 abs_q = (i8)iabs((i32)pulses[i * 16 + k]);
            for (int j = nLS; j > 0; j--) k_ec_enc_icdf(EC_PASS, (abs_q >> j) & 1, sk_lsb_icdf, 8);
            k_ec_enc_icdf(EC_PASS, abs_q & 1, sk_lsb_icdf, 8);
         }
      }
   }
   {  /* silk_encode_signs */
      const int base = sk_mulbb(7, quantOffsetType + shl32(signalType, 1));
      const int n = (frame_length + 8) >> 4;
      for (int i = 0; i < n; i++) {
         const int p = sum_pulses[i];
         if (p > 0) {
            const u8 *icdf = &se_sign_icdf_pairs[2 * (base + imin(p & 0x1F, 6))];
            for (int j = 0; j < 16; j++) { const int q = pulses[i * 16 + j]; if (q != 0) k_ec_enc_icdf(EC_PASS, (q >> 15) + 1, icdf, 8); }
         }
      }
   }
}

/* ---- stereo ----
 * silk_stereo_find_predictor (stereo_find_predictor.c:34) in two halves: the three sums over the frame (two energies with their two-pass scaling, one cross
 * product whose terms are shifted one by one: all order-free) by the whole wave, the fixed-point tail on lane 0 */
struct SeStereoSums { i32 nrgx, nrgy, corr; int scale; };
WV_DEV SeStereoSums se_stereo_sums_wave(const WV_LDS i16 *x, const WV_LDS i16 *y, int length)
{
   SeStereoSums r; int scale1, scale2;
   se_sum_sqr_shift_wave(&r.nrgx, &scale1, x, length);
   se_sum_sqr_shift_wave(&r.nrgy, &scale2, y, length);
   int scale = imax(scale1, scale2);
   scale = scale + (scale & 1);
   r.nrgy >>= scale - scale2; r.nrgx >>= scale - scale1;
   r.nrgx = imax(r.nrgx, 1);
   i32 part = 0;
   FOR_LANES(i, length) part = add32(part, sk_mulbb(x[i], y[i]) >> scale);
   r.corr = wv_sum(part);
   r.scale = scale;
   return r;
}
WV_DEV i32 se_stereo_predictor_tail(i32 *ratio_Q14, SeStereoSums m, WV_LDS i32 *mid_res_amp_Q0, int smooth_coef_Q16)
{
   i32 nrgx = m.nrgx, nrgy = m.nrgy; const i32 corr = m.corr; int scale = m.scale;
   i32 pred_Q13 = sk_div32_varQ(corr, nrgx, 13);
   pred_Q13 = se_limit(pred_Q13, -(1 << 14), 1 << 14);
   const i32 pred2_Q10 = sk_mulwb(pred_Q13, pred_Q13);
   smooth_coef_Q16 = imax(smooth_coef_Q16, iabs(pred2_Q10));
   scale >>= 1;
   mid_res_amp_Q0[0] = sk_mlawb(mid_res_amp_Q0[0], shl32(se_sqrt_approx(nrgx), scale) - mid_res_amp_Q0[0], smooth_coef_Q16);
   nrgy = sub32(nrgy, shl32(sk_mulwb(corr, pred_Q13), 3 + 1));
   nrgy = add32(nrgy, shl32(sk_mulwb(nrgx, pred2_Q10), 6));
   mid_res_amp_Q0[1] = sk_mlawb(mid_res_amp_Q0[1], shl32(se_sqrt_approx(nrgy), scale) - mid_res_amp_Q0[1], smooth_coef_Q16);
   *ratio_Q14 = sk_div32_varQ(mid_res_amp_Q0[1], imax(mid_res_amp_Q0[0], 1), 14);
   *ratio_Q14 = se_limit(*ratio_Q14, 0, 32767);
   return pred_Q13;
}
WV_DEV void se_stereo_quant_pred(i32 *pred_Q13, WV_LDS i8 *ix /* [2][3] */)
{
   i32 quant_pred_Q13 = 0;
   for (int n = 0; n < 2; n++) {
      i32 err_min_Q13 = 2147483647; int done = 0;
      for (int i = 0; i < 15 && !done; i++) {
         const i32 low_Q13 = sk_stereo_pred_quant_q13[i], step_Q13 = sk_mulwb(sk_stereo_pred_quant_q13[i + 1] - low_Q13, SE_FIX(0.5 / 5, 16));
         for (int j = 0; j < 5; j++) {
            const i32 lvl_Q13 = sk_mlabb(low_Q13, step_Q13, 2 * j + 1), err_Q13 = iabs(pred_Q13[n] - lvl_Q13);
            if (err_Q13 < err_min_Q13) { err_min_Q13 = err_Q13; quant_pred_Q13 = lvl_Q13; ix[n * 3 + 0] = (i8)i; ix[n * 3 + 1] = (i8)j; } else { done = 1; break; }
         }
      }
      ix[n * 3 + 2] = (i8)(ix[n * 3 + 0] / 3);
      ix[n * 3 + 0] = (i8)(ix[n * 3 + 0] - ix[n * 3 + 2] * 3);
      pred_Q13[n] = quant_pred_Q13;
   }
   pred_Q13[0] -= pred_Q13[1];
}
WV_DEV void se_stereo_encode_pred(EC_ARGS, const WV_LDS i8 *ix)
{
   k_ec_enc_icdf(EC_PASS, 5 * ix[2] + ix[3 + 2], sk_stereo_pred_joint_icdf, 8);
   for (int n = 0; n < 2; n++) { k_ec_enc_icdf(EC_PASS, ix[n * 3 + 0], sk_uniform3_icdf, 8); k_ec_enc_icdf(EC_PASS, ix[n * 3 + 1], sk_uniform5_icdf, 8); }
}
/* silk_stereo_LR_to_MS (stereo_LR_to_MS.c:35).  x1 = &inputBuf0[2], x2 = &inputBuf1[2].  Every per-sample pass (mid / side, the 3-tap low / high split, the
 * predicted side signal with its 8 ms parameter ramp -- written in closed form: step n uses start + (n + 1) * delta, exact mod 2^32) and every sum over the frame
 * runs on the whole wave; lane 0 keeps the decision logic in between.  mid_side_rates_bps -> hand[0..1]. */
WV_DEV void se_stereo_lr_to_ms_wave(WV_LDS OaSilkEncStereo *state, WV_LDS i16 *x1, WV_LDS i16 *x2, WV_LDS i8 *ix, WV_LDS i8 *mid_only_flag, WV_LDS i32 *hand, i32 total_rate_bps,
      int prev_speech_act_Q8, int toMono, int fs_kHz, int frame_length, WV_LDS SeStereoLds *T)
{
   WV_LDS i16 *mid = &x1[-2], *side = T->side;
   wv_sync();
   FOR_LANES(n, frame_length + 2) {
      const i32 sum = x1[n - 2] + (i32)x2[n - 2], diff = x1[n - 2] - (i32)x2[n - 2];
      mid[n] = (i16)sk_rround(sum, 1); side[n] = (i16)sk_sat16(sk_rround(diff, 1));
   }
   wv_sync();
   LANE0 {
      mid[0] = state->sMid[0]; mid[1] = state->sMid[1]; side[0] = state->sSide[0]; side[1] = state->sSide[1];
      state->sMid[0] = mid[frame_length]; state->sMid[1] = mid[frame_length + 1]; state->sSide[0] = side[frame_length]; state->sSide[1] = side[frame_length + 1];
   }
   wv_sync();
   FOR_LANES(n, frame_length) {
      i32 sum = sk_rround(add32(mid[n] + (i32)mid[n + 2], shl32(mid[n + 1], 1)), 2); T->LP_mid[n] = (i16)sum; T->HP_mid[n] = (i16)(mid[n + 1] - sum);
      sum = sk_rround(add32(side[n] + (i32)side[n + 2], shl32(side[n + 1], 1)), 2); T->LP_side[n] = (i16)sum; T->HP_side[n] = (i16)(side[n + 1] - sum);
   }
   wv_sync();
   const SeStereoSums lp = se_stereo_sums_wave(T->LP_mid, T->LP_side, frame_length), hp = se_stereo_sums_wave(T->HP_mid, T->HP_side, frame_length);
   LANE0 {
      i32 pred_Q13[2], LP_ratio_Q14, HP_ratio_Q14, width_Q14, mid_side_rates_bps[2];
      const int is10msFrame = frame_length == 10 * fs_kHz;
      i32 smooth_coef_Q16 = is10msFrame ? SE_FIX(0.01 / 2, 16) : SE_FIX(0.01, 16);
      smooth_coef_Q16 = sk_mulwb(sk_mulbb(prev_speech_act_Q8, prev_speech_act_Q8), smooth_coef_Q16);
      pred_Q13[0] = se_stereo_predictor_tail(&LP_ratio_Q14, lp, &state->mid_side_amp_Q0[0], smooth_coef_Q16);
      pred_Q13[1] = se_stereo_predictor_tail(&HP_ratio_Q14, hp, &state->mid_side_amp_Q0[2], smooth_coef_Q16);
      i32 frac_Q16 = sk_mlabb(HP_ratio_Q14, LP_ratio_Q14, 3);
      frac_Q16 = imin(frac_Q16, SE_FIX(1, 16));
      total_rate_bps -= is10msFrame ? 1200 : 600;
      if (total_rate_bps < 1) total_rate_bps = 1;
      const i32 min_mid_rate_bps = sk_mlabb(2000, fs_kHz, 600), frac_3_Q16 = 3 * frac_Q16;
      mid_side_rates_bps[0] = sk_div32_varQ(total_rate_bps, SE_FIX(8 + 5, 16) + frac_3_Q16, 16 + 3);
      if (mid_side_rates_bps[0] < min_mid_rate_bps) {
         mid_side_rates_bps[0] = min_mid_rate_bps; mid_side_rates_bps[1] = total_rate_bps - mid_side_rates_bps[0];
         width_Q14 = sk_div32_varQ(shl32(mid_side_rates_bps[1], 1) - min_mid_rate_bps, sk_mulwb(SE_FIX(1, 16) + frac_3_Q16, min_mid_rate_bps), 14 + 2);
         width_Q14 = se_limit(width_Q14, 0, SE_FIX(1, 14));
      } else { mid_side_rates_bps[1] = total_rate_bps - mid_side_rates_bps[0]; width_Q14 = SE_FIX(1, 14); }
      state->smth_width_Q14 = (i16)sk_mlawb(state->smth_width_Q14, width_Q14 - state->smth_width_Q14, smooth_coef_Q16);
      *mid_only_flag = 0;
      if (toMono) { width_Q14 = 0; pred_Q13[0] = 0; pred_Q13[1] = 0; se_stereo_quant_pred(pred_Q13, ix); }
      else if (state->width_prev_Q14 == 0 && (8 * total_rate_bps < 13 * min_mid_rate_bps || sk_mulwb(frac_Q16, state->smth_width_Q14) < SE_FIX(0.05, 14))) {
         pred_Q13[0] = sk_mulbb(state->smth_width_Q14, pred_Q13[0]) >> 14; pred_Q13[1] = sk_mulbb(state->smth_width_Q14, pred_Q13[1]) >> 14;
         se_stereo_quant_pred(pred_Q13, ix);
         width_Q14 = 0; pred_Q13[0] = 0; pred_Q13[1] = 0; mid_side_rates_bps[0] = total_rate_bps; mid_side_rates_bps[1] = 0; *mid_only_flag = 1;
      } else if (state->width_prev_Q14 != 0 && (8 * total_rate_bps < 11 * min_mid_rate_bps || sk_mulwb(frac_Q16, state->smth_width_Q14) < SE_FIX(0.02, 14))) {
         pred_Q13[0] = sk_mulbb(state->smth_width_Q14, pred_Q13[0]) >> 14; pred_Q13[1] = sk_mulbb(state->smth_width_Q14, pred_Q13[1]) >> 14;
         se_stereo_quant_pred(pred_Q13, ix);
         width_Q14 = 0; pred_Q13[0] = 0; pred_Q13[1] = 0;
      } else if (state->smth_width_Q14 > SE_FIX(0.95, 14)) { se_stereo_quant_pred(pred_Q13, ix); width_Q14 = SE_FIX(1, 14); }
      else {
         pred_Q13[0] = sk_mulbb(state->smth_width_Q14, pred_Q13[0]) >> 14; pred_Q13[1] = sk_mulbb(state->smth_width_Q14, pred_Q13[1]) >> 14;
         se_stereo_quant_pred(pred_Q13, ix);
         width_Q14 = state->smth_width_Q14;
      }
      if (*mid_only_flag == 1) {
         state->silent_side_len += frame_length - 8 * fs_kHz;
         if (state->silent_side_len < 5 * fs_kHz) *mid_only_flag = 0; else state->silent_side_len = 10000;
      } else state->silent_side_len = 0;
      if (*mid_only_flag == 0 && mid_side_rates_bps[1] < 1) { mid_side_rates_bps[1] = 1; mid_side_rates_bps[0] = imax(1, total_rate_bps - mid_side_rates_bps[1]); }
      const int denom_Q16 = ((i32)1 << 16) / (8 * fs_kHz);
      hand[0] = mid_side_rates_bps[0]; hand[1] = mid_side_rates_bps[1];
      hand[2] = -state->pred_prev_Q13[0]; hand[3] = -state->pred_prev_Q13[1]; hand[4] = shl32(state->width_prev_Q14, 10);
      hand[5] = -sk_rround(sk_mulbb(pred_Q13[0] - state->pred_prev_Q13[0], denom_Q16), 16); hand[6] = -sk_rround(sk_mulbb(pred_Q13[1] - state->pred_prev_Q13[1], denom_Q16), 16);
      hand[7] = shl32(sk_mulwb(width_Q14 - state->width_prev_Q14, denom_Q16), 10);
      hand[8] = -pred_Q13[0]; hand[9] = -pred_Q13[1]; hand[10] = shl32(width_Q14, 10);
      state->pred_prev_Q13[0] = (i16)pred_Q13[0]; state->pred_prev_Q13[1] = (i16)pred_Q13[1]; state->width_prev_Q14 = (i16)width_Q14;
   }
   wv_sync();
   {
      const i32 p0s = hand[2], p1s = hand[3], ws = hand[4], d0 = hand[5], d1 = hand[6], dw = hand[7], p0e = hand[8], p1e = hand[9], we = hand[10];
      const int ramp = 8 * fs_kHz;
      FOR_LANES(n, frame_length) {
         const i32 k = n + 1;
         const i32 pred0_Q13 = n < ramp ? add32(p0s, (i32)((u32)k * (u32)d0)) : p0e, pred1_Q13 = n < ramp ? add32(p1s, (i32)((u32)k * (u32)d1)) : p1e;
         const i32 w_Q24 = n < ramp ? add32(ws, (i32)((u32)k * (u32)dw)) : we;
         i32 sum = shl32(add32(mid[n] + (i32)mid[n + 2], shl32(mid[n + 1], 1)), 9);
         sum = sk_mlawb(sk_mulwb(w_Q24, side[n + 1]), sum, pred0_Q13);
         sum = sk_mlawb(sum, shl32((i32)mid[n + 1], 11), pred1_Q13);
         x2[n - 1] = (i16)sk_sat16(sk_rround(sum, 8));
      }
   }
   wv_sync();
}

/* ---- stage taps of the emulator build (same word layout as the tapped reference shim of the tests) ---- */
#ifdef K_DUMP_ENABLED
WV_DEV void se_tap(WV_LDS OaSilkEncChannel *c, WV_LDS SeEncCtrl *ctl, int which, const WV_LDS i8 *tl_pulses = nullptr)
{
   i32 w[330]; int n = 0;
   if (which == 0) {
      w[n++] = c->speech_activity_Q8; w[n++] = c->input_tilt_Q15; for (int i = 0; i < 4; i++) w[n++] = c->input_quality_bands_Q15[i];
      w[n++] = c->SNR_dB_Q7;
      for (int i = 0; i < 4; i++) w[n++] = ctl->pitchL[i];
      w[n++] = c->indices.lagIndex; w[n++] = c->indices.contourIndex; w[n++] = c->indices.signalType; w[n++] = c->LTPCorr_Q15; w[n++] = ctl->predGain_Q16;
      K_DUMP("pitch", w, 4 * n);
   } else if (which == 1) {
      for (int i = 0; i < 4; i++) w[n++] = ctl->Gains_Q16[i];
      for (int i = 0; i < 96; i++) w[n++] = ctl->AR_Q13[i];
      for (int i = 0; i < 4; i++) w[n++] = ctl->LF_shp_Q14[i];
      for (int i = 0; i < 4; i++) w[n++] = ctl->Tilt_Q14[i];
      for (int i = 0; i < 4; i++) w[n++] = ctl->HarmShapeGain_Q14[i];
      w[n++] = c->indices.quantOffsetType; w[n++] = ctl->input_quality_Q14; w[n++] = ctl->coding_quality_Q14;
      K_DUMP("shape", w, 4 * n);
   } else if (which == 2) {
      for (int i = 0; i < 32; i++) w[n++] = ctl->PredCoef_Q12[i >> 4][i & 15];
      for (int i = 0; i < 20; i++) w[n++] = ctl->LTPCoef_Q14[i];
      w[n++] = ctl->LTP_scale_Q14;
      for (int i = 0; i < 17; i++) w[n++] = c->indices.NLSFIndices[i];
      w[n++] = c->indices.NLSFInterpCoef_Q2;
      for (int i = 0; i < 4; i++) w[n++] = c->indices.LTPIndex[i];
      w[n++] = c->indices.PERIndex;
      for (int i = 0; i < 4; i++) w[n++] = ctl->ResNrg[i];
      for (int i = 0; i < 4; i++) w[n++] = ctl->ResNrgQ[i];
      w[n++] = ctl->LTPredCodGain_Q7;
      K_DUMP("pred", w, 4 * n);
   } else if (which == 3) {
      for (int i = 0; i < 4; i++) w[n++] = ctl->Gains_Q16[i];
      for (int i = 0; i < 4; i++) w[n++] = c->indices.GainsIndices[i];
      w[n++] = ctl->Lambda_Q10; w[n++] = c->indices.quantOffsetType; w[n++] = c->LastGainIndex;
      K_DUMP("gains", w, 4 * n);
   } else {
      w[n++] = c->indices.Seed;
      for (int i = 0; i < c->frame_length; i++) w[n++] = tl_pulses[i];
      K_DUMP("nsq", w, 4 * n);
   }
}
#define SE_TAP(which) do { wv_sync(); if (wv_lane() == 0) se_tap(c, ctl, which); wv_sync(); } while (0)
#define SE_TAP_Q(which, pulses) do { wv_sync(); if (wv_lane() == 0) se_tap(c, ctl, which, pulses); wv_sync(); } while (0)
#else
#define SE_TAP(which)
#define SE_TAP_Q(which, pulses)
#endif

/* ---- silk_encode_frame_FIX.  ec / packet buffer: the caller's (L->ec, buf) in LDS; returns nBytesOut through S->r[0] ---- */
template <class PD, class PS> WV_DEV void se_copy_words_wave(PD d, PS s, int n) { wv_sync(); wv_copy_batched(d, s, n); wv_sync(); }
/* silk_encode_frame_FIX in four pieces, so that the stages between the analysis and the bookkeeping (the noise-shaping quantiser and the entropy coder: the rate-control
 * loop) can also run in a kernel of their own on a several-streams-per-wave layout (opus_sh_split.h); se_encode_frame_wave below strings them together for the one-kernel path.
 *   se_frame_head_wave      :98-128   seed, variable low-pass, the frame into x_buf
 *   se_frame_analysis_wave  :130-160  pitch, noise shaping, prediction coefficients, gains  -> ctl, c->indices
 *   se_frame_quant_wave     :162-378  LBRR, the rate-control loop: NSQ -> indices -> pulses -> bits
 *   se_frame_finish_wave    :380-389  x_buf shift, what the next frame conditions on */
template <int FRONT = 0> WV_DEV void se_frame_head_wave(WV_LDS SilkEncLds *S, WV_LDS OaSilkEncChannel *c)
{
   WV_LDS i16 *x_frame = c->x_buf + c->ltp_mem_length;
   WV_LDS i16 *inputBuf = se_inbuf<FRONT>(S, c == &se_st(S)->ch[1] ? 1 : 0);
   SE_PHASE(S, 2);
   LANE0 {
      c->indices.Seed = (i8)(c->frameCounter++ & 3);
      se_lp_variable_cutoff(c, inputBuf + 1, c->frame_length);
   }
   FOR_LANES(i, c->frame_length) x_frame[5 * c->fs_kHz + i] = inputBuf[1 + i];
   wv_sync();
}
WV_DEV void se_frame_analysis_wave(WV_LDS SilkEncLds *S, WV_LDS OaSilkEncChannel *c, int condCoding, ShPredIn *pj = nullptr /* the split path with its pred kernel: stop where that kernel takes over */, int pj_corr = 0)
{
   WV_LDS SeEncCtrl *ctl = &S->ctl;
   WV_LDS i16 *x_frame = c->x_buf + c->ltp_mem_length;
   WV_LDS SeAnaLds *A = &S->u.a;
   WV_LDS i16 *res_pitch = A->res_pitch, *res_pitch_frame = res_pitch + c->ltp_mem_length;
   se_find_pitch_lags_wave(c, ctl, res_pitch, x_frame - c->ltp_mem_length, A->u.a.Wsig, A->u.a.xx, A->w32, A->A_Q12s, &A->u.a.pitch);
   wv_sync();
   SE_TAP(0);
   SE_PHASE(S, 3);
   se_noise_shape_analysis_wave(c, ctl, res_pitch_frame, x_frame, A->u.a.Wsig, A->u.a.xx, A->w32, S->stk);
   wv_sync();
   SE_TAP(1);
   SE_PHASE(S, 4);
   se_find_pred_coefs_wave(c, ctl, res_pitch_frame, x_frame, condCoding, &A->u.p.W, A->u.p.W.LPC_in_pre, A->u.p.W.XX, A->u.p.W.LPC_res, &S->r[15], pj, pj_corr);
   SE_TAP(2);
   SE_PHASE(S, 5);
   if (pj) return;
   LANE0 se_process_gains_l0(c, ctl, condCoding);
   SE_TAP(3);
   SE_PHASE(S, 6);
}
WV_DEV void se_frame_quant_wave(WV_LDS SilkEncLds *S, WV_LDS OaSilkEncChannel *c, WV_LDS EcCtx *ecl, WV_LDS u8 *buf, int condCoding, int maxBits, int useCBR, SeRateScratch *G, OaSilkLbrr *lb)
{
   WV_LDS SeEncCtrl *ctl = &S->ctl;
   WV_LDS i16 *x_frame = c->x_buf + c->ltp_mem_length;
   const int bits_margin = useCBR ? 5 : maxBits / 4;
   const int NSQW = (int)(sizeof(OaSilkNsqState) / 4);
   WV_LDS SeQuantLds *Q = &S->u.q;
   WV_LDS OaSilkEncTail *tl = se_tail(S, c);
   se_nsq_apply_reset_wave(S, c);
   if (c->LBRR_enabled && c->speech_activity_Q8 > SE_FIX(0.3f, 8)) {
      /* silk_LBRR_encode_FIX (:392): the same frame once more with raised gains -- the noise-shaping quantiser runs on the live state, which comes
       * back from its HBM snapshot afterwards; indices and pulses go to the stream's HBM store for the next packet */
      const int fi = c->nFramesEncoded, chn = c->channelNb;
      i32 TempGains_Q16[4];
      for (int k = 0; k < 4; k++) TempGains_Q16[k] = ctl->Gains_Q16[k];
      se_copy_words_wave((i32 *)&G->nsq_copy[0], (const WV_LDS i32 *)&tl->nsq, NSQW);
      LANE0 {
         c->LBRR_flags[fi] = 1;
         { WV_LDS i32 *d = (WV_LDS i32 *)&Q->ix_lbrr; const WV_LDS i32 *sr = (const WV_LDS i32 *)&c->indices; for (int k = 0; k < (int)(sizeof(OaSilkEncIndices) / 4); k++) d[k] = sr[k]; }
         if (fi == 0 || c->LBRR_flags[fi - 1] == 0) {
            c->LBRRprevLastGainIndex = c->LastGainIndex;
            Q->ix_lbrr.GainsIndices[0] = (i8)imin(Q->ix_lbrr.GainsIndices[0] + c->LBRR_GainIncreases, 64 - 1);
         }
         i32 g[4]; i8 gi[4]; int prev = c->LBRRprevLastGainIndex;
         for (int k = 0; k < 4; k++) gi[k] = Q->ix_lbrr.GainsIndices[k];
         se_gains_dequant(g, gi, &prev, condCoding == SE_CODE_CONDITIONALLY, c->nb_subfr);
         for (int k = 0; k < c->nb_subfr; k++) ctl->Gains_Q16[k] = g[k];
         c->LBRRprevLastGainIndex = prev;
      }
      if (c->nStatesDelayedDecision > 1 || c->warping_Q16 > 0) se_nsq_del_dec_wave(c, &tl->nsq, &Q->ix_lbrr, &Q->N, ctl, x_frame, Q->pulses_lbrr);
      else se_nsq_wave(c, &tl->nsq, &Q->ix_lbrr, &Q->N, ctl, x_frame, Q->pulses_lbrr);
      wv_sync();
      FOR_LANES(i, c->frame_length) lb->pulses[chn][fi][i] = Q->pulses_lbrr[i];
      { const WV_LDS i32 *src = (const WV_LDS i32 *)&Q->ix_lbrr; i32 *dst = (i32 *)&lb->indices[chn][fi]; FOR_LANES(i, (int)(sizeof(OaSilkEncIndices) / 4)) dst[i] = src[i]; }
      se_copy_words_wave((WV_LDS i32 *)&tl->nsq, (const i32 *)&G->nsq_copy[0], NSQW);
      LANE0 { for (int k = 0; k < c->nb_subfr; k++) ctl->Gains_Q16[k] = TempGains_Q16[k]; }
   }
   const int maxIter = 6;
   int gainMult_Q8 = SE_FIX(1, 8), found_lower = 0, found_upper = 0;
   i32 gainsID = se_gains_ID(c->indices.GainsIndices, c->nb_subfr), gainsID_lower = -1, gainsID_upper = -1;
   i32 nBits = 0, nBits_lower = 0, nBits_upper = 0, gainMult_lower = 0, gainMult_upper = 0;
   int LastGainIndex_copy2 = 0;
   int gain_lock[4] = {0, 0, 0, 0}; i16 best_gain_mult[4] = {0, 0, 0, 0}; int best_sum[4] = {0, 0, 0, 0};
   ec_cp_lds(&Q->ec_copy, ecl);
   se_copy_words_wave((i32 *)&G->nsq_copy[0], (const WV_LDS i32 *)&tl->nsq, NSQW);
   const int seed_copy = c->indices.Seed, ec_prevLagIndex_copy = c->ec_prevLagIndex, ec_prevSignalType_copy = c->ec_prevSignalType;
   for (int iter = 0; ; iter++) {
      if (gainsID == gainsID_lower) nBits = nBits_lower;
      else if (gainsID == gainsID_upper) nBits = nBits_upper;
      else {
         if (iter > 0) {
            wv_sync();
            LANE0 { ec_cp_lds(ecl, &Q->ec_copy); c->indices.Seed = (i8)seed_copy; c->ec_prevLagIndex = ec_prevLagIndex_copy; c->ec_prevSignalType = ec_prevSignalType_copy; }
            se_copy_words_wave((WV_LDS i32 *)&tl->nsq, (const i32 *)&G->nsq_copy[0], NSQW);
         }
         if (c->nStatesDelayedDecision > 1 || c->warping_Q16 > 0) se_nsq_del_dec_wave(c, &tl->nsq, &c->indices, &Q->N, ctl, x_frame, tl->pulses);
         else se_nsq_wave(c, &tl->nsq, &c->indices, &Q->N, ctl, x_frame, tl->pulses);
         wv_sync();
         SE_TAP_Q(4, tl->pulses);
         SE_PHASE(S, 7);
         LANE0 {
            if (iter == maxIter && !found_lower) ec_cp_lds(&Q->ec_copy2, ecl);
            EcCtx ec_; ec_ld(&ec_, ecl); EcCtx *e = &ec_;
            se_encode_indices(c, &c->indices, EC_PASS, condCoding);
            se_encode_pulses(EC_PASS, c->indices.signalType, c->indices.quantOffsetType, tl->pulses, c->frame_length, S->stk);
            int nb = k_ec_tell(EC_PASS);
            if (iter == maxIter && !found_lower && nb > maxBits) {
               ec_ld(&ec_, &Q->ec_copy2);
               c->LastGainIndex = ctl->lastGainIndexPrev;
               for (int i = 0; i < c->nb_subfr; i++) c->indices.GainsIndices[i] = 4;
               if (condCoding != SE_CODE_CONDITIONALLY) c->indices.GainsIndices[0] = (i8)ctl->lastGainIndexPrev;
               c->ec_prevLagIndex = ec_prevLagIndex_copy; c->ec_prevSignalType = ec_prevSignalType_copy;
               for (int i = 0; i < c->frame_length; i++) tl->pulses[i] = 0;
               se_encode_indices(c, &c->indices, EC_PASS, condCoding);
               se_encode_pulses(EC_PASS, c->indices.signalType, c->indices.quantOffsetType, tl->pulses, c->frame_length, S->stk);
               nb = k_ec_tell(EC_PASS);
            }
            ec_st(ecl, &ec_);
            S->r[1] = nb;
         }
         SE_PHASE(S, 8);
         nBits = S->r[1];
         if (useCBR == 0 && iter == 0 && nBits <= maxBits) break;
      }
      if (iter == maxIter) {
         if (found_lower && (gainsID == gainsID_lower || nBits > maxBits)) {
            wv_sync();
            LANE0 { ec_cp_lds(ecl, &Q->ec_copy2); for (u32 i = 0; i < Q->ec_copy2.offs; i++) buf[i] = G->ec_buf_copy[i]; c->LastGainIndex = LastGainIndex_copy2; }
            se_copy_words_wave((WV_LDS i32 *)&tl->nsq, (const i32 *)&G->nsq_copy[1], NSQW);
         }
         break;
      }
      if (nBits > maxBits) {
         if (found_lower == 0 && iter >= 2) { LANE0 ctl->Lambda_Q10 = ctl->Lambda_Q10 + (ctl->Lambda_Q10 >> 1); found_upper = 0; gainsID_upper = -1; }
         else { found_upper = 1; nBits_upper = nBits; gainMult_upper = gainMult_Q8; gainsID_upper = gainsID; }
      } else if (nBits < maxBits - bits_margin) {
         found_lower = 1; nBits_lower = nBits; gainMult_lower = gainMult_Q8;
         if (gainsID != gainsID_lower) {
            gainsID_lower = gainsID;
            wv_sync();
            LANE0 { ec_cp_lds(&Q->ec_copy2, ecl); for (u32 i = 0; i < ecl->offs; i++) G->ec_buf_copy[i] = buf[i]; }
            se_copy_words_wave((i32 *)&G->nsq_copy[1], (const WV_LDS i32 *)&tl->nsq, NSQW);
            LastGainIndex_copy2 = c->LastGainIndex;
         }
      } else break;
      if (!found_lower && nBits > maxBits) {
         for (int i = 0; i < c->nb_subfr; i++) {
            int sum = 0;
            for (int j = i * c->subfr_length; j < (i + 1) * c->subfr_length; j++) sum += iabs((i32)tl->pulses[j]);
            if (iter == 0 || (sum < best_sum[i] && !gain_lock[i])) { best_sum[i] = sum; best_gain_mult[i] = (i16)gainMult_Q8; } else gain_lock[i] = 1;
         }
      }
      if ((found_lower & found_upper) == 0) {
         if (nBits > maxBits) gainMult_Q8 = imin(1024, gainMult_Q8 * 3 / 2); else gainMult_Q8 = imax(64, gainMult_Q8 * 4 / 5);
         gainMult_Q8 = (i16)gainMult_Q8;
      } else {
         gainMult_Q8 = gainMult_lower + ((gainMult_upper - gainMult_lower) * (maxBits - nBits_lower)) / (nBits_upper - nBits_lower);
         gainMult_Q8 = (i16)gainMult_Q8;
         if (gainMult_Q8 > gainMult_lower + ((gainMult_upper - gainMult_lower) >> 2)) gainMult_Q8 = (i16)(gainMult_lower + ((gainMult_upper - gainMult_lower) >> 2));
         else if (gainMult_Q8 < gainMult_upper - ((gainMult_upper - gainMult_lower) >> 2)) gainMult_Q8 = (i16)(gainMult_upper - ((gainMult_upper - gainMult_lower) >> 2));
      }
      wv_sync();
      LANE0 {
         for (int i = 0; i < c->nb_subfr; i++) { const i16 tmp = gain_lock[i] ? best_gain_mult[i] : (i16)gainMult_Q8; ctl->Gains_Q16[i] = sk_shl_sat(sk_mulwb(ctl->GainsUnq_Q16[i], tmp), 8); }
         c->LastGainIndex = ctl->lastGainIndexPrev;
         se_gains_quant(c->indices.GainsIndices, ctl->Gains_Q16, &c->LastGainIndex, condCoding == SE_CODE_CONDITIONALLY, c->nb_subfr);
      }
      gainsID = se_gains_ID(c->indices.GainsIndices, c->nb_subfr);
   }
}
WV_DEV void se_frame_finish_wave(WV_LDS SilkEncLds *S, WV_LDS OaSilkEncChannel *c, const WV_LDS EcCtx *ecl)
{
   const WV_LDS SeEncCtrl *ctl = &S->ctl;
   /* input buffer shift (:381): overlapping move through registers */
   {
      const int n = c->ltp_mem_length + 5 * c->fs_kHz, fl = c->frame_length;
      for (int b = 0; b < n; b += WV_WIDTH) { const int i = b + wv_lane(); i16 v = 0; if (i < n) v = c->x_buf[fl + i]; wv_sync(); if (i < n) c->x_buf[i] = v; wv_sync(); }
   }
   LANE0 {
      if (c->prefillFlag) S->r[0] = 0;
      else {
         c->prevLag = ctl->pitchL[c->nb_subfr - 1]; c->prevSignalType = c->indices.signalType; c->first_frame_after_reset = 0;
         S->r[0] = ((ecl->nbits_total - ec_ilog(ecl->rng)) + 7) >> 3;
      }
   }
}
WV_DEVN void se_encode_frame_wave(WV_LDS SilkEncLds *S, WV_LDS OaSilkEncChannel *c, WV_LDS EcCtx *ecl, WV_LDS u8 *buf, int condCoding, int maxBits, int useCBR, SeRateScratch *G, OaSilkLbrr *lb)
{
   se_frame_head_wave(S, c);
   if (!c->prefillFlag) {
      se_frame_analysis_wave(S, c, condCoding);
      se_frame_quant_wave(S, c, ecl, buf, condCoding, maxBits, useCBR, G, lb);
   }
   se_frame_finish_wave(S, c, ecl);
}

/* ---- silk_Encode.  pcm: the Opus layer's int16 staging of this call's input (interleaved, nChannelsAPI), nSamplesIn per channel.
 * Returns 0 or a negative error; *nBytesOut through S->r[0].  One SILK frame per call for 10/20 ms payloads, 2-3 frames for 40/60 ms. ---- */
struct SePcmSrc {
   const i16 *p; int stride, off, mix;                                       /* the Opus layer's high-passed input (HBM scratch) */
   WV_MEM i32 operator[](int i) const { if (!mix) return p[i * stride + off]; const i32 s = (i32)p[2 * i] + p[2 * i + 1]; return (i16)sk_rround(s, 1); }
   WV_MEM SePcmSrc operator+(int k) const { SePcmSrc r = *this; r.p = p + k * (mix ? 2 : stride); return r; }
};
/* silk_Encode in pieces (the one-kernel path strings them together in silk_encode_wave; the split path -- opus_sh_split.h -- runs the channel loop's quantiser in a kernel of its own) */
struct SeCall { int transition, nBlocksOf10ms, tot_blocks, curr_block, tmp_payloadSize_ms, tmp_complexity, nSamplesToBufferMax; };
/* enc_API.c:166-281: channel bookkeeping, the checks on the input length, the prefill reset, silk_control_encoder per channel */
WV_DEV int se_call_prologue_wave(WV_LDS SilkEncLds *S, SeControl *ec, int nSamplesIn, int prefillFlag, SeCall *k)
{
   WV_LDS OaSilkEnc *E = se_st(S);
   WV_LDS OaSilkEncChannel *c0 = &E->ch[0], *c1 = &E->ch[1];
   LANE0 {
      if (ec->reducedDependency) for (int n = 0; n < ec->nChannelsAPI; n++) E->ch[n].first_frame_after_reset = 1;
      for (int n = 0; n < ec->nChannelsAPI; n++) E->ch[n].nFramesEncoded = 0;
      ec->switchReady = 0;
      if (ec->nChannelsInternal > E->nChannelsInternal) {
         se_init_channel(c1);
         E->st.pred_prev_Q13[0] = E->st.pred_prev_Q13[1] = 0; E->st.sSide[0] = E->st.sSide[1] = 0;
         E->st.mid_side_amp_Q0[0] = 0; E->st.mid_side_amp_Q0[1] = 1; E->st.mid_side_amp_Q0[2] = 0; E->st.mid_side_amp_Q0[3] = 1;
         E->st.width_prev_Q14 = 0; E->st.smth_width_Q14 = SE_FIX(1, 14);
         if (E->nChannelsAPI == 2) { for (int i = 0; i < 9; i++) c1->rs_cfg[i] = c0->rs_cfg[i]; for (int i = 0; i < 90; i++) c1->rs_rows[i] = c0->rs_rows[i]; }
      }
   }
   k->transition = ec->payloadSize_ms != c0->PacketSize_ms || E->nChannelsInternal != ec->nChannelsInternal;
   LANE0 { E->nChannelsAPI = ec->nChannelsAPI; E->nChannelsInternal = ec->nChannelsInternal; }
   k->nBlocksOf10ms = (100 * nSamplesIn) / ec->API_sampleRate;
   k->tot_blocks = k->nBlocksOf10ms > 1 ? k->nBlocksOf10ms >> 1 : 1;
   k->curr_block = 0;
   k->tmp_payloadSize_ms = 0; k->tmp_complexity = 0;
   if (prefillFlag) {
      if (k->nBlocksOf10ms != 1) return -101;
      LANE0 {
         i32 lp[5];
         if (prefillFlag == 2) { lp[0] = c0->lp_In_LP_State[0]; lp[1] = c0->lp_In_LP_State[1]; lp[2] = c0->lp_transition_frame_no; lp[3] = c0->lp_mode; lp[4] = c0->fs_kHz; }   /* saved_fs_kHz = the rate in use */
         for (int n = 0; n < ec->nChannelsInternal; n++) {
            se_init_channel(&E->ch[n]);
            if (prefillFlag == 2) { E->ch[n].lp_In_LP_State[0] = lp[0]; E->ch[n].lp_In_LP_State[1] = lp[1]; E->ch[n].lp_transition_frame_no = lp[2]; E->ch[n].lp_mode = lp[3]; E->ch[n].lp_saved_fs_kHz = lp[4]; }
         }
      }
      k->tmp_payloadSize_ms = ec->payloadSize_ms; ec->payloadSize_ms = 10;
      k->tmp_complexity = ec->complexity; ec->complexity = 0;
      LANE0 { for (int n = 0; n < ec->nChannelsInternal; n++) { E->ch[n].controlled_since_last_payload = 0; E->ch[n].prefillFlag = 1; } }
   } else {
      if (k->nBlocksOf10ms * ec->API_sampleRate != 100 * nSamplesIn || nSamplesIn < 0) return -101;
      if (1000 * (i32)nSamplesIn > ec->payloadSize_ms * ec->API_sampleRate) return -101;
   }
   const int transition = k->transition;
   LANE0 {
      for (int n = 0; n < ec->nChannelsInternal; n++) {
         const int force_fs_kHz = n == 1 ? c0->fs_kHz : 0;
         se_control_encoder(&E->ch[n], ec, E->allowBandwidthSwitch, n, force_fs_kHz, &S->rs, S->u.rs_tmp, S->stk);
         if (E->ch[n].first_frame_after_reset || transition) for (int i = 0; i < c0->nFramesPerPacket; i++) E->ch[n].LBRR_flags[i] = 0;
         E->ch[n].inDTX = E->ch[n].useDTX;
      }
      S->r[2] = ec->maxBits; S->r[3] = ec->switchReady;
   }
   ec->maxBits = S->r[2]; ec->switchReady = S->r[3];                           /* se_control_audio_bw may have changed them on lane 0 */
   k->nSamplesToBufferMax = 10 * k->nBlocksOf10ms * c0->fs_kHz;
   return 0;
}
/* :283-340: resample this call's input to the internal rate, buffer it */
template <int FRONT = 0> WV_DEV void se_call_buffer_wave(WV_LDS SilkEncLds *S, SeControl *ec, const i16 *pcm, int nSamplesFromInput, int nSamplesToBuffer, int nBlocksOf10ms)
{
   WV_LDS OaSilkEnc *E = se_st(S);
   WV_LDS OaSilkEncChannel *c0 = &E->ch[0], *c1 = &E->ch[1];
   WV_LDS i16 *in0 = se_inbuf<FRONT>(S, 0), *in1 = se_inbuf<FRONT>(S, 1);
   for (int n = 0; n < ec->nChannelsAPI; n++) {                                  /* a channel that silk_init_encoder has started over: its input buffer with it (inbuf_reset_req) */
      if (wv_uni(E->ch[n].inbuf_reset_req)) {
         WV_LDS i32 *z = (WV_LDS i32 *)(n ? in1 : in0);
         wv_sync();
         FOR_LANES(i, SE_INBUF_WORDS) z[i] = 0;
         LANE0 E->ch[n].inbuf_reset_req = 0;
      }
   }
   wv_sync();
   const int ix0 = c0->inputBufIx;
   if (ec->nChannelsAPI == 2 && ec->nChannelsInternal == 2) {
      const int ix1 = c1->inputBufIx;
      LANE0 { if (E->nPrevChannelsInternal == 1 && c0->nFramesEncoded == 0) { for (int i = 0; i < 9; i++) c1->rs_cfg[i] = c0->rs_cfg[i]; for (int i = 0; i < 90; i++) c1->rs_rows[i] = c0->rs_rows[i]; } }
      SePcmSrc s0 = {pcm, 2, 0, 0}, s1 = {pcm, 2, 1, 0};
      /* both channels at once where the kernel's phase union has room for a second ring in front of the input buffers (the split path's front kernel: they sit at its end) */
      int both = 0;
      if (FRONT && sizeof(i32) * 2 * (36 + 480 + 4) + 2 * sizeof(int16_t) * (SE_MAX_FRAME + 2) <= SE_FRONT_U_BYTES)
         both = se_resample2_wave(c0->rs_cfg, c0->rs_rows, c1->rs_cfg, c1->rs_rows, S->u.rs_ring, S->u.rs_ring + (36 + 480 + 4), &in0[ix0 + 2], &in1[ix1 + 2], s0, s1, nSamplesFromInput);
      if (!both) {
      se_resample_wave(c0->rs_cfg, c0->rs_rows, &S->rs, S->u.rs_ring, &in0[ix0 + 2], s0, nSamplesFromInput);
      se_resample_wave(c1->rs_cfg, c1->rs_rows, &S->rs, S->u.rs_ring, &in1[ix1 + 2], s1, nSamplesFromInput);
      }
      LANE0 { c0->inputBufIx += nSamplesToBuffer; c1->inputBufIx += imin(c1->frame_length - c1->inputBufIx, 10 * nBlocksOf10ms * c1->fs_kHz); }
   } else if (ec->nChannelsAPI == 2 && ec->nChannelsInternal == 1) {
      SePcmSrc sm = {pcm, 2, 0, 1};
      se_resample_wave(c0->rs_cfg, c0->rs_rows, &S->rs, S->u.rs_ring, &in0[ix0 + 2], sm, nSamplesFromInput);
      if (E->nPrevChannelsInternal == 2 && c0->nFramesEncoded == 0) {
         const int ix1 = c1->inputBufIx;
         se_resample_wave(c1->rs_cfg, c1->rs_rows, &S->rs, S->u.rs_ring, &in1[ix1 + 2], sm, nSamplesFromInput);
         FOR_LANES(n, c0->frame_length) in0[ix0 + n + 2] = (i16)((in0[ix0 + n + 2] + in1[ix1 + n + 2]) >> 1);
      }
      LANE0 c0->inputBufIx += nSamplesToBuffer;
   } else {
      SePcmSrc s0 = {pcm, 1, 0, 0};
      se_resample_wave(c0->rs_cfg, c0->rs_rows, &S->rs, S->u.rs_ring, &in0[ix0 + 2], s0, nSamplesFromInput);
      LANE0 c0->inputBufIx += nSamplesToBuffer;
   }
   LANE0 E->allowBandwidthSwitch = 0;
}
/* :342-470, a full frame is buffered: the LBRR side stream of the previous packet at the head of a new one, variable high-pass, target rate, stereo L/R -> M/S,
 * VAD.  Leaves TargetRate_bps in S->r[4], the mid / side rates in S->r[5], S->r[6]. */
template <int FRONT = 0> WV_DEV void se_call_frame_head_wave(WV_LDS SilkEncLds *S, SeControl *ec, WV_LDS EcCtx *ecl, WV_LDS u8 *buf, OaSilkLbrr *lb, int activity, int prefillFlag)
{
   WV_LDS OaSilkEnc *E = se_st(S);
   WV_LDS OaSilkEncChannel *c0 = &E->ch[0], *c1 = &E->ch[1];
   WV_LDS i16 *in0 = se_inbuf<FRONT>(S, 0), *in1 = se_inbuf<FRONT>(S, 1);
   if (c0->nFramesEncoded == 0 && !prefillFlag) {                              /* LBRR data of the previous packet: HBM store -> LDS (all lanes) before lane 0 codes it */
      int any = 0;
      for (int n = 0; n < ec->nChannelsInternal; n++) for (int i = 0; i < 3; i++) any |= E->ch[n].LBRR_flags[i];
      if (any) { WV_LDS i32 *d = (WV_LDS i32 *)&S->u.lbrr; const i32 *g = (const i32 *)lb; wv_sync(); FOR_LANES(i, (int)(sizeof(OaSilkLbrr) / 4)) d[i] = g[i]; wv_sync(); }
   }
   LANE0 {
      EcCtx ec_; ec_ld(&ec_, ecl); EcCtx *e = &ec_;
      int curr_nBitsUsedLBRR = 0;
      if (c0->nFramesEncoded == 0 && !prefillFlag) {
         u8 iCDF[2] = {0, 0};
         iCDF[0] = (u8)(256 - (256 >> ((c0->nFramesPerPacket + 1) * ec->nChannelsInternal)));
         k_ec_enc_icdf(EC_PASS, 0, iCDF, 8);
         curr_nBitsUsedLBRR = k_ec_tell(EC_PASS);
         for (int n = 0; n < ec->nChannelsInternal; n++) {                        /* LBRR flags (enc_API.c:364-374) */
            int sym = 0;
            for (int i = 0; i < E->ch[n].nFramesPerPacket; i++) sym |= E->ch[n].LBRR_flags[i] << i;
            E->ch[n].LBRR_flag = sym > 0;
            if (sym && E->ch[n].nFramesPerPacket > 1) k_ec_enc_icdf(EC_PASS, sym - 1, &sk_lbrr_flags_icdf[E->ch[n].nFramesPerPacket == 2 ? 0 : 3], 8);
         }
         for (int i = 0; i < c0->nFramesPerPacket; i++) for (int n = 0; n < ec->nChannelsInternal; n++) if (E->ch[n].LBRR_flags[i]) {      /* indices and excitation (:376-400) */
            if (ec->nChannelsInternal == 2 && n == 0) {
               se_stereo_encode_pred(EC_PASS, &E->st.predIx[i][0][0]);
               if (c1->LBRR_flags[i] == 0) k_ec_enc_icdf(EC_PASS, E->st.mid_only_flags[i], sk_stereo_only_code_mid_icdf, 8);
            }
            const int cc = i > 0 && E->ch[n].LBRR_flags[i - 1] ? SE_CODE_CONDITIONALLY : SE_CODE_INDEPENDENTLY;
            se_encode_indices(&E->ch[n], &S->u.lbrr.indices[n][i], EC_PASS, cc);
            se_encode_pulses(EC_PASS, S->u.lbrr.indices[n][i].signalType, S->u.lbrr.indices[n][i].quantOffsetType, S->u.lbrr.pulses[n][i], E->ch[n].frame_length, S->stk);
         }
         for (int n = 0; n < ec->nChannelsInternal; n++) for (int i = 0; i < 3; i++) E->ch[n].LBRR_flags[i] = 0;
         curr_nBitsUsedLBRR = k_ec_tell(EC_PASS) - curr_nBitsUsedLBRR;
      }
      se_hp_variable_cutoff(c0);
      i32 nBits = (ec->bitRate * ec->payloadSize_ms) / 1000;
      if (!prefillFlag) {
         if (curr_nBitsUsedLBRR < 10) E->nBitsUsedLBRR = 0; else if (E->nBitsUsedLBRR < 10) E->nBitsUsedLBRR = curr_nBitsUsedLBRR; else E->nBitsUsedLBRR = (E->nBitsUsedLBRR + curr_nBitsUsedLBRR) / 2;
         nBits -= E->nBitsUsedLBRR;
      }
      nBits = nBits / c0->nFramesPerPacket;
      i32 T = ec->payloadSize_ms == 10 ? sk_mulbb(nBits, 100) : sk_mulbb(nBits, 50);
      T -= (E->nBitsExceeded * 1000) / 500;
      if (c0->nFramesEncoded > 0) { const i32 bitsBalance = k_ec_tell(EC_PASS) - E->nBitsUsedLBRR - nBits * c0->nFramesEncoded; T -= (bitsBalance * 1000) / 500; }
      T = se_limit(T, ec->bitRate, 5000);
      S->r[4] = T;
      ec_st(ecl, &ec_);
   }
   if (ec->nChannelsInternal == 2) {
      se_stereo_lr_to_ms_wave(&E->st, &in0[2], &in1[2], &E->st.predIx[c0->nFramesEncoded][0][0], &E->st.mid_only_flags[c0->nFramesEncoded], S->stk, S->r[4], c0->speech_activity_Q8,
            ec->toMono, c0->fs_kHz, c0->frame_length, &S->u.s);
   }
   LANE0 {
      EcCtx ec_; ec_ld(&ec_, ecl); EcCtx *e = &ec_;
      if (ec->nChannelsInternal == 2) {
         S->r[5] = S->stk[0]; S->r[6] = S->stk[1];
         if (E->st.mid_only_flags[c0->nFramesEncoded] == 0) {
            if (E->prev_decode_only_middle == 1) {
               c1->LastGainIndex = 0; c1->HarmShapeGain_smth_Q16 = 0; c1->Tilt_smth_Q16 = 0;
               c1->nsq_reset_req = 1;
               for (int i = 0; i < 16; i++) c1->prev_NLSFq_Q15[i] = 0;
               c1->lp_In_LP_State[0] = c1->lp_In_LP_State[1] = 0;
               c1->prevLag = 100; c1->LastGainIndex = 10; c1->prevSignalType = SE_TYPE_NO_VOICE; c1->first_frame_after_reset = 1;
            }
            se_vad_l0(c1, in1 + 1, S->u.vadX, activity);
         } else c1->VAD_flags[c0->nFramesEncoded] = 0;
         if (!prefillFlag) {
            se_stereo_encode_pred(EC_PASS, &E->st.predIx[c0->nFramesEncoded][0][0]);
            if (c1->VAD_flags[c0->nFramesEncoded] == 0) k_ec_enc_icdf(EC_PASS, E->st.mid_only_flags[c0->nFramesEncoded], sk_stereo_only_code_mid_icdf, 8);
         }
      } else {
         in0[0] = E->st.sMid[0]; in0[1] = E->st.sMid[1];
         E->st.sMid[0] = in0[c0->frame_length]; E->st.sMid[1] = in0[c0->frame_length + 1];
      }
      se_vad_l0(c0, in0 + 1, S->u.vadX, activity);
      ec_st(ecl, &ec_);
   }
}
/* :472-520, channel n of the frame: bit budget, rate-control mode, SNR target, conditional coding.  channelRate_bps <= 0: the channel is not coded */
struct SeChanParams { int maxBits, useCBR, condCoding; i32 channelRate_bps; };
WV_DEV SeChanParams se_call_channel_params(WV_LDS SilkEncLds *S, const SeControl *ec, int n, int tot_blocks, int curr_block)
{
   WV_LDS OaSilkEnc *E = se_st(S);
   WV_LDS OaSilkEncChannel *c0 = &E->ch[0];
   const i32 TargetRate_bps = S->r[4], MStargetRates_bps[2] = {S->r[5], S->r[6]};
   SeChanParams p;
   p.maxBits = ec->maxBits;
   if (tot_blocks == 2 && curr_block == 0) p.maxBits = p.maxBits * 3 / 5;
   else if (tot_blocks == 3) { if (curr_block == 0) p.maxBits = p.maxBits * 2 / 5; else if (curr_block == 1) p.maxBits = p.maxBits * 3 / 4; }
   p.useCBR = ec->useCBR && curr_block == tot_blocks - 1;
   if (ec->nChannelsInternal == 1) p.channelRate_bps = TargetRate_bps;
   else { p.channelRate_bps = MStargetRates_bps[n]; if (n == 0 && MStargetRates_bps[1] > 0) { p.useCBR = 0; p.maxBits -= ec->maxBits / (tot_blocks * 2); } }
   p.condCoding = SE_CODE_INDEPENDENTLY;
   if (p.channelRate_bps > 0) {
      LANE0 se_control_snr(&E->ch[n], p.channelRate_bps);
      if (c0->nFramesEncoded - n <= 0) p.condCoding = SE_CODE_INDEPENDENTLY;
      else if (n > 0 && E->prev_decode_only_middle) p.condCoding = SE_CODE_INDEPENDENTLY_NO_LTP_SCALING;
      else p.condCoding = SE_CODE_CONDITIONALLY;
   }
   return p;
}
/* :522-560 after the channels of a frame, lane 0.  part 1: what does not depend on the coded bytes (mid-only memory, the packet's VAD / LBRR flag bits -> S->r[7], "every
 * channel is in DTX" -> S->r[8], the bandwidth-switch timer); part 2: the flag bits patched into the first payload byte, the bit reservoir; 3 = both */
WV_DEV void se_call_frame_tail_l0(WV_LDS SilkEncLds *S, SeControl *ec, WV_LDS EcCtx *ecl, WV_LDS u8 *buf, int nBytesOut, int prefillFlag, int part)
{
   WV_LDS OaSilkEnc *E = se_st(S);
   WV_LDS OaSilkEncChannel *c0 = &E->ch[0], *c1 = &E->ch[1];
   if (part & 1) E->prev_decode_only_middle = E->st.mid_only_flags[c0->nFramesEncoded - 1];
   if (nBytesOut > 0 && c0->nFramesEncoded == c0->nFramesPerPacket) {
      if (part & 1) {
         int flags = 0;
         for (int n = 0; n < ec->nChannelsInternal; n++) {
            for (int i = 0; i < E->ch[n].nFramesPerPacket; i++) { flags <<= 1; flags |= E->ch[n].VAD_flags[i]; }
            flags <<= 1; flags |= E->ch[n].LBRR_flag;
         }
         S->r[7] = flags; S->r[8] = c0->inDTX && (ec->nChannelsInternal == 1 || c1->inDTX);
         const int thr = sk_mlawb(SE_FIX(0.05f, 8), SE_FIX((1 - 0.05f) / 5000, 16 + 8), E->timeSinceSwitchAllowed_ms);
         if (c0->speech_activity_Q8 < thr) { E->allowBandwidthSwitch = 1; E->timeSinceSwitchAllowed_ms = 0; } else { E->allowBandwidthSwitch = 0; E->timeSinceSwitchAllowed_ms += ec->payloadSize_ms; }
      }
      if (part & 2) {
         if (!prefillFlag) {
            EcCtx ec_; ec_ld(&ec_, ecl); EcCtx *e = &ec_;
            k_ec_enc_patch_initial_bits(EC_PASS, S->r[7], (c0->nFramesPerPacket + 1) * ec->nChannelsInternal);
            ec_st(ecl, &ec_);
         }
         int nb = nBytesOut;
         if (S->r[8]) nb = 0;
         E->nBitsExceeded += nb * 8;
         E->nBitsExceeded -= (ec->bitRate * ec->payloadSize_ms) / 1000;
         E->nBitsExceeded = se_limit(E->nBitsExceeded, 0, 10000);
         S->r[0] = nb;
      }
   }
}
/* :562-590 */
WV_DEV void se_call_epilogue_wave(WV_LDS SilkEncLds *S, SeControl *ec, int prefillFlag, const SeCall *k)
{
   WV_LDS OaSilkEnc *E = se_st(S);
   WV_LDS OaSilkEncChannel *c0 = &E->ch[0];
   LANE0 E->nPrevChannelsInternal = ec->nChannelsInternal;
   ec->allowBandwidthSwitch = E->allowBandwidthSwitch;
   ec->inWBmodeWithoutVariableLP = c0->fs_kHz == 16 && c0->lp_mode == 0;
   ec->internalSampleRate = sk_mulbb(c0->fs_kHz, 1000);
   ec->stereoWidth_Q14 = ec->toMono ? 0 : E->st.smth_width_Q14;
   if (prefillFlag) {
      ec->payloadSize_ms = k->tmp_payloadSize_ms; ec->complexity = k->tmp_complexity;
      LANE0 { for (int n = 0; n < ec->nChannelsInternal; n++) { E->ch[n].controlled_since_last_payload = 0; E->ch[n].prefillFlag = 0; } }
   }
   ec->signalType = c0->indices.signalType;
   ec->offset = se_quantization_offsets_q10[(c0->indices.signalType >> 1) * 2 + c0->indices.quantOffsetType];
}
WV_DEV int silk_encode_wave(WV_LDS SilkEncLds *S, SeControl *ec, const i16 *pcm, int nSamplesIn, WV_LDS EcCtx *ecl, WV_LDS u8 *buf, int activity, SeRateScratch *G, OaSilkLbrr *lb, int prefillFlag = 0)
{
   prefillFlag = wv_uni(prefillFlag);
   WV_LDS OaSilkEnc *E = se_st(S);
   WV_LDS OaSilkEncChannel *c0 = &E->ch[0];
   int nBytesOut = 0;
   SeCall k;
   { const int r = se_call_prologue_wave(S, ec, nSamplesIn, prefillFlag, &k); if (r) return r; }
   while (1) {
      int nSamplesToBuffer = imin(c0->frame_length - c0->inputBufIx, k.nSamplesToBufferMax);
      const int nSamplesFromInput = (nSamplesToBuffer * c0->API_fs_Hz) / (c0->fs_kHz * 1000);
      se_call_buffer_wave(S, ec, pcm, nSamplesFromInput, nSamplesToBuffer, k.nBlocksOf10ms);
      pcm += nSamplesFromInput * ec->nChannelsAPI;
      nSamplesIn -= nSamplesFromInput;
      if (c0->inputBufIx < c0->frame_length) break;
      /* ---- enough data: encode one frame ---- */
      se_call_frame_head_wave(S, ec, ecl, buf, lb, activity, prefillFlag);
      for (int n = 0; n < ec->nChannelsInternal; n++) {
         const SeChanParams p = se_call_channel_params(S, ec, n, k.tot_blocks, k.curr_block);
         if (p.channelRate_bps > 0) {
            se_encode_frame_wave(S, &E->ch[n], ecl, buf, p.condCoding, p.maxBits, p.useCBR, G, lb);
            nBytesOut = S->r[0];
         }
         wv_sync();
         LANE0 { E->ch[n].controlled_since_last_payload = 0; E->ch[n].inputBufIx = 0; E->ch[n].nFramesEncoded++; }
      }
      LANE0 se_call_frame_tail_l0(S, ec, ecl, buf, nBytesOut, prefillFlag, 3);
      nBytesOut = S->r[0];
      if (nSamplesIn == 0) break;
      k.curr_block++;
   }
   se_call_epilogue_wave(S, ec, prefillFlag, &k);
   S->r[0] = nBytesOut;
   return 0;
}
#endif

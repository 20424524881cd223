/* opus_sh_split.h — the SILK-capable Opus encoder as a pipeline of kernels instead of one: what a call does before, inside and after SILK's rate-control loop, and (pipeline
 * values 3 / 4) the prediction stage in front of that loop as a stage of its own.
 *
 * Why: inside one wave the noise-shaping quantiser (silk/NSQ_del_dec.c:114: a 320-step recurrence per stream whose only parallelism is its <= 4 survivors) and the
 * entropy coder (silk/encode_indices.c:35, encode_pulses.c:60: one symbol after the other) keep 4 and 1 of the 64 lanes busy for 40-50 % of a frame, and their
 * working set (rings, snapshots, 250 VGPRs) sets the occupancy of everything else.  Both are independent ACROSS streams, so they get a kernel whose wave holds 16
 * streams: one quad per stream for the quantiser (lane = survivor: the layout of silk_nsq_dd.h), one lane per stream for the entropy coder and the rate-control
 * decisions (silk/fixed/encode_frame_FIX.c:170-370), each lane coding into its own stream's buffer.
 *
 *   front kernel  (one wave per stream)   oa_sh_front_frame   opus_encode_native's decisions, analysis.c, high-pass, silk_Encode up to and including silk_process_gains_FIX for
 *                                                             every coded channel (src/opus_encoder.c:1182-2189, silk/enc_API.c:150-470, encode_frame_FIX.c:98-160) -> ShCont;
 *                                                             values 3 / 4: only up to the LPC analysis' input (find_pred_coefs_FIX.c:101) -> ShCont.p
 *   pred stage    (values 3 / 4)          oa_sh_pred_frame    silk_find_LPC_FIX, silk_process_NLSFs, silk_residual_energy_FIX, silk_process_gains_FIX per coded channel: one wave
 *                                         oa_sh_preda_tile .. per channel (3), or lane kernels for the serial chains + wave kernels for the passes over the signal (4: the
 *                                                             default of a wide launch; silk_enc_predl.h)
 *   quant kernel  (16 streams per wave)   sq_quant_tile_wave  per coded channel: silk_NSQ_del_dec, silk_encode_indices, silk_encode_pulses inside the rate loop
 *                                                             (encode_frame_FIX.c:162-378); NSQ state, gain indices, coder state and payload bytes back to HBM
 *   back kernel   (one wave per stream)   oa_sh_back_frame    the flag bits of the SILK payload, the bit reservoir (enc_API.c:522-545), then opus_encode_frame_native from
 *                                                             :2211: CELT layer of hybrid frames, redundancy, TOC, DTX, padding, the packet
 *
 * The front kernel takes the split only for calls it can prove simple from the call's own decisions: one coded frame of 10 or 20 ms with a SILK layer, no prefill, no
 * SILK bandwidth switch in progress, no in-band FEC, the delayed-decision quantiser (complexity >= 2).  Every other call goes to the one-kernel path
 * (oa_sh_encode_kernel) unchanged: the front kernel leaves a stream it turns away exactly as it found it, except for the tonality analysis of the call's input, which
 * has then run (oa_sh_encode_frame(..., analysed = 1)).  Both paths share every stage function (silk_enc_frame.h, opus_enc_sh.h) and the stream record. */
#ifndef OPUS_AMD_OPUS_SH_SPLIT_H
#define OPUS_AMD_OPUS_SH_SPLIT_H

enum { SH_CONT_SLOW = 0, SH_CONT_FAST = 1 };
/* one coded channel of the frame, as the quantiser kernel needs it: the argument list of silk_NSQ_del_dec (OaNsqFrame / OaNsqCfg, silk_frame.h) + the rate loop's inputs */
struct ShQuantCh {
   OaNsqFrame fr;
   OaNsqCfg cfg;
   i32 GainsUnq_Q16[4], lastGainIndexPrev, LastGainIndex, condCoding, maxBits, useCBR, ec_prevLagIndex, ec_prevSignalType, chan, nsq_reset;
   i32 lbrr_on, LBRR_GainIncreases, lbrr_fi, lbrr_prev_flag, LBRRprevLastGainIndex, pad_[3];      /* in-band FEC: this frame (number lbrr_fi of its packet) gets a second, coarser quantisation for the NEXT packet's side stream (silk_LBRR_encode_FIX) */
   OaSilkEncIndices indices;
   i16 x16[SE_MAX_FRAME];
};
/* the call between the kernels */
struct ShCont {
   i32 kind, nq;                                         /* SH_CONT_*; coded channels (jobs) of the frame */
   i32 silk_flags, silk_flag_bits, silk_dtx, pad0[3];    /* VAD / LBRR flag bits of the payload's first byte, their count, "every channel is in DTX" */
   EcCtx ec;                                             /* front -> quant -> back */
   SeControl sc;
   SeCall k; i32 blk_from_input, blk_to_buffer;          /* a 40 / 60 ms SILK packet: silk_Encode's loop variables between the blocks (one front -> pred -> quant relay per 20 ms frame) */
   ShShared sh;
   OaShScalars st;
   ShQuantCh q[2];
   ShPredIn p[2];                                        /* front -> pred (pipeline modes 3 / 4): the prediction stage's input, per coded channel */
   ShPredMid m[2];                                       /* between the pred stage's kernels (mode 4) */
   u8 packet[OA_MAX_PACKET + 4];
};

/* ---------------- front ---------------- */
/* the wave hands the call to the one-kernel path: nothing of the stream record has been written (the analysis aside) */
WV_DEV void sh_front_decline(ShCont *ct, int *slow_list, unsigned *slow_count, int s)
{
   wv_sync();
   if (wv_lane() == 0) { ct->kind = SH_CONT_SLOW; slow_list[atomicAdd(slow_count, 1u)] = s; }
   wv_sync();
}
template <class PD, class PS> WV_DEV void sh_copy_words(PD d, PS s, int n) { wv_copy_batched(d, s, n); }
/* One block of silk_Encode's loop (enc_API.c:283-560) in the front kernels: the block's input into the channels' buffers, the head of the frame, every coded channel up to its
 * quantiser job, the call's continuation record.  The first block of a call comes here from oa_sh_front_frame, the later ones of a 40 / 60 ms SILK packet from
 * oa_sh_front_cont_frame, after the quantiser kernel has coded the block before. */
WV_DEV void sh_front_silk_block(WV_LDS ShLds *L, OaShStream *gs, const i16 *pcm_blk, ShCont *ct, SeControl &sc, const SeCall &k, int nSamplesFromInput, int nSamplesToBuffer, int last, int pred_split, int fec_ok)
{
   WV_LDS ShShared *sh = &L->sh; WV_LDS OaShScalars *st = &L->st;
   WV_LDS SilkEncLds *S = &L->S;
   WV_LDS OaSilkEnc *E = se_st(S);
   WV_LDS OaSilkEncChannel *c0 = &E->ch[0];
   const int CC = L->cfg.channels;
   {
      /* the channels' input buffers (OaSilkEnc.inbuf) are scratch in this kernel: every sample of them that a later stage reads is written by this call (the whole frame is
       * buffered in one go, the two samples in front of it come from the stereo state), they live at the end of the phase union (se_inbuf<1>) from the resampler to the heads
       * of the frames and go back to the record before the first analysis overwrites the union.  What this call does not write (a channel the call leaves alone) is brought
       * in first, so that the record ends up as the one-kernel path leaves it */
   for (int n = 0; n < CC; n++) sh_copy_words((WV_LDS i32 *)se_inbuf<1>(S, n), (const i32 *)gs->silk.inbuf[n], SE_INBUF_WORDS);
   wv_sync();
   se_call_buffer_wave<1>(S, &sc, pcm_blk, nSamplesFromInput, nSamplesToBuffer, k.nBlocksOf10ms);
   }
   wv_sync();
   se_call_frame_head_wave<1>(S, &sc, &L->ec, SH_PKT(L) + 1, &gs->lbrr, wv_uni(sh->activity), 0);
   /* the head of every channel that is coded (seed, variable low-pass, the frame into x_buf: all of it the channel's own state) -- the last reader of the input buffers */
   for (int n = 0; n < sc.nChannelsInternal; n++) {
      const i32 rate = sc.nChannelsInternal == 1 ? wv_uni(S->r[4]) : wv_uni(S->r[5 + n]);
      if (rate > 0) se_frame_head_wave<1>(S, &E->ch[n]);
   }
   wv_sync();
   for (int n = 0; n < CC; n++) sh_copy_words((i32 *)gs->silk.inbuf[n], (const WV_LDS i32 *)se_inbuf<1>(S, n), SE_INBUF_WORDS);
   wv_sync();
   int nq = 0;
   for (int n = 0; n < sc.nChannelsInternal; n++) {
      const SeChanParams p = se_call_channel_params(S, &sc, n, k.tot_blocks, k.curr_block);
      if (p.channelRate_bps > 0) {
         WV_LDS OaSilkEncChannel *c = &E->ch[n];
         se_frame_analysis_wave(S, c, p.condCoding, pred_split ? &ct->p[nq] : (ShPredIn *)nullptr, pred_split == 2);
         wv_sync();
         {  /* the channel's job for the quantiser kernel */
            ShQuantCh *q = &ct->q[nq];
            const WV_LDS SeEncCtrl *ctl = &S->ctl;
            const WV_LDS i16 *x_frame = c->x_buf + c->ltp_mem_length;
            FOR_LANES(i, 32) q->fr.PredCoef_Q12[i] = ctl->PredCoef_Q12[i >> 4][i & 15];
            FOR_LANES(i, 20) q->fr.LTPCoef_Q14[i] = ctl->LTPCoef_Q14[i];
            FOR_LANES(i, 4 * 24) q->fr.AR_Q13[i] = ctl->AR_Q13[i];
            FOR_LANES(i, 4) {
               q->fr.HarmShapeGain_Q14[i] = ctl->HarmShapeGain_Q14[i]; q->fr.Tilt_Q14[i] = ctl->Tilt_Q14[i]; q->fr.LF_shp_Q14[i] = ctl->LF_shp_Q14[i];
               q->fr.Gains_Q16[i] = ctl->Gains_Q16[i]; q->fr.pitchL[i] = ctl->pitchL[i]; q->GainsUnq_Q16[i] = ctl->GainsUnq_Q16[i];
            }
            FOR_LANES(i, c->frame_length) q->x16[i] = x_frame[i];
            sh_copy_words((i32 *)&q->indices, (const WV_LDS i32 *)&c->indices, (int)(sizeof(OaSilkEncIndices) / 4));
            if (wv_lane() == 0) {
               q->fr.signalType = c->indices.signalType; q->fr.quantOffsetType = c->indices.quantOffsetType; q->fr.NLSFInterpCoef_Q2 = c->indices.NLSFInterpCoef_Q2; q->fr.Seed = c->indices.Seed;
               q->fr.Lambda_Q10 = ctl->Lambda_Q10; q->fr.LTP_scale_Q14 = ctl->LTP_scale_Q14;
               q->cfg.fs_kHz = c->fs_kHz; q->cfg.nb_subfr = c->nb_subfr; q->cfg.predictLPCOrder = c->predictLPCOrder; q->cfg.shapingLPCOrder = c->shapingLPCOrder;
               q->cfg.nStatesDelayedDecision = c->nStatesDelayedDecision; q->cfg.warping_Q16 = c->warping_Q16;
               q->lastGainIndexPrev = ctl->lastGainIndexPrev; q->LastGainIndex = c->LastGainIndex; q->condCoding = p.condCoding; q->maxBits = p.maxBits; q->useCBR = p.useCBR;
               q->ec_prevLagIndex = c->ec_prevLagIndex; q->ec_prevSignalType = c->ec_prevSignalType; q->chan = n;
               q->nsq_reset = c->nsq_reset_req; c->nsq_reset_req = 0;                          /* the quantiser kernel starts this channel's state over (se_nsq_apply_reset_wave on the one-kernel path) */
               q->lbrr_on = c->LBRR_enabled && c->speech_activity_Q8 > SE_FIX(0.3f, 8); q->LBRR_GainIncreases = c->LBRR_GainIncreases;      /* (encode_frame_FIX.c:392-410) */
               q->lbrr_fi = c->nFramesEncoded; q->lbrr_prev_flag = c->nFramesEncoded > 0 ? c->LBRR_flags[c->nFramesEncoded - 1] : 0; q->LBRRprevLastGainIndex = c->LBRRprevLastGainIndex;
            }
         }
         wv_sync();
         se_frame_finish_wave(S, c, &L->ec);
         nq++;
      }
      wv_sync();
      LANE0 { E->ch[n].controlled_since_last_payload = 0; E->ch[n].inputBufIx = 0; E->ch[n].nFramesEncoded++; }
   }
   LANE0 se_call_frame_tail_l0(S, &sc, &L->ec, SH_PKT(L) + 1, 1, 0, 1);
   if (last) se_call_epilogue_wave(S, &sc, 0, &k);
   /* ---- the call so far -> HBM: the continuation record, the SILK state ---- */
   wv_sync();
   sh_copy_words((i32 *)&ct->sh, (const WV_LDS i32 *)sh, (int)(sizeof(ShShared) / 4));
   sh_copy_words((i32 *)&ct->st, (const WV_LDS i32 *)st, (int)(sizeof(OaShScalars) / 4));
   sh_copy_words((i32 *)&ct->ec, (const WV_LDS i32 *)&L->ec, (int)(sizeof(EcCtx) / 4));
   sh_copy_words((i32 *)ct->packet, (const WV_LDS i32 *)SH_PKT(L), fec_ok ? (int)((wv_uni((i32)L->ec.offs) + 11) / 4) : SH_FRONT_PKT_BYTES / 4);        /* (the header symbols: a handful of bytes at most -- with in-band FEC: + the side stream; the quantiser kernel's lanes code on from there, in HBM) */
   se_state_copy_wave((i32 *)&gs->silk, (const WV_LDS i32 *)se_st(S), CC, 0);
   if (wv_lane() == 0) {
      ct->sc = sc; ct->kind = SH_CONT_FAST; ct->nq = nq; ct->k = k; ct->k.curr_block = k.curr_block + 1; ct->blk_from_input = nSamplesFromInput; ct->blk_to_buffer = nSamplesToBuffer;
      ct->silk_flags = S->r[7]; ct->silk_dtx = S->r[8]; ct->silk_flag_bits = (c0->nFramesPerPacket + 1) * sc.nChannelsInternal;
   }
   wv_sync();
}


WV_DEVN void oa_sh_front_frame(WV_LDS ShLds *L, OaShStream *gs, const i16 *pcm, int frame_size, int max_data_bytes, i16 *pcm_hp, CeltScratch *cs, ShCont *ct, const i32 *apcm,
      int *slow_list, unsigned *slow_count, int s, int analysis_frame_size = 0, int pred_split = 0 /* 1: the prediction stage is the pred kernel's (oa_sh_pred_frame); 2: the pred lane / wave kernels' (mode 4), which want the Burg correlations too */,
      int pkt_window = SH_FRONT_PKT_BYTES /* bytes of packet buffer this launch gave the wave: SH_FRONT_PKT_BYTES (a few header symbols), or SH_PKT_BYTES -- a launch with in-band FEC in its batch: the previous packet's LBRR side stream is coded at the head of this one (enc_API.c:364-404) */)
{
   const int fec_ok = pkt_window >= (int)SH_PKT_BYTES;
   WV_LDS ShShared *sh = &L->sh; WV_LDS OaShScalars *st = &L->st;
   WV_LDS SilkEncLds *S = &L->S;
   LANE0 { L->silk_tail = 0; S->st_off = (i32)SE_FRONT_ST_OFF; }          /* the quantiser tails stay in HBM: they are the quantiser kernel's; the state sits behind the analysis working set */
   wv_sync();
   WV_LDS OaSilkEnc *E = se_st(S);
   sh_call_open_wave(L, gs, pcm, frame_size, max_data_bytes, cs, apcm, 0, analysis_frame_size);
   const int CC = L->cfg.channels, Fs = L->cfg.Fs;
   const int celt_only = wv_uni(st->mode) == OA_MODE_CELT_ONLY;
   /* a SILK-only call of 40 / 60 ms is ONE Opus frame whose SILK layer codes two / three 20 ms frames on one coder (nFramesPerPacket, enc_API.c:283-560): the pipeline runs its
    * front -> pred -> quantiser relay once per frame (oa_sh_front_cont_frame takes the later ones) -- with the full packet window, which later frames code into */
   const int silk_multi = fec_ok && wv_uni(st->mode) == OA_MODE_SILK_ONLY && (frame_size * 25 == Fs || frame_size * 50 == 3 * Fs);
   {
      /* a CELT-only frame (the call's decision: opus_encoder.c:1413-1470) has no SILK layer at all: it skips the quantiser stage and is coded whole by the back kernel, at that
       * kernel's occupancy instead of the one-kernel path's */
      const int simple = !wv_uni(sh->err) && !wv_uni(sh->plc_frame) && wv_uni(sh->nb_frames) == 1 && !wv_uni(sh->prefill) && !wv_uni(st->silk_bw_switch) &&
                         (fec_ok || !wv_uni(L->cfg.use_inband_fec)) && wv_uni(L->cfg.complexity) >= 2 && (celt_only || frame_size * 100 == Fs || frame_size * 50 == Fs || silk_multi);
      if (!simple) { sh_front_decline(ct, slow_list, slow_count, s); return; }
   }
   LANE0 { sh->f_redundancy = sh->redundancy; sh->f_celt_to_silk = sh->celt_to_silk; sh->f_prefill = sh->prefill; sh->f_to_celt = sh->to_celt; sh->f_silence = sh->is_silence; st->nonfinal_frame = 0; }
   SeControl sc;
   if (celt_only) { i32 *z = (i32 *)&sc; for (int i = 0; i < (int)(sizeof(SeControl) / 4); i++) z[i] = 0; }         /* (sh_frame_front_wave leaves it alone for this mode; the record carries it) */
   sh_frame_front_wave(L, gs, pcm, frame_size, wv_uni(sh->max_data_bytes), pcm_hp, &sc);
   if (celt_only) {
      /* ---- the call so far -> HBM: the continuation record (no job for the quantiser kernel; the SILK state is as it was) ---- */
      wv_sync();
      sh_copy_words((i32 *)&ct->sh, (const WV_LDS i32 *)sh, (int)(sizeof(ShShared) / 4));
      sh_copy_words((i32 *)&ct->st, (const WV_LDS i32 *)st, (int)(sizeof(OaShScalars) / 4));
      sh_copy_words((i32 *)&ct->ec, (const WV_LDS i32 *)&L->ec, (int)(sizeof(EcCtx) / 4));
      sh_copy_words((i32 *)ct->packet, (const WV_LDS i32 *)SH_PKT(L), SH_FRONT_PKT_BYTES / 4);
      if (wv_lane() == 0) { ct->sc = sc; ct->kind = SH_CONT_FAST; ct->nq = 0; ct->silk_flags = 0; ct->silk_dtx = 0; ct->silk_flag_bits = 0; ct->k.tot_blocks = 1; ct->k.curr_block = 1; }
      wv_sync();
      return;
   }
   /* ---- silk_Encode, one frame (silk_encode_wave's pieces in its order; the quantiser and the coder of each channel are what is left out) ---- */
   SeCall k;
   if (se_call_prologue_wave(S, &sc, frame_size, 0, &k)) { sh_front_decline(ct, slow_list, slow_count, s); return; }
   WV_LDS OaSilkEncChannel *c0 = &E->ch[0];
   {
      const int nblk = silk_multi ? frame_size * 50 / Fs : 1;
      int ok = k.tot_blocks == nblk && wv_uni(c0->inputBufIx) == 0 && wv_uni(c0->nFramesPerPacket) == nblk;
      for (int n = 0; n < sc.nChannelsInternal; n++) ok = ok && (fec_ok || !wv_uni(E->ch[n].LBRR_enabled)) && (wv_uni(E->ch[n].nStatesDelayedDecision) > 1 || wv_uni(E->ch[n].warping_Q16) > 0);
      /* the previous packet's LBRR side stream is still owed (enc_API.c:364-404 codes it at the head of this packet whatever the FEC setting is now -- the first frame after
       * OPUS_SET_INBAND_FEC goes 1 -> 0): indices and pulses of up to three frames do not fit the front kernel's SH_FRONT_PKT_BYTES window, the one-kernel path codes this call */
      if (!fec_ok) for (int n = 0; n < sc.nChannelsInternal; n++) for (int i = 0; i < 3; i++) ok = ok && !wv_uni(E->ch[n].LBRR_flags[i]);
      const int nSamplesToBuffer = imin(wv_uni(c0->frame_length) - wv_uni(c0->inputBufIx), k.nSamplesToBufferMax);
      const int nSamplesFromInput = (nSamplesToBuffer * wv_uni(c0->API_fs_Hz)) / (wv_uni(c0->fs_kHz) * 1000);
      ok = ok && nSamplesFromInput * nblk == frame_size && nSamplesToBuffer == wv_uni(c0->frame_length);
      if (sc.nChannelsInternal == 2) ok = ok && wv_uni(E->ch[1].inputBufIx) == 0;
      const int last = k.tot_blocks == 1;                                          /* (a 40 / 60 ms SILK packet: the first of its frames; the call's input starts at pcm_hp) */
      if (!ok) { sh_front_decline(ct, slow_list, slow_count, s); return; }
      sh_front_silk_block(L, gs, pcm_hp, ct, sc, k, nSamplesFromInput, nSamplesToBuffer, last, pred_split, fec_ok);
   }
}

/* the later 20 ms frames of a 40 / 60 ms SILK packet: the wave takes the call up again from the continuation record (Opus-layer scalars, coder, loop variables) and the stream
 * record (the SILK state as the front and quantiser kernels of the frame before left it), and runs the next block.  The packet bytes written so far come into the LDS window:
 * the frame's header symbols continue the coder where the quantiser kernel left it. */
WV_DEVN void oa_sh_front_cont_frame(WV_LDS ShLds *L, OaShStream *gs, int frame_size, const i16 *pcm_hp, ShCont *ct, int block, int pred_split)
{
   WV_LDS ShShared *sh = &L->sh; WV_LDS OaShScalars *st = &L->st;
   WV_LDS SilkEncLds *S = &L->S;
   LANE0 { L->silk_tail = 0; S->st_off = (i32)SE_FRONT_ST_OFF; }
   wv_sync();
   sh_copy_words((WV_LDS i32 *)&L->cfg, (const i32 *)&gs->cfg, (int)(sizeof(OaShConfig) / 4));
   sh_copy_words((WV_LDS i32 *)sh, (const i32 *)&ct->sh, (int)(sizeof(ShShared) / 4));
   sh_copy_words((WV_LDS i32 *)st, (const i32 *)&ct->st, (int)(sizeof(OaShScalars) / 4));
   sh_copy_words((WV_LDS i32 *)&L->ec, (const i32 *)&ct->ec, (int)(sizeof(EcCtx) / 4));
   wv_sync();
   const int CC = L->cfg.channels;
   se_state_copy_wave((WV_LDS i32 *)se_st(S), (const i32 *)&gs->silk, CC, 0);
   sh_copy_words((WV_LDS i32 *)SH_PKT(L), (const i32 *)ct->packet, (int)((wv_uni((i32)L->ec.offs) + 8) / 4));
   wv_sync();
   SeControl sc = ct->sc;
   SeCall k = ct->k;
   const int nSamplesFromInput = wv_uni(ct->blk_from_input), nSamplesToBuffer = wv_uni(ct->blk_to_buffer);
   (void)frame_size;
   sh_front_silk_block(L, gs, pcm_hp + (size_t)block * nSamplesFromInput * sc.nChannelsAPI, ct, sc, k, nSamplesFromInput, nSamplesToBuffer, k.curr_block == k.tot_blocks - 1, pred_split, 1);
}

/* ---------------- pred (pipeline mode 3) ---------------- */
/* The prediction stage of a coded channel -- silk_find_LPC_FIX (two Burg recursions, the NLSF interpolation search), silk_process_NLSFs (the 16-survivor trellis quantiser,
 * NLSF -> LPC), silk_residual_energy_FIX, silk_process_gains_FIX: find_pred_coefs_FIX.c:115-144, encode_frame_FIX.c:157 -- as a kernel of its own between the front kernel
 * and the quantiser.  The stage is 43 % of the front kernel's time (profiles/r04_s), a chain of short dependent steps: the wave is latency-bound, so what it needs is
 * company on its SIMD -- and this stage alone fits 64 VGPRs and 3.5 KB of LDS: 32 waves per CU instead of the front kernel's 16 (mono) / 12 (stereo).  Same stage functions,
 * templated on the channel type (SePredChan: the dozen fields of OaSilkEncChannel they read); inputs from the call's continuation record (ShPredIn), results into the
 * quantiser's job (ShQuantCh) and the one piece of stream state the stage owns (prev_NLSFq_Q15). */
struct SePredChan {
   i32 predictLPCOrder, nb_subfr, subfr_length, useInterpolatedNLSFs, first_frame_after_reset, speech_activity_Q8, NLSF_MSVQ_Survivors, SNR_dB_Q7, input_tilt_Q15, nStatesDelayedDecision, LastGainIndex, pad_;
   i16 prev_NLSFq_Q15[16];
   OaSilkEncIndices indices;
};
struct PredLds { SePredChan c; SeEncCtrl ctl; SeLpcWork W; i32 tk[4]; };
WV_DEVN void oa_sh_pred_frame(WV_LDS PredLds *P, OaShStream *gs, ShCont *ct, int j /* the coded channel (job) of the frame: one work item each */,
      int tail /* mode 4: the LPC analysis and the NLSF quantiser have run in the lane kernels -- residual energies and gains only */)
{
   if (wv_uni(ct->kind) != SH_CONT_FAST || j >= wv_uni(ct->nq)) return;
   {
      const ShPredIn *in = &ct->p[j]; ShQuantCh *q = &ct->q[j];
      WV_LDS SePredChan *c = &P->c; WV_LDS SeEncCtrl *ctl = &P->ctl; WV_LDS SeLpcWork *W = &P->W;
      wv_sync();
      const int order = wv_uni(in->predictLPCOrder), nb = wv_uni(in->nb_subfr), sl = wv_uni(in->subfr_length);
      FOR_LANES(i, nb * (sl + order)) W->LPC_in_pre[i] = in->LPC_in_pre[i];
      FOR_LANES(i, 16) c->prev_NLSFq_Q15[i] = in->prev_NLSFq_Q15[i];
      FOR_LANES(i, 4) { W->local_gains[i] = in->local_gains[i]; ctl->Gains_Q16[i] = q->fr.Gains_Q16[i]; }
      sh_copy_words((WV_LDS i32 *)&c->indices, (const i32 *)&q->indices, (int)(sizeof(OaSilkEncIndices) / 4));
      if (wv_lane() == 0) {
         c->predictLPCOrder = order; c->nb_subfr = nb; c->subfr_length = sl; c->useInterpolatedNLSFs = in->useInterpolatedNLSFs; c->first_frame_after_reset = in->first_frame_after_reset;
         c->speech_activity_Q8 = in->speech_activity_Q8; c->NLSF_MSVQ_Survivors = in->NLSF_MSVQ_Survivors; c->SNR_dB_Q7 = in->SNR_dB_Q7; c->input_tilt_Q15 = in->input_tilt_Q15;
         c->nStatesDelayedDecision = in->nStatesDelayedDecision; c->LastGainIndex = q->LastGainIndex;
         ctl->LTPredCodGain_Q7 = in->LTPredCodGain_Q7; ctl->coding_quality_Q14 = in->coding_quality_Q14; ctl->input_quality_Q14 = in->input_quality_Q14;
      }
      wv_sync();
      if (!tail) {
         se_find_lpc_wave(c, W, W->LPC_in_pre, wv_uni(in->minInvGain_Q30), W->LPC_res, P->tk);
         se_process_nlsfs_wave(c, &ctl->PredCoef_Q12[0][0], W);
         wv_sync();
         FOR_LANES(i, nb * (sl + order)) W->LPC_in_pre[i] = in->LPC_in_pre[i];          /* (the quantiser has worked in its bytes) */
      } else { FOR_LANES(i, 32) ctl->PredCoef_Q12[i >> 4][i & 15] = q->fr.PredCoef_Q12[i]; }
      wv_sync();
      se_residual_energy_wave(ctl->ResNrg, ctl->ResNrgQ, W->LPC_in_pre, &ctl->PredCoef_Q12[0][0], W->local_gains, sl, nb, order, W->LPC_res);
      LANE0 se_process_gains_l0(c, ctl, wv_uni(q->condCoding));
      wv_sync();
      /* results: the quantiser's job, the stream's NLSF memory */
      FOR_LANES(i, 32) q->fr.PredCoef_Q12[i] = ctl->PredCoef_Q12[i >> 4][i & 15];
      FOR_LANES(i, 4) { q->fr.Gains_Q16[i] = ctl->Gains_Q16[i]; q->GainsUnq_Q16[i] = ctl->GainsUnq_Q16[i]; }
      sh_copy_words((i32 *)&q->indices, (const WV_LDS i32 *)&c->indices, (int)(sizeof(OaSilkEncIndices) / 4));
      if (!tail) { i16 *pn = gs->silk.ch[wv_uni(q->chan)].prev_NLSFq_Q15; FOR_LANES(i, 16) pn[i] = i < order ? W->NLSF_Q15[i] : (i16)0; }
      if (wv_lane() == 0) {
         q->fr.quantOffsetType = c->indices.quantOffsetType; q->fr.NLSFInterpCoef_Q2 = c->indices.NLSFInterpCoef_Q2; q->fr.Lambda_Q10 = ctl->Lambda_Q10;
         q->lastGainIndexPrev = ctl->lastGainIndexPrev; q->LastGainIndex = c->LastGainIndex;
         /* what silk_Encode reports of the first channel's frame to the Opus layer (enc_API.c:557-562; the hybrid CELT layer reads it as SILKInfo): the quantisation offset
          * follows the quantOffsetType silk_process_gains_FIX has just decided -- the front kernel's epilogue wrote it from the value before */
         if (q->chan == 0) ct->sc.offset = se_quantization_offsets_q10[(c->indices.signalType >> 1) * 2 + c->indices.quantOffsetType];
      }
      wv_sync();
   }
}

/* ---------------- pred, pipeline mode 4: lane kernels for the serial parts, wave kernels for the passes over the signal (silk_enc_predl.h) ----------------
 * A work item = a coded channel (item = stream * 2 + job) of the pred work list; the lane kernels take PL_STREAMS of them per wave. */
WV_DEVN void oa_sh_preda_tile(WV_LDS i32 *F, ShCont *conts, const int *list, int base, int cnt)
{
   const int lane = wv_lane();
   if (lane < cnt) { const int it = list[base + lane]; ShCont *ct = &conts[it >> 1]; pl_stage_a(F + lane * PL_A_WORDS, &ct->p[it & 1], &ct->m[it & 1]); }
}
/* the interpolation choice (find_LPC_FIX.c:88-138): one wave per coded channel on the first half frame */
WV_DEVN void oa_sh_predc_frame(WV_LDS PredLds *P, ShCont *ct, int j)
{
   const ShPredIn *in = &ct->p[j]; ShPredMid *md = &ct->m[j];
   WV_LDS SeLpcWork *W = &P->W;
   const int order = wv_uni(in->predictLPCOrder), subfr_length = wv_uni(in->subfr_length) + order;
   int coef = 4;
   wv_sync();
   if (wv_uni(md->interp)) {
      WV_LDS i16 *cand_a = (WV_LDS i16 *)W->a_Q16;                             /* 4 x 16 coefficients in a_Q16 | a_tmp_Q16 */
      FOR_LANES(i, subfr_length) ((WV_LDS i32 *)W->LPC_in_pre)[i] = ((const i32 *)in->LPC_in_pre)[i];      /* 2 * subfr_length samples */
      FOR_LANES(i, 32) ((WV_LDS i32 *)cand_a)[i] = ((const i32 *)md->cand_a)[i];
      wv_sync();
      i32 res_nrg = wv_uni(md->res_nrg); int res_nrg_Q = wv_uni(md->res_nrg_Q);
      coef = se_interp_search_wave(W->LPC_in_pre, cand_a, W->LPC_res, subfr_length, order, &res_nrg, &res_nrg_Q);
   }
   FOR_LANES(i, 16) md->NLSF_Q15[i] = coef == 4 ? md->NLSF_full[i] : md->NLSF_half[i];
   if (wv_lane() == 0) md->coef = coef;
   wv_sync();
}
WV_DEVN void oa_sh_predb_tile(WV_LDS PlBLane *B, WV_LDS SeNlsfTabs *T /* [2]: order 16, order 10 */, OaShStream *streams, ShCont *conts, const int *list, int base, int cnt)
{
   const int lane = wv_lane();
   wv_sync();
   FOR_LANES(i, 40) se_nlsf_out_tabs(&T[i >= 20], i >= 20 ? i - 20 : i, i >= 20 ? SK_NLSF_NB_MB_QSTEP_Q16 : SK_NLSF_WB_QSTEP_Q16);
   wv_sync();
   if (lane < cnt) {
      const int it = list[base + lane];
      ShCont *ct = &conts[it >> 1]; const ShPredIn *in = &ct->p[it & 1]; const ShPredMid *md = &ct->m[it & 1]; ShQuantCh *q = &ct->q[it & 1];
      WV_LDS PlBLane *c = &B[lane];
      const int order = in->predictLPCOrder, ic = md->coef;
      for (int i = 0; i < 16; i++) { c->NLSF_Q15[i] = md->NLSF_Q15[i]; c->prev[i] = in->prev_NLSFq_Q15[i]; }
      for (int i = 0; i < 17; i++) c->ind[i] = q->indices.NLSFIndices[i];
      pl_stage_b(c, in, ic, q->indices.signalType, &T[order == 16 ? 0 : 1]);
      for (int i = 0; i < 32; i++) q->fr.PredCoef_Q12[i] = c->PredCoef_Q12[i >> 4][i & 15];
      for (int i = 0; i < 17; i++) q->indices.NLSFIndices[i] = c->ind[i];
      q->indices.NLSFInterpCoef_Q2 = (i8)ic;
      { i16 *pn = streams[it >> 1].silk.ch[q->chan].prev_NLSFq_Q15; for (int i = 0; i < 16; i++) pn[i] = i < order ? c->NLSF_Q15[i] : (i16)0; }
   }
   wv_sync();
}

/* ---------------- the CELT layer's transient analysis as a lane pre-pass (celt_enc_front.h: ct_transient_lane), ahead of the back kernel ----------------
 * The back kernel hands celt_encode_with_ec [delay line | this frame's high-passed input] after the hybrid gain fade and the stereo width fade (opus_encoder.c:2304-2349); a lane
 * regenerates its channel of that signal from the stream record and the front kernel's output, with the gains it works out the way sh_frame_back_wave will -- and records them:
 * sh_celt_run uses the lane's value only when the gains that were applied are those (tr [stream][12]: 0 / 1 the channels' values, 2 the frame length they are good for or 0,
 * 3 .. 8 do_gain_fade, hb_g1, hb_g2, do_stereo_fade, fade_g1, fade_g2). */
struct CtSrcSh {
   const i16 *delay, *hp; int CC, c, total_buffer, overlap, inc, do_gain, do_stereo; i32 hb_g1, hb_g2, sg1, sg2;
   WV_MEM void rewind() {}
   WV_MEM i32 gain_at(int n, i32 g1, i32 g2) const
   {
      if (n >= overlap) return g2;
      i16 w = ct_window[n * inc]; w = (i16)mult16_16_q15(w, w);
      return (i16)(mac16_16(mult16_16(w, g2), Q15ONE - w, g1) >> 15);
   }
   WV_MEM void block(int k0, i32 *v)
   {
      i32 a[8], b[8];
#pragma unroll
      for (int j = 0; j < 8; j++) {
         const int n = k0 + j;
         const i16 *p = n < total_buffer ? delay + n * CC : hp + (n - total_buffer) * CC;
         a[j] = p[c]; b[j] = do_stereo ? p[c ^ 1] : 0;
      }
#pragma unroll
      for (int j = 0; j < 8; j++) {
         const int n = k0 + j;
         if (do_gain) { const i32 g = gain_at(n, hb_g1, hb_g2); a[j] = (i16)mult16_16_q15(g, a[j]); b[j] = (i16)mult16_16_q15(g, b[j]); }
         if (do_stereo) {
            const i32 g = gain_at(n, sg1, sg2);
            i32 diff = half32((c == 0 ? a[j] : b[j]) - (c == 0 ? b[j] : a[j]));
            diff = mult16_16_q15(g, diff);
            a[j] = c == 0 ? (i16)(a[j] - diff) : (i16)(a[j] + diff);
         }
         v[j] = a[j];
      }
   }
};
WV_DEVN void oa_sh_transient_tile(const OaShStream *streams, const ShCont *conts, const char *pcm_hp_all, int N, int CC, int n_items, int base, i16 *scr, i32 *tr)
{
   const int lane = wv_lane(), it = base + lane, itc = it < n_items ? it : n_items - 1;
   const int s = CC == 2 ? itc >> 1 : itc, c = CC == 2 ? itc & 1 : 0;
   const OaShStream *gs = streams + s; const ShCont *ct = conts + s;
   const int Fs = gs->cfg.Fs, application = gs->cfg.application, mode = ct->st.mode;
   const int total_buffer = application == OA_APP_RESTRICTED_SILK ? 0 : Fs / 250, encoder_buffer = Fs / 100;
   /* the gains, as sh_frame_back_wave will work them out (opus_encoder.c:2264-2349) */
   const i32 HB_gain = mode != OA_MODE_CELT_ONLY ? ct->sh.HB_gain : Q15ONE;
   i32 sw = mode != OA_MODE_CELT_ONLY ? ct->sc.stereoWidth_Q14 : ct->st.sm_stereoWidth_Q14;
   if (mode != OA_MODE_HYBRID || ct->st.stream_channels == 1) {
      const i32 er = ct->sh.equiv_rate;
      sw = er > 32000 ? 16384 : er < 16000 ? 0 : 16384 - 2048 * (i32)(32000 - er) / (er - 14000);
   }
   const int do_gain = (ct->st.prev_HB_gain < Q15ONE || HB_gain < Q15ONE) && application != OA_APP_RESTRICTED_SILK;
   int do_stereo = 0; i32 fg1 = 0, fg2 = 0;
   if (!gs->cfg.energy_mask_on && CC == 2 && (ct->st.hybrid_stereo_width_Q14 < (1 << 14) || sw < (1 << 14))) {
      const i16 g1 = (i16)ct->st.hybrid_stereo_width_Q14, g2 = (i16)sw;
      do_stereo = application != OA_APP_RESTRICTED_SILK; fg1 = g1 == 16384 ? Q15ONE : shl16(g1, 1); fg2 = g2 == 16384 ? Q15ONE : shl16(g2, 1);
   }
   CtTrGen<CtSrcSh> g;
   g.hist = gs->celt.prefilter_mem + c * OA_MAX_PERIOD; g.pre_mem0 = gs->celt.s.preemph_memE[c];
   g.src.delay = gs->delay_buffer + (encoder_buffer - total_buffer) * CC; g.src.hp = (const i16 *)(pcm_hp_all + (size_t)s * SH_PCM_BYTES(N, CC));
   g.src.CC = CC; g.src.c = c; g.src.total_buffer = total_buffer; g.src.inc = 48000 / Fs; g.src.overlap = OA_OVERLAP / g.src.inc;
   g.src.do_gain = do_gain; g.src.do_stereo = do_stereo; g.src.hb_g1 = (i16)ct->st.prev_HB_gain; g.src.hb_g2 = (i16)HB_gain; g.src.sg1 = (i16)(Q15ONE - fg1); g.src.sg2 = (i16)(Q15ONE - fg2);
   const i32 u = ct_transient_lane(g, N, CC, scr + lane);
   if (it < n_items) {
      i32 *r = tr + 12 * (size_t)s;
      r[c] = u;
      if (c == 0) {
         r[2] = ct->kind == SH_CONT_FAST && mode != OA_MODE_SILK_ONLY && Fs == 48000 ? N + OA_OVERLAP : 0;
         r[3] = do_gain; r[4] = ct->st.prev_HB_gain; r[5] = HB_gain; r[6] = do_stereo; r[7] = fg1; r[8] = fg2;
      }
   }
}

/* ---------------- back ---------------- */
/* The CELT layer's PVQ as a stage of its own (celt_enc_pvq4.h: four streams per wave): the back kernel stops a frame before the PVQ of its CELT pass (a hybrid or CELT-only frame
 * with that one pass), parks the wave's LDS -- this header in the stream's ShBackHdr, the CELT arena in its CeltCont -- and returns 1; oa_celt_pvq_kernel codes the bands;
 * oa_sh_back2_frame reloads the LDS and finishes the call. */
#define SH_BACK_HDR_WORDS ((int)((offsetof(ShLds, S) + 16 + 3) / 4))
struct ShBackHdr { i32 w[SH_BACK_HDR_WORDS]; };
WV_DEV void oa_sh_back_finish(WV_LDS ShLds *L, OaShStream *gs, int ret, u8 *out, int out_cap, i32 *len_out, u32 *rng_out)
{
   WV_LDS ShShared *sh = &L->sh; WV_LDS OaShScalars *st = &L->st;
   const int pad_to = (!L->cfg.use_vbr && ret > 0 && !wv_uni(sh->r[3])) ? wv_uni(sh->max_data_bytes) : 0;
   const int result = ret < 0 ? ret : sh_emit_packet(SH_PKT(L), out, ret, pad_to, out_cap);
   LANE0 { *len_out = result; *rng_out = result < 0 ? 0 : st->rangeFinal; }
   sh_copy_words((i32 *)&gs->s, (const WV_LDS i32 *)st, (int)(sizeof(OaShScalars) / 4));
   wv_sync();
}
WV_DEVN void oa_sh_back2_frame(WV_LDS ShLds *L, OaShStream *gs, int frame_size, u8 *out, int out_cap, CeltScratch *cs, const ShBackHdr *hdr, const CeltCont *cc, int pkt_off, i32 *len_out, u32 *rng_out)
{
   wv_sync();
   sh_copy_words((WV_LDS i32 *)L, hdr->w, SH_BACK_HDR_WORDS);
   sh_copy_words((WV_LDS i32 *)SH_F(L), cc->image, (int)(offsetof(FrameLds, BC) / 4));
   wv_sync();
   LANE0 { L->cs = cs; L->packet_off = pkt_off; SH_F(L)->g = cs; }
   wv_sync();
   const int ret = sh_frame_back_wave(L, gs, frame_size, (i16 *)0, (i16 *)0, (i16 *)0, out, (const SeControl *)0, 1, (const i32 *)0, (CeltCont *)0, 1);
   oa_sh_back_finish(L, gs, ret, out, out_cap, len_out, rng_out);
}
WV_DEVN int oa_sh_back_frame(WV_LDS ShLds *L, OaShStream *gs, int frame_size, u8 *out, int out_cap, i16 *pcm_hp /* the stream's slot: the front kernel's high-passed input */, i16 *pcm_celt, i16 *tmp_prefill, CeltScratch *cs, const ShCont *ct, i32 *len_out, u32 *rng_out, const i32 *tr = nullptr /* the transient pre-pass's record of the stream */,
      CeltCont *cut = nullptr, ShBackHdr *hdr = nullptr)
{
   WV_LDS ShShared *sh = &L->sh; WV_LDS OaShScalars *st = &L->st;
   sh_copy_words((WV_LDS i32 *)&L->cfg, (const i32 *)&gs->cfg, (int)(sizeof(OaShConfig) / 4));
   sh_copy_words((WV_LDS i32 *)sh, (const i32 *)&ct->sh, (int)(sizeof(ShShared) / 4));
   sh_copy_words((WV_LDS i32 *)st, (const i32 *)&ct->st, (int)(sizeof(OaShScalars) / 4));
   sh_copy_words((WV_LDS i32 *)&L->ec, (const i32 *)&ct->ec, (int)(sizeof(EcCtx) / 4));
   sh_copy_words((WV_LDS i32 *)SH_PKT(L), (const i32 *)ct->packet, (OA_MAX_PACKET + 4) / 4);
   wv_sync();
   const SeControl sc = ct->sc;
   const int celt_only = wv_uni(st->mode) == OA_MODE_CELT_ONLY;                              /* no SILK layer: nothing for the flags and the reservoir to do, silk_nBytes as opus_encode_frame_native starts it */
   LANE0 { L->cs = cs; sh->silk_in_lds = 0; sh->r[5] = 1; }
   if (!celt_only) LANE0 {
      /* what silk_Encode does once the channels are coded (enc_API.c:522-545): VAD / LBRR flags into the payload's first bits, the bit reservoir */
      EcCtx e_; ec_ld(&e_, &L->ec); EcCtx *e = &e_; WV_LDS u8 *buf = SH_PKT(L) + 1;
      const int nBytesOut = (k_ec_tell(EC_PASS) + 7) >> 3;
      k_ec_enc_patch_initial_bits(EC_PASS, (unsigned)ct->silk_flags, (unsigned)ct->silk_flag_bits);
      ec_st(&L->ec, e);
      const int nb = ct->silk_dtx ? 0 : nBytesOut;
      i32 ex = gs->silk.nBitsExceeded + nb * 8 - (sc.bitRate * sc.payloadSize_ms) / 1000;
      gs->silk.nBitsExceeded = se_limit(ex, 0, 10000);
      sh->r[5] = nb;
   }
   const int silk_nBytes = wv_uni(sh->r[5]);
   const int ret = sh_frame_back_wave(L, gs, frame_size, pcm_hp, pcm_celt, tmp_prefill, out, &sc, silk_nBytes, tr, cut);
   if (ret == OA_CUT) { wv_sync(); sh_copy_words(hdr->w, (const WV_LDS i32 *)L, SH_BACK_HDR_WORDS); wv_sync(); return 1; }
   oa_sh_back_finish(L, gs, ret, out, out_cap, len_out, rng_out);
   return 0;
}

/* ---------------- quantiser, reference form: one wave per stream on the one-kernel path's own stage function (se_frame_quant_wave).  OPUS_AMD_SH_SPLIT=2 selects it: the
 * split's data flow can then be checked apart from the 16-streams-per-wave kernel below ---------------- */
WV_DEVN void oa_sh_quant0_frame(WV_LDS ShLds *L, OaShStream *gs, ShCont *ct, SeRateScratch *G)
{
   WV_LDS SilkEncLds *S = &L->S;
   WV_LDS OaSilkEnc *E = se_st(S);
   sh_copy_words((WV_LDS i32 *)&L->cfg, (const i32 *)&gs->cfg, (int)(sizeof(OaShConfig) / 4));
   wv_sync();
   const int CC = L->cfg.channels;
   se_state_copy_wave((WV_LDS i32 *)se_st(S), (const i32 *)&gs->silk, CC, 1);
   sh_copy_words((WV_LDS i32 *)&L->ec, (const i32 *)&ct->ec, (int)(sizeof(EcCtx) / 4));
   sh_copy_words((WV_LDS i32 *)SH_PKT(L), (const i32 *)ct->packet, (OA_MAX_PACKET + 4) / 4);
   wv_sync();
   const int nq = wv_uni(ct->nq);
   for (int j = 0; j < nq; j++) {
      const ShQuantCh *q = &ct->q[j];
      const int n = wv_uni(q->chan);
      WV_LDS OaSilkEncChannel *c = &E->ch[n];
      WV_LDS SeEncCtrl *ctl = &S->ctl;
      wv_sync();
      FOR_LANES(i, 32) ctl->PredCoef_Q12[i >> 4][i & 15] = q->fr.PredCoef_Q12[i];
      FOR_LANES(i, 20) ctl->LTPCoef_Q14[i] = q->fr.LTPCoef_Q14[i];
      FOR_LANES(i, 4 * 24) ctl->AR_Q13[i] = q->fr.AR_Q13[i];
      FOR_LANES(i, 4) {
         ctl->HarmShapeGain_Q14[i] = q->fr.HarmShapeGain_Q14[i]; ctl->Tilt_Q14[i] = q->fr.Tilt_Q14[i]; ctl->LF_shp_Q14[i] = q->fr.LF_shp_Q14[i];
         ctl->Gains_Q16[i] = q->fr.Gains_Q16[i]; ctl->pitchL[i] = q->fr.pitchL[i]; ctl->GainsUnq_Q16[i] = q->GainsUnq_Q16[i];
      }
      FOR_LANES(i, c->frame_length) c->x_buf[c->ltp_mem_length + i] = q->x16[i];          /* (LDS copy only: the front kernel has moved x_buf on already) */
      if (wv_lane() == 0) { ctl->Lambda_Q10 = q->fr.Lambda_Q10; ctl->LTP_scale_Q14 = q->fr.LTP_scale_Q14; ctl->lastGainIndexPrev = q->lastGainIndexPrev; c->nsq_reset_req = q->nsq_reset; }
      wv_sync();
      se_frame_quant_wave(S, c, &L->ec, SH_PKT(L) + 1, wv_uni(q->condCoding), wv_uni(q->maxBits), wv_uni(q->useCBR), G, &gs->lbrr);
      wv_sync();
      OaSilkEncChannel *gc = &gs->silk.ch[n];
      sh_copy_words((i32 *)&gs->silk.tail[n], (const WV_LDS i32 *)&E->tail[n], SE_TAIL_WORDS);
      sh_copy_words((i32 *)&gc->indices, (const WV_LDS i32 *)&c->indices, (int)(sizeof(OaSilkEncIndices) / 4));
      if (wv_lane() == 0) { gc->LastGainIndex = c->LastGainIndex; gc->ec_prevLagIndex = c->ec_prevLagIndex; gc->ec_prevSignalType = c->ec_prevSignalType; }
   }
   wv_sync();
   sh_copy_words((i32 *)&ct->ec, (const WV_LDS i32 *)&L->ec, (int)(sizeof(EcCtx) / 4));
   sh_copy_words((i32 *)ct->packet, (const WV_LDS i32 *)SH_PKT(L), (OA_MAX_PACKET + 4) / 4);
   wv_sync();
}

/* ---------------- quantiser: 16 streams per wave ---------------- */
/* the rate-control loop's variables of one stream (silk/fixed/encode_frame_FIX.c:100-120): in LDS, so that the wave's registers belong to whichever stage is running */
struct SqRate {
   EcCtx ec, ec_copy, ec_copy2;
   i32 iter, done, need, gainMult_Q8, found_lower, found_upper, gainsID, gainsID_lower, gainsID_upper, nBits, nBits_lower, nBits_upper, gainMult_lower, gainMult_upper, LastGainIndex_copy2;
   i32 gain_lock[4], best_gain_mult[4], best_sum[4];
   i32 seed_copy, ec_prevLagIndex_copy, ec_prevSignalType_copy;
   i32 condCoding, maxBits, useCBR, lastGainIndexPrev, snap_now, fin, use_lower, chan, nsq_reset;
};
struct SqStream {                                        /* a stream's slice of the wave's LDS */
   OaNsqFrame fr;                                        /* live copy: the rate loop changes Gains_Q16, Lambda_Q10, Seed */
   OaNsqCfg cfg;
   OaSilkEncIndices ix;
   i32 nb_subfr, predictLPCOrder, fs_kHz, ec_prevSignalType, ec_prevLagIndex;     /* what se_encode_indices reads of the channel (and the two words it updates) */
   i32 LastGainIndex, GainsUnq_Q16[4];
   SqRate rc;
   i32 wk[40];
   i8 pulses[SE_MAX_FRAME + 16];
};
struct SqLds { SqStream s[16]; };
/* per-wave HBM scratch: the quantiser's tile (silk_frame.h layout: histories, whitened copies, scalars, delayed-decision rings) + per stream the "lower" snapshot of the rate loop */
struct SqSnap { OaSilkNsqState nsq; u8 ec_buf_copy[OA_MAX_PACKET + 4]; };
#define SQ_TILE_WORDS ((size_t)(OA_SILK_HIST_ROWS * 16 * 2 + OA_NSQ_S_ROWS * 16 + OA_SILK_HIST_ROWS * 16 + 5 * OA_SILK_DD * 64))      /* = oa_nsq_tile_words(16), silk_frame.h */
#define SQ_WAVE_SCRATCH_BYTES (SQ_TILE_WORDS * 4 + 16 * sizeof(SqSnap))

WV_DEV NsqMem sq_mem(i32 *tile, int t)
{
   NsqMem m; const int R = OA_SILK_HIST_ROWS, T = 16;
   m.shp = tile + t; m.q15 = tile + R * T + t; m.scal = tile + 2 * R * T + t;
   m.xq = (i16 *)(tile + 2 * R * T + OA_NSQ_S_ROWS * T) + t; m.wh = (i16 *)(tile + 2 * R * T + OA_NSQ_S_ROWS * T + R * T / 2) + t;
   m.T = T; m.len = 2 * R; m.base = 0;
   return m;
}
/* the channel's quantiser state, stream record -> tile column: the quad's four lanes take every fourth word */
/* reset: the state starts over instead (the front kernel's request, ShQuantCh.nsq_reset): all zero, lagPrev = 100, prev_gain_Q16 = 1.0 */
WV_DEV void sq_tile_load(const NsqMem &m, const OaSilkNsqState *g, int mem, int kk, int reset)
{
   const int T = m.T;
   if (reset) {
      for (int i = kk; i < mem; i += 4) { m.shp[i * T] = 0; m.xq[i * T] = 0; }
      for (int i = kk; i < 16; i += 4) m.scal[(OA_NSQ_S_LPC + i) * T] = 0;
      for (int i = kk; i < 24; i += 4) m.scal[(OA_NSQ_S_AR2 + i) * T] = 0;
      if (kk == 0) { m.scal[OA_NSQ_S_LF_AR * T] = 0; m.scal[OA_NSQ_S_DIFF * T] = 0; m.scal[OA_NSQ_S_LAGPREV * T] = 100; m.scal[OA_NSQ_S_PREVGAIN * T] = 65536; }
      return;
   }
   for (int i = kk; i < mem; i += 4) { m.shp[i * T] = g->sLTP_shp_Q14[i]; m.xq[i * T] = g->xq[i]; }
   for (int i = kk; i < 16; i += 4) m.scal[(OA_NSQ_S_LPC + i) * T] = g->sLPC_Q14[i];
   for (int i = kk; i < 24; i += 4) m.scal[(OA_NSQ_S_AR2 + i) * T] = g->sAR2_Q14[i];
   if (kk == 0) { m.scal[OA_NSQ_S_LF_AR * T] = g->sLF_AR_shp_Q14; m.scal[OA_NSQ_S_DIFF * T] = g->sDiff_shp_Q14; m.scal[OA_NSQ_S_LAGPREV * T] = g->lagPrev; m.scal[OA_NSQ_S_PREVGAIN * T] = g->prev_gain_Q16; }
}
/* ... and back, as silk_NSQ_del_dec leaves silk_nsq_state (NSQ_del_dec.c:299-311: histories moved down by one frame, the frame itself still behind them) */
WV_DEV void sq_tile_store(const NsqMem &m, OaSilkNsqState *g, int mem, int frame, int ltp_end, int kk)
{
   const int T = m.T;
   for (int i = kk; i < mem; i += 4) { g->sLTP_shp_Q14[i] = m.shp[(frame + i) * T]; g->xq[i] = m.xq[(frame + i) * T]; }
   for (int i = kk; i < frame; i += 4) { g->sLTP_shp_Q14[mem + i] = m.shp[(mem + i) * T]; g->xq[mem + i] = m.xq[(mem + i) * T]; }
   for (int i = kk; i < 16; i += 4) g->sLPC_Q14[i] = m.scal[(OA_NSQ_S_LPC + i) * T];
   for (int i = kk; i < 24; i += 4) g->sAR2_Q14[i] = m.scal[(OA_NSQ_S_AR2 + i) * T];
   if (kk == 0) {
      g->sLF_AR_shp_Q14 = m.scal[OA_NSQ_S_LF_AR * T]; g->sDiff_shp_Q14 = m.scal[OA_NSQ_S_DIFF * T]; g->lagPrev = m.scal[OA_NSQ_S_LAGPREV * T]; g->prev_gain_Q16 = m.scal[OA_NSQ_S_PREVGAIN * T];
      g->sLTP_shp_buf_idx = mem + frame; g->sLTP_buf_idx = ltp_end; g->rewhite_flag = 0;
   }
}
/* the same words between two records (the rate loop's "lower" snapshot back into the stream record) */
WV_DEV void sq_state_copy(OaSilkNsqState *d, const OaSilkNsqState *g, int mem, int frame, int kk)
{
   for (int i = kk; i < mem + frame; i += 4) { d->sLTP_shp_Q14[i] = g->sLTP_shp_Q14[i]; d->xq[i] = g->xq[i]; }
   for (int i = kk; i < 16; i += 4) d->sLPC_Q14[i] = g->sLPC_Q14[i];
   for (int i = kk; i < 24; i += 4) d->sAR2_Q14[i] = g->sAR2_Q14[i];
   if (kk == 0) {
      d->sLF_AR_shp_Q14 = g->sLF_AR_shp_Q14; d->sDiff_shp_Q14 = g->sDiff_shp_Q14; d->lagPrev = g->lagPrev; d->prev_gain_Q16 = g->prev_gain_Q16;
      d->sLTP_shp_buf_idx = g->sLTP_shp_buf_idx; d->sLTP_buf_idx = g->sLTP_buf_idx; d->rewhite_flag = g->rewhite_flag;
   }
}
WV_DEV OaNsqCfg sq_cfg_ld(const WV_LDS OaNsqCfg *p) { OaNsqCfg c; c.fs_kHz = p->fs_kHz; c.nb_subfr = p->nb_subfr; c.predictLPCOrder = p->predictLPCOrder; c.shapingLPCOrder = p->shapingLPCOrder; c.nStatesDelayedDecision = p->nStatesDelayedDecision; c.warping_Q16 = p->warping_Q16; return c; }
WV_DEV bool sq_cfg_eq(const OaNsqCfg &a, const OaNsqCfg &b)
{ return a.fs_kHz == b.fs_kHz && a.nb_subfr == b.nb_subfr && a.predictLPCOrder == b.predictLPCOrder && a.shapingLPCOrder == b.shapingLPCOrder && a.nStatesDelayedDecision == b.nStatesDelayedDecision && a.warping_Q16 == b.warping_Q16; }

/* job -> the stream's LDS slice (the quad's four lanes share the copy) and the start of the rate loop (:170-185), coder state from / to *ecg */
WV_DEVN void sq_job_open(WV_LDS SqStream *me, const ShQuantCh *job, const EcCtx *ecg, int first_job, int kk)
{
   WV_LDS i32 *d = (WV_LDS i32 *)&me->fr; const i32 *g = (const i32 *)&job->fr;
   for (int i = kk; i < (int)(sizeof(OaNsqFrame) / 4); i += 4) d[i] = g[i];
   d = (WV_LDS i32 *)&me->cfg; g = (const i32 *)&job->cfg;
   for (int i = kk; i < (int)(sizeof(OaNsqCfg) / 4); i += 4) d[i] = g[i];
   d = (WV_LDS i32 *)&me->ix; g = (const i32 *)&job->indices;
   for (int i = kk; i < (int)(sizeof(OaSilkEncIndices) / 4); i += 4) d[i] = g[i];
   me->GainsUnq_Q16[kk] = job->GainsUnq_Q16[kk];
   if (kk == 0) {
      WV_LDS SqRate *r = &me->rc;
      me->nb_subfr = job->cfg.nb_subfr; me->predictLPCOrder = job->cfg.predictLPCOrder; me->fs_kHz = job->cfg.fs_kHz;
      me->ec_prevSignalType = job->ec_prevSignalType; me->ec_prevLagIndex = job->ec_prevLagIndex; me->LastGainIndex = job->LastGainIndex;
      if (first_job) { WV_LDS i32 *e = (WV_LDS i32 *)&r->ec; const i32 *s = (const i32 *)ecg; for (int i = 0; i < (int)(sizeof(EcCtx) / 4); i++) e[i] = s[i]; }
      ec_cp_lds(&r->ec_copy, &r->ec); ec_cp_lds(&r->ec_copy2, &r->ec);
      r->iter = 0; r->done = 0; r->need = 0; r->gainMult_Q8 = SE_FIX(1, 8); r->found_lower = 0; r->found_upper = 0;
      { i32 id = 0; for (int k = 0; k < job->cfg.nb_subfr; k++) id = add32(job->indices.GainsIndices[k], shl32(id, 8)); r->gainsID = id; }      /* silk_gains_ID (straight from the job: the slice's copy is still being written by the quad's other lanes) */
      r->gainsID_lower = -1; r->gainsID_upper = -1; r->nBits = 0; r->nBits_lower = 0; r->nBits_upper = 0; r->gainMult_lower = 0; r->gainMult_upper = 0; r->LastGainIndex_copy2 = 0;
      for (int i = 0; i < 4; i++) { r->gain_lock[i] = 0; r->best_gain_mult[i] = 0; r->best_sum[i] = 0; }
      r->seed_copy = job->indices.Seed; r->ec_prevLagIndex_copy = job->ec_prevLagIndex; r->ec_prevSignalType_copy = job->ec_prevSignalType;
      r->condCoding = job->condCoding; r->maxBits = job->maxBits; r->useCBR = job->useCBR; r->lastGainIndexPrev = job->lastGainIndexPrev; r->chan = job->chan; r->nsq_reset = job->nsq_reset;
      r->snap_now = 0; r->fin = 0; r->use_lower = 0;
   }
}
/* does this pass quantise?  (a gain set met before: its bit count is known, :186-191) */
WV_DEV void sq_rate_pre(WV_LDS SqStream *me)
{
   WV_LDS SqRate *r = &me->rc;
   r->need = 0;
   if (r->gainsID == r->gainsID_lower) r->nBits = r->nBits_lower;
   else if (r->gainsID == r->gainsID_upper) r->nBits = r->nBits_upper;
   else {
      if (r->iter > 0) { ec_cp_lds(&r->ec, &r->ec_copy); me->ix.Seed = (i8)r->seed_copy; me->ec_prevLagIndex = r->ec_prevLagIndex_copy; me->ec_prevSignalType = r->ec_prevSignalType_copy; }
      me->fr.Seed = me->ix.Seed;
      r->need = 1;
   }
}
/* entropy coding of the pass and the loop's decisions (:214-366), one lane for the stream; buf: the stream's payload bytes in HBM */
WV_DEVN void sq_rate_post(WV_LDS SqStream *me, u8 *buf, SqSnap *snap)
{
   WV_LDS SqRate *r = &me->rc;
   const int maxIter = 6, iter = r->iter, maxBits = r->maxBits, condCoding = r->condCoding, nb_subfr = me->nb_subfr, subfr_length = 5 * me->fs_kHz, frame_length = nb_subfr * subfr_length;
   const int bits_margin = r->useCBR ? 5 : maxBits / 4;
   int brk = 0;
   r->snap_now = 0; r->fin = 0; r->use_lower = 0;
   if (r->need) {
      if (iter == maxIter && !r->found_lower) ec_cp_lds(&r->ec_copy2, &r->ec);
      EcCtx ec_; ec_ld(&ec_, &r->ec); EcCtx *e = &ec_;
      se_encode_indices(me, &me->ix, e, buf, condCoding);
      se_encode_pulses(e, buf, me->ix.signalType, me->ix.quantOffsetType, (WV_LDS i8 *)me->pulses, frame_length, (WV_LDS i32 *)me->wk);
      int nb = k_ec_tell(e, buf);
      if (iter == maxIter && !r->found_lower && nb > maxBits) {
         ec_ld(&ec_, &r->ec_copy2);
         me->LastGainIndex = r->lastGainIndexPrev;
         for (int i = 0; i < nb_subfr; i++) me->ix.GainsIndices[i] = 4;
         if (condCoding != SE_CODE_CONDITIONALLY) me->ix.GainsIndices[0] = (i8)r->lastGainIndexPrev;
         me->ec_prevLagIndex = r->ec_prevLagIndex_copy; me->ec_prevSignalType = r->ec_prevSignalType_copy;
         for (int i = 0; i < frame_length; i++) me->pulses[i] = 0;
         se_encode_indices(me, &me->ix, e, buf, condCoding);
         se_encode_pulses(e, buf, me->ix.signalType, me->ix.quantOffsetType, (WV_LDS i8 *)me->pulses, frame_length, (WV_LDS i32 *)me->wk);
         nb = k_ec_tell(e, buf);
      }
      ec_st(&r->ec, &ec_);
      r->nBits = nb;
      if (r->useCBR == 0 && iter == 0 && nb <= maxBits) brk = 1;
   }
   const i32 nBits = r->nBits, gainsID = r->gainsID;
   if (!brk) {
      if (iter == maxIter) {
         if (r->found_lower && (gainsID == r->gainsID_lower || nBits > maxBits)) {
            ec_cp_lds(&r->ec, &r->ec_copy2); for (u32 i = 0; i < r->ec_copy2.offs; i++) buf[i] = snap->ec_buf_copy[i]; me->LastGainIndex = r->LastGainIndex_copy2;
            r->use_lower = 1;
         }
         brk = 1;
      } else {
         if (nBits > maxBits) {
            if (r->found_lower == 0 && iter >= 2) { me->fr.Lambda_Q10 = me->fr.Lambda_Q10 + (me->fr.Lambda_Q10 >> 1); r->found_upper = 0; r->gainsID_upper = -1; }
            else { r->found_upper = 1; r->nBits_upper = nBits; r->gainMult_upper = r->gainMult_Q8; r->gainsID_upper = gainsID; }
         } else if (nBits < maxBits - bits_margin) {
            r->found_lower = 1; r->nBits_lower = nBits; r->gainMult_lower = r->gainMult_Q8;
            if (gainsID != r->gainsID_lower) {
               r->gainsID_lower = gainsID;
               ec_cp_lds(&r->ec_copy2, &r->ec); for (u32 i = 0; i < r->ec.offs; i++) snap->ec_buf_copy[i] = buf[i];
               r->LastGainIndex_copy2 = me->LastGainIndex;
               r->snap_now = 1;
            }
         } else brk = 1;
      }
   }
   if (brk) { r->done = 1; r->fin = 1; return; }
   int gainMult_Q8 = r->gainMult_Q8;
   if (!r->found_lower && nBits > maxBits) {
      for (int i = 0; i < nb_subfr; i++) {
         int sum = 0;
         for (int t = i * subfr_length; t < (i + 1) * subfr_length; t++) sum += iabs((i32)me->pulses[t]);
         if (iter == 0 || (sum < r->best_sum[i] && !r->gain_lock[i])) { r->best_sum[i] = sum; r->best_gain_mult[i] = (i16)gainMult_Q8; } else r->gain_lock[i] = 1;
      }
   }
   if ((r->found_lower & r->found_upper) == 0) {
      if (nBits > maxBits) gainMult_Q8 = imin(1024, gainMult_Q8 * 3 / 2); else gainMult_Q8 = imax(64, gainMult_Q8 * 4 / 5);
      gainMult_Q8 = (i16)gainMult_Q8;
   } else {
      const i32 gl = r->gainMult_lower, gu = r->gainMult_upper;
      gainMult_Q8 = gl + ((gu - gl) * (maxBits - r->nBits_lower)) / (r->nBits_upper - r->nBits_lower);
      gainMult_Q8 = (i16)gainMult_Q8;
      if (gainMult_Q8 > gl + ((gu - gl) >> 2)) gainMult_Q8 = (i16)(gl + ((gu - gl) >> 2));
      else if (gainMult_Q8 < gu - ((gu - gl) >> 2)) gainMult_Q8 = (i16)(gu - ((gu - gl) >> 2));
   }
   r->gainMult_Q8 = gainMult_Q8;
   for (int i = 0; i < nb_subfr; i++) { const i16 tmp = r->gain_lock[i] ? (i16)r->best_gain_mult[i] : (i16)gainMult_Q8; me->fr.Gains_Q16[i] = sk_shl_sat(sk_mulwb(me->GainsUnq_Q16[i], tmp), 8); }
   me->LastGainIndex = r->lastGainIndexPrev;
   se_gains_quant((WV_LDS i8 *)me->ix.GainsIndices, (WV_LDS i32 *)me->fr.Gains_Q16, (WV_LDS i32 *)&me->LastGainIndex, condCoding == SE_CODE_CONDITIONALLY, nb_subfr);
   r->gainsID = se_gains_ID((const WV_LDS i8 *)me->ix.GainsIndices, nb_subfr);
   r->iter = iter + 1;
}
/* the LBRR pass of a stream (lane kk == 0): the frame's indices with the first gain index raised, the gains those indices stand for; the stream's own gains and indices wait in wk */
WV_DEV void sq_lbrr_pre(WV_LDS SqStream *me, const ShQuantCh *job)
{
   WV_LDS i32 *wk = (WV_LDS i32 *)me->wk; const WV_LDS i32 *ixw = (const WV_LDS i32 *)&me->ix;
   for (int k = 0; k < 4; k++) wk[k] = me->fr.Gains_Q16[k];
   for (int k = 0; k < (int)(sizeof(OaSilkEncIndices) / 4); k++) wk[4 + k] = ixw[k];
   /* first frame of the packet, or no LBRR frame before this one: LBRRprevLastGainIndex = LastGainIndex, the first index raised (:404-410) */
   int prev = job->LBRRprevLastGainIndex;
   if (job->lbrr_fi == 0 || job->lbrr_prev_flag == 0) {
      prev = me->LastGainIndex;
      me->ix.GainsIndices[0] = (i8)imin(me->ix.GainsIndices[0] + job->LBRR_GainIncreases, 64 - 1);
   }
   i32 g[4]; i8 gi[4];
   for (int k = 0; k < 4; k++) gi[k] = me->ix.GainsIndices[k];
   se_gains_dequant(g, gi, &prev, me->rc.condCoding == SE_CODE_CONDITIONALLY, me->nb_subfr);
   for (int k = 0; k < me->nb_subfr; k++) me->fr.Gains_Q16[k] = g[k];
   wk[4 + (int)(sizeof(OaSilkEncIndices) / 4)] = prev;
   me->fr.Seed = me->ix.Seed;
}
/* ... and after the pass (the quad's four lanes): pulses and indices into the side-stream store, the flag and the gain predictor into the channel record, the stream's own values back */
WV_DEV void sq_lbrr_post(WV_LDS SqStream *me, const ShQuantCh *job, OaSilkLbrr *lb, OaSilkEncChannel *gc, int kk)
{
   const int frame_length = me->nb_subfr * 5 * me->fs_kHz, chn = job->chan, fi = job->lbrr_fi, NIX = (int)(sizeof(OaSilkEncIndices) / 4);
   { i32 *d = (i32 *)lb->pulses[chn][fi]; const WV_LDS i32 *g = (const WV_LDS i32 *)me->pulses; for (int i = kk; i < frame_length / 4; i += 4) d[i] = g[i]; }
   { i32 *d = (i32 *)&lb->indices[chn][fi]; const WV_LDS i32 *g = (const WV_LDS i32 *)&me->ix; for (int i = kk; i < NIX; i += 4) d[i] = g[i]; }
}
WV_DEV void sq_lbrr_restore(WV_LDS SqStream *me, const ShQuantCh *job, OaSilkEncChannel *gc, int kk)
{
   const int NIX = (int)(sizeof(OaSilkEncIndices) / 4);
   if (kk == 0) {
      WV_LDS i32 *wk = (WV_LDS i32 *)me->wk; WV_LDS i32 *ixw = (WV_LDS i32 *)&me->ix;
      gc->LBRR_flags[job->lbrr_fi] = 1; gc->LBRRprevLastGainIndex = wk[4 + NIX];
      for (int k = 0; k < 4; k++) me->fr.Gains_Q16[k] = wk[k];
      for (int k = 0; k < NIX; k++) ixw[k] = wk[4 + k];
   }
}
/* one silk_NSQ_del_dec pass over the wave's 16 streams; `same`: this quad's stream takes part (its parameters are *sp == its own slice), the others keep the collectives in step */
WV_DEVN void sq_nsq_pass(const OaNsqCfg cfg, NsqMem mown, i32 *ring, WV_LDS SqStream *sp, const i16 *x16, const OaSilkNsqState *gnsq, int same, int kk, int reset)
{
   if (same) sq_tile_load(mown, gnsq, 20 * cfg.fs_kHz, kk, reset);                          /* every pass starts from the state the frame started from: the stream record is not written before the loop ends */
   wv_sync();
   switch (cfg.shapingLPCOrder) {
   case 24: silk_nsq_dd_wave<24>(cfg, mown, ring, (const WV_LDS OaNsqFrame *)&sp->fr, x16, (WV_LDS i8 *)sp->pulses, (WV_LDS i8 *)&sp->ix.Seed, same != 0); break;
   case 16: silk_nsq_dd_wave<16>(cfg, mown, ring, (const WV_LDS OaNsqFrame *)&sp->fr, x16, (WV_LDS i8 *)sp->pulses, (WV_LDS i8 *)&sp->ix.Seed, same != 0); break;
   default: silk_nsq_dd_wave<0>(cfg, mown, ring, (const WV_LDS OaNsqFrame *)&sp->fr, x16, (WV_LDS i8 *)sp->pulses, (WV_LDS i8 *)&sp->ix.Seed, same != 0); break;
   }
   wv_sync();
}
/* the quad's part of a snapshot / of the end of the loop: quantiser state tile -> HBM, and what else the frame leaves in the channel record */
WV_DEVN void sq_store(WV_LDS SqStream *me, const NsqMem mown, OaSilkEncChannel *gc, OaSilkEncTail *gt, SqSnap *snap, int snap_q, int fin_q, int low_q, int kk)
{
   const int nb_subfr = me->nb_subfr, subfr_length = 5 * me->fs_kHz, frame_length = nb_subfr * subfr_length, ltp_mem = 20 * me->fs_kHz;
   const int interp = me->ix.NLSFInterpCoef_Q2 == 4 ? 0 : 1, voiced = me->ix.signalType == SE_TYPE_VOICED;
   const int k_r = voiced && interp && nb_subfr == 4 ? 2 : 0;                    /* the last subframe that re-whitened the LTP state (:205, :232) */
   const int ltp_end = ltp_mem + (nb_subfr - k_r) * subfr_length;
   if (snap_q) sq_tile_store(mown, &snap->nsq, ltp_mem, frame_length, ltp_end, kk);
   if (fin_q) {
      if (low_q) sq_state_copy(&gt->nsq, &snap->nsq, ltp_mem, frame_length, kk);
      else sq_tile_store(mown, &gt->nsq, ltp_mem, frame_length, ltp_end, kk);
      i32 *d = (i32 *)&gc->indices; const WV_LDS i32 *g = (const WV_LDS i32 *)&me->ix;
      for (int i = kk; i < (int)(sizeof(OaSilkEncIndices) / 4); i += 4) d[i] = g[i];
      d = (i32 *)gt->pulses; g = (const WV_LDS i32 *)me->pulses;
      for (int i = kk; i < (frame_length + (frame_length & 15 ? 16 : 0)) / 4; i += 4) d[i] = g[i];      /* (+ the zeros silk_encode_pulses pads a frame that is not a multiple of 16 with) */
      if (kk == 0) { gc->LastGainIndex = me->LastGainIndex; gc->ec_prevLagIndex = me->ec_prevLagIndex; gc->ec_prevSignalType = me->ec_prevSignalType; }
   }
}

WV_DEV void sq_quant_tile_wave(WV_LDS SqLds *Q, OaShStream *streams, ShCont *conts, int first, int nstreams, i32 *tile, SqSnap *snaps)
{
   const int lane = wv_lane(), kk = lane & 3, qd = lane >> 2;
   const int sidx = first + qd < nstreams ? first + qd : first;
   ShCont *ct = conts + sidx;
   const bool valid = first + qd < nstreams && ct->kind == SH_CONT_FAST;
   const int nq = valid ? ct->nq : 0;
   WV_LDS SqStream *me = &Q->s[qd];
   SqSnap *snap = snaps + qd;
   i32 *ring = tile + (SQ_TILE_WORDS - 5 * OA_SILK_DD * 64);
   const NsqMem mown = sq_mem(tile, qd);
   u8 *buf = ct->packet + 1;
   for (int j = 0; j < 2; j++) {
      if (!wv_ballot(j < nq)) break;
      const bool has = j < nq;
      const ShQuantCh *job = &ct->q[has ? j : 0];
      OaSilkEncChannel *gc = &streams[sidx].silk.ch[has ? job->chan : 0];
      OaSilkEncTail *gt = &streams[sidx].silk.tail[has ? job->chan : 0];
      wv_sync();
      if (has) sq_job_open(me, job, &ct->ec, j == 0, kk); else if (kk == 0) me->rc.done = 1;
      if (has && job->nsq_reset) { i32 *z = (i32 *)&gt->nsq; for (int i = kk; i < (int)(sizeof(OaSilkNsqState) / 4); i += 4) z[i] = 0; }     /* the record too: the passes start from constants (sq_tile_load), the store at the end writes what a frame leaves */
      wv_sync();
      /* silk_LBRR_encode_FIX (encode_frame_FIX.c:172, :392-455) ahead of the rate loop: the same frame once more with raised gains, from the same quantiser state (every pass
       * loads it from the record, nothing is written back); indices and pulses go to the stream's side-stream store for the next packet */
      const int lb_on = has && job->lbrr_on;
      if (wv_ballot(lb_on)) {
         if (kk == 0 && lb_on) sq_lbrr_pre(me, job);
         wv_sync();
         unsigned long long pend = wv_ballot(lb_on);
         while (pend) {
            const int lead = (int)(__builtin_ctzll(pend) >> 2);
            const OaNsqCfg cfg = sq_cfg_ld(&Q->s[lead].cfg);
            const int same = lb_on && sq_cfg_eq(sq_cfg_ld(&me->cfg), cfg);
            const int src = same ? qd : lead;
            sq_nsq_pass(cfg, mown, ring, &Q->s[src], conts[first + src].q[j].x16, &gt->nsq, same, kk, me->rc.nsq_reset);
            pend &= ~wv_ballot(same);
         }
         if (lb_on) sq_lbrr_post(me, job, &streams[sidx].lbrr, gc, kk);
         wv_sync();
         if (lb_on) sq_lbrr_restore(me, job, gc, kk);
         wv_sync();
      }
      /* the rate-control loop of silk_encode_frame_FIX (:170-370), one lane (kk == 0) per stream; the loop itself is the wave's, a stream that has converged sits out */
      for (;;) {
         if (kk == 0 && !me->rc.done) sq_rate_pre(me);
         wv_sync();
         const int need_q = !me->rc.done && me->rc.need;
         /* silk_NSQ_del_dec for the streams that need it: one pass per distinct configuration among them (almost always one) */
         unsigned long long pend = wv_ballot(need_q);
         while (pend) {
            const int lead = (int)(__builtin_ctzll(pend) >> 2);
            const OaNsqCfg cfg = sq_cfg_ld(&Q->s[lead].cfg);
            const int same = need_q && sq_cfg_eq(sq_cfg_ld(&me->cfg), cfg);
            const int src = same ? qd : lead;
            sq_nsq_pass(cfg, mown, ring, &Q->s[src], conts[first + src].q[j].x16, &gt->nsq, same, kk, me->rc.nsq_reset);
            pend &= ~wv_ballot(same);
         }
         if (kk == 0 && !me->rc.done) sq_rate_post(me, buf, snap);
         wv_sync();
         const int act = has && (me->rc.snap_now || me->rc.fin);
         if (act) sq_store(me, mown, gc, gt, snap, me->rc.snap_now, me->rc.fin, me->rc.use_lower, kk);
         wv_sync();
         if (kk == 0 && has) { me->rc.snap_now = 0; me->rc.fin = 0; }
         wv_sync();
         if (!wv_ballot(!me->rc.done)) break;
      }
   }
   if (valid && nq > 0 && kk == 0) { i32 *d = (i32 *)&ct->ec; const WV_LDS i32 *g = (const WV_LDS i32 *)&me->rc.ec; for (int i = 0; i < (int)(sizeof(EcCtx) / 4); i++) d[i] = g[i]; }
   wv_sync();
}
#endif

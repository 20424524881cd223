/* silk_dec_api.h — silk_Decode (silk/dec_API.c:142-470) on lane 0 of the stream's wave, and the LDS scratch it borrows from the CELT decoder.
 * One call decodes one 10/20 ms SILK frame of every internal channel, converts mid/side to left/right and resamples to the API rate into
 * L's staging area; the caller (oa_decode_packet) copies the staged PCM out with all lanes. */
#ifndef OPUS_AMD_SILK_DEC_API_H
#define OPUS_AMD_SILK_DEC_API_H
#include "silk_dec.h"

/* scratch overlaid on the CELT decoder's phase regions (free while SILK runs): A (7,680 B) and BC (8,640 B), contiguous = 16,320 B:
 * 5,904 B synthesis scratch + 4,096 B resampler staging + the hot part of the stream's SILK state (both channels, 6,168 B) so that the lane-0
 * code never waits on HBM (its first version kept the state in HBM: 1.2 ms per frame, two orders of magnitude behind one CPU core) */
struct SilkLdsA { i32 sLTP_Q15[640]; i32 res_Q14[80]; i32 sLPC_Q14[96]; union { i16 pulses[336]; i16 sLTP[320]; } u;   /* pulses are dead once the excitation exists */
                  i16 tmp[16]; i16 xq[2][324]; SdCtrl ctrl; };
struct SilkLdsB { i16 rs_out[2][960]; ResamplerLdsT<1> ring; };
struct SilkLdsAll { SilkLdsA a; SilkLdsB b; i32 hot[(OA_SILK_HOT_BYTES + 3) / 4]; };
struct SdDecControl { i32 nChannelsAPI, nChannelsInternal, API_sampleRate, internalSampleRate, payloadSize_ms; };

WV_DEV void sd_resample(WV_LDS OaSilkChannel *ch, WV_LDS SilkLdsB *B, WV_LDS i16 *out, WV_LDS i16 *in, int inLen)
{
   OaResamplerCfg c;
   c.resampler_function = ch->rs_cfg[0]; c.batchSize = ch->rs_cfg[1]; c.invRatio_Q16 = ch->rs_cfg[2]; c.FIR_Order = ch->rs_cfg[3]; c.FIR_Fracs = ch->rs_cfg[4];
   c.Fs_in_kHz = ch->rs_cfg[5]; c.Fs_out_kHz = ch->rs_cfg[6]; c.inputDelay = ch->rs_cfg[7]; c.coefs_id = ch->rs_cfg[8];
   silk_resampler_lane(c, &B->ring, ch->rs_rows, 1, in, inLen, out, 0);
}
WV_DEV void sd_resampler_init(WV_LDS OaSilkChannel *ch, i32 Fs_in, i32 Fs_out)
{
   OaResamplerCfg c;
   rs_init_cfg(&c, Fs_in, Fs_out, 0);
   ch->rs_cfg[0] = c.resampler_function; ch->rs_cfg[1] = c.batchSize; ch->rs_cfg[2] = c.invRatio_Q16; ch->rs_cfg[3] = c.FIR_Order; ch->rs_cfg[4] = c.FIR_Fracs;
   ch->rs_cfg[5] = c.Fs_in_kHz; ch->rs_cfg[6] = c.Fs_out_kHz; ch->rs_cfg[7] = c.inputDelay; ch->rs_cfg[8] = c.coefs_id;
   for (int i = 0; i < 90; i++) ch->rs_rows[i] = 0;
}

/* ---- the decoder-side resampler, wave-wide: silk_resampler (silk/resampler.c:183) for the up-sampling pairs (8/12/16 kHz -> 48 kHz), i.e.
 * silk_resampler_private_IIR_FIR (resampler_private_IIR_FIR.c:65).  Per <= 10 ms batch: the two allpass chains of the 2x stage (even / odd output
 * phase, resampler_private_up2_HQ.c:38) are independent recursions -> lanes 0 and 1; the 8-tap fractional FIR has one output per lane.
 * `buf` = 8 carried samples + 2*nIn new ones (int16), in the LTP scratch that is free by now. ---- */
WV_DEV int sd_resample_segment_wave(WV_LDS OaSilkChannel *ch, WV_LDS i16 *buf, WV_LDS i16 *out, const WV_LDS i16 *in, const WV_LDS i32 *delay, int ndelay, int len)
{
   const int lane = wv_lane();
   const i32 inc = ch->rs_cfg[2], batch = ch->rs_cfg[1];
   int no = 0;
   for (int done = 0; done < len;) {
      const int nIn = imin(len - done, batch);
      if (lane < 2) {
         const i16 *c = lane ? sk_resampler_up2_hq_1 : sk_resampler_up2_hq_0;
         i32 s0 = ch->rs_rows[3 * lane], s1 = ch->rs_rows[3 * lane + 1], s2 = ch->rs_rows[3 * lane + 2];
         for (int k = 0; k < nIn; k++) {
            const int kk = done + k;
            const i32 in32 = shl32(kk < ndelay ? delay[kk] : (i32)in[kk - ndelay], 10);
            i32 Y = in32 - s0, X = sk_mulwb(Y, c[0]);
            i32 o1 = s0 + X;  s0 = in32 + X;
            Y = o1 - s1;  X = sk_mulwb(Y, c[1]);
            const i32 o2 = s1 + X;  s1 = o1 + X;
            Y = o2 - s2;  X = sk_mlawb(Y, Y, c[2]);
            o1 = s2 + X;  s2 = o2 + X;
            buf[8 + 2 * k + lane] = (i16)sk_sat16(sk_rround(o1, 10));
         }
         ch->rs_rows[3 * lane] = s0; ch->rs_rows[3 * lane + 1] = s1; ch->rs_rows[3 * lane + 2] = s2;
      }
      wv_sync();
      const i32 max_index_Q16 = shl32(nIn, 17);
      const int nout = (int)(((i64)max_index_Q16 + inc - 1) / inc);
      for (int m = lane; m < nout; m += WV_WIDTH) {
         const i32 idx = m * inc;
         const int ti = sk_mulwb(idx & 0xFFFF, 12);
         const WV_LDS i16 *bp = &buf[idx >> 16];
         const i16 *t0 = &sk_resampler_frac_fir_12[4 * ti], *t1 = &sk_resampler_frac_fir_12[4 * (11 - ti)];
         i32 acc = sk_mulbb(bp[0], t0[0]);
         acc = sk_mlabb(acc, bp[1], t0[1]); acc = sk_mlabb(acc, bp[2], t0[2]); acc = sk_mlabb(acc, bp[3], t0[3]);
         acc = sk_mlabb(acc, bp[4], t1[3]); acc = sk_mlabb(acc, bp[5], t1[2]); acc = sk_mlabb(acc, bp[6], t1[1]); acc = sk_mlabb(acc, bp[7], t1[0]);
         out[no + m] = (i16)sk_sat16(sk_rround(acc, 15));
      }
      wv_sync();
      i32 tail = 0;
      if (lane < 8) tail = buf[2 * nIn + lane];
      wv_sync();
      if (lane < 8) buf[lane] = (i16)tail;
      wv_sync();
      no += nout; done += nIn;
   }
   return no;
}
WV_DEV void sd_resample_wave(WV_LDS OaSilkChannel *ch, WV_LDS SilkLdsA *A, WV_LDS SilkLdsB *B, WV_LDS i16 *out, WV_LDS i16 *in, int inLen)
{
   const int lane = wv_lane();
   if (ch->rs_cfg[0] != OA_RS_FN_IIR_FIR) {                      /* (48 kHz output always takes the path above; kept for other API rates) */
      LANE0 sd_resample(ch, B, out, in, inLen);
      return;
   }
   WV_LDS i16 *buf = (WV_LDS i16 *)A->sLTP_Q15;                   /* 2*160 + 8 int16 needed */
   const int Fs_in_kHz = ch->rs_cfg[5], Fs_out_kHz = ch->rs_cfg[6], inputDelay = ch->rs_cfg[7];
   wv_sync();
   if (lane < 8) buf[lane] = (i16)ch->rs_rows[OA_RS_ROW_FIR + lane];
   wv_sync();
   const int nNew = Fs_in_kHz - inputDelay;
   sd_resample_segment_wave(ch, buf, out, in, &ch->rs_rows[OA_RS_ROW_DELAY], inputDelay, Fs_in_kHz);
   sd_resample_segment_wave(ch, buf, out + Fs_out_kHz, in + nNew, &ch->rs_rows[OA_RS_ROW_DELAY], 0, inLen - Fs_in_kHz);
   if (lane < 8) ch->rs_rows[OA_RS_ROW_FIR + lane] = buf[lane];
   if (lane < inputDelay) ch->rs_rows[OA_RS_ROW_DELAY + lane] = in[inLen - inputDelay + lane];
   wv_sync();
}

/* silk_Decode (silk/dec_API.c:142) for one 10/20 ms frame of every internal channel; executed by the whole wave: entropy decoding, parameter
 * decoding and bookkeeping in lane-0 sections, synthesis and resampling wave-wide.  The range decoder is parked in *ecp between sections.
 * sd: the LDS-staged state; cng_exc: &OaSilkDec::cng_exc_buf_Q14[0][0] in HBM; shr: 8 shared words.
 * Returns the number of samples per channel staged in B->rs_out (at the API rate), or a negative OA_ERR_*. */
WV_DEVN int silk_decode_wave(WV_LDS OaSilkDec *sd, i32 *cng_exc, const SdDecControl dc, int lostFlag, int newPacketFlag, WV_LDS EcCtx *ecp, WV_LDS u8 *buf,
                             WV_LDS SilkLdsA *A, WV_LDS SilkLdsB *B, WV_LDS i32 *shr)
{
   const int lane = wv_lane();
   WV_LDS OaSilkChannel *cs = sd->ch;
   SdScratch S; S.cng_exc = cng_exc; S.sLTP_Q15 = A->sLTP_Q15; S.res_Q14 = A->res_Q14; S.sLPC_Q14 = A->sLPC_Q14; S.sLTP = A->u.sLTP; S.pulses = A->u.pulses; S.tmp = A->tmp; S.ctrl = &A->ctrl;
   enum { R_RET = 0, R_ONLYMID, R_PRED0, R_PRED1, R_HASSIDE, R_S2M, R_DEC };
   LANE0 {
      EcCtx ec_; EcCtx *e = &ec_; ec_ld(e, ecp);
      int ret = 0, decode_only_middle = 0;
      i32 MS_pred_Q13[2] = { 0, 0 };
      if (newPacketFlag) for (int n = 0; n < dc.nChannelsInternal; n++) cs[n].nFramesDecoded = 0;
      if (dc.nChannelsInternal > sd->nChannelsInternal) sd_reset(&cs[1], cng_exc + 320);                                /* mono -> stereo: init the side channel (:186) */
      const int stereo_to_mono = dc.nChannelsInternal == 1 && sd->nChannelsInternal == 2 && dc.internalSampleRate == 1000 * cs[0].fs_kHz;
      if (cs[0].nFramesDecoded == 0) {
         for (int n = 0; n < dc.nChannelsInternal; n++) {
            if (dc.payloadSize_ms == 0 || dc.payloadSize_ms == 10) { cs[n].nFramesPerPacket = 1; cs[n].nb_subfr = 2; }
            else if (dc.payloadSize_ms == 20) { cs[n].nFramesPerPacket = 1; cs[n].nb_subfr = 4; }
            else if (dc.payloadSize_ms == 40) { cs[n].nFramesPerPacket = 2; cs[n].nb_subfr = 4; }
            else if (dc.payloadSize_ms == 60) { cs[n].nFramesPerPacket = 3; cs[n].nb_subfr = 4; }
            else { ret = OA_ERR_INTERNAL; break; }
            const int fs_kHz_dec = (dc.internalSampleRate >> 10) + 1;
            if (fs_kHz_dec != 8 && fs_kHz_dec != 12 && fs_kHz_dec != 16) { ret = OA_ERR_INTERNAL; break; }
            if (sd_set_fs(&cs[n], fs_kHz_dec, dc.API_sampleRate)) sd_resampler_init(&cs[n], fs_kHz_dec * 1000, dc.API_sampleRate);
         }
      }
      if (ret == 0) {
         if (dc.nChannelsAPI == 2 && dc.nChannelsInternal == 2 && (sd->nChannelsAPI == 1 || sd->nChannelsInternal == 1)) {
            sd->pred_prev_Q13[0] = sd->pred_prev_Q13[1] = 0; sd->sSide[0] = sd->sSide[1] = 0;
            for (int i = 0; i < 9; i++) cs[1].rs_cfg[i] = cs[0].rs_cfg[i];
            for (int i = 0; i < 90; i++) cs[1].rs_rows[i] = cs[0].rs_rows[i];
         }
         sd->nChannelsAPI = dc.nChannelsAPI; sd->nChannelsInternal = dc.nChannelsInternal;
         if (lostFlag != SD_FLAG_PACKET_LOST && cs[0].nFramesDecoded == 0) {
            /* VAD and LBRR flags, then skip over any LBRR payload (:233-290) */
            for (int n = 0; n < dc.nChannelsInternal; n++) {
               for (int i = 0; i < cs[n].nFramesPerPacket; i++) cs[n].VAD_flags[i] = k_ec_dec_bit_logp(e, buf, 1);
               cs[n].LBRR_flag = k_ec_dec_bit_logp(e, buf, 1);
            }
            for (int n = 0; n < dc.nChannelsInternal; n++) {
               cs[n].LBRR_flags[0] = cs[n].LBRR_flags[1] = cs[n].LBRR_flags[2] = 0;
               if (cs[n].LBRR_flag) {
                  if (cs[n].nFramesPerPacket == 1) cs[n].LBRR_flags[0] = 1;
                  else {
                     const int sym = k_ec_dec_icdf(e, buf, &sk_lbrr_flags_icdf[cs[n].nFramesPerPacket == 2 ? 0 : 3], 8) + 1;
                     for (int i = 0; i < cs[n].nFramesPerPacket; i++) cs[n].LBRR_flags[i] = (sym >> i) & 1;
                  }
               }
            }
            if (lostFlag == SD_FLAG_DECODE_NORMAL) {
               for (int i = 0; i < cs[0].nFramesPerPacket; i++) {
                  for (int n = 0; n < dc.nChannelsInternal; n++) {
                     if (cs[n].LBRR_flags[i]) {
                        if (dc.nChannelsInternal == 2 && n == 0) {
                           sd_stereo_decode_pred(e, buf, MS_pred_Q13);
                           if (cs[1].LBRR_flags[i] == 0) decode_only_middle = k_ec_dec_icdf(e, buf, sk_stereo_only_code_mid_icdf, 8);
                        }
                        const int condCoding = (i > 0 && cs[n].LBRR_flags[i - 1]) ? SD_CODE_CONDITIONALLY : SD_CODE_INDEPENDENTLY;
                        sd_decode_indices(e, buf, &cs[n], i, 1, condCoding);
                        sd_decode_pulses(e, buf, S.pulses, cs[n].indices.signalType, cs[n].indices.quantOffsetType, cs[n].frame_length, S.tmp);
                     }
                  }
               }
            }
         }
         if (dc.nChannelsInternal == 2) {
            if (lostFlag == SD_FLAG_DECODE_NORMAL || (lostFlag == SD_FLAG_DECODE_LBRR && cs[0].LBRR_flags[cs[0].nFramesDecoded] == 1)) {
               sd_stereo_decode_pred(e, buf, MS_pred_Q13);
               if ((lostFlag == SD_FLAG_DECODE_NORMAL && cs[1].VAD_flags[cs[0].nFramesDecoded] == 0) ||
                   (lostFlag == SD_FLAG_DECODE_LBRR && cs[1].LBRR_flags[cs[0].nFramesDecoded] == 0)) decode_only_middle = k_ec_dec_icdf(e, buf, sk_stereo_only_code_mid_icdf, 8);
               else decode_only_middle = 0;
            } else { MS_pred_Q13[0] = sd->pred_prev_Q13[0]; MS_pred_Q13[1] = sd->pred_prev_Q13[1]; }
         }
         if (dc.nChannelsInternal == 2 && decode_only_middle == 0 && sd->prev_decode_only_middle == 1) {
            for (int i = 0; i < 480; i++) cs[1].outBuf[i] = 0;
            for (int i = 0; i < 16; i++) cs[1].sLPC_Q14_buf[i] = 0;
            cs[1].lagPrev = 100; cs[1].LastGainIndex = 10; cs[1].prevSignalType = SD_TYPE_NO_VOICE; cs[1].first_frame_after_reset = 1;
         }
         int has_side;
         if (lostFlag == SD_FLAG_DECODE_NORMAL) has_side = !decode_only_middle;
         else has_side = !sd->prev_decode_only_middle || (dc.nChannelsInternal == 2 && lostFlag == SD_FLAG_DECODE_LBRR && cs[1].LBRR_flags[cs[1].nFramesDecoded] == 1);
         shr[R_ONLYMID] = decode_only_middle; shr[R_PRED0] = MS_pred_Q13[0]; shr[R_PRED1] = MS_pred_Q13[1]; shr[R_HASSIDE] = has_side; shr[R_S2M] = stereo_to_mono;
      }
      shr[R_RET] = ret;
      ec_st(ecp, e);
   }
   if (wv_uni(shr[R_RET]) < 0) return wv_uni(shr[R_RET]);
   const int has_side = wv_uni(shr[R_HASSIDE]);
   const int nSamplesOutDec = wv_uni(cs[0].frame_length);
   for (int n = 0; n < dc.nChannelsInternal; n++) {
      WV_LDS OaSilkChannel *ch = &cs[n];
      if (n == 0 || has_side) {
         S.cng_exc = cng_exc + n * 320;
         LANE0 {
            EcCtx ec_; EcCtx *e = &ec_; ec_ld(e, ecp);
            const int FrameIndex = cs[0].nFramesDecoded - n;
            int condCoding;
            if (FrameIndex <= 0) condCoding = SD_CODE_INDEPENDENTLY;
            else if (lostFlag == SD_FLAG_DECODE_LBRR) condCoding = cs[n].LBRR_flags[FrameIndex - 1] ? SD_CODE_CONDITIONALLY : SD_CODE_INDEPENDENTLY;
            else if (n > 0 && sd->prev_decode_only_middle) condCoding = SD_CODE_INDEPENDENTLY_NO_LTP_SCALING;
            else condCoding = SD_CODE_CONDITIONALLY;
            shr[R_DEC] = sd_decode_frame_front(ch, e, buf, lostFlag, condCoding, S);
            ec_st(ecp, e);
         }
         const int decoded = wv_uni(shr[R_DEC]);
         if (decoded) sd_decode_core_wave(ch, S.ctrl, &A->xq[n][2], S);
         LANE0 sd_decode_frame_back(ch, &A->xq[n][2], decoded, S);
      } else { wv_sync(); for (int i = lane; i < nSamplesOutDec; i += WV_WIDTH) A->xq[n][2 + i] = 0; wv_sync(); }
      LANE0 ch->nFramesDecoded++;
   }
   LANE0 {
      i32 MS_pred_Q13[2] = { shr[R_PRED0], shr[R_PRED1] };
      if (dc.nChannelsAPI == 2 && dc.nChannelsInternal == 2) sd_stereo_ms_to_lr(sd, A->xq[0], A->xq[1], MS_pred_Q13, cs[0].fs_kHz, nSamplesOutDec);
      else { for (int i = 0; i < 2; i++) { A->xq[0][i] = sd->sMid[i]; sd->sMid[i] = A->xq[0][nSamplesOutDec + i]; } }
   }
   const int nOut = (nSamplesOutDec * dc.API_sampleRate) / (wv_uni(cs[0].fs_kHz) * 1000);
   const int nres = imin(dc.nChannelsAPI, dc.nChannelsInternal);
   for (int n = 0; n < nres; n++) sd_resample_wave(&cs[n], A, B, B->rs_out[n], &A->xq[n][1], nSamplesOutDec);
   if (dc.nChannelsAPI == 2 && dc.nChannelsInternal == 1) {
      if (wv_uni(shr[R_S2M])) sd_resample_wave(&cs[1], A, B, B->rs_out[1], &A->xq[0][1], nSamplesOutDec);
      else { wv_sync(); for (int i = lane; i < nOut; i += WV_WIDTH) B->rs_out[1][i] = B->rs_out[0][i]; }
   }
   LANE0 {
      if (lostFlag == SD_FLAG_PACKET_LOST) { for (int i = 0; i < sd->nChannelsInternal; i++) sd->ch[i].LastGainIndex = 10; }
      else sd->prev_decode_only_middle = shr[R_ONLYMID];
   }
   return nOut;
}
#endif

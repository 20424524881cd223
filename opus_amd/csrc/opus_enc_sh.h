/* opus_enc_sh.h — one (stream, frame) of the SILK-capable Opus encoder on one wavefront: HBM state -> LDS -> packet bytes -> HBM.
 *
 * Follows src/opus_encoder.c:1182 opus_encode_native (rate, channel, mode and bandwidth decisions :1310-1700) and :1855 opus_encode_frame_native
 * (high-pass :1958-1992, SILK control block :2024-2160, SILK call :2181, TOC / range finalisation :2310-2560, CBR padding :2646) for the SILK-only
 * mode (API rates 8-48 kHz, mono/stereo, 10-60 ms frames) and for hybrid and CELT-only frames at 48 kHz (:2402-2600, sh_hybrid_celt_wave), i.e. applications
 * VOIP, AUDIO and RESTRICTED_SILK; DTX (:1463, :2242, :2565, decide_dtx_mode :1115) and in-band FEC (decide_fec :940).  What is not built is refused loudly
 * (negative length, on the frame where it arises): mode and SILK-bandwidth transitions that need a CELT redundancy frame or a SILK prefill, CELT below
 * 48 kHz, frames above 60 ms (repacketised multi-frame packets). */
#ifndef OPUS_AMD_OPUS_ENC_SH_H
#define OPUS_AMD_OPUS_ENC_SH_H
#include "opus_sh_state.h"

#define OA_MODE_SILK_ONLY 1000
#define OA_MODE_HYBRID 1001
#define OA_MODE_CELT_ONLY 1002
#define OA_APP_VOIP 2048
#define OA_APP_AUDIO 2049
#define OA_APP_RESTRICTED_SILK 2052
#define OA_SIGNAL_VOICE 3001
#define OA_SIGNAL_MUSIC 3002
#define OA_ERR_UNIMPLEMENTED (-5)
#define OA_ERR_INTERNAL (-3)
#define OA_ERR_BUFFER_TOO_SMALL (-2)

struct ShShared {
   i32 frame_size, max_data_bytes, orig_max_data_bytes, pad_to, plc_frame, ret, err, toc, is_silence, sample_max;
   i32 bitrate_bps, equiv_rate, curr_bandwidth, activity, cutoff_Hz, use_hp_cutoff, bits_target, nBytes, silk_ret;
   i32 silk_bitRate, HB_gain, nb_compr_bytes, silk_signalType, silk_offset, stereo_width;
   i32 r[8];
};
struct ShLds {
   EcCtx ec;
   ShShared sh;
   OaShScalars st;
   OaShConfig cfg;
   u8 packet[OA_MAX_PACKET + 4];
   SilkEncLds S;                                         /* LAST (its own last member is the SILK state): mono batches allocate SH_LDS_BYTES(1) */
};
#define SH_LDS_BYTES(channels) (sizeof(ShLds) - ((channels) == 1 ? sizeof(OaSilkEncChannel) : 0))
/* per-stream HBM scratch: the high-passed input of the call followed by the rate-loop snapshots */
#define SH_SCRATCH_BYTES(frame_size, channels) (((size_t)(frame_size) * (channels) * 2 + 63) / 64 * 64 + sizeof(SeRateScratch))
#define SH_STAGE_SAMPLES 1920

WV_DEV i32 sh_equiv_rate(i32 bitrate, int channels, int frame_rate, int vbr, int mode, int complexity, int loss)      /* compute_equiv_rate :780 */
{
   i32 equiv = bitrate;
   if (frame_rate > 50) equiv -= (40 * channels + 20) * (frame_rate - 50);
   if (!vbr) equiv -= equiv / 12;
   equiv = equiv * (90 + complexity) / 100;
   if (mode == OA_MODE_SILK_ONLY || mode == OA_MODE_HYBRID) { if (complexity < 2) equiv = equiv * 4 / 5; equiv -= equiv * loss / (6 * loss + 10); }
   else if (mode == OA_MODE_CELT_ONLY) { if (complexity < 5) equiv = equiv * 9 / 10; }
   else equiv -= equiv * loss / (12 * loss + 20);
   return equiv;
}
WV_DEV u8 sh_gen_toc(int mode, int framerate, int bandwidth, int channels)                                            /* gen_toc :330 */
{
   int period = 0; u8 toc;
   while (framerate < 400) { framerate <<= 1; period++; }
   if (mode == OA_MODE_SILK_ONLY) toc = (u8)(((bandwidth - OA_BW_NB) << 5) | ((period - 2) << 3));
   else if (mode == OA_MODE_CELT_ONLY) { int tmp = bandwidth - OA_BW_MB; if (tmp < 0) tmp = 0; toc = (u8)(0x80 | (tmp << 5) | (period << 3)); }
   else toc = (u8)(0x60 | ((bandwidth - OA_BW_SWB) << 4) | ((period - 2) << 3));
   return (u8)(toc | ((channels == 2) << 2));
}

/* compute_stereo_width (src/opus_encoder.c:854): inter-channel correlation / loudness-difference tracker on the raw input; the three energy sums are plain
 * int32 sums of per-group terms (order-free) and are reduced over the wave, the smoothing recursion runs on lane 0.  Result in sh->stereo_width. */
WV_DEV void sh_compute_stereo_width_wave(WV_LDS ShLds *L, const i16 *pcm, int frame_size)
{
   WV_LDS OaShScalars *st = &L->st;
   const int shift = celt_ilog2(frame_size) - 2;
   i32 xx = 0, xy = 0, yy = 0;
   FOR_LANES(g, frame_size / 4) {
      i32 pxx = 0, pxy = 0, pyy = 0;
      for (int k = 0; k < 4; k++) { const i32 x = pcm[2 * (4 * g + k)], y = pcm[2 * (4 * g + k) + 1]; pxx += mult16_16(x, x) >> 2; pxy += mult16_16(x, y) >> 2; pyy += mult16_16(y, y) >> 2; }
      xx += pxx >> shift; xy += pxy >> shift; yy += pyy >> shift;
   }
   xx = wv_sum(xx); xy = wv_sum(xy); yy = wv_sum(yy);
   LANE0 {
      const int frame_rate = L->cfg.Fs / frame_size;
      const i16 short_alpha = (i16)(mult16_16(25, Q15ONE) / imax(50, frame_rate));
      st->wm_XX += mult16_32_q15(short_alpha, xx - st->wm_XX);
      st->wm_XY = mult16_32_q15(Q15ONE - short_alpha, st->wm_XY) + mult16_32_q15(short_alpha, xy);
      st->wm_YY += mult16_32_q15(short_alpha, yy - st->wm_YY);
      st->wm_XX = imax(0, st->wm_XX); st->wm_XY = imax(0, st->wm_XY); st->wm_YY = imax(0, st->wm_YY);
      if (imax(st->wm_XX, st->wm_YY) > QC16(8e-4f, 18)) {
         const i16 sqrt_xx = (i16)fx_sqrt(st->wm_XX), sqrt_yy = (i16)fx_sqrt(st->wm_YY), qrrt_xx = (i16)fx_sqrt(sqrt_xx), qrrt_yy = (i16)fx_sqrt(sqrt_yy);
         st->wm_XY = imin(st->wm_XY, sqrt_xx * sqrt_yy);
         const i16 corr = (i16)(fx_frac_div32(st->wm_XY, EPSILON + mult16_16(sqrt_xx, sqrt_yy)) >> 16);
         const i16 ldiff = (i16)(mult16_16(Q15ONE, iabs((i16)(qrrt_xx - qrrt_yy))) / (EPSILON + qrrt_xx + qrrt_yy));
         const i16 width = (i16)mult16_16_q15(imin(Q15ONE, fx_sqrt(QC32(1.f, 30) - mult16_16(corr, corr))), ldiff);
         st->wm_smoothed_width = (i16)(st->wm_smoothed_width + (width - st->wm_smoothed_width) / frame_rate);
         st->wm_max_follower = (i16)imax(st->wm_max_follower - QC16(.02f, 15) / frame_rate, st->wm_smoothed_width);
      }
      L->sh.stereo_width = (i16)imin(Q15ONE, mult16_16(20, st->wm_max_follower));
   }
}

/* lane 0: opus_encode_native's decisions (:1310-1700) */
WV_DEVN void sh_layer_decide(WV_LDS ShLds *L, int frame_size, int out_data_bytes)
{
   WV_LDS ShShared *sh = &L->sh; WV_LDS OaShScalars *st = &L->st; const WV_LDS OaShConfig *cfg = &L->cfg;
   const int Fs = cfg->Fs, channels = cfg->channels;
   i32 max_data_bytes = imin(1276 * 6, out_data_bytes);
   st->rangeFinal = 0;
   sh->plc_frame = 0; sh->ret = 0; sh->err = 0; sh->pad_to = 0; sh->frame_size = frame_size;
   if (max_data_bytes == 1 && Fs == frame_size * 10) { sh->err = OA_ERR_BUFFER_TOO_SMALL; return; }
   const i32 user = cfg->user_bitrate_bps == OA_AUTO ? 60 * Fs / frame_size + Fs * channels : (cfg->user_bitrate_bps == OA_BITRATE_MAX ? 1500000 : cfg->user_bitrate_bps);
   i32 bitrate_bps = imin(user, bits_to_bitrate(max_data_bytes * 8, Fs, frame_size));
   int frame_rate = Fs / frame_size;
   if (!cfg->use_vbr) {
      const i32 cbr_bytes = imin((bitrate_to_bits(bitrate_bps, Fs, frame_size) + 4) / 8, max_data_bytes);
      bitrate_bps = bits_to_bitrate(cbr_bytes * 8, Fs, frame_size);
      max_data_bytes = imax(1, cbr_bytes);
      sh->pad_to = max_data_bytes;
   }
   if (max_data_bytes < 3 || bitrate_bps < 3 * frame_rate * 8 || (frame_rate < 50 && (max_data_bytes * (i32)frame_rate < 300 || bitrate_bps < 2400))) {
      /* 'PLC' frame (:1345-1410) */
      int tocmode = st->mode, bw = st->bandwidth == 0 ? OA_BW_NB : st->bandwidth, packet_code = 0, num_multiframes = 0;
      if (tocmode == 0) tocmode = OA_MODE_SILK_ONLY;
      if (frame_rate > 100) tocmode = OA_MODE_CELT_ONLY;
      if (frame_rate == 25 && tocmode != OA_MODE_SILK_ONLY) { frame_rate = 50; packet_code = 1; }
      if (frame_rate <= 16) {
         if (out_data_bytes == 1 || (tocmode == OA_MODE_SILK_ONLY && frame_rate != 10)) { tocmode = OA_MODE_SILK_ONLY; packet_code = frame_rate <= 12; frame_rate = frame_rate == 12 ? 25 : 16; }
         else { num_multiframes = 50 / frame_rate; frame_rate = 50; packet_code = 3; }
      }
      if (tocmode == OA_MODE_SILK_ONLY && bw > OA_BW_WB) bw = OA_BW_WB;
      else if (tocmode == OA_MODE_CELT_ONLY && bw == OA_BW_MB) bw = OA_BW_NB;
      else if (tocmode == OA_MODE_HYBRID && bw <= OA_BW_SWB) bw = OA_BW_SWB;
      L->packet[0] = (u8)(sh_gen_toc(tocmode, frame_rate, bw, st->stream_channels) | packet_code);
      if (packet_code == 3) L->packet[1] = (u8)num_multiframes;
      sh->plc_frame = 1; sh->ret = packet_code <= 1 ? 1 : 2;
      return;
   }
   const i32 max_rate = bits_to_bitrate(max_data_bytes * 8, Fs, frame_size);
   const int loss = cfg->packet_loss_perc;
   i32 equiv_rate = sh_equiv_rate(bitrate_bps, channels, frame_rate, cfg->use_vbr, 0, cfg->complexity, loss);
   int voice_est;
   if (cfg->signal_type == OA_SIGNAL_VOICE) voice_est = 127; else if (cfg->signal_type == OA_SIGNAL_MUSIC) voice_est = 0;
   else if (cfg->application == OA_APP_VOIP) voice_est = 115; else voice_est = 48;                 /* voice_ratio is -1 without the float analysis */
   if (cfg->force_channels != OA_AUTO && channels == 2) st->stream_channels = cfg->force_channels;
   else if (channels == 2) {
      i32 thr = 17000 + ((voice_est * voice_est * (19000 - 17000)) >> 14);
      if (st->stream_channels == 2) thr -= 1000; else thr += 1000;
      st->stream_channels = equiv_rate > thr ? 2 : 1;
   } else st->stream_channels = channels;
   equiv_rate = sh_equiv_rate(bitrate_bps, st->stream_channels, frame_rate, cfg->use_vbr, 0, cfg->complexity, loss);
   st->sm_useDTX = cfg->use_dtx && !sh->is_silence;                                               /* :1463: SILK's own DTX; digital silence takes the generalised one */
   /* mode (:1487-1560) */
   if (cfg->application == OA_APP_RESTRICTED_SILK) st->mode = OA_MODE_SILK_ONLY;
   else if (cfg->user_forced_mode == OA_AUTO) {
      const i32 stereo_width = sh->stereo_width;
      const i32 mode_voice = (i32)(mult16_32_q15(Q15ONE - stereo_width, 64000) + mult16_32_q15(stereo_width, 44000));
      const i32 mode_music = (i32)(mult16_32_q15(Q15ONE - stereo_width, 10000) + mult16_32_q15(stereo_width, 10000));
      i32 threshold = mode_music + ((voice_est * voice_est * (mode_voice - mode_music)) >> 14);
      if (cfg->application == OA_APP_VOIP) threshold += 8000;
      if (st->prev_mode == OA_MODE_CELT_ONLY) threshold -= 4000; else if (st->prev_mode > 0) threshold += 4000;
      st->mode = equiv_rate >= threshold ? OA_MODE_CELT_ONLY : OA_MODE_SILK_ONLY;
      if (cfg->use_inband_fec && loss > ((128 - voice_est) >> 4) && (cfg->use_inband_fec != 2 || voice_est > 25)) st->mode = OA_MODE_SILK_ONLY;   /* :1517 */
      if (st->sm_useDTX && voice_est > 100) st->mode = OA_MODE_SILK_ONLY;                          /* :1521 */
      if (max_data_bytes < bitrate_to_bits(frame_rate > 50 ? 9000 : 6000, Fs, frame_size) / 8) st->mode = OA_MODE_CELT_ONLY;
   } else st->mode = cfg->user_forced_mode;
   if (st->mode != OA_MODE_CELT_ONLY && frame_size < Fs / 100) st->mode = OA_MODE_CELT_ONLY;
   /* a SILK/hybrid <-> CELT-only switch needs a redundant CELT frame and a SILK prefill (:1568-1590, :2478-2590): not built; CELT below 48 kHz neither */
   if ((st->mode == OA_MODE_CELT_ONLY) != (st->prev_mode == OA_MODE_CELT_ONLY) && st->prev_mode > 0) { sh->err = OA_ERR_UNIMPLEMENTED; return; }
   if (st->mode == OA_MODE_CELT_ONLY && (Fs != 48000 || frame_size > Fs / 50)) { sh->err = OA_ERR_UNIMPLEMENTED; return; }
   if (st->stream_channels == 1 && st->prev_channels == 2 && st->sm_toMono == 0) { st->sm_toMono = 1; st->stream_channels = 2; } else st->sm_toMono = 0;
   equiv_rate = sh_equiv_rate(bitrate_bps, st->stream_channels, frame_rate, cfg->use_vbr, st->mode, cfg->complexity, loss);
   /* bandwidth (:1600-1700) */
   if (st->mode == OA_MODE_CELT_ONLY || st->first || st->sm_allowBandwidthSwitch) {
      const i32 voice_bw[8] = {9000, 700, 9000, 700, 13500, 1000, 14000, 2000}, music_bw[8] = {9000, 700, 9000, 700, 11000, 1000, 12000, 2000};
      int bandwidth = OA_BW_FB;
      do {
         const int k = 2 * (bandwidth - OA_BW_MB);
         int threshold = music_bw[k] + ((voice_est * voice_est * (voice_bw[k] - music_bw[k])) >> 14);
         const int hysteresis = music_bw[k + 1] + ((voice_est * voice_est * (voice_bw[k + 1] - music_bw[k + 1])) >> 14);
         if (!st->first) { if (st->auto_bandwidth >= bandwidth) threshold -= hysteresis; else threshold += hysteresis; }
         if (equiv_rate >= threshold) break;
      } while (--bandwidth > OA_BW_NB);
      if (bandwidth == OA_BW_MB) bandwidth = OA_BW_WB;
      st->bandwidth = st->auto_bandwidth = bandwidth;
      if (!st->first && st->mode != OA_MODE_CELT_ONLY && !st->sm_inWBmodeWithoutVariableLP && st->bandwidth > OA_BW_WB) st->bandwidth = OA_BW_WB;
   }
   if (st->bandwidth > cfg->max_bandwidth) st->bandwidth = cfg->max_bandwidth;
   if (cfg->user_bandwidth != OA_AUTO) st->bandwidth = cfg->user_bandwidth;
   if (st->mode != OA_MODE_CELT_ONLY && max_rate < 15000) st->bandwidth = imin(st->bandwidth, OA_BW_WB);
   if (Fs <= 24000 && st->bandwidth > OA_BW_SWB) st->bandwidth = OA_BW_SWB;
   if (Fs <= 16000 && st->bandwidth > OA_BW_WB) st->bandwidth = OA_BW_WB;
   if (Fs <= 12000 && st->bandwidth > OA_BW_MB) st->bandwidth = OA_BW_MB;
   if (Fs <= 8000 && st->bandwidth > OA_BW_NB) st->bandwidth = OA_BW_NB;
   {  /* decide_fec (:940): enough rate for the LBRR side stream at this bandwidth?  With > 5 % loss the bandwidth comes down until there is. */
      int fec = 0;
      if (cfg->use_inband_fec && loss != 0 && st->mode != OA_MODE_CELT_ONLY) {
         const i32 thr_tab[10] = {12000, 1000, 14000, 1000, 16000, 1000, 20000, 1000, 22000, 1000};
         const int orig_bandwidth = st->bandwidth;
         for (;;) {
            i32 thr = thr_tab[2 * (st->bandwidth - OA_BW_NB)]; const i32 hyst = thr_tab[2 * (st->bandwidth - OA_BW_NB) + 1];
            if (st->sm_LBRR_coded == 1) thr -= hyst;
            if (st->sm_LBRR_coded == 0) thr += hyst;
            thr = sk_mulwb(thr * (125 - imin(loss, 25)), SE_FIX(0.01, 16));
            if (equiv_rate > thr) { fec = 1; break; }
            else if (loss <= 5) break;
            else if (st->bandwidth > OA_BW_NB) st->bandwidth--;
            else { st->bandwidth = orig_bandwidth; break; }
         }
      }
      st->sm_LBRR_coded = fec;
   }
   if (st->mode == OA_MODE_CELT_ONLY && st->bandwidth == OA_BW_MB) st->bandwidth = OA_BW_WB;
   int curr_bandwidth = st->bandwidth;
   if (cfg->application == OA_APP_RESTRICTED_SILK && curr_bandwidth > OA_BW_WB) st->bandwidth = curr_bandwidth = OA_BW_WB;
   if (st->mode == OA_MODE_SILK_ONLY && curr_bandwidth > OA_BW_WB) st->mode = OA_MODE_HYBRID;
   if (st->mode == OA_MODE_HYBRID && curr_bandwidth <= OA_BW_WB) st->mode = OA_MODE_SILK_ONLY;
   if (st->mode == OA_MODE_HYBRID && (Fs != 48000 || frame_size > Fs / 50 || (st->prev_mode > 0 && st->prev_mode != OA_MODE_HYBRID))) { sh->err = OA_ERR_UNIMPLEMENTED; return; }   /* CELT layer: 48 kHz, <= 20 ms; SILK -> hybrid needs the CELT prefill */
   if (frame_size > 3 * Fs / 50) { sh->err = OA_ERR_UNIMPLEMENTED; return; }                       /* 80/100/120 ms: repacketised multi-frame packets */
   if (st->silk_bw_switch) { sh->err = OA_ERR_UNIMPLEMENTED; return; }                             /* bandwidth switch with CELT redundancy */
   sh->bitrate_bps = bitrate_bps; sh->equiv_rate = equiv_rate; sh->curr_bandwidth = curr_bandwidth;
   sh->orig_max_data_bytes = max_data_bytes; sh->max_data_bytes = imin(max_data_bytes, 1276);
   /* opus_encode_frame_native prologue */
   sh->activity = sh->is_silence ? 0 : SE_VAD_NO_DECISION;
   sh->bits_target = imin(8 * sh->max_data_bytes, bitrate_to_bits(bitrate_bps, Fs, frame_size)) - 8;
   const i32 hp_freq_smth1 = st->mode == OA_MODE_CELT_ONLY ? shl32(se_lin2log(60), 8) : L->S.st.ch[0].variable_HP_smth1_Q15;
   st->variable_HP_smth2_Q15 = sk_mlawb(st->variable_HP_smth2_Q15, hp_freq_smth1 - st->variable_HP_smth2_Q15, SE_FIX(0.015f, 16));
   sh->cutoff_Hz = se_log2lin(st->variable_HP_smth2_Q15 >> 8);
   sh->use_hp_cutoff = cfg->application == OA_APP_VOIP;
}

/* hp_cutoff (:441, VOIP) or dc_reject (:479) over one chunk staged in LDS: lane c runs channel c's recursion */
WV_DEV void sh_highpass_chunk(WV_LDS ShLds *L, WV_LDS i16 *io, int len, int channels, const i32 *B_Q28, const i32 *A_Q28)
{
   const int c = wv_lane();
   if (c < channels) {
      if (L->sh.use_hp_cutoff) {
         const i32 A0_L = (-A_Q28[0]) & 0x3FFF, A0_U = (-A_Q28[0]) >> 14, A1_L = (-A_Q28[1]) & 0x3FFF, A1_U = (-A_Q28[1]) >> 14;
         i32 S0 = L->st.hp_mem[2 * c], S1 = L->st.hp_mem[2 * c + 1];
         for (int k = 0; k < len; k++) {
            const i32 inval = io[k * channels + c];
            const i32 out32_Q14 = shl32(sk_mlawb(S0, B_Q28[0], inval), 2);
            S0 = S1 + sk_rround(sk_mulwb(out32_Q14, A0_L), 14); S0 = sk_mlawb(S0, out32_Q14, A0_U); S0 = sk_mlawb(S0, B_Q28[1], inval);
            S1 = sk_rround(sk_mulwb(out32_Q14, A1_L), 14); S1 = sk_mlawb(S1, out32_Q14, A1_U); S1 = sk_mlawb(S1, B_Q28[2], inval);
            io[k * channels + c] = (i16)sk_sat16((out32_Q14 + (1 << 14) - 1) >> 14);
         }
         L->st.hp_mem[2 * c] = S0; L->st.hp_mem[2 * c + 1] = S1;
      } else {
         const int shift = celt_ilog2(L->cfg.Fs / (3 * 4));
         i32 mem = L->st.hp_mem[2 * c];
         for (int k = 0; k < len; k++) {
            const i32 x = shl32(saturate((i32)io[k * channels + c], (1 << 16) - 1), 14), y = x - mem;
            mem = mem + pshr32(y, shift);
            io[k * channels + c] = (i16)saturate(pshr32(y, 14), 32767);
         }
         L->st.hp_mem[2 * c] = mem;
      }
   }
}

/* packet bytes (nbytes at pk, pk[0] = TOC) -> out, or the same frame re-framed as a code-3 packet padded to pad_to bytes (opus_packet_pad, src/repacketizer.c:346) */
WV_DEV int sh_emit_packet(const WV_LDS u8 *pk, u8 *out, int nbytes, int pad_to, int out_cap)
{
   if (nbytes <= 0) return nbytes;
   if (pad_to == 0 || nbytes == pad_to) { if (nbytes > out_cap) return OA_ERR_BUFFER_TOO_SMALL; FOR_LANES(i, nbytes) out[i] = pk[i]; return nbytes; }
   if (nbytes > pad_to) return OA_ERR_INTERNAL;
   if (pad_to > out_cap) return OA_ERR_BUFFER_TOO_SMALL;
   const int L0 = nbytes - 1, pad_amount = pad_to - (L0 + 2);
   const int nb_255s = pad_amount > 0 ? (pad_amount - 1) / 255 : 0, hdr = 2 + (pad_amount > 0 ? nb_255s + 1 : 0);
   FOR_LANES(i, pad_to) {
      u8 v = 0;
      if (i == 0) v = (u8)((pk[0] & 0xFC) | 0x3);
      else if (i == 1) v = (u8)(1 | (pad_amount != 0 ? 0x40 : 0));
      else if (i < hdr) v = i < hdr - 1 ? 255 : (u8)(pad_amount - 255 * nb_255s - 1);
      else if (i < hdr + L0) v = pk[1 + i - hdr];
      out[i] = v;
   }
   return pad_to;
}

/* compute_silk_rate_for_hybrid (src/opus_encoder.c:656) */
WV_DEV i32 sh_silk_rate_for_hybrid(i32 rate, int bandwidth, int frame20ms, int vbr, int fec, int channels)
{
   const i32 rate_table[7][5] = {{0, 0, 0, 0, 0}, {12000, 10000, 10000, 11000, 11000}, {16000, 13500, 13500, 15000, 15000}, {20000, 16000, 16000, 18000, 18000},
                                 {24000, 18000, 18000, 21000, 21000}, {32000, 22000, 22000, 28000, 28000}, {64000, 38000, 38000, 50000, 50000}};
   rate /= channels;
   const int entry = 1 + frame20ms + 2 * fec, N = 7;
   int i; i32 silk_rate;
   for (i = 1; i < N; i++) if (rate_table[i][0] > rate) break;
   if (i == N) { silk_rate = rate_table[i - 1][entry]; silk_rate += (rate - rate_table[i - 1][0]) / 2; }
   else { const i32 lo = rate_table[i - 1][entry], hi = rate_table[i][entry], x0 = rate_table[i - 1][0], x1 = rate_table[i][0]; silk_rate = (lo * (x1 - rate) + hi * (rate - x0)) / (x1 - x0); }
   if (!vbr) silk_rate += 100;
   if (bandwidth == OA_BW_SWB) silk_rate += 300;
   silk_rate *= channels;
   if (channels == 2 && rate >= 12000) silk_rate -= 1000;
   return silk_rate;
}

/* stereo-width decision and the fade gains of the CELT input (:2365-2400), then the per-call bookkeeping (:2596-2600); silk_width = the width SILK reported */
/* the generalised DTX decision at the end of opus_encode_frame_native (:2565-2576, decide_dtx_mode :1115): without the float analysis it only ever sees
 * digital silence (SILK's own DTX covers the rest); after 200 ms of it the packet is the TOC byte alone, at most 400 ms in a row */
WV_DEV int sh_generalised_dtx_l0(WV_LDS ShLds *L, int frame_size, int Fs)
{
   WV_LDS ShShared *sh = &L->sh; WV_LDS OaShScalars *st = &L->st;
   if (L->cfg.use_dtx && !st->sm_useDTX) {
      int dtx = 0;
      if (!sh->activity) {
         st->nb_no_activity_ms_Q1 += 2 * 1000 * frame_size / Fs;
         if (st->nb_no_activity_ms_Q1 > 10 * 20 * 2) { if (st->nb_no_activity_ms_Q1 <= (10 + 20) * 20 * 2) dtx = 1; else st->nb_no_activity_ms_Q1 = 10 * 20 * 2; }
      } else st->nb_no_activity_ms_Q1 = 0;
      return dtx;
   }
   st->nb_no_activity_ms_Q1 = 0;
   return 0;
}
WV_DEV void sh_width_and_bookkeeping_l0(WV_LDS ShLds *L, int frame_size, i32 silk_width)
{
   WV_LDS ShShared *sh = &L->sh; WV_LDS OaShScalars *st = &L->st;
   st->sm_stereoWidth_Q14 = silk_width;
   if (st->mode != OA_MODE_HYBRID || st->stream_channels == 1) {
      if (sh->equiv_rate > 32000) st->sm_stereoWidth_Q14 = 16384; else if (sh->equiv_rate < 16000) st->sm_stereoWidth_Q14 = 0;
      else st->sm_stereoWidth_Q14 = 16384 - 2048 * (i32)(32000 - sh->equiv_rate) / (sh->equiv_rate - 14000);
   }
   sh->r[0] = 0;
   if (L->cfg.channels == 2 && (st->hybrid_stereo_width_Q14 < (1 << 14) || st->sm_stereoWidth_Q14 < (1 << 14))) {
      i16 g1 = (i16)st->hybrid_stereo_width_Q14, g2 = (i16)st->sm_stereoWidth_Q14;
      sh->r[0] = 1; sh->r[1] = g1 == 16384 ? Q15ONE : shl16(g1, 1); sh->r[2] = g2 == 16384 ? Q15ONE : shl16(g2, 1);
      st->hybrid_stereo_width_Q14 = st->sm_stereoWidth_Q14;
   }
   st->prev_mode = st->mode; st->prev_channels = st->stream_channels; st->prev_framesize = frame_size; st->first = 0;
}

/* The CELT layer of a hybrid frame (src/opus_encoder.c:2452-2600): bands 17.. on the coder the SILK layer leaves behind.  The SILK state has served its
 * purpose: it goes back to HBM and the CELT encoder's LDS working set takes its place.  CELT input = the delay-compensated high-passed signal
 * (delay_buffer tail + this call's frame), faded towards the high-band gain and the stereo width decided above. */
WV_DEVN void sh_hybrid_celt_wave(WV_LDS ShLds *L, OaShStream *gs, const i16 *pcm_hp, int frame_size, u8 *out, int out_cap, i32 *len_out, u32 *rng_out)
{
   const int hyb = L->st.mode == OA_MODE_HYBRID;                                       /* else CELT-only inside an AUDIO / VOIP encoder (:2452-2560 with start band 0) */
   WV_LDS ShShared *sh = &L->sh; WV_LDS OaShScalars *st = &L->st;
   const int CC = L->cfg.channels, Fs = L->cfg.Fs;
   if (hyb) {  /* SILK state back to HBM (coalesced); a CELT-only frame leaves it untouched */
      i32 *g = (i32 *)&gs->silk; const WV_LDS i32 *d = (const WV_LDS i32 *)&L->S.st;
      FOR_LANES(i, SE_STATE_WORDS(CC)) g[i] = d[i];
   }
   wv_sync();
   WV_LDS FrameLds *F = (WV_LDS FrameLds *)&L->S;
   WV_LDS FrameShared *fs = &F->sh;
   {
      const i32 *g = (const i32 *)&gs->celt.s; WV_LDS i32 *d = (WV_LDS i32 *)&F->st;
      FOR_LANES(i, (int)(sizeof(OaEncScalars) / 4)) d[i] = g[i];
      FOR_LANES(i, 2 * NBE) { F->oldBandE[i] = gs->celt.oldBandE[i]; F->energyError[i] = gs->celt.energyError[i]; }
      FOR_LANES(i, (OA_MAX_PACKET + 4) / 4) ((WV_LDS i32 *)F->packet)[i] = ((const WV_LDS i32 *)L->packet)[i];
   }
   LANE0 {
      if (hyb) ec_cp_lds(&F->ec, &L->ec);
      const int curr_bandwidth = sh->curr_bandwidth, endband = curr_bandwidth == OA_BW_NB ? 13 : curr_bandwidth <= OA_BW_WB ? 17 : curr_bandwidth == OA_BW_SWB ? 19 : 21;
      fs->CC = CC; fs->C = st->stream_channels; fs->frame_size = frame_size; fs->start = hyb ? 17 : 0; fs->end = endband; fs->effEnd = endband;
      fs->complexity = L->cfg.complexity; fs->lsb_depth = imin(16, L->cfg.lsb_depth); fs->disable_inv = L->cfg.disable_inv; fs->disable_pf = 0; fs->force_intra = 0; fs->loss_rate = L->cfg.packet_loss_perc;
      fs->vbr = L->cfg.use_vbr; fs->constrained_vbr = hyb ? 0 : L->cfg.vbr_constraint;
      fs->bitrate = -1;
      if (L->cfg.use_vbr) { const i32 cb = hyb ? sh->bitrate_bps - sh->silk_bitRate : sh->bitrate_bps; if (cb > 500) fs->bitrate = imin(cb, 750000 * CC); }     /* OPUS_SET_BITRATE rejects <= 500 and keeps OPUS_BITRATE_MAX */
      fs->curr_bandwidth = curr_bandwidth; fs->max_data_bytes = sh->max_data_bytes; fs->orig_max_data_bytes = sh->orig_max_data_bytes; fs->pad_to = sh->pad_to;
      fs->plc_frame = 0; fs->ret = 0; fs->skip_celt = 0;
      fs->toc = sh_gen_toc(st->mode, Fs / frame_size, curr_bandwidth, st->stream_channels);
      fs->silk_signalType = sh->silk_signalType; fs->silk_offset = sh->silk_offset;
      fs->do_stereo_fade = sh->r[0]; fs->fade_g1 = sh->r[1]; fs->fade_g2 = sh->r[2];
   }
   /* ---- CELT input: [delay tail | new frame], then the delay line moves on (:1950, :2340-2353) ---- */
   const int total_buffer = Fs / 250, encoder_buffer = Fs / 100;
   {
      WV_LDS i16 *io = F->A.pcm16;
      FOR_LANES(i, frame_size * CC) { const int n = i / CC, c = i - n * CC; io[i] = n < total_buffer ? gs->delay_buffer[(encoder_buffer - total_buffer + n) * CC + c] : pcm_hp[(n - total_buffer) * CC + c]; }
      wv_sync();
      /* new delay line = the last encoder_buffer samples of [old line | this frame]; ascending in place through registers, one 64-lane trip at a time */
      for (int b0 = 0; b0 < encoder_buffer * CC; b0 += WV_WIDTH) {
         const int i = b0 + wv_lane(), j = i + frame_size * CC;
         i16 v = 0;
         if (i < encoder_buffer * CC) v = j < encoder_buffer * CC ? gs->delay_buffer[j] : pcm_hp[j - encoder_buffer * CC];
         wv_sync();
         if (i < encoder_buffer * CC) gs->delay_buffer[i] = v;
         wv_sync();
      }
      const i16 g1 = (i16)st->prev_HB_gain, g2 = (i16)sh->HB_gain;
      if (g1 < Q15ONE || g2 < Q15ONE) {                                               /* gain_fade (:581) */
         FOR_LANES(i, frame_size * CC) {
            const int n = i / CC; i16 g = g2;
            if (n < OA_OVERLAP) { i16 w = ct_window[n]; w = (i16)mult16_16_q15(w, w); g = (i16)(mac16_16(mult16_16(w, g2), Q15ONE - w, g1) >> 15); }
            io[i] = (i16)mult16_16_q15(g, io[i]);
         }
      }
      wv_sync();
      LANE0 st->prev_HB_gain = sh->HB_gain;
      if (fs->do_stereo_fade) { stereo_fade_lanes(F, frame_size); wv_sync(); }
      const int overlap = OA_OVERLAP;
      i32 a = 0, b = 0;
      FOR_LANES(i, CC * (frame_size - overlap)) a = imax(a, iabs((i32)io[i]));
      FOR_LANES(i, CC * overlap) b = imax(b, iabs((i32)io[CC * (frame_size - overlap) + i]));
      a = wv_max(a); b = wv_max(b);
      LANE0 { fs->r[0] = a; fs->r[1] = b; }
   }
   wv_sync();
   LANE0 celt_prologue(F, hyb ? sh->nb_compr_bytes : 0);
#ifdef SH_DEBUG
   LANE0 printf("hyb: toc %d pk0 %d nb_compr %d tell %d skip %d bitrate %d vbr %d C %d end %d\n", fs->toc, F->packet[0], sh->nb_compr_bytes, fs->tell, fs->skip_celt, fs->bitrate, fs->vbr, fs->C, fs->end);
#endif
   wv_sync();
   if (fs->skip_celt) { LANE0 { F->packet[0] = (u8)fs->toc; F->packet[1] = 0; F->st.rangeFinal = 0; fs->ret = 2; } wv_sync(); }
   else if (hyb) celt_encode_core<true>(F, &gs->celt, out);
   else celt_encode_core<false>(F, &gs->celt, out);
   wv_sync();
#ifdef SH_DEBUG
   LANE0 printf("hyb end: toc %d pk0 %d ret %d rng %u\n", fs->toc, F->packet[0], fs->ret, F->st.rangeFinal);
#endif
   {
      LANE0 { if (fs->ret >= 0 && sh_generalised_dtx_l0(L, frame_size, Fs)) { F->st.rangeFinal = 0; F->packet[0] = (u8)fs->toc; fs->ret = 1; fs->pad_to = 0; } }
      const int nbytes = emit_packet_wave(F, out, fs->ret, fs->pad_to, out_cap);
      LANE0 { *len_out = nbytes; *rng_out = F->st.rangeFinal; st->rangeFinal = F->st.rangeFinal; }
      wv_sync();
      i32 *g = (i32 *)&gs->celt.s; const WV_LDS i32 *d = (const WV_LDS i32 *)&F->st;
      FOR_LANES(i, (int)(sizeof(OaEncScalars) / 4)) g[i] = d[i];
      g = (i32 *)&gs->s; d = (const WV_LDS i32 *)st;
      FOR_LANES(i, (int)(sizeof(OaShScalars) / 4)) g[i] = d[i];
   }
}

WV_DEV void oa_sh_encode_frame(WV_LDS ShLds *L, OaShStream *gs, const i16 *pcm, int frame_size, int max_data_bytes, u8 *out, int out_cap, i16 *pcm_hp, SeRateScratch *G, i32 *len_out, u32 *rng_out)
{
   WV_LDS ShShared *sh = &L->sh; WV_LDS OaShScalars *st = &L->st;
   SE_PHASE_START(&L->S);
   /* ---- load configuration, Opus-layer scalars and the SILK encoder state (coalesced) ---- */
   {
      const i32 *g = (const i32 *)&gs->cfg; WV_LDS i32 *d = (WV_LDS i32 *)&L->cfg;
      FOR_LANES(i, (int)(sizeof(OaShConfig) / 4)) d[i] = g[i];
      g = (const i32 *)&gs->s; d = (WV_LDS i32 *)st;
      FOR_LANES(i, (int)(sizeof(OaShScalars) / 4)) d[i] = g[i];
      g = (const i32 *)&gs->silk; d = (WV_LDS i32 *)&L->S.st;
      FOR_LANES(i, SE_STATE_WORDS(gs->cfg.channels)) d[i] = g[i];
   }
   wv_sync();
   SE_PHASE(&L->S, 0);
   const int CC = L->cfg.channels, Fs = L->cfg.Fs;
   {  /* is_digital_silence (:1060, fixed point: all samples zero) */
      i32 m = 0;
      FOR_LANES(i, frame_size * CC) m = imax(m, iabs((i32)pcm[i]));
      m = wv_max(m);
      LANE0 { sh->sample_max = m; sh->is_silence = m == 0; }
   }
   if (CC == 2 && L->cfg.force_channels != 1) sh_compute_stereo_width_wave(L, pcm, frame_size); else { LANE0 sh->stereo_width = 0; }
   LANE0 sh_layer_decide(L, frame_size, max_data_bytes);
   if (sh->err) { LANE0 { *len_out = sh->err; *rng_out = 0; gs->s.error = sh->err; } return; }
   if (sh->plc_frame) {
      const int n = sh_emit_packet(L->packet, out, sh->ret, sh->pad_to, out_cap);
      LANE0 { *len_out = n; *rng_out = 0; gs->s.rangeFinal = 0; }
      return;
   }
   /* ---- high-pass into the per-stream HBM scratch, staged through LDS in chunks ---- */
   {
      i32 B_Q28[3] = {0, 0, 0}, A_Q28[2] = {0, 0};
      if (sh->use_hp_cutoff) {
         const i32 Fc_Q19 = sk_mulbb(SE_FIX(1.5 * 3.14159 / 1000, 19), sh->cutoff_Hz) / (Fs / 1000);
         const i32 r_Q28 = SE_FIX(1.0, 28) - SE_FIX(0.92, 9) * Fc_Q19;
         B_Q28[0] = r_Q28; B_Q28[1] = shl32(-r_Q28, 1); B_Q28[2] = r_Q28;
         const i32 r_Q22 = r_Q28 >> 6;
         A_Q28[0] = sk_mulww(r_Q22, sk_mulww(Fc_Q19, Fc_Q19) - SE_FIX(2.0, 22));
         A_Q28[1] = sk_mulww(r_Q22, r_Q22);
      }
      WV_LDS i16 *stage = L->S.u.pcm_stage;
      const int chunk = SH_STAGE_SAMPLES / CC;
      for (int i0 = 0; i0 < frame_size; i0 += chunk) {
         const int n = imin(chunk, frame_size - i0);
         wv_sync();
         FOR_LANES(i, n * CC) stage[i] = pcm[i0 * CC + i];
         wv_sync();
         sh_highpass_chunk(L, stage, n, CC, B_Q28, A_Q28);
         wv_sync();
         FOR_LANES(i, n * CC) pcm_hp[i0 * CC + i] = stage[i];
      }
      wv_sync();
   }
   SE_PHASE(&L->S, 1);
   if (st->mode == OA_MODE_CELT_ONLY) {
      LANE0 { sh->HB_gain = Q15ONE; sh_width_and_bookkeeping_l0(L, frame_size, 0); }
      sh_hybrid_celt_wave(L, gs, pcm_hp, frame_size, out, out_cap, len_out, rng_out);
      return;
   }
   /* ---- SILK (:2024-2200) ---- */
   SeControl sc;
   {
      const int mode = st->mode, curr_bandwidth = sh->curr_bandwidth, frame_rate = Fs / frame_size;
      sc.nChannelsAPI = CC; sc.nChannelsInternal = st->stream_channels; sc.API_sampleRate = Fs;
      const i32 total_bitRate = bits_to_bitrate(sh->bits_target, Fs, frame_size);
      sc.bitRate = total_bitRate;
      sh->HB_gain = Q15ONE;
      if (mode == OA_MODE_HYBRID) {                                                  /* :2034-2046 */
         sc.bitRate = sh_silk_rate_for_hybrid(total_bitRate, curr_bandwidth, Fs == 50 * frame_size, L->cfg.use_vbr, st->sm_LBRR_coded, st->stream_channels);
         sh->HB_gain = Q15ONE - (fx_exp2((i16)(-(total_bitRate - sc.bitRate))) >> 1);        /* celt_exp2 takes an opus_val16: the rate difference is truncated as in the reference */
      }
      sh->silk_bitRate = sc.bitRate;
      sc.payloadSize_ms = 1000 * frame_size / Fs;
      sc.desiredInternalSampleRate = curr_bandwidth == OA_BW_NB ? 8000 : curr_bandwidth == OA_BW_MB ? 12000 : 16000;
      sc.minInternalSampleRate = mode == OA_MODE_HYBRID ? 16000 : 8000;
      sc.maxInternalSampleRate = 16000;
      if (mode == OA_MODE_SILK_ONLY) {
         i32 effective_max_rate = bits_to_bitrate(sh->max_data_bytes * 8, Fs, frame_size);
         if (frame_rate > 50) effective_max_rate = effective_max_rate * 2 / 3;
         if (effective_max_rate < 8000) { sc.maxInternalSampleRate = 12000; sc.desiredInternalSampleRate = imin(12000, sc.desiredInternalSampleRate); }
         if (effective_max_rate < 7000) { sc.maxInternalSampleRate = 8000; sc.desiredInternalSampleRate = imin(8000, sc.desiredInternalSampleRate); }
      }
      sc.packetLossPercentage = L->cfg.packet_loss_perc; sc.complexity = L->cfg.complexity; sc.useInBandFEC = L->cfg.use_inband_fec; sc.LBRR_coded = st->sm_LBRR_coded; sc.useDTX = st->sm_useDTX;
      sc.useCBR = !L->cfg.use_vbr;
      sc.maxBits = (sh->max_data_bytes - 1) * 8;
      if (mode == OA_MODE_HYBRID) {                                                  /* :2136-2160 */
         if (sc.useCBR) { const i16 other_bits = (i16)imax(0, sc.maxBits - sc.bitRate * frame_size / Fs); sc.maxBits = imax(0, sc.maxBits - other_bits * 3 / 4); sc.useCBR = 0; }
         else { const i32 maxBitRate = sh_silk_rate_for_hybrid(sc.maxBits * Fs / frame_size, curr_bandwidth, Fs == 50 * frame_size, L->cfg.use_vbr, st->sm_LBRR_coded, st->stream_channels); sc.maxBits = bitrate_to_bits(maxBitRate, Fs, frame_size); }
      }
      sc.toMono = st->sm_toMono; sc.opusCanSwitch = st->sm_opusCanSwitch; sc.reducedDependency = 0;
      sc.internalSampleRate = 0; sc.allowBandwidthSwitch = 0; sc.inWBmodeWithoutVariableLP = 0; sc.stereoWidth_Q14 = 0; sc.switchReady = 0; sc.signalType = 0; sc.offset = 0;
   }
   LANE0 { EcCtx e_; EcCtx *e = &e_; WV_LDS u8 *buf = L->packet + 1; k_ec_enc_init(EC_PASS, (u32)(sh->orig_max_data_bytes - 1)); ec_st(&L->ec, e); }
   const int sret = silk_encode_wave(&L->S, &sc, pcm_hp, frame_size, &L->ec, L->packet + 1, sh->activity, G, &gs->lbrr);
   wv_sync();
   if (sret) { LANE0 { *len_out = sret == -100 ? OA_ERR_UNIMPLEMENTED : OA_ERR_INTERNAL; *rng_out = 0; gs->s.error = sret; } return; }
   SE_PHASE(&L->S, 9);
   /* ---- finalise (:2190-2560) ---- */
   LANE0 {
      int curr_bandwidth = sh->curr_bandwidth;
      if (st->mode == OA_MODE_SILK_ONLY) { if (sc.internalSampleRate == 8000) curr_bandwidth = OA_BW_NB; else if (sc.internalSampleRate == 12000) curr_bandwidth = OA_BW_MB; else if (sc.internalSampleRate == 16000) curr_bandwidth = OA_BW_WB; }
      st->sm_allowBandwidthSwitch = sc.allowBandwidthSwitch; st->sm_inWBmodeWithoutVariableLP = sc.inWBmodeWithoutVariableLP; st->sm_switchReady = sc.switchReady;
      st->sm_opusCanSwitch = sc.switchReady;                                          /* (!nonfinal_frame: single-packet calls only) */
      const int nBytes = L->S.r[0];
      int ret;
      if (nBytes == 0) { st->rangeFinal = 0; L->packet[0] = sh_gen_toc(st->mode, Fs / frame_size, curr_bandwidth, st->stream_channels); ret = 1; sh->pad_to = 0; }   /* SILK DTX (:2242): straight out, no bookkeeping */
      else {
         if (st->sm_opusCanSwitch) {
            if (L->cfg.application != OA_APP_RESTRICTED_SILK) st->error = OA_ERR_UNIMPLEMENTED;          /* the next frame would need a redundant CELT frame */
            st->silk_bw_switch = L->cfg.application != OA_APP_RESTRICTED_SILK;
         }
         if (st->silk_bw_switch) { sh->ret = OA_ERR_UNIMPLEMENTED; }                 /* this very frame would carry the redundant CELT frame (:2251-2262) */
         else {
         sh_width_and_bookkeeping_l0(L, frame_size, sc.stereoWidth_Q14);
         if (st->mode == OA_MODE_HYBRID) {                                            /* :2402-2450: the redundancy flag, then the CELT layer takes the coder over */
            EcCtx e_; ec_ld(&e_, &L->ec); EcCtx *e = &e_; WV_LDS u8 *buf = L->packet + 1;
            if (k_ec_tell(EC_PASS) + 17 + 20 <= 8 * (sh->max_data_bytes - 1)) k_ec_enc_bit_logp(EC_PASS, 0, 12);
            sh->nb_compr_bytes = sh->max_data_bytes - 1;
            k_ec_enc_shrink(EC_PASS, (u32)sh->nb_compr_bytes);
            ec_st(&L->ec, e);
            sh->curr_bandwidth = curr_bandwidth; sh->silk_signalType = sc.signalType; sh->silk_offset = sc.offset;
            sh->ret = -1000;                                                          /* continue in sh_hybrid_celt_wave */
         } else {
         EcCtx e_; ec_ld(&e_, &L->ec); EcCtx *e = &e_; WV_LDS u8 *buf = L->packet + 1;
         const int tell = k_ec_tell(EC_PASS);
         ret = (tell + 7) >> 3;
         st->rangeFinal = e->rng;
         k_ec_enc_done(EC_PASS);
         L->packet[0] = sh_gen_toc(st->mode, Fs / frame_size, curr_bandwidth, st->stream_channels);
         if (tell > (sh->max_data_bytes - 1) * 8) {
            if (sh->max_data_bytes < 2) ret = OA_ERR_BUFFER_TOO_SMALL; else { L->packet[1] = 0; ret = 1; st->rangeFinal = 0; }
         } else while (ret > 2 && L->packet[ret] == 0) ret--;                         /* trailing zeros are implied in SILK-only packets (:2540) */
         if (ret >= 0) ret += 1;
         sh->ret = ret;
         }
         }
      }
      if (nBytes == 0) sh->ret = ret;
      else if (sh->ret >= 0 && sh_generalised_dtx_l0(L, frame_size, Fs)) { st->rangeFinal = 0; L->packet[0] = sh_gen_toc(st->mode, Fs / frame_size, curr_bandwidth, st->stream_channels); sh->ret = 1; sh->pad_to = 0; }
   }
   if (sh->ret == -1000) { SE_CLK_BEGIN(); sh_hybrid_celt_wave(L, gs, pcm_hp, frame_size, out, out_cap, len_out, rng_out); SE_CLK_END(16); return; }
   /* ---- store packet + state (coalesced) ---- */
   {
      const int nbytes = sh->ret < 0 ? sh->ret : sh_emit_packet(L->packet, out, sh->ret, sh->pad_to, out_cap);
      LANE0 { *len_out = nbytes; *rng_out = st->rangeFinal; }
      i32 *g = (i32 *)&gs->s; const WV_LDS i32 *d = (const WV_LDS i32 *)st;
      FOR_LANES(i, (int)(sizeof(OaShScalars) / 4)) g[i] = d[i];
      g = (i32 *)&gs->silk; d = (const WV_LDS i32 *)&L->S.st;
      FOR_LANES(i, SE_STATE_WORDS(CC)) g[i] = d[i];
   }
   SE_PHASE(&L->S, 10);
}
#endif

/* opus_enc_sh.h — one encode call of one stream of the SILK-capable Opus encoder on one wavefront: HBM state -> LDS -> packet bytes -> HBM.
 *
 * Follows src/opus_encoder.c:1182 opus_encode_native (the call: rate, channel, mode, redundancy / prefill and bandwidth decisions :1310-1696, the split of calls
 * above 20 ms (60 ms in SILK-only) into frames that are re-framed as one packet :1698-1838) and :1855 opus_encode_frame_native (one coded frame: activity :1911-1930,
 * SILK bandwidth-switch redundancy :1933-1940, high-pass :1969-2009, SILK control block and call :2043-2261 with the prefill :2191-2209, CELT control :2264-2295,
 * delay line / gain and stereo fades :2297-2349, redundancy signalling :2351-2383, the 5 ms redundant CELT frames :2427-2442 / :2514-2545 and the 2.5 ms CELT
 * prefill :2478-2486, TOC / range / bookkeeping :2549-2562, DTX :2565-2576, CBR padding :2646) for the applications VOIP, AUDIO and RESTRICTED_SILK at every API
 * rate (8-48 kHz), mono / stereo, 2.5-120 ms.  Mode switches in every direction (SILK / hybrid <-> CELT-only with the redundant frame on the right side of the
 * switch, SILK bandwidth switches), in-band FEC, both DTX flavours.
 *
 * LDS protocol: the packet being built lives in SH_PKT(L) (ShLds); the SILK working set and the CELT frame arena alias each other in L->S, so a frame that needs
 * the CELT coder first sends the SILK state back to HBM (sh_enter_celt) and the next frame of the same call brings it back (sh_reload_silk). */
#ifndef OPUS_AMD_OPUS_ENC_SH_H
#define OPUS_AMD_OPUS_ENC_SH_H
#include "opus_sh_state.h"
#include "opus_multiframe.h"

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
   /* the call (opus_encode_native) */
   i32 frame_size, max_data_bytes, plc_frame, ret, err, is_silence, sample_max, stereo_width;
   i32 bitrate_bps, equiv_rate, redundancy, celt_to_silk, to_celt, prefill, cbr_bytes, lsb_depth;
   i32 nb_frames, enc_frame_size, repacketize_len, max_len_sum;
   /* the frame (opus_encode_frame_native) */
   i32 f_size, f_max_data_bytes, f_orig_max_data_bytes, f_redundancy, f_celt_to_silk, f_prefill, f_to_celt, f_silence;
   i32 curr_bandwidth, activity, cutoff_Hz, use_hp_cutoff, bits_target, redundancy_bytes, nb_compr_bytes, silk_bitRate, HB_gain;
   i32 silk_signalType, silk_offset, redundant_rng, start_band, celt_ret, silk_in_lds, tell_frac0;
   i32 do_gain_fade, do_stereo_fade, fade_g1, fade_g2, hb_g1, hb_g2, need_celt_prefill;
   i32 r[8];
};
struct ShLds {
   EcCtx ec;
   ShShared sh;
   OaShScalars st;
   OaShConfig cfg;
   MfLds mf;
   CeltScratch *cs;                                      /* the stream's HBM scratch of the CELT passes (celt_enc_lds.h) */
   i32 silk_tail, pad_;                                  /* 1: this kernel stages the quantiser tails of the SILK state (OaSilkEncTail) too; 0: the split path's front kernel (set by the kernel before the call opens) */
   i32 packet_off, pad2_;                                /* the packet being built: SH_PKT(L), behind everything else the kernel allocates -- OA_MAX_PACKET + 4 bytes, except in the
                                                          * split path's front kernel, which codes a few header symbols and gets SH_FRONT_PKT_BYTES (set by the kernel before the call opens) */
   SilkEncLds S;                                         /* LAST (its own last member is the SILK state): mono batches allocate SH_LDS_BYTES(1) */
};
#define SH_PKT(L) ((WV_LDS u8 *)(L) + (L)->packet_off)
/* the CELT passes' arena (FrameLds) borrows the SILK encoder's LDS behind its 16-byte header */
#define SH_F(L) ((WV_LDS FrameLds *)((WV_LDS char *)&(L)->S + 16))
#define SH_CELT_LDS_BYTES (offsetof(ShLds, S) + 16 + sizeof(FrameLds))
static_assert(sizeof(AnLds) <= SE_FRONT_U_BYTES + offsetof(OaSilkEnc, ch) + sizeof(OaSilkEncChannel) /* (it runs before the state is staged: it may reach into where a mono stream's copy goes) */ && alignof(AnLds) <= 16 && (offsetof(ShLds, S) + offsetof(SilkEncLds, u)) % 16 == 0, "the tonality analysis works in the SILK encoder's phase union, in every kernel");
#define SH_PKT_BYTES (OA_MAX_PACKET + 4)
#define SH_FRONT_PKT_BYTES 64
#define SH_LDS_BYTES(channels) (sizeof(ShLds) - ((channels) == 1 ? sizeof(OaSilkEncTail) : 0))                 /* without the packet: the kernels add it behind (packet_off) */
#define SH_FRONT_LDS_BYTES(channels) (offsetof(ShLds, S) + SE_FRONT_LDS_BYTES(channels))
/* per-stream HBM scratch: the high-passed input of the frame, the faded CELT input of the frame (only written when a frame needs more than one CELT pass),
 * the 2.5 ms CELT prefill, the CELT passes' bulk arrays (CeltScratch), then the rate-loop snapshots */
#define SH_PCM_BYTES(frame_size, channels) (((size_t)(frame_size) * (channels) * 2 + 63) / 64 * 64)
#define SH_SCRATCH_BYTES(frame_size, channels) (2 * SH_PCM_BYTES(frame_size, channels) + 512 + sizeof(CeltScratch) + sizeof(SeRateScratch))
#define SH_STAGE_SAMPLES 1920

/* OPUS_SET_FORCE_CHANNELS as the encoder sees it: 1 while the multi-frame path's own 'force_channels = 1' (OaShScalars.mono_forced_seq) is in force */
#define SH_FORCE_CHANNELS(cfg, st) ((st)->mono_forced_seq == (cfg)->force_channels_seq + 1 ? 1 : (cfg)->force_channels)
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
/* compute_redundancy_bytes :1142 */
WV_DEV int sh_redundancy_bytes(i32 max_data_bytes, i32 bitrate_bps, int frame_rate, int channels)
{
   const int base_bits = 40 * channels + 20;
   i32 redundancy_rate = bitrate_bps + base_bits * (200 - frame_rate);
   redundancy_rate = 3 * redundancy_rate / 2;
   int redundancy_bytes = redundancy_rate / 1600;
   const i32 available_bits = max_data_bytes * 8 - 2 * base_bits;
   const int cap = (available_bits * 240 / (240 + 48000 / frame_rate) + base_bits) / 8;
   redundancy_bytes = imin(redundancy_bytes, cap);
   return redundancy_bytes > 4 + 8 * channels ? imin(257, redundancy_bytes) : 0;
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
/* celt_maxabs16 of n int16 samples in HBM */
WV_DEV i32 sh_maxabs_wave(const i16 *pcm, int n)
{
   if (!((size_t)pcm & 3) && !(n & 1)) return oa_maxabs_wave(pcm, n);      /* (two samples per word, eight trips in flight) */
   i32 m = 0;
   FOR_LANES(i, n) m = imax(m, iabs((i32)pcm[i]));
   return wv_max(m);
}
/* compute_frame_energy (:1080): a plain int32 sum of down-shifted squares (order-free), normalised by the length */
WV_DEV i32 sh_frame_energy_wave(const i16 *pcm, int len)
{
   const i32 sample_max = sh_maxabs_wave(pcm, len);
   const int shift = imax(0, (celt_ilog2(1 + sample_max) << 1) + celt_ilog2(len) - 28);
   if (!((size_t)pcm & 3) && !(len & 1)) return oa_frame_energy_wave(pcm, len, sample_max);
   i32 e = 0;
   FOR_LANES(i, len) e += mult16_16(pcm[i], pcm[i]) >> shift;
   e = wv_sum(e);
   e /= len;
   return shl32(e, shift);
}

/* lane 0: opus_encode_native's decisions for the call (:1325-1696) */
WV_DEVN void sh_layer_decide(WV_LDS ShLds *L, int frame_size, int out_data_bytes, const OaAnalysisInfo *info)
{
   WV_LDS ShShared *sh = &L->sh; WV_LDS OaShScalars *st = &L->st; const WV_LDS OaShConfig *cfg = &L->cfg;
   const int Fs = cfg->Fs, channels = cfg->channels;
   i32 max_data_bytes = imin(1276 * 6, out_data_bytes);
   st->rangeFinal = 0;
   sh->plc_frame = 0; sh->ret = 0; sh->err = 0; sh->frame_size = frame_size; sh->cbr_bytes = -1;
   sh->redundancy = 0; sh->celt_to_silk = 0; sh->to_celt = 0; sh->prefill = 0; sh->nb_frames = 1; sh->enc_frame_size = frame_size;
   sh->lsb_depth = imin(cfg->input_depth ? cfg->input_depth : 16, cfg->lsb_depth);
   if (max_data_bytes == 1 && Fs == frame_size * 10) { sh->err = OA_ERR_BUFFER_TOO_SMALL; return; }
   /* voice_ratio and the detected bandwidth from the analysis of the call (:1273-1308); without the float API voice_ratio is always -1 */
   if (!sh->is_silence || cfg->analysis_off) st->voice_ratio = -1;
   int detected_bandwidth = 0;
   if (info->valid) {
      if (cfg->signal_type == OA_AUTO) st->voice_ratio = an_voice_ratio(info, st->prev_mode);
      detected_bandwidth = an_detected_bandwidth(info->bandwidth);
   }
   const i32 user = cfg->user_bitrate_bps == OA_AUTO ? 60 * Fs / frame_size + Fs * channels : (cfg->user_bitrate_bps == OA_BITRATE_MAX ? 1500000 : cfg->user_bitrate_bps);
   i32 bitrate_bps = imin(user, bits_to_bitrate(max_data_bytes * 8, Fs, frame_size));
   int frame_rate = Fs / frame_size;
   if (!cfg->use_vbr) {
      const i32 cbr_bytes = imin((bitrate_to_bits(bitrate_bps, Fs, frame_size) + 4) / 8, max_data_bytes);
      bitrate_bps = bits_to_bitrate(cbr_bytes * 8, Fs, frame_size);
      max_data_bytes = imax(1, cbr_bytes);
      sh->cbr_bytes = cbr_bytes;
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
      SH_PKT(L)[0] = (u8)(sh_gen_toc(tocmode, frame_rate, bw, st->stream_channels) | packet_code);
      if (packet_code == 3) SH_PKT(L)[1] = (u8)num_multiframes;
      sh->plc_frame = 1; sh->ret = packet_code <= 1 ? 1 : 2;
      sh->max_data_bytes = imax(max_data_bytes, sh->ret);
      return;
   }
   const i32 max_rate = bits_to_bitrate(max_data_bytes * 8, Fs, frame_size);
   const int loss = cfg->packet_loss_perc;
   i32 equiv_rate = sh_equiv_rate(bitrate_bps, channels, frame_rate, cfg->use_vbr, 0, cfg->complexity, loss);
   int voice_est;
   if (cfg->signal_type == OA_SIGNAL_VOICE) voice_est = 127; else if (cfg->signal_type == OA_SIGNAL_MUSIC) voice_est = 0;
   else if (st->voice_ratio >= 0) { voice_est = st->voice_ratio * 327 >> 8; if (cfg->application == OA_APP_AUDIO) voice_est = imin(voice_est, 115); }   /* for AUDIO, never more than 90% confident of having speech */
   else if (cfg->application == OA_APP_VOIP) voice_est = 115; else voice_est = 48;
   const int force_channels = SH_FORCE_CHANNELS(cfg, st);
   if (force_channels != OA_AUTO && channels == 2) st->stream_channels = force_channels;
   else if (channels == 2) {
      i32 thr = 17000 + ((voice_est * voice_est * (19000 - 17000)) >> 14);
      if (st->stream_channels == 2) thr -= 1000; else thr += 1000;
      st->stream_channels = equiv_rate > thr ? 2 : 1;
   } else st->stream_channels = channels;
   equiv_rate = sh_equiv_rate(bitrate_bps, st->stream_channels, frame_rate, cfg->use_vbr, 0, cfg->complexity, loss);
   st->sm_useDTX = cfg->use_dtx && !(info->valid || sh->is_silence);                              /* :1461: SILK's own DTX only where the generalised one cannot be used */
   /* mode (:1466-1539) */
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
   if (cfg->lfe && cfg->application != OA_APP_RESTRICTED_SILK) st->mode = OA_MODE_CELT_ONLY;
   /* a switch between the SILK-based modes and CELT-only carries a 5 ms redundant CELT frame on the SILK side of it (:1541-1558) */
   if (st->prev_mode > 0 && ((st->mode != OA_MODE_CELT_ONLY) != (st->prev_mode != OA_MODE_CELT_ONLY))) {
      sh->redundancy = 1;
      sh->celt_to_silk = st->mode != OA_MODE_CELT_ONLY;
      if (!sh->celt_to_silk) {
         if (frame_size >= Fs / 100) { st->mode = st->prev_mode; sh->to_celt = 1; }            /* this call is still coded in the old mode, the redundant frame follows it */
         else sh->redundancy = 0;
      }
   }
   if (st->stream_channels == 1 && st->prev_channels == 2 && st->sm_toMono == 0 && st->mode != OA_MODE_CELT_ONLY && st->prev_mode != OA_MODE_CELT_ONLY) { st->sm_toMono = 1; st->stream_channels = 2; }
   else st->sm_toMono = 0;
   equiv_rate = sh_equiv_rate(bitrate_bps, st->stream_channels, frame_rate, cfg->use_vbr, st->mode, cfg->complexity, loss);
   if (st->mode != OA_MODE_CELT_ONLY && st->prev_mode == OA_MODE_CELT_ONLY) sh->prefill = 1;      /* + silk_InitEncoder, done by the wave right after this section (:1576-1581) */
   /* bandwidth (:1583-1696) */
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
   if (detected_bandwidth && cfg->user_bandwidth == OA_AUTO) {                                    /* use the detected bandwidth to reduce the encoded bandwidth (:1651-1674); SILK / hybrid never below wideband */
      const i32 sc = st->stream_channels; const int celt = st->mode == OA_MODE_CELT_ONLY;
      const int min_detected_bandwidth = equiv_rate <= 18000 * sc && celt ? OA_BW_NB : equiv_rate <= 24000 * sc && celt ? OA_BW_MB : equiv_rate <= 30000 * sc ? OA_BW_WB : equiv_rate <= 44000 * sc ? OA_BW_SWB : OA_BW_FB;
      st->bandwidth = imin(st->bandwidth, imax(detected_bandwidth, min_detected_bandwidth));
   }
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
   if (cfg->lfe) st->bandwidth = OA_BW_NB;
   int curr_bandwidth = st->bandwidth;
   if (cfg->application == OA_APP_RESTRICTED_SILK && curr_bandwidth > OA_BW_WB) st->bandwidth = curr_bandwidth = OA_BW_WB;
   if (st->mode == OA_MODE_SILK_ONLY && curr_bandwidth > OA_BW_WB) st->mode = OA_MODE_HYBRID;
   if (st->mode == OA_MODE_HYBRID && curr_bandwidth <= OA_BW_WB) st->mode = OA_MODE_SILK_ONLY;
   sh->bitrate_bps = bitrate_bps; sh->equiv_rate = equiv_rate; sh->max_data_bytes = max_data_bytes;
   /* more than one coded frame? (:1698-1755) */
   if ((frame_size > Fs / 50 && st->mode != OA_MODE_SILK_ONLY) || frame_size > 3 * Fs / 50) {
      int enc_frame_size;
      if (st->mode == OA_MODE_SILK_ONLY) enc_frame_size = frame_size == 2 * Fs / 25 ? Fs / 25 : frame_size == 3 * Fs / 25 ? 3 * Fs / 50 : Fs / 50;
      else enc_frame_size = Fs / 50;
      const int nb_frames = frame_size / enc_frame_size, max_header_bytes = nb_frames == 2 ? 3 : 2 + (nb_frames - 1) * 2;
      sh->repacketize_len = (cfg->use_vbr || cfg->user_bitrate_bps == OA_BITRATE_MAX) ? out_data_bytes : imin(sh->cbr_bytes, out_data_bytes);
      sh->max_len_sum = nb_frames + sh->repacketize_len - max_header_bytes;
      sh->nb_frames = nb_frames; sh->enc_frame_size = enc_frame_size;
   }
}

/* hp_cutoff (:441, VOIP) or dc_reject (:479) over one chunk staged in LDS: lane c runs channel c's recursion */
WV_DEV void sh_highpass_chunk(WV_LDS ShLds *L, WV_LDS i16 *io, int len, int channels, const i32 *B_Q28, const i32 *A_Q28)
{
   const int c = wv_lane();
   if (c < channels) {
      if (L->sh.use_hp_cutoff) {
         const i32 A0_L = (-A_Q28[0]) & 0x3FFF, A0_U = (-A_Q28[0]) >> 14, A1_L = (-A_Q28[1]) & 0x3FFF, A1_U = (-A_Q28[1]) >> 14;
         i32 S0 = L->st.hp_mem[2 * c], S1 = L->st.hp_mem[2 * c + 1];
         int k = 0;
         for (; k + 8 <= len; k += 8) {                  /* eight samples per trip: the LDS reads of a trip are issued back to back, only the chain through S0 / S1 is serial */
            i32 x[8];
#pragma unroll
            for (int u = 0; u < 8; u++) x[u] = io[(k + u) * channels + c];
#pragma unroll
            for (int u = 0; u < 8; u++) {
               const i32 inval = x[u];
               const i32 out32_Q14 = shl32(sk_mlawb(S0, B_Q28[0], inval), 2);
               S0 = S1 + sk_rround(sk_mulwb(out32_Q14, A0_L), 14); S0 = sk_mlawb(S0, out32_Q14, A0_U); S0 = sk_mlawb(S0, B_Q28[1], inval);
               S1 = sk_rround(sk_mulwb(out32_Q14, A1_L), 14); S1 = sk_mlawb(S1, out32_Q14, A1_U); S1 = sk_mlawb(S1, B_Q28[2], inval);
               x[u] = sk_sat16((out32_Q14 + (1 << 14) - 1) >> 14);
            }
#pragma unroll
            for (int u = 0; u < 8; u++) io[(k + u) * channels + c] = (i16)x[u];
         }
         for (; k < len; k++) {
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
         int k = 0;
         for (; k + 8 <= len; k += 8) {
            i32 x[8];
#pragma unroll
            for (int u = 0; u < 8; u++) x[u] = shl32(saturate((i32)io[(k + u) * channels + c], (1 << 16) - 1), 14);
#pragma unroll
            for (int u = 0; u < 8; u++) { const i32 y = x[u] - mem; mem = mem + pshr32(y, shift); x[u] = saturate(pshr32(y, 14), 32767); }
#pragma unroll
            for (int u = 0; u < 8; u++) io[(k + u) * channels + c] = (i16)x[u];
         }
         for (; k < len; k++) {
            const i32 x = shl32(saturate((i32)io[k * channels + c], (1 << 16) - 1), 14), y = x - mem;
            mem = mem + pshr32(y, shift);
            io[k * channels + c] = (i16)saturate(pshr32(y, 14), 32767);
         }
         L->st.hp_mem[2 * c] = mem;
      }
   }
}

#define sh_emit_packet oa_emit_packet_wave            /* celt_enc_frame.h: plain store, or the code-3 re-framing of hard CBR */

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

/* the generalised DTX decision at the end of opus_encode_frame_native (:2565-2576, decide_dtx_mode :1115); 1 = send the TOC byte alone */
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

/* ---- SILK <-> CELT arena ---- */
WV_DEV void sh_enter_celt(WV_LDS ShLds *L, OaShStream *gs)                               /* SILK state back to HBM (coalesced) if it is in LDS; CELT scalars in */
{
   const int CC = L->cfg.channels;
   wv_sync();
   if (wv_uni(L->sh.silk_in_lds)) {
      se_state_copy_wave((i32 *)&gs->silk, (const WV_LDS i32 *)se_st(&L->S), CC, wv_uni(L->silk_tail));
      wv_sync();
      LANE0 L->sh.silk_in_lds = 0;
   }
   WV_LDS FrameLds *F = SH_F(L);
   const i32 *g = (const i32 *)&gs->celt.s; WV_LDS i32 *d = (WV_LDS i32 *)&F->st;
   FOR_LANES(i, (int)(sizeof(OaEncScalars) / 4)) d[i] = g[i];
   FOR_LANES(i, 2 * NBE) F->oldBandE[i] = gs->celt.oldBandE[i];
   wv_sync();
}
WV_DEV void sh_leave_celt(WV_LDS ShLds *L, OaShStream *gs)
{
   WV_LDS FrameLds *F = SH_F(L);
   wv_sync();
   i32 *g = (i32 *)&gs->celt.s; const WV_LDS i32 *d = (const WV_LDS i32 *)&F->st;
   FOR_LANES(i, (int)(sizeof(OaEncScalars) / 4)) g[i] = d[i];
   wv_sync();
}
WV_DEV void sh_reload_silk(WV_LDS ShLds *L, const OaShStream *gs)
{
   if (wv_uni(L->sh.silk_in_lds)) return;
   wv_sync();
   se_state_copy_wave((WV_LDS i32 *)se_st(&L->S), (const i32 *)&gs->silk, L->cfg.channels, wv_uni(L->silk_tail));
   wv_sync();
   LANE0 L->sh.silk_in_lds = 1;
}
/* SILK state back to HBM (if it is in LDS): whoever borrows the arena next may overwrite it; sh_reload_silk brings it back */
WV_DEV void sh_park_silk(WV_LDS ShLds *L, OaShStream *gs)
{
   wv_sync();
   if (wv_uni(L->sh.silk_in_lds)) {
      se_state_copy_wave((i32 *)&gs->silk, (const WV_LDS i32 *)se_st(&L->S), L->cfg.channels, wv_uni(L->silk_tail));
      wv_sync();
      LANE0 L->sh.silk_in_lds = 0;
   }
}
/* OPUS_RESET_STATE of the CELT encoder (celt_encoder.c:2972-2992) on the state in HBM + the scalars in the arena; configuration-like words survive */
WV_DEV void sh_celt_reset_wave(WV_LDS ShLds *L, OaShStream *gs)
{
   WV_LDS FrameLds *F = SH_F(L);
   wv_sync();
   LANE0 {
      WV_LDS OaEncScalars *c = &F->st;
      const i32 dpf = c->pad0[0], fi = c->pad0[1];
      WV_LDS i32 *w = (WV_LDS i32 *)c; for (int i = 0; i < (int)(sizeof(OaEncScalars) / 4); i++) w[i] = 0;
      c->pad0[0] = dpf; c->pad0[1] = fi;
      c->spread_decision = 2; c->delayedIntra = 1; c->tonal_average = 256;
      L->st.celt_mask_cleared = 1;                                                       /* ... and so does CELT's energy_mask pointer: no surround masking in CELT until the mask is set again */
      L->sh.silk_signalType = 0; L->sh.silk_offset = 0;                                  /* SILKInfo sits in the reset region too: the CELT passes after a reset see zeros until the next frame sets it */
   }
   FOR_LANES(i, 2 * NBE) { F->oldBandE[i] = 0; gs->celt.oldBandE[i] = 0; gs->celt.energyError[i] = 0; gs->celt.oldLogE[i] = gs->celt.oldLogE2[i] = -(28 << 24); }
   FOR_LANES(i, 2 * OA_OVERLAP) gs->celt.in_mem[i] = 0;
   FOR_LANES(i, 2 * OA_MAX_PERIOD) gs->celt.prefilter_mem[i] = 0;
   FOR_LANES(i, (int)(sizeof(OaAnalysisInfo) / 4)) ((i32 *)&gs->celt.analysis)[i] = 0;       /* the AnalysisInfo of CELT_SET_ANALYSIS sits in the reset region as well (celt_encoder.c:109) */
   wv_sync();
}
/* the persistent CELT_SET_PREDICTION state (celt_encoder.c:2893: disable_pf = value <= 1, force_intra = value == 0) lives in two spare scalar words */
#define SH_CELT_DISABLE_PF(F) ((F)->st.pad0[0])
#define SH_CELT_FORCE_INTRA(F) ((F)->st.pad0[1])

/* One celt_encode_with_ec (celt/celt_encoder.c:1726) on the arena: `src` = nsamp * CC int16 samples at the API rate in HBM (NULL: already staged in the HBM scratch L->cs->pcm16).
 *   raw = 0: the CELT layer of the frame being built, continuing the coder in L->ec on the bytes in SH_PKT(L) (hybrid), or starting it (CELT-only frame)
 *   raw = 1: a self-contained redundancy / prefill frame of `nbytes` bytes; its bytes end up at F->packet + 1, its return value in L->sh.celt_ret */
struct ShCeltCtl { int start, vbr, constrained_vbr, nbytes, raw, cont; i32 bitrate; };     /* raw: own nbytes-byte buffer; cont: continue the frame's coder after the SILK layer */
/* what follows celt_encode_with_ec in a pass: its return value and the coder's position, for the caller */
WV_DEV void sh_celt_run_tail(WV_LDS ShLds *L)
{
   WV_LDS ShShared *sh = &L->sh;
   WV_LDS FrameLds *F = SH_F(L);
   wv_sync();
   LANE0 { EcCtx t; ec_ld(&t, &F->ec); sh->celt_ret = F->sh.ret; sh->r[4] = k_ec_tell(&t, F->packet + 1); }
   wv_sync();
}
/* cut: the stream's continuation record -- the pass stops before the PVQ (celt_enc_frame.h: OA_CUT, returned) and goes on in oa_celt_pvq_kernel / oa_sh_back2_kernel */
WV_DEVN int sh_celt_run(WV_LDS ShLds *L, OaShStream *gs, const i16 *src, int nsamp, const ShCeltCtl ctl, u8 *journal, const i32 *tr = nullptr /* the transient pre-pass's record of the
      stream (opus_sh_split.h: oa_sh_transient_tile), for the frame's own pass on the input this function finds staged */, CeltCont *cut = nullptr)
{
   WV_LDS ShShared *sh = &L->sh; WV_LDS OaShScalars *st = &L->st;
   WV_LDS FrameLds *F = SH_F(L);
   WV_LDS FrameShared *fs = &F->sh;
   const int CC = L->cfg.channels, Fs = L->cfg.Fs, up = 48000 / Fs;
   SE_CLK_BEGIN();                                  /* (profiling build: the pass is timed from a register, the SILK hand-off words are about to be overwritten) */
   wv_sync();
   if (wv_lane() == 0) F->g = L->cs;               /* (the arena aliases the SILK working set: whatever SILK did since the last CELT pass may have overwritten the pointer) */
   wv_sync();
   if (src) { i16 *dst = L->cs->pcm16; const int n = nsamp * CC; if (!((size_t)src & 3) && !(n & 1)) wv_copy_batched((u32 *)dst, (const u32 *)src, n >> 1); else wv_copy_batched(dst, src, n); }
   if (!ctl.raw) { FOR_LANES(i, (OA_MAX_PACKET + 4) / 4) ((WV_LDS i32 *)F->packet)[i] = ((const WV_LDS i32 *)SH_PKT(L))[i]; }
   wv_sync();
   {  /* celt_maxabs over the head and the overlap tail of the input (celt_encoder.c:1970-1973), at the API rate */
      const i16 *p = L->cs->pcm16;
      const int ov = OA_OVERLAP / up;
      i32 a = 0, b = 0;
      const int na = CC * (nsamp - ov), nb = CC * ov;
      if (!((na | nb) & 1)) { a = na ? oa_maxabs_wave(p, na) : 0; b = oa_maxabs_wave(p + na, nb); }           /* (by words, eight trips in flight; pcm16 starts on a word) */
      else {
      FOR_LANES(i, na) a = imax(a, iabs((i32)p[i]));
      FOR_LANES(i, nb) b = imax(b, iabs((i32)p[na + i]));
      a = wv_max(a); b = wv_max(b);
      }
      LANE0 { fs->r[0] = a; fs->r[1] = b; }
   }
   LANE0 {
      if (ctl.cont) ec_cp_lds(&F->ec, &L->ec);
      const int curr_bandwidth = sh->curr_bandwidth, endband = curr_bandwidth == OA_BW_NB ? 13 : curr_bandwidth <= OA_BW_WB ? 17 : curr_bandwidth == OA_BW_SWB ? 19 : 21;
      fs->CC = CC; fs->C = st->stream_channels; fs->frame_size = nsamp; fs->upsample = up; fs->raw_frame = 1;        /* this Opus layer does its own finalisation */
      fs->start = ctl.start; fs->end = endband; fs->effEnd = endband;
      fs->complexity = L->cfg.complexity; fs->lsb_depth = sh->lsb_depth; fs->disable_inv = L->cfg.disable_inv; fs->loss_rate = L->cfg.packet_loss_perc;
      fs->disable_pf = SH_CELT_DISABLE_PF(F); fs->force_intra = SH_CELT_FORCE_INTRA(F);
      fs->vbr = ctl.vbr; fs->constrained_vbr = ctl.constrained_vbr; fs->bitrate = ctl.bitrate;
      fs->curr_bandwidth = curr_bandwidth;
      fs->max_data_bytes = ctl.raw ? ctl.nbytes + 1 : sh->f_max_data_bytes; fs->orig_max_data_bytes = ctl.raw ? ctl.nbytes + 1 : sh->f_orig_max_data_bytes; fs->pad_to = 0;
      fs->plc_frame = 0; fs->ret = 0; fs->skip_celt = 0; fs->toc = 0;
      fs->silk_signalType = sh->silk_signalType; fs->silk_offset = sh->silk_offset;
      fs->do_stereo_fade = 0; fs->lfe = L->cfg.lfe; fs->energy_mask_on = L->cfg.energy_mask_on && !st->celt_mask_cleared; fs->Fs = Fs;
   }
   wv_sync();
   LANE0 celt_prologue(F, ctl.cont ? sh->nb_compr_bytes : 0);
   wv_sync();
   if (fs->skip_celt) { LANE0 sh->celt_ret = -1000; wv_sync(); SE_CLK_END(16); SE_PHASE_START(&L->S); return 0; }                /* budget already gone: the caller emits the "PLC" byte (:2487) */
   /* the pre-pass worked from the gains it expected this function's caller to apply to the input: its values hold if those are the ones that were applied */
   const i32 *tp = nullptr;
   if (tr && !ctl.raw && !src) {
      const int dg = wv_uni(sh->do_gain_fade), ds = wv_uni(sh->do_stereo_fade);
      if (wv_uni(tr[3]) == dg && (!dg || (wv_uni(tr[4]) == wv_uni(sh->hb_g1) && wv_uni(tr[5]) == wv_uni(sh->hb_g2))) && wv_uni(tr[6]) == ds && (!ds || (wv_uni(tr[7]) == wv_uni(sh->fade_g1) && wv_uni(tr[8]) == wv_uni(sh->fade_g2)))) tp = tr;
   }
   int r;
   if (ctl.start != 0) r = celt_encode_core<true>(F, &gs->celt, journal, gs->energy_mask, tp, cut);           /* "hybrid" inside CELT = start band above 0 (celt_encoder.c:1809) */
   else r = celt_encode_core<false>(F, &gs->celt, journal, gs->energy_mask, tp, cut);
   if (r == OA_CUT) { SE_CLK_END(16); return OA_CUT; }
   sh_celt_run_tail(L);
   SE_CLK_END(16); SE_PHASE_START(&L->S);
   return 0;
}

/* gain_fade (:581) / stereo_fade (:548) on `n` frames of CC interleaved int16 (LDS staging or HBM scratch); the cross-fade covers overlap = 120 * Fs / 48000 samples, window read with stride inc */
template <class P16> WV_DEV void sh_gain_fade_lds(P16 io, int n, int CC, i16 g1, i16 g2, int Fs)
{
   const int inc = 48000 / Fs, overlap = OA_OVERLAP / inc;
   FOR_LANES(i, n * CC) {
      const int k = i / CC; i16 g = g2;
      if (k < overlap) { i16 w = ct_window[k * inc]; w = (i16)mult16_16_q15(w, w); g = (i16)(mac16_16(mult16_16(w, g2), Q15ONE - w, g1) >> 15); }
      io[i] = (i16)mult16_16_q15(g, io[i]);
   }
}
template <class P16> WV_DEV void sh_stereo_fade_lds(P16 io, int n, i16 g1_, i16 g2_, int Fs)
{
   const int inc = 48000 / Fs, overlap = OA_OVERLAP / inc;
   const i16 g1 = (i16)(Q15ONE - g1_), g2 = (i16)(Q15ONE - g2_);
   FOR_LANES(i, n) {
      i16 g = g2;
      if (i < overlap) { i16 w = ct_window[i * inc]; w = (i16)mult16_16_q15(w, w); g = (i16)(mac16_16(mult16_16(w, g2), Q15ONE - w, g1) >> 15); }
      i32 diff = half32((i32)io[2 * i] - (i32)io[2 * i + 1]);
      diff = mult16_16_q15(g, diff);
      io[2 * i] = (i16)(io[2 * i] - diff);
      io[2 * i + 1] = (i16)(io[2 * i + 1] + diff);
   }
}

/* One coded frame: opus_encode_frame_native (:1855), in two halves around the SILK layer so that the split path (opus_sh_split.h) can run them in kernels of their own.
 * pcm = this frame's input, frame_size samples per channel; the packet ends up in SH_PKT(L), its length (before CBR padding) is returned and st / the HBM state are
 * updated.  pcm_hp / pcm_celt / pre = per-stream HBM scratch.
 * sh_frame_front_wave: activity (:1911-1930), the frame's redundancy / budget words, coder start, high-pass into pcm_hp (:1969-2009) and, when SILK codes the frame, the
 * SILK control block *scp (:2043-2189). */
WV_DEV void sh_frame_front_wave(WV_LDS ShLds *L, OaShStream *gs, const i16 *pcm, int frame_size, int orig_max_data_bytes, i16 *pcm_hp, SeControl *scp)
{
   WV_LDS ShShared *sh = &L->sh; WV_LDS OaShScalars *st = &L->st;
   const int CC = L->cfg.channels, Fs = L->cfg.Fs, application = L->cfg.application;
   const int frame_rate = Fs / frame_size;
   frame_size = wv_uni(frame_size); orig_max_data_bytes = wv_uni(orig_max_data_bytes);
   /* ---- activity (:1911-1930) ---- */
   {
      i32 noise_energy = 0;
      const int celt_only = wv_uni(st->mode) == OA_MODE_CELT_ONLY, an_valid = wv_uni(gs->an_info.valid);
      const int an_active = an_valid ? an_activity_prob_active(&gs->an_info) : 0;
      if (!wv_uni(sh->f_silence) && (an_valid ? !an_active : celt_only)) noise_energy = sh_frame_energy_wave(pcm, frame_size * CC);
      LANE0 {
         sh->activity = SE_VAD_NO_DECISION;
         if (sh->f_silence) sh->activity = 0;
         else if (an_valid) sh->activity = an_active || an_loud_noise_active(st->peak_signal_energy, noise_energy);     /* the analysis' activity probability; loud noise counts as active (:1916-1924) */
         else if (celt_only) sh->activity = (i64)st->peak_signal_energy < 316 * (i64)half32(noise_energy);
      }
   }
   LANE0 {
      int redundancy = sh->f_redundancy, celt_to_silk = sh->f_celt_to_silk, prefill = sh->f_prefill;
      const int max_data_bytes = imin(orig_max_data_bytes, 1276);
      st->rangeFinal = 0;
      if (st->silk_bw_switch) { redundancy = 1; celt_to_silk = 1; st->silk_bw_switch = 0; prefill = 2; }      /* first frame at a new SILK bandwidth (:1933) */
      if (st->mode == OA_MODE_CELT_ONLY) redundancy = 0;
      int redundancy_bytes = 0;
      if (redundancy) { redundancy_bytes = sh_redundancy_bytes(max_data_bytes, sh->bitrate_bps, frame_rate, st->stream_channels); if (redundancy_bytes == 0) redundancy = 0; }
      if (application == OA_APP_RESTRICTED_SILK) { redundancy = 0; redundancy_bytes = 0; }
      sh->f_redundancy = redundancy; sh->f_celt_to_silk = celt_to_silk; sh->f_prefill = prefill; sh->redundancy_bytes = redundancy_bytes;
      sh->f_max_data_bytes = max_data_bytes; sh->f_orig_max_data_bytes = orig_max_data_bytes;
      sh->bits_target = imin(8 * (max_data_bytes - redundancy_bytes), bitrate_to_bits(sh->bitrate_bps, Fs, frame_size)) - 8;
      sh->curr_bandwidth = st->bandwidth;
      sh->redundant_rng = 0; sh->f_size = frame_size; sh->r[3] = 0;                  /* r[3]: the result is a bare TOC that is never padded (DTX) */
      { EcCtx e_; EcCtx *e = &e_; WV_LDS u8 *buf = SH_PKT(L) + 1; k_ec_enc_init(EC_PASS, (u32)imin(orig_max_data_bytes - 1, 1275)); ec_st(&L->ec, e); }   /* the reference's coder spans the caller's whole buffer (:1964); a frame never fills more than 1275 bytes of it, and what lies beyond only ever gets cleared by ec_enc_done -- which here would run past the LDS packet buffer */
      const i32 hp_freq_smth1 = st->mode == OA_MODE_CELT_ONLY ? shl32(se_lin2log(60), 8) : se_st(&L->S)->ch[0].variable_HP_smth1_Q15;
      st->variable_HP_smth2_Q15 = sk_mlawb(st->variable_HP_smth2_Q15, hp_freq_smth1 - st->variable_HP_smth2_Q15, SE_FIX(0.015f, 16));
      sh->cutoff_Hz = se_log2lin(st->variable_HP_smth2_Q15 >> 8);
      sh->use_hp_cutoff = application == OA_APP_VOIP;
   }
   /* ---- high-pass into the per-stream HBM scratch, staged through LDS in chunks (:1980-2009) ---- */
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
         if (!(((size_t)pcm | (size_t)pcm_hp) & 3) && !((i0 * CC | n * CC) & 1)) {           /* by words, eight trips in flight */
            wv_copy_batched((WV_LDS u32 *)stage, (const u32 *)(pcm + i0 * CC), n * CC >> 1);
            wv_sync();
            sh_highpass_chunk(L, stage, n, CC, B_Q28, A_Q28);
            wv_sync();
            wv_copy_batched((u32 *)(pcm_hp + i0 * CC), (const WV_LDS u32 *)stage, n * CC >> 1);
            continue;
         }
         FOR_LANES(i, n * CC) stage[i] = pcm[i0 * CC + i];
         wv_sync();
         sh_highpass_chunk(L, stage, n, CC, B_Q28, A_Q28);
         wv_sync();
         FOR_LANES(i, n * CC) pcm_hp[i0 * CC + i] = stage[i];
      }
      wv_sync();
   }
   SE_PHASE(&L->S, 1);
   const int mode = wv_uni(st->mode);
   if (mode != OA_MODE_CELT_ONLY) {
      SeControl &sc = *scp;
      /* ---- SILK (:2043-2261) ---- */
      {
         const int curr_bandwidth = sh->curr_bandwidth, redundancy = sh->f_redundancy, redundancy_bytes = sh->redundancy_bytes;
         sc.nChannelsAPI = CC; sc.nChannelsInternal = st->stream_channels; sc.API_sampleRate = Fs;
         const i32 total_bitRate = bits_to_bitrate(sh->bits_target, Fs, frame_size);
         sc.bitRate = total_bitRate;
         i32 HB_gain = Q15ONE;
         if (mode == OA_MODE_HYBRID) {                                                  /* :2052-2062 */
            sc.bitRate = sh_silk_rate_for_hybrid(total_bitRate, curr_bandwidth, Fs == 50 * frame_size, L->cfg.use_vbr, st->sm_LBRR_coded, st->stream_channels);
            if (!L->cfg.energy_mask_on) HB_gain = (i16)(Q15ONE - (fx_exp2((i16)(-(total_bitRate - sc.bitRate))) >> 1));   /* celt_exp2 takes an opus_val16 and HB_gain is one: both truncations as in the reference */
         }
         LANE0 { sh->HB_gain = HB_gain; }
         if (L->cfg.energy_mask_on && L->cfg.use_vbr && !L->cfg.lfe) {                 /* surround masking for SILK (:2069-2108) */
            i32 mask_sum = 0;
            int end = 17; i16 srate = 16000;
            if (st->bandwidth == OA_BW_NB) { end = 13; srate = 8000; } else if (st->bandwidth == OA_BW_MB) { end = 15; srate = 12000; }
            for (int c = 0; c < CC; c++) for (int i = 0; i < end; i++) {
               i32 mask = imax(imin(gs->energy_mask[21 * c + i], GC(.5f)), -GC(2.0f));
               if (mask > 0) mask = half32(mask);
               mask_sum += mask;
            }
            i32 masking_depth = mask_sum / end * CC;
            masking_depth += GC(.2f);
            i32 rate_offset = (i32)pshr32(mult16_16(srate, (i16)(masking_depth >> (DB_SHIFT - 10))), 10);
            rate_offset = imax(rate_offset, -2 * sc.bitRate / 3);
            if (st->bandwidth == OA_BW_SWB || st->bandwidth == OA_BW_FB) sc.bitRate += 3 * rate_offset / 5; else sc.bitRate += rate_offset;
         }
         LANE0 { sh->silk_bitRate = sc.bitRate; }
         sc.payloadSize_ms = 1000 * frame_size / Fs;
         sc.desiredInternalSampleRate = curr_bandwidth == OA_BW_NB ? 8000 : curr_bandwidth == OA_BW_MB ? 12000 : 16000;
         sc.minInternalSampleRate = mode == OA_MODE_HYBRID ? 16000 : 8000;
         sc.maxInternalSampleRate = 16000;
         if (mode == OA_MODE_SILK_ONLY) {
            i32 effective_max_rate = bits_to_bitrate(sh->f_max_data_bytes * 8, Fs, frame_size);
            if (frame_rate > 50) effective_max_rate = effective_max_rate * 2 / 3;
            if (effective_max_rate < 8000) { sc.maxInternalSampleRate = 12000; sc.desiredInternalSampleRate = imin(12000, sc.desiredInternalSampleRate); }
            if (effective_max_rate < 7000) { sc.maxInternalSampleRate = 8000; sc.desiredInternalSampleRate = imin(8000, sc.desiredInternalSampleRate); }
         }
         sc.packetLossPercentage = L->cfg.packet_loss_perc; sc.complexity = L->cfg.complexity; sc.useInBandFEC = L->cfg.use_inband_fec != 0; sc.LBRR_coded = st->sm_LBRR_coded; sc.useDTX = st->sm_useDTX;
         sc.useCBR = !L->cfg.use_vbr;
         sc.maxBits = (sh->f_max_data_bytes - 1) * 8;
         if (redundancy && redundancy_bytes >= 2) { sc.maxBits -= redundancy_bytes * 8 + 1; if (mode == OA_MODE_HYBRID) sc.maxBits -= 20; }     /* :2156 */
         if (sc.useCBR) {
            if (mode == OA_MODE_HYBRID) { const i16 other_bits = (i16)imax(0, sc.maxBits - sc.bitRate * frame_size / Fs); sc.maxBits = imax(0, sc.maxBits - other_bits * 3 / 4); sc.useCBR = 0; }
         } else if (mode == OA_MODE_HYBRID) {
            const i32 maxBitRate = sh_silk_rate_for_hybrid(sc.maxBits * Fs / frame_size, curr_bandwidth, Fs == 50 * frame_size, L->cfg.use_vbr, st->sm_LBRR_coded, st->stream_channels);
            sc.maxBits = bitrate_to_bits(maxBitRate, Fs, frame_size);
         }
         sc.toMono = st->sm_toMono; sc.opusCanSwitch = st->sm_opusCanSwitch; sc.reducedDependency = L->cfg.prediction_disabled;
         sc.internalSampleRate = 0; sc.allowBandwidthSwitch = 0; sc.inWBmodeWithoutVariableLP = 0; sc.stereoWidth_Q14 = 0; sc.switchReady = 0; sc.signalType = 0; sc.offset = 0;
      }
   }
}
/* sh_frame_back_wave: everything after silk_Encode (:2211-2657): what the Opus layer reads back from *scp, the CELT passes, redundancy, TOC, DTX, the frame's length.
 * silk_nBytes: what SILK returned (1 when it did not run) */
/* cut / resume: the kernel pipeline's cut of the frame's own CELT pass before its PVQ (sh_celt_run).  With a continuation record the function returns OA_CUT there (only a
 * frame with no other CELT pass is cut: no redundancy frame, no prefill -- those read the per-wave copy of the input); resume = 1 is the other half, entered with the
 * wave's LDS as the first half left it (oa_sh_back2_frame) and the PVQ coded: everything from celt_encode_with_ec's finalisation on. */
WV_DEV int sh_frame_back_wave(WV_LDS ShLds *L, OaShStream *gs, int frame_size, i16 *pcm_hp, i16 *pcm_celt, i16 *tmp_prefill, u8 *journal, const SeControl *scp, int silk_nBytes, const i32 *tr = nullptr,
      CeltCont *cut = nullptr, const int resume = 0)
{
   WV_LDS ShShared *sh = &L->sh; WV_LDS OaShScalars *st = &L->st;
   WV_LDS FrameLds *F = SH_F(L);
   const int CC = L->cfg.channels, Fs = L->cfg.Fs, application = L->cfg.application;
   const int delay_compensation = application == OA_APP_RESTRICTED_SILK ? 0 : Fs / 250, total_buffer = delay_compensation, encoder_buffer = Fs / 100;
   frame_size = wv_uni(frame_size); silk_nBytes = wv_uni(silk_nBytes);
   const int frame_rate = Fs / frame_size;
   const int mode = wv_uni(st->mode);
   if (resume) {}
   else if (mode != OA_MODE_CELT_ONLY) {
      const SeControl &sc = *scp;
      SE_PHASE(&L->S, 9);
      LANE0 {
         int curr_bandwidth = sh->curr_bandwidth;
         if (st->mode == OA_MODE_SILK_ONLY) { if (sc.internalSampleRate == 8000) curr_bandwidth = OA_BW_NB; else if (sc.internalSampleRate == 12000) curr_bandwidth = OA_BW_MB; else if (sc.internalSampleRate == 16000) curr_bandwidth = OA_BW_WB; }
         sh->curr_bandwidth = curr_bandwidth;
         st->sm_allowBandwidthSwitch = sc.allowBandwidthSwitch; st->sm_inWBmodeWithoutVariableLP = sc.inWBmodeWithoutVariableLP; st->sm_switchReady = sc.switchReady;
         st->sm_opusCanSwitch = sc.switchReady && !st->nonfinal_frame;
         st->sm_stereoWidth_Q14 = sc.stereoWidth_Q14;
         sh->silk_signalType = sc.signalType; sh->silk_offset = sc.offset;
         if (sh->activity == SE_VAD_NO_DECISION) sh->activity = sc.signalType != SE_TYPE_NO_VOICE;
         if (silk_nBytes != 0 && st->sm_opusCanSwitch) {                                 /* SILK is ready to change its bandwidth: announce it with a redundant frame (:2251-2260) */
            if (application != OA_APP_RESTRICTED_SILK) {
               sh->redundancy_bytes = sh_redundancy_bytes(sh->f_max_data_bytes, sh->bitrate_bps, frame_rate, st->stream_channels);
               sh->f_redundancy = sh->redundancy_bytes != 0;
            }
            sh->f_celt_to_silk = 0;
            st->silk_bw_switch = 1;
         }
      }
      if (silk_nBytes == 0) {                                                             /* SILK DTX (:2242): the TOC alone, no bookkeeping */
         LANE0 { st->rangeFinal = 0; SH_PKT(L)[0] = sh_gen_toc(st->mode, Fs / frame_size, sh->curr_bandwidth, st->stream_channels); sh->r[3] = 1; }
         wv_sync();
         return 1;
      }
   } else { LANE0 sh->HB_gain = Q15ONE; }
   wv_sync();
   /* ---- CELT control, delay line, fades (:2264-2349) ---- */
   const int need_celt = mode != OA_MODE_SILK_ONLY || wv_uni(sh->f_redundancy);
   const int celt_prefill = mode != OA_MODE_SILK_ONLY && mode != wv_uni(st->prev_mode) && wv_uni(st->prev_mode) > 0;
   if (!resume) {
   if (celt_prefill) {                                                                    /* tmp_prefill: the 2.5 ms ahead of the delay-compensated input (:2298-2302) */
      const int n4 = Fs / 400;
      FOR_LANES(i, n4 * CC) tmp_prefill[i] = gs->delay_buffer[(encoder_buffer - total_buffer - n4) * CC + i];
      wv_sync();
   }
   if (need_celt) sh_enter_celt(L, gs);
   LANE0 {
      if (mode != OA_MODE_SILK_ONLY) { SH_CELT_DISABLE_PF(F) = 0; SH_CELT_FORCE_INTRA(F) = 0; if (L->cfg.prediction_disabled) { SH_CELT_DISABLE_PF(F) = 1; SH_CELT_FORCE_INTRA(F) = 1; } }   /* CELT_SET_PREDICTION(2 or 0) :2288-2295 */
      /* stereo width (:2320-2328) and fade gains */
      if (st->mode != OA_MODE_HYBRID || st->stream_channels == 1) {
         if (sh->equiv_rate > 32000) st->sm_stereoWidth_Q14 = 16384; else if (sh->equiv_rate < 16000) st->sm_stereoWidth_Q14 = 0;
         else st->sm_stereoWidth_Q14 = 16384 - 2048 * (i32)(32000 - sh->equiv_rate) / (sh->equiv_rate - 14000);
      }
      sh->do_gain_fade = (st->prev_HB_gain < Q15ONE || sh->HB_gain < Q15ONE) && application != OA_APP_RESTRICTED_SILK;
      sh->hb_g1 = st->prev_HB_gain; sh->hb_g2 = sh->HB_gain;
      st->prev_HB_gain = sh->HB_gain;
      sh->do_stereo_fade = 0;
      if (!L->cfg.energy_mask_on && CC == 2 && (st->hybrid_stereo_width_Q14 < (1 << 14) || st->sm_stereoWidth_Q14 < (1 << 14))) {
         i16 g1 = (i16)st->hybrid_stereo_width_Q14, g2 = (i16)st->sm_stereoWidth_Q14;
         sh->do_stereo_fade = application != OA_APP_RESTRICTED_SILK; sh->fade_g1 = g1 == 16384 ? Q15ONE : shl16(g1, 1); sh->fade_g2 = g2 == 16384 ? Q15ONE : shl16(g2, 1);
         st->hybrid_stereo_width_Q14 = st->sm_stereoWidth_Q14;
      }
   }
   /* pcm_buf = [delay tail | this frame] -> the CELT staging area (only when a CELT pass will read it), then the delay line moves on (:2304-2312) */
   if (need_celt) {
      i16 *io = L->cs->pcm16;
      for (int i0 = wv_lane(); i0 < frame_size * CC; i0 += 8 * WV_WIDTH) {                 /* (eight trips' samples in flight) */
         i16 v[8];
#pragma unroll
         for (int u = 0; u < 8; u++) { const int i = imin(i0 + u * WV_WIDTH, frame_size * CC - 1), n = i / CC, c = i - n * CC; v[u] = n < total_buffer ? gs->delay_buffer[(encoder_buffer - total_buffer + n) * CC + c] : pcm_hp[(n - total_buffer) * CC + c]; }
#pragma unroll
         for (int u = 0; u < 8; u++) { const int i = i0 + u * WV_WIDTH; if (i < frame_size * CC) io[i] = v[u]; }
      }
      wv_sync();
   }
   if (application != OA_APP_RESTRICTED_SILK) {
      /* new delay line = the last encoder_buffer samples of [old line | this frame]; ascending in place through registers, eight 64-lane trips at a time: a batch reads
       * [b0 + F, b0 + F + 512) before it writes [b0, b0 + 512), and no later batch reads below b0 + 512 + F -- nothing is read after it was written */
      const int EB = encoder_buffer * CC, F = frame_size * CC;
      for (int b0 = 0; b0 < EB; b0 += 8 * WV_WIDTH) {
         i16 v[8];
#pragma unroll
         for (int u = 0; u < 8; u++) {
            const int i = b0 + u * WV_WIDTH + wv_lane(), j = i + F;
            v[u] = 0;
            if (i < EB) v[u] = j < EB ? gs->delay_buffer[j] : pcm_hp[j - EB];
         }
         wv_sync();
#pragma unroll
         for (int u = 0; u < 8; u++) { const int i = b0 + u * WV_WIDTH + wv_lane(); if (i < EB) gs->delay_buffer[i] = v[u]; }
         wv_sync();
      }
   }
   if (need_celt) {
      i16 *io = L->cs->pcm16;
      if (sh->do_gain_fade) { sh_gain_fade_lds(io, frame_size, CC, (i16)sh->hb_g1, (i16)sh->hb_g2, Fs); wv_sync(); }
      if (sh->do_stereo_fade) { sh_stereo_fade_lds(io, frame_size, (i16)sh->fade_g1, (i16)sh->fade_g2, Fs); wv_sync(); }
      /* more than one CELT pass reads pcm_buf: keep it in the HBM scratch */
      if (wv_uni(sh->f_redundancy) || celt_prefill) { FOR_LANES(i, frame_size * CC) pcm_celt[i] = io[i]; wv_sync(); }
   }
   /* ---- redundancy signalling, end of the SILK layer (:2351-2414) ---- */
   LANE0 {
      EcCtx e_; ec_ld(&e_, &L->ec); EcCtx *e = &e_; WV_LDS u8 *buf = SH_PKT(L) + 1;
      int redundancy = sh->f_redundancy, redundancy_bytes = sh->redundancy_bytes;
      const int max_data_bytes = sh->f_max_data_bytes;
      if (st->mode != OA_MODE_CELT_ONLY && k_ec_tell(EC_PASS) + 17 + 20 * (st->mode == OA_MODE_HYBRID) <= 8 * (max_data_bytes - 1)) {
         if (st->mode == OA_MODE_HYBRID) k_ec_enc_bit_logp(EC_PASS, redundancy, 12);
         if (redundancy) {
            int max_redundancy;
            k_ec_enc_bit_logp(EC_PASS, sh->f_celt_to_silk, 1);
            if (st->mode == OA_MODE_HYBRID) max_redundancy = (max_data_bytes - 1) - ((k_ec_tell(EC_PASS) + 8 + 3 + 7) >> 3);
            else max_redundancy = (max_data_bytes - 1) - ((k_ec_tell(EC_PASS) + 7) >> 3);
            redundancy_bytes = imin(max_redundancy, redundancy_bytes);
            redundancy_bytes = imin(257, imax(2, redundancy_bytes));
            if (st->mode == OA_MODE_HYBRID) k_ec_enc_uint(EC_PASS, redundancy_bytes - 2, 256);
         }
      } else redundancy = 0;
      if (!redundancy) { st->silk_bw_switch = 0; redundancy_bytes = 0; }
      sh->start_band = st->mode != OA_MODE_CELT_ONLY ? 17 : 0;
      if (st->mode == OA_MODE_SILK_ONLY) {
         sh->ret = (k_ec_tell(EC_PASS) + 7) >> 3;
         sh->r[7] = k_ec_tell(EC_PASS);
         st->rangeFinal = e->rng;
         k_ec_enc_done(EC_PASS);
         sh->nb_compr_bytes = sh->ret;
      } else {
         sh->nb_compr_bytes = (max_data_bytes - 1) - redundancy_bytes;
         if (st->mode == OA_MODE_HYBRID) k_ec_enc_shrink(EC_PASS, (u32)sh->nb_compr_bytes);   /* (a CELT-only frame starts its coder in the prologue, shrunk there) */
         sh->r[7] = k_ec_tell(EC_PASS);
      }
      ec_st(&L->ec, e);
      sh->f_redundancy = redundancy; sh->redundancy_bytes = redundancy_bytes;
   }
   }   /* !resume */
   const int redundancy = wv_uni(sh->f_redundancy), celt_to_silk = wv_uni(sh->f_celt_to_silk), redundancy_bytes = wv_uni(sh->redundancy_bytes);
   const int n2 = Fs / 200, n4 = Fs / 400;
   if (resume) {}
   else if (redundancy || mode != OA_MODE_SILK_ONLY) {                                          /* CELT_SET_ANALYSIS (:2416-2419) */
      if (wv_lane() < (int)(sizeof(OaAnalysisInfo) / 4)) ((i32 *)&gs->celt.analysis)[wv_lane()] = ((const i32 *)&gs->an_info)[wv_lane()];
      wv_sync();
   }
   /* ---- 5 ms redundant CELT frame ahead of the SILK / hybrid audio (CELT -> SILK, :2427-2442) ---- */
   if (!resume && redundancy && celt_to_silk) {
      ShCeltCtl c; c.start = 0; c.vbr = 0; c.constrained_vbr = 0; c.nbytes = redundancy_bytes; c.raw = 1; c.cont = 0; c.bitrate = -1;
      sh_celt_run(L, gs, pcm_celt, n2, c, journal);
      if (wv_uni(sh->celt_ret) < 0) return OA_ERR_INTERNAL;
      wv_sync();
      FOR_LANES(i, redundancy_bytes) SH_PKT(L)[1 + sh->nb_compr_bytes + i] = F->packet[1 + i];
      LANE0 sh->redundant_rng = F->st.rangeFinal;
      sh_celt_reset_wave(L, gs);
   }
   /* ---- the CELT layer of the frame (:2447-2512) ---- */
   if (mode != OA_MODE_SILK_ONLY) {
      const int hyb = mode == OA_MODE_HYBRID;
      ShCeltCtl c; c.start = hyb ? 17 : 0; c.vbr = L->cfg.use_vbr; c.constrained_vbr = hyb ? 0 : L->cfg.vbr_constraint; c.nbytes = 0; c.raw = 0; c.cont = hyb; c.bitrate = -1;
      if (L->cfg.use_vbr) { const i32 cb = hyb ? sh->bitrate_bps - sh->silk_bitRate : sh->bitrate_bps; if (cb > 500) c.bitrate = imin(cb, 750000 * CC); }    /* OPUS_SET_BITRATE rejects <= 500 and keeps OPUS_BITRATE_MAX */
      if (!resume && celt_prefill) {                                                      /* mode change: restart CELT on the 2.5 ms before the frame, then no inter-frame prediction (:2478-2486) */
         sh_celt_reset_wave(L, gs);
         ShCeltCtl p = c; p.raw = 1; p.nbytes = 2; p.cont = 0;
         sh_celt_run(L, gs, tmp_prefill, n4, p, journal);
         LANE0 { SH_CELT_DISABLE_PF(F) = 1; SH_CELT_FORCE_INTRA(F) = 1; }
      }
      int ran = 0;
      if (resume || wv_uni(sh->r[7]) <= 8 * wv_uni(sh->nb_compr_bytes)) {                  /* otherwise the budget is gone already and the frame ends up a "PLC frame" (:2487) */
         const int reload = wv_uni(sh->f_redundancy) || celt_prefill;
         if (resume) { if (hyb) celt_encode_core_tail<true>(F, &gs->celt); else celt_encode_core_tail<false>(F, &gs->celt); sh_celt_run_tail(L); }
         else if (sh_celt_run(L, gs, reload ? pcm_celt : (const i16 *)0, frame_size, c, journal, tr, reload ? (CeltCont *)0 : cut) == OA_CUT) return OA_CUT;
         const int cret = wv_uni(sh->celt_ret);
         if (cret != -1000 && cret < 0) return OA_ERR_INTERNAL;
         if (cret >= 0) {
            ran = 1;
            LANE0 sh->ret = cret;
            wv_sync();
            FOR_LANES(i, (OA_MAX_PACKET + 4) / 4) ((WV_LDS i32 *)SH_PKT(L))[i] = ((const WV_LDS i32 *)F->packet)[i];    /* the frame's bytes return to the packet buffer */
            wv_sync();
            if (redundancy && celt_to_silk && hyb && wv_uni(sh->nb_compr_bytes) != cret) {      /* the redundant frame follows the bytes CELT really used (:2503-2507) */
               const int from = wv_uni(sh->nb_compr_bytes);
               for (int b0 = 0; b0 < redundancy_bytes; b0 += WV_WIDTH) {
                  const int i = b0 + wv_lane(); u8 v = 0;
                  if (i < redundancy_bytes) v = SH_PKT(L)[1 + from + i];
                  wv_sync();
                  if (i < redundancy_bytes) SH_PKT(L)[1 + cret + i] = v;
                  wv_sync();
               }
               LANE0 sh->nb_compr_bytes = cret + redundancy_bytes;
            }
         }
      }
      LANE0 { if (ran) st->rangeFinal = F->st.rangeFinal; else { sh->ret = 0; sh->r[4] = sh->r[7]; st->rangeFinal = F->st.rng; } }   /* OPUS_GET_FINAL_RANGE of the CELT encoder (:2509) */
      wv_sync();
   }
   /* ---- 5 ms redundant CELT frame after the SILK / hybrid audio (SILK -> CELT, :2514-2545) ---- */
   if (redundancy && !celt_to_silk) {
      sh_celt_reset_wave(L, gs);
      LANE0 { SH_CELT_DISABLE_PF(F) = 1; SH_CELT_FORCE_INTRA(F) = 1; if (st->mode == OA_MODE_HYBRID) sh->nb_compr_bytes = sh->ret; }      /* hybrid: the packet shrinks to what the coder used */
      ShCeltCtl c; c.start = 0; c.vbr = 0; c.constrained_vbr = 0; c.nbytes = 2; c.raw = 1; c.cont = 0; c.bitrate = -1;
      sh_celt_run(L, gs, pcm_celt + CC * (frame_size - n2 - n4), n4, c, journal);
      c.nbytes = redundancy_bytes;
      sh_celt_run(L, gs, pcm_celt + CC * (frame_size - n2), n2, c, journal);
      if (wv_uni(sh->celt_ret) < 0) return OA_ERR_INTERNAL;
      wv_sync();
      FOR_LANES(i, redundancy_bytes) SH_PKT(L)[1 + sh->nb_compr_bytes + i] = F->packet[1 + i];
      LANE0 sh->redundant_rng = F->st.rangeFinal;
      wv_sync();
   }
   if (need_celt) sh_leave_celt(L, gs);
   /* ---- TOC, bookkeeping, DTX, busted budget (:2549-2601) ---- */
   LANE0 {
      const int curr_bandwidth = sh->curr_bandwidth;
      SH_PKT(L)[0] = sh_gen_toc(st->mode, Fs / frame_size, curr_bandwidth, st->stream_channels);
      st->rangeFinal ^= (u32)sh->redundant_rng;
      st->prev_mode = sh->f_to_celt ? OA_MODE_CELT_ONLY : st->mode;
      st->prev_channels = st->stream_channels; st->prev_framesize = frame_size; st->first = 0;
      int ret = sh->ret;
      if (sh_generalised_dtx_l0(L, frame_size, Fs)) { st->rangeFinal = 0; ret = 1; sh->r[3] = 1; }
      else {
         const int busted = (st->mode == OA_MODE_SILK_ONLY ? sh->r[7] : sh->r[4]) > (sh->f_max_data_bytes - 1) * 8;
         if (busted) {
            if (sh->f_max_data_bytes < 2) ret = OA_ERR_BUFFER_TOO_SMALL;
            else { SH_PKT(L)[1] = 0; ret = 1; st->rangeFinal = 0; }
         } else if (st->mode == OA_MODE_SILK_ONLY && !sh->f_redundancy) while (ret > 2 && SH_PKT(L)[ret] == 0) ret--;     /* trailing zeros are implied in SILK-only packets (:2590) */
         if (ret >= 0) ret += 1 + sh->redundancy_bytes;
      }
      sh->ret = ret;
   }
   wv_sync();
   return wv_uni(sh->ret);
}
WV_DEVN int sh_encode_frame_native(WV_LDS ShLds *L, OaShStream *gs, const i16 *pcm, int frame_size, int orig_max_data_bytes, i16 *pcm_hp, i16 *pcm_celt, i16 *tmp_prefill,
      SeRateScratch *G, u8 *journal)
{
   WV_LDS ShShared *sh = &L->sh;
   const int CC = L->cfg.channels, Fs = L->cfg.Fs, application = L->cfg.application;
   const int delay_compensation = application == OA_APP_RESTRICTED_SILK ? 0 : Fs / 250, encoder_buffer = Fs / 100;
   frame_size = wv_uni(frame_size);
   SeControl sc;
   int silk_nBytes = 1;
   sh_frame_front_wave(L, gs, pcm, frame_size, orig_max_data_bytes, pcm_hp, &sc);
   if (wv_uni(L->st.mode) != OA_MODE_CELT_ONLY) {
      if (wv_uni(sh->f_prefill) && application != OA_APP_RESTRICTED_SILK) {
         /* smooth onset for the SILK prefill (:2191-2209): fade the delay line in over 2.5 ms, silence before it, feed its 10 ms to SILK with nothing coded */
         const int prefill_offset = CC * (encoder_buffer - delay_compensation - Fs / 400), n4 = Fs / 400;
         WV_LDS i16 *stage = L->S.u.pcm_stage;
         wv_sync();
         FOR_LANES(i, n4 * CC) stage[i] = gs->delay_buffer[prefill_offset + i];
         wv_sync();
         sh_gain_fade_lds(stage, n4, CC, 0, Q15ONE, Fs);
         wv_sync();
         FOR_LANES(i, n4 * CC) gs->delay_buffer[prefill_offset + i] = stage[i];
         FOR_LANES(i, prefill_offset) gs->delay_buffer[i] = 0;
         wv_sync();
         const int pr = silk_encode_wave(&L->S, &sc, gs->delay_buffer, encoder_buffer, &L->ec, SH_PKT(L) + 1, sh->activity, G, &gs->lbrr, wv_uni(sh->f_prefill));
         wv_sync();
         if (pr) { LANE0 { gs->s.error = pr; } return OA_ERR_INTERNAL; }
         sc.opusCanSwitch = 0;                                                              /* no second switch in the real call */
      }
      const int sret = silk_encode_wave(&L->S, &sc, pcm_hp, frame_size, &L->ec, SH_PKT(L) + 1, sh->activity, G, &gs->lbrr);
      wv_sync();
      if (sret) { LANE0 { gs->s.error = sret; } return OA_ERR_INTERNAL; }
      silk_nBytes = wv_uni(L->S.r[0]);
   }
   return sh_frame_back_wave(L, gs, frame_size, pcm_hp, pcm_celt, tmp_prefill, journal, &sc, silk_nBytes);
}

/* silk_InitEncoder (silk/enc_API.c:82) on the state staged in LDS: everything, both channels */
WV_DEV void sh_silk_init_wave(WV_LDS ShLds *L)
{
   const int CC = L->cfg.channels;
   wv_sync();
   WV_LDS i32 *w = (WV_LDS i32 *)se_st(&L->S);
   FOR_LANES(i, SE_STATE_LITE_WORDS(CC)) w[i] = 0;
   /* the channels' input buffers and the quantiser tails belong to the state silk_InitEncoder clears (silk_encoder_state.inputBuf, silk_nsq_state: silk/structs.h:176, :56): stale
    * input would be read by the first mono frame after a stereo one (enc_API.c:318-326 averages frame_length samples of channel 1's buffer, of which its resampler -- still at the
    * old internal rate when the rate switches with the channel count -- may have written fewer).  Only kernels that stage them run this (the pipeline's front kernel turns prefill calls away) */
   if (wv_uni(L->silk_tail)) {
      const int o = (int)(offsetof(OaSilkEnc, tail) / 4), b = (int)(offsetof(OaSilkEnc, inbuf) / 4);
      FOR_LANES(i, CC * SE_TAIL_WORDS) w[o + i] = 0;
      FOR_LANES(i, CC * SE_INBUF_WORDS) w[b + i] = 0;
   }
   wv_sync();
   LANE0 {
      WV_LDS OaSilkEnc *E = se_st(&L->S);
      se_init_channel(&E->ch[0]); if (CC == 2) se_init_channel(&E->ch[1]);
      E->nChannelsAPI = 1; E->nChannelsInternal = 1;
   }
}

/* The top of opus_encode_native (:1182-1696) for one call: configuration, Opus-layer scalars and SILK state HBM -> LDS, the tonality analysis of the call's input, digital
 * silence / peak energy / stereo width, the call's decisions (sh_layer_decide).  analysed = 1: the analysis of this call's input has run already (the split path's front
 * kernel ran this very function on the stream and then handed the call to the one-kernel path: the analysis state in HBM is the only thing it changed) */
/* the tonality / music analysis of a call's input (src/opus_encoder.c:1247-1264; the FIXED_POINT build runs it at complexity 10 only); a call the reference turns away before
 * that (:1231) leaves it alone.  A: 6 KB of LDS (AnLds); gscratch: 1,920 words of the wave's HBM scratch.  (As a kernel of its own ahead of the front kernel -- 64 VGPRs, 25 waves
 * per CU -- it took the 1.9 ms it saved there: profiles/r05_o; it stays part of the call's opening.) */
WV_DEV void sh_call_analysis_wave(WV_LDS AnLds *A, OaShStream *gs, int analysis_off, int complexity, int application, int input_depth, int lsb_depth, int CC, int Fs,
      const i16 *pcm, const i32 *apcm, int frame_size, int analysis_frame_size, int max_data_bytes, i32 *gscratch)
{
   if (imin(1276 * 6, max_data_bytes) == 1 && Fs == frame_size * 10) return;
   if (!analysis_off && complexity >= 10 && Fs >= 16000 && application != OA_APP_RESTRICTED_SILK) {
      LANE0 { gs->an_read_pos_bak = gs->an.read_pos; gs->an_read_subframe_bak = gs->an.read_subframe; }
      an_run_analysis_wave(A, &gs->an, pcm, apcm, analysis_frame_size > frame_size ? analysis_frame_size : frame_size, frame_size, CC, Fs, imin(input_depth ? input_depth : 16, lsb_depth), gscratch, &gs->an_info);
   } else {
      const int was_initialized = wv_uni(gs->an.initialized);
      wv_sync();                                                                       /* (every lane has read the flag before any lane clears it) */
      if (was_initialized) { i32 *z = (i32 *)&gs->an; FOR_LANES(i, (int)(sizeof(OaAnalysis) / 4)) z[i] = 0; }                /* tonality_analysis_reset (:1262) */
      LANE0 { gs->an_info.valid = 0; gs->an_read_pos_bak = -1; }
   }
}
WV_DEV void sh_call_open_wave(WV_LDS ShLds *L, OaShStream *gs, const i16 *pcm, int frame_size, int max_data_bytes, CeltScratch *cs, const i32 *apcm, int analysed, int analysis_frame_size = 0 /* samples per channel behind pcm: the caller's look-ahead (src/opus_encoder.c:1247, :2662-2690); 0 = frame_size */)
{
   WV_LDS ShShared *sh = &L->sh; WV_LDS OaShScalars *st = &L->st;
   SE_PHASE_START(&L->S);
   /* ---- load configuration, Opus-layer scalars and the SILK encoder state (coalesced) ---- */
   {
      const i32 *g = (const i32 *)&gs->cfg; WV_LDS i32 *d = (WV_LDS i32 *)&L->cfg;
      FOR_LANES(i, (int)(sizeof(OaShConfig) / 4)) d[i] = g[i];
      g = (const i32 *)&gs->s; d = (WV_LDS i32 *)st;
      FOR_LANES(i, (int)(sizeof(OaShScalars) / 4)) d[i] = g[i];
   }
   wv_sync();
   const int CC = L->cfg.channels, Fs = L->cfg.Fs;
   LANE0 {
      L->cs = cs;
      if (L->cfg.voice_ratio_seq != st->voice_ratio_seq) { st->voice_ratio = L->cfg.voice_ratio; st->voice_ratio_seq = L->cfg.voice_ratio_seq; }   /* OPUS_SET_VOICE_RATIO through a batch ctl */
   }
   /* the tonality / music analysis of the call's input (:1247-1264; the FIXED_POINT build runs it at complexity 10 only), in the SILK encoder's phase union (nothing in it
    * outlives a stage; the words in front of it -- among them where this kernel keeps the staged state, SilkEncLds.st_off -- stay); a call the reference turns away before
    * that (:1231) leaves it alone */
   SE_CLK_BEGIN();
   if (!analysed) sh_call_analysis_wave((WV_LDS AnLds *)&L->S.u, gs, wv_uni(L->cfg.analysis_off), wv_uni(L->cfg.complexity), wv_uni(L->cfg.application), wv_uni(L->cfg.input_depth), wv_uni(L->cfg.lsb_depth), CC, Fs,
         pcm, apcm, frame_size, analysis_frame_size, max_data_bytes, (i32 *)cs->X);
   wv_sync();
   SE_CLK_END(23);
   SE_PHASE_START(&L->S);                                                               /* (profiling build: the analysis borrowed the arena the phase clock lives in) */
   {  /* the SILK encoder state (coalesced) */
      se_state_copy_wave((WV_LDS i32 *)se_st(&L->S), (const i32 *)&gs->silk, CC, wv_uni(L->silk_tail));
   }
   wv_sync();
   LANE0 sh->silk_in_lds = 1;
   SE_PHASE(&L->S, 0);
   {  /* is_digital_silence (:1060, fixed point: all samples zero); peak signal energy tracker (:1310-1320: frames the analysis calls inactive do not feed it) */
      const i32 m = sh_maxabs_wave(pcm, frame_size * CC);
      i32 en = 0;
      const int track = m != 0 && (!wv_uni(gs->an_info.valid) || an_activity_prob_above(&gs->an_info));
      if (track) en = sh_frame_energy_wave(pcm, frame_size * CC);
      LANE0 { sh->sample_max = m; sh->is_silence = m == 0; if (track) st->peak_signal_energy = imax(mult16_32_q15(QC16(0.999f, 15), st->peak_signal_energy), en); }
   }
   if (CC == 2 && SH_FORCE_CHANNELS(&L->cfg, st) != 1) sh_compute_stereo_width_wave(L, pcm, frame_size); else { LANE0 sh->stereo_width = 0; }
   LANE0 sh_layer_decide(L, frame_size, max_data_bytes, &gs->an_info);
}
WV_DEV void oa_sh_encode_frame(WV_LDS ShLds *L, OaShStream *gs, const i16 *pcm, int frame_size, int max_data_bytes, u8 *out, int out_cap, i16 *pcm_hp, SeRateScratch *G, CeltScratch *cs, i32 *len_out, u32 *rng_out,
      const i32 *apcm = nullptr, int analysed = 0, int analysis_frame_size = 0)
{
   WV_LDS ShShared *sh = &L->sh; WV_LDS OaShScalars *st = &L->st;
   sh_call_open_wave(L, gs, pcm, frame_size, max_data_bytes, cs, apcm, analysed, analysis_frame_size);
   const int CC = L->cfg.channels, Fs = L->cfg.Fs;
   i16 *pcm_celt = (i16 *)((char *)pcm_hp + SH_PCM_BYTES(frame_size, CC)), *tmp_prefill = (i16 *)((char *)pcm_hp + 2 * SH_PCM_BYTES(frame_size, CC));
   if (sh->err) { LANE0 { *len_out = sh->err; *rng_out = 0; gs->s.error = sh->err; } return; }
   if (sh->plc_frame) {
      const int n = sh_emit_packet(SH_PKT(L), out, sh->ret, L->cfg.use_vbr ? 0 : sh->max_data_bytes, out_cap);
      /* what opus_encode_native has updated by the time it emits a 'PLC frame' (:1345) stays updated: the voice ratio (:1273-1292), the peak signal energy (:1310-1320), the
       * stereo-width memory (:1322), besides the analysis (in HBM already) */
      LANE0 {
         *len_out = n; *rng_out = 0; gs->s.rangeFinal = 0;
         gs->s.voice_ratio = st->voice_ratio; gs->s.voice_ratio_seq = st->voice_ratio_seq; gs->s.peak_signal_energy = st->peak_signal_energy;
         gs->s.wm_XX = st->wm_XX; gs->s.wm_XY = st->wm_XY; gs->s.wm_YY = st->wm_YY; gs->s.wm_smoothed_width = st->wm_smoothed_width; gs->s.wm_max_follower = st->wm_max_follower;
      }
      return;
   }
   if (wv_uni(sh->prefill)) sh_silk_init_wave(L);                                      /* CELT -> SILK: the SILK encoder starts over (:1576-1581) */
   int result;
   if (wv_uni(sh->nb_frames) == 1) {
      LANE0 { sh->f_redundancy = sh->redundancy; sh->f_celt_to_silk = sh->celt_to_silk; sh->f_prefill = sh->prefill; sh->f_to_celt = sh->to_celt; sh->f_silence = sh->is_silence; st->nonfinal_frame = 0; }
      const int ret = sh_encode_frame_native(L, gs, pcm, frame_size, wv_uni(sh->max_data_bytes), pcm_hp, pcm_celt, tmp_prefill, G, out);
      /* apply_padding (:2646): hard CBR pads every packet, except the bare TOC of a DTX frame, to the byte budget */
      const int pad_to = (!L->cfg.use_vbr && ret > 0 && !wv_uni(sh->r[3])) ? wv_uni(sh->max_data_bytes) : 0;
      result = ret < 0 ? ret : sh_emit_packet(SH_PKT(L), out, ret, pad_to, out_cap);
   } else {
      /* ---- several coded frames, one packet (:1757-1838) ---- */
      const int nb_frames = wv_uni(sh->nb_frames), enc_frame_size = wv_uni(sh->enc_frame_size), max_len_sum = wv_uni(sh->max_len_sum), repacketize_len = wv_uni(sh->repacketize_len);
      const int bak_to_mono = wv_uni(st->sm_toMono);
      LANE0 { if (bak_to_mono) st->mono_forced_seq = L->cfg.force_channels_seq + 1; else st->prev_channels = st->stream_channels; L->mf.n = nb_frames; }   /* (:1764: force_channels = 1, for good) */
      int tot_size = 0, dtx_count = 0, err = 0, staged = 0;
      if (OA_MF_HEADROOM + imin(max_len_sum, 1276 * nb_frames) > out_cap) err = OA_ERR_BUFFER_TOO_SMALL;     /* staging + theta-RDO journal need the slot the host promised */
      const int an_bak = wv_uni(gs->an_read_pos_bak);
      if (an_bak != -1) { LANE0 { gs->an.read_pos = an_bak; gs->an.read_subframe = gs->an_read_subframe_bak; } }   /* the analysis is read one coded frame at a time (:1727-1735) */
      for (int i = 0; i < nb_frames && !err; i++) {
         const i16 *fp = pcm + (size_t)i * CC * enc_frame_size;
         if (an_bak != -1) {                                                              /* (:1796-1800); the arena is borrowed: the SILK state goes back to HBM first */
            sh_park_silk(L, gs);
            an_get_info_wave((WV_LDS AnLds *)&L->S.u, &gs->an, &gs->an_info, enc_frame_size, Fs);
         }
         const i32 fm = sh_maxabs_wave(fp, enc_frame_size * CC);
         int curr_max = imin(bitrate_to_bits(wv_uni(sh->bitrate_bps), Fs, enc_frame_size) / 8, max_len_sum / nb_frames);
         curr_max = imin(max_len_sum - tot_size, curr_max);
         LANE0 {
            st->sm_toMono = 0; st->nonfinal_frame = i < nb_frames - 1;
            const int frame_to_celt = sh->to_celt && i == nb_frames - 1;
            sh->f_to_celt = frame_to_celt; sh->f_redundancy = sh->redundancy && (frame_to_celt || (!sh->to_celt && i == 0));
            sh->f_celt_to_silk = sh->celt_to_silk; sh->f_prefill = sh->prefill; sh->f_silence = fm == 0;
         }
         sh_reload_silk(L, gs);
         const int tmp_len = sh_encode_frame_native(L, gs, fp, enc_frame_size, curr_max, pcm_hp, pcm_celt, tmp_prefill, G, out + OA_MF_HEADROOM + staged);
         if (tmp_len < 0) { err = OA_ERR_INTERNAL; break; }
         if (tmp_len == 1) dtx_count++;
         wv_sync();
         if (i > 0 && ((wv_uni(L->mf.toc) ^ SH_PKT(L)[0]) & 0xFC)) { err = OA_ERR_INTERNAL; break; }   /* opus_repacketizer_cat refuses frames of another configuration */
         LANE0 { if (i == 0) L->mf.toc = SH_PKT(L)[0]; L->mf.len[i] = tmp_len - 1; }
         FOR_LANES(k, tmp_len - 1) out[OA_MF_HEADROOM + staged + k] = SH_PKT(L)[1 + k];
         wv_sync();
         staged += tmp_len - 1; tot_size += tmp_len;
      }
      LANE0 st->sm_toMono = bak_to_mono;
      wv_sync();
      if (err) result = err;
      else {
         result = oa_multiframe_assemble_wave(&L->mf, SH_PKT(L), out, repacketize_len, !L->cfg.use_vbr && dtx_count != nb_frames, out_cap);
         if (result < 0) result = OA_ERR_INTERNAL;
      }
   }
   /* ---- store lengths + state (coalesced) ---- */
   {
      LANE0 { *len_out = result; *rng_out = result < 0 ? 0 : st->rangeFinal; }
      i32 *g = (i32 *)&gs->s; const WV_LDS i32 *d = (const WV_LDS i32 *)st;
      FOR_LANES(i, (int)(sizeof(OaShScalars) / 4)) g[i] = d[i];
      if (wv_uni(sh->silk_in_lds)) {
         se_state_copy_wave((i32 *)&gs->silk, (const WV_LDS i32 *)se_st(&L->S), CC, wv_uni(L->silk_tail));
      }
   }
   SE_PHASE(&L->S, 10);
}
#endif

/* celt_enc_front.h — Opus-layer front end and CELT time-domain analysis, wave-parallel where the math allows.
 *
 *   opus_layer_decide   lane 0    src/opus_encoder.c:1182-1700 (the decisions left when the application pins CELT-only)
 *   dc_reject           lane/ch   src/opus_encoder.c:479 (1st-order recursion with rounding: serial per channel)
 *   stereo_fade         parallel  src/opus_encoder.c:548
 *   celt_prologue       lane 0    celt/celt_encoder.c:1902-2008 (rate/VBR bounds, silence flag)
 *   preemphasis         parallel  celt/celt_encoder.c:557 (an FIR on the input: y[i] = x[i] - .85 x[i-1])
 *   tone_detect         reductions + scalar tail   celt/celt_encoder.c:1272-1403
 *   transient_analysis  lane/ch recursions + parallel normalisation   celt/celt_encoder.c:267
 *   run_prefilter       parallel xcorr / comb FIR, lane-0 decisions   celt/celt_encoder.c:1405, celt/pitch.c, celt/celt.c:166-312
 */
#ifndef OPUS_AMD_CELT_ENC_FRONT_H
#define OPUS_AMD_CELT_ENC_FRONT_H

/* lane-0 serial section, fenced on both sides: other lanes neither race ahead of its inputs nor read its
 * outputs early (on the GPU the fences are LDS waits; the CPU emulator needs them for fiber ordering) */
#ifndef LANE0
#define LANE0 for (int l0_ = (wv_sync(), wv_prio_serial(), 1); l0_; l0_ = (wv_prio_normal(), wv_sync(), 0)) if (wv_lane() == 0)
#define FOR_LANES(i, n) for (int i = wv_lane(); i < (n); i += WV_WIDTH)
#endif
/* d[0 .. n) = s[0 .. n), lane-strided, eight trips' loads in flight before the first store: a trip of a plain lane loop over HBM is a round trip (the store of one trip
 * keeps the next trip's load behind it), and a wave has four or five of those per stage to hide -- measured on the front kernels (profiles/r06_aq, r06_ar) */
template <class PD, class PS> WV_DEV void wv_copy_batched(PD d, PS s, int n)
{
   for (int i0 = wv_lane(); i0 < n; i0 += 8 * WV_WIDTH) {
      decltype(+s[0]) v[8];
#pragma unroll
      for (int u = 0; u < 8; u++) v[u] = s[i0 + u * WV_WIDTH < n ? i0 + u * WV_WIDTH : n - 1];
#pragma unroll
      for (int u = 0; u < 8; u++) { const int i = i0 + u * WV_WIDTH; if (i < n) d[i] = v[u]; }
   }
}

#define OA_AUTO (-1000)
#define OA_BITRATE_MAX (-1)
#define OA_BW_NB 1101
#define OA_BW_MB 1102
#define OA_BW_WB 1103
#define OA_BW_SWB 1104
#define OA_BW_FB 1105

WV_DEV i32 bits_to_bitrate(i32 bits, i32 Fs, i32 frame_size) { return bits * (6 * Fs / frame_size) / 6; }
WV_DEV i32 bitrate_to_bits(i32 bitrate, i32 Fs, i32 frame_size) { return bitrate * 6 / (6 * Fs / frame_size); }
/* compute_equiv_rate (src/opus_encoder.c:1027) for an encoder whose mode can only be CELT-only: before the mode is "known" (the channel decision, :1432) the reference
 * discounts half of SILK's loss penalty whatever the application (:1055) */
WV_DEV i32 compute_equiv_rate(i32 bitrate, int channels, int frame_rate, int vbr, int celt_mode_known, int complexity, int loss)
{
   i32 equiv = bitrate;
   if (frame_rate > 50) equiv -= (40 * channels + 20) * (frame_rate - 50);
   if (!vbr) equiv -= equiv / 12;
   equiv = equiv * (90 + complexity) / 100;
   if (celt_mode_known) { if (complexity < 5) equiv = equiv * 9 / 10; }
   else equiv -= equiv * loss / (12 * loss + 20);
   return equiv;
}
WV_DEV u8 gen_toc_celt(int framerate, int bandwidth, int channels)
{
   int period = 0;
   while (framerate < 400) { framerate <<= 1; period++; }
   int tmp = bandwidth - OA_BW_MB;
   if (tmp < 0) tmp = 0;
   return (u8)(0x80 | (tmp << 5) | (period << 3) | ((channels == 2) << 2));
}

/* lane 0: the call-level decisions of opus_encode_native (src/opus_encoder.c:1325-1755) that are left when the application pins CELT-only
 * (RESTRICTED_LOWDELAY / RESTRICTED_CELT): rate, 'PLC' frames, channels, bandwidth, the split of calls above 20 ms into 20 ms frames. */
WV_DEVN void opus_layer_decide(WV_LDS FrameLds *L, const OaEncConfig *cfg, int frame_size, int out_data_bytes, int signal_type, int float_api, const OaAnalysisInfo *info)
{
   WV_LDS FrameShared *sh = &L->sh;
   WV_LDS OaEncScalars *st = &L->st;
   const int Fs = sh->Fs;
   int channels = cfg->channels;
   i32 max_data_bytes = imin(1276 * 6, out_data_bytes);
   st->rangeFinal = 0;
   sh->plc_frame = 0; sh->ret = 0; sh->skip_celt = 0; sh->cbr_bytes = -1; sh->nb_frames = 1; sh->enc_frame_size = frame_size;
   sh->CC = channels; sh->upsample = 48000 / Fs; sh->raw_frame = 0;
   sh->lsb_depth = imin(cfg->input_depth ? cfg->input_depth : 16, cfg->lsb_depth);
   if (max_data_bytes == 1 && Fs == frame_size * 10) { sh->plc_frame = 2; sh->ret = -2; return; }           /* cannot code 100 ms in one byte: OPUS_BUFFER_TOO_SMALL (:1231) */
   /* voice_ratio and the detected bandwidth from the analysis of the call (:1273-1308); without the float API voice_ratio is always -1 */
   if (!sh->is_silence || !float_api) st->voice_ratio = -1;
   int detected_bandwidth = 0;
   if (info->valid) {
      if (signal_type == OA_AUTO) st->voice_ratio = an_voice_ratio(info, st->prev_mode);
      detected_bandwidth = an_detected_bandwidth(info->bandwidth);
   }
   const int voice_est = signal_type == 3001 /* OPUS_SIGNAL_VOICE */ ? 127 : signal_type == 3002 /* OPUS_SIGNAL_MUSIC */ ? 0 : st->voice_ratio >= 0 ? st->voice_ratio * 327 >> 8 : 48;   /* :1413-1425 */
   i32 user = cfg->user_bitrate_bps == OA_AUTO ? 60 * Fs / frame_size + Fs * channels : (cfg->user_bitrate_bps == OA_BITRATE_MAX ? 1500000 : cfg->user_bitrate_bps);
   i32 bitrate_bps = imin(user, bits_to_bitrate(max_data_bytes * 8, Fs, frame_size));
   int frame_rate = Fs / frame_size;
   if (!cfg->use_vbr) {          /* hard CBR: src/opus_encoder.c:1328-1334; the packet is padded to max_data_bytes at the end (:2646) */
      i32 cbr_bytes = imin((bitrate_to_bits(bitrate_bps, Fs, frame_size) + 4) / 8, max_data_bytes);
      bitrate_bps = bits_to_bitrate(cbr_bytes * 8, Fs, frame_size);
      max_data_bytes = imax(1, cbr_bytes);
      sh->cbr_bytes = cbr_bytes;
   }
   sh->call_max_data_bytes = max_data_bytes;
   if (max_data_bytes < 3 || bitrate_bps < 3 * frame_rate * 8 || (frame_rate < 50 && (max_data_bytes * (i32)frame_rate < 300 || bitrate_bps < 2400))) {
      /* 'PLC' frame (:1345-1405).  st->mode is still its initial MODE_HYBRID until the first coded frame (opus_encoder_init :319), CELT-only afterwards */
      int tocmode = st->prev_mode == 0 ? 1001 : 1002, packet_code = 0, num_multiframes = 0;
      int bw = st->bandwidth == 0 ? OA_BW_NB : st->bandwidth;
      if (frame_rate > 100) tocmode = 1002;
      if (frame_rate == 25) { frame_rate = 50; packet_code = 1; }
      if (frame_rate <= 16) {
         if (out_data_bytes == 1) { tocmode = 1000; packet_code = frame_rate <= 12; frame_rate = frame_rate == 12 ? 25 : 16; }
         else { num_multiframes = 50 / frame_rate; frame_rate = 50; packet_code = 3; }
      }
      if (tocmode == 1000 && bw > OA_BW_WB) bw = OA_BW_WB;
      else if (tocmode == 1002 && bw == OA_BW_MB) bw = OA_BW_NB;
      else if (tocmode == 1001 && bw <= OA_BW_SWB) bw = OA_BW_SWB;
      int period = 0, fr = frame_rate;
      while (fr < 400) { fr <<= 1; period++; }
      u8 toc;
      if (tocmode == 1002) toc = gen_toc_celt(frame_rate, bw, st->stream_channels);
      else if (tocmode == 1000) toc = (u8)(((bw - OA_BW_NB) << 5) | ((period - 2) << 3) | ((st->stream_channels == 2) << 2));
      else toc = (u8)(0x60 | ((bw - OA_BW_SWB) << 4) | ((period - 2) << 3) | ((st->stream_channels == 2) << 2));
      L->packet[0] = (u8)(toc | packet_code);
      if (packet_code == 3) L->packet[1] = (u8)num_multiframes;
      sh->plc_frame = 1; sh->ret = packet_code <= 1 ? 1 : 2;
      sh->call_max_data_bytes = imax(max_data_bytes, sh->ret);
      return;
   }
   i32 equiv_rate = compute_equiv_rate(bitrate_bps, channels, frame_rate, cfg->use_vbr, 0, cfg->complexity, cfg->packet_loss_perc);
   if (cfg->force_channels != OA_AUTO && channels == 2) st->stream_channels = cfg->force_channels;
   else if (channels == 2) {
      i32 thr = 17000 + ((voice_est * voice_est * (19000 - 17000)) >> 14);
      if (st->stream_channels == 2) thr -= 1000; else thr += 1000;
      st->stream_channels = (equiv_rate > thr) ? 2 : 1;
   } else st->stream_channels = channels;
   equiv_rate = compute_equiv_rate(bitrate_bps, st->stream_channels, frame_rate, cfg->use_vbr, 1, cfg->complexity, cfg->packet_loss_perc);
   {
      const i32 voice_bw[8] = {9000, 700, 9000, 700, 13500, 1000, 14000, 2000};
      const i32 music_bw[8] = {9000, 700, 9000, 700, 11000, 1000, 12000, 2000};
      int bandwidth = OA_BW_FB;
      do {
         int k = 2 * (bandwidth - OA_BW_MB);
         int threshold = music_bw[k] + ((voice_est * voice_est * (voice_bw[k] - music_bw[k])) >> 14);
         int hysteresis = music_bw[k + 1] + ((voice_est * voice_est * (voice_bw[k + 1] - music_bw[k + 1])) >> 14);
         if (!st->first) { if (st->auto_bandwidth >= bandwidth) threshold -= hysteresis; else threshold += hysteresis; }
         if (equiv_rate >= threshold) break;
      } while (--bandwidth > OA_BW_NB);
      if (bandwidth == OA_BW_MB) bandwidth = OA_BW_WB;
      st->bandwidth = st->auto_bandwidth = bandwidth;
   }
   if (st->bandwidth > cfg->max_bandwidth) st->bandwidth = cfg->max_bandwidth;
   if (cfg->user_bandwidth != OA_AUTO) st->bandwidth = cfg->user_bandwidth;
   if (Fs <= 24000 && st->bandwidth > OA_BW_SWB) st->bandwidth = OA_BW_SWB;                          /* nothing above the input's Nyquist rate (:1641-1650) */
   if (Fs <= 16000 && st->bandwidth > OA_BW_WB) st->bandwidth = OA_BW_WB;
   if (Fs <= 12000 && st->bandwidth > OA_BW_MB) st->bandwidth = OA_BW_MB;
   if (Fs <= 8000 && st->bandwidth > OA_BW_NB) st->bandwidth = OA_BW_NB;
   if (detected_bandwidth && cfg->user_bandwidth == OA_AUTO) {                                        /* use the detected bandwidth to reduce the encoded bandwidth (:1651-1674) */
      const i32 sc = st->stream_channels;
      const int min_detected_bandwidth = equiv_rate <= 18000 * sc ? OA_BW_NB : equiv_rate <= 24000 * sc ? OA_BW_MB : equiv_rate <= 30000 * sc ? OA_BW_WB : equiv_rate <= 44000 * sc ? OA_BW_SWB : OA_BW_FB;
      st->bandwidth = imin(st->bandwidth, imax(detected_bandwidth, min_detected_bandwidth));
   }
   if (st->bandwidth == OA_BW_MB) st->bandwidth = OA_BW_WB;
   if (cfg->lfe) st->bandwidth = OA_BW_NB;
   sh->curr_bandwidth = st->bandwidth;
   sh->call_bitrate = bitrate_bps; sh->call_equiv_rate = equiv_rate;
   if (frame_size > Fs / 50) {                                                                        /* 40-120 ms: 20 ms frames, one packet (:1698-1755) */
      const int nb_frames = frame_size / (Fs / 50), max_header_bytes = nb_frames == 2 ? 3 : 2 + (nb_frames - 1) * 2;
      sh->repacketize_len = (cfg->use_vbr || cfg->user_bitrate_bps == OA_BITRATE_MAX) ? out_data_bytes : imin(sh->cbr_bytes, out_data_bytes);
      sh->max_len_sum = nb_frames + sh->repacketize_len - max_header_bytes;
      sh->nb_frames = nb_frames; sh->enc_frame_size = Fs / 50;
   }
}
/* lane 0: opus_encode_frame_native's set-up for one coded frame of a CELT-only application (:1893-1909, :2264-2295, :2320-2349, :2447-2464) */
WV_DEVN void opus_layer_frame(WV_LDS FrameLds *L, const OaEncConfig *cfg, int frame_size, int orig_max_data_bytes)
{
   WV_LDS FrameShared *sh = &L->sh;
   WV_LDS OaEncScalars *st = &L->st;
   const int channels = cfg->channels, curr_bandwidth = sh->curr_bandwidth, frame_rate = sh->Fs / frame_size;
   const i32 bitrate_bps = sh->call_bitrate, equiv_rate = sh->call_equiv_rate;
   st->rangeFinal = 0;
   sh->frame_size = frame_size; sh->skip_celt = 0; sh->ret = 0; sh->no_pad = 0;
   sh->orig_max_data_bytes = orig_max_data_bytes;
   sh->max_data_bytes = imin(orig_max_data_bytes, 1276);
   sh->pad_to = 0;
   int endband = 21;
   if (curr_bandwidth == OA_BW_NB) endband = 13;
   else if (curr_bandwidth == OA_BW_MB || curr_bandwidth == OA_BW_WB) endband = 17;
   else if (curr_bandwidth == OA_BW_SWB) endband = 19;
   sh->start = 0; sh->end = endband; sh->effEnd = endband;
   sh->C = st->stream_channels;
   sh->complexity = cfg->complexity; sh->disable_inv = cfg->disable_inv; sh->loss_rate = cfg->packet_loss_perc;
   sh->disable_pf = cfg->prediction_disabled; sh->force_intra = cfg->prediction_disabled;            /* CELT_SET_PREDICTION(reducedDependency ? 0 : 2), :2288-2295 */
   sh->vbr = cfg->use_vbr; sh->constrained_vbr = cfg->vbr_constraint;
   sh->bitrate = -1;
   if (cfg->use_vbr && bitrate_bps > 500) sh->bitrate = imin(bitrate_bps, 750000 * channels);
   i32 stereoWidth_Q14;
   if (equiv_rate > 32000) stereoWidth_Q14 = 16384;
   else if (equiv_rate < 16000) stereoWidth_Q14 = 0;
   else stereoWidth_Q14 = 16384 - 2048 * (i32)(32000 - equiv_rate) / (equiv_rate - 14000);
   sh->do_stereo_fade = 0;
   if (!sh->energy_mask_on && channels == 2 && (st->hybrid_stereo_width_Q14 < (1 << 14) || stereoWidth_Q14 < (1 << 14))) {
      i16 g1 = (i16)st->hybrid_stereo_width_Q14, g2 = (i16)stereoWidth_Q14;
      g1 = g1 == 16384 ? Q15ONE : shl16(g1, 1);
      g2 = g2 == 16384 ? Q15ONE : shl16(g2, 1);
      sh->do_stereo_fade = 1; sh->fade_g1 = g1; sh->fade_g2 = g2;
      st->hybrid_stereo_width_Q14 = stereoWidth_Q14;
   }
   sh->toc = gen_toc_celt(frame_rate, curr_bandwidth, st->stream_channels);
   st->prev_mode = 1002;
   st->first = 0;
   sh->prev_framesize = frame_size;
}

/* dc_reject (opus_encoder.c:479): the raw int16 PCM is staged into LDS with coalesced loads, then lanes 0..CC-1 each run one
 * channel's one-pole recursion in place, eight samples per trip so the LDS reads of a trip are issued back to back and
 * only the 3-operation chain (mem) is serial. */
WV_DEV void dc_reject_lanes(WV_LDS FrameLds *L, const i16 *pcm, int len, int channels)
{
   WV_LDS i16 *io = L->BC.stage16;
   {  /* two samples per word, eight trips' words in flight (the caller's frames start on word boundaries: oa_maxabs_wave reads them the same way) */
      const int n = len * channels, nw = n >> 1;
      const u32 *pw = (const u32 *)pcm; WV_LDS u32 *iw = (WV_LDS u32 *)io;
      for (int i0 = wv_lane(); i0 < nw; i0 += 8 * WV_WIDTH) {
         u32 v[8];
#pragma unroll
         for (int u = 0; u < 8; u++) v[u] = pw[imin(i0 + u * WV_WIDTH, nw - 1)];
#pragma unroll
         for (int u = 0; u < 8; u++) { const int i = i0 + u * WV_WIDTH; if (i < nw) iw[i] = v[u]; }
      }
      if ((n & 1) && wv_lane() == 0) io[n - 1] = pcm[n - 1];
   }
   wv_sync();
   int c = wv_lane();
   if (c < channels) {
      const int shift = celt_ilog2(L->sh.Fs / (3 * 4));
      i32 mem = L->st.hp_mem[2 * c];
      int i0 = 0;
      for (; i0 + 8 <= len; i0 += 8) {
         i32 x[8];
#pragma unroll
         for (int k = 0; k < 8; k++) x[k] = shl32(saturate((i32)io[channels * (i0 + k) + c], (1 << 16) - 1), 14);
#pragma unroll
         for (int k = 0; k < 8; k++) {
            i32 y = x[k] - mem;
            mem = mem + pshr32(y, shift);
            x[k] = saturate(pshr32(y, 14), 32767);
         }
#pragma unroll
         for (int k = 0; k < 8; k++) io[channels * (i0 + k) + c] = (i16)x[k];
      }
      for (; i0 < len; i0++) {                         /* (2.5 ms at 8 / 12 / 24 kHz is not a multiple of 8 samples) */
         const i32 x = shl32(saturate((i32)io[channels * i0 + c], (1 << 16) - 1), 14), y = x - mem;
         mem = mem + pshr32(y, shift);
         io[channels * i0 + c] = (i16)saturate(pshr32(y, 14), 32767);
      }
      L->st.hp_mem[2 * c] = mem;
   }
}
/* stereo_fade (:548): the cross-fade covers overlap = 120 * Fs / 48000 samples, the window is read with stride 48000 / Fs */
WV_DEV void stereo_fade_lanes(WV_LDS FrameLds *L, int frame_size)
{
   WV_LDS i16 *io = L->BC.stage16;
   const int inc = L->sh.upsample > 1 ? L->sh.upsample : 1, overlap = OA_OVERLAP / inc;
   i16 g1 = (i16)(Q15ONE - L->sh.fade_g1), g2 = (i16)(Q15ONE - L->sh.fade_g2);
   FOR_LANES(i, frame_size) {
      i16 g = g2;
      if (i < overlap) {
         i16 w = ct_window[i * inc];
         w = (i16)mult16_16_q15(w, w);
         g = (i16)(mac16_16(mult16_16(w, g2), Q15ONE - w, g1) >> 15);
      }
      i32 diff = half32((i32)io[2 * i] - (i32)io[2 * i + 1]);
      diff = mult16_16_q15(g, diff);
      io[2 * i] = (i16)(io[2 * i] - diff);
      io[2 * i + 1] = (i16)(io[2 * i + 1] + diff);
   }
}

/* celt_encode_with_ec prologue (lane 0): byte budget, VBR bounds, silence flag (celt_encoder.c:1858-2008). */
/* hyb_bytes > 0: hybrid frame, the coder continues after the SILK layer inside a buffer already shrunk to hyb_bytes (src/opus_encoder.c:2446, celt_encoder.c:1858) */
WV_DEVN void celt_prologue(WV_LDS FrameLds *L, int hyb_bytes = 0)
{
   WV_LDS FrameShared *sh = &L->sh;
   WV_LDS OaEncScalars *st = &L->st;
   EcCtx ec_; EcCtx *e = &ec_; WV_LDS u8 *buf = L->packet + 1;
   const int Fs = 48000, frame_size = sh->frame_size * (sh->upsample > 1 ? sh->upsample : 1);      /* the coder runs at 48 kHz (celt_encoder.c:1838) */
   int LM;
   for (LM = 0; LM <= 3; LM++) if (120 << LM == frame_size) break;
   sh->LM = LM; sh->M = 1 << LM; sh->N = 120 << LM;
   /* opus_encode_frame_native: ec_enc_init(data+1, orig_max-1), shrink to max_data_bytes-1 */
   int nbCompressedBytes;
   if (hyb_bytes > 0) { ec_ld(e, &L->ec); nbCompressedBytes = hyb_bytes; }
   else {
      k_ec_enc_init(EC_PASS, sh->orig_max_data_bytes - 1);
      nbCompressedBytes = sh->max_data_bytes - 1;
      k_ec_enc_shrink(EC_PASS, nbCompressedBytes);
   }
   L->packet[0] = 0;
   if (k_ec_tell(EC_PASS) > 8 * nbCompressedBytes) { sh->skip_celt = 1; EC_END; return; }
   int C = sh->C;
   i32 tell = k_ec_tell(EC_PASS), tell0_frac = k_ec_tell_frac(EC_PASS);
   int nbFilledBytes = (tell + 4) >> 3, effectiveBytes, nbAvailableBytes;
   i32 vbr_rate;
   nbCompressedBytes = imin(nbCompressedBytes, 1275);
   if (sh->vbr && sh->bitrate != -1) {
      vbr_rate = bitrate_to_bits(sh->bitrate, Fs, frame_size) << BITRES;
      effectiveBytes = vbr_rate >> (3 + BITRES);
   } else {
      vbr_rate = 0;
      i32 tmp = sh->bitrate * frame_size;
      if (tell > 1) tmp += tell * Fs;
      if (sh->bitrate != -1) {
         nbCompressedBytes = imax(2, imin(nbCompressedBytes, (tmp + 4 * Fs) / (8 * Fs)));
         k_ec_enc_shrink(EC_PASS, nbCompressedBytes);
      }
      effectiveBytes = nbCompressedBytes - nbFilledBytes;
   }
   nbAvailableBytes = nbCompressedBytes - nbFilledBytes;
   i32 equiv_rate = ((i32)nbCompressedBytes * 8 * 50 << (3 - LM)) - (40 * C + 20) * ((400 >> LM) - 50);
   if (sh->bitrate != -1) equiv_rate = imin(equiv_rate, sh->bitrate - (40 * C + 20) * ((400 >> LM) - 50));
   if (vbr_rate > 0 && sh->constrained_vbr) {
      i32 vbr_bound = vbr_rate;
      i32 max_allowed = imin(imax(tell == 1 ? 2 : 0, (vbr_rate + vbr_bound - st->vbr_reservoir) >> (BITRES + 3)), nbAvailableBytes);
      if (max_allowed < nbAvailableBytes) {
         nbCompressedBytes = nbFilledBytes + max_allowed;
         nbAvailableBytes = max_allowed;
         k_ec_enc_shrink(EC_PASS, nbCompressedBytes);
      }
   }
   i32 total_bits = nbCompressedBytes * 8;
   /* sample_max pieces were reduced by the wave into r[0] (head) and r[1] (overlap tail) */
   i32 sample_max = imax(st->overlap_max, sh->r[0]);
   st->overlap_max = sh->r[1];
   sample_max = imax(sample_max, st->overlap_max);
   int silence = (sample_max == 0);
   if (tell == 1) k_ec_enc_bit_logp(EC_PASS, silence, 15);
   else silence = 0;
   if (silence) {
      if (vbr_rate > 0) {
         effectiveBytes = nbCompressedBytes = imin(nbCompressedBytes, nbFilledBytes + 2);
         total_bits = nbCompressedBytes * 8;
         nbAvailableBytes = 2;
         k_ec_enc_shrink(EC_PASS, nbCompressedBytes);
      }
      tell = nbCompressedBytes * 8;
      e->nbits_total += tell - k_ec_tell(EC_PASS);
   }
   sh->nbCompressedBytes = nbCompressedBytes; sh->nbFilledBytes = nbFilledBytes; sh->nbAvailableBytes = nbAvailableBytes;
   sh->effectiveBytes = effectiveBytes; sh->vbr_rate = vbr_rate; sh->total_bits = total_bits; sh->equiv_rate = equiv_rate;
   sh->tell = tell; sh->tell0_frac = tell0_frac; sh->silence = silence; sh->sample_max = sample_max;
   EC_END;
}

/* The unfiltered pre-emphasised signal of one channel, indexed like the reference's pre[c][] (history then new input): the history comes straight from the stream's
 * HBM state, the new samples (x<<12 - .85*prev<<12, celt_encoder.c:557) are worked out once per frame into the frame's spectrum scratch, which nothing else uses before
 * the first MDCT (pre_stage_wave) -- so no 16 KB copy has to live in LDS, and a sample is ONE load without control flow (the compiler keeps many of them in flight). */
struct PreSrc { const i32 *hist; const i32 *xnew; };
WV_DEV i32 pre_at(const PreSrc &p, int j)
{
   const i32 *b = j < OA_MAX_PERIOD ? p.hist : p.xnew - OA_MAX_PERIOD;
   return b[j];
}
WV_DEV i32 pre_calc(const i16 *pcm, int CC, int c, i32 mem0, int up, int i)
{
   if (up > 1) {                       /* API rate below 48 kHz: sample i of the zero-stuffed signal is pcm[i / up] when up divides i, else 0 (celt_encoder.c:583-612) */
      const int q = i / up, r = i - q * up;
      const i32 x = r == 0 ? shl32((i32)pcm[CC * q + c], SIG_SHIFT) : 0;
      const i32 m = i == 0 ? mem0 : (r == 1 ? mult16_32_q15(27853, shl32((i32)pcm[CC * q + c], SIG_SHIFT)) : 0);
      return x - m;
   }
   i32 x = shl32((i32)pcm[CC * i + c], SIG_SHIFT);
   i32 m = i == 0 ? mem0 : mult16_32_q15(27853, shl32((i32)pcm[CC * (i - 1) + c], SIG_SHIFT));
   return x - m;
}
/* the frame's N new pre-emphasised samples of every channel -> xnew[c * N + i] */
WV_DEV void pre_stage_wave(i32 *xnew, const i16 *pcm, int CC, int N, i32 mem0, i32 mem1, int up)
{
   if (CC == 2 && up == 1) {           /* both channels of a sample in one word */
      const u32 *pw = (const u32 *)pcm;
      for (int i0 = wv_lane(); i0 < N; i0 += 4 * WV_WIDTH) {        /* four trips' samples in flight, then the stores */
         u32 cur[4], prv[4];
#pragma unroll
         for (int u = 0; u < 4; u++) { const int i = imin(i0 + u * WV_WIDTH, N - 1); cur[u] = pw[i]; prv[u] = pw[i > 0 ? i - 1 : 0]; }
#pragma unroll
         for (int u = 0; u < 4; u++) {
            const int i = i0 + u * WV_WIDTH;
            if (i < N) {
               const i32 m0 = i == 0 ? mem0 : mult16_32_q15(27853, shl32((i32)(i16)prv[u], SIG_SHIFT)), m1 = i == 0 ? mem1 : mult16_32_q15(27853, shl32((i32)prv[u] >> 16, SIG_SHIFT));
               xnew[i] = shl32((i32)(i16)cur[u], SIG_SHIFT) - m0;
               xnew[N + i] = shl32((i32)cur[u] >> 16, SIG_SHIFT) - m1;
            }
         }
      }
   } else {
      for (int c = 0; c < CC; c++) { FOR_LANES(i, N) xnew[c * N + i] = pre_calc(pcm, CC, c, c ? mem1 : mem0, up, i); }
   }
}

/* tone detector (celt_encoder.c:1272-1403).  x16 built in parallel, correlations by wave reductions
 * (plain int32 sums are order-free), 2x2 solve redundantly on every lane (pure scalar code). */
WV_DEV int acos_approx(i32 x)
{
   int flip = x < 0;
   x = iabs(x);
   i16 x14 = (i16)(x >> 15);
   i32 tmp = (762 * x14 >> 14) - 3308;
   tmp = (tmp * x14 >> 14) + 25726;
   tmp = tmp * fx_sqrt(imax(0, (1 << 30) - (x << 1))) >> 16;
   if (flip) tmp = 25736 - tmp;
   return tmp;
}
WV_DEV int tone_lpc_wave(const WV_LDS i16 *x, int len, int delay, i32 *lpc)
{
   i32 r00 = 0, r01 = 0, r02 = 0, e1 = 0, e2 = 0, e3 = 0;
   FOR_LANES(i, len - 2 * delay) {
      r00 += mult16_16(x[i], x[i]);
      r01 += mult16_16(x[i], x[i + delay]);
      r02 += mult16_16(x[i], x[i + 2 * delay]);
   }
   FOR_LANES(i, delay) {
      e1 += mult16_16(x[len + i - 2 * delay], x[len + i - 2 * delay]) - mult16_16(x[i], x[i]);
      e2 += mult16_16(x[len + i - delay], x[len + i - delay]) - mult16_16(x[i + delay], x[i + delay]);
      e3 += mult16_16(x[len + i - 2 * delay], x[len + i - delay]) - mult16_16(x[i], x[i + delay]);
   }
   r00 = wv_sum(r00); r01 = wv_sum(r01); r02 = wv_sum(r02); e1 = wv_sum(e1); e2 = wv_sum(e2); e3 = wv_sum(e3);
   i32 r11 = r00 + e1, r22 = r11 + e2, r12 = r01 + e3;
   {
      i32 R00 = r00 + r22, R01 = r01 + r12, R11 = 2 * r11, R02 = 2 * r02, R12 = r12 + r01;
      r00 = R00; r01 = R01; r11 = R11; r02 = R02; r12 = R12;
   }
   i32 den = mult32_32_q31(r00, r11) - mult32_32_q31(r01, r01);
   if (den <= (mult32_32_q31(r00, r11) >> 10)) return 1;
   i32 num1 = mult32_32_q31(r02, r11) - mult32_32_q31(r01, r12);
   if (num1 >= den) lpc[1] = QC32(1.f, 29);
   else if (num1 <= -den) lpc[1] = -QC32(1.f, 29);
   else lpc[1] = fx_frac_div32_q29(num1, den);
   i32 num0 = mult32_32_q31(r00, r12) - mult32_32_q31(r02, r01);
   if (half32(num0) >= den) lpc[0] = QC32(1.999999f, 29);
   else if (half32(num0) <= -den) lpc[0] = -QC32(1.999999f, 29);
   else lpc[0] = fx_frac_div32_q29(num0, den);
   return 0;
}
WV_DEVN void tone_detect_wave(WV_LDS FrameLds *L, const PreSrc &p0, const PreSrc &p1)
{
   const int CC = L->sh.CC, N = L->sh.N + OA_OVERLAP;
   WV_LDS i16 *x = L->BC.x16[0];
   const int j0 = OA_MAX_PERIOD - OA_OVERLAP;      /* in[c][i] == pre[c][1024 - overlap + i] */
   i32 ac0 = 0;
   for (int i0 = wv_lane(); i0 < N; i0 += 6 * WV_WIDTH) {           /* six trips' samples in flight (N = 1,080 at 20 ms: three batches) */
      i32 a0[6], a1[6];
#pragma unroll
      for (int u = 0; u < 6; u++) { const int i = imin(i0 + u * WV_WIDTH, N - 1); a0[u] = pre_at(p0, j0 + i); a1[u] = CC == 2 ? pre_at(p1, j0 + i) : 0; }
#pragma unroll
      for (int u = 0; u < 6; u++) {
         const int i = i0 + u * WV_WIDTH;
         if (i < N) {
            i16 v = CC == 2 ? (i16)pshr32(add32(a0[u] >> 1, a1[u] >> 1), SIG_SHIFT + 2) : (i16)pshr32(a0[u], SIG_SHIFT + 2);
            x[i] = v;
            ac0 += mult16_16(v, v) >> 10;
         }
      }
   }
   ac0 = add32(N, wv_sum(ac0));
   int shift = 5 - (28 - celt_ilog2(ac0)) / 2;
   wv_sync();
   if (shift > 0) { FOR_LANES(i, N) x[i] = (i16)pshr32(x[i], shift); }
   wv_sync();
   i32 lpc[2] = {0, 0};
   int delay = 1;
   int fail = tone_lpc_wave(x, N, delay, lpc);
   while (delay <= 48000 / 3000 && (fail || (lpc[0] > QC32(1.f, 29) && lpc[1] < 0))) {
      delay *= 2;
      fail = tone_lpc_wave(x, N, delay, lpc);
   }
   i32 toneishness; i16 freq;
   if (!fail && mult32_32_q31(lpc[0], lpc[0]) + mult32_32_q31(QC32(3.999999, 29), lpc[1]) < 0) {
      toneishness = -lpc[1];
      freq = (i16)((acos_approx(lpc[0] >> 1) + delay / 2) / delay);
   } else { freq = -1; toneishness = 0; }
   wv_sync();
   LANE0 { L->sh.tone_freq = freq; L->sh.toneishness = toneishness; }
}

/* transient_analysis (celt_encoder.c:267): the HP filter and the forward/backward masking followers are
 * recursions with rounding -> one lane per channel runs them; ranges/normalisation use wave reductions. */
WV_DEVN void transient_analysis_wave(WV_LDS FrameLds *L, const PreSrc &p0, const PreSrc &p1, int allow_weak_transients, const i32 *tr_pre = nullptr /* the channels' unmask values from
      ct_transient_tile (below), or NULL */)
{
   const u8 inv_table[128] = {
      255, 255, 156, 110, 86, 70, 59, 51, 45, 40, 37, 33, 31, 28, 26, 25, 23, 22, 21, 20, 19, 18, 17, 16, 16, 15, 15, 14, 13, 13, 12, 12,
      12, 12, 11, 11, 11, 10, 10, 10, 9, 9, 9, 9, 9, 9, 8, 8, 8, 8, 8, 7, 7, 7, 7, 7, 7, 6, 6, 6, 6, 6, 6, 6,
      6, 6, 6, 6, 6, 6, 6, 6, 6, 5, 5, 5, 5, 5, 5, 5, 5, 5, 5, 5, 5, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4,
      4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 3, 3, 3, 3, 3, 3, 3, 3, 3, 3, 3, 3, 3, 3, 3, 3, 3, 2};
   const int C = L->sh.CC, len = L->sh.N + OA_OVERLAP, len2 = len / 2, lane = wv_lane();
   const int forward_shift = allow_weak_transients ? 5 : 4;
   const int j0 = OA_MAX_PERIOD - OA_OVERLAP;      /* in[c][i] == pre[c][1024 - overlap + i] */
   i32 unmask_c = 0;
   const bool have_pre = tr_pre != nullptr && !allow_weak_transients && !wv_uni(L->sh.do_stereo_fade) && wv_uni(L->sh.upsample) <= 1 && wv_uni(tr_pre[2]) == len;   /* (worked out from the frame this call codes, unfaded, 48 kHz) */
   if (have_pre) { if (lane < C) unmask_c = tr_pre[lane]; }
   else {
   i32 mx = 0;
   FOR_LANES(i, len) { mx = imax(mx, iabs(pre_at(p0, j0 + i))); if (C == 2) mx = imax(mx, iabs(pre_at(p1, j0 + i))); }
   mx = wv_max(mx);                       /* celt_maxabs32 over both channels (|INT32_MIN| cannot occur: SIG range) */
   const int in_shift = imax(0, celt_ilog2(1 + mx) - 14);
   wv_sync();
   /* the shifted input fits 16 bits (|x| < 2^15 by construction of in_shift): stage it in place of the filter output */
   FOR_LANES(i, len) { L->BC.x16[0][i] = (i16)(pre_at(p0, j0 + i) >> in_shift); if (C == 2) L->BC.x16[1][i] = (i16)(pre_at(p1, j0 + i) >> in_shift); }
   wv_sync();
   if (lane < C) {
      WV_LDS i16 *tmp = L->BC.x16[lane];
      i32 mem0 = 0, mem1 = 0;
      for (int i0 = 0; i0 < len; i0 += 8) {            /* len = N + 120 is a multiple of 8; eight reads in flight per trip */
         i32 x[8];
#pragma unroll
         for (int k = 0; k < 8; k++) x[k] = tmp[i0 + k];
#pragma unroll
         for (int k = 0; k < 8; k++) {
            i32 y = add32(mem0, x[k]);
            mem0 = mem1 + y - shl32(x[k], 1);
            mem1 = x[k] - (y >> 1);
            x[k] = sround16(y, 2);
         }
#pragma unroll
         for (int k = 0; k < 8; k++) tmp[i0 + k] = (i16)x[k];
      }
      for (int i = 0; i < 12; i++) tmp[i] = 0;
   }
   wv_sync();
   /* per-channel normalisation to max range */
   i32 m0 = 0, m1 = 0;
   FOR_LANES(i, len) { m0 = imax(m0, iabs((i32)L->BC.x16[0][i])); if (C == 2) m1 = imax(m1, iabs((i32)L->BC.x16[1][i])); }
   m0 = wv_max(m0); m1 = wv_max(m1);
   {
      int s0 = 14 - celt_ilog2(imax(1, m0)), s1 = 14 - celt_ilog2(imax(1, m1));
      FOR_LANES(i, len) {
         if (s0 != 0) L->BC.x16[0][i] = shl16(L->BC.x16[0][i], s0);
         if (C == 2 && s1 != 0) L->BC.x16[1][i] = shl16(L->BC.x16[1][i], s1);
      }
   }
   wv_sync();
   if (lane < C) {
      WV_LDS i16 *tmp = L->BC.x16[lane];
      i32 mean = 0, mem0 = 0, norm;
      i16 maxE = 0;
      for (int i0 = 0; i0 < len2; i0 += 4) {           /* len2 is a multiple of 4; the energies are chain-independent */
         i32 x2[4];
#pragma unroll
         for (int k = 0; k < 4; k++) {
            i32 a = tmp[2 * (i0 + k)], b = tmp[2 * (i0 + k) + 1];
            x2[k] = pshr32(mult16_16(a, a) + mult16_16(b, b), 4);
            mean += pshr32(x2[k], 12);
         }
#pragma unroll
         for (int k = 0; k < 4; k++) { mem0 = mem0 + pshr32(x2[k] - mem0, forward_shift); x2[k] = pshr32(mem0, 12); }
#pragma unroll
         for (int k = 0; k < 4; k++) tmp[i0 + k] = (i16)x2[k];
      }
      mem0 = 0;
      for (int i0 = len2 - 4; i0 >= 0; i0 -= 4) {
         i32 t[4];
#pragma unroll
         for (int k = 0; k < 4; k++) t[k] = shl32(tmp[i0 + k], 4);
#pragma unroll
         for (int k = 3; k >= 0; k--) { mem0 = mem0 + pshr32(t[k] - mem0, 3); t[k] = (i16)pshr32(mem0, 4); maxE = (i16)imax(maxE, t[k]); }
#pragma unroll
         for (int k = 0; k < 4; k++) tmp[i0 + k] = (i16)t[k];
      }
      mean = mult16_16(fx_sqrt(mean), fx_sqrt(mult16_16(maxE, len2 >> 1)));
      norm = shl32((i32)len2, 6 + 14) / add32(EPSILON, mean >> 1);
      i32 unmask = 0;
      for (int i = 12; i < len2 - 5; i += 4) {
         int id = imax(0, imin(127, mult16_32_q15(tmp[i] + EPSILON, norm)));
         unmask += inv_table[id];
      }
      unmask_c = 64 * unmask * 4 / (6 * (len2 - 17));
   }
   }
   i32 u0 = wv_bcast(unmask_c, 0), u1 = wv_bcast(unmask_c, 1);
   i32 mask_metric = 0; int tf_chan = L->sh.tf_chan;
   if (u0 > mask_metric) { tf_chan = 0; mask_metric = u0; }
   if (C == 2 && u1 > mask_metric) { tf_chan = 1; mask_metric = u1; }
   int is_transient = mask_metric > 200, weak = 0;
   if (L->sh.toneishness > QC32(.98f, 29) && (i16)L->sh.tone_freq < QC16(0.026f, 13)) { is_transient = 0; mask_metric = 0; }
   if (allow_weak_transients && is_transient && mask_metric < 600) { is_transient = 0; weak = 1; }
   i16 tf_max = (i16)imax(0, fx_sqrt(27 * mask_metric) - 42);
   i16 tf_estimate = (i16)fx_sqrt(imax(0, shl32(mult16_16(QC16(0.0069, 14), imin(163, tf_max)), 14) - QC32(0.139, 28)));
   wv_sync();
   LANE0 { L->sh.isTransient = is_transient; L->sh.weak_transient = weak; L->sh.tf_estimate = tf_estimate; L->sh.tf_chan = tf_chan; }
}

/* The serial part of transient_analysis with one LANE per (stream, channel) -- 64 of them per wave, ahead of the CELT-only encode kernel of a wide launch (oa_celt_transient_kernel):
 * the high-pass and the two masking followers are recursions with rounding over 1,080 / 540 samples that keep two lanes of a stream's wave busy for 4.4 % of a config-2 frame.
 * A lane regenerates its channel's input on the fly -- dc_reject (src/opus_encoder.c:479, from the state as it stands before the call), pre-emphasis (celt_encoder.c:557), the
 * overlap from the stream's history -- twice (range, then filter) instead of storing it; the followers' array lives in the tile's HBM scratch as [sample][lane] (coalesced).  The result, the
 * channel's unmask value, is used by transient_analysis_wave when the frame turns out to be coded the plain way (no stereo fade, 48 kHz, the reference's forward_shift 4);
 * tone override, tf_estimate and tf_chan stay there. */
/* the int16 signal celt_encode_with_ec is handed, eight samples of one channel at a time: dc_reject of the caller's PCM (the CELT-only applications, src/opus_encoder.c:479) */
struct CtSrcDc {
   const i16 *pcm; int CC; i32 mem, mem0; int shift;
   WV_MEM void rewind() { mem = mem0; }
   WV_MEM void block(int k0, i32 *v)
   {
      i32 raw[8];
#pragma unroll
      for (int j = 0; j < 8; j++) raw[j] = pcm[CC * (k0 + j)];
#pragma unroll
      for (int j = 0; j < 8; j++) {
         const i32 x = shl32(saturate(raw[j], (1 << 16) - 1), 14), y = x - mem;
         mem = mem + pshr32(y, shift);
         v[j] = (i16)saturate(pshr32(y, 14), 32767);
      }
   }
};
template <class SRC> struct CtTrGen {                            /* regenerates in[i] = pre[c][1024 - 120 + i], i = 0 .. N + 119, eight at a time (the block's loads in flight together) */
   const i32 *hist; SRC src; i32 pre_mem0, prev;
   WV_MEM void rewind() { src.rewind(); prev = 0; }
   WV_MEM void block(int i0, i32 *v)                              /* v[0..7] = in[i0 .. i0 + 7]; i0 a multiple of 8 (so is the overlap: a block is all history or all new input) */
   {
      if (i0 < OA_OVERLAP) {
#pragma unroll
         for (int j = 0; j < 8; j++) v[j] = hist[OA_MAX_PERIOD - OA_OVERLAP + i0 + j];
         return;
      }
      const int k0 = i0 - OA_OVERLAP;
      src.block(k0, v);
#pragma unroll
      for (int j = 0; j < 8; j++) {                               /* celt_preemphasis at 48 kHz (celt_encoder.c:557) */
         const i32 s = shl32(v[j], SIG_SHIFT);
         const i32 m = k0 + j == 0 ? pre_mem0 : mult16_32_q15(27853, prev);
         prev = s;
         v[j] = s - m;
      }
   }
};
/* one lane = one channel; returns the channel's unmask value.  Every lane of the wave calls it (the two channels of a stream sit on neighbouring lanes and meet in one shuffle) */
template <class SRC> WV_DEV i32 ct_transient_lane(CtTrGen<SRC> &g, int N, int CC, i16 *col /* [N + 120] at stride 64 */)
{
   const u8 inv_table[128] = {
      255, 255, 156, 110, 86, 70, 59, 51, 45, 40, 37, 33, 31, 28, 26, 25, 23, 22, 21, 20, 19, 18, 17, 16, 16, 15, 15, 14, 13, 13, 12, 12,
      12, 12, 11, 11, 11, 10, 10, 10, 9, 9, 9, 9, 9, 9, 8, 8, 8, 8, 8, 7, 7, 7, 7, 7, 7, 6, 6, 6, 6, 6, 6, 6,
      6, 6, 6, 6, 6, 6, 6, 6, 6, 5, 5, 5, 5, 5, 5, 5, 5, 5, 5, 5, 5, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4,
      4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 3, 3, 3, 3, 3, 3, 3, 3, 3, 3, 3, 3, 3, 3, 3, 3, 3, 2};
   const int len = N + OA_OVERLAP, len2 = len / 2;                /* len is a multiple of 8, len2 of 4 (N = 120 .. 960) */
   /* pass 1: the range of the input over both channels (celt_maxabs32, :281) */
   i32 mx = 0;
   g.rewind();
   for (int i0 = 0; i0 < len; i0 += 8) {
      i32 v[8]; g.block(i0, v);
#pragma unroll
      for (int j = 0; j < 8; j++) mx = imax(mx, iabs(v[j]));
   }
   if (CC == 2) mx = imax(mx, wv_shfl(mx, wv_lane() ^ 1));
   const int in_shift = imax(0, celt_ilog2(1 + mx) - 14);
   /* pass 2: high-pass (:298-320), its range */
   i32 mem0 = 0, mem1 = 0, m = 0;
   g.rewind();
   for (int i0 = 0; i0 < len; i0 += 8) {
      i32 v[8]; g.block(i0, v);
#pragma unroll
      for (int j = 0; j < 8; j++) {
         const i32 x = (i16)(v[j] >> in_shift), y = add32(mem0, x);
         mem0 = mem1 + y - shl32(x, 1);
         mem1 = x - (y >> 1);
         const i32 t = i0 + j < 12 ? 0 : sround16(y, 2);
         col[(size_t)(i0 + j) * 64] = (i16)t;
         m = imax(m, iabs(t));
      }
   }
   const int sh = 14 - celt_ilog2(imax(1, m));
   /* pass 3: pair energies, forward follower (:340-362): four pairs per trip, their eight loads ahead of the four stores */
   i32 mean = 0; mem0 = 0;
   for (int i0 = 0; i0 < len2; i0 += 4) {
      i32 t[8];
#pragma unroll
      for (int j = 0; j < 8; j++) t[j] = col[(size_t)(2 * i0 + j) * 64];
#pragma unroll
      for (int j = 0; j < 4; j++) {
         i32 a = t[2 * j], b = t[2 * j + 1];
         if (sh != 0) { a = shl16(a, sh); b = shl16(b, sh); }
         const i32 x2 = pshr32(mult16_16(a, a) + mult16_16(b, b), 4);
         mean += pshr32(x2, 12);
         mem0 = mem0 + pshr32(x2 - mem0, 4);
         t[j] = pshr32(mem0, 12);
      }
#pragma unroll
      for (int j = 0; j < 4; j++) col[(size_t)(i0 + j) * 64] = (i16)t[j];
   }
   /* pass 4: backward follower (:364-380) */
   i16 maxE = 0; mem0 = 0;
   for (int i0 = len2 - 4; i0 >= 0; i0 -= 4) {
      i32 t[4];
#pragma unroll
      for (int j = 0; j < 4; j++) t[j] = shl32((i32)col[(size_t)(i0 + j) * 64], 4);
#pragma unroll
      for (int j = 3; j >= 0; j--) { mem0 = mem0 + pshr32(t[j] - mem0, 3); t[j] = (i16)pshr32(mem0, 4); maxE = (i16)imax(maxE, t[j]); }
#pragma unroll
      for (int j = 0; j < 4; j++) col[(size_t)(i0 + j) * 64] = (i16)t[j];
   }
   mean = mult16_16(fx_sqrt(mean), fx_sqrt(mult16_16(maxE, len2 >> 1)));
   const i32 norm = shl32((i32)len2, 6 + 14) / add32(EPSILON, mean >> 1);
   i32 unmask = 0;
#pragma unroll 8
   for (int i = 12; i < len2 - 5; i += 4) unmask += inv_table[imax(0, imin(127, mult16_32_q15(col[(size_t)i * 64] + EPSILON, norm)))];
   return 64 * unmask * 4 / (6 * (len2 - 17));
}
/* the CELT-only applications: tr [stream][4] = the two channels' values, the frame length they are good for (48 kHz), - */
WV_DEVN void ct_transient_tile(const OaStream *streams, const i16 *pcm, int pcm_row, int N, int CC, int first, int stride, int n_items, int base, i16 *scr /* [N + 120][64] */, i32 *tr)
{
   const int lane = wv_lane(), it = base + lane, itc = it < n_items ? it : n_items - 1;            /* (lanes past the end shadow the last item and write nothing: every lane meets the shuffle) */
   const int si = CC == 2 ? itc >> 1 : itc, c = CC == 2 ? itc & 1 : 0, s = first + si * stride;
   const OaStream *gs = streams + s;
   const int Fs = gs->Fs ? gs->Fs : 48000;
   CtTrGen<CtSrcDc> g;
   g.hist = gs->st.prefilter_mem + c * OA_MAX_PERIOD; g.pre_mem0 = gs->st.s.preemph_memE[c];
   g.src.pcm = pcm + (size_t)s * pcm_row * CC + c; g.src.CC = CC; g.src.mem0 = gs->st.s.hp_mem[2 * c]; g.src.shift = celt_ilog2(Fs / (3 * 4));
   const i32 u = ct_transient_lane(g, N, CC, scr + lane);
   if (it < n_items) {
      tr[4 * s + c] = u;
      if (c == 0) tr[4 * s + 2] = Fs == 48000 ? N + OA_OVERLAP : 0;
   }
}
#endif

/* oc_opus_enc.c — Opus-level glue around the CELT encoder for the CELT-only applications.
 * Oracle restatement of src/opus_encoder.c: :204 opus_encoder_init (defaults), :330 gen_toc, :479 dc_reject (fixed),
 * :548 stereo_fade, :733 user_bitrate_to_bitrate, :1027 compute_equiv_rate, :1182 opus_encode_native (the decisions
 * that survive when the application pins MODE_CELT_ONLY), :1855 opus_encode_frame_native (CELT branch), :2671 opus_encode.
 * Out of scope here (return -5 = OPUS_UNIMPLEMENTED-like): SILK/hybrid modes, Fs != 48000, >20 ms packets, CBR padding, DTX. */
#include "oc_opus_enc.h"

static i32 bits_to_bitrate(i32 bits, i32 Fs, i32 frame_size) { return bits * (6 * Fs / frame_size) / 6; }
static i32 bitrate_to_bits(i32 bitrate, i32 Fs, i32 frame_size) { return bitrate * 6 / (6 * Fs / frame_size); }

int oc_opus_enc_size(void) { return (int)sizeof(oc_opus_enc); }

int oc_opus_enc_init(oc_opus_enc *st, int Fs, int channels, int application)
{
   if (Fs != 48000 || (channels != 1 && channels != 2)) return -1;
   if (application != OC_APPLICATION_RESTRICTED_LOWDELAY && application != OC_APPLICATION_RESTRICTED_CELT) return -5;
   memset(st, 0, sizeof(*st));
   st->Fs = Fs; st->channels = st->stream_channels = channels; st->application = application;
   oc_celt_enc_init(&st->celt, channels);
   st->complexity = 9; st->celt.complexity = 9;
   st->use_vbr = 1; st->vbr_constraint = 1; st->user_bitrate_bps = OC_AUTO; st->bitrate_bps = 3000 + Fs * channels;
   st->user_bandwidth = OC_AUTO; st->max_bandwidth = OC_BANDWIDTH_FULLBAND; st->force_channels = OC_AUTO;
   st->lsb_depth = 24; st->hybrid_stereo_width_Q14 = 1 << 14; st->first = 1; st->bandwidth = OC_BANDWIDTH_FULLBAND;
   return 0;
}
int oc_opus_enc_set(oc_opus_enc *st, int what, int value)
{
   switch (what) {
   case 0:
      if (value != OC_AUTO && value != OC_BITRATE_MAX) {
         if (value <= 0) return -1;
         else if (value <= 500) value = 500;
         else if (value > 750000 * st->channels) value = 750000 * st->channels;
      }
      st->user_bitrate_bps = value; break;
   case 1: if (value < 0 || value > 10) return -1; st->complexity = value; st->celt.complexity = value; break;
   case 2: st->use_vbr = value; break;
   case 3: st->vbr_constraint = value; break;
   case 4: st->force_channels = value; break;
   case 5: st->user_bandwidth = value; break;
   case 6: st->max_bandwidth = value; break;
   case 7: if (value < 8 || value > 24) return -1; st->lsb_depth = value; break;
   case 8: st->celt.disable_inv = value; break;
   case 9: if (value < 0 || value > 100) return -1; st->packet_loss_perc = value; st->celt.loss_rate = value; break;
   default: return -5;
   }
   return 0;
}
u32 oc_opus_enc_final_range(const oc_opus_enc *st) { return st->rangeFinal; }

static u8 gen_toc_celt(int framerate, int bandwidth, int channels)
{
   int period = 0;
   while (framerate < 400) { framerate <<= 1; period++; }
   int tmp = bandwidth - OC_BANDWIDTH_MEDIUMBAND;
   if (tmp < 0) tmp = 0;
   u8 toc = 0x80;
   toc |= tmp << 5;
   toc |= period << 3;
   toc |= (channels == 2) << 2;
   return toc;
}
static void dc_reject(const i16 *in, i32 cutoff_Hz, i16 *out, i32 *hp_mem, int len, int channels, i32 Fs)
{
   int shift = celt_ilog2(Fs / (cutoff_Hz * 4));
   for (int c = 0; c < channels; c++)
      for (int i = 0; i < len; i++) {
         i32 x = saturate((i32)in[channels * i + c], (1 << 16) - 1);
         x = shl32(x, 14);
         i32 y = x - hp_mem[2 * c];
         hp_mem[2 * c] = hp_mem[2 * c] + pshr32(x - hp_mem[2 * c], shift);
         out[channels * i + c] = (i16)saturate(pshr32(y, 14), 32767);
      }
}
static void stereo_fade(const i16 *in, i16 *out, i16 g1, i16 g2, int overlap48, int frame_size, int channels, i32 Fs)
{
   int inc = imax(1, 48000 / Fs), overlap = overlap48 / inc, i;
   g1 = (i16)(Q15ONE - g1);
   g2 = (i16)(Q15ONE - g2);
   for (i = 0; i < overlap; i++) {
      i16 w = oc_window[i * inc];
      w = (i16)mult16_16_q15(w, w);
      i16 g = (i16)(mac16_16(mult16_16(w, g2), Q15ONE - w, g1) >> 15);
      i32 diff = half32((i32)in[i * channels] - (i32)in[i * channels + 1]);
      diff = mult16_16_q15(g, diff);
      out[i * channels] = (i16)(out[i * channels] - diff);
      out[i * channels + 1] = (i16)(out[i * channels + 1] + diff);
   }
   for (; i < frame_size; i++) {
      i32 diff = half32((i32)in[i * channels] - (i32)in[i * channels + 1]);
      diff = mult16_16_q15(g2, diff);
      out[i * channels] = (i16)(out[i * channels] - diff);
      out[i * channels + 1] = (i16)(out[i * channels + 1] + diff);
   }
}
static i32 compute_equiv_rate(i32 bitrate, int channels, int frame_rate, int vbr, int celt_mode_known, int complexity)
{
   i32 equiv = bitrate;
   if (frame_rate > 50) equiv -= (40 * channels + 20) * (frame_rate - 50);
   if (!vbr) equiv -= equiv / 12;
   equiv = equiv * (90 + complexity) / 100;
   if (celt_mode_known) { if (complexity < 5) equiv = equiv * 9 / 10; }
   /* mode not known yet: "equiv -= equiv*loss/(12*loss+20)" with loss handled by the caller (0 in scope) */
   return equiv;
}

static const i32 mono_voice_bw[8] = {9000, 700, 9000, 700, 13500, 1000, 14000, 2000};
static const i32 mono_music_bw[8] = {9000, 700, 9000, 700, 11000, 1000, 12000, 2000};
static const i32 stereo_voice_bw[8] = {9000, 700, 9000, 700, 13500, 1000, 14000, 2000};
static const i32 stereo_music_bw[8] = {9000, 700, 9000, 700, 11000, 1000, 12000, 2000};

/* opus_packet_pad (src/repacketizer.c:346 -> opus_repacketizer_out_range_impl :112 with pad = 1) for the only shape this encoder
 * emits: a code-0 packet [toc][len-1 frame bytes] becomes the code-3 packet [toc|3][0x01 or 0x41][padding length][frame][zeros]. */
static int packet_pad1(u8 *data, int len, int new_len)
{
   if (len < 1) return -1;
   if (len == new_len) return 0;
   if (len > new_len) return -1;
   int L0 = len - 1, tot = L0 + 2;
   if (tot > new_len) return -2;
   int pad_amount = new_len - tot, hdr = 2;
   u8 tmp[1500];
   memcpy(tmp, data + 1, L0);
   data[0] = (data[0] & 0xFC) | 0x3;
   data[1] = 1;
   if (pad_amount != 0) {
      int nb_255s = (pad_amount - 1) / 255;
      data[1] |= 0x40;
      for (int i = 0; i < nb_255s; i++) data[hdr++] = 255;
      data[hdr++] = (u8)(pad_amount - 255 * nb_255s - 1);
   }
   memcpy(data + hdr, tmp, L0);
   for (int i = hdr + L0; i < new_len; i++) data[i] = 0;
   return 0;
}

int oc_opus_encode(oc_opus_enc *st, const i16 *pcm, int frame_size, u8 *data, int out_data_bytes)
{
   i16 pcm_buf[2 * 960];
   oc_ec enc;
   int Fs = st->Fs, lsb_depth = imin(16, st->lsb_depth), frame_rate, voice_est = 48, ret;
   i32 max_data_bytes, equiv_rate, max_rate;
   if (400 * frame_size != Fs && 200 * frame_size != Fs && 100 * frame_size != Fs && 50 * frame_size != Fs) return -1;
   max_data_bytes = imin(1276 * 6, out_data_bytes);
   st->rangeFinal = 0;
   if (frame_size <= 0 || max_data_bytes <= 0) return -1;
   if (st->packet_loss_perc != 0) return -5;   /* loss-aware equiv_rate: not restated */
   st->bitrate_bps = imin(st->user_bitrate_bps == OC_AUTO ? 60 * Fs / frame_size + Fs * st->channels :
         st->user_bitrate_bps == OC_BITRATE_MAX ? 1500000 : st->user_bitrate_bps, bits_to_bitrate(max_data_bytes * 8, Fs, frame_size));
   frame_rate = Fs / frame_size;
   if (!st->use_vbr) {          /* hard CBR: src/opus_encoder.c:1328-1334 */
      i32 cbr_bytes = imin((bitrate_to_bits(st->bitrate_bps, Fs, frame_size) + 4) / 8, max_data_bytes);
      st->bitrate_bps = bits_to_bitrate(cbr_bytes * 8, Fs, frame_size);
      max_data_bytes = imax(1, cbr_bytes);
   }
   if (max_data_bytes < 3 || st->bitrate_bps < 3 * frame_rate * 8 || (frame_rate < 50 && (max_data_bytes * (i32)frame_rate < 300 || st->bitrate_bps < 2400))) {
      /* "PLC frame": TOC only (src/opus_encoder.c:1340-1406).  st->mode is still its initial MODE_HYBRID until the first coded frame
       * (opus_encoder_init :319), CELT-only afterwards; 2.5 ms frames are always CELT. */
      int tocmode = st->prev_mode == 0 ? 1001 : 1002;
      int bw = st->bandwidth == 0 ? OC_BANDWIDTH_NARROWBAND : st->bandwidth;
      if (frame_rate > 100) tocmode = 1002;
      if (tocmode == 1002 && bw == OC_BANDWIDTH_MEDIUMBAND) bw = OC_BANDWIDTH_NARROWBAND;
      else if (tocmode == 1001 && bw <= OC_BANDWIDTH_SUPERWIDEBAND) bw = OC_BANDWIDTH_SUPERWIDEBAND;
      if (tocmode == 1002) data[0] = gen_toc_celt(frame_rate, bw, st->stream_channels);
      else {
         int period = 0, fr = frame_rate;
         while (fr < 400) { fr <<= 1; period++; }
         data[0] = (u8)(0x60 | ((bw - OC_BANDWIDTH_SUPERWIDEBAND) << 4) | ((period - 2) << 3) | ((st->stream_channels == 2) << 2));
      }
      if (!st->use_vbr) { max_data_bytes = imax(max_data_bytes, 1); return packet_pad1(data, 1, max_data_bytes) == 0 ? max_data_bytes : -3; }
      return 1;
   }
   max_rate = bits_to_bitrate(max_data_bytes * 8, Fs, frame_size);
   (void)max_rate;
   equiv_rate = compute_equiv_rate(st->bitrate_bps, st->channels, frame_rate, st->use_vbr, 0, st->complexity);
   if (st->force_channels != OC_AUTO && st->channels == 2) st->stream_channels = st->force_channels;
   else if (st->channels == 2) {
      i32 stereo_threshold = 17000 + ((voice_est * voice_est * (19000 - 17000)) >> 14);
      if (st->stream_channels == 2) stereo_threshold -= 1000; else stereo_threshold += 1000;
      st->stream_channels = (equiv_rate > stereo_threshold) ? 2 : 1;
   } else st->stream_channels = st->channels;
   equiv_rate = compute_equiv_rate(st->bitrate_bps, st->stream_channels, frame_rate, st->use_vbr, 0, st->complexity);
   /* mode is pinned to CELT-only by the application */
   equiv_rate = compute_equiv_rate(st->bitrate_bps, st->stream_channels, frame_rate, st->use_vbr, 1, st->complexity);
   {
      const i32 *vt, *mt;
      i32 thr[8];
      int bandwidth = OC_BANDWIDTH_FULLBAND;
      if (st->channels == 2 && st->force_channels != 1) { vt = stereo_voice_bw; mt = stereo_music_bw; }
      else { vt = mono_voice_bw; mt = mono_music_bw; }
      for (int i = 0; i < 8; i++) thr[i] = mt[i] + ((voice_est * voice_est * (vt[i] - mt[i])) >> 14);
      do {
         int threshold = thr[2 * (bandwidth - OC_BANDWIDTH_MEDIUMBAND)], hysteresis = thr[2 * (bandwidth - OC_BANDWIDTH_MEDIUMBAND) + 1];
         if (!st->first) { if (st->auto_bandwidth >= bandwidth) threshold -= hysteresis; else threshold += hysteresis; }
         if (equiv_rate >= threshold) break;
      } while (--bandwidth > OC_BANDWIDTH_NARROWBAND);
      if (bandwidth == OC_BANDWIDTH_MEDIUMBAND) bandwidth = OC_BANDWIDTH_WIDEBAND;
      st->bandwidth = st->auto_bandwidth = bandwidth;
   }
   if (st->bandwidth > st->max_bandwidth) st->bandwidth = st->max_bandwidth;
   if (st->user_bandwidth != OC_AUTO) st->bandwidth = st->user_bandwidth;
   st->celt.lsb_depth = lsb_depth;
   if (st->bandwidth == OC_BANDWIDTH_MEDIUMBAND) st->bandwidth = OC_BANDWIDTH_WIDEBAND;
   int curr_bandwidth = st->bandwidth;

   /* ---- opus_encode_frame_native, CELT-only branch ---- */
   int orig_max_data_bytes = max_data_bytes;
   max_data_bytes = imin(orig_max_data_bytes, 1276);
   data += 1;
   oc_ec_enc_init(&enc, data, orig_max_data_bytes - 1);
   dc_reject(pcm, 3, pcm_buf, st->hp_mem, frame_size, st->channels, Fs);
   {
      int endband = 21;
      switch (curr_bandwidth) {
      case OC_BANDWIDTH_NARROWBAND: endband = 13; break;
      case OC_BANDWIDTH_MEDIUMBAND: case OC_BANDWIDTH_WIDEBAND: endband = 17; break;
      case OC_BANDWIDTH_SUPERWIDEBAND: endband = 19; break;
      case OC_BANDWIDTH_FULLBAND: endband = 21; break;
      }
      st->celt.end = endband;
      st->celt.stream_channels = st->stream_channels;
      st->celt.bitrate = -1;
      st->celt.disable_pf = 0; st->celt.force_intra = 0;   /* CELT_SET_PREDICTION(2) */
   }
   if (equiv_rate > 32000) st->stereoWidth_Q14 = 16384;
   else if (equiv_rate < 16000) st->stereoWidth_Q14 = 0;
   else st->stereoWidth_Q14 = 16384 - 2048 * (i32)(32000 - equiv_rate) / (equiv_rate - 14000);
   if (st->channels == 2) {
      if (st->hybrid_stereo_width_Q14 < (1 << 14) || st->stereoWidth_Q14 < (1 << 14)) {
         i16 g1 = (i16)st->hybrid_stereo_width_Q14, g2 = (i16)st->stereoWidth_Q14;
         g1 = g1 == 16384 ? Q15ONE : shl16(g1, 1);
         g2 = g2 == 16384 ? Q15ONE : shl16(g2, 1);
         stereo_fade(pcm_buf, pcm_buf, g1, g2, OVERLAP, frame_size, st->channels, Fs);
         st->hybrid_stereo_width_Q14 = st->stereoWidth_Q14;
      }
   }
   int nb_compr_bytes = (max_data_bytes - 1);
   oc_ec_enc_shrink(&enc, nb_compr_bytes);
   st->celt.start = 0;
   data[-1] = 0;
   st->celt.vbr = st->use_vbr;
   if (st->use_vbr) {
      st->celt.vbr = 1;
      st->celt.constrained_vbr = st->vbr_constraint;
      if (st->bitrate_bps > 500) st->celt.bitrate = imin(st->bitrate_bps, 750000 * st->channels);
   }
   ret = 0;
   if (oc_ec_tell(&enc) <= 8 * nb_compr_bytes) {
      ret = oc_celt_encode_with_ec(&st->celt, pcm_buf, frame_size, 0, nb_compr_bytes, &enc);
      if (ret < 0) return -3;
   }
   st->rangeFinal = st->celt.rng;
   data--;
   data[0] |= gen_toc_celt(Fs / frame_size, curr_bandwidth, st->stream_channels);
   st->prev_mode = 1002;
   st->prev_channels = st->stream_channels;
   st->prev_framesize = frame_size;
   st->first = 0;
   if (oc_ec_tell(&enc) > (max_data_bytes - 1) * 8) {
      if (max_data_bytes < 2) return -2;
      data[1] = 0;
      ret = 1;
      st->rangeFinal = 0;
   }
   ret += 1;
   if (!st->use_vbr) {          /* apply_padding, src/opus_encoder.c:2602/:2646 */
      if (packet_pad1(data, ret, orig_max_data_bytes) != 0) return -3;
      ret = orig_max_data_bytes;
   }
   return ret;
}

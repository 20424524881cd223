/* oc_opus_dec.c — Opus-level decode of CELT-only packets at 48 kHz: oracle restatement of src/opus.c:203
 * (opus_packet_get_samples_per_frame), :224 (opus_packet_parse_impl, not self-delimited), src/opus_decoder.c:130
 * (opus_decoder_init), :271 (opus_decode_frame, CELT-only branch without transitions/redundancy), :716
 * (opus_decode_native).  SILK / hybrid packets, PLC and FEC return -5.  TEST INFRASTRUCTURE. */
#include "oc_celt_dec.h"
#include <string.h>

#define MODE_SILK_ONLY 1000
#define MODE_HYBRID 1001
#define MODE_CELT_ONLY 1002

int oc_opus_dec_size(void) { return (int)sizeof(oc_opus_dec); }
int oc_opus_dec_init(oc_opus_dec *st, int Fs, int channels)
{
   if (Fs != 48000 || (channels != 1 && channels != 2)) return -1;
   memset(st, 0, sizeof(*st));
   st->Fs = Fs; st->channels = st->stream_channels = channels;
   st->frame_size = Fs / 400;
   oc_celt_dec_init(&st->celt, channels);
   return 0;
}
u32 oc_opus_dec_final_range(const oc_opus_dec *st) { return st->rangeFinal; }

static int samples_per_frame(const u8 *data, i32 Fs)
{
   int audiosize;
   if (data[0] & 0x80) { audiosize = ((data[0] >> 3) & 0x3); audiosize = (Fs << audiosize) / 400; }
   else if ((data[0] & 0x60) == 0x60) audiosize = (data[0] & 0x08) ? Fs / 50 : Fs / 100;
   else { audiosize = ((data[0] >> 3) & 0x3); audiosize = audiosize == 3 ? Fs * 60 / 1000 : (Fs << audiosize) / 100; }
   return audiosize;
}
static int parse_size(const u8 *data, i32 len, i16 *size)
{
   if (len < 1) { *size = -1; return -1; }
   else if (data[0] < 252) { *size = data[0]; return 1; }
   else if (len < 2) { *size = -1; return -1; }
   else { *size = 4 * data[1] + data[0]; return 2; }
}
/* opus_packet_parse_impl, opus.c:224 with self_delimited = 0 */
int oc_opus_packet_parse(const u8 *data, int len, u8 *out_toc, i16 size[48], int *payload_offset)
{
   int i, bytes, count, cbr = 0, framesize;
   u8 ch, toc;
   i32 last_size, pad = 0;
   const u8 *data0 = data;
   if (size == 0 || len < 0) return -1;
   if (len == 0) return -4;
   framesize = samples_per_frame(data, 48000);
   toc = *data++;
   len--;
   last_size = len;
   switch (toc & 0x3) {
   case 0: count = 1; break;
   case 1:
      count = 2; cbr = 1;
      if (len & 0x1) return -4;
      last_size = len / 2;
      size[0] = (i16)last_size;
      break;
   case 2:
      count = 2;
      bytes = parse_size(data, len, size);
      len -= bytes;
      if (size[0] < 0 || size[0] > len) return -4;
      data += bytes;
      last_size = len - size[0];
      break;
   default:
      if (len < 1) return -4;
      ch = *data++;
      count = ch & 0x3F;
      if (count <= 0 || framesize * (i32)count > 5760) return -4;
      len--;
      if (ch & 0x40) {
         int p;
         do {
            int tmp;
            if (len <= 0) return -4;
            p = *data++;
            len--;
            tmp = p == 255 ? 254 : p;
            len -= tmp;
            pad += tmp;
         } while (p == 255);
      }
      if (len < 0) return -4;
      cbr = !(ch & 0x80);
      if (!cbr) {
         last_size = len;
         for (i = 0; i < count - 1; i++) {
            bytes = parse_size(data, len, size + i);
            len -= bytes;
            if (size[i] < 0 || size[i] > len) return -4;
            data += bytes;
            last_size -= bytes + size[i];
         }
         if (last_size < 0) return -4;
      } else {
         last_size = len / count;
         if (last_size * count != len) return -4;
         for (i = 0; i < count - 1; i++) size[i] = (i16)last_size;
      }
      break;
   }
   if (last_size > 1275) return -4;
   size[count - 1] = (i16)last_size;
   if (payload_offset) *payload_offset = (int)(data - data0);
   if (out_toc) *out_toc = toc;
   (void)pad; (void)cbr;
   return count;
}

/* opus_decode_frame, opus_decoder.c:271: CELT-only; data == NULL (or a frame of <= 1 byte) runs the concealment */
static int decode_frame(oc_opus_dec *st, const u8 *data, i32 len, i16 *pcm, int frame_size)
{
   const int F20 = st->Fs / 50, F10 = F20 >> 1, F5 = F10 >> 1, F2_5 = F5 >> 1;
   oc_ec dec;
   int audiosize, mode;
   if (frame_size < F2_5) return -2;
   frame_size = imin(frame_size, st->Fs / 25 * 3);
   if (len <= 1) { data = 0; frame_size = imin(frame_size, st->frame_size); }
   if (data != 0) {
      audiosize = st->frame_size;
      mode = st->mode;
      oc_ec_dec_init(&dec, data, len);
   } else {
      audiosize = frame_size;
      mode = st->prev_redundancy ? MODE_CELT_ONLY : st->prev_mode;
      if (mode == 0) { for (int i = 0; i < audiosize * st->channels; i++) pcm[i] = 0; return audiosize; }
      if (audiosize > F20) {
         do {
            int ret = decode_frame(st, 0, 0, pcm, imin(audiosize, F20));
            if (ret < 0) return ret;
            pcm += ret * st->channels;
            audiosize -= ret;
         } while (audiosize > 0);
         return frame_size;
      } else if (audiosize < F20) {
         if (audiosize > F10) audiosize = F10;
         else if (mode != MODE_SILK_ONLY && audiosize > F5 && audiosize < F10) audiosize = F5;
      }
   }
   if (mode != MODE_CELT_ONLY) return -5;
   if (data != 0 && st->prev_mode > 0 && st->prev_mode != MODE_CELT_ONLY) return -5;    /* transitions need SILK */
   if (audiosize > frame_size) return -1;
   frame_size = audiosize;
   if (st->bandwidth) {
      int endband = 21;
      switch (st->bandwidth) {
      case 1101: endband = 13; break;
      case 1102: case 1103: endband = 17; break;
      case 1104: endband = 19; break;
      case 1105: endband = 21; break;
      }
      st->celt.end = endband;
   }
   st->celt.stream_channels = st->stream_channels;
   st->celt.start = 0;
   int celt_ret = oc_celt_decode_with_ec(&st->celt, data, len, pcm, imin(F20, frame_size), data ? &dec : 0);
   st->rangeFinal = st->celt.rng;
   st->prev_mode = mode;
   st->prev_redundancy = 0;
   if (len <= 1) st->rangeFinal = 0;
   return celt_ret < 0 ? celt_ret : audiosize;
}

/* opus_decode_native, opus_decoder.c:716 */
int oc_opus_decode(oc_opus_dec *st, const u8 *data, int len, i16 *pcm, int frame_size, int decode_fec)
{
   i16 size[48];
   u8 toc;
   int offset, count, nb_samples = 0;
   if (frame_size <= 0) return -1;
   if (decode_fec < 0 || decode_fec > 1) return -1;
   if ((decode_fec || len == 0 || data == 0) && frame_size % (st->Fs / 400) != 0) return -1;
   if (len == 0 || data == 0) {                               /* packet loss: conceal frame_size samples */
      int pcm_count = 0;
      do {
         int ret = decode_frame(st, 0, 0, pcm + pcm_count * st->channels, frame_size - pcm_count);
         if (ret < 0) return ret;
         pcm_count += ret;
      } while (pcm_count < frame_size);
      st->last_packet_duration = pcm_count;
      return pcm_count;
   }
   if (len < 0) return -1;
   if (decode_fec) return -5;
   int packet_mode = (data[0] & 0x80) ? MODE_CELT_ONLY : ((data[0] & 0x60) == 0x60 ? MODE_HYBRID : MODE_SILK_ONLY);
   int packet_bandwidth;
   if (data[0] & 0x80) { packet_bandwidth = 1102 + ((data[0] >> 5) & 0x3); if (packet_bandwidth == 1102) packet_bandwidth = 1101; }
   else if ((data[0] & 0x60) == 0x60) packet_bandwidth = (data[0] & 0x10) ? 1105 : 1104;
   else packet_bandwidth = 1101 + ((data[0] >> 5) & 0x3);
   int packet_frame_size = samples_per_frame(data, st->Fs);
   int packet_stream_channels = (data[0] & 0x4) ? 2 : 1;
   count = oc_opus_packet_parse(data, len, &toc, size, &offset);
   if (count < 0) return count;
   if (packet_mode != MODE_CELT_ONLY) return -5;
   data += offset;
   if (count * packet_frame_size > frame_size) return -2;
   st->mode = packet_mode; st->bandwidth = packet_bandwidth; st->frame_size = packet_frame_size; st->stream_channels = packet_stream_channels;
   for (int i = 0; i < count; i++) {
      int ret = decode_frame(st, data, size[i], pcm + nb_samples * st->channels, frame_size - nb_samples);
      if (ret < 0) return ret;
      data += size[i];
      nb_samples += ret;
   }
   st->last_packet_duration = nb_samples;
   return nb_samples;
}

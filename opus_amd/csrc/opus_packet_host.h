/* opus_packet_host.h — host-side packet toolkit of the libopus ABI (pure CPU control code in the reference as well):
 * opus_packet_get_* / opus_packet_parse (reference src/opus.c:203-399, src/opus_decoder.c:1252-1340), OpusRepacketizer and
 * packet (un)padding incl. the multistream variants (src/repacketizer.c:36-475).  Padding *extensions* (opus_extension_data,
 * DRED/QEXT payloads carried in code-3 padding) are not interpreted: padding is dropped when frames are re-assembled. */
#ifndef OPUS_AMD_PACKET_HOST_H
#define OPUS_AMD_PACKET_HOST_H

static int oa_encode_size(int size, unsigned char *data)
{
   if (size < 252) { data[0] = (unsigned char)size; return 1; }
   data[0] = (unsigned char)(252 + (size & 0x3));
   data[1] = (unsigned char)((size - (int)data[0]) >> 2);
   return 2;
}
static int oa_parse_size_h(const unsigned char *data, opus_int32 len, opus_int16 *size)
{
   if (len < 1) { *size = -1; return -1; }
   else if (data[0] < 252) { *size = data[0]; return 1; }
   else if (len < 2) { *size = -1; return -1; }
   else { *size = (opus_int16)(4 * data[1] + data[0]); return 2; }
}
/* opus_packet_parse_impl, src/opus.c:224 */
static int oa_packet_parse_impl(const unsigned char *data, opus_int32 len, int self_delimited, unsigned char *out_toc, const unsigned char *frames[48],
      opus_int16 size[48], int *payload_offset, opus_int32 *packet_offset)
{
   int i, bytes, count, cbr = 0, framesize;
   unsigned char ch, toc;
   opus_int32 last_size, pad = 0;
   const unsigned char *data0 = data;
   if (size == NULL || len < 0) return OPUS_BAD_ARG;
   if (len == 0) return OPUS_INVALID_PACKET;
   framesize = opus_packet_get_samples_per_frame(data, 48000);
   toc = *data++;
   len--;
   last_size = len;
   switch (toc & 0x3) {
   case 0: count = 1; break;
   case 1:
      count = 2; cbr = 1;
      if (!self_delimited) {
         if (len & 0x1) return OPUS_INVALID_PACKET;
         last_size = len / 2;
         size[0] = (opus_int16)last_size;
      }
      break;
   case 2:
      count = 2;
      bytes = oa_parse_size_h(data, len, size);
      len -= bytes;
      if (size[0] < 0 || size[0] > len) return OPUS_INVALID_PACKET;
      data += bytes;
      last_size = len - size[0];
      break;
   default:
      if (len < 1) return OPUS_INVALID_PACKET;
      ch = *data++;
      count = ch & 0x3F;
      if (count <= 0 || framesize * (opus_int32)count > 5760) return OPUS_INVALID_PACKET;
      len--;
      if (ch & 0x40) {
         int p;
         do {
            int tmp;
            if (len <= 0) return OPUS_INVALID_PACKET;
            p = *data++;
            len--;
            tmp = p == 255 ? 254 : p;
            len -= tmp;
            pad += tmp;
         } while (p == 255);
      }
      if (len < 0) return OPUS_INVALID_PACKET;
      cbr = !(ch & 0x80);
      if (!cbr) {
         last_size = len;
         for (i = 0; i < count - 1; i++) {
            bytes = oa_parse_size_h(data, len, size + i);
            len -= bytes;
            if (size[i] < 0 || size[i] > len) return OPUS_INVALID_PACKET;
            data += bytes;
            last_size -= bytes + size[i];
         }
         if (last_size < 0) return OPUS_INVALID_PACKET;
      } else if (!self_delimited) {
         last_size = len / count;
         if (last_size * count != len) return OPUS_INVALID_PACKET;
         for (i = 0; i < count - 1; i++) size[i] = (opus_int16)last_size;
      }
      break;
   }
   if (self_delimited) {
      bytes = oa_parse_size_h(data, len, size + count - 1);
      len -= bytes;
      if (size[count - 1] < 0 || size[count - 1] > len) return OPUS_INVALID_PACKET;
      data += bytes;
      if (cbr) {
         if (size[count - 1] * count > len) return OPUS_INVALID_PACKET;
         for (i = 0; i < count - 1; i++) size[i] = size[count - 1];
      } else if (bytes + size[count - 1] > last_size) return OPUS_INVALID_PACKET;
   } else {
      if (last_size > 1275) return OPUS_INVALID_PACKET;
      size[count - 1] = (opus_int16)last_size;
   }
   if (payload_offset) *payload_offset = (int)(data - data0);
   for (i = 0; i < count; i++) {
      if (frames) frames[i] = data;
      data += size[i];
   }
   if (packet_offset) *packet_offset = pad + (opus_int32)(data - data0);
   if (out_toc) *out_toc = toc;
   return count;
}

struct OpusRepacketizer {
   unsigned char toc;
   int nb_frames;
   const unsigned char *frames[48];
   opus_int16 len[48];
   int framesize;
};

static int oa_repacketizer_cat_impl(OpusRepacketizer *rp, const unsigned char *data, opus_int32 len, int self_delimited)
{
   unsigned char tmp_toc;
   int curr_nb_frames, ret;
   if (len < 1) return OPUS_INVALID_PACKET;
   if (rp->nb_frames == 0) { rp->toc = data[0]; rp->framesize = opus_packet_get_samples_per_frame(data, 8000); }
   else if ((rp->toc & 0xFC) != (data[0] & 0xFC)) return OPUS_INVALID_PACKET;
   curr_nb_frames = opus_packet_get_nb_frames(data, len);
   if (curr_nb_frames < 1) return OPUS_INVALID_PACKET;
   if ((curr_nb_frames + rp->nb_frames) * rp->framesize > 960) return OPUS_INVALID_PACKET;
   ret = oa_packet_parse_impl(data, len, self_delimited, &tmp_toc, &rp->frames[rp->nb_frames], &rp->len[rp->nb_frames], NULL, NULL);
   if (ret < 1) return ret;
   rp->nb_frames += curr_nb_frames;
   return OPUS_OK;
}
/* opus_repacketizer_out_range_impl, src/repacketizer.c:112 (no extensions) */
static opus_int32 oa_repacketizer_out_range_impl(OpusRepacketizer *rp, int begin, int end, unsigned char *data, opus_int32 maxlen, int self_delimited, int pad)
{
   int i, count;
   opus_int32 tot_size;
   opus_int16 *len;
   const unsigned char **frames;
   unsigned char *ptr;
   if (begin < 0 || begin >= end || end > rp->nb_frames) return OPUS_BAD_ARG;
   count = end - begin;
   len = rp->len + begin;
   frames = rp->frames + begin;
   tot_size = self_delimited ? 1 + (len[count - 1] >= 252) : 0;
   ptr = data;
   if (count == 1) {
      tot_size += len[0] + 1;
      if (tot_size > maxlen) return OPUS_BUFFER_TOO_SMALL;
      *ptr++ = rp->toc & 0xFC;
   } else if (count == 2) {
      if (len[1] == len[0]) {
         tot_size += 2 * len[0] + 1;
         if (tot_size > maxlen) return OPUS_BUFFER_TOO_SMALL;
         *ptr++ = (rp->toc & 0xFC) | 0x1;
      } else {
         tot_size += len[0] + len[1] + 2 + (len[0] >= 252);
         if (tot_size > maxlen) return OPUS_BUFFER_TOO_SMALL;
         *ptr++ = (rp->toc & 0xFC) | 0x2;
         ptr += oa_encode_size(len[0], ptr);
      }
   }
   if (count > 2 || (pad && tot_size < maxlen)) {
      int vbr = 0, pad_amount = 0;
      ptr = data;
      tot_size = self_delimited ? 1 + (len[count - 1] >= 252) : 0;
      for (i = 1; i < count; i++) if (len[i] != len[0]) { vbr = 1; break; }
      if (vbr) {
         tot_size += 2;
         for (i = 0; i < count - 1; i++) tot_size += 1 + (len[i] >= 252) + len[i];
         tot_size += len[count - 1];
         if (tot_size > maxlen) return OPUS_BUFFER_TOO_SMALL;
         *ptr++ = (rp->toc & 0xFC) | 0x3;
         *ptr++ = (unsigned char)(count | 0x80);
      } else {
         tot_size += count * len[0] + 2;
         if (tot_size > maxlen) return OPUS_BUFFER_TOO_SMALL;
         *ptr++ = (rp->toc & 0xFC) | 0x3;
         *ptr++ = (unsigned char)count;
      }
      pad_amount = pad ? (maxlen - tot_size) : 0;
      if (pad_amount != 0) {
         int nb_255s;
         data[1] |= 0x40;
         nb_255s = (pad_amount - 1) / 255;
         if (tot_size + nb_255s + 1 > maxlen) return OPUS_BUFFER_TOO_SMALL;
         for (i = 0; i < nb_255s; i++) *ptr++ = 255;
         *ptr++ = (unsigned char)(pad_amount - 255 * nb_255s - 1);
         tot_size += pad_amount;
      }
      if (vbr) for (i = 0; i < count - 1; i++) ptr += oa_encode_size(len[i], ptr);
   }
   if (self_delimited) ptr += oa_encode_size(len[count - 1], ptr);
   for (i = 0; i < count; i++) { memmove(ptr, frames[i], (size_t)len[i]); ptr += len[i]; }
   if (pad) while (ptr < data + maxlen) *ptr++ = 0;
   return tot_size;
}

extern "C" {
int opus_packet_get_samples_per_frame(const unsigned char *data, opus_int32 Fs)
{
   int audiosize;
   if (data[0] & 0x80) { audiosize = ((data[0] >> 3) & 0x3); audiosize = (Fs << audiosize) / 400; }
   else if ((data[0] & 0x60) == 0x60) audiosize = (data[0] & 0x08) ? Fs / 50 : Fs / 100;
   else { audiosize = ((data[0] >> 3) & 0x3); audiosize = audiosize == 3 ? Fs * 60 / 1000 : (Fs << audiosize) / 100; }
   return audiosize;
}
int opus_packet_get_bandwidth(const unsigned char *data)
{
   int bandwidth;
   if (data[0] & 0x80) { bandwidth = OPUS_BANDWIDTH_MEDIUMBAND + ((data[0] >> 5) & 0x3); if (bandwidth == OPUS_BANDWIDTH_MEDIUMBAND) bandwidth = OPUS_BANDWIDTH_NARROWBAND; }
   else if ((data[0] & 0x60) == 0x60) bandwidth = (data[0] & 0x10) ? OPUS_BANDWIDTH_FULLBAND : OPUS_BANDWIDTH_SUPERWIDEBAND;
   else bandwidth = OPUS_BANDWIDTH_NARROWBAND + ((data[0] >> 5) & 0x3);
   return bandwidth;
}
int opus_packet_get_nb_channels(const unsigned char *data) { return (data[0] & 0x4) ? 2 : 1; }
int opus_packet_get_nb_frames(const unsigned char packet[], opus_int32 len)
{
   int count;
   if (len < 1) return OPUS_BAD_ARG;
   count = packet[0] & 0x3;
   if (count == 0) return 1;
   else if (count != 3) return 2;
   else if (len < 2) return OPUS_INVALID_PACKET;
   else return packet[1] & 0x3F;
}
int opus_packet_get_nb_samples(const unsigned char packet[], opus_int32 len, opus_int32 Fs)
{
   int count = opus_packet_get_nb_frames(packet, len);
   if (count < 0) return count;
   int samples = count * opus_packet_get_samples_per_frame(packet, Fs);
   if (samples * 25 > Fs * 3) return OPUS_INVALID_PACKET;
   return samples;
}
int opus_packet_parse(const unsigned char *data, opus_int32 len, unsigned char *out_toc, const unsigned char *frames[48], opus_int16 size[48], int *payload_offset)
{
   return oa_packet_parse_impl(data, len, 0, out_toc, frames, size, payload_offset, NULL);
}
int opus_repacketizer_get_size(void) { return (int)sizeof(OpusRepacketizer); }
OpusRepacketizer *opus_repacketizer_init(OpusRepacketizer *rp) { rp->nb_frames = 0; return rp; }
OpusRepacketizer *opus_repacketizer_create(void)
{
   OpusRepacketizer *rp = (OpusRepacketizer *)malloc(sizeof(OpusRepacketizer));
   if (rp == NULL) return NULL;
   return opus_repacketizer_init(rp);
}
void opus_repacketizer_destroy(OpusRepacketizer *rp) { free(rp); }
int opus_repacketizer_cat(OpusRepacketizer *rp, const unsigned char *data, opus_int32 len) { return oa_repacketizer_cat_impl(rp, data, len, 0); }
int opus_repacketizer_get_nb_frames(OpusRepacketizer *rp) { return rp->nb_frames; }
opus_int32 opus_repacketizer_out_range(OpusRepacketizer *rp, int begin, int end, unsigned char *data, opus_int32 maxlen)
{
   return oa_repacketizer_out_range_impl(rp, begin, end, data, maxlen, 0, 0);
}
opus_int32 opus_repacketizer_out(OpusRepacketizer *rp, unsigned char *data, opus_int32 maxlen)
{
   return oa_repacketizer_out_range_impl(rp, 0, rp->nb_frames, data, maxlen, 0, 0);
}
int opus_packet_pad(unsigned char *data, opus_int32 len, opus_int32 new_len)
{
   OpusRepacketizer rp;
   if (len < 1) return OPUS_BAD_ARG;
   if (len == new_len) return OPUS_OK;
   else if (len > new_len) return OPUS_BAD_ARG;
   std::vector<unsigned char> copy(data, data + len);
   opus_repacketizer_init(&rp);
   opus_int32 ret = opus_repacketizer_cat(&rp, copy.data(), len);
   if (ret != OPUS_OK) return ret;
   ret = oa_repacketizer_out_range_impl(&rp, 0, rp.nb_frames, data, new_len, 0, 1);
   return ret > 0 ? OPUS_OK : ret;
}
opus_int32 opus_packet_unpad(unsigned char *data, opus_int32 len)
{
   OpusRepacketizer rp;
   if (len < 1) return OPUS_BAD_ARG;
   opus_repacketizer_init(&rp);
   opus_int32 ret = opus_repacketizer_cat(&rp, data, len);
   if (ret < 0) return ret;
   return oa_repacketizer_out_range_impl(&rp, 0, rp.nb_frames, data, len, 0, 0);
}
int opus_multistream_packet_pad(unsigned char *data, opus_int32 len, opus_int32 new_len, int nb_streams)
{
   unsigned char toc;
   opus_int16 size[48];
   opus_int32 packet_offset;
   if (len < 1) return OPUS_BAD_ARG;
   if (len == new_len) return OPUS_OK;
   else if (len > new_len) return OPUS_BAD_ARG;
   opus_int32 amount = new_len - len;
   for (int s = 0; s < nb_streams - 1; s++) {
      if (len <= 0) return OPUS_INVALID_PACKET;
      int count = oa_packet_parse_impl(data, len, 1, &toc, NULL, size, NULL, &packet_offset);
      if (count < 0) return count;
      data += packet_offset;
      len -= packet_offset;
   }
   return opus_packet_pad(data, len, len + amount);
}
opus_int32 opus_multistream_packet_unpad(unsigned char *data, opus_int32 len, int nb_streams)
{
   unsigned char toc;
   opus_int16 size[48];
   opus_int32 packet_offset, dst_len = 0;
   OpusRepacketizer rp;
   unsigned char *dst = data;
   if (len < 1) return OPUS_BAD_ARG;
   for (int s = 0; s < nb_streams; s++) {
      int self_delimited = s != nb_streams - 1;
      if (len <= 0) return OPUS_INVALID_PACKET;
      opus_repacketizer_init(&rp);
      opus_int32 ret = oa_packet_parse_impl(data, len, self_delimited, &toc, NULL, size, NULL, &packet_offset);
      if (ret < 0) return ret;
      ret = oa_repacketizer_cat_impl(&rp, data, packet_offset, self_delimited);
      if (ret < 0) return ret;
      ret = oa_repacketizer_out_range_impl(&rp, 0, rp.nb_frames, dst, len, self_delimited, 0);
      if (ret < 0) return ret;
      dst_len += ret;
      dst += ret;
      data += packet_offset;
      len -= packet_offset;
   }
   return dst_len;
}
} /* extern "C" */
#endif

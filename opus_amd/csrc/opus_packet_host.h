/* opus_packet_host.h — host-side packet toolkit of the libopus ABI, written from the packet format itself (RFC 6716 §3.1 TOC byte, §3.2 frame
 * packing codes 0-3 and the 1/2-byte frame length coding, §3.4 the malformed-packet rules R1-R7, Appendix B self-delimiting framing).
 * Entry points and their error codes are the ones of include/opus.h:571-800 (opus_packet_parse, opus_packet_get_*), :1004-1167 (OpusRepacketizer,
 * opus_packet_pad / unpad and the multistream variants); behaviour is checked against the compiled reference in tests/test_packet_toolkit.py and by the
 * reference's own tests/test_opus_api.c (test_parse, test_repacketizer_api).
 *
 * Design: a packet is first turned into an OaFrameMap (where each frame sits, what the header said); every query is a read of that map, and every
 * writer (repacketizer, padding) is one call of oa_frames_emit(), which picks the cheapest code that can carry the frames.  Padding *extensions*
 * (DRED / QEXT payloads inside code-3 padding) are not interpreted: padding is dropped when frames are re-assembled. */
#ifndef OPUS_AMD_PACKET_HOST_H
#define OPUS_AMD_PACKET_HOST_H

#define OA_MAX_FRAMES 48                       /* 120 ms of 2.5 ms frames */
#define OA_MAX_FRAME_BYTES 1275

/* ---- TOC byte (RFC 6716 §3.1): config(5) | s(1) | c(2) ---- */
static inline int oa_toc_config(unsigned char toc) { return toc >> 3; }
/* samples per frame at 48 kHz from the 5-bit config: SILK 10/20/40/60 ms (configs 0-11), hybrid 10/20 ms (12-15), CELT 2.5/5/10/20 ms (16-31) */
static int oa_toc_frame_48k(unsigned char toc)
{
   const int cfg = oa_toc_config(toc);
   if (cfg >= 16) return 120 << (cfg & 3);
   if (cfg >= 12) return (cfg & 1) ? 960 : 480;
   return (cfg & 3) == 3 ? 2880 : 480 << (cfg & 3);
}

/* ---- frame length coding (§3.2.1): 0..251 in one byte, 252..1275 as 252+(n&3), (n-first)/4 ---- */
struct OaCursor {
   const unsigned char *p; opus_int32 left;
   bool byte(int *v) { if (left < 1) return false; *v = *p++; left--; return true; }
   /* one coded length; false when the bytes are not there */
   bool length(int *n) { int a, b; if (!byte(&a)) return false; if (a < 252) { *n = a; return true; } if (!byte(&b)) return false; *n = 4 * b + a; return true; }
};
static int oa_put_length(int n, unsigned char *dst)
{
   if (n < 252) { dst[0] = (unsigned char)n; return 1; }
   dst[0] = (unsigned char)(252 + (n & 3)); dst[1] = (unsigned char)((n - dst[0]) >> 2);
   return 2;
}
static inline int oa_length_bytes(int n) { return n < 252 ? 1 : 2; }

struct OaFrameMap {
   unsigned char toc;
   int count;                                  /* frames in the packet */
   opus_int32 offset[OA_MAX_FRAMES];           /* of each frame from the start of the packet */
   opus_int16 size[OA_MAX_FRAMES];
   opus_int32 padding;                         /* code-3 padding bytes (without their length bytes) */
   opus_int32 end;                             /* first byte after the packet: last frame end + padding (== len unless self-delimited) */
};

/* Builds the map; returns the frame count or OPUS_INVALID_PACKET.  `framed` = Appendix B self-delimiting variant (the length of the last frame is coded too). */
static int oa_frame_map(const unsigned char *pkt, opus_int32 len, bool framed, OaFrameMap *m)
{
   if (len < 0) return OPUS_BAD_ARG;
   if (len == 0) return OPUS_INVALID_PACKET;                                                     /* R1 */
   OaCursor c = {pkt + 1, len - 1};
   m->toc = pkt[0]; m->padding = 0;
   const int code = pkt[0] & 3;
   bool vbr = false;
   int n = 1, v;
   int sz[OA_MAX_FRAMES];
   if (code == 1) n = 2;
   else if (code == 2) { n = 2; vbr = true; if (!c.length(&sz[0])) return OPUS_INVALID_PACKET; }
   else if (code == 3) {
      if (!c.byte(&v)) return OPUS_INVALID_PACKET;                                                /* R6: the frame count byte must exist */
      n = v & 0x3F; vbr = (v & 0x80) != 0;
      if (n == 0 || (opus_int32)n * oa_toc_frame_48k(pkt[0]) > 5760) return OPUS_INVALID_PACKET; /* R5: at most 120 ms */
      if (v & 0x40) {                                                                             /* padding: 255 means "254 more, and another length byte follows" */
         int b;
         do { if (!c.byte(&b)) return OPUS_INVALID_PACKET; m->padding += b == 255 ? 254 : b; } while (b == 255);
         c.left -= m->padding;
         if (c.left < 0) return OPUS_INVALID_PACKET;
      }
      if (vbr) for (int i = 0; i < n - 1; i++) if (!c.length(&sz[i])) return OPUS_INVALID_PACKET;
   }
   /* what the explicitly sized frames leave for the rest */
   opus_int32 rest = c.left;
   if (vbr) for (int i = 0; i < n - 1; i++) { rest -= sz[i]; if (rest < 0) return OPUS_INVALID_PACKET; }     /* R3 / R7 */
   if (framed) {
      int last;
      OaCursor t = c;
      if (!c.length(&last)) return OPUS_INVALID_PACKET;
      rest -= (opus_int32)(t.left - c.left);
      if (vbr || n == 1) { if (last > rest) return OPUS_INVALID_PACKET; sz[n - 1] = last; }
      else { if ((opus_int32)last * n > rest) return OPUS_INVALID_PACKET; for (int i = 0; i < n; i++) sz[i] = last; }
   } else if (vbr || n == 1) {
      sz[n - 1] = rest;
   } else {                                                                                      /* CBR: the payload splits evenly (R3, R6) */
      if (rest % n) return OPUS_INVALID_PACKET;
      for (int i = 0; i < n; i++) sz[i] = rest / n;
   }
   opus_int32 at = (opus_int32)(c.p - pkt);
   for (int i = 0; i < n; i++) {
      if (sz[i] > OA_MAX_FRAME_BYTES) return OPUS_INVALID_PACKET;                                 /* R2 */
      m->offset[i] = at; m->size[i] = (opus_int16)sz[i]; at += sz[i];
   }
   m->count = n; m->end = at + m->padding;
   return n;
}

/* Writes frames as one packet with the smallest header that can carry them: code 0 (one frame), 1 (two equal), 2 (two different), 3 (anything else, or
 * whenever the packet has to be padded out to exactly maxlen).  Frame data may alias `out` as long as every frame starts at or after its destination
 * (un-padding in place).  Returns the packet size or OPUS_BUFFER_TOO_SMALL. */
static opus_int32 oa_frames_emit(unsigned char toc, int n, const unsigned char *const *frame, const opus_int16 *size, unsigned char *out, opus_int32 maxlen, bool framed, bool fill)
{
   bool same = true;
   opus_int32 body = 0;
   for (int i = 0; i < n; i++) { body += size[i]; same = same && size[i] == size[0]; }
   const int tail = framed ? oa_length_bytes(size[n - 1]) : 0;
   toc &= 0xFC;
   /* pass 1: the shape of the header (nothing is written yet: the padding length run is as long as the caller's maxlen makes it) */
   opus_int32 h = 0, total = 0, pad = 0, full = 0;
   bool code3 = n > 2;
   if (!code3) {
      h = 1 + (n == 2 && !same ? oa_length_bytes(size[0]) : 0);
      total = h + tail + body;
      if (total > maxlen) return OPUS_BUFFER_TOO_SMALL;
      code3 = fill && total < maxlen;
   }
   if (code3) {
      h = 2;
      if (!same) for (int i = 0; i < n - 1; i++) h += oa_length_bytes(size[i]);
      total = h + tail + body;
      if (total > maxlen) return OPUS_BUFFER_TOO_SMALL;
      pad = fill ? maxlen - total : 0;
      if (pad > 0) { full = (pad - 1) / 255; h += full + 1; total = maxlen; }                     /* the length bytes of the padding count as padding */
   }
   h += tail;
   /* pass 2: frames first (destinations never pass their sources, so ascending order is safe; the header cannot clobber frame 0 before it moved
    * because frame 0 starts at >= h in any packet it came from -- except when padding grows the header: callers that pad in place pass a copy),
    * then the header straight into the packet */
   unsigned char *w = out + h;
   for (int i = 0; i < n; i++) { memmove(w, frame[i], (size_t)size[i]); w += size[i]; }
   unsigned char *q = out;
   if (!code3) {
      *q++ = (unsigned char)(toc | (n == 1 ? 0 : same ? 1 : 2));
      if (n == 2 && !same) q += oa_put_length(size[0], q);
   } else {
      *q++ = (unsigned char)(toc | 3);
      *q++ = (unsigned char)(n | (same ? 0 : 0x80) | (pad > 0 ? 0x40 : 0));
      if (pad > 0) { memset(q, 255, (size_t)full); q += full; *q++ = (unsigned char)(pad - 255 * full - 1); }
      if (!same) for (int i = 0; i < n - 1; i++) q += oa_put_length(size[i], q);
   }
   if (framed) q += oa_put_length(size[n - 1], q);
   if (fill && w < out + total) memset(w, 0, (size_t)(out + total - w));
   return total;
}

struct OpusRepacketizer {
   unsigned char toc;
   int nb_frames;
   int frame_8k;                               /* samples per frame at 8 kHz: 120 ms = 960 */
   const unsigned char *frames[OA_MAX_FRAMES];
   opus_int16 len[OA_MAX_FRAMES];
};
static int oa_rp_add(OpusRepacketizer *rp, const unsigned char *data, opus_int32 len, bool framed)
{
   if (len < 1) return OPUS_INVALID_PACKET;
   if (rp->nb_frames == 0) { rp->toc = data[0]; rp->frame_8k = oa_toc_frame_48k(data[0]) / 6; }
   else if ((rp->toc ^ data[0]) & 0xFC) return OPUS_INVALID_PACKET;                              /* all frames of a packet share one configuration */
   OaFrameMap m;
   const int n = oa_frame_map(data, len, framed, &m);
   if (n < 1) return n;
   if ((rp->nb_frames + n) * rp->frame_8k > 960) return OPUS_INVALID_PACKET;
   for (int i = 0; i < n; i++) { rp->frames[rp->nb_frames + i] = data + m.offset[i]; rp->len[rp->nb_frames + i] = m.size[i]; }
   rp->nb_frames += n;
   return OPUS_OK;
}
static opus_int32 oa_rp_emit(OpusRepacketizer *rp, int begin, int end, unsigned char *data, opus_int32 maxlen, bool framed, bool fill)
{
   if (begin < 0 || begin >= end || end > rp->nb_frames) return OPUS_BAD_ARG;
   return oa_frames_emit(rp->toc, end - begin, rp->frames + begin, rp->len + begin, data, maxlen, framed, fill);
}
/* names the multistream layer uses */
static int oa_repacketizer_cat_impl(OpusRepacketizer *rp, const unsigned char *data, opus_int32 len, int self_delimited) { return oa_rp_add(rp, data, len, self_delimited != 0); }
static opus_int32 oa_repacketizer_out_range_impl(OpusRepacketizer *rp, int begin, int end, unsigned char *data, opus_int32 maxlen, int self_delimited, int pad)
{ return oa_rp_emit(rp, begin, end, data, maxlen, self_delimited != 0, pad != 0); }
/* parse with the argument list of the reference's internal opus_packet_parse_impl (src/opus_private.h:200) */
static int oa_packet_parse_impl(const unsigned char *data, opus_int32 len, int self_delimited, unsigned char *out_toc, const unsigned char *frames[48],
      opus_int16 size[48], int *payload_offset, opus_int32 *packet_offset)
{
   if (size == NULL || len < 0) return OPUS_BAD_ARG;
   OaFrameMap m;
   const int n = oa_frame_map(data, len, self_delimited != 0, &m);
   if (n < 0) return n;
   for (int i = 0; i < n; i++) { size[i] = m.size[i]; if (frames) frames[i] = data + m.offset[i]; }
   if (payload_offset) *payload_offset = (int)m.offset[0];
   if (packet_offset) *packet_offset = m.end;
   if (out_toc) *out_toc = m.toc;
   return n;
}

extern "C" {
int opus_packet_get_samples_per_frame(const unsigned char *data, opus_int32 Fs) { return (int)((long long)oa_toc_frame_48k(data[0]) * Fs / 48000); }
int opus_packet_get_bandwidth(const unsigned char *data)
{
   const int cfg = oa_toc_config(data[0]);
   if (cfg >= 16) { static const int bw[4] = {OPUS_BANDWIDTH_NARROWBAND, OPUS_BANDWIDTH_WIDEBAND, OPUS_BANDWIDTH_SUPERWIDEBAND, OPUS_BANDWIDTH_FULLBAND}; return bw[(cfg - 16) >> 2]; }
   if (cfg >= 12) return cfg >= 14 ? OPUS_BANDWIDTH_FULLBAND : OPUS_BANDWIDTH_SUPERWIDEBAND;
   return OPUS_BANDWIDTH_NARROWBAND + (cfg >> 2);
}
int opus_packet_get_nb_channels(const unsigned char *data) { return (data[0] & 0x4) ? 2 : 1; }
int opus_packet_get_nb_frames(const unsigned char packet[], opus_int32 len)
{
   if (len < 1) return OPUS_BAD_ARG;
   const int code = packet[0] & 3;
   if (code != 3) return code == 0 ? 1 : 2;
   return len < 2 ? OPUS_INVALID_PACKET : packet[1] & 0x3F;
}
int opus_packet_get_nb_samples(const unsigned char packet[], opus_int32 len, opus_int32 Fs)
{
   const int n = opus_packet_get_nb_frames(packet, len);
   if (n < 0) return n;
   const int samples = n * opus_packet_get_samples_per_frame(packet, Fs);
   return samples * 25 > Fs * 3 ? OPUS_INVALID_PACKET : samples;                                 /* more than 120 ms */
}
/* include/opus.h:778: does the packet carry an in-band FEC (LBRR) copy?  CELT-only packets never do; otherwise the flags follow the VAD flags at the
 * head of the first SILK frame (RFC 6716 §4.2.3, §4.2.4: one VAD bit per 20 ms SILK frame, then the LBRR flag, per channel). */
int opus_packet_has_lbrr(const unsigned char packet[], opus_int32 len)
{
   if (len >= 1 && (packet[0] & 0x80)) return 0;                                                  /* (the parse below reports empty / malformed packets) */
   OaFrameMap m;
   const int n = oa_frame_map(packet, len, false, &m);
   if (n <= 0) return n;
   if (m.size[0] == 0) return 0;
   const int f48 = oa_toc_frame_48k(packet[0]), silk_frames = f48 > 960 ? f48 / 960 : 1, nch = opus_packet_get_nb_channels(packet);
   const unsigned char b0 = packet[m.offset[0]];
   int lbrr = (b0 >> (7 - silk_frames)) & 1;
   if (nch == 2) lbrr |= (b0 >> (6 - 2 * silk_frames)) & 1;
   return lbrr;
}
int opus_packet_parse(const unsigned char *data, opus_int32 len, unsigned char *out_toc, const unsigned char *frames[48], opus_int16 size[48], int *payload_offset)
{
   return oa_packet_parse_impl(data, len, 0, out_toc, frames, size, payload_offset, NULL);
}
int opus_repacketizer_get_size(void) { return (int)sizeof(OpusRepacketizer); }
OpusRepacketizer *opus_repacketizer_init(OpusRepacketizer *rp) { rp->nb_frames = 0; return rp; }
OpusRepacketizer *opus_repacketizer_create(void) { OpusRepacketizer *rp = (OpusRepacketizer *)malloc(sizeof(OpusRepacketizer)); return rp ? opus_repacketizer_init(rp) : NULL; }
void opus_repacketizer_destroy(OpusRepacketizer *rp) { free(rp); }
int opus_repacketizer_cat(OpusRepacketizer *rp, const unsigned char *data, opus_int32 len) { return oa_rp_add(rp, data, len, false); }
int opus_repacketizer_get_nb_frames(OpusRepacketizer *rp) { return rp->nb_frames; }
opus_int32 opus_repacketizer_out_range(OpusRepacketizer *rp, int begin, int end, unsigned char *data, opus_int32 maxlen) { return oa_rp_emit(rp, begin, end, data, maxlen, false, false); }
opus_int32 opus_repacketizer_out(OpusRepacketizer *rp, unsigned char *data, opus_int32 maxlen) { return oa_rp_emit(rp, 0, rp->nb_frames, data, maxlen, false, false); }
int opus_packet_pad(unsigned char *data, opus_int32 len, opus_int32 new_len)
{
   if (len < 1) return OPUS_BAD_ARG;
   if (len == new_len) return OPUS_OK;
   if (len > new_len) return OPUS_BAD_ARG;
   std::vector<unsigned char> src(data, data + len);                                              /* the header grows: frames are taken from a copy */
   OpusRepacketizer rp;
   opus_repacketizer_init(&rp);
   opus_int32 r = oa_rp_add(&rp, src.data(), len, false);
   if (r != OPUS_OK) return r;
   r = oa_rp_emit(&rp, 0, rp.nb_frames, data, new_len, false, true);
   return r > 0 ? OPUS_OK : r;
}
opus_int32 opus_packet_unpad(unsigned char *data, opus_int32 len)
{
   if (len < 1) return OPUS_BAD_ARG;
   OpusRepacketizer rp;
   opus_repacketizer_init(&rp);
   const opus_int32 r = oa_rp_add(&rp, data, len, false);
   if (r < 0) return r;
   return oa_rp_emit(&rp, 0, rp.nb_frames, data, len, false, false);                              /* the header can only shrink: in place */
}
/* a multistream packet = nb_streams - 1 self-delimited packets followed by a plain one (RFC 6716 Appendix B); padding goes to the last */
int opus_multistream_packet_pad(unsigned char *data, opus_int32 len, opus_int32 new_len, int nb_streams)
{
   if (len < 1) return OPUS_BAD_ARG;
   if (len == new_len) return OPUS_OK;
   if (len > new_len) return OPUS_BAD_ARG;
   const opus_int32 grow = new_len - len;
   for (int s = 0; s < nb_streams - 1; s++) {
      OaFrameMap m;
      if (len <= 0) return OPUS_INVALID_PACKET;
      const int n = oa_frame_map(data, len, true, &m);
      if (n < 0) return n;
      data += m.end; len -= m.end;
   }
   return opus_packet_pad(data, len, len + grow);
}
opus_int32 opus_multistream_packet_unpad(unsigned char *data, opus_int32 len, int nb_streams)
{
   if (len < 1) return OPUS_BAD_ARG;
   unsigned char *dst = data;
   opus_int32 written = 0;
   for (int s = 0; s < nb_streams; s++) {
      const bool framed = s != nb_streams - 1;
      OaFrameMap m;
      if (len <= 0) return OPUS_INVALID_PACKET;
      const int n = oa_frame_map(data, len, framed, &m);
      if (n < 0) return n;
      OpusRepacketizer rp;
      opus_repacketizer_init(&rp);
      opus_int32 r = oa_rp_add(&rp, data, m.end, framed);
      if (r < 0) return r;
      r = oa_rp_emit(&rp, 0, rp.nb_frames, dst, len, framed, false);
      if (r < 0) return r;
      dst += r; written += r;
      data += m.end; len -= m.end;
   }
   return written;
}
} /* extern "C" */
#endif

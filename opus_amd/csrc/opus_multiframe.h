/* opus_multiframe.h — calls longer than one coded frame (40/60 ms in the CELT and hybrid modes, 80/100/120 ms in every mode) are coded as 2..6 frames
 * and framed as one packet (src/opus_encoder.c:1698-1838: opus_repacketizer_cat per frame, then opus_repacketizer_out_range_impl with pad = hard CBR).
 * On the device the frames' payloads are staged one behind the other in the stream's own output slot, OA_MF_HEADROOM bytes in, and this routine turns
 * the staging area into the packet in place: lane 0 derives the header (RFC 6716 §3.2: code 1 / 2 for two frames, code 3 otherwise or when padding),
 * all lanes slide the payload down behind it (forward copy, destinations never pass their sources) and zero the padding. */
#ifndef OPUS_AMD_MULTIFRAME_H
#define OPUS_AMD_MULTIFRAME_H

#ifndef LANE0
#define LANE0 for (int l0_ = (wv_sync(), wv_prio_serial(), 1); l0_; l0_ = (wv_prio_normal(), wv_sync(), 0)) if (wv_lane() == 0)
#define FOR_LANES(i, n) for (int i = wv_lane(); i < (n); i += WV_WIDTH)
#endif
#define OA_MF_MAX_FRAMES 6
#define OA_MF_HEADROOM 48            /* >= the header of any packet up to 7.6 KB: 2 + 31 padding length bytes + 2 x 5 frame lengths */
#define OA_MF_HDR_CAP 1280           /* bytes of the LDS buffer the header is built in (the frame's packet buffer) */

struct MfLds { i32 len[OA_MF_MAX_FRAMES]; i32 n, toc, total, hdr_len, err; };

WV_DEV int mf_put_length(WV_LDS u8 *dst, int n) { if (n < 252) { dst[0] = (u8)n; return 1; } dst[0] = (u8)(252 + (n & 3)); dst[1] = (u8)((n - (int)dst[0]) >> 2); return 2; }

/* out[OA_MF_HEADROOM ...] holds M->n payloads of M->len[i] bytes back to back; result: the packet at out[0 .. return).  maxlen = repacketize_len, fill = pad to it.
 * hdr: OA_MF_HEADROOM bytes of LDS for the header (the frame's packet buffer: every payload has left it by now); out_cap: bytes of the slot at out. */
WV_DEV int oa_multiframe_assemble_wave(WV_LDS MfLds *M, WV_LDS u8 *hdr, u8 *out, int maxlen, int fill, int out_cap)
{
   LANE0 {
      const int n = M->n;
      int body = 0, same = 1, h = 0, total = 0, err = 0;
      for (int i = 0; i < n; i++) { body += M->len[i]; same &= M->len[i] == M->len[0]; }
      const u8 toc = (u8)(M->toc & 0xFC);
      int code3 = n > 2;
      if (!code3) {
         hdr[h++] = (u8)(toc | (same ? 1 : 2));
         if (!same) h += mf_put_length(hdr + h, M->len[0]);
         total = h + body;
         if (total > maxlen) err = 1;
         code3 = fill && total < maxlen;
      }
      if (code3 && !err) {
         h = 0;
         hdr[h++] = (u8)(toc | 3);
         hdr[h++] = (u8)(n | (same ? 0 : 0x80));
         total = 2 + body;
         if (!same) for (int i = 0; i < n - 1; i++) total += M->len[i] < 252 ? 1 : 2;
         if (total > maxlen) err = 1;
         else {
            const int pad = fill ? maxlen - total : 0;
            if (pad > 0) {
               const int full = (pad - 1) / 255;
               hdr[1] |= 0x40;
               if (2 + full + 1 + 2 * (n - 1) > OA_MF_HDR_CAP) err = 1;             /* (a budget beyond ~320 KB: more length bytes than the header buffer holds) */
               else { for (int i = 0; i < full; i++) hdr[h++] = 255; hdr[h++] = (u8)(pad - 255 * full - 1); }
               total = maxlen;
            }
            if (!same) for (int i = 0; i < n - 1; i++) h += mf_put_length(hdr + h, M->len[i]);
         }
      }
      if (total > out_cap) err = 1;                                                  /* the padded packet must fit the slot the host promised */
      M->hdr_len = h; M->total = total; M->err = err; M->len[0] = body;          /* len[0] now carries the payload size for the copy below */
   }
   if (wv_uni(M->err)) return -2;
   const int h = wv_uni(M->hdr_len), body = wv_uni(M->len[0]), total = wv_uni(M->total);
   /* slide the payload behind the header, one 64-byte trip at a time: down by OA_MF_HEADROOM - h bytes (first trip first), or -- a hard-CBR budget so large that its
    * padding length bytes outgrow the head-room (OPUS_BITRATE_MAX with a buffer of 10 KB and more: the reference pads to the caller's whole buffer, :1757) -- up (last trip first) */
   const int up = h > OA_MF_HEADROOM, trips = (body + WV_WIDTH - 1) / WV_WIDTH;
   for (int t = 0; t < trips; t++) {
      const int b0 = (up ? trips - 1 - t : t) * WV_WIDTH;
      const int i = b0 + wv_lane();
      u8 v = 0;
      if (i < body) v = out[OA_MF_HEADROOM + i];
      wv_sync();
      if (i < body) out[h + i] = v;
      wv_sync();
   }
   FOR_LANES(i, h) out[i] = hdr[i];
   FOR_LANES(i, total - h - body) out[h + body + i] = 0;
   wv_sync();
   return total;
}
#endif

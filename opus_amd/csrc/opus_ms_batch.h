/* opus_ms_batch.h — a device-resident batch of B identical multistream encoders (BASELINE config 5: 257 encoders x 255 mono AUDIO streams).
 *
 * opus_multistream_encode (src/opus_multistream_encoder.c:841) is, per frame: rate allocation -> channel extraction -> one independent opus_encode per
 * elementary stream -> the streams' packets concatenated, all but the last in the self-delimiting framing of RFC 6716 Appendix B (:1016-1060).  The classic
 * entry points of opus_ms_host.h do that per encoder through host memory.  Here the B x N elementary streams ARE two ordinary encoder batches (the coupled
 * streams, the mono streams) whose state never leaves HBM, and a frame-step of all B encoders is four launches on one HIP stream, no host round trip:
 *   oa_ms_split_kernel   [B][frame][channels] interleaved PCM -> the two groups' per-stream inputs (one wave per (encoder, stream), coalesced)
 *   the coupled group's and the mono group's encode kernels (persistent waves over B * nc and B * nm streams)
 *   oa_ms_pack_kernel    one wave per encoder: every lane reads one stream's header and derives the Appendix-B length field, a wave scan turns the sizes into
 *                        offsets, then the wave copies packet after packet; the encoder's final range is the XOR over its streams.
 * The byte budget the reference hands to stream s depends on what streams 0..s-1 used (:1016-1026); when the caller's buffer is large enough for every stream
 * to be offered the elementary encoder's own cap, the budgets are all the same and the streams are independent: two encode launches.  A tighter buffer chains them:
 * the call then steps through the streams in order -- oa_ms_budget_kernel (one lane per encoder) turns what the streams before took into stream k's budget, an
 * encode launch codes stream k of all B encoders with it (oa_encode_launch: a strided subset of the batch with a per-record budget) -- 2 x streams launches, still
 * without a host round trip, packets and error codes those of opus_multistream_encode.  Hard CBR always chains: the packet is the bitrate's size (:918-927), the budget
 * kernel sets the last stream's bitrate from the bytes left (:1027), and the assembly goes through the general packet form (oa_ms_pack_cbr_kernel: padded elementary packets
 * lose their padding, the last one is padded out, :1032-1048).  Mapping families: 0 / 255 (plain), 2 (ambisonics:
 * elementary encoders forced to CELT), 3 (projection: the mixing matrix applied on the device, opus_ms_dec_batch.h) and 1 (surround: the masking analysis of every
 * encoder and the energy masks of its streams are two more launches per frame, oa_surround_kernel + oa_ms_surround_mask_kernel). */
#ifndef OPUS_AMD_MS_BATCH_H
#define OPUS_AMD_MS_BATCH_H

/* one wave per (encoder b, stream s): s < nc -> interleaved stereo pair into pc, else mono into pm */
extern "C" __global__ void __launch_bounds__(64)
oa_ms_split_kernel(const i16 *pcm, int frame, int nch, const i32 *chan /* [2 * nc | nm] source channel of every elementary channel */, int nc, int nm, i16 *pc, i16 *pm)
{
   const int ns = nc + nm, b = (int)blockIdx.x / ns, s = (int)blockIdx.x - b * ns;
   const i16 *src = pcm + (size_t)b * frame * nch;
   if (s < nc) {
      const int l = chan[2 * s], r = chan[2 * s + 1];
      i16 *d = pc + ((size_t)b * nc + s) * frame * 2;
      for (int i = threadIdx.x; i < frame; i += 64) { d[2 * i] = src[(size_t)i * nch + l]; d[2 * i + 1] = src[(size_t)i * nch + r]; }
   } else {
      const int c = chan[2 * nc + (s - nc)];
      i16 *d = pm + ((size_t)b * nm + (s - nc)) * frame;
      for (int i = threadIdx.x; i < frame; i += 64) d[i] = src[(size_t)i * nch + c];
   }
}

/* the same split of the caller's UN-mixed channels into the analysis' signal domain (int32, INT16TOSIG = << 12): what the elementary encoders' tonality analysis looks at when
 * a mixing matrix sits in front of them (opus_multistream_encoder.c:1027 hands opus_encode_native the original pcm with the stream's channel indices) */
extern "C" __global__ void __launch_bounds__(64)
oa_ms_split_sig_kernel(const i16 *pcm, int frame, int nch, const i32 *chan, int nc, int nm, i32 *pc, i32 *pm)
{
   const int ns = nc + nm, b = (int)blockIdx.x / ns, s = (int)blockIdx.x - b * ns;
   const i16 *src = pcm + (size_t)b * frame * nch;
   if (s < nc) {
      const int l = chan[2 * s], r = chan[2 * s + 1];
      i32 *d = pc + ((size_t)b * nc + s) * frame * 2;
      for (int i = threadIdx.x; i < frame; i += 64) { d[2 * i] = shl32((i32)src[(size_t)i * nch + l], 12); d[2 * i + 1] = shl32((i32)src[(size_t)i * nch + r], 12); }
   } else {
      const int c = chan[2 * nc + (s - nc)];
      i32 *d = pm + ((size_t)b * nm + (s - nc)) * frame;
      for (int i = threadIdx.x; i < frame; i += 64) d[i] = shl32((i32)src[(size_t)i * nch + c], 12);
   }
}

/* surround (mapping family 1, > 2 channels): what couples the channels (opus_surround_host.h: oa_surround_couple, src/opus_multistream_encoder.c:310-376) and hands every
 * elementary encoder its energy mask (:1005-1014), for all B encoders on the device.  One wave per encoder, lane = band: the per-channel log energies of oa_surround_kernel in
 * E [B][channels][21] become signal-to-mask ratios in place and go straight into the stream records of the two groups. */
WV_TABLE unsigned char oa_surround_pos_dev[9][8] = {{0}, {0}, {0}, {1, 2, 3}, {1, 3, 1, 3}, {1, 2, 3, 1, 3}, {1, 2, 3, 1, 3, 0}, {1, 2, 3, 1, 3, 2, 0}, {1, 2, 3, 1, 3, 1, 3, 0}};
extern "C" __global__ void __launch_bounds__(64)
oa_ms_surround_mask_kernel(i32 *E, int channels, const i32 *chan, int nc, int nm, OaShStream *shc, OaShStream *shm, OaStream *cc, OaStream *cm)
{
   const int b = blockIdx.x, i = threadIdx.x;
   i32 *e = E + (size_t)b * channels * 21;
   if (i < 21) {
      i32 m0 = -(28 << 24), m2 = -(28 << 24);
      for (int c = 0; c < channels; c++) {
         const int p = oa_surround_pos_dev[channels][c]; const i32 v = e[21 * c + i];
         if (p == 1) m0 = oa_logsum(m0, v);
         else if (p == 3) m2 = oa_logsum(m2, v);
         else if (p == 2) { m0 = oa_logsum(m0, v - (1 << 23)); m2 = oa_logsum(m2, v - (1 << 23)); }
      }
      const i32 m1 = m0 < m2 ? m0 : m2, off = oa_log2_q10(32768 / (channels - 1)) >> 1;
      for (int c = 0; c < channels; c++) {
         const int p = oa_surround_pos_dev[channels][c];
         e[21 * c + i] = p != 0 ? e[21 * c + i] - ((p == 1 ? m0 : p == 2 ? m1 : m2) + off) : 0;
      }
      for (int s = 0; s < nc; s++) {
         const int l = chan[2 * s], r = chan[2 * s + 1]; const size_t k = (size_t)b * nc + s;
         i32 *mask = shc ? shc[k].energy_mask : cc[k].energy_mask;
         mask[i] = e[21 * l + i]; mask[21 + i] = e[21 * r + i];
      }
      for (int s = 0; s < nm; s++) {
         const int c = chan[2 * nc + s]; const size_t k = (size_t)b * nm + s;
         i32 *mask = shm ? shm[k].energy_mask : cm[k].energy_mask;
         mask[i] = e[21 * c + i];
      }
   }
   if (i == 0) {
      for (int s = 0; s < nc; s++) { const size_t k = (size_t)b * nc + s; if (shc) { shc[k].cfg.energy_mask_on = 1; shc[k].s.celt_mask_cleared = 0; } else cc[k].energy_mask_on = 1; }
      for (int s = 0; s < nm; s++) { const size_t k = (size_t)b * nm + s; if (shm) { shm[k].cfg.energy_mask_on = 1; shm[k].s.celt_mask_cleared = 0; } else cm[k].energy_mask_on = 1; }
   }
}

/* bytes of an Opus packet's own header (RFC 6716 section 3.2: TOC, code-3 count byte, explicit lengths; packets of this encoder carry no padding in VBR) and the
 * size of its last frame -- what Appendix B wants coded in addition */
WV_DEV void oa_ms_header(const u8 *p, int len, int *hdr, int *last)
{
   const int code = p[0] & 3;
   if (code == 0) { *hdr = 1; *last = len - 1; }
   else if (code == 1) { *hdr = 1; *last = (len - 1) >> 1; }
   else if (code == 2) { const int a = p[1], lb = a < 252 ? 1 : 2, n0 = a < 252 ? a : 4 * p[2] + a; *hdr = 1 + lb; *last = len - 1 - lb - n0; }
   else {
      const int n = p[1] & 0x3F, vbr = p[1] >> 7;
      int at = 2, body = 0;
      if (vbr) for (int i = 0; i < n - 1; i++) { const int a = p[at], two = a >= 252; body += two ? 4 * p[at + 1] + a : a; at += 1 + two; }
      *hdr = at; *last = vbr ? len - at - body : (len - at) / n;
   }
}

/* The general form, for hard CBR: there an elementary packet may come padded (opus_encode_native pads to the CBR size, src/opus_encoder.c:2533-2544) and the multistream layer's
 * repacketizer (cat + out_range_impl, opus_multistream_encoder.c:1032-1048) drops that padding, writes the smallest header that carries the frames, and pads the LAST stream's
 * packet out to the bytes left.  oa_ms_plan = the frames of an elementary packet (the encoder emits at most six), oa_ms_shape = the header the repacketizer gives them
 * (opus_packet_host.h: oa_frame_map / oa_frames_emit are the host forms of the same rules). */
struct OaMsPlan { int toc, n, same, body, off0, size[6]; };
WV_DEV int oa_ms_plan(const u8 *p, int len, OaMsPlan *m)
{
   if (len < 1) return OPUS_INTERNAL_ERROR;
   const int code = p[0] & 3;
   int n = 1, vbr = 0, at = 1, pad = 0, sz[6] = {0, 0, 0, 0, 0, 0};
   if (code == 1) n = 2;
   else if (code == 2) {
      n = 2; vbr = 1;
      if (len < 2) return OPUS_INTERNAL_ERROR;
      const int a = p[1]; at = 2; sz[0] = a;
      if (a >= 252) { if (len < 3) return OPUS_INTERNAL_ERROR; sz[0] = 4 * p[2] + a; at = 3; }
   } else if (code == 3) {
      if (len < 2) return OPUS_INTERNAL_ERROR;
      const int v = p[1]; at = 2; n = v & 0x3F; vbr = v >> 7;
      if (n < 1 || n > 6) return OPUS_INTERNAL_ERROR;
      if (v & 0x40) { int b; do { if (at >= len) return OPUS_INTERNAL_ERROR; b = p[at++]; pad += b == 255 ? 254 : b; } while (b == 255); }
      if (vbr) {
#pragma unroll
         for (int i = 0; i < 5; i++) if (i < n - 1) {
            if (at >= len) return OPUS_INTERNAL_ERROR;
            const int a = p[at++]; int s = a;
            if (a >= 252) { if (at >= len) return OPUS_INTERNAL_ERROR; s = 4 * p[at++] + a; }
            sz[i] = s;
         }
      }
   }
   int rest = len - at - pad;
   if (rest < 0) return OPUS_INTERNAL_ERROR;
   if (vbr) {
#pragma unroll
      for (int i = 0; i < 5; i++) if (i < n - 1) rest -= sz[i];
      if (rest < 0) return OPUS_INTERNAL_ERROR;
#pragma unroll
      for (int i = 0; i < 6; i++) if (i == n - 1) sz[i] = rest;
   } else {
      if (rest % n) return OPUS_INTERNAL_ERROR;
      const int each = rest / n;
#pragma unroll
      for (int i = 0; i < 6; i++) if (i < n) sz[i] = each;
   }
   int body = 0, same = 1;
#pragma unroll
   for (int i = 0; i < 6; i++) { m->size[i] = sz[i]; if (i < n) { body += sz[i]; same = same && sz[i] == sz[0]; } }
   m->toc = p[0]; m->n = n; m->same = same; m->body = body; m->off0 = at;
   return n;
}
struct OaMsShape { int code3, h, total, pad, full; };      /* h: header bytes incl. padding length bytes and the Appendix-B length field; total: the packet */
WV_DEV int oa_ms_lenbytes(int v) { return v < 252 ? 1 : 2; }
WV_DEV int oa_ms_last_size(const OaMsPlan *m) { int v = 0; for (int i = 0; i < 6; i++) if (i == m->n - 1) v = m->size[i]; return v; }
WV_DEV int oa_ms_shape(const OaMsPlan *m, int maxlen, int framed, int fill, OaMsShape *o)
{
   const int n = m->n, tail = framed ? oa_ms_lenbytes(oa_ms_last_size(m)) : 0;
   int h = 0, total = 0, pad = 0, full = 0, code3 = n > 2;
   if (!code3) {
      h = 1 + (n == 2 && !m->same ? oa_ms_lenbytes(m->size[0]) : 0);
      total = h + tail + m->body;
      if (total > maxlen) return OPUS_BUFFER_TOO_SMALL;
      code3 = fill && total < maxlen;
   }
   if (code3) {
      h = 2;
      if (!m->same) for (int i = 0; i < 5; i++) if (i < n - 1) h += oa_ms_lenbytes(m->size[i]);
      total = h + tail + m->body;
      if (total > maxlen) return OPUS_BUFFER_TOO_SMALL;
      pad = fill ? maxlen - total : 0;
      if (pad > 0) { full = (pad - 1) / 255; h += full + 1; total = maxlen; }
   }
   o->code3 = code3; o->h = h + tail; o->total = total; o->pad = pad; o->full = full;
   return OPUS_OK;
}
WV_DEV int oa_ms_put_length(int v, u8 *d) { if (v < 252) { d[0] = (u8)v; return 1; } d[0] = (u8)(252 + (v & 3)); d[1] = (u8)((v - d[0]) >> 2); return 2; }
WV_DEV void oa_ms_put_header(const OaMsPlan *m, const OaMsShape *o, int framed, u8 *q)
{
   const int n = m->n, toc = m->toc & 0xFC;
   if (!o->code3) {
      *q++ = (u8)(toc | (n == 1 ? 0 : m->same ? 1 : 2));
      if (n == 2 && !m->same) q += oa_ms_put_length(m->size[0], q);
   } else {
      *q++ = (u8)(toc | 3);
      *q++ = (u8)(n | (m->same ? 0 : 0x80) | (o->pad > 0 ? 0x40 : 0));
      if (o->pad > 0) { for (int i = 0; i < o->full; i++) *q++ = 255; *q++ = (u8)(o->pad - 255 * o->full - 1); }
      if (!m->same) for (int i = 0; i < 5; i++) if (i < n - 1) q += oa_ms_put_length(m->size[i], q);
   }
   if (framed) q += oa_ms_put_length(oa_ms_last_size(m), q);
}

/* The chained byte budgets of opus_multistream_encode_native (src/opus_multistream_encoder.c:1016-1027) for all B encoders, one LANE per encoder: before stream s is coded,
 * what stream s - 1 took (its packet plus the Appendix-B length field it will carry) joins the encoder's running total, and stream s's byte budget follows from what is
 * left -- two bytes kept back for each stream still to come (one for the last), one more each at 100 ms, the elementary encoder's own cap, the length field of this stream.
 * An encoder whose stream failed, or whose budget ran out, stops there like the reference's loop does: its later streams get budget 0 and sit the remaining calls out. */
extern "C" __global__ void __launch_bounds__(64)
oa_ms_budget_kernel(int s, int ns, int nc, int B, int Fs, int frame_size, int max_data_bytes, const u8 *pkc, const i32 *lc, const u8 *pkm, const i32 *lm, int stride,
      i32 *tot, i32 *err, i32 *budget_c, i32 *budget_m,
      int cbr /* hard CBR: the general packet form above, and the last stream's bitrate follows from the bytes left (:1027) */, char *last_rate /* &record[0].cfg.user_bitrate_bps of the
      last stream's batch */, long long rate_pitch /* bytes between the last streams of consecutive encoders */, long long rate_first /* bytes to encoder 0's */, int last_channels)
{
   const int b = (int)blockIdx.x * 64 + (int)threadIdx.x, nm = ns - nc;
   if (b >= B) return;
   i32 t = s == 0 ? 0 : tot[b], e = s == 0 ? 0 : err[b];
   if (s > 0 && !e) {
      const int q = s - 1;
      const u8 *p = q < nc ? pkc + ((size_t)b * nc + q) * stride : pkm + ((size_t)b * nm + (q - nc)) * stride;
      const int len = q < nc ? lc[b * nc + q] : lm[b * nm + (q - nc)];
      if (len <= 0) e = len < 0 ? len : OPUS_INTERNAL_ERROR;
      else if (!cbr) { int hdr, last; oa_ms_header(p, len, &hdr, &last); t += len + (last < 252 ? 1 : 2); }
      else { OaMsPlan pl; OaMsShape sh; int r = oa_ms_plan(p, len, &pl); if (r > 0) r = oa_ms_shape(&pl, max_data_bytes - t, 1, 0, &sh); if (r < 0) e = r; else t += sh.total; }
   }
   i32 bud = 0;
   if (!e) {
      i32 curr_max = max_data_bytes - t;
      const int resv = 2 * (ns - s - 1) - 1;
      curr_max -= resv > 0 ? resv : 0;
      if (Fs / frame_size == 10) curr_max -= ns - s - 1;
      if (curr_max > OA_MS_FRAME_TMP) curr_max = OA_MS_FRAME_TMP;
      if (s != ns - 1) curr_max -= curr_max > 253 ? 2 : 1;
      /* INVARIANT (hard CBR): the override below exists in the DEVICE record only -- the host mirror of the last stream (h_sh / h_streams) keeps the rate the surround
       * allocation gave it.  That is sound because (1) every hard-CBR frame writes the value again before the last stream's kernels read it, (2) any ctl on the batch
       * uploads the mirror over the record, after which (1) applies again, and (3) the reference leaves the same trace: its OPUS_SET_BITRATE on the last encoder is
       * overwritten by the next frame's rate allocation (opus_multistream_encoder.c:1005-1030).  Between frames the two copies therefore differ: opusgpu_enc_batch_get answers configuration
       * requests from the mirror (the allocated rate), export_state / copy_states move the device record (the overridden one, what OPUS_GET_BITRATE on the reference's last
       * encoder reports between frames); no frame reads either before (1) has written it again. */
      if (cbr && s == ns - 1) {                                           /* OPUS_SET_BITRATE(bits_to_bitrate(curr_max * 8, Fs, frame_size)) on the last elementary encoder (opus_encoder.c ctl: <= 0 is refused, then 500 .. 750000 per channel) */
         i32 v = curr_max * 8 * (6 * Fs / frame_size) / 6;
         if (v > 0) { v = v <= 500 ? 500 : v > 750000 * last_channels ? 750000 * last_channels : v; *(i32 *)(last_rate + rate_first + (long long)b * rate_pitch) = v; }
      }
      if (curr_max <= 0) e = OPUS_BUFFER_TOO_SMALL; else bud = curr_max;
   }
   tot[b] = t; err[b] = e;
   if (s < nc) budget_c[b * nc + s] = bud; else budget_m[b * nm + (s - nc)] = bud;
}

extern "C" __global__ void __launch_bounds__(64)
oa_ms_pack_kernel(const u8 *pkc, const i32 *lc, const u32 *rc, int nc, const u8 *pkm, const i32 *lm, const u32 *rm, int nm, int stride,
      u8 *out, int out_stride, int max_data_bytes, i32 *lens, u32 *rngs, const i32 *err_in /* NULL, or per encoder: the error its chained-budget loop stopped on */)
{
   const int b = blockIdx.x, ns = nc + nm, lane = threadIdx.x;
   u8 *dst = out + (size_t)b * out_stride;
   if (err_in && err_in[b]) { if (lane == 0) { lens[b] = err_in[b]; rngs[b] = 0; } return; }
   int at = 0, err = 0;
   u32 rx = 0;
   for (int s0 = 0; s0 < ns; s0 += 64) {
      const int s = s0 + lane, mine = s < ns;
      const u8 *p = 0; int len = 0, hdr = 0, last = 0, extra = 0;
      if (mine) {
         if (s < nc) { p = pkc + ((size_t)b * nc + s) * stride; len = lc[b * nc + s]; rx ^= rc[b * nc + s]; }
         else { p = pkm + ((size_t)b * nm + (s - nc)) * stride; len = lm[b * nm + (s - nc)]; rx ^= rm[b * nm + (s - nc)]; }
         if (len <= 0) err = len < 0 ? len : OPUS_INTERNAL_ERROR;
         else { oa_ms_header(p, len, &hdr, &last); extra = s == ns - 1 ? 0 : (last < 252 ? 1 : 2); }
      }
      err = wv_min(err);
      if (err) break;
      const int total = mine ? len + extra : 0;
      const int incl = wv_scan_incl(total), off = at + incl - total;
      const int chunk = imin(64, ns - s0);
      for (int k = 0; k < chunk; k++) {                                   /* packet after packet, all lanes on the bytes */
         const unsigned long long pw = (unsigned long long)p;
         const u8 *q = (const u8 *)(((unsigned long long)(u32)wv_bcast((i32)(pw >> 32), k) << 32) | (unsigned long long)(u32)wv_bcast((i32)(u32)pw, k));
         const int ql = wv_bcast(len, k), qh = wv_bcast(hdr, k), qx = wv_bcast(extra, k), qlast = wv_bcast(last, k), qo = wv_bcast(off, k);
         if (qo + ql + qx > max_data_bytes || qo + ql + qx > out_stride) { err = OPUS_BUFFER_TOO_SMALL; break; }
         for (int i = lane; i < qh; i += 64) dst[qo + i] = q[i];
         if (lane == 0 && qx) { if (qlast < 252) dst[qo + qh] = (u8)qlast; else { const int f = 252 + (qlast & 3); dst[qo + qh] = (u8)f; dst[qo + qh + 1] = (u8)((qlast - f) >> 2); } }
         for (int i = lane; i < ql - qh; i += 64) dst[qo + qh + qx + i] = q[qh + i];
      }
      if (err) break;
      at += wv_bcast(incl, 63);
   }
   for (int d = 1; d < 64; d <<= 1) rx ^= (u32)wv_shfl((i32)rx, lane ^ d);
   if (lane == 0) { lens[b] = err ? err : at; rngs[b] = err ? 0 : rx; }
}

/* the same assembly for hard CBR, through the general packet form: every lane plans its stream's packet (frames, the repacketizer's header), a wave scan places them, the last
 * stream's packet is padded out to max_data_bytes; the owner lane writes its header, the wave copies the frames (contiguous in an elementary packet) and zeroes the padding */
extern "C" __global__ void __launch_bounds__(64)
oa_ms_pack_cbr_kernel(const u8 *pkc, const i32 *lc, const u32 *rc, int nc, const u8 *pkm, const i32 *lm, const u32 *rm, int nm, int stride,
      u8 *out, int out_stride, int max_data_bytes, i32 *lens, u32 *rngs, const i32 *err_in)
{
   const int b = blockIdx.x, ns = nc + nm, lane = threadIdx.x;
   u8 *dst = out + (size_t)b * out_stride;
   if (err_in && err_in[b]) { if (lane == 0) { lens[b] = err_in[b]; rngs[b] = 0; } return; }
   int at = 0, err = 0;
   u32 rx = 0;
   for (int s0 = 0; s0 < ns; s0 += 64) {
      const int s = s0 + lane, mine = s < ns, is_last = s == ns - 1;
      const u8 *p = 0; int len = 0, e = 0;
      OaMsPlan pl; OaMsShape sh; pl.n = 1; pl.body = 0; pl.off0 = 0; sh.total = 0; sh.h = 0;
      if (mine) {
         if (s < nc) { p = pkc + ((size_t)b * nc + s) * stride; len = lc[b * nc + s]; rx ^= rc[b * nc + s]; }
         else { p = pkm + ((size_t)b * nm + (s - nc)) * stride; len = lm[b * nm + (s - nc)]; rx ^= rm[b * nm + (s - nc)]; }
         if (len <= 0) e = len < 0 ? len : OPUS_INTERNAL_ERROR;
         else { const int r = oa_ms_plan(p, len, &pl); if (r < 0) e = r; else if (!is_last) e = oa_ms_shape(&pl, 1 << 30, 1, 0, &sh); }
      }
      err = wv_min(e);
      if (err) break;
      const int total0 = mine && !is_last ? sh.total : 0;
      const int incl = wv_scan_incl(total0);
      int off = at + incl - total0;
      if (mine && is_last) { e = oa_ms_shape(&pl, max_data_bytes - off, 0, 1, &sh); }
      err = wv_min(e);
      if (err) break;
      const int total = mine ? sh.total : 0;
      if (mine && (off + total > max_data_bytes || off + total > out_stride)) e = OPUS_BUFFER_TOO_SMALL;
      err = wv_min(e);
      if (err) break;
      if (mine) oa_ms_put_header(&pl, &sh, !is_last, dst + off);
      const int chunk = imin(64, ns - s0);
      for (int k = 0; k < chunk; k++) {                                   /* packet after packet, all lanes on the bytes */
         const unsigned long long pw = (unsigned long long)p;
         const u8 *q = (const u8 *)(((unsigned long long)(u32)wv_bcast((i32)(pw >> 32), k) << 32) | (unsigned long long)(u32)wv_bcast((i32)(u32)pw, k));
         const int qo = wv_bcast(off, k) + wv_bcast(sh.h, k), qf = wv_bcast(pl.off0, k), qb = wv_bcast(pl.body, k), qt = wv_bcast(off, k) + wv_bcast(total, k);
         for (int i = lane; i < qb; i += 64) dst[qo + i] = q[qf + i];
         for (int i = qo + qb + lane; i < qt; i += 64) dst[i] = 0;
      }
      at += wv_bcast(wv_scan_incl(total), 63);
   }
   for (int d = 1; d < 64; d <<= 1) rx ^= (u32)wv_shfl((i32)rx, lane ^ d);
   if (lane == 0) { lens[b] = err ? err : at; rngs[b] = err ? 0 : rx; }
}

struct OpusGpuMsEncBatch {
   int B, nch, ns, nc, nm, device, application, last_frame_size;
   opus_int32 Fs;
   OpusMSEncoder *proto;                 /* host-side prototype encoder: layout, controls, rate allocation (never encodes) */
   OpusGpuEncBatch *bc, *bm;             /* the B * nc coupled and B * nm mono elementary encoders */
   i32 *d_chan;
   i16 *d_pc, *d_pm; size_t pc_cap, pm_cap;
   i32 *d_surround;                                                       /* surround: [B][channels][120] window memory | [B][channels] pre-emphasis memory | [B][channels][21] energies / ratios */
   i16 *d_M, *d_mixed; size_t mixed_cap; i32 *d_apc, *d_apm; size_t apc_cap, apm_cap;      /* mapping family 3 (projection): the mixing matrix [C][C], the mixed input, the un-mixed channels for the analysis */
   u8 *d_pkc, *d_pkm; i32 *d_lc, *d_lm; u32 *d_rc, *d_rm; opus_int32 stride;
   i32 *d_budget_c, *d_budget_m, *d_tot, *d_err;                          /* chained byte budgets (tight buffers): per elementary stream, per encoder */
   /* staging of the host-pointer entry */
   i16 *d_pcm; size_t pcm_cap; u8 *d_out; size_t out_cap; i32 *d_lens; u32 *d_rng;
};

extern "C" {
void opusgpu_ms_enc_batch_destroy(OpusGpuMsEncBatch *m)
{
   if (!m) return;
   if (m->bc) opusgpu_enc_batch_destroy(m->bc);
   if (m->bm) opusgpu_enc_batch_destroy(m->bm);
   (void)hipSetDevice(m->device);
   void *bufs[] = {m->d_surround, m->d_M, m->d_mixed, m->d_apc, m->d_apm, m->d_chan, m->d_pc, m->d_pm, m->d_pkc, m->d_pkm, m->d_lc, m->d_lm, m->d_rc, m->d_rm, m->d_pcm, m->d_out, m->d_lens, m->d_rng, m->d_budget_c, m->d_budget_m, m->d_tot, m->d_err};
   for (void *p : bufs) if (p) (void)hipFree(p);
   free(m->proto);
   delete m;
}
/* B encoders of opus_multistream_encoder_create(Fs, channels, streams, coupled_streams, mapping, application) (include/opus_multistream.h:260); mapping_family 0 / 255:
 * plain layouts, 2: ambisonics (the caller passes the layout opus_multistream_surround_encoder_create would derive) */
OpusGpuMsEncBatch *opusgpu_ms_enc_batch_create(opus_int32 B, opus_int32 Fs, int channels, int mapping_family, int streams, int coupled_streams, const unsigned char *mapping,
      int application, int device, int *error)
{
   int err = OPUS_OK;
   OpusGpuMsEncBatch *m = nullptr;
   if (B <= 0 || !mapping || (mapping_family != 0 && mapping_family != 255 && mapping_family != 1 && mapping_family != 2 && mapping_family != 3)) err = OPUS_BAD_ARG;
   int o1 = 0, map_type = mapping_family == 2 ? OA_MAP_AMBISONICS : OA_MAP_NONE, lfe = -1;
   unsigned char ident[255];
   if (err == OPUS_OK && mapping_family == 1) {                             /* surround: the Vorbis layout opus_multistream_surround_encoder_create derives (the caller passes the same streams / coupled) */
      int ps = 0, pc = 0;
      if (oa_surround_layout(channels, 1, &ps, &pc, ident, &map_type, &lfe) != OPUS_OK || ps != streams || pc != coupled_streams) err = OPUS_BAD_ARG;
      mapping = ident;
   }
   if (err == OPUS_OK && mapping_family == 3) {                             /* projection: the layout opus_projection_ambisonics_encoder_init derives, identity mapping, mixing matrix of the order */
      int ps = 0, pc = 0;
      if (oa_proj_layout(channels, 3, &ps, &pc, &o1) != OPUS_OK || o1 < 2 || o1 > 6 || ps != streams || pc != coupled_streams || channels > OA_PROJ_MAXC) err = OPUS_BAD_ARG;
      for (int i = 0; i < channels && i < 255; i++) ident[i] = (unsigned char)i;
      mapping = ident;
   }
   OpusMSEncoder *proto = nullptr;
   if (err == OPUS_OK) {
      const opus_int32 sz = opus_multistream_encoder_get_size(streams, coupled_streams);
      proto = sz > 0 ? (OpusMSEncoder *)malloc((size_t)sz + (map_type == OA_MAP_SURROUND ? (size_t)channels * (120 + 1) * sizeof(opus_int32) : 0)) : nullptr;      /* (a surround encoder carries its analysis memory behind the records: opus_multistream_surround_encoder_get_size) */
      if (!proto) err = sz > 0 ? OPUS_ALLOC_FAIL : OPUS_BAD_ARG;
      else err = oa_ms_encoder_init_impl(proto, Fs, channels, streams, coupled_streams, mapping, application, map_type, lfe);      /* (family 3: opus_projection_ambisonics_encoder_init builds a PLAIN multistream encoder behind its matrix, opus_projection_encoder.c:224) */
   }
   if (err == OPUS_OK) {
      m = new OpusGpuMsEncBatch();
      memset(m, 0, sizeof(*m));
      m->B = B; m->nch = channels; m->ns = streams; m->nc = coupled_streams; m->nm = streams - coupled_streams; m->device = device; m->application = application; m->Fs = Fs; m->proto = proto;
      proto = nullptr;
      if (m->nc) m->bc = opusgpu_enc_batch_create(B * m->nc, Fs, 2, application, device, &err);
      if (err == OPUS_OK && m->nm) m->bm = opusgpu_enc_batch_create(B * m->nm, Fs, 1, application, device, &err);
      std::vector<i32> chan((size_t)2 * m->nc + m->nm + 1);
      for (int s = 0; s < m->nc; s++) { chan[2 * s] = oa_get_left(&m->proto->layout, s, -1); chan[2 * s + 1] = oa_get_right(&m->proto->layout, s, -1); }
      for (int s = 0; s < m->nm; s++) chan[2 * m->nc + s] = oa_get_mono(&m->proto->layout, m->nc + s, -1);
      const size_t nstr = (size_t)B * streams;
      m->stride = (oa_enc_out_stride_needed(Fs, Fs / 25 * 3, 1276 * 6) + 15) & ~15;
      if (err == OPUS_OK) {
         const bool ok = hipSetDevice(device) == hipSuccess && hipMalloc((void **)&m->d_chan, chan.size() * 4) == hipSuccess &&
               hipMemcpy(m->d_chan, chan.data(), chan.size() * 4, hipMemcpyHostToDevice) == hipSuccess &&
               hipMalloc((void **)&m->d_pkc, (size_t)B * (m->nc ? m->nc : 1) * m->stride) == hipSuccess && hipMalloc((void **)&m->d_pkm, (size_t)B * (m->nm ? m->nm : 1) * m->stride) == hipSuccess &&
               hipMalloc((void **)&m->d_lc, nstr * 4 + 4) == hipSuccess && hipMalloc((void **)&m->d_lm, nstr * 4 + 4) == hipSuccess &&
               hipMalloc((void **)&m->d_rc, nstr * 4 + 4) == hipSuccess && hipMalloc((void **)&m->d_rm, nstr * 4 + 4) == hipSuccess &&
               hipMalloc((void **)&m->d_lens, (size_t)B * 4) == hipSuccess && hipMalloc((void **)&m->d_rng, (size_t)B * 4) == hipSuccess;
         if (!ok) err = OPUS_ALLOC_FAIL;
      }
      if (err == OPUS_OK && lfe >= 0 && m->bm) for (int b = 0; b < B && err == OPUS_OK; b++) err = opusgpu_enc_batch_ctl(m->bm, b * m->nm + (lfe - m->nc), OPUS_SET_LFE_REQUEST, 1);      /* (:499) */
      if (err == OPUS_OK && map_type == OA_MAP_SURROUND) {
         const size_t nw = (size_t)B * channels * (120 + 1 + 21);
         if (hipMalloc((void **)&m->d_surround, nw * 4) != hipSuccess || hipMemset(m->d_surround, 0, nw * 4) != hipSuccess) err = OPUS_ALLOC_FAIL;
      }
      if (err == OPUS_OK && mapping_family == 3) {
         const OaMatrixDesc *mix = &oa_pm_mixing[o1 - 2];
         std::vector<i16> M((size_t)channels * channels);
         for (int c = 0; c < channels; c++) for (int r = 0; r < channels; r++) M[(size_t)channels * c + r] = mix->data[mix->rows * c + r];
         if (hipMalloc((void **)&m->d_M, M.size() * 2) != hipSuccess || hipMemcpy(m->d_M, M.data(), M.size() * 2, hipMemcpyHostToDevice) != hipSuccess) err = OPUS_ALLOC_FAIL;
      }
      if (err == OPUS_OK && mapping_family == 2) {                        /* ambisonics: CELT-only elementary encoders (opus_multistream_encoder.c:986) */
         if (m->bc) err = opusgpu_enc_batch_ctl(m->bc, -1, OPUS_SET_FORCE_MODE_REQUEST, OPUS_MODE_CELT_ONLY);
         if (err == OPUS_OK && m->bm) err = opusgpu_enc_batch_ctl(m->bm, -1, OPUS_SET_FORCE_MODE_REQUEST, OPUS_MODE_CELT_ONLY);
      }
      if (err != OPUS_OK) { opusgpu_ms_enc_batch_destroy(m); m = nullptr; }
   }
   free(proto);
   if (error) *error = err;
   return m;
}
/* opus_multistream_encoder_ctl for all B encoders: OPUS_SET_BITRATE feeds the rate allocation, every other SET goes to every elementary encoder (:1245-1278) */
int opusgpu_ms_enc_batch_ctl(OpusGpuMsEncBatch *m, int request, opus_int32 value)
{
   if (!m) return OPUS_BAD_ARG;
   if (request & 1) return OPUS_BAD_ARG;                                  /* GETs: ask one encoder through the classic API */
   const int r = opus_multistream_encoder_ctl(m->proto, request, value);
   if (r != OPUS_OK) return r;
   if (request == OPUS_SET_BITRATE_REQUEST) { m->last_frame_size = 0; return OPUS_OK; }
   int e = OPUS_OK;
   if (m->bc) e = opusgpu_enc_batch_ctl(m->bc, -1, request, value);
   if (e == OPUS_OK && m->bm) e = opusgpu_enc_batch_ctl(m->bm, -1, request, value);
   return e;
}
/* one frame of all B encoders, everything in HBM: d_pcm [B][frame_size][channels] int16 -> d_out [B][out_stride] packets, d_lens [B] (byte count or a negative
 * OPUS_* code per encoder), d_final_range [B].  max_data_bytes as in opus_multistream_encode; see the header for the buffer-size condition. */
int opusgpu_ms_encode_batch_dev(OpusGpuMsEncBatch *m, const opus_int16 *d_pcm, int frame_size, unsigned char *d_out, opus_int32 out_stride, opus_int32 max_data_bytes,
      opus_int32 *d_lens, opus_uint32 *d_final_range, void *hip_stream)
{
   if (!m || !d_pcm || !d_out || !d_lens || !d_final_range) return OPUS_BAD_ARG;
   const opus_int32 Fs = m->Fs;
   if (oa_frame_size_select(m->application, frame_size, OPUS_FRAMESIZE_ARG, Fs) != frame_size) return OPUS_BAD_ARG;
   const int nf = frame_size > Fs / 50 ? (frame_size * 50 + Fs - 1) / Fs : 1;
   const long long worst = (long long)(m->ns - 1) * (1276 * nf + 3) + OA_MS_FRAME_TMP + 3 * m->ns + 8;
   opus_int32 vbr = 1;
   (void)opus_multistream_encoder_ctl(m->proto, OPUS_GET_VBR_REQUEST, &vbr);
   opus_int32 smallest_packet = m->ns * 2 - 1;
   if (Fs / frame_size == 10) smallest_packet += m->ns;
   if (max_data_bytes < smallest_packet) return OPUS_BUFFER_TOO_SMALL;      /* (:898-906) */
   if (!vbr) {                                                            /* hard CBR: the packet is the bitrate's size (:918-927), the last stream takes -- and is padded to -- what the others leave (:1027, :1048) */
      std::vector<opus_int32> rates((size_t)m->ns);
      const opus_int32 rate_sum = oa_ms_rate_allocation(m->proto, rates.data(), frame_size);
      if (m->proto->bitrate_bps == OPUS_AUTO) { const opus_int32 c = (rate_sum * 6 / (6 * Fs / frame_size) + 4) / 8; if (c < max_data_bytes) max_data_bytes = c; }
      else if (m->proto->bitrate_bps != OPUS_BITRATE_MAX) {
         opus_int32 c = (m->proto->bitrate_bps * 6 / (6 * Fs / frame_size) + 4) / 8;
         if (c < smallest_packet) c = smallest_packet;
         if (c < max_data_bytes) max_data_bytes = c;
      }
   }
   const bool chained = !vbr || max_data_bytes < worst;                   /* some stream's budget may bind: the streams are stepped in order, each with the bytes its predecessors left (:1016-1027) */
   HIPCHECK(hipSetDevice(m->device));
   if (chained && !m->d_tot) {
      const size_t nstr = (size_t)m->B * m->ns;
      if (hipMalloc((void **)&m->d_budget_c, nstr * 4 + 4) != hipSuccess || hipMalloc((void **)&m->d_budget_m, nstr * 4 + 4) != hipSuccess ||
          hipMalloc((void **)&m->d_tot, (size_t)m->B * 4) != hipSuccess || hipMalloc((void **)&m->d_err, (size_t)m->B * 4) != hipSuccess) return OPUS_ALLOC_FAIL;
   }
   hipStream_t s = hip_stream ? (hipStream_t)hip_stream : (m->bc ? m->bc->stream : m->bm->stream);
   if (frame_size != m->last_frame_size) {                                /* per-stream rates depend on the frame rate (rate_allocation :702): refresh on change */
      std::vector<opus_int32> rates((size_t)m->ns);
      (void)oa_ms_rate_allocation(m->proto, rates.data(), frame_size);
      for (int b = 0; b < m->B; b++) for (int k = 0; k < m->ns; k++) {
         const int e = k < m->nc ? opusgpu_enc_batch_ctl(m->bc, b * m->nc + k, OPUS_SET_BITRATE_REQUEST, rates[k]) : opusgpu_enc_batch_ctl(m->bm, b * m->nm + (k - m->nc), OPUS_SET_BITRATE_REQUEST, rates[k]);
         if (e != OPUS_OK) return e;
      }
      if (m->proto->mapping_type == OA_MAP_SURROUND) {                     /* what opus_multistream_encode_native sets on every elementary encoder of a surround layout (:965-985) */
         opus_int32 equiv_rate = m->proto->bitrate_bps;
         if (frame_size * 50 < Fs) equiv_rate -= 60 * (Fs / frame_size - 50) * m->nch;
         const opus_int32 bw = equiv_rate > 10000 * m->nch ? OPUS_BANDWIDTH_FULLBAND : equiv_rate > 7000 * m->nch ? OPUS_BANDWIDTH_SUPERWIDEBAND : equiv_rate > 5000 * m->nch ? OPUS_BANDWIDTH_WIDEBAND : OPUS_BANDWIDTH_NARROWBAND;
         int e = OPUS_OK;
         if (m->bc) { e = opusgpu_enc_batch_ctl(m->bc, -1, OPUS_SET_BANDWIDTH_REQUEST, bw); if (e == OPUS_OK) e = opusgpu_enc_batch_ctl(m->bc, -1, OPUS_SET_FORCE_MODE_REQUEST, OPUS_MODE_CELT_ONLY); if (e == OPUS_OK) e = opusgpu_enc_batch_ctl(m->bc, -1, OPUS_SET_FORCE_CHANNELS_REQUEST, 2); }
         if (e == OPUS_OK && m->bm) e = opusgpu_enc_batch_ctl(m->bm, -1, OPUS_SET_BANDWIDTH_REQUEST, bw);
         if (e != OPUS_OK) return e;
      }
      m->last_frame_size = frame_size;
   }
   const size_t need_c = (size_t)m->B * m->nc * frame_size * 2 * sizeof(i16), need_m = (size_t)m->B * m->nm * frame_size * sizeof(i16);
   if (need_c > m->pc_cap) { HIPCHECK(hipStreamSynchronize(s)); if (m->d_pc) (void)hipFree(m->d_pc); m->d_pc = nullptr; m->pc_cap = 0; HIPCHECK(hipMalloc((void **)&m->d_pc, need_c)); m->pc_cap = need_c; }
   if (need_m > m->pm_cap) { HIPCHECK(hipStreamSynchronize(s)); if (m->d_pm) (void)hipFree(m->d_pm); m->d_pm = nullptr; m->pm_cap = 0; HIPCHECK(hipMalloc((void **)&m->d_pm, need_m)); m->pm_cap = need_m; }
   if (m->d_surround && m->application != OPUS_APPLICATION_RESTRICTED_SILK) {  /* surround_analysis (:230) of all B encoders, then the masks into the stream records */
      i32 *mem = m->d_surround, *pre = mem + (size_t)m->B * m->nch * 120, *E = pre + (size_t)m->B * m->nch;
      hipLaunchKernelGGL(oa_surround_kernel, dim3((unsigned)(m->B * m->nch)), dim3(64), 0, s, (const i16 *)d_pcm, frame_size, m->nch, (int)Fs, mem, pre, E);
      hipLaunchKernelGGL(oa_ms_surround_mask_kernel, dim3((unsigned)m->B), dim3(64), 0, s, E, m->nch, (const i32 *)m->d_chan, m->nc, m->nm,
            m->bc && m->bc->kind ? m->bc->d_sh : nullptr, m->bm && m->bm->kind ? m->bm->d_sh : nullptr, m->bc && !m->bc->kind ? m->bc->d_streams : nullptr, m->bm && !m->bm->kind ? m->bm->d_streams : nullptr);
      HIPCHECK(hipGetLastError());
   }
   const i16 *src = (const i16 *)d_pcm;
   if (m->d_M) {                                                           /* projection: mix on the device, and give the analyses the caller's un-mixed channels */
      const size_t need_x = (size_t)m->B * frame_size * m->nch * sizeof(i16);
      if (need_x > m->mixed_cap) { HIPCHECK(hipStreamSynchronize(s)); if (m->d_mixed) (void)hipFree(m->d_mixed); m->d_mixed = nullptr; m->mixed_cap = 0; HIPCHECK(hipMalloc((void **)&m->d_mixed, need_x)); m->mixed_cap = need_x; }
      if (2 * need_c > m->apc_cap) { HIPCHECK(hipStreamSynchronize(s)); if (m->d_apc) (void)hipFree(m->d_apc); m->d_apc = nullptr; m->apc_cap = 0; HIPCHECK(hipMalloc((void **)&m->d_apc, 2 * need_c + 4)); m->apc_cap = 2 * need_c; }
      if (2 * need_m > m->apm_cap) { HIPCHECK(hipStreamSynchronize(s)); if (m->d_apm) (void)hipFree(m->d_apm); m->d_apm = nullptr; m->apm_cap = 0; HIPCHECK(hipMalloc((void **)&m->d_apm, 2 * need_m + 4)); m->apm_cap = 2 * need_m; }
      hipLaunchKernelGGL(oa_proj_mix_kernel, dim3((unsigned)(((frame_size + 63) / 64) * m->B)), dim3(64), 0, s, (const i16 *)m->d_M, m->nch, (const i16 *)d_pcm, frame_size, m->d_mixed);
      hipLaunchKernelGGL(oa_ms_split_sig_kernel, dim3((unsigned)(m->B * m->ns)), dim3(64), 0, s, (const i16 *)d_pcm, frame_size, m->nch, (const i32 *)m->d_chan, m->nc, m->nm, m->d_apc, m->d_apm);
      src = m->d_mixed;
   }
   hipLaunchKernelGGL(oa_ms_split_kernel, dim3((unsigned)(m->B * m->ns)), dim3(64), 0, s, src, frame_size, m->nch, (const i32 *)m->d_chan, m->nc, m->nm, m->d_pc, m->d_pm);
   HIPCHECK(hipGetLastError());
   if (!chained) {
      if (m->nc) { const int r = opusgpu_encode_batch_dev_sig(m->bc, m->d_pc, m->d_M ? m->d_apc : nullptr, frame_size, m->d_pkc, m->stride, 1276 * 6, m->d_lc, m->d_rc, s); if (r != OPUS_OK) return r; }
      if (m->nm) { const int r = opusgpu_encode_batch_dev_sig(m->bm, m->d_pm, m->d_M ? m->d_apm : nullptr, frame_size, m->d_pkm, m->stride, 1276 * 6, m->d_lm, m->d_rm, s); if (r != OPUS_OK) return r; }
   } else {
      /* stream after stream, all B encoders at once: the budget kernel (one lane per encoder) turns what the streams before took into this stream's byte budget, then
       * the B records of stream k -- every nc-th of the coupled batch, or every nm-th of the mono batch -- are coded with it.  2 x streams launches instead of 2: what
       * a caller with a tight buffer pays for staying on the device */
      /* hard CBR: where the last stream's user bitrate lives on the device (the budget kernel sets it per encoder) */
      OpusGpuEncBatch *lb = m->nm ? m->bm : m->bc;
      const int lper = m->nm ? m->nm : m->nc;
      char *last_rate = lb->kind ? (char *)&lb->d_sh[0].cfg.user_bitrate_bps : (char *)&lb->d_streams[0].cfg.user_bitrate_bps;
      const long long rec = lb->kind ? (long long)sizeof(OaShStream) : (long long)sizeof(OaStream);
      for (int k = 0; k < m->ns; k++) {
         hipLaunchKernelGGL(oa_ms_budget_kernel, dim3((unsigned)((m->B + 63) / 64)), dim3(64), 0, s, k, m->ns, m->nc, m->B, (int)Fs, frame_size, (int)max_data_bytes,
               (const u8 *)m->d_pkc, (const i32 *)m->d_lc, (const u8 *)m->d_pkm, (const i32 *)m->d_lm, (int)m->stride, m->d_tot, m->d_err, m->d_budget_c, m->d_budget_m,
               vbr ? 0 : 1, last_rate, rec * lper, rec * (lper - 1), lb->channels);
         const int r = k < m->nc
            ? oa_encode_launch(m->bc, m->d_pc, m->d_M ? m->d_apc : nullptr, frame_size, frame_size, m->d_pkc, m->stride, OA_MS_FRAME_TMP, m->d_lc, m->d_rc, s, k, m->nc, m->B, m->d_budget_c)
            : oa_encode_launch(m->bm, m->d_pm, m->d_M ? m->d_apm : nullptr, frame_size, frame_size, m->d_pkm, m->stride, OA_MS_FRAME_TMP, m->d_lm, m->d_rm, s, k - m->nc, m->nm, m->B, m->d_budget_m);
         if (r != OPUS_OK) return r;
      }
   }
   if (!vbr) hipLaunchKernelGGL(oa_ms_pack_cbr_kernel, dim3((unsigned)m->B), dim3(64), 0, s, (const u8 *)m->d_pkc, (const i32 *)m->d_lc, (const u32 *)m->d_rc, m->nc,
         (const u8 *)m->d_pkm, (const i32 *)m->d_lm, (const u32 *)m->d_rm, m->nm, (int)m->stride, (u8 *)d_out, (int)out_stride, (int)max_data_bytes, (i32 *)d_lens, (u32 *)d_final_range, (const i32 *)m->d_err);
   else hipLaunchKernelGGL(oa_ms_pack_kernel, dim3((unsigned)m->B), dim3(64), 0, s, (const u8 *)m->d_pkc, (const i32 *)m->d_lc, (const u32 *)m->d_rc, m->nc,
         (const u8 *)m->d_pkm, (const i32 *)m->d_lm, (const u32 *)m->d_rm, m->nm, (int)m->stride, (u8 *)d_out, (int)out_stride, (int)max_data_bytes, (i32 *)d_lens, (u32 *)d_final_range,
         (const i32 *)(chained ? m->d_err : nullptr));
   HIPCHECK(hipGetLastError());
   return OPUS_OK;
}
/* host-pointer convenience (tests): pcm [B][frame_size][channels], out [B][out_stride] */
int opusgpu_ms_encode_batch(OpusGpuMsEncBatch *m, const opus_int16 *pcm, int frame_size, unsigned char *out, opus_int32 out_stride, opus_int32 max_data_bytes, opus_int32 *lens, opus_uint32 *final_range)
{
   if (!m || !pcm || !out || !lens || !final_range || frame_size <= 0 || out_stride <= 0) return OPUS_BAD_ARG;
   HIPCHECK(hipSetDevice(m->device));
   const size_t npcm = (size_t)m->B * frame_size * m->nch * sizeof(i16), nout = (size_t)m->B * out_stride;
   if (npcm > m->pcm_cap) { if (m->d_pcm) (void)hipFree(m->d_pcm); m->d_pcm = nullptr; m->pcm_cap = 0; HIPCHECK(hipMalloc((void **)&m->d_pcm, npcm)); m->pcm_cap = npcm; }
   if (nout > m->out_cap) { if (m->d_out) (void)hipFree(m->d_out); m->d_out = nullptr; m->out_cap = 0; HIPCHECK(hipMalloc((void **)&m->d_out, nout)); m->out_cap = nout; }
   HIPCHECK(hipMemcpy(m->d_pcm, pcm, npcm, hipMemcpyHostToDevice));
   const int r = opusgpu_ms_encode_batch_dev(m, m->d_pcm, frame_size, m->d_out, out_stride, max_data_bytes, m->d_lens, m->d_rng, nullptr);
   if (r != OPUS_OK) return r;
   hipStream_t s = m->bc ? m->bc->stream : m->bm->stream;
   HIPCHECK(hipStreamSynchronize(s));
   HIPCHECK(hipMemcpy(out, m->d_out, nout, hipMemcpyDeviceToHost));
   HIPCHECK(hipMemcpy(lens, m->d_lens, (size_t)m->B * 4, hipMemcpyDeviceToHost));
   HIPCHECK(hipMemcpy(final_range, m->d_rng, (size_t)m->B * 4, hipMemcpyDeviceToHost));
   return OPUS_OK;
}
}
#endif

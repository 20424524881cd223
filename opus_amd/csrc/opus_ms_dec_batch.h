/* opus_ms_dec_batch.h — device-resident batches of B identical multistream DECODERS (opus_multistream_decode, src/opus_multistream_decoder.c:178) and, with a
 * demixing matrix behind them, B projection decoders (opus_projection_decode, src/opus_projection_decoder.c:213; matrix arithmetic of src/mapping_matrix.c:257-286); plus the
 * projection ENCODER's mixing stage (mapping_matrix_multiply_channel_in_short, src/mapping_matrix.c:148-180) used by opus_ms_batch.h for mapping family 3.
 *
 * A multistream packet is the elementary packets back to back, all but the last in the self-delimiting framing of RFC 6716 Appendix B: the ordinary packet with ONE extra
 * length field (the size of the last / only / each CBR frame) right behind its header.  A frame-step of all B decoders is five launches on one HIP stream, nothing goes
 * through the host:
 *   oa_msd_parse_kernel     one wave per multistream packet; lane 0 walks the streams (the framing is serial by construction: where packet s+1 starts is only known once
 *                           packet s is parsed), checks what opus_multistream_packet_validate (:149) checks and writes one descriptor per stream
 *   oa_msd_scatter_kernel   one wave per (decoder, stream): the elementary packet -> its group's slot as a plain packet (the extra field dropped), coalesced
 *   the coupled group's and the mono group's decode kernels (oa_decode_kernel over B * nc and B * nm streams, state resident in HBM)
 *   oa_msd_merge_kernel     one wave per (decoder, output channel): the stream channel the mapping names -> the caller's interleaved output (muted channels: zeros);
 *                           per decoder the sample count (or the error code) and the XOR of the streams' final ranges
 *   oa_proj_demix_kernel    (projection) out[row] = sum over decoded channels of (M[row, c] * x[c] + 16384) >> 15, accumulated in int16 like the reference's `output[] +=`
 * A packet the validation turns away leaves every elementary decoder untouched (its streams' slots carry an over-long length: oa_decode_kernel returns before touching state)
 * and the decoder's entry of d_nsamples carries the reference's error code. */
#ifndef OPUS_AMD_MS_DEC_BATCH_H
#define OPUS_AMD_MS_DEC_BATCH_H

struct OaMsdDesc { i32 off, cut_at, cut_len, out_len; };              /* elementary packet: bytes [off, off + cut_at) ++ [off + cut_at + cut_len, off + out_len + cut_len) */

WV_DEV int oa_msd_toc_samples48(int toc)                                  /* samples per frame at 48 kHz (opus_packet_get_samples_per_frame, src/opus.c:174) */
{
   if (toc & 0x80) return 120 << ((toc >> 3) & 3);
   if ((toc & 0x60) == 0x60) return (toc & 0x08) ? 960 : 480;
   const int s = (toc >> 3) & 3;
   return s == 3 ? 2880 : 480 << s;
}
/* a length field at p (at most `left` bytes): value and size, or size 0 when it does not fit (parse_size, src/opus.c:116) */
WV_DEV int oa_msd_len_field(const u8 *p, int left, int *val)
{
   if (left < 1) return 0;
   if (p[0] < 252) { *val = p[0]; return 1; }
   if (left < 2) return 0;
   *val = 4 * p[1] + p[0];
   return 2;
}
/* one elementary packet of a multistream packet (opus_packet_parse_impl, src/opus.c:215-380): `framed` = self-delimited.  Returns the bytes it occupies (> 0) or
 * OPUS_INVALID_PACKET; *samples48 = its duration at 48 kHz */
WV_DEV int oa_msd_parse_one(const u8 *p, int left, int framed, OaMsdDesc *d, int *samples48)
{
   if (left < 1) return OPUS_INVALID_PACKET;
   const int toc = p[0], code = toc & 3, fs48 = oa_msd_toc_samples48(toc);
   int at = 1, n = 1, vbr = 0, pad = 0, body = 0, v = 0;
   if (code == 1) n = 2;
   else if (code == 2) {
      n = 2; vbr = 1;
      const int k = oa_msd_len_field(p + at, left - at, &v); if (!k) return OPUS_INVALID_PACKET;
      at += k; body = v;
   } else if (code == 3) {
      if (left - at < 1) return OPUS_INVALID_PACKET;
      const int cb = p[at++];
      n = cb & 0x3F; vbr = (cb & 0x80) != 0;
      if (n == 0 || n * fs48 > 5760) return OPUS_INVALID_PACKET;
      if (cb & 0x40) {
         int b;
         do { if (left - at < 1) return OPUS_INVALID_PACKET; b = p[at++]; pad += b == 255 ? 254 : b; } while (b == 255);
      }
      if (vbr) for (int i = 0; i < n - 1; i++) { const int k = oa_msd_len_field(p + at, left - at, &v); if (!k) return OPUS_INVALID_PACKET; at += k; body += v; }
   }
   if (left - at - pad < 0) return OPUS_INVALID_PACKET;
   int total;
   if (framed) {
      const int k = oa_msd_len_field(p + at, left - at - pad, &v); if (!k) return OPUS_INVALID_PACKET;
      d->cut_at = at; d->cut_len = k;
      const int rest = left - at - k - pad - body;                                    /* what is left for the frame(s) the field sizes */
      if (rest < 0) return OPUS_INVALID_PACKET;
      if (vbr || n == 1) { if (v > rest) return OPUS_INVALID_PACKET; body += v; }
      else { if (v * n > rest) return OPUS_INVALID_PACKET; body = v * n; }
      if (v > 1275) return OPUS_INVALID_PACKET;
      total = at + k + body + pad;
   } else {
      d->cut_at = at; d->cut_len = 0;
      const int rest = left - at - pad - body;
      if (rest < 0) return OPUS_INVALID_PACKET;
      if (!(vbr || n == 1)) { if (rest % n) return OPUS_INVALID_PACKET; if (rest / n > 1275) return OPUS_INVALID_PACKET; }
      else if (rest > 1275) return OPUS_INVALID_PACKET;
      total = left;
   }
   d->out_len = total - d->cut_len;
   *samples48 = n * fs48;
   return total;
}

/* status[b]: samples per channel the packet holds at the decoder's rate (> 0), 0 for a lost packet (len 0: every stream conceals), or a negative OPUS_* code */
extern "C" __global__ void __launch_bounds__(64)
oa_msd_parse_kernel(const u8 *data, int stride, const i32 *lens, int ns, int Fs, int frame_size, OaMsdDesc *desc, i32 *status, int slot /* bytes an elementary packet may have in this batch */)
{
   const int b = blockIdx.x;
   if (threadIdx.x != 0) return;
   const u8 *p = data + (size_t)b * stride;
   const int len = lens[b];
   OaMsdDesc *d = desc + (size_t)b * ns;
   if (len < 0 || len > stride) { status[b] = OPUS_BAD_ARG; return; }
   if (len == 0) { for (int s = 0; s < ns; s++) { d[s].off = 0; d[s].cut_at = 0; d[s].cut_len = 0; d[s].out_len = 0; } status[b] = 0; return; }
   if (len < 2 * ns - 1) { status[b] = OPUS_INVALID_PACKET; return; }
   int at = 0, samples = 0;
   for (int s = 0; s < ns; s++) {
      if (len - at <= 0) { status[b] = OPUS_INVALID_PACKET; return; }
      int s48 = 0;
      const int used = oa_msd_parse_one(p + at, len - at, s != ns - 1, &d[s], &s48);
      if (used < 0) { status[b] = used; return; }
      const int smp = (int)((long long)s48 * Fs / 48000);
      if (s != 0 && smp != samples) { status[b] = OPUS_INVALID_PACKET; return; }
      samples = smp;
      d[s].off = at;
      at += used;
   }
   /* an elementary packet beyond the batch's slot (a limit of this library: include/opus_amd.h) turns the WHOLE multistream packet away here, before any elementary decoder
    * has moved -- not its own stream alone after its siblings have decoded theirs */
   for (int s = 0; s < ns; s++) if (d[s].out_len > slot) { status[b] = OPUS_BAD_ARG; return; }
   status[b] = samples > frame_size ? OPUS_BUFFER_TOO_SMALL : samples;
}
/* lens of a turned-away packet's streams: beyond the slot, so that oa_decode_kernel answers OPUS_BAD_ARG without touching the stream */
extern "C" __global__ void __launch_bounds__(64)
oa_msd_scatter_kernel(const u8 *data, int stride, const OaMsdDesc *desc, const i32 *status, int ns, int nc, u8 *pkc, i32 *lc, u8 *pkm, i32 *lm, int slot)
{
   const int b = (int)blockIdx.x / ns, s = (int)blockIdx.x - b * ns, nm = ns - nc;
   u8 *dst = s < nc ? pkc + ((size_t)b * nc + s) * slot : pkm + ((size_t)b * nm + (s - nc)) * slot;
   i32 *l = s < nc ? lc + (size_t)b * nc + s : lm + (size_t)b * nm + (s - nc);
   const int st = status[b];
   if (st < 0) { if (threadIdx.x == 0) *l = slot + 1; return; }
   const OaMsdDesc d = desc[(size_t)b * ns + s];
   if (d.out_len > slot) { if (threadIdx.x == 0) *l = slot + 1; return; }
   const u8 *src = data + (size_t)b * stride + d.off;
   for (int i = threadIdx.x; i < d.out_len; i += 64) dst[i] = src[i < d.cut_at ? i : i + d.cut_len];
   if (threadIdx.x == 0) *l = d.out_len;
}
/* chan[c]: 255 = muted, < 2 nc = coupled stream c / 2 (left / right by parity), else mono stream (c - 2 nc) */
extern "C" __global__ void __launch_bounds__(64)
oa_msd_merge_kernel(const i16 *oc, const i32 *nsc, const u32 *rc, const i16 *om, const i32 *nsm, const u32 *rm, int nc, int nm, const i32 *chan, int nch, int frame_size,
      const i32 *status, i16 *pcm, i32 *nsamples, u32 *rngs)
{
   const int b = (int)blockIdx.x / nch, c = (int)blockIdx.x - b * nch, ns = nc + nm;
   /* what the streams decoded: all equal (validated), or the first error among them */
   int n = status[b], err = n < 0 ? n : 0;
   if (err == 0) {
      for (int s = 0; s < ns; s++) { const int k = s < nc ? nsc[(size_t)b * nc + s] : nsm[(size_t)b * nm + (s - nc)]; if (k <= 0) { err = k == 0 ? OPUS_INTERNAL_ERROR : k; break; } n = k; }
   }
   if (c == 0 && threadIdx.x == 0) {
      u32 x = 0;
      if (err == 0) for (int s = 0; s < ns; s++) x ^= s < nc ? rc[(size_t)b * nc + s] : rm[(size_t)b * nm + (s - nc)];
      nsamples[b] = err ? err : n; rngs[b] = x;
   }
   if (err) return;
   i16 *dst = pcm + (size_t)b * frame_size * nch + c;
   const int idx = chan[c];
   if (idx == 255) { for (int i = threadIdx.x; i < n; i += 64) dst[(size_t)i * nch] = 0; return; }
   if (idx < 2 * nc) { const i16 *src = oc + ((size_t)b * nc + (idx >> 1)) * frame_size * 2 + (idx & 1); for (int i = threadIdx.x; i < n; i += 64) dst[(size_t)i * nch] = src[2 * i]; }
   else { const i16 *src = om + ((size_t)b * nm + (idx - 2 * nc)) * frame_size; for (int i = threadIdx.x; i < n; i += 64) dst[(size_t)i * nch] = src[i]; }
}
/* projection: M int16 [cols][rows] column-major (as opus_projection_decoder_create receives it); x [B][frame_size][nch] decoded channels -> out [B][frame_size][rows];
 * one workgroup = one decoder's 64 consecutive samples, one lane per sample; matrix and the sample block staged in LDS */
#define OA_PROJ_MAXC 40
extern "C" __global__ void __launch_bounds__(64)
oa_proj_demix_kernel(const i16 *M, int rows, int cols, const i16 *x, int nch, int frame_size, const i32 *nsamples, i16 *out)
{
   __shared__ i16 sM[OA_PROJ_MAXC * OA_PROJ_MAXC], sx[64 * OA_PROJ_MAXC];
   const int nblk = (frame_size + 63) / 64, b = (int)blockIdx.x / nblk, i0 = ((int)blockIdx.x - b * nblk) * 64, n = nsamples[b];
   if (n <= 0 || i0 >= n) return;
   for (int k = threadIdx.x; k < rows * cols; k += 64) sM[k] = M[k];
   const int cnt = imin(64, n - i0);
   const i16 *src = x + ((size_t)b * frame_size + i0) * nch;
   for (int k = threadIdx.x; k < cnt * nch; k += 64) sx[k] = src[k];
   __syncthreads();
   const int i = threadIdx.x;
   if (i < cnt) {
      i16 *dst = out + ((size_t)b * frame_size + i0 + i) * rows;
      for (int r = 0; r < rows; r++) {
         i16 acc = 0;
         for (int c = 0; c < cols && c < nch; c++) acc = (i16)(acc + (i16)(((i32)sM[rows * c + r] * (i32)sx[i * nch + c] + 16384) >> 15));
         dst[r] = acc;
      }
   }
}
/* projection encoder: y[row] = SAT16((((sum over col of (M[row, col] * x[col]) >> 8) + 64) >> 7)) (mapping_matrix_multiply_channel_in_short, fixed point); M [C][C] column-major */
extern "C" __global__ void __launch_bounds__(64)
oa_proj_mix_kernel(const i16 *M, int C, const i16 *x, int frame_size, i16 *y)
{
   __shared__ i16 sM[OA_PROJ_MAXC * OA_PROJ_MAXC], sx[64 * OA_PROJ_MAXC];
   const int nblk = (frame_size + 63) / 64, b = (int)blockIdx.x / nblk, i0 = ((int)blockIdx.x - b * nblk) * 64;
   for (int k = threadIdx.x; k < C * C; k += 64) sM[k] = M[k];
   const int cnt = imin(64, frame_size - i0);
   const i16 *src = x + ((size_t)b * frame_size + i0) * C;
   for (int k = threadIdx.x; k < cnt * C; k += 64) sx[k] = src[k];
   __syncthreads();
   const int i = threadIdx.x;
   if (i < cnt) {
      i16 *dst = y + ((size_t)b * frame_size + i0 + i) * C;
      for (int r = 0; r < C; r++) {
         i32 acc = 0;
         for (int c = 0; c < C; c++) acc += ((i32)sM[C * c + r] * (i32)sx[i * C + c]) >> 8;
         const i32 v = (acc + 64) >> 7;
         dst[r] = (i16)(v > 32767 ? 32767 : v < -32768 ? -32768 : v);
      }
   }
}

struct OpusGpuMsDecBatch {
   opus_int32 B, Fs; int nch, ns, nc, nm, device;
   OpusGpuDecBatch *bc, *bm;
   i32 *d_chan; OaMsdDesc *d_desc; i32 *d_status;
   u8 *d_pkc, *d_pkm; i32 *d_lc, *d_lm, *d_nsc, *d_nsm; u32 *d_rc, *d_rm; opus_int32 slot;
   i16 *d_oc, *d_om; size_t oc_cap, om_cap;
   i16 *d_M; int rows, cols; i16 *d_tmp; size_t tmp_cap;                  /* projection: demixing matrix, the decoded channels before it */
   /* staging of the host-pointer entry */
   u8 *d_data; size_t data_cap; i16 *d_pcm; size_t pcm_cap; i32 *d_lens, *d_ns; u32 *d_rng;
};

extern "C" {
void opusgpu_ms_dec_batch_destroy(OpusGpuMsDecBatch *m)
{
   if (!m) return;
   if (m->bc) opusgpu_dec_batch_destroy(m->bc);
   if (m->bm) opusgpu_dec_batch_destroy(m->bm);
   (void)hipSetDevice(m->device);
   void *bufs[] = {m->d_chan, m->d_desc, m->d_status, m->d_pkc, m->d_pkm, m->d_lc, m->d_lm, m->d_nsc, m->d_nsm, m->d_rc, m->d_rm, m->d_oc, m->d_om, m->d_M, m->d_tmp, m->d_data, m->d_pcm, m->d_lens, m->d_ns, m->d_rng};
   for (void *p : bufs) if (p) (void)hipFree(p);
   delete m;
}
/* B decoders of opus_multistream_decoder_create(Fs, channels, streams, coupled_streams, mapping) (include/opus_multistream.h:461) */
OpusGpuMsDecBatch *opusgpu_ms_dec_batch_create(opus_int32 B, opus_int32 Fs, int channels, int streams, int coupled_streams, const unsigned char *mapping, int device, int *error)
{
   int err = OPUS_OK;
   OpusGpuMsDecBatch *m = nullptr;
   OaLayout lay; memset(&lay, 0, sizeof lay);
   if (B <= 0 || !mapping || channels > 255 || channels < 1 || coupled_streams > streams || streams < 1 || coupled_streams < 0 || streams > 255 - coupled_streams || !oa_fs_ok(Fs)) err = OPUS_BAD_ARG;
   if (err == OPUS_OK) {
      lay.nb_channels = channels; lay.nb_streams = streams; lay.nb_coupled_streams = coupled_streams;
      for (int i = 0; i < channels; i++) lay.mapping[i] = mapping[i];
      if (!oa_validate_layout(&lay)) err = OPUS_BAD_ARG;
   }
   if (err == OPUS_OK) {
      m = new OpusGpuMsDecBatch();
      memset(m, 0, sizeof(*m));
      m->B = B; m->Fs = Fs; m->nch = channels; m->ns = streams; m->nc = coupled_streams; m->nm = streams - coupled_streams; m->device = device;
      if (m->nc) m->bc = opusgpu_dec_batch_create(B * m->nc, Fs, 2, device, &err);
      if (err == OPUS_OK && m->nm) m->bm = opusgpu_dec_batch_create(B * m->nm, Fs, 1, device, &err);
      std::vector<i32> chan((size_t)channels);
      for (int c = 0; c < channels; c++) chan[c] = mapping[c];
      m->slot = 1280 * 6 + 16;                                            /* a plain elementary packet: up to 120 ms of maximum-size frames would be 48 x 1275; the batch takes what six 20 ms frames need and turns longer ones away */
      const size_t nstr = (size_t)B * streams;
      if (err == OPUS_OK) {
         const bool ok = hipSetDevice(device) == hipSuccess && hipMalloc((void **)&m->d_chan, chan.size() * 4) == hipSuccess &&
               hipMemcpy(m->d_chan, chan.data(), chan.size() * 4, hipMemcpyHostToDevice) == hipSuccess &&
               hipMalloc((void **)&m->d_desc, nstr * sizeof(OaMsdDesc)) == hipSuccess && hipMalloc((void **)&m->d_status, (size_t)B * 4) == hipSuccess &&
               hipMalloc((void **)&m->d_pkc, (size_t)B * (m->nc ? m->nc : 1) * m->slot) == hipSuccess && hipMalloc((void **)&m->d_pkm, (size_t)B * (m->nm ? m->nm : 1) * m->slot) == hipSuccess &&
               hipMalloc((void **)&m->d_lc, nstr * 4 + 4) == hipSuccess && hipMalloc((void **)&m->d_lm, nstr * 4 + 4) == hipSuccess &&
               hipMalloc((void **)&m->d_nsc, nstr * 4 + 4) == hipSuccess && hipMalloc((void **)&m->d_nsm, nstr * 4 + 4) == hipSuccess &&
               hipMalloc((void **)&m->d_rc, nstr * 4 + 4) == hipSuccess && hipMalloc((void **)&m->d_rm, nstr * 4 + 4) == hipSuccess &&
               hipMalloc((void **)&m->d_lens, (size_t)B * 4) == hipSuccess && hipMalloc((void **)&m->d_ns, (size_t)B * 4) == hipSuccess && hipMalloc((void **)&m->d_rng, (size_t)B * 4) == hipSuccess;
         if (!ok) err = OPUS_ALLOC_FAIL;
      }
      if (err != OPUS_OK) { opusgpu_ms_dec_batch_destroy(m); m = nullptr; }
   }
   if (error) *error = err;
   return m;
}
/* B projection decoders of opus_projection_decoder_create(Fs, channels, streams, coupled_streams, demixing_matrix, size) (include/opus_projection.h:418): the multistream
 * decoders with the identity mapping, the demixing stage behind them */
OpusGpuMsDecBatch *opusgpu_projection_dec_batch_create(opus_int32 B, opus_int32 Fs, int channels, int streams, int coupled_streams, const unsigned char *demixing_matrix,
      opus_int32 demixing_matrix_size, int device, int *error)
{
   const int nin = streams + coupled_streams;
   if (!demixing_matrix || channels < 1 || channels > OA_PROJ_MAXC || nin > OA_PROJ_MAXC || nin * channels * 2 != demixing_matrix_size) { if (error) *error = OPUS_BAD_ARG; return nullptr; }
   unsigned char mapping[255];
   for (int i = 0; i < channels; i++) mapping[i] = (unsigned char)i;
   int err = OPUS_OK;
   OpusGpuMsDecBatch *m = opusgpu_ms_dec_batch_create(B, Fs, channels, streams, coupled_streams, mapping, device, &err);
   if (m) {
      std::vector<i16> M((size_t)nin * channels);
      for (int i = 0; i < nin * channels; i++) M[i] = (i16)(demixing_matrix[2 * i + 1] << 8 | demixing_matrix[2 * i]);
      m->rows = channels; m->cols = nin;
      if (hipMalloc((void **)&m->d_M, M.size() * 2) != hipSuccess || hipMemcpy(m->d_M, M.data(), M.size() * 2, hipMemcpyHostToDevice) != hipSuccess) { err = OPUS_ALLOC_FAIL; opusgpu_ms_dec_batch_destroy(m); m = nullptr; }
   }
   if (error) *error = err;
   return m;
}
/* one packet per decoder, everything in HBM: d_data [B][stride] multistream packets of d_lens [B] bytes (0 = lost: every stream conceals frame_size samples) ->
 * d_pcm [B][frame_size][channels] int16 interleaved, d_nsamples [B] (samples per channel, or a negative OPUS_* code for that decoder), d_final_range [B] */
int opusgpu_ms_decode_batch_dev(OpusGpuMsDecBatch *m, const unsigned char *d_data, opus_int32 stride, const opus_int32 *d_lens, opus_int16 *d_pcm, int frame_size,
      opus_int32 *d_nsamples, opus_uint32 *d_final_range, void *hip_stream)
{
   if (!m || !d_data || !d_lens || !d_pcm || !d_nsamples || !d_final_range || stride <= 0 || frame_size <= 0) return OPUS_BAD_ARG;
   if (frame_size > m->Fs / 25 * 3) return OPUS_BAD_ARG;                   /* (the classic entry point clamps; a batch of fixed-size rows does not guess) */
   HIPCHECK(hipSetDevice(m->device));
   hipStream_t s = hip_stream ? (hipStream_t)hip_stream : (m->bc ? m->bc->stream : m->bm->stream);
   const size_t need_c = (size_t)m->B * m->nc * frame_size * 2 * sizeof(i16), need_m = (size_t)m->B * m->nm * frame_size * sizeof(i16);
   if (need_c > m->oc_cap) { HIPCHECK(hipStreamSynchronize(s)); if (m->d_oc) (void)hipFree(m->d_oc); m->d_oc = nullptr; m->oc_cap = 0; HIPCHECK(hipMalloc((void **)&m->d_oc, need_c)); m->oc_cap = need_c; }
   if (need_m > m->om_cap) { HIPCHECK(hipStreamSynchronize(s)); if (m->d_om) (void)hipFree(m->d_om); m->d_om = nullptr; m->om_cap = 0; HIPCHECK(hipMalloc((void **)&m->d_om, need_m)); m->om_cap = need_m; }
   i16 *merged = (i16 *)d_pcm;
   if (m->d_M) {
      const size_t need_t = (size_t)m->B * frame_size * m->nch * sizeof(i16);
      if (need_t > m->tmp_cap) { HIPCHECK(hipStreamSynchronize(s)); if (m->d_tmp) (void)hipFree(m->d_tmp); m->d_tmp = nullptr; m->tmp_cap = 0; HIPCHECK(hipMalloc((void **)&m->d_tmp, need_t)); m->tmp_cap = need_t; }
      merged = m->d_tmp;
   }
   hipLaunchKernelGGL(oa_msd_parse_kernel, dim3((unsigned)m->B), dim3(64), 0, s, (const u8 *)d_data, (int)stride, (const i32 *)d_lens, m->ns, (int)m->Fs, frame_size, m->d_desc, m->d_status, (int)m->slot);
   hipLaunchKernelGGL(oa_msd_scatter_kernel, dim3((unsigned)(m->B * m->ns)), dim3(64), 0, s, (const u8 *)d_data, (int)stride, (const OaMsdDesc *)m->d_desc, (const i32 *)m->d_status, m->ns, m->nc,
         m->d_pkc, m->d_lc, m->d_pkm, m->d_lm, (int)m->slot);
   HIPCHECK(hipGetLastError());
   if (m->nc) { const int r = opusgpu_decode_batch_dev(m->bc, m->d_pkc, m->slot, m->d_lc, m->d_oc, frame_size, m->d_nsc, m->d_rc, s); if (r != OPUS_OK) return r; }
   if (m->nm) { const int r = opusgpu_decode_batch_dev(m->bm, m->d_pkm, m->slot, m->d_lm, m->d_om, frame_size, m->d_nsm, m->d_rm, s); if (r != OPUS_OK) return r; }
   hipLaunchKernelGGL(oa_msd_merge_kernel, dim3((unsigned)(m->B * m->nch)), dim3(64), 0, s, (const i16 *)m->d_oc, (const i32 *)m->d_nsc, (const u32 *)m->d_rc, (const i16 *)m->d_om, (const i32 *)m->d_nsm,
         (const u32 *)m->d_rm, m->nc, m->nm, (const i32 *)m->d_chan, m->nch, frame_size, (const i32 *)m->d_status, merged, (i32 *)d_nsamples, (u32 *)d_final_range);
   if (m->d_M)
      hipLaunchKernelGGL(oa_proj_demix_kernel, dim3((unsigned)(((frame_size + 63) / 64) * m->B)), dim3(64), 0, s, (const i16 *)m->d_M, m->rows, m->cols, (const i16 *)merged, m->nch, frame_size,
            (const i32 *)d_nsamples, (i16 *)d_pcm);
   HIPCHECK(hipGetLastError());
   return OPUS_OK;
}
/* host-pointer convenience (tests): data [B][stride], pcm [B][frame_size][channels] */
int opusgpu_ms_decode_batch(OpusGpuMsDecBatch *m, const unsigned char *data, opus_int32 stride, const opus_int32 *lens, opus_int16 *pcm, int frame_size, opus_int32 *nsamples, opus_uint32 *final_range)
{
   if (!m || !data || !lens || !pcm || !nsamples || !final_range || stride <= 0 || frame_size <= 0) return OPUS_BAD_ARG;
   HIPCHECK(hipSetDevice(m->device));
   const size_t nd = (size_t)m->B * stride, npcm = (size_t)m->B * frame_size * m->nch * sizeof(i16);
   if (nd > m->data_cap) { if (m->d_data) (void)hipFree(m->d_data); m->d_data = nullptr; m->data_cap = 0; HIPCHECK(hipMalloc((void **)&m->d_data, nd)); m->data_cap = nd; }
   if (npcm > m->pcm_cap) { if (m->d_pcm) (void)hipFree(m->d_pcm); m->d_pcm = nullptr; m->pcm_cap = 0; HIPCHECK(hipMalloc((void **)&m->d_pcm, npcm)); m->pcm_cap = npcm; }
   HIPCHECK(hipMemcpy(m->d_data, data, nd, hipMemcpyHostToDevice));
   HIPCHECK(hipMemcpy(m->d_lens, lens, (size_t)m->B * 4, hipMemcpyHostToDevice));
   HIPCHECK(hipMemset(m->d_pcm, 0, npcm));
   const int r = opusgpu_ms_decode_batch_dev(m, m->d_data, stride, m->d_lens, m->d_pcm, frame_size, m->d_ns, m->d_rng, nullptr);
   if (r != OPUS_OK) return r;
   hipStream_t s = m->bc ? m->bc->stream : m->bm->stream;
   HIPCHECK(hipStreamSynchronize(s));
   HIPCHECK(hipMemcpy(pcm, m->d_pcm, npcm, hipMemcpyDeviceToHost));
   HIPCHECK(hipMemcpy(nsamples, m->d_ns, (size_t)m->B * 4, hipMemcpyDeviceToHost));
   HIPCHECK(hipMemcpy(final_range, m->d_rng, (size_t)m->B * 4, hipMemcpyDeviceToHost));
   return OPUS_OK;
}
}
#endif

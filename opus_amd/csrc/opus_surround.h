/* opus_surround.h — the masking analysis of the surround (mapping family 1, > 2 channels) multistream encoder, src/opus_multistream_encoder.c:230 surround_analysis.
 *
 * Device part: one wavefront per input channel.  The channel's samples are pre-emphasised (an FIR: y[i] = x[i] - .85 x[i-1], zero-stuffed up to 48 kHz below
 * that rate, celt_encoder.c:557) straight into LDS, every 20 ms (or the one shorter frame) goes through the long forward MDCT of celt_mdct.h with the
 * carried 120-sample overlap, band energies are taken one lane per band (bands.c:95), the maximum over the frames of the call is converted to the log domain
 * (quant_bands.c:553) and spread with -6 dB/band upwards and -12 dB/band downwards (:305-309).  Output: bandLogE[channel][21]; state: the overlap tail and
 * the pre-emphasis memory of the channel (both live in the OpusMSEncoder blob).
 * Host part: what couples the channels -- the three position masks (left / centre / right) by logSum accumulation and the per-channel signal-to-mask
 * ratios (:310-376) -- a few hundred integer operations per call. */
#ifndef OPUS_AMD_SURROUND_H
#define OPUS_AMD_SURROUND_H

struct SurroundLds { i32 body[OA_MAX_FRAME]; i32 freq[OA_MAX_FRAME]; i32 bandE[NBE]; i32 tmpE[NBE]; int aux[32]; };

WV_DEV void oa_surround_channel_wave(WV_LDS SurroundLds *S, const i16 *pcm, int len, int channels, int c, int Fs, i32 *mem, i32 *preemph_mem, i32 *bandLogE_out)
{
   const int up = 48000 / Fs, frame_size = len * up, overlap = OA_OVERLAP;
   int LM;
   for (LM = 0; LM < 3; LM++) if (120 << LM == frame_size) break;
   const int freq_size = 120 << LM, nb_frames = frame_size / freq_size, shift = 3 - LM;
   const int lane = wv_lane();
   i32 m0 = preemph_mem[c];
   FOR_LANES(i, NBE) S->bandE[i] = 0;
   for (int f = 0; f < nb_frames; f++) {
      const int base = f * freq_size;                                                    /* first 48 kHz sample of this frame */
      wv_sync();
      FOR_LANES(i, freq_size) {                                                          /* x48[j] = pcm[j / up] when up divides j, else 0; y = x48[j] - .85 x48[j-1] */
         const int j = base + i, q = j / up, r = j - q * up;
         const i32 x = r == 0 ? shl32((i32)pcm[(size_t)q * channels + c], SIG_SHIFT) : 0;
         i32 m;
         if (j == 0) m = m0;
         else { const int q1 = (j - 1) / up, r1 = (j - 1) - q1 * up; m = r1 == 0 ? mult16_32_q15(27853, shl32((i32)pcm[(size_t)q1 * channels + c], SIG_SHIFT)) : 0; }
         S->body[i] = x - m;
      }
      wv_sync();
      mdct_forward_blocks(mem + c * overlap, S->body, S->freq, shift, 1, S->aux);
      if (up != 1) { const int bound = freq_size / up; FOR_LANES(i, freq_size) S->freq[i] = i < bound ? S->freq[i] * up : 0; wv_sync(); }
      /* the overlap of the next frame = the last 120 pre-emphasised samples of this one; the MDCT above has consumed the old head */
      FOR_LANES(i, overlap) mem[c * overlap + i] = S->body[freq_size - overlap + i];
      FOR_LANES(i, NBE) {                                                                 /* compute_band_energies (bands.c:95), one lane per band */
         const WV_LDS i32 *x = &S->freq[ct_eBands[i] << LM];
         const int n = (ct_eBands[i + 1] - ct_eBands[i]) << LM;
         i32 mx = 0, mn = 0, sum = 0, E;
         for (int j = 0; j < n; j++) { mx = imax(mx, x[j]); mn = imin(mn, x[j]); }
         const i32 maxval = imax(mx, neg32(mn));
         if (maxval > 0) {
            const int sh = imax(0, 30 - celt_ilog2(maxval + (maxval >> 14) + 1) - ((((ct_logN[i] + 7) >> BITRES) + LM + 1) >> 1));
            for (int j = 0; j < n; j++) { const i32 v = shl32(x[j], sh); sum = add32(sum, mult32_32_q31(v, v)); }
            E = imax(maxval, pshr32(fx_sqrt32(sum >> 1), sh));
         } else E = EPSILON;
         S->bandE[i] = imax(S->bandE[i], E);                                              /* several frames: the larger energy counts */
      }
      wv_sync();
   }
   LANE0 {
      i32 lg[NBE];
      for (int i = 0; i < NBE; i++) lg[i] = fx_log2_db(S->bandE[i]) - shl32((i32)ct_eMeans[i], DB_SHIFT - 4) + GC(2.f);    /* amp2Log2 */
      for (int i = 1; i < NBE; i++) lg[i] = imax(lg[i], lg[i - 1] - GC(1.f));
      for (int i = NBE - 2; i >= 0; i--) lg[i] = imax(lg[i], lg[i + 1] - GC(2.f));
      for (int i = 0; i < NBE; i++) bandLogE_out[NBE * c + i] = lg[i];
      /* pre-emphasis memory: .85 x the last (zero-stuffed) sample */
      preemph_mem[c] = up > 1 ? 0 : mult16_32_q15(27853, shl32((i32)pcm[(size_t)(len - 1) * channels + c], SIG_SHIFT));
   }
   (void)lane;
}

#endif

/* opus_surround_host.h — host part of the surround masking analysis (src/opus_multistream_encoder.c:138-230, :310-376): what couples the channels.  The per-channel
 * log band energies come from oa_surround_kernel (opus_surround.h); here the three position masks (left / centre / right) are accumulated with logSum and turned
 * into per-channel signal-to-mask ratios -- a few hundred integer operations per call -- and the kernel is driven (device buffers cached per process). */
#ifndef OPUS_AMD_SURROUND_HOST_H
#define OPUS_AMD_SURROUND_HOST_H
/* ---- host part: couple the channels (src/opus_multistream_encoder.c:138-230, :310-376) ---- */
/* position of each channel of the vorbis layouts in the left-centre-right mix: 0 = not mixed (LFE), 1 = left, 2 = centre, 3 = right */
static const unsigned char oa_surround_pos[9][8] = {{0}, {0}, {0}, {1, 2, 3}, {1, 3, 1, 3}, {1, 2, 3, 1, 3}, {1, 2, 3, 1, 3, 0}, {1, 2, 3, 1, 3, 2, 0}, {1, 2, 3, 1, 3, 1, 3, 0}};
/* celt_log2 of the fixed-point build (celt/mathops.h:391): Q14 in, Q10 out, 4th-order polynomial on the mantissa */
WV_HD int oa_log2_q10(opus_int32 x)
{
   const opus_int16 C[5] = {-6801 + (1 << (13 - 10)), 15746, -5217, 2545, -1401};
   if (x == 0) return -32767;
   int i = 31; while (!(x >> i)) i--;
   const opus_int16 n = (opus_int16)((i - 15 > 0 ? x >> (i - 15) : x << (15 - i)) - 32768 - 16384);
   opus_int16 f = C[4];
   for (int k = 3; k >= 0; k--) f = (opus_int16)(C[k] + (((opus_int32)n * f) >> 15));
   return ((i - 13) << 10) + (f >> (14 - 10));
}
/* log2(2^a + 2^b) in Q24 (DB_SHIFT), piecewise linear in the difference, half-unit steps (logSum :193).  The reference's function is declared to return opus_val16
 * -- 16 bits in the fixed-point build -- so what reaches the masks is the low half of the Q24 sum, sign-extended (gcc's modulo conversion); the elementary encoders'
 * allocation follows from exactly that value, so it is kept bit for bit (found by tools/encode_trace_compare.py on the reference's surround_analysis_uninit regression) */
WV_HD opus_int32 oa_logsum(opus_int32 a, opus_int32 b)
{
   const opus_int32 tab[17] = {8388608, 4907022, 2700528, 1425434, 733691, 372406, 187635, 94181, 47183, 0, 0, 0, 0, 0, 0, 0, 0};   /* GCONST(.5, .2924813, .1609640, ...) */
   const opus_int32 hi = a > b ? a : b, diff = a > b ? a - b : b - a;
   if (!(diff < (8 << 24))) return (opus_int16)hi;
   const int low = diff >> 23;
   const opus_int32 frac = (diff - (low << 23)) >> 8;                                      /* VSHR32(., DB_SHIFT - 16): Q15 of a half unit */
   return (opus_int16)(hi + tab[low] + (opus_int32)(((long long)(opus_int16)frac * (tab[low + 1] - tab[low])) >> 15));
}
/* bandLogE[channels][21] (the per-channel spread log energies) -> signal-to-mask ratios in place */
static void oa_surround_couple(opus_int32 *bandLogE, int channels)
{
   const unsigned char *pos = oa_surround_pos[channels <= 8 ? channels : 0];
   opus_int32 mask[3][21];
   for (int k = 0; k < 3; k++) for (int i = 0; i < 21; i++) mask[k][i] = -(28 << 24);
   for (int c = 0; c < channels; c++) for (int i = 0; i < 21; i++) {
      const opus_int32 e = bandLogE[21 * c + i];
      if (pos[c] == 1) mask[0][i] = oa_logsum(mask[0][i], e);
      else if (pos[c] == 3) mask[2][i] = oa_logsum(mask[2][i], e);
      else if (pos[c] == 2) { mask[0][i] = oa_logsum(mask[0][i], e - (1 << 23)); mask[2][i] = oa_logsum(mask[2][i], e - (1 << 23)); }
   }
   for (int i = 0; i < 21; i++) mask[1][i] = mask[0][i] < mask[2][i] ? mask[0][i] : mask[2][i];
   /* channel_offset = HALF16(celt_log2(QCONST32(2.f, 14) / (channels - 1))) -- an opus_val16 in Q10 that the reference adds to the Q24 masks as is (:345): kept bit for bit */
   const opus_int32 channel_offset = oa_log2_q10(32768 / (channels - 1)) >> 1;
   for (int c = 0; c < channels; c++) for (int i = 0; i < 21; i++)
      bandLogE[21 * c + i] = pos[c] != 0 ? bandLogE[21 * c + i] - (mask[pos[c] - 1][i] + channel_offset) : 0;
}

/* surround_analysis (:230): pcm = len samples of `channels` interleaved int16 at Fs; mem[channels][120] / preemph_mem[channels] = the analysis state in the encoder
 * blob; bandSMR[channels][21] out */
static int oa_surround_analysis(const opus_int16 *pcm, int len, int channels, opus_int32 Fs, opus_int32 *mem, opus_int32 *preemph_mem, opus_int32 *bandSMR)
{
   static opus_int16 *d_pcm = nullptr; static size_t pcm_cap = 0;
   static opus_int32 *d_state = nullptr;                                                /* [8][120] window memory | [8] pre-emphasis memory | [8][21] result */
   HIPCHECK(hipSetDevice(0));
   const size_t npcm = (size_t)len * channels * sizeof(opus_int16);
   if (npcm > pcm_cap) { if (d_pcm) (void)hipFree(d_pcm); HIPCHECK(hipMalloc((void **)&d_pcm, npcm)); pcm_cap = npcm; }
   if (!d_state) HIPCHECK(hipMalloc((void **)&d_state, sizeof(opus_int32) * (8 * 120 + 8 + 8 * 21)));
   opus_int32 *d_mem = d_state, *d_pre = d_state + 8 * 120, *d_out = d_pre + 8;
   HIPCHECK(hipMemcpy(d_pcm, pcm, npcm, hipMemcpyHostToDevice));
   HIPCHECK(hipMemcpy(d_mem, mem, sizeof(opus_int32) * 120 * (size_t)channels, hipMemcpyHostToDevice));
   HIPCHECK(hipMemcpy(d_pre, preemph_mem, sizeof(opus_int32) * (size_t)channels, hipMemcpyHostToDevice));
   hipLaunchKernelGGL(oa_surround_kernel, dim3((unsigned)channels), dim3(64), 0, (hipStream_t)0, (const i16 *)d_pcm, len, channels, (int)Fs, (i32 *)d_mem, (i32 *)d_pre, (i32 *)d_out);
   HIPCHECK(hipGetLastError());
   HIPCHECK(hipMemcpy(mem, d_mem, sizeof(opus_int32) * 120 * (size_t)channels, hipMemcpyDeviceToHost));
   HIPCHECK(hipMemcpy(preemph_mem, d_pre, sizeof(opus_int32) * (size_t)channels, hipMemcpyDeviceToHost));
   HIPCHECK(hipMemcpy(bandSMR, d_out, sizeof(opus_int32) * 21 * (size_t)channels, hipMemcpyDeviceToHost));
   oa_surround_couple(bandSMR, channels);
   return OPUS_OK;
}
#endif

/* silk_lpc.h — silk_LPC_analysis_filter (silk/LPC_analysis_filter.c:49-108) as a wave-per-signal kernel body.
 *
 * out[n] = sat16(round((in[n] << 12) - sum_{j<d} in[n-1-j] * B[j], 12)) for n >= d, out[0..d) = 0; the sum wraps modulo 2^32
 * exactly like the reference's SMLABB_ovflw chain (a wrapped sum does not depend on the order of its terms).
 * It is a d-tap FIR, so the outputs are independent: the signal (<= OA_LPC_MAX_LEN int16) and the d Q12 coefficients are staged in
 * LDS once, lane l computes outputs l, l+64, ... from d+1 LDS reads each, and the result goes back with coalesced 128-byte rows.
 * HBM traffic per signal is the algorithmic 2*len (in) + 2*len (out) + 2*d bytes: the kernel is HBM-bound. */
#ifndef OPUS_AMD_SILK_LPC_H
#define OPUS_AMD_SILK_LPC_H
#define OA_LPC_MAX_LEN 1024
#define OA_LPC_MAX_ORDER 16

struct LpcLds { i16 in[OA_LPC_MAX_LEN + 8]; i16 B[OA_LPC_MAX_ORDER]; };

WV_DEV void silk_lpc_analysis_filter_wave(WV_LDS LpcLds *L, i16 *out, const i16 *in, const i16 *B, int len, int d)
{
   const int lane = wv_lane();
   for (int i = lane; i < len; i += WV_WIDTH) L->in[i] = in[i];
   if (lane < OA_LPC_MAX_ORDER) L->B[lane] = lane < d ? B[lane] : 0;
   wv_sync();
   for (int n = lane; n < len; n += WV_WIDTH) {
      i32 o = 0;
      if (n >= d) {
         i32 pred = 0;
         for (int j = 0; j < OA_LPC_MAX_ORDER; j++) if (j < d) pred = add32(pred, (i32)L->in[n - 1 - j] * (i32)L->B[j]);
         i32 e = sub32(shl32((i32)L->in[n], 12), pred);
         e = ((e >> 11) + 1) >> 1;                                   /* silk_RSHIFT_ROUND(e, 12) */
         o = e > 32767 ? 32767 : e < -32768 ? -32768 : e;
      }
      out[n] = (i16)o;
   }
}
#endif

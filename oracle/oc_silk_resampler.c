/* oracle/oc_silk_resampler.c — TEST INFRASTRUCTURE ONLY.  Plain-C restatement of the SILK fixed-ratio int16 resamplers:
 * silk_resampler_init / silk_resampler (silk/resampler.c:79-224), the AR2 + polyphase-FIR downsampler
 * (silk/resampler_private_down_FIR.c:36-194, silk/resampler_private_AR2.c:36-54), the allpass 2x upsampler
 * (silk/resampler_private_up2_HQ.c:38-113) and the 2x + fractional-FIR upsampler (silk/resampler_private_IIR_FIR.c:36-107).
 * Checked sample-for-sample and state-word-for-state-word against the compiled reference by tests/test_oracle_silk.py. */
#include "oc_silk.h"
#include "oc_silk_tables.h"

static const s8 kDelayEnc[6][3] = { { 6, 0, 3 }, { 0, 7, 3 }, { 0, 1, 10 }, { 0, 2, 6 }, { 18, 10, 12 }, { 0, 0, 44 } };   /* resampler.c:52-60 */
static const s8 kDelayDec[3][6] = { { 4, 0, 2, 0, 0, 0 }, { 0, 9, 4, 7, 4, 4 }, { 0, 3, 12, 7, 7, 7 } };                 /* resampler.c:62-67 */
static int rate_id(s32 R) { int v = ((((R >> 12) - (R > 16000)) >> (R > 24000)) - 1); return v < 5 ? v : 5; }              /* resampler.c:70 */

static const s16 *coef_set(int id)
{
   switch (id) {
   case OC_RS_3_4: return ocs_resampler_3_4_coefs;  case OC_RS_2_3: return ocs_resampler_2_3_coefs;  case OC_RS_1_2: return ocs_resampler_1_2_coefs;
   case OC_RS_1_3: return ocs_resampler_1_3_coefs;  case OC_RS_1_4: return ocs_resampler_1_4_coefs;  case OC_RS_1_6: return ocs_resampler_1_6_coefs;
   }
   return 0;
}

int oc_silk_resampler_init(OcSilkResampler *S, s32 Fs_in, s32 Fs_out, int forEnc)
{
   memset(S, 0, sizeof *S);
   int in_ok = Fs_in == 8000 || Fs_in == 12000 || Fs_in == 16000, out_ok = Fs_out == 8000 || Fs_out == 12000 || Fs_out == 16000;
   if (forEnc) {
      if (!(in_ok || Fs_in == 24000 || Fs_in == 48000) || !out_ok) return -1;
      S->inputDelay = kDelayEnc[rate_id(Fs_in)][rate_id(Fs_out)];
   } else {
      if (!in_ok || !(out_ok || Fs_out == 24000 || Fs_out == 48000)) return -1;
      S->inputDelay = kDelayDec[rate_id(Fs_in)][rate_id(Fs_out)];
   }
   S->Fs_in_kHz = Fs_in / 1000;  S->Fs_out_kHz = Fs_out / 1000;
   S->batchSize = S->Fs_in_kHz * 10;
   int up2x = 0;
   if (Fs_out > Fs_in) {
      if (Fs_out == 2 * Fs_in) S->resampler_function = OC_RS_FN_UP2;
      else { S->resampler_function = OC_RS_FN_IIR_FIR; up2x = 1; }
   } else if (Fs_out < Fs_in) {
      S->resampler_function = OC_RS_FN_DOWN_FIR;
      if (Fs_out * 4 == Fs_in * 3)      { S->FIR_Fracs = 3; S->FIR_Order = 18; S->coefs_id = OC_RS_3_4; }
      else if (Fs_out * 3 == Fs_in * 2) { S->FIR_Fracs = 2; S->FIR_Order = 18; S->coefs_id = OC_RS_2_3; }
      else if (Fs_out * 2 == Fs_in)     { S->FIR_Fracs = 1; S->FIR_Order = 24; S->coefs_id = OC_RS_1_2; }
      else if (Fs_out * 3 == Fs_in)     { S->FIR_Fracs = 1; S->FIR_Order = 36; S->coefs_id = OC_RS_1_3; }
      else if (Fs_out * 4 == Fs_in)     { S->FIR_Fracs = 1; S->FIR_Order = 36; S->coefs_id = OC_RS_1_4; }
      else if (Fs_out * 6 == Fs_in)     { S->FIR_Fracs = 1; S->FIR_Order = 36; S->coefs_id = OC_RS_1_6; }
      else return -1;
   } else S->resampler_function = OC_RS_FN_COPY;
   S->invRatio_Q16 = ((Fs_in << (14 + up2x)) / Fs_out) << 2;
   while (q_mulww(S->invRatio_Q16, Fs_out) < (Fs_in << up2x)) S->invRatio_Q16++;
   return 0;
}

/* resampler_private_up2_HQ.c:38: two 3-section allpass chains (even / odd phase), Q10 state */
static void up2_hq(s32 *S, s16 *out, const s16 *in, s32 len)
{
   for (s32 k = 0; k < len; k++) {
      const s32 in32 = (s32)in[k] << 10;
      for (int ph = 0; ph < 2; ph++) {
         const s16 *c = ph ? ocs_resampler_up2_hq_1 : ocs_resampler_up2_hq_0;
         s32 *st = S + 3 * ph;
         s32 Y = in32 - st[0], X = q_mulwb(Y, c[0]);
         s32 o1 = st[0] + X;  st[0] = in32 + X;
         Y = o1 - st[1];  X = q_mulwb(Y, c[1]);
         s32 o2 = st[1] + X;  st[1] = o1 + X;
         Y = o2 - st[2];  X = q_mlawb(Y, Y, c[2]);
         o1 = st[2] + X;  st[2] = o2 + X;
         out[2 * k + ph] = (s16)q_sat16(q_rshift_round(o1, 10));
      }
   }
}

/* resampler_private_down_FIR.c:145 */
static void down_fir(OcSilkResampler *S, s16 *out, const s16 *in, s32 inLen)
{
   s32 buf[480 + 36];
   const s16 *C = coef_set(S->coefs_id), *F = C + 2;
   const int ord = S->FIR_Order;
   memcpy(buf, S->sFIR.i32, (size_t)ord * sizeof(s32));
   s32 nIn;
   for (;;) {
      nIn = inLen < S->batchSize ? inLen : S->batchSize;
      for (s32 k = 0; k < nIn; k++) {                                   /* resampler_private_AR2.c:36 */
         s32 o = S->sIIR[0] + ((s32)in[k] << 8);
         buf[ord + k] = o;
         o = q_shlw(o, 2);
         S->sIIR[0] = q_mlawb(S->sIIR[1], o, C[0]);
         S->sIIR[1] = q_mulwb(o, C[1]);
      }
      const s32 max_index_Q16 = nIn << 16;
      for (s32 idx = 0; idx < max_index_Q16; idx += S->invRatio_Q16) {
         const s32 *b = buf + (idx >> 16);
         s32 r;
         if (ord == 18) {                                               /* fractional phases, mirrored second half */
            const int ph = q_mulwb(idx & 0xFFFF, S->FIR_Fracs);
            const s16 *c0 = &F[9 * ph], *c1 = &F[9 * (S->FIR_Fracs - 1 - ph)];
            r = q_mulwb(b[0], c0[0]);
            for (int j = 1; j < 9; j++) r = q_mlawb(r, b[j], c0[j]);
            for (int j = 0; j < 9; j++) r = q_mlawb(r, b[17 - j], c1[j]);
         } else {                                                       /* symmetric FIR */
            r = q_mulwb(b[0] + b[ord - 1], F[0]);
            for (int j = 1; j < ord / 2; j++) r = q_mlawb(r, b[j] + b[ord - 1 - j], F[j]);
         }
         *out++ = (s16)q_sat16(q_rshift_round(r, 6));
      }
      in += nIn; inLen -= nIn;
      if (inLen > 1) memcpy(buf, &buf[nIn], (size_t)ord * sizeof(s32));
      else break;
   }
   memcpy(S->sFIR.i32, &buf[nIn], (size_t)ord * sizeof(s32));
}

/* resampler_private_IIR_FIR.c:65 */
static void iir_fir(OcSilkResampler *S, s16 *out, const s16 *in, s32 inLen)
{
   s16 buf[2 * 160 + 8];
   memcpy(buf, S->sFIR.i16, 8 * sizeof(s16));
   s32 nIn;
   for (;;) {
      nIn = inLen < S->batchSize ? inLen : S->batchSize;
      up2_hq(S->sIIR, &buf[8], in, nIn);
      const s32 max_index_Q16 = nIn << 17;
      for (s32 idx = 0; idx < max_index_Q16; idx += S->invRatio_Q16) {
         const int ti = q_mulwb(idx & 0xFFFF, 12);
         const s16 *b = &buf[idx >> 16], *t0 = &ocs_resampler_frac_fir_12[4 * ti], *t1 = &ocs_resampler_frac_fir_12[4 * (11 - ti)];
         s32 r = q_mulbb(b[0], t0[0]);
         r = q_mlabb(r, b[1], t0[1]); r = q_mlabb(r, b[2], t0[2]); r = q_mlabb(r, b[3], t0[3]);
         r = q_mlabb(r, b[4], t1[3]); r = q_mlabb(r, b[5], t1[2]); r = q_mlabb(r, b[6], t1[1]); r = q_mlabb(r, b[7], t1[0]);
         *out++ = (s16)q_sat16(q_rshift_round(r, 15));
      }
      in += nIn; inLen -= nIn;
      if (inLen > 0) memcpy(buf, &buf[nIn << 1], 8 * sizeof(s16));
      else break;
   }
   memcpy(S->sFIR.i16, &buf[nIn << 1], 8 * sizeof(s16));
}

static void run_fn(OcSilkResampler *S, s16 *out, const s16 *in, s32 len)
{
   switch (S->resampler_function) {
   case OC_RS_FN_UP2:      up2_hq(S->sIIR, out, in, len); break;
   case OC_RS_FN_IIR_FIR:  iir_fir(S, out, in, len); break;
   case OC_RS_FN_DOWN_FIR: down_fir(S, out, in, len); break;
   default:                memcpy(out, in, (size_t)len * sizeof(s16));
   }
}

/* resampler.c:183: the first millisecond comes from the delay buffer (delay-compensation between rate pairs) */
int oc_silk_resampler(OcSilkResampler *S, s16 *out, const s16 *in, s32 inLen)
{
   const int nSamples = S->Fs_in_kHz - S->inputDelay;
   memcpy(&S->delayBuf[S->inputDelay], in, (size_t)nSamples * sizeof(s16));
   run_fn(S, out, S->delayBuf, S->Fs_in_kHz);
   run_fn(S, &out[S->Fs_out_kHz], &in[nSamples], inLen - S->Fs_in_kHz);
   memcpy(S->delayBuf, &in[inLen - S->inputDelay], (size_t)S->inputDelay * sizeof(s16));
   return 0;
}

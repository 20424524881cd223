/* silk_dec.h — the SILK decoder (lane-0 serial code of the stream's wave), rows a22/a34 of SURVEY §8.
 *
 * What it computes, with the reference function each block follows:
 *   sd_decode_indices     silk_decode_indices       silk/decode_indices.c:35      side information (gains, NLSF path, pitch, LTP, seed)
 *   sd_decode_pulses      silk_decode_pulses        silk/decode_pulses.c:37       shell-coded excitation (silk/shell_coder.c:118, code_signs.c:74)
 *   sd_decode_parameters  silk_decode_parameters    silk/decode_parameters.c:35   gains (gain_quant.c:100, log2lin.c:36), NLSF (NLSF_decode.c:62,
 *                                                                                 NLSF_unpack.c:35, NLSF_stabilize.c:50) -> LPC (NLSF2A.c:66, LPC_fit.c:35,
 *                                                                                 LPC_inv_pred_gain.c:45/:122, bwexpander_32.c:35), pitch lags (decode_pitch.c:38)
 *   sd_decode_core        silk_decode_core          silk/decode_core.c:38         excitation -> LTP synthesis -> LPC synthesis
 *   sd_decode_frame       silk_decode_frame         silk/decode_frame.c:43
 *   sd_set_fs / sd_reset  silk_decoder_set_fs / silk_reset_decoder   silk/decoder_set_fs.c:35, silk/init_decoder.c:43
 *   sd_stereo_*           silk_stereo_decode_pred / _mid_only / silk_stereo_MS_to_LR   silk/stereo_decode_pred.c:35,:66, silk/stereo_MS_to_LR.c:35
 *   silk_decode_wave      silk_Decode               silk/dec_API.c:142            per-packet frame/channel sequencing, LBRR skipping, resampling to the API rate (silk_dec_api.h)
 * Entropy and parameter decoding are serial chains per stream and run as lane-0 sections of the wave that owns the stream; the synthesis filters
 * (sd_decode_core_wave) and the resampler use the whole wave.  Same kernel as the CELT decoder, whose LDS regions are borrowed between CELT frames.  Packet loss concealment / comfort noise
 * (silk/PLC.c:77-493 silk_PLC / _update / _conceal / _glue_frames, silk/CNG.c:79 silk_CNG) are sd_plc*, sd_cng. */
#ifndef OPUS_AMD_SILK_DEC_H
#define OPUS_AMD_SILK_DEC_H
#include "silk_tables.h"

#define SD_CODE_INDEPENDENTLY 0
#define SD_CODE_INDEPENDENTLY_NO_LTP_SCALING 1
#define SD_CODE_CONDITIONALLY 2
#define SD_TYPE_NO_VOICE 0
#define SD_TYPE_VOICED 2
#define SD_FLAG_DECODE_NORMAL 0
#define SD_FLAG_PACKET_LOST 1
#define SD_FLAG_DECODE_LBRR 2

struct SdCtrl { i32 pitchL[4]; i32 Gains_Q16[4]; i16 PredCoef_Q12[2][16]; i16 LTPCoef_Q14[20]; i32 LTP_scale_Q14; };
struct SdNlsfCb { int nVectors, order, qstep; const u8 *cb1_nlsf, *cb1_icdf, *pred, *ec_sel, *ec_icdf; const i16 *wght, *deltamin; };
WV_DEV SdNlsfCb sd_nlsf_cb(int LPC_order)
{
   SdNlsfCb c;
   if (LPC_order == 16) { c.nVectors = 32; c.order = 16; c.qstep = SK_NLSF_WB_QSTEP_Q16; c.cb1_nlsf = sk_nlsf_wb_cb1_nlsf_q8; c.cb1_icdf = sk_nlsf_wb_cb1_icdf; c.pred = sk_nlsf_wb_pred_q8;
                         c.ec_sel = sk_nlsf_wb_ec_sel; c.ec_icdf = sk_nlsf_wb_ec_icdf; c.wght = sk_nlsf_wb_cb1_wght_q9; c.deltamin = sk_nlsf_wb_deltamin_q15; }
   else { c.nVectors = 32; c.order = 10; c.qstep = SK_NLSF_NB_MB_QSTEP_Q16; c.cb1_nlsf = sk_nlsf_nb_mb_cb1_nlsf_q8; c.cb1_icdf = sk_nlsf_nb_mb_cb1_icdf; c.pred = sk_nlsf_nb_mb_pred_q8;
          c.ec_sel = sk_nlsf_nb_mb_ec_sel; c.ec_icdf = sk_nlsf_nb_mb_ec_icdf; c.wght = sk_nlsf_nb_mb_cb1_wght_q9; c.deltamin = sk_nlsf_nb_mb_deltamin_q15; }
   return c;
}
WV_DEV void sd_nlsf_unpack(i32 *ec_ix, i32 *pred_Q8, const SdNlsfCb &cb, int CB1_index)            /* NLSF_unpack.c:35 */
{
   const u8 *sel = &cb.ec_sel[CB1_index * cb.order / 2];
   for (int i = 0; i < cb.order; i += 2) {
      const int entry = *sel++;
      ec_ix[i] = ((entry >> 1) & 7) * 9;
      pred_Q8[i] = cb.pred[i + (entry & 1) * (cb.order - 1)];
      ec_ix[i + 1] = ((entry >> 5) & 7) * 9;
      pred_Q8[i + 1] = cb.pred[i + ((entry >> 4) & 1) * (cb.order - 1) + 1];
   }
}

WV_DEV const u8 *sd_pitch_contour_icdf(int fs_kHz, int nb_subfr)
{ return fs_kHz == 8 ? (nb_subfr == 4 ? sk_pitch_contour_nb_icdf : sk_pitch_contour_10ms_nb_icdf) : (nb_subfr == 4 ? sk_pitch_contour_icdf : sk_pitch_contour_10ms_icdf); }
WV_DEV const u8 *sd_pitch_low_bits_icdf(int fs_kHz) { return fs_kHz == 16 ? sk_uniform8_icdf : fs_kHz == 12 ? sk_uniform6_icdf : sk_uniform4_icdf; }

template <class ECB, class CH> WV_DEV void sd_decode_indices(EC_ARGS_G, CH ch, int FrameIndex, int decode_LBRR, int condCoding)
{
   auto ix = &ch->indices;
   int Ix;
   if (decode_LBRR || ch->VAD_flags[FrameIndex]) Ix = k_ec_dec_icdf(EC_PASS, sk_type_offset_vad_icdf, 8) + 2;
   else Ix = k_ec_dec_icdf(EC_PASS, sk_type_offset_no_vad_icdf, 8);
   ix->signalType = (i8)(Ix >> 1); ix->quantOffsetType = (i8)(Ix & 1);
   if (condCoding == SD_CODE_CONDITIONALLY) ix->GainsIndices[0] = (i8)k_ec_dec_icdf(EC_PASS, sk_delta_gain_icdf, 8);
   else { ix->GainsIndices[0] = (i8)(k_ec_dec_icdf(EC_PASS, &sk_gain_icdf[ix->signalType * 8], 8) << 3); ix->GainsIndices[0] += (i8)k_ec_dec_icdf(EC_PASS, sk_uniform8_icdf, 8); }
   for (int i = 1; i < ch->nb_subfr; i++) ix->GainsIndices[i] = (i8)k_ec_dec_icdf(EC_PASS, sk_delta_gain_icdf, 8);
   const SdNlsfCb cb = sd_nlsf_cb(ch->LPC_order);
   ix->NLSFIndices[0] = (i8)k_ec_dec_icdf(EC_PASS, &cb.cb1_icdf[(ix->signalType >> 1) * cb.nVectors], 8);
   i32 ec_ix[16], pred_Q8[16];
   sd_nlsf_unpack(ec_ix, pred_Q8, cb, ix->NLSFIndices[0]);
   for (int i = 0; i < cb.order; i++) {
      Ix = k_ec_dec_icdf(EC_PASS, &cb.ec_icdf[ec_ix[i]], 8);
      if (Ix == 0) Ix -= k_ec_dec_icdf(EC_PASS, sk_nlsf_ext_icdf, 8);
      else if (Ix == 8) Ix += k_ec_dec_icdf(EC_PASS, sk_nlsf_ext_icdf, 8);
      ix->NLSFIndices[i + 1] = (i8)(Ix - 4);
   }
   ix->NLSFInterpCoef_Q2 = ch->nb_subfr == 4 ? (i8)k_ec_dec_icdf(EC_PASS, sk_nlsf_interpolation_factor_icdf, 8) : (i8)4;
   if (ix->signalType == SD_TYPE_VOICED) {
      int absolute = 1;
      if (condCoding == SD_CODE_CONDITIONALLY && ch->ec_prevSignalType == SD_TYPE_VOICED) {
         int delta = (i16)k_ec_dec_icdf(EC_PASS, sk_pitch_delta_icdf, 8);
         if (delta > 0) { ix->lagIndex = (i16)(ch->ec_prevLagIndex + delta - 9); absolute = 0; }
      }
      if (absolute) {
         ix->lagIndex = (i16)((i16)k_ec_dec_icdf(EC_PASS, sk_pitch_lag_icdf, 8) * (ch->fs_kHz >> 1));
         ix->lagIndex += (i16)k_ec_dec_icdf(EC_PASS, sd_pitch_low_bits_icdf(ch->fs_kHz), 8);
      }
      ch->ec_prevLagIndex = ix->lagIndex;
      ix->contourIndex = (i8)k_ec_dec_icdf(EC_PASS, sd_pitch_contour_icdf(ch->fs_kHz, ch->nb_subfr), 8);
      ix->PERIndex = (i8)k_ec_dec_icdf(EC_PASS, sk_ltp_per_index_icdf, 8);
      const u8 *gicdf = &sk_ltp_gain_icdf[ix->PERIndex == 0 ? 0 : ix->PERIndex == 1 ? 8 : 24];
      for (int k = 0; k < ch->nb_subfr; k++) ix->LTPIndex[k] = (i8)k_ec_dec_icdf(EC_PASS, gicdf, 8);
      ix->LTP_scaleIndex = condCoding == SD_CODE_INDEPENDENTLY ? (i8)k_ec_dec_icdf(EC_PASS, sk_ltpscale_icdf, 8) : (i8)0;
   }
   ch->ec_prevSignalType = ix->signalType;
   ix->Seed = (i8)k_ec_dec_icdf(EC_PASS, sk_uniform4_icdf, 8);
}

template <class P1, class ECB> WV_DEV void sd_split(P1 c1, P1 c2, EC_ARGS_G, int p, const u8 *table)                 /* shell_coder.c:63 */
{
   if (p > 0) { c1[0] = (i16)k_ec_dec_icdf(EC_PASS, &table[sk_shell_code_table_offsets[p]], 8); c2[0] = (i16)(p - c1[0]); }
   else { c1[0] = 0; c2[0] = 0; }
}
template <class PU, class ECB, class PT> WV_DEV void sd_shell_decoder(PU p0, EC_ARGS_G, int pulses4, PT tmp)                   /* shell_coder.c:118; tmp: 14 words */
{
   PT p3 = tmp, p2 = tmp + 2, p1 = tmp + 6;
   sd_split(p3 + 0, p3 + 1, EC_PASS, pulses4, sk_shell_code_table3);
   sd_split(p2 + 0, p2 + 1, EC_PASS, p3[0], sk_shell_code_table2);
   sd_split(p1 + 0, p1 + 1, EC_PASS, p2[0], sk_shell_code_table1);
   sd_split(p0 + 0, p0 + 1, EC_PASS, p1[0], sk_shell_code_table0);
   sd_split(p0 + 2, p0 + 3, EC_PASS, p1[1], sk_shell_code_table0);
   sd_split(p1 + 2, p1 + 3, EC_PASS, p2[1], sk_shell_code_table1);
   sd_split(p0 + 4, p0 + 5, EC_PASS, p1[2], sk_shell_code_table0);
   sd_split(p0 + 6, p0 + 7, EC_PASS, p1[3], sk_shell_code_table0);
   sd_split(p2 + 2, p2 + 3, EC_PASS, p3[1], sk_shell_code_table2);
   sd_split(p1 + 4, p1 + 5, EC_PASS, p2[2], sk_shell_code_table1);
   sd_split(p0 + 8, p0 + 9, EC_PASS, p1[4], sk_shell_code_table0);
   sd_split(p0 + 10, p0 + 11, EC_PASS, p1[5], sk_shell_code_table0);
   sd_split(p1 + 6, p1 + 7, EC_PASS, p2[3], sk_shell_code_table1);
   sd_split(p0 + 12, p0 + 13, EC_PASS, p1[6], sk_shell_code_table0);
   sd_split(p0 + 14, p0 + 15, EC_PASS, p1[7], sk_shell_code_table0);
}
template <class ECB, class PU, class PT> WV_DEV void sd_decode_pulses(EC_ARGS_G, PU pulses, int signalType, int quantOffsetType, int frame_length, PT tmp)
{
   i32 sum_pulses[20], nLshifts[20];
   const int RateLevelIndex = k_ec_dec_icdf(EC_PASS, &sk_rate_levels_icdf[(signalType >> 1) * 9], 8);
   int iter = frame_length >> 4;
   if (iter * 16 < frame_length) iter++;                                                              /* 10 ms at 12 kHz */
   const u8 *cdf = &sk_pulses_per_block_icdf[RateLevelIndex * 18];
   for (int i = 0; i < iter; i++) {
      nLshifts[i] = 0;
      sum_pulses[i] = k_ec_dec_icdf(EC_PASS, cdf, 8);
      while (sum_pulses[i] == 17) { nLshifts[i]++; sum_pulses[i] = k_ec_dec_icdf(EC_PASS, &sk_pulses_per_block_icdf[9 * 18] + (nLshifts[i] == 10), 8); }
   }
   for (int i = 0; i < iter; i++) {
      if (sum_pulses[i] > 0) sd_shell_decoder(pulses + i * 16, EC_PASS, sum_pulses[i], tmp);
      else for (int k = 0; k < 16; k++) pulses[i * 16 + k] = 0;
   }
   for (int i = 0; i < iter; i++) {
      if (nLshifts[i] > 0) {
         const int nLS = nLshifts[i];
         for (int k = 0; k < 16; k++) {
            int abs_q = pulses[i * 16 + k];
            for (int j = 0; j < nLS; j++) abs_q = (abs_q << 1) + k_ec_dec_icdf(EC_PASS, sk_lsb_icdf, 8);
            pulses[i * 16 + k] = (i16)abs_q;
         }
         sum_pulses[i] |= nLS << 5;
      }
   }
   /* signs (code_signs.c:74) */
   const u8 *icdf_ptr = &sk_sign_icdf[7 * (quantOffsetType + (signalType << 1))];
   const int nblk = (frame_length + 8) >> 4;
   for (int i = 0; i < nblk; i++) {
      const int p = sum_pulses[i];
      if (p > 0) {
         u8 icdf[2]; icdf[0] = icdf_ptr[imin(p & 0x1F, 6)]; icdf[1] = 0;
         for (int j = 0; j < 16; j++) if (pulses[i * 16 + j] > 0) pulses[i * 16 + j] = (i16)(pulses[i * 16 + j] * ((k_ec_dec_icdf(EC_PASS, icdf, 8) << 1) - 1));
      }
   }
}

WV_DEV i32 sd_log2lin(i32 inLog_Q7)                                                                    /* log2lin.c:36 */
{
   if (inLog_Q7 < 0) return 0;
   if (inLog_Q7 >= 3967) return 2147483647;
   i32 out = (i32)1 << (inLog_Q7 >> 7);
   const i32 frac = inLog_Q7 & 0x7F, t = sk_mlawb(frac, sk_mulbb(frac, 128 - frac), -174);
   if (inLog_Q7 < 2048) out = out + ((out * t) >> 7); else out = out + (out >> 7) * t;
   return out;
}
template <class PA> WV_DEV void sd_bwexpander_32(PA ar, int d, i32 chirp_Q16)                                           /* bwexpander_32.c:35 */
{
   const i32 cm1 = chirp_Q16 - 65536;
   for (int i = 0; i < d - 1; i++) { ar[i] = sk_mulww(chirp_Q16, ar[i]); chirp_Q16 += sk_rround(chirp_Q16 * cm1, 16); }
   ar[d - 1] = sk_mulww(chirp_Q16, ar[d - 1]);
}
template <class PA> WV_DEV void sd_bwexpander(PA ar, int d, i32 chirp_Q16)                                              /* bwexpander.c:35 */
{
   const i32 cm1 = chirp_Q16 - 65536;
   for (int i = 0; i < d - 1; i++) { ar[i] = (i16)sk_rround(chirp_Q16 * ar[i], 16); chirp_Q16 += sk_rround(chirp_Q16 * cm1, 16); }
   ar[d - 1] = (i16)sk_rround(chirp_Q16 * ar[d - 1], 16);
}
WV_DEV i32 sd_rround64(i64 a, int s) { return (i32)(s == 1 ? (a >> 1) + (a & 1) : ((a >> (s - 1)) + 1) >> 1); }
WV_DEV i64 sd_rround64w(i64 a, int s) { return s == 1 ? (a >> 1) + (a & 1) : ((a >> (s - 1)) + 1) >> 1; }
/* LPC_inv_pred_gain.c:45 (QA = 24): 0 = unstable, else inverse prediction gain Q30 */
/* A: 16 words of working storage (LDS for lane-0 callers that care about latency: a run-time indexed private array lives in scratch memory) */
template <class PA, class PW> WV_DEV i32 sd_lpc_inverse_pred_gain_w(PA A_Q12, int order, PW A)
{
   i32 DC = 0;
   for (int k = 0; k < order; k++) { DC += A_Q12[k]; A[k] = shl32(A_Q12[k], 12); }
   if (DC >= 4096) return 0;
   const i32 A_LIMIT = 16773022;                                                                      /* SILK_FIX_CONST(0.99975, 24) */
   const i32 MIN_INVGAIN = 107374;                                                                    /* SILK_FIX_CONST(1/1e4, 30) */
   i32 invGain_Q30 = (i32)1 << 30;
   int k;
   for (k = order - 1; k > 0; k--) {
      if (A[k] > A_LIMIT || A[k] < -A_LIMIT) return 0;
      const i32 rc_Q31 = -shl32(A[k], 31 - 24);
      const i32 rc_mult1_Q30 = ((i32)1 << 30) - sk_mulhi(rc_Q31, rc_Q31);
      invGain_Q30 = shl32(sk_mulhi(invGain_Q30, rc_mult1_Q30), 2);
      if (invGain_Q30 < MIN_INVGAIN) return 0;
      const int mult2Q = 32 - sk_clz(rc_mult1_Q30 > 0 ? rc_mult1_Q30 : -rc_mult1_Q30);
      const i32 rc_mult2 = sk_inverse32_varQ(rc_mult1_Q30, mult2Q + 30);
      for (int n = 0; n < (k + 1) >> 1; n++) {
         const i32 tmp1 = A[n], tmp2 = A[k - n - 1];
         i64 t64 = sd_rround64w((i64)sk_sub_sat(tmp1, sd_rround64((i64)tmp2 * rc_Q31, 31)) * rc_mult2, mult2Q);
         if (t64 > 2147483647LL || t64 < -2147483648LL) return 0;
         A[n] = (i32)t64;
         t64 = sd_rround64w((i64)sk_sub_sat(tmp2, sd_rround64((i64)tmp1 * rc_Q31, 31)) * rc_mult2, mult2Q);
         if (t64 > 2147483647LL || t64 < -2147483648LL) return 0;
         A[k - n - 1] = (i32)t64;
      }
   }
   if (A[k] > A_LIMIT || A[k] < -A_LIMIT) return 0;
   const i32 rc_Q31 = -shl32(A[0], 31 - 24);
   const i32 rc_mult1_Q30 = ((i32)1 << 30) - sk_mulhi(rc_Q31, rc_Q31);
   invGain_Q30 = shl32(sk_mulhi(invGain_Q30, rc_mult1_Q30), 2);
   if (invGain_Q30 < MIN_INVGAIN) return 0;
   return invGain_Q30;
}
template <class PA> WV_DEV i32 sd_lpc_inverse_pred_gain(PA A_Q12, int order) { i32 A[16]; return sd_lpc_inverse_pred_gain_w(A_Q12, order, A); }
/* NLSF2A.c:66: NLSF (Q15) -> monic LPC coefficients Q12 via the two symmetric polynomials, then range fit and stability loop */
template <class PO, class PC> WV_DEV void sd_nlsf2a_poly(PO out, PC cLSF, int dd)
{
   out[0] = (i32)1 << 16; out[1] = -cLSF[0];
   for (int k = 1; k < dd; k++) {
      const i32 ftmp = cLSF[2 * k];
      out[k + 1] = shl32(out[k - 1], 1) - sd_rround64((i64)ftmp * out[k], 16);
      for (int n = k; n > 1; n--) out[n] += out[n - 2] - sd_rround64((i64)ftmp * out[n - 1], 16);
      out[1] -= ftmp;
   }
}
WV_TABLE u8 k_sd_ordering16[16] = { 0, 15, 8, 7, 4, 11, 12, 3, 2, 13, 10, 5, 6, 9, 14, 1 };
WV_TABLE u8 k_sd_ordering10[10] = { 0, 9, 6, 3, 4, 5, 8, 1, 2, 7 };
/* wk: 66 words of working storage (cosq 16, P 9, Q 9, a32 16, inverse-gain work 16) */
template <class PA, class PN, class PW> WV_DEV void sd_nlsf2a_w(PA a_Q12, PN NLSF, int d, PW wk)
{
   const u8 *ordering = d == 16 ? k_sd_ordering16 : k_sd_ordering10;
   PW cosq = wk, P = wk + 16, Q = wk + 25, a32 = wk + 34, ipg = wk + 50;
   for (int k = 0; k < d; k++) {
      const i32 f_int = NLSF[k] >> 8, f_frac = NLSF[k] - (f_int << 8);
      const i32 cos_val = sk_lsf_cos_tab_q12[f_int], delta = sk_lsf_cos_tab_q12[f_int + 1] - cos_val;
      cosq[ordering[k]] = sk_rround(shl32(cos_val, 8) + delta * f_frac, 4);
   }
   const int dd = d >> 1;
   sd_nlsf2a_poly(P, &cosq[0], dd);
   sd_nlsf2a_poly(Q, &cosq[1], dd);
   for (int k = 0; k < dd; k++) { const i32 Pt = P[k + 1] + P[k], Qt = Q[k + 1] - Q[k]; a32[k] = -Qt - Pt; a32[d - k - 1] = Qt - Pt; }
   /* silk_LPC_fit(a_Q12, a32, 12, 17, d)  (LPC_fit.c:35) */
   {
      int i, idx = 0;
      for (i = 0; i < 10; i++) {
         i32 maxabs = 0;
         for (int k = 0; k < d; k++) { const i32 av = a32[k] > 0 ? a32[k] : -a32[k]; if (av > maxabs) { maxabs = av; idx = k; } }
         maxabs = sk_rround(maxabs, 5);
         if (maxabs > 32767) {
            maxabs = imin(maxabs, 163838);
            const i32 chirp_Q16 = 65470 - shl32(maxabs - 32767, 14) / ((maxabs * (idx + 1)) >> 2);           /* SILK_FIX_CONST(0.999, 16) */
            sd_bwexpander_32(a32, d, chirp_Q16);
         } else break;
      }
      if (i == 10) for (int k = 0; k < d; k++) { a_Q12[k] = (i16)sk_sat16(sk_rround(a32[k], 5)); a32[k] = shl32(a_Q12[k], 5); }
      else for (int k = 0; k < d; k++) a_Q12[k] = (i16)sk_rround(a32[k], 5);
   }
   for (int i = 0; sd_lpc_inverse_pred_gain_w(a_Q12, d, ipg) == 0 && i < 16; i++) {                         /* MAX_LPC_STABILIZE_ITERATIONS */
      sd_bwexpander_32(a32, d, 65536 - shl32(2, i));
      for (int k = 0; k < d; k++) a_Q12[k] = (i16)sk_rround(a32[k], 5);
   }
}
template <class PA, class PN> WV_DEV void sd_nlsf2a(PA a_Q12, PN NLSF, int d) { i32 wk[66]; sd_nlsf2a_w(a_Q12, NLSF, d, wk); }
WV_DEV void sd_nlsf_stabilize(i16 *NLSF, const i16 *NDeltaMin, int L)                                       /* NLSF_stabilize.c:50 */
{
   for (int loops = 0; loops < 20; loops++) {
      i32 min_diff = NLSF[0] - NDeltaMin[0]; int I = 0;
      for (int i = 1; i <= L - 1; i++) { const i32 diff = NLSF[i] - (NLSF[i - 1] + NDeltaMin[i]); if (diff < min_diff) { min_diff = diff; I = i; } }
      { const i32 diff = (1 << 15) - (NLSF[L - 1] + NDeltaMin[L]); if (diff < min_diff) { min_diff = diff; I = L; } }
      if (min_diff >= 0) return;
      if (I == 0) NLSF[0] = NDeltaMin[0];
      else if (I == L) NLSF[L - 1] = (i16)((1 << 15) - NDeltaMin[L]);
      else {
         i32 min_center = 0, max_center = 1 << 15;
         for (int k = 0; k < I; k++) min_center += NDeltaMin[k];
         min_center += NDeltaMin[I] >> 1;
         for (int k = L; k > I; k--) max_center -= NDeltaMin[k];
         max_center -= NDeltaMin[I] >> 1;
         i32 c = sk_rround((i32)NLSF[I - 1] + (i32)NLSF[I], 1);
         /* silk_LIMIT_32 tolerates min > max */
         if (min_center > max_center) c = c > min_center ? min_center : c < max_center ? max_center : c;
         else c = c > max_center ? max_center : c < min_center ? min_center : c;
         NLSF[I - 1] = (i16)((i16)c - (NDeltaMin[I] >> 1));
         NLSF[I] = (i16)(NLSF[I - 1] + NDeltaMin[I]);
      }
   }
   /* fall-back: sort and clamp (NLSF_stabilize.c:120-141) */
   for (int i = 1; i < L; i++) { const i16 v = NLSF[i]; int j = i - 1; for (; j >= 0 && v < NLSF[j]; j--) NLSF[j + 1] = NLSF[j]; NLSF[j + 1] = v; }
   NLSF[0] = (i16)imax(NLSF[0], NDeltaMin[0]);
   for (int i = 1; i < L; i++) NLSF[i] = (i16)imax(NLSF[i], sk_sat16((i32)NLSF[i - 1] + NDeltaMin[i]));
   NLSF[L - 1] = (i16)imin(NLSF[L - 1], (1 << 15) - NDeltaMin[L]);
   for (int i = L - 2; i >= 0; i--) NLSF[i] = (i16)imin(NLSF[i], NLSF[i + 1] - NDeltaMin[i + 1]);
}
template <class PI> WV_DEV void sd_nlsf_decode(i16 *pNLSF_Q15, PI NLSFIndices, const SdNlsfCb &cb)                      /* NLSF_decode.c:62 */
{
   i32 ec_ix[16], pred_Q8[16]; i32 res_Q10[16];
   sd_nlsf_unpack(ec_ix, pred_Q8, cb, NLSFIndices[0]);
   i32 out_Q10 = 0;
   for (int i = cb.order - 1; i >= 0; i--) {
      const i32 pred_Q10 = sk_mulbb(out_Q10, pred_Q8[i]) >> 8;
      out_Q10 = shl32(NLSFIndices[1 + i], 10);
      if (out_Q10 > 0) out_Q10 -= 102; else if (out_Q10 < 0) out_Q10 += 102;                               /* SILK_FIX_CONST(0.1, 10) */
      out_Q10 = sk_mlawb(pred_Q10, out_Q10, cb.qstep);
      res_Q10[i] = (i16)out_Q10;
      out_Q10 = (i16)out_Q10;
   }
   const u8 *el = &cb.cb1_nlsf[NLSFIndices[0] * cb.order]; const i16 *w = &cb.wght[NLSFIndices[0] * cb.order];
   for (int i = 0; i < cb.order; i++) {
      const i32 t = shl32(res_Q10[i], 14) / w[i] + shl32((i16)el[i], 7);
      pNLSF_Q15[i] = (i16)(t < 0 ? 0 : t > 32767 ? 32767 : t);
   }
   sd_nlsf_stabilize(pNLSF_Q15, cb.deltamin, cb.order);
}
template <class PL> WV_DEV void sd_decode_pitch(int lagIndex, int contourIndex, PL pitch_lags, int Fs_kHz, int nb_subfr)     /* decode_pitch.c:38 */
{
   const i8 *cb; int cbk_size;
   if (Fs_kHz == 8) { if (nb_subfr == 4) { cb = sk_cb_lags_stage2; cbk_size = 11; } else { cb = sk_cb_lags_stage2_10ms; cbk_size = 3; } }
   else { if (nb_subfr == 4) { cb = sk_cb_lags_stage3; cbk_size = 34; } else { cb = sk_cb_lags_stage3_10ms; cbk_size = 12; } }
   const int min_lag = 2 * Fs_kHz, max_lag = 18 * Fs_kHz, lag = min_lag + lagIndex;
   for (int k = 0; k < nb_subfr; k++) { int p = lag + cb[k * cbk_size + contourIndex]; pitch_lags[k] = p < min_lag ? min_lag : p > max_lag ? max_lag : p; }
}

template <class CH, class CT> WV_DEV void sd_decode_parameters(CH ch, CT c, int condCoding)                              /* decode_parameters.c:35 */
{
   auto ix = &ch->indices;
   /* gains (gain_quant.c:100): OFFSET = 2090, INV_SCALE_Q16 = 1907825 */
   for (int k = 0; k < ch->nb_subfr; k++) {
      int prev = ch->LastGainIndex;
      if (k == 0 && condCoding != SD_CODE_CONDITIONALLY) prev = imax(ix->GainsIndices[k], prev - 16);
      else {
         const int ind_tmp = ix->GainsIndices[k] - 4, thr = 2 * 36 - 64 + prev;
         if (ind_tmp > thr) prev += (ind_tmp << 1) - thr; else prev += ind_tmp;
      }
      prev = prev < 0 ? 0 : prev > 63 ? 63 : prev;
      ch->LastGainIndex = prev;
      c->Gains_Q16[k] = sd_log2lin(imin(sk_mulwb(1907825, prev) + 2090, 3967));
   }
   const SdNlsfCb cb = sd_nlsf_cb(ch->LPC_order);
   i16 pNLSF[16], pNLSF0[16];
   sd_nlsf_decode(pNLSF, ix->NLSFIndices, cb);
   sd_nlsf2a(c->PredCoef_Q12[1], pNLSF, ch->LPC_order);
   if (ch->first_frame_after_reset == 1) ix->NLSFInterpCoef_Q2 = 4;
   if (ix->NLSFInterpCoef_Q2 < 4) {
      for (int i = 0; i < ch->LPC_order; i++) pNLSF0[i] = (i16)(ch->prevNLSF_Q15[i] + ((ix->NLSFInterpCoef_Q2 * (pNLSF[i] - ch->prevNLSF_Q15[i])) >> 2));
      sd_nlsf2a(c->PredCoef_Q12[0], pNLSF0, ch->LPC_order);
   } else for (int i = 0; i < ch->LPC_order; i++) c->PredCoef_Q12[0][i] = c->PredCoef_Q12[1][i];
   for (int i = 0; i < ch->LPC_order; i++) ch->prevNLSF_Q15[i] = pNLSF[i];
   if (ch->lossCnt) { sd_bwexpander(c->PredCoef_Q12[0], ch->LPC_order, 63570); sd_bwexpander(c->PredCoef_Q12[1], ch->LPC_order, 63570); }   /* BWE_AFTER_LOSS_Q16 */
   if (ix->signalType == SD_TYPE_VOICED) {
      sd_decode_pitch(ix->lagIndex, ix->contourIndex, c->pitchL, ch->fs_kHz, ch->nb_subfr);
      const i8 *cbk = &sk_ltp_vq_q7[ix->PERIndex == 0 ? 0 : ix->PERIndex == 1 ? 40 : 120];
      for (int k = 0; k < ch->nb_subfr; k++) for (int i = 0; i < 5; i++) c->LTPCoef_Q14[k * 5 + i] = (i16)shl32(cbk[ix->LTPIndex[k] * 5 + i], 7);
      c->LTP_scale_Q14 = sk_ltpscales_table_q14[ix->LTP_scaleIndex];
   } else {
      for (int k = 0; k < ch->nb_subfr; k++) c->pitchL[k] = 0;
      for (int i = 0; i < 5 * ch->nb_subfr; i++) c->LTPCoef_Q14[i] = 0;
      ix->PERIndex = 0; c->LTP_scale_Q14 = 0;
   }
}

struct SdScratch { WV_LDS i32 *sLTP_Q15, *res_Q14, *sLPC_Q14; WV_LDS i16 *sLTP, *pulses, *tmp; WV_LDS SdCtrl *ctrl; i32 *cng_exc; /* this channel's CNG excitation buffer (HBM) */ };

/* cng_exc: this channel's comfort-noise excitation buffer, which lives apart from the hot state (HBM) but belongs to the reference's decoder state all the same:
 * the reset clears it with everything else (a stale one is drawn from by the next concealment) */
WV_DEV void sd_reset(WV_LDS OaSilkChannel *ch, i32 *cng_exc)                                                      /* init_decoder.c:43 (whole state) */
{
   WV_LDS i32 *w = (WV_LDS i32 *)ch;
   for (int i = 0; i < (int)(sizeof(OaSilkChannel) / 4); i++) w[i] = 0;
   for (int i = 0; i < 320; i++) cng_exc[i] = 0;
   ch->first_frame_after_reset = 1;
   ch->prev_gain_Q16 = 65536;
   ch->cng_smth_Gain_Q16 = 0; ch->cng_rand_seed = 3176576;                                                  /* silk_CNG_Reset with LPC_order == 0 (CNG.c:58) */
   ch->plc_pitchL_Q8 = 0; ch->plc_prevGain_Q16[0] = ch->plc_prevGain_Q16[1] = 65536; ch->plc_subfr_length = 20; ch->plc_nb_subfr = 2;   /* silk_PLC_Reset (PLC.c:65) */
}
/* silk_decoder_set_fs (decoder_set_fs.c:35); returns nonzero when the resampler has to be re-initialised by the caller */
WV_DEV int sd_set_fs(WV_LDS OaSilkChannel *ch, int fs_kHz, i32 fs_API_Hz)
{
   int reinit = 0;
   ch->subfr_length = 5 * fs_kHz;
   const int frame_length = ch->nb_subfr * ch->subfr_length;
   if (ch->fs_kHz != fs_kHz || ch->fs_API_hz != fs_API_Hz) { reinit = 1; ch->fs_API_hz = fs_API_Hz; }
   if (ch->fs_kHz != fs_kHz || frame_length != ch->frame_length) {
      if (ch->fs_kHz != fs_kHz) {
         ch->ltp_mem_length = 20 * fs_kHz;
         ch->LPC_order = (fs_kHz == 8 || fs_kHz == 12) ? 10 : 16;
         ch->first_frame_after_reset = 1;
         ch->lagPrev = 100; ch->LastGainIndex = 10; ch->prevSignalType = SD_TYPE_NO_VOICE;
         for (int i = 0; i < 480; i++) ch->outBuf[i] = 0;
         for (int i = 0; i < 16; i++) ch->sLPC_Q14_buf[i] = 0;
      }
      ch->fs_kHz = fs_kHz; ch->frame_length = frame_length;
   }
   return reinit;
}


/* ---- packet loss concealment and comfort noise (silk/PLC.c, silk/CNG.c) ---- */
template <class PX> WV_DEV void sd_sum_sqr_shift(i32 *energy, int *shift, PX x, int len)                         /* sum_sqr_shift.c:36 */
{
   int shft = 31 - sk_clz(len);
   i32 nrg = len;
   for (int pass = 0; pass < 2; pass++) {
      int i;
      if (pass) { shft = imax(0, shft + 3 - sk_clz(nrg)); nrg = 0; }
      for (i = 0; i < len - 1; i += 2) { const u32 t = (u32)((i32)x[i] * x[i]) + (u32)((i32)x[i + 1] * x[i + 1]); nrg = (i32)((u32)nrg + (t >> shft)); }
      if (i < len) { const u32 t = (u32)((i32)x[i] * x[i]); nrg = (i32)((u32)nrg + (t >> shft)); }
   }
   *shift = shft; *energy = nrg;
}
WV_DEV i32 sd_sqrt_approx(i32 x)                                                                            /* Inlines.h:67 */
{
   if (x <= 0) return 0;
   const int lz = sk_clz(x), rot = 24 - lz;
   const u32 u = (u32)x;
   const i32 frac_Q7 = (i32)((rot == 0 ? u : rot < 0 ? ((u << -rot) | (u >> (32 + rot))) : ((u << (32 - rot)) | (u >> rot))) & 0x7f);
   i32 y = (lz & 1) ? 32768 : 46214;
   y >>= lz >> 1;
   return sk_mlawb(y, y, sk_mulbb(213, frac_Q7));
}
template <class CH> WV_DEV void sd_plc_reset(CH ch)                                                                 /* PLC.c:65 */
{ ch->plc_pitchL_Q8 = shl32(ch->frame_length, 7); ch->plc_prevGain_Q16[0] = ch->plc_prevGain_Q16[1] = 65536; ch->plc_subfr_length = 20; ch->plc_nb_subfr = 2; }
template <class CH, class CT> WV_DEV void sd_plc_update(CH ch, CT c)                                               /* PLC.c:107 */
{
   ch->prevSignalType = ch->indices.signalType;
   i32 LTP_Gain_Q14 = 0;
   if (ch->indices.signalType == SD_TYPE_VOICED) {
      for (int j = 0; j * ch->subfr_length < c->pitchL[ch->nb_subfr - 1]; j++) {
         if (j == ch->nb_subfr) break;
         i32 t = 0;
         for (int i = 0; i < 5; i++) t += c->LTPCoef_Q14[(ch->nb_subfr - 1 - j) * 5 + i];
         if (t > LTP_Gain_Q14) {
            LTP_Gain_Q14 = t;
            for (int i = 0; i < 5; i++) ch->plc_LTPCoef_Q14[i] = c->LTPCoef_Q14[(ch->nb_subfr - 1 - j) * 5 + i];
            ch->plc_pitchL_Q8 = shl32(c->pitchL[ch->nb_subfr - 1 - j], 8);
         }
      }
      for (int i = 0; i < 5; i++) ch->plc_LTPCoef_Q14[i] = 0;
      ch->plc_LTPCoef_Q14[2] = (i16)LTP_Gain_Q14;
      if (LTP_Gain_Q14 < 11469) {
         const i32 scale_Q10 = shl32(11469, 10) / imax(LTP_Gain_Q14, 1);
         for (int i = 0; i < 5; i++) ch->plc_LTPCoef_Q14[i] = (i16)(sk_mulbb(ch->plc_LTPCoef_Q14[i], scale_Q10) >> 10);
      } else if (LTP_Gain_Q14 > 15565) {
         const i32 scale_Q14 = shl32(15565, 14) / imax(LTP_Gain_Q14, 1);
         for (int i = 0; i < 5; i++) ch->plc_LTPCoef_Q14[i] = (i16)(sk_mulbb(ch->plc_LTPCoef_Q14[i], scale_Q14) >> 14);
      }
   } else {
      ch->plc_pitchL_Q8 = shl32(sk_mulbb(ch->fs_kHz, 18), 8);
      for (int i = 0; i < 5; i++) ch->plc_LTPCoef_Q14[i] = 0;
   }
   for (int i = 0; i < ch->LPC_order; i++) ch->plc_prevLPC_Q12[i] = c->PredCoef_Q12[1][i];
   ch->plc_prevLTP_scale_Q14 = (i16)c->LTP_scale_Q14;
   ch->plc_prevGain_Q16[0] = c->Gains_Q16[ch->nb_subfr - 2]; ch->plc_prevGain_Q16[1] = c->Gains_Q16[ch->nb_subfr - 1];
   ch->plc_subfr_length = ch->subfr_length; ch->plc_nb_subfr = ch->nb_subfr;
}
WV_DEV void sd_plc_conceal(WV_LDS OaSilkChannel *ch, WV_LDS SdCtrl *c, WV_LDS i16 *frame, const SdScratch &S)             /* PLC.c:198 */
{
   const int mem = ch->ltp_mem_length, L = ch->subfr_length, P = ch->LPC_order;
   WV_LDS i32 *sLTP_Q14 = S.sLTP_Q15;
   i32 prevGain_Q10[2] = { ch->plc_prevGain_Q16[0] >> 6, ch->plc_prevGain_Q16[1] >> 6 };
   if (ch->first_frame_after_reset) for (int i = 0; i < 16; i++) ch->plc_prevLPC_Q12[i] = 0;
   /* energies of the last two subframes of the previous excitation decide where the noise is drawn from (PLC.c:172-196) */
   i32 energy1, energy2; int shift1, shift2;
   {
      WV_LDS i16 *eb = S.pulses;
      for (int k = 0; k < 2; k++) for (int i = 0; i < L; i++) eb[k * L + i] = (i16)sk_sat16(sk_mulww(ch->exc_Q14[i + (k + ch->nb_subfr - 2) * L], prevGain_Q10[k]) >> 8);
      sd_sum_sqr_shift(&energy1, &shift1, eb, L); sd_sum_sqr_shift(&energy2, &shift2, eb + L, L);
   }
   const WV_LDS i32 *rand_ptr;
   if ((energy1 >> shift2) < (energy2 >> shift1)) rand_ptr = &ch->exc_Q14[imax(0, (ch->plc_nb_subfr - 1) * ch->plc_subfr_length - 128)];
   else rand_ptr = &ch->exc_Q14[imax(0, ch->plc_nb_subfr * ch->plc_subfr_length - 128)];
   WV_LDS i16 *B_Q14 = ch->plc_LTPCoef_Q14;
   i16 rand_scale_Q14 = (i16)ch->plc_randScale_Q14;
   const int att = imin(1, ch->lossCnt);
   const i32 harm_Gain_Q15 = att ? 31130 : 32440;
   i32 rand_Gain_Q15 = ch->prevSignalType == SD_TYPE_VOICED ? (att ? 26214 : 31130) : (att ? 29491 : 32440);
   sd_bwexpander(ch->plc_prevLPC_Q12, P, 64881);                                                          /* SILK_FIX_CONST(0.99, 16) */
   i16 A_Q12[16];
   for (int i = 0; i < 16; i++) A_Q12[i] = i < P ? ch->plc_prevLPC_Q12[i] : (i16)0;
   if (ch->lossCnt == 0) {
      rand_scale_Q14 = 1 << 14;
      if (ch->prevSignalType == SD_TYPE_VOICED) {
         for (int i = 0; i < 5; i++) rand_scale_Q14 = (i16)(rand_scale_Q14 - B_Q14[i]);
         rand_scale_Q14 = (i16)imax(3277, rand_scale_Q14);
         rand_scale_Q14 = (i16)(sk_mulbb(rand_scale_Q14, ch->plc_prevLTP_scale_Q14) >> 14);
      } else {
         const i32 invGain_Q30 = sd_lpc_inverse_pred_gain(ch->plc_prevLPC_Q12, P);
         i32 down_scale_Q30 = imin(((i32)1 << 30) >> 3, invGain_Q30);
         down_scale_Q30 = imax(((i32)1 << 30) >> 8, down_scale_Q30);
         down_scale_Q30 = shl32(down_scale_Q30, 3);
         rand_Gain_Q15 = sk_mulwb(down_scale_Q30, rand_Gain_Q15) >> 14;
      }
   }
   i32 rand_seed = ch->plc_rand_seed;
   int lag = sk_rround(ch->plc_pitchL_Q8, 8);
   int sLTP_buf_idx = mem;
   int idx = mem - lag - P - 2;
   for (int n = 0; n < mem - idx; n++) {                                                                  /* silk_LPC_analysis_filter(&sLTP[idx], &outBuf[idx], A_Q12, mem - idx, P) */
      i32 o = 0;
      if (n >= P) {
         const WV_LDS i16 *in = &ch->outBuf[idx + n];
         i32 pred = 0;
         for (int j = 0; j < P; j++) pred = add32(pred, (i32)in[-1 - j] * A_Q12[j]);
         o = sk_sat16(sk_rround(sub32(shl32(in[0], 12), pred), 12));
      }
      S.sLTP[idx + n] = (i16)o;
   }
   i32 inv_gain_Q30 = sk_inverse32_varQ(ch->plc_prevGain_Q16[1], 46);
   inv_gain_Q30 = imin(inv_gain_Q30, 2147483647 >> 1);
   for (int i = idx + P; i < mem; i++) sLTP_Q14[i] = sk_mulwb(inv_gain_Q30, S.sLTP[i]);
   for (int k = 0; k < ch->nb_subfr; k++) {
      for (int i = 0; i < L; i++) {
         const WV_LDS i32 *pl = &sLTP_Q14[sLTP_buf_idx - lag + 2];
         i32 LTP_pred_Q12 = 2;
         for (int j = 0; j < 5; j++) LTP_pred_Q12 = sk_mlawb(LTP_pred_Q12, pl[-j], B_Q14[j]);
         rand_seed = sk_rand(rand_seed);
         const int ri = (rand_seed >> 25) & 127;
         sLTP_Q14[sLTP_buf_idx] = shl32(sk_mlawb(LTP_pred_Q12, rand_ptr[ri], rand_scale_Q14), 2);
         sLTP_buf_idx++;
      }
      for (int j = 0; j < 5; j++) B_Q14[j] = (i16)(sk_mulbb(harm_Gain_Q15, B_Q14[j]) >> 15);
      rand_scale_Q14 = (i16)(sk_mulbb(rand_scale_Q14, rand_Gain_Q15) >> 15);
      ch->plc_pitchL_Q8 = sk_mlawb(ch->plc_pitchL_Q8, ch->plc_pitchL_Q8, 655);
      ch->plc_pitchL_Q8 = imin(ch->plc_pitchL_Q8, shl32(sk_mulbb(18, ch->fs_kHz), 8));
      lag = sk_rround(ch->plc_pitchL_Q8, 8);
   }
   WV_LDS i32 *sLPC = &sLTP_Q14[mem - 16];
   for (int i = 0; i < 16; i++) sLPC[i] = ch->sLPC_Q14_buf[i];
   for (int i = 0; i < ch->frame_length; i++) {
      i32 LPC_pred_Q10 = P >> 1;
      for (int j = 0; j < P; j++) LPC_pred_Q10 = sk_mlawb(LPC_pred_Q10, sLPC[16 + i - 1 - j], A_Q12[j]);
      sLPC[16 + i] = sk_add_sat(sLPC[16 + i], sk_shl_sat(LPC_pred_Q10, 4));
      frame[i] = (i16)sk_sat16(sk_sat16(sk_rround(sk_mulww(sLPC[16 + i], prevGain_Q10[1]), 8)));
   }
   for (int i = 0; i < 16; i++) ch->sLPC_Q14_buf[i] = sLPC[ch->frame_length + i];
   ch->plc_rand_seed = rand_seed;
   ch->plc_randScale_Q14 = rand_scale_Q14;
   for (int i = 0; i < 4; i++) c->pitchL[i] = lag;
}
WV_DEV void sd_plc(WV_LDS OaSilkChannel *ch, WV_LDS SdCtrl *c, WV_LDS i16 *frame, int lost, const SdScratch &S)           /* PLC.c:77 */
{
   if (ch->fs_kHz != ch->plc_fs_kHz) { sd_plc_reset(ch); ch->plc_fs_kHz = ch->fs_kHz; }
   if (lost) { sd_plc_conceal(ch, c, frame, S); ch->lossCnt++; }
   else sd_plc_update(ch, c);
}
template <class CH, class PF> WV_DEV void sd_plc_glue_frames(CH ch, PF frame, int length)                            /* PLC.c:441 */
{
   if (ch->lossCnt) {
      i32 en; int sh; sd_sum_sqr_shift(&en, &sh, frame, length); ch->plc_conc_energy = en; ch->plc_conc_energy_shift = sh;
      ch->plc_last_frame_lost = 1;
   } else {
      if (ch->plc_last_frame_lost) {
         i32 energy; int energy_shift;
         sd_sum_sqr_shift(&energy, &energy_shift, frame, length);
         if (energy_shift > ch->plc_conc_energy_shift) ch->plc_conc_energy = ch->plc_conc_energy >> (energy_shift - ch->plc_conc_energy_shift);
         else if (energy_shift < ch->plc_conc_energy_shift) energy = energy >> (ch->plc_conc_energy_shift - energy_shift);
         if (energy > ch->plc_conc_energy) {
            int LZ = sk_clz(ch->plc_conc_energy) - 1;
            ch->plc_conc_energy = shl32(ch->plc_conc_energy, LZ);
            energy = energy >> imax(24 - LZ, 0);
            const i32 frac_Q24 = ch->plc_conc_energy / imax(energy, 1);
            i32 gain_Q16 = shl32(sd_sqrt_approx(frac_Q24), 4);
            i32 slope_Q16 = (((i32)1 << 16) - gain_Q16) / length;
            slope_Q16 = shl32(slope_Q16, 2);
            for (int i = 0; i < length; i++) {
               frame[i] = (i16)sk_mulwb(gain_Q16, frame[i]);
               gain_Q16 += slope_Q16;
               if (gain_Q16 > (i32)1 << 16) break;
            }
         }
      }
      ch->plc_last_frame_lost = 0;
   }
}
template <class CH> WV_DEV void sd_cng_reset(CH ch)                                                                 /* CNG.c:58 */
{
   const i32 step = 32767 / (ch->LPC_order + 1);
   i32 acc = 0;
   for (int i = 0; i < ch->LPC_order; i++) { acc += step; ch->cng_smth_NLSF_Q15[i] = (i16)acc; }
   ch->cng_smth_Gain_Q16 = 0; ch->cng_rand_seed = 3176576;
}
WV_DEV void sd_cng(WV_LDS OaSilkChannel *ch, const WV_LDS SdCtrl *c, WV_LDS i16 *frame, int length, const SdScratch &S)   /* CNG.c:79 */
{
   if (ch->fs_kHz != ch->cng_fs_kHz) { sd_cng_reset(ch); ch->cng_fs_kHz = ch->fs_kHz; }
   if (ch->lossCnt == 0 && ch->prevSignalType == SD_TYPE_NO_VOICE) {
      for (int i = 0; i < ch->LPC_order; i++) ch->cng_smth_NLSF_Q15[i] = (i16)(ch->cng_smth_NLSF_Q15[i] + sk_mulwb((i32)ch->prevNLSF_Q15[i] - (i32)ch->cng_smth_NLSF_Q15[i], 16348));
      i32 max_Gain_Q16 = 0; int subfr = 0;
      for (int i = 0; i < ch->nb_subfr; i++) if (c->Gains_Q16[i] > max_Gain_Q16) { max_Gain_Q16 = c->Gains_Q16[i]; subfr = i; }
      for (int i = (ch->nb_subfr - 1) * ch->subfr_length - 1; i >= 0; i--) S.cng_exc[ch->subfr_length + i] = S.cng_exc[i];
      for (int i = 0; i < ch->subfr_length; i++) S.cng_exc[i] = ch->exc_Q14[subfr * ch->subfr_length + i];
      for (int i = 0; i < ch->nb_subfr; i++) {
         ch->cng_smth_Gain_Q16 += sk_mulwb(c->Gains_Q16[i] - ch->cng_smth_Gain_Q16, 4634);
         if (sk_mulww(ch->cng_smth_Gain_Q16, 46396) > c->Gains_Q16[i]) ch->cng_smth_Gain_Q16 = c->Gains_Q16[i];
      }
   }
   if (ch->lossCnt) {
      WV_LDS i32 *sig = S.sLTP_Q15;
      i32 gain_Q16 = sk_mulww(ch->plc_randScale_Q14, ch->plc_prevGain_Q16[1]);
      if (gain_Q16 >= (1 << 21) || ch->cng_smth_Gain_Q16 > (1 << 23)) {
         gain_Q16 = (gain_Q16 >> 16) * (gain_Q16 >> 16);
         gain_Q16 = (ch->cng_smth_Gain_Q16 >> 16) * (ch->cng_smth_Gain_Q16 >> 16) - shl32(gain_Q16, 5);
         gain_Q16 = shl32(sd_sqrt_approx(gain_Q16), 16);
      } else {
         gain_Q16 = sk_mulww(gain_Q16, gain_Q16);
         gain_Q16 = sk_mulww(ch->cng_smth_Gain_Q16, ch->cng_smth_Gain_Q16) - shl32(gain_Q16, 5);
         gain_Q16 = shl32(sd_sqrt_approx(gain_Q16), 8);
      }
      const i32 gain_Q10 = gain_Q16 >> 6;
      {  /* silk_CNG_exc (CNG.c:36) */
         int exc_mask = 255; while (exc_mask > length) exc_mask >>= 1;
         i32 seed = ch->cng_rand_seed;
         for (int i = 0; i < length; i++) { seed = sk_rand(seed); sig[16 + i] = S.cng_exc[(seed >> 24) & exc_mask]; }
         ch->cng_rand_seed = seed;
      }
      i16 A_Q12[16];
      for (int i = 0; i < 16; i++) A_Q12[i] = 0;
      sd_nlsf2a(A_Q12, ch->cng_smth_NLSF_Q15, ch->LPC_order);
      for (int i = 0; i < 16; i++) sig[i] = ch->cng_synth_state[i];
      for (int i = 0; i < length; i++) {
         i32 LPC_pred_Q10 = ch->LPC_order >> 1;
         for (int j = 0; j < ch->LPC_order; j++) LPC_pred_Q10 = sk_mlawb(LPC_pred_Q10, sig[16 + i - 1 - j], A_Q12[j]);
         sig[16 + i] = sk_add_sat(sig[16 + i], sk_shl_sat(LPC_pred_Q10, 4));
         frame[i] = (i16)sk_sat16((i32)frame[i] + sk_sat16(sk_rround(sk_mulww(sig[16 + i], gain_Q10), 8)));
      }
      for (int i = 0; i < 16; i++) ch->cng_synth_state[i] = sig[length + i];
   } else for (int i = 0; i < ch->LPC_order; i++) ch->cng_synth_state[i] = 0;
}

/* ---- silk_decode_core, wave-wide (decode_core.c:38): the reference's sample loop spread over the 64 lanes:
 *   excitation    the dither seed is an affine recurrence s <- a*s + c + pulse[i]; affine maps compose associatively, so each lane composes its
 *                 5 samples, a 6-step scan over the lanes gives every lane its starting seed, and the lane replays its 5 samples
 *   re-whitening  an FIR: one output per lane
 *   LTP synthesis a recurrence with delay lag-2 >= 14: chunks of min(64, lag-2) samples, one per lane
 *   LPC synthesis a true 16-tap recursion: lane j holds tap j and the sample of lag j+1; per sample one multiply per lane, a DPP sum over the wave,
 *                 and a one-lane shift of the delay line (wv_shift_up1) */
WV_DEV void sd_decode_core_wave(WV_LDS OaSilkChannel *ch, WV_LDS SdCtrl *c, WV_LDS i16 *xq, const SdScratch &S)
{
   const int lane = wv_lane();
   const WV_LDS OaSilkIndices *ix = &ch->indices;
   const int L = ch->subfr_length, mem = ch->ltp_mem_length, P = ch->LPC_order, FL = ch->frame_length;
   const i32 offset_Q10 = k_silk_quant_offsets_Q10[(ix->signalType >> 1) * 2 + ix->quantOffsetType];
   const int interp_flag = ix->NLSFInterpCoef_Q2 < 4;
   /* ---- excitation ---- */
   {
      const int B = 5, i0 = lane * B;
      u32 Am = 1, Cm = 0;                                                   /* this lane's block as one affine map x -> Am*x + Cm */
      for (int t = 0; t < B; t++) { const int i = i0 + t; if (i < FL) { Am = Am * 196314165u; Cm = Cm * 196314165u + 907633515u + (u32)(i32)S.pulses[i]; } }
      u32 Ai = Am, Ci = Cm;                                                 /* inclusive scan: (Ai, Ci) = composition of blocks 0..lane */
      for (int d = 1; d < WV_WIDTH; d <<= 1) {
         const u32 Ap = (u32)wv_shfl((i32)Ai, lane - d), Cp = (u32)wv_shfl((i32)Ci, lane - d);
         if (lane >= d) { Ci = Ai * Cp + Ci; Ai = Ai * Ap; }
      }
      const u32 Ae = (u32)wv_shfl((i32)Ai, lane - 1), Ce = (u32)wv_shfl((i32)Ci, lane - 1);
      i32 seed = lane == 0 ? (i32)ix->Seed : (i32)(Ae * (u32)(i32)ix->Seed + Ce);
      for (int t = 0; t < B; t++) {
         const int i = i0 + t;
         if (i < FL) {
            seed = sk_rand(seed);
            i32 e = shl32(S.pulses[i], 14);
            if (e > 0) e -= 80 << 4; else if (e < 0) e += 80 << 4;
            e += offset_Q10 << 4;
            if (seed < 0) e = -e;
            ch->exc_Q14[i] = e;
            seed = add32(seed, S.pulses[i]);
         }
      }
   }
   wv_sync();
   i32 st = lane < 16 ? ch->sLPC_Q14_buf[15 - lane] : 0;                     /* lane j: the synthesis output of lag j+1 */
   int sLTP_buf_idx = mem, lag = 0;
   i32 prev_gain = ch->prev_gain_Q16;
   for (int k = 0; k < ch->nb_subfr; k++) {
      const WV_LDS i16 *A_Q12 = c->PredCoef_Q12[k >> 1];
      WV_LDS i16 *B_Q14 = &c->LTPCoef_Q14[k * 5];
      int signalType = ix->signalType;
      const i32 Gain_Q16 = c->Gains_Q16[k], Gain_Q10 = Gain_Q16 >> 6;
      i32 inv_gain_Q31 = sk_inverse32_varQ(Gain_Q16, 47);
      i32 gain_adj_Q16 = (i32)1 << 16;
      if (Gain_Q16 != prev_gain) { gain_adj_Q16 = sk_div32_varQ(prev_gain, Gain_Q16, 16); st = sk_mulww(gain_adj_Q16, st); }
      prev_gain = Gain_Q16;
      if (ch->lossCnt && ch->prevSignalType == SD_TYPE_VOICED && ix->signalType != SD_TYPE_VOICED && k < 2) {
         wv_sync();
         if (lane < 5) B_Q14[lane] = lane == 2 ? (i16)4096 : (i16)0;
         if (lane == 0) c->pitchL[k] = ch->lagPrev;
         signalType = SD_TYPE_VOICED;
         wv_sync();
      }
      const WV_LDS i32 *pexc = ch->exc_Q14 + k * L;
      if (signalType == SD_TYPE_VOICED) {
         lag = c->pitchL[k];
         if (k == 0 || (k == 2 && interp_flag)) {
            const int start_idx = mem - lag - P - 2;
            if (k == 2) { for (int i = lane; i < 2 * L; i += WV_WIDTH) ch->outBuf[mem + i] = xq[i]; wv_sync(); }
            for (int n = lane; n < mem - start_idx; n += WV_WIDTH) {
               i32 o = 0;
               if (n >= P) {
                  const WV_LDS i16 *in = &ch->outBuf[start_idx + k * L + n];
                  i32 pred = 0;
                  for (int j = 0; j < P; j++) pred = add32(pred, (i32)in[-1 - j] * A_Q12[j]);
                  o = sk_sat16(sk_rround(sub32(shl32(in[0], 12), pred), 12));
               }
               S.sLTP[start_idx + n] = (i16)o;
            }
            wv_sync();
            if (k == 0) inv_gain_Q31 = shl32(sk_mulwb(inv_gain_Q31, c->LTP_scale_Q14), 2);
            for (int i = lane; i < lag + 2; i += WV_WIDTH) S.sLTP_Q15[sLTP_buf_idx - i - 1] = sk_mulwb(inv_gain_Q31, S.sLTP[mem - i - 1]);
         } else if (gain_adj_Q16 != (i32)1 << 16) {
            for (int i = lane; i < lag + 2; i += WV_WIDTH) S.sLTP_Q15[sLTP_buf_idx - i - 1] = sk_mulww(gain_adj_Q16, S.sLTP_Q15[sLTP_buf_idx - i - 1]);
         }
         wv_sync();
         /* LTP synthesis in chunks no longer than the recursion delay */
         const int chunk = imin(WV_WIDTH, lag - 2);
         i32 b0 = B_Q14[0], b1 = B_Q14[1], b2 = B_Q14[2], b3 = B_Q14[3], b4 = B_Q14[4];
         for (int i0 = 0; i0 < L; i0 += chunk) {
            const int i = i0 + lane;
            if (lane < chunk && i < L) {
               const WV_LDS i32 *pl = &S.sLTP_Q15[sLTP_buf_idx + i - lag + 2];
               i32 p = 2;
               p = sk_mlawb(p, pl[0], b0); p = sk_mlawb(p, pl[-1], b1); p = sk_mlawb(p, pl[-2], b2); p = sk_mlawb(p, pl[-3], b3); p = sk_mlawb(p, pl[-4], b4);
               const i32 r = pexc[i] + shl32(p, 1);
               S.res_Q14[i] = r;
               S.sLTP_Q15[sLTP_buf_idx + i] = shl32(r, 1);
            }
            wv_sync();
         }
         sLTP_buf_idx += L;
      }
      /* LPC synthesis: lane j = tap j */
      {
         const i32 a = lane < P ? (i32)A_Q12[lane] : 0;
         const WV_LDS i32 *res = signalType == SD_TYPE_VOICED ? S.res_Q14 : pexc;
         for (int i = 0; i < L; i++) {
            const i32 pred = add32(P >> 1, wv_sum(lane < P ? sk_mulwb(st, a) : 0));
            const i32 v = sk_add_sat(res[i], sk_shl_sat(pred, 4));
            st = wv_shift_up1(st, v);
            if (lane == 0) xq[k * L + i] = (i16)sk_sat16(sk_rround(sk_mulww(v, Gain_Q10), 8));
         }
      }
      wv_sync();
   }
   if (lane < 16) ch->sLPC_Q14_buf[15 - lane] = st;
   if (lane == 0) ch->prev_gain_Q16 = prev_gain;
   wv_sync();
}

/* silk_decode_frame (decode_frame.c:43) in three steps: entropy + parameters (lane 0), synthesis (wave), bookkeeping (lane 0) */
WV_DEV int sd_decode_frame_front(WV_LDS OaSilkChannel *ch, EC_ARGS, int lostFlag, int condCoding, const SdScratch &S)      /* 1 = decode, 0 = conceal */
{
   S.ctrl->LTP_scale_Q14 = 0;
   if (lostFlag == SD_FLAG_DECODE_NORMAL || (lostFlag == SD_FLAG_DECODE_LBRR && ch->LBRR_flags[ch->nFramesDecoded] == 1)) {
      sd_decode_indices(EC_PASS, ch, ch->nFramesDecoded, lostFlag, condCoding);
      sd_decode_pulses(EC_PASS, S.pulses, ch->indices.signalType, ch->indices.quantOffsetType, ch->frame_length, S.tmp);
      sd_decode_parameters(ch, S.ctrl, condCoding);
      return 1;
   }
   return 0;
}
WV_DEV void sd_decode_frame_back(WV_LDS OaSilkChannel *ch, WV_LDS i16 *pOut, int decoded, const SdScratch &S)
{
   const int L = ch->frame_length;
   WV_LDS SdCtrl *ctrl = S.ctrl;
   if (decoded) {
      const int mv = ch->ltp_mem_length - L;
      for (int i = 0; i < mv; i++) ch->outBuf[i] = ch->outBuf[L + i];
      for (int i = 0; i < L; i++) ch->outBuf[mv + i] = pOut[i];
      sd_plc(ch, ctrl, pOut, 0, S);
      ch->lossCnt = 0;
      ch->prevSignalType = ch->indices.signalType;
      ch->first_frame_after_reset = 0;
   } else {
      sd_plc(ch, ctrl, pOut, 1, S);
      const int mv = ch->ltp_mem_length - L;
      for (int i = 0; i < mv; i++) ch->outBuf[i] = ch->outBuf[L + i];
      for (int i = 0; i < L; i++) ch->outBuf[mv + i] = pOut[i];
   }
   sd_cng(ch, ctrl, pOut, L, S);
   sd_plc_glue_frames(ch, pOut, L);
   ch->lagPrev = ctrl->pitchL[ch->nb_subfr - 1];
}

template <class ECB> WV_DEV void sd_stereo_decode_pred(EC_ARGS_G, i32 *pred_Q13)                                                   /* stereo_decode_pred.c:35 */
{
   int ixs[2][3];
   int n = k_ec_dec_icdf(EC_PASS, sk_stereo_pred_joint_icdf, 8);
   ixs[0][2] = n / 5; ixs[1][2] = n - 5 * ixs[0][2];
   for (n = 0; n < 2; n++) { ixs[n][0] = k_ec_dec_icdf(EC_PASS, sk_uniform3_icdf, 8); ixs[n][1] = k_ec_dec_icdf(EC_PASS, sk_uniform5_icdf, 8); }
   for (n = 0; n < 2; n++) {
      ixs[n][0] += 3 * ixs[n][2];
      const i32 low = sk_stereo_pred_quant_q13[ixs[n][0]];
      const i32 step = sk_mulwb(sk_stereo_pred_quant_q13[ixs[n][0] + 1] - low, 6554);                        /* SILK_FIX_CONST(0.5 / 5, 16) */
      pred_Q13[n] = sk_mlabb(low, step, 2 * ixs[n][1] + 1);
   }
   pred_Q13[0] -= pred_Q13[1];
}
template <class SD, class PX> WV_DEV void sd_stereo_ms_to_lr(SD sd, PX x1, PX x2, const i32 *pred_Q13, int fs_kHz, int frame_length)   /* stereo_MS_to_LR.c:35 */
{
   for (int i = 0; i < 2; i++) { x1[i] = sd->sMid[i]; x2[i] = sd->sSide[i]; sd->sMid[i] = x1[frame_length + i]; sd->sSide[i] = x2[frame_length + i]; }
   i32 pred0 = sd->pred_prev_Q13[0], pred1 = sd->pred_prev_Q13[1];
   const i32 denom_Q16 = ((i32)1 << 16) / (8 * fs_kHz);
   const i32 delta0 = sk_rround(sk_mulbb(pred_Q13[0] - sd->pred_prev_Q13[0], denom_Q16), 16), delta1 = sk_rround(sk_mulbb(pred_Q13[1] - sd->pred_prev_Q13[1], denom_Q16), 16);
   for (int n = 0; n < frame_length; n++) {
      if (n < 8 * fs_kHz) { pred0 += delta0; pred1 += delta1; } else { pred0 = pred_Q13[0]; pred1 = pred_Q13[1]; }
      i32 sum = shl32((i32)x1[n] + (i32)x1[n + 2] + shl32(x1[n + 1], 1), 9);
      sum = sk_mlawb(shl32((i32)x2[n + 1], 8), sum, pred0);
      sum = sk_mlawb(sum, shl32((i32)x1[n + 1], 11), pred1);
      x2[n + 1] = (i16)sk_sat16(sk_rround(sum, 8));
   }
   sd->pred_prev_Q13[0] = pred_Q13[0]; sd->pred_prev_Q13[1] = pred_Q13[1];
   for (int n = 0; n < frame_length; n++) {
      const i32 s = x1[n + 1] + (i32)x2[n + 1], d = x1[n + 1] - (i32)x2[n + 1];
      x1[n + 1] = (i16)sk_sat16(s); x2[n + 1] = (i16)sk_sat16(d);
   }
}
#endif

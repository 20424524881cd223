/* silk_enc.h — the SILK encoder (fixed-point), rows a16, a18, a19 (glue), a20, a21 (frame-level driver), a22 (encoder half) of SURVEY §8.
 *
 * One wavefront encodes one stream; the state (silk_enc_state.h) is staged in LDS for the duration of the frame.  Functions named *_l0 are
 * serial chains that run as lane-0 sections (filters that round at every step, codebook searches with early exits, the entropy coder);
 * functions named *_wave use all 64 lanes (correlations, FIR filters, the pitch estimator of silk_pitch.h).
 *
 * Reference map (each block cites its source again where it stands):
 *   se_control_encoder      silk_control_encoder + setup_fs / setup_complexity / setup_resamplers   silk/control_codec.c:59-423
 *   se_control_audio_bw     silk_control_audio_bandwidth   silk/control_audio_bandwidth.c:36
 *   se_control_snr          silk_control_SNR               silk/control_SNR.c:81
 *   se_vad                  silk_VAD_GetSA_Q8_c / silk_VAD_GetNoiseLevels / silk_encode_do_VAD_FIX   silk/VAD.c:82,:300; silk/fixed/encode_frame_FIX.c:44
 *   se_hp_variable_cutoff   silk_HP_variable_cutoff        silk/HP_variable_cutoff.c:40
 *   se_lp_variable_cutoff   silk_LP_variable_cutoff        silk/LP_variable_cutoff.c:103
 *   se_find_pitch_lags      silk_find_pitch_lags_FIX       silk/fixed/find_pitch_lags_FIX.c:36
 *   (continued in silk_enc_analysis.h, silk_enc_quant.h, silk_enc_frame.h) */
#ifndef OPUS_AMD_SILK_ENC_H
#define OPUS_AMD_SILK_ENC_H
#include "silk_enc_state.h"
#include "silk_enc_tables.h"

#ifndef SE_PHASE            /* shader-clock marks exist only in the -DOA_PHASE_TIMERS profiling build */
#define SE_PHASE(S_, id)
#define SE_CLK_BEGIN()
#define SE_CLK_END(id)
#define SE_PHASE_START(S_)
#define SE_TICK(tk_, id)
#define SE_LTIC()
#define SE_LTOC(id)
#endif
#define SE_FIX(C, Q) ((i32)((C) * ((i64)1 << (Q)) + 0.5))          /* SILK_FIX_CONST (silk/SigProc_FIX.h:574): the literal keeps the reference's type */
#define SE_TYPE_NO_VOICE 0
#define SE_TYPE_UNVOICED 1
#define SE_TYPE_VOICED 2
#define SE_CODE_INDEPENDENTLY 0
#define SE_CODE_INDEPENDENTLY_NO_LTP_SCALING 1
#define SE_CODE_CONDITIONALLY 2
#define SE_VAD_NO_DECISION (-1)
#define SE_VAD_NO_ACTIVITY 0
#define SE_TRANSITION_FRAMES 256
#define SE_MAX_SHAPE_ORDER 24

WV_DEV i32 se_limit(i32 a, i32 l1, i32 l2) { return l1 > l2 ? (a > l1 ? l1 : a < l2 ? l2 : a) : (a > l2 ? l2 : a < l1 ? l1 : a); }   /* silk_LIMIT */
WV_DEV i32 se_add_pos_sat(i32 a, i32 b) { return ((u32)a + (u32)b) & 0x80000000u ? 2147483647 : a + b; }                            /* silk_ADD_POS_SAT32 */
WV_DEV i32 se_ror32(i32 a32, int rot) { const u32 x = (u32)a32; if (rot == 0) return a32; return rot < 0 ? (i32)((x << -rot) | (x >> (32 + rot))) : (i32)((x << (32 - rot)) | (x >> rot)); }
WV_DEV i32 se_sigm_Q15(int in_Q5)                                                                                                  /* silk/sigm_Q15.c:51 */
{
   const i32 slope[6] = {237, 153, 73, 30, 12, 7}, pos[6] = {16384, 23955, 28861, 31213, 32178, 32548}, neg[6] = {16384, 8812, 3906, 1554, 589, 219};
   if (in_Q5 < 0) { in_Q5 = -in_Q5; if (in_Q5 >= 6 * 32) return 0; const int ind = in_Q5 >> 5; return neg[ind] - sk_mulbb(slope[ind], in_Q5 & 0x1F); }
   if (in_Q5 >= 6 * 32) return 32767;
   const int ind = in_Q5 >> 5; return pos[ind] + sk_mulbb(slope[ind], in_Q5 & 0x1F);
}
#define se_lin2log pe_lin2log
#define se_log2lin sd_log2lin
#define se_sqrt_approx sd_sqrt_approx

/* ---- silk_init_encoder (silk/init_encoder.c:46) + silk_VAD_Init (silk/VAD.c:47) ---- */
WV_DEV void se_init_channel(WV_LDS OaSilkEncChannel *c)
{
   WV_LDS i32 *w = (WV_LDS i32 *)c;
   for (int i = 0; i < (int)(sizeof(OaSilkEncChannel) / 4); i++) w[i] = 0;
   c->variable_HP_smth1_Q15 = shl32(se_lin2log(SE_FIX(60, 16)) - (16 << 7), 8);
   c->first_frame_after_reset = 1;
   c->inbuf_reset_req = 1;
   for (int b = 0; b < 4; b++) c->vad_NoiseLevelBias[b] = imax(50 / (b + 1), 1);
   for (int b = 0; b < 4; b++) { c->vad_NL[b] = 100 * c->vad_NoiseLevelBias[b]; c->vad_inv_NL[b] = 2147483647 / c->vad_NL[b]; }
   c->vad_counter = 15;
   for (int b = 0; b < 4; b++) c->vad_NrgRatioSmth_Q8[b] = 100 * 256;
}

/* ---- silk_control_audio_bandwidth ---- */
WV_DEV int se_control_audio_bw(WV_LDS OaSilkEncChannel *c, SeControl *ec)
{
   int orig_kHz = c->fs_kHz;
   if (orig_kHz == 0) orig_kHz = c->lp_saved_fs_kHz;
   int fs_kHz = orig_kHz;
   i32 fs_Hz = sk_mulbb(fs_kHz, 1000);
   if (fs_Hz == 0) { fs_Hz = imin(c->desiredInternal_fs_Hz, c->API_fs_Hz); fs_kHz = fs_Hz / 1000; }
   else if (fs_Hz > c->API_fs_Hz || fs_Hz > c->maxInternal_fs_Hz || fs_Hz < c->minInternal_fs_Hz) {
      fs_Hz = c->API_fs_Hz; fs_Hz = imin(fs_Hz, c->maxInternal_fs_Hz); fs_Hz = imax(fs_Hz, c->minInternal_fs_Hz); fs_kHz = fs_Hz / 1000;
   } else {
      if (c->lp_transition_frame_no >= SE_TRANSITION_FRAMES) c->lp_mode = 0;
      if (c->allow_bandwidth_switch || ec->opusCanSwitch) {
         if (sk_mulbb(orig_kHz, 1000) > c->desiredInternal_fs_Hz) {
            if (c->lp_mode == 0) { c->lp_transition_frame_no = SE_TRANSITION_FRAMES; c->lp_In_LP_State[0] = c->lp_In_LP_State[1] = 0; }
            if (ec->opusCanSwitch) { c->lp_mode = 0; fs_kHz = orig_kHz == 16 ? 12 : 8; }
            else if (c->lp_transition_frame_no <= 0) { ec->switchReady = 1; ec->maxBits -= ec->maxBits * 5 / (ec->payloadSize_ms + 5); }
            else c->lp_mode = -2;
         } else if (sk_mulbb(orig_kHz, 1000) < c->desiredInternal_fs_Hz) {
            if (ec->opusCanSwitch) { fs_kHz = orig_kHz == 8 ? 12 : 16; c->lp_transition_frame_no = 0; c->lp_In_LP_State[0] = c->lp_In_LP_State[1] = 0; c->lp_mode = 1; }
            else if (c->lp_mode == 0) { ec->switchReady = 1; ec->maxBits -= ec->maxBits * 5 / (ec->payloadSize_ms + 5); }
            else c->lp_mode = 1;
         } else if (c->lp_mode < 0) c->lp_mode = 1;
      }
   }
   return fs_kHz;
}

/* ---- the encoder-side input resampler: silk_resampler on lane 0 (W = 1 instance of silk_resampler.h) ---- */
typedef ResamplerLdsT<1> SeRsLds;
WV_DEV OaResamplerCfg se_rs_cfg(const WV_LDS i32 *w) { OaResamplerCfg c; i32 *d = (i32 *)&c; for (int i = 0; i < 9; i++) d[i] = w[i]; return c; }
WV_DEV void se_resampler_init(WV_LDS i32 *cfgw, WV_LDS i32 *rows, i32 Fs_in, i32 Fs_out, int forEnc)
{
   OaResamplerCfg c; rs_init_cfg(&c, Fs_in, Fs_out, forEnc);
   const i32 *s = (const i32 *)&c; for (int i = 0; i < 9; i++) cfgw[i] = s[i];
   for (int i = 0; i < 90; i++) rows[i] = 0;
}
template <class InP> WV_DEV void se_resample_l0(WV_LDS i32 *cfgw, WV_LDS i32 *rows, WV_LDS SeRsLds *R, WV_LDS i16 *out, InP in, int inLen)
{ silk_resampler_lane(se_rs_cfg(cfgw), R, rows, 1, in, inLen, out, 0); }

/* silk_resampler (silk/resampler.c:183) for the encoder's input direction (API rate >= internal rate: copy or silk_resampler_private_down_FIR,
 * resampler_private_down_FIR.c:144) with the whole wave.  Per <= 10 ms batch: all lanes stage the input (coalesced reads of the Opus layer's HBM
 * scratch) into the FIR buffer, lane 0 runs the second-order AR prefilter in place (a rounding recursion, serial by nature; loads batched four ahead),
 * then one lane per output sample evaluates the FIR.  The reference's logical input is delayBuf ++ in, cut into a 1 ms and a (inLen - 1 ms) call whose
 * batches restart the fractional index at 0.  Rb: i32[36 + 480 + 4] of LDS.  Any other method falls back to the single-lane routine. */
template <class InP> WV_DEV void se_resample_wave(WV_LDS i32 *cfgw, WV_LDS i32 *rows, WV_LDS SeRsLds *R, WV_LDS i32 *Rb, WV_LDS i16 *out, InP in, int inLen)
{
   const OaResamplerCfg c = se_rs_cfg(cfgw);
   const int nd = c.inputDelay, lane = wv_lane();
   wv_sync();
   if (c.resampler_function == OA_RS_FN_COPY) {
      for (int k0 = wv_lane(); k0 < inLen; k0 += 8 * WV_WIDTH) {                       /* (eight trips' input samples in flight) */
         i16 v[8];
#pragma unroll
         for (int u = 0; u < 8; u++) { const int k = imin(k0 + u * WV_WIDTH, inLen - 1); v[u] = k < nd ? (i16)rows[OA_RS_ROW_DELAY + k] : (i16)in[k - nd]; }
#pragma unroll
         for (int u = 0; u < 8; u++) { const int k = k0 + u * WV_WIDTH; if (k < inLen) out[k] = v[u]; }
      }
      wv_sync();
      FOR_LANES(j, nd) rows[OA_RS_ROW_DELAY + j] = in[inLen - nd + j];
      wv_sync();
      return;
   }
   if (c.resampler_function != OA_RS_FN_DOWN_FIR) { LANE0 silk_resampler_lane(c, R, rows, 1, in, inLen, out, 0); return; }
   const int ord = c.FIR_Order;
   const i16 *C = rs_coefs(c.coefs_id), *F = C + 2;
   const i32 C0 = C[0], C1 = C[1], inv = c.invRatio_Q16;
   i32 iir0 = rows[OA_RS_ROW_IIR], iir1 = rows[OA_RS_ROW_IIR + 1];
   FOR_LANES(j, ord) Rb[j] = rows[OA_RS_ROW_FIR + j];
   int pos = 0, no = 0;
   for (int seg = 0; seg < 2; seg++) {
      const int len = seg == 0 ? c.Fs_in_kHz : inLen - c.Fs_in_kHz;
      for (int done = 0; done < len;) {
         const int nIn = imin(len - done, c.batchSize);
         for (int k0 = wv_lane(); k0 < nIn; k0 += 8 * WV_WIDTH) {                     /* (eight trips' input samples in flight) */
            i32 v[8];
#pragma unroll
            for (int u = 0; u < 8; u++) { const int q = pos + imin(k0 + u * WV_WIDTH, nIn - 1); v[u] = q < nd ? rows[OA_RS_ROW_DELAY + q] : (i32)in[q - nd]; }
#pragma unroll
            for (int u = 0; u < 8; u++) { const int k = k0 + u * WV_WIDTH; if (k < nIn) Rb[ord + k] = v[u]; }
         }
         wv_sync();
         if (lane == 0) {                                                            /* silk_resampler_private_AR2 (resampler_private_AR2.c:36) */
            WV_LDS i32 *x = Rb + ord;
            int k = 0;
            for (; k + 4 <= nIn; k += 4) {
               const i32 s0 = x[k], s1 = x[k + 1], s2 = x[k + 2], s3 = x[k + 3];
               i32 o;
               o = iir0 + shl32(s0, 8); x[k] = o; o = shl32(o, 2); iir0 = sk_mlawb(iir1, o, C0); iir1 = sk_mulwb(o, C1);
               o = iir0 + shl32(s1, 8); x[k + 1] = o; o = shl32(o, 2); iir0 = sk_mlawb(iir1, o, C0); iir1 = sk_mulwb(o, C1);
               o = iir0 + shl32(s2, 8); x[k + 2] = o; o = shl32(o, 2); iir0 = sk_mlawb(iir1, o, C0); iir1 = sk_mulwb(o, C1);
               o = iir0 + shl32(s3, 8); x[k + 3] = o; o = shl32(o, 2); iir0 = sk_mlawb(iir1, o, C0); iir1 = sk_mulwb(o, C1);
            }
            for (; k < nIn; k++) { i32 o = iir0 + shl32(x[k], 8); x[k] = o; o = shl32(o, 2); iir0 = sk_mlawb(iir1, o, C0); iir1 = sk_mulwb(o, C1); }
         }
         wv_sync();
         const int nOut = (int)((((i64)nIn << 16) + inv - 1) / inv);                   /* index_Q16 = 0, inv, 2 inv, ... < nIn << 16 */
         FOR_LANES(j, nOut) {
            const i32 idx = j * inv; const int b = idx >> 16;
            i32 a;
            if (ord == 18) {                                                          /* resampler_private_down_FIR.c:56-86 */
               const int ph = sk_mulwb(idx & 0xFFFF, c.FIR_Fracs);
               const i16 *c0 = &F[9 * ph], *c1 = &F[9 * (c.FIR_Fracs - 1 - ph)];
               a = sk_mulwb(Rb[b], c0[0]);
               for (int t = 1; t < 9; t++) a = sk_mlawb(a, Rb[b + t], c0[t]);
               for (int t = 0; t < 9; t++) a = sk_mlawb(a, Rb[b + 17 - t], c1[t]);
            } else {                                                                  /* :88-141, symmetric 24 / 36 taps */
               a = sk_mulwb(Rb[b] + Rb[b + ord - 1], F[0]);
               for (int t = 1; t < ord / 2; t++) a = sk_mlawb(a, Rb[b + t] + Rb[b + ord - 1 - t], F[t]);
            }
            out[no + j] = (i16)sk_sat16(sk_rround(a, 6));
         }
         wv_sync();
         { const i32 t = lane < ord ? Rb[nIn + lane] : 0; wv_sync(); if (lane < ord) Rb[lane] = t; wv_sync(); }   /* the tail becomes the head of the next batch */
         no += nOut; pos += nIn; done += nIn;
      }
   }
   LANE0 { rows[OA_RS_ROW_IIR] = iir0; rows[OA_RS_ROW_IIR + 1] = iir1; }
   FOR_LANES(j, ord) rows[OA_RS_ROW_FIR + j] = Rb[j];
   FOR_LANES(j, nd) rows[OA_RS_ROW_DELAY + j] = in[inLen - nd + j];
   wv_sync();
}
/* The two channels of a stereo input side by side (same rates, the down-FIR method): what se_resample_wave does per channel, with the second-order recursion -- the serial
 * part, 960 steps per channel at 48 kHz -- on lane 0 for channel 0 and lane 1 for channel 1 at the same time, and the FIR of both channels in one lane loop.  Rb0 / Rb1:
 * i32[36 + 480 + 4] each.  Returns 0 when the pair does not qualify (the caller runs the channels one after the other). */
template <class InP> WV_DEV int se_resample2_wave(WV_LDS i32 *cfgw0, WV_LDS i32 *rows0, WV_LDS i32 *cfgw1, WV_LDS i32 *rows1, WV_LDS i32 *Rb0, WV_LDS i32 *Rb1,
      WV_LDS i16 *out0, WV_LDS i16 *out1, InP in0, InP in1, int inLen)
{
   const OaResamplerCfg c = se_rs_cfg(cfgw0);
   {
      int same = c.resampler_function == OA_RS_FN_DOWN_FIR;
      for (int i = 0; i < 9; i++) same &= cfgw0[i] == cfgw1[i];
      if (!wv_uni(same)) return 0;
   }
   const int nd = c.inputDelay, lane = wv_lane(), ord = c.FIR_Order;
   wv_sync();
   const i16 *C = rs_coefs(c.coefs_id), *F = C + 2;
   const i32 C0 = C[0], C1 = C[1], inv = c.invRatio_Q16;
   WV_LDS i32 *const rows_l = lane == 1 ? rows1 : rows0;                             /* (lanes 0 and 1 carry a channel's recursion state) */
   WV_LDS i32 *const Rb_l = lane == 1 ? Rb1 : Rb0;
   i32 iir0 = rows_l[OA_RS_ROW_IIR], iir1 = rows_l[OA_RS_ROW_IIR + 1];
   FOR_LANES(j, ord) { Rb0[j] = rows0[OA_RS_ROW_FIR + j]; Rb1[j] = rows1[OA_RS_ROW_FIR + j]; }
   int pos = 0, no = 0;
   for (int seg = 0; seg < 2; seg++) {
      const int len = seg == 0 ? c.Fs_in_kHz : inLen - c.Fs_in_kHz;
      for (int done = 0; done < len;) {
         const int nIn = imin(len - done, c.batchSize);
         for (int k0 = wv_lane(); k0 < nIn; k0 += 4 * WV_WIDTH) {                     /* (four trips' input samples of both channels in flight) */
            i32 v0[4], v1[4];
#pragma unroll
            for (int u = 0; u < 4; u++) {
               const int q = pos + imin(k0 + u * WV_WIDTH, nIn - 1);
               v0[u] = q < nd ? rows0[OA_RS_ROW_DELAY + q] : (i32)in0[q - nd];
               v1[u] = q < nd ? rows1[OA_RS_ROW_DELAY + q] : (i32)in1[q - nd];
            }
#pragma unroll
            for (int u = 0; u < 4; u++) { const int k = k0 + u * WV_WIDTH; if (k < nIn) { Rb0[ord + k] = v0[u]; Rb1[ord + k] = v1[u]; } }
         }
         wv_sync();
         if (lane < 2) {                                                             /* silk_resampler_private_AR2 (resampler_private_AR2.c:36), a channel per lane */
            WV_LDS i32 *x = Rb_l + ord;
            int k = 0;
            for (; k + 4 <= nIn; k += 4) {
               const i32 s0 = x[k], s1 = x[k + 1], s2 = x[k + 2], s3 = x[k + 3];
               i32 o;
               o = iir0 + shl32(s0, 8); x[k] = o; o = shl32(o, 2); iir0 = sk_mlawb(iir1, o, C0); iir1 = sk_mulwb(o, C1);
               o = iir0 + shl32(s1, 8); x[k + 1] = o; o = shl32(o, 2); iir0 = sk_mlawb(iir1, o, C0); iir1 = sk_mulwb(o, C1);
               o = iir0 + shl32(s2, 8); x[k + 2] = o; o = shl32(o, 2); iir0 = sk_mlawb(iir1, o, C0); iir1 = sk_mulwb(o, C1);
               o = iir0 + shl32(s3, 8); x[k + 3] = o; o = shl32(o, 2); iir0 = sk_mlawb(iir1, o, C0); iir1 = sk_mulwb(o, C1);
            }
            for (; k < nIn; k++) { i32 o = iir0 + shl32(x[k], 8); x[k] = o; o = shl32(o, 2); iir0 = sk_mlawb(iir1, o, C0); iir1 = sk_mulwb(o, C1); }
         }
         wv_sync();
         const int nOut = (int)((((i64)nIn << 16) + inv - 1) / inv);                   /* index_Q16 = 0, inv, 2 inv, ... < nIn << 16 */
         FOR_LANES(jj, 2 * nOut) {
            const int ch = jj >= nOut, j = ch ? jj - nOut : jj;
            const WV_LDS i32 *Rb = ch ? Rb1 : Rb0;
            const i32 idx = j * inv; const int b = idx >> 16;
            i32 a;
            if (ord == 18) {                                                          /* resampler_private_down_FIR.c:56-86 */
               const int ph = sk_mulwb(idx & 0xFFFF, c.FIR_Fracs);
               const i16 *c0 = &F[9 * ph], *c1 = &F[9 * (c.FIR_Fracs - 1 - ph)];
               a = sk_mulwb(Rb[b], c0[0]);
               for (int t = 1; t < 9; t++) a = sk_mlawb(a, Rb[b + t], c0[t]);
               for (int t = 0; t < 9; t++) a = sk_mlawb(a, Rb[b + 17 - t], c1[t]);
            } else {                                                                  /* :88-141, symmetric 24 / 36 taps */
               a = sk_mulwb(Rb[b] + Rb[b + ord - 1], F[0]);
               for (int t = 1; t < ord / 2; t++) a = sk_mlawb(a, Rb[b + t] + Rb[b + ord - 1 - t], F[t]);
            }
            (ch ? out1 : out0)[no + j] = (i16)sk_sat16(sk_rround(a, 6));
         }
         wv_sync();
         { const i32 t0 = lane < ord ? Rb0[nIn + lane] : 0, t1 = lane < ord ? Rb1[nIn + lane] : 0; wv_sync(); if (lane < ord) { Rb0[lane] = t0; Rb1[lane] = t1; } wv_sync(); }   /* the tails become the heads of the next batch */
         no += nOut; pos += nIn; done += nIn;
      }
   }
   if (lane < 2) { rows_l[OA_RS_ROW_IIR] = iir0; rows_l[OA_RS_ROW_IIR + 1] = iir1; }
   wv_sync();
   FOR_LANES(j, ord) { rows0[OA_RS_ROW_FIR + j] = Rb0[j]; rows1[OA_RS_ROW_FIR + j] = Rb1[j]; }
   FOR_LANES(j, nd) { rows0[OA_RS_ROW_DELAY + j] = in0[inLen - nd + j]; rows1[OA_RS_ROW_DELAY + j] = in1[inLen - nd + j]; }
   wv_sync();
   return 1;
}

/* ---- silk_setup_resamplers (control_codec.c:134): on a change of internal rate the buffered signal is carried over by resampling it up to the API
 * rate and down again.  tmp: i16[(2 * 20 + 5) * 48] scratch ---- */
WV_DEV void se_setup_resamplers(WV_LDS OaSilkEncChannel *c, int fs_kHz, WV_LDS SeRsLds *R, WV_LDS i16 *tmp, WV_LDS i32 *tmp_rs)
{
   if (c->fs_kHz != fs_kHz || c->prev_API_fs_Hz != c->API_fs_Hz) {
      if (c->fs_kHz == 0) se_resampler_init(c->rs_cfg, c->rs_rows, c->API_fs_Hz, fs_kHz * 1000, 1);
      else {
         const i32 buf_length_ms = shl32(c->nb_subfr * 5, 1) + 5, old_buf_samples = buf_length_ms * c->fs_kHz;
         se_resampler_init(tmp_rs, tmp_rs + 9, sk_mulbb(c->fs_kHz, 1000), c->API_fs_Hz, 0);
         const i32 api_buf_samples = buf_length_ms * (c->API_fs_Hz / 1000);
         se_resample_l0(tmp_rs, tmp_rs + 9, R, tmp, (const WV_LDS i16 *)c->x_buf, old_buf_samples);
         se_resampler_init(c->rs_cfg, c->rs_rows, c->API_fs_Hz, sk_mulbb(fs_kHz, 1000), 1);
         se_resample_l0(c->rs_cfg, c->rs_rows, R, c->x_buf, (const WV_LDS i16 *)tmp, api_buf_samples);
      }
   }
   c->prev_API_fs_Hz = c->API_fs_Hz;
}


/* ---- silk_setup_fs (control_codec.c:198) ---- */
WV_DEV void se_setup_fs(WV_LDS OaSilkEncChannel *c, int fs_kHz, int PacketSize_ms)
{
   if (PacketSize_ms != c->PacketSize_ms) {
      if (PacketSize_ms <= 10) { c->nFramesPerPacket = 1; c->nb_subfr = PacketSize_ms == 10 ? 2 : 1; c->frame_length = sk_mulbb(PacketSize_ms, fs_kHz); c->pitch_LPC_win_length = sk_mulbb(10 + 4, fs_kHz); }
      else { c->nFramesPerPacket = PacketSize_ms / 20; c->nb_subfr = 4; c->frame_length = sk_mulbb(20, fs_kHz); c->pitch_LPC_win_length = sk_mulbb(20 + 4, fs_kHz); }
      c->PacketSize_ms = PacketSize_ms;
      c->TargetRate_bps = 0;
   }
   if (c->fs_kHz != fs_kHz) {
      c->LastGainIndex = 0; c->HarmShapeGain_smth_Q16 = 0; c->Tilt_smth_Q16 = 0;
      c->nsq_reset_req = 1;                                                            /* (silk_nsq_state zeroed, lagPrev = 100, prev_gain_Q16 = 65536: applied by the quantiser stage, se_nsq_apply_reset_wave) */
      for (int i = 0; i < 16; i++) c->prev_NLSFq_Q15[i] = 0;
      c->lp_In_LP_State[0] = c->lp_In_LP_State[1] = 0;
      c->inputBufIx = 0; c->nFramesEncoded = 0; c->TargetRate_bps = 0;
      c->prevLag = 100; c->first_frame_after_reset = 1; c->LastGainIndex = 10; c->prevSignalType = SE_TYPE_NO_VOICE;
      c->fs_kHz = fs_kHz;
      c->predictLPCOrder = (fs_kHz == 8 || fs_kHz == 12) ? 10 : 16;
      c->subfr_length = 5 * fs_kHz; c->frame_length = sk_mulbb(c->subfr_length, c->nb_subfr); c->ltp_mem_length = sk_mulbb(20, fs_kHz);
      c->la_pitch = sk_mulbb(2, fs_kHz); c->max_pitch_lag = sk_mulbb(18, fs_kHz);
      c->pitch_LPC_win_length = c->nb_subfr == 4 ? sk_mulbb(20 + 4, fs_kHz) : sk_mulbb(10 + 4, fs_kHz);
   }
}

/* ---- silk_setup_complexity (control_codec.c:292) ---- */
WV_DEV void se_setup_complexity(WV_LDS OaSilkEncChannel *c, int Complexity)
{
   int pe_cx, thr, pe_order, shp, la, nst, interp, surv, warp;
   const i32 W = c->fs_kHz * SE_FIX(0.015f, 16);
   if (Complexity < 1)      { pe_cx = 0; thr = SE_FIX(0.8, 16);  pe_order = 6;  shp = 12; la = 3; nst = 1; interp = 0; surv = 2;  warp = 0; }
   else if (Complexity < 2) { pe_cx = 1; thr = SE_FIX(0.76, 16); pe_order = 8;  shp = 14; la = 5; nst = 1; interp = 0; surv = 3;  warp = 0; }
   else if (Complexity < 3) { pe_cx = 0; thr = SE_FIX(0.8, 16);  pe_order = 6;  shp = 12; la = 3; nst = 2; interp = 0; surv = 2;  warp = 0; }
   else if (Complexity < 4) { pe_cx = 1; thr = SE_FIX(0.76, 16); pe_order = 8;  shp = 14; la = 5; nst = 2; interp = 0; surv = 4;  warp = 0; }
   else if (Complexity < 6) { pe_cx = 1; thr = SE_FIX(0.74, 16); pe_order = 10; shp = 16; la = 5; nst = 2; interp = 1; surv = 6;  warp = W; }
   else if (Complexity < 8) { pe_cx = 1; thr = SE_FIX(0.72, 16); pe_order = 12; shp = 20; la = 5; nst = 3; interp = 1; surv = 8;  warp = W; }
   else                     { pe_cx = 2; thr = SE_FIX(0.7, 16);  pe_order = 16; shp = 24; la = 5; nst = 4; interp = 1; surv = 16; warp = W; }
   c->pitchEstimationComplexity = pe_cx; c->pitchEstimationThreshold_Q16 = thr; c->pitchEstimationLPCOrder = imin(pe_order, c->predictLPCOrder);
   c->shapingLPCOrder = shp; c->la_shape = la * c->fs_kHz; c->nStatesDelayedDecision = nst; c->useInterpolatedNLSFs = interp; c->NLSF_MSVQ_Survivors = surv; c->warping_Q16 = warp;
   c->shapeWinLength = 5 * c->fs_kHz + 2 * c->la_shape;
   c->Complexity = Complexity;
}

/* ---- silk_control_encoder (control_codec.c:59) ---- */
WV_DEV void se_control_encoder(WV_LDS OaSilkEncChannel *c, SeControl *ec, int allow_bw_switch, int channelNb, int force_fs_kHz, WV_LDS SeRsLds *R, WV_LDS i16 *tmp, WV_LDS i32 *tmp_rs)
{
   c->useDTX = ec->useDTX; c->useCBR = ec->useCBR; c->API_fs_Hz = ec->API_sampleRate; c->maxInternal_fs_Hz = ec->maxInternalSampleRate; c->minInternal_fs_Hz = ec->minInternalSampleRate;
   c->desiredInternal_fs_Hz = ec->desiredInternalSampleRate; c->useInBandFEC = ec->useInBandFEC; c->nChannelsAPI = ec->nChannelsAPI; c->nChannelsInternal = ec->nChannelsInternal;
   c->allow_bandwidth_switch = allow_bw_switch; c->channelNb = channelNb;
   if (c->controlled_since_last_payload != 0 && c->prefillFlag == 0) {
      if (c->API_fs_Hz != c->prev_API_fs_Hz && c->fs_kHz > 0) se_setup_resamplers(c, c->fs_kHz, R, tmp, tmp_rs);
      return;
   }
   int fs_kHz = se_control_audio_bw(c, ec);
   if (force_fs_kHz) fs_kHz = force_fs_kHz;
   se_setup_resamplers(c, fs_kHz, R, tmp, tmp_rs);
   se_setup_fs(c, fs_kHz, ec->payloadSize_ms);
   se_setup_complexity(c, ec->complexity);
   c->PacketLoss_perc = ec->packetLossPercentage;
   {  /* silk_setup_LBRR (:408) */
      const int prev = c->LBRR_enabled;
      c->LBRR_enabled = ec->LBRR_coded;
      if (c->LBRR_enabled) c->LBRR_GainIncreases = prev == 0 ? 7 : imax(7 - sk_mulwb((i32)c->PacketLoss_perc, SE_FIX(0.2, 16)), 3);
   }
   c->controlled_since_last_payload = 1;
}

/* ---- silk_control_SNR ---- */
WV_DEV void se_control_snr(WV_LDS OaSilkEncChannel *c, i32 TargetRate_bps)
{
   c->TargetRate_bps = TargetRate_bps;
   if (c->nb_subfr == 2) TargetRate_bps -= 2000 + c->fs_kHz / 16;
   int bound; const u8 *tab;
   if (c->fs_kHz == 8) { bound = 107; tab = se_targetrate_nb_21; } else if (c->fs_kHz == 12) { bound = 155; tab = se_targetrate_mb_21; } else { bound = 191; tab = se_targetrate_wb_21; }
   int id = (TargetRate_bps + 200) / 400;
   id = imin(id - 10, bound - 1);
   c->SNR_dB_Q7 = id <= 0 ? 0 : tab[id] * 21;
}

/* ---- silk_ana_filt_bank_1 (silk/ana_filt_bank_1.c:40) ---- */
WV_DEV void se_ana_filt_bank_1(const WV_LDS i16 *in, WV_LDS i32 *S, WV_LDS i16 *outL, WV_LDS i16 *outH, int N)
{
   const int N2 = N >> 1;
   i32 S0 = S[0], S1 = S[1];
   int k = 0;
   /* eight output pairs per trip: the sixteen LDS reads of a trip are issued back to back, then the two all-pass chains, then the stores (in place the low band lands on
    * words the trip has already read, the high band on a region of its own: silk_VAD_GetSA_Q8's X_offset layout) */
   for (; k + 8 <= N2; k += 8) {
      i32 e[8], o[8];
#pragma unroll
      for (int u = 0; u < 8; u++) { e[u] = shl32((i32)in[2 * (k + u)], 10); o[u] = shl32((i32)in[2 * (k + u) + 1], 10); }
#pragma unroll
      for (int u = 0; u < 8; u++) {
         i32 Y = sub32(e[u], S0), X = sk_mlawb(Y, Y, -24290);
         const i32 out_1 = add32(S0, X); S0 = add32(e[u], X);
         Y = sub32(o[u], S1); X = sk_mulwb(Y, 5394 << 1);
         const i32 out_2 = add32(S1, X); S1 = add32(o[u], X);
         e[u] = sk_sat16(sk_rround(add32(out_2, out_1), 11));
         o[u] = sk_sat16(sk_rround(sub32(out_2, out_1), 11));
      }
#pragma unroll
      for (int u = 0; u < 8; u++) { outL[k + u] = (i16)e[u]; outH[k + u] = (i16)o[u]; }
   }
   for (; k < N2; k++) {
      i32 in32 = shl32((i32)in[2 * k], 10), Y = sub32(in32, S0), X = sk_mlawb(Y, Y, -24290);
      const i32 out_1 = add32(S0, X); S0 = add32(in32, X);
      in32 = shl32((i32)in[2 * k + 1], 10); Y = sub32(in32, S1); X = sk_mulwb(Y, 5394 << 1);
      const i32 out_2 = add32(S1, X); S1 = add32(in32, X);
      outL[k] = (i16)sk_sat16(sk_rround(add32(out_2, out_1), 11));
      outH[k] = (i16)sk_sat16(sk_rround(sub32(out_2, out_1), 11));
   }
   S[0] = S0; S[1] = S1;
}

/* ---- silk_VAD_GetSA_Q8_c + silk_encode_do_VAD_FIX.  X: i16 scratch of 7/8 frame_length + frame_length/2 words; pIn = inputBuf + 1 ---- */
WV_DEV void se_vad_l0(WV_LDS OaSilkEncChannel *c, const WV_LDS i16 *pIn, WV_LDS i16 *X, int activity)
{
   const int fl = c->frame_length;
   const int dl1 = fl >> 1, dl2 = fl >> 2, dl = fl >> 3;
   int X_offset[4]; X_offset[0] = 0; X_offset[1] = dl + dl2; X_offset[2] = X_offset[1] + dl; X_offset[3] = X_offset[2] + dl2;
   se_ana_filt_bank_1(pIn, c->vad_AnaState, X, &X[X_offset[3]], fl);
   se_ana_filt_bank_1(X, c->vad_AnaState1, X, &X[X_offset[2]], dl1);
   se_ana_filt_bank_1(X, c->vad_AnaState2, X, &X[X_offset[1]], dl2);
   X[dl - 1] = (i16)(X[dl - 1] >> 1);
   const i16 HPstateTmp = X[dl - 1];
   for (int i = dl - 1; i > 0; i--) { X[i - 1] = (i16)(X[i - 1] >> 1); X[i] = (i16)(X[i] - X[i - 1]); }
   X[0] = (i16)(X[0] - (i16)c->vad_HPstate);
   c->vad_HPstate = HPstateTmp;
   i32 Xnrg[4], NrgToNoiseRatio_Q8[4];
   for (int b = 0; b < 4; b++) {
      const int dfl = fl >> imin(4 - b, 3), dsl = dfl >> 2;
      int off = 0; i32 sumSquared = 0;
      Xnrg[b] = c->vad_XnrgSubfr[b];
      for (int s = 0; s < 4; s++) {
         sumSquared = 0;
         for (int i = 0; i < dsl; i++) { const i32 x_tmp = X[X_offset[b] + i + off] >> 3; sumSquared = sk_mlabb(sumSquared, x_tmp, x_tmp); }
         Xnrg[b] = se_add_pos_sat(Xnrg[b], s < 3 ? sumSquared : sumSquared >> 1);
         off += dsl;
      }
      c->vad_XnrgSubfr[b] = sumSquared;
   }
   {  /* silk_VAD_GetNoiseLevels (VAD.c:300) */
      int min_coef;
      if (c->vad_counter < 1000) { min_coef = 32767 / ((c->vad_counter >> 4) + 1); c->vad_counter++; } else min_coef = 0;
      for (int k = 0; k < 4; k++) {
         i32 nl = c->vad_NL[k];
         const i32 nrg = se_add_pos_sat(Xnrg[k], c->vad_NoiseLevelBias[k]), inv_nrg = 2147483647 / nrg;
         int coef;
         if (nrg > shl32(nl, 3)) coef = 1024 >> 3; else if (nrg < nl) coef = 1024; else coef = sk_mulwb(sk_mulww(inv_nrg, nl), 1024 << 1);
         coef = imax(coef, min_coef);
         c->vad_inv_NL[k] = sk_mlawb(c->vad_inv_NL[k], inv_nrg - c->vad_inv_NL[k], coef);
         nl = 2147483647 / c->vad_inv_NL[k];
         c->vad_NL[k] = imin(nl, 0x00FFFFFF);
      }
   }
   const i32 tiltWeights[4] = {30000, 6000, -12000, -12000};
   i32 sumSquared = 0, input_tilt = 0, speech_nrg;
   for (int b = 0; b < 4; b++) {
      speech_nrg = Xnrg[b] - c->vad_NL[b];
      if (speech_nrg > 0) {
         if ((Xnrg[b] & 0xFF800000) == 0) NrgToNoiseRatio_Q8[b] = shl32(Xnrg[b], 8) / (c->vad_NL[b] + 1);
         else NrgToNoiseRatio_Q8[b] = Xnrg[b] / ((c->vad_NL[b] >> 8) + 1);
         i32 SNR_Q7 = se_lin2log(NrgToNoiseRatio_Q8[b]) - 8 * 128;
         sumSquared = sk_mlabb(sumSquared, SNR_Q7, SNR_Q7);
         if (speech_nrg < ((i32)1 << 20)) SNR_Q7 = sk_mulwb(shl32(se_sqrt_approx(speech_nrg), 6), SNR_Q7);
         input_tilt = sk_mlawb(input_tilt, tiltWeights[b], SNR_Q7);
      } else NrgToNoiseRatio_Q8[b] = 256;
   }
   sumSquared = sumSquared / 4;
   const int pSNR_dB_Q7 = (i16)(3 * se_sqrt_approx(sumSquared));
   i32 SA_Q15 = se_sigm_Q15(sk_mulwb(45000, pSNR_dB_Q7) - 128);
   c->input_tilt_Q15 = shl32(se_sigm_Q15(input_tilt) - 16384, 1);
   speech_nrg = 0;
   for (int b = 0; b < 4; b++) speech_nrg += (b + 1) * ((Xnrg[b] - c->vad_NL[b]) >> 4);
   if (fl == 20 * c->fs_kHz) speech_nrg >>= 1;
   if (speech_nrg <= 0) SA_Q15 >>= 1;
   else if (speech_nrg < 16384) { speech_nrg = shl32(speech_nrg, 16); speech_nrg = se_sqrt_approx(speech_nrg); SA_Q15 = sk_mulwb(32768 + speech_nrg, SA_Q15); }
   c->speech_activity_Q8 = imin(SA_Q15 >> 7, 255);
   i32 smooth_coef_Q16 = sk_mulwb(4096, sk_mulwb((i32)SA_Q15, SA_Q15));
   if (fl == 10 * c->fs_kHz) smooth_coef_Q16 >>= 1;
   for (int b = 0; b < 4; b++) {
      c->vad_NrgRatioSmth_Q8[b] = sk_mlawb(c->vad_NrgRatioSmth_Q8[b], NrgToNoiseRatio_Q8[b] - c->vad_NrgRatioSmth_Q8[b], smooth_coef_Q16);
      const i32 SNR_Q7 = 3 * (se_lin2log(c->vad_NrgRatioSmth_Q8[b]) - 8 * 128);
      c->input_quality_bands_Q15[b] = se_sigm_Q15((SNR_Q7 - 16 * 128) >> 4);
   }
   /* silk_encode_do_VAD_FIX */
   const int thr = SE_FIX(0.05f, 8);
   if (activity == SE_VAD_NO_ACTIVITY && c->speech_activity_Q8 >= thr) c->speech_activity_Q8 = thr - 1;
   if (c->speech_activity_Q8 < thr) {
      c->indices.signalType = SE_TYPE_NO_VOICE;
      c->noSpeechCounter++;
      if (c->noSpeechCounter <= 10) c->inDTX = 0;
      else if (c->noSpeechCounter > 20 + 10) { c->noSpeechCounter = 10; c->inDTX = 0; }
      c->VAD_flags[c->nFramesEncoded] = 0;
   } else { c->noSpeechCounter = 0; c->inDTX = 0; c->indices.signalType = SE_TYPE_UNVOICED; c->VAD_flags[c->nFramesEncoded] = 1; }
}

/* ---- silk_HP_variable_cutoff ---- */
WV_DEV void se_hp_variable_cutoff(WV_LDS OaSilkEncChannel *c)
{
   if (c->prevSignalType != SE_TYPE_VOICED) return;
   const i32 pitch_freq_Hz_Q16 = shl32(c->fs_kHz * 1000, 16) / c->prevLag;
   i32 pitch_freq_log_Q7 = se_lin2log(pitch_freq_Hz_Q16) - (16 << 7);
   const int quality_Q15 = c->input_quality_bands_Q15[0];
   pitch_freq_log_Q7 = sk_mlawb(pitch_freq_log_Q7, sk_mulwb(shl32(-quality_Q15, 2), quality_Q15), pitch_freq_log_Q7 - (se_lin2log(SE_FIX(60, 16)) - (16 << 7)));
   i32 delta_freq_Q7 = pitch_freq_log_Q7 - (c->variable_HP_smth1_Q15 >> 8);
   if (delta_freq_Q7 < 0) delta_freq_Q7 *= 3;
   delta_freq_Q7 = se_limit(delta_freq_Q7, -SE_FIX(0.4f, 7), SE_FIX(0.4f, 7));
   c->variable_HP_smth1_Q15 = sk_mlawb(c->variable_HP_smth1_Q15, sk_mulbb(c->speech_activity_Q8, delta_freq_Q7), SE_FIX(0.1f, 16));
   c->variable_HP_smth1_Q15 = se_limit(c->variable_HP_smth1_Q15, shl32(se_lin2log(60), 8), shl32(se_lin2log(100), 8));
}

/* ---- silk_biquad_alt_stride1 (silk/biquad_alt.c:42), in place ---- */
WV_DEV void se_biquad_alt_stride1(WV_LDS i16 *io, const i32 *B_Q28, const i32 *A_Q28, WV_LDS i32 *S, int len)
{
   const i32 A0_L = (-A_Q28[0]) & 0x3FFF, A0_U = (-A_Q28[0]) >> 14, A1_L = (-A_Q28[1]) & 0x3FFF, A1_U = (-A_Q28[1]) >> 14;
   i32 S0 = S[0], S1 = S[1];
   for (int k = 0; k < len; k++) {
      const i32 inval = io[k];
      const i32 out32_Q14 = shl32(sk_mlawb(S0, B_Q28[0], inval), 2);
      S0 = S1 + sk_rround(sk_mulwb(out32_Q14, A0_L), 14);
      S0 = sk_mlawb(S0, out32_Q14, A0_U);
      S0 = sk_mlawb(S0, B_Q28[1], inval);
      S1 = sk_rround(sk_mulwb(out32_Q14, A1_L), 14);
      S1 = sk_mlawb(S1, out32_Q14, A1_U);
      S1 = sk_mlawb(S1, B_Q28[2], inval);
      io[k] = (i16)sk_sat16((out32_Q14 + (1 << 14) - 1) >> 14);
   }
   S[0] = S0; S[1] = S1;
}
/* ---- silk_LP_variable_cutoff ---- */
WV_DEV void se_lp_variable_cutoff(WV_LDS OaSilkEncChannel *c, WV_LDS i16 *frame, int frame_length)
{
   if (c->lp_mode == 0) return;
   i32 fac_Q16 = shl32(SE_TRANSITION_FRAMES - c->lp_transition_frame_no, 16 - 6);
   const int ind = fac_Q16 >> 16;
   fac_Q16 -= shl32(ind, 16);
   i32 B[3], A[2];
   if (ind < 4) {
      if (fac_Q16 > 0) {
         if (fac_Q16 < 32768) {
            for (int i = 0; i < 3; i++) B[i] = sk_mlawb(se_transition_lp_b_q28[ind * 3 + i], se_transition_lp_b_q28[(ind + 1) * 3 + i] - se_transition_lp_b_q28[ind * 3 + i], fac_Q16);
            for (int i = 0; i < 2; i++) A[i] = sk_mlawb(se_transition_lp_a_q28[ind * 2 + i], se_transition_lp_a_q28[(ind + 1) * 2 + i] - se_transition_lp_a_q28[ind * 2 + i], fac_Q16);
         } else {
            for (int i = 0; i < 3; i++) B[i] = sk_mlawb(se_transition_lp_b_q28[(ind + 1) * 3 + i], se_transition_lp_b_q28[(ind + 1) * 3 + i] - se_transition_lp_b_q28[ind * 3 + i], fac_Q16 - ((i32)1 << 16));
            for (int i = 0; i < 2; i++) A[i] = sk_mlawb(se_transition_lp_a_q28[(ind + 1) * 2 + i], se_transition_lp_a_q28[(ind + 1) * 2 + i] - se_transition_lp_a_q28[ind * 2 + i], fac_Q16 - ((i32)1 << 16));
         }
      } else { for (int i = 0; i < 3; i++) B[i] = se_transition_lp_b_q28[ind * 3 + i]; for (int i = 0; i < 2; i++) A[i] = se_transition_lp_a_q28[ind * 2 + i]; }
   } else { for (int i = 0; i < 3; i++) B[i] = se_transition_lp_b_q28[4 * 3 + i]; for (int i = 0; i < 2; i++) A[i] = se_transition_lp_a_q28[4 * 2 + i]; }
   c->lp_transition_frame_no = se_limit(c->lp_transition_frame_no + c->lp_mode, 0, SE_TRANSITION_FRAMES);
   se_biquad_alt_stride1(frame, B, A, c->lp_In_LP_State, frame_length);
}

/* ================= LPC helpers shared by the analysis stages ================= */
/* silk_apply_sine_window (silk/fixed/apply_sine_window_FIX.c:51) */
WV_DEV void se_apply_sine_window(WV_LDS i16 *px_win, const WV_LDS i16 *px, int win_type, int length)
{
   const i16 freq_table_Q16[27] = {12111, 9804, 8235, 7100, 6239, 5565, 5022, 4575, 4202, 3885, 3612, 3375, 3167, 2984, 2820, 2674, 2542, 2422, 2313, 2214, 2123, 2038, 1961, 1889, 1822, 1760, 1702};
   const int f_Q16 = freq_table_Q16[(length >> 2) - 4], c_Q16 = sk_mulwb((i32)f_Q16, -f_Q16);
   i32 S0, S1;
   if (win_type == 1) { S0 = 0; S1 = f_Q16 + (length >> 3); } else { S0 = (i32)1 << 16; S1 = ((i32)1 << 16) + (c_Q16 >> 1) + (length >> 4); }
   for (int k = 0; k < length; k += 4) {
      px_win[k] = (i16)sk_mulwb((S0 + S1) >> 1, px[k]);
      px_win[k + 1] = (i16)sk_mulwb(S1, px[k + 1]);
      S0 = sk_mulwb(S1, c_Q16) + shl32(S1, 1) - S0 + 1; S0 = imin(S0, (i32)1 << 16);
      px_win[k + 2] = (i16)sk_mulwb((S0 + S1) >> 1, px[k + 2]);
      px_win[k + 3] = (i16)sk_mulwb(S0, px[k + 3]);
      S1 = sk_mulwb(S0, c_Q16) + shl32(S0, 1) - S1; S1 = imin(S1, (i32)1 << 16);
   }
}
/* the multipliers of silk_apply_sine_window (a rounding recursion that depends on the length only): m[k] with px_win[k] = SMULWB(m[k], px[k]) */
WV_DEV void se_sine_window_table(WV_LDS i32 *m, int win_type, int length)
{
   const i16 freq_table_Q16[27] = {12111, 9804, 8235, 7100, 6239, 5565, 5022, 4575, 4202, 3885, 3612, 3375, 3167, 2984, 2820, 2674, 2542, 2422, 2313, 2214, 2123, 2038, 1961, 1889, 1822, 1760, 1702};
   const int f_Q16 = freq_table_Q16[(length >> 2) - 4], c_Q16 = sk_mulwb((i32)f_Q16, -f_Q16);
   i32 S0, S1;
   if (win_type == 1) { S0 = 0; S1 = f_Q16 + (length >> 3); } else { S0 = (i32)1 << 16; S1 = ((i32)1 << 16) + (c_Q16 >> 1) + (length >> 4); }
   for (int k = 0; k < length; k += 4) {
      m[k] = (S0 + S1) >> 1; m[k + 1] = S1;
      S0 = sk_mulwb(S1, c_Q16) + shl32(S1, 1) - S0 + 1; S0 = imin(S0, (i32)1 << 16);
      m[k + 2] = (S0 + S1) >> 1; m[k + 3] = S0;
      S1 = sk_mulwb(S0, c_Q16) + shl32(S0, 1) - S1; S1 = imin(S1, (i32)1 << 16);
   }
}
/* silk_autocorr (silk/fixed/autocorr_FIX.c:36) = _celt_autocorr (celt/celt_lpc.c:284) without window; xx: i16[n] scratch; all lanes.
 * ac[k] = sum_{i>=k} xs[i] xs[i-k] mod 2^32 is order-free, so lane = lag. */
WV_DEV int se_autocorr_wave(WV_LDS i32 *ac, const WV_LDS i16 *x, int n, int count, WV_LDS i16 *xx)
{
   const int lag = imin(n, count) - 1;
   const int ac0_shift = celt_ilog2(n + (n >> 4));
   i32 part = 0;
   FOR_LANES(i, n) part += mult16_16(x[i], x[i]) >> ac0_shift;
   i32 ac0 = 1 + (n << 7) + wv_sum(part);
   ac0 += ac0 >> 7;
   int shift = (celt_ilog2(ac0) - 30 + ac0_shift + 1) / 2;
   if (shift > 0) { FOR_LANES(i, n) xx[i] = (i16)pshr32(x[i], shift); } else { shift = 0; FOR_LANES(i, n) xx[i] = x[i]; }
   wv_sync();
   FOR_LANES(k, lag + 1) { i32 d = 0; for (int i = k; i < n; i++) d = mac16_16(d, xx[i], xx[i - k]); ac[k] = d; }
   wv_sync();
   LANE0 {
      shift = 2 * shift;
      if (shift <= 0) ac[0] += shl32((i32)1, -shift);
      if (ac[0] < 268435456) { const int s2 = 29 - ec_ilog((u32)ac[0]); for (int i = 0; i <= lag; i++) ac[i] = shl32(ac[i], s2); shift -= s2; }
      else if (ac[0] >= 536870912) { int s2 = 1; if (ac[0] >= 1073741824) s2++; for (int i = 0; i <= lag; i++) ac[i] >>= s2; shift += s2; }
      ac[lag + 1] = shift;                                                    /* hand-off of the lane-0 result */
   }
   return ac[lag + 1];
}
/* silk_schur (silk/fixed/schur_FIX.c:36) */
WV_DEV i32 se_schur(i16 *rc_Q15, const WV_LDS i32 *c, int order)
{
   i32 C[SE_MAX_SHAPE_ORDER + 1][2];
   int k, lz = sk_clz(c[0]);
   if (lz < 2) { for (k = 0; k <= order; k++) C[k][0] = C[k][1] = c[k] >> 1; }
   else if (lz > 2) { lz -= 2; for (k = 0; k <= order; k++) C[k][0] = C[k][1] = shl32(c[k], lz); }
   else { for (k = 0; k <= order; k++) C[k][0] = C[k][1] = c[k]; }
   for (k = 0; k < order; k++) {
      if (iabs(C[k + 1][0]) >= C[0][1]) { rc_Q15[k] = C[k + 1][0] > 0 ? (i16)-SE_FIX(.99f, 15) : (i16)SE_FIX(.99f, 15); k++; break; }
      i32 rc_tmp = -(C[k + 1][0] / imax(C[0][1] >> 15, 1));
      rc_tmp = sk_sat16(rc_tmp);
      rc_Q15[k] = (i16)rc_tmp;
      for (int n = 0; n < order - k; n++) {
         const i32 t1 = C[n + k + 1][0], t2 = C[n][1];
         C[n + k + 1][0] = sk_mlawb(t1, shl32(t2, 1), rc_tmp);
         C[n][1] = sk_mlawb(t2, shl32(t1, 1), rc_tmp);
      }
   }
   for (; k < order; k++) rc_Q15[k] = 0;
   return imax(1, C[0][1]);
}
/* silk_k2a (silk/fixed/k2a_FIX.c:36) */
WV_DEV void se_k2a(i32 *A_Q24, const i16 *rc_Q15, int order)
{
   for (int k = 0; k < order; k++) {
      const i32 rc = rc_Q15[k];
      for (int n = 0; n < (k + 1) >> 1; n++) {
         const i32 t1 = A_Q24[n], t2 = A_Q24[k - n - 1];
         A_Q24[n] = sk_mlawb(t1, shl32(t2, 1), rc);
         A_Q24[k - n - 1] = sk_mlawb(t2, shl32(t1, 1), rc);
      }
      A_Q24[k] = -shl32((i32)rc_Q15[k], 9);
   }
}
/* silk_schur / silk_k2a with one lane per correlation pair / coefficient (same layout as se_schur64_wave / se_k2a_Q16_wave); rc: LDS i32[order] */
WV_DEV i32 se_schur_wave(WV_LDS i32 *rc_Q15, const WV_LDS i32 *c, int order)
{
   const int lane = wv_lane();
   const int lz = sk_clz(c[0]);
   i32 a = lane < order ? c[lane + 1] : 0, b = lane <= order ? c[lane] : 0;
   if (lz < 2) { a >>= 1; b >>= 1; } else if (lz > 2) { a = shl32(a, lz - 2); b = shl32(b, lz - 2); }
   wv_sync();
   int k;
   for (k = 0; k < order; k++) {
      const i32 a0 = wv_lane_const<0>(a), b0 = wv_lane_const<0>(b);
      if (iabs(a0) >= b0) { if (lane == 0) rc_Q15[k] = a0 > 0 ? -SE_FIX(.99f, 15) : SE_FIX(.99f, 15); k++; break; }
      const i32 rc_tmp = sk_sat16(-(a0 / imax(b0 >> 15, 1)));
      if (lane == 0) rc_Q15[k] = rc_tmp;
      const i32 na = sk_mlawb(a, shl32(b, 1), rc_tmp), nb = sk_mlawb(b, shl32(a, 1), rc_tmp);
      if (lane < order - k) { a = na; b = nb; }
      a = wv_shift_down1(a, 0);
   }
   if (lane >= k && lane < order) rc_Q15[lane] = 0;
   const i32 nrg = imax(1, wv_lane_const<0>(b));
   wv_sync();
   return nrg;
}
WV_DEV i32 se_k2a_wave(const WV_LDS i32 *rc_Q15, int order)                    /* returns A_Q24[lane] */
{
   const int lane = wv_lane();
   i32 a = 0;
   for (int k = 0; k < order; k++) {
      const i32 rc = rc_Q15[k];
      const i32 other = wv_shfl(a, (k - 1 - lane) & 63);
      if (lane < k) a = sk_mlawb(a, shl32(other, 1), rc);
      if (lane == k) a = -shl32(rc, 9);
   }
   return a;
}
/* silk_bwexpander (silk/bwexpander.c:35) on a plain array */
template <class P> WV_DEV void se_bwexpander(P ar, int d, i32 chirp_Q16)
{
   const i32 cm1 = chirp_Q16 - 65536;
   for (int i = 0; i < d - 1; i++) { ar[i] = (i16)sk_rround(chirp_Q16 * ar[i], 16); chirp_Q16 += sk_rround(chirp_Q16 * cm1, 16); }
   ar[d - 1] = (i16)sk_rround(chirp_Q16 * ar[d - 1], 16);
}
/* silk_LPC_analysis_filter (silk/LPC_analysis_filter.c:49): FIR, every output independent -> lanes */
WV_DEV void se_lpc_analysis_filter_wave(WV_LDS i16 *out, const WV_LDS i16 *in, const WV_LDS i16 *B, int len, int d)
{
   FOR_LANES(ix, len) {
      if (ix < d) { out[ix] = 0; continue; }
      i32 o = 0;
      for (int j = 0; j < d; j++) o = sk_mlabb(o, in[ix - 1 - j], B[j]);                    /* silk_SMLABB_ovflw: wraps */
      o = sub32(shl32((i32)in[ix], 12), o);
      out[ix] = (i16)sk_sat16(sk_rround(o, 12));
   }
}

/* ---- the per-frame control block (silk_encoder_control_FIX, silk/fixed/structs_FIX.h:75-103) ---- */
struct SeEncCtrl {
   i32 Gains_Q16[4];
   i16 PredCoef_Q12[2][16];
   i16 LTPCoef_Q14[20];
   i32 LTP_scale_Q14, pitchL[4];
   i16 AR_Q13[4 * SE_MAX_SHAPE_ORDER];
   i32 LF_shp_Q14[4], Tilt_Q14[4], HarmShapeGain_Q14[4], Lambda_Q10, input_quality_Q14, coding_quality_Q14;
   i32 predGain_Q16, LTPredCodGain_Q7, ResNrg[4], ResNrgQ[4];
   i32 GainsUnq_Q16[4], lastGainIndexPrev;
};

/* ---- silk_find_pitch_lags_FIX.  res: i16[la_pitch + frame + ltp_mem]; x = x_frame - ltp_mem; Wsig: i16[384] window scratch; xx: i16[384]; w32: i32[20] ---- */
WV_DEVN void se_find_pitch_lags_wave(WV_LDS OaSilkEncChannel *c, WV_LDS SeEncCtrl *ctl, WV_LDS i16 *res, const WV_LDS i16 *x, WV_LDS i16 *Wsig, WV_LDS i16 *xx, WV_LDS i32 *w32,
      WV_LDS i16 *A_Q12s, WV_LDS PitchLdsCore *PL)
{
   const int buf_len = c->la_pitch + c->frame_length + c->ltp_mem_length, wl = c->pitch_LPC_win_length, la = c->la_pitch, order = c->pitchEstimationLPCOrder;
   SE_LTIC();
   {
      const WV_LDS i16 *x_ptr = x + buf_len - wl;
      LANE0 { se_apply_sine_window(Wsig, x_ptr, 1, la); se_apply_sine_window(Wsig + wl - la, x_ptr + wl - la, 2, la); }
      FOR_LANES(i, wl - 2 * la) Wsig[la + i] = x_ptr[la + i];
      wv_sync();
   }
   se_autocorr_wave(w32, Wsig, wl, order + 1, xx);
   LANE0 w32[0] = sk_mlawb(w32[0], w32[0], SE_FIX(1e-3f, 16)) + 1;
   {
      WV_LDS i32 *rc = PL->d_srch;                                               /* free until the pitch core runs */
      const i32 res_nrg = se_schur_wave(rc, w32, order);
      LANE0 ctl->predGain_Q16 = sk_div32_varQ(w32[0], imax(res_nrg, 1), 16);
      const i32 a24 = se_k2a_wave(rc, order);
      if (wv_lane() < order) A_Q12s[wv_lane()] = (i16)sk_sat16(a24 >> 12);
      wv_sync();
      LANE0 se_bwexpander(A_Q12s, order, SE_FIX(0.99f, 16));
   }
   SE_LTOC(20);
   se_lpc_analysis_filter_wave(res, x, A_Q12s, buf_len, order);
   wv_sync();
   SE_LTOC(21);
   if (c->indices.signalType != SE_TYPE_NO_VOICE && c->first_frame_after_reset == 0) {
      i32 thrhld_Q13 = SE_FIX(0.6, 13);
      thrhld_Q13 = sk_mlabb(thrhld_Q13, SE_FIX(-0.004, 13), order);
      thrhld_Q13 = sk_mlawb(thrhld_Q13, SE_FIX(-0.1, 21), c->speech_activity_Q8);
      thrhld_Q13 = sk_mlabb(thrhld_Q13, SE_FIX(-0.15, 13), c->prevSignalType >> 1);
      thrhld_Q13 = sk_mlawb(thrhld_Q13, SE_FIX(-0.1, 14), c->input_tilt_Q15);
      thrhld_Q13 = sk_sat16(thrhld_Q13);
      OaPitchCfg pc; pc.Fs_kHz = c->fs_kHz; pc.complexity = c->pitchEstimationComplexity; pc.nb_subfr = c->nb_subfr;
      OaPitchIn pin; pin.prevLag = c->prevLag; pin.LTPCorr_Q15 = c->LTPCorr_Q15; pin.search_thres1_Q16 = c->pitchEstimationThreshold_Q16; pin.search_thres2_Q13 = thrhld_Q13;
      OaPitchOut po;
      wv_sync();
      /* the estimator works in place on the residual (its first step scales the frame down to two bits of headroom: silk_pitch_analysis_core's frame_scaled, :144-155);
       * a residual it has scaled is worked out again for the stages behind it */
      const int scaled = silk_pitch_analysis_wave(pc, PL, res, nullptr, &pin, &po);
      wv_sync();
      if (scaled) { se_lpc_analysis_filter_wave(res, x, A_Q12s, buf_len, order); wv_sync(); }
      LANE0 {
         for (int k = 0; k < 4; k++) ctl->pitchL[k] = po.pitch[k];
         c->indices.lagIndex = po.lagIndex; c->indices.contourIndex = po.contourIndex; c->LTPCorr_Q15 = po.LTPCorr_Q15;
         c->indices.signalType = po.unvoiced ? SE_TYPE_UNVOICED : SE_TYPE_VOICED;
      }
   } else {
      LANE0 { for (int k = 0; k < 4; k++) ctl->pitchL[k] = 0; c->indices.lagIndex = 0; c->indices.contourIndex = 0; c->LTPCorr_Q15 = 0; }
   }
}
#endif

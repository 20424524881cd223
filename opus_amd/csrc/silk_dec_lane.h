/* silk_dec_lane.h — the SILK decoder with one LANE per stream (64 streams per wave): oa_sdec_lane_kernel's body.
 *
 * silk_Decode (silk/dec_API.c:142) is, per stream, serial from end to end: the range decoder (silk/decode_indices.c:35, silk/decode_pulses.c:37, silk/shell_coder.c:118),
 * 16-coefficient chains (silk/decode_parameters.c:35, NLSF_decode.c:62, NLSF2A.c:66), and two recursive filters over the frame (silk/decode_core.c:38: the 5-tap long-term
 * predictor and the 16-tap short-term synthesis), then the resampler's all-pass chains (silk/resampler_private_up2_HQ.c:38).  One wave per stream (silk_dec_api.h, the
 * general kernel) spends a 64-lane instruction on one lane for the first two and ten dependent DPP steps per SAMPLE on the third; here every lane decodes its own stream the
 * way the reference's C does -- the same scalar stage functions the wave code calls from lane 0 (silk_dec.h, generic over where the state lives) -- and 64 of them share a wave.
 *
 * Which packets: the steady state of a SILK-only stream (oa_decode_look_kernel): a SILK-only TOC with one coded frame (10 / 20 / 40 / 60 ms), the stream's last packet SILK-only
 * too, the same internal rate and channel count as last time, nothing lost and no FEC request.  The one thing the look cannot see is a redundant CELT frame behind the SILK
 * data (src/opus_decoder.c:499-526: SILK-only packets carry one when 17 bits are left): the lane works on copies, finds out when its range decoder is done, and then either
 * commits (state, scalars, sample count, final range) or hands the packet to the general kernel's list with the stream record untouched.
 *
 * Where a lane's data lives:
 *   registers / private memory   the channel records (SlCh: silk_decoder_state without its three long arrays), the frame's parameters (SdCtrl), the range decoder, the
 *                                synthesis filter's 16 taps and 16 delayed outputs, the resampler's all-pass states
 *   HBM work rows [i][64 lanes]  pulses, excitation, output history (outBuf), the frame's output, the comfort-noise excitation buffer: every lane reads row i at the same
 *                                time, one 128 / 256-byte line per instruction
 *   HBM, contiguous per lane     the long-term predictor's buffer (sLTP_Q15): its read position depends on the lane's pitch lag
 *   LDS                          the resampler's FIR ring, one column per lane (silk_resampler.h)
 *   the stream record (AoS)      read once at the start and written once at the commit, lane-strided
 * No wave collective is called: the lanes diverge freely (signal types, pulse counts, pitch lags, frame sizes and internal rates may differ from lane to lane). */
#ifndef OPUS_AMD_SILK_DEC_LANE_H
#define OPUS_AMD_SILK_DEC_LANE_H
#include "silk_dec_api.h"

#define SL_STREAMS 64
#ifndef P4_TIC            /* (section timers of the -DOA_PHASE_TIMERS build, tools/phase_profile_sdec.py; nothing in the product) */
#define P4_TIC()
#define P4_TOC(bucket)
#endif

/* a lane's array in a work area: ST = 64 rows interleaved over the wave's lanes, ST = 1 contiguous */
template <class T, int ST> struct LnArr {
   T *p;
   WV_MEM T &operator[](int i) const { return p[(ptrdiff_t)i * ST]; }
   WV_MEM LnArr operator+(int k) const { LnArr r; r.p = p + (ptrdiff_t)k * ST; return r; }
};
typedef LnArr<i16, SL_STREAMS> LnI16;
typedef LnArr<i32, SL_STREAMS> LnI32;
/* the API's interleaved output: sample i of one channel */
struct SlPcmOut {
   i16 *p; int st;
   WV_MEM i16 &operator[](int i) const { return p[(ptrdiff_t)i * st]; }
   WV_MEM SlPcmOut operator+(int k) const { SlPcmOut r; r.p = p + (ptrdiff_t)k * st; r.st = st; return r; }
};

/* per tile of 64 streams, in HBM (bytes): rows of int16 then rows of int32 then the contiguous per-lane buffers */
#define SL_ROWS16 (2 * 480 + 2 * 324 + 336)
#define SL_ROWS32 (2 * 320 + 2 * 320)
#define SL_LTP_WORDS 640
#define SL_WORK_BYTES ((size_t)SL_ROWS16 * SL_STREAMS * 2 + (size_t)SL_ROWS32 * SL_STREAMS * 4 + (size_t)SL_LTP_WORDS * SL_STREAMS * 4)

/* silk_decoder_state (silk/structs.h:236-286) without exc_Q14 / outBuf (work rows) and the resampler's configuration: the field names of OaSilkChannel, so that silk_dec.h's
 * stage functions take either */
struct SlCh {
   i32 prev_gain_Q16;
   i32 sLPC_Q14_buf[16];
   i32 lagPrev, LastGainIndex, fs_kHz, nb_subfr, frame_length, subfr_length, ltp_mem_length, LPC_order;
   i32 first_frame_after_reset, nFramesDecoded, nFramesPerPacket, ec_prevSignalType, ec_prevLagIndex;
   i32 VAD_flags[3], LBRR_flag, LBRR_flags[3];
   i32 lossCnt, prevSignalType;
   i32 rs_rows[90];
   i32 plc_pitchL_Q8, plc_last_frame_lost, plc_conc_energy, plc_conc_energy_shift, plc_prevGain_Q16[2], plc_fs_kHz, plc_nb_subfr, plc_subfr_length;
   i32 plc_prevLTP_scale_Q14;
   i32 cng_synth_state[16], cng_smth_Gain_Q16, cng_rand_seed, cng_fs_kHz;
   i16 prevNLSF_Q15[16], plc_LTPCoef_Q14[6], plc_prevLPC_Q12[16], cng_smth_NLSF_Q15[16];
   OaSilkIndices indices;
   i32 exc_valid;                                                  /* this packet has decoded a frame of this channel: its excitation rows go back at the commit (a side channel that was not coded keeps its old one) */
   i32 cng_loaded;                                                 /* the comfort-noise excitation buffer has been copied into its work rows (and goes back at the commit) */
};
struct SlDec { i32 pred_prev_Q13[2]; i16 sMid[2], sSide[2]; i32 prev_decode_only_middle; };


/* n elements from src to dst, eight loads in flight before the first store: a lane's copy loop is otherwise one memory round trip per element (the compiler cannot tell the
 * rows apart and keeps every load behind the store before it; with one wave per SIMD nothing else hides that latency) */
template <class D, class S> WV_DEV void sl_copy(D dst, S src, int n)
{
   int i = 0;
   for (; i + 8 <= n; i += 8) {
      i32 t[8];
#pragma unroll
      for (int k = 0; k < 8; k++) t[k] = src[i + k];
#pragma unroll
      for (int k = 0; k < 8; k++) dst[i + k] = t[k];
   }
   for (; i < n; i++) dst[i] = src[i];
}

/* The head of every lane's packet in LDS (the tile's waves fill it together, one coalesced row per stream): the range decoder takes a byte every few symbols, and a wave of 64
 * decoders waits for the slowest lane at EVERY symbol -- with the bytes in HBM nearly every symbol of the wave paid a memory round trip for some lane's renormalisation.
 * 33 dwords per lane: the lanes' windows start in different banks.  Bytes beyond the window (long packets) come from HBM as before. */
#define SL_WIN 128
#define SL_WIN_STRIDE 132
struct SlPkt {
   const u8 *g; const WV_LDS u8 *w; u32 sh;                            /* the window holds the packet from its second byte on; the coded frame starts sh bytes into it */
   WV_MEM int operator[](u32 i) const { return i + sh < SL_WIN ? (int)w[i + sh] : (int)g[i]; }
};
/* the pulse coder's tables in LDS, each padded so that a four-entry look-ahead never leaves it (sl_dec_icdf): silk/tables_pulses_per_block.c:34-263 */
struct SlTabs { u8 ppb[180 + 4]; u8 rate[18 + 2]; u8 shell[4][152]; u8 shell_pad[4]; u8 sign[42 + 2]; u8 offs[17 + 3]; };
WV_DEV void sl_tabs_fill(WV_LDS SlTabs *T, int lane)
{
   for (int i = lane; i < 184; i += SL_STREAMS) T->ppb[i] = i < 180 ? sk_pulses_per_block_icdf[i] : (u8)0;
   for (int i = lane; i < 20; i += SL_STREAMS) { T->rate[i] = i < 18 ? sk_rate_levels_icdf[i] : (u8)0; T->offs[i] = i < 17 ? sk_shell_code_table_offsets[i] : (u8)0; }
   for (int i = lane; i < 152; i += SL_STREAMS) { T->shell[0][i] = sk_shell_code_table0[i]; T->shell[1][i] = sk_shell_code_table1[i]; T->shell[2][i] = sk_shell_code_table2[i]; T->shell[3][i] = sk_shell_code_table3[i]; }
   for (int i = lane; i < 44; i += SL_STREAMS) T->sign[i] = i < 42 ? sk_sign_icdf[i] : (u8)0;
   if (lane < 4) T->shell_pad[lane] = 0;
}
/* ec_dec_icdf (celt/entdec.c:181, ftb = 8) on a table in LDS, four entries per look: the scan is a chain of dependent reads otherwise, as long as the slowest lane's symbol */
template <class ECB> WV_DEV int sl_dec_icdf(EcCtx *e, ECB buf, const WV_LDS u8 *icdf)
{
   u32 s = e->rng, t;
   const u32 v = e->val, r = s >> 8;
   int ret = 0;
   for (;; ret += 4) {
      const u32 c0 = icdf[ret], c1 = icdf[ret + 1], c2 = icdf[ret + 2], c3 = icdf[ret + 3];
      t = s; s = r * c0; if (v >= s) break;
      t = s; s = r * c1; if (v >= s) { ret += 1; break; }
      t = s; s = r * c2; if (v >= s) { ret += 2; break; }
      t = s; s = r * c3; if (v >= s) { ret += 3; break; }
   }
   e->val = v - s;
   e->rng = t - s;
   ecd_normalize(e, buf);
   return ret;
}
/* silk_decode_pulses (silk/decode_pulses.c:37) as one lane runs it: a shell block's sixteen counts are built in registers (silk/shell_coder.c:118) and written once; the sign
 * pass (silk/code_signs.c:74) reads a block's sixteen values with one wait, not sixteen */
/* (silk_shell_code_table_offsets[p] = p (p + 1) / 2 - 1 for p >= 1: arithmetic, not a look-up in front of the look-up) */
#define SL_SPLIT(a, b, p, tb) do { if ((p) > 0) { a = sl_dec_icdf(e, buf, &T->shell[tb][(((p) * ((p) + 1)) >> 1) - 1]); b = (p) - a; } else { a = 0; b = 0; } } while (0)
template <class ECB> WV_DEV void sl_decode_pulses(EcCtx *e, ECB buf, LnI16 pulses, int signalType, int quantOffsetType, int frame_length, const WV_LDS SlTabs *T)
{
   i32 sum_pulses[20], nLshifts[20];
   const int RateLevelIndex = sl_dec_icdf(e, buf, &T->rate[(signalType >> 1) * 9]);
   int iter = frame_length >> 4;
   if (iter * 16 < frame_length) iter++;                                                              /* 10 ms at 12 kHz */
   for (int i = 0; i < iter; i++) {
      int nl = 0, sp = sl_dec_icdf(e, buf, &T->ppb[RateLevelIndex * 18]);
      while (sp == 17) { nl++; sp = sl_dec_icdf(e, buf, &T->ppb[9 * 18 + (nl == 10)]); }
      nLshifts[i] = nl; sum_pulses[i] = sp;
   }
   for (int i = 0; i < iter; i++) {
      i32 q[16];
      const int p4 = sum_pulses[i];
      if (p4 > 0) {
         int a3, b3, a2, b2, c2, d2, p1[8];
         SL_SPLIT(a3, b3, p4, 3);
         SL_SPLIT(a2, b2, a3, 2);
         SL_SPLIT(p1[0], p1[1], a2, 1);
         SL_SPLIT(q[0], q[1], p1[0], 0);
         SL_SPLIT(q[2], q[3], p1[1], 0);
         SL_SPLIT(p1[2], p1[3], b2, 1);
         SL_SPLIT(q[4], q[5], p1[2], 0);
         SL_SPLIT(q[6], q[7], p1[3], 0);
         SL_SPLIT(c2, d2, b3, 2);
         SL_SPLIT(p1[4], p1[5], c2, 1);
         SL_SPLIT(q[8], q[9], p1[4], 0);
         SL_SPLIT(q[10], q[11], p1[5], 0);
         SL_SPLIT(p1[6], p1[7], d2, 1);
         SL_SPLIT(q[12], q[13], p1[6], 0);
         SL_SPLIT(q[14], q[15], p1[7], 0);
      } else {
#pragma unroll
         for (int k = 0; k < 16; k++) q[k] = 0;
      }
#pragma unroll
      for (int k = 0; k < 16; k++) pulses[i * 16 + k] = (i16)q[k];
   }
   for (int i = 0; i < iter; i++) {
      if (nLshifts[i] > 0) {
         const int nLS = nLshifts[i];
         for (int k = 0; k < 16; k++) {
            int abs_q = pulses[i * 16 + k];
            for (int j = 0; j < nLS; j++) abs_q = (abs_q << 1) + k_ec_dec_icdf(e, buf, sk_lsb_icdf, 8);
            pulses[i * 16 + k] = (i16)abs_q;
         }
         sum_pulses[i] |= nLS << 5;
      }
   }
   const WV_LDS u8 *icdf_ptr = &T->sign[7 * (quantOffsetType + (signalType << 1))];
   const int nblk = (frame_length + 8) >> 4;
   for (int i = 0; i < nblk; i++) {
      const int p = sum_pulses[i];
      if (p > 0) {
         const u32 c0 = icdf_ptr[imin(p & 0x1F, 6)];
         i32 q[16];
#pragma unroll
         for (int k = 0; k < 16; k++) q[k] = pulses[i * 16 + k];
#pragma unroll
         for (int k = 0; k < 16; k++) {
            if (q[k] > 0) {                                                                          /* ec_dec_icdf on the two-entry table { c0, 0 } */
               const u32 rr = e->rng >> 8, s1 = rr * c0;
               if (e->val >= s1) { e->val -= s1; e->rng -= s1; q[k] = -q[k]; } else e->rng = s1;
               ecd_normalize(e, buf);
            }
         }
#pragma unroll
         for (int k = 0; k < 16; k++) pulses[i * 16 + k] = (i16)q[k];
      }
   }
}
#undef SL_SPLIT

WV_DEV void sl_load_channel(SlCh *d, const OaSilkChannel *g, LnI16 outBuf)
{
   d->prev_gain_Q16 = g->prev_gain_Q16;
   for (int i = 0; i < 16; i++) d->sLPC_Q14_buf[i] = g->sLPC_Q14_buf[i];
   d->lagPrev = g->lagPrev; d->LastGainIndex = g->LastGainIndex; d->fs_kHz = g->fs_kHz; d->nb_subfr = g->nb_subfr; d->frame_length = g->frame_length;
   d->subfr_length = g->subfr_length; d->ltp_mem_length = g->ltp_mem_length; d->LPC_order = g->LPC_order;
   d->first_frame_after_reset = g->first_frame_after_reset; d->nFramesDecoded = g->nFramesDecoded; d->nFramesPerPacket = g->nFramesPerPacket;
   d->ec_prevSignalType = g->ec_prevSignalType; d->ec_prevLagIndex = g->ec_prevLagIndex;
   for (int i = 0; i < 3; i++) { d->VAD_flags[i] = g->VAD_flags[i]; d->LBRR_flags[i] = g->LBRR_flags[i]; }
   d->LBRR_flag = g->LBRR_flag; d->lossCnt = g->lossCnt; d->prevSignalType = g->prevSignalType;
   sl_copy(d->rs_rows, g->rs_rows, 90);
   d->plc_pitchL_Q8 = g->plc_pitchL_Q8; d->plc_last_frame_lost = g->plc_last_frame_lost; d->plc_conc_energy = g->plc_conc_energy; d->plc_conc_energy_shift = g->plc_conc_energy_shift;
   d->plc_prevGain_Q16[0] = g->plc_prevGain_Q16[0]; d->plc_prevGain_Q16[1] = g->plc_prevGain_Q16[1]; d->plc_fs_kHz = g->plc_fs_kHz; d->plc_nb_subfr = g->plc_nb_subfr;
   d->plc_subfr_length = g->plc_subfr_length; d->plc_prevLTP_scale_Q14 = g->plc_prevLTP_scale_Q14;
   for (int i = 0; i < 16; i++) d->cng_synth_state[i] = g->cng_synth_state[i];
   d->cng_smth_Gain_Q16 = g->cng_smth_Gain_Q16; d->cng_rand_seed = g->cng_rand_seed; d->cng_fs_kHz = g->cng_fs_kHz;
   for (int i = 0; i < 16; i++) { d->prevNLSF_Q15[i] = g->prevNLSF_Q15[i]; d->plc_prevLPC_Q12[i] = g->plc_prevLPC_Q12[i]; d->cng_smth_NLSF_Q15[i] = g->cng_smth_NLSF_Q15[i]; }
   for (int i = 0; i < 6; i++) d->plc_LTPCoef_Q14[i] = g->plc_LTPCoef_Q14[i];
   d->indices = g->indices;
   d->cng_loaded = 0; d->exc_valid = 0;
   const i32 *ob = (const i32 *)g->outBuf;                          /* the history the long-term predictor reads (ltp_mem_length samples) and the half frame behind it that
                                                                     * decode_core.c:141 parks there (so that the record comes back byte for byte), as dwords */
   const int nw = (d->ltp_mem_length + 10 * d->fs_kHz) >> 1;
   for (int i = 0; i < nw; i += 8) {
      i32 t[8];
#pragma unroll
      for (int k = 0; k < 8; k++) t[k] = i + k < nw ? ob[i + k] : 0;
#pragma unroll
      for (int k = 0; k < 8; k++) if (i + k < nw) { outBuf[2 * (i + k)] = (i16)(t[k] & 0xFFFF); outBuf[2 * (i + k) + 1] = (i16)(t[k] >> 16); }
   }
}
WV_DEV void sl_store_channel(OaSilkChannel *g, const SlCh *d, LnI16 outBuf, LnI32 exc, LnI32 cng, i32 *cng_exc)
{
   g->prev_gain_Q16 = d->prev_gain_Q16;
   for (int i = 0; i < 16; i++) g->sLPC_Q14_buf[i] = d->sLPC_Q14_buf[i];
   g->lagPrev = d->lagPrev; g->LastGainIndex = d->LastGainIndex; g->nb_subfr = d->nb_subfr; g->frame_length = d->frame_length; g->subfr_length = d->subfr_length;
   g->first_frame_after_reset = d->first_frame_after_reset; g->nFramesDecoded = d->nFramesDecoded; g->nFramesPerPacket = d->nFramesPerPacket;
   g->ec_prevSignalType = d->ec_prevSignalType; g->ec_prevLagIndex = d->ec_prevLagIndex;
   for (int i = 0; i < 3; i++) { g->VAD_flags[i] = d->VAD_flags[i]; g->LBRR_flags[i] = d->LBRR_flags[i]; }
   g->LBRR_flag = d->LBRR_flag; g->lossCnt = d->lossCnt; g->prevSignalType = d->prevSignalType;
   sl_copy(g->rs_rows, d->rs_rows, 90);
   g->plc_pitchL_Q8 = d->plc_pitchL_Q8; g->plc_last_frame_lost = d->plc_last_frame_lost; g->plc_conc_energy = d->plc_conc_energy; g->plc_conc_energy_shift = d->plc_conc_energy_shift;
   g->plc_prevGain_Q16[0] = d->plc_prevGain_Q16[0]; g->plc_prevGain_Q16[1] = d->plc_prevGain_Q16[1]; g->plc_fs_kHz = d->plc_fs_kHz; g->plc_nb_subfr = d->plc_nb_subfr;
   g->plc_subfr_length = d->plc_subfr_length; g->plc_prevLTP_scale_Q14 = d->plc_prevLTP_scale_Q14;
   for (int i = 0; i < 16; i++) g->cng_synth_state[i] = d->cng_synth_state[i];
   g->cng_smth_Gain_Q16 = d->cng_smth_Gain_Q16; g->cng_rand_seed = d->cng_rand_seed; g->cng_fs_kHz = d->cng_fs_kHz;
   for (int i = 0; i < 16; i++) { g->prevNLSF_Q15[i] = d->prevNLSF_Q15[i]; g->plc_prevLPC_Q12[i] = d->plc_prevLPC_Q12[i]; g->cng_smth_NLSF_Q15[i] = d->cng_smth_NLSF_Q15[i]; }
   for (int i = 0; i < 6; i++) g->plc_LTPCoef_Q14[i] = d->plc_LTPCoef_Q14[i];
   g->indices = d->indices;
   i32 *ob = (i32 *)g->outBuf;
   const int nw = (d->ltp_mem_length + 10 * d->fs_kHz) >> 1;
   for (int i = 0; i < nw; i += 8) {
      i32 t[8];
#pragma unroll
      for (int k = 0; k < 8; k++) t[k] = i + k < nw ? (i32)(((u32)outBuf[2 * (i + k)] & 0xFFFFu) | ((u32)outBuf[2 * (i + k) + 1] << 16)) : 0;
#pragma unroll
      for (int k = 0; k < 8; k++) if (i + k < nw) ob[i + k] = t[k];
   }
   if (d->exc_valid) sl_copy(g->exc_Q14, exc, d->frame_length);
   if (d->cng_loaded) sl_copy(cng_exc, cng, 320);
}

/* silk_decode_core (silk/decode_core.c:38) as one lane runs it.  The long-term predictor's state needs only the last lag + 2 whitened samples (:147-160 use sLTP[mem - i - 1],
 * i < lag + 2), so the re-whitening filter (silk_LPC_analysis_filter, silk/LPC_analysis_filter.c:50) runs from the newest sample downwards and stops there: every lane reads
 * the same history row whatever its lag; its outputs go straight into sLTP_Q15.  The residual of a voiced subframe is consumed by the synthesis filter sample by sample
 * (no res_Q14 array), and the synthesis filter keeps its taps and its 16 delayed outputs in registers, four samples per pass so that every index is static. */
template <class CH> WV_DEV void sl_decode_core(CH ch, SdCtrl *c, LnI16 xq, LnI16 pulses, LnI32 exc, LnI16 outBuf, i32 *lt)
{
   const OaSilkIndices *ix = &ch->indices;
   const int L = ch->subfr_length, mem = ch->ltp_mem_length, FL = ch->frame_length, P = ch->LPC_order;
   const i32 offset_Q10 = k_silk_quant_offsets_Q10[(ix->signalType >> 1) * 2 + ix->quantOffsetType];
   const int interp_flag = ix->NLSFInterpCoef_Q2 < 4;
   {
      i32 seed = ix->Seed;
      for (int i0 = 0; i0 < FL; i0 += 8) {                                     /* (frame lengths are multiples of 8) */
         i32 q[8];
#pragma unroll
         for (int k = 0; k < 8; k++) q[k] = pulses[i0 + k];
#pragma unroll
         for (int k = 0; k < 8; k++) {
            seed = sk_rand(seed);
            i32 e = shl32(q[k], 14);
            if (e > 0) e -= 80 << 4; else if (e < 0) e += 80 << 4;
            e += offset_Q10 << 4;
            if (seed < 0) e = -e;
            seed = add32(seed, q[k]);
            q[k] = e;
         }
#pragma unroll
         for (int k = 0; k < 8; k++) exc[i0 + k] = q[k];
      }
   }
   i32 w[20];                                                                 /* w[0..15]: the synthesis outputs of lags 16..1 (oldest first), w[16..19]: the four being made */
#pragma unroll
   for (int j = 0; j < 16; j++) w[j] = ch->sLPC_Q14_buf[j];
   int sLTP_buf_idx = mem, lag = 0;
   i32 prev_gain = ch->prev_gain_Q16;
   const int voiced = ix->signalType == SD_TYPE_VOICED;
   for (int k = 0; k < ch->nb_subfr; k++) {
      i32 A[16];
#pragma unroll
      for (int j = 0; j < 16; j++) A[j] = j < P ? (i32)c->PredCoef_Q12[k >> 1][j] : 0;
      const i32 Gain_Q16 = c->Gains_Q16[k], Gain_Q10 = Gain_Q16 >> 6;
      i32 inv_gain_Q31 = sk_inverse32_varQ(Gain_Q16, 47);
      i32 gain_adj_Q16 = (i32)1 << 16;
      if (Gain_Q16 != prev_gain) {
         gain_adj_Q16 = sk_div32_varQ(prev_gain, Gain_Q16, 16);
#pragma unroll
         for (int j = 0; j < 16; j++) w[j] = sk_mulww(gain_adj_Q16, w[j]);
      }
      prev_gain = Gain_Q16;
      LnI32 pexc = exc + k * L;
      LnI16 pxq = xq + k * L;
      i32 b0 = 0, b1 = 0, b2 = 0, b3 = 0, b4 = 0, t0 = 0, t1 = 0, t2 = 0, t3 = 0;
      const i32 *pl = lt;
      if (voiced) {
         lag = c->pitchL[k];
         if (k == 0 || (k == 2 && interp_flag)) {
            if (k == 2) sl_copy(outBuf + mem, xq, 2 * L);
            if (k == 0) inv_gain_Q31 = shl32(sk_mulwb(inv_gain_Q31, c->LTP_scale_Q14), 2);
            const int top = mem - 1 + k * L;                                   /* newest sample of the history */
            i32 h[20];                                                       /* h[j] = history sample top - i0 - j */
            for (int i0 = 0; i0 < lag + 2; i0 += 4) {
               if (i0 == 0) {
#pragma unroll
                  for (int j = 0; j < 20; j++) { const int p = top - j; h[j] = p >= 0 ? (i32)outBuf[p] : 0; }
               } else {
#pragma unroll
                  for (int j = 0; j < 16; j++) h[j] = h[j + 4];
#pragma unroll
                  for (int j = 16; j < 20; j++) { const int p = top - i0 - j; h[j] = p >= 0 ? (i32)outBuf[p] : 0; }
               }
               i32 o4[4];
#pragma unroll
               for (int u = 0; u < 4; u++) {
                  i32 pred = 0;
#pragma unroll
                  for (int j = 0; j < 16; j++) pred = add32(pred, h[u + 1 + j] * A[j]);
                  o4[u] = sk_mulwb(inv_gain_Q31, sk_sat16(sk_rround(sub32(shl32(h[u], 12), pred), 12)));
               }
#pragma unroll
               for (int u = 0; u < 4; u++) if (i0 + u < lag + 2) lt[sLTP_buf_idx - (i0 + u) - 1] = o4[u];
            }
         } else if (gain_adj_Q16 != (i32)1 << 16) {
            for (int i = 0; i < lag + 2; i++) lt[sLTP_buf_idx - i - 1] = sk_mulww(gain_adj_Q16, lt[sLTP_buf_idx - i - 1]);
         }
         b0 = c->LTPCoef_Q14[k * 5]; b1 = c->LTPCoef_Q14[k * 5 + 1]; b2 = c->LTPCoef_Q14[k * 5 + 2]; b3 = c->LTPCoef_Q14[k * 5 + 3]; b4 = c->LTPCoef_Q14[k * 5 + 4];
         pl = lt + sLTP_buf_idx - lag + 2;                                     /* tap 0 of sample i reads pl[i], tap j pl[i - j] */
         t0 = pl[-1]; t1 = pl[-2]; t2 = pl[-3]; t3 = pl[-4];
      }
      for (int i0 = 0; i0 < L; i0 += 4) {
         i32 r4[4], tn4[4], l4[4], x4[4];                                      /* this pass's reads first (the long-term taps lie at least 14 samples back), its writes last */
#pragma unroll
         for (int u = 0; u < 4; u++) { r4[u] = pexc[i0 + u]; tn4[u] = voiced ? pl[i0 + u] : 0; }
#pragma unroll
         for (int u = 0; u < 4; u++) {
            i32 r = r4[u];
            if (voiced) {
               const i32 tn = tn4[u];
               i32 p = 2;
               p = sk_mlawb(p, tn, b0); p = sk_mlawb(p, t0, b1); p = sk_mlawb(p, t1, b2); p = sk_mlawb(p, t2, b3); p = sk_mlawb(p, t3, b4);
               t3 = t2; t2 = t1; t1 = t0; t0 = tn;
               r = r + shl32(p, 1);
               l4[u] = shl32(r, 1);
            }
            i32 pred = P >> 1;
#pragma unroll
            for (int j = 0; j < 16; j++) pred = sk_mlawb(pred, w[16 + u - 1 - j], A[j]);
            const i32 v = sk_add_sat(r, sk_shl_sat(pred, 4));
            w[16 + u] = v;
            x4[u] = sk_sat16(sk_rround(sk_mulww(v, Gain_Q10), 8));
         }
#pragma unroll
         for (int u = 0; u < 4; u++) { if (voiced) lt[sLTP_buf_idx + i0 + u] = l4[u]; pxq[i0 + u] = (i16)x4[u]; }
#pragma unroll
         for (int j = 0; j < 16; j++) w[j] = w[j + 4];
      }
      if (voiced) sLTP_buf_idx += L;
   }
#pragma unroll
   for (int j = 0; j < 16; j++) ch->sLPC_Q14_buf[j] = w[j];
   ch->prev_gain_Q16 = prev_gain;
}

/* the bookkeeping half of silk_decode_frame for a decoded frame (silk/decode_frame.c:104-143): the output history, silk_PLC's update branch (PLC.c:77), silk_CNG's
 * no-loss half (CNG.c:79-127), silk_PLC_glue_frames (PLC.c:441) */
template <class CH> WV_DEV void sl_decode_frame_back(CH ch, SdCtrl *c, LnI16 pOut, LnI16 outBuf, LnI32 exc, LnI32 cng, const i32 *cng_exc)
{
   const int L = ch->frame_length, mv = ch->ltp_mem_length - L;
   sl_copy(outBuf, outBuf + L, mv);
   sl_copy(outBuf + mv, pOut, L);
   if (ch->fs_kHz != ch->plc_fs_kHz) { sd_plc_reset(ch); ch->plc_fs_kHz = ch->fs_kHz; }
   sd_plc_update(ch, c);
   ch->lossCnt = 0;
   ch->prevSignalType = ch->indices.signalType;
   ch->first_frame_after_reset = 0;
   if (ch->fs_kHz != ch->cng_fs_kHz) { sd_cng_reset(ch); ch->cng_fs_kHz = ch->fs_kHz; }
   if (ch->prevSignalType == SD_TYPE_NO_VOICE) {
      for (int i = 0; i < ch->LPC_order; i++) ch->cng_smth_NLSF_Q15[i] = (i16)(ch->cng_smth_NLSF_Q15[i] + sk_mulwb((i32)ch->prevNLSF_Q15[i] - (i32)ch->cng_smth_NLSF_Q15[i], 16348));
      i32 max_Gain_Q16 = 0; int subfr = 0;
      for (int i = 0; i < ch->nb_subfr; i++) if (c->Gains_Q16[i] > max_Gain_Q16) { max_Gain_Q16 = c->Gains_Q16[i]; subfr = i; }
      if (!ch->cng_loaded) { sl_copy(cng, cng_exc, 320); ch->cng_loaded = 1; }
      const int SL = ch->subfr_length;
      {                                                                        /* memmove up by one subframe, from the top, eight at a time */
         int i = (ch->nb_subfr - 1) * SL;
         for (; i >= 8; i -= 8) {
            i32 t[8];
#pragma unroll
            for (int k = 0; k < 8; k++) t[k] = cng[i - 8 + k];
#pragma unroll
            for (int k = 0; k < 8; k++) cng[SL + i - 8 + k] = t[k];
         }
         for (i--; i >= 0; i--) cng[SL + i] = cng[i];
      }
      sl_copy(cng, exc + subfr * SL, SL);
      for (int i = 0; i < ch->nb_subfr; i++) {
         ch->cng_smth_Gain_Q16 += sk_mulwb(c->Gains_Q16[i] - ch->cng_smth_Gain_Q16, 4634);
         if (sk_mulww(ch->cng_smth_Gain_Q16, 46396) > c->Gains_Q16[i]) ch->cng_smth_Gain_Q16 = c->Gains_Q16[i];
      }
   }
   for (int i = 0; i < ch->LPC_order; i++) ch->cng_synth_state[i] = 0;
   sd_plc_glue_frames(ch, pOut, L);
   ch->lagPrev = c->pitchL[ch->nb_subfr - 1];
}

/* One packet of one stream on this lane.  Returns 1 when the stream record has been updated (and *nsamples_out / *rng_out written), 0 when the packet has to go to the
 * general kernel (a redundant CELT frame follows the SILK data): nothing but the PCM slot has been written then.  A HYBRID packet (the look sends those whose stream's last
 * packet was hybrid too): the SILK layer is decoded and committed here, the range decoder -- behind the redundancy flag, src/opus_decoder.c:503 -- is parked in *hyb_ec, and
 * the return value 2 asks for oa_decode_hyb_kernel (the CELT layer on top of this lane's PCM; it writes the scalars, the sample count and the final range).
 * work: this lane's base in the tile's work area (the tile's base + lane, see SL_WORK_BYTES); ring: the tile's resampler ring in LDS. */
WV_DEVN int oa_sdec_lane_packet(OaDecStream *gs, const u8 *data, int len, i16 *pcm_out, i32 *nsamples_out, u32 *rng_out, OaHybCont *hyb, char *tile_work, WV_LDS ResamplerLds *ring, const WV_LDS SlTabs *tabs, const WV_LDS u8 *win, const int lane)
{
   const int CC = gs->s.channels, Fs = gs->s.Fs ? gs->s.Fs : 48000;
   const int toc = data[0];
   const int nch = (toc & 0x4) ? 2 : 1;
   const int hybrid = (toc & 0x60) == 0x60;
   const int bandwidth = hybrid ? ((toc & 0x10) ? 1105 : 1104) : 1101 + ((toc >> 5) & 0x3);
   const int audiosize = oa_samples_per_frame(toc, Fs);
   const int internalRate = bandwidth == 1101 ? 8000 : bandwidth == 1102 ? 12000 : 16000, fs_kHz = internalRate / 1000;
   const int payload_ms = imax(10, 1000 * audiosize / Fs);
   int off = 1, flen = len - 1;
   if (toc & 3) {                                                   /* a code-3 packet with one frame (what a CBR encoder's padding makes of a packet): opus_packet_parse_impl, src/opus.c:224 */
      i32 size[48];
      if (oa_packet_parse(data, len, size, &off) != 1 || size[0] <= 1) return 0;
      flen = size[0];
   }
   SlPkt buf; buf.g = data + off; buf.w = win + lane * SL_WIN_STRIDE; buf.sh = (u32)(off - 1);

   i16 *r16 = (i16 *)tile_work + lane;
   i32 *r32 = (i32 *)(tile_work + (size_t)SL_ROWS16 * SL_STREAMS * 2) + lane;
   i32 *lt = (i32 *)(tile_work + (size_t)SL_ROWS16 * SL_STREAMS * 2 + (size_t)SL_ROWS32 * SL_STREAMS * 4) + (size_t)lane * SL_LTP_WORDS;
   LnI16 outBuf[2], xq[2], pulses; LnI32 exc[2], cng[2];
   outBuf[0].p = r16; outBuf[1].p = r16 + 480 * SL_STREAMS; xq[0].p = r16 + 960 * SL_STREAMS; xq[1].p = r16 + (960 + 324) * SL_STREAMS; pulses.p = r16 + (960 + 648) * SL_STREAMS;
   exc[0].p = r32; exc[1].p = r32 + 320 * SL_STREAMS; cng[0].p = r32 + 640 * SL_STREAMS; cng[1].p = r32 + 960 * SL_STREAMS;

   SlCh cs[2]; SlDec sd; SdCtrl ctrl;
   P4_TIC();
   for (int n = 0; n < nch; n++) sl_load_channel(&cs[n], &gs->silk.ch[n], outBuf[n]);
   sd.pred_prev_Q13[0] = gs->silk.pred_prev_Q13[0]; sd.pred_prev_Q13[1] = gs->silk.pred_prev_Q13[1];
   sd.sMid[0] = gs->silk.sMid[0]; sd.sMid[1] = gs->silk.sMid[1]; sd.sSide[0] = gs->silk.sSide[0]; sd.sSide[1] = gs->silk.sSide[1];
   sd.prev_decode_only_middle = gs->silk.prev_decode_only_middle;
   OaResamplerCfg rc;
   {
      const i32 *g = gs->silk.ch[0].rs_cfg;
      rc.resampler_function = g[0]; rc.batchSize = g[1]; rc.invRatio_Q16 = g[2]; rc.FIR_Order = g[3]; rc.FIR_Fracs = g[4]; rc.Fs_in_kHz = g[5]; rc.Fs_out_kHz = g[6]; rc.inputDelay = g[7]; rc.coefs_id = g[8];
   }
   EcCtx ec_; EcCtx *e = &ec_;
   k_ec_dec_init(e, buf, (u32)flen);
   P4_TOC(0);

   int decoded = 0;
   do {
      /* ---- silk_Decode for one 10 / 20 ms frame of every internal channel (silk/dec_API.c:142; silk_decode_wave is the commented wave form) ---- */
      int decode_only_middle = 0;
      i32 MS_pred_Q13[2] = { 0, 0 };
      if (decoded == 0) {
         for (int n = 0; n < nch; n++) {
            cs[n].nFramesDecoded = 0;
            if (payload_ms == 10) { cs[n].nFramesPerPacket = 1; cs[n].nb_subfr = 2; }
            else if (payload_ms == 20) { cs[n].nFramesPerPacket = 1; cs[n].nb_subfr = 4; }
            else if (payload_ms == 40) { cs[n].nFramesPerPacket = 2; cs[n].nb_subfr = 4; }
            else { cs[n].nFramesPerPacket = 3; cs[n].nb_subfr = 4; }
            cs[n].subfr_length = 5 * fs_kHz;                                    /* silk_decoder_set_fs with the rate unchanged (decoder_set_fs.c:35): the frame length may change */
            cs[n].frame_length = cs[n].nb_subfr * cs[n].subfr_length;
         }
         for (int n = 0; n < nch; n++) {
            for (int i = 0; i < cs[n].nFramesPerPacket; i++) cs[n].VAD_flags[i] = k_ec_dec_bit_logp(e, buf, 1);
            cs[n].LBRR_flag = k_ec_dec_bit_logp(e, buf, 1);
         }
         for (int n = 0; n < nch; n++) {
            cs[n].LBRR_flags[0] = cs[n].LBRR_flags[1] = cs[n].LBRR_flags[2] = 0;
            if (cs[n].LBRR_flag) {
               if (cs[n].nFramesPerPacket == 1) cs[n].LBRR_flags[0] = 1;
               else {
                  const int sym = k_ec_dec_icdf(e, buf, &sk_lbrr_flags_icdf[cs[n].nFramesPerPacket == 2 ? 0 : 3], 8) + 1;
                  for (int i = 0; i < cs[n].nFramesPerPacket; i++) cs[n].LBRR_flags[i] = (sym >> i) & 1;
               }
            }
         }
         for (int i = 0; i < cs[0].nFramesPerPacket; i++) {                    /* skip the LBRR payload (:254-290) */
            for (int n = 0; n < nch; n++) {
               if (cs[n].LBRR_flags[i]) {
                  if (nch == 2 && n == 0) {
                     sd_stereo_decode_pred(e, buf, MS_pred_Q13);
                     if (cs[1].LBRR_flags[i] == 0) decode_only_middle = k_ec_dec_icdf(e, buf, sk_stereo_only_code_mid_icdf, 8);
                  }
                  const int condCoding = (i > 0 && cs[n].LBRR_flags[i - 1]) ? SD_CODE_CONDITIONALLY : SD_CODE_INDEPENDENTLY;
                  sd_decode_indices(e, buf, &cs[n], i, 1, condCoding);
                  sl_decode_pulses(e, buf, pulses, cs[n].indices.signalType, cs[n].indices.quantOffsetType, cs[n].frame_length, tabs);
               }
            }
         }
      }
      if (nch == 2) {
         sd_stereo_decode_pred(e, buf, MS_pred_Q13);
         decode_only_middle = cs[1].VAD_flags[cs[0].nFramesDecoded] == 0 ? k_ec_dec_icdf(e, buf, sk_stereo_only_code_mid_icdf, 8) : 0;
      }
      if (nch == 2 && decode_only_middle == 0 && sd.prev_decode_only_middle == 1) {
         for (int i = 0; i < 480; i++) outBuf[1][i] = 0;
         for (int i = 0; i < 16; i++) cs[1].sLPC_Q14_buf[i] = 0;
         cs[1].lagPrev = 100; cs[1].LastGainIndex = 10; cs[1].prevSignalType = SD_TYPE_NO_VOICE; cs[1].first_frame_after_reset = 1;
      }
      const int has_side = !decode_only_middle;
      const int nDec = cs[0].frame_length;
      for (int n = 0; n < nch; n++) {
         SlCh *ch = &cs[n];
         if (n == 0 || has_side) {
            const int FrameIndex = cs[0].nFramesDecoded - n;
            int condCoding;
            if (FrameIndex <= 0) condCoding = SD_CODE_INDEPENDENTLY;
            else if (n > 0 && sd.prev_decode_only_middle) condCoding = SD_CODE_INDEPENDENTLY_NO_LTP_SCALING;
            else condCoding = SD_CODE_CONDITIONALLY;
            ctrl.LTP_scale_Q14 = 0; ch->exc_valid = 1;
            P4_TOC(1);
            sd_decode_indices(e, buf, ch, ch->nFramesDecoded, 0, condCoding);
            P4_TOC(2);
            sl_decode_pulses(e, buf, pulses, ch->indices.signalType, ch->indices.quantOffsetType, ch->frame_length, tabs);
            P4_TOC(3);
            sd_decode_parameters(ch, &ctrl, condCoding);
            P4_TOC(4);
            sl_decode_core(ch, &ctrl, xq[n] + 2, pulses, exc[n], outBuf[n], lt);
            P4_TOC(5);
            sl_decode_frame_back(ch, &ctrl, xq[n] + 2, outBuf[n], exc[n], cng[n], &gs->silk.cng_exc_buf_Q14[n][0]);
            P4_TOC(6);
         } else for (int i = 0; i < nDec; i++) xq[n][2 + i] = 0;
         ch->nFramesDecoded++;
      }
      if (CC == 2 && nch == 2) sd_stereo_ms_to_lr(&sd, xq[0], xq[1], MS_pred_Q13, cs[0].fs_kHz, nDec);
      else for (int i = 0; i < 2; i++) { xq[0][i] = sd.sMid[i]; sd.sMid[i] = xq[0][nDec + i]; }
      const int nOut = (nDec * Fs) / (cs[0].fs_kHz * 1000);
      P4_TOC(1);
      for (int n = 0; n < nch; n++) {
         SlPcmOut out; out.p = pcm_out + (size_t)decoded * CC + n; out.st = CC;
         silk_resampler_lane(rc, ring, cs[n].rs_rows, 1, xq[n] + 1, nDec, out, lane);
      }
      if (CC == 2 && nch == 1) {                                    /* a mono packet into a stereo decoder: both channels get the one signal (silk/dec_API.c:401) */
         i16 *pp = pcm_out + (size_t)decoded * 2;
         for (int i0 = 0; i0 < nOut; i0 += 8) {
            i32 t[8];
#pragma unroll
            for (int k = 0; k < 8; k++) t[k] = i0 + k < nOut ? pp[2 * (i0 + k)] : 0;
#pragma unroll
            for (int k = 0; k < 8; k++) if (i0 + k < nOut) pp[2 * (i0 + k) + 1] = (i16)t[k];
         }
      }
      P4_TOC(7);
      sd.prev_decode_only_middle = decode_only_middle;
      decoded += nOut;
   } while (decoded < audiosize);

   /* a SILK-only packet with 17 bits to spare carries a redundant CELT frame (src/opus_decoder.c:499-526): the general kernel's business */
   if (hybrid) { if (k_ec_tell(e, buf) + 17 + 20 <= 8 * flen && k_ec_dec_bit_logp(e, buf, 12)) return 0; }
   else if (k_ec_tell(e, buf) + 17 <= 8 * flen) return 0;

   /* ---- commit ---- */
   P4_TOC(1);
   for (int n = 0; n < nch; n++) sl_store_channel(&gs->silk.ch[n], &cs[n], outBuf[n], exc[n], cng[n], &gs->silk.cng_exc_buf_Q14[n][0]);
   gs->silk.pred_prev_Q13[0] = sd.pred_prev_Q13[0]; gs->silk.pred_prev_Q13[1] = sd.pred_prev_Q13[1];
   gs->silk.sMid[0] = sd.sMid[0]; gs->silk.sMid[1] = sd.sMid[1]; gs->silk.sSide[0] = sd.sSide[0]; gs->silk.sSide[1] = sd.sSide[1];
   gs->silk.prev_decode_only_middle = sd.prev_decode_only_middle;
   P4_TOC(8);
   if (hybrid) { hyb->ec = *e; hyb->off = off; hyb->flen = flen; return 2; }
   gs->s.mode = 1000; gs->s.bandwidth = bandwidth; gs->s.frame_size = audiosize; gs->s.stream_channels = nch;
   gs->s.start = 17; gs->s.end = bandwidth == 1101 ? 13 : 17;                    /* what opus_decode_frame leaves behind for a SILK-only frame (celt_dec_frame.h: oa_decode_frame_wave) */
   gs->s.rangeFinal = e->rng; gs->s.prev_mode = 1000; gs->s.prev_redundancy = 0; gs->s.last_packet_duration = decoded;
   *nsamples_out = decoded; *rng_out = e->rng;
   return 1;
}
#endif

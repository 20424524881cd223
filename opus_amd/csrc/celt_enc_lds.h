/* celt_enc_lds.h — per-wavefront LDS working set of the CELT frame encoder (one wave = one stream-frame).
 *
 * Big regions are phase-aliased:
 *   A (15,872 B)  prefilter: pre[2][1024+960] (history + new input)     -> afterwards: spectrum freq/X[2][960]
 *                                                                          (+ theta-RDO save slots, hadamard tmp)
 *   B ( 8,640 B)  in[2][960+120] (pre-emphasised, comb-filtered input)   -> afterwards: folding memory norm[2][800],
 *                                                                          tf_analysis scratch
 *   C ( 8,192 B)  int16 staging (dc-rejected PCM, tone/transient buffers, pitch buffers), FFT storage, PVQ scratch
 * Total ~40 KB/wave -> 4 waves per CU (160 KB LDS). */
#ifndef OPUS_AMD_CELT_ENC_LDS_H
#define OPUS_AMD_CELT_ENC_LDS_H

#define NBE OA_NB_EBANDS

struct FrameShared {
   /* frame constants derived by the Opus layer / CELT prologue */
   i32 CC, C, LM, M, N, start, end, effEnd, complexity, lsb_depth, vbr, constrained_vbr, disable_inv, disable_pf, force_intra, loss_rate;
   i32 bitrate, curr_bandwidth, toc, max_data_bytes, orig_max_data_bytes, frame_size, do_stereo_fade, fade_g1, fade_g2;
   i32 nbCompressedBytes, nbFilledBytes, nbAvailableBytes, effectiveBytes, vbr_rate, total_bits, equiv_rate, tell, tell0_frac;
   i32 silence, sample_max, skip_celt, ret, plc_frame;
   /* analysis results */
   i32 tone_freq, toneishness, isTransient, tf_estimate, tf_chan, weak_transient, shortBlocks, transient_got_disabled, secondMdct;
   i32 pf_on, pitch_index, gain1, qg, prefilter_tapset, pitch_change, pf_enabled, cancel_pitch;
   i32 maxDepth, tot_boost, temporal_vbr, tf_select, enable_tf_analysis, do_patch;
   i32 alloc_trim, dual_stereo, total_boost, anti_collapse_rsv, anti_collapse_on, codedBands, balance, bits, signalBandwidth, pvq_total_bits;
   i32 r[24];     /* small hand-off slots between lane-0 sections and parallel code */
};

struct PvqScratch {                  /* lives in region A behind the spectrum */
   i32 lowband_scratch[176], X_save[176], Y_save[176], X_save2[176], Y_save2[176], norm_save2[176], hada_tmp[176];
   i32 iy[176 + 8];
   i32 ysearch[176 + 8];
};

struct FrameLds {
   EcCtx ec;
   EcCtx ecsave[4];
   FrameShared sh;
   OaEncScalars st;
   i32 bandE[2 * NBE], bandLogE[2 * NBE], bandLogE2[2 * NBE], error[2 * NBE];
   i32 oldBandE[2 * NBE], oldLogE[2 * NBE], oldLogE2[2 * NBE], energyError[2 * NBE];
   i32 offsets[NBE], importance[NBE], spread_weight[NBE], tf_res[NBE], pulses[NBE], fine_quant[NBE], fine_priority[NBE], cap[NBE];
   i32 in_mem[2 * OA_OVERLAP];
   i32 scr[6 * 2 * NBE];              /* lane-0 scratch (allocation vectors, dynalloc followers, two-pass energies) */
   i32 aux[32];                       /* MDCT headroom/shift bookkeeping, reductions hand-off */
   u8 collapse_masks[2 * NBE + 6];
   u8 packet[OA_MAX_PACKET + 4];      /* packet[0] = TOC, range coder buffer = packet+1 */
   u8 bytes_save[OA_MAX_PACKET + 4];
   union {
      i32 pre[2][OA_MAX_PERIOD + OA_MAX_FRAME];
      struct { i32 X[2 * OA_MAX_FRAME]; PvqScratch pvq; } s;
   } A;
   union {
      i32 in[2][OA_MAX_FRAME + OA_OVERLAP];
      struct { i32 norm[2 * 800]; i32 tf_tmp[560]; } s;
   } B;
   union {
      i16 pcm16[2 * OA_MAX_FRAME];                                     /* dc-rejected input, interleaved */
      i16 x16[2][OA_MAX_FRAME + OA_OVERLAP + 8];                        /* tone detector / transient detector */
      struct { i16 pitch_buf[992 + 8]; i16 x_lp4[240 + 8]; i16 y_lp4[496 + 8]; i32 xcorr[488 + 8]; i32 yy_lookup[514 + 6]; } p;
      i32 fft[OA_MAX_FRAME];                                           /* N/4 complex points x blocks */
   } Cc;
};
#endif

/* celt_enc_lds.h — per-wavefront LDS working set of the CELT frame encoder (one wave = one stream-frame).
 *
 * Big regions are phase-aliased:
 *   A (7,680 B)  dc-rejected int16 PCM (the unfiltered pre-emphasised signal is recomputed from it on the fly)
 *                                                      -> from the MDCT on: spectrum freq/X[2][960]
 *   B (8,640 B)  in[2][960+120] (pre-emphasised, comb-filtered input)   -> afterwards: folding memory norm[2][800],
 *                                                      tf_analysis scratch, theta-RDO byte snapshot
 *   C (5,664 B)  tone/transient int16 buffers | pitch buffers | FFT storage | PVQ save slots
 * The 2x1024-sample pitch history (prefilter_mem) is NOT copied to LDS: it is read from the stream's HBM state where
 * needed (coalesced) and rewritten in place.  Total ~26 KB/wave -> 6 waves per CU (160 KB LDS), 2 per SIMD with <=256 VGPRs. */
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

struct PvqScratch {                  /* region C during the PVQ phase */
   i32 lowband_scratch[176], X_save[176], Y_save[176], X_save2[176], Y_save2[176], norm_save2[176], hada_tmp[176];
   i32 iy[176 + 8];
};

struct FrameLds {
   EcCtx ec;
   EcCtx ecsave[2];
   FrameShared sh;
   OaEncScalars st;
   i32 bandE[2 * NBE], bandLogE[2 * NBE], bandLogE2[2 * NBE], error[2 * NBE];
   i32 oldBandE[2 * NBE], energyError[2 * NBE];
   i32 offsets[NBE], importance[NBE], spread_weight[NBE], tf_res[NBE], pulses[NBE], fine_quant[NBE], fine_priority[NBE], cap[NBE];
   i32 scr[6 * NBE];                  /* lane-0 scratch (allocation vectors, dynalloc followers, two-pass energies) */
   i32 aux[32];                       /* MDCT headroom/shift bookkeeping, reductions hand-off */
   u32 prof[24];                      /* shader-clock buckets, only written by the -DOA_PHASE_TIMERS profiling build */
   u8 collapse_masks[2 * NBE + 6];
   u8 packet[OA_MAX_PACKET + 4];      /* packet[0] = TOC, range coder buffer = packet+1 */
   union {                            /* A: int16 input until the MDCT, then the spectrum */
      i16 pcm16[2 * OA_MAX_FRAME];                                     /* dc-rejected input, interleaved */
      struct { i32 X[2 * OA_MAX_FRAME]; } s;
   } A;
   union {                            /* B: filtered time signal until the MDCT, then folding memory / rollback bytes */
      i32 in[2][OA_MAX_FRAME + OA_OVERLAP];
      struct { i32 norm[2 * 800]; u8 bytes_save[OA_MAX_PACKET + 4]; } s;
   } B;
   union {                            /* C: phase scratch */
      i16 x16[2][OA_MAX_FRAME + OA_OVERLAP + 8];                        /* tone detector / transient detector */
      struct { i16 pitch_buf[992 + 8]; i16 x_lp4[240 + 8]; i16 y_lp4[496 + 8]; union { i32 xcorr[488 + 8]; i32 yy_lookup[514 + 6]; } u; } p;
      i32 fft[OA_MAX_FRAME];                                           /* N/4 complex points x blocks */
      PvqScratch pvq;
   } Cc;
};
#endif

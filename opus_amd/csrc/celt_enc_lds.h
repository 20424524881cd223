/* celt_enc_lds.h — per-wavefront working set of the CELT frame encoder (one wave = one stream-frame).
 *
 * LDS holds what lane-0 serial code and the in-place transforms touch; the bulk arrays that only lane-parallel, coalesced passes touch live in a
 * per-stream HBM scratch record (CeltScratch, L2/MALL-resident while the frame is in flight):
 *   HBM  pcm16[2][960]   dc-rejected int16 input (the unfiltered pre-emphasised signal is recomputed from it on the fly)
 *        in[2][960]      comb-filtered new input, read once by every MDCT of the frame
 *        X[2][960]       spectrum: freq, normalised in place; the PVQ stages one band at a time into LDS
 *        theta-RDO save slots of the second trial
 *   LDS  one phase-aliased region BC (7,136 B): dc_reject staging | tone/transient int16 buffers | pitch buffers | MDCT work buffer of ONE channel
 *        (FFT in place, band energies taken before the channel is written out) | tf_analysis trial buffers | coarse-energy rollback |
 *        both channels' coded bins [2][800] for the spreading / stereo / trim analyses | PVQ: band staging + folding memory norm[624] (+ norm2 for
 *        dual stereo) + band scratch
 * Also not in LDS: the 2x1024-sample pitch history and the 2x120 overlap memory (read from the stream's HBM record where needed, rewritten in place)
 * and the theta-RDO byte journal (the stream's still-unwritten output slot).
 * Total = 10,240 B/wave (8 allocation granules of 1,280 B) -> 16 waves per CU (160 KB LDS) = 4 per SIMD, matched by __launch_bounds__(64, 4) (<= 128 VGPRs). */
#ifndef OPUS_AMD_CELT_ENC_LDS_H
#define OPUS_AMD_CELT_ENC_LDS_H

#define NBE OA_NB_EBANDS
#include "opus_multiframe.h"
#ifndef K_TIC            /* shader-clock section timers exist only in the -DOA_PHASE_TIMERS profiling build */
#define K_TIC()
#define K_TOC(bucket)
#endif
#ifndef AN_TIC           /* ... and so do the sections that add straight to the global totals */
#define AN_TIC()
#define AN_TOC(bucket)
#endif

struct FrameShared {
   /* frame constants derived by the Opus layer / CELT prologue */
   i32 CC, C, LM, M, N, start, end, effEnd, complexity, lsb_depth, vbr, constrained_vbr, disable_inv, disable_pf, force_intra, loss_rate;
   i32 bitrate, curr_bandwidth, toc, max_data_bytes, orig_max_data_bytes, frame_size, do_stereo_fade, fade_g1, fade_g2;
   i32 nbCompressedBytes, nbFilledBytes, nbAvailableBytes, effectiveBytes, vbr_rate, total_bits, equiv_rate, tell, tell0_frac;
   i32 silence, sample_max, skip_celt, ret, plc_frame, pad_to;
   /* analysis results */
   i32 tone_freq, toneishness, isTransient, tf_estimate, tf_chan, weak_transient, shortBlocks, transient_got_disabled, secondMdct;
   i32 pf_on, pitch_index, gain1, qg, prefilter_tapset, pitch_change, pf_enabled, cancel_pitch;
   i32 maxDepth, tot_boost, temporal_vbr, tf_select, enable_tf_analysis, do_patch;
   i32 alloc_trim, dual_stereo, total_boost, anti_collapse_rsv, anti_collapse_on, codedBands, balance, bits, signalBandwidth, pvq_total_bits;
   i32 silk_signalType, silk_offset;   /* hybrid: SILKInfo of the frame (celt/celt.h SILKInfo, src/opus_encoder.c:2486) */
   i32 upsample;                       /* 48000 / API rate: the input is zero-stuffed up to 48 kHz (celt_encoder.c:255, :557, :544); 0 = 1 */
   i32 raw_frame;                      /* plain celt_encode_with_ec (redundancy / prefill frames): no TOC, no Opus-layer finalisation */
   /* the call (opus_encode_native) of the CELT-only applications */
   i32 Fs, call_bitrate, call_max_data_bytes, call_equiv_rate, cbr_bytes, nb_frames, enc_frame_size, repacketize_len, max_len_sum, is_silence, activity, no_pad;
   i32 use_dtx, nb_no_activity_ms_Q1, peak_signal_energy, prev_framesize, lfe, energy_mask_on;        /* the tail of the stream record, staged */
   i32 surround_masking, surround_trim;   /* what the surround masks of the multistream layer contribute to the VBR target / the allocation trim (celt_encoder.c:2112-2186) */
   i32 r[8];      /* small hand-off slots between lane-0 sections and parallel code */
};

#define OA_NORM_LEN 624              /* 8 * eBands[20]: folding memory never extends into the last band */
#define OA_MAX_BAND 176              /* widest band: 8 * (eBands[21] - eBands[20]) */
#define OA_CODED_BINS 800            /* 8 * eBands[21] */

struct CeltScratch {                 /* per-stream HBM scratch of one frame in flight (nothing in it survives the frame) */
   i16 pcm16[2 * OA_MAX_FRAME];
   i32 in[2][OA_MAX_FRAME];
   i32 X[2 * OA_MAX_FRAME];
   i32 X_save2[OA_MAX_BAND], Y_save2[OA_MAX_BAND], norm_save2[OA_MAX_BAND];
};

struct FrameLds {
   EcCtx ec;
   MfLds mf;                          /* multi-frame packet assembly (opus_multiframe.h) */
   union {                            /* (anonymous unions: members whose lifetimes never overlap share their bytes, the names stay) */
      EcCtx ecsave[2];                /* theta-RDO coder snapshots (PVQ phase) */
      i32 aux[32];                    /* MDCT headroom/shift bookkeeping (MDCT phase) */
   };
   FrameShared sh;
   OaEncScalars st;
   CeltScratch *g;                    /* this frame's HBM scratch */
   i32 bandE[2 * NBE], bandLogE[2 * NBE], oldBandE[2 * NBE];
   union {
      i32 bandLogE2[2 * NBE];         /* second-MDCT energies: last read by dynalloc_analysis */
      i32 error[2 * NBE];             /* coarse-energy residual: first written by the coarse quantiser, after dynalloc_analysis */
   };                                 /* (energyError is not staged: one coalesced read and one coalesced write of the HBM state) */
   i32 offsets[NBE], importance[NBE], spread_weight[NBE], tf_res[NBE], pulses[NBE], fine_quant[NBE], fine_priority[NBE], cap[NBE];
   union {
      i32 surround_dynalloc[NBE];     /* surround masking boosts: last read by dynalloc_analysis */
      u8 collapse_masks[2 * NBE + 6]; /* first written by the PVQ */
   };
   i32 scr[4 * NBE];                  /* lane-0 scratch while BC is fully occupied (tf metrics, spreading counts, allocation vectors); the larger users borrow BC */
#ifdef OA_PHASE_TIMERS
   u32 prof[34], prof_t0;             /* shader-clock buckets of the profiling build, start of the open phase */
#endif
   u8 packet[OA_MAX_PACKET + 4];      /* packet[0] = TOC, range coder buffer = packet+1 */
   union {                            /* BC: phase scratch */
      i16 stage16[2 * OA_MAX_FRAME];                                   /* dc_reject: the per-channel recursion runs here, the result goes to g->pcm16 */
      i16 x16[2][OA_MAX_FRAME + OA_OVERLAP + 8];                        /* tone detector / transient detector */
      struct { i16 pitch_buf[992 + 8]; i16 x_lp4[240 + 8]; i16 y_lp4[496 + 8]; union { i32 xcorr[488 + 8]; i32 yy_lookup[514 + 6]; } u; } p;
      i32 W[OA_MAX_FRAME];                                             /* MDCT of one channel, in place */
      i32 tf[2 * OA_CODED_BINS];                                       /* tf_analysis trial buffers */
      u8 coarse_save[OA_MAX_PACKET + 4];                               /* two-pass coarse energy rollback */
      i32 xs[2][OA_CODED_BINS];                                        /* normalised coded bins of both channels: spreading_decision / stereo_analysis / alloc_trim */
      struct {
         i32 norm[OA_NORM_LEN];
         i32 Xb[OA_MAX_BAND];                                          /* the band being coded (channel 0 / the mid) */
         union { i32 norm2[OA_NORM_LEN]; i32 Yb[OA_MAX_BAND]; } u;     /* dual stereo: second folding memory; otherwise the band of channel 1 (never both: after the switch at the intensity band norm2 is dead) */
         i32 lowband_scratch[OA_MAX_BAND];
#ifdef K_DUMP_ENABLED
         i32 iy[OA_MAX_BAND + 8];                                      /* stage dumps of the test build only */
#endif
      } q;
   } BC;
};
#include <stddef.h>
/* what the front kernel hands over and the back kernel takes up again (one record per stream, HBM) */
struct alignas(16) CeltCont {
   i32 state;                                  /* 0: the front kernel finished the call itself; 1: cut before the PVQ */
   i32 sort_key;                               /* what the PVQ kernel's order is sorted by (opus_amd.hip: oa_cut_key) */
   i32 pad_[2];
   i32 image[(offsetof(FrameLds, BC) + 3) / 4];   /* the front wave's LDS up to the phase scratch: coder, frame constants, band arrays, packet */
   i32 X[2][OA_CODED_BINS];                    /* the normalised spectrum, coded bins of each channel */
   i32 norm[2][OA_NORM_LEN];                   /* folding memory (norm, norm2) */
   i32 norm_alt[2][OA_MAX_BAND];               /* the two theta-RDO trials' folding output of the band in flight */
   u8 alt[OA_MAX_PACKET + 4];                  /* the second theta-RDO trial codes into this buffer */
};

#endif

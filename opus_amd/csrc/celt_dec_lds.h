/* celt_dec_lds.h — per-wavefront LDS working set of the decoder (one wave = one stream; the frames of a multi-frame packet are decoded one after the other by the same wave).
 *   BC (8,640 B)  PVQ phase: folding memory norm[2][624], pulse vector, the band being decoded (X / Y staging + lowband scratch); synthesis phase: syn[2][960+120]
 *   A  (7,680 B)  LAST: scratch of the concealment's pitch mode and, together with BC, the SILK decoder's arena (SilkLdsAll at &BC).  The CELT-only fast kernel
 *                 (oa_decode_fast_kernel: packets whose decode touches neither) allocates LDS only up to here.
 * The decoded spectrum X[2][960] itself (normalised, then denormalised in place = the IMDCT's input) lives in a per-wave HBM scratch (Xg): every pass over it is
 * lane-parallel and coalesced (band copy-out, anti-collapse through the staging buffer, denormalise, the IMDCT's pre-rotation reads).
 * packet: the current frame's bytes (range decoder input). */
#ifndef OPUS_AMD_CELT_DEC_LDS_H
#define OPUS_AMD_CELT_DEC_LDS_H
#include <stddef.h>
struct DecShared {
   i32 CC, C, LM, M, N, start, end, effEnd, disable_inv, len, total_bits, silence, ret;
   i32 postfilter_pitch, postfilter_gain, postfilter_tapset, isTransient, shortBlocks, intra_ener, spread, alloc_trim, intensity, dual_stereo;
   i32 anti_collapse_rsv, anti_collapse_on, codedBands, balance, pvq_total_bits;
   /* packet level */
   i32 count, frame_bytes_off, nb_samples, packet_frame_size, max_frame;
   i32 size[48];
   i32 r[8];
};
struct DecLds {
   EcCtx ec;
   EcCtx ec_silk;                     /* the range decoder after the SILK part of a hybrid frame (the CELT part continues from it) */
   DecShared sh;
   OaDecScalars st;
   i32 oldBandE[2 * NBE], oldLogE[2 * NBE], oldLogE2[2 * NBE], backgroundLogE[2 * NBE];
   i32 tf_res[NBE], pulses[NBE], fine_quant[NBE], fine_priority[NBE], cap[NBE], offsets[NBE];
   i32 scr[6 * NBE];
   i32 aux[32];
   u8 collapse_masks[2 * NBE + 6];
   u8 packet[OA_MAX_PACKET + 4];      /* frame bytes at packet + 1 (same convention as the encoder's EC macros) */
   i32 *Xg;                           /* the spectrum X[2][OA_MAX_FRAME] of the frame in flight: per-wave HBM scratch (set by the kernel) */
   union {
      struct { i32 norm[2 * OA_NORM_LEN]; i32 iy[176 + 8]; i32 xb[176], yb[176], lbs[176]; } q;      /* xb / yb: the band being decoded, copied out to Xg when it is done; lbs: quant_band's lowband scratch (the reference borrows the last band of X, bands.c:1642) */
      i32 syn[2][OA_MAX_FRAME + OA_OVERLAP];
   } BC;
   union { i32 w[2 * OA_MAX_FRAME]; } A;
};
#define OA_DEC_FAST_LDS_BYTES (offsetof(DecLds, A))
#define OA_DEC_SCRATCH_BYTES (2 * OA_MAX_FRAME * sizeof(i32))
#endif

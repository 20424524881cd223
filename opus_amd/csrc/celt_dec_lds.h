/* celt_dec_lds.h — per-wavefront LDS working set of the decoder (one wave = one stream; the frames of a multi-frame packet are decoded one after the other by the same wave).
 *   BC (8,640 B)  PVQ phase: pulse vector, the band being decoded (X / Y staging), its folding source and its contribution to the folding memory (3,552 B); synthesis phase: syn[2][960+120].
 *                 The CELT-only fast kernel (oa_decode_fast_kernel) synthesises the channels one after the other in syn[0] and allocates LDS only up to the end of it (4,320 B of BC).
 *   A  (7,680 B)  LAST: scratch of the concealment's pitch mode and, together with BC, the SILK decoder's arena (SilkLdsAll at &BC): the general kernel only.
 * The decoded spectrum X[2][960] itself (normalised, then denormalised in place = the IMDCT's input) and the folding memory norm[2][624] live in a per-wave HBM scratch (Xg): every
 * pass over them is lane-parallel and coalesced (band copy-out, folding source / contribution of a band, anti-collapse through the staging buffer, denormalise, the IMDCT's
 * pre-rotation reads).
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
      /* xb / yb: the band being decoded, copied out to Xg when it is done; lbs: the band's folding source, staged from the folding memory in the HBM scratch -- also quant_band's lowband
       * scratch (the reference borrows the last band of X, bands.c:1642); lbo: the band's contribution to the folding memory (quant_band's lowband_out), copied out when it is complete */
      struct { i32 iy[176 + 8]; i32 xb[176], yb[176], lbs[176], lbo[176]; } q;
      i32 syn[2][OA_MAX_FRAME + OA_OVERLAP];     /* the fast kernel synthesises one channel at a time in syn[0] and allocates no further */
   } BC;
   union { i32 w[2 * OA_MAX_FRAME]; } A;
};
#define OA_DEC_FAST_LDS_BYTES (offsetof(DecLds, BC) + (OA_MAX_FRAME + OA_OVERLAP) * sizeof(i32))
static_assert(sizeof(((DecLds *)0)->BC.q) <= (OA_MAX_FRAME + OA_OVERLAP) * sizeof(i32), "the PVQ phase fits the fast kernel's LDS");
/* per resident wave in HBM: the spectrum X[2][OA_MAX_FRAME] of the frame in flight, then the folding memory norm[2][OA_NORM_LEN] (celt_dec_bands.h) */
#define OA_DEC_SCRATCH_BYTES ((2 * OA_MAX_FRAME + 2 * OA_NORM_LEN) * sizeof(i32))
#endif

/* celt_dec_lds.h — per-wavefront LDS working set of the CELT frame decoder (one wave = one stream; the frames of a
 * multi-frame packet are decoded one after the other by the same wave).
 *   A  (7,680 B)  decoded normalised spectrum X[2][960] -> denormalised in place (freq) -> int16 PCM staging
 *   BC (8,640 B)  PVQ phase: folding memory norm[2][624] + pulse vector; synthesis phase: syn[2][960+120]
 * packet: the current frame's bytes (range decoder input). */
#ifndef OPUS_AMD_CELT_DEC_LDS_H
#define OPUS_AMD_CELT_DEC_LDS_H
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
   union { i32 X[2 * OA_MAX_FRAME]; i16 pcm16[2 * OA_MAX_FRAME]; } A;
   union {
      struct { i32 norm[2 * OA_NORM_LEN]; i32 iy[176 + 8]; } q;
      i32 syn[2][OA_MAX_FRAME + OA_OVERLAP];
   } BC;
};
#endif

/* silk_frame.h — data formats of the batched SILK quantiser kernels (shared by host code, device code and the CPU emulator).
 *
 * Boundary formats mirror the reference exactly so a caller can hand over what it already has:
 *   OaNsqFrame    = the argument list of silk_NSQ_c / silk_NSQ_del_dec_c (silk/NSQ.c:76-93), one record per stream-frame
 *   OaNsqRefState = silk_nsq_state (silk/structs.h:56-69), used only by import/export
 *   OaNsqCfg      = the six silk_encoder_state fields the quantisers read (silk/structs.h:167-207)
 * Device-resident state is tile-SoA: streams are grouped in tiles of T (64 for the plain quantiser = one lane per stream, 16 for
 * delayed decision = one quad of lanes per stream, one lane per survivor); inside a tile every array is [row][T], so the lanes of
 * a wave read/write one row with a single coalesced access.  The 20 ms signal histories (xq, sLTP_shp) are rings of
 * ltp_mem+frame rows: a frame step appends instead of memmove-ing 3.8 KB per stream (silk/NSQ.c:176-178). */
#ifndef OPUS_AMD_SILK_FRAME_H
#define OPUS_AMD_SILK_FRAME_H
#include <stdint.h>

#define OA_SILK_MAX_FRAME   320
#define OA_SILK_MAX_SUBFR   80
#define OA_SILK_LPC_ORDER   16
#define OA_SILK_SHAPE_ORDER 24
#define OA_SILK_LTP_ORDER   5
#define OA_SILK_TYPE_VOICED 2
#define OA_SILK_DD          40        /* DECISION_DELAY */
#define OA_SILK_HIST_ROWS   (2 * OA_SILK_MAX_FRAME)

struct OaNsqCfg { int32_t fs_kHz, nb_subfr, predictLPCOrder, shapingLPCOrder, nStatesDelayedDecision, warping_Q16; };

struct OaNsqFrame {
   int8_t  signalType, quantOffsetType, NLSFInterpCoef_Q2, Seed;
   int16_t PredCoef_Q12[2 * 16];
   int16_t LTPCoef_Q14[20];
   int16_t AR_Q13[4 * 24];
   int32_t HarmShapeGain_Q14[4], Tilt_Q14[4], LF_shp_Q14[4], Gains_Q16[4], pitchL[4];
   int32_t Lambda_Q10, LTP_scale_Q14;
};

struct OaNsqRefState {
   int16_t xq[2 * OA_SILK_MAX_FRAME];
   int32_t sLTP_shp_Q14[2 * OA_SILK_MAX_FRAME];
   int32_t sLPC_Q14[OA_SILK_MAX_SUBFR + 16];
   int32_t sAR2_Q14[24];
   int32_t sLF_AR_shp_Q14, sDiff_shp_Q14, lagPrev, sLTP_buf_idx, sLTP_shp_buf_idx, rand_seed, prev_gain_Q16, rewhite_flag;
};

/* rows of the per-tile scalar block */
enum { OA_NSQ_S_LPC = 0, OA_NSQ_S_AR2 = 16, OA_NSQ_S_LF_AR = 40, OA_NSQ_S_DIFF, OA_NSQ_S_LAGPREV, OA_NSQ_S_PREVGAIN, OA_NSQ_S_RANDSEED,
       OA_NSQ_S_BASE, OA_NSQ_S_ROWS = 48 };

/* per-tile storage, in int32 words: [hist shp: rows*T][q15 scratch: rows*T][scal: 48*T][xq hist (i16): rows*T/2][whitened (i16): rows*T/2]
 * [delayed-decision rings: 5*40*64 (T == 16 only)] */
static inline int64_t oa_nsq_tile_words(int T)
{ return (int64_t)OA_SILK_HIST_ROWS * T * 2 + OA_NSQ_S_ROWS * T + OA_SILK_HIST_ROWS * T + (T == 16 ? 5 * OA_SILK_DD * 64 : 0); }
#endif

/* oracle/oc_silk.h — TEST INFRASTRUCTURE ONLY (CPU restatement of the SILK building blocks named by north_star).
 * Never linked by the product library; used by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg.
 *
 * Arithmetic primitives restate silk/macros.h:40-122 (SMULWB/SMLAWB/SMULWT/SMLAWT/SMULBB/SMLABB/SMULWW/SMLAWW in their
 * OPUS_FAST_INT64 forms, ADD_SAT32, CLZ32), silk/SigProc_FIX.h:447-584 (wrap-around adds, RSHIFT_ROUND, LIMIT, SAT16, RAND)
 * and silk/Inlines.h:93-185 (DIV32_varQ, INVERSE32_varQ).  Pinned against the compiled reference (oracle/_ref) by
 * tests/test_oracle_silk.py. */
#ifndef OC_SILK_H
#define OC_SILK_H
#include <stdint.h>
#include <string.h>

typedef int8_t s8; typedef int16_t s16; typedef int32_t s32; typedef int64_t s64; typedef uint32_t u32;

static inline s32 q_mulwb(s32 a, s32 b)           { return (s32)(((s64)a * (s16)b) >> 16); }                 /* macros.h:44 */
static inline s32 q_mlawb(s32 acc, s32 a, s32 b)  { return (s32)((u32)acc + (u32)q_mulwb(a, b)); }            /* macros.h:52 (the add wraps like the C int add it restates; values never overflow in valid use) */
static inline s32 q_mulwt(s32 a, s32 b)           { return (s32)(((s64)a * (b >> 16)) >> 16); }              /* macros.h:59 */
static inline s32 q_mlawt(s32 acc, s32 a, s32 b)  { return (s32)((u32)acc + (u32)q_mulwt(a, b)); }            /* macros.h:66 */
static inline s32 q_mulbb(s32 a, s32 b)           { return (s32)(s16)a * (s32)(s16)b; }                      /* macros.h:72 */
static inline s32 q_mlabb(s32 acc, s32 a, s32 b)  { return (s32)((u32)acc + (u32)q_mulbb(a, b)); }            /* macros.h:75 / SigProc_FIX.h SMLABB_ovflw */
static inline s32 q_mulww(s32 a, s32 b)           { return (s32)(((s64)a * b) >> 16); }                      /* macros.h:91 */
static inline s32 q_mlaww(s32 acc, s32 a, s32 b)  { return (s32)((u32)acc + (u32)q_mulww(a, b)); }            /* macros.h:99 */
static inline s32 q_smmul(s32 a, s32 b)           { return (s32)(((s64)a * b) >> 32); }                      /* SigProc_FIX.h silk_SMMUL */
static inline s32 q_addw(s32 a, s32 b)            { return (s32)((u32)a + (u32)b); }                         /* SigProc_FIX.h:447 */
static inline s32 q_subw(s32 a, s32 b)            { return (s32)((u32)a - (u32)b); }                         /* SigProc_FIX.h:450 */
static inline s32 q_shlw(s32 a, int s)            { return (s32)((u32)a << s); }                             /* SigProc_FIX.h:513 */
static inline s32 q_add_sat(s32 a, s32 b)         { s64 r = (s64)a + b; return r > 2147483647 ? 2147483647 : r < -2147483647 - 1 ? -2147483647 - 1 : (s32)r; } /* macros.h:105 */
static inline s32 q_rshift_round(s32 a, int s)    { return s == 1 ? (a >> 1) + (a & 1) : ((a >> (s - 1)) + 1) >> 1; }   /* SigProc_FIX.h RSHIFT_ROUND */
static inline s32 q_sat16(s32 a)                  { return a > 32767 ? 32767 : a < -32768 ? -32768 : a; }
static inline s32 q_limit(s32 a, s32 lo, s32 hi)  { return a > hi ? hi : a < lo ? lo : a; }                  /* lo <= hi form of silk_LIMIT */
static inline s32 q_rand(s32 seed)                { return (s32)(907633515u + (u32)seed * 196314165u); }      /* SigProc_FIX.h silk_RAND */
static inline int q_clz32(s32 x)                  { return x ? __builtin_clz((u32)x) : 32; }                 /* macros.h:122 */
static inline s32 q_abs(s32 a)                    { return a > 0 ? a : (s32)(0u - (u32)a); }
static inline s32 q_shl_sat(s32 a, int s)         { s32 lo = (-2147483647 - 1) >> s, hi = 2147483647 >> s; return (s32)((u32)q_limit(a, lo, hi) << s); } /* SigProc_FIX.h:510 */

s32 oc_silk_div32_varQ(s32 a32, s32 b32, int Qres);       /* silk/Inlines.h:93  */
s32 oc_silk_inverse32_varQ(s32 b32, int Qres);            /* silk/Inlines.h:143 */

/* ---- data formats (shared by the restatement, the emulator build of the kernel and the product's C ABI) ---- */
#define OC_SILK_MAX_FRAME     320      /* silk/define.h MAX_FRAME_LENGTH (20 ms @ 16 kHz) */
#define OC_SILK_MAX_SUBFR     80
#define OC_SILK_LPC_BUF       16       /* NSQ_LPC_BUF_LENGTH */
#define OC_SILK_MAX_SHAPE     24       /* MAX_SHAPE_LPC_ORDER */
#define OC_SILK_LTP_ORDER     5
#define OC_SILK_MAX_NB_SUBFR  4
#define OC_SILK_TYPE_VOICED   2
#define OC_SILK_DECISION_DELAY 40
#define OC_SILK_MAX_DEL_DEC   4

typedef struct {                        /* the encoder-state fields silk_NSQ*_c reads (silk/structs.h:167-207) */
   s32 fs_kHz, nb_subfr, predictLPCOrder, shapingLPCOrder, nStatesDelayedDecision, warping_Q16;
} OcSilkNsqCfg;
static inline int oc_cfg_subfr(const OcSilkNsqCfg *c)   { return 5 * c->fs_kHz; }
static inline int oc_cfg_ltp_mem(const OcSilkNsqCfg *c) { return 20 * c->fs_kHz; }
static inline int oc_cfg_frame(const OcSilkNsqCfg *c)   { return c->nb_subfr * 5 * c->fs_kHz; }

typedef struct {                        /* byte-compatible with silk_nsq_state (silk/structs.h:56-69), 4,352 B */
   s16 xq[2 * OC_SILK_MAX_FRAME];
   s32 sLTP_shp_Q14[2 * OC_SILK_MAX_FRAME];
   s32 sLPC_Q14[OC_SILK_MAX_SUBFR + OC_SILK_LPC_BUF];
   s32 sAR2_Q14[OC_SILK_MAX_SHAPE];
   s32 sLF_AR_shp_Q14, sDiff_shp_Q14;
   s32 lagPrev, sLTP_buf_idx, sLTP_shp_buf_idx;
   s32 rand_seed, prev_gain_Q16, rewhite_flag;
} OcSilkNsqState;

typedef struct {                        /* one frame's quantiser inputs = the argument list of silk_NSQ_c (silk/NSQ.c:76-93) */
   s8  signalType, quantOffsetType, NLSFInterpCoef_Q2, Seed;      /* the SideInfoIndices fields read (Seed is also the dither seed) */
   s16 PredCoef_Q12[2 * 16];
   s16 LTPCoef_Q14[OC_SILK_LTP_ORDER * OC_SILK_MAX_NB_SUBFR];
   s16 AR_Q13[OC_SILK_MAX_NB_SUBFR * OC_SILK_MAX_SHAPE];
   s32 HarmShapeGain_Q14[4], Tilt_Q14[4], LF_shp_Q14[4], Gains_Q16[4], pitchL[4];
   s32 Lambda_Q10, LTP_scale_Q14;
} OcSilkNsqFrame;

void oc_silk_lpc_analysis_filter(s16 *out, const s16 *in, const s16 *B, s32 len, s32 d);     /* silk/LPC_analysis_filter.c:49 */
void oc_silk_nsq(const OcSilkNsqCfg *cfg, OcSilkNsqState *st, OcSilkNsqFrame *fr, const s16 *x16, s8 *pulses);          /* silk/NSQ.c:76 */
void oc_silk_nsq_del_dec(const OcSilkNsqCfg *cfg, OcSilkNsqState *st, OcSilkNsqFrame *fr, const s16 *x16, s8 *pulses);  /* silk/NSQ_del_dec.c:114 */

/* ---- resampler (silk/resampler_structs.h:38-52; the coefficient pointer is replaced by a table id) ---- */
enum { OC_RS_FN_COPY = 0, OC_RS_FN_UP2 = 1, OC_RS_FN_IIR_FIR = 2, OC_RS_FN_DOWN_FIR = 3 };          /* resampler.c:72-75 */
enum { OC_RS_NONE = 0, OC_RS_3_4, OC_RS_2_3, OC_RS_1_2, OC_RS_1_3, OC_RS_1_4, OC_RS_1_6 };
typedef struct {
   s32 sIIR[6];
   union { s32 i32[36]; s16 i16[36]; } sFIR;
   s16 delayBuf[96];
   s32 resampler_function, batchSize, invRatio_Q16, FIR_Order, FIR_Fracs, Fs_in_kHz, Fs_out_kHz, inputDelay, coefs_id;
} OcSilkResampler;
int oc_silk_resampler_init(OcSilkResampler *S, s32 Fs_Hz_in, s32 Fs_Hz_out, int forEnc);         /* silk/resampler.c:79  */
int oc_silk_resampler(OcSilkResampler *S, s16 *out, const s16 *in, s32 inLen);                    /* silk/resampler.c:183 */

/* ---- pitch estimator (silk/fixed/pitch_analysis_core_FIX.c:82); returns 0 voiced / 1 unvoiced ---- */
void oc_silk_sum_sqr_shift(s32 *energy, int *shift, const s16 *x, int len);
void oc_silk_resampler_down2(s32 *S, s16 *out, const s16 *in, s32 inLen);
void oc_silk_resampler_down2_3(s32 *S, s16 *out, const s16 *in, s32 inLen);
s32 oc_silk_lin2log(s32 inLin);
int oc_silk_pitch_analysis_core(const s16 *frame, s32 *pitch_out, s16 *lagIndex, s8 *contourIndex, s32 *LTPCorr_Q15, s32 prevLag,
                                s32 search_thres1_Q16, s32 search_thres2_Q13, int Fs_kHz, int complexity, int nb_subfr);
#endif

/* silk_enc_all.h — the SILK encoder body in include order (after celt_enc_all.h and celt_dec_all.h: it reuses the range coder, the arithmetic, the SILK
 * tables, the pitch estimator, the resampler and the NLSF helpers of the decoder). */
#ifndef OPUS_AMD_SILK_ENC_ALL_H
#define OPUS_AMD_SILK_ENC_ALL_H
#include "silk_pitch.h"
#include "silk_enc.h"
#include "silk_enc_analysis.h"
#include "silk_enc_quant.h"
#include "silk_enc_predl.h"
#include "silk_enc_nsq.h"
#include "silk_enc_frame.h"
#include "opus_enc_sh.h"
#include "silk_nsq_dd.h"
#include "opus_sh_split.h"
#endif

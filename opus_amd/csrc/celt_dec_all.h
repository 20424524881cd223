/* celt_dec_all.h — decoder kernel body in include order (after celt_enc_all.h: it reuses the arithmetic, tables, FFT,
 * rotation and band helpers of the encoder). */
#ifndef OPUS_AMD_CELT_DEC_ALL_H
#define OPUS_AMD_CELT_DEC_ALL_H
#include "celt_dec_lds.h"
#include "celt_dec_energy.h"
#include "celt_dec_bands.h"
#include "celt_dec_pvq4.h"
#include "celt_dec_frame.h"
#endif

/* celt_enc_all.h — the kernel body in include order.  The includer provides the wave vocabulary first
 * (opus_amd/csrc/wave.h for gfx950). */
#ifndef OPUS_AMD_CELT_ENC_ALL_H
#define OPUS_AMD_CELT_ENC_ALL_H
#include "fx.h"
#include "celt_tables.h"
#include "celt_frame.h"
#include "celt_ec.h"
#include "celt_ecdec.h"
#include "celt_enc_lds.h"
#include "celt_alloc.h"
#include "celt_enc_energy.h"
#include "celt_mdct.h"
#include "opus_analysis.h"
#include "celt_enc_front.h"
#include "celt_enc_pitch.h"
#include "celt_enc_bands.h"
#include "celt_enc_pvq.h"
#include "celt_enc_pvq4.h"
#include "celt_enc_frame.h"
#endif

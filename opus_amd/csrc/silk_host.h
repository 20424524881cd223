/* silk_host.h — host-side helpers for the SILK quantiser batches: conversion between the reference's silk_nsq_state layout
 * (OaNsqRefState) and one stream's slice of a tile-SoA image held in host memory.  Used by the product library (import/export
 * through a staged tile copy) and by the CPU emulator tests. */
#ifndef OPUS_AMD_SILK_HOST_H
#define OPUS_AMD_SILK_HOST_H
#include "silk_frame.h"
#include <string.h>

struct OaNsqTileView { int32_t *shp, *q15, *scal; int16_t *xq, *wh; int32_t *ring; int T; };
static inline OaNsqTileView oa_nsq_tile_view(int32_t *tile, int T)
{
   OaNsqTileView v; const int R = OA_SILK_HIST_ROWS;
   v.shp = tile; v.q15 = tile + R * T; v.scal = tile + 2 * R * T;
   v.xq = (int16_t *)(tile + 2 * R * T + OA_NSQ_S_ROWS * T); v.wh = v.xq + R * T;
   v.ring = tile + 2 * R * T + OA_NSQ_S_ROWS * T + R * T;
   v.T = T; return v;
}
/* ring base is reset to 0 on import, so logical row == physical row */
static inline void oa_nsq_import(int32_t *tile, int T, int t, const OaNsqRefState *r, const OaNsqCfg *cfg)
{
   OaNsqTileView v = oa_nsq_tile_view(tile, T); const int mem = 20 * cfg->fs_kHz;
   for (int i = 0; i < OA_SILK_HIST_ROWS; i++) { v.shp[i * T + t] = i < mem ? r->sLTP_shp_Q14[i] : 0; v.xq[i * T + t] = i < mem ? r->xq[i] : 0; }
   for (int j = 0; j < 16; j++) v.scal[(OA_NSQ_S_LPC + j) * T + t] = r->sLPC_Q14[j];
   for (int j = 0; j < 24; j++) v.scal[(OA_NSQ_S_AR2 + j) * T + t] = r->sAR2_Q14[j];
   v.scal[OA_NSQ_S_LF_AR * T + t] = r->sLF_AR_shp_Q14;  v.scal[OA_NSQ_S_DIFF * T + t] = r->sDiff_shp_Q14;
   v.scal[OA_NSQ_S_LAGPREV * T + t] = r->lagPrev;       v.scal[OA_NSQ_S_PREVGAIN * T + t] = r->prev_gain_Q16;
   v.scal[OA_NSQ_S_RANDSEED * T + t] = r->rand_seed;    v.scal[OA_NSQ_S_BASE * T + t] = 0;
}
/* sLTP_buf_idx / sLTP_shp_buf_idx / rewhite_flag are call-local in the reference (reset at the top of every call); they are exported as 0 */
static inline void oa_nsq_export(const int32_t *tile, int T, int t, OaNsqRefState *r, const OaNsqCfg *cfg)
{
   OaNsqTileView v = oa_nsq_tile_view((int32_t *)tile, T); const int mem = 20 * cfg->fs_kHz, len = mem + cfg->nb_subfr * 5 * cfg->fs_kHz;
   const int base = v.scal[OA_NSQ_S_BASE * T + t];
   memset(r, 0, sizeof *r);
   for (int i = 0; i < mem; i++) { int p = i + base; if (p >= len) p -= len; r->sLTP_shp_Q14[i] = v.shp[p * T + t]; r->xq[i] = v.xq[p * T + t]; }
   for (int j = 0; j < 16; j++) r->sLPC_Q14[j] = v.scal[(OA_NSQ_S_LPC + j) * T + t];
   for (int j = 0; j < 24; j++) r->sAR2_Q14[j] = v.scal[(OA_NSQ_S_AR2 + j) * T + t];
   r->sLF_AR_shp_Q14 = v.scal[OA_NSQ_S_LF_AR * T + t];  r->sDiff_shp_Q14 = v.scal[OA_NSQ_S_DIFF * T + t];
   r->lagPrev = v.scal[OA_NSQ_S_LAGPREV * T + t];       r->prev_gain_Q16 = v.scal[OA_NSQ_S_PREVGAIN * T + t];
   r->rand_seed = v.scal[OA_NSQ_S_RANDSEED * T + t];
}
#endif

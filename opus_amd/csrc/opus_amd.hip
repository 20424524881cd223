/* opus_amd.hip — kernels + C ABI of the batched Opus (CELT-only) encoder for gfx950.
 * One 64-lane wavefront (= one workgroup) per (stream, frame); see celt_enc_*.h for the body. */
#include "wave.h"
#ifdef OA_PHASE_TIMERS
/* profiling variant: shader-clock ticks per encoder phase, accumulated per wave in LDS (lane 0) and added to the
 * global totals once per frame */
__device__ unsigned long long oa_phase_ticks[50];
#define K_TIC() unsigned long long tic_ = clock64()
#define K_TOC(b) do { if (threadIdx.x == 0) { unsigned long long t_ = clock64(); L->prof[b] += (u32)(t_ - tic_); tic_ = t_; } } while (0)
/* K_PHASE_BEGIN at the top of the call; K_PHASE(0) closes what precedes the coded frame (state load, analysis, decisions: bucket 15), K_PHASE(id) closes phase id - 1 */
#define K_PHASE_BEGIN() do { if (threadIdx.x == 0) { for (int z_ = 0; z_ < 34; z_++) L->prof[z_] = 0; L->prof_t0 = (u32)clock64(); } } while (0)
#define K_PHASE(id) do { if (threadIdx.x == 0) { const u32 t_ = (u32)clock64(); L->prof[(id) > 0 ? (id) - 1 : 15] = t_ - L->prof_t0; L->prof_t0 = t_; \
      if ((id) == 15) for (int z_ = 0; z_ < 31; z_++) atomicAdd(&oa_phase_ticks[z_], (unsigned long long)L->prof[z_]); } } while (0)
/* the analysis has no FrameLds at hand: its sections go straight to the global totals (buckets 31..33) */
#define AN_TIC() unsigned long long an_tic_ = clock64()
#define AN_TOC(b) do { if (threadIdx.x == 0) { unsigned long long t_ = clock64(); atomicAdd(&oa_phase_ticks[b], t_ - an_tic_); an_tic_ = t_; } } while (0)
/* ... and a second clock for the finer sections inside those (buckets 34..49) */
#define AN2_TIC() unsigned long long an2_tic_ = clock64()
#define AN2_TOC(b) do { if (threadIdx.x == 0) { unsigned long long t_ = clock64(); atomicAdd(&oa_phase_ticks[b], t_ - an2_tic_); an2_tic_ = t_; } } while (0)
#endif
#ifdef OA_PHASE_TIMERS
/* the four-streams-per-wave PVQ kernel: a section's wave time (the first active lane adds the ticks) and its lane time (ticks x active lanes: how full the wave was),
 * accumulated per wave in LDS (ds_add), added to the global totals when the wave is done with its four streams */
__device__ unsigned long long oa_p4_ticks[32], oa_p4_lanes[32];
__shared__ unsigned int oa_p4_prof[64];
#define P4_TIC() unsigned long long p4tic_ = clock64()
#define P4_TOC(b) do { const unsigned long long t_ = clock64(); const int me_ = (int)threadIdx.x; const unsigned n_ = (unsigned)__popcll(__ballot(1)); if (me_ == __builtin_amdgcn_readfirstlane(me_)) { \
      atomicAdd(&oa_p4_prof[b], (unsigned)(t_ - p4tic_)); atomicAdd(&oa_p4_prof[32 + (b)], (unsigned)(t_ - p4tic_) * n_ >> 6); } p4tic_ = t_; } while (0)
#define P4_PROF_BEGIN() do { oa_p4_prof[threadIdx.x] = 0; __syncthreads(); } while (0)
#define P4_PROF_END() do { __syncthreads(); if (threadIdx.x < 32) { atomicAdd(&oa_p4_ticks[threadIdx.x], (unsigned long long)oa_p4_prof[threadIdx.x]); atomicAdd(&oa_p4_lanes[threadIdx.x], (unsigned long long)oa_p4_prof[32 + threadIdx.x]); } __syncthreads(); } while (0)
#endif
#include "celt_enc_all.h"
#include "celt_dec_all.h"
#include "silk_dec_lane.h"
#ifdef OA_PHASE_TIMERS
/* SILK-capable kernel: shader-clock ticks between SE_PHASE marks (lane 0), summed over all waves */
__device__ unsigned long long oa_sh_phase_ticks[24];
#define SE_PHASE(S_, id) do { if (threadIdx.x == 0) { const u32 t_ = (u32)clock64(); atomicAdd(&oa_sh_phase_ticks[id], (unsigned long long)(u32)(t_ - (u32)(S_)->r[15])); (S_)->r[15] = (i32)t_; } } while (0)
#define SE_PHASE_START(S_) do { if (threadIdx.x == 0) (S_)->r[15] = (i32)(u32)clock64(); } while (0)
#define SE_TICK(tk_, id) do { if (threadIdx.x == 0) { const u32 t_ = (u32)clock64(); atomicAdd(&oa_sh_phase_ticks[id], (unsigned long long)(u32)(t_ - (u32)*(tk_))); *(tk_) = (i32)t_; } } while (0)
#define SE_CLK_BEGIN() const u32 clk0_ = (u32)clock64()
#define SE_CLK_END(id) do { if (threadIdx.x == 0) atomicAdd(&oa_sh_phase_ticks[id], (unsigned long long)(u32)((u32)clock64() - clk0_)); } while (0)
/* a clock of the function's own (a register): sub-marks that leave the parent phase's clock alone */
#define SE_LTIC() u32 ltic_ = (u32)clock64()
#define SE_LTOC(id) do { const u32 t_ = (u32)clock64(); if (threadIdx.x == 0) atomicAdd(&oa_sh_phase_ticks[id], (unsigned long long)(u32)(t_ - ltic_)); ltic_ = t_; } while (0)
#endif
#include "silk_enc_all.h"
#include "opus_surround.h"
#include "../../include/opus_amd.h"
#include <stdarg.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include <stdio.h>
#include <mutex>
#include <vector>
#include "opus_call_combiner.h"

/* The encoder kernels are persistent: the launch fills the chip once (grid = resident waves, opusgpu host code), every wave pops stream indices from a
 * device-side queue until it is empty.  Streams cost different amounts (transients, VBR), so the queue balances what a static blockIdx -> stream map
 * cannot, and the bulk scratch of a frame (CeltScratch) belongs to the wave, not the stream: grid x 21 KB that stays in the XCD's L2 / the MALL instead
 * of streams x 21 KB streaming through HBM. */
WV_DEV int oa_queue_pop(unsigned *queue) { int s = 0; if (wv_lane() == 0) s = (int)atomicAdd(queue, 1u); return wv_bcast(s, 0); }

#ifndef OA_ENC_WAVES_PER_EU
#define OA_ENC_WAVES_PER_EU 4
#endif
#ifndef OA_KEY_BYTES_SHIFT
#define OA_KEY_BYTES_SHIFT 4      /* the byte budget in steps of 16 */
#endif
#ifndef OA_KEY_TRIM_BITS
#define OA_KEY_TRIM_BITS 0
#endif
#define OA_KEY_BYTES_N (512 >> OA_KEY_BYTES_SHIFT)
#define OA_SORT_KEYS (16 * (1 << OA_KEY_TRIM_BITS) * OA_KEY_BYTES_N)          /* the order the PVQ kernel takes the cut frames in: see oa_celt_sort_kernel */
WV_DEV int oa_cut_key(const WV_LDS FrameLds *F)
{
   int k = (F->sh.shortBlocks ? 2 : 0) | (F->sh.dual_stereo ? 1 : 0);
   k = k * 4 + (F->st.spread_decision & 3);
   if (OA_KEY_TRIM_BITS) k = (k << OA_KEY_TRIM_BITS) + (imin(11, imax(0, F->sh.alloc_trim)) >> (4 - OA_KEY_TRIM_BITS));
   return k * OA_KEY_BYTES_N + imin(OA_KEY_BYTES_N - 1, F->sh.nbCompressedBytes >> OA_KEY_BYTES_SHIFT);
}
template <bool NOPVQ> WV_DEV void oa_encode_kernel_body(OaStream *streams, const i16 *pcm, const i32 *apcm, int frame_size, int max_data_bytes, u8 *out, int out_stride, i32 *lens, u32 *rngs, int nstreams, CeltScratch *scratch, unsigned *queue,
      int pcm_row /* samples per channel of a stream's row of pcm / apcm: frame_size, or more when the caller hands the analysis a look-ahead */,
      int first, int stride /* the call's streams: first, first + stride, ... (nstreams of them; 0, 1: the first nstreams records) */,
      const i32 *budget /* NULL, or per stream record: this call's max_data_bytes for it, <= 0 = the stream sits this call out (opus_ms_batch.h: chained byte budgets) */,
      const i32 *tr /* NULL, or [record][4]: the unmask values oa_celt_transient_kernel worked out for this call's frames */,
      CeltCont *conts /* NULL, or [record]: the kernel pipeline -- a single-frame call of 10 / 20 ms stops before the PVQ, its stream goes on the list (queue[1] counts) for oa_celt_pvq_kernel / oa_celt_back_kernel */,
      int *cut_list, unsigned *srt /* the key counts of the PVQ kernel's order (oa_celt_sort_kernel) */)
{
   extern __shared__ __attribute__((aligned(16))) char smem[];
   WV_LDS FrameLds *L = (WV_LDS FrameLds *)smem;
   for (;;) {
      const int i_ = oa_queue_pop(queue);
      if (i_ >= nstreams) break;
      const int s = first + i_ * stride;
      if (budget) { max_data_bytes = wv_uni(budget[s]); if (max_data_bytes <= 0) continue; }
      if (threadIdx.x == 0) L->g = scratch + blockIdx.x;
      __syncthreads();
      OaStream *gs = streams + s;
      const int ch = gs->cfg.channels;
      const int cut = oa_encode_frame<NOPVQ>(L, gs, pcm + (size_t)s * pcm_row * ch, frame_size, max_data_bytes, out + (size_t)s * out_stride, out_stride, lens + s, rngs + s,
            apcm ? apcm + (size_t)s * pcm_row * ch : nullptr, pcm_row, tr ? tr + 4 * (size_t)s : nullptr, conts ? conts + s : nullptr);
      if (cut) { if (threadIdx.x == 0) { const int key = oa_cut_key(L); conts[s].sort_key = key; atomicAdd(srt + key, 1u); cut_list[atomicAdd(queue + 1, 1u)] = s; } }
      __syncthreads();
   }
}
#define OA_ENC_KERNEL_PARAMS OaStream *streams, const i16 *pcm, const i32 *apcm, int frame_size, int max_data_bytes, u8 *out, int out_stride, i32 *lens, u32 *rngs, int nstreams, CeltScratch *scratch, unsigned *queue, \
      int pcm_row, int first, int stride, const i32 *budget, const i32 *tr, CeltCont *conts, int *cut_list, unsigned *srt
#define OA_ENC_KERNEL_ARGS streams, pcm, apcm, frame_size, max_data_bytes, out, out_stride, lens, rngs, nstreams, scratch, queue, pcm_row, first, stride, budget, tr, conts, cut_list, srt
extern "C" __global__ void __launch_bounds__(64, OA_ENC_WAVES_PER_EU) oa_encode_kernel(OA_ENC_KERNEL_PARAMS) { oa_encode_kernel_body<false>(OA_ENC_KERNEL_ARGS); }
/* the pipeline's front kernel: oa_encode_kernel on single-frame calls of 10 / 20 ms with continuation records -- every frame that reaches the PVQ is cut there, so the PVQ
 * (and the multi-frame loop) are not in its code */
#ifndef OA_FRONT_WAVES_PER_EU
#define OA_FRONT_WAVES_PER_EU 4
#endif
extern "C" __global__ void __launch_bounds__(64, OA_FRONT_WAVES_PER_EU) oa_celt_front_kernel(OA_ENC_KERNEL_PARAMS) { oa_encode_kernel_body<true>(OA_ENC_KERNEL_ARGS); }
/* The four streams of a PVQ wave go through their bands side by side: the more alike their band trees, the fewer instructions the wave spends on branches only some of them
 * take.  The frames that were cut are therefore handed to the PVQ kernel sorted by what shapes the tree -- block switching, dual stereo, the spreading decision (which
 * leaves are rotated), the frame's byte budget -- with a counting sort: the kernel that cuts a frame counts its key (srt[key]), oa_celt_sort_kernel places every list
 * entry behind the keys below it (srt[OA_SORT_KEYS + key] fills). */
extern "C" __global__ void __launch_bounds__(64)
oa_celt_sort_kernel(const CeltCont *conts, const int *cut_list, int *order, const unsigned *queue, unsigned *srt)
{
   const int n = (int)queue[1], lane = (int)threadIdx.x;
   /* exclusive prefix of the key counts, OA_SORT_KEYS / 64 consecutive keys per lane */
   __shared__ i32 base[OA_SORT_KEYS];
   {
      const int per = OA_SORT_KEYS / 64;
      i32 c[per], sum = 0;
#pragma unroll
      for (int u = 0; u < per; u++) { c[u] = (i32)srt[per * lane + u]; sum += c[u]; }
      i32 run = wv_scan_incl(sum) - sum;
#pragma unroll
      for (int u = 0; u < per; u++) { base[per * lane + u] = run; run += c[u]; }
      __syncthreads();
   }
   for (int k0 = (int)blockIdx.x * 64; k0 < n; k0 += (int)gridDim.x * 64) {
      const int k = k0 + lane;
      int s = 0, key = 0;
      if (k < n) { s = cut_list[k]; key = conts[s].sort_key; }
      if (k < n) order[base[key] + (int)atomicAdd(srt + OA_SORT_KEYS + key, 1u)] = s;
   }
}
/* the PVQ of the frames the encode kernel cut: four streams per wave, one 16-lane group each (celt_enc_pvq4.h) */
#ifndef OA_PVQ4_WAVES_PER_EU
#define OA_PVQ4_WAVES_PER_EU 3
#endif
extern "C" __global__ void __launch_bounds__(64, OA_PVQ4_WAVES_PER_EU)
oa_celt_pvq_kernel(CeltCont *conts, const int *cut_list, unsigned *queue)
{
   extern __shared__ __attribute__((aligned(16))) char smem[];
   WV_LDS P4Lds *L4 = (WV_LDS P4Lds *)smem;
   const int n = (int)wv_uni((i32)queue[1]);
   for (;;) {
      int base = 0;
      if (wv_lane() == 0) base = (int)atomicAdd(queue + 2, 4u);
      base = wv_bcast(base, 0);
      if (base >= n) break;
      const int k = base + wg_id();
#ifdef OA_PHASE_TIMERS
      P4_PROF_BEGIN();
#endif
      p4_quant_all_bands(L4, k < n ? conts + cut_list[k] : (CeltCont *)0);
#ifdef OA_PHASE_TIMERS
      P4_PROF_END();
#endif
      __syncthreads();
   }
}
/* ... and the rest of their call: one wave per stream again, the front wave's LDS reloaded from the continuation record */
extern "C" __global__ void __launch_bounds__(64, OA_ENC_WAVES_PER_EU)
oa_celt_back_kernel(OaStream *streams, CeltCont *conts, const int *cut_list, unsigned *queue, int frame_size, u8 *out, int out_stride, i32 *lens, u32 *rngs, CeltScratch *scratch)
{
   extern __shared__ __attribute__((aligned(16))) char smem[];
   WV_LDS FrameLds *L = (WV_LDS FrameLds *)smem;
   const int n = (int)wv_uni((i32)queue[1]);
   for (;;) {
      int k = 0;
      if (wv_lane() == 0) k = (int)atomicAdd(queue + 3, 1u);
      k = wv_bcast(k, 0);
      if (k >= n) break;
      const int s = wv_uni(cut_list[k]);
      CeltCont *c = conts + s;
      __syncthreads();
      FOR_LANES(i, (int)(offsetof(FrameLds, BC) / 4)) ((WV_LDS i32 *)L)[i] = c->image[i];
      __syncthreads();
      if (threadIdx.x == 0) L->g = scratch + blockIdx.x;
      __syncthreads();
      oa_encode_frame_back(L, streams + s, frame_size, out + (size_t)s * out_stride, out_stride, lens + s, rngs + s);
      __syncthreads();
   }
}
/* ahead of oa_encode_kernel in a wide launch of 48 kHz frames up to 20 ms: the serial part of every stream's transient analysis, one lane per (stream, channel) (celt_enc_front.h) */
extern "C" __global__ void __launch_bounds__(64)
oa_celt_transient_kernel(const OaStream *streams, const i16 *pcm, int pcm_row, int frame_size, int channels, int first, int stride, int n_items, i16 *scratch, i32 *tr)
{
   const int ntiles = (n_items + 63) / 64;
   for (int t = (int)blockIdx.x; t < ntiles; t += (int)gridDim.x)
      ct_transient_tile(streams, pcm, pcm_row, frame_size, channels, first, stride, n_items, t * 64, scratch + (size_t)blockIdx.x * (OA_MAX_FRAME + OA_OVERLAP) * 64, tr);
}

/* The decoder: persistent waves fed from a queue, one packet of one stream at a time; the spectrum of the frame in flight lives in the wave's HBM scratch (celt_dec_lds.h).
 * Two kernels on one HIP stream:
 *   oa_decode_fast_kernel  packets whose decode is the CELT steady state and nothing else -- a CELT-only TOC with one coded frame, a stream whose last packet was CELT-only too
 *                          (or that has not decoded anything yet), no FEC request, no pending fold of the concealment -- with the SILK decoder, the concealment and every
 *                          transition compiled out (oa_decode_packet<true>): it needs neither the A arena nor their registers and synthesises one channel at a time, so 16 waves
 *                          share a CU; it marks the streams it decoded in `taken`
 *   oa_decode_kernel       the general decoder over the streams not taken (taken == NULL: over every stream) */
extern "C" __global__ void __launch_bounds__(64, 2)
oa_decode_kernel(OaDecStream *streams, const u8 *packets, int packet_stride, const i32 *lens, int frame_size, i16 *pcm, int pcm_stride, i32 *nsamples, u32 *rngs, int nstreams, int decode_fec,
      char *scratch, unsigned *queue, const int *list /* NULL: every stream; else the streams oa_decode_look_kernel left to this kernel */, const unsigned *list_count)
{
   extern __shared__ __attribute__((aligned(16))) char smem[];
   WV_LDS DecLds *L = (WV_LDS DecLds *)smem;
   const int n = list ? (int)*list_count : nstreams;
   for (;;) {
      int s = oa_queue_pop(queue);
      if (s >= n) break;
      if (list) s = wv_uni(list[s]);
      const int len = lens[s];
      if (len > packet_stride) { if (threadIdx.x == 0) { nsamples[s] = OPUS_BAD_ARG; rngs[s] = 0; } }       /* a length beyond the stream's slot would read the neighbour's packet */
      else {
         if (threadIdx.x == 0) L->Xg = (i32 *)(scratch + (size_t)blockIdx.x * OA_DEC_SCRATCH_BYTES);
         __syncthreads();
         oa_decode_packet<false>(L, streams + s, packets + (size_t)s * packet_stride, len, frame_size, pcm + (size_t)s * pcm_stride, nsamples + s, rngs + s, decode_fec);
      }
      __syncthreads();
   }
}
#ifndef OA_DEC_FAST_WAVES_PER_EU
#define OA_DEC_FAST_WAVES_PER_EU 4
#endif
#ifdef OA_PHASE_TIMERS
#define OA_DEC_FAST_DYN_LDS_MAX (159 * 1024)          /* (the timers' static LDS words come on top) */
#else
#define OA_DEC_FAST_DYN_LDS_MAX (160 * 1024)
#endif
extern "C" __global__ void __launch_bounds__(64, OA_DEC_FAST_WAVES_PER_EU)
oa_decode_fast_kernel(OaDecStream *streams, const u8 *packets, int packet_stride, const i32 *lens, int frame_size, i16 *pcm, int pcm_stride, i32 *nsamples, u32 *rngs, int nstreams,
      char *scratch, unsigned *queue, const int *list /* the streams oa_decode_look_kernel found in the CELT steady state */, const unsigned *list_count,
      CeltDecCont *conts /* NULL, or [stream]: the kernel pipeline -- a packet of one 10 / 20 ms frame stops in front of its bands, its stream goes on the list of its frame size
                          * (cut_list [2][nstreams]: 20 ms, 10 ms; cut_count[2]) for oa_celt_dpvq_kernel / oa_celt_dback_kernel; the spectrum of every packet then lives in the stream's record */,
      int *cut_list, unsigned *cut_count)
{
   extern __shared__ __attribute__((aligned(16))) char smem[];
   WV_LDS DecLds *L = (WV_LDS DecLds *)smem;
   const int n = (int)*list_count;
   for (;;) {
      const int i = oa_queue_pop(queue);
      if (i >= n) break;
      const int s = wv_uni(list[i]);
      if (threadIdx.x == 0) L->Xg = conts ? conts[s].xg : (i32 *)(scratch + (size_t)blockIdx.x * OA_DEC_SCRATCH_BYTES);
      __syncthreads();
#ifdef OA_PHASE_TIMERS
      P4_PROF_BEGIN();
#endif
      const int cut = oa_decode_packet<true>(L, streams + s, packets + (size_t)s * packet_stride, lens[s], frame_size, pcm + (size_t)s * pcm_stride, nsamples + s, rngs + s, 0, conts ? conts + s : (CeltDecCont *)0);
      if (cut) { if (threadIdx.x == 0) { const int w = L->sh.LM == 3 ? 0 : 1; cut_list[(size_t)w * nstreams + atomicAdd(cut_count + w, 1u)] = s; } }
#ifdef OA_PHASE_TIMERS
      P4_PROF_END();
#endif
      __syncthreads();
   }
}
/* The CELT layer of the hybrid packets whose SILK layer oa_sdec_lane_kernel has decoded: the fast kernel's frame function from band 17 on top of the SILK audio
 * (celt_dec_frame.h: oa_decode_hybrid_tail), persistent waves over the list the lane kernel built */
extern "C" __global__ void __launch_bounds__(64, OA_DEC_FAST_WAVES_PER_EU)
oa_decode_hyb_kernel(OaDecStream *streams, const u8 *packets, int packet_stride, const i32 *lens, i16 *pcm, int pcm_stride, i32 *nsamples, u32 *rngs,
      char *scratch, unsigned *queue, const int *list, const unsigned *list_count, const OaHybCont *hyb_ec,
      CeltDecCont *conts /* as in oa_decode_fast_kernel */, int *cut_list, unsigned *cut_count, int nstreams)
{
   extern __shared__ __attribute__((aligned(16))) char smem[];
   WV_LDS DecLds *L = (WV_LDS DecLds *)smem;
   const int n = (int)*list_count;
   for (;;) {
      const int i = oa_queue_pop(queue);
      if (i >= n) break;
      const int s = wv_uni(list[i]);
      if (threadIdx.x == 0) L->Xg = conts ? conts[s].xg : (i32 *)(scratch + (size_t)blockIdx.x * OA_DEC_SCRATCH_BYTES);
      __syncthreads();
      const int cut = oa_decode_hybrid_tail(L, streams + s, packets + (size_t)s * packet_stride, pcm + (size_t)s * pcm_stride, nsamples + s, rngs + s, hyb_ec + s, conts ? conts + s : (CeltDecCont *)0);
      if (cut) { if (threadIdx.x == 0) { const int w = L->sh.LM == 3 ? 0 : 1; cut_list[(size_t)w * nstreams + atomicAdd(cut_count + w, 1u)] = s; } }
      __syncthreads();
   }
}
/* The bands of the frames the two kernels above stopped: four streams per wave, one 16-lane group each (celt_dec_pvq4.h); a wave takes four entries of ONE list (the bands of
 * a wave go in lockstep, their sizes depend on the frame size).  queue: the tile counter. */
#ifndef OA_DPVQ4_WAVES_PER_EU
#define OA_DPVQ4_WAVES_PER_EU 3
#endif
extern "C" __global__ void __launch_bounds__(64, OA_DPVQ4_WAVES_PER_EU)
oa_celt_dpvq_kernel(CeltDecCont *conts, const int *cut_list, const unsigned *cut_count, unsigned *queue, int nstreams)
{
   extern __shared__ __attribute__((aligned(16))) char smem[];
   WV_LDS P4Lds *L4 = (WV_LDS P4Lds *)smem;
   const int n0 = (int)wv_uni((i32)cut_count[0]), n1 = (int)wv_uni((i32)cut_count[1]);
   const int t0 = (n0 + 3) >> 2, t1 = (n1 + 3) >> 2;
   for (;;) {
      int t = 0;
      if (wv_lane() == 0) t = (int)atomicAdd(queue, 1u);
      t = wv_bcast(t, 0);
      if (t >= t0 + t1) break;
      const int w = t >= t0, k = (w ? t - t0 : t) * 4 + wg_id(), n = w ? n1 : n0;
#ifdef OA_PHASE_TIMERS
      P4_PROF_BEGIN();
#endif
      p4d_quant_all_bands(L4, k < n ? conts + cut_list[(size_t)w * nstreams + k] : (CeltDecCont *)0);
#ifdef OA_PHASE_TIMERS
      P4_PROF_END();
#endif
      __syncthreads();
   }
}
/* ... and the rest of their packets: one wave per stream again, the front wave's LDS reloaded from the continuation record (celt_dec_frame.h: oa_decode_packet_back) */
extern "C" __global__ void __launch_bounds__(64, OA_DEC_FAST_WAVES_PER_EU)
oa_celt_dback_kernel(OaDecStream *streams, CeltDecCont *conts, const int *cut_list, const unsigned *cut_count, unsigned *queue, i16 *pcm, int pcm_stride, i32 *nsamples, u32 *rngs, int nstreams,
      int defer_deemph /* the de-emphasis and the PCM store are oa_celt_deemph_kernel's */)
{
   extern __shared__ __attribute__((aligned(16))) char smem[];
   WV_LDS DecLds *L = (WV_LDS DecLds *)smem;
   const int n0 = (int)wv_uni((i32)cut_count[0]), n1 = (int)wv_uni((i32)cut_count[1]);
   for (;;) {
      const int k = oa_queue_pop(queue);
      if (k >= n0 + n1) break;
      const int s = wv_uni(k < n0 ? cut_list[k] : cut_list[(size_t)nstreams + (k - n0)]);
      __syncthreads();
#ifdef OA_PHASE_TIMERS
      P4_PROF_BEGIN();
#endif
      oa_decode_packet_back(L, streams + s, conts + s, pcm + (size_t)s * pcm_stride, nsamples + s, rngs + s, defer_deemph);
#ifdef OA_PHASE_TIMERS
      P4_PROF_END();
#endif
      __syncthreads();
   }
}
/* ... whose de-emphasis recursion and PCM store run one lane per stream (celt_dec_frame.h: oa_deemph_lane) */
extern "C" __global__ void __launch_bounds__(64)
oa_celt_deemph_kernel(OaDecStream *streams, const CeltDecCont *conts, const int *cut_list, const unsigned *cut_count, i16 *pcm, int pcm_stride, int nstreams)
{
   const int n0 = (int)cut_count[0], n1 = (int)cut_count[1];
   for (int k = (int)blockIdx.x * 64 + (int)threadIdx.x; k < n0 + n1; k += (int)gridDim.x * 64) {
      const int s = k < n0 ? cut_list[k] : cut_list[(size_t)nstreams + (k - n0)];
      oa_deemph_lane(streams + s, conts + s, pcm + (size_t)s * pcm_stride);
   }
}
/* The look that sorts a call's packets between the decoder's kernels, one LANE per stream (64 streams per wave, one ballot and one atomic per list and wave):
 *   fast list   the CELT steady state and nothing else -- a CELT-only TOC with one coded frame that fits a frame's slot, a stream whose last packet was CELT-only too (or that
 *               has not decoded anything yet), no pending fold of the concealment
 *   lane list   (lane_list != NULL) the SILK steady state -- a SILK-only TOC with one coded frame, the stream's last packet SILK-only too, the internal rate and the channel
 *               count of last time (so that silk_decoder_set_fs and the resampler set-up have nothing to do), API channels = coded channels or a mono packet into a stereo decoder, nothing lost last time:
 *               oa_sdec_lane_kernel, 64 streams per wave (silk_dec_lane.h) -- and the hybrid steady state (a hybrid TOC with one coded frame after a hybrid packet, no
 *               pending fold of the concealment): the same kernel for the SILK layer, then oa_decode_hyb_kernel for the CELT layer
 *   slow list   everything else: the general kernel
 * counters: [0] number of fast streams, [1] number of the others, [2] number of lane streams. */
extern "C" __global__ void __launch_bounds__(64)
oa_decode_look_kernel(const OaDecStream *streams, const u8 *packets, int packet_stride, const i32 *lens, int nstreams, int frame_size, int *fast_list, int *slow_list, int *lane_list, unsigned *counters)
{
   const int lane = (int)threadIdx.x, s = (int)blockIdx.x * 64 + lane;
   int fast = 0, ln = 0;
   if (s < nstreams) {
      const int len = lens[s];
      if (len >= 3 && len <= packet_stride && len <= 1276) {
         const int toc = packets[(size_t)s * packet_stride];
         const OaDecStream *g = streams + s;
         const int prev = g->s.prev_mode;
         fast = fast_list && (toc & 0x80) && (toc & 3) == 0 && (prev == 0 || prev == 1002) && g->s.prefilter_and_fold == 0;
         const int hyb = (toc & 0x60) == 0x60;
         const int one = (toc & 3) == 0 || ((toc & 3) == 3 && (packets[(size_t)s * packet_stride + 1] & 0x3F) == 1);           /* one coded frame: code 0, or code 3 with M = 1 (a padded packet) */
         if (lane_list && !(toc & 0x80) && one && (hyb ? prev == 1001 && g->s.prefilter_and_fold == 0 : prev == 1000)) {
            const int Fs = g->s.Fs ? g->s.Fs : 48000, nch = (toc & 0x4) ? 2 : 1, bw = 1101 + ((toc >> 5) & 0x3);
            const int rate = hyb ? 16000 : bw == 1101 ? 8000 : bw == 1102 ? 12000 : 16000;
            ln = oa_samples_per_frame(toc, Fs) <= frame_size && (nch == g->s.channels || nch == 1) && g->silk.nChannelsInternal == nch && g->silk.nChannelsAPI == g->s.channels && g->silk.lastChannelsInternal == nch &&
                 g->silk.lastInternalRate == rate;
            for (int n = 0; n < nch && ln; n++) ln = g->silk.ch[n].fs_kHz * 1000 == rate && g->silk.ch[n].fs_API_hz == Fs && g->silk.ch[n].lossCnt == 0 && g->silk.ch[n].rs_cfg[5] * 1000 == rate;
         }
      }
   }
   const int slow = s < nstreams && !fast && !ln;
   const unsigned long long mf = wv_ballot(fast), ms = wv_ballot(slow), ml = wv_ballot(ln), below = lane ? (~0ull >> (64 - lane)) : 0ull;
   int bf = 0, bs = 0, bl = 0;
   if (lane == 0) {
      bf = mf ? (int)atomicAdd(counters, (unsigned)__builtin_popcountll(mf)) : 0; bs = ms ? (int)atomicAdd(counters + 1, (unsigned)__builtin_popcountll(ms)) : 0;
      bl = ml ? (int)atomicAdd(counters + 2, (unsigned)__builtin_popcountll(ml)) : 0;
   }
   bf = wv_bcast(bf, 0); bs = wv_bcast(bs, 0); bl = wv_bcast(bl, 0);
   if (fast) fast_list[bf + __builtin_popcountll(mf & below)] = s;
   else if (ln) lane_list[bl + __builtin_popcountll(ml & below)] = s;
   else if (slow) slow_list[bs + __builtin_popcountll(ms & below)] = s;
}
/* The SILK steady state, one lane per stream (silk_dec_lane.h): tiles of 64 entries of the look's lane list, a static split over the grid; a lane whose packet turns out to
 * carry a redundant CELT frame appends its stream to the general kernel's list (which is launched behind this kernel on the same HIP stream).
 * tile_width: the lanes of a wave that take a stream.  The lane code is a chain of dependent steps (a symbol of the range decoder, a sample of a recursive filter): a wave
 * alone on its SIMD waits out every latency, so a call with too few streams to give every SIMD two full waves runs half-filled waves, two per SIMD, instead.
 * work: SL_WORK_BYTES per block. */
#ifndef OA_SDEC_WAVES_PER_EU
#define OA_SDEC_WAVES_PER_EU 2
#endif
extern "C" __global__ void __launch_bounds__(64, OA_SDEC_WAVES_PER_EU)
oa_sdec_lane_kernel(OaDecStream *streams, const u8 *packets, int packet_stride, const i32 *lens, i16 *pcm, int pcm_stride, i32 *nsamples, u32 *rngs, char *work, int tile_width,
      const int *list, const unsigned *list_count, int *slow_list, unsigned *slow_count, unsigned *rejected, int *hyb_list, unsigned *hyb_count, OaHybCont *hyb_ec)
{
   extern __shared__ __attribute__((aligned(16))) char smem[];
   const int n = (int)*list_count, ntiles = (n + tile_width - 1) / tile_width, lane = (int)threadIdx.x;
   sl_tabs_fill((WV_LDS SlTabs *)(smem + sizeof(ResamplerLds)), lane);
   __syncthreads();
   for (int t = (int)blockIdx.x; t < ntiles; t += (int)gridDim.x) {
      const int i = lane < tile_width ? t * tile_width + lane : n;
      const int s = i < n ? list[i] : -1, len_s = s >= 0 ? lens[s] : 0;
      WV_LDS u8 *win = (WV_LDS u8 *)(smem + sizeof(ResamplerLds) + sizeof(SlTabs));
      for (int k = 0; k < tile_width; k++) {                           /* the packets' heads -> LDS, a stream at a time, every lane a byte of it */
         const int sk = wv_shfl(s, k), nb = imin(wv_shfl(len_s, k) - 1, SL_WIN);
         if (sk >= 0) { const u8 *src = packets + (size_t)sk * packet_stride + 1; for (int j = lane; j < nb; j += 64) win[k * SL_WIN_STRIDE + j] = src[j]; }
      }
      __syncthreads();
#ifdef OA_PHASE_TIMERS
      P4_PROF_BEGIN();
#endif
      if (i < n) {
         const int r = oa_sdec_lane_packet(streams + s, packets + (size_t)s * packet_stride, len_s, pcm + (size_t)s * pcm_stride, nsamples + s, rngs + s, hyb_ec + s,
                                           work + (size_t)blockIdx.x * SL_WORK_BYTES, (WV_LDS ResamplerLds *)smem, (const WV_LDS SlTabs *)(smem + sizeof(ResamplerLds)), win, lane);
         if (r == 0) { slow_list[atomicAdd(slow_count, 1u)] = s; atomicAdd(rejected, 1u); }
         else if (r == 2) hyb_list[atomicAdd(hyb_count, 1u)] = s;
      }
#ifdef OA_PHASE_TIMERS
      P4_PROF_END();
#endif
      __syncthreads();
   }
}

/* the SILK-capable encoder (applications VOIP / AUDIO / RESTRICTED_SILK): one wave per stream at a time, SILK state staged in LDS; persistent like oa_encode_kernel,
 * the per-frame HBM scratch (scratch_bytes per wave) belongs to the wave.  list != NULL: the calls the split path's front kernel turned away (their analysis has run) */
extern "C" __global__ void __launch_bounds__(64, 2)
oa_sh_encode_kernel(OaShStream *streams, const i16 *pcm, const i32 *apcm, int frame_size, int max_data_bytes, u8 *out, int out_stride, char *scratch, i32 *lens, u32 *rngs, int nstreams, unsigned *queue,
      const int *list, const unsigned *list_count, int pkt_off, int pcm_row, int first, int stride, const i32 *budget /* as in oa_encode_kernel */)
{
   extern __shared__ __attribute__((aligned(16))) char smem[];
   WV_LDS ShLds *L = (WV_LDS ShLds *)smem;
   const int n = list ? (int)*list_count : nstreams;
   for (;;) {
      int s = oa_queue_pop(queue);
      if (s >= n) break;
      if (list) s = list[s]; else s = first + s * stride;
      if (budget) { max_data_bytes = wv_uni(budget[s]); if (max_data_bytes <= 0) continue; }
      OaShStream *gs = streams + s;
      const int ch = gs->cfg.channels;
      char *scr = scratch + (size_t)blockIdx.x * SH_SCRATCH_BYTES(frame_size, ch);
      char *tail = scr + SH_SCRATCH_BYTES(frame_size, ch);
      if (threadIdx.x == 0) { L->silk_tail = 1; L->packet_off = pkt_off; L->S.st_off = (i32)offsetof(SilkEncLds, st); }
      __syncthreads();
      oa_sh_encode_frame(L, gs, pcm + (size_t)s * pcm_row * ch, frame_size, max_data_bytes, out + (size_t)s * out_stride, out_stride, (i16 *)scr,
            (SeRateScratch *)(tail - sizeof(SeRateScratch)), (CeltScratch *)(tail - sizeof(SeRateScratch) - sizeof(CeltScratch)), lens + s, rngs + s,
            apcm ? apcm + (size_t)s * pcm_row * ch : nullptr, list != nullptr, pcm_row);
      __syncthreads();
   }
}
/* the split path (opus_sh_split.h).  counters: [0] front queue, [1] quantiser queue, [2] back queue, [3] queue of the one-kernel pass over the calls turned away, [4] their count, [5] queue of the one-wave pred kernel (value 3), [6] count of the pred work list */
#ifndef OA_SH_FRONT_WAVES_PER_EU
#define OA_SH_FRONT_WAVES_PER_EU 4
#endif
extern "C" __global__ void __launch_bounds__(64, OA_SH_FRONT_WAVES_PER_EU)
oa_sh_front_kernel(OaShStream *streams, const i16 *pcm, const i32 *apcm, int frame_size, int max_data_bytes, char *pcm_hp_all, CeltScratch *scratch, ShCont *conts, int *slow_list, unsigned *counters, int nstreams, int pkt_off, int pcm_row, int pred_split, int pkt_window /* bytes of packet buffer behind pkt_off: SH_FRONT_PKT_BYTES, or SH_PKT_BYTES when the batch has in-band FEC on */)
{
   extern __shared__ __attribute__((aligned(16))) char smem[];
   WV_LDS ShLds *L = (WV_LDS ShLds *)smem;
   unsigned kept = 0, seen = 0;
   for (;;) {
      const int s = oa_queue_pop(counters);
      if (s >= nstreams) break;
      OaShStream *gs = streams + s;
      const int ch = gs->cfg.channels;
      seen++;
      if (threadIdx.x == 0) { L->packet_off = pkt_off; L->S.st_off = (i32)SE_FRONT_ST_OFF; }
      __syncthreads();
      oa_sh_front_frame(L, gs, pcm + (size_t)s * pcm_row * ch, frame_size, max_data_bytes, (i16 *)(pcm_hp_all + (size_t)s * SH_PCM_BYTES(frame_size, ch)), scratch + blockIdx.x, conts + s,
            apcm ? apcm + (size_t)s * pcm_row * ch : nullptr, slow_list, counters + 4, s, pcm_row, pred_split, pkt_window);
      __syncthreads();
      kept += conts[s].kind == SH_CONT_FAST;
      if (pred_split && threadIdx.x == 0 && conts[s].kind == SH_CONT_FAST && conts[s].nq > 0) {      /* the pred kernel's work list: one item (stream * 2 + job) per coded channel, behind the list of the calls turned away */
         const int nq = conts[s].nq; const unsigned at = atomicAdd(counters + 6, (unsigned)nq);
         for (int j = 0; j < nq; j++) slow_list[nstreams + at + j] = 2 * s + j;
      }
   }
   if (threadIdx.x == 0 && seen) { atomicAdd(counters + 16, kept); atomicAdd(counters + 17, seen - kept); }     /* running totals of the batch (opusgpu_enc_batch_split_stats) */
}
/* a 40 / 60 ms SILK packet's second / third frame (oa_sh_front_cont_frame): every stream of the launch the front kernel kept, after the quantiser kernel coded the frame before */
extern "C" __global__ void __launch_bounds__(64, OA_SH_FRONT_WAVES_PER_EU)
oa_sh_frontc_kernel(OaShStream *streams, int frame_size, char *pcm_hp_all, ShCont *conts, int *slow_list, unsigned *counters, int nstreams, int pkt_off, int pred_split, int block)
{
   extern __shared__ __attribute__((aligned(16))) char smem[];
   WV_LDS ShLds *L = (WV_LDS ShLds *)smem;
   for (;;) {
      const int s = oa_queue_pop(counters);
      if (s >= nstreams) break;
      if (wv_uni(conts[s].kind) != SH_CONT_FAST || wv_uni(conts[s].k.tot_blocks) <= block || wv_uni(conts[s].k.curr_block) != block) continue;
      OaShStream *gs = streams + s;
      const int ch = gs->cfg.channels;
      if (threadIdx.x == 0) { L->packet_off = pkt_off; L->S.st_off = (i32)SE_FRONT_ST_OFF; }
      __syncthreads();
      oa_sh_front_cont_frame(L, gs, frame_size, (const i16 *)(pcm_hp_all + (size_t)s * SH_PCM_BYTES(frame_size, ch)), conts + s, block, pred_split);
      __syncthreads();
      if (pred_split && threadIdx.x == 0 && conts[s].nq > 0) {
         const int nq = conts[s].nq; const unsigned at = atomicAdd(counters + 6, (unsigned)nq);
         for (int j = 0; j < nq; j++) slow_list[nstreams + at + j] = 2 * s + j;
      }
   }
}
/* pipeline mode 3: the prediction stage of every coded channel the front kernel kept (oa_sh_pred_frame, opus_sh_split.h), persistent, 8 waves per SIMD */
#ifndef OA_SH_PRED_WAVES_PER_EU
#define OA_SH_PRED_WAVES_PER_EU 8
#endif
extern "C" __global__ void __launch_bounds__(64, OA_SH_PRED_WAVES_PER_EU)
oa_sh_pred_kernel(OaShStream *streams, ShCont *conts, const int *list, const unsigned *list_count, unsigned *queue, int tail /* mode 4: the stage's last part only */)
{
   extern __shared__ __attribute__((aligned(16))) char smem[];
   WV_LDS PredLds *P = (WV_LDS PredLds *)smem;
   const int n = (int)*list_count;                                       /* coded channels with a SILK job: none in a batch of CELT-only frames, whose launch of this kernel then costs a few microseconds */
   if (tail) {                                                           /* short uniform items: a static split (65,536 pops of one counter would cost more than the items, profiles/r05_r) */
      for (int i = (int)blockIdx.x; i < n; i += (int)gridDim.x) { const int it = wv_uni(list[i]); oa_sh_pred_frame(P, streams + (it >> 1), conts + (it >> 1), it & 1, 1); __syncthreads(); }
      return;
   }
   for (;;) {
      const int i = oa_queue_pop(queue);
      if (i >= n) break;
      const int it = wv_uni(list[i]);
      oa_sh_pred_frame(P, streams + (it >> 1), conts + (it >> 1), it & 1, 0);
      __syncthreads();
   }
}
/* pipeline mode 4: the stage's serial parts on lanes, its passes over the signal on waves (silk_enc_predl.h, opus_sh_split.h) -- four launches over the same work list */
extern "C" __global__ void __launch_bounds__(64, 1)
oa_sh_preda_kernel(ShCont *conts, const int *list, const unsigned *list_count)
{
   extern __shared__ __attribute__((aligned(16))) char smem[];
   const int n = (int)*list_count, ntiles = (n + PL_STREAMS - 1) / PL_STREAMS;
   for (int t = (int)blockIdx.x; t < ntiles; t += (int)gridDim.x) {
      oa_sh_preda_tile((WV_LDS i32 *)smem, conts, list, t * PL_STREAMS, imin(PL_STREAMS, n - t * PL_STREAMS));
      __syncthreads();
   }
}
extern "C" __global__ void __launch_bounds__(64, OA_SH_PRED_WAVES_PER_EU)
oa_sh_predc_kernel(ShCont *conts, const int *list, const unsigned *list_count)
{
   extern __shared__ __attribute__((aligned(16))) char smem[];
   const int n = (int)*list_count;
   for (int i = (int)blockIdx.x; i < n; i += (int)gridDim.x) {           /* (short uniform items: a static split, as in the stage's tail) */
      const int it = wv_uni(list[i]);
      oa_sh_predc_frame((WV_LDS PredLds *)smem, conts + (it >> 1), it & 1);
      __syncthreads();
   }
}
extern "C" __global__ void __launch_bounds__(64, 1)
oa_sh_predb_kernel(OaShStream *streams, ShCont *conts, const int *list, const unsigned *list_count)
{
   extern __shared__ __attribute__((aligned(16))) char smem[];
   const int n = (int)*list_count, ntiles = (n + PL_STREAMS - 1) / PL_STREAMS;
   for (int t = (int)blockIdx.x; t < ntiles; t += (int)gridDim.x) {
      oa_sh_predb_tile((WV_LDS PlBLane *)smem, (WV_LDS SeNlsfTabs *)(smem + sizeof(PlBLane) * PL_STREAMS), streams, conts, list, t * PL_STREAMS, imin(PL_STREAMS, n - t * PL_STREAMS));
      __syncthreads();
   }
}
#ifndef OA_SH_QUANT_WAVES_PER_EU
#define OA_SH_QUANT_WAVES_PER_EU 2
#endif
extern "C" __global__ void __launch_bounds__(64, OA_SH_QUANT_WAVES_PER_EU)
oa_sh_quant_kernel(OaShStream *streams, ShCont *conts, int nstreams, char *scratch, unsigned *counters)
{
   extern __shared__ __attribute__((aligned(16))) char smem[];
   WV_LDS SqLds *Q = (WV_LDS SqLds *)smem;
   char *scr = scratch + (size_t)blockIdx.x * SQ_WAVE_SCRATCH_BYTES;
   const int ntiles = (nstreams + 15) / 16;
   for (;;) {
      const int t = oa_queue_pop(counters + 1);
      if (t >= ntiles) break;
      sq_quant_tile_wave(Q, streams, conts, t * 16, nstreams, (i32 *)scr, (SqSnap *)(scr + SQ_TILE_WORDS * 4));
      __syncthreads();
   }
}
extern "C" __global__ void __launch_bounds__(64, 2)
oa_sh_quant0_kernel(OaShStream *streams, ShCont *conts, int nstreams, char *scratch, unsigned *counters, int pkt_off)
{
   extern __shared__ __attribute__((aligned(16))) char smem[];
   WV_LDS ShLds *L = (WV_LDS ShLds *)smem;
   for (;;) {
      const int s = oa_queue_pop(counters + 1);
      if (s >= nstreams) break;
      if (threadIdx.x == 0) { L->silk_tail = 1; L->packet_off = pkt_off; L->S.st_off = (i32)offsetof(SilkEncLds, st); }
      __syncthreads();
      if (conts[s].kind == SH_CONT_FAST) oa_sh_quant0_frame(L, streams + s, conts + s, (SeRateScratch *)(scratch + (size_t)blockIdx.x * sizeof(SeRateScratch)));
      __syncthreads();
   }
}
/* ahead of the back kernel in a 48 kHz launch: the serial part of the CELT layer's transient analysis, one lane per (stream, channel) (opus_sh_split.h: oa_sh_transient_tile) */
extern "C" __global__ void __launch_bounds__(64)
oa_sh_transient_kernel(const OaShStream *streams, const ShCont *conts, const char *pcm_hp_all, int frame_size, int channels, int n_items, i16 *scratch, i32 *tr)
{
   const int ntiles = (n_items + 63) / 64;
   for (int t = (int)blockIdx.x; t < ntiles; t += (int)gridDim.x)
      oa_sh_transient_tile(streams, conts, pcm_hp_all, frame_size, channels, n_items, t * 64, scratch + (size_t)blockIdx.x * (OA_MAX_FRAME + OA_OVERLAP) * 64, tr);
}
#ifndef OA_SH_BACK_WAVES_PER_EU
#define OA_SH_BACK_WAVES_PER_EU 3
#endif
extern "C" __global__ void __launch_bounds__(64, OA_SH_BACK_WAVES_PER_EU)
oa_sh_back_kernel(OaShStream *streams, int frame_size, u8 *out, int out_stride, char *pcm_hp_all, char *scratch, const ShCont *conts, i32 *lens, u32 *rngs, int nstreams, unsigned *counters, int pkt_off,
      int chunk /* streams per pop of the queue: 1, or several where the frames are all light (a batch pinned to SILK-only: 65,536 pops of one counter take longer than their frames) */,
      const i32 *tr /* NULL, or [stream][12]: the transient pre-pass's records (oa_sh_transient_kernel) */,
      CeltCont *cconts /* NULL, or [stream]: the CELT layer's PVQ as a stage of its own -- a frame with one CELT pass stops before its PVQ, its stream goes on cut_list (cutq[1] counts) */,
      ShBackHdr *hdrs, int *cut_list, unsigned *cutq, unsigned *srt)
{
   extern __shared__ __attribute__((aligned(16))) char smem[];
   WV_LDS ShLds *L = (WV_LDS ShLds *)smem;
   for (int s = 0, end = 0;; s++) {
      if (s >= end) { int b0 = 0; if (wv_lane() == 0) b0 = (int)atomicAdd(counters + 2, (unsigned)chunk); s = wv_bcast(b0, 0); end = s + chunk; }
      if (s >= nstreams) break;
      if (threadIdx.x == 0) { L->packet_off = pkt_off; L->S.st_off = (i32)offsetof(SilkEncLds, st); }
      __syncthreads();
      if (conts[s].kind == SH_CONT_FAST) {
         OaShStream *gs = streams + s;
         const int ch = gs->cfg.channels;
         char *scr = scratch + (size_t)blockIdx.x * SH_SCRATCH_BYTES(frame_size, ch);
         const int cut = oa_sh_back_frame(L, gs, frame_size, out + (size_t)s * out_stride, out_stride, (i16 *)(pcm_hp_all + (size_t)s * SH_PCM_BYTES(frame_size, ch)),
               (i16 *)(scr + SH_PCM_BYTES(frame_size, ch)), (i16 *)(scr + 2 * SH_PCM_BYTES(frame_size, ch)), (CeltScratch *)(scr + 2 * SH_PCM_BYTES(frame_size, ch) + 512), conts + s, lens + s, rngs + s, tr ? tr + 12 * (size_t)s : nullptr,
               cconts ? cconts + s : (CeltCont *)0, cconts ? hdrs + s : (ShBackHdr *)0);
         if (cut) { if (threadIdx.x == 0) { const int key = oa_cut_key(SH_F(L)); cconts[s].sort_key = key; atomicAdd(srt + key, 1u); cut_list[atomicAdd(cutq + 1, 1u)] = s; } }
      }
      __syncthreads();
   }
}
/* the rest of the calls the back kernel cut before their PVQ (oa_celt_pvq_kernel in between) */
extern "C" __global__ void __launch_bounds__(64, OA_SH_BACK_WAVES_PER_EU)
oa_sh_back2_kernel(OaShStream *streams, int frame_size, u8 *out, int out_stride, char *scratch, i32 *lens, u32 *rngs, int pkt_off, const CeltCont *cconts, const ShBackHdr *hdrs, const int *cut_list, unsigned *cutq)
{
   extern __shared__ __attribute__((aligned(16))) char smem[];
   WV_LDS ShLds *L = (WV_LDS ShLds *)smem;
   const int n = (int)wv_uni((i32)cutq[1]);
   for (;;) {
      int k = 0;
      if (wv_lane() == 0) k = (int)atomicAdd(cutq + 3, 1u);
      k = wv_bcast(k, 0);
      if (k >= n) break;
      const int s = wv_uni(cut_list[k]);
      OaShStream *gs = streams + s;
      const int ch = gs->cfg.channels;
      char *scr = scratch + (size_t)blockIdx.x * SH_SCRATCH_BYTES(frame_size, ch);
      __syncthreads();
      oa_sh_back2_frame(L, gs, frame_size, out + (size_t)s * out_stride, out_stride, (CeltScratch *)(scr + 2 * SH_PCM_BYTES(frame_size, ch) + 512), hdrs + s, cconts + s, pkt_off, lens + s, rngs + s);
      __syncthreads();
   }
}

/* masking analysis of the surround multistream encoder: one wave per input channel (opus_surround.h) */
extern "C" __global__ void __launch_bounds__(64, 2)
oa_surround_kernel(const i16 *pcm, int len, int channels, int Fs, i32 *mem, i32 *preemph_mem, i32 *bandLogE)
{
   __shared__ SurroundLds lds;
   /* grid = encoders x channels (the classic entry point launches one encoder): encoder b's input, state and result rows */
   const int b = (int)blockIdx.x / channels, c = (int)blockIdx.x - b * channels;
   oa_surround_channel_wave((WV_LDS SurroundLds *)&lds, pcm + (size_t)b * len * channels, len, channels, c, Fs, mem + (size_t)b * channels * 120, preemph_mem + (size_t)b * channels,
         bandLogE + (size_t)b * channels * 21);
}

/* final-gather compaction: packet s (lens[s] bytes of its out slot) -> packed[offs[s] ...], one wave per packet */
extern "C" __global__ void __launch_bounds__(64)
oa_pack_kernel(const u8 *out, int stride, const i32 *lens, const long long *offs, u8 *packed, int n, long long capacity)
{
   const int s = blockIdx.x;
   if (s >= n) return;
   const long long o = offs[s];
   long long len = lens[s];
   if (o + len > capacity) len = capacity - o;            /* a fixed-size wire record: what does not fit is dropped (the receiver sees it from the lengths) */
   const u8 *src = out + (size_t)s * stride; u8 *dst = packed + o;
   for (int i = threadIdx.x; i < len; i += 64) dst[i] = src[i];
}
/* T consecutive frame-steps of every stream in one launch: the wave keeps a stream for T frames (pcm [T][S][frame*ch], out [T][S][stride], lens / rngs [T][S]) */
extern "C" __global__ void __launch_bounds__(64, OA_ENC_WAVES_PER_EU)
oa_encode_frames_kernel(OaStream *streams, const i16 *pcm, int frame_size, int T, int max_data_bytes, u8 *out, int out_stride, i32 *lens, u32 *rngs, int nstreams, CeltScratch *scratch, unsigned *queue)
{
   extern __shared__ __attribute__((aligned(16))) char smem[];
   WV_LDS FrameLds *L = (WV_LDS FrameLds *)smem;
   for (;;) {
      const int s = oa_queue_pop(queue);
      if (s >= nstreams) break;
      OaStream *gs = streams + s;
      const int ch = gs->cfg.channels;
      for (int t = 0; t < T; t++) {
         const size_t row = (size_t)t * nstreams + s;
         if (threadIdx.x == 0) L->g = scratch + blockIdx.x;
         __syncthreads();
         oa_encode_frame(L, gs, pcm + row * frame_size * ch, frame_size, max_data_bytes, out + row * out_stride, out_stride, lens + row, rngs + row);
         __syncthreads();
      }
   }
}

#include "opus_packet_host.h"

#define HIPCHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "opus_amd: %s failed: %s\n", #x, hipGetErrorString(e_)); return OPUS_INTERNAL_ERROR; } } while (0)

#include "opus_enc_host.h"
/* frame sizes an encode call accepts: 2.5, 5, 10, 20, 40, 60, 80, 100, 120 ms at the API rate (frame_size_select :845; the SILK-only application starts at 10 ms) */
/* bytes of the per-stream output slot a call needs: the largest packet the call can return -- a coded frame never exceeds 1276 bytes, but a hard-CBR call is
 * padded to its byte budget whatever the frame size, up to the 1276*6 the reference clamps max_data_bytes to -- and for calls above 20 ms (repacketised
 * multi-frame packets, src/opus_encoder.c:1698-1838) the 48-byte staging head-room of oa_multiframe_* (opus_multiframe.h) */
static opus_int32 oa_enc_out_stride_needed(opus_int32 Fs, int frame_size, opus_int32 max_data_bytes, int hard_cbr = 1)
{
   const int nf = frame_size > Fs / 50 ? (frame_size * 50 + Fs - 1) / Fs : 1;
   const opus_int32 mdb = max_data_bytes < 1276 * 6 ? max_data_bytes : 1276 * 6;          /* a single coded frame never gets more than 1276 * 6 bytes of budget (:1221) */
   if (nf > 1) return max_data_bytes > 0x7fffffff - 64 ? 0x7fffffff - 16 : max_data_bytes + 48;   /* multi-frame packets: with hard CBR / OPUS_BITRATE_MAX the reference pads the repacketised packet to the caller's WHOLE buffer, beyond 1276 * 6 (:1757, :1823), + the staging head-room */
   if (!hard_cbr) return mdb < 1276 ? mdb : 1276;                                        /* one VBR frame: TOC + at most 1275 bytes */
   return mdb;                                                                           /* one frame of a hard-CBR stream: padded to min(max_data_bytes, 1276 * 6) (:1330, :2646) */
}
static int oa_enc_frame_size_code(opus_int32 Fs, int application, int frame_size)
{
   return oa_frame_size_select(application, frame_size, OPUS_FRAMESIZE_ARG, Fs) == frame_size ? OPUS_OK : OPUS_BAD_ARG;
}

/* ---------------- batch object ---------------- */
struct OpusGpuEncBatch {
   int kind;                            /* 0: CELT-only kernel (OaStream), 1: SILK-capable kernel (OaShStream) */
   opus_int32 Fs; int application;
   OaShStream *d_sh;
   std::vector<OaShStream> h_sh;
   char *d_scratch; size_t scratch_cap;  /* per-wave HBM scratch of the frames in flight (CeltScratch; kind 1: SH_SCRATCH_BYTES) */
   unsigned *d_queue;                    /* the launch's stream queue (next unclaimed stream) */
   int num_cu;
   const void *occ_kernel; size_t occ_lds; int occ_per_cu;   /* last occupancy query (it is a host-side call per launch otherwise) */
   /* the split path of the SILK-capable encoder (opus_sh_split.h): per-stream continuation records, per-stream high-passed input, the calls handed to the one-kernel path */
   ShCont *d_cont; char *d_pcm_hp; size_t pcm_hp_cap; int *d_slow_list;
   int timing; int n_stamps; hipEvent_t stamp_ev[20]; const char *stamp_name[20];   /* OPUS_AMD_SET_KERNEL_TIMING: HIP events between the launches of the last call (opusgpu_enc_batch_kernel_times) */
   int pvq4_last;                                                           /* the last call launched oa_celt_pvq_kernel */
   ShBackHdr *d_back_hdr;                                                   /* SILK-capable batches: the back kernel's LDS header of the calls cut before their CELT pass's PVQ */
   unsigned *d_srt;                                                         /* [OA_SORT_KEYS] key counts, [OA_SORT_KEYS] fill counters of the PVQ kernel's sorted order */
   CeltCont *d_ccont; int *d_cut_list; int celt_pipe_last /* streams of the last pipelined call, 0 = the last call was not pipelined */;                                       /* CELT-only batches, kernel pipeline: per-stream continuation records, the list of the streams whose call was cut before the PVQ */
   i32 *d_tr; i16 *d_tr_scratch; size_t tr_scratch_cap;                      /* CELT-only batches: the transient pre-pass's records [S][4] and its per-wave scratch */
   struct { const void *kernel; size_t lds; int per_cu; } occ[12];
   int device;
   opus_int32 S;
   opus_int32 n_act;                     /* streams a call processes: the first n_act records (== S except under the classic API's call combiner) */
   int channels;
   hipStream_t stream;
   OaStream *d_streams;
   std::vector<OaStream> h_streams;     /* host mirror of the configuration (state is authoritative on device) */
   bool cfg_dirty;                      /* the host mirror changed since all_silk_pinned was derived */
   int any_fec;                         /* some stream of the mirror has in-band FEC on (-1: not derived since the mirror last changed): sizes the front kernel's packet window */
   int any_cbr;                         /* some stream of the mirror is hard CBR (-1: not derived since the mirror last changed): sizes the output slot a call needs */
   int all_silk_pinned;
   int tr_pre, pvq_stage;               /* OPUS_AMD_SET_TRANSIENT_PREPASS, OPUS_AMD_SET_PVQ_STAGE: -1 the library chooses, 0 off, 1 on (include/opus_amd.h) */
   int pipeline;                        /* OPUS_AMD_SET_KERNEL_PIPELINE: -1 the library chooses, 0 one kernel, 1 .. 4 the kernel pipeline (include/opus_amd.h) */
   /* staging for the host-pointer entry */
   opus_int16 *d_pcm; size_t pcm_cap;
   opus_int32 *d_apcm; size_t apcm_cap;  /* signal-domain copy of the input for the analysis (24-bit / float entry points) */
   unsigned char *d_out; size_t out_cap;
   opus_int32 *d_lens; opus_uint32 *d_rng;
};

extern "C" {

int opusgpu_device_count(void) { int n = 0; if (hipGetDeviceCount(&n) != hipSuccess) return 0; return n; }
#ifndef OA_SOURCE_HASH
#define OA_SOURCE_HASH "unknown"
#endif
const char *opusgpu_build_info(void) { return "OA_SRC_HASH=" OA_SOURCE_HASH; }
int opusgpu_enc_state_size(void) { return (int)sizeof(OaStream); }
int opusgpu_enc_sh_state_size(void) { return (int)sizeof(OaShStream); }
int opusgpu_sh_kernel_lds_bytes(void) { return (int)SH_LDS_BYTES(1); }
/* dynamic LDS of one wave: the SILK working set, or -- when the batch can reach the CELT layer (48 kHz) -- at least the CELT frame arena that aliases it */
static size_t sh_lds_bytes(int channels, int silk_only) { size_t n = SH_LDS_BYTES(channels); const size_t celt = SH_CELT_LDS_BYTES; if (!silk_only && celt > n) n = celt; return n; }
int opusgpu_kernel_lds_bytes(void) { return (int)sizeof(FrameLds); }
opus_int32 opusgpu_enc_batch_streams(const OpusGpuEncBatch *b) { return b ? b->S : 0; }

OpusGpuEncBatch *opusgpu_enc_batch_create(opus_int32 nstreams, opus_int32 Fs, int channels, int application, int device, int *error)
{
   int err = OPUS_OK;
   OpusGpuEncBatch *b = nullptr;
   OaStream proto;
   const int kind = oa_app_is_sh(application);
   OaShStream *shproto = kind ? new OaShStream : nullptr;
   if (nstreams <= 0) err = OPUS_BAD_ARG;
   if (err == OPUS_OK) err = kind ? sh_init_stream(shproto, Fs, channels, application) : oa_init_stream(&proto, Fs, channels, application);
   if (err == OPUS_OK) {
      int ndev = 0;
      if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0 || device < 0 || device >= ndev) {
         fprintf(stderr, "opus_amd: no usable HIP device (requested %d of %d) — this library has no CPU fallback\n", device, ndev);
         err = OPUS_INTERNAL_ERROR;
      }
   }
   if (err == OPUS_OK) {
      b = new OpusGpuEncBatch();
      b->device = device; b->S = nstreams; b->n_act = nstreams; b->channels = channels; b->cfg_dirty = true; b->all_silk_pinned = 0; b->any_cbr = -1; b->any_fec = -1; b->pipeline = -1; b->tr_pre = -1; b->pvq_stage = -1;
      b->kind = kind; b->Fs = Fs; b->application = application; b->d_sh = nullptr; b->d_scratch = nullptr; b->scratch_cap = 0; b->d_queue = nullptr; b->num_cu = 0; b->occ_kernel = nullptr; b->occ_lds = 0; b->occ_per_cu = 0;
      b->d_cont = nullptr; b->d_pcm_hp = nullptr; b->pcm_hp_cap = 0; b->d_slow_list = nullptr; memset(b->occ, 0, sizeof b->occ); b->d_tr = nullptr; b->d_tr_scratch = nullptr; b->tr_scratch_cap = 0; b->d_ccont = nullptr; b->d_cut_list = nullptr; b->celt_pipe_last = 0; b->d_back_hdr = nullptr; b->pvq4_last = 0; b->d_srt = nullptr; b->timing = 0; b->n_stamps = 0; memset(b->stamp_ev, 0, sizeof b->stamp_ev);
      b->d_pcm = nullptr; b->pcm_cap = 0; b->d_apcm = nullptr; b->apcm_cap = 0; b->d_out = nullptr; b->out_cap = 0; b->d_lens = nullptr; b->d_rng = nullptr; b->d_streams = nullptr; b->stream = nullptr;
      if (kind) b->h_sh.assign(nstreams, *shproto); else b->h_streams.assign(nstreams, proto);
      bool ok = hipSetDevice(device) == hipSuccess && hipStreamCreate(&b->stream) == hipSuccess &&
                (kind ? hipMalloc((void **)&b->d_sh, sizeof(OaShStream) * (size_t)nstreams) == hipSuccess &&
                        hipMemcpy(b->d_sh, b->h_sh.data(), sizeof(OaShStream) * (size_t)nstreams, hipMemcpyHostToDevice) == hipSuccess &&
                        hipFuncSetAttribute((const void *)oa_sh_encode_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) == hipSuccess
                      : true) &&
                hipMalloc((void **)&b->d_streams, sizeof(OaStream) * (size_t)(kind ? 1 : nstreams)) == hipSuccess &&
                hipMalloc((void **)&b->d_lens, sizeof(opus_int32) * (size_t)nstreams) == hipSuccess &&
                hipMalloc((void **)&b->d_rng, sizeof(opus_uint32) * (size_t)nstreams) == hipSuccess &&
                hipMalloc((void **)&b->d_queue, 128) == hipSuccess && hipMemset(b->d_queue, 0, 128) == hipSuccess &&
                hipDeviceGetAttribute(&b->num_cu, hipDeviceAttributeMultiprocessorCount, device) == hipSuccess &&
                (kind || hipMemcpy(b->d_streams, b->h_streams.data(), sizeof(OaStream) * (size_t)nstreams, hipMemcpyHostToDevice) == hipSuccess) &&
                hipFuncSetAttribute((const void *)oa_encode_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) == hipSuccess;
      if (!ok) { opusgpu_enc_batch_destroy(b); b = nullptr; err = OPUS_ALLOC_FAIL; }
   }
   delete shproto;
   if (error) *error = err;
   return b;
}
void opusgpu_enc_batch_destroy(OpusGpuEncBatch *b)
{
   if (!b) return;
   (void)hipSetDevice(b->device);
   if (b->stream) (void)hipStreamSynchronize(b->stream);
   if (b->d_streams) (void)hipFree(b->d_streams);
   if (b->d_sh) (void)hipFree(b->d_sh);
   if (b->d_scratch) (void)hipFree(b->d_scratch);
   if (b->d_queue) (void)hipFree(b->d_queue);
   if (b->d_ccont) (void)hipFree(b->d_ccont);
   if (b->d_cut_list) (void)hipFree(b->d_cut_list);
   if (b->d_back_hdr) (void)hipFree(b->d_back_hdr);
   if (b->d_srt) (void)hipFree(b->d_srt);
   for (int i = 0; i < 20; i++) if (b->stamp_ev[i]) (void)hipEventDestroy(b->stamp_ev[i]);
   if (b->d_cont) (void)hipFree(b->d_cont);
   if (b->d_pcm_hp) (void)hipFree(b->d_pcm_hp);
   if (b->d_slow_list) (void)hipFree(b->d_slow_list);
   if (b->d_tr) (void)hipFree(b->d_tr);
   if (b->d_tr_scratch) (void)hipFree(b->d_tr_scratch);
   if (b->d_pcm) (void)hipFree(b->d_pcm);
   if (b->d_apcm) (void)hipFree(b->d_apcm);
   if (b->d_out) (void)hipFree(b->d_out);
   if (b->d_lens) (void)hipFree(b->d_lens);
   if (b->d_rng) (void)hipFree(b->d_rng);
   if (b->stream) (void)hipStreamDestroy(b->stream);
   delete b;
}
/* configuration lives in the first bytes of each OaStream; ctl edits the host mirror and pushes only cfg (or the whole stream on RESET) */
int opusgpu_enc_batch_ctl(OpusGpuEncBatch *b, opus_int32 stream, int request, opus_int32 value)
{
   if (!b || stream < -1 || stream >= b->S) return OPUS_BAD_ARG;
   HIPCHECK(hipSetDevice(b->device));
   HIPCHECK(hipStreamSynchronize(b->stream));
   opus_int32 lo = stream < 0 ? 0 : stream, hi = stream < 0 ? b->S : stream + 1;
   if (request == OPUS_AMD_SET_TRANSIENT_PREPASS_REQUEST) { if (value < -1 || value > 1) return OPUS_BAD_ARG; b->tr_pre = value; return OPUS_OK; }     /* the launch's, not a stream's */
   if (request == OPUS_AMD_SET_PVQ_STAGE_REQUEST) { if (value < -1 || value > 1) return OPUS_BAD_ARG; b->pvq_stage = value; return OPUS_OK; }
   if (request == OPUS_AMD_SET_KERNEL_TIMING_REQUEST) { b->timing = value != 0; b->n_stamps = 0; return OPUS_OK; }      /* the launch's, not a stream's */
   if (request == OPUS_AMD_SET_KERNEL_PIPELINE_REQUEST) { if (value < -1 || value > 4) return OPUS_BAD_ARG; b->pipeline = value; return OPUS_OK; }   /* the launch's, not a stream's */
   b->any_cbr = -1; b->any_fec = -1;
   if (request == OPUS_RESET_STATE) {
      /* what the reference keeps across a reset (voice_ratio, the sticky force_channels, SILK's control structure) lives in the scalars, and those are the device's: bring
       * them into the mirror first (gathered on the device, one contiguous transfer) */
      const size_t n = (size_t)(hi - lo), w = b->kind ? sizeof(OaShScalars) : sizeof(OaEncScalars), pitch = b->kind ? sizeof(OaShStream) : sizeof(OaStream);
      const char *src = b->kind ? (const char *)&b->d_sh[lo].s : (const char *)&b->d_streams[lo].st.s;
      char *d_tmp = nullptr; std::vector<char> h_tmp(n * w);
      HIPCHECK(hipMalloc((void **)&d_tmp, n * w));
      hipError_t e_ = hipMemcpy2D(d_tmp, w, src, pitch, w, n, hipMemcpyDeviceToDevice);
      if (e_ == hipSuccess) e_ = hipMemcpy(h_tmp.data(), d_tmp, n * w, hipMemcpyDeviceToHost);
      (void)hipFree(d_tmp);
      if (e_ != hipSuccess) return OPUS_INTERNAL_ERROR;
      for (size_t i = 0; i < n; i++) { if (b->kind) memcpy(&b->h_sh[lo + i].s, h_tmp.data() + i * w, w); else memcpy(&b->h_streams[lo + i].st.s, h_tmp.data() + i * w, w); }
   }
   if (b->kind) {
      b->cfg_dirty = true;
      for (opus_int32 s = lo; s < hi; s++) { int r = sh_ctl_set(&b->h_sh[s], request, value); if (r != OPUS_OK) return r; }
      if (request == OPUS_RESET_STATE) HIPCHECK(hipMemcpy(b->d_sh + lo, &b->h_sh[lo], sizeof(OaShStream) * (size_t)(hi - lo), hipMemcpyHostToDevice));
      else HIPCHECK(hipMemcpy2D(&b->d_sh[lo].cfg, sizeof(OaShStream), &b->h_sh[lo].cfg, sizeof(OaShStream), sizeof(OaShConfig), (size_t)(hi - lo), hipMemcpyHostToDevice));
      return OPUS_OK;
   }
   for (opus_int32 s = lo; s < hi; s++) {
      int r = oa_ctl_set(&b->h_streams[s], request, value);
      if (r != OPUS_OK) return r;
   }
   /* one transfer for the whole range: full records on RESET, otherwise only the cfg prefix of every record (strided copy) */
   if (request == OPUS_RESET_STATE) HIPCHECK(hipMemcpy(b->d_streams + lo, &b->h_streams[lo], sizeof(OaStream) * (size_t)(hi - lo), hipMemcpyHostToDevice));
   else {
      HIPCHECK(hipMemcpy2D(&b->d_streams[lo].cfg, sizeof(OaStream), &b->h_streams[lo].cfg, sizeof(OaStream), sizeof(OaEncConfig), (size_t)(hi - lo), hipMemcpyHostToDevice));
      HIPCHECK(hipMemcpy2D(&b->d_streams[lo].Fs, sizeof(OaStream), &b->h_streams[lo].Fs, sizeof(OaStream), sizeof(opus_int32) * OA_STREAM_CFG2_WORDS, (size_t)(hi - lo), hipMemcpyHostToDevice));   /* DTX, signal type, ... live behind the state */
   }
   return OPUS_OK;
}
int opusgpu_enc_batch_get(OpusGpuEncBatch *b, opus_int32 stream, int request, opus_int32 *value)
{
   if (!b || stream < 0 || stream >= b->S) return OPUS_BAD_ARG;
   HIPCHECK(hipSetDevice(b->device));
   HIPCHECK(hipStreamSynchronize(b->stream));
   if (request == OPUS_AMD_GET_KERNEL_PIPELINE_REQUEST) { if (!value) return OPUS_BAD_ARG; *value = b->pipeline; return OPUS_OK; }
   if (request == OPUS_AMD_GET_TRANSIENT_PREPASS_REQUEST) { if (!value) return OPUS_BAD_ARG; *value = b->tr_pre; return OPUS_OK; }
   if (request == OPUS_AMD_GET_PVQ_STAGE_REQUEST) { if (!value) return OPUS_BAD_ARG; *value = b->pvq_stage; return OPUS_OK; }
   if (b->kind) {
      OaShStream *t = new OaShStream(b->h_sh[stream]);
      hipError_t e_ = request == OPUS_GET_IN_DTX_REQUEST ? hipMemcpy(t, &b->d_sh[stream], sizeof(OaShStream), hipMemcpyDeviceToHost)      /* needs the SILK channel counters */
                                                         : hipMemcpy(&t->s, &b->d_sh[stream].s, sizeof(OaShScalars), hipMemcpyDeviceToHost);
      const int r = e_ == hipSuccess ? sh_ctl_get(t, request, value) : OPUS_INTERNAL_ERROR;
      delete t;
      return r;
   }
   OaStream tmp = b->h_streams[stream];
   HIPCHECK(hipMemcpy(&tmp.st.s, &b->d_streams[stream].st.s, sizeof(OaEncScalars), hipMemcpyDeviceToHost));
   HIPCHECK(hipMemcpy(&tmp.nb_no_activity_ms_Q1, &b->d_streams[stream].nb_no_activity_ms_Q1, 3 * sizeof(opus_int32), hipMemcpyDeviceToHost));   /* DTX counter, peak energy, prev_framesize */
   return oa_ctl_get(&tmp, request, value);
}
int opusgpu_enc_batch_export_state(OpusGpuEncBatch *b, opus_int32 stream, void *blob)
{
   if (!b || !blob || stream < 0 || stream >= b->S) return OPUS_BAD_ARG;
   HIPCHECK(hipSetDevice(b->device));
   HIPCHECK(hipStreamSynchronize(b->stream));
   if (b->kind) HIPCHECK(hipMemcpy(blob, b->d_sh + stream, sizeof(OaShStream), hipMemcpyDeviceToHost));
   else HIPCHECK(hipMemcpy(blob, b->d_streams + stream, sizeof(OaStream), hipMemcpyDeviceToHost));
   return OPUS_OK;
}
int opusgpu_enc_batch_import_state(OpusGpuEncBatch *b, opus_int32 stream, const void *blob)
{
   if (!b || !blob || stream < 0 || stream >= b->S) return OPUS_BAD_ARG;
   if (b->kind) {
      const OaShStream *src = (const OaShStream *)blob;
      if (src->cfg.channels != b->channels || src->cfg.Fs != b->Fs) return OPUS_BAD_ARG;
      if ((src->cfg.application == OPUS_APPLICATION_RESTRICTED_SILK) != (b->application == OPUS_APPLICATION_RESTRICTED_SILK)) return OPUS_BAD_ARG;      /* (as in opusgpu_enc_batch_copy_states) */
      HIPCHECK(hipSetDevice(b->device));
      HIPCHECK(hipStreamSynchronize(b->stream));
      b->h_sh[stream] = *src; b->cfg_dirty = true; b->any_cbr = -1; b->any_fec = -1;
      HIPCHECK(hipMemcpy(b->d_sh + stream, src, sizeof(OaShStream), hipMemcpyHostToDevice));
      return OPUS_OK;
   }
   const OaStream *src = (const OaStream *)blob;
   if (src->cfg.channels != b->channels) return OPUS_BAD_ARG;
   HIPCHECK(hipSetDevice(b->device));
   HIPCHECK(hipStreamSynchronize(b->stream));
   b->h_streams[stream] = *src; b->any_cbr = -1; b->any_fec = -1;
   HIPCHECK(hipMemcpy(b->d_streams + stream, src, sizeof(OaStream), hipMemcpyHostToDevice));
   return OPUS_OK;
}
/* n stream records (configuration AND state, as they stand on the device) from one batch into another of the same shape, device to device: a service that moves streams
 * between batches, or fans a warmed-up stream out (bench.py's steady-state leg) -- what export_state + import_state do per stream through host memory */
int opusgpu_enc_batch_copy_states(OpusGpuEncBatch *dst, opus_int32 dst_first, OpusGpuEncBatch *src, opus_int32 src_first, opus_int32 n)
{
   if (!dst || !src || n < 0 || dst_first < 0 || src_first < 0 || dst_first + n > dst->S || src_first + n > src->S) return OPUS_BAD_ARG;
   if (dst->kind != src->kind || dst->channels != src->channels || dst->Fs != src->Fs || dst->device != src->device) return OPUS_BAD_ARG;
   /* a RESTRICTED_SILK batch launches its back kernel without the CELT arena (oa_sh_encode_split: lds_back): records of the other SILK-capable applications do not go there, nor its own elsewhere */
   if (dst->kind && (dst->application == OPUS_APPLICATION_RESTRICTED_SILK) != (src->application == OPUS_APPLICATION_RESTRICTED_SILK)) return OPUS_BAD_ARG;
   if (dst == src && dst_first < src_first + n && src_first < dst_first + n && n > 0) return OPUS_BAD_ARG;                     /* overlapping ranges of one batch */
   HIPCHECK(hipSetDevice(dst->device));
   HIPCHECK(hipStreamSynchronize(src->stream)); HIPCHECK(hipStreamSynchronize(dst->stream));
   if (n == 0) return OPUS_OK;
   if (dst->kind) {
      HIPCHECK(hipMemcpy(dst->d_sh + dst_first, src->d_sh + src_first, sizeof(OaShStream) * (size_t)n, hipMemcpyDeviceToDevice));
      for (opus_int32 i = 0; i < n; i++) dst->h_sh[dst_first + i] = src->h_sh[src_first + i];
      dst->cfg_dirty = true;
   } else {
      HIPCHECK(hipMemcpy(dst->d_streams + dst_first, src->d_streams + src_first, sizeof(OaStream) * (size_t)n, hipMemcpyDeviceToDevice));
      for (opus_int32 i = 0; i < n; i++) dst->h_streams[dst_first + i] = src->h_streams[src_first + i];
   }
   dst->any_cbr = -1; dst->any_fec = -1;
   return OPUS_OK;
}
/* calls the split path's front kernel has kept / handed to the one-kernel path since the batch was created (diagnostics, tests) */
int opusgpu_enc_batch_split_stats(OpusGpuEncBatch *b, opus_uint32 *kept, opus_uint32 *declined)
{
   if (!b || !kept || !declined) return OPUS_BAD_ARG;
   HIPCHECK(hipSetDevice(b->device));
   HIPCHECK(hipStreamSynchronize(b->stream));
   unsigned v[2] = {0, 0};
   if (!b->kind) {      /* a CELT-only batch: the LAST call -- the streams it cut before the PVQ (oa_celt_pvq_kernel coded their bands), and those the encode kernel kept whole */
      if (!b->d_ccont || !b->celt_pipe_last) { *kept = 0; *declined = 0; return OPUS_OK; }
      HIPCHECK(hipMemcpy(v, b->d_queue + 1, sizeof(unsigned), hipMemcpyDeviceToHost));
      *kept = v[0]; *declined = (opus_uint32)b->celt_pipe_last - v[0];
      return OPUS_OK;
   }
   HIPCHECK(hipMemcpy(v, b->d_queue + 16, sizeof v, hipMemcpyDeviceToHost));
   *kept = v[0]; *declined = v[1];
   return OPUS_OK;
}
int opusgpu_enc_batch_kernel_times(OpusGpuEncBatch *b, char *names, int names_cap, float *ms, int max_kernels)
{
   if (!b || !names || !ms || names_cap < 1 || max_kernels < 0) return OPUS_BAD_ARG;
   names[0] = 0;
   if (b->n_stamps < 2) return 0;
   HIPCHECK(hipSetDevice(b->device));
   HIPCHECK(hipEventSynchronize(b->stamp_ev[b->n_stamps - 1]));
   int n = 0; size_t used = 0;
   for (int i = 1; i < b->n_stamps && n < max_kernels; i++) {
      float t = 0;
      HIPCHECK(hipEventElapsedTime(&t, b->stamp_ev[i - 1], b->stamp_ev[i]));
      const size_t l = strlen(b->stamp_name[i]);
      if (used + l + 2 > (size_t)names_cap) break;
      if (n) names[used++] = ',';
      memcpy(names + used, b->stamp_name[i], l); used += l; names[used] = 0;
      ms[n++] = t;
   }
   return n;
}
int opusgpu_enc_batch_pvq_stage_stats(OpusGpuEncBatch *b, opus_uint32 *streams)
{
   if (!b || !streams) return OPUS_BAD_ARG;
   *streams = 0;
   if (!b->d_ccont || !b->pvq4_last) return OPUS_OK;
   HIPCHECK(hipSetDevice(b->device));
   HIPCHECK(hipStreamSynchronize(b->stream));
   unsigned v = 0;
   HIPCHECK(hipMemcpy(&v, b->d_queue + (b->kind ? 25 : 1), sizeof v, hipMemcpyDeviceToHost));
   *streams = v;
   return OPUS_OK;
}
int opusgpu_enc_batch_reset(OpusGpuEncBatch *b) { return opusgpu_enc_batch_ctl(b, -1, OPUS_RESET_STATE, 0); }
int opusgpu_enc_batch_sync(OpusGpuEncBatch *b) { if (!b) return OPUS_BAD_ARG; HIPCHECK(hipSetDevice(b->device)); HIPCHECK(hipStreamSynchronize(b->stream)); return OPUS_OK; }

/* OPUS_AMD_SET_KERNEL_TIMING(1): an event on the launch stream before the call's first launch and after each of its launches; opusgpu_enc_batch_kernel_times reads the last call's */
static void oa_stamp(OpusGpuEncBatch *b, hipStream_t s, const char *name)
{
   if (!b->timing || b->n_stamps >= 20) return;
   const int i = b->n_stamps++;
   if (!b->stamp_ev[i] && hipEventCreate(&b->stamp_ev[i]) != hipSuccess) { b->stamp_ev[i] = nullptr; b->n_stamps = i; return; }
   b->stamp_name[i] = name;
   (void)hipEventRecord(b->stamp_ev[i], s);
}
static void oa_stamp_begin(OpusGpuEncBatch *b, hipStream_t s) { b->n_stamps = 0; oa_stamp(b, s, "begin"); }
/* grid of a persistent encoder launch = the waves the chip holds at this kernel's register / LDS footprint (never more than there are streams); makes sure the
 * per-wave scratch covers it and resets the stream queue on the launch's HIP stream */
static int oa_persistent_grid(OpusGpuEncBatch *b, const void *kernel, size_t lds_bytes, size_t scratch_per_wave, hipStream_t s, int *grid_out)
{
   int per_cu = b->occ_per_cu;
   if (b->occ_kernel != kernel || b->occ_lds != lds_bytes) {
      HIPCHECK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kernel, 64, lds_bytes));
      if (per_cu < 1) per_cu = 1;
      b->occ_kernel = kernel; b->occ_lds = lds_bytes; b->occ_per_cu = per_cu;
   }
   long long grid = (long long)per_cu * (b->num_cu > 0 ? b->num_cu : 1);
   static const int grid_env = getenv("OPUS_AMD_GRID") ? atoi(getenv("OPUS_AMD_GRID")) : 0;                 /* experiments only */
   if (grid_env > 0) grid = grid_env;
   if (grid > b->n_act) grid = b->n_act;
   const size_t need = (size_t)grid * scratch_per_wave;
   if (need > b->scratch_cap) { HIPCHECK(hipStreamSynchronize(s)); if (b->d_scratch) (void)hipFree(b->d_scratch); b->d_scratch = nullptr; b->scratch_cap = 0; HIPCHECK(hipMalloc((void **)&b->d_scratch, need)); b->scratch_cap = need; }
   HIPCHECK(hipMemsetAsync(b->d_queue, 0, sizeof(unsigned), s));
   *grid_out = (int)grid;
   return OPUS_OK;
}
/* One encode call of a SILK-capable batch as the kernel pipeline of opus_sh_split.h, all on HIP stream s: front (every stream) -> quantiser (16 streams per wave) ->
 * back (every stream the front kernel kept) -> the one-kernel path over the streams it turned away (no wave finds work there when there are none).  The launches are
 * persistent: each grid is what the chip holds of its kernel, the per-wave scratch is shared between them (they run one after the other). */
static int oa_sh_grid(OpusGpuEncBatch *b, int slot, const void *kernel, size_t lds, long long work_items, int *grid_out)
{
   if (b->occ[slot].kernel != kernel || b->occ[slot].lds != lds) {
      int per_cu = 0;
#ifdef OA_PHASE_TIMERS
      HIPCHECK(hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 159 * 1024));      /* (the profiling build's kernels carry static LDS counters) */
#else
      HIPCHECK(hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
#endif
      HIPCHECK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kernel, 64, lds));
      b->occ[slot].kernel = kernel; b->occ[slot].lds = lds; b->occ[slot].per_cu = per_cu < 1 ? 1 : per_cu;
   }
   long long grid = (long long)b->occ[slot].per_cu * (b->num_cu > 0 ? b->num_cu : 1);
   if (grid > work_items) grid = work_items;
   *grid_out = (int)(grid < 1 ? 1 : grid);
   return OPUS_OK;
}
static int oa_sh_encode_split(OpusGpuEncBatch *b, const opus_int16 *d_pcm, const opus_int32 *d_apcm, int frame_size, unsigned char *d_out, opus_int32 out_stride, opus_int32 max_data_bytes,
      opus_int32 *d_lens, opus_uint32 *d_final_range, hipStream_t s, size_t lds_full, int silk_only, int mode, int pcm_row)
{
   const int n = (int)b->n_act, ch = b->channels;
   if (!b->d_cont) {
      HIPCHECK(hipMalloc((void **)&b->d_cont, sizeof(ShCont) * (size_t)b->S));
      HIPCHECK(hipMalloc((void **)&b->d_slow_list, 3 * sizeof(int) * (size_t)b->S));         /* [S] the calls the front kernel turned away | [2 S] the pred kernel's work list (coded channels) */
   }
   { const size_t need = SH_PCM_BYTES(frame_size, ch) * (size_t)b->S; if (need > b->pcm_hp_cap) { HIPCHECK(hipStreamSynchronize(s)); if (b->d_pcm_hp) (void)hipFree(b->d_pcm_hp); b->d_pcm_hp = nullptr; b->pcm_hp_cap = 0; HIPCHECK(hipMalloc((void **)&b->d_pcm_hp, need)); b->pcm_hp_cap = need; } }
   static const size_t lds_pad = getenv("OPUS_AMD_SH_LDS_PAD") ? (size_t)atoi(getenv("OPUS_AMD_SH_LDS_PAD")) : 0;   /* occupancy experiments only */
   /* the front kernel holds the SILK state without the quantiser tails; the arena behind the head must also hold the tonality analysis' working set */
   /* every kernel's packet buffer sits behind the rest of its LDS (ShLds.packet_off): the full OA_MAX_PACKET + 4 bytes, or the front kernel's few header bytes */
   auto al16 = [](size_t v) { return (v + 15) & ~(size_t)15; };
   size_t lds_front = SH_FRONT_LDS_BYTES(ch) + lds_pad;
   if (lds_front < offsetof(ShLds, S) + offsetof(SilkEncLds, u) + sizeof(AnLds)) lds_front = offsetof(ShLds, S) + offsetof(SilkEncLds, u) + sizeof(AnLds);
   /* a batch with in-band FEC on somewhere: the front kernel codes the previous packet's LBRR side stream at the head of the packet, which can be most of a packet: full window */
   if (b->any_fec < 0) { int f = 0; for (opus_int32 i = 0; !f && i < b->S; i++) f = b->h_sh[i].cfg.use_inband_fec != 0; b->any_fec = f; }
   const int pkt_window = (b->any_fec || frame_size * 50 > b->Fs) && mode != 2 /* (the one-wave-per-stream reference quantiser of value 2 has no LBRR pass: those calls stay on the one-kernel path) */ ? (int)SH_PKT_BYTES : (int)SH_FRONT_PKT_BYTES;
   const int po_front = (int)al16(lds_front); lds_front = po_front + pkt_window;
   /* the back kernel enters the CELT arena for hybrid frames AND for the redundant CELT frame that announces a SILK bandwidth switch (opus_encoder.c:2251-2260), which a
    * stream pinned to SILK-only can still ask for: only RESTRICTED_SILK never does */
   size_t lds_back = b->application == OPUS_APPLICATION_RESTRICTED_SILK ? offsetof(ShLds, S) + 256 : SH_CELT_LDS_BYTES;
   const int po_back = (int)al16(lds_back); lds_back = po_back + SH_PKT_BYTES;
   const int po_full = (int)al16(lds_full); lds_full = po_full + SH_PKT_BYTES;
   const void *kq = mode == 2 ? (const void *)oa_sh_quant0_kernel : (const void *)oa_sh_quant_kernel;
   const size_t lds_q = mode == 2 ? lds_full : sizeof(SqLds), scr_q = mode == 2 ? sizeof(SeRateScratch) : SQ_WAVE_SCRATCH_BYTES;
   const int pred_split = mode >= 3;                                    /* front -> pred -> quantiser -> back (3: one wave per coded channel; 4: the stage cut into lane and wave kernels) */
   const size_t lds_pa = (size_t)PL_A_WORDS * 4 * PL_STREAMS, lds_pb = sizeof(PlBLane) * PL_STREAMS + 2 * sizeof(SeNlsfTabs);
   int g_front = 0, g_quant = 0, g_back = 0, g_slow = 0, g_pred = 0;
   { int r = oa_sh_grid(b, 0, (const void *)oa_sh_front_kernel, lds_front, n, &g_front); if (r != OPUS_OK) return r; }
   { int r = oa_sh_grid(b, 1, kq, lds_q, mode == 2 ? n : (n + 15) / 16, &g_quant); if (r != OPUS_OK) return r; }
   { int r = oa_sh_grid(b, 2, (const void *)oa_sh_back_kernel, lds_back, n, &g_back); if (r != OPUS_OK) return r; }
   { int r = oa_sh_grid(b, 3, (const void *)oa_sh_encode_kernel, lds_full, n, &g_slow); if (r != OPUS_OK) return r; }
   int g_pa = 0, g_pb = 0, g_pc = 0;
   if (pred_split) { int r = oa_sh_grid(b, 4, (const void *)oa_sh_pred_kernel, sizeof(PredLds), n * ch, &g_pred); if (r != OPUS_OK) return r; }
   if (mode == 4) {
      const long long tiles = ((long long)n * ch + PL_STREAMS - 1) / PL_STREAMS;
      int r = oa_sh_grid(b, 5, (const void *)oa_sh_preda_kernel, lds_pa, tiles, &g_pa); if (r != OPUS_OK) return r;
      r = oa_sh_grid(b, 6, (const void *)oa_sh_predb_kernel, lds_pb, tiles, &g_pb); if (r != OPUS_OK) return r;
      r = oa_sh_grid(b, 7, (const void *)oa_sh_predc_kernel, sizeof(PredLds), n * ch, &g_pc); if (r != OPUS_OK) return r;
   }
   size_t need = (size_t)g_front * sizeof(CeltScratch);
   if ((size_t)g_quant * scr_q > need) need = (size_t)g_quant * scr_q;
   if ((size_t)g_back * SH_SCRATCH_BYTES(frame_size, ch) > need) need = (size_t)g_back * SH_SCRATCH_BYTES(frame_size, ch);
   if ((size_t)g_slow * SH_SCRATCH_BYTES(frame_size, ch) > need) need = (size_t)g_slow * SH_SCRATCH_BYTES(frame_size, ch);
   if (need > b->scratch_cap) { HIPCHECK(hipStreamSynchronize(s)); if (b->d_scratch) (void)hipFree(b->d_scratch); b->d_scratch = nullptr; b->scratch_cap = 0; HIPCHECK(hipMalloc((void **)&b->d_scratch, need)); b->scratch_cap = need; }
   HIPCHECK(hipMemsetAsync(b->d_queue, 0, 64, s));
   oa_stamp_begin(b, s);
   hipLaunchKernelGGL(oa_sh_front_kernel, dim3((unsigned)g_front), dim3(64), lds_front, s,
         b->d_sh, (const i16 *)d_pcm, (const i32 *)d_apcm, frame_size, (int)max_data_bytes, b->d_pcm_hp, (CeltScratch *)b->d_scratch, b->d_cont, b->d_slow_list, b->d_queue, n, po_front, pcm_row, mode == 4 ? 2 : pred_split, pkt_window);
   oa_stamp(b, s, "oa_sh_front_kernel");
   /* a 40 / 60 ms call: the SILK-only packets among its streams code two / three 20 ms frames on one coder -- the pred / quantiser stages run once per frame, with
    * oa_sh_frontc_kernel (the next frame's analysis) in between; every other stream of such a launch was handed to the one-kernel path by the front kernel */
   const int nblk = frame_size * 50 > b->Fs ? frame_size * 50 / b->Fs : 1;
   for (int blk = 0; blk < nblk; blk++) {
   if (blk > 0) {
      HIPCHECK(hipMemsetAsync(b->d_queue, 0, 2 * sizeof(unsigned), s)); HIPCHECK(hipMemsetAsync(b->d_queue + 5, 0, 2 * sizeof(unsigned), s));
      hipLaunchKernelGGL(oa_sh_frontc_kernel, dim3((unsigned)g_front), dim3(64), lds_front, s, b->d_sh, frame_size, b->d_pcm_hp, b->d_cont, b->d_slow_list, b->d_queue, n, po_front, mode == 4 ? 2 : pred_split, blk);
      oa_stamp(b, s, "oa_sh_frontc_kernel");
   }
   if (mode == 4) {
      const int *pl = (const int *)(b->d_slow_list + n); const unsigned *pc = (const unsigned *)(b->d_queue + 6);
      hipLaunchKernelGGL(oa_sh_preda_kernel, dim3((unsigned)g_pa), dim3(64), lds_pa, s, b->d_cont, pl, pc);
      hipLaunchKernelGGL(oa_sh_predc_kernel, dim3((unsigned)g_pc), dim3(64), sizeof(PredLds), s, b->d_cont, pl, pc);
      hipLaunchKernelGGL(oa_sh_predb_kernel, dim3((unsigned)g_pb), dim3(64), lds_pb, s, b->d_sh, b->d_cont, pl, pc);
      oa_stamp(b, s, "oa_sh_preda_kernel+oa_sh_predc_kernel+oa_sh_predb_kernel");
   }
   if (pred_split) hipLaunchKernelGGL(oa_sh_pred_kernel, dim3((unsigned)g_pred), dim3(64), sizeof(PredLds), s, b->d_sh, b->d_cont, (const int *)(b->d_slow_list + n), (const unsigned *)(b->d_queue + 6), b->d_queue + 5, mode == 4 ? 1 : 0);
   if (mode == 2) hipLaunchKernelGGL(oa_sh_quant0_kernel, dim3((unsigned)g_quant), dim3(64), lds_q, s, b->d_sh, b->d_cont, n, b->d_scratch, b->d_queue, po_full);
   else hipLaunchKernelGGL(oa_sh_quant_kernel, dim3((unsigned)g_quant), dim3(64), lds_q, s, b->d_sh, b->d_cont, n, b->d_scratch, b->d_queue);
   oa_stamp(b, s, pred_split ? "oa_sh_pred_kernel+oa_sh_quant_kernel" : "oa_sh_quant_kernel");
   }
   /* the CELT layer's transient recursions on lanes first (48 kHz, AUDIO / VOIP: frames with a CELT layer); OPUS_AMD_TR_PRE=0 keeps them in the back kernel */
   static const int tr_env = getenv("OPUS_AMD_TR_PRE") ? atoi(getenv("OPUS_AMD_TR_PRE")) : 1;               /* (process default behind OPUS_AMD_SET_TRANSIENT_PREPASS(-1)) */
   const int tr_on = b->tr_pre >= 0 ? b->tr_pre : tr_env;
   const i32 *d_tr = nullptr;
   if (tr_on && b->Fs == 48000 && b->application != OPUS_APPLICATION_RESTRICTED_SILK && !silk_only && nblk == 1 /* (a 40 / 60 ms launch keeps SILK-only packets only: no CELT layer to prepare) */) {
      const int items = n * ch, tiles = (items + 63) / 64;
      const int g = tiles < 8 * (b->num_cu > 0 ? b->num_cu : 1) ? tiles : 8 * (b->num_cu > 0 ? b->num_cu : 1);
      const size_t need_tr = (size_t)g * (OA_MAX_FRAME + OA_OVERLAP) * 64 * sizeof(i16);
      if (!b->d_tr) { HIPCHECK(hipMalloc((void **)&b->d_tr, (size_t)b->S * 12 * sizeof(i32))); HIPCHECK(hipMemsetAsync(b->d_tr, 0, (size_t)b->S * 12 * sizeof(i32), s)); }
      if (need_tr > b->tr_scratch_cap) { HIPCHECK(hipStreamSynchronize(s)); if (b->d_tr_scratch) (void)hipFree(b->d_tr_scratch); b->d_tr_scratch = nullptr; b->tr_scratch_cap = 0; HIPCHECK(hipMalloc((void **)&b->d_tr_scratch, need_tr)); b->tr_scratch_cap = need_tr; }
      hipLaunchKernelGGL(oa_sh_transient_kernel, dim3((unsigned)g), dim3(64), 0, s, (const OaShStream *)b->d_sh, (const ShCont *)b->d_cont, (const char *)b->d_pcm_hp, frame_size, ch, items, b->d_tr_scratch, b->d_tr);
      oa_stamp(b, s, "oa_sh_transient_kernel");
      d_tr = b->d_tr;
   }
   /* the CELT layer's PVQ as a stage of its own (celt_enc_pvq4.h: four streams per wave) wherever the launch can carry CELT frames; OPUS_AMD_SH_PVQ4=0: inside the back kernel */
   static const int pvq4_env = getenv("OPUS_AMD_SH_PVQ4") ? atoi(getenv("OPUS_AMD_SH_PVQ4")) : 1;          /* (process default behind OPUS_AMD_SET_PVQ_STAGE(-1)) */
   const bool pvq4 = (b->pvq_stage >= 0 ? b->pvq_stage : (pvq4_env && (pvq4_env > 1 || n > (long long)b->occ[2].per_cu * (b->num_cu > 0 ? b->num_cu : 1)))) /* -1: where the back kernel needs more than one round of the chip (profiles/r06_j) */ && !silk_only && b->application != OPUS_APPLICATION_RESTRICTED_SILK;
   unsigned *cutq = b->d_queue + 24;
   b->pvq4_last = pvq4;
   if (pvq4) {
      if (!b->d_ccont) {
         HIPCHECK(hipMalloc((void **)&b->d_ccont, sizeof(CeltCont) * (size_t)b->S));
         HIPCHECK(hipMalloc((void **)&b->d_cut_list, 2 * sizeof(int) * (size_t)b->S));
         HIPCHECK(hipMalloc((void **)&b->d_back_hdr, sizeof(ShBackHdr) * (size_t)b->S));
         HIPCHECK(hipMalloc((void **)&b->d_srt, 2 * OA_SORT_KEYS * sizeof(unsigned)));
      }
      HIPCHECK(hipMemsetAsync(cutq, 0, 4 * sizeof(unsigned), s));
      HIPCHECK(hipMemsetAsync(b->d_srt, 0, 2 * OA_SORT_KEYS * sizeof(unsigned), s));
   }
   hipLaunchKernelGGL(oa_sh_back_kernel, dim3((unsigned)g_back), dim3(64), lds_back, s,
         b->d_sh, frame_size, (u8 *)d_out, (int)out_stride, b->d_pcm_hp, b->d_scratch, (const ShCont *)b->d_cont, (i32 *)d_lens, (u32 *)d_final_range, n, b->d_queue, po_back, silk_only ? 8 : 1, d_tr,
         pvq4 ? b->d_ccont : (CeltCont *)nullptr, b->d_back_hdr, b->d_cut_list, cutq, b->d_srt);
   oa_stamp(b, s, "oa_sh_back_kernel");
   if (pvq4) {
      int g_pvq = 0;
      { const int r = oa_sh_grid(b, 8, (const void *)oa_celt_pvq_kernel, sizeof(P4Lds), ((long long)n + 3) / 4, &g_pvq); if (r != OPUS_OK) return r; }
      { const int gs_ = (n + 63) / 64; hipLaunchKernelGGL(oa_celt_sort_kernel, dim3((unsigned)(gs_ < 1024 ? gs_ : 1024)), dim3(64), 0, s, (const CeltCont *)b->d_ccont, (const int *)b->d_cut_list, b->d_cut_list + b->S, (const unsigned *)cutq, b->d_srt); }
      oa_stamp(b, s, "oa_celt_sort_kernel");
      hipLaunchKernelGGL(oa_celt_pvq_kernel, dim3((unsigned)g_pvq), dim3(64), sizeof(P4Lds), s, b->d_ccont, (const int *)(b->d_cut_list + b->S), cutq);
      oa_stamp(b, s, "oa_celt_pvq_kernel");
      hipLaunchKernelGGL(oa_sh_back2_kernel, dim3((unsigned)g_back), dim3(64), lds_back, s, b->d_sh, frame_size, (u8 *)d_out, (int)out_stride, b->d_scratch, (i32 *)d_lens, (u32 *)d_final_range, po_back,
            (const CeltCont *)b->d_ccont, (const ShBackHdr *)b->d_back_hdr, (const int *)b->d_cut_list, cutq);
      oa_stamp(b, s, "oa_sh_back2_kernel");
   }
   hipLaunchKernelGGL(oa_sh_encode_kernel, dim3((unsigned)g_slow), dim3(64), lds_full, s,
         b->d_sh, (const i16 *)d_pcm, (const i32 *)d_apcm, frame_size, (int)max_data_bytes, (u8 *)d_out, (int)out_stride, b->d_scratch, (i32 *)d_lens, (u32 *)d_final_range, n, b->d_queue + 3,
         (const int *)b->d_slow_list, (const unsigned *)(b->d_queue + 4), po_full, pcm_row, 0, 1, (const i32 *)nullptr);
   oa_stamp(b, s, "oa_sh_encode_kernel(declined calls)");
   HIPCHECK(hipGetLastError());
   return OPUS_OK;
}
static int oa_batch_any_cbr(OpusGpuEncBatch *b)
{
   if (b->any_cbr < 0) {
      int any = 0;
      for (opus_int32 i = 0; !any && i < b->n_act; i++) any = b->kind ? !b->h_sh[i].cfg.use_vbr : !b->h_streams[i].cfg.use_vbr;
      b->any_cbr = any;
   }
   return b->any_cbr;
}
/* d_apcm (may be NULL): the same samples in the encoder's signal domain (int32, Q12 below int16 full scale: src/opus_encoder.c FLOAT2SIG / INT24TOSIG), which the
 * 24-bit and float entry points hand to the analysis instead of the rounded int16 samples (downmix_int24 :804, downmix_float :748) */
int opusgpu_encode_batch_lookahead_dev(OpusGpuEncBatch *b, const opus_int16 *d_pcm, const opus_int32 *d_apcm, int frame_size, int analysis_frame_size, unsigned char *d_out,
      opus_int32 out_stride, opus_int32 max_data_bytes, opus_int32 *d_lens, opus_uint32 *d_final_range, void *hip_stream);
int opusgpu_encode_batch_dev_sig(OpusGpuEncBatch *b, const opus_int16 *d_pcm, const opus_int32 *d_apcm, int frame_size, unsigned char *d_out,
      opus_int32 out_stride, opus_int32 max_data_bytes, opus_int32 *d_lens, opus_uint32 *d_final_range, void *hip_stream)
{
   return opusgpu_encode_batch_lookahead_dev(b, d_pcm, d_apcm, frame_size, frame_size, d_out, out_stride, max_data_bytes, d_lens, d_final_range, hip_stream);
}
/* The same with the reference's look-ahead (src/opus_encoder.c:1247, :2662-2690; src/analysis.c:954): every stream's row of d_pcm (and d_apcm) holds analysis_frame_size >=
 * frame_size samples per channel -- what opus_encode() is handed when OPUS_SET_EXPERT_FRAME_DURATION selects a frame shorter than the caller's buffer -- of which the first
 * frame_size are coded; the tonality analysis runs over the whole row (the part it has not seen yet: OaAnalysis.analysis_offset carries that between calls) */
static int oa_encode_launch(OpusGpuEncBatch *b, const opus_int16 *d_pcm, const opus_int32 *d_apcm, int frame_size, int analysis_frame_size, unsigned char *d_out,
      opus_int32 out_stride, opus_int32 max_data_bytes, opus_int32 *d_lens, opus_uint32 *d_final_range, void *hip_stream, int first, int stride, opus_int32 count, const opus_int32 *d_budget);
int opusgpu_encode_batch_lookahead_dev(OpusGpuEncBatch *b, const opus_int16 *d_pcm, const opus_int32 *d_apcm, int frame_size, int analysis_frame_size, unsigned char *d_out,
      opus_int32 out_stride, opus_int32 max_data_bytes, opus_int32 *d_lens, opus_uint32 *d_final_range, void *hip_stream)
{
   if (!b) return OPUS_BAD_ARG;
   return oa_encode_launch(b, d_pcm, d_apcm, frame_size, analysis_frame_size, d_out, out_stride, max_data_bytes, d_lens, d_final_range, hip_stream, 0, 1, b->n_act, nullptr);
}
/* the launch itself.  first / stride / count: the call's streams are the records first, first + stride, ... (count of them; inputs, outputs and d_budget are indexed by the
 * RECORD, as always); d_budget: per record, this call's max_data_bytes for it (<= 0: the stream sits the call out) -- both for the multistream batch's chained byte budgets
 * (opus_ms_batch.h), which step through the streams of all encoders in order; such a call takes the one-kernel path */
static int oa_encode_launch(OpusGpuEncBatch *b, const opus_int16 *d_pcm, const opus_int32 *d_apcm, int frame_size, int analysis_frame_size, unsigned char *d_out,
      opus_int32 out_stride, opus_int32 max_data_bytes, opus_int32 *d_lens, opus_uint32 *d_final_range, void *hip_stream, int first, int stride, opus_int32 count, const opus_int32 *d_budget)
{
   if (!b || !d_pcm || !d_out || !d_lens || !d_final_range) return OPUS_BAD_ARG;
   if (analysis_frame_size < frame_size) return OPUS_BAD_ARG;
   const bool subset = first != 0 || stride != 1 || count != b->n_act || d_budget != nullptr;
   if (count <= 0 || stride < 1 || first < 0 || (long long)first + (long long)(count - 1) * stride >= b->S) return OPUS_BAD_ARG;
   const opus_int32 n_act_saved = b->n_act;
   struct Restore { OpusGpuEncBatch *b; opus_int32 n; ~Restore() { b->n_act = n; } } restore_{b, n_act_saved};
   b->n_act = count;                                                        /* (the grid and the kernels' stream count follow the call's streams) */
   const int pcm_row = analysis_frame_size;
   { const int fr = oa_enc_frame_size_code(b->Fs, b->application, frame_size); if (fr != OPUS_OK) return fr; }
   if (max_data_bytes <= 0) return OPUS_BAD_ARG;
   if (out_stride < oa_enc_out_stride_needed(b->Fs, frame_size, max_data_bytes, oa_batch_any_cbr(b))) return OPUS_BUFFER_TOO_SMALL;
   HIPCHECK(hipSetDevice(b->device));
   hipStream_t s = hip_stream ? (hipStream_t)hip_stream : b->stream;
   if (b->kind) {
      /* a launch whose streams are all pinned to the SILK layer (RESTRICTED_SILK, or OPUS_SET_FORCE_MODE(SILK_ONLY) with >= 10 ms frames at <= wideband) never enters the
       * CELT arena and gets the smaller LDS footprint (one more wave per CU) */
      if (subset) { b->all_silk_pinned = 0; b->cfg_dirty = true; }
      else if (b->cfg_dirty) {
         int pinned = 1;
         for (opus_int32 i = 0; pinned && i < b->n_act; i++) {
            const OaShConfig &c = b->h_sh[i].cfg;
            pinned = c.application == OPUS_APPLICATION_RESTRICTED_SILK || (c.user_forced_mode == OPUS_MODE_SILK_ONLY && !c.lfe && (b->Fs <= 16000 || (c.user_bandwidth != OPUS_AUTO && c.user_bandwidth <= OPUS_BANDWIDTH_WIDEBAND) || c.max_bandwidth <= OPUS_BANDWIDTH_WIDEBAND));
         }
         b->all_silk_pinned = pinned; b->cfg_dirty = false;
      }
      const int silk_only = b->all_silk_pinned && frame_size >= b->Fs / 100;
      const size_t lds = sh_lds_bytes(b->channels, silk_only);                                                 /* (without the packet buffer: it goes behind, ShLds.packet_off) */
      const int po = (int)((lds + 15) & ~(size_t)15); const size_t lds_pk = (size_t)po + SH_PKT_BYTES;
      /* 0: one kernel; 1: front / quantiser / back kernels; 2: the same with the one-wave-per-stream reference quantiser; 3 / 4: with the pred stage as kernel(s) of its own; unset: the kernel pipeline (4) when the launch is wide
       * enough for it to pay -- a handful of streams (the classic API's lone caller: profiles/r04_j) finish sooner in one launch than in four */
      static const int split_env = getenv("OPUS_AMD_SH_SPLIT") ? atoi(getenv("OPUS_AMD_SH_SPLIT")) : -1;
      const int split_mode = b->pipeline >= 0 ? b->pipeline : split_env >= 0 ? split_env : (b->n_act >= 64 ? 4 : 0);
      b->pvq4_last = 0;
      if (split_mode && !subset && (frame_size * 100 == b->Fs || frame_size * 50 == b->Fs || (split_mode != 2 && (frame_size * 25 == b->Fs || frame_size * 50 == 3 * b->Fs)))) return oa_sh_encode_split(b, d_pcm, d_apcm, frame_size, d_out, out_stride, max_data_bytes, d_lens, d_final_range, s, lds, silk_only, split_mode, pcm_row);
      int grid = 0;
      { const int r = oa_persistent_grid(b, (const void *)oa_sh_encode_kernel, lds_pk, SH_SCRATCH_BYTES(frame_size, b->channels), s, &grid); if (r != OPUS_OK) return r; }
      hipLaunchKernelGGL(oa_sh_encode_kernel, dim3((unsigned)grid), dim3(64), lds_pk, s,
            b->d_sh, (const i16 *)d_pcm, (const i32 *)d_apcm, frame_size, (int)max_data_bytes, (u8 *)d_out, (int)out_stride, b->d_scratch, (i32 *)d_lens, (u32 *)d_final_range, (int)b->n_act, b->d_queue,
            (const int *)nullptr, (const unsigned *)nullptr, po, pcm_row, first, stride, (const i32 *)d_budget);
      HIPCHECK(hipGetLastError());
      return OPUS_OK;
   }
   static const size_t lds_pad = getenv("OPUS_AMD_LDS_PAD") ? (size_t)atoi(getenv("OPUS_AMD_LDS_PAD")) : 0;   /* occupancy experiments only */
   /* a wide launch of 48 kHz frames up to 20 ms: the transient analysis' recursions first, one lane per (stream, channel) (oa_celt_transient_kernel); OPUS_AMD_TR_PRE=0 keeps them in the encode kernel */
   static const int tr_env = getenv("OPUS_AMD_TR_PRE") ? atoi(getenv("OPUS_AMD_TR_PRE")) : 1;               /* (process default behind OPUS_AMD_SET_TRANSIENT_PREPASS(-1): 0 never, 1 wide launches, 2 always) */
   const i32 *d_tr = nullptr;
   oa_stamp_begin(b, s);
   if ((b->tr_pre >= 0 ? b->tr_pre : tr_env) && b->Fs == 48000 && frame_size <= OA_MAX_FRAME && (b->n_act >= 64 || b->tr_pre == 1 || (b->tr_pre < 0 && tr_env == 2))) {
      const int items = (int)b->n_act * b->channels, tiles = (items + 63) / 64;
      const int g = tiles < 8 * (b->num_cu > 0 ? b->num_cu : 1) ? tiles : 8 * (b->num_cu > 0 ? b->num_cu : 1);
      const size_t need = (size_t)g * (OA_MAX_FRAME + OA_OVERLAP) * 64 * sizeof(i16);
      if (!b->d_tr) { HIPCHECK(hipMalloc((void **)&b->d_tr, (size_t)b->S * 4 * sizeof(i32))); HIPCHECK(hipMemsetAsync(b->d_tr, 0, (size_t)b->S * 4 * sizeof(i32), s)); }
      if (need > b->tr_scratch_cap) { HIPCHECK(hipStreamSynchronize(s)); if (b->d_tr_scratch) (void)hipFree(b->d_tr_scratch); b->d_tr_scratch = nullptr; b->tr_scratch_cap = 0; HIPCHECK(hipMalloc((void **)&b->d_tr_scratch, need)); b->tr_scratch_cap = need; }
      hipLaunchKernelGGL(oa_celt_transient_kernel, dim3((unsigned)g), dim3(64), 0, s, (const OaStream *)b->d_streams, (const i16 *)d_pcm, pcm_row, frame_size, b->channels, first, stride, items, b->d_tr_scratch, b->d_tr);
      oa_stamp(b, s, "oa_celt_transient_kernel");
      d_tr = b->d_tr;
   }
   /* the kernel pipeline of the CELT-only applications (OPUS_AMD_SET_KERNEL_PIPELINE: -1 = wide launches, 0 = never, >= 1 = always; process default OPUS_AMD_CELT_PIPE): 10 / 20 ms
    * calls stop before the PVQ (oa_encode_kernel with continuation records), oa_celt_pvq_kernel codes the bands of four streams per wave, oa_celt_back_kernel finishes the calls */
   static const int pipe_env = getenv("OPUS_AMD_CELT_PIPE") ? atoi(getenv("OPUS_AMD_CELT_PIPE")) : -1;
   const int pipe_mode = b->pipeline >= 0 ? b->pipeline : pipe_env;
   /* -1: the pipeline when the one-kernel path would need more than one round of the chip (its waves = the streams it holds at once): below that a lone frame's serial
    * chain is the call's latency, and one wave per stream walks it faster than the three-kernel relay (profiles/r06_j: 4,096 streams 4.6 vs 5.2 ms, 16,384 13.9 vs 11.0 ms) */
   long long one_round = 0;
   { int g1 = 0; const int r = oa_sh_grid(b, 9, (const void *)oa_encode_kernel, sizeof(FrameLds) + lds_pad, 1LL << 40, &g1); if (r != OPUS_OK) return r; one_round = g1; }
   const bool pipe = (pipe_mode < 0 ? b->n_act > one_round : pipe_mode > 0) && b->pvq_stage != 0 && (frame_size * 100 == b->Fs || frame_size * 50 == b->Fs);
   if (pipe && !b->d_ccont) {
      HIPCHECK(hipMalloc((void **)&b->d_ccont, sizeof(CeltCont) * (size_t)b->S));
      HIPCHECK(hipMalloc((void **)&b->d_cut_list, 2 * sizeof(int) * (size_t)b->S));                 /* [S] the streams that were cut, in the order they were; [S] sorted for the PVQ kernel */
      HIPCHECK(hipMalloc((void **)&b->d_srt, 2 * OA_SORT_KEYS * sizeof(unsigned)));
   }
   b->celt_pipe_last = pipe ? (int)b->n_act : 0; b->pvq4_last = pipe;
   int grid = 0;
   { const int r = oa_persistent_grid(b, pipe ? (const void *)oa_celt_front_kernel : (const void *)oa_encode_kernel, sizeof(FrameLds) + lds_pad, sizeof(CeltScratch), s, &grid); if (r != OPUS_OK) return r; }
   if (pipe) {
      HIPCHECK(hipMemsetAsync(b->d_queue, 0, 4 * sizeof(unsigned), s));
      HIPCHECK(hipMemsetAsync(b->d_srt, 0, 2 * OA_SORT_KEYS * sizeof(unsigned), s));
      hipLaunchKernelGGL(oa_celt_front_kernel, dim3((unsigned)grid), dim3(64), sizeof(FrameLds) + lds_pad, s,
            b->d_streams, (const i16 *)d_pcm, (const i32 *)d_apcm, frame_size, (int)max_data_bytes, (u8 *)d_out, (int)out_stride, (i32 *)d_lens, (u32 *)d_final_range, (int)b->n_act, (CeltScratch *)b->d_scratch, b->d_queue, pcm_row, first, stride, (const i32 *)d_budget, d_tr,
            b->d_ccont, b->d_cut_list, b->d_srt);
   } else hipLaunchKernelGGL(oa_encode_kernel, dim3((unsigned)grid), dim3(64), sizeof(FrameLds) + lds_pad, s,
         b->d_streams, (const i16 *)d_pcm, (const i32 *)d_apcm, frame_size, (int)max_data_bytes, (u8 *)d_out, (int)out_stride, (i32 *)d_lens, (u32 *)d_final_range, (int)b->n_act, (CeltScratch *)b->d_scratch, b->d_queue, pcm_row, first, stride, (const i32 *)d_budget, d_tr,
         (CeltCont *)nullptr, (int *)nullptr, (unsigned *)nullptr);
   oa_stamp(b, s, pipe ? "oa_celt_front_kernel" : "oa_encode_kernel");
   HIPCHECK(hipGetLastError());
   if (pipe) {
      int g_pvq = 0, g_back = 0;
      { const int r = oa_sh_grid(b, 6, (const void *)oa_celt_pvq_kernel, sizeof(P4Lds), ((long long)b->n_act + 3) / 4, &g_pvq); if (r != OPUS_OK) return r; }
      { const int r = oa_sh_grid(b, 7, (const void *)oa_celt_back_kernel, offsetof(FrameLds, BC), b->n_act, &g_back); if (r != OPUS_OK) return r; }
      if (g_back > grid) g_back = grid;                                   /* (the per-wave scratch was sized for the encode kernel's grid) */
      { const int gs_ = (int)((b->n_act + 63) / 64); hipLaunchKernelGGL(oa_celt_sort_kernel, dim3((unsigned)(gs_ < 1024 ? gs_ : 1024)), dim3(64), 0, s, (const CeltCont *)b->d_ccont, (const int *)b->d_cut_list, b->d_cut_list + b->S, (const unsigned *)b->d_queue, b->d_srt); }
      oa_stamp(b, s, "oa_celt_sort_kernel");
      hipLaunchKernelGGL(oa_celt_pvq_kernel, dim3((unsigned)g_pvq), dim3(64), sizeof(P4Lds), s, b->d_ccont, (const int *)(b->d_cut_list + b->S), b->d_queue);
      oa_stamp(b, s, "oa_celt_pvq_kernel");
      hipLaunchKernelGGL(oa_celt_back_kernel, dim3((unsigned)g_back), dim3(64), offsetof(FrameLds, BC), s, b->d_streams, b->d_ccont, (const int *)b->d_cut_list, b->d_queue, frame_size, (u8 *)d_out, (int)out_stride, (i32 *)d_lens, (u32 *)d_final_range,
            (CeltScratch *)b->d_scratch);
      oa_stamp(b, s, "oa_celt_back_kernel");
      HIPCHECK(hipGetLastError());
   }
   return OPUS_OK;
}
int opusgpu_encode_batch_dev(OpusGpuEncBatch *b, const opus_int16 *d_pcm, int frame_size, unsigned char *d_out,
      opus_int32 out_stride, opus_int32 max_data_bytes, opus_int32 *d_lens, opus_uint32 *d_final_range, void *hip_stream)
{
   return opusgpu_encode_batch_dev_sig(b, d_pcm, nullptr, frame_size, d_out, out_stride, max_data_bytes, d_lens, d_final_range, hip_stream);
}
int opusgpu_pack_packets_cap_dev(const unsigned char *d_out, opus_int32 stride, const opus_int32 *d_lens, const long long *d_offsets, unsigned char *d_packed, opus_int32 n, long long capacity, void *hip_stream)
{
   if (!d_out || !d_lens || !d_offsets || !d_packed || n <= 0 || stride <= 0 || capacity < 0) return OPUS_BAD_ARG;
   hipLaunchKernelGGL(oa_pack_kernel, dim3((unsigned)n), dim3(64), 0, (hipStream_t)hip_stream, (const u8 *)d_out, (int)stride, (const i32 *)d_lens, d_offsets, (u8 *)d_packed, (int)n, capacity);
   HIPCHECK(hipGetLastError());
   return OPUS_OK;
}
int opusgpu_pack_packets_dev(const unsigned char *d_out, opus_int32 stride, const opus_int32 *d_lens, const long long *d_offsets, unsigned char *d_packed, opus_int32 n, void *hip_stream)
{
   return opusgpu_pack_packets_cap_dev(d_out, stride, d_lens, d_offsets, d_packed, n, (long long)n * stride, hip_stream);
}
/* state bytes a frame-step reads plus writes (for the roofline's algorithmic traffic): the CELT-only record moves its scalars, the four energy arrays, the overlap and the
 * pitch history; the SILK-capable record moves configuration, scalars and the SILK state of the coded channels, plus -- hybrid -- the CELT state and the delay line */
int opusgpu_enc_moved_state_bytes(int application, int channels, int hybrid)
{
   if (!oa_app_is_sh(application)) return 2 * (int)(sizeof(OaEncScalars) + sizeof(int32_t) * (4 * channels * OA_NB_EBANDS + channels * OA_OVERLAP + channels * OA_MAX_PERIOD));
   int n = (int)(sizeof(OaShConfig) + 2 * sizeof(OaShScalars)) + 2 * 4 * SE_STATE_WORDS(channels);
   if (hybrid) n += 2 * (int)(sizeof(OaEncScalars) + sizeof(int32_t) * (4 * channels * OA_NB_EBANDS + channels * OA_OVERLAP + channels * OA_MAX_PERIOD)) + 2 * 2 * channels * OA_SH_MAX_DELAY;
   return n;
}
/* state bytes the tonality analysis of a 20 ms frame-step reads plus writes: the phase history of the 239 bins (read + written), the 30 ms input window (read; 20 ms of it
 * rewritten), one row of the band-energy rings written and all eight read, the small feature / network state both ways, one info record written and the four
 * fields of the ring that tonality_get_info walks read, the call's info handed to CELT */
int opusgpu_enc_analysis_moved_bytes(void)
{
   return 2 * 3 * 240 * 4 + (AN_BUF_SIZE + 240 + 480) * 4 + (2 * AN_NB_FRAMES * AN_NB_TBANDS + 2 * AN_NB_TBANDS) * 4 + 2 * (3 * AN_NB_TBANDS + 1 + 32 + 8 + 9 + 24 + 16) * 4
        + (int)sizeof(OaAnalysisInfo) * 4 + 4 * AN_DETECT_SIZE * 4;
}
int opusgpu_encode_batch_dev_frames(OpusGpuEncBatch *b, const opus_int16 *d_pcm, int frame_size, int T, unsigned char *d_out, opus_int32 out_stride, opus_int32 max_data_bytes,
      opus_int32 *d_lens, opus_uint32 *d_final_range, void *hip_stream)
{
   if (!b || !d_pcm || !d_out || !d_lens || !d_final_range || T <= 0) return OPUS_BAD_ARG;
   if (b->kind) return OPUS_UNIMPLEMENTED;                                            /* the CELT-only kernel (the headline configuration) */
   { const int fr = oa_enc_frame_size_code(b->Fs, b->application, frame_size); if (fr != OPUS_OK) return fr; }
   if (max_data_bytes <= 0) return OPUS_BAD_ARG;
   if (out_stride < oa_enc_out_stride_needed(b->Fs, frame_size, max_data_bytes)) return OPUS_BUFFER_TOO_SMALL;
   HIPCHECK(hipSetDevice(b->device));
   hipStream_t s = hip_stream ? (hipStream_t)hip_stream : b->stream;
   HIPCHECK(hipFuncSetAttribute((const void *)oa_encode_frames_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
   int grid = 0;
   { const int r = oa_persistent_grid(b, (const void *)oa_encode_frames_kernel, sizeof(FrameLds), sizeof(CeltScratch), s, &grid); if (r != OPUS_OK) return r; }
   hipLaunchKernelGGL(oa_encode_frames_kernel, dim3((unsigned)grid), dim3(64), sizeof(FrameLds), s,
         b->d_streams, (const i16 *)d_pcm, frame_size, T, (int)max_data_bytes, (u8 *)d_out, (int)out_stride, (i32 *)d_lens, (u32 *)d_final_range, (int)b->S, (CeltScratch *)b->d_scratch, b->d_queue);
   HIPCHECK(hipGetLastError());
   return OPUS_OK;
}
int opusgpu_time_encode_dev(OpusGpuEncBatch *b, const opus_int16 *d_pcm, int frame_size, unsigned char *d_out,
      opus_int32 out_stride, opus_int32 max_data_bytes, opus_int32 *d_lens, opus_uint32 *d_final_range, int steps, float *ms)
{
   if (!b || !ms || steps <= 0) return OPUS_BAD_ARG;
   HIPCHECK(hipSetDevice(b->device));
   hipEvent_t e0, e1;
   HIPCHECK(hipEventCreate(&e0)); HIPCHECK(hipEventCreate(&e1));
   HIPCHECK(hipEventRecord(e0, b->stream));
   size_t step_elems = (size_t)b->S * frame_size * b->channels;
   for (int k = 0; k < steps; k++) {
      int r = opusgpu_encode_batch_dev(b, d_pcm + (size_t)k * step_elems, frame_size, d_out, out_stride, max_data_bytes, d_lens, d_final_range, nullptr);
      if (r != OPUS_OK) { (void)hipEventDestroy(e0); (void)hipEventDestroy(e1); return r; }
   }
   HIPCHECK(hipEventRecord(e1, b->stream));
   HIPCHECK(hipEventSynchronize(e1));
   HIPCHECK(hipEventElapsedTime(ms, e0, e1));
   (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
   return OPUS_OK;
}
int opusgpu_encode_batch_sig(OpusGpuEncBatch *b, const opus_int16 *pcm, const opus_int32 *apcm, int frame_size, unsigned char *out,
      opus_int32 out_stride, opus_int32 max_data_bytes, opus_int32 *lens, opus_uint32 *final_range);
int opusgpu_encode_batch_lookahead(OpusGpuEncBatch *b, const opus_int16 *pcm, const opus_int32 *apcm, int frame_size, int analysis_frame_size, unsigned char *out,
      opus_int32 out_stride, opus_int32 max_data_bytes, opus_int32 *lens, opus_uint32 *final_range);
int opusgpu_encode_batch(OpusGpuEncBatch *b, const opus_int16 *pcm, int frame_size, unsigned char *out,
      opus_int32 out_stride, opus_int32 max_data_bytes, opus_int32 *lens, opus_uint32 *final_range)
{
   return opusgpu_encode_batch_sig(b, pcm, nullptr, frame_size, out, out_stride, max_data_bytes, lens, final_range);
}
int opusgpu_encode_batch_sig(OpusGpuEncBatch *b, const opus_int16 *pcm, const opus_int32 *apcm, int frame_size, unsigned char *out,
      opus_int32 out_stride, opus_int32 max_data_bytes, opus_int32 *lens, opus_uint32 *final_range)
{
   return opusgpu_encode_batch_lookahead(b, pcm, apcm, frame_size, frame_size, out, out_stride, max_data_bytes, lens, final_range);
}
int opusgpu_encode_batch_lookahead(OpusGpuEncBatch *b, const opus_int16 *pcm, const opus_int32 *apcm, int frame_size, int analysis_frame_size, unsigned char *out,
      opus_int32 out_stride, opus_int32 max_data_bytes, opus_int32 *lens, opus_uint32 *final_range)
{
   if (!b || !pcm || !out || !lens || analysis_frame_size < frame_size) return OPUS_BAD_ARG;
   { const int fr = oa_enc_frame_size_code(b->Fs, b->application, frame_size); if (fr != OPUS_OK) return fr; }
   HIPCHECK(hipSetDevice(b->device));
   size_t npcm = (size_t)b->n_act * analysis_frame_size * b->channels * sizeof(opus_int16), nout = (size_t)b->n_act * out_stride;
   if (npcm > b->pcm_cap) { if (b->d_pcm) (void)hipFree(b->d_pcm); HIPCHECK(hipMalloc((void **)&b->d_pcm, npcm)); b->pcm_cap = npcm; }
   if (nout > b->out_cap) { if (b->d_out) (void)hipFree(b->d_out); HIPCHECK(hipMalloc((void **)&b->d_out, nout)); b->out_cap = nout; }
   HIPCHECK(hipMemcpyAsync(b->d_pcm, pcm, npcm, hipMemcpyHostToDevice, b->stream));
   if (apcm) {
      if (2 * npcm > b->apcm_cap) { if (b->d_apcm) (void)hipFree(b->d_apcm); b->d_apcm = nullptr; b->apcm_cap = 0; HIPCHECK(hipMalloc((void **)&b->d_apcm, 2 * npcm)); b->apcm_cap = 2 * npcm; }
      HIPCHECK(hipMemcpyAsync(b->d_apcm, apcm, 2 * npcm, hipMemcpyHostToDevice, b->stream));
   }
   int r = opusgpu_encode_batch_lookahead_dev(b, b->d_pcm, apcm ? b->d_apcm : nullptr, frame_size, analysis_frame_size, b->d_out, out_stride, max_data_bytes, b->d_lens, b->d_rng, nullptr);
   if (r != OPUS_OK) return r;
   HIPCHECK(hipMemcpyAsync(out, b->d_out, nout, hipMemcpyDeviceToHost, b->stream));
   HIPCHECK(hipMemcpyAsync(lens, b->d_lens, sizeof(opus_int32) * (size_t)b->n_act, hipMemcpyDeviceToHost, b->stream));
   if (final_range) HIPCHECK(hipMemcpyAsync(final_range, b->d_rng, sizeof(opus_uint32) * (size_t)b->n_act, hipMemcpyDeviceToHost, b->stream));
   HIPCHECK(hipStreamSynchronize(b->stream));
   return OPUS_OK;
}

/* ---------------- classic libopus encoder API on top of a process-wide batch-of-one ---------------- */
#define OA_MAGIC 0x4f41454eu /* "OAEN" */
struct OpusEncoder { uint32_t magic; uint32_t kind; uint32_t pipeline_p2 /* OPUS_AMD_SET_KERNEL_PIPELINE + 2; 0 = never set (-1) */; uint32_t launch_opts /* bits 0-1: OPUS_AMD_SET_TRANSIENT_PREPASS + 2, bits 2-3: OPUS_AMD_SET_PVQ_STAGE + 2; 0 = never set (-1) */; union { OaStream s; OaShStream sh; }; };   /* flat, no device handles: memcpy-able (include/opus.h:108) */
static OpusGpuEncBatch *g_classic[2][5][2];                   /* [record kind][API rate][channels - 1]: created by the first call of that shape, used by one launch at a time */
/* one classic call waiting for (or leading) a launch: opus_call_combiner.h */
struct OaEncCall {
   OpusEncoder *st; const opus_int16 *pcm; const opus_int32 *apcm; unsigned char *data;
   int kind, frame_size, application, channels; opus_int32 Fs, max_data_bytes;
   int ret; bool done; size_t tid; int pipeline; unsigned launch_opts; int analysis_frame_size;    /* samples per channel behind pcm (>= frame_size: OPUS_SET_EXPERT_FRAME_DURATION) */
   const void *who() const { return st; }
   bool same_shape(const OaEncCall &o) const
   {
      return kind == o.kind && Fs == o.Fs && channels == o.channels && application == o.application && frame_size == o.frame_size
          && max_data_bytes == o.max_data_bytes && (apcm != nullptr) == (o.apcm != nullptr) && pipeline == o.pipeline && launch_opts == o.launch_opts && analysis_frame_size == o.analysis_frame_size;
   }
};
static OaCallCombiner<OaEncCall> g_enc_calls;
/* page-locked staging for the state records of a launch's group (one launch at a time per combiner owns it): packed here, sent with one copy, received back, unpacked */
struct OaPinned { char *p = nullptr; size_t cap = 0; };
static OaPinned g_enc_pin, g_dec_pin;
static char *oa_pinned(OaPinned &b, size_t n)
{
   if (n > b.cap) {
      if (b.p) (void)hipHostFree(b.p);
      b.p = nullptr; b.cap = 0;
      if (hipHostMalloc((void **)&b.p, n, 0) != hipSuccess) { b.p = nullptr; return nullptr; }
      b.cap = n;
   }
   return b.p;
}
/* the device the classic entry points (opus_encode, opus_decode, the multistream and projection calls) run on: OPUS_AMD_DEVICE=<index>, default 0.  The batch ABI takes
 * its device as an argument; a process that serves N GPUs through the classic API runs one process per GPU */
static int oa_classic_device()
{
   static const int dev = getenv("OPUS_AMD_DEVICE") && atoi(getenv("OPUS_AMD_DEVICE")) > 0 ? atoi(getenv("OPUS_AMD_DEVICE")) : 0;
   return dev;
}
static int oa_classic_cap()      /* states one launch of the classic API carries at most (the device arrays of a shape are sized for it once) */
{
   static const int cap = getenv("OPUS_AMD_CLASSIC_BATCH") && atoi(getenv("OPUS_AMD_CLASSIC_BATCH")) > 0 ? atoi(getenv("OPUS_AMD_CLASSIC_BATCH")) : 256;
   return cap;
}
static int oa_classic_linger_us() /* how long the leader of a launch waits for the callers it expects (opus_call_combiner.h); 0 = never */
{
   static const int us = getenv("OPUS_AMD_CLASSIC_LINGER_US") ? atoi(getenv("OPUS_AMD_CLASSIC_LINGER_US")) : 200;
   return us;
}
static int oa_fs_index(opus_int32 Fs) { return Fs == 8000 ? 0 : Fs == 12000 ? 1 : Fs == 16000 ? 2 : Fs == 24000 ? 3 : 4; }

int opus_encoder_get_size(int channels) { if (channels < 1 || channels > 2) return 0; return (int)sizeof(OpusEncoder); }
int opus_encoder_init(OpusEncoder *st, opus_int32 Fs, int channels, int application)
{
   if (!st) return OPUS_BAD_ARG;
   if (!oa_fs_ok(Fs) || (channels != 1 && channels != 2) || !oa_app_ok(application)) return OPUS_BAD_ARG;
   memset(st, 0, sizeof(*st));
   st->magic = OA_MAGIC; st->kind = (uint32_t)oa_app_is_sh(application);
   return st->kind ? sh_init_stream(&st->sh, Fs, channels, application) : oa_init_stream(&st->s, Fs, channels, application);
}
OpusEncoder *opus_encoder_create(opus_int32 Fs, int channels, int application, int *error)
{
   if (!oa_fs_ok(Fs) || (channels != 1 && channels != 2) || !oa_app_ok(application)) { if (error) *error = OPUS_BAD_ARG; return nullptr; }
   OpusEncoder *st = (OpusEncoder *)malloc(sizeof(OpusEncoder));
   if (!st) { if (error) *error = OPUS_ALLOC_FAIL; return nullptr; }
   int r = opus_encoder_init(st, Fs, channels, application);
   if (error) *error = r;
   if (r != OPUS_OK) { free(st); return nullptr; }
   return st;
}
void opus_encoder_destroy(OpusEncoder *st) { free(st); }
/* one launch for a group of classic calls of one shape: their records go to the first n slots of the shape's batch, the kernel runs n waves, the records come back */
static int oa_classic_encode_group_run(std::vector<OaEncCall *> &g)
{
   const OaEncCall &h = *g[0];
   const int n = (int)g.size(), kind = h.kind;
   OpusGpuEncBatch **slot = &g_classic[kind][oa_fs_index(h.Fs)][h.channels - 1];
   if (!*slot) {
      int err;
      *slot = opusgpu_enc_batch_create(oa_classic_cap(), h.Fs, h.channels, kind ? OPUS_APPLICATION_AUDIO : OPUS_APPLICATION_RESTRICTED_LOWDELAY, oa_classic_device(), &err);
      if (!*slot) return err == OPUS_OK ? OPUS_INTERNAL_ERROR : err;
   }
   OpusGpuEncBatch *b = *slot;
   b->application = h.application; b->n_act = n; b->pipeline = h.pipeline; b->tr_pre = (h.launch_opts & 3) ? (int)(h.launch_opts & 3) - 2 : -1; b->pvq_stage = ((h.launch_opts >> 2) & 3) ? (int)((h.launch_opts >> 2) & 3) - 2 : -1;
   opus_int32 stride = oa_enc_out_stride_needed(h.Fs, h.frame_size, h.max_data_bytes) + 8;
   if (stride < 1288) stride = 1288;
   const size_t per = (size_t)h.analysis_frame_size * h.channels, rec = kind ? sizeof(OaShStream) : sizeof(OaStream);
   HIPCHECK(hipSetDevice(b->device));
   HIPCHECK(hipStreamSynchronize(b->stream));
   const opus_int16 *pcm = h.pcm; const opus_int32 *apcm = h.apcm;
   std::vector<opus_int16> pcm_all; std::vector<opus_int32> apcm_all; char *recs = nullptr;
   if (n > 1) {                                                            /* inputs and records of the group side by side: one contiguous transfer each */
      pcm_all.resize(per * n);
      recs = oa_pinned(g_enc_pin, rec * (size_t)oa_classic_cap());
      if (!recs) return OPUS_ALLOC_FAIL;
      if (apcm) apcm_all.resize(per * n);
      for (int i = 0; i < n; i++) {
         memcpy(pcm_all.data() + per * i, g[i]->pcm, per * sizeof(opus_int16));
         if (apcm) memcpy(apcm_all.data() + per * i, g[i]->apcm, per * sizeof(opus_int32));
         memcpy(recs + rec * i, kind ? (const void *)&g[i]->st->sh : (const void *)&g[i]->st->s, rec);
      }
      pcm = pcm_all.data(); if (apcm) apcm = apcm_all.data();
   }
   const void *rec_src = n > 1 ? (const void *)recs : kind ? (const void *)&h.st->sh : (const void *)&h.st->s;
   HIPCHECK(hipMemcpy(kind ? (void *)b->d_sh : (void *)b->d_streams, rec_src, rec * (size_t)n, hipMemcpyHostToDevice));
   b->any_cbr = -1; b->any_fec = -1;
   if (kind) { for (int i = 0; i < n; i++) b->h_sh[i].cfg = g[i]->st->sh.cfg; b->cfg_dirty = true; }     /* the launch's host-side decisions follow the records it carries */
   else for (int i = 0; i < n; i++) b->h_streams[i].cfg = g[i]->st->s.cfg;
   std::vector<unsigned char> out((size_t)stride * n);
   std::vector<opus_int32> lens((size_t)n); std::vector<opus_uint32> rng((size_t)n);
   int r = opusgpu_encode_batch_lookahead(b, pcm, apcm, h.frame_size, h.analysis_frame_size, out.data(), stride, h.max_data_bytes, lens.data(), rng.data());
   if (r != OPUS_OK) return r;
   if (n > 1) {
      HIPCHECK(hipMemcpy(recs, kind ? (const void *)b->d_sh : (const void *)b->d_streams, rec * (size_t)n, hipMemcpyDeviceToHost));
      for (int i = 0; i < n; i++) memcpy(kind ? (void *)&g[i]->st->sh : (void *)&g[i]->st->s, recs + rec * i, rec);
   } else HIPCHECK(hipMemcpy(kind ? (void *)&h.st->sh : (void *)&h.st->s, kind ? (const void *)b->d_sh : (const void *)b->d_streams, rec, hipMemcpyDeviceToHost));
   for (int i = 0; i < n; i++) {
      const opus_int32 len = lens[i];
      if (len > 0) memcpy(g[i]->data, out.data() + (size_t)stride * i, (size_t)(len < h.max_data_bytes ? len : h.max_data_bytes));
      g[i]->ret = len;
   }
   return OPUS_OK;
}
static void oa_classic_encode_group(std::vector<OaEncCall *> &g)
{
   const int r = oa_classic_encode_group_run(g);
   if (r != OPUS_OK) for (OaEncCall *c : g) c->ret = r;
}
/* one frame of one classic encoder: the record goes to the device, the kernel runs one wave for it (and one for every other call waiting with it), the record comes back.  depth = the sample depth
 * of the entry point (the lsb_depth argument of opus_encode_native, src/opus_encoder.c:2667,:2722: 16 for every entry point of this build, see OA_MAX_ENCODING_DEPTH) */
/* MAX_ENCODING_DEPTH of the build this library reproduces (FIXED_POINT without ENABLE_RES24, celt/arch.h:176): the 24-bit and float entry points convert to 16-bit samples and
 * say so -- lsb_depth 16 reaches the codec and the analysis, whatever the caller's samples held (src/opus_encoder.c:2724,:2762; the analysis still sees the unrounded samples) */
#define OA_MAX_ENCODING_DEPTH 16
static opus_int32 oa_classic_encode(OpusEncoder *st, const opus_int16 *pcm, int analysis_frame_size, unsigned char *data, opus_int32 max_data_bytes, int depth, const opus_int32 *apcm = nullptr)
{
   if (!st || st->magic != OA_MAGIC || !pcm || !data) return OPUS_BAD_ARG;
   const int channels = st->kind ? st->sh.cfg.channels : st->s.cfg.channels, application = st->kind ? st->sh.cfg.application : st->s.cfg.application;
   const opus_int32 Fs = st->kind ? st->sh.cfg.Fs : st->s.Fs;
   const int frame_size = (int)oa_frame_size_select(application, analysis_frame_size, st->kind ? st->sh.cfg.variable_duration : st->s.cfg.variable_duration, Fs);
   /* opus_encode_native clears the final range before it looks at its arguments (src/opus_encoder.c:1223-1228); the int16 entry point hands it frame_size_select's -1 as it
    * comes (:2659-2667), the 24-bit and float ones turn an illegal frame size away before that (:2713-2718, :2752-2757) */
   const int fr_code = frame_size <= 0 ? OPUS_BAD_ARG : oa_enc_frame_size_code(Fs, application, frame_size);
   if (fr_code != OPUS_OK || max_data_bytes <= 0) {
      if (fr_code == OPUS_OK || apcm == nullptr) { if (st->kind) st->sh.s.rangeFinal = 0; else st->s.st.s.rangeFinal = 0; }
      return fr_code != OPUS_OK ? fr_code : OPUS_BAD_ARG;
   }
   if (st->kind) st->sh.cfg.input_depth = depth; else st->s.cfg.input_depth = depth;
   OaEncCall call = {st, pcm, apcm, data, (int)st->kind, frame_size, application, channels, Fs, max_data_bytes, OPUS_INTERNAL_ERROR, false, 0, st->pipeline_p2 ? (int)st->pipeline_p2 - 2 : -1, (unsigned)st->launch_opts, analysis_frame_size > frame_size ? analysis_frame_size : frame_size};
   g_enc_calls.submit(&call, oa_classic_cap(), oa_classic_linger_us(), oa_classic_encode_group);
   return call.ret;
}
opus_int32 opus_encode(OpusEncoder *st, const opus_int16 *pcm, int frame_size, unsigned char *data, opus_int32 max_data_bytes)
{
   return oa_classic_encode(st, pcm, frame_size, data, max_data_bytes, 16);
}
/* int16-resolution build of the reference (FIXED_POINT without ENABLE_RES24): the 24-bit and float inputs are rounded to int16 first
 * (INT24TORES = SAT16(PSHR32(a, 8)), FLOAT2RES = FLOAT2INT16; celt/arch.h:171-173, src/opus_encoder.c:2703-2722, :2745-2765) */
static inline opus_int16 oa_sat16(opus_int32 x) { return (opus_int16)(x > 32767 ? 32767 : x < -32768 ? -32768 : x); }
static inline opus_int16 oa_float2int16(float x)
{
   x = x * 32768.f;
   x = x > -32768.f ? x : -32768.f;
   x = x < 32767.f ? x : 32767.f;
   return (opus_int16)lrintf(x);
}
static inline opus_int32 oa_float2sig(float x)
{
   x = x * ((opus_int32)32768 << 12);
   x = x > -(65536 << 12) ? x : -(65536 << 12);
   x = x < (65536 << 12) ? x : (65536 << 12);
   return (opus_int32)lrintf(x);
}
opus_int32 opus_encode24(OpusEncoder *st, const opus_int32 *pcm, int frame_size, unsigned char *data, opus_int32 max_data_bytes)
{
   if (!st || st->magic != OA_MAGIC || !pcm || frame_size <= 0 || frame_size > 5760 * 2) return OPUS_BAD_ARG;
   const int channels = st->kind ? st->sh.cfg.channels : st->s.cfg.channels;
   std::vector<opus_int16> in((size_t)frame_size * channels);
   std::vector<opus_int32> sig((size_t)frame_size * channels);                                 /* what the analysis sees: INT24TOSIG (downmix_int24, src/opus_encoder.c:804) */
   for (size_t i = 0; i < in.size(); i++) { in[i] = oa_sat16((pcm[i] + 128) >> 8); sig[i] = (opus_int32)((opus_uint32)pcm[i] << 4); }
   return oa_classic_encode(st, in.data(), frame_size, data, max_data_bytes, OA_MAX_ENCODING_DEPTH, sig.data());
}
opus_int32 opus_encode_float(OpusEncoder *st, const float *pcm, int frame_size, unsigned char *data, opus_int32 max_data_bytes)
{
   if (!st || st->magic != OA_MAGIC || !pcm || frame_size <= 0 || frame_size > 5760 * 2) return OPUS_BAD_ARG;
   const int channels = st->kind ? st->sh.cfg.channels : st->s.cfg.channels;
   std::vector<opus_int16> in((size_t)frame_size * channels);
   std::vector<opus_int32> sig((size_t)frame_size * channels);                                 /* what the analysis sees: FLOAT2SIG (downmix_float :748, celt/float_cast.h:166) */
   for (size_t i = 0; i < in.size(); i++) { in[i] = oa_float2int16(pcm[i]); sig[i] = oa_float2sig(pcm[i]); }
   return oa_classic_encode(st, in.data(), frame_size, data, max_data_bytes, OA_MAX_ENCODING_DEPTH, sig.data());
}
int opus_encoder_ctl(OpusEncoder *st, int request, ...)
{
   if (!st || st->magic != OA_MAGIC) return OPUS_BAD_ARG;
   va_list ap;
   va_start(ap, request);
   int ret;
   if (request == OPUS_RESET_STATE) ret = st->kind ? sh_ctl_set(&st->sh, request, 0) : oa_ctl_set(&st->s, request, 0);
   else if (request == OPUS_AMD_SET_KERNEL_PIPELINE_REQUEST) { const opus_int32 v = va_arg(ap, opus_int32); if (v < -1 || v > 4) ret = OPUS_BAD_ARG; else { st->pipeline_p2 = (uint32_t)(v + 2); ret = OPUS_OK; } }
   else if (request == OPUS_AMD_GET_KERNEL_PIPELINE_REQUEST) { opus_int32 *p = va_arg(ap, opus_int32 *); if (!p) ret = OPUS_BAD_ARG; else { *p = st->pipeline_p2 ? (opus_int32)st->pipeline_p2 - 2 : -1; ret = OPUS_OK; } }
   else if (request == OPUS_AMD_SET_TRANSIENT_PREPASS_REQUEST || request == OPUS_AMD_SET_PVQ_STAGE_REQUEST) {
      const opus_int32 v = va_arg(ap, opus_int32); const int sh_ = request == OPUS_AMD_SET_PVQ_STAGE_REQUEST ? 2 : 0;
      if (v < -1 || v > 1) ret = OPUS_BAD_ARG; else { st->launch_opts = (st->launch_opts & ~(3u << sh_)) | ((uint32_t)(v + 2) << sh_); ret = OPUS_OK; }
   }
   else if (request == OPUS_AMD_GET_TRANSIENT_PREPASS_REQUEST || request == OPUS_AMD_GET_PVQ_STAGE_REQUEST) {
      opus_int32 *p = va_arg(ap, opus_int32 *); const int sh_ = request == OPUS_AMD_GET_PVQ_STAGE_REQUEST ? 2 : 0;
      if (!p) ret = OPUS_BAD_ARG; else { const uint32_t f = (st->launch_opts >> sh_) & 3; *p = f ? (opus_int32)f - 2 : -1; ret = OPUS_OK; }
   }
   else if (request == OPUS_SET_ENERGY_MASK_REQUEST) {                                  /* internal (src/opus_private.h): multistream surround masking, 21 values per channel or NULL */
      const opus_int32 *m = va_arg(ap, const opus_int32 *);
      opus_int32 *dst = st->kind ? st->sh.energy_mask : st->s.energy_mask;
      const int n = 21 * (st->kind ? st->sh.cfg.channels : st->s.cfg.channels);
      if (m) memcpy(dst, m, sizeof(opus_int32) * (size_t)n);
      if (st->kind) { st->sh.cfg.energy_mask_on = m != nullptr; st->sh.s.celt_mask_cleared = 0; } else st->s.energy_mask_on = m != nullptr;
      ret = OPUS_OK;
   }
   else if (request == OPUS_SET_APPLICATION_REQUEST) {
      const opus_int32 v = va_arg(ap, opus_int32);
      ret = st->kind ? sh_ctl_set(&st->sh, request, v) : oa_ctl_set(&st->s, request, v);
      if (ret == OPUS_UNIMPLEMENTED) {                                                  /* legal before the first frame, but the new application lives in the other record type: convert */
         OpusEncoder *o = (OpusEncoder *)malloc(sizeof(OpusEncoder));
         if (!o) ret = OPUS_ALLOC_FAIL;
         else {
            memcpy(o, st, sizeof(*o));
            ret = opus_encoder_init(st, o->kind ? o->sh.cfg.Fs : o->s.Fs, o->kind ? o->sh.cfg.channels : o->s.cfg.channels, v);
            st->pipeline_p2 = o->pipeline_p2; st->launch_opts = o->launch_opts;
            static const int carry[] = {OPUS_SET_BITRATE_REQUEST, OPUS_SET_COMPLEXITY_REQUEST, OPUS_SET_VBR_REQUEST, OPUS_SET_VBR_CONSTRAINT_REQUEST, OPUS_SET_FORCE_CHANNELS_REQUEST,
               OPUS_SET_BANDWIDTH_REQUEST, OPUS_SET_MAX_BANDWIDTH_REQUEST, OPUS_SET_LSB_DEPTH_REQUEST, OPUS_SET_PHASE_INVERSION_DISABLED_REQUEST, OPUS_SET_FORCE_MODE_REQUEST, OPUS_SET_SIGNAL_REQUEST,
               OPUS_SET_PACKET_LOSS_PERC_REQUEST, OPUS_SET_INBAND_FEC_REQUEST, OPUS_SET_DTX_REQUEST, OPUS_SET_VOICE_RATIO_REQUEST, OPUS_SET_EXPERT_FRAME_DURATION_REQUEST, OPUS_SET_PREDICTION_DISABLED_REQUEST};
            for (size_t i = 0; ret == OPUS_OK && i < sizeof(carry) / sizeof(carry[0]); i++) {
               opus_int32 cur = 0;
               if (carry[i] == OPUS_SET_BITRATE_REQUEST) cur = o->kind ? o->sh.cfg.user_bitrate_bps : o->s.cfg.user_bitrate_bps;
               else if (carry[i] == OPUS_SET_BANDWIDTH_REQUEST) cur = o->kind ? o->sh.cfg.user_bandwidth : o->s.cfg.user_bandwidth;
               else if (carry[i] == OPUS_SET_FORCE_MODE_REQUEST) cur = o->kind ? o->sh.cfg.user_forced_mode : o->s.user_forced_mode;
               else if ((o->kind ? sh_ctl_get(&o->sh, carry[i] + 1, &cur) : oa_ctl_get(&o->s, carry[i] + 1, &cur)) != OPUS_OK) continue;
               (void)(st->kind ? sh_ctl_set(&st->sh, carry[i], cur) : oa_ctl_set(&st->s, carry[i], cur));
            }
            free(o);
         }
      }
   }
   else if (request == 10015 /* CELT_GET_MODE (celt/celt.h; the reference's multistream code asks its elementary encoders for it): an opaque handle here */) {
      const void **p = va_arg(ap, const void **);
      static const int oa_mode48000_960 = 0;
      if (!p) ret = OPUS_BAD_ARG; else { *p = &oa_mode48000_960; ret = OPUS_OK; }
   }
   else if (request & 1) { opus_int32 *p = va_arg(ap, opus_int32 *); ret = st->kind ? sh_ctl_get(&st->sh, request, p) : oa_ctl_get(&st->s, request, p); }   /* GET requests are odd */
   else { opus_int32 v = va_arg(ap, opus_int32); ret = st->kind ? sh_ctl_set(&st->sh, request, v) : oa_ctl_set(&st->s, request, v); }
   va_end(ap);
   return ret;
}
const char *opus_strerror(int error)
{
   static const char *const s[8] = {"success", "invalid argument", "buffer too small", "internal error", "corrupted stream", "request not implemented", "invalid state", "memory allocation failed"};
   if (error > 0 || error < -7) return "unknown error";
   return s[-error];
}
#ifdef OA_PHASE_TIMERS
OPUS_AMD_EXPORT int opusgpu_debug_sh_phase_ticks(unsigned long long *out, int reset)
{
   HIPCHECK(hipDeviceSynchronize());
   HIPCHECK(hipMemcpyFromSymbol(out, HIP_SYMBOL(oa_sh_phase_ticks), sizeof(unsigned long long) * 24));
   if (reset) { unsigned long long z[24] = {0}; HIPCHECK(hipMemcpyToSymbol(HIP_SYMBOL(oa_sh_phase_ticks), z, sizeof(z))); }
   return OPUS_OK;
}
OPUS_AMD_EXPORT int opusgpu_debug_phase_ticks(unsigned long long *out, int reset)
{
   HIPCHECK(hipDeviceSynchronize());
   HIPCHECK(hipMemcpyFromSymbol(out, HIP_SYMBOL(oa_phase_ticks), sizeof(unsigned long long) * 50));
   if (reset) { unsigned long long z[50] = {0}; HIPCHECK(hipMemcpyToSymbol(HIP_SYMBOL(oa_phase_ticks), z, sizeof(z))); }
   return OPUS_OK;
}
OPUS_AMD_EXPORT int opusgpu_debug_p4_ticks(unsigned long long *ticks, unsigned long long *lanes, int reset)
{
   HIPCHECK(hipDeviceSynchronize());
   HIPCHECK(hipMemcpyFromSymbol(ticks, HIP_SYMBOL(oa_p4_ticks), sizeof(unsigned long long) * 32));
   HIPCHECK(hipMemcpyFromSymbol(lanes, HIP_SYMBOL(oa_p4_lanes), sizeof(unsigned long long) * 32));
   if (reset) { unsigned long long z[32] = {0}; HIPCHECK(hipMemcpyToSymbol(HIP_SYMBOL(oa_p4_ticks), z, sizeof(z))); HIPCHECK(hipMemcpyToSymbol(HIP_SYMBOL(oa_p4_lanes), z, sizeof(z))); }
   return OPUS_OK;
}
#endif

/* ================= decoder (CELT-only packets at 48 kHz) ================= */
static int oa_dec_init_stream(OaDecStream *st, opus_int32 Fs, int channels)
{
   if ((Fs != 48000 && Fs != 24000 && Fs != 16000 && Fs != 12000 && Fs != 8000) || (channels != 1 && channels != 2)) return OPUS_BAD_ARG;
   oa_dec_stream_reset(st, channels);
   st->s.Fs = Fs; st->s.frame_size = Fs / 400;
   return OPUS_OK;
}
struct OpusGpuDecBatch {
   int device; opus_int32 S; opus_int32 n_act /* as in the encoder batch */; int channels; opus_int32 Fs; int decode_fec; hipStream_t stream;
   OaDecStream *d_streams;
   unsigned char *d_pkt; size_t pkt_cap; opus_int16 *d_pcm; size_t pcm_cap; opus_int32 *d_lens, *d_ns; opus_uint32 *d_rng;
   char *d_scratch; size_t scratch_cap;     /* per resident wave: the spectrum of the frame in flight (OA_DEC_SCRATCH_BYTES) */
   unsigned *d_queue; int *d_slow;          /* d_queue [0] fast kernel's queue, [1] general kernel's queue; [2] / [3] the lengths of the two lists; d_slow [2][S]: the streams oa_decode_look_kernel sent to the fast kernel, then those it left to the general one */
   int num_cu, occ_fast, occ_gen;
   char *d_lane_work; size_t lane_work_cap; OaHybCont *d_hyb_ec;   /* [S] the range decoders of the hybrid packets between the lane kernel and oa_decode_hyb_kernel */
    /* oa_sdec_lane_kernel's work rows, SL_WORK_BYTES per block of its grid */
   int no_lane;                             /* opusgpu_dec_batch_set_lane_kernel(b, 0): SILK-only packets go to the general kernel too */
   int no_fast;                             /* opusgpu_dec_batch_set_fast_kernel(b, 0): every packet goes to the general kernel */
   int pvq_stage;                           /* opusgpu_dec_batch_set_pvq_stage: -1 wide launches (the default), 0 never, 1 always -- the bands of the steady-state CELT frames by oa_celt_dpvq_kernel, four streams per wave */
   CeltDecCont *d_dcont; int *d_cut;        /* [S] continuation records of the decoder's kernel pipeline; [2][S] the streams the front kernels stopped, by frame size (20 ms, 10 ms) */
   int occ_dpvq;
};
int opusgpu_dec_state_size(void) { return (int)sizeof(OaDecStream); }
/* decode_fec of the following calls (opus_decode's last argument, include/opus.h:516): 1 = decode the in-band FEC (LBRR) copy the packets carry for
 * the frame BEFORE them, concealing where there is none */
int opusgpu_dec_batch_set_fec(OpusGpuDecBatch *b, int decode_fec) { if (!b || decode_fec < 0 || decode_fec > 1) return OPUS_BAD_ARG; b->decode_fec = decode_fec; return OPUS_OK; }
/* 0: the following calls skip the CELT-only fast kernel and its look at every stream (a batch that knows it carries no CELT-only packets -- a SILK-only or hybrid service -- saves
 * one launch and one queue pop per stream; any other batch only loses the fast kernel's occupancy); 1 (the default): fast kernel first, the general kernel takes the rest.  The output is the same either way. */
int opusgpu_dec_batch_set_fast_kernel(OpusGpuDecBatch *b, int enable) { if (!b || enable < 0 || enable > 1) return OPUS_BAD_ARG; b->no_fast = !enable; return OPUS_OK; }
/* the last call's split (after a sync): packets the look gave to oa_sdec_lane_kernel, and how many of those it handed on to the general kernel (a redundant CELT frame) */
int opusgpu_dec_batch_lane_stats(OpusGpuDecBatch *b, opus_uint32 *taken, opus_uint32 *handed_on)
{
   if (!b || !taken || !handed_on) return OPUS_BAD_ARG;
   unsigned q[2] = { 0, 0 };
   HIPCHECK(hipSetDevice(b->device)); HIPCHECK(hipStreamSynchronize(b->stream));
   HIPCHECK(hipMemcpy(q, b->d_queue + 4, sizeof(q), hipMemcpyDeviceToHost));
   *taken = q[0]; *handed_on = q[1];
   return OPUS_OK;
}
int opusgpu_dec_batch_set_lane_kernel(OpusGpuDecBatch *b, int enable) { if (!b || enable < 0 || enable > 1) return OPUS_BAD_ARG; b->no_lane = !enable; return OPUS_OK; }
/* the band decoding (quant_all_bands) of the CELT-only and hybrid steady-state packets of one 10 / 20 ms frame as a kernel of its own with four streams per wave
 * (oa_celt_dpvq_kernel, celt_dec_pvq4.h) between the kernels that decode the rest of those frames: -1 (the default) when the call is wider than one round of the fast kernel's
 * waves, 0 never, 1 always.  The output is the same either way.  The last call's count: opusgpu_dec_batch_pvq_stats. */
int opusgpu_dec_batch_set_pvq_stage(OpusGpuDecBatch *b, int mode) { if (!b || mode < -1 || mode > 1) return OPUS_BAD_ARG; b->pvq_stage = mode; return OPUS_OK; }
int opusgpu_dec_batch_pvq_stats(OpusGpuDecBatch *b, opus_uint32 *frames)
{
   if (!b || !frames) return OPUS_BAD_ARG;
   unsigned q[2] = { 0, 0 };
   HIPCHECK(hipSetDevice(b->device)); HIPCHECK(hipStreamSynchronize(b->stream));
   HIPCHECK(hipMemcpy(q, b->d_queue + 8, sizeof(q), hipMemcpyDeviceToHost));
   *frames = q[0] + q[1];
   return OPUS_OK;
}
int opusgpu_dec_kernel_lds_bytes(void) { return (int)sizeof(DecLds); }
int opusgpu_dec_fast_kernel_lds_bytes(void) { return (int)OA_DEC_FAST_LDS_BYTES; }
opus_int32 opusgpu_dec_batch_streams(const OpusGpuDecBatch *b) { return b ? b->S : 0; }
void opusgpu_dec_batch_destroy(OpusGpuDecBatch *b)
{
   if (!b) return;
   (void)hipSetDevice(b->device);
   if (b->stream) (void)hipStreamSynchronize(b->stream);
   if (b->d_streams) (void)hipFree(b->d_streams);
   if (b->d_pkt) (void)hipFree(b->d_pkt);
   if (b->d_pcm) (void)hipFree(b->d_pcm);
   if (b->d_lens) (void)hipFree(b->d_lens);
   if (b->d_ns) (void)hipFree(b->d_ns);
   if (b->d_rng) (void)hipFree(b->d_rng);
   if (b->d_scratch) (void)hipFree(b->d_scratch);
   if (b->d_queue) (void)hipFree(b->d_queue);
   if (b->d_slow) (void)hipFree(b->d_slow);
   if (b->d_lane_work) (void)hipFree(b->d_lane_work);
   if (b->d_hyb_ec) (void)hipFree(b->d_hyb_ec);
   if (b->d_dcont) (void)hipFree(b->d_dcont);
   if (b->d_cut) (void)hipFree(b->d_cut);
   if (b->stream) (void)hipStreamDestroy(b->stream);
   delete b;
}
OpusGpuDecBatch *opusgpu_dec_batch_create(opus_int32 nstreams, opus_int32 Fs, int channels, int device, int *error)
{
   int err = OPUS_OK;
   OpusGpuDecBatch *b = nullptr;
   OaDecStream *proto = new OaDecStream;
   if (nstreams <= 0) err = OPUS_BAD_ARG;
   if (err == OPUS_OK) err = oa_dec_init_stream(proto, Fs, channels);
   if (err == OPUS_OK) {
      int ndev = 0;
      if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0 || device < 0 || device >= ndev) {
         fprintf(stderr, "opus_amd: no usable HIP device (requested %d of %d) — this library has no CPU fallback\n", device, ndev);
         err = OPUS_INTERNAL_ERROR;
      }
   }
   if (err == OPUS_OK) {
      b = new OpusGpuDecBatch();
      b->device = device; b->S = nstreams; b->n_act = nstreams; b->channels = channels; b->Fs = Fs; b->decode_fec = 0; b->stream = nullptr; b->d_streams = nullptr;
      b->d_pkt = nullptr; b->pkt_cap = 0; b->d_pcm = nullptr; b->pcm_cap = 0; b->d_lens = nullptr; b->d_ns = nullptr; b->d_rng = nullptr;
      b->d_scratch = nullptr; b->scratch_cap = 0; b->d_queue = nullptr; b->d_slow = nullptr; b->num_cu = 0; b->occ_fast = 0; b->occ_gen = 0; b->no_fast = 0; b->d_lane_work = nullptr; b->lane_work_cap = 0; b->no_lane = 0; b->d_hyb_ec = nullptr; b->pvq_stage = -1; b->d_dcont = nullptr; b->d_cut = nullptr; b->occ_dpvq = 0;
      std::vector<OaDecStream> init((size_t)(nstreams < 256 ? nstreams : 256), *proto);
      bool ok = hipSetDevice(device) == hipSuccess && hipStreamCreate(&b->stream) == hipSuccess &&
                hipMalloc((void **)&b->d_streams, sizeof(OaDecStream) * (size_t)nstreams) == hipSuccess &&
                hipMalloc((void **)&b->d_lens, sizeof(opus_int32) * (size_t)nstreams) == hipSuccess &&
                hipMalloc((void **)&b->d_ns, sizeof(opus_int32) * (size_t)nstreams) == hipSuccess &&
                hipMalloc((void **)&b->d_rng, sizeof(opus_uint32) * (size_t)nstreams) == hipSuccess &&
                hipMalloc((void **)&b->d_queue, 64) == hipSuccess && hipMalloc((void **)&b->d_slow, 4 * sizeof(int) * (size_t)nstreams) == hipSuccess && hipMalloc((void **)&b->d_hyb_ec, sizeof(OaHybCont) * (size_t)nstreams) == hipSuccess &&
                hipDeviceGetAttribute(&b->num_cu, hipDeviceAttributeMultiprocessorCount, device) == hipSuccess &&
                hipFuncSetAttribute((const void *)oa_decode_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) == hipSuccess &&
                hipFuncSetAttribute((const void *)oa_decode_fast_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, OA_DEC_FAST_DYN_LDS_MAX) == hipSuccess &&
                hipFuncSetAttribute((const void *)oa_decode_hyb_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, OA_DEC_FAST_DYN_LDS_MAX) == hipSuccess &&
                hipOccupancyMaxActiveBlocksPerMultiprocessor(&b->occ_gen, (const void *)oa_decode_kernel, 64, sizeof(DecLds)) == hipSuccess &&
                hipOccupancyMaxActiveBlocksPerMultiprocessor(&b->occ_fast, (const void *)oa_decode_fast_kernel, 64, OA_DEC_FAST_LDS_BYTES) == hipSuccess &&
                hipOccupancyMaxActiveBlocksPerMultiprocessor(&b->occ_dpvq, (const void *)oa_celt_dpvq_kernel, 64, sizeof(P4Lds)) == hipSuccess;
      for (opus_int32 s0 = 0; ok && s0 < nstreams; s0 += 256) {
         opus_int32 n = nstreams - s0 < 256 ? nstreams - s0 : 256;
         ok = hipMemcpy(b->d_streams + s0, init.data(), sizeof(OaDecStream) * (size_t)n, hipMemcpyHostToDevice) == hipSuccess;
      }
      if (!ok) { opusgpu_dec_batch_destroy(b); b = nullptr; err = OPUS_ALLOC_FAIL; }
   }
   delete proto;
   if (error) *error = err;
   return b;
}
int opusgpu_dec_batch_sync(OpusGpuDecBatch *b) { if (!b) return OPUS_BAD_ARG; HIPCHECK(hipSetDevice(b->device)); HIPCHECK(hipStreamSynchronize(b->stream)); return OPUS_OK; }
int opusgpu_dec_batch_reset(OpusGpuDecBatch *b)
{
   if (!b) return OPUS_BAD_ARG;
   OaDecStream *proto = new OaDecStream;
   oa_dec_init_stream(proto, b->Fs, b->channels);
   HIPCHECK(hipSetDevice(b->device));
   HIPCHECK(hipStreamSynchronize(b->stream));
   std::vector<OaDecStream> init((size_t)(b->S < 256 ? b->S : 256), *proto);
   delete proto;
   for (opus_int32 s0 = 0; s0 < b->S; s0 += 256) {
      opus_int32 n = b->S - s0 < 256 ? b->S - s0 : 256;
      HIPCHECK(hipMemcpy(b->d_streams + s0, init.data(), sizeof(OaDecStream) * (size_t)n, hipMemcpyHostToDevice));
   }
   return OPUS_OK;
}
int opusgpu_dec_batch_export_state(OpusGpuDecBatch *b, opus_int32 stream, void *blob)
{
   if (!b || !blob || stream < 0 || stream >= b->S) return OPUS_BAD_ARG;
   HIPCHECK(hipSetDevice(b->device)); HIPCHECK(hipStreamSynchronize(b->stream));
   HIPCHECK(hipMemcpy(blob, b->d_streams + stream, sizeof(OaDecStream), hipMemcpyDeviceToHost));
   return OPUS_OK;
}
int opusgpu_dec_batch_import_state(OpusGpuDecBatch *b, opus_int32 stream, const void *blob)
{
   if (!b || !blob || stream < 0 || stream >= b->S) return OPUS_BAD_ARG;
   if (((const OaDecStream *)blob)->s.channels != b->channels) return OPUS_BAD_ARG;
   HIPCHECK(hipSetDevice(b->device)); HIPCHECK(hipStreamSynchronize(b->stream));
   HIPCHECK(hipMemcpy(b->d_streams + stream, blob, sizeof(OaDecStream), hipMemcpyHostToDevice));
   return OPUS_OK;
}
int opusgpu_decode_batch_dev(OpusGpuDecBatch *b, const unsigned char *d_packets, opus_int32 packet_stride, const opus_int32 *d_lens, opus_int16 *d_pcm,
      int frame_size, opus_int32 *d_nsamples, opus_uint32 *d_final_range, void *hip_stream)
{
   if (!b || !d_packets || !d_lens || !d_pcm || !d_nsamples || !d_final_range || packet_stride <= 0) return OPUS_BAD_ARG;
   if (frame_size <= 0 || frame_size > 5760) return OPUS_BAD_ARG;
   HIPCHECK(hipSetDevice(b->device));
   hipStream_t s = hip_stream ? (hipStream_t)hip_stream : b->stream;
   /* persistent launches: as many waves as the chip holds of each kernel (never more than there are streams), every wave with its own spectrum scratch; the fast kernel
    * first, then the general one over the streams it handed over (its waves find an empty list when there are none) */
   static const int fast_env = getenv("OPUS_AMD_DEC_FAST") ? atoi(getenv("OPUS_AMD_DEC_FAST")) : 1;                 /* 0: the general kernel for every packet (A/B, tests) */
   static const int lane_env = getenv("OPUS_AMD_DEC_LANE") ? atoi(getenv("OPUS_AMD_DEC_LANE")) : 1;                 /* 0: no lane = stream SILK kernel */
   const long long cu = b->num_cu > 0 ? b->num_cu : 1;
   long long g_fast = (long long)(b->occ_fast < 1 ? 1 : b->occ_fast) * cu, g_gen = (long long)(b->occ_gen < 1 ? 1 : b->occ_gen) * cu;
   if (g_fast > b->n_act) g_fast = b->n_act;
   if (g_gen > b->n_act) g_gen = b->n_act;
   const size_t need = (size_t)(g_fast > g_gen ? g_fast : g_gen) * OA_DEC_SCRATCH_BYTES;
   if (need > b->scratch_cap) { HIPCHECK(hipStreamSynchronize(s)); if (b->d_scratch) (void)hipFree(b->d_scratch); b->d_scratch = nullptr; b->scratch_cap = 0; HIPCHECK(hipMalloc((void **)&b->d_scratch, need)); b->scratch_cap = need; }
   HIPCHECK(hipMemsetAsync(b->d_queue, 0, 64, s));
   const int use_fast = fast_env && !b->no_fast && !b->decode_fec;
   const int use_lane = fast_env && lane_env && !b->no_lane && !b->decode_fec, use_look = use_fast || use_lane;
   long long g_lane = 0; int lane_tw = SL_STREAMS;
   if (use_lane) {                                                    /* one block per tile, as many as the chip holds at most */
      static const int tw_env = getenv("OPUS_AMD_SDEC_TILE") ? atoi(getenv("OPUS_AMD_SDEC_TILE")) : 0;           /* 16 / 32 / 64: fixed tile width (experiments) */
      const long long slots = (long long)OA_SDEC_WAVES_PER_EU * 4 * cu;
      lane_tw = tw_env >= 1 && tw_env <= SL_STREAMS ? tw_env : (long long)b->n_act >= 4 * cu * SL_STREAMS ? SL_STREAMS : 32;   /* (full waves as soon as every SIMD has one: measured, profiles/r06_r) */
      g_lane = ((long long)b->n_act + lane_tw - 1) / lane_tw;
      if (g_lane > slots) g_lane = slots;
      const size_t lneed = (size_t)g_lane * SL_WORK_BYTES;
      if (lneed > b->lane_work_cap) { HIPCHECK(hipStreamSynchronize(s)); if (b->d_lane_work) (void)hipFree(b->d_lane_work); b->d_lane_work = nullptr; b->lane_work_cap = 0; HIPCHECK(hipMalloc((void **)&b->d_lane_work, lneed)); b->lane_work_cap = lneed; }
   }
   /* d_queue: [0] the fast kernel's queue, [1] the general kernel's, [2] / [3] / [4] the lengths of the fast, general and lane lists; d_slow: [S] the fast list, [S] the
    * general kernel's list, [S] the lane kernel's list */
   if (use_look) {
      hipLaunchKernelGGL(oa_decode_look_kernel, dim3((unsigned)((b->n_act + 63) / 64)), dim3(64), 0, s,
            (const OaDecStream *)b->d_streams, (const u8 *)d_packets, (int)packet_stride, (const i32 *)d_lens, (int)b->n_act, frame_size, use_fast ? b->d_slow : (int *)nullptr, b->d_slow + b->S,
            use_lane ? b->d_slow + 2 * (size_t)b->S : (int *)nullptr, b->d_queue + 2);
      if (use_lane)
         hipLaunchKernelGGL(oa_sdec_lane_kernel, dim3((unsigned)g_lane), dim3(64), sizeof(ResamplerLds) + sizeof(SlTabs) + SL_STREAMS * SL_WIN_STRIDE, s,
               b->d_streams, (const u8 *)d_packets, (int)packet_stride, (const i32 *)d_lens, (i16 *)d_pcm, frame_size * b->channels, (i32 *)d_nsamples, (u32 *)d_final_range, b->d_lane_work, lane_tw,
               (const int *)(b->d_slow + 2 * (size_t)b->S), (const unsigned *)(b->d_queue + 4), b->d_slow + b->S, b->d_queue + 3, b->d_queue + 5,
               b->d_slow + 3 * (size_t)b->S, b->d_queue + 6, b->d_hyb_ec);
      /* the kernel pipeline of the steady-state CELT frames (opusgpu_dec_batch_set_pvq_stage; process default OPUS_AMD_DEC_PVQ4): the two kernels below stop a packet of one
       * 10 / 20 ms frame in front of its bands, oa_celt_dpvq_kernel decodes the bands of four streams per wave, oa_celt_dback_kernel finishes the packets.
       * d_queue [8] / [9] the lengths of the 20 ms / 10 ms lists, [10] the PVQ kernel's tile counter, [11] the back kernel's queue */
      static const int pvq_env = getenv("OPUS_AMD_DEC_PVQ4") ? atoi(getenv("OPUS_AMD_DEC_PVQ4")) : -1;
      const int pvq_mode = b->pvq_stage >= 0 ? b->pvq_stage : pvq_env;
      const bool dpipe = (use_fast || use_lane) && (pvq_mode < 0 ? (long long)b->n_act > g_fast : pvq_mode > 0);
      if (dpipe && !b->d_dcont) {
         HIPCHECK(hipMalloc((void **)&b->d_dcont, sizeof(CeltDecCont) * (size_t)b->S));
         HIPCHECK(hipMalloc((void **)&b->d_cut, 2 * sizeof(int) * (size_t)b->S));
      }
      CeltDecCont *const dc = dpipe ? b->d_dcont : (CeltDecCont *)nullptr;
      if (use_fast) hipLaunchKernelGGL(oa_decode_fast_kernel, dim3((unsigned)g_fast), dim3(64), OA_DEC_FAST_LDS_BYTES, s,
            b->d_streams, (const u8 *)d_packets, (int)packet_stride, (const i32 *)d_lens, frame_size, (i16 *)d_pcm, frame_size * b->channels, (i32 *)d_nsamples,
            (u32 *)d_final_range, (int)b->S, b->d_scratch, b->d_queue, (const int *)b->d_slow, (const unsigned *)(b->d_queue + 2), dc, b->d_cut, b->d_queue + 8);
      if (use_lane)                                                   /* (d_queue [6] the hybrid list's length, [7] the kernel's queue; an empty list costs the launch) */
         hipLaunchKernelGGL(oa_decode_hyb_kernel, dim3((unsigned)g_fast), dim3(64), OA_DEC_FAST_LDS_BYTES, s,
               b->d_streams, (const u8 *)d_packets, (int)packet_stride, (const i32 *)d_lens, (i16 *)d_pcm, frame_size * b->channels, (i32 *)d_nsamples, (u32 *)d_final_range,
               b->d_scratch, b->d_queue + 7, (const int *)(b->d_slow + 3 * (size_t)b->S), (const unsigned *)(b->d_queue + 6), (const OaHybCont *)b->d_hyb_ec, dc, b->d_cut, b->d_queue + 8, (int)b->S);
      static const int deemph_lane = getenv("OPUS_AMD_DEC_DEEMPH_LANE") ? atoi(getenv("OPUS_AMD_DEC_DEEMPH_LANE")) : 1;      /* 0: the de-emphasis stays in the back kernel (A/B) */
      if (dpipe) {
         long long g_pvq = (long long)(b->occ_dpvq < 1 ? 1 : b->occ_dpvq) * cu;
         const long long tiles = ((long long)b->n_act + 3) / 4 + 1;
         if (g_pvq > tiles) g_pvq = tiles;
         hipLaunchKernelGGL(oa_celt_dpvq_kernel, dim3((unsigned)g_pvq), dim3(64), sizeof(P4Lds), s, b->d_dcont, (const int *)b->d_cut, (const unsigned *)(b->d_queue + 8), b->d_queue + 10, (int)b->S);
         hipLaunchKernelGGL(oa_celt_dback_kernel, dim3((unsigned)g_fast), dim3(64), OA_DEC_FAST_LDS_BYTES, s,
               b->d_streams, b->d_dcont, (const int *)b->d_cut, (const unsigned *)(b->d_queue + 8), b->d_queue + 11, (i16 *)d_pcm, frame_size * b->channels, (i32 *)d_nsamples, (u32 *)d_final_range, (int)b->S, deemph_lane);
         if (deemph_lane)
            hipLaunchKernelGGL(oa_celt_deemph_kernel, dim3((unsigned)((b->n_act + 63) / 64)), dim3(64), 0, s,
                  b->d_streams, (const CeltDecCont *)b->d_dcont, (const int *)b->d_cut, (const unsigned *)(b->d_queue + 8), (i16 *)d_pcm, frame_size * b->channels, (int)b->S);
      }
   }
   hipLaunchKernelGGL(oa_decode_kernel, dim3((unsigned)g_gen), dim3(64), sizeof(DecLds), s,
         b->d_streams, (const u8 *)d_packets, (int)packet_stride, (const i32 *)d_lens, frame_size, (i16 *)d_pcm, frame_size * b->channels, (i32 *)d_nsamples,
         (u32 *)d_final_range, (int)b->n_act, b->decode_fec, b->d_scratch, b->d_queue + 1, use_look ? (const int *)(b->d_slow + b->S) : (const int *)nullptr, (const unsigned *)(b->d_queue + 3));
   HIPCHECK(hipGetLastError());
   return OPUS_OK;
}
int opusgpu_decode_batch(OpusGpuDecBatch *b, const unsigned char *packets, opus_int32 packet_stride, const opus_int32 *lens, opus_int16 *pcm,
      int frame_size, opus_int32 *nsamples, opus_uint32 *final_range)
{
   if (!b || !packets || !lens || !pcm || !nsamples || packet_stride <= 0) return OPUS_BAD_ARG;
   if (frame_size <= 0 || frame_size > 5760) return OPUS_BAD_ARG;
   HIPCHECK(hipSetDevice(b->device));
   size_t npkt = (size_t)b->n_act * packet_stride, npcm = (size_t)b->n_act * frame_size * b->channels * sizeof(opus_int16);
   if (npkt > b->pkt_cap) { if (b->d_pkt) (void)hipFree(b->d_pkt); HIPCHECK(hipMalloc((void **)&b->d_pkt, npkt)); b->pkt_cap = npkt; }
   if (npcm > b->pcm_cap) { if (b->d_pcm) (void)hipFree(b->d_pcm); HIPCHECK(hipMalloc((void **)&b->d_pcm, npcm)); b->pcm_cap = npcm; }
   HIPCHECK(hipMemcpyAsync(b->d_pkt, packets, npkt, hipMemcpyHostToDevice, b->stream));
   HIPCHECK(hipMemcpyAsync(b->d_lens, lens, sizeof(opus_int32) * (size_t)b->n_act, hipMemcpyHostToDevice, b->stream));
   HIPCHECK(hipMemsetAsync(b->d_pcm, 0, npcm, b->stream));
   int r = opusgpu_decode_batch_dev(b, b->d_pkt, packet_stride, b->d_lens, b->d_pcm, frame_size, b->d_ns, b->d_rng, nullptr);
   if (r != OPUS_OK) return r;
   HIPCHECK(hipMemcpyAsync(pcm, b->d_pcm, npcm, hipMemcpyDeviceToHost, b->stream));
   HIPCHECK(hipMemcpyAsync(nsamples, b->d_ns, sizeof(opus_int32) * (size_t)b->n_act, hipMemcpyDeviceToHost, b->stream));
   if (final_range) HIPCHECK(hipMemcpyAsync(final_range, b->d_rng, sizeof(opus_uint32) * (size_t)b->n_act, hipMemcpyDeviceToHost, b->stream));
   HIPCHECK(hipStreamSynchronize(b->stream));
   return OPUS_OK;
}
int opusgpu_time_decode_dev(OpusGpuDecBatch *b, const unsigned char *d_packets, opus_int32 packet_stride, const opus_int32 *d_lens, opus_int16 *d_pcm,
      int frame_size, opus_int32 *d_nsamples, opus_uint32 *d_final_range, int steps, float *ms)
{
   if (!b || !ms || steps <= 0) return OPUS_BAD_ARG;
   HIPCHECK(hipSetDevice(b->device));
   hipEvent_t e0, e1;
   HIPCHECK(hipEventCreate(&e0)); HIPCHECK(hipEventCreate(&e1));
   HIPCHECK(hipEventRecord(e0, b->stream));
   for (int k = 0; k < steps; k++) {
      int r = opusgpu_decode_batch_dev(b, d_packets + (size_t)k * b->S * packet_stride, packet_stride, d_lens + (size_t)k * b->S, d_pcm, frame_size, d_nsamples, d_final_range, nullptr);
      if (r != OPUS_OK) { (void)hipEventDestroy(e0); (void)hipEventDestroy(e1); return r; }
   }
   HIPCHECK(hipEventRecord(e1, b->stream));
   HIPCHECK(hipEventSynchronize(e1));
   HIPCHECK(hipEventElapsedTime(ms, e0, e1));
   (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
   return OPUS_OK;
}

/* ---- classic decoder API: flat host blob, one wave per call, concurrent calls share a launch (opus_call_combiner.h) (reference src/opus_decoder.c:121 get_size, :135 init, :186 create,
 *      :890 opus_decode, :1033 ctl, :1246 destroy) ---- */
#define OA_DEC_MAGIC 0x4f414443u
#define OPUS_SET_GAIN_REQUEST 4034
#define OPUS_GET_GAIN_REQUEST 4045
struct OpusDecoder { opus_uint32 magic; opus_int32 Fs; opus_int32 decode_gain; opus_int32 pad[1]; OaDecStream s; };
static OpusGpuDecBatch *g_classic_dec[5][2];
struct OaDecCall {
   OpusDecoder *st; const unsigned char *data; opus_int32 len; opus_int16 *pcm; int frame_size, decode_fec, channels; opus_int32 Fs;
   int ret; bool done; size_t tid;
   const void *who() const { return st; }
   bool same_shape(const OaDecCall &o) const { return Fs == o.Fs && channels == o.channels && frame_size == o.frame_size && decode_fec == o.decode_fec; }
};
static OaCallCombiner<OaDecCall> g_dec_calls;
int opus_decoder_get_size(int channels) { return (channels < 1 || channels > 2) ? 0 : (int)sizeof(OpusDecoder); }
int opus_decoder_init(OpusDecoder *st, opus_int32 Fs, int channels)
{
   if (!st) return OPUS_BAD_ARG;
   OaDecStream *tmp = new OaDecStream;
   int r = oa_dec_init_stream(tmp, Fs, channels);
   if (r == OPUS_OK) { memset(st, 0, sizeof(*st)); st->magic = OA_DEC_MAGIC; st->Fs = Fs; st->s = *tmp; }
   delete tmp;
   return r;
}
OpusDecoder *opus_decoder_create(opus_int32 Fs, int channels, int *error)
{
   if ((Fs != 48000 && Fs != 24000 && Fs != 16000 && Fs != 12000 && Fs != 8000) || (channels != 1 && channels != 2)) { if (error) *error = OPUS_BAD_ARG; return nullptr; }
   OpusDecoder *st = (OpusDecoder *)malloc(sizeof(OpusDecoder));
   if (!st) { if (error) *error = OPUS_ALLOC_FAIL; return nullptr; }
   int r = opus_decoder_init(st, Fs, channels);
   if (error) *error = r;
   if (r != OPUS_OK) { free(st); return nullptr; }
   return st;
}
void opus_decoder_destroy(OpusDecoder *st) { free(st); }
/* OPUS_SET_GAIN (src/opus_decoder.c:700-712): gain = celt_exp2(6.48814081e-4 * decode_gain) in Q16-ish fixed point, applied with saturation */
static opus_int32 oa_decode_gain_q16(int decode_gain)
{
   if (!decode_gain) return 0;
   /* celt_exp2(MULT16_16_P15(QCONST16(6.48814081e-4f, 25), decode_gain)) with the reference's fixed-point celt_exp2 (celt/mathops.h) */
   const opus_int32 x = ((opus_int32)21771 * (opus_int16)decode_gain + 16384) >> 15;   /* QCONST16(6.48814081e-4, 25) = 21771; result Q10 */
   opus_int32 gain;
   {  /* celt_exp2(x), x in Q10 -> Q16 */
      const int integer = x >> 10;
      if (integer > 14) gain = 0x7f000000;
      else if (integer < -15) gain = 0;
      else {
         const opus_int32 fr = (opus_int32)(opus_int16)(x - (integer << 10)) << 4;      /* celt_exp2_frac, Q14 */
         opus_int32 f = 16383 + (((opus_int32)(opus_int16)fr * (22804 + (((opus_int32)(opus_int16)fr * (14819 + ((10204 * (opus_int32)(opus_int16)fr) >> 15))) >> 15))) >> 15);
         gain = integer + 2 >= 0 ? f << (integer + 2) : f >> (-2 - integer);                 /* VSHR32(frac, -integer-2) */
      }
   }
   return gain;
}
static void oa_apply_decode_gain(opus_int16 *pcm, int n, int decode_gain)
{
   const opus_int32 gain = oa_decode_gain_q16(decode_gain);
   for (int i = 0; i < n; i++) {
      const opus_int32 y = (opus_int32)(((long long)(opus_int16)pcm[i] * gain + 32768) >> 16);   /* MULT16_32_P16 */
      pcm[i] = (opus_int16)(y > 32767 ? 32767 : y < -32767 ? -32767 : y);
   }
}
/* one launch for a group of classic decode calls of one shape (rate, channels, output size, FEC flag): packets of any mode and length side by side, one wave each */
static int oa_classic_decode_group_run(std::vector<OaDecCall *> &g)
{
   const OaDecCall &h = *g[0];
   const int n = (int)g.size(), ch = h.st->s.s.channels, ci = ch - 1, fi = oa_fs_index(h.st->Fs);
   if (!g_classic_dec[fi][ci]) {
      int err;
      g_classic_dec[fi][ci] = opusgpu_dec_batch_create(oa_classic_cap(), h.st->Fs, ch, oa_classic_device(), &err);
      if (!g_classic_dec[fi][ci]) return err == OPUS_OK ? OPUS_INTERNAL_ERROR : err;
   }
   OpusGpuDecBatch *b = g_classic_dec[fi][ci];
   b->n_act = n; b->decode_fec = h.decode_fec;
   opus_int32 stride = 8;
   for (int i = 0; i < n; i++) if (g[i]->len + 8 > stride) stride = g[i]->len + 8;
   std::vector<unsigned char> pkt((size_t)stride * n, 0);
   std::vector<opus_int32> lens((size_t)n), ns((size_t)n); std::vector<opus_uint32> rng((size_t)n);
   char *recs = oa_pinned(g_dec_pin, sizeof(OaDecStream) * (size_t)oa_classic_cap());
   if (!recs) return OPUS_ALLOC_FAIL;
   const size_t recs_bytes = sizeof(OaDecStream) * (size_t)n;
   for (int i = 0; i < n; i++) {
      if (g[i]->len > 0) memcpy(pkt.data() + (size_t)stride * i, g[i]->data, (size_t)g[i]->len);
      lens[i] = g[i]->len;
      g[i]->st->s.s.transition_gain_Q16 = oa_decode_gain_q16(g[i]->st->decode_gain);
      memcpy(recs + sizeof(OaDecStream) * i, &g[i]->st->s, sizeof(OaDecStream));
   }
   const size_t per = (size_t)h.frame_size * ch;
   std::vector<opus_int16> out(per * n);
   HIPCHECK(hipSetDevice(b->device));
   HIPCHECK(hipStreamSynchronize(b->stream));
   HIPCHECK(hipMemcpy(b->d_streams, recs, recs_bytes, hipMemcpyHostToDevice));
   int r = opusgpu_decode_batch(b, pkt.data(), stride, lens.data(), out.data(), h.frame_size, ns.data(), rng.data());
   if (r != OPUS_OK) return r;
   HIPCHECK(hipMemcpy(recs, b->d_streams, recs_bytes, hipMemcpyDeviceToHost));
   for (int i = 0; i < n; i++) {
      OpusDecoder *st = g[i]->st;
      memcpy(&st->s, recs + sizeof(OaDecStream) * i, sizeof(OaDecStream));
      const opus_int32 m = ns[i];
      if (m > 0) {
         opus_int16 *o = out.data() + per * i;
         if (st->decode_gain) oa_apply_decode_gain(o, m * ch, st->decode_gain);
         memcpy(g[i]->pcm, o, (size_t)m * ch * sizeof(opus_int16));
      }
      g[i]->ret = m;
   }
   return OPUS_OK;
}
static void oa_classic_decode_group(std::vector<OaDecCall *> &g)
{
   const int r = oa_classic_decode_group_run(g);
   if (r != OPUS_OK) for (OaDecCall *c : g) c->ret = r;
}
int opus_decode(OpusDecoder *st, const unsigned char *data, opus_int32 len, opus_int16 *pcm, int frame_size, int decode_fec)
{
   if (!st || st->magic != OA_DEC_MAGIC || !pcm) return OPUS_BAD_ARG;
   if (frame_size <= 0 || decode_fec < 0 || decode_fec > 1) return OPUS_BAD_ARG;
   if ((decode_fec || len == 0 || data == nullptr) && frame_size % (st->Fs / 400) != 0) return OPUS_BAD_ARG;
   if (data == nullptr) len = 0;                    /* packet loss (takes precedence over a negative length, src/opus_decoder.c:738-750) */
   if (len < 0) return OPUS_BAD_ARG;
   if (len == 0) decode_fec = 0;
   const int cap = st->Fs / 25 * 3;                 /* 120 ms */
   if (len == 0 && frame_size > cap) {              /* a loss longer than the longest packet: the reference conceals all of it, 20 ms or less at a time (src/opus_decoder.c:756-769) */
      int total = 0;
      while (total < frame_size) {
         const int r = opus_decode(st, nullptr, 0, pcm + (size_t)total * st->s.s.channels, frame_size - total < cap ? frame_size - total : cap, 0);
         if (r < 0) return r;
         total += r;
      }
      st->s.s.last_packet_duration = total;
      return total;
   }
   if (frame_size > cap) frame_size = cap;
   OaDecCall call = {st, data, len, pcm, frame_size, decode_fec, st->s.s.channels, st->Fs, OPUS_INTERNAL_ERROR, false};
   g_dec_calls.submit(&call, oa_classic_cap(), oa_classic_linger_us(), oa_classic_decode_group);
   return call.ret;
}
/* opus_decode24 / opus_decode_float (reference include/opus.h:541,:566; src/opus_decoder.c:947-1030, the int16-resolution build): the frame size is
 * first limited to what the packet holds (so the temporary is no larger than needed), then decode, then RES2INT24 = << 8 / RES2FLOAT = * 1/32768 */
static int oa_decode_limit(OpusDecoder *st, const unsigned char *data, opus_int32 len, int frame_size, int decode_fec)
{
   if (frame_size <= 0) return OPUS_BAD_ARG;
   if (data != nullptr && len > 0 && !decode_fec) {
      const int nb = opus_packet_get_nb_samples(data, len, st->Fs);
      if (nb > 0) frame_size = frame_size < nb ? frame_size : nb; else return OPUS_INVALID_PACKET;
   }
   return frame_size;
}
int opus_decode24(OpusDecoder *st, const unsigned char *data, opus_int32 len, opus_int32 *pcm, int frame_size, int decode_fec)
{
   if (!st || st->magic != OA_DEC_MAGIC || !pcm) return OPUS_BAD_ARG;
   frame_size = oa_decode_limit(st, data, len, frame_size, decode_fec);
   if (frame_size < 0) return frame_size;
   std::vector<opus_int16> out((size_t)frame_size * st->s.s.channels);
   const int ret = opus_decode(st, data, len, out.data(), frame_size, decode_fec);
   for (int i = 0; i < ret * st->s.s.channels; i++) pcm[i] = (opus_int32)out[i] * 256;
   return ret;
}
int opus_decode_float(OpusDecoder *st, const unsigned char *data, opus_int32 len, float *pcm, int frame_size, int decode_fec)
{
   if (!st || st->magic != OA_DEC_MAGIC || !pcm) return OPUS_BAD_ARG;
   frame_size = oa_decode_limit(st, data, len, frame_size, decode_fec);
   if (frame_size < 0) return frame_size;
   std::vector<opus_int16> out((size_t)frame_size * st->s.s.channels);
   const int ret = opus_decode(st, data, len, out.data(), frame_size, decode_fec);
   for (int i = 0; i < ret * st->s.s.channels; i++) pcm[i] = (1.f / 32768.f) * out[i];
   return ret;
}
/* how the classic entry points were served so far: {opus_encode* calls, launches they shared, opus_decode* calls, launches they shared} (opus_call_combiner.h) */
void opusgpu_classic_call_stats(long long out[4])
{
   if (!out) return;
   { std::lock_guard<std::mutex> l(g_enc_calls.mu); out[0] = g_enc_calls.calls; out[1] = g_enc_calls.launches; }
   { std::lock_guard<std::mutex> l(g_dec_calls.mu); out[2] = g_dec_calls.calls; out[3] = g_dec_calls.launches; }
}
int opus_decoder_get_nb_samples(const OpusDecoder *dec, const unsigned char packet[], opus_int32 len) { return opus_packet_get_nb_samples(packet, len, dec->Fs); }
int opus_decoder_ctl(OpusDecoder *st, int request, ...)
{
   if (!st || st->magic != OA_DEC_MAGIC) return OPUS_BAD_ARG;
   va_list ap;
   va_start(ap, request);
   int ret = OPUS_OK;
   switch (request) {
   case OPUS_RESET_STATE: { OaDecStream *tmp = new OaDecStream; oa_dec_init_stream(tmp, st->Fs, st->s.s.channels); tmp->s.disable_inv = st->s.s.disable_inv; st->s = *tmp; delete tmp; } break;
   case OPUS_GET_FINAL_RANGE_REQUEST: { opus_uint32 *p = va_arg(ap, opus_uint32 *); if (!p) ret = OPUS_BAD_ARG; else *p = st->s.s.rangeFinal; } break;
   case OPUS_GET_SAMPLE_RATE_REQUEST: { opus_int32 *p = va_arg(ap, opus_int32 *); if (!p) ret = OPUS_BAD_ARG; else *p = st->Fs; } break;
   case OPUS_GET_BANDWIDTH_REQUEST: { opus_int32 *p = va_arg(ap, opus_int32 *); if (!p) ret = OPUS_BAD_ARG; else *p = st->s.s.bandwidth; } break;
   case 4039 /* OPUS_GET_LAST_PACKET_DURATION */: { opus_int32 *p = va_arg(ap, opus_int32 *); if (!p) ret = OPUS_BAD_ARG; else *p = st->s.s.last_packet_duration; } break;
   case 4033 /* OPUS_GET_PITCH (:1090): the CELT post-filter period, or the SILK lag scaled to 48 kHz */: {
      opus_int32 *p = va_arg(ap, opus_int32 *);
      if (!p) ret = OPUS_BAD_ARG;
      else if (st->s.s.prev_mode == 1002) *p = st->s.s.postfilter_period;
      else *p = st->s.silk.ch[0].prevSignalType == 2 ? st->s.silk.ch[0].lagPrev * 48 / (st->s.silk.ch[0].fs_kHz ? st->s.silk.ch[0].fs_kHz : 16) : 0;
   } break;
   case OPUS_SET_GAIN_REQUEST: { opus_int32 v = va_arg(ap, opus_int32); if (v < -32768 || v > 32767) ret = OPUS_BAD_ARG; else st->decode_gain = v; } break;
   case OPUS_GET_GAIN_REQUEST: { opus_int32 *p = va_arg(ap, opus_int32 *); if (!p) ret = OPUS_BAD_ARG; else *p = st->decode_gain; } break;
   case OPUS_SET_COMPLEXITY_REQUEST: { opus_int32 v = va_arg(ap, opus_int32); if (v < 0 || v > 10) ret = OPUS_BAD_ARG; else st->pad[0] = (st->pad[0] & ~0xff) | v; } break;   /* decoder complexity only gates the DNN options (:1041) */
   case OPUS_GET_COMPLEXITY_REQUEST: { opus_int32 *p = va_arg(ap, opus_int32 *); if (!p) ret = OPUS_BAD_ARG; else *p = st->pad[0] & 0xff; } break;
   /* OPUS_SET / GET_IGNORE_EXTENSIONS (:1199-1217): stored (bit 8 of the word the complexity lives in) and read back; this decoder never looks at the extensions in a packet's padding either way */
   case 4058: { opus_int32 v = va_arg(ap, opus_int32); if (v < 0 || v > 1) ret = OPUS_BAD_ARG; else st->pad[0] = (st->pad[0] & ~0x100) | (v << 8); } break;
   case 4059: { opus_int32 *p = va_arg(ap, opus_int32 *); if (!p) ret = OPUS_BAD_ARG; else *p = (st->pad[0] >> 8) & 1; } break;
   case OPUS_SET_PHASE_INVERSION_DISABLED_REQUEST: { opus_int32 v = va_arg(ap, opus_int32); if (v < 0 || v > 1) ret = OPUS_BAD_ARG; else st->s.s.disable_inv = v; } break;
   case OPUS_GET_PHASE_INVERSION_DISABLED_REQUEST: { opus_int32 *p = va_arg(ap, opus_int32 *); if (!p) ret = OPUS_BAD_ARG; else *p = st->s.s.disable_inv; } break;
   default: ret = OPUS_UNIMPLEMENTED;
   }
   va_end(ap);
   return ret;
}
} /* extern "C" */
#include "opus_ms_host.h"
#include "opus_api_host.h"
#include "opus_projection_host.h"
#include "opus_ms_dec_batch.h"
#include "opus_ms_batch.h"
#include "silk_batch.h"
extern "C" {
const char *opus_get_version_string(void) { return "opus-amd 0.4 (gfx950; bit-exact fixed-point Opus encoder and decoder: CELT, SILK, hybrid, multistream, projection)"; }

} /* extern "C" */

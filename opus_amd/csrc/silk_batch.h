/* silk_batch.h — kernels and C-ABI of the batched SILK building blocks (included by opus_amd.hip).
 *
 * opusgpu_nsq_batch_*: N independent SILK channels' noise-shaping quantiser states resident in HBM (tile-SoA, silk_frame.h); one
 * call quantises one frame of every stream.  The per-stream inputs are the reference's own argument list (OpusGpuNsqFrame = the
 * parameters of silk_NSQ_c, silk/NSQ.c:76-93), dispatch follows the reference (delayed decision iff nStatesDelayedDecision > 1 ||
 * warping_Q16 > 0, silk/fixed/encode_frame_FIX.c / silk/float/wrappers_FLP.c:163-169). */
#ifndef OPUS_AMD_SILK_BATCH_H
#define OPUS_AMD_SILK_BATCH_H
#include "silk_nsq.h"
#include "silk_nsq_dd.h"
#include "silk_host.h"
#include "silk_lpc.h"
#include "silk_resampler.h"
#include "silk_pitch.h"

template <int SS> __global__ __launch_bounds__(64) void oa_silk_nsq_kernel(OaNsqCfg cfg, i32 *tiles, long tile_words, const OaNsqFrame *frames, const i16 *x16, i8 *pulses, int n)
{
   const int lane = (int)threadIdx.x, first = (int)blockIdx.x * 64;
   int sidx = first + lane; const bool act = sidx < n; if (!act) sidx = first;
   const int frame = cfg.nb_subfr * 5 * cfg.fs_kHz;
   NsqMem m = nsq_mem(tiles + (size_t)blockIdx.x * tile_words, 64, lane, 20 * cfg.fs_kHz + frame);
   silk_nsq_lane<SS>(cfg, m, &frames[sidx], x16 + (size_t)sidx * frame, pulses + (size_t)sidx * frame, act);
}

template <int SS> __global__ __launch_bounds__(64) void oa_silk_nsq_dd_kernel(OaNsqCfg cfg, i32 *tiles, long tile_words, const OaNsqFrame *frames, const i16 *x16, i8 *pulses, i8 *seed_out, int n)
{
   const int lane = (int)threadIdx.x, first = (int)blockIdx.x * 16;
   int sidx = first + (lane >> 2); const bool act = sidx < n; if (!act) sidx = first;
   const int frame = cfg.nb_subfr * 5 * cfg.fs_kHz;
   i32 *tile = tiles + (size_t)blockIdx.x * tile_words;
   NsqMem m = nsq_mem(tile, 16, lane >> 2, 20 * cfg.fs_kHz + frame);
   silk_nsq_dd_wave<SS>(cfg, m, tile + (tile_words - 5 * OA_SILK_DD * 64), &frames[sidx], x16 + (size_t)sidx * frame, pulses + (size_t)sidx * frame, seed_out + sidx, act);
}

__global__ __launch_bounds__(64) void oa_silk_lpc_analysis_kernel(i16 *out, const i16 *in, const i16 *B, int len, int d)
{
   __shared__ LpcLds lds;
   const size_t s = blockIdx.x;
   silk_lpc_analysis_filter_wave((WV_LDS LpcLds *)&lds, out + s * (size_t)len, in + s * (size_t)len, B + s * (size_t)d, len, d);
}

__global__ __launch_bounds__(64) void oa_silk_resampler_kernel(OaResamplerCfg cfg, i32 *state, int n, const i16 *in, int inLen, i16 *out, int outLen)
{
   __shared__ ResamplerLds lds;
   const int ch = (int)blockIdx.x * 64 + (int)threadIdx.x;
   if (ch >= n) return;                              /* no cross-lane operation anywhere in this kernel */
   silk_resampler_lane(cfg, (WV_LDS ResamplerLds *)&lds, state + ch, n, in + (size_t)ch * inLen, inLen, out + (size_t)ch * outLen, (int)threadIdx.x);
}

__global__ __launch_bounds__(64) void oa_silk_pitch_kernel(OaPitchCfg cfg, const i16 *frames, int flen, const OaPitchIn *in, OaPitchOut *out)
{
   __shared__ PitchLds lds;
   const size_t s = blockIdx.x;
   (void)silk_pitch_analysis_wave(cfg, (WV_LDS PitchLdsCore *)(WV_LDS PitchLds *)&lds, ((WV_LDS PitchLds *)&lds)->frame, frames + s * (size_t)flen, in + s, out + s);
}

extern "C" {
/* ---- silk_pitch_analysis_core for n independent analysis buffers (silk/fixed/pitch_analysis_core_FIX.c:82) ---- */
int opusgpu_silk_pitch_analysis_batch_dev(int device, opus_int32 n, const opus_int16 *d_frames, const OpusGpuPitchIn *d_in, OpusGpuPitchOut *d_out,
      int Fs_kHz, int complexity, int nb_subfr, void *hip_stream)
{
   if (n <= 0 || !d_frames || !d_in || !d_out || (Fs_kHz != 8 && Fs_kHz != 12 && Fs_kHz != 16) || complexity < 0 || complexity > 2 || (nb_subfr != 2 && nb_subfr != 4)) return OPUS_BAD_ARG;   /* the reference's asserts (:131-135) */
   HIPCHECK(hipSetDevice(device));
   OaPitchCfg c = { Fs_kHz, complexity, nb_subfr };
   hipLaunchKernelGGL(oa_silk_pitch_kernel, dim3((unsigned)n), dim3(64), 0, (hipStream_t)hip_stream, c, (const i16 *)d_frames, (20 + 5 * nb_subfr) * Fs_kHz, (const OaPitchIn *)d_in, (OaPitchOut *)d_out);
   HIPCHECK(hipGetLastError());
   return OPUS_OK;
}
int opusgpu_silk_pitch_analysis_batch(int device, opus_int32 n, const opus_int16 *frames, const OpusGpuPitchIn *in, OpusGpuPitchOut *out, int Fs_kHz, int complexity, int nb_subfr)
{
   if (n <= 0 || !frames || !in || !out || (Fs_kHz != 8 && Fs_kHz != 12 && Fs_kHz != 16) || (nb_subfr != 2 && nb_subfr != 4)) return OPUS_BAD_ARG;
   int ndev = 0;
   if (hipGetDeviceCount(&ndev) != hipSuccess || device < 0 || device >= ndev) { fprintf(stderr, "opus_amd: no usable HIP device (requested %d of %d) — this library has no CPU fallback\n", device, ndev); return OPUS_INTERNAL_ERROR; }
   HIPCHECK(hipSetDevice(device));
   const size_t fb = sizeof(opus_int16) * (size_t)n * (size_t)((20 + 5 * nb_subfr) * Fs_kHz);
   opus_int16 *d_f = nullptr; OpusGpuPitchIn *d_i = nullptr; OpusGpuPitchOut *d_o = nullptr;
   int r = OPUS_OK;
   if (hipMalloc((void **)&d_f, fb) != hipSuccess || hipMalloc((void **)&d_i, sizeof(OpusGpuPitchIn) * (size_t)n) != hipSuccess || hipMalloc((void **)&d_o, sizeof(OpusGpuPitchOut) * (size_t)n) != hipSuccess) r = OPUS_ALLOC_FAIL;
   if (r == OPUS_OK && (hipMemcpy(d_f, frames, fb, hipMemcpyHostToDevice) != hipSuccess || hipMemcpy(d_i, in, sizeof(OpusGpuPitchIn) * (size_t)n, hipMemcpyHostToDevice) != hipSuccess)) r = OPUS_INTERNAL_ERROR;
   if (r == OPUS_OK) r = opusgpu_silk_pitch_analysis_batch_dev(device, n, d_f, d_i, d_o, Fs_kHz, complexity, nb_subfr, nullptr);
   if (r == OPUS_OK && (hipDeviceSynchronize() != hipSuccess || hipMemcpy(out, d_o, sizeof(OpusGpuPitchOut) * (size_t)n, hipMemcpyDeviceToHost) != hipSuccess)) r = OPUS_INTERNAL_ERROR;
   if (d_f) (void)hipFree(d_f);
   if (d_i) (void)hipFree(d_i);
   if (d_o) (void)hipFree(d_o);
   return r;
}

/* ---- silk_resampler for n independent channels of one rate pair (silk/resampler.c:79 init, :183 run) ---- */
struct OpusGpuResamplerBatch { int device; opus_int32 n; OaResamplerCfg cfg; hipStream_t stream; i32 *d_state; opus_int16 *d_in, *d_out; size_t in_cap, out_cap; };
int opusgpu_resampler_state_size(void) { return (int)sizeof(OaResamplerState); }
void opusgpu_resampler_batch_destroy(OpusGpuResamplerBatch *b)
{
   if (!b) return;
   (void)hipSetDevice(b->device);
   if (b->stream) (void)hipStreamSynchronize(b->stream);
   if (b->d_state) (void)hipFree(b->d_state);
   if (b->d_in) (void)hipFree(b->d_in);
   if (b->d_out) (void)hipFree(b->d_out);
   if (b->stream) (void)hipStreamDestroy(b->stream);
   delete b;
}
int opusgpu_resampler_batch_reset(OpusGpuResamplerBatch *b)
{
   if (!b) return OPUS_BAD_ARG;
   HIPCHECK(hipSetDevice(b->device));
   HIPCHECK(hipMemsetAsync(b->d_state, 0, sizeof(i32) * OA_RS_ROWS * (size_t)b->n, b->stream));
   HIPCHECK(hipStreamSynchronize(b->stream));
   return OPUS_OK;
}
OpusGpuResamplerBatch *opusgpu_resampler_batch_create(opus_int32 nchannels, opus_int32 Fs_Hz_in, opus_int32 Fs_Hz_out, int forEnc, int device, int *error)
{
   int err = OPUS_OK; OpusGpuResamplerBatch *b = nullptr; OaResamplerCfg c;
   if (nchannels <= 0 || rs_init_cfg(&c, Fs_Hz_in, Fs_Hz_out, forEnc) != 0) err = OPUS_BAD_ARG;
   if (err == OPUS_OK) {
      int ndev = 0;
      if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0 || device < 0 || device >= ndev) {
         fprintf(stderr, "opus_amd: no usable HIP device (requested %d of %d) — this library has no CPU fallback\n", device, ndev);
         err = OPUS_INTERNAL_ERROR;
      }
   }
   if (err == OPUS_OK) {
      b = new OpusGpuResamplerBatch(); memset(b, 0, sizeof *b);
      b->device = device; b->n = nchannels; b->cfg = c;
      bool ok = hipSetDevice(device) == hipSuccess && hipStreamCreate(&b->stream) == hipSuccess &&
                hipMalloc((void **)&b->d_state, sizeof(i32) * OA_RS_ROWS * (size_t)nchannels) == hipSuccess && opusgpu_resampler_batch_reset(b) == OPUS_OK;
      if (!ok) { opusgpu_resampler_batch_destroy(b); b = nullptr; err = OPUS_ALLOC_FAIL; }
   }
   if (error) *error = err;
   return b;
}
opus_int32 opusgpu_resampler_batch_out_len(const OpusGpuResamplerBatch *b, opus_int32 inLen)
{ return (!b || inLen < b->cfg.Fs_in_kHz || inLen % b->cfg.Fs_in_kHz) ? OPUS_BAD_ARG : inLen / b->cfg.Fs_in_kHz * b->cfg.Fs_out_kHz; }
int opusgpu_resampler_batch_run_dev(OpusGpuResamplerBatch *b, opus_int16 *d_out, const opus_int16 *d_in, opus_int32 inLen, void *hip_stream)
{
   if (!b || !d_out || !d_in) return OPUS_BAD_ARG;
   const opus_int32 outLen = opusgpu_resampler_batch_out_len(b, inLen);           /* whole milliseconds, >= 1 ms (silk/resampler.c:193) */
   if (outLen < 0) return OPUS_BAD_ARG;
   HIPCHECK(hipSetDevice(b->device));
   hipLaunchKernelGGL(oa_silk_resampler_kernel, dim3((unsigned)((b->n + 63) / 64)), dim3(64), 0, hip_stream ? (hipStream_t)hip_stream : b->stream, b->cfg, b->d_state, (int)b->n,
                      (const i16 *)d_in, (int)inLen, (i16 *)d_out, (int)outLen);
   HIPCHECK(hipGetLastError());
   return OPUS_OK;
}
int opusgpu_resampler_batch_sync(OpusGpuResamplerBatch *b) { if (!b) return OPUS_BAD_ARG; HIPCHECK(hipSetDevice(b->device)); HIPCHECK(hipStreamSynchronize(b->stream)); return OPUS_OK; }
int opusgpu_resampler_batch_run(OpusGpuResamplerBatch *b, opus_int16 *out, const opus_int16 *in, opus_int32 inLen)
{
   if (!b || !out || !in) return OPUS_BAD_ARG;
   const opus_int32 outLen = opusgpu_resampler_batch_out_len(b, inLen);
   if (outLen < 0) return OPUS_BAD_ARG;
   HIPCHECK(hipSetDevice(b->device));
   const size_t ib = sizeof(opus_int16) * (size_t)b->n * inLen, ob = sizeof(opus_int16) * (size_t)b->n * outLen;
   if (ib > b->in_cap) { if (b->d_in) (void)hipFree(b->d_in); b->d_in = nullptr; b->in_cap = 0; if (hipMalloc((void **)&b->d_in, ib) != hipSuccess) return OPUS_ALLOC_FAIL; b->in_cap = ib; }
   if (ob > b->out_cap) { if (b->d_out) (void)hipFree(b->d_out); b->d_out = nullptr; b->out_cap = 0; if (hipMalloc((void **)&b->d_out, ob) != hipSuccess) return OPUS_ALLOC_FAIL; b->out_cap = ob; }
   HIPCHECK(hipMemcpyAsync(b->d_in, in, ib, hipMemcpyHostToDevice, b->stream));
   int r = opusgpu_resampler_batch_run_dev(b, b->d_out, b->d_in, inLen, nullptr);
   if (r != OPUS_OK) return r;
   HIPCHECK(hipMemcpyAsync(out, b->d_out, ob, hipMemcpyDeviceToHost, b->stream));
   HIPCHECK(hipStreamSynchronize(b->stream));
   return OPUS_OK;
}
static int oa_resampler_state_rw(OpusGpuResamplerBatch *b, opus_int32 i, const OaResamplerState *in, OaResamplerState *out)
{
   if (!b || i < 0 || i >= b->n) return OPUS_BAD_ARG;
   HIPCHECK(hipSetDevice(b->device));
   HIPCHECK(hipStreamSynchronize(b->stream));
   i32 rows[OA_RS_ROWS];
   const bool fir16 = b->cfg.resampler_function == OA_RS_FN_IIR_FIR;
   if (out) {
      HIPCHECK(hipMemcpy2D(rows, sizeof(i32), b->d_state + i, sizeof(i32) * (size_t)b->n, sizeof(i32), OA_RS_ROWS, hipMemcpyDeviceToHost));
      memset(out, 0, sizeof *out);
      for (int j = 0; j < 6; j++) out->sIIR[j] = rows[OA_RS_ROW_IIR + j];
      for (int j = 0; j < 36; j++) { if (fir16) { if (j < 8) out->sFIR.w16[j] = (i16)rows[OA_RS_ROW_FIR + j]; } else out->sFIR.w32[j] = rows[OA_RS_ROW_FIR + j]; }
      for (int j = 0; j < 48; j++) out->delayBuf[j] = (i16)rows[OA_RS_ROW_DELAY + j];
      out->cfg = b->cfg;
   }
   if (in) {
      if (memcmp(&in->cfg, &b->cfg, sizeof(OaResamplerCfg)) != 0) return OPUS_BAD_ARG;       /* a state only fits a batch of its own rate pair */
      for (int j = 0; j < 6; j++) rows[OA_RS_ROW_IIR + j] = in->sIIR[j];
      for (int j = 0; j < 36; j++) rows[OA_RS_ROW_FIR + j] = fir16 ? (j < 8 ? in->sFIR.w16[j] : 0) : in->sFIR.w32[j];
      for (int j = 0; j < 48; j++) rows[OA_RS_ROW_DELAY + j] = in->delayBuf[j];
      HIPCHECK(hipMemcpy2D(b->d_state + i, sizeof(i32) * (size_t)b->n, rows, sizeof(i32), sizeof(i32), OA_RS_ROWS, hipMemcpyHostToDevice));
   }
   return OPUS_OK;
}
int opusgpu_resampler_batch_export_state(OpusGpuResamplerBatch *b, opus_int32 channel, void *state) { return state ? oa_resampler_state_rw(b, channel, nullptr, (OaResamplerState *)state) : OPUS_BAD_ARG; }
int opusgpu_resampler_batch_import_state(OpusGpuResamplerBatch *b, opus_int32 channel, const void *state) { return state ? oa_resampler_state_rw(b, channel, (const OaResamplerState *)state, nullptr) : OPUS_BAD_ARG; }

/* ---- silk_LPC_analysis_filter for n independent signals of one length and order (silk/LPC_analysis_filter.c:49) ---- */
int opusgpu_silk_lpc_analysis_filter_batch_dev(int device, opus_int32 n, opus_int16 *d_out, const opus_int16 *d_in, const opus_int16 *d_B, opus_int32 len, opus_int32 d, void *hip_stream)
{
   if (n <= 0 || !d_out || !d_in || !d_B || d < 6 || d > OA_LPC_MAX_ORDER || (d & 1) || len < d || len > OA_LPC_MAX_LEN) return OPUS_BAD_ARG;   /* the reference's asserts (:65-67) */
   HIPCHECK(hipSetDevice(device));
   hipLaunchKernelGGL(oa_silk_lpc_analysis_kernel, dim3((unsigned)n), dim3(64), 0, (hipStream_t)hip_stream, (i16 *)d_out, (const i16 *)d_in, (const i16 *)d_B, (int)len, (int)d);
   HIPCHECK(hipGetLastError());
   return OPUS_OK;
}
int opusgpu_silk_lpc_analysis_filter_batch(int device, opus_int32 n, opus_int16 *out, const opus_int16 *in, const opus_int16 *B, opus_int32 len, opus_int32 d)
{
   if (n <= 0 || !out || !in || !B) return OPUS_BAD_ARG;
   int ndev = 0;
   if (hipGetDeviceCount(&ndev) != hipSuccess || device < 0 || device >= ndev) { fprintf(stderr, "opus_amd: no usable HIP device (requested %d of %d) — this library has no CPU fallback\n", device, ndev); return OPUS_INTERNAL_ERROR; }
   HIPCHECK(hipSetDevice(device));
   opus_int16 *d_in = nullptr, *d_out = nullptr, *d_B = nullptr;
   const size_t sig = sizeof(opus_int16) * (size_t)n * (size_t)len, cb = sizeof(opus_int16) * (size_t)n * (size_t)d;
   int r = OPUS_OK;
   if (hipMalloc((void **)&d_in, sig) != hipSuccess || hipMalloc((void **)&d_out, sig) != hipSuccess || hipMalloc((void **)&d_B, cb) != hipSuccess) r = OPUS_ALLOC_FAIL;
   if (r == OPUS_OK && (hipMemcpy(d_in, in, sig, hipMemcpyHostToDevice) != hipSuccess || hipMemcpy(d_B, B, cb, hipMemcpyHostToDevice) != hipSuccess)) r = OPUS_INTERNAL_ERROR;
   if (r == OPUS_OK) r = opusgpu_silk_lpc_analysis_filter_batch_dev(device, n, d_out, d_in, d_B, len, d, nullptr);
   if (r == OPUS_OK && (hipDeviceSynchronize() != hipSuccess || hipMemcpy(out, d_out, sig, hipMemcpyDeviceToHost) != hipSuccess)) r = OPUS_INTERNAL_ERROR;
   if (d_in) (void)hipFree(d_in);
   if (d_out) (void)hipFree(d_out);
   if (d_B) (void)hipFree(d_B);
   return r;
}

struct OpusGpuNsqBatch {
   int device; opus_int32 n; OaNsqCfg cfg; int T; long tile_words; opus_int32 ntiles; bool dd; hipStream_t stream;
   i32 *d_tiles; OaNsqFrame *d_frames; opus_int16 *d_x16; opus_int8 *d_pulses, *d_seed;
};

static int oa_nsq_cfg_ok(const OaNsqCfg &c)
{
   if (c.fs_kHz != 8 && c.fs_kHz != 12 && c.fs_kHz != 16) return 0;
   if (c.nb_subfr != 2 && c.nb_subfr != 4) return 0;
   if (c.predictLPCOrder != 10 && c.predictLPCOrder != 16) return 0;
   if (c.shapingLPCOrder < 2 || c.shapingLPCOrder > 24 || (c.shapingLPCOrder & 1)) return 0;
   if (c.nStatesDelayedDecision < 1 || c.nStatesDelayedDecision > 4) return 0;
   if (c.warping_Q16 < 0 || c.warping_Q16 > 32767) return 0;
   return 1;
}
int opusgpu_nsq_state_size(void) { return (int)sizeof(OaNsqRefState); }
opus_int32 opusgpu_nsq_batch_streams(const OpusGpuNsqBatch *b) { return b ? b->n : 0; }
int opusgpu_nsq_batch_frame_length(const OpusGpuNsqBatch *b) { return b ? b->cfg.nb_subfr * 5 * b->cfg.fs_kHz : 0; }
void opusgpu_nsq_batch_destroy(OpusGpuNsqBatch *b)
{
   if (!b) return;
   (void)hipSetDevice(b->device);
   if (b->stream) (void)hipStreamSynchronize(b->stream);
   if (b->d_tiles) (void)hipFree(b->d_tiles);
   if (b->d_frames) (void)hipFree(b->d_frames);
   if (b->d_x16) (void)hipFree(b->d_x16);
   if (b->d_pulses) (void)hipFree(b->d_pulses);
   if (b->d_seed) (void)hipFree(b->d_seed);
   if (b->stream) (void)hipStreamDestroy(b->stream);
   delete b;
}
/* all streams to the encoder's reset state (silk/control_codec.c:247-258: zeroed, lagPrev = 100, prev_gain_Q16 = 65536) */
int opusgpu_nsq_batch_reset(OpusGpuNsqBatch *b)
{
   if (!b) return OPUS_BAD_ARG;
   HIPCHECK(hipSetDevice(b->device));
   std::vector<i32> img((size_t)b->tile_words, 0);
   OaNsqRefState r; memset(&r, 0, sizeof r); r.lagPrev = 100; r.prev_gain_Q16 = 65536;
   for (int t = 0; t < b->T; t++) oa_nsq_import(img.data(), b->T, t, &r, &b->cfg);
   for (opus_int32 tl = 0; tl < b->ntiles; tl++)
      HIPCHECK(hipMemcpyAsync(b->d_tiles + (size_t)tl * b->tile_words, img.data(), sizeof(i32) * (size_t)b->tile_words, hipMemcpyHostToDevice, b->stream));
   HIPCHECK(hipStreamSynchronize(b->stream));
   return OPUS_OK;
}
OpusGpuNsqBatch *opusgpu_nsq_batch_create(opus_int32 nstreams, const OpusGpuNsqConfig *config, int device, int *error)
{
   int err = OPUS_OK;
   OpusGpuNsqBatch *b = nullptr;
   OaNsqCfg c; memset(&c, 0, sizeof c);
   if (nstreams <= 0 || !config) err = OPUS_BAD_ARG;
   else { memcpy(&c, config, sizeof c); if (!oa_nsq_cfg_ok(c)) err = OPUS_BAD_ARG; }
   if (err == OPUS_OK) {
      int ndev = 0;
      if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0 || device < 0 || device >= ndev) {
         fprintf(stderr, "opus_amd: no usable HIP device (requested %d of %d) — this library has no CPU fallback\n", device, ndev);
         err = OPUS_INTERNAL_ERROR;
      }
   }
   if (err == OPUS_OK) {
      b = new OpusGpuNsqBatch();
      memset(b, 0, sizeof *b);
      b->device = device; b->n = nstreams; b->cfg = c;
      b->dd = c.nStatesDelayedDecision > 1 || c.warping_Q16 > 0;
      b->T = b->dd ? 16 : 64; b->tile_words = (long)oa_nsq_tile_words(b->T); b->ntiles = (nstreams + b->T - 1) / b->T;
      const size_t frame = (size_t)c.nb_subfr * 5 * c.fs_kHz;
      bool ok = hipSetDevice(device) == hipSuccess && hipStreamCreate(&b->stream) == hipSuccess &&
                hipMalloc((void **)&b->d_tiles, sizeof(i32) * (size_t)b->tile_words * b->ntiles) == hipSuccess &&
                hipMalloc((void **)&b->d_frames, sizeof(OaNsqFrame) * (size_t)nstreams) == hipSuccess &&
                hipMalloc((void **)&b->d_x16, sizeof(opus_int16) * frame * nstreams) == hipSuccess &&
                hipMalloc((void **)&b->d_pulses, frame * nstreams) == hipSuccess &&
                hipMalloc((void **)&b->d_seed, (size_t)nstreams) == hipSuccess;
      if (ok) ok = opusgpu_nsq_batch_reset(b) == OPUS_OK;
      if (!ok) { opusgpu_nsq_batch_destroy(b); b = nullptr; err = OPUS_ALLOC_FAIL; }
   }
   if (error) *error = err;
   return b;
}
static int oa_nsq_tile_rw(OpusGpuNsqBatch *b, opus_int32 i, const void *in, void *out)
{
   if (!b || i < 0 || i >= b->n || (!in && !out)) return OPUS_BAD_ARG;
   HIPCHECK(hipSetDevice(b->device));
   HIPCHECK(hipStreamSynchronize(b->stream));
   std::vector<i32> img((size_t)b->tile_words);
   i32 *d = b->d_tiles + (size_t)(i / b->T) * b->tile_words;
   HIPCHECK(hipMemcpy(img.data(), d, sizeof(i32) * (size_t)b->tile_words, hipMemcpyDeviceToHost));
   if (out) oa_nsq_export(img.data(), b->T, i % b->T, (OaNsqRefState *)out, &b->cfg);
   if (in) { oa_nsq_import(img.data(), b->T, i % b->T, (const OaNsqRefState *)in, &b->cfg); HIPCHECK(hipMemcpy(d, img.data(), sizeof(i32) * (size_t)b->tile_words, hipMemcpyHostToDevice)); }
   return OPUS_OK;
}
/* NOTE: import re-bases the stream's history ring to row 0; the other streams of the tile are untouched only if they share that
 * base, which they do whenever every stream of a batch has run the same number of frames (the only way to run them). */
int opusgpu_nsq_batch_import_state(OpusGpuNsqBatch *b, opus_int32 i, const void *silk_nsq_state)
{
   if (!b || i < 0 || i >= b->n || !silk_nsq_state) return OPUS_BAD_ARG;
   /* bring the whole tile to base 0 first so that mixed bases never exist inside a tile */
   HIPCHECK(hipSetDevice(b->device));
   HIPCHECK(hipStreamSynchronize(b->stream));
   std::vector<i32> img((size_t)b->tile_words), img2((size_t)b->tile_words, 0);
   i32 *d = b->d_tiles + (size_t)(i / b->T) * b->tile_words;
   HIPCHECK(hipMemcpy(img.data(), d, sizeof(i32) * (size_t)b->tile_words, hipMemcpyDeviceToHost));
   OaNsqRefState r;
   for (int t = 0; t < b->T; t++) {
      if (t == i % b->T) { oa_nsq_import(img2.data(), b->T, t, (const OaNsqRefState *)silk_nsq_state, &b->cfg); continue; }
      oa_nsq_export(img.data(), b->T, t, &r, &b->cfg); oa_nsq_import(img2.data(), b->T, t, &r, &b->cfg);
   }
   HIPCHECK(hipMemcpy(d, img2.data(), sizeof(i32) * (size_t)b->tile_words, hipMemcpyHostToDevice));
   return OPUS_OK;
}
int opusgpu_nsq_batch_export_state(OpusGpuNsqBatch *b, opus_int32 i, void *silk_nsq_state) { return oa_nsq_tile_rw(b, i, nullptr, silk_nsq_state); }

int opusgpu_nsq_batch_run_dev(OpusGpuNsqBatch *b, const OpusGpuNsqFrame *d_frames, const opus_int16 *d_x16, opus_int8 *d_pulses, opus_int8 *d_seed_out, void *hip_stream)
{
   if (!b || !d_frames || !d_x16 || !d_pulses) return OPUS_BAD_ARG;
   HIPCHECK(hipSetDevice(b->device));
   hipStream_t s = hip_stream ? (hipStream_t)hip_stream : b->stream;
   /* the shaping orders the reference's complexity table uses (silk/control_codec.c:320-385) get compile-time-specialised tap loops */
#define OA_NSQ_LAUNCH(SS) do { \
      if (b->dd) hipLaunchKernelGGL(oa_silk_nsq_dd_kernel<SS>, dim3((unsigned)b->ntiles), dim3(64), 0, s, b->cfg, b->d_tiles, b->tile_words, (const OaNsqFrame *)d_frames, \
                                    (const i16 *)d_x16, (i8 *)d_pulses, (i8 *)(d_seed_out ? d_seed_out : b->d_seed), (int)b->n); \
      else hipLaunchKernelGGL(oa_silk_nsq_kernel<SS>, dim3((unsigned)b->ntiles), dim3(64), 0, s, b->cfg, b->d_tiles, b->tile_words, (const OaNsqFrame *)d_frames, \
                              (const i16 *)d_x16, (i8 *)d_pulses, (int)b->n); } while (0)
   switch (b->cfg.shapingLPCOrder) {
   case 12: OA_NSQ_LAUNCH(12); break;  case 14: OA_NSQ_LAUNCH(14); break;  case 16: OA_NSQ_LAUNCH(16); break;
   case 20: OA_NSQ_LAUNCH(20); break;  case 24: OA_NSQ_LAUNCH(24); break;  default: OA_NSQ_LAUNCH(0);
   }
#undef OA_NSQ_LAUNCH
   HIPCHECK(hipGetLastError());
   return OPUS_OK;
}
int opusgpu_nsq_batch_sync(OpusGpuNsqBatch *b) { if (!b) return OPUS_BAD_ARG; HIPCHECK(hipSetDevice(b->device)); HIPCHECK(hipStreamSynchronize(b->stream)); return OPUS_OK; }
/* host-buffer convenience: frames[n], x16[n][frame_length] -> pulses[n][frame_length]; seed_out[n] (may be NULL) receives the Seed index the
 * delayed-decision winner started from (psIndices->Seed, NSQ_del_dec.c:286); for the plain quantiser it echoes frames[i].Seed */
int opusgpu_nsq_batch_run(OpusGpuNsqBatch *b, const OpusGpuNsqFrame *frames, const opus_int16 *x16, opus_int8 *pulses, opus_int8 *seed_out)
{
   if (!b || !frames || !x16 || !pulses) return OPUS_BAD_ARG;
   HIPCHECK(hipSetDevice(b->device));
   const size_t frame = (size_t)b->cfg.nb_subfr * 5 * b->cfg.fs_kHz, n = (size_t)b->n;
   HIPCHECK(hipMemcpyAsync(b->d_frames, frames, sizeof(OaNsqFrame) * n, hipMemcpyHostToDevice, b->stream));
   HIPCHECK(hipMemcpyAsync(b->d_x16, x16, sizeof(opus_int16) * frame * n, hipMemcpyHostToDevice, b->stream));
   int r = opusgpu_nsq_batch_run_dev(b, (const OpusGpuNsqFrame *)b->d_frames, b->d_x16, b->d_pulses, b->d_seed, nullptr);
   if (r != OPUS_OK) return r;
   HIPCHECK(hipMemcpyAsync(pulses, b->d_pulses, frame * n, hipMemcpyDeviceToHost, b->stream));
   if (seed_out && b->dd) HIPCHECK(hipMemcpyAsync(seed_out, b->d_seed, n, hipMemcpyDeviceToHost, b->stream));
   HIPCHECK(hipStreamSynchronize(b->stream));
   if (seed_out && !b->dd) for (size_t i = 0; i < n; i++) seed_out[i] = frames[i].Seed;
   return OPUS_OK;
}
/* average launch duration of `steps` back-to-back frames on the batch's stream (HIP events), inputs already resident */
int opusgpu_nsq_time_dev(OpusGpuNsqBatch *b, const OpusGpuNsqFrame *d_frames, const opus_int16 *d_x16, opus_int8 *d_pulses, int steps, float *ms)
{
   if (!b || !ms || steps <= 0) return OPUS_BAD_ARG;
   HIPCHECK(hipSetDevice(b->device));
   hipEvent_t e0, e1;
   HIPCHECK(hipEventCreate(&e0)); HIPCHECK(hipEventCreate(&e1));
   HIPCHECK(hipEventRecord(e0, b->stream));
   for (int k = 0; k < steps; k++) { int r = opusgpu_nsq_batch_run_dev(b, d_frames, d_x16, d_pulses, nullptr, nullptr); if (r != OPUS_OK) return r; }
   HIPCHECK(hipEventRecord(e1, b->stream));
   HIPCHECK(hipEventSynchronize(e1));
   HIPCHECK(hipEventElapsedTime(ms, e0, e1));
   (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
   return OPUS_OK;
}
} /* extern "C" */
#endif

/* wave.h — the CDNA4 execution vocabulary used by the codec kernels.
 *
 * One wavefront (64 lanes) encodes one (stream, frame).  Workgroups are exactly one wave
 * (__launch_bounds__(64)), so a workgroup barrier is free: the compiler drops s_barrier and keeps only
 * the LDS ordering (s_waitcnt lgkmcnt(0)).  Lane-parallel phases alternate with lane-0 serial phases
 * (range coder, bit allocation); cross-lane traffic goes through DPP/ds_bpermute shuffles, never
 * through global memory.  All working arrays live in LDS and are addressed through address_space(3)
 * pointers so every access is a ds_* instruction, not a flat_* one. */
#ifndef OPUS_AMD_WAVE_H
#define OPUS_AMD_WAVE_H
#include <hip/hip_runtime.h>
#include <stdint.h>

#define WV_DEV  __device__ __forceinline__
#ifndef WV_DEVN
#define WV_DEVN __device__ __noinline__
#endif
#define WV_MEM  __device__ __forceinline__          /* member functions */
#define WV_HD   __host__ __device__ inline          /* small pure helpers shared with host code */
#define WV_LDS  __attribute__((address_space(3)))
#define WV_TABLE __device__ const
#define WV_WIDTH 64

WV_DEV int wv_lane() { return (int)threadIdx.x; }
WV_DEV float wv_rcpf(float x) { return __builtin_amdgcn_rcpf(x); }           /* v_rcp_f32: 1 ulp */
/* Ordering of memory traffic between the lanes of the wave (block == wave).  A wavefront executes its LDS and its vector-memory instructions in issue order
 * and all its lanes share one L1, so a later access by any lane observes an earlier write by any other lane without waiting for anything: a wavefront-scope
 * fence (wv_order: a compiler barrier, no s_waitcnt) is all the ordering a one-wave workgroup needs, for LDS and for the per-wave HBM scratch alike.
 * wv_sync is the workgroup-scope __syncthreads(), which also drains every outstanding LDS and global operation at each lane-0 section boundary.
 * Measured on the MI355X at 16 waves per CU (profiles/r02_e): 1,265,577 frames/s with wavefront-scope ordering everywhere (-DOA_LIGHT_SYNC, the whole
 * GPU parity suite green) against 1,268,870 with the drains -- the other waves hide them -- so the conservative form stays the default. */
WV_DEV void wv_order() { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront"); }
#ifdef OA_LIGHT_SYNC
WV_DEV void wv_sync() { wv_order(); }
#else
WV_DEV void wv_sync() { __syncthreads(); }
#endif

/* the wave's issue priority (s_setprio 0..3): a section in which one lane walks a dependent chain raises it, so that the chain's next instruction does not queue
 * behind the other waves' parallel work (which fills the slots the chain leaves anyway) */
#ifdef OA_SERIAL_PRIO
WV_DEV void wv_prio_serial() { __builtin_amdgcn_s_setprio(OA_SERIAL_PRIO); }
WV_DEV void wv_prio_normal() { __builtin_amdgcn_s_setprio(0); }
#else
WV_DEV void wv_prio_serial() {}
WV_DEV void wv_prio_normal() {}
#endif

WV_DEV int32_t wv_shfl(int32_t v, int src) { return __shfl(v, src, 64); }
/* src must be wave-uniform */
WV_DEV int32_t wv_bcast(int32_t v, int src) { return __builtin_amdgcn_readlane(v, __builtin_amdgcn_readfirstlane(src)); }
/* v is the same in every lane: move it to a scalar register so that control flow and address math derived from it run
 * on the scalar unit (arguments of non-inlined device functions arrive in VGPRs and would otherwise stay there) */
WV_DEV int32_t wv_uni(int32_t v) { return __builtin_amdgcn_readfirstlane(v); }
/* old with lane `lane` replaced by the (uniform) value val */
WV_DEV int32_t wv_writelane(int32_t val, int lane, int32_t old)
{
   /* no clang builtin for llvm.amdgcn.writelane in this toolchain; a VOP3 may read one SGPR only, so the lane select goes through M0.  M0 cannot be named
    * in a clobber list (reserved register), so the sequence saves and restores it itself */
   int32_t m0_save;
   asm("s_mov_b32 %1, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tv_writelane_b32 %0, %2, m0\n\ts_mov_b32 m0, %1" : "+v"(old), "=&s"(m0_save) : "s"(val), "s"(lane));
   return old;
}

/* lane i receives v of lane i-1, lane 0 receives `fill` (DPP wave_shr:1: one instruction, no LDS) */
/* value of lane J (0..3) of the caller's quad, in all four lanes: one DPP quad_perm move, no LDS crossbar */
/* value of lane Q as a wave-uniform scalar (v_readlane_b32 with a constant lane select: the result lives in an SGPR) */
template <int Q> WV_DEV int32_t wv_lane_const(int32_t v) { return __builtin_amdgcn_readlane(v, Q); }
template <int J> WV_DEV int32_t wv_quad_bcast(int32_t v) { return __builtin_amdgcn_update_dpp(0, v, J * 0x55, 0xf, 0xf, false); }
WV_DEV int32_t wv_shift_up1(int32_t v, int32_t fill) { return __builtin_amdgcn_update_dpp(fill, v, 0x138, 0xf, 0xf, false); }
/* lane i receives v of lane i+1, lane 63 receives `fill` (DPP wave_shl:1) */
WV_DEV int32_t wv_shift_down1(int32_t v, int32_t fill) { return __builtin_amdgcn_update_dpp(fill, v, 0x130, 0xf, 0xf, false); }

/* Wave-wide reductions on the DPP cross-lane network (no LDS round trips, unlike ds_bpermute shuffles):
 * quad_perm swaps -> row rotations (every lane of a 16-lane row holds the row result) -> row_bcast:15 / row_bcast:31
 * carry the partial results up the rows; lane 63 ends with the wave result, v_readlane broadcasts it. */
#define WV_DPP(old, src, ctrl, rmask) __builtin_amdgcn_update_dpp((old), (src), (ctrl), (rmask), 0xf, false)
#define WV_DPP_QP_1032 0xB1
#define WV_DPP_QP_2301 0x4E
#define WV_DPP_ROW_ROR4 0x124
#define WV_DPP_ROW_ROR8 0x128
#define WV_DPP_BCAST15 0x142
#define WV_DPP_BCAST31 0x143
WV_DEV int32_t wv_sum(int32_t v)
{
   v += WV_DPP(0, v, WV_DPP_QP_1032, 0xf);
   v += WV_DPP(0, v, WV_DPP_QP_2301, 0xf);
   v += WV_DPP(0, v, WV_DPP_ROW_ROR4, 0xf);
   v += WV_DPP(0, v, WV_DPP_ROW_ROR8, 0xf);
   v += WV_DPP(0, v, WV_DPP_BCAST15, 0xa);
   v += WV_DPP(0, v, WV_DPP_BCAST31, 0xc);
   return __builtin_amdgcn_readlane(v, 63);
}
WV_DEV uint32_t wv_sumu(uint32_t v) { return (uint32_t)wv_sum((int32_t)v); }
WV_DEV int64_t wv_sum64(int64_t v)
{
   /* two independent 32-bit lanes of work; carries are rebuilt exactly by summing 16-bit quarters */
   uint32_t lo = (uint32_t)v, hi = (uint32_t)((uint64_t)v >> 32);
   uint32_t a = wv_sumu(lo & 0xffff), b = wv_sumu(lo >> 16), c = wv_sumu(hi & 0xffff), d = wv_sumu(hi >> 16);
   uint64_t r = (uint64_t)a + ((uint64_t)b << 16) + ((uint64_t)c << 32) + ((uint64_t)d << 48);
   return (int64_t)r;
}
WV_DEV int32_t wv_max(int32_t v)
{
   int32_t t;
   t = WV_DPP(v, v, WV_DPP_QP_1032, 0xf); v = t > v ? t : v;
   t = WV_DPP(v, v, WV_DPP_QP_2301, 0xf); v = t > v ? t : v;
   t = WV_DPP(v, v, WV_DPP_ROW_ROR4, 0xf); v = t > v ? t : v;
   t = WV_DPP(v, v, WV_DPP_ROW_ROR8, 0xf); v = t > v ? t : v;
   t = WV_DPP(v, v, WV_DPP_BCAST15, 0xa); v = t > v ? t : v;
   t = WV_DPP(v, v, WV_DPP_BCAST31, 0xc); v = t > v ? t : v;
   return __builtin_amdgcn_readlane(v, 63);
}
WV_DEV int32_t wv_min(int32_t v) { return -wv_max(-v); }   /* callers never pass INT32_MIN */
WV_DEV uint32_t wv_or(uint32_t v)
{
   v |= (uint32_t)WV_DPP(0, (int)v, WV_DPP_QP_1032, 0xf);
   v |= (uint32_t)WV_DPP(0, (int)v, WV_DPP_QP_2301, 0xf);
   v |= (uint32_t)WV_DPP(0, (int)v, WV_DPP_ROW_ROR4, 0xf);
   v |= (uint32_t)WV_DPP(0, (int)v, WV_DPP_ROW_ROR8, 0xf);
   v |= (uint32_t)WV_DPP(0, (int)v, WV_DPP_BCAST15, 0xa);
   v |= (uint32_t)WV_DPP(0, (int)v, WV_DPP_BCAST31, 0xc);
   return (uint32_t)__builtin_amdgcn_readlane((int)v, 63);
}
WV_DEV uint64_t wv_ballot(int pred) { return __ballot(pred); }
/* inclusive prefix sum over lanes */
WV_DEV int32_t wv_scan_incl(int32_t v)
{
   /* on the DPP network like the reductions above: shifts by 1, 2, 4, 8 inside each row of 16 lanes (lanes without a source add 0), then the totals of the rows
    * below are broadcast up (row_bcast:15 into rows 1 and 3, row_bcast:31 into rows 2 and 3) -- six adds, no LDS crossbar */
   v += WV_DPP(0, v, 0x111, 0xf);
   v += WV_DPP(0, v, 0x112, 0xf);
   v += WV_DPP(0, v, 0x114, 0xf);
   v += WV_DPP(0, v, 0x118, 0xf);
   v += WV_DPP(0, v, WV_DPP_BCAST15, 0xa);
   v += WV_DPP(0, v, WV_DPP_BCAST31, 0xc);
   return v;
}
/* PVQ greedy-search arg-max: maximise num/den (den > 0, exact 16x16 cross products), lowest index wins ties.
 * Every lane receives the winning (num, den, idx).  The order is total, so the DPP tree may combine in any shape. */
#define WV_ARGMAX_STEP(ctrl, rmask) do { \
      int32_t n2 = WV_DPP(num, num, ctrl, rmask), d2 = WV_DPP(den, den, ctrl, rmask), i2 = WV_DPP(idx, idx, ctrl, rmask); \
      int32_t lhs = (int32_t)(int16_t)den * (int32_t)(int16_t)n2, rhs = (int32_t)(int16_t)d2 * (int32_t)(int16_t)num; \
      bool take = lhs > rhs || (lhs == rhs && i2 < idx); \
      num = take ? n2 : num; den = take ? d2 : den; idx = take ? i2 : idx; } while (0)
WV_DEV void wv_argmax_ratio(int32_t &num, int32_t &den, int32_t &idx)
{
   WV_ARGMAX_STEP(WV_DPP_QP_1032, 0xf);
   WV_ARGMAX_STEP(WV_DPP_QP_2301, 0xf);
   WV_ARGMAX_STEP(WV_DPP_ROW_ROR4, 0xf);
   WV_ARGMAX_STEP(WV_DPP_ROW_ROR8, 0xf);
   WV_ARGMAX_STEP(WV_DPP_BCAST15, 0xa);
   WV_ARGMAX_STEP(WV_DPP_BCAST31, 0xc);
   num = __builtin_amdgcn_readlane(num, 63); den = __builtin_amdgcn_readlane(den, 63); idx = __builtin_amdgcn_readlane(idx, 63);
}
/* Same arg-max with (num, den) packed in one register (den in the high half, both 15-bit unsigned) and WITHOUT carrying the index:
 * the DPP tree finds a maximal ratio, every lane then tests itself against it by exact cross-multiplication and the lowest
 * matching lane (s_ff1 of the ballot) is the winner -- half the cross-lane traffic and no index tie-break per stage.
 * `valid` lanes only may win; invalid lanes must pass num = 0, den = 1.  nlanes (uniform) bounds the occupied lanes so that short
 * bands skip the cross-row stages.  Returns the winning lane. */
#define WV_ARGMAXP_STEP(ctrl, rmask) do { \
      uint32_t o = (uint32_t)WV_DPP((int)pk, (int)pk, ctrl, rmask); \
      uint32_t lhs = (pk >> 16) * (o & 0xffffu), rhs = (o >> 16) * (pk & 0xffffu); \
      pk = lhs > rhs ? o : pk; } while (0)
WV_DEV int wv_argmax_ratio_packed(uint32_t num, uint32_t den, bool valid, int nlanes)
{
   uint32_t pk = den << 16 | num;
   const uint32_t mine = pk;
   WV_ARGMAXP_STEP(WV_DPP_QP_1032, 0xf);
   WV_ARGMAXP_STEP(WV_DPP_QP_2301, 0xf);
   WV_ARGMAXP_STEP(WV_DPP_ROW_ROR4, 0xf);
   WV_ARGMAXP_STEP(WV_DPP_ROW_ROR8, 0xf);
   uint32_t best;
   if (nlanes <= 16) best = (uint32_t)__builtin_amdgcn_readlane((int)pk, 0);
   else {
      WV_ARGMAXP_STEP(WV_DPP_BCAST15, 0xa);
      if (nlanes <= 32) best = (uint32_t)__builtin_amdgcn_readlane((int)pk, 31);
      else { WV_ARGMAXP_STEP(WV_DPP_BCAST31, 0xc); best = (uint32_t)__builtin_amdgcn_readlane((int)pk, 63); }
   }
   const bool hit = valid && (mine >> 16) * (best & 0xffffu) == (best >> 16) * (mine & 0xffffu);
   return (int)__builtin_ctzll(__ballot(hit));
}
/* ---- 16-lane groups: four independent streams share one wave (celt_enc_pvq4.h).  A group is one DPP row, so its reductions are the row part of the wave trees above
 * (every lane of the row ends with the row's result: no v_readlane), and groups may sit in different branches: the collectives below only touch lanes of the caller's own
 * row, which a group enters and leaves together.  Values that are "uniform" per stream live in VGPRs here (one copy per lane of the row). ---- */
#define WG_WIDTH 16
WV_DEV int wg_lane() { return (int)threadIdx.x & 15; }
WV_DEV int wg_id() { return (int)threadIdx.x >> 4; }
WV_DEV void wg_sync() { wv_order(); }          /* one wave: the LDS pipeline is in order, a compiler fence is all the ordering there is to ask for */
WV_DEV bool wv_any(int pred) { return __ballot(pred) != 0; }      /* WAVE-level vote (every lane must call it) */
WV_DEV int32_t wg_sum(int32_t v)
{
   v += WV_DPP(0, v, WV_DPP_QP_1032, 0xf);
   v += WV_DPP(0, v, WV_DPP_QP_2301, 0xf);
   v += WV_DPP(0, v, WV_DPP_ROW_ROR4, 0xf);
   v += WV_DPP(0, v, WV_DPP_ROW_ROR8, 0xf);
   return v;
}
WV_DEV uint32_t wg_sumu(uint32_t v) { return (uint32_t)wg_sum((int32_t)v); }
WV_DEV int64_t wg_sum64(int64_t v)
{
   uint32_t lo = (uint32_t)v, hi = (uint32_t)((uint64_t)v >> 32);
   uint32_t a = wg_sumu(lo & 0xffff), b = wg_sumu(lo >> 16), c = wg_sumu(hi & 0xffff), d = wg_sumu(hi >> 16);
   return (int64_t)((uint64_t)a + ((uint64_t)b << 16) + ((uint64_t)c << 32) + ((uint64_t)d << 48));
}
WV_DEV int32_t wg_max(int32_t v)
{
   int32_t t;
   t = WV_DPP(v, v, WV_DPP_QP_1032, 0xf); v = t > v ? t : v;
   t = WV_DPP(v, v, WV_DPP_QP_2301, 0xf); v = t > v ? t : v;
   t = WV_DPP(v, v, WV_DPP_ROW_ROR4, 0xf); v = t > v ? t : v;
   t = WV_DPP(v, v, WV_DPP_ROW_ROR8, 0xf); v = t > v ? t : v;
   return v;
}
WV_DEV uint32_t wg_or(uint32_t v)
{
   v |= (uint32_t)WV_DPP(0, (int)v, WV_DPP_QP_1032, 0xf);
   v |= (uint32_t)WV_DPP(0, (int)v, WV_DPP_QP_2301, 0xf);
   v |= (uint32_t)WV_DPP(0, (int)v, WV_DPP_ROW_ROR4, 0xf);
   v |= (uint32_t)WV_DPP(0, (int)v, WV_DPP_ROW_ROR8, 0xf);
   return v;
}
/* inclusive prefix sum inside the row (row_shr:1/2/4/8, lanes without a source add 0) */
WV_DEV int32_t wg_scan_incl(int32_t v)
{
   v += WV_DPP(0, v, 0x111, 0xf);
   v += WV_DPP(0, v, 0x112, 0xf);
   v += WV_DPP(0, v, 0x114, 0xf);
   v += WV_DPP(0, v, 0x118, 0xf);
   return v;
}
/* lane i of the group receives v of lane i + D (wg_shl) / i - D (wg_shr) of the same group, 0 where there is none: DPP row_shl / row_shr, one instruction */
template <int D> WV_DEV int32_t wg_shl(int32_t v) { return WV_DPP(0, v, 0x100 + D, 0xf); }
template <int D> WV_DEV int32_t wg_shr(int32_t v) { return WV_DPP(0, v, 0x110 + D, 0xf); }
/* value of lane src (0..15, the same in all lanes of the group, free to differ between groups) of the caller's group: ds_bpermute_b32, the LDS crossbar without memory */
WV_DEV int32_t wg_bcast(int32_t v, int src) { return __builtin_amdgcn_ds_bpermute((int)(((threadIdx.x & 48u) | (unsigned)src) << 2), v); }
WV_DEV uint32_t wg_ballot(int pred) { return (uint32_t)(__ballot(pred) >> (threadIdx.x & 48u)) & 0xffffu; }
/* the packed-ratio arg-max of wv_argmax_ratio_packed inside a group: returns the winning lane (0..15) */
WV_DEV int wg_argmax_ratio_packed(uint32_t num, uint32_t den, bool valid)
{
   uint32_t pk = den << 16 | num;
   const uint32_t mine = pk;
   WV_ARGMAXP_STEP(WV_DPP_QP_1032, 0xf);
   WV_ARGMAXP_STEP(WV_DPP_QP_2301, 0xf);
   WV_ARGMAXP_STEP(WV_DPP_ROW_ROR4, 0xf);
   WV_ARGMAXP_STEP(WV_DPP_ROW_ROR8, 0xf);
   /* the row tree is not a total order on ties (equal ratios with different (num, den) may leave different representatives in different lanes), but every lane holds A
    * maximal ratio, and the hit test is by cross-multiplication: the set of hits is the set of maximal lanes whichever representative a lane holds */
   const bool hit = valid && (mine >> 16) * (pk & 0xffffu) == (pk >> 16) * (mine & 0xffffu);
   return (int)__builtin_ctz(wg_ballot(hit) | 0x10000u);
}
#endif

/* celt_enc_pvq4.h — the PVQ stage (quant_all_bands) with FOUR streams per wavefront: one 16-lane group (one DPP row) per stream.
 * Reference: the same lines celt_enc_pvq.h follows -- celt/bands.c :638 compute_qn, :700 compute_theta, :973 quant_partition, :1248 quant_band, :1387 quant_band_stereo,
 * :1589 quant_all_bands (the theta RDO :1842-1912), celt/vq.c :75 exp_rotation1, :104 exp_rotation, :205 op_pvq_search_c, :552 alg_quant, :695 renormalise_vector,
 * :724 stereo_itheta, celt/cwrs.c:444 encode_pulses, celt/rate.h:48-66.
 *
 * Why: with one wave per stream the PVQ works on partitions of 4-12 coefficients -- a quarter of the wave at best -- and everything the band recursion carries is one
 * scalar per stream; a wave instruction is spent per scalar step.  Here the same instruction serves four streams: what was wave-uniform (SGPRs, scalar branches) is
 * group-uniform (one VGPR copy per lane of the row, EXEC-masked branches), the range coder runs on lanes 0 / 16 / 32 / 48 at once, reductions are the row part of the DPP
 * trees.  The recursion of quant_partition is an explicit stack (P4Frame) walked by a loop whose trips are aligned across the groups: every trip takes every group down
 * through its split nodes (compute_theta) to its next leaf (alg_quant), codes the leaves side by side, and unwinds.  No function of the stage is out of line, nothing
 * is spilled across calls, and the stage's LDS is what a band needs (the band, its second channel, the folding source: 2.9 KB per stream): three waves per SIMD.
 * The bands themselves go in lockstep (band sizes only depend on the frame size, which a launch shares), so the pulse-cache rows of a band are staged once per wave.
 * What is NOT here: frames shorter than 10 ms (bands of one or two coefficients: quant_band_n1, the N == 2 stereo case) -- the front kernel keeps those. */
#ifndef OPUS_AMD_CELT_ENC_PVQ4_H
#define OPUS_AMD_CELT_ENC_PVQ4_H

#define FOR_GL(i, n) for (int i = wg_lane(); i < (n); i += WG_WIDTH)
#define GLANE0 for (int l0_ = (wg_sync(), 1); l0_; l0_ = (wg_sync(), 0)) if (wg_lane() == 0)
#ifndef P4_TIC
#define P4_TIC()
#define P4_TOC(bucket)
#endif

struct P4Frame { i32 xo, N, B, B0, LM, lb, gm, gs, fill, mbits, sbits, itheta, rebal, mid_first, phase, cm; };
struct P4Group {
   i32 Xb[OA_MAX_BAND], Yb[OA_MAX_BAND];       /* the band being coded: channel 0 / mid, channel 1 / side (contiguous: a node's X is an offset from Xb) */
   i32 lbs[OA_MAX_BAND];                       /* this band's folding source, private copy (transformed in place) */
   EcCtx ec, ecsave[2];
   i32 pulses[NBE], tf_res[NBE], bandE[2 * NBE];
   u8 cmask[2 * NBE + 2];
   P4Frame stk[4];                             /* 720 words in all = 16 mod 64: the four groups' j-th words fall into four different quarters of the 64 LDS banks */
};
struct P4Lds { P4Group g[4]; u8 rows[5 * 64]; };
static_assert(sizeof(P4Group) == 720 * 4, "P4Group: 720 words");

WV_TABLE u8 k_bit_interleave_table[16] = {0, 1, 1, 1, 2, 3, 3, 3, 2, 3, 3, 3, 2, 3, 3, 3};
WV_TABLE u8 k_bit_deinterleave_table[16] = {0x00, 0x03, 0x0C, 0x0F, 0x30, 0x33, 0x3C, 0x3F, 0xC0, 0xC3, 0xCC, 0xCF, 0xF0, 0xF3, 0xFC, 0xFF};
WV_TABLE i16 k_exp2_table8[8] = {16384, 17866, 19483, 21247, 23170, 25267, 27554, 30048};

/* ---- the range coder of a group: state parked in the group's LDS, bytes straight into the stream's packet in HBM (append-only during the PVQ) ---- */
#define P4_EC_BEGIN EcCtx ec_; ec_ld(&ec_, &G->ec); EcCtx *e = &ec_; u8 *buf = ecbuf
#define P4_EC_END ec_st(&G->ec, &ec_)

WV_DEV int p4_compute_qn(int N, int b, int offset, int pulse_cap, int stereo)
{
   int qn, qb, N2 = 2 * N - 1;
   if (stereo && N == 2) N2--;
   qb = fx_sdiv24(b + N2 * offset, N2);
   qb = imin(b - pulse_cap - (4 << BITRES), qb);
   qb = imin(8 << BITRES, qb);
   if (qb < (1 << BITRES >> 1)) qn = 1;
   else {
      qn = k_exp2_table8[qb & 0x7] >> (14 - (qb >> BITRES));
      qn = (qn + 1) >> 1 << 1;
   }
   return qn;
}

WV_DEV i32 p4_inner_prod(const WV_LDS i32 *x, const WV_LDS i32 *y, int len)
{
   i64 sum = 0;
   FOR_GL(i, len) sum += x[i] * (i64)y[i];
   return (i32)(wg_sum64(sum) >> 2 * (NORM_SHIFT - 14));
}
WV_DEV i32 p4_inner_prod_g(const i32 *x, const WV_LDS i32 *y, int len)            /* x in HBM */
{
   i64 sum = 0;
   FOR_GL(i, len) sum += x[i] * (i64)y[i];
   return (i32)(wg_sum64(sum) >> 2 * (NORM_SHIFT - 14));
}

/* haar1 (bands.c:623); stride is a power of two */
WV_DEV void p4_haar1(WV_LDS i32 *X, int N0, int stride)
{
   N0 >>= 1;
   const int ls = ec_ilog((u32)stride) - 1;
   wg_sync();
   FOR_GL(p, N0 * stride) {
      const int i = p & (stride - 1), j = p >> ls;
      i32 t1 = mult32_32_q31(QC32(.70710678f, 31), X[stride * 2 * j + i]);
      i32 t2 = mult32_32_q31(QC32(.70710678f, 31), X[stride * (2 * j + 1) + i]);
      X[stride * 2 * j + i] = add32(t1, t2);
      X[stride * (2 * j + 1) + i] = sub32(t1, t2);
   }
   wg_sync();
}
/* (de)interleave_hadamard (bands.c:574/:600): a permutation of N0 * stride <= 176 words through registers; nmax: a wave-uniform bound of N0 * stride */
#define P4_MAXR ((OA_MAX_BAND + WG_WIDTH - 1) / WG_WIDTH)
WV_DEV void p4_deinterleave_hadamard(WV_LDS i32 *X, int N0, int stride, int hadamard, int nmax)
{
   const int N = N0 * stride, ls = ec_ilog((u32)stride) - 1;
   i32 v[P4_MAXR];
   wg_sync();
#pragma unroll
   for (int t = 0; t < P4_MAXR; t++) { const int p = wg_lane() + t * WG_WIDTH; if (t * WG_WIDTH < nmax && p < N) v[t] = X[p]; }
   wg_sync();
#pragma unroll
   for (int t = 0; t < P4_MAXR; t++) {
      const int p = wg_lane() + t * WG_WIDTH;                 /* p = j * stride + i */
      if (t * WG_WIDTH < nmax && p < N) { const int i = p & (stride - 1), j = p >> ls; X[(hadamard ? k_ordery_table[stride - 2 + i] : i) * N0 + j] = v[t]; }
   }
   wg_sync();
}
WV_DEV void p4_interleave_hadamard(WV_LDS i32 *X, int N0, int stride, int hadamard, int nmax)
{
   const int N = N0 * stride, ls = ec_ilog((u32)stride) - 1;
   i32 v[P4_MAXR];
   wg_sync();
#pragma unroll
   for (int t = 0; t < P4_MAXR; t++) {
      const int p = wg_lane() + t * WG_WIDTH;
      if (t * WG_WIDTH < nmax && p < N) { const int i = p & (stride - 1), j = p >> ls; v[t] = X[(hadamard ? k_ordery_table[stride - 2 + i] : i) * N0 + j]; }
   }
   wg_sync();
#pragma unroll
   for (int t = 0; t < P4_MAXR; t++) { const int p = wg_lane() + t * WG_WIDTH; if (t * WG_WIDTH < nmax && p < N) X[p] = v[t]; }
   wg_sync();
}

WV_DEV void p4_intensity_stereo(const WV_LDS i32 *bandE, WV_LDS i32 *X, const WV_LDS i32 *Y, int i, int N)
{
   int shift = celt_zlog2(imax(bandE[i], bandE[i + NBE])) - 13;
   i16 left = (i16)vshr32(bandE[i], shift), right = (i16)vshr32(bandE[i + NBE], shift);
   i16 norm = (i16)(EPSILON + fx_sqrt(EPSILON + mult16_16(left, left) + mult16_16(right, right)));
   left = (i16)imin(left, norm - 1);
   right = (i16)imin(right, norm - 1);
   i16 a1 = (i16)(shl32((i32)left, 15) / norm), a2 = (i16)(shl32((i32)right, 15) / norm);
   wg_sync();
   FOR_GL(j, N) X[j] = add32(mult16_32_q15(a1, X[j]), mult16_32_q15(a2, Y[j]));
   wg_sync();
}
WV_DEV void p4_stereo_merge(WV_LDS i32 *X, WV_LDS i32 *Y, i32 mid, int N)
{
   wg_sync();
   i32 xp = p4_inner_prod(Y, X, N), side = p4_inner_prod(Y, Y, N);
   xp = mult32_32_q31(mid, xp);
   i32 El = (mult32_32_q31(mid, mid) >> 3) + side - 2 * xp;
   i32 Er = (mult32_32_q31(mid, mid) >> 3) + side + 2 * xp;
   if (Er < QC32(6e-4f, 28) || El < QC32(6e-4f, 28)) { FOR_GL(j, N) Y[j] = X[j]; wg_sync(); return; }
   int kl = celt_ilog2(El) >> 1, kr = celt_ilog2(Er) >> 1;
   i32 t = vshr32(El, (kl << 1) - 29);
   i32 lgain = fx_rsqrt_norm32(t);
   t = vshr32(Er, (kr << 1) - 29);
   i32 rgain = fx_rsqrt_norm32(t);
   if (kl < 7) kl = 7;
   if (kr < 7) kr = 7;
   FOR_GL(j, N) {
      i32 l = mult32_32_q31(mid, X[j]), r = Y[j];
      X[j] = vshr32(mult32_32_q31(lgain, sub32(l, r)), kl - 15);
      Y[j] = vshr32(mult32_32_q31(rgain, add32(l, r)), kr - 15);
   }
   wg_sync();
}
WV_DEV i32 p4_stereo_itheta(const WV_LDS i32 *X, const WV_LDS i32 *Y, int stereo, int N)
{
   i32 Emid = 0, Eside = 0;
   wg_sync();
   if (stereo) {
      FOR_GL(i, N) {
         i32 m = pshr32(add32(X[i], Y[i]), NORM_SHIFT - 13);
         i32 s = pshr32(sub32(X[i], Y[i]), NORM_SHIFT - 13);
         Emid = mac16_16(Emid, m, m);
         Eside = mac16_16(Eside, s, s);
      }
      Emid = wg_sum(Emid); Eside = wg_sum(Eside);
   } else {
      Emid = p4_inner_prod(X, X, N);
      Eside = p4_inner_prod(Y, Y, N);
   }
   i32 mid = fx_sqrt32(Emid), side = fx_sqrt32(Eside);
   return fx_atan2p_norm(side, mid);
}
WV_DEV void p4_renormalise_vector(WV_LDS i32 *X, int N, i32 gain)
{
   i32 e = 0;
   wg_sync();
   FOR_GL(i, N) { i32 v = pshr32(X[i], NORM_SHIFT - 14); e = add32(e, (i32)((u32)v * (u32)v)); }
   i32 E = add32(EPSILON, wg_sum(e));
   int k = celt_ilog2(E) >> 1;
   i32 t = vshr32(E, 2 * (k - 7));
   i16 g = (i16)mult32_32_q31(fx_rsqrt_norm(t), gain);
   FOR_GL(i, N) { i32 v = pshr32(X[i], NORM_SHIFT - 14); X[i] = shl32((i32)extract16(pshr32(mult16_16(g, v), k + 15 - 14)), NORM_SHIFT - 14); }
   wg_sync();
}

/* ---- exp_rotation (vq.c:104) in place on the band in LDS (already scaled down to Q14): one lane per independent chain (block, residue mod d) ---- */
WV_DEV void p4_rot_pass(WV_LDS i32 *T, int nblk, int len, int d, i32 c_, i32 s_)
{
   /* (fetching eight steps' operands ahead of the chain -- what a sweep reads ahead of itself is untouched data -- was tried: 16 more live registers at the 168 the kernel
    * is held to, more scratch, config 2 1.86 -> 1.79 M frames/s, profiles/r06_d; the kernel is VALU-issue-bound, not LDS-latency-bound: profiles/r06_e) */
   const i32 c = (i16)c_, s = (i16)s_;
   const int top = len - 2 * d - 1;
   wg_sync();
   for (int ch = wg_lane(); ch < nblk * d; ch += WG_WIDTH) {
      const int blk = (int)fx_udiv24((u32)ch, (u32)d), r = ch - blk * d;
      WV_LDS i32 *X = T + blk * len;
      if (r < len - d) {
         i32 x1 = X[r];
         int i = r;
         for (; i < len - d; i += d) {
            const i32 x2 = X[i + d];
            X[i] = (i32)(i16)(add32(sub32(mult16_16(c, x1), s * x2), 16384) >> 15);
            x1 = (i32)(i16)(add32(add32(mult16_16(c, x2), s * x1), 16384) >> 15);
         }
         X[i] = x1;
      }
      if (top >= r) {
         int i = top - (int)((u32)(top - r) % (u32)d);
         i32 y = X[i + d];
         for (; i >= 0; i -= d) {
            const i32 x1 = X[i];
            X[i + d] = (i32)(i16)(add32(add32(mult16_16(c, y), s * x1), 16384) >> 15);
            y = (i32)(i16)(add32(sub32(mult16_16(c, x1), s * y), 16384) >> 15);
         }
         X[i + d] = y;
      }
   }
   wg_sync();
}
/* returns 1 when the rotation applies (the caller scales the band down to Q14 before and up after: the up / down pairs between passes cancel exactly) */
WV_DEV int p4_rot_applies(int len, int K, int spread) { return !(2 * K >= len || spread == 0); }
WV_DEV void p4_exp_rotation_q14(WV_LDS i32 *T, int len, int dir, int stride, int K, int spread)
{
   int stride2 = 0;
   int factor = spread == 1 ? 15 : (spread == 2 ? 10 : 5);
   i16 gain = (i16)fx_div(mult16_16(Q15ONE, len), (i32)(len + factor * K));
   i16 theta = (i16)(mult16_16_q15(gain, gain) >> 1);
   i32 c = fx_cos_norm(theta);
   i32 s = fx_cos_norm(sub16(Q15ONE, theta));
   if (len >= 8 * stride) {
      stride2 = 1;
      while ((stride2 * stride2 + stride2) * stride + (stride >> 2) < len) stride2++;
   }
   len = fx_div_pow2(len, stride);
   if (dir < 0) {
      if (stride2) p4_rot_pass(T, stride, len, stride2, s, c);
      p4_rot_pass(T, stride, len, 1, c, s);
   } else {
      p4_rot_pass(T, stride, len, 1, c, -s);
      if (stride2) p4_rot_pass(T, stride, len, stride2, s, -c);
   }
}

/* The same rotation on a leaf of at most 16 coefficients held one per lane (v; pos = the lane's place in its block, out of range for lanes beyond the leaf).  Only the chain
 * itself is serial: x1' = (A + s x1) >> 15 with A from the not-yet-touched neighbour, handed from lane to lane by a DPP row shift; every other product is elementwise
 * (the scheme of celt_enc_pvq.h: rot_pass, there on the scalar unit).  A step is ~8 instructions without an LDS round trip. */
template <int D> WV_DEV i32 p4_rot_pass_reg(i32 v, int pos, int len, i32 c_, i32 s_)
{
   const i32 c = (i16)c_, s = (i16)s_;
   if (len - D > 0) {            /* upwards: (X[i], X[i+D]) <- (c X[i] - s X[i+D], c X[i+D] + s X[i]), i = 0 .. len-D-1 */
      const i32 xs = wg_shl<D>(v);
      const i32 A = add32(mult16_16(c, xs), 16384), Bv = mult16_16(s, xs);
      i32 x1 = v;
      const int nsteps = (len - 1) / D;                                  /* ceil((len - D) / D) */
      for (int m = 0; m < nsteps; m++) {
         const i32 t = wg_shr<D>((i32)(i16)(add32(A, s * x1) >> 15));
         if (pos >= (m + 1) * D && pos < (m + 2) * D) x1 = t;
      }
      v = pos < len - D ? (i32)(i16)(add32(sub32(mult16_16(c, x1), Bv), 16384) >> 15) : x1;
   }
   const int top = len - 2 * D - 1;
   if (top >= 0) {               /* downwards from i = len-2D-1, same butterfly: the chain carries the new X[i] down as the X[i+D] of step i-D */
      const i32 Cv = add32(mult16_16(c, v), 16384), Sv = mult16_16(s, v);
      i32 y = wg_shl<D>(v), h = v;
      const int nsteps = top / D + 1;
      for (int m = 0; m < nsteps; m++) {
         const i32 yn = (i32)(i16)(sub32(Cv, s * y) >> 15);
         const i32 t = wg_shl<D>(yn);
         if (pos < D && pos <= top - m * D && pos > top - (m + 1) * D) h = yn;
         if (pos <= top - (m + 1) * D && pos > top - (m + 2) * D) y = t;
      }
      const i32 ou = wg_shr<D>((i32)(i16)(add32(add32(mult16_16(c, y), Sv), 16384) >> 15));
      v = (pos >= D && pos <= len - D - 1) ? ou : h;
   }
   return v;
}
struct P4Rot { int on, len, stride2; i32 c, s; };
WV_DEV P4Rot p4_rot_setup(int len, int stride, int K, int spread)
{
   P4Rot r; r.on = p4_rot_applies(len, K, spread); r.len = len; r.stride2 = 0; r.c = 0; r.s = 0;
   if (r.on) {
      const int factor = spread == 1 ? 15 : (spread == 2 ? 10 : 5);
      const i16 gain = (i16)fx_div(mult16_16(Q15ONE, len), (i32)(len + factor * K));
      const i16 theta = (i16)(mult16_16_q15(gain, gain) >> 1);
      r.c = fx_cos_norm(theta);
      r.s = fx_cos_norm(sub16(Q15ONE, theta));
      if (len >= 8 * stride) {
         r.stride2 = 1;
         while ((r.stride2 * r.stride2 + r.stride2) * stride + (stride >> 2) < len) r.stride2++;
      }
      r.len = fx_div_pow2(len, stride);
   }
   return r;
}
/* v in Q14 (scaled down by the caller); T: the leaf's LDS words, the way round for a stride the row shifts are not instantiated for */
WV_DEV i32 p4_rot_pass_any(i32 v, WV_LDS i32 *T, int N, int pos, int nblk, int len, int d, i32 c, i32 s)
{
   if (d == 1) return p4_rot_pass_reg<1>(v, pos, len, c, s);
   if (d == 2) return p4_rot_pass_reg<2>(v, pos, len, c, s);
   if (d == 3) return p4_rot_pass_reg<3>(v, pos, len, c, s);
   if (d == 4) return p4_rot_pass_reg<4>(v, pos, len, c, s);
   wg_sync();
   if (wg_lane() < N) T[wg_lane()] = v;
   p4_rot_pass(T, nblk, len, d, c, s);
   v = wg_lane() < N ? T[wg_lane()] : 0;
   wg_sync();
   return v;
}
WV_DEV i32 p4_exp_rotation_reg(i32 v, WV_LDS i32 *T, int N, const P4Rot &r, int dir, int stride)
{
   const int gl = wg_lane();
   const int pos = gl < N ? gl - (int)fx_udiv24((u32)gl, (u32)r.len) * r.len : 1 << 20;
   v = pshr32(v, NORM_SHIFT - 14);
   if (dir < 0) {
      if (r.stride2) v = p4_rot_pass_any(v, T, N, pos, stride, r.len, r.stride2, r.s, r.c);
      v = p4_rot_pass_any(v, T, N, pos, stride, r.len, 1, r.c, r.s);
   } else {
      v = p4_rot_pass_any(v, T, N, pos, stride, r.len, 1, r.c, -r.s);
      if (r.stride2) v = p4_rot_pass_any(v, T, N, pos, stride, r.len, r.stride2, r.s, -r.c);
   }
   return shl32(v, NORM_SHIFT - 14);
}

/* ---- alg_quant (vq.c:552) of the leaf X[0 .. N) in the group's LDS.  The leaf is searched in registers when it fits one per lane (N <= 16: most leaves), else word by word
 * from LDS.  During the search a word of X holds |x| (15 bits) | sign << 15 | 2 * pulses << 16. ---- */
WV_DEV unsigned p4_alg_quant(WV_LDS P4Group *G, u8 *ecbuf, WV_LDS i32 *X, int N, int K, int spread, int B, i32 gain, int resynth)
{
   const int gl = wg_lane();
   const u32 ft = pvq_u(N, K) + pvq_u(N, K + 1);
   const int reg = N <= WG_WIDTH;
   const int rot = p4_rot_applies(N, K, spread);
#ifdef P4_STAT
   if (gl == 0) fprintf(stderr, "P4LEAF N %d K %d B %d spread %d rot %d\n", N, K, B, spread, rot);
#endif
   P4_TIC();
   wg_sync();
   if (rot && !reg) {
      FOR_GL(j, N) X[j] = pshr32(X[j], NORM_SHIFT - 14);
      p4_exp_rotation_q14(X, N, 1, B, K, spread);
      FOR_GL(j, N) X[j] = shl32(X[j], NORM_SHIFT - 14);
      wg_sync();
      P4_TOC(2);
   }
   /* op_pvq_search (vq.c:205) */
   i32 yy_out;
   unsigned cm = 1;
   u32 idx = 0;
   if (reg) {
      /* the leaf lives in registers from here to its last store: rotation, search, index, resynthesis */
      const bool vld = gl < N;
      i32 xv0 = vld ? X[gl] : 0;
      P4_TOC(1);
      const P4Rot rr = p4_rot_setup(N, B, K, spread);
      P4_TOC(3);
      if (rot) xv0 = p4_exp_rotation_reg(xv0, X, N, rr, 1, B);
      P4_TOC(16);
      i64 e2 = xv0 * (i64)xv0;
      int shift = (celt_ilog2(1 + (i32)(wg_sum64(e2) >> 2 * (NORM_SHIFT - 14))) + 1) / 2;
      shift = imax(0, shift + (NORM_SHIFT - 14) - 14);
      i32 xv = vld ? pshr32(xv0, shift) : 0;
      const i32 sg = xv < 0;
      i32 x = iabs(xv), y = 0, q = 0;
      i32 xy = 0; i16 yy = 0;
      int pulsesLeft = K;
      if (K > (N >> 1)) {
         i32 sum = wg_sum(x);
         if (sum <= K) { x = gl == 0 ? QC16(1.f, 14) : 0; sum = QC16(1.f, 14); }
         i16 rcp = extract16(mult16_32_q16(K, fx_rcp(sum)));
         q = mult16_16_q15(x, rcp);
         y = 2 * q;
         yy = (i16)wg_sum(mult16_16(q, q));
         xy = wg_sum(mult16_16(x, q));
         pulsesLeft -= wg_sum(q);
      }
      if (pulsesLeft > N + 3) {
         i16 tmp = (i16)pulsesLeft;
         i32 yfirst = wg_bcast(y, 0);
         yy = (i16)mac16_16(yy, tmp, tmp);
         yy = (i16)mac16_16(yy, tmp, yfirst);
         if (gl == 0) q += pulsesLeft;
         pulsesLeft = 0;
      }
      for (int i = 0; i < pulsesLeft; i++) {
         int rshift = 1 + celt_ilog2(K - pulsesLeft + i + 1);
         yy = add16(yy, 1);
         i16 Rxy = extract16(add32(xy, x) >> rshift); i16 Ryy = add16(yy, y); Rxy = (i16)mult16_16_q15(Rxy, Rxy);
         const int owner = wg_argmax_ratio_packed(vld ? (u32)Rxy : 0u, vld ? (u32)Ryy : 1u, vld);
         const i32 w = wg_bcast(x | (y << 16), owner);
         xy = add32(xy, w & 0xffff);
         yy = add16(yy, w >> 16);
         if (gl == owner) { y += 2; q++; }
      }
      q = (q ^ -sg) + sg;
      yy_out = yy;
      P4_TOC(17);
      if (B > 1) { const int N0 = fx_div_pow2(N, B); cm = wg_or(q != 0 ? 1u << ((u32)gl / (u32)N0) : 0u); }
      /* encode_pulses (cwrs.c:444) */
      {
         const i32 a = iabs(q), incl = wg_scan_incl(a), tot = wg_bcast(incl, WG_WIDTH - 1);
         const i32 kafter = tot - incl;
         if (gl < N - 1) { idx += pvq_u(N - gl, kafter); if (q < 0) idx += pvq_u(N - gl, kafter + a + 1); }
         else if (gl == N - 1) idx += q < 0;
         idx = wg_sumu(idx);
      }
      GLANE0 { P4_EC_BEGIN; k_ec_enc_uint(EC_PASS, idx, ft); P4_EC_END; }
      P4_TOC(18);
      if (resynth) {
         int k = celt_ilog2(yy_out) >> 1;
         i32 t_ = vshr32(yy_out, 2 * (k - 7) - 15);
         i32 g = mult32_32_q31(fx_rsqrt_norm32(t_), gain);
         i32 v = vld ? vshr32(mult16_32_q15(q, g), k + 15 - NORM_SHIFT) : 0;
         if (rot) v = p4_exp_rotation_reg(v, X, N, rr, -1, B);
         wg_sync();
         if (vld) X[gl] = v;
         wg_sync();
      }
   } else {
      P4_TOC(16);
      i64 e2 = 0;
      FOR_GL(j, N) e2 += X[j] * (i64)X[j];
      int shift = (celt_ilog2(1 + (i32)(wg_sum64(e2) >> 2 * (NORM_SHIFT - 14))) + 1) / 2;
      shift = imax(0, shift + (NORM_SHIFT - 14) - 14);
      i32 xsum = 0;
      FOR_GL(j, N) { const i32 xv = pshr32(X[j], shift), ax = iabs(xv); X[j] = ax | (xv < 0 ? 0x8000 : 0); xsum += ax; }
      wg_sync();
      i32 xy = 0; i16 yy = 0;
      int pulsesLeft = K;
      if (K > (N >> 1)) {
         i32 sum = wg_sum(xsum);
         if (sum <= K) { FOR_GL(j, N) X[j] = (X[j] & 0x8000) | (j == 0 ? QC16(1.f, 14) : 0); sum = QC16(1.f, 14); wg_sync(); }
         i16 rcp = extract16(mult16_32_q16(K, fx_rcp(sum)));
         i32 yyp = 0, xyp = 0, qs = 0;
         FOR_GL(j, N) {
            const i32 w = X[j], x = w & 0x7fff, q = mult16_16_q15(x, rcp);
            yyp = mac16_16(yyp, q, q); xyp = mac16_16(xyp, x, q); qs += q;
            X[j] = (w & 0xffff) | (2 * q << 16);
         }
         wg_sync();
         yy = (i16)wg_sum(yyp);
         xy = wg_sum(xyp);
         pulsesLeft -= wg_sum(qs);
      }
      if (pulsesLeft > N + 3) {
         i16 tmp = (i16)pulsesLeft;
         i32 yfirst = X[0] >> 16;
         yy = (i16)mac16_16(yy, tmp, tmp);
         yy = (i16)mac16_16(yy, tmp, yfirst);
         wg_sync();
         if (gl == 0) X[0] = (X[0] & 0xffff) | ((yfirst + 2 * pulsesLeft) << 16);
         wg_sync();
         pulsesLeft = 0;
      }
      for (int i = 0; i < pulsesLeft; i++) {
         int rshift = 1 + celt_ilog2(K - pulsesLeft + i + 1);
         yy = add16(yy, 1);
         /* per-lane best over its elements gl, gl + 16, ...; ties inside a lane go to the lower index */
         u32 bn = 0, bd = 1; int bj = -1;
         FOR_GL(j, N) {
            const i32 w = X[j];
            i16 Rxy = extract16(add32(xy, w & 0x7fff) >> rshift); const i16 Ryy = add16(yy, w >> 16); Rxy = (i16)mult16_16_q15(Rxy, Rxy);
            if (bj < 0 || bd * (u32)Rxy > (u32)Ryy * bn) { bn = (u32)Rxy; bd = (u32)Ryy; bj = j; }
         }
         /* global order is by index: maximal ratio first, then the lowest index that attains it */
         const int any = wg_argmax_ratio_packed(bn, bd, bj >= 0);
         const u32 gn = (u32)wg_bcast((i32)bn, any), gd = (u32)wg_bcast((i32)bd, any);
         int cand = 0x7fffffff;
         FOR_GL(j, N) {
            const i32 w = X[j];
            i16 Rxy = extract16(add32(xy, w & 0x7fff) >> rshift); const i16 Ryy = add16(yy, w >> 16); Rxy = (i16)mult16_16_q15(Rxy, Rxy);
            if ((u32)Ryy * gn == gd * (u32)Rxy) { cand = j; break; }
         }
         const int win = -wg_max(-cand);
         const i32 w = X[win];
         xy = add32(xy, w & 0x7fff);
         yy = add16(yy, w >> 16);
         wg_sync();
         if (gl == (win & (WG_WIDTH - 1))) X[win] = w + (2 << 16);
         wg_sync();
      }
      yy_out = yy;
      P4_TOC(17);
      /* collapse mask, encode_pulses: signed pulse counts q_j = (y_j / 2) with the sign of x_j */
      if (B > 1) {
         const int N0 = fx_div_pow2(N, B);
         u32 m = 0;
         FOR_GL(j, N) if (X[j] >> 16) m |= 1u << ((u32)j / (u32)N0);
         cm = wg_or(m);
      }
      {
         const int nr = (N + WG_WIDTH - 1) / WG_WIDTH;
         i32 above = 0;
         for (int t = nr - 1; t >= 0; t--) {
            const int j = gl + WG_WIDTH * t;
            const i32 w = j < N ? X[j] : 0, a = w >> 17, neg = (w >> 15) & 1;
            const i32 incl = wg_scan_incl(a), tot = wg_bcast(incl, WG_WIDTH - 1);
            const i32 kafter = above + tot - incl;
            if (j < N - 1) { idx += pvq_u(N - j, kafter); if (neg && a) idx += pvq_u(N - j, kafter + a + 1); }
            else if (j == N - 1) idx += (neg && a);
            above += tot;
         }
         idx = wg_sumu(idx);
      }
      GLANE0 { P4_EC_BEGIN; k_ec_enc_uint(EC_PASS, idx, ft); P4_EC_END; }
      P4_TOC(18);
      if (resynth) {
         int k = celt_ilog2(yy_out) >> 1;
         i32 t_ = vshr32(yy_out, 2 * (k - 7) - 15);
         i32 g = mult32_32_q31(fx_rsqrt_norm32(t_), gain);
         wg_sync();
         FOR_GL(j, N) { const i32 w = X[j]; i32 q = w >> 17; if (w & 0x8000) q = -q; const i32 v = vshr32(mult16_32_q15(q, g), k + 15 - NORM_SHIFT); X[j] = rot ? pshr32(v, NORM_SHIFT - 14) : v; }
         wg_sync();
      }
   }
   if (resynth && rot && !reg) {
      p4_exp_rotation_q14(X, N, -1, B, K, spread);
      FOR_GL(j, N) X[j] = shl32(X[j], NORM_SHIFT - 14);
      wg_sync();
   }
   P4_TOC(19);
   return cm;
}

/* pulse-cache row of (LM, band): staged once per band and wave (bands go in lockstep) */
WV_DEV void p4_rows_stage(WV_LDS P4Lds *L, int band)
{
   wv_sync();
   for (int t = wv_lane(); t < 5 * 64; t += WV_WIDTH) {
      const int d = t >> 6, e = t & 63;
      const int off = ct_cache_index[d * OA_NB_EBANDS + band];
      L->rows[t] = ct_cache_bits[imax(0, imin(off + imin(e, 40), (int)sizeof(ct_cache_bits) - 1))];
   }
   wv_sync();
}
WV_DEV int p4_row(const WV_LDS u8 *row, int k) { return (int)row[k]; }
WV_DEV int p4_bits2pulses(const WV_LDS u8 *row, int bits)
{
   int lo = 0, hi = p4_row(row, 0);
   bits--;
   for (int i = 0; i < LOG_MAX_PSEUDO; i++) {
      int mid = (lo + hi + 1) >> 1;
      if (p4_row(row, mid) >= bits) hi = mid; else lo = mid;
   }
   if (bits - (lo == 0 ? -1 : p4_row(row, lo)) <= p4_row(row, hi) - bits) return lo;
   return hi;
}
WV_DEV int p4_pulses2bits(const WV_LDS u8 *row, int pulses) { return pulses == 0 ? 0 : p4_row(row, pulses) + 1; }

/* the per-band context of a group (the read-only part of band_ctx, bands.c:664) */
struct P4Cfg { int i, resynth, intensity, spread, tf_change, theta_round, disable_inv, avoid_split_noise; };
struct P4Theta { int inv, imid, iside, delta, itheta, qalloc, b, fill; };

/* compute_theta (bands.c:700) */
WV_DEV P4Theta p4_compute_theta(WV_LDS P4Group *G, u8 *ecbuf, const P4Cfg &cfg, i32 remaining_bits, WV_LDS i32 *X, WV_LDS i32 *Y, int N, int b, int B, int B0, int LM, int stereo, int fill)
{
   int qn, itheta = 0, delta, imid, iside, qalloc, pulse_cap, offset, inv = 0;
   const int i = cfg.i, intensity = cfg.intensity;
   P4_TIC();
   pulse_cap = ct_logN[i] + LM * (1 << BITRES);
   offset = (pulse_cap >> 1) - (stereo && N == 2 ? 16 : 4);
   qn = p4_compute_qn(N, b, offset, pulse_cap, stereo);
   if (stereo && i >= intensity) qn = 1;
   itheta = p4_stereo_itheta(X, Y, stereo, N) >> 16;
   wg_sync();
   i32 tell = ec_tell_frac_lds(&G->ec);
   wg_sync();
   if (qn != 1) {
      if (!stereo || cfg.theta_round == 0) {
         itheta = (itheta * (i32)qn + 8192) >> 14;
         if (!stereo && cfg.avoid_split_noise && itheta > 0 && itheta < qn) {
            int unquantized = (int)fx_udiv24((u32)((i32)itheta * 16384), (u32)qn);
            imid = bitexact_cos((i16)unquantized);
            iside = bitexact_cos((i16)(16384 - unquantized));
            delta = frac_mul16((N - 1) << 7, bitexact_log2tan(iside, imid));
            if (delta > b) itheta = qn;
            else if (delta < -b) itheta = 0;
         }
      } else {
         const int q32767 = (int)fx_udiv24(32767u, (u32)qn);
         int bias = itheta > 8192 ? q32767 : -q32767;
         int down = imin(qn - 1, imax(0, (itheta * (i32)qn + bias) >> 14));
         itheta = cfg.theta_round < 0 ? down : down + 1;
      }
      GLANE0 {
         P4_EC_BEGIN;
         if (stereo && N > 2) {
            int p0 = 3, x = itheta, x0 = qn / 2, ft = p0 * (x0 + 1) + x0;
            k_ec_encode(EC_PASS, x <= x0 ? p0 * x : (x - 1 - x0) + (x0 + 1) * p0, x <= x0 ? p0 * (x + 1) : (x - x0) + (x0 + 1) * p0, ft);
         } else if (B0 > 1 || stereo) {
            k_ec_enc_uint(EC_PASS, itheta, qn + 1);
         } else {
            int ft = ((qn >> 1) + 1) * ((qn >> 1) + 1);
            int fs = itheta <= (qn >> 1) ? itheta + 1 : qn + 1 - itheta;
            int fl = itheta <= (qn >> 1) ? itheta * (itheta + 1) >> 1 : ft - ((qn + 1 - itheta) * (qn + 2 - itheta) >> 1);
            k_ec_encode(EC_PASS, fl, fl + fs, ft);
         }
         P4_EC_END;
      }
      itheta = (int)fx_udiv24((u32)((i32)itheta * 16384), (u32)qn);
      if (stereo) {
         if (itheta == 0) p4_intensity_stereo(G->bandE, X, Y, i, N);
         else {
            wg_sync();
            FOR_GL(j, N) {
               i32 l = mult32_32_q31(QC32(.70710678f, 31), X[j]);
               i32 r = mult32_32_q31(QC32(.70710678f, 31), Y[j]);
               X[j] = add32(l, r);
               Y[j] = sub32(r, l);
            }
            wg_sync();
         }
      }
   } else if (stereo) {
      inv = itheta > 8192 && !cfg.disable_inv;
      if (inv) { wg_sync(); FOR_GL(j, N) Y[j] = neg32(Y[j]); wg_sync(); }
      p4_intensity_stereo(G->bandE, X, Y, i, N);
      if (b > 2 << BITRES && remaining_bits > 2 << BITRES) {
         GLANE0 { P4_EC_BEGIN; k_ec_enc_bit_logp(EC_PASS, inv, 2); P4_EC_END; }
      } else inv = 0;
      if (cfg.disable_inv) inv = 0;
      itheta = 0;
   }
   wg_sync();
   qalloc = (int)ec_tell_frac_lds(&G->ec) - tell;
   b -= qalloc;
   if (itheta == 0) { imid = 32767; iside = 0; fill &= (1 << B) - 1; delta = -16384; }
   else if (itheta == 16384) { imid = 0; iside = 32767; fill &= ((1 << B) - 1) << B; delta = 16384; }
   else {
      imid = bitexact_cos((i16)itheta);
      iside = bitexact_cos((i16)(16384 - itheta));
      delta = frac_mul16((N - 1) << 7, bitexact_log2tan(iside, imid));
   }
   P4_TOC(stereo ? 31 : 20);
   P4Theta r = {inv, imid, iside, delta, itheta, qalloc, b, fill};
   return r;
}

/* ---- quant_partition (bands.c:973) as a walk over an explicit stack.  The state of a group: the node it stands on (nd_*), its depth, the running budget and seed.
 * p4_tree_run takes all four groups through their trees side by side; on return tr.cm is the root's collapse mask. ---- */
struct P4Tree { int act; int xo, N, b, B, LM, lb, fill; i32 gain; int depth, done; unsigned cm; i32 remaining_bits; u32 seed; };

WV_DEV void p4_tree_run(WV_LDS P4Lds *L4, WV_LDS P4Group *G, u8 *ecbuf, const P4Cfg &cfg, P4Tree &tr)
{
   tr.depth = 0; tr.done = !tr.act; tr.cm = 0;
   P4_TIC();
   while (wv_any(!tr.done)) {
      /* down through the split nodes to the next leaf */
      for (;;) {
         int split = 0;
         const WV_LDS u8 *row = L4->rows + (tr.LM + 1) * 64;
         if (!tr.done) split = tr.LM != -1 && tr.b > p4_row(row, p4_row(row, 0)) + 12 && tr.N > 2;
         if (!wv_any(split)) break;
         if (split) {
            const int B0 = tr.B;
            int N = tr.N >> 1, LM = tr.LM - 1, fill = tr.fill, B;
            if (B0 == 1) fill = (fill & 1) | (fill << 1);
            B = (B0 + 1) >> 1;
            WV_LDS i32 *X = G->Xb + tr.xo;
            const P4Theta th = p4_compute_theta(G, ecbuf, cfg, tr.remaining_bits, X, X + N, N, tr.b, B, B0, LM, 0, fill);
            int delta = th.delta;
            const int itheta = th.itheta, b = th.b;
            fill = th.fill;
            const i32 mid = shl32((i32)th.imid, 16), side = shl32((i32)th.iside, 16);
            if (B0 > 1 && (itheta & 0x3fff)) {
               if (itheta > 8192) delta -= delta >> (4 - LM);
               else delta = imin(0, delta + (N << BITRES >> (5 - LM)));
            }
            const int mbits = imax(0, imin(b, (b - delta) / 2)), sbits = b - mbits;
            tr.remaining_bits -= th.qalloc;
            const int mid_first = mbits >= sbits;
            const i32 gm = mult32_32_q31(tr.gain, mid), gs = mult32_32_q31(tr.gain, side);
            WV_LDS P4Frame *f = &G->stk[tr.depth];
            GLANE0 {
               f->xo = tr.xo; f->N = N; f->B = B; f->B0 = B0; f->LM = LM; f->lb = tr.lb; f->gm = gm; f->gs = gs; f->fill = fill; f->mbits = mbits; f->sbits = sbits;
               f->itheta = itheta; f->rebal = tr.remaining_bits; f->mid_first = mid_first; f->phase = 0; f->cm = 0;
            }
            tr.N = N; tr.B = B; tr.LM = LM;
            if (mid_first) { tr.b = mbits; tr.gain = gm; tr.fill = fill; }
            else { tr.xo += N; tr.b = sbits; tr.gain = gs; tr.fill = fill >> B; tr.lb = tr.lb >= 0 ? tr.lb + N : -1; }
            tr.depth++;
         }
      }
      /* the leaf */
      P4_TOC(25);
      unsigned cm = 0;
      if (!tr.done) {
         const WV_LDS u8 *row = L4->rows + (tr.LM + 1) * 64;
         WV_LDS i32 *X = G->Xb + tr.xo;
         const int N = tr.N, B = tr.B;
         int q = p4_bits2pulses(row, tr.b);
         int curr_bits = p4_pulses2bits(row, q);
         tr.remaining_bits -= curr_bits;
         while (tr.remaining_bits < 0 && q > 0) {
            tr.remaining_bits += curr_bits;
            q--;
            curr_bits = p4_pulses2bits(row, q);
            tr.remaining_bits -= curr_bits;
         }
         P4_TOC(26);
         if (q != 0) cm = p4_alg_quant(G, ecbuf, X, N, k_get_pulses(q), cfg.spread, B, tr.gain, cfg.resynth);
         else if (cfg.resynth) {
            const unsigned cm_mask = (unsigned)(1UL << B) - 1;
            const int fill = tr.fill & (int)cm_mask;
            wg_sync();
            if (!fill) { FOR_GL(j, N) X[j] = 0; wg_sync(); }
            else {
               /* noise, or the folded band plus noise: X[j] depends on the j-th step of the LCG -- every lane takes its own steps (seed_j = A^j seed + c_j: j serial
                * steps of a two-instruction recurrence cost less than a look-up of the jump-ahead constants) */
               const WV_LDS i32 *lowband = tr.lb >= 0 ? G->lbs + tr.lb : (const WV_LDS i32 *)0;
               u32 s = tr.seed;
               int jdone = 0;
               FOR_GL(j, N) {
                  for (; jdone <= j; jdone++) s = lcg_rand(s);
                  if (lowband == 0) X[j] = shl32((i32)((i32)s >> 20), NORM_SHIFT - 14);
                  else { i16 tmp = QC16(1.0f / 256, NORM_SHIFT - 4); tmp = (s) & 0x8000 ? tmp : -tmp; X[j] = lowband[j] + tmp; }
               }
               cm = lowband == 0 ? cm_mask : (unsigned)fill;
               for (int j = 0; j < N; j++) tr.seed = lcg_rand(tr.seed);
               wg_sync();
               p4_renormalise_vector(X, N, tr.gain);
            }
         }
      }
      /* back up: finished nodes hand their mask to the parent; the first parent with a child to go sends the group down again */
      P4_TOC(30);
      if (!tr.done) {
         for (;;) {
            if (tr.depth == 0) { tr.done = 1; tr.cm = cm; break; }
            WV_LDS P4Frame *f = &G->stk[tr.depth - 1];
            wg_sync();
            const int phase = f->phase, mid_first = f->mid_first, B0 = f->B0;
            if (phase == 0) {
               int mbits = f->mbits, sbits = f->sbits;
               const int itheta = f->itheta, N = f->N, B = f->B;
               i32 rebalance = f->rebal;
               unsigned fcm;
               tr.N = N; tr.B = B; tr.LM = f->LM;
               if (mid_first) {
                  fcm = cm;
                  rebalance = mbits - (rebalance - tr.remaining_bits);
                  if (rebalance > 3 << BITRES && itheta != 0) sbits += rebalance - (3 << BITRES);
                  tr.xo = f->xo + N; tr.b = sbits; tr.gain = f->gs; tr.fill = f->fill >> B; tr.lb = f->lb >= 0 ? f->lb + N : -1;
               } else {
                  fcm = cm << (B0 >> 1);
                  rebalance = sbits - (rebalance - tr.remaining_bits);
                  if (rebalance > 3 << BITRES && itheta != 16384) mbits += rebalance - (3 << BITRES);
                  tr.xo = f->xo; tr.b = mbits; tr.gain = f->gm; tr.fill = f->fill; tr.lb = f->lb;
               }
               wg_sync();
               GLANE0 { f->phase = 1; f->cm = (i32)fcm; }
               break;
            }
            cm = (unsigned)f->cm | (mid_first ? cm << (B0 >> 1) : cm);
            tr.depth--;
         }
      }
      P4_TOC(27);
   }
}

/* quant_band (bands.c:1248) around the tree: what is done to the band before (p4_qb_pre) and after (p4_qb_post) quant_partition */
struct P4Qb { int N0, recombine, time_divide, B0, N_B0, longBlocks, B; };

WV_DEV void p4_qb_pre(WV_LDS P4Group *G, P4Tree &tr, P4Qb &qb, int tf_change, int nmax, int do_x = 1 /* 0: the decoder -- the band has no content yet, only the folding source is transformed */)
{
   WV_LDS i32 *X = G->Xb + tr.xo;
   WV_LDS i32 *lowband = tr.lb >= 0 ? G->lbs + tr.lb : (WV_LDS i32 *)0;
   const int N = tr.N;
   int B = tr.B, fill = tr.fill, N_B = fx_div_pow2(N, B);
   qb.N0 = N; qb.recombine = 0; qb.time_divide = 0; qb.longBlocks = B == 1;
   P4_TIC();
   if (tf_change > 0) qb.recombine = tf_change;
   for (int k = 0; k < qb.recombine; k++) {
      if (do_x) p4_haar1(X, N >> k, 1 << k);
      if (lowband) p4_haar1(lowband, N >> k, 1 << k);
      fill = k_bit_interleave_table[fill & 0xF] | k_bit_interleave_table[fill >> 4] << 2;
   }
   B >>= qb.recombine;
   N_B <<= qb.recombine;
   while ((N_B & 1) == 0 && tf_change < 0) {
      if (do_x) p4_haar1(X, N_B, B);
      if (lowband) p4_haar1(lowband, N_B, B);
      fill |= fill << B;
      B <<= 1;
      N_B >>= 1;
      qb.time_divide++;
      tf_change++;
   }
   qb.B0 = B; qb.N_B0 = N_B;
   if (B > 1) {
      if (do_x) p4_deinterleave_hadamard(X, N_B >> qb.recombine, B << qb.recombine, qb.longBlocks, nmax);
      if (lowband) p4_deinterleave_hadamard(lowband, N_B >> qb.recombine, B << qb.recombine, qb.longBlocks, nmax);
   }
   tr.B = B; tr.fill = fill;
   P4_TOC(22);
}
/* returns the band's collapse mask; lowband_out: HBM (the folding memory or a trial's slot), or NULL */
WV_DEV unsigned p4_qb_post(WV_LDS P4Group *G, const P4Tree &tr, const P4Qb &qb, int xo, int resynth, i32 *lowband_out, int nmax)
{
   unsigned cm = tr.cm;
   P4_TIC();
   if (resynth) {
      WV_LDS i32 *X = G->Xb + xo;
      int B = qb.B0, N_B = qb.N_B0;
      if (qb.B0 > 1) p4_interleave_hadamard(X, N_B >> qb.recombine, qb.B0 << qb.recombine, qb.longBlocks, nmax);
      for (int k = 0; k < qb.time_divide; k++) {
         B >>= 1;
         N_B <<= 1;
         cm |= cm >> B;
         p4_haar1(X, N_B, B);
      }
      for (int k = 0; k < qb.recombine; k++) {
         cm = k_bit_deinterleave_table[cm];
         p4_haar1(X, qb.N0 >> k, 1 << k);
      }
      B <<= qb.recombine;
      if (lowband_out) {
         const i16 n = (i16)fx_sqrt(shl32((i32)qb.N0, 22));
         wg_sync();
         FOR_GL(j, qb.N0) lowband_out[j] = mult16_32_q15(n, X[j]);
      }
      cm &= (1u << B) - 1;
   }
   P4_TOC(22);
   return cm;
}

WV_DEV void p4_channel_weights(i32 Ex, i32 Ey, i32 &w0, i32 &w1)
{
   i32 minE = imin(Ex, Ey);
   Ex = add32(Ex, minE / 3);
   Ey = add32(Ey, minE / 3);
   int shift = celt_ilog2(EPSILON + imax(Ex, Ey)) - 14;
   w0 = (i16)vshr32(Ex, shift);
   w1 = (i16)vshr32(Ey, shift);
}

/* quant_all_bands (bands.c:1589), encoder side, for the (up to) four streams of a wave.  cont: this lane's group's record, or NULL (a group without a stream). */
WV_DEV void p4_quant_all_bands(WV_LDS P4Lds *L4, CeltCont *cont)
{
   WV_LDS P4Group *G = &L4->g[wg_id()];
   const int active = cont != 0;
   FrameLds *img = (FrameLds *)(cont ? cont->image : 0);
   /* the stream's frame constants (group-uniform registers) */
   int start = 0, end = 0, LM = 3, C = 1, shortBlocks = 0, spread = 0, dual_stereo = 0, intensity = 0, codedBands = 0, complexity = 0, disable_inv = 0;
   i32 total_bits = 0, balance = 0;
   u32 seed = 0;
   if (active) {
      start = img->sh.start; end = img->sh.end; LM = img->sh.LM; C = img->sh.C; shortBlocks = img->sh.shortBlocks; spread = img->st.spread_decision; dual_stereo = img->sh.dual_stereo;
      intensity = img->st.intensity; codedBands = img->sh.codedBands; complexity = img->sh.complexity; disable_inv = img->sh.disable_inv; balance = img->sh.balance;
      total_bits = img->sh.nbCompressedBytes * (8 << BITRES) - img->sh.anti_collapse_rsv;
      seed = img->st.rng;
      wg_sync();
      FOR_GL(k, NBE) { G->pulses[k] = img->pulses[k]; G->tf_res[k] = img->tf_res[k]; }
      FOR_GL(k, 2 * NBE) G->bandE[k] = img->bandE[k];
      FOR_GL(k, (int)(sizeof(EcCtx) / 4)) ((WV_LDS i32 *)&G->ec)[k] = ((const i32 *)&img->ec)[k];
      wg_sync();
   }
   u8 *const pkt = active ? img->packet + 1 : (u8 *)0;
   const i32 *const X_ = active ? cont->X[0] : (const i32 *)0, *const Y_ = active && C == 2 ? cont->X[1] : (const i32 *)0;
   i32 *const norm = active ? cont->norm[0] : (i32 *)0, *const norm2 = active ? cont->norm[1] : (i32 *)0;
   /* wave-uniform: the frame size class, the union of the groups' band ranges */
   const int LMu = wv_max(active ? LM : 0), Mu = 1 << LMu;
   const int i_lo = wv_min(active ? start : NBE), i_hi = wv_max(active ? end : 0);
   const int M = Mu;
   int B = shortBlocks ? M : 1, lowband_offset = 0, update_lowband = 1;
   const int norm_offset = M * ct_eBands[start];
   const int theta_rdo = Y_ != 0 && !dual_stereo && complexity >= 8;
   const int resynth = theta_rdo;
   P4Cfg cfg;
   cfg.intensity = intensity; cfg.spread = spread; cfg.disable_inv = disable_inv; cfg.resynth = resynth; cfg.theta_round = 0; cfg.avoid_split_noise = B > 1; cfg.i = 0; cfg.tf_change = 0;
   for (int i = i_lo; i < i_hi; i++) {
      const int N = M * ct_eBands[i + 1] - M * ct_eBands[i];          /* wave-uniform */
      P4_TIC();
      p4_rows_stage(L4, i);
      const int act = active && i >= start && i < end;
      const int last = i == end - 1;
      const i32 *Xg = X_ + M * ct_eBands[i], *Yg = Y_ != 0 ? Y_ + M * ct_eBands[i] : (const i32 *)0;
      i32 remaining_bits = 0, tell = 0;
      int b = 0, effective_lowband = -1, tf_change = 0;
      unsigned x_cm = 0, y_cm = 0;
      cfg.i = i;
      if (act) {
         wg_sync();
         tell = (i32)ec_tell_frac_lds(&G->ec);
         if (i != start) balance -= tell;
         remaining_bits = total_bits - tell - 1;
         if (i <= codedBands - 1) {
            const i32 curr_balance = fx_sdiv24(balance, imin(3, codedBands - i));
            b = imax(0, imin(16383, imin(remaining_bits + 1, G->pulses[i] + curr_balance)));
         } else b = 0;
         if (resynth && (M * ct_eBands[i] - N >= M * ct_eBands[start] || i == start + 1) && (update_lowband || lowband_offset == 0)) lowband_offset = i;
         if (resynth && i == start + 1) {                                  /* special_hybrid_folding (bands.c:1575) */
            const int hf_n1 = M * (ct_eBands[start + 1] - ct_eBands[start]), hf_n2 = M * (ct_eBands[start + 2] - ct_eBands[start + 1]);
            FOR_GL(j, hf_n2 - hf_n1) { norm[hf_n1 + j] = norm[2 * hf_n1 - hf_n2 + j]; if (dual_stereo) norm2[hf_n1 + j] = norm2[2 * hf_n1 - hf_n2 + j]; }
         }
         tf_change = G->tf_res[i];
         cfg.tf_change = tf_change;
         if (lowband_offset != 0 && (spread != 3 || B > 1 || tf_change < 0)) {
            int fold_start, fold_end, fold_i;
            effective_lowband = imax(0, M * ct_eBands[lowband_offset] - norm_offset - N);
            fold_start = lowband_offset;
            while (M * ct_eBands[--fold_start] > effective_lowband + norm_offset);
            fold_end = lowband_offset - 1;
            while (++fold_end < i && M * ct_eBands[fold_end] < effective_lowband + norm_offset + N);
            fold_i = fold_start;
            do {
               x_cm |= G->cmask[fold_i * C + 0];
               y_cm |= G->cmask[fold_i * C + C - 1];
            } while (++fold_i < fold_end);
         } else x_cm = y_cm = (1u << B) - 1;
         if (dual_stereo && i == intensity) {
            dual_stereo = 0;
            if (resynth) { FOR_GL(j, M * ct_eBands[i] - norm_offset) norm[j] = half32(norm[j] + norm2[j]); }
         }
      }
      /* the slots of this band: mono -- one quant_band; dual stereo -- the two channels one after the other through Xb with half the budget each (bands.c:1831-1841);
       * joint stereo -- quant_band_stereo: theta, then mid and side in the order of their budgets; with the theta RDO twice (rounded down, then up: :1842-1912) */
      const int joint = act && Y_ != 0 && !dual_stereo;
      const int rdo = joint && theta_rdo && i < intensity;
      const int ntrials = act ? 1 + rdo : 0, nslots = act ? ((dual_stereo || joint) ? 2 : 1) : 0;
      i32 *const lbo = last ? (i32 *)0 : norm + M * ct_eBands[i] - norm_offset, *const lbo2 = last ? (i32 *)0 : norm2 + M * ct_eBands[i] - norm_offset;
      const i32 rem0 = remaining_bits; const u32 seed0 = seed;
      const unsigned cm_in = x_cm | y_cm;
      P4_TOC(0);
      i32 dist0 = 0, rem1 = 0, w0 = 0, w1 = 0;
      u32 seed1 = 0;
      unsigned cm2 = 0;
      if (rdo) {
         p4_channel_weights(G->bandE[i], G->bandE[i + NBE], w0, w1);
         wg_sync();
         FOR_GL(k, (int)(sizeof(EcCtx) / 4)) ((WV_LDS i32 *)&G->ecsave[0])[k] = ((const WV_LDS i32 *)&G->ec)[k];
         wg_sync();
      }
      for (int tr_i = 0; tr_i < 2; tr_i++) {
         const int t_act = tr_i < ntrials;
         if (!wv_any(t_act)) break;
         u8 *ecbuf = pkt;
         int mbits = 0, sbits = 0, itheta = 0, inv = 0, mid_first = 1, fill_j = 0;
         i32 mid = 0, side = 0, rebalance = 0;
         if (t_act) {
            P4_TIC();
            if (tr_i == 1) {
               /* the first trial's outcome is parked (its bytes stay in the packet, its folding output in norm_alt[0]); the second starts from the same coder state,
                * budget and seed and codes into the stream's alternative buffer */
               wg_sync();
               dist0 = mult16_32_q15(w0, p4_inner_prod_g(Xg, G->Xb, N)) + mult16_32_q15(w1, p4_inner_prod_g(Yg, G->Yb, N));
               cm2 = x_cm; rem1 = remaining_bits; seed1 = seed;
               wg_sync();
               FOR_GL(k, (int)(sizeof(EcCtx) / 4)) { ((WV_LDS i32 *)&G->ecsave[1])[k] = ((const WV_LDS i32 *)&G->ec)[k]; }
               wg_sync();
               FOR_GL(k, (int)(sizeof(EcCtx) / 4)) { ((WV_LDS i32 *)&G->ec)[k] = ((const WV_LDS i32 *)&G->ecsave[0])[k]; }
               wg_sync();
               remaining_bits = rem0; seed = seed0;
               ecbuf = cont->alt;
            }
            /* stage the band (and its folding source: always a private copy) */
            wg_sync();
            FOR_GL(j, N) { G->Xb[j] = Xg[j]; if (joint) G->Yb[j] = Yg[j]; if (effective_lowband != -1) G->lbs[j] = norm[effective_lowband + j]; }
            wg_sync();
            P4_TOC(21);
            if (joint) {
               /* quant_band_stereo (bands.c:1387) up to the first quant_band */
               if (G->bandE[i] < 2 || G->bandE[NBE + i] < 2) {
                  wg_sync();
                  if (G->bandE[i] > G->bandE[NBE + i]) { FOR_GL(j, N) G->Yb[j] = G->Xb[j]; }
                  else { FOR_GL(j, N) G->Xb[j] = G->Yb[j]; }
                  wg_sync();
               }
               cfg.theta_round = rdo ? 2 * tr_i - 1 : 0;
               const P4Theta th = p4_compute_theta(G, ecbuf, cfg, remaining_bits, G->Xb, G->Yb, N, b, B, B, LMu, 1, (int)cm_in);
               inv = th.inv; itheta = th.itheta; fill_j = th.fill;
               mid = shl32((i32)th.imid, 16); side = shl32((i32)th.iside, 16);
               mbits = imax(0, imin(th.b, (th.b - th.delta) / 2));
               sbits = th.b - mbits;
               remaining_bits -= th.qalloc;
               rebalance = remaining_bits;
               mid_first = mbits >= sbits;
            }
         }
         for (int sl = 0; sl < 2; sl++) {
            const int s_act = t_act && sl < nslots;
            if (!wv_any(s_act)) break;
            P4Tree tr; P4Qb qb;
            tr.act = s_act; tr.xo = 0; tr.N = N; tr.b = 0; tr.B = B; tr.LM = LMu; tr.lb = -1; tr.fill = 0; tr.gain = Q31ONE; tr.remaining_bits = remaining_bits; tr.seed = seed;
            tr.depth = 0; tr.done = 1; tr.cm = 0;
            qb.N0 = N; qb.recombine = 0; qb.time_divide = 0; qb.B0 = B; qb.N_B0 = N; qb.longBlocks = 1; qb.B = B;
            i32 *lb_out = (i32 *)0;
            int do_mid = 1;
            if (s_act) {
               if (joint) {
                  do_mid = (sl == 0) == (mid_first != 0);
                  if (sl == 1) {
                     if (mid_first) { rebalance = mbits - (rebalance - remaining_bits); if (rebalance > 3 << BITRES && itheta != 0) sbits += rebalance - (3 << BITRES); }
                     else { rebalance = sbits - (rebalance - remaining_bits); if (rebalance > 3 << BITRES && itheta != 16384) mbits += rebalance - (3 << BITRES); }
                  }
                  tr.xo = do_mid ? 0 : OA_MAX_BAND; tr.b = do_mid ? mbits : sbits; tr.lb = do_mid && effective_lowband != -1 ? 0 : -1; tr.gain = do_mid ? Q31ONE : side;
                  tr.fill = do_mid ? fill_j : fill_j >> B;
                  lb_out = do_mid && !last ? (rdo ? cont->norm_alt[tr_i] : lbo) : (i32 *)0;
               } else if (dual_stereo) {
                  if (sl == 1) { wg_sync(); FOR_GL(j, N) { G->Xb[j] = Yg[j]; if (effective_lowband != -1) G->lbs[j] = norm2[effective_lowband + j]; } wg_sync(); }
                  tr.b = b / 2; tr.lb = effective_lowband != -1 ? 0 : -1; tr.fill = (int)(sl ? y_cm : x_cm);
                  lb_out = sl ? lbo2 : lbo;
               } else {
                  tr.b = b; tr.lb = effective_lowband != -1 ? 0 : -1; tr.fill = (int)cm_in;
                  lb_out = lbo;
               }
               p4_qb_pre(G, tr, qb, tf_change, N);
            }
            const int xo = tr.xo;
            P4_TIC();
            p4_tree_run(L4, G, ecbuf, cfg, tr);
            P4_TOC(24);
            if (s_act) {
               const unsigned cm = p4_qb_post(G, tr, qb, xo, resynth, lb_out, N);
               remaining_bits = tr.remaining_bits; seed = tr.seed;
               if (joint) x_cm = sl == 0 ? cm : (x_cm | cm);
               else if (dual_stereo) { if (sl == 0) x_cm = cm; else y_cm = cm; }
               else { x_cm = cm; y_cm = cm; }
            }
         }
         if (t_act && joint) {
            if (resynth) {
               P4_TIC();
               p4_stereo_merge(G->Xb, G->Yb, mid, N);
               if (inv) { wg_sync(); FOR_GL(j, N) G->Yb[j] = neg32(G->Yb[j]); wg_sync(); }
               P4_TOC(23);
            }
            y_cm = x_cm;
         }
      }
      if (rdo) {
         /* keep the trial with the larger weighted correlation (bands.c:1889-1911); the first one wins ties */
         P4_TIC();
         wg_sync();
         const i32 dist1 = mult16_32_q15(w0, p4_inner_prod_g(Xg, G->Xb, N)) + mult16_32_q15(w1, p4_inner_prod_g(Yg, G->Yb, N));
         const int first = dist0 >= dist1;
         if (first) {
            x_cm = cm2; y_cm = cm2; remaining_bits = rem1; seed = seed1;
            wg_sync();
            FOR_GL(k, (int)(sizeof(EcCtx) / 4)) { ((WV_LDS i32 *)&G->ec)[k] = ((const WV_LDS i32 *)&G->ecsave[1])[k]; }
            wg_sync();
         } else {
            /* the second trial's bytes: the front run from the common start, the raw bits at the tail */
            const int f0 = (int)G->ecsave[0].offs, f1 = (int)G->ec.offs, st = (int)G->ec.storage, e0 = (int)G->ecsave[0].end_offs, e1 = (int)G->ec.end_offs;
            const u8 *alt = cont->alt;
            FOR_GL(k, f1 - f0) pkt[f0 + k] = alt[f0 + k];
            FOR_GL(k, e1 - e0) pkt[st - e1 + k] = alt[st - e1 + k];
         }
         if (!last) { const i32 *src = cont->norm_alt[first ? 0 : 1]; FOR_GL(j, N) lbo[j] = src[j]; }
         P4_TOC(21);
      }
      if (act) {
         wg_sync();
         GLANE0 { G->cmask[i * C + 0] = (u8)x_cm; G->cmask[i * C + C - 1] = (u8)y_cm; }
         balance += G->pulses[i] + tell;
         update_lowband = b > (N << BITRES);
         cfg.avoid_split_noise = 0;
      }
   }
   /* hand the coder back (the bytes are in the image's packet already) */
   if (active) {
      wg_sync();
      FOR_GL(k, (int)(sizeof(EcCtx) / 4)) ((i32 *)&img->ec)[k] = ((const WV_LDS i32 *)&G->ec)[k];
      GLANE0 img->st.rng = seed;
   }
}
#endif

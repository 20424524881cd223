/* fx.h — fixed-point arithmetic and math approximations for the CELT kernels (device side).
 * Integer semantics are those of the reference's fixed-point build: celt/fixed_generic.h:36-218,
 * celt/arch.h:100-224, celt/mathops.c:45-316, celt/mathops.h:352-623 (file:line cited per helper).
 * Included after a wave vocabulary header (wave.h) that defines WV_DEV. */
#ifndef OPUS_AMD_FX_H
#define OPUS_AMD_FX_H
#include <stdint.h>
typedef int16_t  i16;
typedef int32_t  i32;
typedef int64_t  i64;
typedef uint32_t u32;
typedef uint64_t u64;
typedef uint8_t  u8;
typedef int8_t   i8;
#define SIG_SHIFT 12
#define SIG_SAT 536870911
#define NORM_SHIFT 24
#define DB_SHIFT 24
#define Q15ONE 32767
#define Q31ONE 2147483647
#define EPSILON 1
#define BITRES 3

/* compile-time constants: QCONST16/QCONST32/GCONST (fixed_generic.h:95-104) */
#define QC16(x,bits) ((i16)(.5+(x)*(((i32)1)<<(bits))))
#define QC32(x,bits) ((i32)(.5+(x)*(((i64)1)<<(bits))))
#define GC(x) ((i32)(.5+(x)*(((i32)1)<<DB_SHIFT)))

WV_DEV i32 imin(i32 a, i32 b) { return a < b ? a : b; }
WV_DEV i32 imax(i32 a, i32 b) { return a > b ? a : b; }
WV_DEV i32 iabs(i32 a) { return a < 0 ? -a : a; }

/* wrap-around 32-bit add/sub/neg/shl (ADD32_ovflw.. fixed_generic.h:157-166, SHL32 :120) */
WV_DEV i32 add32(i32 a, i32 b) { return (i32)((u32)a + (u32)b); }
WV_DEV i32 sub32(i32 a, i32 b) { return (i32)((u32)a - (u32)b); }
WV_DEV i32 neg32(i32 a) { return (i32)(0u - (u32)a); }
WV_DEV i32 shl32(i32 a, int s) { return (i32)((u32)a << s); }
WV_DEV i32 shr32(i32 a, int s) { return a >> s; }
WV_DEV i32 pshr32(i32 a, int s) { return add32(a, ((i32)1 << s) >> 1) >> s; }   /* PSHR32 :123 */
WV_DEV i32 vshr32(i32 a, int s) { return s > 0 ? (a >> s) : shl32(a, -s); }     /* VSHR32 :125 */
WV_DEV i32 half32(i32 a) { return a >> 1; }
WV_DEV i16 shl16(i32 a, int s) { return (i16)((uint16_t)a << s); }              /* SHL16 :116 */
WV_DEV i16 add16(i32 a, i32 b) { return (i16)((i16)a + (i16)b); }               /* ADD16 :148 */
WV_DEV i32 sub16(i32 a, i32 b) { return (i16)a - (i16)b; }                      /* SUB16 :150 (no truncation) */
WV_DEV i16 extract16(i32 a) { return (i16)a; }
WV_DEV i16 sat16(i32 x) { return x > 32767 ? 32767 : x < -32768 ? -32768 : (i16)x; }
WV_DEV i32 saturate(i32 x, i32 a) { return x > a ? a : x < -a ? -a : x; }       /* SATURATE :134 */
WV_DEV i16 round16(i32 x, int a) { return (i16)pshr32(x, a); }                  /* ROUND16 :139 */
WV_DEV i16 sround16(i32 x, int a) { return (i16)saturate(pshr32(x, a), 32767); }/* SROUND16 :141 */

/* products */
WV_DEV i32 mult16_16(i32 a, i32 b) { return (i32)(i16)a * (i32)(i16)b; }        /* :176 */
WV_DEV i32 mac16_16(i32 c, i32 a, i32 b) { return add32(c, mult16_16(a, b)); }  /* :179 */
WV_DEV i32 mult16_16_q11(i32 a, i32 b) { return mult16_16(a, b) >> 11; }
WV_DEV i32 mult16_16_q13(i32 a, i32 b) { return mult16_16(a, b) >> 13; }
WV_DEV i32 mult16_16_q14(i32 a, i32 b) { return mult16_16(a, b) >> 14; }
WV_DEV i32 mult16_16_q15(i32 a, i32 b) { return mult16_16(a, b) >> 15; }
WV_DEV i32 mult16_16_p13(i32 a, i32 b) { return add32(4096, mult16_16(a, b)) >> 13; }
WV_DEV i32 mult16_16_p14(i32 a, i32 b) { return add32(8192, mult16_16(a, b)) >> 14; }
WV_DEV i32 mult16_16_p15(i32 a, i32 b) { return add32(16384, mult16_16(a, b)) >> 15; }
WV_DEV i32 mult16_32_q15(i32 a, i32 b) { return (i32)(((i64)(i16)a * b) >> 15); } /* :55 */
WV_DEV i32 mult16_32_q16(i32 a, i32 b) { return (i32)(((i64)(i16)a * b) >> 16); } /* :41 */
WV_DEV i32 mult16_32_p16(i32 a, i32 b) { return (i32)((((i64)(i16)a * b) + 32768) >> 16); } /* :48 */
WV_DEV i32 mult32_32_q16(i32 a, i32 b) { return (i32)(((i64)a * (i64)b) >> 16); }
WV_DEV i32 mult32_32_q31(i32 a, i32 b) { return (i32)(((i64)a * (i64)b) >> 31); } /* :69 */
WV_DEV i32 mult32_32_p31(i32 a, i32 b) { return (i32)((1073741824 + (i64)a * (i64)b) >> 31); } /* :76 */
WV_DEV i32 mult32_32_q32(i32 a, i32 b) { return (i32)(((i64)a * (i64)b) >> 32); }
WV_DEV i32 frac_mul16(i32 a, i32 b) { return (16384 + (i32)(i16)a * (i16)b) >> 15; } /* mathops.h:49 */
WV_DEV i32 mac16_32_q15(i32 c, i32 a, i32 b)  /* fixed_generic.h:183 (split form is what the build uses) */
{ return add32(c, add32(mult16_16(a, b >> 15), mult16_16(a, b & 0x7fff) >> 15)); }
WV_DEV i32 mac16_32_q16(i32 c, i32 a, i32 b)  /* :187 */
{ return add32(c, add32(mult16_16(a, b >> 16), ((i32)(i16)a * (i32)(uint16_t)(b & 0xffff)) >> 16)); }

/* celt_coef is 16-bit in this build (arch.h:186-193) */
WV_DEV i32 mult_coef_32(i32 a, i32 b) { return mult16_32_q15(a, b); }
WV_DEV i32 mult_coef(i32 a, i32 b) { return mult16_16_q15(a, b); }
WV_DEV i32 mult_coef_taps(i32 a, i32 b) { return mult16_16_p15(a, b); }

/* signal conversions (arch.h:163-180 with RES_SHIFT 0; fixed_generic.h:208) */
WV_DEV i16 sig2word16(i32 x) { x = pshr32(x, SIG_SHIFT); x = imax(x, -32768); x = imin(x, 32767); return (i16)x; }

/* EC_ILOG: 1+floor(log2(v)), 0 for v==0 (celt/ecintrin.h, entcode.c:41) */
WV_DEV int ec_ilog(u32 v) { return v ? 32 - __builtin_clz(v) : 0; }
WV_DEV int celt_ilog2(i32 x) { return ec_ilog((u32)x) - 1; }                    /* mathops.h:352 */
WV_DEV int celt_zlog2(i32 x) { return x <= 0 ? 0 : celt_ilog2(x); }
/* n / d for operands whose QUOTIENT stays below 2^20 (n < 2^24, d >= 1): the float estimate n * rcp(d) is within a quarter of the true quotient (v_rcp_f32 is good to 1 ulp,
 * n and d are exact as floats), so its truncation is the quotient or one off either way and one remainder test settles it -- exact, and a third shorter a dependent chain
 * than the 32-bit division sequence (reciprocal, Newton step, two corrections) the compiler emits.  The PVQ's control arithmetic (compute_qn, the theta grid, band
 * budgets) divides small numbers by small numbers several times per partition, on a wave that is latency-bound. */
WV_DEV u32 fx_udiv24(u32 n, u32 d)
{
   u32 q = (u32)((float)n * wv_rcpf((float)d));
   const i32 r = (i32)(n - q * d);
   if (r < 0) q--; else if ((u32)r >= d) q++;
   return q;
}
WV_DEV i32 fx_sdiv24(i32 n, i32 d) { const u32 q = fx_udiv24((u32)(n < 0 ? -n : n), (u32)d); return n < 0 ? -(i32)q : (i32)q; }     /* d > 0; truncates towards zero like C */
/* x / b for b a power of two (block counts) */
WV_DEV int fx_div_pow2(int x, int b) { return (int)((u32)x >> (ec_ilog((u32)b) - 1)); }


/* isqrt32: exact floor(sqrt(v)), celt/mathops.c:45 */
WV_DEV unsigned fx_isqrt32(u32 val)
{
   unsigned g = 0;
   int bshift = (ec_ilog(val) - 1) >> 1;
   unsigned b = 1U << bshift;
   do {
      u32 t = (((u32)g << 1) + b) << bshift;
      if (t <= val) { g += b; val -= t; }
      b >>= 1; bshift--;
   } while (bshift >= 0);
   return g;
}

/* celt_rcp_norm16: Q15 normalised reciprocal, mathops.c:243 */
WV_DEV i16 fx_rcp_norm16(i32 x)
{
   i16 r = add16(30840, mult16_16_q15(-15420, x));
   r = (i16)sub16(r, mult16_16_q15(r, add16(mult16_16_q15(r, x), add16(r, -32768))));
   return (i16)sub16(r, add16(1, mult16_16_q15(r, add16(mult16_16_q15(r, x), add16(r, -32768)))));
}

/* celt_rcp_norm32: Q31 in [0.5,1) -> Q30, mathops.c:264 */
WV_DEV i32 fx_rcp_norm32(i32 x)
{
   i32 r = shl32((i32)fx_rcp_norm16((x >> 15) - 32768), 16);
   return sub32(r, add32(shl32(mult32_32_q31(add32(mult32_32_q31(r, x), -1073741824), r), 1), 1));
}

/* celt_rcp: Q15 in, Q16 out, mathops.c:287 */
WV_DEV i32 fx_rcp(i32 x)
{
   int i = celt_ilog2(x);
   i16 r = fx_rcp_norm16(vshr32(x, i - 15) - 32768);
   return vshr32((i32)r, i - 16);
}

/* frac_div32_q29 / frac_div32, mathops.c:70/:87 */
WV_DEV i32 fx_frac_div32_q29(i32 a, i32 b)
{
   int shift = celt_ilog2(b) - 29;
   a = vshr32(a, shift);
   b = vshr32(b, shift);
   i16 rcp = round16(fx_rcp(round16(b, 16)), 3);
   i32 result = mult16_32_q15(rcp, a);
   i32 rem = sub32(pshr32(a, 2), mult32_32_q31(result, b));
   return add32(result, shl32(mult16_32_q15(rcp, rem), 2));
}
WV_DEV i32 fx_frac_div32(i32 a, i32 b)
{
   i32 r = fx_frac_div32_q29(a, b);
   if (r >= 536870912) return 2147483647;
   if (r <= -536870912) return -2147483647;
   return shl32(r, 2);
}

/* celt_rsqrt_norm: Q16 in [0.25,1) -> Q14, mathops.c:98 */
WV_DEV i16 fx_rsqrt_norm(i32 x)
{
   i16 n = (i16)(x - 32768);
   i16 r = add16(23557, mult16_16_q15(n, add16(-13490, mult16_16_q15(n, 6713))));
   i16 r2 = (i16)mult16_16_q15(r, r);
   i16 y = shl16(sub16(add16(mult16_16_q15(r2, n), r2), 16384), 1);
   return add16(r, mult16_16_q15(r, mult16_16_q15(y, sub16(mult16_16_q15(y, 12288), 16384))));
}
/* celt_rsqrt_norm32: Q31 -> Q29, mathops.c:126 */
WV_DEV i32 fx_rsqrt_norm32(i32 x)
{
   i32 r = shl32((i32)fx_rsqrt_norm(x >> 15), 15);
   i32 t = mult32_32_q31(r, r);
   t = mult32_32_q31(1073741824, t);
   t = mult32_32_q31(x, t);
   return shl32(mult32_32_q31(r, sub32(201326592, t)), 4);
}

/* celt_sqrt (QX -> QX/2), mathops.c:140 */
WV_DEV i32 fx_sqrt(i32 x)
{
   const i16 C[6] = {23171, 11574, -2901, 1592, -1002, 336};
   if (x == 0) return 0;
   if (x >= 1073741824) return 32767;
   int k = (celt_ilog2(x) >> 1) - 7;
   x = vshr32(x, 2 * k);
   i16 n = (i16)(x - 32768);
   i32 rt = add32(C[0], mult16_16_q15(n, add16(C[1], mult16_16_q15(n, add16(C[2],
             mult16_16_q15(n, add16(C[3], mult16_16_q15(n, add16(C[4], mult16_16_q15(n, C[5]))))))))));
   return vshr32(rt, 7 - k);
}
/* celt_sqrt32 (Qx -> Q(x/2+16)), mathops.c:164 */
WV_DEV i32 fx_sqrt32(i32 x)
{
   if (x == 0) return 0;
   if (x >= 1073741824) return 2147483647;
   int k = celt_ilog2(x) >> 1;
   i32 xf = vshr32(x, 2 * (k - 14) - 1);
   xf = mult32_32_q31(fx_rsqrt_norm32(xf), xf);
   if (k < 12) return pshr32(xf, 12 - k);
   return shl32(xf, k - 12);
}

/* celt_cos_norm (Q16 period 2^17 -> Q15), mathops.c:198; _celt_cos_pi_2 :184 */
WV_DEV i16 cos_pi_2(i16 x)
{
   i16 x2 = (i16)mult16_16_p15(x, x);
   i32 v = add32(sub16(32767, x2), mult16_16_p15(x2, add32(-7651, mult16_16_p15(x2, add32(8277, mult16_16_p15(-626, x2))))));
   return add16(1, imin(32766, v));
}
WV_DEV i16 fx_cos_norm(i32 x)
{
   x = x & 0x0001ffff;
   if (x > (1 << 16)) x = (1 << 17) - x;
   if (x & 0x00007fff) {
      if (x < (1 << 15)) return cos_pi_2((i16)x);
      return (i16)(-cos_pi_2((i16)(65536 - x)));
   }
   if (x & 0x0000ffff) return 0;
   if (x & 0x0001ffff) return -32767;
   return 32767;
}
/* celt_cos_norm32 (Q30 -> Q31), mathops.c:222 */
WV_DEV i32 fx_cos_norm32(i32 x)
{
   if (iabs(x) == 1 << 30) return 0;
   i32 xs = mult32_32_q31(x, x);
   i32 t = add32(-178761936, mult32_32_q31(xs, 29487206));
   t = add32(544710848, mult32_32_q31(xs, t));
   t = add32(-662336704, mult32_32_q31(xs, t));
   return shl32(add32(134217720, mult32_32_q31(xs, t)), 4);
}

/* celt_log2 (Q14 -> Q10), mathops.h:362 */
WV_DEV i16 fx_log2(i32 x)
{
   const i16 C[5] = {-6801 + (1 << (13 - 10)), 15746, -5217, 2545, -1401};
   if (x == 0) return -32767;
   int i = celt_ilog2(x);
   i16 n = (i16)(vshr32(x, i - 15) - 32768 - 16384);
   i16 frac = add16(C[0], mult16_16_q15(n, add16(C[1], mult16_16_q15(n, add16(C[2], mult16_16_q15(n, add16(C[3], mult16_16_q15(n, C[4]))))))));
   return (i16)(shl32(i - 13, 10) + (frac >> (14 - 10)));
}
/* celt_exp2_frac / celt_exp2 (Q10 -> Q16), mathops.h:383/:395 */
WV_DEV i32 fx_exp2_frac(i32 x)
{
   i16 frac = shl16(x, 4);
   return add16(16383, mult16_16_q15(frac, add16(22804, mult16_16_q15(frac, add16(14819, mult16_16_q15(10204, frac))))));
}
WV_DEV i32 fx_exp2(i32 x)
{
   int integer = (i16)x >> 10;
   if (integer > 14) return 0x7f000000;
   if (integer < -15) return 0;
   i16 frac = (i16)fx_exp2_frac((i16)((i16)x - shl16(integer, 10)));
   return vshr32((i32)frac, -integer - 2);
}
/* non-QEXT DB forms, mathops.h:520-522 */
WV_DEV i32 fx_log2_db(i32 x) { return shl32((i32)fx_log2(x), DB_SHIFT - 10); }
WV_DEV i32 fx_exp2_db_frac(i32 x) { return shl32(fx_exp2_frac(pshr32(x, DB_SHIFT - 10)), 14); }
WV_DEV i32 fx_exp2_db(i32 x) { return fx_exp2(pshr32(x, DB_SHIFT - 10)); }

/* celt_atan_norm / celt_atan2p_norm (Q30), mathops.h:537/:585 */
WV_DEV i32 fx_atan_norm(i32 x)
{
   if (x == 1073741824) return 536870912;
   if (x == -1073741824) return -536870912;
   i32 xq31 = shl32(x, 1);
   i32 xs = mult32_32_q31(xq31, x);
   i32 t = mult32_32_q31(xs, -598602432);
   t = mult32_32_q31(xs, add32(1583306112, t));
   t = mult32_32_q31(xs, add32(-1985085440, t));
   t = mult32_32_q31(xs, add32(1682636672, t));
   t = mult32_32_q31(xs, add32(-1200579328, t));
   t = mult32_32_q31(xs, add32(857391616, t));
   t = mult32_32_q31(xs, add32(-715791936, t));
   t = add32(x, mult32_32_q31(xq31, t));
   return mult32_32_q31(1367130551, t);
}
WV_DEV i32 fx_atan2p_norm(i32 y, i32 x)
{
   if (y == 0 && x == 0) return 0;
   if (y < x) return fx_atan_norm(fx_frac_div32(y, x) >> 1);
   return 1073741824 - fx_atan_norm(fx_frac_div32(x, y) >> 1);
}

WV_DEV i32 fx_div(i32 a, i32 b) { return mult32_32_q31(a, fx_rcp(b)); }
#endif

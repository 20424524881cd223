/* oc_math.c — fixed-point transcendental approximations (oracle restatement).
 * Follows celt/mathops.c:45-316 and celt/mathops.h:362-403,520-523 (non-QEXT forms). */
#include "oc_celt.h"

/* isqrt32: exact floor(sqrt(v)), celt/mathops.c:45 */
unsigned oc_isqrt32(u32 val)
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
i16 oc_rcp_norm16(i32 x)
{
   i16 r = add16(30840, mult16_16_q15(-15420, x));
   r = (i16)sub16(r, mult16_16_q15(r, add16(mult16_16_q15(r, x), add16(r, -32768))));
   return (i16)sub16(r, add16(1, mult16_16_q15(r, add16(mult16_16_q15(r, x), add16(r, -32768)))));
}

/* celt_rcp_norm32: Q31 in [0.5,1) -> Q30, mathops.c:264 */
i32 oc_rcp_norm32(i32 x)
{
   i32 r = shl32((i32)oc_rcp_norm16((x >> 15) - 32768), 16);
   return sub32(r, add32(shl32(mult32_32_q31(add32(mult32_32_q31(r, x), -1073741824), r), 1), 1));
}

/* celt_rcp: Q15 in, Q16 out, mathops.c:287 */
i32 oc_rcp(i32 x)
{
   int i = celt_ilog2(x);
   i16 r = oc_rcp_norm16(vshr32(x, i - 15) - 32768);
   return vshr32((i32)r, i - 16);
}

/* frac_div32_q29 / frac_div32, mathops.c:70/:87 */
i32 oc_frac_div32_q29(i32 a, i32 b)
{
   int shift = celt_ilog2(b) - 29;
   a = vshr32(a, shift);
   b = vshr32(b, shift);
   i16 rcp = round16(oc_rcp(round16(b, 16)), 3);
   i32 result = mult16_32_q15(rcp, a);
   i32 rem = sub32(pshr32(a, 2), mult32_32_q31(result, b));
   return add32(result, shl32(mult16_32_q15(rcp, rem), 2));
}
i32 oc_frac_div32(i32 a, i32 b)
{
   i32 r = oc_frac_div32_q29(a, b);
   if (r >= 536870912) return 2147483647;
   if (r <= -536870912) return -2147483647;
   return shl32(r, 2);
}

/* celt_rsqrt_norm: Q16 in [0.25,1) -> Q14, mathops.c:98 */
i16 oc_rsqrt_norm(i32 x)
{
   i16 n = (i16)(x - 32768);
   i16 r = add16(23557, mult16_16_q15(n, add16(-13490, mult16_16_q15(n, 6713))));
   i16 r2 = (i16)mult16_16_q15(r, r);
   i16 y = shl16(sub16(add16(mult16_16_q15(r2, n), r2), 16384), 1);
   return add16(r, mult16_16_q15(r, mult16_16_q15(y, sub16(mult16_16_q15(y, 12288), 16384))));
}
/* celt_rsqrt_norm32: Q31 -> Q29, mathops.c:126 */
i32 oc_rsqrt_norm32(i32 x)
{
   i32 r = shl32((i32)oc_rsqrt_norm(x >> 15), 15);
   i32 t = mult32_32_q31(r, r);
   t = mult32_32_q31(1073741824, t);
   t = mult32_32_q31(x, t);
   return shl32(mult32_32_q31(r, sub32(201326592, t)), 4);
}

/* celt_sqrt (QX -> QX/2), mathops.c:140 */
i32 oc_sqrt(i32 x)
{
   static const i16 C[6] = {23171, 11574, -2901, 1592, -1002, 336};
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
i32 oc_sqrt32(i32 x)
{
   if (x == 0) return 0;
   if (x >= 1073741824) return 2147483647;
   int k = celt_ilog2(x) >> 1;
   i32 xf = vshr32(x, 2 * (k - 14) - 1);
   xf = mult32_32_q31(oc_rsqrt_norm32(xf), xf);
   if (k < 12) return pshr32(xf, 12 - k);
   return shl32(xf, k - 12);
}

/* celt_cos_norm (Q16 period 2^17 -> Q15), mathops.c:198; _celt_cos_pi_2 :184 */
static i16 cos_pi_2(i16 x)
{
   i16 x2 = (i16)mult16_16_p15(x, x);
   i32 v = add32(sub16(32767, x2), mult16_16_p15(x2, add32(-7651, mult16_16_p15(x2, add32(8277, mult16_16_p15(-626, x2))))));
   return add16(1, imin(32766, v));
}
i16 oc_cos_norm(i32 x)
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
i32 oc_cos_norm32(i32 x)
{
   if (iabs(x) == 1 << 30) return 0;
   i32 xs = mult32_32_q31(x, x);
   i32 t = add32(-178761936, mult32_32_q31(xs, 29487206));
   t = add32(544710848, mult32_32_q31(xs, t));
   t = add32(-662336704, mult32_32_q31(xs, t));
   return shl32(add32(134217720, mult32_32_q31(xs, t)), 4);
}

/* celt_log2 (Q14 -> Q10), mathops.h:362 */
i16 oc_log2(i32 x)
{
   static const i16 C[5] = {-6801 + (1 << (13 - 10)), 15746, -5217, 2545, -1401};
   if (x == 0) return -32767;
   int i = celt_ilog2(x);
   i16 n = (i16)(vshr32(x, i - 15) - 32768 - 16384);
   i16 frac = add16(C[0], mult16_16_q15(n, add16(C[1], mult16_16_q15(n, add16(C[2], mult16_16_q15(n, add16(C[3], mult16_16_q15(n, C[4]))))))));
   return (i16)(shl32(i - 13, 10) + (frac >> (14 - 10)));
}
/* celt_exp2_frac / celt_exp2 (Q10 -> Q16), mathops.h:383/:395 */
i32 oc_exp2_frac(i32 x)
{
   i16 frac = shl16(x, 4);
   return add16(16383, mult16_16_q15(frac, add16(22804, mult16_16_q15(frac, add16(14819, mult16_16_q15(10204, frac))))));
}
i32 oc_exp2(i32 x)
{
   int integer = (i16)x >> 10;
   if (integer > 14) return 0x7f000000;
   if (integer < -15) return 0;
   i16 frac = (i16)oc_exp2_frac((i16)((i16)x - shl16(integer, 10)));
   return vshr32((i32)frac, -integer - 2);
}
/* non-QEXT DB forms, mathops.h:520-522 */
i32 oc_log2_db(i32 x) { return shl32((i32)oc_log2(x), DB_SHIFT - 10); }
i32 oc_exp2_db_frac(i32 x) { return shl32(oc_exp2_frac(pshr32(x, DB_SHIFT - 10)), 14); }
i32 oc_exp2_db(i32 x) { return oc_exp2(pshr32(x, DB_SHIFT - 10)); }

/* celt_atan_norm / celt_atan2p_norm (Q30), mathops.h:537/:585 */
i32 oc_atan_norm(i32 x)
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
i32 oc_atan2p_norm(i32 y, i32 x)
{
   if (y == 0 && x == 0) return 0;
   if (y < x) return oc_atan_norm(oc_frac_div32(y, x) >> 1);
   return 1073741824 - oc_atan_norm(oc_frac_div32(x, y) >> 1);
}

/* oc_arith.h — fixed-point arithmetic vocabulary of the CPU oracle.
 *
 * TEST INFRASTRUCTURE: this directory is the CPU restatement ("oracle") of the
 * xiph/opus fixed-point CELT path.  Only tests/, __graft_entry__.smoke() and
 * bench.py's cpu_baseline leg may use it; the product (opus_amd/) never does.
 *
 * Every helper states the integer semantics of the reference macro it mirrors
 * (celt/fixed_generic.h:36-218, celt/arch.h:100-224, celt/mathops.h:49,
 * celt/ecintrin.h EC_ILOG).  Two's complement, arithmetic >>, wrap-around adds via
 * unsigned — the implementation-defined behaviours the reference relies on.
 */
#ifndef OC_ARITH_H
#define OC_ARITH_H
#include <stdint.h>
#include <string.h>

typedef int16_t  i16;
typedef int32_t  i32;
typedef int64_t  i64;
typedef uint32_t u32;
typedef uint8_t  u8;

#define OC_INLINE static inline __attribute__((always_inline))

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

OC_INLINE i32 imin(i32 a, i32 b) { return a < b ? a : b; }
OC_INLINE i32 imax(i32 a, i32 b) { return a > b ? a : b; }
OC_INLINE i32 iabs(i32 a) { return a < 0 ? -a : a; }

/* wrap-around 32-bit add/sub/neg/shl (ADD32_ovflw.. fixed_generic.h:157-166, SHL32 :120) */
OC_INLINE i32 add32(i32 a, i32 b) { return (i32)((u32)a + (u32)b); }
OC_INLINE i32 sub32(i32 a, i32 b) { return (i32)((u32)a - (u32)b); }
OC_INLINE i32 neg32(i32 a) { return (i32)(0u - (u32)a); }
OC_INLINE i32 shl32(i32 a, int s) { return (i32)((u32)a << s); }
OC_INLINE i32 shr32(i32 a, int s) { return a >> s; }
OC_INLINE i32 pshr32(i32 a, int s) { return add32(a, ((i32)1 << s) >> 1) >> s; }   /* PSHR32 :123 */
OC_INLINE i32 vshr32(i32 a, int s) { return s > 0 ? (a >> s) : shl32(a, -s); }     /* VSHR32 :125 */
OC_INLINE i32 half32(i32 a) { return a >> 1; }
OC_INLINE i16 shl16(i32 a, int s) { return (i16)((uint16_t)a << s); }              /* SHL16 :116 */
OC_INLINE i16 add16(i32 a, i32 b) { return (i16)((i16)a + (i16)b); }               /* ADD16 :148 */
OC_INLINE i32 sub16(i32 a, i32 b) { return (i16)a - (i16)b; }                      /* SUB16 :150 (no truncation) */
OC_INLINE i16 extract16(i32 a) { return (i16)a; }
OC_INLINE i16 sat16(i32 x) { return x > 32767 ? 32767 : x < -32768 ? -32768 : (i16)x; }
OC_INLINE i32 saturate(i32 x, i32 a) { return x > a ? a : x < -a ? -a : x; }       /* SATURATE :134 */
OC_INLINE i16 round16(i32 x, int a) { return (i16)pshr32(x, a); }                  /* ROUND16 :139 */
OC_INLINE i16 sround16(i32 x, int a) { return (i16)saturate(pshr32(x, a), 32767); }/* SROUND16 :141 */

/* products */
OC_INLINE i32 mult16_16(i32 a, i32 b) { return (i32)(i16)a * (i32)(i16)b; }        /* :176 */
OC_INLINE i32 mac16_16(i32 c, i32 a, i32 b) { return add32(c, mult16_16(a, b)); }  /* :179 */
OC_INLINE i32 mult16_16_q11(i32 a, i32 b) { return mult16_16(a, b) >> 11; }
OC_INLINE i32 mult16_16_q13(i32 a, i32 b) { return mult16_16(a, b) >> 13; }
OC_INLINE i32 mult16_16_q14(i32 a, i32 b) { return mult16_16(a, b) >> 14; }
OC_INLINE i32 mult16_16_q15(i32 a, i32 b) { return mult16_16(a, b) >> 15; }
OC_INLINE i32 mult16_16_p13(i32 a, i32 b) { return add32(4096, mult16_16(a, b)) >> 13; }
OC_INLINE i32 mult16_16_p14(i32 a, i32 b) { return add32(8192, mult16_16(a, b)) >> 14; }
OC_INLINE i32 mult16_16_p15(i32 a, i32 b) { return add32(16384, mult16_16(a, b)) >> 15; }
OC_INLINE i32 mult16_32_q15(i32 a, i32 b) { return (i32)(((i64)(i16)a * b) >> 15); } /* :55 */
OC_INLINE i32 mult16_32_q16(i32 a, i32 b) { return (i32)(((i64)(i16)a * b) >> 16); } /* :41 */
OC_INLINE i32 mult16_32_p16(i32 a, i32 b) { return (i32)((((i64)(i16)a * b) + 32768) >> 16); } /* :48 */
OC_INLINE i32 mult32_32_q16(i32 a, i32 b) { return (i32)(((i64)a * (i64)b) >> 16); }
OC_INLINE i32 mult32_32_q31(i32 a, i32 b) { return (i32)(((i64)a * (i64)b) >> 31); } /* :69 */
OC_INLINE i32 mult32_32_p31(i32 a, i32 b) { return (i32)((1073741824 + (i64)a * (i64)b) >> 31); } /* :76 */
OC_INLINE i32 mult32_32_q32(i32 a, i32 b) { return (i32)(((i64)a * (i64)b) >> 32); }
OC_INLINE i32 frac_mul16(i32 a, i32 b) { return (16384 + (i32)(i16)a * (i16)b) >> 15; } /* mathops.h:49 */
OC_INLINE i32 mac16_32_q15(i32 c, i32 a, i32 b)  /* fixed_generic.h:183 (split form is what the build uses) */
{ return add32(c, add32(mult16_16(a, b >> 15), mult16_16(a, b & 0x7fff) >> 15)); }
OC_INLINE i32 mac16_32_q16(i32 c, i32 a, i32 b)  /* :187 */
{ return add32(c, add32(mult16_16(a, b >> 16), ((i32)(i16)a * (i32)(uint16_t)(b & 0xffff)) >> 16)); }

/* celt_coef is 16-bit in this build (arch.h:186-193) */
OC_INLINE i32 mult_coef_32(i32 a, i32 b) { return mult16_32_q15(a, b); }
OC_INLINE i32 mult_coef(i32 a, i32 b) { return mult16_16_q15(a, b); }
OC_INLINE i32 mult_coef_taps(i32 a, i32 b) { return mult16_16_p15(a, b); }

/* signal conversions (arch.h:163-180 with RES_SHIFT 0; fixed_generic.h:208) */
OC_INLINE i16 sig2word16(i32 x) { x = pshr32(x, SIG_SHIFT); x = imax(x, -32768); x = imin(x, 32767); return (i16)x; }

/* EC_ILOG: 1+floor(log2(v)), 0 for v==0 (celt/ecintrin.h, entcode.c:41) */
OC_INLINE int ec_ilog(u32 v) { return v ? 32 - __builtin_clz(v) : 0; }
OC_INLINE int celt_ilog2(i32 x) { return ec_ilog((u32)x) - 1; }                    /* mathops.h:352 */
OC_INLINE int celt_zlog2(i32 x) { return x <= 0 ? 0 : celt_ilog2(x); }

#endif

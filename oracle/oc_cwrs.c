/* oc_cwrs.c — PVQ codeword <-> index (combinatorial numbering), oracle restatement of
 * celt/cwrs.c:199-204 (U/V accessors), :444-465 (icwrs/encode_pulses). */
#include "oc_celt.h"
#include <stdlib.h>

static u32 pvq_u(int n, int k)
{
   int lo = n < k ? n : k, hi = n < k ? k : n;
   return oc_pvq_u_data[oc_pvq_u_row[lo] + hi];
}
u32 oc_pvq_v(int n, int k) { return pvq_u(n, k) + pvq_u(n, k + 1); }

/* icwrs, cwrs.c:444 */
static u32 icwrs(int n, const int *y)
{
   int j = n - 1;
   u32 i = y[j] < 0;
   int k = abs(y[j]);
   do {
      j--;
      i += pvq_u(n - j, k);
      k += abs(y[j]);
      if (y[j] < 0) i += pvq_u(n - j, k + 1);
   } while (j > 0);
   return i;
}
void oc_encode_pulses(const int *y, int n, int k, oc_ec *enc)
{
   oc_ec_enc_uint(enc, icwrs(n, y), oc_pvq_v(n, k));
}

/* cwrsi, cwrs.c:467 — index -> pulse vector; returns yy = sum y^2.  The reference's row pointers CELT_PVQ_U_ROW[a][b]
 * are U(a,b) with a <= b, which pvq_u() resolves. */
static i32 cwrsi(int n, int k, u32 i, int *y)
{
   u32 p;
   int s, k0;
   i16 val;
   i32 yy = 0;
   while (n > 2) {
      u32 q;
      if (k >= n) {                                 /* lots of pulses */
         p = pvq_u(n, k + 1);
         s = -(i >= p);
         i -= p & s;
         k0 = k;
         q = pvq_u(n, n);
         if (q > i) {
            k = n;
            do p = pvq_u(--k, n); while (p > i);
         } else for (p = pvq_u(n, k); p > i; p = pvq_u(n, k)) k--;
         i -= p;
         val = (i16)((k0 - k + s) ^ s);
         *y++ = val;
         yy = mac16_16(yy, val, val);
      } else {                                      /* lots of dimensions */
         p = pvq_u(k, n);
         q = pvq_u(k + 1, n);
         if (p <= i && i < q) {
            i -= p;
            *y++ = 0;
         } else {
            s = -(i >= q);
            i -= q & s;
            k0 = k;
            do p = pvq_u(--k, n); while (p > i);
            i -= p;
            val = (i16)((k0 - k + s) ^ s);
            *y++ = val;
            yy = mac16_16(yy, val, val);
         }
      }
      n--;
   }
   p = 2 * k + 1;                                    /* n == 2 */
   s = -(i >= p);
   i -= p & s;
   k0 = k;
   k = (i + 1) >> 1;
   if (k) i -= 2 * k - 1;
   val = (i16)((k0 - k + s) ^ s);
   *y++ = val;
   yy = mac16_16(yy, val, val);
   s = -(int)i;                                      /* n == 1 */
   val = (i16)((k + s) ^ s);
   *y = val;
   yy = mac16_16(yy, val, val);
   return yy;
}
/* decode_pulses, cwrs.c:543 */
i32 oc_decode_pulses(int *y, int n, int k, oc_ec *dec) { return cwrsi(n, k, oc_ec_dec_uint(dec, oc_pvq_v(n, k)), y); }

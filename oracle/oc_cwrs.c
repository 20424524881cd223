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

/* tests/gpu_wave_prims2.hip -- TEST INFRASTRUCTURE (MI355X): the row-limited reductions and the float-pre-filtered arg-max of opus_amd/csrc/wave.h against plain loops on the host:
 * wv_sum_n / wv_sum64_n for every lane count 1..64, wv_argmax_ratio_fast on ratios with exact ties, near ties (cross products that differ by one) and invalid lanes. */
#include "wave.h"
#include <stdio.h>
#include <stdlib.h>
__global__ void __launch_bounds__(64) k(const int *in, const long long *in64, const unsigned *num, const unsigned *den, const int *nl, int *out, long long *out64)
{
   const int l = threadIdx.x, b = blockIdx.x, n = __builtin_amdgcn_readfirstlane(nl[b]);
   const int v = l < n ? in[b * 64 + l] : 0;
   const long long w = l < n ? in64[b * 64 + l] : 0;
   out[b * 4 + 0] = wv_sum_n(v, n);
   out64[b] = wv_sum64_n(w, n);
   const bool valid = l < n && den[b * 64 + l] != 0;
   out[b * 4 + 1] = wv_argmax_ratio_fast(valid ? num[b * 64 + l] : 0u, valid ? den[b * 64 + l] : 1u, valid, n);
   out[b * 4 + 2] = wv_argmax_ratio_packed(valid ? num[b * 64 + l] : 0u, valid ? den[b * 64 + l] : 1u, valid, n);
}
int main() {
   const int NB = 20000;
   int *h = (int *)malloc(NB * 64 * 4), *hn = (int *)malloc(NB * 4); long long *h64 = (long long *)malloc(NB * 64 * 8); unsigned *hnum = (unsigned *)malloc(NB * 64 * 4), *hden = (unsigned *)malloc(NB * 64 * 4);
   srand(7);
   for (int b = 0; b < NB; b++) {
      hn[b] = 1 + rand() % 64;
      const int mode = b % 5;
      for (int l = 0; l < 64; l++) {
         const int i = b * 64 + l;
         h[i] = (rand() << 8) ^ rand() ^ (rand() << 20); h64[i] = ((long long)h[i] << 22) ^ rand();
         unsigned nu = rand() & 0x7fff, de = 1 + (rand() & 0x7ffe);
         if (mode == 1) { nu = rand() % 8; de = 1 + rand() % 4; }                               /* many exact ties */
         if (mode == 2) { de = 1000 + rand() % 3; nu = 3 * de + rand() % 2; }                   /* near ties around 3 */
         if (mode == 3) { const unsigned kq = 1 + rand() % 50; nu = 7 * kq; de = 11 * kq; }     /* one ratio in many forms */
         if (mode == 4 && rand() % 3 == 0) de = 0;                                              /* invalid lanes */
         hnum[i] = nu; hden[i] = de;
      }
      if (mode == 4) hden[b * 64 + rand() % hn[b]] = 1 + rand() % 100;                          /* at least one valid lane */
   }
   int *d, *dn, *dout; long long *d64, *dout64; unsigned *dnum, *dden;
   hipMalloc(&d, NB * 64 * 4); hipMalloc(&dn, NB * 4); hipMalloc(&dout, NB * 4 * 4); hipMalloc(&d64, NB * 64 * 8); hipMalloc(&dout64, NB * 8); hipMalloc(&dnum, NB * 64 * 4); hipMalloc(&dden, NB * 64 * 4);
   hipMemcpy(d, h, NB * 64 * 4, hipMemcpyHostToDevice); hipMemcpy(dn, hn, NB * 4, hipMemcpyHostToDevice); hipMemcpy(d64, h64, NB * 64 * 8, hipMemcpyHostToDevice);
   hipMemcpy(dnum, hnum, NB * 64 * 4, hipMemcpyHostToDevice); hipMemcpy(dden, hden, NB * 64 * 4, hipMemcpyHostToDevice);
   k<<<NB, 64>>>(d, d64, dnum, dden, dn, dout, dout64);
   int *o = (int *)malloc(NB * 4 * 4); long long *o64 = (long long *)malloc(NB * 8);
   hipMemcpy(o, dout, NB * 4 * 4, hipMemcpyDeviceToHost); hipMemcpy(o64, dout64, NB * 8, hipMemcpyDeviceToHost);
   int bad = 0;
   for (int b = 0; b < NB; b++) {
      unsigned s = 0; unsigned long long s64 = 0; int best = -1;
      for (int l = 0; l < hn[b]; l++) {
         const int i = b * 64 + l; s += (unsigned)h[i]; s64 += (unsigned long long)h64[i];
         if (hden[i] == 0) continue;
         if (best < 0 || (unsigned long long)hden[b * 64 + best] * hnum[i] > (unsigned long long)hden[i] * hnum[b * 64 + best]) best = l;
      }
      if (o[b * 4] != (int)s || o64[b] != (long long)s64 || o[b * 4 + 1] != best || o[b * 4 + 2] != best) {
         if (bad < 5) printf("block %d (n %d, mode %d): sum %d/%d s64 %lld/%lld argmax fast %d packed %d want %d\n", b, hn[b], b % 5, o[b * 4], (int)s, o64[b], (long long)s64, o[b * 4 + 1], o[b * 4 + 2], best);
         bad++;
      }
   }
   printf("wave primitives 2: %s (%d bad of %d)\n", bad ? "BAD" : "ok", bad, NB);
   return bad != 0;
}

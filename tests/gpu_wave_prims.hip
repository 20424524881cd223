#include "wave.h"
#include <stdio.h>
#include <stdlib.h>
__global__ void __launch_bounds__(64) k(const int *in, const long long *in64, int *out, long long *out64)
{
   int l = threadIdx.x, b = blockIdx.x;
   int v = in[b * 64 + l];
   out[b * 8 + 0] = wv_sum(v);
   out[b * 8 + 1] = wv_max(v);
   out[b * 8 + 2] = wv_min(v);
   out[b * 8 + 3] = (int)wv_or((unsigned)v);
   int num = (v >> 3) & 0x7fff, den = 1 + (v & 0x3ff), idx = l * 3 + (v & 1);
   if ((v & 0xf000) == 0) { num = -1; den = 1; idx = 0x7fffffff; }
   wv_argmax_ratio(num, den, idx);
   out[b * 8 + 4] = num; out[b * 8 + 5] = den; out[b * 8 + 6] = idx;
   out[b * 8 + 7] = wv_bcast(v, (b * 7) & 63);
   out64[b] = wv_sum64(in64[b * 64 + l]);
}
int main() {
   const int NB = 2000; int *h = (int*)malloc(NB*64*4); long long *h64 = (long long*)malloc(NB*64*8);
   srand(1); for (int i = 0; i < NB*64; i++) { h[i] = (rand() << 8) ^ rand() ^ (rand() << 20); if (i % 977 == 0) h[i] = 0; h64[i] = ((long long)h[i] << 20) ^ rand(); }
   int *d, *dout; long long *d64, *dout64; hipMalloc(&d, NB*64*4); hipMalloc(&dout, NB*8*4); hipMalloc(&d64, NB*64*8); hipMalloc(&dout64, NB*8);
   hipMemcpy(d, h, NB*64*4, hipMemcpyHostToDevice); hipMemcpy(d64, h64, NB*64*8, hipMemcpyHostToDevice);
   k<<<NB, 64>>>(d, d64, dout, dout64);
   int *o = (int*)malloc(NB*8*4); long long *o64 = (long long*)malloc(NB*8); hipMemcpy(o, dout, NB*8*4, hipMemcpyDeviceToHost); hipMemcpy(o64, dout64, NB*8, hipMemcpyDeviceToHost);
   int bad = 0;
   for (int b = 0; b < NB; b++) {
      unsigned s = 0; int mx = -2147483647-1, mn = 2147483647; unsigned orv = 0; long long s64 = 0;
      int bn = 0, bd = 0, bi = 0; bool have = false;
      for (int l = 0; l < 64; l++) { int v = h[b*64+l]; s += (unsigned)v; if (v > mx) mx = v; if (v < mn) mn = v; orv |= (unsigned)v; s64 += h64[b*64+l];
         int num = (v >> 3) & 0x7fff, den = 1 + (v & 0x3ff), idx = l * 3 + (v & 1);
         if ((v & 0xf000) == 0) { num = -1; den = 1; idx = 0x7fffffff; }
         if (!have) { bn = num; bd = den; bi = idx; have = true; }
         else { int lhs = (int)(short)bd * (int)(short)num, rhs = (int)(short)den * (int)(short)bn; if (lhs > rhs || (lhs == rhs && idx < bi)) { bn = num; bd = den; bi = idx; } } }
      bool mn_ok = (mn == -2147483647-1) || o[b*8+2] == mn;
      if (o[b*8+0] != (int)s || o[b*8+1] != mx || !mn_ok || o[b*8+3] != (int)orv || o[b*8+4] != bn || o[b*8+5] != bd || o[b*8+6] != bi || o[b*8+7] != h[b*64 + ((b*7)&63)] || o64[b] != s64) {
         if (bad < 5) printf("block %d mismatch: sum %d/%d max %d/%d min %d/%d or %x/%x argmax (%d,%d,%d)/(%d,%d,%d) s64 %lld/%lld\n", b, o[b*8], (int)s, o[b*8+1], mx, o[b*8+2], mn, o[b*8+3], orv, o[b*8+4], o[b*8+5], o[b*8+6], bn, bd, bi, o64[b], s64);
         bad++; }
   }
   printf("wave primitives: %s (%d bad of %d)\n", bad ? "BAD" : "ok", bad, NB);
   return bad != 0;
}

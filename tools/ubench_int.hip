// tools/ubench_int.hip — instruction-rate microbenchmark for the integer ops the SILK kernels lean on (gfx950).
// One wave per block, N dependent or independent ops, cycles from s_memtime.  Output: cycles per instruction per wave.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#define REP 256
template <int MODE> __global__ __launch_bounds__(64) void k(int32_t *out, int32_t seed, long long *cyc)
{
   int32_t a[8]; for (int j = 0; j < 8; j++) a[j] = seed + threadIdx.x * 7 + j;
   int32_t b = seed * 3 + 1;
   long long t0 = __builtin_amdgcn_s_memtime();
   #pragma unroll 1
   for (int r = 0; r < REP; r++) {
      #pragma unroll
      for (int u = 0; u < 4; u++) {
         #pragma unroll
         for (int j = 0; j < 8; j++) {
            if (MODE == 0) a[j] = __mulhi(a[j], b) + 1;                                  // v_mul_hi_i32 + add
            if (MODE == 1) a[j] = a[j] * b + 1;                                          // v_mul_lo_u32 (+add / mad_u64?)
            if (MODE == 2) a[j] = __mul24(a[j], b) + 1;                                  // v_mad_i32_i24
            if (MODE == 3) a[j] = a[j] + b;                                              // v_add
            if (MODE == 4) a[j] = __builtin_amdgcn_ds_bpermute((threadIdx.x ^ 1) * 4, a[j]);   // ds_bpermute
            if (MODE == 5) a[j] = __builtin_amdgcn_update_dpp(0, a[j], 0xB1, 0xf, 0xf, false) + 1;  // dpp quad_perm mov + add
            if (MODE == 6) a[j] = (int32_t)(((int64_t)a[j] * (int16_t)b) >> 16) + 1;     // SMLAWB as the compiler sees it
            if (MODE == 7) { int32_t hi = (a[j] >> 16) * (int16_t)b; int32_t lo = ((a[j] & 0xffff) * (int16_t)b) >> 16; a[j] = hi + lo + 1; }  // split form
         }
      }
   }
   long long t1 = __builtin_amdgcn_s_memtime();
   int32_t s = 0; for (int j = 0; j < 8; j++) s += a[j];
   out[blockIdx.x * 64 + threadIdx.x] = s;
   if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
template <int MODE> void run(const char *name, int blocks, int32_t *d_out, long long *d_cyc)
{
   hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(64), 0, 0, d_out, 12345, d_cyc);
   hipDeviceSynchronize();
   hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
   hipEventRecord(e0); hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(64), 0, 0, d_out, 12345, d_cyc); hipEventRecord(e1); hipEventSynchronize(e1);
   float ms; hipEventElapsedTime(&ms, e0, e1);
   long long c[4]; hipMemcpy(c, d_cyc, sizeof c, hipMemcpyDeviceToHost);
   double n = (double)REP * 4 * 8;
   printf("%-28s blocks=%5d  memtime ticks/op (wave0) = %.2f   wall ns/op/wave = %.3f\n", name, blocks, (double)c[0] / n, ms * 1e6 / n / ((blocks + 1023) / 1024));
}
int main()
{
   int32_t *d_out; long long *d_cyc; hipMalloc(&d_out, 8192 * 64 * 4); hipMalloc(&d_cyc, 8192 * 8);
   for (int blocks : {1024, 2048, 4096}) {
      run<0>("mul_hi_i32+add", blocks, d_out, d_cyc);  run<1>("mul_lo+add", blocks, d_out, d_cyc);  run<2>("mul24+add", blocks, d_out, d_cyc);
      run<3>("add", blocks, d_out, d_cyc);             run<4>("ds_bpermute", blocks, d_out, d_cyc); run<5>("dpp quad_perm+add", blocks, d_out, d_cyc);
      run<6>("smlawb (i64 form)", blocks, d_out, d_cyc); run<7>("smlawb (split form)", blocks, d_out, d_cyc);
   }
   return 0;
}

// tools/pmc_calibrate.hip — known-byte-count kernels in the encoder's own access pattern (4 B per lane, 256 B per wave request)
// to calibrate rocprofv3's FETCH_SIZE / WRITE_SIZE on gfx950 (MI355X_MICROARCH.md: only 16 B/lane streaming reads are pre-calibrated).
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void calib_read4(const int *src, int *sink, size_t n)
{
   size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x, stride = (size_t)gridDim.x * blockDim.x;
   int acc = 0;
   for (; i < n; i += stride) acc ^= src[i];
   if (acc == 0x7fffffff) sink[0] = acc;          // never true for the fill pattern: keeps the loads alive without writing
}
__global__ void calib_write4(int *dst, size_t n)
{
   size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x, stride = (size_t)gridDim.x * blockDim.x;
   for (; i < n; i += stride) dst[i] = (int)i;
}
int main()
{
   const size_t n = (size_t)1 << 28;               // 1 GiB of int32: far beyond L2 + Infinity Cache
   int *a, *sink;
   if (hipMalloc(&a, n * 4) != hipSuccess || hipMalloc(&sink, 4) != hipSuccess) return 1;
   hipMemset(a, 1, n * 4);
   hipDeviceSynchronize();
   calib_write4<<<4096, 256>>>(a, n);
   hipDeviceSynchronize();
   calib_read4<<<4096, 256>>>(a, sink, n);
   hipDeviceSynchronize();
   printf("bytes_per_kernel %zu\n", n * 4);
   return 0;
}

/* tests/emu/hip_stub.h — TEST INFRASTRUCTURE.  Just enough of the HIP runtime API for opus_amd/csrc/opus_amd.hip (host side + kernels) to be
 * compiled by g++ on top of the CPU wave emulator (wave_emu.h): "device" memory is host memory, a kernel launch runs the kernel function once per
 * workgroup on 64 fibers.  This is how the reference's own C test programs are linked against the complete C ABI in the GPU-less container
 * (tests/hostemu.py).  The product never includes this file and has no CPU path. */
#ifndef HIP_STUB_H
#define HIP_STUB_H
#include "wave_emu.h"
#include <functional>
#include <chrono>
#include <mutex>

typedef int hipError_t;
enum { hipSuccess = 0, hipErrorUnknown = 999 };
enum hipMemcpyKind { hipMemcpyHostToHost, hipMemcpyHostToDevice, hipMemcpyDeviceToHost, hipMemcpyDeviceToDevice, hipMemcpyDefault };
enum { hipFuncAttributeMaxDynamicSharedMemorySize = 8 };
enum { hipDeviceAttributeMultiprocessorCount = 63 };
typedef struct EmuStream_ *hipStream_t;
typedef struct EmuEvent_ { std::chrono::steady_clock::time_point t; } *hipEvent_t;
struct dim3 { unsigned x, y, z; dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {} };

#define __global__
#define __device__
#define __host__
#define __forceinline__ inline
#define __noinline__
#define __launch_bounds__(...)
#define __shared__ thread_local
#define HIP_SYMBOL(x) x

/* the launch's dynamic LDS: a window that ENDS at a PROT_NONE page (emu_lds_window), so that a load or a store beyond the launch's allocation faults here -- on the GPU the
 * store is dropped and the load returns zero without any sign.  The kernels' own `extern __shared__ char smem[]` declarations resolve to this pointer-to-array. */
extern thread_local char (*emu_smem_p)[];
#define smem (*emu_smem_p)
char *emu_lds_window(size_t lds, const char *kernel_name);      /* emu_host.cpp */
extern thread_local unsigned emu_block_x, emu_grid_x;
struct EmuIdx { unsigned x, y, z; };
static inline EmuIdx emu_tidx() { EmuIdx i = {(unsigned)(emu_cur->cur ^ emu_flip), 0, 0}; return i; }
static inline EmuIdx emu_bidx() { EmuIdx i = {emu_block_x, 0, 0}; return i; }
static inline void __syncthreads() { emu_rendezvous(); }
#define threadIdx (emu_tidx())
#define blockIdx (emu_bidx())
static inline EmuIdx emu_gdim() { EmuIdx i = {emu_grid_x, 1, 1}; return i; }
#define gridDim (emu_gdim())

static inline const char *hipGetErrorString(hipError_t) { return "emulated HIP error"; }
static inline hipError_t hipGetLastError() { return hipSuccess; }
static inline hipError_t hipGetDeviceCount(int *n) { *n = 1; return hipSuccess; }
static inline hipError_t hipSetDevice(int) { return hipSuccess; }
static inline hipError_t hipDeviceSynchronize() { return hipSuccess; }
static inline hipError_t hipStreamCreate(hipStream_t *s) { *s = (hipStream_t)malloc(8); return hipSuccess; }
static inline hipError_t hipStreamDestroy(hipStream_t s) { free(s); return hipSuccess; }
static inline hipError_t hipStreamSynchronize(hipStream_t) { return hipSuccess; }
/* device memory is uninitialised on the GPU: fill it with a loud pattern (OA_EMU_FILL=<byte>, default 0xA5 = large negative words; 0x5A = large positive words finds the
 * garbage-as-index reads a negative pattern hides) */
static inline int emu_fill_byte() { static const int v = getenv("OA_EMU_FILL") ? (int)strtol(getenv("OA_EMU_FILL"), nullptr, 0) : 0xA5; return v; }
static inline hipError_t hipMalloc(void **p, size_t n) { *p = aligned_alloc(64, (n + 63) & ~(size_t)63); if (*p) memset(*p, emu_fill_byte(), n); return *p ? hipSuccess : hipErrorUnknown; }
static inline hipError_t hipFree(void *p) { free(p); return hipSuccess; }
static inline hipError_t hipHostMalloc(void **p, size_t n, unsigned = 0) { *p = aligned_alloc(64, (n + 63) & ~(size_t)63); return *p ? hipSuccess : hipErrorUnknown; }
static inline hipError_t hipHostFree(void *p) { free(p); return hipSuccess; }
static inline hipError_t hipMemcpy(void *d, const void *s, size_t n, hipMemcpyKind) { memmove(d, s, n); return hipSuccess; }
static inline hipError_t hipMemcpyAsync(void *d, const void *s, size_t n, hipMemcpyKind, hipStream_t) { memmove(d, s, n); return hipSuccess; }
static inline hipError_t hipMemsetAsync(void *d, int v, size_t n, hipStream_t) { memset(d, v, n); return hipSuccess; }
static inline hipError_t hipMemset(void *d, int v, size_t n) { memset(d, v, n); return hipSuccess; }
static inline hipError_t hipMemcpy2D(void *d, size_t dp, const void *s, size_t sp, size_t w, size_t h, hipMemcpyKind)
{ for (size_t i = 0; i < h; i++) memmove((char *)d + i * dp, (const char *)s + i * sp, w); return hipSuccess; }
static inline hipError_t hipMemcpy2DAsync(void *d, size_t dp, const void *s, size_t sp, size_t w, size_t h, hipMemcpyKind k, hipStream_t) { return hipMemcpy2D(d, dp, s, sp, w, h, k); }
template <class F> static inline hipError_t hipFuncSetAttribute(F, int, int) { return hipSuccess; }
static inline hipError_t hipDeviceGetAttribute(int *v, int, int) { *v = 2; return hipSuccess; }                                 /* a 2-CU, 2-waves-per-CU "chip": persistent launches get a grid of 4 */
static inline hipError_t hipOccupancyMaxActiveBlocksPerMultiprocessor(int *n, const void *, int, size_t) { *n = 2; return hipSuccess; }
static inline unsigned atomicAdd(unsigned *p, unsigned v) { const unsigned o = *p; *p = o + v; return o; }                     /* fibers are cooperative: no race */
static inline hipError_t hipEventCreate(hipEvent_t *e) { *e = new EmuEvent_; return hipSuccess; }
static inline hipError_t hipEventDestroy(hipEvent_t e) { delete e; return hipSuccess; }
static inline hipError_t hipEventRecord(hipEvent_t e, hipStream_t) { e->t = std::chrono::steady_clock::now(); return hipSuccess; }
static inline hipError_t hipEventSynchronize(hipEvent_t) { return hipSuccess; }
static inline hipError_t hipEventElapsedTime(float *ms, hipEvent_t a, hipEvent_t b) { *ms = std::chrono::duration<float, std::milli>(b->t - a->t).count(); return hipSuccess; }

static void emu_launch_tramp(void *p) { (*(std::function<void()> *)p)(); }
/* lds = the launch's dynamic LDS size: on the GPU an access beyond it is dropped / reads zero without any fault.  Here the window ends (to 16 bytes) at an inaccessible page:
 * a kernel that reads or writes there (a struct member outside the part of the layout the launch allocated, a packet window too small for what is coded into it) dies with
 * the kernel's name instead of passing on the emulator and failing -- or silently differing -- on the device */
static inline void emu_launch(dim3 grid, size_t lds, const char *name, std::function<void()> body)
{
   emu_grid_x = grid.x;
   for (unsigned b = 0; b < grid.x; b++) {
      emu_block_x = b;
      char *w = emu_lds_window(lds, name);
      memset(w, emu_fill_byte(), (lds + 15) & ~(size_t)15);         /* LDS is uninitialised on the GPU: make stale reads loud */
      emu_run_wave(emu_launch_tramp, &body);
   }
   emu_lds_window(0, nullptr);
}
#define hipLaunchKernelGGL(kern, grid, block, lds, stream, ...) emu_launch((grid), (size_t)(lds), #kern, [&]() { kern(__VA_ARGS__); })
#endif

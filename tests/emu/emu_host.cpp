/* tests/emu/emu_host.cpp — TEST INFRASTRUCTURE: the complete C ABI of the product (opus_amd/csrc/opus_amd.hip: classic opus_* entry points, the
 * opusgpu_* batch ABI, multistream, packet toolkit) compiled for the CPU on top of the wave emulator, so that the reference's own C test programs
 * (tests/test_opus_api.c, test_opus_encode.c, test_opus_decode.c, src/opus_demo.c) can be linked against it and debugged in this GPU-less
 * container.  "Device" memory is host memory and a launch runs every workgroup on 64 fibers (hip_stub.h).  Never part of the product. */
#include "hip_stub.h"
#include <sys/mman.h>
#include <signal.h>
#include <unistd.h>
/* the LDS window of the running launch: [end - lds, end) of a per-thread mapping whose next page is PROT_NONE (hip_stub.h: emu_launch) */
thread_local char (*emu_smem_p)[] = nullptr;
static thread_local char *emu_lds_map = nullptr;
static thread_local const char *emu_lds_kernel = nullptr;
static thread_local size_t emu_lds_bytes = 0;
static const size_t EMU_LDS_MAX = 160 * 1024, EMU_LDS_GUARD = 64 * 1024;
static struct sigaction emu_old_segv;
static void emu_segv(int sig, siginfo_t *si, void *uc)
{
   const char *a = (const char *)si->si_addr;
   if (emu_lds_map && emu_lds_kernel && a >= emu_lds_map + EMU_LDS_MAX && a < emu_lds_map + EMU_LDS_MAX + EMU_LDS_GUARD) {
      char msg[256];
      const int n = snprintf(msg, sizeof msg, "wave_emu: kernel %s touched LDS byte %zu, beyond its dynamic allocation of %zu bytes\n", emu_lds_kernel,
                             (size_t)(a - (emu_lds_map + EMU_LDS_MAX)) + ((emu_lds_bytes + 15) & ~(size_t)15), emu_lds_bytes);
      if (n > 0) { ssize_t r = write(2, msg, (size_t)n); (void)r; }
      abort();
   }
   if (emu_old_segv.sa_flags & SA_SIGINFO) { if (emu_old_segv.sa_sigaction) { emu_old_segv.sa_sigaction(sig, si, uc); return; } }
   else if (emu_old_segv.sa_handler != SIG_DFL && emu_old_segv.sa_handler != SIG_IGN) { emu_old_segv.sa_handler(sig); return; }
   signal(SIGSEGV, SIG_DFL); raise(SIGSEGV);
}
char *emu_lds_window(size_t lds, const char *kernel_name)
{
   static std::once_flag once;
   std::call_once(once, [] { struct sigaction sa; memset(&sa, 0, sizeof sa); sa.sa_sigaction = emu_segv; sa.sa_flags = SA_SIGINFO | SA_ONSTACK | SA_NODEFER; sigemptyset(&sa.sa_mask); sigaction(SIGSEGV, &sa, &emu_old_segv); });
   if (!emu_lds_map) {
      char *m = (char *)mmap(nullptr, EMU_LDS_MAX + EMU_LDS_GUARD, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
      if (m == (char *)MAP_FAILED || mprotect(m + EMU_LDS_MAX, EMU_LDS_GUARD, PROT_NONE)) { fprintf(stderr, "wave_emu: cannot map the LDS window\n"); abort(); }
      emu_lds_map = m;
   }
   if (lds > EMU_LDS_MAX) { fprintf(stderr, "wave_emu: kernel %s asks for %zu bytes of LDS (a CU has %zu)\n", kernel_name ? kernel_name : "?", lds, EMU_LDS_MAX); abort(); }
   emu_lds_kernel = kernel_name; emu_lds_bytes = lds;
   char *w = emu_lds_map + EMU_LDS_MAX - ((lds + 15) & ~(size_t)15);
   emu_smem_p = (char (*)[])w;
   return w;
}
thread_local unsigned emu_block_x = 0, emu_grid_x = 1;
#define OPUS_AMD_WAVE_H            /* wave_emu.h is the wave vocabulary here */
#define OPUS_AMD_EMU_HOST 1
#include "../../opus_amd/csrc/opus_amd.hip"

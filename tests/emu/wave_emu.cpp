/* tests/emu/wave_emu.cpp — fiber scheduler of the CPU wave emulator (TEST INFRASTRUCTURE). */
#include "wave_emu.h"
thread_local EmuWave *emu_cur = nullptr;
int emu_check_uni = getenv("OA_EMU_CHECK_UNI") && atoi(getenv("OA_EMU_CHECK_UNI"));
int emu_flip = getenv("OA_EMU_REVERSE") && atoi(getenv("OA_EMU_REVERSE")) ? 63 : 0;

__asm__(
   ".text\n.globl emu_switch\n.type emu_switch,@function\nemu_switch:\n"
   "  pushq %rbp\n  pushq %rbx\n  pushq %r12\n  pushq %r13\n  pushq %r14\n  pushq %r15\n"
   "  movq %rsp, (%rdi)\n  movq %rsi, %rsp\n"
   "  popq %r15\n  popq %r14\n  popq %r13\n  popq %r12\n  popq %rbx\n  popq %rbp\n  ret\n"
   ".size emu_switch,.-emu_switch\n");

static void emu_trampoline()
{
   EmuWave *w = emu_cur;
   int me = w->cur;
   w->entry(w->arg);
   w = emu_cur;
   w->f[me].done = 1;
   void *dummy;
   if (me == 63) emu_switch(&dummy, w->main_sp);
   else {
      if (w->f[me + 1].nsync != w->f[me].nsync) {
         fprintf(stderr, "wave_emu: lane %d exited after %ld collectives but lane %d is at %ld\n", me, w->f[me].nsync, me + 1, w->f[me + 1].nsync); abort();
      }
      w->cur = me + 1; emu_switch(&dummy, w->f[me + 1].sp);
   }
   abort();
}

void emu_run_wave(void (*entry)(void *), void *arg)
{
   static const size_t STACK = 1 << 20;
   EmuWave *w = (EmuWave *)calloc(1, sizeof(EmuWave));
   w->entry = entry; w->arg = arg;
   for (int i = 0; i < 64; i++) {
      w->f[i].stack = (char *)aligned_alloc(64, STACK);
      uintptr_t top = ((uintptr_t)w->f[i].stack + STACK) & ~(uintptr_t)15;
      void **sp = (void **)(top - 16);          /* ret slot at a 16-byte boundary */
      sp[0] = (void *)emu_trampoline;
      sp -= 6;                                   /* rbp rbx r12 r13 r14 r15 */
      memset(sp, 0, 6 * sizeof(void *));
      w->f[i].sp = sp;
   }
   EmuWave *saved = emu_cur;
   emu_cur = w;
   w->cur = 0;
   emu_switch(&w->main_sp, w->f[0].sp);
   emu_cur = saved;
   for (int i = 0; i < 64; i++) free(w->f[i].stack);
   free(w);
}

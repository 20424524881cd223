/* tests/emu/demo_rand.c — TEST INFRASTRUCTURE.  The reference's opus_demo draws its random frame sizes, FEC switches and simulated losses from libc's rand(), whose
 * state is shared with everything else in the process; linked against the product library, the ROCm runtime's own threads draw from it too, and the schedule of the
 * run on the GPU is no longer the schedule of the run linked against the reference (seen on the MI355X: a packet lost in one run and not in the other, a different
 * one on every run).  tests/hostemu.py therefore compiles opus_demo.c -- unmodified -- with -Drand=oa_demo_rand -Dsrand=oa_demo_srand and links this file: a private
 * 31-bit sequence (RAND_MAX is 2^31 - 1 here) that only opus_demo advances, the same in all three builds. */
static unsigned long long oa_demo_state = 1;
void oa_demo_srand(unsigned seed) { oa_demo_state = seed; }
int oa_demo_rand(void)
{
   oa_demo_state = oa_demo_state * 6364136223846793005ULL + 1442695040888963407ULL;
   return (int)(oa_demo_state >> 33);
}

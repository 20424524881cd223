/* tests/emu/emu_host.cpp — TEST INFRASTRUCTURE: the complete C ABI of the product (opus_amd/csrc/opus_amd.hip: classic opus_* entry points, the
 * opusgpu_* batch ABI, multistream, packet toolkit) compiled for the CPU on top of the wave emulator, so that the reference's own C test programs
 * (tests/test_opus_api.c, test_opus_encode.c, test_opus_decode.c, src/opus_demo.c) can be linked against it and debugged in this GPU-less
 * container.  "Device" memory is host memory and a launch runs every workgroup on 64 fibers (hip_stub.h).  Never part of the product. */
#include "hip_stub.h"
thread_local __attribute__((aligned(64))) char smem[160 * 1024];
thread_local unsigned emu_block_x = 0;
#define OPUS_AMD_WAVE_H            /* wave_emu.h is the wave vocabulary here */
#define OPUS_AMD_EMU_HOST 1
#include "../../opus_amd/csrc/opus_amd.hip"

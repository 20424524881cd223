#!/bin/bash
# tools/build_variant.sh <name> <extra hipcc flags ...> — an experiment build of the product library as opus_amd/libopus_amd_<name>.so (A/B legs of tools/gpu_r6.sh: exp_lib).
cd "$(dirname "$0")/.."
N=$1; shift
H=$(python -c "import opus_amd; print(opus_amd.source_hash())")
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -Wl,-Bsymbolic -fvisibility=hidden -DOA_SOURCE_HASH=\"$H\" -Iopus_amd/csrc -Iinclude "$@" opus_amd/csrc/opus_amd.hip -o opus_amd/libopus_amd_$N.so

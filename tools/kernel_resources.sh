#!/bin/bash
# tools/kernel_resources.sh — registers / scratch / LDS of every kernel in the product library's gfx950 code object (the numbers occupancy follows from).
# usage: tools/kernel_resources.sh [path/to/libopus_amd.so]
set -e
LIB=${1:-$(dirname "$0")/../opus_amd/libopus_amd.so}
T=$(mktemp -d); trap 'rm -rf $T' EXIT
/opt/rocm/lib/llvm/bin/clang-offload-bundler --type=o --unbundle --targets=hipv4-amdgcn-amd-amdhsa--gfx950 --input="$LIB" --output=$T/co.o 2>/dev/null \
  || { objcopy -O binary --only-section=.hip_fatbin "$LIB" $T/fat.bin; /opt/rocm/lib/llvm/bin/clang-offload-bundler --type=o --unbundle --targets=hipv4-amdgcn-amd-amdhsa--gfx950 --input=$T/fat.bin --output=$T/co.o; }
/opt/rocm/lib/llvm/bin/llvm-readelf --notes $T/co.o | awk '
  /\.name:/ {name=$2} /\.vgpr_count:/ {v=$2} /\.agpr_count:/ {a=$2} /\.sgpr_count:/ {s=$2} /\.private_segment_fixed_size:/ {p=$2} /\.group_segment_fixed_size:/ {g=$2}
  /\.wavefront_size:/ {printf "%-40s vgpr %3d agpr %3d sgpr %3d scratch %5d B/lane lds(static) %6d\n", name, v, a, s, p, g}'

#!/bin/bash
# tools/gpu_profile_sh.sh <tag> — on the GPU box: bench lines + rocprofv3 kernel-trace stats + PMC passes (instruction mix, FETCH_SIZE, WRITE_SIZE in
# separate runs) for the SILK-capable encoder kernel oa_sh_encode_kernel on BASELINE config 3 (SILK-only) and config 4 (hybrid).
set -u
TAG=${1:-cur}
REPO=$(cd "$(dirname "$0")/.." && pwd)
OUT=$REPO/gpurun_out/prof_sh_$TAG
mkdir -p "$OUT"
cd "$REPO"
timeout 300 python tools/silk_enc_bench.py --streams 65536 --steps 10 --warmup 3 > "$OUT/bench_config3_silk.json.log" 2>&1
timeout 300 python tools/silk_enc_bench.py --mode hybrid --streams 65536 --steps 8 --warmup 3 > "$OUT/bench_config4_hybrid.json.log" 2>&1
cd /tmp && export TMPDIR=/tmp
for m in silk hybrid; do
  timeout 300 rocprofv3 --kernel-trace --stats -f csv -d /tmp/kts_${TAG}_$m -- python $REPO/tools/silk_enc_bench.py --mode $m --streams 65536 --steps 5 --warmup 2 --cpu-seconds 0 > "$OUT/under_rocprof_$m.log" 2>&1
  find /tmp/kts_${TAG}_$m -name '*kernel_stats.csv' -exec cp {} "$OUT/rocprofv3_kernel_stats_$m.csv" \;
  timeout 300 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR --kernel-include-regex oa_sh_encode -f csv -d /tmp/pmca_${TAG}_$m -- python $REPO/tools/silk_enc_bench.py --mode $m --streams 16384 --steps 2 --warmup 1 --cpu-seconds 0 > /dev/null 2>&1
  find /tmp/pmca_${TAG}_$m -name '*counter_collection.csv' -exec cp {} "$OUT/pmc_sq_insts_$m.csv" \;
  timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-include-regex oa_sh_encode -f csv -d /tmp/pmcf_${TAG}_$m -- python $REPO/tools/silk_enc_bench.py --mode $m --streams 16384 --steps 2 --warmup 1 --cpu-seconds 0 > /dev/null 2>&1
  find /tmp/pmcf_${TAG}_$m -name '*counter_collection.csv' -exec cp {} "$OUT/pmc_fetch_$m.csv" \;
  timeout 300 rocprofv3 --pmc WRITE_SIZE --kernel-include-regex oa_sh_encode -f csv -d /tmp/pmcw_${TAG}_$m -- python $REPO/tools/silk_enc_bench.py --mode $m --streams 16384 --steps 2 --warmup 1 --cpu-seconds 0 > /dev/null 2>&1
  find /tmp/pmcw_${TAG}_$m -name '*counter_collection.csv' -exec cp {} "$OUT/pmc_write_$m.csv" \;
done
ls -la "$OUT"

#!/bin/bash
# tools/gpu_profile_silk.sh <tag> — on the GPU box: bench lines + rocprofv3 kernel-trace stats + two PMC passes for the SILK kernels.
set -u
TAG=${1:-cur}
REPO=$(cd "$(dirname "$0")/.." && pwd)
OUT=$REPO/gpurun_out/prof_silk_$TAG
mkdir -p "$OUT"
cd "$REPO"
timeout 200 python tools/silk_bench.py --streams 65536 --steps 5 > "$OUT/bench_nsq_del_dec.json.log" 2>&1
timeout 200 python tools/silk_bench.py --plain --streams 65536 --steps 5 > "$OUT/bench_nsq_plain.json.log" 2>&1
timeout 200 python tools/silk_bench.py --kernel resampler --streams 65536 --steps 5 > "$OUT/bench_resampler.json.log" 2>&1
timeout 200 python tools/silk_bench.py --kernel lpc --streams 65536 --steps 5 > "$OUT/bench_lpc.json.log" 2>&1
cd /tmp && export TMPDIR=/tmp
for k in "" "--kernel resampler" "--kernel lpc"; do
  n=$(echo "$k" | tr -d ' -'); n=${n:-nsq}
  timeout 300 rocprofv3 --kernel-trace --stats -f csv -d /tmp/kts_${TAG}_$n -- python $REPO/tools/silk_bench.py $k --streams 65536 --steps 5 --cpu-frames 0 > "$OUT/under_rocprof_$n.log" 2>&1
  find /tmp/kts_${TAG}_$n -name '*kernel_stats.csv' -exec cp {} "$OUT/rocprofv3_kernel_stats_$n.csv" \;
done
timeout 300 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY --kernel-include-regex oa_silk -f csv -d /tmp/pmcs_${TAG}_a -- python $REPO/tools/silk_bench.py --streams 16384 --steps 2 --cpu-frames 0 > /dev/null 2>&1
find /tmp/pmcs_${TAG}_a -name '*counter_collection.csv' -exec cp {} "$OUT/pmc_sq_insts_nsq.csv" \;
timeout 300 rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_INSTS_VALU SQ_BUSY_CYCLES --kernel-include-regex oa_silk -f csv -d /tmp/pmcs_${TAG}_b -- python $REPO/tools/silk_bench.py --streams 16384 --steps 2 --cpu-frames 0 > /dev/null 2>&1
find /tmp/pmcs_${TAG}_b -name '*counter_collection.csv' -exec cp {} "$OUT/pmc_valu_lds_nsq.csv" \;
ls -la "$OUT"

#!/bin/bash
# tools/gpu_profile.sh <tag> — on the GPU box: rocprofv3 kernel-trace stats of the bench command, separate PMC passes for the
# encode kernel, and the per-phase shader-clock breakdown.  Writes only small CSV/TXT summaries into gpurun_out/prof_<tag>/.
set -u
TAG=${1:-cur}
REPO=$(cd "$(dirname "$0")/.." && pwd)
OUT=$REPO/gpurun_out/prof_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
BENCH="python $REPO/bench.py --steps 3 --warmup 1 --no-cpu-baseline"
timeout 300 rocprofv3 --kernel-trace --stats -f csv -d /tmp/kt_$TAG -- $BENCH > "$OUT/bench_under_rocprof.log" 2>&1
find /tmp/kt_$TAG -name '*kernel_stats.csv' -exec cp {} "$OUT/rocprofv3_kernel_stats.csv" \;
pmc() {   # name, counters...
  local name=$1; shift
  timeout 300 rocprofv3 --pmc "$@" --kernel-include-regex oa_encode -f csv -d /tmp/pmc_${TAG}_$name -- $BENCH --streams 16384 > /dev/null 2>&1
  find /tmp/pmc_${TAG}_$name -name '*counter_collection.csv' -exec cp {} "$OUT/pmc_$name.csv" \;
}
pmc sq_insts SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY
pmc lds_vmem SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_INSTS_FLAT SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_WAIT_ANY
pmc valu_busy SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_INSTS_VALU SQ_BUSY_CYCLES
pmc lanes SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAVES
pmc fetch FETCH_SIZE
pmc write WRITE_SIZE
cd "$REPO" && timeout 300 python tools/phase_profile.py 8192 > "$OUT/phase_ticks.txt" 2>&1
ls -la "$OUT"
# counter calibration on a known byte count in the encoder's access width (4 B/lane)
cd /tmp
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 120 rocprofv3 --pmc $c -f csv -d /tmp/cal_${TAG}_$c -- "$REPO/tools/pmc_calibrate" > "$OUT/calib_$c.log" 2>&1
  find /tmp/cal_${TAG}_$c -name '*counter_collection.csv' -exec cp {} "$OUT/calib_$c.csv" \;
done
ls -la "$OUT"

#!/bin/bash
# tools/gpu_pmc.sh <config 2|3|4> <out dir> [--decode] — rocprofv3 counter passes of one bench configuration on the GPU box (separate passes, never with a trace:
# /opt/skills/guides/MI355X_MICROARCH.md), 16,384 streams, every oa_* kernel of the call.  Writes the raw counter CSVs + calibration CSVs into <out dir>; tools/pmc_summary4.py
# condenses them.
set -u
C=$1; OUT=$(realpath -m "$2"); shift 2
REPO=$(cd "$(dirname "$0")/.." && pwd)
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
BENCH="python $REPO/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extra-configs --streams 16384 --config $C $*"
pmc() {   # name, counters...
  local name=$1; shift
  rm -rf /tmp/pmc_$$_$name
  timeout 300 rocprofv3 --pmc "$@" --kernel-include-regex "oa_" -f csv -d /tmp/pmc_$$_$name -- $BENCH > /dev/null 2>&1
  find /tmp/pmc_$$_$name -name '*counter_collection.csv' -exec cp {} "$OUT/pmc_$name.csv" \;
}
pmc sq_insts SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR
pmc valu_busy SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_INSTS_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES
pmc lanes SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAVES
pmc fetch FETCH_SIZE
pmc write WRITE_SIZE
if [ ! -f "$OUT/../calib_FETCH_SIZE.csv" ]; then
  [ -x "$REPO/tools/pmc_calibrate" ] || hipcc --offload-arch=gfx950 -O2 "$REPO/tools/pmc_calibrate.hip" -o "$REPO/tools/pmc_calibrate" > /dev/null 2>&1
  for c in FETCH_SIZE WRITE_SIZE; do
    rm -rf /tmp/cal_$$_$c
    timeout 120 rocprofv3 --pmc $c -f csv -d /tmp/cal_$$_$c -- "$REPO/tools/pmc_calibrate" > /dev/null 2>&1
    find /tmp/cal_$$_$c -name '*counter_collection.csv' -exec cp {} "$OUT/../calib_$c.csv" \;
  done
fi
ls "$OUT"

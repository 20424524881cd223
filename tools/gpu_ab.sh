#!/bin/bash
# tools/gpu_ab.sh <tag> [bench args...] — quick A/B of a build on the GPU box: the bench line (no CPU leg) and the two HBM traffic passes of the encode kernel -> gpurun_out/ab_<tag>/
TAG=${1:-cur}; shift
REPO=$(cd "$(dirname "$0")/.." && pwd)
OUT=$REPO/gpurun_out/ab_$TAG; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
timeout 300 python $REPO/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extra-configs "$@" > "$OUT/bench.log" 2>&1
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 300 rocprofv3 --pmc $c --kernel-include-regex "oa_encode|oa_sh_encode" -f csv -d /tmp/ab_${TAG}_$c -- python $REPO/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extra-configs --streams 16384 "$@" > /dev/null 2>&1
  find /tmp/ab_${TAG}_$c -name '*counter_collection.csv' -exec cp {} "$OUT/pmc_$c.csv" \;
done
python3 - "$OUT" <<'PY'
import csv, sys, collections
d = sys.argv[1]
for c, scale in (("FETCH_SIZE", 2048.0), ("WRITE_SIZE", 1024.0)):
    v = [float(r["Counter_Value"]) for r in csv.DictReader(open("%s/pmc_%s.csv" % (d, c))) if "encode" in r["Kernel_Name"]]
    print(c, "bytes/frame", round(sum(v) / len(v) * scale / 16384))
PY
tail -n 1 "$OUT/bench.log" | cut -c1-400

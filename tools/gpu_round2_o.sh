cd $GRAFT_REPO_ROOT && O=gpurun_out/r02o && mkdir -p $O && export TMPDIR=/tmp
B="python bench.py --no-cpu-baseline --no-extra-configs --steps 8"
( $B ) > $O/bench_default.log 2>&1
( OPUS_AMD_LIB=$PWD/build/libopus_amd_qbinl.so $B ) > $O/bench_qbinl.log 2>&1
for f in $O/bench_*.log; do echo $f; grep -o '"value": [0-9.]*' $f | head -1; done
cd /tmp
for c in FETCH_SIZE WRITE_SIZE; do
  OPUS_AMD_LIB=$GRAFT_REPO_ROOT/build/libopus_amd_qbinl.so timeout 300 rocprofv3 --pmc $c --kernel-include-regex oa_encode -f csv -d /tmp/pmc_o_$c -- python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extra-configs --streams 16384 > /dev/null 2>&1
  find /tmp/pmc_o_$c -name '*counter_collection.csv' -exec cp {} $GRAFT_REPO_ROOT/$O/pmc_qbinl_$c.csv \;
done
python3 - <<'PY'
import csv,glob,os
for f in sorted(glob.glob(os.environ["GRAFT_REPO_ROOT"]+"/gpurun_out/r02o/pmc_*.csv")):
    v=[float(r["Counter_Value"]) for r in csv.DictReader(open(f))]
    print(os.path.basename(f), sum(v)/len(v))
PY

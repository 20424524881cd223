cd $GRAFT_REPO_ROOT && O=gpurun_out/r02k && mkdir -p $O && export TMPDIR=/tmp
B="python bench.py --no-cpu-baseline --no-extra-configs --steps 8"
( $B ) > $O/bench_c2.log 2>&1
( $B --frames-per-launch 8 ) > $O/bench_c2_fpl8.log 2>&1
for f in $O/bench_*.log; do echo $f; grep -o '"value": [0-9.]*\|"frames_per_s": [0-9.]*' $f | head -2; done
( time timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -k "not soak" ) > $O/pytest_parity.log 2>&1; tail -3 $O/pytest_parity.log
( timeout 300 python tools/classic_latency.py 200 ) > $O/classic_latency.log 2>&1; cat $O/classic_latency.log

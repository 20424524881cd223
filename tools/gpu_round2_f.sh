# tools/gpu_round2_f.sh — the driver's view of the tree: whole -m gpu suite (incl. the parity soak and the reference's test programs), smoke, default bench
cd $GRAFT_REPO_ROOT && O=gpurun_out/r02f && mkdir -p $O && export TMPDIR=/tmp
( time python -c "import __graft_entry__ as g; g.smoke()" ) > $O/smoke.log 2>&1; echo "smoke rc=$?" >> $O/summary.txt
( time python bench.py ) > $O/bench_default.log 2>&1; echo "bench rc=$?" >> $O/summary.txt
( time timeout 2400 python -m pytest tests -x -q -m gpu ) > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $O/summary.txt
cat $O/summary.txt; tail -3 $O/smoke.log; tail -5 $O/pytest_gpu.log; grep -o '"value": [0-9.]*' $O/bench_default.log | head -3

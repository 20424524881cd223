# tools/gpu_round2_final.sh — the driver's view of the final tree (smoke, default bench, whole -m gpu suite) + the profile set of the final build
cd $GRAFT_REPO_ROOT && O=gpurun_out/r02final3 && mkdir -p $O && export TMPDIR=/tmp
( time python -c "import __graft_entry__ as g; g.smoke()" ) > $O/smoke.log 2>&1; echo "smoke rc=$?" >> $O/summary.txt
( time python bench.py ) > $O/bench_default.log 2>&1; echo "bench rc=$?" >> $O/summary.txt
OPUS_AMD_PROF_PREBUILT=1 bash tools/gpu_profile.sh r02final3 > $O/profile.log 2>&1; cp -r gpurun_out/prof_r02final3 $O/
( time timeout 2400 python -m pytest tests -x -q -m gpu ) > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $O/summary.txt
cat $O/summary.txt; tail -3 $O/smoke.log; tail -5 $O/pytest_gpu.log; grep -o '"value": [0-9.]*' $O/bench_default.log | head -4; tail -34 gpurun_out/prof_r02final3/phase_ticks.txt

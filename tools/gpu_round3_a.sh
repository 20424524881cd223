#!/bin/bash
# round 3, first GPU pass: the analysis on the device (IEEE behaviour of the real hardware), the ADVICE regression tests, smoke, config 2 with and without the analysis
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out/r03_a
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r03_a/smoke.log 2>&1
timeout 900 python -m pytest tests/test_gpu_analysis.py -x -q > gpurun_out/r03_a/pytest_analysis.log 2>&1
timeout 600 python -m pytest tests/test_gpu_classic_api.py -x -q -k "hard_cbr or bitrate_max or packet_pad or ms_" > gpurun_out/r03_a/pytest_api_limits.log 2>&1
timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extra-configs > gpurun_out/r03_a/bench_analysis_on.log 2>&1
timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extra-configs --no-analysis > gpurun_out/r03_a/bench_analysis_off.log 2>&1
for f in gpurun_out/r03_a/*.log; do tail -n 3 "$f"; done

#!/bin/bash
# round 2, validation r: dynalloc / tf_analysis / temporal VBR / patch_transient on the wave -- parity subset, float gate, config 2 bench
cd $GRAFT_REPO_ROOT; O=gpurun_out/r02r; mkdir -p $O; export TMPDIR=/tmp
( time timeout 120 python -m pytest tests/test_gpu_parity.py tests/test_gpu_classic_api.py tests/test_gpu_float_decoder_gate.py -x -q -k "not soak" ) > $O/pytest_subset.log 2>&1; tail -4 $O/pytest_subset.log
( timeout 100 python bench.py --no-cpu-baseline --steps 5 --config 2 --no-extra-configs ) > $O/bench.log 2>&1; grep -o '"value": [0-9.]*' $O/bench.log | head -3

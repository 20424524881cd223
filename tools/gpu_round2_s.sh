#!/bin/bash
# round 2, validation s: SILK LTP / NLSF interpolation / stereo LR->MS on the wave -- SILK + hybrid parity tests, default bench (configs 2, 3, 4)
cd $GRAFT_REPO_ROOT; O=gpurun_out/r02s; mkdir -p $O; export TMPDIR=/tmp
( time timeout 100 python -m pytest tests/test_gpu_silkenc.py tests/test_gpu_ms_batch.py -x -q ) > $O/pytest_silk.log 2>&1; tail -4 $O/pytest_silk.log
( timeout 150 python bench.py --no-cpu-baseline --steps 4 ) > $O/bench.log 2>&1; grep -o '"value": [0-9.]*' $O/bench.log | head -3

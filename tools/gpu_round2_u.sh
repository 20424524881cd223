#!/bin/bash
# round 2, validation u (last GPU seconds of the round): lane-per-chain exp_rotation -- config 2 encode + decode rates, then the CELT parity tests
cd $GRAFT_REPO_ROOT; O=gpurun_out/r02u; mkdir -p $O; export TMPDIR=/tmp
( timeout 40 python bench.py --no-cpu-baseline --steps 4 --config 2 --no-extra-configs ) > $O/bench.log 2>&1; grep -o '"value": [0-9.]*' $O/bench.log | head -1
( timeout 60 python -m pytest tests/test_gpu_parity.py tests/test_gpu_decoder.py -x -q -k "not soak and not full_size" ) > $O/pytest.log 2>&1; tail -2 $O/pytest.log

cd $GRAFT_REPO_ROOT && O=gpurun_out/r02p && mkdir -p $O && export TMPDIR=/tmp
( time timeout 200 python -m pytest tests/test_zz_reference_programs.py -x -q -k "gpu_opus_demo" ) > $O/pytest_demo.log 2>&1; tail -3 $O/pytest_demo.log
( time timeout 200 python -m pytest tests/test_gpu_silkenc.py tests/test_gpu_classic_api.py tests/test_gpu_parity.py -x -q -k "not soak" ) > $O/pytest_subset.log 2>&1; tail -3 $O/pytest_subset.log
( python bench.py --no-cpu-baseline --steps 5 ) > $O/bench.log 2>&1; grep -o '"value": [0-9.]*' $O/bench.log | head -3

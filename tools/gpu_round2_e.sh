# tools/gpu_round2_e.sh — wavefront-scope ordering (no drains at lane-0 section boundaries): bench A/B + the whole GPU parity suite on the light-sync build
cd $GRAFT_REPO_ROOT && O=gpurun_out/r02e && mkdir -p $O && export TMPDIR=/tmp
B="python bench.py --no-cpu-baseline --no-extra-configs --steps 5"
( $B ) > $O/bench_heavy.log 2>&1
export OPUS_AMD_LIB=$PWD/build/libopus_amd_light.so
( $B ) > $O/bench_light.log 2>&1
( $B --config 3 ) > $O/bench_light_config3.log 2>&1
( $B --config 4 ) > $O/bench_light_config4.log 2>&1
for f in $O/bench_*.log; do echo $f; grep -o '"value": [0-9.]*' $f | head -1; done
( time timeout 1200 python -m pytest tests -m gpu -q -x --deselect tests/test_zz_reference_programs.py ) > $O/pytest_gpu_light.log 2>&1
tail -5 $O/pytest_gpu_light.log

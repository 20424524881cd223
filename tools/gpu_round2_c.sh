# tools/gpu_round2_c.sh — A/B of the LDS-diet / persistent-wave encoder: waves-per-SIMD variants, then parity on the new build
cd $GRAFT_REPO_ROOT && O=gpurun_out/r02c && mkdir -p $O && export TMPDIR=/tmp
B="python bench.py --no-cpu-baseline --no-extra-configs --steps 5"
( $B ) > $O/bench_w3.log 2>&1
( OPUS_AMD_LIB=$PWD/build/libopus_amd_w2.so $B ) > $O/bench_w2.log 2>&1
( OPUS_AMD_LIB=$PWD/build/libopus_amd_w4.so $B ) > $O/bench_w4.log 2>&1
for g in 2048 2560 3584 4096; do ( OPUS_AMD_GRID=$g $B ) > $O/bench_w3_grid$g.log 2>&1; done
( $B --config 4 ) > $O/bench_config4.log 2>&1
( $B --config 3 ) > $O/bench_config3.log 2>&1
( time timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_silkenc.py tests/test_gpu_classic_api.py tests/test_gpu_multistream.py -x -q ) > $O/pytest_parity.log 2>&1
grep -h -o '"value": [0-9.]*\|"ms_per_step": [0-9.]*' $O/bench_*.log | paste - - ; tail -3 $O/pytest_parity.log
for f in $O/bench_*.log; do echo $f; grep -o '"value": [0-9.]*' $f | head -1; done

# tools/gpu_round2_i.sh — decoder rewrite on the GPU (parity + decode bench), inline-variant A/B of the encoder
cd $GRAFT_REPO_ROOT && O=gpurun_out/r02i && mkdir -p $O && export TMPDIR=/tmp
( time timeout 900 python -m pytest tests/test_gpu_decoder.py tests/test_gpu_silkdec.py tests/test_gpu_parity.py -x -q -k "not soak" ) > $O/pytest_dec.log 2>&1; tail -3 $O/pytest_dec.log
( python bench.py --decode --no-cpu-baseline --no-extra-configs --steps 5 ) > $O/bench_decode.log 2>&1; tail -2 $O/bench_decode.log | cut -c1-600
B="python bench.py --no-cpu-baseline --no-extra-configs --steps 5"
( $B ) > $O/bench_default.log 2>&1
( OPUS_AMD_LIB=$PWD/build/libopus_amd_inl.so $B ) > $O/bench_inl.log 2>&1
( OPUS_AMD_LIB=$PWD/build/libopus_amd_inl.so $B --config 3 ) > $O/bench_inl_c3.log 2>&1
( OPUS_AMD_LIB=$PWD/build/libopus_amd_inl.so $B --config 4 ) > $O/bench_inl_c4.log 2>&1
for f in $O/bench_*.log; do echo $f; grep -o '"value": [0-9.]*' $f | head -1; done

# tools/gpu_round2_d.sh — 16-waves-per-CU build: A/B against the 3-per-SIMD variant, parity, then the profile set (kernel stats, PMC passes, phase ticks)
cd $GRAFT_REPO_ROOT && O=gpurun_out/r02d && mkdir -p $O && export TMPDIR=/tmp
B="python bench.py --no-cpu-baseline --no-extra-configs --steps 5"
( $B ) > $O/bench_w4.log 2>&1
( OPUS_AMD_LIB=$PWD/build/libopus_amd_w3.so $B ) > $O/bench_w3.log 2>&1
( time timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_silkenc.py tests/test_gpu_classic_api.py tests/test_gpu_multistream.py -x -q ) > $O/pytest_parity.log 2>&1
for f in $O/bench_*.log; do echo $f; grep -o '"value": [0-9.]*' $f | head -1; done; tail -3 $O/pytest_parity.log
OPUS_AMD_PROF_PREBUILT=1 bash tools/gpu_profile.sh r02d > $O/profile.log 2>&1
cp -r gpurun_out/prof_r02d $O/
tail -45 gpurun_out/prof_r02d/phase_ticks.txt

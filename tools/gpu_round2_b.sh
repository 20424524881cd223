# tools/gpu_round2_b.sh — second GPU call of round 2: bench (headline + configs 3/4 + frames-per-launch), rocprofv3 kernel stats, whole -m gpu suite,
# committed bitstream vectors on the GPU, classic-API latency, and the reference's long test programs in the background with line-buffered logs.
cd $GRAFT_REPO_ROOT && O=gpurun_out/r02b && mkdir -p $O && export TMPDIR=/tmp
( time python bench.py ) > $O/bench_default.log 2>&1
( time python bench.py --config 3 --no-cpu-baseline --steps 5 ) > $O/bench_config3.log 2>&1
( time python bench.py --config 4 --no-cpu-baseline --steps 5 ) > $O/bench_config4.log 2>&1
( time python bench.py --config 2 --no-cpu-baseline --no-extra-configs --steps 2 --frames-per-launch 50 ) > $O/bench_fpl50.log 2>&1
( cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats -f csv -d /tmp/kt_b -- python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $GRAFT_REPO_ROOT/$O/bench_under_rocprof.log 2>&1; find /tmp/kt_b -name '*kernel_stats.csv' -exec cp {} $GRAFT_REPO_ROOT/$O/rocprofv3_kernel_stats.csv \; )
# the long reference programs: start now, collect at the end
( time TEST_OPUS_NOFUZZ=1 SEED=20260922 timeout 1500 stdbuf -oL oracle/_ref/reftests/gpu/test_opus_encode ) > $O/ref_test_opus_encode_nofuzz.log 2>&1 &
P1=$!
( time SEED=20260922 timeout 1500 stdbuf -oL oracle/_ref/reftests/gpu/test_opus_decode ) > $O/ref_test_opus_decode.log 2>&1 &
P2=$!
( time timeout 300 python tools/classic_latency.py 200 ) > $O/classic_latency.log 2>&1
( time timeout 1500 python -m pytest tests -m gpu -q --deselect tests/test_zz_reference_programs.py::test_gpu_test_opus_decode --deselect tests/test_zz_reference_programs.py::test_gpu_test_opus_encode ) > $O/pytest_gpu.log 2>&1
( time timeout 600 python tools/run_vectors_gpu.py ) > $O/run_vectors.log 2>&1
wait $P1; echo "encode_nofuzz rc=$?" >> $O/summary.txt
wait $P2; echo "decode rc=$?" >> $O/summary.txt
tail -4 $O/*.log; cat $O/summary.txt

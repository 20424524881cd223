cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out/r02a && export TMPDIR=/tmp
( time timeout 900 python -m pytest tests/test_gpu_classic_api.py -x -q ) > gpurun_out/r02a/classic_api.log 2>&1
( time timeout 300 oracle/_ref/reftests/gpu/test_opus_api ) > gpurun_out/r02a/ref_test_opus_api.log 2>&1; echo "api rc=$?" >> gpurun_out/r02a/summary.txt
( time timeout 120 oracle/_ref/reftests/gpu/test_opus_padding ) > gpurun_out/r02a/ref_test_opus_padding.log 2>&1; echo "padding rc=$?" >> gpurun_out/r02a/summary.txt
( time timeout 420 oracle/_ref/reftests/gpu/test_opus_encode ) > gpurun_out/r02a/ref_test_opus_encode.log 2>&1; echo "encode rc=$?" >> gpurun_out/r02a/summary.txt
( time timeout 300 oracle/_ref/reftests/gpu/test_opus_decode ) > gpurun_out/r02a/ref_test_opus_decode.log 2>&1; echo "decode rc=$?" >> gpurun_out/r02a/summary.txt
( time python bench.py ) > gpurun_out/r02a/bench.log 2>&1
( time timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_silkenc.py tests/test_gpu_multistream.py -x -q ) > gpurun_out/r02a/old_gpu_tests.log 2>&1
tail -3 gpurun_out/r02a/*.log; cat gpurun_out/r02a/summary.txt

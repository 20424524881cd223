#!/bin/bash
# tools/gpu_encode_trace.sh — on the MI355X box: the reference's unmodified test_opus_encode (mode matrix, multistream, frame-size switching, settings fuzz, regression
# cases) linked to opus_amd/libopus_amd.so, every opus_encode / opus_multistream_encode call logged by tools/encode_trace_shim.c, compared call by call with the log of
# the same program linked to the compiled reference (float API on), which was recorded in the build container and travels as oracle/_ref/enctrace/fxa_<seed>.log.gz.
#   gpurun --timeout 2700 -- 'bash tools/gpu_encode_trace.sh r03_m 12345 20260922'
out=gpurun_out/$1; shift
mkdir -p $out
gcc -O2 -shared -fPIC tools/encode_trace_shim.c -o gpurun_out/enc_shim.so -ldl || exit 1
for seed in "$@"; do
  ( SEED=$seed OPUS_AMD_FLOAT_ANALYSIS=1 OPUS_TRACE_FILE=gpurun_out/gpu_$seed.log LD_PRELOAD=$PWD/gpurun_out/enc_shim.so timeout 2400 oracle/_ref/reftests/gpu/test_opus_encode > $out/test_opus_encode_$seed.out 2>&1; echo "rc=$?" >> $out/test_opus_encode_$seed.out ) &
done
wait
for seed in "$@"; do
  zcat oracle/_ref/enctrace/fxa_$seed.log.gz > /tmp/fxa_$seed.log
  python tools/encode_trace_compare.py /tmp/fxa_$seed.log gpurun_out/gpu_$seed.log > $out/compare_$seed.txt 2>&1
  echo "seed $seed: $(head -1 $out/compare_$seed.txt); $(tail -2 $out/test_opus_encode_$seed.out | tr '\n' ' ')"
  gzip -f gpurun_out/gpu_$seed.log; mv gpurun_out/gpu_$seed.log.gz $out/
done
rm -f gpurun_out/enc_shim.so

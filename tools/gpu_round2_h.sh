# tools/gpu_round2_h.sh — multistream batch on the GPU: parity tests, config-5 bench, SH-kernel instruction-cache counters
cd $GRAFT_REPO_ROOT && O=gpurun_out/r02h && mkdir -p $O && export TMPDIR=/tmp
( time timeout 900 python -m pytest tests/test_gpu_ms_batch.py tests/test_gpu_multistream.py -x -q ) > $O/pytest_ms.log 2>&1; tail -4 $O/pytest_ms.log
( time python bench.py --config 5 --steps 5 ) > $O/bench_config5.log 2>&1; tail -3 $O/bench_config5.log | cut -c1-1500
cd /tmp
for set in "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE" "SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU"; do
  n=$(echo $set | cut -d' ' -f1)
  for c in 3 4; do
    timeout 300 rocprofv3 --pmc $set --kernel-include-regex oa_sh_encode -f csv -d /tmp/pmc_h_${n}_$c -- python $GRAFT_REPO_ROOT/bench.py --config $c --steps 3 --warmup 1 --no-cpu-baseline --streams 16384 > $GRAFT_REPO_ROOT/$O/pmc_${n}_c$c.log 2>&1
    find /tmp/pmc_h_${n}_$c -name '*counter_collection.csv' -exec cp {} $GRAFT_REPO_ROOT/$O/pmc_${n}_c$c.csv \;
  done
done
ls $GRAFT_REPO_ROOT/$O

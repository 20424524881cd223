# tools/gpu_round2_g.sh — code-size experiments (-Os / -O2 builds) and instruction-cache counters of the encode kernel
cd $GRAFT_REPO_ROOT && O=gpurun_out/r02g && mkdir -p $O && export TMPDIR=/tmp
B="python bench.py --no-cpu-baseline --no-extra-configs --steps 5"
( $B ) > $O/bench_O3.log 2>&1
( OPUS_AMD_LIB=$PWD/build/libopus_amd_Os.so $B ) > $O/bench_Os.log 2>&1
( OPUS_AMD_LIB=$PWD/build/libopus_amd_O2.so $B ) > $O/bench_O2.log 2>&1
( OPUS_AMD_LIB=$PWD/build/libopus_amd_Os.so $B --config 4 ) > $O/bench_Os_config4.log 2>&1
( OPUS_AMD_LIB=$PWD/build/libopus_amd_Os.so $B --config 3 ) > $O/bench_Os_config3.log 2>&1
for f in $O/bench_*.log; do echo $f; grep -o '"value": [0-9.]*' $f | head -1; done
cd /tmp
for set in "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE" "SQ_IFETCH SQ_IFETCH_LEVEL SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQC_DCACHE_REQ SQC_DCACHE_HITS SQC_DCACHE_MISSES SQ_INSTS_SMEM"; do
  n=$(echo $set | cut -d' ' -f1)
  timeout 300 rocprofv3 --pmc $set --kernel-include-regex oa_encode -f csv -d /tmp/pmc_g_$n -- python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extra-configs --streams 16384 > $GRAFT_REPO_ROOT/$O/pmc_$n.log 2>&1
  find /tmp/pmc_g_$n -name '*counter_collection.csv' -exec cp {} $GRAFT_REPO_ROOT/$O/pmc_$n.csv \;
done
ls -la $GRAFT_REPO_ROOT/$O

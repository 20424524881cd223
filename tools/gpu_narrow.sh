cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06_j
for S in 64 256 1024 4096 16384; do for m in 0 1; do
  OPUS_AMD_CELT_PIPE=$m timeout 200 python bench.py --steps 12 --warmup 3 --no-cpu-baseline --no-extra-configs --steady-state 0 --streams $S --config 2 2>/dev/null | tail -1 | python -c "
import sys, json
r = json.loads(sys.stdin.read()); print('config2 S=$S pipe=$m', r['value'], r['ms_per_step'], r['roofline'].get('kernels_ms'))"
done; done > gpurun_out/r06_j/narrow.log 2>&1
for S in 256 1024 4096; do for m in 0 1; do
  OPUS_AMD_SH_PVQ4=$m timeout 200 python bench.py --steps 12 --warmup 3 --no-cpu-baseline --no-extra-configs --steady-state 0 --streams $S --config 4 2>/dev/null | tail -1 | python -c "
import sys, json
r = json.loads(sys.stdin.read()); print('config4 S=$S pvq4=$m', r['value'], r['ms_per_step'], r['roofline'].get('kernels_ms'))"
done; done >> gpurun_out/r06_j/narrow.log 2>&1
cat gpurun_out/r06_j/narrow.log

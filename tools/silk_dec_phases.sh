#!/bin/bash
# tools/silk_dec_phases.sh — on the GPU box: where does the lane-0 SILK layer spend its time?  Rebuilds the library with one stage compiled out
# at a time (results are then wrong on purpose) and times SILK-only decode.  Writes gpurun_out/silk_dec_phases.txt.
cd "$(dirname "$0")/.."
for f in "" "-DSD_PROF_SKIP_RESAMPLE" "-DSD_PROF_SKIP_CORE" "-DSD_PROF_SKIP_PARAMS" "-DSD_PROF_SKIP_RESAMPLE -DSD_PROF_SKIP_CORE -DSD_PROF_SKIP_PARAMS"; do
  OPUS_AMD_EXTRA_CFLAGS="$f" python -c "import opus_amd; opus_amd.build(force=True)" > /dev/null 2>&1
  echo "== flags: [$f]"
  timeout 300 python tools/silk_bench.py --kernel decode --streams 32768 --steps 4 --warmup 1 --cpu-frames 0 2>&1 | tail -1 | python -c "import sys, json; d = json.loads(sys.stdin.read()); print({k: round(v['ms_per_step'], 2) for k, v in d.items() if isinstance(v, dict)})"
done
python -c "import opus_amd; opus_amd.build(force=True)" > /dev/null 2>&1

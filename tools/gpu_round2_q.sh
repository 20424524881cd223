#!/bin/bash
# round 2, diagnostic q: which opus_demo schedule cases differ on the MI355X, and the call-by-call trace of case 7 (tools/demo_trace.py)
cd $GRAFT_REPO_ROOT; O=gpurun_out/r02q; mkdir -p $O
timeout 100 python tools/demo_trace.py 7 gpu $O > $O/trace7.txt 2>&1
rm -f $O/in.pcm
timeout 120 python -m pytest tests/test_zz_reference_programs.py -q -k "gpu_opus_demo_schedules" > $O/schedules.log 2>&1
tail -5 $O/schedules.log; grep -n "^FAILED" $O/schedules.log

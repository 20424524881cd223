#!/bin/bash
# round 2, profile t: stage shares of the SILK-capable encoder kernel (lane-0 shader clocks) on configs 3 and 4 at the end-of-round build
cd $GRAFT_REPO_ROOT; O=gpurun_out/r02t; mkdir -p $O; export TMPDIR=/tmp OPUS_AMD_PROF_PREBUILT=1
timeout 70 python tools/phase_profile_sh.py 8192 10 > $O/silk_phases.txt 2> $O/silk_phases.err
timeout 70 python tools/phase_profile_sh.py 8192 10 hybrid > $O/hybrid_phases.txt 2> $O/hybrid_phases.err
tail -19 $O/silk_phases.txt; tail -19 $O/hybrid_phases.txt

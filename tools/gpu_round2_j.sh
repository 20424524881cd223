# tools/gpu_round2_j.sh — decode bench on the rewritten decoder side information, SILK / hybrid phase shares of the round-2 build
cd $GRAFT_REPO_ROOT && O=gpurun_out/r02j && mkdir -p $O && export TMPDIR=/tmp
( python bench.py --decode --no-cpu-baseline --steps 5 ) > $O/bench_decode_c2.log 2>&1; grep -o '"value": [0-9.]*\|"all_packets_valid": [a-z]*' $O/bench_decode_c2.log | head -3
( python bench.py --decode --config 3 --no-cpu-baseline --steps 5 ) > $O/bench_decode_c3.log 2>&1; grep -o '"value": [0-9.]*\|"all_packets_valid": [a-z]*' $O/bench_decode_c3.log | head -3
( python bench.py --decode --config 4 --no-cpu-baseline --steps 5 ) > $O/bench_decode_c4.log 2>&1; grep -o '"value": [0-9.]*\|"all_packets_valid": [a-z]*' $O/bench_decode_c4.log | head -3
( OPUS_AMD_PROF_PREBUILT=1 python tools/phase_profile_sh.py 8192 10 ) > $O/silk_phases.txt 2>&1; tail -20 $O/silk_phases.txt
( OPUS_AMD_PROF_PREBUILT=1 python tools/phase_profile_sh.py 8192 10 hybrid ) > $O/hybrid_phases.txt 2>&1; tail -20 $O/hybrid_phases.txt

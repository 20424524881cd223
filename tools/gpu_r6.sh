#!/bin/bash
# tools/gpu_r6.sh <tag> <step> [<step> ...] — the round-6 GPU passes, one parameterised script (output under gpurun_out/r06_<tag>/).
# steps: pipeline | icache2..4 | phases2 | split_check | silkenc | bench3 | bench4 | bench2 | bench_default | prof3 | prof4 | prof2 | pmc2 | pmc3 | pmc4 | full | smoke | decode | decfast | phases | bench34p | broad
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
TAG=$1; shift
O=gpurun_out/r06_$TAG; mkdir -p $O
export TMPDIR=/tmp
B="--steps 10 --warmup 3 --no-cpu-baseline --no-extra-configs $BENCH_EXTRA"
for step in "$@"; do
case $step in
pipeline) timeout 1500 python -m pytest tests/test_gpu_pipeline.py -q -n 4 --timeout 900 > $O/pytest_pipeline.log 2>&1 ;;
icache2|icache3|icache4) c=${step#icache}; for v in ${EXP_LIBS:-libopus_amd.so}; do (cd /tmp && rm -rf /tmp/ic_$c && OPUS_AMD_LIB=$OLDPWD/opus_amd/$v timeout 300 rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQ_IFETCH SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY --kernel-include-regex "oa_" -f csv -d /tmp/ic_$c -- python $OLDPWD/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extra-configs --streams 16384 --config $c > /dev/null 2>&1; find /tmp/ic_$c -name '*counter_collection.csv' -exec cp {} $OLDPWD/$O/icache${c}_$v.csv \;); python tools/pmc_rows.py $O/icache${c}_$v.csv > $O/icache${c}_$v.log 2>&1; done ;;
pcs2|pcs3|pcs4) c=${step#pcs}; M=${PCS_METHOD:-stochastic}; if [ $M = stochastic ]; then PU="--pc-sampling-unit cycles --pc-sampling-interval ${PCS_INTERVAL:-1048576}"; else PU="--pc-sampling-unit time --pc-sampling-interval ${PCS_INTERVAL:-100}"; fi
  (cd /tmp && rm -rf /tmp/pcs_$c && ROCPROFILER_PC_SAMPLING_BETA_ENABLED=1 timeout 200 rocprofv3 --pc-sampling-beta-enabled --pc-sampling-method $M $PU -f csv -d /tmp/pcs_$c -- python $OLDPWD/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extra-configs --streams ${PCS_STREAMS:-16384} --config $c > $OLDPWD/$O/pcs${c}_$M.log 2>&1; ls -laR /tmp/pcs_$c >> $OLDPWD/$O/pcs${c}_$M.log; for f in $(find /tmp/pcs_$c -name '*.csv'); do python $OLDPWD/tools/pcs_hist.py $f > $OLDPWD/$O/pcs${c}_${M}_$(basename $f).hist 2>&1; done) ;;
phases2) OPUS_AMD_PROF_PREBUILT=1 timeout 200 python tools/phase_profile.py 16384 > $O/phases_config2.txt 2>&1 ;;
celtpipe) timeout 600 python tests/celt_pipe_check.py gpu > $O/celt_pipe_check.log 2>&1 ;;
bench2ab) for m in 1 0; do OPUS_AMD_CELT_PIPE=$m timeout 300 python bench.py $B > $O/bench2_pipe$m.log 2>&1; done ;;
p4phases) OPUS_AMD_PROF_PREBUILT=1 timeout 200 python tools/phase_profile_p4.py 16384 > $O/p4_phases_mixed.txt 2>&1; OPUS_AMD_PROF_PREBUILT=1 timeout 200 python tools/phase_profile_p4.py 16384 same > $O/p4_phases_same.txt 2>&1 ;;
bench45ab) for c in 4 5; do for m in 1 0; do OPUS_AMD_SH_PVQ4=$m timeout 300 python bench.py $B --config $c > $O/bench${c}_pvq4_$m.log 2>&1; done; done ;;
smoke) python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1 ;;
split_check) timeout 900 python tools/split_check.py gpu > $O/split_check.log 2>&1 ;;
silkenc) timeout 1200 python -m pytest tests/test_gpu_silkenc.py -x -q > $O/pytest_silkenc.log 2>&1 ;;
classic) timeout 1200 python -m pytest tests/test_gpu_classic_api.py -x -q > $O/pytest_classic.log 2>&1 ;;
bench2) timeout 300 python bench.py $B > $O/bench2.log 2>&1 ;;
pool) for c in 2 3 4; do timeout 300 python bench.py $B --config $c --corpus pool > $O/bench${c}_pool.log 2>&1; done ;;
bench3) for m in ${SPLIT_MODES:-0 1}; do OPUS_AMD_SH_SPLIT=$m timeout 300 python bench.py $B --config 3 > $O/bench3_split$m.log 2>&1; done ;;
bench4) for m in ${SPLIT_MODES:-0 1}; do OPUS_AMD_SH_SPLIT=$m timeout 300 python bench.py $B --config 4 > $O/bench4_split$m.log 2>&1; done ;;
bench5) timeout 300 python bench.py $B --config 5 > $O/bench5.log 2>&1 ;;
decode) for f in ${DEC_FAST_MODES:-0 1}; do for c in 2 3 4; do OPUS_AMD_DEC_FAST=$f timeout 120 python bench.py $B --config $c --decode > $O/decode${c}_fast$f.log 2>&1; done; done ;;
dectests) timeout 400 python -m pytest tests/test_gpu_decoder.py tests/test_gpu_silkdec.py tests/test_gpu_float_decoder_gate.py tests/test_gpu_ms_batch.py -x -q --timeout 90 > $O/pytest_decoder.log 2>&1 ;;
bench_default) timeout 900 python bench.py > $O/bench_default.log 2>&1 ;;
final) timeout 1500 python bench.py > $O/bench_default.log 2>&1   # the driver's line first, in the SAME call as the profiles below (profiles/INDEX.md states the delta between the two)
  # the round's closing measurement, on the build that is in the tree: rocprofv3 kernel stats of every bench leg (configs 2-5, the three decoder legs), the five counter passes
  # of each, condensed into profiles-ready files under $O (copy to profiles/r06_final + profiles/pmc_traffic_r06.json)
  for c in 2 3 4 5; do (cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -f csv -d $OLDPWD/$O/prof$c -o p -- python $OLDPWD/bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-extra-configs --steady-state 0 --config $c > $OLDPWD/$O/prof$c.log 2>&1); find $O/prof$c -name '*kernel_trace*' -delete; find $O/prof$c -name '*agent_info*' -delete; done
  for c in 2 3 4; do (cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -f csv -d $OLDPWD/$O/profd$c -o p -- python $OLDPWD/bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-extra-configs --steady-state 0 --config $c --decode > $OLDPWD/$O/profd$c.log 2>&1); find $O/profd$c -name '*kernel_trace*' -delete; find $O/profd$c -name '*agent_info*' -delete; done
  for c in 2 3 4 5; do timeout 900 bash tools/gpu_pmc.sh $c $O/pmc/pmc$c --steady-state 0 > $O/pmc$c.log 2>&1; done
  for c in 2 3 4; do timeout 900 bash tools/gpu_pmc.sh $c $O/pmc/pmcd$c --steady-state 0 --decode > $O/pmcd$c.log 2>&1; done
  python tools/pmc_summary4.py $O/pmc $O/pmc_traffic_r06.json > $O/pmc_summary.log 2>&1
  bash tools/kernel_resources.sh > $O/kernel_resources.txt 2>&1 ;;
prof2|prof3|prof4) c=${step#prof}; (cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -f csv -d $OLDPWD/$O/prof$c -o p -- python $OLDPWD/bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-extra-configs --config $c > $OLDPWD/$O/prof$c.log 2>&1); find $O/prof$c -name '*kernel_trace*' -delete; find $O/prof$c -name '*agent_info*' -delete ;;
pmc2|pmc3|pmc4) c=${step#pmc}; timeout 900 bash tools/gpu_pmc.sh $c $O/pmc/pmc$c > $O/pmc$c.log 2>&1 ;;
pmcd2|pmcd3|pmcd4) c=${step#pmcd}; timeout 900 bash tools/gpu_pmc.sh $c $O/pmc/pmcd$c --decode > $O/pmcd$c.log 2>&1 ;;
exp_occ) for pad in 0 20000 60000; do OPUS_AMD_SH_LDS_PAD=$pad timeout 300 python bench.py $B --config 3 > $O/bench3_pad$pad.log 2>&1; done ;;
exp_lib) for f in $EXP_LIBS; do for c in $EXP_CONFIGS; do OPUS_AMD_LIB=$PWD/opus_amd/$f timeout 300 python bench.py $B --config $c > $O/bench${c}_$f.log 2>&1; done; done ;;
msbatch) timeout 1500 python -m pytest tests/test_gpu_ms_batch.py tests/test_gpu_float_decoder_gate.py -x -q -s > $O/pytest_ms_batch.log 2>&1 ;;
latency) for m in 0 1; do OPUS_AMD_SH_SPLIT=$m timeout 600 python tools/classic_latency.py 200 > $O/classic_latency_split$m.log 2>&1; done; for c in 2 3 4; do timeout 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-extra-configs --streams 1 --config $c > $O/bench${c}_one_stream.log 2>&1; done ;;
soak) timeout 1500 python tools/parity_soak.py --float-analysis --streams ${SOAK_STREAMS:-512} --frames ${SOAK_FRAMES:-600} --configs 2,3,4 > $O/parity_soak_analysis.log 2>&1 ;;
ranks) timeout 1800 python -m pytest tests/test_gpu_bench_ranks.py -x -q -s > $O/pytest_bench_ranks.log 2>&1 ;;
declane) timeout 300 python tools/dec_fast_check.py gpu > $O/dec_fast_check.log 2>&1; for c in 3 4 2; do timeout 220 python bench.py --steps 8 --warmup 2 --no-extra-configs --config $c --decode > $O/decode$c.log 2>&1; done
  for m in 0 1; do OPUS_AMD_DEC_LANE=$m timeout 200 python bench.py --steps 8 --warmup 2 --no-extra-configs --no-cpu-baseline --config 3 --decode > $O/decode3_lane$m.log 2>&1; done
  (cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -f csv -d $OLDPWD/$O/profd3 -o p -- python $OLDPWD/bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-extra-configs --steady-state 0 --config 3 --decode > $OLDPWD/$O/profd3.log 2>&1); find $O/profd3 -name '*kernel_trace*' -delete; find $O/profd3 -name '*agent_info*' -delete
  (cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -f csv -d $OLDPWD/$O/profd4 -o p -- python $OLDPWD/bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-extra-configs --steady-state 0 --config 4 --decode > $OLDPWD/$O/profd4.log 2>&1); find $O/profd4 -name '*kernel_trace*' -delete; find $O/profd4 -name '*agent_info*' -delete
  timeout 600 python -m pytest tests/test_gpu_decoder.py tests/test_gpu_dec_fast.py -x -q --timeout 300 > $O/pytest_decoder.log 2>&1 ;;
gpufuzz) OPUS_AMD_DEC_PVQ4=${FUZZ_DEC_PVQ4:-1} OPUS_AMD_TEST_TRPRE=1 timeout 2400 python tools/fuzz_sweep.py --which gpu --fuzzers fuzz_dec,fuzz_ms_dec --first ${FUZZ_FIRST:-9990000} --count ${FUZZ_DEC_COUNT:-1500} --pipeline 4 --workers 8 --log $O/fuzz_gpu_decoders.log > /dev/null 2>&1
  OPUS_AMD_TEST_TRPRE=1 timeout 3000 python tools/fuzz_sweep.py --which gpu --fuzzers fuzz,fuzz_sparse,fuzz_batch --first ${FUZZ_FIRST:-9990000} --count ${FUZZ_ENC_COUNT:-800} --pipeline 4 --workers 8 --log $O/fuzz_gpu_encoders_pipeline4.log > /dev/null 2>&1 ;;
dpvq) OPUS_AMD_DEC_PVQ4=1 timeout 500 python tools/dec_fast_check.py gpu > $O/dec_fast_check_pvq4.log 2>&1
  for m in 0 1; do for c in ${DPVQ_CONFIGS:-2 4}; do OPUS_AMD_DEC_PVQ4=$m timeout 220 python bench.py --steps 8 --warmup 2 --no-extra-configs --no-cpu-baseline --steady-state 0 --config $c --decode > $O/decode${c}_pvq4_$m.log 2>&1; done; done
  for c in ${DPVQ_CONFIGS:-2 4}; do (cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -f csv -d $OLDPWD/$O/profd$c -o p -- python $OLDPWD/bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-extra-configs --steady-state 0 --config $c --decode > $OLDPWD/$O/profd$c.log 2>&1); find $O/profd$c -name '*kernel_trace*' -delete; find $O/profd$c -name '*agent_info*' -delete; done ;;
dpvqprof) OPUS_AMD_PROF_PREBUILT=1 timeout 300 python tools/phase_profile_dpvq.py 16384 > $O/phases_dpvq_config2.txt 2>&1 ;;
deemph) for m in 0 1; do for c in ${DPVQ_CONFIGS:-2 4}; do OPUS_AMD_DEC_DEEMPH_LANE=$m timeout 220 python bench.py --steps 8 --warmup 2 --no-extra-configs --no-cpu-baseline --steady-state 0 --config $c --decode > $O/decode${c}_deemph_$m.log 2>&1; done; done ;;
decfast) timeout 220 python bench.py --steps 8 --warmup 2 --no-extra-configs --config 2 --decode > $O/decode2.log 2>&1; timeout 80 python tools/dec_fast_check.py gpu > $O/dec_fast_check.log 2>&1; timeout 100 python -m pytest tests/test_gpu_decoder.py tests/test_gpu_dec_fast.py -x -q --timeout 60 > $O/pytest_decoder.log 2>&1 ;;
phases) for m in 1 0; do for k in silk hybrid; do OPUS_AMD_PROF_PREBUILT=1 OPUS_AMD_SH_SPLIT=$m timeout 90 python tools/phase_profile_sh.py 16384 10 $k > $O/phases_${k}_split$m.txt 2>&1; done; done ;;
bench34p) for c in 3 4; do timeout 200 python bench.py --steps 10 --warmup 3 --no-extra-configs --config $c > $O/bench${c}_parity.log 2>&1; done ;;
broad) timeout ${BROAD_TIMEOUT:-190} python -m pytest tests -m gpu -q -n 6 --timeout 150 -k "not bench_ranks and not reference_programs" --durations=15 > $O/pytest_gpu_broad.log 2>&1 ;;
full) timeout 1200 python -m pytest tests -m gpu -x -q --timeout 600 > $O/pytest_gpu_full.log 2>&1 ;;
*) echo "unknown step $step" ;;
esac
done
for f in $O/*.log; do echo "== $f"; tail -n 4 "$f"; done

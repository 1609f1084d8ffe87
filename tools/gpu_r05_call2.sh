#!/bin/bash
# Round 5, second GPU call: fit the complex-shape decoder fixture, the MFMA clock probe with random operand data, the K1 ablation variants,
# the cluster / co-tenant tests again, and a rocprofv3 kernel trace of a short bench run split into legs (tools/rocpd_legs.py).
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r05b
mkdir -p $OUT $R/gpurun_out/fit
cd $R
timeout 200 tools/probes/mfma16_probe.bin 2.0 > $OUT/mfma16_probe.md 2>&1; echo "probe rc=$?"; cat $OUT/mfma16_probe.md
echo "| variant | kernel ms | MHz | TFLOP/s | frac of 157.3 | cycles per 64-pt tile | MFMA-bound / measured |" > $OUT/k1_ablation.md
echo "|---|---|---|---|---|---|---|" >> $OUT/k1_ablation.md
for v in main nobar nolds nodma noepi main; do
  if [ $v = main ]; then unset DSPGN_LIB; else export DSPGN_LIB=$R/dsp_slam_amd/lib/libdspgn_$v.so; fi
  timeout 120 python tools/probes/gpu_k1_ablation.py $v 2>> $OUT/k1_ablation.err | grep "^|" >> $OUT/k1_ablation.md
done
unset DSPGN_LIB
cat $OUT/k1_ablation.md
timeout 600 python -m pytest tests/test_gpu_round4.py tests/test_gpu_cotenant.py --maxfail=10 -q -m gpu -s -k "cluster or cotenant" > $OUT/cluster_tests.log 2>&1; echo "cluster tests rc=$?"
grep -E "narrow decoder|lost hand-off|co-tenant|passed|failed|Error|assert" $OUT/cluster_tests.log | head -30
cd /tmp && export TMPDIR=/tmp
timeout 400 rocprofv3 --kernel-trace -d /tmp/prof_legs -o legs -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --latency-runs 1 > $OUT/bench_under_rocprof.txt 2>&1; echo "rocprof rc=$?"
DB=$(find /tmp/prof_legs -name "*.db" | head -1)
python $R/tools/rocpd_legs.py $DB $OUT/bench_under_rocprof.txt $OUT/legs.md > /dev/null 2>$OUT/legs.err; tail -3 $OUT/legs.err; cat $OUT/legs.md | head -60
cd $R
timeout 900 python tools/fit_decoder_gpu.py --shape complex --name complex --steps 20000 --out gpurun_out/fit/decoder_complex.npz > $OUT/fit_complex.log 2>&1; echo "fit rc=$?"; tail -5 $OUT/fit_complex.log

#!/bin/bash
# Round 5, fourth GPU call: K0 scheduling A/B -- main (16x16x32, one wait per step, reads / epilogue dealt over the gaps), lpburst (epilogue in one
# burst), lpuse (per-use waits + burst: the first 16x16x32 form), lp32 (round 4's 32x32x16 kernel): bare kernel probe + exactness tests + short bench.
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r05d
mkdir -p $OUT
cd $R
for v in main lpburst lpuse lp32 main; do
  if [ $v = main ]; then unset DSPGN_LIB; else export DSPGN_LIB=$R/dsp_slam_amd/lib/libdspgn_$v.so; fi
  echo "== $v"; timeout 200 python tools/probes/gpu_prepass_probe.py 2>&1 | grep "^f16" | tee -a $OUT/probe_$v.log
done
unset DSPGN_LIB
bash tools/gpu_ab.sh lpburst lp32 2>&1 | tee $OUT/ab.log

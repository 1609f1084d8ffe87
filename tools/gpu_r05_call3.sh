#!/bin/bash
# Round 5, third GPU call: the prepass kernel on v_mfma_f32_16x16x32 (main) against the round-4 kernel on 32x32x16 (libdspgn_lp32.so, built
# from the same tree before the rewrite): exactness tests + a short bench each; the narrow-decoder cluster test; the cfg4 shard balance.
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r05c
mkdir -p $OUT
cd $R
timeout 300 python -m pytest tests/test_gpu_decoders.py tests/test_gpu_round4.py -q -m gpu -s -k "narrower or prepass or embedded" > $OUT/narrow.log 2>&1; echo "narrow rc=$?"; grep -E "narrow decoder|passed|failed|Error" $OUT/narrow.log | head
bash tools/gpu_ab.sh lp32 2>&1 | tee $OUT/ab.log
timeout 200 python tools/probes/gpu_prepass_probe.py > $OUT/prepass_probe_main.log 2>&1; tail -6 $OUT/prepass_probe_main.log
DSPGN_LIB=$R/dsp_slam_amd/lib/libdspgn_lp32.so timeout 200 python tools/probes/gpu_prepass_probe.py > $OUT/prepass_probe_lp32.log 2>&1; tail -6 $OUT/prepass_probe_lp32.log
timeout 600 python tools/gpu_cfg4_balance.py 1024 8 > $OUT/cfg4_balance.md 2> $OUT/cfg4_balance.err; echo "balance rc=$?"; cat $OUT/cfg4_balance.md | grep -E "slowest|^## "

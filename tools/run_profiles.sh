#!/bin/bash
# Profile sequence on the GPU box (one gpurun call): the bench lines, a rocprofv3 kernel trace of the DRIVER'S bench command split into legs,
# PMC passes (each in its own run, --pmc only), latency-sized kernel stats.  Outputs under gpurun_out/<tag>/; tools/make_profiles.py turns
# them into profiles/<tag>_*.md.
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=${1:-r06}
STEPS=${2:-20}
WARM=${3:-5}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd $R
timeout 900 python bench.py --steps $STEPS --warmup $WARM 2> $OUT/bench.err | tail -1 > $OUT/bench.json
timeout 900 python bench.py --steps 2 --warmup 1 --config cfg4 --no-cpu-baseline 2> $OUT/bench_cfg4.err | tail -1 > $OUT/bench_cfg4.json
timeout 400 python bench.py --steps 3 --warmup 1 --config cfg5 --no-cpu-baseline 2> $OUT/bench_cfg5.err | tail -1 > $OUT/bench_cfg5.json
cd /tmp && export TMPDIR=/tmp
# the command exactly as the driver runs it, under the profiler: its JSON line carries roofline.rocprof_check (per leg), tools/rocpd_legs.py
# splits the trace at the marker kernel's dispatches and lays both side by side
timeout 900 rocprofv3 --kernel-trace --stats -d /tmp/prof_stats_full -o stats -- python $R/bench.py --steps $STEPS --warmup $WARM > $OUT/bench_under_rocprof_full.txt 2>&1
DB=$(find /tmp/prof_stats_full -name "*.db" | head -1)
python $R/tools/rocpd_stats.py $DB > $OUT/kernel_stats_full.md 2>&1
python $R/tools/rocpd_legs.py $DB $OUT/bench_under_rocprof_full.txt $OUT/legs.md > /dev/null 2> $OUT/legs.err
for pass in "fetch:FETCH_SIZE" "write:WRITE_SIZE" "mfma:SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES" "lds:SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY"; do
  name=${pass%%:*}; ctrs=${pass#*:}
  timeout 300 rocprofv3 --pmc $ctrs -d /tmp/prof_pmc_$name -o pmc -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --latency-runs 1 --no-prepass-off > $OUT/pmc_$name.log 2>&1
  DB=$(find /tmp/prof_pmc_$name -name "*.db" | head -1)
  # markers: 1 headline_warmup, 2 headline_timed, 3 lp_compute_warmup, 4 lp_compute_timed, 5 restore_fp32_results, 6 clock_probe
  # -> [1, 3) the headline leg's two steps, [3, 5) the low-precision compute leg's two steps
  python $R/tools/rocpd_pmc.py $DB mlp_ --between 1 3 > $OUT/pmc_$name.md 2>&1
  python $R/tools/rocpd_pmc.py $DB mlp_ --between 3 5 > $OUT/pmc_lp_$name.md 2>&1
done
# latency-sized: one real-KITTI-size detection per call
timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_lat -o stats -- python $R/tools/gpu_small_loop.py 250 200 50 > $OUT/latency_run.txt 2>&1
DB=$(find /tmp/prof_lat -name "*.db" | head -1)
python $R/tools/rocpd_stats.py $DB > $OUT/latency_kernel_stats.md 2>&1
timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_lat2 -o stats -- python $R/tools/gpu_small_loop.py 2000 500 20 > $OUT/latency_cfg2_run.txt 2>&1
DB=$(find /tmp/prof_lat2 -name "*.db" | head -1)
python $R/tools/rocpd_stats.py $DB > $OUT/latency_cfg2_kernel_stats.md 2>&1
cd $R
timeout 600 python tools/gpu_latency_ab.py 15 > $OUT/latency_ab.txt 2>&1
cut -c1-600 $OUT/bench.json; head -30 $OUT/legs.md; tail -4 $OUT/latency_run.txt

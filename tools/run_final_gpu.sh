#!/bin/bash
# Round-end measurement sequence on the GPU box (one gpurun call): tests, bench, rocprofv3 kernel stats, three PMC passes, probes.
# Outputs under gpurun_out/final/ ; copy what is to be judged into profiles/.
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/final
mkdir -p $OUT
cd $R
timeout 600 python -m pytest tests -m gpu -q 2>&1 | tail -5 > $OUT/pytest_gpu.txt
timeout 300 python bench.py 2> $OUT/bench.err | tail -1 > $OUT/bench.json
timeout 100 python tools/gpu_mesh_probe.py > $OUT/mesh_probe.txt 2>&1
timeout 100 python tools/gpu_call_overhead.py > $OUT/call_overhead.txt 2>&1
timeout 100 python tools/gpu_split_probe.py > $OUT/split_probe.txt 2>&1
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_stats -o stats -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --latency-runs 3 > $OUT/bench_under_rocprof.txt 2>&1
DB=$(find /tmp/prof_stats -name "*.db" | head -1)
python $R/tools/rocpd_stats.py $DB > $OUT/kernel_stats.md 2>&1
for pass in "fetch:FETCH_SIZE" "write:WRITE_SIZE" "mfma:SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES"; do
  name=${pass%%:*}; ctrs=${pass#*:}
  timeout 300 rocprofv3 --pmc $ctrs -d /tmp/prof_pmc_$name -o pmc -- python $R/bench.py --steps 1 --warmup 1 --objects-per-gpu 32 --no-cpu-baseline --latency-runs 1 > $OUT/pmc_$name.log 2>&1
  DB=$(find /tmp/prof_pmc_$name -name "*.db" | head -1)
  python $R/tools/rocpd_pmc.py $DB mlp_ > $OUT/pmc_$name.md 2>&1
done
cd $R
tail -3 $OUT/pytest_gpu.txt; cat $OUT/bench.json | cut -c1-400; cat $OUT/mesh_probe.txt; tail -3 $OUT/call_overhead.txt

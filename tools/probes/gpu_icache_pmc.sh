#!/bin/bash
# GPU box: does the instruction cache matter for the decoder kernels?  Lists the SQC instruction-cache counters this device exposes and collects
# them (own --pmc run) over the bench's headline and low-precision legs.  bash tools/probes/gpu_icache_pmc.sh
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/icache
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L > $OUT/avail.txt 2>&1
grep -i -o "SQC_[A-Z_]*\|SQ_IFETCH[A-Z_]*\|SQ_INST_LEVEL[A-Z_]*\|SQ_WAIT_INST[A-Z_]*" $OUT/avail.txt | sort -u > $OUT/candidates.txt
cat $OUT/candidates.txt | tr '\n' ' '; echo
for ctrs in "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES" "SQ_IFETCH SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQC_ICACHE_MISSES_DUPLICATE SQC_ICACHE_INPUT_VALID_READYB"; do
  name=$(echo $ctrs | cut -d' ' -f1)
  timeout 300 rocprofv3 --pmc $ctrs -d /tmp/prof_ic_$name -o pmc -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --latency-runs 1 --no-prepass-off > $OUT/pmc_$name.log 2>&1
  DB=$(find /tmp/prof_ic_$name -name "*.db" | head -1)
  if [ -n "$DB" ]; then
    python $R/tools/rocpd_pmc.py $DB mlp_ --between 1 3 > $OUT/head_$name.md 2>&1
    python $R/tools/rocpd_pmc.py $DB mlp_ --between 3 5 > $OUT/lp_$name.md 2>&1
    grep "^| _ZN" $OUT/head_$name.md $OUT/lp_$name.md | cut -c1-220
  else
    tail -3 $OUT/pmc_$name.log
  fi
done

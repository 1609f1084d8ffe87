#!/bin/bash
# GPU box: the whole GPU suite (-x, as the driver runs it), smoke(), then the driver's bench command.
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/${1:-sb}
mkdir -p $OUT
cd $R
timeout 2400 python -m pytest tests -x -q -m gpu > $OUT/gpu_tests.log 2>&1; echo "gpu tests rc=$?"; tail -3 $OUT/gpu_tests.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; echo "smoke rc=$?"; tail -1 $OUT/smoke.log
timeout 900 python bench.py --steps 20 --warmup 5 > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"
python - <<PY
import json
r = json.loads(open("$OUT/bench.json").read().strip().splitlines()[-1])
print({k: r.get(k) for k in ("value", "ms_per_step", "value_fp32_only", "value_lp")}, r["roofline"]["ms_per_step_by_kernel"], r["lp_compute"]["roofline"]["ms_per_step_by_kernel"])
PY

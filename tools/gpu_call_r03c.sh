set -u
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03c
timeout 1200 python -m pytest tests -m gpu -q > gpurun_out/r03c/gpu_tests.log 2>&1; tail -4 gpurun_out/r03c/gpu_tests.log
python tools/gpu_power_probe.py 4 > gpurun_out/r03c/power_probe.jsonl 2> gpurun_out/r03c/power_probe.err; cut -c1-420 gpurun_out/r03c/power_probe.jsonl
DSP_BENCH_FORCE_DIST=1 timeout 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --latency-runs 1 2> gpurun_out/r03c/bench_force_dist.err | tail -1 > gpurun_out/r03c/bench_force_dist.json; cut -c1-300 gpurun_out/r03c/bench_force_dist.json
python examples/export_example_data.py /tmp/dsp_example 16 250 100 > /dev/null && ./examples/multi_gpu_c_abi /tmp/dsp_example > gpurun_out/r03c/multi_gpu_c_abi.txt 2>&1; cat gpurun_out/r03c/multi_gpu_c_abi.txt
rocm-smi --showpower --showclocks --json > gpurun_out/r03c/rocm_smi_idle.json 2>&1

"""examples/multi_gpu_c_abi.cpp: a plain C++ program on the C ABI -- one dsp_handle per GPU, one host thread per handle, ONE RCCL gather
straight from the device-resident batches (dsp_gather_batch_results) -- must return, bit for bit, what one GPU returns for all objects.
The GPU boxes of this pool have one GPU (a communicator of one rank); on an 8-GPU node the same binary shards over all of them."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "examples", "multi_gpu_c_abi")


def _build():
    src = EXE + ".cpp"
    lib = os.path.join(ROOT, "dsp_slam_amd", "lib", "libdspgn.so")
    if not os.path.exists(EXE) or os.path.getmtime(EXE) < max(os.path.getmtime(src), os.path.getmtime(lib)):
        subprocess.check_call([os.path.join(ROOT, "examples", "build.sh")])


def test_example_builds_cpu():
    from dsp_slam_amd import _lib
    _lib.load()
    _build()
    assert os.access(EXE, os.X_OK)


@pytest.mark.gpu
def test_multi_gpu_c_abi_example(tmp_path):
    _build()
    subprocess.check_call(["python3", os.path.join(ROOT, "examples", "export_example_data.py"), str(tmp_path), "12", "250", "100"])
    out = subprocess.run([EXE, str(tmp_path)], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=600)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    last = [line for line in out.stdout.splitlines() if "sharded == single GPU" in line]
    assert last and last[0].endswith("yes") and "12 objects" in last[0], out.stdout
    good = int(last[0].split(":")[1].split("good")[0])
    assert good >= 10, out.stdout
    print(out.stdout)


@pytest.mark.gpu
def test_bench_launch_path_with_two_ranks_on_one_gpu(tmp_path):
    """The driver launches the multi-GPU bench as `python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1
    --master-port P bench.py --gpus N ...`.  This box has one GPU, so the LAUNCH PATH is exercised with both ranks on device 0 and the
    gather on gloo (DSP_BENCH_SHARE_GPU / DSP_BENCH_BACKEND: plumbing switches, the numbers mean nothing): rendezvous, RANK / LOCAL_RANK /
    WORLD_SIZE handling, two engines created concurrently (build lock), per-rank shards, gather order, barrier + max-over-ranks timing, ONE
    JSON line from rank 0 with n_gpus = 2 and both ranks' times."""
    import json
    import socket
    import sys
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ, DSP_BENCH_SHARE_GPU="1", DSP_BENCH_BACKEND="gloo")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", str(port),
           os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--objects-per-gpu", "3", "--no-prepass-off"]
    out = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=900, env=env, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith('{"metric"')]
    assert len(lines) == 1, out.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 2 and len(d["ms_per_step_by_rank"]) == 2 and d["value"] > 0
    assert d["config"]["objects_good"] >= 2 and "PLUMBING TEST" in d["config"]["parallelism"]


@pytest.mark.gpu
@pytest.mark.parametrize("partition", ["static", "measured"])
def test_cfg4_strong_scaling_job_with_two_ranks_on_one_gpu(partition):
    """BASELINE configs[3] as bench.py runs it under the driver's launch command: a FIXED object list (here 10 objects instead of 1024) cut
    into per-rank shards by distributed.shard_objects -- by the static cost (default) or by the measured first-iteration cost, whose per-rank
    measurements are all-gathered before the cut -- then one gather of the result rows per step.  Both ranks on device 0, gloo (plumbing
    switches: the numbers mean nothing): `scaling` is "strong", the shards cover the list, every object comes back good."""
    import json
    import socket
    import sys
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ, DSP_BENCH_SHARE_GPU="1", DSP_BENCH_BACKEND="gloo")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", str(port),
           os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "1", "--config", "cfg4", "--total-objects", "10", "--partition", partition]
    out = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=900, env=env, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith('{"metric"')]
    assert len(lines) == 1, out.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["scaling"] == "strong" and d["config"]["name"] == "cfg4"
    assert sum(r["objects"] for r in d["by_rank"]) == 10 and d["config"]["objects_good"] == 10
    assert partition in d["config"]["workload"]
    assert all(r["sum_V_per_step"] > 0 for r in d["by_rank"])


@pytest.mark.gpu
@pytest.mark.parametrize("mode", ["atexit", "reverse"])
def test_a_process_that_exits_with_live_objects_exits(mode):
    """A process that ends with an Engine and a Batch still alive (a test that failed half way, a script without close()) must END: finalisers
    run in no particular order at interpreter exit, and dsp_batch_destroy after dsp_destroy used to lock a mutex inside the freed handle --
    a hang that took a whole gpurun call with it (profiles/r05_cluster_exchange_stores.md).  `atexit`: both are left to the atexit handler of
    dsp_slam_amd.engine (batches first).  `reverse`: dsp_destroy(handle) first, dsp_batch_destroy(batch) after it -- the library takes a
    handle's live batches with it and ignores the late call.  In a child process, under a time limit."""
    import sys
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "gpu_exit_leak.py"), mode], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=240)
    assert out.returncode == 0, out.stdout[-2000:]
    assert "results ok" in out.stdout

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

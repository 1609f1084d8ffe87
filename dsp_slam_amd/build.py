"""Build libdspgn.so (HIP, gfx950 only) in-tree:  python -m dsp_slam_amd.build [--force]"""
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CSRC = os.path.join(HERE, "csrc")
LIB_DIR = os.path.join(HERE, "lib")
LIB_PATH = os.path.join(LIB_DIR, "libdspgn.so")
SOURCES = ["mlp_kernel.hip", "mlp_split_kernel.hip", "mlp_lp_kernel.hip", "gn_kernels.hip", "mesh_kernels.hip", "dsp_gn.hip", "pose_graph.cpp"]
HEADERS = [os.path.join(CSRC, "dsp_internal.h"), os.path.join(CSRC, "mlp_common.h"), os.path.join(ROOT, "include", "dsp_gn.h"),
           os.path.join(ROOT, "include", "dsp_pose_graph.h")]


def _hipcc():
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found (needs ROCm >= 7.0)")


def is_stale():
    if not os.path.exists(LIB_PATH):
        return True
    t = os.path.getmtime(LIB_PATH)
    deps = [os.path.join(CSRC, s) for s in SOURCES] + HEADERS + [os.path.abspath(__file__)]
    return any(os.path.getmtime(d) > t for d in deps)


# Bookkeeping and mesh kernels restate fp32 formulas of the reference operation by operation (point transforms, depth samples,
# edge interpolation): no fused multiply-add may be formed where the reference rounds twice.  The `__fmul_rn`/`__fadd_rn`
# spellings do not prevent that on their own (they are plain `*`/`+` in clang's HIP headers), so these files are compiled with
# contraction off; explicit fmaf() calls stay FMAs.  mlp_kernel.hip keeps the default (its MFMA / fmaf use is explicit anyway).
EXTRA_FLAGS = {"gn_kernels.hip": ["-ffp-contract=off"], "mesh_kernels.hip": ["-ffp-contract=off"],
               "pose_graph.cpp": ["-ffp-contract=off"]}
OBJ_DIR = os.path.join(LIB_DIR, "obj")


def _compile_one(hipcc, src, verbose):
    obj = os.path.join(OBJ_DIR, src + ".o")
    deps = [os.path.join(CSRC, src)] + HEADERS + [os.path.abspath(__file__)]
    if os.path.exists(obj) and all(os.path.getmtime(d) <= os.path.getmtime(obj) for d in deps):
        return obj
    cmd = [hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-c", "-Wno-unused-value",
           "-I" + os.path.join(ROOT, "include"), "-I" + CSRC] + EXTRA_FLAGS.get(src, []) + [os.path.join(CSRC, src), "-o", obj + ".tmp"]
    if verbose:
        print(" ".join(cmd))
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if r.returncode != 0:
        raise RuntimeError("hipcc failed on %s:\n%s" % (src, r.stdout))
    os.replace(obj + ".tmp", obj)
    return obj


def build(force=False, verbose=False):
    """Compile the HIP kernels + C ABI for gfx950 into dsp_slam_amd/lib/libdspgn.so."""
    if not force and not is_stale():
        return LIB_PATH
    os.makedirs(OBJ_DIR, exist_ok=True)
    if force:
        for f in os.listdir(OBJ_DIR):
            os.remove(os.path.join(OBJ_DIR, f))
    hipcc = _hipcc()
    from concurrent.futures import ThreadPoolExecutor
    with ThreadPoolExecutor(max_workers=len(SOURCES)) as ex:
        objs = list(ex.map(lambda s: _compile_one(hipcc, s, verbose), SOURCES))
    cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC"] + objs + ["-o", LIB_PATH + ".tmp"]
    if verbose:
        print(" ".join(cmd))
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if r.returncode != 0:
        raise RuntimeError("hipcc link failed:\n" + r.stdout)
    os.replace(LIB_PATH + ".tmp", LIB_PATH)
    return LIB_PATH


def build_locked(force=False, verbose=False):
    """build() under an inter-process lock: with one process per GPU every rank of a fresh checkout would otherwise run hipcc
    into the same temporary files at once.  The first holder builds; the others block, then see a fresh library."""
    import fcntl
    os.makedirs(LIB_DIR, exist_ok=True)
    with open(os.path.join(LIB_DIR, ".build.lock"), "w") as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        try:
            if force or is_stale():
                return build(force=force, verbose=verbose)
            return LIB_PATH
        finally:
            fcntl.flock(lock, fcntl.LOCK_UN)


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))

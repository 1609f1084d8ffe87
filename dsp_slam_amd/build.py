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
SOURCES = ["mlp_kernel.hip", "mlp_split_kernel.hip", "mlp_cluster_kernel.hip", "mlp_lp_kernel.hip", "mlp_lpj_kernel.hip", "gn_kernels.hip", "mesh_kernels.hip", "dsp_gn.hip",
           "pose_graph.cpp"]
HEADERS = [os.path.join(CSRC, "dsp_internal.h"), os.path.join(CSRC, "mlp_common.h"), os.path.join(CSRC, "mlp_lp_common.h"), os.path.join(ROOT, "include", "dsp_gn.h"),
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
    if src == "dsp_gn.hip":
        deps.append(os.path.join(LIB_DIR, "build_info.h"))
    if os.path.exists(obj) and all(os.path.getmtime(d) <= os.path.getmtime(obj) for d in deps):
        return obj
    cmd = [hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-c", "-Wno-unused-value",
           "-I" + os.path.join(ROOT, "include"), "-I" + CSRC, "-I" + LIB_DIR] + EXTRA_FLAGS.get(src, []) + [os.path.join(CSRC, src), "-o", obj + ".tmp"]
    if verbose:
        print(" ".join(cmd))
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if r.returncode != 0:
        raise RuntimeError("hipcc failed on %s:\n%s" % (src, r.stdout))
    os.replace(obj + ".tmp", obj)
    return obj


# ---- ISA assumptions, enforced where the library is BUILT ----------------------------------------------------------------------------
# The decoder kernels rest on three properties of the generated code that hipcc does not promise (DESIGN.md "K1"):
#   * M0 (the LDS-DMA destination) is written ONLY by mlp_common.h's helpers (`s_mov_b32 m0, sN` + hazard `s_nop`) and never by hipcc:
#     the helpers leave it set across the k-steps that issue a chunk's pieces and do not restore it;
#   * the fp32 decoder kernels and the f16 prepass kernel keep their activation slabs in registers: no scratch memory, no spills;
#   * the LDS-DMA loads are really there (global_load_lds_dwordx4), i.e. the inline asm was not dropped.
# A different hipcc may break any of them silently (wrong results for M0, a 10x slowdown for scratch), so the build FAILS instead.
OBJDUMP_CANDIDATES = ("/opt/rocm/lib/llvm/bin/llvm-objdump", "/opt/rocm/llvm/bin/llvm-objdump")
READELF_CANDIDATES = ("/opt/rocm/lib/llvm/bin/llvm-readelf", "/opt/rocm/llvm/bin/llvm-readelf")
NO_SCRATCH_KERNELS = ("mlp_kernelILi0", "mlp_kernelILi1", "mlp_kernelILi2", "mlp_kernelILi3", "mlp_split_kernel", "mlp_cluster_kernel", "mlp_lp_kernelILb0",
                      "mlp_lpj_fwd_kernel", "mlp_lpj_bwd_kernel")


class IsaCheckError(RuntimeError):
    pass


def check_isa(verbose=False):
    """Disassemble the gfx950 code objects of the decoder kernels and check the assumptions above.  Returns a dict of what was counted;
    raises IsaCheckError on a violation, RuntimeError when the LLVM tools are missing (set DSP_SKIP_ISA_CHECK=1 to build anyway)."""
    import tempfile
    objdump = next((c for c in OBJDUMP_CANDIDATES if os.path.exists(c)), None)
    readelf = next((c for c in READELF_CANDIDATES if os.path.exists(c)), None)
    if not objdump or not readelf:
        raise RuntimeError("llvm-objdump / llvm-readelf not found: cannot check the ISA assumptions of the decoder kernels")
    report = {"m0_writes": 0, "lds_dma_loads": 0, "kernels": {}}
    for src in ("mlp_kernel.hip", "mlp_lp_kernel.hip", "mlp_lpj_kernel.hip", "mlp_split_kernel.hip", "mlp_cluster_kernel.hip"):
        obj = os.path.join(OBJ_DIR, src + ".o")
        with tempfile.TemporaryDirectory(prefix="dsp_isa_") as work:
            shutil.copy(obj, os.path.join(work, "k.o"))
            subprocess.run([objdump, "--offloading", "k.o"], cwd=work, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, check=True)
            co = [f for f in os.listdir(work) if "gfx950" in f]
            if len(co) != 1:
                raise IsaCheckError("%s: expected one gfx950 code object, found %s" % (src, co))
            dis = subprocess.run([objdump, "-d", co[0]], cwd=work, stdout=subprocess.PIPE, text=True, check=True).stdout.splitlines()
            notes = subprocess.run([readelf, "--notes", co[0]], cwd=work, stdout=subprocess.PIPE, text=True, check=True).stdout.splitlines()
        ins = [ln.split("//")[0].split() for ln in dis if "\t" in ln and not ln.rstrip().endswith(":")]
        ins = [t for t in ins if t]
        for i, t in enumerate(ins):
            if any(x.rstrip(",") == "m0" for x in t):
                if not (t[0] == "s_mov_b32" and t[1].rstrip(",") == "m0" and i + 1 < len(ins) and ins[i + 1][0] == "s_nop"):
                    raise IsaCheckError("%s: M0 is touched outside the LDS-DMA helpers: `%s`" % (src, " ".join(t)))
                report["m0_writes"] += 1
        loads = sum(1 for t in ins if t[0] == "global_load_lds_dwordx4")
        if loads < (64 if src == "mlp_cluster_kernel.hip" else 400):       # the cluster form is ONE kernel with one unrolled chunk loop (16 + 64 loads)
            raise IsaCheckError("%s: only %d global_load_lds_dwordx4 instructions (the LDS-DMA stream is missing)" % (src, loads))
        report["lds_dma_loads"] += loads
        cur = {}
        for ln in notes:          # AMDGPU metadata: one block of `.key: value` lines per kernel
            ln = ln.strip().lstrip("- ").strip()
            for key in (".name", ".private_segment_fixed_size", ".vgpr_count", ".agpr_count", ".vgpr_spill_count", ".sgpr_spill_count"):
                if ln.startswith(key + ":"):
                    cur[key] = ln.split(":", 1)[1].strip()
            if ".name" in cur and ".private_segment_fixed_size" in cur and ".vgpr_count" in cur and ".vgpr_spill_count" in cur and (
                    ".agpr_count" in cur) and ".sgpr_spill_count" in cur:
                report["kernels"][cur[".name"]] = {k[1:]: int(v) for k, v in cur.items() if k != ".name"}
                cur = {}
    for name, k in report["kernels"].items():
        if any(tag in name for tag in NO_SCRATCH_KERNELS) and (k["private_segment_fixed_size"] or k["vgpr_spill_count"]):
            raise IsaCheckError("%s uses scratch memory (%d B, %d spilled VGPRs): the activation slabs no longer live in registers" % (
                name, k["private_segment_fixed_size"], k["vgpr_spill_count"]))
    if report["m0_writes"] < 250 or len(report["kernels"]) < 8:
        raise IsaCheckError("ISA check saw too little: %d M0 writes, %d kernels" % (report["m0_writes"], len(report["kernels"])))
    if verbose:
        print("ISA check: %d M0 writes (all by the LDS-DMA helpers), %d LDS-DMA loads, %d decoder kernels without scratch" % (
            report["m0_writes"], report["lds_dma_loads"], len(report["kernels"])))
    return report


def hipcc_version(hipcc):
    r = subprocess.run([hipcc, "--version"], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    lines = [ln.strip() for ln in r.stdout.splitlines() if ln.strip()]
    return " | ".join(lines[:2]) if lines else "unknown"


def build(force=False, verbose=False):
    """Compile the HIP kernels + C ABI for gfx950 into dsp_slam_amd/lib/libdspgn.so."""
    if not force and not is_stale():
        return LIB_PATH
    os.makedirs(OBJ_DIR, exist_ok=True)
    if force:
        for f in os.listdir(OBJ_DIR):
            os.remove(os.path.join(OBJ_DIR, f))
    hipcc = _hipcc()
    # what compiled this library travels inside it (dsp_build_info): the GPU tests log it next to the runtime's version
    info = os.path.join(LIB_DIR, "build_info.h")
    text = '#define DSP_BUILD_HIPCC "%s"\n' % hipcc_version(hipcc).replace("\\", "/").replace('"', "'")
    if not os.path.exists(info) or open(info).read() != text:
        with open(info, "w") as f:
            f.write(text)
    from concurrent.futures import ThreadPoolExecutor
    with ThreadPoolExecutor(max_workers=len(SOURCES)) as ex:
        objs = list(ex.map(lambda s: _compile_one(hipcc, s, verbose), SOURCES))
    if os.environ.get("DSP_SKIP_ISA_CHECK") != "1":
        check_isa(verbose)          # before linking: a library that violates the assumptions is never produced
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


def build_variant(name, flags, verbose=False):
    """Development aid for A/B measurements inside one gpurun call: the same sources compiled with extra hipcc flags (e.g. -DLP_ABL_NOBAR)
    into lib/libdspgn_<name>.so (own object directory; same ISA checks).  Select it at run time with DSPGN_LIB=<path> (dsp_slam_amd/_lib.py)."""
    global OBJ_DIR, LIB_PATH
    saved = (OBJ_DIR, LIB_PATH, dict(EXTRA_FLAGS))
    try:
        OBJ_DIR = os.path.join(LIB_DIR, "obj_" + name)
        LIB_PATH = os.path.join(LIB_DIR, "libdspgn_%s.so" % name)
        for src in SOURCES:
            EXTRA_FLAGS[src] = EXTRA_FLAGS.get(src, []) + list(flags)
        return build(force=True, verbose=verbose)
    finally:
        OBJ_DIR, LIB_PATH = saved[0], saved[1]
        EXTRA_FLAGS.clear()
        EXTRA_FLAGS.update(saved[2])


if __name__ == "__main__":
    if "--variant" in sys.argv:        # python -m dsp_slam_amd.build --variant NAME -DFLAG ...
        i = sys.argv.index("--variant")
        print(build_variant(sys.argv[i + 1], sys.argv[i + 2:], verbose=False))
        sys.exit(0)
    print(build(force="--force" in sys.argv, verbose=True))

"""CPU-only: the C-ABI library loads and exports every symbol include/*.h declares; the product path fails
loudly (no fallback) when there is no GPU."""
import os
import re

import pytest

from dsp_slam_amd import _lib as L, engine as E, fixtures

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_symbols():
    inc = os.path.join(ROOT, "include")
    src = "".join(open(os.path.join(inc, f)).read() for f in sorted(os.listdir(inc)) if f.endswith(".h"))
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(dsp_[a-z0-9_]+)\s*\(", src)))


def test_library_exports_every_declared_symbol():
    lib = L.load()
    names = header_symbols()
    assert len(names) >= 16 and "dsp_pg_edge_error" in names and "dsp_create" in names
    for n in names:
        assert hasattr(lib, n), "libdspgn.so does not export %s" % n
    bound = {n for n, _, _ in L.SYMBOLS}
    assert set(names) <= bound, "ctypes binding misses %s" % (set(names) - bound)
    assert lib.dsp_abi_version() == L.ABI_VERSION == 6    # 2: dsp_stats grew the prepass fields; 3: the guard fields; 4: kernel-timing setter, partial guard re-run; 5: dsp_debug_lie, dsp_trim, cluster time-out; 6: batch tokens, five setters + dsp_batch_set_debug
    setters = [n for n in names if n.startswith("dsp_batch_set_")]
    assert sorted(setters) == ["dsp_batch_set_compute", "dsp_batch_set_debug", "dsp_batch_set_iterations", "dsp_batch_set_kernel_timing", "dsp_batch_set_prepass", "dsp_batch_set_prepass_guard",
                               "dsp_batch_set_ray_passes"], setters


def test_gfx950_code_object_present():
    """The shared object carries a gfx950 device code object (hipcc --offload-arch=gfx950)."""
    data = open(L.lib_path(), "rb").read()
    assert b"gfx950" in data and b"mlp_kernel" in data


def test_no_silent_fallback_without_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from dsp_slam_amd.deep_sdf.deep_sdf_decoder import fold_weight_norm
    layers = fold_weight_norm(fixtures.random_state_dict(0), 9)
    with pytest.raises(L.DspError):
        E.Engine(layers, [4], 64, device=0)


def test_product_never_imports_oracle():
    pkg = os.path.join(ROOT, "dsp_slam_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                txt = open(os.path.join(dirpath, f)).read()
                assert "dsp_oracle" not in txt and "from oracle" not in txt and "import oracle" not in txt, f


def test_isa_assumptions_hold_on_the_built_code_objects():
    """The decoder kernels rely on three properties of the generated code (M0 written only by the LDS-DMA helpers and never restored,
    no scratch memory in the register-resident kernels, the LDS-DMA loads present).  dsp_slam_amd/build.py checks them by disassembly
    BEFORE it links -- a build with a hipcc that breaks one of them fails -- and this test re-runs the same check on the objects the
    loaded library was linked from, so the CPU tier shows its numbers."""
    from dsp_slam_amd import build as B
    L.load()        # builds (and checks) if stale
    if not os.path.exists(os.path.join(B.OBJ_DIR, "mlp_kernel.hip.o")):
        pytest.skip("object files not kept on this box")
    rep = B.check_isa()
    assert rep["m0_writes"] > 250 and rep["lds_dma_loads"] >= 1200
    fp32 = [k for k in rep["kernels"] if "mlp_kernelILi" in k]
    assert len(fp32) == 4 and all(rep["kernels"][k]["private_segment_fixed_size"] == 0 for k in fp32)
    assert b"ISA assumptions checked at build time" in L.load().dsp_build_info()


def test_isa_check_rejects_a_violation(tmp_path, monkeypatch):
    """The check is not vacuous: pointed at the bookkeeping kernels (which do use hipcc-managed registers freely and carry no LDS-DMA
    stream) it must refuse."""
    from dsp_slam_amd import build as B
    if not os.path.exists(os.path.join(B.OBJ_DIR, "gn_kernels.hip.o")):
        pytest.skip("object files not kept on this box")
    import shutil
    fake = tmp_path / "obj"
    fake.mkdir()
    for name in ("mlp_kernel.hip.o", "mlp_lp_kernel.hip.o", "mlp_split_kernel.hip.o"):
        shutil.copy(os.path.join(B.OBJ_DIR, "gn_kernels.hip.o"), fake / name)
    monkeypatch.setattr(B, "OBJ_DIR", str(fake))
    with pytest.raises(B.IsaCheckError):
        B.check_isa()

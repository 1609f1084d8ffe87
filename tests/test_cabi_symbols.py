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
    assert lib.dsp_abi_version() == 2    # 2: dsp_stats grew the prepass fields


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

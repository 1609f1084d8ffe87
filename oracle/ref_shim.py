"""Import shim for running the UNMODIFIED reference Python hot path on CPU.

TEST INFRASTRUCTURE ONLY.  Used by tools/make_golden.py (in the build container, where
/root/reference exists) to generate the golden vectors under tests/golden/ and by the
optional `test_oracle_vs_live_reference` test.  Nothing in the product path imports this.

What it does (no reference file is modified or copied):
  * puts /root/reference on sys.path so `reconstruct` / `deep_sdf` resolve to the reference;
  * stubs the three third-party imports of reconstruct/utils.py:21-23 that are absent here
    (`addict`, `plyfile`, `skimage.measure`) -- only `addict.Dict` is used on the hot path
    (ForceKeyErrorDict, reconstruct/utils.py:82-84);
  * neutralises the hard-coded `.cuda()` calls (reconstruct/loss.py:32,63,84,
    reconstruct/optimizer.py:56-57,98-114, ...) when no GPU is visible.
"""
import os
import sys
import types

REFERENCE_ROOT = os.environ.get("DSP_REFERENCE_ROOT", "/root/reference")


def reference_available():
    return os.path.isdir(os.path.join(REFERENCE_ROOT, "reconstruct"))


class _Dict(dict):
    """Minimal stand-in for addict.Dict (attribute access, nested dict promotion)."""

    def __init__(self, *args, **kwargs):
        super().__init__()
        for a in args:
            for k, v in dict(a).items():
                self[k] = v
        for k, v in kwargs.items():
            self[k] = v

    def __setitem__(self, k, v):
        if isinstance(v, dict) and not isinstance(v, _Dict):
            v = type(self)(v)
        super().__setitem__(k, v)

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError:
            return self.__missing__(k)

    def __setattr__(self, k, v):
        self[k] = v

    def __missing__(self, k):
        v = _Dict()
        self[k] = v
        return v


_installed = False


def install(force_cpu=False):
    """Make `import reconstruct...` / `import deep_sdf...` resolve to the reference.
    force_cpu: neutralise the reference's `.cuda()` calls even when a GPU is visible (the CPU-baseline leg of bench.py)."""
    global _installed
    if _installed:
        return
    if not reference_available():
        raise RuntimeError("reference tree not found at %s" % REFERENCE_ROOT)
    import torch

    # our own mirror packages must not shadow the reference inside this process
    for name in list(sys.modules):
        if name == "reconstruct" or name.startswith("reconstruct.") or \
           name == "deep_sdf" or name.startswith("deep_sdf."):
            del sys.modules[name]
    sys.path.insert(0, REFERENCE_ROOT)

    m = types.ModuleType("addict")
    m.Dict = _Dict
    sys.modules["addict"] = m
    sys.modules["plyfile"] = types.ModuleType("plyfile")
    sk = types.ModuleType("skimage")
    sk.measure = types.ModuleType("skimage.measure")
    sys.modules["skimage"] = sk
    sys.modules["skimage.measure"] = sk.measure

    if force_cpu or not torch.cuda.is_available():
        torch.Tensor.cuda = lambda self, *a, **k: self
        torch.nn.Module.cuda = lambda self, *a, **k: self
        torch.cuda.synchronize = lambda *a, **k: None
    _installed = True

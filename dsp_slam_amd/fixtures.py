"""Decoder fixtures: compact storage <-> the reference's on-disk DeepSDF format.

The reference loads `<dir>/specs.json` + `<dir>/ModelParameters/latest.pth`, a dict whose
"model_state_dict" was saved from nn.DataParallel, i.e. keys `module.linK.weight_g|weight_v|bias`
and `module.lin8.weight|bias` (deep_sdf/workspace.py:202-223).
The synthetic fixtures are kept in tests/golden/ as .npz with weight_v stored as bf16 bit
patterns (the values ARE exactly bf16-representable fp32 numbers: they were rounded before the
golden vectors were generated, so nothing is lost on reload).
"""
import json
import os

import numpy as np

# Upstream DeepSDF example spec with CodeLength 64 (assumed architecture)
SPECS = {
    "NetworkArch": "deep_sdf_decoder",
    "CodeLength": 64,
    "NetworkSpecs": {
        "dims": [512] * 8,
        "dropout": [0, 1, 2, 3, 4, 5, 6, 7],
        "dropout_prob": 0.2,
        "norm_layers": [0, 1, 2, 3, 4, 5, 6, 7],
        "latent_in": [4],
        "xyz_in_all": False,
        "use_tanh": False,
        "latent_dropout": False,
        "weight_norm": True,
    },
}

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")


def _round_to_bf16(a):
    """fp32 -> nearest-even bf16, returned as (uint16 bits, fp32 value)."""
    a = np.ascontiguousarray(a, dtype=np.float32)
    u = a.view(np.uint32).astype(np.uint64)
    u = u + 0x7FFF + ((u >> 16) & 1)
    bits = (u >> 16).astype(np.uint16)
    val = (bits.astype(np.uint32) << 16).view(np.float32)
    return bits, val


def save_decoder_npz(state_dict, path, code_len=None):
    """state_dict: torch or numpy tensors keyed as Decoder.state_dict() (no `module.` prefix)."""
    out = {}
    if code_len is not None and code_len != SPECS["CodeLength"]:
        out["meta:code_len"] = np.array(code_len, np.int32)
    for k, v in state_dict.items():
        a = v.detach().cpu().numpy() if hasattr(v, "detach") else np.asarray(v)
        if k.endswith("weight_v") or k == "lin8.weight":
            bits, _ = _round_to_bf16(a)
            out[k + ":bf16"] = bits
        else:
            out[k] = a.astype(np.float32)
    np.savez_compressed(path, **out)


def load_decoder_npz(path):
    """-> dict name -> float32 ndarray, keyed as Decoder.state_dict()."""
    z = np.load(path)
    sd = {}
    for k in z.files:
        if k.startswith("meta:"):
            continue
        if k.endswith(":bf16"):
            sd[k[:-5]] = (z[k].astype(np.uint32) << 16).view(np.float32)
        else:
            sd[k] = z[k].astype(np.float32)
    return sd


def fixture_specs(name_or_npz):
    """specs.json content of a fixture: the upstream example spec with the fixture's CodeLength."""
    path = name_or_npz if os.path.isfile(name_or_npz) else fixture_path(name_or_npz)
    z = np.load(path)
    specs = json.loads(json.dumps(SPECS))
    if "meta:code_len" in z.files:
        specs["CodeLength"] = int(z["meta:code_len"])
    return specs


def fixture_path(name):
    return os.path.join(GOLDEN_DIR, "decoder_%s.npz" % name)


def materialize_decoder_dir(name_or_npz, out_dir, specs=None):
    """Write the reference's on-disk layout for a fixture; returns out_dir (a DeepSDF_DIR)."""
    import torch

    path = name_or_npz if os.path.isfile(name_or_npz) else fixture_path(name_or_npz)
    sd = load_decoder_npz(path)
    os.makedirs(os.path.join(out_dir, "ModelParameters"), exist_ok=True)
    with open(os.path.join(out_dir, "specs.json"), "w") as f:
        json.dump(specs or fixture_specs(path), f, indent=2)
    tsd = {"module." + k: torch.from_numpy(np.ascontiguousarray(v)) for k, v in sd.items()}
    torch.save({"epoch": 0, "model_state_dict": tsd},
               os.path.join(out_dir, "ModelParameters", "latest.pth"))
    return out_dir


def random_state_dict(seed, specs=None):
    """Random-init decoder weights (PyTorch-default-like uniform init), numpy only.

    Used for mechanics tests and the failure path (no zero crossing => K = 0 => is_good False).
    """
    specs = specs or SPECS
    rng = np.random.default_rng(seed)
    ns = specs["NetworkSpecs"]
    d0 = specs["CodeLength"] + 3
    dims = [d0] + list(ns["dims"]) + [1]      # any CodeLength / hidden widths the spec names
    sd = {}
    for layer in range(len(dims) - 1):
        out_dim = dims[layer + 1] - d0 if (layer + 1) in ns["latent_in"] else dims[layer + 1]
        in_dim = dims[layer]
        bound = 1.0 / np.sqrt(in_dim)
        w = rng.uniform(-bound, bound, size=(out_dim, in_dim)).astype(np.float32)
        b = rng.uniform(-bound, bound, size=(out_dim,)).astype(np.float32)
        if ns["weight_norm"] and layer in ns["norm_layers"]:
            _, w = _round_to_bf16(w)
            sd["lin%d.weight_g" % layer] = np.linalg.norm(w, axis=1, keepdims=True).astype(np.float32)
            sd["lin%d.weight_v" % layer] = w
        else:
            _, w = _round_to_bf16(w)
            sd["lin%d.weight" % layer] = w
        sd["lin%d.bias" % layer] = b
    return sd

#!/usr/bin/env python3
"""Development aid: compare every pass's activation slab on the GPU with the oracle's layer activations."""
import os, sys, ctypes as C
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import dsp_oracle as O
from dsp_slam_amd import fixtures, engine as E, _lib as L

dec = O.fold_decoder(fixtures.load_decoder_npz(fixtures.fixture_path("cars")), fixtures.SPECS)
eng = E.Engine(dec.layers, dec.latent_in, dec.code_len, device=0)
lib = L.load()
lib.dsp_debug_slabs.restype = C.c_int
lib.dsp_debug_slabs.argtypes = [C.c_void_p, L.c_f32p, L.c_f32p, C.c_int, L.c_f32p, L.c_f32p]
rng = np.random.default_rng(0)
code = (rng.normal(size=64) * 0.2).astype(np.float32)
n = 64
pts = rng.uniform(-1, 1, size=(n, 3)).astype(np.float32)
slabs = np.zeros((16, 4, 128, 64), np.float32)
grad = np.zeros((n, 68), np.float32)
L.check(lib.dsp_debug_slabs(eng._h, L.ptr(code), L.ptr(pts), n, L.ptr(slabs), L.ptr(grad)), eng._h, "dbg")
x = np.concatenate([np.broadcast_to(code, (n, 64)), pts], -1).astype(np.float32)
y, pre = O.decoder_forward(dec, x, keep=True)
lanes = np.arange(64); G = lanes >> 4; PL = lanes & 15
for k in range(8):
    h = np.maximum(pre[k], 0)           # (n, out_k)
    od = h.shape[1]
    worst = 0.0; nbad = 0
    for w in range(4):
        for reg in range(128):
            rows = 16 * (reg >> 2) + 4 * G + (reg & 3)
            ok = rows < od
            ref = np.where(ok, h[16 * w + PL, np.minimum(rows, od - 1)], np.nan)
            got = slabs[k, w, reg]
            d = np.abs(got - ref)[ok]
            if d.size:
                worst = max(worst, d.max()); nbad += int((d > 1e-4).sum())
    print("fwd pass %d (out %d): max |gpu - oracle| = %.3e, entries off by >1e-4: %d" % (k, od, worst, nbad))
    if nbad and k <= 1:
        # show where
        w = 0
        for reg in range(0, 128, 1):
            rows = 16 * (reg >> 2) + 4 * G + (reg & 3)
            ref = h[16 * w + PL, np.minimum(rows, od - 1)]
            d = np.abs(slabs[k, w, reg] - ref)
            if d.max() > 1e-4:
                print("   wave0 reg %3d: bad lanes %s" % (reg, np.where(d > 1e-4)[0][:16]))
                if reg > 24: break
print("sdf diff", np.abs(grad[:, 67] - y).max())

#!/usr/bin/env python3
"""Development aid: effective shader clock of the decoder kernel under full load + kernel-only time."""
import os, sys, time, ctypes as C
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from dsp_slam_amd import fixtures, engine as E, _lib as L
from dsp_slam_amd.deep_sdf.deep_sdf_decoder import fold_weight_norm
layers = fold_weight_norm(fixtures.load_decoder_npz(fixtures.fixture_path("cars")), 9)
eng = E.Engine(layers, [4], 64, device=0)
lib = L.load()
lib.dsp_debug_last_clocks.restype = C.c_int
lib.dsp_debug_last_clocks.argtypes = [C.c_void_p, C.POINTER(C.c_uint64)]
rng = np.random.default_rng(0)
code = (rng.normal(size=64) * 0.2).astype(np.float32)
for n_rounds, bwd in ((24, False), (24, False), (12, True)):
    n = 64 * 256 * n_rounds
    pts = rng.uniform(-1, 1, size=(n, 3)).astype(np.float32)
    if bwd:
        eng.sdf_jacobian(code, pts)
    else:
        eng.decode_sdf(code, pts)
    clk = (C.c_uint64 * 4)()
    L.check(lib.dsp_debug_last_clocks(eng._h, clk), eng._h, "clk")
    cyc, wall = clk[2] - clk[0], clk[3] - clk[1]
    secs = wall / 100e6
    flop = n * (7.34208e6 if bwd else 3.67104e6)
    print("%s %d tiles/CU: WG0 %d shader cycles in %.3f ms -> %.0f MHz effective; kernel %.1f TFLOP/s = %.1f%% of 157.3; "
          "cycles per tile %.0f (ideal MFMA-bound %d)" % ("jac" if bwd else "fwd", n_rounds, cyc, secs * 1e3, cyc / secs / 1e6, flop / secs / 1e12,
          100 * flop / secs / 157.3e12, cyc / n_rounds, (26624 if not bwd else 26624 + 28672) * 32), flush=True)

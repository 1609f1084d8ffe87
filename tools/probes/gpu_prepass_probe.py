#!/usr/bin/env python3
"""Development aid: throughput of the low-precision prepass kernel alone (dsp_decode_sdf_prepass on resident points)."""
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
for dt, name in ((L.PREPASS_F16, "f16"), (L.PREPASS_BF16, "bf16")):
    for n_rounds in (4, 32, 32):
        n = 128 * 256 * n_rounds
        pts = rng.uniform(-0.6, 0.6, size=(n, 3)).astype(np.float32)
        t0 = time.perf_counter()
        out = eng.decode_sdf_prepass(code, pts, dt)
        t1 = time.perf_counter()
        clk = (C.c_uint64 * 4)()
        L.check(lib.dsp_debug_last_clocks(eng._h, clk), eng._h, "clk")
        cyc, wall = clk[2] - clk[0], clk[3] - clk[1]
        secs = wall / 100e6
        flop = n * 3.67104e6
        print("%s %d tiles/CU: WG0 %d shader cycles in %.3f ms -> %.0f MHz; kernel %.0f TFLOP/s = %.1f%% of 2500; cycles per 128-pt tile %.0f "
              "(MFMA-bound %d); host call %.1f ms" % (name, n_rounds, cyc, secs * 1e3, cyc / secs / 1e6, flop / secs / 1e12, 100 * flop / secs / 2.5e15,
              cyc / n_rounds, 228 * 16 * 32, (t1 - t0) * 1e3), flush=True)

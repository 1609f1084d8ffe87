#!/usr/bin/env python3
"""Where the fp32 decoder kernel's matrix pipe idles: the bare forward kernel (dsp_decode_sdf on resident points: mlp_kernel<0>) and the
same launch in library VARIANTS built with one element of the hot loop removed (-DK1_ABL_NOBAR: no per-chunk s_barrier; -DK1_ABL_NOLDS: no
A-operand ds_reads / lgkmcnt waits; -DK1_ABL_NODMA: no LDS-DMA refills; -DK1_ABL_NOEPI: no relu / mask epilogue).  The variants compute
WRONG values; only their time means anything.  One process per variant (DSPGN_LIB selects the library):

    for v in main nobar nolds nodma noepi; do DSPGN_LIB=... python tools/probes/gpu_k1_ablation.py $v; done   (tools/gpu_r05_call2.sh)
"""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from dsp_slam_amd import fixtures, engine as E, _lib as L  # noqa: E402
from dsp_slam_amd.deep_sdf.deep_sdf_decoder import fold_weight_norm  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "main"
layers = fold_weight_norm(fixtures.load_decoder_npz(fixtures.fixture_path("cars")), 9)
eng = E.Engine(layers, [4], 64, device=0)
lib = L.load()
lib.dsp_debug_last_clocks.restype = C.c_int
lib.dsp_debug_last_clocks.argtypes = [C.c_void_p, C.POINTER(C.c_uint64)]
rng = np.random.default_rng(0)
code = (rng.normal(size=64) * 0.2).astype(np.float32)
n_rounds = 24
n = 64 * 256 * n_rounds
pts = rng.uniform(-0.6, 0.6, size=(n, 3)).astype(np.float32)
best = None
for rep in range(4):
    eng.decode_sdf(code, pts)
    clk = (C.c_uint64 * 4)()
    L.check(lib.dsp_debug_last_clocks(eng._h, clk), eng._h, "clk")
    cyc, wall = clk[2] - clk[0], clk[3] - clk[1]
    secs = wall / 100e6
    rec = (secs, cyc)
    if rep and (best is None or secs < best[0]):
        best = rec
secs, cyc = best
flop = n * 3671040.0
mfma_cycles = 416 * 64 * 32          # 416 forward chunks x 64 MFMAs per wave and chunk x 32 cycles
print("| %s | %.3f | %.0f | %.1f | %.4f | %.0f | %.4f |" % (name, secs * 1e3, cyc / secs / 1e6, flop / secs / 1e12, flop / secs / 157.3e12,
                                                          cyc / n_rounds, mfma_cycles / (cyc / n_rounds)), flush=True)
eng.close()

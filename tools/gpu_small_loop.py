#!/usr/bin/env python3
"""Development aid: a loop of single-object runs (for rocprofv3 --kernel-trace --stats): python tools/gpu_small_loop.py M Bg reps"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from dsp_slam_amd import fixtures, synth, engine as E
from dsp_slam_amd.deep_sdf.deep_sdf_decoder import fold_weight_norm
M, Bg, reps = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
eng = E.Engine(fold_weight_norm(fixtures.load_decoder_npz(fixtures.fixture_path("cars")), 9), [4], 64, device=0)
o = synth.make_object(1, n_surface=M, n_background=Bg)
b = eng.batch(E.gn_params(), [o["t_cam_obj_init"]], [o["pts"]], [o["rays"]], [o["depth"]])
b.run()
t0 = time.perf_counter()
for _ in range(reps):
    b.run()
print("M=%d Bg=%d: %.2f ms per run" % (M, Bg, (time.perf_counter() - t0) / reps * 1e3))
import ctypes as C
from dsp_slam_amd import _lib as L
clk = (C.c_uint64 * 8)()
lib = L.load()
lib.dsp_debug_solve_clocks.argtypes = [C.c_void_p, C.POINTER(C.c_uint64)]
if lib.dsp_debug_solve_clocks(eng._h, clk) == 0:
    c = list(clk)
    print("k_solve stages (us): setup %.1f, eliminate %.1f, divide+update+exp %.1f, derive_iter_state %.1f" % (
        (c[1] - c[0]) / 100.0, (c[2] - c[1]) / 100.0, (c[3] - c[2]) / 100.0, (c[4] - c[3]) / 100.0))
    if c[7] > c[1]:
        print("   eliminate = factorisation %.1f + back substitution %.1f" % ((c[7] - c[1]) / 100.0, (c[2] - c[7]) / 100.0))
    print("   shader clock during the elimination: %.0f MHz (%d cycles)" % ((c[6] - c[5]) / ((c[2] - c[1]) / 100.0), c[6] - c[5]))

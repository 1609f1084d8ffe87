#!/usr/bin/env python3
"""GPU check: a process that exits with an Engine and a Batch still alive must exit (no hang), in both orders of destruction.
   python tools/gpu_exit_leak.py atexit     -> leaves both to the atexit handler of dsp_slam_amd.engine
   python tools/gpu_exit_leak.py reverse    -> calls dsp_destroy(handle) first, dsp_batch_destroy(batch) after it (ignored by the library)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from dsp_slam_amd import fixtures, synth, engine as E, _lib as L
from dsp_slam_amd.deep_sdf.deep_sdf_decoder import fold_weight_norm
mode = sys.argv[1]
eng = E.Engine(fold_weight_norm(fixtures.load_decoder_npz(fixtures.fixture_path("cars")), 9), [4], 64, device=0)
o = synth.make_object(1, n_surface=250, n_background=200)
b = eng.batch(E.gn_params(num_iterations=2), [o["t_cam_obj_init"]], [o["pts"]], [o["rays"]], [o["depth"]])
b.run()
r = b.results()
if mode == "reverse":
    lib = L.load()
    h, bh = eng._h, b._h
    lib.dsp_destroy(h)                 # takes the batch with it
    lib.dsp_batch_destroy(bh)          # ... so this is ignored
    import ctypes as C
    eng._h = C.c_void_p(); b._h = C.c_void_p()
print("exit-leak check '%s': results %s, leaving with the engine %s" % (mode, "ok" if r[3][0] == 0 else "status %d" % r[3][0], "destroyed handle-first" if mode == "reverse" else "and the batch still open"))

#!/usr/bin/env python3
"""Development aid: wall time of MeshExtractor-style mesh extraction on the device (grid decode + marching cubes + copy back)."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from dsp_slam_amd import fixtures, engine as E
from dsp_slam_amd.deep_sdf.deep_sdf_decoder import fold_weight_norm
eng = E.Engine(fold_weight_norm(fixtures.load_decoder_npz(fixtures.fixture_path("cars")), 9), [4], 64, device=0)
code = np.zeros(64, np.float32); code[:3] = (0.3, -0.2, 0.1)
for n in (32, 64, 128):
    eng.extract_mesh(code, n)
    ts = []
    for _ in range(5):
        t0 = time.perf_counter(); v, f = eng.extract_mesh(code, n); ts.append(time.perf_counter() - t0)
    g = np.linspace(-1, 1, n, dtype=np.float32)
    pts = np.stack(np.meshgrid(g, g, g, indexing="ij"), -1).reshape(-1, 3)
    td = []
    for _ in range(3):
        t0 = time.perf_counter(); eng.decode_sdf(code, pts); td.append(time.perf_counter() - t0)
    print("extract_mesh %d^3: %.2f ms (%d vertices, %d faces); decode_sdf of the same grid through host buffers alone: %.2f ms" % (
        n, np.median(ts) * 1e3, len(v), len(f), np.median(td) * 1e3), flush=True)

#!/usr/bin/env python3
"""Writes the inputs of examples/multi_gpu_c_abi.cpp: the fixture decoder with weight-norm folded (decoder.bin) and n seeded synthetic
objects (objects.bin), as flat little-endian binaries a C program can read without any dependency.

    python examples/export_example_data.py <out dir> [n objects] [surface points] [background rays]
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from dsp_slam_amd import fixtures, synth  # noqa: E402
from dsp_slam_amd.deep_sdf.deep_sdf_decoder import fold_weight_norm  # noqa: E402


def main():
    out = sys.argv[1]
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 16
    m = int(sys.argv[3]) if len(sys.argv) > 3 else 250
    bg = int(sys.argv[4]) if len(sys.argv) > 4 else 100
    os.makedirs(out, exist_ok=True)
    specs = fixtures.SPECS
    layers = fold_weight_norm(fixtures.load_decoder_npz(fixtures.fixture_path("cars")), len(specs["NetworkSpecs"]["dims"]) + 1)
    with open(os.path.join(out, "decoder.bin"), "wb") as f:
        f.write(np.array([len(layers), specs["CodeLength"], specs["NetworkSpecs"]["latent_in"][0]], np.int32).tobytes())
        for w, b in layers:
            w, b = np.ascontiguousarray(w, np.float32), np.ascontiguousarray(b, np.float32)
            f.write(np.array(w.shape, np.int32).tobytes())
            f.write(w.tobytes())
            f.write(b.tobytes())
    with open(os.path.join(out, "objects.bin"), "wb") as f:
        f.write(np.array([n], np.int32).tobytes())
        for i in range(n):
            o = synth.make_object(7000 + i, n_surface=m, n_background=bg)
            f.write(np.array([o["pts"].shape[0], o["rays"].shape[0], o["depth"].shape[0]], np.int32).tobytes())
            for k in ("t_cam_obj_init", "pts", "rays", "depth"):
                f.write(np.ascontiguousarray(o[k], np.float32).tobytes())
    print("wrote", out)


if __name__ == "__main__":
    main()

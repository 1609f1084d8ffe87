#!/usr/bin/env python3
"""The reference's call surface on the MI355X path, end to end on a synthetic detection (needs one MI355X):

    python examples/reconstruct_synthetic.py

Mirrors what DSP-SLAM's C++ does through pybind11 (src/LocalMapping.cc:38-40, src/LocalMapping_util.cc:109-110,179-196):
get_configs / get_decoder -> Optimizer, MeshExtractor -> estimate_pose_cam_obj, reconstruct_object, extract_mesh_from_code.
"""
import json
import os
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "dsp_slam_amd"))      # the mirror packages under the reference's own names
sys.path.insert(0, ROOT)


def main():
    from dsp_slam_amd import fixtures, synth
    from reconstruct.utils import get_configs, get_decoder, write_mesh_to_ply
    from reconstruct.optimizer import Optimizer, MeshExtractor

    work = tempfile.mkdtemp(prefix="dsp_example_")
    # a decoder directory in the reference's on-disk format (specs.json + ModelParameters/latest.pth), from the test fixture
    decoder_dir = fixtures.materialize_decoder_dir("cars", os.path.join(work, "cars_64"))
    cfg = json.load(open(os.path.join(ROOT, "tests", "golden", "config_kitti_optimizer.json")))
    cfg.update(data_type="KITTI", DeepSDF_DIR=decoder_dir, voxels_dim=64)
    cfg_path = os.path.join(work, "config.json")
    json.dump(cfg, open(cfg_path, "w"))

    configs = get_configs(cfg_path)
    decoder = get_decoder(configs)
    optimizer = Optimizer(decoder, configs)
    mesher = MeshExtractor(decoder, configs.optimizer.code_len, configs.voxels_dim)

    det = synth.make_object(seed=3, n_surface=250, n_background=200)        # a KITTI-sized detection
    # Eigen hands pybind11 column-major arrays; the mirror accepts any strides
    rst = optimizer.reconstruct_object(np.asfortranarray(det["t_cam_obj_init"]), np.asfortranarray(det["pts"]),
                                       np.asfortranarray(det["rays"]), det["depth"])
    print("is_good", rst.is_good, "loss", float(rst.loss))
    err0 = np.linalg.norm(det["t_cam_obj_init"][:3, 3] - det["t_cam_obj_gt"][:3, 3])
    err1 = np.linalg.norm(rst.t_cam_obj[:3, 3] - det["t_cam_obj_gt"][:3, 3])
    print("translation error %.3f m -> %.3f m" % (err0, err1))
    mesh = mesher.extract_mesh_from_code(rst.code)
    ply = os.path.join(work, "object.ply")
    write_mesh_to_ply(mesh.vertices, mesh.faces, ply)
    print("mesh: %d vertices, %d faces -> %s" % (len(mesh.vertices), len(mesh.faces), ply))


if __name__ == "__main__":
    main()

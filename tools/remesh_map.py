#!/usr/bin/env python3
"""Batch re-decode every object of a saved DSP-SLAM map (MapObjects.txt) on the MI355X: one kernel launch decodes the
D^3 SDF grid of ALL objects (saved as <id>_sdf.npy); marching cubes then runs on the device on each decoded grid and the mesh is
written as <id>.ply next to the pose <id>.npy -- the files the reference's extract_map_objects.py:46-63 produces.

    python tools/remesh_map.py --config configs/config_kitti.json --map_dir map/kitti/07 [--voxels_dim 64]
"""
import argparse
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "dsp_slam_amd"))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", required=True)
    ap.add_argument("--map_dir", required=True)
    ap.add_argument("--voxels_dim", type=int, default=64)
    args = ap.parse_args()
    from reconstruct.utils import get_configs, get_decoder, write_mesh_to_ply, convert_sdf_voxels_to_mesh
    from reconstruct.optimizer import MeshExtractor
    from dsp_slam_amd.map_objects import read_map_objects
    cfg = get_configs(args.config)
    objs = read_map_objects(os.path.join(args.map_dir, "MapObjects.txt"))
    ext = MeshExtractor(get_decoder(cfg), cfg.optimizer.code_len, args.voxels_dim)
    t0 = time.time()
    grids = ext.decode_grids([o["code"] for o in objs])
    print("decoded %d grids of %d^3 in %.3f s" % (len(objs), args.voxels_dim, time.time() - t0))
    save_dir = os.path.join(args.map_dir, "objects")
    os.makedirs(save_dir, exist_ok=True)
    for o, g in zip(objs, grids):
        np.save(os.path.join(save_dir, "%d.npy" % o["id"]), o["pose"])
        np.save(os.path.join(save_dir, "%d_sdf.npy" % o["id"]), g)
    t0 = time.time()
    n_ok = 0
    for o, g in zip(objs, grids):
        try:
            vertices, faces = convert_sdf_voxels_to_mesh(g)      # marching cubes on the grid decoded above: no second decode
        except ValueError as e:          # no zero crossing inside the grid
            print("object %d: %s" % (o["id"], e))
            continue
        write_mesh_to_ply(vertices, faces, os.path.join(save_dir, "%d.ply" % o["id"]))
        n_ok += 1
    print("meshed %d objects in %.3f s" % (n_ok, time.time() - t0))


if __name__ == "__main__":
    main()

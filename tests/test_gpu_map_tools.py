"""GPU: the saved-map tools of SURVEY.md 8(f-2), end to end on the reference's on-disk formats.

  * tools/remesh_map.py: MapObjects.txt (System_util.cc:123-146) -> per object <id>.npy (4x4 pose, as extract_map_objects.py:46-63 saves
    it), <id>_sdf.npy (the decoded grid) and <id>.ply -- every mesh equal to oracle/mc_oracle.py on that object's decoded grid, every
    grid equal to a separate single-object decode;
  * tools/reoptimise_map.py: read a saved map + the detections' sidecar files, re-optimise all objects as ONE ragged batch per shard
    (the cfg4 job), write the map back: sharded == unsharded bit for bit, the written file re-reads to the optimised values, objects
    without observations or with a failed optimisation keep what the map held.
"""
import json
import os
import sys

import numpy as np
import pytest

from conftest import ROOT, golden, parity_log
from oracle import mc_oracle as M
from dsp_slam_amd import fixtures, synth, engine as E
from dsp_slam_amd.map_objects import read_map_objects, write_map_objects

pytestmark = pytest.mark.gpu


@pytest.fixture
def mirror():
    pkg = os.path.join(ROOT, "dsp_slam_amd")
    sys.path.insert(0, pkg)
    for m in [k for k in sys.modules if k.split(".")[0] in ("reconstruct", "deep_sdf")]:
        del sys.modules[m]
    yield
    sys.path.remove(pkg)
    for m in [k for k in sys.modules if k.split(".")[0] in ("reconstruct", "deep_sdf")]:
        del sys.modules[m]


def _config(tmp_path):
    cars = fixtures.materialize_decoder_dir("cars", str(tmp_path / "cars_64"))
    cfg = json.load(open(os.path.join(ROOT, "tests", "golden", "config_kitti_optimizer.json")))
    cfg["DeepSDF_DIR"] = cars
    cfg.setdefault("data_type", "KITTI")
    cfg.setdefault("voxels_dim", 32)
    p = str(tmp_path / "cfg.json")
    with open(p, "w") as f:
        json.dump(cfg, f)
    return p


def _run_tool(name, argv):
    import runpy
    old = sys.argv
    sys.argv = [name] + argv
    try:
        runpy.run_path(os.path.join(ROOT, "tools", name), run_name="__main__")
    finally:
        sys.argv = old


def test_remesh_map_end_to_end(tmp_path, mirror, oracle_decoder):
    g = golden("golden_map_objects.npz")
    map_dir = tmp_path / "map"
    map_dir.mkdir()
    with open(map_dir / "MapObjects.txt", "wb") as f:
        f.write(g["text"].tobytes())          # the bytes the reference's own parse loop was pinned on (tools/make_golden_map.py)
    # the recorded codes are random (no surface inside the grid for some): give two objects a real shape so that meshes come out
    objs = read_map_objects(str(map_dir / "MapObjects.txt"))
    for o, c3 in zip(objs[:2], ((0.3, -0.2, 0.1), (-0.4, 0.5, 0.0))):
        o["code"] = np.zeros(64, np.float32)
        o["code"][:3] = c3
    write_map_objects(str(map_dir / "MapObjects.txt"), objs)
    objs = read_map_objects(str(map_dir / "MapObjects.txt"))
    n = 32
    _run_tool("remesh_map.py", ["--config", _config(tmp_path), "--map_dir", str(map_dir), "--voxels_dim", str(n)])
    from reconstruct.utils import read_mesh_from_ply, create_voxel_grid
    eng = E.Engine(oracle_decoder.layers, oracle_decoder.latent_in, oracle_decoder.code_len, device=0)
    n_mesh = 0
    for o in objs:
        d = map_dir / "objects"
        pose = np.load(d / ("%d.npy" % o["id"]))
        assert pose.shape == (4, 4) and np.array_equal(pose, o["pose"])
        grid = np.load(d / ("%d_sdf.npy" % o["id"]))
        assert grid.shape == (n, n, n) and grid.dtype == np.float32
        assert np.array_equal(grid.reshape(-1), eng.decode_sdf(o["code"], create_voxel_grid(n)))      # batched decode == single-object decode
        ply = d / ("%d.ply" % o["id"])
        try:
            ov, of = M.convert_sdf_voxels_to_mesh(grid)
        except ValueError:
            assert not ply.exists()           # no zero crossing inside the grid: the tool reports it and writes no mesh
            continue
        v, f = read_mesh_from_ply(str(ply))
        assert np.array_equal(v, ov) and np.array_equal(f, of)
        n_mesh += 1
    eng.close()
    assert n_mesh >= 2


def _make_map(tmp_path, n_obj, rng):
    """A synthetic saved map: objects observed by cameras at random world poses; saved pose / code = the perturbed initial estimates."""
    map_dir = tmp_path / "map"
    (map_dir / "observations").mkdir(parents=True)
    objs = []
    for i in range(n_obj):
        o = synth.make_object(7000 + i, n_surface=int(rng.integers(120, 400)), n_background=int(rng.integers(30, 120)))
        t_wc = np.eye(4)
        a = rng.uniform(-np.pi, np.pi)
        t_wc[:3, :3] = [[np.cos(a), 0, np.sin(a)], [0, 1, 0], [-np.sin(a), 0, np.cos(a)]]
        t_wc[:3, 3] = rng.uniform(-30, 30, size=3)
        code = np.zeros(64, np.float32)
        code[:3] = 0.5 * o["code_gt"][:3]
        objs.append(dict(id=3 * i + 1, pose=t_wc @ o["t_cam_obj_init"].astype(np.float64), code=code))
        if i != 2:                      # object 2 was never observed again: it must keep what the map holds
            np.savez(map_dir / "observations" / ("%d.npz" % (3 * i + 1)), pts=o["pts"], rays=o["rays"], depth=o["depth"], t_world_cam=t_wc)
    write_map_objects(str(map_dir / "MapObjects.txt"), objs)
    return map_dir


def test_reoptimise_map_sharded_equals_unsharded(tmp_path, mirror, oracle_decoder):
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import reoptimise_map as R
    rng = np.random.default_rng(5)
    map_dir = _make_map(tmp_path, 9, rng)
    objs = read_map_objects(str(map_dir / "MapObjects.txt"))
    obs = R.load_observations(str(map_dir), objs)
    assert sum(o is None for o in obs) == 1
    eng = E.Engine(oracle_decoder.layers, oracle_decoder.latent_in, oracle_decoder.code_len, device=0)
    prm = E.gn_params(num_iterations=4)
    whole, st1 = R.reoptimise([eng], prm, objs, obs, 64)
    # three uneven shards (one of them empty) on the same GPU, each its own batch: the block partition a 3-GPU job would use
    shard, st3 = R.reoptimise([eng], prm, objs, obs, 64, shards=[(0, 3), (3, 3), (3, 8)])
    assert np.array_equal(st1["packed"], st3["packed"])
    for a, b in zip(whole, shard):
        assert np.array_equal(a["pose"], b["pose"]) and np.array_equal(a["code"], b["code"])
    assert st1["n_observed"] == 8 and st1["n_good"] == 8
    assert np.array_equal(whole[2]["pose"], objs[2]["pose"]) and np.array_equal(whole[2]["code"], objs[2]["code"])     # not observed: untouched
    # the optimisation moved the others towards the truth: the warm-started code grew towards code_gt, the loss is small
    moved = [float(np.abs(w["code"] - o["code"]).max()) for w, o in zip(whole, objs)]
    assert sorted(moved)[1] > 1e-3
    eng.close()
    # the command-line tool on the same map: writes the reference's format, which re-reads to float32(values)
    _run_tool("reoptimise_map.py", ["--config", _config(tmp_path), "--map_dir", str(map_dir), "--gpus", "1"])
    back = read_map_objects(str(map_dir / "MapObjects.reopt.txt"))
    assert [o["id"] for o in back] == [o["id"] for o in objs]
    # (the tool ran the config's 10 iterations on a decoder it loaded from the reference's on-disk format: the same here)
    from deep_sdf.workspace import config_decoder
    cfg = json.load(open(_config(tmp_path)))
    dec = config_decoder(cfg["DeepSDF_DIR"]).cuda(0)
    ten, _ = R.reoptimise([dec.engine], E.params_from_configs(cfg), objs, obs, 64)
    dec.engine.close()
    for o, w in zip(back, ten):
        assert np.allclose(o["pose"][:3], np.asarray(w["pose"], np.float32)[:3], rtol=0, atol=1e-6 * max(1.0, float(np.abs(w["pose"]).max())))
        assert np.allclose(o["code"], w["code"], rtol=0, atol=2e-9 + 1e-7 * float(np.abs(w["code"]).max()))
    # --compute f16: the opt-in low-precision mode on the same map (NOT the parity path): every observed object still good, poses close to the fp32 run's
    _run_tool("reoptimise_map.py", ["--config", _config(tmp_path), "--map_dir", str(map_dir), "--gpus", "1", "--compute", "f16", "--out", str(map_dir / "MapObjects.f16.txt")])
    lp = read_map_objects(str(map_dir / "MapObjects.f16.txt"))
    d = [float(np.abs(np.asarray(a["pose"]) - np.asarray(w["pose"])).max() / np.abs(np.asarray(w["pose"])).max()) for a, w in zip(lp, back)]
    assert max(d) < 5e-2, d
    parity_log(kind="reoptimise_map", case="9-object synthetic map, 8 observed", n_good=int(st1["n_good"]), seconds=float(st1["seconds"]),
               sharded_equals_unsharded=True, lp_pose_rel_max=max(d))

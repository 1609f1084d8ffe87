"""The unchanged-C++ drop-in must START (missing 2).

src/System.cc:90-99 does, in this order: `sys.path.append("./")`, `import reconstruct.utils`, `get_configs`, `get_decoder`, and then
`reconstruct.get_sequence(strSequencePath, pyCfg)` -- whose classes live in the reference's reconstruct/kitti_sequence.py / mono_sequence.py
and import `read_calib_file`, `load_velo_scan`, `ForceKeyErrorDict` from `reconstruct.utils`, `get_rays`, `get_time` from
`reconstruct.loss_utils` and `get_detectors` from `reconstruct`.  The mirror package serves the hot-path modules itself and appends the
reference's `reconstruct/` directory to its `__path__` for the rest (dsp_slam_amd/reconstruct/__init__.py).

Every case runs in a fresh interpreter (the parent test process has the mirror imported already), laid out as System.cc leaves it: the
mirror first on sys.path, the process's working directory = the DSP-SLAM source directory, "./" appended.
  * against the REAL reference checkout where there is one (/root/reference in the build container), with cv2 / mmcv / mmdet / mmdet3d
    stubbed in sys.modules (absent from this image; only module-level imports need them on the paths taken);
  * against a stand-in checkout written by the test (a `reconstruct/kitti_sequence.py` with the reference module's import lines), so the
    mechanism is covered on a box without the reference too.
No GPU: get_decoder is left out here; tests/test_embed_harness.py makes the same get_sequence call from C++ after get_decoder.
"""
import json
import os
import subprocess
import sys
import textwrap

import numpy as np
import pytest

from conftest import ROOT

MIRROR = os.path.join(ROOT, "dsp_slam_amd")
REFERENCE = "/root/reference"
have_reference = os.path.isfile(os.path.join(REFERENCE, "reconstruct", "kitti_sequence.py"))

STUBS = textwrap.dedent('''
    import sys, types
    def _stub(name, **attrs):
        m = types.ModuleType(name)
        m.__dict__.update(attrs)
        sys.modules[name] = m
        return m
    _any = lambda *a, **k: None
    _stub("cv2")
    _stub("mmcv", Config=type("Config", (), {}))
    _stub("mmcv.runner", load_checkpoint=_any)
    _stub("mmdet"); _stub("mmdet.models", build_detector=_any); _stub("mmdet.core", get_classes=_any); _stub("mmdet.apis", inference_detector=_any)
    _stub("mmdet3d"); _stub("mmdet3d.models", build_model=_any); _stub("mmdet3d.apis", inference_detector=_any, convert_SyncBN=_any)
''')


def write_kitti_dir(d):
    """The three things KITIISequence.__init__ reads (kitti_sequence.py:219-225,241-256): calib.txt with P2 and Tr, image_2/, velodyne/."""
    os.makedirs(os.path.join(d, "image_2"))
    os.makedirs(os.path.join(d, "velodyne"))
    open(os.path.join(d, "image_2", "000000.png"), "wb").close()
    np.arange(8, dtype=np.float32).tofile(os.path.join(d, "velodyne", "000000.bin"))
    p2 = [718.856, 0, 607.1928, 45.38225, 0, 718.856, 185.2157, -0.1130887, 0, 0, 1, 0.003779761]
    tr = [0.0004276802, -0.9999672, -0.008084491, -0.01198459, -0.007210626, 0.008081198, -0.9999413, -0.05403984, 0.9999738, 0.0004859485, -0.007206933, -0.2921968]
    with open(os.path.join(d, "calib.txt"), "w") as f:
        f.write("P0: 1 0 0 0 0 1 0 0 0 0 1 0\n")
        f.write("P2: " + " ".join("%.9g" % v for v in p2) + "\n")
        f.write("calib_time: 09-Jan-2012 13:57:47\n")          # non-numeric values are skipped (utils.py:68-72)
        f.write("Tr: " + " ".join("%.9g" % v for v in tr) + "\n")
        f.write("\nignored: 1 2 3\n")                           # parsing stops at the first empty line (utils.py:64-65)
    return np.array(p2).reshape(3, 4), np.array(tr).reshape(3, 4)


def write_cfg(path, data_type="KITTI", online=False):
    cfg = {"data_type": data_type, "detect_online": online, "path_label_2d": "labels/2d", "path_label_3d": "labels/3d", "DeepSDF_DIR": "weights/cars_64",
           "voxels_dim": 32, "slam_config_path": "configs/none.yaml",
           "optimizer": {"code_len": 64, "num_depth_samples": 50, "cut_off_threshold": 0.01,
                         "joint_optim": {"k1": 1.0, "k2": 100.0, "k3": 0.25, "k4": 1e7, "b1": 0.2, "b2": 0.025, "num_iterations": 10, "learning_rate": 1.0, "scale_damping": 1.0},
                         "pose_only_optim": {"num_iterations": 5, "learning_rate": 1.0}}}
    with open(path, "w") as f:
        json.dump(cfg, f)


def run_as_system_cc(cwd, body, env=None):
    """A fresh interpreter set up the way System.cc:90-94 leaves it -- PYTHONPATH puts the mirror first, cwd = the DSP-SLAM source directory,
    "./" appended -- runs `body`; its stdout lines `key value` come back as a dict."""
    # (`python -c` puts the working directory FIRST on sys.path; an embedded interpreter -- py::initialize_interpreter, System.cc:90 -- does not
    # have that entry, which is why System.cc:93 appends "./" itself: drop it, then append as C++ does)
    script = STUBS + 'import sys\nsys.path = [p for p in sys.path if p != ""]\nsys.path.append("./")\n' + textwrap.dedent(body)
    e = dict(os.environ, PYTHONPATH=MIRROR)
    e.pop("DSP_REFERENCE_ROOT", None)
    e.update(env or {})
    r = subprocess.run([sys.executable, "-c", script], cwd=cwd, env=e, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-4000:]
    return dict(line.split(" ", 1) for line in r.stdout.splitlines() if " " in line)


BODY_KITTI = '''
    import reconstruct.utils as io_utils                       # System.cc:94
    cfg = io_utils.get_configs(%(cfg)r)                        # System.cc:96
    import reconstruct
    seq = reconstruct.get_sequence(%(data)r, cfg)              # System.cc:98
    import reconstruct.kitti_sequence, reconstruct.optimizer, reconstruct.loss, reconstruct.loss_utils
    print("sequence_class", type(seq).__name__)
    print("sequence_file", reconstruct.kitti_sequence.__file__)
    print("optimizer_file", reconstruct.optimizer.__file__)
    print("loss_file", reconstruct.loss.__file__)
    print("loss_utils_file", reconstruct.loss_utils.__file__)
    print("utils_file", io_utils.__file__)
    print("detectors", repr((seq.detector_2d, seq.detector_3d)))
    print("k_cam", " ".join("%%.9g" %% v for v in seq.K_cam.reshape(-1)))
    print("t_cam_velo", " ".join("%%.9g" %% v for v in seq.T_cam_velo.reshape(-1)))
    print("num_frames", seq.num_frames)
    print("velo", " ".join("%%g" %% v for v in io_utils.load_velo_scan(%(data)r + "/velodyne/000000.bin").reshape(-1)))
'''


def _check_kitti(out, where, p2, tr):
    assert out["sequence_class"] == "KITIISequence"
    assert os.path.realpath(out["sequence_file"]).startswith(os.path.realpath(where))                    # the loader is the (reference) checkout's file
    for k in ("optimizer_file", "loss_file", "loss_utils_file", "utils_file"):                              # the hot path is this repository's
        assert os.path.realpath(out[k]).startswith(os.path.realpath(MIRROR)), (k, out[k])
    assert out["detectors"] == "(None, None)"                                                               # detect_online = false (reconstruct/__init__.py:9-11)
    k = np.array(out["k_cam"].split(), np.float64).reshape(3, 3)
    assert np.array_equal(k.astype(np.float32), p2[:, :3].astype(np.float32))                              # kitti_sequence.py:246-248 through the mirror's read_calib_file
    t0, t2 = np.eye(4), np.eye(4)
    t0[:3] = tr
    t2[0, 3] = p2[0, 3] / p2[0, 0]
    assert np.allclose(np.array(out["t_cam_velo"].split(), np.float64).reshape(4, 4), t2.dot(t0).astype(np.float32), rtol=0, atol=1e-6)
    assert out["num_frames"] == "1" and out["velo"] == "0 1 2 3 4 5 6 7"


@pytest.mark.skipif(not have_reference, reason="no reference checkout on this box")
def test_system_cc_sequence_starts_on_the_real_reference(tmp_path):
    """VERDICT r5's probe: mirror first, reference appended -> get_sequence returns the REFERENCE'S KITIISequence, constructed."""
    data = str(tmp_path / "kitti07")
    os.makedirs(data)
    p2, tr = write_kitti_dir(data)
    write_cfg(str(tmp_path / "cfg.json"))
    out = run_as_system_cc(REFERENCE, BODY_KITTI % dict(cfg=str(tmp_path / "cfg.json"), data=data))
    _check_kitti(out, REFERENCE, p2, tr)


@pytest.mark.skipif(not have_reference, reason="no reference checkout on this box")
def test_every_reference_only_module_resolves_to_the_reference(tmp_path):
    """mono_sequence / detector2d / detector3d import from the reference's files (module-level imports satisfied by the stubs), the mono
    dispatch reaches MonoSequence.__init__ (which stops at the stubbed cv2.FileStorage), online detectors reach the reference's Detector2D."""
    write_cfg(str(tmp_path / "mono.json"), "Freiburg")
    write_cfg(str(tmp_path / "online.json"), "KITTI", online=True)
    out = run_as_system_cc(REFERENCE, '''
        import reconstruct, reconstruct.utils as io_utils
        import reconstruct.mono_sequence, reconstruct.detector2d, reconstruct.detector3d
        for m in (reconstruct.mono_sequence, reconstruct.detector2d, reconstruct.detector3d):
            print(m.__name__.split(".")[-1], m.__file__)
        try:
            reconstruct.get_sequence("/nowhere", io_utils.get_configs(%r))
            print("mono", "constructed")
        except AttributeError as e:                       # module 'cv2' has no attribute 'FileStorage': raised INSIDE MonoSequence.__init__ (mono_sequence.py:122)
            print("mono", "reached_init" if "FileStorage" in str(e) else "other " + str(e))
        try:
            reconstruct.get_detectors(io_utils.get_configs(%r))
            print("online", "constructed")
        except KeyError as e:                             # configs.Detector2D: raised INSIDE Detector2D.__init__ (detector2d.py:40)
            print("online", "reached_init" if "Detector2D" in str(e) else "other " + str(e))
        print("color_table_rows", len(io_utils.color_table) if "addict" in sys.modules or __import__("importlib").util.find_spec("addict") else -1)
    ''' % (str(tmp_path / "mono.json"), str(tmp_path / "online.json")))
    for m in ("mono_sequence", "detector2d", "detector3d"):
        assert os.path.realpath(out[m]).startswith(os.path.realpath(REFERENCE)), out[m]
    assert out["mono"] == "reached_init" and out["online"] == "reached_init"


STANDIN = '''
    """Stand-in for the reference's reconstruct/kitti_sequence.py: the import lines of the real module (kitti_sequence.py:18-24) and a class of
    the same name whose constructor uses what it imported.  Written by tests/test_dropin_delegation.py."""
    import os
    import numpy as np
    from reconstruct.loss_utils import get_rays, get_time
    from reconstruct.utils import ForceKeyErrorDict, read_calib_file, load_velo_scan
    from reconstruct import get_detectors


    class KITIISequence:
        def __init__(self, data_dir, configs):
            self.rgb_dir = os.path.join(data_dir, "image_2")
            calib = read_calib_file(os.path.join(data_dir, "calib.txt"))
            p2 = np.reshape(calib["P2"], (3, 4))
            self.K_cam = p2[0:3, 0:3].astype(np.float32)
            t0, t2 = np.eye(4), np.eye(4)
            t0[:3, :] = np.reshape(calib["Tr"], (3, 4))
            t2[0, 3] = p2[0, 3] / p2[0, 0]
            self.T_cam_velo = t2.dot(t0).astype(np.float32)
            self.num_frames = len(os.listdir(self.rgb_dir))
            self.configs = configs
            self.detector_2d, self.detector_3d = get_detectors(configs)
'''


def write_standin_checkout(d):
    os.makedirs(os.path.join(d, "reconstruct"))
    with open(os.path.join(d, "reconstruct", "kitti_sequence.py"), "w") as f:
        f.write(textwrap.dedent(STANDIN))
    with open(os.path.join(d, "reconstruct", "__init__.py"), "w") as f:       # never imported: the mirror's package is first on sys.path
        f.write("raise ImportError('the checkout\\'s own reconstruct/__init__.py must not be imported when the mirror is first on sys.path')\n")
    with open(os.path.join(d, "reconstruct", "optimizer.py"), "w") as f:      # a same-named module must NOT shadow the mirror's
        f.write("raise ImportError('the checkout\\'s optimizer.py must not be imported: the mirror serves reconstruct.optimizer')\n")


@pytest.mark.parametrize("how", ["cwd", "env"])
def test_system_cc_sequence_starts_on_a_standin_checkout(tmp_path, how):
    """The same call sequence against a stand-in checkout, found through "./" on sys.path (how C++ runs) or through DSP_REFERENCE_ROOT."""
    co = str(tmp_path / "dsp_slam_src")
    write_standin_checkout(co)
    data = str(tmp_path / "kitti07")
    os.makedirs(data)
    p2, tr = write_kitti_dir(data)
    write_cfg(str(tmp_path / "cfg.json"))
    body = BODY_KITTI % dict(cfg=str(tmp_path / "cfg.json"), data=data)
    if how == "cwd":
        out = run_as_system_cc(co, body)
    else:
        other = str(tmp_path / "elsewhere")
        os.makedirs(other)
        out = run_as_system_cc(other, body, env={"DSP_REFERENCE_ROOT": co})
    _check_kitti(out, co, p2, tr)


def test_without_a_checkout_the_error_says_what_to_do(tmp_path):
    write_cfg(str(tmp_path / "cfg.json"))
    out = run_as_system_cc(str(tmp_path), '''
        import reconstruct, reconstruct.utils as io_utils
        cfg = io_utils.get_configs(%r)
        print("offline_detectors", repr(reconstruct.get_detectors(cfg)))          # needs no checkout (reconstruct/__init__.py:9-13)
        try:
            reconstruct.get_sequence("/nowhere", cfg)
        except ImportError as e:
            print("error", str(e))
        try:
            io_utils.no_such_name
        except AttributeError as e:
            print("attr", "AttributeError")
    ''' % str(tmp_path / "cfg.json"))
    assert out["offline_detectors"] == "(None, None)"
    assert "DSP_REFERENCE_ROOT" in out["error"] and out["attr"] == "AttributeError"


def test_read_calib_file_matches_the_reference(tmp_path):
    """The mirror's two loader helpers against the reference's own functions on the same files (where a reference checkout exists)."""
    d = str(tmp_path / "k")
    os.makedirs(d)
    write_kitti_dir(d)
    sys.path.insert(0, MIRROR)
    try:
        for m in [k for k in sys.modules if k.split(".")[0] in ("reconstruct", "deep_sdf")]:
            del sys.modules[m]
        from reconstruct.utils import read_calib_file, load_velo_scan
        mine = read_calib_file(os.path.join(d, "calib.txt"))
        assert sorted(mine) == ["P0", "P2", "Tr"] and mine["P2"].dtype == np.float64 and mine["P2"].shape == (12,)
        scan = load_velo_scan(os.path.join(d, "velodyne", "000000.bin"))
        assert scan.shape == (2, 4) and scan.dtype == np.float32
    finally:
        sys.path.remove(MIRROR)
        for m in [k for k in sys.modules if k.split(".")[0] in ("reconstruct", "deep_sdf")]:
            del sys.modules[m]
    if have_reference:
        from oracle import ref_shim
        ref_shim.install()
        import importlib.util
        spec = importlib.util.spec_from_file_location("_ref_utils_for_test", os.path.join(REFERENCE, "reconstruct", "utils.py"))
        ref = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(ref)
        theirs = ref.read_calib_file(os.path.join(d, "calib.txt"))
        assert sorted(theirs) == sorted(mine) and all(np.array_equal(theirs[k], mine[k]) for k in mine)
        assert np.array_equal(ref.load_velo_scan(os.path.join(d, "velodyne", "000000.bin")), scan)

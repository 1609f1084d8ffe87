"""GPU: the error paths (ADVICE r5 on batch lifetime).

The reference reports numeric failure through the RESULT, never through an exception: `is_good=False, t_cam_obj=None, code=None`
(reconstruct/optimizer.py:131,135-136,143,149-150 -- `math.isnan(sdf_loss)`, `render_rst is None`, `math.isnan(render_loss)`).  Here that is a
per-object status (DSP_OBJ_NAN / DSP_OBJ_FEW_SAMPLES): a poisoned object must fail ALONE -- its neighbours in the batch keep the bits they get
in a batch without it -- and the handle must stay usable after every kind of refused or failed call:

  * NaN / Inf in pts, rays, depth, t_cam_obj                         -> that object's status != 0, neighbours bit-identical;
  * zero rays, zero surface points                                    -> status != 0 (compute_render_loss returns None / mean of nothing is NaN);
  * depth longer than rays, bad sizes                                 -> DSP_E_ARG, and the next call on the handle works;
  * a failed device allocation (injected: dsp_debug_fail_alloc)       -> DSP_E_NOMEM, nothing leaked into the batch, handle usable after dsp_trim;
  * stale batch tokens (the batch's handle destroyed first; destroyed twice; a token that outlived its batch while ANOTHER handle's batch
    reuses the memory)                                                -> refused / ignored, the other handle's batch untouched.
"""
import ctypes as C

import numpy as np
import pytest

from dsp_slam_amd import synth, engine as E, _lib as L

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def eng(oracle_decoder):
    e = E.Engine(oracle_decoder.layers, oracle_decoder.latent_in, oracle_decoder.code_len, device=0)
    yield e
    e.close()


def _objs():
    return [synth.make_object(3100 + i, n_surface=180 + 40 * i, n_background=60 + 10 * i) for i in range(3)]


def _args(objs):
    return ([o["t_cam_obj_init"] for o in objs], [o["pts"] for o in objs], [o["rays"] for o in objs], [o["depth"] for o in objs])


def _alone(eng, prm, o):
    t, c, l, s = eng.reconstruct_batch(prm, *_args([o]))
    return t[0], c[0], l[0], s[0]


def _poison(o, field, value, where="one"):
    o = {k: (np.array(v, copy=True) if isinstance(v, np.ndarray) else v) for k, v in o.items()}
    a = o[field]
    if where == "all":
        a[...] = value
    elif field == "t_cam_obj_init":
        a[1, 2] = value
    else:
        a.reshape(-1)[a.size // 2] = value
    return o


# (field, value, where, what the reference's arithmetic does with it)
#   fail: a NaN reaches one of the two losses (optimizer.py:134-136,148-150) or fewer than 10 samples are inside the sphere (loss.py:73-74), or
#         the normal equations turn NaN and the next linearisation fails -> is_good False;
#   ok:   the poisoned element drops out by itself -- a NaN / Inf RAY fails `norm < 1` (loss.py:68: both compare false) and is simply not
#         in the sphere; an infinite observed depth is clamped to the +-0.30 residual bound (loss.py:135-141) -- the reference optimises on.
CASES = [("pts", np.nan, "one", "fail"), ("pts", np.inf, "one", "fail"), ("pts", np.nan, "all", "fail"),
         ("rays", np.nan, "one", "ok"), ("rays", np.inf, "one", "ok"), ("rays", np.nan, "all", "fail"),
         ("depth", np.nan, "all", "fail"), ("depth", np.inf, "one", "ok"), ("depth", -np.inf, "one", "ok"),
         ("t_cam_obj_init", np.nan, "one", "fail"), ("t_cam_obj_init", np.inf, "one", "fail")]


@pytest.mark.parametrize("field,value,where,expect", CASES)
def test_a_poisoned_object_fails_alone(eng, field, value, where, expect):
    prm = E.gn_params(num_iterations=3)
    objs = _objs()
    clean = [_alone(eng, prm, o) for o in objs]
    assert all(c[3] == 0 for c in clean)
    bad = list(objs)
    bad[1] = _poison(objs[1], field, value, where)
    t, c, l, s = eng.reconstruct_batch(prm, *_args(bad))
    if expect == "ok":
        assert s[1] == 0 and np.isfinite(t[1]).all() and np.isfinite(c[1]).all() and np.isfinite(l[1]), (field, value, where, s[1])
    else:
        assert s[1] in (L.OBJ_NAN, L.OBJ_FEW_SAMPLES), (field, value, where, s)
        if field == "rays":
            assert s[1] == L.OBJ_FEW_SAMPLES          # no sample inside the sphere: compute_render_loss returns None
        if field == "pts" and value != value:
            assert s[1] == L.OBJ_NAN                  # math.isnan(sdf_loss)
    for i in (0, 2):        # the neighbours: the bits of a run without the poisoned object
        assert s[i] == 0
        assert np.array_equal(t[i], clean[i][0]) and np.array_equal(c[i], clean[i][1]) and l[i] == clean[i][2], (field, value, where, i)


def test_failure_through_the_mirror_is_the_references_result_dict(tmp_path):
    """The reference's failure convention at the Python boundary (optimizer.py:131,136,143,150): is_good False, t_cam_obj None, code None."""
    import json
    import os
    import sys
    from conftest import ROOT
    from dsp_slam_amd import fixtures
    pkg = os.path.join(ROOT, "dsp_slam_amd")
    sys.path.insert(0, pkg)
    try:
        for m in [k for k in sys.modules if k.split(".")[0] in ("reconstruct", "deep_sdf")]:
            del sys.modules[m]
        from reconstruct.utils import get_configs, get_decoder
        from reconstruct.optimizer import Optimizer
        cfg_d = json.load(open(os.path.join(ROOT, "tests", "golden", "config_kitti_optimizer.json")))
        cfg_d.update(data_type="KITTI", DeepSDF_DIR=fixtures.materialize_decoder_dir("cars", str(tmp_path / "cars_64")), voxels_dim=16)
        with open(tmp_path / "cfg.json", "w") as f:
            json.dump(cfg_d, f)
        cfg = get_configs(str(tmp_path / "cfg.json"))
        opt = Optimizer(get_decoder(cfg), cfg)
        opt.verbose = False
        o = _poison(_objs()[0], "pts", np.nan, "one")
        rst = opt.reconstruct_object(o["t_cam_obj_init"], o["pts"], o["rays"], o["depth"])
        assert rst.is_good is False and rst.t_cam_obj is None and rst.code is None and isinstance(float(rst.loss), float)
        ok = _objs()[0]
        rst = opt.reconstruct_object(ok["t_cam_obj_init"], ok["pts"], ok["rays"], ok["depth"])       # the same Optimizer still works
        assert rst.is_good is True and np.isfinite(rst.t_cam_obj).all()
    finally:
        sys.path.remove(pkg)
        for m in [k for k in sys.modules if k.split(".")[0] in ("reconstruct", "deep_sdf")]:
            del sys.modules[m]


def test_empty_inputs(eng):
    prm = E.gn_params(num_iterations=2)
    o = _objs()[0]
    good = _alone(eng, prm, o)
    # zero rays: compute_render_loss returns None (< 10 in-sphere samples, loss.py:73-74)
    t, c, l, s = eng.reconstruct_batch(prm, [o["t_cam_obj_init"]] * 2, [o["pts"], o["pts"]], [np.zeros((0, 3), np.float32), o["rays"]],
                                       [np.zeros(0, np.float32), o["depth"]])
    assert s[0] == L.OBJ_FEW_SAMPLES and s[1] == 0 and np.array_equal(t[1], good[0])
    # zero surface points: the mean of no residuals is NaN (optimizer.py:134-136)
    t, c, l, s = eng.reconstruct_batch(prm, [o["t_cam_obj_init"]] * 2, [np.zeros((0, 3), np.float32), o["pts"]], [o["rays"], o["rays"]],
                                       [o["depth"], o["depth"]])
    assert s[0] == L.OBJ_NAN and s[1] == 0 and np.array_equal(t[1], good[0])
    # pose-only with no points at all
    out = eng.estimate_pose_batch(prm, [np.eye(4, dtype=np.float32)], [1.0], [np.zeros((0, 3), np.float32)], [np.zeros(64, np.float32)])
    assert out.shape == (1, 4, 4)


def test_refused_arguments_leave_the_handle_usable(eng):
    prm = E.gn_params(num_iterations=2)
    o = _objs()[0]
    good = _alone(eng, prm, o)
    lib = L.load()
    # depth longer than rays: the foreground rows of `rays` pair with `depth` (optimizer.py:108-112) -- more depths than rays cannot be paired
    with pytest.raises(L.DspError, match="more depths than rays"):
        eng.reconstruct_batch(prm, [o["t_cam_obj_init"]], [o["pts"]], [o["rays"][:10]], [o["depth"][:20]])
    assert np.array_equal(_alone(eng, prm, o)[0], good[0])
    # num_depth_samples out of range
    with pytest.raises(L.DspError):
        eng.reconstruct_batch(E.gn_params(num_depth_samples=65), *_args([o]))
    with pytest.raises(L.DspError):
        eng.reconstruct_batch(E.gn_params(num_depth_samples=1), *_args([o]))
    # NULL arguments at the C ABI
    assert lib.dsp_reconstruct_batch(eng._h, C.byref(prm), 1, None, None, None, None, None, None, None, None, None, None, None, None) == -1
    assert lib.dsp_decode_sdf(eng._h, None, None, 4, None) == -1
    assert lib.dsp_batch_run(None) == -1 and lib.dsp_batch_set_debug(None, 1, 1) == -1
    # an unknown debug key, an out-of-range value
    b = eng.batch(prm, *_args([o]))
    assert lib.dsp_batch_set_debug(b._h, 999, 1) == -1 and lib.dsp_batch_set_debug(b._h, L.DBG_MASK_REUSE, 7) == -1
    assert lib.dsp_batch_set_iterations(b._h, 0) == -1 and lib.dsp_batch_set_prepass(b._h, 3, -1.0) == -1
    # results before the first run: DSP_E_STATE, then the batch still runs
    with pytest.raises(L.DspError, match="not been run"):
        b.results()
    b.run()
    assert np.array_equal(b.results()[0][0], good[0])
    b.close()
    assert np.array_equal(_alone(eng, prm, o)[0], good[0])


def test_a_failed_device_allocation_is_nomem_and_nothing_sticks(eng):
    prm = E.gn_params(num_iterations=2)
    o = _objs()[0]
    good = _alone(eng, prm, o)
    lib = L.load()
    eng.trim()                           # an empty cache: the next batch has to allocate every array afresh
    for nth in (0, 3, 17):               # the first array of the batch, one in the middle, a late one
        eng.fail_alloc(nth)
        po = np.array([0, o["pts"].shape[0]], np.int64)
        ro = np.array([0, o["rays"].shape[0]], np.int64)
        do = np.array([0, o["depth"].shape[0]], np.int64)
        tok = C.c_void_p()
        rc = lib.dsp_batch_create(eng._h, C.byref(prm), 1, L.ptr(po, L.c_i64p), L.ptr(L.f32(o["pts"])), L.ptr(ro, L.c_i64p), L.ptr(L.f32(o["rays"])),
                                  L.ptr(do, L.c_i64p), L.ptr(L.f32(o["depth"])), L.ptr(L.f32(o["t_cam_obj_init"])), None, C.byref(tok))
        assert rc == -3, (nth, rc)                                           # DSP_E_NOMEM
        assert not tok.value and b"out of device memory" in lib.dsp_last_error(eng._h)
        eng.fail_alloc(-1)
        eng.trim()
        assert np.array_equal(_alone(eng, prm, o)[0], good[0]), nth          # the handle works, same bits
        eng.trim()
    # the same through the one-shot entry point
    eng.fail_alloc(5)
    with pytest.raises(L.DspError, match="out of device memory"):
        eng.reconstruct_batch(prm, *_args([o]))
    eng.fail_alloc(-1)
    assert np.array_equal(_alone(eng, prm, o)[0], good[0])


def test_stale_batch_tokens_are_refused(oracle_decoder):
    """dsp_destroy of a handle takes its batches; a token that outlives its batch must never reach ANOTHER handle's batch, even when
    the allocator hands that batch the same address.  Tokens are generation-tagged (include/dsp_gn.h: dsp_destroy)."""
    lib = L.load()
    prm = E.gn_params(num_iterations=2)
    o = _objs()[0]
    e1 = E.Engine(oracle_decoder.layers, oracle_decoder.latent_in, oracle_decoder.code_len, device=0)
    e2 = E.Engine(oracle_decoder.layers, oracle_decoder.latent_in, oracle_decoder.code_len, device=0)
    b1 = e1.batch(prm, *_args([o]))
    b1.run()
    stale = C.c_void_p(b1._h.value)
    assert stale.value & 1, "a batch token is not an object address"
    e1._batches.clear()                  # the engine forgets the batch: dsp_destroy itself has to take it
    e1.close()                           # dsp_destroy(h1)
    # ... many new batches on the OTHER handle: the allocator reuses the freed batch's memory for some of them
    others = [e2.batch(prm, *_args([o])) for _ in range(8)]
    for b in others:
        b.run()
    want = [b.results() for b in others]
    assert lib.dsp_batch_run(stale) == -1 and lib.dsp_batch_set_iterations(stale, 3) == -1
    st = L.Stats()
    assert lib.dsp_batch_stats(stale, C.byref(st)) == -1
    lib.dsp_batch_destroy(stale)         # ignored
    lib.dsp_batch_destroy(stale)         # ... twice
    for b, w in zip(others, want):       # every batch of the other handle is alive and unchanged
        b.run()
        for x, y in zip(b.results(), w):
            assert np.array_equal(x, y)
    # destroying a live batch twice: the second call finds a retired token
    tok = C.c_void_p(others[0]._h.value)
    others[0].close()
    lib.dsp_batch_destroy(tok)
    assert lib.dsp_batch_run(tok) == -1
    nb = e2.batch(prm, *_args([o]))      # the slot is reused with a new generation: the old token still resolves to nothing
    assert nb._h.value != tok.value and lib.dsp_batch_run(tok) == -1
    nb.run()
    b1._h = C.c_void_p()                 # (nothing left for the Python finaliser to do)
    e2.close()


def test_concurrent_destroy_from_a_finaliser_thread(oracle_decoder):
    """dsp_destroy from one thread while another is inside dsp_batch_run on one of the handle's batches: the destroy waits for the call in
    flight (pinned handle), later calls with the token are refused -- no dead mutex, no hang."""
    import threading
    lib = L.load()
    prm = E.gn_params(num_iterations=10)
    objs = [synth.make_object(3300 + i, n_surface=600, n_background=200) for i in range(8)]
    e = E.Engine(oracle_decoder.layers, oracle_decoder.latent_in, oracle_decoder.code_len, device=0)
    b = e.batch(prm, *_args(objs))
    b.run()
    tok, h = C.c_void_p(b._h.value), C.c_void_p(e._h.value)
    rcs = []

    def runner():
        for _ in range(6):
            rcs.append(lib.dsp_batch_run(tok))

    th = threading.Thread(target=runner)
    th.start()
    while not rcs:                       # at least one run has completed: the next is in flight or about to be
        pass
    e._batches.clear()
    e._h = C.c_void_p()                  # (Python's own close must not run a second destroy)
    b._h = C.c_void_p()
    lib.dsp_destroy(h)
    th.join(timeout=60)
    assert not th.is_alive(), "dsp_batch_run did not return after its handle was destroyed"
    assert rcs[0] == 0 and all(r in (0, -1) for r in rcs) and rcs[-1] == -1, rcs


def test_concurrent_batch_destroy_waits_for_the_run_in_flight(oracle_decoder):
    """dsp_batch_destroy from one thread while another is inside dsp_batch_run on THAT batch: the destroy retires the token, waits for the call
    in flight (pinned batch), frees; the runner's later calls see a stale token.  The handle stays usable."""
    import threading
    lib = L.load()
    prm = E.gn_params(num_iterations=10)
    objs = [synth.make_object(3400 + i, n_surface=600, n_background=200) for i in range(8)]
    e = E.Engine(oracle_decoder.layers, oracle_decoder.latent_in, oracle_decoder.code_len, device=0)
    b = e.batch(prm, *_args(objs))
    b.run()
    want = b.results()
    tok = C.c_void_p(b._h.value)
    rcs = []

    def runner():
        for _ in range(6):
            rcs.append(lib.dsp_batch_run(tok))

    th = threading.Thread(target=runner)
    th.start()
    while not rcs:
        pass
    b._h = C.c_void_p()                  # (Python's own close must not destroy it a second time)
    lib.dsp_batch_destroy(tok)
    th.join(timeout=60)
    assert not th.is_alive()
    assert rcs[0] == 0 and all(r in (0, -1) for r in rcs) and rcs[-1] == -1, rcs
    b2 = e.batch(prm, *_args(objs))      # the handle is intact: same inputs, same bits
    b2.run()
    got = b2.results()
    for x, y in zip(want, got):
        assert np.array_equal(np.asarray(x), np.asarray(y), equal_nan=True)
    e.close()

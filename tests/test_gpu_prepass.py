"""GPU: the low-precision prepass kernel (mlp_lp_kernel.hip, through dsp_decode_sdf_prepass) against the direct statement
of its arithmetic and against the fp32 oracle.

The prepass is never a result -- it only classifies samples whose occupancy is exactly 0 or 1 -- so what matters is
(a) it computes what it says (same rounding points as lp_emulator.reference_forward: only accumulation order differs) and
(b) its distance to the fp32 decoder stays inside the margin the optimiser uses (DSP default delta: 4x the measured max)."""
import numpy as np
import pytest

from oracle import dsp_oracle as O
from dsp_slam_amd import engine as E, _lib as L
import lp_emulator as LE

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def eng(oracle_decoder):
    e = E.Engine(oracle_decoder.layers, oracle_decoder.latent_in, oracle_decoder.code_len, device=0)
    yield e
    e.close()


@pytest.mark.parametrize("dtype,tol_lp,tol_fp32", [(L.PREPASS_F16, 1.5e-4, 4e-4), (L.PREPASS_BF16, 1e-3, 4e-3)])
@pytest.mark.parametrize("n", [1, 31, 32, 33, 127, 128, 129, 1000, 40000])
def test_prepass_decode(eng, oracle_decoder, dtype, tol_lp, tol_fp32, n):
    rng = np.random.default_rng(7 * n + dtype)
    code = (rng.normal(size=64) * 0.2).astype(np.float32)
    pts = rng.uniform(-1.0, 1.0, size=(n, 3)).astype(np.float32)
    got = eng.decode_sdf_prepass(code, pts, dtype)
    ref32 = O.decode_sdf(oracle_decoder, code, pts)
    want = LE.reference_forward(oracle_decoder, code, pts, dtype)
    assert got.shape == (n,)
    assert np.isfinite(got).all()
    assert np.abs(got - want).max() < tol_lp, np.abs(got - want).max()
    assert np.abs(got - ref32).max() < tol_fp32, np.abs(got - ref32).max()


def test_prepass_matches_fp32_kernel_closely(eng):
    """device fp32 kernel vs device prepass on a large sample: the number the default margin is derived from."""
    rng = np.random.default_rng(3)
    worst = {}
    for trial in range(4):
        code = (rng.normal(size=64) * (0.1 + 0.1 * trial)).astype(np.float32)
        u = rng.normal(size=(60000, 3))
        pts = (u / np.linalg.norm(u, axis=1, keepdims=True) * rng.uniform(0, 1, size=(60000, 1)) ** (1 / 3)).astype(np.float32)
        ref = eng.decode_sdf(code, pts)
        for name, dt in (("f16", L.PREPASS_F16), ("bf16", L.PREPASS_BF16)):
            worst[name] = max(worst.get(name, 0.0), float(np.abs(eng.decode_sdf_prepass(code, pts, dt) - ref).max()))
    print("prepass max |sdf_lp - sdf_fp32| over 240k unit-ball points:", worst)
    assert worst["f16"] < 4e-4 and worst["bf16"] < 4e-3


# ---------------------------------------------------------------------------------------------------
# the prepass inside the optimiser: bit-identical to prepass off
# ---------------------------------------------------------------------------------------------------
import json

from conftest import golden, parity_log
from dsp_slam_amd import synth

TRACE_KEYS = ("H", "b", "dx", "V", "K", "set_sums", "t_obj_cam", "code")


def _run_traced(eng, prm, args, mode, delta=-1.0, passes=0, reuse=-1, audit=True, iters=None, guard=True):
    b = eng.batch(prm, *args, trace=True)
    b.set_prepass(mode, delta)
    b.set_ray_passes(passes)
    b.set_mask_reuse(reuse)
    b.set_prepass_guard(guard)
    if mode:
        b.set_prepass_audit(audit)
    b.run()
    res = b.results()
    tr = [b.trace(e) for e in range(iters if iters is not None else prm.num_iterations)]
    st = b.stats()
    b.close()
    return res, tr, st


def _assert_identical(run, ref, what):
    for a, c in zip(run[0], ref[0]):
        assert np.array_equal(a, c), what
    for e, (ta, tc) in enumerate(zip(run[1], ref[1])):
        for k in TRACE_KEYS:
            assert np.array_equal(ta[k], tc[k]), "%s: iteration %d, %s" % (what, e, k)


def test_prepass_is_exact_every_mode(eng):
    """Every prepass setting (dtype x pass count x mask reuse) gives every bit of every iteration that prepass-off gives, the
    audit (all samples also decoded in fp32) finds no misclassified sample and an error well inside the margin, and the fp32
    kernel really sees far fewer samples."""
    prm = E.gn_params(num_iterations=4)
    objs = synth.make_batch(3, first_seed=950, n_surface=400, n_background=120)
    args = ([o["t_cam_obj_init"] for o in objs], [o["pts"] for o in objs], [o["rays"] for o in objs], [o["depth"] for o in objs])
    ref = _run_traced(eng, prm, args, L.PREPASS_OFF, passes=1)
    assert ref[2]["n_prepass_points"] == 0 and ref[2]["prepass_mode"] == 0
    for mode, name in ((L.PREPASS_F16, "f16"), (L.PREPASS_BF16, "bf16")):
        for passes in (0, 1, 3, 10):
            for reuse in (0, 1):
                run = _run_traced(eng, prm, args, mode, passes=passes, reuse=reuse)
                what = "%s passes=%d reuse=%d" % (name, passes, reuse)
                _assert_identical(run, ref, what)
                st = run[2]
                assert st["prepass_mode"] == mode and st["n_mlp_prepass_launches"] > 0
                assert st["prepass_misclassified"] == 0 and st["prepass_audited"] > 0
                assert 3.0 * st["prepass_max_err"] <= st["prepass_delta"], (what, st["prepass_max_err"], st["prepass_delta"])
                assert st["n_insphere_points"] == ref[2]["n_insphere_points"]
                assert st["n_fwd_points"] < 0.5 * ref[2]["n_fwd_points"], what      # the fp32 kernel's share
                if passes == 1:
                    assert st["n_prepass_points"] == st["n_insphere_points"]
    # automatic mode is the f16 prepass
    auto = _run_traced(eng, prm, args, -1)
    assert auto[2]["prepass_mode"] == L.PREPASS_F16
    _assert_identical(auto, ref, "auto")


def test_prepass_margin_too_small_is_caught_by_the_audit(eng):
    """delta = 0 classifies on the raw low-precision value: the audit must then report misclassified samples on a batch of
    this size (otherwise it could never catch a margin that is too small)."""
    prm = E.gn_params(num_iterations=3)
    objs = synth.make_batch(4, first_seed=970, n_surface=1000, n_background=250)
    args = ([o["t_cam_obj_init"] for o in objs], [o["pts"] for o in objs], [o["rays"] for o in objs], [o["depth"] for o in objs])
    st = _run_traced(eng, prm, args, L.PREPASS_BF16, delta=0.0, guard=False)[2]     # (with the guard on, the run would be repeated with the prepass off)
    assert st["prepass_misclassified"] > 0


@pytest.mark.parametrize("name", ["golden_recon_small.npz", "golden_recon_cfg1.npz", "golden_recon_redwood.npz", "golden_recon_freiburg.npz",
                                  "golden_recon_cfg2.npz"])
def test_prepass_is_exact_on_reference_goldens(eng, name):
    g = golden(name)
    cfg = json.loads(str(g["cfg_json"]))
    prm = E.params_from_configs(cfg)
    args = ([g["in_t_cam_obj_init"]], [g["in_pts"]], [g["in_rays"]], [g["in_depth"]])
    ref = _run_traced(eng, prm, args, L.PREPASS_OFF)
    for mode in (L.PREPASS_F16, L.PREPASS_BF16):
        run = _run_traced(eng, prm, args, mode)
        _assert_identical(run, ref, "%s mode %d" % (name, mode))
        st = run[2]
        parity_log(kind="prepass", case=name, dtype=["off", "f16", "bf16"][mode], audited=st["prepass_audited"], max_err=st["prepass_max_err"],
                   delta=st["prepass_delta"], misclassified=int(st["prepass_misclassified"]),
                   fwd_over_insphere=st["n_fwd_points"] / st["n_insphere_points"], identical=True)
        assert run[2]["prepass_misclassified"] == 0
        assert 3.0 * run[2]["prepass_max_err"] <= run[2]["prepass_delta"]


def test_prepass_is_exact_on_64_cfg2_objects(eng):
    """The bench workload (BASELINE configs[2]: 64 x cfg2, 10 iterations): prepass on == prepass off, every bit."""
    prm = E.gn_params()
    objs = synth.make_batch(64, first_seed=300, n_surface=2000, n_background=500)
    args = ([o["t_cam_obj_init"] for o in objs], [o["pts"] for o in objs], [o["rays"] for o in objs], [o["depth"] for o in objs])
    ref = _run_traced(eng, prm, args, L.PREPASS_OFF, audit=False)
    run = _run_traced(eng, prm, args, L.PREPASS_F16, audit=True)
    _assert_identical(run, ref, "64 x cfg2")
    st = run[2]
    assert (run[0][3] == 0).all()
    parity_log(kind="prepass", case="64 x cfg2 (bench workload), 10 iterations", dtype="f16", audited=st["prepass_audited"],
               max_err=st["prepass_max_err"], delta=st["prepass_delta"], misclassified=int(st["prepass_misclassified"]),
               fwd_over_insphere=st["n_fwd_points"] / st["n_insphere_points"], identical=True)
    assert st["prepass_misclassified"] == 0 and 3.0 * st["prepass_max_err"] <= st["prepass_delta"]
    assert st["prepass_guard_trips"] == 0 and st["prepass_guard_rerun"] == 0
    print("64 x cfg2: fp32 forward points %.3g -> %.3g (%.1f %% of in-sphere), prepass points %.3g, max |sdf_lp - sdf_fp32| %.3g, delta %.3g" % (
        ref[2]["n_fwd_points"], st["n_fwd_points"], 100 * st["n_fwd_points"] / st["n_insphere_points"], st["n_prepass_points"],
        st["prepass_max_err"], st["prepass_delta"]))


def test_margin_is_calibrated_per_decoder(eng, oracle_decoder):
    """dsp_create measures the prepass error of the decoder it was given and derives the margin from it: the fixture gets the
    floor with f16 (its error is 5x below it) and 5x its measured error with bf16; a decoder whose hidden activations are 30x larger gets a proportionally wider band
    instead of misclassified samples -- and still gives prepass-on == prepass-off."""
    for dt, floor in ((L.PREPASS_F16, 5e-4), (L.PREPASS_BF16, 3e-3)):
        err, delta = eng.prepass_calibration(dt)          # the zero-code entry of the table
        assert 0 < err < 0.1 and abs(delta - max(floor, 6 * err)) < 1e-7
    # scale the last hidden layer's output by 30 and the final layer's weights by 1/30: same function, 30x the activations of layer 7
    layers = [(w.copy(), b.copy()) for w, b in oracle_decoder.layers]
    w7, b7 = layers[7]
    layers[7] = (w7 * 30.0, b7 * 30.0)
    layers[8] = (layers[8][0] / 30.0, layers[8][1])
    big = E.Engine(layers, oracle_decoder.latent_in, oracle_decoder.code_len, device=0)
    e1, d1 = big.prepass_calibration(L.PREPASS_F16)
    assert d1 >= 6 * e1 * 0.999
    prm = E.gn_params(num_iterations=3)
    objs = synth.make_batch(2, first_seed=930, n_surface=300, n_background=80)
    args = ([o["t_cam_obj_init"] for o in objs], [o["pts"] for o in objs], [o["rays"] for o in objs], [o["depth"] for o in objs])
    ref = _run_traced(big, prm, args, L.PREPASS_OFF)
    run = _run_traced(big, prm, args, L.PREPASS_F16)
    _assert_identical(run, ref, "rescaled decoder")
    assert run[2]["prepass_misclassified"] == 0 and abs(run[2]["prepass_delta"] - d1) < 1e-9
    big.close()


# ---------------------------------------------------------------------------------------------------
# safe WITHOUT the audit: code-aware margins + the always-on guard
# ---------------------------------------------------------------------------------------------------
def _args(objs, codes=None):
    a = [[o["t_cam_obj_init"] for o in objs], [o["pts"] for o in objs], [o["rays"] for o in objs], [o["depth"] for o in objs]]
    if codes is not None:
        a.append(codes)
    return tuple(a)


def test_margin_table_follows_the_code_magnitude(eng):
    """dsp_create measures the prepass error with codes drawn at |z|_inf = 0, 0.15, 0.5, 1, 2: the margins are monotone in the
    magnitude, never below the floor, and at least 6x the largest error measured up to that magnitude."""
    for dt, floor in ((L.PREPASS_F16, 5e-4), (L.PREPASS_BF16, 3e-3)):
        t = eng.prepass_calibration_table(dt)
        assert list(t["mags"]) == [0.0, 0.15000000596046448, 0.5, 1.0, 2.0]
        assert np.all(np.diff(t["delta"]) >= 0) and t["delta"][0] >= floor and t["delta"][-1] <= 0.5
        assert np.all(t["delta"] >= np.minimum(0.5, 6 * np.maximum.accumulate(t["max_err"])) - 1e-9)
        assert np.all(t["max_err"] > 0) and t["guard_err"] == 0.0
        # (on the fixture decoders the error hardly depends on the code -- their code columns carry three shape parameters; a decoder
        # whose activations do grow with the code gets the wider margins this table then holds, test_chairs32_and_rescaled_... below)
        print("dtype %d: max_err %s delta %s" % (dt, t["max_err"], t["delta"]))


@pytest.mark.parametrize("mag", [0.5, 1.0, 2.0])
def test_warm_start_codes_are_exact_without_the_audit(eng, mag):
    """The C++ caller warm-starts with arbitrary codes (LocalMapping_util.cc:391-392).  With codes whose entries reach +-mag the prepass
    error is far above the zero-code margin; the per-object margin follows the code, so prepass-on still reproduces prepass-off bit for
    bit -- checked WITHOUT the audit switch (a production run) -- and the guard, which compares every re-decoded sample, stays quiet."""
    prm = E.gn_params(num_iterations=3)
    objs = synth.make_batch(3, first_seed=960, n_surface=300, n_background=80)
    rng = np.random.default_rng(int(mag * 10))
    codes = [(rng.uniform(-mag, mag, size=64)).astype(np.float32) for _ in objs]
    codes[1][:] = 0.0
    codes[1][:3] = mag * np.array([1.0, -1.0, 0.5], np.float32)       # a code the decoder was fitted on, at this magnitude
    ref = _run_traced(eng, prm, _args(objs, codes), L.PREPASS_OFF)
    for mode in (L.PREPASS_F16, L.PREPASS_BF16):
        run = _run_traced(eng, prm, _args(objs, codes), mode, audit=False)
        _assert_identical(run, ref, "warm start |z|=%g mode %d" % (mag, mode))
        st = run[2]
        assert st["prepass_guard_trips"] == 0 and st["prepass_guard_rerun"] == 0, st
        assert st["prepass_audited"] == 0                       # the audit really was off
        assert st["prepass_guard_max_err"] > 0                  # ... and the guard really compared samples
        parity_log(kind="prepass_guard", case="warm start |z|_inf = %g" % mag, dtype=["off", "f16", "bf16"][mode],
                   guard_max_err=st["prepass_guard_max_err"], delta_zero_code=st["prepass_delta"], trips=int(st["prepass_guard_trips"]), identical=True)


def test_reference_goldens_are_exact_without_the_audit(eng):
    for name in ("golden_recon_small.npz", "golden_recon_redwood.npz", "golden_recon_cfg2.npz"):
        g = golden(name)
        prm = E.params_from_configs(json.loads(str(g["cfg_json"])))
        args = [[g["in_t_cam_obj_init"]], [g["in_pts"]], [g["in_rays"]], [g["in_depth"]]]
        if "in_code" in g.files:
            args.append([g["in_code"]])
        ref = _run_traced(eng, prm, tuple(args), L.PREPASS_OFF)
        run = _run_traced(eng, prm, tuple(args), L.PREPASS_F16, audit=False)
        _assert_identical(run, ref, name)
        assert run[2]["prepass_guard_trips"] == 0 and run[2]["prepass_guard_rerun"] == 0
        # the guard's cost: the stratified sample of the classified samples that the fp32 kernel re-decodes
        off = _run_traced(eng, prm, tuple(args), L.PREPASS_F16, audit=False, guard=False)
        _assert_identical(off, ref, name + " (guard off)")
        # (latency-sized objects send their band samples straight into the jacobian launch: count both kinds of fp32 points)
        pts = lambda st: st["n_fwd_points"] + st["n_jac_points"] + st["n_render_rows"]      # noqa: E731
        extra = pts(run[2]) / pts(off[2]) - 1.0
        assert 0.0 < extra < 0.25, extra
        parity_log(kind="prepass_guard", case=name, dtype="f16", guard_max_err=run[2]["prepass_guard_max_err"], delta_zero_code=run[2]["prepass_delta"],
                   trips=0, identical=True, extra_fp32_points=extra)


def test_chairs32_and_rescaled_decoders_are_exact_without_the_audit(chairs32_decoder, oracle_decoder):
    prm = E.gn_params(k1=10.0, k3=2.5, k4=0.0, b2=0.02, s_damp=100.0, num_iterations=4)
    objs = synth.make_batch(2, first_seed=940, n_surface=300, n_background=80, code_len=32, half=synth.CHAIR_HALF)
    ch = E.Engine(chairs32_decoder.layers, chairs32_decoder.latent_in, chairs32_decoder.code_len, device=0)
    ref = _run_traced(ch, prm, _args(objs), L.PREPASS_OFF)
    run = _run_traced(ch, prm, _args(objs), L.PREPASS_F16, audit=False)
    _assert_identical(run, ref, "chairs32")
    assert run[2]["prepass_guard_trips"] == 0
    ch.close()
    layers = [(w.copy(), b.copy()) for w, b in oracle_decoder.layers]
    layers[7] = (layers[7][0] * 30.0, layers[7][1] * 30.0)          # same function, 30x the activations of the last hidden layer
    layers[8] = (layers[8][0] / 30.0, layers[8][1])
    big = E.Engine(layers, oracle_decoder.latent_in, oracle_decoder.code_len, device=0)
    objs = synth.make_batch(2, first_seed=930, n_surface=300, n_background=80)
    ref = _run_traced(big, E.gn_params(num_iterations=3), _args(objs), L.PREPASS_OFF)
    run = _run_traced(big, E.gn_params(num_iterations=3), _args(objs), L.PREPASS_F16, audit=False)
    _assert_identical(run, ref, "rescaled decoder")
    assert run[2]["prepass_guard_trips"] == 0
    big.close()


def test_guard_fires_on_a_margin_that_is_too_small_and_the_rerun_is_exact(oracle_decoder):
    """A margin below the decoder's real prepass error (here forced: 2e-5 with bf16, whose error is ~1e-3): the guard sees re-decoded
    samples that are off by more than half the margin, the objects it tripped on are run again with the prepass off, and the caller gets
    exactly the prepass-off results with prepass_guard_rerun = 1.  The margin was FORCED by the caller, so the trip says nothing about
    the decoder's calibration: the handle's table stays as it was (it used to be raised for the handle's lifetime, and a
    non-finite value could switch the prepass off for good).  The same run with the guard switched off returns DIFFERENT results (so the
    guard is what saved it), as the audit confirms."""
    own = E.Engine(oracle_decoder.layers, oracle_decoder.latent_in, oracle_decoder.code_len, device=0)      # its margins get raised: not shared
    prm = E.gn_params(num_iterations=3)
    objs = synth.make_batch(4, first_seed=970, n_surface=1000, n_background=250)
    ref = _run_traced(own, prm, _args(objs), L.PREPASS_OFF)
    before = own.prepass_calibration_table(L.PREPASS_BF16)
    run = _run_traced(own, prm, _args(objs), L.PREPASS_BF16, delta=2e-5, audit=False)
    st = run[2]
    assert st["prepass_guard_trips"] > 0 and st["prepass_guard_objects"] >= 1 and st["prepass_guard_rerun"] == 1, st
    assert st["prepass_guard_max_err"] >= 1e-5
    _assert_identical(run, ref, "guarded run with a margin that is too small")
    after = own.prepass_calibration_table(L.PREPASS_BF16)
    assert after["guard_err"] == 0.0 and np.array_equal(after["delta"], before["delta"])
    own.prepass_reset_guard()
    assert np.array_equal(own.prepass_calibration_table(L.PREPASS_BF16)["delta"], before["delta"])
    # unguarded, the same margin silently changes set membership
    bad = _run_traced(own, prm, _args(objs), L.PREPASS_BF16, delta=2e-5, audit=True, guard=False)
    assert bad[2]["prepass_misclassified"] > 0 and bad[2]["prepass_guard_rerun"] == 0
    same = all(np.array_equal(ta[k], tc[k]) for ta, tc in zip(bad[1], ref[1]) for k in TRACE_KEYS)
    assert not same, "an unguarded run with misclassified samples should not reproduce prepass-off"
    parity_log(kind="prepass_guard", case="forced margin 2e-5 (bf16)", dtype="bf16", guard_max_err=st["prepass_guard_max_err"],
               delta_zero_code=2e-5, trips=int(st["prepass_guard_trips"]), identical=True, rerun=True)
    own.close()


@pytest.mark.parametrize("dtype", [L.PREPASS_F16, L.PREPASS_BF16])
def test_small_prepass_tiles_give_the_same_values(eng, dtype):
    """The prepass kernel's 64-point-tile form (round 5: one 16-point column block per wave instead of two; chosen automatically where an
    iteration's 128-point tiles would leave most CUs idle -- one detection of SLAM's real size) runs the same arithmetic per point: every
    prepass value is bit-identical to the 128-point form's, for ragged tile ends too."""
    rng = np.random.default_rng(31)
    code = (rng.normal(size=64) * 0.2).astype(np.float32)
    for n in (1, 15, 16, 17, 63, 64, 65, 127, 129, 5000, 16387):
        pts = rng.uniform(-1.0, 1.0, size=(n, 3)).astype(np.float32)
        a = eng.decode_sdf_prepass(code, pts, dtype)
        b = eng.decode_sdf_prepass(code, pts, dtype | L.PREPASS_SMALL_TILES)
        assert np.array_equal(a, b), (n, np.abs(a - b).max())


def test_detection_with_small_prepass_tiles_is_exact_and_faster(eng):
    """One KITTI-size detection: 64-point prepass tiles (automatic for a list this short) against 128-point tiles -- every bit of every
    iteration equal, the same work counters; the timing goes to the log."""
    import time
    det = synth.make_object(4242, n_surface=250, n_background=200)
    prm = E.gn_params()
    out, ms = {}, {}
    for tile in (128, 64, -1):
        b = eng.batch(prm, [det["t_cam_obj_init"]], [det["pts"]], [det["rays"]], [det["depth"]], trace=True)
        b.set_prepass_tile(tile)
        b.run()
        ts = []
        for _ in range(9):
            t0 = time.perf_counter()
            b.run()
            b.results()
            ts.append((time.perf_counter() - t0) * 1e3)
        out[tile] = (b.results(), [b.trace(e) for e in range(10)], b.stats())
        ms[tile] = float(np.median(ts))
        b.close()
    for tile in (64, -1):
        for x, y in zip(out[128][0], out[tile][0]):
            assert np.array_equal(x, y), tile
        for ta, tb in zip(out[128][1], out[tile][1]):
            for k in ("H", "b", "dx", "V", "m", "K", "set_sums", "t_obj_cam", "code"):
                assert np.array_equal(ta[k], tb[k]), (tile, k)
        for k in ("n_prepass_points", "n_jac_points", "n_insphere_points", "n_cluster_tiles"):
            assert out[128][2][k] == out[tile][2][k], (tile, k)
    print("KITTI-size detection p50 (traced batch): 128-point prepass tiles %.3f ms, 64-point %.3f ms, automatic %.3f ms" % (ms[128], ms[64], ms[-1]))
    parity_log(kind="prepass_tile", case="KITTI-size detection, prepass tile 128 vs 64 points", ms_128=ms[128], ms_64=ms[64], ms_auto=ms[-1])
    assert ms[-1] <= ms[128] * 1.02

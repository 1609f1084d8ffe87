"""GPU parity: the HIP path (through the C ABI, via dsp_slam_amd.engine / the reconstruct mirror) against
the oracle on the same seeded inputs and against the committed golden vectors of the reference.

Tolerances (float32 path, north_star: 1e-4 relative):
  * decoder outputs: the MFMA fmaf chain vs BLAS sgemm differ by summation order only -> abs 5e-6 on sdf,
    1e-5 relative (max-norm) on gradients;
  * one Gauss-Newton linearisation from an identical state: H, b, dx within 1e-4 relative (max-norm) when the
    ragged sets (V, m, K) are identical, which is asserted;
  * ten chained iterations: see test_chained_run_within_the_references_own_spread for the sensitivity-calibrated bound.
"""
import json

import numpy as np
import pytest

from conftest import golden, parity_log
from oracle import dsp_oracle as O
from dsp_slam_amd import fixtures, synth, engine as E

pytestmark = pytest.mark.gpu


def rel(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return np.abs(a - b).max() / max(np.abs(b).max(), 1e-30)


@pytest.fixture(scope="module")
def eng(oracle_decoder):
    e = E.Engine(oracle_decoder.layers, oracle_decoder.latent_in, oracle_decoder.code_len, device=0)
    yield e
    e.close()


@pytest.fixture(scope="module")
def eng_random():
    dec = O.fold_decoder(fixtures.random_state_dict(5), fixtures.SPECS)
    e = E.Engine(dec.layers, dec.latent_in, dec.code_len, device=0)
    yield dec, e
    e.close()


def prm_from(cfg):
    return E.params_from_configs(cfg), O.GNParams.from_configs(cfg)


KITTI_CFG = {"optimizer": {"code_len": 64, "num_depth_samples": 50, "cut_off_threshold": 0.01,
                           "joint_optim": {"k1": 1.0, "k2": 100.0, "k3": 0.25, "k4": 1e7, "b1": 0.2, "b2": 0.025,
                                           "num_iterations": 10, "learning_rate": 1.0, "scale_damping": 1.0},
                           "pose_only_optim": {"num_iterations": 5, "learning_rate": 1.0}}}


# ---------------------------------------------------------------------------------------------------
# decoder
# ---------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("n", [1, 15, 16, 63, 64, 65, 1000, 16387])
def test_decode_sdf_vs_oracle(eng, oracle_decoder, n):
    rng = np.random.default_rng(n)
    code = (rng.normal(size=64) * 0.2).astype(np.float32)
    pts = rng.uniform(-1.0, 1.0, size=(n, 3)).astype(np.float32)
    out = eng.decode_sdf(code, pts)
    ref = O.decode_sdf(oracle_decoder, code, pts)
    assert out.shape == ref.shape
    assert np.abs(out - ref).max() < 5e-6


@pytest.mark.parametrize("n", [1, 17, 64, 200, 4099])
def test_sdf_jacobian_vs_oracle(eng, oracle_decoder, n):
    rng = np.random.default_rng(100 + n)
    code = (rng.normal(size=64) * 0.2).astype(np.float32)
    pts = rng.uniform(-1.0, 1.0, size=(n, 3)).astype(np.float32)
    sdf, grad = eng.sdf_jacobian(code, pts)
    y, g = O.get_batch_sdf_jacobian(oracle_decoder, code, pts)
    assert np.abs(sdf - y).max() < 5e-6
    assert np.abs(grad - g).max() < 1e-5 * max(1.0, np.abs(g).max())


def test_decoder_vs_reference_golden(eng):
    g = golden("golden_decoder.npz")
    assert np.abs(eng.decode_sdf(g["code"], g["pts"]) - g["sdf"]).max() < 5e-6
    sdf, grad = eng.sdf_jacobian(g["code"], g["pts"])
    assert np.abs(sdf - g["y_jac"]).max() < 5e-6
    assert rel(grad, g["grad"]) < 2e-5


def test_empty_inputs(eng):
    assert eng.decode_sdf(np.zeros(64, np.float32), np.zeros((0, 3), np.float32)).shape == (0,)


def test_decoder_linear_in_nothing_but_deterministic(eng):
    """Idempotence / determinism at full tile counts: two runs over 300k points are bit-identical and every
    tile position gives the same value for the same point (tile-independent results)."""
    rng = np.random.default_rng(5)
    code = (rng.normal(size=64) * 0.2).astype(np.float32)
    base = rng.uniform(-1, 1, size=(977, 3)).astype(np.float32)
    pts = np.tile(base, (307, 1))                      # 299 939 points, every point at many tile offsets
    a = eng.decode_sdf(code, pts)
    b = eng.decode_sdf(code, pts)
    assert np.array_equal(a, b)
    assert np.array_equal(a.reshape(307, 977), np.broadcast_to(a[:977], (307, 977)))


# ---------------------------------------------------------------------------------------------------
# residual terms vs the reference's golden outputs
# ---------------------------------------------------------------------------------------------------
def test_sdf_term_vs_reference_golden(eng):
    g = golden("golden_terms.npz")
    j7, jc, r = eng.compute_sdf_loss(g["pts"], g["t_obj_cam"], g["code"])
    assert np.abs(r - g["sdf_r"]).max() < 5e-6
    assert rel(j7, g["sdf_j7"]) < 2e-5
    assert rel(jc, g["sdf_jc"]) < 2e-5


def test_render_term_vs_reference_golden(eng):
    g = golden("golden_terms.npz")
    out, st = eng.compute_render_loss(g["rays"], g["depth_obs"], g["t_obj_cam"], g["sampled"], g["code"], th=0.01)
    assert out is not None
    j7, jc, r = out
    assert j7.shape == g["ren_j7"].shape, "ragged set K differs from the reference (%s vs %s)" % (j7.shape, g["ren_j7"].shape)
    assert np.abs(r - g["ren_r"]).max() < 2e-5
    assert rel(j7, g["ren_j7"]) < 5e-5
    assert rel(jc, g["ren_jc"]) < 5e-5


def test_render_term_too_few_samples(eng):
    g = golden("golden_terms.npz")
    far = g["sampled"] + 100.0          # sample range nowhere near the object: no in-sphere samples
    out, st = eng.compute_render_loss(g["rays"], g["depth_obs"], g["t_obj_cam"], far, g["code"], th=0.01)
    assert out is None and st["V"] < 10


# ---------------------------------------------------------------------------------------------------
# Gauss-Newton: every iteration is one linearisation that must match the oracle started from the same state
# ---------------------------------------------------------------------------------------------------
def _run_traced(eng, cfg, obj, code=None):
    prm, oprm = prm_from(cfg)
    b = eng.batch(prm, [obj["t_cam_obj_init"]], [obj["pts"]], [obj["rays"]], [obj["depth"]],
                  None if code is None else [code], trace=True)
    b.run()
    res = b.results()
    traces = [b.trace(e) for e in range(prm.num_iterations)]
    b.close()
    return res, traces, oprm


# chained runs: bound = this x the largest of the reference's 9 recorded round-off draws (see test_chained_run_within_the_references_own_spread).  Measured on
# MI355X (profiles/parity_r03.md): device / spread <= 0.9 on five of the six goldens, 1.07 on `small` (the device is one more draw: it
# exceeds the largest of nine exchangeable draws one time in ten per quantity).  Round 2 used 3.0.
E2E_SPREAD_FACTOR = 1.5
SDF_ROUNDOFF = 2e-7      # the two decoders agree to ~1e-7 (test_decode_sdf_vs_oracle); this is what propagates
LAST_LINEARISATION = {}  # what the last compare_linearisation call measured (written to the parity report by its callers)


def compare_linearisation(tr, i, its, k4, tol_b=1e-4):
    """One GN linearisation of the device (trace tr, object i) against the oracle's from the same state.
    its = (oracle trace on the device's own depth samples, the same with the decoded sdf values jittered by
    +-SDF_ROUNDOFF, the oracle's own derivation of the depth samples).  The device derives T_co, scale and the 50 depth
    samples in fp64 and rounds; the reference does it in fp32 LAPACK / powf; they must agree within 2 ulp, and the
    linearisation is compared on identical samples because H is violently sensitive to them (next paragraph).

    The render rows are de_ds * d sdf with de_ds = (sum_l T_l) / (1 - o_k) * ..., and 1 - o_k = 0.5 + sdf/(2 th) -> 0 at the
    inner edge of the |sdf| < th band: decoder round-off of 1e-7 is amplified by 1/(2 th (1 - o_k)) in exactly the rows
    that dominate H.  So the bound is 1e-4 relative PLUS what the oracle itself moves under that round-off -- when both
    implementations selected exactly the same sample SETS (membership checksums, not just counts).  A sample within
    round-off of a threshold may legitimately switch sets; then the comparison is O(flips / K) and says so.
    Returns True when the strict comparison was made."""
    it, itj, own_depths = its
    nd = own_depths.shape[0]
    assert np.abs(tr["depths"][i][:nd] - own_depths).max() <= 2.5 * np.spacing(np.abs(own_depths).max())
    same_sets = int(tr["set_sums"][i][0]) == it["vsum"] and int(tr["set_sums"][i][1]) == it["ksum"]
    jitter_same = it["vsum"] == itj["vsum"] and it["ksum"] == itj["ksum"]
    # (m is not compared: the device does not decode samples behind a solid one, where the transmittance is exactly 0)
    flips = abs(int(tr["V"][i]) - it["V"]) + abs(int(tr["K"][i]) - it["K"])
    hs, bs = np.abs(it["H"]).max(), np.abs(it["b"]).max()
    mask = np.ones(it["b"].shape[0], bool)
    mask[3:6] = False
    if same_sets and jitter_same:
        assert flips == 0
        amp_h = np.abs(itj["H"] - it["H"]).max()
        amp_b = np.abs(itj["b"] - it["b"])[mask].max()
        assert np.abs(tr["H"][i] - it["H"]).max() < 1e-4 * hs + 2 * amp_h
        assert np.abs(tr["b"][i][mask] - it["b"][mask]).max() < tol_b * bs + 2 * amp_b
        # rotation-prior entries: k4 * J_rot * (1 + R_co[1,1]) is ulp-quantised in fp32 (see test_oracle_golden)
        j_rot = np.sqrt(np.abs(np.diag(it["H"])[3:6]) / max(k4, 1.0))
        tol_rot = k4 * (j_rot + 1e-3) * 2.4e-7 + tol_b * bs + 2 * amp_b
        assert np.all(np.abs(tr["b"][i][3:6] - it["b"][3:6]) <= tol_rot)
        # dx = H^-1 b: whatever difference is accepted on b (above) maps to |H^-1| tol_b on dx -- near convergence b, hence dx, is
        # a difference of large terms and a bound relative to |dx| alone would be ill-posed
        tol_bv = np.full(it["b"].shape[0], tol_b * bs + 2 * amp_b)
        tol_bv[3:6] = np.maximum(tol_bv[3:6], tol_rot)
        tol_dx = np.abs(np.linalg.inv(it["H"].astype(np.float64))) @ tol_bv
        assert np.all(np.abs(tr["dx"][i] - it["dx"]) <= 2e-4 * np.abs(it["dx"]).max() + 2 * np.abs(itj["dx"] - it["dx"]).max() + tol_dx)
        LAST_LINEARISATION.update(same_sets=True, flips=0, rel_H=float(np.abs(tr["H"][i] - it["H"]).max() / hs),
                                  rel_b=float(np.abs(tr["b"][i][mask] - it["b"][mask]).max() / bs), oracle_jitter_rel_H=float(amp_h / hs),
                                  V=int(it["V"]), K=int(it["K"]))
        return amp_h < 1e-3 * hs
    assert flips <= max(4, it["K"] // 250), "too many threshold flips: V %d/%d m %d/%d K %d/%d" % (
        tr["V"][i], it["V"], tr["m"][i], it["m"], tr["K"][i], it["K"])   # m informational
    loose = 8.0 * max(flips, 2) / max(it["K"], 1)
    LAST_LINEARISATION.update(same_sets=False, flips=int(flips), rel_H=float(np.abs(tr["H"][i] - it["H"]).max() / hs),
                              rel_b=float(np.abs(tr["b"][i][mask] - it["b"][mask]).max() / bs),
                              oracle_jitter_rel_H=float(np.abs(itj["H"] - it["H"]).max() / hs), V=int(it["V"]), K=int(it["K"]))
    assert np.abs(tr["H"][i] - it["H"]).max() < loose * hs + 4 * np.abs(itj["H"] - it["H"]).max()
    assert np.abs(tr["b"][i][mask] - it["b"][mask]).max() < loose * bs + 4 * np.abs(itj["b"] - it["b"])[mask].max()
    return False


def one_iteration_oracle(oracle_decoder, oprm, obj, tr, i=0):
    o1 = O.GNParams(oprm.k1, oprm.k2, oprm.k3, oprm.k4, oprm.b1, oprm.b2, oprm.lr, oprm.s_damp, 1, oprm.code_len,
                    oprm.num_depth_samples, oprm.cut_off)
    out = []
    for jit in (0.0, SDF_ROUNDOFF):
        otr = []
        O.reconstruct_object(oracle_decoder, o1, None, obj["pts"], obj["rays"], obj["depth"], tr["code"][i], trace=otr,
                             t_obj_cam0=tr["t_obj_cam"][i], sdf_jitter=jit, sampled_override=tr["depths"][i])
        out.append(otr[0])
    otr = []
    O.reconstruct_object(oracle_decoder, o1, None, obj["pts"], obj["rays"], obj["depth"], tr["code"][i], trace=otr,
                         t_obj_cam0=tr["t_obj_cam"][i])
    out.append(otr[0]["depths"])
    return tuple(out)


def explain_flips(eng, prm, oprm, oracle_decoder, obj, tr, code_in=None):
    """The device and the oracle selected different sample sets from the SAME state and depth samples (trace tr): re-run that one
    iteration, fetch the device's per-sample decisions and name every differing sample with its distance to the threshold it crossed
    (tests/forensics.py).  Each must lie within round-off of the threshold -- nothing drifted, so the margins are the tight ones."""
    import forensics as F
    n_rays, n_d = obj["rays"].shape[0], oprm.num_depth_samples
    b = eng.batch(prm, [obj["t_cam_obj_init"]], [obj["pts"]], [obj["rays"]], [obj["depth"]], trace=True)
    t1, st = F.device_linearisation(b, tr["t_obj_cam"][0], tr["code"][0], tr["depths"][0][:n_d])
    assert st == 0 and np.array_equal(t1["set_sums"][0], tr["set_sums"][0]), "the re-run from the traced state does not reproduce the traced sets"
    m, sdf, deds = b.debug_samples(0, n_rays, n_d)
    b.close()
    ot = F.oracle_linearisation(oracle_decoder, oprm, obj["pts"], obj["rays"], obj["depth"], tr["t_obj_cam"][0], tr["code"][0][:oprm.code_len], tr["depths"][0][:n_d])
    flips = F.name_flips(m, sdf, deds, F.oracle_grids(ot["sets"], n_rays, n_d), oprm.cut_off)
    assert flips, "checksums differ but no differing sample was found"
    assert all(f["explained"] for f in flips), flips
    return flips


def _check_iterations(oracle_decoder, obj, traces, oprm, k4, name="", explain=None):
    """Every iteration strict (identical sets, 1e-4) -- or, where the sets differ, every differing sample named and within round-off of
    its threshold (explain = (engine, device params); without it a non-strict iteration fails)."""
    strict, per_iter, named = 0, [], []
    for e, tr in enumerate(traces):
        ok = bool(compare_linearisation(tr, 0, one_iteration_oracle(oracle_decoder, oprm, obj, tr), k4))
        strict += ok
        per_iter.append(dict(LAST_LINEARISATION))
        if not LAST_LINEARISATION["same_sets"]:
            assert explain is not None, "iteration %d: sample sets differ from the oracle's at the same state" % e
            fl = explain_flips(explain[0], explain[1], oprm, oracle_decoder, obj, tr)
            named.append(dict(iteration=e, flips=[(f["ray"], f["depth_index"], f["threshold"], f["margin"]) for f in fl]))
    parity_log(kind="iterations", case=name, n=len(traces), strict=strict, same_sets=sum(1 for p in per_iter if p["same_sets"]),
               flips=[p["flips"] for p in per_iter], rel_H=[p["rel_H"] for p in per_iter], rel_b=[p["rel_b"] for p in per_iter],
               oracle_jitter_rel_H=[p["oracle_jitter_rel_H"] for p in per_iter], K=[p["K"] for p in per_iter], named_flips=named)
    n_same = sum(1 for p in per_iter if p["same_sets"])
    assert n_same + len(named) == len(traces) and len(named) <= 1, "iterations whose sets differ from the oracle's: %s" % named      # measured on MI355X: 0


def test_reconstruct_small_each_iteration(eng, oracle_decoder):
    g = golden("golden_recon_small.npz")
    cfg = json.loads(str(g["cfg_json"]))
    obj = dict(t_cam_obj_init=g["in_t_cam_obj_init"], pts=g["in_pts"], rays=g["in_rays"], depth=g["in_depth"])
    res, traces, oprm = _run_traced(eng, cfg, obj)
    assert res[3][0] == 0
    prm, _ = prm_from(cfg)
    _check_iterations(oracle_decoder, obj, traces, oprm, cfg["optimizer"]["joint_optim"]["k4"], "small (KITTI hyper-parameters)", explain=(eng, prm))
    # (against the reference's OWN recorded numbers, at its own states: tests/test_gpu_forensics.py::test_linearisation_at_reference_states)


def test_reconstruct_redwood_each_iteration(eng, oracle_decoder):
    g = golden("golden_recon_redwood.npz")
    cfg = json.loads(str(g["cfg_json"]))
    obj = dict(t_cam_obj_init=g["in_t_cam_obj_init"], pts=g["in_pts"], rays=g["in_rays"], depth=g["in_depth"])
    res, traces, oprm = _run_traced(eng, cfg, obj, code=g["in_code"])
    assert res[3][0] == 0 and len(traces) == 5
    _check_iterations(oracle_decoder, obj, traces, oprm, cfg["optimizer"]["joint_optim"]["k4"], "redwood", explain=(eng, prm_from(cfg)[0]))


def end_to_end_differences(g, t44, code):
    """Chained result against a golden: rotation (R / scale: absolute), scale (relative), translation (relative to |t|), code
    (absolute), whole matrix (absolute) -- and the same quantities for the reference's own spread (its 1 + 8 re-runs with every input
    element moved to an adjacent float32, golden fields ulp_* / ulps_*).  Returns (measured, spread, number of draws)."""
    def split(m44):
        m44 = np.asarray(m44, np.float64)
        sc = np.cbrt(np.linalg.det(m44[:3, :3]))
        return m44[:3, :3] / sc, sc, m44[:3, 3]

    r_g, s_g, p_g = split(g["t_cam_obj"])

    def diffs(m44, cd):
        r_a, s_a, p_a = split(m44)
        return dict(rot=float(np.abs(r_a - r_g).max()), scale=float(abs(s_a - s_g) / s_g),
                    trans=float(np.linalg.norm(p_a - p_g) / np.linalg.norm(p_g)), code=float(np.abs(cd - g["code"]).max()),
                    t_abs=float(np.abs(np.asarray(m44, np.float64) - g["t_cam_obj"]).max()))

    m = diffs(t44, code)
    draws = [diffs(g["ulp_t_cam_obj"], g["ulp_code"])] + [diffs(a, c) for a, c in zip(g["ulps_t_cam_obj"], g["ulps_code"])]
    # the reference's spread with NO input change at all: the same arrays at torch.set_num_threads(1) and (4) instead of the recording's 8
    # (tools/make_golden_threads.py) -- only the accumulation order of its CPU sgemm / reductions differs.  The yardstick is the larger of the two.
    files = g.files if hasattr(g, "files") else g.keys()
    if "thr_t_cam_obj" in files:
        thr = [diffs(a, c) for a, c in zip(g["thr_t_cam_obj"], g["thr_code"])]
        m["thread_spread"] = {k: max(d[k] for d in thr) for k in ("rot", "scale", "trans", "code")}
        draws += thr
    sens = {k: max(d[k] for d in draws) for k in ("rot", "scale", "trans", "code", "t_abs")}
    return m, sens, len(draws)


@pytest.mark.parametrize("name", ["golden_recon_small.npz", "golden_recon_cfg1.npz", "golden_recon_redwood.npz", "golden_recon_freiburg.npz",
                                  "golden_recon_cfg2.npz"])
def test_chained_run_within_the_references_own_spread(eng, name):
    """All iterations chained, against the reference's final pose / code (north_star: 1e-4 relative).

    Tolerance per quantity: 1e-4, or E2E_SPREAD_FACTOR x the REFERENCE'S OWN spread when every element of its inputs moves to an adjacent
    float32 (golden ulps_*: 8 seeded draws + the original one-direction draw; tools/make_golden_sensitivity.py), whichever is larger.
    The chained map amplifies round-off (set flips and up to x100 per iteration without flips, measured in the
    reference itself), so this comparison is a sanity bound; what pins parity is tests/test_gpu_forensics.py -- the device at the
    reference's own recorded states (identical sets, H / b to 1e-5, all iterations) and the chained run decomposed step by step.
    Rotation (R / scale: absolute), scale (relative), translation (relative to |t|) and code (absolute) are checked separately so that the
    18 m translation does not set the scale for the rotation entries.  The measured differences go to profiles/parity_rNN.md."""
    g = golden(name)
    cfg = json.loads(str(g["cfg_json"]))
    prm, oprm = prm_from(cfg)
    code0 = [g["in_code"]] if "in_code" in g.files else None
    t, code, loss, status = eng.reconstruct_batch(prm, [g["in_t_cam_obj_init"]], [g["in_pts"]], [g["in_rays"]], [g["in_depth"]], code0)
    assert status[0] == 0 and bool(g["is_good"])

    m, sens, n_draws = end_to_end_differences(g, t[0], code[0])
    rec = {k: m[k] for k in m if k != "thread_spread"}
    rec.update({k + "_sens": sens[k] for k in sens})
    rec.update({k + "_thread_spread": v for k, v in m.get("thread_spread", {}).items()})
    rec["loss"] = float(abs(loss[0] - float(g["loss"])) / max(abs(float(g["loss"])), 1e-12))
    parity_log(kind="end_to_end", case=name, n_draws=n_draws, **rec)
    print("%s: rot %.2e (reference spread under 1-ulp inputs %.2e) scale %.2e (%.2e) trans %.2e (%.2e) code %.2e (%.2e)" % (
        name, m["rot"], sens["rot"], m["scale"], sens["scale"], m["trans"], sens["trans"], m["code"], sens["code"]))
    for q in ("rot", "scale", "trans", "code"):
        assert m[q] <= max(1e-4, E2E_SPREAD_FACTOR * sens[q]), (q, m[q], sens[q])


def test_failure_path_is_good_false(eng_random):
    dec, e = eng_random
    g = golden("golden_recon_fail.npz")
    prm, oprm = prm_from(json.loads(str(g["cfg_json"])))
    t, code, loss, status = e.reconstruct_batch(prm, [g["in_t_cam_obj_init"]], [g["in_pts"]], [g["in_rays"]], [g["in_depth"]])
    assert status[0] == 2 and not bool(g["is_good"])
    assert loss[0] == float(g["loss"]) == 0.0


def test_pose_only_vs_reference_golden(eng, oracle_decoder):
    g = golden("golden_pose_only.npz")
    prm, oprm = prm_from(KITTI_CFG)
    out = eng.estimate_pose_batch(prm, [g["t_co_se3"]], [float(g["scale"])], [g["pts"]], [g["code"]])
    assert rel(out[0], g["out"]) < 1e-4
    ref = O.estimate_pose_cam_obj(oracle_decoder, oprm, g["t_co_se3"], float(g["scale"]), g["pts"], g["code"])
    assert rel(out[0], ref) < 1e-4


def test_ragged_batch_equals_single_runs(eng):
    """Objects are independent: a ragged batch (different M, R, one object that fails) gives each object the
    result it gets alone, bit for bit."""
    prm, _ = prm_from(KITTI_CFG)
    objs = [synth.make_object(40, 130, 20), synth.make_object(41, 64, 0), synth.make_object(42, 257, 33)]
    bad = synth.make_object(43, 50, 10)
    bad["t_cam_obj_init"] = bad["t_cam_obj_init"].copy()
    bad["t_cam_obj_init"][:3, 3] += 500.0          # nowhere near its rays: < 10 in-sphere samples
    objs.insert(1, bad)
    tb, cb, lb, sb = eng.reconstruct_batch(prm, [o["t_cam_obj_init"] for o in objs], [o["pts"] for o in objs],
                                           [o["rays"] for o in objs], [o["depth"] for o in objs])
    assert sb[1] == 1 and list(sb[[0, 2, 3]]) == [0, 0, 0]
    for i, o in enumerate(objs):
        t1, c1, l1, s1 = eng.reconstruct_batch(prm, [o["t_cam_obj_init"]], [o["pts"]], [o["rays"]], [o["depth"]])
        assert s1[0] == sb[i]
        assert np.array_equal(t1[0], tb[i]) and np.array_equal(c1[0], cb[i]) and l1[0] == lb[i]


def test_python_mirror_api(cars_state_dict, tmp_path):
    """The reference's call surface on the mirror package, with Eigen-style Fortran-ordered inputs."""
    import os
    import sys
    pkg = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "dsp_slam_amd")
    sys.path.insert(0, pkg)
    try:
        for m in [k for k in sys.modules if k.split(".")[0] in ("reconstruct", "deep_sdf")]:
            del sys.modules[m]
        from reconstruct.utils import get_configs, get_decoder
        from reconstruct.optimizer import Optimizer, MeshExtractor
        ddir = fixtures.materialize_decoder_dir("cars", str(tmp_path / "cars_64"))
        cfg_d = dict(KITTI_CFG, data_type="KITTI", DeepSDF_DIR=ddir, voxels_dim=32)
        with open(tmp_path / "cfg.json", "w") as f:
            json.dump(cfg_d, f)
        cfg = get_configs(str(tmp_path / "cfg.json"))
        decoder = get_decoder(cfg)
        opt = Optimizer(decoder, cfg)
        opt.verbose = False
        assert opt.code_len == 64
        g = golden("golden_recon_small.npz")
        f_order = lambda a: np.asfortranarray(a)          # noqa: E731  what pybind11's Eigen caster hands over
        rst = opt.reconstruct_object(f_order(g["in_t_cam_obj_init"]), f_order(g["in_pts"]), f_order(g["in_rays"]), g["in_depth"])
        assert rst.is_good is True and rst.t_cam_obj.shape == (4, 4) and rst.t_cam_obj.dtype == np.float32
        assert rst.code.shape == (64,) and float(rst.loss) > 0
        with pytest.raises(KeyError):
            rst.no_such_field
        assert Optimizer.get_shape_code(rst) is rst.code
        gp = golden("golden_pose_only.npz")
        t_se3 = f_order(gp["t_co_se3"])
        out = opt.estimate_pose_cam_obj(t_se3, float(gp["scale"]), f_order(gp["pts"]), gp["code"])
        assert tuple(out.shape) == (4, 4) and rel(out.numpy(), gp["out"]) < 1e-4
        grid = MeshExtractor(decoder, 64, 16).decode_grid(rst.code)
        assert grid.shape == (16, 16, 16) and np.isfinite(grid).all() and grid.min() < 0 < grid.max()
    finally:
        sys.path.remove(pkg)
        for m in [k for k in sys.modules if k.split(".")[0] in ("reconstruct", "deep_sdf")]:
            del sys.modules[m]

"""Chained-result parity, demonstrated instead of argued.

(c) test_linearisation_at_reference_states: the device linearises at the REFERENCE'S OWN recorded states -- camera->object matrix, code
    and depth samples injected bit for bit (dsp_batch_set_start_state) -- for EVERY iteration of every recorded run, and H, b, dx, V, K are
    compared with what the unmodified reference recorded (tests/golden/golden_recon_*.npz: it_*), not with the oracle.  Every iteration
    must be strict (identical sets, 1e-4) unless every differing sample is NAMED and lies within round-off of the threshold it crossed.
(b) test_chained_divergence_is_the_maps_own: the chained device run is laid beside the reference's recorded trajectory iteration by
    iteration and every step's difference is decomposed into the device's LOCAL error (device step vs oracle step from the device's own
    state: asserted, identical sets or named samples, dx to 1e-4) and the PROPAGATED part (the map's own response to the state
    difference that came in: reported).  Where the device's sets first depart from the recorded ones, the samples that switched are
    listed with their distance to the threshold (`|‖p‖-1|`, `||sdf|-th|`, `|de_do-1e-2|`).
Reports go to gpurun_out/parity/ (parity_log) and gpurun_out/forensics_<case>.md; tools/make_parity_report.py collects them.
"""
import json
import os

import numpy as np
import pytest

import forensics as F
from conftest import ROOT, golden, parity_log
from oracle import dsp_oracle as O
from dsp_slam_amd import engine as E

pytestmark = pytest.mark.gpu

# the last two run on the 32-D chairs decoder; cfg5 is BASELINE configs[4] at full size (4000 surface points + 500 background rays,
# Redwood hyper-parameters) -- and a fixture on which the reference's OWN 1-ulp spread is below 1e-4 (9.3e-5 pose / 4.4e-5 code)
CASES = ["golden_recon_small.npz", "golden_recon_cfg1.npz", "golden_recon_redwood.npz", "golden_recon_freiburg.npz", "golden_recon_cfg2.npz",
         "golden_recon_chairs32.npz", "golden_recon_cfg5.npz"]
# round 5: one cfg2-size object on the decoder fitted to the complex (non-convex, all-64-dims) shape family
if os.path.exists(os.path.join(ROOT, "tests", "golden", "golden_recon_complex.npz")) and os.path.exists(os.path.join(ROOT, "tests", "golden", "decoder_complex.npz")):
    CASES.append("golden_recon_complex.npz")


class _Engines(object):
    """The engine + oracle decoder a golden was recorded with (64-D cars or 32-D chairs), created on first use."""

    def __init__(self, decoders):
        self.decoders, self.engines = decoders, {}

    def __call__(self, code_len):
        if code_len not in self.engines:
            d = self.decoders[code_len]
            if callable(d):                  # created on first use (the complex fixture may be absent)
                d = self.decoders[code_len] = d()
            self.engines[code_len] = E.Engine(d.layers, d.latent_in, d.code_len, device=0)
        return self.engines[code_len], self.decoders[code_len]

    def close(self):
        for e in self.engines.values():
            e.close()


@pytest.fixture(scope="module")
def eng(oracle_decoder, chairs32_decoder):
    def complex_dec():
        from dsp_slam_amd import fixtures
        return O.fold_decoder(fixtures.load_decoder_npz(fixtures.fixture_path("complex")), fixtures.fixture_specs("complex"))
    es = _Engines({64: oracle_decoder, 32: chairs32_decoder, "complex": complex_dec})
    yield es
    es.close()


def _setup(engines, g):
    cfg = json.loads(str(g["cfg_json"]))
    prm, oprm = E.params_from_configs(cfg), O.GNParams.from_configs(cfg)
    e, dec = engines("complex" if os.path.basename(cfg["DeepSDF_DIR"]).startswith("complex") else cfg["optimizer"]["code_len"])
    code0 = [g["in_code"]] if "in_code" in g.files else None
    b = e.batch(prm, [g["in_t_cam_obj_init"]], [g["in_pts"]], [g["in_rays"]], [g["in_depth"]], code0, trace=True)
    return cfg, prm, oprm, b, dec


def _rot_prior_bound(h_ref, k4):
    """b[3:6] carries k4 * J_rot * (1 + R_co[1,1]) with k4 = 1e7: the residual is a difference of two numbers ~1, i.e. quantised to
    ulp(1) = 1.2e-7 BEFORE the factor 1e7 -- in the reference itself.  Two ulp of that, times the jacobian entry."""
    j_rot = np.sqrt(np.abs(np.diag(h_ref)[3:6]) / max(k4, 1.0))
    return k4 * (j_rot + 1e-3) * 2.4e-7


def _write_report(case, lines):
    out = os.path.join(ROOT, "gpurun_out")
    os.makedirs(out, exist_ok=True)
    with open(os.path.join(out, "forensics_%s.md" % case.replace("golden_recon_", "").replace(".npz", "")), "a") as f:
        f.write("\n".join(lines) + "\n\n")


def _fmt_flip(f):
    return "ray %d depth %d: %s device=%s reference=%s, threshold `%s`, oracle value %.9g, margin %.2e%s" % (
        f["ray"], f["depth_index"], f["set"], f["device"], f["oracle"], f["threshold"], f["value"], f["margin"], "" if f["explained"] else "  <-- NOT within round-off")


@pytest.mark.parametrize("name", CASES)
def test_linearisation_at_reference_states(eng, name):
    g = golden(name)
    cfg, prm, oprm, b, oracle_decoder = _setup(eng, g)
    k4 = cfg["optimizer"]["joint_optim"]["k4"]
    n_it = g["it_H"].shape[0]
    n_rays, n_d = g["in_rays"].shape[0], oprm.num_depth_samples
    obj = dict(pts=g["in_pts"], rays=g["in_rays"], depth=g["in_depth"])
    rows, strict, report = [], 0, ["## %s: device linearised at the reference's recorded states (injected bit for bit)" % name, "",
                                   "| it | V dev/ref | K dev/ref | rel dH | rel db | rel ddx | rel dloss | flips |", "|---|---|---|---|---|---|---|---|"]
    n_unk = 7 + oprm.code_len                # 71, or 39 with the 32-D decoder (the device carries 64 code slots; the unused ones are pinned)
    mask = np.ones(n_unk, bool)
    mask[3:6] = False
    for e in range(n_it):
        tr, status = F.device_linearisation(b, g["it_t_obj_cam"][e], g["it_code"][e], g["it_depths"][e])
        assert status == 0
        tr = dict(tr, H=tr["H"][:, :n_unk, :n_unk], b=tr["b"][:, :n_unk], dx=tr["dx"][:, :n_unk])
        # the state really is the reference's, bit for bit
        assert np.array_equal(tr["t_obj_cam"][0], g["it_t_obj_cam"][e]) and np.array_equal(tr["code"][0][:g["it_code"].shape[1]], g["it_code"][e])
        assert np.array_equal(tr["depths"][0][:n_d], g["it_depths"][e])
        v_dev, k_dev = int(tr["V"][0]), int(tr["K"][0])
        v_ref, k_ref = int(g["it_V"][e]), int(g["it_K"][e])
        h_ref, b_ref, dx_ref = g["it_H"][e], g["it_b"][e], g["it_dx"][e]
        rh, rb = F.rel_max(tr["H"][0], h_ref), F.rel_max(tr["b"][0][mask], b_ref[mask])
        rdx = F.rel_max(tr["dx"][0], dx_ref)
        # the result's `loss` field after one iteration from this state IS the reference's loss at this state (optimizer.py:155,200-203;
        # the field LocalMapping_util.cc:405-406 branches on); it_loss: the reference's own functions at the recorded state
        # (tools/make_golden_it_loss.py; its last entry is bit-identical to the loss the recorded run returned)
        rl = F.loss_rel(tr["loss"][0], g["it_loss"][e])
        if e == n_it - 1:
            assert float(g["it_loss"][e]) == float(g["loss"])
        flips = []
        if (v_dev, k_dev) != (v_ref, k_ref):
            ot = F.oracle_linearisation(oracle_decoder, oprm, obj["pts"], obj["rays"], obj["depth"], g["it_t_obj_cam"][e], g["it_code"][e], g["it_depths"][e])
            assert (ot["V"], ot["K"]) == (v_ref, k_ref), "the oracle does not reproduce the reference's sets at its own state"
            m, sdf, deds = b.debug_samples(0, n_rays, n_d)
            flips = F.name_flips(m, sdf, deds, F.oracle_grids(ot["sets"], n_rays, n_d), oprm.cut_off)
            assert flips, "set sizes differ but no differing sample was found"
            assert all(f["explained"] for f in flips), "\n".join(_fmt_flip(f) for f in flips)
            assert len(flips) <= 4
            assert rl <= F.LOSS_RTOL_FLIPPED, (e, rl)
            report += ["", "iteration %d:" % e] + ["* " + _fmt_flip(f) for f in flips] + [""]
        else:
            strict += 1
            assert rl <= F.LOSS_RTOL, (e, float(tr["loss"][0]), float(g["it_loss"][e]))
            # measured on MI355X (profiles/parity_r03.md): rel dH <= 9.2e-6, rel db <= 3.7e-5 over all 35 recorded iterations; the bounds
            # leave a factor of three, so a 10x regression of the per-step agreement fails here
            assert rh < 3e-5, (e, rh)
            assert rb < 1.2e-4, (e, rb)
            assert np.all(np.abs(tr["b"][0][3:6] - b_ref[3:6]) <= _rot_prior_bound(h_ref, k4) + 2e-4 * np.abs(b_ref).max())
            # dx = H^-1 b inherits what is accepted on b through |H^-1| (near convergence b, hence dx, is a difference of large terms)
            tol_b = np.full(n_unk, 2e-4 * np.abs(b_ref[mask]).max())
            tol_b[3:6] += _rot_prior_bound(h_ref, k4)
            tol_dx = np.abs(np.linalg.inv(h_ref.astype(np.float64))) @ tol_b + 1e-4 * np.abs(dx_ref).max()
            assert np.all(np.abs(tr["dx"][0] - dx_ref) <= tol_dx), (e, np.abs(tr["dx"][0] - dx_ref).max(), tol_dx.max())
        rows.append(dict(V=(v_dev, v_ref), K=(k_dev, k_ref), rel_H=rh, rel_b=rb, rel_dx=rdx, rel_loss=rl, flips=len(flips)))
        report.append("| %d | %d / %d | %d / %d | %.2e | %.2e | %.2e | %.2e | %d |" % (e, v_dev, v_ref, k_dev, k_ref, rh, rb, rdx, rl, len(flips)))
    b.close()
    _write_report(name, report)
    parity_log(kind="at_reference_states", case=name, n=n_it, strict=strict, rel_H=[r["rel_H"] for r in rows], rel_b=[r["rel_b"] for r in rows],
               rel_dx=[r["rel_dx"] for r in rows], rel_loss=[r["rel_loss"] for r in rows], flips=[r["flips"] for r in rows], K=[r["K"][1] for r in rows])
    # measured on MI355X: strict in 45 of 45 recorded iterations (profiles/parity_r03.md); one named flip per run is the most a changed
    # instruction schedule could plausibly add
    assert strict >= n_it - 1, "more than one iteration with (named) flips at the reference's own states: %d of %d strict" % (strict, n_it)


@pytest.mark.parametrize("name", CASES)
def test_chained_divergence_is_the_maps_own(eng, name):
    """The chained device run beside the reference's recorded trajectory, with the difference of every step DECOMPOSED:

        state_dev(e+1) - state_ref(e+1)  =  [ step_dev(state_dev(e)) - step_oracle(state_dev(e)) ]        local: the device's own error
                                          + [ step_oracle(state_dev(e)) - step_ref(state_ref(e)) ]        propagated: the MAP's response to the
                                                                                                          difference that came in
    (step_oracle == step_ref at identical states to ~1e-6: tests/test_oracle_golden.py checks every recorded state.)  The local term is
    asserted for EVERY iteration: identical sample sets -- else every differing sample is named and must lie within round-off of its
    threshold -- and dx within 1e-4.  The propagated term is reported with the amplification it implies; it is what a 1-ulp change of
    the inputs does to the unmodified reference (golden ulps_*: the yardstick of the final bound), including set flips, which are
    named where the device's sets first depart from the recorded ones."""
    g = golden(name)
    cfg, prm, oprm, b, oracle_decoder = _setup(eng, g)
    k4 = cfg["optimizer"]["joint_optim"]["k4"]
    n_it = g["it_H"].shape[0]
    n_rays, n_d = g["in_rays"].shape[0], oprm.num_depth_samples
    b.run()
    t_fin, code_fin, loss, status = b.results()
    assert status[0] == 0
    traces = [b.trace(e) for e in range(n_it)]
    report = ["## %s: chained device run beside the reference's recorded trajectory" % name, "",
              "| it | incoming state diff (rot / trans / code) | V dev/ref | K dev/ref | local: rel d(dx) dev vs oracle at the device's state | propagated: rel d(dx) oracle(dev state) vs reference | local flips |",
              "|---|---|---|---|---|---|---|"]
    n_unk = 7 + oprm.code_len
    traces = [dict(tr, H=tr["H"][:, :n_unk, :n_unk], b=tr["b"][:, :n_unk], dx=tr["dx"][:, :n_unk]) for tr in traces]
    local, prop, drift_in, first = [], [], [], None
    for e, tr in enumerate(traces):
        sd = F.state_difference(tr["t_obj_cam"][0], tr["code"][0], g["it_t_obj_cam"][e], g["it_code"][e])
        drift_in.append(max(sd.values()))
        if first is None and (int(tr["V"][0]), int(tr["K"][0])) != (int(g["it_V"][e]), int(g["it_K"][e])):
            first = e
        ot = F.oracle_linearisation(oracle_decoder, oprm, g["in_pts"], g["in_rays"], g["in_depth"], tr["t_obj_cam"][0], tr["code"][0][:oprm.code_len], tr["depths"][0][:n_d])
        same = int(tr["set_sums"][0][0]) == ot["vsum"] and int(tr["set_sums"][0][1]) == ot["ksum"]
        flips = []
        if not same:      # the device and the oracle disagree at the SAME state and depths: name the samples (tight margins: nothing drifted)
            t1, st1 = F.device_linearisation(b, tr["t_obj_cam"][0], tr["code"][0], tr["depths"][0][:n_d])
            assert st1 == 0 and np.array_equal(t1["set_sums"][0], tr["set_sums"][0])
            m, sdf, deds = b.debug_samples(0, n_rays, n_d)
            flips = F.name_flips(m, sdf, deds, F.oracle_grids(ot["sets"], n_rays, n_d), oprm.cut_off)
            assert flips and all(f["explained"] for f in flips), "\n".join(_fmt_flip(f) for f in flips)
            assert len(flips) <= 4
            report += ["", "iteration %d, device vs oracle at the same state:" % e] + ["* " + _fmt_flip(f) for f in flips] + [""]
        loc = F.rel_max(tr["dx"][0], ot["dx"])
        pro = F.rel_max(ot["dx"], g["it_dx"][e])
        local.append(loc)
        prop.append(pro)
        if same:
            tol_b = np.full(n_unk, 1e-4 * np.abs(ot["b"]).max())
            tol_b[3:6] += _rot_prior_bound(ot["H"], k4)
            tol_dx = np.abs(np.linalg.inv(ot["H"].astype(np.float64))) @ tol_b + 1e-4 * np.abs(ot["dx"]).max()
            assert F.rel_max(tr["H"][0], ot["H"]) < 1e-4, (e, F.rel_max(tr["H"][0], ot["H"]))
            assert np.all(np.abs(tr["dx"][0] - ot["dx"]) <= tol_dx), (e, loc)
            if e == n_it - 1:
                # the chained run's returned `loss` is the loss at the device's OWN last linearisation point (optimizer.py:155,200-203): against
                # the oracle evaluated at exactly that state (the oracle reproduces the reference's it_loss at every recorded state to 1e-5,
                # tests/test_oracle_golden.py) -- a statement about the returned field that the chained map's chaos cannot blur
                chained_loss_rel = F.loss_rel(loss[0], ot["loss"])
                assert chained_loss_rel <= F.LOSS_RTOL, (float(loss[0]), ot["loss"])
        report.append("| %d | %.1e / %.1e / %.1e | %d / %d | %d / %d | %.2e | %.2e | %d |" % (
            e, sd["rot"], sd["trans"], sd["code"], tr["V"][0], g["it_V"][e], tr["K"][0], g["it_K"][e], loc, pro, len(flips)))
    import test_gpu_parity as P
    m, sens, n_draws = P.end_to_end_differences(g, t_fin[0], code_fin[0])
    report += ["", "final: rot %.2e scale %.2e trans %.2e code %.2e  (reference's own spread under 1-ulp inputs, %d draws: %.2e / %.2e / %.2e / %.2e)" % (
        m["rot"], m["scale"], m["trans"], m["code"], n_draws, sens["rot"], sens["scale"], sens["trans"], sens["code"])]
    named = []
    if first is not None:
        # where the device's sets first depart from the RECORDED ones: the samples, with margins widened by the state difference that came in
        t1, st1 = F.device_linearisation(b, traces[first]["t_obj_cam"][0], traces[first]["code"][0], traces[first]["depths"][0][:n_d])
        mask, sdf, deds = b.debug_samples(0, n_rays, n_d)
        ot = F.oracle_linearisation(oracle_decoder, oprm, g["in_pts"], g["in_rays"], g["in_depth"], g["it_t_obj_cam"][first], g["it_code"][first], g["it_depths"][first])
        assert (ot["V"], ot["K"]) == (int(g["it_V"][first]), int(g["it_K"][first]))
        named = F.name_flips(mask, sdf, deds, F.oracle_grids(ot["sets"], n_rays, n_d), oprm.cut_off)
        scale = float(np.cbrt(np.linalg.det(np.linalg.inv(g["it_t_obj_cam"][first].astype(np.float64))[:3, :3])))
        tol = F.flip_tolerances(drift_in[first], float(np.abs(traces[first]["depths"][0][:n_d] - g["it_depths"][first]).max()) / scale)
        for f in named:
            f["explained"] = bool(f["margin"] <= tol[f["threshold"]])
        report += ["", "the device's sets first depart from the recorded ones at iteration %d (incoming state difference %.1e):" % (first, drift_in[first])]
        report += ["* " + _fmt_flip(f) for f in named[:12]] + (["* ... %d more" % (len(named) - 12)] if len(named) > 12 else [])
    else:
        report.append("the device selected the reference's sample sets in all %d iterations" % n_it)
    _write_report(name, report)
    parity_log(kind="chained_forensic", case=name, first_flip_iteration=first, final=m, reference_spread=sens, local_rel_dx=local,
               loss_vs_reference_rel=F.loss_rel(loss[0], g["loss"]),
               propagated_rel_dx=prop, incoming_state_diff=drift_in, flips_named=len(named), flips_explained=sum(1 for f in named if f["explained"]))
    b.close()
    if first is not None and drift_in[first] <= 2e-5:
        assert named and all(f["explained"] for f in named), "\n".join(_fmt_flip(f) for f in named)
    # Where the REFERENCE is stable in a quantity (its own 1-ulp spread below 1e-4: Freiburg, cfg1's pose, all of full-size cfg5) the chained
    # device result must be within north_star's 1e-4 of it, flip or no flip; elsewhere the reference's own spread is the yardstick.
    bound = {q: (1e-4 if sens[q] < 1e-4 else max(1e-4 if first is None else 5e-3, P.E2E_SPREAD_FACTOR * sens[q])) for q in sens}
    for q in ("rot", "scale", "trans", "code"):
        assert m[q] <= bound[q], (q, m[q], sens[q])

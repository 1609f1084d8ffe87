"""Reference-held evidence for the HEADLINE workload: the 64 objects bench.py times at its default
configuration -- synth.make_batch(64, first_seed=1, 2000, 500), KITTI hyper-parameters -- against what the UNMODIFIED reference recorded
for exactly these objects (tests/golden/golden_bench_cfg2x64.npz, tools/make_golden_bench.py):

  all 64   final pose / code / loss / is_good and V, m, K of every iteration;
  16 of 64 full per-iteration traces (state, depth samples, H, b, dx) + the reference's own spread under eight 1-ulp input draws: the
           first and the last bench object, the three with the most render rows, the one with the fewest, the largest first step, the
           largest initial yaw error (round 4, first batch), and the eight next-largest K (second batch, --extend: their all_* entries are
           the traced run's -- the reference does not reproduce an earlier process's run bit for bit, tr<i>_rerun_dT records by how much).

The device is checked INSIDE THE RESIDENT 64-OBJECT BATCH (the thing the bench times), not on single-object batches:
  (a) at the reference's own recorded states (pose, code, depth samples injected bit for bit for the 8 traced objects, every iteration):
      identical V and K, H / b within 3e-5 / 1.2e-4 of the reference's recorded values -- or every differing sample named and within
      round-off of the threshold it crossed;
  (b) iteration 0 of ALL 64 objects from the device's own start state against the reference's recorded V and K;
  (c) all ten iterations chained, each traced object's result against that object's OWN reference spread, every object's result against
      the reference's.
"""
import json
import os

import numpy as np
import pytest

import forensics as F
from conftest import GOLDEN, golden, parity_log
from oracle import dsp_oracle as O
from dsp_slam_amd import synth, engine as E

GOLD = "golden_bench_cfg2x64.npz"


def _golden_complete():
    """tools/make_golden_bench.py checkpoints the file after every traced object; only a finished file (all ulp draws present) is used."""
    p = os.path.join(GOLDEN, GOLD)
    if not os.path.exists(p):
        return False
    try:
        z = np.load(p, allow_pickle=False)
        return all(("tr%d_ulps_code" % int(i)) in z.files for i in z["full_objects"])
    except Exception:
        return False


have_golden = _golden_complete()


def _objects(g):
    return synth.make_batch(int(g["all_it_V"].shape[0]), first_seed=int(g["first_seed"]), n_surface=int(g["n_surface"]), n_background=int(g["n_background"]))


@pytest.mark.skipif(not have_golden, reason="tests/golden/%s not generated yet (tools/make_golden_bench.py)" % GOLD)
def test_bench_objects_regenerate_bit_for_bit():
    """The golden holds no inputs: the test regenerates them from the seeds.  The digest recorded next to the reference's results proves that
    these are the arrays the reference saw (CPU tier)."""
    import hashlib
    g = golden(GOLD)
    for i, o in enumerate(_objects(g)):
        h = hashlib.sha256()
        for k in ("t_cam_obj_init", "pts", "rays", "depth"):
            h.update(np.ascontiguousarray(o[k], np.float32).tobytes())
        assert np.frombuffer(h.digest()[:8], np.uint64)[0] == g["all_input_digest"][i], i
    full = [int(i) for i in g["full_objects"]]
    assert len(full) >= 8 and 0 in full and g["all_it_V"].shape[0] - 1 in full
    assert bool(g["all_is_good"].all())
    for i in full:
        assert g["tr%d_it_H" % i].shape == (10, 71, 71) and g["tr%d_ulps_code" % i].shape[0] == 8
        assert np.array_equal(g["tr%d_it_K" % i], g["all_it_K"][i])


@pytest.fixture(scope="module")
def eng(oracle_decoder):
    e = E.Engine(oracle_decoder.layers, oracle_decoder.latent_in, oracle_decoder.code_len, device=0)
    yield e
    e.close()


@pytest.fixture(scope="module")
def batch64(eng):
    g = golden(GOLD)
    objs = _objects(g)
    prm = E.params_from_configs(json.loads(str(g["cfg_json"])))
    b = eng.batch(prm, [o["t_cam_obj_init"] for o in objs], [o["pts"] for o in objs], [o["rays"] for o in objs], [o["depth"] for o in objs], trace=True)
    yield g, objs, prm, b
    b.close()


def _rot_prior_bound(h_ref, k4):
    j_rot = np.sqrt(np.abs(np.diag(h_ref)[3:6]) / max(k4, 1.0))
    return k4 * (j_rot + 1e-3) * 2.4e-7


@pytest.mark.gpu
@pytest.mark.skipif(not have_golden, reason="golden not generated")
def test_batch64_at_the_references_recorded_states(batch64, oracle_decoder):
    g, objs, prm, b = batch64
    cfg = json.loads(str(g["cfg_json"]))
    oprm = O.GNParams.from_configs(cfg)
    k4 = cfg["optimizer"]["joint_optim"]["k4"]
    B, n_d = len(objs), oprm.num_depth_samples
    full = [int(i) for i in g["full_objects"]]
    # every object's own start state and depth samples as the device derives them (iteration 0 of a plain run): what the 56 untraced
    # objects keep being started from while the 8 traced ones walk through the reference's recorded iterations
    zero_codes = [np.zeros(64, np.float32)] * B
    b.set_start_state(None, zero_codes, None)
    b.set_iterations(1)
    b.run()
    base = b.trace(0)
    mask = np.ones(71, bool)
    mask[3:6] = False
    rows, strict, named_total, jitter_rows = [], 0, 0, []
    for e in range(10):
        t_oc = [base["t_obj_cam"][i] for i in range(B)]
        codes = [base["code"][i] for i in range(B)]
        depths = [base["depths"][i][:n_d] for i in range(B)]
        for i in full:
            t_oc[i], codes[i], depths[i] = g["tr%d_it_t_obj_cam" % i][e], g["tr%d_it_code" % i][e], g["tr%d_it_depths" % i][e]
        b.set_start_state(t_oc, codes, depths)
        b.set_iterations(1)
        b.run()
        res_e = b.results()
        assert (res_e[3] == 0).all()
        tr = b.trace(0)
        for i in full:
            assert np.array_equal(tr["t_obj_cam"][i], g["tr%d_it_t_obj_cam" % i][e]) and np.array_equal(tr["code"][i], g["tr%d_it_code" % i][e])
            assert np.array_equal(tr["depths"][i][:n_d], g["tr%d_it_depths" % i][e])
            v_ref, k_ref = int(g["tr%d_it_V" % i][e]), int(g["tr%d_it_K" % i][e])
            h_ref, b_ref, dx_ref = g["tr%d_it_H" % i][e], g["tr%d_it_b" % i][e], g["tr%d_it_dx" % i][e]
            # The 3 x 3 rotation block of H carries k4 * J_rot^T J_rot with k4 = 1e7, and the reference derives J_rot = (R_oc n_g) x e_y through
            # two float32 inversions, a float32 pow and a division (loss.py:155-178): a few ulp of 1.0 on entries of ~1e-3, times 1e7.  For a
            # tilted state that is 1e-4 ... 1e-3 of |H| of noise IN THE REFERENCE (object 54, iteration 7: the numpy oracle, which derives J_rot
            # in fp64, differs from the recorded H[3, 3] by 3.1e-4 of |H|max exactly as the device does, while its response to sdf jitter is
            # 7e-7).  So, as for b[3:6]: the block is held to the prior's own noise, the rest of H to the tight bound.
            hm = np.ones((71, 71), bool)
            hm[3:6, 3:6] = False
            hs = np.abs(h_ref).max()
            rh = float(np.abs(tr["H"][i] - h_ref)[hm].max() / hs)
            rb = F.rel_max(tr["b"][i][mask], b_ref[mask])
            j_rot = np.sqrt(np.abs(np.diag(h_ref)[3:6]) / max(k4, 1.0)) + 1e-3
            tol_rot_h = k4 * (j_rot[:, None] + j_rot[None, :]) * 1e-6 + 3e-5 * hs
            assert np.all(np.abs(tr["H"][i] - h_ref)[3:6, 3:6] <= tol_rot_h), (i, e, np.abs(tr["H"][i] - h_ref)[3:6, 3:6].max(), tol_rot_h.max())
            n_named = 0
            # `loss` after one iteration from the recorded state = the reference's loss AT that state (optimizer.py:155; tr<i>_it_loss:
            # tools/make_golden_it_loss.py --bench); at the last state it is the value the recorded run returned (all_loss)
            rl = F.loss_rel(res_e[2][i], g["tr%d_it_loss" % i][e])
            if e == 9:
                assert float(g["tr%d_it_loss" % i][e]) == float(g["all_loss"][i])
            if (int(tr["V"][i]), int(tr["K"][i])) != (v_ref, k_ref):
                o = objs[i]
                ot = F.oracle_linearisation(oracle_decoder, oprm, o["pts"], o["rays"], o["depth"], g["tr%d_it_t_obj_cam" % i][e], g["tr%d_it_code" % i][e],
                                            g["tr%d_it_depths" % i][e])
                assert (ot["V"], ot["K"]) == (v_ref, k_ref), "object %d iteration %d: the oracle does not reproduce the reference's sets at its own state" % (i, e)
                m, sdf, deds = b.debug_samples(i, o["rays"].shape[0], n_d)
                flips = F.name_flips(m, sdf, deds, F.oracle_grids(ot["sets"], o["rays"].shape[0], n_d), oprm.cut_off)
                assert flips and all(f["explained"] for f in flips) and len(flips) <= 4, (i, e, flips)
                n_named = len(flips)
                named_total += 1
                assert rl <= F.LOSS_RTOL_FLIPPED, (i, e, rl)
            else:
                strict += 1
                assert rl <= F.LOSS_RTOL, (i, e, float(res_e[2][i]), float(g["tr%d_it_loss" % i][e]))
                # Identical sample sets: H and b within 3e-5 / 1.2e-4 of the reference's recorded values (the bound of the seven
                # single-object goldens, 3x what they measure) -- or, for the linearisations whose render rows sit on the amplifying inner
                # edge of the band (de_ds ~ 1 / (1 - o) -> 1 / (2 th (1 - o)) ~ 5e3...5e4 per unit of sdf: these eight objects include the
                # ones with the most render rows, up to 22 670), within 1e-4 + twice what the ORACLE's own H / b move when its decoded sdf
                # values are jittered by the decoders' agreement of 2e-7 (compare_linearisation's bound, tests/test_gpu_parity.py).
                amp_h = amp_b = 0.0
                if rh >= 3e-5 or rb >= 1.2e-4:
                    o = objs[i]
                    args = (oracle_decoder, oprm, o["pts"], o["rays"], o["depth"], g["tr%d_it_t_obj_cam" % i][e], g["tr%d_it_code" % i][e], g["tr%d_it_depths" % i][e])
                    it0, itj = F.oracle_linearisation(*args), F.oracle_linearisation(*args, sdf_jitter=2e-7)
                    assert (it0["V"], it0["K"]) == (v_ref, k_ref)
                    # (where 2e-7 of sdf moves a sample of the ORACLE across a threshold, a whole row enters or leaves: that response is the yardstick too)
                    amp_h = float(np.abs(itj["H"] - it0["H"])[hm].max() / hs)
                    amp_b = float(np.abs(itj["b"] - it0["b"])[mask].max() / np.abs(it0["b"][mask]).max())
                    jitter_rows.append(dict(object=i, iteration=e, rel_H=rh, rel_b=rb, oracle_jitter_rel_H=amp_h, oracle_jitter_rel_b=amp_b,
                                            oracle_jitter_flips_a_sample=bool((it0["vsum"], it0["ksum"]) != (itj["vsum"], itj["ksum"]))))
                    assert rh < 1e-4 + 2 * amp_h, (i, e, rh, amp_h)
                    assert rb < 1.2e-4 + 2 * amp_b, (i, e, rb, amp_b)
                assert np.all(np.abs(tr["b"][i][3:6] - b_ref[3:6]) <= _rot_prior_bound(h_ref, k4) + 2e-4 * np.abs(b_ref).max())
                if amp_h == 0.0:        # (dx = H^-1 b inherits the sensitivity: checked where the tight bounds hold)
                    tol_b = np.full(71, 2e-4 * np.abs(b_ref[mask]).max())
                    tol_b[3:6] += _rot_prior_bound(h_ref, k4)
                    tol_dx = np.abs(np.linalg.inv(h_ref.astype(np.float64))) @ tol_b + 1e-4 * np.abs(dx_ref).max()
                    assert np.all(np.abs(tr["dx"][i] - dx_ref) <= tol_dx), (i, e)
            rows.append(dict(object=i, iteration=e, V=v_ref, K=k_ref, rel_H=rh, rel_b=rb, rel_loss=rl, named=n_named))
    b.set_start_state(None, zero_codes, None)
    parity_log(kind="bench_at_reference_states", case="64 x cfg2 bench batch, %d traced objects x 10 iterations inside the resident batch" % len(full), objects=full,
               n=len(rows), strict=strict, with_named_flips=named_total, beyond_tight_bounds=jitter_rows, max_rel_H=max(r["rel_H"] for r in rows), max_rel_b=max(r["rel_b"] for r in rows),
               max_rel_loss=max(r["rel_loss"] for r in rows), n_loss_comparisons=len(rows),
               per_object={str(i): dict(max_rel_H=max(r["rel_H"] for r in rows if r["object"] == i), max_rel_b=max(r["rel_b"] for r in rows if r["object"] == i),
                                        max_rel_loss=max(r["rel_loss"] for r in rows if r["object"] == i),
                                        K=[r["K"] for r in rows if r["object"] == i], named=sum(r["named"] > 0 for r in rows if r["object"] == i)) for i in full})
    assert strict >= len(rows) - 2, "more than two of %d linearisations with (named) flips at the reference's own states" % len(rows)


@pytest.mark.gpu
@pytest.mark.skipif(not have_golden, reason="golden not generated")
def test_batch64_first_iteration_and_chained_result_vs_reference(batch64):
    """(b) + (c): all 64 objects, from the device's own start state.  The device inverts the initial pose in fp64 and rounds (the reference:
    float32 LAPACK), so a sample within round-off of the unit sphere or of a threshold may fall on the other side: V and K within 2 of the
    reference per object, identical for at least 56 of the 64 (measured on MI355X: identical for 61, the other three differ by 1, 2 and
    1 in-sphere samples out of ~90 000).  Chained, an object's result is held to ITS OWN reference spread where that was recorded (1.5 x
    the largest of eight 1-ulp draws, as tests/test_gpu_parity.py does); for the 56 objects without recorded draws the yardstick is 3 x
    the largest spread recorded among the eight -- a sanity bound: the reference's own spread varies by two orders of magnitude between
    these objects (2e-4 ... 2e-2 in rotation), every measured value goes to the parity report."""
    import test_gpu_parity as P
    g, objs, prm, b = batch64
    B = len(objs)
    b.set_start_state(None, [np.zeros(64, np.float32)] * B, None)      # the uploaded initial estimates, zero codes (optimizer.py:96-99)
    b.set_iterations(10)
    b.run()
    t, code, loss, status = b.results()
    assert (status == 0).all() and bool(g["all_is_good"].all())
    tr0, tr9 = b.trace(0), b.trace(9)
    dv = np.abs(tr0["V"] - g["all_it_V"][:, 0])
    dk = np.abs(tr0["K"] - g["all_it_K"][:, 0])
    exact0 = int(((dv == 0) & (dk == 0)).sum())
    assert dv.max() <= 2 and dk.max() <= 2, (dv.max(), dk.max())
    full = [int(i) for i in g["full_objects"]]
    per, worst_spread = {}, dict(rot=0.0, scale=0.0, trans=0.0, code=0.0)
    for i in full:
        gi = dict(t_cam_obj=g["all_t_cam_obj"][i], code=g["all_code"][i], ulp_t_cam_obj=g["tr%d_ulps_t_cam_obj" % i][0], ulp_code=g["tr%d_ulps_code" % i][0],
                  ulps_t_cam_obj=g["tr%d_ulps_t_cam_obj" % i], ulps_code=g["tr%d_ulps_code" % i])
        if "tr%d_thr_t_cam_obj" % i in g.files:        # the reference's no-input-change (thread count) spread of this object, tools/make_golden_threads.py --bench
            gi.update(thr_t_cam_obj=g["tr%d_thr_t_cam_obj" % i], thr_code=g["tr%d_thr_code" % i])
        m, sens, _ = P.end_to_end_differences(gi, t[i], code[i])
        per[str(i)] = dict(measured={q: m[q] for q in ("rot", "scale", "trans", "code")}, reference_spread={q: sens[q] for q in ("rot", "scale", "trans", "code")})
        for q in worst_spread:
            worst_spread[q] = max(worst_spread[q], sens[q])
            assert m[q] <= max(1e-4, P.E2E_SPREAD_FACTOR * sens[q]), (i, q, m[q], sens[q])
    worst = dict(rot=0.0, scale=0.0, trans=0.0, code=0.0)
    for i in range(B):
        gi = dict(t_cam_obj=g["all_t_cam_obj"][i], code=g["all_code"][i], ulp_t_cam_obj=g["all_t_cam_obj"][i], ulp_code=g["all_code"][i],
                  ulps_t_cam_obj=g["all_t_cam_obj"][i][None], ulps_code=g["all_code"][i][None])
        m, _, _ = P.end_to_end_differences(gi, t[i], code[i])
        for q in worst:
            worst[q] = max(worst[q], m[q])
            assert m[q] <= max(1e-4, 3.0 * worst_spread[q]), (i, q, m[q], worst_spread[q])
    k_same9 = int((tr9["K"] == g["all_it_K"][:, 9]).sum())
    parity_log(kind="bench_chained", case="64 x cfg2 bench batch, all objects vs the reference", objects_identical_sets_iteration0=exact0, n_objects=B,
               max_dV_iteration0=int(dv.max()), max_dK_iteration0=int(dk.max()), objects_same_K_iteration9=k_same9, traced=per,
               worst_over_all_objects=worst, largest_recorded_reference_spread=worst_spread,
               loss_rel_max=float(np.max(np.abs(loss - g["all_loss"]) / np.maximum(np.abs(g["all_loss"]), 1e-12))))
    assert exact0 >= B - 8, "iteration 0: %d of %d objects select exactly the reference's sets" % (exact0, B)

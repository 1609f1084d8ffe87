"""GPU: the BASELINE.json configurations that are parity cases rather than bench lines (cfg2 full size, cfg3 = 64-object
batch, cfg4 = objects sharded over ranks + gather, cfg5 = 4000-point objects, Redwood hyper-parameters, second decoder,
mixed batch) plus size-independent properties at full size: run-to-run determinism, batch-permutation equivariance,
shard/gather == unsharded."""
import json
import os

import numpy as np
import pytest

from conftest import golden, parity_log
from oracle import dsp_oracle as O
from dsp_slam_amd import fixtures, synth, engine as E, distributed as D
from test_gpu_parity import compare_linearisation, one_iteration_oracle, LAST_LINEARISATION

pytestmark = pytest.mark.gpu

REDWOOD = dict(k1=10.0, k2=100.0, k3=2.5, k4=0.0, b1=0.2, b2=0.02, lr=1.0, s_damp=100.0, num_iterations=5)


@pytest.fixture(scope="module")
def eng(oracle_decoder):
    e = E.Engine(oracle_decoder.layers, oracle_decoder.latent_in, oracle_decoder.code_len, device=0)
    yield e
    e.close()


def chairs_like_layers(dec):
    """A second, different decoder without another 3 MB fixture: the cars decoder composed with a 90-degree roll about z,
    f'(x, y, z) = f(-y, x, z), applied to the xyz columns of the two layers that see xyz."""
    layers = [(w.copy(), b.copy()) for w, b in dec.layers]
    for k in (0, 4):
        w = layers[k][0]
        wx, wy = w[:, -3].copy(), w[:, -2].copy()
        w[:, -3] = wy            # coefficient of x' := old coefficient of y
        w[:, -2] = -wx           # coefficient of y' := -(old coefficient of x)
    return layers


def _run(eng, prm, objs, codes=None):
    return eng.reconstruct_batch(prm, [o["t_cam_obj_init"] for o in objs], [o["pts"] for o in objs],
                                 [o["rays"] for o in objs], [o["depth"] for o in objs], codes)


def test_cfg2_full_size_first_iteration_vs_reference(eng, oracle_decoder):
    """cfg2 (2000 + 500 rays x 50): the first linearisation, from the device's own start state, against the reference's golden trace,
    then three later iterations against the oracle restarted from the GPU's state.  (ALL ten iterations are covered, at the reference's
    own recorded states and at the device's, by tests/test_gpu_forensics.py.)"""
    g = golden("golden_recon_cfg2.npz")
    cfg = json.loads(str(g["cfg_json"]))
    prm = E.params_from_configs(cfg)
    b = eng.batch(prm, [g["in_t_cam_obj_init"]], [g["in_pts"]], [g["in_rays"]], [g["in_depth"]], trace=True)
    b.run()
    tr0 = b.trace(0)
    assert abs(int(tr0["V"][0]) - int(g["it_V"][0])) <= 1      # own start state (fp64 inverse of t_cam_obj): a sample on the unit sphere may switch sides
    dk = abs(int(tr0["K"][0]) - int(g["it_K"][0]))
    assert dk <= 2, "more than two threshold flips against the reference"
    # identical ragged sets -> 1e-4; one sample within round-off of a threshold changes H, b by O(1/K)
    dv = abs(int(tr0["V"][0]) - int(g["it_V"][0]))
    tol = 1e-4 if (dk == 0 and dv == 0) else 4.0 * max(dk, dv) / float(g["it_K"][0])
    assert np.abs(tr0["H"][0] - g["it_H"][0]).max() < tol * np.abs(g["it_H"][0]).max()
    mask = np.ones(71, bool)
    mask[3:6] = False
    assert np.abs(tr0["b"][0][mask] - g["it_b"][0][mask]).max() < tol * np.abs(g["it_b"][0]).max()
    oprm = O.GNParams.from_configs(cfg)
    strict, per_iter = 0, []
    for e in (3, 6, 9):
        tr = b.trace(e)
        obj = dict(pts=g["in_pts"], rays=g["in_rays"], depth=g["in_depth"])
        strict += bool(compare_linearisation(tr, 0, one_iteration_oracle(oracle_decoder, oprm, obj, tr), oprm.k4))
        per_iter.append(dict(LAST_LINEARISATION, iteration=e))
    parity_log(kind="iterations", case="cfg2 single object, iterations 3/6/9 vs oracle", n=3, strict=strict,
               same_sets=sum(1 for p in per_iter if p["same_sets"]), flips=[p["flips"] for p in per_iter], rel_H=[p["rel_H"] for p in per_iter],
               rel_b=[p["rel_b"] for p in per_iter], oracle_jitter_rel_H=[p["oracle_jitter_rel_H"] for p in per_iter], K=[p["K"] for p in per_iter],
               first_iteration_vs_reference=dict(dK=int(dk), rel_H=float(np.abs(tr0["H"][0] - g["it_H"][0]).max() / np.abs(g["it_H"][0]).max())))
    assert strict == 3          # measured on MI355X (profiles/parity_r03.md): 3 of 3 strict
    b.close()


def test_cfg3_batch64_objects_vs_oracle(eng, oracle_decoder):
    """Inside a full 64-object cfg2 batch (the bench workload) the first and the last object, at the first and the last GN
    iteration, are each one linearisation the oracle reproduces from the device's own state (optimizer.py:118-192) --
    i.e. parity holds at full batch size, not only for single objects."""
    prm = E.gn_params()
    oprm = O.GNParams()
    objs = synth.make_batch(64, first_seed=300, n_surface=2000, n_background=500)
    b = eng.batch(prm, [o["t_cam_obj_init"] for o in objs], [o["pts"] for o in objs], [o["rays"] for o in objs],
                  [o["depth"] for o in objs], trace=True)
    b.run()
    assert (b.results()[3] == 0).all()
    out = []
    for e in (0, prm.num_iterations - 1):
        tr = b.trace(e)
        for i in (0, 63):
            strict = bool(compare_linearisation(tr, i, one_iteration_oracle(oracle_decoder, oprm, objs[i], tr, i), oprm.k4))
            out.append(dict(LAST_LINEARISATION, iteration=e, object=i, strict=strict))
    b.close()
    parity_log(kind="batch64", case="64 x cfg2 batch: objects 0 and 63, iterations 0 and 9 vs oracle", checks=out)
    assert sum(1 for o in out if o["same_sets"]) == 4        # measured on MI355X: identical sets in 4 of 4 (profiles/parity_r03.md)


def test_cfg3_batch64_deterministic_and_permutation_equivariant(eng):
    prm = E.gn_params()
    objs = synth.make_batch(64, first_seed=300, n_surface=2000, n_background=500)
    t1, c1, l1, s1 = _run(eng, prm, objs)
    t2, c2, l2, s2 = _run(eng, prm, objs)
    assert np.array_equal(t1, t2) and np.array_equal(c1, c2) and np.array_equal(l1, l2) and np.array_equal(s1, s2)
    assert (s1 == 0).all() and np.isfinite(t1).all() and np.isfinite(c1).all()
    perm = np.random.default_rng(0).permutation(64)
    tp, cp, lp, sp = _run(eng, prm, [objs[i] for i in perm])
    assert np.array_equal(tp, t1[perm]) and np.array_equal(cp, c1[perm]) and np.array_equal(lp, l1[perm])
    # objects converge towards their ground truth on average (the optimiser does its job at this size)
    err0 = np.mean([np.linalg.norm(o["t_cam_obj_init"][:3, 3] - o["t_cam_obj_gt"][:3, 3]) for o in objs])
    err1 = np.mean([np.linalg.norm(t1[i][:3, 3] - o["t_cam_obj_gt"][:3, 3]) for i, o in enumerate(objs)])
    assert err1 < err0


def test_cfg4_sharded_equals_unsharded(eng):
    """1024-object job, scaled to what one GPU does in seconds (96 small objects): block-sharded over 8 'ranks' run one
    after the other, packed and concatenated exactly as gather_results does == one unsharded run."""
    prm = E.gn_params(num_iterations=4)
    objs = [synth.make_object(500 + i, n_surface=100 + 7 * (i % 9), n_background=25 + (i % 4)) for i in range(96)]
    full = D.pack_results(*_run(eng, prm, objs))
    shards = D.shard_objects([D.object_cost(o["pts"].shape[0], o["rays"].shape[0]) for o in objs], 8)
    parts = [D.pack_results(*_run(eng, prm, objs[a:b])) for a, b in shards if b > a]
    assert np.array_equal(np.concatenate(parts, 0), full)
    assert max(b - a for a, b in shards) - min(b - a for a, b in shards) <= 3     # balanced


def test_more_than_256_objects_in_one_batch(eng):
    """300 tiny ragged objects in ONE batch (the tile-list builder walks objects 256 at a time; per-object kernels use the grid's
    y / x dimension) == the same objects in three batches of 100, bit for bit, including two objects that fail."""
    prm = E.gn_params(num_iterations=3)
    objs = [synth.make_object(3000 + i, n_surface=40 + 5 * (i % 7), n_background=12 + (i % 5)) for i in range(300)]
    objs[17] = dict(objs[17], rays=np.zeros((0, 3), np.float32), depth=np.zeros((0,), np.float32))    # no samples: status 1
    objs[290] = dict(objs[290], pts=np.zeros((0, 3), np.float32))                                      # no surface points: status 2
    full = _run(eng, prm, objs)
    parts = [_run(eng, prm, objs[a:a + 100]) for a in (0, 100, 200)]
    for k in range(4):
        assert np.array_equal(np.concatenate([p[k] for p in parts], 0), full[k])
    assert full[3][17] == 1 and full[3][290] == 2 and (full[3] == 0).sum() >= 290


def test_cfg5_redwood_4000_points_two_decoders_mixed_batch(eng, oracle_decoder):
    """Redwood hyper-parameters, 4000 surface points + render term, a second decoder resident on the same GPU, and a
    mixed batch: car objects on the car handle, 'chair' objects on the chair handle; each handle's results equal the
    oracle's linearisation with the matching decoder, and the two handles do not disturb each other."""
    chairs = chairs_like_layers(oracle_decoder)
    dec_ch = O.FoldedDecoder(chairs, oracle_decoder.latent_in, oracle_decoder.code_len)
    eng_ch = E.Engine(chairs, oracle_decoder.latent_in, oracle_decoder.code_len, device=0)
    prm = E.gn_params(**REDWOOD)
    cars = synth.make_batch(3, first_seed=700, n_surface=4000, n_background=500)
    chairs_objs = []
    for o in synth.make_batch(3, first_seed=800, n_surface=4000, n_background=500):
        # present the same world points to the rolled decoder: p_old = (-y', x', z')  =>  T' = T @ P, det(P) = +1
        o = dict(o)
        p = np.array([[0, -1, 0, 0], [1, 0, 0, 0], [0, 0, 1, 0], [0, 0, 0, 1]], np.float32)
        o["t_cam_obj_init"] = (o["t_cam_obj_init"] @ p).astype(np.float32)
        chairs_objs.append(o)
    bc = eng.batch(prm, [o["t_cam_obj_init"] for o in cars], [o["pts"] for o in cars], [o["rays"] for o in cars],
                   [o["depth"] for o in cars], trace=True)
    bh = eng_ch.batch(prm, [o["t_cam_obj_init"] for o in chairs_objs], [o["pts"] for o in chairs_objs],
                      [o["rays"] for o in chairs_objs], [o["depth"] for o in chairs_objs], trace=True)
    bc.run(); bh.run(); bc.run()                # interleaved use of the two handles
    rc, rh = bc.results(), bh.results()
    assert (rc[3] == 0).all() and (rh[3] == 0).all()
    for batch, dec, objs in ((bc, oracle_decoder, cars), (bh, dec_ch, chairs_objs)):
        tr = batch.trace(0)
        oprm = O.GNParams(k1=10.0, k2=100.0, k3=2.5, k4=0.0, b1=0.2, b2=0.02, lr=1.0, s_damp=100.0, num_iterations=1)
        for i in (0, 2):
            compare_linearisation(tr, i, one_iteration_oracle(dec, oprm, objs[i], tr, i), 0.0)
    # the car results are what the car handle gives alone
    alone = _run(eng, prm, cars)
    assert np.array_equal(alone[0], rc[0]) and np.array_equal(alone[1], rc[1])
    bc.close(); bh.close(); eng_ch.close()


def test_pose_only_batch_ragged(eng, oracle_decoder):
    prm = E.gn_params()
    oprm = O.GNParams()
    objs = [synth.make_object(900 + i, n_surface=150 + 50 * i, n_background=0) for i in range(4)]
    t_se3, scales, codes = [], [], []
    for o in objs:
        s = float(o["scale"])
        t = o["t_cam_obj_init"].copy()
        t[:3, :3] /= s
        t_se3.append(t); scales.append(s)
        c = np.zeros(64, np.float32); c[:3] = o["code_gt"][:3]
        codes.append(c)
    out = eng.estimate_pose_batch(prm, t_se3, scales, [o["pts"] for o in objs], codes)
    for i, o in enumerate(objs):
        ref = O.estimate_pose_cam_obj(oracle_decoder, oprm, t_se3[i], scales[i], o["pts"], codes[i])
        assert np.abs(out[i] - ref).max() < 1e-4 * np.abs(ref).max()
        one = eng.estimate_pose_batch(prm, [t_se3[i]], [scales[i]], [o["pts"]], [codes[i]])
        assert np.array_equal(one[0], out[i])


def test_multi_code_grid_decode_equals_single(eng, oracle_decoder):
    """MeshExtractor-style grid decode for a whole map in one launch == one launch per object, bit for bit, and == oracle."""
    rng = np.random.default_rng(3)
    codes = (rng.normal(size=(5, 64)) * 0.2).astype(np.float32)
    n = 17
    lin = np.linspace(-1, 1, n, dtype=np.float32)
    grid = np.stack(np.meshgrid(lin, lin, lin, indexing="ij"), -1).reshape(-1, 3)
    multi = eng.decode_sdf_multi(codes, grid)
    assert multi.shape == (5, n ** 3)
    for c in range(5):
        assert np.array_equal(multi[c], eng.decode_sdf(codes[c], grid))
    assert np.abs(multi[2] - O.decode_sdf(oracle_decoder, codes[2], grid)).max() < 5e-6


def test_degenerate_objects_do_not_poison_the_batch(eng):
    """Ragged edge cases inside one batch: an object with no rays at all, one with no surface points, one with a single
    ray and point, one with rays but no foreground depths -- each ends with a failure status (or runs), none hangs or
    disturbs the healthy object next to it."""
    prm = E.gn_params(num_iterations=3)
    good = synth.make_object(600, n_surface=90, n_background=20)
    no_rays = dict(good, rays=np.zeros((0, 3), np.float32), depth=np.zeros((0,), np.float32))
    no_pts = dict(good, pts=np.zeros((0, 3), np.float32))
    single = dict(good, pts=good["pts"][:1], rays=good["rays"][:1], depth=good["depth"][:1])
    all_bg = dict(good, depth=np.zeros((0,), np.float32))
    objs = [no_rays, good, no_pts, single, all_bg]
    t, c, l, s = _run(eng, prm, objs)
    assert s[0] == 1            # < 10 in-sphere samples  -> compute_render_loss returns None in the reference
    assert s[1] == 0
    assert s[2] == 2            # empty surface set -> NaN loss in the reference
    assert s[3] in (0, 1, 2)    # one ray through the object: it runs, or stops for lack of samples / kept rows
    assert s[4] in (0, 2)
    alone = _run(eng, prm, [good])
    assert np.array_equal(alone[0][0], t[1]) and np.array_equal(alone[1][0], c[1])
    assert np.isfinite(t[1]).all()


def test_early_ray_termination_is_exact(eng):
    """Front-to-back ray passes skip decoder evaluations behind the first solid sample of a ray (transmittance exactly 0).
    Every pass count gives bit-identical results to decoding all in-sphere samples (1 pass), and it does skip work."""
    prm = E.gn_params(num_iterations=4)
    objs = synth.make_batch(3, first_seed=950, n_surface=400, n_background=120)
    args = ([o["t_cam_obj_init"] for o in objs], [o["pts"] for o in objs], [o["rays"] for o in objs], [o["depth"] for o in objs])
    ref = None
    evaluated = {}
    for n_passes in (1, 0, 2, 5, 10, 50):      # 0 = automatic: per-ray ranges steered by the previous iteration's hits
        b = eng.batch(prm, *args, trace=True)
        b.set_prepass(0)              # this test is about the fp32 passes (tests/test_gpu_prepass.py covers the prepass)
        b.set_ray_passes(n_passes)
        b.run()
        res = b.results()
        tr = [b.trace(e) for e in range(4)]
        st = b.stats()
        evaluated[n_passes] = st["n_fwd_points"]
        b.close()
        if ref is None:
            ref = (res, tr)
            assert st["n_fwd_points"] == st["n_insphere_points"]
        else:
            for a, c in zip(res, ref[0]):
                assert np.array_equal(a, c)
            for ta, tc in zip(tr, ref[1]):
                for k in ("H", "b", "dx", "V", "K", "set_sums"):
                    assert np.array_equal(ta[k], tc[k]), k
            assert st["n_insphere_points"] == ref_v
        ref_v = st["n_insphere_points"]
    assert evaluated[50] < evaluated[5] < evaluated[2] < evaluated[1]
    assert evaluated[5] < 0.85 * evaluated[1] and evaluated[0] < 0.85 * evaluated[1]


def test_mask_reuse_is_exact(eng):
    """Render rows either share the surface points' forward+backward launch or run backward-only from the relu masks the
    forward launches exported: every bit of every iteration (H, b, dx, sets, results) must agree, and the backward-only
    launch must actually have run when asked for."""
    prm = E.gn_params(num_iterations=4)
    objs = synth.make_batch(3, first_seed=960, n_surface=500, n_background=150)
    args = ([o["t_cam_obj_init"] for o in objs], [o["pts"] for o in objs], [o["rays"] for o in objs], [o["depth"] for o in objs])
    out = {}
    for mode in (0, 1):
        for n_passes in (0, 10):
            b = eng.batch(prm, *args, trace=True)
            b.set_mask_reuse(mode)
            b.set_ray_passes(n_passes)
            b.run()
            out[mode, n_passes] = (b.results(), [b.trace(e) for e in range(4)], b.stats())
            b.close()
    ref = out[0, 0]
    assert ref[2]["n_render_rows"] == 0 and out[1, 0][2]["n_render_rows"] > 0
    assert out[1, 0][2]["n_mlp_jac_launches"] == 2 * ref[2]["n_mlp_jac_launches"]
    assert out[1, 0][2]["n_jac_points"] + out[1, 0][2]["n_render_rows"] == ref[2]["n_jac_points"]
    for key, (res, tr, st) in out.items():
        for a, c in zip(res, ref[0]):
            assert np.array_equal(a, c), key
        for ta, tc in zip(tr, ref[1]):
            for k in ("H", "b", "dx", "V", "K", "set_sums"):
                assert np.array_equal(ta[k], tc[k]), (key, k)


def test_split_kernel_is_exact(eng, oracle_decoder):
    """The latency form of the jacobian launch (16-point tiles, every layer's rows split over the four waves, per-wave weight
    streams) must reproduce the throughput form bit for bit -- joint optimisation traces, results, and the pose-only path --
    for ragged tile counts (tiles with 1..16 valid points)."""
    prm = E.gn_params(num_iterations=3)
    objs = [synth.make_object(970, n_surface=333, n_background=90), synth.make_object(971, n_surface=17, n_background=40),
            synth.make_object(972, n_surface=512, n_background=0)]
    args = ([o["t_cam_obj_init"] for o in objs], [o["pts"] for o in objs], [o["rays"] for o in objs], [o["depth"] for o in objs])
    out = {}
    for split in (0, 1):
        b = eng.batch(prm, *args, trace=True)
        b.set_mask_reuse(0)
        b.set_split_rows(split)
        b.run()
        out[split] = (b.results(), [b.trace(e) for e in range(3)])
        b.close()
    for a, c in zip(out[1][0], out[0][0]):
        assert np.array_equal(a, c)
    for ta, tc in zip(out[1][1], out[0][1]):
        for k in ("H", "b", "dx", "V", "K", "set_sums"):
            assert np.array_equal(ta[k], tc[k]), k
    assert (out[0][0][3] == 0).all()
    # pose-only batches take the latency form automatically; compare with the golden pose-only result and with a decode
    gp = golden("golden_pose_only.npz")
    t = eng.estimate_pose_batch(E.gn_params(), [gp["t_co_se3"]], [float(gp["scale"])], [gp["pts"]], [gp["code"]])
    assert np.abs(t[0] - gp["out"]).max() / np.abs(gp["out"]).max() < 1e-4


def test_tail_split_is_exact(eng):
    """The last, mostly empty round of a 64-point forward launch handed to the latency-form kernel as 16-point tiles (k_tail_tiles):
    one cfg2-size object (a few hundred band tiles on 256 CUs: one full round + a remainder) and a two-object batch, with the prepass on
    and off -- every bit of every iteration equals the run without the split, and the forward launches get shorter."""
    prm = E.gn_params(num_iterations=5)
    for objs in ([synth.make_object(1, n_surface=2000, n_background=500)], synth.make_batch(2, first_seed=985, n_surface=1500, n_background=400)):
        args = ([o["t_cam_obj_init"] for o in objs], [o["pts"] for o in objs], [o["rays"] for o in objs], [o["depth"] for o in objs])
        for prepass in (1, 0):
            out = {}
            for tail in (0, 1):
                b = eng.batch(prm, *args, trace=True)
                b.set_prepass(prepass)
                b.set_mask_reuse(0)
                b.set_split_rows(-1)
                b.set_tail_split(tail)
                b.run()
                b.run()
                out[tail] = (b.results(), [b.trace(e) for e in range(5)], b.stats())
                b.close()
            for a, c in zip(out[1][0], out[0][0]):
                assert np.array_equal(a, c)
            for ta, tc in zip(out[1][1], out[0][1]):
                for k in ("H", "b", "dx", "V", "K", "set_sums"):
                    assert np.array_equal(ta[k], tc[k]), k
            assert out[1][2]["n_fwd_points"] == out[0][2]["n_fwd_points"]
            print("tail split, %d object(s), prepass %d: fp32 forward launches %.3f ms -> %.3f ms per run" % (
                len(objs), prepass, out[0][2]["ms_mlp_fwd"], out[1][2]["ms_mlp_fwd"]))


def test_speculative_band_rows_are_exact(eng, oracle_decoder):
    """Latency path: the samples the prepass could not classify go straight into the jacobian launch (forward + backward, sdf
    scattered back) and the Gram kernel picks the kept rows' gradients up where that launch left them -- every bit of every
    iteration equals the path with a separate forward launch, the stand-alone render term included."""
    prm = E.gn_params(num_iterations=5)
    objs = synth.make_batch(3, first_seed=990, n_surface=250, n_background=200)
    args = ([o["t_cam_obj_init"] for o in objs], [o["pts"] for o in objs], [o["rays"] for o in objs], [o["depth"] for o in objs])
    out = {}
    for spec in (0, 1):
        b = eng.batch(prm, *args, trace=True)
        b.set_prepass(1)
        b.set_speculative_band(spec)
        b.set_prepass_audit(True)
        b.run()
        out[spec] = (b.results(), [b.trace(e) for e in range(5)], b.stats())
        b.close()
    assert (out[0][0][3] == 0).all()
    for a, c in zip(out[0][0], out[1][0]):
        assert np.array_equal(a, c)
    for ta, tc in zip(out[0][1], out[1][1]):
        for k in ("H", "b", "dx", "V", "m", "K", "set_sums"):
            assert np.array_equal(ta[k], tc[k]), k
    assert out[0][2]["n_fwd_points"] > 0 and out[1][2]["n_fwd_points"] == 0          # no forward launch of their own
    assert out[1][2]["n_mlp_fwd_launches"] == 0 and out[1][2]["n_mlp_jac_launches"] == 5
    # (without the speculative rows the kept render rows either repeat their forward sweep inside the jacobian launch -- counted in
    # n_jac_points -- or, with the mixed form of mask reuse the latency path now picks, run backward-only -- counted in n_render_rows)
    assert out[1][2]["n_jac_points"] == (out[0][2]["n_jac_points"] + out[0][2]["n_render_rows"] - sum(int(t["K"].sum()) for t in out[0][1])
                                         + out[0][2]["n_fwd_points"])
    assert out[1][2]["prepass_misclassified"] == 0
    # automatic mode picks it for a detection of this size
    b = eng.batch(prm, *[a[:1] for a in args])
    b.run()
    assert b.stats()["n_mlp_fwd_launches"] == 0
    b.close()
    # stand-alone render term (one-object batch inside the library takes the same path) against the oracle's rows
    g = golden("golden_terms.npz")
    res, st = eng.compute_render_loss(g["rays"], g["depth_obs"], g["t_obj_cam"], g["sampled"], g["code"], th=0.01)
    assert res[0].shape == g["ren_j7"].shape
    assert np.abs(res[2] - g["ren_r"]).max() < 2e-5


def test_c_abi_gather_over_rccl(eng, oracle_decoder):
    """dsp_pack_results / dsp_gather_results: the multi-GPU gather of the C ABI (one process, one handle per GPU, ONE ncclGather).
    This box has one GPU, so the communicator has one rank: the block goes up to the device, through RCCL and back unchanged;
    two handles on the same GPU are refused."""
    prm = E.gn_params(num_iterations=2)
    objs = synth.make_batch(3, first_seed=40, n_surface=120, n_background=30)
    res = _run(eng, prm, objs)
    packed = E.pack_results_c(*res)
    assert np.array_equal(packed, D.pack_results(*res))                       # same layout as the Python mirror's
    got = E.gather_results_c([eng], [packed])
    assert np.array_equal(got, packed)
    assert np.array_equal(E.gather_results_c([eng], [packed[:0]]), packed[:0])   # an empty shard still joins the collective
    eng2 = E.Engine(oracle_decoder.layers, oracle_decoder.latent_in, oracle_decoder.code_len, device=0)
    with pytest.raises(Exception):
        E.gather_results_c([eng, eng2], [packed, packed])
    with pytest.raises(Exception):
        E.gather_results_c([eng, eng], [packed, packed])          # the same handle twice
    eng2.close()
    # ... and straight from a device-resident batch (no host bounce in front of the collective): dsp_gather_batch_results
    bt = eng.batch(prm, [o["t_cam_obj_init"] for o in objs], [o["pts"] for o in objs], [o["rays"] for o in objs], [o["depth"] for o in objs])
    with pytest.raises(Exception):
        E.gather_batches_c([bt])                                  # not run yet
    bt.run()
    assert np.array_equal(E.gather_batches_c([bt]), D.pack_results(*bt.results()))
    bt.close()


def test_device_resident_gather_through_torch_distributed(eng):
    """distributed.gather_results_device (what bench.py's step does under torch.distributed.run): the library copies each batch's packed
    rows device-to-device into a torch tensor, ONE dist.gather over RCCL, one copy to the host.  One rank here (one GPU per box); the
    2- and 3-rank logic incl. an empty shard runs on gloo in tests/test_sharding_gloo.py."""
    import os
    import torch
    import torch.distributed as dist
    prm = E.gn_params(num_iterations=2)
    objs = synth.make_batch(5, first_seed=60, n_surface=100, n_background=25)
    mk = lambda ol: eng.batch(prm, [o["t_cam_obj_init"] for o in ol], [o["pts"] for o in ol], [o["rays"] for o in ol], [o["depth"] for o in ol])  # noqa: E731
    b1, b2 = mk(objs[:2]), mk(objs[2:])
    b1.run(); b2.run()
    want = np.concatenate([D.pack_results(*b1.results()), D.pack_results(*b2.results())], 0)
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29547")
    own = not dist.is_initialized()
    if own:
        dist.init_process_group(backend="nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        got = D.gather_results_device([b1, b2], [(0, 5)], dist, torch.device("cuda", 0))
    finally:
        if own:
            dist.destroy_process_group()
    assert np.array_equal(got, want)
    b1.close(); b2.close()


def test_build_info_of_the_library_that_ran(eng):
    """Which compiler produced the code object these tests just executed, and which HIP runtime executed it: the decoder kernels rely on
    codegen properties checked at build time (dsp_slam_amd/build.py: check_isa), so a library rebuilt by a different hipcc on the GPU box
    must be visible in the test record, not silent."""
    import ctypes as C
    from dsp_slam_amd import _lib as L
    lib = L.load()
    info = lib.dsp_build_info().decode()
    rt, drv = C.c_int32(0), C.c_int32(0)
    assert lib.dsp_runtime_versions(C.byref(rt), C.byref(drv)) == 0
    assert "ISA assumptions checked at build time" in info and "clang" in info and rt.value > 0
    assert lib.dsp_device_count() >= 1
    print("\n%s\nHIP runtime %d, driver %d, library %s" % (info, rt.value, drv.value, L.lib_path()))
    parity_log(kind="build_info", case="libdspgn", info=info, hip_runtime=rt.value, hip_driver=drv.value, lib=os.path.basename(L.lib_path()))

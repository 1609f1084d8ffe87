"""GPU: the LOW-PRECISION COMPUTE MODE (dsp_batch_set_compute(DSP_COMPUTE_F16 | _BF16), include/dsp_gn.h) -- opt-in, NOT the parity path.

BASELINE.json's north_star names "fp32/bf16 GEMMs"; SURVEY.md 8(d) allows a 16-bit path "reported separately, never mixed into the fp32
fraction".  Here it is measured, not trusted: everything below reports HOW FAR the mode's numbers sit from the fp32 path and from the
reference, next to the reference's own spread, and asserts only what the arithmetic guarantees.

  * kernel level: mlp_lpj_fwd_kernel's sdf IS the prepass kernel's (same passes: bit for bit); its relu masks + mlp_lpj_bwd_kernel's
    transposed sweep give d sdf / d [code, xyz] within f16 rounding of the fp32 kernel's (median ~4e-4 of the largest entry; a point whose
    16-bit forward flips a relu unit that sits at zero differs by that unit's contribution); the numpy emulator (tests/lp_emulator.py) agrees
    with the device to fp32 summation order; ragged tile sizes, three decoders (cars 64-D, chairs 32-D, the complex 64-D one), f16 and bf16;
  * Gauss-Newton level: at the reference's own recorded states (the forensic goldens) the mode's V, K, H, b, loss against the reference's --
    the accuracy table of profiles/r06_lp_compute.md; chained runs land inside a small multiple of the reference's own spread; results are
    run-to-run deterministic; the default (fp32) path is untouched by a batch that ran in the mode before.
"""
import json
import os

import numpy as np
import pytest

import forensics as F
from conftest import GOLDEN, ROOT, golden, have_complex_fixture, parity_log
from oracle import dsp_oracle as O
from dsp_slam_amd import engine as E, synth, _lib as L, fixtures

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def eng(oracle_decoder):
    e = E.Engine(oracle_decoder.layers, oracle_decoder.latent_in, oracle_decoder.code_len, device=0)
    yield e
    e.close()


def _grad_err(g, g_ref):
    per_point = np.abs(g - g_ref).max(1) / np.abs(g_ref).max()
    return float(np.median(per_point)), float(np.percentile(per_point, 99)), float(per_point.max())


@pytest.mark.parametrize("n", [1, 31, 128, 129, 1000, 16387])
def test_lp_jacobian_kernels_vs_fp32(eng, oracle_decoder, n):
    rng = np.random.default_rng(n)
    code = (rng.normal(size=64) * 0.1).astype(np.float32)
    code[:3] = [0.2, -0.3, 0.1]
    pts = rng.uniform(-0.7, 0.7, size=(n, 3)).astype(np.float32)
    y32, g32 = eng.sdf_jacobian(code, pts)
    for dtype, tol in ((L.COMPUTE_F16, 1.0), (L.COMPUTE_BF16, 16.0)):
        y, g = eng.sdf_jacobian_lp(code, pts, dtype)
        # the forward half is the prepass kernel's arithmetic, pass for pass (that kernel only clamps an exact 1.0 away)
        yp = eng.decode_sdf_prepass(code, pts, dtype)
        assert np.array_equal(y[y < 1.0], yp[y < 1.0]), "the 16-bit jacobian kernel's sdf differs from the prepass kernel's"
        assert np.abs(y - y32).max() < 2e-3 * tol
        med, p99, worst = _grad_err(g, g32)
        if n >= 1000:
            print("%s jacobian, %d points: |dg| / max|g| median %.2e, 99 %% %.2e, max %.2e; |dsdf| max %.2e" % (
                "f16" if dtype == L.COMPUTE_F16 else "bf16", n, med, p99, worst, np.abs(y - y32).max()))
            parity_log(kind="lp_jacobian_kernel", case="%s, cars decoder, %d unit-ball points" % ("f16" if dtype == L.COMPUTE_F16 else "bf16", n),
                       grad_rel_median=med, grad_rel_p99=p99, grad_rel_max=worst, sdf_abs_max=float(np.abs(y - y32).max()))
        assert med < 1.5e-3 * tol and worst < 0.2 * min(tol, 4.0), (med, p99, worst)
    # fp32 kernel vs the oracle, for scale: the same points through the parity path
    _, go = O.get_batch_sdf_jacobian(oracle_decoder, code, pts[:64])
    assert np.abs(g32[:64] - go).max() <= 2e-5 * np.abs(go).max()


def test_lp_jacobian_matches_the_cpu_emulator(oracle_decoder, eng):
    """One wave's 32 points: the device against tests/lp_emulator.py's replay of the SAME packed streams -- differences are fp32 summation
    order inside an MFMA (the emulator sums in float64 and rounds once) and nothing else; a wrong slot, mask bit or kept row would be O(1)."""
    import kernel_emulator as KE
    import lp_emulator as LE
    pk = KE.debug_pack(oracle_decoder.layers, oracle_decoder.latent_in, oracle_decoder.code_len)
    rng = np.random.default_rng(5)
    code = (rng.normal(size=64) * 0.1).astype(np.float32)
    pts = rng.uniform(-0.7, 0.7, size=(32, 3)).astype(np.float32)
    for dtype in (L.COMPUTE_F16, L.COMPUTE_BF16):
        lp, lpj = LE.debug_pack(pk["_holder"], dtype), LE.debug_pack_lpj(pk["_holder"], dtype)
        ye, ge = LE.run_wave_jac(pk, lp, lpj, code, pts, dtype)
        y, g = eng.sdf_jacobian_lp(code, pts, dtype)
        assert np.abs(y - ye).max() < 2e-6
        per_point = np.abs(g - ge).max(1) / np.abs(ge).max()
        # (a relu unit within fp32 round-off of zero may flip between the two: its contribution, on a point or two at most)
        assert np.median(per_point) < 2e-5 and (per_point > 1e-3).sum() <= 2, per_point


def test_lp_jacobian_other_decoders(chairs32_decoder):
    decs = [("chairs32", chairs32_decoder)]
    if have_complex_fixture():
        decs.append(("complex", O.fold_decoder(fixtures.load_decoder_npz(fixtures.fixture_path("complex")), fixtures.fixture_specs("complex"))))
    for name, dec in decs:
        e = E.Engine(dec.layers, dec.latent_in, dec.code_len, device=0)
        rng = np.random.default_rng(3)
        code = (rng.normal(size=dec.code_len) * 0.1).astype(np.float32)
        pts = rng.uniform(-0.7, 0.7, size=(4097, 3)).astype(np.float32)
        y32, g32 = e.sdf_jacobian(code, pts)
        y, g = e.sdf_jacobian_lp(code, pts, L.COMPUTE_F16)
        med, p99, worst = _grad_err(g, g32)
        print("%s decoder, f16 jacobian: median %.2e, 99 %% %.2e, max %.2e; |dsdf| %.2e" % (name, med, p99, worst, np.abs(y - y32).max()))
        parity_log(kind="lp_jacobian_kernel", case="f16, %s decoder, 4097 unit-ball points" % name, grad_rel_median=med, grad_rel_p99=p99, grad_rel_max=worst,
                   sdf_abs_max=float(np.abs(y - y32).max()))
        assert g.shape == g32.shape and med < 1.5e-3 and worst < 0.2 and np.abs(y - y32).max() < 2e-3
        e.close()


def test_lp_mode_is_refused_where_it_does_not_apply():
    import copy
    sp = copy.deepcopy(fixtures.SPECS)
    sp["NetworkSpecs"].update(dims=[512] * 6, latent_in=[3], norm_layers=list(range(6)), dropout=list(range(6)))
    dec = O.fold_decoder(fixtures.random_state_dict(9, sp), sp)
    e = E.Engine(dec.layers, dec.latent_in, dec.code_len, device=0)
    o = synth.make_object(1, 100, 30)
    b = e.batch(E.gn_params(num_iterations=1), [o["t_cam_obj_init"]], [o["pts"]], [o["rays"]], [o["depth"]])
    with pytest.raises(L.DspError, match="eight hidden layers"):
        b.set_compute(L.COMPUTE_F16)
    with pytest.raises(L.DspError):
        e.sdf_jacobian_lp(np.zeros(64, np.float32), o["pts"][:4])
    b.run()                                   # ... and the batch runs in fp32 as ever
    assert b.results()[3][0] in (0, 1, 2)
    b.close()
    e.close()


CASES = ["golden_recon_small.npz", "golden_recon_cfg1.npz", "golden_recon_redwood.npz", "golden_recon_freiburg.npz", "golden_recon_cfg2.npz"]


@pytest.mark.parametrize("dtype", [L.COMPUTE_F16, L.COMPUTE_BF16])
def test_lp_compute_at_the_references_recorded_states(eng, dtype):
    """The accuracy table: at every recorded state of five goldens (35 linearisations) the mode's V, K, H, b, dx and loss against the REFERENCE'S
    recorded values.  V is identical by construction (the in-sphere test is fp32 geometry); K moves by the samples whose 16-bit sdf falls on
    the other side of +-th or whose de_do crosses 1e-2; H and b carry f16 rounding of every decoder output plus those rows."""
    tag = "f16" if dtype == L.COMPUTE_F16 else "bf16"
    rows = []
    for name in CASES:
        g = golden(name)
        cfg = json.loads(str(g["cfg_json"]))
        prm = E.params_from_configs(cfg)
        code0 = [g["in_code"]] if "in_code" in g.files else None
        b = eng.batch(prm, [g["in_t_cam_obj_init"]], [g["in_pts"]], [g["in_rays"]], [g["in_depth"]], code0, trace=True)
        b.set_compute(dtype)
        b.set_lp_small_batches(1)          # (four of the five are detection-sized: by itself the mode would leave them on the fp32 latency path)
        mask = np.ones(71, bool)
        mask[3:6] = False
        for e in range(g["it_H"].shape[0]):
            tr, status = F.device_linearisation(b, g["it_t_obj_cam"][e], g["it_code"][e], g["it_depths"][e])
            assert status == 0
            rows.append(dict(case=name, it=e, dV=int(tr["V"][0]) - int(g["it_V"][e]), dK=int(tr["K"][0]) - int(g["it_K"][e]), K=int(g["it_K"][e]),
                             rel_H=F.rel_max(tr["H"][0], g["it_H"][e]), rel_b=F.rel_max(tr["b"][0][mask], g["it_b"][e][mask]),
                             rel_dx=F.rel_max(tr["dx"][0], g["it_dx"][e]), rel_loss=F.loss_rel(tr["loss"][0], g["it_loss"][e])))
        st = b.stats()
        assert st["n_mlp_fwd_launches"] == 0 and st["n_mlp_prepass_launches"] > 0 and st["prepass_guard_trips"] == 0     # no fp32 decoder launch at all
        b.close()
    worst = {k: max(abs(r[k]) for r in rows) for k in ("dV", "dK", "rel_H", "rel_b", "rel_dx", "rel_loss")}
    med = {k: float(np.median([abs(r[k]) for r in rows])) for k in ("rel_H", "rel_b", "rel_dx", "rel_loss")}
    rel_dk = max(abs(r["dK"]) / max(r["K"], 1) for r in rows)
    print("%s compute at %d recorded reference states: |dV| <= %d, |dK| <= %d (%.3f of K), rel dH median %.2e max %.2e, rel db median %.2e max %.2e, "
          "rel ddx median %.2e max %.2e, rel dloss median %.2e max %.2e" % (tag, len(rows), worst["dV"], worst["dK"], rel_dk, med["rel_H"], worst["rel_H"],
                                                                          med["rel_b"], worst["rel_b"], med["rel_dx"], worst["rel_dx"], med["rel_loss"], worst["rel_loss"]))
    parity_log(kind="lp_compute_at_reference_states", case="%s compute mode, five goldens, every recorded iteration" % tag, n=len(rows), rows=rows, worst=worst, median=med,
               worst_dK_over_K=rel_dk)
    assert worst["dV"] == 0
    bound = 1.0 if dtype == L.COMPUTE_F16 else 8.0
    assert rel_dk <= 0.05 * bound and worst["rel_H"] <= 0.05 * bound and worst["rel_loss"] <= 0.05 * bound, worst


BENCH_GOLD = "golden_bench_cfg2x64.npz"


@pytest.mark.skipif(not os.path.exists(os.path.join(GOLDEN, BENCH_GOLD)), reason="tests/golden/%s not generated" % BENCH_GOLD)
def test_lp_compute_at_the_bench_objects_recorded_states(eng):
    """The same table on the HEADLINE workload: the sixteen traced bench objects at the reference's recorded states, all ten iterations, inside
    the resident 64-object batch running in the f16 mode (160 linearisations; with the 35 above: 195 of the 205 the fp32 path is held to --
    the other ten are the 32-D decoder's and the complex one's, whose kernels test_lp_jacobian_other_decoders covers)."""
    g = golden(BENCH_GOLD)
    objs = synth.make_batch(int(g["all_it_V"].shape[0]), first_seed=int(g["first_seed"]), n_surface=int(g["n_surface"]), n_background=int(g["n_background"]))
    cfg = json.loads(str(g["cfg_json"]))
    prm = E.params_from_configs(cfg)
    n_d = int(cfg["optimizer"]["num_depth_samples"]) if "num_depth_samples" in cfg["optimizer"] else 50
    B = len(objs)
    full = [int(i) for i in g["full_objects"]]
    b = eng.batch(prm, [o["t_cam_obj_init"] for o in objs], [o["pts"] for o in objs], [o["rays"] for o in objs], [o["depth"] for o in objs], trace=True)
    b.set_compute(L.COMPUTE_F16)
    zero_codes = [np.zeros(64, np.float32)] * B
    b.set_start_state(None, zero_codes, None)
    b.set_iterations(1)
    b.run()
    base = b.trace(0)
    mask = np.ones(71, bool)
    mask[3:6] = False
    rows = []
    for e in range(10):
        t_oc = [base["t_obj_cam"][i] for i in range(B)]
        codes = [base["code"][i] for i in range(B)]
        depths = [base["depths"][i][:n_d] for i in range(B)]
        for i in full:
            t_oc[i], codes[i], depths[i] = g["tr%d_it_t_obj_cam" % i][e], g["tr%d_it_code" % i][e], g["tr%d_it_depths" % i][e]
        b.set_start_state(t_oc, codes, depths)
        b.set_iterations(1)
        b.run()
        res = b.results()
        tr = b.trace(0)
        for i in full:
            assert res[3][i] == 0
            assert np.array_equal(tr["t_obj_cam"][i], g["tr%d_it_t_obj_cam" % i][e]) and np.array_equal(tr["code"][i], g["tr%d_it_code" % i][e])
            k_ref = int(g["tr%d_it_K" % i][e])
            rows.append(dict(case="bench object %d" % i, it=e, dV=int(tr["V"][i]) - int(g["tr%d_it_V" % i][e]), dK=int(tr["K"][i]) - k_ref, K=k_ref,
                             rel_H=F.rel_max(tr["H"][i], g["tr%d_it_H" % i][e]), rel_b=F.rel_max(tr["b"][i][mask], g["tr%d_it_b" % i][e][mask]),
                             rel_dx=F.rel_max(tr["dx"][i], g["tr%d_it_dx" % i][e]), rel_loss=F.loss_rel(res[2][i], g["tr%d_it_loss" % i][e])))
    st = b.stats()
    assert st["n_mlp_fwd_launches"] == 0 and st["n_mlp_prepass_launches"] > 0
    b.close()
    worst = {k: max(abs(r[k]) for r in rows) for k in ("dV", "dK", "rel_H", "rel_b", "rel_dx", "rel_loss")}
    med = {k: float(np.median([abs(r[k]) for r in rows])) for k in ("rel_H", "rel_b", "rel_dx", "rel_loss")}
    rel_dk = max(abs(r["dK"]) / max(r["K"], 1) for r in rows)
    print("f16 compute at %d recorded states of the bench objects: |dV| <= %d, |dK| <= %d (%.4f of K), rel dH median %.2e max %.2e, rel db median %.2e max %.2e, "
          "rel ddx median %.2e max %.2e, rel dloss median %.2e max %.2e" % (len(rows), worst["dV"], worst["dK"], rel_dk, med["rel_H"], worst["rel_H"],
                                                                          med["rel_b"], worst["rel_b"], med["rel_dx"], worst["rel_dx"], med["rel_loss"], worst["rel_loss"]))
    parity_log(kind="lp_compute_at_reference_states", case="f16 compute mode, 64 x cfg2 bench batch: 16 traced objects x 10 iterations inside the resident batch", n=len(rows),
               rows=rows, worst=worst, median=med, worst_dK_over_K=rel_dk)
    assert worst["dV"] == 0 and len(rows) == 160
    assert rel_dk <= 0.05 and worst["rel_H"] <= 0.05 and worst["rel_loss"] <= 0.05, worst


def test_lp_compute_chained_runs(eng):
    """Chained ten-iteration runs in the f16 mode: good status wherever the fp32 path's is, run-to-run deterministic, and the final pose / code
    against the fp32 path's and the reference's own spread (golden ulps_* / thr_*) -- REPORTED with a sanity bound; the default path returns
    its usual bits on the same handle afterwards."""
    import test_gpu_parity as P
    out = []
    for name in CASES:
        g = golden(name)
        cfg = json.loads(str(g["cfg_json"]))
        prm = E.params_from_configs(cfg)
        code0 = [g["in_code"]] if "in_code" in g.files else None
        args = ([g["in_t_cam_obj_init"]], [g["in_pts"]], [g["in_rays"]], [g["in_depth"]], code0)
        t32, c32, l32, s32 = eng.reconstruct_batch(prm, *args)
        b = eng.batch(prm, *args)
        b.set_compute(L.COMPUTE_F16)
        b.set_lp_small_batches(1)
        b.run()
        t, c, l, s = b.results()
        b.run()
        for x, y in zip(b.results(), (t, c, l, s)):
            assert np.array_equal(x, y), "the low-precision mode is not run-to-run deterministic"
        b.set_compute(L.COMPUTE_F32)
        b.run()
        for x, y in zip(b.results(), (t32, c32, l32, s32)):
            assert np.array_equal(x, y), "the fp32 path changed after the batch ran in the low-precision mode"
        b.close()
        assert s[0] == 0 == s32[0]
        m, sens, _ = P.end_to_end_differences(g, t[0], c[0])
        m32, _, _ = P.end_to_end_differences(g, t32[0], c32[0])
        rec = dict(case=name, lp_vs_reference={q: m[q] for q in ("rot", "scale", "trans", "code")}, fp32_vs_reference={q: m32[q] for q in ("rot", "scale", "trans", "code")},
                   reference_spread={q: sens[q] for q in ("rot", "scale", "trans", "code")}, loss_rel_vs_reference=F.loss_rel(l[0], g["loss"]))
        out.append(rec)
        print("%s: f16 mode vs reference rot %.1e trans %.1e code %.1e | fp32 path %.1e %.1e %.1e | reference's own spread %.1e %.1e %.1e" % (
            name, m["rot"], m["trans"], m["code"], m32["rot"], m32["trans"], m32["code"], sens["rot"], sens["trans"], sens["code"]))
        for q in ("rot", "scale", "trans", "code"):
            assert m[q] <= max(2e-2, 10.0 * sens[q]), (name, q, m[q], sens[q])
    parity_log(kind="lp_compute_chained", case="f16 compute mode, chained runs of five goldens", rows=out)


def test_lp_compute_batch_throughput_shape(eng):
    """Eight cfg2-size objects in one batch: every object good, the jacobian launches are the 16-bit pair, and the per-iteration decoder time is
    well below the fp32 path's (the bench's value_lp is the measured figure)."""
    objs = synth.make_batch(8, first_seed=1, n_surface=2000, n_background=500)
    prm = E.gn_params()
    args = ([o["t_cam_obj_init"] for o in objs], [o["pts"] for o in objs], [o["rays"] for o in objs], [o["depth"] for o in objs])
    res = {}
    for mode in (L.COMPUTE_F32, L.COMPUTE_F16):
        b = eng.batch(prm, *args)
        b.set_kernel_timing(1)
        b.set_compute(mode)
        b.run()
        b.run()
        res[mode] = (b.results(), b.stats())
        b.close()
    (r32, s32), (r16, s16) = res[L.COMPUTE_F32], res[L.COMPUTE_F16]
    assert (r16[3] == 0).all() and (r32[3] == 0).all()
    assert s16["n_mlp_fwd_launches"] == 0 and s16["n_mlp_jac_launches"] == prm.num_iterations
    dec32 = s32["ms_mlp_fwd"] + s32["ms_mlp_jac"] + s32["ms_mlp_prepass"]
    dec16 = s16["ms_mlp_fwd"] + s16["ms_mlp_jac"] + s16["ms_mlp_prepass"]
    print("8 cfg2 objects: decoder time per run %.1f ms (fp32 path with f16 classifier) -> %.1f ms (f16 compute mode); whole run %.1f -> %.1f ms" % (
        dec32, dec16, s32["ms_total"], s16["ms_total"]))
    parity_log(kind="lp_compute_speed", case="8 cfg2 objects, one batch", decoder_ms_fp32_path=dec32, decoder_ms_lp=dec16, run_ms_fp32_path=s32["ms_total"], run_ms_lp=s16["ms_total"])
    assert dec16 < 0.6 * dec32
    dt = np.abs(r16[0] - r32[0]).max(axis=(1, 2)) / np.abs(r32[0]).max(axis=(1, 2))
    print("   final pose, f16 mode vs fp32 path, per object (relative): %s" % np.array2string(dt, precision=1))


def test_detection_sized_batches_keep_the_fp32_latency_path(eng):
    """With the mode set, a batch of SLAM's own per-detection size keeps the fp32 latency path (faster there: 2.89 against 3.40 ms, and exact):
    same bits as a batch that never heard of the mode, cluster tiles in use, no 16-bit jacobian launch.  A cfg2-size object does switch."""
    prm = E.gn_params()
    o = synth.make_object(4242, n_surface=250, n_background=200)
    args = ([o["t_cam_obj_init"]], [o["pts"]], [o["rays"]], [o["depth"]])
    want = eng.reconstruct_batch(prm, *args)
    b = eng.batch(prm, *args)
    b.set_compute(L.COMPUTE_F16)
    b.run()
    st = b.stats()
    for x, y in zip(b.results(), want):
        assert np.array_equal(x, y, equal_nan=True)
    assert st["n_cluster_tiles"] > 0
    b.set_lp_small_batches(1)                  # pinned on: now it is the 16-bit pair
    b.run()
    st1 = b.stats()
    assert st1["n_cluster_tiles"] == 0 and b.results()[3][0] == 0
    assert not np.array_equal(b.results()[0], want[0])
    b.close()
    big = synth.make_object(1, n_surface=2000, n_background=500)
    b = eng.batch(prm, [big["t_cam_obj_init"]], [big["pts"]], [big["rays"]], [big["depth"]])
    b.run()
    r32 = b.results()
    b.set_compute(L.COMPUTE_F16)
    b.run()
    st = b.stats()
    assert st["n_mlp_fwd_launches"] == 0 and b.results()[3][0] == 0 and not np.array_equal(b.results()[0], r32[0])
    b.close()


def test_mirror_api_opt_in(tmp_path):
    """`"compute_dtype": "f16"` under "optimizer" (an ADDITION: the reference's configs do not have the key, its absence is fp32) switches the
    mirror's Optimizer to the low-precision mode: same call surface, same result dict, a result within the mode's accuracy of the fp32 one."""
    import sys
    pkg = os.path.join(ROOT, "dsp_slam_amd")
    sys.path.insert(0, pkg)
    try:
        for m in [k for k in sys.modules if k.split(".")[0] in ("reconstruct", "deep_sdf")]:
            del sys.modules[m]
        from reconstruct.utils import get_configs, get_decoder
        from reconstruct.optimizer import Optimizer
        g = golden("golden_recon_freiburg.npz")
        cfg_d = json.loads(str(g["cfg_json"]))
        cfg_d["DeepSDF_DIR"] = fixtures.materialize_decoder_dir("cars", str(tmp_path / "cars_64"))
        out, big = {}, {}
        o2 = synth.make_object(1, n_surface=2000, n_background=500)
        for dt in ("f32", "f16"):
            c = json.loads(json.dumps(cfg_d))
            if dt != "f32":
                c["optimizer"]["compute_dtype"] = dt
            with open(tmp_path / ("cfg_%s.json" % dt), "w") as f:
                json.dump(c, f)
            cfg = get_configs(str(tmp_path / ("cfg_%s.json" % dt)))
            opt = Optimizer(get_decoder(cfg), cfg)
            opt.verbose = False
            assert opt.compute == (L.COMPUTE_F16 if dt == "f16" else L.COMPUTE_F32)
            out[dt] = opt.reconstruct_object(g["in_t_cam_obj_init"], g["in_pts"], g["in_rays"], g["in_depth"])
            big[dt] = opt.reconstruct_object(o2["t_cam_obj_init"], o2["pts"], o2["rays"], o2["depth"])
            assert out[dt].is_good is True and big[dt].is_good is True
        assert np.abs(out["f32"].t_cam_obj - g["t_cam_obj"]).max() <= 1e-4 * np.abs(g["t_cam_obj"]).max()        # the parity path, as ever
        # a detection of SLAM's own size keeps the fp32 latency path with the key set (faster there): the same bits
        assert np.array_equal(out["f16"].t_cam_obj, out["f32"].t_cam_obj) and np.array_equal(out["f16"].code, out["f32"].code)
        # a cfg2-size object runs in the mode (chained ten iterations of a chaotic object: the bench line's pose_rel_max is 5e-3 over 64 of them)
        d = np.abs(big["f16"].t_cam_obj - big["f32"].t_cam_obj).max() / np.abs(big["f32"].t_cam_obj).max()
        assert 0 < d < 5e-2, d
        c = json.loads(json.dumps(cfg_d))
        c["optimizer"]["compute_dtype"] = "fp8"
        with open(tmp_path / "cfg_bad.json", "w") as f:
            json.dump(c, f)
        with pytest.raises(ValueError):
            Optimizer(None, get_configs(str(tmp_path / "cfg_bad.json")))
    finally:
        sys.path.remove(pkg)
        for m in [k for k in sys.modules if k.split(".")[0] in ("reconstruct", "deep_sdf")]:
            del sys.modules[m]

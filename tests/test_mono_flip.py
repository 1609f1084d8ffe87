"""The result field DSP-SLAM's mono path BRANCHES on.

src/LocalMapping_util.cc:391-406: an object that is not reconstructed yet is optimised twice -- from its pose and from the pose turned by
180 degrees about the object's up axis -- and C++ keeps the hypothesis whose returned `loss` is smaller (`if (a.loss > b.loss) a = b`).
tests/golden/golden_mono_flip.npz (tools/make_golden_mono_flip.py) holds both runs of the UNMODIFIED reference on one SLAM-size detection
(250 surface points, 200 background rays, Freiburg hyper-parameters, the fore / aft asymmetric complex-car decoder): `a_*` / `b_*`,
full per-iteration traces with `it_loss`.  The reference's losses are 0.1214 (a) and 0.0951 (b): C++ keeps the flipped hypothesis.

CPU tier: the oracle reproduces the reference's loss at all ten recorded states (1e-5) and takes the same branch from its own chained runs.
GPU tier: the device at all ten recorded states (loss to 1e-4, V and K identical, H / b to the forensics bounds), and the two chained runs through the mirror's
`Optimizer.reconstruct_object` -- the five-argument form C++ calls -- take the reference's branch, each chained loss within 1e-2 of the
reference's (the gap between the hypotheses is 0.28).
"""
import json
import os
import sys

import numpy as np
import pytest

import forensics as F
from conftest import ROOT, golden, have_complex_fixture, parity_log
from oracle import dsp_oracle as O

GOLD = "golden_mono_flip.npz"
have = have_complex_fixture() and os.path.exists(os.path.join(ROOT, "tests", "golden", GOLD))
pytestmark = pytest.mark.skipif(not have, reason="golden_mono_flip.npz / the complex decoder fixture not generated")


def _hyp(g, tag):
    return {k[len(tag):]: g[k] for k in g.files if k.startswith(tag)}


def test_golden_is_the_cpp_sequence():
    """What the golden holds: the same detection twice, the second start pose = the first with columns 0 and 2 negated
    (`flipped_Two.col(0) *= -1; flipped_Two.col(2) *= -1`, LocalMapping_util.cc:399-401), both runs good, a clear gap between the losses."""
    g = golden(GOLD)
    a, b = _hyp(g, "a_"), _hyp(g, "b_")
    for k in ("in_pts", "in_rays", "in_depth"):
        assert np.array_equal(a[k], b[k])
    flip = np.diag([-1.0, 1.0, -1.0, 1.0]).astype(np.float32)
    assert np.array_equal(b["in_t_cam_obj_init"], (a["in_t_cam_obj_init"] @ flip).astype(np.float32))
    assert bool(a["is_good"]) and bool(b["is_good"])
    assert a["it_loss"].shape == (5,) and float(a["it_loss"][-1]) == float(a["loss"]) and float(b["it_loss"][-1]) == float(b["loss"])
    gap = abs(float(a["loss"]) - float(b["loss"])) / min(float(a["loss"]), float(b["loss"]))
    assert gap > 0.1, "the two hypotheses must not be a coin toss for this test to mean anything"


def test_oracle_takes_the_references_branch(complex_decoder):
    g = golden(GOLD)
    chained = {}
    for tag in ("a_", "b_"):
        h = _hyp(g, tag)
        prm = O.GNParams.from_configs(json.loads(str(h["cfg_json"])))
        for e in range(h["it_H"].shape[0]):
            it = F.oracle_linearisation(complex_decoder, prm, h["in_pts"], h["in_rays"], h["in_depth"], h["it_t_obj_cam"][e], h["it_code"][e], h["it_depths"][e])
            assert (it["V"], it["K"]) == (int(h["it_V"][e]), int(h["it_K"][e]))
            assert F.loss_rel(it["loss"], h["it_loss"][e]) <= 1e-5, (tag, e)
        rst = O.reconstruct_object(complex_decoder, prm, h["in_t_cam_obj_init"], h["in_pts"], h["in_rays"], h["in_depth"], np.zeros(64, np.float32))
        assert rst["is_good"]
        chained[tag] = rst["loss"]
        assert F.loss_rel(rst["loss"], h["loss"]) <= 1e-3, (tag, rst["loss"], float(h["loss"]))
    assert (chained["a_"] > chained["b_"]) == (float(g["a_loss"]) > float(g["b_loss"]))


@pytest.mark.gpu
def test_device_loss_at_both_hypotheses_recorded_states(complex_decoder):
    from dsp_slam_amd import engine as E
    g = golden(GOLD)
    eng = E.Engine(complex_decoder.layers, complex_decoder.latent_in, complex_decoder.code_len, device=0)
    try:
        out = {}
        for tag in ("a_", "b_"):
            h = _hyp(g, tag)
            cfg = json.loads(str(h["cfg_json"]))
            prm = E.params_from_configs(cfg)
            b = eng.batch(prm, [h["in_t_cam_obj_init"]], [h["in_pts"]], [h["in_rays"]], [h["in_depth"]], trace=True)
            rels = []
            for e in range(h["it_H"].shape[0]):
                tr, status = F.device_linearisation(b, h["it_t_obj_cam"][e], h["it_code"][e], h["it_depths"][e])
                assert status == 0
                assert (int(tr["V"][0]), int(tr["K"][0])) == (int(h["it_V"][e]), int(h["it_K"][e])), (tag, e)
                assert F.rel_max(tr["H"][0], h["it_H"][e]) < 3e-5 and F.rel_max(tr["b"][0], h["it_b"][e]) < 1.2e-4      # k4 = 0 here: no rotation-prior rows to exempt
                rels.append(F.loss_rel(tr["loss"][0], h["it_loss"][e]))
                assert rels[-1] <= F.LOSS_RTOL, (tag, e, float(tr["loss"][0]), float(h["it_loss"][e]))
            b.close()
            out[tag] = rels
        parity_log(kind="mono_flip_at_reference_states", case=GOLD, rel_loss_a=out["a_"], rel_loss_b=out["b_"], n=len(out["a_"]) + len(out["b_"]))
    finally:
        eng.close()


@pytest.mark.gpu
def test_mirror_api_takes_the_references_branch(tmp_path):
    """The C++ sequence itself on the mirror package: two reconstruct_object calls in the five-argument form, then the comparison of the two
    `loss` fields cast to float (LocalMapping_util.cc:391-406)."""
    from dsp_slam_amd import fixtures
    g = golden(GOLD)
    a, b = _hyp(g, "a_"), _hyp(g, "b_")
    pkg = os.path.join(ROOT, "dsp_slam_amd")
    sys.path.insert(0, pkg)
    try:
        for m in [k for k in sys.modules if k.split(".")[0] in ("reconstruct", "deep_sdf")]:
            del sys.modules[m]
        from reconstruct.utils import get_configs, get_decoder
        from reconstruct.optimizer import Optimizer
        cfg_d = json.loads(str(a["cfg_json"]))
        cfg_d["DeepSDF_DIR"] = fixtures.materialize_decoder_dir("complex", str(tmp_path / "complex_64"))
        with open(tmp_path / "cfg.json", "w") as f:
            json.dump(cfg_d, f)
        cfg = get_configs(str(tmp_path / "cfg.json"))
        opt = Optimizer(get_decoder(cfg), cfg)
        opt.verbose = False
        code0 = np.zeros(64, np.float32)                     # pMO->vShapeCode of a fresh object
        f_order = lambda x: np.asfortranarray(x)             # noqa: E731  what pybind11's Eigen caster hands over
        r_a = opt.reconstruct_object(f_order(a["in_t_cam_obj_init"]), f_order(a["in_pts"]), f_order(a["in_rays"]), a["in_depth"], code0)
        r_b = opt.reconstruct_object(f_order(b["in_t_cam_obj_init"]), f_order(b["in_pts"]), f_order(b["in_rays"]), b["in_depth"], code0)
        la, lb = float(r_a.loss), float(r_b.loss)            # .attr("loss").cast<float>()
        keep_flipped_dev = la > lb
        keep_flipped_ref = float(a["loss"]) > float(b["loss"])
        rel_a, rel_b = F.loss_rel(la, a["loss"]), F.loss_rel(lb, b["loss"])
        print("mono flip: device losses %.6f / %.6f, reference %.6f / %.6f (rel %.1e / %.1e); C++ keeps %s" % (
            la, lb, float(a["loss"]), float(b["loss"]), rel_a, rel_b, "the flipped hypothesis" if keep_flipped_dev else "the detection's pose"))
        parity_log(kind="mono_flip_chained", case=GOLD, device_loss=[la, lb], reference_loss=[float(a["loss"]), float(b["loss"])], rel=[rel_a, rel_b],
                   same_branch=bool(keep_flipped_dev == keep_flipped_ref))
        assert keep_flipped_dev == keep_flipped_ref
        # chained over five iterations the states drift apart as in every chained comparison (measured on MI355X: 1.0e-4 and
        # 2.0e-3, the oracle's own chained run: < 1e-3); what pins the field is the test above (1e-4 at every recorded state).  Here the
        # bound is what the BRANCH needs: an order of magnitude below the gap between the hypotheses (0.28)
        assert rel_a <= 1e-2 and rel_b <= 1e-2
        kept, kept_ref = (r_b, b) if keep_flipped_dev else (r_a, a)
        assert kept.is_good is True
        assert np.abs(kept.t_cam_obj - kept_ref["t_cam_obj"]).max() <= 5e-3 * np.abs(kept_ref["t_cam_obj"]).max()
    finally:
        sys.path.remove(pkg)
        for m in [k for k in sys.modules if k.split(".")[0] in ("reconstruct", "deep_sdf")]:
            del sys.modules[m]

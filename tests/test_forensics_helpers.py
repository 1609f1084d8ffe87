"""CPU: tests/forensics.py -- the helpers that NAME samples which switched sets between two linearisations -- on hand-made grids, so that
their classification does not depend on a GPU run: which set, which threshold, which margin, and the explained / not-explained verdict."""
import numpy as np

import forensics as F


def _grids(n_rays=4, n_d=8, th=0.01):
    in_sphere = np.ones((n_rays, n_d), bool)
    in_sphere[:, 0] = False
    sdf = np.full((n_rays, n_d), 0.5, np.float32)
    sdf[:, 0] = np.nan
    sdf[:, 4] = 0.004           # one band sample per ray
    kept = np.zeros((n_rays, n_d), bool)
    kept[:, 4] = True
    de_do = np.full((n_rays, n_d), np.nan, np.float32)
    de_do[:, 4] = 1.5
    norm = np.full((n_rays, n_d), 0.7, np.float32)
    norm[:, 0] = 1.2
    return dict(in_sphere=in_sphere, sdf=sdf, kept=kept, de_do=de_do, norm=norm)


def test_identical_grids_have_no_flips():
    og = _grids()
    assert F.name_flips(*F.as_device_grids(og), og, 0.01) == []


def test_sphere_flip_is_named_with_its_margin():
    og = _grids()
    og["norm"][2, 0] = np.float32(1.0 + 6e-7)          # oracle: just outside
    dev = [a.copy() for a in F.as_device_grids(og)]
    dev[0][2, 0] = True                                  # device: inside
    dev[1][2, 0] = 0.3
    dev[2][2, 0] = 0.0
    fl = F.name_flips(dev[0], dev[1], dev[2], og, 0.01)
    assert len(fl) == 1 and fl[0]["set"] == "in_sphere" and fl[0]["threshold"] == "norm<1" and (fl[0]["ray"], fl[0]["depth_index"]) == (2, 0)
    assert fl[0]["explained"] and abs(fl[0]["margin"] - 6e-7) < 2e-7
    og["norm"][2, 0] = np.float32(1.01)                  # a sample 1e-2 outside cannot be round-off
    assert not F.name_flips(dev[0], dev[1], dev[2], og, 0.01)[0]["explained"]


def test_band_edge_flip_and_de_do_flip_are_told_apart():
    og = _grids()
    og["sdf"][1, 4] = np.float32(0.0099996)              # oracle: inside the band by 4e-7, kept
    dev = [a.copy() for a in F.as_device_grids(og)]
    dev[1][1, 4] = np.float32(0.0100003)                 # device: outside -> not kept
    dev[2][1, 4] = 0.0
    fl = F.name_flips(dev[0], dev[1], dev[2], og, 0.01)
    assert len(fl) == 1 and fl[0]["threshold"] == "|sdf|<th" and fl[0]["explained"] and not fl[0]["device"] and fl[0]["oracle"]
    # same band membership on both sides, different kept flag: the de_do > 1e-2 test decided
    og = _grids()
    og["de_do"][3, 4] = np.float32(0.010001)
    dev = [a.copy() for a in F.as_device_grids(og)]
    dev[2][3, 4] = 0.0
    fl = F.name_flips(dev[0], dev[1], dev[2], og, 0.01)
    assert len(fl) == 1 and fl[0]["threshold"] == "de_do>1e-2" and fl[0]["explained"] and fl[0]["margin"] < 2e-4
    og["de_do"][3, 4] = np.float32(0.5)
    assert not F.name_flips(dev[0], dev[1], dev[2], og, 0.01)[0]["explained"]


def test_flip_tolerances_widen_with_the_incoming_difference():
    tight, loose = F.flip_tolerances(0.0, 0.0), F.flip_tolerances(1e-4, 2e-6)
    assert tight["norm<1"] == F.TOL_NORM and tight["|sdf|<th"] == F.TOL_SDF
    assert all(loose[k] > tight[k] for k in tight)
    assert F.first_differing_iteration([(5, 2), (5, 3)], {"it_V": [5, 5], "it_K": [2, 2]}) == 1
    assert F.first_differing_iteration([(5, 2)], {"it_V": [5], "it_K": [2]}) is None

"""CPU-only: the packed weight stream + pass table (exactly what dsp_create uploads) replayed through a
numpy model of the kernel's register data flow must reproduce the oracle's decoder forward and
input-gradient.  Guards the MFMA operand permutation, row conventions, masks and skip gradients."""
import numpy as np
import pytest

from oracle import dsp_oracle as O
from dsp_slam_amd import fixtures
import kernel_emulator as KE


@pytest.fixture(scope="module")
def packed():
    sd = fixtures.random_state_dict(3)
    # make the random net lively: larger last-layer weights so tanh' and the masks matter
    dec = O.fold_decoder(sd, fixtures.SPECS)
    pk = KE.debug_pack(dec.layers, dec.latent_in, dec.code_len)
    return dec, pk


def test_pass_table(packed):
    dec, pk = packed
    # layer 0 runs on the VALU (code folded into a per-object bias), so 7 forward + 8 backward passes stream weights
    assert pk["n_fwd"] == 7 and pk["n_pass"] == 15 and pk["n_bias_rows"] == 11
    assert pk["wlast_row"] == 7 and pk["w0_row"] == 8
    p = pk["passes"]
    assert list(p[:7, 0]) == [8, 8, 7, 8, 8, 8, 8]             # 445-wide layer 3 -> 7 output groups
    assert list(p[:7, 1]) == [8, 8, 8, 7, 8, 8, 8]             # latent_in layer: K = 445 + 3 -> 7 chunks
    assert list(p[:7, 2]) == [0, 1, 2, -2, 4, 5, 6]            # bias rows; -2 = per-object code bias
    assert list(p[7:, 5]) == [3, 3, 3, 4, 3, 3, 3, 5]          # backward kinds, latent_in at layer 4
    assert list(p[7:, 1]) == [8, 8, 8, 8, 7, 8, 8, 8]          # K = 448 for the 445-wide layer
    assert list(p[7:, 0]) == [8, 8, 8, 8, 8, 8, 8, 1]          # first layer backward: the 64 code rows only
    assert pk["chunks_fwd"] == 64 * 2 + 56 + 56 + 64 * 3
    assert pk["stream"].shape[0] == pk["chunks_all"] == pk["chunks_fwd"] + 64 * 3 + 64 + 56 + 64 * 2 + 8


def test_emulated_forward_matches_oracle(packed):
    dec, pk = packed
    rng = np.random.default_rng(0)
    code = (rng.normal(size=64) * 0.3).astype(np.float32)
    pts = rng.uniform(-0.8, 0.8, size=(16, 3)).astype(np.float32)
    sdf = KE.run_wave(pk, code, pts, bwd=False)
    ref = O.decode_sdf(dec, code, pts)
    assert np.abs(sdf - ref).max() < 2e-6


def test_emulated_jacobian_matches_oracle(packed):
    dec, pk = packed
    rng = np.random.default_rng(1)
    code = (rng.normal(size=64) * 0.3).astype(np.float32)
    pts = rng.uniform(-0.8, 0.8, size=(16, 3)).astype(np.float32)
    sdf, grad = KE.run_wave(pk, code, pts, bwd=True)
    y, g = O.get_batch_sdf_jacobian(dec, code, pts)
    assert np.abs(sdf - y).max() < 2e-6
    assert np.abs(grad - g).max() < 1e-5 * max(1.0, np.abs(g).max())
    assert np.abs(g[:, :64]).max() > 1e-4 and np.abs(g[:, 64:]).max() > 1e-4   # the check is not vacuous

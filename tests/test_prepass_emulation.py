"""CPU-only: the packed PREPASS weight stream (exactly what dsp_create uploads for mlp_lp_kernel.hip) replayed through a
numpy model of one wave's register data flow must reproduce the prepass arithmetic stated directly on the folded
decoder, and stay within the calibrated distance of the fp32 oracle.  Guards the 32x32x16 operand permutation, the
split-precision xyz k-steps, the ping-pong slabs and the 445-row layer's padding."""
import numpy as np
import pytest

from oracle import dsp_oracle as O
from dsp_slam_amd import fixtures, _lib as L
import kernel_emulator as KE
import lp_emulator as LE


@pytest.fixture(scope="module")
def packed():
    sd = fixtures.load_decoder_npz(fixtures.fixture_path("cars"))
    dec = O.fold_decoder(sd, fixtures.SPECS)
    pk = KE.debug_pack(dec.layers, dec.latent_in, dec.code_len)
    return dec, pk


@pytest.mark.parametrize("dtype", [L.PREPASS_F16, L.PREPASS_BF16])
def test_pass_table(packed, dtype):
    dec, pk = packed
    lp = LE.debug_pack(pk["_holder"], dtype)
    p = lp["passes"]
    assert lp["n_pass"] == 8
    assert list(p[:, 0]) == [8] * 8                             # the 445-wide layer 3 is padded to 8 groups of zero rows
    assert list(p[:, 1]) == [1, 4, 4, 4, 4, 4, 4, 4]            # first layer: the xyz k-steps only
    assert list(p[:, 2]) == [-3, 0, 1, 2, -2, 4, 5, 6]          # -3 / -2: per-object code bias of layer 0 / latent_in
    assert list(p[:, 3]) == [0, 1, 1, 1, 2, 1, 1, 1]
    assert list(p[:, 4]) == [0, 0, 0, 0, 2, 0, 0, 0]            # latent_in: 28 slab k-steps, 2 padding, 2 xyz
    assert list(p[:, 5]) == [0, 0, 0, 0, 0, 0, 0, 1]
    assert lp["chunks"] == 8 + 32 * 7 == lp["stream"].shape[0]


@pytest.mark.parametrize("dtype,tol_lp,tol_fp32", [(L.PREPASS_F16, 2e-5, 4e-4), (L.PREPASS_BF16, 2e-4, 4e-3)])
def test_emulated_wave_matches_direct_statement(packed, dtype, tol_lp, tol_fp32):
    dec, pk = packed
    lp = LE.debug_pack(pk["_holder"], dtype)
    rng = np.random.default_rng(5)
    code = (rng.normal(size=64) * 0.2).astype(np.float32)
    pts = rng.uniform(-0.7, 0.7, size=(32, 3)).astype(np.float32)
    got = LE.run_wave(pk, lp, code, pts, dtype)
    want = LE.reference_forward(dec, code, pts, dtype)
    ref32 = O.decode_sdf(dec, code, pts)
    # same rounding points; only the fp32-vs-fp64 accumulation differs (an activation may land on the other side of a
    # rounding boundary, one 16-bit ulp in one place)
    assert np.abs(got - want).max() < tol_lp
    assert np.abs(got - ref32).max() < tol_fp32
    assert np.abs(ref32).max() > 0.05                             # not vacuous

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


def test_split_stream_layout(packed):
    """Latency-form kernel: every wave streams exactly the chunks of its two output groups, in consumption order (pass, group,
    chunk), the four per-wave streams partition the throughput stream, and the forward part is a prefix of each."""
    import ctypes as C
    from dsp_slam_amd import _lib as L
    dec, pk = packed
    lib = L.load()
    layout = np.zeros(12, np.int32)
    n = C.c_int64(0)
    L.check(lib.dsp_debug_split_layout(C.byref(pk["_holder"].desc), L.ptr(layout, L.c_i32p), None, C.byref(n)), None, "split(size)")
    assert n.value == pk["chunks_all"]
    ids = np.zeros(n.value, np.int32)
    L.check(lib.dsp_debug_split_layout(C.byref(pk["_holder"].desc), L.ptr(layout, L.c_i32p), L.ptr(ids, L.c_i32p), C.byref(n)), None, "split")
    off, ln, lf = layout[:4], layout[4:8], layout[8:]
    assert sorted(ids.tolist()) == list(range(pk["chunks_all"]))            # a permutation: every chunk once
    assert off[0] == 0 and all(off[w + 1] == off[w] + ln[w] for w in range(3)) and off[3] + ln[3] == pk["chunks_all"]
    p = pk["passes"]
    for w in range(4):
        want, want_fwd = [], 0
        for ps in range(pk["n_pass"]):
            if ps == pk["n_fwd"]:
                want_fwd = len(want)
            nog, nchunks, chunk_base = int(p[ps, 0]), int(p[ps, 1]), int(p[ps, 6])
            for og in (2 * w, 2 * w + 1):
                if og < nog:
                    want += [chunk_base + og * nchunks + c for c in range(nchunks)]
        assert ids[off[w]:off[w] + ln[w]].tolist() == want
        assert lf[w] == want_fwd and all(i < pk["chunks_fwd"] for i in want[:want_fwd])
    # waves 0..2 carry two groups of every pass; wave 3 one group fewer in the two 445/448-row passes; wave 0 the first-layer backward
    assert ln[0] > ln[1] == ln[2] > ln[3]


def test_cluster_stream_layout(packed):
    """Cluster-form kernel (mlp_cluster_kernel.hip): wave slot u of the 16 in a cluster streams row tiles 2u, 2u + 1 of every pass whose
    output group u / 2 exists, as k-step PAIRS: lane element = {tile 0 @ k-step 2p, tile 1 @ 2p, tile 0 @ 2p + 1, tile 1 @ 2p + 1}.  Checked
    against the throughput stream (itself pinned by the emulated forward / jacobian above): same A operands, every one exactly once."""
    import ctypes as C
    from dsp_slam_amd import _lib as L
    dec, pk = packed
    lib = L.load()
    lib.dsp_debug_cluster_stream.restype = C.c_int
    lib.dsp_debug_cluster_stream.argtypes = [C.c_void_p, L.c_i32p, L.c_f32p, L.c_i64p]
    layout = np.zeros(32, np.int32)
    n = C.c_int64(0)
    L.check(lib.dsp_debug_cluster_stream(C.byref(pk["_holder"].desc), L.ptr(layout, L.c_i32p), None, C.byref(n)), None, "cluster(size)")
    assert n.value == pk["stream"].size
    cs = np.zeros(n.value, np.float32)
    L.check(lib.dsp_debug_cluster_stream(C.byref(pk["_holder"].desc), L.ptr(layout, L.c_i32p), L.ptr(cs), C.byref(n)), None, "cluster")
    off, ln = layout[:16], layout[16:]
    minis = cs.reshape(-1, 4, 64, 4)                     # [mini-chunk][pair in mini][lane][element]
    assert off[0] == 0 and all(off[u + 1] == off[u] + ln[u] for u in range(15)) and off[15] + ln[15] == minis.shape[0]
    p = pk["passes"]
    for u in range(16):
        og, half = u >> 1, u & 1
        pos = int(off[u])
        for ps in range(pk["n_pass"]):
            nog, nchunks, chunk_base = int(p[ps, 0]), int(p[ps, 1]), int(p[ps, 6])
            if og >= nog:
                continue
            for c in range(nchunks):
                ch = pk["stream"][chunk_base + og * nchunks + c]          # [k-step 16][lane 64][tile-in-group 4]
                got = minis[pos:pos + 2].reshape(8, 64, 4)                 # 8 pairs
                pos += 2
                for e in range(4):
                    assert np.array_equal(got[:, :, e], ch[(e >> 1)::2, :, 2 * half + (e & 1)]), (u, ps, c, e)
        assert pos == off[u] + ln[u]
    # slots 0 and 1 also carry the first layer's backward pass (one output group); the two top groups skip the 445/448-row passes
    assert ln[0] == ln[1] > ln[2] == ln[13] > ln[14] == ln[15]


def _specs(code_len=64, width=512):
    import copy
    sp = copy.deepcopy(fixtures.SPECS)
    sp["CodeLength"] = code_len
    sp["NetworkSpecs"]["dims"] = [width] * 8
    return sp


@pytest.mark.parametrize("code_len,width", [(32, 512), (64, 256), (32, 256), (64, 384)])
def test_other_code_lengths_and_widths(code_len, width):
    """32-D codes (the Redwood chairs option, LocalMapping_util.cc:415-423; Decoder.__init__ is generic over latent_size and
    dims, deep_sdf_decoder.py:10-73): the re-injected input moves to slab rows 477.., and narrower decoders run embedded in
    the 512-row slabs with zero rows / columns.  Forward, input gradient and the prepass stream against the oracle."""
    sp = _specs(code_len, width)
    dec = O.fold_decoder(fixtures.random_state_dict(11 + code_len + width, sp), sp)
    assert dec.code_len == code_len and dec.layers[1][0].shape == (width, width) and dec.layers[3][0].shape[0] == width - code_len - 3
    pk = KE.debug_pack(dec.layers, dec.latent_in, dec.code_len)
    assert pk["code_len"] == code_len and pk["lat_tile"] == (29 if code_len == 32 else 27)
    rng = np.random.default_rng(2)
    code = (rng.normal(size=code_len) * 0.3).astype(np.float32)
    pts = rng.uniform(-0.8, 0.8, size=(16, 3)).astype(np.float32)
    sdf, grad = KE.run_wave(pk, code, pts, bwd=True)
    y, g = O.get_batch_sdf_jacobian(dec, code, pts)
    assert grad.shape == g.shape == (16, code_len + 3)
    assert np.abs(sdf - y).max() < 2e-6
    assert np.abs(grad - g).max() < 1e-5 * max(1.0, np.abs(g).max())
    assert np.abs(g[:, :code_len]).max() > 1e-5 and np.abs(g[:, code_len:]).max() > 1e-5
    assert np.abs(KE.run_wave(pk, code, pts, bwd=False) - y).max() < 2e-6
    # prepass stream of the same decoder
    import lp_emulator as LE
    from dsp_slam_amd import _lib as L
    lp = LE.debug_pack(pk["_holder"], L.PREPASS_F16)
    pts32 = rng.uniform(-0.8, 0.8, size=(32, 3)).astype(np.float32)
    got = LE.run_wave(pk, lp, code, pts32, L.PREPASS_F16)
    assert np.abs(got - LE.reference_forward(dec, code, pts32, L.PREPASS_F16)).max() < 5e-5
    assert np.abs(got - O.decode_sdf(dec, code, pts32)).max() < 2e-3


@pytest.mark.parametrize("depth,lat", [(6, 3), (7, 4), (4, 2)])
def test_other_depths(depth, lat):
    """Decoder.__init__ is generic over the number of hidden layers too (deep_sdf_decoder.py:27-47): 4, 6 and 7 hidden layers through
    the packed fp32 stream (forward + input gradient) and, for an even number of passes, the prepass stream; with an odd number the
    prepass is refused (its last-layer body reads slab X) and the library runs every sample through the fp32 kernel."""
    import copy
    import lp_emulator as LE
    from dsp_slam_amd import _lib as L
    sp = copy.deepcopy(fixtures.SPECS)
    sp["NetworkSpecs"].update(dims=[512] * depth, latent_in=[lat], norm_layers=list(range(depth)), dropout=list(range(depth)))
    dec = O.fold_decoder(fixtures.random_state_dict(5 + depth, sp), sp)
    assert len(dec.layers) == depth + 1
    pk = KE.debug_pack(dec.layers, dec.latent_in, dec.code_len)
    rng = np.random.default_rng(depth)
    code = (rng.normal(size=64) * 0.3).astype(np.float32)
    pts = rng.uniform(-0.8, 0.8, size=(16, 3)).astype(np.float32)
    sdf, grad = KE.run_wave(pk, code, pts, bwd=True)
    y, g = O.get_batch_sdf_jacobian(dec, code, pts)
    assert np.abs(sdf - y).max() < 2e-6 and np.abs(grad - g).max() < 1e-5 * max(1.0, np.abs(g).max())
    if depth % 2 == 0:
        lp = LE.debug_pack(pk["_holder"], L.PREPASS_F16)
        assert lp["passes"].shape[0] == depth
        pts32 = rng.uniform(-0.8, 0.8, size=(32, 3)).astype(np.float32)
        got = LE.run_wave(pk, lp, code, pts32, L.PREPASS_F16)
        assert np.abs(got - LE.reference_forward(dec, code, pts32, L.PREPASS_F16)).max() < 5e-5
    else:
        with pytest.raises(L.DspError):
            LE.debug_pack(pk["_holder"], L.PREPASS_F16)


@pytest.mark.parametrize("code_len", [64, 32])
def test_lp_jacobian_stream_emulated(code_len):
    """The low-precision compute mode's backward stream (dsp_debug_pack_lpj) replayed through a numpy model of mlp_lpj_fwd_kernel +
    mlp_lpj_bwd_kernel: its sdf IS the prepass emulation's (same passes, bit for bit), and its input gradient agrees with the oracle's fp32
    gradient to what rounding the weights, the activations and the back-propagated gradient to f16 once per layer allows -- which pins the
    transposed slot order, the mask bit bookkeeping, the re-injected rows of the latent_in layer (rows 445.. / 477..) and the first layer's
    row layout (code 0 .., xyz 77..79) on the CPU.  bf16 (8-bit mantissas) goes through the same data flow with a 16x looser bound."""
    import copy
    import lp_emulator as LE
    from dsp_slam_amd import _lib as L
    sp = copy.deepcopy(fixtures.SPECS)
    sp["CodeLength"] = code_len
    dec = O.fold_decoder(fixtures.random_state_dict(11, sp), sp)
    pk = KE.debug_pack(dec.layers, dec.latent_in, dec.code_len)
    rng = np.random.default_rng(code_len)
    code = (rng.normal(size=code_len) * 0.3).astype(np.float32)
    pts32 = rng.uniform(-0.8, 0.8, size=(32, 3)).astype(np.float32)
    y_ref, g_ref = O.get_batch_sdf_jacobian(dec, code, pts32)
    assert np.abs(g_ref[:, :code_len]).max() > 1e-4 and np.abs(g_ref[:, code_len:]).max() > 1e-4
    for dtype, tol in ((L.COMPUTE_F16, 1.0), (L.COMPUTE_BF16, 16.0)):
        lp = LE.debug_pack(pk["_holder"], dtype)
        lpj = LE.debug_pack_lpj(pk["_holder"], dtype)
        assert lpj["n_pass"] == 8 and list(lpj["passes"][:, 3]) == [3, 3, 3, 4, 3, 3, 3, 5] and list(lpj["passes"][:, 0]) == [8] * 7 + [2]
        assert lpj["lat_tile"] == (27 if code_len == 64 else 29) and lpj["chunks"] == 7 * 32 + 8
        y, g = LE.run_wave_jac(pk, lp, lpj, code, pts32, dtype)
        assert np.array_equal(y, LE.run_wave(pk, lp, code, pts32, dtype))
        g = np.concatenate([g[:, :code_len], g[:, 64:]], 1)
        scale = np.abs(g_ref).max()
        per_point = np.abs(g - g_ref).max(1) / scale
        print("emulated %s jacobian, %d-D codes: |dg| / max |g| per point: median %.2e, 90 %% %.2e, max %.2e" % (
            "f16" if dtype == L.COMPUTE_F16 else "bf16", code_len, np.median(per_point), np.percentile(per_point, 90), per_point.max()))
        # rounding alone: ~4e-4 (f16); a point whose 16-bit forward flips ONE relu unit that sits within round-off of zero (random weights have
        # many) differs by that unit's whole contribution -- a few points in 32, a few % at most
        assert np.median(per_point) < 1e-3 * tol and per_point.max() < (5e-2 if dtype == L.COMPUTE_F16 else 0.5), per_point
        if code_len == 32:
            gfull = LE.run_wave_jac(pk, lp, lpj, code, pts32, dtype)[1]
            assert not gfull[:, 32:64].any()          # code entries beyond the decoder's code length: zero gradient

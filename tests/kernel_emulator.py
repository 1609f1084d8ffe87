"""Numpy model of ONE WAVE of dsp_slam_amd/csrc/mlp_kernel.hip (16 points, 64 lanes).

Test infrastructure: it consumes the exact packed weight stream / pass table the library uploads
(dsp_debug_pack, host-only) and replays the kernel's register-level data flow -- slabs indexed
[reg][lane], v_mfma_f32_16x16x4_f32 operand/result lane maps, relu masks, skip-gradient capture,
final-layer dot product -- so the packing and the row conventions are checked without a GPU.

MFMA 16x16x4 lane maps (cdna_hip_programming.md section 3):  A[i = l&15][k = l>>4],  B[k = l>>4][j = l&15],
D[row = 4*(l>>4) + r][col = l&15] for accumulator register r.
"""
import ctypes as C

import numpy as np

from dsp_slam_amd import _lib as L

LANES = np.arange(64)
G = LANES >> 4
PL = LANES & 15


def debug_pack(layers, latent_in, code_len=64):
    lib = L.load()
    holder = L.DecoderDescHolder(layers, latent_in, code_len)
    slen, blen = C.c_int64(0), C.c_int64(0)
    meta = np.zeros(9, np.int32)
    L.check(lib.dsp_debug_pack(C.byref(holder.desc), None, C.byref(slen), None, C.byref(blen), None,
                               L.ptr(meta, L.c_i32p), None), None, "dsp_debug_pack(size)")
    stream = np.zeros(slen.value, np.float32)
    bias = np.zeros(blen.value, np.float32)
    passes = np.zeros((meta[1], 8), np.int32)
    b_last = C.c_float(0)
    L.check(lib.dsp_debug_pack(C.byref(holder.desc), L.ptr(stream), C.byref(slen), L.ptr(bias), C.byref(blen),
                               L.ptr(passes, L.c_i32p), L.ptr(meta, L.c_i32p), C.byref(b_last)), None, "dsp_debug_pack")
    def code_bias(code):
        out = np.zeros(1024, np.float32)
        L.check(lib.dsp_debug_code_bias(C.byref(holder.desc), L.ptr(L.code64(code)), L.ptr(out)), None, "dsp_debug_code_bias")
        return out

    return dict(stream=stream.reshape(-1, 16, 64, 4), bias=bias.reshape(-1, 512), passes=passes,
                n_fwd=int(meta[0]), n_pass=int(meta[1]), chunks_fwd=int(meta[2]), chunks_all=int(meta[3]),
                n_bias_rows=int(meta[4]), wlast_row=int(meta[5]), w0_row=int(meta[6]), lat_tile=int(meta[7]), code_len=int(meta[8]),
                b_last=float(b_last.value),
                code_bias=code_bias, _holder=holder)


def mfma16(a, b, acc):
    """a, b: (64,) per-lane operands; acc: (4, 64) [reg][lane].  Returns acc + A @ B in the D lane map."""
    A = np.zeros((16, 4), np.float64)
    B = np.zeros((4, 16), np.float64)
    A[PL, G] = a
    B[G, PL] = b
    D = A @ B                                     # (16 rows, 16 cols)
    out = acc.astype(np.float64).copy()
    for r in range(4):
        out[r] += D[4 * G + r, PL]
    return out


def run_wave(pk, code, pts16, bwd):
    """pts16: (16,3) object-frame points.  Returns sdf (16,), and if bwd grad (16, 67)."""
    sin = np.zeros((128, 64))
    masks = {}
    cb = pk["code_bias"](code).astype(np.float64)          # [0:512] layer 0, [512:1024] latent_in layer
    px, py, pz = pts16[PL, 0], pts16[PL, 1], pts16[PL, 2]
    w0 = pk["bias"][pk["w0_row"]:pk["w0_row"] + 3].astype(np.float64)      # W0[:, x|y|z]
    # layer 0 on the VALU: relu(code bias + W0[:, xyz] . p)
    for o in range(8):
        bits = np.zeros((16, 64), bool)
        for j in range(4):
            for r in range(4):
                row = 16 * (4 * o + j) + 4 * G + r
                pre = cb[row] + w0[0][row] * px + w0[1][row] * py + w0[2][row] * pz
                bits[4 * j + r] = pre > 0
                sin[16 * o + 4 * j + r] = np.maximum(pre, 0)
        masks[(0, o)] = bits
    skipc = np.zeros((16, 64))
    skipx = np.zeros((3, 64))
    gfirst = np.zeros(64)
    y = None
    chunk = 0
    n_pass = pk["n_pass"] if bwd else pk["n_fwd"]
    for ps in range(n_pass):
        nog, nchunks, bias_row, relu, mask_slot, kind, chunk_base, _ = pk["passes"][ps]
        assert chunk == chunk_base
        if kind == 2:
            g3 = G == 3
            r0 = 4 * pk["lat_tile"] + 1          # row 445 (tile 27) or 477 (tile 29, 32-D codes): row 13 of the tile
            sin[r0][g3], sin[r0 + 1][g3], sin[r0 + 2][g3] = px[g3], py[g3], pz[g3]
        elif bwd and kind == 5:
            part = np.zeros((3, 64))
            for t in range(32):
                for r in range(4):
                    row = 16 * t + 4 * G + r
                    for c3 in range(3):
                        part[c3] += w0[c3][row] * sin[4 * t + r]
            tot = np.zeros((3, 64))
            for l in range(64):
                tot[:, l] = part[:, l & 15] + part[:, (l & 15) + 16] + part[:, (l & 15) + 32] + part[:, (l & 15) + 48]
            gfirst = np.where(G == 0, tot[0], np.where(G == 1, tot[1], tot[2]))
        out = sin.copy()
        for og in range(nog):
            acc = np.zeros((4, 4, 64))           # [j][reg][lane]
            for j in range(4):
                for r in range(4):
                    row = 64 * og + 16 * j + 4 * G + r
                    if bias_row == -2:
                        acc[j][r] = cb[512 + row]
                    elif bias_row >= 0:
                        acc[j][r] = pk["bias"][bias_row][row]
            for c in range(nchunks):
                ch = pk["stream"][chunk]
                chunk += 1
                for s_ in range(16):
                    b = sin[16 * c + s_]
                    for j in range(4):
                        acc[j] = mfma16(ch[s_, :, j], b, acc[j])
            v = acc.reshape(16, 64)
            if relu:
                bits = (v > 0)
                v = np.maximum(v, 0)
                masks[(mask_slot, og)] = bits
            elif bwd and mask_slot >= 0:
                if kind == 4:
                    if pk["lat_tile"] != 29:
                        if og == 6:
                            skipx = v[13:16].copy()
                        if og == 7:
                            skipc = v.copy()
                    elif og == 7:
                        skipx = v[5:8].copy()
                        skipc = np.zeros((16, 64))
                        skipc[:8] = v[8:16]
                bits = masks.get((mask_slot, og))
                if bits is None:      # never written by the forward sweep (stale LDS in the kernel): must not matter
                    bits = np.ones((16, 64), bool)
                v = np.where(bits, v, 0.0)
            out[16 * og:16 * og + 16] = v
        sin = out
        if ps == pk["n_fwd"] - 1:
            wl = np.zeros((128, 64))
            for t in range(32):
                for r in range(4):
                    wl[4 * t + r] = pk["bias"][pk["wlast_row"]][16 * t + 4 * G + r]
            part = (sin * wl).sum(0)
            tot = np.zeros(64)
            for l in range(64):
                tot[l] = part[l & 15] + part[(l & 15) + 16] + part[(l & 15) + 32] + part[(l & 15) + 48]
            y = np.tanh(tot + pk["b_last"])
            if bwd:
                d = 1.0 - y * y
                for o in range(8):
                    # (groups a narrower decoder never produced read stale mask bits in the kernel: their final-layer weights are 0)
                    sin[16 * o:16 * o + 16] = np.where(masks.get((mask_slot, o), np.ones((16, 64), bool)), d * wl[16 * o:16 * o + 16], 0.0)
    sdf = y[:16]
    if not bwd:
        return sdf.astype(np.float32)
    grad = np.zeros((16, 67))
    for t in range(4):
        for r in range(4):
            grad[PL, 16 * t + 4 * G + r] = sin[4 * t + r] + skipc[4 * t + r]
    for cidx in range(3):
        lanes = np.where(G == cidx)[0]
        grad[PL[lanes], 64 + cidx] = gfirst[lanes] + skipx[cidx][(lanes & 15) + 48]
    cl = pk["code_len"]
    if cl != 64:            # the kernel's row is always [64 code slots | xyz]: slots beyond the code length must be zero
        assert np.all(grad[:, cl:64] == 0)
        grad = np.concatenate([grad[:, :cl], grad[:, 64:]], 1)
    return sdf.astype(np.float32), grad.astype(np.float32)

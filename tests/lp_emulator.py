"""Numpy model of ONE WAVE of dsp_slam_amd/csrc/mlp_lp_kernel.hip (32 points as two 16-point column blocks, 64 lanes): the low-precision
prepass.

Test infrastructure.  It consumes the exact packed 16-bit weight stream / pass table the library uploads
(dsp_debug_pack_prepass, host-only) and replays the kernel's register-level data flow: slabs indexed
[32-k step][column block][lane][8 halves], v_mfma_f32_16x16x32 operand / result lane maps, the split-precision xyz step,
fp32 bias as accumulator seed, relu + round-to-nearest-even packing into the next layer's slab, the final
fp32 dot product.  Lane maps (cdna_hip_programming.md section 3; round 5: 16x16x32, rounds 2-4 used 32x32x16):
  A[i = l & 15][k slot = 8 (l >> 4) + e],  B[k slot = 8 (l >> 4) + e][j = l & 15],
  D[row = 4 (l >> 4) + r][col = l & 15]  for accumulator register r in [0, 4).
(The hardware's k index of a slot is irrelevant: A and B use the same slot -> k map, and the sum over k does not care.)
"""
import ctypes as C

import numpy as np

from dsp_slam_amd import _lib as L

LANES = np.arange(64)
GQ = LANES >> 4
PL = LANES & 15
NCH = 4
KQ = 4              # 32-k steps per chunk
# dsp_internal.h LP_XYZ_TERMS: [dtype][k-step][term] = 4 * xpart + wpart, 0 = unused
XYZ_TERMS = {
    L.PREPASS_F16: [[5, 9, 6, 10, 0], [0, 0, 0, 0, 0]],
    L.PREPASS_BF16: [[5, 9, 13, 6, 10], [7, 14, 11, 0, 0]],
}


def lp_round(x, dtype):
    """fp32 -> f16 / bf16 -> fp32, round to nearest even."""
    x = np.ascontiguousarray(x, np.float32)
    if dtype == L.PREPASS_F16:
        return x.astype(np.float16).astype(np.float32)
    u = x.view(np.uint32).astype(np.uint64)
    u = (u + 0x7FFF + ((u >> 16) & 1)) >> 16
    return (u.astype(np.uint32) << 16).view(np.float32)


def lp_decode(bits, dtype):
    bits = np.ascontiguousarray(bits, np.uint16)
    if dtype == L.PREPASS_F16:
        return bits.view(np.float16).astype(np.float32)
    return (bits.astype(np.uint32) << 16).view(np.float32)


def debug_pack(holder, dtype):
    lib = L.load()
    slen = C.c_int64(0)
    meta = np.zeros(2, np.int32)
    L.check(lib.dsp_debug_pack_prepass(C.byref(holder.desc), dtype, None, C.byref(slen), None, L.ptr(meta, L.c_i32p)), None,
            "dsp_debug_pack_prepass(size)")
    stream = np.zeros(slen.value, np.uint16)
    passes = np.zeros((meta[0], 8), np.int32)
    L.check(lib.dsp_debug_pack_prepass(C.byref(holder.desc), dtype, stream.ctypes.data_as(C.POINTER(C.c_uint16)), C.byref(slen),
                                       L.ptr(passes, L.c_i32p), L.ptr(meta, L.c_i32p)), None, "dsp_debug_pack_prepass")
    # [chunk][step of 32 k: 4][16-row tile: 4][lane 64][8 halves]
    return dict(stream=lp_decode(stream, dtype).reshape(-1, KQ, 4, 64, 8), passes=passes, n_pass=int(meta[0]), chunks=int(meta[1]))


def mfma16(a, b, acc):
    """a, b: (64, 8) per-lane operands; acc (4, 64) [reg][lane] fp32.  D = A @ B + C in the lane maps above (fp64 sum, rounded once)."""
    A = np.zeros((16, 32), np.float64)
    B = np.zeros((32, 16), np.float64)
    for e in range(8):
        A[PL, 8 * GQ + e] = a[:, e]
        B[8 * GQ + e, PL] = b[:, e]
    D = A @ B
    out = acc.astype(np.float64).copy()
    for r in range(4):
        out[r] += D[4 * GQ + r, PL]
    return out.astype(np.float32)


def rows_in_d_order(tab512, g, rt):
    """fp32 table rows of group g, 16-row tile rt as [reg][lane]."""
    out = np.zeros((4, 64), np.float32)
    for r in range(4):
        out[r] = tab512[64 * g + 16 * rt + 4 * GQ + r]
    return out


def run_wave(pk_fp32, pk_lp, code, pts32, dtype):
    """pk_fp32: kernel_emulator.debug_pack(...) (bias table, code bias, b_last); pk_lp: debug_pack above.
    pts32: (32,3) object-frame points.  Returns the prepass sdf (32,)."""
    bias = pk_fp32["bias"]
    cb = pk_fp32["code_bias"](code)               # [0:512] layer 0, [512:1024] latent_in layer
    wl = bias[pk_fp32["wlast_row"]]
    pts32 = np.asarray(pts32, np.float32)
    xb = np.zeros((2, 64, 8), np.float32)          # [column block][lane][8 slots]
    for blk in range(2):
        p = pts32[16 * blk + PL]                   # (64, 3): the four lane groups of a point hold it
        xp = np.zeros((4, 64, 3), np.float32)
        xp[1] = lp_round(p, dtype)
        xp[2] = lp_round(p - xp[1], dtype)
        xp[3] = lp_round(p - xp[1] - xp[2], dtype)
        for e in range(8):
            kk = 8 * GQ + e                        # slot 0..31 = 16 u + 3 t + c
            for lane in range(64):
                u, k16 = kk[lane] >> 4, kk[lane] & 15
                t = k16 // 3
                ent = XYZ_TERMS[dtype][u][t] if t < 5 else 0
                xb[blk, lane, e] = xp[ent >> 2, lane, k16 % 3] if ent else 0.0
    slabs = [np.zeros((16, 2, 64, 8), np.float32), np.zeros((16, 2, 64, 8), np.float32)]   # X, Y: [step][block][lane][slot]
    part = np.zeros((2, 64), np.float32)
    for ps in range(pk_lp["n_pass"]):
        nog, nchunks, bias_row, kind, npad, last, chunk_base, _ = [int(v) for v in pk_lp["passes"][ps]]
        src, dst = (slabs[1], slabs[0]) if ps % 2 == 0 else (slabs[0], slabs[1])      # even passes read Y, write X
        if kind == 0:
            src[:KQ] = 0.0
            src[0] = xb
        elif kind == 2:
            kx = KQ * NCH - 1
            src[kx] = xb
            for t in range(1, npad + 1):           # padding 16-row tile 2 kx - t = half (T & 1) of step T >> 1
                T = 2 * kx - t
                src[T >> 1, :, :, 4 * (T & 1):4 * (T & 1) + 4] = 0.0
        tab = cb[512:] if bias_row == -2 else (cb[:512] if bias_row == -3 else bias[bias_row])
        for g in range(nog):
            acc = [[rows_in_d_order(tab, g, rt) for _ in range(2)] for rt in range(4)]
            for c in range(nchunks):
                chunk = pk_lp["stream"][chunk_base + g * nchunks + c]
                for kq in range(KQ):
                    ks = KQ * c + kq
                    for rt in range(4):
                        for blk in range(2):
                            acc[rt][blk] = mfma16(chunk[kq, rt], src[ks, blk], acc[rt][blk])
            for rt in range(4):
                T = 4 * g + rt
                for blk in range(2):
                    if last:
                        w = rows_in_d_order(wl, g, rt)
                        for r in range(4):
                            part[blk] = (part[blk] + np.maximum(acc[rt][blk][r], 0) * w[r]).astype(np.float32)
                    else:
                        v = lp_round(np.maximum(acc[rt][blk], 0.0), dtype)           # (4, 64)
                        for r in range(4):
                            dst[T >> 1, blk, :, 4 * (T & 1) + r] = v[r]
    out = np.zeros(32, np.float32)
    for blk in range(2):
        tot = part[blk][:16] + part[blk][16:32] + part[blk][32:48] + part[blk][48:]
        out[16 * blk:16 * blk + 16] = tot
    return np.tanh(out + np.float32(pk_fp32["b_last"])).astype(np.float32)


def debug_pack_lpj(holder, dtype):
    """The backward kernel's transposed stream + pass table (dsp_debug_pack_lpj, host only)."""
    lib = L.load()
    slen = C.c_int64(0)
    meta = np.zeros(3, np.int32)
    L.check(lib.dsp_debug_pack_lpj(C.byref(holder.desc), dtype, None, C.byref(slen), None, L.ptr(meta, L.c_i32p)), None, "dsp_debug_pack_lpj(size)")
    stream = np.zeros(slen.value, np.uint16)
    passes = np.zeros((meta[0], 8), np.int32)
    L.check(lib.dsp_debug_pack_lpj(C.byref(holder.desc), dtype, stream.ctypes.data_as(C.POINTER(C.c_uint16)), C.byref(slen), L.ptr(passes, L.c_i32p),
                                   L.ptr(meta, L.c_i32p)), None, "dsp_debug_pack_lpj")
    return dict(stream=lp_decode(stream, dtype).reshape(-1, KQ, 4, 64, 8), passes=passes, n_pass=int(meta[0]), chunks=int(meta[1]), lat_tile=int(meta[2]))


SEED_SCALE = np.float32(16.0)


def run_wave_jac(pk_fp32, pk_lp, pk_lpj, code, pts32, dtype):
    """Numpy model of ONE WAVE of mlp_lpj_fwd_kernel + mlp_lpj_bwd_kernel (dsp_slam_amd/csrc/mlp_lpj_kernel.hip): the prepass forward with its
    relu masks kept (bit = accumulator > 0), the backward sweep's input slab S w_last under the last layer's mask, eight passes over the
    transposed stream with the masks ANDed in before rounding, the latent_in pass's re-injected rows kept as rounded pairs, and the final
    (acc + kept) (1 - y^2) / S.  Returns (sdf (32,), grad (32, 67): d/dcode[64] then d/dxyz[3])."""
    bias = pk_fp32["bias"]
    cb = pk_fp32["code_bias"](code)
    wl = bias[pk_fp32["wlast_row"]]
    pts32 = np.asarray(pts32, np.float32)
    xb = np.zeros((2, 64, 8), np.float32)
    for blk in range(2):
        p = pts32[16 * blk + PL]
        xp = np.zeros((4, 64, 3), np.float32)
        xp[1] = lp_round(p, dtype)
        xp[2] = lp_round(p - xp[1], dtype)
        xp[3] = lp_round(p - xp[1] - xp[2], dtype)
        for e in range(8):
            kk = 8 * GQ + e
            for lane in range(64):
                u, k16 = kk[lane] >> 4, kk[lane] & 15
                t = k16 // 3
                ent = XYZ_TERMS[dtype][u][t] if t < 5 else 0
                xb[blk, lane, e] = xp[ent >> 2, lane, k16 % 3] if ent else 0.0
    slabs = [np.zeros((16, 2, 64, 8), np.float32), np.zeros((16, 2, 64, 8), np.float32)]
    part = np.zeros((2, 64), np.float32)
    masks = np.zeros((pk_lp["n_pass"], 32, 2, 4, 64), bool)       # [layer][tile][block][reg][lane]
    # ---- forward (run_wave), masks kept ----
    for ps in range(pk_lp["n_pass"]):
        nog, nchunks, bias_row, kind, npad, last, chunk_base, _ = [int(v) for v in pk_lp["passes"][ps]]
        src, dst = (slabs[1], slabs[0]) if ps % 2 == 0 else (slabs[0], slabs[1])
        if kind == 0:
            src[:KQ] = 0.0
            src[0] = xb
        elif kind == 2:
            kx = KQ * NCH - 1
            src[kx] = xb
            for t in range(1, npad + 1):
                T = 2 * kx - t
                src[T >> 1, :, :, 4 * (T & 1):4 * (T & 1) + 4] = 0.0
        tab = cb[512:] if bias_row == -2 else (cb[:512] if bias_row == -3 else bias[bias_row])
        for g in range(nog):
            acc = [[rows_in_d_order(tab, g, rt) for _ in range(2)] for rt in range(4)]
            for c in range(nchunks):
                chunk = pk_lp["stream"][chunk_base + g * nchunks + c]
                for kq in range(KQ):
                    ks = KQ * c + kq
                    for rt in range(4):
                        for blk in range(2):
                            acc[rt][blk] = mfma16(chunk[kq, rt], src[ks, blk], acc[rt][blk])
            for rt in range(4):
                T = 4 * g + rt
                for blk in range(2):
                    masks[ps, T, blk] = acc[rt][blk] > 0
                    if last:
                        w = rows_in_d_order(wl, g, rt)
                        for r in range(4):
                            part[blk] = (part[blk] + np.maximum(acc[rt][blk][r], 0) * w[r]).astype(np.float32)
                    else:
                        v = lp_round(np.maximum(acc[rt][blk], 0.0), dtype)
                        for r in range(4):
                            dst[T >> 1, blk, :, 4 * (T & 1) + r] = v[r]
    y = np.zeros(32, np.float32)
    for blk in range(2):
        tot = part[blk][:16] + part[blk][16:32] + part[blk][32:48] + part[blk][48:]
        y[16 * blk:16 * blk + 16] = tot
    y = np.tanh(y + np.float32(pk_fp32["b_last"])).astype(np.float32)
    # ---- backward ----
    X, Y = slabs[0], slabs[1]
    n_fwd = pk_lp["n_pass"]
    wls = (SEED_SCALE * wl).astype(np.float32)
    for T in range(32):
        ws = rows_in_d_order(wls, T >> 2, T & 3)                      # [reg][lane]
        for blk in range(2):
            v = lp_round(np.where(masks[n_fwd - 1, T, blk], ws, 0.0), dtype)
            for r in range(4):
                Y[T >> 1, blk, :, 4 * (T & 1) + r] = v[r]
    skip = {}
    src, dst = Y, X
    acc_final = None
    for ps in range(pk_lpj["n_pass"]):
        nog, nchunks, bias_row, kind, npad, last, chunk_base, _ = [int(v) for v in pk_lpj["passes"][ps]]
        layer = n_fwd - 1 - ps                                       # this pass goes back through hidden layer `layer`; its output is masked by layer - 1
        outs = {}
        for g in range(nog):
            acc = [[np.zeros((4, 64), np.float32) for _ in range(2)] for rt in range(4)]
            for c in range(nchunks):
                chunk = pk_lpj["stream"][chunk_base + g * nchunks + c]
                for kq in range(KQ):
                    ks = KQ * c + kq
                    for rt in range(4):
                        for blk in range(2):
                            acc[rt][blk] = mfma16(chunk[kq, rt], src[ks, blk], acc[rt][blk])
            for rt in range(4):
                T = 4 * g + rt
                for blk in range(2):
                    outs[(T, blk)] = acc[rt][blk]
                    if kind == 5:
                        continue
                    if kind == 4 and T >= 27:
                        skip[(T, blk)] = lp_round(acc[rt][blk], dtype)
                    v = lp_round(np.where(masks[layer - 1, T, blk], acc[rt][blk], 0.0), dtype)
                    for r in range(4):
                        dst[T >> 1, blk, :, 4 * (T & 1) + r] = v[r]
        if kind == 5:
            acc_final = outs
        src, dst = dst, src
    lt = pk_lpj["lat_tile"]
    grad = np.zeros((32, 67), np.float32)
    for blk in range(2):
        sc = ((np.float32(1.0) - y[16 * blk:16 * blk + 16] ** 2) * (np.float32(1.0) / SEED_SCALE)).astype(np.float32)      # per point (lane & 15)
        n_code_tiles = 31 - lt
        for j in range(4):
            g4 = acc_final[(j, blk)]                                                         # [reg][lane]: row 16 j + 4 gq + r
            k = skip[(lt + 1 + j, blk)] if j < n_code_tiles else np.zeros((4, 64), np.float32)
            val = ((g4 + k) * sc[PL][None, :]).astype(np.float32)
            for r in range(4):
                for lane in range(64):
                    grad[16 * blk + PL[lane], 16 * j + 4 * GQ[lane] + r] = val[r, lane]
        gx = acc_final[(4, blk)]
        kx = skip[(lt, blk)]
        val = ((gx + kx) * sc[PL][None, :]).astype(np.float32)
        for lane in range(48, 64):                                                           # lane group 3, registers 1..3
            for c in range(3):
                grad[16 * blk + PL[lane], 64 + c] = val[1 + c, lane]
    return y, grad


def reference_forward(dec, code, pts, dtype):
    """The prepass arithmetic stated directly on the folded decoder (no packing): layer 0 and the xyz / code columns of the
    latent_in layer at fp32 accuracy, hidden activations and hidden weights rounded to 16 bits, fp32 accumulation."""
    pts = np.asarray(pts, np.float32)
    n = pts.shape[0]
    x = np.concatenate([np.broadcast_to(np.asarray(code, np.float32), (n, dec.code_len)), pts], -1).astype(np.float64)
    h = None
    n_lin = len(dec.layers)
    for k, (w, b) in enumerate(dec.layers):
        w64 = w.astype(np.float64)
        if k == 0:
            a = x @ w64.T + b
        elif k == n_lin - 1:
            a = h @ w64.T + b
        elif k in dec.latent_in:
            p = w.shape[1] - dec.in_dim
            a = lp_round(h[:, :p], dtype).astype(np.float64) @ lp_round(w[:, :p], dtype).astype(np.float64).T + x @ w64[:, p:].T + b
        else:
            a = lp_round(h, dtype).astype(np.float64) @ lp_round(w, dtype).astype(np.float64).T + b
        h = np.maximum(a, 0.0).astype(np.float32) if k < n_lin - 1 else a
    return np.tanh(h[:, 0]).astype(np.float32)

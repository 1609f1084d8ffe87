"""ORACLE -- CPU restatement (numpy, float32) of DSP-SLAM's DeepSDF shape/pose Gauss-Newton path.

THIS IS TEST INFRASTRUCTURE, NOT PRODUCT CODE.  Only tests/, __graft_entry__.smoke() and the
`cpu_baseline` leg of bench.py may import it; the shipped path (dsp_slam_amd/) never does and
fails loudly when the HIP library is missing.

Every function restates one reference function and cites the file:line it follows (paths are
relative to the reference tree).  Parity status: PINNED -- the reference has no tests or golden
vectors of its own, so this oracle is pinned against outputs of the
UNMODIFIED reference Python run on CPU in the build container (tools/make_golden.py ->
tests/golden/*.npz; checked by tests/test_oracle_golden.py).

Arithmetic is float32 throughout, like the reference (torch default dtype).  Dense products go
through `_mm`, which uses a BLAS sgemm (torch CPU when importable -- ~6x faster than numpy's
bundled OpenBLAS on this image -- else numpy); that is a choice of BLAS, not of algorithm.
"""
import math
import os

import numpy as np

F32 = np.float32

_BLAS = os.environ.get("DSP_ORACLE_BLAS", "auto")
_torch = None
if _BLAS in ("auto", "torch"):
    try:
        import torch as _torch  # noqa: N812  (sgemm provider only)
    except Exception:  # pragma: no cover
        _torch = None


def _mm(a, b):
    """(n,k) @ (k,m) float32 sgemm."""
    a = np.ascontiguousarray(a, dtype=F32)
    b = np.ascontiguousarray(b, dtype=F32)
    if _torch is not None and a.shape[0] >= 64:
        return _torch.mm(_torch.from_numpy(a), _torch.from_numpy(b)).numpy()
    return a @ b


# ----------------------------------------------------------------------------------------------
# Decoder  (deep_sdf/deep_sdf_decoder.py:9-110, deep_sdf/workspace.py:202-223)
# ----------------------------------------------------------------------------------------------
class FoldedDecoder(object):
    """Weight-norm-folded DeepSDF MLP.  layers[k] = (W (out,in) f32, b (out,) f32)."""

    def __init__(self, layers, latent_in, code_len):
        self.layers = layers
        self.latent_in = tuple(latent_in)
        self.code_len = int(code_len)
        self.in_dim = self.code_len + 3


def fold_decoder(state_dict, specs):
    """state_dict keyed like Decoder.state_dict() (optional `module.` prefix), numpy or torch values.

    Folds nn.utils.weight_norm: W = g * v / ||v||_2 per output row (deep_sdf_decoder.py:49-54).
    Layer shapes follow deep_sdf_decoder.py:27-47 (layer k+1 in latent_in => out_dim -= dims[0]).
    """
    ns = specs["NetworkSpecs"]
    if ns.get("xyz_in_all") or ns.get("use_tanh") or ns.get("latent_dropout"):
        raise NotImplementedError("only the DSP-SLAM decoder configuration is restated")
    if not ns.get("weight_norm") and ns.get("norm_layers"):
        raise NotImplementedError("LayerNorm variant (weight_norm=False) not restated")
    sd = {}
    for k, v in state_dict.items():
        a = v.detach().cpu().numpy() if hasattr(v, "detach") else np.asarray(v)
        sd[k[7:] if k.startswith("module.") else k] = a.astype(F32)
    n_lin = len(ns["dims"]) + 1
    layers = []
    for k in range(n_lin):
        name = "lin%d" % k
        if name + ".weight_v" in sd:
            v = sd[name + ".weight_v"]
            g = sd[name + ".weight_g"].reshape(-1, 1)
            nrm = np.sqrt((v.astype(np.float64) ** 2).sum(axis=1, keepdims=True)).astype(F32)
            w = (v * (g / nrm)).astype(F32)  # torch._weight_norm: v * (g / norm)
        else:
            w = sd[name + ".weight"]
        layers.append((np.ascontiguousarray(w, dtype=F32), sd[name + ".bias"].astype(F32)))
    return FoldedDecoder(layers, ns["latent_in"], specs["CodeLength"])


def decoder_forward(dec, x, keep=False):
    """Decoder.forward (deep_sdf_decoder.py:75-110): x (N, code_len+3) -> y (N,).

    With keep=True also returns the pre-activations needed by the backward restatement.
    """
    x = np.ascontiguousarray(x, dtype=F32)
    h = x
    pre = []
    n_lin = len(dec.layers)
    for k, (w, b) in enumerate(dec.layers):
        if k in dec.latent_in:
            h = np.concatenate([h, x], axis=-1)              # :89-90
        a = _mm(h, w.T) + b                                   # :93
        if keep:
            pre.append(a)
        h = np.maximum(a, F32(0)) if k < n_lin - 1 else a    # :97-104 (relu; LayerNorm/dropout inert)
    y = np.tanh(h[:, 0]).astype(F32)                          # :107-108
    return (y, pre) if keep else y


def decoder_forward_backward(dec, x):
    """y and dy/dx -- what get_batch_sdf_jacobian's autograd call yields
    (reconstruct/loss_utils.py:82-103)."""
    x = np.ascontiguousarray(x, dtype=F32)
    y, pre = decoder_forward(dec, x, keep=True)
    n_lin = len(dec.layers)
    g = ((F32(1) - y * y)[:, None] * dec.layers[-1][0]).astype(F32)   # d tanh * W8 -> (N, 512)
    g_skip = np.zeros_like(x)
    for k in range(n_lin - 2, -1, -1):
        w = dec.layers[k][0]
        g = g * (pre[k] > 0)                                   # relu'(0) = 0
        g = _mm(g, w)                                          # (N, in_k)
        if k in dec.latent_in:
            g_skip = g_skip + g[:, -dec.in_dim:]
            g = g[:, :-dec.in_dim]
    return y, (g + g_skip).astype(F32)


def decode_sdf(dec, code, pts, max_batch=64 ** 3):
    """reconstruct/loss_utils.py:51-79 -- no-grad forward in chunks; code (C,), pts (N,3) -> (N,)."""
    pts = np.ascontiguousarray(pts, dtype=F32)
    out = []
    for head in range(0, pts.shape[0], max_batch):
        sub = pts[head:head + max_batch]
        x = np.concatenate([np.broadcast_to(code.astype(F32), (sub.shape[0], code.shape[0])), sub], -1)
        out.append(decoder_forward(dec, x))
    return np.concatenate(out, 0) if out else np.zeros((0,), F32)


def get_batch_sdf_jacobian(dec, code, pts):
    """reconstruct/loss_utils.py:82-103 -> y (N,), dy/d[code,xyz] (N, C+3)."""
    pts = np.ascontiguousarray(pts, dtype=F32)
    if pts.shape[0] == 0:
        return np.zeros((0,), F32), np.zeros((0, dec.in_dim), F32)
    x = np.concatenate([np.broadcast_to(code.astype(F32), (pts.shape[0], code.shape[0])), pts], -1)
    return decoder_forward_backward(dec, x)


# ----------------------------------------------------------------------------------------------
# Lie-group helpers  (reconstruct/loss_utils.py:107-233)
# ----------------------------------------------------------------------------------------------
def points_to_pose_jacobian_se3(p):
    """loss_utils.py:107-126: (N,3) -> (N,3,6) = [I | -[p]x]."""
    n = p.shape[0]
    j = np.zeros((n, 3, 6), F32)
    j[:, 0, 0] = j[:, 1, 1] = j[:, 2, 2] = 1
    x, y, z = p[:, 0], p[:, 1], p[:, 2]
    j[:, 0, 4] = z
    j[:, 0, 5] = -y
    j[:, 1, 3] = -z
    j[:, 1, 5] = x
    j[:, 2, 3] = y
    j[:, 2, 4] = -x
    return j


def points_to_pose_jacobian_sim3(p):
    """loss_utils.py:166-185: (N,3) -> (N,3,7) = [I | -[p]x | p]."""
    return np.concatenate([points_to_pose_jacobian_se3(p), p[:, :, None].astype(F32)], axis=-1)


# torch's float32 exp / sin / cos on the CPU (Sleef u10) return the correctly rounded value for 99 % / 95 % / 95 % of arguments; numpy's float32
# exp does so for only 60 %.  The difference matters: exp_sim3 forms c = (e^s - 1) / s, which amplifies the last bit of e^s by 1 / s (s = 1e-6:
# one ulp of e^s is 12 % of c).  Rounding the float64 value is the closest restatement of the reference's arithmetic that does not copy Sleef.
def _exp32(x):
    return F32(np.exp(np.float64(x)))


def _sin32(x):
    return F32(np.sin(np.float64(x)))


def _cos32(x):
    return F32(np.cos(np.float64(x)))


def _hat(w):
    return np.array([[0., -w[2], w[1]], [w[2], 0., -w[0]], [-w[1], w[0], 0.]], F32)


def exp_se3(x):
    """loss_utils.py:129-163 (float32, as torch on CPU)."""
    x = np.asarray(x, F32)
    v, w = x[:3], x[3:6]
    w_hat = _hat(w)
    w_hat2 = (w_hat @ w_hat).astype(F32)
    theta = F32(np.sqrt(np.sum(w * w, dtype=F32)))
    eye = np.eye(3, dtype=F32)
    if theta <= 1e-8:
        e_w, j = eye, eye
    else:
        s, c = _sin32(theta), _cos32(theta)
        t2, t3 = F32(theta ** 2), F32(theta ** 3)
        e_w = eye + w_hat * s / theta + w_hat2 * (F32(1.) - c) / t2
        k1 = (F32(1) - c) / t2
        k2 = (theta - s) / t3
        j = eye + k1 * w_hat + k2 * w_hat2
    rst = np.eye(4, dtype=F32)
    rst[:3, :3] = e_w
    rst[:3, 3] = (j.astype(F32) @ v).astype(F32)
    return rst


def exp_sim3(x):
    """loss_utils.py:188-233, quirks included: c = 0 when s <= 1e-8 in the theta > 1e-8 branch (:223)."""
    x = np.asarray(x, F32)
    v, w, s = x[:3], x[3:6], F32(x[6])
    w_hat = _hat(w)
    w_hat2 = (w_hat @ w_hat).astype(F32)
    theta = F32(np.sqrt(np.sum(w * w, dtype=F32)))
    t2 = F32(theta ** 2)
    sn, cs = _sin32(theta), _cos32(theta)
    e_s = _exp32(s)
    s2 = F32(s ** 2)
    eye = np.eye(3, dtype=F32)
    eps = 1e-8
    if theta <= 1e-8:
        e_w = eye
        if s == 0:
            j = eye
        else:
            j = ((e_s - F32(1.)) / s) * eye
    else:
        e_w = eye + w_hat * sn / theta + w_hat2 * (F32(1.) - cs) / t2
        a = e_s * sn
        b = e_s * cs
        c = F32(0.) if s <= eps else (e_s - F32(1.)) / s
        k0 = c * eye
        k1 = (a * s + (F32(1) - b) * theta) / (s2 + t2)
        k2 = c - ((b - F32(1)) * s + a * theta) / (s2 + t2)
        j = k0 + k1 * w_hat / theta + k2 * w_hat2 / t2
    rst = np.eye(4, dtype=F32)
    rst[:3, :3] = (e_s * e_w).astype(F32)
    rst[:3, 3] = (j.astype(F32) @ v).astype(F32)
    return rst


def _inv(a):
    """torch.inverse on CPU float32 == LAPACK sgetrf-based inverse; numpy float32 inv is the same class."""
    return np.linalg.inv(np.asarray(a, F32)).astype(F32)


def _det3_cuberoot(r):
    """torch.det(R) ** (1/3) in float32 (optimizer.py:122, loss.py:163)."""
    d = F32(np.linalg.det(np.asarray(r, F32)))
    return F32(np.power(d, F32(1.0 / 3.0)))


def linspace_f32(start, end, steps):
    """torch.linspace on CPU float32: step = (end-start)/(steps-1); first half start + i*step,
    second half end - (steps-1-i)*step (ATen RangeFactories linspace kernel)."""
    start, end = F32(start), F32(end)
    step = F32((end - start) / F32(steps - 1))
    i = np.arange(steps)
    lo = (start + step * i.astype(F32)).astype(F32)
    hi = (end - step * (steps - 1 - i).astype(F32)).astype(F32)
    return np.where(i < steps // 2, lo, hi).astype(F32)


# ----------------------------------------------------------------------------------------------
# Robust kernel  (reconstruct/loss_utils.py:236-265)
# ----------------------------------------------------------------------------------------------
def huber_norm_weights(x, b):
    """loss_utils.py:236-248: x = |r| (N,), w = sqrt(rho(x)) / x, w(0) = 0."""
    x = np.asarray(x, F32).copy()
    b = F32(b)
    rho = np.where(x <= b, x * x, F32(2) * b * x - b * b).astype(F32)
    x[x == 0] = 1.
    return (np.sqrt(rho) / x).astype(F32)


def get_robust_res(res, b):
    """loss_utils.py:251-265 -> (w*res, mean((w*res)^2), w).  Empty input => loss NaN, like torch.mean."""
    res = np.asarray(res, F32).reshape(-1)
    w = huber_norm_weights(np.abs(res), b)
    rr = (w * res).astype(F32)
    loss = F32(np.mean(rr * rr, dtype=F32)) if rr.size else F32(np.nan)
    return rr, loss, w


# ----------------------------------------------------------------------------------------------
# Residual terms  (reconstruct/loss.py)
# ----------------------------------------------------------------------------------------------
def transform_points(t_obj_cam, p):
    """(p[..., None, :] * R).sum(-1) + t  (loss.py:31-32, 62-63), float32, products summed in column order."""
    r = np.asarray(t_obj_cam, F32)[:3, :3]
    t = np.asarray(t_obj_cam, F32)[:3, 3]
    p = np.asarray(p, F32)
    out = np.empty(p.shape, F32)
    for i in range(3):
        acc = (p[..., 0] * r[i, 0]).astype(F32)
        acc = (acc + (p[..., 1] * r[i, 1]).astype(F32)).astype(F32)
        acc = (acc + (p[..., 2] * r[i, 2]).astype(F32)).astype(F32)
        out[..., i] = acc + t[i]
    return out


def compute_sdf_loss(dec, pts_surface_cam, t_obj_cam, code):
    """loss.py:22-43 -> J_pose (N,7), J_code (N,C), residual (N,)."""
    p_o = transform_points(t_obj_cam, pts_surface_cam)
    res, de_di = get_batch_sdf_jacobian(dec, code, p_o)
    de_dxo = de_di[:, -3:]
    dxo = points_to_pose_jacobian_sim3(p_o)
    jac_toc = np.einsum("ni,nij->nj", de_dxo, dxo).astype(F32)
    return jac_toc, de_di[:, :-3], res


def sdf_to_occupancy(sdf, th):
    """loss_utils.py:40-48."""
    th = F32(th)
    return (F32(0.5) - np.clip(sdf, -th, th) / (F32(2) * th)).astype(F32)


def compute_render_loss(dec, ray_directions, depth_obs, t_obj_cam, sampled_ray_depth, code, th=0.01,
                        stats=None, sdf_jitter=0.0):
    """loss.py:46-152 -> (J_pose (K,7), J_code (K,C), residual (K,)) or None (<10 in-sphere samples).

    stats (optional dict) receives the ragged set sizes V, m, K and the index sets.
    """
    rays = np.asarray(ray_directions, F32)
    d = np.asarray(sampled_ray_depth, F32)
    th = F32(th)
    n_rays, n_d = rays.shape[0], d.shape[0]
    pts_cam = (rays[:, None, :] * d[None, :, None]).astype(F32)          # :60
    pts_obj = transform_points(t_obj_cam, pts_cam)                          # :62-63
    nrm = np.sqrt(np.sum(pts_obj * pts_obj, axis=-1, dtype=F32)).astype(F32)
    vx, vy = np.where(nrm < F32(1.0))                                       # :68 (row-major order)
    query = pts_obj[vx, vy, :]
    if stats is not None:
        stats["V"] = int(query.shape[0])
    if query.shape[0] < 10:                                                 # :73-74
        return None
    sdf = decode_sdf(dec, code, query)                                      # :77-78
    if sdf_jitter:   # tests only: +-jitter on the decoded values, to measure how round-off in the decoder propagates
        sdf = (sdf + F32(sdf_jitter) * np.where(np.arange(sdf.shape[0]) % 2 == 0, F32(1), F32(-1))).astype(F32)
    occ = np.zeros((n_rays, n_d), F32)
    occ[vx, vy] = sdf_to_occupancy(sdf, th)                                 # :84-86
    wg = (sdf > -th) & (sdf < th)                                           # :88
    gx, gy = vx[wg], vy[wg]
    occ_g = occ[gx, :]                                                      # (m, D)  :93
    m = occ_g.shape[0]
    d_min, d_max = d[0], d[-1]
    acc = np.cumprod(F32(1) - occ_g, axis=-1, dtype=F32)                    # :99
    acc_aug = np.concatenate([np.ones((m, 1), F32), acc], -1)
    o = np.concatenate([occ_g, np.ones((m, 1), F32)], -1)
    dd = np.concatenate([d, np.array([F32(1.1) * d_max], F32)], -1)
    term = (o * acc_aug).astype(F32)
    d_u = np.sum(dd * term, axis=-1, dtype=F32)                             # :112-114
    o_k = occ[gx, gy]
    l_idx = np.arange(n_d)[None, :]
    acc_z = np.where(l_idx < gy[:, None], F32(0), acc)                      # :121
    de_do = (np.sum(acc_z, axis=-1, dtype=F32) / (F32(1.) - o_k)).astype(F32)
    nz = de_do > F32(1e-2)                                                  # :125
    if stats is not None:      # forensics: everything a threshold decision was made on (tests name flipped samples by these)
        stats.update(norm=nrm, band=(gx.copy(), gy.copy()), de_do_band=de_do.copy())
    de_do = de_do[nz]
    d_u = d_u[nz]
    delta_d = F32((d_max - d_min) / F32(n_d - 1))
    do_ds = F32(-1.) / (F32(2) * th)
    de_ds = (de_do * delta_d * do_ds).astype(F32)                           # :130
    gx, gy = gx[nz], gy[nz]
    res = (np.asarray(depth_obs, F32)[gx] - d_u).astype(F32)                # :135-136
    res = np.clip(res, F32(-0.30), F32(0.30))                               # :139-140
    pts_g = pts_obj[gx, gy]
    _, ds_di = get_batch_sdf_jacobian(dec, code, pts_g)                     # :144
    de_di = (de_ds[:, None] * ds_di).astype(F32)
    dxo = points_to_pose_jacobian_sim3(pts_g)
    jac_toc = np.einsum("ni,nij->nj", de_di[:, -3:], dxo).astype(F32)
    if stats is not None:
        stats.update(m=int(m), K=int(gx.shape[0]), valid=(vx, vy), kept=(gx, gy), sdf=sdf, de_ds=de_ds)
    return jac_toc, de_di[:, :-3], res


def compute_rotation_loss_sim3(t_obj_cam):
    """loss.py:155-178 (CPU float32) -> (J (7,), residual)."""
    t_cam_obj = _inv(t_obj_cam)
    r_co = t_cam_obj[:3, :3].copy()
    scale = _det3_cuberoot(r_co)
    r_co = (r_co / scale).astype(F32)
    r_oc = _inv(r_co)
    ey = np.array([0., 1., 0.], F32)
    ng = np.array([0., -1., 0.], F32)
    ry = (r_co @ ey).astype(F32)
    res = F32(1.) - F32(np.dot(ry, ng))
    if res < 1e-7:
        return np.zeros(7, F32), F32(0.)
    j = np.zeros(7, F32)
    j[3:6] = np.cross((r_oc @ ng).astype(F32), ey).astype(F32)
    return j, res


# ----------------------------------------------------------------------------------------------
# Optimiser  (reconstruct/optimizer.py)
# ----------------------------------------------------------------------------------------------
class GNParams(object):
    """Hyper-parameters read by Optimizer.__init__ (optimizer.py:27-43)."""

    def __init__(self, k1=1.0, k2=100.0, k3=0.25, k4=1e7, b1=0.2, b2=0.025, lr=1.0, s_damp=1.0,
                 num_iterations=10, code_len=64, num_depth_samples=50, cut_off=0.01,
                 num_iterations_pose_only=5):
        self.k1, self.k2, self.k3, self.k4 = k1, k2, k3, k4
        self.b1, self.b2, self.lr, self.s_damp = b1, b2, lr, s_damp
        self.num_iterations = num_iterations
        self.code_len = code_len
        self.num_depth_samples = num_depth_samples
        self.cut_off = cut_off
        self.num_iterations_pose_only = num_iterations_pose_only

    @classmethod
    def from_configs(cls, cfg):
        o = cfg["optimizer"]
        j = o["joint_optim"]
        p = o.get("pose_only_optim", {"num_iterations": 5})
        return cls(j["k1"], j["k2"], j["k3"], j["k4"], j["b1"], j["b2"], j["learning_rate"],
                   j["scale_damping"], j["num_iterations"], o["code_len"], o["num_depth_samples"],
                   o["cut_off_threshold"], p["num_iterations"])


def set_checksum(ray_idx, depth_idx):
    """Order-independent checksum of a sample set, same definition as the device's (include/dsp_gn.h, dsp_batch_trace)."""
    ids = (np.asarray(ray_idx, np.uint64) << np.uint64(6)) | np.asarray(depth_idx, np.uint64)
    ids = ids & np.uint64(0xFFFFFFFF)
    h = ((ids * np.uint64(2654435761)) & np.uint64(0xFFFFFFFF)) ^ (ids >> np.uint64(7))
    return int(h.sum() & np.uint64(0xFFFFFFFF)) if ids.size else 0


def _gram(j, r):
    """sum_n J_n^T J_n and sum_n J_n^T r_n for row Jacobians (optimizer.py:162-167)."""
    j = np.ascontiguousarray(j, F32)
    return (j.T @ j).astype(F32), (j.T @ np.asarray(r, F32)).astype(F32)


def reconstruct_object(dec, prm, t_cam_obj, pts, rays, depth, code=None, trace=None, t_obj_cam0=None, sdf_jitter=0.0,
                       scale_jitter=0.0, sampled_override=None):
    """Optimizer.reconstruct_object (optimizer.py:88-203).

    Returns dict(t_cam_obj (4,4) f32 | None, code (C,) f32 | None, is_good bool, loss float).
    trace (optional list) receives one dict per GN iteration.
    """
    c_len = prm.code_len
    z = np.zeros(c_len, F32) if code is None else np.asarray(code, F32)[:c_len].copy()
    # t_obj_cam0 (tests only): start from a given camera->object matrix instead of inverting t_cam_obj
    t_obj_cam = _inv(np.asarray(t_cam_obj, F32)) if t_obj_cam0 is None else np.asarray(t_obj_cam0, F32).copy()
    rays = np.asarray(rays, F32)
    depth = np.asarray(depth, F32)
    n_fg = depth.shape[0]
    n_bg = rays.shape[0] - n_fg
    depth_obs = np.concatenate([depth, np.zeros(n_bg, F32)]).astype(F32)
    pts = np.asarray(pts, F32)
    loss = 0.
    fail = lambda: dict(t_cam_obj=None, code=None, is_good=False, loss=loss)  # noqa: E731
    for e in range(prm.num_iterations):
        t_co = _inv(t_obj_cam)                                               # :120
        scale = _det3_cuberoot(t_co[:3, :3])                                 # :122
        if scale_jitter:   # tests only: the derived scale moved by a relative round-off (fp32 det/pow vs other exact paths)
            scale = F32(scale * F32(1.0 + scale_jitter))
        d_min = F32(t_co[2, 3] - F32(1.0) * scale)
        d_max = F32(t_co[2, 3] + F32(1.0) * scale)
        sampled = linspace_f32(d_min, d_max, prm.num_depth_samples)          # :125
        if sampled_override is not None:   # tests only: linearise on exactly these depth samples (one row per iteration, or one row for a single-iteration run)
            so = np.asarray(sampled_override, F32)
            sampled = (so[e] if so.ndim == 2 else so)[:prm.num_depth_samples].copy()
            d_max = sampled[-1]
        derived_depths = sampled.copy()
        depth_obs[n_fg:] = F32(1.1) * d_max                                  # :126
        j7_s, jc_s, r_s = compute_sdf_loss(dec, pts, t_obj_cam, z)           # :129
        rr_s, sdf_loss, _ = get_robust_res(r_s, prm.b2)                      # :134
        if math.isnan(sdf_loss):
            return fail()
        st = {}
        rend = compute_render_loss(dec, rays, depth_obs, t_obj_cam, sampled, z, th=prm.cut_off, stats=st, sdf_jitter=sdf_jitter)
        if rend is None:                                                     # :142-143
            return fail()
        j7_r, jc_r, r_r = rend
        rr_r, render_loss, _ = get_robust_res(r_r, prm.b1)                   # :148
        if math.isnan(render_loss):
            return fail()
        j_rot, res_rot = compute_rotation_loss_sim3(t_obj_cam)               # :153
        loss = float(F32(prm.k1) * render_loss + F32(prm.k2) * sdf_loss)     # :155
        pd = 7
        j_s = np.concatenate([j7_s, jc_s], -1)
        hs, bs = _gram(j_s, rr_s)
        h_sdf = (F32(prm.k2) * hs / F32(j_s.shape[0])).astype(F32)           # :162
        b_sdf = (-F32(prm.k2) * bs / F32(j_s.shape[0])).astype(F32)
        j_r = np.concatenate([j7_r, jc_r], -1)
        hr, br = _gram(j_r, rr_r)
        h_r = (F32(prm.k1) * hr / F32(j_r.shape[0])).astype(F32)             # :166
        b_r = (-F32(prm.k1) * br / F32(j_r.shape[0])).astype(F32)
        h = (h_r + h_sdf).astype(F32)
        h[pd:, pd:] += F32(prm.k3) * np.eye(c_len, dtype=F32)                # :170
        b = (b_r + b_sdf).astype(F32)
        b[pd:] -= F32(prm.k3) * z                                            # :172
        h_rot = np.outer(j_rot, j_rot).astype(F32)                           # :176
        b_rot = -(j_rot * res_rot).astype(F32)                               # :177
        h[:pd, :pd] += F32(prm.k4) * h_rot
        b[:pd] -= F32(prm.k4) * b_rot                                        # :179 (sign as written)
        h[:pd, :pd] += np.eye(pd, dtype=F32)                                 # :183
        h[pd - 1, pd - 1] += F32(prm.s_damp)                                 # :184
        dx = (_inv(h) @ b).astype(F32)                                       # :186
        delta_t = exp_sim3(F32(prm.lr) * dx[:pd])                            # :190
        if trace is not None:
            trace.append(dict(V=st["V"], m=st["m"], K=st["K"], vsum=set_checksum(*st["valid"]), ksum=set_checksum(*st["kept"]), sets=st,
                              H=h.copy(), b=b.copy(), dx=dx.copy(), depths=derived_depths,
                              t_obj_cam=t_obj_cam.copy(), code=z.copy(), loss=loss,
                              sdf_loss=float(sdf_loss), render_loss=float(render_loss)))
        t_obj_cam = (delta_t @ t_obj_cam).astype(F32)                        # :191
        z = (z + F32(prm.lr) * dx[pd:pd + c_len]).astype(F32)                # :192
    return dict(t_cam_obj=_inv(t_obj_cam), code=z, is_good=True, loss=loss)


def estimate_pose_cam_obj(dec, prm, t_co_se3, scale, pts, code, trace=None):
    """Optimizer.estimate_pose_cam_obj (optimizer.py:45-86) -> (4,4) f32 SE(3) object->camera.

    (The reference scales the caller's array in place, :53-54; this restatement works on a copy.)
    """
    t_cam_obj = np.asarray(t_co_se3, F32).copy()
    t_cam_obj[:3, :3] *= F32(scale)
    t_obj_cam = _inv(t_cam_obj)
    z = np.asarray(code, F32)
    pts = np.asarray(pts, F32)
    for e in range(prm.num_iterations_pose_only):
        j7, _, res = compute_sdf_loss(dec, pts, t_obj_cam, z)
        j6 = j7[:, :6]
        n = F32(j6.shape[0])
        hess = ((j6.T @ j6) / n).astype(F32) + F32(1e-2) * np.eye(6, dtype=F32)   # :69-70
        b = (-(j6.T @ res) / n).astype(F32)                                       # :71 raw residual
        dx = (_inv(hess) @ b).astype(F32)
        if trace is not None:
            trace.append(dict(H=hess.copy(), b=b.copy(), dx=dx.copy(), t_obj_cam=t_obj_cam.copy()))
        t_obj_cam = (exp_se3(dx) @ t_obj_cam).astype(F32)                         # :73-74
        if e == 4:                                                                # :76-78
            keep = np.abs(res) <= F32(0.05)
            pts = pts[keep]
    out = _inv(t_obj_cam)
    out[:3, :3] /= F32(scale)
    return out

"""CPU baseline for bench.py: the reference's PyTorch op sequence for one reconstruct_object call, restated from scratch.

TEST / MEASUREMENT INFRASTRUCTURE ONLY -- nothing under dsp_slam_amd/ imports this; only bench.py's `cpu_baseline` leg and
tests/ do.  The GPU box has no checkout of the reference, so the baseline that travels has to be a restatement; the numpy oracle
(oracle/dsp_oracle.py) is written for checking, not for speed, and runs ~3x slower than the reference itself (VERDICT round 2,
weak #5).  This file instead issues the SAME TORCH OPERATIONS the reference issues, so that its wall time on a host is the
reference's wall time on that host:

  * the decoder is an nn.Module of weight-normed nn.Linear layers in eval mode whose parameters still require grad
    (deep_sdf/workspace.py:213-221 never freezes them), so every jacobian call pays for the weight gradients autograd
    accumulates and nobody reads (reconstruct/loss_utils.py:82-103);
  * decode_sdf = expand + cat + one no-grad forward over all in-sphere samples (loss_utils.py:51-79);
  * the render term builds the same (m, D) cumprod / masked-sum tensors (reconstruct/loss.py:84-141);
  * the normal equations are bmm(J^T, J).sum(0) over (N, 71, 1) x (N, 1, 71) batches (reconstruct/optimizer.py:159-171) and
    torch.inverse of the 71 x 71 system (:186).

tools/calibrate_cpu_baseline.py times it against the unmodified reference (oracle/ref_shim.py) in the build container and commits
the ratio under profiles/; bench.py reports `kind: "port"` (`port: "torch-restatement ..."`) with that ratio.  tests/test_torch_baseline.py checks that
its result agrees with the numpy oracle's (same algorithm, same fixture).
"""
import math

import numpy as np
import torch
import torch.nn as nn


class BaselineDecoder(nn.Module):
    """deep_sdf/deep_sdf_decoder.py:10-110 for the DeepSDF spec DSP-SLAM ships (weight norm, no layer norm, dropout inert in eval)."""

    def __init__(self, code_len, dims, latent_in, norm_layers, weight_norm=True):
        super().__init__()
        widths = [code_len + 3] + list(dims) + [1]
        self.n_lin = len(widths) - 1
        self.latent_in = tuple(latent_in)
        for k in range(self.n_lin):
            n_out = widths[k + 1] - widths[0] if (k + 1) in self.latent_in else widths[k + 1]
            lin = nn.Linear(widths[k], n_out)
            if weight_norm and k in norm_layers:
                lin = nn.utils.weight_norm(lin)
            self.add_module("lin%d" % k, lin)

    def forward(self, inp):
        h = inp
        for k in range(self.n_lin):
            if k in self.latent_in:
                h = torch.cat([h, inp], dim=-1)
            h = getattr(self, "lin%d" % k)(h)
            if k < self.n_lin - 1:
                h = torch.relu(h)
        return torch.tanh(h)


def build_decoder(state_dict, specs):
    """state_dict keyed as Decoder.state_dict() (numpy or torch values).  Parameters are left requiring grad, as the reference leaves them."""
    import warnings
    ns = specs["NetworkSpecs"]
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        dec = BaselineDecoder(specs["CodeLength"], ns["dims"], ns["latent_in"], ns["norm_layers"], ns["weight_norm"])
    dec.load_state_dict({k: torch.as_tensor(np.asarray(v)) for k, v in state_dict.items()})
    dec.eval()
    return dec


def _forward_all(dec, z, x, chunk=64 ** 3):
    outs = []
    with torch.no_grad():
        for lo in range(0, x.shape[0], chunk):
            xs = x[lo:lo + chunk, :3]
            outs.append(dec(torch.cat([z.expand(xs.shape[0], -1), xs], dim=-1)).squeeze())
    return torch.cat([o.reshape(-1) for o in outs], 0)


def _input_jacobian(dec, z, x):
    n = x.shape[0]
    inp = torch.cat([z.expand(n, -1), x.clone().detach()], 1).unsqueeze(1).repeat(1, 1, 1)
    inp.requires_grad = True
    y = dec(inp)
    y.backward(torch.eye(1).view(1, 1, 1).repeat(n, 1, 1), retain_graph=False)
    return y.detach(), inp.grad.data.detach()


def _pose_jacobian_sim3(p):
    n = p.shape[0]
    zero = torch.zeros(n)
    x, y, z = p[:, 0], p[:, 1], p[:, 2]
    skew = torch.stack([torch.stack([zero, -z, y], dim=-1), torch.stack([z, zero, -x], dim=-1), torch.stack([-y, x, zero], dim=-1)], dim=-1)
    return torch.cat((torch.eye(3).view(1, 3, 3).repeat(n, 1, 1), skew, p[..., None]), dim=-1)


def _to_object(t_oc, p):
    return (p[..., None, :] * t_oc[:3, :3]).sum(-1) + t_oc[:3, 3]


def _surface_term(dec, pts_cam, t_oc, z):
    p_o = _to_object(t_oc, pts_cam)
    res, g = _input_jacobian(dec, z, p_o)
    return torch.bmm(g[..., -3:], _pose_jacobian_sim3(p_o)), g[..., :-3], res


def _render_term(dec, rays, depth_obs, t_oc, depths, z, th):
    p_o = _to_object(t_oc, rays[..., None, :] * depths[:, None])
    n_rays, n_d = p_o.shape[0], depths.shape[0]
    vr, vd = torch.where(torch.norm(p_o, dim=-1) < 1.0)
    q = p_o[vr, vd, :]
    if q.shape[0] < 10:
        return None
    sdf = _forward_all(dec, z, q)
    occ = torch.full((n_rays, n_d), 0.)
    occ[vr, vd] = 0.5 - torch.clamp(sdf, min=-th, max=th) / (2 * th)
    band = (sdf > -th) & (sdf < th)
    br, bd = vr[band], vd[band]
    rows = occ[br, :]
    m = rows.shape[0]
    trans = torch.cumprod(1 - rows, dim=-1)
    trans_aug = torch.cat((torch.ones(m, 1), trans), dim=-1)
    o_aug = torch.cat((rows, torch.ones(m, 1)), dim=-1)
    d_aug = torch.cat((depths, torch.tensor([1.1 * depths[-1]])), dim=-1)
    prob = o_aug * trans_aug
    d_u = torch.sum(d_aug * prob, dim=-1)
    _ = torch.sum(prob * (d_aug[None, :] - d_u[:, None]) ** 2, dim=-1)      # the reference computes the variance and drops it (loss.py:115)
    o_k = occ[br, bd]
    idx = torch.arange(n_d)[None, :].repeat(m, 1)
    trans[idx < bd[:, None]] = 0.
    de_do = trans.sum(dim=-1) / (1. - o_k)
    keep = de_do > 1e-2
    de_ds = (de_do[keep] * ((depths[-1] - depths[0]) / (n_d - 1)) * (-1. / (2 * th))).view(-1, 1, 1)
    br, bd = br[keep], bd[keep]
    res = depth_obs[br] - d_u[keep]
    res[res > 0.30] = 0.30
    res[res < -0.30] = -0.30
    pk = p_o[br, bd]
    _, g = _input_jacobian(dec, z, pk)
    g = de_ds * g
    return torch.bmm(g[..., -3:], _pose_jacobian_sim3(pk)), g[..., :-3], res.view(-1, 1, 1)


def _robust(res, b):
    res = res.view(-1, 1, 1)
    a = torch.abs(res)
    rho = torch.zeros_like(a)
    rho[a <= b] = a[a <= b] ** 2
    rho[a > b] = 2 * b * a[a > b] - b ** 2
    a[a == 0] = 1.
    r = torch.sqrt(rho) / a * res
    return r, torch.mean(r ** 2)


def _rotation_prior(t_oc):
    t_co = torch.inverse(t_oc)
    r_co = t_co[:3, :3]
    r_co = r_co / torch.det(r_co) ** (1 / 3)
    res = 1. - torch.dot(torch.mv(r_co, torch.tensor([0., 1., 0.])), torch.tensor([0., -1., 0.]))
    j = torch.zeros(7)
    if res < 1e-7:
        return j, 0.
    j[3:6] = torch.linalg.cross(torch.mv(torch.inverse(r_co), torch.tensor([0., -1., 0.])), torch.tensor([0., 1., 0.]))
    return j, res


def _exp_sim3(x):
    v, w, s = x[:3], x[3:6], x[6]
    wh = torch.tensor([[0., -w[2], w[1]], [w[2], 0., -w[0]], [-w[1], w[0], 0.]])
    wh2 = torch.mm(wh, wh)
    th = torch.norm(w)
    es, eye = torch.exp(s), torch.eye(3)
    if th <= 1e-8:
        rot = eye
        jac = eye if s == 0 else (es - 1.) / s * eye
    else:
        rot = eye + wh * torch.sin(th) / th + wh2 * (1. - torch.cos(th)) / th ** 2
        a, b = es * torch.sin(th), es * torch.cos(th)
        c = 0. if s <= 1e-8 else (es - 1.) / s        # the reference's quirk (loss_utils.py:223)
        jac = c * eye + (a * s + (1 - b) * th) / (s ** 2 + th ** 2) * wh / th + (c - ((b - 1) * s + a * th) / (s ** 2 + th ** 2)) * wh2 / th ** 2
    out = torch.eye(4)
    out[:3, :3] = es * rot
    out[:3, 3] = torch.mv(jac, v)
    return out


def reconstruct_object(dec, prm, t_cam_obj, pts, rays, depth, code=None):
    """reconstruct/optimizer.py:88-203 on the CPU.  prm: anything with k1..k4, b1, b2, lr, s_damp, num_iterations, code_len,
    num_depth_samples, cut_off (oracle.dsp_oracle.GNParams).  Returns dict(t_cam_obj, code, is_good, loss)."""
    n_code = prm.code_len
    z = torch.zeros(n_code) if code is None else torch.from_numpy(np.ascontiguousarray(code[:n_code], np.float32)).clone()
    t_oc = torch.inverse(torch.from_numpy(np.array(t_cam_obj, np.float32)))
    dirs = torch.from_numpy(np.ascontiguousarray(rays, np.float32))
    n_fg = depth.shape[0]
    depth_obs = torch.from_numpy(np.concatenate([depth, np.zeros(rays.shape[0] - n_fg)]).astype(np.float32))
    pts_t = torch.from_numpy(np.ascontiguousarray(pts, np.float32))
    bad = dict(t_cam_obj=None, code=None, is_good=False, loss=0.)
    loss = 0.
    for _ in range(prm.num_iterations):
        t_co = torch.inverse(t_oc)
        scale = torch.det(t_co[:3, :3]) ** (1 / 3)
        d_lo, d_hi = t_co[2, 3] - 1.0 * scale, t_co[2, 3] + 1.0 * scale
        depths = torch.linspace(d_lo, d_hi, prm.num_depth_samples)
        depth_obs[n_fg:] = 1.1 * d_hi
        jp_s, jc_s, r_s = _surface_term(dec, pts_t, t_oc, z)
        rr_s, l_s = _robust(r_s, prm.b2)
        if math.isnan(l_s):
            return dict(bad, loss=loss)
        rend = _render_term(dec, dirs, depth_obs, t_oc, depths, z, prm.cut_off)
        if rend is None:
            return dict(bad, loss=loss)
        jp_r, jc_r, r_r = rend
        rr_r, l_r = _robust(r_r, prm.b1)
        if math.isnan(l_r):
            return dict(bad, loss=loss)
        j_rot, r_rot = _rotation_prior(t_oc)
        loss = prm.k1 * l_r + prm.k2 * l_s
        j_s = torch.cat([jp_s, jc_s], dim=-1)
        h_s = prm.k2 * torch.bmm(j_s.transpose(-2, -1), j_s).sum(0).squeeze() / j_s.shape[0]
        b_s = -prm.k2 * torch.bmm(j_s.transpose(-2, -1), rr_s).sum(0).squeeze() / j_s.shape[0]
        j_r = torch.cat([jp_r, jc_r], dim=-1)
        h_r = prm.k1 * torch.bmm(j_r.transpose(-2, -1), j_r).sum(0).squeeze() / j_r.shape[0]
        b_r = -prm.k1 * torch.bmm(j_r.transpose(-2, -1), rr_r).sum(0).squeeze() / j_r.shape[0]
        h = h_r + h_s
        h[7:7 + n_code, 7:7 + n_code] += prm.k3 * torch.eye(n_code)
        b = b_r + b_s
        b[7:7 + n_code] -= prm.k3 * z
        j_rot = j_rot.unsqueeze(0)
        h[:7, :7] += prm.k4 * torch.mm(j_rot.transpose(-2, -1), j_rot)
        b[:7] -= prm.k4 * (-(j_rot.transpose(-2, -1) * r_rot).squeeze())
        h[:7, :7] += torch.eye(7)
        h[6, 6] += prm.s_damp
        dx = torch.mv(torch.inverse(h), b)
        t_oc = torch.mm(_exp_sim3(prm.lr * dx[:7]), t_oc)
        z += prm.lr * dx[7:7 + n_code]
    return dict(t_cam_obj=torch.inverse(t_oc).numpy(), code=z.numpy(), is_good=True, loss=float(loss))

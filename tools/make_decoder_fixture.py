#!/usr/bin/env python3
"""Fit a DeepSDF decoder (reference architecture, reference state-dict layout) to the analytic
rounded-box family of dsp_slam_amd/synth.py and store it as a compact fixture.

Why: the real `cars_64` / `chairs_64` weights are not in the reference tree (README.md:112) and
cannot be downloaded; a random-weight decoder has no zero crossing, so the render branch of
compute_render_loss (reconstruct/loss.py:88-150) would never execute.

Runs only in the build container (needs /root/reference for the Decoder class):
    python tools/make_decoder_fixture.py --name cars --seed 0 --steps 2000
Output: tests/golden/decoder_<name>.npz  (weight_v rounded to bf16-representable fp32 and stored
as uint16; weight_g / bias fp32) -- see dsp_slam_amd/fixtures.py for the reader, which can also
materialise the reference's on-disk format (specs.json + ModelParameters/latest.pth).
"""
import argparse
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import ref_shim  # noqa: E402
from dsp_slam_amd import synth  # noqa: E402
from dsp_slam_amd.fixtures import SPECS, save_decoder_npz  # noqa: E402


def sample_batch(rng, n, code_len, half=None):
    codes = np.zeros((n, code_len))
    codes[:, :3] = rng.normal(scale=0.45, size=(n, 3))
    codes[:, 3:] = rng.normal(scale=0.05, size=(n, code_len - 3))
    n_uni = n // 2
    # uniform in a ball of radius 1.05
    u = rng.normal(size=(n_uni, 3))
    u /= np.linalg.norm(u, axis=-1, keepdims=True)
    p_uni = u * (1.05 * rng.uniform(size=(n_uni, 1)) ** (1.0 / 3.0))
    # near-surface samples: per-point code, so do the bisection vectorised over codes
    m = n - n_uni
    d = rng.normal(size=(m, 3))
    d /= np.linalg.norm(d, axis=-1, keepdims=True)
    lo = np.zeros(m)
    hi = np.full(m, 1.6)
    c3 = codes[n_uni:, :3]
    for _ in range(30):
        mid = 0.5 * (lo + hi)
        s = synth.rounded_box_sdf(d * mid[:, None], c3, half)
        inside = s < 0
        lo = np.where(inside, mid, lo)
        hi = np.where(inside, hi, mid)
    p_surf = d * (0.5 * (lo + hi))[:, None]
    sig = np.where(rng.uniform(size=(m, 1)) < 0.5, 0.01, 0.05)
    p_near = p_surf + rng.normal(size=(m, 3)) * sig
    p = np.concatenate([p_uni, p_near], 0)
    sdf = synth.rounded_box_sdf(p, codes[:, :3], half)
    x = np.concatenate([codes, p], -1).astype(np.float32)
    return torch.from_numpy(x), torch.from_numpy(sdf.astype(np.float32))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--name", default="cars")
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--steps", type=int, default=2000)
    ap.add_argument("--batch", type=int, default=16384)
    ap.add_argument("--code-len", type=int, default=SPECS["CodeLength"])      # 32: the Redwood chairs option of LocalMapping_util.cc:415-423
    ap.add_argument("--half", type=float, nargs=3, default=None)               # base half extents of the shape family (default: synth.BOX_HALF)
    args = ap.parse_args()

    ref_shim.install()
    from deep_sdf.deep_sdf_decoder import Decoder  # the reference's class

    torch.manual_seed(args.seed)
    rng = np.random.default_rng(args.seed)
    code_len = args.code_len
    dec = Decoder(code_len, **SPECS["NetworkSpecs"])
    dec.eval()  # dropout inert, as in deep_sdf/workspace.py:221
    opt = torch.optim.Adam(dec.parameters(), lr=1e-3)
    sched = torch.optim.lr_scheduler.CosineAnnealingLR(opt, T_max=args.steps, eta_min=5e-5)
    clamp = 0.1
    t0 = time.time()
    for step in range(args.steps):
        x, y = sample_batch(rng, args.batch, code_len, args.half)
        pred = dec(x).squeeze(-1)
        loss = (torch.clamp(pred, -clamp, clamp) - torch.clamp(y, -clamp, clamp)).abs().mean()
        opt.zero_grad()
        loss.backward()
        opt.step()
        sched.step()
        if step % 50 == 0 or step == args.steps - 1:
            print("step %d  clamped-L1 %.5f  (%.0fs)" % (step, loss.item(), time.time() - t0), flush=True)

    sd = {k: v.detach().clone() for k, v in dec.state_dict().items()}
    out = os.path.join(ROOT, "tests", "golden", "decoder_%s.npz" % args.name)
    save_decoder_npz(sd, out, code_len=code_len)
    print("wrote", out, os.path.getsize(out) / 1e6, "MB")


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""Power / clock under the two decoder kernels ("prove the power wall or raise K0").

Runs the f16 prepass kernel (K0) and the fp32 forward kernel (K1) back to back for a few seconds each on resident points while a
background thread samples `rocm-smi --showpower --showclocks --json` (socket power, sclk), and prints the kernel's own clock measurement
(shader cycles / wall ticks of workgroup 0, dsp_debug_last_clocks) next to it.  Output: one JSON line per phase.
"""
import ctypes as C
import json
import os
import subprocess
import sys
import threading
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from dsp_slam_amd import fixtures, engine as E, _lib as L  # noqa: E402
from dsp_slam_amd.deep_sdf.deep_sdf_decoder import fold_weight_norm  # noqa: E402


def sampler(stop, out):
    while not stop.is_set():
        t = time.time()
        try:
            r = subprocess.run(["rocm-smi", "--showpower", "--showclocks", "--json"], stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True, timeout=5)
            d = json.loads(r.stdout)
            card = d[sorted(k for k in d if k.startswith("card"))[0]]
            out.append((t, card))
        except Exception as e:      # never let the probe die on a parse problem
            out.append((t, {"error": repr(e)}))
        time.sleep(0.05)


def numbers(card):
    pw = [float(v) for k, v in card.items() if "ower" in k and isinstance(v, str) and v.replace(".", "", 1).isdigit()]
    sclk = [v for k, v in card.items() if k.lower().startswith("sclk")]
    mhz = None
    for v in sclk:
        digits = "".join(c for c in str(v).split("Mhz")[0].replace("(", "") if c.isdigit())
        if digits:
            mhz = float(digits)
    return (pw[0] if pw else None), mhz


def main():
    layers = fold_weight_norm(fixtures.load_decoder_npz(fixtures.fixture_path("cars")), 9)
    eng = E.Engine(layers, [4], 64, device=0)
    lib = L.load()
    lib.dsp_debug_last_clocks.restype = C.c_int
    lib.dsp_debug_last_clocks.argtypes = [C.c_void_p, C.POINTER(C.c_uint64)]
    rng = np.random.default_rng(0)
    code = (rng.normal(size=64) * 0.2).astype(np.float32)
    seconds = float(sys.argv[1]) if len(sys.argv) > 1 else 4.0
    phases = [("idle", None, 0), ("K0 f16 prepass (mlp_lp_kernel<f16>)", lambda p: eng.decode_sdf_prepass(code, p, L.PREPASS_F16), 128 * 256 * 48),
              ("K1 fp32 forward (mlp_kernel<0>)", lambda p: eng.decode_sdf(code, p), 64 * 256 * 12)]
    for name, fn, n in phases:
        samples, stop = [], threading.Event()
        th = threading.Thread(target=sampler, args=(stop, samples))
        th.start()
        t0 = time.time()
        kclk, calls, flops = [], 0, 0.0
        if fn is None:
            time.sleep(seconds)
        else:
            pts = rng.uniform(-0.6, 0.6, size=(n, 3)).astype(np.float32)
            fn(pts)
            t0 = time.time()
            while time.time() - t0 < seconds:
                fn(pts)
                clk = (C.c_uint64 * 4)()
                L.check(lib.dsp_debug_last_clocks(eng._h, clk), eng._h, "clk")
                kclk.append((clk[2] - clk[0]) / ((clk[3] - clk[1]) / 100e6) / 1e6)
                flops += (clk[3] - clk[1]) / 100e6 and n * 3.67104e6 / ((clk[3] - clk[1]) / 100e6)
                calls += 1
        stop.set()
        th.join()
        vals = [numbers(c) for t, c in samples if t >= t0 and "error" not in c]
        pw = [p for p, _ in vals if p is not None]
        mh = [m for _, m in vals if m is not None]
        print(json.dumps({"phase": name, "seconds": seconds, "smi_samples": len(vals), "power_w_mean": round(float(np.mean(pw)), 1) if pw else None,
                          "power_w_max": max(pw) if pw else None, "smi_sclk_mhz_mean": round(float(np.mean(mh))) if mh else None,
                          "kernel_clock_mhz_mean": round(float(np.mean(kclk))) if kclk else None, "kernel_clock_mhz_min": round(min(kclk)) if kclk else None,
                          "kernel_tflops_mean": round(flops / max(calls, 1) / 1e12, 1) if calls else None, "calls": calls,
                          "raw_first_sample": samples[len(samples) // 2][1] if samples else None}), flush=True)
    eng.close()


if __name__ == "__main__":
    main()

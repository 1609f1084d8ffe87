"""GPU: the per-detection path on a SHARED GPU.  In DSP-SLAM the optimiser (LocalMapping thread, src/LocalMapping_util.cc:165-203) and the
detectors (Tracking thread: Mask R-CNN / PointPillars through PyTorch, src/Tracking_util.cc:31-57) use the same device.  The cluster form of
the jacobian launch needs its four workgroups per tile co-resident; a co-tenant that fills the CUs delays some of them.  What must hold:

  * results bit-identical to the solo run, always;
  * no stall: every spin is bounded to 2 ms and the fallback (latency-form kernel takes the list) happens on the device -- a detection never
    costs more than a bounded number of milliseconds on top of what the queueing behind the co-tenant's kernels costs any kernel;
  * the numbers (solo p50, co-tenant p50 / p99, how often the fallback fired) go to the parity log -> profiles/r05_latency_cotenant.md.

The co-tenant is a second handle driven from another host thread (ctypes releases the GIL): back-to-back dsp_decode_sdf launches of N
points -- N = 16 384 (256 tiles: one ~1 ms round over the chip per launch, the grain of a detector's convolution kernels) and N = 100 000
(~6 ms launches of persistent workgroups that hold every CU: the worst case for co-residency; fewer detections, to bound the test's time).
"""
import threading
import time

import numpy as np
import pytest

from conftest import parity_log
from dsp_slam_amd import engine as E, synth

pytestmark = pytest.mark.gpu


def _args(o):
    return ([o["t_cam_obj_init"]], [o["pts"]], [o["rays"]], [o["depth"]])


# (The highest stream priority was a parameter of this test in round 5: it changed nothing on average -- p50 10.1 against 9.9 ms -- and one run
# of 50 detections had outliers of 54 and 93 ms where the default had p99 13.4 ms: a detection's kernels are short and dependent, what they wait for
# is a co-tenant WORKGROUP leaving a CU, not the queue arbiter.  Measured, documented in profiles/r05_latency_cotenant.md, not asserted.)
@pytest.mark.parametrize("co_points,priority", [(16384, 0), (100000, 0)])
def test_detections_beside_a_cu_saturating_cotenant(oracle_decoder, co_points, priority):
    eng = E.Engine(oracle_decoder.layers, oracle_decoder.latent_in, oracle_decoder.code_len, device=0)
    if priority:
        eng.set_stream_priority(priority)       # dsp_set_stream_priority: this handle's kernels go first when a CU frees up
    co = E.Engine(oracle_decoder.layers, oracle_decoder.latent_in, oracle_decoder.code_len, device=0)
    prm = E.gn_params()
    dets = [synth.make_object(5000 + i, n_surface=250, n_background=200) for i in range(10)]
    batches = [eng.batch(prm, *_args(o)) for o in dets]
    # ---- solo ----
    want, solo = [], []
    for b in batches:
        b.run()
        want.append(b.results())
        assert b.stats()["n_cluster_tiles"] > 0
    for rep in range(3):
        for b in batches:
            t0 = time.perf_counter()
            b.run()
            b.results()
            solo.append((time.perf_counter() - t0) * 1e3)
    # ---- beside the co-tenant ----
    rng = np.random.default_rng(0)
    pts = rng.uniform(-0.7, 0.7, size=(co_points, 3)).astype(np.float32)
    code = np.zeros(64, np.float32)
    stop = threading.Event()
    launches = [0]

    def cotenant():
        while not stop.is_set():
            co.decode_sdf(code, pts)
            launches[0] += 1

    th = threading.Thread(target=cotenant)
    th.start()
    try:
        time.sleep(0.2)
        shared, fallbacks, cluster_runs = [], 0, 0
        n_rep, n_det = (5, 10) if co_points <= 16384 else (1, 6)
        for rep in range(n_rep):
            for b, w in list(zip(batches, want))[:n_det]:
                t0 = time.perf_counter()
                b.run()
                got = b.results()
                shared.append((time.perf_counter() - t0) * 1e3)
                st = b.stats()
                fallbacks += st["cluster_fallback"]
                cluster_runs += 1 if st["n_cluster_tiles"] > 0 else 0
                for x, y in zip(got, w):
                    assert np.array_equal(x, y), "a detection beside the co-tenant differs from its solo run"
    finally:
        stop.set()
        th.join()
    solo_p50 = float(np.median(solo))
    p50, p99, worst = float(np.median(shared)), float(np.percentile(shared, 99)), float(np.max(shared))
    print("co-tenant %d points/launch, stream priority %d (%d launches meanwhile): solo p50 %.2f ms; shared p50 %.2f p99 %.2f max %.2f ms; %d of %d runs fell back, %d used the cluster form" % (
        co_points, priority, launches[0], solo_p50, p50, p99, worst, fallbacks, len(shared), cluster_runs))
    parity_log(kind="cotenant", case="KITTI-size detections beside back-to-back dsp_decode_sdf(%d points) from a second handle / thread%s" % (co_points, ", this handle's stream at the highest priority" if priority else ""),
               solo_p50_ms=solo_p50, shared_p50_ms=p50, shared_p99_ms=p99, shared_max_ms=worst, runs=len(shared), fallback_runs=int(fallbacks),
               runs_with_cluster_tiles=int(cluster_runs), cotenant_launches=int(launches[0]))
    assert launches[0] > 0
    # Bounds that a regression CAN break (measured on two boxes, profiles/r05_latency_cotenant.md: 1 ms co-tenant p50 2.4-3.0 x solo, p99 2.9-3.4 x;
    # 6 ms persistent co-tenant p50 5.0-5.6 x, max 5.4-6.3 x).  A lost hand-off costs 2 ms once per run (then the cool-down keeps the cluster form
    # off); round 4's ~1 s stall per lost hand-off, or a detection queueing behind EVERY co-tenant launch instead of the one in flight, is far outside.
    if co_points <= 16384:
        assert p99 <= 4.0 * solo_p50, (p99, solo_p50)
        assert fallbacks <= 1, fallbacks                       # (none in any measured run)
    else:
        assert p50 <= 7.0 * solo_p50 and worst <= 10.0 * solo_p50, (p50, worst, solo_p50)
    for b in batches:
        b.close()
    eng.close()
    co.close()

"""Multi-GPU scale-out of the Gauss-Newton path: objects are independent, so ranks take contiguous blocks of
the object list (balanced by estimated work) and the only exchange is ONE gather of the per-object results
(pose 16 + code 64 + loss + status = 82 float32) to rank 0 -- RCCL over xGMI with backend "nccl", gloo on CPU.

One process per GPU (torch.distributed); nothing here touches the data path of a rank.
"""
import numpy as np

RESULT_WIDTH = 82


def object_cost(n_pts, n_rays, n_depth=50):
    """Relative work of one object: forward-only decoder points dominate (R*D), jacobian points cost 2x."""
    return float(n_rays) * n_depth + 2.0 * float(n_pts)


# Seconds per in-sphere sample and per kept render row over a whole run, REGRESSED on an MI355X from the step times of sixteen 119..133-object
# shards of the cfg4 job (profiles/r05_cfg4_balance.md: time = 1.25 ms per 1e6 sum-V + 81 ms per 1e6 sum-K, rms residual 0.4 %): the kept rows
# drive the band the fp32 forward kernel decodes and the backward-only launch; only the RATIO of the two matters to the partitioner.
_T_V = 1.25e-9
_T_K = 81e-9
_T_SURFACE = 2.0 * 3671040.0 / (0.871 * 157.3e12)      # forward + backward of one surface point at the jacobian kernels' rate


def measured_cost(n_pts, v, band, k, prepass=True):
    """Work of one object per Gauss-Newton iteration from its MEASURED set sizes: V in-sphere samples, K kept render rows (`band`, the
    samples the fp32 forward kernel decodes, follows K and is folded into its coefficient), n_pts surface points.  For objects of
    different SIZE (M, R) this separates them far better than the static R*D + 2M; among same-sized objects the first iteration's K
    predicts only half of the variance of the ten-iteration sum (correlation 0.51 over the 64 bench objects), and equal counts balance
    better (profiles/r05_cfg4_balance.md: 1.012 static against 1.051 measured)."""
    fwd = float(v) * _T_V * (1.0 if prepass else 9.6)       # prepass off: every in-sphere sample at the fp32 rate (1347 vs 140 TFLOP/s)
    return fwd + float(n_pts) * _T_SURFACE + float(k) * _T_K


def measure_costs(engine, prm, objs, chunk=128):
    """measured_cost of every object in `objs` from ONE Gauss-Newton iteration on `engine`'s GPU (a tenth of the job's work; the set sizes of
    the later iterations follow the first one's closely enough to balance shards: profiles/r05_cfg4_balance.md).  In a multi-GPU job every
    rank measures its equal-count slice and the costs are all-gathered before shard_objects runs (bench.py --config cfg4)."""
    costs = []
    for a in range(0, len(objs), chunk):
        ol = objs[a:a + chunk]
        bt = engine.batch(prm, [o["t_cam_obj_init"] for o in ol], [o["pts"] for o in ol], [o["rays"] for o in ol], [o["depth"] for o in ol], trace=True)
        try:
            bt.set_iterations(1)
            bt.run()
            tr = bt.trace(0)
            pre = bt.stats()["prepass_mode"] != 0
        finally:
            bt.close()
        for i, o in enumerate(ol):
            # m counts |sdf| < th among the decoded samples; the widened band + guard samples the fp32 kernel really decodes is ~1.15 m
            costs.append(measured_cost(len(o["pts"]), tr["V"][i], 1.15 * tr["m"][i], tr["K"][i], pre))
    return costs


def shard_objects(costs, world_size):
    """Contiguous block partition of objects 0..n-1 over ranks, balancing the summed cost.
    Returns a list of (start, stop) per rank; every object belongs to exactly one rank."""
    costs = np.asarray(costs, np.float64)
    n = costs.shape[0]
    bounds = [0]
    cum = np.concatenate([[0.0], np.cumsum(costs)])
    total = cum[-1]
    for r in range(1, world_size):
        target = total * r / world_size
        k = int(np.searchsorted(cum, target, side="left"))
        if k > 0 and abs(cum[k - 1] - target) <= abs(cum[min(k, n)] - target):
            k -= 1
        k = min(max(k, bounds[-1]), n)
        bounds.append(k)
    bounds.append(n)
    return [(bounds[r], bounds[r + 1]) for r in range(world_size)]


def pack_results(t_cam_obj, codes, loss, status):
    n = len(loss)
    out = np.zeros((n, RESULT_WIDTH), np.float32)
    out[:, :16] = np.asarray(t_cam_obj, np.float32).reshape(n, 16)
    if n:
        codes = np.asarray(codes, np.float32).reshape(n, -1)      # 32-D codes occupy the first 32 of the 64 slots
        out[:, 16:16 + codes.shape[1]] = codes
    out[:, 80] = loss
    out[:, 81] = np.asarray(status, np.float32)
    return out


def unpack_results(packed):
    packed = np.asarray(packed, np.float32)
    n = packed.shape[0]
    return packed[:, :16].reshape(n, 4, 4), packed[:, 16:80], packed[:, 80], packed[:, 81].astype(np.int32)


def gather_results(local_packed, shards, dist, device=None, dst=0):
    """One collective: every rank contributes its (n_local, 82) block; rank `dst` returns the (n_total, 82)
    array in object order, other ranks return None.  Blocks are padded to the largest shard so that a single
    dist.gather suffices (uneven shards)."""
    import torch
    rank, world = dist.get_rank(), dist.get_world_size()
    n_max = max(b - a for a, b in shards)
    pad = np.zeros((n_max, RESULT_WIDTH), np.float32)
    pad[:local_packed.shape[0]] = local_packed
    mine = torch.from_numpy(pad)
    if device is not None:
        mine = mine.to(device)
    bufs = [torch.empty_like(mine) for _ in range(world)] if rank == dst else None
    dist.gather(mine, bufs, dst=dst)
    if rank != dst:
        return None
    parts = [bufs[r][: shards[r][1] - shards[r][0]].cpu().numpy() for r in range(world)]
    return np.concatenate(parts, 0)


def gather_results_device(batches, shards, dist, device, dst=0):
    """The same single collective without a host bounce: every batch's packed result rows (kept in HBM by the library) are copied
    device-to-device into this rank's send tensor, ONE dist.gather (RCCL over xGMI) moves the blocks to rank `dst`, ONE device-to-host copy
    delivers them.  batches: this rank's engine.Batch objects, in object order (their object counts add up to this rank's shard)."""
    import torch
    rank, world = dist.get_rank(), dist.get_world_size()
    n_max = max(b - a for a, b in shards)
    mine = torch.zeros((max(n_max, 1), RESULT_WIDTH), dtype=torch.float32, device=device)
    # torch.zeros fills on TORCH's current stream; the library copies the rows on its own (non-blocking) stream, which nothing orders
    # behind that fill -- without this wait a late fill could zero rows the copy has already delivered (zeros read as "status GOOD")
    if torch.device(device).type == "cuda":
        torch.cuda.current_stream(device).synchronize()
    row = 0
    for bt in batches:
        bt.results_packed_to_device(mine.data_ptr() + row * RESULT_WIDTH * 4)
        row += bt.n
    assert row == shards[rank][1] - shards[rank][0], "the batches do not cover this rank's shard"
    bufs = [torch.empty_like(mine) for _ in range(world)] if rank == dst else None
    dist.gather(mine, bufs, dst=dst)
    if rank != dst:
        return None
    host = torch.stack(bufs).cpu().numpy()
    return np.concatenate([host[r, : shards[r][1] - shards[r][0]] for r in range(world)], 0)

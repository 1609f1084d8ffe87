"""CPU-only, world_size 2 over gloo: the multi-GPU path (dsp_slam_amd.distributed -- block sharding of independent
objects + ONE gather of the 82-float results to rank 0) returns, in object order, exactly what a single process
computes.  The per-object worker here is the CPU oracle on tiny objects (no GPU in this container); on the GPU box
bench.py plugs the HIP engine into the same shard/gather code with backend "nccl" (RCCL)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

from dsp_slam_amd import distributed as D, fixtures, synth

N_OBJ = 5           # uneven over 2 ranks on purpose


def _objects():
    return [synth.make_object(200 + i, n_surface=24 + 8 * i, n_background=6) for i in range(N_OBJ)]


def _solve(objs):
    from oracle import dsp_oracle as O
    dec = O.fold_decoder(fixtures.load_decoder_npz(fixtures.fixture_path("cars")), fixtures.SPECS)
    prm = O.GNParams(num_iterations=2)
    t, c, l, s = [], [], [], []
    for o in objs:
        r = O.reconstruct_object(dec, prm, o["t_cam_obj_init"], o["pts"], o["rays"], o["depth"])
        good = r["is_good"]
        t.append(r["t_cam_obj"] if good else np.zeros((4, 4), np.float32))
        c.append(r["code"] if good else np.zeros(64, np.float32))
        l.append(r["loss"])
        s.append(0 if good else 2)
    return D.pack_results(np.stack(t), np.stack(c), np.array(l, np.float32), np.array(s))


def _worker(rank, world, port, out_path, n_obj=N_OBJ):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.set_num_threads(2)
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    objs = _objects()[:n_obj]
    shards = D.shard_objects([D.object_cost(o["pts"].shape[0], o["rays"].shape[0]) for o in objs], world)
    a, b = shards[rank]
    local = _solve(objs[a:b]) if b > a else np.zeros((0, D.RESULT_WIDTH), np.float32)    # an empty shard still joins the gather
    full = D.gather_results(local, shards, dist)
    # the device-resident form bench.py uses (gather_results_device): the library copies each batch's packed rows to the address the
    # collective sends from.  Here the "batches" are host stand-ins that do that copy with memmove into a CPU tensor (gloo); on the GPU
    # box engine.Batch.results_packed_to_device does it device-to-device (tests/test_gpu_configs.py).  Two batches per rank where the
    # shard has two or more objects: their rows must land back to back.
    import ctypes

    class HostBatch(object):
        def __init__(self, rows):
            self.rows, self.n = np.ascontiguousarray(rows, np.float32), rows.shape[0]

        def results_packed_to_device(self, dst_ptr):
            ctypes.memmove(int(dst_ptr), self.rows.ctypes.data, self.rows.nbytes)

    k = local.shape[0] // 2
    batches = [HostBatch(local[:k]), HostBatch(local[k:])] if local.shape[0] >= 2 else [HostBatch(local)]
    full_dev = D.gather_results_device(batches, shards, dist, torch.device("cpu"))
    assert (full_dev is None) == (full is None) and (full is None or np.array_equal(full_dev, full))
    dist.barrier()
    if rank == 0:
        np.save(out_path, full)
        np.save(out_path + ".shards.npy", np.array(shards))
    else:
        assert full is None
    dist.destroy_process_group()


def test_two_rank_shard_and_gather(tmp_path):
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    out = str(tmp_path / "gathered.npy")
    mp.spawn(_worker, args=(2, port, out), nprocs=2, join=True)
    got = np.load(out)
    shards = np.load(out + ".shards.npy")
    assert shards[0][0] == 0 and shards[-1][1] == N_OBJ and shards[0][1] == shards[1][0]
    assert 0 < shards[0][1] < N_OBJ                       # both ranks had work
    torch.set_num_threads(2)
    want = _solve(_objects())
    assert got.shape == (N_OBJ, D.RESULT_WIDTH)
    assert np.array_equal(got, want)


def test_more_ranks_than_objects(tmp_path):
    """3 ranks, 2 objects: a rank with an empty shard contributes nothing but still takes part in the one collective
    (otherwise the other ranks hang in it)."""
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    out = str(tmp_path / "gathered3.npy")
    mp.spawn(_worker, args=(3, port, out, 2), nprocs=3, join=True)
    got = np.load(out)
    shards = np.load(out + ".shards.npy")
    assert sorted(b - a for a, b in shards) == [0, 1, 1]
    torch.set_num_threads(2)
    assert np.array_equal(got, _solve(_objects()[:2]))


def test_measured_cost_partitioner_balances_uneven_objects():
    """distributed.measured_cost / shard_objects (round 5): objects of identical (M, R) but very different kept-row counts K -- the bench
    batch spans 4.8 k ... 21 k -- get shards of equal summed cost, not of equal count; every object lands in exactly one shard, in order;
    the static cost R*D + 2M (identical for all of them) would cut equal counts."""
    rng = np.random.default_rng(3)
    n = 1024
    V = rng.integers(80000, 110000, size=n)
    K = np.where(np.arange(n) < n // 2, rng.integers(4000, 7000, size=n), rng.integers(15000, 21000, size=n))      # the heavy half at the end
    costs = [D.measured_cost(2000, V[i], 0.14 * V[i], K[i]) for i in range(n)]
    for world in (2, 3, 8):
        shards = D.shard_objects(costs, world)
        assert shards[0][0] == 0 and shards[-1][1] == n and all(shards[r][1] == shards[r + 1][0] for r in range(world - 1))
        per = np.array([sum(costs[a:b]) for a, b in shards])
        assert per.max() / per.mean() < 1.02, (world, per / per.mean())
        counts = np.array([b - a for a, b in shards])
        assert counts.max() > counts.min()                   # equal cost, not equal count
        static = D.shard_objects([D.object_cost(2000, 2500)] * n, world)
        per_static = np.array([sum(costs[a:b]) for a, b in static])
        assert per_static.max() / per_static.mean() > per.max() / per.mean()
    # the model's terms: prepass on -> V is cheap (f16 rate), the band and the rows dominate; prepass off -> V at the fp32 rate
    assert D.measured_cost(2000, 100000, 14000, 5000, prepass=False) > 2.0 * D.measured_cost(2000, 100000, 14000, 5000, prepass=True)

"""N>1 path on CPU: world_size-2 gloo run of graph-neural-net_amd/data_parallel.py.

The GPU engine cannot run here, so the test drives the SAME DataParallelStep / shard_rows code
with a CPU engine built on the oracle (test infrastructure) and checks that the sharded
compute -> all_reduce(SUM) -> update(B_global) equals one full-batch gradientStep (SCE:297-346)
on a single process, and that the replicas stay in lock-step."""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

DIMS = [24, 17, 9, 5]
B_LOCAL, WORLD, STEPS = 6, 2, 4


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


class OracleEngine:
    """CPU stand-in for HipEngine: same interface, arithmetic by the fp64 oracle."""

    def __init__(self, net, X, Y, torch):
        self.net, self.X, self.Y, self.torch = net, X, Y, torch
        self.grad_tensor = torch.zeros(net.n_params, dtype=torch.float64)
        self.v = np.zeros(net.n_params)

    def compute_gradient_range(self, first, B):
        g = sum(self.net.calculate_weight_gradient(self.X[r], self.Y[r]) for r in range(first, first + B))
        self.grad_tensor.copy_(self.torch.from_numpy(g))

    def apply_update(self, B_global, step, momentum):
        g = self.grad_tensor.numpy()
        adj = step * g / B_global + momentum * self.v     # SCE:333
        self.net.set_weights(self.net.get_weights() - adj)
        self.v = adj

    def weights_checksum(self):
        w = self.net.get_weights()
        return np.array([w.sum(), np.abs(w).sum()])


def _worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch
    import torch.distributed as dist
    import gnn_amd
    from gnn_amd import data_parallel as dp
    from oracle import oracle
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        Bg = B_LOCAL * world
        net = oracle.OracleNet(DIMS)
        X, Y = oracle.synthetic_batch(DIMS, Bg * STEPS, 77)
        eng = OracleEngine(net, X, Y, torch)
        stepper = dp.DataParallelStep(eng, dist)
        assert stepper.world == world and stepper.rank == rank
        for s in range(STEPS):
            lo, hi = dp.shard_rows(Bg, rank, world)
            assert hi - lo == B_LOCAL
            stepper.step(s * Bg + lo, B_LOCAL, 0.05, 0.9)
        assert stepper.replicas_in_lockstep(torch)
        np.save(os.path.join(out_dir, "w%d.npy" % rank), net.get_weights())
    finally:
        dist.destroy_process_group()


def test_shard_rows_partitions():
    from gnn_amd import data_parallel as dp
    for n in (1, 7, 128, 1024, 1000):
        for world in (1, 2, 3, 8):
            spans = [dp.shard_rows(n, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            for a, b in zip(spans, spans[1:]):
                assert a[1] == b[0]
            sizes = [hi - lo for lo, hi in spans]
            assert max(sizes) - min(sizes) <= 1


@pytest.mark.timeout(180)
def test_two_rank_gloo_equals_single_process(tmp_path, oracle_mod):
    import torch.multiprocessing as mp
    import gnn_amd  # noqa: F401  (registers the import shim for the workers' parent)
    port = _free_port()
    mp.spawn(_worker, args=(WORLD, port, str(tmp_path)), nprocs=WORLD, join=True)
    w0 = np.load(tmp_path / "w0.npy")
    w1 = np.load(tmp_path / "w1.npy")
    assert np.array_equal(w0, w1)                          # replicas bitwise identical
    ref = oracle_mod.OracleNet(DIMS)
    Bg = B_LOCAL * WORLD
    X, Y = oracle_mod.synthetic_batch(DIMS, Bg * STEPS, 77)
    for s in range(STEPS):
        ref.gradient_step(X[s * Bg:(s + 1) * Bg], Y[s * Bg:(s + 1) * Bg], 0.05, 0.9)
    # only the summation order differs (two partial sums instead of one serial sum)
    assert np.abs(w0 - ref.get_weights()).max() < 1e-13


@pytest.mark.parametrize("world", [4, 8])
def test_loopback_reducer_world_4_and_8(oracle_mod, world):
    """The sharding + reduction + update logic at the world sizes of the scaling runs (4 and 8 ranks), with the
    host loopback reducer standing in for RCCL: N ranks in one process, rank-ordered sum.
    Ragged global batch (not divisible by the world size)."""
    import torch
    import gnn_amd  # noqa: F401
    from gnn_amd import data_parallel as dp
    Bg, steps = 21, 3
    X, Y = oracle_mod.synthetic_batch(DIMS, Bg * steps, 99)
    group = dp.LoopbackGroup(world)

    def rank_body(rank, dist):
        net = oracle_mod.OracleNet(DIMS)
        eng = OracleEngine(net, X, Y, torch)
        stepper = dp.DataParallelStep(eng, dist)
        assert stepper.world == world and stepper.rank == rank
        for s in range(steps):
            lo, hi = dp.shard_rows(Bg, rank, world)
            # DataParallelStep takes B_global = B_local * world; with ragged shards the update size is passed explicitly
            eng.compute_gradient_range(s * Bg + lo, hi - lo)
            dist.all_reduce(eng.grad_tensor, op=dist.ReduceOp.SUM)
            eng.apply_update(Bg, 0.05, 0.9)
        assert stepper.replicas_in_lockstep(torch)
        return net.get_weights()
    ws = group.run(rank_body)
    for w in ws[1:]:
        assert np.array_equal(ws[0], w)
    ref = oracle_mod.OracleNet(DIMS)
    for s in range(steps):
        ref.gradient_step(X[s * Bg:(s + 1) * Bg], Y[s * Bg:(s + 1) * Bg], 0.05, 0.9)
    assert np.abs(ws[0] - ref.get_weights()).max() < 1e-13


def test_loopback_reducer_even_shards_through_the_stepper(oracle_mod):
    """DataParallelStep.step itself (B_global = B_local * world) over the loopback group, world 4."""
    import torch
    import gnn_amd  # noqa: F401
    from gnn_amd import data_parallel as dp
    world, B_local, steps = 4, 3, 3
    Bg = world * B_local
    X, Y = oracle_mod.synthetic_batch(DIMS, Bg * steps, 7)

    def rank_body(rank, dist):
        net = oracle_mod.OracleNet(DIMS)
        stepper = dp.DataParallelStep(OracleEngine(net, X, Y, torch), dist)
        for s in range(steps):
            stepper.step(s * Bg + rank * B_local, B_local, 0.05, 0.9)
        return net.get_weights()
    ws = dp.LoopbackGroup(world).run(rank_body)
    ref = oracle_mod.OracleNet(DIMS)
    for s in range(steps):
        ref.gradient_step(X[s * Bg:(s + 1) * Bg], Y[s * Bg:(s + 1) * Bg], 0.05, 0.9)
    assert all(np.array_equal(ws[0], w) for w in ws[1:])
    assert np.abs(ws[0] - ref.get_weights()).max() < 1e-13

"""Two ranks, ONE GPU: the real data-parallel code path (HipEngine + DataParallelStep + the HIP
kernels through the C ABI) with torch.distributed `gloo` carrying the all-reduce of the CUDA
gradient tensor.  RCCL needs one GPU per rank, and the test box has one, so this is as close as a
single-GPU box gets to the N>1 run: only the collective's transport differs from bench.py --gpus N.

Checks: (1) the two replicas stay bitwise identical; (2) they equal one process stepping on the
whole global batch, up to the summation order of the two partial gradients."""
import os
import socket
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DIMS, B_LOCAL, WORLD, STEPS = [784, 300, 100, 10], 64, 2, 6


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _data():
    rng = np.random.default_rng(123)
    n = B_LOCAL * WORLD * STEPS
    X = rng.random((n, DIMS[0])) * (rng.random((n, DIMS[0])) < 0.3)
    Y = np.eye(DIMS[-1])[rng.integers(0, DIMS[-1], n)]
    return X, Y


def _worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch
    import torch.distributed as dist
    import gnn_amd
    from gnn_amd import data_parallel as dp
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        X, Y = _data()
        Bg = B_LOCAL * world
        # this rank's shard of every global batch, laid out step after step
        rows = np.concatenate([np.arange(s * Bg + rank * B_LOCAL, s * Bg + (rank + 1) * B_LOCAL) for s in range(STEPS)])
        net = gnn_amd.SoftmaxCrossEntropyNeuralNet(DIMS, device=0, max_batch=B_LOCAL)
        net.upload_dataset(X[rows], Y[rows])
        stepper = dp.DataParallelStep(dp.HipEngine(net, torch), dist)
        assert stepper.world == world
        for s in range(STEPS):
            stepper.step(s * B_LOCAL, B_LOCAL, 0.0125, 0.9)
        torch.cuda.synchronize()
        assert net.time == STEPS
        assert stepper.replicas_in_lockstep(torch)
        np.save(os.path.join(out_dir, "w%d.npy" % rank), net.get_weights())
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_two_ranks_one_gpu_equal_single_process(gnn, tmp_path):
    import torch.multiprocessing as mp
    port = _free_port()
    mp.spawn(_worker, args=(WORLD, port, str(tmp_path)), nprocs=WORLD, join=True)
    w0, w1 = np.load(tmp_path / "w0.npy"), np.load(tmp_path / "w1.npy")
    assert np.array_equal(w0, w1)
    X, Y = _data()
    Bg = B_LOCAL * WORLD
    ref = gnn.SoftmaxCrossEntropyNeuralNet(DIMS, max_batch=Bg)
    ref.upload_dataset(X, Y)
    for s in range(STEPS):
        ref.compute_gradient_range(s * Bg, Bg)       # same split path: G, then the flat update
        ref.apply_update(Bg, 0.0125, 0.9)
    assert np.abs(w0 - ref.get_weights()).max() <= 1e-6

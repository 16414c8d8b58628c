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
DIMS, B_LOCAL, WORLD, STEPS = [784, 300, 100, 10], 64, 2, 24


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


def _worker(rank, world, port, out_dir, bf16):
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
        net = gnn_amd.SoftmaxCrossEntropyNeuralNet(DIMS, device=0, max_batch=B_LOCAL,
                                                   dtype=gnn_amd.DTYPE_BF16 if bf16 else gnn_amd.DTYPE_F32)
        net.upload_dataset(X[rows], Y[rows])
        # HipEngine makes its own side stream: kernels, the collective's copies and the update are
        # ordered by that ONE stream (a race between them would show up as replicas that differ,
        # or as weights off the single-process run, over these 24 back-to-back steps)
        stepper = dp.DataParallelStep(dp.HipEngine(net, torch), dist)
        assert stepper.engine.stream.cuda_stream != 0
        assert stepper.world == world
        for s in range(STEPS):
            stepper.step(s * B_LOCAL, B_LOCAL, 0.0125, 0.9)
            if s == 2:
                np.save(os.path.join(out_dir, "w%d_s3.npy" % rank), net.get_weights())
        torch.cuda.synchronize()
        assert net.time == STEPS
        assert stepper.replicas_in_lockstep(torch)
        np.save(os.path.join(out_dir, "w%d.npy" % rank), net.get_weights())
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(300)
@pytest.mark.parametrize("bf16", [False, True], ids=["f32", "bf16"])
def test_two_ranks_one_gpu_equal_single_process(gnn, tmp_path, bf16):
    """f32 = configs[1]'s arithmetic sharded; bf16 = BASELINE configs[2] (bf16 x data parallel)."""
    import torch.multiprocessing as mp
    port = _free_port()
    mp.spawn(_worker, args=(WORLD, port, str(tmp_path), bf16), nprocs=WORLD, join=True)
    w0, w1 = np.load(tmp_path / "w0.npy"), np.load(tmp_path / "w1.npy")
    assert np.array_equal(w0, w1)
    X, Y = _data()
    Bg = B_LOCAL * WORLD
    ref = gnn.SoftmaxCrossEntropyNeuralNet(DIMS, max_batch=Bg, dtype=gnn.DTYPE_BF16 if bf16 else gnn.DTYPE_F32)
    ref.upload_dataset(X, Y)
    for s in range(STEPS):
        ref.compute_gradient_range(s * Bg, Bg)       # same split path: G, then the flat update
        ref.apply_update(Bg, 0.0125, 0.9)
    # one process on the global batch: only the summation order of the two partial gradients differs.
    # f32: that stays at rounding level over the 24 steps.  bf16: a weight that differs in its last f32 bit can
    # round to the NEIGHBOURING bf16 operand value (a 0.4 % step), and with logits of +-100 at this init such a
    # flip moves saturated softmax rows: two correct runs drift apart, so the 24-step comparison is statistical
    # and the tight comparison is made after 3 steps below
    d = np.abs(w0 - ref.get_weights())
    if bf16:
        assert d.mean() <= 2e-4 and d.max() <= 5e-2
    else:
        assert d.max() <= 4e-6
    if bf16:
        # and the bf16-aware fp64 oracle on the GLOBAL batch (tests/np_oracle.py: every GEMM operand
        # rounded to bf16, f32-exact inputs)
        from tests import np_oracle
        ini = gnn.SoftmaxCrossEntropyNeuralNet(DIMS, max_batch=16)
        w, v = ini.get_weights(), np.zeros(ini.n_params)
        X32 = X.astype(np.float32).astype(np.float64)
        n_or = 3
        for s in range(n_or):
            w, v = np_oracle.gradient_step_bf16(w, v, DIMS, X32[s * Bg:(s + 1) * Bg], Y[s * Bg:(s + 1) * Bg], 0.0125, 0.9, 0)
        # the data-parallel replicas themselves after 3 global steps (tests/test_bf16_gpu.py allows one GPU 2e-4
        # after 3 steps; the two partial sums add their own reordering)
        assert np.abs(np.load(tmp_path / "w0_s3.npy") - w).max() <= 3e-4


@pytest.mark.timeout(300)
def test_config3_world8_one_process_per_gpu_form_on_one_gpu(gnn):
    """BASELINE configs[2] at its stated split through the one-process-per-GPU code (HipEngine + DataParallelStep):
    world 8, 128 rows per rank, global batch 1024, bf16 operands.  The box's process guard allows six GPU processes, so
    the eight ranks are eight THREADS of this process (LoopbackGroup: each with its own net, stream and gradient tensor
    on cuda:0; the rank-ordered sum crosses the host).  What differs from `bench.py --gpus 8` is the collective's
    transport only.  Replicas bitwise identical; equal to the bf16-aware fp64 restatement on the global batch."""
    import torch
    from gnn_amd import data_parallel as dp
    from tests import np_oracle
    world, B_local, steps = 8, 128, 3
    Bg = world * B_local
    rng = np.random.default_rng(77)
    X = rng.random((Bg * steps, DIMS[0])) * (rng.random((Bg * steps, DIMS[0])) < 0.3)
    Y = np.eye(DIMS[-1])[rng.integers(0, DIMS[-1], Bg * steps)]
    group = dp.LoopbackGroup(world)

    def rank_body(rank, dist):
        rows = np.concatenate([np.arange(s * Bg + rank * B_local, s * Bg + (rank + 1) * B_local) for s in range(steps)])
        net = gnn.SoftmaxCrossEntropyNeuralNet(DIMS, device=0, max_batch=B_local, dtype=gnn.DTYPE_BF16)
        net.upload_dataset(X[rows], Y[rows])
        stepper = dp.DataParallelStep(dp.HipEngine(net, torch), dist)
        assert stepper.world == world and stepper.rank == rank
        for s in range(steps):
            nxt = (s + 1) * B_local if s + 1 < steps else None
            stepper.step(s * B_local, B_local, 0.0125, 0.9, next_first=nxt)
        net.synchronize()
        assert net.time == steps
        return net.get_weights(), net.get_momentum()

    out = group.run(rank_body)
    for w_r, v_r in out[1:]:
        assert np.array_equal(w_r, out[0][0]) and np.array_equal(v_r, out[0][1])
    ini = gnn.SoftmaxCrossEntropyNeuralNet(DIMS, max_batch=16)
    w, v = ini.get_weights(), np.zeros(ini.n_params)
    X32 = X.astype(np.float32).astype(np.float64)
    for s in range(steps):
        w, v = np_oracle.gradient_step_bf16(w, v, DIMS, X32[s * Bg:(s + 1) * Bg], Y[s * Bg:(s + 1) * Bg], 0.0125, 0.9, 0)
    assert np.abs(out[0][0] - w).max() <= 3e-4
    assert np.abs(out[0][1] - v).max() <= 3e-4


@pytest.mark.parametrize("dtype,dims", [("f32", [784, 300, 100, 10]), ("bf16", [784, 300, 100, 10]), ("f32", [200, 512, 272, 10])],
                         ids=["f32-two-launch", "bf16-two-launch", "f32-per-layer-gemms"])
def test_rccl_exchange_inside_the_library_world_of_one(gnn, dtype, dims):
    """gnn_mlp_rccl_*: a rank attaches an RCCL communicator (ncclGetUniqueId -> ncclCommInitRank, a world of ONE here) and runs n
    steps in one call: gradient kernels -> ncclAllReduce of the flat gradient on the same stream -> update kernel.  With one rank
    the sum is the identity, so the weights equal, bit for bit, the same steps through the hooks (compute_gradient_range +
    apply_update with the next batch announced): the same kernels in the same order."""
    B, nb, steps = 128, 4, 9
    rng = np.random.default_rng(44)
    X = rng.random((B * nb, dims[0])) * (rng.random((B * nb, dims[0])) < 0.3)
    Y = np.eye(dims[-1])[rng.integers(0, dims[-1], B * nb)]
    dt = gnn.DTYPE_BF16 if dtype == "bf16" else gnn.DTYPE_F32
    a = gnn.SoftmaxCrossEntropyNeuralNet(dims, max_batch=B, dtype=dt)
    b = gnn.SoftmaxCrossEntropyNeuralNet(dims, max_batch=B, dtype=dt)
    a.upload_dataset(X, Y); b.upload_dataset(X, Y)
    with pytest.raises(gnn.GnnError):
        a.rccl_train_range(0, B, 1, 0.0125, 0.9)          # no communicator yet
    uid = gnn.NeuralNet.rccl_unique_id()
    assert len(uid) == 128
    a.rccl_attach(uid, 1, 0)
    with pytest.raises(gnn.GnnError):
        a.rccl_attach(uid, 1, 0)                          # already attached
    a.rccl_train_range(B, B, steps, 0.0125, 0.9)
    for s in range(steps):
        b.hint_next_range(((1 + s + 1) % nb) * B, B)
        b.compute_gradient_range(((1 + s) % nb) * B, B)
        b.apply_update(B, 0.0125, 0.9)
    assert a.time == steps == b.time
    assert np.array_equal(a.get_weights(), b.get_weights()) and np.array_equal(a.get_momentum(), b.get_momentum())
    a.rccl_detach()
    a.train_range(0, B, 2, 0.0125, 0.9); b.train_range(0, B, 2, 0.0125, 0.9)   # the handle steps on without a communicator
    assert np.array_equal(a.get_weights(), b.get_weights())

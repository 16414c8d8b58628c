"""gnn_mlp_dp_*: ONE handle over N device replicas, the data-parallel gradientStep inside the library
(the reference's caller is a single thread, NNT:83).  On the one-GPU box:
  * GNN_REDUCE_RCCL with n_dev = 1 (ncclCommInitAll + ncclAllReduce of the flat gradient on a world of one);
  * GNN_REDUCE_DIRECT with 2, 3 and 4 replicas that SHARE device 0: peer pointers, stream events between the
    replicas' streams and the rank-ordered sum + update kernel are exercised exactly as across devices.
Checks: replicas bitwise identical; equal to ONE net stepping on the whole batch up to the summation
order of the partial gradients; ragged shards (B not divisible, B < replicas); both dtypes."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
DIMS = [784, 300, 100, 10]


def data(n, seed=5):
    rng = np.random.default_rng(seed)
    X = rng.random((n, DIMS[0])) * (rng.random((n, DIMS[0])) < 0.3)
    Y = np.eye(DIMS[-1])[rng.integers(0, DIMS[-1], n)]
    return X, Y


def single_net_reference(gnn, X, Y, B, steps, dtype):
    ref = gnn.SoftmaxCrossEntropyNeuralNet(DIMS, max_batch=B, dtype=dtype)
    ref.upload_dataset(X, Y)
    nb = X.shape[0] // B
    for s in range(steps):
        ref.compute_gradient_range((s % nb) * B, B)     # the split path: gradient buffer, then the flat update
        ref.apply_update(B, 0.0125, 0.9)
    return ref


def test_rccl_world_of_one(gnn):
    B, steps = 96, 6
    X, Y = data(B * 3)
    net = gnn.DataParallelNeuralNet(DIMS, devices=[0], max_batch=B, reducer=gnn.REDUCE_RCCL)
    net.upload_dataset(X, Y)
    net.train_range(0, B, steps, 0.0125, 0.9)
    assert net.time == steps and net.replicas_identical()
    ref = single_net_reference(gnn, X, Y, B, steps, gnn.DTYPE_F32)
    assert np.array_equal(net.get_weights(), ref.get_weights())   # one replica: same kernels, same order
    with pytest.raises(gnn.GnnError, match="distinct device"):
        gnn.DataParallelNeuralNet(DIMS, devices=[0, 0], max_batch=B, reducer=gnn.REDUCE_RCCL)


@pytest.mark.parametrize("rs", [False, True], ids=["direct", "direct_rs"])
@pytest.mark.parametrize("n_rep,B", [(2, 128), (3, 100), (4, 64)])
def test_direct_reducer_replicas_sharing_one_device(gnn, n_rep, B, rs):
    """Both peer-memory reducers: DIRECT (every replica sums all partial gradients) and DIRECT_RS (reduce-scatter, then
    gather while updating).  train_range names each step's successor, so the reduction runs inside the tile-owner kernel
    (tile_step_kernel<GSRC = 3 / 4>); the last step, with no successor, takes the flat kernels."""
    steps = 7
    X, Y = data(B * 3, seed=n_rep)
    net = gnn.DataParallelNeuralNet(DIMS, devices=[0] * n_rep, max_batch=B, reducer=gnn.REDUCE_DIRECT_RS if rs else gnn.REDUCE_DIRECT)
    assert len(net.replicas) == n_rep
    net.upload_dataset(X, Y)
    net.train_range(0, B, steps, 0.0125, 0.9)
    net.synchronize()
    assert net.time == steps
    assert net.replicas_identical()                                # rank-ordered sum: the same bits on every replica
    ref = single_net_reference(gnn, X, Y, B, steps, gnn.DTYPE_F32)
    assert np.abs(net.get_weights() - ref.get_weights()).max() <= 2e-6   # only the partial sums' order differs
    p, pr = net.propagate(X[:B]), ref.propagate(X[:B])
    assert np.abs(p - pr).max() <= 1e-4


def test_host_batches_ragged_shards_and_tiny_batches(gnn, oracle_mod):
    """NeuralNet.gradientStep(double[] rows) through the dp handle: shard sizes 34/33/33, then a batch
    smaller than the replica count (a replica with no rows contributes a zero gradient)."""
    net = gnn.DataParallelNeuralNet(DIMS, devices=[0, 0, 0], max_batch=100, reducer=gnn.REDUCE_DIRECT)
    ref = oracle_mod.OracleNet(DIMS)
    ref.set_alloc_per_sample(0)
    for s, B in enumerate([100, 2, 37, 1]):
        X, Y = data(B, seed=20 + s)
        net.gradientStep(X, 0.0125, 0.9, False, expected=Y)
        ref.gradient_step(X, Y, 0.0125, 0.9)
    assert net.replicas_identical() and net.time == 4
    assert np.abs(net.get_weights() - ref.get_weights()).max() <= 4 * 2e-6
    with pytest.raises(gnn.GnnError):
        X, Y = data(101)
        net.gradientStep(X, 0.0125, 0.9, False, expected=Y)        # beyond max_batch
    with pytest.raises(gnn.GnnError):
        X, Y = data(4)
        net.gradientStep(X, 0.0125, 0.9, True, expected=Y)         # noise: unsupported, as on one device


def test_direct_reducer_bf16(gnn):
    """configs[2]'s arithmetic (bf16 operands) sharded inside the library: the bf16 shadow of W follows the
    masters in the reduce+update kernel; replicas identical, close to one net on the whole batch."""
    B, steps = 128, 3
    X, Y = data(B * 2, seed=9)
    net = gnn.DataParallelNeuralNet(DIMS, devices=[0, 0], max_batch=B, reducer=gnn.REDUCE_DIRECT, dtype=gnn.DTYPE_BF16)
    net.upload_dataset(X, Y)
    net.train_range(0, B, steps, 0.0125, 0.9)
    assert net.replicas_identical()
    ref = single_net_reference(gnn, X, Y, B, steps, gnn.DTYPE_BF16)
    assert np.abs(net.get_weights() - ref.get_weights()).max() <= 3e-4


def test_config3_eight_way_global_batch_1024_bf16(gnn):
    """BASELINE configs[2] AS STATED, on the one GPU there is: 784-300-100-10, bf16 operands, 8-way data parallel,
    global batch 1024 (128 rows per replica), the eight replicas sharing device 0 behind ONE handle (the direct
    reducer: peer pointers, cross-stream events, rank-ordered sum).  Replicas bitwise identical; weights equal to the
    bf16-aware fp64 restatement of gradientStep on the GLOBAL batch (SCE:297-346 with every GEMM operand rounded to
    bf16).  RCCL needs eight distinct devices, and the box's process guard allows six GPU processes: those legs are
    the driver's 8-GPU run."""
    from tests import np_oracle
    Bg, steps = 1024, 3
    X, Y = data(Bg * steps, seed=31)
    net = gnn.DataParallelNeuralNet(DIMS, devices=[0] * 8, max_batch=Bg, reducer=gnn.REDUCE_DIRECT, dtype=gnn.DTYPE_BF16)
    assert len(net.replicas) == 8
    w, v = net.get_weights(), np.zeros(net.n_params)
    net.upload_dataset(X, Y)
    net.train_range(0, Bg, steps, 0.0125, 0.9)
    net.synchronize()
    assert net.time == steps and net.replicas_identical()
    X32 = X.astype(np.float32).astype(np.float64)
    for s in range(steps):
        w, v = np_oracle.gradient_step_bf16(w, v, DIMS, X32[s * Bg:(s + 1) * Bg], Y[s * Bg:(s + 1) * Bg], 0.0125, 0.9, 0)
    assert np.abs(net.get_weights() - w).max() <= 3e-4
    assert np.abs(net.get_momentum() - v).max() <= 3e-4
    # the same three steps through the host-batch entry point (NeuralNet.gradientStep(rows): shards of 128 rows)
    net2 = gnn.DataParallelNeuralNet(DIMS, devices=[0] * 8, max_batch=Bg, reducer=gnn.REDUCE_DIRECT, dtype=gnn.DTYPE_BF16)
    for s in range(steps):
        net2.gradientStep(X[s * Bg:(s + 1) * Bg], 0.0125, 0.9, False, expected=Y[s * Bg:(s + 1) * Bg])
    assert net2.replicas_identical()
    assert np.array_equal(net2.get_weights(), net.get_weights())


def test_direct_reducers_agree_bitwise_and_with_stepwise_calls(gnn):
    """The sum of the partial gradients is taken in rank order by every form of the direct reducers -- fused into the tile
    kernel or flat, all-read-all or reduce-scatter + gather -- and the update has one spelling: the same bits from all of
    them, whether the steps come as one train_range (successors known: fused) or one call per step (flat kernels)."""
    B, steps, n_rep = 96, 5, 3
    X, Y = data(B * 2, seed=41)
    ws = []
    for reducer, stepwise in [(gnn.REDUCE_DIRECT, False), (gnn.REDUCE_DIRECT, True), (gnn.REDUCE_DIRECT_RS, False), (gnn.REDUCE_DIRECT_RS, True)]:
        net = gnn.DataParallelNeuralNet(DIMS, devices=[0] * n_rep, max_batch=B, reducer=reducer)
        net.upload_dataset(X, Y)
        if stepwise:
            for s in range(steps):
                net.gradient_step_range((s % 2) * B, B, 0.0125, 0.9)
        else:
            net.train_range(0, B, steps, 0.0125, 0.9)
        net.synchronize()
        assert net.replicas_identical() and net.time == steps
        ws.append((net.get_weights(), net.get_momentum()))
    for w, v in ws[1:]:
        assert np.array_equal(w, ws[0][0]) and np.array_equal(v, ws[0][1])

"""Data-parallel gradientStep: one process per GPU, one all-reduce per step.

The reference sums the per-sample gradients of a batch serially (SCE:305-322) and divides by
batch.size() in the update (SCE:333).  Samples are independent up to that sum, so the batch
shards by rows: every rank holds the same weights and momentum, computes the partial sum G_r
over its rows into ONE flat fp32 buffer (all layers contiguous), the ranks all-reduce(SUM) that
buffer once (RCCL over xGMI through torch.distributed; gloo on CPU in the tests), and every
rank applies the identical update with batchSize = the global batch.  No other collective is on
the path.

The engine behind `DataParallelStep` is anything with
    grad_tensor                      flat torch tensor the all-reduce runs on, in place
    compute_gradient_range(first, B) partial gradient of local dataset rows [first, first+B)
    apply_update(B_global, step, momentum)
`HipEngine` adapts a gnn_amd NeuralNet (the C ABI); tests drive the same class with a CPU
engine over gloo.
"""
import contextlib

import numpy as np


class CaptureFailed(RuntimeError):
    """Stream capture of the step sequence did not produce a graph.  The stream that was being
    captured is in the invalidated state; the process that holds it must not go on issuing GPU
    work (bench.py exits and lets a fresh process run the eager path)."""


def shard_rows(n_rows, rank, world):
    """Contiguous row block [lo, hi) of `rank`; the first n_rows % world ranks get one more."""
    base, extra = divmod(int(n_rows), int(world))
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


class HipEngine:
    """Binds a torch-owned CUDA tensor as the net's gradient buffer and moves the net's kernels
    onto a torch stream, so that compute -> all_reduce -> update are stream-ordered with no host
    synchronisation."""

    def __init__(self, net, torch_module, stream=None):
        torch = torch_module
        self.net = net
        self.torch = torch
        self.grad_tensor = torch.zeros(net.grad_elems, dtype=torch.float32, device="cuda")
        torch.cuda.synchronize()  # the fill ran on torch's current stream; the kernels may use another
        # Kernels, collective and update must share ONE stream.  torch's default stream has the
        # handle 0, which gnn_mlp_set_stream reads as "the handle's own (non-blocking) stream" --
        # a stream torch.distributed knows nothing about -- so a side stream is always made here
        # and DataParallelStep.step makes it torch's current stream around the collective.
        self.stream = stream if stream is not None else torch.cuda.Stream()
        net.set_stream(self.stream.cuda_stream)
        net.bind_grad_buffer(self.grad_tensor.data_ptr(), self.grad_tensor.numel())

    def stream_context(self):
        """Makes the engine's stream torch's current stream (the collective is enqueued on it)."""
        return self.torch.cuda.stream(self.stream)

    def rebind_stream(self, stream):
        """Move the net's kernels to another torch stream (after a capture that left the old one
        unusable)."""
        self.stream = stream
        self.net.set_stream(stream.cuda_stream)

    def compute_gradient_range(self, first, B):
        self.net.compute_gradient_range(first, B)

    def apply_update(self, B_global, step, momentum):
        self.net.apply_update(B_global, step, momentum)

    def hint_next_range(self, first, B):
        """The next compute_gradient_range will be on local rows [first, first+B): the update kernel
        then also starts that step (gnn_mlp_hint_next_range; speed only)."""
        self.net.hint_next_range(first, B)

    def weights_checksum(self):
        w = self.net.get_weights()
        return np.array([w.sum(), np.abs(w).sum()])


class DataParallelStep:
    def __init__(self, engine, dist_module=None, group=None, always_reduce=False):
        self.engine = engine
        self.always_reduce = always_reduce  # run the collective even on one rank (plumbing tests)
        self.dist = dist_module
        self.group = group
        self.world = dist_module.get_world_size(group) if dist_module is not None else 1
        self.rank = dist_module.get_rank(group) if dist_module is not None else 0

    def step(self, first, B_local, step, momentum, next_first=None):
        """One global gradientStep; every rank passes its own local rows.  The three stages are
        ordered by ONE stream: the engine's (CPU engines have none).  `next_first`: the local rows of
        the NEXT step, when the caller knows them (same B_local) -- results do not depend on it."""
        ctx = self.engine.stream_context() if hasattr(self.engine, "stream_context") else contextlib.nullcontext()
        with ctx:
            if next_first is not None and hasattr(self.engine, "hint_next_range"):
                self.engine.hint_next_range(next_first, B_local)
            self.engine.compute_gradient_range(first, B_local)
            B_global = B_local
            if self.dist is not None and (self.world > 1 or self.always_reduce):
                self.dist.all_reduce(self.engine.grad_tensor, op=self.dist.ReduceOp.SUM, group=self.group)
                B_global = B_local * self.world
            self.engine.apply_update(B_global, step, momentum)

    def replicas_in_lockstep(self, torch_module, device="cpu"):
        """True when every rank holds the same weights (the all-reduce result is bitwise identical
        on all ranks, so replicas never drift)."""
        if self.dist is None or self.world == 1:
            return True
        torch = torch_module
        mine = torch.tensor(self.engine.weights_checksum(), dtype=torch.float64, device=device)
        lo, hi = mine.clone(), mine.clone()
        self.dist.all_reduce(lo, op=self.dist.ReduceOp.MIN, group=self.group)
        self.dist.all_reduce(hi, op=self.dist.ReduceOp.MAX, group=self.group)
        return bool((lo == hi).all().item())


class GraphedSteps:
    """A fixed sequence of global gradientSteps captured ONCE into a HIP graph (stream capture of
    the engine's kernels and the RCCL all-reduce on one side stream) and replayed with a single
    launch: the per-step host work (three ABI calls and one collective launch) disappears, which
    is what bounds a small-net step in eager mode.  Needs a HipEngine built on `stream`."""

    def __init__(self, stepper, torch_module, stream, firsts, B_local, step, momentum, inject_failure=False):
        torch = torch_module
        self.stepper, self.torch, self.stream, self.n = stepper, torch, stream, len(firsts)
        self.graph = torch.cuda.CUDAGraph()
        nxt = firsts[1:] + firsts[:1]     # the sequence is replayed as a cycle: every step names its successor
        with torch.cuda.stream(stream):   # one eager pass: lazy RCCL / allocator initialisation
            for f, fn in zip(firsts, nxt):
                stepper.step(f, B_local, step, momentum, next_first=fn)
        stream.synchronize()
        net = stepper.engine.net
        t_before = net.time
        # The captured sequence must not depend on what ran before it: on the two-launch path a gradient computation
        # skips its chain-start launch when the handle believes the previous step already made this batch's first-layer
        # sums.  Forgetting that here puts the chain start INTO the graph; forgetting it again after the capture (the
        # bookkeeping then describes steps that were enqueued, not run) and after every replay keeps eager steps that
        # follow correct as well.
        self._forget = getattr(net, "forget_lookahead", lambda: None)
        self._forget()
        try:
            with torch.cuda.graph(self.graph, stream=stream):
                for f, fn in zip(firsts, nxt):
                    stepper.step(f, B_local, step, momentum, next_first=fn)
                if inject_failure:   # test hook: a call that is not permitted while capturing, so the capture really fails
                    torch.cuda.synchronize()
        except Exception as e:
            # A collective (or anything else) that cannot be captured invalidates the capture.  On
            # this ROCm the stream then stays invalidated and later HIP calls of the process crashed
            # (round 1), so nothing is repaired here: only the host-side step counter is put back,
            # and the caller is told to stop using this process for GPU work.
            net.advance_time(t_before - net.time)
            self._forget()
            raise CaptureFailed("%s: %s" % (type(e).__name__, str(e).splitlines()[0] if str(e) else "")) from e
        net.advance_time(-self.n)                 # the capture pass enqueued, it did not run
        self._forget()
        self.eager_steps = self.n                 # steps really executed by the warm pass

    def replay(self):
        with self.torch.cuda.stream(self.stream):
            self.graph.replay()
        self.stepper.engine.net.advance_time(self.n)
        self._forget()


class LoopbackGroup:
    """A host 'loopback' reducer: N ranks inside ONE process (one thread each) behind the part of the
    torch.distributed interface DataParallelStep uses (all_reduce with SUM / MIN / MAX, get_world_size,
    get_rank, barrier, ReduceOp).  It exists so that the sharding + reduction + lock-step logic can be
    run at world sizes 4 and 8 where there are neither 8 GPUs nor a process group; the
    sum is taken in RANK ORDER, like the library's direct reducer, so every rank gets the same bits.
    Works on numpy arrays and on torch tensors (CPU, or device tensors, which cross through the host: that is how
    the GPU engine is run at world 8 on a one-GPU box, eight ranks as threads of one process)."""

    class ReduceOp:
        SUM, MIN, MAX = "sum", "min", "max"

    def __init__(self, world):
        import threading
        self.world = int(world)
        self._slots = [None] * self.world
        self._bar = threading.Barrier(self.world)
        self._local = threading.local()

    def rank_view(self, rank):
        """The `dist`-like object rank `rank`'s thread passes to DataParallelStep."""
        return _LoopbackRank(self, int(rank))

    def run(self, fn):
        """Runs fn(rank, dist_like) on one thread per rank; returns the results in rank order; re-raises
        the first failure."""
        import threading
        out, err = [None] * self.world, [None] * self.world

        def body(r):
            try:
                out[r] = fn(r, self.rank_view(r))
            except BaseException as e:     # a failed rank must not leave the others waiting
                err[r] = e
                self._bar.abort()
        ts = [threading.Thread(target=body, args=(r,)) for r in range(self.world)]
        for t in ts:
            t.start()
        for t in ts:
            t.join()
        for e in err:
            if e is not None and not isinstance(e, __import__("threading").BrokenBarrierError):
                raise e
        for e in err:
            if e is not None:
                raise e
        return out


class _LoopbackRank:
    def __init__(self, group, rank):
        self.group, self.rank = group, rank
        self.ReduceOp = LoopbackGroup.ReduceOp

    def get_world_size(self, group=None):
        return self.group.world

    def get_rank(self, group=None):
        return self.rank

    def barrier(self, group=None):
        self.group._bar.wait()

    def all_reduce(self, tensor, op="sum", group=None):
        g = self.group
        on_device = bool(getattr(tensor, "is_cuda", False))
        if on_device:   # a device tensor: through the host (`.cpu()` waits for the work enqueued before it on the current stream)
            arr = tensor.detach().cpu().numpy()
        else:
            arr = tensor.numpy() if hasattr(tensor, "numpy") else np.asarray(tensor)
        g._slots[self.rank] = np.array(arr, copy=True)
        g._bar.wait()                                   # every rank's contribution is in
        acc = g._slots[0].copy()
        for r in range(1, g.world):                     # rank order: identical bits on every rank
            if op == "sum":
                acc = acc + g._slots[r]
            elif op == "min":
                acc = np.minimum(acc, g._slots[r])
            else:
                acc = np.maximum(acc, g._slots[r])
        g._bar.wait()                                   # nobody overwrites a slot another rank still reads
        if hasattr(tensor, "copy_"):
            import torch
            tensor.copy_(torch.from_numpy(acc))   # (device tensors: an H2D copy on the caller's current stream)
        else:
            arr[...] = acc

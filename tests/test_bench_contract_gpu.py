"""bench.py's output contract: exactly ONE line on stdout, a JSON object with the driver's keys
plus the `roofline` and `cpu_baseline` objects (the latter null when skipped), on the plain
single-process path and on the data-parallel path (RCCL, one rank)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KEYS = ["metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
        "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline", "host_batch_samples_per_s"]


def run_bench(*flags, want_stderr=False):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env.update(MASTER_ADDR="127.0.0.1", MASTER_PORT="29577")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "128", "--warmup", "64",
                          "--no-cpu-baseline", *flags], capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [l for l in out.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, out.stdout[:2000]
    return (json.loads(lines[0]), out.stderr) if want_stderr else json.loads(lines[0])


def check(line, n_gpus=1, dtype="f32", shared_gpu=False):
    for k in KEYS:
        assert k in line, k
    assert line["metric"].startswith("training samples/sec") and line["unit"] == "samples/s"
    assert line["steps"] == 128 and line["warmup"] == 64 and line["n_gpus"] == n_gpus
    assert line["higher_is_better"] is True and line["scaling"] == "weak" and line["vs_baseline"] is None
    assert line["dtype"] == dtype and line["data"] == "synthetic" and "workload" in line["config"]
    assert line["config"]["global_batch"] == 128 * n_gpus
    assert abs(line["value"] - n_gpus * 128 * 128 / (line["ms_per_step"] * 1e-3 * 128)) <= 0.01 * line["value"]
    r = line["roofline"]
    for k in ["bound", "achieved", "peak", "unit", "frac", "traffic", "kernel", "share_of_step", "other"]:
        assert k in r, k
    # (ranks time-slicing ONE GPU stretch a kernel's begin-to-end time to milliseconds: frac rounds to 0 there)
    assert r["bound"] in ("hbm", "mfma") and (0 if shared_gpu else 1e-4) <= r["frac"] < 1 and r["peak"] > 0
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3
    # the roofline kernel is the one with the largest share of the step
    assert all(r["avg_launch_us"] * r["launches"] >= o["avg_us"] * o["launches"] for o in r["other"].values())


def test_bench_line_single_process():
    line = run_bench()
    check(line)
    assert line["cpu_baseline"] is None          # --no-cpu-baseline
    assert line["roofline"]["traffic"] is None or isinstance(line["roofline"]["traffic"], int)
    assert 0 < line["roofline"]["gemm_784x300_mfma_frac"] < 1
    assert 0 < line["host_batch_samples_per_s"] < line["value"]   # PCIe-inclusive NeuralNet.gradientStep(double[]) rate
    # the same K steps between two events on the launch stream: within a few per cent of the wall clock
    assert 0.5 * line["ms_per_step"] <= line["ms_per_step_events"] <= 1.05 * line["ms_per_step"]
    oc = line["other_configs"]                   # short same-run timings of configs[0], [3], [4]
    for name in ("configs[0] f32", "configs[3] f32", "configs[3] bf16", "configs[4] f32 eager", "configs[4] f32 hipGraph", "configs[4] bf16"):
        assert "error" not in oc[name], oc[name]
        assert oc[name]["samples_per_s"] > 0 and oc[name]["us_per_step_events"] > 0
    assert oc["configs[4] f32 hipGraph"]["hipgraph"] is True
    for name in ("configs[1] GeneralNeuralNet sigmoid", "configs[1] GeneralNeuralNet leaky_relu/sigmoid"):
        assert "error" not in oc[name] and oc[name]["samples_per_s"] > 0
    inf = [k for k in oc if k.startswith("inference configs[1]")]
    assert len(inf) == 2
    for k in inf:   # evaluation over MNIST's 60 000 rows as one call (MT:181-197), beside the same loop one call per block
        assert "error" not in oc[k] and oc[k]["samples_per_s"] > oc[k]["per_block_calls_samples_per_s"] > 0


def test_bench_line_bf16():
    line = run_bench("--dtype", "bf16")
    check(line, dtype="bf16")
    assert "bf16" in line["config"]["workload"]


def test_bench_two_ranks_self_launched():
    """The driver's call shape at N > 1: `bench.py --gpus N` with no WORLD_SIZE.  bench.py starts the
    ranks itself (here both on the one GPU, gloo carrying the all-reduce: numbers meaningless)."""
    line = run_bench("--gpus", "2", "--backend", "gloo", "--share-gpu")
    check(line, n_gpus=2, shared_gpu=True)
    c = line["config"]
    assert c["backend"] == "gloo" and c["world_size"] == 2 and c["dp_mode"] == "eager"
    assert c["dp_replicas_identical"] is True and c["parallelism"] == "dp2"
    check_dp_variants(line, 2)


def check_dp_variants(line, n):
    """The rest of the N > 1 record, measured by the same invocation in fresh children (rehearsed here with both replicas / ranks
    on the one GPU, gloo carrying the ranks' all-reduce): every variant either a summary with value + roofline, or {"error": ..}.
    The in-library RCCL form needs one DISTINCT device per replica, so on this box it is the variant that fails -- and shows that a
    failing variant leaves the line intact."""
    v = line["dp_variants"]
    assert len(v) == 9
    by = lambda frag: next(v[k] for k in v if frag in k)
    assert "error" in by("library rccl") and any("distinct device" in l for l in by("library rccl")["stderr_tail"])
    assert "error" in by("RCCL inside the library's step loop")   # (two ranks of one RCCL communicator cannot share a device either)
    for frag, n_gpus, dtype in (("one GPU alone", 1, "f32"), ("all-reduce through torch.distributed", n, "f32"), ("library direct f32", n, "f32"), ("library direct_rs f32", n, "f32"),
                                ("ranks bf16", n, "bf16"), ("library direct bf16", n, "bf16"), ("ranks configs[4]", n, "f32")):
        e = by(frag)
        assert "error" not in e, (frag, e)
        assert e["value"] > 0 and e["n_gpus"] == n_gpus and e["dtype"] == dtype and e["roofline"] is not None, (frag, e)
        assert e["roofline"]["bound"] in ("hbm", "mfma") and e["roofline"]["frac"] >= 0
        if n_gpus > 1:
            assert e["config"]["dp_replicas_identical"] is True, frag
    assert by("ranks configs[4]")["config"]["global_batch"] == 256 * n and "784-1024-1024-1024-10" in by("ranks configs[4]")["config"]["workload"]
    assert by("ranks bf16")["config"]["global_batch"] == 128 * n
    assert line["single_gpu_value"] == by("one GPU alone")["value"]


def test_bench_two_ranks_started_by_torch_distributed_run():
    """The contract's call shape at N > 1: `python -m torch.distributed.run --nproc-per-node N bench.py --gpus N`.  Every rank the
    launcher starts supervises a child (the rank proper); rank 0's supervisor adds the dp variants once the ranks are gone and
    prints the ONE line."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "GNN_BENCH_LAUNCHER")}
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                          "--master-port", "29533", os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "128", "--warmup", "64",
                          "--no-cpu-baseline", "--backend", "gloo", "--share-gpu"], capture_output=True, text=True, timeout=1200, env=env, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [l for l in out.stdout.splitlines() if l.strip().startswith("{")]
    assert len(lines) == 1, out.stdout[:2000]
    line = json.loads(lines[0])
    check(line, n_gpus=2, shared_gpu=True)
    check_dp_variants(line, 2)


def test_bench_capture_failure_hands_over_to_fresh_eager_ranks():
    """A hipGraph capture that fails (here: a gloo collective, which cannot be captured, plus a
    call that is not permitted while capturing) leaves the capturing stream invalidated; the ranks
    must stop using the GPU and exit, and the eager run must come from FRESH processes."""
    line, err = run_bench("--gpus", "2", "--backend", "gloo", "--share-gpu", "--inject-capture-failure", "--no-dp-variants", want_stderr=True)
    check(line, n_gpus=2, shared_gpu=True)
    assert "hipGraph capture failed" in err and "starting fresh ranks: eager steps" in err
    assert line["config"]["dp_mode"] == "eager" and line["config"]["dp_replicas_identical"] is True


def test_bench_line_data_parallel_path():
    line, err = run_bench("--dp-path", want_stderr=True)
    check(line)
    assert line["config"]["dp_replicas_identical"] is True
    if line["config"]["dp_mode"] == "eager":
        # RCCL on a world of one is capturable and the graph attempt normally succeeds; once in the round's ~15 runs of this
        # test, on one box, it did not, and the launcher did what it is built to do -- which is then what has to be seen
        assert "starting fresh ranks: eager steps" in err, err[-2000:]
    else:
        assert line["config"]["dp_mode"] == "hipGraph replay of 64 steps"
    assert line["config"]["backend"] == "rccl" and line["config"]["world_size"] == 1


def test_bench_line_ranks_with_the_exchange_inside_the_library():
    """--dp-exchange library on a world of one rank: the rank attaches an RCCL communicator to its handle (unique id through the
    process group's store) and the K steps are ONE gnn_mlp_rccl_train_range call -- gradient kernels, ncclAllReduce, update kernel."""
    line = run_bench("--dp-path", "--dp-exchange", "library")
    check(line)
    c = line["config"]
    assert c["dp_replicas_identical"] is True and c["dp_mode"] == "eager" and c["world_size"] == 1
    assert "inside the library" in c["backend"]


def test_bench_capture_failure_under_external_launcher():
    """Started the way torch.distributed.run starts a rank (WORLD_SIZE in the environment, no launcher of
    ours): the rank supervises itself -- the process that was started stays off the GPU, the graph attempt
    runs in a child (with RCCL: the case in which the capturing process is killed by torch's NCCL watchdog
    ~100 ms after the failed capture), and the eager run in a second, fresh child."""
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29591", RANK="0", LOCAL_RANK="0", WORLD_SIZE="1")
    env.pop("GNN_BENCH_LAUNCHER", None)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "128", "--warmup", "64",
                          "--no-cpu-baseline", "--dp-path", "--inject-capture-failure"],
                         capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [l for l in out.stdout.splitlines() if l.strip()]
    assert len(lines) == 1
    line = json.loads(lines[0])
    check(line)
    assert "hipGraph capture failed" in out.stderr and "rank supervisor" in out.stderr
    assert line["config"]["dp_mode"] == "eager"


@pytest.mark.parametrize("reducer", ["direct", "direct_rs"])
def test_bench_library_data_parallel_rehearsal(reducer):
    """--dp-impl library: ONE process, one handle over N replicas (gnn_mlp_dp_train_range); on this box the replicas share
    device 0, which the peer-memory reducers accept.  Same driver keys; replicas identical."""
    line = run_bench("--gpus", "4", "--dp-impl", "library", "--dp-reducer", reducer, "--share-gpu")
    for k in KEYS:
        if k != "host_batch_samples_per_s":
            assert k in line, k
    assert line["n_gpus"] == 4 and line["config"]["global_batch"] == 512 and line["config"]["world_size"] == 4
    assert line["config"]["dp_replicas_identical"] is True and line["config"]["backend"] == reducer
    assert abs(line["value"] - 512 * 128 / (line["ms_per_step"] * 1e-3 * 128)) <= 0.01 * line["value"]
    r = line["roofline"]                      # as on the rank path: the kernel with the largest share of replica 0's step
    assert r is not None and r["bound"] in ("hbm", "mfma") and r["peak"] > 0 and "kernel" in r and "other" in r
    assert line["cpu_baseline"] is None      # --no-cpu-baseline; without the flag the library form reports it like the rank form

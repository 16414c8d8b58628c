#!/usr/bin/env python3
"""bench.py -- training samples/sec of the MLP mini-batch SGD path on MI355X.

  python bench.py --gpus N --steps K --warmup W

A "step" is one gradientStep (SCE:297-346: forward + backward + [all-reduce] + momentum update)
on one batch of synthetic 784-dim inputs already resident in HBM.  The workload at every N is
BASELINE.json configs[1] per GPU: 784-300-100-10, fp32, batch 128 per GPU (weak scaling: the
global batch is 128*N, sharded by rows, ONE all-reduce(SUM) of the flat weight gradient per step
over RCCL, identical update on every rank).  `--dtype bf16` runs configs[2]'s arithmetic (bf16
GEMM operands, f32 accumulate and masters) on the same shape; the headline stays f32.
Rank 0 prints ONE JSON line.

N > 1 is one process per GPU.  Either the caller starts the ranks (torch.distributed.run: RANK /
LOCAL_RANK / WORLD_SIZE / MASTER_* in the environment) or -- when `--gpus N` arrives with no
WORLD_SIZE, which is how the driver calls it -- this file starts them itself: the launcher below
imports neither torch nor the HIP library, spawns N fresh children of this script, relays rank 0's
line and returns the worst exit code.

Graph or eager (data-parallel path) is decided BEFORE anything runs: `--dp-mode graph` captures
one pass over the resident batches -- kernels and the RCCL all-reduce -- into one HIP graph.  If
that capture does not produce a graph, the stream it ran on stays invalidated, and a process that
holds an RCCL process group then dies within ~100 ms (the round-1 "later HIP calls crashed").  The
suspected cause -- not evidenced by a kept log -- is torch's ProcessGroupNCCL watchdog thread polling the
end event of the collective it enqueued on that stream.
So a rank whose capture failed reports it through a CPU-side store and leaves AT ONCE with
EXIT_CAPTURE_FAILED, its peers follow, and the eager run happens in FRESH processes: a process that
never touches the GPU supervises the attempt(s) -- the launcher for a group it started itself, a
per-rank supervisor when the ranks were started by torch.distributed.run.  In `auto` mode any
failure of the graph attempt (not only a reported capture failure) is followed by one eager attempt.
"""
import argparse
import datetime
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

DIMS = [784, 300, 100, 10]
BATCH = 128
N_BATCHES = 64                 # synthetic batches resident in HBM per rank
# --workload: the headline is configs[1]; configs[4] (BASELINE: "1 and 8 GPUs") is the second data-parallel workload, run as a
# variant of the N > 1 record.  (dims, rows per GPU, resident batches, factor on the Random(1) weights -- 0.05 keeps the softmax
# of a 1024-wide net unsaturated, as tools/bench_configs.py does --, label)
WORKLOADS = {
    "configs1": ([784, 300, 100, 10], 128, 64, 1.0, "784-300-100-10 SoftmaxCrossEntropyNeuralNet gradientStep"),
    "configs4": ([784, 1024, 1024, 1024, 10], 256, 16, 0.05, "784-1024-1024-1024-10 SoftmaxCrossEntropyNeuralNet gradientStep"),
}
STEP, MOMENTUM = 0.0125, 0.9   # MT:227-229
FP32_MFMA_PEAK_TFLOPS = 157.3  # MI355X_MICROARCH.md: v_mfma_f32_16x16x4_f32 dense peak
BF16_MFMA_PEAK_TFLOPS = 2500.0  # MI355X_MICROARCH.md: bf16 MFMA dense peak (the 5 PF figure is 2:1 sparse)
HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured copy)
EXIT_CAPTURE_FAILED = 75       # a rank's stream capture did not produce a graph: rerun eager in fresh processes
PMC_FILE = os.path.join("profiles", "r04", "pmc_traffic.json")
MFMA_FILE = os.path.join("profiles", "r04", "mfma_counters.json")


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2000)
    ap.add_argument("--warmup", type=int, default=200)
    ap.add_argument("--dtype", choices=["f32", "bf16"], default="f32",
                    help="GEMM operand type (f32 = BASELINE configs[1], the headline; bf16 = configs[2]'s arithmetic)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-other-configs", action="store_true",
                    help="skip the short same-run timings of BASELINE configs[0], [3], [4] (N=1 only; a few seconds)")
    ap.add_argument("--dp-mode", choices=["auto", "graph", "eager"], default="auto",
                    help="data-parallel path: hipGraph replay of the resident batches, or eager steps; "
                         "auto = eager on more than one rank, graph for the one-rank --dp-path with RCCL")
    ap.add_argument("--no-graph", action="store_true", help="same as --dp-mode eager")
    ap.add_argument("--dp-path", action="store_true",
                    help="run the data-parallel code path (compute -> all_reduce -> update) even at N=1")
    ap.add_argument("--dp-impl", choices=["ranks", "library"], default="ranks",
                    help="N > 1: one process per GPU over torch.distributed/RCCL (default), or ONE process whose handle owns the N "
                         "devices (gnn_mlp_dp_*: what a single-threaded JVM caller uses)")
    ap.add_argument("--dp-reducer", choices=["rccl", "direct", "direct_rs"], default="rccl",
                    help="--dp-impl library: RCCL all-reduce, or the peer-memory reducers (all-read-all / reduce-scatter + gather)")
    ap.add_argument("--dp-exchange", choices=["auto", "torch", "library"], default="auto",
                    help="--dp-impl ranks: the all-reduce through torch.distributed, or by the library itself inside its step loop "
                         "(gnn_mlp_rccl_*: each rank attaches an RCCL communicator to its handle; K steps are ONE call).  auto = on "
                         "more than one rank with RCCL the library's loop first and, should that attempt fail or overrun "
                         "--attempt-timeout, FRESH ranks through torch.distributed; torch on one rank and with other backends")
    ap.add_argument("--attempt-timeout", type=float, default=420.0,
                    help="seconds an attempt of the headline may take when another attempt can follow it")
    ap.add_argument("--backend", default="nccl",
                    help="torch.distributed backend; 'gloo' + --share-gpu rehearses the N>1 control flow on a one-GPU box")
    ap.add_argument("--share-gpu", action="store_true", help="rehearsal only: every rank uses cuda:0")
    ap.add_argument("--inject-capture-failure", action="store_true",
                    help="test hook: a non-capturable call inside the capture, so that the capture really fails")
    ap.add_argument("--workload", choices=sorted(WORKLOADS), default="configs1",
                    help="configs1 = the headline (784-300-100-10, 128 rows per GPU); configs4 = 784-1024-1024-1024-10, 256 rows per GPU")
    ap.add_argument("--no-dp-variants", action="store_true",
                    help="N > 1: only the headline (one process per GPU, RCCL); by default the same invocation also measures the "
                         "in-library reducers, bf16 configs[2], configs[4] and one GPU alone, each in fresh child processes")
    ap.add_argument("--variant-timeout", type=float, default=150.0, help="seconds one dp variant may take before it is ended")
    ap.add_argument("--variants-budget", type=float, default=600.0,
                    help="seconds all dp variants together may take: once spent, the remaining ones are skipped (reported as such)")
    ap.add_argument("--variant-child", action="store_true", help="(internal) this process measures ONE variant: no further children")
    args = ap.parse_args(argv)
    if args.no_graph:
        args.dp_mode = "eager"
    return args


# ---------------------------------------------------------------------------------------------
# launcher: no torch, no HIP -- only child processes
# ---------------------------------------------------------------------------------------------
def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def child_argv(argv, dp_mode, exchange=None):
    """The caller's flags with the graph/eager decision (and, when given, the exchange) made explicit."""
    out, skip = [], False
    for a in argv:
        if skip:
            skip = False
            continue
        if a == "--dp-mode" or (exchange is not None and a == "--dp-exchange"):
            skip = True
            continue
        if a.startswith("--dp-mode=") or a == "--no-graph" or a == "--inject-capture-failure":
            continue
        if exchange is not None and a.startswith("--dp-exchange="):
            continue
        out.append(a)
    return out + ["--dp-mode", dp_mode] + ([] if exchange is None else ["--dp-exchange", exchange])


def attempt_plan(args, world):
    """[(dp_mode, exchange)] in the order they are tried, each attempt in FRESH processes.
    More than one rank over RCCL: first the library's own step loop (gnn_mlp_rccl_*: gradient kernels, ncclAllReduce, update
    kernel enqueued by ONE call per K steps -- on a world of one 16.6 us per step where the per-step Python of the
    torch.distributed form costs 41, profiles/r04/dp_path_world1.log), then eager steps through torch.distributed.  Graph replay
    of the torch form on more than one rank stays opt-in (`--dp-mode graph`): the RCCL + hipGraph capture has only ever run on a
    world of one, and a capture that HANGS would leave the run without a number; the test hook asks for the failing attempt
    followed by the eager one.  One rank (`--dp-path`): the graph attempt, then eager."""
    if args.dp_mode != "auto":
        modes = [args.dp_mode]
    elif world > 1:
        modes = ["graph", "eager"] if args.inject_capture_failure else ["eager"]
    else:
        modes = ["graph", "eager"]
    ex = args.dp_exchange
    if ex == "auto":
        if world > 1 and args.backend == "nccl" and not args.share_gpu and not args.inject_capture_failure and args.dp_mode in ("auto", "eager"):
            return [("eager", "library")] + [(m, "torch") for m in modes]
        ex = "torch"
    if ex == "library":
        return [("eager", "library")]
    return [(m, ex) for m in modes]


def clean_env():
    """The environment of a child that is NOT a rank of the group this process belongs to: nothing of an outer launcher's
    rendezvous (torch.distributed.run exports RANK / WORLD_SIZE / MASTER_* / TORCHELASTIC_*; with TORCHELASTIC_USE_AGENT_STORE
    set, rank 0 of a fresh group would wait for the outer agent's store instead of hosting its own)."""
    drop = {"RANK", "LOCAL_RANK", "WORLD_SIZE", "LOCAL_WORLD_SIZE", "GROUP_RANK", "GROUP_WORLD_SIZE", "ROLE_RANK", "ROLE_WORLD_SIZE",
            "ROLE_NAME", "MASTER_ADDR", "MASTER_PORT", "GNN_BENCH_LAUNCHER"}
    return {k: v for k, v in os.environ.items() if k not in drop and not k.startswith("TORCHELASTIC_")}


def child_setup(new_session):
    """preexec of every child: it dies with this process (PR_SET_PDEATHSIG: a launcher the driver ends at its time limit must not
    leave ranks on the GPUs); a child that may have to be ended at a time limit also leads a session of its own."""
    def setup():
        import ctypes
        import signal
        if new_session:
            os.setsid()
        try:
            ctypes.CDLL(None).prctl(1, int(signal.SIGKILL))   # PR_SET_PDEATHSIG
        except Exception:
            pass
    return setup


def end_process_group(p):
    """Ends child `p` -- and, when it was started as the leader of its own session, exactly that process group."""
    import signal
    try:
        if os.getpgid(p.pid) == p.pid:
            os.killpg(p.pid, signal.SIGKILL)
    except (ProcessLookupError, PermissionError):
        pass
    try:
        p.kill()
    except Exception:
        pass


def run_group(argv, world, dp_mode, inject, timeout=None, stderr_to=None, exchange=None):
    """Starts `world` fresh ranks of this script; returns (worst exit code, rank 0's stdout).  timeout: seconds after which
    every rank is ended and 124 returned; stderr_to: a file the ranks' stderr goes to instead of this process's."""
    port = free_port()
    cmd = [sys.executable, os.path.abspath(__file__)] + child_argv(argv, dp_mode, exchange)
    if inject:
        cmd.append("--inject-capture-failure")
    procs = []
    err = stderr_to if stderr_to is not None else sys.stderr
    base = clean_env()
    for r in range(world):
        env = dict(base, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(world), LOCAL_WORLD_SIZE=str(world),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), GNN_BENCH_LAUNCHER="1")
        procs.append(subprocess.Popen(cmd, env=env, cwd=ROOT, stdout=subprocess.PIPE if r == 0 else err,
                                      stderr=err, text=(r == 0), preexec_fn=child_setup(timeout is not None)))
    # rank 0's stdout is one short line: reading it at the end cannot fill the pipe
    deadline_after_failure = None
    t_end = None if timeout is None else time.time() + timeout
    timed_out = False
    ended_here = set()
    while True:
        codes = [p.poll() for p in procs]
        if all(c is not None for c in codes):
            break
        if t_end is not None and time.time() > t_end:
            timed_out = True
            for i, p in enumerate(procs):
                if p.poll() is None:
                    end_process_group(p)      # exact process groups of the children started above
                    ended_here.add(i)
            t_end = None
        if any(c not in (None, 0) for c in codes):
            # a rank died: its peers would wait in a collective for ever; give them a moment, then end them
            if deadline_after_failure is None:
                deadline_after_failure = time.time() + 20.0
            elif time.time() > deadline_after_failure:
                for i, p in enumerate(procs):
                    if p.poll() is None:
                        end_process_group(p)
                        ended_here.add(i)
        time.sleep(0.05)
    out0 = procs[0].stdout.read() if procs[0].stdout else ""
    codes = [p.returncode for p in procs]
    if timed_out:
        return 124, out0
    if any(c == EXIT_CAPTURE_FAILED for c in codes):
        return EXIT_CAPTURE_FAILED, out0
    worst = 0
    for i, c in enumerate(codes):   # the code of a rank that failed by itself, not of the peers ended above
        if c != 0 and (i not in ended_here or worst == 0):
            worst = c if c > 0 else 128 - c
            if i not in ended_here:
                break
    return worst, out0


def run_single(argv, timeout, stderr_to):
    """ONE fresh child of this script that is not a rank of anything: (exit code, stdout); 124 when it was ended at `timeout`."""
    p = subprocess.Popen([sys.executable, os.path.abspath(__file__)] + list(argv), env=clean_env(), cwd=ROOT, stdout=subprocess.PIPE,
                         stderr=stderr_to, text=True, preexec_fn=child_setup(True))
    try:
        out, _ = p.communicate(timeout=timeout)
        rc = p.returncode
        return (rc if rc >= 0 else 128 - rc), out
    except subprocess.TimeoutExpired:
        end_process_group(p)
        try:
            out, _ = p.communicate(timeout=10)
        except Exception:
            out = ""
        return 124, out


# ---- the rest of the N > 1 record: every other data-parallel form, each in fresh child processes ---------------------------
def dp_variant_specs(args):
    """(name, "single" | "group", argv) of every variant measured beside the headline at N > 1.  `single` = one process (the
    in-library forms own all N devices; `one GPU alone` is the same workload at N = 1); `group` = N ranks, one per GPU."""
    n, K, W = args.gpus, args.steps, args.warmup
    tail = ["--no-cpu-baseline", "--no-other-configs", "--no-dp-variants", "--variant-child"]
    if args.share_gpu:
        tail.append("--share-gpu")
    ranks = ["--gpus", str(n), "--steps", str(K), "--warmup", str(W), "--backend", args.backend] + tail
    lib = ["--gpus", str(n), "--steps", str(K), "--warmup", str(W), "--dp-impl", "library"] + tail
    k4, w4 = min(K, 640), min(W, 64)
    return [
        ("one GPU alone (configs[1] f32, N = 1, same invocation)", "single",
         ["--gpus", "1", "--steps", str(K), "--warmup", str(W)] + [t for t in tail if t != "--share-gpu"]),
        ("library rccl f32 (one process over N devices, ncclAllReduce inside the library)", "single", lib + ["--dp-reducer", "rccl"]),
        ("library direct f32 (peer-memory reduction fused into the tile kernel)", "single", lib + ["--dp-reducer", "direct"]),
        ("library direct_rs f32 (peer-memory reduce-scatter, gather while updating)", "single", lib + ["--dp-reducer", "direct_rs"]),
        ("ranks f32, RCCL inside the library's step loop (gnn_mlp_rccl_*: K steps are one call per rank)", "group", ranks + ["--dp-exchange", "library"]),
        ("ranks f32, all-reduce through torch.distributed, eager steps", "group", ranks + ["--dp-exchange", "torch"]),
        ("ranks bf16 (BASELINE configs[2]: bf16 operands, global batch 128 N, RCCL all-reduce)", "group", ranks + ["--dtype", "bf16", "--dp-exchange", "torch"]),
        ("library direct bf16 (configs[2]'s arithmetic, one process over N devices)", "single", lib + ["--dp-reducer", "direct", "--dtype", "bf16"]),
        ("ranks configs[4] f32 (784-1024-1024-1024-10, 256 rows per GPU, RCCL all-reduce of 11.6 MB)", "group",
         ["--gpus", str(n), "--steps", str(k4), "--warmup", str(w4), "--backend", args.backend, "--workload", "configs4", "--dp-exchange", "torch"] + tail),
    ]


def summarize_variant(line, wall_s):
    c, r = line.get("config") or {}, line.get("roofline")
    out = {k: line.get(k) for k in ("value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "dtype")}
    out["config"] = {k: c[k] for k in ("workload", "global_batch", "parallelism", "dp_impl", "backend", "world_size", "dp_mode",
                                       "dp_replicas_identical", "devices_shared") if k in c}
    out["roofline"] = None if not r else {k: r.get(k) for k in ("bound", "kernel", "achieved", "peak", "unit", "frac", "avg_launch_us",
                                                                "traffic", "share_of_step")}
    out["child_wall_s"] = round(wall_s, 1)
    return out


def collect_variants(args):
    """Runs every variant of dp_variant_specs in FRESH child processes, one after the other, from a process that never touches the
    GPU.  A variant that fails or runs into --variant-timeout becomes {"error": ...} carrying the tail of its stderr; nothing a
    variant does can take the headline down."""
    import tempfile
    out = {}
    t_begin = time.time()
    for name, kind, argv in dp_variant_specs(args):
        t0 = time.time()
        left = args.variants_budget - (t0 - t_begin)
        if left < 5.0:   # (a hung collective in one variant after the other must not hold the headline back for ever)
            out[name] = {"error": "skipped: the variants' time budget of %.0f s is spent" % args.variants_budget}
            continue
        limit = min(args.variant_timeout, left)
        with tempfile.TemporaryFile(mode="w+") as errf:
            try:
                if kind == "group":
                    rc, text = run_group(argv, args.gpus, "eager", False, timeout=limit, stderr_to=errf)
                else:
                    rc, text = run_single(argv, limit, errf)
                lines = [l for l in (text or "").splitlines() if l.strip().startswith("{")]
                if rc == 0 and len(lines) == 1:
                    out[name] = summarize_variant(json.loads(lines[0]), time.time() - t0)
                    continue
                errf.seek(0)
                tail = [l for l in errf.read().splitlines() if l.strip()][-6:]
                what = ("ended at the %.0f s limit" % limit) if rc == 124 else "exit code %d" % rc
                out[name] = {"error": what, "stderr_tail": [l[-300:] for l in tail], "child_wall_s": round(time.time() - t0, 1)}
            except Exception as e:   # (a variant must not take the headline down)
                out[name] = {"error": "%s: %s" % (type(e).__name__, e)}
        print("bench.py: dp variant '%s': %s" % (name, "ok" if "error" not in out[name] else out[name]["error"]), file=sys.stderr)
    return out


def merge_variants(headline_text, args):
    """The headline's line with `dp_variants`, `single_gpu_value` (the same workload on ONE GPU, measured in this invocation) added."""
    line = json.loads(headline_text)
    variants = collect_variants(args)
    single = next((v for k, v in variants.items() if k.startswith("one GPU alone")), None)
    line["dp_variants"] = variants
    line["single_gpu_value"] = single.get("value") if single and "error" not in single else None
    line["dp_variants_note"] = ("each variant ran in its own fresh child process(es) after the headline; `value` above is the "
                                "one-process-per-GPU RCCL form; scaling efficiency is left to the reader of the per-N values")
    return json.dumps(line)


def launch(args, argv):
    plan = attempt_plan(args, args.gpus)
    rc, out0 = 1, ""
    for i, (mode, exchange) in enumerate(plan):
        last = i + 1 == len(plan)
        rc, out0 = run_group(argv, args.gpus, mode, args.inject_capture_failure and mode == "graph",
                             timeout=None if last else args.attempt_timeout, exchange=exchange)
        if rc == 0:
            lines = [l for l in out0.splitlines() if l.strip()]
            if len(lines) != 1:
                print("bench.py launcher: rank 0 printed %d lines instead of one" % len(lines), file=sys.stderr)
                return 1
            text = lines[0]
            if args.gpus > 1 and not args.no_dp_variants and not args.variant_child:
                text = merge_variants(text, args)
            print(text, flush=True)
            return 0
        if last:
            break
        what = "capture failed" if rc == EXIT_CAPTURE_FAILED else ("ended at the %.0f s limit" % args.attempt_timeout) if rc == 124 else "exit code %d" % rc
        print("bench.py launcher: the %s attempt failed in the ranks (%s); starting fresh ranks: %s steps, exchange through %s"
              % ("hipGraph" if mode == "graph" else "library-exchange" if exchange == "library" else mode, what, plan[i + 1][0],
                 "the library" if plan[i + 1][1] == "library" else "torch.distributed"), file=sys.stderr)
    print("bench.py launcher: ranks failed (exit code %d)" % rc, file=sys.stderr)
    return rc


# ---------------------------------------------------------------------------------------------
# one rank
# ---------------------------------------------------------------------------------------------
def pmc_traffic(kernel_substring):
    """L2<->fabric bytes per launch of a kernel from the committed rocprofv3 PMC passes (separate
    FETCH_SIZE / WRITE_SIZE runs of this very command, FETCH_SIZE doubled as the gfx950 guide
    prescribes; tools/pmc_summary.py).  bench.py cannot collect PMCs itself: the figure is from the
    file named in roofline.traffic_source, and null when the file has no kernel of that name."""
    try:
        with open(os.path.join(ROOT, PMC_FILE)) as f:
            for name, d in json.load(f)["kernels"].items():
                if kernel_substring in name:
                    return round(d["traffic_bytes_per_launch"])
    except Exception:
        pass
    return None


def mfma_counter(kernel_substring):
    """MFMA-pipe utilisation of a kernel from the committed rocprofv3 counter pass over this very command
    (SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE per XCD x 1024 SIMDs), tools/mfma_summary.py): the counter figure
    that stands beside FLOP / time / peak.  Null when the file has no kernel of that name."""
    try:
        with open(os.path.join(ROOT, MFMA_FILE)) as f:
            for name, d in json.load(f)["kernels"].items():
                if kernel_substring in name:
                    return d["mfma_util"]
    except Exception:
        pass
    return None


def other_configs():
    """Short same-run timings of the BASELINE configs bench.py's headline is NOT quoted on, so that the driver's line
    carries them: configs[0] (the reference's CPU-runnable case), configs[3] (f32 and bf16), configs[4] (f32 eager and
    under the hipGraph-captured step BASELINE names, bf16).  tools/bench_configs.py holds the accounting."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("bench_configs", os.path.join(ROOT, "tools", "bench_configs.py"))
    bc = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bc)
    out = {}
    # (name, config key, dtype, steps, hipGraph, resident batches, GeneralNeuralNet (inner, last) or None)
    # configs[4]: SIXTEEN batches resident, so that one graph launch carries 16 steps (112 kernel nodes) -- with 4 (round 3) a
    # launch carried 28 nodes and its cost was spread over 4 steps only: the captured line read 8 % SLOWER than the eager one.
    for name, key, dtype, steps, graph, nb, general in [
            ("configs[0] f32", "1", "f32", 1000, False, 4, None),
            ("configs[1] GeneralNeuralNet sigmoid", "2", "f32", 1000, False, 4, ("sigmoid", "sigmoid")),
            ("configs[1] GeneralNeuralNet leaky_relu/sigmoid", "2", "f32", 1000, False, 4, ("leaky_relu", "sigmoid")),
            ("configs[3] f32", "4", "f32", 120, False, 4, None), ("configs[3] bf16", "4", "bf16", 200, False, 4, None),   # (43 + 25 ms: with 30 / 50 steps the lines read 4-5 % over the 200-step runs of tools/bench_configs.py)
            ("configs[4] f32 eager", "5", "f32", 160, False, 16, None),
            ("configs[4] f32 hipGraph", "5", "f32", 160, True, 16, None),
            ("configs[4] bf16", "5", "bf16", 160, False, 16, None)]:
        try:
            line = bc.run(key, dtype, steps, graph=graph, n_batches=nb, timed_kernels=False, general=general)
            r = line["roofline"]
            out[name] = {"workload": line["config"]["workload"], "samples_per_s": line["value"], "us_per_step": round(line["ms_per_step"] * 1e3, 2),
                         "us_per_step_events": line.get("us_per_step_events"), "steps": steps, "dtype": dtype,
                         "hipgraph": graph, "batches_resident": nb, "bound": r["bound"], "mfma_frac": r["mfma_frac"], "hbm_frac": r["hbm_frac"]}
        except Exception as e:   # a config that fails must not take the headline down with it
            out[name] = {"error": "%s: %s" % (type(e).__name__, (str(e).splitlines() or [""])[0])}
    # inference / evaluation throughput: testOnTrainingData (MT:181-197) over 60 000 resident rows as ONE call
    # (gnn_mlp_count_hits_range: blocks of up to 16 384 rows through the handle's evaluation workspace), on a handle sized for
    # training (max_batch 128: the one-call-per-block form beside it walks 469 blocks) and on one sized for evaluation
    for name, mb in [("inference configs[1], 60000 rows, a training handle (max_batch 128)", 128),
                     ("inference configs[1], 60000 rows, max_batch 16384", 16384)]:
        try:
            line = bc.run_inference("2", 60000, mb)
            out[name] = {"samples_per_s": line["value"], "ms_per_pass": line["ms_per_pass"], "rows": line["rows"], "max_batch": mb,
                         "per_block_calls_samples_per_s": line["per_block_calls_samples_per_s"], "dtype": "f32",
                         "bound": line["roofline"]["bound"], "mfma_frac": line["roofline"]["frac"], "hbm_frac": line["roofline"]["hbm_frac"]}
        except Exception as e:
            out[name] = {"error": "%s: %s" % (type(e).__name__, (str(e).splitlines() or [""])[0])}
    return out


def synthetic(n, seed, dims=None):
    """X ~ U[0,1) 784-dim, uniform one-hot labels; generated here, never shipped."""
    import numpy as np
    dims = dims or DIMS
    rng = np.random.default_rng(seed)
    X = rng.random((n, dims[0]))
    Y = np.eye(dims[-1])[rng.integers(0, dims[-1], n)]
    return X, Y


def whole_step_roofline(dims, B, bf16, step_us):
    """`roofline` quoted on the WHOLE step (the per-layer GEMM path of the wide configs: a chain of 3 (L - 1) GEMMs that all run at
    the same roof) with SURVEY 8d's accounting: FLOP = (6P - 2 d0 d1) B; bytes = weights (P forward + (P - d0 d1) backward + P
    gradient + 5P update) + activations B (d0 + 4 sum_{l>=1} d_l); bound = MFMA when the arithmetic intensity is above the ridge."""
    P = sum(dims[l] * dims[l + 1] for l in range(len(dims) - 1))
    flop = (6 * P - 2 * dims[0] * dims[1]) * B
    eo = 2 if bf16 else 4
    nbytes = eo * (2 * P - dims[0] * dims[1]) + 4 * 6 * P + eo * B * (dims[0] + 4 * sum(dims[1:]))
    peak_tf = BF16_MFMA_PEAK_TFLOPS if bf16 else FP32_MFMA_PEAK_TFLOPS
    tf, gbs = flop / (step_us * 1e-6) / 1e12, nbytes / (step_us * 1e-6) / 1e9
    ridge, ai = peak_tf * 1e12 / (HBM_PEAK_GBS * 1e9), flop / nbytes
    bound = "mfma" if ai >= ridge else "hbm"
    ach, peak, unit = (tf, peak_tf, "TFLOP/s") if bound == "mfma" else (gbs, HBM_PEAK_GBS, "GB/s")
    return {"bound": bound, "kernel": "whole gradientStep (forward + backward GEMM chain + exchange + update)", "achieved": round(ach, 2),
            "peak": peak, "unit": unit, "frac": round(ach / peak, 4), "traffic": None, "avg_launch_us": round(step_us, 3), "launches": 1,
            "flop_per_step": flop, "algorithmic_bytes_per_step": nbytes, "arithmetic_intensity_flop_per_byte": round(ai, 1),
            "ridge_flop_per_byte": round(ridge, 1), "mfma_frac": round(tf / peak_tf, 4), "hbm_frac": round(gbs / HBM_PEAK_GBS, 4),
            "share_of_step": 1.0, "other": {}}


def cpu_baseline(seconds=12.0):
    """The oracle (serial fp64 C restatement of the Java loop, NOT a JVM run) timed on one host
    core on a bounded sample of the same workload.  Reported baseline only."""
    from oracle import oracle
    oracle.build()
    net = oracle.OracleNet(DIMS)
    X, Y = synthetic(BATCH * 4, 1234)
    net.gradient_step(X[:BATCH], Y[:BATCH], STEP, MOMENTUM)  # warm
    t0 = time.perf_counter()
    steps = 0
    while time.perf_counter() - t0 < seconds:
        r = (steps % 4) * BATCH
        net.gradient_step(X[r:r + BATCH], Y[r:r + BATCH], STEP, MOMENTUM)
        steps += 1
    dt = time.perf_counter() - t0
    return {"value": round(steps * BATCH / dt, 1), "unit": "samples/s", "cores": 1, "kind": "port",
            "sample": "%d gradientSteps of batch %d on 784-300-100-10 (fp64, per-sample loop order of "
                      "SCE:297-346; C restatement, not a JVM run)" % (steps, BATCH)}


def kernel_roofline(timing_read, DIMS, BATCH, bf16, is_dp, two_launch, step_us, nt):
    """The `roofline` object of the line from the per-kernel-class mean durations of the timed pass (the dispatches' own begin / end
    timestamps, gnn_mlp_timing_read): quoted on the kernel with the largest share of the step.  is_dp: the gradient kernel stores G
    and the update runs after the exchange; two_launch: row-block kernel + tile-owner kernel (csrc/tile_step_kernel.h)."""
    roofline = None
    fwd_us, fwd_n = timing_read(0)
    grad_us, grad_n = timing_read(1)
    mid_us, mid_n = timing_read(3)
    upd_us, upd_n = timing_read(4)
    P_all = sum(DIMS[l] * DIMS[l + 1] for l in range(len(DIMS) - 1))
    P_mid = P_all - DIMS[0] * DIMS[1]
    eo = 2 if bf16 else 4  # bytes per GEMM operand element
    mfma_peak = BF16_MFMA_PEAK_TFLOPS if bf16 else FP32_MFMA_PEAK_TFLOPS
    # algorithmic work per launch (SURVEY 8d accounting), f32 masters throughout:
    #   fwd_first : 2*B*d0*d1 FLOP; reads A_0 (B*d0) + W_0 (d0*d1), writes A_1 (B*d1)
    #   middle    : 2*B*2*(P - d0 d1) FLOP; reads W_1.. once per use (fwd + bwd), A_1, Y; writes A_2.., delta_1..
    #   grad      : 2*B*P FLOP; reads A_l, delta_{l+1} for every layer and W, V (2P f32); writes W, V (2P f32)
    #   (data-parallel path: the same kernel stores G instead -- P written, W and V untouched;
    #    the update is sgd_momentum_kernel after the all-reduce)
    f01 = 2.0 * BATCH * DIMS[0] * DIMS[1]     # one 784 x 300 product over the batch
    if two_launch and not is_dp:
        # the timed class "grad" is the tile-owner kernel: every layer's gradient + the update + the NEXT batch's
        # first-layer product; "fwd" is the forward-only launch that opens a chain (once per training call)
        names = {"fwd": "tile_step<forward only> (first layer of the chain's first batch, 128x784x300)",
                 "mid": "row-block kernel (K-slab sum + f, layers 2.., softmax/CE, backward data)",
                 "grad": "tile_step (G = A^T.delta all layers + momentum update + next batch's 128x784x300)"}
        kernels = {
            "fwd": (fwd_us, fwd_n, f01, eo * (BATCH * DIMS[0] + DIMS[0] * DIMS[1]) + 4 * BATCH * DIMS[1]),
            "mid": (mid_us, mid_n, 2.0 * BATCH * 2 * P_mid,
                    eo * 2 * P_mid + 4 * BATCH * DIMS[1] + eo * BATCH * 2 * sum(DIMS[1:]) + 4 * BATCH * 2 * DIMS[-1]),
            "grad": (grad_us, grad_n, 2.0 * BATCH * P_all + f01,
                     4 * 4 * P_all + (2 * P_all if bf16 else 0) + eo * BATCH * (sum(DIMS[:-1]) + sum(DIMS[1:]))
                     + eo * BATCH * DIMS[0] + 4 * BATCH * DIMS[1]),
        }
        gemm01 = ("grad", 2 * f01)            # both 784 x 300 products of a step run inside this kernel
    else:
        names = {"fwd": "first layer (128x784x300)", "mid": "middle(fwd L2.. + softmax + bwd-data)",
                 "grad": "gradient kernel (all layers, 784x300xB + ...%s)" % (", stores G; update after the all-reduce" if is_dp else " + update")}
        kernels = {
            "fwd": (fwd_us, fwd_n, f01, eo * (BATCH * DIMS[0] + DIMS[0] * DIMS[1]) + eo * BATCH * DIMS[1]),
            "mid": (mid_us, mid_n, 2.0 * BATCH * 2 * P_mid,
                    eo * 2 * P_mid + eo * BATCH * (DIMS[1] + 2 * sum(DIMS[1:])) + 4 * BATCH * 2 * DIMS[-1]),
            "grad": (grad_us, grad_n, 2.0 * BATCH * P_all,
                     4 * (1 if is_dp else 4) * P_all + eo * BATCH * (sum(DIMS[:-1]) + sum(DIMS[1:]))),
        }
        gemm01 = ("fwd", f01)
        if is_dp and upd_n:
            # data-parallel path: the update after the all-reduce -- by weight tiles with the next batch's first layer
            # fused in (two-launch path, a next-batch hint given), else the flat momentum kernel
            names["upd"] = "update after the all-reduce (+ next batch's 128x784x300 on the two-launch path)"
            kernels["upd"] = (upd_us, upd_n, f01 if two_launch else 0.0,
                              4 * 5 * P_all + (eo * BATCH * DIMS[0] + 4 * BATCH * DIMS[1] if two_launch else 0))
            if two_launch:
                gemm01 = ("upd", f01)
    # kernel names in the PMC file: the two-launch path's kernels first, the three-launch ones as the fallback
    pmc_key = {"fwd": ["tile_step_kernel<0, 0, true", "fwd_first"], "mid": ["rowblock", "middle4"],
               "grad": ["tile_step_kernel<1, 2, true", "grad_update"], "upd": ["tile_step_kernel<2, 2, true", "sgd_momentum"]}
    kernels = {k: v for k, v in kernels.items() if v[1] > 0 and v[0] > 0}
    whole_flop = (6 * P_all - 2 * DIMS[0] * DIMS[1]) * BATCH
    whole_bytes = 4 * (8 * P_all - DIMS[0] * DIMS[1] + BATCH * (DIMS[0] + 4 * sum(DIMS[1:])))

    def entry(k):
        us, n, flop, nbytes = kernels[k]
        return {"name": names[k], "avg_us": round(us, 3), "launches": n, "flop": flop, "algorithmic_bytes": nbytes,
                "tflops": round(flop / (us * 1e-6) / 1e12, 3), "gbs": round(nbytes / (us * 1e-6) / 1e9, 1),
                "mfma_frac": round(flop / (us * 1e-6) / 1e12 / mfma_peak, 4),
                "hbm_frac": round(nbytes / (us * 1e-6) / 1e9 / HBM_PEAK_GBS, 4),
                "share_of_step": round(us * n / (step_us * nt), 3),
                "mfma_util_counter": None if is_dp or bf16 else next((v for v in map(mfma_counter, pmc_key[k]) if v is not None), None),
                "traffic": None if is_dp or bf16 else next((v for v in map(pmc_traffic, pmc_key[k]) if v is not None), None)}
    if kernels:
        # roofline kernel = the one with the largest share of the step's time.  Which roof: its
        # arithmetic intensity against the ridge of this dtype (peak FLOP/s / 8 TB/s).
        dom = max(kernels, key=lambda k: kernels[k][0] * kernels[k][1])   # total time in the timed pass = avg x launches
        e = entry(dom)
        ai = e["flop"] / e["algorithmic_bytes"]
        ridge = mfma_peak * 1e12 / (HBM_PEAK_GBS * 1e9)
        if ai >= ridge:
            bound, ach, peak, unit = "mfma", e["tflops"], mfma_peak, "TFLOP/s"
        else:
            bound, ach, peak, unit = "hbm", e["gbs"], HBM_PEAK_GBS, "GB/s"
        return {"bound": bound, "kernel": e["name"], "achieved": ach, "peak": peak, "unit": unit,
                    "frac": round(ach / peak, 4), "traffic": e["traffic"],
                    "traffic_source": PMC_FILE if e["traffic"] is not None else None,
                    "avg_launch_us": e["avg_us"], "launches": e["launches"],
                    # the MFMA pipe's busy share by COUNTER (rocprofv3 pass in profiles/, same command), beside FLOP/time/peak
                    "mfma_util_counter": None if is_dp or bf16 else next((v for v in map(mfma_counter, pmc_key[dom]) if v is not None), None),
                    "mfma_util_counter_source": MFMA_FILE,
                    "algorithmic_bytes_per_launch": e["algorithmic_bytes"], "flop_per_launch": e["flop"],
                    "arithmetic_intensity_flop_per_byte": round(ai, 2), "ridge_flop_per_byte": round(ridge, 1),
                    "share_of_step": e["share_of_step"],
                    "limiter": "launch + memory latency and instruction issue, not bandwidth or MFMA rate: the launch moves "
                               "<1 MB and <0.1 GFLOP (DESIGN.md 3.3)",
                    # the north-star GEMM: FLOPs of the 784x300 product(s) / the time of the kernel they run in / MFMA peak
                    "gemm_784x300_mfma_frac": (round(gemm01[1] / (kernels[gemm01[0]][0] * 1e-6) / 1e12 / mfma_peak, 4)
                                               if gemm01[0] in kernels else None),
                    "gemm_784x300_kernel": names[gemm01[0]] if gemm01[0] in kernels else None,
                    # SURVEY 8d, whole step: 6P - 2 d0 d1 FLOP per sample; bytes = weights (P + (P - d0 d1) + P + 5P)
                    # + activations B (d0 + 4 sum_{l>=1} d_l), 4 B each
                    "whole_step": {"flop": whole_flop, "algorithmic_bytes": whole_bytes, "us": round(step_us, 3),
                                   "mfma_frac": round(whole_flop / (step_us * 1e-6) / 1e12 / mfma_peak, 4),
                                   "hbm_frac": round(whole_bytes / (step_us * 1e-6) / 1e9 / HBM_PEAK_GBS, 4)},
                    "other": {names[k]: entry(k) for k in kernels if k != dom}}
    return roofline


def library_worker(args):
    """--dp-impl library: ONE process, one handle over N devices (include/gnn_mlp.h, gnn_mlp_dp_*).  Same workload and the
    same JSON line as the one-process-per-GPU form: 128 rows per device, global batch 128 N, one sum of the flat gradient
    per step inside the library.  --share-gpu puts every replica on device 0 (peer-memory reducers only): a rehearsal of
    the control flow on a one-GPU box, not a scaling number."""
    import numpy as np
    import torch
    import gnn_amd
    if not torch.cuda.is_available():
        sys.exit("bench.py needs a GPU (the HIP path has no CPU fallback)")
    n = args.gpus
    if not args.share_gpu and torch.cuda.device_count() < n:
        sys.exit("bench.py --dp-impl library: %d devices asked, %d visible" % (n, torch.cuda.device_count()))
    reducer = {"rccl": gnn_amd.REDUCE_RCCL, "direct": gnn_amd.REDUCE_DIRECT, "direct_rs": gnn_amd.REDUCE_DIRECT_RS}[args.dp_reducer]
    K, W = args.steps, args.warmup
    bf16 = args.dtype == "bf16"
    Bg = BATCH * n
    X, Y = synthetic(Bg * N_BATCHES, 1000)
    net = gnn_amd.DataParallelNeuralNet(DIMS, devices=[0] * n if args.share_gpu else list(range(n)), max_batch=Bg, reducer=reducer,
                                        dtype=gnn_amd.DTYPE_BF16 if bf16 else gnn_amd.DTYPE_F32)
    net.upload_dataset(X, Y)
    net.train_range(0, Bg, W, STEP, MOMENTUM)
    net.synchronize()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    net.train_range((W % N_BATCHES) * Bg, Bg, K, STEP, MOMENTUM)
    net.synchronize()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    # per-kernel roofline as on the rank path: the dispatches' own timestamps on replica 0 (every replica runs the same launches
    # on its 128 rows), over a further pass of the same loop
    nt = min(K, 256)
    r0 = net.replicas[0]
    r0.timing_enable(True)
    net.train_range(((W + K) % N_BATCHES) * Bg, Bg, nt, STEP, MOMENTUM)
    net.synchronize()
    roofline = kernel_roofline(r0.timing_read, DIMS, BATCH, bf16, True, r0.step_launches == 2, dt / K * 1e6, nt)
    r0.timing_enable(False)
    identical = net.replicas_identical()
    cpu = None if args.no_cpu_baseline else cpu_baseline()
    line = {
        "metric": "training samples/sec, 784-300-100-10 MLP batch 128",
        "value": round(K * Bg / dt, 1), "unit": "samples/s", "n_gpus": n, "steps": K, "warmup": W,
        "ms_per_step": round(dt / K * 1e3, 5), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": args.dtype, "data": "synthetic",
        "config": {"workload": "784-300-100-10 SoftmaxCrossEntropyNeuralNet gradientStep, %s, batch 128 per GPU (BASELINE configs[%d])"
                               % (("bf16 GEMM operands / f32 accumulate and masters", 2) if bf16 else ("fp32", 1)),
                   "global_batch": Bg, "parallelism": "dp%d" % n, "dp_impl": "library: one process, one handle over %d devices" % n,
                   "backend": args.dp_reducer, "world_size": n, "devices_shared": bool(args.share_gpu),
                   "dp_replicas_identical": identical, "step": STEP, "momentum": MOMENTUM, "inner_activation": "leaky_relu"},
        "roofline": roofline, "cpu_baseline": cpu,
    }
    print(json.dumps(line), flush=True)


def supervise_own_rank(args, argv):
    """This process was started as ONE rank by someone else (torch.distributed.run), or is the single
    process of `--dp-path`.  It stays off the GPU and runs the rank as a child: on one rank the graph attempt
    first, and if that fails a fresh child in eager mode on the next rendezvous port (every rank's
    supervisor takes the same decision from the same exit code); on more than one rank eager steps (see launch()).
    Rank 0's supervisor then measures the other data-parallel forms (collect_variants) once the ranks are gone,
    and prints the merged line."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    plan = attempt_plan(args, world)
    rc = 1
    for i, (mode, exchange) in enumerate(plan):
        last = i + 1 == len(plan)
        env = dict(os.environ, GNN_BENCH_LAUNCHER="1")
        if i > 0:
            env["MASTER_PORT"] = str(int(os.environ.get("MASTER_PORT", "29511")) + i)
            env.pop("TORCHELASTIC_USE_AGENT_STORE", None)   # rank 0 of the fresh group hosts the new store itself
        cmd = [sys.executable, os.path.abspath(__file__)] + child_argv(argv, mode, exchange)
        if args.inject_capture_failure and mode == "graph":
            cmd.append("--inject-capture-failure")
        p = subprocess.Popen(cmd, env=env, cwd=ROOT, stdout=subprocess.PIPE, stderr=sys.stderr, text=True, preexec_fn=child_setup(not last))
        try:
            out_text, _ = p.communicate(timeout=None if last else args.attempt_timeout)
            rc = p.returncode
        except subprocess.TimeoutExpired:   # (every rank's supervisor takes the same decision at the same limit)
            end_process_group(p)
            try:
                out_text, _ = p.communicate(timeout=10)
            except Exception:
                out_text = ""
            rc = 124
        if rc == 0:
            text = (out_text or "").strip()
            if text and world > 1 and os.environ.get("RANK", "0") == "0" and not args.no_dp_variants and not args.variant_child:
                time.sleep(2.0)   # (the other ranks' children left the barrier with this one's: let them release their devices)
                try:
                    text = merge_variants(text.splitlines()[-1], args)
                except Exception as e:
                    print("bench.py rank supervisor: dp variants failed (%s: %s); headline only" % (type(e).__name__, e), file=sys.stderr)
            if text:
                print(text, flush=True)
            return 0
        if not last:
            what = "capture failed" if rc == EXIT_CAPTURE_FAILED else ("ended at the %.0f s limit" % args.attempt_timeout) if rc == 124 else "exit code %d" % rc
            print("bench.py rank supervisor: the %s attempt failed (%s); fresh child in %s mode, exchange through %s"
                  % ("hipGraph" if mode == "graph" else "library-exchange" if exchange == "library" else mode, what, plan[i + 1][0],
                     "the library" if plan[i + 1][1] == "library" else "torch.distributed"), file=sys.stderr)
    return rc if rc > 0 else 128 - rc


class _ChecksumOnly:
    """The part of data_parallel.HipEngine the lock-step check needs (the library-exchange ranks run their steps in C)."""

    def __init__(self, net):
        self.net = net

    def weights_checksum(self):
        import numpy as np
        w = self.net.get_weights()
        return np.array([w.sum(), np.abs(w).sum()])


def worker(args, argv):
    import numpy as np

    # stdout carries ONE JSON line: libraries that chat on fd 1 (RCCL prints a version banner there,
    # gloo its connection report) are routed to stderr until the line is printed
    sys.stdout.flush()
    stdout_fd = os.dup(1)
    os.dup2(2, 1)

    def emit(text):
        sys.stdout.flush()
        os.dup2(stdout_fd, 1)
        print(text, flush=True)

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))

    import torch
    import gnn_amd

    # The interpreter's cyclic collector: with torch imported a full (generation-2) pass walks ~10^6 import-time objects and
    # takes ~40 ms -- once, somewhere inside the per-call loops below (the host-batch loop of 1 000 calls read 70 us per call
    # instead of 27 whenever it caught it: tools/_build/hp_variants.py, profiles/r04/host_path_gc_pause.log).  The objects
    # alive now are moved to the permanent generation; the timed K steps are ONE library call and were never affected.
    import gc
    gc.collect()
    gc.freeze()

    if not torch.cuda.is_available():
        sys.exit("bench.py needs a GPU (the HIP path has no CPU fallback)")
    if args.share_gpu:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dist = None
    dp_mode = None
    if world > 1 or args.dp_path:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(args.backend)
        dp_mode = args.dp_mode
        if dp_mode == "auto":
            dp_mode = "graph" if (args.backend == "nccl" and world == 1) else "eager"   # (world > 1: see launch())

    K, W = args.steps, args.warmup
    bf16 = args.dtype == "bf16"
    DIMS, BATCH, N_BATCHES, w_scale, wl_label = WORKLOADS[args.workload]   # (the headline's constants unless --workload says otherwise)
    headline_shape = args.workload == "configs1"
    X, Y = synthetic(BATCH * N_BATCHES, 1000 + rank, DIMS)  # each rank owns its row shard of the global batch
    net = gnn_amd.SoftmaxCrossEntropyNeuralNet(DIMS, device=local_rank, max_batch=BATCH,
                                               dtype=gnn_amd.DTYPE_BF16 if bf16 else gnn_amd.DTYPE_F32)
    if w_scale != 1.0:
        net.set_weights(net.get_weights() * w_scale)   # (the same on every rank: Random(1) draws, one factor)
    net.upload_dataset(X, Y)

    def barrier():
        if dist is not None:
            dist.barrier()
        net.synchronize()
        torch.cuda.synchronize()

    graphed = None
    if dist is None:
        # the net's kernels on a torch stream, so that torch events bracket the timed region ON THE LAUNCH STREAM
        side = torch.cuda.Stream()
        net.set_stream(side.cuda_stream)

        def run(first_batch, n, eager=False):
            net.train_range((first_batch % N_BATCHES) * BATCH, BATCH, n, STEP, MOMENTUM)
    elif args.dp_exchange == "library":   # ("auto" is resolved by the launcher / the rank's supervisor; a bare rank takes torch)
        # one process per GPU, the exchange inside the library's own step loop (include/gnn_mlp.h, gnn_mlp_rccl_*): rank 0's RCCL
        # unique id travels through the process group's store, every rank attaches a communicator to its handle, and K steps are
        # ONE call -- gradient kernels, ncclAllReduce on the same stream, update kernel, no Python between steps
        from gnn_amd import data_parallel as dp
        side = torch.cuda.Stream()
        net.set_stream(side.cuda_stream)
        store = dist.distributed_c10d._get_default_store()
        if rank == 0:
            store.set("gnn_mlp_rccl_unique_id", net.rccl_unique_id())
        net.rccl_attach(store.get("gnn_mlp_rccl_unique_id"), world, rank)
        dp_mode = "eager"
        stepper = dp.DataParallelStep(_ChecksumOnly(net), dist)   # (for the lock-step check and the world size only)

        def run(first_batch, n, eager=False):
            net.rccl_train_range((first_batch % N_BATCHES) * BATCH, BATCH, n, STEP, MOMENTUM)
    else:
        # data parallel (graph-neural-net_amd/data_parallel.py): kernels on a torch side stream that
        # is also the collective's stream, gradient buffer owned by torch so that RCCL reduces it in
        # place, one all-reduce per step
        from gnn_amd import data_parallel as dp
        side = torch.cuda.Stream()
        stepper = dp.DataParallelStep(dp.HipEngine(net, torch, stream=side), dist, always_reduce=args.dp_path)
        if dp_mode == "graph":
            # one pass over the resident batches as ONE graph launch.  The ranks agree on the outcome
            # through a CPU-side store (no GPU call after a failed capture), then either all go on or
            # all leave.
            ok = True
            store = None
            if world > 1:   # made BEFORE the attempt: after a failure there is no time to set one up
                store = dist.TCPStore("127.0.0.1", int(os.environ["MASTER_PORT"]) + 3, world, rank == 0,
                                      timeout=datetime.timedelta(seconds=120))
            try:
                graphed = dp.GraphedSteps(stepper, torch, side, [b * BATCH for b in range(N_BATCHES)],
                                          BATCH, STEP, MOMENTUM, inject_failure=args.inject_capture_failure)
            except Exception as e:
                ok = False
                print("rank %d: hipGraph capture failed (%s: %s)" % (rank, type(e).__name__,
                                                                     (str(e).splitlines() or [""])[0]), file=sys.stderr)
            if not ok:
                # The stream is invalidated and this process is about to die (suspected: the RCCL watchdog): tell the peers and
                # leave at once -- no further GPU call, no teardown, no waiting.
                try:
                    if store is not None:
                        store.set("cap%d" % rank, "0")
                finally:
                    sys.stderr.flush()
                    os._exit(EXIT_CAPTURE_FAILED)
            if store is not None:
                store.set("cap%d" % rank, "1")
                try:
                    peers_ok = all(store.get("cap%d" % r) == b"1" for r in range(world))
                except Exception:   # the store's host (rank 0) has already left
                    peers_ok = False
                if not peers_ok:
                    sys.stderr.flush()
                    os._exit(EXIT_CAPTURE_FAILED)

        def run(first_batch, n, eager=False):
            s = 0
            while s < n:
                b = (first_batch + s) % N_BATCHES
                if graphed is not None and not eager and b == 0 and n - s >= N_BATCHES:
                    graphed.replay()
                    s += N_BATCHES
                else:
                    stepper.step(b * BATCH, BATCH, STEP, MOMENTUM, next_first=((b + 1) % N_BATCHES) * BATCH)
                    s += 1

    run(0, W)
    barrier()
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    ev0.record(side)
    run(0 if dist is not None else W, K)  # data parallel: start on a graph boundary (the resident batches)
    ev1.record(side)
    barrier()
    dt = time.perf_counter() - t0
    dt_events = ev0.elapsed_time(ev1) * 1e-3   # the same K steps between two events on the stream the kernels run on
    if dist is not None:
        t = torch.tensor([dt], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    # ---- per-kernel roofline: the dispatches' own begin/end timestamps, over the same step loop
    roofline = None
    cpu = None
    host_rate = None
    # The kernel-timing pass steps the net again. On the data-parallel path a step contains a
    # collective, so EVERY rank takes part (eagerly: kernels inside a replayed graph are not
    # timed); only rank 0 records and reports.
    nt = min(K, 1000) if dist is None else min(K, 256)
    if rank == 0:
        net.timing_enable(True)
    run(W + K if dist is None else K, nt, eager=True)
    barrier()
    if rank == 0:
        if headline_shape:
            roofline = kernel_roofline(net.timing_read, DIMS, BATCH, bf16, dist is not None, net.step_launches == 2, dt / K * 1e6, nt)
        else:
            roofline = whole_step_roofline(DIMS, BATCH, bf16, dt / K * 1e6)
        net.timing_enable(False)
        if dist is None and headline_shape and not args.variant_child:
            # PCIe-inclusive rate of the literal NeuralNet.gradientStep(double[] rows) call shape: fp64 host
            # batch -> f32 in a pinned slot -> staging kernel reading it over PCIe -> step.  Reported beside `value`, never as `value`.
            nh = 1000   # (~30 ms: the conversion helpers and the pinned slots are in their steady state after the first few dozen calls)
            for s in range(50):
                net.gradientStep(X[:BATCH], STEP, MOMENTUM, False, expected=Y[:BATCH])
            net.synchronize()
            th = time.perf_counter()
            for s in range(nh):
                r = (s % N_BATCHES) * BATCH
                net.gradientStep(X[r:r + BATCH], STEP, MOMENTUM, False, expected=Y[r:r + BATCH])
            net.synchronize()
            host_rate = round(nh * BATCH / (time.perf_counter() - th), 1)
            if not args.no_cpu_baseline:
                cpu = cpu_baseline()

    others = None
    if rank == 0 and dist is None and not args.no_other_configs and not args.variant_child:
        net.close()
        others = other_configs()

    lockstep = None
    if dist is not None:
        # every rank applied the same all-reduced gradient: the replicas must hold identical weights
        lockstep = stepper.replicas_in_lockstep(torch, device="cuda" if args.backend == "nccl" else "cpu")
        dist.barrier()
        dist.destroy_process_group()

    if rank == 0:
        total = K * BATCH * world
        line = {
            "metric": "training samples/sec, 784-300-100-10 MLP batch 128" if headline_shape else
                      "training samples/sec, %s MLP batch %d" % ("-".join(map(str, DIMS)), BATCH),
            "value": round(total / dt, 1), "unit": "samples/s",
            "n_gpus": world, "steps": K, "warmup": W,
            "ms_per_step": round(dt / K * 1e3, 5),
            "ms_per_step_events": round(dt_events / K * 1e3, 5),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": args.dtype, "data": "synthetic",
            "host_batch_samples_per_s": host_rate,
            "config": {"workload": "%s, %s, batch %d per GPU (BASELINE configs[%d])"
                                   % (wl_label, "bf16 GEMM operands / f32 accumulate and masters" if bf16 else "fp32", BATCH,
                                      4 if not headline_shape else 2 if bf16 else 1),
                       "global_batch": BATCH * world, "parallelism": "dp%d" % world,
                       "backend": None if dist is None else ("rccl inside the library's step loop (gnn_mlp_rccl_*)" if args.dp_exchange == "library"
                                                             else "rccl" if args.backend == "nccl" else args.backend),
                       "world_size": None if dist is None else stepper.world,
                       "dp_mode": (None if dist is None else
                                   ("hipGraph replay of %d steps" % N_BATCHES if graphed is not None else "eager")),
                       "dp_replicas_identical": lockstep,
                       "step": STEP, "momentum": MOMENTUM, "inner_activation": "leaky_relu"},
            "roofline": roofline,
            "cpu_baseline": cpu,
            "other_configs": others,
        }
        emit(json.dumps(line))


def main():
    argv = sys.argv[1:]
    args = parse_args(argv)
    world_env = os.environ.get("WORLD_SIZE")
    if args.dp_impl == "library" and world_env is None:
        library_worker(args)
        return
    if world_env is None and args.gpus > 1:
        sys.exit(launch(args, argv))       # nothing GPU-related has been imported in this process
    if world_env is not None and int(world_env) != args.gpus:
        args.gpus = int(world_env)
    is_dp = args.gpus > 1 or args.dp_path
    supervised = os.environ.get("GNN_BENCH_LAUNCHER") == "1"
    # a graph attempt -- and any attempt that another one may have to follow -- is never made in an unsupervised process; and a rank
    # of a world > 1 that someone else started (torch.distributed.run) is supervised too, so that rank 0's supervisor can add the
    # dp variants afterwards
    if is_dp and not supervised:
        plan = attempt_plan(args, args.gpus)
        if plan[0][0] == "graph" or len(plan) > 1 or (world_env is not None and args.gpus > 1 and not args.no_dp_variants):
            sys.exit(supervise_own_rank(args, argv))
        args.dp_mode, args.dp_exchange = plan[0]
    if args.dp_exchange == "auto":
        args.dp_exchange = "torch"
    worker(args, argv)


if __name__ == "__main__":
    main()

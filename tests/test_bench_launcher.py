"""bench.py's self-launcher (the driver calls `python bench.py --gpus N` with no WORLD_SIZE): the
launcher must start the ranks WITHOUT importing torch or the HIP library, relay rank 0's single line,
return the worst exit code, and hand a failed hipGraph capture (exit EXIT_CAPTURE_FAILED) to FRESH
ranks in eager mode.  CPU-only: the ranks themselves are faked."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench():
    sys.path.insert(0, ROOT)
    import bench
    return bench


def test_child_argv_makes_the_mode_explicit():
    b = _bench()
    got = b.child_argv(["--gpus", "8", "--dp-mode", "auto", "--steps", "5", "--no-graph", "--inject-capture-failure",
                        "--dp-mode=graph"], "eager")
    assert got == ["--gpus", "8", "--steps", "5", "--dp-mode", "eager"]


def test_launcher_relays_one_line_and_falls_back_to_fresh_eager_ranks(monkeypatch, capsys):
    b = _bench()
    calls, exchanges = [], []

    def fake_group(argv, world, mode, inject, timeout=None, stderr_to=None, exchange=None):
        calls.append((world, mode))
        exchanges.append((exchange, timeout))
        if mode == "graph":
            return b.EXIT_CAPTURE_FAILED, ""
        return 0, '{"n_gpus": %d}\n' % world
    monkeypatch.setattr(b, "run_group", fake_group)
    # the driver's call shape: eager ranks, ONE attempt (graph replay on more than one rank is opt-in)
    args = b.parse_args(["--gpus", "4", "--no-dp-variants"])
    assert b.launch(args, ["--gpus", "4", "--no-dp-variants"]) == 0
    assert calls == [(4, "eager")]
    assert exchanges == [("library", args.attempt_timeout)]   # more than one rank over RCCL: the library's own step loop first, with a time limit
    assert capsys.readouterr().out.strip() == '{"n_gpus": 4}'
    # the test hook: a graph attempt that fails, then fresh ranks in eager mode
    calls.clear()
    args = b.parse_args(["--gpus", "4", "--inject-capture-failure", "--no-dp-variants"])
    assert b.launch(args, ["--gpus", "4", "--inject-capture-failure", "--no-dp-variants"]) == 0
    assert calls == [(4, "graph"), (4, "eager")]
    assert capsys.readouterr().out.strip() == '{"n_gpus": 4}'
    # an explicit mode is not second-guessed; any other failure is passed through
    calls.clear()
    args = b.parse_args(["--gpus", "2", "--dp-mode", "graph"])
    assert b.launch(args, []) == b.EXIT_CAPTURE_FAILED and calls == [(2, "graph")]
    monkeypatch.setattr(b, "run_group", lambda *a, **k: (3, ""))
    assert b.launch(b.parse_args(["--gpus", "2", "--backend", "gloo"]), []) == 3


def test_launcher_process_never_imports_torch():
    code = ("import sys; sys.path.insert(0, %r); import bench\n"
            "bench.run_group = lambda *a, **k: (0, '{\"ok\": 1}\\n')\n"
            "sys.argv = ['bench.py', '--gpus', '8', '--no-dp-variants']\n"
            "try:\n    bench.main()\nexcept SystemExit as e:\n    assert e.code == 0, e.code\n"
            "assert 'torch' not in sys.modules and 'gnn_amd' not in sys.modules\n") % ROOT
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=env, timeout=60)
    assert out.returncode == 0, out.stderr
    assert out.stdout.strip() == '{"ok": 1}'


def test_run_group_real_children_worst_code_and_rank_env(tmp_path, monkeypatch):
    """run_group with real child processes (a stand-in script instead of bench.py): per-rank
    environment, rank 0's stdout relayed, peers of a dead rank ended, worst code returned."""
    b = _bench()
    script = tmp_path / "fake_rank.py"
    script.write_text(
        "import os, sys, time\n"
        "r, w = int(os.environ['RANK']), int(os.environ['WORLD_SIZE'])\n"
        "assert os.environ['LOCAL_RANK'] == str(r) and os.environ['MASTER_ADDR'] == '127.0.0.1'\n"
        "assert os.environ['GNN_BENCH_LAUNCHER'] == '1' and '--dp-mode' in sys.argv\n"
        "if '--fail' in sys.argv and r == 1: sys.exit(7)\n"
        "if '--fail' in sys.argv and r == 2: time.sleep(600)\n"
        "print('rank %d of %d' % (r, w)) if r == 0 else print('noise')\n")
    monkeypatch.setattr(b.os.path, "abspath", lambda p: str(script) if p == b.__file__ else os.path.normpath(os.path.join(os.getcwd(), p)))
    rc, out = b.run_group([], 3, "eager", False)
    assert rc == 0 and out.strip() == "rank 0 of 3"
    monkeypatch.setattr(b.time, "time", lambda _t=[0.0]: _t.__setitem__(0, _t[0] + 5.0) or _t[0])  # fast-forward the grace period
    rc, out = b.run_group(["--fail"], 3, "eager", False)
    assert rc == 7


def test_dp_variants_are_merged_and_a_failing_one_does_not_take_the_headline_down(monkeypatch, capsys):
    """`bench.py --gpus N` (N > 1): after the headline every other data-parallel form is measured in fresh children -- the
    in-library reducers, bf16 configs[2], configs[4], one GPU alone -- and merged into the ONE line; a variant that fails or runs
    into its time limit becomes {"error": ...} with the tail of its stderr."""
    import json
    b = _bench()
    seen = []

    def fake_group(argv, world, mode, inject, timeout=None, stderr_to=None, exchange=None):
        seen.append(("group", tuple(argv)))
        if "--variant-child" not in argv:
            return 0, json.dumps({"value": 800.0, "n_gpus": world, "config": {"backend": "rccl"}}) + "\n"
        if "--workload" in argv:                       # configs[4]: a rank dies
            stderr_to.write("noise\nRuntimeError: rank 1 fell over\n"); stderr_to.flush()
            return 3, ""
        return 0, json.dumps({"value": 700.0, "unit": "samples/s", "n_gpus": world, "dtype": "bf16", "ms_per_step": 0.1,
                              "config": {"backend": "rccl", "global_batch": 128 * world}, "roofline": {"bound": "hbm", "frac": 0.01}}) + "\n"

    def fake_single(argv, timeout, stderr_to):
        seen.append(("single", tuple(argv)))
        if argv[:2] == ["--gpus", "1"]:
            return 0, json.dumps({"value": 110.0, "n_gpus": 1, "config": {}, "roofline": None}) + "\n"
        if "rccl" in argv:
            return 124, ""                              # hangs: ended at the limit
        return 0, "RCCL banner on stdout\n" + json.dumps({"value": 650.0, "n_gpus": 8, "config": {"backend": argv[argv.index("--dp-reducer") + 1]},
                                                            "roofline": {"bound": "hbm", "frac": 0.02, "kernel": "k"}}) + "\n"
    monkeypatch.setattr(b, "run_group", fake_group)
    monkeypatch.setattr(b, "run_single", fake_single)
    args = b.parse_args(["--gpus", "8", "--steps", "50", "--warmup", "10"])
    assert b.launch(args, ["--gpus", "8", "--steps", "50", "--warmup", "10"]) == 0
    line = json.loads(capsys.readouterr().out.strip())
    assert line["value"] == 800.0 and line["n_gpus"] == 8           # the headline is the one-process-per-GPU RCCL form
    v = line["dp_variants"]
    assert len(v) == 9 and line["single_gpu_value"] == 110.0
    names = list(v)
    assert any("library rccl" in n for n in names) and any("direct_rs" in n for n in names) and any("configs[2]" in n for n in names)
    rccl = next(v[n] for n in names if "library rccl" in n)
    assert rccl["error"].startswith("ended at the") and "stderr_tail" in rccl
    c4 = next(v[n] for n in names if "configs[4]" in n)
    assert c4["error"] == "exit code 3" and c4["stderr_tail"][-1] == "RuntimeError: rank 1 fell over"
    direct = next(v[n] for n in names if "library direct f32" in n)
    assert direct["value"] == 650.0 and direct["roofline"]["frac"] == 0.02 and direct["config"]["backend"] == "direct"
    # once the variants' time budget is spent the rest is skipped, and says so
    args = b.parse_args(["--gpus", "8", "--steps", "50", "--warmup", "10", "--variants-budget", "0"])
    assert b.launch(args, ["--gpus", "8", "--steps", "50", "--warmup", "10", "--variants-budget", "0"]) == 0
    skipped = json.loads(capsys.readouterr().out.strip())
    assert skipped["value"] == 800.0 and len(skipped["dp_variants"]) == 9 and skipped["single_gpu_value"] is None
    assert all(e["error"].startswith("skipped") for e in skipped["dp_variants"].values())
    # every variant child is told not to spawn anything itself, and keeps the caller's K / W
    seen[:] = [x for x in seen if "--variants-budget" not in x[1]]
    for kind, argv in seen[1:]:
        assert "--variant-child" in argv and "--no-dp-variants" in argv and "--no-cpu-baseline" in argv
    assert all("50" in argv for kind, argv in seen[1:] if "--workload" not in argv)


def test_children_do_not_inherit_an_outer_launchers_rendezvous(monkeypatch):
    b = _bench()
    monkeypatch.setenv("TORCHELASTIC_USE_AGENT_STORE", "True")
    monkeypatch.setenv("WORLD_SIZE", "8"); monkeypatch.setenv("RANK", "0"); monkeypatch.setenv("MASTER_PORT", "1234")
    monkeypatch.setenv("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    e = b.clean_env()
    assert not any(k.startswith("TORCHELASTIC_") for k in e) and "WORLD_SIZE" not in e and "RANK" not in e and "MASTER_PORT" not in e
    assert e["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"


def test_library_exchange_attempt_falls_back_to_fresh_ranks_through_torch(monkeypatch, capsys):
    """More than one rank over RCCL: the first attempt runs the ranks with the exchange inside the library's step loop (K steps = one
    call per rank); if it fails -- or overruns --attempt-timeout -- FRESH ranks run the torch.distributed form, and the line says which."""
    import json
    b = _bench()
    seen = []

    def fake_group(argv, world, mode, inject, timeout=None, stderr_to=None, exchange=None):
        seen.append((mode, exchange, timeout))
        if exchange == "library":
            return 124, ""                                   # hangs in the communicator's set-up: ended at the limit
        return 0, json.dumps({"n_gpus": world, "config": {"backend": "rccl"}}) + "\n"
    monkeypatch.setattr(b, "run_group", fake_group)
    args = b.parse_args(["--gpus", "8", "--no-dp-variants", "--attempt-timeout", "7"])
    assert b.launch(args, ["--gpus", "8", "--no-dp-variants", "--attempt-timeout", "7"]) == 0
    assert seen == [("eager", "library", 7.0), ("eager", "torch", None)]
    out = capsys.readouterr()
    assert json.loads(out.out.strip())["config"]["backend"] == "rccl" and "library-exchange attempt failed" in out.err
    # the plans: one rank keeps the graph attempt; other backends, shared devices and explicit choices are not second-guessed
    assert b.attempt_plan(b.parse_args(["--gpus", "1", "--dp-path"]), 1) == [("graph", "torch"), ("eager", "torch")]
    assert b.attempt_plan(b.parse_args(["--gpus", "2", "--backend", "gloo"]), 2) == [("eager", "torch")]
    assert b.attempt_plan(b.parse_args(["--gpus", "2", "--share-gpu"]), 2) == [("eager", "torch")]
    assert b.attempt_plan(b.parse_args(["--gpus", "8", "--dp-exchange", "torch"]), 8) == [("eager", "torch")]
    assert b.attempt_plan(b.parse_args(["--gpus", "8", "--dp-exchange", "library"]), 8) == [("eager", "library")]
    assert b.attempt_plan(b.parse_args(["--gpus", "8", "--dp-mode", "graph"]), 8) == [("graph", "torch")]
    got = b.child_argv(["--gpus", "8", "--dp-exchange", "auto", "--steps", "5"], "eager", "library")
    assert got == ["--gpus", "8", "--steps", "5", "--dp-mode", "eager", "--dp-exchange", "library"]

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
    calls = []

    def fake_group(argv, world, mode, inject):
        calls.append((world, mode))
        if mode == "graph":
            return b.EXIT_CAPTURE_FAILED, ""
        return 0, '{"n_gpus": %d}\n' % world
    monkeypatch.setattr(b, "run_group", fake_group)
    # the driver's call shape: eager ranks, ONE attempt (graph replay on more than one rank is opt-in)
    args = b.parse_args(["--gpus", "4"])
    assert b.launch(args, ["--gpus", "4"]) == 0
    assert calls == [(4, "eager")]
    assert capsys.readouterr().out.strip() == '{"n_gpus": 4}'
    # the test hook: a graph attempt that fails, then fresh ranks in eager mode
    calls.clear()
    args = b.parse_args(["--gpus", "4", "--inject-capture-failure"])
    assert b.launch(args, ["--gpus", "4", "--inject-capture-failure"]) == 0
    assert calls == [(4, "graph"), (4, "eager")]
    assert capsys.readouterr().out.strip() == '{"n_gpus": 4}'
    # an explicit mode is not second-guessed; any other failure is passed through
    calls.clear()
    args = b.parse_args(["--gpus", "2", "--dp-mode", "graph"])
    assert b.launch(args, []) == b.EXIT_CAPTURE_FAILED and calls == [(2, "graph")]
    monkeypatch.setattr(b, "run_group", lambda *a: (3, ""))
    assert b.launch(b.parse_args(["--gpus", "2", "--backend", "gloo"]), []) == 3


def test_launcher_process_never_imports_torch():
    code = ("import sys; sys.path.insert(0, %r); import bench\n"
            "bench.run_group = lambda *a: (0, '{\"ok\": 1}\\n')\n"
            "sys.argv = ['bench.py', '--gpus', '8']\n"
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

"""csrc/convert_helper.h: the fp64 -> f32 conversion of a host batch, shared between the calling thread and up to three
helper threads that poll for the next batch before they sleep (the literal NeuralNet.gradientStep(double[] rows) call of
NeuralNetTrainer.java:83).  Host code only: built with g++ -fsanitize=thread and run here, no GPU."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "tests", "native", "convert_helper_check.cpp")
INC = os.path.join(ROOT, "graph-neural-net_amd", "csrc")


@pytest.mark.parametrize("threads", ["3", "1", "0"])
@pytest.mark.parametrize("sanitize", [True, False])
def test_conversion_with_helpers(tmp_path, threads, sanitize):
    gxx = shutil.which("g++")
    if not gxx:
        pytest.skip("no g++")
    out = str(tmp_path / "convert_helper_check")
    flags = ["-fsanitize=thread", "-O1", "-g"] if sanitize else ["-O2"]
    subprocess.run([gxx, "-std=c++17", "-pthread", "-I", INC] + flags + ["-o", out, SRC], check=True, capture_output=True, timeout=300)
    env = dict(os.environ, GNN_MLP_CONVERT_THREADS=threads, TSAN_OPTIONS="halt_on_error=1")
    r = subprocess.run([out], capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode == 0, r.stdout + r.stderr
    assert r.stdout.strip().startswith("ok 5")
    assert "ThreadSanitizer" not in r.stderr

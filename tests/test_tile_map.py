"""make_tile_map (csrc/tile_step_kernel.h): the host-built workgroup -> tile map of the two-launch step's tile kernel.
Host code only: compiled with hipcc and run here, no GPU.  Every tile exactly once, no idle entry in front of a live
one, tile counts balanced over the XCDs, and -- where there are enough small-layer tiles -- no CU shared by two
layer-0 tiles (the pair that set the kernel's duration before the map existed)."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "tests", "native", "tile_map_check.hip")


@pytest.fixture(scope="module")
def checker(tmp_path_factory):
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("no hipcc")
    out = str(tmp_path_factory.mktemp("tile_map") / "tile_map_check")
    subprocess.run([hipcc, "--offload-arch=gfx950", "-O1", "-std=c++17", "-o", out, SRC], check=True, capture_output=True, timeout=600)
    return out


def run(checker, cus, dims):
    r = subprocess.run([checker, str(cus)] + [str(d) for d in dims], check=True, capture_output=True, text=True, timeout=60)
    lines = r.stdout.strip().splitlines()
    head = dict(zip(lines[0].split()[0::2], map(int, lines[0].split()[1::2])))
    pack = dict(zip(lines[1].split()[0::2], map(int, lines[1].split()[1::2])))
    xcds = [dict(zip(l.split()[0::2], map(int, l.split()[1::2]))) for l in lines[2:]]
    return head, pack, xcds


@pytest.mark.parametrize("dims", [[784, 300, 100, 10], [784, 100, 50, 10], [300, 40, 10], [100, 64, 48, 32, 10], [1000, 512, 256, 16]])
def test_every_tile_once_and_balanced(checker, dims):
    head, pack, xcds = run(checker, 32, dims)
    assert head["live"] == head["expected"] and head["duplicates"] == 0
    assert head["grid"] % 8 == 0 and head["grid"] - head["live"] < 8 + 7   # at most one short XCD row of idle entries
    counts = [x["tiles"] for x in xcds]
    assert max(counts) - min(counts) <= 4           # (layer-0 rectangles differ by a row or a column of tiles)
    assert all(x["idle_before_live"] == 0 for x in xcds)
    assert pack["packed"] == (head["grid"] <= 640) and (not pack["packed"] or pack["roundtrip"] == 1)


def test_headline_shape_shares_no_cu_between_two_layer0_tiles(checker):
    head, pack, xcds = run(checker, 32, [784, 300, 100, 10])
    assert head["live"] == 284 and head["grid"] == 288
    assert sum(x["layer0_pairs"] for x in xcds) == 0

"""GPU: `bench.py --mode train` in a FRESH process (one child, a few steps).  The graphed training step once faulted only
there - a memset node inside the captured hipGraph raced with the multi-workgroup EMD auction that follows it - while every
in-process test passed, so this path gets its own test."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("graph", ["1", "0"])
def test_bench_train_fresh_process(graph):
    env = dict(os.environ, PF_BENCH_GRAPH=graph)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--mode", "train", "--steps", "4", "--warmup", "2"],
                         cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    line = [l for l in out.stdout.splitlines() if l.startswith("{")][-1]
    rec = json.loads(line)
    assert rec["unit"] == "patches/s" and rec["value"] > 0 and rec["steps"] == 4
    assert rec["loss"] == rec["loss"] and abs(rec["loss"]) < 1e3          # finite, sane
    assert "capture failed" not in out.stderr

"""bench.py's stdout contract (CPU): rank 0 prints ONE JSON line and nothing else, whatever the libraries underneath write to file descriptor 1
(RCCL announces its version there when a process group comes up)."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_stdout_carries_the_result_line_only():
    code = ("import os, sys; sys.path.insert(0, %r); import bench; bench.claim_stdout(); "
            "os.write(1, b'library banner on fd 1\\n'); print('python print after the claim'); "
            "bench.emit({'metric': 'm', 'value': 1.5}); os.write(1, b'more noise\\n')") % ROOT
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = r.stdout.splitlines()
    assert len(lines) == 1 and json.loads(lines[0]) == {"metric": "m", "value": 1.5}
    assert "library banner on fd 1" in r.stderr and "python print after the claim" in r.stderr and "more noise" in r.stderr


def test_emit_without_a_claim_still_prints_the_line():
    code = "import sys; sys.path.insert(0, %r); import bench; bench.emit({'k': [1, 2]})" % ROOT
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    assert [json.loads(ln) for ln in r.stdout.splitlines()] == [{"k": [1, 2]}]

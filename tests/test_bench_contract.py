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


def test_configs3_roofline_block_arithmetic():
    """The N > 1 line's `roofline` block (bench.c4_roofline): per-GPU fraction of the HBM roofline from SURVEY 8(d)'s bytes per frame, the
    exchange against 7 xGMI links, for both exchanges SURVEY 8(e) names (G = 14 -> 112 MB, G = 59 -> 472 MB at 2 M Gaussians)."""
    sys.path.insert(0, ROOT)
    import bench
    N, D, px = 2_000_000, 4_500_000, 640 * 480
    assert bench.frame_bytes(N, D, px, sh=True) == N * 832 + D * 160 + px * 48        # C3 render, 2.40 GB
    for sh, G in ((False, 14), (True, 59)):
        S = N * G * 4
        r = bench.c4_roofline(N, D, px, sh, 8, 64, 6e-3, 1.0, S, single_fps=1000.0)
        B = bench.frame_bytes(N, D, px, sh=sh)
        assert r["alg_bytes_per_keyframe"] == B and r["keyframes_per_gpu_per_step"] == 8
        assert abs(r["frame_frac"] - 8 * B / 6e-3 / 8e12) < 1e-5 and r["frac"] == r["frame_frac"] and r["bound"] == "hbm"
        assert r["exchange_buffer_bytes"] == S == {14: 112_000_000, 59: 472_000_000}[G]
        assert r["exchange_wire_bytes_per_rank"] == int(2 * S * 7 / 8)
        assert abs(r["exchange_frac_xgmi"] - (2 * S * 7 / 8) / 1e-3 / (7 * 153e9)) < 1e-4
        assert abs(r["single_gpu_frame_frac"] - 1000.0 * B / 8e12) < 1e-5
    one = bench.c4_roofline(N, D, px, False, 1, 64, 32e-3, None, None)
    assert "exchange_frac_xgmi" not in one and one["keyframes_per_gpu_per_step"] == 64


def test_profiled_kernels_merges_the_rocprof_table_with_the_counter_summary():
    """bench.profiled_kernels: the committed rocprofv3 kernel table of a leg + its PMC summary -> per-kernel duration, counter bytes (FETCH doubled
    + WRITE), fraction of 8 TB/s.  Checked on the committed 2 M profile (any round's: the newest is taken)."""
    sys.path.insert(0, ROOT)
    import bench
    pr = bench.profiled_kernels("2m")
    assert pr and any(s.endswith("_2m_kernel_stats.csv") for s in pr["source"]) and any(s.endswith("_2m_pmc_summary.json") for s in pr["source"])
    by, us = bench.counters_for(pr, "blend_backward_kernel<false,1,false>")
    assert by and us and 100 < us < 400 and 2e8 < by < 6e8                      # ~180 us, ~367 MB at 2 M Gaussians / 4.84 M instances
    e = next(v for k, v in pr["kernels"].items() if k.startswith("blend_backward_kernel<"))
    assert abs(e["frac_hbm_pmc"] - by / (us * 1e-6) / 8e12) < 1e-3 and 0.3 < e["valu_busy"] < 1.0
    assert bench.profiled_kernels("no_such_leg") is None and bench.counters_for(None, "x") == (None, None)

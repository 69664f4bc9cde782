"""Development helper: where do the backward blend's vector instructions go?  Replays the CONTROL FLOW of blend_backward_kernel (blend.hip) in numpy
from the integer artefacts of a debug forward -- per quadrant walker: chunks visited / skipped, per-block list lengths, phase-A iterations, phase-B
batches (and how full they are), flush groups -- and prices them with the instruction counts read off the kernel's ISA.
GPU box: N=2000000 SH=3 python scripts/exp/bwd_work.py"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from activesplat_amd import GaussianRasterizer, setup_camera  # noqa: E402
from activesplat_amd import synthetic as syn  # noqa: E402
from tests import util  # noqa: E402

dev = torch.device("cuda")
W, H = int(os.environ.get("W", 640)), int(os.environ.get("H", 480))
N = int(os.environ.get("N", 2_000_000))
cam = setup_camera(W, H, syn.intrinsics(W, H), np.eye(4), device=dev)
rv = {k: v.to(dev) for k, v in syn.activate(syn.make_params(N, W, H, seed=0)).items()}
util.run_product(cam, rv)
art = util.artefacts()
nc = art["n_contrib"].astype(np.int64)
geom, ranges, plist = art["geom"], art["ranges"].astype(np.int64), art["point_list"].astype(np.int64)
gx, gy = (W + 15) // 16, (H + 15) // 16
mx, my, ex, ey = geom[:, 0], geom[:, 1], geom[:, 10], geom[:, 11]

SLACKS = ((4, 0), (4, 32), (4, 30), (2, 32), (4, 48), (0, 32))
pol = {sl: dict(rounds=0, positions=0, batches=0, iters=0, flush_groups=0, scans=0, misses=0, pairs=0) for sl in SLACKS}
PIECES = int(os.environ.get("PIECES", 3))          # chained pieces per quadrant walk (each ends with a partial round)
rt = dict(rounds=0, positions=0, pairs=0, batches=0, iters=0, flush_groups=0, scans=0)
# DECOUPLED STREAMS (priced before building): per-stream ring queues -- a stream whose list for one round is short goes on with the
# next round's entries instead of idling to the longest list.  Upper bound of what that buys: per walker (piece) the trip count becomes the LONGEST
# stream's total over the whole piece instead of the sum over rounds of each round's longest list; scans, gathers per batch and flushes stay.
dec = dict(positions=0, pairs=0, batches=0, iters=0)
tot = dict(walkers=0, chunks=0, skipped=0, iters=0, batches=0, positions=0, pairs=0, flush_groups=0, any_records=0, records=0, full_batches=0)
hist_ntrips = np.zeros(66, np.int64)
for t in range(gx * gy):
    tx, ty = t % gx, t // gx
    ids = plist[ranges[t, 0]:ranges[t, 1]]
    if ids.size == 0:
        continue
    x, y, ax, ay = mx[ids], my[ids], ex[ids], ey[ids]
    for q in range(4):
        qx0, qy0 = tx * 16 + (q & 1) * 8, ty * 16 + (q >> 1) * 8
        blk = nc[qy0:qy0 + 8, qx0:qx0 + 8]
        if blk.size == 0:
            continue
        wmax = int(blk.max())
        if wmax == 0:
            continue
        tot["walkers"] += 1
        nch = (wmax - 1) // 64 + 1
        L = nch * 64
        hits = np.zeros((4, L), bool)
        for r in range(4):
            x0, y0 = qx0 + (r & 1) * 4, qy0 + (r >> 1) * 4
            sub = nc[y0:y0 + 4, x0:x0 + 4]
            rm = int(sub.max()) if sub.size else 0
            n = min(wmax, ids.size)
            h = (ax[:n] >= 0) & (x[:n] + ax[:n] >= x0) & (x[:n] - ax[:n] <= x0 + 3) & (y[:n] + ay[:n] >= y0) & (y[:n] - ay[:n] <= y0 + 3)
            cpos = (np.arange(n) // 64) * 64
            hits[r, :n] = h & (cpos < rm)
        hc = hits.reshape(4, nch, 64)
        nr = hc.sum(2)                                   # [4, nch]
        ntr = nr.max(0)
        anyc = hc.any(0).sum(1)                          # records in the union list per chunk
        vis = ntr > 0
        tot["chunks"] += int(vis.sum()); tot["skipped"] += int((~vis).sum())
        tot["records"] += L
        nt = ntr[vis]
        np.add.at(hist_ntrips, np.minimum(nt, 65), 1)
        b = (nt + 15) // 16
        tot["batches"] += int(b.sum())
        tot["full_batches"] += int((nt // 16).sum())
        tot["positions"] += int(nt.sum())
        tot["iters"] += int(((nt // 16) * 8 + ((nt % 16) + 1) // 2).sum())
        tot["pairs"] += int(nr.sum())
        tot["flush_groups"] += int(((anyc[vis] + 5) // 6).sum())
        tot["any_records"] += int(anyc.sum())
        # the same walk with COMPACTED staging: hit records of successive chunks are staged until the next chunk's would not fit in 64 slots
        order = list(range(nch - 1, -1, -1))                     # deepest chunk first
        bounds = [nch - (nch * (p + 1)) // PIECES for p in range(PIECES)]      # piece p covers chunks [bounds[p], previous bound)
        hi = nch
        for lo in bounds:
            S_ = nr[:, lo:hi].sum(1)
            if int(S_.max()) > 0:
                nt_ = int(S_.max())
                dec["positions"] += nt_; dec["pairs"] += int(S_.sum()); dec["batches"] += (nt_ + 15) // 16
                dec["iters"] += (nt_ // 16) * 8 + ((nt_ % 16) + 1) // 2
            cnt = 0; acc = np.zeros(4, np.int64)
            def close():
                global rt
                nt_ = int(acc.max())
                if nt_ == 0: return
                rt["rounds"] += 1; rt["positions"] += nt_; rt["pairs"] += int(acc.sum()); rt["batches"] += (nt_ + 15) // 16
                rt["iters"] += (nt_ // 16) * 8 + ((nt_ % 16) + 1) // 2; rt["flush_groups"] += (cnt + 5) // 6
            for chn in range(hi - 1, lo - 1, -1):
                a = int(anyc[chn])
                if a == 0: continue
                rt["scans"] += 1
                if cnt + a > 64:
                    close(); cnt = 0; acc[:] = 0
                cnt += a; acc += nr[:, chn]
            close()
            # policy P(slack): stage, then process at once if another chunk like this one would not fit; a chunk that does not fit all the same is
            # SPLIT (the deepest hits fill the round, the chunk is scanned again for the rest)
            for sl in SLACKS:
                d = pol[sl]
                cnt = 0; acc = np.zeros(4, np.int64)
                def close2():
                    nt_ = int(acc.max())
                    if nt_ == 0: return
                    d["rounds"] += 1; d["positions"] += nt_; d["pairs"] += int(acc.sum()); d["batches"] += (nt_ + 15) // 16
                    d["iters"] += (nt_ // 16) * 8 + ((nt_ % 16) + 1) // 2; d["flush_groups"] += (cnt + 5) // 6
                for chn in range(hi - 1, lo - 1, -1):
                    a = int(anyc[chn])
                    if a == 0: continue
                    d["scans"] += 1
                    if cnt + a > 64:
                        # split: the deepest (64 - cnt) hit records of the chunk complete the round
                        d["misses"] += 1; d["scans"] += 1
                        take = 64 - cnt
                        hm = hc[:, chn, :]                               # [4, 64] lane 63 deepest
                        anyl = hm.any(0)
                        idx = np.nonzero(anyl)[0][::-1]                  # hit lanes, deepest first
                        first = np.zeros(64, bool); first[idx[:take]] = True
                        acc += (hm & first).sum(1); cnt = 64
                        close2(); cnt = 0; acc[:] = 0
                        acc += (hm & ~first).sum(1); cnt = a - take
                    else:
                        cnt += a; acc += nr[:, chn]
                    if cnt + a + sl[0] > 64 or (sl[1] and int((acc + nr[:, chn]).max()) > sl[1]):
                        close2(); cnt = 0; acc[:] = 0
                close2()
            hi = lo
print(tot)
print("Gaussians %d, visible (tiles touched > 0) %d = %.1f %%, contributing to some quadrant's staging %d" % (N, int((art["tiles_touched"] > 0).sum()), 100.0 * float((art["tiles_touched"] > 0).mean()),
      tot["any_records"]))
c = tot
print("per visited chunk: positions %.1f (lane-rows filled %.2f), batches %.2f (%.0f %% full), union records %.1f, flush groups %.1f" % (
    c["positions"] / c["chunks"], c["pairs"] / (4.0 * c["positions"]), c["batches"] / c["chunks"], 100.0 * c["full_batches"] / c["batches"],
    c["any_records"] / c["chunks"], c["flush_groups"] / c["chunks"]))
print("ntrips histogram (visited chunks):", {int(i): int(v) for i, v in enumerate(hist_ntrips) if v})
# instruction prices (VALU wave-instructions), read off the ISA of blend_backward_kernel<false,1,false> (scripts/exp/README in profiles/README.md)
A_ITER, B_BATCH, GATHER, CHUNK, SKIP, FLUSH_SETUP, FLUSH_GROUP = 62, 150, 52, 130, 45, 25, 8
parts = dict(phase_A=c["iters"] * A_ITER, phase_B=c["batches"] * B_BATCH, gather=c["batches"] * GATHER, chunk=c["chunks"] * CHUNK, skipped=c["skipped"] * SKIP,
             flush=c["chunks"] * FLUSH_SETUP + c["flush_groups"] * FLUSH_GROUP)
s = sum(parts.values())
print("modelled VALU wave-instructions: %.1f M" % (s / 1e6), {k: "%.1f M (%.0f %%)" % (v / 1e6, 100.0 * v / s) for k, v in parts.items()})
print("compacted staging:", rt, "row fill %.2f, positions per round %.1f, batches per round %.2f" % (rt["pairs"] / (4.0 * rt["positions"]), rt["positions"] / rt["rounds"],
      rt["batches"] / rt["rounds"]))
SCAN, ROUND = 50, 80
parts = dict(phase_A=rt["iters"] * A_ITER, phase_B=rt["batches"] * B_BATCH, gather=rt["batches"] * GATHER, scan=rt["scans"] * SCAN, round=rt["rounds"] * ROUND,
             flush=rt["rounds"] * FLUSH_SETUP + rt["flush_groups"] * FLUSH_GROUP)
s2 = sum(parts.values())
print("modelled with compacted staging: %.1f M (%.3f x)" % (s2 / 1e6, s2 / s), {k: "%.1f M" % (v / 1e6) for k, v in parts.items()})
for sl in SLACKS:
    d = pol[sl]
    parts = dict(phase_A=d["iters"] * A_ITER, phase_B=d["batches"] * B_BATCH, gather=d["batches"] * GATHER, scan=d["scans"] * SCAN, round=d["rounds"] * ROUND,
                 flush=d["rounds"] * FLUSH_SETUP + d["flush_groups"] * FLUSH_GROUP)
    s3 = sum(parts.values())
    print("policy slack %s: rounds %d, records per round %.1f, batches per round %.2f, row fill %.2f, split chunks %d (%.1f %% of scans): %.1f M (%.3f x)" % (
        sl, d["rounds"], c["any_records"] / d["rounds"], d["batches"] / d["rounds"], d["pairs"] / (4.0 * d["positions"]), d["misses"], 100.0 * d["misses"] / d["scans"], s3 / 1e6, s3 / s))

# decoupled streams, priced on top of the shipped policy (4, 0): its scans / rounds / flushes, the decoupled trip counts, and RING instructions of
# bookkeeping per scan (per-stream head / tail, a record's "all four streams done" count before its slot is flushed and freed)
d = pol[(4, 0)]
base = dict(phase_A=d["iters"] * A_ITER, phase_B=d["batches"] * B_BATCH, gather=d["batches"] * GATHER, scan=d["scans"] * SCAN, round=d["rounds"] * ROUND,
            flush=d["rounds"] * FLUSH_SETUP + d["flush_groups"] * FLUSH_GROUP)
sb = sum(base.values())
for RING in (0, 40, 80):
    parts = dict(base, phase_A=dec["iters"] * A_ITER, phase_B=dec["batches"] * B_BATCH, gather=dec["batches"] * GATHER, ring=d["scans"] * RING)
    sd_ = sum(parts.values())
    print("decoupled streams (upper bound, %d ring instructions per scan): row fill %.3f (shipped %.3f), positions %d (shipped %d): %.1f M = %.3f x the shipped %.1f M" % (
        RING, dec["pairs"] / (4.0 * dec["positions"]), d["pairs"] / (4.0 * d["positions"]), dec["positions"], d["positions"], sd_ / 1e6, sd_ / sb, sb / 1e6))

#!/usr/bin/env python3
"""probe_order.py — dev probe: the bench frame handed to one GPU as one region (the shipped way: blocks bottom-up), as 4-row strips bottom-up and as 4-row strips TOP-DOWN
(the expensive middle of the frame early, the uniform ground plane last): kernel time and how the waves finish."""
import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import numpy as np
from __graft_entry__ import load_package, BUILT
import bench
pkg = load_package(); api = pkg.api; abi = pkg.abi
W = bench.WORKLOAD
w, h, spp, b = W["width"], W["height"], W["samples"], W["bounces"]
ctx = api.Context(0); ctx.set_option(abi.OPT_COUNTER_LEVEL, 1); ctx.set_option(abi.OPT_WAVE_STATS, 1)
ctx.upload(api.Scene(os.path.join(BUILT, W["blob"] + ".blob")))
fb = ctx.framebuffer(w, h)
R = int(os.environ.get("ROWS", "8"))
cases = [("one region", [(0, 0, w, h)]), (f"{R}-row strips bottom-up", [(0, y, w, min(y + R, h)) for y in range(0, h, R)]),
         (f"{R}-row strips top-down", [(0, y, w, min(y + R, h)) for y in reversed(range(0, h, R))]),
         ("three bands: top third, middle, bottom", [(0, 2 * h // 3, w, h), (0, h // 3, w, 2 * h // 3), (0, 0, w, h // 3)]),
         ("three bands: middle, top, bottom", [(0, h // 3, w, 2 * h // 3), (0, 2 * h // 3, w, h), (0, 0, w, h // 3)])]
for world, rank in ((1, 0), (8, 0)):
    for label, tiles in cases:
        if world > 1:
            if "strips" not in label: continue
            tiles = [(0, y, w, min(y + 4, h)) for y in (range(0, h, 4) if "bottom-up" in label else reversed(range(0, h, 4)))][rank::world]       # a rank's share: 4-row strips
        best = None
        for rep in range(3):
            ctx.clear(fb, w, h); ctx.reset_counters(); ctx.render_tiles(fb, w, h, spp, b, tiles); ctx.synchronize()
            ms = ctx.kernel_time_ms()[0]; ws = ctx.wave_stats(); busy = ws[:, 0] / 1e5
            if best is None or ms < best[0]: best = (ms, busy.mean(), np.median(busy), busy.min(), busy.max())
        print(f"world {world} {label}: {best[0]:.2f} ms; wave busy mean {best[1]:.2f} median {best[2]:.2f} min {best[3]:.2f} max {best[4]:.2f}", flush=True)

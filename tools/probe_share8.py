#!/usr/bin/env python3
"""probe_share8.py — dev probe: one rank's share of the bench frame at world size 8 (every 8th 4-row strip), all 256 passes: kernel time,
wave busy time (mean / max) and units per wave for several unit sizes — where the 1/8 share loses time against 1/8 of the full frame."""
import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from __graft_entry__ import load_package, BUILT
import bench
pkg = load_package(); api = pkg.api; abi = pkg.abi
W = bench.WORKLOAD
ctx = api.Context(0); ctx.set_option(abi.OPT_COUNTER_LEVEL, 1); ctx.set_option(abi.OPT_WAVE_STATS, 1)
ctx.upload(api.Scene(os.path.join(BUILT, W["blob"] + ".blob")))
w, h, spp, b = W["width"], W["height"], W["samples"], W["bounces"]
fb = ctx.framebuffer(w, h)
def run(tiles, label):
    best = None
    for rep in range(3):
        ctx.clear(fb, w, h); ctx.reset_counters(); ctx.render_tiles(fb, w, h, spp, b, tiles); ctx.synchronize()
        ms = ctx.kernel_time_ms()[0]; rays = ctx.counters()["rays"]; ws = ctx.wave_stats()
        if best is None or ms < best[0]: best = (ms, rays, ws[:, 0].mean() / 1e5, ws[:, 0].max() / 1e5, ws[:, 0].min() / 1e5, ws[:, 1].mean())
    print(f"{label}: {best[0]:.2f} ms {best[1]/best[0]/1e3:.0f} Mray/s; wave busy mean {best[2]:.2f} max {best[3]:.2f} min {best[4]:.2f} ms; units/wave {best[5]:.1f}", flush=True)
    return best[0]
full = pkg.render.owned_tiles(w, h, 64, 64, 1, 0, 1)
t1 = run(full, "world 1 (full frame)")
for items, upw, tail, tail2 in ((2048, 8, 16, 0), (2048, 8, 16, 2), (2048, 8, 16, 4), (2048, 8, 16, 8), (2048, 8, 30, 8), (2048, 8, 30, 4), (4096, 8, 30, 8), (2048, 8, 24, 6)):
    ctx.set_option(abi.OPT_UNIT_ITEMS, items); ctx.set_option(abi.OPT_UNITS_PER_WAVE, upw); ctx.set_option(abi.OPT_TAIL_PERCENT, tail | ((tail2 + 1) << 8))
    full_ms = run(full, f"world 1 items {items} tail {tail}/{tail2}%")
    worst = 0
    for rank in (0, 7):
        worst = max(worst, run(pkg.render.owned_tiles(w, h, 64, 64, 1, rank, 8), f"world 8 rank {rank} items {items} units/wave>={upw} tail {tail}/{tail2}%"))
    print(f"   -> ceiling {full_ms / worst:.2f}x (vs the same settings at world 1)", flush=True)

#!/usr/bin/env python3
"""probe_rank_share.py — dev probe: kernel time of one rank's tile share of the bench frame for world sizes 1, 2, 4, 8
(all on this one GPU): the strong-scaling ceiling of bench.py before reduce / launch overheads."""
import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from __graft_entry__ import load_package, BUILT
import bench
pkg = load_package(); api = pkg.api; abi = pkg.abi
W = bench.WORKLOAD
ctx = api.Context(0); ctx.set_option(abi.OPT_COUNTER_LEVEL, 1)
ctx.upload(api.Scene(os.path.join(BUILT, W["blob"] + ".blob")))
w, h, spp, b = W["width"], W["height"], W["samples"], W["bounces"]
fb = ctx.framebuffer(w, h)
t1 = None
for world in (1, 2, 4, 8):
    worst = 0.0; rays_total = 0
    for rank in sorted({0, world // 2, world - 1}):
        tiles = pkg.render.owned_tiles(w, h, W["tile"][0], W["tile"][1], W["tile_order"], rank, world)
        best = 1e9
        for rep in range(2):
            ctx.clear(fb, w, h); ctx.reset_counters(); ctx.render_tiles(fb, w, h, spp, b, tiles); ctx.synchronize()
            best = min(best, ctx.kernel_time_ms()[0])
        worst = max(worst, best)
    if world == 1: t1 = worst
    print(f"world {world}: slowest sampled rank {worst:.2f} ms -> speedup ceiling {t1/worst:.2f}x ({100*t1/worst/world:.0f}% efficiency)", flush=True)

#!/usr/bin/env python3
"""probe_share8_kernels.py — dev probe: the full bench frame and one rank's 1/8 share (every 8th 4-row strip) with the kernel forms a build holds
(0 = k_pathtrace, one unit at a time; 2 = k_pathtrace_roll, the default): where do the small work units of a small share lose their time?"""
import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from __graft_entry__ import load_package, BUILT
import bench
pkg = load_package(); api = pkg.api; abi = pkg.abi
W = bench.WORKLOAD
w, h, spp, b = W["width"], W["height"], W["samples"], W["bounces"]
kernels = [int(k) for k in (sys.argv[1:] or ["0", "2"])]
for kern in kernels:
    ctx = api.Context(0); ctx.set_option(abi.OPT_COUNTER_LEVEL, 1); ctx.set_option(abi.OPT_WAVE_STATS, 1)
    if kern: ctx.set_option(abi.OPT_KERNEL, kern)
    ctx.upload(api.Scene(os.path.join(BUILT, W["blob"] + ".blob")))
    fb = ctx.framebuffer(w, h)
    def run(tiles, label):
        best = None
        for rep in range(3):
            ctx.clear(fb, w, h); ctx.reset_counters(); ctx.render_tiles(fb, w, h, spp, b, tiles); ctx.synchronize()
            ms = ctx.kernel_time_ms()[0]; rays = ctx.counters()["rays"]; ws = ctx.wave_stats()
            if best is None or ms < best[0]: best = (ms, rays, ws[:, 0].mean() / 1e5, ws[:, 0].max() / 1e5, ws[:, 0].min() / 1e5, ws[:, 1].mean())
        print(f"kernel {kern} {label}: {best[0]:.2f} ms {best[1]/best[0]/1e3:.0f} Mray/s; wave busy mean {best[2]:.2f} max {best[3]:.2f} min {best[4]:.2f} ms; units/wave {best[5]:.1f}", flush=True)
        return best[0]
    full = run(pkg.render.owned_tiles(w, h, 64, 64, 1, 0, 1), "world 1")
    for world in (2, 4, 8):
        worst = max(run(pkg.render.owned_tiles(w, h, 64, 64, 1, rank, world), f"world {world} rank {rank}") for rank in (0, world - 1))
        print(f"   kernel {kern}: world {world} ceiling {full / worst:.2f}x", flush=True)
    ctx.close()

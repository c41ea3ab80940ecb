#!/usr/bin/env python3
"""probe_small_jobs.py — dev probe: unit granularity vs the end-of-kernel tail when a GPU gets little work (the per-rank
share of cfg2 at 8 ranks = 32 spp of the frame): kernel time, mean / max wave busy time."""
import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from __graft_entry__ import load_package, BUILT
pkg = load_package(); api = pkg.api; abi = pkg.abi
ctx = api.Context(0)
ctx.set_option(abi.OPT_COUNTER_LEVEL, 1); ctx.set_option(abi.OPT_WAVE_STATS, 1)
scene = api.Scene(os.path.join(BUILT, "cfg2_hdr.blob")); ctx.upload(scene)
w, h, b = 1280, 720, 8
fb = ctx.framebuffer(w, h)
for spp in (32, 256):
    for items, upw, tail in ((2048, 8, 0), (2048, 8, 16), (1024, 16, 16), (1024, 16, 24), (512, 32, 16)):
        ctx.set_option(abi.OPT_UNIT_ITEMS, items); ctx.set_option(abi.OPT_UNITS_PER_WAVE, upw); ctx.set_option(abi.OPT_TAIL_PERCENT, tail)
        best = None
        for rep in range(2):
            ctx.clear(fb, w, h); ctx.reset_counters(); ctx.render_region(fb, w, h, spp, b); ctx.synchronize()
            ms = ctx.kernel_time_ms()[0]; rays = ctx.counters()["rays"]; ws = ctx.wave_stats()
            if best is None or ms < best[0]: best = (ms, rays, ws[:, 0].mean() / 1e5, ws[:, 0].max() / 1e5, ws[:, 1].mean())
        print(f"spp {spp} items {items} units/wave>={upw} tail {tail}%: {best[0]:.2f} ms {best[1]/best[0]/1e3:.0f} Mray/s; wave busy mean {best[2]:.2f} max {best[3]:.2f} ms; units/wave {best[4]:.1f}", flush=True)

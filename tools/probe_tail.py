#!/usr/bin/env python3
"""probe_tail.py — dev probe: taper of the work units (CRH_OPT_TAIL_PERCENT: last p1 % of the pixels in quarter blocks, last p2 % in sixteenth
blocks) on full frames submitted as one region; kernel time, per-wave busy time (mean / max)."""
import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from __graft_entry__ import load_package, BUILT
pkg = load_package(); api = pkg.api; abi = pkg.abi
ctx = api.Context(0); ctx.set_option(abi.OPT_COUNTER_LEVEL, 1); ctx.set_option(abi.OPT_WAVE_STATS, 1)
for name, w, h, spp, b in (("cfg2_hdr", 1280, 720, 256, 8), ("cfg4_statues", 3840, 2160, 8, 30), ("soup_1m", 2560, 1440, 16, 8), ("cfg3_venus", 1920, 1080, 32, 32)):
    ctx.upload(api.Scene(os.path.join(BUILT, name + ".blob")))
    fb = ctx.framebuffer(w, h)
    for tail, tail2 in [tuple(int(v) for v in t.split("/")) for t in (os.environ.get("TAILS") or "16/4,8/2,12/3,24/6,30/4,30/8,40/10,50/12,0/0").split(",")]:
        ctx.set_option(abi.OPT_TAIL_PERCENT, tail | ((tail2 + 1) << 8))
        best = None
        for rep in range(3):
            ctx.clear(fb, w, h); ctx.reset_counters(); ctx.render_region(fb, w, h, spp, b); ctx.synchronize()
            ms = ctx.kernel_time_ms()[0]; ws = ctx.wave_stats()
            if best is None or ms < best[0]: best = (ms, ws[:, 0].mean() / 1e5, ws[:, 0].max() / 1e5)
        print(f"{name} tail {tail}/{tail2}%: {best[0]:.2f} ms; wave busy mean {best[1]:.2f} max {best[2]:.2f}", flush=True)

#!/usr/bin/env python3
"""probe_sched_sweep.py — dev probe: scheduler parameter sweep on cfg2, the 1M soup, statues and venus.
   args: node,tri,ctrl,swapMin[,fillTo,runNum,triInRun,ctrlInRun] ..."""
import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from __graft_entry__ import load_package, BUILT
pkg = load_package(); api = pkg.api; abi = pkg.abi
ctx = api.Context(0)
ctx.set_option(abi.OPT_COUNTER_LEVEL, 1)
SWEEP = [tuple(int(v) for v in s.split(",")) for s in sys.argv[1:]] or [(70, 160, 120, 16)]
for name, w, h, spp, b in (("cfg2_hdr", 1280, 720, 128, 8), ("soup_1m", 2560, 1440, 16, 8), ("cfg4_statues", 3840, 2160, 4, 30), ("cfg3_venus", 1920, 1080, 16, 32)):
    scene = api.Scene(os.path.join(BUILT, name + ".blob"))
    ctx.upload(scene)
    fb = ctx.framebuffer(w, h)
    base = None
    for cfg in SWEEP:
        ctx.set_sched(*cfg)
        best = None
        for rep in range(2):
            ctx.clear(fb, w, h); ctx.reset_counters()
            ctx.render_region(fb, w, h, spp, b); ctx.synchronize()
            ms = ctx.kernel_time_ms()[0]; rays = ctx.counters()["rays"]
            best = ms if best is None else min(best, ms)
        base = base or best
        print(f"{name} sched {cfg}: {best:.1f} ms {rays/best/1e3:.0f} Mray/s  {base/best:.3f}", flush=True)

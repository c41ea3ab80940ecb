#!/usr/bin/env python3
"""pcs_render.py NAME W H SPP BOUNCES REPS — dev: render a built scene a few times (the workload under tools/pc_sample.sh)."""
import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from __graft_entry__ import load_package, BUILT
pkg = load_package(); api = pkg.api; abi = pkg.abi
name, w, h, spp, b, reps = sys.argv[1], *map(int, sys.argv[2:7])
ctx = api.Context(0)
ctx.set_option(abi.OPT_COUNTER_LEVEL, 1)
ctx.upload(api.Scene(name if name.endswith(".blob") else os.path.join(BUILT, name + ".blob")))
fb = ctx.framebuffer(w, h)
for _ in range(reps):
    ctx.clear(fb, w, h); ctx.reset_counters()
    ctx.render_region(fb, w, h, spp, b); ctx.synchronize()
    print(name, ctx.kernel_time_ms()[0], "ms", ctx.counters()["rays"], "rays", flush=True)

#!/usr/bin/env python3
"""probe_scenes.py — dev probe: timed kernel (counter level 1) on the three reference scenes."""
import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from __graft_entry__ import load_package, BUILT
pkg = load_package(); api = pkg.api; abi = pkg.abi
ctx = api.Context(0)
ctx.set_option(abi.OPT_COUNTER_LEVEL, 1)
if os.environ.get('CRH_BPC'): ctx.set_option(abi.OPT_BLOCKS_PER_CU, int(os.environ['CRH_BPC']))
for name, w, h, spp, b in (("cfg2_hdr", 1280, 720, 256, 8), ("cfg3_venus", 1920, 1080, 64, 32), ("cfg4_statues", 3840, 2160, 16, 30), ("soup_1m", 2560, 1440, 32, 8)):
    scene = api.Scene(os.path.join(BUILT, name + ".blob"))
    ctx.upload(scene)
    fb = ctx.framebuffer(w, h)
    for items, upw in ((2048, 8),):
        ctx.set_option(abi.OPT_UNIT_ITEMS, items); ctx.set_option(abi.OPT_UNITS_PER_WAVE, upw)
        best = None
        for rep in range(2):
            ctx.clear(fb, w, h); ctx.reset_counters()
            ctx.render_region(fb, w, h, spp, b); ctx.synchronize()
            ms = ctx.kernel_time_ms()[0]; rays = ctx.counters()["rays"]
            best = ms if best is None else min(best, ms)
        print(f"{name} items {items} units/wave>={upw}: {best:.1f} ms {rays/best/1e3:.0f} Mray/s", flush=True)

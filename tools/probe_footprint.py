#!/usr/bin/env python3
"""probe_footprint.py — dev probe: work-unit size (CRH_OPT_UNIT_ITEMS: samples staged per wave) x paths in flight (fillTo) sweep."""
import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from __graft_entry__ import load_package, BUILT
pkg = load_package(); api = pkg.api; abi = pkg.abi
ctx = api.Context(0)
ctx.set_option(abi.OPT_COUNTER_LEVEL, 1)
UNITS = [int(v) for v in (os.environ.get("UNITS") or "1024,1536,2048,3072,4096").split(",")]
FILLS = [int(v) for v in (os.environ.get("FILLS") or "128,160,192").split(",")]
for name, w, h, spp, b in (("cfg2_hdr", 1280, 720, 256, 8), ("soup_1m", 2560, 1440, 16, 8), ("cfg4_statues", 3840, 2160, 8, 30)):
    scene = api.Scene(os.path.join(BUILT, name + ".blob"))
    ctx.upload(scene)
    fb = ctx.framebuffer(w, h)
    for u in UNITS:
        ctx.set_option(abi.OPT_UNIT_ITEMS, u)
        for f in FILLS:
            ctx.set_sched(70, 160, 120, 16, fill_to=f)
            best = None
            for rep in range(2):
                ctx.clear(fb, w, h); ctx.reset_counters()
                ctx.render_region(fb, w, h, spp, b); ctx.synchronize()
                ms = ctx.kernel_time_ms()[0]; rays = ctx.counters()["rays"]
                best = ms if best is None else min(best, ms)
            print(f"{name} unit {u} fill {f}: {best:.1f} ms {rays/best/1e3:.0f} Mray/s", flush=True)

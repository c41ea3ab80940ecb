#!/usr/bin/env python3
"""probe_tile_order.py — dev probe: one dispatch over the reference's tile list (each order) vs over one full-frame region."""
import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from __graft_entry__ import load_package, BUILT
pkg = load_package(); api = pkg.api; abi = pkg.abi; T = pkg.tiles
ctx = api.Context(0)
ctx.set_option(abi.OPT_COUNTER_LEVEL, 1)
for name, w, h, spp, b in (("cfg2_hdr", 1280, 720, 256, 8), ("cfg4_statues", 3840, 2160, 8, 30)):
    ctx.upload(api.Scene(os.path.join(BUILT, name + ".blob")))
    fb = ctx.framebuffer(w, h)
    cases = [("region", None)]
    for oname, order in (("from_middle", T.ORDER_FROM_MIDDLE), ("top_to_bottom", T.ORDER_TOP_TO_BOTTOM), ("normal", T.ORDER_NORMAL), ("random", T.ORDER_RANDOM)):
        for ts in (64, 128):
            cases.append((f"tiles {ts} {oname}", T.quantize_image(w, h, ts, ts, order)))
    cases.append(("strips of 4 rows", [(0, y, w, min(y + 4, h)) for y in range(0, h, 4)]))
    for label, tiles in cases:
        best = None
        for rep in range(3):
            ctx.clear(fb, w, h); ctx.reset_counters()
            if tiles is None: ctx.render_region(fb, w, h, spp, b)
            else: ctx.render_tiles(fb, w, h, spp, b, tiles)
            ctx.synchronize()
            ms = ctx.kernel_time_ms()[0]
            best = ms if best is None else min(best, ms)
        print(f"{name} {label}: {best:.2f} ms {ctx.counters()['rays']/best/1e3:.0f} Mray/s", flush=True)

#!/usr/bin/env python3
"""gpu_probe4.py — dev probe: where a wave spends its time (setup / BVH walk / shading), counting kernel."""
import os, sys
import numpy as np
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from __graft_entry__ import load_package, BUILT
pkg = load_package(); api = pkg.api; abi = pkg.abi
ctx = api.Context(0)
import itertools
SCHEDS = [(70, 120, 30, 32, 1024), (70, 120, 30, 48, 1024), (70, 100, 30, 40, 1024), (70, 160, 40, 40, 1024)]
for name, w, h, spp, b in (("cfg2_hdr", 1280, 720, 256, 8), ("cfg3_venus", 1920, 1080, 64, 32), ("soup_1m", 2560, 1440, 32, 8)):
    scene = api.Scene(os.path.join(BUILT, name + ".blob"))
    ctx.upload(scene)
    fb = ctx.framebuffer(w, h)
    for sc in SCHEDS:
        ctx.set_sched(*sc[:4]); ctx.set_option(abi.OPT_UNIT_ITEMS, sc[4])
        ctx.set_option(abi.OPT_COUNTER_LEVEL, 1)
        ctx.clear(fb, w, h); ctx.reset_counters()
        ctx.render_region(fb, w, h, spp, b); ctx.synchronize()
        ms1 = ctx.kernel_time_ms()[0]; rays = ctx.counters()["rays"]
        ctx.set_option(abi.OPT_COUNTER_LEVEL, 2)
        ctx.clear(fb, w, h); ctx.reset_counters()
        ctx.render_region(fb, w, h, spp, b); ctx.synchronize()
        c = ctx.counters(); t = ctx.phase_ticks()
        tot = max(t["setup"] + t["traverse"] + t["shade"], 1)
        wb = c["rays"] / 64.0
        print(name, sc, f"{ms1:.1f} ms {rays/ms1/1e3:.0f} Mray/s |", {k: f"{100*t[k]/tot:.0f}%" for k in ("setup", "traverse", "shade")},
              "| per wave-bounce: node %.1f tri %.1f ctrl %.1f serve %.2f | lanes/step: node %.1f serve %.1f" % (
                  t["w_node"] / wb, t["w_tri"] / wb, t["w_ctrl"] / wb, t["w_shade"] / wb,
                  t["u_node"] / max(t["w_node"], 1), t["u_shade"] / max(t["w_shade"], 1)), flush=True)

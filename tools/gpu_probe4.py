#!/usr/bin/env python3
"""gpu_probe4.py — dev probe: where a wave spends its time (setup / BVH walk / shading), counting kernel."""
import os, sys
import numpy as np
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from __graft_entry__ import load_package, BUILT
pkg = load_package(); api = pkg.api; abi = pkg.abi
ctx = api.Context(0)
for name, w, h, spp, b in (("cfg2_hdr", 1280, 720, 64, 8), ("cfg3_venus", 1920, 1080, 16, 32), ("soup_1m", 2560, 1440, 8, 8)):
    scene = api.Scene(os.path.join(BUILT, name + ".blob"))
    ctx.upload(scene)
    fb = ctx.framebuffer(w, h)
    ctx.set_option(abi.OPT_COUNTER_LEVEL, 2)
    ctx.reset_counters()
    ctx.render_region(fb, w, h, spp, b); ctx.synchronize()
    ms = ctx.kernel_time_ms()[0]; c = ctx.counters(); t = ctx.phase_ticks()
    tot = t["setup"] + t["traverse"] + t["shade"]
    wb = c["rays"] / 64.0   # wave-bounces if every lane were busy
    print(name, f"{ms:.1f} ms {c['rays']/ms/1e3:.0f} Mray/s", {k: f"{100*t[k]/tot:.1f}%" for k in ("setup", "traverse", "shade")}, flush=True)
    print("   per ray: node steps %.2f tri %.2f inst %.2f | per wave-bounce: rounds %.1f node iters %.1f tri iters %.1f ctrl %.1f | lane utilisation: node %.1f%% tri %.1f%% ctrl %.1f%%" % (
        c["node_tests"] / 2 / c["rays"], c["tri_tests"] / c["rays"], c["inst_visits"] / c["rays"],
        t["w_round"] / wb, t["w_node"] / wb, t["w_tri"] / wb, t["w_ctrl"] / wb,
        100 * (c["node_tests"] / 2) / (64 * t["w_node"]), 100 * c["tri_tests"] / (64 * max(t["w_tri"], 1)), 100 * c["inst_visits"] / (64 * max(t["w_ctrl"], 1))), flush=True)

#!/usr/bin/env python3
"""gpu_probe6.py — dev probe: wall-clock share of node / triangle / control / serve steps (counting kernel)."""
import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from __graft_entry__ import load_package, BUILT
pkg = load_package(); api = pkg.api; abi = pkg.abi
ctx = api.Context(0)
for name, w, h, spp, b in (("cfg2_hdr", 1280, 720, 64, 8), ("cfg3_venus", 1920, 1080, 16, 32), ("cfg4_statues", 3840, 2160, 4, 30), ("soup_1m", 2560, 1440, 8, 8)):
    scene = api.Scene(os.path.join(BUILT, name + ".blob"))
    ctx.upload(scene)
    fb = ctx.framebuffer(w, h)
    ctx.set_option(abi.OPT_COUNTER_LEVEL, 2)
    ctx.reset_counters()
    ctx.render_region(fb, w, h, spp, b); ctx.synchronize()
    c = ctx.counters(); t = ctx.phase_ticks()
    node, tri, ctrl, serve = t["traverse"], t["setup"], t["w_setup"], t["shade"]
    tot = node + tri + ctrl + serve
    wb = c["rays"] / 64.0
    print(name, "share: node %.0f%% tri %.0f%% ctrl %.0f%% serve %.0f%% | steps per wave-bounce: node %.1f tri %.1f ctrl %.1f serve %.2f | us per step: node %.2f tri %.2f ctrl %.2f serve %.2f" % (
        100*node/tot, 100*tri/tot, 100*ctrl/tot, 100*serve/tot, t["w_node"]/wb, t["w_tri"]/wb, t["w_ctrl"]/wb, t["w_shade"]/wb,
        node/100/max(t["w_node"],1), tri/100/max(t["w_tri"],1), ctrl/100/max(t["w_ctrl"],1), serve/100/max(t["w_shade"],1)), flush=True)

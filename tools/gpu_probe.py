#!/usr/bin/env python3
"""gpu_probe.py — dev probe: per-wave busy-time distribution and spp scaling on cfg2."""
import os, sys, json
import numpy as np
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from __graft_entry__ import load_package, BUILT
pkg = load_package(); api = pkg.api; abi = pkg.abi
ctx = api.Context(0)
scene = api.Scene(os.path.join(BUILT, "cfg2_hdr.blob"))
ctx.upload(scene)
w, h, b = 1280, 720, 8
fb = ctx.framebuffer(w, h)
ctx.set_option(abi.OPT_WAVE_STATS, 1)
ctx.set_option(abi.OPT_COUNTER_LEVEL, 1)
for wps, bpc in ((1, 2), (4, 3)):
    ctx.set_option(abi.OPT_WAVES_PER_SIMD, wps); ctx.set_option(abi.OPT_BLOCKS_PER_CU, bpc)
    for spp in (4, 64, 256):
        for items in (256, 1024, 4096):
            ctx.set_option(abi.OPT_UNIT_ITEMS, items)
            ctx.clear(fb, w, h); ctx.reset_counters()
            ctx.render_region(fb, w, h, spp, b); ctx.synchronize()
            ms = ctx.kernel_time_ms()[0]; rays = ctx.counters()["rays"]
            ws = ctx.wave_stats().astype(np.float64)
            t = ws[:, 0] / 100e3  # ms at 100 MHz
            print(f"wps{wps} bpc{bpc} spp{spp} items{items}: kernel {ms:.1f} ms {rays/ms/1e3:.0f} Mray/s | waves {len(t)} busy ms min {t.min():.1f} mean {t.mean():.1f} max {t.max():.1f} | units/wave mean {ws[:,1].mean():.1f}", flush=True)
ctx.set_option(abi.OPT_UNIT_ITEMS, 1024)
# region test: sky-only crop vs statue crop at 64 spp
ctx.set_option(abi.OPT_PASS_CHUNK, 64)
for name, reg in (("sky", (0, 560, 1280, 720)), ("statue", (840, 0, 1000, 520)), ("floor", (0, 0, 800, 160))):
    ctx.clear(fb, w, h); ctx.reset_counters()
    ctx.render_region(fb, w, h, 64, b, region=reg); ctx.synchronize()
    ms = ctx.kernel_time_ms()[0]; rays = ctx.counters()["rays"]
    ws = ctx.wave_stats().astype(np.float64); t = ws[:, 0] / 100e3
    print(f"region {name} {reg}: {ms:.1f} ms rays {rays} {rays/ms/1e3:.0f} Mray/s busy mean {t.mean():.1f} max {t.max():.1f}", flush=True)

#!/usr/bin/env python3
"""gpu_probe3.py — dev probe: cfg2 at 64 / 256 spp over kernel variants (timing only)."""
import os, sys
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
for wps, bpcs in ((4, (3, 4)),):
    ctx.set_option(abi.OPT_WAVES_PER_SIMD, wps)
    for bpc in bpcs:
        ctx.set_option(abi.OPT_BLOCKS_PER_CU, bpc)
        for spp in (64, 256):
            ctx.clear(fb, w, h); ctx.reset_counters()
            ctx.render_region(fb, w, h, spp, b); ctx.synchronize()
            ms = ctx.kernel_time_ms()[0]; rays = ctx.counters()["rays"]
            ws = ctx.wave_stats().astype(np.float64); t = ws[:, 0] / 100e3
            print(f"wps{wps} bpc{bpc} spp{spp}: kernel {ms:.1f} ms {rays/ms/1e3:.0f} Mray/s | waves {len(t)} busy mean {t.mean():.1f} max {t.max():.1f}", flush=True)

#!/usr/bin/env python3
"""probe_tail_hist.py — dev probe: how the waves of one dispatch of the bench frame finish: histogram of per-wave busy times (CRH_OPT_WAVE_STATS), the units
the latest waves worked on, for the full frame and for a 1/8 share."""
import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import numpy as np
from __graft_entry__ import load_package, BUILT
import bench
pkg = load_package(); api = pkg.api; abi = pkg.abi
W = bench.WORKLOAD
w, h, spp, b = W["width"], W["height"], W["samples"], W["bounces"]
ctx = api.Context(0); ctx.set_option(abi.OPT_COUNTER_LEVEL, 1); ctx.set_option(abi.OPT_WAVE_STATS, 1)
ctx.upload(api.Scene(os.path.join(BUILT, W["blob"] + ".blob")))
fb = ctx.framebuffer(w, h)
for label, tiles in (("world 1", pkg.render.owned_tiles(w, h, 64, 64, 1, 0, 1)), ("world 8 rank 0", pkg.render.owned_tiles(w, h, 64, 64, 1, 0, 8))):
    for rep in range(2):
        ctx.clear(fb, w, h); ctx.reset_counters(); ctx.render_tiles(fb, w, h, spp, b, tiles); ctx.synchronize()
    ms = ctx.kernel_time_ms()[0]; ws = ctx.wave_stats()
    busy = ws[:, 0] / 1e5
    print(f"{label}: kernel {ms:.2f} ms; wave busy mean {busy.mean():.2f} median {np.median(busy):.2f} min {busy.min():.2f} max {busy.max():.2f}; units/wave mean {ws[:,1].mean():.1f}")
    edges = np.quantile(busy, [0.5, 0.9, 0.99, 0.999, 1.0])
    print("   quantiles 50/90/99/99.9/100 %:", " ".join(f"{e:.2f}" for e in edges))
    late = np.argsort(busy)[-12:]
    print("   the 12 latest waves: busy ms", " ".join(f"{busy[i]:.2f}" for i in late), "; units", " ".join(str(int(ws[i, 1])) for i in late))
    hist, be = np.histogram(busy, bins=12)
    print("   histogram:", " ".join(f"{be[i]:.2f}:{hist[i]}" for i in range(len(hist))))

#!/usr/bin/env python3
"""probe_smoke_pixels.py — dev probe: the smoke() comparison with the offending pixels listed."""
import os, sys
import numpy as np
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "oracle"))
from __graft_entry__ import load_package, BUILT
import oracle_py
pkg = load_package(); api = pkg.api
blob = os.path.join(BUILT, "cfg1_scene.blob")
w, h, spp, bounces = 160, 100, 4, 4
ctx = api.Context(0); ctx.upload(api.Scene(blob)); fb = ctx.framebuffer(w, h)
ctx.render_region(fb, w, h, spp, bounces); img = ctx.download(fb, w, h); cnt = ctx.counters()
ref, ocnt = oracle_py.render(oracle_py.OracleScene(blob), w, h, spp, bounces)
d = np.abs(img - ref).max(axis=2)
print("lib", os.environ.get("CRH_LIB", "default"), "rays", cnt["rays"], ocnt["rays"], "rmse", float(np.sqrt(((img - ref) ** 2).mean())))
for y, x in zip(*np.where(d > 1e-3)):
    print("  pixel", x, y, "gpu", img[y, x], "oracle", ref[y, x])

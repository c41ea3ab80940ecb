#!/usr/bin/env python3
"""gpu_probe2.py — dev probe: find the tile/pixel where the GPU does far more traversal work than the oracle."""
import os, sys
import numpy as np
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "oracle"))
from __graft_entry__ import load_package, BUILT
import oracle_py as O
pkg = load_package(); api = pkg.api; abi = pkg.abi
ctx = api.Context(0)
blob = os.path.join(BUILT, "cfg2_hdr.blob")
scene = api.Scene(blob); osc = O.OracleScene(blob)
ctx.upload(scene)
w, h, b, spp = 1280, 720, 8, 16
fb = ctx.framebuffer(w, h)

def gpu(reg):
    ctx.clear(fb, w, h); ctx.reset_counters()
    ctx.render_region(fb, w, h, spp, b, region=reg); ctx.synchronize()
    return ctx.counters(), ctx.kernel_time_ms()[0]

def search(reg, depth=0):
    c, ms = gpu(reg)
    x0, y0, x1, y1 = reg
    print("  " * depth, reg, f"{ms:.1f} ms", c["node_tests"], c["tri_tests"], flush=True)
    if ms < 50 or (x1 - x0 <= 1 and y1 - y0 <= 1):
        return reg if ms >= 50 else None
    if x1 - x0 >= y1 - y0:
        xm = (x0 + x1) // 2
        halves = [(x0, y0, xm, y1), (xm, y0, x1, y1)]
    else:
        ym = (y0 + y1) // 2
        halves = [(x0, y0, x1, ym), (x0, ym, x1, y1)]
    for hreg in halves:
        r = search(hreg, depth + 1)
        if r:
            return r
    return None

pix = search((0, 560, 1280, 720))
print("slow pixel:", pix)
if pix:
    c, ms = gpu(pix)
    f2 = np.zeros((h, w, 3), np.float32)
    _, oc = O.render(osc, w, h, spp, b, region=pix, fb=f2)
    print("gpu", c, ms); print("cpu", oc)
    img = ctx.download(fb, w, h)
    x, y = pix[0], pix[1]
    print("gpu px", img[h - 1 - y, x], "cpu px", f2[h - 1 - y, x])
    # which pass?
    for p in range(spp):
        ctx.clear(fb, w, h); ctx.reset_counters()
        ctx.render_region(fb, w, h, spp, b, region=pix, first_pass=p, pass_count=1); ctx.synchronize()
        cc = ctx.counters(); ms = ctx.kernel_time_ms()[0]
        f3 = np.zeros((h, w, 3), np.float32)
        _, o3 = O.render(osc, w, h, spp, b, region=pix, first_pass=p, pass_count=1, fb=f3)
        if ms > 5 or cc["node_tests"] != o3["node_tests"]:
            print("pass", p, ms, cc, o3)

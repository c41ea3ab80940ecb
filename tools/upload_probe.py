#!/usr/bin/env python3
"""upload_probe.py — how long the boundary's host->device hand-over takes (PCIe-inclusive note):
blob load, context creation, crh_scene_upload (host-side layout compile + copies), framebuffer download."""
import os, sys, time
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from __graft_entry__ import load_package, BUILT
pkg = load_package(); api = pkg.api
t = time.perf_counter(); ctx = api.Context(0); t_ctx = time.perf_counter() - t
for name, w, h in (("cfg2_hdr", 1280, 720), ("soup_1m", 2560, 1440)):
    t = time.perf_counter(); scene = api.Scene(os.path.join(BUILT, name + ".blob")); t_load = time.perf_counter() - t
    d = scene.desc
    t = time.perf_counter(); ctx.upload(scene); ctx.synchronize(); t_up = time.perf_counter() - t
    t = time.perf_counter(); ctx.upload(scene); ctx.synchronize(); t_up2 = time.perf_counter() - t
    fb = ctx.framebuffer(w, h)
    t = time.perf_counter(); img = ctx.download(fb, w, h); t_down = time.perf_counter() - t
    print(f"{name}: context {t_ctx*1e3:.0f} ms, blob load {t_load*1e3:.0f} ms, upload {t_up*1e3:.0f} ms (again {t_up2*1e3:.0f} ms), "
          f"download {w}x{h} {t_down*1e3:.1f} ms; polys {d.poly_count} nodes {d.node_count} texture bytes {d.texture_bytes}", flush=True)

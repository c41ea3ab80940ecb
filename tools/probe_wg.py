#!/usr/bin/env python3
"""probe_wg.py — dev probe: the two forms of the path-tracing kernel side by side (CRH_OPT_KERNEL) on BASELINE.json's scenes,
with a sweep of the workgroup scheduler (linger, drainAt, maxDrainers, partialMin, walkMin, fillTo). First checks that both kernels
produce the same frame bit for bit on a small case. Writes gpurun_out/probe_wg.json.

    python tools/probe_wg.py [--quick] [linger,drainAt,maxDrainers,partialMin,walkMin,fillTo ...]
"""
import json, os, sys
import numpy as np
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from __graft_entry__ import load_package, BUILT
pkg = load_package(); api = pkg.api; abi = pkg.abi
args = [a for a in sys.argv[1:] if not a.startswith("--")]
quick = "--quick" in sys.argv
SWEEP = [tuple(int(v) for v in s.split(",")) for s in args] or [(8, 192, 1, 16, 32, 768)]
CASES = [("cfg2_hdr", 1280, 720, 256, 8), ("cfg3_venus", 1920, 1080, 16, 32), ("cfg4_statues", 3840, 2160, 4, 30), ("soup_1m", 2560, 1440, 16, 8)]
if quick:
    CASES = [("cfg2_hdr", 1280, 720, 64, 8), ("soup_1m", 2560, 1440, 8, 8)]
ctx = api.Context(0)
out = {"cases": []}

# bit identity first (small frame, full counters)
scene = api.Scene(os.path.join(BUILT, "cfg1_scene.blob"))
ctx.upload(scene)
fb = ctx.framebuffer(320, 200)
frames = {}
for kern in (abi.KERNEL_WAVE, abi.KERNEL_WG):
    ctx.set_option(abi.OPT_KERNEL, kern)
    ctx.set_option(abi.OPT_COUNTER_LEVEL, 2)
    ctx.clear(fb, 320, 200); ctx.reset_counters()
    ctx.render_region(fb, 320, 200, 4, 4); ctx.synchronize()
    frames[kern] = (ctx.download(fb, 320, 200), ctx.counters())
same = bool(np.array_equal(frames[0][0], frames[1][0])) and frames[0][1] == frames[1][1]
out["bit_identical_cfg1"] = same
print("bit identical (cfg1, counters too):", same, flush=True)
if not same:
    d = np.abs(frames[0][0] - frames[1][0])
    print("  max diff", float(d.max()), "pixels", int((d.max(axis=2) > 0).sum()), frames[0][1], frames[1][1], flush=True)

ctx.set_option(abi.OPT_COUNTER_LEVEL, 1)
for name, w, h, spp, b in CASES:
    path = os.path.join(BUILT, name + ".blob")
    if not os.path.exists(path):
        continue
    ctx.upload(api.Scene(path))
    fb = ctx.framebuffer(w, h)
    rec = {"case": name, "w": w, "h": h, "spp": spp, "bounces": b, "runs": []}
    ref_img = None
    for kern, cfgs in ((abi.KERNEL_WAVE, [None]), (abi.KERNEL_WG, SWEEP)):
        ctx.set_option(abi.OPT_KERNEL, kern)
        for cfg in cfgs:
            if cfg:
                ctx.set_sched_wg(*cfg)
            best = None
            for rep in range(2):
                ctx.clear(fb, w, h); ctx.reset_counters()
                ctx.render_region(fb, w, h, spp, b); ctx.synchronize()
                ms = ctx.kernel_time_ms()[0]; rays = ctx.counters()["rays"]
                best = ms if best is None else min(best, ms)
            img = ctx.download(fb, w, h)
            if ref_img is None:
                ref_img = img
            same = bool(np.array_equal(img, ref_img))
            r = {"kernel": "wg" if kern else "wave", "sched": cfg, "ms": round(best, 2), "mrays": round(rays / best / 1e3, 1), "same_frame": same}
            rec["runs"].append(r)
            print(name, r, flush=True)
    out["cases"].append(rec)
os.makedirs(os.path.join(REPO, "gpurun_out"), exist_ok=True)
json.dump(out, open(os.path.join(REPO, "gpurun_out", "probe_wg.json"), "w"), indent=1)

#!/usr/bin/env python3
"""ab_walk.py — dev / evidence (round 5): CRH_OPT_WALK = CRH_WALK_BINARY (the contract) against CRH_WALK_WIDE4 (the 4-ary copy of the BVHs) on the BASELINE scenes, on the GPU,
in one process: kernel time (best of N), Mray/s, rays, and how the two frames compare (floats / pixels that differ, largest difference).

    python tools/ab_walk.py [--soup10m] [--reps N] [name ...]          -> gpurun_out/ab_walk.json
"""
import hashlib, json, os, sys, time
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import numpy as np
from __graft_entry__ import load_package, BUILT
pkg = load_package(); api = pkg.api; abi = pkg.abi
CASES = [("cfg2_hdr", 1280, 720, 256, 8), ("cfg4_statues", 3840, 2160, 16, 30), ("soup_1m", 2560, 1440, 16, 8), ("cfg3_venus", 1920, 1080, 32, 32)]
args = [a for a in sys.argv[1:] if not a.startswith("--")]
reps = int(sys.argv[sys.argv.index("--reps") + 1]) if "--reps" in sys.argv else 3
if "--reps" in sys.argv: args = [a for a in args if a != sys.argv[sys.argv.index("--reps") + 1]]
if "--soup10m" in sys.argv:
    path = os.path.join(BUILT, "soup_10m.blob")
    if not os.path.exists(path):
        sys.path.insert(0, os.path.join(REPO, "tools"))
        import make_soup_blob
        t0 = time.time(); make_soup_blob.build(10_000_000, path); print(f"soup_10m.blob built in {time.time() - t0:.1f} s", flush=True)
    CASES.append(("soup_10m", 2560, 1440, 16, 8))
if args: CASES = [c for c in CASES if c[0] in args]
out = {}
for name, w, h, spp, b in CASES:
    path = os.path.join(BUILT, name + ".blob")
    if not os.path.exists(path): continue
    res = {}
    imgs = {}
    for walk, tag in ((abi.WALK_BINARY, "binary"), (abi.WALK_WIDE4, "wide4")):
        ctx = api.Context(0)
        ctx.set_option(abi.OPT_COUNTER_LEVEL, 1)
        ctx.set_option(abi.OPT_WALK, walk)
        t0 = time.time(); ctx.upload(api.Scene(path)); up = time.time() - t0
        fb = ctx.framebuffer(w, h)
        best = None
        for _ in range(reps):
            ctx.clear(fb, w, h); ctx.reset_counters()
            ctx.render_region(fb, w, h, spp, b); ctx.synchronize()
            ms = ctx.kernel_time_ms()[0]
            best = ms if best is None else min(best, ms)
        rays = ctx.counters()["rays"]
        imgs[tag] = ctx.download(fb, w, h)
        res[tag] = {"ms": round(best, 3), "mrays": round(rays / best / 1e3, 1), "rays": rays, "upload_s": round(up, 3), "kernel": ctx.last_kernel_name() if hasattr(ctx, "last_kernel_name") else None,
                    "md5": hashlib.md5(imgs[tag].tobytes()).hexdigest()}
        ctx.close()
    a, bimg = imgs["binary"], imgs["wide4"]
    ne = a.view(np.uint32) != bimg.view(np.uint32)
    res["speedup"] = round(res["binary"]["ms"] / res["wide4"]["ms"], 4)
    res["frames"] = {"floats_that_differ": int(ne.sum()), "pixels_that_differ": int(ne.any(axis=2).sum()), "of_pixels": w * h, "largest_abs_difference": float(np.abs(a.astype(np.float64) - bimg.astype(np.float64)).max()),
                     "rays_equal": res["binary"]["rays"] == res["wide4"]["rays"]}
    out[name] = res
    print(f"{name:14s} {spp:4d} spp  binary {res['binary']['mrays']:8.1f}  wide4 {res['wide4']['mrays']:8.1f} Mray/s  x{res['speedup']:.3f}  pixels that differ {res['frames']['pixels_that_differ']} of {w * h} (max {res['frames']['largest_abs_difference']:.3g})  kernel {res['wide4']['kernel']}", flush=True)
os.makedirs(os.path.join(REPO, "gpurun_out"), exist_ok=True)
json.dump(out, open(os.path.join(REPO, "gpurun_out", "ab_walk.json"), "w"), indent=1)

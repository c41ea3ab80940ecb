#!/usr/bin/env python3
"""gpu_check.py — quick on-GPU sanity run (parity vs the oracle + first timings). Dev tool, not a test."""
import json
import os
import sys
import time

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "oracle"))
from __graft_entry__ import load_package, BUILT  # noqa: E402

pkg = load_package()
api = pkg.api
import oracle_py  # noqa: E402


def stats(img, ref):
    d = np.abs(img - ref)
    return {"rmse": float(np.sqrt(((img - ref) ** 2).mean())), "max": float(d.max()),
            "pix_gt_1e-3": float((d.max(axis=2) > 1e-3).mean()), "pix_ne": float((d.max(axis=2) > 0).mean())}


def main():
    out = {}
    print("devices", api.device_count(), flush=True)
    ctx = api.Context(0)
    # ---- cfg1: exact traversal parity + image parity
    blob = os.path.join(BUILT, "cfg1_scene.blob")
    scene = api.Scene(blob)
    oscene = oracle_py.OracleScene(blob)
    ctx.upload(scene)
    rng = np.random.default_rng(7)
    n = 200000
    cam = scene.desc.camera
    rays = np.zeros((n, 6), np.float32)
    rays[:, 0:3] = [cam.A[3], cam.A[7], cam.A[11]]
    rays[:, 3:6] = rng.normal(size=(n, 3)).astype(np.float32)
    t = time.time(); hg = ctx.trace_rays(rays); tg = time.time() - t
    ho = oracle_py.trace_rays(oscene, rays)
    eq = {f: bool(np.array_equal(hg[f], ho[f])) for f in ["inst", "poly", "distance", "node_tests", "tri_tests", "material", "point", "normal", "uv"]}
    mesh = ho["poly"] >= 0
    eq["point_mesh"] = bool(np.array_equal(hg["point"][mesh], ho["point"][mesh]))
    eq["normal_mesh"] = bool(np.array_equal(hg["normal"][mesh], ho["normal"][mesh]))
    eq["uv_maxdiff"] = float(np.abs(hg["uv"] - ho["uv"]).max())
    eq["normal_maxdiff"] = float(np.abs(hg["normal"] - ho["normal"]).max())
    out["trace_rays_cfg1"] = eq
    print("trace_rays", eq, "gpu s", tg, flush=True)
    w, h, spp, b = 320, 200, 4, 4
    fb = ctx.framebuffer(w, h)
    ctx.reset_counters()
    ctx.render_region(fb, w, h, spp, b)
    img = ctx.download(fb, w, h)
    cnt = ctx.counters()
    ref, ocnt = oracle_py.render(oscene, w, h, spp, b)
    out["cfg1"] = {"stats": stats(img, ref), "gpu_counters": cnt, "oracle_counters": ocnt, "kernel_ms": ctx.kernel_time_ms()[0]}
    print("cfg1", out["cfg1"], flush=True)
    # ---- cfg2 hdr
    blob2 = os.path.join(BUILT, "cfg2_hdr.blob")
    if os.path.exists(blob2):
        scene2 = api.Scene(blob2)
        oscene2 = oracle_py.OracleScene(blob2)
        ctx.upload(scene2)
        w, h, b = 1280, 720, 8
        fb2 = ctx.framebuffer(w, h)
        for spp in (4,):
            ctx.clear(fb2, w, h)
            ctx.reset_counters()
            ctx.render_region(fb2, w, h, spp, b)
            img = ctx.download(fb2, w, h)
            cnt = ctx.counters()
            t = time.time(); ref, ocnt = oracle_py.render(oscene2, w, h, spp, b); tc = time.time() - t
            ms = ctx.kernel_time_ms()[0]
            out[f"cfg2_{spp}spp"] = {"stats": stats(img, ref), "gpu_counters": cnt, "oracle_counters": ocnt, "kernel_ms": ms,
                                     "gpu_mrays": cnt["rays"] / ms / 1e3, "cpu_s": tc, "cpu_mrays": ocnt["rays"] / tc / 1e6}
            print(f"cfg2 {spp}spp", out[f"cfg2_{spp}spp"], flush=True)
        for wps in (1, 4):
            ctx.set_option(pkg.abi.OPT_WAVES_PER_SIMD, wps)
            for level in (2, 1):
                ctx.set_option(pkg.abi.OPT_COUNTER_LEVEL, level)
                for bpc in ((1, 2) if wps == 1 else (2, 4)):
                    for chunk in (16, 64):
                        ctx.set_option(pkg.abi.OPT_BLOCKS_PER_CU, bpc)
                        ctx.set_option(pkg.abi.OPT_PASS_CHUNK, chunk)
                        ctx.clear(fb2, w, h)
                        ctx.reset_counters()
                        spp = 64
                        ctx.render_region(fb2, w, h, spp, b)
                        ctx.synchronize()
                        cnt = ctx.counters()
                        ms = ctx.kernel_time_ms()[0]
                        key = f"cfg2_{spp}spp_w{wps}_l{level}_b{bpc}_c{chunk}"
                        out[key] = {"kernel_ms": ms, "rays": cnt["rays"], "mrays": cnt["rays"] / ms / 1e3}
                        print(key, out[key], flush=True)
    os.makedirs(os.path.join(REPO, "gpurun_out"), exist_ok=True)
    with open(os.path.join(REPO, "gpurun_out", "gpu_check.json"), "w") as f:
        json.dump(out, f, indent=1)
    ctx.close()


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""run_config.py — render one scene blob on the GPU and report Mray/s, counters, algorithmic bytes and (optionally)
parity against the oracle at a reduced sample count. Used for BASELINE.json configs 3-5 (profiles/*.json).

    python tools/run_config.py --blob scenes/_built/cfg3_venus.blob --width 1920 --height 1080 --spp 64 --bounces 32 [--parity-spp 4]
"""
import argparse
import json
import os
import sys
import time

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "oracle"))
from __graft_entry__ import load_package  # noqa: E402
from bench import algorithmic_bytes, HBM_PEAK_GBS  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--blob", required=True)
    ap.add_argument("--width", type=int, required=True)
    ap.add_argument("--height", type=int, required=True)
    ap.add_argument("--spp", type=int, required=True)
    ap.add_argument("--bounces", type=int, required=True)
    ap.add_argument("--parity-spp", type=int, default=0)
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--tag", default="")
    a = ap.parse_args()
    pkg = load_package()
    api, abi = pkg.api, pkg.abi
    t0 = time.time()
    scene = api.Scene(a.blob)
    ctx = api.Context(0)
    ctx.upload(scene)
    upload_s = time.time() - t0
    d = scene.desc
    w, h = a.width, a.height
    fb = ctx.framebuffer(w, h)
    out = {"tag": a.tag or os.path.basename(a.blob), "width": w, "height": h, "spp": a.spp, "bounces": a.bounces,
           "nodes": int(d.node_count), "polys": int(d.poly_count), "instances": int(d.instance_count), "upload_s": round(upload_s, 2)}
    ctx.set_option(abi.OPT_COUNTER_LEVEL, 2)
    ctx.reset_counters()
    ctx.render_region(fb, w, h, a.spp, a.bounces)
    ctx.synchronize()
    full = ctx.counters()
    out["counting_run_ms"] = round(ctx.kernel_time_ms()[0], 2)
    ctx.set_option(abi.OPT_COUNTER_LEVEL, 1)
    ctx.reset_counters()
    for _ in range(a.steps):
        ctx.clear(fb, w, h)
        ctx.render_region(fb, w, h, a.spp, a.bounces)
    ctx.synchronize()
    _, total_ms, n = ctx.kernel_time_ms()
    ms = total_ms / n
    alg = algorithmic_bytes(full)
    out.update({"kernel_ms": round(ms, 2), "rays": full["rays"], "mrays": round(full["rays"] / ms / 1e3, 1),
                "rays_per_path": round(full["rays"] / full["paths"], 3), "node_tests_per_ray": round(full["node_tests"] / full["rays"], 2),
                "tri_tests_per_ray": round(full["tri_tests"] / full["rays"], 2), "bytes_per_ray": round(alg / full["rays"], 1),
                "achieved_GBs": round(alg / ms / 1e6, 1), "frac_of_hbm_peak": round(alg / ms / 1e6 / HBM_PEAK_GBS, 4), "counters": full})
    img = ctx.download(fb, w, h)
    out["finite"] = bool(np.isfinite(img).all())
    out["mean"] = float(img.mean())
    if a.parity_spp:
        import oracle_py
        osc = oracle_py.OracleScene(a.blob)
        ctx.clear(fb, w, h)
        ctx.set_option(abi.OPT_COUNTER_LEVEL, 2)
        ctx.reset_counters()
        ctx.render_region(fb, w, h, a.parity_spp, a.bounces)
        g = ctx.download(fb, w, h)
        gc = ctx.counters()
        t = time.time()
        ref, oc = oracle_py.render(osc, w, h, a.parity_spp, a.bounces)
        cpu_s = time.time() - t
        dd = np.abs(g.astype(np.float64) - ref)
        per_px = np.sqrt((dd ** 2).sum(axis=2))
        # BASELINE.md section 3 report row: pixels > 1 LSB (8-bit sRGB), mean and p99.9 per-pixel L2, RMSE — and the bar itself: identical floats
        g8, r8 = oracle_py.to_srgb8(np.ascontiguousarray(g)).astype(np.int32), oracle_py.to_srgb8(np.ascontiguousarray(ref)).astype(np.int32)
        out["parity"] = {"spp": a.parity_spp, "floats_that_differ": int((g.view(np.uint32) != ref.view(np.uint32)).sum()), "bit_identical": bool(np.array_equal(g, ref)),
                         "pixels_gt_1lsb_pct": float(100.0 * (np.abs(g8 - r8).max(axis=2) > 1).mean()), "p999_l2": float(np.quantile(per_px, 0.999)),
                         "rmse": float(np.sqrt((dd ** 2).mean())), "frac_px_gt_1e-3": float((per_px > 1e-3).mean()),
                         "mean_l2": float(per_px.mean()), "gpu_rays": gc["rays"], "oracle_rays": oc["rays"],
                         "oracle_s": round(cpu_s, 2), "oracle_mrays": round(oc["rays"] / cpu_s / 1e6, 2), "cores": os.cpu_count()}
    print(json.dumps(out), flush=True)
    ctx.close()


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""ab_stream.py — the streaming form (CRH_KERNEL_STREAM, csrc/pathtrace_stream.h) against the rolling megakernel on the bench workloads, one GPU call:
same frame (bit for bit) and same ray count required; wall time of render + synchronize (the streaming form's host loop is part of a dispatch) and the events' time.

    python tools/ab_stream.py [workload:spp ...]          default: cfg2:64 cfg4:8 soup:16 cfg3:16 soup10m:8
    AB_COHORTS=16384,32768  AB_GROUP=8  AB_REPS=2         pool sizes / iterations per group to try
"""
import hashlib
import json
import os
import sys
import time

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from __graft_entry__ import load_package, BUILT  # noqa: E402
import bench  # noqa: E402

pkg = load_package(); api = pkg.api; abi = pkg.abi
jobs = [a for a in sys.argv[1:] if ":" in a] or ["cfg2:64", "cfg4:8", "soup:16", "cfg3:16", "soup10m:8"]
cohorts_list = [int(x) for x in os.environ.get("AB_COHORTS", "16384").split(",")]
reps = int(os.environ.get("AB_REPS", "2"))
results = {}


def timed(ctx, fb, w, h, spp, b):
    best = None
    for _ in range(reps):
        ctx.clear(fb, w, h); ctx.reset_counters(); ctx.synchronize()
        t0 = time.perf_counter()
        ctx.render_region(fb, w, h, spp, b)
        ctx.synchronize()
        wall = (time.perf_counter() - t0) * 1e3
        ev = ctx.kernel_time_ms()[0]
        if best is None or wall < best[0]:
            best = (wall, ev)
    img = ctx.download(fb, w, h)
    return best, hashlib.md5(img.tobytes()).hexdigest(), ctx.counters()


for job in jobs:
    key, spp = job.split(":"); spp = int(spp)
    wl = bench.WORKLOADS[key]
    blob = bench.workload_blob(key, BUILT)
    if not os.path.exists(blob):
        print(f"{job}: {blob} not built", flush=True); continue
    scene = api.Scene(blob)
    w, h, b = wl["width"], wl["height"], wl["bounces"]
    row = {}
    ctx = api.Context(0)
    ctx.set_option(abi.OPT_COUNTER_LEVEL, 1)
    ctx.set_option(abi.OPT_KERNEL, abi.KERNEL_ROLL)
    ctx.upload(scene)
    fb = ctx.framebuffer(w, h)
    (wall, ev), md5, cnt = timed(ctx, fb, w, h, spp, b)
    rays = cnt["rays"]
    row["roll"] = {"wall_ms": round(wall, 2), "event_ms": round(ev, 2), "mrays": round(rays / wall / 1e3, 1), "kernel": ctx.last_kernel_name()}
    print(f"{key} {w}x{h} {spp} spp: {rays} rays; rolling kernel {wall:8.2f} ms wall ({ev:8.2f} events) = {rays / wall / 1e3:8.1f} Mray/s", flush=True)
    ctx.close()
    for co in cohorts_list:
        ctx = api.Context(0)
        ctx.set_option(abi.OPT_COUNTER_LEVEL, 1)
        ctx.set_option(abi.OPT_KERNEL, abi.KERNEL_STREAM)
        ctx.set_option(abi.OPT_STREAM_COHORTS, co)
        ctx.upload(scene)
        fb = ctx.framebuffer(w, h)
        try:
            (wall2, ev2), md52, cnt2 = timed(ctx, fb, w, h, spp, b)
        except api.CrhError as e:
            print(f"   stream {co}: {e}", flush=True); ctx.close(); continue
        same = md52 == md5 and cnt2["rays"] == rays and cnt2["paths"] == cnt["paths"]
        row[f"stream_{co}"] = {"wall_ms": round(wall2, 2), "event_ms": round(ev2, 2), "mrays": round(rays / wall2 / 1e3, 1), "vs_roll": round(wall / wall2, 3), "same_frame": same, "kernel": ctx.last_kernel_name()}
        print(f"   streaming, {co:6d} cohorts: {wall2:8.2f} ms wall ({ev2:8.2f} events) = {rays / wall2 / 1e3:8.1f} Mray/s = {wall / wall2:5.3f} x; frame and counters {'identical' if same else 'DIFFER'}  [{ctx.last_kernel_name()}]", flush=True)
        ctx.close()
    results[job] = row
    scene.close()
os.makedirs(os.path.join(REPO, "gpurun_out"), exist_ok=True)
json.dump(results, open(os.path.join(REPO, "gpurun_out", "ab_stream.json"), "w"), indent=1)

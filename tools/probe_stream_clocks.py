#!/usr/bin/env python3
"""probe_stream_clocks.py — dev: where k_stream_shade's workgroups spend their time (a variant library built with -DCRH_STREAM_CLOCKS: tools/build_variant.sh sclk -DCRH_STREAM_CLOCKS;
CRH_LIB=c-ray_amd/_lib/variants/sclk.so python tools/probe_stream_clocks.py cfg4:8). Phases, thread 0's wall clock between the workgroup's barriers, summed over workgroups and iterations:
fetch = the cohort counter's atomic, classify = fill level + hit records + ordered lists, hits / misses = shadeCore on their batches, alloc = the refill's fetch-and-add + the
counters' flush, refill = new camera rays (+ the wait at the top of the next cohort)."""
import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from __graft_entry__ import load_package, BUILT
import bench
pkg = load_package(); api = pkg.api; abi = pkg.abi
for job in sys.argv[1:] or ["cfg4:8"]:
    key, spp = job.split(":"); spp = int(spp)
    wl = bench.WORKLOADS[key]
    scene = api.Scene(bench.workload_blob(key, BUILT))
    ctx = api.Context(0)
    ctx.set_option(abi.OPT_COUNTER_LEVEL, 1); ctx.set_option(abi.OPT_KERNEL, abi.KERNEL_STREAM)
    ctx.upload(scene)
    w, h, b = wl["width"], wl["height"], wl["bounces"]
    fb = ctx.framebuffer(w, h)
    ctx.render_region(fb, w, h, spp, b); ctx.synchronize(); ctx.reset_counters(); ctx.clear(fb, w, h)
    ctx.render_region(fb, w, h, spp, b); ctx.synchronize()
    t = list(ctx.phase_ticks().values())
    names = ["fetch", "classify", "hits", "misses", "alloc", "refill"]
    tot = float(sum(t[:6])) or 1.0
    print(key, spp, "spp:", ctx.kernel_time_ms()[0], "ms;", ", ".join(f"{n} {100 * t[i] / tot:.1f} %" for i, n in enumerate(names)), f"; {tot / 100e6 / 1024 * 1e3:.2f} ms per workgroup (1024 workgroups)")

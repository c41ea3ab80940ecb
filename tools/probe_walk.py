#!/usr/bin/env python3
"""probe_walk.py — round 6, VERDICT r05 item 1 step A: what does the WALK do at more than four waves per SIMD?

For every scene: (1) a counting dispatch of the path tracer (counter level 2) records every ray its waves start, wave by wave, in the order they start them
(crh_debug_ray_dump) and the per-step-kind clocks that say which share of the path tracer's time is the walk; (2) the timed kernel (counter level 1) renders the
same dispatch: the path tracer's ray rate, and — divided by the walk's share of its step time — the rate of its walk; (3) k_walk_probe (csrc/walk_probe.h: the same lane
code and node run, no generation / shading / fold) walks the recorded rays at 4 ... 8 waves per SIMD, every variant checked bit for bit against the one-ray-per-lane walk.

    python tools/probe_walk.py [scene ...]            PROBE_RAYS=64e6 (rays to record per scene)  PROBE_UNIT=512 (rays per queue unit)  PROBE_VARIANTS=4.12.1.1,...
"""
import json, os, sys, time
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from __graft_entry__ import load_package, BUILT
import bench
pkg = load_package(); api = pkg.api; abi = pkg.abi

SCENES = {          # name -> (bench workload key, width, height, bounces, rays per path (to size the dispatch))
    "cfg4_statues": ("cfg4", 3840, 2160, 30, 3.45), "soup_10m": ("soup10m", 2560, 1440, 8, 3.52), "cfg2_hdr": ("cfg2", 1280, 720, 8, 2.47),
    "soup_1m": ("soup", 2560, 1440, 8, 3.36), "cfg3_venus": ("cfg3", 1920, 1080, 32, 6.12)}
VARIANTS = [(4, 12, 1, 1), (4, 12, 1, 0), (2, 12, 1, 3), (3, 12, 1, 3), (4, 12, 1, 3), (4, 7, 1, 3), (4, 3, 1, 3), (4, 4, 0, 3), (5, 12, 1, 3), (6, 7, 1, 3), (7, 3, 1, 3), (8, 4, 0, 3)]
if os.environ.get("PROBE_VARIANTS"):
    VARIANTS = [tuple(int(x) for x in v.split(".")) for v in os.environ["PROBE_VARIANTS"].split(",")]
want_rays = float(os.environ.get("PROBE_RAYS", "64e6"))
unit = int(os.environ.get("PROBE_UNIT", "512"))
names = [a for a in sys.argv[1:] if not a.startswith("-")] or ["cfg4_statues", "soup_10m", "cfg2_hdr"]
out = {}
ctx = api.Context(0)
for name in names:
    key, w, h, b, rpp = SCENES[name]
    blob = bench.workload_blob(key, BUILT)
    if not os.path.exists(blob):
        print(name, "blob missing", flush=True); continue
    spp = max(1, int(round(want_rays / (w * h * rpp))))
    ctx.upload(api.Scene(blob))
    fb = ctx.framebuffer(w, h)
    # the timed kernel on this dispatch
    ctx.set_option(abi.OPT_COUNTER_LEVEL, 1)
    best = None
    for _ in range(2):
        ctx.clear(fb, w, h); ctx.reset_counters(); ctx.render_region(fb, w, h, spp, b); ctx.synchronize()
        ms = ctx.kernel_time_ms()[0]; best = ms if best is None else min(best, ms)
    rays_timed = ctx.counters()["rays"]
    # the counting kernel: step clocks + the ray dump
    ctx.set_option(abi.OPT_COUNTER_LEVEL, 2)
    waves = 4096
    cap = int(2.5 * rays_timed / waves) + 1024
    ctx.ray_dump(cap)
    ctx.clear(fb, w, h); ctx.reset_counters(); ctx.render_region(fb, w, h, spp, b); ctx.synchronize()
    ms_count = ctx.kernel_time_ms()[0]
    c = ctx.counters(); t = ctx.phase_ticks()
    total, per = ctx.ray_dump_counts()
    walk_ticks = t["traverse"] + t["setup"] + t["w_setup"]          # node runs (incl. the triangle / instance / refill steps inside them), triangle steps, control steps
    all_ticks = walk_ticks + t["shade"] + t["t_swap"] + t["t_gen"]
    share = walk_ticks / max(all_ticks, 1)
    share_swap = (walk_ticks + t["t_swap"]) / max(all_ticks, 1)          # ... with the round-level retire / refill steps counted as walk (the probe does them too)
    mk = rays_timed / best / 1e3
    res = {"spp": spp, "rays": rays_timed, "rays_recorded": total, "waves_that_overran": int((per >= cap).sum()), "megakernel_ms": round(best, 2), "megakernel_mrays": round(mk, 1),
           "counting_kernel_ms": round(ms_count, 2), "walk_share_of_step_time": round(share, 4), "walk_plus_refill_share": round(share_swap, 4),
           "megakernel_walk_mrays": round(mk / share_swap, 1), "node_tests_per_ray": round(c["node_tests"] / c["rays"], 1), "tri_tests_per_ray": round(c["tri_tests"] / c["rays"], 2), "variants": {}}
    print(f"{name}: {spp} spp, {rays_timed} rays ({total} recorded, {res['waves_that_overran']} waves overran); path tracer {best:.1f} ms = {mk:.0f} Mray/s; walk {100 * share:.1f} % (+ refill {100 * share_swap:.1f} %) "
          f"of step time -> its walk runs at {mk / share_swap:.0f} Mray/s", flush=True)
    ms0, _ = ctx.walk_probe(0, slot=0)
    print(f"  reference form (one ray per lane, k_trace_rays' loop): {ms0:.1f} ms = {total / ms0 / 1e3:.0f} Mray/s", flush=True)
    res["reference_form_mrays"] = round(total / ms0 / 1e3, 1)
    for wps, nlds, inst, fused in VARIANTS:
        try:
            msv = None
            for _ in range(2):
                m, _r = ctx.walk_probe(wps, nlds, bool(inst), int(fused), unit_rays=unit, slot=1)
                msv = m if msv is None else min(msv, m)
            differ = ctx.walk_probe_compare()
        except api.CrhError as e:
            print(f"  {wps} waves/SIMD stack {nlds} inst {inst} form {fused}: {e}", flush=True); continue
        kn = ctx.last_kernel_name()
        rate = total / msv / 1e3
        res["variants"][f"{wps}.{nlds}.{inst}.{fused}"] = {"ms": round(msv, 2), "mrays": round(rate, 1), "vs_megakernel_walk": round(rate / (mk / share_swap), 3), "hits_that_differ": differ, "kernel": kn}
        print(f"  {kn:52s} {msv:8.1f} ms = {rate:7.0f} Mray/s = {rate / (mk / share_swap):.3f} x the path tracer's walk; hits that differ from the reference form: {differ}", flush=True)
    ctx.ray_dump(0)
    ctx.set_option(abi.OPT_WAVE_STATS, 0)
    out[name] = res
os.makedirs(os.path.join(REPO, "gpurun_out"), exist_ok=True)
json.dump(out, open(os.path.join(REPO, "gpurun_out", "probe_walk.json"), "w"), indent=1)

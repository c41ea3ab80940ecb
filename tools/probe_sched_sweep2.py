#!/usr/bin/env python3
"""probe_sched_sweep2.py — dev probe (round 5): the scheduler's run parameters on the traversal-heavy configs at sample counts where a dispatch is long (statues 32 spp, the soups 64 / 32 spp),
one parameter at a time around the defaults. name=value[,value...] ... ; default: run_num tri_in_run ctrl_in_run swap_in_run swap_min fill_to."""
import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from __graft_entry__ import load_package, BUILT
pkg = load_package(); api = pkg.api; abi = pkg.abi
DEFAULT = dict(node=70, tri=160, ctrl=120, swap_min=16, fill_to=160, run_num=4, tri_in_run=12, ctrl_in_run=12, shade_min=48, swap_in_run=20)
SWEEP = {"run_num": [3, 5, 6, 8], "tri_in_run": [8, 16, 24], "ctrl_in_run": [8, 16, 24], "swap_in_run": [16, 28], "swap_min": [12, 24], "fill_to": [128, 192], "tri": [120, 240], "node": [50, 100]}
if len(sys.argv) > 1:
    SWEEP = {a.split("=")[0]: [int(v) for v in a.split("=")[1].split(",")] for a in sys.argv[1:]}
cases = [("cfg4_statues", 3840, 2160, 32, 30), ("soup_1m", 2560, 1440, 64, 8)]
p10 = os.path.join(BUILT, "soup_10m.blob")
if not os.path.exists(p10):
    sys.path.insert(0, os.path.join(REPO, "tools"))
    import make_soup_blob
    make_soup_blob.build(10_000_000, p10)
cases.append(("soup_10m", 2560, 1440, 32, 8))
for name, w, h, spp, b in cases:
    ctx = api.Context(0)
    ctx.set_option(abi.OPT_COUNTER_LEVEL, 1)
    ctx.upload(api.Scene(os.path.join(BUILT, name + ".blob")))
    fb = ctx.framebuffer(w, h)
    def run(**kw):
        ctx.set_sched(**dict(DEFAULT, **kw))
        best = None
        for rep in range(2):
            ctx.clear(fb, w, h); ctx.reset_counters()
            ctx.render_region(fb, w, h, spp, b); ctx.synchronize()
            ms = ctx.kernel_time_ms()[0]
            best = ms if best is None else min(best, ms)
        return best, ctx.counters()["rays"]
    base, rays = run()
    print(f"{name} {spp} spp defaults: {base:.1f} ms {rays / base / 1e3:.0f} Mray/s", flush=True)
    for k, vals in SWEEP.items():
        for v in vals:
            ms, _ = run(**{k: v})
            print(f"  {name} {k}={v}: {base / ms:.3f}", flush=True)
    base2, _ = run()
    print(f"  {name} defaults again: {base / base2:.3f}", flush=True)
    ctx.close()

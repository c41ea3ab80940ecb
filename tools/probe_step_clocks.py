#!/usr/bin/env python3
"""probe_step_clocks.py — dev probe: per-step-kind clocks of the wave scheduler (counter level 2)."""
import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from __graft_entry__ import load_package, BUILT
pkg = load_package(); api = pkg.api; abi = pkg.abi
ctx = api.Context(0)
ctx.set_option(abi.OPT_COUNTER_LEVEL, 2)
if os.environ.get("PROBE_KERNEL"):          # 0: the one-unit-at-a-time form, 2: the rolling-units form (the default)
    ctx.set_option(abi.OPT_KERNEL, int(os.environ["PROBE_KERNEL"]))
only = sys.argv[1:]          # optional scene names
# PROBE_SPP=256 PROBE_UNIT_ITEMS=2048,512: the passes per pixel and the work-unit sizes (one run each) instead of the defaults
units = [int(v) for v in os.environ["PROBE_UNIT_ITEMS"].split(",")] if os.environ.get("PROBE_UNIT_ITEMS") else [None]
if os.environ.get("PROBE_WALK"):            # 1: CRH_OPT_WALK = CRH_WALK_WIDE4 (round 5)
    ctx.set_option(abi.OPT_WALK, int(os.environ["PROBE_WALK"]))
for name, w, h, spp, b in (("cfg2_hdr", 1280, 720, 64, 8), ("cfg3_venus", 1920, 1080, 16, 32), ("cfg4_statues", 3840, 2160, 4, 30), ("soup_1m", 2560, 1440, 16, 8), ("soup_10m", 2560, 1440, 8, 8)):
  if only and name not in only:
      continue
  if not os.path.exists(os.path.join(BUILT, name + ".blob")):          # (soup_10m.blob is built on the GPU box: tools/make_soup_blob.py)
      continue
  if os.environ.get("PROBE_SPP"): spp = int(os.environ["PROBE_SPP"])
  scene = api.Scene(os.path.join(BUILT, name + ".blob"))
  ctx.upload(scene)
  fb = ctx.framebuffer(w, h)
  for items in units:
    if items: ctx.set_option(abi.OPT_UNIT_ITEMS, items)
    for rep in range(2):
        ctx.clear(fb, w, h); ctx.reset_counters()
        ctx.render_region(fb, w, h, spp, b); ctx.synchronize()
    ms = ctx.kernel_time_ms()[0]; c = ctx.counters(); t = ctx.phase_ticks()
    rays = c["rays"]
    print(f"{name}{' unit %d' % items if items else ''}: {ms:.1f} ms {rays/ms/1e3:.0f} Mray/s  rays {rays} node_tests/ray {c['node_tests']/rays:.1f} tri/ray {c['tri_tests']/rays:.2f}")
    tot = t["traverse"] + t["setup"] + t["w_setup"] + t["shade"] + t["t_swap"] + t["t_gen"]
    def line(k, ticks, n, lanes=None):
        s = f"  {k:6s} {100.0*ticks/max(tot,1):5.1f}% of step time = {ticks / 1e5 / 4096:6.2f} ms per wave (4096 waves), {n} steps, {ticks*10.0/max(n,1):8.1f} ns/step"
        if lanes is not None: s += f", {lanes/max(n,1):5.1f} lanes/step"
        print(s)
    line("node", t["traverse"], t["w_node"], t["u_node"])
    line("tri", t["setup"], t["w_tri"], t["u_tri"])
    line("ctrl", t["w_setup"], t["w_ctrl"], t["u_ctrl"])
    line("swap", t["t_swap"], t["n_swap"], t["u_swap"])
    line("gen", t["t_gen"], t["n_gen"])
    line("shade", t["shade"], t["w_shade"], t["u_shade"])
    print(f"  rounds {t['w_round']}  steps per ray: node {t['u_node']/rays:.1f} lane-steps; node_tests/lane-step {c['node_tests']/max(t['u_node'],1):.2f}", flush=True)

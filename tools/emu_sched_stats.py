#!/usr/bin/env python3
"""emu_sched_stats.py — dev: the wave scheduler's step statistics WITHOUT a GPU, from the kernel emulation (tests/emu/libcray_hip_emu.so:
cray_hip.hip on the HIP-on-CPU shim). A unit's schedule depends only on the unit (the wave's path table starts empty), so with the GPU's
CU count (HIPEMU_CUS=256: the same work plan) the counts are the GPU's own, step for step: tools/probe_step_clocks.py on the GPU and this
script print the same node / tri / ctrl / swap / gen / shade step counts and lanes per step. Two cost models are printed from the ns / step the GPU
measured for that scene (profiles/r02d_probe_step_clocks.log): per step (A) and per lane-step (B); the hardware follows B.

    python tools/emu_sched_stats.py SCENE W H SPP BOUNCES [name=value ...]     e.g. cfg2_hdr 320 180 64 8 unit_items=4096 fill_to=192
"""
import os, sys, time
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ["CRH_LIB"] = os.path.join(REPO, "tests", "emu", "libcray_hip_emu.so")
os.environ["CRH_ALLOW_EMULATION"] = "1"
os.environ.setdefault("HIPEMU_CUS", "256")
sys.path.insert(0, REPO)
from __graft_entry__ import load_package, BUILT
pkg = load_package(); api = pkg.api; abi = pkg.abi
# ns per step on the MI355X (counting kernel, 16 waves per CU resident): profiles/r02d_probe_step_clocks.log
GPU_NS = {"cfg2_hdr": dict(node=1662.5, tri=1367.1, ctrl=1635.2, swap=1108.1, gen=4448.8, shade=18261.7),
          "cfg3_venus": dict(node=1283.1, tri=1175.9, ctrl=1602.5, swap=623.6, gen=2729.5, shade=10134.1),
          "soup_1m": dict(node=1291.7, tri=994.5, ctrl=1704.8, swap=956.3, gen=2770.6, shade=10263.5)}
name, w, h, spp, b = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5])
opts = dict(a.split("=") for a in sys.argv[6:])
path = os.path.join(BUILT, name + ".blob")
if not os.path.exists(path):
    import gzip, tempfile
    path = os.path.join(tempfile.gettempdir(), name + ".blob")
    open(path, "wb").write(gzip.open(os.path.join(REPO, "tests", "golden", name + ".blob.gz")).read())
ctx = api.Context(0)
ctx.set_option(abi.OPT_COUNTER_LEVEL, 2)
if "unit_items" in opts: ctx.set_option(abi.OPT_UNIT_ITEMS, int(opts["unit_items"]))
if "units_per_wave" in opts: ctx.set_option(abi.OPT_UNITS_PER_WAVE, int(opts["units_per_wave"]))
if "pass_chunk" in opts: ctx.set_option(abi.OPT_PASS_CHUNK, int(opts["pass_chunk"]))
if "kernel" in opts: ctx.set_option(abi.OPT_KERNEL, int(opts["kernel"]))        # 0 = one unit at a time, 1 = workgroup form, 2 = rolling units (the default)
if "tail" in opts: ctx.set_option(abi.OPT_TAIL_PERCENT, int(opts["tail"]))
sched = dict(node=70, tri=160, ctrl=120, swap_min=16, fill_to=160, run_num=4, tri_in_run=12, ctrl_in_run=12, shade_min=48, swap_in_run=20)
sched.update({k: int(v) for k, v in opts.items() if k in sched})
ctx.set_sched(**sched)
scene = api.Scene(path)
cam = scene.desc.camera
cam.width, cam.height = w, h
ctx.upload(scene)
fb = ctx.framebuffer(w, h)
ctx.reset_counters()
t0 = time.time()
ctx.render_region(fb, w, h, spp, b); ctx.synchronize()
secs = time.time() - t0
c = ctx.counters(); t = ctx.phase_ticks()
rays = c["rays"]
print(f"{name} {w}x{h} {spp} spp {b} bounces {opts}: emulated in {secs:.1f} s; rays {rays} paths {c['paths']} node_tests/ray {c['node_tests']/rays:.2f} tri/ray {c['tri_tests']/rays:.2f}")
rows = (("node", t["w_node"], t["u_node"]), ("tri", t["w_tri"], t["u_tri"]), ("ctrl", t["w_ctrl"], t["u_ctrl"]), ("swap", t["n_swap"], t["u_swap"]),
        ("gen", t["n_gen"], None), ("shade", t["w_shade"], t["u_shade"]))
ns = GPU_NS.get(name)
LANES = {"cfg2_hdr": dict(node=34.9, tri=19.9, ctrl=13.3, swap=52.5, shade=53.1),        # lanes per step of the runs GPU_NS comes from
         "cfg3_venus": dict(node=24.0, tri=2.7, ctrl=2.9, swap=22.6, shade=43.8), "soup_1m": dict(node=35.0, tri=5.8, ctrl=29.9, swap=32.2, shade=57.1)}
per_step = per_lane = 0.0
for k, n, lanes in rows:
    s = f"  {k:6s} {n:12d} steps"
    if lanes is not None: s += f", {lanes / max(n, 1):5.1f} lanes/step"
    if ns:
        per_step += n * ns[k]
        per_lane += (lanes * ns[k] / LANES[name][k]) if lanes is not None else n * ns[k]
        s += f", {n * ns[k] / 1e6:10.1f} wave-ms per step"
    print(s)
print(f"  rounds {t['w_round']}")
if t.get("w_tri_in") is not None and t["w_node"]:
    print(f"  inside node runs: {t['w_tri_in']} triangle steps at {t['u_tri_in'] / max(t['w_tri_in'], 1):.1f} lanes, {t['w_ctrl_in']} control steps at {t['u_ctrl_in'] / max(t['w_ctrl_in'], 1):.1f} lanes; "
          f"per node step {t['u_wait_tri'] / t['w_node']:.1f} lanes waited for a triangle step, {t['u_wait_fin'] / t['w_node']:.1f} for a retire / refill")
if ns:
    print(f"  model A (a step costs what the GPU measured per STEP):      {per_step / 1e6:9.1f} wave-ms = {per_step / 1e6 / 4096:.2f} ms on 4096 waves, {per_step / max(rays, 1):.1f} wave-ns per ray")
    print(f"  model B (a step costs what the GPU measured per LANE-step): {per_lane / 1e6:9.1f} wave-ms = {per_lane / 1e6 / 4096:.2f} ms on 4096 waves, {per_lane / max(rays, 1):.1f} wave-ns per ray")
    print("  (the MI355X follows B, not A: r02e_roll_step_clocks.log — 17 % fewer node steps at 20 % more lanes took the same time; neither model sees the cache misses\n"
          "   that a bigger working set adds: r02e_pmc_base_vs_roll.txt)", flush=True)

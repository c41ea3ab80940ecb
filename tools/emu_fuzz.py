#!/usr/bin/env python3
"""emu_fuzz.py — dev / test: schedule fuzzing of the REAL kernels on the kernel emulation (tests/emu/libcray_hip_emu.so). Every result is
independent of how a frame is dispatched and scheduled — so random work plans (unit size, units per wave, pass chunk, taper), scheduler
parameters (weights, run lengths, in-run thresholds, paths in flight, shade batch size), kernel forms (wave / workgroup / rolling
units / 3 = the streaming form of round 6 with a random pool size), device sizes (CUs, blocks per CU), tile decompositions and pass splits must all give the reference's frame bit for bit, with the
reference's ray count. Seeds are deterministic; a failing case prints its full configuration.

    python tools/emu_fuzz.py [--seeds A:B] [--fixtures fence,refraction,...] [--kernels 0,1,2]      (one JSON line per case)
"""
import argparse, gzip, json, os, random, sys, tempfile, time
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ap = argparse.ArgumentParser()
ap.add_argument("--seeds", default="0:8")
ap.add_argument("--fixtures", default="fence,refraction,glowmetal,volumes")
ap.add_argument("--kernels", default="0,1,2")
a = ap.parse_args()
lo, hi = (int(v) for v in a.seeds.split(":"))
fixtures = a.fixtures.split(",")
kernels = [int(k) for k in a.kernels.split(",")]
os.environ["CRH_LIB"] = os.environ.get("FUZZ_LIB") or os.path.join(REPO, "tests", "emu", "libcray_hip_emu.so")       # FUZZ_LIB: an emulation build with an experiment's macro
os.environ["CRH_ALLOW_EMULATION"] = "1"
sys.path.insert(0, REPO)
import numpy as np
from __graft_entry__ import load_package
man = json.load(open(os.path.join(REPO, "tests", "golden", "manifest.json")))
failures = 0
for seed in range(lo, hi):
    rng = random.Random(seed)
    name = fixtures[seed % len(fixtures)]
    m = man[name]
    w, h, s, b = m["width"], m["height"], m["samples"], m["bounces"]
    cfg = {
        "seed": seed, "fixture": name, "cus": rng.choice([1, 2, 3, 5]), "kernel": int(os.environ["FUZZ_KERNEL"]) if os.environ.get("FUZZ_KERNEL") else kernels[(seed // len(fixtures)) % len(kernels)],
        "blocks_per_cu": rng.choice([1, 2, 4]), "unit_items": rng.choice([64, 100, 256, 777, 2048, 5000, 1 << 16]),
        "units_per_wave": rng.choice([1, 2, 8, 64]), "pass_chunk": rng.choice([1, 2, 3, 64]), "tail": rng.choice([0, 7, 16, 50]) | (rng.choice([0, 1, 5, 31]) << 8),
        "w_node": rng.choice([1, 70, 500]), "w_tri": rng.choice([1, 160, 900]), "w_ctrl": rng.choice([1, 120, 4000]), "swap_min": rng.choice([1, 8, 16, 40, 64]),
        "fill_to": rng.choice([0, 1, 64, 100, 160, 192]), "run_num": rng.choice([1, 4, 8]), "tri_in_run": rng.choice([1, 12, 65]), "ctrl_in_run": rng.choice([1, 12, 65]),
        "shade_min": rng.choice([1, 17, 48, 64, 128]), "sort_from": rng.choice([0, 0, 1, 2, 4]),
        # the workgroup kernel's own scheduler (used when kernel == 1)
        "wg_linger": rng.choice([0, 1, 8, 255]), "wg_drain_at": rng.choice([1, 64, 192, 1000, 4095]), "wg_max_drainers": rng.choice([0, 1, 4]),
        "wg_partial_min": rng.choice([1, 16, 64, 255]), "wg_walk_min": rng.choice([1, 32, 64]), "wg_fill_to": rng.choice([0, 64, 768, 960]),
    }
    # a random cover of the frame by rectangles (guillotine cuts), rendered in 1..3 pass ranges
    rects = [(0, 0, w, h)]
    for _ in range(rng.choice([0, 1, 3, 9])):
        i = rng.randrange(len(rects)); x0, y0, x1, y1 = rects.pop(i)
        if rng.random() < 0.5 and x1 - x0 > 1: c = rng.randrange(x0 + 1, x1); rects += [(x0, y0, c, y1), (c, y0, x1, y1)]
        elif y1 - y0 > 1: c = rng.randrange(y0 + 1, y1); rects += [(x0, y0, x1, c), (x0, c, x1, y1)]
        else: rects.append((x0, y0, x1, y1))
    rng.shuffle(rects)
    cuts = sorted(set([0, s] + [rng.randrange(0, s + 1) for _ in range(rng.choice([0, 1, 2]))]))
    cfg["tiles"] = len(rects); cfg["pass_cuts"] = cuts
    cfg["tail_split"] = rng.choice([0, 1, 4, 64])          # (the rolling kernel's 64-path units at the end of the queue; pass segments of split pixels: tools/emu_fuzz_split.py)
    cfg["swap_in_run"] = rng.choice([1, 4, 20, 33, 65])          # (drawn after everything else: the earlier fields of a seed stay what they were)
    cfg["stream_cohorts"] = rng.choice([1, 2, 5, 37, 16384])          # (round 6: the pool size of the streaming form, kernel == 3 — one cohort: hundreds of iterations and a turning sample ring)
    os.environ["HIPEMU_CUS"] = str(cfg["cus"])
    pkg = load_package(); api, abi = pkg.api, pkg.abi
    t0 = time.time()
    with tempfile.NamedTemporaryFile(suffix=".blob") as f:
        f.write(gzip.open(os.path.join(REPO, "tests", "golden", name + ".blob.gz")).read()); f.flush()
        scene = api.Scene(f.name)
    ref = np.frombuffer(gzip.open(os.path.join(REPO, "tests", "golden", name + ".ref.f32.gz")).read(), dtype=np.float32)
    ctx = api.Context(0)
    ctx.set_option(abi.OPT_COUNTER_LEVEL, rng.choice([1, 2]))
    ctx.set_option(abi.OPT_KERNEL, cfg["kernel"])
    ctx.set_option(abi.OPT_BLOCKS_PER_CU, cfg["blocks_per_cu"])
    ctx.set_option(abi.OPT_UNIT_ITEMS, cfg["unit_items"]); ctx.set_option(abi.OPT_UNITS_PER_WAVE, cfg["units_per_wave"])
    ctx.set_option(abi.OPT_PASS_CHUNK, cfg["pass_chunk"]); ctx.set_option(abi.OPT_TAIL_PERCENT, cfg["tail"])
    ctx.set_sched(cfg["w_node"], cfg["w_tri"], cfg["w_ctrl"], cfg["swap_min"], fill_to=cfg["fill_to"], run_num=cfg["run_num"],
                  tri_in_run=cfg["tri_in_run"], ctrl_in_run=cfg["ctrl_in_run"], shade_min=cfg["shade_min"], swap_in_run=cfg["swap_in_run"])
    ctx.set_option(abi.OPT_SHADE_SORT, cfg["sort_from"])
    ctx.set_option(abi.OPT_TAIL_SPLIT, cfg["tail_split"])
    ctx.set_option(abi.OPT_STREAM_COHORTS, cfg["stream_cohorts"])
    if cfg["kernel"] == 1:
        ctx.set_sched_wg(linger=cfg["wg_linger"], drain_at=cfg["wg_drain_at"], max_drainers=cfg["wg_max_drainers"], partial_min=cfg["wg_partial_min"],
                         walk_min=cfg["wg_walk_min"], fill_to=cfg["wg_fill_to"])
    ctx.upload(scene)
    fb = ctx.framebuffer(w, h)
    ctx.reset_counters()
    for p0, p1 in zip(cuts[:-1], cuts[1:]):
        ctx.render_tiles(fb, w, h, s, b, rects, first_pass=p0, pass_count=p1 - p0)
    img = ctx.download(fb, w, h)
    cnt = ctx.counters()
    ctx.close()
    ok = bool(np.array_equal(img.ravel().view(np.uint32), ref.view(np.uint32))) and cnt["rays"] == m["rays"] and cnt["paths"] == w * h * s
    failures += 0 if ok else 1
    print(json.dumps({"ok": ok, "secs": round(time.time() - t0, 1), **cfg, "rays": cnt["rays"], "want_rays": m["rays"]}), flush=True)
sys.exit(1 if failures else 0)

#!/usr/bin/env python3
"""emu_fuzz_split.py — dev / test: the rolling kernel's SPLIT PIXELS (CRH_OPT_TAIL_SPLIT, include/cray_hip.h) fuzzed on the kernel emulation
(tests/emu/libcray_hip_emu.so). From 128 passes per dispatch on the work queue ends with pass segments of single pixels; whichever waves pull
them trace them, their samples are staged per pixel and k_fold_deferred folds them behind the kernel. The golden fixtures hold few passes per
pixel, so this fuzz renders small crops of them with 128..330 passes and compares with the one-unit-at-a-time kernel (CRH_KERNEL_WAVE: no pixel is
ever split there; its frames are pinned to the reference's by the golden fixtures): random device sizes, split units per wave, unit sizes, taper,
tile covers with ragged edges and pass ranges (a later dispatch continues the running mean of an earlier one; short ranges are not split) must
all give the same frame bit for bit and the same ray count. Seeds are deterministic; a failing case prints its configuration.

    python tools/emu_fuzz_split.py [--seeds A:B] [--fixtures refraction,glowmetal,...]      (one JSON line per case)
"""
import argparse, gzip, json, os, random, sys, tempfile, time
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ap = argparse.ArgumentParser()
ap.add_argument("--seeds", default="0:6")
ap.add_argument("--fixtures", default="refraction,glowmetal,volumes,fence")
a = ap.parse_args()
lo, hi = (int(v) for v in a.seeds.split(":"))
fixtures = a.fixtures.split(",")
os.environ["CRH_LIB"] = os.environ.get("FUZZ_LIB") or os.path.join(REPO, "tests", "emu", "libcray_hip_emu.so")       # FUZZ_LIB: an emulation build with an experiment's macro
os.environ["CRH_ALLOW_EMULATION"] = "1"
sys.path.insert(0, REPO)
import numpy as np
from __graft_entry__ import load_package
man = json.load(open(os.path.join(REPO, "tests", "golden", "manifest.json")))
failures = 0
for seed in range(lo, hi):
    rng = random.Random(1000 + seed)
    name = fixtures[seed % len(fixtures)]
    b = man[name]["bounces"]
    w, h = rng.randrange(9, 40), rng.randrange(5, 24)
    s = rng.choice([128, 129, 160, 191, 192, 256, 257, 330])
    cfg = {"seed": seed, "fixture": name, "width": w, "height": h, "samples": s, "cus": rng.choice([1, 2, 3, 5]), "blocks_per_cu": rng.choice([1, 2, 4]),
           "tail_split": rng.choice([1, 2, 4, 8, 64]), "unit_items": rng.choice([64, 256, 2048, 1 << 16]), "units_per_wave": rng.choice([1, 8, 64]),
           "pass_chunk": rng.choice([1, 3, 64]), "tail": rng.choice([0, 16, 50]) | (rng.choice([0, 5, 31]) << 8), "fill_to": rng.choice([0, 64, 160, 192]),
           "counter_level": rng.choice([1, 2])}
    rects = [(0, 0, w, h)]
    for _ in range(rng.choice([0, 1, 3, 6])):
        i = rng.randrange(len(rects)); x0, y0, x1, y1 = rects.pop(i)
        if rng.random() < 0.5 and x1 - x0 > 1: c = rng.randrange(x0 + 1, x1); rects += [(x0, y0, c, y1), (c, y0, x1, y1)]
        elif y1 - y0 > 1: c = rng.randrange(y0 + 1, y1); rects += [(x0, y0, x1, c), (x0, c, x1, y1)]
        else: rects.append((x0, y0, x1, y1))
    rng.shuffle(rects)
    cuts = sorted(set([0, s] + [rng.randrange(0, s + 1) for _ in range(rng.choice([0, 1, 2]))]))
    cfg["tiles"] = len(rects); cfg["pass_cuts"] = cuts
    os.environ["HIPEMU_CUS"] = str(cfg["cus"])
    pkg = load_package(); api, abi = pkg.api, pkg.abi
    t0 = time.time()
    with tempfile.NamedTemporaryFile(suffix=".blob") as f:
        f.write(gzip.open(os.path.join(REPO, "tests", "golden", name + ".blob.gz")).read()); f.flush()
        scene = api.Scene(f.name)
    frames, rays, split_units = [], [], 0
    for kernel in (abi.KERNEL_WAVE, abi.KERNEL_ROLL):
        ctx = api.Context(0)
        ctx.set_option(abi.OPT_KERNEL, kernel)
        ctx.set_option(abi.OPT_COUNTER_LEVEL, cfg["counter_level"])
        if kernel == abi.KERNEL_ROLL:
            ctx.set_option(abi.OPT_BLOCKS_PER_CU, cfg["blocks_per_cu"])
            ctx.set_option(abi.OPT_UNIT_ITEMS, cfg["unit_items"]); ctx.set_option(abi.OPT_UNITS_PER_WAVE, cfg["units_per_wave"])
            ctx.set_option(abi.OPT_PASS_CHUNK, cfg["pass_chunk"]); ctx.set_option(abi.OPT_TAIL_PERCENT, cfg["tail"])
            ctx.set_sched(70, 160, 120, 16, fill_to=cfg["fill_to"])
            ctx.set_option(abi.OPT_TAIL_SPLIT, cfg["tail_split"])
        ctx.upload(scene)
        fb = ctx.framebuffer(w, h)
        ctx.reset_counters()
        for p0, p1 in zip(cuts[:-1], cuts[1:]):
            ctx.render_tiles(fb, w, h, s, b, rects, first_pass=p0, pass_count=p1 - p0)
        frames.append(ctx.download(fb, w, h))
        c = ctx.counters()
        rays.append((c["rays"], c["paths"]))
        ctx.close()
    ok = bool(np.array_equal(frames[0].view(np.uint32), frames[1].view(np.uint32))) and rays[0] == rays[1] and rays[0][1] == w * h * s
    failures += 0 if ok else 1
    print(json.dumps({"ok": ok, "secs": round(time.time() - t0, 1), **cfg, "rays": rays[1][0], "want_rays": rays[0][0]}), flush=True)
sys.exit(1 if failures else 0)

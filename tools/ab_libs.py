#!/usr/bin/env python3
"""ab_libs.py — dev probe: A/B several builds of libcray_hip.so (c-ray_amd/_lib/variants/*.so, built by tools/build_variants.sh) on
the BASELINE scenes. Every build renders in its own process (CRH_LIB); frames must hash equal to the first build's.

    python tools/ab_libs.py [--quick] [name ...]          -> gpurun_out/ab_libs.json
"""
import glob, hashlib, json, os, subprocess, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CASES = [("cfg2_hdr", 1280, 720, 256, 8, 3), ("cfg4_statues", 3840, 2160, 4, 30, 2), ("soup_1m", 2560, 1440, 16, 8, 2), ("cfg3_venus", 1920, 1080, 16, 32, 2)]

if os.environ.get("AB_CHILD"):
    sys.path.insert(0, REPO)
    from __graft_entry__ import load_package, BUILT
    pkg = load_package(); api = pkg.api; abi = pkg.abi
    quick = os.environ.get("AB_QUICK") == "1"
    ctx = api.Context(0)
    ctx.set_option(abi.OPT_COUNTER_LEVEL, 1)
    if os.environ.get("AB_SCHED"):                 # "fill_to,shade_min" (NAME.env of a build)
        f, sm = (int(v) for v in os.environ["AB_SCHED"].split(","))
        ctx.set_sched(70, 160, 120, 16, fill_to=f, shade_min=sm)
    if os.environ.get("AB_KERNEL"):                # CRH_OPT_KERNEL of a build with an experimental kernel form (NAME.env)
        ctx.set_option(abi.OPT_KERNEL, int(os.environ["AB_KERNEL"]))
    if os.environ.get("AB_UNIT_ITEMS"):
        ctx.set_option(abi.OPT_UNIT_ITEMS, int(os.environ["AB_UNIT_ITEMS"]))
    res = {}
    cases = CASES[:3] if quick else CASES
    if os.environ.get("AB_CASES"):                 # e.g. AB_CASES=cfg2_hdr,soup_1m
        cases = [c for c in CASES if c[0] in os.environ["AB_CASES"].split(",")]
    for name, w, h, spp, b, reps in cases:
        path = os.path.join(BUILT, name + ".blob")
        if not os.path.exists(path):
            continue
        ctx.upload(api.Scene(path))
        fb = ctx.framebuffer(w, h)
        best = None
        for _ in range(reps):
            ctx.clear(fb, w, h); ctx.reset_counters()
            ctx.render_region(fb, w, h, spp, b); ctx.synchronize()
            ms = ctx.kernel_time_ms()[0]
            best = ms if best is None else min(best, ms)
        rays = ctx.counters()["rays"]
        img = ctx.download(fb, w, h)
        res[name] = {"ms": round(best, 2), "mrays": round(rays / best / 1e3, 1), "rays": rays, "md5": hashlib.md5(img.tobytes()).hexdigest()}
    print("AB_RESULT " + json.dumps(res), flush=True)
    sys.exit(0)

names = [a for a in sys.argv[1:] if not a.startswith("--")]
libs = sorted(glob.glob(os.path.join(REPO, "c-ray_amd", "_lib", "variants", "*.so")))
if names:
    libs = [l for l in libs if os.path.basename(l)[:-3] in names]
out = {}
first = None
for lib in libs:
    tag = os.path.basename(lib)[:-3]
    env = dict(os.environ, CRH_LIB=lib, AB_CHILD="1", AB_QUICK="1" if "--quick" in sys.argv else "0")
    if os.path.exists(lib[:-3] + ".env"):           # NAME.env: extra environment of that build (e.g. CRH_BLOCKS_PER_CU=3)
        env.update(l.strip().split("=", 1) for l in open(lib[:-3] + ".env") if "=" in l)
    try:
        r = subprocess.run([sys.executable, __file__], env=env, capture_output=True, text=True, timeout=int(os.environ.get("AB_TIMEOUT", "240")))
    except subprocess.TimeoutExpired:
        print(tag, "TIMEOUT", flush=True)
        continue
    line = [l for l in r.stdout.splitlines() if l.startswith("AB_RESULT ")]
    if not line:
        print(tag, "FAILED rc", r.returncode, r.stderr.strip().splitlines()[-3:], flush=True)
        continue
    res = json.loads(line[0][len("AB_RESULT "):])
    if first is None:
        first = res
    for k, v in res.items():
        v["same_frame"] = v["md5"] == first[k]["md5"] and v["rays"] == first[k]["rays"]
        v["vs_first"] = round(first[k]["ms"] / v["ms"], 4)
    out[tag] = res
    print(f"{tag:28s} " + "  ".join(f"{k}: {v['mrays']:7.1f} ({v['vs_first']:.3f}{'' if v['same_frame'] else ' FRAME DIFFERS'})" for k, v in res.items()), flush=True)
os.makedirs(os.path.join(REPO, "gpurun_out"), exist_ok=True)
json.dump(out, open(os.path.join(REPO, "gpurun_out", "ab_libs.json"), "w"), indent=1)

#!/usr/bin/env python3
"""probe_fold_any.py — dev probe (one GPU): the product library against a variant (default: c-ray_amd/_lib/variants/fold_any.so, built with -DCRH_EXP_FOLD_ANY: the rolling kernel folds a
complete job whatever its slot instead of in ring order) for several unit sizes: the bench frame as one dispatch and as the slower of ranks 0 / 7 of a 1/8 share, and the other BASELINE
workloads at bench.py's pass counts. Every library renders in its own process; frames must hash equal. Kernel time = the best of three dispatches."""
import hashlib, json, os, subprocess, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CASES = [("cfg2_hdr", 1280, 720, 256, 8), ("cfg4_statues", 3840, 2160, 4, 30), ("soup_1m", 2560, 1440, 16, 8), ("cfg3_venus", 1920, 1080, 16, 32)]
UNITS = [2048, 1024, 512, 256]
if os.environ.get("PFA_CASES"):          # e.g. PFA_CASES=cfg2_hdr,cfg4_statues ; PFA_UNITS=2048,512
    CASES = [c for c in CASES if c[0] in os.environ["PFA_CASES"].split(",")]
if os.environ.get("PFA_UNITS"):
    UNITS = [int(v) for v in os.environ["PFA_UNITS"].split(",")]
if os.environ.get("PFA_CHILD"):
    sys.path.insert(0, REPO)
    from __graft_entry__ import load_package, BUILT
    pkg = load_package(); api = pkg.api; abi = pkg.abi
    ctx = api.Context(0); ctx.set_option(abi.OPT_COUNTER_LEVEL, 1); ctx.set_option(abi.OPT_WAVE_STATS, 1)
    out = {}
    for name, w, h, spp, b in CASES:
        ctx.upload(api.Scene(os.path.join(BUILT, name + ".blob")))
        fb = ctx.framebuffer(w, h)
        def run(tiles):
            best = None
            for rep in range(3):
                ctx.clear(fb, w, h); ctx.reset_counters(); ctx.render_tiles(fb, w, h, spp, b, tiles); ctx.synchronize()
                ms = ctx.kernel_time_ms()[0]; ws = ctx.wave_stats()
                if best is None or ms < best[0]: best = (ms, float(ws[:, 0].mean() / 1e5))
            return best
        for items in (UNITS if name == "cfg2_hdr" or os.environ.get("PFA_UNITS") else UNITS[:3:2]):
            ctx.set_option(abi.OPT_UNIT_ITEMS, items)
            full = run(pkg.render.owned_tiles(w, h, 64, 64, 1, 0, 1))
            md5 = hashlib.md5(ctx.download(fb, w, h).tobytes()).hexdigest()
            share = max(run(pkg.render.owned_tiles(w, h, 64, 64, 1, r, 8)) for r in (0, 7))
            out[f"{name} unit {items}"] = {"full_ms": round(full[0], 2), "full_mean_wave_ms": round(full[1], 2), "share8_ms": round(share[0], 2), "share8_mean_wave_ms": round(share[1], 2), "md5": md5}
        ctx.set_option(abi.OPT_UNIT_ITEMS, 2048)
    print("PFA_RESULT " + json.dumps(out), flush=True)
    sys.exit(0)
# VARIANTS=a.so,b.so (paths or names under c-ray_amd/_lib/variants): several variants against the product
vnames = (os.environ.get("VARIANTS") or "fold_any").split(",")
vpath = lambda n: n if os.path.sep in n else os.path.join(REPO, "c-ray_amd", "_lib", "variants", n if n.endswith(".so") else n + ".so")
libs = [("product", os.path.join(REPO, "c-ray_amd", "_lib", "libcray_hip.so"))] + [(os.path.basename(vpath(n))[:-3], vpath(n)) for n in vnames]
res = {}
for tag, lib in libs * (1 if os.environ.get("PFA_ONCE") else 2):                     # each library twice, alternating (the later numbers count: warm clocks); PFA_ONCE=1: once
    r = subprocess.run([sys.executable, __file__], env=dict(os.environ, CRH_LIB=lib, PFA_CHILD="1"), capture_output=True, text=True, timeout=200)
    line = [l for l in r.stdout.splitlines() if l.startswith("PFA_RESULT ")]
    if not line:
        print(tag, "FAILED", r.stderr[-800:]); continue
    res[tag] = json.loads(line[0][len("PFA_RESULT "):])
for vtag in [t for t, _ in libs[1:]]:
  print(f"== {vtag} against the product", flush=True)
  for key in res.get("product", {}):
    p, v = res["product"][key], res.get(vtag, {}).get(key)
    if not v: continue
    same = "same frame" if p["md5"] == v["md5"] else "FRAMES DIFFER"
    print(f"  {key:26s} full {p['full_ms']:7.2f} -> {v['full_ms']:7.2f} ms ({p['full_ms'] / v['full_ms']:.3f}x)   1/8 share {p['share8_ms']:6.2f} -> {v['share8_ms']:6.2f} ms ({p['share8_ms'] / v['share8_ms']:.3f}x; "
            f"mean wave {p['share8_mean_wave_ms']:.2f} -> {v['share8_mean_wave_ms']:.2f}; ceiling {p['full_ms'] / p['share8_ms']:.2f} -> {v['full_ms'] / v['share8_ms']:.2f}x)   {same}", flush=True)

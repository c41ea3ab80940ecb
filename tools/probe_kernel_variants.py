#!/usr/bin/env python3
"""probe_kernel_variants.py — dev probe: run each kernel variant (waves/SIMD x counter level) in its own process, report faults."""
import os, subprocess, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 1:
    sys.path.insert(0, REPO)
    import numpy as np
    from __graft_entry__ import load_package
    pkg = load_package(); api = pkg.api; abi = pkg.abi
    wps, level, name, w, h, s, b = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3], *map(int, sys.argv[4:8])
    ctx = api.Context(0)
    import gzip, tempfile
    tmp = os.path.join(tempfile.gettempdir(), name + ".blob")
    open(tmp, "wb").write(gzip.open(os.path.join(REPO, "tests", "golden", name + ".blob.gz"), "rb").read())
    ctx.upload(api.Scene(tmp))
    fb = ctx.framebuffer(w, h)
    ctx.set_option(abi.OPT_WAVES_PER_SIMD, wps); ctx.set_option(abi.OPT_COUNTER_LEVEL, level)
    ctx.clear(fb, w, h); ctx.render_region(fb, w, h, s, b)
    img = ctx.download(fb, w, h)
    print("ok", wps, level, name, float(img.sum()), ctx.counters()["rays"], flush=True)
    sys.exit(0)
import json
man = json.load(open(os.path.join(REPO, "tests", "golden", "manifest.json")))
for name in sorted(man):
    m = man["cases"][name] if "cases" in man else man[name]
    for wps, level in ((4, 1), (1, 1)):
        r = subprocess.run([sys.executable, __file__, str(wps), str(level), name, str(m["width"]), str(m["height"]), str(m["samples"]), str(m["bounces"])],
                           capture_output=True, text=True, timeout=300)
        print(name, wps, level, "rc", r.returncode, (r.stdout.strip().splitlines() or ["-"])[-1], (r.stderr.strip().splitlines() or ["-"])[-1][:200], flush=True)

#!/usr/bin/env python3
"""render_fixture.py — child process of tests/test_kernel_emu.py: render golden fixtures through api.py with the library CRH_LIB names
(the kernel emulation) and a given CRH_OPT_KERNEL; prints one JSON line per fixture {name, equal, rays, paths}. TEST INFRASTRUCTURE."""
import gzip, json, os, sys, tempfile
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO)
import numpy as np
from __graft_entry__ import load_package
pkg = load_package(); api, abi = pkg.api, pkg.abi
kernel = int(sys.argv[1])
golden = os.path.join(REPO, "tests", "golden")
man = json.load(open(os.path.join(golden, "manifest.json")))
for name in sys.argv[2:]:
    m = man[name]
    w, h, s, b = m["width"], m["height"], m["samples"], m["bounces"]
    with tempfile.NamedTemporaryFile(suffix=".blob") as f:
        f.write(gzip.open(os.path.join(golden, name + ".blob.gz")).read()); f.flush()
        scene = api.Scene(f.name)
    ref = np.frombuffer(gzip.open(os.path.join(golden, name + ".ref.f32.gz")).read(), dtype=np.float32)
    frames = []
    for unit_items in (2048, 256):            # 256: many small jobs per wave (every slot of the ring in use, units of a few pixels)
        ctx = api.Context(0)
        ctx.set_option(abi.OPT_KERNEL, kernel)
        ctx.set_option(abi.OPT_UNIT_ITEMS, unit_items)
        ctx.upload(scene)
        fb = ctx.framebuffer(w, h)
        ctx.reset_counters()
        ctx.render_region(fb, w, h, s, b)
        img = ctx.download(fb, w, h)
        cnt = ctx.counters()
        ctx.close()
        frames.append(bool(np.array_equal(img.ravel().view(np.uint32), ref.view(np.uint32))) and cnt["rays"] == m["rays"] and cnt["paths"] == w * h * s)
    print(json.dumps({"name": name, "equal": all(frames), "rays": cnt["rays"], "paths": cnt["paths"]}), flush=True)

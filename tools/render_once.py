#!/usr/bin/env python3
"""render_once.py — dev: one dispatch through the C-ABI (no torch), for rocprofv3 --pmc passes on a chosen kernel form.
    CRH_LIB=<variant .so> python tools/render_once.py SCENE W H SPP BOUNCES [CRH_OPT_KERNEL]"""
import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from __graft_entry__ import load_package, BUILT
pkg = load_package(); api = pkg.api; abi = pkg.abi
name, w, h, spp, b = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5])
ctx = api.Context(0)
ctx.set_option(abi.OPT_COUNTER_LEVEL, 1)
if len(sys.argv) > 6:
    ctx.set_option(abi.OPT_KERNEL, int(sys.argv[6]))
ctx.upload(api.Scene(os.path.join(BUILT, name + ".blob")))
fb = ctx.framebuffer(w, h)
ctx.reset_counters()
ctx.render_region(fb, w, h, spp, b); ctx.synchronize()
print(name, w, h, spp, b, sys.argv[6:] or "default kernel", f"{ctx.kernel_time_ms()[0]:.2f} ms", ctx.counters()["rays"], "rays", flush=True)

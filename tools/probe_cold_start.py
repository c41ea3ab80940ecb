import os, sys, time
sys.path.insert(0, os.getcwd())
from __graft_entry__ import load_package, BUILT
pkg = load_package(); api = pkg.api; abi = pkg.abi
ctx = api.Context(0)
ctx.set_option(abi.OPT_COUNTER_LEVEL, 1)
ctx.upload(api.Scene(os.path.join(BUILT, "cfg2_hdr.blob")))
w, h = 1280, 720
fb = ctx.framebuffer(w, h)
for i in range(4):
    ctx.clear(fb, w, h); ctx.reset_counters()
    t = time.perf_counter()
    ctx.render_region(fb, w, h, 256, 8); ctx.synchronize()
    wall = (time.perf_counter() - t) * 1e3
    print(i, "kernel ms", round(ctx.kernel_time_ms()[0], 2), "wall ms", round(wall, 2), flush=True)

#!/usr/bin/env python3
"""probe_share_scaling.py — dev probe (round 5): what ONE GPU can say about the 1/2/4/8-GPU curve of the bench line's scaling objects — a rank's share (every N-th 4-row strip: render.py
owned_tiles) of BASELINE configs[3] (statues.json 3840x2160) and configs[4] (the 10 M-triangle soup 2560x1440) at the sample count bench.py's scaling_cfg4 / scaling_soup10m use, the slowest of
three sampled ranks against the full frame on the same GPU: the ceiling of the strong-scaling factor before the strip gather (1 / N of the float frame per xGMI link) and the launch."""
import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from __graft_entry__ import load_package, BUILT
import bench
pkg = load_package(); api = pkg.api; abi = pkg.abi
for key in ("cfg4", "soup10m"):
    W = bench.WORKLOADS[key]
    blob = bench.workload_blob(key, BUILT)
    ctx = api.Context(0); ctx.set_option(abi.OPT_COUNTER_LEVEL, 1)
    ctx.upload(api.Scene(blob))
    w, h, spp, b = W["width"], W["height"], bench.SCALING_SPP[key], W["bounces"]
    fb = ctx.framebuffer(w, h)
    def run(tiles):
        best = None
        for rep in range(2):
            ctx.clear(fb, w, h); ctx.reset_counters(); ctx.render_tiles(fb, w, h, spp, b, tiles); ctx.synchronize()
            ms = ctx.kernel_time_ms()[0]; rays = ctx.counters()["rays"]
            if best is None or ms < best[0]: best = (ms, rays)
        return best
    full = run(pkg.render.owned_tiles(w, h, 64, 64, 1, 0, 1))
    print(f"{key} {w}x{h} {spp} spp: full frame {full[0]:.1f} ms, {full[1] / full[0] / 1e3:.0f} Mray/s", flush=True)
    for world in (2, 4, 8):
        shares = [run(pkg.render.owned_tiles(w, h, 64, 64, 1, r, world)) for r in sorted({0, world // 2, world - 1})]
        worst = max(s[0] for s in shares)
        print(f"  world {world}: shares {', '.join('%.1f' % s[0] for s in shares)} ms ({', '.join('%.0f' % (s[1] / 1e6) for s in shares)} Mrays) -> ceiling {full[0] / worst:.2f}x", flush=True)
    ctx.close()

#!/usr/bin/env python3
"""probe_tail_split.py — dev probe (one GPU): what CRH_OPT_TAIL_SPLIT buys. The bench frame (hdr.json, 256 passes) as one dispatch and as one rank's
share at world size 2 / 4 / 8 (every N-th 4-row strip), and the other BASELINE workloads at the pass counts bench.py times them with, for several numbers of
64-path units per wave at the end of the work queue (0 = the taper ends with single pixels x all passes, as before): kernel time (the best of three
dispatches; it includes k_fold_deferred), how far the last wave finishes behind the average one, and the share ceilings (full frame / slowest share).
Writes the setting that is best for the bench frame's 1/8 share without costing any workload more than 0.5 % to gpurun_out/tail_split_best.txt."""
import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from __graft_entry__ import load_package, BUILT
pkg = load_package(); api = pkg.api; abi = pkg.abi
SPLITS = [int(v) for v in os.environ.get("SPLITS", "0,2,4,8,16").split(",")]
CASES = [("cfg2_hdr", 1280, 720, 256, 8), ("cfg4_statues", 3840, 2160, 4, 30), ("soup_1m", 2560, 1440, 16, 8), ("cfg3_venus", 1920, 1080, 16, 32)]
if os.environ.get("PROBE_CASES"):          # dry runs (the emulation): "name,w,h,spp,bounces;..."
    CASES = [(c.split(",")[0], *[int(v) for v in c.split(",")[1:]]) for c in os.environ["PROBE_CASES"].split(";")]
ctx = api.Context(0); ctx.set_option(abi.OPT_COUNTER_LEVEL, 1); ctx.set_option(abi.OPT_WAVE_STATS, 1)
res = {}
for name, w, h, spp, b in CASES:
    path = os.path.join(BUILT, name + ".blob")
    if not os.path.exists(path):
        continue
    ctx.upload(api.Scene(path))
    fb = ctx.framebuffer(w, h)

    def run(tiles):
        best = None
        for rep in range(3):
            ctx.clear(fb, w, h); ctx.reset_counters(); ctx.render_tiles(fb, w, h, spp, b, tiles); ctx.synchronize()
            ms = ctx.kernel_time_ms()[0]; rays = ctx.counters()["rays"]; ws = ctx.wave_stats()
            if best is None or ms < best[0]: best = (ms, rays, ws[:, 0].mean() / 1e5, ws[:, 0].max() / 1e5, ws[:, 1].mean())
        return best
    worlds = (1, 2, 4, 8) if name == CASES[0][0] else (1, 8)
    for split in SPLITS:
        ctx.set_option(abi.OPT_TAIL_SPLIT, split)
        line = []
        for world in worlds:
            worst = None
            for rank in sorted({0, world - 1}):
                r = run(pkg.render.owned_tiles(w, h, 64, 64, 1, rank, world))
                if worst is None or r[0] > worst[0]: worst = r
            res[(name, split, world)] = worst
            full = res[(name, split, 1)][0]
            line.append(f"world {world}: {worst[0]:7.2f} ms ({worst[1] / worst[0] / 1e3:6.0f} Mray/s, last wave {worst[3] - worst[2]:4.2f} ms behind the mean {worst[2]:6.2f}, {worst[4]:5.1f} units/wave"
                        + (f", ceiling {full / worst[0]:.2f}x)" if world > 1 else ")"))
        print(f"{name:13s} split {split:2d}  " + "  ".join(line), flush=True)
# the setting: best 1/8 share of the bench frame among those that cost no workload's full dispatch more than 0.5 % against split 0
ok = [s for s in SPLITS if all(res[(n, s, 1)][0] <= 1.005 * res[(n, 0, 1)][0] for n, *_ in CASES if (n, s, 1) in res)]
best = min(ok or [0], key=lambda s: res[(CASES[0][0], s, 8)][0])
os.makedirs(os.path.join(REPO, "gpurun_out"), exist_ok=True)
open(os.path.join(REPO, "gpurun_out", "tail_split_best.txt"), "w").write(str(best) + "\n")
print(f"settings within 0.5 % of split 0 on every full dispatch: {ok}; best 1/8 share of the bench frame: split {best}", flush=True)

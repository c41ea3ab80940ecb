#!/usr/bin/env python3
"""probe_finish.py — dev probe (one GPU; needs a library built with -DCRH_EXP_ABS_TIMES: CRH_LIB=c-ray_amd/_lib/variants/abs_times.so): WHERE the end of a
dispatch goes. Every wave of k_pathtrace_roll records, on the chip-wide 100 MHz clock, when it started, when it first found the work queue empty and when it
ended. For the bench frame (hdr.json, 256 passes) as one dispatch and as a 1/8 share: how far apart the waves start, when the queue runs dry, how long a wave
needs from there to its end (the drain of its open jobs and its path table), and what the latest waves look like."""
import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import numpy as np
from __graft_entry__ import load_package, BUILT
import bench
pkg = load_package(); api = pkg.api; abi = pkg.abi
W = bench.WORKLOAD
w, h, spp, b = W["width"], W["height"], W["samples"], W["bounces"]
blob = W["blob"]
if os.environ.get("PROBE_CASE"):          # a dry run on the emulation: "name,w,h,spp,bounces"
    blob, w, h, spp, b = (lambda c: (c[0], int(c[1]), int(c[2]), int(c[3]), int(c[4])))(os.environ["PROBE_CASE"].split(","))
ctx = api.Context(0); ctx.set_option(abi.OPT_COUNTER_LEVEL, 1); ctx.set_option(abi.OPT_WAVE_STATS, 1)
ctx.upload(api.Scene(os.path.join(BUILT, blob + ".blob")))
fb = ctx.framebuffer(w, h)
q = lambda a, ps: " ".join(f"{v:.2f}" for v in np.quantile(a, ps))
for split in [int(v) for v in os.environ.get("PROBE_SPLITS", "0,16").split(",")]:
    ctx.set_option(abi.OPT_TAIL_SPLIT, split)
    for label, tiles in (("world 1", pkg.render.owned_tiles(w, h, 64, 64, 1, 0, 1)), ("world 8 rank 0", pkg.render.owned_tiles(w, h, 64, 64, 1, 0, 8))):
        for rep in range(2):
            ctx.clear(fb, w, h); ctx.reset_counters(); ctx.render_tiles(fb, w, h, spp, b, tiles); ctx.synchronize()
        ms = ctx.kernel_time_ms()[0]
        ws = ctx.wave_stats().astype(np.uint64)
        snap = ws[len(ws) // 2:]; ws = ws[:len(ws) // 2]          # second record per wave: what it held when the work counter passed the last unit
        start = (ws[:, 0] - ws[:, 0].min()).astype(np.float64) / 1e5                       # ms after the first wave's start
        busy = (ws[:, 1] & np.uint64(0xFFFFFFFF)).astype(np.float64) / 1e5
        dry = (ws[:, 1] >> np.uint64(32)).astype(np.float64) / 1e5                          # own start -> queue found empty (= busy for a wave that never asked again)
        end = start + busy
        drain = busy - dry
        print(f"== tail split {split}, {label}: kernel {ms:.2f} ms, {len(ws)} waves")
        print(f"   starts: all within {start.max():.3f} ms (quantiles 50/90/99/100 %: {q(start, [0.5, 0.9, 0.99, 1.0])})")
        print(f"   queue found empty at (ms after the first start) 1/50/99 %: {q(start + dry, [0.01, 0.5, 0.99])}")
        print(f"   ends 1/10/50/90/99/100 %: {q(end, [0.01, 0.1, 0.5, 0.9, 0.99, 1.0])}; mean {end.mean():.2f}")
        print(f"   from queue-empty to the wave's end 10/50/90/99/100 %: {q(drain, [0.1, 0.5, 0.9, 0.99, 1.0])}; mean {drain.mean():.2f}; waves that never saw it empty: {(drain == 0).sum()}")
        seen = (snap[:, 0] >> np.uint64(32)).astype(np.float64) / 1e5
        not_gen = (snap[:, 0] & np.uint64(0xFFFFFFFF)).astype(np.float64)
        n_open = (snap[:, 1] >> np.uint64(48)).astype(np.float64); in_flight = ((snap[:, 1] >> np.uint64(32)) & np.uint64(0xFFFF)).astype(np.float64)
        not_staged = (snap[:, 1] & np.uint64(0xFFFFFFFF)).astype(np.float64)
        have = seen > 0
        order = np.argsort(end)
        groups = (("the earliest half", order[:len(order) // 2]), ("the latest tenth", order[-len(order) // 10:]), ("the latest 1 %", order[-max(1, len(order) // 100):]))
        print(f"   when the work counter passed the last unit (seen {q(start[have] + seen[have], [0.01, 0.5, 0.99])} ms; {int((~have).sum())} waves never saw it):")
        for name, idx in groups:
            idx = idx[have[idx]]
            print(f"     {name:18s}: {n_open[idx].mean():.2f} jobs open, {not_gen[idx].mean():7.1f} items not yet generated (max {not_gen[idx].max():.0f}), {in_flight[idx].mean():6.1f} paths in the table, "
                  f"{not_staged[idx].mean():7.1f} generated paths unfinished; end {end[idx].mean():.2f} ms")
        if have.sum() > 10:
            print(f"   correlation of a wave's end with: items not generated {np.corrcoef(end[have], not_gen[have])[0, 1]:.2f}, paths unfinished {np.corrcoef(end[have], not_staged[have])[0, 1]:.2f}, "
                  f"both {np.corrcoef(end[have], (not_gen + not_staged)[have])[0, 1]:.2f}")
        late = np.argsort(end)[-8:]
        print("   the 8 latest waves: end", " ".join(f"{end[i]:.2f}" for i in late), "| saw the queue empty at", " ".join(f"{start[i] + dry[i]:.2f}" for i in late))

#!/usr/bin/env python3
"""pmc_walk_table.py DIR — per kernel (every k_walk_probe variant, the path tracer's timed and counting instantiations): the counters tools/pmc_walk.sh collected, averaged
per dispatch, and the ratios that say where a wave's cycles go and how the caches fare at each occupancy."""
import glob, os, re, sqlite3, sys
root = sys.argv[1]
per = {}          # kernel -> counter -> [sum, dispatches]
for f in sorted(glob.glob(os.path.join(root, "**", "*.db"), recursive=True)):
    try:
        c = sqlite3.connect(f)
        cols = [r[1] for r in c.execute("pragma table_info(counters_collection)")]
        ki = [i for i, x in enumerate(cols) if "kernel" in x.lower() and "name" in x.lower()][0]
        ci, vi = cols.index("counter_name"), cols.index("value")
        di = cols.index("dispatch_id") if "dispatch_id" in cols else None
        seen = {}
        for r in c.execute("select * from counters_collection"):
            k = str(r[ki])
            if "k_walk_probe" not in k and "k_pathtrace" not in k and "k_stream" not in k:
                continue
            k = re.sub(r"^void ", "", re.sub(r"\(.*", "", k))
            e = per.setdefault(k, {}).setdefault(r[ci], [0.0, set()])
            e[0] += r[vi]
            e[1].add(r[di] if di is not None else len(e[1]))
    except Exception as e:
        print(f"({os.path.basename(f)}: {e})")
times = {}
for log in glob.glob(os.path.join(root, "*.log")):
    for m in re.finditer(r"(k_walk_probe<[^>]*>)[^\n]*?([0-9.]+) ms =\s+([0-9]+) Mray/s", open(log).read()):
        times.setdefault(m.group(1).replace(",", ", "), []).append(float(m.group(3)))
for k in sorted(per):
    c = {n: v[0] / max(len(v[1]), 1) for n, v in per[k].items()}
    g = lambda n: c.get(n, float("nan"))
    wc = g("SQ_WAVE_CYCLES")
    print(f"== {k}   ({max(len(v[1]) for v in per[k].values())} dispatches averaged)")
    print(f"   wave cycles {wc:.4g}: parked on s_waitcnt {g('SQ_WAIT_ANY') / wc:.3f}, issue-stalled {g('SQ_WAIT_INST_ANY') / wc:.3f}, issuing {g('SQ_ACTIVE_INST_ANY') / wc:.3f}"
          f" (VALU {g('SQ_ACTIVE_INST_VALU') / wc:.3f}, VMEM {g('SQ_ACTIVE_INST_VMEM') / wc:.3f}, LDS {g('SQ_ACTIVE_INST_LDS') / wc:.3f})")
    print(f"   instructions: VALU {g('SQ_INSTS_VALU'):.4g}, SALU {g('SQ_INSTS_SALU'):.4g}, VMEM rd {g('SQ_INSTS_VMEM_RD'):.4g}, wr {g('SQ_INSTS_VMEM_WR'):.4g}, LDS {g('SQ_INSTS_LDS'):.4g}; VMEM in flight per wave-cycle {g('SQ_INST_LEVEL_VMEM') / wc:.3f}")
    print(f"   L1: {g('TCP_TOTAL_CACHE_ACCESSES_sum'):.4g} line accesses, {g('TCP_TCC_READ_REQ_sum'):.4g} read requests to L2 -> hit rate by lines {1 - g('TCP_TCC_READ_REQ_sum') / g('TCP_TOTAL_CACHE_ACCESSES_sum'):.3f};"
          f" L1 -> L2 read latency {g('TCP_TCC_READ_REQ_LATENCY_sum') / g('TCP_TCC_READ_REQ_sum'):.0f} cycles; clocked cycles (GATE_EN1, all L1s) {g('TCP_GATE_EN1_sum'):.4g}, miss-pending stall {g('TCP_PENDING_STALL_CYCLES_sum'):.4g}, tag-conflict stall {g('TCP_READ_TAGCONFLICT_STALL_CYCLES_sum'):.4g}")
    print(f"   L2: {g('TCC_REQ_sum'):.4g} requests, hit rate {g('TCC_HIT_sum') / (g('TCC_HIT_sum') + g('TCC_MISS_sum')):.3f}; fabric reads {g('TCC_EA0_RDREQ_sum'):.4g}, average latency {g('TCC_EA0_RDREQ_LEVEL_sum') / g('TCC_EA0_RDREQ_sum'):.0f} cycles")

#!/usr/bin/env python3
"""pmc_sweep_table.py DIR — the counters tools/pmc_sweep.sh collected (rocprofv3 rocpd databases), one line per option set: kernel time, VALU
instructions, VALU pipe busy, lane utilisation, L1 -> L2 read requests, L2 hit rate, L2 misses; each also relative to the first set."""
import glob, os, re, sqlite3, sys
root = sys.argv[1]
def counters(db):
    c = sqlite3.connect(db)
    cols = [r[1] for r in c.execute("pragma table_info(counters_collection)")]
    ki = [i for i, x in enumerate(cols) if "kernel" in x.lower() and "name" in x.lower()][0]
    ci, vi = cols.index("counter_name"), cols.index("value")
    out = {}
    for r in c.execute("select * from counters_collection"):
        if "pathtrace" in str(r[ki]):
            out[r[ci]] = out.get(r[ci], 0.0) + r[vi]
    return out
rows = []
for d in sorted(glob.glob(os.path.join(root, "*")), key=os.path.getmtime):
    if not os.path.isdir(d):
        continue
    c = {}
    for f in glob.glob(os.path.join(d, "**", "*.db"), recursive=True):
        c.update(counters(f))
    ms = None
    for log in ("sq.log", "tcc.log"):
        m = re.findall(r"([0-9.]+) ms ", open(os.path.join(d, log)).read()) if os.path.exists(os.path.join(d, log)) else []
        if m: ms = float(m[-1])
    if c:
        rows.append((os.path.basename(d), ms, c))
if not rows:
    sys.exit("no counter databases under " + root)
base = rows[0]
print(f"{'set':12s} {'ms':>8s} {'VALU insts':>11s} {'pipe busy':>9s} {'lane util':>9s} {'L1->L2 req':>11s} {'L2 hit':>7s} {'L2 miss':>10s}   relative to '{base[0]}': ms / VALU / L2 miss")
for name, ms, c in rows:
    g = lambda k: c.get(k, float("nan"))
    busy = g("SQ_ACTIVE_INST_VALU") / (g("SQ_WAVE_CYCLES") / 4)
    util = g("SQ_THREAD_CYCLES_VALU") / (64 * g("SQ_INSTS_VALU"))
    hit = g("TCC_HIT_sum") / (g("TCC_HIT_sum") + g("TCC_MISS_sum"))
    bc = base[2]
    rel = f"{(ms or 0) / (base[1] or 1):.3f} / {g('SQ_INSTS_VALU') / bc.get('SQ_INSTS_VALU', float('nan')):.3f} / {g('TCC_MISS_sum') / bc.get('TCC_MISS_sum', float('nan')):.3f}"
    print(f"{name:12s} {ms or float('nan'):8.2f} {g('SQ_INSTS_VALU'):11.4g} {busy:9.3f} {util:9.3f} {g('TCP_TCC_READ_REQ_sum'):11.4g} {hit:7.3f} {g('TCC_MISS_sum'):10.4g}   {rel}")

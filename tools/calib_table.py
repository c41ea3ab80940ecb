#!/usr/bin/env python3
"""calib_table.py DIR — the table of tools/calib.sh (round 5): per micro-kernel of tools/ubench_calib.hip the KNOWN bytes / vector wave-instructions
beside what rocprofv3's counters report for the same launch, and the factors that follow:
  read_factor  = known bytes / (FETCH_SIZE x 1024)      (MI355X_MICROARCH.md: 2 for wide coalesced streaming reads)
  write_factor = known bytes / (WRITE_SIZE x 1024)
  pipe_busy    = SQ_ACTIVE_INST_VALU / (SQ_WAVE_CYCLES / 4)   at KNOWN saturation of the vector pipe (what bench.py's `valu.pipe_busy` reads when nothing else limits)
Prints text; --json FILE also writes the factors (tools/parse_prof.py and bench.py read profiles/calibration.json)."""
import json, os, sqlite3, sys
d = sys.argv[1]
plain = {}
for l in open(os.path.join(d, "plain.log")):
    if l.startswith("{"):
        j = json.loads(l); plain[j["kernel"]] = j
def counters(db):
    if not os.path.exists(db): return {}
    cur = sqlite3.connect(db).cursor()
    out = {}
    # the profiled run launches every kernel twice (warm + timed): average per dispatch
    for k, c, n, v in cur.execute("select kernel_name, counter_name, count(*), avg(value) from counters_collection group by kernel_name, counter_name"):
        name = k.split("(")[0]
        out.setdefault(name, {})[c] = v
    return out
groups = {g: counters(os.path.join(d, g + "_results.db")) for g in ("fetch", "write", "valu", "ea")}
res = {}
print(f"# {d}: known traffic / instruction counts of tools/ubench_calib.hip against rocprofv3 --pmc (one group per run)")
print(f"{'kernel':12s} {'ms':>8s} {'known GB':>9s} {'GB/s':>8s} {'FETCH_SIZE GB':>14s} {'read factor':>11s} {'WRITE_SIZE GB':>14s} {'write factor':>12s}")
for k, p in plain.items():
    if not p["known_bytes"]: continue
    f = groups["fetch"].get(k, {}).get("FETCH_SIZE"); w = groups["write"].get(k, {}).get("WRITE_SIZE")
    fb = f * 1024 if f is not None else None; wb = w * 1024 if w is not None else None
    reads = k != "k_write128"
    rf = p["known_bytes"] / fb if (fb and reads) else None
    wf = p["known_bytes"] / wb if (wb and not reads) else None
    res[k] = {"ms": p["ms"], "known_bytes": p["known_bytes"], "GBps": p["GBps"], "fetch_size_bytes": fb, "write_size_bytes": wb, "read_factor": rf, "write_factor": wf,
              "ea": groups["ea"].get(k)}
    print(f"{k:12s} {p['ms']:8.3f} {p['known_bytes'] / 1e9:9.3f} {p['GBps']:8.1f} {(fb or 0) / 1e9:14.3f} {rf if rf else float('nan'):11.3f} {(wb or 0) / 1e9:14.3f} {wf if wf else float('nan'):12.3f}")
    if groups["ea"].get(k): print("             L2 -> fabric requests:", {a: round(b) for a, b in groups["ea"][k].items()})
print()
print(f"{'kernel':12s} {'ms':>8s} {'known wave-insts':>17s} {'per SIMD-cycle':>14s} {'SQ_INSTS_VALU':>14s} {'ACTIVE_INST_VALU':>17s} {'WAVE_CYCLES':>13s} {'pipe_busy':>9s} {'cycles/inst':>11s}")
for k, p in plain.items():
    if not p["known_valu_wave_instructions"]: continue
    c = groups["valu"].get(k, {})
    busy = c["SQ_ACTIVE_INST_VALU"] / (c["SQ_WAVE_CYCLES"] / 4.0) if c.get("SQ_WAVE_CYCLES") else None
    cpi = c["SQ_ACTIVE_INST_VALU"] / c["SQ_INSTS_VALU"] if c.get("SQ_INSTS_VALU") else None      # busy cycles the SQ books per vector instruction
    res[k] = {"ms": p["ms"], "known_valu_wave_instructions": p["known_valu_wave_instructions"], "per_simd_cycle": p["wave_instructions_per_simd_cycle_at_2400MHz"], "counters": c,
              "pipe_busy_at_saturation": busy, "active_cycles_per_instruction": cpi}
    print(f"{k:12s} {p['ms']:8.3f} {p['known_valu_wave_instructions']:17.0f} {p['wave_instructions_per_simd_cycle_at_2400MHz']:14.4f} {c.get('SQ_INSTS_VALU', 0):14.0f} {c.get('SQ_ACTIVE_INST_VALU', 0):17.0f} {c.get('SQ_WAVE_CYCLES', 0):13.0f} "
          f"{busy if busy else float('nan'):9.3f} {cpi if cpi else float('nan'):11.3f}")
if "--json" in sys.argv:
    json.dump(res, open(sys.argv[sys.argv.index("--json") + 1], "w"), indent=1)

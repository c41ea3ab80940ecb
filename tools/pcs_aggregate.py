#!/usr/bin/env python3
"""pcs_aggregate.py CSV OUT.json — dev: histogram of a rocprofv3 pc-sampling CSV by (instruction comment = source line) and by instruction,
plus, when present, the stochastic columns (issued / stall reason / instruction type)."""
import collections, csv, json, sys
src, dst = sys.argv[1], sys.argv[2]
csv.field_size_limit(1 << 30)
n = 0
by_line = collections.Counter(); by_inst = collections.Counter(); extra = collections.defaultdict(collections.Counter)
lanes = collections.Counter()
with open(src, newline="") as f:
    rd = csv.DictReader(f)
    cols = rd.fieldnames
    for row in rd:
        n += 1
        c = row.get("Instruction_Comment", "") or ""
        i = row.get("Instruction", "") or ""
        by_line[c] += 1
        by_inst[(c, i)] += 1
        em = row.get("Exec_Mask")
        if em:
            try:
                lanes[c] += bin(int(em)).count("1")
            except ValueError:
                pass
        for k in cols:
            if k not in ("Sample_Timestamp", "Exec_Mask", "Dispatch_Id", "Instruction", "Instruction_Comment", "Correlation_Id", "Timestamp", "Wave_Id", "Chiplet", "Hw_Id"):
                v = row.get(k)
                if v is not None and len(v) < 40:
                    extra[k][v] += 1
out = {"columns": cols, "samples": n,
       "by_line": [[k, v, round(lanes[k] / max(v, 1), 1)] for k, v in by_line.most_common(400)],
       "by_inst": [[k[0], k[1], v] for k, v in by_inst.most_common(1500)],
       "extra": {k: dict(v.most_common(40)) for k, v in extra.items() if len(v) <= 4000}}
json.dump(out, open(dst, "w"), indent=0)
print("aggregated", n, "samples ->", dst, "columns:", cols)

#!/usr/bin/env python3
"""stream_trace.py DB — per-launch durations of the streaming form's kernels from a rocprofv3 --kernel-trace database, in launch order (W walk, S shade, F fold; microseconds)."""
import sqlite3, sys
db = sqlite3.connect(sys.argv[1])
tabs = {r[0].split('_0000')[0]: r[0] for r in db.execute("select name from sqlite_master where type='table'")}
kd, ks = tabs['rocpd_kernel_dispatch'], tabs['rocpd_info_kernel_symbol']
rows = db.execute(f"select s.kernel_name, d.start, d.end from {kd} d join {ks} s on d.kernel_id=s.id order by d.start").fetchall()
tot = {}
line = []
for n, s, e in rows:
    k = 'W' if 'stream_walk' in n else 'S' if 'stream_shade' in n else 'F' if 'stream_fold' in n else 'I' if 'stream_init' in n else 'R' if 'pathtrace_roll' in n else None
    if k is None: continue
    tot.setdefault(k, [0, 0.0]); tot[k][0] += 1; tot[k][1] += (e - s) / 1e6
    line.append(f"{k}{(e - s) / 1e3:.0f}")
print(' '.join(line))
for k, (n, ms) in sorted(tot.items()): print(f"{k}: {n} launches, {ms:.2f} ms")
st = [r for r in rows if 'k_stream' in r[0]]
if st: print(f"stream span {(st[-1][2] - st[0][1]) / 1e6:.2f} ms, busy {sum((e - s) for _, s, e in st) / 1e6:.2f} ms")

#!/usr/bin/env python3
"""probe_dropin.py — dev probe: the drop-in program's phase times (bench.py: dropin_timing), a few runs in a row."""
import json, os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import bench
wl = bench.WORKLOADS[sys.argv[1] if len(sys.argv) > 1 else "cfg2"]
for i in range(int(os.environ.get("RUNS", "3"))):
    d = bench.dropin_timing(wl)
    print(json.dumps({k: v for k, v in d.items() if k not in ("note", "program")}), flush=True)

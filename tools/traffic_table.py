#!/usr/bin/env python3
"""traffic_table.py DIR TAG — what tools/traffic_workloads.sh collected -> profiles/hbm_traffic_<workload>.json: bytes per dispatch = FETCH_SIZE x 1024 x the read
factor calibrated on divergent 64-byte gathers (profiles/calibration.json; the guide's x2 holds for 128-byte requests only: kept as the upper bound) + WRITE_SIZE x 1024,
with the sample count and ray count of the dispatch and the md5 of the device code it ran on."""
import glob, json, os, re, sqlite3, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from bench import kernel_source_md5
root, tag = sys.argv[1], sys.argv[2]


def counter(db, name):
    c = sqlite3.connect(db)
    cols = [r[1] for r in c.execute("pragma table_info(counters_collection)")]
    ki = [i for i, x in enumerate(cols) if "kernel" in x.lower() and "name" in x.lower()][0]
    ci, vi = cols.index("counter_name"), cols.index("value")
    return sum(r[vi] for r in c.execute("select * from counters_collection") if "pathtrace" in str(r[ki]) and r[ci] == name)


for d in sorted(glob.glob(os.path.join(root, "*"))):
    if not os.path.isdir(d):
        continue
    key = os.path.basename(d)
    try:
        rd = counter(glob.glob(os.path.join(d, "**", "fetch_results.db"), recursive=True)[0], "FETCH_SIZE") * 1024.0
        wr = counter(glob.glob(os.path.join(d, "**", "write_results.db"), recursive=True)[0], "WRITE_SIZE") * 1024.0
    except Exception as e:
        print(key, "no counters:", e)
        continue
    log = open(os.path.join(d, "write.log")).read()
    ms = [float(x) for x in re.findall(r"([0-9.]+) ms ", log)]
    rays = [int(x) for x in re.findall(r" ms ([0-9]+) rays", log)]
    from bench import calibrated_traffic
    out = {"workload": key, "spp": int(open(os.path.join(d, "spp")).read()), "rays": rays[-1] if rays else None, "fetch_bytes_raw": rd, "write_bytes": wr,
           "kernel_ms_under_the_counter": ms[-1] if ms else None, "tag": tag, "source_md5": kernel_source_md5(),
           "method": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate runs of one dispatch (tools/traffic_workloads.sh); bytes = counter * 1024. Round 5: the read "
                     "side is NOT doubled any more — profiles/calibration.json (tools/ubench_calib.hip) measured FETCH_SIZE x 1024 = 1.05 x the bytes of divergent 64-byte gathers "
                     "(this kernel's child pairs, texels, record quarters) and 0.53 x the bytes of 128-byte requests (wide streaming reads, the guide's x2 case): hbm_bytes_per_launch "
                     "uses the 64-byte factor, hbm_bytes_upper the 128-byte one. Infinity Cache hits are counted: this is L2 <-> fabric traffic"}
    out["hbm_bytes_per_launch"], out["hbm_bytes_upper"] = calibrated_traffic(out)
    json.dump(out, open(os.path.join(REPO, "profiles", f"hbm_traffic_{key}.json"), "w"), indent=1)
    print(key, f"{out['hbm_bytes_per_launch'] / 1e9:.1f} GB per dispatch", f"({(out['hbm_bytes_per_launch'] / 1e9) / (ms[-1] / 1e3):.0f} GB/s)" if ms else "")

#!/usr/bin/env python3
"""parse_prof.py — turn the rocprofv3 (rocpd SQLite) outputs under gpurun_out/prof/ into the text / JSON
summaries committed under profiles/ (the numbers bench.py's `roofline.traffic` and DESIGN.md quote).

    python tools/parse_prof.py <round-tag>       e.g. r01

HBM bytes follow MI355X_MICROARCH.md §HBM: FETCH_SIZE / WRITE_SIZE are reported in KiB-units of 1 KB... the
tool prints the raw counter value; bytes = value * 1024 for the *_SIZE derived counters, and the read side is
DOUBLED on gfx950 (FETCH_SIZE tallies 128-B requests at 64 B for wide coalesced reads; for this kernel's
16-64 B scattered reads that factor is an upper bound, so both the raw and the corrected figure are kept).
"""
import json
import os
import sqlite3
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PROF = os.path.join(REPO, "gpurun_out", "prof" + os.environ.get("PROF_SUFFIX", ""))
OUT = os.path.join(REPO, "profiles")


def kernel_stats(db):
    cur = sqlite3.connect(db).cursor()
    rows = cur.execute("select name, count(*), sum(duration), avg(duration), min(duration), max(duration), max(vgpr_count), "
                       "max(accum_vgpr_count), max(sgpr_count), max(lds_size), max(scratch_size), max(grid_x), max(workgroup_x) "
                       "from kernels group by name order by sum(duration) desc").fetchall()
    total = sum(r[2] for r in rows) or 1
    return [{"kernel": r[0], "calls": r[1], "total_ms": r[2] / 1e6, "avg_ms": r[3] / 1e6, "min_ms": r[4] / 1e6, "max_ms": r[5] / 1e6,
             "pct": 100.0 * r[2] / total, "vgpr": r[6], "agpr": r[7], "sgpr": r[8], "lds": r[9], "scratch": r[10], "grid": r[11], "wg": r[12]}
            for r in rows]


def counters(db, kernel_like="k_pathtrace"):
    """Per-dispatch averages over the TIMED launches (counter level 1: bench.py's counting pass runs the level-2 instantiation)."""
    cur = sqlite3.connect(db).cursor()
    q = ("select counter_name, count(*), sum(value), avg(value) from counters_collection "
         "where (kernel_name like ? or kernel_name like ?) group by counter_name")
    rows = cur.execute(q, (f"%{kernel_like}%<1,%", f"%{kernel_like}%ILi1E%")).fetchall()       # k_pathtrace<1, ...> and k_pathtrace_roll<1, ...>
    if not rows:
        rows = cur.execute(q, (f"%{kernel_like}%", f"%{kernel_like}%")).fetchall()
    return {r[0]: {"dispatches": r[1], "sum": r[2], "per_dispatch": r[3]} for r in rows}


def main():
    tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
    os.makedirs(OUT, exist_ok=True)
    summary = {"tag": tag}
    tdb = os.path.join(PROF, "trace", f"{tag}_results.db")
    lines = []
    if os.path.exists(tdb):
        ks = kernel_stats(tdb)
        summary["kernel_trace"] = ks
        lines.append(f"# rocprofv3 --kernel-trace --stats -- python bench.py --steps 3 --warmup 1 --no-cpu   ({tag})")
        lines.append(f"{'kernel':70s} {'calls':>5s} {'total_ms':>10s} {'avg_ms':>10s} {'min_ms':>10s} {'max_ms':>10s} {'%':>6s} {'vgpr':>5s} {'agpr':>5s} {'sgpr':>5s} {'lds':>6s} {'scratch':>7s} {'grid':>8s} {'wg':>4s}")
        for k in ks:
            lines.append(f"{k['kernel'][:70]:70s} {k['calls']:5d} {k['total_ms']:10.3f} {k['avg_ms']:10.3f} {k['min_ms']:10.3f} {k['max_ms']:10.3f} {k['pct']:6.2f} "
                         f"{k['vgpr']:5d} {k['agpr']:5d} {k['sgpr']:5d} {k['lds']:6d} {k['scratch']:7d} {k['grid']:8d} {k['wg']:4d}")
    pmc = {}
    for sub in sorted(os.listdir(PROF)) if os.path.isdir(PROF) else []:
        db = os.path.join(PROF, sub, f"{tag}_results.db")
        if sub.startswith("pmc") and os.path.exists(db):
            for name, v in counters(db).items():
                pmc[name] = v
    if pmc:
        summary["pmc_k_pathtrace"] = pmc
        lines.append("")
        lines.append(f"# rocprofv3 --pmc <one group per run> -- python bench.py --steps 2 --warmup 0 --no-cpu   ({tag}); k_pathtrace dispatches only")
        for name in sorted(pmc):
            lines.append(f"{name:28s} dispatches {pmc[name]['dispatches']:3d}   per dispatch {pmc[name]['per_dispatch']:.6g}")
        if "FETCH_SIZE" in pmc or "WRITE_SIZE" in pmc:
            rd = pmc.get("FETCH_SIZE", {}).get("per_dispatch", 0.0) * 1024
            wr = pmc.get("WRITE_SIZE", {}).get("per_dispatch", 0.0) * 1024
            sys.path.insert(0, REPO)
            from bench import kernel_source_md5
            valu = None
            need = ("SQ_ACTIVE_INST_VALU", "SQ_WAVE_CYCLES", "SQ_THREAD_CYCLES_VALU", "SQ_INSTS_VALU")
            if all(k in pmc for k in need):
                g = lambda k: pmc[k]["per_dispatch"]
                valu = {"bound": "see bench.py: measured_profile (the formula's reading is held against profiles/calibration.json there)", "pipe_busy": round(g("SQ_ACTIVE_INST_VALU") / (g("SQ_WAVE_CYCLES") / 4.0), 4),
                        "lane_utilisation": round(g("SQ_THREAD_CYCLES_VALU") / (64.0 * g("SQ_INSTS_VALU")), 4),
                        "valu_wave_instructions_per_launch": g("SQ_INSTS_VALU"),
                        "formulas": "pipe_busy = SQ_ACTIVE_INST_VALU / (SQ_WAVE_CYCLES / 4); lane_utilisation = SQ_THREAD_CYCLES_VALU / (64 * SQ_INSTS_VALU)",
                        "source": f"profiles/{tag}_rocprof_summary.json"}
            from bench import calibrated_traffic
            hbm = {"fetch_bytes_raw": rd, "write_bytes": wr,
                   "tag": tag, "workload": os.environ.get("PROFILE_WORKLOAD", "cfg2"),
                   "source_md5": kernel_source_md5(), "valu": valu,
                   "method": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate runs; bytes = counter * 1024. Round 5: the read side times the factor measured on divergent "
                             "64-byte gathers (profiles/calibration.json, tools/ubench_calib.hip: FETCH_SIZE counts a request at 64 bytes whatever its size — 1.05 x the bytes of "
                             "64-byte gathers, 0.5 x those of 128-byte requests, the guide's x2 case); hbm_bytes_upper = as if every request had been 128 bytes"}
            hbm["hbm_bytes_per_launch"], hbm["hbm_bytes_upper"] = calibrated_traffic(hbm)
            summary["hbm"] = hbm
            lines.append("")
            lines.append(f"L2 <-> fabric read  per launch: FETCH_SIZE x 1024 = {rd/1e9:.3f} GB; calibrated (64-byte gathers) {(hbm['hbm_bytes_per_launch'] - wr)/1e9:.3f} GB; upper bound (every request 128 bytes) {(hbm['hbm_bytes_upper'] - wr)/1e9:.3f} GB")
            lines.append(f"L2 <-> fabric write per launch: {wr/1e9:.3f} GB")
            with open(os.path.join(OUT, "hbm_traffic" + os.environ.get("PROF_SUFFIX", "") + ".json"), "w") as f:
                json.dump(hbm, f, indent=1)
    with open(os.path.join(OUT, f"{tag}_rocprof_summary.txt"), "w") as f:
        f.write("\n".join(lines) + "\n")
    with open(os.path.join(OUT, f"{tag}_rocprof_summary.json"), "w") as f:
        json.dump(summary, f, indent=1)
    print("\n".join(lines))


if __name__ == "__main__":
    main()

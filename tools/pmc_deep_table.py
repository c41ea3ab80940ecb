#!/usr/bin/env python3
"""pmc_deep_table.py DIR — the counters tools/pmc_deep.sh collected (rocprofv3 rocpd databases): per workload, every counter of the
path-tracing kernel and the ratios that say where a wave's time goes."""
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


for d in sorted(glob.glob(os.path.join(root, "*"))):
    if not os.path.isdir(d):
        continue
    c, ms = {}, []
    for f in sorted(glob.glob(os.path.join(d, "**", "*.db"), recursive=True)):
        try:
            c.update(counters(f))
        except Exception as e:        # a group that did not fit its block's slots leaves no table
            print(f"  ({os.path.basename(f)}: {e})")
    for log in glob.glob(os.path.join(d, "*.log")):
        ms += [float(x) for x in re.findall(r"([0-9.]+) ms ", open(log).read())]
    if not c:
        continue
    g = lambda k: c.get(k, float("nan"))
    print(f"== {os.path.basename(d)}: kernel {min(ms) if ms else float('nan'):.2f} .. {max(ms) if ms else float('nan'):.2f} ms under the counters")
    for k in sorted(c):
        print(f"   {k:42s} {c[k]:.6g}")
    wc = g("SQ_WAVE_CYCLES")
    print("   -- derived")
    print(f"   wave cycles: parked on s_waitcnt {g('SQ_WAIT_ANY') / wc:.3f}, issue-stalled {g('SQ_WAIT_INST_ANY') / wc:.3f}, issuing {g('SQ_ACTIVE_INST_ANY') / wc:.3f}"
          f" (VALU {g('SQ_ACTIVE_INST_VALU') / wc:.3f}, VMEM {g('SQ_ACTIVE_INST_VMEM') / wc:.3f}, LDS {g('SQ_ACTIVE_INST_LDS') / wc:.3f}, scalar {g('SQ_ACTIVE_INST_SCA') / wc:.3f})")
    print(f"   VALU pipe busy (ACTIVE_INST_VALU / (WAVE_CYCLES / 4 waves)): {g('SQ_ACTIVE_INST_VALU') / (wc / 4):.3f}")
    print(f"   per VMEM read instruction: {g('SQ_INSTS_VALU') / g('SQ_INSTS_VMEM_RD'):.1f} VALU, {g('SQ_INSTS_SALU') / g('SQ_INSTS_VMEM_RD'):.1f} SALU, {g('SQ_INSTS_LDS') / g('SQ_INSTS_VMEM_RD'):.2f} LDS;"
          f" VMEM write / read instructions {g('SQ_INSTS_VMEM_WR') / g('SQ_INSTS_VMEM_RD'):.3f}")
    print(f"   TA busy / GPU active cycles (per TA: / 256): {g('TA_TA_BUSY_sum') / g('GRBM_GUI_ACTIVE') / 256:.3f};"
          f" TA address side stalled by L1 {g('TA_ADDR_STALLED_BY_TC_CYCLES_sum') / g('TA_TA_BUSY_sum'):.3f} of busy, data side {g('TA_DATA_STALLED_BY_TC_CYCLES_sum') / g('TA_TA_BUSY_sum'):.3f}")
    print(f"   L1: {g('TCP_TOTAL_CACHE_ACCESSES_sum'):.4g} line accesses, {g('TCP_TCC_READ_REQ_sum'):.4g} read requests to L2 (L1 hit rate by lines {1 - g('TCP_TCC_READ_REQ_sum') / g('TCP_TOTAL_CACHE_ACCESSES_sum'):.3f}),"
          f" {g('TCP_TCC_WRITE_REQ_sum'):.4g} write requests; cycles per L1 with a miss pending {g('TCP_PENDING_STALL_CYCLES_sum') / 256:.4g}")
    print(f"   L1 -> L2 read latency (LATENCY / READ_REQ): {g('TCP_TCC_READ_REQ_LATENCY_sum') / g('TCP_TCC_READ_REQ_sum'):.0f} cycles; L1 access latency (TCP_TCP_LATENCY / accesses): {g('TCP_TCP_LATENCY_sum') / g('TCP_TOTAL_ACCESSES_sum'):.0f}")
    print(f"   L1 TLB: {g('TCP_UTCL1_REQUEST_sum'):.4g} requests, miss rate {g('TCP_UTCL1_TRANSLATION_MISS_sum') / g('TCP_UTCL1_REQUEST_sum'):.4f}")
    print(f"   L2: {g('TCC_REQ_sum'):.4g} requests, hit rate {g('TCC_HIT_sum') / (g('TCC_HIT_sum') + g('TCC_MISS_sum')):.3f}, {g('TCC_MISS_sum'):.4g} misses, tag stall / busy {g('TCC_TAG_STALL_sum') / g('TCC_BUSY_sum'):.3f}")
    print(f"   L2 -> fabric: {g('TCC_EA0_RDREQ_sum'):.4g} read requests, average latency {g('TCC_EA0_RDREQ_LEVEL_sum') / g('TCC_EA0_RDREQ_sum'):.0f} cycles, to DRAM {g('TCC_EA0_RDREQ_DRAM_sum') / g('TCC_EA0_RDREQ_sum'):.3f}; FETCH_SIZE {g('FETCH_SIZE'):.4g} KB, WRITE_SIZE {g('WRITE_SIZE'):.4g} KB")

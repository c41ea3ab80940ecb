#!/bin/bash
# pmc_deep.sh TAG [render_once options ...] — ONE GPU call: the counters DESIGN.md section 7(c) listed as never collected, for the default
# kernel on four workloads (cfg2 at 64 spp, statues at 4 spp, the 1 M soup at 8 spp and — round 5 — the 10 M soup at 8 spp, the one scene larger than the Infinity
# Cache: its blob is built on the box first). PMC_WORKLOADS="soup10m cfg2" / PMC_GROUPS="waits insts" restrict the passes. One rocprofv3 --pmc pass per counter group
# (tools/render_once.py, no torch: ~4 s per pass); a group that does not fit the block's slots fails alone (its log says so).
#   waits     where wave cycles go: parked on s_waitcnt, issue-stalled, issuing (SQ_WAIT_ANY + SQ_WAIT_INST_ANY + SQ_ACTIVE_INST_ANY = SQ_WAVE_CYCLES)
#   insts     instruction mix (VALU / SALU / SMEM / VMEM read, write / LDS) + average VMEM instructions in flight
#   (the TA / TD groups hang rocprofv3 on this image — 60 s timeouts in r03a — and are left out)
#   tcp_req   L1: accesses, tag look-ups, requests to L2 (read / write), stall cycles while a miss is pending
#   tcp_lat   L1: summed latency of its read requests to L2, of its own accesses, tag-conflict stalls
#   utcl1     L1 TLB: requests, hits, misses
#   tcc       L2: requests, hits, misses, reads
#   tcc_ea    L2 -> fabric: read requests, their summed time in flight (average latency = LEVEL / RDREQ), 32-B-granular DRAM reads, write requests
# Output: gpurun_out/pmc_deep_TAG/<workload>/<group>_results.db + logs; then, here: python tools/pmc_deep_table.py gpurun_out/pmc_deep_TAG
TAG=${1:-r03}; shift
OPTS="$*"
cd "$(dirname "$0")/.." || exit 1
R=$(pwd); OUT=$R/gpurun_out/pmc_deep_$TAG; mkdir -p "$OUT"
export TMPDIR=/tmp
if [ ! -f "$R/scenes/_built/soup_10m.blob" ] && { [ -z "$PMC_WORKLOADS" ] || [[ " $PMC_WORKLOADS " == *" soup10m "* ]]; }; then
	python "$R/tools/make_soup_blob.py" 10000000 "$R/scenes/_built/soup_10m.blob" > "$OUT/make_soup_10m.log" 2>&1 || echo "soup_10m.blob: build failed"
fi
cd /tmp || exit 1
while read -r wl scene w h spp b; do
	[ -z "$wl" ] && continue
	[ -n "$PMC_WORKLOADS" ] && [[ " $PMC_WORKLOADS " != *" $wl "* ]] && continue
	[ -f "$R/scenes/_built/$scene.blob" ] || continue
	while read -r grp ctrs; do
		[ -z "$grp" ] && continue
		[ -n "$PMC_GROUPS" ] && [[ " $PMC_GROUPS " != *" $grp "* ]] && continue
		d=$OUT/$wl; mkdir -p "$d"
		# shellcheck disable=SC2086
		timeout 60 rocprofv3 --pmc $ctrs -d "$d" -o "$grp" -- python "$R/tools/render_once.py" $scene $w $h $spp $b $OPTS > "$d/$grp.log" 2>&1
		echo "$wl $grp rc=$? $(grep ' ms ' "$d/$grp.log" | tail -1)"
	done <<'GROUPS'
waits SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS
insts SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INST_LEVEL_VMEM SQ_ACTIVE_INST_SCA
tcp_req TCP_TOTAL_CACHE_ACCESSES_sum TCP_TOTAL_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum
tcp_stall TCP_PENDING_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TCP_GATE_EN1_sum
tcp_lat TCP_TCC_READ_REQ_LATENCY_sum TCP_TCP_LATENCY_sum TCP_TA_TCP_STATE_READ_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum
utcl1 TCP_UTCL1_REQUEST_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_PERMISSION_MISS_sum
tcc TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum TCC_READ_sum
tcc2 TCC_WRITE_sum TCC_WRITEBACK_sum TCC_TAG_STALL_sum TCC_BUSY_sum
tcc_ea TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_LEVEL_sum TCC_EA0_RDREQ_DRAM_sum TCC_EA0_WRREQ_sum
fetch FETCH_SIZE
write WRITE_SIZE
GROUPS
done <<'WORKLOADS'
cfg2 cfg2_hdr 1280 720 64 8
statues cfg4_statues 3840 2160 4 30
soup1m soup_1m 2560 1440 8 8
soup10m soup_10m 2560 1440 8 8
WORKLOADS

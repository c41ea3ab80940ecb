#!/bin/bash
# pmc_walk.sh TAG [scene ...] — round 6: the counters of tools/pmc_deep.sh for the WALK-ONLY probe (tools/probe_walk.py, csrc/walk_probe.h) at each occupancy, and for the
# path tracer's own dispatches of the same rays beside them: one rocprofv3 --pmc pass per counter group over the whole probe script; tools/pmc_walk_table.py gives the
# per-kernel averages. PROBE_VARIANTS / PROBE_RAYS as for probe_walk.py.
TAG=${1:-r06}; shift
SCENES=${*:-soup_10m}
cd "$(dirname "$0")/.." || exit 1
R=$(pwd); OUT=$R/gpurun_out/pmc_walk_$TAG; mkdir -p "$OUT"
export TMPDIR=/tmp
export PROBE_VARIANTS=${PROBE_VARIANTS:-2.12.1.3,3.12.1.3,4.12.1.3,5.12.1.3,6.7.1.3,7.3.1.3,8.4.0.3}
export PROBE_RAYS=${PROBE_RAYS:-32e6}
cd /tmp || exit 1
while read -r grp ctrs; do
	[ -z "$grp" ] && continue
	[ -n "$PMC_GROUPS" ] && [[ " $PMC_GROUPS " != *" $grp "* ]] && continue
	# shellcheck disable=SC2086
	timeout 300 rocprofv3 --pmc $ctrs -d "$OUT" -o "$grp" -- python "$R/tools/probe_walk.py" $SCENES > "$OUT/$grp.log" 2>&1
	echo "$grp rc=$?"
done <<'GROUPS'
waits SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS
insts SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INST_LEVEL_VMEM SQ_WAVES
tcp_req TCP_TOTAL_CACHE_ACCESSES_sum TCP_TOTAL_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum
tcp_stall TCP_PENDING_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TCP_GATE_EN1_sum
tcp_lat TCP_TCC_READ_REQ_LATENCY_sum TCP_TCP_LATENCY_sum TCP_TA_TCP_STATE_READ_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum
tcc TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum TCC_READ_sum
tcc_ea TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_LEVEL_sum TCC_EA0_RDREQ_DRAM_sum TCC_EA0_WRREQ_sum
GROUPS
python "$R/tools/pmc_walk_table.py" "$OUT" > "$OUT/table.txt" 2>&1
cat "$OUT/table.txt"

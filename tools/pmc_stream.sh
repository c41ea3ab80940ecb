#!/bin/bash
# pmc_stream.sh TAG [workload:spp] — round 6: where the streaming form's kernels (k_stream_walk / k_stream_shade, csrc/pathtrace_stream.h) spend their wave cycles, beside the rolling
# kernel's dispatch of the same frame: one rocprofv3 --pmc pass per counter group over tools/ab_stream.py (both forms render the job once); tools/pmc_walk_table.py averages per kernel.
TAG=${1:-r06}; JOB=${2:-cfg4:8}
cd "$(dirname "$0")/.." || exit 1
R=$(pwd); OUT=$R/gpurun_out/pmc_stream_$TAG; mkdir -p "$OUT"
export TMPDIR=/tmp AB_REPS=1
cd /tmp || exit 1
while read -r grp ctrs; do
	[ -z "$grp" ] && continue
	# shellcheck disable=SC2086
	timeout 300 rocprofv3 --pmc $ctrs -d "$OUT" -o "$grp" -- python "$R/tools/ab_stream.py" $JOB > "$OUT/$grp.log" 2>&1
	echo "$grp rc=$?"
done <<'GROUPS'
waits SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS
insts SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INST_LEVEL_VMEM SQ_WAVES
tcp_req TCP_TOTAL_CACHE_ACCESSES_sum TCP_TOTAL_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum
tcp_lat TCP_TCC_READ_REQ_LATENCY_sum TCP_TCP_LATENCY_sum TCP_TA_TCP_STATE_READ_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum
tcc TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum TCC_READ_sum
tcc_ea TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_LEVEL_sum TCC_EA0_RDREQ_DRAM_sum TCC_EA0_WRREQ_sum
GROUPS
python "$R/tools/pmc_walk_table.py" "$OUT" > "$OUT/table.txt" 2>&1
cat "$OUT/table.txt"

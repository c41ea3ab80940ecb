#!/bin/bash
# calib.sh TAG — ONE GPU call (round 5): tools/ubench_calib.hip's kernels of KNOWN traffic and KNOWN vector-instruction counts, once plain and once per
# rocprofv3 counter group (one --pmc pass per group, nothing else traced). Output: gpurun_out/calib_TAG/{plain.log, <group>_results.db, <group>.log}; then, here:
#   python tools/calib_table.py gpurun_out/calib_TAG > profiles/TAG_calibration.txt
TAG=${1:-r05}
cd "$(dirname "$0")/.." || exit 1
R=$(pwd); OUT=$R/gpurun_out/calib_$TAG; mkdir -p "$OUT"
BIN=$R/c-ray_amd/_lib/ubench_calib
[ -x "$BIN" ] || /opt/rocm/bin/hipcc --offload-arch=gfx950 -O2 "$R/tools/ubench_calib.hip" -o "$BIN" || exit 1
export TMPDIR=/tmp
cd /tmp || exit 1
timeout 120 "$BIN" > "$OUT/plain.log" 2>&1; echo "plain rc=$?"
while read -r grp ctrs; do
	[ -z "$grp" ] && continue
	# shellcheck disable=SC2086
	timeout 120 rocprofv3 --pmc $ctrs -d "$OUT" -o "$grp" -- "$BIN" > "$OUT/$grp.log" 2>&1
	echo "$grp rc=$?"
done <<'GROUPS'
fetch FETCH_SIZE
write WRITE_SIZE
valu SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_BUSY_CYCLES SQ_THREAD_CYCLES_VALU SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_ANY
ea TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum
GROUPS

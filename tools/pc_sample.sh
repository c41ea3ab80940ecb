#!/bin/bash
# pc_sample.sh — dev (GPU box): rocprofv3 PC sampling of the path-tracing kernel (a -gline-tables-only build: c-ray_amd/_lib/variants/pcs_g.so),
# aggregated per source line / instruction by tools/pcs_aggregate.py into gpurun_out/pcs/*.json.
cd "$(dirname "$0")/.." || exit 1
REPO=$(pwd)
export TMPDIR=/tmp
OUT=$REPO/gpurun_out/pcs
mkdir -p "$OUT"
export CRH_LIB=$REPO/c-ray_amd/_lib/variants/pcs_g.so
cd /tmp || exit 1
run() {   # tag method unit interval scene...
	local tag=$1 method=$2 unit=$3 interval=$4; shift 4
	rm -rf "/tmp/pcs_$tag"
	timeout 170 rocprofv3 --pc-sampling-beta-enabled --pc-sampling-method "$method" --pc-sampling-unit "$unit" --pc-sampling-interval "$interval" \
		--kernel-trace --output-format csv -d "/tmp/pcs_$tag" -o "$tag" -- python "$REPO/tools/pcs_render.py" "$@" > "$OUT/$tag.log" 2>&1
	echo "$tag rc $?"; tail -3 "$OUT/$tag.log"
	find "/tmp/pcs_$tag" -type f | head -20
	for f in $(find "/tmp/pcs_$tag" -name '*pc_sampling*' -type f); do
		ls -la "$f"; head -5 "$f" > "$OUT/$tag.$(basename "$f").head"
		head -150000 "$f" | gzip > "$OUT/$tag.$(basename "$f").gz"
		python "$REPO/tools/pcs_aggregate.py" "$f" "$OUT/$tag.$(basename "$f").agg.json"
	done
}
run cfg2_st stochastic cycles ${PCS_INTERVAL:-4194304} cfg2_hdr 1280 720 256 8 3
if ! ls "$OUT"/cfg2_st.*agg.json > /dev/null 2>&1; then
	run cfg2_ht host_trap time ${PCS_US:-50} cfg2_hdr 1280 720 256 8 3
	run soup_ht host_trap time ${PCS_US:-50} soup_1m 2560 1440 16 8 3
else
	run soup_st stochastic cycles ${PCS_INTERVAL:-4194304} soup_1m 2560 1440 16 8 3
fi

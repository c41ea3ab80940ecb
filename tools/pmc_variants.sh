#!/bin/bash
# pmc_variants.sh TAG VARIANT... — ONE GPU call: for every library variant (c-ray_amd/_lib/variants/NAME.so, tools/build_variant.sh) two rocprofv3 --pmc
# passes of one dispatch on cfg2 (64 spp) and on the 1 M soup (8 spp): the SQ group and the cache group of tools/pmc_sweep.sh.
# Output: gpurun_out/pmc_variants_TAG_<workload>/<variant>/...; then, here: python tools/pmc_sweep_table.py gpurun_out/pmc_variants_TAG_<workload>
TAG=$1; shift
cd "$(dirname "$0")/.." || exit 1
R=$(pwd)
export TMPDIR=/tmp
cd /tmp || exit 1
SQ="SQ_WAVES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_INSTS_VMEM"
TCC="TCC_HIT_sum TCC_MISS_sum TCP_TCC_READ_REQ_sum GRBM_GUI_ACTIVE"
while read -r wl scene w h spp b; do
	for v in "$@"; do
		d=$R/gpurun_out/pmc_variants_${TAG}_$wl/$v; mkdir -p "$d"
		# shellcheck disable=SC2086
		CRH_LIB=$R/c-ray_amd/_lib/variants/$v.so timeout 60 rocprofv3 --pmc $SQ -d "$d" -o sq -- python "$R/tools/render_once.py" $scene $w $h $spp $b > "$d/sq.log" 2>&1
		# shellcheck disable=SC2086
		CRH_LIB=$R/c-ray_amd/_lib/variants/$v.so timeout 60 rocprofv3 --pmc $TCC -d "$d" -o tcc -- python "$R/tools/render_once.py" $scene $w $h $spp $b > "$d/tcc.log" 2>&1
		echo "$wl $v $(grep ' ms ' "$d/tcc.log" | tail -1)"
	done
done <<'WORKLOADS'
cfg2 cfg2_hdr 1280 720 64 8
soup1m soup_1m 2560 1440 8 8
WORKLOADS

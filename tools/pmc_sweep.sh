#!/bin/bash
# pmc_sweep.sh TAG [SCENE W H SPP BOUNCES] — ONE GPU call: for every option set in the list below, two rocprofv3 --pmc passes of one dispatch
# (tools/render_once.py, no torch: ~4 s per pass): the SQ group (VALU instructions, active cycles, lane cycles, wave cycles, VMEM) and the cache
# group (L1 -> L2 read requests, L2 hits / misses). What r02e learnt: the walk waits for cache misses, so settings are
# compared by misses and waiting, not by lane utilisation. Output: gpurun_out/pmc_sweep_TAG/<set>/{sq,tcc}_results.db + log; then, here:
#     python tools/pmc_sweep_table.py gpurun_out/pmc_sweep_TAG      -> one line per option set
TAG=${1:-r03}; shift
SCENE=${1:-cfg2_hdr}; W=${2:-1280}; H=${3:-720}; SPP=${4:-64}; B=${5:-8}
cd "$(dirname "$0")/.." || exit 1
R=$(pwd); OUT=$R/gpurun_out/pmc_sweep_$TAG; mkdir -p "$OUT"
export TMPDIR=/tmp
cd /tmp || exit 1
SQ="SQ_WAVES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_INSTS_VMEM"
TCC="TCC_HIT_sum TCC_MISS_sum TCP_TCC_READ_REQ_sum GRBM_GUI_ACTIVE"
i=0
while read -r name opts; do
	[ -z "$name" ] && continue
	d=$OUT/$name; mkdir -p "$d"
	# shellcheck disable=SC2086
	timeout 40 rocprofv3 --pmc $SQ -d "$d" -o sq -- python "$R/tools/render_once.py" $SCENE $W $H $SPP $B $opts > "$d/sq.log" 2>&1
	# shellcheck disable=SC2086
	timeout 40 rocprofv3 --pmc $TCC -d "$d" -o tcc -- python "$R/tools/render_once.py" $SCENE $W $H $SPP $B $opts > "$d/tcc.log" 2>&1
	grep " ms " "$d/tcc.log" | tail -1
done <<'LIST'
default
fill128 fill_to=128
fill96 fill_to=96
fill192 fill_to=192
unit1024 unit_items=1024
unit4096 unit_items=4096
swap8 swap_min=8
swap32 swap_min=32
blocks3 blocks_per_cu=3
wg kernel=1
LIST

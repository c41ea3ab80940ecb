#!/bin/bash
# profile_round.sh TAG — the rocprofv3 passes behind profiles/TAG_rocprof_summary.{txt,json} (run on the GPU box):
#   one --kernel-trace --stats pass of bench.py, then one --pmc pass per counter group (PMC is never combined with tracing).
# Afterwards, here:  python tools/parse_prof.py TAG
TAG=${1:-r01}
WORKLOAD=${2:-cfg2}          # bench.py --workload; SAMPLES (env) = dev override for the long configs
EXTRA="--workload $WORKLOAD --no-cpu --no-dropin --no-others ${SAMPLES:+--samples $SAMPLES}"
cd "$(dirname "$0")/.." || exit 1
REPO=$(pwd)
export TMPDIR=/tmp
OUT=$REPO/gpurun_out/prof${PROF_SUFFIX:-}
mkdir -p "$OUT"
cd /tmp || exit 1
rocprofv3 --kernel-trace --stats -d "$OUT/trace" -o "$TAG" -- python "$REPO/bench.py" --steps 3 --warmup 1 $EXTRA > "$OUT/trace_$TAG.log" 2>&1
echo "trace rc $?"; tail -1 "$OUT/trace_$TAG.log"
i=0
for group in "FETCH_SIZE" "WRITE_SIZE" \
             "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU" \
             "SQ_INSTS_SALU SQ_INSTS_VMEM SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY" \
             "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_INSTS_SMEM" \
             "TCC_HIT_sum TCC_MISS_sum TCP_TCC_READ_REQ_sum GRBM_GUI_ACTIVE"; do
	i=$((i + 1))
	if [ -n "$TRAFFIC_ONLY" ] && [ $i -gt 2 ]; then break; fi      # TRAFFIC_ONLY=1: FETCH_SIZE / WRITE_SIZE passes only
	# shellcheck disable=SC2086
	rocprofv3 --pmc $group -d "$OUT/pmc$i" -o "$TAG" -- python "$REPO/bench.py" --steps 2 --warmup 0 $EXTRA > "$OUT/pmc${i}_$TAG.log" 2>&1
	echo "pmc$i ($group) rc $?"
done

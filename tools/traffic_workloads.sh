#!/bin/bash
# traffic_workloads.sh TAG — ONE GPU call: L2 <-> fabric traffic (rocprofv3 --pmc FETCH_SIZE, then WRITE_SIZE: separate passes, no tracing) of one dispatch
# of each workload bench.py's `other_workloads` measures, at the sample counts of ITS timed dispatches (bench.py: OTHER_WORKLOADS — round 6: BASELINE's own, so that the line
# quotes the traffic of the dispatch it times instead of carrying bytes per ray over from a 4-32 spp run). The 10 M soup is built here first
# (tools/make_soup_blob.py). Output: gpurun_out/traffic_TAG/<workload>/{fetch,write}_results.db; then, here:
#     python tools/traffic_table.py gpurun_out/traffic_TAG TAG      -> profiles/hbm_traffic_<workload>.json (what bench.py quotes, gated by the device code's md5)
TAG=${1:-r03}
cd "$(dirname "$0")/.." || exit 1
R=$(pwd); OUT=$R/gpurun_out/traffic_$TAG; mkdir -p "$OUT"
export TMPDIR=/tmp
[ -n "$SKIP_SOUP10M" ] || [ -f /tmp/crh_soup_10m.blob ] || python tools/make_soup_blob.py 10000000 /tmp/crh_soup_10m.blob > "$OUT/soup10m_build.json" 2>&1      # (SKIP_SOUP10M=1: a short call)
cd /tmp || exit 1
while read -r key scene w h spp b; do
	[ -z "$key" ] && continue
	[ -n "$SKIP_SOUP10M" ] && [ "$key" = soup10m ] && continue
	d=$OUT/$key; mkdir -p "$d"
	echo "$spp" > "$d/spp"
	timeout 400 rocprofv3 --pmc FETCH_SIZE -d "$d" -o fetch -- python "$R/tools/render_once.py" $scene $w $h $spp $b > "$d/fetch.log" 2>&1
	timeout 400 rocprofv3 --pmc WRITE_SIZE -d "$d" -o write -- python "$R/tools/render_once.py" $scene $w $h $spp $b > "$d/write.log" 2>&1
	echo "$key $(grep ' ms ' "$d/write.log" | tail -1)"
done <<'WORKLOADS'
cfg3 cfg3_venus 1920 1080 1024 32
cfg4 cfg4_statues 3840 2160 2048 30
soup soup_1m 2560 1440 512 8
soup10m /tmp/crh_soup_10m.blob 2560 1440 512 8
WORKLOADS

#!/bin/bash
# evidence_round.sh TAG — everything profiles/ quotes for a round, in one GPU call (about 5 minutes of box time):
#   GPU test suite, the bench line (CPU baseline + drop-in timing), rocprofv3 trace + PMC groups on cfg2, FETCH_SIZE / WRITE_SIZE of every workload bench.py's other_workloads
#   measures (tools/traffic_workloads.sh), the wait / L1 / L2 / fabric counters (tools/pmc_deep.sh), and BASELINE.json configs 2-5 with counters, roofline figures and parity against the oracle.
# Afterwards, here: python tools/parse_prof.py TAG; python tools/traffic_table.py gpurun_out/traffic_TAG TAG; python tools/pmc_deep_table.py gpurun_out/pmc_deep_TAG
TAG=${1:-r03}
cd "$(dirname "$0")/.." || exit 1
O=gpurun_out/evidence_$TAG
mkdir -p $O
timeout 900 python -m pytest tests -m gpu -q -rfE --tb=short > $O/pytest_gpu.log 2>&1; echo "pytest rc $?"; tail -2 $O/pytest_gpu.log
timeout 900 python bench.py --steps 10 --warmup 2 > $O/bench_1gpu.log 2>&1; echo "bench rc $?"
timeout 400 tools/profile_round.sh $TAG cfg2 > $O/profile_cfg2.log 2>&1
B=scenes/_built
timeout 600 python tools/run_config.py --blob $B/cfg2_hdr.blob --width 1280 --height 720 --spp 256 --bounces 8 --parity-spp 256 --tag cfg2_hdr_full > $O/cfg2_hdr_full_parity.json 2> $O/cfg2.err
timeout 600 python tools/run_config.py --blob $B/cfg3_venus.blob --width 1920 --height 1080 --spp 64 --bounces 32 --parity-spp 8 --tag cfg3_venus > $O/cfg3_venus.json 2> $O/cfg3.err
timeout 600 python tools/run_config.py --blob $B/cfg4_statues.blob --width 3840 --height 2160 --spp 16 --bounces 30 --parity-spp 2 --tag cfg4_statues > $O/cfg4_statues.json 2> $O/cfg4.err
timeout 600 python tools/run_config.py --blob $B/soup_1m.blob --width 2560 --height 1440 --spp 32 --bounces 8 --parity-spp 2 --tag soup_1m > $O/soup_1m.json 2> $O/soup1m.err
if [ -n "$SOUP10M" ]; then
	timeout 900 python tools/make_soup_blob.py 10000000 /tmp/crh_soup_10m.blob > $O/soup_10m_build.log 2>&1
	timeout 600 python tools/run_config.py --blob /tmp/crh_soup_10m.blob --width 2560 --height 1440 --spp 16 --bounces 8 --parity-spp 1 --tag soup_10m > $O/soup_10m.json 2> $O/soup10m.err
	timeout 300 python tools/bvh_bench.py --blob /tmp/crh_soup_10m.blob --tag soup_10m > $O/bvh_build_soup_10m.json 2>&1
	CRH_BVH_TRACE=1 timeout 300 python tools/bvh_bench.py --blob /tmp/crh_soup_10m.blob --tag soup_10m --no-cpu 2>&1 | tail -20 > $O/bvh_trace_soup_10m.log
fi
timeout 1500 bash tools/traffic_workloads.sh $TAG > $O/traffic_workloads.log 2>&1
timeout 900 bash tools/pmc_deep.sh $TAG > $O/pmc_deep.log 2>&1
CRH_BVH_TRACE=1 timeout 300 python tools/bvh_bench.py --blob $B/soup_1m.blob --tag soup_1m --no-cpu 2>&1 | tail -16 > $O/bvh_trace_soup_1m.log
timeout 300 python tools/probe_exact.py > $O/probe_exact.log 2>&1
timeout 300 python tools/probe_step_clocks.py > $O/probe_step_clocks.log 2>&1
timeout 300 python tools/bvh_bench.py --blob $B/soup_1m.blob --tag soup_1m > $O/bvh_build_soup_1m.json 2>&1
timeout 300 python tools/bvh_bench.py --blob $B/cfg2_hdr.blob --tag cfg2_hdr > $O/bvh_build_cfg2_hdr.json 2>&1
timeout 300 python tools/probe_share8.py > $O/probe_share8.log 2>&1
CRH_FORCE_PROGRAMS=1 timeout 200 python bench.py --steps 3 --no-cpu --no-dropin > $O/bench_forced_rare_features_variant.log 2>&1
for f in $O/*.json; do echo "$f: $(head -c 300 $f)"; done

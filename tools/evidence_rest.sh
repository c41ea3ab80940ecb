#!/bin/bash
# evidence_rest.sh TAG — the part of tools/evidence_round.sh that tools/evidence_tail_split.sh leaves out (second GPU call of a split evidence round): BASELINE configs 3-5 with counters,
# roofline figures and parity against the oracle, the share ceilings, the degenerate-ray probe, the wait / L1 / L2 / fabric counters, the 10 M-triangle soup (built here) last.
# Afterwards, here: python tools/pmc_deep_table.py gpurun_out/pmc_deep_TAG; python tools/traffic_table.py gpurun_out/traffic_TAG TAG
TAG=${1:-r03zb}
cd "$(dirname "$0")/.." || exit 1
O=gpurun_out/evidence_$TAG
mkdir -p $O
B=scenes/_built
timeout 200 python tools/run_config.py --blob $B/cfg3_venus.blob --width 1920 --height 1080 --spp 64 --bounces 32 --parity-spp 8 --tag cfg3_venus > $O/cfg3_venus.json 2> $O/cfg3.err
timeout 200 python tools/run_config.py --blob $B/cfg4_statues.blob --width 3840 --height 2160 --spp 16 --bounces 30 --parity-spp 2 --tag cfg4_statues > $O/cfg4_statues.json 2> $O/cfg4.err
timeout 200 python tools/run_config.py --blob $B/soup_1m.blob --width 2560 --height 1440 --spp 32 --bounces 8 --parity-spp 2 --tag soup_1m > $O/soup_1m.json 2> $O/soup1m.err
timeout 120 python tools/probe_share8.py > $O/probe_share8.log 2>&1
timeout 120 python tools/probe_exact.py > $O/probe_exact.log 2>&1
timeout 300 bash tools/pmc_deep.sh $TAG > $O/pmc_deep.log 2>&1
timeout 200 python tools/make_soup_blob.py 10000000 /tmp/crh_soup_10m.blob > $O/soup_10m_build.log 2>&1
timeout 200 python tools/run_config.py --blob /tmp/crh_soup_10m.blob --width 2560 --height 1440 --spp 16 --bounces 8 --parity-spp 1 --tag soup_10m > $O/soup_10m.json 2> $O/soup10m.err
timeout 100 python tools/bvh_bench.py --blob /tmp/crh_soup_10m.blob --tag soup_10m > $O/bvh_build_soup_10m.json 2>&1
R=$(pwd); d=$R/gpurun_out/traffic_$TAG/soup10m; mkdir -p $d; echo 8 > $d/spp
(cd /tmp && export TMPDIR=/tmp && timeout 100 rocprofv3 --pmc FETCH_SIZE -d $d -o fetch -- python $R/tools/render_once.py /tmp/crh_soup_10m.blob 2560 1440 8 8 > $d/fetch.log 2>&1; timeout 100 rocprofv3 --pmc WRITE_SIZE -d $d -o write -- python $R/tools/render_once.py /tmp/crh_soup_10m.blob 2560 1440 8 8 > $d/write.log 2>&1)
for f in $O/cfg3_venus.json $O/cfg4_statues.json $O/soup_1m.json $O/soup_10m.json; do echo "$f: $(head -c 260 $f)"; done
tail -3 $O/probe_share8.log | cut -c1-200

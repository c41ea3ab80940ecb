#!/bin/bash
# evidence_tail_split.sh TAG — round 3's last GPU call (about 6 minutes of box time): what CRH_OPT_TAIL_SPLIT buys (tools/probe_tail_split.py picks the setting), then, WITH that setting
# (CRH_TAIL_SPLIT: the process default), the GPU test suite, the bench line, the rocprofv3 trace + PMC groups of the bench command, hdr.json at full size against the oracle, the
# rare-features variant, and — time permitting — the other workloads' FETCH_SIZE / WRITE_SIZE.
# Afterwards, here: python tools/parse_prof.py TAG; python tools/traffic_table.py gpurun_out/traffic_TAG TAG
TAG=${1:-r03zb}
cd "$(dirname "$0")/.." || exit 1
O=gpurun_out/evidence_$TAG
mkdir -p $O
timeout 240 python tools/probe_tail_split.py > $O/probe_tail_split.log 2>&1; echo "probe rc $?"; tail -2 $O/probe_tail_split.log | cut -c1-400
CRH_TAIL_SPLIT=$(cat gpurun_out/tail_split_best.txt 2>/dev/null || echo 0); export CRH_TAIL_SPLIT
echo "CRH_TAIL_SPLIT=$CRH_TAIL_SPLIT"
timeout 600 python -m pytest tests -m gpu -q -rfE --tb=short > $O/pytest_gpu.log 2>&1; echo "pytest rc $?"; tail -2 $O/pytest_gpu.log
timeout 400 python bench.py --steps 10 --warmup 2 > $O/bench_1gpu.log 2>&1; echo "bench rc $?"; tail -1 $O/bench_1gpu.log | cut -c1-600
timeout 400 tools/profile_round.sh $TAG cfg2 > $O/profile_cfg2.log 2>&1
B=scenes/_built
timeout 300 python tools/run_config.py --blob $B/cfg2_hdr.blob --width 1280 --height 720 --spp 256 --bounces 8 --parity-spp 256 --tag cfg2_hdr_full > $O/cfg2_hdr_full_parity.json 2> $O/cfg2.err
head -c 900 $O/cfg2_hdr_full_parity.json; echo
CRH_FORCE_PROGRAMS=1 timeout 200 python bench.py --steps 3 --no-cpu --no-dropin > $O/bench_forced_rare_features_variant.log 2>&1; tail -1 $O/bench_forced_rare_features_variant.log | cut -c1-300
timeout 300 python tools/probe_step_clocks.py > $O/probe_step_clocks.log 2>&1
SKIP_SOUP10M=1 timeout 400 bash tools/traffic_workloads.sh $TAG > $O/traffic_workloads.log 2>&1; echo "traffic rc $?"

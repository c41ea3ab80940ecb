#!/bin/bash
# pmc_quick.sh [scene w h spp bounces reps] — dev (GPU box): two rocprofv3 --pmc passes over tools/pcs_render.py; prints the per-dispatch mean of each
# counter for k_pathtrace and the derived VALU figures. (PMC passes only: never combined with tracing.)
cd "$(dirname "$0")/.." || exit 1
REPO=$(pwd)
export TMPDIR=/tmp
ARGS=${*:-cfg2_hdr 1280 720 256 8 2}
cd /tmp || exit 1
i=0
if [ -n "$TRAFFIC_ONLY" ]; then GROUPS_=("FETCH_SIZE" "WRITE_SIZE"); else GROUPS_=("SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU" \
             "SQ_INSTS_SALU SQ_INSTS_VMEM SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY" \
             "SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM"); fi
rm -rf /tmp/pmcq[0-9]*
for group in "${GROUPS_[@]}"; do
	i=$((i + 1))
	rm -rf /tmp/pmcq$i
	# shellcheck disable=SC2086
	timeout 200 rocprofv3 --pmc $group -d /tmp/pmcq$i -o q --output-format csv -- python "$REPO/tools/pcs_render.py" $ARGS > /tmp/pmcq$i.log 2>&1
	echo "pass $i rc $?"
done
python3 - <<'PY'
import csv, glob, collections
tot = collections.defaultdict(float); n = collections.Counter()
for f in glob.glob('/tmp/pmcq*/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if 'k_pathtrace' not in r.get('Kernel_Name', ''): continue
        tot[r['Counter_Name']] += float(r['Counter_Value']); n[r['Counter_Name']] += 1
disp = {}
for f in glob.glob('/tmp/pmcq*/**/*counter_collection.csv', recursive=True):
    ids = set()
    for r in csv.DictReader(open(f)):
        if 'k_pathtrace' in r.get('Kernel_Name', ''): ids.add(r['Dispatch_Id'])
    for r in csv.DictReader(open(f)):
        if 'k_pathtrace' in r.get('Kernel_Name', ''): disp[r['Counter_Name']] = len(ids)
m = {k: tot[k] / max(disp.get(k, 1), 1) for k in tot}
for k in sorted(m): print(f"{k:28s} {m[k]:.4e}")
g = m.get
if g('SQ_WAVE_CYCLES') and g('SQ_ACTIVE_INST_VALU'):
    print("valu pipe busy      ", round(g('SQ_ACTIVE_INST_VALU') / (g('SQ_WAVE_CYCLES') / 4), 4))
    print("lane utilisation    ", round(g('SQ_THREAD_CYCLES_VALU') / (64 * g('SQ_INSTS_VALU')), 4))
if g('SQ_WAIT_ANY') and g('SQ_WAVE_CYCLES'): print("wait any / wave cyc ", round(g('SQ_WAIT_ANY') / g('SQ_WAVE_CYCLES'), 4))
if g('FETCH_SIZE') is not None and g('WRITE_SIZE') is not None:
    # per REAL dispatch: the preload launch of crh_scene_upload is an empty dispatch of the same kernel and is counted in disp
    real = max(min(disp.get('FETCH_SIZE', 1), disp.get('WRITE_SIZE', 1)) - 1, 1)
    f = tot['FETCH_SIZE'] * 1024 * 2 / real; w = tot['WRITE_SIZE'] * 1024 / real
    print(f"L2<->fabric per frame: read {f/1e9:.1f} GB (FETCH_SIZE x 1024 x 2, gfx950) + write {w/1e9:.1f} GB = {(f+w)/1e9:.1f} GB over {real} frame(s)")
PY

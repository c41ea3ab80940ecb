#!/bin/bash
# make_soup.sh N — build soup_<N>.obj and its scene blob ON THE BOX THAT RUNS IT (the 10 M-triangle blob is 1.1 GB and
# is not shipped): gen_soup (C) -> OBJ in the asset overlay -> crh-flatten (reference loader + BVH builder + flattener).
set -e
N=${1:-10000000}
REPO=$(cd "$(dirname "$0")/.." && pwd)
OUT=${2:-/tmp/soup_${N}.blob}
python3 "$REPO/tools/gen_assets.py" --soup "$N"
python3 "$REPO/tools/refrun.py" "soup_${N}.json" --blob "$OUT" --width 2560 --height 1440 --samples 512 --bounces 8 | tail -2
ls -la "$OUT"

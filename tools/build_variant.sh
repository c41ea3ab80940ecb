#!/bin/bash
# build_variant.sh NAME [hipcc flags ...] — dev: build c-ray_amd/_lib/variants/NAME.so for tools/ab_libs.py. Only the bench instantiation
# of the path-tracing kernel is compiled (-DCRH_DEV_ONLY_BENCH_VARIANT: seconds instead of minutes). CSRC=<dir> builds another copy of csrc/.
set -e
NAME=$1; shift
REPO=$(cd "$(dirname "$0")/.." && pwd)
CSRC=${CSRC:-$REPO/c-ray_amd/csrc}
OUT=$REPO/c-ray_amd/_lib/variants
mkdir -p "$OUT"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O2 -std=c++17 -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt -mllvm -disable-machine-licm -fno-slp-vectorize -fPIC -Wno-unused-function \
	-I"$REPO/include" -I"$CSRC" -DCRH_DEV_ONLY_BENCH_VARIANT "$@" -x hip "$CSRC/cray_hip.hip" "$CSRC/bvh_build.hip" -x none \
	"$REPO/c-ray_amd/_lib/scene_blob.c.o" "$REPO/c-ray_amd/_lib/scene_compile.cpp.o" -shared -ldl -o "$OUT/$NAME.so"
echo "$OUT/$NAME.so"

#!/usr/bin/env python3
"""build.py — compile libcray_hip.so (the product) for gfx950 with hipcc, in-tree.

    python c-ray_amd/build.py [--force]

hipcc cross-compiles without a GPU. Flags that matter for parity with the reference CPU render:
  -ffp-contract=off                              only the explicit fmaf() of the slab test fuses (bvh.c:318-324)
  -fhip-fp32-correctly-rounded-divide-sqrt       IEEE divide / sqrt like the host
and three that matter for speed (-O2 rather than -O3: +0.4..1.7 %, two rounds in a row):
  -mllvm -disable-machine-licm                   the path-tracing kernel is one big loop; hoisting every loop-invariant constant and
                                                 address out of it keeps them live across everything: 87 -> 44 spilled VGPRs, +3..11 %
  -fno-slp-vectorize                             the SLP vectoriser pairs float operations into v_pk_mul_f32 / v_pk_add_f32 / v_pk_fma_f32 (350 of them in the
                                                 bench kernel), and on gfx950 those are far slower than the two plain instructions they replace (measured in
                                                 round 3: six v_pk_fma_f32 in the node step cost 19-30 %): without them +4..5.6 % on every workload, 29
                                                 instead of 38 spilled VGPRs (profiles/r03h_ab_noslp.log); per-component IEEE results are the same
The library has no CPU path: without a HIP device every entry point returns CRH_ERR_NO_DEVICE.
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(HERE)
CSRC = os.path.join(HERE, "csrc")
OUT_DIR = os.path.join(HERE, "_lib")
LIB = os.path.join(OUT_DIR, "libcray_hip.so")

SOURCES = [os.path.join(CSRC, "cray_hip.hip"), os.path.join(CSRC, "bvh_build.hip")]
CXX_SOURCES = [os.path.join(CSRC, "scene_compile.cpp")]      # host-only C++ (g++, -ffp-contract=off: prepared triangles)
C_SOURCES = [os.path.join(HERE, "host", "scene_blob.c")]
def deps():
    """Everything the library is built from: every file of csrc/ (the kernels are header-heavy: pathtrace_roll.h holds the hot kernel) and of include/,
    the C sources, and this script. tests/test_abi.py checks that each file cray_hip.hip / bvh_build.hip #include is in here."""
    out = list(C_SOURCES) + [os.path.abspath(__file__)]
    for d in (CSRC, os.path.join(REPO, "include")):
        out += sorted(os.path.join(d, f) for f in os.listdir(d) if f.endswith((".h", ".hip", ".cpp", ".c")))
    return out


HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=gfx950", "-O2", "-std=c++17", "-ffp-contract=off", "-fhip-fp32-correctly-rounded-divide-sqrt", "-mllvm", "-disable-machine-licm", "-fno-slp-vectorize",
         "-fPIC", "-Wall", "-Wno-unused-function", "-I" + os.path.join(REPO, "include"), "-I" + CSRC]


def up_to_date():
    if not os.path.exists(LIB):
        return False
    t = os.path.getmtime(LIB)
    return all(os.path.getmtime(d) <= t for d in deps())


def build(force=False, verbose=True, extra=()):
    if not force and up_to_date():
        return LIB
    os.makedirs(OUT_DIR, exist_ok=True)
    objs = []
    for src in C_SOURCES:
        obj = os.path.join(OUT_DIR, os.path.basename(src) + ".o")
        subprocess.check_call(["gcc", "-std=gnu99", "-O2", "-fPIC", "-D_GNU_SOURCE", "-I" + os.path.join(REPO, "include"),
                               "-c", src, "-o", obj])
        objs.append(obj)
    for src in CXX_SOURCES:
        obj = os.path.join(OUT_DIR, os.path.basename(src) + ".o")
        subprocess.check_call(["g++", "-std=c++17", "-O2", "-fPIC", "-ffp-contract=off", "-Wall", "-I" + os.path.join(REPO, "include"),
                               "-I" + CSRC, "-c", src, "-o", obj])
        objs.append(obj)
    cmd = [HIPCC] + FLAGS + list(extra) + ["-x", "hip"] + SOURCES + ["-x", "none"] + objs + ["-shared", "-ldl", "-o", LIB + ".tmp"]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    os.replace(LIB + ".tmp", LIB)
    return LIB


if __name__ == "__main__":
    extra = [a for a in sys.argv[1:] if a != "--force"]
    print(build(force="--force" in sys.argv, extra=extra))

#!/usr/bin/env python3
"""kernel_regs.py — compile a .hip source for the device only (gfx950) and print each kernel's register / spill /
scratch / LDS metadata (llvm-readelf --notes). No GPU needed.

    tools/kernel_regs.py [--src FILE] [--filter SUBSTR] [extra hipcc flags, e.g. -DCRH_STACK_LDS=17]
"""
import os
import re
import subprocess
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LLVM = "/opt/rocm/lib/llvm/bin"


def main():
    args = sys.argv[1:]
    src = os.path.join(REPO, "c-ray_amd", "csrc", "cray_hip.hip")
    flt = ""
    extra = []
    i = 0
    while i < len(args):
        if args[i] == "--src":
            src = args[i + 1]; i += 2
        elif args[i] == "--filter":
            flt = args[i + 1]; i += 2
        else:
            extra.append(args[i]); i += 1
    out = os.environ.get("KREGS_OUT", "/tmp/kregs")
    os.makedirs(out, exist_ok=True)
    co = os.path.join(out, os.path.basename(src) + ".co")
    cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O2", "-std=c++17", "-ffp-contract=off",
           "-fhip-fp32-correctly-rounded-divide-sqrt", "-mllvm", "-disable-machine-licm", "-fno-slp-vectorize", "-fPIC", "-Wno-unused-function", "-Wno-pass-failed", "-I" + os.path.join(REPO, "include"),
           "-I" + os.path.join(REPO, "c-ray_amd", "csrc"), "--cuda-device-only", "--no-gpu-bundle-output", "-c", src, "-o", co] + extra
    if not os.environ.get("KREGS_REUSE"):
        subprocess.check_call(cmd)
    notes = subprocess.check_output([os.path.join(LLVM, "llvm-readelf"), "--notes", co]).decode()
    cur = {}
    rows = []
    keys = ("agpr_count", "group_segment_fixed_size", "private_segment_fixed_size", "sgpr_count", "sgpr_spill_count", "vgpr_count",
            "vgpr_spill_count", "symbol")
    for line in notes.splitlines():
        m = re.match(r"\s+(?:- )?\.(\w+):\s+(.*)", line)
        if not m or m.group(1) not in keys:
            continue
        k, v = m.group(1), m.group(2).strip()
        if k in cur:          # a key repeats: the previous kernel's block is complete
            rows.append(cur)
            cur = {}
        cur[k] = v
    if cur.get("symbol"):
        rows.append(cur)
    for r in rows:
        sym = r["symbol"].replace(".kd", "").strip("'")
        name = subprocess.check_output(["c++filt", sym]).decode().strip()
        name = re.sub(r"\(.*", "", name).replace("void ", "")
        if flt and flt not in name:
            continue
        print(f"{name:44s} vgpr {r.get('vgpr_count', '?'):>4s} agpr {r.get('agpr_count', '0'):>3s} vspill {r.get('vgpr_spill_count', '0'):>4s} "
              f"sgpr {r.get('sgpr_count', '?'):>4s} sspill {r.get('sgpr_spill_count', '0'):>4s} scratch {r.get('private_segment_fixed_size', '?'):>5s} "
              f"lds {r.get('group_segment_fixed_size', '?'):>6s}")


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""static_profile.py — dev: an instruction-level cost estimate of the bench kernel without a GPU.
   static:  every instruction of the -g build of k_pathtrace<1,4,false,0>, attributed to its innermost inlined (function, file:line)
            with llvm-symbolizer --inlines;
   dynamic: how often each source line of pt_device.h / exact_math.h runs per lane, from a gcov build of the host emulation (tests/emu)
            rendering the given scene at low resolution;
   estimate = executions(line) x instructions(line) / inlined copies(line), summed per function.
   tools/static_profile.py [scene w h spp bounces]      (needs /tmp write access; ~1 min)"""
import collections, json, os, re, subprocess, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LLVM = "/opt/rocm/lib/llvm/bin"
W = "/tmp/static_profile"
os.makedirs(W, exist_ok=True)
scene = sys.argv[1:6] or ["cfg2_hdr", "320", "180", "8", "8"]
flags = ["--offload-arch=gfx950", "-O2", "-std=c++17", "-ffp-contract=off", "-fhip-fp32-correctly-rounded-divide-sqrt", "-mllvm", "-disable-machine-licm", "-fno-slp-vectorize",
         "-fPIC", "-I" + REPO + "/include", "-I" + REPO + "/c-ray_amd/csrc", "-DCRH_DEV_ONLY_BENCH_VARIANT", "-g", "--cuda-device-only", "--no-gpu-bundle-output"]
co = W + "/dbg.co"
if not os.environ.get("SP_REUSE"):
    subprocess.check_call(["/opt/rocm/bin/hipcc"] + flags + os.environ.get("SP_FLAGS", "").split() + ["-c", "-o", co, "-x", "hip", REPO + "/c-ray_amd/csrc/cray_hip.hip"], stderr=subprocess.DEVNULL)
dis = subprocess.check_output([LLVM + "/llvm-objdump", "-d", "--no-show-raw-insn", co]).decode().splitlines()
ins = []
on = False
for l in dis:
    if re.match(r"^[0-9a-f]+ <", l):
        on = "k_pathtrace_rollILi1ELi4ELb0ELi0ELb0EE" in l          # the bench instantiation (binary walk)
        continue
    if on:
        m = re.match(r"\s+(\S+).*//\s*([0-9A-F]+):", l)
        if m:
            ins.append((int(m.group(2), 16), m.group(1)))
print("instructions:", len(ins))
sym = subprocess.run([LLVM + "/llvm-symbolizer", "--obj=" + co, "--inlines", "--output-style=JSON"], input="\n".join(hex(a) for a, _ in ins), capture_output=True, text=True).stdout
frames = []
for l in sym.splitlines():
    j = json.loads(l)
    fs = [(re.sub(r"\(.*", "", f["FunctionName"]).replace("crh::", ""), os.path.basename(f["FileName"]), f["Line"]) for f in j["Symbol"]]
    frames.append(fs)
# dynamic counts
if not os.environ.get("SP_REUSE_COV"):
    for f in os.listdir(W):
        if f.endswith((".gcda", ".gcov")): os.remove(os.path.join(W, f))
    subprocess.check_call(["g++", "-std=c++17", "-O0", "-g", "--coverage", "-march=x86-64-v3", "-ffp-contract=off", "-fPIC", "-shared", "-I" + REPO + "/include", "-I" + REPO + "/tests/emu",
                           REPO + "/tests/emu/emu.cpp", REPO + "/c-ray_amd/csrc/scene_compile.cpp", "-o", W + "/libcray_emu_cov.so"], cwd=W)
    drv = f"""
import sys, ctypes as C
sys.path.insert(0,{REPO!r}); sys.path.insert(0,{REPO!r}+'/oracle'); sys.path.insert(0,{REPO!r}+'/tests')
import numpy as np, oracle_py as oracle
abi = oracle.abi
L = C.CDLL({W!r}+'/libcray_emu_cov.so')
L.emu_render_region.argtypes = [C.POINTER(abi.SceneDesc), C.POINTER(abi.RenderParams), C.c_void_p, C.POINTER(abi.Counters), C.POINTER(C.c_uint32), C.c_int, C.c_int, C.c_int]
name, w, h, s, b = {scene[0]!r}, {int(scene[1])}, {int(scene[2])}, {int(scene[3])}, {int(scene[4])}
sc = oracle.OracleScene({REPO!r}+'/scenes/_built/%s.blob' % name)
sc.desc.camera.width, sc.desc.camera.height = w, h
fb = np.zeros((h, w, 3), np.float32); cnt, hi = abi.Counters(), C.c_uint32()
p = abi.RenderParams(0, 0, w, h, w, h, 0, s, s, b)
print(L.emu_render_region(sc.ptr, C.byref(p), fb.ctypes.data, C.byref(cnt), C.byref(hi), 8, 8, 64), cnt.as_dict())
"""
    r = subprocess.run([sys.executable, "-c", drv], cwd=W, capture_output=True, text=True)
    print(r.stdout.strip(), r.stderr.strip()[-300:])
    subprocess.run(["gcov", "-o", ".", "libcray_emu_cov.so-emu.gcno"], cwd=W, capture_output=True)
dyn = {}
rays = None
for fn in ("pt_device.h", "exact_math.h"):
    p = os.path.join(W, fn + ".gcov")
    if not os.path.exists(p): continue
    for l in open(p, errors="replace"):
        m = re.match(r"\s*([0-9#=\-\*]+)\*?:\s*(\d+):", l)
        if m and m.group(1)[0].isdigit():
            dyn[(fn, int(m.group(2)))] = max(dyn.get((fn, int(m.group(2))), 0), int(m.group(1).rstrip("*")))
# per own-source line: static count and inlined copies (distinct outer call chains)
OWN = ("pt_device.h", "exact_math.h")
stat = collections.Counter(); chains = collections.defaultdict(set); stat_f = collections.Counter()
rows = []
for (a, mn), fs in zip(ins, frames):
    i = next((j for j, f in enumerate(fs) if f[1] in OWN), None)
    if i is None:
        stat_f["(kernel: " + fs[0][1] + ")"] += 1
        continue
    own = fs[i]
    k = (own[1], own[2])
    stat[k] += 1
    stat_f[own[0]] += 1
    chains[k].add(tuple((f[0], f[2]) for f in fs[i + 1:]))
    rows.append((k, own[0]))
est_f = collections.Counter(); est_l = collections.Counter()
for k, fn in rows:
    c = dyn.get(k, 0) / max(len(chains[k]), 1)
    est_f[fn] += c
    est_l[k] += c
tot = sum(est_f.values())
print(f"estimated lane-instructions (pt_device.h + exact_math.h code only): {tot:.3e}")
print("by function (est share, static instrs):")
for k, v in est_f.most_common(45): print(f"  {k:34s} {100*v/tot:5.1f}%  static {stat_f[k]}")
print("by line:")
for k, v in est_l.most_common(60): print(f"  {k[0]}:{k[1]:5d} {100*v/tot:5.1f}%  static {stat[k]:5d} copies {len(chains[k]):3d} runs {dyn.get(k,0)}")

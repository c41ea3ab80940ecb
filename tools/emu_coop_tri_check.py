#!/usr/bin/env python3
"""emu_coop_tri_check.py — dev: the cooperative leaf step (profiles/r06g_exp_coop_tri.patch: -DCRH_EXP_COOP_TRI, pathtrace_roll.h: coopTriStep; round 6; a
measured negative, so the product source does not carry it: `git apply profiles/r06g_exp_coop_tri.patch` first) against the shipped two-triangle
step in the kernel emulation: same frame and the same eight counters (tri_tests among them) required. Builds nothing: first
    cd tests/emu && g++ -std=c++17 -O1 -march=x86-64-v3 -ffp-contract=off -fPIC -pthread -Wno-attributes -Wno-unknown-pragmas -Ihipemu -I../../include -I../../c-ray_amd/csrc \
        -DCRH_DEV_ONLY_BENCH_VARIANT -DCRH_DEV_ONLY_LEVEL2 -DCRH_EXP_COOP_TRI -c kernel_emu.cpp -o _obj/kernel_emu_coop.o && \
        g++ -shared -pthread _obj/kernel_emu_coop.o _obj/bvh_emu.o _obj/hipemu.o _obj/scene_compile.o _obj/scene_blob.o -ldl -o _obj/libcray_hip_emu_coop.so
then  python tools/emu_coop_tri_check.py cfg1_scene:320:200:4:4 fence:128:100:4:6 soup_1m:160:90:2:8 cfg4_statues:160:90:2:30      (scene:width:height:spp:bounces)
On the MI355X: tools/build_variant.sh base; tools/build_variant.sh coop -DCRH_EXP_COOP_TRI; tools/ab_libs.py base coop (profiles/r06g_ab_coop_tri.log: slower)."""
import sys, os, subprocess, json, hashlib
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if os.environ.get("CHILD"):
    import numpy as np
    sys.path.insert(0, REPO)
    from __graft_entry__ import load_package
    pkg = load_package(); api = pkg.api; abi = pkg.abi
    res = {}
    for spec in sys.argv[1:]:
        name, w, h, spp, b = spec.split(":"); w, h, spp, b = int(w), int(h), int(spp), int(b)
        blob = name if name.startswith("/") else f"{REPO}/scenes/_built/{name}.blob"
        if not os.path.exists(blob):
            import gzip, shutil, tempfile
            src = f"{REPO}/tests/golden/{name}.blob.gz"; blob = tempfile.mktemp(suffix=".blob")
            with gzip.open(src) as f, open(blob, "wb") as o: shutil.copyfileobj(f, o)
        scene = api.Scene(blob)
        if name.startswith("cfg") or name.startswith("soup"):
            sys.path.insert(0, f"{REPO}/tests")
            from conftest import resize_camera
            scene = resize_camera(scene, w, h)
        ctx = api.Context(0)
        ctx.upload(scene)
        fb = ctx.framebuffer(w, h)
        ctx.reset_counters()
        ctx.render_region(fb, w, h, spp, b)
        img = ctx.download(fb, w, h)
        res[spec] = {"md5": hashlib.md5(img.tobytes()).hexdigest(), "cnt": ctx.counters(), "kernel": ctx.last_kernel_name()}
        ctx.close()
    print("RESULT " + json.dumps(res))
    sys.exit(0)
out = {}
for tag, lib in (("base", f"{REPO}/tests/emu/libcray_hip_emu.so"), ("coop", f"{REPO}/tests/emu/_obj/libcray_hip_emu_coop.so")):
    env = dict(os.environ, CHILD="1", CRH_LIB=lib, CRH_ALLOW_EMULATION="1", HIPEMU_CUS="2", HIPEMU_THREADS="6")
    r = subprocess.run([sys.executable, __file__] + sys.argv[1:], env=env, capture_output=True, text=True)
    line = [l for l in r.stdout.splitlines() if l.startswith("RESULT ")]
    if not line: print(tag, "FAILED", r.stdout[-2000:], r.stderr[-3000:]); sys.exit(1)
    out[tag] = json.loads(line[0][7:])
for spec in out["base"]:
    a, b = out["base"][spec], out["coop"][spec]
    print(spec, "frame", "SAME" if a["md5"] == b["md5"] else "DIFFERS", "counters", "SAME" if a["cnt"] == b["cnt"] else f"DIFFER {a['cnt']} {b['cnt']}", b["kernel"])

#!/usr/bin/env python3
"""gen_bmp_golden.py — md5 of the BMP files the REAL reference (oracle/_ref/c-ray-ref-strict) writes for config 1, normal and --iterative
("BMP is byte-deterministic -> use for golden hashes"), into tests/golden/manifest.json (`bmp_md5`). The drop-in
program's BMP — the reference's own encoder fed by renderer_hip.c's 8-bit output, which is converted ON THE DEVICE — must hash equal
(tests/test_gpu_parity.py: test_dropin_binary...). Needs /root/reference-built oracle/_ref; run where that exists."""
import glob, hashlib, json, os, subprocess, sys, tempfile
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "tools"))
import refrun
path = os.path.join(REPO, "tests", "golden", "manifest.json")
man = json.load(open(path))
for key, args in (("cfg1_scene", ()), ("cfg1_scene_iterative", ("--iterative",))):
    m = man[key]
    w, h = man["cfg1_scene"]["width"], man["cfg1_scene"]["height"]
    with tempfile.TemporaryDirectory() as tmp:
        scene = refrun.rewrite_scene("scene.json", w, h, m["samples"], m["bounces"], out_dir=tmp)
        exe = os.path.join(refrun.REF_DIR, "c-ray-ref-strict")
        subprocess.run([exe, "-j", "1" if args else "8", *args], input=json.dumps(scene).encode(), cwd=refrun.INPUT_DIR, check=True,
                       stdout=subprocess.DEVNULL, stderr=subprocess.STDOUT, timeout=1800)
        bmp = glob.glob(os.path.join(tmp, "*.bmp"))
        assert len(bmp) == 1, bmp
        m["bmp_md5"] = hashlib.md5(open(bmp[0], "rb").read()).hexdigest()
        print(key, os.path.basename(bmp[0]), m["bmp_md5"])
json.dump(man, open(path, "w"), indent=1)

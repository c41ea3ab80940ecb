#!/usr/bin/env python3
"""gen_golden.py — (re)generate tests/golden/ from the REAL reference (oracle/_ref, built from /root/reference).

For every case: the scene blob written by crh-flatten (reference loader + product flattener), the linear
float render buffer of c-ray-ref-strict (the unmodified reference sources, -ffp-contract=off), and the
ray / node-test / triangle-test counts of c-ray-ref-count. Runs only where /root/reference exists; the
fixtures are committed so that the tests can run on the GPU box, where it does not.
"""
import gzip
import hashlib
import json
import os
import sys

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "tools"))
import refrun  # noqa: E402

GOLDEN = os.path.join(REPO, "tests", "golden")

# name: (scene json, W, H, spp, bounces)   bounces None = keep the JSON's own value
CASES = {
    "cfg1_scene": ("scene.json", 320, 200, 4, 4),            # BASELINE.json configs[0]
    "alphanode": ("alphanode.json", 160, 100, 4, None),      # node-graph materials: alpha / mix / texture nodes
    "fence": ("fence.json", 160, 100, 4, 6),                 # alpha-textured mesh (map_d style cut-outs)
    "glowmetal": ("glowmetal.json", 160, 100, 4, 6),         # emissive + metal
    "refraction": ("refraction.json", 160, 100, 4, 8),       # glass spheres
    "uvsphere": ("uvsphere.json", 160, 100, 4, 6),           # textured sphere (getTexMapSphere)
}


# BASELINE.json configs[1..4] at a reduced 16:9 frame (the sensor depends only on FOV and aspect ratio, camera.c:30-32, so the
# committed fixture is only the reference's float buffer: the scene comes from scenes/_built/<blob>.blob with camera.width /
# height patched). Next to the strict buffer, the "chaos floor": how far the SAME reference sources built with the upstream
# default flags (FMA contraction on: c-ray-ref) land from the strict build — what any other correct fp32 implementation is allowed.
BIG_CASES = {
    "cfg2_hdr_small": ("hdr.json", "cfg2_hdr", 320, 180, 4, 8),
    "cfg3_venus_small": ("venus.json", "cfg3_venus", 320, 180, 4, 32),
    "cfg4_statues_small": ("statues.json", "cfg4_statues", 320, 180, 4, 30),
    "soup_1m_small": ("soup_1000000.json", "soup_1m", 320, 180, 4, 8),
}


def per_pixel_stats(a, b):
    d = a.astype(np.float64) - b.astype(np.float64)
    l2 = np.sqrt((d ** 2).sum(axis=2))
    return {"rmse": float(np.sqrt((d ** 2).mean())), "frac_gt_1e-3": float((l2 > 1e-3).mean()), "mean_l2": float(l2.mean())}


def gz_write(path, data):
    with gzip.GzipFile(path, "wb", compresslevel=9, mtime=0) as f:
        f.write(data)


def main():
    os.makedirs(GOLDEN, exist_ok=True)
    manifest = {}
    only = sys.argv[1:]
    mpath = os.path.join(GOLDEN, "manifest.json")
    if only and os.path.exists(mpath):
        manifest = json.load(open(mpath))
    for name, (scene, w, h, spp, bounces) in CASES.items():
        if only and name not in only:
            continue
        if bounces is None:
            bounces = json.load(open(os.path.join(refrun.INPUT_DIR, scene)))["renderer"]["bounces"]
        tmp_blob = os.path.join("/tmp", f"golden_{name}.blob")
        refrun.flatten_scene(scene, tmp_blob, w, h, spp, bounces)
        buf, st = refrun.render_reference(scene, w, h, spp, bounces, "strict")
        _, cst = refrun.render_reference(scene, w, h, spp, bounces, "count")
        blob = open(tmp_blob, "rb").read()
        gz_write(os.path.join(GOLDEN, name + ".blob.gz"), blob)
        gz_write(os.path.join(GOLDEN, name + ".ref.f32.gz"), buf.tobytes())
        manifest[name] = {"scene": scene, "width": w, "height": h, "samples": spp, "bounces": bounces,
                          "ref_flavour": "c-ray-ref-strict (unmodified reference sources, gcc -O2 -march=x86-64-v3 -ffp-contract=off)",
                          "ref_md5": hashlib.md5(buf.tobytes()).hexdigest(), "blob_md5": hashlib.md5(blob).hexdigest(),
                          "rays": cst["rays"], "node_tests": cst["node_tests"], "tri_tests": cst["tri_tests"],
                          "mean": float(buf.mean())}
        print(name, manifest[name], len(blob))
        if name == "cfg1_scene":
            # the interactive mode (renderThreadInteractive: Halton sampler, passes 1..samples-1),
            # same scene blob, reference run with --iterative on one thread (the only reproducible way to run it)
            ispp = 6
            ibuf, _ = refrun.render_reference(scene, w, h, ispp, bounces, "strict", iterative=True)
            gz_write(os.path.join(GOLDEN, name + "_iterative.ref.f32.gz"), ibuf.tobytes())
            manifest[name + "_iterative"] = {"scene": scene, "blob": name, "width": w, "height": h, "samples": ispp, "passes": ispp - 1,
                                             "bounces": bounces, "ref_flavour": "c-ray-ref-strict --iterative -j 1",
                                             "ref_md5": hashlib.md5(ibuf.tobytes()).hexdigest(), "mean": float(ibuf.mean())}
            print(name + "_iterative", manifest[name + "_iterative"])
    for name, (scene, blob, w, h, spp, bounces) in BIG_CASES.items():
        if only and name not in only:
            continue
        buf, _ = refrun.render_reference(scene, w, h, spp, bounces, "strict")
        fma, _ = refrun.render_reference(scene, w, h, spp, bounces, "default")
        _, cst = refrun.render_reference(scene, w, h, spp, bounces, "count")
        gz_write(os.path.join(GOLDEN, name + ".ref.f32.gz"), buf.tobytes())
        manifest[name] = {"scene": scene, "built_blob": blob, "width": w, "height": h, "samples": spp, "bounces": bounces,
                          "ref_flavour": "c-ray-ref-strict", "ref_md5": hashlib.md5(buf.tobytes()).hexdigest(),
                          "rays": cst["rays"], "node_tests": cst["node_tests"], "tri_tests": cst["tri_tests"], "mean": float(buf.mean()),
                          "floor": dict(per_pixel_stats(fma, buf), flavour="c-ray-ref (same sources, FMA contraction on) vs c-ray-ref-strict")}
        print(name, manifest[name])
    with open(mpath, "w") as f:
        json.dump(manifest, f, indent=1, sort_keys=True)


if __name__ == "__main__":
    main()

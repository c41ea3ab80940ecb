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


# Fixtures for the nodes no JSON path reaches: the scene is written here,
# the exotic graphs are built INSIDE the reference by oracle/ref_node_patch.c (CRH_NODE_PATCH). name: (scene, patch, W, H, spp, bounces)
PATCHED_CASES = {
    "nodezoo": ("nodezoo.json", {"CRH_NODE_PATCH": "zoo", "CRH_ZOO_TAME": "1"}, 320, 192, 4, 6),   # exotic + JSON graphs, full paths
    "nodezoo_display": ("nodezoo.json", {"CRH_NODE_PATCH": "zoo"}, 320, 192, 1, 2),    # known answers: first hit x white background
    # participating media. Sphere volumes (radius marked ...7) and a cube-shaped mesh volume with solid objects inside and
    # behind them, in front of the HDR environment (instance.c:62-92, 187-216; isotropic.c:40-47)
    "volumes": ("volumes.json", {"CRH_NODE_PATCH": "volumes", "CRH_VOLUME_MESHES": "0", "CRH_VOLUME_DENSITY": "9"}, 240, 160, 8, 12),
}
ZOO_COLS, ZOO_ROWS, ZOO_PITCH, ZOO_RADIUS = 10, 6, 0.1, 0.042


def zoo_sphere_center(i):
    """World position of nodezoo sphere i (row-major from the top left as the camera sees it)."""
    col, row = i % ZOO_COLS, i // ZOO_COLS
    return ((col - (ZOO_COLS - 1) / 2) * ZOO_PITCH, ((ZOO_ROWS - 1) / 2 - row) * ZOO_PITCH, 0.0)


def write_nodezoo_scene():
    """60 spheres in a 10 x 6 grid facing the camera + an untextured cube and a textured plane. Spheres 0..49 are re-materialised
    by ref_node_patch.c; 50..59 and the meshes carry the node graphs the JSON loader itself can build (sceneloader.c:766-870)."""
    grid = "shapes/grid.png"
    json_materials = [
        {"type": "add", "A": {"type": "diffuse", "color": [0.5, 0.1, 0.1]}, "B": {"type": "metal", "color": [0.1, 0.1, 0.5], "roughness": 0.1}},
        {"type": "diffuse", "color": {"type": "checkerboard", "size": 30}},
        {"type": "diffuse", "color": {"type": "blackbody", "degrees": 3500}},
        {"type": "metal", "color": [0.9, 0.9, 0.9], "roughness": {"path": grid, "lerp": False, "transform": False}},
        {"type": "diffuse", "color": {"path": grid, "lerp": True, "transform": True}},
        {"type": "diffuse", "color": {"path": grid}},
        {"type": "glass", "color": [1.0, 1.0, 1.0], "roughness": 0.05, "IOR": 1.5},
        {"type": "plastic", "color": [0.2, 0.3, 0.8]},
        {"type": "mix", "A": {"type": "diffuse", "color": [0.8, 0.2, 0.2]}, "B": {"type": "transparent", "color": [1.0, 1.0, 1.0]},
         "factor": {"path": grid, "lerp": True}},
        {"type": "emissive", "color": [1.0, 0.8, 0.6], "strength": 3.0},
    ]
    prims = []
    for i in range(ZOO_COLS * ZOO_ROWS):
        x, y, z = zoo_sphere_center(i)
        p = {"type": "sphere", "bsdf": "lambertian", "color": {"r": 0.8, "g": 0.8, "b": 0.8}, "radius": ZOO_RADIUS, "IOR": 1.45,
             "instances": [{"transforms": [{"type": "rotateY", "degrees": 20 + 3 * i}, {"type": "translate", "x": x, "y": y, "z": z}]}]}
        if i >= 50:
            p["material"] = json_materials[i - 50]
        prims.append(p)
    meshes = [
        {"fileName": "shapes/cube.obj", "bsdf": "lambertian",       # no texture coordinates: checker.c takes the hit-point branch
         "material": {"type": "diffuse", "color": {"type": "checkerboard", "size": 40}},
         "instances": [{"transforms": [{"type": "scaleUniform", "scale": 0.04}, {"type": "rotateY", "degrees": 30},
                                       {"type": "translate", "x": -0.62, "y": 0.0, "z": 0.0}]}]},
        {"fileName": "shapes/gridplane.obj", "bsdf": "lambertian",  # textured, far behind the grid
         "instances": [{"transforms": [{"type": "scaleUniform", "scale": 0.3}, {"type": "rotateX", "degrees": -90},
                                       {"type": "translate", "x": 0.62, "y": 0.0, "z": 0.3}]}]},
    ]
    scene = {"version": 1.0,
             "renderer": {"threads": 0, "samples": 4, "bounces": 6, "antialiasing": True, "tileWidth": 32, "tileHeight": 32, "tileOrder": "fromMiddle",
                          "outputFilePath": "/tmp/", "outputFileName": "nodezoo", "fileType": "bmp", "count": 0, "width": 320, "height": 192},
             "display": {"isFullscreen": False, "isBorderless": False, "windowScale": 1.0},
             "camera": {"FOV": 28.0, "focalDistance": 3.0, "fstops": 0, "transforms": [{"type": "translate", "x": 0, "y": 0, "z": -3.0}]},
             "scene": {"ambientColor": {"offset": 0, "down": {"r": 1.0, "g": 1.0, "b": 1.0}, "up": {"r": 0.5, "g": 0.7, "b": 1.0}},
                       "primitives": prims, "meshes": meshes}}
    with open(os.path.join(refrun.INPUT_DIR, "nodezoo.json"), "w") as f:
        json.dump(scene, f, indent=1)


def write_volumes_scene():
    def sphere(x, y, z, radius, color, bsdf="lambertian", **kw):
        p = {"type": "sphere", "bsdf": bsdf, "color": dict(zip("rgb", color)), "radius": radius, "IOR": 1.45,
             "instances": [{"transforms": [{"type": "translate", "x": x, "y": y, "z": z}]}]}
        p.update(kw)
        return p
    prims = [
        sphere(-0.45, 0.0, 0.0, 0.2507, (0.5, 0.7, 0.9)),             # volume (radius marked ...7), nothing inside
        sphere(0.45, 0.05, 0.1, 0.3007, (0.5, 0.7, 0.9)),             # volume with a solid metal sphere inside and one poking through
        sphere(0.45, 0.05, 0.1, 0.1, (0.9, 0.6, 0.2), "metal", roughness=0.1),
        sphere(0.75, 0.0, 0.0, 0.12, (0.8, 0.2, 0.2)),
        sphere(0.0, -100.3, 0.0, 100.0, (0.6, 0.6, 0.6)),             # floor
        sphere(-0.1, 0.55, 0.4, 0.12, (1.0, 0.9, 0.7), "emissive", intensity=8.0),
        sphere(0.0, 0.0, 0.05, 0.08, (0.2, 0.8, 0.3)),                # solid sphere inside the mesh volume
    ]
    meshes = [
        {"fileName": "shapes/cube.obj", "bsdf": "lambertian",          # becomes the mesh volume (CRH_VOLUME_MESHES=0), two instances
         "instances": [{"transforms": [{"type": "scaleUniform", "scale": 0.2}, {"type": "rotateY", "degrees": 25}, {"type": "translate", "x": 0.0, "y": 0.0, "z": 0.0}]},
                       {"transforms": [{"type": "scale", "x": 0.1, "y": 0.25, "z": 0.1}, {"type": "rotateZ", "degrees": 15}, {"type": "translate", "x": -0.05, "y": 0.1, "z": -0.6}]}]},
        {"fileName": "shapes/torus.obj", "bsdf": "lambertian",
         "instances": [{"transforms": [{"type": "scaleUniform", "scale": 0.15}, {"type": "rotateX", "degrees": 60}, {"type": "translate", "x": -0.45, "y": 0.0, "z": 0.0}]}]},
    ]
    scene = {"version": 1.0,
             "renderer": {"threads": 0, "samples": 8, "bounces": 12, "antialiasing": True, "tileWidth": 32, "tileHeight": 32, "tileOrder": "fromMiddle",
                          "outputFilePath": "/tmp/", "outputFileName": "volumes", "fileType": "bmp", "count": 0, "width": 240, "height": 160},
             "display": {"isFullscreen": False, "isBorderless": False, "windowScale": 1.0},
             "camera": {"FOV": 40.0, "focalDistance": 2.5, "fstops": 0, "transforms": [{"type": "translate", "x": 0, "y": 0.15, "z": -2.4}, {"type": "rotateX", "degrees": 3}]},
             "scene": {"ambientColor": {"hdr": "HDRs/roof_garden_1k.hdr", "offset": 0, "down": {"r": 1.0, "g": 1.0, "b": 1.0}, "up": {"r": 0.5, "g": 0.7, "b": 1.0}},
                       "primitives": prims, "meshes": meshes}}
    with open(os.path.join(refrun.INPUT_DIR, "volumes.json"), "w") as f:
        json.dump(scene, f, indent=1)


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
    blobs_done = {}
    for name, (scene, env, w, h, spp, bounces) in PATCHED_CASES.items():
        if only and name not in only:
            continue
        if scene == "nodezoo.json":
            write_nodezoo_scene()
        if scene == "volumes.json":
            write_volumes_scene()
        entry = {"scene": scene, "patch": env, "width": w, "height": h, "samples": spp, "bounces": bounces,
                 "ref_flavour": "c-ray-ref-strict + oracle/ref_node_patch.c"}
        if (scene, str(env)) not in blobs_done:
            tmp_blob = os.path.join("/tmp", f"golden_{name}.blob")
            refrun.flatten_scene(scene, tmp_blob, w, h, spp, bounces, env=env)
            blob = open(tmp_blob, "rb").read()
            gz_write(os.path.join(GOLDEN, name + ".blob.gz"), blob)
            blobs_done[(scene, str(env))] = name
            entry["blob_md5"] = hashlib.md5(blob).hexdigest()
        entry["blob"] = blobs_done[(scene, str(env))]
        buf, st = refrun.render_reference(scene, w, h, spp, bounces, "strict", env=env)
        _, cst = refrun.render_reference(scene, w, h, spp, bounces, "count", env=env)
        gz_write(os.path.join(GOLDEN, name + ".ref.f32.gz"), buf.tobytes())
        entry.update({"ref_md5": hashlib.md5(buf.tobytes()).hexdigest(), "rays": cst["rays"], "node_tests": cst["node_tests"],
                      "tri_tests": cst["tri_tests"], "mean": float(buf.mean())})
        manifest[name] = entry
        print(name, entry)
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

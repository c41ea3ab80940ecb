#!/usr/bin/env python3
"""gen_assets.py — build the asset overlay `oracle/_ref/input/` (TEST / BENCH INFRASTRUCTURE).

The reference reads assets relative to its cwd and `/root/reference` is read-only, so every run of
the reference binaries and of the flattener uses an overlay directory:
  * a copy of /root/reference/input (scene JSONs, OBJ/MTL, PNG, HDR),
  * a deterministic STAND-IN for `venusscaled.obj`, which the reference tree does not ship
    (/root/reference/.MISSING_LARGE_BLOBS lists it; SURVEY.md §8(d) "Missing asset"). The stand-in is a
    procedurally displaced closed surface of revolution, 524 288 triangles with `vn` normals,
    base at y = 0, ~3.2 units tall (so the scenes' scaleUniform 0.05 / 70 framing works), using
    the shipped `venusscaled.mtl` (material `default`). Every report that uses it says so.
  * the synthetic triangle soups of BASELINE.json config 5 (`soup_<N>.obj` + `soup_<N>.json`),
    written by tools/gen_soup (C) on demand: `gen_assets.py --soup 1000000`.

Nothing here is copied into git: oracle/_ref/ is git-ignored (it still travels to the GPU box).
"""
import argparse
import os
import shutil
import subprocess
import sys

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF_INPUT = "/root/reference/input"
OVERLAY = os.path.join(REPO, "oracle", "_ref", "input")


def copy_reference_input():
    if not os.path.isdir(REF_INPUT):
        return False
    for root, dirs, files in os.walk(REF_INPUT):
        rel = os.path.relpath(root, REF_INPUT)
        dst = os.path.join(OVERLAY, rel) if rel != "." else OVERLAY
        os.makedirs(dst, exist_ok=True)
        for f in files:
            d = os.path.join(dst, f)
            if not os.path.exists(d):
                shutil.copyfile(os.path.join(root, f), d)
                os.chmod(d, 0o644)
    return True


def statue_vertices(rows, cols):
    """Closed surface: pole at the bottom (y=0), pole at the top, `rows` interior rings of `cols` vertices."""
    v = (np.arange(1, rows + 1, dtype=np.float64) / (rows + 1))[:, None]          # (rows,1) in (0,1)
    u = (np.arange(cols, dtype=np.float64) / cols * 2.0 * np.pi)[None, :]         # (1,cols)
    height = 3.2
    y = v * height
    # lathe profile: plinth, legs/drape, waist, torso, shoulders, neck, head
    prof = (0.62 * np.exp(-((v - 0.02) / 0.05) ** 2)
            + 0.42 * np.exp(-((v - 0.25) / 0.22) ** 2)
            + 0.30 * np.exp(-((v - 0.55) / 0.10) ** 2)
            + 0.40 * np.exp(-((v - 0.70) / 0.08) ** 2)
            + 0.14 * np.exp(-((v - 0.82) / 0.04) ** 2)
            + 0.20 * np.exp(-((v - 0.92) / 0.05) ** 2))
    prof = prof * np.sin(np.pi * v) ** 0.35 + 0.01
    # asymmetric low-frequency shape + drapery folds + fine chisel detail
    shape = 1.0 + 0.18 * np.cos(u - 2.0 * v) + 0.10 * np.cos(2.0 * u + 5.0 * v)
    folds = 0.035 * np.sin(14.0 * u + 9.0 * np.sin(6.0 * v)) * np.exp(-((v - 0.28) / 0.2) ** 2)
    detail = 0.006 * np.sin(61.0 * u + 3.0) * np.sin(173.0 * v) + 0.004 * np.sin(127.0 * u) * np.cos(97.0 * v + 1.0)
    r = prof * shape + folds + detail
    lean = 0.12 * np.sin(2.2 * v)                                                  # contrapposto lean in x
    x = r * np.cos(u) + lean
    z = r * np.sin(u) * 0.8
    yy = np.broadcast_to(y, x.shape)
    ring = np.stack([x, yy, z], axis=-1)                                           # (rows, cols, 3)
    bottom = np.array([[0.0, 0.0, 0.0]])
    top = np.array([[lean[-1, 0], height, 0.0]])
    return ring, bottom, top


def write_statue(path, rows=512, cols=512):
    ring, bottom, top = statue_vertices(rows, cols)
    # vertex normals from central differences of the ring grid (wrap in u, clamp in v)
    du = np.roll(ring, -1, axis=1) - np.roll(ring, 1, axis=1)
    dv = np.empty_like(ring)
    dv[1:-1] = ring[2:] - ring[:-2]
    dv[0] = ring[1] - bottom
    dv[-1] = top - ring[-2]
    n = np.cross(dv, du)
    n /= np.maximum(np.linalg.norm(n, axis=-1, keepdims=True), 1e-20)
    verts = np.concatenate([bottom, ring.reshape(-1, 3), top]).astype(np.float32)
    norms = np.concatenate([[[0.0, -1.0, 0.0]], n.reshape(-1, 3), [[0.0, 1.0, 0.0]]]).astype(np.float32)

    def vid(i, j):  # 1-based OBJ index of ring vertex (i, j)
        return 2 + i * cols + (j % cols)

    i = np.arange(rows - 1)[:, None]
    j = np.arange(cols)[None, :]
    a, b = vid(i, j), vid(i, j + 1)
    c, d = vid(i + 1, j), vid(i + 1, j + 1)
    quads1 = np.stack([np.broadcast_to(a, c.shape), c, d], axis=-1).reshape(-1, 3)
    quads2 = np.stack([np.broadcast_to(a, c.shape), d, np.broadcast_to(b, c.shape)], axis=-1).reshape(-1, 3)
    jj = np.arange(cols)
    capb = np.stack([np.full(cols, 1), vid(0, jj), vid(0, jj + 1)], axis=-1)
    top_id = 2 + rows * cols
    capt = np.stack([np.full(cols, top_id), vid(rows - 1, jj + 1), vid(rows - 1, jj)], axis=-1)
    faces = np.concatenate([capb, np.stack([quads1, quads2], axis=1).reshape(-1, 3), capt])
    tmp = path + ".tmp"
    with open(tmp, "w") as f:
        f.write("# STAND-IN for venusscaled.obj generated by tools/gen_assets.py (the original is not in the reference tree)\n")
        f.write("mtllib venusscaled.mtl\no venus_standin\n")
        f.write("".join("v %.6f %.6f %.6f\n" % tuple(p) for p in verts))
        f.write("".join("vn %.6f %.6f %.6f\n" % tuple(p) for p in norms))
        f.write("usemtl default\ns 1\n")
        f.write("".join("f %d//%d %d//%d %d//%d\n" % (p, p, q, q, r, r) for p, q, r in faces))
    os.replace(tmp, path)
    return len(faces)


SOUP_JSON = """{
	"version": 1.0,
	"renderer": {"threads": 0, "samples": 512, "bounces": 8, "antialiasing": true, "tileWidth": 64, "tileHeight": 64,
		"tileOrder": "fromMiddle", "outputFilePath": "output/", "outputFileName": "soup", "fileType": "png", "count": 0,
		"width": 2560, "height": 1440},
	"display": {"isFullscreen": false, "isBorderless": false, "windowScale": 1.0},
	"camera": {"FOV": 60.0, "focalDistance": 3.5, "fstops": 0.0,
		"transforms": [{"type": "translate", "x": 0, "y": 0, "z": -3.5}]},
	"scene": {
		"ambientColor": {"down": {"r": 1.0, "g": 1.0, "b": 1.0}, "up": {"r": 0.5, "g": 0.7, "b": 1.0}},
		"primitives": [],
		"meshes": [{"fileName": "%s", "bsdf": "lambertian", "instances": [{"transforms": []}]}]
	}
}
"""


def ensure_soup(n):
    """soup_<n>.obj/.mtl/.json: SURVEY.md §8(d) synthetic soup (PCG32 seed 42), written by tools/gen_soup.c."""
    obj = os.path.join(OVERLAY, f"soup_{n}.obj")
    js = os.path.join(OVERLAY, f"soup_{n}.json")
    if not os.path.exists(obj):
        exe = os.path.join(REPO, "oracle", "_ref", "gen_soup")
        src = os.path.join(REPO, "tools", "gen_soup.c")
        if not os.path.exists(exe) or os.path.getmtime(exe) < os.path.getmtime(src):
            subprocess.check_call(["gcc", "-O2", "-o", exe, src, "-lm"])
        subprocess.check_call([exe, str(n), obj + ".tmp"])
        os.replace(obj + ".tmp", obj)
    mtl = os.path.join(OVERLAY, "soup.mtl")
    if not os.path.exists(mtl):
        with open(mtl, "w") as f:
            f.write("newmtl grey\nKd 0.5 0.5 0.5\nKe 0 0 0\nNi 1.0\nd 1.0\nillum 2\n")
    if not os.path.exists(js):
        with open(js, "w") as f:
            f.write(SOUP_JSON % f"soup_{n}.obj")
    return js


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--soup", type=int, action="append", default=[], help="also write soup_<N>.obj/.json")
    a = ap.parse_args()
    os.makedirs(OVERLAY, exist_ok=True)
    have_ref = copy_reference_input()
    if not have_ref and not os.path.exists(os.path.join(OVERLAY, "scene.json")):
        print("gen_assets: /root/reference/input absent and no overlay present", file=sys.stderr)
        return 1
    venus = os.path.join(OVERLAY, "venusscaled.obj")
    if not os.path.exists(venus):
        n = write_statue(venus)
        print(f"gen_assets: wrote stand-in venusscaled.obj ({n} triangles)")
    for n in a.soup:
        print("gen_assets:", ensure_soup(n))
    return 0


if __name__ == "__main__":
    sys.exit(main())

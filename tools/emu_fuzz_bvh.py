#!/usr/bin/env python3
"""emu_fuzz_bvh.py — dev / test: the GPU BVH builder (csrc/bvh_build.hip, on the kernel emulation) against the restated reference builder
(oracle/bvh_oracle.c, pinned against the reference's own trees) on random meshes chosen to be awkward for a PARALLEL builder: coordinates on a
coarse grid (equal centres, equal bin boundaries, equal SAH costs: every tie rule of bvh.c:132-287 decides), duplicates, clusters with
outliers, flat and needle-shaped extents, sizes around the phase boundaries (16, 512, chunk size). Same tree or it prints the case.

    python tools/emu_fuzz_bvh.py [--seeds A:B]
"""
import argparse, json, os, sys, time
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ap = argparse.ArgumentParser()
ap.add_argument("--seeds", default="0:10")
a = ap.parse_args()
lo, hi = (int(v) for v in a.seeds.split(":"))
os.environ["CRH_LIB"] = os.path.join(REPO, "tests", "emu", "libcray_hip_emu.so")
os.environ["CRH_ALLOW_EMULATION"] = "1"
os.environ.setdefault("HIPEMU_CUS", "4")
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "oracle"))
import subprocess
subprocess.check_call(["make", "-s", "-C", os.path.join(REPO, "oracle"), "oracle"])
import numpy as np
import oracle_py
from __graft_entry__ import load_package
pkg = load_package(); api = pkg.api
ctx = api.Context(0)
bad = 0
for seed in range(lo, hi):
    rng = np.random.default_rng(seed)
    kind = ["grid", "dups", "cluster", "flat", "needle", "uniform"][seed % 6]
    count = int(rng.choice([1, 2, 15, 16, 17, 33, 100, 511, 512, 513, 700, 2049, 5000, 9000, 20000]))
    c = rng.uniform(-1, 1, (count, 3))
    if kind == "grid": c = np.round(c * rng.choice([2, 4, 16])) / 4.0
    elif kind == "dups": c = c[rng.integers(0, max(1, count // 7), count)]
    elif kind == "cluster": c = c * 0.01; c[rng.integers(0, count, max(1, count // 50))] += rng.uniform(-1e4, 1e4, 3)
    elif kind == "flat": c[:, int(rng.integers(0, 3))] = 0.25
    elif kind == "needle": c[:, 1:] *= 1e-6
    size = rng.choice([0.0, 1e-3, 0.25])
    e = rng.uniform(-1, 1, (count, 2, 3)) * size
    if kind == "grid": e = np.round(e * 8) / 8.0
    verts = np.empty((count * 3, 3), np.float32)
    verts[0::3] = c; verts[1::3] = c + e[:, 0]; verts[2::3] = c + e[:, 1]
    polys = np.zeros((count, 10), np.int32)
    polys[:, 0] = np.arange(count) * 3; polys[:, 1] = polys[:, 0] + 1; polys[:, 2] = polys[:, 0] + 2; polys[:, 3:9] = -1
    if seed % 5 == 0: polys = polys[rng.permutation(count)]
    t0 = time.time()
    try:
        ref_nodes, ref_prims = oracle_py.bvh_build_triangles(polys.ctypes.data, verts.ctypes.data, count)
    except OverflowError:
        ref_nodes = None          # the reference's own node array overflows on this mesh (bvh.c:271): the builder must refuse it, not follow it
    try:
        nodes, prims, st = ctx.bvh_build_triangles(polys.ctypes.data, count, verts.ctypes.data, len(verts))
    except api.CrhError as e:
        nodes = None
        refused = e.code == pkg.abi.ERR_UNSUPPORTED
    if ref_nodes is None or nodes is None:
        ok = ref_nodes is None and nodes is None and refused
        bad += 0 if ok else 1
        print(json.dumps({"ok": bool(ok), "seed": seed, "kind": kind, "triangles": count, "size": float(size), "refused": True, "secs": round(time.time() - t0, 2)}), flush=True)
        continue
    leaf = ((ref_nodes[:, 7] >> 30) & 1) == 1
    ok = (nodes.shape == ref_nodes.shape and np.array_equal(nodes[:, :7], ref_nodes[:, :7]) and np.array_equal(((nodes[:, 7] >> 30) & 1) == 1, leaf)
          and np.array_equal(nodes[leaf, 7] & 0x7FFFFFFF, ref_nodes[leaf, 7] & 0x7FFFFFFF) and np.array_equal(prims, ref_prims))
    bad += 0 if ok else 1
    print(json.dumps({"ok": bool(ok), "seed": seed, "kind": kind, "triangles": count, "size": float(size), "nodes": int(ref_nodes.shape[0]), "got_nodes": int(nodes.shape[0]),
                      "secs": round(time.time() - t0, 2)}), flush=True)
ctx.close()
sys.exit(1 if bad else 0)

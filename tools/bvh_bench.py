#!/usr/bin/env python3
"""bvh_bench.py — SURVEY.md §8(f) row 1 measurement: build every mesh BVH of a scene blob on the GPU
(crh_bvh_build_triangles), check it against the BVH the reference's builder left in the blob, and time the CPU
restatement of the same builder (oracle/bvh_oracle.c, one thread, like the reference: one thread per mesh) beside it.

    python tools/bvh_bench.py --blob scenes/_built/soup_1m.blob [--tag soup_1m]
"""
import argparse, ctypes as C, json, os, sys, time
import numpy as np
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "oracle")); sys.path.insert(0, os.path.join(REPO, "tests"))
from __graft_entry__ import load_package  # noqa: E402
import oracle_py  # noqa: E402
from test_bvh_build import mesh_views, assert_same_bvh  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--blob", required=True); ap.add_argument("--tag", default=""); ap.add_argument("--no-cpu", action="store_true")
a = ap.parse_args()
pkg = load_package(); api = pkg.api
scene = api.Scene(a.blob); d = scene.desc
ctx = api.Context(0)
out = {"tag": a.tag or os.path.basename(a.blob), "meshes": []}
verts = C.cast(d.vertices, C.c_void_p).value
for m in range(int(d.mesh_count)):
    ref_nodes, ref_prims, polys, count = mesh_views(d, m)
    if count == 0:
        continue
    best = None
    for rep in range(3):
        t = time.perf_counter()
        nodes, prims, st = ctx.bvh_build_triangles(polys, count, verts, int(d.vertex_count))
        st["wall_ms"] = (time.perf_counter() - t) * 1e3
        if best is None or st["build_ms"] < best["build_ms"]:
            best = st
    assert_same_bvh(nodes, prims, ref_nodes, ref_prims, (a.blob, m))
    rec = {"mesh": m, "triangles": int(count), "nodes": int(len(nodes)), "identical_to_reference": True, "gpu": best,
           "gpu_Mtri_per_s": round(count / best["build_ms"] / 1e3, 2),
           # what the build has to touch at least once per level it is alive: box + centre + index per primitive per level
           "bytes_per_level": int(count) * (24 + 12 + 4)}
    if not a.no_cpu:
        t = time.perf_counter()
        onodes, oprims = oracle_py.bvh_build_triangles(polys, verts, count)
        rec["cpu_restatement_ms"] = round((time.perf_counter() - t) * 1e3, 1)
        rec["speedup_vs_cpu_restatement"] = round(rec["cpu_restatement_ms"] / best["build_ms"], 1)
        assert np.array_equal(oprims, prims)
    out["meshes"].append(rec)
print(json.dumps(out))

#!/usr/bin/env python3
"""make_soup_blob.py — BASELINE.json configs[4] (the synthetic N-triangle soup) as a scene blob, built WHERE IT RUNS and
without the reference's loader: the 10 M-triangle blob is 1.1 GB and does not travel to the GPU box, and the reference tree is not there.
BENCH / TEST INFRASTRUCTURE.

    python tools/make_soup_blob.py N out.blob [--builder gpu|oracle]

  vertices   tools/gen_soup.c --bin: the PCG32 triangles exactly as the reference's OBJ loader would read them from the text file (printed with
             %.7f, read back with atof) — at 1 M triangles bit-equal to the vertex buffer in scenes/_built/soup_1m.blob (the loader's);
  BLAS       crh_bvh_build_triangles (the GPU builder of the product: the reference's tree, tests/test_bvh_build.py) — or, `--builder oracle`
             (CPU tier of the tests only), the restated builder oracle/bvh_oracle.c;
  the rest   camera, materials, node graph, instance, render prefs: copied from the template blob scenes/_built/soup_1m.blob (the soups' scene
             files differ in the OBJ's name only); the one-leaf TLAS and the mesh's ray offset follow from the BLAS root box exactly as in
             instance.c:222-227 / transforms.c:86-94 / bbox.h:39-46 (identity instance transform), in float32 with the reference's operation order.
The result for N = 1 000 000 must equal the template in every section (tests/test_soup_blob.py); the 10 M blob is then the same code at another N.
"""
import ctypes as C
import os
import subprocess
import sys
import tempfile

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)


def soup_vertices(n):
    """float32 [3 n, 3]: the loader's view of the soup's vertices (gen_soup --bin, compiled on first use)."""
    exe = os.path.join(tempfile.gettempdir(), "crh_gen_soup")
    src = os.path.join(REPO, "tools", "gen_soup.c")
    if not os.path.exists(exe) or os.path.getmtime(exe) < os.path.getmtime(src):
        subprocess.check_call(["gcc", "-O2", "-fopenmp", src, "-o", exe + ".tmp"])
        os.replace(exe + ".tmp", exe)
    with tempfile.TemporaryDirectory() as tmp:
        out = os.path.join(tmp, "v.f32")
        subprocess.check_call([exe, "--bin", str(int(n)), out])
        return np.fromfile(out, np.float32).reshape(3 * n, 3)


def tlas_box_and_offset(root_bounds):
    """The instance's box (= the bounds of the one-leaf TLAS) and mesh->rayOffset from the BLAS root box {minx, maxx, miny, maxy, minz, maxz}:
    getMeshBBoxAndCenter (instance.c:222-227) with composite.A = identity: transformBBox recentres the box (transforms.c:86-94, every product with the
    identity matrix is exact), rayOffset = 0.0001f * |max - min| of the recentred box (bbox.h:39-46, vector.h: vecLength = sqrtf(x*x + y*y + z*z))."""
    f = np.float32
    b = np.asarray(root_bounds, np.float32)
    mn, mx = b[0::2], b[1::2]
    half = f(0.5)
    center = (mn + mx) * half
    ext = (mx - mn) * half
    nmn, nmx = center - ext, center + ext
    e = nmx - nmn
    length = np.sqrt(f(f(f(e[0] * e[0]) + f(e[1] * e[1])) + f(e[2] * e[2])), dtype=np.float32)
    out = np.empty(6, np.float32)
    out[0::2], out[1::2] = nmn, nmx
    return out, f(f(0.0001) * length)


def build(n, out_path, builder="gpu", template=None):
    from __graft_entry__ import load_package, BUILT
    pkg = load_package()
    api, abi = pkg.api, pkg.abi
    tpl = api.Scene(template or os.path.join(BUILT, "soup_1m.blob"))
    t = tpl.desc
    assert int(t.mesh_count) == 1 and int(t.instance_count) == 1 and int(t.tlas_node_count) == 1 and int(t.sphere_count) == 0, "template is not a soup blob"
    verts = np.ascontiguousarray(soup_vertices(n))
    polys = np.full((n, 10), -1, np.int32)                # struct poly: v[3], n[3] = -1, t[3] = -1, bits (wavefront.c:110-126: vertexCount 3, no normals, material 0)
    polys[:, 0:3] = np.arange(3 * n, dtype=np.int32).reshape(n, 3)
    polys[:, 9] = t.polys[0].bits
    stats = {}
    if builder == "gpu":
        ctx = api.Context(0)
        nodes, prims, stats = ctx.bvh_build_triangles(polys.ctypes.data, n, verts.ctypes.data, 3 * n)
        nodes, prims = nodes.copy(), prims.copy()
        ctx.close()
    else:                                                  # tests only (CPU tier): the restated reference builder
        sys.path.insert(0, os.path.join(REPO, "oracle"))
        import oracle_py
        nodes, prims = oracle_py.bvh_build_triangles(polys.ctypes.data, verts.ctypes.data, n)
    nn = len(nodes)
    all_nodes = np.zeros((nn + 1, 8), np.uint32)
    all_nodes[:nn] = nodes
    box, ray_offset = tlas_box_and_offset(nodes[0, :6].view(np.float32))
    tl = np.ctypeslib.as_array(C.cast(t.nodes, C.POINTER(C.c_uint32)), shape=(int(t.node_count), 8))[int(t.tlas_node_base)].copy()
    tl[:6] = box.view(np.uint32)
    all_nodes[nn] = tl
    all_prims = np.zeros(n + 1, np.int32)
    all_prims[:n] = prims
    all_prims[n] = t.prim_indices[int(t.tlas_prim_base)]
    mesh = abi.Mesh.from_buffer_copy(t.meshes[0])
    mesh.node_count, mesh.poly_count, mesh.ray_offset = nn, n, float(ray_offset)
    d = abi.SceneDesc.from_buffer_copy(t)
    d.nodes, d.node_count = all_nodes.ctypes.data_as(C.POINTER(abi.BvhNode)), nn + 1
    d.prim_indices, d.prim_index_count = all_prims.ctypes.data_as(C.POINTER(C.c_int32)), n + 1
    d.tlas_node_base, d.tlas_prim_base = nn, n
    d.polys, d.poly_count = polys.ctypes.data_as(C.POINTER(abi.Poly)), n
    d.vertices, d.vertex_count = verts.ctypes.data_as(C.POINTER(C.c_float)), 3 * n
    d.meshes = C.pointer(mesh)
    rc = api.library().crh_blob_save(os.fsencode(out_path), C.byref(d), C.byref(tpl.prefs))
    if rc != 0:
        raise RuntimeError(f"crh_blob_save({out_path}) failed: {rc}")
    return {"triangles": n, "nodes": nn, "bytes": os.path.getsize(out_path), "builder": builder, **{k: (round(v, 3) if isinstance(v, float) else v) for k, v in stats.items()}}


if __name__ == "__main__":
    import argparse
    import json
    ap = argparse.ArgumentParser()
    ap.add_argument("n", type=int)
    ap.add_argument("out")
    ap.add_argument("--builder", choices=("gpu", "oracle"), default="gpu")
    a = ap.parse_args()
    if a.builder == "oracle":
        os.environ.setdefault("CRH_ALLOW_EMULATION", "0")
    print(json.dumps(build(a.n, a.out, a.builder)))

"""SURVEY.md §8(f) row 1 — the binned-SAH BVH builder (src/accelerators/bvh.c:87-316).

The fixtures' scene blobs hold, for every mesh, the node array and the primitive order that the REFERENCE's own builder
produced (oracle/_ref/crh-flatten = reference loader + builder, -ffp-contract=off). A rebuilt BVH must equal them:
leaf nodes byte for byte, inner nodes in bounds / child index / leaf flag (their primCount bits are uninitialised heap
in the reference, bvh.c:219-236 never writes them), the primitive order index for index.
  not gpu: oracle/bvh_oracle.c (the CPU restatement) is pinned this way;
  gpu:     crh_bvh_build_triangles (the HIP builder) through the C-ABI.
"""
import ctypes as C
import os

import numpy as np
import pytest

CASES = ["cfg1_scene", "alphanode", "fence", "glowmetal", "refraction", "uvsphere"]
BIG = ["cfg2_hdr", "soup_1m"]        # scenes/_built (present where __graft_entry__.build() ran with the reference tree)


def mesh_views(desc, m):
    """(reference nodes uint32[n, 8], reference prim order int32[count], polys pointer, count) of mesh m."""
    mesh = desc.meshes[m]
    nodes = np.ctypeslib.as_array(C.cast(desc.nodes, C.POINTER(C.c_uint32)), shape=(int(desc.node_count), 8))
    prims = np.ctypeslib.as_array(C.cast(desc.prim_indices, C.POINTER(C.c_int32)), shape=(int(desc.prim_index_count),))
    polys = C.cast(desc.polys, C.c_void_p).value + 40 * mesh.poly_base
    n, count = mesh.node_count, mesh.poly_count
    return nodes[mesh.node_base:mesh.node_base + n], prims[mesh.prim_base:mesh.prim_base + (count if n else 0)], polys, count


def assert_same_bvh(nodes, prims, ref_nodes, ref_prims, what):
    assert nodes.shape == ref_nodes.shape, (what, nodes.shape, ref_nodes.shape)
    leaf = ((ref_nodes[:, 7] >> 30) & 1) == 1
    assert np.array_equal(((nodes[:, 7] >> 30) & 1) == 1, leaf), what
    assert np.array_equal(nodes[:, :7], ref_nodes[:, :7]), what           # bounds (bit patterns) + first child / first prim
    assert np.array_equal(nodes[leaf, 7] & 0x7FFFFFFF, ref_nodes[leaf, 7] & 0x7FFFFFFF), what
    assert np.array_equal(prims, ref_prims), what


def blob_path(name, golden_blob):
    from __graft_entry__ import BUILT
    if name in CASES:
        return golden_blob(name)
    p = os.path.join(BUILT, name + ".blob")
    if not os.path.exists(p):
        pytest.skip(f"{p} not built here")
    return p


@pytest.mark.parametrize("name", CASES + BIG)
def test_restated_builder_reproduces_the_reference_bvh(name, oracle, golden_blob):
    scene = oracle.OracleScene(blob_path(name, golden_blob))
    d = scene.desc
    built = 0
    for m in range(int(d.mesh_count)):
        ref_nodes, ref_prims, polys, count = mesh_views(d, m)
        if len(ref_nodes) == 0:
            continue
        nodes, prims = oracle.bvh_build_triangles(polys, C.cast(d.vertices, C.c_void_p).value, count)
        assert_same_bvh(nodes, prims, ref_nodes, ref_prims, (name, m))
        built += 1
    assert built > 0 or name in ("alphanode", "uvsphere", "glowmetal", "refraction", "fence")


@pytest.mark.gpu
@pytest.mark.parametrize("name", CASES + BIG)
def test_gpu_builder_reproduces_the_reference_bvh(name, pkg, golden_blob):
    if pkg.api.device_count() < 1:
        pytest.fail("no HIP device visible: the GPU tier needs one (libcray_hip has no CPU fallback)")
    scene = pkg.api.Scene(blob_path(name, golden_blob))
    d = scene.desc
    ctx = pkg.api.Context(0)
    for m in range(int(d.mesh_count)):
        ref_nodes, ref_prims, polys, count = mesh_views(d, m)
        nodes, prims, st = ctx.bvh_build_triangles(polys, count, C.cast(d.vertices, C.c_void_p).value, int(d.vertex_count))
        if len(ref_nodes) == 0:
            assert len(nodes) == 0
            continue
        assert_same_bvh(nodes, prims, ref_nodes, ref_prims, (name, m, st))
    ctx.close()


@pytest.mark.gpu
def test_gpu_builder_degenerate_inputs(pkg, oracle):
    """Coincident triangles (every centre in one bin, hi == lo on every axis: the reference splits nothing off until
    the depth limit), signed zeros in the bounds, a single triangle, and 600 triangles in a row (median fallback)."""
    ctx = pkg.api.Context(0)
    rng = np.random.default_rng(7)
    cases = []
    one = np.array([[0, 0, 0], [1, 0, 0], [0, 1, 0]], np.float32)
    cases.append(("single", one, 1))
    cases.append(("coincident", one, 700))
    z = np.array([[-0.0, 0.0, 1], [0.0, -0.0, 2], [-0.0, -0.0, 3], [0.0, 0.0, 4]], np.float32)
    cases.append(("signed zeros", z, 40))
    for label, base, count in cases:
        nv = len(base)
        verts = np.ascontiguousarray(base, np.float32)
        polys = np.zeros((count, 10), np.int32)
        for i in range(count):
            polys[i, 0:3] = [(i + k) % nv for k in range(3)] if label != "coincident" else [0, 1, 2]
            polys[i, 3:9] = -1
        ref_nodes, ref_prims = oracle.bvh_build_triangles(polys.ctypes.data, verts.ctypes.data, count)
        nodes, prims, st = ctx.bvh_build_triangles(polys.ctypes.data, count, verts.ctypes.data, nv)
        assert_same_bvh(nodes, prims, ref_nodes, ref_prims, (label, st))
    # a line of tiny triangles along x with two far outliers: SAH keeps losing to the leaf cost, median fallback splits
    count = 3000
    verts = np.zeros((count * 3, 3), np.float32)
    xs = np.sort(rng.uniform(0, 1, count)).astype(np.float32)
    xs[-1] = 1e6
    for i in range(count):
        verts[3 * i] = [xs[i], 0, 0]; verts[3 * i + 1] = [xs[i] + 1e-4, 1e-4, 0]; verts[3 * i + 2] = [xs[i], 0, 1e-4]
    polys = np.zeros((count, 10), np.int32)
    polys[:, 0] = np.arange(count) * 3; polys[:, 1] = polys[:, 0] + 1; polys[:, 2] = polys[:, 0] + 2; polys[:, 3:9] = -1
    ref_nodes, ref_prims = oracle.bvh_build_triangles(polys.ctypes.data, verts.ctypes.data, count)
    nodes, prims, st = ctx.bvh_build_triangles(polys.ctypes.data, count, verts.ctypes.data, len(verts))
    assert_same_bvh(nodes, prims, ref_nodes, ref_prims, ("outlier line", st))
    # a polygon that names a vertex the mesh does not have (checked where the boxes are made, on the device, since round 4): refused, and the context builds the next mesh
    for bad in (len(verts), -1):
        broken = polys.copy()
        broken[1234, 1] = bad
        with pytest.raises(pkg.api.CrhError) as e:
            ctx.bvh_build_triangles(broken.ctypes.data, count, verts.ctypes.data, len(verts))
        assert e.value.code == pkg.abi.ERR_INVALID and "vertex index out of range" in str(e.value)
    nodes, prims, st = ctx.bvh_build_triangles(polys.ctypes.data, count, verts.ctypes.data, len(verts))
    assert_same_bvh(nodes, prims, ref_nodes, ref_prims, ("after a refused mesh", st))
    ctx.close()


@pytest.mark.gpu
def test_gpu_builder_refuses_a_mesh_the_reference_overflows_on(pkg, oracle):
    """bvh.c:271 allocates 2 n - 1 nodes, and bvh.c:220 splits a node whose primitives all land on the left: clusters of more than 16
    coincident primitives become chains of (all | none) splits down to the depth limit, and enough of them need more nodes than that — the
    reference writes past its heap array (found by tools/emu_fuzz_bvh.py). No reference tree exists; the builder must say so
    (CRH_ERR_UNSUPPORTED) with its memory intact, and build the next mesh correctly."""
    rng = np.random.default_rng(18)
    count = 9000
    c = np.round(rng.uniform(-1, 1, (count, 3)) * 4) / 4.0                 # 729 distinct centres, zero-size triangles
    verts = np.repeat(c, 3, axis=0).astype(np.float32)
    polys = np.zeros((count, 10), np.int32)
    polys[:, 0] = np.arange(count) * 3; polys[:, 1] = polys[:, 0] + 1; polys[:, 2] = polys[:, 0] + 2; polys[:, 3:9] = -1
    with pytest.raises(OverflowError):
        oracle.bvh_build_triangles(polys.ctypes.data, verts.ctypes.data, count)
    ctx = pkg.api.Context(0)
    try:
        with pytest.raises(pkg.api.CrhError) as e:
            ctx.bvh_build_triangles(polys.ctypes.data, count, verts.ctypes.data, len(verts))
        assert e.value.code == pkg.abi.ERR_UNSUPPORTED and "2 n - 1" in str(e.value)
        good = count // 3                                                    # the same context builds a sane mesh afterwards
        verts2 = rng.uniform(-1, 1, (good * 3, 3)).astype(np.float32)
        polys2 = np.ascontiguousarray(polys[:good])
        ref_nodes, ref_prims = oracle.bvh_build_triangles(polys2.ctypes.data, verts2.ctypes.data, good)
        nodes, prims, st = ctx.bvh_build_triangles(polys2.ctypes.data, good, verts2.ctypes.data, len(verts2))
        assert_same_bvh(nodes, prims, ref_nodes, ref_prims, ("after a refused mesh", st))
    finally:
        ctx.close()

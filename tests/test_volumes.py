"""participating media — newSphereVolume / newMeshVolume instances with an isotropic phase function
(/root/reference/src/datatypes/instance.c:62-92, 187-216; src/nodes/shaders/isotropic.c:40-47). No scene JSON builds them, so the
fixture comes from the real reference with the instances converted by its own C constructors (oracle/ref_node_patch.c:
CRH_NODE_PATCH=volumes): two sphere volumes (one empty, one with solid spheres inside and poking through), a cube-shaped mesh volume
instanced twice (a solid sphere inside one), solid geometry and an HDR environment around them, 8 spp, 12 bounces.

What makes volumes special on the device: the intersection draws from the PATH's sampler inside the traversal, and a mesh volume
needs two BLAS walks per visit (entry, then exit from just behind the entry point) bounded by the closest hit so far."""
import ctypes as C

import numpy as np
import pytest

from conftest import camera_rays, image_stats, kernel_forms


def test_oracle_bit_exact_on_volumes(oracle, manifest, golden_blob, golden_ref):
    m = manifest["volumes"]
    scene = oracle.OracleScene(golden_blob("volumes"))
    kinds = sorted(scene.desc.instances[i].kind for i in range(scene.desc.instance_count))
    assert kinds.count(2) == 2 and kinds.count(3) == 2, kinds          # CRH_INSTANCE_SPHERE_VOLUME / MESH_VOLUME
    img, cnt = oracle.render(scene, m["width"], m["height"], m["samples"], m["bounces"])
    ref = golden_ref("volumes")
    assert np.array_equal(img.view(np.uint32), ref.view(np.uint32)), f"{(img != ref).sum()} floats differ"
    assert cnt["rays"] == m["rays"] and cnt["node_tests"] == m["node_tests"] and cnt["tri_tests"] == m["tri_tests"]


@pytest.mark.parametrize("shape,chunk", [((8, 8), 64), ((2, 2), 3)])
def test_emulated_kernel_bit_exact_on_volumes(shape, chunk, emu, oracle, manifest, golden_blob, golden_ref):
    """The device lane code (walk state machine with the two volume phases, sampler draw through the path port) built for the host."""
    from test_emu_parity import emu_render
    m = manifest["volumes"]
    scene = oracle.OracleScene(golden_blob("volumes"))
    fb, cnt, _ = emu_render(emu, oracle, scene, m["width"], m["height"], m["samples"], m["bounces"], shape=shape, chunk=chunk)
    ref = golden_ref("volumes")
    assert np.array_equal(fb.view(np.uint32), ref.view(np.uint32)), f"{(fb != ref).sum()} floats differ"
    assert cnt["rays"] == m["rays"] and cnt["tri_tests"] <= m["tri_tests"] and cnt["tri_tests"] >= 0.98 * m["tri_tests"]


def test_caller_rays_are_refused_on_scenes_with_volumes(emu, oracle, golden_blob):
    """getClosestIsect needs the path's sampler once a volume is in the scene (instance.c:74, 199): crh_trace_rays has none."""
    scene = oracle.OracleScene(golden_blob("volumes"))
    rays = camera_rays(scene.desc, 16, 1)
    with pytest.raises(RuntimeError):
        oracle.trace_rays(scene, rays)
    he = np.zeros(len(rays), dtype=oracle.abi.HIT_DTYPE)
    assert emu.emu_trace_rays(scene.ptr, rays.ctypes.data, len(rays), he.ctypes.data) == oracle.abi.ERR_UNSUPPORTED


@pytest.mark.gpu
def test_gpu_volumes_vs_reference(pkg, manifest, golden_blob, golden_ref):
    if pkg.api.device_count() < 1:
        pytest.fail("GPU tier needs a HIP device; libcray_hip has no CPU fallback")
    m = manifest["volumes"]
    w, h, s, b = m["width"], m["height"], m["samples"], m["bounces"]
    ctx = pkg.api.Context(0)
    try:
        ctx.upload(pkg.api.Scene(golden_blob("volumes")))
        fb = ctx.framebuffer(w, h)
        frames = []
        forms = kernel_forms(pkg, ctx)          # the product library: the rolling kernel; the emulation tier: all three forms
        for kern in forms:
            ctx.set_option(pkg.abi.OPT_KERNEL, kern)
            ctx.clear(fb, w, h)
            ctx.reset_counters()
            ctx.render_region(fb, w, h, s, b)
            frames.append((ctx.download(fb, w, h), ctx.counters()))
        with pytest.raises(pkg.api.CrhError) as e:
            ctx.trace_rays(np.zeros((4, 6), np.float32) + 1.0)
        assert e.value.code == pkg.abi.ERR_UNSUPPORTED
    finally:
        ctx.close()
    img, cnt = frames[0]
    for other, other_cnt in frames[1:]:        # every kernel form the library holds: the same frame
        assert np.array_equal(img, other) and cnt == other_cnt
    ref = golden_ref("volumes")
    assert np.array_equal(img.view(np.uint32), ref.view(np.uint32)), image_stats(img, ref)        # logf of the free-flight draw included
    assert cnt["rays"] == m["rays"], (cnt["rays"], m["rays"])

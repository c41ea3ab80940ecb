"""CPU tier for the KERNEL LOGIC: the lane-level device code (c-ray_amd/csrc/pt_device.h) and the product's
scene compiler, built for the host (tests/emu), must match the oracle bit for bit — same traversal order,
RNG draw order, node programs, block/chunk schedule and fold order as the reference."""
import ctypes as C

import numpy as np
import pytest

from conftest import BIG_CASES, built_blob, camera_rays, resize_camera

CASES = ["cfg1_scene", "alphanode", "fence", "glowmetal", "refraction", "uvsphere"]


def emu_render(emu, oracle, scene, w, h, s, b, region=None, shape=(8, 8), chunk=64):
    abi = oracle.abi
    x0, y0, x1, y1 = region or (0, 0, w, h)
    fb = np.zeros((h, w, 3), np.float32)
    cnt, hi = abi.Counters(), C.c_uint32()
    p = abi.RenderParams(x0, y0, x1, y1, w, h, 0, s, s, b)
    rc = emu.emu_render_region(scene.ptr, C.byref(p), fb.ctypes.data, C.byref(cnt), C.byref(hi), shape[0], shape[1], chunk)
    assert rc == 0, emu.emu_last_error()
    return fb, cnt.as_dict(), hi.value


@pytest.mark.parametrize("name", CASES)
def test_emulated_kernel_bit_exact_vs_reference(name, emu, oracle, manifest, golden_blob, golden_ref):
    m = manifest[name]
    scene = oracle.OracleScene(golden_blob(name))
    fb, cnt, high = emu_render(emu, oracle, scene, m["width"], m["height"], m["samples"], m["bounces"])
    ref = golden_ref(name)
    assert np.array_equal(fb.view(np.uint32), ref.view(np.uint32)), f"{name}: {(fb != ref).sum()} floats differ"
    assert cnt["rays"] == m["rays"] and cnt["paths"] == m["width"] * m["height"] * m["samples"]
    # exact node / triangle visit counts too, unless a zero-component ray took the exact-slab path (fewer visits)
    assert cnt["node_tests"] <= m["node_tests"] and cnt["tri_tests"] <= m["tri_tests"]
    assert cnt["node_tests"] >= 0.98 * m["node_tests"]
    mx = C.c_uint32()
    assert emu.emu_compile_check(scene.ptr, C.byref(mx), None, None) == 0
    assert high <= mx.value, "traversal stack bound computed by the scene compiler was exceeded"


@pytest.mark.parametrize("name", BIG_CASES)
def test_emulated_kernel_bit_exact_on_baseline_configs_reduced_frame(name, emu, oracle, manifest, golden_ref):
    """BASELINE.json configs[1..4] at 320x180 through the device lane code built for the host: the reference's frame bit for bit."""
    m = manifest[name]
    scene = resize_camera(oracle.OracleScene(built_blob(m["built_blob"])), m["width"], m["height"])
    fb, cnt, _ = emu_render(emu, oracle, scene, m["width"], m["height"], m["samples"], m["bounces"])
    ref = golden_ref(name)
    assert np.array_equal(fb.view(np.uint32), ref.view(np.uint32)), f"{name}: {(fb != ref).sum()} floats differ"
    assert cnt["rays"] == m["rays"] and cnt["node_tests"] <= m["node_tests"] and cnt["node_tests"] >= 0.98 * m["node_tests"]


@pytest.mark.parametrize("name", ["cfg1_scene", "fence", "refraction", "uvsphere"])
def test_wide_walk_lane_code_renders_the_fixtures(name, emu, oracle, manifest, golden_blob, golden_ref):
    """CRH_OPT_WALK = CRH_WALK_WIDE4 (round 5, an option; the binary walk is the contract): the scene compiler's 4-ary copy of the BVHs and pt_device.h's wide node step,
    built for the host — these fixtures' frames are the reference's bit for bit (no ray of theirs meets a tie or a near tie: tools/wide_walk_study.py counts where others do),
    with the same rays, at most 0.62 of the binary walk's node steps, and a stack that stays inside the wide walk's own bound."""
    m = manifest[name]
    scene = oracle.OracleScene(golden_blob(name))
    emu.emu_set_walk.argtypes = [C.c_int]
    emu.emu_set_walk(1)
    try:
        fb, cnt, high = emu_render(emu, oracle, scene, m["width"], m["height"], m["samples"], m["bounces"])
    finally:
        emu.emu_set_walk(0)
    assert np.array_equal(fb.view(np.uint32), golden_ref(name).view(np.uint32)), f"{name}: {(fb != golden_ref(name)).sum()} floats differ"
    assert cnt["rays"] == m["rays"]
    if m["node_tests"] > 100 * m["rays"] // 10:          # (scenes with a BVH worth the name)
        assert cnt["node_tests"] / 4 < 0.62 * m["node_tests"] / 2
    assert high <= 134, "deeper than the device's stack (12 LDS entries + 122 in the overflow columns)"


def test_emulated_kernel_interactive_mode_bit_exact(emu, oracle, manifest, golden_blob, golden_ref):
    """The Halton-sampler instantiation of the lane code (CRH_OPT_SAMPLER = CRH_SAMPLER_HALTON) against the reference's
    --iterative -j 1 frame: passes 1 .. samples-1, any chunking."""
    m = manifest["cfg1_scene_iterative"]
    scene = oracle.OracleScene(golden_blob(m["blob"]))
    emu.emu_set_sampler(1)
    try:
        p = oracle.abi.RenderParams(0, 0, m["width"], m["height"], m["width"], m["height"], 0, m["passes"], m["samples"], m["bounces"])
        fb = np.zeros((m["height"], m["width"], 3), np.float32)
        cnt, hi = oracle.abi.Counters(), C.c_uint32()
        assert emu.emu_render_region(scene.ptr, C.byref(p), fb.ctypes.data, C.byref(cnt), C.byref(hi), 8, 8, 2) == 0, emu.emu_last_error()
    finally:
        emu.emu_set_sampler(0)
    ref = golden_ref("cfg1_scene_iterative")
    assert np.array_equal(fb.view(np.uint32), ref.view(np.uint32)), f"{(fb != ref).sum()} floats differ"


@pytest.mark.parametrize("shape,chunk", [((8, 8), 3), ((4, 4), 1), ((2, 2), 64), ((16, 16), 2), ((1, 1), 4)])
def test_block_schedule_is_order_independent(shape, chunk, emu, oracle, manifest, golden_blob, golden_ref):
    """Any block shape / pass chunking folds the samples in pass order -> identical frame (ragged edges included)."""
    m = manifest["fence"]
    scene = oracle.OracleScene(golden_blob("fence"))
    fb, _, _ = emu_render(emu, oracle, scene, m["width"], m["height"], m["samples"], m["bounces"], shape=shape, chunk=chunk)
    assert np.array_equal(fb, golden_ref("fence"))


@pytest.mark.parametrize("name", ["cfg1_scene", "refraction", "uvsphere"])
def test_trace_rays_records_identical(name, emu, oracle, golden_blob):
    scene = oracle.OracleScene(golden_blob(name))
    rays = camera_rays(scene.desc, 20000, 3)
    ho = oracle.trace_rays(scene, rays)
    he = np.zeros(len(rays), dtype=oracle.abi.HIT_DTYPE)
    assert emu.emu_trace_rays(scene.ptr, rays.ctypes.data, len(rays), he.ctypes.data) == 0
    assert ho.tobytes() == he.tobytes()
    assert (ho["inst"] >= 0).sum() > 500


def test_zero_component_rays_same_hit_fewer_visits(emu, oracle, golden_blob):
    """Rays with an exactly zero direction component: the reference's NaN slab arithmetic visits (nearly) every
    node; the device code tests that slab exactly. Same hit record, never more node visits."""
    scene = oracle.OracleScene(golden_blob("cfg1_scene"))
    rays = camera_rays(scene.desc, 3000, 11)
    rays[0::3, 3] = 0.0
    rays[1::3, 4] = 0.0
    rays[2::3, 5] = 0.0
    ho = oracle.trace_rays(scene, rays)
    he = np.zeros(len(rays), dtype=oracle.abi.HIT_DTYPE)
    assert emu.emu_trace_rays(scene.ptr, rays.ctypes.data, len(rays), he.ctypes.data) == 0
    for f in ("inst", "poly", "distance", "uv", "point", "normal", "material"):
        assert np.array_equal(ho[f], he[f]), f
    assert (he["node_tests"] <= ho["node_tests"]).all() and he["node_tests"].sum() < ho["node_tests"].sum()


def test_scene_compiler_rejects_malformed_scenes(emu, oracle, golden_blob):
    abi = oracle.abi
    scene = oracle.OracleScene(golden_blob("fence"))
    d = scene.desc

    def check(expect):
        rc = emu.emu_compile_check(scene.ptr, None, None, None)
        assert rc == expect, (rc, emu.emu_last_error())

    check(0)
    old = d.abi_version; d.abi_version = 99; check(abi.ERR_INVALID); d.abi_version = old
    old = d.background; d.background = 0; check(abi.ERR_INVALID); d.background = old
    old = d.nodes[0].first; d.nodes[0].first = 10 ** 6; check(abi.ERR_INVALID); d.nodes[0].first = old
    old = d.polys[0].v[0]; d.polys[0].v[0] = -5; check(abi.ERR_INVALID); d.polys[0].v[0] = old
    old = d.instances[0].kind; d.instances[0].kind = 4; check(abi.ERR_UNSUPPORTED); d.instances[0].kind = old      # 2 / 3 are the volume kinds
    old = d.materials[0].bsdf; d.materials[0].bsdf = abi.NODE_NONE; check(abi.ERR_INVALID); d.materials[0].bsdf = old
    check(0)


def test_shade_classes_follow_the_shading_code_path(emu, oracle):
    """Scene compiler: instances get a shade class by what their hits branch on — sphere or mesh, the bsdf kinds in the material graph,
    uv use, emission. hdr.json: the three metal spheres share a class, the four glass spheres another, the three lights a third, the plastic
    sphere, the textured ground mesh and the plastic mesh one each (six); a one-material soup has a single class (the kernel then skips
    the per-class bookkeeping, as it does below four classes)."""
    scene = oracle.OracleScene(built_blob("cfg2_hdr"))
    n = int(scene.desc.instance_count)
    cls = (C.c_uint32 * n)()
    emu.emu_shade_classes.argtypes = [C.POINTER(oracle.abi.SceneDesc), C.POINTER(C.c_uint32), C.c_uint64]
    count = emu.emu_shade_classes(scene.ptr, cls, n)
    assert count == 6 and max(cls) == 5
    cls = list(cls)
    # input/hdr.json order: spheres metal, plastic, metal, glass x4, metal, emissive x3, then the two meshes
    assert cls[0] == cls[2] == cls[7] and cls[3] == cls[4] == cls[5] == cls[6] and cls[8] == cls[9] == cls[10]
    assert len({cls[0], cls[1], cls[3], cls[8], cls[11], cls[12]}) == 6
    soup = oracle.OracleScene(built_blob("soup_1m"))
    one = (C.c_uint32 * int(soup.desc.instance_count))()
    assert emu.emu_shade_classes(soup.ptr, one, len(one)) == 1 and list(one) == [0]

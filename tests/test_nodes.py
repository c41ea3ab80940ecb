"""Table N rows a12 / a13 / a14: the nodes no scene JSON reaches — math (15 ops), vecMath (10 ops), fresnel, rayLength,
normal, combineValue, combineRGB, vecToColor, isotropic, nested add — plus the JSON-reachable ones no other fixture uses (add, checker on
a mesh without texture coordinates, blackbody, grayscale(image), NO_BILINEAR fetches).

Fixtures: tests/golden/nodezoo*.{blob,ref.f32}.gz, rendered by the REAL reference with the graphs built by its own C constructors
(oracle/ref_node_patch.c, tools/gen_golden.py). `nodezoo_display` (1 spp, 2 bounces, white background) shows each graph's value as the
colour of one sphere: spheres 0..8 / 18..27 hold the known answers of /root/reference/tests/test_nodes.h as constant graphs (folded by
the scene compiler), 9..17 / 28..37 the same graphs with hit-dependent operands (evaluated per hit by the device VM). `nodezoo` (4 spp,
6 bounces) runs the exotic and JSON graphs through whole paths.

CPU tier: oracle and host-built lane code equal the reference bit for bit. GPU tier: bit for bit as well (c-ray_amd/csrc/exact_math.h);
the Math node's Tangent op included."""
import ctypes as C
import math

import numpy as np
import pytest

from conftest import image_stats

ZOO_COLS, ZOO_ROWS, ZOO_PITCH = 10, 6, 0.1          # tools/gen_golden.py: write_nodezoo_scene()
W, H, FOV, CAM_Z = 320, 192, 28.0, 3.0
PI32 = float(np.float32(math.pi))
H32 = float(np.sqrt(np.float32(0.5)))

# /root/reference/tests/test_nodes.h — (expected r, g, b), exact unless listed in ROUGH
MATH_EXPECT = [
    (256.0, 0.0, 0.0),             # :26-40 add 128+128, -128+128; :42-56 subtract 128-128
    (-256.0, 16384.0, 1.0),        # subtract -128-128; :58-66 multiply; :68-82 divide 128/128
    (-128.0, 65536.0, -128.0),     # divide -128/1; :84-103 power 2^16, (-128)^1
    (1.0, 0.0, 1.0),               # power 128^0; :105-129 log10 1, 10
    (2.0, 3.0, 4.0),               # log10 100, 1000, 10000
    (3.0, 128.0, 128.0),           # :131-139 sqrt 9; :141-153 abs
    (-128.0, 42.0, 128.0),         # :155-169 min; :171-185 max -128,128
    (128.0, 0.0, -1.0),            # max 128,42; :188-196 sin(pi) ~ 0; :198-206 cos(pi) ~ -1
    (0.0, PI32, 180.0),            # :208-216 tan(pi) ~ 0; :218-226 toRadians(180) ~ pi; :228-236 toDegrees(pi) ~ 180
]
VEC_EXPECT = [
    (2.0, 4.0, 6.0), (0.0, 0.0, 0.0), (1.0, 4.0, 9.0), (2.5, 2.5, 2.5),     # :245-303 add, subtract, multiply, average
    (0.0, 0.0, 0.0),                                                         # :305-326 dot: the float result is not the vector
    (0.0, 0.0, 1.0),                                                         # :328-341 cross
    tuple(float(v) for v in (np.float32([1, 2, 3]) / np.sqrt(np.float32(14.0)))),   # :343-353 normalize
    (H32, -H32, 0.0),                                                        # :355-371 reflect
    (0.0, 0.0, 0.0),                                                         # :373-383 length
    (10.0, 2.0, 3.0),                                                        # :385-395 abs
]
EXPECT = {i: v for i, v in enumerate(MATH_EXPECT)}
EXPECT.update({9 + i: v for i, v in enumerate(MATH_EXPECT)})
EXPECT.update({18 + i: v for i, v in enumerate(VEC_EXPECT)})
EXPECT.update({28 + i: v for i, v in enumerate(VEC_EXPECT)})
LIBM = {2, 3, 4, 7, 8}            # math spheres whose values go through powf / log10f / sinf / cosf / tanf (roughly_equals in the reference's tests too)


def sphere_pixel(i):
    """Image (row, col) of the centre of nodezoo sphere i (stored rows run top-down: texture.c:24-28)."""
    col, row = i % ZOO_COLS, i // ZOO_COLS
    x, y = (col - (ZOO_COLS - 1) / 2) * ZOO_PITCH, ((ZOO_ROWS - 1) / 2 - row) * ZOO_PITCH
    pix = 2.0 * math.tan(math.radians(FOV) / 2.0) / W * CAM_Z
    return int(round(H / 2 - y / pix - 0.5)), int(round(x / pix + W / 2 - 0.5))


def sphere_value(img, i):
    """Colour shown by sphere i: per-channel median of the 5 x 5 pixels around its centre (a bounce that hits a neighbour is black)."""
    r, c = sphere_pixel(i)
    return np.median(img[r - 2:r + 3, c - 2:c + 3].reshape(-1, 3), axis=0)


def check_known_answers(img):
    """Exact for everything IEEE arithmetic decides; the reference's own `roughly_equals` for the values that pass through libm."""
    for i, want in EXPECT.items():
        got = sphere_value(img, i)
        if i < 18 and (i % 9) in LIBM:
            assert np.allclose(got, want, rtol=1e-6, atol=2e-7), (i, got, want)
        else:
            assert np.array_equal(got, np.float32(want)), (i, got, want)


def test_reference_fixture_shows_the_reference_tests_known_answers(golden_ref):
    """The fixture itself (the real reference's render) reproduces tests/test_nodes.h."""
    check_known_answers(golden_ref("nodezoo_display"))


@pytest.mark.parametrize("name", ["nodezoo_display", "nodezoo"])
def test_oracle_bit_exact_on_node_zoo(name, oracle, manifest, golden_blob, golden_ref):
    m = manifest[name]
    scene = oracle.OracleScene(golden_blob(m["blob"]))
    img, cnt = oracle.render(scene, m["width"], m["height"], m["samples"], m["bounces"])
    ref = golden_ref(name)
    assert np.array_equal(img.view(np.uint32), ref.view(np.uint32)), f"{name}: {(img != ref).sum()} floats differ"
    assert cnt["rays"] == m["rays"] and cnt["node_tests"] == m["node_tests"] and cnt["tri_tests"] == m["tri_tests"]


@pytest.mark.parametrize("name", ["nodezoo_display", "nodezoo"])
def test_emulated_kernel_bit_exact_on_node_zoo(name, emu, oracle, manifest, golden_blob, golden_ref):
    """The device lane code + the product's scene compiler (constant folding, operand kinds, postfix programs) built for the host."""
    from test_emu_parity import emu_render
    m = manifest[name]
    scene = oracle.OracleScene(golden_blob(m["blob"]))
    fb, cnt, _ = emu_render(emu, oracle, scene, m["width"], m["height"], m["samples"], m["bounces"])
    ref = golden_ref(name)
    assert np.array_equal(fb.view(np.uint32), ref.view(np.uint32)), f"{name}: {(fb != ref).sum()} floats differ"
    assert cnt["rays"] == m["rays"]


def test_scene_compiler_folds_constant_graphs_and_keeps_dynamic_ones(emu, oracle, golden_blob):
    """Spheres 0..8 / 18..27 are constant graphs: the compiler folds them (no program); 9..17 / 28..37 must stay programs."""
    scene = oracle.OracleScene(golden_blob("nodezoo_display"))
    nprog = C.c_uint32()
    assert emu.emu_compile_check(scene.ptr, None, C.byref(nprog), None) == 0, emu.emu_last_error()
    assert nprog.value >= 19, nprog.value


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["nodezoo_display", "nodezoo"])
def test_gpu_node_zoo_vs_reference(name, pkg, manifest, golden_blob, golden_ref):
    if pkg.api.device_count() < 1:
        pytest.fail("GPU tier needs a HIP device; libcray_hip has no CPU fallback")
    m = manifest[name]
    w, h, s, b = m["width"], m["height"], m["samples"], m["bounces"]
    ctx = pkg.api.Context(0)
    try:
        ctx.upload(pkg.api.Scene(golden_blob(m["blob"])))
        fb = ctx.framebuffer(w, h)
        ctx.reset_counters()
        ctx.render_region(fb, w, h, s, b)
        img, cnt = ctx.download(fb, w, h), ctx.counters()
    finally:
        ctx.close()
    ref = golden_ref(name)
    assert cnt["rays"] == m["rays"], (cnt["rays"], m["rays"])
    if name == "nodezoo":
        assert np.array_equal(img.view(np.uint32), ref.view(np.uint32)), image_stats(img, ref)       # every exotic / JSON graph, whole paths: bit-exact
    else:
        check_known_answers(img)
        assert np.array_equal(img.view(np.uint32), ref.view(np.uint32)), image_stats(img, ref)

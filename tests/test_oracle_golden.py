"""Pins the oracle (oracle/cray_oracle.c): its render buffer must equal, BIT FOR BIT, the buffer the real
reference (c-ray-ref-strict = the unmodified reference sources) produced for every golden case, and its
ray / node-test / triangle-test counters must equal the instrumented reference's (c-ray-ref-count)."""
import numpy as np
import pytest

from conftest import BIG_CASES, built_blob, resize_camera

CASES = ["cfg1_scene", "alphanode", "fence", "glowmetal", "refraction", "uvsphere"]


@pytest.mark.parametrize("name", CASES)
def test_oracle_bit_exact_vs_reference(name, oracle, manifest, golden_blob, golden_ref):
    m = manifest[name]
    scene = oracle.OracleScene(golden_blob(name))
    img, cnt = oracle.render(scene, m["width"], m["height"], m["samples"], m["bounces"])
    ref = golden_ref(name)
    assert np.array_equal(img.view(np.uint32), ref.view(np.uint32)), f"{name}: {(img != ref).sum()} floats differ"
    assert cnt["rays"] == m["rays"] and cnt["node_tests"] == m["node_tests"] and cnt["tri_tests"] == m["tri_tests"]
    assert cnt["paths"] == m["width"] * m["height"] * m["samples"]


def test_oracle_interactive_mode_bit_exact_vs_reference(oracle, manifest, golden_blob, golden_ref):
    """renderThreadInteractive (renderer.c:184-250) = Halton sampler seeded with
    state.finishedPasses (1-based), passes 1 .. samples-1. Fixture: c-ray-ref-strict --iterative -j 1."""
    m = manifest["cfg1_scene_iterative"]
    scene = oracle.OracleScene(golden_blob(m["blob"]))
    img, cnt = oracle.render(scene, m["width"], m["height"], m["samples"], m["bounces"], pass_count=m["passes"], halton=True)
    ref = golden_ref("cfg1_scene_iterative")
    assert np.array_equal(img.view(np.uint32), ref.view(np.uint32)), f"{(img != ref).sum()} floats differ"
    assert cnt["paths"] == m["width"] * m["height"] * m["passes"]
    # progressive display: pass by pass into the same buffer is the same image
    fb = np.zeros_like(img)
    for p in range(m["passes"]):
        oracle.render(scene, m["width"], m["height"], m["samples"], m["bounces"], first_pass=p, pass_count=1, fb=fb, halton=True)
    assert np.array_equal(fb, ref)


def test_oracle_region_and_pass_splits_compose(oracle, manifest, golden_blob, golden_ref):
    """Tiles are disjoint and passes fold in order: any tiling / pass split reproduces the frame exactly."""
    m = manifest["fence"]
    w, h, s, b = m["width"], m["height"], m["samples"], m["bounces"]
    scene = oracle.OracleScene(golden_blob("fence"))
    fb = np.zeros((h, w, 3), np.float32)
    for (x0, y0, x1, y1) in [(0, 0, 70, 33), (70, 0, w, 33), (0, 33, w, h)]:
        oracle.render(scene, w, h, s, b, region=(x0, y0, x1, y1), pass_count=1, fb=fb)
        oracle.render(scene, w, h, s, b, region=(x0, y0, x1, y1), first_pass=1, fb=fb)
    assert np.array_equal(fb, golden_ref("fence"))


def test_sampler_known_answers(oracle):
    """PCG32 / hash64 seeding (sampler.c:41-44): first draws of stream (pixel 0, pass 0) and the 32-bit key wrap."""
    d = oracle.sampler_draws(0, 0, 4, 4)
    assert ((d >= 0) & (d <= 1)).all()
    # key = pixel * maxPasses + pass wraps in 32 bits: (2^31, 2) collides with (0, 0)
    assert np.array_equal(oracle.sampler_draws(2 ** 31, 0, 2, 8), oracle.sampler_draws(0, 0, 2, 8))
    assert not np.array_equal(oracle.sampler_draws(1, 0, 2, 8), oracle.sampler_draws(0, 0, 2, 8))


def test_srgb8_truncates(oracle):
    fb = np.array([[[0.0, 0.0031308, 1.0], [0.5, 2.0, 0.2]]], np.float32)
    out = oracle.to_srgb8(fb)
    assert out.tolist() == [[[0, 10, 254], [187, 255, 123]]]   # 1.055 * 1 - 0.055 = 0.99999994 -> 254: truncation, texture.c:18-22


@pytest.mark.parametrize("name", BIG_CASES)
def test_oracle_bit_exact_on_baseline_configs_reduced_frame(name, oracle, manifest, golden_ref):
    """BASELINE.json configs[1..4] (HDR environment + DOF, deep BLAS at 32 bounces, instanced TLAS, triangle soup) at 320x180:
    the restatement equals c-ray-ref-strict bit for bit, counters equal c-ray-ref-count."""
    m = manifest[name]
    scene = resize_camera(oracle.OracleScene(built_blob(m["built_blob"])), m["width"], m["height"])
    img, cnt = oracle.render(scene, m["width"], m["height"], m["samples"], m["bounces"])
    ref = golden_ref(name)
    assert np.array_equal(img.view(np.uint32), ref.view(np.uint32)), f"{name}: {(img != ref).sum()} floats differ"
    assert cnt["rays"] == m["rays"] and cnt["node_tests"] == m["node_tests"] and cnt["tri_tests"] == m["tri_tests"]

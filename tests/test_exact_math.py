"""c-ray_amd/csrc/exact_math.h — the libm functions of the hot path restated with the bits of the reference's host libm (glibc 2.35,
x86-64 FMA ifunc variants): sinf, cosf (+ the shared-reduction sincosf), tanf, powf, logf, log10f, atanf, atan2f, acosf, asinf.

CPU tier: the host build of the header against the installed libm, bit for bit (tests/emu/exact_math_check.cpp). By default every
4099th float per unary function / per powf exponent of the path plus 2 M random pairs (a second); CRH_EXACT_MATH_FULL=1 runs all 2^32
inputs per function and 3e8 random pairs (~10 min on 8 cores) — that run was clean when the header was written and after every change.
GPU tier: the DEVICE build of the same header (crh_debug_eval_math) against the host libm on edge values, dense samples of the ranges
the path feeds each function, and random bit patterns. NaN results compare equal to NaN (payload / sign of a NaN is not part of the bar).
"""
import math
import os
import subprocess

import numpy as np
import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_host_build_matches_the_installed_libm(tmp_path):
    exe = str(tmp_path / "exact_math_check")
    subprocess.check_call(["g++", "-std=c++17", "-O2", "-march=x86-64-v3", "-ffp-contract=off", "-fopenmp",
                           os.path.join(REPO, "tests", "emu", "exact_math_check.cpp"), "-o", exe, "-lm"])
    full = os.environ.get("CRH_EXACT_MATH_FULL") == "1"
    out = subprocess.run([exe, "1" if full else "4099", "300000000" if full else "2000000"], stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
    text = out.stdout.decode()
    assert out.returncode == 0, text
    lines = [l for l in text.splitlines() if "mismatches" in l]
    assert len(lines) >= 25 and all(l.rstrip().endswith("mismatches 0") for l in lines), text


def test_wrap01_shortcuts_match_wrapminmax(tmp_path):
    """pt_device.h: wrap01() skips the two fmodf of wrapMinMax(x, 0, 1) (vector.h:215-221) for -1 <= x < 1; same bits for every float
    (every 4099th by default, all 2^32 with CRH_EXACT_MATH_FULL=1)."""
    exe = str(tmp_path / "wrap01_check")
    subprocess.check_call(["g++", "-std=c++17", "-O2", "-march=x86-64-v3", "-ffp-contract=off", "-fopenmp", "-I" + os.path.join(REPO, "include"),
                           os.path.join(REPO, "tests", "emu", "wrap01_check.cpp"), "-o", exe, "-lm"])
    out = subprocess.run([exe, "1" if os.environ.get("CRH_EXACT_MATH_FULL") == "1" else "4099"], stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
    assert out.returncode == 0 and out.stdout.decode().rstrip().endswith("mismatches 0"), out.stdout.decode()


def _samples(rng):
    edge = np.array([0.0, -0.0, 1.0, -1.0, 0.5, -0.5, 2.0, 1e-45, -1e-45, 1.17549435e-38, 3.4028235e38, -3.4028235e38, np.inf, -np.inf, np.nan,
                     math.pi, -math.pi, math.pi / 4, 0.785398185253143, 0.7853982, 120.0, 119.99999, 1e9, 2.4e-4, 0.975, 0.9750001, 0.4375, 0.6875, 1.1875, 2.4375,
                     33554432.0, 5.9604645e-08, 0.99999994, 1.0000001], np.float32)
    unit = rng.random(400000, dtype=np.float32)                                  # what getDimension() returns
    bits = rng.integers(0, 2 ** 32, 400000, dtype=np.uint64).astype(np.uint32).view(np.float32)
    return {"edge": edge, "unit": unit, "angle": unit * np.float32(2 * math.pi), "sym": unit * 2 - 1, "wide": (unit * 2 - 1) * np.float32(300.0), "bits": bits}


def _same(a, b):
    return ((a.view(np.uint32) == b.view(np.uint32)) | (np.isnan(a) & np.isnan(b)))


@pytest.mark.gpu
def test_device_build_matches_the_host_libm(pkg):
    """numpy's float32 ufuncs are NOT the reference's libm on every platform, so the expected values come from the C library itself (ctypes)."""
    import ctypes as C
    if pkg.api.device_count() < 1:
        pytest.fail("GPU tier needs a HIP device; libcray_hip has no CPU fallback")
    libm = C.CDLL("libm.so.6")
    def host(name, x, y=None):
        f = getattr(libm, name)
        f.restype = C.c_float
        f.argtypes = [C.c_float] * (1 if y is None else 2)
        if y is None:
            return np.array([f(float(v)) for v in x], np.float32)
        return np.array([f(float(a), float(b)) for a, b in zip(x, y)], np.float32)
    rng = np.random.default_rng(7)
    S = _samples(rng)
    ctx = pkg.api.Context(0)
    try:
        unary = {"sinf": ("sinf", ["edge", "angle", "wide", "bits"]), "cosf": ("cosf", ["edge", "angle", "wide", "bits"]),
                 "sincosf_sin": ("sinf", ["edge", "angle", "bits"]), "sincosf_cos": ("cosf", ["edge", "angle", "bits"]),
                 "logf": ("logf", ["edge", "unit", "bits"]), "log10f": ("log10f", ["edge", "unit", "bits"]),
                 "atanf": ("atanf", ["edge", "wide", "bits"]), "tanf": ("tanf", ["edge", "angle", "wide", "bits"]), "acosf": ("acosf", ["edge", "sym", "bits"]), "asinf": ("asinf", ["edge", "sym", "bits"])}
        for dev, (cname, sets) in unary.items():
            for sname in sets:
                x = S[sname][:60000]
                got, want = ctx.eval_math(dev, x), host(cname, x)
                ok = _same(got, want)
                assert ok.all(), (dev, sname, x[~ok][:4], got[~ok][:4], want[~ok][:4])
        n = 60000
        for yv in (5.0, 2.4, 0.4166666667, 2.0, -0.1332047592, -0.0755148492):       # schlick, sRGB, grayscale, blackbody
            for sname in ("edge", "unit", "bits"):
                x = S[sname][:n]
                y = np.full(x.shape, yv, np.float32)
                got, want = ctx.eval_math("powf", x, y), host("powf", x, y)
                ok = _same(got, want)
                assert ok.all(), ("powf", yv, sname, x[~ok][:4], got[~ok][:4], want[~ok][:4])
        for xa, ya in ((S["bits"][:n], S["bits"][n:2 * n]), (S["sym"][:n], S["wide"][:n]), (np.repeat(S["edge"], len(S["edge"])), np.tile(S["edge"], len(S["edge"])))):
            for dev, cname in (("powf", "powf"), ("atan2f", "atan2f")):
                got, want = ctx.eval_math(dev, xa, ya), host(cname, xa, ya)
                ok = _same(got, want)
                assert ok.all(), (dev, xa[~ok][:4], ya[~ok][:4], got[~ok][:4], want[~ok][:4])
    finally:
        ctx.close()

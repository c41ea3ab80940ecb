"""Shared fixtures. CPU tier: `pytest -m "not gpu"`; GPU tier (parity through the C-ABI): `pytest -m gpu`."""
import gzip
import json
import os
import subprocess
import sys

import numpy as np
import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "oracle"))
GOLDEN = os.path.join(REPO, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def pkg():
    from __graft_entry__ import load_package
    return load_package()


@pytest.fixture(scope="session")
def manifest():
    with open(os.path.join(GOLDEN, "manifest.json")) as f:
        return json.load(f)


@pytest.fixture(scope="session")
def golden_blob(tmp_path_factory):
    """name -> path of the decompressed scene blob fixture."""
    cache = {}
    root = tmp_path_factory.mktemp("blobs")

    def get(name):
        if name not in cache:
            out = os.path.join(root, name + ".blob")
            with gzip.open(os.path.join(GOLDEN, name + ".blob.gz"), "rb") as f, open(out, "wb") as g:
                g.write(f.read())
            cache[name] = out
        return cache[name]
    return get


@pytest.fixture(scope="session")
def golden_ref(manifest):
    """name -> float32 [H, W, 3] render buffer of the real reference (c-ray-ref-strict)."""
    def get(name):
        m = manifest[name]
        with gzip.open(os.path.join(GOLDEN, name + ".ref.f32.gz"), "rb") as f:
            return np.frombuffer(f.read(), dtype=np.float32).reshape(m["height"], m["width"], 3).copy()
    return get


def locked_make(cmd):
    """`make` behind a file lock: with pytest-xdist several workers reach a session fixture at once, and two makes in one directory corrupt each other's objects."""
    import fcntl
    with open(os.path.join(REPO, "tests", ".make.lock"), "w") as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        try:
            subprocess.check_call(cmd)
        finally:
            fcntl.flock(lock, fcntl.LOCK_UN)


@pytest.fixture(scope="session")
def oracle():
    locked_make(["make", "-s", "-C", os.path.join(REPO, "oracle"), "oracle"])
    import oracle_py
    return oracle_py


@pytest.fixture(scope="session")
def emu(oracle):
    """ctypes handle of the host emulation of the device lane code (tests/emu, test infrastructure)."""
    import ctypes as C
    locked_make(["make", "-s", "-C", os.path.join(REPO, "tests", "emu")])
    abi = oracle.abi
    L = C.CDLL(os.path.join(REPO, "tests", "emu", "libcray_emu.so"))
    L.emu_render_region.argtypes = [C.POINTER(abi.SceneDesc), C.POINTER(abi.RenderParams), C.c_void_p, C.POINTER(abi.Counters),
                                    C.POINTER(C.c_uint32), C.c_int, C.c_int, C.c_int]
    L.emu_trace_rays.argtypes = [C.POINTER(abi.SceneDesc), C.c_void_p, C.c_uint64, C.c_void_p]
    L.emu_compile_check.argtypes = [C.POINTER(abi.SceneDesc), C.POINTER(C.c_uint32), C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]
    L.emu_last_error.restype = C.c_char_p
    L.emu_set_sampler.argtypes = [C.c_int]
    L.emu_set_sampler.restype = None
    return L


def image_stats(img, ref):
    d = np.abs(img.astype(np.float64) - ref.astype(np.float64))
    per_px = np.sqrt((d ** 2).sum(axis=2))
    return {"rmse": float(np.sqrt((d ** 2).mean())), "max": float(d.max()), "mean_l2": float(per_px.mean()),
            "frac_gt_1e-3": float((per_px > 1e-3).mean()), "frac_ne": float((d.max(axis=2) > 0).mean())}


def camera_rays(desc, n, seed):
    """n world-space rays from the camera position, spread around the viewing direction (6 floats each)."""
    rng = np.random.default_rng(seed)
    rays = np.zeros((n, 6), np.float32)
    cam = desc.camera
    A = np.array(list(cam.A), dtype=np.float64).reshape(3, 4)
    local = np.stack([rng.normal(size=n) * 0.35, rng.normal(size=n) * 0.25, np.ones(n)], axis=1)
    rays[:, 0:3] = A[:, 3]
    rays[:, 3:6] = (local @ A[:, :3].T).astype(np.float32)
    return rays


BIG_CASES = ["cfg2_hdr_small", "cfg3_venus_small", "cfg4_statues_small", "soup_1m_small"]


def built_blob(name):
    """Path of scenes/_built/<name>.blob (made by __graft_entry__.build() where /root/reference exists; travels to the GPU box)."""
    from __graft_entry__ import BUILT
    path = os.path.join(BUILT, name + ".blob")
    if not os.path.exists(path):
        pytest.skip(f"scenes/_built/{name}.blob not built")
    return path


def resize_camera(scene, width, height):
    """The blobs of BASELINE.json configs[1..4] are flattened at full frame size; a frame of the same aspect ratio differs only in
    camera.width / height (camera.c:27-32: the sensor is a function of FOV and aspect ratio)."""
    cam = scene.desc.camera
    assert cam.width * height == cam.height * width, "resize_camera keeps the aspect ratio"
    cam.width, cam.height = width, height
    return scene


def kernel_forms(pkg, ctx):
    """The forms of the path-tracing kernel the loaded library holds (CRH_OPT_KERNEL). The product library holds k_pathtrace_roll alone (round 4); the kernel
    emulation of tests/emu (and A/B variant libraries built with -DCRH_WITH_ALT_KERNELS) also hold the one-unit-at-a-time and the workgroup-cooperative forms,
    so the tests that compare the forms with each other run in full in the emulation tier and compare the rolling kernel with itself / the fixtures on the GPU."""
    abi = pkg.abi
    forms = [abi.KERNEL_ROLL]
    for k in (abi.KERNEL_WAVE, abi.KERNEL_WG):
        try:
            ctx.set_option(abi.OPT_KERNEL, k)
            forms.append(k)
        except pkg.api.CrhError as e:
            assert e.code == abi.ERR_UNSUPPORTED, e
    ctx.set_option(abi.OPT_KERNEL, abi.KERNEL_ROLL)
    return forms

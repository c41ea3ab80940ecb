"""ctypes binding of oracle/libcray_oracle.so (TEST INFRASTRUCTURE).

Importable only from tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg; the product
(c-ray_amd/) never imports this module. Scene structs come from the product's ABI mirror.
"""
import ctypes as C
import importlib.util
import os
import sys

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_REPO = os.path.dirname(_HERE)


def _abi():
    name = "cray_amd.abi"
    if name in sys.modules:
        return sys.modules[name]
    spec = importlib.util.spec_from_file_location("_cray_amd_abi_for_oracle", os.path.join(_REPO, "c-ray_amd", "abi.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


abi = _abi()
_lib = None


def lib():
    global _lib
    if _lib is None:
        path = os.path.join(_HERE, "libcray_oracle.so")
        if not os.path.exists(path):
            raise FileNotFoundError(f"{path} missing: run `make -C oracle oracle`")
        L = C.CDLL(path)
        L.oracle_render_region.argtypes = [C.POINTER(abi.SceneDesc), C.POINTER(abi.RenderParams), C.c_void_p,
                                           C.POINTER(abi.Counters), C.c_int]
        L.oracle_render_region.restype = C.c_int
        L.oracle_trace_rays.argtypes = [C.POINTER(abi.SceneDesc), C.c_void_p, C.c_uint64, C.c_void_p]
        L.oracle_trace_rays.restype = C.c_int
        L.oracle_to_srgb8.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p]
        L.oracle_to_srgb8.restype = None
        L.oracle_sampler_draws.argtypes = [C.c_uint32, C.c_int, C.c_int, C.c_int, C.c_void_p]
        L.oracle_sampler_draws.restype = None
        L.oracle_camera_ray.argtypes = [C.POINTER(abi.SceneDesc), C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]
        L.oracle_camera_ray.restype = None
        L.oracle_set_sampler.argtypes = [C.c_int]
        L.oracle_set_sampler.restype = None
        L.orc_bvh_triangle_bounds.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p]
        L.orc_bvh_triangle_bounds.restype = None
        L.orc_bvh_build.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p, C.POINTER(C.c_uint32)]
        L.orc_bvh_build.restype = C.c_int
        L.crh_blob_load.argtypes = [C.c_char_p, C.POINTER(C.POINTER(abi.SceneDesc)), C.POINTER(abi.BlobPrefs)]
        L.crh_blob_load.restype = C.c_int
        L.crh_blob_free.argtypes = [C.POINTER(abi.SceneDesc)]
        L.crh_blob_free.restype = None
        _lib = L
    return _lib


class OracleScene:
    """A scene blob loaded through the oracle library's own copy of the blob reader."""

    def __init__(self, path):
        self.ptr = C.POINTER(abi.SceneDesc)()
        self.prefs = abi.BlobPrefs()
        rc = lib().crh_blob_load(os.fsencode(path), C.byref(self.ptr), C.byref(self.prefs))
        if rc != 0:
            raise RuntimeError(f"crh_blob_load({path}) failed: {rc}")

    @property
    def desc(self):
        return self.ptr.contents

    def close(self):
        if self.ptr:
            lib().crh_blob_free(self.ptr)
            self.ptr = C.POINTER(abi.SceneDesc)()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def render(scene, width, height, samples, bounces, region=None, first_pass=0, pass_count=None, fb=None, threads=0, halton=False):
    """renderThread() (halton=True: renderThreadInteractive(), run single-threaded: with more threads the reference
    races on state.finishedPasses and is not reproducible) restated. Returns (float32 [H, W, 3] in stored row order, counters dict)."""
    x0, y0, x1, y1 = region if region else (0, 0, width, height)
    p = abi.RenderParams(x0, y0, x1, y1, width, height, first_pass,
                         samples - first_pass if pass_count is None else pass_count, samples, bounces)
    if fb is None:
        fb = np.zeros((height, width, 3), dtype=np.float32)
    assert fb.dtype == np.float32 and fb.flags["C_CONTIGUOUS"]
    cnt = abi.Counters()
    lib().oracle_set_sampler(1 if halton else 0)
    try:
        rc = lib().oracle_render_region(scene.ptr, C.byref(p), fb.ctypes.data, C.byref(cnt), threads)
    finally:
        lib().oracle_set_sampler(0)
    if rc != 0:
        raise RuntimeError(f"oracle_render_region failed: {rc}")
    return fb, cnt.as_dict()


def trace_rays(scene, rays):
    rays = np.ascontiguousarray(rays, dtype=np.float32).reshape(-1, 6)
    hits = np.zeros(len(rays), dtype=abi.HIT_DTYPE)
    rc = lib().oracle_trace_rays(scene.ptr, rays.ctypes.data, len(rays), hits.ctypes.data)
    if rc != 0:
        raise RuntimeError(f"oracle_trace_rays failed: {rc}")
    return hits


def to_srgb8(fb):
    h, w, _ = fb.shape
    out = np.zeros((h, w, 3), dtype=np.uint8)
    lib().oracle_to_srgb8(np.ascontiguousarray(fb).ctypes.data, w, h, out.ctypes.data)
    return out


def sampler_draws(pixel_index, pass_, max_passes, n):
    out = np.zeros(n, dtype=np.float32)
    lib().oracle_sampler_draws(pixel_index, pass_, max_passes, n, out.ctypes.data)
    return out


def camera_ray(scene, x, y, pass_, max_passes):
    out = np.zeros(6, dtype=np.float32)
    lib().oracle_camera_ray(scene.ptr, x, y, pass_, max_passes, out.ctypes.data)
    return out


def bvh_build_triangles(polys_ptr, vertices_ptr, count):
    """Restated reference builder (bvh.c:87-316) over `count` triangles: (nodes uint32[n, 8], prim order int32[count])."""
    import numpy as np
    boxes = np.empty((max(count, 1), 6), np.float32)
    centers = np.empty((max(count, 1), 3), np.float32)
    lib().orc_bvh_triangle_bounds(polys_ptr, vertices_ptr, count, boxes.ctypes.data, centers.ctypes.data)
    nodes = np.zeros((max(2 * count - 1, 1), 8), np.uint32)
    prims = np.zeros(max(count, 1), np.int32)
    n = C.c_uint32(0)
    rc = lib().orc_bvh_build(boxes.ctypes.data, centers.ctypes.data, count, nodes.ctypes.data, prims.ctypes.data, C.byref(n))
    if rc == -2:
        raise OverflowError("orc_bvh_build: the reference's builder overflows its 2 * count - 1 node array on this input (bvh.c:271)")
    if rc != 0:
        raise MemoryError("orc_bvh_build")
    return nodes[:n.value], prims[:count]

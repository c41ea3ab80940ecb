"""api.py — thin ctypes binding of libcray_hip.so (include/cray_hip.h).

Plumbing, not the product: everything that computes lives behind the C-ABI in c-ray_amd/csrc. There is
no CPU fallback here either: `library()` raises if the HIP library has not been built, and every call
raises `CrhError` on a non-zero return code (the reference's convention is 0 = ok, negative = failure,
src/datatypes/scene.c:122-134).
"""
import ctypes as C
import os

import numpy as np

from . import abi

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("CRH_LIB") or os.path.join(HERE, "_lib", "libcray_hip.so")    # CRH_LIB: A/B another build (dev)


class CrhError(RuntimeError):
    def __init__(self, code, where, detail=""):
        self.code = code
        super().__init__(f"{where} failed with {code}: {detail}")


_lib = None


def library():
    """Load libcray_hip.so (built by c-ray_amd/build.py). Raises loudly if it is missing."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise FileNotFoundError(f"{LIB_PATH} is missing: run `python c-ray_amd/build.py` "
                                "(or __graft_entry__.build()); there is no CPU fallback")
    # torch wheels bundle their own libamdhip64; whichever HIP runtime is loaded first owns the device, and a second copy
    # then reports "No HIP GPUs". Let torch (when present) load its copy first: libcray_hip's NEEDED libamdhip64 resolves to it.
    try:
        import torch  # noqa: F401
    except Exception:
        pass
    L = C.CDLL(LIB_PATH)
    # The emulation library of tests/emu — the kernels compiled for the CPU on a HIP shim — exports the same C-ABI. It is test infrastructure: only
    # a caller that says so explicitly (the emulation tier's child processes and the tools/emu_* scripts) may bind to it.
    if hasattr(L, "crh_emu_stats") and os.environ.get("CRH_ALLOW_EMULATION") != "1":
        raise RuntimeError(f"{LIB_PATH} is the CPU emulation of the kernels (test infrastructure): the product has no CPU path "
                           "(set CRH_ALLOW_EMULATION=1 only in tests / tools that mean it)")
    ctx = C.c_void_p
    sig = {
        "crh_device_count": (C.c_int, []),
        "crh_last_error": (C.c_char_p, []),
        "crh_abi_version": (C.c_int, []),
        "crh_context_create": (C.c_int, [C.c_int, C.c_void_p, C.POINTER(ctx)]),
        "crh_context_destroy": (C.c_int, [ctx]),
        "crh_set_option": (C.c_int, [ctx, C.c_int, C.c_int64]),
        "crh_debug_wave_stats": (C.c_int, [ctx, C.c_void_p, C.c_uint32]),
        "crh_debug_phase_ticks": (C.c_int, [ctx, C.c_void_p]),
        "crh_scene_upload": (C.c_int, [ctx, C.POINTER(abi.SceneDesc)]),
        "crh_scene_compile": (C.c_int, [C.POINTER(abi.SceneDesc), C.c_int, C.POINTER(C.c_void_p)]),
        "crh_scene_upload_compiled": (C.c_int, [ctx, C.c_void_p]),
        "crh_compiled_scene_free": (None, [C.c_void_p]),
        "crh_debug_upload_counts": (None, [C.POINTER(C.c_int), C.POINTER(C.c_int)]),
        "crh_framebuffer_alloc": (C.c_int, [ctx, C.c_int, C.c_int, C.POINTER(C.c_void_p)]),
        "crh_framebuffer_free": (C.c_int, [ctx, C.c_void_p]),
        "crh_framebuffer_clear": (C.c_int, [ctx, C.c_void_p, C.c_int, C.c_int]),
        "crh_framebuffer_download": (C.c_int, [ctx, C.c_void_p, C.c_int, C.c_int, C.c_void_p]),
        "crh_framebuffer_to_srgb8": (C.c_int, [ctx, C.c_void_p, C.c_int, C.c_int, C.c_void_p]),
        "crh_framebuffer_strips_to_srgb8": (C.c_int, [ctx, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]),
        "crh_render_region": (C.c_int, [ctx, C.POINTER(abi.RenderParams), C.c_void_p]),
        "crh_render_tiles": (C.c_int, [ctx, C.POINTER(abi.RenderParams), C.POINTER(abi.Tile), C.c_uint32, C.c_void_p]),
        "crh_synchronize": (C.c_int, [ctx]),
        "crh_frames_reduce": (C.c_int, [C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), C.c_int, C.c_int, C.c_int]),
        "crh_frames_gather": (C.c_int, [C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), C.c_int, C.c_int, C.c_int, C.c_int]),
        "crh_frames_prepare": (C.c_int, [C.POINTER(C.c_int), C.c_int]),
        "crh_context_prepare": (C.c_int, [ctx]),
        "crh_counters_get": (C.c_int, [ctx, C.POINTER(abi.Counters)]),
        "crh_counters_reset": (C.c_int, [ctx]),
        "crh_kernel_time_ms": (C.c_int, [ctx, C.POINTER(C.c_float), C.POINTER(C.c_double), C.POINTER(C.c_uint64)]),
        "crh_last_kernel_name": (C.c_char_p, [ctx]),
        "crh_trace_rays": (C.c_int, [ctx, C.c_void_p, C.c_uint64, C.c_void_p]),
        "crh_debug_eval_math": (C.c_int, [ctx, C.c_int, C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p]),
        "crh_bvh_build_triangles": (C.c_int, [ctx, C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p,
                                              C.POINTER(C.c_uint32), C.POINTER(abi.BvhBuildStats)]),
        "crh_blob_save": (C.c_int, [C.c_char_p, C.POINTER(abi.SceneDesc), C.POINTER(abi.BlobPrefs)]),
        "crh_blob_load": (C.c_int, [C.c_char_p, C.POINTER(C.POINTER(abi.SceneDesc)), C.POINTER(abi.BlobPrefs)]),
        "crh_blob_free": (None, [C.POINTER(abi.SceneDesc)]),
        "crh_debug_ray_dump": (C.c_int, [ctx, C.c_uint32]),
        "crh_debug_ray_dump_counts": (C.c_int, [ctx, C.POINTER(C.c_uint64), C.c_void_p, C.c_uint32]),
        "crh_debug_ray_dump_fetch": (C.c_int, [ctx, C.c_uint32, C.c_uint32, C.c_uint32, C.c_void_p]),
        "crh_debug_walk_probe": (C.c_int, [ctx, C.c_int, C.c_int, C.c_int, C.c_int, C.c_uint32, C.c_int, C.POINTER(C.c_float), C.POINTER(C.c_uint64)]),
        "crh_debug_walk_probe_fetch": (C.c_int, [ctx, C.c_int, C.c_uint32, C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p]),
        "crh_debug_walk_probe_compare": (C.c_int, [ctx, C.POINTER(C.c_uint64)]),
        "crh_debug_plan_units": (C.c_int, [C.POINTER(abi.RenderParams), C.POINTER(abi.Tile), C.c_uint32, C.c_uint32, C.c_void_p, C.c_uint64,
                                           C.POINTER(C.c_uint64), C.POINTER(C.c_int32)]),
    }
    for name, (res, args) in sig.items():
        fn = getattr(L, name)
        fn.restype = res
        fn.argtypes = args
    _lib = L
    return L


def _check(rc, where):
    if rc != 0:
        raise CrhError(rc, where, (library().crh_last_error() or b"").decode(errors="replace"))


def device_count():
    return library().crh_device_count()


class CompiledScene:
    """crh_scene_compile: the device layout of a scene, derived once on the host; any number of contexts upload it (Context.upload_compiled)."""

    def __init__(self, scene, walk=0):
        desc = scene.ptr if hasattr(scene, "ptr") else C.pointer(scene)
        self.h = C.c_void_p()
        _check(library().crh_scene_compile(desc, walk, C.byref(self.h)), "crh_scene_compile")

    def close(self):
        if self.h:
            library().crh_compiled_scene_free(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def upload_counts():
    """(layout compiles, uploads) of this process so far (crh_debug_upload_counts)."""
    a, b = C.c_int(0), C.c_int(0)
    library().crh_debug_upload_counts(C.byref(a), C.byref(b))
    return a.value, b.value


def plan_units(width, height, samples, tiles, cu_count=256, first_pass=0, pass_count=None):
    """crh_debug_plan_units (no device needed): the work units of one dispatch in hand-out order, as an int32 array [n, 8] of
    x0, y0, x1, y1, block area, taper level, first pass, pass count — and the pass chunk."""
    p = abi.RenderParams(0, 0, 0, 0, width, height, first_pass, samples - first_pass if pass_count is None else pass_count, samples, 1)
    arr = (abi.Tile * len(tiles))(*[abi.Tile(*t) for t in tiles])
    n, chunk = C.c_uint64(0), C.c_int32(0)
    _check(library().crh_debug_plan_units(C.byref(p), arr, len(tiles), cu_count, None, 0, C.byref(n), C.byref(chunk)), "crh_debug_plan_units")
    units = np.zeros((n.value, 8), np.int32)
    _check(library().crh_debug_plan_units(C.byref(p), arr, len(tiles), cu_count, units.ctypes.data, n.value, C.byref(n), C.byref(chunk)), "crh_debug_plan_units")
    return units, chunk.value


class Scene:
    """A flat scene blob (written by the flattener, c-ray_amd/host/flatten.c) loaded through the library."""

    def __init__(self, path):
        self.path = path
        self.ptr = C.POINTER(abi.SceneDesc)()
        self.prefs = abi.BlobPrefs()
        rc = library().crh_blob_load(os.fsencode(path), C.byref(self.ptr), C.byref(self.prefs))
        if rc != 0:
            raise CrhError(rc, f"crh_blob_load({path})")

    @property
    def desc(self):
        return self.ptr.contents

    def close(self):
        if self.ptr:
            library().crh_blob_free(self.ptr)
            self.ptr = C.POINTER(abi.SceneDesc)()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Context:
    """One crh_ctx = one GPU. `stream` may be a raw hipStream_t (e.g. torch.cuda.current_stream().cuda_stream)."""

    def __init__(self, device=0, stream=None):
        self.L = library()
        self.h = C.c_void_p()
        _check(self.L.crh_context_create(int(device), C.c_void_p(stream) if stream else None, C.byref(self.h)),
               "crh_context_create")
        self._owned_fbs = []

    def close(self):
        if self.h:
            for fb in self._owned_fbs:
                self.L.crh_framebuffer_free(self.h, fb)
            self._owned_fbs = []
            self.L.crh_context_destroy(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_option(self, option, value):
        _check(self.L.crh_set_option(self.h, int(option), int(value)), "crh_set_option")

    def wave_stats(self):
        buf = np.zeros((8192, 2), dtype=np.uint64)
        n = self.L.crh_debug_wave_stats(self.h, buf.ctypes.data, 8192)
        if n < 0:
            _check(n, "crh_debug_wave_stats")
        return buf[:n]

    def phase_ticks(self):
        buf = np.zeros(24, dtype=np.uint64)
        _check(self.L.crh_debug_phase_ticks(self.h, buf.ctypes.data), "crh_debug_phase_ticks")
        keys = ("setup", "traverse", "shade", "w_node", "w_tri", "w_ctrl", "w_round", "w_shade", "w_setup", "u_node", "u_shade",
                "t_swap", "t_gen", "n_swap", "n_gen", "u_swap", "u_tri", "u_ctrl", "w_tri_in", "u_tri_in", "w_ctrl_in", "u_ctrl_in", "u_wait_tri", "u_wait_fin")
        return {k: int(v) for k, v in zip(keys, buf)}

    def prepare(self):
        """crh_context_prepare: per-wave buffers + code objects before the scene is there (optional)."""
        _check(self.L.crh_context_prepare(self.h), "crh_context_prepare")

    def set_sched(self, node, tri, ctrl, swap_min, fill_to=160, run_num=4, tri_in_run=12, ctrl_in_run=12, shade_min=0, swap_in_run=20):
        """shade_min = 0 keeps the library's tuned value (include/cray_hip.h: CRH_OPT_SCHED_RUNS)."""
        self.set_option(abi.OPT_SCHED_WEIGHTS, node | (tri << 12) | (ctrl << 24) | (swap_min << 36))
        self.set_option(abi.OPT_SCHED_RUNS, fill_to | (run_num << 12) | (tri_in_run << 16) | (ctrl_in_run << 24) | (shade_min << 32) | (swap_in_run << 40))

    def set_sched_wg(self, linger=8, drain_at=192, max_drainers=1, partial_min=16, walk_min=32, fill_to=768):
        """Scheduler of the workgroup kernel (CRH_OPT_KERNEL = KERNEL_WG), see cray_hip.hip: k_pathtrace_wg."""
        self.set_option(abi.OPT_SCHED_WG, linger | (drain_at << 8) | (max_drainers << 20) | (partial_min << 24) | (walk_min << 32) | (fill_to << 40))

    def upload(self, scene):
        desc = scene.ptr if hasattr(scene, "ptr") else C.pointer(scene)
        _check(self.L.crh_scene_upload(self.h, desc), "crh_scene_upload")

    def upload_compiled(self, compiled):
        """crh_scene_upload_compiled: the copies only, from a CompiledScene (one layout compile for several contexts)."""
        _check(self.L.crh_scene_upload_compiled(self.h, compiled.h), "crh_scene_upload_compiled")

    def framebuffer(self, width, height):
        p = C.c_void_p()
        _check(self.L.crh_framebuffer_alloc(self.h, width, height, C.byref(p)), "crh_framebuffer_alloc")
        self._owned_fbs.append(p)
        return p

    def clear(self, fb, width, height):
        _check(self.L.crh_framebuffer_clear(self.h, fb, width, height), "crh_framebuffer_clear")

    def render_region(self, fb, width, height, samples, bounces, region=None, first_pass=0, pass_count=None):
        x0, y0, x1, y1 = region if region else (0, 0, width, height)
        p = abi.RenderParams(x0, y0, x1, y1, width, height, first_pass,
                             samples - first_pass if pass_count is None else pass_count, samples, bounces)
        _check(self.L.crh_render_region(self.h, C.byref(p), fb), "crh_render_region")

    def render_tiles(self, fb, width, height, samples, bounces, tiles, first_pass=0, pass_count=None):
        p = abi.RenderParams(0, 0, 0, 0, width, height, first_pass,
                             samples - first_pass if pass_count is None else pass_count, samples, bounces)
        arr = (abi.Tile * len(tiles))(*[abi.Tile(*t) for t in tiles])
        _check(self.L.crh_render_tiles(self.h, C.byref(p), arr, len(tiles), fb), "crh_render_tiles")

    def synchronize(self):
        _check(self.L.crh_synchronize(self.h), "crh_synchronize")

    def download(self, fb, width, height):
        out = np.empty((height, width, 3), dtype=np.float32)
        _check(self.L.crh_framebuffer_download(self.h, fb, width, height, out.ctypes.data), "crh_framebuffer_download")
        return out

    def to_srgb8(self, fb, width, height):
        out = np.empty((height, width, 3), dtype=np.uint8)
        _check(self.L.crh_framebuffer_to_srgb8(self.h, fb, width, height, out.ctypes.data), "crh_framebuffer_to_srgb8")
        return out

    def counters(self):
        c = abi.Counters()
        _check(self.L.crh_counters_get(self.h, C.byref(c)), "crh_counters_get")
        return c.as_dict()

    def reset_counters(self):
        _check(self.L.crh_counters_reset(self.h), "crh_counters_reset")

    def kernel_time_ms(self):
        last, total, n = C.c_float(), C.c_double(), C.c_uint64()
        _check(self.L.crh_kernel_time_ms(self.h, C.byref(last), C.byref(total), C.byref(n)), "crh_kernel_time_ms")
        return last.value, total.value, n.value

    def strips_to_srgb8(self, fb, width, height, strip_rows, g, n_gpus, out):
        """crh_framebuffer_strips_to_srgb8: the rows of GPU g's strips of the 8-bit frame into `out` (uint8 [height, width, 3]); the other rows stay as they are."""
        _check(self.L.crh_framebuffer_strips_to_srgb8(self.h, fb, width, height, strip_rows, g, n_gpus, out.ctypes.data), "crh_framebuffer_strips_to_srgb8")
        return out

    def last_kernel_name(self):
        """The instantiation of the path-tracing kernel launched last, as a profiler names it ("" before the first dispatch)."""
        return (self.L.crh_last_kernel_name(self.h) or b"").decode()

    def bvh_build_triangles(self, polys_ptr, poly_count, vertices_ptr, vertex_count):
        """buildBottomLevelBvh on the GPU: (nodes uint32[n, 8] = crh_bvh_node records, prim order int32[count], stats dict)."""
        nodes = np.zeros((max(2 * poly_count - 1, 1), 8), np.uint32)
        prims = np.zeros(max(poly_count, 1), np.int32)
        n = C.c_uint32(0)
        st = abi.BvhBuildStats()
        _check(self.L.crh_bvh_build_triangles(self.h, polys_ptr, poly_count, vertices_ptr, vertex_count, nodes.ctypes.data,
                                              prims.ctypes.data, C.byref(n), C.byref(st)), "crh_bvh_build_triangles")
        return nodes[:n.value], prims[:poly_count], {k: getattr(st, k) for k, _ in abi.BvhBuildStats._fields_ if k != "pad"}

    def frames_reduce(self, fb, width, height):
        """crh_frames_reduce over this one context (n = 1: a no-op unless CRH_FORCE_RCCL is set)."""
        ctxs = (C.c_void_p * 1)(self.h)
        fbs = (C.c_void_p * 1)(fb)
        _check(self.L.crh_frames_reduce(ctxs, fbs, 1, width, height), "crh_frames_reduce")

    def eval_math(self, function, x, y=None):
        """crh_debug_eval_math: the device build of exact_math.h on caller values (function = a name from abi.MATH_FUNCTIONS)."""
        x = np.ascontiguousarray(x, dtype=np.float32).ravel()
        out = np.empty_like(x)
        yp = None
        if y is not None:
            y = np.ascontiguousarray(y, dtype=np.float32).ravel()
            assert y.size == x.size
            yp = y.ctypes.data
        _check(self.L.crh_debug_eval_math(self.h, abi.MATH_FUNCTIONS.index(function), x.ctypes.data, yp, x.size, out.ctypes.data), "crh_debug_eval_math")
        return out

    # ---- round 6: the walk-only probe (debug / measurement; include/cray_hip.h) ----
    def ray_dump(self, rays_per_wave):
        _check(self.L.crh_debug_ray_dump(self.h, rays_per_wave), "crh_debug_ray_dump")

    def ray_dump_counts(self, max_waves=8192):
        total = C.c_uint64(0)
        per = np.zeros(max_waves, dtype=np.uint32)
        _check(self.L.crh_debug_ray_dump_counts(self.h, C.byref(total), per.ctypes.data, max_waves), "crh_debug_ray_dump_counts")
        return int(total.value), per

    def ray_dump_fetch(self, wave, first, n):
        out = np.zeros((n, 6), dtype=np.float32)
        _check(self.L.crh_debug_ray_dump_fetch(self.h, wave, first, n, out.ctypes.data), "crh_debug_ray_dump_fetch")
        return out

    def walk_probe(self, wps, stack_lds=12, inst_lds=True, fused=True, unit_rays=512, slot=0):
        """-> (kernel ms, rays walked)"""
        ms = C.c_float(0.0)
        rays = C.c_uint64(0)
        _check(self.L.crh_debug_walk_probe(self.h, wps, stack_lds, 1 if inst_lds else 0, int(fused), unit_rays, slot, C.byref(ms), C.byref(rays)), "crh_debug_walk_probe")
        return float(ms.value), int(rays.value)

    def walk_probe_fetch(self, slot, wave, first, n):
        hits = np.zeros((n, 4), dtype=np.float32)
        inst = np.zeros(n, dtype=np.int32)
        _check(self.L.crh_debug_walk_probe_fetch(self.h, slot, wave, first, n, hits.ctypes.data, inst.ctypes.data), "crh_debug_walk_probe_fetch")
        return hits, inst

    def walk_probe_compare(self):
        d = C.c_uint64(0)
        _check(self.L.crh_debug_walk_probe_compare(self.h, C.byref(d)), "crh_debug_walk_probe_compare")
        return int(d.value)

    def trace_rays(self, rays):
        rays = np.ascontiguousarray(rays, dtype=np.float32).reshape(-1, 6)
        hits = np.zeros(len(rays), dtype=abi.HIT_DTYPE)
        _check(self.L.crh_trace_rays(self.h, rays.ctypes.data, len(rays), hits.ctypes.data), "crh_trace_rays")
        return hits

"""ctypes mirror of include/cray_hip.h (the C-ABI of libcray_hip.so).

Plumbing only: the product is the C-ABI library; this file lets tests/ and bench.py call it.
Field order and sizes must match the header exactly (tests/test_abi.py checks the sizes).
"""
import ctypes as C

ABI_VERSION = 3

OK, ERR_INVALID, ERR_NO_DEVICE, ERR_HIP, ERR_UNSUPPORTED, ERR_IO, ERR_NOMEM = 0, -1, -2, -3, -4, -5, -6
NODE_NONE = 0xFFFFFFFF
OPT_COUNTER_LEVEL, OPT_BLOCKS_PER_CU, OPT_PASS_CHUNK, OPT_WAVES_PER_SIMD, OPT_WAVE_STATS, OPT_UNIT_ITEMS, OPT_SCHED_WEIGHTS, OPT_UNITS_PER_WAVE, OPT_SAMPLER, OPT_TAIL_PERCENT, OPT_SCHED_RUNS, OPT_KERNEL, OPT_SCHED_WG, OPT_TRACE_SLABS, OPT_SHADE_SORT, OPT_TAIL_SPLIT, OPT_ROUND_LIMIT, OPT_RENDER_SLABS, OPT_WALK, OPT_STREAM_COHORTS = 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16, 17, 18, 19, 20
WALK_BINARY, WALK_WIDE4 = 0, 1
TAIL_SPLIT_DEFAULT = 0        # CRH_TAIL_SPLIT_DEFAULT
TRACE_SLABS_LITERAL, TRACE_SLABS_EXACT = 0, 1
KERNEL_WAVE, KERNEL_WG, KERNEL_ROLL, KERNEL_STREAM = 0, 1, 2, 3
SAMPLER_RANDOM, SAMPLER_HALTON = 0, 1


class BvhNode(C.Structure):
    _fields_ = [("bounds", C.c_float * 6), ("first", C.c_uint32), ("count_leaf", C.c_uint32)]


class Poly(C.Structure):
    _fields_ = [("v", C.c_int32 * 3), ("n", C.c_int32 * 3), ("t", C.c_int32 * 3), ("bits", C.c_uint32)]


class Instance(C.Structure):
    _fields_ = [("Ainv", C.c_float * 12), ("A", C.c_float * 12), ("kind", C.c_uint32), ("object", C.c_uint32),
                ("density", C.c_float), ("pad", C.c_uint32 * 5)]


class Mesh(C.Structure):
    _fields_ = [("node_base", C.c_uint32), ("node_count", C.c_uint32), ("prim_base", C.c_uint32),
                ("poly_base", C.c_uint32), ("poly_count", C.c_uint32), ("material_base", C.c_uint32),
                ("material_count", C.c_uint32), ("texcoord_count", C.c_uint32), ("ray_offset", C.c_float),
                ("pad", C.c_uint32 * 3)]


class Sphere(C.Structure):
    _fields_ = [("radius", C.c_float), ("ray_offset", C.c_float), ("material", C.c_uint32), ("pad", C.c_uint32)]


class Material(C.Structure):
    _fields_ = [("emission", C.c_float * 4), ("ior", C.c_float), ("bsdf", C.c_uint32), ("pad", C.c_uint32 * 2)]


class GNode(C.Structure):
    _fields_ = [("kind", C.c_uint32), ("a", C.c_uint32), ("b", C.c_uint32), ("c", C.c_uint32), ("f", C.c_float * 8)]


class Texture(C.Structure):
    _fields_ = [("offset", C.c_uint64), ("width", C.c_uint32), ("height", C.c_uint32), ("channels", C.c_uint32),
                ("is_float", C.c_uint32), ("has_alpha", C.c_uint32), ("pad", C.c_uint32)]


class Camera(C.Structure):
    _fields_ = [("right", C.c_float * 3), ("up", C.c_float * 3), ("forward", C.c_float * 3),
                ("sensor", C.c_float * 2), ("aperture", C.c_float), ("focal_distance", C.c_float),
                ("width", C.c_int32), ("height", C.c_int32), ("A", C.c_float * 12)]


class SceneDesc(C.Structure):
    _fields_ = [
        ("struct_size", C.c_uint32), ("abi_version", C.c_uint32),
        ("nodes", C.POINTER(BvhNode)), ("node_count", C.c_uint64),
        ("prim_indices", C.POINTER(C.c_int32)), ("prim_index_count", C.c_uint64),
        ("tlas_node_base", C.c_uint32), ("tlas_node_count", C.c_uint32),
        ("tlas_prim_base", C.c_uint32), ("tlas_prim_count", C.c_uint32),
        ("polys", C.POINTER(Poly)), ("poly_count", C.c_uint64),
        ("vertices", C.POINTER(C.c_float)), ("vertex_count", C.c_uint64),
        ("normals", C.POINTER(C.c_float)), ("normal_count", C.c_uint64),
        ("texcoords", C.POINTER(C.c_float)), ("texcoord_count", C.c_uint64),
        ("instances", C.POINTER(Instance)), ("instance_count", C.c_uint64),
        ("meshes", C.POINTER(Mesh)), ("mesh_count", C.c_uint64),
        ("spheres", C.POINTER(Sphere)), ("sphere_count", C.c_uint64),
        ("materials", C.POINTER(Material)), ("material_count", C.c_uint64),
        ("gnodes", C.POINTER(GNode)), ("gnode_count", C.c_uint64),
        ("textures", C.POINTER(Texture)), ("texture_count", C.c_uint64),
        ("texture_data", C.POINTER(C.c_uint8)), ("texture_bytes", C.c_uint64),
        ("camera", Camera),
        ("background", C.c_uint32), ("pad", C.c_uint32),
    ]


class RenderParams(C.Structure):
    _fields_ = [("x0", C.c_int32), ("y0", C.c_int32), ("x1", C.c_int32), ("y1", C.c_int32),
                ("image_width", C.c_int32), ("image_height", C.c_int32),
                ("first_pass", C.c_int32), ("pass_count", C.c_int32), ("max_passes", C.c_int32),
                ("bounces", C.c_int32)]


class Counters(C.Structure):
    _fields_ = [(n, C.c_uint64) for n in
                ("paths", "rays", "node_tests", "tri_tests", "inst_visits", "inst_hits", "sphere_tests", "tex_fetches")]

    def as_dict(self):
        return {n: int(getattr(self, n)) for n, _ in self._fields_}


class Hit(C.Structure):
    _fields_ = [("inst", C.c_int32), ("poly", C.c_int32), ("distance", C.c_float), ("uv", C.c_float * 2),
                ("point", C.c_float * 3), ("normal", C.c_float * 3), ("node_tests", C.c_uint32),
                ("tri_tests", C.c_uint32), ("material", C.c_uint32)]


class BlobPrefs(C.Structure):
    _fields_ = [("image_width", C.c_int32), ("image_height", C.c_int32), ("sample_count", C.c_int32),
                ("bounces", C.c_int32), ("tile_width", C.c_int32), ("tile_height", C.c_int32),
                ("tile_order", C.c_int32), ("pad", C.c_int32)]


class Tile(C.Structure):
    """crh_tile: one rectangle of a multi-tile dispatch (reference tile coordinates, y from the bottom)."""
    _fields_ = [("x0", C.c_int32), ("y0", C.c_int32), ("x1", C.c_int32), ("y1", C.c_int32)]


# numpy dtype of crh_hit for bulk comparisons
HIT_DTYPE = [("inst", "<i4"), ("poly", "<i4"), ("distance", "<f4"), ("uv", "<f4", (2,)), ("point", "<f4", (3,)),
             ("normal", "<f4", (3,)), ("node_tests", "<u4"), ("tri_tests", "<u4"), ("material", "<u4")]

# every symbol include/cray_hip.h declares (tests/test_abi.py checks the .so exports all of them)
EXPORTED_SYMBOLS = [
    "crh_device_count", "crh_last_error", "crh_abi_version", "crh_context_create", "crh_context_destroy",
    "crh_set_option", "crh_debug_wave_stats", "crh_debug_phase_ticks",
    "crh_scene_upload", "crh_framebuffer_alloc", "crh_framebuffer_free", "crh_framebuffer_clear",
    "crh_framebuffer_download", "crh_framebuffer_to_srgb8", "crh_render_region", "crh_render_tiles",
    "crh_synchronize", "crh_frames_reduce", "crh_frames_gather", "crh_frames_prepare", "crh_context_prepare", "crh_counters_get", "crh_counters_reset", "crh_kernel_time_ms", "crh_trace_rays",
    "crh_blob_save", "crh_blob_load", "crh_blob_free", "crh_bvh_build_triangles", "crh_debug_eval_math", "crh_debug_plan_units", "crh_last_kernel_name", "crh_framebuffer_strips_to_srgb8",
    "crh_scene_compile", "crh_scene_upload_compiled", "crh_compiled_scene_free", "crh_debug_upload_counts",
    "crh_debug_ray_dump", "crh_debug_ray_dump_counts", "crh_debug_ray_dump_fetch", "crh_debug_walk_probe", "crh_debug_walk_probe_fetch", "crh_debug_walk_probe_compare",
]
MATH_FUNCTIONS = ("sinf", "cosf", "sincosf_sin", "sincosf_cos", "logf", "log10f", "atanf", "acosf", "asinf", "tanf", "powf", "atan2f")   # enum crh_math_function


class BvhBuildStats(C.Structure):
    _fields_ = [("upload_ms", C.c_double), ("build_ms", C.c_double), ("download_ms", C.c_double),
                ("levels", C.c_uint32), ("upper_nodes", C.c_uint32), ("subtrees", C.c_uint32), ("pad", C.c_uint32)]

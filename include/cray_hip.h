/*
 * cray_hip.h — C-ABI of libcray_hip.so, the MI355X (gfx950) path-tracing backend for c-ray.
 *
 * This is the drop-in boundary for ONE hot path of the reference renderer (VKoskiv/c-ray v0.6.3):
 *   renderThread()          src/renderer/renderer.c:258-327   (pixel x pass loop, running mean)
 *   pathTrace()             src/renderer/pathtrace.c:32-60    (bounce loop, Russian roulette)
 *   traverseTopLevelBvh()   src/accelerators/bvh.c:488-496    (two-level BVH walk)
 *   bsdfNode.sample()       src/nodes/...                     (node-graph shading)
 * The reference has no FFI of its own: the call that a maintainer replaces is
 *   struct texture *renderFrame(struct renderer *r)   (src/renderer/renderer.h:104),
 * and the host-side replacement (c-ray_amd/host/renderer_hip.c) is a thin C file that flattens
 * `struct world` into the POD arrays below and calls the functions declared here.
 * INTEGRATION.md shows that binding.
 *
 * Conventions (mirroring the reference's C conventions):
 *   - plain C, no C++ types, no exceptions; every call returns int: 0 = ok, negative = failure
 *     (the reference returns 0 / -1 / -2: src/datatypes/scene.c:122-134, src/utils/platform/thread.c:38-50).
 *   - host buffers are caller-owned; device memory is owned by the context unless the caller passes
 *     its own device pointer (e.g. a torch tensor's data_ptr for the framebuffer).
 *   - one crh_ctx per GPU; calls on one ctx must be serialised by the caller; different ctxs are independent.
 *   - there is NO CPU fallback: every entry point that needs a device fails with CRH_ERR_NO_DEVICE if none exists.
 */
#ifndef CRAY_HIP_H
#define CRAY_HIP_H

#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

/* 2 (round 4): crh_debug_plan_units writes EIGHT ints per unit (six in version 1), crh_frames_gather / crh_frames_prepare / crh_context_prepare exist,
 * CRH_OPT_ROUND_LIMIT / CRH_OPT_RENDER_SLABS exist, CRH_KERNEL_WAVE / CRH_KERNEL_WG are refused by the product build.
 * 3 (round 5): crh_scene_compile / crh_scene_upload_compiled / crh_compiled_scene_free (one layout compile for the contexts of a multi-GPU frame) and CRH_OPT_WALK exist.
 * (round 6 adds values, not entry points — CRH_KERNEL_STREAM for CRH_OPT_KERNEL, CRH_OPT_STREAM_COHORTS, and the debug entries crh_debug_ray_dump* / crh_debug_walk_probe* of the
 * walk-only probe —: a version-3 host is unaffected, the version stays 3.)
 * A host checks crh_abi_version() == CRH_ABI_VERSION. */
#define CRH_ABI_VERSION 3
/* layout version of crh_scene_desc and of the scene blobs (crh_blob_save / crh_blob_load): the records have not changed since round 1 */
#define CRH_SCENE_VERSION 1

/* ---- error codes ------------------------------------------------------------------------- */
#define CRH_OK                0
#define CRH_ERR_INVALID      -1   /* bad argument / malformed scene description            */
#define CRH_ERR_NO_DEVICE    -2   /* no HIP device (there is no CPU fallback)               */
#define CRH_ERR_HIP          -3   /* a HIP runtime call failed; see crh_last_error()        */
#define CRH_ERR_UNSUPPORTED  -4   /* scene uses a construct the device VM does not implement */
#define CRH_ERR_IO           -5   /* scene blob could not be read / written                 */
#define CRH_ERR_NOMEM        -6

/* ---- flattened scene records (all little-endian POD) --------------------------------------- */

/* struct bvhNode verbatim (src/accelerators/bvh.c:37-42), 32 B.
 * bounds = {minx,maxx,miny,maxy,minz,maxz}; count_leaf: primCount = bits 0..29, isLeaf = bit 30.
 * Children of an inner node are nodes first, first+1 *relative to the owning BVH's node range*;
 * a leaf indexes prim_indices[first .. first+primCount) relative to the owning BVH's prim range. */
typedef struct crh_bvh_node {
	float    bounds[6];
	uint32_t first;
	uint32_t count_leaf;
} crh_bvh_node;
#define CRH_NODE_PRIMCOUNT(n) ((n).count_leaf & 0x3FFFFFFFu)
#define CRH_NODE_ISLEAF(n)    (((n).count_leaf >> 30) & 1u)

/* struct poly verbatim (src/datatypes/poly.h:11-18), 40 B. Indices are ABSOLUTE indices into the
 * global vertex / normal / texcoord arrays (src/utils/loaders/formats/wavefront/wavefront.c:110-126).
 * bits: materialIndex = bits & 0xFFFF, vertexCount = (bits >> 16) & 7, hasNormals = (bits >> 24) & 0xFF. */
typedef struct crh_poly {
	int32_t  v[3];
	int32_t  n[3];
	int32_t  t[3];
	uint32_t bits;
} crh_poly;
#define CRH_POLY_MATERIAL(p)   ((p).bits & 0xFFFFu)
#define CRH_POLY_HASNORMALS(p) ((((p).bits >> 24) & 0xFFu) != 0)

/* struct instance (src/datatypes/instance.h:23-28) reduced to what the path reads: rows 0..2 of
 * composite.A / composite.Ainv (src/datatypes/transforms.c:76-116 never touch row 3). */
#define CRH_INSTANCE_SPHERE        0u
#define CRH_INSTANCE_MESH          1u
#define CRH_INSTANCE_SPHERE_VOLUME 2u   /* newSphereVolume (instance.c:137-147): constant-density medium bounded by the sphere */
#define CRH_INSTANCE_MESH_VOLUME   3u   /* newMeshVolume   (instance.c:253-263): ... bounded by the mesh                         */
typedef struct crh_instance {
	float    Ainv[12];   /* row-major 3x4 */
	float    A[12];      /* row-major 3x4 */
	uint32_t kind;       /* CRH_INSTANCE_*                                   */
	uint32_t object;     /* index into spheres[] or meshes[]                  */
	float    density;    /* volumes: sphereVolume / meshVolume .density (instance.c:22-30) */
	uint32_t pad[5];
} crh_instance;          /* 128 B */

/* struct mesh (src/datatypes/mesh.h:20-46) as offsets into the concatenated arrays. */
typedef struct crh_mesh {
	uint32_t node_base;       /* first node of this BLAS in nodes[]                */
	uint32_t node_count;      /* bvh->nodeCount (0 = empty BVH, bvh.c:362-365)      */
	uint32_t prim_base;       /* first entry of this BLAS in prim_indices[]         */
	uint32_t poly_base;       /* first polygon of this mesh in polys[]              */
	uint32_t poly_count;
	uint32_t material_base;   /* mesh->materials[0] in materials[]                  */
	uint32_t material_count;
	uint32_t texcoord_count;  /* mesh->textureCoordCount (instance.c:151)           */
	float    ray_offset;      /* mesh->rayOffset, read AFTER the TLAS build (instance.c:227) */
	uint32_t pad[3];
} crh_mesh;               /* 48 B */

typedef struct crh_sphere {
	float    radius;
	float    ray_offset;      /* sphere->rayOffset (instance.c:106) */
	uint32_t material;        /* index into materials[] */
	uint32_t pad;
} crh_sphere;             /* 16 B */

/* The three fields of struct material read at render time (pathtrace.c:44,46; plastic.c:68-77). */
typedef struct crh_material {
	float    emission[4];
	float    ior;
	uint32_t bsdf;            /* root node index in gnodes[] */
	uint32_t pad[2];
} crh_material;           /* 32 B */

/* Node graph (src/nodes/...), hash-consed by the reference and flattened 1:1: one record per
 * distinct reference node; a/b/c are child node indices (CRH_NODE_NONE if absent). */
#define CRH_NODE_NONE 0xFFFFFFFFu
enum crh_node_kind {
	/* bsdf nodes: src/nodes/shaders/ *.c */
	CRH_BSDF_DIFFUSE = 1,   /* a=color                                   diffuse.c:40-47   */
	CRH_BSDF_METAL,         /* a=color b=roughness(value)                metal.c:40-55     */
	CRH_BSDF_GLASS,         /* a=color b=roughness(value) c=IOR(value)   glass.c:41-87     */
	CRH_BSDF_PLASTIC,       /* a=color b=roughness(color) c=diffuse bsdf plastic.c:42-87   */
	CRH_BSDF_MIX,           /* a=A b=B c=factor(value)                   mix.c:42-50       */
	CRH_BSDF_ADD,           /* a=A b=B                                   add.c:42-49       */
	CRH_BSDF_TRANSPARENT,   /* a=color                                   transparent.c:40-44 */
	CRH_BSDF_EMISSION,      /* a=color b=strength(value)                 emission.c:42-49  */
	CRH_BSDF_ISOTROPIC,     /* a=color                                   isotropic.c:40-47 */
	CRH_BSDF_BACKGROUND,    /* a=color b=strength(value) c=offset(value) background.c:39-66 */
	/* color nodes: src/nodes/textures, src/nodes/converter */
	CRH_COLOR_CONSTANT = 32,/* f[0..3]=rgba                              constant.c:39-42  */
	CRH_COLOR_IMAGE,        /* a=texture index (NONE = NULL tex) b=options image.c:31-68   */
	CRH_COLOR_CHECKER,      /* a=A b=B c=scale(value)                    checker.c:31-74   */
	CRH_COLOR_GRADIENT,     /* f[0..3]=down f[4..7]=up                   gradient.c:40-45  */
	CRH_COLOR_BLACKBODY,    /* a=temperature(value)                      blackbody.c:38-42 */
	CRH_COLOR_COMBINE,      /* a=value                                   combine.c:38-43   */
	CRH_COLOR_COMBINERGB,   /* a=R b=G c=B (values)                      combinergb.c:42-51*/
	CRH_COLOR_VECTOCOLOR,   /* a=vector                                  vectocolor.c:38-43*/
	/* value nodes */
	CRH_VALUE_CONSTANT = 64,/* f[0]                                      valuenode.c:36-40 */
	CRH_VALUE_ALPHA,        /* a=color                                   alpha.c:38-41     */
	CRH_VALUE_GRAYSCALE,    /* a=color                                   grayscale.c:38-41 */
	CRH_VALUE_MATH,         /* a=A b=B c=op (enum mathOp, math.h:11-27)  math.c:42-95      */
	CRH_VALUE_FRESNEL,      /* a=IOR(value) b=normal(vector, unused)     fresnel.c:38-51   */
	CRH_VALUE_RAYLENGTH,    /*                                            raylength.c:36-40 */
	/* vector nodes */
	CRH_VEC_CONSTANT = 96,  /* f[0..2]                                   vectornode.c:38-42*/
	CRH_VEC_NORMAL,         /*                                            normal.c:37-41    */
	CRH_VEC_VECMATH         /* a=A b=B c=op (enum vecOp, vecmath.h:11-22) vecmath.c:42-81   */
};
#define CRH_IMAGE_SRGB_TRANSFORM 0x01u  /* src/nodes/textures/image.h:12 */
#define CRH_IMAGE_NO_BILINEAR    0x02u  /* src/nodes/textures/image.h:13 */

typedef struct crh_gnode {
	uint32_t kind;
	uint32_t a, b, c;
	float    f[8];
} crh_gnode;              /* 48 B */

/* struct texture (src/datatypes/image/texture.h:25-36); pixel data lives in texture_data[offset..]. */
typedef struct crh_texture {
	uint64_t offset;      /* byte offset into texture_data, 16-byte aligned */
	uint32_t width, height, channels;
	uint32_t is_float;    /* precision == float_p */
	uint32_t has_alpha;
	uint32_t pad;
} crh_texture;            /* 32 B */

/* struct camera (src/datatypes/camera.h:15-33), fields read by getCameraRay (camera.c:58-87). */
typedef struct crh_camera {
	float   right[3], up[3], forward[3];
	float   sensor[2];
	float   aperture;
	float   focal_distance;
	int32_t width, height;
	float   A[12];        /* composite.A rows 0..2 */
} crh_camera;

/* Everything renderFrame() can see in struct world + the global vertex buffers, as POD. */
typedef struct crh_scene_desc {
	uint32_t struct_size;     /* sizeof(crh_scene_desc), checked by crh_scene_upload */
	uint32_t abi_version;     /* CRH_SCENE_VERSION: the layout of this description's records (the entry points' version is crh_abi_version()) */

	const crh_bvh_node *nodes;        uint64_t node_count;        /* all BLAS then the TLAS */
	const int32_t      *prim_indices; uint64_t prim_index_count;
	uint32_t tlas_node_base, tlas_node_count;                     /* scene->topLevel */
	uint32_t tlas_prim_base, tlas_prim_count;

	const crh_poly     *polys;        uint64_t poly_count;
	const float        *vertices;     uint64_t vertex_count;      /* g_vertices, xyz       */
	const float        *normals;      uint64_t normal_count;      /* g_normals, xyz        */
	const float        *texcoords;    uint64_t texcoord_count;    /* g_textureCoords, xy   */

	const crh_instance *instances;    uint64_t instance_count;
	const crh_mesh     *meshes;       uint64_t mesh_count;
	const crh_sphere   *spheres;      uint64_t sphere_count;
	const crh_material *materials;    uint64_t material_count;
	const crh_gnode    *gnodes;       uint64_t gnode_count;
	const crh_texture  *textures;     uint64_t texture_count;
	const uint8_t      *texture_data; uint64_t texture_bytes;

	crh_camera camera;
	uint32_t   background;    /* scene->background root in gnodes[] */
	uint32_t   pad;
} crh_scene_desc;

/* One dispatch of the hot path: passes [first_pass, first_pass+pass_count) of every pixel of the
 * region [x0,x1) x [y0,y1) (reference tile coordinates: y counts from the BOTTOM of the image,
 * renderer.c:277-280), folded into the running mean exactly like renderer.c:288-291. */
typedef struct crh_render_params {
	int32_t x0, y0, x1, y1;
	int32_t image_width, image_height;
	int32_t first_pass;       /* completedSamples-1 of the first pass to run        */
	int32_t pass_count;
	int32_t max_passes;       /* prefs.sampleCount: part of the seed (sampler.c:42) */
	int32_t bounces;          /* prefs.bounces                                      */
} crh_render_params;

/* Counters kept by the kernels (the N_* of the algorithmic-bytes formula). */
typedef struct crh_counters {
	uint64_t paths;           /* camera rays started                                  */
	uint64_t rays;            /* getClosestIsect calls (pathtrace.c:26,38)            */
	uint64_t node_tests;      /* intersectNode calls (bvh.c:326), TLAS + BLAS         */
	uint64_t tri_tests;       /* rayIntersectsWithPolygon calls (poly.c:17)           */
	uint64_t inst_visits;     /* instance.intersectFn calls from TLAS leaves          */
	uint64_t inst_hits;       /* ... that returned true                               */
	uint64_t sphere_tests;    /* rayIntersectsWithSphere calls                        */
	uint64_t tex_fetches;     /* textureGetPixelInternal calls                        */
} crh_counters;

/* Closest-hit record returned by crh_trace_rays (test/diagnostic entry; = struct hitRecord,
 * src/datatypes/hitrecord.h:14-23, after getClosestIsect). */
typedef struct crh_hit {
	int32_t  inst;            /* instIndex, -1 = miss                                 */
	int32_t  poly;            /* polygon index in polys[], -1 for spheres / miss      */
	float    distance;
	float    uv[2];           /* texture-mapped uv (instance.c:150-167, 33-43)        */
	float    point[3];        /* hitPoint, world space                                */
	float    normal[3];       /* surfaceNormal, world space                           */
	uint32_t node_tests;      /* per-ray intersectNode count                          */
	uint32_t tri_tests;       /* per-ray triangle test count                          */
	uint32_t material;        /* index in materials[] of the hit material             */
} crh_hit;

typedef struct crh_ctx crh_ctx;

/* ---- entry points -------------------------------------------------------------------------- */

/* Number of HIP devices visible, 0 if none (never negative). */
int crh_device_count(void);
/* Human-readable text for the most recent failure on this thread. */
const char *crh_last_error(void);
/* ABI version of the loaded library (== CRH_ABI_VERSION of the header it was built from). */
int crh_abi_version(void);

/* Create / destroy the per-GPU context. `stream` may be NULL (the context then creates its own
 * stream) or an existing hipStream_t passed as void* (e.g. torch.cuda.current_stream().cuda_stream). */
int crh_context_create(int device, void *stream, crh_ctx **out);
int crh_context_destroy(crh_ctx *ctx);
/* Optional: the part of crh_scene_upload that does not need the scene — the per-wave buffers of a full-size dispatch and the code object
 * of the kernel the options set so far select (HIP loads kernels lazily; the plain variant: a scene with node programs or volumes loads its own). A host calls it after crh_set_option and before its scene is
 * flattened, on the context's own thread, so that this work overlaps the flattening (renderer_hip.c); crh_scene_upload then skips it. */
int crh_context_prepare(crh_ctx *ctx);

/* Per-context knobs. */
#define CRH_OPT_COUNTER_LEVEL 1   /* 2 (default): every crh_counters field; 1: paths + rays only (timed runs)      */
#define CRH_OPT_BLOCKS_PER_CU 2   /* persistent 256-thread workgroups launched per compute unit (default 4)        */
#define CRH_OPT_PASS_CHUNK    3   /* passes a wave traces per pixel block before folding them (default 64)         */
#define CRH_OPT_WAVES_PER_SIMD 4  /* register budget variant of the path-tracing kernel: 4 (<=128 VGPRs, default) or 1 */
#define CRH_OPT_UNIT_ITEMS    6   /* paths per work unit (pixel block x passes) the block shape aims for (default 2048) */
#define CRH_OPT_SCHED_WEIGHTS 7   /* wave scheduler, four 12-bit fields: node | tri<<12 | ctrl<<24 step weights, finished+idle lanes that trigger a swap step <<36 (default 70,160,120,16) */
#define CRH_OPT_SCHED_RUNS   11   /* wave scheduler: paths a wave keeps in flight before it waits for idle lanes to generate more (0..192, default 160)
                                   * | n<<12: a node / triangle run continues while n/8 of its lanes still want that step (default 4)
                                   * | m<<16: inside a node run, triangles are tested as soon as m lanes wait for them (default 12; 65 = never)
                                   * | k<<24: likewise instance entries / sphere tests (control steps), as soon as k lanes wait (default 12; 65 = never)
                                   * | s<<32: a shading step starts once 64 hits wait; in scenes with four or more shade classes (instances whose hits run
                                   *          the same shading code) it takes whole classes, largest first, until at least s lanes are busy (1..128, default 48; 0 = keep)
                                   * | r<<40: inside a node run, lanes whose walk has ended are retired and idle lanes refilled in place once r of them wait (round 4; default 20; 65 = never; 0 = keep) */
#define CRH_OPT_UNITS_PER_WAVE 8  /* shrink the pixel blocks until every wave gets at least this many work units (default 8) */
#define CRH_OPT_SAMPLER       9   /* which sampler seeds a (pixel, pass): CRH_SAMPLER_RANDOM = renderThread (sampler.c:41-44, default),
                                   * CRH_SAMPLER_HALTON = renderThreadInteractive (renderer.c:204: Halton index = pass + 1, halton.c:16-31) */
#define CRH_OPT_TAIL_PERCENT 10   /* share (0..50, default 16) of a dispatch's pixels that ends the work queue as quarter-size blocks, so the waves finish close together;
                                   * | (p2 + 1) << 8 also sets the share (default 4) at the very end that is cut into sixteenth-size blocks */
#define CRH_OPT_KERNEL       12   /* which form of the path-tracing kernel: CRH_KERNEL_ROLL (default, and since round 4 the only form in the product library) = every wave a
                                   * self-contained machine that keeps up to four work units open, so that its path table stays full across unit boundaries. Builds with
                                   * -DCRH_WITH_ALT_KERNELS (the kernel emulation of tests/emu, A/B variant libraries) also hold CRH_KERNEL_WAVE = the same machine, one unit
                                   * at a time (the default until round 3) and CRH_KERNEL_WG = the four waves of a workgroup share one path table and take walker / shader
                                   * roles; all three compute the same frame bit for bit. The product build answers CRH_ERR_UNSUPPORTED for the other two */
#define CRH_OPT_SCHED_WG     13   /* workgroup kernel scheduler: linger | drainAt<<8 | maxDrainers<<20 | partialMin<<24 | walkMin<<32 | fillTo<<40 */
#define CRH_OPT_TRACE_SLABS  14   /* crh_trace_rays and rays with a zero / denormal direction component (a slab the reference's arithmetic turns into NaN, bvh.c:326-352):
                                   * CRH_TRACE_SLABS_LITERAL (default) = the reference's select chain followed literally: its record and its node / triangle test counts,
                                   * whatever the ray (such a ray visits most of the scene, like the reference's); CRH_TRACE_SLABS_EXACT = what the render kernels do:
                                   * that slab is tested exactly (inside iff min <= start <= max): never more node visits, the same hit except an origin one ulp
                                   * beside an axis-aligned face — a frame's one such camera ray costs microseconds instead of a third of a second */
#define CRH_OPT_SHADE_SORT   15   /* 0 (default): hits are shaded in the order they were found. N in 1..8: in scenes with at least N shade classes (instances whose hits
                                   * run the same surface-shader code path) the wave and rolling kernels shade them in batches of few classes. Pure scheduling: the frame
                                   * is the same bit for bit. (Default until round 3: 4; slower than 0 since the shading code got shorter.) */
#define CRH_TRACE_SLABS_LITERAL 0
#define CRH_TRACE_SLABS_EXACT   1
#define CRH_KERNEL_WAVE 0
#define CRH_KERNEL_WG   1
#define CRH_KERNEL_ROLL 2
#define CRH_KERNEL_STREAM 3       /* round 6: the same machine as three kernels that take turns — walk / shade + refill / fold — over two pools of paths in global memory, so that
                                   * the BVH walk runs at six or seven waves per SIMD instead of the megakernel's four (csrc/pathtrace_stream.h); the same frame bit for bit.
                                   * Dispatches it cannot serve (volumes, the Halton sampler, the 4-ary walk, bounces <= 0) are rendered by CRH_KERNEL_ROLL. crh_render_tiles
                                   * returns when the dispatch's last iteration has been enqueued, i.e. shortly before it is finished (the host feeds the device iteration by iteration) */
#define CRH_SAMPLER_RANDOM 0
#define CRH_SAMPLER_HALTON 1
#define CRH_OPT_TAIL_SPLIT   16   /* rolling kernel: the work queue ENDS with this many units per wave (0..64; 0 = none, the default) of about 64 paths — blocks of 64 / passes
                                   * pixels, or, from 128 passes per dispatch on, single pixels whose passes are split into segments of 64: whichever waves pull the segments
                                   * trace them, the samples are staged per pixel and folded into the frame in pass order behind the kernel (k_fold_deferred) — the same frame
                                   * bit for bit. Built to bring the last waves of a dispatch in earlier (a pixel's passes need not be one wave's); measured: it does not —
                                   * the last waves are those whose path table is full when the queue runs dry: draining it takes ~1.4 ms (DESIGN.md §6, r03zb_probe_finish*).
                                   * The environment variable CRH_TAIL_SPLIT sets a process's default (A/B runs of unmodified hosts) */
#define CRH_TAIL_SPLIT_DEFAULT 0
#define CRH_OPT_ROUND_LIMIT  17   /* scheduling rounds (2..2e9, default 2e9: hours of one wave's work) after which a wave of the path-tracing kernel gives up instead of spinning
                                   * on: the dispatch is flagged, and crh_synchronize / crh_framebuffer_download / crh_framebuffer_to_srgb8 return CRH_ERR_HIP "incomplete frame"
                                   * (the reference's convention: an error code and a message, src/datatypes/scene.c:122-134). A tiny value is the test hook for that path */
#define CRH_OPT_RENDER_SLABS 18   /* the render kernels and rays with a zero / denormal direction component: CRH_TRACE_SLABS_EXACT (default: see CRH_OPT_TRACE_SLABS) or
                                   * CRH_TRACE_SLABS_LITERAL = the reference's NaN arithmetic followed literally (bvh.c:326-352) — the same node visits as the reference for
                                   * every ray of the frame, at the reference's price for such a ray: a walk of most of the scene with the rest of its wave waiting */
#define CRH_OPT_WALK         19   /* round 5, an EXPERIMENT kept as an option (set before crh_scene_upload): CRH_WALK_BINARY (default, the contract) = the reference's walk over the
                                   * reference's binary tree, node for node (bvh.c:354-441); CRH_WALK_WIDE4 = the render kernel steps through a derived 4-ary copy of every BVH
                                   * (the same boxes, bit for bit; half the dependent round trips), which reaches leaves in another ORDER: the closest hit is the same
                                   * except where two triangles are hit at exactly the same distance (poly.c:33 keeps the first one tested) or where the reference's own
                                   * culling is inconsistent by a rounding error — counted, not assumed (one ray in 4e5 on statues.json). Node-test
                                   * counts are the wide walk's own. Scenes with node programs / volumes, the Halton sampler and crh_trace_rays keep the binary walk */
#define CRH_WALK_BINARY 0
#define CRH_WALK_WIDE4  1
#define CRH_OPT_STREAM_COHORTS 20 /* CRH_KERNEL_STREAM: the most cohorts of 1024 paths a pool holds (default 16384: 16.8 M paths in flight, 2.5 GB of pools + up to 3.2 GB of sample slabs) */
#define CRH_OPT_WAVE_STATS    5   /* debug: record per-wave busy time / units of each dispatch (crh_debug_wave_stats) */
int crh_set_option(crh_ctx *ctx, int option, int64_t value);
int crh_debug_wave_stats(crh_ctx *ctx, uint64_t *out_pairs, uint32_t max_waves);
int crh_debug_phase_ticks(crh_ctx *ctx, uint64_t *out24);   /* debug: 24 values — per-step-kind clocks (10 ns ticks), step counts and lanes served of the counter-level-2 kernel */

/* Round 6, debug / measurement entries (nothing of the drop-in path calls them): the WALK of the path-tracing kernel on its own — getClosestIsect (src/renderer/pathtrace.c:26-30 ->
 * src/accelerators/bvh.c:354-441, 443-496) with the render kernel's lane code and node run, without generation, shading or fold — at occupancies the render kernel cannot reach,
 * on the render kernel's own rays. crh_debug_ray_dump(ctx, n): the following dispatches at CRH_OPT_COUNTER_LEVEL 2 record every ray a wave starts to walk, in the order it starts
 * them, up to n per wave (n = 0: off, buffers released). crh_debug_walk_probe walks the recorded rays with k_walk_probe<wps, stack_lds, inst_lds, fused> (csrc/walk_probe.h: waves per
 * SIMD, traversal-stack entries in LDS, instance records in LDS, the fused node run or the lean one; wps = 0: one ray per lane from start to end, the form the others are checked
 * against) into output `slot` (0 / 1) and reports the kernel's time; crh_debug_walk_probe_compare counts the hits whose bits differ between the two outputs;
 * the two fetch entries copy a stretch of one wave's rays (six floats each: origin, direction) / hits (t, u, v, BLAS prim slot bits; instance in top-level leaf order or -1). */
int crh_debug_ray_dump(crh_ctx *ctx, uint32_t rays_per_wave);
int crh_debug_ray_dump_counts(crh_ctx *ctx, uint64_t *total_out, uint32_t *per_wave_out, uint32_t max_waves);
int crh_debug_ray_dump_fetch(crh_ctx *ctx, uint32_t wave, uint32_t first, uint32_t n, float *rays6_host);
int crh_debug_walk_probe(crh_ctx *ctx, int wps, int stack_lds, int inst_lds, int form, uint32_t unit_rays, int slot, float *ms_out, uint64_t *rays_out);
int crh_debug_walk_probe_fetch(crh_ctx *ctx, int slot, uint32_t wave, uint32_t first, uint32_t n, float *hits4_host, int32_t *inst_host);
int crh_debug_walk_probe_compare(crh_ctx *ctx, uint64_t *differ_out);

/* SURVEY.md 8(f) row 1 — replaces buildBottomLevelBvh() (src/accelerators/bvh.c:299-301 -> buildBvhGeneric, bvh.c:245-287,
 * with getPolyBBoxAndCenter, bvh.c:289-297): the reference's binned-SAH builder on the GPU. The result is THE reference's
 * tree: nodes_out[0 .. *node_count_out) equal its struct bvhNode array (bounds bit for bit, child / first-prim indices,
 * leaf flags and leaf sizes; inner nodes carry primCount 0 where the reference leaves heap garbage) and
 * prim_indices_out[0 .. poly_count) equals bvh->primIndices. Host pointers in and out; polys[i].v[] index `vertices`
 * (3 floats each). nodes_out needs room for 2 * poly_count - 1 nodes. poly_count == 0 gives the empty BVH (node count 0).
 * CRH_ERR_UNSUPPORTED: the mesh is one on which the reference itself writes past the 2 n - 1 nodes it allocates (bvh.c:271; clusters of
 * more than 16 coincident primitives are split into (all | none) down to the depth limit, bvh.c:220) — no reference tree exists; the
 * context stays usable. */
typedef struct crh_bvh_build_stats {
	double   upload_ms, build_ms, download_ms;   /* host->device copies; every kernel + the host's level bookkeeping; results back */
	uint32_t levels, upper_nodes, subtrees, pad;  /* level-synchronous passes; nodes kept by the host; subtrees built by one wave each */
} crh_bvh_build_stats;
int crh_bvh_build_triangles(crh_ctx *ctx, const crh_poly *polys, uint32_t poly_count, const float *vertices, uint64_t vertex_count,
                            crh_bvh_node *nodes_out, int32_t *prim_indices_out, uint32_t *node_count_out, crh_bvh_build_stats *stats /* may be NULL */);

/* Replaces "the CPU reads struct world directly": copies the flattened scene to HBM, derives the
 * device-side acceleration layout (leaf-ordered prepared triangles) and validates the node graph.
 * Lifetime contract: the library does NOT retain the description or any array it points to, and no copy out of them is still in flight when the call returns —
 * the caller may release or overwrite them at once (renderer_hip.c frees the flattened scene while the first dispatch runs; tests/test_gpu_parity.py poisons them). */
int crh_scene_upload(crh_ctx *ctx, const crh_scene_desc *scene);
/* The same in two steps, for a process that renders one scene on several GPUs (round 5; the reference builds its scene once and every worker reads it:
 * src/datatypes/scene.c:111-213): crh_scene_compile derives the device layout ONCE, on the host (no context, no device; `walk` = the CRH_OPT_WALK the contexts use),
 * crh_scene_upload_compiled copies it to a context's GPU — it only reads the handle, so the contexts' threads may call it at the same time —, and
 * crh_compiled_scene_free releases it (any time after the last upload has returned). The same lifetime contract: nothing of `scene` is retained. */
typedef struct crh_compiled_scene crh_compiled_scene;
int crh_scene_compile(const crh_scene_desc *scene, int walk, crh_compiled_scene **out);
int crh_scene_upload_compiled(crh_ctx *ctx, const crh_compiled_scene *compiled);
void crh_compiled_scene_free(crh_compiled_scene *compiled);
void crh_debug_upload_counts(int *compiles, int *uploads);   /* debug / tests: layout compiles and uploads of this process so far */

/* Device float-RGB framebuffer helpers (layout = state.renderBuffer: index (x + (H-1-y)*W)*3,
 * src/datatypes/image/texture.c:24-28). The framebuffer may also be any caller-owned device pointer. */
int crh_framebuffer_alloc(crh_ctx *ctx, int width, int height, float **dev_out);
int crh_framebuffer_free(crh_ctx *ctx, float *dev_fb);
int crh_framebuffer_clear(crh_ctx *ctx, float *dev_fb, int width, int height);
int crh_framebuffer_download(crh_ctx *ctx, const float *dev_fb, int width, int height, float *host_rgb);
/* colorToSRGB + setPixel's 8-bit truncation (color.h:60-84, texture.c:18-22) of the float buffer. */
int crh_framebuffer_to_srgb8(crh_ctx *ctx, const float *dev_fb, int width, int height, uint8_t *host_rgb8);
/* ... for a GPU that owns strips g, g + n_gpus, ... of strip_rows image rows each (a multi-GPU host's share of the frame: host/share.h): only those rows of the 8-bit frame
 * are written into host_rgb8 (a full width x height x 3 buffer), the rest of it is left untouched. n_gpus = 1: the whole frame. */
int crh_framebuffer_strips_to_srgb8(crh_ctx *ctx, const float *dev_fb, int width, int height, int strip_rows, int g, int n_gpus, uint8_t *host_rgb8);

/* THE hot path: replaces the renderThread() pixel x pass loop for one region (renderer.c:275-301).
 * dev_fb is the device running-mean buffer; asynchronous on the context's stream. */
int crh_render_region(crh_ctx *ctx, const crh_render_params *params, float *dev_fb);
/* Same, for a LIST of rectangles in one dispatch (x0..y1 of `params` are ignored): this is how one
 * GPU takes "its" tiles of the reference's ordered tile list (struct renderTile, tile.h:28-37) without
 * one launch per tile. Tiles must not overlap. */
typedef struct crh_tile { int32_t x0, y0, x1, y1; } crh_tile;
int crh_render_tiles(crh_ctx *ctx, const crh_render_params *params, const crh_tile *tiles, uint32_t tile_count, float *dev_fb);
/* Multi-GPU inside one process (the C host, c-ray_amd/host/renderer_hip.c: one crh_ctx + one dispatch thread per
 * GPU): sum the n per-GPU float framebuffers onto ctxs[0]'s with ONE RCCL reduce over xGMI (ncclReduce, float,
 * sum, root 0; communicators from ncclCommInitAll). Tiles are disjoint and non-owned pixels are 0, so the sum is
 * a gather. Replaces the TCP submitWork of 8-bit tiles (src/utils/protocol/worker.c:128-136, server.c:159-174).
 * librccl is loaded on first use; n == 1 is a no-op. (Python hosts use torch.distributed instead: render.py.) */
int crh_frames_reduce(crh_ctx **ctxs, float **dev_fbs, int n, int width, int height);
/* The same frame with 1/n of the bytes on the links: GPU g owns the strips g, g + n, ... of strip_rows pixel rows (host/share.h; render.py:
 * owned_tiles), so it packs exactly those rows, sends them to GPU 0 with ONE ncclSend (GPU 0: one ncclRecv per peer, all in one group — the
 * peers' xGMI links work in parallel), and GPU 0 writes them into its framebuffer. Bit-identical to crh_frames_reduce for strip shares
 * (adding 0.0f changes nothing); CRH_ERR_UNSUPPORTED if the librccl found has no send / receive. */
int crh_frames_gather(crh_ctx **ctxs, float **dev_fbs, int n, int width, int height, int strip_rows);
/* Load librccl and create the communicators of these devices now (ncclCommInitAll: tens to hundreds of milliseconds) — from any thread,
 * before or while the contexts are being created — so that the frame's first reduce / gather does not pay for it. Optional. */
int crh_frames_prepare(const int *devices, int n);
/* Block until everything queued on the context's stream has finished. */
int crh_synchronize(crh_ctx *ctx);

/* Counters accumulated since the last reset (synchronises the stream). */
int crh_counters_get(crh_ctx *ctx, crh_counters *out);
int crh_counters_reset(crh_ctx *ctx);
/* Duration in milliseconds of the most recent crh_render_region's path-tracing kernel, measured
 * with HIP events on the context's stream; also the launch count and total since the last reset. */
int crh_kernel_time_ms(crh_ctx *ctx, float *last_ms, double *total_ms, uint64_t *launches);
/* The instantiation of the path-tracing kernel the context launched last, as rocprofv3 names it (e.g. "k_pathtrace_roll<1,4,false,0>":
 * counter level, waves per SIMD, rare features, sampler); "" before the first dispatch. bench.py quotes it in `roofline.kernel`. */
const char *crh_last_kernel_name(crh_ctx *ctx);

/* Debug / parity: evaluate one function of the device math library (c-ray_amd/csrc/exact_math.h: the libm functions of the hot path
 * restated with the bits of the reference's host libm) on n caller values; y_host is the second argument of powf(x, y) /
 * atan2f(x = first argument y, second argument x) and may be NULL for the unary functions. */
enum crh_math_function { CRH_MATH_SINF = 0, CRH_MATH_COSF, CRH_MATH_SINCOSF_SIN, CRH_MATH_SINCOSF_COS, CRH_MATH_LOGF, CRH_MATH_LOG10F,
                         CRH_MATH_ATANF, CRH_MATH_ACOSF, CRH_MATH_ASINF, CRH_MATH_TANF, CRH_MATH_POWF, CRH_MATH_ATAN2F };
int crh_debug_eval_math(crh_ctx *ctx, int function, const float *x_host, const float *y_host, uint64_t n, float *out_host);
/* Debug / test entry that needs NO device: the work units crh_render_tiles would hand to the kernel for this dispatch on a GPU with
 * cu_count compute units, at the default options, in hand-out order. units_out (may be NULL): eight ints per unit — the pixel rectangle
 * x0, y0, x1, y1 (clipped to its tile), the block area in pixels, the taper level (0 regular, 1 quarter blocks, 2 sixteenth blocks, 3 the 64-path
 * units of CRH_OPT_TAIL_SPLIT), the unit's first pass and pass count (the dispatch's, except for the pass segments of split pixels). */
int crh_debug_plan_units(const crh_render_params *params, const crh_tile *tiles, uint32_t tile_count, uint32_t cu_count, int32_t *units_out,
						 uint64_t max_units, uint64_t *unit_count_out, int32_t *pass_chunk_out);

/* Diagnostic / parity entry: getClosestIsect (pathtrace.c:26-30) for n caller-supplied world-space
 * rays (6 floats each: start xyz, direction xyz). Host in, host out. */
int crh_trace_rays(crh_ctx *ctx, const float *rays_host, uint64_t n, crh_hit *hits_host);

/* Scene blob I/O (c-ray_amd/host/scene_blob.c): a flat file holding one crh_scene_desc, written by
 * the flattener so that tests/bench can run where the reference loader and assets are absent. */
typedef struct crh_blob_prefs {        /* the struct prefs fields the path needs (renderer.h:58-87) */
	int32_t image_width, image_height, sample_count, bounces, tile_width, tile_height, tile_order, pad;
} crh_blob_prefs;
int  crh_blob_save(const char *path, const crh_scene_desc *scene, const crh_blob_prefs *prefs);
int  crh_blob_load(const char *path, crh_scene_desc **scene_out, crh_blob_prefs *prefs_out);
void crh_blob_free(crh_scene_desc *scene);

#ifdef __cplusplus
}
#endif
#endif /* CRAY_HIP_H */

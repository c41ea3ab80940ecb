/*
 * pt_device.h — lane-level logic of the MI355X path-tracing kernels (one ray per lane).
 *
 * Everything here is written from scratch for the flattened scene (include/cray_hip.h); each block
 * cites the reference function (file:line under /root/reference/src) whose RESULT it must reproduce.
 * The float expressions keep the reference's operand order; the translation unit is compiled with
 * -ffp-contract=off so that only the explicit fmaf() of the slab test fuses (bvh.c:318-324, where
 * FP_FAST_FMAF is defined for the reference build), and with correctly rounded fp32 divide/sqrt.
 *
 * The same header compiles for the device (hipcc, the product: csrc/cray_hip.hip) and for the host
 * (g++, tests/emu/ only: a lane-by-lane emulation that lets the CPU-only test tier check the kernel
 * logic bit-for-bit against the oracle). The host build is never linked into libcray_hip.so.
 */
#pragma once
#include <stdint.h>
#include <math.h>
#include <float.h>
#include <type_traits>
#include <utility>
#include "cray_hip.h"
#include "exact_math.h"     /* sinf, cosf, tanf, powf, logf, log10f, atan2f, acosf, asinf with the host libm's bits (namespace crh::em) */

#if defined(__HIPCC__)
#define CRH_DEV __device__ __forceinline__
#define CRH_DEV_NOINLINE __device__ __noinline__
#define CRH_MEM __device__ __forceinline__            /* member functions */
#else
#define CRH_MEM inline __attribute__((always_inline))
#define CRH_DEV static inline __attribute__((always_inline))
#define CRH_DEV_NOINLINE static __attribute__((noinline))
#endif

namespace crh {

#define CRH_PI 3.141592653589793238462643383279502f        /* includes.h:13 */
#define CRH_NONE 0xFFFFFFFFu

struct alignas(16) f4 { float x, y, z, w; };
struct v3 { float x, y, z; };
struct v2 { float x, y; };
struct rgba { float r, g, b, a; };

/* ---- device-side scene ----------------------------------------------------------------------- */

/* One instance visit (stepCtrl) reads the first 64-B line of this record, finishing a hit both (N_instVisit,
 * N_instHit). instances[] is in TLAS leaf order: the record of TLAS prim slot s is instances[s - tlas_prim_base], so a
 * TLAS leaf is walked without the prim-index indirection; `orig` is the reference's instance index (reported by
 * crh_trace_rays). */
#define CRH_DINST_SPHERE     0u
#define CRH_DINST_MESH       1u   /* BLAS with an inner root: root = device index of the root's child pair */
#define CRH_DINST_MESH_LEAF  2u   /* bvh->nodeCount == 1: root = device index of the root leaf (bvh.c:382-387) */
#define CRH_DINST_MESH_EMPTY 3u   /* bvh->nodeCount == 0 (bvh.c:362-365) */
#define CRH_DINST_KIND(k)    ((k) & 15u)
#define CRH_DINST_VOLUME     16u  /* flag: the sphere / mesh bounds a constant-density medium (instance.c:62-92, 187-216) */
#define CRH_DINST_CLASS_SHIFT 8u /* bits 8..10: shade class — instances whose hits run the same shading code path (a scheduling hint, scene_compile.cpp) */
#define CRH_DINST_CLASSES    8u
#define CRH_DINST_CLASS(k)   (((k) >> CRH_DINST_CLASS_SHIFT) & (CRH_DINST_CLASSES - 1u))
struct alignas(16) DInstance {
	float Ainv[12];
	uint32_t kind;        /* CRH_DINST_* */
	uint32_t root;
	float    ray_offset;  /* mesh->rayOffset / sphere->rayOffset */
	float    radius;      /* sphere */
	float A[12];
	uint32_t orig;        /* index in crh_scene_desc.instances */
	uint32_t poly_base;   /* mesh: first polygon in polys[] (crh_trace_rays reports polygon indices) */
	uint32_t material;    /* sphere: material index; mesh: material_base */
	float    density;     /* volumes */
};
static_assert(sizeof(DInstance) == 128, "two 64-byte lines (InstLine)");

/* Per BLAS prim slot, next to tris[]: what finishing a hit on that triangle needs, so that poly.c:37-48 + instance.c:150-167
 * read ONE 64-B record instead of prim index -> polygon -> three normals / three texture coordinates -> mesh. */
#define CRH_SHADE_HASNORMALS 0x80000000u   /* n0..n2 are the vertex normals; otherwise n0 = e1 x e2 (poly.c:45-47) */
#define CRH_SHADE_HASUV      0x40000000u   /* mesh has texture coordinates and the polygon references them (instance.c:152) */
struct alignas(16) DShadeTri {
	float n0[3], n1[3], n2[3];
	float t0[2], t1[2], t2[2];
	uint32_t flags;       /* CRH_SHADE_* | polygon materialIndex */
};

/* pure-node operand reference (compiled at upload, see scene_compile.cpp: operand()).
 * Sub-graphs that do not depend on the hit are folded to constants on the host; the common hit-dependent
 * leaves get their own kinds so that only rare graphs (checker, grayscale(image), ...) run a program. */
#define CRH_OPR_CONST       0u   /* index into consts[] (f4)                                   */
#define CRH_OPR_IMAGE       1u   /* index into images[]: colour = image fetch                  */
#define CRH_OPR_IMAGE_ALPHA 2u   /* index into images[]: value = alpha of the fetch (alpha.c)  */
#define CRH_OPR_GRADIENT    3u   /* index into consts[]: down, up (gradient.c:40-45)           */
#define CRH_OPR_PROGRAM     4u   /* offset into prog[]                                         */
#define CRH_OPR(kind, idx) (((uint32_t)(kind) << 29) | (uint32_t)(idx))
#define CRH_OPR_KIND(r) ((r) >> 29)
#define CRH_OPR_IDX(r)  ((r) & 0x1FFFFFFFu)

struct DImage { uint32_t tex; uint32_t options; };

/* Device texture: texels expanded at upload to one f4 {r,g,b,a} each (the /255.0f of 8-bit data, the channel
 * replication of 1-channel data and the alpha default of texture.c:32-63 are applied once, with the same IEEE
 * operations), rows stored so that texel (x, y) of textureGetPixelInternal is texels[first + x + y * width]. */
struct DTexture {
	uint32_t first;        /* index of the first texel in texels[] */
	uint32_t width, height;
	uint32_t m64w, m64h;   /* 2^64 mod width / height: (size_t)(negative int) % W of texture.c:37-38 in 32-bit math */
	uint32_t pad[3];
};

/* postfix program op: dst = op(src...) over a small operand file of f4 slots */
struct alignas(16) DOp {
	uint16_t kind;        /* enum crh_node_kind of the pure node, or CRH_OP_END */
	uint8_t  dst, s0, s1, s2;
	uint16_t pad;
	uint32_t u;           /* image index / math op / vec op */
	uint32_t cidx;        /* constants: index into consts[] */
};
#define CRH_OP_END 0xFFFFu
#define CRH_PROG_SLOTS 8

/* bsdf node compiled 1:1 from crh_gnode: a/b/c are bsdf indices or operand refs depending on kind */
struct alignas(16) DBsdf { uint32_t kind, a, b, c; };

struct DScene {
	const f4 *nodes;            /* 2 x f4 per node; node i of a BVH lives at device index base+1+i (child pairs 64-B aligned) */
	const f4 *tris;             /* 3 x f4 per BLAS prim slot: v0, e1, e2, n (poly.c:20-22 precomputed with the same fp32 ops) */
	const DShadeTri *shade;     /* per BLAS prim slot */
	const int32_t *prims;       /* prim_indices verbatim (BLAS slots -> polygon index in mesh): only crh_trace_rays reads it */
	const DInstance *instances; /* TLAS leaf order */
	const crh_material *materials;
	const DBsdf *bsdfs;         /* indexed by gnode index (only bsdf-kind entries are meaningful) */
	const f4 *consts;
	const DImage *images;
	const DOp *prog;
	const DTexture *textures;
	const f4 *texels;
	uint32_t tlas_root;         /* device index of the TLAS root's child pair (or of the root leaf if tlas_node_count == 1); for a WIDE walk (round 5, CRH_OPT_WALK) the launch passes
	                             * the ref of the top-level BVH's wide root here instead (tlas_node_count > 1) */
	uint32_t tlas_node_count;
	uint32_t tlas_prim_base;
	uint32_t background;        /* gnode index of the background bsdf */
	uint32_t shade_classes;     /* distinct shade classes among the instances (CRH_DINST_CLASS); <= 1: hits need no sorting */
	uint32_t instance_count;    /* records in instances[] */
	uint32_t image_count, texture_count;
	uint32_t material_count, bsdf_count, const_count;       /* records in materials[] / bsdfs[] / consts[] (small tables may be staged in LDS: loadMaterial / loadBsdf / loadConst) */
	uint32_t tlas_first;        /* device index of the first TLAS node after the root (the TLAS nodes tlas_first .. tlas_first + tlas_node_count - 2 are contiguous) */
	const crh_camera *camera;   /* (a record in memory rather than 26 words of kernel argument: only path generation reads it — scalar loads when a GEN step runs instead of scalar
	                             * registers that live, and spill, across the whole machine) */
};

/* Counter levels: 0 none, 1 rays + paths only (timed runs), 2 everything (parity / roofline runs).
 * The counter type also carries the compile-time switch `programs` = "rare features": kernels instantiated with programs = false
 * contain no call to runProgram() (a device function call in the persistent loop costs ~200 SGPR spills and
 * the callee's register budget) and no volume code (instance.c:62-92, 187-216: two extra walk states and a sampler draw inside
 * the traversal); the host picks that variant when the compiled scene has neither node programs nor volume instances. */
/* Round 5: the type carries a second compile-time switch, `wide` = the walk steps through the derived 4-ary copy of the BVHs (CRH_OPT_WALK = CRH_WALK_WIDE4, an option;
 * the binary walk is the contract). Instantiations without it are the code they were. */
template <int LEVEL, bool PROGRAMS, bool WIDE = false> struct CountersT;
template <bool PROGRAMS, bool WIDE> struct CountersT<2, PROGRAMS, WIDE> {
	static constexpr int level = 2;
	static constexpr bool programs = PROGRAMS;
	static constexpr bool wide = WIDE;
	uint32_t rays, node_tests, tri_tests, inst_visits, inst_hits, sphere_tests, tex_fetches, paths;
	uint32_t t_setup, t_trav, t_shade;   /* debug: wall-clock ticks (100 MHz) this wave spent per phase */
	uint32_t w_node, w_tri, w_ctrl, w_round, w_shade, w_setup;   /* debug: WAVE-level step counts by kind (lane 0 counts) */
	uint32_t u_node, u_shade;                                     /* debug: lanes served by the node / shade steps */
	uint32_t t_swap, t_gen, n_swap, n_gen, u_swap, u_tri, u_ctrl;                /* debug: swap / gen step clocks, counts, lanes moved by swaps */
#ifdef CRH_CENSUS
	uint32_t w_tri_in, u_tri_in, w_ctrl_in, u_ctrl_in;                           /* debug (rolling kernel): triangle / control steps served INSIDE node runs, and their lanes */
	uint32_t u_wait_tri, u_wait_fin;                                              /* debug (rolling kernel): summed over node steps, the lanes that sat the step out waiting for a triangle step / for a retire + refill */
#endif
};
template <bool PROGRAMS, bool WIDE> struct CountersT<1, PROGRAMS, WIDE> {
	static constexpr int level = 1;
	static constexpr bool programs = PROGRAMS;
	static constexpr bool wide = WIDE;
	uint32_t rays, paths;
};
typedef CountersT<2, true> Counters;
typedef CountersT<1, true> LiteCounters;
struct NoCounters { static constexpr int level = 0; static constexpr bool programs = true; static constexpr bool wide = false; };
template <class T> struct cnt_traits { static constexpr int level = T::level; static constexpr bool programs = T::programs; static constexpr bool wide = T::wide; };
template <class T> struct cnt_traits<T &> { static constexpr int level = T::level; static constexpr bool programs = T::programs; static constexpr bool wide = T::wide; };
/* CRH_COUNT: detailed counters (level 2); CRH_COUNT1: rays / paths (level >= 1) */
#define CRH_COUNT(c, field, n) do { if constexpr (crh::cnt_traits<decltype(c)>::level >= 2) (c).field += (n); } while (0)
#if defined(__HIPCC__)
#define CRH_TICK() ((uint32_t)wall_clock64())
#else
#define CRH_TICK() 0u
#endif
#define CRH_COUNT1(c, field, n) do { if constexpr (crh::cnt_traits<decltype(c)>::level >= 1) (c).field += (n); } while (0)

/* ---- vector.h / color.h ---------------------------------------------------------------------- */
CRH_DEV v3 vadd(v3 a, v3 b) { return v3{a.x + b.x, a.y + b.y, a.z + b.z}; }
CRH_DEV v3 vsub(v3 a, v3 b) { return v3{a.x - b.x, a.y - b.y, a.z - b.z}; }
CRH_DEV v3 vmul(v3 a, v3 b) { return v3{a.x * b.x, a.y * b.y, a.z * b.z}; }
CRH_DEV float vdot(v3 a, v3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
CRH_DEV v3 vscale(v3 v, float c) { return v3{v.x * c, v.y * c, v.z * c}; }
CRH_DEV v3 vcross(v3 a, v3 b) { return v3{(a.y * b.z) - (a.z * b.y), (a.z * b.x) - (a.x * b.z), (a.x * b.y) - (a.y * b.x)}; }
CRH_DEV float vlen(v3 v) { return sqrtf(vdot(v, v)); }
CRH_DEV v3 vnorm(v3 v) { float l = vlen(v); return v3{v.x / l, v.y / l, v.z / l}; }
CRH_DEV v3 vneg(v3 v) { return v3{-v.x, -v.y, -v.z}; }
CRH_DEV v3 vreflect(v3 I, v3 N) { return vsub(I, vscale(N, vdot(N, I) * 2.0f)); }
CRH_DEV float wrapMax(float x, float mx) { return fmodf(mx + fmodf(x, mx), mx); }
CRH_DEV float wrapMinMax(float x, float mn, float mx) { return mn + wrapMax(x - mn, mx - mn); }
/* wrapMinMax(x, 0, 1) (vector.h:215-221) without the two fmodf for -1 <= x < 1 (every texture coordinate the shipped scenes produce; an fmodf is
 * ~150 instructions, and half of all background look-ups have a negative v). 0 <= x < 1: fmodf(x, 1) = x exactly, 1 + x rounds once into [1, 2], and
 * fmodf(that, 1) is exact (that - 1, or 0 when the sum rounded up to 2). -1 <= x < 0: fmodf(x, 1) = x again (-0 for x = -1), 1 + x rounds once into
 * [0, 1], and fmodf(that, 1) is that, or 0 when the sum rounded up to 1. Same bits (tests/test_exact_math.py: every float of both ranges). */
CRH_DEV float wrap01(float x) {
	if (x >= 0.0f && x < 1.0f) { const float y = 1.0f + x; return 0.0f + (y < 2.0f ? y - 1.0f : 0.0f); }
	if (x >= -1.0f && x < 0.0f) { const float y = 1.0f + x; return 0.0f + (y < 1.0f ? y : 0.0f); }
	return wrapMinMax(x, 0.0f, 1.0f);
}
CRH_DEV float rmin(float a, float b) { return a < b ? a : b; }     /* includes.h:20 */
CRH_DEV float rmax(float a, float b) { return a > b ? a : b; }     /* includes.h:21 */

CRH_DEV rgba cmul(rgba a, rgba b) { return rgba{a.r * b.r, a.g * b.g, a.b * b.b, a.a * b.a}; }
CRH_DEV rgba cadd(rgba a, rgba b) { return rgba{a.r + b.r, a.g + b.g, a.b + b.b, a.a + b.a}; }
CRH_DEV rgba ccoef(float c, rgba a) { return rgba{a.r * c, a.g * c, a.b * c, a.a * c}; }
CRH_DEV rgba cmix(rgba c1, rgba c2, float k) { return cadd(ccoef(1.0f - k, c1), ccoef(k, c2)); }   /* color.h:46 */
CRH_DEV float linearToSRGB(float c) {                                                               /* color.h:51 */
	if (c <= 0.0031308f) return 12.92f * c;
	return (1.055f * em::powf_(c, 0.4166666667f)) - 0.055f;
}
CRH_DEV float SRGBToLinear(float c) {                                                               /* color.h:59 */
	if (c <= 0.04045f) return c / 12.92f;
	return em::powf_(((c + 0.055f) / 1.055f), 2.4f);
}
/* color.h:37-40: 0.587 is a double constant, so the sum is carried in double */
CRH_DEV float grayscaleOf(rgba c) {
	return sqrtf((float)(0.299f * em::powf_(c.r, 2.0f) + 0.587 * (double)em::powf_(c.g, 2.0f) + (double)(0.114f * em::powf_(c.b, 2.0f))));
}
/* color.c:27-70 */
CRH_DEV rgba colorForKelvin(float kelvin) {
	float r, g, b;
	float temp = kelvin >= 40000.0f ? 40000.0f : kelvin;
	temp = temp / 100.0f;
	if (temp <= 66.0f) {
		r = 255.0f;
	} else {
		r = temp - 60.0f;
		r = 329.698727446f * em::powf_(r, -0.1332047592f);
		r = r < 0.0f ? 0.0f : r;
		r = r > 255.0f ? 255.0f : r;
	}
	if (temp <= 66.0f) {
		g = temp;
		g = 99.4708025861f * em::logf_(g) - 161.1195681661f;
	} else {
		g = temp - 60.0f;
		g = 288.1221695283f * em::powf_(g, -0.0755148492f);
	}
	g = g < 0.0f ? 0.0f : g;
	g = g > 255.0f ? 255.0f : g;
	if (temp >= 66.0f) {
		b = 255.0f;
	} else if (temp <= 19.0f) {
		b = 0.0f;
	} else {
		b = temp - 10.0f;
		b = 138.5177312231f * em::logf_(b) - 305.0447927307f;
		b = b < 0.0f ? 0.0f : b;
		b = b > 255.0f ? 255.0f : b;
	}
	return rgba{r / 255.0f, g / 255.0f, b / 255.0f, 0.0f};
}

CRH_DEV uint32_t asU32(float f) { union { float f; uint32_t u; } c; c.f = f; return c.u; }
CRH_DEV float asF32(uint32_t u) { union { float f; uint32_t u; } c; c.u = u; return c.f; }

/* ---- samplers: sampler.c:31-58. KIND 0 = Random (pcg_basic.c:42-68, samplers/common.h:22-27, random.c:12-21): the
 * sampler of renderThread. KIND 1 = Halton (halton.c:16-31, common.h:14-57): the sampler of renderThreadInteractive.
 * The kind is a compile-time parameter of everything that draws (no branch per draw); both keep their state in 64 bits. */
template <int KIND> struct RngT { uint64_t state; };
typedef RngT<0> Rng;              /* PCG32: state; inc is always 1 (stream 0) */
typedef RngT<1> HaltonRng;        /* bits 0..31 rndOffset (float bits), 32..39 currPrime mod 6, 40..63 currPass */
CRH_DEV uint32_t pcg32_next(Rng &r) {
	uint64_t old = r.state;
	r.state = old * 6364136223846793005ULL + 1ULL;
	uint32_t xs = (uint32_t)(((old >> 18u) ^ old) >> 27u);
	uint32_t rot = (uint32_t)(old >> 59u);
	return (xs >> rot) | (xs << ((0u - rot) & 31u));
}
CRH_DEV uint64_t hash64(uint64_t x) {
	x = (x ^ (x >> 30)) * 0xbf58476d1ce4e5b9ULL;
	x = (x ^ (x >> 27)) * 0x94d049bb133111ebULL;
	x = x ^ (x >> 31);
	return x;
}
CRH_DEV uint32_t hash32(uint32_t x) {                       /* common.h:14-20 */
	x = (x ^ 12345391u) * 2654435769u;
	x ^= (x << 6) ^ (x >> 26);
	x *= 2654435769u;
	x += (x << 5) ^ (x >> 12);
	return x;
}
CRH_DEV float radicalInverse(int pass, int base) {          /* common.h:34-46 */
	const float invBase = 1.0f / (float)base;
	int reversedDigits = 0;
	float invBaseN = 1.0f;
	while (pass) {
		const int next = pass / base;
		const int digit = pass - base * next;
		reversedDigits = reversedDigits * base + digit;
		invBaseN *= invBase;
		pass = next;
	}
	return rmin((float)reversedDigits * invBaseN, 0.99999994f);
}
CRH_DEV void initSampler(Rng &r, int pass, int maxPasses, uint32_t pixelIndex) {
	uint32_t key = pixelIndex * (uint32_t)maxPasses + (uint32_t)pass;   /* 32-bit wrap: sampler.c:42 */
	r.state = 0u;
	pcg32_next(r);
	r.state += hash64((uint64_t)key);
	pcg32_next(r);
}
/* renderThreadInteractive seeds with state.finishedPasses, which starts at 1 (renderer.c:204, 333): pass p (0-based, like
 * completedSamples - 1 everywhere else here) is Halton index p + 1. uintToUnitReal: common.h:48-57. */
CRH_DEV void initSampler(HaltonRng &r, int pass, int, uint32_t pixelIndex) {
	const uint32_t offsetBits = asU32(asF32((hash32(pixelIndex) >> 9) | 0x3f800000u) - 1.0f);
	r.state = (uint64_t)offsetBits | ((uint64_t)(uint32_t)(pass + 1) << 40);
}
CRH_DEV float getDimension(Rng &r) { return (1.0f / 4294967296.0f) * (float)pcg32_next(r); }
CRH_DEV float getDimension(HaltonRng &r) {                  /* halton.c:25-31 */
	const uint32_t prime = (uint32_t)(r.state >> 32) & 0xFFu;
	const int base = prime == 0u ? 2 : prime == 1u ? 3 : prime == 2u ? 5 : prime == 3u ? 7 : prime == 4u ? 11 : 13;
	r.state = (r.state & ~(0xFFull << 32)) | ((uint64_t)(prime == 5u ? 0u : prime + 1u) << 32);
	const float u = radicalInverse((int)(r.state >> 40), base), v = asF32((uint32_t)r.state);
	return (u + v < 1.0f) ? u + v : u + v - 1.0f;           /* wrapAdd, common.h:30-32 */
}

/* vector.h:190-198 */
template <class R>
CRH_DEV v2 randomCoordOnUnitDisc(R &rng) {
	float r = sqrtf(getDimension(rng));
	float theta = ((getDimension(rng)) * ((2.0f * CRH_PI) - 0.0f)) + 0.0f;
	float sn, cs;
	em::sincosf_(theta, sn, cs);
	return v2{r * cs, r * sn};
}
/* vector.h:243-249 */
template <class R>
CRH_DEV v3 randomOnUnitSphere(R &rng) {
	const float sample_x = getDimension(rng);
	const float sample_y = getDimension(rng);
	const float a = sample_x * (2.0f * CRH_PI);
	const float s = 2.0f * sqrtf(rmax(0.0f, sample_y * (1.0f - sample_y)));
	float sn, cs;
	em::sincosf_(a, sn, cs);
	return v3{cs * s, sn * s, 1.0f - 2.0f * sample_y};
}
/* vector.h:251-266 */
CRH_DEV bool refract(v3 in, v3 normal, float niOverNt, v3 &refracted) {
	const v3 uv = vnorm(in);
	const float dt = vdot(uv, normal);
	const float discriminant = 1.0f - niOverNt * niOverNt * (1.0f - dt * dt);
	if (discriminant > 0.0f) {
		const v3 A = vscale(normal, dt);
		const v3 B = vsub(uv, A);
		const v3 C = vscale(B, niOverNt);
		const v3 D = vscale(normal, sqrtf(discriminant));
		refracted = vsub(C, D);
		return true;
	}
	return false;
}
/* vector.h:268-272 */
CRH_DEV float schlick(float cosine, float IOR) {
	float r0 = (1.0f - IOR) / (1.0f + IOR);
	r0 = r0 * r0;
	return r0 + (1.0f - r0) * em::powf_((1.0f - cosine), 5.0f);
}

/* ---- transforms.c:76-116 on 3x4 row-major matrices --------------------------------------------- */
CRH_DEV v3 xfPoint(v3 v, const float *m) {
	return v3{(m[0] * v.x) + (m[1] * v.y) + (m[2] * v.z) + m[3],
			  (m[4] * v.x) + (m[5] * v.y) + (m[6] * v.z) + m[7],
			  (m[8] * v.x) + (m[9] * v.y) + (m[10] * v.z) + m[11]};
}
CRH_DEV v3 xfVector(v3 v, const float *m) {
	return v3{(m[0] * v.x) + (m[1] * v.y) + (m[2] * v.z),
			  (m[4] * v.x) + (m[5] * v.y) + (m[6] * v.z),
			  (m[8] * v.x) + (m[9] * v.y) + (m[10] * v.z)};
}
CRH_DEV v3 xfVectorT(v3 v, const float *m) {
	return v3{(m[0] * v.x) + (m[4] * v.y) + (m[8] * v.z),
			  (m[1] * v.x) + (m[5] * v.y) + (m[9] * v.z),
			  (m[2] * v.x) + (m[6] * v.y) + (m[10] * v.z)};
}
CRH_DEV v3 alongRay(v3 o, v3 d, float t) { return vadd(o, vscale(d, t)); }   /* lightray.h:31 */

/* ---- instance records: one 64-byte LINE at a time -------------------------------------------------
 * Line 0 = what entering an instance needs (Ainv rows, then kind / root / ray_offset / radius), line 1 = what finishing a hit needs (A rows,
 * then orig / poly_base / material / density). The lines come from the scene's array in global memory — or from wherever the caller's
 * "hot table source" keeps a copy: k_pathtrace stages the records of scenes with few instances in LDS (cray_hip.hip: LdsStack::instLine;
 * the vector L1 is the walk's scarcest resource, and an instance visit is four 16-byte look-ups per lane that always hit the same few lines).
 * A source is any object with instLine(S, idx, line); everything else (host emulation, crh_trace_rays, the other kernel forms) reads global memory. */
struct InstLine { f4 a, b, c, d; };
CRH_DEV uint32_t instKind(const InstLine &l0) { return asU32(l0.d.x); }
CRH_DEV uint32_t instRoot(const InstLine &l0) { return asU32(l0.d.y); }
CRH_DEV float instRayOffset(const InstLine &l0) { return l0.d.z; }
CRH_DEV float instRadius(const InstLine &l0) { return l0.d.w; }
CRH_DEV uint32_t instOrig(const InstLine &l1) { return asU32(l1.d.x); }
CRH_DEV uint32_t instPolyBase(const InstLine &l1) { return asU32(l1.d.y); }
CRH_DEV uint32_t instMaterial(const InstLine &l1) { return asU32(l1.d.z); }
CRH_DEV float instDensity(const InstLine &l1) { return l1.d.w; }
struct GlobalTables {};          /* "no hot tables": every record comes from global memory */
template <class T, class = void> struct has_inst_line : std::false_type {};
template <class T> struct has_inst_line<T, std::void_t<decltype(std::declval<const T &>().instLine(std::declval<const DScene &>(), 0, 0))>> : std::true_type {};
template <class Src>
CRH_DEV InstLine instLine(const DScene &S, const Src &src, int32_t idx, int line) {
	if constexpr (has_inst_line<Src>::value) return src.instLine(S, idx, line);
	else {
		(void)src;
		const f4 *g = (const f4 *)(S.instances + idx) + 4 * line;
		return InstLine{g[0], g[1], g[2], g[3]};
	}
}
/* The shading tables — materials, bsdf nodes, constants: a few dozen records per scene, read through a chain of dependent look-ups by every shaded hit
 * (material -> bsdf node -> [mix: child node ->] operand constant). A hot-table source with bsdfNode() serves all three (k_pathtrace: LDS). */
template <class T, class = void> struct has_shade_tables : std::false_type {};
template <class T> struct has_shade_tables<T, std::void_t<decltype(std::declval<const T &>().bsdfNode(std::declval<const DScene &>(), 0u))>> : std::true_type {};
template <class Src> CRH_DEV DBsdf loadBsdf(const DScene &S, const Src &src, uint32_t i) {
	if constexpr (has_shade_tables<Src>::value) return src.bsdfNode(S, i); else { (void)src; return S.bsdfs[i]; }
}
template <class Src> CRH_DEV f4 loadConst(const DScene &S, const Src &src, uint32_t i) {
	if constexpr (has_shade_tables<Src>::value) return src.constant(S, i); else { (void)src; return S.consts[i]; }
}
/* an image operand: its descriptor and its texture's (zeros when the image has none: evalImage does not read them then) */
struct ImageRef { DImage im; DTexture t; };
template <class Src> CRH_DEV ImageRef loadImage(const DScene &S, const Src &src, uint32_t i) {
	if constexpr (has_shade_tables<Src>::value) return src.image(S, i);
	else { (void)src; const DImage im = S.images[i]; return ImageRef{im, im.tex == CRH_NONE ? DTexture{} : S.textures[im.tex]}; }
}
template <class Src> CRH_DEV crh_material loadMaterial(const DScene &S, const Src &src, uint32_t i) {
	if constexpr (has_shade_tables<Src>::value) return src.material(S, i); else { (void)src; return S.materials[i]; }
}
/* transforms.c:76-116 on the three rows of a line (the same operations in the same order as xfPoint / xfVector / xfVectorT on float[12]) */
CRH_DEV v3 xfPoint(v3 v, const InstLine &m) {
	return v3{(m.a.x * v.x) + (m.a.y * v.y) + (m.a.z * v.z) + m.a.w,
			  (m.b.x * v.x) + (m.b.y * v.y) + (m.b.z * v.z) + m.b.w,
			  (m.c.x * v.x) + (m.c.y * v.y) + (m.c.z * v.z) + m.c.w};
}
CRH_DEV v3 xfVector(v3 v, const InstLine &m) {
	return v3{(m.a.x * v.x) + (m.a.y * v.y) + (m.a.z * v.z),
			  (m.b.x * v.x) + (m.b.y * v.y) + (m.b.z * v.z),
			  (m.c.x * v.x) + (m.c.y * v.y) + (m.c.z * v.z)};
}
CRH_DEV v3 xfVectorT(v3 v, const InstLine &m) {
	return v3{(m.a.x * v.x) + (m.b.x * v.y) + (m.c.x * v.z),
			  (m.a.y * v.x) + (m.b.y * v.y) + (m.c.y * v.z),
			  (m.a.z * v.x) + (m.b.z * v.y) + (m.c.z * v.z)};
}

/* ---- camera.c:46-87 ---------------------------------------------------------------------------- */
CRH_DEV float triangleDistribution(float v) {
	const float orig = v * 2.0f - 1.0f;
	v = orig / sqrtf(fabsf(orig));
	v = rmin(rmax(v, -1.0f), 1.0f);          /* clamp(): vector.h:55 = min(max(value, min), max) */
	v = v - ((orig >= 0.0f) ? 1.0f : -1.0f);
	return v;
}
template <class R>
CRH_DEV void getCameraRay(const crh_camera &cam, R &rng, int x, int y, v3 &ro, v3 &rd) {
	const v3 right = v3{cam.right[0], cam.right[1], cam.right[2]};
	const v3 up = v3{cam.up[0], cam.up[1], cam.up[2]};
	const v3 forward = v3{cam.forward[0], cam.forward[1], cam.forward[2]};
	v3 start = v3{0.0f, 0.0f, 0.0f};
	const float jitterX = triangleDistribution(getDimension(rng));
	const float jitterY = triangleDistribution(getDimension(rng));
	v3 pixX = vscale(right, (cam.sensor[0] / (float)cam.width));
	v3 pixY = vscale(up, (cam.sensor[1] / (float)cam.height));
	v3 pixV = vadd(forward, vadd(vscale(pixX, (float)x - (float)cam.width * 0.5f + jitterX + 0.5f),
								 vscale(pixY, (float)y - (float)cam.height * 0.5f + jitterY + 0.5f)));
	v3 dir = vnorm(pixV);
	if (cam.aperture > 0.0f) {
		float ft = cam.focal_distance / vdot(dir, forward);
		v3 focusPoint = alongRay(start, dir, ft);
		v2 disc = randomCoordOnUnitDisc(rng);
		v2 lensPoint = v2{disc.x * cam.aperture, disc.y * cam.aperture};
		start = vadd(start, vadd(vscale(right, lensPoint.x), vscale(up, lensPoint.y)));
		dir = vnorm(vsub(focusPoint, start));
	}
	ro = xfPoint(start, cam.A);
	rd = xfVector(dir, cam.A);
}

/* ---- textures: texture.c:32-79 ----------------------------------------------------------------- */
/* (size_t)i % W for a possibly negative int i: the reference sign-extends to 64 bits, so a negative i wraps
 * as (2^64 - |i|) % W = (m64 + W - |i| % W) % W with m64 = 2^64 mod W — all in 32-bit arithmetic here. */
CRH_DEV uint32_t wrapIndex(int i, uint32_t W, uint32_t m64) {
	if ((W & (W - 1u)) == 0u) return (uint32_t)i & (W - 1u);      /* power of two: the two's-complement low bits ARE the 64-bit modulo */
	if (i >= 0) return (uint32_t)i % W;
	const uint32_t a = (uint32_t)(-(int64_t)i) % W;
	return (m64 + W - a) % W;
}
struct TexCtx { const DTexture *textures; const f4 *texels; };
template <class Cnt>
CRH_DEV rgba texel(const TexCtx S, const DTexture &t, uint32_t x, uint32_t y, Cnt &cnt) {
	CRH_COUNT(cnt, tex_fetches, 1);
	const f4 o = *(const f4 *)((const char *)S.texels + (uint32_t)((t.first + x + y * t.width) << 4));      /* base + 32-bit byte offset: see stepNode */
	return rgba{o.x, o.y, o.z, o.w};
}
template <class Cnt>
CRH_DEV rgba textureGetPixelFiltered(const TexCtx S, const DTexture &t, float x, float y, Cnt &cnt) {
	x = x * (float)t.width;
	y = y * (float)t.height;
	float xcopy = x - 0.5f;
	float ycopy = y - 0.5f;
	int xint = (int)xcopy;
	int yint = (int)ycopy;
	const uint32_t x0 = wrapIndex(xint, t.width, t.m64w), x1 = wrapIndex(xint + 1, t.width, t.m64w);
	const uint32_t y0 = wrapIndex(yint, t.height, t.m64h), y1 = wrapIndex(yint + 1, t.height, t.m64h);
	rgba topleft = texel(S, t, x0, y0, cnt);
	rgba topright = texel(S, t, x1, y0, cnt);
	rgba botleft = texel(S, t, x0, y1, cnt);
	rgba botright = texel(S, t, x1, y1, cnt);
	const float fx = xcopy - (float)xint, fy = ycopy - (float)yint;
	return cmix(cmix(topleft, topright, fx), cmix(botleft, botright, fx), fy);
}
/* image.c:31-48 */
/* (t = the image's texture descriptor, read by the caller: from textures[] or from a hot-table source's copy; unused when the image has no texture) */
template <class Cnt>
CRH_DEV rgba evalImage(const TexCtx S, const DImage im, const DTexture t, v2 uv, Cnt &cnt) {
	if (im.tex == CRH_NONE) return rgba{1.0f, 0.0f, 0.5f, 1.0f};   /* warningMaterial().diffuse, material.c:38 */
	rgba out;
	if (im.options & CRH_IMAGE_NO_BILINEAR) {
		float x = uv.x * (float)t.width;
		float y = uv.y * (float)t.height;
		/* (size_t)x % W: exact for any finite non-negative float via fmodf (a negative x is undefined in C) */
		const uint32_t xi = (uint32_t)fmodf(truncf(x), (float)t.width) % t.width;
		const uint32_t yi = (uint32_t)fmodf(truncf(y), (float)t.height) % t.height;
		out = texel(S, t, xi, yi, cnt);
	} else {
		out = textureGetPixelFiltered(S, t, uv.x, uv.y, cnt);
	}
	if (im.options & CRH_IMAGE_SRGB_TRANSFORM) {       /* color.h:66-73 on r, g, b: the channels rotate through ONE inlined powf */
		/* a grey sample (the three channels hold the same bits: grids, masks, most of a typical albedo map's filtered texels) is transformed once —
		 * the same function of the same bits — instead of three times: an exact powf is ~140 instructions of mostly double arithmetic */
		const bool grey = asU32(out.r) == asU32(out.g) && asU32(out.g) == asU32(out.b);
		const int n = grey ? 1 : 3;
#if defined(__HIPCC__)
#pragma clang loop unroll(disable)
#endif
		for (int i = 0; i < n; ++i) { const float t = SRGBToLinear(out.r); out.r = out.g; out.g = out.b; out.b = t; }
		if (grey) { out.r = out.b; out.g = out.b; }
	}
	return out;
}

/* ---- shading record (struct hitRecord, hitrecord.h:14-23, as the nodes read it) --------------- */
struct ShadeRec {
	v3 dir;        /* incident.direction */
	v3 point;      /* hitPoint           */
	v3 normal;     /* surfaceNormal      */
	v2 uv;
	float distance;
	float ior;     /* material.IOR       */
};

/* gradient.c:40-45 */
CRH_DEV rgba evalGradient(const f4 *consts, uint32_t cidx, const ShadeRec &rec) {
	v3 unitDir = vnorm(rec.dir);
	float t = 0.5f * (unitDir.y + 1.0f);
	const f4 dn = consts[cidx], up = consts[cidx + 1];
	return cadd(ccoef(1.0f - t, rgba{dn.x, dn.y, dn.z, dn.w}), ccoef(t, rgba{up.x, up.y, up.z, up.w}));
}
template <class Src>
CRH_DEV rgba evalGradient(const DScene &S, const Src &src, uint32_t cidx, const ShadeRec &rec) {
	v3 unitDir = vnorm(rec.dir);
	float t = 0.5f * (unitDir.y + 1.0f);
	const f4 dn = loadConst(S, src, cidx), up = loadConst(S, src, cidx + 1);
	return cadd(ccoef(1.0f - t, rgba{dn.x, dn.y, dn.z, dn.w}), ccoef(t, rgba{up.x, up.y, up.z, up.w}));
}
/* ---- pure nodes (colour / value / vector): postfix programs compiled at upload ---------------- */
/* Out-of-line on the device (rare graphs only: checker, grayscale(image), ...). Everything is passed and
 * returned BY VALUE: a reference parameter would pin the caller's scene / hit record / counters in scratch. */
struct ProgCtx { const f4 *consts; const DImage *images; const DOp *prog; TexCtx tex; };
struct ProgResult { f4 v; uint32_t fetches; };
struct FetchCounter { static constexpr int level = 2; static constexpr bool programs = true; uint32_t tex_fetches; };
CRH_DEV ProgResult runProgram(const ProgCtx S, uint32_t pc, const ShadeRec rec) {
	f4 slot[CRH_PROG_SLOTS];
	FetchCounter cnt;
	cnt.tex_fetches = 0;
	for (;;) {
		const DOp op = S.prog[pc++];
		if (op.kind == CRH_OP_END) return ProgResult{slot[op.s0], cnt.tex_fetches};
		const f4 a = slot[op.s0], b = slot[op.s1], c = slot[op.s2];
		f4 r = f4{0.0f, 0.0f, 0.0f, 0.0f};
		switch (op.kind) {
			case CRH_COLOR_CONSTANT: case CRH_VALUE_CONSTANT: case CRH_VEC_CONSTANT:
				r = S.consts[op.cidx]; break;
			case CRH_COLOR_IMAGE: {
				const DImage im = S.images[op.u];
				rgba o = evalImage(S.tex, im, im.tex == CRH_NONE ? DTexture{} : S.tex.textures[im.tex], rec.uv, cnt);
				r = f4{o.r, o.g, o.b, o.a}; break;
			}
			case CRH_COLOR_CHECKER: {          /* checker.c:31-54; a=A b=B c=scale (all already evaluated: pure) */
				const float coef = c.x;
				float sines;
				if (rec.uv.x >= 0.0f) sines = em::sinf_(coef * rec.uv.x) * em::sinf_(coef * rec.uv.y);
				else sines = em::sinf_(coef * rec.point.x) * em::sinf_(coef * rec.point.y) * em::sinf_(coef * rec.point.z);
				r = sines < 0.0f ? a : b; break;
			}
			case CRH_COLOR_GRADIENT: { rgba o = evalGradient(S.consts, op.cidx, rec); r = f4{o.r, o.g, o.b, o.a}; break; }
			case CRH_COLOR_BLACKBODY: { rgba o = colorForKelvin(a.x); r = f4{o.r, o.g, o.b, o.a}; break; }
			case CRH_COLOR_COMBINE: r = f4{a.x, a.x, a.x, 1.0f}; break;
			case CRH_COLOR_COMBINERGB: r = f4{a.x, b.x, c.x, 1.0f}; break;
			case CRH_COLOR_VECTOCOLOR: r = f4{a.x, a.y, a.z, 0.0f}; break;
			case CRH_VALUE_ALPHA: r.x = a.w; break;
			case CRH_VALUE_GRAYSCALE: r.x = grayscaleOf(rgba{a.x, a.y, a.z, a.w}); break;
			case CRH_VALUE_RAYLENGTH: r.x = rec.distance; break;
			case CRH_VALUE_FRESNEL: {          /* fresnel.c:38-51 */
				const float IOR = a.x;
				float cosine;
				if (vdot(rec.dir, rec.normal) > 0.0f) cosine = IOR * vdot(rec.dir, rec.normal) / vlen(rec.dir);
				else cosine = -(vdot(rec.dir, rec.normal) / vlen(rec.dir));
				r.x = schlick(cosine, IOR); break;
			}
			case CRH_VALUE_MATH: {             /* math.c:42-95 */
				const float x = a.x, y = b.x;
				switch (op.u) {
					case 0: r.x = x + y; break;
					case 1: r.x = x - y; break;
					case 2: r.x = x * y; break;
					case 3: r.x = x / y; break;
					case 4: r.x = em::powf_(x, y); break;
					case 5: r.x = em::log10f_(x); break;
					case 6: r.x = sqrtf(x); break;
					case 7: r.x = fabsf(x); break;
					case 8: r.x = rmin(x, y); break;
					case 9: r.x = rmax(x, y); break;
					case 10: r.x = em::sinf_(x); break;
					case 11: r.x = em::cosf_(x); break;
					case 12: r.x = em::tanf_(x); break;
					case 13: r.x = (x * CRH_PI) / 180.0f; break;
					case 14: r.x = x * (180.0f / CRH_PI); break;
					default: break;
				}
				break;
			}
			case CRH_VEC_NORMAL: r = f4{rec.normal.x, rec.normal.y, rec.normal.z, 0.0f}; break;
			case CRH_VEC_VECMATH: {            /* vecmath.c:42-81 (dot / length return their float in .f, the vector is zero) */
				const v3 x = v3{a.x, a.y, a.z}, y = v3{b.x, b.y, b.z};
				v3 o = v3{0.0f, 0.0f, 0.0f};
				switch (op.u) {
					case 0: o = vadd(x, y); break;
					case 1: o = vsub(x, y); break;
					case 2: o = vmul(x, y); break;
					case 3: o = vscale(vadd(x, y), 0.5f); break;
					case 5: o = vcross(x, y); break;
					case 6: o = vnorm(x); break;
					case 7: o = vreflect(x, y); break;
					case 9: o = v3{fabsf(x.x), fabsf(x.y), fabsf(x.z)}; break;
					default: break;
				}
				r = f4{o.x, o.y, o.z, 0.0f}; break;
			}
			default: break;
		}
		slot[op.dst] = r;
	}
}

CRH_DEV ProgCtx progCtx(const DScene &S) { return ProgCtx{S.consts, S.images, S.prog, TexCtx{S.textures, S.texels}}; }
template <class Cnt, class Src = GlobalTables>
CRH_DEV rgba evalColor(const DScene &S, uint32_t opr, const ShadeRec &rec, Cnt &cnt, const Src &src = Src()) {
	const uint32_t k = CRH_OPR_KIND(opr), i = CRH_OPR_IDX(opr);
	if (k == CRH_OPR_CONST) { const f4 c = loadConst(S, src, i); return rgba{c.x, c.y, c.z, c.w}; }
	if (k == CRH_OPR_IMAGE) { const ImageRef ir = loadImage(S, src, i); return evalImage(TexCtx{S.textures, S.texels}, ir.im, ir.t, rec.uv, cnt); }
	if (k == CRH_OPR_GRADIENT) return evalGradient(S, src, i, rec);
	if constexpr (cnt_traits<Cnt>::programs) {
		const ProgResult r = runProgram(progCtx(S), i, rec);
		CRH_COUNT(cnt, tex_fetches, r.fetches);
		return rgba{r.v.x, r.v.y, r.v.z, r.v.w};
	}
	return rgba{0.0f, 0.0f, 0.0f, 0.0f};
}
template <class Cnt, class Src = GlobalTables>
CRH_DEV float evalValue(const DScene &S, uint32_t opr, const ShadeRec &rec, Cnt &cnt, const Src &src = Src()) {
	const uint32_t k = CRH_OPR_KIND(opr), i = CRH_OPR_IDX(opr);
	if (k == CRH_OPR_CONST) return loadConst(S, src, i).x;
	if (k == CRH_OPR_IMAGE_ALPHA) { const ImageRef ir = loadImage(S, src, i); return evalImage(TexCtx{S.textures, S.texels}, ir.im, ir.t, rec.uv, cnt).a; }
	if constexpr (cnt_traits<Cnt>::programs) {
		const ProgResult r = runProgram(progCtx(S), i, rec);
		CRH_COUNT(cnt, tex_fetches, r.fetches);
		return r.v.x;
	}
	return 0.0f;
}

/* ---- bsdf nodes: src/nodes/shaders ----------------------------------------------------------- */
struct BsdfSample { v3 out; float r, g, b; };   /* bsdfnode.h:19-23; colour alpha never reaches RGB (pathtrace.c:51-57) */
#define CRH_ADD_DEPTH 4

template <class R, class Cnt, class Src = GlobalTables>
CRH_DEV BsdfSample sampleBsdf(const DScene &S, uint32_t root, const ShadeRec &rec, R &rng, Cnt &cnt, const Src &src = Src()) {
	uint32_t addStack[CRH_ADD_DEPTH];      /* pending add.c frames: gnode index | (A done ? 1<<31 : 0) */
	BsdfSample resStack[CRH_ADD_DEPTH];
	int asp = 0, rsp = 0;
	uint32_t cur = root;
	for (;;) {
		const DBsdf n = loadBsdf(S, src, cur);
		const uint32_t kind = n.kind;
		BsdfSample res;
		res.out = v3{0.0f, 0.0f, 0.0f};
		if (kind == CRH_BSDF_ADD) {           /* add.c:42-49: A fully, then B fully */
			if (asp < CRH_ADD_DEPTH) addStack[asp++] = cur;
			cur = n.a;
			continue;
		}
		/* Operands are pure functions of the hit, so each CLASS of operand is evaluated at ONE place for all node kinds (one inlined copy
		 * of the texture fetch / sRGB transform / program interpreter instead of a dozen), still only when the reference would evaluate it,
		 * and the sampler draws keep the reference's order. */
		float vc = 0.0f;                      /* value operand c: mix factor (mix.c:45), glass IOR (glass.c:47) */
		if (kind == CRH_BSDF_MIX || kind == CRH_BSDF_GLASS) vc = evalValue(S, n.c, rec, cnt, src);
		if (kind == CRH_BSDF_MIX) {           /* mix.c:42-50 */
			cur = (getDimension(rng) > vc) ? n.a : n.b;
			continue;
		}
		/* the dielectric interface of glass (IOR from its node) and plastic (the material's IOR): glass.c:48-66, plastic.c:47-65 */
		float reflectionProbability = 1.0f;
		v3 refracted = v3{0.0f, 0.0f, 0.0f};
		if (kind == CRH_BSDF_GLASS || kind == CRH_BSDF_PLASTIC) {
			const float IOR = (kind == CRH_BSDF_GLASS) ? vc : rec.ior;
			v3 outwardNormal;
			float niOverNt, cosine;
			if (vdot(rec.dir, rec.normal) > 0.0f) {
				outwardNormal = vneg(rec.normal);
				niOverNt = IOR;
				cosine = IOR * vdot(rec.dir, rec.normal) / vlen(rec.dir);
			} else {
				outwardNormal = rec.normal;
				niOverNt = 1.0f / IOR;
				cosine = -(vdot(rec.dir, rec.normal) / vlen(rec.dir));
			}
			if (refract(rec.dir, outwardNormal, niOverNt, refracted)) reflectionProbability = schlick(cosine, IOR);
			else reflectionProbability = 1.0f;
		}
		if (kind == CRH_BSDF_PLASTIC && !(getDimension(rng) < reflectionProbability)) {      /* plastic.c:66-86: the diffuse layer below the coat */
			cur = n.c;
			continue;
		}
		float vb = 0.0f;                      /* value operand b: roughness (metal.c:44, glass.c:67), emission strength (emission.c:46) */
		if (kind == CRH_BSDF_METAL || kind == CRH_BSDF_GLASS || kind == CRH_BSDF_EMISSION) vb = evalValue(S, n.b, rec, cnt, src);
		rgba col = rgba{0.0f, 0.0f, 0.0f, 0.0f};      /* the colour operand: a, for the plastic coat its roughness b (plastic.c:68) */
		if (kind >= CRH_BSDF_DIFFUSE && kind <= CRH_BSDF_ISOTROPIC) col = evalColor(S, kind == CRH_BSDF_PLASTIC ? n.b : n.a, rec, cnt, src);
		/* The outgoing direction. A batch of hits runs the union of its lanes' code, so what several kinds compute is computed at ONE site each
		 * (round 3: randomOnUnitSphere was inlined six times, vnorm four times, vreflect three times): the random unit vector — the first draw of
		 * every kind that uses it, so the sampler's order is the reference's —, one normalisation, one reflection. Per kind:
		 *   plastic coat (plastic.c:66-72)  reflect(dir, n) [+ sphere * roughness]                colour white
		 *   diffuse      (diffuse.c:40-47)  norm(n + sphere)
		 *   metal        (metal.c:40-55)    reflect(norm(dir), n) [+ sphere * roughness]
		 *   glass        (glass.c:41-87)    reflect(dir, n) [+ fuzz] or refracted [+ fuzz], fuzz = sphere * roughness, chosen by one more draw
		 *   transparent  (transparent.c:40-44) dir
		 *   emission     (emission.c:42-49) norm(n + sphere)                                       colour * strength
		 *   isotropic    (isotropic.c:40-47) norm(sphere) */
		const bool lambert = kind == CRH_BSDF_DIFFUSE || kind == CRH_BSDF_EMISSION, iso = kind == CRH_BSDF_ISOTROPIC;
		const bool mirror = kind == CRH_BSDF_PLASTIC || kind == CRH_BSDF_METAL || kind == CRH_BSDF_GLASS;
		const float rough = kind == CRH_BSDF_PLASTIC ? col.r : vb;         /* read by the mirror kinds only */
		const bool fuzzy = mirror && rough > 0.0f;
		v3 sph = v3{0.0f, 0.0f, 0.0f};
		if (lambert || iso || fuzzy) sph = randomOnUnitSphere(rng);
		v3 unit = v3{0.0f, 0.0f, 0.0f};
		if (lambert || iso || kind == CRH_BSDF_METAL) unit = vnorm(kind == CRH_BSDF_METAL ? rec.dir : (lambert ? vadd(rec.normal, sph) : sph));
		res.out = unit;                                                    /* diffuse, emission, isotropic */
		if (mirror) {
			v3 reflected = vreflect(kind == CRH_BSDF_METAL ? unit : rec.dir, rec.normal);
			if (fuzzy) {
				const v3 fuzz = vscale(sph, rough);
				reflected = vadd(reflected, fuzz);
				refracted = vadd(refracted, fuzz);                         /* (glass; dead otherwise) */
			}
			res.out = reflected;
			if (kind == CRH_BSDF_GLASS) res.out = (getDimension(rng) < reflectionProbability) ? reflected : refracted;
		}
		if (kind == CRH_BSDF_TRANSPARENT) res.out = rec.dir;
		if (kind == CRH_BSDF_PLASTIC) col = rgba{1.0f, 1.0f, 1.0f, 1.0f};
		if (kind == CRH_BSDF_EMISSION) col = ccoef(vb, col);
		/* (background as a surface bsdf never happens; unknown kinds are rejected at upload) */
		res.r = col.r; res.g = col.g; res.b = col.b;
		/* unwind pending add frames */
		for (;;) {
			if (asp == 0) return res;
			const uint32_t top = addStack[asp - 1];
			if (!(top & 0x80000000u)) {
				resStack[rsp++] = res;
				addStack[asp - 1] = top | 0x80000000u;
				cur = loadBsdf(S, src, top).b;
				break;
			}
			const BsdfSample A = resStack[--rsp];
			res.out = vadd(A.out, res.out);
			res.r = A.r + res.r; res.g = A.g + res.g; res.b = A.b + res.b;
			--asp;
		}
	}
}

/* background.c:39-66 (on a miss): rec.dir = incident direction, everything else zero */
template <class Cnt, class Src = GlobalTables>
CRH_DEV rgba sampleBackground(const DScene &S, ShadeRec &rec, Cnt &cnt, const Src &src = Src()) {
	const DBsdf n = loadBsdf(S, src, S.background);
	v3 ud = vnorm(rec.dir);
	float phi = (em::atan2f_(ud.z, ud.x) / 4.0f) + evalValue(S, n.c, rec, cnt, src);
	float theta = em::acosf_((-ud.y / 1.0f));
	float u = theta / CRH_PI;
	float v = (phi / (CRH_PI / 2.0f));
	u = wrap01(u);
	v = wrap01(v);
	rec.uv = v2{v, u};
	float strength = evalValue(S, n.b, rec, cnt, src);
	return ccoef(strength, evalColor(S, n.a, rec, cnt, src));
}

/* ---- intersection ------------------------------------------------------------------------------ */
/* oct bits 0..2: signbit of d (bvh.c:368-372); bits 4..6: that slab is tested exactly (|1/d| > 1e30, incl. d == 0);
 * bit 7: some component of the ray is not finite (reference NaN semantics are then followed literally). */
struct RayK { v3 o, d, inv, ss; uint32_t oct; };
#define CRH_RAY_SLOW 0xF0u
#define CRH_RAY_PHASE_SHIFT 16
#define CRH_RAY_LITERAL 0x100u   /* degenerate slabs follow the reference's NaN arithmetic literally instead of being tested exactly (set by the caller of walkBegin) */
CRH_DEV bool finitef(float x) { return fabsf(x) <= FLT_MAX; }
CRH_DEV RayK makeRayK(v3 o, v3 d) {                     /* bvh.c:368-376 */
	RayK k;
	k.o = o; k.d = d;
	k.inv = v3{1.0f / d.x, 1.0f / d.y, 1.0f / d.z};
	k.ss = vscale(vmul(o, k.inv), -1.0f);
	const bool fin = finitef(o.x) && finitef(o.y) && finitef(o.z) && finitef(d.x) && finitef(d.y) && finitef(d.z);
	k.oct = (signbit(d.x) ? 1u : 0u) | (signbit(d.y) ? 2u : 0u) | (signbit(d.z) ? 4u : 0u)
		  | (fabsf(k.inv.x) > 1e30f ? 16u : 0u) | (fabsf(k.inv.y) > 1e30f ? 32u : 0u) | (fabsf(k.inv.z) > 1e30f ? 64u : 0u)
		  | (fin ? 0u : 128u);
	/* bits 16..18: the phase a lane with this ray is in when a child pair is next (PH_NODE, or PH_NODE_SLOW for a degenerate ray) — walkAdvance sets it with one
	 * shift instead of mask + compare + select after every node and triangle step */
	k.oct |= ((k.oct & CRH_RAY_SLOW) ? 7u /* PH_NODE_SLOW */ : 1u /* PH_NODE */) << CRH_RAY_PHASE_SHIFT;
	return k;
}
/* bvh.c:326-352; n0 = {minx,maxx,miny,maxy}, n1 = {minz,maxz,first,countLeaf}.
 *
 * Fast path (every ordinary ray): with finite inputs and |invDir| <= 1e30 no product overflows, no NaN can arise,
 * and the reference's compare-select chain IS max / min (the octant-selected bound is the smaller / larger of the
 * two slab parameters because invDir carries the sign the octant was taken from), so v_min / v_max / v_max3 /
 * v_min3 give the same boolean and the same tEntry (up to the sign of zero, which no comparison sees).
 *
 * Slow path. With CRH_RAY_LITERAL in k.oct (crh_trace_rays by default since round 3) every ray follows the reference's select chain literally — NaN
 * slabs and all: the same node visits, the same record, whatever the ray. Without it (the render kernels), two deliberate differences
 * ("degenerate rays"; result-preserving except where the reference's extra visits find a
 * hit its own rounding error allows — an origin one ulp beside an axis-aligned face, the ray parallel to it: 5 of 190 750 adversarial
 * zero / denormal-component rays in tools/emu_fuzz_rays.py, none of the rendered fixtures and configurations):
 *  - a direction component that is zero (or so small that 1/d > 1e30): the reference computes
 *    fma(bound, inf, -start*inf) = inf - inf = NaN whenever bound and start have the same sign, and its
 *    NaN-ordered selects then drop that slab AND the x slab, so the walk visits every node and triangle of the
 *    scene (measured: 451 585 node + 524 290 triangle tests for one camera ray of input/hdr.json, about one ray
 *    per frame on the image row that crosses the horizon; ~10 ms on a CPU core, a third of a second for one GPU
 *    lane with the chip waiting). Box tests only cull — the closest hit is decided by the triangle / sphere tests,
 *    which never use invDir — so such a slab is tested exactly (inside iff min <= start <= max): same hit (up to
 *    exact-tie order), after a normal number of node visits. tests: test_zero_component_rays.
 *  - non-finite rays follow the reference's select chain literally.
 */
/* v_min / v_max / v_max3 / v_min3 on values that are already canonical (results of fma on finite inputs): written as
 * instructions so that the compiler does not put a quieting v_max x,x in front of every operand */
#if defined(__HIPCC__)
CRH_DEV float hwmin(float a, float b) { float r; asm("v_min_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
CRH_DEV float hwmax(float a, float b) { float r; asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
CRH_DEV float hwmax3(float a, float b, float c) { float r; asm("v_max3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c)); return r; }
CRH_DEV float hwmin3(float a, float b, float c) { float r; asm("v_min3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c)); return r; }
#else
CRH_DEV float hwmin(float a, float b) { return fminf(a, b); }
CRH_DEV float hwmax(float a, float b) { return fmaxf(a, b); }
CRH_DEV float hwmax3(float a, float b, float c) { return fmaxf(fmaxf(a, b), c); }
CRH_DEV float hwmin3(float a, float b, float c) { return fminf(fminf(a, b), c); }
#endif
/* FAST = true: the caller knows the ray is regular (no zero / non-finite direction component: !(oct & CRH_RAY_SLOW)) */
template <bool FAST = false>
CRH_DEV bool intersectNode(const f4 n0, const f4 n1, const RayK &k, float maxDist, float &tEntry) {
	const float xa = __builtin_fmaf(n0.x, k.inv.x, k.ss.x), xb = __builtin_fmaf(n0.y, k.inv.x, k.ss.x);
	const float ya = __builtin_fmaf(n0.z, k.inv.y, k.ss.y), yb = __builtin_fmaf(n0.w, k.inv.y, k.ss.y);
	const float za = __builtin_fmaf(n1.x, k.inv.z, k.ss.z), zb = __builtin_fmaf(n1.y, k.inv.z, k.ss.z);
	if (FAST || !(k.oct & CRH_RAY_SLOW)) {
		const float tMin = hwmax3(hwmax(hwmin(xa, xb), hwmin(ya, yb)), hwmin(za, zb), 0.0f);
		const float tMax = hwmin3(hwmin(hwmax(xa, xb), hwmax(ya, yb)), hwmax(za, zb), maxDist);
		tEntry = tMin;
		return tMin <= tMax;
	}
	const bool ox = k.oct & 1u, oy = k.oct & 2u, oz = k.oct & 4u;
	float tMinX = ox ? xb : xa, tMaxX = ox ? xa : xb;
	float tMinY = oy ? yb : ya, tMaxY = oy ? ya : yb;
	float tMinZ = oz ? zb : za, tMaxZ = oz ? za : zb;
	if (!(k.oct & (128u | CRH_RAY_LITERAL))) {
		const float inf = __builtin_inff();
		if (k.oct & 16u) { const bool in = (n0.x <= k.o.x) && (k.o.x <= n0.y); tMinX = in ? -inf : inf; tMaxX = in ? inf : -inf; }
		if (k.oct & 32u) { const bool in = (n0.z <= k.o.y) && (k.o.y <= n0.w); tMinY = in ? -inf : inf; tMaxY = in ? inf : -inf; }
		if (k.oct & 64u) { const bool in = (n1.x <= k.o.z) && (k.o.z <= n1.y); tMinZ = in ? -inf : inf; tMaxZ = in ? inf : -inf; }
	}
	float tMin = tMinX > tMinY ? tMinX : tMinY;
	float tMax = tMaxX < tMaxY ? tMaxX : tMaxY;
	tMin = tMin > tMinZ ? tMin : tMinZ;
	tMax = tMax < tMaxZ ? tMax : tMaxZ;
	tMin = tMin > 0.0f ? tMin : 0.0f;
	tMax = tMax < maxDist ? tMax : maxDist;
	tEntry = tMin;
	return tMin <= tMax;
}
#define CRH_DNODE_FIRST(n1) asU32((n1).z)
/* (the leaf flag is the SIGN bit of the device record's last word — one signed compare in the node step instead of mask + compare; scene_compile.cpp: relayoutBvh) */
#define CRH_DNODE_LEAF_BIT 0x80000000u
#define CRH_DNODE_COUNT(n1) (asU32((n1).w) & 0x3FFFFFFFu)
#define CRH_DNODE_ISLEAF(n1) ((int32_t)asU32((n1).w) < 0)

/* The WIDE walk's child references (round 5, CRH_OPT_WALK = CRH_WALK_WIDE4; scene_compile.cpp: buildWide): bit 31 clear = a wide node, its offset from S.nodes in
 * 16-byte units; bit 31 set = a leaf, count << 25 | first absolute prim slot; CRH_NONE = an unused slot. The same words are the wide walk's stack entries. */
#define CRH_WREF_LEAF 0x80000000u
#define CRH_WREF_COUNT_SHIFT 25u
#define CRH_WREF_COUNT_MAX 63u
#define CRH_WREF_FIRST_MASK 0x01FFFFFFu

struct TravHit {
	float t;           /* isect->distance */
	float u, v;        /* barycentrics of the closest triangle (isect->uv before getTexMapMesh) */
	int32_t slot;      /* BLAS prim slot of the closest triangle, -1 for spheres / miss */
	int32_t inst;      /* isect->instIndex */
};

/*
 * getClosestIsect (pathtrace.c:26-30) -> traverseTopLevelBvh (bvh.c:488-496) as a per-lane STATE MACHINE.
 *
 * A lane's walk is cut into steps of three kinds — NODE (test one child pair, bvh.c:391-436), TRI (one leaf
 * triangle, poly.c:17-53) and CTRL (enter the next instance of a TLAS leaf / leave a finished BLAS / finish,
 * instance.c:45-60,169-185, bvh.c:468-486) — and `phase` says which kind the lane needs next. The order of every
 * box / triangle / instance test is the reference's: both children tested with the old maxDist, leaf children
 * intersected left then right before descending, nearer inner child first. Because each step is a short
 * straight-line body, the wave-level scheduler (cray_hip.hip: k_pathtrace) can run, at every iteration, the ONE
 * kind of step that most lanes are waiting for — lanes never sit out a whole divergent loop of their neighbours.
 * Hit attributes are derived after the walk (finishHit).
 *
 * Stack: LDS-resident (device) / local array (host emulation); entries are device node indices. While a lane is inside
 * a BLAS the interrupted top-level walk (resume pair, pending instance ranges) and the world-space ray with its slab
 * constants are parked in fixed LDS slots (PK_*).
 */
#define CRH_TLAS_SAVE 0    /* stack entries a BLAS visit adds on top of the node entries: none since round 4 (the interrupted top-level walk — resume pair and pending
                            * instance ranges — is parked in fixed slots, PK_NODE ... PK_PBE, beside the world ray; until then it went through five stack pushes and pops, each
                            * with its own LDS-or-overflow branch: ~15 vector and ~50 scalar instructions per BLAS visit in the node step that leaves and in the control step that enters) */
/* fixed per-lane park slots (LDS on the device): the world-space ray and its slab constants while a lane is inside a BLAS */
/* (parking the ray only and recomputing the slab constants on the way out was measured: -2.4 % on hdr.json, -6 % on statues.json; profiles/r03_exp_variants.patch) */
/* (the slab offsets ss = -(o * inv) are not parked: three multiplications and three negations of parked values give the same bits back, and the
 * three LDS words per lane they occupied until round 3 — 3 KB per workgroup — now hold the instance records, cray_hip.hip: CRH_INST_LDS0_MAX) */
enum { PK_OX, PK_OY, PK_OZ, PK_DX, PK_DY, PK_DZ, PK_IX, PK_IY, PK_IZ, PK_OCT, PK_NODE, PK_PA, PK_PAE, PK_PB, PK_PBE, CRH_PARK_SLOTS };

static_assert(true, "");   /* (PH_NODE = 1 and PH_NODE_SLOW = 7 are what makeRayK stores in the ray flag word: CRH_RAY_PHASE_SHIFT) */
enum { PH_SETUP = 0, PH_NODE = 1, PH_TRI = 2, PH_CTRL = 3, PH_SHADE = 4, PH_DONE = 5, PH_IDLE = 6, PH_NODE_SLOW = 7 };   /* PH_NODE_SLOW: a node step for a degenerate ray (rare; served with the control steps) */   /* PH_IDLE: a worker lane without a ray (queue driver) */

struct Walk {
	uint32_t phase;
	RayK k;                                  /* current-level ray: world ray in the TLAS, object-space ray inside a BLAS */
	uint32_t node;                           /* device index of the child PAIR tested next (children are adjacent: bvh.c:393-394) */
	uint32_t pA, pAe, pB, pBe;               /* pending leaf prim ranges [p, pe) at the current level: left leaf, right leaf */
	uint32_t sp, spBase;
	uint32_t inBlas, instFound;
	int32_t curInst;
	TravHit hit;
};

/* What a walk needs from its PATH (volumes only, instance.c:62-92, 187-216): one sampler draw inside the traversal, and a few words
 * of per-path memory while two BLAS walks bracket the medium. The kernels pass the path's record in the path table, the host
 * emulation the lane's own path; walks of caller rays (crh_trace_rays) have no path — scenes with volumes are refused there. */
struct NullPort {
	CRH_MEM float draw() { return 0.5f; }
	CRH_MEM void save(int, uint32_t) {}
	CRH_MEM uint32_t load(int) { return 0u; }
};
enum { VP_T, VP_U, VP_V, VP_SLOT, VP_T1, VP_WORDS };       /* saved closest hit (t, u, v, slot) + entry distance of the medium */
enum { BLAS_NONE = 0, BLAS_SOLID = 1, BLAS_VOL_ENTRY = 2, BLAS_VOL_EXIT = 3 };   /* Walk::inBlas */

/* sphere.c:20-50 without the hit-point / normal part: true and the distance if the sphere is hit in (0.00001, bound] */
template <class Cnt>
CRH_DEV bool sphereTest(const v3 o, const v3 d, float radius, float bound, float &t, Cnt &cnt) {
	CRH_COUNT(cnt, sphere_tests, 1);
	const float A = vdot(d, d);
	const float B = 2.0f * vdot(d, o);
	const float C = vdot(o, o) - (radius * radius);
	const float disc = B * B - 4.0f * A * C;
	if (disc < 0.0f) return false;
	const float sq = sqrtf(disc);
	float t0 = (-B + sq) / 2.0f;
	const float t1 = (-B - sq) / 2.0f;
	if (t0 > t1 && t1 > 0.0f) t0 = t1;
	if (t0 < 0.00001f || t0 > bound) return false;
	t = t0;
	return true;
}

/* one instance under a one-leaf top-level BVH (the triangle soups of BASELINE.json configs[4]): when its BLAS has been walked, the walk is over */
CRH_DEV bool sceneIsOneInstance(const DScene &S) { return S.instance_count == 1u && S.tlas_node_count == 1u; }

/* after a step that left no pending prims: continue with the next pair, pop one, leave the BLAS, or hand over (CTRL / SHADE) */
/* A volume instance's BLAS walk ended (instance.c:196-214). Entry walk found the medium's near side -> start the exit walk from just
 * behind it (true: the lane keeps walking); exit walk found the far side -> sample the free flight; anything else -> no hit. */
template <class Stack, class Cnt, class Port>
CRH_DEV bool volumeAdvance(const DScene &S, Walk &w, Stack &stk, Cnt &cnt, Port &port) {
	const bool found = w.instFound != 0u;
	const float tWalk = w.hit.t;
	if (found) {      /* both walks ran on a copy of the record: the closest hit so far comes back */
		w.hit.t = asF32(port.load(VP_T)); w.hit.u = asF32(port.load(VP_U)); w.hit.v = asF32(port.load(VP_V)); w.hit.slot = (int32_t)port.load(VP_SLOT);
	}
	w.instFound = 0;
	if (w.inBlas == BLAS_VOL_ENTRY && found) {
		port.save(VP_T1, asU32(tWalk));
		const InstLine inst = instLine(S, stk, w.curInst, 0);
		const v3 d = w.k.d;
		{ const uint32_t lit = w.k.oct & CRH_RAY_LITERAL; w.k = makeRayK(alongRay(w.k.o, d, tWalk + 0.0001f), d); w.k.oct |= lit; }
		w.inBlas = BLAS_VOL_EXIT;
		w.pA = w.pAe = w.pB = w.pBe = 0;
		w.node = CRH_NONE;
		if (CRH_DINST_KIND(instKind(inst)) == CRH_DINST_MESH_LEAF) {                  /* bvh.c:382-387 */
			const f4 n0 = S.nodes[2u * instRoot(inst)], n1 = S.nodes[2u * instRoot(inst) + 1u];
			float tE;
			CRH_COUNT(cnt, node_tests, 1);
			if (intersectNode(n0, n1, w.k, w.hit.t, tE)) { w.pA = CRH_DNODE_FIRST(n1); w.pAe = w.pA + CRH_DNODE_COUNT(n1); }
		} else {
			w.node = cnt_traits<Cnt>::wide ? asU32(instRadius(inst)) : instRoot(inst);
		}
		if (w.pA != w.pAe) { w.phase = PH_TRI; return true; }
		if (w.node != CRH_NONE) { w.phase = w.k.oct >> CRH_RAY_PHASE_SHIFT; return true; }
		return false;                                                            /* single-leaf BLAS missed by the exit ray */
	}
	if (w.inBlas == BLAS_VOL_EXIT && found) {
		float t1 = asF32(port.load(VP_T1));
		if (t1 < 0.0f) t1 = 0.0f;
		const float distanceInsideVolume = tWalk;
		const float hitDistance = -(1.0f / instDensity(instLine(S, stk, w.curInst, 1))) * em::logf_(port.draw());
		if (hitDistance < distanceInsideVolume) {
			w.hit.t = t1 + hitDistance; w.hit.u = 0.0f; w.hit.v = 0.0f; w.hit.slot = -2; w.hit.inst = w.curInst;
			CRH_COUNT(cnt, inst_hits, 1);
		}
	}
	return false;
}

template <class Stack, class Cnt, class Port>
CRH_DEV void walkAdvance(const DScene &S, Walk &w, Stack &stk, Cnt &cnt, Port &port) {
	if (w.pA != w.pAe) { w.phase = w.inBlas ? PH_TRI : PH_CTRL; return; }
	if constexpr (cnt_traits<Cnt>::wide) {      /* a wide walk's stack holds leaves too (stepWideLoaded): a popped leaf is the pending range */
		if (w.node == CRH_NONE && w.sp > w.spBase) {
			const uint32_t e = stk.pop(--w.sp);
			if (e & CRH_WREF_LEAF) { w.pA = e & CRH_WREF_FIRST_MASK; w.pAe = w.pA + ((e >> CRH_WREF_COUNT_SHIFT) & CRH_WREF_COUNT_MAX); w.phase = w.inBlas ? PH_TRI : PH_CTRL; return; }
			w.node = e;
		}
	} else {
		if (w.node == CRH_NONE && w.sp > w.spBase) w.node = stk.pop(--w.sp);
	}
	if (w.node != CRH_NONE) { w.phase = w.k.oct >> CRH_RAY_PHASE_SHIFT; return; }
	if (!w.inBlas) { w.phase = PH_SHADE; return; }                /* TLAS exhausted -> the walk is over */
	if constexpr (cnt_traits<Cnt>::programs) {       /* volumes exist only in the rare-features instantiations (see CountersT) */
		if (__builtin_expect(w.inBlas >= BLAS_VOL_ENTRY, 0)) { if (volumeAdvance(S, w, stk, cnt, port)) return; }
	}
	/* BLAS exhausted -> back to the TLAS walk (instance.c:176-183): the world ray and the TLAS cursor come back from LDS.
	 * Done here, at the end of whichever step emptied the BLAS, rather than as a step of its own: a leave is a dozen LDS
	 * reads, far cheaper than a scheduling round at the occupancy such a step would get. */
	if (w.instFound) { w.hit.inst = w.curInst; CRH_COUNT(cnt, inst_hits, 1); }
	w.inBlas = 0; w.instFound = 0;
	/* (round 5) a scene of ONE instance under a one-leaf top level — BASELINE's triangle soups — has nothing to return to: no resume pair, no further instance, an empty stack.
	 * Nothing was parked (stepCtrl), and the walk is over; the restored state would have led here too (no pending range, no node, nothing to pop -> PH_SHADE). Until now a node or
	 * triangle step ran the fifteen LDS reads below for the ONE lane that left its BLAS in every other iteration of a soup's wave. The test is on kernel arguments: a scalar branch */
	if (sceneIsOneInstance(S)) { w.spBase = 0; w.phase = PH_SHADE; return; }
	w.k.o = v3{asF32(stk.unpark(PK_OX)), asF32(stk.unpark(PK_OY)), asF32(stk.unpark(PK_OZ))};
	w.k.d = v3{asF32(stk.unpark(PK_DX)), asF32(stk.unpark(PK_DY)), asF32(stk.unpark(PK_DZ))};
	w.k.inv = v3{asF32(stk.unpark(PK_IX)), asF32(stk.unpark(PK_IY)), asF32(stk.unpark(PK_IZ))};
	w.k.ss = vscale(vmul(w.k.o, w.k.inv), -1.0f);                    /* makeRayK's own expression on the same values */
	w.k.oct = stk.unpark(PK_OCT);
	w.node = stk.unpark(PK_NODE); w.pA = stk.unpark(PK_PA); w.pAe = stk.unpark(PK_PAE); w.pB = stk.unpark(PK_PB); w.pBe = stk.unpark(PK_PBE);
	w.spBase = 0;
	if (w.pA != w.pAe) { w.phase = PH_CTRL; return; }
	if constexpr (cnt_traits<Cnt>::wide) {
		if (w.node == CRH_NONE && w.sp > 0u) {
			const uint32_t e = stk.pop(--w.sp);
			if (e & CRH_WREF_LEAF) { w.pA = e & CRH_WREF_FIRST_MASK; w.pAe = w.pA + ((e >> CRH_WREF_COUNT_SHIFT) & CRH_WREF_COUNT_MAX); w.phase = PH_CTRL; return; }
			w.node = e;
		}
	} else {
		if (w.node == CRH_NONE && w.sp > 0u) w.node = stk.pop(--w.sp);
	}
	w.phase = (w.node != CRH_NONE) ? (w.k.oct >> CRH_RAY_PHASE_SHIFT) : PH_SHADE;
}

template <class Stack, class Cnt, class Port>
CRH_DEV void walkBegin(const DScene &S, Walk &w, Stack &stk, const v3 o, const v3 d, Cnt &cnt, Port &port, uint32_t rayFlags = 0u) {
	w.hit.t = FLT_MAX; w.hit.u = 0.0f; w.hit.v = 0.0f; w.hit.slot = -1; w.hit.inst = -1;
	w.node = CRH_NONE; w.pA = w.pAe = w.pB = w.pBe = 0; w.sp = 0; w.spBase = 0;
	w.inBlas = 0; w.instFound = 0; w.curInst = -1;
	CRH_COUNT1(cnt, rays, 1);
	w.k = makeRayK(o, d);
	w.k.oct |= rayFlags;                                                    /* CRH_RAY_LITERAL travels with the ray into every BLAS */
	if (S.tlas_node_count < 1u) { w.phase = PH_SHADE; return; }              /* bvh.c:362-365 */
	if (S.tlas_node_count == 1u) {                                          /* bvh.c:382-387 */
		const f4 n0 = S.nodes[2u * S.tlas_root], n1 = S.nodes[2u * S.tlas_root + 1u];
		float tE;
		CRH_COUNT(cnt, node_tests, 1);
		if (intersectNode(n0, n1, w.k, w.hit.t, tE)) { w.pA = CRH_DNODE_FIRST(n1); w.pAe = w.pA + CRH_DNODE_COUNT(n1); }
	} else {
		w.node = S.tlas_root;
	}
	walkAdvance(S, w, stk, cnt, port);
}

/* NODE: bvh.c:391-436 */
/* FAST = true serves PH_NODE lanes (regular rays: no degenerate-slab code in the step at all), FAST = false PH_NODE_SLOW lanes */
/* (the child pair arrives as four 16-byte quarters: stepNode loads them itself; k_pathtrace fetches the pairs of a whole wave quad-cooperatively and calls this) */
template <bool FAST = true, class Stack, class Cnt, class Port>
CRH_DEV void stepNodeLoaded(const DScene &S, Walk &w, Stack &stk, Cnt &cnt, Port &port, const f4 l0, const f4 l1, const f4 r0, const f4 r1) {
	float tL, tR;
	CRH_COUNT(cnt, node_tests, 2);
	const bool hitL = intersectNode<FAST>(l0, l1, w.k, w.hit.t, tL);
	const bool hitR = intersectNode<FAST>(r0, r1, w.k, w.hit.t, tR);
	const bool leafL = CRH_DNODE_ISLEAF(l1), leafR = CRH_DNODE_ISLEAF(r1);
	const uint32_t fl = CRH_DNODE_FIRST(l1), fr = CRH_DNODE_FIRST(r1);
	/* pending leaf ranges (none are pending when a node step runs): left leaf first, then the right one */
	const bool lL = hitL && leafL, lR = hitR && leafR;
	const uint32_t nL = lL ? CRH_DNODE_COUNT(l1) : 0u, nR = lR ? CRH_DNODE_COUNT(r1) : 0u;
	w.pA = lL ? fl : fr;
	w.pAe = w.pA + (lL ? nL : nR);
	w.pB = fr;
	w.pBe = fr + (lL ? nR : 0u);
	/* descent: both inner -> nearer first, the other one on the stack */
	const bool inL = hitL && !leafL, inR = hitR && !leafR;
	const bool swap = tL > tR;
	if (inL && inR) stk.push(w.sp++, swap ? fl : fr);
	w.node = (inL && inR) ? (swap ? fr : fl) : (inL ? fl : (inR ? fr : CRH_NONE));
	walkAdvance(S, w, stk, cnt, port);
}
template <bool FAST = true, class Stack, class Cnt, class Port>
CRH_DEV void stepNode(const DScene &S, Walk &w, Stack &stk, Cnt &cnt, Port &port) {
	/* the pair's byte offset as ONE 32-bit value added to the (wave-uniform) array base: global_load with a scalar base, a 32-bit vector offset and immediate
	 * offsets for the four quarters instead of two 64-bit address computations per node step (round 3: profiles/r03q_ab_offset32.log). The scene compiler
	 * refuses node / triangle / shading-record / texel arrays of 4 GB and more (134 M nodes, 89 M triangles per scene) */
	const char *pair = (const char *)S.nodes + (uint32_t)(w.node << 5);
	const f4 l0 = *(const f4 *)pair, l1 = *(const f4 *)(pair + 16), r0 = *(const f4 *)(pair + 32), r1 = *(const f4 *)(pair + 48);
	stepNodeLoaded<FAST>(S, w, stk, cnt, port, l0, l1, r0, r1);
}

/* WIDE NODE (round 5, an option: DESIGN.md section 7): one step tests the FOUR boxes of a collapsed node — the reference's own boxes, bit for bit, of the binary nodes the
 * collapse kept — against the same maxDist with the same slab arithmetic, orders the children that are hit by entry distance, continues with the nearest and leaves the
 * others on the stack, farthest first. A leaf child is an entry like any other: it becomes the pending range when it is the nearest or when it is popped.
 * What this changes against bvh.c:391-436: the ORDER in which leaves are reached (the reference tests a pair's leaves before it descends, and the nearer of two inner
 * children first; here every child waits its turn by entry distance) and the boxes that are never tested (the binary nodes the collapse skipped: a child box is inside its
 * parent's, and the slab test is monotone in the bounds, so a skipped box would have passed whenever one of its children does). The closest hit is the same whenever it
 * is unique and the reference's own culling is consistent; of two triangles at exactly the same distance the FIRST one tested wins (poly.c:33: t < distance), so an
 * exact tie may resolve differently — counted, not assumed: tools/wide_walk_study.py. */
template <bool FAST = true, class Stack, class Cnt, class Port>
CRH_DEV void stepWideLoaded(const DScene &S, Walk &w, Stack &stk, Cnt &cnt, Port &port, const f4 a0, const f4 a1, const f4 b0, const f4 b1, const f4 c0, const f4 c1, const f4 d0, const f4 d1) {
	float t0, t1, t2, t3;
	CRH_COUNT(cnt, node_tests, 4);
	const bool h0 = intersectNode<FAST>(a0, a1, w.k, w.hit.t, t0);
	const bool h1 = intersectNode<FAST>(b0, b1, w.k, w.hit.t, t1);
	const bool h2 = intersectNode<FAST>(c0, c1, w.k, w.hit.t, t2);
	const bool h3 = intersectNode<FAST>(d0, d1, w.k, w.hit.t, t3);
	const float inf = __builtin_inff();
	float k0 = h0 ? t0 : inf, k1 = h1 ? t1 : inf, k2 = h2 ? t2 : inf, k3 = h3 ? t3 : inf;
	uint32_t r0 = h0 ? asU32(a1.z) : CRH_NONE, r1 = h1 ? asU32(b1.z) : CRH_NONE, r2 = h2 ? asU32(c1.z) : CRH_NONE, r3 = h3 ? asU32(d1.z) : CRH_NONE;
	/* five compare-exchanges: (0,1) (2,3) (0,2) (1,3) (1,2); ascending, misses (inf, CRH_NONE) last; a strict compare keeps slot order among equal distances */
#define CRH_CSWAP(ka, ra, kb, rb) do { const bool sw_ = kb < ka; const float kt_ = sw_ ? kb : ka; kb = sw_ ? ka : kb; ka = kt_; const uint32_t rt_ = sw_ ? rb : ra; rb = sw_ ? ra : rb; ra = rt_; } while (0)
	CRH_CSWAP(k0, r0, k1, r1); CRH_CSWAP(k2, r2, k3, r3); CRH_CSWAP(k0, r0, k2, r2); CRH_CSWAP(k1, r1, k3, r3); CRH_CSWAP(k1, r1, k2, r2);
#undef CRH_CSWAP
	if (r3 != CRH_NONE) stk.push(w.sp++, r3);
	if (r2 != CRH_NONE) stk.push(w.sp++, r2);
	if (r1 != CRH_NONE) stk.push(w.sp++, r1);
	w.node = CRH_NONE;
	if (r0 != CRH_NONE) {
		if (r0 & CRH_WREF_LEAF) { w.pA = r0 & CRH_WREF_FIRST_MASK; w.pAe = w.pA + ((r0 >> CRH_WREF_COUNT_SHIFT) & CRH_WREF_COUNT_MAX); }
		else w.node = r0;
	}
	walkAdvance(S, w, stk, cnt, port);
}
template <bool FAST = true, class Stack, class Cnt, class Port>
CRH_DEV void stepWide(const DScene &S, Walk &w, Stack &stk, Cnt &cnt, Port &port) {
	const char *rec = (const char *)S.nodes + (uint32_t)(w.node << 4);
	const f4 a0 = *(const f4 *)rec, a1 = *(const f4 *)(rec + 16), b0 = *(const f4 *)(rec + 32), b1 = *(const f4 *)(rec + 48);
	const f4 c0 = *(const f4 *)(rec + 64), c1 = *(const f4 *)(rec + 80), d0 = *(const f4 *)(rec + 96), d1 = *(const f4 *)(rec + 112);
	stepWideLoaded<FAST>(S, w, stk, cnt, port, a0, a1, b0, b1, c0, c1, d0, d1);
}
/* the node step of a walk whose form the counter type names */
template <bool FAST, class Stack, class Cnt, class Port>
CRH_DEV void stepNodeAny(const DScene &S, Walk &w, Stack &stk, Cnt &cnt, Port &port) {
	if constexpr (cnt_traits<Cnt>::wide) stepWide<FAST>(S, w, stk, cnt, port); else stepNode<FAST>(S, w, stk, cnt, port);
}

/* TRI: poly.c:17-53 on the prepared record (one triangle per step) */
template <class Cnt>
CRH_DEV void testTriangle(const f4 q0, const f4 q1, const f4 q2, uint32_t slot, Walk &w, Cnt &cnt) {
	const v3 v0 = v3{q0.x, q0.y, q0.z}, e1 = v3{q0.w, q1.x, q1.y}, e2 = v3{q1.z, q1.w, q2.x}, n = v3{q2.y, q2.z, q2.w};
	CRH_COUNT(cnt, tri_tests, 1);
	const v3 c = vsub(v0, w.k.o);
	const v3 r = vcross(w.k.d, c);
	const float invDet = 1.0f / vdot(n, w.k.d);
	const float u = vdot(r, e2) * invDet;
	const float v = vdot(r, e1) * invDet;
	if (u >= 0.0f && v >= 0.0f && u + v <= 1.0f) {
		const float t = vdot(n, c) * invDet;
		if (t >= 0.0f && t < w.hit.t) { w.hit.t = t; w.hit.u = u; w.hit.v = v; w.hit.slot = (int32_t)slot; w.instFound = 1; }
	}
}
/* One triangle step = the next TWO triangles of the pending leaf range when it holds two (in order, the second sees the
 * first's hit distance: poly.c:17-36 via bvh.c:449-458); both records are requested before either is used. */
/* (the records arrive loaded: stepTri fetches them itself; the rolling kernel's node run requests them together with the node pairs of the lanes that keep descending
 * and calls this — fused walk steps, pathtrace_roll.h) */
template <class Stack, class Cnt, class Port>
CRH_DEV void stepTriLoaded(const DScene &S, Walk &w, Stack &stk, Cnt &cnt, Port &port, const f4 a0, const f4 a1, const f4 a2, const f4 b0, const f4 b1, const f4 b2) {
	const uint32_t slot = w.pA;
	const bool two = slot + 1u < w.pAe;
	const uint32_t slot2 = two ? slot + 1u : slot;
	w.pA = slot2 + 1u;
	if (w.pA == w.pAe) { w.pA = w.pB; w.pAe = w.pBe; w.pB = w.pBe = 0; }
	testTriangle(a0, a1, a2, slot, w, cnt);
	if (two) testTriangle(b0, b1, b2, slot2, w, cnt);
	walkAdvance(S, w, stk, cnt, port);
}
template <class Stack, class Cnt, class Port>
CRH_DEV void stepTri(const DScene &S, Walk &w, Stack &stk, Cnt &cnt, Port &port) {
	const uint32_t slot = w.pA;
	const uint32_t slot2 = slot + 1u < w.pAe ? slot + 1u : slot;
	const char *ta = (const char *)S.tris + (uint32_t)(slot * 48u), *tb = (const char *)S.tris + (uint32_t)(slot2 * 48u);
	const f4 a0 = *(const f4 *)ta, a1 = *(const f4 *)(ta + 16), a2 = *(const f4 *)(ta + 32);
	const f4 b0 = *(const f4 *)tb, b1 = *(const f4 *)(tb + 16), b2 = *(const f4 *)(tb + 32);
	stepTriLoaded(S, w, stk, cnt, port, a0, a1, a2, b0, b1, b2);
}

/* CTRL: leave a finished BLAS (bvh.c:468-486 loop body tail) and / or visit the next instance of a TLAS leaf (bvh.c:472-484) */
template <class Stack, class Cnt, class Port>
CRH_DEV void stepCtrl(const DScene &S, Walk &w, Stack &stk, Cnt &cnt, Port &port) {
	/* next instance */
	const uint32_t slot = w.pA++;
	if (w.pA == w.pAe) { w.pA = w.pB; w.pAe = w.pBe; w.pB = w.pBe = 0; }
	const int32_t idx = (int32_t)(slot - S.tlas_prim_base);   /* leaf.first is an absolute prim slot; instances[] is in slot order */
	const InstLine inst = instLine(S, stk, idx, 0);
	CRH_COUNT(cnt, inst_visits, 1);
	/* transformRay(Ainv) + offset: instance.c:46-50 / 170-174 */
	v3 o = xfPoint(w.k.o, inst);
	const v3 d = xfVector(w.k.d, inst);
	o = vadd(o, vscale(d, instRayOffset(inst)));
	const uint32_t kind = CRH_DINST_KIND(instKind(inst));
	const bool volume = cnt_traits<Cnt>::programs && (instKind(inst) & CRH_DINST_VOLUME) != 0u;
	if (kind == CRH_DINST_SPHERE) {
		float t0;
		if (__builtin_expect(volume, 0)) {
			/* instance.c:62-92: where the ray enters the sphere, where — from just behind that point — it leaves it, then a free-flight
			 * distance drawn from the path's sampler; both tests are bounded by the closest hit so far */
			float tExit;
			if (sphereTest(o, d, instRadius(inst), w.hit.t, t0, cnt) && sphereTest(alongRay(o, d, t0 + 0.0001f), d, instRadius(inst), w.hit.t, tExit, cnt)) {
				if (t0 < 0.0f) t0 = 0.0f;
				const float hitDistance = -(1.0f / instDensity(instLine(S, stk, idx, 1))) * em::logf_(port.draw());
				if (hitDistance < tExit) {
					w.hit.t = t0 + hitDistance; w.hit.u = 0.0f; w.hit.v = 0.0f; w.hit.slot = -2; w.hit.inst = idx;
					CRH_COUNT(cnt, inst_hits, 1);
				}
			}
		} else if (sphereTest(o, d, instRadius(inst), w.hit.t, t0, cnt)) {   /* sphere.c:20-50 */
			w.hit.t = t0; w.hit.slot = -1; w.hit.inst = idx;
			CRH_COUNT(cnt, inst_hits, 1);
		}
	} else if (kind == CRH_DINST_MESH_EMPTY) {
		if (!volume) w.hit.inst = -1;                                        /* bvh.c:362-365 via instance.c:175 (a volume walks a COPY of the record) */
	} else if (!(w.k.oct & CRH_RAY_LITERAL) && (o.x != o.x || o.y != o.y || o.z != o.z || d.x != d.x || d.y != d.y || d.z != d.z)) {
		/* A NaN anywhere in the ray makes u (poly.c:30) NaN for every triangle, so no triangle can be accepted;
		 * the reference still walks the whole BLAS (every box test passes on NaN). Same result, no walk. */
	} else {
		RayK ko = makeRayK(o, d);
		ko.oct |= w.k.oct & CRH_RAY_LITERAL;
		bool enter = true;
		uint32_t rootA = 0, rootAe = 0;
		if (kind == CRH_DINST_MESH_LEAF) {                                 /* bvh.c:382-387 */
			const f4 n0 = S.nodes[2u * instRoot(inst)], n1 = S.nodes[2u * instRoot(inst) + 1u];
			float tE;
			CRH_COUNT(cnt, node_tests, 1);
			enter = intersectNode(n0, n1, ko, w.hit.t, tE);
			rootA = CRH_DNODE_FIRST(n1); rootAe = rootA + CRH_DNODE_COUNT(n1);
		}
		if (enter) {
			if (!sceneIsOneInstance(S)) {          /* (a one-instance scene's top-level walk has nothing left behind its only BLAS visit: walkAdvance does not come back for it) */
			stk.park(PK_NODE, w.node); stk.park(PK_PA, w.pA); stk.park(PK_PAE, w.pAe); stk.park(PK_PB, w.pB); stk.park(PK_PBE, w.pBe);
			stk.park(PK_OX, asU32(w.k.o.x)); stk.park(PK_OY, asU32(w.k.o.y)); stk.park(PK_OZ, asU32(w.k.o.z));
			stk.park(PK_DX, asU32(w.k.d.x)); stk.park(PK_DY, asU32(w.k.d.y)); stk.park(PK_DZ, asU32(w.k.d.z));
			stk.park(PK_IX, asU32(w.k.inv.x)); stk.park(PK_IY, asU32(w.k.inv.y)); stk.park(PK_IZ, asU32(w.k.inv.z));
			stk.park(PK_OCT, w.k.oct);
			}
			if (kind == CRH_DINST_MESH_LEAF) { w.node = CRH_NONE; w.pA = rootA; w.pAe = rootAe; }
			else { w.node = cnt_traits<Cnt>::wide ? asU32(instRadius(inst)) /* a mesh's wide root stands where a sphere's radius does */ : instRoot(inst); w.pA = w.pAe = 0; }
			w.pB = w.pBe = 0;
			w.spBase = w.sp; w.inBlas = BLAS_SOLID; w.instFound = 0; w.curInst = idx; w.k = ko;
			if (__builtin_expect(volume, 0)) {                                /* instance.c:188-196: the entry walk runs on a copy of the record */
				w.inBlas = BLAS_VOL_ENTRY;
				port.save(VP_T, asU32(w.hit.t)); port.save(VP_U, asU32(w.hit.u)); port.save(VP_V, asU32(w.hit.v)); port.save(VP_SLOT, (uint32_t)w.hit.slot);
			}
		}
	}
	walkAdvance(S, w, stk, cnt, port);
}

/* The whole walk for one lane (k_trace_rays, host emulation): run steps until the walk hands over to shading. */
template <class Stack, class Cnt>
CRH_DEV void traverse(const DScene &S, Stack &stk, const v3 rayO, const v3 rayD, TravHit &hit, Cnt &cnt, uint32_t rayFlags = 0u) {
	Walk w;
	NullPort port;                    /* caller rays have no path: scenes with volumes are refused before this runs */
	walkBegin(S, w, stk, rayO, rayD, cnt, port, rayFlags);
	while (w.phase != PH_SHADE) {
		if (w.phase == PH_NODE) stepNodeAny<true>(S, w, stk, cnt, port);
		else if (w.phase == PH_NODE_SLOW) stepNodeAny<false>(S, w, stk, cnt, port);
		else if (w.phase == PH_TRI) stepTri(S, w, stk, cnt, port);
		else stepCtrl(S, w, stk, cnt, port);
	}
	hit = w.hit;
}

/* Hit attributes after the walk: exactly what instance.c:45-60 / 169-185 + poly.c:37-48 leave in the record. */
struct HitInfo {
	v3 point, normal;
	v2 uv;
	uint32_t material;
};
/* polygon index (into crh_scene_desc.polys) of a finished hit, -1 for spheres: reported by crh_trace_rays only */
CRH_DEV int32_t hitPoly(const DScene &S, const TravHit &hit) {
	const DInstance *inst = &S.instances[hit.inst];
	return CRH_DINST_KIND(inst->kind) == CRH_DINST_SPHERE ? -1 : (int32_t)inst->poly_base + S.prims[hit.slot];
}
CRH_DEV v3 loadV3(const float *base, int64_t i) { const float *p = base + 3 * i; return v3{p[0], p[1], p[2]}; }
/* LAZY_UV: shading skips a sphere's texture coordinates (atan2f + asinf, instance.c:33-43) when no node of the material's
 * graph reads them (crh_material.pad[0], set by the scene compiler); crh_trace_rays always reports them. */
template <bool LAZY_UV = true, bool VOLUMES = true, class Src = GlobalTables>
CRH_DEV HitInfo finishHit(const DScene &S, const v3 wo, const v3 wd, const TravHit &hit, const Src &src = Src()) {
	HitInfo h;
	const InstLine inst = instLine(S, src, hit.inst, 0), inst1 = instLine(S, src, hit.inst, 1);     /* inst: Ainv + kind / offset; inst1: A + material */
	v3 o = xfPoint(wo, inst);
	const v3 d = xfVector(wd, inst);
	o = vadd(o, vscale(d, instRayOffset(inst)));
	if (VOLUMES && __builtin_expect(hit.slot == -2, 0)) {         /* a scattering event inside a volume: instance.c:81-88 / 205-211 */
		h.uv = v2{-1.0f, -1.0f};
		h.material = instMaterial(inst1);                         /* the sphere's material / mesh->materials[0] */
		h.point = xfPoint(alongRay(wo, wd, hit.t), inst1);        /* the WORLD ray's point, transformed again — as the reference does */
		h.normal = xfVectorT(v3{1.0f, 0.0f, 0.0f}, inst);         /* "will be ignored by material anyway" */
		return h;
	}
	/* (round 3: one normalisation and one xfPoint shared by the sphere and the triangle branch — a batch of hits holds both — was built and measured:
	 * nine more spilled VGPRs, hdr.json -3 %, profiles/r03k_ab_shade.log: b3 / b4. The two branches stay.) */
	const v3 objPoint = alongRay(o, d, hit.t);
	if (CRH_DINST_KIND(instKind(inst)) == CRH_DINST_SPHERE) {
		v3 n = vnorm(objPoint);                                   /* sphere.c:48 */
		h.uv = v2{0.0f, 0.0f};
		if (!LAZY_UV || loadMaterial(S, src, instMaterial(inst1)).pad[0]) {       /* getTexMapSphere: instance.c:33-43 (object-space normal) */
			float phi = em::atan2f_(n.z, n.x);
			float theta = em::asinf_(n.y);
			float v = (theta + CRH_PI / 2.0f) / CRH_PI;
			float u = 1.0f - (phi + CRH_PI) / (CRH_PI * 2.0f);
			u = wrap01(u);
			v = wrap01(v);
			h.uv = v2{u, v};
		}
		h.material = instMaterial(inst1);
		h.point = xfPoint(objPoint, inst1);
		h.normal = xfVectorT(n, inst);                            /* not renormalised: instance.c:56 */
		return h;
	}
	const DShadeTri *st = (const DShadeTri *)((const char *)S.shade + (uint32_t)((uint32_t)hit.slot << 6));      /* base + 32-bit byte offset: see stepNode */
	const uint32_t flags = st->flags;
	const float u = hit.u, v = hit.v;
	const float w = 1.0f - u - v;
	v3 n = v3{st->n0[0], st->n0[1], st->n0[2]};                   /* poly.c:45-47: un-normalised e1 x e2 */
	if (flags & CRH_SHADE_HASNORMALS) {                           /* poly.c:40-44 */
		const v3 upcomp = vscale(v3{st->n1[0], st->n1[1], st->n1[2]}, u);
		const v3 vpcomp = vscale(v3{st->n2[0], st->n2[1], st->n2[2]}, v);
		const v3 wpcomp = vscale(n, w);
		n = vadd(vadd(upcomp, vpcomp), wpcomp);
	}
	/* getTexMapMesh: instance.c:150-167 */
	if (!(flags & CRH_SHADE_HASUV)) {
		h.uv = v2{-1.0f, -1.0f};
	} else {
		h.uv = v2{((st->t1[0] * u) + (st->t2[0] * v)) + (st->t0[0] * w), ((st->t1[1] * u) + (st->t2[1] * v)) + (st->t0[1] * w)};
	}
	h.material = instMaterial(inst1) + (flags & 0x3FFFFFFFu);
	h.point = xfPoint(objPoint, inst1);
	h.normal = vnorm(xfVectorT(n, inst));                         /* instance.c:180-181 */
	return h;
}

/* ---- one path: pathtrace.c:32-60 as SHADE steps; renderer.c:275-301 as SETUP steps -------------------- */
struct BlockJob {
	int x0, y0;          /* block origin in reference coordinates (y from the bottom) */
	int w, h;            /* valid extent inside the block (ragged tile edges) */
	int bw, bh;          /* block shape in pixels */
	int passBegin;       /* first pass of this chunk (completedSamples - 1) */
	int passCount;       /* passes in this chunk */
};

/* running mean: renderer.c:288-291 */
CRH_DEV void foldSample(float &r, float &g, float &b, float sr, float sg, float sb, int completedSamples) {
	const float n1 = (float)(completedSamples - 1);
	const float t = 1.0f / (float)completedSamples;
	r = ((r * n1) + sr) * t;
	g = ((g * n1) + sg) * t;
	b = ((b * n1) + sb) * t;
}

struct Item { uint32_t next, cur; };    /* next item this lane will take / the item of the path in flight */

/* What a path carries from bounce to bounce besides its ray (pathtrace.c:33-35 + the sampler). */
template <class R> struct PathRecT { float wr, wg, wb; float fr, fg, fb; R rng; int depth; };
typedef PathRecT<Rng> PathRec;

/* Item index -> pixel / pass of a block chunk. Items are numbered pixel-major, so consecutive items are passes of
 * the same pixel (coherent primary rays). Returns false for the padding items of a ragged tile edge. */
CRH_DEV bool decodeItem(const BlockJob &J, uint32_t item, int &x, int &y, int &pass) {
	const uint32_t pc = (uint32_t)J.passCount, bw = (uint32_t)J.bw;       /* bw is a power of two; pc usually is */
	const bool pcPow2 = (pc & (pc - 1u)) == 0u;
	const uint32_t pcShift = 31u - (uint32_t)__builtin_clz(pc | 1u), bwShift = 31u - (uint32_t)__builtin_clz(bw | 1u);
	const uint32_t pix = pcPow2 ? (item >> pcShift) : (item / pc);
	const int px = (int)(pix & (bw - 1u)), py = (int)(pix >> bwShift);
	x = J.x0 + px; y = J.y0 + py;
	pass = J.passBegin + (int)(pcPow2 ? (item & (pc - 1u)) : (item - pix * pc));
	return px < J.w && py < J.h;
}

/* A new path: initSampler + getCameraRay (renderer.c:280-284). Needs P.bounces > 0. */
template <class PR, class Cnt>
CRH_DEV void beginPath(const DScene &S, const crh_render_params &P, int x, int y, int pass, v3 &ro, v3 &rd, PR &r, Cnt &cnt) {
	CRH_COUNT1(cnt, paths, 1);
	const uint32_t pixIdx = (uint32_t)(y * P.image_width + x);          /* renderer.c:280 */
	initSampler(r.rng, pass, P.max_passes, pixIdx);                       /* :281 */
	getCameraRay(*S.camera, r.rng, x, y, ro, rd);                          /* :284 */
	r.wr = r.wg = r.wb = 1.0f; r.fr = r.fg = r.fb = 0.0f; r.depth = 0;
}

/*
 * One iteration of the pathTrace() loop body after getClosestIsect (pathtrace.c:39-57), on explicit records: the
 * ray (ro, rd) and its closest hit go in; either the path continues (true: ro / rd are the next ray, r updated) or it
 * is complete (false: r.fr/fg/fb is the sample).
 */
template <class PR, class Cnt, class Src = GlobalTables>
CRH_DEV bool shadeCore(const DScene &S, const crh_render_params &P, v3 &ro, v3 &rd, const TravHit &hit, PR &r, Cnt &cnt, const Src &src = Src()) {
	ShadeRec rec;
	rec.dir = rd;
	if (hit.inst < 0) {                                            /* pathtrace.c:39-42 */
		rec.point = v3{0.0f, 0.0f, 0.0f}; rec.normal = v3{0.0f, 0.0f, 0.0f}; rec.uv = v2{0.0f, 0.0f};
		rec.distance = hit.t; rec.ior = 0.0f;
		const rgba bg = sampleBackground(S, rec, cnt, src);
		r.fr = r.fr + (r.wr * bg.r); r.fg = r.fg + (r.wg * bg.g); r.fb = r.fb + (r.wb * bg.b);
		return false;
	}
	const HitInfo h = finishHit<true, cnt_traits<Cnt>::programs>(S, ro, rd, hit, src);
	const crh_material mat = loadMaterial(S, src, h.material);
	r.fr = r.fr + (r.wr * mat.emission[0]); r.fg = r.fg + (r.wg * mat.emission[1]); r.fb = r.fb + (r.wb * mat.emission[2]);   /* :44 */
	rec.point = h.point; rec.normal = h.normal; rec.uv = h.uv; rec.distance = hit.t; rec.ior = mat.ior;
	const BsdfSample s = sampleBsdf(S, mat.bsdf, rec, r.rng, cnt, src);     /* :46 */
	float probability = 1.0f;
	if (r.depth >= 4) {                                                /* :51-55 */
		probability = rmax(s.r, rmax(s.g, s.b));
		if (getDimension(r.rng) > probability) return false;
	}
	const float ip = 1.0f / probability;                             /* :57 */
	r.wr = (s.r * r.wr) * ip; r.wg = (s.g * r.wg) * ip; r.wb = (s.b * r.wb) * ip;
	++r.depth;
	if (r.depth >= P.bounces) return false;
	ro = h.point; rd = s.out;                                        /* :47 */
	return true;
}

/* A lane that owns its paths from start to end (host emulation; the device driver keeps paths in a per-wave table
 * instead and lets lanes work on whichever path needs a step): */
template <class R> struct LanePathT { Item it; PathRecT<R> r; v3 ro, rd; uint32_t vol[VP_WORDS]; };
template <class R> struct LanePort {
	LanePathT<R> &lp;
	CRH_MEM float draw() { return getDimension(lp.r.rng); }
	CRH_MEM void save(int i, uint32_t v) { lp.vol[i] = v; }
	CRH_MEM uint32_t load(int i) { return lp.vol[i]; }
};
template <class R> CRH_DEV LanePort<R> lanePort(LanePathT<R> &lp) { return LanePort<R>{lp}; }

template <class LP, class Stack, class Cnt>
CRH_DEV void stepSetup(const DScene &S, const crh_render_params &P, const BlockJob &J, uint32_t laneStride, Walk &w, LP &lp,
					   Stack &stk, float *stage, Cnt &cnt) {
	const uint32_t nItems = (uint32_t)(J.bw * J.bh * J.passCount);
	for (;;) {
		if (lp.it.next >= nItems) { w.phase = PH_DONE; return; }
		int x = 0, y = 0, pass = 0;
		const bool valid = decodeItem(J, lp.it.next, x, y, pass);
		lp.it.cur = lp.it.next;
		lp.it.next += laneStride;
		if (!valid) continue;
		if (P.bounces <= 0) {                      /* pathTrace() with maxDepth 0 returns black */
			CRH_COUNT1(cnt, paths, 1);
			float *o = stage + (size_t)lp.it.cur * 3;
			o[0] = 0.0f; o[1] = 0.0f; o[2] = 0.0f;
			continue;
		}
		beginPath(S, P, x, y, pass, lp.ro, lp.rd, lp.r, cnt);
		auto port = lanePort(lp);
		walkBegin(S, w, stk, lp.ro, lp.rd, cnt, port);
		return;
	}
}

template <class LP, class Stack, class Cnt>
CRH_DEV void stepShade(const DScene &S, const crh_render_params &P, Walk &w, LP &lp, Stack &stk, float *stage, Cnt &cnt) {
	if (shadeCore(S, P, lp.ro, lp.rd, w.hit, lp.r, cnt)) {
		auto port = lanePort(lp);
		walkBegin(S, w, stk, lp.ro, lp.rd, cnt, port);
		return;
	}
	float *o = stage + (size_t)lp.it.cur * 3;
	o[0] = lp.r.fr; o[1] = lp.r.fg; o[2] = lp.r.fb;
	w.phase = PH_SETUP;
}

/* One lane, all steps in sequence (host emulation and any non-scheduled use): the block chunk's items of this lane. */
template <class R = Rng, class Stack, class Cnt>
CRH_DEV void renderItems(const DScene &S, const crh_render_params &P, Stack &stk, const BlockJob &J, uint32_t lane, uint32_t laneStride,
						 float *stage, Cnt &cnt) {
	Walk w;
	LanePathT<R> lp;
	lp.it.next = lane; lp.it.cur = 0;
	w.phase = PH_SETUP;
	auto port = lanePort(lp);
	for (;;) {
		switch (w.phase) {
			case PH_SETUP: stepSetup(S, P, J, laneStride, w, lp, stk, stage, cnt); break;
			case PH_NODE: stepNodeAny<true>(S, w, stk, cnt, port); break;
			case PH_NODE_SLOW: stepNodeAny<false>(S, w, stk, cnt, port); break;
			case PH_TRI: stepTri(S, w, stk, cnt, port); break;
			case PH_CTRL: stepCtrl(S, w, stk, cnt, port); break;
			case PH_SHADE: stepShade(S, P, w, lp, stk, stage, cnt); break;
			default: return;
		}
	}
}

/* Fold the staged samples of block pixel `pix` into the float framebuffer (texture.c:24-28 layout). */
CRH_DEV void foldBlockPixel(const crh_render_params &P, const BlockJob &J, uint32_t pix, const float *stage, float *fb) {
	const int px = (int)(pix % (uint32_t)J.bw), py = (int)(pix / (uint32_t)J.bw);
	if (px >= J.w || py >= J.h) return;
	const int x = J.x0 + px, y = J.y0 + py;
	float *out = fb + ((size_t)x + (size_t)(P.image_height - (y + 1)) * (size_t)P.image_width) * 3;
	float r = out[0], g = out[1], b = out[2];
	const float *sp = stage + (size_t)pix * (size_t)J.passCount * 3;
	/* the mean is a serial chain, the loads are not: fetch eight samples at a time, then fold them in pass order (one
	 * load latency per eight passes instead of one per pass — a block's fold is on the critical path of its wave) */
	int k = 0;
	for (; k + 8 <= J.passCount; k += 8) {
		float s[24];
		for (int i = 0; i < 24; ++i) s[i] = sp[3 * k + i];
		for (int j = 0; j < 8; ++j) foldSample(r, g, b, s[3 * j], s[3 * j + 1], s[3 * j + 2], J.passBegin + k + j + 1);
	}
	for (; k < J.passCount; ++k) foldSample(r, g, b, sp[3 * k], sp[3 * k + 1], sp[3 * k + 2], J.passBegin + k + 1);
	out[0] = r; out[1] = g; out[2] = b;
}

}  // namespace crh

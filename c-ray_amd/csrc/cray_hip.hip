/*
 * cray_hip.hip — libcray_hip.so: the C-ABI of include/cray_hip.h and the gfx950 kernels behind it.
 *
 * Kernels (all hand-written for CDNA4, wave = 64):
 *   k_pathtrace_roll<LEVEL, WPS, PROG, SAMP>  (pathtrace_roll.h) the hot kernel since the end of round 3: k_pathtrace's machine with up to four work units
 *                        open per wave, so that the path table stays full across unit boundaries.
 *   (k_pathtrace<LEVEL, WPS, PROG, SAMP> and k_pathtrace_wg, pathtrace_alt.h, only with -DCRH_WITH_ALT_KERNELS: the same machine one work unit at a time / shared by the
 *                        four waves of a workgroup — the forms the hot kernel is proven bit-identical to in the emulation tier.) The machine: persistent grid, every wave a small wavefront machine. A wave
 *                        pulls pixel blocks from a global queue (one atomic per wave, readfirstlane broadcast); the
 *                        block's (pixel, pass) paths live in a per-wave table of 128-byte records (global memory),
 *                        their ids on LDS byte stacks (rays / hits / misses / free); lanes are workers, and each
 *                        iteration the wave ballots its lanes' needs and runs ONE kind of step for all of them: BVH node
 *                        pair, two triangles, instance entry, retire + refill, generate 64 camera rays, shade 64 hits,
 *                        64 background misses (see the comment inside the kernel and DESIGN.md section 3). Samples are
 *                        staged per wave and folded into the running mean in pass order (renderer.c:288-291).
 *                        Traversal stack in LDS (entry-major, conflict-free); deeper entries in a private array.
 *   k_fold_black         bounces <= 0: every sample is black, only the running mean moves.
 *   k_trace_rays         getClosestIsect for caller rays (diagnostic / parity entry).
 *   k_to_srgb8           colorToSRGB + setPixel truncation.
 * bvh_build.hip (same library): the reference's binned-SAH BVH builder on the GPU (crh_bvh_build_triangles).
 * No CPU fallback: every entry point fails with CRH_ERR_NO_DEVICE when there is no GPU.
 * Build: hipcc --offload-arch=gfx950 -O2 -ffp-contract=off -fno-slp-vectorize (see c-ray_amd/build.py).
 */
#include <hip/hip_runtime.h>
#include <dlfcn.h>

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstring>
#include <condition_variable>
#include <atomic>
#include <memory>
#include <mutex>
#include <thread>
#include <string>
#include <vector>

#include "cray_hip.h"
#define CRH_EM_POW_TABLES_IN_LDS         /* powf's two lookup tables: 512 B of LDS per workgroup (exact_math.h) */
#include "pt_device.h"
#include "scene_compile.h"
#include "ctx_access.h"

using namespace crh;

/* ---- tunables ---------------------------------------------------------------------------------- */
#define CRH_BLOCK 256            /* 4 waves of 64 */
/* Instance records in LDS (k_pathtrace; c-ray scenes are a handful of spheres and meshes): the workgroup stages line 0 of every record — what an instance
 * VISIT reads: Ainv, kind, root, ray offset, radius — when the scene has at most CRH_INST_LDS0_MAX instances, and line 1 — what FINISHING a hit reads besides:
 * A, material — when it has at most CRH_INST_LDS1_MAX; LdsStack::instLine serves them with ds_read_b128 instead of four divergent 16-byte look-ups in the
 * vector L1 per lane and line (profiles/r03a_pmc_deep.txt: that unit is busy 93 % of the time; hdr.json visits 2.05 instances per ray and finishes a hit on
 * 0.6 of them: a fifth of all its L1 look-ups). Measured against the same kernel without the tables (profiles/r03k_ab_inst_lds.log, d3 vs d4): hdr.json +4.5 %,
 * statues.json (55 instances, 4.7 visits per ray) +3 %, venus.json +3 %, the 1 M soup +0.3 %. 8 KB of LDS: the three park slots the slab offsets no longer
 * occupy (pt_device.h: PK_*) and five of the 23 traversal-stack entries (18 stay; deeper entries live in the global overflow columns — 18 / 21 / 24 / 26 entries
 * measured equal). 0 / 0 = off. */
#ifndef CRH_INST_LDS0_MAX
#define CRH_INST_LDS0_MAX 64u
#endif
#ifndef CRH_INST_LDS1_MAX
#define CRH_INST_LDS1_MAX 32u
#endif
#define CRH_INST_LDS_BYTES ((CRH_INST_LDS0_MAX + CRH_INST_LDS1_MAX) * 64u)
/* ... and the shading tables (materials 32 B, bsdf nodes 16 B, constants 16 B each) of scenes that have at most this many of each: a shaded hit reads them
 * through a chain of five or six DEPENDENT look-ups (material -> bsdf node -> mix child -> operand constant ...). 0 = off. */
#ifndef CRH_SHADE_LDS
#define CRH_SHADE_LDS 1
#endif
#define CRH_SHADE_LDS_MATERIALS 24u
#define CRH_SHADE_LDS_BSDFS 64u
#define CRH_SHADE_LDS_CONSTS 48u
#define CRH_SHADE_LDS_IMAGES 8u           /* image descriptors (8 B) and texture descriptors (32 B) */
#define CRH_SHADE_LDS_BYTES (CRH_SHADE_LDS * (CRH_SHADE_LDS_MATERIALS * 32u + CRH_SHADE_LDS_BSDFS * 16u + CRH_SHADE_LDS_CONSTS * 16u + CRH_SHADE_LDS_IMAGES * 40u))
#ifndef CRH_STACK_LDS
#define CRH_STACK_LDS ((40960 - 3968 - (int)CRH_INST_LDS_BYTES - (int)CRH_SHADE_LDS_BYTES) / 1024 - 15)     /* what the LDS holds after the id stacks, tables and the 15 park slots: 12 entries with the default tables —
                                                                                                                 * as many NODE entries as before round 4, when five of 17 held the parked top-level walk of a lane inside a BLAS */
#endif
/* CRH_LOCKSTEP(): marks a place where the lanes of a wave hand data to each other through LDS with no wave collective in between,
 * relying on what the hardware guarantees anyway — a wave executes in lockstep and its LDS operations in program order. It expands to
 * nothing here; the CPU emulation of these kernels (tests/emu/hipemu, test infrastructure), whose lanes run one after the other from
 * collective to collective, makes it a rendezvous of the wave. */
#ifndef CRH_LOCKSTEP
#define CRH_LOCKSTEP() do { } while (0)
#endif

/* ---- error plumbing ---------------------------------------------------------------------------- */
static thread_local std::string t_err;
static int fail(int code, const std::string &msg) { t_err = msg; return code; }
#define HIP_TRY(expr)                                                                               \
	do {                                                                                            \
		hipError_t e_ = (expr);                                                                     \
		if (e_ != hipSuccess) return fail(CRH_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(e_)); \
	} while (0)

/* ---- device-side helpers ------------------------------------------------------------------------ */
/* Per-lane traversal stack: the first CRH_STACK_LDS entries live in LDS (entry-major: entry i of lane l at
 * word i * 256 + l, bank = l mod 32, conflict-free), deeper entries in a private (scratch) array. The two are
 * addressed through their own address spaces — never through one generic pointer, which would turn every stack
 * access into a flat_load / flat_store. LDS + overflow cover the worst case the scene compiler can report
 * (64 + 5 + 64 + 1, bvh.c:32). The park slots (pt_device.h: PK_*) are LDS too. */
#define CRH_STACK_OVF (134 - CRH_STACK_LDS)
#define CRH_OVF_WORDS_PER_WAVE (128u * 64u)    /* overflow columns of one wave in the context's global buffer (both kernel forms: >= 134 - 6 entries x 64 lanes, for any CRH_STACK_LDS >= 6) */
typedef __attribute__((address_space(3))) uint32_t lds_u32;
typedef __attribute__((address_space(1))) uint32_t glb_u32;
/* k_pathtrace: the overflow entries live in a per-wave column block of a global buffer (entry i of lane l at ovf[(i - CRH_STACK_LDS) * 64]
 * from the lane's own base: coalesced, and no private memory behind every lane for a depth real scenes almost never reach) */
#if defined(__clang__)
typedef float crh_v4f __attribute__((ext_vector_type(4)));
__device__ __forceinline__ f4 ldsLoadF4(const lds_u32 *p) { const crh_v4f v = *(const __attribute__((address_space(3))) crh_v4f *)p; return f4{v.x, v.y, v.z, v.w}; }     /* ds_read_b128 */
#else
static inline f4 ldsLoadF4(const uint32_t *p) { f4 r; memcpy(&r, p, sizeof(r)); return r; }
#endif
struct LdsStack {
	lds_u32 *lds;        /* &s_stack[threadIdx.x] */
	lds_u32 *parkp;      /* &s_park[threadIdx.x]  */
	glb_u32 *ovf;        /* wave-uniform: &ovfAll[wave * CRH_OVF_WORDS_PER_WAVE]; the lane's column starts at its lane index */
#if CRH_SHADE_LDS
	const lds_u32 *shadeTab; /* workgroup-uniform: materials (8 words each), then bsdf nodes (4), then constants (4), or null */
	__device__ __forceinline__ DBsdf bsdfNode(const DScene &S, uint32_t i) const {
		if (shadeTab) { const f4 v = ldsLoadF4(shadeTab + CRH_SHADE_LDS_MATERIALS * 8u + i * 4u); return DBsdf{asU32(v.x), asU32(v.y), asU32(v.z), asU32(v.w)}; }
		return S.bsdfs[i];
	}
	__device__ __forceinline__ f4 constant(const DScene &S, uint32_t i) const {
		if (shadeTab) return ldsLoadF4(shadeTab + CRH_SHADE_LDS_MATERIALS * 8u + CRH_SHADE_LDS_BSDFS * 4u + i * 4u);
		return S.consts[i];
	}
	__device__ __forceinline__ ImageRef image(const DScene &S, uint32_t i) const {
		if (shadeTab) {
			const lds_u32 *it = shadeTab + CRH_SHADE_LDS_MATERIALS * 8u + CRH_SHADE_LDS_BSDFS * 4u + CRH_SHADE_LDS_CONSTS * 4u;      /* textures (8 words each), then images (2) */
			ImageRef r;
			r.im.tex = it[CRH_SHADE_LDS_IMAGES * 8u + i * 2u]; r.im.options = it[CRH_SHADE_LDS_IMAGES * 8u + i * 2u + 1u];
			r.t = DTexture{};
			if (r.im.tex != CRH_NONE) {
				const f4 a = ldsLoadF4(it + r.im.tex * 8u);
				r.t.first = asU32(a.x); r.t.width = asU32(a.y); r.t.height = asU32(a.z); r.t.m64w = asU32(a.w); r.t.m64h = it[r.im.tex * 8u + 4u];
			}
			return r;
		}
		const DImage im = S.images[i];
		return ImageRef{im, im.tex == CRH_NONE ? DTexture{} : S.textures[im.tex]};
	}
	__device__ __forceinline__ crh_material material(const DScene &S, uint32_t i) const {
		if (shadeTab) {
			const f4 a = ldsLoadF4(shadeTab + i * 8u), b = ldsLoadF4(shadeTab + i * 8u + 4u);
			crh_material m;
			m.emission[0] = a.x; m.emission[1] = a.y; m.emission[2] = a.z; m.emission[3] = a.w;
			m.ior = b.x; m.bsdf = asU32(b.y); m.pad[0] = asU32(b.z); m.pad[1] = asU32(b.w);
			return m;
		}
		return S.materials[i];
	}
#endif
#if CRH_INST_LDS_BYTES > 0
	const lds_u32 *inst0, *inst1; /* workgroup-uniform: the LDS copies of the instance records' lines 0 / 1 (16 words per instance each), or null */
	__device__ __forceinline__ InstLine instLine(const DScene &S, int32_t idx, int line) const {
		const lds_u32 *t = line == 0 ? inst0 : inst1;
		if (t) {
			const lds_u32 *p = t + (uint32_t)idx * 16u;
			return InstLine{ldsLoadF4(p), ldsLoadF4(p + 4), ldsLoadF4(p + 8), ldsLoadF4(p + 12)};
		}
		const f4 *g = (const f4 *)(S.instances + idx) + 4 * line;
		return InstLine{g[0], g[1], g[2], g[3]};
	}
#endif
	__device__ __forceinline__ void park(int i, uint32_t v) { parkp[i * CRH_BLOCK] = v; }
	__device__ __forceinline__ uint32_t unpark(int i) { return parkp[i * CRH_BLOCK]; }
	__device__ __forceinline__ void push(uint32_t i, uint32_t v) {
		if (__builtin_expect(i < CRH_STACK_LDS, 1)) lds[i * CRH_BLOCK] = v;
		else ovf[(i - CRH_STACK_LDS) * 64u + (threadIdx.x & 63u)] = v;
	}
	__device__ __forceinline__ uint32_t pop(uint32_t i) {
		uint32_t v;
		if (__builtin_expect(i < CRH_STACK_LDS, 1)) v = lds[i * CRH_BLOCK];
		else v = ovf[(i - CRH_STACK_LDS) * 64u + (threadIdx.x & 63u)];
		return v;
	}
};
/* k_trace_rays (test entry, one ray per lane, no persistent waves): overflow in a private array */
struct LdsStackPrivate {
	lds_u32 *lds;
	lds_u32 *parkp;
	uint32_t ovf[CRH_STACK_OVF];
	__device__ __forceinline__ void park(int i, uint32_t v) { parkp[i * CRH_BLOCK] = v; }
	__device__ __forceinline__ uint32_t unpark(int i) { return parkp[i * CRH_BLOCK]; }
	__device__ __forceinline__ void push(uint32_t i, uint32_t v) {
		if (__builtin_expect(i < CRH_STACK_LDS, 1)) lds[i * CRH_BLOCK] = v;
		else ovf[i - CRH_STACK_LDS] = v;
	}
	__device__ __forceinline__ uint32_t pop(uint32_t i) {
		uint32_t v;
		if (__builtin_expect(i < CRH_STACK_LDS, 1)) v = lds[i * CRH_BLOCK];
		else v = ovf[i - CRH_STACK_LDS];
		return v;
	}
};

/* Work queue over the pixel blocks of a tile list: tiles in list order, inside a tile bw x bh pixel blocks in
 * row-major order. One unit = one block for ALL passes of the dispatch (chunks of a block are folded in
 * order by the wave that owns it). */
struct BlockQueue {
	const crh_tile *tiles;
	const uint32_t *start;     /* start[t] = first block of tile t; start[ntiles] = total */
	uint32_t ntiles, total;
	uint32_t *counter;
	int bw, bh;
	/* the last tiles of the list (from firstSmall on) are cut into smaller blocks (sbw x sbh): the units handed out last
	 * are short, so the waves finish close together instead of up to one full unit apart */
	uint32_t firstSmall;
	int sbw, sbh;
	/* ... and the very last ones (from firstTiny on) into blocks of a sixteenth (tbw x tbh): what remains of the finish-line spread is about
	 * one such unit, which matters when a GPU's share of a frame is small (1/8 of it at 8 GPUs) */
	uint32_t firstTiny;
	int tbw, tbh;
	/* rolling kernel only (CRH_OPT_TAIL_SPLIT): the queue's very end (from firstMicro on) is handed out in units of about 64 paths, which is what a wave needs
	 * to keep its lanes busy while its last jobs finish — blocks of mbw x mbh pixels for all passes, or, from 128 passes per dispatch on, ONE pixel's passes in
	 * `segs` segments of segPasses: unit = start[t] + pixel * segs + segment. A pixel's segments are traced by whichever waves pull them, so their samples are
	 * staged per pixel (defer: unit (u - unit0) owns segPasses samples) and folded into the frame in pass order by k_fold_deferred behind the kernel */
	uint32_t firstMicro;
	int mbw, mbh;
	int segs, segPasses;
	uint32_t unit0;
	float *defer;
};

/* Pointers that arrive inside a by-value kernel-argument struct are generic ("flat") to the compiler; a round
 * trip through the global address space lets it emit global_load instead of flat_load for the scene arrays. */
template <class T>
__device__ __forceinline__ const T *asGlobal(const T *p) {
	return (const T *)(const __attribute__((address_space(1))) T *)p;
}
__device__ __forceinline__ DScene globalize(const DScene &S) {
	DScene G = S;
	G.nodes = asGlobal(S.nodes); G.tris = asGlobal(S.tris); G.prims = asGlobal(S.prims); G.shade = asGlobal(S.shade);
	G.instances = asGlobal(S.instances); G.materials = asGlobal(S.materials);
	G.bsdfs = asGlobal(S.bsdfs); G.consts = asGlobal(S.consts); G.images = asGlobal(S.images); G.prog = asGlobal(S.prog);
	G.textures = asGlobal(S.textures); G.texels = asGlobal(S.texels);
	G.camera = (const crh_camera *)(const __attribute__((address_space(4))) crh_camera *)S.camera;          /* constant address space: uniform scalar loads */
	return G;
}

__device__ __forceinline__ uint32_t waveSum(uint32_t v) {
	for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
	return v;
}

#define CRH_NCOUNTERS 32
/* the counter block of a context: CRH_NCOUNTERS global 64-bit counters, then CRH_NCOUNTERS 64-bit words per wave (the rolling kernel's wave-level numbers of the counting
 * instantiation: pathtrace_roll.h, CRH_WCTR) for the largest grid a context launches (8 workgroups per CU) */
#define CRH_COUNTER_WAVES_MAX(cus) ((size_t)(cus) * 8u * 4u)
/* (round 5, ADVICE r04: the per-wave words are 64-bit too — a wave's step clocks accumulate over every dispatch until crh_counters_reset, and a 32-bit word wraps after
 * about two seconds of a busy wave at the shader clock; bench.py's other_workloads dispatches run for three) */
#define CRH_COUNTER_BYTES(cus) (CRH_NCOUNTERS * sizeof(unsigned long long) + CRH_COUNTER_WAVES_MAX(cus) * CRH_NCOUNTERS * sizeof(unsigned long long))

/* scheduler weights: score of a step kind = lanes waiting for it x weight (weight ~ 1 / cost of the step) */
struct Sched { int wNode, wTri, wCtrl, swapMin, fillTo, runNum, triInRun, ctrlInRun, shadeMin;
               int sortFrom;        /* CRH_OPT_SHADE_SORT: hits are shaded in batches of few shade classes in scenes with at least this many classes; 0 = never (default) */
               int swapInRun;       /* finished + idle lanes from which a node run retires / refills in place (CRH_OPT_SCHED_RUNS << 40; default 20, 65: never; dev env CRH_SWAP_IN_RUN) */
               int rayFlags;        /* CRH_OPT_RENDER_SLABS: CRH_RAY_LITERAL or 0, given to every walk of the dispatch */
               int roundLimit; };   /* CRH_OPT_ROUND_LIMIT: scheduling rounds after which a wave gives up and flags the dispatch incomplete (k_pathtrace_roll) */
/* scheduler of the workgroup-cooperative form (pathtrace_alt.h; a plain struct, so that the context and crh_set_option do not depend on the build) */
struct SchedWg { int wNode, wTri, wCtrl, swapMin, fillTo, runNum, triInRun, ctrlInRun, linger, drainAt, maxDrainers, partialMin, walkMin; };
#define CRH_WG_PATHS 1024u
/* bits of a context's error word (host-visible memory; a kernel ORs them in, crh_synchronize / crh_framebuffer_download report and clear them) */
#define CRH_ERRFLAG_WG_WATCHDOG 1u
#define CRH_ERRFLAG_ROUND_LIMIT 2u
#define CRH_JANITOR_MIN_PATHS ((uint64_t)1 << 26)      /* a dispatch of at least this many paths (>= 20 ms of device time; the release takes 15 ms for hdr.json's arrays, huge pages or not:
                                                          * profiles/r04zz6_janitor_time.log) hides the release of an upload's host arrays; shorter dispatches — a 1 / 8 share of a frame — leave them to
                                                          * the next upload or the context's end rather than have their synchronisation wait for it */

/* Per-wave PATH TABLE in global memory: a path lives in one 128-B record (one cache line, one lane reads or writes it
 * with a few 16-B accesses) from its camera ray to its last bounce; what moves between the work stacks is its one-byte
 * slot id (LDS). Record = 8 x f4: {o, depth} {d, item} {weight, rng.lo} {radiance, rng.hi} {t, u, v, slot} {inst, -, -, -} - - */
#define CRH_PATHS 256u        /* slots per wave = the most paths a wave keeps in flight */
#define CRH_PATH_F4 8u
#define CRH_WAVE_QUEUE_FLOATS (CRH_PATHS * CRH_PATH_F4 * 4u)
/* id stacks (LDS): ids of rays waiting for a walker grow up from byte 0, the free slots grow down from byte 255 (a slot is in
 * at most one place, so the two never meet); surface hits waiting for shading are 16-bit entries (id | shade class << 8: hits are
 * shaded in batches of one class, see ST_SHADE); misses are bytes */
#define CRH_IDS_RAYS 0u
#define CRH_IDS_FREE_END 256u  /* free slot i (0 = next to be taken) sits at byte FREE_END - freeQ + i */
#define CRH_IDS_HITS 256u      /* 192 x u16 (< 64 waiting + 64 retired by one SWAP, with room to spare) */
#define CRH_HITS_MAX 192u
#define CRH_IDS_MISSES 640u    /* < 64 waiting + 64 retired by one SWAP */
#define CRH_IDS_BYTES 768u

template <class PR>
__device__ __forceinline__ void putPathRay(f4 *q, const v3 &o, const v3 &d, const PR &r, uint32_t item) {
	q[0] = f4{o.x, o.y, o.z, asF32((uint32_t)r.depth)};
	q[1] = f4{d.x, d.y, d.z, asF32(item)};
	q[2] = f4{r.wr, r.wg, r.wb, asF32((uint32_t)r.rng.state)};
	q[3] = f4{r.fr, r.fg, r.fb, asF32((uint32_t)(r.rng.state >> 32))};
}

/* What a walk may ask of its path (pt_device.h: volumes): the sampler lives in words 2.w / 3.w of the path's record, words 6..7 are free */
template <int SAMP> struct TablePort {
	f4 *q;
	__device__ __forceinline__ float draw() {
		RngT<SAMP> r;
		r.state = (uint64_t)asU32(q[2].w) | ((uint64_t)asU32(q[3].w) << 32);
		const float v = getDimension(r);
		q[2].w = asF32((uint32_t)r.state); q[3].w = asF32((uint32_t)(r.state >> 32));
		return v;
	}
	__device__ __forceinline__ void save(int i, uint32_t v) { ((uint32_t *)(q + 6))[i] = v; }
	__device__ __forceinline__ uint32_t load(int i) { return ((const uint32_t *)(q + 6))[i]; }
};

/* rank of this lane among the set bits of a ballot mask below it */
__device__ __forceinline__ uint32_t laneRank(unsigned long long m) {
	return __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
}

/* The workgroup's copies of the hot records (k_pathtrace and k_pathtrace_roll; `S` and `stk` of the calling kernel) */
#if CRH_SHADE_LDS
#define CRH_STAGE_SHADE_TABLES() \
	__shared__ __attribute__((aligned(16))) uint32_t s_shade[CRH_SHADE_LDS_BYTES / 4u]; \
	stk.shadeTab = nullptr; \
	if (S.material_count <= CRH_SHADE_LDS_MATERIALS && S.bsdf_count <= CRH_SHADE_LDS_BSDFS && S.const_count <= CRH_SHADE_LDS_CONSTS && \
		S.image_count <= CRH_SHADE_LDS_IMAGES && S.texture_count <= CRH_SHADE_LDS_IMAGES) { \
		const uint32_t texBase = CRH_SHADE_LDS_MATERIALS * 8u + CRH_SHADE_LDS_BSDFS * 4u + CRH_SHADE_LDS_CONSTS * 4u; \
		for (uint32_t i = threadIdx.x; i < S.texture_count * 8u; i += CRH_BLOCK) s_shade[texBase + i] = ((const uint32_t *)S.textures)[i]; \
		for (uint32_t i = threadIdx.x; i < S.image_count * 2u; i += CRH_BLOCK) s_shade[texBase + CRH_SHADE_LDS_IMAGES * 8u + i] = ((const uint32_t *)S.images)[i]; \
		for (uint32_t i = threadIdx.x; i < S.material_count * 8u; i += CRH_BLOCK) s_shade[i] = ((const uint32_t *)S.materials)[i]; \
		for (uint32_t i = threadIdx.x; i < S.bsdf_count * 4u; i += CRH_BLOCK) s_shade[CRH_SHADE_LDS_MATERIALS * 8u + i] = ((const uint32_t *)S.bsdfs)[i]; \
		for (uint32_t i = threadIdx.x; i < S.const_count * 4u; i += CRH_BLOCK) s_shade[CRH_SHADE_LDS_MATERIALS * 8u + CRH_SHADE_LDS_BSDFS * 4u + i] = ((const uint32_t *)S.consts)[i]; \
		__syncthreads(); \
		stk.shadeTab = (const lds_u32 *)s_shade; \
	}
#else
#define CRH_STAGE_SHADE_TABLES() do { } while (0)
#endif
#if CRH_INST_LDS_BYTES > 0
#define CRH_STAGE_INSTANCE_TABLES() \
	__shared__ __attribute__((aligned(16))) uint32_t s_inst[CRH_INST_LDS_BYTES / 4u]; \
	stk.inst0 = stk.inst1 = nullptr; \
	{ \
		const bool st0 = S.instance_count <= CRH_INST_LDS0_MAX, st1 = S.instance_count <= CRH_INST_LDS1_MAX; \
		if (st0) for (uint32_t i = threadIdx.x; i < S.instance_count * 16u; i += CRH_BLOCK) s_inst[i] = ((const uint32_t *)(S.instances + (i >> 4)))[i & 15u]; \
		if (st1) for (uint32_t i = threadIdx.x; i < S.instance_count * 16u; i += CRH_BLOCK) s_inst[CRH_INST_LDS0_MAX * 16u + i] = ((const uint32_t *)(S.instances + (i >> 4)))[16u + (i & 15u)]; \
		if (st0 || st1) __syncthreads(); \
		if (st0) stk.inst0 = (const lds_u32 *)s_inst; \
		if (st1) stk.inst1 = (const lds_u32 *)s_inst + CRH_INST_LDS0_MAX * 16u; \
	}
#else
#define CRH_STAGE_INSTANCE_TABLES() do { } while (0)
#endif

/* WPS = minimum waves per SIMD the register allocator must leave room for (1: unconstrained). */
#ifndef CRH_WPS_OVERRIDE
#define CRH_WPS_OVERRIDE WPS
#endif
/* SAMP: 0 = Random sampler (renderThread), 1 = Halton (renderThreadInteractive) */
/* the words of the wave-statistics buffer (CRH_OPT_WAVE_STATS): two per wave for up to CRH_WAVE_STATS_MAX waves, then the ray dump's descriptor — list address, rays per wave region, one
 * count per wave (crh_debug_ray_dump; written by the counting instantiations of k_pathtrace_roll only) */
#define CRH_WAVE_STATS_MAX 8192u
#define CRH_DUMP_HDR (2u * CRH_WAVE_STATS_MAX)
#define CRH_WAVE_STATS_WORDS (CRH_DUMP_HDR + 2u + CRH_WAVE_STATS_MAX)
#include "pathtrace_roll.h"          /* k_pathtrace_roll: the hot kernel (CRH_KERNEL_ROLL) */
#ifdef CRH_WITH_ALT_KERNELS
#include "pathtrace_alt.h"           /* k_pathtrace (one unit at a time) and k_pathtrace_wg (workgroup-cooperative): emulation tier and A/B variant libraries only */
#endif

__global__ __launch_bounds__(CRH_BLOCK) void k_trace_rays(const DScene Sarg, const float *rays, uint64_t n, crh_hit *hits, uint32_t rayFlags) {
	__shared__ uint32_t s_stack[CRH_STACK_LDS * CRH_BLOCK];
	__shared__ uint32_t s_park[CRH_PARK_SLOTS * CRH_BLOCK];
	const DScene S = globalize(Sarg);
	LdsStackPrivate stk;
	stk.lds = (lds_u32 *)&s_stack[threadIdx.x];
	stk.parkp = (lds_u32 *)&s_park[threadIdx.x];
	for (uint64_t i = (uint64_t)blockIdx.x * CRH_BLOCK + threadIdx.x; i < n; i += (uint64_t)gridDim.x * CRH_BLOCK) {
		Counters cnt;
		memset(&cnt, 0, sizeof(cnt));
		const v3 o{rays[6 * i], rays[6 * i + 1], rays[6 * i + 2]}, d{rays[6 * i + 3], rays[6 * i + 4], rays[6 * i + 5]};
		TravHit h;
		traverse(S, stk, o, d, h, cnt, rayFlags);
		crh_hit out;
		memset(&out, 0, sizeof(out));
		out.inst = h.inst < 0 ? -1 : (int32_t)S.instances[h.inst].orig; out.distance = h.t; out.node_tests = cnt.node_tests; out.tri_tests = cnt.tri_tests;
		if (h.inst < 0) {
			out.poly = -1; out.material = CRH_NODE_NONE;
		} else {
			const HitInfo hi = finishHit<false>(S, o, d, h);
			out.poly = hitPoly(S, h); out.uv[0] = hi.uv.x; out.uv[1] = hi.uv.y;
			out.point[0] = hi.point.x; out.point[1] = hi.point.y; out.point[2] = hi.point.z;
			out.normal[0] = hi.normal.x; out.normal[1] = hi.normal.y; out.normal[2] = hi.normal.z;
			out.material = hi.material;
		}
		hits[i] = out;
	}
}

#include "pathtrace_stream.h"        /* k_stream_walk / k_stream_shade / k_stream_fold: the streaming form (CRH_KERNEL_STREAM, round 6) */
#include "walk_probe.h"              /* k_walk_probe: the walk of k_pathtrace_roll on its own, on the path tracer's own rays (round 6: a measurement entry, crh_debug_walk_probe) */

/* bounces <= 0: pathTrace() returns black (pathtrace.c:36); only the running mean moves (renderer.c:288-291) */
__global__ void k_fold_black(const crh_render_params P, const crh_tile *tiles, uint32_t ntiles, float *fb, unsigned long long *counters) {
	for (uint32_t t = blockIdx.y; t < ntiles; t += gridDim.y) {
		const crh_tile r = tiles[t];
		const int w = r.x1 - r.x0, n = w * (r.y1 - r.y0);
		if (blockIdx.x == 0 && threadIdx.x == 0) atomicAdd(&counters[0], (unsigned long long)n * (unsigned long long)P.pass_count);   /* paths are still counted (renderer.c:283) */
		for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
			const int x = r.x0 + i % w, y = r.y0 + i / w;
			float *out = fb + ((size_t)x + (size_t)(P.image_height - (y + 1)) * (size_t)P.image_width) * 3;
			float a = out[0], b = out[1], c = out[2];
			for (int k = 0; k < P.pass_count; ++k) foldSample(a, b, c, 0.0f, 0.0f, 0.0f, P.first_pass + k + 1);
			out[0] = a; out[1] = b; out[2] = c;
		}
	}
}

/* The split pixels of a dispatch (BlockQueue: from firstMicro on, segs > 1): their passes were traced segment by segment by whichever waves pulled the
 * segments, the samples wait in Q.defer — pixel q's at [q * segs * segPasses + (pass - first_pass)] — and go into the frame here, in pass order: the
 * same running mean over the same values as foldBlockPixel's (renderer.c:288-291). One lane per pixel. */
__global__ __launch_bounds__(256) void k_fold_deferred(const crh_render_params P, const BlockQueue Q, uint32_t pixels, float *fb) {
	const uint32_t q = blockIdx.x * blockDim.x + threadIdx.x;
	if (q >= pixels) return;
	const uint32_t unit = Q.unit0 + q * (uint32_t)Q.segs;
	uint32_t lo = Q.firstMicro, hi = Q.ntiles;
	while (hi - lo > 1) { const uint32_t mid = (lo + hi) >> 1; if (Q.start[mid] <= unit) lo = mid; else hi = mid; }
	const crh_tile t = Q.tiles[lo];
	const uint32_t pix = (unit - Q.start[lo]) / (uint32_t)Q.segs, w = (uint32_t)(t.x1 - t.x0);       /* split units are single pixels: the tile's pixels row by row */
	const int x = t.x0 + (int)(pix % w), y = t.y0 + (int)(pix / w);
	float *out = fb + ((size_t)x + (size_t)(P.image_height - (y + 1)) * (size_t)P.image_width) * 3;
	const float *sp = Q.defer + (size_t)q * (size_t)Q.segs * (size_t)Q.segPasses * 3;
	float r = out[0], g = out[1], b = out[2];
	int k = 0;
	for (; k + 8 <= P.pass_count; k += 8) {          /* the loads of eight passes in flight together, the mean a serial chain (foldBlockPixel) */
		float v[24];
		for (int i = 0; i < 24; ++i) v[i] = sp[3 * k + i];
		for (int j = 0; j < 8; ++j) foldSample(r, g, b, v[3 * j], v[3 * j + 1], v[3 * j + 2], P.first_pass + k + j + 1);
	}
	for (; k < P.pass_count; ++k) foldSample(r, g, b, sp[3 * k], sp[3 * k + 1], sp[3 * k + 2], P.first_pass + k + 1);
	out[0] = r; out[1] = g; out[2] = b;
}

/* color.h:60-84 + texture.c:18-22 */
__global__ void k_to_srgb8(const float *fb, size_t n, uint8_t *out) {
	CRH_EM_POW_TABLES_INIT();
	for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
		const float v = linearToSRGB(fb[i]);
		out[i] = (unsigned char)rmin(v * 255.0f, 255.0f);
	}
}

/* debug / parity entry: the device build of exact_math.h on caller values (tests compare with the host libm bit for bit) */
__global__ void k_eval_math(int fn, const float *x, const float *y, uint64_t n, float *out) {
	CRH_EM_POW_TABLES_INIT();
	for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
		const float a = x[i], b = y ? y[i] : 0.0f;
		float r = 0.0f, c = 0.0f;
		switch (fn) {
			case CRH_MATH_SINF: r = em::sinf_(a); break;
			case CRH_MATH_COSF: r = em::cosf_(a); break;
			case CRH_MATH_SINCOSF_SIN: em::sincosf_(a, r, c); break;
			case CRH_MATH_SINCOSF_COS: em::sincosf_(a, c, r); break;
			case CRH_MATH_LOGF: r = em::logf_(a); break;
			case CRH_MATH_LOG10F: r = em::log10f_(a); break;
			case CRH_MATH_ATANF: r = em::atanf_(a); break;
			case CRH_MATH_ACOSF: r = em::acosf_(a); break;
			case CRH_MATH_ASINF: r = em::asinf_(a); break;
			case CRH_MATH_TANF: r = em::tanf_(a); break;
			case CRH_MATH_POWF: r = em::powf_(a, b); break;
			case CRH_MATH_ATAN2F: r = em::atan2f_(a, b); break;
			default: break;
		}
		out[i] = r;
	}
}

/* ---- context ------------------------------------------------------------------------------------ */
struct crh_ctx {
	int device = 0;
	void *pinned = nullptr;                  /* page-locked host scratch of the BVH builder (crh_internal_pinned) */
	/* The last upload's compiled host arrays (150 MB of touched pages for hdr.json) are released by this thread, and only once a dispatch of some length is on the device
	 * (or the next upload / the context's end asks for it): giving pages back takes 20 ms in a process that has the GPU open, and every launch or synchronisation of the
	 * caller that falls into that time waits for it (round 4, profiles/r04q_*, r04r_*) — while the device traces a frame the host has nothing else to do. */
	std::thread janitor;
	std::mutex janitorMu;
	std::condition_variable janitorCv;
	bool janitorGo = false;
	std::atomic<bool> janitorWaiting{false};
	size_t pinnedBytes = 0;
	hipStream_t stream = nullptr;
	bool ownStream = false;
	int cuCount = 0;
	int blocksPerCU = 4;
	int counterLevel = 2;
	int passChunk = 64;
	int unitItems = 2048;
	int unitsPerWave = 8;
	Sched sched = {70, 160, 120, 16, 160, 4, 12, 12, 48, 0, 20, 0, 2000000000};
	int kernel = CRH_KERNEL_ROLL;            /* CRH_OPT_KERNEL */
	SchedWg schedWg = {70, 160, 120, 16, 768, 4, 12, 12, 8, 192, 1, 16, 32};
	uint32_t *dOvf = nullptr;                /* workgroup kernel: traversal-stack overflow columns */
	size_t ovfWords = 0;
	unsigned int *hErr = nullptr, *dErr = nullptr;   /* the dispatches' error word: pinned host memory the kernels OR bits into (CRH_ERRFLAG_*), and its device address */
	std::vector<int> preloaded;              /* kernel instantiations whose code object is loaded (variantKey) */
	float *dGather = nullptr;                /* crh_frames_gather: this GPU's strips packed (senders) / every sender's slab (GPU 0) */
	size_t gatherFloats = 0;
	uint8_t *dSrgb = nullptr;                /* crh_framebuffer_to_srgb8: the 8-bit frame on the device (grown on demand, kept) */
	size_t srgbBytes = 0;
	bool traceExactSlabs = false;            /* CRH_OPT_TRACE_SLABS: crh_trace_rays walks degenerate rays like the render kernels do (exact slabs) instead of like the reference (NaN arithmetic) */
	float *dQueues = nullptr;
	size_t queueFloats = 0;
	int wavesPerSimd = 4;
	int sampler = CRH_SAMPLER_RANDOM;
	int tailPercent = 16;       /* share of a dispatch's pixels that is cut into quarter-size blocks at the end of the work queue */
	int tail2Percent = 4;       /* ... and the share at the very end that is cut into sixteenth-size blocks */
	int tailSplit = CRH_TAIL_SPLIT_DEFAULT;   /* CRH_OPT_TAIL_SPLIT: 64-path units per wave at the very end of the queue (rolling kernel); 0 = none */
	float *dDefer = nullptr;    /* samples of the split pixels of the dispatch in flight (k_fold_deferred folds them) */
	size_t deferFloats = 0;
	unsigned long long *dWaveStats = nullptr;   /* debug (CRH_OPT_WAVE_STATS) */
	/* debug (crh_debug_ray_dump / crh_debug_walk_probe): the rays the counting kernel's waves started, one region of dumpCap rays per wave; the probe's two outputs, its
	 * unit list / counter and its own stack-overflow columns (up to eight workgroups per CU) */
	float *dDump = nullptr;
	uint32_t dumpCap = 0;
	f4 *dProbeHits[2] = {nullptr, nullptr};
	int32_t *dProbeInst[2] = {nullptr, nullptr};
	uint32_t *dProbeOvf = nullptr;
	void *dProbeUnits = nullptr;
	size_t probeUnitCap = 0;
	/* the streaming form (CRH_KERNEL_STREAM; pathtrace_stream.h): two path pools of streamSlots slots (eight 16-byte planes), the walk's hit records, the cohorts' fill
	 * levels, the ring of sample slabs, the dispatch's state, the walk kernel's stack-overflow columns, and the host-visible word that says a dispatch is over */
	int streamCohorts = 16384;               /* CRH_OPT_STREAM_COHORTS: cohorts of 1024 paths in a pool (the most; small dispatches take fewer) */
	int streamGroup = 8;                     /* iterations enqueued between two looks at the completion word */
	f4 *dStreamPlanes = nullptr;
	f4 *dStreamHit = nullptr;
	int32_t *dStreamHitInst = nullptr;
	uint32_t *dStreamCount = nullptr;
	size_t streamSlots = 0;
	float *dStreamSlab = nullptr;
	size_t streamSlabFloats = 0;
	StreamCtl *dStreamCtl = nullptr;
	uint32_t *dStreamOvf = nullptr;
	size_t streamOvfWords = 0;
	unsigned int *hStreamDone = nullptr, *dStreamDone = nullptr;
	unsigned int streamSeq = 0;
	uint64_t streamIterations = 0;           /* iterations the last streamed dispatch enqueued (crh_debug_stream_stats) */
	hipEvent_t streamEv[2] = {nullptr, nullptr};
	uint32_t lastGrid = 0;
	char lastKernel[64] = "";               /* the instantiation launchPathtrace launched last (crh_last_kernel_name) */
	float *dStage = nullptr;
	size_t stageFloats = 0;
	bool haveScene = false;
	bool hasPrograms = true;     /* the compiled scene contains node programs -> kernel variant with runProgram() */
	bool hasVolumes = false;     /* walks draw from the path's sampler: crh_trace_rays (caller rays, no path) refuses such scenes */
	DScene d;                              /* device pointers */
	int walk = CRH_WALK_BINARY;            /* CRH_OPT_WALK: what the NEXT upload prepares and the render kernel walks */
	bool haveWide = false;                 /* the resident scene has a wide copy of its BVHs (behind the triangles, in the nodes' allocation) */
	uint32_t wideTlasRoot = 0;             /* ... whose top-level root is this reference (scenes with more than one top-level node) */
	std::vector<void *> sceneAllocs;
	unsigned long long *dCounters = nullptr;
	uint32_t *dWork = nullptr;             /* ring of work counters, one per in-flight launch */
	uint32_t workSlot = 0;
	/* per-launch tile lists: a ring of persistent device buffers, each with a pinned host twin (no hipMalloc and no blocking copy per
	 * launch: one hipMemcpyAsync on the launch stream); a slot is reused only after the launch that read it has finished */
	struct TileSlot { void *dev = nullptr; void *host = nullptr; size_t cap = 0; hipEvent_t done = nullptr; bool inFlight = false; };
	TileSlot tileSlots[64];
	struct Timed { hipEvent_t a, b; };
	std::vector<Timed> pendingTimes;
	std::vector<Timed> eventPool;
	float lastMs = 0.0f;
	double totalMs = 0.0;
	uint64_t launches = 0;
};
#define CRH_WORK_SLOTS 64
static_assert(sizeof(((crh_ctx *)nullptr)->tileSlots) / sizeof(crh_ctx::TileSlot) == CRH_WORK_SLOTS, "one tile slot per work counter");

static int setDevice(crh_ctx *c) {
	HIP_TRY(hipSetDevice(c->device));
	return CRH_OK;
}

static void freeScene(crh_ctx *c) {
	for (void *p : c->sceneAllocs) (void)hipFree(p);
	c->sceneAllocs.clear();
	c->haveScene = false;
}

static int resolveTimes(crh_ctx *c, bool wait) {
	size_t done = 0;
	for (auto &t : c->pendingTimes) {
		if (wait) HIP_TRY(hipEventSynchronize(t.b));
		else if (hipEventQuery(t.b) != hipSuccess) break;
		float ms = 0.0f;
		HIP_TRY(hipEventElapsedTime(&ms, t.a, t.b));
		c->lastMs = ms;
		c->totalMs += ms;
		c->eventPool.push_back(t);
		++done;
	}
	c->pendingTimes.erase(c->pendingTimes.begin(), c->pendingTimes.begin() + done);
	return CRH_OK;
}

/* A kernel that gave up instead of hanging has said so in the context's error word: report it (call with the stream drained). The word is pinned host memory
 * the device writes through its mapped address — reading it costs nothing, so every kernel form is covered on every synchronize / download. */
static int checkWatchdog(crh_ctx *c) {
	const unsigned int err = c->hErr ? *(volatile unsigned int *)c->hErr : 0u;
	if (!err) return CRH_OK;
	*(volatile unsigned int *)c->hErr = 0u;
	if (err & CRH_ERRFLAG_ROUND_LIMIT) return fail(CRH_ERR_HIP, "k_pathtrace_roll: a wave reached the round limit (CRH_OPT_ROUND_LIMIT) and gave up: incomplete frame");
	return fail(CRH_ERR_HIP, "k_pathtrace_wg: watchdog abort (flag " + std::to_string(err) + "): incomplete frame");
}

template <class T>
static int upload(crh_ctx *c, const T *host, size_t count, const T **dev) {
	void *p = nullptr;
	const size_t bytes = std::max<size_t>(count, 1) * sizeof(T);
	HIP_TRY(hipMalloc(&p, bytes));
	c->sceneAllocs.push_back(p);
	if (count) HIP_TRY(hipMemcpy(p, host, count * sizeof(T), hipMemcpyHostToDevice));
	else HIP_TRY(hipMemset(p, 0, bytes));
	*dev = (const T *)p;
	return CRH_OK;
}

/* the rolling kernel renders the dispatch: it is the selected form, or the selected form is the streaming one (CRH_KERNEL_STREAM) and cannot serve it (streamServes) */
static bool rollForm(const crh_ctx *c) { return c->kernel == CRH_KERNEL_ROLL || c->kernel == CRH_KERNEL_STREAM; }
/* the dispatch walks the wide copy: the option is on, the resident scene has one, and neither rare features nor the Halton sampler are in play */
static bool wideWalk(const crh_ctx *c) {
	return c->walk == CRH_WALK_WIDE4 && c->haveWide && rollForm(c) && !c->hasPrograms && c->sampler == CRH_SAMPLER_RANDOM;
}

/* Launch the instantiation the context's options select (counter level, rare features, sampler; the kernel form in builds that hold more than one). */
static hipError_t launchPathtrace(crh_ctx *c, uint32_t grid, const crh_render_params *P, const BlockQueue &Q, float *dev_fb, int chunk) {
#define CRH_LAUNCH_ROLL(LEVEL, PROG, SAMP) do { snprintf(c->lastKernel, sizeof(c->lastKernel), "k_pathtrace_roll<%d,4,%s,%d>", LEVEL, PROG ? "true" : "false", SAMP); \
		hipLaunchKernelGGL((k_pathtrace_roll<LEVEL, 4, PROG, SAMP>), dim3(grid), dim3(CRH_BLOCK), 0, c->stream, c->d, *P, Q, dev_fb, \
						   c->dCounters, c->dStage, chunk, c->dWaveStats, c->sched, c->dQueues, c->dOvf, c->dErr); } while (0)
#define CRH_LAUNCH_ROLL_WIDE(LEVEL) do { snprintf(c->lastKernel, sizeof(c->lastKernel), "k_pathtrace_roll<%d,4,false,0,wide4>", LEVEL); \
		hipLaunchKernelGGL((k_pathtrace_roll<LEVEL, 4, false, 0, true>), dim3(grid), dim3(CRH_BLOCK), 0, c->stream, dw, *P, Q, dev_fb, \
						   c->dCounters, c->dStage, chunk, c->dWaveStats, c->sched, c->dQueues, c->dOvf, c->dErr); } while (0)
	const bool halton = c->sampler == CRH_SAMPLER_HALTON;
	(void)halton;
	if (rollForm(c)) {
		/* CRH_OPT_WALK = CRH_WALK_WIDE4 (an option, round 5): scenes without rare features, the random sampler */
		if (wideWalk(c)) {
			DScene dw = c->d;
			if (dw.tlas_node_count > 1u) dw.tlas_root = c->wideTlasRoot;
#ifdef CRH_DEV_ONLY_BENCH_VARIANT
#if defined(CRH_DEV_ONLY_LEVEL2)
			CRH_LAUNCH_ROLL_WIDE(2);
#else
			CRH_LAUNCH_ROLL_WIDE(1);
#endif
#else
			if (c->counterLevel >= 2) CRH_LAUNCH_ROLL_WIDE(2); else CRH_LAUNCH_ROLL_WIDE(1);
#endif
			return hipGetLastError();
		}
#ifdef CRH_DEV_ONLY_BENCH_VARIANT                 /* development builds (tools/build_variant.sh, tools/kernel_regs.py): one instantiation compiles in seconds */
#if defined(CRH_DEV_ONLY_LEVEL2)              /* the counting instantiation (tools/emu_sched_stats.py) */
		CRH_LAUNCH_ROLL(2, true, 0);
#elif defined(CRH_DEV_ONLY_PROG)
		CRH_LAUNCH_ROLL(1, true, 0);
#else
		CRH_LAUNCH_ROLL(1, false, 0);
#endif
#else
		if (c->counterLevel >= 2) {
			if (c->hasPrograms) { if (halton) CRH_LAUNCH_ROLL(2, true, 1); else CRH_LAUNCH_ROLL(2, true, 0); }
			else { if (halton) CRH_LAUNCH_ROLL(2, false, 1); else CRH_LAUNCH_ROLL(2, false, 0); }
		} else {
			if (c->hasPrograms) { if (halton) CRH_LAUNCH_ROLL(1, true, 1); else CRH_LAUNCH_ROLL(1, true, 0); }
			else { if (halton) CRH_LAUNCH_ROLL(1, false, 1); else CRH_LAUNCH_ROLL(1, false, 0); }
		}
#endif
		return hipGetLastError();
	}
#undef CRH_LAUNCH_ROLL
#undef CRH_LAUNCH_ROLL_WIDE
#ifdef CRH_WITH_ALT_KERNELS
	const bool wg = c->kernel == CRH_KERNEL_WG;
#define CRH_LAUNCH(LEVEL, WPS, PROG, SAMP) do { snprintf(c->lastKernel, sizeof(c->lastKernel), "k_pathtrace<%d,%d,%s,%d>", LEVEL, WPS, PROG ? "true" : "false", SAMP); \
		hipLaunchKernelGGL((k_pathtrace<LEVEL, WPS, PROG, SAMP>), dim3(grid), dim3(CRH_BLOCK), 0, c->stream, c->d, *P, Q, dev_fb, \
						   c->dCounters, c->dStage, chunk, c->dWaveStats, c->sched, c->dQueues, c->dOvf); } while (0)
#define CRH_LAUNCH2(LEVEL, WPS) do { if (c->hasPrograms) CRH_LAUNCH(LEVEL, WPS, true, 0); else CRH_LAUNCH(LEVEL, WPS, false, 0); } while (0)
#define CRH_LAUNCH_WG(LEVEL, PROG, SAMP) do { snprintf(c->lastKernel, sizeof(c->lastKernel), "k_pathtrace_wg<%d,%s,%d>", LEVEL, PROG ? "true" : "false", SAMP); \
		hipLaunchKernelGGL((k_pathtrace_wg<LEVEL, PROG, SAMP>), dim3(grid), dim3(CRH_BLOCK), 0, c->stream, c->d, *P, Q, dev_fb, \
						   c->dCounters, c->dStage, chunk, c->schedWg, c->dQueues, c->dOvf, c->dErr); } while (0)
#ifdef CRH_DEV_ONLY_BENCH_VARIANT
#ifdef CRH_DEV_ONLY_PROG
	if (wg) CRH_LAUNCH_WG(1, true, 0); else CRH_LAUNCH(1, 4, true, 0);
#else
	if (wg) CRH_LAUNCH_WG(1, false, 0); else CRH_LAUNCH(1, 4, false, 0);
#endif
#else
	if (wg) {
		if (c->counterLevel >= 2) {
			if (c->hasPrograms) { if (halton) CRH_LAUNCH_WG(2, true, 1); else CRH_LAUNCH_WG(2, true, 0); }
			else { if (halton) CRH_LAUNCH_WG(2, false, 1); else CRH_LAUNCH_WG(2, false, 0); }
		} else {
			if (c->hasPrograms) { if (halton) CRH_LAUNCH_WG(1, true, 1); else CRH_LAUNCH_WG(1, true, 0); }
			else { if (halton) CRH_LAUNCH_WG(1, false, 1); else CRH_LAUNCH_WG(1, false, 0); }
		}
	} else
	if (halton) {          /* interactive mode: the 128-register variants only */
		if (c->counterLevel >= 2) { if (c->hasPrograms) CRH_LAUNCH(2, 4, true, 1); else CRH_LAUNCH(2, 4, false, 1); }
		else { if (c->hasPrograms) CRH_LAUNCH(1, 4, true, 1); else CRH_LAUNCH(1, 4, false, 1); }
	}
	else if (c->counterLevel >= 2) { if (c->wavesPerSimd >= 4) CRH_LAUNCH2(2, 4); else CRH_LAUNCH2(2, 1); }
	else { if (c->wavesPerSimd >= 4) CRH_LAUNCH2(1, 4); else CRH_LAUNCH2(1, 1); }
#endif
#undef CRH_LAUNCH2
#undef CRH_LAUNCH
#undef CRH_LAUNCH_WG
	return hipGetLastError();
#else
	return hipErrorInvalidDeviceFunction;          /* (crh_set_option refuses the other forms in this build) */
#endif
}

/* Load the code object of the selected instantiation now (HIP loads kernels lazily, ~40 ms on first launch) with a launch that finds
 * an empty work queue: crh_scene_upload calls it, so a renderer's first frame is not the one that pays for it. */
static int variantKey(const crh_ctx *c) { return (c->hasPrograms ? 1 : 0) | (c->sampler << 1) | (c->counterLevel << 2) | (c->wavesPerSimd << 4) | (c->kernel << 8) | (wideWalk(c) ? 1 << 12 : 0); }
static int preloadKernel(crh_ctx *c, bool again = false) {
	crh_render_params P;
	memset(&P, 0, sizeof(P));
	BlockQueue Q;
	memset(&Q, 0, sizeof(Q));
	Q.counter = c->dWork;                 /* any valid counter: total = 0, every wave leaves at once */
	Q.bw = Q.bh = Q.sbw = Q.sbh = Q.tbw = Q.tbh = Q.mbw = Q.mbh = Q.segs = 1;
	/* the per-wave path tables, stack-overflow columns and sample slabs of a full-size dispatch at the default unit size: allocated here rather than by the first frame */
	const size_t waves = (size_t)c->cuCount * c->blocksPerCU * (CRH_BLOCK / 64);
	if (waves * CRH_OVF_WORDS_PER_WAVE > c->ovfWords) {
		if (c->dOvf) HIP_TRY(hipFree(c->dOvf));
		c->dOvf = nullptr; c->ovfWords = 0;
		HIP_TRY(hipMalloc((void **)&c->dOvf, waves * CRH_OVF_WORDS_PER_WAVE * sizeof(uint32_t)));
		c->ovfWords = waves * CRH_OVF_WORDS_PER_WAVE;
	}
	if (waves * CRH_WAVE_QUEUE_FLOATS > c->queueFloats) {
		if (c->dQueues) HIP_TRY(hipFree(c->dQueues));
		c->dQueues = nullptr; c->queueFloats = 0;
		HIP_TRY(hipMalloc((void **)&c->dQueues, waves * CRH_WAVE_QUEUE_FLOATS * sizeof(float)));
		c->queueFloats = waves * CRH_WAVE_QUEUE_FLOATS;
	}
	const size_t slabs = rollForm(c) ? CRH_ROLL_SLOTS : 1u;        /* one sample slab per open job */
	if (waves * slabs * (size_t)c->unitItems * 3 > c->stageFloats) {
		if (c->dStage) HIP_TRY(hipFree(c->dStage));
		c->dStage = nullptr; c->stageFloats = 0;
		HIP_TRY(hipMalloc((void **)&c->dStage, waves * slabs * (size_t)c->unitItems * 3 * sizeof(float)));
		c->stageFloats = waves * slabs * (size_t)c->unitItems * 3;
	}
	const int key = variantKey(c);
	const bool known = std::find(c->preloaded.begin(), c->preloaded.end(), key) != c->preloaded.end();
	if (known && !again) return CRH_OK;      /* crh_context_prepare has been here */
	{   /* the copy paths a dispatch uses — pinned host -> device (its tile list) and back — are set up by the runtime on first use (measured: the first
		 * dispatch's kernel started 7-15 ms after its launch, behind its own 64-byte tile list): first use is here */
		crh_ctx::TileSlot &ts = c->tileSlots[0];
		if (!ts.dev) {
			HIP_TRY(hipMalloc(&ts.dev, 4096));
			HIP_TRY(hipHostMalloc(&ts.host, 4096, hipHostMallocDefault));
			ts.cap = 4096;
			memset(ts.host, 0, 4096);
		}
		HIP_TRY(hipMemcpyAsync(ts.dev, ts.host, 64, hipMemcpyHostToDevice, c->stream));
		HIP_TRY(hipMemcpyAsync(ts.host, ts.dev, 64, hipMemcpyDeviceToHost, c->stream));
	}
	HIP_TRY(hipMemsetAsync(c->dWork, 0, sizeof(uint32_t), c->stream));
	/* a FULL-SIZE grid (every wave finds the queue empty and leaves): the runtime sizes the queue's scratch memory by the waves a dispatch can have in flight,
	 * and allocates it when a dispatch first needs that much — for a one-workgroup preload that was the first frame: its kernel started 8-24 ms after its
	 * launch (round 3, CRH_TRACE_SYNC; an empty launch and every synchronize the API offers directly in front of it changed nothing) */
	/* ... between the two timing events a dispatch records around its kernel (their first use on a stream is a set-up step of the runtime as well) */
	crh_ctx::Timed ev;
	if (!c->eventPool.empty()) { ev = c->eventPool.back(); c->eventPool.pop_back(); }
	else { HIP_TRY(hipEventCreate(&ev.a)); HIP_TRY(hipEventCreate(&ev.b)); }
	HIP_TRY(hipEventRecord(ev.a, c->stream));
	const hipError_t e = launchPathtrace(c, (uint32_t)(c->cuCount * c->blocksPerCU), &P, Q, nullptr, 1);
	HIP_TRY(hipEventRecord(ev.b, c->stream));
	c->eventPool.push_back(ev);
	if (e != hipSuccess) return fail(CRH_ERR_HIP, std::string("kernel preload: ") + hipGetErrorString(e));
	HIP_TRY(hipMemsetAsync(c->dWork, 0, sizeof(uint32_t), c->stream));          /* its waves have drawn from the counter: zero again for the dispatch that takes slot 0 */
	HIP_TRY(hipStreamSynchronize(c->stream));
	if (!known) c->preloaded.push_back(key);
	return CRH_OK;
}

extern "C" {

int crh_abi_version(void) { return CRH_ABI_VERSION; }
const char *crh_last_error(void) { return t_err.c_str(); }

int crh_device_count(void) {
	int n = 0;
	if (hipGetDeviceCount(&n) != hipSuccess) return 0;
	return n < 0 ? 0 : n;
}

int crh_context_create(int device, void *stream, crh_ctx **out) {
	if (!out) return fail(CRH_ERR_INVALID, "crh_context_create: out is NULL");
	*out = nullptr;
	const int n = crh_device_count();
	if (n <= 0) return fail(CRH_ERR_NO_DEVICE, "no HIP device visible (libcray_hip has no CPU fallback)");
	if (device < 0 || device >= n) return fail(CRH_ERR_INVALID, "device index out of range");
	crh_ctx *c = new (std::nothrow) crh_ctx();
	if (!c) return fail(CRH_ERR_NOMEM, "out of host memory");
	c->device = device;
	hipError_t e = hipSetDevice(device);
	hipDeviceProp_t prop;
	if (e == hipSuccess) e = hipGetDeviceProperties(&prop, device);
	if (e == hipSuccess) {
		c->cuCount = prop.multiProcessorCount;
		if (stream) c->stream = (hipStream_t)stream;
		else { e = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking); c->ownStream = (e == hipSuccess); }
	}
	if (e == hipSuccess) e = hipMalloc((void **)&c->dCounters, CRH_COUNTER_BYTES(c->cuCount));
	if (e == hipSuccess) e = hipMemset(c->dCounters, 0, CRH_COUNTER_BYTES(c->cuCount));
	if (e == hipSuccess) e = hipMalloc((void **)&c->dWork, CRH_WORK_SLOTS * sizeof(uint32_t));
	if (e == hipSuccess) e = hipMemset(c->dWork, 0, CRH_WORK_SLOTS * sizeof(uint32_t));      /* a work counter is zero whenever a dispatch takes it: crh_render_tiles resets it BEHIND the kernel */
	if (e == hipSuccess) e = hipHostMalloc((void **)&c->hErr, sizeof(unsigned int), hipHostMallocDefault);
	if (e == hipSuccess) { *c->hErr = 0u; e = hipHostGetDevicePointer((void **)&c->dErr, c->hErr, 0); }
	if (e != hipSuccess) {
		const std::string msg = std::string("crh_context_create: ") + hipGetErrorString(e);
		crh_context_destroy(c);
		return fail(CRH_ERR_HIP, msg);
	}
	const char *env = getenv("CRH_BLOCKS_PER_CU");
	if (env && atoi(env) > 0) c->blocksPerCU = atoi(env) > 8 ? 8 : atoi(env);          /* (1..8, like CRH_OPT_BLOCKS_PER_CU: the counter block is sized for 8) */
	env = getenv("CRH_SWAP_IN_RUN");              /* dev: A/B of the in-run retire / refill threshold */
	if (env && atoi(env) >= 1 && atoi(env) <= 65) c->sched.swapInRun = atoi(env);
	env = getenv("CRH_KERNEL");                     /* dev: a process's default kernel form (A/B runs of unmodified hosts): "stream" or "roll" */
	if (env && !strcmp(env, "stream")) c->kernel = CRH_KERNEL_STREAM;
	if (env && !strcmp(env, "roll")) c->kernel = CRH_KERNEL_ROLL;
	env = getenv("CRH_STREAM_COHORTS");
	if (env && atoi(env) >= 1 && atoi(env) <= 262144) c->streamCohorts = atoi(env);
	env = getenv("CRH_STREAM_GROUP");
	if (env && atoi(env) >= 1 && atoi(env) <= 1024) c->streamGroup = atoi(env);
	env = getenv("CRH_TAIL_SPLIT");                 /* dev: the default of CRH_OPT_TAIL_SPLIT for this process (A/B runs of unmodified hosts) */
	if (env && atoi(env) >= 0 && atoi(env) <= 64) c->tailSplit = atoi(env);
	*out = c;
	return CRH_OK;
}

/* lets the janitor thread (see crh_ctx) go; join: the caller needs it gone (next upload, end of the context) */
static void releaseJanitor(crh_ctx *c, bool join) {
	if (c->janitorWaiting.exchange(false)) {
		{ std::lock_guard<std::mutex> lk(c->janitorMu); c->janitorGo = true; }
		c->janitorCv.notify_one();
	}
	if (join && c->janitor.joinable()) c->janitor.join();
}

int crh_context_destroy(crh_ctx *c) {
	if (!c) return CRH_OK;
	(void)hipSetDevice(c->device);
	if (c->stream) (void)hipStreamSynchronize(c->stream);
	freeScene(c);
	for (auto &ts : c->tileSlots) {
		if (ts.dev) (void)hipFree(ts.dev);
		if (ts.host) (void)hipHostFree(ts.host);
		if (ts.done) (void)hipEventDestroy(ts.done);
	}
	for (auto &t : c->pendingTimes) { (void)hipEventDestroy(t.a); (void)hipEventDestroy(t.b); }
	for (auto &t : c->eventPool) { (void)hipEventDestroy(t.a); (void)hipEventDestroy(t.b); }
	if (c->dCounters) (void)hipFree(c->dCounters);
	if (c->dWork) (void)hipFree(c->dWork);
	if (c->dStage) (void)hipFree(c->dStage);
	if (c->dDefer) (void)hipFree(c->dDefer);
	if (c->dQueues) (void)hipFree(c->dQueues);
	if (c->dOvf) (void)hipFree(c->dOvf);
	if (c->hErr) (void)hipHostFree(c->hErr);
	if (c->pinned) (void)hipHostFree(c->pinned);
	releaseJanitor(c, true);
	if (c->dSrgb) (void)hipFree(c->dSrgb);
	if (c->dGather) (void)hipFree(c->dGather);
	if (c->dWaveStats) (void)hipFree(c->dWaveStats);
	if (c->dDump) (void)hipFree(c->dDump);
	for (int i = 0; i < 2; ++i) { if (c->dProbeHits[i]) (void)hipFree(c->dProbeHits[i]); if (c->dProbeInst[i]) (void)hipFree(c->dProbeInst[i]); }
	if (c->dProbeOvf) (void)hipFree(c->dProbeOvf);
	if (c->dProbeUnits) (void)hipFree(c->dProbeUnits);
	if (c->dStreamPlanes) (void)hipFree(c->dStreamPlanes);
	if (c->dStreamHit) (void)hipFree(c->dStreamHit);
	if (c->dStreamHitInst) (void)hipFree(c->dStreamHitInst);
	if (c->dStreamCount) (void)hipFree(c->dStreamCount);
	if (c->dStreamSlab) (void)hipFree(c->dStreamSlab);
	if (c->dStreamCtl) (void)hipFree(c->dStreamCtl);
	if (c->dStreamOvf) (void)hipFree(c->dStreamOvf);
	if (c->hStreamDone) (void)hipHostFree(c->hStreamDone);
	for (int i = 0; i < 2; ++i) if (c->streamEv[i]) (void)hipEventDestroy(c->streamEv[i]);
	if (c->ownStream && c->stream) (void)hipStreamDestroy(c->stream);
	delete c;
	return CRH_OK;
}

int crh_set_option(crh_ctx *c, int option, int64_t value) {
	if (!c) return fail(CRH_ERR_INVALID, "crh_set_option: ctx is NULL");
	switch (option) {
		case CRH_OPT_COUNTER_LEVEL:
			if (value < 1 || value > 2) return fail(CRH_ERR_INVALID, "counter level must be 1 or 2");
			c->counterLevel = (int)value; return CRH_OK;
		case CRH_OPT_BLOCKS_PER_CU:
			if (value < 1 || value > 8) return fail(CRH_ERR_INVALID, "blocks per CU must be 1..8");
			c->blocksPerCU = (int)value; return CRH_OK;
		case CRH_OPT_WAVE_STATS:
			if (value && !c->dWaveStats) { if (hipMalloc((void **)&c->dWaveStats, CRH_WAVE_STATS_WORDS * sizeof(unsigned long long)) != hipSuccess) return fail(CRH_ERR_HIP, "wave stats alloc");
				if (hipMemset(c->dWaveStats, 0, CRH_WAVE_STATS_WORDS * sizeof(unsigned long long)) != hipSuccess) return fail(CRH_ERR_HIP, "wave stats clear"); }
			if (!value && c->dWaveStats) { (void)hipFree(c->dWaveStats); c->dWaveStats = nullptr; }
			return CRH_OK;
		case CRH_OPT_WAVES_PER_SIMD:
			if (value != 1 && value != 4) return fail(CRH_ERR_INVALID, "waves per SIMD must be 1 (unconstrained) or 4");
			c->wavesPerSimd = (int)value; return CRH_OK;
		case CRH_OPT_SCHED_WEIGHTS: {  /* four 12-bit fields, low to high: node, tri, ctrl weights; finished + idle lanes that trigger a swap step */
			Sched k = c->sched;
			k.wNode = (int)(value & 0xFFF); k.wTri = (int)((value >> 12) & 0xFFF); k.wCtrl = (int)((value >> 24) & 0xFFF); k.swapMin = (int)((value >> 36) & 0xFFF);
			if (value < 0 || k.wNode < 1 || k.wTri < 1 || k.wCtrl < 1 || k.swapMin < 1 || k.swapMin > 64) return fail(CRH_ERR_INVALID, "bad scheduler parameters");
			c->sched = k;
			c->schedWg.wNode = k.wNode; c->schedWg.wTri = k.wTri; c->schedWg.wCtrl = k.wCtrl; c->schedWg.swapMin = k.swapMin;
			return CRH_OK;
		}
		case CRH_OPT_SCHED_RUNS: {     /* fillTo | runNum << 12 | triInRun << 16 | ctrlInRun << 24 | shadeMin << 32 (0: keep) | swapInRun << 40 (0: keep) */
			Sched k = c->sched;
			k.fillTo = (int)(value & 0xFFF); k.runNum = (int)((value >> 12) & 0xF); k.triInRun = (int)((value >> 16) & 0xFF); k.ctrlInRun = (int)((value >> 24) & 0xFF);
			if ((value >> 32) & 0xFF) k.shadeMin = (int)((value >> 32) & 0xFF);
			if ((value >> 40) & 0xFF) k.swapInRun = (int)((value >> 40) & 0xFF);
			if (value < 0 || k.fillTo > 192 || k.runNum < 1 || k.runNum > 8 || k.triInRun < 1 || k.triInRun > 65 || k.ctrlInRun < 1 || k.ctrlInRun > 65 || k.shadeMin < 1 || k.shadeMin > 128 || k.swapInRun < 1 || k.swapInRun > 65) return fail(CRH_ERR_INVALID, "bad scheduler run parameters");
			c->sched = k;
			c->schedWg.runNum = k.runNum; c->schedWg.triInRun = k.triInRun; c->schedWg.ctrlInRun = k.ctrlInRun;
			return CRH_OK;
		}
		case CRH_OPT_TAIL_PERCENT:
			if (value < 0 || (value & 0xFF) > 50 || (value >> 8) > 51) return fail(CRH_ERR_INVALID, "tail percent must be 0..50 (| (second-level percent + 1) << 8)");
			c->tailPercent = (int)(value & 0xFF) > 50 ? 50 : (int)(value & 0xFF);
			if (value >> 8) c->tail2Percent = (int)((value >> 8) & 0xFF) - 1;       /* second level: (percent + 1) << 8, so that plain values keep their meaning */
			return CRH_OK;
		case CRH_OPT_SAMPLER:
			if (value != CRH_SAMPLER_RANDOM && value != CRH_SAMPLER_HALTON) return fail(CRH_ERR_INVALID, "sampler must be CRH_SAMPLER_RANDOM or CRH_SAMPLER_HALTON");
			c->sampler = (int)value; return CRH_OK;
		case CRH_OPT_UNITS_PER_WAVE:
			if (value < 1 || value > 1024) return fail(CRH_ERR_INVALID, "units per wave must be 1..1024");
			c->unitsPerWave = (int)value; return CRH_OK;
		case CRH_OPT_UNIT_ITEMS:
			if (value < 64 || value > (1 << 20)) return fail(CRH_ERR_INVALID, "unit items must be 64..2^20");
			c->unitItems = (int)value; return CRH_OK;
		case CRH_OPT_PASS_CHUNK:
			if (value < 1 || value > 4096) return fail(CRH_ERR_INVALID, "pass chunk must be 1..4096");
			c->passChunk = (int)value; return CRH_OK;
		case CRH_OPT_TRACE_SLABS:
			if (value != CRH_TRACE_SLABS_LITERAL && value != CRH_TRACE_SLABS_EXACT) return fail(CRH_ERR_INVALID, "trace slabs must be CRH_TRACE_SLABS_LITERAL or CRH_TRACE_SLABS_EXACT");
			c->traceExactSlabs = value == CRH_TRACE_SLABS_EXACT; return CRH_OK;
		case CRH_OPT_TAIL_SPLIT:
			if (value < 0 || value > 64) return fail(CRH_ERR_INVALID, "tail split: 0 (off) or the number of 64-path units per wave (1..64) the work queue ends with");
			c->tailSplit = (int)value; return CRH_OK;
		case CRH_OPT_SHADE_SORT:
			if (value < 0 || value > 8) return fail(CRH_ERR_INVALID, "shade sort: 0 (never) or the number of shade classes (1..8) from which a scene's hits are shaded in batches of few classes");
			c->sched.sortFrom = (int)value; return CRH_OK;
		case CRH_OPT_KERNEL:
			if (value != CRH_KERNEL_WAVE && value != CRH_KERNEL_WG && value != CRH_KERNEL_ROLL && value != CRH_KERNEL_STREAM) return fail(CRH_ERR_INVALID, "kernel must be CRH_KERNEL_ROLL, CRH_KERNEL_STREAM, CRH_KERNEL_WAVE or CRH_KERNEL_WG");
#ifndef CRH_WITH_ALT_KERNELS
			if (value != CRH_KERNEL_ROLL && value != CRH_KERNEL_STREAM) return fail(CRH_ERR_UNSUPPORTED, "this library holds k_pathtrace_roll only: CRH_KERNEL_WAVE / CRH_KERNEL_WG need a build with -DCRH_WITH_ALT_KERNELS (tests/emu, tools/build_variant.sh)");
#endif
			c->kernel = (int)value; return CRH_OK;
		case CRH_OPT_STREAM_COHORTS:
			if (value < 1 || value > 262144) return fail(CRH_ERR_INVALID, "stream cohorts: 1..262144 cohorts of 1024 paths in a pool");
			c->streamCohorts = (int)value; return CRH_OK;
		case CRH_OPT_RENDER_SLABS:
			if (value != CRH_TRACE_SLABS_LITERAL && value != CRH_TRACE_SLABS_EXACT) return fail(CRH_ERR_INVALID, "render slabs must be CRH_TRACE_SLABS_LITERAL or CRH_TRACE_SLABS_EXACT");
			c->sched.rayFlags = value == CRH_TRACE_SLABS_LITERAL ? (int)CRH_RAY_LITERAL : 0; return CRH_OK;
		case CRH_OPT_WALK:
			if (value != CRH_WALK_BINARY && value != CRH_WALK_WIDE4) return fail(CRH_ERR_INVALID, "walk must be CRH_WALK_BINARY or CRH_WALK_WIDE4");
			c->walk = (int)value; return CRH_OK;
		case CRH_OPT_ROUND_LIMIT:
			if (value < 2 || value > 2000000000) return fail(CRH_ERR_INVALID, "round limit must be 2..2e9 scheduling rounds per wave");
			c->sched.roundLimit = (int)value; return CRH_OK;
		case CRH_OPT_SCHED_WG: {       /* linger | drainAt << 8 | maxDrainers << 20 | partialMin << 24 | walkMin << 32 | fillTo << 40 */
			SchedWg k = c->schedWg;
			k.linger = (int)(value & 0xFF); k.drainAt = (int)((value >> 8) & 0xFFF); k.maxDrainers = (int)((value >> 20) & 0xF);
			k.partialMin = (int)((value >> 24) & 0xFF); k.walkMin = (int)((value >> 32) & 0xFF); k.fillTo = (int)((value >> 40) & 0xFFF);
			if (value < 0 || k.drainAt < 1 || k.maxDrainers > 4 || k.partialMin < 1 || k.walkMin < 1 || k.walkMin > 64 || k.fillTo > 960)
				return fail(CRH_ERR_INVALID, "bad workgroup scheduler parameters");
			c->schedWg = k;
			return CRH_OK;
		}
		default: return fail(CRH_ERR_INVALID, "unknown option");
	}
}

/* Everything of crh_scene_upload that does not need the scene: the per-wave buffers of a full-size dispatch and the code objects of both
 * feature variants of the kernel the current options select. A host calls it while it is still flattening its scene (renderer_hip.c). */
int crh_context_prepare(crh_ctx *c) {
	if (!c) return fail(CRH_ERR_INVALID, "crh_context_prepare: ctx is NULL");
	int rc = setDevice(c);
	if (rc) return rc;
	/* the plain variant only: scenes with node programs or volumes are rare (crh_scene_upload loads theirs), and a launch of that variant —
	 * twice the scratch per lane — would make the runtime size the queue's scratch memory for it */
	const bool had = c->hasPrograms;
	c->hasPrograms = getenv("CRH_FORCE_PROGRAMS") != nullptr;
	rc = preloadKernel(c);
	c->hasPrograms = had;
	return rc;
}

static std::atomic<int> g_sceneCompiles{0}, g_sceneUploads{0};          /* layout compiles / uploads of this process (tests: crh_debug_upload_counts) */

/* The copies of an upload, with the compiled layout in hand: `cs` is only read (several contexts may upload the same compiled scene at once — renderer_hip.c's
 * multi-GPU frame compiles ONCE, round 5); takeTexels() hands over texels that are already on the device (crh_scene_upload copies them beside its compile) or null. */
static int uploadCompiled(crh_ctx *c, const CompiledScene &cs, const int32_t *prims, size_t primCount, const std::function<int(void **)> &takeTexels, double *tCopiesMs,
                          const std::chrono::steady_clock::time_point &tUp0) {
	auto upMs = [&]() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - tUp0).count(); };
	int rc = CRH_OK;
	HIP_TRY(hipStreamSynchronize(c->stream));
	freeScene(c);
	DScene d;
	memset(&d, 0, sizeof(d));
#define UP(field, ptr, count) do { rc = upload(c, ptr, count, &d.field); if (rc) { freeScene(c); return rc; } } while (0)
	{	/* BVH nodes and prepared triangles share ONE allocation — nodes, then (256-byte aligned) triangles, then one record of padding: a fused walk step
		 * (pathtrace_roll.h) addresses a lane's child pair OR its next two triangles as (S.nodes, a kernel argument in SGPRs) + one 32-bit byte offset, and reads
		 * the two triangles as six consecutive quarters (the second one unused when the leaf range holds one: at the array's end that is the padding) */
		const size_t nodeBytes = sceneNodeBytes(cs), triBytes = cs.tris.size() * sizeof(f4);
		/* (CRH_OPT_WALK = CRH_WALK_WIDE4: the wide nodes stand behind the padding, 128-byte aligned — scene_compile.h: sceneWideOffset, which their references were made for) */
		const size_t wideOff = sceneWideOffset(cs), wideBytes = cs.wide.size() * sizeof(f4);
		const size_t total = wideBytes ? wideOff + wideBytes : nodeBytes + triBytes + 96u;
		if (total >= (1ull << 32)) { freeScene(c); return fail(CRH_ERR_UNSUPPORTED, "crh_scene_upload: BVH nodes + prepared triangles of 4 GB and more"); }
		void *p = nullptr;
		HIP_TRY(hipMalloc(&p, total));
		c->sceneAllocs.push_back(p);
		HIP_TRY(hipMemset((char *)p + nodeBytes + triBytes, 0, (wideBytes ? wideOff : total) - nodeBytes - triBytes));
		if (cs.nodes.size()) HIP_TRY(hipMemcpy(p, cs.nodes.data(), cs.nodes.size() * sizeof(f4), hipMemcpyHostToDevice));
		if (triBytes) HIP_TRY(hipMemcpy((char *)p + nodeBytes, cs.tris.data(), triBytes, hipMemcpyHostToDevice));
		if (wideBytes) HIP_TRY(hipMemcpy((char *)p + wideOff, cs.wide.data(), wideBytes, hipMemcpyHostToDevice));
		d.nodes = (const f4 *)p;
		d.tris = (const f4 *)((const char *)p + nodeBytes);
	}
	UP(shade, cs.shade.data(), cs.shade.size());
	UP(prims, prims, primCount);
	UP(instances, cs.instances.data(), cs.instances.size());
	UP(materials, cs.materials.data(), cs.materials.size());
	UP(bsdfs, cs.bsdfs.data(), cs.bsdfs.size());
	UP(consts, cs.consts.data(), cs.consts.size());
	UP(images, cs.images.data(), cs.images.size());
	UP(prog, cs.prog.data(), cs.prog.size());
	UP(textures, cs.textures.data(), cs.textures.size());
	{
		void *texDev = nullptr;
		rc = takeTexels ? takeTexels(&texDev) : CRH_OK;
		if (rc) { freeScene(c); return rc; }
		if (texDev) { c->sceneAllocs.push_back(texDev); d.texels = (const f4 *)texDev; }
		else UP(texels, cs.texels.data(), cs.texels.size());
	}
#undef UP
	d.tlas_first = cs.tlas_first;
	d.material_count = (uint32_t)cs.materials.size(); d.bsdf_count = (uint32_t)cs.bsdfs.size(); d.const_count = (uint32_t)cs.consts.size();
	d.image_count = (uint32_t)cs.images.size(); d.texture_count = (uint32_t)cs.textures.size();
	d.tlas_root = cs.tlas_root; d.tlas_node_count = cs.tlas_node_count; d.tlas_prim_base = cs.tlas_prim_base; d.shade_classes = cs.shade_classes; d.instance_count = (uint32_t)cs.instances.size();
	d.background = cs.background;
	{ const crh_camera *cam = nullptr; rc = upload(c, &cs.camera, 1, &cam); if (rc) { freeScene(c); return rc; } d.camera = cam; }
	c->d = d;
	c->haveWide = !cs.wide.empty(); c->wideTlasRoot = cs.wide_tlas_root;
	if (cs.want_wide && !c->haveWide && !cs.wide_refused.empty())          /* (not only under CRH_TRACE_UPLOAD — the frame would silently be the binary walk's) */
		fprintf(stderr, "libcray_hip: CRH_WALK_WIDE4 was asked for, but this scene has no 4-ary copy (%s): the binary walk renders it\n", cs.wide_refused.c_str());
	c->hasPrograms = cs.prog.size() > 1 || cs.has_volumes || getenv("CRH_FORCE_PROGRAMS") != nullptr;    /* the rare-features kernel variant */
	c->hasVolumes = cs.has_volumes;
	c->haveScene = true;
	/* The scene is resident, and the device has nothing left to do, when this function returns: a blocking device-to-host copy on the NULL stream ends the
	 * set-up. Measured in round 3 (CRH_TRACE_SYNC): without it the first dispatch's kernel starts 7-25 ms after its launch — behind work the runtime
	 * still owes the pageable host-to-device copies above, which neither hipStreamSynchronize on the context's (non-blocking) stream nor
	 * hipDeviceSynchronize waits for; with it, 1-5 us. (Until round 3 the watchdog flag's copy in crh_synchronize was this barrier by accident.) */
	*tCopiesMs = upMs();
	rc = preloadKernel(c, true);         /* ... and an (empty) launch of the kernel on the context's stream is waited for: see below */
	if (rc != CRH_OK) return rc;
	unsigned int flag = 0;
	HIP_TRY(hipMemcpy(&flag, c->dWork + (CRH_WORK_SLOTS - 1), sizeof(flag), hipMemcpyDeviceToHost));
	return CRH_OK;
}

/* (every C++ exception a compile or an upload can raise ends at the C boundary as an error code: ADVICE r04) */
static int guarded(const char *what, const std::function<int()> &f) {
	try { return f(); }
	catch (const std::bad_alloc &) { return fail(CRH_ERR_NOMEM, std::string(what) + ": out of host memory"); }
	catch (const std::exception &e) { return fail(CRH_ERR_HIP, std::string(what) + ": " + e.what()); }
}

int crh_scene_upload(crh_ctx *c, const crh_scene_desc *scene) {
	if (!c || !scene) return fail(CRH_ERR_INVALID, "crh_scene_upload: NULL argument");
	int rc = setDevice(c);
	if (rc) return rc;
	return guarded("crh_scene_upload", [&]() -> int {
	std::unique_ptr<CompiledScene> compiled(new CompiledScene);
	CompiledScene &cs = *compiled;
	std::string err;
	const auto tUp0 = std::chrono::steady_clock::now();
	auto upMs = [&]() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - tUp0).count(); };
	/* the texels — nine tenths of a textured scene's bytes, and the first thing the compiler finishes — are copied by a helper thread WHILE the BVHs and triangles are
	 * prepared (round 4: traced in the drop-in, the pageable copies' DMA was still under way 21 ms after hipMemcpy had returned, in front of the first dispatch) */
	struct TexelJob {                   /* joined, and its memory released unless the scene took it, on every way out of this function */
		std::thread thread;
		void *dev = nullptr;
		hipError_t status = hipSuccess;
		~TexelJob() { if (thread.joinable()) thread.join(); if (dev) (void)hipFree(dev); }
	} texelJob;
	cs.want_wide = c->walk == CRH_WALK_WIDE4;
	int rc = compile_scene(scene, cs, err, [&]() {
		try {
			texelJob.thread = std::thread([&]() {
				texelJob.status = hipSetDevice(c->device);
				const size_t bytes = std::max<size_t>(cs.texels.size(), 1) * sizeof(f4);
				if (texelJob.status == hipSuccess) texelJob.status = hipMalloc(&texelJob.dev, bytes);
				if (texelJob.status == hipSuccess) texelJob.status = hipMemcpy(texelJob.dev, cs.texels.data(), cs.texels.size() * sizeof(f4), hipMemcpyHostToDevice);
			});
		} catch (const std::exception &) { /* no helper thread: the texels are copied with the rest */ }
	});
	if (rc != CRH_OK) return fail(rc, "crh_scene_upload: " + err);
	++g_sceneCompiles;
	const double tCompile = upMs();
	double tCopies = 0.0;
	rc = uploadCompiled(c, cs, scene->prim_indices, (size_t)scene->prim_index_count, [&](void **dev) -> int {
		if (!texelJob.thread.joinable()) return CRH_OK;
		texelJob.thread.join();
		if (texelJob.status != hipSuccess) return fail(CRH_ERR_HIP, std::string("crh_scene_upload: texels: ") + hipGetErrorString(texelJob.status));
		*dev = texelJob.dev;
		texelJob.dev = nullptr;
		return CRH_OK;
	}, &tCopies, tUp0);
	if (rc != CRH_OK) return rc;
	++g_sceneUploads;
	if (getenv("CRH_TRACE_UPLOAD"))         /* dev: where crh_scene_upload's time goes */
		fprintf(stderr, "crh_scene_upload trace: layout compile %.1f ms, allocations + copies %.1f ms (%.1f MB), code object + barrier %.1f ms\n", tCompile, tCopies - tCompile,
				(double)(cs.nodes.size() * 16 + cs.tris.size() * 16 + cs.shade.size() * sizeof(DShadeTri) + cs.texels.size() * 16) / 1e6, upMs() - tCopies);
	releaseJanitor(c, true);
	CompiledScene *const trash = compiled.release();
	c->janitorGo = false;
	c->janitorWaiting.store(true);
	try {
		c->janitor = std::thread([c, trash]() {
			{ std::unique_lock<std::mutex> lk(c->janitorMu); c->janitorCv.wait(lk, [c]() { return c->janitorGo; }); }
			const auto t0 = std::chrono::steady_clock::now();
			delete trash;
			if (getenv("CRH_TRACE_UPLOAD"))
				fprintf(stderr, "crh_scene_upload trace: the compiled host arrays were released in %.1f ms (behind a dispatch, or at the next upload / the context's end)\n",
						std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count());
		});
	} catch (const std::system_error &) {          /* no thread to be had: the scene IS resident — release the host arrays here and now, and say OK */
		c->janitorWaiting.store(false);
		delete trash;
	}
	return CRH_OK;
	});
}

/* ONE compile for the contexts of an in-process multi-GPU frame (round 5; the reference builds its scene once and every worker reads it:
 * src/datatypes/scene.c:111-213): crh_scene_compile derives the device layout on the host — no context, no device —, crh_scene_upload_compiled copies it to a
 * context's GPU (only reads the handle: the GPU threads of renderer_hip.c call it at the same time), crh_compiled_scene_free releases it. */
struct crh_compiled_scene {
	CompiledScene cs;
	std::vector<int32_t> prims;          /* crh_scene_desc.prim_indices: the caller's arrays are not retained */
};

int crh_scene_compile(const crh_scene_desc *scene, int walk, crh_compiled_scene **out) {
	if (!scene || !out) return fail(CRH_ERR_INVALID, "crh_scene_compile: NULL argument");
	*out = nullptr;          /* (before anything can fail: a caller that frees the handle on error must not free garbage) */
	if (walk != CRH_WALK_BINARY && walk != CRH_WALK_WIDE4) return fail(CRH_ERR_INVALID, "crh_scene_compile: walk must be CRH_WALK_BINARY or CRH_WALK_WIDE4");
	return guarded("crh_scene_compile", [&]() -> int {
		const auto t0 = std::chrono::steady_clock::now();
		std::unique_ptr<crh_compiled_scene> h(new crh_compiled_scene);
		h->cs.want_wide = walk == CRH_WALK_WIDE4;
		std::string err;
		const int rc = compile_scene(scene, h->cs, err);
		if (rc != CRH_OK) return fail(rc, "crh_scene_compile: " + err);
		h->prims.assign(scene->prim_indices, scene->prim_indices + scene->prim_index_count);
		++g_sceneCompiles;
		if (getenv("CRH_TRACE_UPLOAD")) fprintf(stderr, "crh_scene_compile trace: layout compile %.1f ms (one compile for every context that uploads it)\n",
		                                         std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count());
		*out = h.release();
		return CRH_OK;
	});
}

int crh_scene_upload_compiled(crh_ctx *c, const crh_compiled_scene *h) {
	if (!c || !h) return fail(CRH_ERR_INVALID, "crh_scene_upload_compiled: NULL argument");
	int rc = setDevice(c);
	if (rc) return rc;
	if ((c->walk == CRH_WALK_WIDE4) != h->cs.want_wide) return fail(CRH_ERR_INVALID, "crh_scene_upload_compiled: the scene was compiled for another CRH_OPT_WALK than the context's");
	return guarded("crh_scene_upload_compiled", [&]() -> int {
		const auto t0 = std::chrono::steady_clock::now();
		double tCopies = 0.0;
		releaseJanitor(c, true);
		const int rc2 = uploadCompiled(c, h->cs, h->prims.data(), h->prims.size(), nullptr, &tCopies, t0);
		if (rc2 != CRH_OK) return rc2;
		++g_sceneUploads;
		if (getenv("CRH_TRACE_UPLOAD")) fprintf(stderr, "crh_scene_upload_compiled trace: device %d: allocations + copies %.1f ms, code object + barrier %.1f ms\n", c->device, tCopies,
		                                         std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count() - tCopies);
		return CRH_OK;
	});
}

void crh_compiled_scene_free(crh_compiled_scene *h) { delete h; }

/* tests: scene compiles / compiled-scene uploads of this process so far */
void crh_debug_upload_counts(int *compiles, int *uploads) { if (compiles) *compiles = g_sceneCompiles.load(); if (uploads) *uploads = g_sceneUploads.load(); }

int crh_framebuffer_alloc(crh_ctx *c, int width, int height, float **dev_out) {
	if (!c || !dev_out || width <= 0 || height <= 0) return fail(CRH_ERR_INVALID, "crh_framebuffer_alloc: bad argument");
	int rc = setDevice(c);
	if (rc) return rc;
	void *p = nullptr;
	const size_t bytes = (size_t)width * height * 3 * sizeof(float);
	HIP_TRY(hipMalloc(&p, bytes));
	HIP_TRY(hipMemsetAsync(p, 0, bytes, c->stream));
	*dev_out = (float *)p;
	return CRH_OK;
}

int crh_framebuffer_free(crh_ctx *c, float *dev_fb) {
	if (!c) return fail(CRH_ERR_INVALID, "crh_framebuffer_free: ctx is NULL");
	int rc = setDevice(c);
	if (rc) return rc;
	HIP_TRY(hipStreamSynchronize(c->stream));
	if (dev_fb) HIP_TRY(hipFree(dev_fb));
	return CRH_OK;
}

int crh_framebuffer_clear(crh_ctx *c, float *dev_fb, int width, int height) {
	if (!c || !dev_fb || width <= 0 || height <= 0) return fail(CRH_ERR_INVALID, "crh_framebuffer_clear: bad argument");
	int rc = setDevice(c);
	if (rc) return rc;
	HIP_TRY(hipMemsetAsync(dev_fb, 0, (size_t)width * height * 3 * sizeof(float), c->stream));
	return CRH_OK;
}

int crh_framebuffer_download(crh_ctx *c, const float *dev_fb, int width, int height, float *host_rgb) {
	if (!c || !dev_fb || !host_rgb || width <= 0 || height <= 0) return fail(CRH_ERR_INVALID, "crh_framebuffer_download: bad argument");
	int rc = setDevice(c);
	if (rc) return rc;
	HIP_TRY(hipMemcpyAsync(host_rgb, dev_fb, (size_t)width * height * 3 * sizeof(float), hipMemcpyDeviceToHost, c->stream));
	HIP_TRY(hipStreamSynchronize(c->stream));
	return checkWatchdog(c);
}

int crh_framebuffer_to_srgb8(crh_ctx *c, const float *dev_fb, int width, int height, uint8_t *host_rgb8) {
	if (!c || !dev_fb || !host_rgb8 || width <= 0 || height <= 0) return fail(CRH_ERR_INVALID, "crh_framebuffer_to_srgb8: bad argument");
	int rc = setDevice(c);
	if (rc) return rc;
	const size_t n = (size_t)width * height * 3;
	if (n > c->srgbBytes) {               /* kept by the context: an interactive host converts after every pass chunk */
		if (c->dSrgb) HIP_TRY(hipFree(c->dSrgb));
		c->dSrgb = nullptr; c->srgbBytes = 0;
		HIP_TRY(hipMalloc((void **)&c->dSrgb, n));
		c->srgbBytes = n;
	}
	const int grid = (int)std::min<size_t>((n + 255) / 256, 4096);
	hipLaunchKernelGGL(k_to_srgb8, dim3(grid), dim3(256), 0, c->stream, dev_fb, n, c->dSrgb);
	hipError_t e = hipGetLastError();
	if (e == hipSuccess) e = hipMemcpyAsync(host_rgb8, c->dSrgb, n, hipMemcpyDeviceToHost, c->stream);
	if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
	if (e != hipSuccess) return fail(CRH_ERR_HIP, std::string("crh_framebuffer_to_srgb8: ") + hipGetErrorString(e));
	return CRH_OK;
}

/* The same conversion for a GPU that owns every n-th strip of the frame (host/share.h): the whole frame is converted on the device (a few microseconds), but only the
 * rows of strips g, g + n, g + 2 n, ... travel — one strided copy straight into the caller's frame-sized buffer, whose other rows are left alone. (Until round 4 a
 * multi-GPU host downloaded the whole 8-bit frame from every GPU after every dispatch to keep 1 / n of it.) */
int crh_framebuffer_strips_to_srgb8(crh_ctx *c, const float *dev_fb, int width, int height, int strip_rows, int g, int n_gpus, uint8_t *host_rgb8) {
	if (!c || !dev_fb || !host_rgb8 || width <= 0 || height <= 0 || strip_rows <= 0 || n_gpus <= 0 || g < 0 || g >= n_gpus) return fail(CRH_ERR_INVALID, "crh_framebuffer_strips_to_srgb8: bad argument");
	if (n_gpus == 1) return crh_framebuffer_to_srgb8(c, dev_fb, width, height, host_rgb8);
	int rc = setDevice(c);
	if (rc) return rc;
	const size_t n = (size_t)width * height * 3, rowBytes = (size_t)width * 3;
	if (n > c->srgbBytes) {
		if (c->dSrgb) HIP_TRY(hipFree(c->dSrgb));
		c->dSrgb = nullptr; c->srgbBytes = 0;
		HIP_TRY(hipMalloc((void **)&c->dSrgb, n));
		c->srgbBytes = n;
	}
	const int grid = (int)std::min<size_t>((n + 255) / 256, 4096);
	hipLaunchKernelGGL(k_to_srgb8, dim3(grid), dim3(256), 0, c->stream, dev_fb, n, c->dSrgb);
	hipError_t e = hipGetLastError();
	/* strip k of this GPU covers image rows [(k n + g) R, ... + R) = framebuffer rows [H - y1, H - y0) (texture.c:24-28: row H - 1 - y): the full strips are
	 * R-row blocks a constant n R rows apart — one 2-D copy, from the block nearest the top of the buffer (the last full strip) downwards */
	const int R = strip_rows, period = n_gpus * R;
	int fullStrips = 0, raggedY0 = -1;
	for (int y0 = g * R; y0 < height; y0 += period) { if (y0 + R <= height) ++fullStrips; else raggedY0 = y0; }
	if (e == hipSuccess && fullStrips > 0) {
		const size_t base = (size_t)(height - (g * R + (fullStrips - 1) * period) - R) * rowBytes;
		e = hipMemcpy2DAsync(host_rgb8 + base, (size_t)period * rowBytes, c->dSrgb + base, (size_t)period * rowBytes, (size_t)R * rowBytes, (size_t)fullStrips, hipMemcpyDeviceToHost, c->stream);
	}
	if (e == hipSuccess && raggedY0 >= 0) e = hipMemcpyAsync(host_rgb8, c->dSrgb, (size_t)(height - raggedY0) * rowBytes, hipMemcpyDeviceToHost, c->stream);      /* the frame's last, shorter strip: the buffer's first rows */
	if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
	if (e != hipSuccess) return fail(CRH_ERR_HIP, std::string("crh_framebuffer_strips_to_srgb8: ") + hipGetErrorString(e));
	return CRH_OK;
}

/* ---- work plan of one dispatch (host only, no device needed: crh_debug_plan_units runs it for the CPU tests) ------------------------- */
struct PlanKnobs { int unitItems, unitsPerWave, tailPercent, tail2Percent, passChunk, cuCount, blocksPerCU; bool wg; int tailSplit; /* 0 unless the rolling kernel runs the plan */ };
struct WorkPlan {
	std::vector<crh_tile> work;            /* the caller's tiles, the tail ones split by rows */
	std::vector<uint32_t> start;           /* start[t] = first unit of work[t]; start[work.size()] = total */
	uint64_t total = 0;
	int bw = 1, bh = 1, sbw = 1, sbh = 1, tbw = 1, tbh = 1;      /* block shapes: regular, from firstSmall on, from firstTiny on */
	uint32_t firstSmall = 0, firstTiny = 0;
	uint32_t firstMicro = 0;               /* BlockQueue: the 64-path units at the very end (rolling kernel) */
	int mbw = 1, mbh = 1, segs = 1, segPasses = 0;
	int area = 1, chunk = 1;
	uint32_t grid = 0;
};
#define CRH_MICRO_ITEMS 64
static int planWork(const crh_render_params *P, const crh_tile *tiles, uint32_t tile_count, const PlanKnobs &K, WorkPlan &W, std::string &err) {
	/* Block shape: one work unit (a block for all passes of the dispatch) should hold about unitItems paths, so
	 * that every wave gets many units (load balance) whatever the sample count: 16x16 pixels at 4 spp ... 2x2 at
	 * 256 spp, 1x1 beyond. Smaller blocks also keep the 64 lanes of a wave on fewer pixels (coherent walks). */
	uint64_t pixels = 0;
	for (uint32_t t = 0; t < tile_count; ++t) pixels += (uint64_t)std::max(0, tiles[t].x1 - tiles[t].x0) * std::max(0, tiles[t].y1 - tiles[t].y0);
	/* the workgroup kernel's unit is worked on by four waves: four times the paths, a quarter of the consumers */
	const int unitItems = K.wg ? K.unitItems * 4 : K.unitItems;
	const uint64_t wavesMax = (uint64_t)K.cuCount * K.blocksPerCU * (K.wg ? 1 : CRH_BLOCK / 64);
	int area = 1;
	while (area < 256 && (int64_t)area * P->pass_count < unitItems) area *= 2;
	while (area > 1 && pixels / area < (uint64_t)K.unitsPerWave * wavesMax) area /= 2;       /* few pixels (or few passes): keep every wave fed */
	int capW = 1, capH = 1;                   /* no block wider / taller than the widest / tallest tile (power of two below it): a 16x16 block clipped
	                                           * to a 4-row strip would generate three quarters of its items as padding */
	for (uint32_t t = 0; t < tile_count; ++t) {
		const crh_tile &r = tiles[t];
		if (r.x0 < 0 || r.y0 < 0 || r.x1 > P->image_width || r.y1 > P->image_height || r.x0 > r.x1 || r.y0 > r.y1)
			{ err = "crh_render_tiles: tile outside the image"; return CRH_ERR_INVALID; }
		while (capW * 2 <= r.x1 - r.x0) capW *= 2;
		while (capH * 2 <= r.y1 - r.y0) capH *= 2;
	}
	auto shapeOf = [capW, capH](int a, int &w, int &h) {
		w = 1; h = 1;
		while (w * h < a) {
			const bool canW = w * 2 <= capW, canH = h * 2 <= capH;
			if (!canW && !canH) break;
			if ((w <= h && canW) || !canH) w *= 2; else h *= 2;
		}
	};
	int &bw = W.bw, &bh = W.bh;
	bw = 1; bh = 1;
	shapeOf(area, bw, bh);
	/* Tapered units: the work queue is consumed in list order, so the tail of the list decides how far apart the waves
	 * finish. The last tailPercent of the pixels (whole tiles from the end of the list, the boundary tile split by rows)
	 * are cut into blocks of a quarter of the area. */
	std::vector<crh_tile> &work = W.work;
	work.assign(tiles, tiles + tile_count);
	/* index of the first tile of the tail that holds the last `want` pixels (the boundary tile is split by rows) */
	int64_t insertedAt = -1;                /* position of the tile the last cutTail() inserted (-1: it split none) */
	auto cutTail = [&work, &insertedAt](uint64_t want) -> uint32_t {
		uint64_t got = 0;
		uint32_t t = (uint32_t)work.size();
		insertedAt = -1;
		while (t > 0 && got < want) {
			const crh_tile r = work[t - 1];
			const uint64_t a = (uint64_t)(r.x1 - r.x0) * (uint64_t)(r.y1 - r.y0);
			if (got + a <= want + want / 4 || r.y1 - r.y0 < 2) { got += a; --t; continue; }
			/* split this tile by rows: the upper part (in list order: first) keeps the bigger blocks */
			const int w = r.x1 - r.x0;
			int rowsSmall = (int)((want - got + (uint64_t)w - 1) / (uint64_t)(w ? w : 1));
			rowsSmall = std::min(std::max(rowsSmall, 1), r.y1 - r.y0 - 1);
			const int ySplit = r.y1 - rowsSmall;                 /* rows are independent: which part goes first is free */
			work[t - 1] = crh_tile{r.x0, r.y0, r.x1, ySplit};
			work.insert(work.begin() + t, crh_tile{r.x0, ySplit, r.x1, r.y1});
			insertedAt = (int64_t)t;
			break;
		}
		return t;
	};
	uint32_t &firstSmall = W.firstSmall, &firstTiny = W.firstTiny;
	firstSmall = firstTiny = (uint32_t)work.size();
	int &sbw = W.sbw, &sbh = W.sbh, &tbw = W.tbw, &tbh = W.tbh;
	sbw = bw; sbh = bh; tbw = bw; tbh = bh;
	if (area >= 2 && K.tailPercent > 0) {
		/* a quarter / a sixteenth of the block — but never fewer than 512 / 256 paths per unit (few passes per dispatch): a unit that cannot
		 * even fill the wave's path table is all ramp and drain */
		auto pixelsFor = [&](int paths) { int a = 1; while ((int64_t)a * P->pass_count < paths && a < area) a *= 2; return a; };
		const int smallArea = std::max(std::max(area / 4, 1), std::min(pixelsFor(512), area)), tinyArea = std::max(std::max(area / 16, 1), std::min(pixelsFor(256), area));
		shapeOf(smallArea, sbw, sbh);
		firstSmall = cutTail(pixels * (uint64_t)K.tailPercent / 100);
		firstTiny = (uint32_t)work.size();
		tbw = sbw; tbh = sbh;
		if (tinyArea < smallArea && K.tail2Percent > 0 && K.tail2Percent < K.tailPercent) {
			shapeOf(tinyArea, tbw, tbh);
			const uint32_t t2 = cutTail(pixels * (uint64_t)K.tail2Percent / 100);
			/* with its + 25 % tolerance the second cut can reach a tile in front of the first one's (tail2Percent close to tailPercent): the
			 * tile it inserts then shifts the quarter-size tail by one */
			if (insertedAt >= 0 && (uint32_t)insertedAt <= firstSmall) ++firstSmall;
			firstTiny = std::max(firstSmall, t2);
		}
	}
	/* The very end in 64-path units (rolling kernel): tailSplit of them per wave, at most a quarter of the dispatch. The late finishers of a dispatch are waves
	 * that pulled an expensive unit just before the queue ran dry (profiles/r03x_probe_tail_hist.log); with four jobs open a wave needs only 64 paths per job to
	 * keep its table full, so the last units can be that short — and where a single pixel holds more than that (>= 128 passes), its passes are split. */
	uint32_t &firstMicro = W.firstMicro;
	firstMicro = (uint32_t)work.size();
	W.mbw = tbw; W.mbh = tbh; W.segs = 1; W.segPasses = P->pass_count;
	if (K.tailSplit > 0 && P->pass_count > 0 && pixels > 0) {
		int microArea = 1;
		while ((int64_t)microArea * 2 * P->pass_count <= CRH_MICRO_ITEMS && microArea < area) microArea *= 2;
		const int segs = P->pass_count >= 2 * CRH_MICRO_ITEMS ? (P->pass_count + CRH_MICRO_ITEMS - 1) / CRH_MICRO_ITEMS : 1;
		const int tinyAreaNow = tbw * tbh;
		if (segs > 1 || microArea < tinyAreaNow) {
			const uint64_t unitsWanted = (uint64_t)K.tailSplit * wavesMax;
			uint64_t want = std::min<uint64_t>(pixels / 4, (unitsWanted * (uint64_t)microArea + (uint64_t)segs - 1) / (uint64_t)segs);
			if (want > 0) {
				shapeOf(microArea, W.mbw, W.mbh);
				W.segs = segs; W.segPasses = segs > 1 ? CRH_MICRO_ITEMS : P->pass_count;
				const uint32_t t3 = cutTail(want);
				if (insertedAt >= 0 && (uint32_t)insertedAt <= firstSmall) ++firstSmall;
				if (insertedAt >= 0 && (uint32_t)insertedAt <= firstTiny) ++firstTiny;
				firstMicro = t3;                                   /* the levels stay in order; a dispatch without a taper (single-pixel blocks) still ends in split pixels */
				firstTiny = std::min(firstTiny, firstMicro);
				firstSmall = std::min(firstSmall, firstTiny);
			}
		}
	}
	const uint32_t work_count = (uint32_t)work.size();
	std::vector<uint32_t> &start = W.start;
	start.assign(work_count + 1, 0);
	uint64_t &total = W.total;
	total = 0;
	for (uint32_t t = 0; t < work_count; ++t) {
		const crh_tile &r = work[t];
		const int ubw = t >= firstMicro ? W.mbw : t >= firstTiny ? tbw : t >= firstSmall ? sbw : bw, ubh = t >= firstMicro ? W.mbh : t >= firstTiny ? tbh : t >= firstSmall ? sbh : bh;
		start[t] = (uint32_t)total;
		total += (uint64_t)((r.x1 - r.x0 + ubw - 1) / ubw) * ((r.y1 - r.y0 + ubh - 1) / ubh) * (uint64_t)(t >= firstMicro ? W.segs : 1);
		if (total > 0xFFFFFFF0ull) { err = "crh_render_tiles: more than 2^32 pixel blocks in one dispatch"; return CRH_ERR_UNSUPPORTED; }
	}
	start[work_count] = (uint32_t)total;
	W.grid = 0; W.chunk = 1; W.area = area;
	if (total == 0 || P->pass_count == 0) return CRH_OK;
	W.grid = (uint32_t)std::min<uint64_t>((uint64_t)K.cuCount * K.blocksPerCU, K.wg ? total : (total + 3) / 4);
	/* passes per chunk: a chunk (block x passes) should also hold about unitItems paths, so that each lane runs >= 16
	 * paths between two wave-wide folds */
	W.chunk = std::min(P->pass_count, std::max(K.passChunk, (unitItems + bw * bh - 1) / (bw * bh)));      /* bw x bh < area when the tiles are thinner than the block (strips) */
	return CRH_OK;
}

/* The work units crh_render_tiles would hand to the kernel for this dispatch on a GPU with `cu_count` compute units at the default options:
 * one record of eight ints per unit, in hand-out order — pixel rectangle x0, y0, x1, y1 (clipped to its tile), block area in pixels, the
 * taper level (0 regular, 1 quarter blocks, 2 sixteenth blocks, 3 the 64-path units of the very end), and the unit's passes (first, count: all
 * of the dispatch's except for the pass segments of split pixels). Needs no device: the CPU tests check cover, order and unit sizes. */
int crh_debug_plan_units(const crh_render_params *P, const crh_tile *tiles, uint32_t tile_count, uint32_t cu_count, int32_t *units_out, uint64_t max_units,
						 uint64_t *unit_count_out, int32_t *pass_chunk_out) {
	if (!P || (!tiles && tile_count) || !unit_count_out || cu_count < 1) return fail(CRH_ERR_INVALID, "crh_debug_plan_units: bad argument");
	if (P->image_width <= 0 || P->image_height <= 0 || P->pass_count < 0) return fail(CRH_ERR_INVALID, "crh_debug_plan_units: bad render parameters");
	crh_ctx defaults{};
	if (const char *env = getenv("CRH_TAIL_SPLIT")) { if (atoi(env) >= 0 && atoi(env) <= 64) defaults.tailSplit = atoi(env); }       /* like crh_context_create */
	const PlanKnobs knobs{defaults.unitItems, defaults.unitsPerWave, defaults.tailPercent, defaults.tail2Percent, defaults.passChunk, (int)cu_count, defaults.blocksPerCU, false,
	                      defaults.kernel == CRH_KERNEL_ROLL ? defaults.tailSplit : 0};
	WorkPlan W;
	std::string err;
	const int rc = planWork(P, tiles, tile_count, knobs, W, err);
	if (rc != CRH_OK) return fail(rc, err);
	*unit_count_out = W.total;
	if (pass_chunk_out) *pass_chunk_out = W.chunk;
	uint64_t u = 0;
	for (uint32_t t = 0; t < W.work.size() && units_out; ++t) {          /* the kernel's unit -> block arithmetic (k_pathtrace: "pull a work unit") */
		const crh_tile &r = W.work[t];
		const int level = t >= W.firstMicro ? 3 : t >= W.firstTiny ? 2 : t >= W.firstSmall ? 1 : 0;
		const int ubw = level == 3 ? W.mbw : level == 2 ? W.tbw : level == 1 ? W.sbw : W.bw, ubh = level == 3 ? W.mbh : level == 2 ? W.tbh : level == 1 ? W.sbh : W.bh;
		const uint32_t nbx = (uint32_t)(r.x1 - r.x0 + ubw - 1) / (uint32_t)ubw;
		const uint32_t segs = level == 3 ? (uint32_t)W.segs : 1u;
		for (uint32_t local = 0; local < W.start[t + 1] - W.start[t]; ++local, ++u) {
			if (u >= max_units) continue;
			const uint32_t blk = local / segs, seg = local % segs;          /* k_pathtrace_roll: ST_OPEN */
			const int x0 = r.x0 + (int)(blk % nbx) * ubw, y0 = r.y0 + (int)(blk / nbx) * ubh;
			int32_t *o = units_out + 8 * u;
			o[0] = x0; o[1] = y0; o[2] = std::min(x0 + ubw, r.x1); o[3] = std::min(y0 + ubh, r.y1); o[4] = ubw * ubh; o[5] = level;
			o[6] = P->first_pass + (segs > 1 ? (int)seg * W.segPasses : 0);
			o[7] = segs > 1 ? std::min(W.segPasses, P->first_pass + P->pass_count - o[6]) : P->pass_count;
		}
	}
	return CRH_OK;
}

/* ---- the streaming form (CRH_KERNEL_STREAM; csrc/pathtrace_stream.h) ------------------------------------------------------------------------------------------------ */
/* can this dispatch be streamed? (the forms it cannot serve are rendered by the rolling kernel: a sampler draw inside the walk (volumes), the Halton sampler of the
 * interactive mode, the 4-ary walk; bounces <= 0 never reaches the path tracer) */
static bool streamServes(const crh_ctx *c, const crh_render_params *P) {
	return c->kernel == CRH_KERNEL_STREAM && !c->hasVolumes && c->sampler == CRH_SAMPLER_RANDOM && !(c->walk == CRH_WALK_WIDE4 && c->haveWide) && P->bounces > 0;
}
static int growDevice(crh_ctx *c, void **p, size_t *have, size_t need, size_t elemBytes) {
	if (need <= *have) return CRH_OK;
	HIP_TRY(hipStreamSynchronize(c->stream));
	if (*p) HIP_TRY(hipFree(*p));
	*p = nullptr; *have = 0;
	HIP_TRY(hipMalloc(p, need * elemBytes));
	*have = need;
	return CRH_OK;
}
/* walk-kernel instantiations: waves per SIMD the register allocator leaves room for / traversal-stack entries in LDS. One-instance scenes (the triangle soups) walk ONE deep
 * BVH and want the deeper LDS stack at five workgroups per CU; everything else the shallower one at six (profiles/r06c_probe_walk.log) */
#define CRH_STREAM_WALK_A_WPS 6
#define CRH_STREAM_WALK_A_NLDS 7
#define CRH_STREAM_WALK_B_WPS 5
#define CRH_STREAM_WALK_B_NLDS 12
static int renderStream(crh_ctx *c, const crh_render_params *P, const crh_tile *tiles, uint32_t tile_count, float *dev_fb) {
	/* the dispatch's pixels in list order */
	std::vector<crh_tile> work;
	std::vector<uint32_t> start;
	uint64_t npix64 = 0;
	for (uint32_t i = 0; i < tile_count; ++i) {
		crh_tile t = tiles[i];
		if (t.x0 < 0 || t.y0 < 0 || t.x1 > P->image_width || t.y1 > P->image_height || t.x0 > t.x1 || t.y0 > t.y1) return fail(CRH_ERR_INVALID, "crh_render_tiles: tile outside the image");          /* (planWork's rule) */
		if (t.x1 <= t.x0 || t.y1 <= t.y0) continue;
		start.push_back((uint32_t)npix64);
		work.push_back(t);
		npix64 += (uint64_t)(t.x1 - t.x0) * (uint64_t)(t.y1 - t.y0);
		if (npix64 >= CRH_SF_CHUNK_ITEMS_MAX) return fail(CRH_ERR_UNSUPPORTED, "crh_render_tiles: a streamed dispatch holds fewer than 2^28 pixels");
	}
	if (npix64 == 0 || P->pass_count == 0) return CRH_OK;
	start.push_back((uint32_t)npix64);
	const uint32_t npix = (uint32_t)npix64, ntiles = (uint32_t)work.size();
	const uint64_t totalItems = npix64 * (uint64_t)P->pass_count;
	const uint32_t cohorts = (uint32_t)std::min<uint64_t>((uint64_t)c->streamCohorts, (totalItems + CRH_SF_COHORT - 1) / CRH_SF_COHORT);
	const size_t slots = (size_t)cohorts * CRH_SF_COHORT;
	/* chunks: about one pool's worth of items each (all pixels x a few passes), fewer than 2^28 */
	uint32_t C = (uint32_t)std::min<uint64_t>((uint64_t)P->pass_count, (slots + npix - 1) / npix);
	if (C < 1u) C = 1u;
	while (C > 1u && (uint64_t)npix * C >= CRH_SF_CHUNK_ITEMS_MAX) --C;
	const uint32_t chunkCount = ((uint32_t)P->pass_count + C - 1u) / C, lastPasses = (uint32_t)P->pass_count - (chunkCount - 1u) * C;
	const uint32_t chunkItems = npix * C;
	const size_t slabFloats = (size_t)std::min<uint32_t>(CRH_SF_RING, chunkCount) * chunkItems * 3;

	int rc;
	{
		size_t have = c->streamSlots;
		if (slots > have) {
			HIP_TRY(hipStreamSynchronize(c->stream));
			if (c->dStreamPlanes) HIP_TRY(hipFree(c->dStreamPlanes));
			if (c->dStreamHit) HIP_TRY(hipFree(c->dStreamHit));
			if (c->dStreamHitInst) HIP_TRY(hipFree(c->dStreamHitInst));
			if (c->dStreamCount) HIP_TRY(hipFree(c->dStreamCount));
			c->dStreamPlanes = nullptr; c->dStreamHit = nullptr; c->dStreamHitInst = nullptr; c->dStreamCount = nullptr; c->streamSlots = 0;
			HIP_TRY(hipMalloc((void **)&c->dStreamPlanes, slots * 8 * sizeof(f4)));
			HIP_TRY(hipMalloc((void **)&c->dStreamHit, slots * sizeof(f4)));
			HIP_TRY(hipMalloc((void **)&c->dStreamHitInst, slots * sizeof(int32_t)));
			HIP_TRY(hipMalloc((void **)&c->dStreamCount, (slots / CRH_SF_COHORT) * 2 * sizeof(uint32_t)));
			c->streamSlots = slots;
		}
	}
	rc = growDevice(c, (void **)&c->dStreamSlab, &c->streamSlabFloats, slabFloats, sizeof(float));
	if (rc) return rc;
	const bool deepStack = c->d.instance_count == 1u && c->d.tlas_node_count == 1u;
	const uint32_t walkGrid = (uint32_t)c->cuCount * (uint32_t)(deepStack ? CRH_STREAM_WALK_B_WPS : CRH_STREAM_WALK_A_WPS);
	rc = growDevice(c, (void **)&c->dStreamOvf, &c->streamOvfWords, (size_t)walkGrid * (CRH_BLOCK / 64) * CRH_OVF_WORDS_PER_WAVE, sizeof(uint32_t));
	if (rc) return rc;
	if (!c->dStreamCtl) HIP_TRY(hipMalloc((void **)&c->dStreamCtl, sizeof(StreamCtl)));
	if (!c->hStreamDone) {
		HIP_TRY(hipHostMalloc((void **)&c->hStreamDone, sizeof(unsigned int), hipHostMallocDefault));
		*c->hStreamDone = 0u;
		HIP_TRY(hipHostGetDevicePointer((void **)&c->dStreamDone, c->hStreamDone, 0));
	}
	for (int i = 0; i < 2; ++i) if (!c->streamEv[i]) HIP_TRY(hipEventCreateWithFlags(&c->streamEv[i], hipEventDisableTiming));

	/* the tile list: pinned host slot, copied into the device slot by the dispatch's first kernel (no copy engine in front of a kernel: see crh_render_tiles) */
	const uint32_t slot = c->workSlot % CRH_WORK_SLOTS;
	crh_ctx::TileSlot &ts = c->tileSlots[slot];
	const size_t tileBytes = ntiles * sizeof(crh_tile), startBytes = (ntiles + 1) * sizeof(uint32_t);
	if (ts.inFlight) { HIP_TRY(hipEventSynchronize(ts.done)); ts.inFlight = false; }
	if (tileBytes + startBytes > ts.cap) {
		if (ts.dev) HIP_TRY(hipFree(ts.dev));
		if (ts.host) HIP_TRY(hipHostFree(ts.host));
		ts.dev = ts.host = nullptr; ts.cap = 0;
		const size_t cap = std::max<size_t>(4096, 2 * (tileBytes + startBytes));
		HIP_TRY(hipMalloc(&ts.dev, cap));
		HIP_TRY(hipHostMalloc(&ts.host, cap, hipHostMallocDefault));
		ts.cap = cap;
	}
	if (!ts.done) HIP_TRY(hipEventCreateWithFlags(&ts.done, hipEventDisableTiming));
	memcpy(ts.host, work.data(), tileBytes);
	memcpy((char *)ts.host + tileBytes, start.data(), startBytes);
	void *hostView = nullptr;
	HIP_TRY(hipHostGetDevicePointer(&hostView, ts.host, 0));
	c->workSlot++;

	StreamPlan Pl;
	Pl.tiles = (const crh_tile *)ts.dev;
	Pl.start = (const uint32_t *)((char *)ts.dev + tileBytes);
	Pl.ntiles = ntiles; Pl.npix = npix; Pl.cohorts = cohorts;
	Pl.passesPerChunk = C; Pl.lastPasses = lastPasses; Pl.chunkCount = chunkCount; Pl.chunkItems = chunkItems;
	Pl.genTotal = totalItems;
	Pl.slab = c->dStreamSlab; Pl.hit = c->dStreamHit; Pl.hitInst = c->dStreamHitInst;
	Pl.done = c->dStreamDone; Pl.seq = ++c->streamSeq;
	if (Pl.seq == 0u) Pl.seq = ++c->streamSeq;
	StreamPool pool[2];
	for (int b = 0; b < 2; ++b) {
		f4 *base = c->dStreamPlanes + (size_t)b * 4 * c->streamSlots;
		pool[b].p0 = base; pool[b].p1 = base + c->streamSlots; pool[b].p2 = base + 2 * c->streamSlots; pool[b].p3 = base + 3 * c->streamSlots;
		pool[b].count = c->dStreamCount + (size_t)b * (c->streamSlots / CRH_SF_COHORT);
	}

	crh_ctx::Timed ev;
	if (!c->eventPool.empty()) { ev = c->eventPool.back(); c->eventPool.pop_back(); }
	else { HIP_TRY(hipEventCreate(&ev.a)); HIP_TRY(hipEventCreate(&ev.b)); }
	HIP_TRY(hipEventRecord(ev.a, c->stream));
	hipLaunchKernelGGL(k_stream_init, dim3(std::min<uint32_t>(64u, (cohorts + 255u) / 256u)), dim3(256), 0, c->stream, Pl, pool[0].count, pool[1].count, c->dStreamCtl,
	                   (const uint32_t *)hostView, (uint32_t *)ts.dev, (uint32_t)((tileBytes + startBytes) / 4));
	hipError_t e = hipGetLastError();
	const uint32_t shadeGrid = std::min<uint32_t>((uint32_t)c->cuCount * (uint32_t)CRH_STREAM_SHADE_WPS, cohorts);
	const uint32_t foldGrid = std::max<uint32_t>(1u, std::min<uint32_t>((uint32_t)c->cuCount * 4u, (npix + CRH_BLOCK - 1u) / CRH_BLOCK));
	snprintf(c->lastKernel, sizeof(c->lastKernel), "k_stream<%d,%s> walk<%d,%d>", c->counterLevel >= 2 ? 2 : 1, c->hasPrograms ? "true" : "false",
	         deepStack ? CRH_STREAM_WALK_B_WPS : CRH_STREAM_WALK_A_WPS, deepStack ? CRH_STREAM_WALK_B_NLDS : CRH_STREAM_WALK_A_NLDS);
	c->lastGrid = walkGrid;
	auto iteration = [&](uint64_t it) {
		const StreamPool &in = pool[it & 1u], &out = pool[(it + 1u) & 1u];
#define CRH_STREAM_WALK(W, N, L) hipLaunchKernelGGL((k_stream_walk<W, N, true, L>), dim3(walkGrid), dim3(CRH_BLOCK), 0, c->stream, c->d, in, c->dStreamHit, c->dStreamHitInst, cohorts, \
		                                             c->dStreamCtl, c->sched, c->dStreamOvf, c->dCounters)
#define CRH_STREAM_SHADE(L, PROG) hipLaunchKernelGGL((k_stream_shade<L, PROG, 0>), dim3(shadeGrid), dim3(CRH_BLOCK), 0, c->stream, c->d, *P, Pl, in, out, c->dStreamCtl, c->dCounters)
#ifdef CRH_DEV_ONLY_BENCH_VARIANT
		if (deepStack) CRH_STREAM_WALK(CRH_STREAM_WALK_B_WPS, CRH_STREAM_WALK_B_NLDS, 1); else CRH_STREAM_WALK(CRH_STREAM_WALK_A_WPS, CRH_STREAM_WALK_A_NLDS, 1);
		CRH_STREAM_SHADE(1, false);
#else
		if (c->counterLevel >= 2) { if (deepStack) CRH_STREAM_WALK(CRH_STREAM_WALK_B_WPS, CRH_STREAM_WALK_B_NLDS, 2); else CRH_STREAM_WALK(CRH_STREAM_WALK_A_WPS, CRH_STREAM_WALK_A_NLDS, 2); }
		else { if (deepStack) CRH_STREAM_WALK(CRH_STREAM_WALK_B_WPS, CRH_STREAM_WALK_B_NLDS, 1); else CRH_STREAM_WALK(CRH_STREAM_WALK_A_WPS, CRH_STREAM_WALK_A_NLDS, 1); }
		if (c->counterLevel >= 2) { if (c->hasPrograms) CRH_STREAM_SHADE(2, true); else CRH_STREAM_SHADE(2, false); }
		else { if (c->hasPrograms) CRH_STREAM_SHADE(1, true); else CRH_STREAM_SHADE(1, false); }
#endif
#undef CRH_STREAM_WALK
#undef CRH_STREAM_SHADE
		hipLaunchKernelGGL(k_stream_fold, dim3(foldGrid), dim3(CRH_BLOCK), 0, c->stream, *P, Pl, c->dStreamCtl, dev_fb);
	};
	/* iterations in groups; behind every group an event. With group g + 1 enqueued the host waits for group g and looks at the completion word: the device always has a
	 * group ahead of it, and the iterations enqueued beyond the dispatch's end find nothing to do (each kernel leaves at its first instruction) */
	const uint64_t poolsOfWork = totalItems / slots + 2u;
	const uint64_t iterLimit = poolsOfWork * ((uint64_t)P->bounces + 2u) * 2u + 64u;          /* (every path ends within bounces iterations of its generation) */
	uint64_t it = 0;
	int pending = -1;                 /* the event slot of the group the host has not waited for yet */
	bool over = false;
	while (e == hipSuccess && !over) {
		const int g = pending == 0 ? 1 : 0;
		for (int k = 0; k < c->streamGroup; ++k) iteration(it++);
		e = hipGetLastError();
		if (e != hipSuccess) break;
		HIP_TRY(hipEventRecord(c->streamEv[g], c->stream));
		if (pending >= 0) {
			HIP_TRY(hipEventSynchronize(c->streamEv[pending]));
			over = *(volatile unsigned int *)c->hStreamDone == Pl.seq;
		}
		pending = g;
		if (!over && it > iterLimit) {
			HIP_TRY(hipStreamSynchronize(c->stream));
			if (*(volatile unsigned int *)c->hStreamDone == Pl.seq) break;
			c->eventPool.push_back(ev);
			return fail(CRH_ERR_HIP, "k_stream: the dispatch did not finish within its iteration limit: incomplete frame");
		}
	}
	c->streamIterations = it;
	HIP_TRY(hipEventRecord(ev.b, c->stream));
	HIP_TRY(hipEventRecord(ts.done, c->stream));
	ts.inFlight = true;
	c->pendingTimes.push_back(ev);
	c->launches++;
	if (c->janitorWaiting.load(std::memory_order_relaxed)) releaseJanitor(c, false);
	if (e != hipSuccess) return fail(CRH_ERR_HIP, std::string("k_stream launch: ") + hipGetErrorString(e));
	return CRH_OK;
}

int crh_render_tiles(crh_ctx *c, const crh_render_params *P, const crh_tile *tiles, uint32_t tile_count, float *dev_fb) {
	if (!c || !P || !dev_fb || (!tiles && tile_count)) return fail(CRH_ERR_INVALID, "crh_render_tiles: NULL argument");
	if (!c->haveScene) return fail(CRH_ERR_INVALID, "crh_render_tiles: no scene uploaded");
	if (P->image_width <= 0 || P->image_height <= 0 || P->pass_count < 0 || P->first_pass < 0 || P->max_passes < P->first_pass + P->pass_count)
		return fail(CRH_ERR_INVALID, "crh_render_tiles: bad render parameters");
	int rc = setDevice(c);
	if (rc) return rc;
	(void)resolveTimes(c, false);
	if (streamServes(c, P)) return renderStream(c, P, tiles, tile_count, dev_fb);
	const bool wg = c->kernel == CRH_KERNEL_WG;
	const PlanKnobs knobs{c->unitItems, c->unitsPerWave, c->tailPercent, c->tail2Percent, c->passChunk, c->cuCount, c->blocksPerCU, wg,
	                      rollForm(c) && P->bounces > 0 ? c->tailSplit : 0};
	WorkPlan plan;
	{
		std::string perr;
		const int prc = planWork(P, tiles, tile_count, knobs, plan, perr);
		if (prc != CRH_OK) return fail(prc, perr);
	}
	if (plan.total == 0 || P->pass_count == 0) return CRH_OK;
	const std::vector<crh_tile> &work = plan.work;
	const std::vector<uint32_t> &start = plan.start;
	const uint32_t work_count = (uint32_t)work.size();
	const uint64_t total = plan.total;
	const int bw = plan.bw, bh = plan.bh, sbw = plan.sbw, sbh = plan.sbh, tbw = plan.tbw, tbh = plan.tbh, chunk = plan.chunk;
	const uint32_t firstSmall = plan.firstSmall, firstTiny = plan.firstTiny, grid = plan.grid;
	c->lastGrid = grid;
	if (c->dWaveStats && grid * (CRH_BLOCK / 64) > CRH_WAVE_STATS_MAX) return fail(CRH_ERR_INVALID, "wave stats: grid too large");
	{
		size_t need = (size_t)grid * (wg ? 1 : CRH_BLOCK / 64) * (size_t)(bw * bh) * (size_t)chunk * 3;
		if (rollForm(c)) need *= CRH_ROLL_SLOTS;      /* one sample slab per open job */
		if (need > c->stageFloats) {
			HIP_TRY(hipStreamSynchronize(c->stream));
			if (c->dStage) HIP_TRY(hipFree(c->dStage));
			c->dStage = nullptr; c->stageFloats = 0;
			HIP_TRY(hipMalloc((void **)&c->dStage, need * sizeof(float)));
			c->stageFloats = need;
		}
	}

	const uint32_t deferUnits = plan.segs > 1 ? (uint32_t)total - start[plan.firstMicro] : 0u;       /* split pixels x segments */
	if (deferUnits) {
		const size_t need = (size_t)deferUnits * (size_t)plan.segPasses * 3;
		if (need > c->deferFloats) {
			HIP_TRY(hipStreamSynchronize(c->stream));
			if (c->dDefer) HIP_TRY(hipFree(c->dDefer));
			c->dDefer = nullptr; c->deferFloats = 0;
			HIP_TRY(hipMalloc((void **)&c->dDefer, need * sizeof(float)));
			c->deferFloats = need;
		}
	}
	{
		const size_t need = (size_t)grid * (CRH_BLOCK / 64) * CRH_WAVE_QUEUE_FLOATS;      /* = grid x CRH_WG_PATHS records for the workgroup kernel */
		static_assert(CRH_WG_PATHS * CRH_PATH_F4 * 4u == (CRH_BLOCK / 64) * CRH_WAVE_QUEUE_FLOATS, "both kernels use the same path-table footprint per workgroup");
		if (need > c->queueFloats) {
			HIP_TRY(hipStreamSynchronize(c->stream));
			if (c->dQueues) HIP_TRY(hipFree(c->dQueues));
			c->dQueues = nullptr; c->queueFloats = 0;
			HIP_TRY(hipMalloc((void **)&c->dQueues, need * sizeof(float)));
			c->queueFloats = need;
		}
	}

	{
		const size_t need = (size_t)grid * (CRH_BLOCK / 64) * CRH_OVF_WORDS_PER_WAVE;
		if (need > c->ovfWords) {
			HIP_TRY(hipStreamSynchronize(c->stream));
			if (c->dOvf) HIP_TRY(hipFree(c->dOvf));
			c->dOvf = nullptr; c->ovfWords = 0;
			HIP_TRY(hipMalloc((void **)&c->dOvf, need * sizeof(uint32_t)));
			c->ovfWords = need;
		}
	}

	/* per-launch tile list: pinned host slot -> device slot, asynchronously on the launch stream */
	const uint32_t slot = c->workSlot % CRH_WORK_SLOTS;          /* consumed below, once nothing can fail before the launch */
	crh_ctx::TileSlot &ts = c->tileSlots[slot];
	const size_t tileBytes = work_count * sizeof(crh_tile), startBytes = (work_count + 1) * sizeof(uint32_t);
	if (ts.inFlight) { HIP_TRY(hipEventSynchronize(ts.done)); ts.inFlight = false; }
	if (tileBytes + startBytes > ts.cap) {
		if (ts.dev) HIP_TRY(hipFree(ts.dev));
		if (ts.host) HIP_TRY(hipHostFree(ts.host));
		ts.dev = ts.host = nullptr; ts.cap = 0;
		const size_t cap = std::max<size_t>(4096, 2 * (tileBytes + startBytes));
		HIP_TRY(hipMalloc(&ts.dev, cap));
		HIP_TRY(hipHostMalloc(&ts.host, cap, hipHostMallocDefault));
		ts.cap = cap;
	}
	if (!ts.done) HIP_TRY(hipEventCreateWithFlags(&ts.done, hipEventDisableTiming));
	memcpy(ts.host, work.data(), tileBytes);
	memcpy((char *)ts.host + tileBytes, start.data(), startBytes);
	/* Nothing but the kernel itself is put on the stream in front of the kernel. Measured in round 3 (CRH_TRACE_SYNC, the drop-in's first dispatch): behind a
	 * 64-byte host-to-device copy and a 4-byte memset the kernel started 9-22 ms after its launch, whatever had been warmed up or waited for before — the copy
	 * engine's wake-up is the frame's. A short tile list (a frame, a GPU's strips) is therefore read by the waves straight from the pinned host slot (a wave
	 * looks up one tile per work unit: a few reads over the host link per millisecond of work); long lists (a cluster worker's batch) are copied as before. The
	 * work counter is reset behind the kernel instead of in front of it. CRH_TILES=copy forces the copy. */
	static const bool forceCopy = getenv("CRH_TILES") && !strcmp(getenv("CRH_TILES"), "copy");
	const bool zeroCopy = work_count <= 16 && !forceCopy;
	void *dTiles = ts.dev;
	if (zeroCopy) HIP_TRY(hipHostGetDevicePointer(&dTiles, ts.host, 0));
	else HIP_TRY(hipMemcpyAsync(ts.dev, ts.host, tileBytes + startBytes, hipMemcpyHostToDevice, c->stream));

	BlockQueue Q;
	Q.tiles = (const crh_tile *)dTiles;
	Q.start = (const uint32_t *)((char *)dTiles + tileBytes);
	Q.ntiles = work_count;
	Q.total = (uint32_t)total;
	Q.counter = c->dWork + slot;
	Q.bw = bw; Q.bh = bh;
	Q.firstSmall = firstSmall; Q.sbw = sbw; Q.sbh = sbh;
	Q.firstTiny = firstTiny; Q.tbw = tbw; Q.tbh = tbh;
	Q.firstMicro = plan.firstMicro; Q.mbw = plan.mbw; Q.mbh = plan.mbh; Q.segs = plan.segs; Q.segPasses = plan.segPasses;
	Q.unit0 = start[plan.firstMicro]; Q.defer = deferUnits ? c->dDefer : nullptr;
	c->workSlot++;

	if (P->bounces <= 0) {           /* every sample is black: no walk, only the running mean moves; paths are still counted */
		hipLaunchKernelGGL(k_fold_black, dim3(64, std::min<uint32_t>(work_count, 1024u)), dim3(256), 0, c->stream, *P, Q.tiles, work_count, dev_fb, c->dCounters);
		hipError_t e0 = hipGetLastError();
		if (e0 != hipSuccess) return fail(CRH_ERR_HIP, std::string("k_fold_black launch: ") + hipGetErrorString(e0));
		HIP_TRY(hipEventRecord(ts.done, c->stream));
		ts.inFlight = true;
		return CRH_OK;                                                   /* (k_fold_black takes no work units: the counter stays zero) */
	}
	crh_ctx::Timed ev;
	if (!c->eventPool.empty()) { ev = c->eventPool.back(); c->eventPool.pop_back(); }
	else { HIP_TRY(hipEventCreate(&ev.a)); HIP_TRY(hipEventCreate(&ev.b)); }
	HIP_TRY(hipEventRecord(ev.a, c->stream));
	hipError_t e = launchPathtrace(c, grid, P, Q, dev_fb, chunk);
	if (e == hipSuccess && deferUnits) {              /* the split pixels' samples -> the frame, in pass order (part of the dispatch and of its time) */
		const uint32_t px = deferUnits / (uint32_t)plan.segs;
		hipLaunchKernelGGL(k_fold_deferred, dim3((px + 255u) / 256u), dim3(256), 0, c->stream, *P, Q, px, dev_fb);
		e = hipGetLastError();
	}
	HIP_TRY(hipEventRecord(ev.b, c->stream));
	HIP_TRY(hipMemsetAsync(Q.counter, 0, sizeof(uint32_t), c->stream));          /* ready for the dispatch that takes this slot next */
	HIP_TRY(hipEventRecord(ts.done, c->stream));
	ts.inFlight = true;
	c->pendingTimes.push_back(ev);
	c->launches++;
	if (c->janitorWaiting.load(std::memory_order_relaxed)) {          /* a dispatch of some length is on the device: the host memory of the last upload can go back now */
		uint64_t paths = 0;
		for (const crh_tile &t : work) paths += (uint64_t)(t.x1 - t.x0) * (uint64_t)(t.y1 - t.y0);
		static const uint64_t minPaths = getenv("CRH_JANITOR_MIN_PATHS") ? strtoull(getenv("CRH_JANITOR_MIN_PATHS"), nullptr, 10) : CRH_JANITOR_MIN_PATHS;      /* (tests: 1 = at the first dispatch) */
		if (paths * (uint64_t)P->pass_count >= minPaths) releaseJanitor(c, false);
	}
	if (e != hipSuccess) return fail(CRH_ERR_HIP, std::string("k_pathtrace launch: ") + hipGetErrorString(e));
	return CRH_OK;
}

int crh_render_region(crh_ctx *c, const crh_render_params *P, float *dev_fb) {
	if (!P) return fail(CRH_ERR_INVALID, "crh_render_region: params is NULL");
	const crh_tile t{P->x0, P->y0, P->x1, P->y1};
	return crh_render_tiles(c, P, &t, 1, dev_fb);
}

/* ---- RCCL (loaded lazily: single-GPU users never need it) ---------------------------------------- */
namespace {
typedef void *ncclComm_t;
struct Rccl {
	void *lib = nullptr;
	int (*CommInitAll)(ncclComm_t *, int, const int *) = nullptr;
	int (*GroupStart)() = nullptr;
	int (*GroupEnd)() = nullptr;
	int (*Reduce)(const void *, void *, size_t, int, int, int, ncclComm_t, hipStream_t) = nullptr;
	int (*Send)(const void *, size_t, int, int, ncclComm_t, hipStream_t) = nullptr;
	int (*Recv)(void *, size_t, int, int, ncclComm_t, hipStream_t) = nullptr;
	const char *(*GetErrorString)(int) = nullptr;
	std::vector<int> devices;
	std::vector<ncclComm_t> comms;
	std::mutex mu;
} g_rccl;
const int kNcclFloat32 = 7, kNcclSum = 0;      /* ncclDataType_t / ncclRedOp_t values of rccl.h */

/* librccl + one communicator per device of `devs` (ncclCommInitAll); g_rccl.mu held by the caller. Creating the communicators takes
 * tens to hundreds of milliseconds: a host calls crh_frames_prepare() beside its scene set-up so that the frame does not pay for it. */
int rcclReady(const std::vector<int> &devs) {
	if (!g_rccl.lib) {
		g_rccl.lib = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
		if (!g_rccl.lib) g_rccl.lib = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
		if (!g_rccl.lib) return fail(CRH_ERR_HIP, std::string("cannot load librccl: ") + dlerror());
		g_rccl.CommInitAll = (decltype(g_rccl.CommInitAll))dlsym(g_rccl.lib, "ncclCommInitAll");
		g_rccl.GroupStart = (decltype(g_rccl.GroupStart))dlsym(g_rccl.lib, "ncclGroupStart");
		g_rccl.GroupEnd = (decltype(g_rccl.GroupEnd))dlsym(g_rccl.lib, "ncclGroupEnd");
		g_rccl.Reduce = (decltype(g_rccl.Reduce))dlsym(g_rccl.lib, "ncclReduce");
		g_rccl.Send = (decltype(g_rccl.Send))dlsym(g_rccl.lib, "ncclSend");
		g_rccl.Recv = (decltype(g_rccl.Recv))dlsym(g_rccl.lib, "ncclRecv");
		g_rccl.GetErrorString = (decltype(g_rccl.GetErrorString))dlsym(g_rccl.lib, "ncclGetErrorString");
		if (!g_rccl.CommInitAll || !g_rccl.GroupStart || !g_rccl.GroupEnd || !g_rccl.Reduce) {
			g_rccl.lib = nullptr;
			return fail(CRH_ERR_HIP, "librccl lacks the expected symbols");
		}
	}
	if (devs != g_rccl.devices) {
		g_rccl.comms.assign(devs.size(), nullptr);
		const int rc = g_rccl.CommInitAll(g_rccl.comms.data(), (int)devs.size(), devs.data());
		if (rc != 0) { g_rccl.devices.clear(); return fail(CRH_ERR_HIP, std::string("ncclCommInitAll: ") + (g_rccl.GetErrorString ? g_rccl.GetErrorString(rc) : "error")); }
		g_rccl.devices = devs;
	}
	return CRH_OK;
}
static std::string rcclError(const char *what, int rc) { return std::string(what) + ": " + (g_rccl.GetErrorString && rc > 0 ? g_rccl.GetErrorString(rc) : "error"); }
}

int crh_frames_prepare(const int *devices, int n) {
	if (!devices || n < 1) return fail(CRH_ERR_INVALID, "crh_frames_prepare: bad argument");
	if (n == 1 && !getenv("CRH_FORCE_RCCL")) return CRH_OK;
	std::lock_guard<std::mutex> lock(g_rccl.mu);
	return rcclReady(std::vector<int>(devices, devices + n));
}

int crh_frames_reduce(crh_ctx **ctxs, float **fbs, int n, int width, int height) {
	if (!ctxs || !fbs || n < 1 || width <= 0 || height <= 0) return fail(CRH_ERR_INVALID, "crh_frames_reduce: bad argument");
	for (int i = 0; i < n; ++i) if (!ctxs[i] || !fbs[i]) return fail(CRH_ERR_INVALID, "crh_frames_reduce: NULL context or framebuffer");
	if (n == 1 && !getenv("CRH_FORCE_RCCL")) return CRH_OK;     /* CRH_FORCE_RCCL: run the one-rank reduce through RCCL anyway (tests) */
	std::lock_guard<std::mutex> lock(g_rccl.mu);
	std::vector<int> devs(n);
	for (int i = 0; i < n; ++i) devs[i] = ctxs[i]->device;
	int rc = rcclReady(devs);                                   /* a no-op after crh_frames_prepare() */
	if (rc != CRH_OK) return rc;
	const size_t count = (size_t)width * height * 3;
	rc = g_rccl.GroupStart();
	for (int i = 0; i < n && rc == 0; ++i) {
		if (hipSetDevice(devs[i]) != hipSuccess) { rc = -1; break; }
		rc = g_rccl.Reduce(fbs[i], fbs[i], count, kNcclFloat32, kNcclSum, 0, g_rccl.comms[i], ctxs[i]->stream);
	}
	const int rcEnd = g_rccl.GroupEnd();
	if (rc == 0) rc = rcEnd;
	if (rc != 0) return fail(CRH_ERR_HIP, rcclError("ncclReduce", rc));
	for (int i = 0; i < n; ++i) {
		HIP_TRY(hipSetDevice(devs[i]));
		HIP_TRY(hipStreamSynchronize(ctxs[i]->stream));
	}
	return CRH_OK;
}

/* The rows of GPU g's strips (host/share.h: strip i = rows [i R, i R + R) counted from the bottom of the image, owned by GPU i mod n), in strip
 * order, between the float framebuffer (texture.c:24-28: row H - 1 - y) and a dense buffer. dir 0: pack, dir 1: unpack. */
__global__ void k_strip_rows(float *fb, float *dense, int width, int height, int stripRows, int g, int n, int rows, int dir) {
	const size_t rowFloats = (size_t)width * 3;
	for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < (size_t)rows * rowFloats; i += (size_t)gridDim.x * blockDim.x) {
		const int p = (int)(i / rowFloats);
		const int y = (g + (p / stripRows) * n) * stripRows + p % stripRows;       /* only the top strip of the image can be ragged, and it is the last one of its owner */
		if (y >= height) continue;
		float *at = fb + (size_t)(height - 1 - y) * rowFloats + (i - (size_t)p * rowFloats);
		if (dir == 0) dense[i] = *at; else *at = dense[i];
	}
}
static int stripRowsOwned(int height, int stripRows, int g, int n) {
	int rows = 0;
	for (int y = g * stripRows; y < height; y += n * stripRows) rows += std::min(stripRows, height - y);
	return rows;
}

int crh_frames_gather(crh_ctx **ctxs, float **fbs, int n, int width, int height, int strip_rows) {
	if (!ctxs || !fbs || n < 1 || width <= 0 || height <= 0 || strip_rows < 1) return fail(CRH_ERR_INVALID, "crh_frames_gather: bad argument");
	for (int i = 0; i < n; ++i) if (!ctxs[i] || !fbs[i]) return fail(CRH_ERR_INVALID, "crh_frames_gather: NULL context or framebuffer");
	if (n == 1) return CRH_OK;
	std::lock_guard<std::mutex> lock(g_rccl.mu);
	std::vector<int> devs(n);
	for (int i = 0; i < n; ++i) devs[i] = ctxs[i]->device;
	int rc = rcclReady(devs);
	if (rc != CRH_OK) return rc;
	if (!g_rccl.Send || !g_rccl.Recv) return fail(CRH_ERR_UNSUPPORTED, "crh_frames_gather: librccl has no ncclSend / ncclRecv");
	const size_t rowFloats = (size_t)width * 3;
	/* dense buffers: every sender's own rows on its device; on GPU 0 one slab per sender */
	std::vector<size_t> rows(n), at(n, 0);
	size_t total = 0;
	for (int g = 1; g < n; ++g) { rows[g] = (size_t)stripRowsOwned(height, strip_rows, g, n); at[g] = total; total += rows[g] * rowFloats; }
	auto room = [](crh_ctx *c, size_t floats) -> int {
		if (floats <= c->gatherFloats) return CRH_OK;
		HIP_TRY(hipSetDevice(c->device));
		if (c->dGather) HIP_TRY(hipFree(c->dGather));
		c->dGather = nullptr; c->gatherFloats = 0;
		HIP_TRY(hipMalloc((void **)&c->dGather, std::max<size_t>(floats, 1) * sizeof(float)));
		c->gatherFloats = floats;
		return CRH_OK;
	};
	if ((rc = room(ctxs[0], total)) != CRH_OK) return rc;
	for (int g = 1; g < n; ++g) {
		if ((rc = room(ctxs[g], rows[g] * rowFloats)) != CRH_OK) return rc;
		if (!rows[g]) continue;
		HIP_TRY(hipSetDevice(devs[g]));
		hipLaunchKernelGGL(k_strip_rows, dim3(1024), dim3(256), 0, ctxs[g]->stream, fbs[g], ctxs[g]->dGather, width, height, strip_rows, g, n, (int)rows[g], 0);
		HIP_TRY(hipGetLastError());
	}
	rc = g_rccl.GroupStart();
	for (int g = 1; g < n && rc == 0; ++g) {
		if (!rows[g]) continue;
		if (hipSetDevice(devs[g]) != hipSuccess) { rc = -1; break; }
		rc = g_rccl.Send(ctxs[g]->dGather, rows[g] * rowFloats, kNcclFloat32, 0, g_rccl.comms[g], ctxs[g]->stream);
		if (rc != 0) break;
		if (hipSetDevice(devs[0]) != hipSuccess) { rc = -1; break; }
		rc = g_rccl.Recv(ctxs[0]->dGather + at[g], rows[g] * rowFloats, kNcclFloat32, g, g_rccl.comms[0], ctxs[0]->stream);
	}
	const int rcEnd = g_rccl.GroupEnd();
	if (rc == 0) rc = rcEnd;
	if (rc != 0) return fail(CRH_ERR_HIP, rcclError("ncclSend / ncclRecv", rc));
	HIP_TRY(hipSetDevice(devs[0]));
	for (int g = 1; g < n; ++g) {
		if (!rows[g]) continue;
		hipLaunchKernelGGL(k_strip_rows, dim3(1024), dim3(256), 0, ctxs[0]->stream, fbs[0], ctxs[0]->dGather + at[g], width, height, strip_rows, g, n, (int)rows[g], 1);
		HIP_TRY(hipGetLastError());
	}
	for (int i = 0; i < n; ++i) {
		HIP_TRY(hipSetDevice(devs[i]));
		HIP_TRY(hipStreamSynchronize(ctxs[i]->stream));
	}
	return CRH_OK;
}

int crh_synchronize(crh_ctx *c) {
	if (!c) return fail(CRH_ERR_INVALID, "crh_synchronize: ctx is NULL");
	int rc = setDevice(c);
	if (rc) return rc;
	if (getenv("CRH_TRACE_SYNC") && !c->pendingTimes.empty()) {        /* dev: when does the stream reach the kernel, when does it leave it, when does the host notice? */
		const auto t0 = std::chrono::steady_clock::now();
		auto us = [&]() { return (long)std::chrono::duration_cast<std::chrono::microseconds>(std::chrono::steady_clock::now() - t0).count(); };
		const crh_ctx::Timed ev = c->pendingTimes.back();
		while (hipEventQuery(ev.a) == hipErrorNotReady) {}
		const long ta = us();
		while (hipEventQuery(ev.b) == hipErrorNotReady) {}
		const long tb = us();
		HIP_TRY(hipStreamSynchronize(c->stream));
		fprintf(stderr, "crh_synchronize trace: event before the kernel done after %ld us, event after it after %ld us, stream synchronized after %ld us\n", ta, tb, us());
	}
	HIP_TRY(hipStreamSynchronize(c->stream));
	rc = resolveTimes(c, true);
	for (auto &ts : c->tileSlots) ts.inFlight = false;
	if (rc == CRH_OK) rc = checkWatchdog(c);
	return rc;
}

int crh_counters_get(crh_ctx *c, crh_counters *out) {
	if (!c || !out) return fail(CRH_ERR_INVALID, "crh_counters_get: NULL argument");
	int rc = crh_synchronize(c);
	if (rc) return rc;
	unsigned long long h[8];
	HIP_TRY(hipMemcpy(h, c->dCounters, sizeof(h), hipMemcpyDeviceToHost));
	out->paths = h[0]; out->rays = h[1]; out->node_tests = h[2]; out->tri_tests = h[3];
	out->inst_visits = h[4]; out->inst_hits = h[5]; out->sphere_tests = h[6]; out->tex_fetches = h[7];
	return CRH_OK;
}

int crh_counters_reset(crh_ctx *c) {
	if (!c) return fail(CRH_ERR_INVALID, "crh_counters_reset: ctx is NULL");
	int rc = crh_synchronize(c);
	if (rc) return rc;
	HIP_TRY(hipMemset(c->dCounters, 0, CRH_COUNTER_BYTES(c->cuCount)));
	c->lastMs = 0.0f; c->totalMs = 0.0; c->launches = 0;
	return CRH_OK;
}

const char *crh_last_kernel_name(crh_ctx *c) { return c ? c->lastKernel : ""; }

int crh_kernel_time_ms(crh_ctx *c, float *last_ms, double *total_ms, uint64_t *launches) {
	if (!c) return fail(CRH_ERR_INVALID, "crh_kernel_time_ms: ctx is NULL");
	int rc = setDevice(c);
	if (rc) return rc;
	rc = resolveTimes(c, true);
	if (rc) return rc;
	if (last_ms) *last_ms = c->lastMs;
	if (total_ms) *total_ms = c->totalMs;
	if (launches) *launches = c->launches;
	return CRH_OK;
}

/* debug: wall-clock ticks (100 MHz), summed over waves, spent in {item setup, BVH walk, shading} by the counting kernel */
int crh_debug_phase_ticks(crh_ctx *c, uint64_t *out3 /* CRH_NCOUNTERS - 8 = 24 values: clocks {tri, node, shade}, wave steps {node, tri, ctrl, rounds, shade}, ctrl clock, lanes served {node, shade}, swap / gen clocks and counts, lanes {swap, tri, ctrl} */) {
	if (!c || !out3) return fail(CRH_ERR_INVALID, "crh_debug_phase_ticks: NULL argument");
	int rc = crh_synchronize(c);
	if (rc) return rc;
	unsigned long long h[CRH_NCOUNTERS - 8];
	HIP_TRY(hipMemcpy(h, c->dCounters + 8, sizeof(h), hipMemcpyDeviceToHost));
	for (int i = 0; i < CRH_NCOUNTERS - 8; ++i) out3[i] = h[i];
	/* ... plus the per-wave words of the rolling kernel (the other kernel forms add to the global counters) */
	const size_t waves = CRH_COUNTER_WAVES_MAX(c->cuCount);
	std::vector<unsigned long long> w(waves * CRH_NCOUNTERS);
	HIP_TRY(hipMemcpy(w.data(), c->dCounters + CRH_NCOUNTERS, w.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost));
	for (size_t v = 0; v < waves; ++v)
		for (int i = 8; i < CRH_NCOUNTERS; ++i) out3[i - 8] += w[v * CRH_NCOUNTERS + i];
	return CRH_OK;
}

/* debug: copy the per-wave {busy ticks @100 MHz, units} pairs of the last dispatch; returns the wave count */
int crh_debug_wave_stats(crh_ctx *c, uint64_t *out, uint32_t max_waves) {
	if (!c || !c->dWaveStats || !out) return fail(CRH_ERR_INVALID, "wave stats not enabled");
	int rc = crh_synchronize(c);
	if (rc) return rc;
#ifdef CRH_EXP_ABS_TIMES          /* tools/probe_finish.py: the rolling kernel writes a second record per wave behind the first ones */
	const uint32_t n = std::min<uint32_t>(max_waves, 2 * c->lastGrid * (CRH_BLOCK / 64));
#else
	const uint32_t n = std::min<uint32_t>(max_waves, c->lastGrid * (CRH_BLOCK / 64));
#endif
	HIP_TRY(hipMemcpy(out, c->dWaveStats, (size_t)n * 2 * sizeof(uint64_t), hipMemcpyDeviceToHost));
	return (int)n;
}

int crh_debug_eval_math(crh_ctx *c, int function, const float *x_host, const float *y_host, uint64_t n, float *out_host) {
	if (!c || !x_host || !out_host || function < 0 || function > CRH_MATH_ATAN2F) return fail(CRH_ERR_INVALID, "crh_debug_eval_math: bad argument");
	if (n == 0) return CRH_OK;
	int rc = setDevice(c);
	if (rc) return rc;
	float *dx = nullptr, *dy = nullptr, *dout = nullptr;
	hipError_t e = hipMalloc((void **)&dx, n * sizeof(float));
	if (e == hipSuccess) e = hipMalloc((void **)&dout, n * sizeof(float));
	if (e == hipSuccess && y_host) e = hipMalloc((void **)&dy, n * sizeof(float));
	if (e == hipSuccess) e = hipMemcpyAsync(dx, x_host, n * sizeof(float), hipMemcpyHostToDevice, c->stream);
	if (e == hipSuccess && y_host) e = hipMemcpyAsync(dy, y_host, n * sizeof(float), hipMemcpyHostToDevice, c->stream);
	if (e == hipSuccess) {
		hipLaunchKernelGGL(k_eval_math, dim3((unsigned)std::min<uint64_t>((n + 255) / 256, 16384)), dim3(256), 0, c->stream, function, dx, dy, n, dout);
		e = hipGetLastError();
	}
	if (e == hipSuccess) e = hipMemcpyAsync(out_host, dout, n * sizeof(float), hipMemcpyDeviceToHost, c->stream);
	if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
	if (dx) (void)hipFree(dx);
	if (dy) (void)hipFree(dy);
	if (dout) (void)hipFree(dout);
	if (e != hipSuccess) return fail(CRH_ERR_HIP, std::string("crh_debug_eval_math: ") + hipGetErrorString(e));
	return CRH_OK;
}

int crh_internal_device(crh_ctx *c) { return c->device; }
void *crh_internal_stream(crh_ctx *c) { return (void *)c->stream; }
int crh_internal_fail(int code, const char *message) { return fail(code, message); }
void *crh_internal_pinned(crh_ctx *c, size_t bytes) {
	if (bytes <= c->pinnedBytes) return c->pinned;
	if (c->pinned) (void)hipHostFree(c->pinned);
	c->pinned = nullptr; c->pinnedBytes = 0;
	bytes = (bytes + ((size_t)1 << 20)) & ~(((size_t)1 << 20) - 1);
	if (hipHostMalloc(&c->pinned, bytes, hipHostMallocDefault) != hipSuccess) { c->pinned = nullptr; return nullptr; }
	c->pinnedBytes = bytes;
	return c->pinned;
}

int crh_trace_rays(crh_ctx *c, const float *rays_host, uint64_t n, crh_hit *hits_host) {
	if (!c || (!rays_host && n) || (!hits_host && n)) return fail(CRH_ERR_INVALID, "crh_trace_rays: NULL argument");
	if (!c->haveScene) return fail(CRH_ERR_INVALID, "crh_trace_rays: no scene uploaded");
	if (c->hasVolumes) return fail(CRH_ERR_UNSUPPORTED, "crh_trace_rays: the scene has volume instances, whose intersection draws from a path's sampler (instance.c:74, 199)");
	if (n == 0) return CRH_OK;
	int rc = setDevice(c);
	if (rc) return rc;
	const uint32_t grid = (uint32_t)std::min<uint64_t>((uint64_t)c->cuCount * c->blocksPerCU, (n + CRH_BLOCK - 1) / CRH_BLOCK);
	float *dRays = nullptr;
	crh_hit *dHits = nullptr;
	hipError_t e = hipMalloc((void **)&dRays, n * 6 * sizeof(float));
	if (e == hipSuccess) e = hipMalloc((void **)&dHits, n * sizeof(crh_hit));
	if (e == hipSuccess) e = hipMemcpyAsync(dRays, rays_host, n * 6 * sizeof(float), hipMemcpyHostToDevice, c->stream);
	if (e == hipSuccess) {
		hipLaunchKernelGGL(k_trace_rays, dim3(grid), dim3(CRH_BLOCK), 0, c->stream, c->d, dRays, n, dHits, c->traceExactSlabs ? 0u : (uint32_t)CRH_RAY_LITERAL);
		e = hipGetLastError();
	}
	if (e == hipSuccess) e = hipMemcpyAsync(hits_host, dHits, n * sizeof(crh_hit), hipMemcpyDeviceToHost, c->stream);
	if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
	if (dRays) (void)hipFree(dRays);
	if (dHits) (void)hipFree(dHits);
	if (e != hipSuccess) return fail(CRH_ERR_HIP, std::string("crh_trace_rays: ") + hipGetErrorString(e));
	return CRH_OK;
}

/* ---- round 6: the walk-only probe (walk_probe.h). Debug entries: a measurement of the walk at occupancies the render kernel cannot reach, on the render
 * kernel's own rays. Nothing of the product path calls them. ---- */

/* The next dispatches of the counting kernel (CRH_OPT_COUNTER_LEVEL 2) record every ray a wave starts to walk, in the order it starts them, rays_per_wave of them per wave (0: off,
 * the buffers are released). Turns CRH_OPT_WAVE_STATS on (the descriptor lives behind the wave statistics). */
int crh_debug_ray_dump(crh_ctx *c, uint32_t rays_per_wave) {
	if (!c) return fail(CRH_ERR_INVALID, "crh_debug_ray_dump: ctx is NULL");
	int rc = crh_synchronize(c);
	if (rc) return rc;
	if (c->dDump) { (void)hipFree(c->dDump); c->dDump = nullptr; }
	for (int i = 0; i < 2; ++i) {
		if (c->dProbeHits[i]) { (void)hipFree(c->dProbeHits[i]); c->dProbeHits[i] = nullptr; }
		if (c->dProbeInst[i]) { (void)hipFree(c->dProbeInst[i]); c->dProbeInst[i] = nullptr; }
	}
	c->dumpCap = 0;
	if (!rays_per_wave) {
		if (c->dWaveStats) HIP_TRY(hipMemset(c->dWaveStats + CRH_DUMP_HDR, 0, (2u + CRH_WAVE_STATS_MAX) * sizeof(unsigned long long)));
		return CRH_OK;
	}
	rc = crh_set_option(c, CRH_OPT_WAVE_STATS, 1);
	if (rc) return rc;
	const size_t waves = (size_t)c->cuCount * c->blocksPerCU * (CRH_BLOCK / 64);
	if (waves > CRH_WAVE_STATS_MAX) return fail(CRH_ERR_INVALID, "crh_debug_ray_dump: too many waves");
	const size_t slots = waves * (size_t)rays_per_wave;
	if (slots >= ((size_t)1 << 32)) return fail(CRH_ERR_INVALID, "crh_debug_ray_dump: more than 2^32 ray slots");
	HIP_TRY(hipMalloc((void **)&c->dDump, slots * 6 * sizeof(float)));
	for (int i = 0; i < 2; ++i) {
		HIP_TRY(hipMalloc((void **)&c->dProbeHits[i], slots * sizeof(f4)));
		HIP_TRY(hipMalloc((void **)&c->dProbeInst[i], slots * sizeof(int32_t)));
		HIP_TRY(hipMemset(c->dProbeHits[i], 0, slots * sizeof(f4)));
		HIP_TRY(hipMemset(c->dProbeInst[i], 0, slots * sizeof(int32_t)));
	}
	c->dumpCap = rays_per_wave;
	std::vector<unsigned long long> hdr(2u + CRH_WAVE_STATS_MAX, 0ull);
	hdr[0] = (unsigned long long)(uintptr_t)c->dDump; hdr[1] = rays_per_wave;
	HIP_TRY(hipMemcpy(c->dWaveStats + CRH_DUMP_HDR, hdr.data(), hdr.size() * sizeof(unsigned long long), hipMemcpyHostToDevice));
	return CRH_OK;
}

/* rays in the list now (summed over the waves of the last counting dispatch that dumped); optionally every wave's count */
int crh_debug_ray_dump_counts(crh_ctx *c, uint64_t *total, uint32_t *per_wave, uint32_t max_waves) {
	if (!c || !c->dDump || !total) return fail(CRH_ERR_INVALID, "crh_debug_ray_dump_counts: no ray dump");
	int rc = crh_synchronize(c);
	if (rc) return rc;
	std::vector<unsigned long long> cnt(CRH_WAVE_STATS_MAX);
	HIP_TRY(hipMemcpy(cnt.data(), c->dWaveStats + CRH_DUMP_HDR + 2, cnt.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost));
	*total = 0;
	for (size_t i = 0; i < cnt.size(); ++i) { *total += cnt[i]; if (per_wave && i < max_waves) per_wave[i] = (uint32_t)cnt[i]; }
	return CRH_OK;
}

/* copy rays [first, first + n) of wave region `wave` (six floats each) to the host: lets a test hand the same rays to crh_trace_rays */
int crh_debug_ray_dump_fetch(crh_ctx *c, uint32_t wave, uint32_t first, uint32_t n, float *rays6_host) {
	if (!c || !c->dDump || !rays6_host) return fail(CRH_ERR_INVALID, "crh_debug_ray_dump_fetch: no ray dump");
	if ((uint64_t)first + n > c->dumpCap) return fail(CRH_ERR_INVALID, "crh_debug_ray_dump_fetch: beyond the wave's region");
	int rc = crh_synchronize(c);
	if (rc) return rc;
	HIP_TRY(hipMemcpy(rays6_host, c->dDump + ((size_t)wave * c->dumpCap + first) * 6, (size_t)n * 6 * sizeof(float), hipMemcpyDeviceToHost));
	return CRH_OK;
}

/* ... and the probe's hits for the same rays (out slot 0 / 1): t, u, v, slot bits per ray, and the instance (TLAS leaf order, -1: miss) */
int crh_debug_walk_probe_fetch(crh_ctx *c, int slot, uint32_t wave, uint32_t first, uint32_t n, float *hits4_host, int32_t *inst_host) {
	if (!c || !c->dDump || slot < 0 || slot > 1 || !hits4_host || !inst_host) return fail(CRH_ERR_INVALID, "crh_debug_walk_probe_fetch: bad argument");
	if ((uint64_t)first + n > c->dumpCap) return fail(CRH_ERR_INVALID, "crh_debug_walk_probe_fetch: beyond the wave's region");
	int rc = crh_synchronize(c);
	if (rc) return rc;
	HIP_TRY(hipMemcpy(hits4_host, c->dProbeHits[slot] + ((size_t)wave * c->dumpCap + first), (size_t)n * sizeof(f4), hipMemcpyDeviceToHost));
	HIP_TRY(hipMemcpy(inst_host, c->dProbeInst[slot] + ((size_t)wave * c->dumpCap + first), (size_t)n * sizeof(int32_t), hipMemcpyDeviceToHost));
	return CRH_OK;
}

/* Walk the dumped rays with k_walk_probe<wps, stack_lds, inst_lds, fused> (wps 0: the one-ray-per-lane reference form) into output `slot`; unit_rays = rays per unit of its work queue.
 * Supported (wps, stack_lds, inst_lds): see the table below. Reports the kernel's time (HIP events on the context's stream) and the rays walked. */
int crh_debug_walk_probe(crh_ctx *c, int wps, int stack_lds, int inst_lds, int fused, uint32_t unit_rays, int slot, float *ms_out, uint64_t *rays_out) {
	if (!c || !c->dDump || slot < 0 || slot > 1) return fail(CRH_ERR_INVALID, "crh_debug_walk_probe: no ray dump (crh_debug_ray_dump, then a counting dispatch)");
	if (!c->haveScene) return fail(CRH_ERR_INVALID, "crh_debug_walk_probe: no scene uploaded");
	if (c->hasVolumes) return fail(CRH_ERR_UNSUPPORTED, "crh_debug_walk_probe: the scene has volume instances (their walks draw from the path's sampler)");
	if (unit_rays < 64) return fail(CRH_ERR_INVALID, "crh_debug_walk_probe: at least 64 rays per unit");
	int rc = crh_synchronize(c);
	if (rc) return rc;
	const size_t dumpWaves = (size_t)c->cuCount * c->blocksPerCU * (CRH_BLOCK / 64);
	std::vector<unsigned long long> cnt(CRH_WAVE_STATS_MAX);
	HIP_TRY(hipMemcpy(cnt.data(), c->dWaveStats + CRH_DUMP_HDR + 2, cnt.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost));
	/* the queue: piece k of every wave's region before piece k + 1 of any, so that the waves of the probe work at about the same depth of the path tracer's waves' histories */
	std::vector<ProbeUnit> units;
	uint64_t total = 0;
	unsigned long long deepest = 0;
	for (size_t w = 0; w < dumpWaves; ++w) { total += cnt[w]; deepest = std::max(deepest, cnt[w]); }
	for (unsigned long long k = 0; k * unit_rays < deepest; ++k)
		for (size_t w = 0; w < dumpWaves; ++w)
			if (k * unit_rays < cnt[w]) units.push_back(ProbeUnit{(uint32_t)(w * c->dumpCap + k * unit_rays), (uint32_t)std::min<unsigned long long>(unit_rays, cnt[w] - k * unit_rays)});
	if (rays_out) *rays_out = total;
	if (units.empty()) return fail(CRH_ERR_INVALID, "crh_debug_walk_probe: the ray list is empty");
	if ((units.size() + 1) * sizeof(ProbeUnit) > c->probeUnitCap) {
		if (c->dProbeUnits) (void)hipFree(c->dProbeUnits);
		c->dProbeUnits = nullptr; c->probeUnitCap = 0;
		HIP_TRY(hipMalloc(&c->dProbeUnits, (units.size() + 1) * sizeof(ProbeUnit)));
		c->probeUnitCap = (units.size() + 1) * sizeof(ProbeUnit);
	}
	/* word 0 of the buffer is the queue's counter, the units follow */
	HIP_TRY(hipMemset(c->dProbeUnits, 0, sizeof(ProbeUnit)));
	HIP_TRY(hipMemcpy((ProbeUnit *)c->dProbeUnits + 1, units.data(), units.size() * sizeof(ProbeUnit), hipMemcpyHostToDevice));
	const size_t ovfWords = (size_t)c->cuCount * 8u * (CRH_BLOCK / 64) * CRH_OVF_WORDS_PER_WAVE;
	if (!c->dProbeOvf) HIP_TRY(hipMalloc((void **)&c->dProbeOvf, ovfWords * sizeof(uint32_t)));
	crh_ctx::Timed ev;
	if (!c->eventPool.empty()) { ev = c->eventPool.back(); c->eventPool.pop_back(); }
	else { HIP_TRY(hipEventCreate(&ev.a)); HIP_TRY(hipEventCreate(&ev.b)); }
	hipError_t e = hipSuccess;
	bool launched = false;
	const uint32_t slots = (uint32_t)(dumpWaves * c->dumpCap);
	if (wps == 0) {
		HIP_TRY(hipEventRecord(ev.a, c->stream));
		hipLaunchKernelGGL(k_walk_simple, dim3((uint32_t)c->cuCount * 8u), dim3(CRH_BLOCK), 0, c->stream, c->d, c->dDump, (uint64_t)slots, c->dWaveStats + CRH_DUMP_HDR + 2, c->dumpCap, c->dProbeHits[slot], c->dProbeInst[slot], (uint32_t)c->sched.rayFlags);
		e = hipGetLastError();
		HIP_TRY(hipEventRecord(ev.b, c->stream));
		launched = true;
	}
#if defined(__HIPCC__)          /* (a workgroup of more than 64 KB of LDS — two workgroups per CU — has to be allowed) */
#define CRH_PROBE_ALLOW_LDS(kernel, bytes) do { if ((bytes) > 32768) HIP_TRY(hipFuncSetAttribute((const void *)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(bytes))); } while (0)
#else
#define CRH_PROBE_ALLOW_LDS(kernel, bytes) do { } while (0)
#endif
#define CRH_PROBE_VARIANT(W, N, I, F) \
	if (!launched && wps == W && stack_lds == N && (inst_lds != 0) == I && fused == F) { \
		hipFuncAttributes fa; \
		HIP_TRY(hipFuncGetAttributes(&fa, (const void *)k_walk_probe<W, N, I, F>)); \
		/* a CU's 160 KB must hold exactly W workgroups: the allocation granule is not documented (r06b: 3 x 54272 B and 5 x 32768 B did NOT fit, 4 x 40960 B does — 1280 B \
		 * would explain all three), so the size aimed for leaves 2 KB of slack per workgroup and W + 1 of them still exceed the 160 KB */ \
		const size_t want = std::max<size_t>(fa.sharedSizeBytes, ((163840u / W - 2048u) & ~255u)); \
		if ((W + 1) * want <= 163840u) return fail(CRH_ERR_INVALID, "crh_debug_walk_probe: this variant's LDS does not pin its occupancy"); \
		const size_t pad = want - fa.sharedSizeBytes; \
		CRH_PROBE_ALLOW_LDS((k_walk_probe<W, N, I, F>), pad); \
		snprintf(c->lastKernel, sizeof(c->lastKernel), "k_walk_probe<%d,%d,%s,%s> vgpr %d lds %zu+%zu", W, N, I ? "true" : "false", F == 1 ? "fused" : F == 2 ? "flat" : F == 3 ? "lean1" : "lean", fa.numRegs, (size_t)fa.sharedSizeBytes, pad); \
		HIP_TRY(hipEventRecord(ev.a, c->stream)); \
		hipLaunchKernelGGL((k_walk_probe<W, N, I, F>), dim3((uint32_t)c->cuCount * W), dim3(CRH_BLOCK), pad, c->stream, c->d, c->dDump, (const ProbeUnit *)c->dProbeUnits + 1, (uint32_t)units.size(), \
		                   (uint32_t *)c->dProbeUnits, c->dProbeHits[slot], c->dProbeInst[slot], c->sched, c->dProbeOvf); \
		e = hipGetLastError(); \
		HIP_TRY(hipEventRecord(ev.b, c->stream)); \
		launched = true; \
	}
	CRH_PROBE_VARIANT(4, 12, true, 1)
	CRH_PROBE_VARIANT(4, 12, true, 0)
	CRH_PROBE_VARIANT(2, 12, true, 3)
	CRH_PROBE_VARIANT(3, 12, true, 3)
	CRH_PROBE_VARIANT(4, 12, true, 3)
	CRH_PROBE_VARIANT(4, 7, true, 3)
	CRH_PROBE_VARIANT(4, 3, true, 3)
	CRH_PROBE_VARIANT(4, 4, false, 3)
	CRH_PROBE_VARIANT(5, 12, true, 3)
	CRH_PROBE_VARIANT(6, 7, true, 3)
	CRH_PROBE_VARIANT(6, 7, true, 0)
	CRH_PROBE_VARIANT(7, 3, true, 3)
	CRH_PROBE_VARIANT(8, 4, false, 3)
#undef CRH_PROBE_VARIANT
	if (!launched) { c->eventPool.push_back(ev); return fail(CRH_ERR_INVALID, "crh_debug_walk_probe: no such variant (wps, stack_lds, inst_lds)"); }
	if (e != hipSuccess) { c->eventPool.push_back(ev); return fail(CRH_ERR_HIP, std::string("crh_debug_walk_probe: ") + hipGetErrorString(e)); }
	HIP_TRY(hipStreamSynchronize(c->stream));
	float ms = 0.0f;
	HIP_TRY(hipEventElapsedTime(&ms, ev.a, ev.b));
	c->eventPool.push_back(ev);
	if (ms_out) *ms_out = ms;
	return CRH_OK;
}

/* hits (bit patterns of t, u, v, slot; instance) that differ between the probe's two outputs, over every slot of the list */
int crh_debug_walk_probe_compare(crh_ctx *c, uint64_t *differ) {
	if (!c || !c->dDump || !differ) return fail(CRH_ERR_INVALID, "crh_debug_walk_probe_compare: no ray dump");
	int rc = crh_synchronize(c);
	if (rc) return rc;
	const uint64_t slots = (uint64_t)c->cuCount * c->blocksPerCU * (CRH_BLOCK / 64) * c->dumpCap;
	unsigned long long *d = nullptr, h = 0;
	HIP_TRY(hipMalloc((void **)&d, sizeof(h)));
	HIP_TRY(hipMemset(d, 0, sizeof(h)));
	hipLaunchKernelGGL(k_probe_compare, dim3((uint32_t)c->cuCount * 8u), dim3(256), 0, c->stream, c->dProbeHits[0], c->dProbeInst[0], c->dProbeHits[1], c->dProbeInst[1], slots, d);
	hipError_t e = hipGetLastError();
	if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
	if (e == hipSuccess) e = hipMemcpy(&h, d, sizeof(h), hipMemcpyDeviceToHost);
	(void)hipFree(d);
	if (e != hipSuccess) return fail(CRH_ERR_HIP, std::string("crh_debug_walk_probe_compare: ") + hipGetErrorString(e));
	*differ = h;
	return CRH_OK;
}

}  // extern "C"

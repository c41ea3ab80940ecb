/*
 * cray_hip.hip — libcray_hip.so: the C-ABI of include/cray_hip.h and the gfx950 kernels behind it.
 *
 * Kernels (all hand-written for CDNA4, wave = 64):
 *   k_pathtrace_roll<LEVEL, WPS, PROG, SAMP>  (pathtrace_roll.h) the hot kernel since the end of round 3: k_pathtrace's machine with up to four work units
 *                        open per wave, so that the path table stays full across unit boundaries.
 *   k_pathtrace<LEVEL, WPS, PROG, SAMP>   the same machine, one work unit at a time (CRH_KERNEL_WAVE): persistent grid, every wave a small wavefront machine. A wave
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
#include <mutex>
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
/* Quad-cooperative record fetch (round 3 experiment, -DCRH_EXP_COOP_FETCH, fetch64 below): the vector L1 prices a divergent 16-byte-per-lane load by the
 * cache lines it touches per instruction, and a lane that reads a 64-byte record with four loads pays for the same line four times (tools/ubench_l1.hip on
 * the MI355X, 16 waves per CU, dependent chain: 1201 ns per wave-step in an L2-resident set, 64 lanes; the same bytes fetched by the four lanes of a quad
 * side by side, one instruction per quad member: 470 ns; 826 -> 479 ns at 32 lanes in a 160 MB set). The records land in LDS (global_load_lds_dwordx4),
 * 4160 bytes per wave, paid for with ten stack entries and seven park slots. In the real kernel it LOSES: the ~22 vector and ~16 scalar instructions per node
 * step it adds for ALL 64 lanes (quad broadcasts, exec masks, M0) cost more than the L1 look-ups it saves, because the vector ALU is the other
 * resource the walk is short of (74 % busy): hdr.json -12 %, statues -8 %, 1 M soup -11 % on top of the -2...-9 % of the smaller LDS stack and
 * the re-derived slab constants (profiles/r03c_ab_coop_fetch.log; frames bit-identical). Kept as a tested option. */
#ifdef CRH_EXP_COOP_FETCH          /* measured slower in the real kernel (profiles/r03c_ab_coop_fetch.log): not in the default library */
#define CRH_COOP_FETCH 1
#endif
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
/* Top-level BVH in LDS (-DCRH_TLAS_LDS=1, with the instance records: scenes with at most CRH_INST_LDS_MAX instances have at most 31 TLAS nodes = 1 KB): every
 * ray starts with three or four node steps in that tiny tree, which always hit the L1 and still cost it four look-ups per lane each. */
#ifndef CRH_TLAS_LDS
#define CRH_TLAS_LDS 0
#endif
#define CRH_TLAS_LDS_NODES 32u
#ifndef CRH_STACK_LDS
#ifdef CRH_COOP_FETCH
#define CRH_STACK_LDS 13         /* traversal stack entries kept in LDS per lane; with the 6 park slots, the fetch slabs, the id stacks and cursors: < 40 KB per block, 4 blocks per CU */
#else
#define CRH_STACK_LDS ((40960 - 3968 - (int)CRH_INST_LDS_BYTES - (int)CRH_SHADE_LDS_BYTES - CRH_TLAS_LDS * 1024) / 1024 - 10)     /* what the LDS holds after the id stacks, tables and park slots (18 with the default instance tables) */
#endif
#endif
#define CRH_REC_STRIDE_WORDS 260u                          /* one slab = what one global_load_lds_dwordx4 writes (64 lanes x 16 B) + 16 B of skew: the four lanes of a quad
                                                            * read their records from four slabs, and the skew puts the 16 lanes of a ds_read_b128 pass on distinct banks */
#define CRH_REC_WORDS_PER_WAVE (4u * CRH_REC_STRIDE_WORDS)

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
#if CRH_TLAS_LDS
	const lds_u32 *tlas; /* workgroup-uniform: the LDS copy of TLAS nodes 1 .. (8 words each), or null */
	__device__ __forceinline__ bool tlasInLds() const { return tlas != nullptr; }
	__device__ __forceinline__ void nodePair(const DScene &S, uint32_t node, f4 &l0, f4 &l1, f4 &r0, f4 &r1) const {
		const lds_u32 *p = tlas + (node - S.tlas_first) * 8u;
		l0 = ldsLoadF4(p); l1 = ldsLoadF4(p + 4); r0 = ldsLoadF4(p + 8); r1 = ldsLoadF4(p + 12);
	}
#endif
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
	return G;
}

__device__ __forceinline__ uint32_t waveSum(uint32_t v) {
	for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
	return v;
}

#define CRH_NCOUNTERS 32

/* scheduler weights: score of a step kind = lanes waiting for it x weight (weight ~ 1 / cost of the step) */
struct Sched { int wNode, wTri, wCtrl, swapMin, fillTo, runNum, triInRun, ctrlInRun, shadeMin;
               int sortFrom; };     /* CRH_OPT_SHADE_SORT: hits are shaded in batches of few shade classes in scenes with at least this many classes; 0 = never (default) */

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

/* value of lane m of this lane's quad (DPP quad_perm [m,m,m,m]); every lane of the wave must execute it */
__device__ __forceinline__ uint32_t quadBcast(uint32_t v, int m) {
	switch (m) {
		case 0: return (uint32_t)__builtin_amdgcn_mov_dpp((int)v, 0x00, 0xF, 0xF, true);
		case 1: return (uint32_t)__builtin_amdgcn_mov_dpp((int)v, 0x55, 0xF, 0xF, true);
		case 2: return (uint32_t)__builtin_amdgcn_mov_dpp((int)v, 0xAA, 0xF, 0xF, true);
		default: return (uint32_t)__builtin_amdgcn_mov_dpp((int)v, 0xFF, 0xF, 0xF, true);
	}
}
#ifndef CRH_WAIT_VMEM             /* every vector-memory operation of the wave has completed (the LDS writes of global_load_lds among them); the emulation's loads are synchronous */
#define CRH_WAIT_VMEM() asm volatile("s_waitcnt vmcnt(0)" ::: "memory")
#endif
/* One 64-byte record per lane — base[fi .. fi + 4), fi = CRH_NONE for a lane that wants none — fetched by the QUADS of the wave: for each of the four members of a
 * quad in turn, its four lanes load one quarter each (64 contiguous bytes: one cache-line look-up per quad and instruction instead of four per lane), straight
 * into the wave's LDS slabs; then every lane reads its own record back. ALL 64 lanes must call it (a lane serves its quad's members even when it wants nothing). */
__device__ __forceinline__ void fetch64(lds_u32 *slab, uint32_t lane, const f4 *base, uint32_t fi, f4 &a, f4 &b, f4 &c, f4 &d) {
	const uint32_t q = lane & 3u;
#pragma unroll
	for (int m = 0; m < 4; ++m) {
		const uint32_t fm = quadBcast(fi, m);
		if (fm != CRH_NONE)
			__builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(base + fm + q), (__attribute__((address_space(3))) void *)(slab + (uint32_t)m * CRH_REC_STRIDE_WORDS), 16, 0, 0);
	}
	CRH_WAIT_VMEM();
	CRH_LOCKSTEP();               /* the quarters other lanes fetched are in LDS */
	if (fi != CRH_NONE) {
		const lds_u32 *mine = slab + q * CRH_REC_STRIDE_WORDS + (lane & ~3u) * 4u;
		a = ldsLoadF4(mine); b = ldsLoadF4(mine + 4); c = ldsLoadF4(mine + 8); d = ldsLoadF4(mine + 12);
	}
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
template <int LEVEL, int WPS, bool PROG, int SAMP>
__global__ __launch_bounds__(CRH_BLOCK, CRH_WPS_OVERRIDE) void k_pathtrace(const DScene Sarg, const crh_render_params P, const BlockQueue Q, float *fb,
														   unsigned long long *counters,
														   float *stage, int chunk, unsigned long long *waveStats, const Sched K, float *queues, uint32_t *ovfAll) {
	__shared__ uint32_t s_stack[CRH_STACK_LDS * CRH_BLOCK];
	__shared__ uint32_t s_park[CRH_PARK_SLOTS * CRH_BLOCK];
#ifdef CRH_COOP_FETCH
	__shared__ __attribute__((aligned(16))) uint32_t s_rec[(CRH_BLOCK / 64) * CRH_REC_WORDS_PER_WAVE];
	static_assert((CRH_STACK_LDS + CRH_PARK_SLOTS) * CRH_BLOCK * 4 + (CRH_BLOCK / 64) * (CRH_IDS_BYTES + 32 + CRH_REC_WORDS_PER_WAVE * 4) + 512 + 256 <= 40960, "4 blocks per CU share 160 KB of LDS (incl. powf's tables)");
	lds_u32 *const rec = (lds_u32 *)&s_rec[(threadIdx.x >> 6) * CRH_REC_WORDS_PER_WAVE];
#else
	static_assert((CRH_STACK_LDS + CRH_PARK_SLOTS) * CRH_BLOCK * 4 + (CRH_BLOCK / 64) * (CRH_IDS_BYTES + 32) + 512 + 256 + CRH_INST_LDS_BYTES + CRH_SHADE_LDS_BYTES + CRH_TLAS_LDS * CRH_TLAS_LDS_NODES * 32 <= 40960, "4 blocks per CU share 160 KB of LDS (incl. powf's tables)");
#endif
	const DScene S = globalize(Sarg);
	CRH_EM_POW_TABLES_INIT();
	const unsigned long long tStart = wall_clock64();
	uint32_t unitsDone = 0;
	LdsStack stk;
	stk.lds = (lds_u32 *)&s_stack[threadIdx.x];
	stk.parkp = (lds_u32 *)&s_park[threadIdx.x];
	CountersT<LEVEL, PROG> cnt;
	memset(&cnt, 0, sizeof(cnt));
	const uint32_t lane = threadIdx.x & 63u;
	const uint32_t wave = (blockIdx.x * CRH_BLOCK + threadIdx.x) >> 6;
	/* hits are shaded in batches of few shade classes (ST_SHADE) when the scene has many of them: with two or three classes a mixed batch
	 * runs little extra code and the bookkeeping costs more than it saves (measured: statues -1 %, venus -3 %; hdr.json, six classes: +3 %).
	 * The instances' classes sit in an LDS table (a retiring walk looks its class up): scenes with more than 256 instances do not sort. */
	__shared__ uint8_t s_cls[256];
	/* (Round 2 turned this on for scenes with four or more classes: hdr.json +2...3 %. Since round 3's shading code runs far less per kind, the bookkeeping costs more
	 * than the purer batches save: hdr.json +1.5 % WITHOUT it, profiles/r03z_ab_shade_sort.log. Off by default; CRH_OPT_SHADE_SORT turns it on.) */
	const bool sorted = K.sortFrom > 0 && S.shade_classes >= (uint32_t)K.sortFrom && S.instance_count <= 256u;
	if (sorted) {
		for (uint32_t i = threadIdx.x; i < S.instance_count; i += CRH_BLOCK) s_cls[i] = (uint8_t)CRH_DINST_CLASS(S.instances[i].kind);
		__syncthreads();
	}
	stk.ovf = (glb_u32 *)ovfAll + (size_t)__builtin_amdgcn_readfirstlane(wave) * CRH_OVF_WORDS_PER_WAVE;
#if CRH_TLAS_LDS
	__shared__ __attribute__((aligned(16))) uint32_t s_tlas[CRH_TLAS_LDS_NODES * 8u];
	stk.tlas = nullptr;
	if (S.tlas_node_count > 1u && S.tlas_node_count - 1u <= CRH_TLAS_LDS_NODES) {
		for (uint32_t i = threadIdx.x; i < (S.tlas_node_count - 1u) * 8u; i += CRH_BLOCK) s_tlas[i] = ((const uint32_t *)(S.nodes + 2u * S.tlas_first))[i];
		__syncthreads();
		stk.tlas = (const lds_u32 *)s_tlas;
	}
#endif
	CRH_STAGE_SHADE_TABLES();
	CRH_STAGE_INSTANCE_TABLES();
#ifndef CRH_EXP_NO_UNIFORM_BASES  /* the wave's slab and path table start at wave-uniform addresses: said so (readfirstlane), their accesses use a scalar base + a 32-bit lane offset instead
                                   * of 64-bit vector address arithmetic and two more VGPRs each: 26 -> 16 spilled VGPRs, hdr.json +4 %, the others +1...2 % (profiles/r03q_ab_uniform_bases.log) */
	float *myStage = stage + (size_t)__builtin_amdgcn_readfirstlane(wave) * ((size_t)Q.bw * Q.bh * chunk * 3);
#else
	float *myStage = stage + (size_t)wave * ((size_t)Q.bw * Q.bh * chunk * 3);
#endif
	const int passEnd = P.first_pass + P.pass_count;
	/* the wave's path table, its id stacks and their wave-uniform fill levels (LDS: lane 0 writes, every lane reads; as
	 * plain variables they would be scalar registers live across the whole machine, and the register allocator is past
	 * its limits there — measured slower, and wrong images in the variant that calls runProgram) */
#ifndef CRH_EXP_NO_UNIFORM_BASES
	f4 *const ptab = (f4 *)(queues + (size_t)__builtin_amdgcn_readfirstlane(wave) * CRH_WAVE_QUEUE_FLOATS);
#else
	f4 *const ptab = (f4 *)(queues + (size_t)wave * CRH_WAVE_QUEUE_FLOATS);
#endif
	enum { WQ_RAYS, WQ_HITS, WQ_MISSES, WQ_FREE, WQ_NEXT_ITEM, WQ_CLS_LO, WQ_CLS_HI, WQ_WORDS };      /* CLS_LO / CLS_HI: hits waiting per shade class, 8 bits each */
	__shared__ int s_wq[(CRH_BLOCK / 64) * WQ_WORDS];
	__shared__ __attribute__((aligned(2))) uint8_t s_ids[(CRH_BLOCK / 64) * CRH_IDS_BYTES];
	typedef volatile __attribute__((address_space(3))) int lds_int;
	typedef volatile __attribute__((address_space(3))) uint8_t lds_u8;
	typedef volatile __attribute__((address_space(3))) uint16_t lds_u16;
	lds_int *const wq = (lds_int *)&s_wq[(threadIdx.x >> 6) * WQ_WORDS];
	lds_u8 *const ids = (lds_u8 *)&s_ids[(threadIdx.x >> 6) * CRH_IDS_BYTES];
	lds_u16 *const hits = (lds_u16 *)&s_ids[(threadIdx.x >> 6) * CRH_IDS_BYTES + CRH_IDS_HITS];
	for (;;) {
		uint32_t unit = 0;
		if (lane == 0) unit = atomicAdd((uint32_t *)(__attribute__((address_space(1))) uint32_t *)Q.counter, 1u);
		unit = __builtin_amdgcn_readfirstlane(unit);
		if (unit >= Q.total) break;
		++unitsDone;
		uint32_t lo = 0, hi = Q.ntiles;           /* largest t with start[t] <= unit */
		while (hi - lo > 1) { const uint32_t mid = (lo + hi) >> 1; if (asGlobal(Q.start)[mid] <= unit) lo = mid; else hi = mid; }
		const crh_tile t = asGlobal(Q.tiles)[lo];
		const uint32_t local = unit - asGlobal(Q.start)[lo];
		const int ubw = lo >= Q.firstTiny ? Q.tbw : lo >= Q.firstSmall ? Q.sbw : Q.bw, ubh = lo >= Q.firstTiny ? Q.tbh : lo >= Q.firstSmall ? Q.sbh : Q.bh;
		const uint32_t nbx = (uint32_t)(t.x1 - t.x0 + ubw - 1) / (uint32_t)ubw;
		BlockJob J;
		J.bw = ubw; J.bh = ubh;
		J.x0 = t.x0 + (int)(local % nbx) * ubw;
		J.y0 = t.y0 + (int)(local / nbx) * ubh;
		J.w = min(ubw, t.x1 - J.x0);
		J.h = min(ubh, t.y1 - J.y0);
		for (int c0 = P.first_pass; c0 < passEnd; c0 += chunk) {
			J.passBegin = c0;
			J.passCount = min(chunk, passEnd - c0);
			const uint32_t nItems = (uint32_t)(J.bw * J.bh * J.passCount);       /* incl. the padding of ragged tile edges */
			const uint32_t validItems = (uint32_t)(J.w * J.h * J.passCount);
			/*
			 * ---- the wave as a small wavefront machine -----------------------------------------------------------------
			 * Paths are decoupled from lanes. A path's state (ray, weight, radiance, RNG, depth, item, last hit) lives in
			 * one slot of the wave's path table (global memory, cache-resident); three LDS byte stacks hold the slot ids of
			 * RAYS waiting for a walker, surface HITS and MISSES waiting for shading, a fourth the free slots. Lanes are
			 * workers: every iteration the wave ballots what its lanes need and runs ONE kind of step:
			 *   NODE / TRI / CTRL  walk steps (state machine of pt_device.h), picked by lanes x weight;
			 *   SWAP   lanes whose walk ended write the hit into their path's slot and push its id (compacted with
			 *          ballot + mbcnt); they and the idle lanes pop ray ids and start those walks (6 words each);
			 *   GEN    all 64 lanes start the next 64 items (initSampler + getCameraRay) in free slots;
			 *   SHADE  all 64 lanes shade 64 surface hits (finishHit, emission, bsdf sample, roulette): continuing paths
			 *          are updated in place and their ids pushed on the ray stack, finished samples staged;
			 *   MISS   all 64 lanes evaluate the background for 64 rays that left the scene (always the end of a path).
			 * The expensive steps therefore run at (close to) full occupancy, without the divergence between surface and
			 * background code, and the walk steps always have rays. Each path's own sequence of operations — hence
			 * every result — is independent of the schedule.
			 */
			if (lane == 0) { wq[WQ_RAYS] = 0; wq[WQ_HITS] = 0; wq[WQ_MISSES] = 0; wq[WQ_FREE] = (int)CRH_PATHS; wq[WQ_NEXT_ITEM] = 0; wq[WQ_CLS_LO] = 0; wq[WQ_CLS_HI] = 0; }
			for (uint32_t i = lane; i < CRH_PATHS; i += 64u) ids[i] = (uint8_t)i;              /* all slots free: the free stack covers bytes 0..255 */
			Walk w;
			memset(&w, 0, sizeof(w));
			w.phase = PH_IDLE;
			uint32_t myPath = 0;
			for (;;) {
				const uint32_t ph = w.phase;
				TablePort<SAMP> port{ptab + myPath * CRH_PATH_F4};           /* the walking lane's path (volumes draw from its sampler) */
				const int nN = __popcll(__ballot(ph == PH_NODE)), nT = __popcll(__ballot(ph == PH_TRI)), nC = __popcll(__ballot(ph == PH_CTRL || ph == PH_NODE_SLOW));
				const int nF = __popcll(__ballot(ph == PH_SHADE));           /* walks that ended, result not yet queued */
				const int nE = 64 - nN - nT - nC - nF;                        /* idle lanes */
#ifndef CRH_EXP_SCALAR_SCHED
				/* (wave-uniform by construction — lane 0 wrote them — but left as vector values: the decision below then compiles to exec-masked straight-line
				 * code. Declared uniform with readfirstlane — a scalar decision, a scalar step switch, 22 instead of 26 spilled VGPRs — it is 2-7 % SLOWER on every
				 * scene: the scalar form waits for the five LDS words before anything else and takes a chain of branches; profiles/r03o_ab_scalar_sched.log) */
				const int raysQ = wq[WQ_RAYS], hitsQ = wq[WQ_HITS], missQn = wq[WQ_MISSES], freeQ = wq[WQ_FREE];
				const uint32_t nextItem = (uint32_t)wq[WQ_NEXT_ITEM];
#else
				const int raysQ = __builtin_amdgcn_readfirstlane(wq[WQ_RAYS]), hitsQ = __builtin_amdgcn_readfirstlane(wq[WQ_HITS]),
						  missQn = __builtin_amdgcn_readfirstlane(wq[WQ_MISSES]), freeQ = __builtin_amdgcn_readfirstlane(wq[WQ_FREE]);
				const uint32_t nextItem = (uint32_t)__builtin_amdgcn_readfirstlane(wq[WQ_NEXT_ITEM]);
#endif
				CRH_LOCKSTEP();               /* every lane has read the fill levels before lane 0 updates them at the end of the step */
				const bool canGen = nextItem < nItems && freeQ >= 64;
				const int walkers = nN + nT + nC;
				enum { ST_NODE, ST_TRI, ST_CTRL, ST_SWAP, ST_GEN, ST_SHADE, ST_MISS, ST_END };
				int pick;
				if (hitsQ >= 64) pick = ST_SHADE;
				else if (missQn >= 64) pick = ST_MISS;
				else if (nF + nE >= K.swapMin && (nF > 0 || (nE > 0 && raysQ > 0))) pick = ST_SWAP;
				/* generate when lanes are out of rays — and whenever fewer than fillTo paths are in flight: a full table means full
				 * shading batches from the start of a job on */
				else if (canGen && raysQ < 64 && (((int)CRH_PATHS - freeQ) < K.fillTo || (nE + nF > 0 && raysQ < nE + nF))) pick = ST_GEN;
				else if (walkers > 0) {
					int best = nN * K.wNode;
					pick = ST_NODE;
					if (nT * K.wTri > best) { best = nT * K.wTri; pick = ST_TRI; }
					if (nC * K.wCtrl > best) { best = nC * K.wCtrl; pick = ST_CTRL; }
				}
				else if (nF > 0 || (nE > 0 && raysQ > 0)) pick = ST_SWAP;
				else if (hitsQ > 0) pick = ST_SHADE;
				else if (canGen) pick = ST_GEN;
				else if (missQn > 0) pick = ST_MISS;
				else pick = ST_END;
				if (pick == ST_END) break;
				uint32_t tk = 0;
				if constexpr (LEVEL >= 2) tk = CRH_TICK();
				switch (pick) {
					case ST_NODE: {          /* keep stepping while at least runNum/8 (half) of the lanes that started this run still want node steps */
						int now = nN;
						do {
#ifdef CRH_COOP_FETCH
							{
								const bool act = w.phase == PH_NODE;
								f4 l0, l1, r0, r1;
								fetch64(rec, lane, S.nodes, act ? 2u * w.node : CRH_NONE, l0, l1, r0, r1);
								if (act) stepNodeLoaded<true>(S, w, stk, cnt, port, l0, l1, r0, r1);
							}
#else
							if (w.phase == PH_NODE) stepNode<true>(S, w, stk, cnt, port);
#endif
							if constexpr (LEVEL >= 2) { if (lane == 0) { cnt.w_node += 1; cnt.u_node += (uint32_t)now; } }
							/* lanes that reached a leaf or an instance: serve them inside the run once enough of them wait (no scheduling
							 * round in between, and the node lanes they become again rejoin this run) */
							if ((int)__popcll(__ballot(w.phase == PH_TRI)) >= K.triInRun) {
								if (w.phase == PH_TRI) stepTri(S, w, stk, cnt, port);
							}
							if ((int)__popcll(__ballot(w.phase == PH_CTRL)) >= K.ctrlInRun) {
								if (w.phase == PH_CTRL) stepCtrl(S, w, stk, cnt, port);
							}
							now = __popcll(__ballot(w.phase == PH_NODE));
						} while (now * 8 >= nN * K.runNum);
						break;
					}
					case ST_TRI: {
						int now = nT;
						do {
							if (w.phase == PH_TRI) stepTri(S, w, stk, cnt, port);
							if constexpr (LEVEL >= 2) { if (lane == 0) { cnt.w_tri += 1; cnt.u_tri += (uint32_t)now; } }
							now = __popcll(__ballot(w.phase == PH_TRI));
						} while (now * 8 >= nT * K.runNum);
						break;
					}
					case ST_CTRL:
						if (ph == PH_CTRL) stepCtrl(S, w, stk, cnt, port);
						if (__ballot(ph == PH_NODE_SLOW)) { if (ph == PH_NODE_SLOW) stepNode<false>(S, w, stk, cnt, port); }   /* degenerate rays: rare */
						break;
					case ST_SWAP: {
						/* retire: a walk that ended leaves its result in the path's slot; the id goes on the hit or the miss stack */
						const bool fin = (ph == PH_SHADE);
						const bool finHit = fin && w.hit.inst >= 0, finMiss = fin && w.hit.inst < 0;
						const unsigned long long hm = __ballot(finHit), mm = __ballot(finMiss);
						uint32_t cls = 0;
						if (fin) {
							f4 *q = ptab + myPath * CRH_PATH_F4;
							q[4] = f4{w.hit.t, w.hit.u, w.hit.v, asF32((uint32_t)w.hit.slot)};
							if (finHit) {
								q[5].x = asF32((uint32_t)w.hit.inst);
								if (sorted) cls = (uint32_t)((volatile __attribute__((address_space(3))) uint8_t *)s_cls)[w.hit.inst];
								hits[(uint32_t)hitsQ + laneRank(hm)] = (uint16_t)(myPath | (cls << 8));
							} else {
								ids[CRH_IDS_MISSES + (uint32_t)missQn + laneRank(mm)] = (uint8_t)myPath;
							}
							w.phase = PH_IDLE;
						}
						if (sorted && hm) {        /* hits waiting per class (wave-uniform, lane 0 stores them) */
							uint32_t addLo = 0, addHi = 0;
#pragma unroll
							for (uint32_t b = 0; b < 8u; ++b) {
								const uint32_t nb = (uint32_t)__popcll(__ballot(finHit && cls == b));
								if (b < 4u) addLo += nb << (8u * b); else addHi += nb << (8u * (b - 4u));
							}
							if (lane == 0) { wq[WQ_CLS_LO] = wq[WQ_CLS_LO] + (int)addLo; wq[WQ_CLS_HI] = wq[WQ_CLS_HI] + (int)addHi; }
						}
						/* refill: idle lanes pop the top ray ids and start those walks */
						const bool idle = (w.phase == PH_IDLE);
						const unsigned long long em = __ballot(idle);
						const uint32_t er = laneRank(em);
						const int take = min(raysQ, (int)__popcll(em));
						if (idle && (int)er < take) {
							myPath = ids[CRH_IDS_RAYS + (uint32_t)(raysQ - take) + er];
							const f4 *q = ptab + myPath * CRH_PATH_F4;
							const f4 q0 = q[0], q1 = q[1];
							{ TablePort<SAMP> port{ptab + myPath * CRH_PATH_F4}; walkBegin(S, w, stk, v3{q0.x, q0.y, q0.z}, v3{q1.x, q1.y, q1.z}, cnt, port); }
						}
						if (lane == 0) { wq[WQ_HITS] = hitsQ + (int)__popcll(hm); wq[WQ_MISSES] = missQn + (int)__popcll(mm); wq[WQ_RAYS] = raysQ - take; }
						__threadfence_block();
						break;
					}
					case ST_GEN: {
						const uint32_t item = nextItem + lane;
						int x = 0, y = 0, pass = 0;
						const bool valid = item < nItems && decodeItem(J, item, x, y, pass);
						const unsigned long long vm = __ballot(valid);
						const int n = (int)__popcll(vm);
						if (valid) {
							const uint32_t rk = laneRank(vm);
							const uint32_t id = ids[CRH_IDS_FREE_END - (uint32_t)freeQ + rk];
							v3 o, d;
							PathRecT<RngT<SAMP>> r;
							beginPath(S, P, x, y, pass, o, d, r, cnt);
							putPathRay(ptab + id * CRH_PATH_F4, o, d, r, item);
							ids[CRH_IDS_RAYS + (uint32_t)raysQ + rk] = (uint8_t)id;
						}
						if (lane == 0) { wq[WQ_RAYS] = raysQ + n; wq[WQ_FREE] = freeQ - n; wq[WQ_NEXT_ITEM] = (int)(nextItem + 64u); }
						__threadfence_block();
						break;
					}
					case ST_MISS: {          /* pathtrace.c:39-42 for up to 64 rays that left the scene: background, then the sample is complete */
						const int n = min(missQn, 64);
						if ((int)lane < n) {
							const uint32_t id = ids[CRH_IDS_MISSES + (uint32_t)(missQn - n) + lane];
							const f4 *q = ptab + id * CRH_PATH_F4;
							const f4 q1 = q[1], q2 = q[2], q3 = q[3];
							v3 o{0.0f, 0.0f, 0.0f}, d{q1.x, q1.y, q1.z};
							PathRecT<RngT<SAMP>> r;
							r.wr = q2.x; r.wg = q2.y; r.wb = q2.z;
							r.fr = q3.x; r.fg = q3.y; r.fb = q3.z;
							r.rng.state = 0; r.depth = 0;
							const uint32_t item = asU32(q1.w);
							TravHit h;
							h.t = q[4].x; h.u = h.v = 0.0f; h.slot = -1; h.inst = -1;
							(void)shadeCore(S, P, o, d, h, r, cnt, stk);
							float *so = myStage + (size_t)item * 3; so[0] = r.fr; so[1] = r.fg; so[2] = r.fb;
							ids[CRH_IDS_FREE_END - 1u - (uint32_t)freeQ - lane] = (uint8_t)id;
						}
						if (lane == 0) { wq[WQ_MISSES] = missQn - n; wq[WQ_FREE] = freeQ + n; }
						__threadfence_block();
						break;
					}
					default: {   /* ST_SHADE: up to 64 surface hits of as few shade classes as fill the wave */
						/* Which hits: whole classes, largest first, while they fit into 64 lanes; if that leaves fewer than shadeMin lanes busy, the
						 * first hits of the next class as well. Hits of the other (small) classes wait for a later batch: the batch runs two or
						 * three surface-shader code paths instead of all of them. Scalar code on the per-class counts. */
						uint32_t clsLo = 0, clsHi = 0;
						int n = min(hitsQ, 64);
						if (sorted) {
						clsLo = (uint32_t)__builtin_amdgcn_readfirstlane(wq[WQ_CLS_LO]); clsHi = (uint32_t)__builtin_amdgcn_readfirstlane(wq[WQ_CLS_HI]);
						int c8[8];
#pragma unroll
						for (int b = 0; b < 8; ++b) c8[b] = (int)(((b < 4 ? clsLo : clsHi) >> (8 * (b & 3))) & 255u);
						uint32_t fullMask = 0;
						int partCls = -1, partN = 0;
						n = 0;
#pragma unroll
						for (int it = 0; it < 8; ++it) {
							int bc = 0, bb = -1;
#pragma unroll
							for (int b = 0; b < 8; ++b) if (!((fullMask >> b) & 1u) && c8[b] > bc) { bc = c8[b]; bb = b; }
							if (bb < 0 || n >= K.shadeMin || partCls >= 0) break;
							if (n + bc <= 64) { fullMask |= 1u << bb; n += bc; }
							else { partCls = bb; partN = 64 - n; n = 64; }
						}
						if (n < hitsQ) {
							/* bring the chosen hits to the top of the stack: every entry is read (up to three per lane), then written to its new
							 * place — the chosen ones in [hitsQ - n, hitsQ), the others below, both in their old order (LDS operations of a wave
							 * execute in program order, so all reads precede all writes) */
							uint32_t e[3];
							bool take[3], keep[3];
							uint32_t tr[3], kr[3];
							int tBase = 0, kBase = 0, pBase = 0;
#pragma unroll
							for (int p = 0; p < 3; ++p) {
								const uint32_t i = (uint32_t)p * 64u + lane;
								const bool valid = (int)i < hitsQ;
								e[p] = valid ? (uint32_t)hits[i] : 0u;
								const uint32_t ec = e[p] >> 8;
								const bool part = valid && (int)ec == partCls;
								const unsigned long long pm = __ballot(part);
								take[p] = valid && (((fullMask >> ec) & 1u) || (part && pBase + (int)laneRank(pm) < partN));
								pBase += (int)__popcll(pm);
								const unsigned long long tm = __ballot(take[p]);
								tr[p] = (uint32_t)tBase + laneRank(tm);
								keep[p] = valid && !take[p];
								const unsigned long long km = __ballot(keep[p]);
								kr[p] = (uint32_t)kBase + laneRank(km);
								tBase += (int)__popcll(tm);
								kBase += (int)__popcll(km);
							}
#pragma unroll
							for (int p = 0; p < 3; ++p) {
								if (take[p]) hits[(uint32_t)(hitsQ - n) + tr[p]] = (uint16_t)e[p];
								if (keep[p]) hits[kr[p]] = (uint16_t)e[p];
							}
							CRH_LOCKSTEP();        /* the batch below reads entries other lanes have just written */
						}
						/* the per-class counts after this batch */
#pragma unroll
						for (int b = 0; b < 8; ++b) {
							const uint32_t gone = ((fullMask >> b) & 1u) ? (uint32_t)c8[b] : (b == partCls ? (uint32_t)partN : 0u);
							if (b < 4) clsLo -= gone << (8 * b); else clsHi -= gone << (8 * (b - 4));
						}
						}
						if constexpr (LEVEL >= 2) { if (lane == 0) cnt.u_shade += (uint32_t)n; }
						bool cont = false, done = false;
						uint32_t id = 0;
						if ((int)lane < n) {
							id = (uint32_t)hits[(uint32_t)(hitsQ - n) + lane] & 255u;
							f4 *q = ptab + id * CRH_PATH_F4;
							const f4 q0 = q[0], q1 = q[1], q2 = q[2], q3 = q[3], q4 = q[4];
							v3 o{q0.x, q0.y, q0.z}, d{q1.x, q1.y, q1.z};
							PathRecT<RngT<SAMP>> r;
							r.wr = q2.x; r.wg = q2.y; r.wb = q2.z;
							r.fr = q3.x; r.fg = q3.y; r.fb = q3.z;
							r.rng.state = (uint64_t)asU32(q2.w) | ((uint64_t)asU32(q3.w) << 32);
							r.depth = (int)asU32(q0.w);
							const uint32_t item = asU32(q1.w);
							TravHit h;
							h.t = q4.x; h.u = q4.y; h.v = q4.z;
							h.slot = (int32_t)asU32(q4.w); h.inst = (int32_t)asU32(q[5].x);
							__builtin_assume(h.inst >= 0);
							cont = shadeCore(S, P, o, d, h, r, cnt, stk);
							done = !cont;
							if (cont) putPathRay(q, o, d, r, item);
							else {
								float *so = myStage + (size_t)item * 3; so[0] = r.fr; so[1] = r.fg; so[2] = r.fb;
							}
						}
						const unsigned long long cm = __ballot(cont), dm = __ballot(done);
						if (cont) ids[CRH_IDS_RAYS + (uint32_t)raysQ + laneRank(cm)] = (uint8_t)id;
						if (done) ids[CRH_IDS_FREE_END - 1u - (uint32_t)freeQ - laneRank(dm)] = (uint8_t)id;
						if (lane == 0) {
							wq[WQ_HITS] = hitsQ - n; wq[WQ_RAYS] = raysQ + (int)__popcll(cm); wq[WQ_FREE] = freeQ + (int)__popcll(dm);
							wq[WQ_CLS_LO] = (int)clsLo; wq[WQ_CLS_HI] = (int)clsHi;
						}
						__threadfence_block();
						break;
					}
				}
				if constexpr (LEVEL >= 2) {
					if (lane == 0) {
						const uint32_t dt = CRH_TICK() - tk;
						cnt.w_round += 1;
						if (pick == ST_NODE) { cnt.t_trav += dt; }
						else if (pick == ST_TRI) { cnt.t_setup += dt; }
						else if (pick == ST_CTRL) { cnt.w_ctrl += 1; cnt.w_setup += dt; cnt.u_ctrl += (uint32_t)nC; }
						else if (pick == ST_SWAP) { cnt.n_swap += 1; cnt.t_swap += dt; cnt.u_swap += (uint32_t)(nF + min(nE + nF, raysQ)); }
						else if (pick == ST_GEN || pick == ST_MISS) { cnt.n_gen += 1; cnt.t_gen += dt; }
						else { cnt.w_shade += 1; cnt.t_shade += dt; }
					}
				}
			}
			(void)validItems;
			__threadfence_block();                 /* the staged samples of all lanes are visible to the folding lanes */
			for (uint32_t pix = lane; pix < (uint32_t)(J.bw * J.bh); pix += 64u) foldBlockPixel(P, J, pix, myStage, fb);
			__threadfence_block();                 /* ... and read before the next chunk overwrites them */
		}
	}
	if (waveStats && lane == 0) {     /* debug: per-wave busy time (100 MHz ticks) and units processed */
		waveStats[2 * wave] = wall_clock64() - tStart;
		waveStats[2 * wave + 1] = unitsDone;
	}
	/* one atomic per wave and counter */
	const bool lead = (lane == 0);
	uint32_t v;
	v = waveSum(cnt.paths); if (lead && v) atomicAdd(&counters[0], (unsigned long long)v);
	v = waveSum(cnt.rays); if (lead && v) atomicAdd(&counters[1], (unsigned long long)v);
	if constexpr (LEVEL >= 2) {
		v = waveSum(cnt.node_tests); if (lead && v) atomicAdd(&counters[2], (unsigned long long)v);
		v = waveSum(cnt.tri_tests); if (lead && v) atomicAdd(&counters[3], (unsigned long long)v);
		v = waveSum(cnt.inst_visits); if (lead && v) atomicAdd(&counters[4], (unsigned long long)v);
		v = waveSum(cnt.inst_hits); if (lead && v) atomicAdd(&counters[5], (unsigned long long)v);
		v = waveSum(cnt.sphere_tests); if (lead && v) atomicAdd(&counters[6], (unsigned long long)v);
		v = waveSum(cnt.tex_fetches); if (lead && v) atomicAdd(&counters[7], (unsigned long long)v);
		if (lead) {   /* debug phase clocks: one sample per wave (lane 0) */
			atomicAdd(&counters[8], (unsigned long long)cnt.t_setup);
			atomicAdd(&counters[9], (unsigned long long)cnt.t_trav);
			atomicAdd(&counters[10], (unsigned long long)cnt.t_shade);
		}
		v = waveSum(cnt.w_node); if (lead && v) atomicAdd(&counters[11], (unsigned long long)v);
		v = waveSum(cnt.w_tri); if (lead && v) atomicAdd(&counters[12], (unsigned long long)v);
		v = waveSum(cnt.w_ctrl); if (lead && v) atomicAdd(&counters[13], (unsigned long long)v);
		v = waveSum(cnt.w_round); if (lead && v) atomicAdd(&counters[14], (unsigned long long)v);
		v = waveSum(cnt.w_shade); if (lead && v) atomicAdd(&counters[15], (unsigned long long)v);
		v = waveSum(cnt.w_setup); if (lead && v) atomicAdd(&counters[16], (unsigned long long)v);
		v = waveSum(cnt.u_node); if (lead && v) atomicAdd(&counters[17], (unsigned long long)v);
		v = waveSum(cnt.u_shade); if (lead && v) atomicAdd(&counters[18], (unsigned long long)v);
		v = waveSum(cnt.t_swap); if (lead && v) atomicAdd(&counters[19], (unsigned long long)v);
		v = waveSum(cnt.t_gen); if (lead && v) atomicAdd(&counters[20], (unsigned long long)v);
		v = waveSum(cnt.n_swap); if (lead && v) atomicAdd(&counters[21], (unsigned long long)v);
		v = waveSum(cnt.n_gen); if (lead && v) atomicAdd(&counters[22], (unsigned long long)v);
		v = waveSum(cnt.u_swap); if (lead && v) atomicAdd(&counters[23], (unsigned long long)v);
		v = waveSum(cnt.u_tri); if (lead && v) atomicAdd(&counters[24], (unsigned long long)v);
		v = waveSum(cnt.u_ctrl); if (lead && v) atomicAdd(&counters[25], (unsigned long long)v);
	}
}

#include "pathtrace_roll.h"          /* k_pathtrace_roll: the same machine with rolling work units — the DEFAULT form since the end of round 3 (CRH_KERNEL_ROLL) */
#define CRH_EXP_ROLLING_UNITS 1      /* (the name the form was developed under, kept for the tools that test for it) */

/* ================================================================================================================================
 * k_pathtrace_wg — the WORKGROUP-cooperative form of the machine above (CRH_OPT_KERNEL = CRH_KERNEL_WG).
 *
 * Why: in k_pathtrace one wave alternates between walking and shading, so the ~30 VGPRs of walk state stay live across the
 * shading code (which alone wants ~140): at the 128-register budget of 4 waves / SIMD that is ~90 spilled VGPRs and ~120 B of
 * scratch traffic per ray. Here the four waves of a workgroup share ONE path table (1024 records) and ONE set of id stacks, and
 * a wave only ever shades / generates / evaluates misses when it holds NO walk: walk state and shading state are never live
 * at the same program point, so neither is spilled. Roles are dynamic:
 *   - a wave without live walks (top of the loop) takes the job with the most pending work: SHADE (>= 64 hits queued), MISS,
 *     GEN (table below its fill level), or WALK (pop up to 64 ray ids and walk them);
 *   - a walking wave retires finished walks and refills idle lanes from the shared ray stack for as long as rays are there; it
 *     returns to the top only when it has drained — because the ray stack is empty, or because it took the workgroup's DRAIN
 *     token (backlog of hits + misses >= drainAt and fewer than maxDrainers waves already draining): that is how walkers
 *     become servers when shading falls behind;
 *   - a wave that just served lingers (sleeps, up to `linger` polls) for the next full batch before it walks again: that is how
 *     a server stays a server while the workload keeps it busy, without oscillating.
 * Stacks are LIFO, mutated only under the workgroup's LDS spin lock (critical sections touch LDS only; path records are
 * written before the lock is taken and published by the release fence). Every path's own sequence of operations is the same
 * as in k_pathtrace, hence the same frame bit for bit. A watchdog (wall clock) aborts the dispatch instead of hanging.
 * ================================================================================================================================ */
#define CRH_WG_PATHS 1024u
#define CRH_WG_STACK_LDS 22          /* (22 + 13 park) x 1 KB + 2 x 2 KB id arrays + control words <= 40 KB: 4 workgroups per CU */
struct SchedWg { int wNode, wTri, wCtrl, swapMin, fillTo, runNum, triInRun, ctrlInRun, linger, drainAt, maxDrainers, partialMin, walkMin; };

/* traversal stack of the workgroup kernel: LDS first, deeper entries in a per-lane column of a global array (never scratch) */
struct WgStack {
	lds_u32 *lds, *parkp;
	uint32_t *ovf;       /* wave-uniform: &ovfAll[wave * OVF * 64]; entry i of lane l at ovf[i * 64 + l] */
	uint32_t lane;
	__device__ __forceinline__ void park(int i, uint32_t v) { parkp[i * CRH_BLOCK] = v; }
	__device__ __forceinline__ uint32_t unpark(int i) { return parkp[i * CRH_BLOCK]; }
	__device__ __forceinline__ void push(uint32_t i, uint32_t v) {
		if (__builtin_expect(i < CRH_WG_STACK_LDS, 1)) lds[i * CRH_BLOCK] = v;
		else ovf[(i - CRH_WG_STACK_LDS) * 64u + lane] = v;
	}
	__device__ __forceinline__ uint32_t pop(uint32_t i) {
		if (__builtin_expect(i < CRH_WG_STACK_LDS, 1)) return lds[i * CRH_BLOCK];
		return ovf[(i - CRH_WG_STACK_LDS) * 64u + lane];
	}
};
#define CRH_WG_OVF (134 - CRH_WG_STACK_LDS)
static_assert(CRH_WG_OVF * 64u <= CRH_OVF_WORDS_PER_WAVE && CRH_STACK_OVF * 64u <= CRH_OVF_WORDS_PER_WAVE, "overflow columns fit the per-wave block");

enum { CT_LOCK, CT_RAYS, CT_HITS, CT_MISSES, CT_FREE, CT_NEXT, CT_DRAINERS, CT_ABORT, CT_UNIT, CT_WORDS };
typedef volatile __attribute__((address_space(3))) int wg_int;
typedef volatile __attribute__((address_space(3))) uint16_t wg_u16;

/* spin lock of the workgroup's queues; false = the dispatch is being aborted (watchdog) */
__device__ __forceinline__ bool wgLock(int *lockWord, wg_int *ctl, uint32_t lane, unsigned int *errFlag) {
	if (lane == 0) {
		uint32_t spins = 0;
		while (atomicCAS(lockWord, 0, 1) != 0) {
			__builtin_amdgcn_s_sleep(2);
			if (++spins > (1u << 24) || ctl[CT_ABORT]) { ctl[CT_ABORT] = 1; atomicOr(errFlag, 1u); break; }
		}
	}
	CRH_LOCKSTEP();          /* the other lanes wait for lane 0's spin */
	__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
	return ctl[CT_ABORT] == 0;
}
__device__ __forceinline__ void wgUnlock(int *lockWord, uint32_t lane) {
	CRH_LOCKSTEP();          /* every lane is through the critical section before lane 0 opens the lock */
	__builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");          /* this wave's id / counter writes are in LDS before the lock opens */
	if (lane == 0) __hip_atomic_store(lockWord, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}

template <int LEVEL, bool PROG, int SAMP>
__global__ __launch_bounds__(CRH_BLOCK, 4) void k_pathtrace_wg(const DScene Sarg, const crh_render_params P, const BlockQueue Q, float *fb,
																unsigned long long *counters, float *stage, int chunk, const SchedWg K, float *queues,
																uint32_t *ovfAll, unsigned int *errFlag) {
	__shared__ uint32_t s_stack[CRH_WG_STACK_LDS * CRH_BLOCK];
	__shared__ uint32_t s_park[CRH_PARK_SLOTS * CRH_BLOCK];
	__shared__ uint16_t s_idsA[CRH_WG_PATHS];      /* rays grow up from 0, hits grow down from the end */
	__shared__ uint16_t s_idsB[CRH_WG_PATHS];      /* misses grow up from 0, free slots grow down from the end */
	__shared__ int s_ctl[CT_WORDS];
	static_assert((CRH_WG_STACK_LDS + CRH_PARK_SLOTS) * CRH_BLOCK * 4 + 2 * CRH_WG_PATHS * 2 + CT_WORDS * 4 + 512 <= 40960, "4 workgroups per CU share 160 KB of LDS (incl. powf's tables)");
	const DScene S = globalize(Sarg);
	CRH_EM_POW_TABLES_INIT();
	const uint32_t lane = threadIdx.x & 63u;
	const uint32_t wave = (blockIdx.x * CRH_BLOCK + threadIdx.x) >> 6;
	WgStack stk;
	stk.lds = (lds_u32 *)&s_stack[threadIdx.x];
	stk.parkp = (lds_u32 *)&s_park[threadIdx.x];
	stk.ovf = (uint32_t *)(__attribute__((address_space(1))) uint32_t *)(ovfAll + (size_t)__builtin_amdgcn_readfirstlane(wave) * CRH_OVF_WORDS_PER_WAVE);
	stk.lane = lane;
	CountersT<LEVEL, PROG> cnt;
	memset(&cnt, 0, sizeof(cnt));
	float *const myStage = stage + (size_t)blockIdx.x * ((size_t)Q.bw * Q.bh * chunk * 3);
	const int passEnd = P.first_pass + P.pass_count;
	f4 *const ptab = (f4 *)(queues + (size_t)blockIdx.x * (CRH_WG_PATHS * CRH_PATH_F4 * 4u));
	wg_int *const ctl = (wg_int *)s_ctl;
	wg_u16 *const idsA = (wg_u16 *)s_idsA;
	wg_u16 *const idsB = (wg_u16 *)s_idsB;
	int *const lockWord = &s_ctl[CT_LOCK];
	const int NP = (int)CRH_WG_PATHS;
	if (threadIdx.x == 0) s_ctl[CT_ABORT] = 0;
	for (;;) {
		if (threadIdx.x == 0) {
			s_ctl[CT_UNIT] = (int)atomicAdd((uint32_t *)(__attribute__((address_space(1))) uint32_t *)Q.counter, 1u);
			s_ctl[CT_LOCK] = 0; s_ctl[CT_DRAINERS] = 0;
		}
		__syncthreads();
		const uint32_t unit = (uint32_t)ctl[CT_UNIT];
		if (unit >= Q.total || ctl[CT_ABORT]) break;
		uint32_t lo = 0, hi = Q.ntiles;           /* largest t with start[t] <= unit */
		while (hi - lo > 1) { const uint32_t mid = (lo + hi) >> 1; if (asGlobal(Q.start)[mid] <= unit) lo = mid; else hi = mid; }
		const crh_tile t = asGlobal(Q.tiles)[lo];
		const uint32_t local = unit - asGlobal(Q.start)[lo];
		const int ubw = lo >= Q.firstTiny ? Q.tbw : lo >= Q.firstSmall ? Q.sbw : Q.bw, ubh = lo >= Q.firstTiny ? Q.tbh : lo >= Q.firstSmall ? Q.sbh : Q.bh;
		const uint32_t nbx = (uint32_t)(t.x1 - t.x0 + ubw - 1) / (uint32_t)ubw;
		BlockJob J;
		J.bw = ubw; J.bh = ubh;
		J.x0 = t.x0 + (int)(local % nbx) * ubw;
		J.y0 = t.y0 + (int)(local / nbx) * ubh;
		J.w = min(ubw, t.x1 - J.x0);
		J.h = min(ubh, t.y1 - J.y0);
		for (int c0 = P.first_pass; c0 < passEnd; c0 += chunk) {
			J.passBegin = c0;
			J.passCount = min(chunk, passEnd - c0);
			const uint32_t nItems = (uint32_t)(J.bw * J.bh * J.passCount);       /* incl. the padding of ragged tile edges */
			if (threadIdx.x == 0) { s_ctl[CT_RAYS] = 0; s_ctl[CT_HITS] = 0; s_ctl[CT_MISSES] = 0; s_ctl[CT_FREE] = NP; s_ctl[CT_NEXT] = 0; }
			for (uint32_t i = threadIdx.x; i < CRH_WG_PATHS; i += CRH_BLOCK) s_idsB[i] = (uint16_t)i;   /* all slots free */
			__syncthreads();
			int idlePolls = K.linger;                     /* a wave that has not served yet does not linger */
			uint32_t waitStart = 0;
			bool waiting = false;
			const uint32_t chunkStart = CRH_TICK();
			for (;;) {   /* ---- top of the machine: this wave holds no walk ---- */
				if (ctl[CT_ABORT]) break;
				if (CRH_TICK() - chunkStart > 3000000000u) { if (lane == 0) { ctl[CT_ABORT] = 1; atomicOr(errFlag, 4u); } break; }   /* 30 s in one chunk: watchdog */
				const int nH = ctl[CT_HITS], nM = ctl[CT_MISSES], nR = ctl[CT_RAYS];
				const uint32_t nextItem = (uint32_t)ctl[CT_NEXT];
				const int nFree = ctl[CT_FREE];
				CRH_LOCKSTEP();               /* one consistent reading of the control words for the whole wave */
				const bool canGen = nextItem < nItems && nFree >= 64;
				enum { JB_SHADE, JB_MISS, JB_GEN, JB_WALK, JB_WAIT };
				int job = JB_WAIT;
				if (nH >= 64) job = JB_SHADE;
				else if (nM >= 64) job = JB_MISS;
				else if (canGen && nR < 64 && (NP - nFree) < K.fillTo) job = JB_GEN;
				else if (nR == 0 && nH >= K.partialMin) job = JB_SHADE;        /* walkers are out of rays: a partial batch now beats a full one later */
				else if (idlePolls >= K.linger) {          /* not (or no longer) waiting for a full batch: take what is there */
					if (nR >= K.walkMin) job = JB_WALK;
					else if (nH >= K.partialMin) job = JB_SHADE;
					else if (nR > 0) job = JB_WALK;
					else if (nH > 0) job = JB_SHADE;
					else if (nM > 0) job = JB_MISS;
					else if (canGen) job = JB_GEN;
				}
				if (job == JB_WAIT) {
					/* nothing queued at all: finished, or the other waves still hold the remaining paths. The unlocked test is sound
					 * (GEN lowers CT_FREE before it raises CT_NEXT, and CT_NEXT was read first); the locked one is belt and braces. */
					if (nFree == NP && nextItem >= nItems) {
						if (!wgLock(lockWord, ctl, lane, errFlag)) break;
						const bool finished = ctl[CT_FREE] == NP && (uint32_t)ctl[CT_NEXT] >= nItems;
						wgUnlock(lockWord, lane);
						if (finished) break;
					}
					++idlePolls;
					const uint32_t now = CRH_TICK();
					if (!waiting) { waiting = true; waitStart = now; }
					else if (now - waitStart > 400000000u) { if (lane == 0) { ctl[CT_ABORT] = 1; atomicOr(errFlag, 2u); } }    /* 4 s without work: watchdog */
					__builtin_amdgcn_s_sleep(16);
					continue;
				}
				waiting = false;
				if (job == JB_SHADE) {           /* pathtrace.c:44-57 for up to 64 surface hits */
					if (!wgLock(lockWord, ctl, lane, errFlag)) break;
					const int hq = ctl[CT_HITS];
					const int n = min(hq, 64);
					uint32_t id = 0;
					if ((int)lane < n) id = idsA[NP - hq + (int)lane];
					CRH_LOCKSTEP();
					if (lane == 0) ctl[CT_HITS] = hq - n;
					wgUnlock(lockWord, lane);
					if (n == 0) continue;
					bool cont = false, done = false;
					if ((int)lane < n) {
						f4 *q = ptab + id * CRH_PATH_F4;
						const f4 q0 = q[0], q1 = q[1], q2 = q[2], q3 = q[3], q4 = q[4];
						v3 o{q0.x, q0.y, q0.z}, d{q1.x, q1.y, q1.z};
						PathRecT<RngT<SAMP>> r;
						r.wr = q2.x; r.wg = q2.y; r.wb = q2.z;
						r.fr = q3.x; r.fg = q3.y; r.fb = q3.z;
						r.rng.state = (uint64_t)asU32(q2.w) | ((uint64_t)asU32(q3.w) << 32);
						r.depth = (int)asU32(q0.w);
						const uint32_t item = asU32(q1.w);
						TravHit h;
						h.t = q4.x; h.u = q4.y; h.v = q4.z;
						h.slot = (int32_t)asU32(q4.w); h.inst = (int32_t)asU32(q[5].x);
						__builtin_assume(h.inst >= 0);
						cont = shadeCore(S, P, o, d, h, r, cnt);
						done = !cont;
						if (cont) putPathRay(q, o, d, r, item);
						else { float *so = myStage + (size_t)item * 3; so[0] = r.fr; so[1] = r.fg; so[2] = r.fb; }
					}
					const unsigned long long cm = __ballot(cont), dm = __ballot(done);
					__builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");      /* records and samples are written before their ids are published */
					if (!wgLock(lockWord, ctl, lane, errFlag)) break;
					const int rq = ctl[CT_RAYS], fq = ctl[CT_FREE];
					if (cont) idsA[rq + (int)laneRank(cm)] = (uint16_t)id;
					if (done) idsB[NP - 1 - fq - (int)laneRank(dm)] = (uint16_t)id;
					CRH_LOCKSTEP();
					if (lane == 0) { ctl[CT_RAYS] = rq + (int)__popcll(cm); ctl[CT_FREE] = fq + (int)__popcll(dm); }
					wgUnlock(lockWord, lane);
					idlePolls = 0;
					continue;
				}
				if (job == JB_MISS) {            /* pathtrace.c:39-42: background for up to 64 rays that left the scene; the sample is complete */
					if (!wgLock(lockWord, ctl, lane, errFlag)) break;
					const int mq = ctl[CT_MISSES];
					const int n = min(mq, 64);
					uint32_t id = 0;
					if ((int)lane < n) id = idsB[mq - n + (int)lane];
					CRH_LOCKSTEP();
					if (lane == 0) ctl[CT_MISSES] = mq - n;
					wgUnlock(lockWord, lane);
					if (n == 0) continue;
					if ((int)lane < n) {
						const f4 *q = ptab + id * CRH_PATH_F4;
						const f4 q1 = q[1], q2 = q[2], q3 = q[3];
						v3 o{0.0f, 0.0f, 0.0f}, d{q1.x, q1.y, q1.z};
						PathRecT<RngT<SAMP>> r;
						r.wr = q2.x; r.wg = q2.y; r.wb = q2.z;
						r.fr = q3.x; r.fg = q3.y; r.fb = q3.z;
						r.rng.state = 0; r.depth = 0;
						const uint32_t item = asU32(q1.w);
						TravHit h;
						h.t = q[4].x; h.u = h.v = 0.0f; h.slot = -1; h.inst = -1;
						(void)shadeCore(S, P, o, d, h, r, cnt);
						float *so = myStage + (size_t)item * 3; so[0] = r.fr; so[1] = r.fg; so[2] = r.fb;
					}
					__builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
					if (!wgLock(lockWord, ctl, lane, errFlag)) break;
					const int fq = ctl[CT_FREE];
					if ((int)lane < n) idsB[NP - 1 - fq - (int)lane] = (uint16_t)id;
					CRH_LOCKSTEP();
					if (lane == 0) ctl[CT_FREE] = fq + n;
					wgUnlock(lockWord, lane);
					idlePolls = 0;
					continue;
				}
				if (job == JB_GEN) {             /* renderer.c:280-284 for the next 64 items */
					if (!wgLock(lockWord, ctl, lane, errFlag)) break;
					const uint32_t it0 = (uint32_t)ctl[CT_NEXT];
					const int fq = ctl[CT_FREE];
					const bool ok = it0 < nItems && fq >= 64;
					const uint32_t item = it0 + lane;
					int x = 0, y = 0, pass = 0;
					const bool valid = ok && item < nItems && decodeItem(J, item, x, y, pass);
					const unsigned long long vm = __ballot(valid);
					const int n = (int)__popcll(vm);
					uint32_t id = 0;
					if (valid) id = idsB[NP - fq + (int)laneRank(vm)];
					CRH_LOCKSTEP();
					if (lane == 0 && ok) { ctl[CT_FREE] = fq - n; ctl[CT_NEXT] = (int)(it0 + 64u); }
					wgUnlock(lockWord, lane);
					if (!ok) continue;
					if (valid) {
						v3 o, d;
						PathRecT<RngT<SAMP>> r;
						beginPath(S, P, x, y, pass, o, d, r, cnt);
						putPathRay(ptab + id * CRH_PATH_F4, o, d, r, item);
					}
					__builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
					if (n > 0) {
						if (!wgLock(lockWord, ctl, lane, errFlag)) break;
						const int rq = ctl[CT_RAYS];
						if (valid) idsA[rq + (int)laneRank(vm)] = (uint16_t)id;
						CRH_LOCKSTEP();
						if (lane == 0) ctl[CT_RAYS] = rq + n;
						wgUnlock(lockWord, lane);
					}
					continue;
				}
				/* ---- JB_WALK: this wave walks rays until it has drained ---- */
				{
					Walk w;
					memset(&w, 0, sizeof(w));
					w.phase = PH_IDLE;
					uint32_t myPath = 0;
					bool draining = false;
					for (;;) {
						const uint32_t ph = w.phase;
						TablePort<SAMP> port{ptab + myPath * CRH_PATH_F4};
						const int nN = __popcll(__ballot(ph == PH_NODE)), nT = __popcll(__ballot(ph == PH_TRI)), nC = __popcll(__ballot(ph == PH_CTRL || ph == PH_NODE_SLOW));
						const int nF = __popcll(__ballot(ph == PH_SHADE));
						const int nE = 64 - nN - nT - nC - nF;
						const int walkers = nN + nT + nC;
						const int raysQ = ctl[CT_RAYS];
						if (walkers == 0 || (nF + nE >= K.swapMin && (nF > 0 || (raysQ > 0 && !draining)))) {
							/* SWAP: finished walks leave their result in the path's record ... */
							const bool fin = (ph == PH_SHADE);
							const bool finHit = fin && w.hit.inst >= 0, finMiss = fin && w.hit.inst < 0;
							const unsigned long long hm = __ballot(finHit), mm = __ballot(finMiss);
							if (fin) {
								f4 *q = ptab + myPath * CRH_PATH_F4;
								q[4] = f4{w.hit.t, w.hit.u, w.hit.v, asF32((uint32_t)w.hit.slot)};
								if (finHit) q[5].x = asF32((uint32_t)w.hit.inst);
								w.phase = PH_IDLE;
							}
							__builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
							if (!wgLock(lockWord, ctl, lane, errFlag)) break;
							const int rq = ctl[CT_RAYS], hq = ctl[CT_HITS], mq = ctl[CT_MISSES];
							/* ... their ids go on the hit / miss stacks ... */
							if (finHit) idsA[NP - 1 - hq - (int)laneRank(hm)] = (uint16_t)myPath;
							if (finMiss) idsB[mq + (int)laneRank(mm)] = (uint16_t)myPath;
							const int hq2 = hq + (int)__popcll(hm), mq2 = mq + (int)__popcll(mm);
							/* ... the drain token: shading has fallen behind -> this wave stops taking rays and becomes a server once its walks end */
							int drainers = ctl[CT_DRAINERS];
							if (!draining && walkers + nF > 0 && drainers < K.maxDrainers && hq2 + mq2 >= K.drainAt) { draining = true; ++drainers; }    /* (a wave with nothing in flight has nothing to drain: it takes rays, so every WALK job makes progress) */
							else if (draining && hq2 + mq2 < 64) { draining = false; --drainers; }
							/* ... and idle lanes pop ray ids */
							const bool idle = (w.phase == PH_IDLE);
							const unsigned long long em = __ballot(idle);
							const uint32_t er = laneRank(em);
							const int take = draining ? 0 : min(rq, (int)__popcll(em));
							const bool got = idle && (int)er < take;
							if (got) myPath = idsA[rq - take + (int)er];
							CRH_LOCKSTEP();
							if (lane == 0) { ctl[CT_HITS] = hq2; ctl[CT_MISSES] = mq2; ctl[CT_RAYS] = rq - take; ctl[CT_DRAINERS] = drainers; }
							wgUnlock(lockWord, lane);
							if (got) {
								const f4 *q = ptab + myPath * CRH_PATH_F4;
								const f4 q0 = q[0], q1 = q[1];
								{ TablePort<SAMP> port{ptab + myPath * CRH_PATH_F4}; walkBegin(S, w, stk, v3{q0.x, q0.y, q0.z}, v3{q1.x, q1.y, q1.z}, cnt, port); }
							}
							if (__ballot(w.phase != PH_IDLE) == 0ull) break;        /* drained: back to the top */
							continue;
						}
						int pick = 0, best = nN * K.wNode;
						if (nT * K.wTri > best) { best = nT * K.wTri; pick = 1; }
						if (nC * K.wCtrl > best) { best = nC * K.wCtrl; pick = 2; }
						if (pick == 0) {          /* node run, with leaf / instance steps served in place (see k_pathtrace) */
							int now = nN;
							do {
								if (w.phase == PH_NODE) stepNode<true>(S, w, stk, cnt, port);
								if ((int)__popcll(__ballot(w.phase == PH_TRI)) >= K.triInRun) { if (w.phase == PH_TRI) stepTri(S, w, stk, cnt, port); }
								if ((int)__popcll(__ballot(w.phase == PH_CTRL)) >= K.ctrlInRun) { if (w.phase == PH_CTRL) stepCtrl(S, w, stk, cnt, port); }
								now = __popcll(__ballot(w.phase == PH_NODE));
							} while (now * 8 >= nN * K.runNum);
						} else if (pick == 1) {
							int now = nT;
							do {
								if (w.phase == PH_TRI) stepTri(S, w, stk, cnt, port);
								now = __popcll(__ballot(w.phase == PH_TRI));
							} while (now * 8 >= nT * K.runNum);
						} else {
							if (ph == PH_CTRL) stepCtrl(S, w, stk, cnt, port);
							if (__ballot(ph == PH_NODE_SLOW)) { if (ph == PH_NODE_SLOW) stepNode<false>(S, w, stk, cnt, port); }   /* degenerate rays: rare */
						}
					}
					if (draining) {
						if (!wgLock(lockWord, ctl, lane, errFlag)) break;
						if (lane == 0) ctl[CT_DRAINERS] = ctl[CT_DRAINERS] - 1;
						wgUnlock(lockWord, lane);
					}
					idlePolls = K.linger;          /* a drained walker takes whatever is there */
				}
			}
			__syncthreads();                       /* every sample of the chunk is staged (the barrier is a workgroup-scope fence) */
			if (!ctl[CT_ABORT])
				for (uint32_t pix = threadIdx.x; pix < (uint32_t)(J.bw * J.bh); pix += CRH_BLOCK) foldBlockPixel(P, J, pix, myStage, fb);
			__syncthreads();                       /* ... and folded before the next chunk overwrites the slab */
		}
		if (ctl[CT_ABORT]) break;
		__syncthreads();                           /* everyone has read CT_UNIT before thread 0 replaces it */
	}
	const bool lead = (lane == 0);
	uint32_t v;
	v = waveSum(cnt.paths); if (lead && v) atomicAdd(&counters[0], (unsigned long long)v);
	v = waveSum(cnt.rays); if (lead && v) atomicAdd(&counters[1], (unsigned long long)v);
	if constexpr (LEVEL >= 2) {
		v = waveSum(cnt.node_tests); if (lead && v) atomicAdd(&counters[2], (unsigned long long)v);
		v = waveSum(cnt.tri_tests); if (lead && v) atomicAdd(&counters[3], (unsigned long long)v);
		v = waveSum(cnt.inst_visits); if (lead && v) atomicAdd(&counters[4], (unsigned long long)v);
		v = waveSum(cnt.inst_hits); if (lead && v) atomicAdd(&counters[5], (unsigned long long)v);
		v = waveSum(cnt.sphere_tests); if (lead && v) atomicAdd(&counters[6], (unsigned long long)v);
		v = waveSum(cnt.tex_fetches); if (lead && v) atomicAdd(&counters[7], (unsigned long long)v);
	}
}

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
	hipStream_t stream = nullptr;
	bool ownStream = false;
	int cuCount = 0;
	int blocksPerCU = 4;
	int counterLevel = 2;
	int passChunk = 64;
	int unitItems = 2048;
	int unitsPerWave = 8;
	Sched sched = {70, 160, 120, 16, 160, 4, 12, 12, 48, 0};
	int kernel = CRH_KERNEL_ROLL;            /* CRH_OPT_KERNEL */
	SchedWg schedWg = {70, 160, 120, 16, 768, 4, 12, 12, 8, 192, 1, 16, 32};
	uint32_t *dOvf = nullptr;                /* workgroup kernel: traversal-stack overflow columns */
	size_t ovfWords = 0;
	unsigned int *dErr = nullptr;            /* workgroup kernel: watchdog flag */
	std::vector<int> preloaded;              /* kernel instantiations whose code object is loaded (variantKey) */
	float *dGather = nullptr;                /* crh_frames_gather: this GPU's strips packed (senders) / every sender's slab (GPU 0) */
	size_t gatherFloats = 0;
	uint8_t *dSrgb = nullptr;                /* crh_framebuffer_to_srgb8: the 8-bit frame on the device (grown on demand, kept) */
	size_t srgbBytes = 0;
	bool traceExactSlabs = false;            /* CRH_OPT_TRACE_SLABS: crh_trace_rays walks degenerate rays like the render kernels do (exact slabs) instead of like the reference (NaN arithmetic) */
	bool wgSinceCheck = false;               /* a workgroup-kernel launch has happened since the flag was last read (the default kernel never writes it) */
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
	uint32_t lastGrid = 0;
	float *dStage = nullptr;
	size_t stageFloats = 0;
	bool haveScene = false;
	bool hasPrograms = true;     /* the compiled scene contains node programs -> kernel variant with runProgram() */
	bool hasVolumes = false;     /* walks draw from the path's sampler: crh_trace_rays (caller rays, no path) refuses such scenes */
	DScene d;                              /* device pointers */
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

/* the workgroup kernel aborts instead of hanging: report it (call with the stream drained) */
static int checkWatchdog(crh_ctx *c) {
	if (!c->wgSinceCheck) return CRH_OK;          /* no blocking device-to-host copy per download / synchronize for the default kernel */
	c->wgSinceCheck = false;
	unsigned int err = 0;
	HIP_TRY(hipMemcpy(&err, c->dErr, sizeof(err), hipMemcpyDeviceToHost));
	if (err) {
		HIP_TRY(hipMemset(c->dErr, 0, sizeof(err)));
		return fail(CRH_ERR_HIP, "k_pathtrace_wg: watchdog abort (flag " + std::to_string(err) + "): the frame is incomplete");
	}
	return CRH_OK;
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

/* Launch the instantiation the context's options select (counter level, register budget, rare features, sampler, kernel form). */
static hipError_t launchPathtrace(crh_ctx *c, uint32_t grid, const crh_render_params *P, const BlockQueue &Q, float *dev_fb, int chunk) {
	const bool wg = c->kernel == CRH_KERNEL_WG;
	if (wg) c->wgSinceCheck = true;
#define CRH_LAUNCH(LEVEL, WPS, PROG, SAMP) hipLaunchKernelGGL((k_pathtrace<LEVEL, WPS, PROG, SAMP>), dim3(grid), dim3(CRH_BLOCK), 0, c->stream, c->d, *P, Q, dev_fb, \
												  c->dCounters, c->dStage, chunk, c->dWaveStats, c->sched, c->dQueues, c->dOvf)
#define CRH_LAUNCH2(LEVEL, WPS) do { if (c->hasPrograms) CRH_LAUNCH(LEVEL, WPS, true, 0); else CRH_LAUNCH(LEVEL, WPS, false, 0); } while (0)
#define CRH_LAUNCH_WG(LEVEL, PROG, SAMP) hipLaunchKernelGGL((k_pathtrace_wg<LEVEL, PROG, SAMP>), dim3(grid), dim3(CRH_BLOCK), 0, c->stream, c->d, *P, Q, dev_fb, \
													  c->dCounters, c->dStage, chunk, c->schedWg, c->dQueues, c->dOvf, c->dErr)
#define CRH_LAUNCH_ROLL(LEVEL, PROG, SAMP) hipLaunchKernelGGL((k_pathtrace_roll<LEVEL, 4, PROG, SAMP>), dim3(grid), dim3(CRH_BLOCK), 0, c->stream, c->d, *P, Q, dev_fb, \
														  c->dCounters, c->dStage, chunk, c->dWaveStats, c->sched, c->dQueues, c->dOvf)
	if (c->kernel == CRH_KERNEL_ROLL) {
#ifdef CRH_DEV_ONLY_BENCH_VARIANT
#if defined(CRH_DEV_ONLY_LEVEL2)              /* the counting instantiation (tools/emu_sched_stats.py) */
		CRH_LAUNCH_ROLL(2, true, 0);
#elif defined(CRH_DEV_ONLY_PROG)
		CRH_LAUNCH_ROLL(1, true, 0);
#else
		CRH_LAUNCH_ROLL(1, false, 0);
#endif
#else
		const bool halton = c->sampler == CRH_SAMPLER_HALTON;
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
#ifdef CRH_DEV_ONLY_BENCH_VARIANT                 /* development builds (tools/kernel_regs.py): one instantiation compiles in seconds */
#ifdef CRH_DEV_ONLY_PROG
	if (wg) CRH_LAUNCH_WG(1, true, 0); else CRH_LAUNCH(1, 4, true, 0);
#else
	if (wg) CRH_LAUNCH_WG(1, false, 0); else CRH_LAUNCH(1, 4, false, 0);
#endif
#else
	if (wg) {
		const bool halton = c->sampler == CRH_SAMPLER_HALTON;
		if (c->counterLevel >= 2) {
			if (c->hasPrograms) { if (halton) CRH_LAUNCH_WG(2, true, 1); else CRH_LAUNCH_WG(2, true, 0); }
			else { if (halton) CRH_LAUNCH_WG(2, false, 1); else CRH_LAUNCH_WG(2, false, 0); }
		} else {
			if (c->hasPrograms) { if (halton) CRH_LAUNCH_WG(1, true, 1); else CRH_LAUNCH_WG(1, true, 0); }
			else { if (halton) CRH_LAUNCH_WG(1, false, 1); else CRH_LAUNCH_WG(1, false, 0); }
		}
	} else
	if (c->sampler == CRH_SAMPLER_HALTON) {          /* interactive mode: the 128-register variants only */
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
}

/* Load the code object of the selected instantiation now (HIP loads kernels lazily, ~40 ms on first launch) with a launch that finds
 * an empty work queue: crh_scene_upload calls it, so a renderer's first frame is not the one that pays for it. */
static int variantKey(const crh_ctx *c) { return (c->hasPrograms ? 1 : 0) | (c->sampler << 1) | (c->counterLevel << 2) | (c->wavesPerSimd << 4) | (c->kernel << 8); }
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
	const size_t slabs = c->kernel == CRH_KERNEL_ROLL ? CRH_ROLL_SLOTS : 1u;        /* one sample slab per open job */
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
	if (e == hipSuccess) e = hipMalloc((void **)&c->dCounters, CRH_NCOUNTERS * sizeof(unsigned long long));
	if (e == hipSuccess) e = hipMemset(c->dCounters, 0, CRH_NCOUNTERS * sizeof(unsigned long long));
	if (e == hipSuccess) e = hipMalloc((void **)&c->dWork, CRH_WORK_SLOTS * sizeof(uint32_t));
	if (e == hipSuccess) e = hipMemset(c->dWork, 0, CRH_WORK_SLOTS * sizeof(uint32_t));      /* a work counter is zero whenever a dispatch takes it: crh_render_tiles resets it BEHIND the kernel */
	if (e == hipSuccess) e = hipMalloc((void **)&c->dErr, sizeof(unsigned int));
	if (e == hipSuccess) e = hipMemset(c->dErr, 0, sizeof(unsigned int));
	if (e != hipSuccess) {
		const std::string msg = std::string("crh_context_create: ") + hipGetErrorString(e);
		crh_context_destroy(c);
		return fail(CRH_ERR_HIP, msg);
	}
	const char *env = getenv("CRH_BLOCKS_PER_CU");
	if (env && atoi(env) > 0) c->blocksPerCU = atoi(env);
	env = getenv("CRH_TAIL_SPLIT");                 /* dev: the default of CRH_OPT_TAIL_SPLIT for this process (A/B runs of unmodified hosts) */
	if (env && atoi(env) >= 0 && atoi(env) <= 64) c->tailSplit = atoi(env);
	*out = c;
	return CRH_OK;
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
	if (c->dErr) (void)hipFree(c->dErr);
	if (c->dSrgb) (void)hipFree(c->dSrgb);
	if (c->dGather) (void)hipFree(c->dGather);
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
			if (value && !c->dWaveStats) { if (hipMalloc((void **)&c->dWaveStats, 2 * 8192 * sizeof(unsigned long long)) != hipSuccess) return fail(CRH_ERR_HIP, "wave stats alloc"); }
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
		case CRH_OPT_SCHED_RUNS: {     /* fillTo | runNum << 12 | triInRun << 16 | ctrlInRun << 24 | shadeMin << 32 (0: keep) */
			Sched k = c->sched;
			k.fillTo = (int)(value & 0xFFF); k.runNum = (int)((value >> 12) & 0xF); k.triInRun = (int)((value >> 16) & 0xFF); k.ctrlInRun = (int)((value >> 24) & 0xFF);
			if ((value >> 32) & 0xFF) k.shadeMin = (int)((value >> 32) & 0xFF);
			if (value < 0 || k.fillTo > 192 || k.runNum < 1 || k.runNum > 8 || k.triInRun < 1 || k.triInRun > 65 || k.ctrlInRun < 1 || k.ctrlInRun > 65 || k.shadeMin < 1 || k.shadeMin > 128) return fail(CRH_ERR_INVALID, "bad scheduler run parameters");
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
			if (value != CRH_KERNEL_WAVE && value != CRH_KERNEL_WG && value != CRH_KERNEL_ROLL) return fail(CRH_ERR_INVALID, "kernel must be CRH_KERNEL_ROLL, CRH_KERNEL_WAVE or CRH_KERNEL_WG");
			c->kernel = (int)value; return CRH_OK;
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

int crh_scene_upload(crh_ctx *c, const crh_scene_desc *scene) {
	if (!c || !scene) return fail(CRH_ERR_INVALID, "crh_scene_upload: NULL argument");
	int rc = setDevice(c);
	if (rc) return rc;
	CompiledScene cs;
	std::string err;
	rc = compile_scene(scene, cs, err);
	if (rc != CRH_OK) return fail(rc, "crh_scene_upload: " + err);
	HIP_TRY(hipStreamSynchronize(c->stream));
	freeScene(c);
	DScene d;
	memset(&d, 0, sizeof(d));
#define UP(field, ptr, count) do { rc = upload(c, ptr, count, &d.field); if (rc) { freeScene(c); return rc; } } while (0)
	UP(nodes, cs.nodes.data(), cs.nodes.size());
	UP(tris, cs.tris.data(), cs.tris.size());
	UP(shade, cs.shade.data(), cs.shade.size());
	UP(prims, scene->prim_indices, (size_t)scene->prim_index_count);
	UP(instances, cs.instances.data(), cs.instances.size());
	UP(materials, cs.materials.data(), cs.materials.size());
	UP(bsdfs, cs.bsdfs.data(), cs.bsdfs.size());
	UP(consts, cs.consts.data(), cs.consts.size());
	UP(images, cs.images.data(), cs.images.size());
	UP(prog, cs.prog.data(), cs.prog.size());
	UP(textures, cs.textures.data(), cs.textures.size());
	UP(texels, cs.texels.data(), cs.texels.size());
#undef UP
	d.tlas_first = cs.tlas_first;
	d.material_count = (uint32_t)cs.materials.size(); d.bsdf_count = (uint32_t)cs.bsdfs.size(); d.const_count = (uint32_t)cs.consts.size();
	d.image_count = (uint32_t)cs.images.size(); d.texture_count = (uint32_t)cs.textures.size();
	d.tlas_root = cs.tlas_root; d.tlas_node_count = cs.tlas_node_count; d.tlas_prim_base = cs.tlas_prim_base; d.shade_classes = cs.shade_classes; d.instance_count = (uint32_t)cs.instances.size();
	d.background = cs.background; d.camera = cs.camera;
	c->d = d;
	c->hasPrograms = cs.prog.size() > 1 || cs.has_volumes || getenv("CRH_FORCE_PROGRAMS") != nullptr;    /* the rare-features kernel variant */
	c->hasVolumes = cs.has_volumes;
	c->haveScene = true;
	/* The scene is resident, and the device has nothing left to do, when this function returns: a blocking device-to-host copy on the NULL stream ends the
	 * set-up. Measured in round 3 (CRH_TRACE_SYNC): without it the first dispatch's kernel starts 7-25 ms after its launch — behind work the runtime
	 * still owes the pageable host-to-device copies above, which neither hipStreamSynchronize on the context's (non-blocking) stream nor
	 * hipDeviceSynchronize waits for; with it, 1-5 us. (Until round 3 the watchdog flag's copy in crh_synchronize was this barrier by accident.) */
	rc = preloadKernel(c, true);         /* ... and an (empty) launch of the kernel on the context's stream is waited for: see below */
	if (rc != CRH_OK) return rc;
	unsigned int flag = 0;
	HIP_TRY(hipMemcpy(&flag, c->dErr, sizeof(flag), hipMemcpyDeviceToHost));
	return CRH_OK;
}

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

int crh_render_tiles(crh_ctx *c, const crh_render_params *P, const crh_tile *tiles, uint32_t tile_count, float *dev_fb) {
	if (!c || !P || !dev_fb || (!tiles && tile_count)) return fail(CRH_ERR_INVALID, "crh_render_tiles: NULL argument");
	if (!c->haveScene) return fail(CRH_ERR_INVALID, "crh_render_tiles: no scene uploaded");
	if (P->image_width <= 0 || P->image_height <= 0 || P->pass_count < 0 || P->first_pass < 0 || P->max_passes < P->first_pass + P->pass_count)
		return fail(CRH_ERR_INVALID, "crh_render_tiles: bad render parameters");
	int rc = setDevice(c);
	if (rc) return rc;
	(void)resolveTimes(c, false);
	const bool wg = c->kernel == CRH_KERNEL_WG;
	const PlanKnobs knobs{c->unitItems, c->unitsPerWave, c->tailPercent, c->tail2Percent, c->passChunk, c->cuCount, c->blocksPerCU, wg,
	                      c->kernel == CRH_KERNEL_ROLL && P->bounces > 0 ? c->tailSplit : 0};
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
	if (c->dWaveStats && grid * (CRH_BLOCK / 64) > 8192) return fail(CRH_ERR_INVALID, "wave stats: grid too large");
	{
		size_t need = (size_t)grid * (wg ? 1 : CRH_BLOCK / 64) * (size_t)(bw * bh) * (size_t)chunk * 3;
		if (c->kernel == CRH_KERNEL_ROLL) need *= CRH_ROLL_SLOTS;      /* one sample slab per open job */
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
std::string rcclError(const char *what, int rc) { return std::string(what) + ": " + (g_rccl.GetErrorString && rc > 0 ? g_rccl.GetErrorString(rc) : "error"); }
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
	HIP_TRY(hipMemset(c->dCounters, 0, CRH_NCOUNTERS * sizeof(unsigned long long)));
	c->lastMs = 0.0f; c->totalMs = 0.0; c->launches = 0;
	return CRH_OK;
}

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

}  // extern "C"

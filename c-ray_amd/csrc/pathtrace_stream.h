/*
 * pathtrace_stream.h — the STREAMING form of the path tracer (round 6; CRH_OPT_KERNEL = CRH_KERNEL_STREAM): the wave machine of pathtrace_roll.h cut into three kernels
 * that take turns on the device, so that the WALK — getClosestIsect, bvh.c:354-441 via pathtrace.c:26-30 — runs in a kernel of its own at the occupancy the megakernel cannot
 * reach (72 VGPRs / 26 KB of LDS: six or seven waves per SIMD instead of four; profiles/r06c_probe_walk.log: 1.2-1.7 x the megakernel's walk on the path tracer's own rays).
 *
 * The paths live in a POOL in global memory: `cohorts` x 1024 slots, one path per slot, structure-of-arrays in four 16-byte planes (origin | depth, direction | item,
 * weight | sampler lo, radiance | sampler hi — the four quarters of the megakernel's path record) so that the 64 lanes of a wave read and write consecutive 16-byte words.
 * There are two pools; an ITERATION reads one and writes the other:
 *   k_stream_walk   persistent waves (the walk-only machine of walk_probe.h, one site per step kind): a unit = 128 consecutive slots of a cohort; lanes take rays from the
 *                   unit in hand as they fall idle, walk them with the megakernel's lane code (walkBegin / stepNode / stepTri / stepCtrl), and leave the closest hit
 *                   (t, u, v, prim slot | instance) in the slot's hit record.
 *   k_stream_shade  one workgroup per cohort (persistent, cohorts from a counter): sorts the cohort's slots into surface hits and misses (an ordered list in LDS), runs
 *                   shadeCore (pathtrace.c:39-57) on full waves of hits, then of misses; a path that continues goes — with its next ray — to the next free slot at the
 *                   front of the SAME cohort of the other pool, a path that ends stages its sample. The slots that stay free are REFILLED in place with new camera rays
 *                   (renderer.c:280-284: the next items of the dispatch, handed out by one fetch-and-add per cohort), so every cohort stays full until the dispatch
 *                   runs out of items: no global compaction, no global queue.
 *   k_stream_fold   the running mean (renderer.c:288-291) in pass order: the dispatch's passes are cut into CHUNKS (all pixels x a few passes); a chunk's samples wait in one
 *                   slot of a ring of slabs, a counter per slot says how many are still missing, and the chunk that is complete — and next in order — is folded
 *                   into the frame, which frees its slot for the chunk sixteen further on. Generation never enters a chunk whose slot is not free.
 * The host (cray_hip.hip: renderStream) enqueues iterations in groups and watches a word in host-visible memory that the fold kernel sets when the last chunk is folded.
 *
 * Every step of a path is the one the other kernel forms run (same lane code, same arithmetic): the frame is the same bit for bit (tests/test_gpu_parity.py,
 * tests/test_kernel_emu.py), whatever the pool size, the chunk size or the order in which cohorts are served.
 *
 * MEASURED (MI355X, profiles/r06f_ab_stream_long.log, r06e_stream_trace.txt, r06j_pmc_stream_statues.txt): the walk kernel runs at the probe's rate — 16.8 M rays
 * in 5.5 ms on statues.json (3.0 Gray/s; the megakernel's walk: 2.6), 1.25 ms on hdr.json — and the FRAME is slower than the megakernel's: 0.82 x statues, 0.72 x / 0.77 x
 * the soups, 0.51 x hdr.json, because the shade pass (2.3 ms per 16.8 M paths, parked on memory 86 % of its cycles behind HBM-latency chains) is serial with the walk and the
 * occupancy it buys the walk is 0-24 %. This form is an OPTION (CRH_OPT_KERNEL = CRH_KERNEL_STREAM, env CRH_KERNEL=stream), not the default; it is kept because it is the
 * measured answer to "what does the walk do at more than four waves per SIMD", bit-identical and tested.
 */
#pragma once

#ifndef CRH_STREAM_SHADE_WPS
#define CRH_STREAM_SHADE_WPS 4         /* workgroups of k_stream_shade per CU the register allocator leaves room for (97 VGPRs as it is; 5 was measured: see DESIGN.md section 3) */
#endif
#define CRH_SF_COHORT 1024u            /* slots per cohort (= what one workgroup of k_stream_shade sorts in LDS: 16-bit indices) */
#define CRH_SF_UNIT 128u               /* slots per unit of the walk kernel's work counter */
#define CRH_SF_COUNTERS 16u            /* unit counters of the walk kernel (a power of two) ... */
#define CRH_SF_CTR_STRIDE 32u          /* ... this many words apart */
#define CRH_SF_RING 16u                /* slabs in the sample ring (item word: ring slot << 28 | index inside the chunk) */
#define CRH_SF_SLOT_SHIFT 28u
#define CRH_SF_IDX_MASK 0x0FFFFFFFu
#define CRH_SF_CHUNK_ITEMS_MAX (1u << CRH_SF_SLOT_SHIFT)

/* one of the two path pools */
struct StreamPool {
	f4 *p0, *p1, *p2, *p3;             /* {o, depth} {d, item} {weight, rng.lo} {radiance, rng.hi}: cohorts * 1024 entries each */
	uint32_t *count;                   /* per cohort: its live paths are its first count[] slots */
};
/* what a dispatch's kernels share and nobody changes (by value) */
struct StreamPlan {
	const crh_tile *tiles;             /* the dispatch's rectangles, ... */
	const uint32_t *start;             /* ... start[t] = first pixel (in list order) of tile t, start[ntiles] = npix */
	uint32_t ntiles, npix;
	uint32_t cohorts;
	uint32_t passesPerChunk, lastPasses, chunkCount;       /* chunk c = passes [c * passesPerChunk, ...) of every pixel; the last chunk holds lastPasses */
	uint32_t chunkItems;               /* npix * passesPerChunk: items of a full chunk; a slab holds this many samples */
	unsigned long long genTotal;       /* items of the dispatch: npix * pass_count */
	float *slab;                       /* the ring: slot s at slab + s * chunkItems * 3 */
	f4 *hit;                           /* per pool slot: the closest hit of the walk (t, u, v, prim slot) ... */
	int32_t *hitInst;                  /* ... and its instance (< 0: miss) */
	unsigned int *done;                /* host-visible: set to `seq` when the last chunk is folded */
	unsigned int seq;
};
/* the dispatch's state on the device */
struct StreamCtl {
	unsigned long long genNext;        /* items handed out so far (chunk-major numbering) */
	uint32_t foldNext;                 /* the chunk that folds next */
	uint32_t foldDone;                 /* workgroups of the running k_stream_fold that are through */
	uint32_t shadeCtr;                 /* work counter of k_stream_shade (k_stream_fold zeroes it) */
	uint32_t liveIn, liveOut;          /* paths in the pool the iteration reads / paths the shade kernel has put into the pool it writes */
	uint32_t left[CRH_SF_RING];        /* samples still missing in the chunk that occupies the ring slot */
	uint32_t walkCtr[CRH_SF_COUNTERS * CRH_SF_CTR_STRIDE];          /* work counters of k_stream_walk, 128 bytes apart (k_stream_fold zeroes them) */
};

/* pixel `q` of a w x h rectangle in 8 x 8 blocks (bands of eight rows, blocks left to right, rows inside a block): neighbours in the item order are neighbours in the frame */
CRH_DEV void streamPixelOf(const crh_tile t, uint32_t q, int &x, int &y) {
	const uint32_t w = (uint32_t)(t.x1 - t.x0), h = (uint32_t)(t.y1 - t.y0);
	const uint32_t band = q / (w * 8u);
	const uint32_t hb = h - band * 8u < 8u ? h - band * 8u : 8u;
	const uint32_t r = q - band * w * 8u;
	const uint32_t k = r / (8u * hb);                  /* (only a band's last block can be narrower, and it comes last) */
	const uint32_t wb = w - k * 8u < 8u ? w - k * 8u : 8u;
	const uint32_t s = r - k * 8u * hb;
	const uint32_t ly = s / wb, lx = s - ly * wb;
	x = t.x0 + (int)(k * 8u + lx);
	y = t.y0 + (int)(band * 8u + ly);
}
/* pixel p of the dispatch (list order) -> frame coordinates */
CRH_DEV void streamPixel(const StreamPlan &Pl, uint32_t p, int &x, int &y) {
	uint32_t lo = 0, hi = Pl.ntiles;
	while (hi - lo > 1u) { const uint32_t mid = (lo + hi) >> 1; if (Pl.start[mid] <= p) lo = mid; else hi = mid; }
	streamPixelOf(Pl.tiles[lo], p - Pl.start[lo], x, y);
}

/* the walk kernel's traversal stack: NLDS entries in LDS (entry-major), deeper ones in the wave's overflow columns; the 15 park slots; line 0 of the instance records in LDS
 * when the scene has at most CRH_INST_LDS0_MAX instances (INST) */
template <int NLDS, bool INST>
struct WalkStack {
	lds_u32 *lds;
	lds_u32 *parkp;
	glb_u32 *ovf;
	const lds_u32 *inst0;
	__device__ __forceinline__ InstLine instLine(const DScene &S, int32_t idx, int line) const {
		if (INST && line == 0 && inst0) {
			const lds_u32 *p = inst0 + (uint32_t)idx * 16u;
			return InstLine{ldsLoadF4(p), ldsLoadF4(p + 4), ldsLoadF4(p + 8), ldsLoadF4(p + 12)};
		}
		const f4 *g = (const f4 *)(S.instances + idx) + 4 * line;
		return InstLine{g[0], g[1], g[2], g[3]};
	}
	__device__ __forceinline__ void park(int i, uint32_t v) { parkp[i * CRH_BLOCK] = v; }
	__device__ __forceinline__ uint32_t unpark(int i) { return parkp[i * CRH_BLOCK]; }
	__device__ __forceinline__ void push(uint32_t i, uint32_t v) {
		if (__builtin_expect(i < (uint32_t)NLDS, 1)) lds[i * CRH_BLOCK] = v;
		else ovf[(i - (uint32_t)NLDS) * 64u + (threadIdx.x & 63u)] = v;
	}
	__device__ __forceinline__ uint32_t pop(uint32_t i) {
		uint32_t v;
		if (__builtin_expect(i < (uint32_t)NLDS, 1)) v = lds[i * CRH_BLOCK];
		else v = ovf[(i - (uint32_t)NLDS) * 64u + (threadIdx.x & 63u)];
		return v;
	}
};

/* start of a dispatch: the pools empty, the ring's first slabs waiting for the first chunks, the tile list (tiles, then their first pixels) fetched from the host's pinned copy */
__global__ void k_stream_init(const StreamPlan Pl, uint32_t *countA, uint32_t *countB, StreamCtl *ctl, const uint32_t *listHost, uint32_t *listDev, uint32_t listWords) {
	for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < Pl.cohorts; i += gridDim.x * blockDim.x) { countA[i] = 0u; countB[i] = 0u; }
	for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < listWords; i += gridDim.x * blockDim.x) listDev[i] = listHost[i];
	if (blockIdx.x == 0 && threadIdx.x == 0) {
		ctl->genNext = 0ull; ctl->foldNext = 0u; ctl->foldDone = 0u; ctl->shadeCtr = 0u;
		for (uint32_t i = 0; i < CRH_SF_COUNTERS; ++i) ctl->walkCtr[i * CRH_SF_CTR_STRIDE] = 0u; ctl->liveIn = 0u; ctl->liveOut = 0u;
		for (uint32_t s = 0; s < CRH_SF_RING; ++s)
			ctl->left[s] = s < Pl.chunkCount ? Pl.npix * (s + 1u == Pl.chunkCount ? Pl.lastPasses : Pl.passesPerChunk) : 0u;
	}
}

/* WALK. WPS / NLDS / INST as in walk_probe.h (the kernel is its form 3: a lean run with one site per step kind); LEVEL: counter level (1: rays; 2: every walk counter) */
template <int WPS, int NLDS, bool INST, int LEVEL>
__global__ __launch_bounds__(CRH_BLOCK, WPS) void k_stream_walk(const DScene Sarg, const StreamPool in, f4 *hitsArg, int32_t *hitInstArg, uint32_t cohorts, StreamCtl *ctlArg,
                                                                const Sched K, uint32_t *ovfAll, unsigned long long *counters) {
	__shared__ uint32_t s_stack[(NLDS > 0 ? NLDS : 1) * CRH_BLOCK];
	__shared__ uint32_t s_park[CRH_PARK_SLOTS * CRH_BLOCK];
	__shared__ __attribute__((aligned(16))) uint32_t s_inst0[INST ? CRH_INST_LDS0_MAX * 16u : 4u];
	StreamCtl *const ctl = (StreamCtl *)(__attribute__((address_space(1))) StreamCtl *)ctlArg;
	if (ctl->liveIn == 0u) return;               /* nothing to walk (the first iteration of a dispatch, and the iterations a group holds beyond the dispatch's end) */
	const DScene S = globalize(Sarg);
	const f4 *const rayO = asGlobal(in.p0), *const rayD = asGlobal(in.p1);
	const uint32_t *const count = asGlobal(in.count);
	f4 *const hits = (f4 *)(__attribute__((address_space(1))) f4 *)hitsArg;
	int32_t *const hitInst = (int32_t *)(__attribute__((address_space(1))) int32_t *)hitInstArg;
	WalkStack<NLDS, INST> stk;
	stk.lds = (lds_u32 *)&s_stack[threadIdx.x];
	stk.parkp = (lds_u32 *)&s_park[threadIdx.x];
	stk.inst0 = nullptr;
	if (INST && S.instance_count <= CRH_INST_LDS0_MAX) {
		for (uint32_t i = threadIdx.x; i < S.instance_count * 16u; i += CRH_BLOCK) s_inst0[i] = ((const uint32_t *)(S.instances + (i >> 4)))[i & 15u];
		__syncthreads();
		stk.inst0 = (const lds_u32 *)s_inst0;
	}
	const uint32_t lane = threadIdx.x & 63u;
	const uint32_t wave = (blockIdx.x * CRH_BLOCK + threadIdx.x) >> 6;
	stk.ovf = (glb_u32 *)ovfAll + (size_t)__builtin_amdgcn_readfirstlane(wave) * CRH_OVF_WORDS_PER_WAVE;
	CountersT<LEVEL, false, false> cnt;
	memset(&cnt, 0, sizeof(cnt));
	NullPort port;                               /* (scenes with volumes — a sampler draw inside the walk — are rendered by the other kernel form) */
	Walk w;
	memset(&w, 0, sizeof(w));
	w.phase = PH_IDLE;
	uint32_t mySlot = 0;
	uint32_t cur = 0, end = 0;                   /* wave-uniform: the slots of the unit in hand whose rays have not started */
	bool dry = false;                            /* wave-uniform: every counter has run past its last unit */
	uint32_t part = __builtin_amdgcn_readfirstlane(wave) & (CRH_SF_COUNTERS - 1u), partsDry = 0;          /* wave-uniform: the counter in use, and how many have been seen dry */
	const uint32_t unitsPerCohort = CRH_SF_COHORT / CRH_SF_UNIT, nUnits = cohorts * unitsPerCohort;
	/* retire + refill (pathtrace_roll.h: retireRefill): lanes whose walk ended leave the hit in their slot's record; they and the idle lanes take the next rays of the unit in hand */
	auto retireRefill = [&]() __attribute__((always_inline)) {
		if (w.phase == PH_SHADE) {
			hits[mySlot] = f4{w.hit.t, w.hit.u, w.hit.v, asF32((uint32_t)w.hit.slot)};
			hitInst[mySlot] = w.hit.inst;
			w.phase = PH_IDLE;
		}
		const bool idle = (w.phase == PH_IDLE);
		const unsigned long long em = __ballot(idle);
		const uint32_t er = laneRank(em);
		for (int tries = 0; tries < 8 && cur == end && !dry; ++tries) {          /* (a unit beyond its cohort's live slots is empty: try the next one) */
			/* the units are dealt over CRH_SF_COUNTERS counters (unit = k * CRH_SF_COUNTERS + counter: sixteen addresses, sixteen L2 channels — one counter served an
			 * atomic every 15 ns, a floor of 2 ms per iteration whatever the pool held); a wave starts at its own counter and moves on when one runs out */
			uint32_t k = 0;
			if (lane == 0) k = atomicAdd((uint32_t *)&ctl->walkCtr[part * CRH_SF_CTR_STRIDE], 1u);
			k = __builtin_amdgcn_readfirstlane(k);
			const uint32_t u = k * CRH_SF_COUNTERS + part;
			if (u < nUnits) {
				const uint32_t t = u / unitsPerCohort, off = (u % unitsPerCohort) * CRH_SF_UNIT;
				const uint32_t n = __builtin_amdgcn_readfirstlane(count[t]);
				if (off < n) { cur = t * CRH_SF_COHORT + off; end = t * CRH_SF_COHORT + (n < off + CRH_SF_UNIT ? n : off + CRH_SF_UNIT); }
			} else {
				part = (part + 1u) & (CRH_SF_COUNTERS - 1u);
				if (++partsDry == CRH_SF_COUNTERS) dry = true;
			}
		}
		const uint32_t take = min(end - cur, (uint32_t)__popcll(em));
		if (idle && er < take) {
			mySlot = cur + er;
			const f4 q0 = rayO[mySlot], q1 = rayD[mySlot];
			walkBegin(S, w, stk, v3{q0.x, q0.y, q0.z}, v3{q1.x, q1.y, q1.z}, cnt, port, (uint32_t)K.rayFlags);
		}
		cur += take;
	};
	for (;;) {
		const uint32_t ph = w.phase;
		const int nN = __popcll(__ballot(ph == PH_NODE)), nT = __popcll(__ballot(ph == PH_TRI)), nC = __popcll(__ballot(ph == PH_CTRL || ph == PH_NODE_SLOW));
		const int nF = __popcll(__ballot(ph == PH_SHADE));
		const int nE = 64 - nN - nT - nC - nF;
		const int walkers = nN + nT + nC;
		const bool more = !dry || cur != end;
		if (walkers == 0 && nF == 0 && !more) break;
		/* the round picks a mode by the megakernel's rules — 0 a node run (which serves triangle, instance-entry and retire / refill steps in place once enough lanes wait
		 * for them), 1 a triangle run, 2 one control step, 3 retire + refill — and ONE loop body serves all four, so that the register allocator sees the largest step and not
		 * the sum of the copies (72 VGPRs; the megakernel's fused node run needs 126) */
		int mode = 0;
		if (walkers == 0 || (nF + nE >= K.swapMin && (nF > 0 || more))) mode = 3;
		else { int best = nN * K.wNode; if (nT * K.wTri > best) { best = nT * K.wTri; mode = 1; } if (nC * K.wCtrl > best) mode = 2; }
		const int n0 = mode == 1 ? nT : nN;
		bool again;
		do {
			if (mode == 0 && w.phase == PH_NODE) stepNode<true>(S, w, stk, cnt, port);
			const int nTw = (int)__popcll(__ballot(w.phase == PH_TRI));
			if (mode == 1 || (mode == 0 && nTw >= K.triInRun)) { if (w.phase == PH_TRI) stepTri(S, w, stk, cnt, port); }
			const int nCw = (int)__popcll(__ballot(w.phase == PH_CTRL));
			if (mode == 2 || (mode == 0 && nCw >= K.ctrlInRun)) {
				if (w.phase == PH_CTRL) stepCtrl(S, w, stk, cnt, port);
				if (mode == 2 && __ballot(w.phase == PH_NODE_SLOW)) { if (w.phase == PH_NODE_SLOW) stepNodeAny<false>(S, w, stk, cnt, port); }
			}
			const int nFi = (int)__popcll(__ballot(w.phase == PH_SHADE)), nEi = (int)__popcll(__ballot(w.phase == PH_IDLE));
			if (mode == 3 || (mode == 0 && nFi + nEi >= K.swapInRun && (nFi > 0 || !dry || cur != end))) retireRefill();
			again = mode == 0 ? (int)__popcll(__ballot(w.phase == PH_NODE)) * 8 >= n0 * K.runNum : mode == 1 ? (int)__popcll(__ballot(w.phase == PH_TRI)) * 8 >= n0 * K.runNum : false;
		} while (again);
	}
	const bool lead = (lane == 0);
	uint32_t v;
	v = waveSum(cnt.rays); if (lead && v) atomicAdd(&counters[1], (unsigned long long)v);
	if constexpr (LEVEL >= 2) {
		v = waveSum(cnt.node_tests); if (lead && v) atomicAdd(&counters[2], (unsigned long long)v);
		v = waveSum(cnt.tri_tests); if (lead && v) atomicAdd(&counters[3], (unsigned long long)v);
		v = waveSum(cnt.inst_visits); if (lead && v) atomicAdd(&counters[4], (unsigned long long)v);
		v = waveSum(cnt.inst_hits); if (lead && v) atomicAdd(&counters[5], (unsigned long long)v);
		v = waveSum(cnt.sphere_tests); if (lead && v) atomicAdd(&counters[6], (unsigned long long)v);
	}
}

/* SHADE + REFILL: pathtrace.c:39-57 on the walked paths of one pool, their continuations and the camera rays that take the freed slots into the other pool */
template <int LEVEL, bool PROG, int SAMP>
__global__ __launch_bounds__(CRH_BLOCK, CRH_STREAM_SHADE_WPS) void k_stream_shade(const DScene Sarg, const crh_render_params P, const StreamPlan Pl, const StreamPool in, const StreamPool out, StreamCtl *ctlArg,
                                                               unsigned long long *counters) {
	__shared__ uint16_t s_list[CRH_SF_COHORT];            /* slot indices inside the cohort: surface hits from the front, misses from the back, both in slot order */
	__shared__ uint32_t s_seg[2][16];                     /* hits / misses per (pass, wave) segment of the classification */
	__shared__ uint32_t s_fin[CRH_SF_RING];               /* samples this cohort has staged, per ring slot */
	__shared__ uint32_t s_word[8];
	enum { SW_COHORT, SW_OUT, SW_NEW, SW_G0_CHUNK, SW_G0_IDX };
	StreamCtl *const ctl = (StreamCtl *)(__attribute__((address_space(1))) StreamCtl *)ctlArg;
	const unsigned long long genLimit0 = (unsigned long long)(ctl->foldNext + CRH_SF_RING) * (unsigned long long)Pl.chunkItems;        /* (k_stream_fold moves foldNext: not while this kernel runs) */
	const unsigned long long genLimit = genLimit0 < Pl.genTotal ? genLimit0 : Pl.genTotal;
	const uint32_t liveIn = ctl->liveIn;
	if (liveIn == 0u && ctl->genNext >= genLimit) return;          /* nothing to shade, nothing to generate */
	const DScene S = globalize(Sarg);
	CRH_EM_POW_TABLES_INIT();
	LdsStack stk;
	stk.lds = nullptr; stk.parkp = nullptr; stk.ovf = nullptr;
	CRH_STAGE_SHADE_TABLES();
	CRH_STAGE_INSTANCE_TABLES();
	CountersT<LEVEL, PROG, false> cnt;
	memset(&cnt, 0, sizeof(cnt));
	const uint32_t lane = threadIdx.x & 63u, wv = threadIdx.x >> 6;
	const f4 *const i0 = asGlobal(in.p0), *const i1 = asGlobal(in.p1), *const i2 = asGlobal(in.p2), *const i3 = asGlobal(in.p3);
	f4 *const o0 = (f4 *)(__attribute__((address_space(1))) f4 *)out.p0, *const o1 = (f4 *)(__attribute__((address_space(1))) f4 *)out.p1;
	f4 *const o2 = (f4 *)(__attribute__((address_space(1))) f4 *)out.p2, *const o3 = (f4 *)(__attribute__((address_space(1))) f4 *)out.p3;
	const f4 *const hit = asGlobal(Pl.hit);
	const int32_t *const hitInst = asGlobal(Pl.hitInst);
	float *const slab = (float *)(__attribute__((address_space(1))) float *)Pl.slab;
	typedef volatile __attribute__((address_space(3))) uint32_t lds_word;
	lds_word *const sw = (lds_word *)s_word;
	lds_word *const fin = (lds_word *)s_fin;
#ifdef CRH_STREAM_CLOCKS          /* dev (tools/probe_stream_clocks.py): where a workgroup's time goes, in 100 MHz ticks summed over workgroups: counters[8 + phase] */
	unsigned long long tPh = wall_clock64();
#define CRH_SCLK(k) do { if (threadIdx.x == 0) { const unsigned long long t_ = wall_clock64(); atomicAdd(&counters[8 + (k)], t_ - tPh); tPh = t_; } } while (0)
#else
#define CRH_SCLK(k) do { } while (0)
#endif
	for (;;) {
		__syncthreads();                      /* every thread is through with the last cohort's words */
		CRH_SCLK(5);
		if (threadIdx.x == 0) { s_word[SW_COHORT] = atomicAdd((uint32_t *)&ctl->shadeCtr, 1u); s_word[SW_OUT] = 0u; s_word[SW_NEW] = 0u; }
		if (threadIdx.x < CRH_SF_RING) s_fin[threadIdx.x] = 0u;
		__syncthreads();
		const uint32_t t = sw[SW_COHORT];
		CRH_SCLK(0);
		if (t >= Pl.cohorts) break;
		const uint32_t base = t * CRH_SF_COHORT;
		const uint32_t nIn = liveIn ? asGlobal(in.count)[t] : 0u;          /* (a pool nobody wrote last iteration holds nothing) */
		/* sort the cohort's slots into hits and misses, keeping slot order (consecutive lanes then read consecutive records) */
		uint32_t nHit = 0, nMiss = 0;
		if (nIn) {
			bool isHit[4], isMiss[4];
			uint32_t rkH[4], rkM[4];
#pragma unroll
			for (int k = 0; k < 4; ++k) {
				const uint32_t e = (uint32_t)k * CRH_BLOCK + threadIdx.x;
				const bool valid = e < nIn;
				const int32_t inst = valid ? hitInst[base + e] : 0;
				isHit[k] = valid && inst >= 0; isMiss[k] = valid && inst < 0;
				const unsigned long long hm = __ballot(isHit[k]), mm = __ballot(isMiss[k]);
				rkH[k] = laneRank(hm); rkM[k] = laneRank(mm);
				if (lane == 0) { s_seg[0][k * 4 + (int)wv] = (uint32_t)__popcll(hm); s_seg[1][k * 4 + (int)wv] = (uint32_t)__popcll(mm); }
			}
			__syncthreads();
			uint32_t offH[4] = {0, 0, 0, 0}, offM[4] = {0, 0, 0, 0};
#pragma unroll
			for (int k = 0; k < 4; ++k) {
#pragma unroll
				for (int w2 = 0; w2 < 4; ++w2) {
					if ((uint32_t)w2 == wv) { offH[k] = nHit; offM[k] = nMiss; }
					nHit += ((volatile __attribute__((address_space(3))) uint32_t *)s_seg[0])[k * 4 + w2];
					nMiss += ((volatile __attribute__((address_space(3))) uint32_t *)s_seg[1])[k * 4 + w2];
				}
			}
#pragma unroll
			for (int k = 0; k < 4; ++k) {
				const uint32_t e = (uint32_t)k * CRH_BLOCK + threadIdx.x;
				if (isHit[k]) s_list[offH[k] + rkH[k]] = (uint16_t)e;
				if (isMiss[k]) s_list[CRH_SF_COHORT - 1u - (offM[k] + rkM[k])] = (uint16_t)e;
			}
			__syncthreads();
		}
		CRH_SCLK(1);
		/* surface hits, a wave's 64 at a time: pathtrace.c:43-57 */
		for (uint32_t jb = wv * 64u; jb < nHit; jb += CRH_BLOCK) {
			const uint32_t j = jb + lane;
			const bool act = j < nHit;
			bool cont = false;
			uint32_t slotDone = CRH_NONE;
			v3 ro{0.0f, 0.0f, 0.0f}, rd{0.0f, 0.0f, 0.0f};
			PathRecT<RngT<SAMP>> r;
			memset(&r, 0, sizeof(r));
			uint32_t item = 0;
			if (act) {
				const uint32_t i = base + (uint32_t)((volatile __attribute__((address_space(3))) uint16_t *)s_list)[j];
				const f4 q0 = i0[i], q1 = i1[i], q2 = i2[i], q3 = i3[i], q4 = hit[i];
				ro = v3{q0.x, q0.y, q0.z}; rd = v3{q1.x, q1.y, q1.z};
				r.wr = q2.x; r.wg = q2.y; r.wb = q2.z;
				r.fr = q3.x; r.fg = q3.y; r.fb = q3.z;
				r.rng.state = (uint64_t)asU32(q2.w) | ((uint64_t)asU32(q3.w) << 32);
				r.depth = (int)asU32(q0.w);
				item = asU32(q1.w);
				TravHit h;
				h.t = q4.x; h.u = q4.y; h.v = q4.z; h.slot = (int32_t)asU32(q4.w); h.inst = hitInst[i];
				__builtin_assume(h.inst >= 0);
				cont = shadeCore(S, P, ro, rd, h, r, cnt, stk);
				if (!cont) {
					slotDone = item >> CRH_SF_SLOT_SHIFT;
					float *so = slab + ((size_t)slotDone * Pl.chunkItems + (item & CRH_SF_IDX_MASK)) * 3u;
					so[0] = r.fr; so[1] = r.fg; so[2] = r.fb;
				}
			}
			const unsigned long long cm = __ballot(cont);
			if (cm) {
				uint32_t ob = 0;
				if (lane == 0) ob = atomicAdd((uint32_t *)&s_word[SW_OUT], (uint32_t)__popcll(cm));
				ob = __builtin_amdgcn_readfirstlane(ob);
				if (cont) {
					const uint32_t o = base + ob + laneRank(cm);
					o0[o] = f4{ro.x, ro.y, ro.z, asF32((uint32_t)r.depth)};
					o1[o] = f4{rd.x, rd.y, rd.z, asF32(item)};
					o2[o] = f4{r.wr, r.wg, r.wb, asF32((uint32_t)r.rng.state)};
					o3[o] = f4{r.fr, r.fg, r.fb, asF32((uint32_t)(r.rng.state >> 32))};
				}
			}
			/* the staged samples, per ring slot (a wave's paths are nearly always of one chunk) */
			unsigned long long dm = __ballot(slotDone != CRH_NONE);
			while (dm) {
				const uint32_t s0 = __builtin_amdgcn_readfirstlane(__shfl(slotDone, (int)__builtin_ctzll(dm)));
				const unsigned long long sm = __ballot(slotDone == s0);
				if (lane == 0) atomicAdd((uint32_t *)&s_fin[s0], (uint32_t)__popcll(sm));
				dm &= ~sm;
			}
		}
#ifdef CRH_STREAM_CLOCKS
		__syncthreads();
#endif
		CRH_SCLK(2);
		/* misses: the background (pathtrace.c:39-42); every one of these paths ends */
		for (uint32_t jb = wv * 64u; jb < nMiss; jb += CRH_BLOCK) {
			const uint32_t j = jb + lane;
			uint32_t slotDone = CRH_NONE;
			if (j < nMiss) {
				const uint32_t i = base + (uint32_t)((volatile __attribute__((address_space(3))) uint16_t *)s_list)[CRH_SF_COHORT - 1u - j];
				const f4 q1 = i1[i], q2 = i2[i], q3 = i3[i];
				v3 ro{0.0f, 0.0f, 0.0f}, rd{q1.x, q1.y, q1.z};
				PathRecT<RngT<SAMP>> r;
				r.wr = q2.x; r.wg = q2.y; r.wb = q2.z;
				r.fr = q3.x; r.fg = q3.y; r.fb = q3.z;
				r.rng.state = 0; r.depth = 0;
				const uint32_t item = asU32(q1.w);
				TravHit h;
				h.t = hit[i].x; h.u = h.v = 0.0f; h.slot = -1; h.inst = -1;
				(void)shadeCore(S, P, ro, rd, h, r, cnt, stk);
				slotDone = item >> CRH_SF_SLOT_SHIFT;
				float *so = slab + ((size_t)slotDone * Pl.chunkItems + (item & CRH_SF_IDX_MASK)) * 3u;
				so[0] = r.fr; so[1] = r.fg; so[2] = r.fb;
			}
			unsigned long long dm = __ballot(slotDone != CRH_NONE);
			while (dm) {
				const uint32_t s0 = __builtin_amdgcn_readfirstlane(__shfl(slotDone, (int)__builtin_ctzll(dm)));
				const unsigned long long sm = __ballot(slotDone == s0);
				if (lane == 0) atomicAdd((uint32_t *)&s_fin[s0], (uint32_t)__popcll(sm));
				dm &= ~sm;
			}
		}
		__syncthreads();
		CRH_SCLK(3);
		/* the slots that stay free take the dispatch's next items (renderer.c:280-284): one fetch-and-add per cohort hands them out */
		const uint32_t nOut = sw[SW_OUT];
		if (threadIdx.x == 0) {
			/* (a fetch-and-add, not a compare-and-swap: a thousand workgroups retrying against each other took 60 ms per iteration. The counter may run past the limit — the
			 * items beyond it are nobody's: k_stream_fold, which moves the limit, takes the counter back to it first) */
			const uint32_t want = CRH_SF_COHORT - nOut;
			uint32_t n = 0;
			unsigned long long g = __hip_atomic_load(&ctl->genNext, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
			if (want && g < genLimit) {
				g = atomicAdd((unsigned long long *)&ctl->genNext, (unsigned long long)want);
				n = g < genLimit ? (uint32_t)(genLimit - g < (unsigned long long)want ? genLimit - g : (unsigned long long)want) : 0u;
			}
			s_word[SW_NEW] = n;
			if (n) { s_word[SW_G0_CHUNK] = (uint32_t)(g / Pl.chunkItems); s_word[SW_G0_IDX] = (uint32_t)(g % Pl.chunkItems); }
			((uint32_t *)(__attribute__((address_space(1))) uint32_t *)out.count)[t] = nOut + n;
			if (nOut + n) atomicAdd((uint32_t *)&ctl->liveOut, nOut + n);
		}
		if (threadIdx.x < CRH_SF_RING) { const uint32_t f = fin[threadIdx.x]; if (f) atomicAdd((uint32_t *)&ctl->left[threadIdx.x], 0u - f); }
		__syncthreads();
		CRH_SCLK(4);
		const uint32_t nNew = sw[SW_NEW];
		if (nNew) {
			const uint32_t c0 = sw[SW_G0_CHUNK], idx0 = sw[SW_G0_IDX];
			for (uint32_t j = threadIdx.x; j < nNew; j += CRH_BLOCK) {
				/* (a cohort's new items span at most two chunks: a dispatch of several chunks has chunks of at least 1024 items) */
				uint32_t c = c0, idx = idx0 + j;
				if (idx >= Pl.chunkItems) { idx -= Pl.chunkItems; ++c; }
				const uint32_t pc = c + 1u == Pl.chunkCount ? Pl.lastPasses : Pl.passesPerChunk;
				const uint32_t pix = (pc & (pc - 1u)) == 0u ? idx >> (31u - (uint32_t)__builtin_clz(pc)) : idx / pc;
				const int pass = P.first_pass + (int)(c * Pl.passesPerChunk + (idx - pix * pc));
				int x = 0, y = 0;
				streamPixel(Pl, pix, x, y);
				v3 ro, rd;
				PathRecT<RngT<SAMP>> r;
				beginPath(S, P, x, y, pass, ro, rd, r, cnt);
				const uint32_t o = base + nOut + j;
				const uint32_t item = ((c & (CRH_SF_RING - 1u)) << CRH_SF_SLOT_SHIFT) | idx;
				o0[o] = f4{ro.x, ro.y, ro.z, asF32((uint32_t)r.depth)};
				o1[o] = f4{rd.x, rd.y, rd.z, asF32(item)};
				o2[o] = f4{r.wr, r.wg, r.wb, asF32((uint32_t)r.rng.state)};
				o3[o] = f4{r.fr, r.fg, r.fb, asF32((uint32_t)(r.rng.state >> 32))};
			}
		}
	}
	const bool lead = (lane == 0);
	uint32_t v;
	v = waveSum(cnt.paths); if (lead && v) atomicAdd(&counters[0], (unsigned long long)v);
	if constexpr (LEVEL >= 2) { v = waveSum(cnt.tex_fetches); if (lead && v) atomicAdd(&counters[7], (unsigned long long)v); }
}

/* FOLD: the chunks that are complete and next in order go into the frame, pass by pass (renderer.c:288-291); the last workgroup through frees their ring slots, resets the
 * iteration's work counters, and says so when the dispatch is over. One lane per pixel. */
__global__ __launch_bounds__(CRH_BLOCK) void k_stream_fold(const crh_render_params P, const StreamPlan Pl, StreamCtl *ctlArg, float *fb) {
	StreamCtl *const ctl = (StreamCtl *)(__attribute__((address_space(1))) StreamCtl *)ctlArg;
	const uint32_t first = ctl->foldNext;
	uint32_t n = 0;                    /* (every workgroup sees the same words: nothing else runs, and only the last workgroup through writes them) */
	while (n < CRH_SF_RING && first + n < Pl.chunkCount && ctl->left[(first + n) & (CRH_SF_RING - 1u)] == 0u) ++n;
	const float *const slab = asGlobal(Pl.slab);
	for (uint32_t k = 0; k < n; ++k) {
		const uint32_t c = first + k;
		const uint32_t pc = c + 1u == Pl.chunkCount ? Pl.lastPasses : Pl.passesPerChunk;
		const float *const chunk = slab + (size_t)(c & (CRH_SF_RING - 1u)) * Pl.chunkItems * 3u;
		const int pass0 = P.first_pass + (int)(c * Pl.passesPerChunk);
		for (uint32_t p = blockIdx.x * CRH_BLOCK + threadIdx.x; p < Pl.npix; p += gridDim.x * CRH_BLOCK) {
			int x = 0, y = 0;
			streamPixel(Pl, p, x, y);
			float *out = fb + ((size_t)x + (size_t)(P.image_height - (y + 1)) * (size_t)P.image_width) * 3;
			float r = out[0], g = out[1], b = out[2];
			const float *sp = chunk + (size_t)p * pc * 3u;
			uint32_t i = 0;
			for (; i + 8u <= pc; i += 8u) {          /* eight samples' loads in flight together, the mean a serial chain (foldBlockPixel) */
				float s[24];
				for (int q = 0; q < 24; ++q) s[q] = sp[3u * i + (uint32_t)q];
				for (int q = 0; q < 8; ++q) foldSample(r, g, b, s[3 * q], s[3 * q + 1], s[3 * q + 2], pass0 + (int)i + q + 1);
			}
			for (; i < pc; ++i) foldSample(r, g, b, sp[3u * i], sp[3u * i + 1u], sp[3u * i + 2u], pass0 + (int)i + 1);
			out[0] = r; out[1] = g; out[2] = b;
		}
	}
	__syncthreads();
	if (threadIdx.x == 0) {
		__threadfence();
		if (atomicAdd((uint32_t *)&ctl->foldDone, 1u) == gridDim.x - 1u) {
			for (uint32_t k = 0; k < n; ++k) {
				const uint32_t c = first + k + CRH_SF_RING;          /* the chunk that takes the freed slab */
				ctl->left[(first + k) & (CRH_SF_RING - 1u)] = c < Pl.chunkCount ? Pl.npix * (c + 1u == Pl.chunkCount ? Pl.lastPasses : Pl.passesPerChunk) : 0u;
			}
			{          /* what the shade kernel really handed out: its limit was the ring's end as it stood, or the dispatch's */
				const unsigned long long lim0 = (unsigned long long)(first + CRH_SF_RING) * (unsigned long long)Pl.chunkItems, lim = lim0 < Pl.genTotal ? lim0 : Pl.genTotal;
				if (ctl->genNext > lim) ctl->genNext = lim;
			}
			ctl->foldNext = first + n;
			ctl->foldDone = 0u;
			ctl->shadeCtr = 0u;
			for (uint32_t i = 0; i < CRH_SF_COUNTERS; ++i) ctl->walkCtr[i * CRH_SF_CTR_STRIDE] = 0u;
			ctl->liveIn = ctl->liveOut; ctl->liveOut = 0u;
			__threadfence();
			if (first + n == Pl.chunkCount) *(volatile unsigned int *)Pl.done = Pl.seq;
		}
	}
}
/*
 * walk_probe.h — k_walk_probe: the WALK of k_pathtrace_roll on its own (round 6, VERDICT r05 item 1, step A: a measurement, not a product path).
 *
 * The question: the path-tracing kernel is nailed at 4 waves per SIMD by 128 VGPRs + 40.9 KB of LDS, its waves are parked on memory half of their cycles, and every occupancy
 * experiment so far squeezed the WHOLE machine into fewer registers and measured spills. What does the walk — getClosestIsect, bvh.c:354-441 via pathtrace.c:26-30 — do when more
 * than four waves share a SIMD? This kernel is the same lane code (walkBegin / stepNodeLoaded / stepTriLoaded / stepCtrl / walkAdvance of pt_device.h), the same fused node run
 * with in-run instance entries and in-run retire + refill, the same ballot scheduling — without path generation, shading, the job ring or the fold: rays come in from a global
 * list, hits go out to a global list. The list is the path tracer's own: a counting dispatch of k_pathtrace_roll dumps every ray a wave starts to walk, wave by wave, in the
 * order it started them (crh_debug_ray_dump), so that a probe wave works through the rays ONE wave of the path tracer walked together — the coherence is the path tracer's.
 *
 * Template parameters: FORM = 1: the node run of pathtrace_roll.h as it is (node pairs and triangles requested together, six quarters carried through the run), 0: the lean
 * run (every step loads its own records), 3: the lean run with one site per step kind; WPS = waves per SIMD the register allocator must leave room for (and, with the LDS pad of the launch, the workgroups a CU holds); NLDS = traversal-stack
 * entries in LDS (deeper ones in the per-wave overflow columns, as in the render kernel); INST = line 0 of the instance records staged in LDS (scenes of <= 64 instances).
 * WPS = 0 is the reference form for the check: one ray per lane, traverse() to the end (k_trace_rays' loop), same output format — the probe's hits must equal it bit for bit.
 */
#pragma once

template <int NLDS, bool INST>
struct ProbeStack {
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
struct WalkOnlyCounters { static constexpr int level = 0; static constexpr bool programs = false; static constexpr bool wide = false; };

/* a unit of the probe's work queue: `count` consecutive rays of the list from `first` on (one wave's rays of the dump, cut into pieces) */
struct ProbeUnit { uint32_t first, count; };

template <int WPS, int NLDS, bool INST, int FORM>
__global__ __launch_bounds__(CRH_BLOCK, WPS) void k_walk_probe(const DScene Sarg, const float *raysArg, const ProbeUnit *unitsArg, uint32_t nUnits, uint32_t *unitCounter,
                                                               f4 *hitsArg, int32_t *hitInstArg, const Sched K, uint32_t *ovfAll) {
	__shared__ uint32_t s_stack[(NLDS > 0 ? NLDS : 1) * CRH_BLOCK];
	__shared__ uint32_t s_park[CRH_PARK_SLOTS * CRH_BLOCK];
	__shared__ __attribute__((aligned(16))) uint32_t s_inst0[INST ? CRH_INST_LDS0_MAX * 16u : 4u];
	/* (the launch adds dynamic LDS that pads the workgroup to 160 KB / WPS, so that a CU holds exactly WPS workgroups whatever the stack depth) */
	const DScene S = globalize(Sarg);
	const float *const rays = asGlobal(raysArg);
	const ProbeUnit *const units = asGlobal(unitsArg);
	f4 *const hits = (f4 *)(__attribute__((address_space(1))) f4 *)hitsArg;
	int32_t *const hitInst = (int32_t *)(__attribute__((address_space(1))) int32_t *)hitInstArg;
	ProbeStack<NLDS, INST> stk;
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
	const uint32_t trisOff = (uint32_t)((const char *)S.tris - (const char *)S.nodes);
	WalkOnlyCounters cnt;
	NullPort port;
	Walk w;
	memset(&w, 0, sizeof(w));
	w.phase = PH_IDLE;
	uint32_t myRay = 0;
	uint32_t cur = 0, end = 0;          /* wave-uniform: the rays of the unit in hand that have not started */
	bool dry = false;                   /* wave-uniform: the queue has no more units */
	/* retire + refill (k_pathtrace_roll: retireRefill): lanes whose walk ended write the hit of their ray; they and the idle lanes take the next rays of the unit in hand */
	auto retireRefill = [&]() __attribute__((always_inline)) {
		if (w.phase == PH_SHADE) {
			hits[myRay] = f4{w.hit.t, w.hit.u, w.hit.v, asF32((uint32_t)w.hit.slot)};
			hitInst[myRay] = w.hit.inst;
			w.phase = PH_IDLE;
		}
		const bool idle = (w.phase == PH_IDLE);
		const unsigned long long em = __ballot(idle);
		const uint32_t er = laneRank(em);
		if (cur == end && !dry) {
			uint32_t u = 0;
			if (lane == 0) u = atomicAdd((uint32_t *)(__attribute__((address_space(1))) uint32_t *)unitCounter, 1u);
			u = __builtin_amdgcn_readfirstlane(u);
			if (u < nUnits) { const ProbeUnit pu = units[u]; cur = __builtin_amdgcn_readfirstlane(pu.first); end = cur + __builtin_amdgcn_readfirstlane(pu.count); }
			else dry = true;
		}
		const uint32_t take = min(end - cur, (uint32_t)__popcll(em));
		if (idle && er < take) {
			myRay = cur + er;
			const float *r = rays + (size_t)myRay * 6u;
			walkBegin(S, w, stk, v3{r[0], r[1], r[2]}, v3{r[3], r[4], r[5]}, cnt, port, (uint32_t)K.rayFlags);
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
		if constexpr (FORM == 3) {
			/* the lean run with ONE site per step kind (the register allocator sees the largest step, not the sum of the copies that the round-level steps and the run's
			 * in-place steps inline): the round picks a mode by the render kernel's rules — 0 a node run, 1 a triangle run, 2 one control step, 3 retire + refill — and one loop
			 * body serves all four */
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
			continue;
		}
		if (walkers == 0 || (nF + nE >= K.swapMin && (nF > 0 || more))) { retireRefill(); continue; }
		int best = nN * K.wNode, pick = 0;
		if (nT * K.wTri > best) { best = nT * K.wTri; pick = 1; }
		if (nC * K.wCtrl > best) { best = nC * K.wCtrl; pick = 2; }
		if (pick == 0) {
			int now = nN;
			if constexpr (FORM == 0) {
				/* the LEAN run: the same steps and thresholds, but every step requests its own records and waits for them (the unfused loop of round 3) — no six quarters live
				 * across the run, which is what lets the register allocator go below 100 registers without spilling */
				do {
					if (w.phase == PH_NODE) stepNode<true>(S, w, stk, cnt, port);
					if ((int)__popcll(__ballot(w.phase == PH_TRI)) >= K.triInRun) { if (w.phase == PH_TRI) stepTri(S, w, stk, cnt, port); }
					if ((int)__popcll(__ballot(w.phase == PH_CTRL)) >= K.ctrlInRun) { if (w.phase == PH_CTRL) stepCtrl(S, w, stk, cnt, port); }
					{
						const int nFi = (int)__popcll(__ballot(w.phase == PH_SHADE)), nEi = (int)__popcll(__ballot(w.phase == PH_IDLE));
						if (nFi + nEi >= K.swapInRun && (nFi > 0 || !dry || cur != end)) retireRefill();
					}
					now = __popcll(__ballot(w.phase == PH_NODE));
				} while (now * 8 >= nN * K.runNum);
			} else {
			f4 q0 = f4{0.0f, 0.0f, 0.0f, 0.0f}, q1 = q0, q2 = q0, q3 = q0, q4 = q0, q5 = q0;
			do {          /* the fused node run of pathtrace_roll.h (ST_NODE), step for step */
				const bool isN = w.phase == PH_NODE;
				const int nTw = (int)__popcll(__ballot(w.phase == PH_TRI));
				const bool isT = nTw >= K.triInRun && w.phase == PH_TRI;
				if (isN || isT) {
					const uint32_t off = isN ? (uint32_t)(w.node << 5) : trisOff + w.pA * 48u;
					const char *rec = (const char *)S.nodes + off;
					q0 = *(const f4 *)rec; q1 = *(const f4 *)(rec + 16); q2 = *(const f4 *)(rec + 32); q3 = *(const f4 *)(rec + 48);
					if (isT) { q4 = *(const f4 *)(rec + 64); q5 = *(const f4 *)(rec + 80); }
				}
				if (isN) stepNodeLoaded<true>(S, w, stk, cnt, port, q0, q1, q2, q3);
				if (isT) stepTriLoaded(S, w, stk, cnt, port, q0, q1, q2, q3, q4, q5);
				if ((int)__popcll(__ballot(w.phase == PH_CTRL)) >= K.ctrlInRun) {
					if (w.phase == PH_CTRL) stepCtrl(S, w, stk, cnt, port);
				}
				{
					const int nFi = (int)__popcll(__ballot(w.phase == PH_SHADE)), nEi = (int)__popcll(__ballot(w.phase == PH_IDLE));
					if (nFi + nEi >= K.swapInRun && (nFi > 0 || !dry || cur != end)) retireRefill();
				}
				now = __popcll(__ballot(w.phase == PH_NODE));
			} while (now * 8 >= nN * K.runNum || (int)__popcll(__ballot(w.phase == PH_TRI)) >= K.triInRun);
			}
		} else if (pick == 1) {
			int now = nT;
			do {
				if (w.phase == PH_TRI) stepTri(S, w, stk, cnt, port);
				now = __popcll(__ballot(w.phase == PH_TRI));
			} while (now * 8 >= nT * K.runNum);
		} else {
			if (ph == PH_CTRL) stepCtrl(S, w, stk, cnt, port);
			if (__ballot(ph == PH_NODE_SLOW)) { if (ph == PH_NODE_SLOW) stepNodeAny<false>(S, w, stk, cnt, port); }
		}
	}
}

/* the check's reference form: one ray per lane from start to end (k_trace_rays' loop), same output */
__global__ __launch_bounds__(CRH_BLOCK) void k_walk_simple(const DScene Sarg, const float *rays, uint64_t n, const unsigned long long *regionCounts, uint32_t regionCap, f4 *hits, int32_t *hitInst, uint32_t rayFlags) {
	__shared__ uint32_t s_stack[CRH_STACK_LDS * CRH_BLOCK];
	__shared__ uint32_t s_park[CRH_PARK_SLOTS * CRH_BLOCK];
	const DScene S = globalize(Sarg);
	LdsStackPrivate stk;
	stk.lds = (lds_u32 *)&s_stack[threadIdx.x];
	stk.parkp = (lds_u32 *)&s_park[threadIdx.x];
	for (uint64_t i = (uint64_t)blockIdx.x * CRH_BLOCK + threadIdx.x; i < n; i += (uint64_t)gridDim.x * CRH_BLOCK) {
		if (i % regionCap >= regionCounts[i / regionCap]) continue;          /* (the list is one region per wave of the dump: only its first `count` slots hold rays) */
		WalkOnlyCounters cnt;
		const v3 o{rays[6 * i], rays[6 * i + 1], rays[6 * i + 2]}, d{rays[6 * i + 3], rays[6 * i + 4], rays[6 * i + 5]};
		TravHit h;
		traverse(S, stk, o, d, h, cnt, rayFlags);
		hits[i] = f4{h.t, h.u, h.v, asF32((uint32_t)h.slot)};
		hitInst[i] = h.inst;
	}
}

/* how many of n hits differ (bit patterns) between two outputs */
__global__ void k_probe_compare(const f4 *a, const int32_t *ai, const f4 *b, const int32_t *bi, uint64_t n, unsigned long long *differ) {
	unsigned long long d = 0;
	for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
		const f4 x = a[i], y = b[i];
		if (asU32(x.x) != asU32(y.x) || asU32(x.y) != asU32(y.y) || asU32(x.z) != asU32(y.z) || asU32(x.w) != asU32(y.w) || ai[i] != bi[i]) ++d;
	}
	if (d) atomicAdd(differ, d);
}

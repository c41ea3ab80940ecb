/*
 * pathtrace_alt.h — the two ALTERNATIVE forms of the path-tracing kernel, compiled only with -DCRH_WITH_ALT_KERNELS (the kernel emulation of tests/emu
 * and A/B variant libraries; the product library holds k_pathtrace_roll alone since round 4):
 *   k_pathtrace     (CRH_KERNEL_WAVE) the wave machine working ONE unit at a time — the default until round 3, the form the rolling kernel is proven bit-identical to;
 *   k_pathtrace_wg  (CRH_KERNEL_WG)   the workgroup-cooperative form (walker / shader roles, shared path table) — measured 2-27 % slower (profiles/r02_probe_wg_kernel.log).
 * Same lane code (pt_device.h), same id stacks and scheduler rules, the same frame bit for bit.
 */
#pragma once

template <int LEVEL, int WPS, bool PROG, int SAMP>
__global__ __launch_bounds__(CRH_BLOCK, CRH_WPS_OVERRIDE) void k_pathtrace(const DScene Sarg, const crh_render_params P, const BlockQueue Q, float *fb,
														   unsigned long long *counters,
														   float *stage, int chunk, unsigned long long *waveStats, const Sched K, float *queues, uint32_t *ovfAll) {
	__shared__ uint32_t s_stack[CRH_STACK_LDS * CRH_BLOCK];
	__shared__ uint32_t s_park[CRH_PARK_SLOTS * CRH_BLOCK];
	static_assert((CRH_STACK_LDS + CRH_PARK_SLOTS) * CRH_BLOCK * 4 + (CRH_BLOCK / 64) * (CRH_IDS_BYTES + 32) + 512 + 256 + CRH_INST_LDS_BYTES + CRH_SHADE_LDS_BYTES <= 40960, "4 blocks per CU share 160 KB of LDS (incl. powf's tables)");
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
	CRH_STAGE_SHADE_TABLES();
	CRH_STAGE_INSTANCE_TABLES();
	/* the wave's slab and path table start at wave-uniform addresses: said so (readfirstlane), their accesses use a scalar base + a 32-bit lane offset (profiles/r03q_ab_uniform_bases.log) */
	float *myStage = stage + (size_t)__builtin_amdgcn_readfirstlane(wave) * ((size_t)Q.bw * Q.bh * chunk * 3);
	const int passEnd = P.first_pass + P.pass_count;
	/* the wave's path table, its id stacks and their wave-uniform fill levels (LDS: lane 0 writes, every lane reads; as
	 * plain variables they would be scalar registers live across the whole machine, and the register allocator is past
	 * its limits there — measured slower, and wrong images in the variant that calls runProgram) */
	f4 *const ptab = (f4 *)(queues + (size_t)__builtin_amdgcn_readfirstlane(wave) * CRH_WAVE_QUEUE_FLOATS);
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
				/* (wave-uniform by construction — lane 0 wrote them — but left as vector values: the decision below then compiles to exec-masked straight-line
				 * code. Declared uniform with readfirstlane — a scalar decision, a scalar step switch, 22 instead of 26 spilled VGPRs — it is 2-7 % SLOWER on every
				 * scene: the scalar form waits for the five LDS words before anything else and takes a chain of branches; profiles/r03o_ab_scalar_sched.log) */
				const int raysQ = wq[WQ_RAYS], hitsQ = wq[WQ_HITS], missQn = wq[WQ_MISSES], freeQ = wq[WQ_FREE];
				const uint32_t nextItem = (uint32_t)wq[WQ_NEXT_ITEM];
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
							if (w.phase == PH_NODE) stepNode<true>(S, w, stk, cnt, port);
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
							{ TablePort<SAMP> port{ptab + myPath * CRH_PATH_F4}; walkBegin(S, w, stk, v3{q0.x, q0.y, q0.z}, v3{q1.x, q1.y, q1.z}, cnt, port, (uint32_t)K.rayFlags); }
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
#define CRH_WG_STACK_LDS 17          /* (17 + 15 park) x 1 KB + 2 x 2 KB id arrays + control words <= 40 KB: 4 workgroups per CU */

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


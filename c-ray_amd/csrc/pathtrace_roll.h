/*
 * pathtrace_roll.h — k_pathtrace_roll: the wave machine of k_pathtrace (cray_hip.hip) with ROLLING work units.
 * THE DEFAULT KERNEL FORM since the end of round 3 (CRH_OPT_KERNEL = CRH_KERNEL_ROLL). Built and verified in round 2, when it was 8-14 % slower than
 * k_pathtrace on hdr.json and 2-4 % faster on the soup (the walk was short of L1 look-ups and waiting for cache misses: always-full path tables are a bigger
 * working set). Round 3 moved the hot records into LDS and the address arithmetic onto the scalar unit: vector issue binds the kernel now, fuller steps pay,
 * and this form equals k_pathtrace on hdr.json at 256 spp and beats it by 6-30 % wherever a dispatch holds few passes (profiles/r03t_ab_rolling_units.log).
 *
 * Why. k_pathtrace works a unit (a pixel block x a chunk of passes) to its last path before it pulls the next one: every unit ramps the
 * wave's path table up and drains it again, and while it drains the walk steps run with ever fewer lanes. The kernel emulation
 * (tests/emu, tools/emu_sched_stats.py) prices that exactly: on hdr.json the default 2048-path units take 1.049 M node steps at 38.4
 * lanes per step where 32768-path units take 0.912 M at 44.2 — ramp and drain are 13 % of all node steps, 10 % of the modelled time —
 * but bigger units are slower on the GPU (r02_probe_unit_size.log: their sample slabs and their coarser tail cost more than they save).
 * Rolling units take the drain away without growing the unit: a wave keeps up to CRH_ROLL_SLOTS jobs open (a ring). When the items of
 * the job it generates from run out, it pulls the next job into the next free slot and keeps the path table full from there, while the
 * last paths of the older jobs finish; a job is folded into the frame when its last sample is staged (in job order, so a pixel's chunks
 * still fold in pass order), which frees its slot. Only the last jobs of a wave drain.
 *
 * What stays: every step of a path — and so every result — is the one k_pathtrace runs (same lane code, same id stacks, same scheduler
 * rules); the frame is the same bit for bit (tests/test_kernel_emu.py on the CPU, tests/test_gpu_parity.py on the GPU).
 * Job state lives in LDS next to the stack fill levels (wave-uniform: lane 0 writes, every lane reads); the hit stack is 136 entries
 * instead of 192 (it never holds more than 64 waiting + 64 retired hits), which pays for it: the LDS footprint stays below 40 KB.
 */
#pragma once

#define CRH_ROLL_HITS_MAX 136u
#define CRH_ROLL_IDS_MISSES (CRH_IDS_HITS + 2u * CRH_ROLL_HITS_MAX)
#define CRH_ROLL_IDS_BYTES (CRH_ROLL_IDS_MISSES + 128u)
#ifndef CRH_ROLL_SLOTS
#define CRH_ROLL_SLOTS 4u                      /* jobs a wave keeps open (a ring: opened and folded in order); 2..4 */
#endif
#ifndef CRH_ROLL_FIFO
#define CRH_ROLL_FIFO 0                        /* 1: rays are walked oldest first (measured in the emulation: no fewer steps — the last paths of a job are late
                                                * because they are long, not because they wait under newer rays) */
#endif
#define CRH_ROLL_SLOT_SHIFT 30u                /* item word of a path record: slot of its job << 30 | item index inside the job */
#define CRH_ROLL_ITEM_MASK 0x3FFFFFFFu


/* WIDE (round 5, CRH_OPT_WALK = CRH_WALK_WIDE4): the walk steps through the 4-ary copy of the BVHs (pt_device.h: stepWideLoaded; Sarg.tlas_root is the wide root) */
template <int LEVEL, int WPS, bool PROG, int SAMP, bool WIDE = false>
__global__ __launch_bounds__(CRH_BLOCK, CRH_WPS_OVERRIDE) void k_pathtrace_roll(const DScene Sarg, const crh_render_params P, const BlockQueue Q, float *fb,
																 unsigned long long *counters,
																 float *stage, int chunk, unsigned long long *waveStats, const Sched K, float *queues, uint32_t *ovfAll, unsigned int *errFlag) {
	__shared__ uint32_t s_stack[CRH_STACK_LDS * CRH_BLOCK];
	__shared__ uint32_t s_park[CRH_PARK_SLOTS * CRH_BLOCK];
	/* wave-uniform words: stack fill levels, shade-class counts, the ring of job slots */
	enum { RQ_RAYS, RQ_HITS, RQ_MISSES, RQ_FREE, RQ_CLS_LO, RQ_CLS_HI,
	       RQ_GENLEFT,   /* items not yet generated of the youngest open job (0: none open, or all generated) */
	       RQ_FLAGS,     /* bit 0: the oldest open job is complete (all items generated, every sample staged) -> fold it;
	                      * bit 1: a slot is free and the work queue is not known to be empty -> the next job can be opened.
	                      * Both words are derived from the ones below by lane 0 whenever those change (refreshJobWords), so that the
	                      * scheduler reads two words per round instead of the slots */
	       RQ_HEAD,      /* the open jobs are the slots head, head + 1, ... (mod CRH_ROLL_SLOTS): head folds first, ... */
	       RQ_OPEN,      /* ... this many of them; ... */
	       RQ_GEN,       /* ... items are generated from the youngest: slot (head + open - 1) mod CRH_ROLL_SLOTS */
	       RQ_DRY,       /* the work queue is empty */
	       RQ_SLOT0 };
	enum { SJ_X0, SJ_Y0, SJ_WH /* w | h << 16 */, SJ_BWBH /* bw | bh << 16 */, SJ_PASS0, SJ_PASSN, SJ_NEXT, SJ_OUT,     /* SJ_OUT: paths generated, sample not yet staged */
	       SJ_BASE_LO, SJ_BASE_HI,   /* where the job's samples wait for their fold (a global address): sample of item i at base + 12 i. A block job's slab — or, for a pass
	                                  * segment of a split pixel (BlockQueue: firstMicro, segs > 1; SJ_WH bit 30), the segment's place in Q.defer: k_fold_deferred folds those */
	       SJ_WORDS };
	enum { SJ_WH_DEFER = 1 << 30 };
	enum { SJ_WH_H_MASK = 0x3FFF };
#define CRH_RQ_IS_DRY(v) ((v) != 0)
	enum { NS = (int)CRH_ROLL_SLOTS, RQ_WORDS = RQ_SLOT0 + NS * SJ_WORDS };
	static_assert(NS >= 2 && NS <= 4, "2..4 job slots (two bits of the item word)");
	static_assert((CRH_STACK_LDS + CRH_PARK_SLOTS) * CRH_BLOCK * 4 + (CRH_BLOCK / 64) * (CRH_ROLL_IDS_BYTES + RQ_WORDS * 4) + 512 + 256 + CRH_INST_LDS_BYTES + CRH_SHADE_LDS_BYTES <= 40960, "4 blocks per CU share 160 KB of LDS (incl. powf's tables)");
	const DScene S = globalize(Sarg);
	CRH_EM_POW_TABLES_INIT();
	const unsigned long long tStart = wall_clock64();
	uint32_t unitsDone = 0;
#ifdef CRH_EXP_ABS_TIMES       /* dev probe (tools/probe_finish.py, a variant library): when — on the chip-wide clock — a wave starts, first finds the work queue empty, and ends */
	unsigned long long tDry = 0;
	bool snapDone = false;          /* ... and what it holds at the first scheduling round (checked every 16th) at which the work counter has passed the last unit: waveStats[2 * (waves + wave)] */
#endif
	LdsStack stk;
	stk.lds = (lds_u32 *)&s_stack[threadIdx.x];
	stk.parkp = (lds_u32 *)&s_park[threadIdx.x];
	CRH_STAGE_SHADE_TABLES();
	CRH_STAGE_INSTANCE_TABLES();
	CountersT<LEVEL, PROG, WIDE> cnt;
	memset(&cnt, 0, sizeof(cnt));
	const uint32_t lane = threadIdx.x & 63u;
	const uint32_t wave = (blockIdx.x * CRH_BLOCK + threadIdx.x) >> 6;
	__shared__ uint8_t s_cls[256];
	const bool sorted = K.sortFrom > 0 && S.shade_classes >= (uint32_t)K.sortFrom && S.instance_count <= 256u;        /* see k_pathtrace */
	if (sorted) {
		for (uint32_t i = threadIdx.x; i < S.instance_count; i += CRH_BLOCK) s_cls[i] = (uint8_t)CRH_DINST_CLASS(S.instances[i].kind);
		__syncthreads();
	}
	stk.ovf = (glb_u32 *)ovfAll + (size_t)__builtin_amdgcn_readfirstlane(wave) * CRH_OVF_WORDS_PER_WAVE;
	const size_t slabFloats = (size_t)Q.bw * Q.bh * chunk * 3;           /* one job's samples; a wave owns one slab per slot */
	float *const myStage = stage + (size_t)__builtin_amdgcn_readfirstlane(wave) * (size_t)NS * slabFloats;
	const int passEnd = P.first_pass + P.pass_count;
	const uint32_t trisOff = (uint32_t)((const char *)S.tris - (const char *)S.nodes);          /* (one allocation: crh_scene_upload) */
	f4 *const ptab = (f4 *)(queues + (size_t)__builtin_amdgcn_readfirstlane(wave) * CRH_WAVE_QUEUE_FLOATS);
	/* the counting kernel's wave-level numbers (steps, lanes served, clocks per step kind: indices 8.. of the counter block) go straight to this wave's own words behind
	 * the global counters — one fire-and-forget atomic by lane 0 per event, no contention — instead of through two dozen registers per lane that are summed at the end:
	 * those registers were 70 spilled VGPRs, and the counting kernel's step clocks were not the timed kernel's */
	unsigned long long *const waveCtr = counters + CRH_NCOUNTERS + (size_t)__builtin_amdgcn_readfirstlane(wave) * CRH_NCOUNTERS;          /* (64-bit words: a busy wave's clocks wrap 32 bits in two seconds) */
	/* round 6, the counting instantiations only (crh_debug_ray_dump; walk_probe.h): every ray this wave starts to walk goes — in the order it starts them — into the wave's own
	 * region of a global list, so that a walk-only kernel can be timed on the path tracer's own rays in the path tracer's own order. The descriptor sits behind the wave
	 * statistics (CRH_OPT_WAVE_STATS): list address, rays per region, then one count per wave. */
	float *dumpRays = nullptr;
	uint32_t dumpCap = 0, dumpCount = 0;
	if constexpr (LEVEL >= 2) {
		if (waveStats) {
			const unsigned long long base = waveStats[CRH_DUMP_HDR];
			if (base) { dumpCap = (uint32_t)waveStats[CRH_DUMP_HDR + 1]; dumpRays = (float *)(__attribute__((address_space(1))) float *)(uintptr_t)base + (size_t)__builtin_amdgcn_readfirstlane(wave) * dumpCap * 6u; }
		}
	}
#define CRH_WCTR(k, v) atomicAdd(&waveCtr[k], (unsigned long long)(uint32_t)(v))
	__shared__ int s_rq[(CRH_BLOCK / 64) * RQ_WORDS];
	__shared__ __attribute__((aligned(2))) uint8_t s_ids[(CRH_BLOCK / 64) * CRH_ROLL_IDS_BYTES];
	typedef volatile __attribute__((address_space(3))) int lds_int;
	typedef volatile __attribute__((address_space(3))) uint8_t lds_u8;
	typedef volatile __attribute__((address_space(3))) uint16_t lds_u16;
	lds_int *const wq = (lds_int *)&s_rq[(threadIdx.x >> 6) * RQ_WORDS];
	lds_u8 *const ids = (lds_u8 *)&s_ids[(threadIdx.x >> 6) * CRH_ROLL_IDS_BYTES];
	lds_u16 *const hits = (lds_u16 *)&s_ids[(threadIdx.x >> 6) * CRH_ROLL_IDS_BYTES + CRH_IDS_HITS];
	auto loadJob = [&](int s) {
		lds_int *j = wq + RQ_SLOT0 + s * SJ_WORDS;
		BlockJob J;
		const int wh = j[SJ_WH], bwbh = j[SJ_BWBH];
		J.x0 = j[SJ_X0]; J.y0 = j[SJ_Y0]; J.w = wh & 0xFFFF; J.h = (wh >> 16) & SJ_WH_H_MASK; J.bw = bwbh & 0xFFFF; J.bh = bwbh >> 16; J.passBegin = j[SJ_PASS0]; J.passCount = j[SJ_PASSN];
		return J;
	};
	/* where a finished path's sample waits for its fold (SJ_BASE) */
	auto sampleSlot = [&](int slot, uint32_t idx) -> float * {
		const uint32_t lo = (uint32_t)wq[RQ_SLOT0 + slot * SJ_WORDS + SJ_BASE_LO], hi = (uint32_t)wq[RQ_SLOT0 + slot * SJ_WORDS + SJ_BASE_HI];
		return (float *)(__attribute__((address_space(1))) float *)(uintptr_t)(((unsigned long long)hi << 32) | lo) + idx * 3u;
	};
	auto jobDeferred = [&](int slot) { return (wq[RQ_SLOT0 + slot * SJ_WORDS + SJ_WH] & SJ_WH_DEFER) != 0; };
	/* lane 0, after it changed a job's words: the two words the scheduler reads every round */
	auto refreshJobWords = [&]() {
		const int head = wq[RQ_HEAD], open = wq[RQ_OPEN];
		int left = 0, flags = 0;
		bool moreChunks = false;             /* the youngest job's unit has passes left: the next job is its following chunk, whatever the queue holds */
		if (open > 0) {
			lds_int *gj = wq + RQ_SLOT0 + wq[RQ_GEN] * SJ_WORDS, *oj = wq + RQ_SLOT0 + head * SJ_WORDS;
			const int gb = gj[SJ_BWBH], ob = oj[SJ_BWBH];
			left = (gb & 0xFFFF) * (gb >> 16) * gj[SJ_PASSN] - gj[SJ_NEXT];
			if (left < 0) left = 0;
			moreChunks = !(gj[SJ_WH] & SJ_WH_DEFER) && gj[SJ_PASS0] + gj[SJ_PASSN] < passEnd;          /* (a pass segment is a unit of its own) */
			/* a job that is the only open one AND not its unit's last chunk stays open until the following chunk has been opened from it
			 * (ST_OPEN reads the unit from the youngest job): folded earlier, the unit's remaining passes would never be generated */
			if (oj[SJ_NEXT] >= (ob & 0xFFFF) * (ob >> 16) * oj[SJ_PASSN] && oj[SJ_OUT] == 0 && (open > 1 || !moreChunks)) flags |= 1;
		}
		if (open < NS && (!CRH_RQ_IS_DRY(wq[RQ_DRY]) || moreChunks)) flags |= 2;
		wq[RQ_GENLEFT] = left; wq[RQ_FLAGS] = flags;
	};

	if (lane == 0) {
		wq[RQ_RAYS] = 0; wq[RQ_HITS] = 0; wq[RQ_MISSES] = 0; wq[RQ_FREE] = (int)CRH_PATHS; wq[RQ_CLS_LO] = 0; wq[RQ_CLS_HI] = 0;
		wq[RQ_HEAD] = 0; wq[RQ_OPEN] = 0; wq[RQ_GEN] = 0; wq[RQ_DRY] = 0;
		for (int i = 0; i < NS * SJ_WORDS; ++i) wq[RQ_SLOT0 + i] = 0;
		refreshJobWords();
	}
	for (uint32_t i = lane; i < CRH_PATHS; i += 64u) ids[i] = (uint8_t)i;              /* all slots free */
	__threadfence_block();
	Walk w;
	memset(&w, 0, sizeof(w));
	w.phase = PH_IDLE;
	uint32_t myPath = 0;
	/* a wave that has run K.roundLimit scheduling rounds without finishing gives up — never a hung GPU — and says so: the dispatch's error word (host-visible) makes
	 * crh_synchronize / crh_framebuffer_download return CRH_ERR_HIP "incomplete frame" (the default limit is hours of one wave's work; CRH_OPT_ROUND_LIMIT) */
	/* retire + refill (ST_SWAP; round 4: also inside node runs): lanes whose walk ended leave the result in their path's record and push its id on the hit or the miss stack;
	 * they and the idle lanes pop ray ids and start those walks. rq / hq / mq: the ray, hit and miss stack levels, updated here and in LDS */
	auto retireRefill = [&](int &rq, int &hq, int &mq) __attribute__((always_inline)) {
		const bool fin = (w.phase == PH_SHADE);
		const bool finHit = fin && w.hit.inst >= 0, finMiss = fin && w.hit.inst < 0;
		const unsigned long long hm = __ballot(finHit), mm = __ballot(finMiss);
		uint32_t cls = 0;
		if (fin) {
			f4 *q = ptab + myPath * CRH_PATH_F4;
			q[4] = f4{w.hit.t, w.hit.u, w.hit.v, asF32((uint32_t)w.hit.slot)};
			if (finHit) {
				q[5].x = asF32((uint32_t)w.hit.inst);
				if (sorted) cls = (uint32_t)((volatile __attribute__((address_space(3))) uint8_t *)s_cls)[w.hit.inst];
				hits[(uint32_t)hq + laneRank(hm)] = (uint16_t)(myPath | (cls << 8));
			} else {
				ids[CRH_ROLL_IDS_MISSES + (uint32_t)mq + laneRank(mm)] = (uint8_t)myPath;
			}
			w.phase = PH_IDLE;
		}
		if (sorted && hm) {
			uint32_t addLo = 0, addHi = 0;
#pragma unroll
			for (uint32_t b = 0; b < 8u; ++b) {
				const uint32_t nb = (uint32_t)__popcll(__ballot(finHit && cls == b));
				if (b < 4u) addLo += nb << (8u * b); else addHi += nb << (8u * (b - 4u));
			}
			if (lane == 0) { wq[RQ_CLS_LO] = wq[RQ_CLS_LO] + (int)addLo; wq[RQ_CLS_HI] = wq[RQ_CLS_HI] + (int)addHi; }
		}
		const bool idle = (w.phase == PH_IDLE);
		const unsigned long long em = __ballot(idle);
		const uint32_t er = laneRank(em);
		const int take = min(rq, (int)__popcll(em));
#if CRH_ROLL_FIFO
		/* oldest rays first (k_pathtrace pops the newest): the last paths of an older job must not wait under the rays of the
		 * job being generated, or its fold — and with it its slot — is held up for that job's whole length. The remaining ids move
		 * down to close the gap: every entry is read, then written (LDS operations of a wave execute in program order). */
		uint32_t myRay = 0;
		if (idle && (int)er < take) myRay = ids[CRH_IDS_RAYS + er];
		{
			uint32_t mv[4];
#pragma unroll
			for (int k = 0; k < 4; ++k) { const uint32_t i = (uint32_t)take + (uint32_t)k * 64u + lane; mv[k] = (int)i < rq ? (uint32_t)ids[CRH_IDS_RAYS + i] : 0u; }
			CRH_LOCKSTEP();
#pragma unroll
			for (int k = 0; k < 4; ++k) { const uint32_t i = (uint32_t)take + (uint32_t)k * 64u + lane; if ((int)i < rq) ids[CRH_IDS_RAYS + i - (uint32_t)take] = (uint8_t)mv[k]; }
		}
		if (idle && (int)er < take) {
			myPath = myRay;
#else
		if (idle && (int)er < take) {
			myPath = ids[CRH_IDS_RAYS + (uint32_t)(rq - take) + er];
#endif
			const f4 *q = ptab + myPath * CRH_PATH_F4;
			const f4 q0 = q[0], q1 = q[1];
			{ TablePort<SAMP> port2{ptab + myPath * CRH_PATH_F4}; walkBegin(S, w, stk, v3{q0.x, q0.y, q0.z}, v3{q1.x, q1.y, q1.z}, cnt, port2, (uint32_t)K.rayFlags); }
			if constexpr (LEVEL >= 2) {
				if (dumpRays && dumpCount + er < dumpCap) { float *o = dumpRays + (size_t)(dumpCount + er) * 6u; o[0] = q0.x; o[1] = q0.y; o[2] = q0.z; o[3] = q1.x; o[4] = q1.y; o[5] = q1.z; }
			}
		}
		if constexpr (LEVEL >= 2) dumpCount += (uint32_t)take;
		hq += (int)__popcll(hm); mq += (int)__popcll(mm); rq -= take;
		if (lane == 0) { wq[RQ_HITS] = hq; wq[RQ_MISSES] = mq; wq[RQ_RAYS] = rq; }
		__threadfence_block();
	};
	uint32_t guard = (uint32_t)K.roundLimit;
	for (;;) {
		if (--guard == 0u) { if (lane == 0) atomicOr(errFlag, CRH_ERRFLAG_ROUND_LIMIT); break; }
#ifdef CRH_EXP_ABS_TIMES
		if (waveStats && !snapDone && (guard & 15u) == 0u) {
			uint32_t handedOut = 0;
			if (lane == 0) handedOut = __hip_atomic_load((uint32_t *)Q.counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
			handedOut = __builtin_amdgcn_readfirstlane(handedOut);
			if (handedOut >= Q.total) {
				snapDone = true;
				if (lane == 0) {
					const int head = wq[RQ_HEAD], open = wq[RQ_OPEN];
					unsigned long long notGenerated = 0, notStaged = 0;
					for (int k = 0; k < open; ++k) {
						lds_int *oj = wq + RQ_SLOT0 + ((head + k) % NS) * SJ_WORDS;
						const int ob = oj[SJ_BWBH], items = (ob & 0xFFFF) * (ob >> 16) * oj[SJ_PASSN];
						notGenerated += (unsigned long long)max(items - oj[SJ_NEXT], 0);
						notStaged += (unsigned long long)oj[SJ_OUT];
					}
					const size_t waves = (size_t)gridDim.x * (CRH_BLOCK / 64u);
					waveStats[2 * (waves + wave)] = ((wall_clock64() - tStart) << 32) | (notGenerated & 0xFFFFFFFFull);
					waveStats[2 * (waves + wave) + 1] = ((unsigned long long)open << 48) | ((unsigned long long)((int)CRH_PATHS - wq[RQ_FREE]) << 32) | (notStaged & 0xFFFFFFFFull);
				}
			}
		}
#endif
		const uint32_t ph = w.phase;
		TablePort<SAMP> port{ptab + myPath * CRH_PATH_F4};
		const int nN = __popcll(__ballot(ph == PH_NODE)), nT = __popcll(__ballot(ph == PH_TRI)), nC = __popcll(__ballot(ph == PH_CTRL || ph == PH_NODE_SLOW));
		const int nF = __popcll(__ballot(ph == PH_SHADE));
		const int nE = 64 - nN - nT - nC - nF;
		const int raysQ = wq[RQ_RAYS], hitsQ = wq[RQ_HITS], missQn = wq[RQ_MISSES], freeQ = wq[RQ_FREE];
		const int itemsLeft = wq[RQ_GENLEFT], jobFlags = wq[RQ_FLAGS];
		CRH_LOCKSTEP();               /* every lane has read the wave's words before lane 0 updates them at the end of the step */
		const bool genLeft = itemsLeft > 0;
		const bool canGen = genLeft && freeQ >= 64;
		/* the next job can be opened when the job being generated has no items left and a slot is free */
		const bool canOpen = !genLeft && (jobFlags & 2);
		const bool foldReady = (jobFlags & 1) != 0;
		const bool wantGen = raysQ < 64 && (((int)CRH_PATHS - freeQ) < K.fillTo || (nE + nF > 0 && raysQ < nE + nF));
		const int walkers = nN + nT + nC;
		enum { ST_NODE, ST_TRI, ST_CTRL, ST_SWAP, ST_GEN, ST_SHADE, ST_MISS, ST_OPEN, ST_FOLD, ST_END };
		int pick;
		if (foldReady) pick = ST_FOLD;
		else if (hitsQ >= 64) pick = ST_SHADE;
		else if (missQn >= 64) pick = ST_MISS;
		else if (nF + nE >= K.swapMin && (nF > 0 || (nE > 0 && raysQ > 0))) pick = ST_SWAP;
		else if (wantGen && canGen) pick = ST_GEN;
		else if (wantGen && canOpen && freeQ >= 64) pick = ST_OPEN;
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
		else if (canOpen) pick = ST_OPEN;
		else pick = ST_END;               /* nothing in flight, nothing queued, no job open, the work queue empty */
		if (pick == ST_END) break;
		uint32_t tk = 0;
		if constexpr (LEVEL >= 2) tk = CRH_TICK();
		switch (pick) {
			case ST_NODE: {
				int now = nN;
				/* Fused walk steps (round 4): a lane that descends needs a child pair, a lane at a leaf its next two triangles — both kinds of record are requested at the
				 * top of an iteration and waited for ONCE, then the node lanes step, then (when enough of them wait: K.triInRun) the triangle lanes. The steps, their order per
				 * path and their thresholds are the unfused loop's (round 3; git history); what changes is that a triangle step's memory round trip overlaps the node step's instead of
				 * following it — and that a lane does ONE step per iteration: 5-10 % more iterations, each with one round trip instead of up to two. Measured (profiles/r04b_ab_*.log,
				 * r04c_ab_kept.log): hdr.json +1.4...2.6 %, statues +0.2...0.9 %, 1 M soup +1.0 %, venus -0.9...-1.3 %; the share of a frame's last 2 ms does not move (r04d_probe_share8.log). */
				int rqRun = raysQ, hqRun = hitsQ, mqRun = missQn;          /* the stack levels as this run's in-place retire / refill steps leave them */
				f4 q0 = f4{0.0f, 0.0f, 0.0f, 0.0f}, q1 = q0, q2 = q0, q3 = q0, q4 = q0, q5 = q0;          /* (defined once per run: a lane reads only what it loaded in the same iteration, but an
				                                                                                            * undefined value that meets a loaded one at every join costs the register allocator 70 spills) */
				f4 q6 = q0, q7 = q0;          /* (WIDE: a wide node is eight quarters) */
				do {
					const bool isN = w.phase == PH_NODE;
					const int nTw = (int)__popcll(__ballot(w.phase == PH_TRI));
					const bool isT = nTw >= K.triInRun && w.phase == PH_TRI;
					if constexpr (LEVEL >= 2) {
						if (lane == 0) { CRH_WCTR(11, 1); CRH_WCTR(17, now); }
#ifdef CRH_CENSUS          /* the node-run census (tools/emu_sched_stats.py: who sits a node step out, and why): six more per-lane counters, which the kernel emulation can afford and the
                            * counting kernel on the GPU cannot — with them its spilled VGPRs went from 60 to 95-176 and its step clocks stopped resembling the timed kernel's */
						const uint32_t wf = (uint32_t)__popcll(__ballot(w.phase == PH_SHADE || w.phase == PH_IDLE));
						if (lane == 0) { CRH_WCTR(30, nTw >= K.triInRun ? 0u : (uint32_t)nTw); CRH_WCTR(31, wf); if (nTw >= K.triInRun) { CRH_WCTR(26, 1); CRH_WCTR(27, nTw); } }
#endif
					}
					/* one 32-bit byte offset from S.nodes for either kind of record (crh_scene_upload puts the triangles behind the nodes in the same allocation), six
					 * quarters in the same registers: a child pair is the first four, two triangles (48 bytes each, consecutive) all six */
					if (isN || isT) {
						const uint32_t off = isN ? (uint32_t)(w.node << (WIDE ? 4 : 5)) : trisOff + w.pA * 48u;
						const char *rec = (const char *)S.nodes + off;
						q0 = *(const f4 *)rec; q1 = *(const f4 *)(rec + 16); q2 = *(const f4 *)(rec + 32); q3 = *(const f4 *)(rec + 48);
						if (WIDE || isT) { q4 = *(const f4 *)(rec + 64); q5 = *(const f4 *)(rec + 80); }
						if (WIDE && isN) { q6 = *(const f4 *)(rec + 96); q7 = *(const f4 *)(rec + 112); }
					}
					if constexpr (WIDE) { if (isN) stepWideLoaded<true>(S, w, stk, cnt, port, q0, q1, q2, q3, q4, q5, q6, q7); }
					else
					if (isN) stepNodeLoaded<true>(S, w, stk, cnt, port, q0, q1, q2, q3);
					if (isT) stepTriLoaded(S, w, stk, cnt, port, q0, q1, q2, q3, q4, q5);
					if ((int)__popcll(__ballot(w.phase == PH_CTRL)) >= K.ctrlInRun) {
#ifdef CRH_CENSUS
						if constexpr (LEVEL >= 2) { const uint32_t n2 = (uint32_t)__popcll(__ballot(w.phase == PH_CTRL)); if (lane == 0) { CRH_WCTR(28, 1); CRH_WCTR(29, n2); } }
#endif
						if (w.phase == PH_CTRL) stepCtrl(S, w, stk, cnt, port);
					}
					/* retire + refill inside the run (round 4): once K.swapInRun lanes have ended their walk or sit idle with rays waiting — and the hit / miss stacks have room for
					 * one more batch — they are served here instead of ending the run for it; the lanes that start a walk join the next iteration */
					{
						const int nFi = (int)__popcll(__ballot(w.phase == PH_SHADE)), nEi = (int)__popcll(__ballot(w.phase == PH_IDLE));
						if (nFi + nEi >= K.swapInRun && (nFi > 0 || rqRun > 0) && hqRun < 64 && mqRun < 64) {
							if constexpr (LEVEL >= 2) { if (lane == 0) { CRH_WCTR(21, 1); CRH_WCTR(23, nFi + min(nEi + nFi, rqRun)); } }
							retireRefill(rqRun, hqRun, mqRun);
							port = TablePort<SAMP>{ptab + myPath * CRH_PATH_F4};
						}
					}
					now = __popcll(__ballot(w.phase == PH_NODE));
					/* (the lanes this iteration's node step sent to a leaf are served at the top of the next one: enough of them keep the run going, as their in-place
					 * triangle step — after which they counted as node lanes again — did before the steps were fused) */
				} while (now * 8 >= nN * K.runNum || (int)__popcll(__ballot(w.phase == PH_TRI)) >= K.triInRun);
				break;
			}
			case ST_TRI: {
				int now = nT;
				do {
					if (w.phase == PH_TRI) stepTri(S, w, stk, cnt, port);
					if constexpr (LEVEL >= 2) { if (lane == 0) { CRH_WCTR(12, 1); CRH_WCTR(24, now); } }
					now = __popcll(__ballot(w.phase == PH_TRI));
				} while (now * 8 >= nT * K.runNum);
				break;
			}
			case ST_CTRL:
				if (ph == PH_CTRL) stepCtrl(S, w, stk, cnt, port);
				if (__ballot(ph == PH_NODE_SLOW)) { if (ph == PH_NODE_SLOW) stepNodeAny<false>(S, w, stk, cnt, port); }
				break;
			case ST_SWAP: {
				int rq = raysQ, hq = hitsQ, mq = missQn;
				retireRefill(rq, hq, mq);
				break;
			}
			case ST_OPEN: {          /* the next job: the following chunk of the youngest job's unit, or a new unit from the queue */
				const int o = wq[RQ_HEAD], open = wq[RQ_OPEN];
				const int s = (o + open) % NS;
				BlockJob J;
				bool have = false;
				const float *base = myStage + (size_t)s * slabFloats;
				int defer = 0;
				if (open > 0 && !jobDeferred(wq[RQ_GEN])) {
					J = loadJob(wq[RQ_GEN]);
					if (J.passBegin + J.passCount < passEnd) {
						J.passBegin += J.passCount; J.passCount = min(chunk, passEnd - J.passBegin); have = true;
					}
				}
				if (!have) {
					uint32_t unit = 0;
					if (lane == 0) unit = atomicAdd((uint32_t *)(__attribute__((address_space(1))) uint32_t *)Q.counter, 1u);
					unit = __builtin_amdgcn_readfirstlane(unit);
					if (unit < Q.total) {
						++unitsDone;
						uint32_t lo = 0, hi = Q.ntiles;
						while (hi - lo > 1) { const uint32_t mid = (lo + hi) >> 1; if (asGlobal(Q.start)[mid] <= unit) lo = mid; else hi = mid; }
						const crh_tile t = asGlobal(Q.tiles)[lo];
						uint32_t local = unit - asGlobal(Q.start)[lo];
						const bool micro = lo >= Q.firstMicro;
						const int ubw = micro ? Q.mbw : lo >= Q.firstTiny ? Q.tbw : lo >= Q.firstSmall ? Q.sbw : Q.bw, ubh = micro ? Q.mbh : lo >= Q.firstTiny ? Q.tbh : lo >= Q.firstSmall ? Q.sbh : Q.bh;
						const uint32_t nbx = (uint32_t)(t.x1 - t.x0 + ubw - 1) / (uint32_t)ubw;
						J.passBegin = P.first_pass;
						J.passCount = min(chunk, passEnd - P.first_pass);
						if (micro && Q.segs > 1) {          /* one pass segment of a split pixel */
							const uint32_t seg = local % (uint32_t)Q.segs;
							local /= (uint32_t)Q.segs;
							J.passBegin = P.first_pass + (int)seg * Q.segPasses;
							J.passCount = min(Q.segPasses, passEnd - J.passBegin);
							defer = SJ_WH_DEFER;
							base = Q.defer + (size_t)(unit - Q.unit0) * (size_t)Q.segPasses * 3;
						}
						J.bw = ubw; J.bh = ubh;
						J.x0 = t.x0 + (int)(local % nbx) * ubw;
						J.y0 = t.y0 + (int)(local / nbx) * ubh;
						J.w = min(ubw, t.x1 - J.x0);
						J.h = min(ubh, t.y1 - J.y0);
						have = true;
					}
				}
				CRH_LOCKSTEP();          /* every lane has read the ring's words */
				if (lane == 0) {
					if (have) {
						lds_int *j = wq + RQ_SLOT0 + s * SJ_WORDS;
						j[SJ_X0] = J.x0; j[SJ_Y0] = J.y0; j[SJ_WH] = J.w | (J.h << 16) | defer
						          ; j[SJ_BWBH] = J.bw | (J.bh << 16); j[SJ_PASS0] = J.passBegin; j[SJ_PASSN] = J.passCount;
						j[SJ_NEXT] = 0; j[SJ_OUT] = 0;
						j[SJ_BASE_LO] = (int)(uint32_t)(uintptr_t)base; j[SJ_BASE_HI] = (int)(uint32_t)((uintptr_t)base >> 32);
						wq[RQ_OPEN] = open + 1; wq[RQ_GEN] = s;
					} else {
						wq[RQ_DRY] = 1;
					}
#ifdef CRH_EXP_ABS_TIMES
					if (!have && !tDry) tDry = wall_clock64();
#endif
					refreshJobWords();
				}
				__threadfence_block();
				break;
			}
			case ST_GEN: {
				const int g = wq[RQ_GEN];
				const BlockJob J = loadJob(g);
				const uint32_t genNext = (uint32_t)wq[RQ_SLOT0 + g * SJ_WORDS + SJ_NEXT], genItems = genNext + (uint32_t)itemsLeft;
				const uint32_t item = genNext + lane;
				int x = 0, y = 0, pass = 0;
				const bool valid = item < genItems && decodeItem(J, item, x, y, pass);         /* (the ballot below: every lane has read the job's words) */
				const unsigned long long vm = __ballot(valid);
				const int n = (int)__popcll(vm);
				if (valid) {
					const uint32_t rk = laneRank(vm);
					const uint32_t id = ids[CRH_IDS_FREE_END - (uint32_t)freeQ + rk];
					v3 ro, rd;
					PathRecT<RngT<SAMP>> r;
					beginPath(S, P, x, y, pass, ro, rd, r, cnt);
					putPathRay(ptab + id * CRH_PATH_F4, ro, rd, r, item | ((uint32_t)g << CRH_ROLL_SLOT_SHIFT));
					ids[CRH_IDS_RAYS + (uint32_t)raysQ + rk] = (uint8_t)id;
				}
				if (lane == 0) {
					wq[RQ_RAYS] = raysQ + n; wq[RQ_FREE] = freeQ - n;
					lds_int *j = wq + RQ_SLOT0 + g * SJ_WORDS;
					j[SJ_NEXT] = (int)(genNext + 64u); j[SJ_OUT] = j[SJ_OUT] + n;
					refreshJobWords();
				}
				__threadfence_block();
				break;
			}
			case ST_MISS: {
				const int n = min(missQn, 64);
				int mySlot = -1;
				if ((int)lane < n) {
					const uint32_t id = ids[CRH_ROLL_IDS_MISSES + (uint32_t)(missQn - n) + lane];
					const f4 *q = ptab + id * CRH_PATH_F4;
					const f4 q1 = q[1], q2 = q[2], q3 = q[3];
					v3 ro{0.0f, 0.0f, 0.0f}, rd{q1.x, q1.y, q1.z};
					PathRecT<RngT<SAMP>> r;
					r.wr = q2.x; r.wg = q2.y; r.wb = q2.z;
					r.fr = q3.x; r.fg = q3.y; r.fb = q3.z;
					r.rng.state = 0; r.depth = 0;
					const uint32_t item = asU32(q1.w);
					mySlot = (int)(item >> CRH_ROLL_SLOT_SHIFT);
					TravHit h;
					h.t = q[4].x; h.u = h.v = 0.0f; h.slot = -1; h.inst = -1;
					(void)shadeCore(S, P, ro, rd, h, r, cnt, stk);
					float *so = sampleSlot(mySlot, item & CRH_ROLL_ITEM_MASK); so[0] = r.fr; so[1] = r.fg; so[2] = r.fb;
					ids[CRH_IDS_FREE_END - 1u - (uint32_t)freeQ - lane] = (uint8_t)id;
				}
				int fin[NS];
#pragma unroll
				for (int s = 0; s < NS; ++s) fin[s] = (int)__popcll(__ballot(mySlot == s));
				if (lane == 0) {
					wq[RQ_MISSES] = missQn - n; wq[RQ_FREE] = freeQ + n;
#pragma unroll
					for (int s = 0; s < NS; ++s) if (fin[s]) wq[RQ_SLOT0 + s * SJ_WORDS + SJ_OUT] = wq[RQ_SLOT0 + s * SJ_WORDS + SJ_OUT] - fin[s];
					refreshJobWords();
				}
				__threadfence_block();
				break;
			}
			case ST_FOLD: {          /* the older job is complete: its samples go into the frame in pass order; its slot is free again */
				const int o = wq[RQ_HEAD], open = wq[RQ_OPEN];
				const BlockJob J = loadJob(o);
				__threadfence_block();                 /* the staged samples of all lanes are visible to the folding lanes */
				const float *slab = myStage + (size_t)o * slabFloats;
				if (!jobDeferred(o)) {          /* (a pass segment's samples are folded behind the kernel) */
					for (uint32_t pix = lane; pix < (uint32_t)(J.bw * J.bh); pix += 64u) foldBlockPixel(P, J, pix, slab, fb);
				}
				__threadfence_block();                 /* ... and read before a later job overwrites them */
				CRH_LOCKSTEP();          /* every lane has read the ring's words */
				if (lane == 0) { wq[RQ_HEAD] = (o + 1) % NS; wq[RQ_OPEN] = open - 1; refreshJobWords(); }
				__threadfence_block();
				break;
			}
			default: {   /* ST_SHADE */
				uint32_t clsLo = 0, clsHi = 0;
				int n = min(hitsQ, 64);
				if (sorted) {
				clsLo = (uint32_t)__builtin_amdgcn_readfirstlane(wq[RQ_CLS_LO]); clsHi = (uint32_t)__builtin_amdgcn_readfirstlane(wq[RQ_CLS_HI]);
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
#pragma unroll
				for (int b = 0; b < 8; ++b) {
					const uint32_t gone = ((fullMask >> b) & 1u) ? (uint32_t)c8[b] : (b == partCls ? (uint32_t)partN : 0u);
					if (b < 4) clsLo -= gone << (8 * b); else clsHi -= gone << (8 * (b - 4));
				}
				}
				if constexpr (LEVEL >= 2) { if (lane == 0) CRH_WCTR(18, n); }
				bool cont = false, done = false;
				int mySlot = -1;
				uint32_t id = 0;
				if ((int)lane < n) {
					id = (uint32_t)hits[(uint32_t)(hitsQ - n) + lane] & 255u;
					f4 *q = ptab + id * CRH_PATH_F4;
					const f4 q0 = q[0], q1 = q[1], q2 = q[2], q3 = q[3], q4 = q[4];
					v3 ro{q0.x, q0.y, q0.z}, rd{q1.x, q1.y, q1.z};
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
					cont = shadeCore(S, P, ro, rd, h, r, cnt, stk);
					done = !cont;
					if (cont) putPathRay(q, ro, rd, r, item);
					else {
						mySlot = (int)(item >> CRH_ROLL_SLOT_SHIFT);
						float *so = sampleSlot(mySlot, item & CRH_ROLL_ITEM_MASK); so[0] = r.fr; so[1] = r.fg; so[2] = r.fb;
					}
				}
				const unsigned long long cm = __ballot(cont), dm = __ballot(done);
				int fin[NS];
#pragma unroll
				for (int s = 0; s < NS; ++s) fin[s] = (int)__popcll(__ballot(mySlot == s));
				if (cont) ids[CRH_IDS_RAYS + (uint32_t)raysQ + laneRank(cm)] = (uint8_t)id;
				if (done) ids[CRH_IDS_FREE_END - 1u - (uint32_t)freeQ - laneRank(dm)] = (uint8_t)id;
				if (lane == 0) {
					const int nd = (int)__popcll(dm);
					wq[RQ_HITS] = hitsQ - n; wq[RQ_RAYS] = raysQ + (int)__popcll(cm); wq[RQ_FREE] = freeQ + nd;
					wq[RQ_CLS_LO] = (int)clsLo; wq[RQ_CLS_HI] = (int)clsHi;
#pragma unroll
					for (int s = 0; s < NS; ++s) if (fin[s]) wq[RQ_SLOT0 + s * SJ_WORDS + SJ_OUT] = wq[RQ_SLOT0 + s * SJ_WORDS + SJ_OUT] - fin[s];
					if (nd) refreshJobWords();
				}
				__threadfence_block();
				break;
			}
		}
		if constexpr (LEVEL >= 2) {
			if (lane == 0) {
				const uint32_t dt = CRH_TICK() - tk;
				CRH_WCTR(14, 1);
				if (pick == ST_NODE) { CRH_WCTR(9, dt); }
				else if (pick == ST_TRI) { CRH_WCTR(8, dt); }
				else if (pick == ST_CTRL) { CRH_WCTR(13, 1); CRH_WCTR(16, dt); CRH_WCTR(25, nC); }
				else if (pick == ST_SWAP) { CRH_WCTR(21, 1); CRH_WCTR(19, dt); CRH_WCTR(23, nF + min(nE + nF, raysQ)); }
				else if (pick == ST_GEN || pick == ST_MISS || pick == ST_OPEN || pick == ST_FOLD) { CRH_WCTR(22, 1); CRH_WCTR(20, dt); }
				else { CRH_WCTR(15, 1); CRH_WCTR(10, dt); }
			}
		}
	}
	if (waveStats && lane == 0) {
#ifdef CRH_EXP_ABS_TIMES
		const unsigned long long tEnd = wall_clock64();
		waveStats[2 * wave] = tStart;
		waveStats[2 * wave + 1] = (((tDry ? tDry : tEnd) - tStart) << 32) | ((tEnd - tStart) & 0xFFFFFFFFull);
		if (!snapDone) { const size_t waves = (size_t)gridDim.x * (CRH_BLOCK / 64u); waveStats[2 * (waves + wave)] = 0; waveStats[2 * (waves + wave) + 1] = 0; }
#else
		waveStats[2 * wave] = wall_clock64() - tStart;
		waveStats[2 * wave + 1] = unitsDone;
#endif
	}
	const bool lead = (lane == 0);
	if constexpr (LEVEL >= 2) { if (dumpRays && lead) waveStats[CRH_DUMP_HDR + 2 + wave] = min(dumpCount, dumpCap); }
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
#undef CRH_WCTR

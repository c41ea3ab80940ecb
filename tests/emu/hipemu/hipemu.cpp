/*
 * hipemu.cpp — runtime of the HIP-on-CPU shim (see hip/hip_runtime.h): fibers, the wave / block scheduler, and the
 * bookkeeping versions of the runtime API. TEST INFRASTRUCTURE ONLY.
 */
#include "hip/hip_runtime.h"

#include <sys/mman.h>

#include <atomic>
#include <cstdio>
#include <thread>
#include <vector>

#if !defined(__x86_64__)
#error "hipemu's context switch is written for x86-64"
#endif

/* void hipemu_switch(void **save_sp, void *new_sp): park the caller (callee-saved registers on its stack, stack pointer in *save_sp)
 * and continue whoever parked at new_sp. */
extern "C" void hipemu_switch(void **save_sp, void *new_sp);
asm(R"(
	.text
	.globl hipemu_switch
	.type hipemu_switch, @function
hipemu_switch:
	pushq %rbp
	pushq %rbx
	pushq %r12
	pushq %r13
	pushq %r14
	pushq %r15
	movq %rsp, (%rdi)
	movq %rsi, %rsp
	popq %r15
	popq %r14
	popq %r13
	popq %r12
	popq %rbx
	popq %rbp
	ret
	.size hipemu_switch, .-hipemu_switch
)");

namespace hipemu {

thread_local Lane *t_lane = nullptr;

namespace {

constexpr size_t kStackBytes = 256u << 10;      /* per lane; mapped lazily */
constexpr unsigned kWave = 64;

std::atomic<uint64_t> g_launches{0}, g_blocks{0}, g_collectives{0}, g_switches{0};

/* one OS thread's worth of fibers: the lanes of the block it is running */
struct Worker {
	std::vector<Lane> lanes;
	std::vector<char *> stacks;
	void *schedSp = nullptr;                     /* the scheduler's own context while a lane runs */
	const std::function<void()> *body = nullptr;
	uint64_t collectives = 0, switches = 0;
	~Worker() { for (char *s : stacks) munmap(s, kStackBytes); }
};
thread_local Worker *t_worker = nullptr;

[[noreturn]] void die(const char *what) {
	fprintf(stderr, "hipemu: %s\n", what);
	abort();
}

/* first frame of every lane */
void laneMain() {
	Worker *w = t_worker;
	(*w->body)();
	Lane *me = t_lane;
	me->state = W_DONE;
	hipemu_switch(&me->sp, w->schedSp);
	die("a finished lane was resumed");
}

void prepare(Worker &w, unsigned i) {
	if (i >= w.stacks.size()) {
		void *m = mmap(nullptr, kStackBytes, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS | MAP_NORESERVE, -1, 0);
		if (m == MAP_FAILED) die("cannot map a lane stack");
		(void)mprotect(m, 4096, PROT_NONE);          /* guard page: a lane that outgrows its stack faults instead of writing into its neighbour's */
		w.stacks.push_back((char *)m);
	}
	uintptr_t top = ((uintptr_t)w.stacks[i] + kStackBytes) & ~(uintptr_t)15;
	void **sp = (void **)top;
	*--sp = nullptr;                 /* where laneMain would return to (it never does): keeps rsp = 8 mod 16 at its entry */
	*--sp = (void *)&laneMain;       /* popped by hipemu_switch's ret */
	for (int r = 0; r < 6; ++r) *--sp = nullptr;
	w.lanes[i].sp = (void *)sp;
	w.lanes[i].state = W_RUN;
	w.lanes[i].site = nullptr;
}

inline void resume(Worker &w, Lane &l) {
	t_lane = &l;
	++w.switches;
	hipemu_switch(&w.schedSp, l.sp);
}

/* All live lanes of the wave wait at a wave collective: resolve it. */
void resolve(Lane *lanes, unsigned n, int kind) {
	const void *site = nullptr;
	bool have = false;
	for (unsigned i = 0; i < n; ++i) {
		if (lanes[i].state == W_DONE) continue;
		if (!have) { site = lanes[i].site; have = true; }
		else if (lanes[i].site != site) die("lanes of one wave wait at different collectives (divergent collective: not modelled)");
	}
	if (kind == W_BALLOT) {
		uint64_t m = 0;
		for (unsigned i = 0; i < n; ++i) if (lanes[i].state != W_DONE && lanes[i].arg) m |= 1ull << i;
		for (unsigned i = 0; i < n; ++i) lanes[i].res = m;
	} else if (kind == W_FIRSTLANE) {
		uint64_t v = 0;
		for (unsigned i = 0; i < n; ++i) if (lanes[i].state != W_DONE) { v = lanes[i].arg; break; }
		for (unsigned i = 0; i < n; ++i) lanes[i].res = v;
	} else if (kind == W_LOCKSTEP) {
		/* a rendezvous only */
	} else if (kind == W_READ_LANE) {
		for (unsigned i = 0; i < n; ++i) {
			if (lanes[i].state == W_DONE) continue;
			const unsigned j = (unsigned)lanes[i].arg2;
			lanes[i].res = (j < n && lanes[j].state != W_DONE) ? lanes[j].arg : lanes[i].arg;
		}
	} else if (kind == W_QUAD_PERM) {
		for (unsigned i = 0; i < n; ++i) {
			if (lanes[i].state == W_DONE) continue;
			const unsigned j = (i & ~3u) | (unsigned)((lanes[i].arg2 >> (2u * (i & 3u))) & 3u);
			lanes[i].res = (j < n && lanes[j].state != W_DONE) ? lanes[j].arg : 0u;
		}
	} else {           /* W_SHFL_XOR */
		for (unsigned i = 0; i < n; ++i) {
			if (lanes[i].state == W_DONE) continue;
			const unsigned j = i ^ (unsigned)lanes[i].arg2;
			lanes[i].res = (j < n && lanes[j].state != W_DONE) ? lanes[j].arg : lanes[i].arg;
		}
	}
	for (unsigned i = 0; i < n; ++i) if (lanes[i].state != W_DONE) lanes[i].state = W_RUN;
}

/* One turn of a wave: run every runnable lane to its next wait; resolve the wave collective if all live lanes reached it.
 * Returns false when every lane is done or waits at the block barrier (nothing more to do until the barrier opens). */
bool turn(Worker &w, Lane *lanes, unsigned n) {
	bool ran = false;
	/* Lowest lane first. On the hardware a wave runs in lockstep; here a lane runs alone from one collective to the next. Where the
	 * kernels let lanes exchange data through LDS without a collective in between they say so (CRH_LOCKSTEP, a rendezvous here); what
	 * remains are stretches where a lane writes a byte that a LOWER lane reads first in program order (k_pathtrace, ST_GEN: the ray
	 * stack growing into the free stack) — in this order the reader has already run. Anything else fails the frame comparison. */
	for (unsigned i = 0; i < n; ++i) {
		Lane &l = lanes[i];
		if (l.state == W_SLEEP) l.state = W_RUN;            /* s_sleep: the lane gave the other waves a turn */
		if (l.state != W_RUN) continue;
		resume(w, l);
		ran = true;
	}
	int kind = -1;
	unsigned live = 0, atKind = 0;
	for (unsigned i = 0; i < n; ++i) {
		const int s = lanes[i].state;
		if (s == W_DONE) continue;
		++live;
		if (s == W_BALLOT || s == W_SHFL_XOR || s == W_FIRSTLANE || s == W_LOCKSTEP || s == W_QUAD_PERM || s == W_READ_LANE) {
			if (kind < 0) kind = s;
			if (s == kind) ++atKind;
		}
	}
	if (live && atKind == live) { resolve(lanes, n, kind); ++w.collectives; return true; }
	if (kind >= 0 && atKind != live) {
		/* part of the wave waits at a collective, the rest sleeps or sits at the barrier: legal only while sleepers catch up */
		for (unsigned i = 0; i < n; ++i) if (lanes[i].state == W_SLEEP) return true;
		for (unsigned i = 0; i < n; ++i) if (lanes[i].state == W_BARRIER) die("part of a wave is at __syncthreads while the rest waits at a wave collective");
		for (unsigned i = 0; i < n; ++i) fprintf(stderr, "lane %u state %d site %p arg %llu\n", i, lanes[i].state, lanes[i].site, (unsigned long long)lanes[i].arg);
		die("lanes of one wave wait at different kinds of collectives");
	}
	for (unsigned i = 0; i < n; ++i) if (lanes[i].state == W_SLEEP) return true;
	return ran;
}

void runBlock(Worker &w, dim3 grid, dim3 block, unsigned bx, unsigned by) {
	const unsigned nThreads = block.x * block.y * block.z;
	if (nThreads == 0 || nThreads > 1024) die("block size must be 1..1024");
	w.lanes.assign(nThreads, Lane{});
	for (unsigned i = 0; i < nThreads; ++i) {
		prepare(w, i);
		Lane &l = w.lanes[i];
		l.tIdx = {i % block.x, (i / block.x) % block.y, i / (block.x * block.y)};
		l.bIdx = {bx, by, 0};
		l.bDim = {block.x, block.y, block.z};
		l.gDim = {grid.x, grid.y, grid.z};
	}
	const unsigned nWaves = (nThreads + kWave - 1) / kWave;
	for (;;) {
		bool progress = false;
		for (unsigned v = 0; v < nWaves; ++v) {
			const unsigned first = v * kWave, n = std::min(kWave, nThreads - first);
			if (turn(w, &w.lanes[first], n)) progress = true;
		}
		if (progress) continue;
		/* nothing moved: everyone is done or at the barrier */
		unsigned live = 0, atBarrier = 0;
		for (Lane &l : w.lanes) { if (l.state == W_DONE) continue; ++live; if (l.state == W_BARRIER) ++atBarrier; }
		if (live == 0) break;
		if (atBarrier != live) die("deadlock: lanes wait for a collective that cannot complete");
		for (Lane &l : w.lanes) if (l.state == W_BARRIER) l.state = W_RUN;
	}
	t_lane = nullptr;
}

}  // namespace

uint64_t collective(int kind, uint64_t arg, uint64_t arg2, const void *site) {
	Lane *me = t_lane;
	if (!me) die("wave intrinsic outside a kernel");
	me->arg = arg; me->arg2 = arg2; me->site = site; me->state = kind;
	hipemu_switch(&me->sp, t_worker->schedSp);
	return me->res;
}

void launch(dim3 grid, dim3 block, const std::function<void()> &body) {
	if (t_lane) die("nested kernel launch");
	const uint64_t nBlocks = (uint64_t)grid.x * grid.y * grid.z;
	g_launches.fetch_add(1);
	g_blocks.fetch_add(nBlocks);
	if (nBlocks == 0) return;
	if (grid.z != 1) die("grid.z > 1 is not supported");
	unsigned nThreads = 1;
	if (const char *e = getenv("HIPEMU_THREADS")) nThreads = (unsigned)std::max(1, atoi(e));
	else nThreads = std::min(8u, std::max(1u, std::thread::hardware_concurrency()));
	nThreads = (unsigned)std::min<uint64_t>(nThreads, nBlocks);
	std::atomic<uint64_t> next{0};
	auto work = [&]() {
		Worker w;
		w.body = &body;
		t_worker = &w;
		for (;;) {
			const uint64_t b = next.fetch_add(1);
			if (b >= nBlocks) break;
			runBlock(w, grid, block, (unsigned)(b % grid.x), (unsigned)(b / grid.x));
		}
		g_collectives.fetch_add(w.collectives);
		g_switches.fetch_add(w.switches);
		t_worker = nullptr;
	};
	if (nThreads <= 1) { work(); return; }
	std::vector<std::thread> pool;
	for (unsigned i = 0; i < nThreads; ++i) pool.emplace_back(work);
	for (auto &t : pool) t.join();
}

Stats stats() { return Stats{g_launches.load(), g_blocks.load(), g_collectives.load(), g_switches.load()}; }

}  // namespace hipemu

/* ---- runtime API ------------------------------------------------------------------------------------------------------------------- */
struct hipemu_stream { int unused; };
struct hipemu_event { std::chrono::steady_clock::time_point t; bool recorded; };

const char *hipGetErrorString(hipError_t e) {
	switch (e) {
		case hipSuccess: return "no error";
		case hipErrorInvalidValue: return "invalid argument";
		case hipErrorOutOfMemory: return "out of memory";
		case hipErrorLaunchFailure: return "unspecified launch failure";
		case hipErrorInvalidDeviceFunction: return "invalid device function";
		case hipErrorNotReady: return "not ready";
		case hipErrorInvalidDevice: return "invalid device ordinal";
	}
	return "unknown error";
}
/* HIPEMU_DEVICES=<n>: that many identical "devices" (all of them this CPU and this heap): lets a host that drives several GPUs from several
 * threads — renderer_hip.c — run its partition and gather logic */
static int deviceCount() { const char *e = getenv("HIPEMU_DEVICES"); const int n = e ? atoi(e) : 1; return n < 1 ? 1 : (n > 16 ? 16 : n); }
hipError_t hipGetDeviceCount(int *n) { if (!n) return hipErrorInvalidValue; *n = deviceCount(); return hipSuccess; }
/* HIPEMU_FAIL_DEVICE=<d>: every hipMalloc of a thread whose current device is d fails (a GPU that cannot be set up: renderer_hip.c's failed-worker path) */
static thread_local int t_device = 0;
static int failDevice() { const char *e = getenv("HIPEMU_FAIL_DEVICE"); return e ? atoi(e) : -1; }
/* HIPEMU_FAIL_LAUNCH=<d>:<n>: the n-th hipGetLastError() (n >= 1: the n-th kernel launch a host checks) of a thread whose current device is d reports a launch
 * failure, and so does every later one — a GPU that dies in the middle of a frame (renderer_hip.c re-deals its strips to the GPUs that are left) */
static thread_local int t_launchChecks = 0;
hipError_t hipGetLastError(void) {
	static const char *e = getenv("HIPEMU_FAIL_LAUNCH");
	if (!e) return hipSuccess;
	int d = -1, n = 0;
	if (sscanf(e, "%d:%d", &d, &n) != 2 || t_device != d) return hipSuccess;
	return ++t_launchChecks >= n ? hipErrorLaunchFailure : hipSuccess;
}
hipError_t hipSetDevice(int device) { if (device < 0 || device >= deviceCount()) return hipErrorInvalidDevice; t_device = device; return hipSuccess; }
hipError_t hipGetDeviceProperties(hipDeviceProp_t *prop, int device) {
	if (!prop || device < 0 || device >= deviceCount()) return hipErrorInvalidValue;
	memset(prop, 0, sizeof(*prop));
	snprintf(prop->name, sizeof(prop->name), "hipemu (CPU emulation, tests only)");
	const char *e = getenv("HIPEMU_CUS");
	prop->multiProcessorCount = e && atoi(e) > 0 ? atoi(e) : 2;
	prop->totalGlobalMem = (size_t)8 << 30;
	return hipSuccess;
}
hipError_t hipMalloc(void **p, size_t bytes) {
	if (!p) return hipErrorInvalidValue;
	*p = nullptr;
	if (t_device == failDevice()) return hipErrorOutOfMemory;
	if (posix_memalign(p, 256, bytes ? bytes : 1) != 0) return hipErrorOutOfMemory;
	return hipSuccess;
}
hipError_t hipFree(void *p) { free(p); return hipSuccess; }
hipError_t hipHostMalloc(void **p, size_t bytes, unsigned) { return hipMalloc(p, bytes); }
hipError_t hipHostFree(void *p) { free(p); return hipSuccess; }
hipError_t hipHostGetDevicePointer(void **dev, void *host, unsigned) { if (!dev) return hipErrorInvalidValue; *dev = host; return hipSuccess; }
hipError_t hipMemcpy(void *dst, const void *src, size_t bytes, hipMemcpyKind) { if (bytes) memmove(dst, src, bytes); return hipSuccess; }
hipError_t hipMemcpy2DAsync(void *dst, size_t dpitch, const void *src, size_t spitch, size_t width, size_t height, hipMemcpyKind, hipStream_t) {
	for (size_t r = 0; r < height; ++r) memcpy((char *)dst + r * dpitch, (const char *)src + r * spitch, width);
	return hipSuccess;
}
hipError_t hipMemcpyAsync(void *dst, const void *src, size_t bytes, hipMemcpyKind k, hipStream_t) { return hipMemcpy(dst, src, bytes, k); }
hipError_t hipMemset(void *dst, int value, size_t bytes) { if (bytes) memset(dst, value, bytes); return hipSuccess; }
hipError_t hipMemsetAsync(void *dst, int value, size_t bytes, hipStream_t) { return hipMemset(dst, value, bytes); }
hipError_t hipStreamCreateWithFlags(hipStream_t *s, unsigned) { if (!s) return hipErrorInvalidValue; *s = new hipemu_stream{0}; return hipSuccess; }
hipError_t hipStreamDestroy(hipStream_t s) { delete s; return hipSuccess; }
hipError_t hipStreamSynchronize(hipStream_t) { return hipSuccess; }
hipError_t hipDeviceSynchronize(void) { return hipSuccess; }
hipError_t hipEventCreate(hipEvent_t *e) { if (!e) return hipErrorInvalidValue; *e = new hipemu_event{{}, false}; return hipSuccess; }
hipError_t hipEventCreateWithFlags(hipEvent_t *e, unsigned) { return hipEventCreate(e); }
hipError_t hipEventDestroy(hipEvent_t e) { delete e; return hipSuccess; }
hipError_t hipEventRecord(hipEvent_t e, hipStream_t) { if (!e) return hipErrorInvalidValue; e->t = std::chrono::steady_clock::now(); e->recorded = true; return hipSuccess; }
hipError_t hipEventSynchronize(hipEvent_t) { return hipSuccess; }
hipError_t hipEventQuery(hipEvent_t) { return hipSuccess; }
hipError_t hipEventElapsedTime(float *ms, hipEvent_t a, hipEvent_t b) {
	if (!ms || !a || !b || !a->recorded || !b->recorded) return hipErrorInvalidValue;
	*ms = std::chrono::duration<float, std::milli>(b->t - a->t).count();
	return hipSuccess;
}

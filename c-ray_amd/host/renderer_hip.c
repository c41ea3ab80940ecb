/*
 * renderer_hip.c — drop-in replacement for the reference's src/renderer/renderer.c.
 *
 * It exports exactly the three symbols the rest of c-ray links against (src/renderer/renderer.h:101-107):
 *     struct renderer *newRenderer(void);
 *     struct texture  *renderFrame(struct renderer *r);     <- THE boundary (called from src/c-ray.c:280)
 *     void             destroyRenderer(struct renderer *r);
 * and leaves struct renderer / state / prefs / renderThreadState untouched, so src/c-ray.c, the JSON loader,
 * tile.c, ui.c (SDL2 preview) and the PNG/BMP encoders keep working unmodified.
 *
 * What changes is who consumes the tile queue: instead of N pthread workers running the pixel x pass loop
 * (renderer.c:258-327) there is ONE host thread per GPU. The scene is flattened once (flatten.c) by the main thread WHILE the GPU threads
 * create their contexts, allocate the per-wave buffers and load the kernels' code objects (crh_context_prepare) and a helper thread
 * creates the RCCL communicators (crh_frames_prepare): none of that is left inside the frame's timed part. Each GPU thread then uploads the
 * scene and renders ITS share of the frame — one GPU: the whole frame as one region; several: every G-th 4-row strip — in dispatches of
 * as many passes as take about a second (the persistent kernel balances the work units inside a dispatch itself; a frame is cut only along the
 * pass axis, so that the preview window, the abort and the pause key get a turn). Between two dispatches the thread converts its share to
 * 8-bit sRGB ON THE DEVICE (crh_framebuffer_to_srgb8: colorToSRGB + setPixel's truncation, bit for bit) into `output`, which the main
 * thread keeps drawing (renderer.c:294-300 wrote it per sample). At the end the GPUs' strips are gathered onto GPU 0 with RCCL over xGMI
 * (crh_frames_gather: 1/G of the frame per link; CRH_FRAMES=reduce selects the one-ncclReduce form), the 8-bit frame and the float buffer
 * come back with one download each.
 *
 * --iterative (renderThreadInteractive, renderer.c:184-250): passes 1 .. sampleCount-1 with the Halton sampler, the frame
 * shown while it converges. Here the main thread drives every GPU (every G-th 4-row strip belongs to one GPU for the whole frame, so each
 * pixel's running mean stays on one GPU) in dispatches of as many passes as take about 16 ms, converts to 8-bit on the device after each
 * and redraws; the float buffer is gathered once, at the end. The result is the reference's single-thread result: with several threads the
 * reference itself races on state.finishedPasses and is not reproducible.
 *
 * A GPU that fails — at set-up or in the middle of the frame — does not end the program: its strips (a pure function of (g, G): share.h) are dealt to the
 * GPUs that finished and rendered there from pass 0, the way the reference's cluster master re-issues the tiles of a worker it lost (src/datatypes/tile.c:31-42);
 * that frame is assembled on the host (the survivors' buffers merged: a pixel has one owner, everybody else holds 0.0f there). Only when NO GPU is left
 * does renderFrame() end with logr(error, ...).
 *
 * Environment: CRAY_HIP_DEVICES=<n> caps the number of GPUs used; CRH_DROPIN_PASSES=<n> (dev / tests) fixes the passes per dispatch; CRH_DUMP_F32=<path> dumps the float buffer; CRH_DUMP_STATS=<path>
 * writes where the frame's time went; CRH_FRAMES=reduce: see above.
 * No GPU => logr(error, ...) (which exits, src/utils/logging.c:69-73): there is no CPU fallback in this file.
 */
#include <stdlib.h>
#include <stdio.h>
#include <string.h>
#include <stdbool.h>
#include <pthread.h>
#include <errno.h>
#include <time.h>

#include "includes.h"
#include "datatypes/image/imagefile.h"
#include "renderer/renderer.h"
#include "datatypes/image/texture.h"
#include "datatypes/color.h"
#include "datatypes/scene.h"
#include "datatypes/tile.h"
#include "datatypes/vertexbuffer.h"
#include "utils/ui.h"
#include "utils/logging.h"
#include "utils/timer.h"
#include "utils/platform/thread.h"
#include "utils/platform/mutex.h"
#include "utils/args.h"

#include "cray_hip.h"
#include "flatten.h"
#include "share.h"

#define MAX_GPUS 16

/* what the main thread and the GPU threads tell each other: "the scene is flattened" (main -> GPUs), "a GPU thread has finished" (GPUs -> main) */
struct frameSync {
	pthread_mutex_t mu;
	pthread_cond_t cv;
	int sceneReady;          /* 1: flattened (and, for several GPUs, compiled), -1: the flattener / the layout compile failed */
	const crh_compiled_scene *compiled;          /* several GPUs: the device layout, derived ONCE by the main thread (round 5: every GPU thread used to derive it for itself); NULL: one GPU,
	                                              * whose crh_scene_upload compiles with its texel copy beside it */
	int finished;
	int launched;            /* GPU threads whose first dispatch is on its device (or that ended before one): the flattened scene has been read for the last time */
};

struct gpuWorker {
	struct renderer *r;
	struct renderThreadState *state;
	const crh_scene_desc *scene;
	struct frameSync *sync;
	struct texture *output;
	struct timeval frameStart;   /* renderFrame() entry */
	crh_ctx *ctx;
	float *fb;
	int device;
	int failed;
	char error[256];      /* crh_last_error() is per thread: the failing dispatch thread keeps its message here */
	uint64_t rays;
	int announced;                        /* sync->launched counts this thread */
	long contextUs, uploadUs, renderUs;   /* context + buffers + code objects (beside the flattener); scene upload + framebuffer; dispatch loop (first launch to last sync) */
	long readyUs;             /* renderFrame() entry -> this GPU ready to dispatch */
	double kernelMs;          /* GPU time of the dispatches (HIP events around the kernels) */
	long launchUs;            /* host time inside crh_render_tiles (work list, copies, launch) */
	int dispatches;
};

/* A dispatch keeps the whole GPU busy from its first work unit to its last (no drain in between), so a frame is cut only along the pass
 * axis, and only when it is long: the first dispatch takes as many passes as make FIRST_DISPATCH_PATHS paths (the bench frame: all 256),
 * the following ones as many as took about DISPATCH_TARGET_MS at the measured rate — the preview, the abort and the pause key wait no
 * longer than that. */
#define FIRST_DISPATCH_PATHS (256ull << 20)
#define DISPATCH_TARGET_MS 1000.0

/* The share of GPU g of G: G == 1 -> the whole frame; G > 1 -> every G-th 4-row strip
 * (static shares must be balanced, and dealing out the tile list is not: DESIGN.md section 6). Returns the tile count. */
static uint32_t gpuShare(const struct renderer *r, int g, int G, crh_tile **out) {
	const int W = (int)r->prefs.imageWidth, H = (int)r->prefs.imageHeight;
	if (G == 1) {
		/* the whole frame as one region: inside a dispatch the library hands out the pixel blocks bottom-up, row by row. The order of the
		 * reference's tile list (tile.c:119-241) is a preview preference, not part of the result — and measured on one GPU it costs 2 %
		 * (fromMiddle) to 10 % (topToBottom: the expensive bottom of a frame ends up in the small blocks of the dispatch's tail) */
		crh_tile *t = calloc(1, sizeof(*t));
		t[0] = (crh_tile){0, 0, W, H};
		*out = t;
		return 1;
	}
	crh_tile *t = calloc((size_t)crh_strip_share_max(H, G), sizeof(*t));
	*out = t;
	return crh_strip_share(W, H, g, G, t);
}

/* renderer.c:294-300 for this GPU's share: colorToSRGB + setPixel's truncation on the device, then the rows of ITS strips into `output` (texture.c:18-22
 * layout, the float buffer's) — one strided copy of 1 / G of the 8-bit frame (crh_framebuffer_strips_to_srgb8). */
static int refreshOutput(crh_ctx *ctx, const float *fb, int W, int H, struct texture *output, int g, int G) {
	return crh_framebuffer_strips_to_srgb8(ctx, fb, W, H, CRH_STRIP_ROWS, g, G, output->data.byte_p);
}

/* A throw-away dispatch ends a GPU's set-up: a few pixels, one pass, into the (still empty) framebuffer, which is cleared again. Measured in round 3
 * (CRH_TRACE_SYNC, profiles/r03l_probe_dropin.log): the first kernel of a process that does real work starts 5-24 ms after its launch — whatever was copied,
 * synchronized, launched empty or waited for before it — and the second does not; that first one should not be the frame. CRH_DROPIN_NO_WARM=1 leaves it out. */
static void warmUp(crh_ctx *ctx, float *fb, const crh_render_params *p, int device) {
	if (getenv("CRH_DROPIN_NO_WARM")) return;
	crh_render_params wp = *p;
	wp.x0 = 0; wp.y0 = 0; wp.x1 = p->image_width < 64 ? p->image_width : 64; wp.y1 = p->image_height < 16 ? p->image_height : 16;
	wp.first_pass = 0; wp.pass_count = 1;
	if (crh_render_region(ctx, &wp, fb) != CRH_OK || crh_synchronize(ctx) != CRH_OK || crh_framebuffer_clear(ctx, fb, p->image_width, p->image_height) != CRH_OK ||
		crh_counters_reset(ctx) != CRH_OK || crh_synchronize(ctx) != CRH_OK)
		logr(warning, "GPU %d: warm-up dispatch: %s\n", device, crh_last_error());
}

/* ---- the process's render contexts ------------------------------------------------------------------------------------------------------------
 * A context ready to dispatch — stream, counters, code objects, 600 MB of per-wave buffers — takes 13-30 ms to make (round 4, CRH_TRACE_UPLOAD: 8 ms until it
 * exists, 5-21 ms until crh_context_prepare returns) and 5 ms to take down, and none of it depends on the scene. They are therefore made — one thread per
 * device — from newRenderer() on, while the program spends hundreds of milliseconds parsing JSON and OBJ files, and a frame's contexts go back to this pool instead
 * of being destroyed inside renderFrame(); destroyRenderer() ends them. A dispatch thread whose device has no context yet (the maker failed, more GPUs than
 * MAX_GPUS makers) makes it beside the flattener, as before. CRH_DROPIN_NO_PREFETCH=1: every context is made inside renderFrame();
 * CRH_DROPIN_PREFETCH_GPUS=n: makers for the first n devices only (a scene file that asks for fewer GPUs than the node has). */
static struct {
	pthread_mutex_t mu, joinMu;          /* joinMu: one thread joins the master, the others wait until it has (a maker that is not started yet would be missed) */
	crh_ctx *ctx[MAX_GPUS];
	pthread_t master, maker[MAX_GPUS];
	int masterLive, makerLive[MAX_GPUS];
} g_pool = {.mu = PTHREAD_MUTEX_INITIALIZER, .joinMu = PTHREAD_MUTEX_INITIALIZER};

/* CRAY_HIP_WALK=wide4: the frame renderer walks the 4-ary copy of the BVHs (CRH_OPT_WALK; venus.json +10 %, statues / soups +3 %, frames not bit-identical to the
 * reference's where two hits nearly tie: an opt-in). Anything else: the reference's walk */
static int frameWalk(void) {
	const char *e = getenv("CRAY_HIP_WALK");
	return e && !strcmp(e, "wide4") ? CRH_WALK_WIDE4 : CRH_WALK_BINARY;
}

static crh_ctx *makeContext(int device) {
	crh_ctx *c = NULL;
	/* counter level 1: this host reports rays only (the detailed counters cost ~20 % of the kernel's time) */
	if (crh_context_create(device, NULL, &c) != CRH_OK) return NULL;
	if (crh_set_option(c, CRH_OPT_COUNTER_LEVEL, 1) != CRH_OK || crh_set_option(c, CRH_OPT_WALK, frameWalk()) != CRH_OK ||
		(!getenv("CRH_DROPIN_NO_PREPARE") && crh_context_prepare(c) != CRH_OK)) {
		crh_context_destroy(c);
		return NULL;
	}
	return c;
}

static void *makerThread(void *arg) {
	const int device = (int)(intptr_t)arg;
	crh_ctx *c = makeContext(device);
	pthread_mutex_lock(&g_pool.mu);
	g_pool.ctx[device] = c;
	pthread_mutex_unlock(&g_pool.mu);
	return NULL;
}

/* counts the devices (the first HIP call of the process: not on the thread that is about to parse the scene) and starts one maker per device the frame will use:
 * renderFrame() renders on every visible GPU, CRAY_HIP_DEVICES=n on the first n */
static void *masterThread(void *arg) {
	(void)arg;
	int n = crh_device_count();
	if (n > MAX_GPUS) n = MAX_GPUS;
	const char *cap = getenv("CRAY_HIP_DEVICES");          /* the same cap renderFrame() applies: no context on a GPU the frame will not use */
	if (cap && atoi(cap) > 0 && atoi(cap) < n) n = atoi(cap);
	if (getenv("CRH_DROPIN_PREFETCH_GPUS") && atoi(getenv("CRH_DROPIN_PREFETCH_GPUS")) < n) n = atoi(getenv("CRH_DROPIN_PREFETCH_GPUS"));
	for (int g = 0; g < n; ++g) {
		pthread_mutex_lock(&g_pool.mu);
		if (!g_pool.makerLive[g] && !g_pool.ctx[g] && pthread_create(&g_pool.maker[g], NULL, makerThread, (void *)(intptr_t)g) == 0) g_pool.makerLive[g] = 1;
		pthread_mutex_unlock(&g_pool.mu);
	}
	return NULL;
}

static void poolJoin(int device);
/* (a stated limit: this handler covers the context MAKERS. A logr(error) -> exit() from inside renderFrame / renderInteractive while the frame's GPU threads have
 * dispatches in flight is not waited for here — those threads block in crh_synchronize for the length of a frame, and exit() would hang as long; the reference's own worker
 * threads are not joined on that path either: renderer.c:96-117 —, and a maker that hangs inside hipMalloc holds exit() up with it.) */
/* the process never exits with HIP calls in flight (logr(error) exit()s on a scene the parser rejects, while the makers may be inside hipMalloc or a code-object
 * load — ROCm's teardown is known to crash then): exit() waits for the makers here; the contexts themselves are left to the process's end */
static void poolAtExit(void) {
	for (int g = 0; g < MAX_GPUS; ++g) poolJoin(g);
}

static void poolPrefetch(void) {
	if (getenv("CRH_DROPIN_NO_PREFETCH")) return;
	static int hooked;
	if (!hooked) { hooked = 1; atexit(poolAtExit); }
	pthread_mutex_lock(&g_pool.mu);
	if (!g_pool.masterLive && pthread_create(&g_pool.master, NULL, masterThread, NULL) == 0) g_pool.masterLive = 1;
	pthread_mutex_unlock(&g_pool.mu);
}

static void poolJoin(int device) {
	pthread_mutex_lock(&g_pool.joinMu);
	pthread_mutex_lock(&g_pool.mu);
	const int master = g_pool.masterLive;
	g_pool.masterLive = 0;
	pthread_mutex_unlock(&g_pool.mu);
	if (master) pthread_join(g_pool.master, NULL);          /* (it only counts and starts) */
	pthread_mutex_unlock(&g_pool.joinMu);
	pthread_mutex_lock(&g_pool.mu);
	const int live = g_pool.makerLive[device];
	g_pool.makerLive[device] = 0;
	pthread_mutex_unlock(&g_pool.mu);
	if (live) pthread_join(g_pool.maker[device], NULL);
}

/* a context for `device`, ready to dispatch: the pool's, or a new one (NULL: crh_last_error() of the calling thread says why) */
static crh_ctx *poolAcquire(int device) {
	poolJoin(device);
	pthread_mutex_lock(&g_pool.mu);
	crh_ctx *c = g_pool.ctx[device];
	g_pool.ctx[device] = NULL;
	pthread_mutex_unlock(&g_pool.mu);
	return c ? c : makeContext(device);
}

static void poolRelease(int device, crh_ctx *c, int healthy) {
	if (!c) return;
	pthread_mutex_lock(&g_pool.mu);
	const int keep = healthy && !g_pool.ctx[device];
	if (keep) g_pool.ctx[device] = c;
	pthread_mutex_unlock(&g_pool.mu);
	if (!keep) crh_context_destroy(c);
}

static void poolDestroy(void) {
	for (int g = 0; g < MAX_GPUS; ++g) {
		poolJoin(g);
		pthread_mutex_lock(&g_pool.mu);
		crh_ctx *c = g_pool.ctx[g];
		g_pool.ctx[g] = NULL;
		pthread_mutex_unlock(&g_pool.mu);
		if (c) crh_context_destroy(c);
	}
}

static void announceLaunch(struct gpuWorker *w) {
	if (w->announced) return;
	w->announced = 1;
	pthread_mutex_lock(&w->sync->mu);
	w->sync->launched++;
	pthread_cond_broadcast(&w->sync->cv);
	pthread_mutex_unlock(&w->sync->mu);
}

static void gpuThreadDone(struct gpuWorker *w) {
	announceLaunch(w);
	w->state->currentTileNum = -1;
	w->state->threadComplete = true;
	pthread_mutex_lock(&w->sync->mu);
	w->sync->finished++;
	pthread_cond_broadcast(&w->sync->cv);
	pthread_mutex_unlock(&w->sync->mu);
}

static void *gpuThread(void *arg) {
	struct gpuWorker *w = threadUserData(arg);
	struct renderer *r = w->r;
	const int W = (int)r->prefs.imageWidth, H = (int)r->prefs.imageHeight;
	const int G = r->prefs.threadCount;
	crh_render_params p;
	memset(&p, 0, sizeof(p));
	p.image_width = W; p.image_height = H;
	p.max_passes = r->prefs.sampleCount;
	p.bounces = r->prefs.bounces;
	struct timeval phase;
	startTimer(&phase);

	/* beside the flattener: the context, the per-wave buffers, the code objects.
	 * counter level 1: this host reports rays only (the detailed counters cost ~20 % of the kernel's time) */
	w->ctx = poolAcquire(w->device);
	const long preparedUs = getUs(phase);
	/* (a pooled context has rendered before in this process: the frame's ray count starts at zero — ADVICE r05) */
	int ok = w->ctx != NULL && crh_counters_reset(w->ctx) == CRH_OK && crh_framebuffer_alloc(w->ctx, W, H, &w->fb) == CRH_OK;
	if (getenv("CRH_TRACE_UPLOAD"))
		fprintf(stderr, "gpuThread trace: GPU %d has its context (pool, or created + code objects + per-wave buffers) after %.1f ms, its framebuffer after %.1f ms\n",
				w->device, preparedUs / 1e3, getUs(phase) / 1e3);
	if (!ok) snprintf(w->error, sizeof(w->error), "%s", crh_last_error());
	w->contextUs = getUs(phase);
	pthread_mutex_lock(&w->sync->mu);
	while (!w->sync->sceneReady) pthread_cond_wait(&w->sync->cv, &w->sync->mu);
	const int sceneOk = w->sync->sceneReady > 0;
	pthread_mutex_unlock(&w->sync->mu);
	startTimer(&phase);
	if (ok && sceneOk && ((w->sync->compiled ? crh_scene_upload_compiled(w->ctx, w->sync->compiled) : crh_scene_upload(w->ctx, w->scene)) != CRH_OK ||
		crh_synchronize(w->ctx) != CRH_OK)) {           /* the scene copies are asynchronous: they belong to the setup, not to the first dispatch */
		snprintf(w->error, sizeof(w->error), "%s", crh_last_error());
		ok = 0;
	}
	if (!ok || !sceneOk) {
		if (!ok) logr(warning, "GPU %d: %s\n", w->device, w->error);
		w->failed = 1;
		gpuThreadDone(w);
		return NULL;
	}
	warmUp(w->ctx, w->fb, &p, w->device);
	crh_tile *share = NULL;
	const uint32_t n = gpuShare(r, w->device, G, &share);
	uint64_t pixels = 0;
	for (uint32_t i = 0; i < n; ++i) pixels += (uint64_t)(share[i].x1 - share[i].x0) * (uint64_t)(share[i].y1 - share[i].y0);
	w->state->currentTileNum = 0;
	w->uploadUs = getUs(phase);
	w->readyUs = getUs(w->frameStart);
	startTimer(&phase);
	int passes = r->prefs.sampleCount;
	if (pixels && (uint64_t)passes * pixels > FIRST_DISPATCH_PATHS) passes = (int)(FIRST_DISPATCH_PATHS / pixels);
	const int fixedPasses = getenv("CRH_DROPIN_PASSES") ? atoi(getenv("CRH_DROPIN_PASSES")) : 0;          /* dev / tests: that many passes per dispatch */
	if (fixedPasses > 0) passes = fixedPasses;
	if (passes < 1) passes = 1;
	for (int done = 0; done < r->prefs.sampleCount && r->state.isRendering && !r->state.renderAborted; ) {
		p.first_pass = done;
		p.pass_count = r->prefs.sampleCount - done < passes ? r->prefs.sampleCount - done : passes;
		struct timeval tl;
		startTimer(&tl);
		int rc = n ? crh_render_tiles(w->ctx, &p, share, n, w->fb) : CRH_OK;
		w->launchUs += getUs(tl);
		announceLaunch(w);
		if (n && (rc != CRH_OK || crh_synchronize(w->ctx) != CRH_OK)) {
			snprintf(w->error, sizeof(w->error), "%s", crh_last_error());
			logr(warning, "GPU %d: %s\n", w->device, w->error);
			w->failed = 1;
			break;
		}
		const long us = getUs(tl);
		w->dispatches++;
		done += p.pass_count;
		w->state->completedSamples = done;
		w->state->totalSamples = pixels * (uint64_t)done;
		if (done < r->prefs.sampleCount) {
			/* more to come: show what there is, and size the next dispatch by what this one cost */
			if (n && refreshOutput(w->ctx, w->fb, W, H, w->output, w->device, G) != CRH_OK) logr(warning, "GPU %d: preview: %s\n", w->device, crh_last_error());
			const double msPerPass = (double)us / 1e3 / (double)p.pass_count;
			passes = msPerPass > 0.0 ? (int)(DISPATCH_TARGET_MS / msPerPass) : passes;
			if (fixedPasses > 0) passes = fixedPasses;
			if (passes < 1) passes = 1;
		}
		while (w->state->paused && !r->state.renderAborted) sleepMSec(100);
	}
	w->renderUs = getUs(phase);
	crh_counters c;
	if (!w->failed && crh_counters_get(w->ctx, &c) == CRH_OK) w->rays = c.rays;
	{ float last = 0.0f; uint64_t launches = 0; if (!w->failed) crh_kernel_time_ms(w->ctx, &last, &w->kernelMs, &launches); }
	free(share);
	gpuThreadDone(w);
	return NULL;
}

struct commWarm { int devices[MAX_GPUS]; int n; int rc; long us; };
static void *commThread(void *arg) {
	struct commWarm *cw = threadUserData(arg);
	struct timeval t;
	startTimer(&t);
	cw->rc = crh_frames_prepare(cw->devices, cw->n);
	cw->us = getUs(t);
	return NULL;
}

/* the G per-GPU framebuffers -> GPU 0's: the strips (1 / G of the frame per link), or — CRH_FRAMES=reduce, or a librccl without send / receive — one ncclReduce */
static void assembleFrame(crh_ctx **ctxs, float **fbs, int gpus, int W, int H) {
	if (gpus < 2) return;
	const char *how = getenv("CRH_FRAMES");
	int rc = (how && !strcmp(how, "reduce")) ? CRH_ERR_UNSUPPORTED : crh_frames_gather(ctxs, fbs, gpus, W, H, CRH_STRIP_ROWS);
	if (rc == CRH_ERR_UNSUPPORTED) rc = crh_frames_reduce(ctxs, fbs, gpus, W, H);
	if (rc != CRH_OK) logr(error, "c-ray-hip: assembling the frame on GPU 0 failed: %s\n", crh_last_error());
}

/* --iterative: returns the rays traced, fills state.renderBuffer and output */
static uint64_t renderInteractive(struct renderer *r, struct texture *output, const crh_scene_desc *scene, int gpus) {
	const int W = (int)r->prefs.imageWidth, H = (int)r->prefs.imageHeight;
	crh_ctx *ctx[MAX_GPUS];
	float *fb[MAX_GPUS];
	crh_tile *tiles[MAX_GPUS];
	uint32_t ntiles[MAX_GPUS];
	/* the process's pooled contexts (round 5, ADVICE r04: this mode used to create a second context per GPU beside the pool's); several GPUs: one layout compile */
	crh_compiled_scene *compiled = NULL;
	if (gpus > 1 && crh_scene_compile(scene, frameWalk(), &compiled) != CRH_OK) logr(error, "c-ray-hip: %s\n", crh_last_error());          /* (the pooled contexts' CRH_OPT_WALK; the Halton sampler keeps the binary walk either way) */
	for (int g = 0; g < gpus; ++g) {
		ctx[g] = poolAcquire(g);
		if (!ctx[g] || crh_counters_reset(ctx[g]) != CRH_OK || crh_set_option(ctx[g], CRH_OPT_SAMPLER, CRH_SAMPLER_HALTON) != CRH_OK ||
			crh_set_option(ctx[g], CRH_OPT_COUNTER_LEVEL, 1) != CRH_OK || (compiled ? crh_scene_upload_compiled(ctx[g], compiled) : crh_scene_upload(ctx[g], scene)) != CRH_OK ||
			crh_framebuffer_alloc(ctx[g], W, H, &fb[g]) != CRH_OK)
			logr(error, "c-ray-hip: GPU %i: %s\n", g, crh_last_error());
		ntiles[g] = gpuShare(r, g, gpus, &tiles[g]);
	}
	crh_compiled_scene_free(compiled);
	const int passes = r->prefs.sampleCount - 1;             /* finishedPasses runs from 1 while < sampleCount (renderer.c:199, tile.c:52) */
	crh_render_params p;
	memset(&p, 0, sizeof(p));
	p.image_width = W; p.image_height = H; p.max_passes = r->prefs.sampleCount; p.bounces = r->prefs.bounces;
	for (int g = 0; g < gpus; ++g) warmUp(ctx[g], fb[g], &p, g);
	int done = 0;
	int chunk = 1;                                           /* the first preview after one pass; then as many passes per dispatch as take about a display refresh */
	int dispatches = 0;
	struct timeval loop;
	startTimer(&loop);
	while (done < passes && !r->state.renderAborted) {
		if (chunk > passes - done) chunk = passes - done;
		p.first_pass = done; p.pass_count = chunk;
		struct timeval tc;
		startTimer(&tc);
		for (int g = 0; g < gpus; ++g)
			if (ntiles[g] && crh_render_tiles(ctx[g], &p, tiles[g], ntiles[g], fb[g]) != CRH_OK) logr(error, "c-ray-hip: GPU %i: %s\n", g, crh_last_error());
		/* the 8-bit frame of this chunk, converted where the float buffer lives (stream order: after the chunk's kernel) */
		for (int g = 0; g < gpus; ++g)
			if (ntiles[g] && refreshOutput(ctx[g], fb[g], W, H, output, g, gpus) != CRH_OK) logr(error, "c-ray-hip: GPU %i: %s\n", g, crh_last_error());
		const double chunkMs = (double)getUs(tc) / 1e3;
		done += chunk;
		++dispatches;
		r->state.finishedPasses = done + 1;
		for (int g = 0; g < gpus; ++g) {
			r->state.threadStates[g].completedSamples = done;
			r->state.threadStates[g].totalSamples += (uint64_t)chunk;
		}
		getKeyboardInput(r);
		drawWindow(r, output);
		while (r->state.threadStates[0].paused && !r->state.renderAborted) { getKeyboardInput(r); sleepMSec(100); }
		/* ~16 ms of GPU work between two redraws: the host side of a chunk (launch, 8-bit download, redraw) stays a small part of it */
		const double msPerPass = chunkMs / (double)chunk;
		int next = msPerPass > 0.0 ? (int)(16.0 / msPerPass) : chunk;
		if (next > 2 * chunk) next = 2 * chunk;              /* grow gently: the first passes' previews are the ones a user watches */
		chunk = next < 1 ? 1 : (next > 256 ? 256 : next);
	}
	const long loopUs = getUs(loop);
	/* CRH_DUMP_STATS: how busy the GPU was while the frame converged (kernel time of GPU 0 / wall time of the loop: launches, 8-bit downloads, redraws are the rest) */
	const char *statsPath = getenv("CRH_DUMP_STATS");
	if (statsPath) {
		double kernelMs = 0.0;
		float last = 0.0f;
		uint64_t launches = 0;
		crh_kernel_time_ms(ctx[0], &last, &kernelMs, &launches);
		FILE *f = fopen(statsPath, "w");
		if (f) {
			fprintf(f, "{\"mode\": \"iterative\", \"gpus\": %d, \"width\": %d, \"height\": %d, \"passes\": %d, \"dispatches\": %d, \"loop_ms\": %.3f, \"kernel_ms\": %.3f, \"gpu_busy\": %.4f}\n",
					gpus, W, H, done, dispatches, loopUs / 1e3, kernelMs, loopUs > 0 ? kernelMs / (loopUs / 1e3) : 0.0);
			fclose(f);
		}
	}
	/* the float buffer: once, at the end */
	assembleFrame(ctx, fb, gpus, W, H);
	if (crh_framebuffer_download(ctx[0], fb[0], W, H, r->state.renderBuffer->data.float_p) != CRH_OK) logr(error, "c-ray-hip: download: %s\n", crh_last_error());
	uint64_t rays = 0;
	for (int g = 0; g < gpus; ++g) {
		crh_counters c;
		if (crh_counters_get(ctx[g], &c) == CRH_OK) rays += c.rays;
		crh_framebuffer_free(ctx[g], fb[g]);
		poolRelease(g, ctx[g], crh_set_option(ctx[g], CRH_OPT_SAMPLER, CRH_SAMPLER_RANDOM) == CRH_OK);          /* (back to the pool as it came: the frame renderer's sampler) */
		free(tiles[g]);
		r->state.threadStates[g].threadComplete = true;
	}
	return rays;
}

/* The strips of the GPUs that failed, rendered by the ones that did not (tile.c:31-42 is the reference's version of this: a lost worker's tiles are handed out
 * again). Strip i of the frame belongs to GPU i mod G (share.h); the strips of a failed GPU are dealt round-robin to the survivors, which render them for ALL
 * passes into their own (there still zero) framebuffers, whatever the failed GPU had got done. Returns 0, or -1 with the survivor's message logged. */
static int redealFailedShares(struct renderer *r, struct gpuWorker *workers, int gpus) {
	const int W = (int)r->prefs.imageWidth, H = (int)r->prefs.imageHeight;
	int survivors[MAX_GPUS], ns = 0;
	for (int g = 0; g < gpus; ++g) if (!workers[g].failed) survivors[ns++] = g;
	if (!ns) return -1;
	crh_tile *extra[MAX_GPUS];
	uint32_t nExtra[MAX_GPUS];
	const size_t cap = (size_t)crh_strip_share_max(H, gpus) * (size_t)gpus;
	for (int s = 0; s < ns; ++s) { extra[s] = calloc(cap ? cap : 1, sizeof(crh_tile)); nExtra[s] = 0; }
	int next = 0, strips = 0;
	for (int g = 0; g < gpus; ++g) {
		if (!workers[g].failed) continue;
		crh_tile *t = NULL;
		const uint32_t n = gpuShare(r, g, gpus, &t);
		for (uint32_t i = 0; i < n; ++i, ++strips) { extra[next][nExtra[next]++] = t[i]; next = (next + 1) % ns; }
		free(t);
	}
	logr(warning, "c-ray-hip: re-dealing the %i strips of the failed GPUs to the %i remaining GPU%s\n", strips, ns, PLURAL(ns));
	crh_render_params p;
	memset(&p, 0, sizeof(p));
	p.image_width = W; p.image_height = H; p.max_passes = r->prefs.sampleCount; p.bounces = r->prefs.bounces;
	uint64_t most = 1;
	for (int s = 0; s < ns; ++s) {
		uint64_t px = 0;
		for (uint32_t i = 0; i < nExtra[s]; ++i) px += (uint64_t)(extra[s][i].x1 - extra[s][i].x0) * (uint64_t)(extra[s][i].y1 - extra[s][i].y0);
		if (px > most) most = px;
	}
	int passes = r->prefs.sampleCount;
	if ((uint64_t)passes * most > FIRST_DISPATCH_PATHS) passes = (int)(FIRST_DISPATCH_PATHS / most);
	if (passes < 1) passes = 1;
	int rc = 0;
	for (int done = 0; done < r->prefs.sampleCount && !rc && !r->state.renderAborted; done += p.pass_count) {
		p.first_pass = done;
		p.pass_count = r->prefs.sampleCount - done < passes ? r->prefs.sampleCount - done : passes;
		for (int s = 0; s < ns && !rc; ++s)          /* every survivor's dispatch is in flight before the first one is waited for */
			if (nExtra[s] && crh_render_tiles(workers[survivors[s]].ctx, &p, extra[s], nExtra[s], workers[survivors[s]].fb) != CRH_OK) rc = -1;
		for (int s = 0; s < ns && !rc; ++s)
			if (nExtra[s] && crh_synchronize(workers[survivors[s]].ctx) != CRH_OK) rc = -1;
		if (rc) logr(warning, "c-ray-hip: re-dealt strips: %s\n", crh_last_error());
		getKeyboardInput(r);
	}
	for (int s = 0; s < ns; ++s) {
		crh_counters c;
		if (!rc && crh_counters_get(workers[survivors[s]].ctx, &c) == CRH_OK) workers[survivors[s]].rays = c.rays;
		free(extra[s]);
	}
	return rc;
}

/* The frame of a run that lost a GPU: the survivors' float buffers merged on the host (one owner per pixel, 0.0f everywhere else), and the 8-bit output the way
 * renderer.c:296-300 makes it, pixel by pixel, with the reference's own colorToSRGB / setPixel. */
static int assembleOnHost(struct renderer *r, struct gpuWorker *workers, int gpus, struct texture *output) {
	const int W = (int)r->prefs.imageWidth, H = (int)r->prefs.imageHeight;
	float *dst = r->state.renderBuffer->data.float_p;
	float *tmp = malloc((size_t)W * H * 3 * sizeof(float));
	if (!tmp) return -1;
	memset(dst, 0, (size_t)W * H * 3 * sizeof(float));
	for (int g = 0; g < gpus; ++g) {
		if (workers[g].failed) continue;
		if (crh_framebuffer_download(workers[g].ctx, workers[g].fb, W, H, tmp) != CRH_OK) { free(tmp); return -1; }
		for (size_t i = 0; i < (size_t)W * H * 3; ++i) if (tmp[i] != 0.0f) dst[i] = tmp[i];
	}
	free(tmp);
	for (int y = 0; y < H; ++y)
		for (int x = 0; x < W; ++x)
			setPixel(output, colorToSRGB(textureGetPixel(r->state.renderBuffer, x, y, false)), x, y);
	return 0;
}

struct texture *renderFrame(struct renderer *r) {
	const int W = (int)r->prefs.imageWidth, H = (int)r->prefs.imageHeight;
	struct timeval frame;
	startTimer(&frame);
	struct texture *output = newTexture(char_p, r->prefs.imageWidth, r->prefs.imageHeight, 3);

	int gpus = crh_device_count();
	const char *cap = getenv("CRAY_HIP_DEVICES");
	if (cap && atoi(cap) > 0 && atoi(cap) < gpus) gpus = atoi(cap);
	if (gpus > MAX_GPUS) gpus = MAX_GPUS;
	if (gpus < 1) logr(error, "c-ray-hip: no HIP device visible (this renderer has no CPU path)\n");

	logr(info, "Starting C-ray MI355X renderer for frame %i\n", r->prefs.imgCount);
	logr(info, "Rendering at %i x %i, %i samples, %i bounces on %i GPU%s.\n", W, H, r->prefs.sampleCount, r->prefs.bounces, gpus, PLURAL(gpus));

	r->state.isRendering = true;
	r->state.renderAborted = false;
	r->state.saveImage = true;
	r->prefs.threadCount = gpus;              /* ui.c indexes threadStates[0..threadCount) */
	r->state.threads = calloc((size_t)gpus, sizeof(*r->state.threads));
	r->state.threadStates = calloc((size_t)gpus, sizeof(*r->state.threadStates));

	/* the RCCL communicators of a multi-GPU frame: created beside everything else, not inside the frame's gather */
	struct commWarm warm = {.n = gpus};
	struct crThread warmThread = {.threadFunc = commThread, .userData = &warm};
	bool warming = false;
	if (gpus > 1) {
		for (int g = 0; g < gpus; ++g) warm.devices[g] = g;
		warming = threadStart(&warmThread) == 0;
	}

	crh_scene_desc scene;
	struct timeval phase;
	if (isSet("interactive")) {
		startTimer(&phase);
		if (crh_flatten_world(r, &scene) != CRH_OK) logr(error, "c-ray-hip: the scene cannot be flattened for the GPU\n");
		logr(info, "Pathtracing iteratively...\n");
		for (int g = 0; g < gpus; ++g)
			r->state.threadStates[g] = (struct renderThreadState){.thread_num = g, .renderer = r, .output = output, .currentTileNum = -1};
		if (warming) threadWait(&warmThread);
		const uint64_t rays = renderInteractive(r, output, &scene, gpus);
		r->state.isRendering = false;
		const char *dumpi = getenv("CRH_DUMP_F32");
		if (dumpi) {
			FILE *f = fopen(dumpi, "wb");
			if (f) { fwrite(r->state.renderBuffer->data.float_p, sizeof(float), (size_t)W * H * 3, f); fclose(f); }
		}
		logr(info, "%llu rays traced on %i GPU%s.\n", (unsigned long long)rays, gpus, PLURAL(gpus));
		crh_flatten_free(&scene);
		return output;
	}
	struct frameSync sync;
	pthread_mutex_init(&sync.mu, NULL);
	pthread_cond_init(&sync.cv, NULL);
	sync.sceneReady = 0; sync.finished = 0; sync.launched = 0; sync.compiled = NULL;
	struct gpuWorker workers[MAX_GPUS];
	memset(workers, 0, sizeof(workers));
	for (int i = 0; i < r->state.tileCount; ++i) r->state.renderTiles[i].isRendering = true;      /* every GPU works on the whole frame (strips): all tiles are "being rendered" until the frame is done */
	for (int g = 0; g < gpus; ++g) {
		r->state.threadStates[g] = (struct renderThreadState){.thread_num = g, .renderer = r, .output = output, .currentTileNum = -1};
		workers[g] = (struct gpuWorker){.r = r, .state = &r->state.threadStates[g], .scene = &scene, .device = g, .sync = &sync, .output = output, .frameStart = frame};
		r->state.threads[g] = (struct crThread){.threadFunc = gpuThread, .userData = &workers[g]};
		if (threadStart(&r->state.threads[g])) logr(error, "Failed to create the dispatch thread of GPU %i.\n", g);
		r->state.activeThreads++;
	}
	/* main thread: flatten the scene while the GPU threads set themselves up */
	startTimer(&phase);
	int frc = crh_flatten_world(r, &scene);
	const long flattenUs = getUs(phase);
	/* several GPUs: ONE layout compile, here, behind the flattener; the GPU threads only copy (the reference builds its scene once and every worker reads it:
	 * src/datatypes/scene.c:111-213). One GPU: its crh_scene_upload compiles with the texels' copy beside it */
	crh_compiled_scene *compiled = NULL;
	if (frc == CRH_OK && gpus > 1 && !getenv("CRH_DROPIN_COMPILE_PER_GPU") && crh_scene_compile(&scene, frameWalk(), &compiled) != CRH_OK) {
		logr(warning, "c-ray-hip: %s\n", crh_last_error());
		frc = CRH_ERR_INVALID;
	}
	const long compileUs = getUs(phase) - flattenUs;
	pthread_mutex_lock(&sync.mu);
	sync.compiled = compiled;
	sync.sceneReady = frc == CRH_OK ? 1 : -1;
	pthread_cond_broadcast(&sync.cv);
	pthread_mutex_unlock(&sync.mu);

	/* main thread: keep the preview window / key handling alive while the GPUs work (ui.c contract); a finishing GPU thread wakes it at once */
	for (;;) {
		pthread_mutex_lock(&sync.mu);
		if (sync.finished < gpus) {
			struct timespec until;
			clock_gettime(CLOCK_REALTIME, &until);
			until.tv_nsec += (r->state.renderAborted ? 1 : 16) * 1000000L;
			if (until.tv_nsec >= 1000000000L) { until.tv_sec++; until.tv_nsec -= 1000000000L; }
			pthread_cond_timedwait(&sync.cv, &sync.mu, &until);
		}
		const int finished = sync.finished, launched = sync.launched;
		pthread_mutex_unlock(&sync.mu);
		/* every GPU has its copy of the scene and is busy with its first dispatch: the flattened arrays (70 MB for hdr.json) go back NOW, while this thread has nothing
		 * to do — at the end of renderFrame() the same free() is milliseconds of the frame (a process that has the GPU open gives pages back slowly, and what its threads
		 * launch or wait for meanwhile waits too: only when a GPU's share is long enough to hide it — 2^26 paths, like the library's own release, cray_hip.hip) */
		if (launched == gpus && scene.struct_size && (uint64_t)W * H * (uint64_t)r->prefs.sampleCount / (uint64_t)gpus >= ((uint64_t)1 << 26)) {
			crh_flatten_free(&scene);
			crh_compiled_scene_free(compiled); compiled = NULL;          /* (every GPU thread has announced its launch: the compiled layout has been read for the last time) */
		}
		if (finished == gpus) break;
		getKeyboardInput(r);
		drawWindow(r, output);
	}
	for (int g = 0; g < gpus; ++g) threadWait(&r->state.threads[g]);
	if (warming) threadWait(&warmThread);
	r->state.activeThreads = 0;
	r->state.isRendering = false;
	if (frc != CRH_OK) logr(error, "c-ray-hip: the scene cannot be flattened for the GPU (%i)\n", frc);

	int failed = 0;
	uint64_t rays = 0;
	const char *firstError = "";
	for (int g = gpus - 1; g >= 0; --g) { failed += workers[g].failed; rays += workers[g].rays; if (workers[g].failed) firstError = workers[g].error; }
	if (failed) {
		logr(warning, "c-ray-hip: %i GPU dispatch thread%s failed: %s\n", failed, PLURAL(failed), firstError);
		if (failed == gpus) logr(error, "c-ray-hip: no GPU is left to render the frame: %s\n", firstError);
		if (!r->state.renderAborted && redealFailedShares(r, workers, gpus) != 0) logr(error, "c-ray-hip: the strips of the failed GPU%s could not be rendered elsewhere\n", PLURAL(failed));
		rays = 0;
		for (int g = 0; g < gpus; ++g) if (!workers[g].failed) rays += workers[g].rays;
	}
	if (warming && warm.rc != CRH_OK) logr(warning, "c-ray-hip: RCCL set-up beside the frame failed; the gather will retry\n");
	if (!r->state.renderAborted)
		for (int i = 0; i < r->state.tileCount; ++i) { r->state.renderTiles[i].isRendering = false; r->state.renderTiles[i].renderComplete = true; }

	startTimer(&phase);
	struct texture *buf = r->state.renderBuffer;
	long gatherUs = 0, resolveUs = 0, downloadUs = 0;
	if (failed) {
		/* a GPU was lost: the survivors hold the whole frame between them, but not in the strip pattern the gather moves */
		if (assembleOnHost(r, workers, gpus, output) != 0) logr(error, "c-ray-hip: assembling the frame on the host failed: %s\n", crh_last_error());
		gatherUs = getUs(phase);
	} else {
		/* assemble the frame on GPU 0 (RCCL over xGMI; the shares are disjoint) */
		crh_ctx *ctxs[MAX_GPUS];
		float *fbs[MAX_GPUS];
		for (int g = 0; g < gpus; ++g) { ctxs[g] = workers[g].ctx; fbs[g] = workers[g].fb; }
		assembleFrame(ctxs, fbs, gpus, W, H);
		gatherUs = getUs(phase);
		startTimer(&phase);
		/* 8-bit output exactly like renderer.c:294-300 — on the device — and the float buffer: one download each */
		if (crh_framebuffer_to_srgb8(workers[0].ctx, workers[0].fb, W, H, output->data.byte_p) != CRH_OK)
			logr(error, "c-ray-hip: sRGB conversion failed: %s\n", crh_last_error());
		resolveUs = getUs(phase);
		startTimer(&phase);
		if (crh_framebuffer_download(workers[0].ctx, workers[0].fb, W, H, buf->data.float_p) != CRH_OK)
			logr(error, "c-ray-hip: framebuffer download failed: %s\n", crh_last_error());
		downloadUs = getUs(phase);
	}
	const long workUs = getUs(frame);            /* set-up + dispatches + gather + conversion + downloads */
	const char *dump = getenv("CRH_DUMP_F32");
	if (dump) {
		FILE *f = fopen(dump, "wb");
		if (f) { fwrite(buf->data.float_p, sizeof(float), (size_t)W * H * 3, f); fclose(f); }
	}
	logr(info, "%llu rays traced on %i GPU%s.\n", (unsigned long long)rays, gpus, PLURAL(gpus));

	for (int g = 0; g < gpus; ++g) {
		if (workers[g].ctx && workers[g].fb) crh_framebuffer_free(workers[g].ctx, workers[g].fb);
		poolRelease(g, workers[g].ctx, !workers[g].failed);
	}
	pthread_mutex_destroy(&sync.mu);
	pthread_cond_destroy(&sync.cv);
	crh_flatten_free(&scene);
	crh_compiled_scene_free(compiled);
	const long frameUs = getUs(frame);          /* the whole of renderFrame(), teardown included: what the timer of c-ray.c:279-281 sees */
	/* CRH_DUMP_STATS=<path>: where the frame went, for bench.py's `dropin` object. render_phase_ms is SURVEY.md 8(d)'s phase — the timer
	 * of src/c-ray.c:279-281 around renderFrame() minus the set-up (flatten / context / upload: everything before the slowest GPU was ready
	 * to dispatch): dispatches + gather + 8-bit conversion + downloads + the host in between. */
	const char *statsPath = getenv("CRH_DUMP_STATS");
	if (statsPath) {
		long contextUs = 0, uploadUs = 0, renderUs = 0, launchUs = 0, readyUs = 0;
		double kernelMs = 0.0;
		int dispatches = 0;
		for (int g = 0; g < gpus; ++g) {
			if (workers[g].contextUs > contextUs) contextUs = workers[g].contextUs;
			if (workers[g].uploadUs > uploadUs) uploadUs = workers[g].uploadUs;
			if (workers[g].readyUs > readyUs) readyUs = workers[g].readyUs;
			if (workers[g].renderUs > renderUs) renderUs = workers[g].renderUs;
			if (workers[g].launchUs > launchUs) launchUs = workers[g].launchUs;
			if (workers[g].kernelMs > kernelMs) kernelMs = workers[g].kernelMs;
			if (workers[g].dispatches > dispatches) dispatches = workers[g].dispatches;
		}
		FILE *f = fopen(statsPath, "w");
		if (f) {
			fprintf(f, "{\"gpus\": %d, \"width\": %d, \"height\": %d, \"samples\": %d, \"bounces\": %d, \"rays\": %llu, \"flatten_ms\": %.3f, \"compile_once_ms\": %.3f, "
					"\"context_ms\": %.3f, \"upload_ms\": %.3f, \"context_upload_ms\": %.3f, \"setup_ms\": %.3f, \"render_ms\": %.3f, \"kernel_ms\": %.3f, \"launch_host_ms\": %.3f, "
					"\"dispatches\": %d, \"reduce_download_ms\": %.3f, \"gather_ms\": %.3f, \"resolve_srgb_ms\": %.3f, \"download_ms\": %.3f, \"rccl_setup_ms\": %.3f, "
					"\"frame_ms\": %.3f, \"teardown_ms\": %.3f, \"render_phase_ms\": %.3f}\n",
					gpus, W, H, r->prefs.sampleCount, r->prefs.bounces, (unsigned long long)rays, flattenUs / 1e3, compileUs / 1e3, contextUs / 1e3, uploadUs / 1e3,
					(contextUs + uploadUs) / 1e3, readyUs / 1e3, renderUs / 1e3, kernelMs, launchUs / 1e3, dispatches, (gatherUs + downloadUs) / 1e3, gatherUs / 1e3,
					resolveUs / 1e3, downloadUs / 1e3, warm.us / 1e3, frameUs / 1e3, (frameUs - workUs) / 1e3, (workUs - readyUs) / 1e3);
			fclose(f);
		}
	}

	return output;
}

struct renderer *newRenderer(void) {
	struct renderer *r = calloc(1, sizeof(*r));
	/* the estimators in ui.c / renderer statistics divide by these, so they start at 1 (renderer.c:331-333) */
	r->state.timeSampleCount = 1;
	r->state.finishedPasses = 1;
	r->state.avgTileTime = 1;
	r->state.tileMutex = createMutex();
	r->state.timer = calloc(1, sizeof(*r->state.timer));
	if (!g_vertices) allocVertexBuffers();
	if (!isSet("is_worker")) poolPrefetch();          /* (a cluster worker makes its own context when its first tile arrives: host/access/worker_access.c) */
	return r;
}

void destroyRenderer(struct renderer *r) {
	if (!r) return;
	poolDestroy();
	destroyScene(r->scene);
	destroyTexture(r->state.uiBuffer);
	destroyTexture(r->state.renderBuffer);
	destroyVertexBuffers();
	free(r->state.threadStates);
	free(r->state.threads);
	free(r->state.renderTiles);
	free(r->state.tileMutex);
	free(r->state.timer);
	free(r->prefs.assetPath);
	free(r->prefs.imgFilePath);
	free(r->prefs.imgFileName);
	free(r);
}

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
 * (renderer.c:258-327) there is ONE host thread per GPU. Each flattens nothing itself — the scene is
 * flattened once (flatten.c) — owns a crh_ctx (libcray_hip.so) and renders ITS share of the frame — one GPU: the whole frame as one
 * region; several: every G-th 4-row strip — with ONE crh_render_tiles() dispatch per 512 passes (the
 * persistent kernel balances the work units inside a dispatch itself; a drain per tile batch would idle the GPU). Then
 * the per-GPU float framebuffers (disjoint pixels, zero elsewhere) are summed onto GPU 0 with
 * RCCL over xGMI (crh_frames_reduce), downloaded into state.renderBuffer, and converted to the 8-bit sRGB
 * output with the reference's own colorToSRGB()/setPixel() on the host.
 *
 * --iterative (renderThreadInteractive, renderer.c:184-250): passes 1 .. sampleCount-1 with the Halton sampler, the frame
 * shown while it converges. Here the main thread drives every GPU pass-chunk by pass-chunk (every G-th 4-row strip belongs to
 * one GPU for the whole frame, so each pixel's running mean stays on one GPU), gathers the strips on the host after each chunk and redraws.
 * The result is the reference's single-thread result: with several threads the reference itself races on
 * state.finishedPasses and is not reproducible.
 *
 * Environment: CRAY_HIP_DEVICES=<n> caps the number of GPUs used; CRH_DUMP_F32=<path> dumps the float buffer.
 * No GPU => logr(error, ...) (which exits, src/utils/logging.c:69-73): there is no CPU fallback in this file.
 */
#include <stdlib.h>
#include <stdio.h>
#include <string.h>
#include <stdbool.h>

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

struct gpuWorker {
	struct renderer *r;
	struct renderThreadState *state;
	const crh_scene_desc *scene;
	crh_ctx *ctx;
	float *fb;
	int device;
	int failed;
	char error[256];      /* crh_last_error() is per thread: the failing dispatch thread keeps its message here */
	uint64_t rays;
	long setupUs, renderUs;   /* context + upload + framebuffer; dispatch loop (first launch to last sync) */
	double kernelMs;          /* GPU time of the dispatches (HIP events around the kernels) */
	long launchUs;            /* host time inside crh_render_tiles (work list, copies, launch) */
};

/* passes per dispatch: one dispatch keeps the whole GPU busy from its first work unit to its last (no drain in between), so a
 * frame is cut only along the pass axis, and only when it is long enough that the preview window / abort key should get a turn */
#define PASSES_PER_DISPATCH 512

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

	/* counter level 1: this host reports rays only (the detailed counters cost ~20 % of the kernel's time) */
	if (crh_context_create(w->device, NULL, &w->ctx) != CRH_OK || crh_set_option(w->ctx, CRH_OPT_COUNTER_LEVEL, 1) != CRH_OK ||
		crh_scene_upload(w->ctx, w->scene) != CRH_OK || crh_framebuffer_alloc(w->ctx, W, H, &w->fb) != CRH_OK ||
		crh_synchronize(w->ctx) != CRH_OK) {           /* the scene copies are asynchronous: they belong to the setup, not to the first dispatch */
		snprintf(w->error, sizeof(w->error), "%s", crh_last_error());
		logr(warning, "GPU %d: %s\n", w->device, w->error);
		w->failed = 1;
		w->state->threadComplete = true;
		return NULL;
	}
	crh_tile *share = NULL;
	const uint32_t n = gpuShare(r, w->device, G, &share);
	uint64_t pixels = 0;
	for (uint32_t i = 0; i < n; ++i) pixels += (uint64_t)(share[i].x1 - share[i].x0) * (uint64_t)(share[i].y1 - share[i].y0);
	if (G == 1) for (int i = 0; i < r->state.tileCount; ++i) r->state.renderTiles[i].isRendering = true;
	w->state->currentTileNum = 0;
	w->setupUs = getUs(phase);
	startTimer(&phase);
	for (int done = 0; done < r->prefs.sampleCount && r->state.isRendering && !r->state.renderAborted; ) {
		p.first_pass = done;
		p.pass_count = r->prefs.sampleCount - done < PASSES_PER_DISPATCH ? r->prefs.sampleCount - done : PASSES_PER_DISPATCH;
		struct timeval tl;
		startTimer(&tl);
		int rc = n ? crh_render_tiles(w->ctx, &p, share, n, w->fb) : CRH_OK;
		w->launchUs += getUs(tl);
		if (n && (rc != CRH_OK || crh_synchronize(w->ctx) != CRH_OK)) {
			snprintf(w->error, sizeof(w->error), "%s", crh_last_error());
			logr(warning, "GPU %d: %s\n", w->device, w->error);
			w->failed = 1;
			break;
		}
		done += p.pass_count;
		w->state->completedSamples = done;
		w->state->totalSamples = pixels * (uint64_t)done;
		while (w->state->paused && !r->state.renderAborted) sleepMSec(100);
	}
	w->renderUs = getUs(phase);
	if (G == 1 && !w->failed && !r->state.renderAborted)
		for (int i = 0; i < r->state.tileCount; ++i) { r->state.renderTiles[i].isRendering = false; r->state.renderTiles[i].renderComplete = true; }
	crh_counters c;
	if (!w->failed && crh_counters_get(w->ctx, &c) == CRH_OK) w->rays = c.rays;
	{ float last = 0.0f; uint64_t launches = 0; if (!w->failed) crh_kernel_time_ms(w->ctx, &last, &w->kernelMs, &launches); }
	free(share);
	w->state->currentTileNum = -1;
	w->state->threadComplete = true;
	return NULL;
}

/* renderer.c:294-300 for the whole frame: colorToSRGB + setPixel truncation on the host */
static void resolveOutput(struct renderer *r, struct texture *output) {
	const int W = (int)r->prefs.imageWidth, H = (int)r->prefs.imageHeight;
	for (int y = 0; y < H; ++y)
		for (int x = 0; x < W; ++x)
			setPixel(output, colorToSRGB(textureGetPixel(r->state.renderBuffer, x, y, false)), x, y);
}

/* --iterative: returns the rays traced, fills state.renderBuffer and output */
static uint64_t renderInteractive(struct renderer *r, struct texture *output, const crh_scene_desc *scene, int gpus) {
	const int W = (int)r->prefs.imageWidth, H = (int)r->prefs.imageHeight;
	crh_ctx *ctx[MAX_GPUS];
	float *fb[MAX_GPUS];
	crh_tile *tiles[MAX_GPUS];
	uint32_t ntiles[MAX_GPUS];
	float *gather = malloc(sizeof(float) * (size_t)W * H * 3);
	for (int g = 0; g < gpus; ++g) {
		if (crh_context_create(g, NULL, &ctx[g]) != CRH_OK || crh_set_option(ctx[g], CRH_OPT_SAMPLER, CRH_SAMPLER_HALTON) != CRH_OK ||
			crh_set_option(ctx[g], CRH_OPT_COUNTER_LEVEL, 1) != CRH_OK || crh_scene_upload(ctx[g], scene) != CRH_OK ||
			crh_framebuffer_alloc(ctx[g], W, H, &fb[g]) != CRH_OK)
			logr(error, "c-ray-hip: GPU %i: %s\n", g, crh_last_error());
		ntiles[g] = gpuShare(r, g, gpus, &tiles[g]);
	}
	const int passes = r->prefs.sampleCount - 1;             /* finishedPasses runs from 1 while < sampleCount (renderer.c:199, tile.c:52) */
	crh_render_params p;
	memset(&p, 0, sizeof(p));
	p.image_width = W; p.image_height = H; p.max_passes = r->prefs.sampleCount; p.bounces = r->prefs.bounces;
	int done = 0;
	while (done < passes && !r->state.renderAborted) {
		const int chunk = done < 8 ? 1 : (passes - done < 8 ? passes - done : 8);      /* first previews quickly, then fewer round trips */
		p.first_pass = done; p.pass_count = chunk;
		for (int g = 0; g < gpus; ++g)
			if (ntiles[g] && crh_render_tiles(ctx[g], &p, tiles[g], ntiles[g], fb[g]) != CRH_OK) logr(error, "c-ray-hip: GPU %i: %s\n", g, crh_last_error());
		for (int g = 0; g < gpus; ++g)
			if (crh_synchronize(ctx[g]) != CRH_OK) logr(error, "c-ray-hip: GPU %i: %s\n", g, crh_last_error());
		done += chunk;
		r->state.finishedPasses = done + 1;
		for (int g = 0; g < gpus; ++g) {                      /* gather: every tile from the GPU that owns it */
			r->state.threadStates[g].completedSamples = done;
			r->state.threadStates[g].totalSamples += (uint64_t)chunk;
			float *dst = r->state.renderBuffer->data.float_p;
			const float *src = gather;
			if (gpus == 1) dst = r->state.renderBuffer->data.float_p, src = NULL;
			if (crh_framebuffer_download(ctx[g], fb[g], W, H, gpus == 1 ? dst : gather) != CRH_OK) logr(error, "c-ray-hip: download: %s\n", crh_last_error());
			if (!src) continue;
			for (uint32_t t = 0; t < ntiles[g]; ++t)
				for (int y = tiles[g][t].y0; y < tiles[g][t].y1; ++y) {
					const size_t row = ((size_t)(H - (y + 1)) * W + (size_t)tiles[g][t].x0) * 3;      /* texture.c:24-28 row order */
					memcpy(dst + row, src + row, sizeof(float) * 3 * (size_t)(tiles[g][t].x1 - tiles[g][t].x0));
				}
		}
		resolveOutput(r, output);
		getKeyboardInput(r);
		drawWindow(r, output);
		while (r->state.threadStates[0].paused && !r->state.renderAborted) { getKeyboardInput(r); sleepMSec(100); }
	}
	uint64_t rays = 0;
	for (int g = 0; g < gpus; ++g) {
		crh_counters c;
		if (crh_counters_get(ctx[g], &c) == CRH_OK) rays += c.rays;
		crh_framebuffer_free(ctx[g], fb[g]);
		crh_context_destroy(ctx[g]);
		free(tiles[g]);
		r->state.threadStates[g].threadComplete = true;
	}
	free(gather);
	return rays;
}

struct texture *renderFrame(struct renderer *r) {
	const int W = (int)r->prefs.imageWidth, H = (int)r->prefs.imageHeight;
	struct texture *output = newTexture(char_p, r->prefs.imageWidth, r->prefs.imageHeight, 3);

	int gpus = crh_device_count();
	const char *cap = getenv("CRAY_HIP_DEVICES");
	if (cap && atoi(cap) > 0 && atoi(cap) < gpus) gpus = atoi(cap);
	if (gpus > MAX_GPUS) gpus = MAX_GPUS;
	if (gpus < 1) logr(error, "c-ray-hip: no HIP device visible (this renderer has no CPU path)\n");

	logr(info, "Starting C-ray MI355X renderer for frame %i\n", r->prefs.imgCount);
	logr(info, "Rendering at %i x %i, %i samples, %i bounces on %i GPU%s.\n", W, H, r->prefs.sampleCount, r->prefs.bounces, gpus, PLURAL(gpus));

	crh_scene_desc scene;
	struct timeval phase;
	startTimer(&phase);
	const int frc = crh_flatten_world(r, &scene);
	const long flattenUs = getUs(phase);
	if (frc != CRH_OK) logr(error, "c-ray-hip: the scene cannot be flattened for the GPU (%i)\n", frc);

	r->state.isRendering = true;
	r->state.renderAborted = false;
	r->state.saveImage = true;
	r->prefs.threadCount = gpus;              /* ui.c indexes threadStates[0..threadCount) */
	r->state.threads = calloc((size_t)gpus, sizeof(*r->state.threads));
	r->state.threadStates = calloc((size_t)gpus, sizeof(*r->state.threadStates));
	if (isSet("interactive")) {
		logr(info, "Pathtracing iteratively...\n");
		for (int g = 0; g < gpus; ++g)
			r->state.threadStates[g] = (struct renderThreadState){.thread_num = g, .renderer = r, .output = output, .currentTileNum = -1};
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
	struct gpuWorker workers[MAX_GPUS];
	memset(workers, 0, sizeof(workers));
	for (int g = 0; g < gpus; ++g) {
		r->state.threadStates[g] = (struct renderThreadState){.thread_num = g, .renderer = r, .output = output, .currentTileNum = -1};
		workers[g] = (struct gpuWorker){.r = r, .state = &r->state.threadStates[g], .scene = &scene, .device = g};
		r->state.threads[g] = (struct crThread){.threadFunc = gpuThread, .userData = &workers[g]};
		if (threadStart(&r->state.threads[g])) logr(error, "Failed to create the dispatch thread of GPU %i.\n", g);
		r->state.activeThreads++;
	}

	/* main thread: keep the preview window / key handling alive while the GPUs work (ui.c contract) */
	for (;;) {
		int done = 0;
		for (int g = 0; g < gpus; ++g) done += r->state.threadStates[g].threadComplete ? 1 : 0;
		if (done == gpus) break;
		getKeyboardInput(r);
		drawWindow(r, output);
		sleepMSec(r->state.renderAborted ? 1 : 4);
	}
	for (int g = 0; g < gpus; ++g) threadWait(&r->state.threads[g]);
	r->state.activeThreads = 0;
	r->state.isRendering = false;

	int failed = 0;
	uint64_t rays = 0;
	const char *firstError = "";
	for (int g = gpus - 1; g >= 0; --g) { failed += workers[g].failed; rays += workers[g].rays; if (workers[g].failed) firstError = workers[g].error; }
	if (failed) logr(error, "c-ray-hip: %i GPU dispatch thread%s failed: %s\n", failed, PLURAL(failed), firstError);

	startTimer(&phase);
	/* assemble the frame on GPU 0 (RCCL reduce over xGMI; tiles are disjoint so the sum is a gather) */
	if (gpus > 1) {
		crh_ctx *ctxs[MAX_GPUS];
		float *fbs[MAX_GPUS];
		for (int g = 0; g < gpus; ++g) { ctxs[g] = workers[g].ctx; fbs[g] = workers[g].fb; }
		if (crh_frames_reduce(ctxs, fbs, gpus, W, H) != CRH_OK) logr(error, "c-ray-hip: framebuffer reduce failed: %s\n", crh_last_error());
	}
	struct texture *buf = r->state.renderBuffer;
	if (crh_framebuffer_download(workers[0].ctx, workers[0].fb, W, H, buf->data.float_p) != CRH_OK)
		logr(error, "c-ray-hip: framebuffer download failed: %s\n", crh_last_error());

	const long gatherUs = getUs(phase);
	startTimer(&phase);
	resolveOutput(r, output);      /* 8-bit output exactly like renderer.c:294-300 */
	const long resolveUs = getUs(phase);
	/* CRH_DUMP_STATS=<path>: where the render phase (src/c-ray.c:279-281) went, for bench.py's `dropin` object */
	const char *statsPath = getenv("CRH_DUMP_STATS");
	if (statsPath) {
		long setupUs = 0, renderUs = 0, launchUs = 0;
		double kernelMs = 0.0;
		for (int g = 0; g < gpus; ++g) {
			if (workers[g].setupUs > setupUs) setupUs = workers[g].setupUs;
			if (workers[g].renderUs > renderUs) renderUs = workers[g].renderUs;
			if (workers[g].launchUs > launchUs) launchUs = workers[g].launchUs;
			if (workers[g].kernelMs > kernelMs) kernelMs = workers[g].kernelMs;
		}
		FILE *f = fopen(statsPath, "w");
		if (f) {
			fprintf(f, "{\"gpus\": %d, \"width\": %d, \"height\": %d, \"samples\": %d, \"bounces\": %d, \"rays\": %llu, \"flatten_ms\": %.3f, "
					"\"context_upload_ms\": %.3f, \"render_ms\": %.3f, \"kernel_ms\": %.3f, \"launch_host_ms\": %.3f, \"reduce_download_ms\": %.3f, \"resolve_srgb_ms\": %.3f}\n",
					gpus, W, H, r->prefs.sampleCount, r->prefs.bounces, (unsigned long long)rays, flattenUs / 1e3, setupUs / 1e3, renderUs / 1e3,
					kernelMs, launchUs / 1e3, gatherUs / 1e3, resolveUs / 1e3);
			fclose(f);
		}
	}

	const char *dump = getenv("CRH_DUMP_F32");
	if (dump) {
		FILE *f = fopen(dump, "wb");
		if (f) { fwrite(buf->data.float_p, sizeof(float), (size_t)W * H * 3, f); fclose(f); }
	}
	logr(info, "%llu rays traced on %i GPU%s.\n", (unsigned long long)rays, gpus, PLURAL(gpus));

	for (int g = 0; g < gpus; ++g) {
		crh_framebuffer_free(workers[g].ctx, workers[g].fb);
		crh_context_destroy(workers[g].ctx);
	}
	crh_flatten_free(&scene);
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
	return r;
}

void destroyRenderer(struct renderer *r) {
	if (!r) return;
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

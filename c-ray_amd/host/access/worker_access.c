/* Wrapper TU for the cluster WORKER: compiles the UNMODIFIED reference src/utils/protocol/worker.c — handshake,
 * scene / asset transfer, the TCP server loop, statistics — and swaps only the body its render threads run: instead of the CPU
 * pixel x pass loop (worker.c:138-217) a worker thread per GPU pulls BATCHES of tiles from the master with the file's own getWork(),
 * renders a batch with one crh_render_tiles() dispatch, and hands every tile back with the file's own submitWork() as the same 8-bit
 * sRGB tile texture the reference ships (protocol.c:102-114). The master cannot tell the difference: tiles are bit-identical.
 * (CRH_WORKER_FLOAT_TILES=1 ships the linear float means instead — the second half of SURVEY 8(f) rank 3 — for masters that keep a float frame.)
 *
 * How the body is swapped without editing the reference: inside this TU `workerThread` is renamed (the reference's loop stays in the
 * object, unused) and `threadStart` is routed through a shim that replaces that thread function by the GPU one. Only c-ray-hip is
 * built this way (-DCRH_GPU_BVH, c-ray_amd/host/Makefile); the oracle's binaries compile worker.c as it is. */
#ifdef CRH_GPU_BVH
#include <stdlib.h>
#include <string.h>
struct crThread;
int crh_worker_threadStart(struct crThread *t);
#define workerThread crh_reference_cpu_workerThread
#define threadStart crh_worker_threadStart
#include "utils/protocol/worker.c"
#undef threadStart
#undef workerThread
int threadStart(struct crThread *t);           /* src/utils/platform/thread.h:38 (its declaration above went to the shim's name) */

#include "cray_hip.h"
#include "../flatten.h"

#ifndef WINDOWS
/* tiles per dispatch: enough paths to fill the GPU (the persistent kernel wants >= ~8 M paths), few enough that the master's queue still
 * balances this worker against the others */
static int tilesPerBatch(const struct renderer *r) {
	const double perTile = (double)r->prefs.tileWidth * r->prefs.tileHeight * (double)r->prefs.sampleCount;
	int n = (int)(16.0e6 / (perTile > 1.0 ? perTile : 1.0)) + 1;
	if (n > 64) n = 64;
	return n;
}

static void *gpuWorkerThread(void *arg) {
	struct workerThreadState *threadState = (struct workerThreadState *)threadUserData(arg);
	struct renderer *r = threadState->renderer;
	const int sock = threadState->connectionSocket;
	struct crMutex *sockMutex = threadState->socketMutex;
	int gpus = crh_device_count();
	const char *cap = getenv("CRAY_HIP_DEVICES");
	if (cap && atoi(cap) > 0 && atoi(cap) < gpus) gpus = atoi(cap);
	if (gpus < 1) logr(error, "c-ray-hip --worker: no HIP device visible (this worker has no CPU path)\n");
	if (threadState->thread_num >= gpus) {          /* the reference starts one thread per host core: one per GPU does the work */
		threadState->threadComplete = true;
		return 0;
	}
	const int W = (int)r->prefs.imageWidth, H = (int)r->prefs.imageHeight;
	crh_scene_desc scene;
	crh_ctx *ctx = NULL;
	float *fb = NULL;
	/* A GPU that cannot be set up, or that fails in the middle of the job, ends THIS thread only (threadComplete): the tiles it held are re-issued by the
	 * master once its queue is empty (tile.c:33-42) and rendered by the other GPUs / workers. logr(error) would exit the whole worker with them. */
	if (crh_flatten_world(r, &scene) != CRH_OK) {
		logr(warning, "c-ray-hip --worker: the scene cannot be flattened for the GPU\n");
		threadState->threadComplete = true;
		return 0;
	}
	if (crh_context_create(threadState->thread_num, NULL, &ctx) != CRH_OK || crh_set_option(ctx, CRH_OPT_COUNTER_LEVEL, 1) != CRH_OK ||
		crh_scene_upload(ctx, &scene) != CRH_OK || crh_framebuffer_alloc(ctx, W, H, &fb) != CRH_OK) {
		logr(warning, "c-ray-hip --worker: GPU %i: %s\n", threadState->thread_num, crh_last_error());
		if (ctx) crh_context_destroy(ctx);
		crh_flatten_free(&scene);
		threadState->threadComplete = true;
		return 0;
	}
	crh_render_params p;
	memset(&p, 0, sizeof(p));
	p.image_width = W; p.image_height = H;
	p.first_pass = 0; p.pass_count = r->prefs.sampleCount; p.max_passes = r->prefs.sampleCount;
	p.bounces = r->prefs.bounces;

	const int batchMax = tilesPerBatch(r);
	struct renderTile *tiles = calloc((size_t)batchMax, sizeof(*tiles));
	crh_tile *rects = calloc((size_t)batchMax, sizeof(*rects));
	float *rows = NULL;
	size_t rowsCap = 0;
	bool more = true;
	threadState->completedSamples = 1;
	while (more && r->state.isRendering && !r->state.renderAborted) {
		int n = 0;
		int want = batchMax;
		while (n < want) {
			/* one request per lock: the statistics sender and the other GPUs' threads of this worker get their turn in between */
			lockMutex(sockMutex);
			struct renderTile t = getWork(sock);
			releaseMutex(sockMutex);
			if (t.tileNum == -1) { more = false; break; }
			/* tiles leave the master's queue in list order (tile.c:22-45), so a first-issue tile's number says how many are left: near the end
			 * of the frame a batch takes a quarter of them at most, and the other workers are not left idle while this one holds the tail */
			const int left = r->state.tileCount - 1 - t.tileNum;
			if (left / 4 + 1 < want - n) want = n + left / 4 + 1;
			/* once its queue is empty the master re-issues network tiles that are not back yet (tile.c:33-42) — i.e. the ones of this very
			 * batch: such a tile ends the batch (it is already in it; rendering it twice in one dispatch would race on its pixels) */
			bool mine = false;
			for (int k = 0; k < n; ++k) mine = mine || tiles[k].tileNum == t.tileNum;
			if (mine) break;
			tiles[n] = t;
			rects[n] = (crh_tile){t.begin.x, t.begin.y, t.end.x, t.end.y};
			++n;
		}
		if (n == 0) break;
		if (crh_render_tiles(ctx, &p, rects, (uint32_t)n, fb) != CRH_OK || crh_synchronize(ctx) != CRH_OK) {
			logr(warning, "c-ray-hip --worker: GPU %i: %s\n", threadState->thread_num, crh_last_error());
			break;                                     /* the batch is never submitted: the master re-issues its tiles */
		}
		threadState->completedSamples = r->prefs.sampleCount;
		for (int i = 0; i < n; ++i) {
			const struct renderTile tile = tiles[i];
			/* the tile's rows of the float buffer (stored top-down: texture.c:24-28), then exactly worker.c:163-176 per pixel */
			const int storedRow0 = H - tile.end.y, nrows = tile.end.y - tile.begin.y;
			const size_t need = (size_t)nrows * W * 3;
			if (need > rowsCap) { free(rows); rows = malloc(need * sizeof(float)); rowsCap = need; }
			if (crh_framebuffer_download(ctx, fb + (size_t)storedRow0 * W * 3, W, nrows, rows) != CRH_OK) {
				logr(warning, "c-ray-hip --worker: download: %s\n", crh_last_error());
				more = false;
				break;
			}
			/* CRH_WORKER_FLOAT_TILES=1: ship the tile's linear float means instead of 8-bit sRGB — the wire format carries either (protocol.c:102-127:
			 * isFloatPrecision). Opt-in: the reference's own master pastes whatever it gets into its 8-bit output without the sRGB transform
			 * (server.c:164-169), so it shows float tiles too dark; a master that keeps a float frame (the test's, a multi-node reduce) wants them. */
			const bool floatTiles = getenv("CRH_WORKER_FLOAT_TILES") != NULL;
			struct texture *tileBuffer = newTexture(floatTiles ? float_p : char_p, tile.width, tile.height, 3);
			for (int y = tile.end.y - 1; y > tile.begin.y - 1; --y) {
				for (int x = tile.begin.x; x < tile.end.x; ++x) {
					const float *px = rows + ((size_t)(H - (y + 1) - storedRow0) * W + (size_t)x) * 3;
					struct color output = {px[0], px[1], px[2], 0.0f};
					setPixel(r->state.renderBuffer, output, x, y);
					if (!floatTiles) output = colorToSRGB(output);
					setPixel(tileBuffer, output, x - tile.begin.x, y - tile.begin.y);
				}
			}
			threadState->totalSamples += (uint64_t)r->prefs.sampleCount;
			lockMutex(sockMutex);
			bool ok = submitWork(sock, tileBuffer, tile);
			if (ok) {
				cJSON *resp = readJSON(sock);
				const cJSON *action = resp ? cJSON_GetObjectItem(resp, "action") : NULL;
				ok = action && cJSON_IsString(action) && stringEquals(action->valuestring, "ok");
				if (resp) cJSON_Delete(resp);
			}
			releaseMutex(sockMutex);
			destroyTexture(tileBuffer);
			if (!ok) { more = false; break; }
		}
		threadState->completedSamples = 1;
	}
	free(rows); free(tiles); free(rects);
	crh_framebuffer_free(ctx, fb);
	crh_context_destroy(ctx);
	crh_flatten_free(&scene);
	threadState->threadComplete = true;
	return 0;
}

int crh_worker_threadStart(struct crThread *t) {
	if (t->threadFunc == crh_reference_cpu_workerThread) t->threadFunc = gpuWorkerThread;
	return threadStart(t);
}
#else
int crh_worker_threadStart(struct crThread *t) { return threadStart(t); }
#endif
#endif

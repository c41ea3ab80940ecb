/* Wrapper TU: compiles the UNMODIFIED reference src/accelerators/bvh.c (builder + traversal) and
 * appends read-only accessors for the file-private `struct bvh` (bvh.c:44-48). See describe.h.
 *
 * With -DCRH_GPU_BVH (the c-ray-hip host; NOT the oracle's crh-flatten, whose blobs must stay the reference's own
 * output) buildBottomLevelBvh() (bvh.c:299-301) is replaced by the GPU builder behind crh_bvh_build_triangles()
 * (SURVEY.md 8(f) row 1): same struct bvh out, same tree bit for bit; builder threads (the reference builds every mesh on its
 * own, scene.c:50-78) borrow a context from a small pool. The reference's function
 * stays in the object under another name; nothing calls it. */
#ifdef CRH_GPU_BVH
#define buildBottomLevelBvh crh_reference_buildBottomLevelBvh
#endif
#include <sys/mman.h>
#include "accelerators/bvh.c"
#include "describe.h"

const void *crh_access_bvh_nodes(const struct bvh *b) { return b ? b->nodes : NULL; }
const int  *crh_access_bvh_prims(const struct bvh *b) { return b ? b->primIndices : NULL; }
unsigned    crh_access_bvh_node_count(const struct bvh *b) { return b ? b->nodeCount : 0; }

_Static_assert(sizeof(struct bvhNode) == sizeof(crh_bvh_node), "crh_bvh_node must mirror struct bvhNode (32 B)");

#ifdef CRH_GPU_BVH
#undef buildBottomLevelBvh
#include <pthread.h>
#include <stdio.h>
#include "utils/timer.h"
#include "datatypes/vertexbuffer.h"
#include "utils/logging.h"
_Static_assert(sizeof(struct poly) == sizeof(crh_poly), "crh_poly must mirror struct poly (40 B)");

/* A small pool of builder contexts: the reference builds every mesh on its own thread (scene.c:50-78) and a crh_ctx belongs to one thread at a time, so a
 * builder thread borrows a free context (each has its own stream and scratch: uploads, builds and downloads of different meshes overlap) or waits for one.
 * Created on first use, destroyed when the process exits. */
#define CRH_BVH_POOL 4
static pthread_mutex_t g_bvh_lock = PTHREAD_MUTEX_INITIALIZER;
static pthread_cond_t g_bvh_free = PTHREAD_COND_INITIALIZER;
static struct { crh_ctx *ctx; int busy; } g_bvh_pool[CRH_BVH_POOL];
static int g_bvh_exit_hooked;

static void bvhPoolDestroy(void) {
	pthread_mutex_lock(&g_bvh_lock);
	for (int i = 0; i < CRH_BVH_POOL; ++i)
		if (g_bvh_pool[i].ctx && !g_bvh_pool[i].busy) { crh_context_destroy(g_bvh_pool[i].ctx); g_bvh_pool[i].ctx = NULL; }
	pthread_mutex_unlock(&g_bvh_lock);
}

/* a free builder context (index in *slot), created if the pool has room; NULL + rc if the device cannot give one and none exists to wait for */
static crh_ctx *bvhPoolAcquire(int *slot, int *rc) {
	pthread_mutex_lock(&g_bvh_lock);
	if (!g_bvh_exit_hooked) { g_bvh_exit_hooked = 1; atexit(bvhPoolDestroy); }
	for (;;) {
		int empty = -1, existing = 0;
		for (int i = 0; i < CRH_BVH_POOL; ++i) {
			if (g_bvh_pool[i].ctx && !g_bvh_pool[i].busy) { g_bvh_pool[i].busy = 1; *slot = i; pthread_mutex_unlock(&g_bvh_lock); return g_bvh_pool[i].ctx; }
			if (g_bvh_pool[i].ctx) ++existing;
			else if (empty < 0) empty = i;
		}
		if (empty >= 0) {
			crh_ctx *c = NULL;
			*rc = crh_context_create(0, NULL, &c);
			if (*rc == CRH_OK) { g_bvh_pool[empty].ctx = c; g_bvh_pool[empty].busy = 1; *slot = empty; pthread_mutex_unlock(&g_bvh_lock); return c; }
			if (!existing) { pthread_mutex_unlock(&g_bvh_lock); return NULL; }
		}
		pthread_cond_wait(&g_bvh_free, &g_bvh_lock);
	}
}
static void bvhPoolRelease(int slot) {
	pthread_mutex_lock(&g_bvh_lock);
	g_bvh_pool[slot].busy = 0;
	pthread_cond_signal(&g_bvh_free);
	pthread_mutex_unlock(&g_bvh_lock);
}

/* malloc for the arrays the builder fills: from 8 MB on, on a 2 MB boundary with transparent huge pages asked for (free()- and realloc()-compatible, which is what
 * the reference's destroyBvh and bvh.c:283 do to them). A 10 M-triangle mesh is 240 MB of nodes written by a device-to-host copy into fresh memory: with 4 KB pages
 * that is 58 000 page faults in a process that has the GPU open (round 4: the same finding as for the scene upload). */
static void *bigMalloc(size_t bytes) {
	if (bytes < ((size_t)8 << 20) || getenv("CRH_NO_HUGEPAGES")) return malloc(bytes);          /* (the variable: dev, for an A/B) */
	void *p = NULL;
	const size_t rounded = (bytes + ((size_t)2 << 20) - 1) & ~(((size_t)2 << 20) - 1);
	if (posix_memalign(&p, (size_t)2 << 20, rounded) != 0) return malloc(bytes);
	(void)madvise(p, rounded, MADV_HUGEPAGE);
	return p;
}

struct bvh *buildBottomLevelBvh(struct poly *polys, unsigned count) {
	struct bvh *bvh = malloc(sizeof(*bvh));
	bvh->nodeCount = 0; bvh->nodes = NULL; bvh->primIndices = NULL;
	if (count < 1) return bvh;                                  /* bvh.c:250-256 */
	struct timeval tTrace;
	startTimer(&tTrace);
	bvh->nodes = bigMalloc(sizeof(struct bvhNode) * (2 * (size_t)count - 1));
	bvh->primIndices = bigMalloc(sizeof(int) * count);
	/* Upload only the vertex range this mesh references (the loader gives every mesh a contiguous range of g_vertices,
	 * wavefront.c:110-126), not the whole global buffer once per mesh: polygons are handed over with indices rebased to it. */
	int vmin = polys[0].vertexIndex[0], vmax = vmin;
	for (unsigned i = 0; i < count; ++i)
		for (int k = 0; k < 3; ++k) {
			const int v = polys[i].vertexIndex[k];
			if (v < vmin) vmin = v;
			if (v > vmax) vmax = v;
		}
	if (vmin < 0 || vmax >= vertexCount) logr(error, "c-ray-hip: mesh references vertex %i outside g_vertices[0..%i)\n", vmin < 0 ? vmin : vmax, vertexCount);
	crh_poly *local = bigMalloc(sizeof(*local) * count);
	memcpy(local, polys, sizeof(*local) * count);
	for (unsigned i = 0; i < count; ++i)
		for (int k = 0; k < 3; ++k) local[i].v[k] -= vmin;
	const long preparedUs = getUs(tTrace);
	int rc = CRH_OK, slot = -1;
	crh_ctx *ctx = bvhPoolAcquire(&slot, &rc);
	const long acquiredUs = getUs(tTrace);
	crh_bvh_build_stats stats;
	memset(&stats, 0, sizeof(stats));
	if (ctx) {
		rc = crh_bvh_build_triangles(ctx, local, count, (const float *)(g_vertices + vmin), (uint64_t)(vmax - vmin + 1),
		                             (crh_bvh_node *)bvh->nodes, bvh->primIndices, &bvh->nodeCount, &stats);
		bvhPoolRelease(slot);
	}
	free(local);
	if (rc != CRH_OK) logr(error, "c-ray-hip: GPU BVH build failed (%i): %s\n", rc, crh_last_error());   /* exits: no CPU path here */
	const long builtUs = getUs(tTrace);
	bvh->nodes = realloc(bvh->nodes, sizeof(struct bvhNode) * bvh->nodeCount);   /* bvh.c:283 */
	if (getenv("CRH_TRACE_UPLOAD") && count > 10000)
		fprintf(stderr, "buildBottomLevelBvh trace: %u triangles: index range + rebased copy %.1f ms, builder context after %.1f ms, GPU builder call %.1f ms (upload %.1f, build %.1f, download %.1f), shrink %.1f ms\n",
				count, preparedUs / 1e3, (acquiredUs - preparedUs) / 1e3, (builtUs - acquiredUs) / 1e3, stats.upload_ms, stats.build_ms, stats.download_ms, (getUs(tTrace) - builtUs) / 1e3);
	return bvh;
}
#endif

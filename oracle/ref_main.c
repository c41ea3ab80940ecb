/*
 * ref_main.c — entry point of the oracle/_ref binaries (c-ray-ref, c-ray-ref-strict, c-ray-ref-count).
 *
 * TEST INFRASTRUCTURE, not product. It replaces only src/main.c of the reference (same call sequence,
 * src/main.c:14-42) and adds what the reference cannot do on its own:
 *   CRH_DUMP_F32=<path>    after the render, write state.renderBuffer (linear float RGB, the
 *                          reference's y-flipped layout, src/datatypes/image/texture.c:24-28) as raw f32
 *   CRH_NODE_PATCH=<name>  after the scene is loaded, rebuild some materials / instances from the reference's own C node
 *                          constructors (oracle/ref_node_patch.c): the nodes no JSON path reaches
 *   CRH_DUMP_STATS=<path>  write one JSON line: render wall ms, threads, W, H, spp, bounces, and —
 *                          in the -count flavour — rays / node tests / triangle tests
 * Every other translation unit of these binaries is the UNMODIFIED reference source compiled where
 * it lies under /root/reference/src (see oracle/Makefile).
 */
#include <stdlib.h>
#include <stdbool.h>
#include <stdio.h>
#include <stdint.h>
#include <signal.h>

#include "c-ray.h"
#include "renderer/renderer.h"
#include "datatypes/image/texture.h"
#include "utils/timer.h"

extern struct renderer *g_renderer;
void crh_apply_node_patch(struct renderer *r);      /* oracle/ref_node_patch.c */

/* Defined by the -count flavour's wrapper TUs (ref_count_*.c); absent (NULL) otherwise. */
extern uint64_t crh_count_rays __attribute__((weak));
extern uint64_t crh_count_node_tests __attribute__((weak));
extern uint64_t crh_count_tri_tests __attribute__((weak));

static void dumpFloatBuffer(const char *path) {
	const struct texture *t = g_renderer->state.renderBuffer;
	FILE *f = fopen(path, "wb");
	if (!f) { crLog("CRH_DUMP_F32: cannot open %s\n", path); return; }
	fwrite(t->data.float_p, sizeof(float), t->width * t->height * t->channels, f);
	fclose(f);
}

static void dumpStats(const char *path, long renderMs) {
	FILE *f = fopen(path, "w");
	if (!f) return;
	fprintf(f, "{\"render_ms\": %ld, \"threads\": %d, \"width\": %u, \"height\": %u, \"samples\": %d, \"bounces\": %d",
			renderMs, g_renderer->prefs.threadCount, g_renderer->prefs.imageWidth, g_renderer->prefs.imageHeight,
			g_renderer->prefs.sampleCount, g_renderer->prefs.bounces);
	if (&crh_count_rays)
		fprintf(f, ", \"rays\": %llu, \"node_tests\": %llu, \"tri_tests\": %llu",
				(unsigned long long)crh_count_rays, (unsigned long long)crh_count_node_tests, (unsigned long long)crh_count_tri_tests);
	fprintf(f, "}\n");
	fclose(f);
}

int main(int argc, char *argv[]) {
	crLog("C-ray v%s [%.8s] (oracle/_ref build)\n", crGetVersion(), crGitHash());
	crInitialize();
	crParseArgs(argc, argv);
	/* cluster master: at the end of a render the reference answers the worker's goodbye on a socket the worker may already have closed
	 * (server.c:213-216 vs worker.c:409-413); the worker ignores SIGPIPE (worker.c:349), the master does not and dies of it now and then.
	 * The harness ignores it so that cluster tests are deterministic. */
	signal(SIGPIPE, SIG_IGN);
	crInitRenderer();
	if (crOptionIsSet("is_worker")) {          /* src/main.c:19,33-36: `--worker [port]` = the reference's own (CPU) cluster worker */
		crStartRenderWorker();
		crDestroyRenderer();
		crDestroyOptions();
		return 0;
	}
	size_t bytes = 0;
	char *input = crOptionIsSet("inputFile") ? crReadFile(&bytes) : crReadStdin(&bytes);
	if (!input) {
		crLog("No input provided, exiting.\n");
		crDestroyRenderer();
		crDestroyOptions();
		return -1;
	}
	if (crLoadSceneFromBuf(input) != 0) return -1;
	free(input);
	crh_apply_node_patch(g_renderer);

	crStartRenderer();
	long renderMs = (long)getMs(*g_renderer->state.timer);
	if (getenv("CRH_DUMP_F32")) dumpFloatBuffer(getenv("CRH_DUMP_F32"));
	if (getenv("CRH_DUMP_STATS")) dumpStats(getenv("CRH_DUMP_STATS"), renderMs);
	if (!getenv("CRH_NO_IMAGE")) crWriteImage();

	crDestroyRenderer();
	crDestroyOptions();
	return 0;
}

/*
 * share.h — which pixels GPU g of G owns when a frame is rendered by several GPUs of one node (C host and, restated, render.py).
 *
 * Horizontal strips of CRH_STRIP_ROWS pixel rows, strip i owned by GPU i mod G: every GPU samples the whole image, so the shares are
 * balanced whatever the scene (dealing out the reference's ordered tile list, tile.c:66-117, is not: DESIGN.md section 6). Any
 * disjoint cover reproduces the frame bit for bit — a pixel's passes fold on its owner, non-owned pixels stay exactly 0.0f and
 * the RCCL reduce(SUM) is a gather.
 */
#ifndef CRH_SHARE_H
#define CRH_SHARE_H
#include <stdint.h>
#include "cray_hip.h"

#define CRH_STRIP_ROWS 4

/* upper bound of the strips one GPU gets */
static inline uint32_t crh_strip_share_max(int height, int gpus) { return (uint32_t)((height + CRH_STRIP_ROWS - 1) / CRH_STRIP_ROWS / (gpus > 0 ? gpus : 1) + 1); }

/* writes GPU g's strips into out[] (room for crh_strip_share_max entries), returns how many */
static inline uint32_t crh_strip_share(int width, int height, int g, int gpus, crh_tile *out) {
	uint32_t n = 0;
	for (int y = 0, i = 0; y < height; y += CRH_STRIP_ROWS, ++i)
		if (i % gpus == g) out[n++] = (crh_tile){0, y, width, y + CRH_STRIP_ROWS < height ? y + CRH_STRIP_ROWS : height};
	return n;
}
#endif

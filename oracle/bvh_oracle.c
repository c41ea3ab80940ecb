/*
 * bvh_oracle.c — TEST INFRASTRUCTURE: CPU restatement of the reference's binned-SAH BVH builder
 * (src/accelerators/bvh.c:87-316, VKoskiv/c-ray v0.6.3) for SURVEY.md §8(f) row 1 (GPU BVH build).
 *
 * Only tests/ may use this file. It is pinned against the real thing: every scene blob holds the node array and
 * the primitive order produced by the reference's own builder (compiled from the reference sources into
 * oracle/_ref/crh-flatten with -ffp-contract=off), and tests/test_bvh_build.py requires this restatement to
 * reproduce them byte for byte (nodes) and index for index (prim order) for every mesh of every fixture.
 *
 * Written as an explicit work stack over index ranges instead of the reference's recursion; the arithmetic, the
 * comparison directions and the order in which boxes are folded are the reference's, line by line:
 *   bin index            bvh.c:87-93   (float -> unsigned conversion as gcc/x86-64 does it: 64-bit cvttss2si, low word)
 *   bin fill             bvh.c:158-165 (extendBBox = vecMin/vecMax with includes.h:20-21 min/max: on a tie the NEW
 *                                        operand wins, which only shows for -0 / +0)
 *   right-to-left sweep  bvh.c:170-177, left-to-right sweep bvh.c:180-191 (strict <, first minimum wins)
 *   axis choice          bvh.c:195-197, leaf cost bvh.c:200, median fallback bvh.c:202-211
 *   partition            bvh.c:95-130  (two-pointer swap)
 *   children             bvh.c:219-238 (pair allocated when the parent splits, left subtree numbered before the right)
 *   triangle bounds      bvh.c:289-297 (getMidPoint: vector.h:186-188)
 * Build flags: the same as the pinned reference flavour (-O2 -march=x86-64-v3 -ffp-contract=off).
 */
#include <float.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "cray_hip.h"

#define ORC_BINS 32
#define ORC_MAX_DEPTH 64u
#define ORC_MAX_LEAF 16u
#define ORC_TRAVERSAL_COST 1.5f

typedef struct { float lo[3], hi[3]; } box3;
typedef struct { box3 box; unsigned n; float cost; } sah_bin;

static inline float pick_lo(float a, float b) { return a < b ? a : b; }   /* includes.h:20 */
static inline float pick_hi(float a, float b) { return a > b ? a : b; }   /* includes.h:21 */

static inline void box_reset(box3 *b) {
	for (int k = 0; k < 3; ++k) { b->lo[k] = FLT_MAX; b->hi[k] = -FLT_MAX; }           /* bbox.h:20-23 */
}
static inline void box_grow(box3 *dst, const box3 *src) {                                /* bbox.h:30-33 */
	for (int k = 0; k < 3; ++k) { dst->lo[k] = pick_lo(dst->lo[k], src->lo[k]); dst->hi[k] = pick_hi(dst->hi[k], src->hi[k]); }
}
static inline float box_half_area(const box3 *b) {                                       /* bbox.h:25-28 */
	const float ex = b->hi[0] - b->lo[0], ey = b->hi[1] - b->lo[1], ez = b->hi[2] - b->lo[2];
	return ex * (ey + ez) + ey * ez;
}
static inline unsigned bin_of(float coord, float lo, float hi) {                         /* bvh.c:87-93 */
	const float scale = ORC_BINS / (hi - lo);
	const float f = (coord - lo) * scale;
	const unsigned b = f < 0 ? 0 : (unsigned)f;
	return b >= ORC_BINS ? ORC_BINS - 1 : b;
}

/* bvh.c:289-297: bounds and centre of every triangle */
void orc_bvh_triangle_bounds(const crh_poly *polys, const float *vertices, uint32_t count, float *boxes6, float *centers3) {
	for (uint32_t i = 0; i < count; ++i) {
		const float *a = vertices + 3 * (size_t)polys[i].v[0], *b = vertices + 3 * (size_t)polys[i].v[1], *c = vertices + 3 * (size_t)polys[i].v[2];
		for (int k = 0; k < 3; ++k) {
			centers3[3 * (size_t)i + k] = ((a[k] + b[k]) + c[k]) * (1.0f / 3.0f);
			boxes6[6 * (size_t)i + k] = pick_lo(a[k], pick_lo(b[k], c[k]));
			boxes6[6 * (size_t)i + 3 + k] = pick_hi(a[k], pick_hi(b[k], c[k]));
		}
	}
}

typedef struct { uint32_t node, first, last, depth; } job;

/* boxes6: {min xyz, max xyz} per primitive; nodes: capacity 2*count-1; prims: capacity count. Returns 0, -1 (no memory) or -2 (the reference's
 * own node array would overflow on this input: no reference result exists). */
int orc_bvh_build(const float *boxes6, const float *centers3, uint32_t count, crh_bvh_node *nodes, int32_t *prims, uint32_t *node_count) {
	if (count < 1) { *node_count = 0; return 0; }                                           /* bvh.c:250-256 */
	box3 root;
	box_reset(&root);
	for (uint32_t i = 0; i < count; ++i) {                                                  /* bvh.c:264-269 */
		box3 b;
		memcpy(b.lo, boxes6 + 6 * (size_t)i, 12); memcpy(b.hi, boxes6 + 6 * (size_t)i + 3, 12);
		prims[i] = (int32_t)i;
		box_grow(&root, &b);
	}
	memset(nodes, 0, sizeof(*nodes) * (2 * (size_t)count - 1));
	for (int k = 0; k < 3; ++k) { nodes[0].bounds[2 * k] = root.lo[k]; nodes[0].bounds[2 * k + 1] = root.hi[k]; }
	uint32_t used = 1;

	/* depth-first, left before right: the right sibling waits on the stack while the left subtree is numbered */
	job *stack = malloc(sizeof(job) * (2 * ORC_MAX_DEPTH + 4));
	if (!stack) return -1;
	size_t sp = 0;
	stack[sp++] = (job){0, 0, count, 0};
	sah_bin (*bins)[ORC_BINS] = malloc(sizeof(sah_bin) * 3 * ORC_BINS);
	if (!bins) { free(stack); return -1; }
	while (sp) {
		const job j = stack[--sp];
		crh_bvh_node *nd = &nodes[j.node];
		const uint32_t n = j.last - j.first;
#define LEAF() do { nd->first = j.first; nd->count_leaf = (n & 0x3FFFFFFFu) | (1u << 30); } while (0)
		if (j.depth >= ORC_MAX_DEPTH || n < 2) { LEAF(); continue; }                         /* bvh.c:143-146 */

		float best_cost[3] = {FLT_MAX, FLT_MAX, FLT_MAX};
		unsigned best_bin[3] = {1, 1, 1};
		for (int ax = 0; ax < 3; ++ax) {
			const float lo = nd->bounds[2 * ax], hi = nd->bounds[2 * ax + 1];
			for (int b = 0; b < ORC_BINS; ++b) { box_reset(&bins[ax][b].box); bins[ax][b].n = 0; }
			for (uint32_t p = j.first; p < j.last; ++p) {                                   /* bvh.c:158-165 */
				const int32_t id = prims[p];
				sah_bin *bn = &bins[ax][bin_of(centers3[3 * (size_t)id + ax], lo, hi)];
				box3 b;
				memcpy(b.lo, boxes6 + 6 * (size_t)id, 12); memcpy(b.hi, boxes6 + 6 * (size_t)id + 3, 12);
				box_grow(&bn->box, &b);
				bn->n++;
			}
			box3 acc;
			unsigned cnt = 0;
			box_reset(&acc);
			for (unsigned b = ORC_BINS; b > 1; --b) {                                       /* bvh.c:170-177 */
				sah_bin *bn = &bins[ax][b - 1];
				cnt += bn->n;
				box_grow(&acc, &bn->box);
				bn->cost = cnt * box_half_area(&acc);
			}
			box_reset(&acc);
			cnt = 0;
			for (unsigned b = 0; b < ORC_BINS - 1; ++b) {                                   /* bvh.c:180-191 */
				sah_bin *bn = &bins[ax][b];
				cnt += bn->n;
				box_grow(&acc, &bn->box);
				const float cost = cnt * box_half_area(&acc) + bins[ax][b + 1].cost;
				if (cost < best_cost[ax]) { best_bin[ax] = b + 1; best_cost[ax] = cost; }
			}
		}
		unsigned ax = 0;                                                                    /* bvh.c:195-197 */
		if (best_cost[1] < best_cost[0]) ax = 1;
		if (best_cost[2] < best_cost[ax]) ax = 2;

		box3 self;
		for (int k = 0; k < 3; ++k) { self.lo[k] = nd->bounds[2 * k]; self.hi[k] = nd->bounds[2 * k + 1]; }
		const float leaf_cost = box_half_area(&self) * (n - ORC_TRAVERSAL_COST);            /* bvh.c:200 */
		if (best_cost[ax] > leaf_cost) {
			if (n > ORC_MAX_LEAF) {                                                         /* bvh.c:202-211: bin boundary nearest the median */
				unsigned seen = 0, closest = n;
				for (unsigned b = 0; b < ORC_BINS - 1; ++b) {
					seen += bins[ax][b].n;
					const unsigned off = (unsigned)abs((int)n / 2 - (int)seen);
					if (off < closest) { closest = off; best_bin[ax] = b + 1; }
				}
			} else { LEAF(); continue; }
		}
		const unsigned split = best_bin[ax];
		const float lo = nd->bounds[2 * ax], hi = nd->bounds[2 * ax + 1];
		uint32_t l = j.first, r = j.last;                                                   /* bvh.c:95-130 */
		while (l < r) {
			while (l < r && bin_of(centers3[3 * (size_t)prims[l] + ax], lo, hi) < split) ++l;
			while (l < r && bin_of(centers3[3 * (size_t)prims[r - 1] + ax], lo, hi) >= split) --r;
			if (l >= r) break;
			const int32_t t = prims[r - 1]; prims[r - 1] = prims[l]; prims[l] = t;
			--r; ++l;
		}
		if (l <= j.first) { LEAF(); continue; }                                             /* bvh.c:239-241 */
		/* The reference allocates 2 * count - 1 nodes (bvh.c:271: "binary tree property") and then splits nodes whose primitives all land on
		 * the left (bvh.c:220 only asks beginRight > begin): more than 16 coincident primitives give a chain of (everything | nothing)
		 * splits down to the depth limit, and enough such clusters overflow its heap array. There is no reference result then. */
		if ((size_t)used + 2 > 2 * (size_t)count - 1) { free(bins); free(stack); return -2; }
		const uint32_t kids = used;                                                         /* bvh.c:221-223 */
		used += 2;
		box3 lb, rb;                                                                        /* bvh.c:226-233 */
		box_reset(&lb); box_reset(&rb);
		for (unsigned b = 0; b < split; ++b) box_grow(&lb, &bins[ax][b].box);
		for (unsigned b = split; b < ORC_BINS; ++b) box_grow(&rb, &bins[ax][b].box);
		for (int k = 0; k < 3; ++k) {
			nodes[kids].bounds[2 * k] = lb.lo[k]; nodes[kids].bounds[2 * k + 1] = lb.hi[k];
			nodes[kids + 1].bounds[2 * k] = rb.lo[k]; nodes[kids + 1].bounds[2 * k + 1] = rb.hi[k];
		}
		nd->first = kids;
		nd->count_leaf = 0;
		stack[sp++] = (job){kids + 1, l, j.last, j.depth + 1};                              /* right waits ... */
		stack[sp++] = (job){kids, j.first, l, j.depth + 1};                                 /* ... left is built first (bvh.c:237-238) */
#undef LEAF
	}
	free(bins);
	free(stack);
	*node_count = used;
	return 0;
}

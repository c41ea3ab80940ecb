/* -count flavour only: compiles the UNMODIFIED reference src/accelerators/bvh.c with
 *   - fmaf() (used only by fastMultiplyAdd, bvh.c:318-324: six per intersectNode) counted, and
 *   - rayIntersectsWithPolygon (bvh.c:455) routed through a counting trampoline.
 * node tests = fma count / 6. Requires FP_FAST_FMAF (true for -march=x86-64-v3). */
#include <stdint.h>
#include <stdbool.h>
#include <math.h>
#ifndef FP_FAST_FMAF
#error "the -count flavour needs FP_FAST_FMAF so that fastMultiplyAdd calls fmaf"
#endif
uint64_t crh_count_node_tests = 0;
uint64_t crh_count_tri_tests = 0;
static uint64_t crh_count_fma = 0;

static inline float crh_counted_fmaf(float a, float b, float c) {
	uint64_t n = __atomic_add_fetch(&crh_count_fma, 1, __ATOMIC_RELAXED);
	if (n % 6 == 0) __atomic_fetch_add(&crh_count_node_tests, 1, __ATOMIC_RELAXED);
	return __builtin_fmaf(a, b, c);
}
#define fmaf(a, b, c) crh_counted_fmaf(a, b, c)
#define rayIntersectsWithPolygon crh_counted_rayIntersectsWithPolygon
#include "accelerators/bvh.c"
#undef rayIntersectsWithPolygon
#undef fmaf

bool rayIntersectsWithPolygon(const struct lightRay *ray, const struct poly *poly, struct hitRecord *isect);

bool crh_counted_rayIntersectsWithPolygon(const struct lightRay *ray, const struct poly *poly, struct hitRecord *isect) {
	__atomic_fetch_add(&crh_count_tri_tests, 1, __ATOMIC_RELAXED);
	return rayIntersectsWithPolygon(ray, poly, isect);
}

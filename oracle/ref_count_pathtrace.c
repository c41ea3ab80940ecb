/* -count flavour only: compiles the UNMODIFIED reference src/renderer/pathtrace.c with its one call
 * to traverseTopLevelBvh (pathtrace.c:28) routed through a counting trampoline, so that
 * crh_count_rays == number of getClosestIsect calls == primary + secondary rays. */
#include <stdint.h>
#include <stdbool.h>
uint64_t crh_count_rays = 0;

#define traverseTopLevelBvh crh_counted_traverseTopLevelBvh
#include "renderer/pathtrace.c"
#undef traverseTopLevelBvh

bool traverseTopLevelBvh(const struct instance *instances, const struct bvh *bvh, const struct lightRay *ray, struct hitRecord *isect, sampler *sampler);

bool crh_counted_traverseTopLevelBvh(const struct instance *instances, const struct bvh *bvh, const struct lightRay *ray, struct hitRecord *isect, sampler *sampler) {
	__atomic_fetch_add(&crh_count_rays, 1, __ATOMIC_RELAXED);
	return traverseTopLevelBvh(instances, bvh, ray, isect, sampler);
}

/* Wrapper TU: compiles the UNMODIFIED reference file nodes/shaders/emission.c and appends a describer (see describe.h). */
#include "nodes/shaders/emission.c"
#include "describe.h"

bool crh_describe_emission(const void *node, struct crh_node_desc *d) {
	const struct bsdfNode *base = node;
	if (base->sample != sample) return false;
	const struct emissiveBsdf *t = node;
	(void)t;
	d->kind = CRH_BSDF_EMISSION;
	d->child[0] = t->color; d->cls[0] = CRH_CLS_COLOR;
	d->child[1] = t->strength; d->cls[1] = CRH_CLS_VALUE;
	return true;
}

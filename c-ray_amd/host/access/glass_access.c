/* Wrapper TU: compiles the UNMODIFIED reference file nodes/shaders/glass.c and appends a describer (see describe.h). */
#include "nodes/shaders/glass.c"
#include "describe.h"

bool crh_describe_glass(const void *node, struct crh_node_desc *d) {
	const struct bsdfNode *base = node;
	if (base->sample != sample) return false;
	const struct glassBsdf *t = node;
	(void)t;
	d->kind = CRH_BSDF_GLASS;
	d->child[0] = t->color; d->cls[0] = CRH_CLS_COLOR;
	d->child[1] = t->roughness; d->cls[1] = CRH_CLS_VALUE;
	d->child[2] = t->IOR; d->cls[2] = CRH_CLS_VALUE;
	return true;
}

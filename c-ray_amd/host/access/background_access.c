/* Wrapper TU: compiles the UNMODIFIED reference file nodes/shaders/background.c and appends a describer (see describe.h). */
#include "nodes/shaders/background.c"
#include "describe.h"

bool crh_describe_background(const void *node, struct crh_node_desc *d) {
	const struct bsdfNode *base = node;
	if (base->sample != sample) return false;
	const struct backgroundBsdf *t = node;
	(void)t;
	d->kind = CRH_BSDF_BACKGROUND;
	d->child[0] = t->color; d->cls[0] = CRH_CLS_COLOR;
	d->child[1] = t->strength; d->cls[1] = CRH_CLS_VALUE;
	d->child[2] = t->offset; d->cls[2] = CRH_CLS_VALUE;
	return true;
}

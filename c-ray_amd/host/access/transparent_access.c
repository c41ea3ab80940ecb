/* Wrapper TU: compiles the UNMODIFIED reference file nodes/shaders/transparent.c and appends a describer (see describe.h). */
#include "nodes/shaders/transparent.c"
#include "describe.h"

bool crh_describe_transparent(const void *node, struct crh_node_desc *d) {
	const struct bsdfNode *base = node;
	if (base->sample != sample) return false;
	const struct transparent *t = node;
	(void)t;
	d->kind = CRH_BSDF_TRANSPARENT;
	d->child[0] = t->color; d->cls[0] = CRH_CLS_COLOR;
	return true;
}

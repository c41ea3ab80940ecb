/* Wrapper TU: compiles the UNMODIFIED reference file nodes/shaders/diffuse.c and appends a describer (see describe.h). */
#include "nodes/shaders/diffuse.c"
#include "describe.h"

bool crh_describe_diffuse(const void *node, struct crh_node_desc *d) {
	const struct bsdfNode *base = node;
	if (base->sample != sample) return false;
	const struct diffuseBsdf *t = node;
	(void)t;
	d->kind = CRH_BSDF_DIFFUSE;
	d->child[0] = t->color; d->cls[0] = CRH_CLS_COLOR;
	return true;
}

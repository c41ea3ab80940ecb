/* Wrapper TU: compiles the UNMODIFIED reference file nodes/shaders/plastic.c and appends a describer (see describe.h). */
#include "nodes/shaders/plastic.c"
#include "describe.h"

bool crh_describe_plastic(const void *node, struct crh_node_desc *d) {
	const struct bsdfNode *base = node;
	if (base->sample != sample) return false;
	const struct plasticBsdf *t = node;
	(void)t;
	d->kind = CRH_BSDF_PLASTIC;
	d->child[0] = t->color; d->cls[0] = CRH_CLS_COLOR;
	d->child[1] = t->roughness; d->cls[1] = CRH_CLS_COLOR;
	d->child[2] = t->diffuse; d->cls[2] = CRH_CLS_BSDF;
	return true;
}

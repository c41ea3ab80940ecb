/* Wrapper TU: compiles the UNMODIFIED reference file nodes/shaders/add.c and appends a describer (see describe.h). */
#include "nodes/shaders/add.c"
#include "describe.h"

bool crh_describe_add(const void *node, struct crh_node_desc *d) {
	const struct bsdfNode *base = node;
	if (base->sample != sample) return false;
	const struct addBsdf *t = node;
	(void)t;
	d->kind = CRH_BSDF_ADD;
	d->child[0] = t->A; d->cls[0] = CRH_CLS_BSDF;
	d->child[1] = t->B; d->cls[1] = CRH_CLS_BSDF;
	return true;
}

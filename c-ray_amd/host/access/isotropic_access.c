/* Wrapper TU: compiles the UNMODIFIED reference file nodes/shaders/isotropic.c and appends a describer (see describe.h). */
#include "nodes/shaders/isotropic.c"
#include "describe.h"

bool crh_describe_isotropic(const void *node, struct crh_node_desc *d) {
	const struct bsdfNode *base = node;
	if (base->sample != sample) return false;
	const struct isotropicBsdf *t = node;
	(void)t;
	d->kind = CRH_BSDF_ISOTROPIC;
	d->child[0] = t->color; d->cls[0] = CRH_CLS_COLOR;
	return true;
}

/* Wrapper TU: compiles the UNMODIFIED reference file nodes/input/fresnel.c and appends a describer (see describe.h). */
#include "nodes/input/fresnel.c"
#include "describe.h"

bool crh_describe_fresnel(const void *node, struct crh_node_desc *d) {
	const struct valueNode *base = node;
	if (base->eval != eval) return false;
	const struct fresnelNode *t = node;
	(void)t;
	d->kind = CRH_VALUE_FRESNEL;
	d->child[0] = t->IOR; d->cls[0] = CRH_CLS_VALUE;
	d->child[1] = t->normal; d->cls[1] = CRH_CLS_VECTOR;
	return true;
}

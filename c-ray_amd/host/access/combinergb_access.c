/* Wrapper TU: compiles the UNMODIFIED reference file nodes/converter/combinergb.c and appends a describer (see describe.h). */
#include "nodes/converter/combinergb.c"
#include "describe.h"

bool crh_describe_combinergb(const void *node, struct crh_node_desc *d) {
	const struct colorNode *base = node;
	if (base->eval != eval) return false;
	const struct combineRGB *t = node;
	(void)t;
	d->kind = CRH_COLOR_COMBINERGB;
	d->child[0] = t->R; d->cls[0] = CRH_CLS_VALUE;
	d->child[1] = t->G; d->cls[1] = CRH_CLS_VALUE;
	d->child[2] = t->B; d->cls[2] = CRH_CLS_VALUE;
	return true;
}

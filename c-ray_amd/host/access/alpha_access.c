/* Wrapper TU: compiles the UNMODIFIED reference file nodes/textures/alpha.c and appends a describer (see describe.h). */
#include "nodes/textures/alpha.c"
#include "describe.h"

bool crh_describe_alpha(const void *node, struct crh_node_desc *d) {
	const struct valueNode *base = node;
	if (base->eval != eval) return false;
	const struct alphaNode *t = node;
	(void)t;
	d->kind = CRH_VALUE_ALPHA;
	d->child[0] = t->color; d->cls[0] = CRH_CLS_COLOR;
	return true;
}

/* Wrapper TU: compiles the UNMODIFIED reference file nodes/textures/checker.c and appends a describer (see describe.h). */
#include "nodes/textures/checker.c"
#include "describe.h"

bool crh_describe_checker(const void *node, struct crh_node_desc *d) {
	const struct colorNode *base = node;
	if (base->eval != eval) return false;
	const struct checkerTexture *t = node;
	(void)t;
	d->kind = CRH_COLOR_CHECKER;
	d->child[0] = t->A; d->cls[0] = CRH_CLS_COLOR;
	d->child[1] = t->B; d->cls[1] = CRH_CLS_COLOR;
	d->child[2] = t->scale; d->cls[2] = CRH_CLS_VALUE;
	return true;
}

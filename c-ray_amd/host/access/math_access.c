/* Wrapper TU: compiles the UNMODIFIED reference file nodes/converter/math.c and appends a describer (see describe.h). */
#include "nodes/converter/math.c"
#include "describe.h"

bool crh_describe_math(const void *node, struct crh_node_desc *d) {
	const struct valueNode *base = node;
	if (base->eval != eval) return false;
	const struct mathNode *t = node;
	(void)t;
	d->kind = CRH_VALUE_MATH;
	d->child[0] = t->A; d->cls[0] = CRH_CLS_VALUE;
	d->child[1] = t->B; d->cls[1] = CRH_CLS_VALUE;
	d->u = (uint32_t)t->op;
	return true;
}

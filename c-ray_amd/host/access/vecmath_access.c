/* Wrapper TU: compiles the UNMODIFIED reference file nodes/converter/vecmath.c and appends a describer (see describe.h). */
#include "nodes/converter/vecmath.c"
#include "describe.h"

bool crh_describe_vecmath(const void *node, struct crh_node_desc *d) {
	const struct vectorNode *base = node;
	if (base->eval != eval) return false;
	const struct vecMathNode *t = node;
	(void)t;
	d->kind = CRH_VEC_VECMATH;
	d->child[0] = t->A; d->cls[0] = CRH_CLS_VECTOR;
	d->child[1] = t->B; d->cls[1] = CRH_CLS_VECTOR;
	d->u = (uint32_t)t->op;
	return true;
}

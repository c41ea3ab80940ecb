/* Wrapper TU: compiles the UNMODIFIED reference file nodes/converter/combine.c and appends a describer (see describe.h). */
#include "nodes/converter/combine.c"
#include "describe.h"

bool crh_describe_combine(const void *node, struct crh_node_desc *d) {
	const struct colorNode *base = node;
	if (base->eval != eval) return false;
	const struct combineValue *t = node;
	(void)t;
	d->kind = CRH_COLOR_COMBINE;
	d->child[0] = t->input; d->cls[0] = CRH_CLS_VALUE;
	return true;
}

/* Wrapper TU: compiles the UNMODIFIED reference file nodes/converter/grayscale.c and appends a describer (see describe.h). */
#include "nodes/converter/grayscale.c"
#include "describe.h"

bool crh_describe_grayscale(const void *node, struct crh_node_desc *d) {
	const struct valueNode *base = node;
	if (base->eval != eval) return false;
	const struct grayscale *t = node;
	(void)t;
	d->kind = CRH_VALUE_GRAYSCALE;
	d->child[0] = t->input; d->cls[0] = CRH_CLS_COLOR;
	return true;
}

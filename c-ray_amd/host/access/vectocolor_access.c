/* Wrapper TU: compiles the UNMODIFIED reference file nodes/converter/vectocolor.c and appends a describer (see describe.h). */
#include "nodes/converter/vectocolor.c"
#include "describe.h"

bool crh_describe_vectocolor(const void *node, struct crh_node_desc *d) {
	const struct colorNode *base = node;
	if (base->eval != eval) return false;
	const struct vecToColorNode *t = node;
	(void)t;
	d->kind = CRH_COLOR_VECTOCOLOR;
	d->child[0] = t->vec; d->cls[0] = CRH_CLS_VECTOR;
	return true;
}

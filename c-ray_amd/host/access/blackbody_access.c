/* Wrapper TU: compiles the UNMODIFIED reference file nodes/converter/blackbody.c and appends a describer (see describe.h). */
#include "nodes/converter/blackbody.c"
#include "describe.h"

bool crh_describe_blackbody(const void *node, struct crh_node_desc *d) {
	const struct colorNode *base = node;
	if (base->eval != eval) return false;
	const struct blackbodyNode *t = node;
	(void)t;
	d->kind = CRH_COLOR_BLACKBODY;
	d->child[0] = t->temperature; d->cls[0] = CRH_CLS_VALUE;
	return true;
}

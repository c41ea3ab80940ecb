/* Wrapper TU: compiles the UNMODIFIED reference file nodes/input/raylength.c and appends a describer (see describe.h). */
#include "nodes/input/raylength.c"
#include "describe.h"

bool crh_describe_raylength(const void *node, struct crh_node_desc *d) {
	const struct valueNode *base = node;
	if (base->eval != eval) return false;
	const struct rayLengthNode *t = node;
	(void)t;
	d->kind = CRH_VALUE_RAYLENGTH;
	return true;
}

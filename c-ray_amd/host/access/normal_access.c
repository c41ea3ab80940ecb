/* Wrapper TU: compiles the UNMODIFIED reference file nodes/input/normal.c and appends a describer (see describe.h). */
#include "nodes/input/normal.c"
#include "describe.h"

bool crh_describe_normal(const void *node, struct crh_node_desc *d) {
	const struct vectorNode *base = node;
	if (base->eval != eval) return false;
	const struct normalNode *t = node;
	(void)t;
	d->kind = CRH_VEC_NORMAL;
	return true;
}

/* Wrapper TU: compiles the UNMODIFIED reference file nodes/valuenode.c and appends a describer (see describe.h). */
#include "nodes/valuenode.c"
#include "describe.h"

bool crh_describe_constvalue(const void *node, struct crh_node_desc *d) {
	const struct valueNode *base = node;
	if (base->eval != eval) return false;
	const struct constantValue *t = node;
	(void)t;
	d->kind = CRH_VALUE_CONSTANT;
	d->f[0] = t->value;
	return true;
}

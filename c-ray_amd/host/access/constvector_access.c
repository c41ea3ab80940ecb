/* Wrapper TU: compiles the UNMODIFIED reference file nodes/vectornode.c and appends a describer (see describe.h). */
#include "nodes/vectornode.c"
#include "describe.h"

bool crh_describe_constvector(const void *node, struct crh_node_desc *d) {
	const struct vectorNode *base = node;
	if (base->eval != eval) return false;
	const struct constantVector *t = node;
	(void)t;
	d->kind = CRH_VEC_CONSTANT;
	d->f[0] = t->vector.x; d->f[1] = t->vector.y; d->f[2] = t->vector.z;
	return true;
}

/* Wrapper TU: compiles the UNMODIFIED reference file nodes/textures/constant.c and appends a describer (see describe.h). */
#include "nodes/textures/constant.c"
#include "describe.h"

bool crh_describe_constant(const void *node, struct crh_node_desc *d) {
	const struct colorNode *base = node;
	if (base->eval != eval) return false;
	const struct constantTexture *t = node;
	(void)t;
	d->kind = CRH_COLOR_CONSTANT;
	d->f[0] = t->color.red; d->f[1] = t->color.green; d->f[2] = t->color.blue; d->f[3] = t->color.alpha;
	return true;
}

/* Wrapper TU: compiles the UNMODIFIED reference file nodes/textures/gradient.c and appends a describer (see describe.h). */
#include "nodes/textures/gradient.c"
#include "describe.h"

bool crh_describe_gradient(const void *node, struct crh_node_desc *d) {
	const struct colorNode *base = node;
	if (base->eval != eval) return false;
	const struct gradientTexture *t = node;
	(void)t;
	d->kind = CRH_COLOR_GRADIENT;
	d->f[0] = t->down.red; d->f[1] = t->down.green; d->f[2] = t->down.blue; d->f[3] = t->down.alpha;
	d->f[4] = t->up.red; d->f[5] = t->up.green; d->f[6] = t->up.blue; d->f[7] = t->up.alpha;
	return true;
}

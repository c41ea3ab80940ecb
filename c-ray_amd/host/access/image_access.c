/* Wrapper TU: compiles the UNMODIFIED reference file nodes/textures/image.c and appends a describer (see describe.h). */
#include "nodes/textures/image.c"
#include "describe.h"

bool crh_describe_image(const void *node, struct crh_node_desc *d) {
	const struct colorNode *base = node;
	if (base->eval != eval) return false;
	const struct imageTexture *t = node;
	(void)t;
	d->kind = CRH_COLOR_IMAGE;
	d->tex = t->tex; d->u = t->options;
	return true;
}

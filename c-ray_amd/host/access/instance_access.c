/* Wrapper TU: compiles the UNMODIFIED reference src/datatypes/instance.c and appends a classifier
 * for the file-private intersect callbacks (instance.c:45,62,169,187). See describe.h. */
#include "datatypes/instance.c"
#include "describe.h"

int crh_access_instance_kind(const struct instance *i) {
	if (i->intersectFn == intersectSphere) return 0;
	if (i->intersectFn == intersectMesh) return 1;
	if (i->intersectFn == intersectSphereVolume) return 2;
	if (i->intersectFn == intersectMeshVolume) return 3;
	return -1;
}

/* volumes keep their sphere / mesh behind a file-private wrapper struct (instance.c:22-30) */
const void *crh_access_instance_object(const struct instance *i, float *density) {
	*density = 0.0f;
	if (i->intersectFn == intersectSphereVolume) { const struct sphereVolume *v = i->object; *density = v->density; return v->sphere; }
	if (i->intersectFn == intersectMeshVolume) { const struct meshVolume *v = i->object; *density = v->density; return v->mesh; }
	return i->object;
}

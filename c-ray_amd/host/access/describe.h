/*
 * describe.h — accessor protocol between the wrapper translation units in this directory and
 * the flattener (../flatten.c).
 *
 * Every node struct of the reference (src/nodes/ ** / *.c) and `struct bvh` (src/accelerators/bvh.c:44-48)
 * is private to its .c file, and every node file reuses the static names sample/eval/compare/hash.
 * To read them WITHOUT editing or re-implementing the reference, each file here textually
 * #includes ONE unmodified reference .c (compiled instead of that .c) and appends a describer that
 * recognises the node by its private function pointer and reports its fields.
 * The reference's constructors, hash-consing and lossy comparators therefore stay byte-for-byte the
 * reference's, so node identities (and BVH node indices) are the reference's by construction.
 */
#pragma once
#include <stdbool.h>
#include <stdint.h>
#include "cray_hip.h"

enum crh_node_class { CRH_CLS_NONE = 0, CRH_CLS_BSDF, CRH_CLS_COLOR, CRH_CLS_VALUE, CRH_CLS_VECTOR };

struct texture;
struct crh_node_desc {
	uint32_t kind;                 /* enum crh_node_kind */
	const void *child[3];          /* child node pointers (NULL = absent) */
	int cls[3];                    /* enum crh_node_class of each child   */
	float f[8];                    /* immediates                          */
	const struct texture *tex;     /* image nodes                         */
	uint32_t u;                    /* image options / math op / vec op    */
};

/* One describer per wrapped reference file; returns false if `node` is not that file's kind. */
typedef bool (*crh_describe_fn)(const void *node, struct crh_node_desc *out);

#define CRH_DESCRIBERS_BSDF(X)  X(diffuse) X(metal) X(glass) X(plastic) X(mix) X(add) X(transparent) X(emission) X(isotropic) X(background)
#define CRH_DESCRIBERS_COLOR(X) X(constant) X(image) X(checker) X(gradient) X(blackbody) X(combine) X(combinergb) X(vectocolor)
#define CRH_DESCRIBERS_VALUE(X) X(constvalue) X(alpha) X(grayscale) X(math) X(fresnel) X(raylength)
#define CRH_DESCRIBERS_VECTOR(X) X(constvector) X(normal) X(vecmath)

#define CRH_DECL(name) bool crh_describe_##name(const void *node, struct crh_node_desc *out);
CRH_DESCRIBERS_BSDF(CRH_DECL)
CRH_DESCRIBERS_COLOR(CRH_DECL)
CRH_DESCRIBERS_VALUE(CRH_DECL)
CRH_DESCRIBERS_VECTOR(CRH_DECL)
#undef CRH_DECL

/* accelerators/bvh.c */
struct bvh;
const void *crh_access_bvh_nodes(const struct bvh *b);       /* struct bvhNode[], 32 B each */
const int  *crh_access_bvh_prims(const struct bvh *b);
unsigned    crh_access_bvh_node_count(const struct bvh *b);

/* datatypes/instance.c: 0 sphere solid, 1 mesh solid, 2 sphere volume, 3 mesh volume, -1 unknown */
struct instance;
int crh_access_instance_kind(const struct instance *i);
const void *crh_access_instance_object(const struct instance *i, float *density);   /* the sphere / mesh (also behind a volume wrapper) */

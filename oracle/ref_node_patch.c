/*
 * ref_node_patch.c — TEST INFRASTRUCTURE (linked into the oracle/_ref binaries and crh-flatten only, never into the product).
 *
 * The reference's JSON loader reaches only part of its node library (sceneloader.c:766-870: diffuse, metal, glass, plastic,
 * mix, add, transparent, emissive; constant / image / checkerboard / blackbody colours; grayscale(image) values). Everything
 * else — math (15 ops), vecMath (10 ops), fresnel, rayLength, normal, combineValue, combineRGB, vecToColor, isotropic, and
 * volume instances — exists only behind the C constructors (src/nodes/ ... /new*()). This file builds graphs from those
 * constructors inside the reference itself, after its loader ran, so that fixtures for them come out of the REAL reference:
 *
 *   CRH_NODE_PATCH=zoo      scene = tools/gen_golden.py's nodezoo.json (a grid of 60 spheres): spheres 0..49 get the graphs
 *                           below (tests/test_nodes.py holds the expected values of the known-answer ones, which restate
 *                           /root/reference/tests/test_nodes.h), the background becomes constant white. With CRH_ZOO_TAME=1
 *                           the known-answer spheres 0..37 keep their plain material (multi-bounce frames stay in [0, ~3]).
 *   CRH_NODE_PATCH=volumes  sphere / mesh instances flagged in the scene become newSphereVolume / newMeshVolume instances
 *                           with an isotropic medium (instance.c:62-92, 187-216); the TLAS is rebuilt.
 *
 * Called between crLoadSceneFromBuf() and the render / flatten step (oracle/ref_main.c, tools/flatten_main.c).
 */
#include <stdlib.h>
#include <stdio.h>
#include <string.h>
#include <math.h>

#include "includes.h"
#include "datatypes/scene.h"
#include "datatypes/sphere.h"
#include "datatypes/mesh.h"
#include "datatypes/instance.h"
#include "datatypes/image/texture.h"
#include "datatypes/color.h"
#include "accelerators/bvh.h"
#include "renderer/renderer.h"
#include "utils/loaders/textureloader.h"
#include "nodes/bsdfnode.h"
#include "nodes/valuenode.h"
#include "nodes/vectornode.h"
#include "nodes/colornode.h"
#include "nodes/converter/math.h"
#include "nodes/converter/vecmath.h"
#include "nodes/converter/combine.h"
#include "nodes/converter/combinergb.h"
#include "nodes/converter/vectocolor.h"
#include "nodes/converter/grayscale.h"
#include "nodes/converter/blackbody.h"
#include "nodes/input/fresnel.h"
#include "nodes/input/normal.h"
#include "nodes/input/raylength.h"
#include "nodes/textures/image.h"
#include "nodes/textures/alpha.h"
#include "nodes/shaders/isotropic.h"
#include "nodes/shaders/add.h"

typedef const struct valueNode *V;
typedef const struct vectorNode *X;
typedef const struct colorNode *C;
typedef const struct bsdfNode *B;

static struct world *W;
static int g_dynamic;   /* known-answer operands as hit-dependent expressions (evaluated per hit, not folded at scene-compile time) */

static V num(float f) {
	V c = newConstantValue(W, f);
	/* 0 * rayLength + f == f exactly for every finite hit distance */
	return g_dynamic ? newMath(W, newMath(W, newRayLength(W), newConstantValue(W, 0.0f), Multiply), c, Add) : c;
}
static X vec(float x, float y, float z) {
	X c = newConstantVector(W, (struct vector){x, y, z});
	/* (0,0,0) * normal + v == v exactly */
	return g_dynamic ? newVecMath(W, newVecMath(W, newNormal(W), newConstantVector(W, (struct vector){0.0f, 0.0f, 0.0f}), VecMultiply), c, VecAdd) : c;
}
static V m2(V a, V b, enum mathOp op) { return newMath(W, a, b, op); }
static V m1(V a, enum mathOp op) { return newMath(W, a, NULL, op); }
static B show3(V r, V g, V b) { return newDiffuse(W, newCombineRGB(W, r, g, b)); }
static B showv(X v) { return newDiffuse(W, newVecToColor(W, v)); }
static C rgb(float r, float g, float b) { return newConstantTexture(W, (struct color){r, g, b, 1.0f}); }

/* tests/test_nodes.h:26-243 (mathnode_*): three results per sphere */
static void mathSpheres(struct sphere *s) {
	const float pi = (float)M_PI;
	s[0].material.bsdf = show3(m2(num(128.0f), num(128.0f), Add), m2(num(-128.0f), num(128.0f), Add), m2(num(128.0f), num(128.0f), Subtract));
	s[1].material.bsdf = show3(m2(num(-128.0f), num(128.0f), Subtract), m2(num(128.0f), num(128.0f), Multiply), m2(num(128.0f), num(128.0f), Divide));
	s[2].material.bsdf = show3(m2(num(-128.0f), num(1.0f), Divide), m2(num(2.0f), num(16.0f), Power), m2(num(-128.0f), num(1.0f), Power));
	s[3].material.bsdf = show3(m2(num(128.0f), num(0.0f), Power), m1(num(1.0f), Log), m1(num(10.0f), Log));
	s[4].material.bsdf = show3(m1(num(100.0f), Log), m1(num(1000.0f), Log), m1(num(10000.0f), Log));
	s[5].material.bsdf = show3(m1(num(9.0f), SquareRoot), m1(num(-128.0f), Absolute), m1(num(128.0f), Absolute));
	s[6].material.bsdf = show3(m2(num(-128.0f), num(128.0f), Min), m2(num(128.0f), num(42.0f), Min), m2(num(-128.0f), num(128.0f), Max));
	s[7].material.bsdf = show3(m2(num(128.0f), num(42.0f), Max), m1(num(pi), Sine), m1(num(pi), Cosine));
	s[8].material.bsdf = show3(m1(num(pi), Tangent), m1(num(180.0f), ToRadians), m1(num(pi), ToDegrees));
}

/* tests/test_nodes.h:245-395 (vecmath_*): one vector per sphere (dot / length return their float in .f, the shown vector is zero) */
static void vecSpheres(struct sphere *s) {
	const float h = sqrtf(0.5f);
	s[0].material.bsdf = showv(newVecMath(W, vec(1, 2, 3), vec(1, 2, 3), VecAdd));
	s[1].material.bsdf = showv(newVecMath(W, vec(1, 2, 3), vec(1, 2, 3), VecSubtract));
	s[2].material.bsdf = showv(newVecMath(W, vec(1, 2, 3), vec(1, 2, 3), VecMultiply));
	s[3].material.bsdf = showv(newVecMath(W, vec(0, 0, 0), vec(5, 5, 5), VecAverage));
	s[4].material.bsdf = showv(newVecMath(W, vec(0, 1, 0), vec(1, 0, 0), VecDot));
	s[5].material.bsdf = showv(newVecMath(W, vec(1, 0, 0), vec(0, 1, 0), VecCross));
	s[6].material.bsdf = showv(newVecMath(W, vec(1, 2, 3), NULL, VecNormalize));
	s[7].material.bsdf = showv(newVecMath(W, vec(h, h, 0), vec(0, -1, 0), VecReflect));
	s[8].material.bsdf = showv(newVecMath(W, vec(0, 2, 0), NULL, VecLength));
	s[9].material.bsdf = showv(newVecMath(W, vec(-10, 2, -3), NULL, VecAbs));
}

static void exoticSpheres(struct sphere *s) {
	char gridPath[] = "shapes/grid.png";         /* loadTexture() writes into its path argument (textureloader.c:54) */
	struct texture *grid = loadTexture(gridPath, &W->nodePool);
	V len = newRayLength(W);
	X n = newNormal(W);
	s[0].material.bsdf = show3(newFresnel(W, newConstantValue(W, 1.45f), NULL), m1(m2(len, newConstantValue(W, 3.0f), Multiply), Sine),
							   m1(m1(m2(len, newConstantValue(W, 7.0f), Multiply), Cosine), Absolute));
	s[1].material.bsdf = showv(newVecMath(W, n, NULL, VecAbs));
	s[2].material.bsdf = showv(newVecMath(W, newVecMath(W, n, newConstantVector(W, (struct vector){0.0f, 1.0f, 0.0f}), VecReflect),
										   newConstantVector(W, (struct vector){1.0f, 1.0f, 1.0f}), VecAverage));
	s[3].material.bsdf = newAdd(W, newDiffuse(W, rgb(0.5f, 0.2f, 0.1f)), newMetal(W, rgb(0.3f, 0.3f, 0.6f), newConstantValue(W, 0.2f)));
	s[4].material.bsdf = newAdd(W, newAdd(W, newDiffuse(W, rgb(0.2f, 0.3f, 0.1f)), newDiffuse(W, rgb(0.1f, 0.1f, 0.4f))),
								newMix(W, newMetal(W, rgb(0.9f, 0.6f, 0.2f), newConstantValue(W, 0.1f)), newDiffuse(W, rgb(0.3f, 0.3f, 0.3f)),
									   newFresnel(W, newConstantValue(W, 1.5f), NULL)));
	s[5].material.bsdf = newMix(W, newAdd(W, newDiffuse(W, rgb(0.4f, 0.1f, 0.1f)), newGlass(W, rgb(0.9f, 0.9f, 0.9f), newConstantValue(W, 0.0f), newConstantValue(W, 1.4f))),
								newPlastic(W, rgb(0.1f, 0.5f, 0.2f)), newConstantValue(W, 0.5f));
	s[6].material.bsdf = newIsotropic(W, rgb(0.8f, 0.5f, 0.3f));
	s[7].material.bsdf = newMix(W, newDiffuse(W, rgb(0.7f, 0.1f, 0.1f)), newMetal(W, rgb(0.9f, 0.9f, 0.9f), newConstantValue(W, 0.05f)),
								newFresnel(W, newConstantValue(W, 1.45f), n));
	s[8].material.bsdf = newEmission(W, newCombineValue(W, m2(newConstantValue(W, 1.0f), m2(len, newConstantValue(W, 1.0f), Add), Divide)),
									 m2(newConstantValue(W, 2.0f), newConstantValue(W, 1.5f), Power));
	s[9].material.bsdf = newMetal(W, rgb(0.8f, 0.8f, 0.3f), newGrayscaleConverter(W, newImageTexture(W, grid, NO_BILINEAR)));
	s[10].material.bsdf = newDiffuse(W, newImageTexture(W, grid, 0));
	s[11].material.bsdf = newGlass(W, rgb(1.0f, 1.0f, 1.0f), newConstantValue(W, 0.0f),
								   m2(newConstantValue(W, 1.0f), m2(newFresnel(W, newConstantValue(W, 1.2f), NULL), newConstantValue(W, 0.5f), Multiply), Add));
}

static void patchZoo(struct renderer *r) {
	struct sphere *s = W->spheres;
	if (W->sphereCount < 60) { fprintf(stderr, "CRH_NODE_PATCH=zoo needs the 60-sphere nodezoo scene (%d spheres)\n", W->sphereCount); exit(3); }
	if (!getenv("CRH_ZOO_TAME")) {      /* CRH_ZOO_TAME: leave the known-answer spheres (colours like 65536 or -256) plain grey for multi-bounce frames */
		g_dynamic = 0; mathSpheres(s + 0);
		g_dynamic = 1; mathSpheres(s + 9);
		g_dynamic = 0; vecSpheres(s + 18);
		g_dynamic = 1; vecSpheres(s + 28);
	}
	g_dynamic = 0; exoticSpheres(s + 38);
	W->background = newBackground(W, newConstantTexture(W, (struct color){1.0f, 1.0f, 1.0f, 1.0f}), NULL, NULL);
	(void)r;
}

/* Spheres whose radius is marked by a trailing ...7 in the fourth decimal (tools/gen_golden.py writes them so) and every instance of
 * the meshes whose indices are listed in CRH_VOLUME_MESHES ("0,2") become volumes. Density: CRH_VOLUME_DENSITY (default 6). */
static void patchVolumes(struct renderer *r) {
	const char *ds = getenv("CRH_VOLUME_DENSITY");
	const float density = ds ? (float)atof(ds) : 6.0f;
	const char *meshes = getenv("CRH_VOLUME_MESHES");
	int made = 0;
	for (int i = 0; i < W->instanceCount; ++i) {
		struct instance *inst = &W->instances[i];
		const struct transform keep = inst->composite;
		if (isMesh(inst)) {
			struct mesh *m = inst->object;
			char key[16];
			snprintf(key, sizeof(key), ",%d,", (int)(m - W->meshes));
			char list[256];
			snprintf(list, sizeof(list), ",%s,", meshes ? meshes : "");
			if (!strstr(list, key)) continue;
			m->materials[0].bsdf = newIsotropic(W, rgb(0.9f, 0.6f, 0.3f));
			*inst = newMeshVolume(m, density);
		} else {
			struct sphere *sp = inst->object;
			if (sp < W->spheres || sp >= W->spheres + W->sphereCount) continue;
			const int mark = (int)lroundf(sp->radius * 10000.0f) % 10;
			if (mark != 7) continue;
			sp->material.bsdf = newIsotropic(W, rgb(0.5f, 0.7f, 0.9f));
			*inst = newSphereVolume(sp, density);
		}
		inst->composite = keep;
		++made;
	}
	if (!made) { fprintf(stderr, "CRH_NODE_PATCH=volumes: nothing to convert\n"); exit(3); }
	destroyBvh(W->topLevel);
	W->topLevel = buildTopLevelBvh(W->instances, W->instanceCount);
	(void)r;
}

void crh_apply_node_patch(struct renderer *r) {
	const char *p = getenv("CRH_NODE_PATCH");
	if (!p || !*p) return;
	W = r->scene;
	if (!strcmp(p, "zoo")) patchZoo(r);
	else if (!strcmp(p, "volumes")) patchVolumes(r);
	else { fprintf(stderr, "unknown CRH_NODE_PATCH=%s\n", p); exit(3); }
}

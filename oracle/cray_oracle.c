/*
 * cray_oracle.c — CPU restatement of the c-ray v0.6.3 path-tracing hot path, operating on the
 * flattened scene (crh_scene_desc, include/cray_hip.h).
 *
 * THIS IS TEST INFRASTRUCTURE. It is the checker for the HIP path, never the thing measured or
 * shipped: only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load it.
 *
 * Parity status: PINNED. Compiled with the same flags as oracle/_ref/c-ray-ref-strict
 * (-O2 -march=x86-64-v3 -ffp-contract=off) it reproduces that binary's float render buffer
 * bit-for-bit on the golden fixtures (tests/test_oracle_golden.py, tests/golden/).
 *
 * Every function cites the reference file:line (under /root/reference/src) it restates. The float
 * expressions keep the reference's operand order and associativity; do not "simplify" them.
 */
#include <math.h>
#include <float.h>
#include <stdbool.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#include "cray_oracle.h"

#define PI 3.141592653589793238462643383279502f          /* includes.h:13 */
#define RMIN(a,b) (((a) < (b)) ? (a) : (b))              /* includes.h:20 */
#define RMAX(a,b) (((a) > (b)) ? (a) : (b))              /* includes.h:21 */

typedef struct { float x, y, z; } vec;                   /* vector.h:16-18 */
typedef struct { float x, y; } coord;                    /* vector.h:24-26 */
typedef struct { float r, g, b, a; } color;              /* color.h:13-15 */
typedef struct { vec start, direction; } ray_t;          /* lightray.h:21-25 */

typedef struct { uint64_t state, inc; } pcg32;           /* pcg_basic.h */

/* struct hitRecord (hitrecord.h:14-23); `material` is an index instead of a by-value copy
 * (only emission / IOR / bsdf are read: pathtrace.c:44,46, plastic.c:68-77). */
typedef struct {
	ray_t incident;
	uint32_t material;
	vec hitPoint;
	vec surfaceNormal;
	coord uv;
	float distance;
	int32_t polygon;     /* index into scene->polys, -1 = NULL */
	int32_t instIndex;
} hit_t;

typedef struct {
	const crh_scene_desc *s;
	pcg32 rng;
	int halton;                  /* sampler type: 0 = Random (renderThread), 1 = Halton (renderThreadInteractive) */
	float h_offset; int h_pass; unsigned h_prime;     /* haltonSampler: halton.h */
	crh_counters cnt;
	uint32_t ray_node_tests, ray_tri_tests;
} ctx_t;

/* ---- vector.h ------------------------------------------------------------------------------ */
static inline vec vecAdd(vec a, vec b) { return (vec){a.x + b.x, a.y + b.y, a.z + b.z}; }          /* vector.h:66 */
static inline vec vecSub(vec a, vec b) { return (vec){a.x - b.x, a.y - b.y, a.z - b.z}; }          /* vector.h:77 */
static inline vec vecMul(vec a, vec b) { return (vec){a.x * b.x, a.y * b.y, a.z * b.z}; }          /* vector.h:81 */
static inline float vecDot(vec a, vec b) { return a.x * b.x + a.y * b.y + a.z * b.z; }             /* vector.h:92 */
static inline vec vecScale(vec v, float c) { return (vec){v.x * c, v.y * c, v.z * c}; }            /* vector.h:103 */
static inline vec vecCross(vec a, vec b) {                                                          /* vector.h:122 */
	return (vec){ (a.y * b.z) - (a.z * b.y), (a.z * b.x) - (a.x * b.z), (a.x * b.y) - (a.y * b.x) };
}
static inline float vecLength(vec v) { return sqrtf(vecDot(v, v)); }                               /* vector.h:162 */
static inline vec vecNormalize(vec v) { float l = vecLength(v); return (vec){v.x / l, v.y / l, v.z / l}; } /* vector.h:173 */
static inline vec vecNegate(vec v) { return (vec){-v.x, -v.y, -v.z}; }                             /* vector.h:200 */
static inline vec vecReflect(vec I, vec N) { return vecSub(I, vecScale(N, vecDot(N, I) * 2.0f)); }  /* vector.h:211 */
static inline float wrapMax(float x, float max) { return fmodf(max + fmodf(x, max), max); }        /* vector.h:215 */
static inline float wrapMinMax(float x, float min, float max) { return min + wrapMax(x - min, max - min); } /* vector.h:219 */
static inline float clampf(float value, float mn, float mx) { return RMIN(RMAX(value, mn), mx); }  /* vector.h:55 */

/* ---- color.h ------------------------------------------------------------------------------- */
static inline color colorMul(color a, color b) { return (color){a.r * b.r, a.g * b.g, a.b * b.b, a.a * b.a}; } /* color.h:29 */
static inline color colorAdd(color a, color b) { return (color){a.r + b.r, a.g + b.g, a.b + b.b, a.a + b.a}; } /* color.h:33 */
static inline color colorCoef(float c, color a) { return (color){a.r * c, a.g * c, a.b * c, a.a * c}; }        /* color.h:42 */
static inline color colorMix(color c1, color c2, float coeff) {                                     /* color.h:46 */
	return colorAdd(colorCoef(1.0f - coeff, c1), colorCoef(coeff, c2));
}
static inline float linearToSRGB(float c) {                                                         /* color.h:51 */
	if (c <= 0.0031308f) return 12.92f * c;
	return (1.055f * powf(c, 0.4166666667f)) - 0.055f;
}
static inline float SRGBToLinear(float c) {                                                         /* color.h:59 */
	if (c <= 0.04045f) return c / 12.92f;
	return powf(((c + 0.055f) / 1.055f), 2.4f);
}
static inline color colorFromSRGB(color c) { return (color){SRGBToLinear(c.r), SRGBToLinear(c.g), SRGBToLinear(c.b), c.a}; } /* color.h:76 */
/* color.h:37-40 — note the double constant 0.587 (no f suffix): the sum is carried in double. */
static inline float grayscaleOf(color c) {
	return sqrtf(0.299f * powf(c.r, 2) + 0.587 * powf(c.g, 2) + 0.114f * powf(c.b, 2));
}
/* color.c:27-70 */
static color colorForKelvin(float kelvin) {
	color ret = {0};
	float temp = kelvin >= 40000.0f ? 40000.0f : kelvin;
	temp = temp / 100.0f;
	if (temp <= 66.0f) {
		ret.r = 255.0f;
	} else {
		ret.r = temp - 60.0f;
		ret.r = 329.698727446f * powf(ret.r, -0.1332047592f);
		ret.r = ret.r < 0.0f ? 0.0f : ret.r;
		ret.r = ret.r > 255.0f ? 255.0f : ret.r;
	}
	if (temp <= 66.0f) {
		ret.g = temp;
		ret.g = 99.4708025861f * logf(ret.g) - 161.1195681661f;
		ret.g = ret.g < 0.0f ? 0.0f : ret.g;
		ret.g = ret.g > 255.0f ? 255.0f : ret.g;
	} else {
		ret.g = temp - 60.0f;
		ret.g = 288.1221695283f * powf(ret.g, -0.0755148492f);
		ret.g = ret.g < 0.0f ? 0.0f : ret.g;
		ret.g = ret.g > 255.0f ? 255.0f : ret.g;
	}
	if (temp >= 66.0f) {
		ret.b = 255.0f;
	} else {
		if (temp <= 19.0f) {
			ret.b = 0.0f;
		} else {
			ret.b = temp - 10.0f;
			ret.b = 138.5177312231f * logf(ret.b) - 305.0447927307f;
			ret.b = ret.b < 0.0f ? 0.0f : ret.b;
			ret.b = ret.b > 255.0f ? 255.0f : ret.b;
		}
	}
	return (color){ret.r / 255.0f, ret.g / 255.0f, ret.b / 255.0f, 0};
}

/* ---- sampler: pcg_basic.c:42-68, samplers/common.h:22-27, sampler.c:41-44, random.c:12-21 ---- */
static inline uint32_t pcg32_random_r(pcg32 *rng) {
	uint64_t oldstate = rng->state;
	rng->state = oldstate * 6364136223846793005ULL + rng->inc;
	uint32_t xorshifted = (uint32_t)(((oldstate >> 18u) ^ oldstate) >> 27u);
	uint32_t rot = (uint32_t)(oldstate >> 59u);
	return (xorshifted >> rot) | (xorshifted << ((-rot) & 31));
}
static inline void pcg32_srandom_r(pcg32 *rng, uint64_t initstate, uint64_t initseq) {
	rng->state = 0U;
	rng->inc = (initseq << 1u) | 1u;
	pcg32_random_r(rng);
	rng->state += initstate;
	pcg32_random_r(rng);
}
static inline uint64_t hash64(uint64_t x) {
	x = (x ^ (x >> 30)) * UINT64_C(0xbf58476d1ce4e5b9);
	x = (x ^ (x >> 27)) * UINT64_C(0x94d049bb133111eb);
	x = x ^ (x >> 31);
	return x;
}
/* samplers/common.h:14-20, 30-57 + halton.c:16-31: the sampler of renderThreadInteractive (renderer.c:204) */
static inline uint32_t hash32(uint32_t x) {
	x = (x ^ 12345391u) * 2654435769u;
	x ^= (x << 6) ^ (x >> 26);
	x *= 2654435769u;
	x += (x << 5) ^ (x >> 12);
	return x;
}
static inline float radicalInverse(int pass, int base) {
	const float invBase = 1.0f / base;
	int reversedDigits = 0;
	float invBaseN = 1.0f;
	while (pass) {
		const int next = pass / base;
		const int digit = pass - base * next;
		reversedDigits = reversedDigits * base + digit;
		invBaseN *= invBase;
		pass = next;
	}
	const float v = reversedDigits * invBaseN;
	return v < 0.99999994f ? v : 0.99999994f;
}
static inline float uintToUnitReal(uint32_t v) {
	union { uint32_t u; float f; } x;
	x.u = (v >> 9) | 0x3f800000u;
	return x.f - 1.0f;
}
/* initSampler(.., Random, pass, maxPasses, pixelIndex): the multiply-add is 32-bit and wraps.
 * initSampler(.., Halton, pass, ..): renderThreadInteractive passes state.finishedPasses, which starts at 1
 * (renderer.c:333): interactive pass p (0-based here, like completedSamples - 1) is Halton index p + 1. */
static inline void initSampler(ctx_t *c, int pass, int maxPasses, uint32_t pixelIndex) {
	if (c->halton) {
		c->h_offset = uintToUnitReal(hash32(pixelIndex));
		c->h_pass = pass + 1;
		c->h_prime = 0;
		return;
	}
	uint32_t key = pixelIndex * (uint32_t)maxPasses + (uint32_t)pass;
	pcg32_srandom_r(&c->rng, hash64((uint64_t)key), 0);
}
static inline float getDimension(ctx_t *c) {
	if (c->halton) {
		static const int primes[6] = {2, 3, 5, 7, 11, 13};
		const float u = radicalInverse(c->h_pass, primes[c->h_prime++ % 6u]);
		return (u + c->h_offset < 1.0f) ? u + c->h_offset : u + c->h_offset - 1.0f;     /* wrapAdd, common.h:30-32 */
	}
	return (1.0f / (1ull << 32)) * pcg32_random_r(&c->rng);
}

/* vector.h:190-198 */
static inline coord randomCoordOnUnitDisc(ctx_t *c) {
	float r = sqrtf(getDimension(c));
	float theta = ((getDimension(c)) * ((2.0f * PI) - 0.0f)) + 0.0f;
	return (coord){r * cosf(theta), r * sinf(theta)};
}
/* vector.h:243-249 */
static inline vec randomOnUnitSphere(ctx_t *c) {
	const float sample_x = getDimension(c);
	const float sample_y = getDimension(c);
	const float a = sample_x * (2.0f * PI);
	const float s = 2.0f * sqrtf(RMAX(0.0f, sample_y * (1.0f - sample_y)));
	return (vec){cosf(a) * s, sinf(a) * s, 1.0f - 2.0f * sample_y};
}
/* vector.h:251-266 */
static inline bool refract(const vec *in, const vec normal, float niOverNt, vec *refracted) {
	const vec uv = vecNormalize(*in);
	const float dt = vecDot(uv, normal);
	const float discriminant = 1.0f - niOverNt * niOverNt * (1.0f - dt * dt);
	if (discriminant > 0.0f) {
		const vec A = vecScale(normal, dt);
		const vec B = vecSub(uv, A);
		const vec C = vecScale(B, niOverNt);
		const vec D = vecScale(normal, sqrtf(discriminant));
		*refracted = vecSub(C, D);
		return true;
	}
	return false;
}
/* vector.h:268-272 */
static inline float schlick(float cosine, float IOR) {
	float r0 = (1.0f - IOR) / (1.0f + IOR);
	r0 = r0 * r0;
	return r0 + (1.0f - r0) * powf((1.0f - cosine), 5.0f);
}

/* ---- transforms.c:76-116 on 3x4 row-major matrices --------------------------------------------- */
static inline void transformPoint(vec *v, const float *m) {
	vec t;
	t.x = (m[0] * v->x) + (m[1] * v->y) + (m[2] * v->z) + m[3];
	t.y = (m[4] * v->x) + (m[5] * v->y) + (m[6] * v->z) + m[7];
	t.z = (m[8] * v->x) + (m[9] * v->y) + (m[10] * v->z) + m[11];
	*v = t;
}
static inline void transformVector(vec *v, const float *m) {
	vec t;
	t.x = (m[0] * v->x) + (m[1] * v->y) + (m[2] * v->z);
	t.y = (m[4] * v->x) + (m[5] * v->y) + (m[6] * v->z);
	t.z = (m[8] * v->x) + (m[9] * v->y) + (m[10] * v->z);
	*v = t;
}
static inline void transformVectorWithTranspose(vec *v, const float *m) {
	vec t;
	t.x = (m[0] * v->x) + (m[4] * v->y) + (m[8] * v->z);
	t.y = (m[1] * v->x) + (m[5] * v->y) + (m[9] * v->z);
	t.z = (m[2] * v->x) + (m[6] * v->y) + (m[10] * v->z);
	*v = t;
}
static inline void transformRay(ray_t *r, const float *m) {
	transformPoint(&r->start, m);
	transformVector(&r->direction, m);
}
static inline vec alongRay(const ray_t *r, float t) { return vecAdd(r->start, vecScale(r->direction, t)); } /* lightray.h:31 */

static inline vec loadVec(const float *base, int64_t idx) { const float *p = base + 3 * idx; return (vec){p[0], p[1], p[2]}; }
static inline coord loadCoord(const float *base, int64_t idx) { const float *p = base + 2 * idx; return (coord){p[0], p[1]}; }

/* ---- camera.c:46-87 ---------------------------------------------------------------------------- */
static inline float signf_(float v) { return (v >= 0.0f) ? 1.0f : -1.0f; }
static inline float triangleDistribution(float v) {
	const float orig = v * 2.0f - 1.0f;
	v = orig / sqrtf(fabsf(orig));
	v = clampf(v, -1.0f, 1.0f);
	v = v - signf_(orig);
	return v;
}
static ray_t getCameraRay(ctx_t *c, int x, int y) {
	const crh_camera *cam = &c->s->camera;
	const vec right = {cam->right[0], cam->right[1], cam->right[2]};
	const vec up = {cam->up[0], cam->up[1], cam->up[2]};
	const vec forward = {cam->forward[0], cam->forward[1], cam->forward[2]};
	ray_t newRay;
	newRay.start = (vec){0.0f, 0.0f, 0.0f};
	const float jitterX = triangleDistribution(getDimension(c));
	const float jitterY = triangleDistribution(getDimension(c));
	vec pixX = vecScale(right, (cam->sensor[0] / cam->width));
	vec pixY = vecScale(up, (cam->sensor[1] / cam->height));
	vec pixV = vecAdd(forward,
					  vecAdd(vecScale(pixX, x - cam->width * 0.5f + jitterX + 0.5f),
							 vecScale(pixY, y - cam->height * 0.5f + jitterY + 0.5f)));
	newRay.direction = vecNormalize(pixV);
	if (cam->aperture > 0.0f) {
		float ft = cam->focal_distance / vecDot(newRay.direction, forward);
		vec focusPoint = alongRay(&newRay, ft);
		coord disc = randomCoordOnUnitDisc(c);
		coord lensPoint = {disc.x * cam->aperture, disc.y * cam->aperture};
		newRay.start = vecAdd(newRay.start, vecAdd(vecScale(right, lensPoint.x), vecScale(up, lensPoint.y)));
		newRay.direction = vecNormalize(vecSub(focusPoint, newRay.start));
	}
	transformRay(&newRay, cam->A);
	return newRay;
}

/* ---- textures: texture.c:32-79, image.c:31-48 -------------------------------------------------- */
static color textureGetPixelInternal(ctx_t *c, const crh_texture *t, size_t x, size_t y) {
	color o = {0.0f, 0.0f, 0.0f, 0.0f};
	const uint8_t *bytes = c->s->texture_data + t->offset;
	const float *floats = (const float *)bytes;
	const size_t W = t->width, H = t->height, C = t->channels;
	c->cnt.tex_fetches++;
	x = x % W;
	y = y % H;
	const size_t base = (x + ((H - 1) - y) * W) * C;
	if (C == 1) {
		o.r = t->is_float ? floats[base] : bytes[base] / 255.0f;
		o.g = o.r; o.b = o.r; o.a = 1.0f;
	} else if (t->is_float) {
		o.r = floats[base + 0]; o.g = floats[base + 1]; o.b = floats[base + 2];
		o.a = t->has_alpha ? floats[base + 3] : 1.0f;
	} else {
		o.r = bytes[base + 0] / 255.0f; o.g = bytes[base + 1] / 255.0f; o.b = bytes[base + 2] / 255.0f;
		o.a = t->has_alpha ? bytes[base + 3] / 255.0f : 1.0f;
	}
	return o;
}
static color textureGetPixel(ctx_t *c, const crh_texture *t, float x, float y, bool filtered) {
	if (!filtered) return textureGetPixelInternal(c, t, (size_t)x, (size_t)y);
	x = x * t->width;
	y = y * t->height;
	float xcopy = x - 0.5f;
	float ycopy = y - 0.5f;
	int xint = (int)xcopy;
	int yint = (int)ycopy;
	color topleft = textureGetPixelInternal(c, t, (size_t)xint, (size_t)yint);
	color topright = textureGetPixelInternal(c, t, (size_t)(xint + 1), (size_t)yint);
	color botleft = textureGetPixelInternal(c, t, (size_t)xint, (size_t)(yint + 1));
	color botright = textureGetPixelInternal(c, t, (size_t)(xint + 1), (size_t)(yint + 1));
	return colorMix(colorMix(topleft, topright, xcopy - xint), colorMix(botleft, botright, xcopy - xint), ycopy - yint);
}

/* ---- node graph: src/nodes/ ---------------------------------------------------------------------- */
static float evalValue(ctx_t *c, uint32_t idx, const hit_t *rec);
static vec evalVector(ctx_t *c, uint32_t idx, const hit_t *rec, float *fOut);

static color evalColor(ctx_t *c, uint32_t idx, const hit_t *rec) {
	const crh_gnode *n = &c->s->gnodes[idx];
	switch (n->kind) {
		case CRH_COLOR_CONSTANT:   /* constant.c:39-42 */
			return (color){n->f[0], n->f[1], n->f[2], n->f[3]};
		case CRH_COLOR_IMAGE: {    /* image.c:31-48 */
			if (n->a == CRH_NODE_NONE) return (color){1.0f, 0.0f, 0.5f, 1.0f}; /* warningMaterial().diffuse, material.c:38 */
			const crh_texture *t = &c->s->textures[n->a];
			color out;
			if (n->b & CRH_IMAGE_NO_BILINEAR) {
				float x = rec->uv.x * t->width;
				float y = rec->uv.y * t->height;
				out = textureGetPixel(c, t, x, y, false);
			} else {
				out = textureGetPixel(c, t, rec->uv.x, rec->uv.y, true);
			}
			if (n->b & CRH_IMAGE_SRGB_TRANSFORM) out = colorFromSRGB(out);
			return out;
		}
		case CRH_COLOR_CHECKER: {  /* checker.c:31-54 */
			const float coef = evalValue(c, n->c, rec);
			float sines;
			if (rec->uv.x >= 0) sines = sinf(coef * rec->uv.x) * sinf(coef * rec->uv.y);
			else sines = sinf(coef * rec->hitPoint.x) * sinf(coef * rec->hitPoint.y) * sinf(coef * rec->hitPoint.z);
			return evalColor(c, sines < 0.0f ? n->a : n->b, rec);
		}
		case CRH_COLOR_GRADIENT: { /* gradient.c:40-45 */
			vec unitDir = vecNormalize(rec->incident.direction);
			float t = 0.5f * (unitDir.y + 1.0f);
			color down = {n->f[0], n->f[1], n->f[2], n->f[3]}, up = {n->f[4], n->f[5], n->f[6], n->f[7]};
			return colorAdd(colorCoef(1.0f - t, down), colorCoef(t, up));
		}
		case CRH_COLOR_BLACKBODY:  /* blackbody.c:38-42 */
			return colorForKelvin(evalValue(c, n->a, rec));
		case CRH_COLOR_COMBINE: {  /* combine.c:38-43 */
			float v = evalValue(c, n->a, rec);
			return (color){v, v, v, 1.0f};
		}
		case CRH_COLOR_COMBINERGB: { /* combinergb.c:42-51 */
			color o;
			o.r = evalValue(c, n->a, rec); o.g = evalValue(c, n->b, rec); o.b = evalValue(c, n->c, rec); o.a = 1.0f;
			return o;
		}
		case CRH_COLOR_VECTOCOLOR: { /* vectocolor.c:38-43 */
			vec v = evalVector(c, n->a, rec, NULL);
			return (color){v.x, v.y, v.z, 0.0f};
		}
		default:
			return (color){0, 0, 0, 0};
	}
}

static float evalValue(ctx_t *c, uint32_t idx, const hit_t *rec) {
	const crh_gnode *n = &c->s->gnodes[idx];
	switch (n->kind) {
		case CRH_VALUE_CONSTANT: return n->f[0];                               /* valuenode.c:36-40 */
		case CRH_VALUE_ALPHA: return evalColor(c, n->a, rec).a;                /* alpha.c:38-41 */
		case CRH_VALUE_GRAYSCALE: return grayscaleOf(evalColor(c, n->a, rec)); /* grayscale.c:38-41 */
		case CRH_VALUE_RAYLENGTH: return rec->distance;                        /* raylength.c:36-40 */
		case CRH_VALUE_FRESNEL: {                                              /* fresnel.c:38-51 */
			float IOR = evalValue(c, n->a, rec);
			float cosine;
			if (vecDot(rec->incident.direction, rec->surfaceNormal) > 0.0f)
				cosine = IOR * vecDot(rec->incident.direction, rec->surfaceNormal) / vecLength(rec->incident.direction);
			else
				cosine = -(vecDot(rec->incident.direction, rec->surfaceNormal) / vecLength(rec->incident.direction));
			return schlick(cosine, evalValue(c, n->a, rec));
		}
		case CRH_VALUE_MATH: {                                                 /* math.c:42-95 */
			const float a = evalValue(c, n->a, rec);
			const float b = evalValue(c, n->b, rec);
			switch (n->c) {
				case 0: return a + b;
				case 1: return a - b;
				case 2: return a * b;
				case 3: return a / b;
				case 4: return powf(a, b);
				case 5: return log10f(a);
				case 6: return sqrtf(a);
				case 7: return fabsf(a);
				case 8: return RMIN(a, b);
				case 9: return RMAX(a, b);
				case 10: return sinf(a);
				case 11: return cosf(a);
				case 12: return tanf(a);
				case 13: return (a * PI) / 180.0f;          /* toRadians, transforms.c:18 */
				case 14: return a * (180.0f / PI);          /* fromRadians, transforms.c:22 */
			}
			return 0.0f;
		}
		default: return 0.0f;
	}
}

static vec evalVector(ctx_t *c, uint32_t idx, const hit_t *rec, float *fOut) {
	const crh_gnode *n = &c->s->gnodes[idx];
	if (fOut) *fOut = 0.0f;
	switch (n->kind) {
		case CRH_VEC_CONSTANT: return (vec){n->f[0], n->f[1], n->f[2]};        /* vectornode.c:38-42 */
		case CRH_VEC_NORMAL: return rec->surfaceNormal;                        /* normal.c:37-41 */
		case CRH_VEC_VECMATH: {                                                /* vecmath.c:42-81 */
			const vec a = evalVector(c, n->a, rec, NULL);
			const vec b = evalVector(c, n->b, rec, NULL);
			switch (n->c) {
				case 0: return vecAdd(a, b);
				case 1: return vecSub(a, b);
				case 2: return vecMul(a, b);
				case 3: return vecScale(vecAdd(a, b), 0.5f);
				case 4: if (fOut) *fOut = vecDot(a, b); return (vec){0, 0, 0};
				case 5: return vecCross(a, b);
				case 6: return vecNormalize(a);
				case 7: return vecReflect(a, b);
				case 8: if (fOut) *fOut = vecLength(a); return (vec){0, 0, 0};
				case 9: return (vec){fabsf(a.x), fabsf(a.y), fabsf(a.z)};
			}
			return (vec){0, 0, 0};
		}
		default: return (vec){0, 0, 0};
	}
}

typedef struct { vec out; color col; } bsdf_sample;   /* bsdfnode.h:19-23 (pdf unused) */

static bsdf_sample sampleBsdf(ctx_t *c, uint32_t idx, hit_t *rec) {
	const crh_gnode *n = &c->s->gnodes[idx];
	const crh_material *mat = &c->s->materials[rec->material];
	switch (n->kind) {
		case CRH_BSDF_DIFFUSE: {      /* diffuse.c:40-47 */
			const vec scatterDir = vecNormalize(vecAdd(rec->surfaceNormal, randomOnUnitSphere(c)));
			return (bsdf_sample){scatterDir, evalColor(c, n->a, rec)};
		}
		case CRH_BSDF_METAL: {        /* metal.c:40-55 */
			const vec normalizedDir = vecNormalize(rec->incident.direction);
			vec reflected = vecReflect(normalizedDir, rec->surfaceNormal);
			float roughness = evalValue(c, n->b, rec);
			if (roughness > 0.0f) {
				const vec fuzz = vecScale(randomOnUnitSphere(c), roughness);
				reflected = vecAdd(reflected, fuzz);
			}
			return (bsdf_sample){reflected, evalColor(c, n->a, rec)};
		}
		case CRH_BSDF_GLASS: {        /* glass.c:41-87 */
			vec outwardNormal;
			vec reflected = vecReflect(rec->incident.direction, rec->surfaceNormal);
			float niOverNt;
			vec refracted = {0, 0, 0};
			float reflectionProbability;
			float cosine;
			float IOR = evalValue(c, n->c, rec);
			if (vecDot(rec->incident.direction, rec->surfaceNormal) > 0.0f) {
				outwardNormal = vecNegate(rec->surfaceNormal);
				niOverNt = IOR;
				cosine = IOR * vecDot(rec->incident.direction, rec->surfaceNormal) / vecLength(rec->incident.direction);
			} else {
				outwardNormal = rec->surfaceNormal;
				niOverNt = 1.0f / IOR;
				cosine = -(vecDot(rec->incident.direction, rec->surfaceNormal) / vecLength(rec->incident.direction));
			}
			if (refract(&rec->incident.direction, outwardNormal, niOverNt, &refracted)) {
				reflectionProbability = schlick(cosine, IOR);
			} else {
				reflectionProbability = 1.0f;
			}
			float roughness = evalValue(c, n->b, rec);
			if (roughness > 0.0f) {
				vec fuzz = vecScale(randomOnUnitSphere(c), roughness);
				reflected = vecAdd(reflected, fuzz);
				refracted = vecAdd(refracted, fuzz);
			}
			vec scatterDir;
			if (getDimension(c) < reflectionProbability) scatterDir = reflected;
			else scatterDir = refracted;
			return (bsdf_sample){scatterDir, evalColor(c, n->a, rec)};
		}
		case CRH_BSDF_PLASTIC: {      /* plastic.c:42-87 */
			vec outwardNormal;
			float niOverNt;
			vec refracted;
			float reflectionProbability;
			float cosine;
			if (vecDot(rec->incident.direction, rec->surfaceNormal) > 0.0f) {
				outwardNormal = vecNegate(rec->surfaceNormal);
				niOverNt = mat->ior;
				cosine = mat->ior * vecDot(rec->incident.direction, rec->surfaceNormal) / vecLength(rec->incident.direction);
			} else {
				outwardNormal = rec->surfaceNormal;
				niOverNt = 1.0f / mat->ior;
				cosine = -(vecDot(rec->incident.direction, rec->surfaceNormal) / vecLength(rec->incident.direction));
			}
			if (refract(&rec->incident.direction, outwardNormal, niOverNt, &refracted)) {
				reflectionProbability = schlick(cosine, mat->ior);
			} else {
				reflectionProbability = 1.0f;
			}
			if (getDimension(c) < reflectionProbability) {
				/* sampleShiny, plastic.c:42-55 */
				vec reflected = vecReflect(rec->incident.direction, rec->surfaceNormal);
				float roughness = evalColor(c, n->b, rec).r;
				if (roughness > 0.0f) {
					const vec fuzz = vecScale(randomOnUnitSphere(c), roughness);
					reflected = vecAdd(reflected, fuzz);
				}
				return (bsdf_sample){reflected, (color){1.0f, 1.0f, 1.0f, 1.0f}};
			}
			return sampleBsdf(c, n->c, rec);
		}
		case CRH_BSDF_MIX: {          /* mix.c:42-50 */
			const float lerp = evalValue(c, n->c, rec);
			if (getDimension(c) > lerp) return sampleBsdf(c, n->a, rec);
			return sampleBsdf(c, n->b, rec);
		}
		case CRH_BSDF_ADD: {          /* add.c:42-49 */
			bsdf_sample A = sampleBsdf(c, n->a, rec);
			bsdf_sample B = sampleBsdf(c, n->b, rec);
			return (bsdf_sample){vecAdd(A.out, B.out), colorAdd(A.col, B.col)};
		}
		case CRH_BSDF_TRANSPARENT:    /* transparent.c:40-44 */
			return (bsdf_sample){rec->incident.direction, evalColor(c, n->a, rec)};
		case CRH_BSDF_EMISSION: {     /* emission.c:42-49 */
			const vec scatterDir = vecNormalize(vecAdd(rec->surfaceNormal, randomOnUnitSphere(c)));
			return (bsdf_sample){scatterDir, colorCoef(evalValue(c, n->b, rec), evalColor(c, n->a, rec))};
		}
		case CRH_BSDF_ISOTROPIC: {    /* isotropic.c:40-47 */
			const vec scatterDir = vecNormalize(randomOnUnitSphere(c));
			return (bsdf_sample){scatterDir, evalColor(c, n->a, rec)};
		}
		case CRH_BSDF_BACKGROUND: {   /* background.c:39-66 */
			vec ud = vecNormalize(rec->incident.direction);
			float r = 1.0f;
			float phi = (atan2f(ud.z, ud.x) / 4.0f) + evalValue(c, n->c, rec);
			float theta = acosf((-ud.y / r));
			float u = theta / PI;
			float v = (phi / (PI / 2.0f));
			u = wrapMinMax(u, 0.0f, 1.0f);
			v = wrapMinMax(v, 0.0f, 1.0f);
			rec->uv = (coord){v, u};
			float strength = evalValue(c, n->b, rec);
			return (bsdf_sample){(vec){0, 0, 0}, colorCoef(strength, evalColor(c, n->a, rec))};
		}
		default:
			return (bsdf_sample){(vec){0, 0, 0}, (color){0, 0, 0, 0}};
	}
}

/* ---- intersection ------------------------------------------------------------------------------ */
/* poly.c:17-53 */
static bool rayIntersectsWithPolygon(ctx_t *c, const ray_t *ray, int32_t polyIndex, hit_t *isect) {
	const crh_scene_desc *s = c->s;
	const crh_poly *poly = &s->polys[polyIndex];
	c->cnt.tri_tests++; c->ray_tri_tests++;
	const vec v0 = loadVec(s->vertices, poly->v[0]), v1 = loadVec(s->vertices, poly->v[1]), v2 = loadVec(s->vertices, poly->v[2]);
	vec e1 = vecSub(v0, v1);
	vec e2 = vecSub(v2, v0);
	vec n = vecCross(e1, e2);
	vec cc = vecSub(v0, ray->start);
	vec r = vecCross(ray->direction, cc);
	float invDet = 1.0f / vecDot(n, ray->direction);
	float u = vecDot(r, e2) * invDet;
	float v = vecDot(r, e1) * invDet;
	float w = 1.0f - u - v;
	if (u >= 0.0f && v >= 0.0f && u + v <= 1.0f) {
		float t = vecDot(n, cc) * invDet;
		if (t >= 0.0f && t < isect->distance) {
			isect->uv = (coord){u, v};
			isect->distance = t;
			if (CRH_POLY_HASNORMALS(*poly)) {
				vec upcomp = vecScale(loadVec(s->normals, poly->n[1]), u);
				vec vpcomp = vecScale(loadVec(s->normals, poly->n[2]), v);
				vec wpcomp = vecScale(loadVec(s->normals, poly->n[0]), w);
				isect->surfaceNormal = vecAdd(vecAdd(upcomp, vpcomp), wpcomp);
			} else {
				isect->surfaceNormal = n;
			}
			isect->hitPoint = alongRay(ray, t);
			return true;
		}
	}
	return false;
}

/* sphere.c:20-61 */
static bool rayIntersectsWithSphere(ctx_t *c, const ray_t *ray, const crh_sphere *sphere, hit_t *isect) {
	c->cnt.sphere_tests++;
	float A = vecDot(ray->direction, ray->direction);
	float B = 2.0f * vecDot(ray->direction, ray->start);
	float C = vecDot(ray->start, ray->start) - (sphere->radius * sphere->radius);
	float trigDiscriminant = B * B - 4.0f * A * C;
	if (trigDiscriminant < 0.0f) return false;
	float sqrtOfDiscriminant = sqrtf(trigDiscriminant);
	float t0 = (-B + sqrtOfDiscriminant) / 2.0f;
	float t1 = (-B - sqrtOfDiscriminant) / 2.0f;
	if (t0 > t1 && t1 > 0.0f) t0 = t1;
	if (t0 < 0.00001f || t0 > isect->distance) return false;
	isect->distance = t0;
	isect->hitPoint = alongRay(ray, isect->distance);
	isect->surfaceNormal = vecNormalize(isect->hitPoint);
	isect->polygon = -1;
	return true;
}

/* bvh.c:326-352 — FP_FAST_FMAF is defined for the reference build, so fastMultiplyAdd is fmaf */
static inline bool intersectNode(ctx_t *c, const crh_bvh_node *node, const vec *invDir, const vec *scaledStart,
								 const int *octant, float maxDist, float *tEntry) {
	c->cnt.node_tests++; c->ray_node_tests++;
	float tMinX = fmaf(node->bounds[0 +     octant[0]], invDir->x, scaledStart->x);
	float tMaxX = fmaf(node->bounds[0 + 1 - octant[0]], invDir->x, scaledStart->x);
	float tMinY = fmaf(node->bounds[2 +     octant[1]], invDir->y, scaledStart->y);
	float tMaxY = fmaf(node->bounds[2 + 1 - octant[1]], invDir->y, scaledStart->y);
	float tMinZ = fmaf(node->bounds[4 +     octant[2]], invDir->z, scaledStart->z);
	float tMaxZ = fmaf(node->bounds[4 + 1 - octant[2]], invDir->z, scaledStart->z);
	float tMin = tMinX > tMinY ? tMinX : tMinY;
	float tMax = tMaxX < tMaxY ? tMaxX : tMaxY;
	tMin = tMin > tMinZ ? tMin : tMinZ;
	tMax = tMax < tMaxZ ? tMax : tMaxZ;
	tMin = tMin > 0 ? tMin : 0;
	tMax = tMax < maxDist ? tMax : maxDist;
	*tEntry = tMin;
	return tMin <= tMax;
}

struct bvh_view { const crh_bvh_node *nodes; const int32_t *prims; uint32_t nodeCount; };
typedef bool (*leaf_fn)(ctx_t *, void *user, const struct bvh_view *, const crh_bvh_node *, const ray_t *, hit_t *);

/* bvh.c:354-441 */
static bool traverseBvhGeneric(ctx_t *c, void *user, const struct bvh_view *bvh, leaf_fn intersectLeaf, const ray_t *ray, hit_t *isect) {
	if (bvh->nodeCount < 1) {
		isect->instIndex = -1;
		return false;
	}
	const crh_bvh_node *stack[64 + 1];
	int stackSize = 0;
	int octant[] = {
		signbit(ray->direction.x) ? 1 : 0,
		signbit(ray->direction.y) ? 1 : 0,
		signbit(ray->direction.z) ? 1 : 0
	};
	vec invDir = {1.0f / ray->direction.x, 1.0f / ray->direction.y, 1.0f / ray->direction.z};
	vec scaledStart = vecScale(vecMul(ray->start, invDir), -1.0f);
	float maxDist = isect->distance;

	if (bvh->nodeCount == 1) {
		float tEntry;
		if (intersectNode(c, bvh->nodes, &invDir, &scaledStart, octant, maxDist, &tEntry))
			return intersectLeaf(c, user, bvh, bvh->nodes, ray, isect);
		return false;
	}

	const crh_bvh_node *node = bvh->nodes;
	bool hasHit = false;
	while (true) {
		unsigned firstChild = node->first;
		const crh_bvh_node *leftNode = &bvh->nodes[firstChild];
		const crh_bvh_node *rightNode = &bvh->nodes[firstChild + 1];
		float tEntryLeft, tEntryRight;
		bool hitLeft = intersectNode(c, leftNode, &invDir, &scaledStart, octant, maxDist, &tEntryLeft);
		bool hitRight = intersectNode(c, rightNode, &invDir, &scaledStart, octant, maxDist, &tEntryRight);
		if (hitLeft) {
			if (CRH_NODE_ISLEAF(*leftNode)) {
				if (intersectLeaf(c, user, bvh, leftNode, ray, isect)) {
					maxDist = isect->distance;
					hasHit = true;
				}
				leftNode = NULL;
			}
		} else
			leftNode = NULL;
		if (hitRight) {
			if (CRH_NODE_ISLEAF(*rightNode)) {
				if (intersectLeaf(c, user, bvh, rightNode, ray, isect)) {
					maxDist = isect->distance;
					hasHit = true;
				}
				rightNode = NULL;
			}
		} else
			rightNode = NULL;
		if ((rightNode != NULL) & (leftNode != NULL)) {
			if (tEntryLeft > tEntryRight) {
				node = leftNode;
				leftNode = rightNode;
				rightNode = node;
			}
			node = leftNode;
			stack[stackSize++] = rightNode;
		} else if ((rightNode != NULL) ^ (leftNode != NULL)) {
			node = rightNode != NULL ? rightNode : leftNode;
		} else {
			if (stackSize == 0) break;
			node = stack[--stackSize];
		}
	}
	return hasHit;
}

/* bvh.c:443-462 */
static bool intersectBottomLevelLeaf(ctx_t *c, void *user, const struct bvh_view *bvh, const crh_bvh_node *leaf, const ray_t *ray, hit_t *isect) {
	const crh_mesh *mesh = user;
	bool found = false;
	for (int i = 0; i < (int)CRH_NODE_PRIMCOUNT(*leaf); ++i) {
		int32_t p = (int32_t)mesh->poly_base + bvh->prims[leaf->first + i];
		if (rayIntersectsWithPolygon(c, ray, p, isect)) {
			isect->polygon = p;
			found = true;
		}
	}
	return found;
}

/* instance.c:33-43 */
static coord getTexMapSphere(const hit_t *isect) {
	vec ud = isect->surfaceNormal;
	float phi = atan2f(ud.z, ud.x);
	float theta = asinf(ud.y);
	float v = (theta + PI / 2.0f) / PI;
	float u = 1.0f - (phi + PI) / (PI * 2.0f);
	u = wrapMinMax(u, 0.0f, 1.0f);
	v = wrapMinMax(v, 0.0f, 1.0f);
	return (coord){u, v};
}

/* instance.c:45-60 */
static bool intersectSphere(ctx_t *c, const crh_instance *inst, const ray_t *ray, hit_t *isect) {
	ray_t copy = *ray;
	transformRay(&copy, inst->Ainv);
	const crh_sphere *sphere = &c->s->spheres[inst->object];
	copy.start = vecAdd(copy.start, vecScale(copy.direction, sphere->ray_offset));
	if (rayIntersectsWithSphere(c, &copy, sphere, isect)) {
		isect->uv = getTexMapSphere(isect);
		isect->polygon = -1;
		isect->material = sphere->material;
		transformPoint(&isect->hitPoint, inst->A);
		transformVectorWithTranspose(&isect->surfaceNormal, inst->Ainv);
		return true;
	}
	return false;
}

/* instance.c:150-167 */
static coord getTexMapMesh(const crh_scene_desc *s, const crh_mesh *mesh, const hit_t *isect) {
	if (mesh->texcoord_count == 0) return (coord){-1.0f, -1.0f};
	const crh_poly *p = &s->polys[isect->polygon];
	if (p->t[0] == -1) return (coord){-1.0f, -1.0f};
	const float u = isect->uv.x;
	const float v = isect->uv.y;
	const float w = 1.0f - u - v;
	const coord t1 = loadCoord(s->texcoords, p->t[1]), t2 = loadCoord(s->texcoords, p->t[2]), t0 = loadCoord(s->texcoords, p->t[0]);
	const coord ucomponent = {t1.x * u, t1.y * u};
	const coord vcomponent = {t2.x * v, t2.y * v};
	const coord wcomponent = {t0.x * w, t0.y * w};
	return (coord){(ucomponent.x + vcomponent.x) + wcomponent.x, (ucomponent.y + vcomponent.y) + wcomponent.y};
}

/* instance.c:169-185 + bvh.c:464-466 */
static bool intersectMesh(ctx_t *c, const crh_instance *inst, const ray_t *ray, hit_t *isect) {
	const crh_scene_desc *s = c->s;
	ray_t copy = *ray;
	transformRay(&copy, inst->Ainv);
	const crh_mesh *mesh = &s->meshes[inst->object];
	float offset = mesh->ray_offset;
	copy.start = vecAdd(copy.start, vecScale(copy.direction, offset));
	struct bvh_view blas = { s->nodes + mesh->node_base, s->prim_indices + mesh->prim_base, mesh->node_count };
	if (traverseBvhGeneric(c, (void *)mesh, &blas, intersectBottomLevelLeaf, &copy, isect)) {
		isect->uv = getTexMapMesh(s, mesh, isect);
		isect->material = mesh->material_base + CRH_POLY_MATERIAL(s->polys[isect->polygon]);
		transformPoint(&isect->hitPoint, inst->A);
		transformVectorWithTranspose(&isect->surfaceNormal, inst->Ainv);
		isect->surfaceNormal = vecNormalize(isect->surfaceNormal);
		return true;
	}
	return false;
}

/* instance.c:62-92. The medium is sampled INSIDE the traversal: one sampler draw, and only when the ray enters and leaves the sphere
 * within the current closest distance. The hit point is built from the WORLD ray and then transformed again (as the reference does). */
static bool intersectSphereVolume(ctx_t *c, const crh_instance *inst, const ray_t *ray, hit_t *isect) {
	hit_t record1 = *isect, record2 = *isect;
	ray_t copy1 = *ray;
	transformRay(&copy1, inst->Ainv);
	const crh_sphere *sphere = &c->s->spheres[inst->object];
	copy1.start = vecAdd(copy1.start, vecScale(copy1.direction, sphere->ray_offset));
	if (rayIntersectsWithSphere(c, &copy1, sphere, &record1)) {
		ray_t copy2 = { alongRay(&copy1, record1.distance + 0.0001f), copy1.direction };
		if (rayIntersectsWithSphere(c, &copy2, sphere, &record2)) {
			if (record1.distance < 0.0f) record1.distance = 0.0f;
			float distanceInsideVolume = record2.distance;
			float hitDistance = -(1.0f / inst->density) * logf(getDimension(c));
			if (hitDistance < distanceInsideVolume) {
				isect->distance = record1.distance + hitDistance;
				isect->hitPoint = alongRay(ray, isect->distance);
				isect->uv = (coord){-1.0f, -1.0f};
				isect->polygon = -1;
				isect->material = sphere->material;
				transformPoint(&isect->hitPoint, inst->A);
				isect->surfaceNormal = (vec){1.0f, 0.0f, 0.0f};
				transformVectorWithTranspose(&isect->surfaceNormal, inst->Ainv);
				return true;
			}
		}
	}
	return false;
}

/* instance.c:187-216: two BLAS walks (entry, then exit from just behind the entry point), then the same free-flight sampling */
static bool intersectMeshVolume(ctx_t *c, const crh_instance *inst, const ray_t *ray, hit_t *isect) {
	const crh_scene_desc *s = c->s;
	hit_t record1 = *isect, record2 = *isect;
	ray_t copy = *ray;
	transformRay(&copy, inst->Ainv);
	const crh_mesh *mesh = &s->meshes[inst->object];
	float offset = mesh->ray_offset;
	copy.start = vecAdd(copy.start, vecScale(copy.direction, offset));
	struct bvh_view blas = { s->nodes + mesh->node_base, s->prim_indices + mesh->prim_base, mesh->node_count };
	if (traverseBvhGeneric(c, (void *)mesh, &blas, intersectBottomLevelLeaf, &copy, &record1)) {
		ray_t copy2 = { alongRay(&copy, record1.distance + 0.0001f), copy.direction };
		if (traverseBvhGeneric(c, (void *)mesh, &blas, intersectBottomLevelLeaf, &copy2, &record2)) {
			if (record1.distance < 0.0f) record1.distance = 0.0f;
			float distanceInsideVolume = record2.distance;
			float hitDistance = -(1.0f / inst->density) * logf(getDimension(c));
			if (hitDistance < distanceInsideVolume) {
				isect->distance = record1.distance + hitDistance;
				isect->hitPoint = alongRay(ray, isect->distance);
				isect->uv = (coord){-1.0f, -1.0f};
				isect->material = mesh->material_base;          /* mesh->materials[0] */
				transformPoint(&isect->hitPoint, inst->A);
				isect->surfaceNormal = (vec){1.0f, 0.0f, 0.0f};
				transformVectorWithTranspose(&isect->surfaceNormal, inst->Ainv);
				return true;
			}
		}
	}
	return false;
}

/* bvh.c:468-486 */
static bool intersectTopLevelLeaf(ctx_t *c, void *user, const struct bvh_view *bvh, const crh_bvh_node *leaf, const ray_t *ray, hit_t *isect) {
	(void)user;
	const crh_scene_desc *s = c->s;
	bool found = false;
	for (int i = 0; i < (int)CRH_NODE_PRIMCOUNT(*leaf); ++i) {
		int currIndex = bvh->prims[leaf->first + i];
		const crh_instance *inst = &s->instances[currIndex];
		c->cnt.inst_visits++;
		bool hit;
		switch (inst->kind) {          /* instance->intersectFn */
			case CRH_INSTANCE_SPHERE: hit = intersectSphere(c, inst, ray, isect); break;
			case CRH_INSTANCE_SPHERE_VOLUME: hit = intersectSphereVolume(c, inst, ray, isect); break;
			case CRH_INSTANCE_MESH_VOLUME: hit = intersectMeshVolume(c, inst, ray, isect); break;
			default: hit = intersectMesh(c, inst, ray, isect); break;
		}
		if (hit) {
			c->cnt.inst_hits++;
			isect->instIndex = currIndex;
			found = true;
		}
	}
	return found;
}

/* pathtrace.c:26-30 */
static hit_t getClosestIsect(ctx_t *c, const ray_t *incidentRay) {
	const crh_scene_desc *s = c->s;
	hit_t isect;
	memset(&isect, 0, sizeof(isect));
	isect.incident = *incidentRay;
	isect.instIndex = -1;
	isect.distance = FLT_MAX;
	isect.polygon = -1;
	c->cnt.rays++;
	c->ray_node_tests = 0; c->ray_tri_tests = 0;
	struct bvh_view tlas = { s->nodes + s->tlas_node_base, s->prim_indices + s->tlas_prim_base, s->tlas_node_count };
	traverseBvhGeneric(c, NULL, &tlas, intersectTopLevelLeaf, incidentRay, &isect);
	return isect;
}

/* pathtrace.c:32-60 */
static color pathTrace(ctx_t *c, const ray_t *incidentRay, int maxDepth) {
	const crh_scene_desc *s = c->s;
	color weight = {1.0f, 1.0f, 1.0f, 1.0f};
	color finalColor = {0.0f, 0.0f, 0.0f, 1.0f};
	ray_t currentRay = *incidentRay;
	for (int depth = 0; depth < maxDepth; ++depth) {
		hit_t isect = getClosestIsect(c, &currentRay);
		if (isect.instIndex < 0) {
			finalColor = colorAdd(finalColor, colorMul(weight, sampleBsdf(c, s->background, &isect).col));
			break;
		}
		const crh_material *mat = &s->materials[isect.material];
		finalColor = colorAdd(finalColor, colorMul(weight, (color){mat->emission[0], mat->emission[1], mat->emission[2], mat->emission[3]}));
		const bsdf_sample sample = sampleBsdf(c, mat->bsdf, &isect);
		currentRay = (ray_t){isect.hitPoint, sample.out};
		const color attenuation = sample.col;
		float probability = 1.0f;
		if (depth >= 4) {
			probability = RMAX(attenuation.r, RMAX(attenuation.g, attenuation.b));
			if (getDimension(c) > probability) break;
		}
		weight = colorCoef(1.0f / probability, colorMul(attenuation, weight));
	}
	return finalColor;
}

/* ---- public entry points ----------------------------------------------------------------------- */

/* renderer.c:275-301, per pixel: running mean over passes in pass order */
static int g_sampler;   /* 0 Random, 1 Halton: what renderThread / renderThreadInteractive pass to initSampler */
void oracle_set_sampler(int halton) { g_sampler = halton ? 1 : 0; }

int oracle_render_region(const crh_scene_desc *scene, const crh_render_params *p, float *fb, crh_counters *counters_out, int threads) {
	if (!scene || !p || !fb) return CRH_ERR_INVALID;
	const int W = p->image_width, H = p->image_height;
	if (p->x0 < 0 || p->y0 < 0 || p->x1 > W || p->y1 > H || p->pass_count < 0) return CRH_ERR_INVALID;
	crh_counters total;
	memset(&total, 0, sizeof(total));
#ifdef _OPENMP
	if (threads > 0) omp_set_num_threads(threads);
#else
	(void)threads;
#endif
	#pragma omp parallel
	{
		ctx_t c;
		memset(&c, 0, sizeof(c));
		c.s = scene;
		c.halton = g_sampler;
		#pragma omp for schedule(dynamic, 1)
		for (int y = p->y1 - 1; y >= p->y0; --y) {
			for (int x = p->x0; x < p->x1; ++x) {
				uint32_t pixIdx = (uint32_t)(y * W + x);
				float *px = fb + ((size_t)x + (size_t)(H - (y + 1)) * (size_t)W) * 3;
				color output = {px[0], px[1], px[2], 1.0f};
				for (int pass = p->first_pass; pass < p->first_pass + p->pass_count; ++pass) {
					const int completedSamples = pass + 1;
					initSampler(&c, pass, p->max_passes, pixIdx);
					ray_t incidentRay = getCameraRay(&c, x, y);
					c.cnt.paths++;
					color sample = pathTrace(&c, &incidentRay, p->bounces);
					output = colorCoef((float)(completedSamples - 1), output);
					output = colorAdd(output, sample);
					float t = 1.0f / completedSamples;
					output = colorCoef(t, output);
				}
				px[0] = output.r; px[1] = output.g; px[2] = output.b;
			}
		}
		#pragma omp critical
		{
			total.paths += c.cnt.paths; total.rays += c.cnt.rays; total.node_tests += c.cnt.node_tests;
			total.tri_tests += c.cnt.tri_tests; total.inst_visits += c.cnt.inst_visits; total.inst_hits += c.cnt.inst_hits;
			total.sphere_tests += c.cnt.sphere_tests; total.tex_fetches += c.cnt.tex_fetches;
		}
	}
	if (counters_out) *counters_out = total;
	return CRH_OK;
}

int oracle_trace_rays(const crh_scene_desc *scene, const float *rays, uint64_t n, crh_hit *hits) {
	if (!scene || !rays || !hits) return CRH_ERR_INVALID;
	for (uint64_t i = 0; i < scene->instance_count; ++i)          /* volumes draw from the path's sampler inside the walk: caller rays have none */
		if (scene->instances[i].kind == CRH_INSTANCE_SPHERE_VOLUME || scene->instances[i].kind == CRH_INSTANCE_MESH_VOLUME) return CRH_ERR_UNSUPPORTED;
	#pragma omp parallel
	{
		ctx_t c;
		memset(&c, 0, sizeof(c));
		c.s = scene;
		#pragma omp for schedule(static)
		for (int64_t i = 0; i < (int64_t)n; ++i) {
			ray_t r = { {rays[6 * i + 0], rays[6 * i + 1], rays[6 * i + 2]}, {rays[6 * i + 3], rays[6 * i + 4], rays[6 * i + 5]} };
			hit_t h = getClosestIsect(&c, &r);
			crh_hit *o = &hits[i];
			memset(o, 0, sizeof(*o));
			o->inst = h.instIndex;
			o->node_tests = c.ray_node_tests; o->tri_tests = c.ray_tri_tests;
			if (h.instIndex < 0) { o->poly = -1; o->distance = h.distance; o->material = CRH_NODE_NONE; continue; }
			o->poly = h.polygon;
			o->distance = h.distance;
			o->uv[0] = h.uv.x; o->uv[1] = h.uv.y;
			o->point[0] = h.hitPoint.x; o->point[1] = h.hitPoint.y; o->point[2] = h.hitPoint.z;
			o->normal[0] = h.surfaceNormal.x; o->normal[1] = h.surfaceNormal.y; o->normal[2] = h.surfaceNormal.z;
			o->material = h.material;
		}
	}
	return CRH_OK;
}

/* color.h:60-84 + texture.c:18-22: (unsigned char)min(c * 255.0f, 255.0f) on the already y-flipped buffer */
void oracle_to_srgb8(const float *fb, int width, int height, uint8_t *rgb8) {
	const size_t n = (size_t)width * (size_t)height * 3;
	for (size_t i = 0; i < n; ++i) {
		float v = linearToSRGB(fb[i]);
		rgb8[i] = (unsigned char)RMIN(v * 255.0f, 255.0f);
	}
}

void oracle_sampler_draws(uint32_t pixel_index, int pass, int max_passes, int n, float *out) {
	ctx_t c;
	memset(&c, 0, sizeof(c));
	c.halton = g_sampler;
	initSampler(&c, pass, max_passes, pixel_index);
	for (int i = 0; i < n; ++i) out[i] = getDimension(&c);
}

void oracle_camera_ray(const crh_scene_desc *scene, int x, int y, int pass, int max_passes, float *out6) {
	ctx_t c;
	memset(&c, 0, sizeof(c));
	c.s = scene;
	initSampler(&c, pass, max_passes, (uint32_t)(y * scene->camera.width + x));
	ray_t r = getCameraRay(&c, x, y);
	out6[0] = r.start.x; out6[1] = r.start.y; out6[2] = r.start.z;
	out6[3] = r.direction.x; out6[4] = r.direction.y; out6[5] = r.direction.z;
}

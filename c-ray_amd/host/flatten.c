/*
 * flatten.c — walks the reference's `struct world` after loadScene() and emits the POD arrays of
 * crh_scene_desc (include/cray_hip.h). Host C, linked against the reference's own objects.
 *
 * What it reads (all produced by the UNTOUCHED reference loader, src/datatypes/scene.c:111-213):
 *   r->scene->instances / topLevel / meshes / spheres / camera / background   (scene.h:14-39)
 *   g_vertices / g_normals / g_textureCoords                                   (vertexbuffer.h:11-18)
 * BVHs are the ones the reference's builder made (bvh.c:132-316), copied node-for-node, so node
 * indices are bit-exact by construction. mesh->rayOffset / sphere->rayOffset are read here, i.e.
 * AFTER the TLAS build wrote them (instance.c:106,227).
 */
#include <sys/mman.h>
#include <stdint.h>
#include <pthread.h>
#include <stdio.h>
#include <time.h>
#include <stdlib.h>
#include <string.h>

#include "includes.h"
#include "datatypes/scene.h"
#include "datatypes/camera.h"
#include "datatypes/mesh.h"
#include "datatypes/sphere.h"
#include "datatypes/poly.h"
#include "datatypes/instance.h"
#include "datatypes/vertexbuffer.h"
#include "datatypes/material.h"
#include "datatypes/image/texture.h"
#include "renderer/renderer.h"
#include "nodes/bsdfnode.h"

#include "flatten.h"
#include "access/describe.h"

_Static_assert(sizeof(struct poly) == sizeof(crh_poly), "crh_poly must mirror struct poly (40 B)");
_Static_assert(sizeof(struct vector) == 12 && sizeof(struct coord) == 8, "vector/coord layout");

/* ---- tiny growable arrays + pointer->index map --------------------------------------------- */
struct ptrmap { const void **keys; uint32_t *vals; size_t cap, count; };

static size_t ptrhash(const void *p) { uintptr_t x = (uintptr_t)p; x ^= x >> 33; x *= 0xff51afd7ed558ccdULL; x ^= x >> 33; return (size_t)x; }

static void ptrmap_grow(struct ptrmap *m) {
	size_t ncap = m->cap ? m->cap * 2 : 256;
	const void **nk = calloc(ncap, sizeof(*nk));
	uint32_t *nv = calloc(ncap, sizeof(*nv));
	for (size_t i = 0; i < m->cap; ++i) {
		if (!m->keys[i]) continue;
		size_t h = ptrhash(m->keys[i]) & (ncap - 1);
		while (nk[h]) h = (h + 1) & (ncap - 1);
		nk[h] = m->keys[i]; nv[h] = m->vals[i];
	}
	free(m->keys); free(m->vals);
	m->keys = nk; m->vals = nv; m->cap = ncap;
}

static bool ptrmap_get(const struct ptrmap *m, const void *k, uint32_t *v) {
	if (!m->cap) return false;
	size_t h = ptrhash(k) & (m->cap - 1);
	while (m->keys[h]) {
		if (m->keys[h] == k) { *v = m->vals[h]; return true; }
		h = (h + 1) & (m->cap - 1);
	}
	return false;
}

static void ptrmap_put(struct ptrmap *m, const void *k, uint32_t v) {
	if ((m->count + 1) * 2 > m->cap) ptrmap_grow(m);
	size_t h = ptrhash(k) & (m->cap - 1);
	while (m->keys[h]) h = (h + 1) & (m->cap - 1);
	m->keys[h] = k; m->vals[h] = v; m->count++;
}

struct flat {
	crh_gnode *gnodes; size_t gnode_count, gnode_cap;
	crh_texture *textures; size_t texture_count, texture_cap;
	uint8_t *texdata; size_t texbytes, texcap;
	struct ptrmap nodemap, texmap;
	int error;
};

/* The big arrays of a flattened scene (BVH nodes, polygons, texture bytes: 70 MB for hdr.json) come as zero pages straight from mmap, on 2 MB boundaries, with
 * transparent huge pages asked for: written first touch by first touch by one thread, 4 KB pages cost 17 000 page faults — more than the copying (round 4; the same
 * finding as for the layout compile, scene_compile.h: PodBuf). A 64-byte header in front of the block tells big_free how to let it go. */
#define BIG_FROM ((size_t)4 << 20)
#define BIG_PAGE ((size_t)2 << 20)
#define BIG_MAGIC 0x4352482d42494721ull
struct big_hdr { uint64_t magic; void *map; size_t span; char pad[40]; };
/* (round 5, ADVICE r04: EVERY block carries the header — the small ones in front of a calloc block, map = NULL — so that big_free reads its own header whatever the
 * block's address; until then it told the two kinds apart by the pointer's 2 MB alignment, which another malloc can hand out too) */
static void *small_zalloc(size_t bytes) {
	char *raw = calloc((bytes ? bytes : 1) + sizeof(struct big_hdr), 1);
	if (!raw) return NULL;
	struct big_hdr *h = (struct big_hdr *)raw;
	h->magic = BIG_MAGIC; h->map = NULL; h->span = 0;
	return raw + sizeof(struct big_hdr);
}
static void *big_zalloc(size_t bytes) {
	if (bytes < BIG_FROM) return small_zalloc(bytes);
	const size_t span = ((bytes + BIG_PAGE - 1) & ~(BIG_PAGE - 1)) + BIG_PAGE;
	char *map = mmap(NULL, span, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
	if (map == MAP_FAILED) return small_zalloc(bytes);
	char *data = (char *)(((uintptr_t)map + sizeof(struct big_hdr) + BIG_PAGE - 1) & ~(uintptr_t)(BIG_PAGE - 1));
	(void)madvise(data, span - (size_t)(data - map), MADV_HUGEPAGE);
	struct big_hdr *h = (struct big_hdr *)(data - sizeof(struct big_hdr));
	h->magic = BIG_MAGIC; h->map = map; h->span = span;
	return data;
}
static void big_free(const void *p) {
	if (!p) return;
	struct big_hdr *h = (struct big_hdr *)((char *)p - sizeof(struct big_hdr));          /* (every block big_zalloc returns has one) */
	if (h->magic != BIG_MAGIC) return;          /* not ours (or freed twice): leak rather than guess */
	h->magic = 0;
	if (h->map) munmap(h->map, h->span); else free(h);
}

/* fn(ctx, begin, end) over [0, n) on up to CRH_FLATTEN_THREADS threads (the calling one included; default: that one alone, see below): the flattener's big loops
 * are copies — 60 MB for hdr.json */
#define FLAT_THREADS 8
#define FLAT_THREADS_DEFAULT 1      /* measured (profiles/r04z4_flatten_threads.log): 8 threads flatten hdr.json in 5 ms instead of 8, and the upload that follows takes 11 OR 20 ms
                                    * instead of 12 — the pages are first touched wherever the helper threads happen to run; CRH_FLATTEN_THREADS=n for a host that pins its threads */
struct par_job { void (*fn)(void *, size_t, size_t); void *ctx; size_t b, e; };
static void *par_run(void *arg) { struct par_job *j = arg; j->fn(j->ctx, j->b, j->e); return NULL; }
static void parallel_ranges(size_t n, size_t grain, void (*fn)(void *, size_t, size_t), void *ctx) {
	size_t parts = n / (grain ? grain : 1);
	static int threads = 0;
	if (!threads) { const char *e = getenv("CRH_FLATTEN_THREADS"); threads = e && atoi(e) > 0 ? (atoi(e) > FLAT_THREADS ? FLAT_THREADS : atoi(e)) : FLAT_THREADS_DEFAULT; }
	if (parts > (size_t)threads) parts = (size_t)threads;
	if (parts <= 1) { fn(ctx, 0, n); return; }
	pthread_t th[FLAT_THREADS];
	struct par_job job[FLAT_THREADS];
	int live[FLAT_THREADS];
	for (size_t p = 0; p < parts; ++p) {
		job[p] = (struct par_job){fn, ctx, n * p / parts, n * (p + 1) / parts};
		live[p] = p > 0 && pthread_create(&th[p], NULL, par_run, &job[p]) == 0;
	}
	par_run(&job[0]);
	for (size_t p = 1; p < parts; ++p) { if (live[p]) pthread_join(th[p], NULL); else par_run(&job[p]); }
}
struct copy_job { char *dst; const char *src; };
static void copy_range(void *ctx, size_t b, size_t e) { const struct copy_job *c = ctx; memcpy(c->dst + b, c->src + b, e - b); }
static void big_memcpy(void *dst, const void *src, size_t bytes) {
	struct copy_job c = {dst, src};
	parallel_ranges(bytes, (size_t)1 << 20, copy_range, &c);
}
struct bits_job { crh_bvh_node *nodes; crh_poly *polys; };
/* (an inner node's primCount is never written by the reference's builders — bvh.c:235 sets isLeaf only — and nothing reads it: whatever malloc left there is cleared too) */
static void node_bits_range(void *ctx, size_t b, size_t e) { struct bits_job *j = ctx; for (size_t i = b; i < e; ++i) j->nodes[i].count_leaf &= ((j->nodes[i].count_leaf >> 30) & 1u) ? 0x7FFFFFFFu : 0u; }
static void poly_bits_range(void *ctx, size_t b, size_t e) { struct bits_job *j = ctx; for (size_t i = b; i < e; ++i) j->polys[i].bits &= 0xFF07FFFFu; }
/* the vertex buffers in two passes: which slots does a polygon name (one byte per slot; several polygons, and several threads, store the same 1), then the named slots
 * copied in index order — instead of every polygon copying its nine slots to wherever they lie */
struct vert_job { const crh_poly *polys; float *verts, *norms, *texs; uint8_t *usedV, *usedN, *usedT; };
static void mark_range(void *ctx, size_t b, size_t e) {
	const struct vert_job *j = ctx;
	for (size_t p = b; p < e; ++p)
		for (int k = 0; k < 3; ++k) {
			int vi = j->polys[p].v[k], ni = j->polys[p].n[k], ti = j->polys[p].t[k];
			if (vi >= 0 && vi < vertexCount) j->usedV[vi] = 1;
			if (ni >= 0 && ni < normalCount) j->usedN[ni] = 1;
			if (ti >= 0 && ti < textureCount) j->usedT[ti] = 1;
		}
}
static void copy_verts_range(void *ctx, size_t b, size_t e) { const struct vert_job *j = ctx; for (size_t i = b; i < e; ++i) if (j->usedV[i]) memcpy(j->verts + 3 * i, &g_vertices[i], 12); }
static void copy_norms_range(void *ctx, size_t b, size_t e) { const struct vert_job *j = ctx; for (size_t i = b; i < e; ++i) if (j->usedN[i]) memcpy(j->norms + 3 * i, &g_normals[i], 12); }
static void copy_texs_range(void *ctx, size_t b, size_t e) { const struct vert_job *j = ctx; for (size_t i = b; i < e; ++i) if (j->usedT[i]) memcpy(j->texs + 2 * i, &g_textureCoords[i], 8); }

static uint32_t add_texture(struct flat *f, const struct texture *t) {
	if (!t) return CRH_NODE_NONE;
	uint32_t idx;
	if (ptrmap_get(&f->texmap, t, &idx)) return idx;
	if (f->texture_count == f->texture_cap) {
		f->texture_cap = f->texture_cap ? f->texture_cap * 2 : 16;
		f->textures = realloc(f->textures, f->texture_cap * sizeof(*f->textures));
	}
	size_t unit = t->precision == float_p ? sizeof(float) : 1;
	size_t bytes = t->width * t->height * t->channels * unit;
	size_t off = (f->texbytes + 15) & ~(size_t)15;
	if (off + bytes > f->texcap) {
		f->texcap = (off + bytes) * 2;
		uint8_t *grown = big_zalloc(f->texcap);
		if (f->texbytes) memcpy(grown, f->texdata, f->texbytes);
		big_free(f->texdata);
		f->texdata = grown;
	}
	memset(f->texdata + f->texbytes, 0, off - f->texbytes);
	big_memcpy(f->texdata + off, t->data.byte_p, bytes);
	f->texbytes = off + bytes;
	idx = (uint32_t)f->texture_count++;
	f->textures[idx] = (crh_texture){
		.offset = off, .width = (uint32_t)t->width, .height = (uint32_t)t->height, .channels = (uint32_t)t->channels,
		.is_float = t->precision == float_p, .has_alpha = t->hasAlpha ? 1u : 0u
	};
	ptrmap_put(&f->texmap, t, idx);
	return idx;
}

static bool describe(const void *node, int cls, struct crh_node_desc *d) {
	memset(d, 0, sizeof(*d));
#define TRY(name) if (crh_describe_##name(node, d)) return true;
	switch (cls) {
		case CRH_CLS_BSDF:   CRH_DESCRIBERS_BSDF(TRY)   break;
		case CRH_CLS_COLOR:  CRH_DESCRIBERS_COLOR(TRY)  break;
		case CRH_CLS_VALUE:  CRH_DESCRIBERS_VALUE(TRY)  break;
		case CRH_CLS_VECTOR: CRH_DESCRIBERS_VECTOR(TRY) break;
		default: break;
	}
#undef TRY
	return false;
}

/* Post-order: children first, so child indices are always smaller than their parent's. */
static uint32_t add_node(struct flat *f, const void *node, int cls) {
	if (!node) return CRH_NODE_NONE;
	uint32_t idx;
	if (ptrmap_get(&f->nodemap, node, &idx)) return idx;
	struct crh_node_desc d;
	if (!describe(node, cls, &d)) {
		fprintf(stderr, "crh_flatten: unrecognised node %p of class %d\n", node, cls);
		f->error = CRH_ERR_UNSUPPORTED;
		return CRH_NODE_NONE;
	}
	crh_gnode g;
	memset(&g, 0, sizeof(g));
	g.kind = d.kind;
	g.a = add_node(f, d.child[0], d.cls[0]);
	g.b = add_node(f, d.child[1], d.cls[1]);
	g.c = add_node(f, d.child[2], d.cls[2]);
	memcpy(g.f, d.f, sizeof(g.f));
	if (d.kind == CRH_COLOR_IMAGE) { g.a = add_texture(f, d.tex); g.b = d.u; }
	if (d.kind == CRH_VALUE_MATH || d.kind == CRH_VEC_VECMATH) g.c = d.u;
	if (f->gnode_count == f->gnode_cap) {
		f->gnode_cap = f->gnode_cap ? f->gnode_cap * 2 : 64;
		f->gnodes = realloc(f->gnodes, f->gnode_cap * sizeof(*f->gnodes));
	}
	idx = (uint32_t)f->gnode_count++;
	f->gnodes[idx] = g;
	ptrmap_put(&f->nodemap, node, idx);
	return idx;
}

static void copy_rows(float dst[12], const struct matrix4x4 *m) {
	for (int r = 0; r < 3; ++r) for (int c = 0; c < 4; ++c) dst[r * 4 + c] = m->mtx[r][c];
}

static crh_material flat_material(struct flat *f, const struct material *m) {
	crh_material out;
	memset(&out, 0, sizeof(out));
	out.emission[0] = m->emission.red; out.emission[1] = m->emission.green;
	out.emission[2] = m->emission.blue; out.emission[3] = m->emission.alpha;
	out.ior = m->IOR;
	out.bsdf = add_node(f, m->bsdf, CRH_CLS_BSDF);
	return out;
}

int crh_flatten_world(const struct renderer *r, crh_scene_desc *out) {
	if (!r || !r->scene || !out) return CRH_ERR_INVALID;
	const struct world *w = r->scene;
	struct flat f;
	memset(&f, 0, sizeof(f));
	memset(out, 0, sizeof(*out));
	out->struct_size = sizeof(*out);
	out->abi_version = CRH_SCENE_VERSION;
	struct timespec tr0;
	clock_gettime(CLOCK_MONOTONIC, &tr0);
	double trAt[5] = {0, 0, 0, 0, 0};
#define TRACE_LAP(i) do { struct timespec t_; clock_gettime(CLOCK_MONOTONIC, &t_); trAt[i] = (t_.tv_sec - tr0.tv_sec) * 1e3 + (t_.tv_nsec - tr0.tv_nsec) / 1e6; } while (0)

	/* --- BVHs: every BLAS, then the TLAS, concatenated --- */
	size_t totalNodes = crh_access_bvh_node_count(w->topLevel), totalPrims = (size_t)w->instanceCount, totalPolys = 0, totalMats = (size_t)w->sphereCount;
	for (int m = 0; m < w->meshCount; ++m) {
		totalNodes += crh_access_bvh_node_count(w->meshes[m].bvh);
		totalPrims += (size_t)w->meshes[m].polyCount;
		totalPolys += (size_t)w->meshes[m].polyCount;
		totalMats += (size_t)w->meshes[m].materialCount;
	}
	crh_bvh_node *nodes = big_zalloc((totalNodes ? totalNodes : 1) * sizeof(*nodes));
	int32_t *prims = big_zalloc((totalPrims ? totalPrims : 1) * sizeof(*prims));
	crh_poly *polys = big_zalloc((totalPolys ? totalPolys : 1) * sizeof(*polys));
	crh_mesh *meshes = calloc(w->meshCount ? w->meshCount : 1, sizeof(*meshes));
	crh_sphere *spheres = calloc(w->sphereCount ? w->sphereCount : 1, sizeof(*spheres));
	crh_material *materials = calloc(totalMats ? totalMats : 1, sizeof(*materials));
	crh_instance *instances = calloc(w->instanceCount ? w->instanceCount : 1, sizeof(*instances));

	size_t nodeAt = 0, primAt = 0, polyAt = 0, matAt = 0;
	for (int m = 0; m < w->meshCount; ++m) {
		const struct mesh *mesh = &w->meshes[m];
		unsigned nc = crh_access_bvh_node_count(mesh->bvh);
		crh_mesh *fm = &meshes[m];
		fm->node_base = (uint32_t)nodeAt; fm->node_count = nc;
		fm->prim_base = (uint32_t)primAt;
		fm->poly_base = (uint32_t)polyAt; fm->poly_count = (uint32_t)mesh->polyCount;
		fm->material_base = (uint32_t)matAt; fm->material_count = (uint32_t)mesh->materialCount;
		fm->texcoord_count = (uint32_t)mesh->textureCoordCount;
		fm->ray_offset = mesh->rayOffset;
		if (nc) {
			big_memcpy(nodes + nodeAt, crh_access_bvh_nodes(mesh->bvh), nc * sizeof(*nodes));
			big_memcpy(prims + primAt, crh_access_bvh_prims(mesh->bvh), (size_t)mesh->polyCount * sizeof(*prims));
		}
		if (mesh->polyCount) big_memcpy(polys + polyAt, mesh->polygons, (size_t)mesh->polyCount * sizeof(*polys));
		for (int k = 0; k < mesh->materialCount; ++k) materials[matAt + k] = flat_material(&f, &mesh->materials[k]);
		nodeAt += nc; primAt += (size_t)mesh->polyCount; polyAt += (size_t)mesh->polyCount; matAt += (size_t)mesh->materialCount;
	}
	TRACE_LAP(0);
	/* bvhNode's padding bits (count_leaf bit 31) and poly's padding bits are indeterminate in the
	 * reference; clear them so that blobs are reproducible. */
	{
		struct bits_job bj = {nodes, polys};
		parallel_ranges(nodeAt, (size_t)1 << 16, node_bits_range, &bj);
		parallel_ranges(polyAt, (size_t)1 << 16, poly_bits_range, &bj);
	}

	TRACE_LAP(1);
	out->tlas_node_base = (uint32_t)nodeAt;
	out->tlas_node_count = crh_access_bvh_node_count(w->topLevel);
	out->tlas_prim_base = (uint32_t)primAt;
	out->tlas_prim_count = out->tlas_node_count ? (uint32_t)w->instanceCount : 0;
	if (out->tlas_node_count) {
		memcpy(nodes + nodeAt, crh_access_bvh_nodes(w->topLevel), out->tlas_node_count * sizeof(*nodes));
		memcpy(prims + primAt, crh_access_bvh_prims(w->topLevel), (size_t)w->instanceCount * sizeof(*prims));
		{ struct bits_job tj = {nodes, polys}; node_bits_range(&tj, nodeAt, nodeAt + out->tlas_node_count); }
		nodeAt += out->tlas_node_count; primAt += (size_t)w->instanceCount;
	}

	for (int s = 0; s < w->sphereCount; ++s) {
		spheres[s].radius = w->spheres[s].radius;
		spheres[s].ray_offset = w->spheres[s].rayOffset;
		spheres[s].material = (uint32_t)matAt;
		materials[matAt++] = flat_material(&f, &w->spheres[s].material);
	}

	for (int i = 0; i < w->instanceCount; ++i) {
		const struct instance *inst = &w->instances[i];
		crh_instance *fi = &instances[i];
		copy_rows(fi->Ainv, &inst->composite.Ainv);
		copy_rows(fi->A, &inst->composite.A);
		int kind = crh_access_instance_kind(inst);
		const void *object = crh_access_instance_object(inst, &fi->density);
		if (kind == 0 || kind == 2) {
			fi->kind = kind == 0 ? CRH_INSTANCE_SPHERE : CRH_INSTANCE_SPHERE_VOLUME;
			fi->object = (uint32_t)((const struct sphere *)object - w->spheres);
		} else if (kind == 1 || kind == 3) {
			fi->kind = kind == 1 ? CRH_INSTANCE_MESH : CRH_INSTANCE_MESH_VOLUME;
			fi->object = (uint32_t)((const struct mesh *)object - w->meshes);
		} else {
			fprintf(stderr, "crh_flatten: instance %d has an unknown intersect function\n", i);
			f.error = CRH_ERR_UNSUPPORTED;
		}
	}

	out->background = add_node(&f, w->background, CRH_CLS_BSDF);

	TRACE_LAP(2);
	/* --- global vertex buffers; slots no polygon references are zeroed (the loader over-allocates:
	 * wavefront.c:148 counts every line starting with 'v') so the blob is deterministic --- */
	float *verts = big_zalloc((size_t)(vertexCount > 0 ? vertexCount : 1) * 3 * sizeof(float));
	float *norms = big_zalloc((size_t)(normalCount > 0 ? normalCount : 1) * 3 * sizeof(float));
	float *texs = big_zalloc((size_t)(textureCount > 0 ? textureCount : 1) * 2 * sizeof(float));
	{
		struct vert_job vj = {polys, verts, norms, texs, calloc((size_t)(vertexCount > 0 ? vertexCount : 1), 1), calloc((size_t)(normalCount > 0 ? normalCount : 1), 1),
		                      calloc((size_t)(textureCount > 0 ? textureCount : 1), 1)};
		parallel_ranges(polyAt, (size_t)1 << 15, mark_range, &vj);
		parallel_ranges((size_t)(vertexCount > 0 ? vertexCount : 0), (size_t)1 << 15, copy_verts_range, &vj);
		parallel_ranges((size_t)(normalCount > 0 ? normalCount : 0), (size_t)1 << 15, copy_norms_range, &vj);
		parallel_ranges((size_t)(textureCount > 0 ? textureCount : 0), (size_t)1 << 15, copy_texs_range, &vj);
		free(vj.usedV); free(vj.usedN); free(vj.usedT);
	}

	TRACE_LAP(3);
	if (getenv("CRH_TRACE_UPLOAD"))
		fprintf(stderr, "crh_flatten_world trace: BVH / polygon copies %.1f ms, padding bits %.1f ms, instances + node graphs + textures %.1f ms, vertex buffers %.1f ms\n",
				trAt[0], trAt[1] - trAt[0], trAt[2] - trAt[1], trAt[3] - trAt[2]);
	const struct camera *cam = w->camera;
	crh_camera *fc = &out->camera;
	fc->right[0] = cam->right.x; fc->right[1] = cam->right.y; fc->right[2] = cam->right.z;
	fc->up[0] = cam->up.x; fc->up[1] = cam->up.y; fc->up[2] = cam->up.z;
	fc->forward[0] = cam->forward.x; fc->forward[1] = cam->forward.y; fc->forward[2] = cam->forward.z;
	fc->sensor[0] = cam->sensorSize.x; fc->sensor[1] = cam->sensorSize.y;
	fc->aperture = cam->aperture;
	fc->focal_distance = cam->focalDistance;
	fc->width = cam->width; fc->height = cam->height;
	copy_rows(fc->A, &cam->composite.A);

	out->nodes = nodes;               out->node_count = nodeAt;
	out->prim_indices = prims;        out->prim_index_count = primAt;
	out->polys = polys;               out->poly_count = polyAt;
	out->vertices = verts;            out->vertex_count = (uint64_t)(vertexCount > 0 ? vertexCount : 0);
	out->normals = norms;             out->normal_count = (uint64_t)(normalCount > 0 ? normalCount : 0);
	out->texcoords = texs;            out->texcoord_count = (uint64_t)(textureCount > 0 ? textureCount : 0);
	out->instances = instances;       out->instance_count = (uint64_t)w->instanceCount;
	out->meshes = meshes;             out->mesh_count = (uint64_t)w->meshCount;
	out->spheres = spheres;           out->sphere_count = (uint64_t)w->sphereCount;
	out->materials = materials;       out->material_count = matAt;
	out->gnodes = f.gnodes;           out->gnode_count = f.gnode_count;
	out->textures = f.textures;       out->texture_count = f.texture_count;
	out->texture_data = f.texdata;    out->texture_bytes = f.texbytes;
	free(f.nodemap.keys); free(f.nodemap.vals);
	free(f.texmap.keys); free(f.texmap.vals);
	if (f.error) { crh_flatten_free(out); return f.error; }
	return CRH_OK;
}

void crh_flatten_free(crh_scene_desc *d) {
	if (!d) return;
	big_free(d->nodes); big_free(d->prim_indices); big_free(d->polys);
	big_free(d->vertices); big_free(d->normals); big_free(d->texcoords);
	free((void *)d->instances); free((void *)d->meshes); free((void *)d->spheres);
	free((void *)d->materials); free((void *)d->gnodes); free((void *)d->textures);
	big_free(d->texture_data);
	memset(d, 0, sizeof(*d));
}

crh_blob_prefs crh_flatten_prefs(const struct renderer *r) {
	crh_blob_prefs p;
	memset(&p, 0, sizeof(p));
	p.image_width = (int32_t)r->prefs.imageWidth;   p.image_height = (int32_t)r->prefs.imageHeight;
	p.sample_count = r->prefs.sampleCount;           p.bounces = r->prefs.bounces;
	p.tile_width = (int32_t)r->prefs.tileWidth;      p.tile_height = (int32_t)r->prefs.tileHeight;
	p.tile_order = (int32_t)r->prefs.tileOrder;
	return p;
}

/*
 * scene_blob.c — flat-file (de)serialisation of one crh_scene_desc.
 *
 * The reference has no such format: its scene lives in `struct world` (src/datatypes/scene.h:14-39)
 * plus the global vertex buffers (src/datatypes/vertexbuffer.h:11-18), produced by the JSON loader.
 * The blob exists so that the flattened scene can travel to machines that have neither the reference
 * loader nor its assets (the GPU box); it is written by the flattener (flatten.c / crh-flatten) and
 * read by the C-ABI (crh_blob_load), the tests, bench.py and the CPU oracle.
 *
 * Layout (little-endian):
 *   blob_header | section table (SEC_COUNT entries) | 16-byte aligned section payloads
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "cray_hip.h"

#define BLOB_MAGIC "CRHSCN01"

enum {
	SEC_NODES = 0, SEC_PRIMS, SEC_POLYS, SEC_VERTS, SEC_NORMALS, SEC_TEXCOORDS, SEC_INSTANCES,
	SEC_MESHES, SEC_SPHERES, SEC_MATERIALS, SEC_GNODES, SEC_TEXTURES, SEC_TEXDATA, SEC_COUNT
};

struct blob_section {
	uint32_t id;
	uint32_t elem_size;
	uint64_t count;
	uint64_t offset;
};

struct blob_header {
	char     magic[8];
	uint32_t abi_version;
	uint32_t section_count;
	uint32_t tlas_node_base, tlas_node_count, tlas_prim_base, tlas_prim_count;
	uint32_t background;
	uint32_t pad;
	crh_camera     camera;
	crh_blob_prefs prefs;
};

struct blob_owner {
	crh_scene_desc desc;   /* must be first: crh_blob_free() receives &owner->desc */
	void          *buffer;
};

static uint64_t align16(uint64_t v) { return (v + 15u) & ~(uint64_t)15u; }

static void fill_sections(const crh_scene_desc *s, struct blob_section *sec, const void **ptr) {
#define SEC(ID, P, ES, N) do { sec[ID].id = ID; sec[ID].elem_size = (uint32_t)(ES); sec[ID].count = (N); ptr[ID] = (P); } while (0)
	SEC(SEC_NODES,     s->nodes,        sizeof(crh_bvh_node), s->node_count);
	SEC(SEC_PRIMS,     s->prim_indices, sizeof(int32_t),      s->prim_index_count);
	SEC(SEC_POLYS,     s->polys,        sizeof(crh_poly),     s->poly_count);
	SEC(SEC_VERTS,     s->vertices,     3 * sizeof(float),    s->vertex_count);
	SEC(SEC_NORMALS,   s->normals,      3 * sizeof(float),    s->normal_count);
	SEC(SEC_TEXCOORDS, s->texcoords,    2 * sizeof(float),    s->texcoord_count);
	SEC(SEC_INSTANCES, s->instances,    sizeof(crh_instance), s->instance_count);
	SEC(SEC_MESHES,    s->meshes,       sizeof(crh_mesh),     s->mesh_count);
	SEC(SEC_SPHERES,   s->spheres,      sizeof(crh_sphere),   s->sphere_count);
	SEC(SEC_MATERIALS, s->materials,    sizeof(crh_material), s->material_count);
	SEC(SEC_GNODES,    s->gnodes,       sizeof(crh_gnode),    s->gnode_count);
	SEC(SEC_TEXTURES,  s->textures,     sizeof(crh_texture),  s->texture_count);
	SEC(SEC_TEXDATA,   s->texture_data, 1,                    s->texture_bytes);
#undef SEC
}

int crh_blob_save(const char *path, const crh_scene_desc *scene, const crh_blob_prefs *prefs) {
	if (!path || !scene || scene->struct_size != sizeof(*scene)) return CRH_ERR_INVALID;
	struct blob_header h;
	memset(&h, 0, sizeof(h));
	memcpy(h.magic, BLOB_MAGIC, 8);
	h.abi_version = CRH_SCENE_VERSION;
	h.section_count = SEC_COUNT;
	h.tlas_node_base = scene->tlas_node_base;  h.tlas_node_count = scene->tlas_node_count;
	h.tlas_prim_base = scene->tlas_prim_base;  h.tlas_prim_count = scene->tlas_prim_count;
	h.background = scene->background;
	h.camera = scene->camera;
	if (prefs) h.prefs = *prefs;

	struct blob_section sec[SEC_COUNT];
	const void *ptr[SEC_COUNT];
	memset(sec, 0, sizeof(sec));
	fill_sections(scene, sec, ptr);
	uint64_t off = align16(sizeof(h) + sizeof(sec));
	for (int i = 0; i < SEC_COUNT; ++i) {
		sec[i].offset = off;
		off = align16(off + sec[i].count * sec[i].elem_size);
	}

	FILE *f = fopen(path, "wb");
	if (!f) return CRH_ERR_IO;
	static const char zeros[16] = {0};
	int ok = fwrite(&h, sizeof(h), 1, f) == 1 && fwrite(sec, sizeof(sec), 1, f) == 1;
	uint64_t pos = sizeof(h) + sizeof(sec);
	for (int i = 0; ok && i < SEC_COUNT; ++i) {
		if (pos < sec[i].offset) { ok = fwrite(zeros, 1, sec[i].offset - pos, f) == sec[i].offset - pos; pos = sec[i].offset; }
		uint64_t bytes = sec[i].count * sec[i].elem_size;
		if (ok && bytes) { ok = fwrite(ptr[i], 1, bytes, f) == bytes; pos += bytes; }
	}
	if (ok && (pos & 15u)) ok = fwrite(zeros, 1, 16 - (pos & 15u), f) == 16 - (pos & 15u);
	ok = (fclose(f) == 0) && ok;
	return ok ? CRH_OK : CRH_ERR_IO;
}

int crh_blob_load(const char *path, crh_scene_desc **scene_out, crh_blob_prefs *prefs_out) {
	if (!path || !scene_out) return CRH_ERR_INVALID;
	*scene_out = NULL;
	FILE *f = fopen(path, "rb");
	if (!f) return CRH_ERR_IO;
	if (fseek(f, 0, SEEK_END) != 0) { fclose(f); return CRH_ERR_IO; }
	long size = ftell(f);
	rewind(f);
	if (size < 0 || size < (long)(sizeof(struct blob_header) + SEC_COUNT * sizeof(struct blob_section))) { fclose(f); return CRH_ERR_IO; }
	char *buf = malloc((size_t)size);
	if (!buf) { fclose(f); return CRH_ERR_NOMEM; }
	if (fread(buf, 1, (size_t)size, f) != (size_t)size) { fclose(f); free(buf); return CRH_ERR_IO; }
	fclose(f);

	const struct blob_header *h = (const struct blob_header *)buf;
	if (memcmp(h->magic, BLOB_MAGIC, 8) != 0 || h->abi_version != CRH_SCENE_VERSION || h->section_count != SEC_COUNT) {
		free(buf);
		return CRH_ERR_INVALID;
	}
	const struct blob_section *sec = (const struct blob_section *)(buf + sizeof(*h));
	static const uint32_t elem[SEC_COUNT] = {
		sizeof(crh_bvh_node), sizeof(int32_t), sizeof(crh_poly), 12, 12, 8, sizeof(crh_instance), sizeof(crh_mesh),
		sizeof(crh_sphere), sizeof(crh_material), sizeof(crh_gnode), sizeof(crh_texture), 1
	};
	for (int i = 0; i < SEC_COUNT; ++i) {
		/* no wrap-around: offset inside the file and past the tables, 16-byte aligned (typed reads), count bounded by what is left */
		const uint64_t tables = sizeof(struct blob_header) + SEC_COUNT * sizeof(struct blob_section);
		if (sec[i].id != (uint32_t)i || sec[i].elem_size != elem[i] || sec[i].offset > (uint64_t)size || sec[i].offset < tables ||
			(sec[i].offset & 15u) || sec[i].count > ((uint64_t)size - sec[i].offset) / elem[i]) {
			free(buf);
			return CRH_ERR_INVALID;
		}
	}
	struct blob_owner *o = calloc(1, sizeof(*o));
	if (!o) { free(buf); return CRH_ERR_NOMEM; }
	o->buffer = buf;
	crh_scene_desc *s = &o->desc;
	s->struct_size = sizeof(*s);
	s->abi_version = CRH_SCENE_VERSION;
#define GET(ID, FIELD, TYPE, COUNT) do { s->FIELD = (const TYPE *)(buf + sec[ID].offset); s->COUNT = sec[ID].count; } while (0)
	GET(SEC_NODES,     nodes,        crh_bvh_node, node_count);
	GET(SEC_PRIMS,     prim_indices, int32_t,      prim_index_count);
	GET(SEC_POLYS,     polys,        crh_poly,     poly_count);
	GET(SEC_VERTS,     vertices,     float,        vertex_count);
	GET(SEC_NORMALS,   normals,      float,        normal_count);
	GET(SEC_TEXCOORDS, texcoords,    float,        texcoord_count);
	GET(SEC_INSTANCES, instances,    crh_instance, instance_count);
	GET(SEC_MESHES,    meshes,       crh_mesh,     mesh_count);
	GET(SEC_SPHERES,   spheres,      crh_sphere,   sphere_count);
	GET(SEC_MATERIALS, materials,    crh_material, material_count);
	GET(SEC_GNODES,    gnodes,       crh_gnode,    gnode_count);
	GET(SEC_TEXTURES,  textures,     crh_texture,  texture_count);
	GET(SEC_TEXDATA,   texture_data, uint8_t,      texture_bytes);
#undef GET
	s->tlas_node_base = h->tlas_node_base;  s->tlas_node_count = h->tlas_node_count;
	s->tlas_prim_base = h->tlas_prim_base;  s->tlas_prim_count = h->tlas_prim_count;
	s->background = h->background;
	s->camera = h->camera;
	if (prefs_out) *prefs_out = h->prefs;
	*scene_out = s;
	return CRH_OK;
}

void crh_blob_free(crh_scene_desc *scene) {
	if (!scene) return;
	struct blob_owner *o = (struct blob_owner *)scene;
	free(o->buffer);
	free(o);
}

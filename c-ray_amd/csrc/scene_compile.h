/*
 * scene_compile.h — host-side derivation of the device scene layout from a crh_scene_desc.
 *
 * Pure host C++ (no HIP): crh_scene_upload (cray_hip.hip) runs it and copies the arrays to HBM; the
 * CPU-only test tier (tests/emu) runs the same code so that the layout, the validation and the
 * node-program compiler are exercised without a GPU.
 *
 * What is derived (nothing changes WHICH node / triangle / instance a ray visits — BVH node indices stay
 * the reference's, bvh.c:132-316):
 *   nodes   every BVH shifted by one slot so that a child pair (first, first+1 — always (odd, even),
 *           bvh.c:221-223) is one 64-byte aligned record; `first` of inner nodes rewritten to the device
 *           index of the left child, `first` of leaves to the absolute prim slot.
 *   tris    per BLAS prim slot (leaf order) the prepared triangle v0, e1 = v0-v1, e2 = v2-v0, n = e1 x e2
 *           (poly.c:20-22, same fp32 operations, this TU is built with -ffp-contract=off): 48 B instead of
 *           4 B index + 40 B poly + 3 x 12 B scattered vertices.
 *   shade   per BLAS prim slot the 64-B record finishing a hit needs (vertex normals or e1 x e2, texture coordinates,
 *           material index): one load instead of prim index -> polygon -> normals / texcoords -> mesh.
 *   instances   in TLAS leaf order (slot - tlas_prim_base indexes them directly), enter data in the first 64-B line.
 *   textures / texels   every texture expanded to f4 texels (see DTexture).
 *   bsdfs / consts / images / prog   the node graph: bsdf nodes 1:1, colour/value/vector sub-graphs
 *           compiled to constants, image fetches or short postfix programs (pure functions of the hit).
 */
#pragma once
#include <string>
#include <vector>
#include "pt_device.h"

namespace crh {

struct CompiledScene {
	std::vector<f4> nodes;
	std::vector<f4> tris;
	std::vector<DShadeTri> shade;
	std::vector<DInstance> instances;
	std::vector<DBsdf> bsdfs;
	std::vector<crh_material> materials;   /* scene materials + pad[0] = 1 when the material's bsdf graph reads the hit's uv */
	std::vector<f4> consts;
	std::vector<DImage> images;
	std::vector<DOp> prog;
	std::vector<DTexture> textures;
	std::vector<f4> texels;
	uint32_t tlas_root = 0, tlas_node_count = 0, tlas_prim_base = 0, background = 0, shade_classes = 0;
	uint32_t tlas_first = 0;    /* device index of TLAS node 1 (the TLAS nodes are contiguous: node i at tlas_first - 1 + i) */
	uint32_t max_stack = 0;     /* worst-case traversal stack entries (TLAS depth + saved TLAS state + deepest BLAS) */
	uint32_t max_add_depth = 0;
	bool has_volumes = false;   /* some instance is a sphere / mesh volume: walks draw from the path's sampler (no crh_trace_rays) */
	crh_camera camera;
};

/* Returns CRH_OK or a negative CRH_ERR_* with a message in `err`. */
int compile_scene(const crh_scene_desc *scene, CompiledScene &out, std::string &err);

}  // namespace crh

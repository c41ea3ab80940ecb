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
#include <cstdlib>
#include <cstring>
#include <sys/mman.h>
#include <functional>
#include <new>
#include <string>
#include <vector>
#include "pt_device.h"

namespace crh {

/* The four big arrays of a compiled scene (nodes, prepared triangles, shading records, texels: 150 MB for BASELINE configs[1]) live in plain heap blocks that grow
 * WITHOUT value-initialising what the compiler is about to overwrite anyway (std::vector::resize zero-fills: 190 MB of memset for the texels alone), and are
 * filled by several threads (scene_compile.cpp: parallelFor). Plain-old-data only. */
#define CRH_PODBUF_HUGE_PAGE ((size_t)2 << 20)
#define CRH_PODBUF_HUGE_FROM ((size_t)8 << 20)
template <class T> struct PodBuf {
	T *p = nullptr;
	size_t n = 0, cap = 0;
	PodBuf() = default;
	PodBuf(const PodBuf &) = delete;
	PodBuf &operator=(const PodBuf &) = delete;
	~PodBuf() { free(p); }
	T *data() { return p; }
	const T *data() const { return p; }
	size_t size() const { return n; }
	bool empty() const { return n == 0; }
	T &operator[](size_t i) { return p[i]; }
	const T &operator[](size_t i) const { return p[i]; }
	void reserve(size_t c) {
		if (c <= cap) return;
		size_t nc = cap ? cap : (c > 16 ? c : 16);          /* a first reservation is exact */
		while (nc < c) nc *= 2;
		T *q;
		if (nc * sizeof(T) >= CRH_PODBUF_HUGE_FROM) {
			/* big blocks sit on 2 MB boundaries and ask for transparent huge pages: a compile writes 150 MB for hdr.json, first touch by first touch — 37 000 page faults from
			 * sixteen threads of a process that has the GPU open, which is where half of its time went (round 4); 75 faults of 2 MB now. (The callers reserve the totals
			 * first: a block of this kind grows by copying.) */
			const size_t bytes = (nc * sizeof(T) + CRH_PODBUF_HUGE_PAGE - 1) & ~(size_t)(CRH_PODBUF_HUGE_PAGE - 1);
			void *m = nullptr;
			if (posix_memalign(&m, CRH_PODBUF_HUGE_PAGE, bytes) != 0 || !m) throw std::bad_alloc();
			(void)madvise(m, bytes, MADV_HUGEPAGE);
			q = (T *)m;
			if (n) memcpy((void *)q, (const void *)p, n * sizeof(T));
			free(p);
		} else {
			q = (T *)realloc((void *)p, nc * sizeof(T));
			if (!q) throw std::bad_alloc();
		}
		p = q; cap = nc;
	}
	void resize(size_t c) { reserve(c); n = c; }                                  /* new elements are NOT initialised */
	void resize(size_t c, const T &v) { const size_t o = n; resize(c); for (size_t i = o; i < c; ++i) p[i] = v; }
	void assign(size_t c, const T &v) { n = 0; resize(c, v); }
	void push_back(const T &v) { reserve(n + 1); p[n++] = v; }
};

struct CompiledScene {
	PodBuf<f4> nodes;
	PodBuf<f4> tris;
	PodBuf<DShadeTri> shade;
	std::vector<DInstance> instances;
	std::vector<DBsdf> bsdfs;
	std::vector<crh_material> materials;   /* scene materials + pad[0] = 1 when the material's bsdf graph reads the hit's uv */
	std::vector<f4> consts;
	std::vector<DImage> images;
	std::vector<DOp> prog;
	std::vector<DTexture> textures;
	PodBuf<f4> texels;
	uint32_t tlas_root = 0, tlas_node_count = 0, tlas_prim_base = 0, background = 0, shade_classes = 0;
	uint32_t tlas_first = 0;    /* device index of TLAS node 1 (the TLAS nodes are contiguous: node i at tlas_first - 1 + i) */
	uint32_t max_stack = 0;     /* worst-case traversal stack entries (TLAS depth + saved TLAS state + deepest BLAS) */
	uint32_t max_add_depth = 0;
	/* CRH_OPT_WALK = CRH_WALK_WIDE4 (round 5, an OPTION: the binary walk over `nodes` stays the contract): a derived 4-ary copy of every BVH with at least one inner
	 * node, collapsed from the reference's binary tree (the inner child with the largest surface is replaced by its two children until four children stand; the boxes are
	 * the binary nodes' own bits). A wide node = 4 x 32 B: {minx, maxx, miny, maxy} {minz, maxz, ref, 0}; ref = CRH_NONE for an unused slot (its box is never hit), the
	 * child wide node's offset from the start of `nodes` in 16-byte units (bit 31 clear), or a leaf: CRH_WREF_LEAF | count << 25 | first absolute prim slot.
	 * In the device allocation the array stands behind the triangles (sceneWideOffset). Empty when not asked for, or when the scene cannot be encoded (wide_refused). */
	PodBuf<f4> wide;
	uint32_t wide_tlas_root = 0;    /* ref of the top-level BVH's wide root (tlas_node_count > 1) */
	uint32_t wide_max_stack = 0;    /* worst-case stack entries of the wide walk: three per level */
	bool want_wide = false;         /* in: build `wide` */
	std::string wide_refused;       /* why `wide` is empty although it was asked for */
	bool has_volumes = false;   /* some instance is a sphere / mesh volume: walks draw from the path's sampler (no crh_trace_rays) */
	crh_camera camera;
};

/* where the arrays stand in the ONE device allocation the walk addresses with 32-bit offsets from `nodes` (crh_scene_upload; tests/emu builds the same block):
 * nodes, (256-byte aligned) prepared triangles, 96 bytes of padding, (128-byte aligned) the wide nodes */
inline size_t sceneNodeBytes(const CompiledScene &c) { return (c.nodes.size() * sizeof(f4) + 255u) & ~(size_t)255u; }
inline size_t sceneWideOffset(const CompiledScene &c) { return (sceneNodeBytes(c) + c.tris.size() * sizeof(f4) + 96u + 127u) & ~(size_t)127u; }

/* Returns CRH_OK or a negative CRH_ERR_* with a message in `err`. */
/* texelsReady (optional) is called — on the compiling thread — as soon as out.textures / out.texels are final, while the BVHs and triangles are still to be prepared: the
 * uploader starts the copy of the texels (nine tenths of a textured scene's bytes) beside the rest of the compile. */
int compile_scene(const crh_scene_desc *scene, CompiledScene &out, std::string &err, const std::function<void()> &texelsReady = std::function<void()>());

}  // namespace crh

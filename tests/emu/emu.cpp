/*
 * emu.cpp — HOST EMULATION of the device lane code (c-ray_amd/csrc/pt_device.h), TEST INFRASTRUCTURE ONLY.
 *
 * The CPU-only test tier has no GPU, so this builds the very same lane-level functions the HIP kernels
 * call (traverse / finishHit / sampleBsdf / renderLane ...) with g++ and runs them one lane at a time
 * over the device scene layout produced by the product's scene compiler (csrc/scene_compile.cpp).
 * Compiled with the oracle's flags it must match the oracle BIT FOR BIT (tests/test_emu_parity.py): that
 * pins the kernel logic (traversal order, RNG draw order, node programs, layout derivation) on the CPU;
 * the GPU tier then only has to absorb the libm differences. Never linked into libcray_hip.so and never
 * used by bench.py or the product.
 */
#include <cstring>
#include <string>
#include <vector>

#include "../../c-ray_amd/csrc/pt_device.h"
#include "../../c-ray_amd/csrc/scene_compile.h"

using namespace crh;

namespace {
struct ArrayStack {
	uint32_t e[160];
	uint32_t high = 0;
	void push(uint32_t i, uint32_t v) { e[i] = v; if (i + 1 > high) high = i + 1; }
	uint32_t pop(uint32_t i) { return e[i]; }
};
struct RegionWork {
	int x0, y0, x1, y1, x, y;
	bool next(int &ox, int &oy) {
		if (x0 >= x1 || y < y0) return false;
		ox = x; oy = y;
		if (++x == x1) { x = x0; --y; }
		return true;
	}
};
DScene make_dscene(const crh_scene_desc *s, const CompiledScene &c) {
	DScene d;
	memset(&d, 0, sizeof(d));
	d.nodes = c.nodes.data(); d.tris = c.tris.data(); d.prims = s->prim_indices; d.polys = s->polys;
	d.vertices = s->vertices; d.normals = s->normals; d.texcoords = s->texcoords;
	d.instances = c.instances.data(); d.meshes = s->meshes; d.materials = s->materials;
	d.bsdfs = c.bsdfs.data(); d.consts = c.consts.data(); d.images = c.images.data(); d.prog = c.prog.data();
	d.textures = s->textures; d.texdata = s->texture_data;
	d.tlas_root = c.tlas_root; d.tlas_node_count = c.tlas_node_count; d.tlas_prim_base = c.tlas_prim_base;
	d.background = c.background; d.camera = c.camera;
	return d;
}
std::string g_err;
}  // namespace

extern "C" {

const char *emu_last_error(void) { return g_err.c_str(); }

int emu_compile_check(const crh_scene_desc *scene, uint32_t *max_stack, uint32_t *prog_ops, uint32_t *consts) {
	CompiledScene c;
	int rc = compile_scene(scene, c, g_err);
	if (rc != CRH_OK) return rc;
	if (max_stack) *max_stack = c.max_stack;
	if (prog_ops) *prog_ops = (uint32_t)c.prog.size();
	if (consts) *consts = (uint32_t)c.consts.size();
	return CRH_OK;
}

int emu_render_region(const crh_scene_desc *scene, const crh_render_params *p, float *fb, crh_counters *out, uint32_t *stack_high) {
	CompiledScene c;
	int rc = compile_scene(scene, c, g_err);
	if (rc != CRH_OK) return rc;
	const DScene d = make_dscene(scene, c);
	crh_counters total;
	memset(&total, 0, sizeof(total));
	uint32_t high = 0;
	#pragma omp parallel
	{
		Counters cnt;
		memset(&cnt, 0, sizeof(cnt));
		crh_counters mine;
		memset(&mine, 0, sizeof(mine));
		uint32_t myHigh = 0;
		#pragma omp for schedule(dynamic, 1)
		for (int y = p->y1 - 1; y >= p->y0; --y) {
			ArrayStack stk;
			RegionWork w{p->x0, y, p->x1, y + 1, p->x0, y};
			renderLane(d, *p, stk, w, fb, cnt);
			if (stk.high > myHigh) myHigh = stk.high;
			mine.paths += cnt.paths; mine.rays += cnt.rays; mine.node_tests += cnt.node_tests; mine.tri_tests += cnt.tri_tests;
			mine.inst_visits += cnt.inst_visits; mine.inst_hits += cnt.inst_hits; mine.sphere_tests += cnt.sphere_tests; mine.tex_fetches += cnt.tex_fetches;
			memset(&cnt, 0, sizeof(cnt));
		}
		#pragma omp critical
		{
			total.paths += mine.paths; total.rays += mine.rays; total.node_tests += mine.node_tests; total.tri_tests += mine.tri_tests;
			total.inst_visits += mine.inst_visits; total.inst_hits += mine.inst_hits; total.sphere_tests += mine.sphere_tests; total.tex_fetches += mine.tex_fetches;
			if (myHigh > high) high = myHigh;
		}
	}
	if (out) *out = total;
	if (stack_high) *stack_high = high;
	return CRH_OK;
}

int emu_trace_rays(const crh_scene_desc *scene, const float *rays, uint64_t n, crh_hit *hits) {
	CompiledScene c;
	int rc = compile_scene(scene, c, g_err);
	if (rc != CRH_OK) return rc;
	const DScene d = make_dscene(scene, c);
	#pragma omp parallel for schedule(static)
	for (int64_t i = 0; i < (int64_t)n; ++i) {
		ArrayStack stk;
		Counters cnt;
		memset(&cnt, 0, sizeof(cnt));
		const v3 o{rays[6 * i], rays[6 * i + 1], rays[6 * i + 2]}, dd{rays[6 * i + 3], rays[6 * i + 4], rays[6 * i + 5]};
		TravHit h;
		traverse(d, stk, o, dd, h, cnt);
		crh_hit *oh = &hits[i];
		memset(oh, 0, sizeof(*oh));
		oh->inst = h.inst; oh->distance = h.t; oh->node_tests = cnt.node_tests; oh->tri_tests = cnt.tri_tests;
		if (h.inst < 0) { oh->poly = -1; oh->material = CRH_NODE_NONE; continue; }
		const HitInfo hi = finishHit(d, o, dd, h);
		oh->poly = hi.poly; oh->uv[0] = hi.uv.x; oh->uv[1] = hi.uv.y;
		oh->point[0] = hi.point.x; oh->point[1] = hi.point.y; oh->point[2] = hi.point.z;
		oh->normal[0] = hi.normal.x; oh->normal[1] = hi.normal.y; oh->normal[2] = hi.normal.z;
		oh->material = hi.material;
	}
	return CRH_OK;
}

}  // extern "C"

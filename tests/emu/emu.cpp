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
#include <algorithm>
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
	uint32_t pk[CRH_PARK_SLOTS];
	void park(int i, uint32_t v) { pk[i] = v; }
	uint32_t unpark(int i) { return pk[i]; }
};
/* CRH_OPT_WALK = CRH_WALK_WIDE4 in the lane emulation (emu_set_walk(1)): nodes, triangles and the wide nodes in ONE block, laid out as crh_scene_upload lays them out
 * (scene_compile.h: sceneWideOffset) — a wide reference is an offset from the start of the nodes */
int g_wide;
uint32_t g_stack_high;
typedef CountersT<2, true, true> WideCounters;
struct WideBlock {
	std::vector<char> mem;
	void build(const CompiledScene &c) {
		mem.assign(sceneWideOffset(c) + c.wide.size() * sizeof(f4) + 256, 0);
		memcpy(mem.data(), c.nodes.data(), c.nodes.size() * sizeof(f4));
		memcpy(mem.data() + sceneNodeBytes(c), c.tris.data(), c.tris.size() * sizeof(f4));
		memcpy(mem.data() + sceneWideOffset(c), c.wide.data(), c.wide.size() * sizeof(f4));
	}
};
DScene make_dscene(const crh_scene_desc *s, const CompiledScene &c, const WideBlock *wb = nullptr) {
	DScene d;
	memset(&d, 0, sizeof(d));
	d.nodes = c.nodes.data(); d.tris = c.tris.data(); d.prims = s->prim_indices; d.shade = c.shade.data();
	if (wb) { d.nodes = (const f4 *)wb->mem.data(); d.tris = (const f4 *)(wb->mem.data() + sceneNodeBytes(c)); }
	d.instances = c.instances.data(); d.materials = c.materials.data();
	d.bsdfs = c.bsdfs.data(); d.consts = c.consts.data(); d.images = c.images.data(); d.prog = c.prog.data();
	d.textures = c.textures.data(); d.texels = c.texels.data();
	d.tlas_root = c.tlas_root; d.tlas_node_count = c.tlas_node_count; d.tlas_prim_base = c.tlas_prim_base; d.tlas_first = c.tlas_first;
	d.material_count = (uint32_t)c.materials.size(); d.bsdf_count = (uint32_t)c.bsdfs.size(); d.const_count = (uint32_t)c.consts.size(); d.image_count = (uint32_t)c.images.size(); d.texture_count = (uint32_t)c.textures.size();
	d.background = c.background; d.camera = &c.camera;
	d.instance_count = (uint32_t)c.instances.size(); d.shade_classes = c.shade_classes;
	if (wb && c.tlas_node_count > 1) d.tlas_root = c.wide_tlas_root;
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

/* The kernel's schedule, one "wave" at a time: blocks of bw x bh pixels, chunks of `chunk` passes, 64 lanes
 * striding over the block's items, then the per-pixel fold. */
static int g_halton;
void emu_set_sampler(int halton) { g_halton = halton ? 1 : 0; }     /* 0 Random (renderThread), 1 Halton (renderThreadInteractive) */

void emu_set_walk(int wide) { g_wide = wide ? 1 : 0; }
uint32_t emu_stack_high(int reset) { const uint32_t h = g_stack_high; if (reset) g_stack_high = 0; return h; }     /* deepest stack of the emu_trace_rays walks since the last reset */              /* 0 the binary walk (the contract), 1 CRH_WALK_WIDE4 */

int emu_render_region(const crh_scene_desc *scene, const crh_render_params *p, float *fb, crh_counters *out, uint32_t *stack_high,
					  int bw, int bh, int chunk) {
	CompiledScene c;
	c.want_wide = g_wide != 0;
	int rc = compile_scene(scene, c, g_err);
	if (rc != CRH_OK) return rc;
	if (g_wide && c.wide.empty()) { g_err = "no wide copy: " + c.wide_refused; return CRH_ERR_UNSUPPORTED; }
	if (bw <= 0 || bh <= 0 || bw * bh > 256 || chunk <= 0) { g_err = "bad block shape"; return CRH_ERR_INVALID; }
	WideBlock wb;
	if (g_wide) wb.build(c);
	const DScene d = make_dscene(scene, c, g_wide ? &wb : nullptr);
	crh_counters total;
	memset(&total, 0, sizeof(total));
	uint32_t high = 0;
	const int W = p->x1 - p->x0, H = p->y1 - p->y0;
	const int nbx = (W + bw - 1) / bw, nby = (H + bh - 1) / bh;
	#pragma omp parallel
	{
		Counters cnt;
		memset(&cnt, 0, sizeof(cnt));
		WideCounters wcnt;
		memset(&wcnt, 0, sizeof(wcnt));
		crh_counters mine;
		memset(&mine, 0, sizeof(mine));
		uint32_t myHigh = 0;
		std::vector<float> stage((size_t)bw * bh * chunk * 3);
		#pragma omp for schedule(dynamic, 1)
		for (int blk = 0; blk < nbx * nby; ++blk) {
			BlockJob J;
			J.bw = bw; J.bh = bh;
			J.x0 = p->x0 + (blk % nbx) * bw; J.y0 = p->y0 + (blk / nbx) * bh;
			J.w = std::min(bw, p->x1 - J.x0); J.h = std::min(bh, p->y1 - J.y0);
			for (int c0 = p->first_pass; c0 < p->first_pass + p->pass_count; c0 += chunk) {
				J.passBegin = c0; J.passCount = std::min(chunk, p->first_pass + p->pass_count - c0);
				for (uint32_t lane = 0; lane < 64; ++lane) {
					ArrayStack stk;
					if (g_wide) { if (g_halton) renderItems<HaltonRng>(d, *p, stk, J, lane, 64u, stage.data(), wcnt); else renderItems<Rng>(d, *p, stk, J, lane, 64u, stage.data(), wcnt); }
					else if (g_halton) renderItems<HaltonRng>(d, *p, stk, J, lane, 64u, stage.data(), cnt);
					else renderItems<Rng>(d, *p, stk, J, lane, 64u, stage.data(), cnt);
					if (stk.high > myHigh) myHigh = stk.high;
				}
				for (uint32_t pix = 0; pix < (uint32_t)(bw * bh); ++pix) foldBlockPixel(*p, J, pix, stage.data(), fb);
			}
			if (g_wide) { memcpy(&cnt, &wcnt, sizeof(cnt)); memset(&wcnt, 0, sizeof(wcnt)); }
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
	for (uint64_t i = 0; scene && i < scene->instance_count; ++i)     /* like crh_trace_rays: caller rays have no sampler for a volume to draw from */
		if (scene->instances[i].kind == CRH_INSTANCE_SPHERE_VOLUME || scene->instances[i].kind == CRH_INSTANCE_MESH_VOLUME) { g_err = "volumes"; return CRH_ERR_UNSUPPORTED; }
	CompiledScene c;
	c.want_wide = g_wide != 0;
	int rc = compile_scene(scene, c, g_err);
	if (rc != CRH_OK) return rc;
	if (g_wide && c.wide.empty()) { g_err = "no wide copy: " + c.wide_refused; return CRH_ERR_UNSUPPORTED; }
	WideBlock wb;
	if (g_wide) wb.build(c);
	const DScene d = make_dscene(scene, c, g_wide ? &wb : nullptr);
	#pragma omp parallel for schedule(static)
	for (int64_t i = 0; i < (int64_t)n; ++i) {
		ArrayStack stk;
		Counters cnt;
		memset(&cnt, 0, sizeof(cnt));
		const v3 o{rays[6 * i], rays[6 * i + 1], rays[6 * i + 2]}, dd{rays[6 * i + 3], rays[6 * i + 4], rays[6 * i + 5]};
		TravHit h;
		if (g_wide) { WideCounters wc; memset(&wc, 0, sizeof(wc)); traverse(d, stk, o, dd, h, wc); memcpy(&cnt, &wc, sizeof(cnt)); }
		else traverse(d, stk, o, dd, h, cnt);          /* the render kernels' walk: degenerate slabs tested exactly (crh_trace_rays' CRH_TRACE_SLABS_EXACT) */
		if (stk.high > g_stack_high) {
			#pragma omp critical
			if (stk.high > g_stack_high) g_stack_high = stk.high;
		}
		crh_hit *oh = &hits[i];
		memset(oh, 0, sizeof(*oh));
		oh->inst = h.inst < 0 ? -1 : (int32_t)d.instances[h.inst].orig; oh->distance = h.t; oh->node_tests = cnt.node_tests; oh->tri_tests = cnt.tri_tests;
		if (h.inst < 0) { oh->poly = -1; oh->material = CRH_NODE_NONE; continue; }
		const HitInfo hi = finishHit<false>(d, o, dd, h);
		oh->poly = hitPoly(d, h); oh->uv[0] = hi.uv.x; oh->uv[1] = hi.uv.y;
		oh->point[0] = hi.point.x; oh->point[1] = hi.point.y; oh->point[2] = hi.point.z;
		oh->normal[0] = hi.normal.x; oh->normal[1] = hi.normal.y; oh->normal[2] = hi.normal.z;
		oh->material = hi.material;
	}
	return CRH_OK;
}

/* The scene compiler's shade classes (scheduling hint of the device kernel): out[orig instance index] = class, returns the number of classes. */
int emu_shade_classes(const crh_scene_desc *scene, uint32_t *out, uint64_t n) {
	CompiledScene c;
	int rc = compile_scene(scene, c, g_err);
	if (rc != CRH_OK) return rc;
	for (const DInstance &d : c.instances) if (d.orig < n) out[d.orig] = CRH_DINST_CLASS(d.kind);
	return (int)c.shade_classes;
}

}  // extern "C"

/* debug: the most expensive single path (by node tests) in a region: out = {x, y, pass, node_tests, tri_tests, rays} */
extern "C" int emu_find_heaviest_path(const crh_scene_desc *scene, const crh_render_params *p, uint64_t *out6) {
	CompiledScene c;
	int rc = compile_scene(scene, c, g_err);
	if (rc != CRH_OK) return rc;
	const DScene d = make_dscene(scene, c);
	uint64_t best[6] = {0, 0, 0, 0, 0, 0};
	#pragma omp parallel for schedule(dynamic, 1)
	for (int y = p->y0; y < p->y1; ++y) {
		uint64_t mine[6] = {0, 0, 0, 0, 0, 0};
		std::vector<float> stage(3);
		for (int x = p->x0; x < p->x1; ++x)
			for (int pass = p->first_pass; pass < p->first_pass + p->pass_count; ++pass) {
				BlockJob J{x, y, 1, 1, 1, 1, pass, 1};
				ArrayStack stk;
				Counters cnt;
				memset(&cnt, 0, sizeof(cnt));
				renderItems(d, *p, stk, J, 0u, 64u, stage.data(), cnt);
				if (cnt.node_tests > mine[3]) { mine[0] = x; mine[1] = y; mine[2] = pass; mine[3] = cnt.node_tests; mine[4] = cnt.tri_tests; mine[5] = cnt.rays; }
			}
		#pragma omp critical
		if (mine[3] > best[3]) memcpy(best, mine, sizeof(best));
	}
	memcpy(out6, best, sizeof(best));
	return CRH_OK;
}

/*
 * scene_compile.cpp — see scene_compile.h. Host C++; must be built with -ffp-contract=off (the prepared
 * triangles restate poly.c:20-22 and have to come out bit-identical to the reference's per-test values).
 */
#include "scene_compile.h"

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <chrono>
#include <map>
#include <mutex>
#include <thread>

namespace crh {

namespace {

struct Fail {
	int code;
	std::string msg;
};

#define CHECK(cond, code, ...)                                 \
	do {                                                       \
		if (!(cond)) {                                         \
			char buf_[256];                                    \
			snprintf(buf_, sizeof(buf_), __VA_ARGS__);         \
			throw Fail{code, buf_};                            \
		}                                                      \
	} while (0)

/* fn(begin, end) over [0, n) on up to sixteen threads (chunks of at least `grain`); a Fail thrown by any chunk is rethrown here (the first one in index order
 * would need a sort: any one of them is a correct diagnosis). CRH_COMPILE_THREADS=1 keeps everything on the calling thread. */
template <class F> void parallelFor(size_t n, size_t grain, F fn) {
	static const unsigned want = [] { const char *e = getenv("CRH_COMPILE_THREADS"); const unsigned h = std::thread::hardware_concurrency(); return e && atoi(e) > 0 ? (unsigned)atoi(e) : (h > 16u ? 16u : (h ? h : 1u)); }();
	const size_t chunks = std::min<size_t>(want, std::max<size_t>(n / std::max<size_t>(grain, 1), 1));
	if (chunks <= 1) { fn((size_t)0, n); return; }
	std::mutex mu;
	bool failed = false;
	Fail first{0, ""};
	std::vector<std::thread> pool;
	auto run = [&](size_t c) {
		const size_t b = n * c / chunks, e = n * (c + 1) / chunks;
		try { fn(b, e); }
		catch (const Fail &f) { std::lock_guard<std::mutex> g(mu); if (!failed) { failed = true; first = f; } }
		catch (const std::exception &x) { std::lock_guard<std::mutex> g(mu); if (!failed) { failed = true; first = Fail{CRH_ERR_NOMEM, std::string("scene compile: ") + x.what()}; } }
	};
	/* (round 5, ADVICE r04: a thread that cannot be started — std::system_error — must not leave joinable threads behind it, which would end the process in
	 * std::terminate: the chunks that got no thread run here, and the ones that did are joined) */
	pool.reserve(chunks);
	size_t started = 1;
	for (; started < chunks; ++started) {
		try { pool.emplace_back(run, started); }
		catch (const std::exception &) { break; }
	}
	run(0);
	for (size_t c = started; c < chunks; ++c) run(c);
	for (auto &t : pool) t.join();
	if (failed) throw first;
}

inline bool isBsdfKind(uint32_t k) { return k >= CRH_BSDF_DIFFUSE && k <= CRH_BSDF_BACKGROUND; }
inline bool isColorKind(uint32_t k) { return k >= CRH_COLOR_CONSTANT && k <= CRH_COLOR_VECTOCOLOR; }
inline bool isValueKind(uint32_t k) { return k >= CRH_VALUE_CONSTANT && k <= CRH_VALUE_RAYLENGTH; }
inline bool isVectorKind(uint32_t k) { return k >= CRH_VEC_CONSTANT && k <= CRH_VEC_VECMATH; }
enum Cls { COLOR, VALUE, VECTOR };

struct Compiler {
	const crh_scene_desc *s;
	CompiledScene &out;
	std::map<uint32_t, uint32_t> oprMemo;
	std::vector<uint8_t> needUv;          /* per bsdf gnode: its graph reads the hit's uv */
	std::function<void()> texelsReady;
	double tRelayoutPar = 0, tRelayoutDepth = 0, tTriLoop = 0, tResize = 0;          /* CRH_TRACE_UPLOAD: where "BLAS + prepared triangles" goes */
	static double nowMs() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

	Compiler(const crh_scene_desc *scene, CompiledScene &o) : s(scene), out(o) {}

	/* ---- BVHs ---- */
	/* root = device index of the root's child pair (node_count > 1) or of the root leaf itself (node_count == 1) */
	struct BvhInfo { uint32_t dev_base; uint32_t depth; uint32_t root; };

	/* ---- CRH_OPT_WALK = CRH_WALK_WIDE4: the 4-ary copy of one BVH (scene_compile.h: CompiledScene::wide) ---- */
	/* rootPair = device index of the binary root's left child (BvhInfo::root of a BVH with more than one node). Returns the wide root's index in out.wide (in nodes of
	 * 8 x f4); refs are written as such indices with CRH_WIDE_LOCAL set and made absolute by finishWide() once the arrays' sizes are final. Sibling wide nodes are
	 * allocated side by side, then the first sibling's children, ... (depth first). */
	static constexpr uint32_t CRH_WIDE_LOCAL = 0x40000000u;
	uint32_t wideDepthMax = 0;
	static float halfArea(const f4 &a, const f4 &b) {
		const float dx = a.y - a.x, dy = a.w - a.z, dz = b.y - b.x;
		return dx * dy + dy * dz + dz * dx;
	}
	uint32_t buildWide(uint32_t rootPair, const char *what) {
		struct Job { uint32_t pair; uint32_t slot; uint32_t depth; };
		std::vector<Job> todo;
		const uint32_t rootSlot = (uint32_t)(out.wide.size() / 8);
		out.wide.resize(out.wide.size() + 8);
		todo.push_back(Job{rootPair, rootSlot, 1});
		uint32_t depthMax = 0;
		while (!todo.empty()) {
			const Job j = todo.back(); todo.pop_back();
			depthMax = std::max(depthMax, j.depth);
			uint32_t kids[4] = {j.pair, j.pair + 1, 0, 0};
			int n = 2;
			auto isLeaf = [&](uint32_t c) { return CRH_DNODE_ISLEAF(out.nodes[(size_t)c * 2 + 1]); };
			while (n < 4) {          /* the inner child with the largest surface makes room for its two children, in its place */
				int best = -1; float bestA = -1.0f;
				for (int i = 0; i < n; ++i) if (!isLeaf(kids[i])) { const float a = halfArea(out.nodes[(size_t)kids[i] * 2], out.nodes[(size_t)kids[i] * 2 + 1]); if (best < 0 || a > bestA) { best = i; bestA = a; } }
				if (best < 0) break;
				const uint32_t first = CRH_DNODE_FIRST(out.nodes[(size_t)kids[best] * 2 + 1]);
				for (int i = n; i > best + 1; --i) kids[i] = kids[i - 1];
				kids[best] = first; kids[best + 1] = first + 1;
				++n;
			}
			int inner = 0;
			for (int i = 0; i < n; ++i) if (!isLeaf(kids[i])) ++inner;
			const uint32_t firstKid = (uint32_t)(out.wide.size() / 8);
			if (inner) out.wide.resize(out.wide.size() + (size_t)inner * 8);
			f4 *rec = out.wide.data() + (size_t)j.slot * 8;
			int k = 0;
			for (int i = 0; i < 4; ++i) {
				const float inf = __builtin_inff();
				if (i >= n) { rec[2 * i] = f4{inf, inf, inf, inf}; rec[2 * i + 1] = f4{inf, inf, asF32(CRH_NONE), 0.0f}; continue; }          /* min = max = +inf: no regular ray hits it */
				const f4 a = out.nodes[(size_t)kids[i] * 2], b = out.nodes[(size_t)kids[i] * 2 + 1];
				uint32_t ref;
				if (isLeaf(kids[i])) {
					const uint32_t first = CRH_DNODE_FIRST(b), count = CRH_DNODE_COUNT(b);
					CHECK(count <= CRH_WREF_COUNT_MAX && first <= CRH_WREF_FIRST_MASK, CRH_ERR_UNSUPPORTED, "%s: a leaf of %u primitives at slot %u does not fit a wide reference", what, count, first);
					ref = count ? (CRH_WREF_LEAF | (count << CRH_WREF_COUNT_SHIFT) | first) : CRH_NONE;          /* (an empty leaf — bvh.c:220 can make one — is an unused slot) */
				} else {
					ref = CRH_WIDE_LOCAL | (firstKid + (uint32_t)k);
					todo.push_back(Job{CRH_DNODE_FIRST(b), firstKid + (uint32_t)k, j.depth + 1});
					++k;
				}
				rec[2 * i] = a; rec[2 * i + 1] = f4{b.x, b.y, asF32(ref), 0.0f};
			}
			/* (the jobs were pushed first child first: reverse them so that the first child's subtree is laid out next) */
			std::reverse(todo.end() - k, todo.end());
		}
		wideDepthMax = std::max(wideDepthMax, depthMax);
		return rootSlot;
	}
	uint32_t wideAbs(uint32_t localSlot) const { return (uint32_t)((sceneWideOffset(out) >> 4) + (size_t)localSlot * 8u); }
	void finishWide() {
		CHECK(sceneWideOffset(out) + out.wide.size() * sizeof(f4) < (1ull << 32), CRH_ERR_UNSUPPORTED, "BVH nodes + prepared triangles + wide nodes of 4 GB and more");
		parallelFor(out.wide.size() / 2, 1u << 14, [&](size_t b0, size_t e0) {
			for (size_t i = b0; i < e0; ++i) {
				f4 &q = out.wide[2 * i + 1];
				const uint32_t r = asU32(q.z);
				if (r != CRH_NONE && !(r & CRH_WREF_LEAF) && (r & CRH_WIDE_LOCAL)) q.z = asF32(wideAbs(r & ~CRH_WIDE_LOCAL));
			}
		});
	}

	BvhInfo relayoutBvh(uint32_t node_base, uint32_t node_count, uint32_t prim_base, uint32_t prim_count, const char *what) {
		BvhInfo info{0, 0, 0};
		if (out.nodes.size() % 4) out.nodes.resize(out.nodes.size() + 2, f4{0, 0, 0, 0});   /* even device node index */
		const uint32_t dev_base = (uint32_t)(out.nodes.size() / 2);
		info.dev_base = dev_base;
		info.root = dev_base + 1;
		if (node_count == 0) return info;
		CHECK((uint64_t)node_base + node_count <= s->node_count, CRH_ERR_INVALID, "%s: node range out of bounds", what);
		CHECK((uint64_t)prim_base + prim_count <= s->prim_index_count, CRH_ERR_INVALID, "%s: prim range out of bounds", what);
		const double tr0 = nowMs();
		out.nodes.resize((size_t)(dev_base + 1 + node_count) * 2);
		const double tr1 = nowMs();
		tResize += tr1 - tr0;
		out.nodes[(size_t)dev_base * 2] = f4{0, 0, 0, 0}; out.nodes[(size_t)dev_base * 2 + 1] = f4{0, 0, 0, 0};          /* the unused slot in front of the root */
		/* validation + records: every node on its own (threads) ... */
		parallelFor(node_count, 1u << 13, [&](size_t b0, size_t e0) {
			for (uint32_t i = (uint32_t)b0; i < (uint32_t)e0; ++i) {
				const crh_bvh_node &n = s->nodes[node_base + i];
				const uint32_t count = CRH_NODE_PRIMCOUNT(n);
				const bool leaf = CRH_NODE_ISLEAF(n);
				uint32_t first;
				if (leaf) {
					CHECK((uint64_t)n.first + count <= prim_count, CRH_ERR_INVALID, "%s: leaf %u prim range out of bounds", what, i);
					first = prim_base + n.first;
				} else {
					/* children are allocated after their parent, as an (odd, even) pair: bvh.c:221-223 */
					CHECK(n.first > i && (uint64_t)n.first + 1 < node_count && (n.first & 1u), CRH_ERR_INVALID,
						  "%s: inner node %u has child index %u (count %u)", what, i, n.first, node_count);
					first = dev_base + 1 + n.first;
				}
				f4 a{n.bounds[0], n.bounds[1], n.bounds[2], n.bounds[3]};
				f4 b{n.bounds[4], n.bounds[5], asF32(first), asF32((count & 0x3FFFFFFFu) | (leaf ? CRH_DNODE_LEAF_BIT : 0u))};
				out.nodes[(size_t)(dev_base + 1 + i) * 2] = a;
				out.nodes[(size_t)(dev_base + 1 + i) * 2 + 1] = b;
			}
		});
		const double tr2 = nowMs();
		tRelayoutPar += tr2 - tr1;
		/* ... the depth in index order (children come after their parent) */
		std::vector<uint8_t> depth(node_count, 0);
		for (uint32_t i = 0; i < node_count; ++i) {
			const crh_bvh_node &n = s->nodes[node_base + i];
			if (CRH_NODE_ISLEAF(n)) continue;
			const uint32_t d = (uint32_t)depth[i] + 1;
			CHECK(d <= 64, CRH_ERR_UNSUPPORTED, "%s deeper than MAX_BVH_DEPTH 64 (bvh.c:32)", what);
			depth[n.first] = (uint8_t)std::max<uint32_t>(depth[n.first], d);
			depth[n.first + 1] = (uint8_t)std::max<uint32_t>(depth[n.first + 1], d);
			info.depth = std::max(info.depth, d);
		}
		tRelayoutDepth += nowMs() - tr2;
		if (node_count > 1) {
			CHECK(!CRH_NODE_ISLEAF(s->nodes[node_base]), CRH_ERR_INVALID, "%s: multi-node BVH with a leaf root", what);
			info.root = dev_base + 1 + s->nodes[node_base].first;
		}
		return info;
	}

	/* ---- node graph ---- */
	uint32_t addConst(float x, float y, float z, float w) {
		out.consts.push_back(f4{x, y, z, w});
		return (uint32_t)out.consts.size() - 1;
	}
	uint32_t addImage(const crh_gnode &g) {
		if (g.a != CRH_NODE_NONE) CHECK(g.a < s->texture_count, CRH_ERR_INVALID, "image node references texture %u of %llu", g.a, (unsigned long long)s->texture_count);
		out.images.push_back(DImage{g.a, g.b});
		return (uint32_t)out.images.size() - 1;
	}
	const crh_gnode &gnode(uint32_t g, uint32_t parent, Cls cls) {
		CHECK(g != CRH_NODE_NONE, CRH_ERR_INVALID, "node %u has a missing operand", parent);
		CHECK(g < s->gnode_count && (parent == CRH_NODE_NONE || g < parent), CRH_ERR_INVALID, "node %u references node %u (graph must be post-ordered)", parent, g);
		const crh_gnode &n = s->gnodes[g];
		const bool ok = cls == COLOR ? isColorKind(n.kind) : cls == VALUE ? isValueKind(n.kind) : isVectorKind(n.kind);
		CHECK(ok, CRH_ERR_UNSUPPORTED, "node %u (kind %u) is not of the class node %u expects", g, n.kind, parent);
		return n;
	}

	struct Slots {
		uint32_t used = 0;
		int alloc() {
			for (int i = 0; i < CRH_PROG_SLOTS; ++i) if (!(used & (1u << i))) { used |= 1u << i; return i; }
			throw Fail{CRH_ERR_UNSUPPORTED, "node program needs more than CRH_PROG_SLOTS operand slots"};
		}
		void release(int i) { used &= ~(1u << i); }
	};

	int emit(uint32_t g, uint32_t parent, Cls cls, Slots &sl) {
		const crh_gnode &n = gnode(g, parent, cls);
		DOp op;
		memset(&op, 0, sizeof(op));
		op.kind = (uint16_t)n.kind;
		int a = -1, b = -1, c = -1;
		switch (n.kind) {
			case CRH_COLOR_CONSTANT: op.cidx = addConst(n.f[0], n.f[1], n.f[2], n.f[3]); break;
			case CRH_VALUE_CONSTANT: op.cidx = addConst(n.f[0], 0, 0, 0); break;
			case CRH_VEC_CONSTANT: op.cidx = addConst(n.f[0], n.f[1], n.f[2], 0); break;
			case CRH_COLOR_IMAGE: op.u = addImage(n); break;
			case CRH_COLOR_GRADIENT: op.cidx = addConst(n.f[0], n.f[1], n.f[2], n.f[3]); addConst(n.f[4], n.f[5], n.f[6], n.f[7]); break;
			case CRH_COLOR_CHECKER: a = emit(n.a, g, COLOR, sl); b = emit(n.b, g, COLOR, sl); c = emit(n.c, g, VALUE, sl); break;
			case CRH_COLOR_BLACKBODY: case CRH_COLOR_COMBINE: a = emit(n.a, g, VALUE, sl); break;
			case CRH_COLOR_COMBINERGB: a = emit(n.a, g, VALUE, sl); b = emit(n.b, g, VALUE, sl); c = emit(n.c, g, VALUE, sl); break;
			case CRH_COLOR_VECTOCOLOR: a = emit(n.a, g, VECTOR, sl); break;
			case CRH_VALUE_ALPHA: case CRH_VALUE_GRAYSCALE: a = emit(n.a, g, COLOR, sl); break;
			case CRH_VALUE_MATH: a = emit(n.a, g, VALUE, sl); b = emit(n.b, g, VALUE, sl); op.u = n.c; break;
			case CRH_VALUE_FRESNEL: a = emit(n.a, g, VALUE, sl); break;
			case CRH_VALUE_RAYLENGTH: case CRH_VEC_NORMAL: break;
			case CRH_VEC_VECMATH: a = emit(n.a, g, VECTOR, sl); b = emit(n.b, g, VECTOR, sl); op.u = n.c; break;
			default: throw Fail{CRH_ERR_UNSUPPORTED, "unknown pure node kind " + std::to_string(n.kind)};
		}
		if (a >= 0) { op.s0 = (uint8_t)a; sl.release(a); }
		if (b >= 0) { op.s1 = (uint8_t)b; sl.release(b); }
		if (c >= 0) { op.s2 = (uint8_t)c; sl.release(c); }
		const int dst = sl.alloc();
		op.dst = (uint8_t)dst;
		out.prog.push_back(op);
		return dst;
	}

	/* true if the sub-graph rooted at g reads nothing from the hit record (pure function of constants) */
	bool hitIndependent(uint32_t g) {
		if (g == CRH_NODE_NONE || g >= s->gnode_count) return false;
		const crh_gnode &n = s->gnodes[g];
		switch (n.kind) {
			case CRH_COLOR_CONSTANT: case CRH_VALUE_CONSTANT: case CRH_VEC_CONSTANT: return true;
			case CRH_COLOR_BLACKBODY: case CRH_COLOR_COMBINE: case CRH_COLOR_VECTOCOLOR: case CRH_VALUE_ALPHA: case CRH_VALUE_GRAYSCALE:
				return n.a < g && hitIndependent(n.a);
			case CRH_COLOR_COMBINERGB: return n.a < g && n.b < g && n.c < g && hitIndependent(n.a) && hitIndependent(n.b) && hitIndependent(n.c);
			case CRH_VALUE_MATH: case CRH_VEC_VECMATH: return n.a < g && n.b < g && hitIndependent(n.a) && hitIndependent(n.b);
			default: return false;   /* image, checker, gradient, fresnel, rayLength, normal read the hit */
		}
	}

	/* Fold a hit-independent sub-graph by running its program once on the host: the same lane code
	 * (pt_device.h, host build, -ffp-contract=off) with the host libm, i.e. exactly the value the reference
	 * would recompute at every hit. */
	f4 foldOnHost(uint32_t g, uint32_t parent, Cls cls) {
		const size_t progMark = out.prog.size(), constMark = out.consts.size();
		Slots sl;
		const int res = emit(g, parent, cls, sl);
		DOp end;
		memset(&end, 0, sizeof(end));
		end.kind = CRH_OP_END;
		end.s0 = (uint8_t)res;
		out.prog.push_back(end);
		ProgCtx d;
		memset(&d, 0, sizeof(d));
		d.consts = out.consts.data(); d.prog = out.prog.data(); d.images = out.images.data();
		ShadeRec rec;
		memset(&rec, 0, sizeof(rec));
		const f4 v = runProgram(d, (uint32_t)progMark, rec).v;
		out.prog.resize(progMark);
		out.consts.resize(constMark);
		return v;
	}

	uint32_t operand(uint32_t g, uint32_t parent, Cls cls) {
		const crh_gnode &n = gnode(g, parent, cls);
		auto it = oprMemo.find(g);
		if (it != oprMemo.end()) return it->second;
		uint32_t r;
		if (hitIndependent(g)) {
			const f4 v = foldOnHost(g, parent, cls);
			r = CRH_OPR(CRH_OPR_CONST, addConst(v.x, v.y, v.z, v.w));
		} else if (n.kind == CRH_COLOR_IMAGE) {
			r = CRH_OPR(CRH_OPR_IMAGE, addImage(n));
		} else if (n.kind == CRH_VALUE_ALPHA && n.a < g && s->gnodes[n.a].kind == CRH_COLOR_IMAGE) {
			r = CRH_OPR(CRH_OPR_IMAGE_ALPHA, addImage(s->gnodes[n.a]));
		} else if (n.kind == CRH_COLOR_GRADIENT) {
			const uint32_t c = addConst(n.f[0], n.f[1], n.f[2], n.f[3]);
			addConst(n.f[4], n.f[5], n.f[6], n.f[7]);
			r = CRH_OPR(CRH_OPR_GRADIENT, c);
		} else {
			const uint32_t pc = (uint32_t)out.prog.size();
			Slots sl;
			const int res = emit(g, parent, cls, sl);
			DOp end;
			memset(&end, 0, sizeof(end));
			end.kind = CRH_OP_END;
			end.s0 = (uint8_t)res;
			out.prog.push_back(end);
			r = CRH_OPR(CRH_OPR_PROGRAM, pc);
		}
		oprMemo[g] = r;
		return r;
	}

	uint32_t bsdfChild(uint32_t g, uint32_t parent) {
		CHECK(g != CRH_NODE_NONE && g < s->gnode_count && g < parent, CRH_ERR_INVALID, "bsdf node %u references node %u", parent, g);
		CHECK(isBsdfKind(s->gnodes[g].kind) && s->gnodes[g].kind != CRH_BSDF_BACKGROUND, CRH_ERR_UNSUPPORTED, "bsdf node %u: child %u is not a surface bsdf", parent, g);
		return g;
	}

	void compileGraph() {
		const uint32_t N = (uint32_t)s->gnode_count;
		out.bsdfs.assign(N ? N : 1, DBsdf{0, CRH_NONE, CRH_NONE, CRH_NONE});
		std::vector<uint32_t> addDepth(N, 0);
		needUv.assign(N, 0);
		auto oprUv = [](uint32_t opr) { const uint32_t k = opr >> 29; return opr != CRH_NONE && (k == CRH_OPR_IMAGE || k == CRH_OPR_IMAGE_ALPHA || k == CRH_OPR_PROGRAM); };
		for (uint32_t g = 0; g < N; ++g) {
			const crh_gnode &n = s->gnodes[g];
			if (!isBsdfKind(n.kind)) continue;
			DBsdf d{n.kind, CRH_NONE, CRH_NONE, CRH_NONE};
			switch (n.kind) {
				case CRH_BSDF_DIFFUSE: case CRH_BSDF_TRANSPARENT: case CRH_BSDF_ISOTROPIC:
					d.a = operand(n.a, g, COLOR); break;
				case CRH_BSDF_METAL: d.a = operand(n.a, g, COLOR); d.b = operand(n.b, g, VALUE); break;
				case CRH_BSDF_GLASS: d.a = operand(n.a, g, COLOR); d.b = operand(n.b, g, VALUE); d.c = operand(n.c, g, VALUE); break;
				case CRH_BSDF_PLASTIC:
					d.a = operand(n.a, g, COLOR); d.b = operand(n.b, g, COLOR); d.c = bsdfChild(n.c, g);
					addDepth[g] = addDepth[n.c]; break;
				case CRH_BSDF_MIX:
					d.a = bsdfChild(n.a, g); d.b = bsdfChild(n.b, g); d.c = operand(n.c, g, VALUE);
					addDepth[g] = std::max(addDepth[n.a], addDepth[n.b]); break;
				case CRH_BSDF_ADD:
					d.a = bsdfChild(n.a, g); d.b = bsdfChild(n.b, g);
					addDepth[g] = 1 + std::max(addDepth[n.a], addDepth[n.b]); break;
				case CRH_BSDF_EMISSION: d.a = operand(n.a, g, COLOR); d.b = operand(n.b, g, VALUE); break;
				case CRH_BSDF_BACKGROUND: d.a = operand(n.a, g, COLOR); d.b = operand(n.b, g, VALUE); d.c = operand(n.c, g, VALUE); break;
				default: break;
			}
			{   /* does anything below this bsdf read rec.uv? (images and programs do; constants and gradients do not) */
				const bool kids = n.kind == CRH_BSDF_MIX || n.kind == CRH_BSDF_ADD;
				bool u = false;
				if (kids) u = needUv[n.a] || needUv[n.b] || (n.kind == CRH_BSDF_MIX && oprUv(d.c));
				else if (n.kind == CRH_BSDF_PLASTIC) u = oprUv(d.a) || oprUv(d.b) || needUv[n.c];
				else u = oprUv(d.a) || oprUv(d.b) || oprUv(d.c);
				needUv[g] = u ? 1 : 0;
			}
			CHECK(addDepth[g] <= CRH_ADD_DEPTH, CRH_ERR_UNSUPPORTED, "bsdf node %u nests add nodes %u deep (device limit %d)", g, addDepth[g], CRH_ADD_DEPTH);
			out.max_add_depth = std::max(out.max_add_depth, addDepth[g]);
			out.bsdfs[g] = d;
		}
		if (out.consts.empty()) addConst(0, 0, 0, 0);
		if (out.images.empty()) out.images.push_back(DImage{CRH_NONE, 0});
		if (out.prog.empty()) { DOp end; memset(&end, 0, sizeof(end)); end.kind = CRH_OP_END; out.prog.push_back(end); }
	}

	void run() {
		auto t0 = std::chrono::steady_clock::now();
		auto lap = [&]() { const auto t1 = std::chrono::steady_clock::now(); const double ms = std::chrono::duration<double, std::milli>(t1 - t0).count(); t0 = t1; return ms; };
		CHECK(s->struct_size == sizeof(crh_scene_desc), CRH_ERR_INVALID, "crh_scene_desc.struct_size %u != %zu", s->struct_size, sizeof(crh_scene_desc));
		CHECK(s->abi_version == CRH_SCENE_VERSION, CRH_ERR_INVALID, "scene description version %u != %d", s->abi_version, CRH_SCENE_VERSION);
		CHECK(s->node_count < 0x3FFFFFFFull && s->prim_index_count < 0x3FFFFFFFull && s->poly_count < 0x7FFFFFFFull, CRH_ERR_UNSUPPORTED, "scene too large for 32-bit device indices");
		CHECK(s->camera.width > 0 && s->camera.height > 0, CRH_ERR_INVALID, "camera has no image size");
		out.camera = s->camera;

		{	/* the totals first: the big arrays are allocated once (PodBuf: huge pages; growth would copy) */
			uint64_t texelTotal = 0, nodeTotal = 4;
			for (uint64_t t = 0; t < s->texture_count; ++t) texelTotal += (uint64_t)s->textures[t].width * s->textures[t].height;
			for (uint64_t m = 0; m < s->mesh_count; ++m) nodeTotal += ((uint64_t)s->meshes[m].node_count + 3) * 2;
			nodeTotal += ((uint64_t)s->tlas_node_count + 3) * 2;
			if (texelTotal < 0xFFFFFFFFull) out.texels.reserve((size_t)std::max<uint64_t>(texelTotal, 1));
			if (nodeTotal < 0xFFFFFFFFull) out.nodes.reserve((size_t)nodeTotal);
		}
		for (uint64_t t = 0; t < s->texture_count; ++t) {
			const crh_texture &tx = s->textures[t];
			CHECK(tx.width > 0 && tx.height > 0 && (tx.channels == 1 || tx.channels == 3 || tx.channels == 4), CRH_ERR_INVALID, "texture %llu has a bad shape", (unsigned long long)t);
			const uint64_t bytes = (uint64_t)tx.width * tx.height * tx.channels * (tx.is_float ? 4u : 1u);
			CHECK(tx.offset % 4 == 0 && tx.offset + bytes <= s->texture_bytes, CRH_ERR_INVALID, "texture %llu data out of bounds", (unsigned long long)t);
			CHECK(!tx.has_alpha || tx.channels == 4, CRH_ERR_INVALID, "texture %llu: has_alpha needs 4 channels", (unsigned long long)t);
			CHECK(out.texels.size() + (uint64_t)tx.width * tx.height < 0xFFFFFFFFull, CRH_ERR_UNSUPPORTED, "more than 2^32 texels");
			DTexture d;
			memset(&d, 0, sizeof(d));
			d.first = (uint32_t)out.texels.size();
			d.width = tx.width; d.height = tx.height;
			d.m64w = (uint32_t)((0xFFFFFFFFFFFFFFFFull % tx.width + 1) % tx.width);
			d.m64h = (uint32_t)((0xFFFFFFFFFFFFFFFFull % tx.height + 1) % tx.height);
			out.textures.push_back(d);
			/* texture.c:32-63, once per texel instead of once per fetch */
			const uint8_t *bytesp = s->texture_data + tx.offset;
			const float *floats = (const float *)bytesp;
			const size_t W = tx.width, H = tx.height, Cn = tx.channels;
			out.texels.resize(out.texels.size() + W * H);
			f4 *dst = out.texels.data() + d.first;
			parallelFor(H, 64, [&](size_t y0, size_t y1) {
			for (size_t y = y0; y < y1; ++y)
				for (size_t x = 0; x < W; ++x) {
					const size_t base = (x + ((H - 1) - y) * W) * Cn;
					f4 o;
					if (Cn == 1) {
						o.x = tx.is_float ? floats[base] : (float)bytesp[base] / 255.0f;
						o.y = o.x; o.z = o.x; o.w = 1.0f;
					} else if (tx.is_float) {
						o.x = floats[base]; o.y = floats[base + 1]; o.z = floats[base + 2];
						o.w = tx.has_alpha ? floats[base + 3] : 1.0f;
					} else {
						o.x = (float)bytesp[base] / 255.0f; o.y = (float)bytesp[base + 1] / 255.0f; o.z = (float)bytesp[base + 2] / 255.0f;
						o.w = tx.has_alpha ? (float)bytesp[base + 3] / 255.0f : 1.0f;
					}
					dst[x + y * W] = o;
				}
			});
		}
		if (out.textures.empty()) { DTexture d; memset(&d, 0, sizeof(d)); d.width = d.height = 1; out.textures.push_back(d); }
		if (out.texels.empty()) out.texels.push_back(f4{0, 0, 0, 0});
		if (texelsReady) texelsReady();

		const double tTex = lap();
		compileGraph();
		const double tGraph = lap();
		CHECK(s->background < s->gnode_count && s->gnodes[s->background].kind == CRH_BSDF_BACKGROUND, CRH_ERR_INVALID, "scene.background is not a background node");
		out.background = s->background;
		out.materials.assign(s->materials, s->materials + s->material_count);
		if (out.materials.empty()) { crh_material z; memset(&z, 0, sizeof(z)); out.materials.push_back(z); }
		for (uint64_t m = 0; m < s->material_count; ++m) {
			const uint32_t b = s->materials[m].bsdf;
			CHECK(b < s->gnode_count && isBsdfKind(s->gnodes[b].kind) && s->gnodes[b].kind != CRH_BSDF_BACKGROUND, CRH_ERR_INVALID, "material %llu has no surface bsdf", (unsigned long long)m);
			out.materials[m].pad[0] = needUv[b];
		}

		const double tBlas0 = nowMs();
		/* BLAS per mesh, prepared triangles */
		/* (every prim slot of a mesh with a BVH is written below; slots no mesh owns — none in scenes the flattener makes — are zeroed first) */
		out.tris.resize((size_t)std::max<uint64_t>(s->prim_index_count, 1) * 3);
		out.shade.resize((size_t)std::max<uint64_t>(s->prim_index_count, 1));
		{	/* ... only them: the top-level BVH's own prim slots are such slots in EVERY scene, and zeroing both arrays whole for their sake was a single-threaded
			 * pass over 75 MB — 2.5 of hdr.json's 8.7 ms (round 4, CRH_TRACE_UPLOAD) */
			std::vector<std::pair<uint64_t, uint64_t>> owned;
			for (uint64_t m = 0; m < s->mesh_count; ++m)
				if (s->meshes[m].node_count && s->meshes[m].poly_count) owned.push_back({s->meshes[m].prim_base, (uint64_t)s->meshes[m].prim_base + s->meshes[m].poly_count});
			std::sort(owned.begin(), owned.end());
			const uint64_t slots = std::max<uint64_t>(s->prim_index_count, 1);
			auto zero = [&](uint64_t b, uint64_t e) {
				e = std::min(e, slots);
				if (b >= e) return;
				memset((void *)(out.tris.data() + b * 3), 0, (size_t)(e - b) * 3 * sizeof(f4));
				memset((void *)(out.shade.data() + b), 0, (size_t)(e - b) * sizeof(DShadeTri));
			};
			uint64_t at = 0;
			for (const auto &r : owned) { zero(at, r.first); at = std::max(at, r.second); }
			zero(at, slots);
		}
		const double tPrelude = nowMs();
		std::vector<BvhInfo> meshBvh(s->mesh_count);
		uint32_t maxBlasDepth = 0;
		for (uint64_t m = 0; m < s->mesh_count; ++m) {
			const crh_mesh &mesh = s->meshes[m];
			CHECK((uint64_t)mesh.poly_base + mesh.poly_count <= s->poly_count, CRH_ERR_INVALID, "mesh %llu: polygon range out of bounds", (unsigned long long)m);
			CHECK((uint64_t)mesh.material_base + mesh.material_count <= s->material_count && mesh.material_count > 0, CRH_ERR_INVALID, "mesh %llu: material range out of bounds", (unsigned long long)m);
			meshBvh[m] = relayoutBvh(mesh.node_base, mesh.node_count, mesh.prim_base, mesh.node_count ? mesh.poly_count : 0, "BLAS");
			maxBlasDepth = std::max(maxBlasDepth, meshBvh[m].depth);
			if (!mesh.node_count) continue;
			const double tt0 = nowMs();
			parallelFor(mesh.poly_count, 1u << 14, [&](size_t k0, size_t k1) {
			for (uint32_t k = (uint32_t)k0; k < (uint32_t)k1; ++k) {
				const int32_t pi = s->prim_indices[mesh.prim_base + k];
				CHECK(pi >= 0 && (uint32_t)pi < mesh.poly_count, CRH_ERR_INVALID, "mesh %llu: prim index %d out of range", (unsigned long long)m, pi);
				const crh_poly &p = s->polys[mesh.poly_base + pi];
				for (int j = 0; j < 3; ++j) {
					CHECK(p.v[j] >= 0 && (uint64_t)p.v[j] < s->vertex_count, CRH_ERR_INVALID, "polygon %u: vertex index out of range", mesh.poly_base + pi);
					if (CRH_POLY_HASNORMALS(p)) CHECK(p.n[j] >= 0 && (uint64_t)p.n[j] < s->normal_count, CRH_ERR_INVALID, "polygon %u: normal index out of range", mesh.poly_base + pi);
					if (mesh.texcoord_count && p.t[0] != -1) CHECK(p.t[j] >= 0 && (uint64_t)p.t[j] < s->texcoord_count, CRH_ERR_INVALID, "polygon %u: texcoord index out of range", mesh.poly_base + pi);
				}
				CHECK(CRH_POLY_MATERIAL(p) < mesh.material_count, CRH_ERR_INVALID, "polygon %u: material index out of range", mesh.poly_base + pi);
				const float *V = s->vertices;
				const v3 v0{V[3 * (size_t)p.v[0]], V[3 * (size_t)p.v[0] + 1], V[3 * (size_t)p.v[0] + 2]};
				const v3 v1{V[3 * (size_t)p.v[1]], V[3 * (size_t)p.v[1] + 1], V[3 * (size_t)p.v[1] + 2]};
				const v3 v2{V[3 * (size_t)p.v[2]], V[3 * (size_t)p.v[2] + 1], V[3 * (size_t)p.v[2] + 2]};
				const v3 e1 = vsub(v0, v1);          /* poly.c:20 */
				const v3 e2 = vsub(v2, v0);          /* poly.c:21 */
				const v3 n = vcross(e1, e2);         /* poly.c:22 */
				f4 *q = &out.tris[(size_t)(mesh.prim_base + k) * 3];
				q[0] = f4{v0.x, v0.y, v0.z, e1.x};
				q[1] = f4{e1.y, e1.z, e2.x, e2.y};
				q[2] = f4{e2.z, n.x, n.y, n.z};
				DShadeTri &st = out.shade[(size_t)mesh.prim_base + k];
				memset((void *)&st, 0, sizeof(st));
				st.flags = CRH_POLY_MATERIAL(p);
				if (CRH_POLY_HASNORMALS(p)) {
					st.flags |= CRH_SHADE_HASNORMALS;
					const float *N = s->normals;
					for (int c = 0; c < 3; ++c) { st.n0[c] = N[3 * (size_t)p.n[0] + c]; st.n1[c] = N[3 * (size_t)p.n[1] + c]; st.n2[c] = N[3 * (size_t)p.n[2] + c]; }
				} else {
					st.n0[0] = n.x; st.n0[1] = n.y; st.n0[2] = n.z;
				}
				if (mesh.texcoord_count && p.t[0] != -1) {
					st.flags |= CRH_SHADE_HASUV;
					const float *T = s->texcoords;
					for (int c = 0; c < 2; ++c) { st.t0[c] = T[2 * (size_t)p.t[0] + c]; st.t1[c] = T[2 * (size_t)p.t[1] + c]; st.t2[c] = T[2 * (size_t)p.t[2] + c]; }
				}
			}
			});
			tTriLoop += nowMs() - tt0;
		}

		const double tBlas = lap();
		if (getenv("CRH_TRACE_UPLOAD")) fprintf(stderr, "compile_scene trace: textures %.1f ms, node graph %.1f ms, BLAS + prepared triangles %.1f ms (reservations %.1f, node records %.1f, depth pass %.1f, triangle loop %.1f, resizes %.1f)\n", tTex, tGraph, tBlas, tPrelude - tBlas0, tRelayoutPar, tRelayoutDepth, tTriLoop, tResize);
		/* TLAS */
		CHECK(s->tlas_prim_count == (s->tlas_node_count ? s->instance_count : 0), CRH_ERR_INVALID, "TLAS prim count does not match the instance count");
		const BvhInfo tlas = relayoutBvh(s->tlas_node_base, s->tlas_node_count, s->tlas_prim_base, s->tlas_prim_count, "TLAS");
		out.tlas_root = tlas.root;
		out.tlas_first = tlas.dev_base + 2;
		out.tlas_node_count = s->tlas_node_count;
		out.tlas_prim_base = s->tlas_prim_base;
		for (uint32_t k = 0; k < s->tlas_prim_count; ++k) {
			const int32_t ii = s->prim_indices[s->tlas_prim_base + k];
			CHECK(ii >= 0 && (uint64_t)ii < s->instance_count, CRH_ERR_INVALID, "TLAS prim %u: instance index out of range", k);
		}
		out.max_stack = tlas.depth + CRH_TLAS_SAVE + maxBlasDepth + 1;

		/* instances in TLAS leaf order: the TLAS prim indices are a permutation of the instances (bvh.c:289-316) */
		{ DInstance z; memset(&z, 0, sizeof(z)); z.kind = CRH_DINST_MESH_EMPTY; out.instances.assign(std::max<uint64_t>(s->tlas_prim_count, 1), z); }
		std::vector<uint8_t> seen(s->instance_count, 0);
		for (uint32_t k = 0; k < s->tlas_prim_count; ++k) {
			const uint32_t i = (uint32_t)s->prim_indices[s->tlas_prim_base + k];
			CHECK(!seen[i], CRH_ERR_INVALID, "TLAS references instance %u twice", i);
			seen[i] = 1;
			const crh_instance &in = s->instances[i];
			DInstance d;
			memset(&d, 0, sizeof(d));
			memcpy(d.Ainv, in.Ainv, sizeof(d.Ainv));
			memcpy(d.A, in.A, sizeof(d.A));
			d.orig = i;
			const bool volume = in.kind == CRH_INSTANCE_SPHERE_VOLUME || in.kind == CRH_INSTANCE_MESH_VOLUME;
			if (volume) {        /* instance.c:62-92, 187-216: -(1 / density) * logf(u) */
				out.has_volumes = true;
				d.density = in.density;
			}
			if (in.kind == CRH_INSTANCE_SPHERE || in.kind == CRH_INSTANCE_SPHERE_VOLUME) {
				CHECK(in.object < s->sphere_count, CRH_ERR_INVALID, "instance %u: sphere index out of range", i);
				const crh_sphere &sp = s->spheres[in.object];
				CHECK(sp.material < s->material_count, CRH_ERR_INVALID, "sphere %u: material out of range", in.object);
				d.kind = CRH_DINST_SPHERE;
				d.radius = sp.radius; d.ray_offset = sp.ray_offset; d.material = sp.material;
			} else if (in.kind == CRH_INSTANCE_MESH || in.kind == CRH_INSTANCE_MESH_VOLUME) {
				CHECK(in.object < s->mesh_count, CRH_ERR_INVALID, "instance %u: mesh index out of range", i);
				const crh_mesh &mesh = s->meshes[in.object];
				d.kind = mesh.node_count > 1 ? CRH_DINST_MESH : mesh.node_count == 1 ? CRH_DINST_MESH_LEAF : CRH_DINST_MESH_EMPTY;
				d.root = meshBvh[in.object].root;
				d.ray_offset = mesh.ray_offset;
				d.material = mesh.material_base;
				d.poly_base = mesh.poly_base;
			} else {
				throw Fail{CRH_ERR_UNSUPPORTED, "unknown instance kind " + std::to_string(in.kind)};
			}
			if (volume) {
				d.kind |= CRH_DINST_VOLUME;
				if (in.kind == CRH_INSTANCE_MESH_VOLUME) CHECK(s->meshes[in.object].material_count >= 1, CRH_ERR_INVALID, "instance %u: a mesh volume needs materials[0]", i);
			}
			out.instances[k] = d;
		}
		assignShadeClasses();
		if (getenv("CRH_TRACE_UPLOAD")) fprintf(stderr, "compile_scene trace: TLAS + instances + classes %.1f ms\n", lap());
		if (out.nodes.empty()) out.nodes.resize(4, f4{0, 0, 0, 0});
		/*  a scene with node programs or volumes is rendered by the rare-features kernel, which has no wide instantiation (cray_hip.hip: wideWalk): no copy is
		 * built or uploaded for it — the binary walk renders it, as it did, and the context says so */
		if (out.want_wide && (out.prog.size() > 1 || out.has_volumes)) out.wide_refused = "node programs / volumes: the rare-features kernel walks the binary tree";
		else if (out.want_wide) {
			try {
				std::vector<uint32_t> wideRoot(s->mesh_count, CRH_NONE);
				uint32_t blasDepth = 0;
				for (uint64_t m = 0; m < s->mesh_count; ++m) if (s->meshes[m].node_count > 1) { wideDepthMax = 0; wideRoot[m] = buildWide(meshBvh[m].root, "BLAS"); blasDepth = std::max(blasDepth, wideDepthMax); }
				uint32_t tlasWide = CRH_NONE;
				wideDepthMax = 0;
				if (s->tlas_node_count > 1) tlasWide = buildWide(tlas.root, "TLAS");
				out.wide_max_stack = 3u * (wideDepthMax + blasDepth) + 1u;
				CHECK(out.wide_max_stack <= 134u, CRH_ERR_UNSUPPORTED, "the wide walk could need %u stack entries (device: 134)", out.wide_max_stack);
				finishWide();
				if (tlasWide != CRH_NONE) out.wide_tlas_root = wideAbs(tlasWide);
				for (uint32_t k = 0; k < s->tlas_prim_count; ++k) {
					const crh_instance &in = s->instances[out.instances[k].orig];
					if ((in.kind == CRH_INSTANCE_MESH || in.kind == CRH_INSTANCE_MESH_VOLUME) && wideRoot[in.object] != CRH_NONE) out.instances[k].radius = asF32(wideAbs(wideRoot[in.object]));
				}
				if (out.wide.empty()) out.wide.resize(8, f4{0, 0, 0, 0});
			} catch (const Fail &f) {
				out.wide.resize(0); out.wide_refused = f.msg;
				for (uint32_t k = 0; k < s->tlas_prim_count; ++k) if (CRH_DINST_KIND(out.instances[k].kind) != CRH_DINST_SPHERE) out.instances[k].radius = 0.0f;
			}
			if (getenv("CRH_TRACE_UPLOAD")) fprintf(stderr, "compile_scene trace: wide BVH copy %.1f ms (%zu wide nodes%s%s)\n", lap(), out.wide.size() / 8, out.wide_refused.empty() ? "" : "; refused: ", out.wide_refused.c_str());
		}
		for (uint64_t i = 0; i < s->instance_count; ++i) {           /* instances outside the TLAS (none with the reference's builder) are still validated */
			const crh_instance &in = s->instances[i];
			CHECK(in.kind <= CRH_INSTANCE_MESH_VOLUME, CRH_ERR_UNSUPPORTED, "unknown instance kind %u", in.kind);
			const bool sphere = in.kind == CRH_INSTANCE_SPHERE || in.kind == CRH_INSTANCE_SPHERE_VOLUME;
			CHECK(sphere ? in.object < s->sphere_count : in.object < s->mesh_count, CRH_ERR_INVALID, "instance %llu: object index out of range", (unsigned long long)i);
		}
		if (out.nodes.empty()) out.nodes.resize(4, f4{0, 0, 0, 0});
	}

	/* Shade classes (pt_device.h: CRH_DINST_CLASS): the path-tracing kernel shades hits in batches of one class, so that a batch runs ONE
	 * surface-shader code path instead of the union of all of them. A class = what the shading code branches on: sphere or mesh, the
	 * root bsdf kind of the material, whether the graph reads uv, whether the material emits. Meshes whose polygons use materials of
	 * different signatures are "mixed". The seven most frequent signatures get a class of their own, the rest share the last one.
	 * Purely a scheduling hint: every path's own sequence of operations, hence every result, is independent of it. */
	void assignShadeClasses() {
		std::vector<uint32_t> sigOfMesh(s->mesh_count, 0xFFFFFFFFu);
		std::vector<uint32_t> sigMemo((size_t)s->material_count * 2u, 0xFFFFFFFFu);          /* a mesh asks once per POLYGON: the walk of a material's graph is done once */
		auto sigOfMaterial = [&](uint32_t m, bool sphere) -> uint32_t {
			if (m >= s->material_count) return 0xFFFFFFu;
			uint32_t &memo = sigMemo[(size_t)m * 2u + (sphere ? 1u : 0u)];
			if (memo != 0xFFFFFFFFu) return memo;
			const crh_material &mt = s->materials[m];
			/* the bsdf kinds in the material's graph (the loader wraps every JSON material in mix(transparent, X, alpha): the root says nothing) */
			uint32_t kinds = 0;
			std::vector<uint32_t> todo{mt.bsdf};
			for (int guard = 0; !todo.empty() && guard < 4096; ++guard) {
				const uint32_t g = todo.back(); todo.pop_back();
				if (g >= s->gnode_count || !isBsdfKind(s->gnodes[g].kind)) continue;
				kinds |= 1u << (s->gnodes[g].kind & 15u);
				todo.push_back(s->gnodes[g].a); todo.push_back(s->gnodes[g].b); todo.push_back(s->gnodes[g].c);
			}
			const bool emits = mt.emission[0] > 0.0f || mt.emission[1] > 0.0f || mt.emission[2] > 0.0f;
			memo = kinds | (out.materials[m].pad[0] ? 1u << 16 : 0u) | (emits ? 1u << 17 : 0u) | (sphere ? 1u << 18 : 0u);
			return memo;
		};
		std::vector<uint32_t> sig(out.instances.size(), 0xFFFFFFu);
		for (uint32_t k = 0; k < s->tlas_prim_count; ++k) {
			const crh_instance &in = s->instances[out.instances[k].orig];
			if (in.kind == CRH_INSTANCE_SPHERE || in.kind == CRH_INSTANCE_SPHERE_VOLUME) {
				sig[k] = sigOfMaterial(s->spheres[in.object].material, true);
			} else {
				uint32_t &ms = sigOfMesh[in.object];
				if (ms == 0xFFFFFFFFu) {
					const crh_mesh &mesh = s->meshes[in.object];
					ms = 0xFFFFFFu;
					for (uint64_t pi = 0; pi < mesh.poly_count && mesh.poly_base + pi < s->poly_count; ++pi) {
						const uint32_t one = sigOfMaterial(mesh.material_base + CRH_POLY_MATERIAL(s->polys[mesh.poly_base + pi]), false);
						if (pi == 0) ms = one;
						else if (one != ms) { ms = 0xFFFFFEu; break; }          /* mixed */
					}
				}
				sig[k] = ms;
			}
		}
		std::vector<std::pair<uint32_t, uint32_t>> freq;             /* (count, signature) */
		for (uint32_t k = 0; k < s->tlas_prim_count; ++k) {
			auto it = std::find_if(freq.begin(), freq.end(), [&](const std::pair<uint32_t, uint32_t> &f) { return f.second == sig[k]; });
			if (it == freq.end()) freq.push_back({1u, sig[k]}); else ++it->first;
		}
		std::stable_sort(freq.begin(), freq.end(), [](const std::pair<uint32_t, uint32_t> &a, const std::pair<uint32_t, uint32_t> &b) { return a.first > b.first; });
		for (uint32_t k = 0; k < s->tlas_prim_count; ++k) {
			uint32_t cls = CRH_DINST_CLASSES - 1u;
			for (uint32_t f = 0; f < freq.size() && f < CRH_DINST_CLASSES - 1u; ++f) if (freq[f].second == sig[k]) { cls = f; break; }
			out.instances[k].kind |= cls << CRH_DINST_CLASS_SHIFT;
		}
		out.shade_classes = (uint32_t)std::min<size_t>(freq.size(), CRH_DINST_CLASSES);
		if (getenv("CRH_DEBUG_CLASSES")) for (auto &f : freq) fprintf(stderr, "shade class signature 0x%x: %u instances\n", f.second, f.first);
	}
};

}  // namespace

int compile_scene(const crh_scene_desc *scene, CompiledScene &out, std::string &err, const std::function<void()> &texelsReady) {
	if (!scene) { err = "null scene"; return CRH_ERR_INVALID; }
	try {
		Compiler c(scene, out);
		c.texelsReady = texelsReady;
		c.run();
		/* the kernels address a record as (array base) + a 32-BIT byte offset (pt_device.h: one scalar base, one vector offset per load): no device array of
		 * 4 GB or more — 134 M BVH nodes, 89 M triangles, 67 M shading records, 268 M texels per scene */
		const uint64_t lim = 1ull << 32;
		if (out.nodes.size() * sizeof(f4) >= lim || out.tris.size() * sizeof(f4) >= lim || out.shade.size() * sizeof(DShadeTri) >= lim || out.texels.size() * sizeof(f4) >= lim)
			throw Fail{CRH_ERR_UNSUPPORTED, "scene too large: a device array of 4 GB or more (32-bit record offsets)"};
	} catch (const Fail &f) {
		err = f.msg;
		return f.code;
	} catch (const std::bad_alloc &) {
		err = "out of host memory";
		return CRH_ERR_NOMEM;
	} catch (const std::exception &x) {          /* (e.g. std::system_error from the texelsReady callback's helper thread: nothing C++ crosses the C boundary) */
		err = std::string("scene compile: ") + x.what();
		return CRH_ERR_NOMEM;
	}
	return CRH_OK;
}

}  // namespace crh

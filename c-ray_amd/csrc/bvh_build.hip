/*
 * bvh_build.hip — SURVEY.md §8(f) row 1: the reference's binned-SAH BVH builder (src/accelerators/bvh.c:87-316) on the
 * GPU, producing THE SAME tree: node numbering, node bounds (bit patterns) and primitive order are the reference's,
 * because the hot path's parity (per-ray node / triangle test counts, tie order of equal-distance hits) hangs on them.
 *
 * The reference is a depth-first recursion over index ranges; what it computes per node is
 *   (1) 3 x 32 bins over the NODE bounds (bvh.c:87-93, 158-165): count + box of the primitives whose centre falls in the bin;
 *   (2) two 31-step sweeps per axis (bvh.c:170-191), the axis / leaf / median-fallback decision (bvh.c:195-215);
 *   (3) a two-pointer in-place partition of the index range (bvh.c:95-130);
 *   (4) child boxes = folds of the bin boxes (bvh.c:226-233); children numbered when the parent splits, left subtree first.
 * (1) and (3) are the work; they are data-parallel once two order dependences are made explicit:
 *   * box folds use includes.h:20-21 `min(a,b) = a < b ? a : b`: on a tie the LATER operand wins, which is visible only
 *     for -0 / +0. Boxes are therefore reduced as 64-bit keys {sortable value with -0 == +0, sequence position, sign
 *     of zero} with atomic min / max: the result is the reference's sequential fold, bit for bit, in any order.
 *   * the partition swaps the k-th misplaced element from the left with the k-th misplaced element from the right:
 *     ranks come from prefix sums, the swaps are independent.
 * Node numbering (depth-first, a pair allocated per split) is a prefix sum over split counts, done last.
 *
 * Two phases:
 *   LARGE  level-synchronous over the nodes that still hold more than CRH_BVH_SMALL primitives: per level four kernels over
 *          2048-primitive chunks (bin -> decide -> rank -> swap); the host keeps the (small) upper tree.
 *   SMALL  every remaining subtree is built by ONE WAVE, depth-first like the reference, with its index range, bins and
 *          rank lists in LDS; it numbers its nodes locally in the reference's allocation order.
 * Then the host numbers the upper tree + subtrees depth-first and one kernel emits the final node array.
 * Everything is plain fp32 with the reference's operation order (this TU is built with -ffp-contract=off and
 * correctly rounded division like the rest of the library); bvh.c:87-93's float -> unsigned conversion is the x86-64
 * one (64-bit cvttss2si, low word), spelled out in binOf().
 */
#include <hip/hip_runtime.h>
#include <float.h>
#include <stdint.h>
#include <string.h>
#include <stdio.h>
#include <stdlib.h>
#include <algorithm>
#include <chrono>
#include <string>
#include <vector>

#include "cray_hip.h"
#include "ctx_access.h"

namespace crhb {

#define CRH_BVH_BINS 32
#define CRH_BVH_MAX_DEPTH 64u
#define CRH_BVH_MAX_LEAF 16u
#ifndef CRH_BVH_SMALL
#define CRH_BVH_SMALL 512u          /* subtrees of at most this many primitives are built by one wave */
#endif
#define CRH_BVH_CHUNK 2048u         /* primitives per workgroup in the large phase */

/* ---- order-independent folds with the reference's tie rule ----------------------------------------------------- */
__host__ __device__ inline uint32_t sortable(float f) {                  /* monotone in the value; -0 and +0 share a key */
	uint32_t u;
	memcpy(&u, &f, 4);
	if (u == 0x80000000u) u = 0u;
	return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__host__ __device__ inline float unsortable(uint32_t s) {
	const uint32_t u = (s & 0x80000000u) ? (s & 0x7FFFFFFFu) : ~s;
	float f;
	memcpy(&f, &u, 4);
	return f;
}
__host__ __device__ inline uint32_t isNegZero(float f) { uint32_t u; memcpy(&u, &f, 4); return u == 0x80000000u ? 1u : 0u; }
/* min fold: smaller value wins, on a tie the LATER position (atomicMin: smaller low word = later position) */
__host__ __device__ inline unsigned long long keyLo(float f, uint32_t pos) {
	return ((unsigned long long)sortable(f) << 32) | (unsigned long long)(((0x7FFFFFFFu - pos) << 1) | isNegZero(f));
}
/* max fold: larger value wins, on a tie the LATER position (atomicMax: larger low word = later position) */
__host__ __device__ inline unsigned long long keyHi(float f, uint32_t pos) {
	return ((unsigned long long)sortable(f) << 32) | (unsigned long long)((pos << 1) | isNegZero(f));
}
__host__ __device__ inline float keyValue(unsigned long long k) {
	const uint32_t s = (uint32_t)(k >> 32);
	if (s == 0x80000000u) return (k & 1ull) ? -0.0f : 0.0f;
	return unsortable(s);
}
#define CRH_KEY_LO_EMPTY ((((unsigned long long)0xFF7FFFFFu) << 32) | 0xFFFFFFFFull)   /* sortable(+FLT_MAX), loses every tie */
#define CRH_KEY_HI_EMPTY ((((unsigned long long)0x00800000u) << 32) | 0ull)            /* sortable(-FLT_MAX), loses every tie */

__host__ __device__ inline float pickLo(float a, float b) { return a < b ? a : b; }   /* includes.h:20 */
__host__ __device__ inline float pickHi(float a, float b) { return a > b ? a : b; }   /* includes.h:21 */

/* bvh.c:87-93. The (unsigned) cast of the reference is gcc/x86-64's: cvttss2si to 64 bits, low word kept — NaN and
 * anything >= 2^63 give 0 (0x8000000000000000 truncated), in-range values truncate. */
__host__ __device__ inline uint32_t binOf(float coord, float lo, float hi) {
	const float scale = 32.0f / (hi - lo);
	const float f = (coord - lo) * scale;
	uint32_t b;
	if (f < 0.0f) b = 0u;
	else if (!(f < 9223372036854775808.0f)) b = 0u;
	else b = (uint32_t)(unsigned long long)f;
	return b >= CRH_BVH_BINS ? CRH_BVH_BINS - 1u : b;
}

struct Box { float lo[3], hi[3]; };
__host__ __device__ inline void boxReset(Box &b) { for (int k = 0; k < 3; ++k) { b.lo[k] = FLT_MAX; b.hi[k] = -FLT_MAX; } }
__host__ __device__ inline void boxGrow(Box &d, const Box &s) {
	for (int k = 0; k < 3; ++k) { d.lo[k] = pickLo(d.lo[k], s.lo[k]); d.hi[k] = pickHi(d.hi[k], s.hi[k]); }
}
__host__ __device__ inline float boxHalfArea(const Box &b) {              /* bbox.h:25-28 */
	const float ex = b.hi[0] - b.lo[0], ey = b.hi[1] - b.lo[1], ez = b.hi[2] - b.lo[2];
	return ex * (ey + ez) + ey * ez;
}

/* one bin: 3 min keys, 3 max keys, count */
struct BinKeys { unsigned long long lo[3], hi[3]; };

/* per node (large phase) or per wave (small phase): what bvh.c:148-215 decides */
struct Decision {
	uint32_t axis, split, leaf, nLeft;
	float childL[6], childR[6];              /* {minx,maxx,miny,maxy,minz,maxz} like bvhNode.bounds */
};

/* ---- the sweeps + decision of one node, by one wave -------------------------------------------------------------------
 * bvh.c:170-191 are folds over the 32 bins in a fixed order; folds with pickLo / pickHi are associative as long as the operand
 * order is kept (on a tie the later operand wins), so they are inclusive scans: lane b (b = lane & 31; both halves of the
 * wave compute the same) holds bin b and combines with lane b -/+ 1, 2, 4, 8, 16. */
struct LaneBin { float lo[3], hi[3]; uint32_t n; };
__device__ inline LaneBin binJoin(const LaneBin &earlier, const LaneBin &later) {
	LaneBin r;
	for (int k = 0; k < 3; ++k) { r.lo[k] = pickLo(earlier.lo[k], later.lo[k]); r.hi[k] = pickHi(earlier.hi[k], later.hi[k]); }
	r.n = earlier.n + later.n;
	return r;
}
__device__ inline LaneBin binFrom(const LaneBin &v, int srcBin) {        /* the value lane `srcBin` of this half holds */
	LaneBin r;
	for (int k = 0; k < 3; ++k) { r.lo[k] = __shfl(v.lo[k], srcBin, 32); r.hi[k] = __shfl(v.hi[k], srcBin, 32); }
	r.n = __shfl(v.n, srcBin, 32);
	return r;
}
__device__ inline float binHalfArea(const LaneBin &v) {
	const float ex = v.hi[0] - v.lo[0], ey = v.hi[1] - v.lo[1], ez = v.hi[2] - v.lo[2];
	return ex * (ey + ez) + ey * ez;                                     /* bbox.h:25-28 */
}
/* Inclusive scan over the 32 lanes of each half-wave with DPP data movement (no LDS round trips): row_shr 1, 2, 4, 8 inside the
 * 16-lane rows, then row_bcast15 hands row 0's total to row 1 (and row 2's to row 3). OWN_FIRST = false: result(l) = fold(v[0], ...,
 * v[l]) in lane order; OWN_FIRST = true: the operands the other way round, fold(v[l], v[l-1], ..., v[0]) — binJoin is associative but on
 * ties the later operand wins, so the order is part of the result. */
template <int CTRL, int ROW_MASK>
__device__ inline LaneBin dppFetch(const LaneBin &v) {
	LaneBin r;
	for (int k = 0; k < 3; ++k) {
		r.lo[k] = __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(v.lo[k]), __float_as_int(v.lo[k]), CTRL, ROW_MASK, 0xF, false));
		r.hi[k] = __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(v.hi[k]), __float_as_int(v.hi[k]), CTRL, ROW_MASK, 0xF, false));
	}
	r.n = (uint32_t)__builtin_amdgcn_update_dpp((int)v.n, (int)v.n, CTRL, ROW_MASK, 0xF, false);
	return r;
}
template <bool OWN_FIRST>
__device__ inline LaneBin scan32(LaneBin v) {
	const uint32_t inRow = threadIdx.x & 15u;
#define CRH_SCAN_STEP(N) { const LaneBin o = dppFetch<0x110 + N, 0xF>(v); if (inRow >= N) v = OWN_FIRST ? binJoin(v, o) : binJoin(o, v); }
	CRH_SCAN_STEP(1) CRH_SCAN_STEP(2) CRH_SCAN_STEP(4) CRH_SCAN_STEP(8)
#undef CRH_SCAN_STEP
	const LaneBin o = dppFetch<0x142, 0xA>(v);              /* row_bcast15 into rows 1 and 3 */
	if (threadIdx.x & 16u) v = OWN_FIRST ? binJoin(v, o) : binJoin(o, v);
	return v;
}

/* All 64 lanes call this with the node's 3 x 32 bins readable through bins / cnts (LDS or global). Returns bvh.c:148-233's
 * outcome in every lane. Lane b of each half-wave holds bin b ("forward") and bin 31 - b ("reversed"): with the reversed copy the
 * right-to-left sweep and the right child's fold are prefix scans too. */
__device__ inline LaneBin loadLaneBin(const BinKeys *bins, const uint32_t *cnts, int idx) {
	LaneBin v;
	const BinKeys k = bins[idx];
	for (int c = 0; c < 3; ++c) { v.lo[c] = keyValue(k.lo[c]); v.hi[c] = keyValue(k.hi[c]); }
	v.n = cnts[idx];
	return v;
}
__device__ inline Decision waveDecide(const BinKeys *bins, const uint32_t *cnts, const float *bounds, uint32_t n) {
	const int b = (int)(threadIdx.x & 31u);
	const bool upper = (threadIdx.x & 32u) != 0u;
	float bestCost[3];
	uint32_t bestBin[3];
	LaneBin leftOf[3];                       /* meaningful in the lower half-wave: lane b holds fold(bin 0 ... b) */
	for (int ax = 0; ax < 3; ++ax) {
		/* ONE scan serves both sweeps: the lower half-wave holds the bins forward (lane b: fold(bin 0 ... b), bvh.c:180-191), the upper
		 * half reversed (lane 32 + r: fold(bin 31, 30, ..., 31 - r), bvh.c:170-177) */
		const LaneBin both = scan32<false>(loadLaneBin(bins, cnts, ax * CRH_BVH_BINS + (upper ? 31 - b : b)));
		leftOf[ax] = both;
		const float part = both.n * binHalfArea(both);                        /* lower: left cost of bin b; upper: bins[31 - r].cost */
		const float costRnext = __shfl(part, 32 + ((30 - b) & 31), 64);       /* bins[b + 1].cost sits in upper lane 31 - (b + 1) */
		const float cost = part + costRnext;
		/* first b in 0..30 with the smallest cost below FLT_MAX (strict <, NaN never wins); none -> (FLT_MAX, bin 1). Reduced in the
		 * lower half (the xor partners of a lane below 32 are below 32), then handed to every lane */
		float c = (b <= 30 && cost < FLT_MAX) ? cost : __builtin_inff();
		int at = b;
		for (int off = 16; off > 0; off >>= 1) {
			const float c2 = __shfl_xor(c, off, 64);
			const int at2 = __shfl_xor(at, off, 64);
			if (c2 < c || (c2 == c && at2 < at)) { c = c2; at = at2; }
		}
		c = __shfl(c, 0, 64); at = __shfl(at, 0, 64);
		if (c < FLT_MAX) { bestCost[ax] = c; bestBin[ax] = (uint32_t)at + 1u; }
		else { bestCost[ax] = FLT_MAX; bestBin[ax] = 1u; }
	}
	Decision d;
	uint32_t ax = 0;                                                     /* bvh.c:195-197 */
	if (bestCost[1] < bestCost[0]) ax = 1;
	if (bestCost[2] < bestCost[ax]) ax = 2;
	uint32_t split = bestBin[ax];
	const LaneBin left = ax == 0 ? leftOf[0] : (ax == 1 ? leftOf[1] : leftOf[2]);
	Box self;
	for (int k = 0; k < 3; ++k) { self.lo[k] = bounds[2 * k]; self.hi[k] = bounds[2 * k + 1]; }
	const float leafCost = boxHalfArea(self) * (n - 1.5f);               /* bvh.c:200 */
	d.leaf = 0; d.axis = ax; d.nLeft = 0;
	for (int k = 0; k < 6; ++k) { d.childL[k] = 0.0f; d.childR[k] = 0.0f; }
	if (bestCost[ax] > leafCost) {
		if (n > CRH_BVH_MAX_LEAF) {                                      /* bvh.c:202-211: first bin boundary nearest the median, if nearer than n */
			const int diff = (int)n / 2 - (int)left.n;
			uint32_t off = (b <= 30) ? (uint32_t)(diff < 0 ? -diff : diff) : 0xFFFFFFFFu;
			int at = b;
			for (int sh = 16; sh > 0; sh >>= 1) {
				const uint32_t o2 = __shfl_xor(off, sh, 64);
				const int at2 = __shfl_xor(at, sh, 64);
				if (o2 < off || (o2 == off && at2 < at)) { off = o2; at = at2; }
			}
			off = __shfl(off, 0, 64); at = __shfl(at, 0, 64);
			if (off < n) split = (uint32_t)at + 1u;
		} else {                                                         /* bvh.c:212-214: a leaf; the child boxes are not needed */
			d.leaf = 1; d.split = split;
			return d;
		}
	}
	d.split = split;
	LaneBin l;                                                           /* bvh.c:226-233: fold(bin 0 ... split - 1) from the lower half */
	for (int k = 0; k < 3; ++k) { l.lo[k] = __shfl(left.lo[k], (int)split - 1, 64); l.hi[k] = __shfl(left.hi[k], (int)split - 1, 64); }
	l.n = __shfl(left.n, (int)split - 1, 64);
	/* right child: fold(bin split, ..., 31) in ascending order = lane 31 - split of the own-first scan over the reversed copy */
	const LaneBin r = binFrom(scan32<true>(loadLaneBin(bins, cnts, (int)ax * CRH_BVH_BINS + 31 - b)), (int)((31u - split) & 31u));
	d.nLeft = l.n;
	if (l.n == 0) d.leaf = 1;                                            /* bvh.c:218, 239-241: beginRight == begin */
	for (int k = 0; k < 3; ++k) {
		d.childL[2 * k] = l.lo[k]; d.childL[2 * k + 1] = l.hi[k];
		d.childR[2 * k] = split < CRH_BVH_BINS ? r.lo[k] : FLT_MAX; d.childR[2 * k + 1] = split < CRH_BVH_BINS ? r.hi[k] : -FLT_MAX;
	}
	return d;
}

/* ---- the decision of a TINY node (at most CRH_BVH_TINY = 8 primitives), lane-parallel (round 4) -------------------------------------------
 * Seven in ten inner nodes of a subtree hold at most eight primitives (85 % at most sixteen), and waveDecide costs them what it costs the root: three scans over 32 bins that are almost all
 * empty, behind a binning pass of 21 LDS atomics per primitive. For such a node the SAME outcome follows from the primitives directly:
 *   - bvh.c:170-191's cost of split i (left = bins 0 .. i-1) changes only behind an OCCUPIED bin, and of equal costs the lowest i wins: the candidates are i = bin(p) + 1 of
 *     the node's primitives p; a candidate whose right side is empty has the cost NaN (0 x the area of the empty box: inf), as in the sweeps, and loses;
 *   - a cost needs the two sides' boxes only as VALUES (which zero of a -0 / +0 tie survives changes an extent by the sign of a zero at most, and no comparison sees that):
 *     lane (side, axis, p) folds the boxes of its side of candidate (axis, p) over the node's primitives — 48 lanes, one pass per eight candidates;
 *   - the two child boxes, whose BIT PATTERNS go into the tree, are folded with the keys the binning uses (keyLo / keyHi: on a tie the later operand wins), the sequence number
 *     being (bin, position): the bins' own folds run in position order and bvh.c:226-233 folds the bins in ascending order.
 * Returns false (every lane the same) when no candidate is valid — the caller then takes the wave-wide path, which also holds the reference's fallbacks for that case. */
#ifndef CRH_BVH_TINY
#define CRH_BVH_TINY 8u
#ifndef CRH_BVH_FOLD_NODES
#define CRH_BVH_FOLD_NODES 16u       /* levels of at most this many nodes fold per-chunk bin rows instead of flushing with global atomics */
#endif          /* 8 or 16 (with 16 the candidates of primitives 8 .. 15 take a second pass of the same 48 lanes). Measured on the 10 M soup (profiles/r04k_bvh_variants.log): 8 -> 21.6 ms, 16 -> 21.7 ms — a node of 9 .. 16 primitives is no cheaper in two passes than in the wave-wide path */
#endif
struct TinyPrim { float lo[3], hi[3]; uint32_t bin[3], pos, pad[2]; };
__device__ inline bool tinyDecide(const TinyPrim *T, const float *bounds, uint32_t n, Decision *out) {
	const uint32_t lane = threadIdx.x;
	const uint32_t side = lane / 24u, rem = lane % 24u, a = rem >> 3;
	float c = __builtin_inff();
	uint32_t at = 0u;
	for (uint32_t pBase = 0; pBase < n; pBase += 8u) {                    /* (n <= 8: one pass) */
		const uint32_t p = pBase + (rem & 7u);
		const bool act = lane < 48u && p < n;
		const uint32_t myBin = act ? T[p].bin[a] : 0u;
		float lo[3] = {FLT_MAX, FLT_MAX, FLT_MAX}, hi[3] = {-FLT_MAX, -FLT_MAX, -FLT_MAX};
		uint32_t cnt = 0;
		for (uint32_t q = 0; q < n; ++q) {
			const bool in = act && ((T[q].bin[a] <= myBin) == (side == 0u));
			if (in) {
				for (int k = 0; k < 3; ++k) { lo[k] = pickLo(lo[k], T[q].lo[k]); hi[k] = pickHi(hi[k], T[q].hi[k]); }
				++cnt;
			}
		}
		const float ex = hi[0] - lo[0], ey = hi[1] - lo[1], ez = hi[2] - lo[2];
		const float part = cnt * (ex * (ey + ez) + ey * ez);             /* bvh.c:176 / 187: count x half area (bbox.h:25-28) */
		const float partR = __shfl(part, (int)((lane + 24u) & 63u), 64);
		const float cost = part + partR;
		/* the first split (axis-major, as bvh.c:195-197 prefers the lower axis on a tie, then the lowest bin) with the smallest cost below FLT_MAX */
		const float c1 = (act && side == 0u && myBin <= 30u && cost < FLT_MAX) ? cost : __builtin_inff();
		const uint32_t at1 = a * 32u + myBin;
		if (c1 < c || (c1 == c && at1 < at)) { c = c1; at = at1; }
	}
	for (int off = 16; off > 0; off >>= 1) {
		const float c2 = __shfl_xor(c, off, 64);
		const uint32_t at2 = (uint32_t)__shfl_xor((int)at, off, 64);
		if (c2 < c || (c2 == c && at2 < at)) { c = c2; at = at2; }
	}
	c = __shfl(c, 0, 64); at = (uint32_t)__shfl((int)at, 0, 64);
	if (!(c < FLT_MAX)) return false;
	const uint32_t ax = at >> 5, split = (at & 31u) + 1u;
	Box self;
	for (int k = 0; k < 3; ++k) { self.lo[k] = bounds[2 * k]; self.hi[k] = bounds[2 * k + 1]; }
	const float leafCost = boxHalfArea(self) * (n - 1.5f);               /* bvh.c:200 */
	if (c > leafCost) {                                                  /* bvh.c:212-214 (n <= CRH_BVH_MAX_LEAF): a leaf */
		if (lane == 0) { out->leaf = 1; out->axis = ax; out->split = split; out->nLeft = 0; }
		if (lane < 12u) { if (lane < 6u) out->childL[lane] = 0.0f; else out->childR[lane - 6u] = 0.0f; }
		return true;
	}
	const uint32_t nLeft = (uint32_t)__popcll(__ballot(lane < n && T[lane < n ? lane : 0u].bin[ax] < split));
	if (lane < 12u) {                                                    /* bvh.c:226-233: lane (child, bound) */
		const uint32_t child = lane / 6u, k = lane % 6u, c3 = k >> 1;          /* bounds are {minx, maxx, miny, maxy, minz, maxz} */
		const bool isLo = (k & 1u) == 0u;
		unsigned long long best = isLo ? CRH_KEY_LO_EMPTY : CRH_KEY_HI_EMPTY;
		for (uint32_t q = 0; q < n; ++q) {
			if ((T[q].bin[ax] < split) != (child == 0u)) continue;
			const uint32_t seq = (T[q].bin[ax] << 10) | T[q].pos;
			const unsigned long long key = isLo ? keyLo(T[q].lo[c3], seq) : keyHi(T[q].hi[c3], seq);
			if (isLo ? key < best : key > best) best = key;
		}
		const float v = keyValue(best);
		if (child == 0u) out->childL[k] = v; else out->childR[k] = v;
	}
	if (lane == 0) { out->leaf = nLeft == 0u ? 1u : 0u; out->axis = ax; out->split = split; out->nLeft = nLeft; }
	return true;
}

__device__ inline Box binBox(const BinKeys &k) {
	Box b;
	for (int c = 0; c < 3; ++c) { b.lo[c] = keyValue(k.lo[c]); b.hi[c] = keyValue(k.hi[c]); }
	return b;
}

/* ---- kernels: preparation -------------------------------------------------------------------------------------------- */
/* bvh.c:264-269 + 289-297: per-triangle box and centre, identity order, root box keys */
__global__ void k_prepare(const crh_poly *polys, const float *vertices, uint64_t vertexCount, uint32_t count, float *boxes, float *centers, int32_t *prims, unsigned long long *rootKeys) {
	__shared__ unsigned long long s_keys[6];
	if (threadIdx.x < 3) { s_keys[threadIdx.x] = CRH_KEY_LO_EMPTY; s_keys[3 + threadIdx.x] = CRH_KEY_HI_EMPTY; }
	__syncthreads();
	/* (a grid-stride loop over a few thousand workgroups: with one workgroup per 256 polygons, 39 063 of them folded their six keys into the same six words with
	 * device-scope atomics; rootKeys[6] collects "a polygon names a vertex that does not exist" — the host used to check that in a loop of its own, 10 M polygons at a time) */
	for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < count; i += gridDim.x * blockDim.x) {
		const crh_poly p = polys[i];
		if (p.v[0] < 0 || p.v[1] < 0 || p.v[2] < 0 || (uint64_t)p.v[0] >= vertexCount || (uint64_t)p.v[1] >= vertexCount || (uint64_t)p.v[2] >= vertexCount) {
			atomicMax(&rootKeys[6], 1ull);
			continue;
		}
		const float *a = vertices + 3 * (size_t)p.v[0], *b = vertices + 3 * (size_t)p.v[1], *c = vertices + 3 * (size_t)p.v[2];
		for (int k = 0; k < 3; ++k) {
			const float lo = pickLo(a[k], pickLo(b[k], c[k])), hi = pickHi(a[k], pickHi(b[k], c[k]));
			centers[3 * (size_t)i + k] = ((a[k] + b[k]) + c[k]) * (1.0f / 3.0f);               /* vector.h:186-188 */
			boxes[6 * (size_t)i + k] = lo;
			boxes[6 * (size_t)i + 3 + k] = hi;
			atomicMin(&s_keys[k], keyLo(lo, i));
			atomicMax(&s_keys[3 + k], keyHi(hi, i));
		}
		prims[i] = (int32_t)i;
	}
	__syncthreads();
	if (threadIdx.x < 3) atomicMin(&rootKeys[threadIdx.x], s_keys[threadIdx.x]);
	else if (threadIdx.x < 6) atomicMax(&rootKeys[threadIdx.x], s_keys[threadIdx.x]);
}

/* ---- kernels: large phase ------------------------------------------------------------------------------------------------ */
struct LargeNode {            /* device mirror of one node of the current level */
	float bounds[6];
	uint32_t begin, end;
};
struct Chunk { uint32_t node, start, len, single; };   /* positions [start, start+len) of prims[], all inside level node `node`; single: the node's only chunk */

__global__ void k_init_bins(BinKeys *bins, uint32_t *counts, uint32_t nBins) {
	const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= nBins) return;
	for (int c = 0; c < 3; ++c) { bins[i].lo[c] = CRH_KEY_LO_EMPTY; bins[i].hi[c] = CRH_KEY_HI_EMPTY; }
	counts[i] = 0;
}

/* (1) bins of every large node of the level: workgroup-private bins in LDS, flushed with one atomic per non-empty bin */
__global__ __launch_bounds__(256) void k_bin(const LargeNode *nodes, const Chunk *chunks, const int32_t *prims, const float *boxes, const float *centers,
											 BinKeys *gbins, uint32_t *gcounts, BinKeys *pbins, uint32_t *pcounts) {
	__shared__ BinKeys s_bins[3 * CRH_BVH_BINS];
	__shared__ uint32_t s_cnt[3 * CRH_BVH_BINS];
	const Chunk ch = chunks[blockIdx.x];
	const LargeNode nd = nodes[ch.node];
	for (uint32_t i = threadIdx.x; i < 3 * CRH_BVH_BINS; i += blockDim.x) {
		for (int c = 0; c < 3; ++c) { s_bins[i].lo[c] = CRH_KEY_LO_EMPTY; s_bins[i].hi[c] = CRH_KEY_HI_EMPTY; }
		s_cnt[i] = 0;
	}
	__syncthreads();
	for (uint32_t q = threadIdx.x; q < ch.len; q += blockDim.x) {
		const uint32_t p = ch.start + q;
		float lo[3], hi[3], ce[3];        /* boxes / centers travel with the index range (k_swap): position p holds the data of prims[p] */
		for (int k = 0; k < 3; ++k) { lo[k] = boxes[6 * (size_t)p + k]; hi[k] = boxes[6 * (size_t)p + 3 + k]; ce[k] = centers[3 * (size_t)p + k]; }
		for (int ax = 0; ax < 3; ++ax) {
			const uint32_t b = ax * CRH_BVH_BINS + binOf(ce[ax], nd.bounds[2 * ax], nd.bounds[2 * ax + 1]);
			for (int k = 0; k < 3; ++k) { atomicMin(&s_bins[b].lo[k], keyLo(lo[k], p)); atomicMax(&s_bins[b].hi[k], keyHi(hi[k], p)); }
			atomicAdd(&s_cnt[b], 1u);
		}
	}
	__syncthreads();
	for (uint32_t i = threadIdx.x; i < 3 * CRH_BVH_BINS; i += blockDim.x) {
		const size_t g = (size_t)ch.node * 3 * CRH_BVH_BINS + i;
		if (pbins) {                     /* a level of few, huge nodes: the chunk's bins go to its own row and k_fold_bins folds the rows (below) */
			pbins[(size_t)blockIdx.x * 3 * CRH_BVH_BINS + i] = s_bins[i];
			pcounts[(size_t)blockIdx.x * 3 * CRH_BVH_BINS + i] = s_cnt[i];
			continue;
		}
		if (ch.single) {                 /* the node's only chunk: these ARE its bins (most nodes of the deep levels) — plain stores, no atomics */
			gbins[g] = s_bins[i];
			gcounts[g] = s_cnt[i];
			continue;
		}
		if (!s_cnt[i]) continue;
		for (int c = 0; c < 3; ++c) { atomicMin(&gbins[g].lo[c], s_bins[i].lo[c]); atomicMax(&gbins[g].hi[c], s_bins[i].hi[c]); }
		atomicAdd(&gcounts[g], s_cnt[i]);
	}
}

/* (1b) the top levels: thousands of chunks of ONE node flushing into the same 96 x 7 words serialise on device-scope atomics (level 1 of the 10 M soup took 1.0 ms where a
 * middle level takes 0.55, round 4) — there every chunk stores its bins as a row and one block per (node, bin) folds that column. min / max of keys and a sum: any order. */
__global__ __launch_bounds__(256) void k_fold_bins(const uint32_t *nodeChunk0, const BinKeys *pbins, const uint32_t *pcounts, BinKeys *gbins, uint32_t *gcounts) {
	__shared__ BinKeys s_bin;
	__shared__ uint32_t s_n;
	const uint32_t node = blockIdx.x / (3 * CRH_BVH_BINS), i = blockIdx.x % (3 * CRH_BVH_BINS);
	if (threadIdx.x == 0) { for (int c = 0; c < 3; ++c) { s_bin.lo[c] = CRH_KEY_LO_EMPTY; s_bin.hi[c] = CRH_KEY_HI_EMPTY; } s_n = 0; }
	__syncthreads();
	BinKeys mine;
	for (int c = 0; c < 3; ++c) { mine.lo[c] = CRH_KEY_LO_EMPTY; mine.hi[c] = CRH_KEY_HI_EMPTY; }
	uint32_t n = 0;
	for (uint32_t ch = nodeChunk0[node] + threadIdx.x; ch < nodeChunk0[node + 1]; ch += blockDim.x) {
		const size_t r = (size_t)ch * 3 * CRH_BVH_BINS + i;
		const uint32_t k = pcounts[r];
		if (!k) continue;
		const BinKeys b = pbins[r];
		for (int c = 0; c < 3; ++c) { mine.lo[c] = b.lo[c] < mine.lo[c] ? b.lo[c] : mine.lo[c]; mine.hi[c] = b.hi[c] > mine.hi[c] ? b.hi[c] : mine.hi[c]; }
		n += k;
	}
	if (n) {
		for (int c = 0; c < 3; ++c) { atomicMin(&s_bin.lo[c], mine.lo[c]); atomicMax(&s_bin.hi[c], mine.hi[c]); }
		atomicAdd(&s_n, n);
	}
	__syncthreads();
	if (threadIdx.x == 0) { gbins[(size_t)node * 3 * CRH_BVH_BINS + i] = s_bin; gcounts[(size_t)node * 3 * CRH_BVH_BINS + i] = s_n; }
}

/* (2) one wave per node: lanes 0..2 sweep one axis each, lane 0 decides */
__global__ __launch_bounds__(64) void k_decide(const LargeNode *nodes, uint32_t nNodes, const BinKeys *gbins, const uint32_t *gcounts, Decision *out) {
	const uint32_t n = blockIdx.x;
	if (n >= nNodes) return;
	const LargeNode nd = nodes[n];
	const Decision d = waveDecide(gbins + (size_t)n * 3 * CRH_BVH_BINS, gcounts + (size_t)n * 3 * CRH_BVH_BINS, nd.bounds, nd.end - nd.begin);
	if (threadIdx.x == 0) out[n] = d;
}

/* (3a) per chunk: how many elements sit on the wrong side (bvh.c:95-130 stops exactly at begin + nLeft) */
__global__ __launch_bounds__(256) void k_count_misplaced(const LargeNode *nodes, const Decision *dec, const Chunk *chunks, const int32_t *prims, const float *centers,
														 uint32_t *chunkML, uint32_t *chunkMR) {
	__shared__ uint32_t s_ml, s_mr;
	if (threadIdx.x == 0) { s_ml = 0; s_mr = 0; }
	__syncthreads();
	const Chunk ch = chunks[blockIdx.x];
	const Decision d = dec[ch.node];
	if (!d.leaf) {
		const LargeNode nd = nodes[ch.node];
		const uint32_t mid = nd.begin + d.nLeft;
		uint32_t ml = 0, mr = 0;
		for (uint32_t q = threadIdx.x; q < ch.len; q += blockDim.x) {
			const uint32_t p = ch.start + q;
			const bool right = binOf(centers[3 * (size_t)p + d.axis], nd.bounds[2 * d.axis], nd.bounds[2 * d.axis + 1]) >= d.split;
			if (p < mid && right) ++ml;
			if (p >= mid && !right) ++mr;
		}
		if (ml) atomicAdd(&s_ml, ml);
		if (mr) atomicAdd(&s_mr, mr);
	}
	__syncthreads();
	if (threadIdx.x == 0) { chunkML[blockIdx.x] = s_ml; chunkMR[blockIdx.x] = s_mr; }
}

/* (3b) per node: exclusive prefix of its chunks' counts (left-misplaced from the left, right-misplaced from the right) */
__global__ __launch_bounds__(64) void k_scan_chunks(const uint32_t *nodeChunk0, uint32_t nNodes, uint32_t *chunkML, uint32_t *chunkMR, uint32_t *nodeSwaps) {
	const uint32_t n = blockIdx.x;
	if (n >= nNodes) return;
	const uint32_t lane = threadIdx.x;
	const uint32_t c0 = nodeChunk0[n], c1 = nodeChunk0[n + 1];
	uint32_t acc = 0;
	for (uint32_t c = c0; c < c1; c += 64u) {                 /* 64 chunks per step: wave-wide inclusive scan, carried total */
		const bool in = c + lane < c1;
		const uint32_t v = in ? chunkML[c + lane] : 0u;
		uint32_t incl = v;
		for (int off = 1; off < 64; off <<= 1) { const uint32_t o = __shfl_up(incl, off); if ((int)lane >= off) incl += o; }
		if (in) chunkML[c + lane] = acc + incl - v;
		acc += __shfl(incl, 63);
	}
	uint32_t accR = 0;
	for (uint32_t done = 0; c0 + done < c1; done += 64u) {     /* the same from the right end */
		const bool in = c0 + done + lane < c1;
		const uint32_t c = in ? c1 - 1u - done - lane : c0;
		const uint32_t v = in ? chunkMR[c] : 0u;
		uint32_t incl = v;
		for (int off = 1; off < 64; off <<= 1) { const uint32_t o = __shfl_up(incl, off); if ((int)lane >= off) incl += o; }
		if (in) chunkMR[c] = accR + incl - v;
		accR += __shfl(incl, 63);
	}
	if (lane == 0) nodeSwaps[n] = acc;        /* == accR: as many left-misplaced as right-misplaced */
}

/* (3c) per chunk: the k-th left-misplaced position goes to listL[begin + k], the k-th right-misplaced FROM THE RIGHT to listR[begin + k] */
__global__ __launch_bounds__(256) void k_list_misplaced(const LargeNode *nodes, const Decision *dec, const Chunk *chunks, const int32_t *prims, const float *centers,
														const uint32_t *chunkML, const uint32_t *chunkMR, uint32_t *listL, uint32_t *listR) {
	__shared__ uint32_t s_scan[256];
	const Chunk ch = chunks[blockIdx.x];
	const Decision d = dec[ch.node];
	if (d.leaf) return;
	const LargeNode nd = nodes[ch.node];
	const uint32_t mid = nd.begin + d.nLeft;
	/* each thread owns a contiguous run so that ranks follow positions */
	const uint32_t per = (ch.len + blockDim.x - 1) / blockDim.x;
	const uint32_t q0 = min(threadIdx.x * per, ch.len), q1 = min(q0 + per, ch.len);
	uint32_t ml = 0, mr = 0;
	for (uint32_t q = q0; q < q1; ++q) {
		const uint32_t p = ch.start + q;
		const bool right = binOf(centers[3 * (size_t)p + d.axis], nd.bounds[2 * d.axis], nd.bounds[2 * d.axis + 1]) >= d.split;
		if (p < mid && right) ++ml;
		if (p >= mid && !right) ++mr;
	}
	/* exclusive scan of ml (ascending threads) */
	s_scan[threadIdx.x] = ml;
	__syncthreads();
	for (uint32_t off = 1; off < blockDim.x; off <<= 1) {
		const uint32_t v = threadIdx.x >= off ? s_scan[threadIdx.x - off] : 0u;
		__syncthreads();
		s_scan[threadIdx.x] += v;
		__syncthreads();
	}
	uint32_t kL = chunkML[blockIdx.x] + s_scan[threadIdx.x] - ml;
	__syncthreads();
	/* exclusive scan of mr from the right (descending threads) */
	s_scan[threadIdx.x] = mr;
	__syncthreads();
	for (uint32_t off = 1; off < blockDim.x; off <<= 1) {
		const uint32_t v = threadIdx.x + off < blockDim.x ? s_scan[threadIdx.x + off] : 0u;
		__syncthreads();
		s_scan[threadIdx.x] += v;
		__syncthreads();
	}
	uint32_t kR = chunkMR[blockIdx.x] + s_scan[threadIdx.x];        /* inclusive from the right: rank of this thread's LAST position + 1 ... */
	for (uint32_t q = q0; q < q1; ++q) {
		const uint32_t p = ch.start + q;
		const bool right = binOf(centers[3 * (size_t)p + d.axis], nd.bounds[2 * d.axis], nd.bounds[2 * d.axis + 1]) >= d.split;
		if (p < mid && right) listL[nd.begin + kL++] = p;
		if (p >= mid && !right) listR[nd.begin + --kR] = p;          /* ... so ascending positions get descending ranks */
	}
}

/* (3d) the swaps of bvh.c:121-123, all independent */
__global__ __launch_bounds__(256) void k_swap(const LargeNode *nodes, const Decision *dec, const Chunk *chunks, const uint32_t *nodeSwaps, const uint32_t *listL, const uint32_t *listR, int32_t *prims, float *boxes, float *centers) {
	const Chunk ch = chunks[blockIdx.x];
	if (dec[ch.node].leaf) return;
	const LargeNode nd = nodes[ch.node];
	const uint32_t m = nodeSwaps[ch.node];
	for (uint32_t q = threadIdx.x; q < ch.len; q += blockDim.x) {
		const uint32_t k = ch.start + q - nd.begin;
		if (k >= m) continue;
		const uint32_t a = listL[nd.begin + k], b = listR[nd.begin + k];
		const int32_t t = prims[a]; prims[a] = prims[b]; prims[b] = t;
		for (int c = 0; c < 6; ++c) { const float f = boxes[6 * (size_t)a + c]; boxes[6 * (size_t)a + c] = boxes[6 * (size_t)b + c]; boxes[6 * (size_t)b + c] = f; }
		for (int c = 0; c < 3; ++c) { const float f = centers[3 * (size_t)a + c]; centers[3 * (size_t)a + c] = centers[3 * (size_t)b + c]; centers[3 * (size_t)b + c] = f; }
	}
}

/* ---- kernel: small phase ------------------------------------------------------------------------------------------------- */
struct SmallRoot { float bounds[6]; uint32_t begin, end, depth, forceLeaf; uint32_t localOff, pad[3]; };

/* One wave builds one subtree depth-first (left first), exactly like the recursion it replaces. Its nodes go to
 * local[localOff ...] (a subtree over n primitives has at most max(2n-1, 1) nodes); local index 0 is the subtree root, child
 * pairs are appended in allocation order, inner nodes point at LOCAL indices. Roots that are leaves by rule (depth limit,
 * fewer than two primitives, or a large node whose split left nothing on the left: forceLeaf) are emitted directly. */
__global__ __launch_bounds__(64) void k_small(const SmallRoot *roots, uint32_t nRoots, int32_t *prims, const float *boxes, const float *centers,
											  crh_bvh_node *local, uint32_t *localCount, unsigned long long *prof, uint32_t *overflow) {
	/* the index range is kept as SLOT numbers (= position at load time): boxes[] / centers[] are in position order (k_swap moves them
	 * with the indices), so slot s of this subtree has its data at base + s — 18 KB of contiguous, cache-resident records per subtree.
	 * (Keeping those records in LDS instead was measured: 30 KB per wave leaves 5 waves per CU, and the build got slower.) */
	__shared__ uint16_t s_prim[CRH_BVH_SMALL];
	__shared__ BinKeys s_bins[3 * CRH_BVH_BINS];
	__shared__ uint32_t s_cnt[3 * CRH_BVH_BINS];
	__shared__ Decision s_dec;
	/* the rank lists of the partition live where the bins were (the decision has consumed them): 8 KB of LDS per wave, 19 waves per CU */
	static_assert(2 * CRH_BVH_SMALL * sizeof(uint16_t) <= sizeof(BinKeys) * 3 * CRH_BVH_BINS, "rank lists alias the bins");
	uint16_t *const s_listL = reinterpret_cast<uint16_t *>(s_bins), *const s_listR = s_listL + CRH_BVH_SMALL;
	struct Job { uint16_t node, first, last, depth; float bounds[6]; };
	__shared__ Job s_stack[CRH_BVH_MAX_DEPTH + 8];          /* depth-first: one waiting right sibling per level, plus the job on top */
	const uint32_t r = blockIdx.x;
	if (r >= nRoots) return;
	const uint32_t lane = threadIdx.x;
	const SmallRoot root = roots[r];
	const uint32_t base = root.begin, total = root.end - root.begin;
	crh_bvh_node *out = local + root.localOff;
	if (root.forceLeaf || root.depth >= CRH_BVH_MAX_DEPTH || total < 2 || total > CRH_BVH_SMALL) {   /* > SMALL only arrives with one of the former */
		if (lane == 0) {
			crh_bvh_node n0;
			memset(&n0, 0, sizeof(n0));
			for (int k = 0; k < 6; ++k) n0.bounds[k] = root.bounds[k];
			n0.first = base; n0.count_leaf = (total & 0x3FFFFFFFu) | (1u << 30);
			out[0] = n0;
			localCount[r] = 1;
		}
		return;
	}
	for (uint32_t i = lane; i < total; i += 64) s_prim[i] = (uint16_t)i;
	if (lane == 0) {
		crh_bvh_node n0;
		memset(&n0, 0, sizeof(n0));
		for (int k = 0; k < 6; ++k) n0.bounds[k] = root.bounds[k];
		out[0] = n0;
		Job j0;
		j0.node = 0; j0.first = 0; j0.last = (uint16_t)total; j0.depth = (uint16_t)root.depth;
		for (int k = 0; k < 6; ++k) j0.bounds[k] = root.bounds[k];
		s_stack[0] = j0;
	}
	__syncthreads();
	uint32_t sp = 1, used = 1;
	const uint32_t cap = 2u * total - 1u;                     /* this subtree's share of the node array (the reference's bound, bvh.c:271) */
	unsigned long long pt[8] = {0, 0, 0, 0, 0, 0, 0, 0};     /* CRH_BVH_TRACE: wall-clock ticks per part (lane 0 of every wave) */
	unsigned long long pc = prof ? wall_clock64() : 0ull;
#define CRH_PROF(i) do { if (prof) { const unsigned long long now_ = wall_clock64(); pt[i] += now_ - pc; pc = now_; } } while (0)
	while (sp) {
		const Job j = s_stack[--sp];
		const uint32_t first = j.first, last = j.last, n = last - first;
		float bounds[6];
		for (int k = 0; k < 6; ++k) bounds[k] = j.bounds[k];
		bool leaf = (j.depth >= CRH_BVH_MAX_DEPTH || n < 2);                 /* bvh.c:143-146 */
		bool decided = false;
		if (!leaf && n <= CRH_BVH_TINY) {                                    /* tiny node: decided from its primitives directly (tinyDecide) */
			TinyPrim *T = reinterpret_cast<TinyPrim *>(s_bins);
			if (lane < n) {
				const uint32_t q = first + lane;
				const size_t at = (size_t)base + s_prim[q];
				TinyPrim t;
				for (int k = 0; k < 3; ++k) { t.lo[k] = boxes[6 * at + k]; t.hi[k] = boxes[6 * at + 3 + k]; t.bin[k] = binOf(centers[3 * at + k], bounds[2 * k], bounds[2 * k + 1]); }
				t.pos = q; t.pad[0] = t.pad[1] = 0;
				T[lane] = t;
			}
			__syncthreads();
			decided = tinyDecide(T, bounds, n, &s_dec);
			__syncthreads();
			CRH_PROF(3);
			if (decided) {
				leaf = s_dec.leaf != 0;
				if (!leaf && used + 2u > cap) { leaf = true; if (lane == 0) atomicOr(overflow, 1u); }
			}
		}
		if (!leaf && !decided) {
			for (uint32_t i = lane; i < 3 * CRH_BVH_BINS; i += 64) {
				for (int c = 0; c < 3; ++c) { s_bins[i].lo[c] = CRH_KEY_LO_EMPTY; s_bins[i].hi[c] = CRH_KEY_HI_EMPTY; }
				s_cnt[i] = 0;
			}
			__syncthreads();
			for (uint32_t q = first + lane; q < last; q += 64) {             /* bvh.c:158-165; the tie order is the position in the range */
				const size_t at = (size_t)base + s_prim[q];
				float lo[3], hi[3], ce[3];
				for (int k = 0; k < 3; ++k) { lo[k] = boxes[6 * at + k]; hi[k] = boxes[6 * at + 3 + k]; ce[k] = centers[3 * at + k]; }
				for (int ax = 0; ax < 3; ++ax) {
					const uint32_t b = ax * CRH_BVH_BINS + binOf(ce[ax], bounds[2 * ax], bounds[2 * ax + 1]);
					for (int k = 0; k < 3; ++k) { atomicMin(&s_bins[b].lo[k], keyLo(lo[k], q)); atomicMax(&s_bins[b].hi[k], keyHi(hi[k], q)); }
					atomicAdd(&s_cnt[b], 1u);
				}
			}
			__syncthreads();
			CRH_PROF(2);
			{
				const Decision dd = waveDecide(s_bins, s_cnt, bounds, n);
				if (lane == 0) s_dec = dd;
			}
			__syncthreads();
			CRH_PROF(3);
			leaf = s_dec.leaf != 0;
			/* The reference splits a node whose primitives ALL land on the left (bvh.c:220 asks only beginRight > begin): clusters of more
			 * than 16 coincident primitives become chains of (everything | nothing) splits, and with enough of them the tree has more than
			 * 2 n - 1 nodes — the reference overflows its heap array there. Here the subtree stops (memory stays intact) and the build
			 * reports that no reference tree exists. */
			if (!leaf && used + 2u > cap) { leaf = true; if (lane == 0) atomicOr(overflow, 1u); }
		}
		if (leaf) {
			if (lane == 0) { out[j.node].first = base + first; out[j.node].count_leaf = (n & 0x3FFFFFFFu) | (1u << 30); }
			__syncthreads();
			CRH_PROF(0);
			continue;
		}
		const Decision d = s_dec;
		const uint32_t mid = first + d.nLeft;
		const float aLo = bounds[2 * d.axis], aHi = bounds[2 * d.axis + 1];
		const float *ce = centers + 3 * (size_t)base + d.axis;          /* centre of slot s along the split axis: ce[3 * s] */
		/* bvh.c:95-130: k-th misplaced from the left <-> k-th misplaced from the right */
		uint32_t kL = 0, kR = 0;
		for (uint32_t q0 = first; q0 < mid; q0 += 64) {
			const uint32_t q = q0 + lane;
			const bool mis = q < mid && binOf(ce[3u * s_prim[q < mid ? q : first]], aLo, aHi) >= d.split;
			const unsigned long long m = __ballot(mis);
			if (mis) s_listL[kL + __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u))] = (uint16_t)q;
			kL += (uint32_t)__popcll(m);
		}
		for (uint32_t done = 0; mid + done < last; done += 64) {             /* from the right end downwards */
			const uint32_t off = done + lane;
			const bool in = mid + off < last;
			const uint32_t q = in ? last - 1u - off : mid;
			const bool mis = in && binOf(ce[3u * s_prim[q]], aLo, aHi) < d.split;
			const unsigned long long m = __ballot(mis);
			if (mis) s_listR[kR + __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u))] = (uint16_t)q;
			kR += (uint32_t)__popcll(m);
		}
		__syncthreads();
		for (uint32_t k = lane; k < kL; k += 64) {
			const uint32_t a = s_listL[k], b = s_listR[k];
			const uint16_t t = s_prim[a]; s_prim[a] = s_prim[b]; s_prim[b] = t;
		}
		if (lane == 0) {                                                     /* bvh.c:219-238 */
			crh_bvh_node l, rr;
			memset(&l, 0, sizeof(l)); memset(&rr, 0, sizeof(rr));
			for (int k = 0; k < 6; ++k) { l.bounds[k] = d.childL[k]; rr.bounds[k] = d.childR[k]; }
			out[used] = l; out[used + 1] = rr;
			out[j.node].first = used;
			out[j.node].count_leaf = 0;
			Job jr, jl;
			jr.node = (uint16_t)(used + 1); jr.first = (uint16_t)mid; jr.last = (uint16_t)last; jr.depth = (uint16_t)(j.depth + 1);
			jl.node = (uint16_t)used; jl.first = (uint16_t)first; jl.last = (uint16_t)mid; jl.depth = (uint16_t)(j.depth + 1);
			for (int k = 0; k < 6; ++k) { jr.bounds[k] = d.childR[k]; jl.bounds[k] = d.childL[k]; }
			s_stack[sp] = jr;              /* right waits ... */
			s_stack[sp + 1] = jl;          /* ... left is built first */
		}
		sp += 2; used += 2;
		__syncthreads();
		CRH_PROF(4);
	}
	{	/* the index range in its new order: every old entry is read before any is overwritten */
		int32_t moved[CRH_BVH_SMALL / 64];
		for (uint32_t k = 0; k < CRH_BVH_SMALL / 64; ++k) { const uint32_t i = k * 64 + lane; moved[k] = i < total ? prims[base + s_prim[i]] : 0; }
		__syncthreads();
		for (uint32_t k = 0; k < CRH_BVH_SMALL / 64; ++k) { const uint32_t i = k * 64 + lane; if (i < total) prims[base + i] = moved[k]; }
	}
	if (lane == 0) localCount[r] = used;
	if (prof && lane == 0) for (int i = 0; i < 8; ++i) if (pt[i]) atomicAdd(&prof[i], pt[i]);
#undef CRH_PROF
}

/* final numbering: subtree r's local node 0 is global node rootId[r]; local node j >= 1 is global node firstId[r] + j - 1 */
__global__ __launch_bounds__(256) void k_emit(const SmallRoot *roots, uint32_t nRoots, const uint32_t *localCount, const uint32_t *rootId, const uint32_t *firstId,
											  const crh_bvh_node *local, crh_bvh_node *nodes) {
	for (uint32_t r = blockIdx.x; r < nRoots; r += gridDim.x) {
		const crh_bvh_node *src = local + roots[r].localOff;
		const uint32_t cnt = localCount[r], first = firstId[r];
		for (uint32_t j = threadIdx.x; j < cnt; j += blockDim.x) {
			crh_bvh_node n = src[j];
			if (!((n.count_leaf >> 30) & 1u)) n.first = first + n.first - 1u;
			nodes[j == 0 ? rootId[r] : first + j - 1u] = n;
		}
	}
}

}  // namespace crhb

/* ---- host driver -------------------------------------------------------------------------------------------------------- */
using namespace crhb;

namespace {
struct UpperNode {
	float bounds[6];
	uint32_t begin, end, depth;
	int32_t left = -1, right = -1;     /* upper-tree indices of the children when split */
	int32_t small = -1;                /* index into the small-root list when handed to the small phase */
	uint32_t id = 0;                   /* final node index */
};
template <class T> struct DevBuf {
	T *p = nullptr;
	DevBuf() = default;
	DevBuf(const DevBuf &) = delete;                 /* owning: a copy-assignment would drop the live allocation without freeing it */
	DevBuf &operator=(const DevBuf &) = delete;
	~DevBuf() { release(); }
	void release() { if (p) (void)hipFree(p); p = nullptr; }
	hipError_t alloc(size_t n) { release(); return hipMalloc((void **)&p, std::max<size_t>(n, 1) * sizeof(T)); }
};
struct DecisionEvent {             /* recorded behind a level's decisions: the host prepares the next level while the partition kernels run */
	hipEvent_t e = nullptr;
	~DecisionEvent() { if (e) (void)hipEventDestroy(e); }
};
#define BVH_TRY(expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) return crh_internal_fail(CRH_ERR_HIP, (std::string("crh_bvh_build_triangles: ") + #expr + ": " + hipGetErrorString(e_)).c_str()); } while (0)
}  // namespace

extern "C" int crh_bvh_build_triangles(crh_ctx *ctx, const crh_poly *polys, uint32_t poly_count, const float *vertices, uint64_t vertex_count,
										crh_bvh_node *nodes_out, int32_t *prim_indices_out, uint32_t *node_count_out, crh_bvh_build_stats *stats) {
	if (!ctx || !node_count_out || (poly_count && (!polys || !vertices || !nodes_out || !prim_indices_out)))
		return crh_internal_fail(CRH_ERR_INVALID, "crh_bvh_build_triangles: NULL argument");
	if (stats) memset(stats, 0, sizeof(*stats));
	if (poly_count < 1) { *node_count_out = 0; return CRH_OK; }                               /* bvh.c:250-256 */
	if (poly_count >= 0x40000000u) return crh_internal_fail(CRH_ERR_UNSUPPORTED, "crh_bvh_build_triangles: more than 2^30 primitives");
	BVH_TRY(hipSetDevice(crh_internal_device(ctx)));
	hipStream_t st = (hipStream_t)crh_internal_stream(ctx);
	const auto t0 = std::chrono::steady_clock::now();
	const uint32_t N = poly_count;

	DevBuf<crh_poly> dPolys; DevBuf<float> dVerts, dBoxes, dCenters; DevBuf<int32_t> dPrims; DevBuf<unsigned long long> dRootKeys;
	DevBuf<uint32_t> dListL, dListR; DevBuf<crh_bvh_node> dLocal, dNodes;
	BVH_TRY(dPolys.alloc(N)); BVH_TRY(dVerts.alloc((size_t)vertex_count * 3)); BVH_TRY(dBoxes.alloc((size_t)N * 6)); BVH_TRY(dCenters.alloc((size_t)N * 3));
	BVH_TRY(dPrims.alloc(N)); BVH_TRY(dRootKeys.alloc(7)); BVH_TRY(dListL.alloc(N)); BVH_TRY(dListR.alloc(N));
	BVH_TRY(dNodes.alloc(2 * (size_t)N));
	BVH_TRY(hipMemcpyAsync(dPolys.p, polys, (size_t)N * sizeof(crh_poly), hipMemcpyHostToDevice, st));
	BVH_TRY(hipMemcpyAsync(dVerts.p, vertices, (size_t)vertex_count * 3 * sizeof(float), hipMemcpyHostToDevice, st));
	const unsigned long long rootInit[7] = {CRH_KEY_LO_EMPTY, CRH_KEY_LO_EMPTY, CRH_KEY_LO_EMPTY, CRH_KEY_HI_EMPTY, CRH_KEY_HI_EMPTY, CRH_KEY_HI_EMPTY, 0ull};      /* [6]: a vertex index out of range */
	BVH_TRY(hipMemcpyAsync(dRootKeys.p, rootInit, sizeof(rootInit), hipMemcpyHostToDevice, st));
	BVH_TRY(hipStreamSynchronize(st));
	const auto t1 = std::chrono::steady_clock::now();

	hipLaunchKernelGGL(k_prepare, dim3(std::min<uint32_t>((N + 255) / 256, 4096u)), dim3(256), 0, st, dPolys.p, dVerts.p, vertex_count, N, dBoxes.p, dCenters.p, dPrims.p, dRootKeys.p);
	unsigned long long rootKeys[7];
	BVH_TRY(hipMemcpyAsync(rootKeys, dRootKeys.p, sizeof(rootKeys), hipMemcpyDeviceToHost, st));
	BVH_TRY(hipStreamSynchronize(st));
	if (rootKeys[6]) return crh_internal_fail(CRH_ERR_INVALID, "crh_bvh_build_triangles: vertex index out of range");

	std::vector<UpperNode> upper;
	upper.reserve(1024);
	{
		UpperNode r;
		for (int k = 0; k < 3; ++k) { r.bounds[2 * k] = keyValue(rootKeys[k]); r.bounds[2 * k + 1] = keyValue(rootKeys[3 + k]); }
		r.begin = 0; r.end = N; r.depth = 0;
		upper.push_back(r);
	}
	std::vector<SmallRoot> smallRoots;
	std::vector<uint32_t> level;          /* upper indices of the large nodes of the current level */
	size_t localNodes = 0;
	auto route = [&](uint32_t u, bool forceLeaf) {   /* a new node is either large (next level) or the root of a wave-built subtree */
		const UpperNode &n = upper[u];
		const uint32_t cnt = n.end - n.begin;
		if (!forceLeaf && cnt > CRH_BVH_SMALL && n.depth < CRH_BVH_MAX_DEPTH) { level.push_back(u); return; }
		SmallRoot s;
		memset(&s, 0, sizeof(s));
		memcpy(s.bounds, n.bounds, sizeof(s.bounds));
		s.begin = n.begin; s.end = n.end; s.depth = n.depth; s.forceLeaf = forceLeaf ? 1u : 0u;
		s.localOff = (uint32_t)localNodes;
		localNodes += (cnt > CRH_BVH_SMALL || cnt < 2) ? 1 : 2 * (size_t)cnt - 1;
		upper[u].small = (int32_t)smallRoots.size();
		smallRoots.push_back(s);
	};
	route(0, false);

	uint32_t levels = 0;
	std::vector<uint32_t> level_leaves;
	const bool trace = getenv("CRH_BVH_TRACE") != nullptr;
	auto tl = std::chrono::steady_clock::now();
	DevBuf<LargeNode> dLevel; DevBuf<Chunk> dChunks; DevBuf<BinKeys> dBins; DevBuf<uint32_t> dCounts, dChunkML, dChunkMR, dNodeChunk0, dNodeSwaps; DevBuf<Decision> dDec;
	size_t capNodes = 0, capChunks = 0;
	DevBuf<BinKeys> dRowBins; DevBuf<uint32_t> dRowCounts;          /* per-chunk bin rows of the top levels (k_fold_bins) */
	const size_t capRows = (size_t)N / CRH_BVH_CHUNK + CRH_BVH_FOLD_NODES + 1;
	const uint32_t foldMinChunks = getenv("CRH_BVH_FOLD_MIN_CHUNKS") ? (uint32_t)atoi(getenv("CRH_BVH_FOLD_MIN_CHUNKS")) : 64u;          /* (tests: 1 sends small meshes through k_fold_bins) */
	{	/* level scratch for the whole build at once (a level's nodes hold > CRH_BVH_SMALL primitives each, its chunks are whole or node tails):
		 * growing it level by level put a hipFree + hipMalloc into every second level */
		capNodes = (size_t)N / CRH_BVH_SMALL + 64;
		capChunks = (size_t)N / CRH_BVH_CHUNK + capNodes + 64;
		BVH_TRY(dLevel.alloc(capNodes)); BVH_TRY(dBins.alloc(capNodes * 3 * CRH_BVH_BINS)); BVH_TRY(dCounts.alloc(capNodes * 3 * CRH_BVH_BINS));
		BVH_TRY(dDec.alloc(capNodes)); BVH_TRY(dNodeChunk0.alloc(capNodes + 1)); BVH_TRY(dNodeSwaps.alloc(capNodes));
		BVH_TRY(dChunks.alloc(capChunks)); BVH_TRY(dChunkML.alloc(capChunks)); BVH_TRY(dChunkMR.alloc(capChunks));
	}
	const uint32_t chunkLen = CRH_BVH_CHUNK;      /* larger chunks for the top levels (fewer global-atomic flushes) were measured: slower from level 3 on */
	/* A level's lists go up from, and its decisions come down into, page-locked memory of the context (round 4: the copies are asynchronous for real), and the decisions are
	 * fetched right behind k_decide: the host routes the children and writes the next level's lists WHILE the four partition kernels of this level run — until then every level
	 * ended with the device idle for a stream synchronisation, a loop over the level on the host, three staged copies and seven launches. */
	DecisionEvent decided;
	BVH_TRY(hipEventCreateWithFlags(&decided.e, hipEventDisableTiming));
	LargeNode *hNodes = nullptr; Chunk *hChunks = nullptr; uint32_t *hChunk0 = nullptr; Decision *hDec = nullptr;
	auto carveStaging = [&]() -> bool {
		const size_t bytes = capNodes * (sizeof(LargeNode) + sizeof(Decision)) + capChunks * sizeof(Chunk) + (capNodes + 1) * sizeof(uint32_t) + 64;
		char *base = (char *)crh_internal_pinned(ctx, bytes);
		if (!base) return false;
		hDec = (Decision *)base; base += capNodes * sizeof(Decision);
		hNodes = (LargeNode *)base; base += capNodes * sizeof(LargeNode);
		hChunks = (Chunk *)base; base += capChunks * sizeof(Chunk);
		hChunk0 = (uint32_t *)base;
		return true;
	};
	if (!carveStaging()) return crh_internal_fail(CRH_ERR_HIP, "crh_bvh_build_triangles: no page-locked host memory for the level lists");
	while (!level.empty()) {
		++levels;
		const std::vector<uint32_t> cur = level;
		level.clear();
		const uint32_t nNodes = (uint32_t)cur.size();
		size_t wantChunks = 0;
		for (uint32_t i = 0; i < nNodes; ++i) wantChunks += (upper[cur[i]].end - upper[cur[i]].begin + chunkLen - 1) / chunkLen;
		if (nNodes > capNodes || wantChunks > capChunks) {          /* (rare: chains of lopsided splits) the previous level's partition kernels still use the scratch */
			BVH_TRY(hipStreamSynchronize(st));
			if (nNodes > capNodes) {
				capNodes = (size_t)nNodes * 2;
				BVH_TRY(dLevel.alloc(capNodes)); BVH_TRY(dBins.alloc(capNodes * 3 * CRH_BVH_BINS)); BVH_TRY(dCounts.alloc(capNodes * 3 * CRH_BVH_BINS));
				BVH_TRY(dDec.alloc(capNodes)); BVH_TRY(dNodeChunk0.alloc(capNodes + 1)); BVH_TRY(dNodeSwaps.alloc(capNodes));
			}
			if (wantChunks > capChunks) {
				capChunks = wantChunks * 2;
				BVH_TRY(dChunks.alloc(capChunks)); BVH_TRY(dChunkML.alloc(capChunks)); BVH_TRY(dChunkMR.alloc(capChunks));
			}
			if (!carveStaging()) return crh_internal_fail(CRH_ERR_HIP, "crh_bvh_build_triangles: no page-locked host memory for the level lists");
		}
		uint32_t nChunks = 0;
		for (uint32_t i = 0; i < nNodes; ++i) {
			const UpperNode &u = upper[cur[i]];
			memcpy(hNodes[i].bounds, u.bounds, sizeof(u.bounds));
			hNodes[i].begin = u.begin; hNodes[i].end = u.end;
			hChunk0[i] = nChunks;
			for (uint32_t s = u.begin; s < u.end; s += chunkLen) hChunks[nChunks++] = Chunk{i, s, std::min(chunkLen, u.end - s), u.end - u.begin <= chunkLen ? 1u : 0u};
		}
		hChunk0[nNodes] = nChunks;
		BVH_TRY(hipMemcpyAsync(dLevel.p, hNodes, nNodes * sizeof(LargeNode), hipMemcpyHostToDevice, st));
		BVH_TRY(hipMemcpyAsync(dChunks.p, hChunks, nChunks * sizeof(Chunk), hipMemcpyHostToDevice, st));
		BVH_TRY(hipMemcpyAsync(dNodeChunk0.p, hChunk0, (nNodes + 1) * sizeof(uint32_t), hipMemcpyHostToDevice, st));
		const uint32_t nBins = nNodes * 3 * CRH_BVH_BINS;
		const bool foldRows = nNodes <= CRH_BVH_FOLD_NODES && nChunks >= foldMinChunks * nNodes && nChunks <= capRows;          /* few nodes of very many chunks each */
		if (foldRows) {
			if (!dRowBins.p) { BVH_TRY(dRowBins.alloc(capRows * 3 * CRH_BVH_BINS)); BVH_TRY(dRowCounts.alloc(capRows * 3 * CRH_BVH_BINS)); }
			hipLaunchKernelGGL(k_bin, dim3(nChunks), dim3(256), 0, st, dLevel.p, dChunks.p, dPrims.p, dBoxes.p, dCenters.p, dBins.p, dCounts.p, dRowBins.p, dRowCounts.p);
			hipLaunchKernelGGL(k_fold_bins, dim3(nBins), dim3(256), 0, st, dNodeChunk0.p, dRowBins.p, dRowCounts.p, dBins.p, dCounts.p);
		} else {
			hipLaunchKernelGGL(k_init_bins, dim3((nBins + 255) / 256), dim3(256), 0, st, dBins.p, dCounts.p, nBins);
			hipLaunchKernelGGL(k_bin, dim3(nChunks), dim3(256), 0, st, dLevel.p, dChunks.p, dPrims.p, dBoxes.p, dCenters.p, dBins.p, dCounts.p, (BinKeys *)nullptr, (uint32_t *)nullptr);
		}
		hipLaunchKernelGGL(k_decide, dim3(nNodes), dim3(64), 0, st, dLevel.p, nNodes, dBins.p, dCounts.p, dDec.p);
		BVH_TRY(hipMemcpyAsync(hDec, dDec.p, nNodes * sizeof(Decision), hipMemcpyDeviceToHost, st));
		BVH_TRY(hipEventRecord(decided.e, st));
		hipLaunchKernelGGL(k_count_misplaced, dim3(nChunks), dim3(256), 0, st, dLevel.p, dDec.p, dChunks.p, dPrims.p, dCenters.p, dChunkML.p, dChunkMR.p);
		hipLaunchKernelGGL(k_scan_chunks, dim3(nNodes), dim3(64), 0, st, dNodeChunk0.p, nNodes, dChunkML.p, dChunkMR.p, dNodeSwaps.p);
		hipLaunchKernelGGL(k_list_misplaced, dim3(nChunks), dim3(256), 0, st, dLevel.p, dDec.p, dChunks.p, dPrims.p, dCenters.p, dChunkML.p, dChunkMR.p, dListL.p, dListR.p);
		hipLaunchKernelGGL(k_swap, dim3(nChunks), dim3(256), 0, st, dLevel.p, dDec.p, dChunks.p, dNodeSwaps.p, dListL.p, dListR.p, dPrims.p, dBoxes.p, dCenters.p);
		BVH_TRY(hipGetLastError());
		BVH_TRY(hipEventSynchronize(decided.e));
		for (uint32_t i = 0; i < nNodes; ++i) {
			const uint32_t u = cur[i];
			const Decision &d = hDec[i];
			if (d.leaf) {          /* bvh.c:239-241: nothing ended up on the left — the node stays a leaf whatever its size */
				level_leaves.push_back(u);
				continue;
			}
			UpperNode l, r;
			memcpy(l.bounds, d.childL, sizeof(l.bounds)); memcpy(r.bounds, d.childR, sizeof(r.bounds));
			l.begin = upper[u].begin; l.end = upper[u].begin + d.nLeft; l.depth = upper[u].depth + 1;
			r.begin = l.end; r.end = upper[u].end; r.depth = l.depth;
			const uint32_t li = (uint32_t)upper.size();
			upper.push_back(l); upper.push_back(r);
			upper[u].left = (int32_t)li; upper[u].right = (int32_t)li + 1;
			route(li, false); route(li + 1, false);
		}
		for (uint32_t u : level_leaves) route(u, true);
		level_leaves.clear();
		if (trace) { BVH_TRY(hipStreamSynchronize(st));          /* (tracing gives up the overlap: a level's time is all of its kernels) */
			const auto tn = std::chrono::steady_clock::now(); fprintf(stderr, "bvh level %u: %u nodes %u chunks%s %.3f ms\n", levels, nNodes, nChunks, foldRows ? " (bin rows folded)" : "", std::chrono::duration<double, std::milli>(tn - tl).count()); tl = tn; }
	}

	/* small phase: one wave per subtree */
	const uint32_t nRoots = (uint32_t)smallRoots.size();
	DevBuf<SmallRoot> dRoots; DevBuf<uint32_t> dLocalCount, dRootId, dFirstId;
	BVH_TRY(dLocal.alloc(localNodes)); BVH_TRY(dRoots.alloc(nRoots)); BVH_TRY(dLocalCount.alloc(nRoots)); BVH_TRY(dRootId.alloc(nRoots)); BVH_TRY(dFirstId.alloc(nRoots));
	/* (the level lists are dead: their copies ran before the last decisions came back) the same page-locked scratch carries the subtree roots, their node counts and their numbers */
	char *const hSmall = (char *)crh_internal_pinned(ctx, (size_t)nRoots * (sizeof(SmallRoot) + 3 * sizeof(uint32_t)) + 64);
	if (!hSmall) return crh_internal_fail(CRH_ERR_HIP, "crh_bvh_build_triangles: no page-locked host memory for the subtree lists");
	uint32_t *const localCount = (uint32_t *)hSmall, *const rootId = localCount + nRoots, *const firstId = rootId + nRoots, *const overflowedWord = firstId + nRoots;
	SmallRoot *const hRoots = (SmallRoot *)(hSmall + (((size_t)nRoots * 3 + 1) * sizeof(uint32_t) + 15u & ~(size_t)15u));
	memcpy(hRoots, smallRoots.data(), nRoots * sizeof(SmallRoot));
	BVH_TRY(hipMemcpyAsync(dRoots.p, hRoots, nRoots * sizeof(SmallRoot), hipMemcpyHostToDevice, st));
	DevBuf<unsigned long long> dProf;
	if (trace) { BVH_TRY(dProf.alloc(8)); BVH_TRY(hipMemsetAsync(dProf.p, 0, 8 * sizeof(unsigned long long), st)); }
	DevBuf<uint32_t> dOverflow;
	BVH_TRY(dOverflow.alloc(1)); BVH_TRY(hipMemsetAsync(dOverflow.p, 0, sizeof(uint32_t), st));
	hipLaunchKernelGGL(k_small, dim3(nRoots), dim3(64), 0, st, dRoots.p, nRoots, dPrims.p, dBoxes.p, dCenters.p, dLocal.p, dLocalCount.p, trace ? dProf.p : nullptr, dOverflow.p);
	BVH_TRY(hipGetLastError());
	BVH_TRY(hipMemcpyAsync(localCount, dLocalCount.p, nRoots * sizeof(uint32_t), hipMemcpyDeviceToHost, st));
	BVH_TRY(hipMemcpyAsync(overflowedWord, dOverflow.p, sizeof(uint32_t), hipMemcpyDeviceToHost, st));
	BVH_TRY(hipStreamSynchronize(st));
	const uint32_t overflowed = *overflowedWord;
	static const char *const kNoReferenceTree = "crh_bvh_build_triangles: degenerate mesh — the reference's builder needs more than the 2 n - 1 nodes it allocates "
		"(bvh.c:271; clusters of more than 16 coincident primitives are split into (all | none) down to the depth limit): no reference tree exists";
	if (overflowed) return crh_internal_fail(CRH_ERR_UNSUPPORTED, kNoReferenceTree);
	if (trace) {
		const auto tn = std::chrono::steady_clock::now(); fprintf(stderr, "bvh small phase: %u subtrees %.3f ms\n", nRoots, std::chrono::duration<double, std::milli>(tn - tl).count()); tl = tn;
		unsigned long long hp[8];
		BVH_TRY(hipMemcpy(hp, dProf.p, sizeof(hp), hipMemcpyDeviceToHost));
		fprintf(stderr, "  per subtree (10 ns ticks summed over waves / subtrees): leaf pops %.1f us, bin %.1f us, decide %.1f us, partition+emit %.1f us\n",
				hp[0] * 0.01 / nRoots, hp[2] * 0.01 / nRoots, hp[3] * 0.01 / nRoots, hp[4] * 0.01 / nRoots);
	}

	/* numbering (bvh.c:221-223, 237-238): depth-first, a pair per split, left subtree before the right one */
	uint32_t next = 1;
	{
		std::vector<uint32_t> stack{0u};
		upper[0].id = 0;
		while (!stack.empty()) {
			const uint32_t u = stack.back();
			stack.pop_back();
			UpperNode &n = upper[u];
			if (n.small >= 0) {
				rootId[n.small] = n.id;
				firstId[n.small] = next;
				next += localCount[n.small] - 1;
				continue;
			}
			upper[n.left].id = next; upper[n.right].id = next + 1;
			next += 2;
			stack.push_back((uint32_t)n.right);
			stack.push_back((uint32_t)n.left);
		}
	}
	const uint32_t nodeCount = next;
	if ((size_t)nodeCount > 2 * (size_t)N - 1) return crh_internal_fail(CRH_ERR_UNSUPPORTED, kNoReferenceTree);      /* the same, with the chains in the upper tree */
	BVH_TRY(hipMemcpyAsync(dRootId.p, rootId, nRoots * sizeof(uint32_t), hipMemcpyHostToDevice, st));
	BVH_TRY(hipMemcpyAsync(dFirstId.p, firstId, nRoots * sizeof(uint32_t), hipMemcpyHostToDevice, st));
	hipLaunchKernelGGL(k_emit, dim3(std::min<uint32_t>(nRoots, 65535u)), dim3(256), 0, st, dRoots.p, nRoots, dLocalCount.p, dRootId.p, dFirstId.p, dLocal.p, dNodes.p);
	BVH_TRY(hipGetLastError());
	BVH_TRY(hipStreamSynchronize(st));
	const auto t2 = std::chrono::steady_clock::now();
	BVH_TRY(hipMemcpyAsync(nodes_out, dNodes.p, (size_t)nodeCount * sizeof(crh_bvh_node), hipMemcpyDeviceToHost, st));
	BVH_TRY(hipMemcpyAsync(prim_indices_out, dPrims.p, (size_t)N * sizeof(int32_t), hipMemcpyDeviceToHost, st));
	BVH_TRY(hipStreamSynchronize(st));
	for (const UpperNode &n : upper) {            /* the split nodes of the upper tree (a few thousand) are written by the host */
		if (n.small >= 0) continue;
		crh_bvh_node o;
		memset(&o, 0, sizeof(o));
		memcpy(o.bounds, n.bounds, sizeof(o.bounds));
		o.first = upper[n.left].id;
		nodes_out[n.id] = o;
	}
	const auto t3 = std::chrono::steady_clock::now();
	*node_count_out = nodeCount;
	if (stats) {
		auto ms = [](auto a, auto b) { return std::chrono::duration<double, std::milli>(b - a).count(); };
		stats->upload_ms = ms(t0, t1); stats->build_ms = ms(t1, t2); stats->download_ms = ms(t2, t3);
		stats->levels = levels; stats->upper_nodes = (uint32_t)upper.size(); stats->subtrees = nRoots;
	}
	return CRH_OK;
}

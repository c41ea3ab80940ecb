/*
 * exact_math_check.cpp — TEST INFRASTRUCTURE: the host build of c-ray_amd/csrc/exact_math.h against the installed libm, bit for bit.
 *
 *   exact_math_check <stride> [pairs]      stride 1 = all 2^32 floats per unary function (and per powf exponent the path uses);
 *                                          pairs = random (x, y) pairs for powf / atan2f (default 2e7)
 * Prints one line per function: inputs tested, mismatches (NaN results compare equal to NaN), first mismatching input. Exit code 1 on any
 * mismatch. Built with -ffp-contract=off: only the explicit fma() calls of the header fuse, as in the device build.
 */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "../../c-ray_amd/csrc/exact_math.h"

using namespace crh::em;

static inline bool same(float a, float b) {
	if (a != a && b != b) return true;
	return fbits(a) == fbits(b);
}
static uint64_t rng_state = 0x9e3779b97f4a7c15ull;
static inline uint64_t splitmix(uint64_t &s) { uint64_t z = (s += 0x9e3779b97f4a7c15ull); z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ull; z = (z ^ (z >> 27)) * 0x94d049bb133111ebull; return z ^ (z >> 31); }

template <class F, class G>
static int unary(const char *name, F mine, G ref, uint32_t stride) {
	uint64_t bad = 0, n = 0;
	uint32_t first = 0;
	#pragma omp parallel for reduction(+ : bad, n) schedule(static)
	for (int64_t i = 0; i < (int64_t)1 << 32; i += stride) {
		const float x = ffrom((uint32_t)i);
		++n;
		if (!same(mine(x), ref(x))) {
			if (!bad) { first = (uint32_t)i; }
			++bad;
		}
	}
	printf("%-28s tested %llu mismatches %llu", name, (unsigned long long)n, (unsigned long long)bad);
	if (bad) printf("  e.g. x = %a (0x%08x): mine %a libm %a", ffrom(first), first, mine(ffrom(first)), ref(ffrom(first)));
	printf("\n");
	fflush(stdout);
	return bad ? 1 : 0;
}

int main(int argc, char **argv) {
	const uint32_t stride = argc > 1 ? (uint32_t)strtoul(argv[1], NULL, 10) : 4099u;
	const uint64_t pairs = argc > 2 ? strtoull(argv[2], NULL, 10) : 20000000ull;
	int rc = 0;
	rc |= unary("sinf", [](float x) { return sinf_(x); }, [](float x) { return sinf(x); }, stride);
	rc |= unary("cosf", [](float x) { return cosf_(x); }, [](float x) { return cosf(x); }, stride);
	rc |= unary("sincosf.sin", [](float x) { float s, c; sincosf_(x, s, c); return s; }, [](float x) { return sinf(x); }, stride);
	rc |= unary("sincosf.cos", [](float x) { float s, c; sincosf_(x, s, c); return c; }, [](float x) { return cosf(x); }, stride);
	rc |= unary("logf", [](float x) { return logf_(x); }, [](float x) { return logf(x); }, stride);
	rc |= unary("log10f", [](float x) { return log10f_(x); }, [](float x) { return log10f(x); }, stride);
	rc |= unary("tanf", [](float x) { return tanf_(x); }, [](float x) { return tanf(x); }, stride);
	rc |= unary("atanf", [](float x) { return atanf_(x); }, [](float x) { return atanf(x); }, stride);
	rc |= unary("acosf", [](float x) { return acosf_(x); }, [](float x) { return acosf(x); }, stride);
	rc |= unary("asinf", [](float x) { return asinf_(x); }, [](float x) { return asinf(x); }, stride);
	/* the exponents of the hot path: schlick (vector.h:271), sRGB (color.h:51-64), grayscale (color.h:39), blackbody (color.c:39,50) */
	static const float ys[] = {5.0f, 2.4f, 0.4166666667f, 2.0f, -0.1332047592f, -0.0755148492f, 0.5f, -1.0f, 3.0f};
	for (float y : ys) {
		char name[64];
		snprintf(name, sizeof(name), "powf(x, %g)", (double)y);
		rc |= unary(name, [y](float x) { return powf_(x, y); }, [y](float x) { return powf(x, y); }, stride);
	}
	rc |= unary("atan2f(x, 1.5)", [](float x) { return atan2f_(x, 1.5f); }, [](float x) { return atan2f(x, 1.5f); }, stride);
	rc |= unary("atan2f(x, 1.0)", [](float x) { return atan2f_(x, 1.0f); }, [](float x) { return atan2f(x, 1.0f); }, stride);
	rc |= unary("atan2f(x, -inf)", [](float x) { return atan2f_(x, -INFINITY); }, [](float x) { return atan2f(x, -INFINITY); }, stride);
	rc |= unary("atan2f(0, x)", [](float x) { return atan2f_(-0.0f, x); }, [](float x) { return atan2f(-0.0f, x); }, stride);
	rc |= unary("atan2f(inf, x)", [](float x) { return atan2f_(INFINITY, x); }, [](float x) { return atan2f(INFINITY, x); }, stride);
	rc |= unary("atan2f(x, 0)", [](float x) { return atan2f_(x, 0.0f); }, [](float x) { return atan2f(x, 0.0f); }, stride);
	rc |= unary("atan2f(1e-30, x)", [](float x) { return atan2f_(1e-30f, x); }, [](float x) { return atan2f(1e-30f, x); }, stride);
	rc |= unary("atan2f(-0.3, x)", [](float x) { return atan2f_(-0.3f, x); }, [](float x) { return atan2f(-0.3f, x); }, stride);
	/* random pairs: raw bit patterns (every class of special value) and unit-range values (what the path feeds them) */
	uint64_t badp = 0, bada = 0;
	uint64_t s = rng_state;
	for (uint64_t i = 0; i < pairs; ++i) {
		const uint64_t r = splitmix(s);
		float x = ffrom((uint32_t)r), y = ffrom((uint32_t)(r >> 32));
		if (i & 1) { x = (float)((r & 0xffffff) / 16777216.0 * 4.0 - 2.0); y = (float)(((r >> 24) & 0xffffff) / 16777216.0 * 8.0 - 4.0); }
		if (!same(powf_(x, y), powf(x, y))) { if (!badp) printf("powf mismatch: x %a y %a mine %a libm %a\n", x, y, powf_(x, y), powf(x, y)); ++badp; }
		if (!same(atan2f_(y, x), atan2f(y, x))) { if (!bada) printf("atan2f mismatch: y %a x %a mine %a libm %a\n", y, x, atan2f_(y, x), atan2f(y, x)); ++bada; }
	}
	printf("%-28s tested %llu mismatches %llu\n", "powf(random pairs)", (unsigned long long)pairs, (unsigned long long)badp);
	printf("%-28s tested %llu mismatches %llu\n", "atan2f(random pairs)", (unsigned long long)pairs, (unsigned long long)bada);
	return (rc || badp || bada) ? 1 : 0;
}

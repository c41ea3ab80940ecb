/*
 * wrap01_check.cpp — TEST INFRASTRUCTURE: pt_device.h's wrap01() (wrapMinMax(x, 0, 1) of vector.h:215-221 without the two fmodf for -1 <= x < 1)
 * against wrapMinMax itself on the host's fmodf, bit for bit.
 *   wrap01_check <stride>      stride 1 = all 2^32 floats (a minute on 8 cores)
 */
#include <stdio.h>
#include <stdlib.h>
#include "../../c-ray_amd/csrc/pt_device.h"

int main(int argc, char **argv) {
	const uint32_t stride = argc > 1 ? (uint32_t)strtoul(argv[1], 0, 10) : 4099u;
	uint64_t bad = 0, n = 0;
	#pragma omp parallel for reduction(+ : bad, n) schedule(static)
	for (int64_t i = 0; i < (int64_t)1 << 32; i += stride) {
		const float x = crh::asF32((uint32_t)i);
		const float a = crh::wrap01(x), b = crh::wrapMinMax(x, 0.0f, 1.0f);
		++n;
		if (!((a != a && b != b) || crh::asU32(a) == crh::asU32(b))) ++bad;
	}
	/* the boundaries of the two shortcut ranges, whatever the stride */
	const uint32_t edges[] = {0x00000000u, 0x80000000u, 0x00000001u, 0x80000001u, 0x3f7fffffu, 0x3f800000u, 0xbf7fffffu, 0xbf800000u, 0xbf800001u, 0x33800000u, 0xb3800000u, 0x33000000u, 0xb3000000u, 0x32ffffffu, 0xb2ffffffu};
	for (uint32_t e : edges) {
		const float x = crh::asF32(e), a = crh::wrap01(x), b = crh::wrapMinMax(x, 0.0f, 1.0f);
		++n;
		if (crh::asU32(a) != crh::asU32(b)) ++bad;
	}
	printf("wrap01 tested %llu mismatches %llu\n", (unsigned long long)n, (unsigned long long)bad);
	return bad ? 1 : 0;
}

/* podbuf_check.cpp — test infrastructure: scene_compile.h's PodBuf across its two allocation regimes (realloc below CRH_PODBUF_HUGE_FROM; 2 MB-aligned blocks that ask for
 * transparent huge pages from there on, which grow by copying) — contents survive every growth, the big blocks are aligned, resize does not initialise, a first
 * reservation is exact. The fixtures of the CPU tier are too small to reach the second regime (hdr.json's 134 MB of texels do, on the GPU tier). */
#include <cstdint>
#include <cstdio>
#include "../../c-ray_amd/csrc/scene_compile.h"
using crh::PodBuf;
using crh::f4;

static int fail(const char *what) { std::printf("podbuf_check: %s\n", what); return 1; }

int main() {
	{	/* growth by push_back from nothing, through the boundary: 1 M elements of 16 bytes = 16 MB */
		PodBuf<f4> b;
		for (uint32_t i = 0; i < (1u << 20); ++i) b.push_back(f4{(float)i, 1.0f, 2.0f, 3.0f});
		if (b.size() != (1u << 20)) return fail("size after push_back");
		for (uint32_t i = 0; i < (1u << 20); i += 4097) if (b[i].x != (float)i || b[i].w != 3.0f) return fail("contents lost while growing");
		if (((uintptr_t)b.data() & (CRH_PODBUF_HUGE_PAGE - 1)) != 0) return fail("a 16 MB block is not on a 2 MB boundary");
	}
	{	/* an exact first reservation, then resize inside it keeps the block and what is in it */
		PodBuf<uint32_t> b;
		b.reserve(5000000);
		if (b.cap != 5000000) return fail("first reservation is not exact");
		const uint32_t *p0 = b.data();
		b.resize(1000); for (uint32_t i = 0; i < 1000; ++i) b[i] = i * 7u;
		b.resize(4000000);
		if (b.data() != p0) return fail("resize inside the reservation moved the block");
		for (uint32_t i = 0; i < 1000; ++i) if (b[i] != i * 7u) return fail("resize inside the reservation lost contents");
		b.resize(9000000);                    /* beyond it: a bigger block, the old contents copied */
		for (uint32_t i = 0; i < 1000; ++i) if (b[i] != i * 7u) return fail("growth of a big block lost contents");
		if (((uintptr_t)b.data() & (CRH_PODBUF_HUGE_PAGE - 1)) != 0) return fail("a grown big block is not on a 2 MB boundary");
	}
	{	/* small blocks stay with realloc, assign / resize(n, v) fill */
		PodBuf<uint8_t> b;
		b.assign(1000, (uint8_t)9);
		b.resize(3000, (uint8_t)4);
		if (b[999] != 9 || b[1000] != 4 || b[2999] != 4) return fail("fill values");
	}
	std::printf("podbuf_check ok\n");
	return 0;
}

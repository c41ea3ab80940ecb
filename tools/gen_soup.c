/*
 * gen_soup.c — writes the synthetic triangle soup of BASELINE.json config 5 as a Wavefront OBJ that the
 * UNMODIFIED reference loader consumes (SURVEY.md §8(d) "Synthetic soup"). BENCH/TEST INFRASTRUCTURE.
 *
 *   gen_soup <N> <out.obj>
 *
 * PCG32 (seed 42, stream 0): per triangle a centre c ~ U([-1,1]^3) and two edge vectors ~ U([-s,s]^3),
 * s = 0.02; vertices c, c+e1, c+e2; no normals (exercises the flat-normal branch, poly.c:45-47), one grey
 * diffuse material from soup.mtl.
 */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

static uint64_t state, inc;
static uint32_t pcg32(void) {
	uint64_t old = state;
	state = old * 6364136223846793005ULL + inc;
	uint32_t xs = (uint32_t)(((old >> 18u) ^ old) >> 27u), rot = (uint32_t)(old >> 59u);
	return (xs >> rot) | (xs << ((-rot) & 31));
}
static void seed(uint64_t s, uint64_t seq) { state = 0; inc = (seq << 1u) | 1u; pcg32(); state += s; pcg32(); }
static float uni(float lo, float hi) { return lo + (hi - lo) * (float)(pcg32() >> 8) * (1.0f / 16777216.0f); }

int main(int argc, char **argv) {
	if (argc < 3) { fprintf(stderr, "usage: gen_soup <N> <out.obj>\n"); return 2; }
	long n = atol(argv[1]);
	FILE *f = fopen(argv[2], "w");
	if (!f) { perror(argv[2]); return 1; }
	static char buf[1 << 22];
	setvbuf(f, buf, _IOFBF, sizeof(buf));
	seed(42, 0);
	const float s = 0.02f;
	fprintf(f, "# synthetic soup, %ld triangles (tools/gen_soup.c)\nmtllib soup.mtl\no soup\n", n);
	for (long i = 0; i < n; ++i) {
		float c[3], e1[3], e2[3];
		for (int k = 0; k < 3; ++k) c[k] = uni(-1.0f, 1.0f);
		for (int k = 0; k < 3; ++k) e1[k] = uni(-s, s);
		for (int k = 0; k < 3; ++k) e2[k] = uni(-s, s);
		fprintf(f, "v %.7f %.7f %.7f\nv %.7f %.7f %.7f\nv %.7f %.7f %.7f\n", c[0], c[1], c[2],
				c[0] + e1[0], c[1] + e1[1], c[2] + e1[2], c[0] + e2[0], c[1] + e2[1], c[2] + e2[2]);
	}
	fprintf(f, "usemtl grey\n");
	/* "v//": the reference tokenises every face vertex on / and reads three fields (wavefront.c:97-103), so a
	 * bare "f 1 2 3" would dereference NULL there; empty fields parse as 0 = "unused" */
	for (long i = 0; i < n; ++i) fprintf(f, "f %ld// %ld// %ld//\n", 3 * i + 1, 3 * i + 2, 3 * i + 3);
	fclose(f);
	return 0;
}

/*
 * gen_soup.c — writes the synthetic triangle soup of BASELINE.json config 5 as a Wavefront OBJ that the
 * UNMODIFIED reference loader consumes (SURVEY.md §8(d) "Synthetic soup"). BENCH/TEST INFRASTRUCTURE.
 *
 *   gen_soup <N> <out.obj>
 *   gen_soup --bin <N> <out.f32>      (see writeBinary below)
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

/* gen_soup --bin <N> <out.f32>: the 9 N vertex coordinates AS THE REFERENCE'S LOADER SEES THEM — printed with %.7f like the OBJ below and read back
 * with atof (wavefront.c:68), narrowed to float — as raw floats, without the 1 GB text file in between (tools/make_soup_blob.py builds the 10 M-triangle
 * scene blob on the GPU box from them; at 1 M triangles the blob equals the one the reference's loader made, byte for byte). */
static int writeBinary(long n, const char *path) {
	float *v = malloc(sizeof(float) * 9 * (size_t)n);
	if (!v) return 1;
	seed(42, 0);
	const float s = 0.02f;
	for (long i = 0; i < n; ++i) {
		float c[3], e1[3], e2[3];
		for (int k = 0; k < 3; ++k) c[k] = uni(-1.0f, 1.0f);
		for (int k = 0; k < 3; ++k) e1[k] = uni(-s, s);
		for (int k = 0; k < 3; ++k) e2[k] = uni(-s, s);
		for (int k = 0; k < 3; ++k) { v[9 * i + k] = c[k]; v[9 * i + 3 + k] = c[k] + e1[k]; v[9 * i + 6 + k] = c[k] + e2[k]; }
	}
#pragma omp parallel for schedule(static)
	for (long i = 0; i < 9 * n; ++i) {
		char txt[64];
		snprintf(txt, sizeof(txt), "%.7f", v[i]);
		v[i] = (float)atof(txt);
	}
	FILE *f = fopen(path, "wb");
	if (!f) { perror(path); return 1; }
	const size_t wrote = fwrite(v, sizeof(float), 9 * (size_t)n, f);
	fclose(f);
	free(v);
	return wrote == 9 * (size_t)n ? 0 : 1;
}

int main(int argc, char **argv) {
	if (argc >= 4 && argv[1][0] == '-' && argv[1][1] == '-' && argv[1][2] == 'b') return writeBinary(atol(argv[2]), argv[3]);
	if (argc < 3) { fprintf(stderr, "usage: gen_soup <N> <out.obj> | gen_soup --bin <N> <out.f32>\n"); return 2; }
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

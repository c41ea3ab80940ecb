/*
 * exact_math.h — the libm functions of the hot path, restated so that the device returns the SAME BITS as the reference's host.
 *
 * Why: every other operation of the path is IEEE-exact on both sides (add / mul / div / sqrt / the one explicit fma), so the only
 * source of GPU-vs-reference differences were the last-ulp differences between ocml and the host's libm in sinf, cosf, powf, logf,
 * atan2f, acosf, asinf (and tanf / log10f in the math node). One ulp occasionally flips a hit / miss or a Russian-roulette decision
 * and that path decorrelates; on a chaotic scene (statues.json) that is a few percent of the pixels. With these the device frame
 * equals the reference's frame bit for bit.
 *
 * The reference is C and calls its platform's libm: here glibc 2.35 on x86-64 with FMA (the image of both this container and the GPU
 * box; the fixtures under tests/golden were rendered with it). That library is a third-party dependency that is not part of
 * /root/reference; its algorithms are published (ARM optimized-routines for sinf / cosf / powf / logf, which glibc adopted in 2.28;
 * the float port of Sun's fdlibm for atanf / atan2f / acosf / asinf / tanf / log10f). They are restated here with every table and
 * coefficient read from the installed libm.so.6, and with the fused multiply-adds exactly where its x86-64 "fma" ifunc variants have
 * them (sinf / cosf / powf / logf evaluate in double with contracted polynomials; the fdlibm ports are plain float code built without
 * contraction). tests/test_exact_math.py checks the host build of this header against the installed libm: all 2^32 inputs of the
 * unary functions, all 2^32 bases for each exponent the path uses and random pairs for powf / atan2f.
 *
 * Everything is branch-light scalar code over uint32 / float / double: fp64 runs at half the fp32 rate on CDNA4, so the double
 * polynomials cost about what ocml's float versions do.
 */
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#define CRH_EM __device__ __forceinline__
#define CRH_EM_TAB static __device__ const
#else
#define CRH_EM static inline __attribute__((always_inline))
#define CRH_EM_TAB static const
#endif

namespace crh {
namespace em {

CRH_EM uint32_t fbits(float f) { union { float f; uint32_t u; } c; c.f = f; return c.u; }
CRH_EM float ffrom(uint32_t u) { union { float f; uint32_t u; } c; c.u = u; return c.f; }
CRH_EM uint64_t dbits(double f) { union { double f; uint64_t u; } c; c.f = f; return c.u; }
CRH_EM double dfrom(uint64_t u) { union { double f; uint64_t u; } c; c.u = u; return c.f; }
CRH_EM double fmad(double a, double b, double c) { return __builtin_fma(a, b, c); }
CRH_EM float fabsf_(float x) { return ffrom(fbits(x) & 0x7fffffffu); }
CRH_EM float sqrtf_(float x) { return __builtin_sqrtf(x); }          /* IEEE on both sides */
CRH_EM float invalidf(float x) { return (x - x) / (x - x); }         /* __math_invalidf */

/* ---- sinf / cosf: sysdeps/ieee754/flt-32/s_sinf.c, s_cosf.c, sincosf.h, sincosf_data.c ------------------------------------- */
/* __sincosf_table[0]: hpi_inv, hpi, the cosine polynomial c0..c4 and the sine polynomial s1..s3. The second table of the original is the
 * same with the cosine coefficients negated, and sign[4] = {1, -1, -1, 1}: negating every coefficient of an fma chain negates its result
 * exactly (round-to-nearest is symmetric), and a multiplication by -1 is a sign flip — so both are done on the sign bit here, and every
 * coefficient is a literal (a scalar register pair on the device, never a table load through a per-lane pointer). */
#define CRH_SC_HPI_INV 0x1.45f306dc9c883p+23
#define CRH_SC_HPI     0x1.921fb54442d18p+0
#define CRH_SC_C0 0x1p0
#define CRH_SC_C1 -0x1.ffffffd0c621cp-2
#define CRH_SC_C2 0x1.55553e1068f19p-5
#define CRH_SC_C3 -0x1.6c087e89a359dp-10
#define CRH_SC_C4 0x1.99343027bf8c3p-16
#define CRH_SC_S1 -0x1.555545995a603p-3
#define CRH_SC_S2 0x1.1107605230bc4p-7
#define CRH_SC_S3 -0x1.994eb3774cf24p-13
/* 4/pi in 32-bit pieces, each entry shifted by 8 bits (__inv_pio4) */
CRH_EM_TAB uint32_t kInvPio4[24] = {
	0xa2, 0xa2f9, 0xa2f983, 0xa2f9836e, 0xf9836e4e, 0x836e4e44, 0x6e4e4415, 0x4e441529, 0x441529fc, 0x1529fc27, 0x29fc2757, 0xfc2757d1,
	0x2757d1f5, 0x57d1f534, 0xd1f534dd, 0xf534ddc0, 0x34ddc0db, 0xddc0db62, 0xc0db6295, 0xdb629599, 0x6295993c, 0x95993c43, 0x993c4390, 0x3c439041,
};
CRH_EM double negIf(double x, uint32_t neg) { return dfrom(dbits(x) ^ ((uint64_t)(neg & 1u) << 63)); }

/* sinf_poly(): the sine polynomial (even n) and the cosine polynomial (odd n) of x2 = x * x; contraction as in __sinf_fma */
CRH_EM double sinPoly(double x, double x2) {
	const double x3 = x * x2;
	const double s1 = fmad(CRH_SC_S3, x2, CRH_SC_S2);
	const double x7 = x2 * x3;
	const double s = fmad(x3, CRH_SC_S1, x);
	return fmad(s1, x7, s);
}
CRH_EM double cosPoly(double x2) {
	const double x4 = x2 * x2;
	const double c = fmad(CRH_SC_C1, x2, CRH_SC_C0);
	const double c2 = fmad(CRH_SC_C4, x2, CRH_SC_C3);
	const double x6 = x2 * x4;
	const double c1 = fmad(x4, CRH_SC_C2, c);
	return fmad(c2, x6, c1);
}
/* reduce_fast(): |x| < 120; x - n * pi/2 with n = round(x * 2/pi) */
CRH_EM double reduceFast(double x, int &n) {
	const double r = x * CRH_SC_HPI_INV;
	n = ((int32_t)r + 0x800000) >> 24;
	return fmad(-(double)n, CRH_SC_HPI, x);
}
/* reduce_large(): |x| >= 120, 96 bits of 4/pi */
CRH_EM double reduceLarge(uint32_t xi, int &n) {
	const uint32_t *arr = &kInvPio4[(xi >> 26) & 15];
	const int shift = (xi >> 23) & 7;
	xi = (xi & 0xffffff) | 0x800000;
	xi <<= shift;
	uint64_t res0 = (uint32_t)(xi * arr[0]);
	const uint64_t res1 = (uint64_t)xi * arr[4];
	const uint64_t res2 = (uint64_t)xi * arr[8];
	res0 = (res2 >> 32) | (res0 << 32);
	res0 += res1;
	const uint64_t nn = (res0 + (1ULL << 61)) >> 62;
	res0 -= nn << 62;
	n = (int)nn;
	return (double)(int64_t)res0 * 0x1.921fb54442d18p-62;
}
/* Both functions of one angle (vector.h:190-198, 243-249 take the sine and the cosine of the same random angle): ONE argument reduction.
 * q = quadrant (n, or n + sign for the large path); sine: n even -> sign[q & 3] * sinPoly, n odd -> (q & 2 ? -1 : 1) * cosPoly;
 * cosine: n even -> (q & 2 ? -1 : 1) * cosPoly, n odd -> sign[q & 3] * sinPoly (s_sinf.c:63-92, s_cosf.c:63-92). */
CRH_EM void sincosf_(float y, float &sn, float &cs) {
	const uint32_t iy = fbits(y), top = (iy >> 20) & 0x7ffu;
	double x = (double)y;
	int n = 0;
	uint32_t q = 0;
	if (top <= 0x3f3u) {                       /* |y| < pi/4 */
		if (top <= 0x397u) { sn = y; cs = 1.0f; return; }   /* |y| < 2^-12 */
	} else if (top <= 0x42eu) {                /* |y| < 120 */
		x = reduceFast(x, n);
		q = (uint32_t)n;
	} else if (top <= 0x7f7u) {
		x = reduceLarge(iy, n);
		q = (uint32_t)n + (iy >> 31);
	} else {
		sn = cs = invalidf(y);
		return;
	}
	const double x2 = x * x;
	const double sp = sinPoly(negIf(x, q ^ (q >> 1)), x2);      /* x * sign[q & 3] */
	const double cp = negIf(cosPoly(x2), q >> 1);               /* the negated table for q & 2 */
	sn = (float)((n & 1) ? cp : sp);
	cs = (float)((n & 1) ? sp : cp);
}
CRH_EM float sinf_(float y) { float s, c; sincosf_(y, s, c); return s; }
CRH_EM float cosf_(float y) { float s, c; sincosf_(y, s, c); return c; }

/* ---- logf: sysdeps/ieee754/flt-32/e_logf.c, e_logf_data.c (LOGF_TABLE_BITS 4, LOGF_POLY_ORDER 4) --------------------------- */
CRH_EM_TAB double kLogfTab[16][2] = {       /* {invc, logc} */
	{0x1.661ec79f8f3bep+0, -0x1.57bf7808caadep-2}, {0x1.571ed4aaf883dp+0, -0x1.2bef0a7c06ddbp-2}, {0x1.49539f0f010bp+0, -0x1.01eae7f513a67p-2},
	{0x1.3c995b0b80385p+0, -0x1.b31d8a68224e9p-3}, {0x1.30d190c8864a5p+0, -0x1.6574f0ac07758p-3}, {0x1.25e227b0b8eap+0, -0x1.1aa2bc79c81p-3},
	{0x1.1bb4a4a1a343fp+0, -0x1.a4e76ce8c0e5ep-4}, {0x1.12358f08ae5bap+0, -0x1.1973c5a611cccp-4}, {0x1.0953f419900a7p+0, -0x1.252f438e10c1ep-5},
	{0x1p+0, 0x0p+0}, {0x1.e608cfd9a47acp-1, 0x1.aa5aa5df25984p-5}, {0x1.ca4b31f026aap-1, 0x1.c5e53aa362eb4p-4},
	{0x1.b2036576afce6p-1, 0x1.526e57720db08p-3}, {0x1.9c2d163a1aa2dp-1, 0x1.bc2860d22477p-3}, {0x1.886e6037841edp-1, 0x1.1058bc8a07ee1p-2},
	{0x1.767dcf5534862p-1, 0x1.4043057b6ee09p-2},
};
CRH_EM float logf_(float x) {
	uint32_t ix = fbits(x);
	if (ix == 0x3f800000u) return 0.0f;
	if (ix - 0x00800000u >= 0x7f800000u - 0x00800000u) {
		if (ix * 2u == 0u) return -1.0f / 0.0f;                                  /* __math_divzerof (1) */
		if (ix == 0x7f800000u) return x;                                          /* log(inf) == inf */
		if ((ix & 0x80000000u) || ix * 2u >= 0xff000000u) return invalidf(x);
		ix = fbits(x * 0x1p23f);                                                  /* subnormal: normalise */
		ix -= 23u << 23;
	}
	const uint32_t tmp = ix - 0x3f330000u;
	const int i = (int)((tmp >> 19) & 15u);
	const int k = (int32_t)tmp >> 23;
	const uint32_t iz = ix - (tmp & 0xff800000u);
	const double invc = kLogfTab[i][0], logc = kLogfTab[i][1];
	const double z = (double)ffrom(iz);
	const double r = fmad(z, invc, -1.0);
	const double y0 = fmad((double)k, 0x1.62e42fefa39efp-1, logc);
	const double r2 = r * r;
	double y = fmad(0x1.5575b0be00b6ap-2, r, -0x1.ffffef20a4123p-2);
	y = fmad(r2, -0x1.00ea348b88334p-2, y);
	y = fmad(r2, y, r + y0);
	return (float)y;
}

/* ---- powf: sysdeps/ieee754/flt-32/e_powf.c, e_powf_log2_data.c, e_exp2f_data.c (POWF_LOG2_TABLE_BITS 4, EXP2F_TABLE_BITS 5) -- */
CRH_EM_TAB double kPowLog2Tab[16][2] = {    /* {invc, logc} */
	{0x1.661ec79f8f3bep+0, -0x1.efec65b963019p-2}, {0x1.571ed4aaf883dp+0, -0x1.b0b6832d4fca4p-2}, {0x1.49539f0f010bp+0, -0x1.7418b0a1fb77bp-2},
	{0x1.3c995b0b80385p+0, -0x1.39de91a6dcf7bp-2}, {0x1.30d190c8864a5p+0, -0x1.01d9bf3f2b631p-2}, {0x1.25e227b0b8eap+0, -0x1.97c1d1b3b7afp-3},
	{0x1.1bb4a4a1a343fp+0, -0x1.2f9e393af3c9fp-3}, {0x1.12358f08ae5bap+0, -0x1.960cbbf788d5cp-4}, {0x1.0953f419900a7p+0, -0x1.a6f9db6475fcep-5},
	{0x1p+0, 0x0p+0}, {0x1.e608cfd9a47acp-1, 0x1.338ca9f24f53dp-4}, {0x1.ca4b31f026aap-1, 0x1.476a9543891bap-3},
	{0x1.b2036576afce6p-1, 0x1.e840b4ac4e4d2p-3}, {0x1.9c2d163a1aa2dp-1, 0x1.40645f0c6651cp-2}, {0x1.886e6037841edp-1, 0x1.88e9c2c1b9ff8p-2},
	{0x1.767dcf5534862p-1, 0x1.ce0a44eb17bccp-2},
};
CRH_EM_TAB uint64_t kExp2fTab[32] = {
	0x3ff0000000000000, 0x3fefd9b0d3158574, 0x3fefb5586cf9890f, 0x3fef9301d0125b51, 0x3fef72b83c7d517b, 0x3fef54873168b9aa, 0x3fef387a6e756238, 0x3fef1e9df51fdee1,
	0x3fef06fe0a31b715, 0x3feef1a7373aa9cb, 0x3feedea64c123422, 0x3feece086061892d, 0x3feebfdad5362a27, 0x3feeb42b569d4f82, 0x3feeab07dd485429, 0x3feea47eb03a5585,
	0x3feea09e667f3bcd, 0x3fee9f75e8ec5f74, 0x3feea11473eb0187, 0x3feea589994cce13, 0x3feeace5422aa0db, 0x3feeb737b0cdc5e5, 0x3feec49182a3f090, 0x3feed503b23e255d,
	0x3feee89f995ad3ad, 0x3feeff76f2fb5e47, 0x3fef199bdd85529c, 0x3fef3720dcef9069, 0x3fef5818dcfba487, 0x3fef7c97337b9b5f, 0x3fefa4afa2a490da, 0x3fefd0765b6e4540,
};
/* On the device the two tables of powf can live in LDS (512 B per workgroup, copied in by the kernel: CRH_EM_POW_TABLES_INIT): both lookups
 * sit on powf's critical path (the exp2 index depends on the log2 result), and an LDS read answers several times faster than a
 * per-lane load through the vector cache. Kernels that define CRH_EM_POW_TABLES_IN_LDS before including this header get them. */
#if defined(__HIPCC__) && defined(CRH_EM_POW_TABLES_IN_LDS)
__shared__ double s_powLog2Tab[32];
__shared__ uint64_t s_powExp2Tab[32];
#define CRH_EM_LOG2TAB(i, j) s_powLog2Tab[2 * (i) + (j)]
#define CRH_EM_EXP2TAB(i) s_powExp2Tab[i]
/* every thread of the block calls this once, before any powf_ (ends with a barrier) */
#define CRH_EM_POW_TABLES_INIT() do { \
		if (threadIdx.x < 32u) { crh::em::s_powLog2Tab[threadIdx.x] = crh::em::kPowLog2Tab[threadIdx.x >> 1][threadIdx.x & 1u]; crh::em::s_powExp2Tab[threadIdx.x] = crh::em::kExp2fTab[threadIdx.x]; } \
		__syncthreads(); \
	} while (0)
#else
#define CRH_EM_LOG2TAB(i, j) kPowLog2Tab[i][j]
#define CRH_EM_EXP2TAB(i) kExp2fTab[i]
#define CRH_EM_POW_TABLES_INIT() do { } while (0)
#endif
CRH_EM int powCheckInt(uint32_t iy) {       /* 0: not an integer, 1: odd, 2: even */
	const int e = (int)((iy >> 23) & 0xffu);
	if (e < 0x7f) return 0;
	if (e > 0x7f + 23) return 2;
	if (iy & ((1u << (0x7f + 23 - e)) - 1u)) return 0;
	if (iy & (1u << (0x7f + 23 - e))) return 1;
	return 2;
}
CRH_EM bool powZeroInfNan(uint32_t ix) { return 2u * ix - 1u >= 2u * 0x7f800000u - 1u; }
CRH_EM bool isSignalingF(float x) { return ((fbits(x) ^ 0x00400000u) & 0x7fffffffu) > 0x7fc00000u; }
CRH_EM float powXflow(uint32_t sign, float y) { return (sign ? -y : y) * y; }
CRH_EM float powf_(float x, float y) {
	uint32_t signBias = 0;
	uint32_t ix = fbits(x);
	const uint32_t iy = fbits(y);
	if (ix - 0x00800000u >= 0x7f800000u - 0x00800000u || powZeroInfNan(iy)) {
		/* either (x < 0x1p-126 or inf or nan) or (y is 0 or inf or nan) */
		if (powZeroInfNan(iy)) {
			if (2u * iy == 0u) return isSignalingF(x) ? x + y : 1.0f;
			if (ix == 0x3f800000u) return isSignalingF(y) ? x + y : 1.0f;
			if (2u * ix > 2u * 0x7f800000u || 2u * iy > 2u * 0x7f800000u) return x + y;
			if (2u * ix == 2u * 0x3f800000u) return 1.0f;
			if ((2u * ix < 2u * 0x3f800000u) == !(iy & 0x80000000u)) return 0.0f;   /* |x| < 1 && y == inf or |x| > 1 && y == -inf */
			return y * y;
		}
		if (powZeroInfNan(ix)) {
			float x2 = x * x;
			if ((ix & 0x80000000u) && powCheckInt(iy) == 1) { x2 = -x2; signBias = 1; }
			if (2u * ix == 0u && (iy & 0x80000000u)) return (signBias ? -1.0f : 1.0f) / 0.0f;   /* __math_divzerof */
			return (iy & 0x80000000u) ? 1.0f / x2 : x2;
		}
		/* x and y are non-zero finite */
		if (ix & 0x80000000u) {
			const int yint = powCheckInt(iy);
			if (yint == 0) return invalidf(x);
			if (yint == 1) signBias = 1u << 16;         /* SIGN_BIAS = 1 << (EXP2F_TABLE_BITS + 11) */
			ix &= 0x7fffffffu;
		}
		if (ix < 0x00800000u) {                          /* subnormal x: normalise */
			ix = fbits(x * 0x1p23f);
			ix &= 0x7fffffffu;
			ix -= 23u << 23;
		}
	}
	/* log2_inline */
	const uint32_t tmp = ix - 0x3f330000u;
	const int i = (int)((tmp >> 19) & 15u);
	const uint32_t top = tmp & 0xff800000u;
	const uint32_t iz = ix - top;
	const int k = (int32_t)top >> 23;
	const double invc = CRH_EM_LOG2TAB(i, 0), logc = CRH_EM_LOG2TAB(i, 1);
	const double z = (double)ffrom(iz);
	const double r = fmad(z, invc, -1.0);
	const double y0 = (double)k + logc;
	const double a = fmad(0x1.27616c9496e0bp-2, r, -0x1.71969a075c67ap-2);
	const double p = fmad(0x1.ec70a6ca7baddp-2, r, -0x1.7154748bef6c8p-1);
	const double r2 = r * r;
	double q = fmad(r, 0x1.71547652ab82bp+0, y0);
	const double r4 = r2 * r2;
	q = fmad(r2, p, q);
	const double logx = fmad(a, r4, q);
	const double ylogx = (double)y * logx;               /* cannot overflow: y is single precision */
	if (((dbits(ylogx) >> 47) & 0xffffu) >= 0x80bfu) {    /* |y * log2(x)| >= 126 */
		if (ylogx > 0x1.fffffffd1d571p+6) return powXflow(signBias, 0x1p97f);      /* overflow */
		if (ylogx <= -150.0) return powXflow(signBias, 0x1p-95f);                    /* underflow */
		if (ylogx < -149.0) return powXflow(signBias, 0x1.4p-75f);                   /* may underflow */
	}
	/* exp2_inline */
	double kd = ylogx + 0x1.8p+47;                        /* shift_scaled: rounds to a multiple of 1/32 */
	const uint64_t ki = dbits(kd);
	kd -= 0x1.8p+47;
	const double rr = ylogx - kd;
	uint64_t t = CRH_EM_EXP2TAB(ki & 31u);
	t += (ki + signBias) << 47;
	const double s = dfrom(t);
	const double zz = fmad(0x1.c6af84b912394p-5, rr, 0x1.ebfce50fac4f3p-3);
	const double rr2 = rr * rr;
	double yy = fmad(rr, 0x1.62e42ff0c52d6p-1, 1.0);
	yy = fmad(zz, rr2, yy);
	return (float)(yy * s);
}

/* ---- atanf / atan2f: sysdeps/ieee754/flt-32/s_atanf.c, e_atan2f.c (plain float code, no contraction) ------------------------- */
/* Written without divergent branches (one division, one polynomial for every lane; the ranges pick operands and constants): the same
 * IEEE operations on the same operands as the branchy original, hence the same bits. */
CRH_EM float atanf_(float x) {
	const float aT0 = 3.3333334327e-01f, aT1 = -2.0000000298e-01f, aT2 = 1.4285714924e-01f, aT3 = -1.1111110449e-01f, aT4 = 9.0908870101e-02f,
				aT5 = -7.6918758452e-02f, aT6 = 6.6610731184e-02f, aT7 = -5.8335702866e-02f, aT8 = 4.9768779427e-02f, aT9 = -3.6531571299e-02f,
				aT10 = 1.6285819933e-02f;
	const int32_t hx = (int32_t)fbits(x), ix = hx & 0x7fffffff;
	const float ax = fabsf_(x);
	/* argument reduction: id -1 (|x| < 0.4375: x itself), 0 (< 0.6875), 1 (< 1.1875), 2 (< 2.4375), 3 */
	const bool r0 = ix < 0x3ee00000, r1 = ix < 0x3f300000, r2 = ix < 0x3f980000, r3 = ix < 0x401c0000;
	const float num = r0 ? x : r1 ? (2.0f * ax - 1.0f) : r2 ? (ax - 1.0f) : r3 ? (ax - 1.5f) : -1.0f;
	const float den = r0 ? 1.0f : r1 ? (2.0f + ax) : r2 ? (ax + 1.0f) : r3 ? (1.0f + 1.5f * ax) : ax;
	const float hi = r1 ? 4.6364760399e-01f : r2 ? 7.8539812565e-01f : r3 ? 9.8279368877e-01f : 1.5707962513e+00f;
	const float lo = r1 ? 5.0121582440e-09f : r2 ? 3.7748947079e-08f : r3 ? 3.4473217170e-08f : 7.5497894159e-08f;
	const float xr = num / den;                          /* x / 1 == x for the small range */
	const float z = xr * xr;
	const float w = z * z;
	const float s1 = z * (aT0 + w * (aT2 + w * (aT4 + w * (aT6 + w * (aT8 + w * aT10)))));
	const float s2 = w * (aT1 + w * (aT3 + w * (aT5 + w * (aT7 + w * aT9))));
	const float small = xr - xr * (s1 + s2);
	const float zz = hi - ((xr * (s1 + s2) - lo) - xr);
	float res = r0 ? small : ((hx < 0) ? -zz : zz);
	if (ix < 0x31000000) res = x;                        /* |x| < 2^-29 */
	if (ix >= 0x4c000000) {                              /* |x| >= 2^25 */
		const float big = 1.5707962513e+00f + 7.5497894159e-08f;
		res = (ix > 0x7f800000) ? x + x : (hx > 0) ? big : -1.5707962513e+00f - 7.5497894159e-08f;
	}
	return res;
}
CRH_EM float atan2f_(float y, float x) {
	const float tiny = 1.0e-30f, pi_o_4 = 7.8539818525e-01f, pi_o_2 = 1.5707963705e+00f, pi = 3.1415927410e+00f, pi_lo = -8.7422776573e-08f;
	const int32_t hx = (int32_t)fbits(x), ix = hx & 0x7fffffff;
	const int32_t hy = (int32_t)fbits(y), iy = hy & 0x7fffffff;
	const int m = ((hy >> 31) & 1) | ((hx >> 30) & 2);   /* 2 * sign(x) + sign(y) */
	/* the ordinary case first, for every lane: ONE atanf (of y when x == 1.0, of |y / x| otherwise), then the quadrant */
	const bool xIsOne = hx == 0x3f800000;
	const int k = (iy - ix) >> 23;
	const float at = atanf_(xIsOne ? y : fabsf_(y / x));
	float z = at;
	if (k > 60) z = pi_o_2 + 0.5f * pi_lo;               /* |y / x| > 2^60 */
	else if (hx < 0 && k < -60) z = 0.0f;                /* |y| / x < -2^60 */
	float res = (m == 0) ? z : (m == 1) ? ffrom(fbits(z) ^ 0x80000000u) : (m == 2) ? pi - (z - pi_lo) : (z - pi_lo) - pi;
	/* the special cases, in reverse order of the original's early returns (the earliest one wins) */
	if (iy == 0x7f800000) res = (hy < 0) ? -pi_o_2 - tiny : pi_o_2 + tiny;
	if (ix == 0x7f800000) {
		if (iy == 0x7f800000) res = (m == 0) ? pi_o_4 + tiny : (m == 1) ? -pi_o_4 - tiny : (m == 2) ? 3.0f * pi_o_4 + tiny : -3.0f * pi_o_4 - tiny;
		else res = (m == 0) ? 0.0f : (m == 1) ? -0.0f : (m == 2) ? pi + tiny : -pi - tiny;
	}
	if (ix == 0) res = (hy < 0) ? -pi_o_2 - tiny : pi_o_2 + tiny;
	if (iy == 0) res = (m <= 1) ? y : (m == 2) ? pi + tiny : -pi - tiny;
	if (xIsOne) res = at;                                /* atan2f(y, 1.0) = atanf(y) */
	if (ix > 0x7f800000 || iy > 0x7f800000) res = x + y;
	return res;
}

/* ---- acosf / asinf: sysdeps/ieee754/flt-32/e_acosf.c, e_asinf.c --------------------------------------------------------------- */
CRH_EM float acosf_(float x) {
	const float pi = 3.1415925026e+00f, pio2_hi = 1.5707962513e+00f, pio2_lo = 7.5497894159e-08f,
				pS0 = 1.6666667163e-01f, pS1 = -3.2556581497e-01f, pS2 = 2.0121252537e-01f, pS3 = -4.0055535734e-02f, pS4 = 7.9153501429e-04f,
				pS5 = 3.4793309169e-05f, qS1 = -2.4033949375e+00f, qS2 = 2.0209457874e+00f, qS3 = -6.8828397989e-01f, qS4 = 7.7038154006e-02f;
	const int32_t hx = (int32_t)fbits(x), ix = hx & 0x7fffffff;
	/* three ranges, one rational p(z) / q(z), one square root, evaluated by every lane (no divergent branches; same operations and
	 * operands as the original's three branches): |x| < 0.5: z = x^2; x <= -0.5: z = (1 + x) / 2; x >= 0.5: z = (1 - x) / 2 */
	const bool mid = ix < 0x3f000000, neg = hx < 0;
	const float z = mid ? x * x : neg ? (1.0f + x) * 0.5f : (1.0f - x) * 0.5f;
	const float p = z * (pS0 + z * (pS1 + z * (pS2 + z * (pS3 + z * (pS4 + z * pS5)))));
	const float q = 1.0f + z * (qS1 + z * (qS2 + z * (qS3 + z * qS4)));
	const float r = p / q;
	const float s = sqrtf_(z);
	const float rMid = pio2_hi - (x - (pio2_lo - x * r));
	const float wNeg = r * s - pio2_lo;
	const float rNeg = pi - 2.0f * (s + wNeg);
	const float df = ffrom(fbits(s) & 0xfffff000u);
	const float c = (z - df * df) / (s + df);
	const float wPos = r * s + c;
	const float rPos = 2.0f * (df + wPos);
	float res = mid ? rMid : neg ? rNeg : rPos;
	if (mid && ix <= 0x23000000) res = pio2_hi + pio2_lo;               /* |x| <= 2^-57 */
	if (ix == 0x3f800000) res = (hx > 0) ? 0.0f : pi + 2.0f * pio2_lo;  /* |x| == 1 */
	if (ix > 0x3f800000) res = invalidf(x);
	return res;
}
CRH_EM float asinf_(float x) {
	const float pio2_hi = 1.57079637050628662109375f, pio2_lo = -4.37113900018624283e-8f, pio4_hi = 0.785398185253143310546875f,
				p0 = 1.666675248e-1f, p1 = 7.495297643e-2f, p2 = 4.547037598e-2f, p3 = 2.417951451e-2f, p4 = 4.216630880e-2f;
	const int32_t hx = (int32_t)fbits(x), ix = hx & 0x7fffffff;
	if (ix == 0x3f800000) return x * pio2_hi + x * pio2_lo;
	if (ix > 0x3f800000) return invalidf(x);
	float t, w, p, q, c, r, s;
	if (ix < 0x3f000000) {                               /* |x| < 0.5 */
		if (ix < 0x32000000) return x;                   /* |x| < 2^-27 */
		t = x * x;
		w = t * (p0 + t * (p1 + t * (p2 + t * (p3 + t * p4))));
		return x + x * w;
	}
	w = 1.0f - fabsf_(x);
	t = w * 0.5f;
	p = t * (p0 + t * (p1 + t * (p2 + t * (p3 + t * p4))));
	s = sqrtf_(t);
	if (ix >= 0x3F79999A) {                              /* |x| > 0.975 */
		t = pio2_hi - (2.0f * (s + s * p) - pio2_lo);
	} else {
		w = ffrom(fbits(s) & 0xfffff000u);
		c = (t - w * w) / (s + w);
		r = p;
		p = 2.0f * s * r - (pio2_lo - 2.0f * c);
		q = pio4_hi - 2.0f * w;
		t = pio4_hi - (p - q);
	}
	return (hx > 0) ? t : -t;
}

/* ---- tanf: sysdeps/ieee754/flt-32/s_tanf.c, k_tanf.c, e_rem_pio2f.c (2.35: the sincosf reduction in double, built WITHOUT contraction) -- */
CRH_EM float kernelTanf(float x, float y, int iy) {
	const float pio4 = 0x1.921fb4p-1f, pio4lo = 0x1.4442dp-25f,
				T0 = 0x1.555556p-2f, T1 = 0x1.111112p-3f, T2 = 0x1.ba1ba2p-5f, T3 = 0x1.664f48p-6f, T4 = 0x1.226e3ep-7f, T5 = 0x1.d6d22cp-9f, T6 = 0x1.7dbc9p-10f,
				T7 = 0x1.344d9p-11f, T8 = 0x1.026f72p-12f, T9 = 0x1.47e88ap-14f, T10 = 0x1.2b80f4p-14f, T11 = -0x1.375cbep-16f, T12 = 0x1.b2a708p-16f;
	const int32_t hx = (int32_t)fbits(x), ix = hx & 0x7fffffff;
	if (ix < 0x39000000) {                               /* |x| < 2^-13 */
		if ((int)x == 0) {
			if ((ix | (iy + 1)) == 0) return 1.0f / fabsf_(x);
			if (iy == 1) return x;
			return -1.0f / x;
		}
	}
	if (ix >= 0x3f2ca140) {                              /* |x| >= 0.6744 */
		if (hx < 0) { x = -x; y = -y; }
		const float z0 = pio4 - x;
		const float w0 = pio4lo - y;
		x = z0 + w0; y = 0.0f;
		if (fabsf_(x) < 0x1p-13f) return (float)((1 - ((hx >> 30) & 2)) * iy) * (1.0f - (float)(2 * iy) * x);
	}
	float z = x * x;
	float w = z * z;
	float r = T1 + w * (T3 + w * (T5 + w * (T7 + w * (T9 + w * T11))));
	float v = z * (T2 + w * (T4 + w * (T6 + w * (T8 + w * (T10 + w * T12)))));
	float s = z * x;
	r = y + z * (s * (r + v) + y);
	r += T0 * s;
	w = x + r;
	if (ix >= 0x3f2ca140) {
		v = (float)iy;
		return (float)(1 - ((hx >> 30) & 2)) * (v - 2.0f * (x - (w * w / (w + v) - r)));
	}
	if (iy == 1) return w;
	/* -1.0 / (x + r), accurately */
	z = ffrom(fbits(w) & 0xfffff000u);
	v = r - (z - x);
	const float a = -1.0f / w;
	const float t = ffrom(fbits(a) & 0xfffff000u);
	s = 1.0f + t * z;
	return t + a * (s + t * v);
}
CRH_EM float tanf_(float x) {
	const uint32_t bits = fbits(x), ix = bits & 0x7fffffffu;
	if (ix <= 0x3f490fdau) return kernelTanf(x, 0.0f, 1);        /* |x| <~ pi/4 */
	if (ix >= 0x7f800000u) return x - x;                           /* tan(Inf or NaN) is NaN */
	double dx = (double)x;
	int n;
	if (((bits >> 20) & 0x7ffu) <= 0x42eu) {                       /* |x| < 120: reduce_fast, mul and sub NOT fused in this translation unit */
		const double r = dx * CRH_SC_HPI_INV;
		n = ((int32_t)r + 0x800000) >> 24;
		dx = dx - (double)n * CRH_SC_HPI;
	} else {
		dx = reduceLarge(bits, n);
		if (bits >> 31) dx = -dx;
	}
	const float y0 = (float)dx;
	const float y1 = (float)(dx - (double)y0);
	return kernelTanf(y0, y1, 1 - ((n & 1) << 1));
}

/* ---- log10f: sysdeps/ieee754/flt-32/e_log10f.c (calls logf above) ----------------------------------------------------------- */
CRH_EM float log10f_(float x) {
	const float two25 = 3.3554432000e+07f, ivln10 = 4.3429449201e-01f, log10_2hi = 3.0102920532e-01f, log10_2lo = 7.9034151668e-07f;
	int32_t hx = (int32_t)fbits(x);
	int32_t k = 0;
	if (hx < 0x00800000) {
		if ((hx & 0x7fffffff) == 0) return -two25 / fabsf_(x);
		if (hx < 0) return invalidf(x);
		k -= 25; x *= two25;
		hx = (int32_t)fbits(x);
	}
	if (hx >= 0x7f800000) return x + x;
	k += (hx >> 23) - 127;
	const int32_t i = (int32_t)(((uint32_t)k & 0x80000000u) >> 31);
	hx = (hx & 0x007fffff) | ((0x7f - i) << 23);
	const float y = (float)(k + i);
	const float z = y * log10_2lo + ivln10 * logf_(ffrom((uint32_t)hx));
	return z + y * log10_2hi;
}

}  // namespace em
}  // namespace crh

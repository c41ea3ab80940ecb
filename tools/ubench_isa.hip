// ubench_isa.hip — dev micro-benchmark (MI355X): issue cost of the instruction kinds the path-tracing kernel is made of, in SIMD cycles per wave64
// instruction, with 1 and with 4 waves per SIMD (the kernel's occupancy). Found in round 3 the hard way: the packed-float instructions the SLP
// vectoriser likes (v_pk_fma_f32, v_pk_mul_f32, v_pk_add_f32) are several times dearer than the two plain instructions they replace.
//   hipcc --offload-arch=gfx950 -O2 tools/ubench_isa.hip -o c-ray_amd/_lib/ubench_isa && c-ray_amd/_lib/ubench_isa
// Each test: 8 independent accumulator chains x 16 unrolled copies x LOOPS iterations of ONE instruction; clock = s_memtime (shader clock).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
#define LOOPS 256
#define REP8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)

// 32-bit destination, two 32-bit sources
#define DEF_V32(NAME, ASM) \
__global__ void NAME(unsigned long long *out, float seed) { \
	float a0 = seed, a1 = seed + 1, a2 = seed + 2, a3 = seed + 3, a4 = seed + 4, a5 = seed + 5, a6 = seed + 6, a7 = seed + 7; \
	const float b = seed * 0.5f + 1.0f; \
	const unsigned long long t0 = __builtin_amdgcn_s_memtime(); \
	for (int i = 0; i < LOOPS; ++i) { \
		_Pragma("unroll") for (int u = 0; u < 16; ++u) { \
			asm volatile(ASM : "+v"(a0) : "v"(b)); asm volatile(ASM : "+v"(a1) : "v"(b)); asm volatile(ASM : "+v"(a2) : "v"(b)); asm volatile(ASM : "+v"(a3) : "v"(b)); \
			asm volatile(ASM : "+v"(a4) : "v"(b)); asm volatile(ASM : "+v"(a5) : "v"(b)); asm volatile(ASM : "+v"(a6) : "v"(b)); asm volatile(ASM : "+v"(a7) : "v"(b)); \
		} \
	} \
	const unsigned long long t1 = __builtin_amdgcn_s_memtime(); \
	if (threadIdx.x == 0) out[blockIdx.x] = t1 - t0; \
	if (a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 == 12345.678f) out[0] = 0; \
}
// 64-bit register pairs (packed float, double)
typedef float f2 __attribute__((ext_vector_type(2)));
#define DEF_V64(NAME, ASM, T) \
__global__ void NAME(unsigned long long *out, float seed) { \
	T a0 = (T)seed, a1 = a0 + (T)1, a2 = a0 + (T)2, a3 = a0 + (T)3, a4 = a0 + (T)4, a5 = a0 + (T)5, a6 = a0 + (T)6, a7 = a0 + (T)7; \
	const T b = a0 * (T)0.5 + (T)1; \
	const unsigned long long t0 = __builtin_amdgcn_s_memtime(); \
	for (int i = 0; i < LOOPS; ++i) { \
		_Pragma("unroll") for (int u = 0; u < 16; ++u) { \
			asm volatile(ASM : "+v"(a0) : "v"(b)); asm volatile(ASM : "+v"(a1) : "v"(b)); asm volatile(ASM : "+v"(a2) : "v"(b)); asm volatile(ASM : "+v"(a3) : "v"(b)); \
			asm volatile(ASM : "+v"(a4) : "v"(b)); asm volatile(ASM : "+v"(a5) : "v"(b)); asm volatile(ASM : "+v"(a6) : "v"(b)); asm volatile(ASM : "+v"(a7) : "v"(b)); \
		} \
	} \
	const unsigned long long t1 = __builtin_amdgcn_s_memtime(); \
	if (threadIdx.x == 0) out[blockIdx.x] = t1 - t0; \
	T s = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7; \
	if (*(unsigned long long *)&s == 0x123456789ull) out[0] = 0; \
}

DEF_V32(k_fma_f32, "v_fma_f32 %0, %0, %1, %1")
DEF_V32(k_mul_f32, "v_mul_f32 %0, %0, %1")
DEF_V32(k_add_f32, "v_add_f32 %0, %0, %1")
DEF_V32(k_max_f32, "v_max_f32 %0, %0, %1")
DEF_V32(k_max3_f32, "v_max3_f32 %0, %0, %1, %1")
DEF_V32(k_cndmask, "v_cndmask_b32 %0, %0, %1, vcc")
DEF_V32(k_cndmask_sgpr, "v_cndmask_b32_e64 %0, %0, %1, s[20:21]")
DEF_V32(k_cmp_vcc, "v_cmp_lt_f32_e32 vcc, %0, %1")
DEF_V32(k_cmp_sgpr, "v_cmp_lt_f32_e64 s[20:21], %0, %1")
DEF_V32(k_cmp_then_cnd, "v_cmp_lt_f32_e32 vcc, %0, %1\n\tv_cndmask_b32 %0, %0, %1, vcc")
DEF_V32(k_and_b32, "v_and_b32 %0, %0, %1")
DEF_V32(k_bfi_b32, "v_bfi_b32 %0, %0, %1, %1")
DEF_V32(k_med3_f32, "v_med3_f32 %0, %0, %1, %1")
DEF_V32(k_min3_f32, "v_min3_f32 %0, %0, %1, %1")
DEF_V32(k_add_co_u32, "v_add_co_u32 %0, vcc, %0, %1")
DEF_V32(k_mbcnt, "v_mbcnt_lo_u32_b32 %0, %1, %0")
DEF_V32(k_rcp_f32, "v_rcp_f32 %0, %0")
DEF_V32(k_sqrt_f32, "v_sqrt_f32 %0, %0")
DEF_V32(k_mul_lo_u32, "v_mul_lo_u32 %0, %0, %1")
DEF_V32(k_mul_hi_u32, "v_mul_hi_u32 %0, %0, %1")
DEF_V32(k_add_u32, "v_add_u32 %0, %0, %1")
DEF_V32(k_lshl_add_u32, "v_lshl_add_u32 %0, %0, 2, %1")
DEF_V32(k_mov_dpp, "v_mov_b32_dpp %0, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf")
DEF_V32(k_readlane, "v_readlane_b32 s20, %0, 5")
DEF_V32(k_cvt_f32_u32, "v_cvt_f32_u32 %0, %0")
DEF_V32(k_ldexp_f32, "v_ldexp_f32 %0, %0, %1")
DEF_V32(k_div_fixup, "v_div_fixup_f32 %0, %0, %1, %1")
DEF_V64(k_pk_fma_f32, "v_pk_fma_f32 %0, %0, %1, %1", f2)
DEF_V64(k_pk_mul_f32, "v_pk_mul_f32 %0, %0, %1", f2)
DEF_V64(k_pk_add_f32, "v_pk_add_f32 %0, %0, %1", f2)
DEF_V64(k_pk_mov_b32, "v_pk_mov_b32 %0, %0, %1", f2)
DEF_V64(k_fma_f64, "v_fma_f64 %0, %0, %1, %1", double)
DEF_V64(k_mul_f64, "v_mul_f64 %0, %0, %1", double)
DEF_V64(k_add_f64, "v_add_f64 %0, %0, %1", double)
DEF_V64(k_lshl_add_u64, "v_lshl_add_u64 %0, %0, 1, %1", double)
DEF_V64(k_mov_b64, "v_mov_b64 %0, %1", double)

struct T { const char *name; void (*fn)(unsigned long long *, float); };

int main() {
	CK(hipSetDevice(0));
	hipDeviceProp_t prop;
	CK(hipGetDeviceProperties(&prop, 0));
	unsigned long long *out;
	const int cus = prop.multiProcessorCount;
	CK(hipMalloc((void **)&out, sizeof(*out) * cus * 8));
	std::vector<unsigned long long> h(cus * 8);
#define E(k) {#k, k}
	const T tests[] = {E(k_fma_f32), E(k_mul_f32), E(k_add_f32), E(k_max_f32), E(k_max3_f32), E(k_cndmask), E(k_cndmask_sgpr), E(k_cmp_vcc), E(k_cmp_sgpr), E(k_cmp_then_cnd), E(k_and_b32), E(k_bfi_b32), E(k_med3_f32), E(k_min3_f32), E(k_add_co_u32), E(k_mbcnt), E(k_rcp_f32), E(k_sqrt_f32), E(k_mul_lo_u32), E(k_mul_hi_u32),
					   E(k_add_u32), E(k_lshl_add_u32), E(k_mov_dpp), E(k_readlane), E(k_cvt_f32_u32), E(k_ldexp_f32), E(k_div_fixup),
					   E(k_pk_fma_f32), E(k_pk_mul_f32), E(k_pk_add_f32), E(k_pk_mov_b32), E(k_fma_f64), E(k_mul_f64), E(k_add_f64), E(k_lshl_add_u64), E(k_mov_b64)};
	printf("# %s: s_memtime ticks per wave64 instruction (8 independent chains); 100 MHz reference clock ticks x (shader clock / 100 MHz) = SIMD cycles\n", prop.name);
	printf("%-18s %14s %14s\n", "instruction", "1 wave/SIMD", "4 waves/SIMD");
	for (const T &t : tests) {
		double r[2], wall[2];
		for (int occ = 0; occ < 2; ++occ) {
			const int block = occ == 0 ? 256 : 1024;           // 4 waves per block = 1 per SIMD; 16 waves = 4 per SIMD
			hipEvent_t e0, e1;
			CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
			hipLaunchKernelGGL(t.fn, dim3(cus), dim3(block), 0, 0, out, 1.5f);
			CK(hipEventRecord(e0));
			hipLaunchKernelGGL(t.fn, dim3(cus), dim3(block), 0, 0, out, 1.5f);
			CK(hipEventRecord(e1));
			CK(hipDeviceSynchronize());
			float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1));
			wall[occ] = ms * 1e6 / (double)(LOOPS * 16 * 8);           // ns per instruction of one wave (all waves run side by side)
			CK(hipMemcpy(h.data(), out, sizeof(*out) * cus, hipMemcpyDeviceToHost));
			double s = 0;
			for (int i = 0; i < cus; ++i) s += (double)h[i];
			r[occ] = s / cus / (double)(LOOPS * 16 * 8);
		}
		printf("%-18s %14.3f %14.3f   wall ns per wave-instruction: %7.3f %7.3f\n", t.name + 2, r[0], r[1], wall[0], wall[1]);
	}
	return 0;
}

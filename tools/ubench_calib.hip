// ubench_calib.hip — dev micro-benchmark (MI355X; round 5): kernels whose memory traffic and vector-instruction counts are KNOWN, run under the same
// rocprofv3 counter passes as the path-tracing kernel, so that the figures quoted from those counters are calibrated on this kernel's own access patterns:
//   k_gather64   every lane reads RECORDS 64-byte records (four global_load_dwordx4, like a BVH child pair) at hashed indices of a 2 GB array — divergent 64-B gathers, no reuse,
//                eight times the Infinity Cache: bytes that must come from HBM = lanes x RECORDS x 64
//   k_gather128  the same with 128-byte records (eight quarters: a wide node, a path record)
//   k_stream     wide coalesced streaming read of the array (16 B per lane, consecutive lanes consecutive quarters): the pattern MI355X_MICROARCH.md's FETCH_SIZE x 2 was measured on
//   k_write128   every lane writes 128-byte records (eight global_store_dwordx4) of its own, like the path table's records: bytes = lanes x RECORDS x 128
//   k_fma / k_add / k_mix   pure vector-issue loops at 16 waves per CU (4 per SIMD): wave-instructions = waves x LOOPS x 16 x 8 (+ the loop's own handful); k_mix = the node step's
//                blend (fma, min, max, max3, cmp + cndmask)
// The program prints one JSON line per kernel: known bytes / wave-instructions, event time, GB/s or wave-instructions per SIMD-cycle at the 2.4 GHz shader clock.
// tools/calib_table.py joins them with the counters of the rocprofv3 passes (FETCH_SIZE, WRITE_SIZE, SQ_ACTIVE_INST_VALU / SQ_WAVE_CYCLES / SQ_INSTS_VALU / SQ_BUSY_CYCLES).
//   hipcc --offload-arch=gfx950 -O2 tools/ubench_calib.hip -o c-ray_amd/_lib/ubench_calib
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
typedef float f4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ uint32_t hash32(uint32_t x) { x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16; return x; }

#define RECORDS 64
__global__ __launch_bounds__(256) void k_gather64(const f4 *arr, uint32_t recMask, float *out) {
	const uint32_t tid = blockIdx.x * 256u + threadIdx.x;
	f4 acc = {0, 0, 0, 0};
	for (uint32_t i = 0; i < RECORDS; ++i) {
		const uint32_t r = hash32(tid * RECORDS + i) & recMask;          /* a 64-byte record */
		const f4 *p = arr + (size_t)r * 4u;
		acc += p[0] + p[1] + p[2] + p[3];
	}
	if (acc.x + acc.y + acc.z + acc.w == 12345.678f) out[0] = acc.x;
}
__global__ __launch_bounds__(256) void k_gather128(const f4 *arr, uint32_t recMask, float *out) {
	const uint32_t tid = blockIdx.x * 256u + threadIdx.x;
	f4 acc = {0, 0, 0, 0};
	for (uint32_t i = 0; i < RECORDS; ++i) {
		const uint32_t r = hash32(tid * RECORDS + i) & recMask;          /* a 128-byte record */
		const f4 *p = arr + (size_t)r * 8u;
		acc += p[0] + p[1] + p[2] + p[3] + p[4] + p[5] + p[6] + p[7];
	}
	if (acc.x + acc.y + acc.z + acc.w == 12345.678f) out[0] = acc.x;
}
__global__ __launch_bounds__(256) void k_stream(const f4 *arr, size_t quarters, float *out) {
	f4 acc = {0, 0, 0, 0};
	for (size_t i = (size_t)blockIdx.x * 256u + threadIdx.x; i < quarters; i += (size_t)gridDim.x * 256u) acc += arr[i];
	if (acc.x + acc.y + acc.z + acc.w == 12345.678f) out[0] = acc.x;
}
__global__ __launch_bounds__(256) void k_write128(f4 *arr, uint32_t recMask, float seed) {
	const uint32_t tid = blockIdx.x * 256u + threadIdx.x;
	const f4 v = {seed, seed + 1, seed + 2, (float)tid};
	for (uint32_t i = 0; i < RECORDS; ++i) {
		const uint32_t r = (tid * RECORDS + i) & recMask;          /* records of this lane's own, one after the other (a path table's column) */
		f4 *p = arr + (size_t)r * 8u;
		p[0] = v; p[1] = v; p[2] = v; p[3] = v; p[4] = v; p[5] = v; p[6] = v; p[7] = v;
	}
}
#define LOOPS 4096
#define VALU_KERNEL(NAME, BODY) \
__global__ __launch_bounds__(256) void NAME(float *out, float seed) { \
	float a0 = seed, a1 = seed + 1, a2 = seed + 2, a3 = seed + 3, a4 = seed + 4, a5 = seed + 5, a6 = seed + 6, a7 = seed + 7; \
	const float b = seed * 0.5f + 1.0f; \
	for (int i = 0; i < LOOPS; ++i) { \
		_Pragma("unroll") for (int u = 0; u < 16; ++u) { BODY } \
	} \
	if (a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 == 12345.678f) out[0] = a0; \
}
#define ALL8(ASM) asm volatile(ASM : "+v"(a0) : "v"(b)); asm volatile(ASM : "+v"(a1) : "v"(b)); asm volatile(ASM : "+v"(a2) : "v"(b)); asm volatile(ASM : "+v"(a3) : "v"(b)); \
	asm volatile(ASM : "+v"(a4) : "v"(b)); asm volatile(ASM : "+v"(a5) : "v"(b)); asm volatile(ASM : "+v"(a6) : "v"(b)); asm volatile(ASM : "+v"(a7) : "v"(b));
VALU_KERNEL(k_fma, ALL8("v_fma_f32 %0, %0, %1, %1"))
VALU_KERNEL(k_add, ALL8("v_add_f32 %0, %0, %1"))
/* the node step's blend: 3 fma, 2 min / max, 1 max3, 1 cmp, 1 cndmask per eight */
VALU_KERNEL(k_mix, asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(a0) : "v"(b)); asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(a1) : "v"(b)); asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(a2) : "v"(b)); \
	asm volatile("v_min_f32 %0, %0, %1" : "+v"(a3) : "v"(b)); asm volatile("v_max_f32 %0, %0, %1" : "+v"(a4) : "v"(b)); asm volatile("v_max3_f32 %0, %0, %1, %1" : "+v"(a5) : "v"(b)); \
	asm volatile("v_cmp_lt_f32_e32 vcc, %0, %1" : : "v"(a6), "v"(b) : "vcc"); asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a7) : "v"(b) : "vcc");)

/* does the vector ALU skip a half (or three quarters) of a wave whose lanes are all masked off? the same fma loop with exec = lanes 0-31 / lanes 0-15 only */
#define VALU_MASKED(NAME, MASK) \
__global__ __launch_bounds__(256) void NAME(float *out, float seed) { \
	float a0 = seed, a1 = seed + 1, a2 = seed + 2, a3 = seed + 3, a4 = seed + 4, a5 = seed + 5, a6 = seed + 6, a7 = seed + 7; \
	const float b = seed * 0.5f + 1.0f; \
	if ((threadIdx.x & 63u) < MASK) { \
		for (int i = 0; i < LOOPS; ++i) { \
			_Pragma("unroll") for (int u = 0; u < 16; ++u) { ALL8("v_fma_f32 %0, %0, %1, %1") } \
		} \
	} \
	if (a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 == 12345.678f) out[0] = a0; \
}
VALU_MASKED(k_fma_half, 32u)
VALU_MASKED(k_fma_quarter, 16u)

int main(int argc, char **argv) {
	const std::string only = argc > 1 ? argv[1] : "";
	hipDeviceProp_t prop;
	CK(hipGetDeviceProperties(&prop, 0));
	const int cus = prop.multiProcessorCount;
	const size_t bytes = (size_t)2 << 30;
	f4 *arr; float *out;
	CK(hipMalloc((void **)&arr, bytes)); CK(hipMalloc((void **)&out, 4096));
	CK(hipMemset(arr, 0, bytes));
	hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
	const uint32_t blocks = (uint32_t)cus * 4u;          /* 16 waves per CU, like the path-tracing kernel */
	const double lanes = (double)blocks * 256.0, waves = lanes / 64.0;
	auto timed = [&](const char *name, auto launch, double knownBytes, double knownInsts) {
		if (!only.empty() && only != name) return;
		launch(); CK(hipDeviceSynchronize());          /* warm */
		CK(hipEventRecord(e0)); launch(); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
		float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1));
		printf("{\"kernel\": \"%s\", \"ms\": %.4f, \"known_bytes\": %.0f, \"GBps\": %.1f, \"known_valu_wave_instructions\": %.0f, \"wave_instructions_per_simd_cycle_at_2400MHz\": %.4f, \"cus\": %d}\n",
		       name, ms, knownBytes, knownBytes / ms / 1e6, knownInsts, knownInsts / (ms * 1e-3 * 2.4e9 * cus * 4.0), cus);
		fflush(stdout);
	};
	timed("k_gather64", [&]() { hipLaunchKernelGGL(k_gather64, dim3(blocks * 8u), dim3(256), 0, 0, arr, (uint32_t)(bytes / 64u - 1u), out); }, lanes * 8.0 * RECORDS * 64.0, 0);
	timed("k_gather128", [&]() { hipLaunchKernelGGL(k_gather128, dim3(blocks * 8u), dim3(256), 0, 0, arr, (uint32_t)(bytes / 128u - 1u), out); }, lanes * 8.0 * RECORDS * 128.0, 0);
	timed("k_stream", [&]() { hipLaunchKernelGGL(k_stream, dim3(blocks * 4u), dim3(256), 0, 0, arr, bytes / 16u, out); }, (double)bytes, 0);
	timed("k_write128", [&]() { hipLaunchKernelGGL(k_write128, dim3(blocks * 8u), dim3(256), 0, 0, arr, (uint32_t)(bytes / 128u - 1u), 1.0f); }, lanes * 8.0 * RECORDS * 128.0, 0);
	timed("k_fma", [&]() { hipLaunchKernelGGL(k_fma, dim3(blocks), dim3(256), 0, 0, out, 1.0f); }, 0, waves * LOOPS * 16.0 * 8.0);
	timed("k_fma_half", [&]() { hipLaunchKernelGGL(k_fma_half, dim3(blocks), dim3(256), 0, 0, out, 1.0f); }, 0, waves * LOOPS * 16.0 * 8.0);
	timed("k_fma_quarter", [&]() { hipLaunchKernelGGL(k_fma_quarter, dim3(blocks), dim3(256), 0, 0, out, 1.0f); }, 0, waves * LOOPS * 16.0 * 8.0);
	timed("k_add", [&]() { hipLaunchKernelGGL(k_add, dim3(blocks), dim3(256), 0, 0, out, 1.0f); }, 0, waves * LOOPS * 16.0 * 8.0);
	timed("k_mix", [&]() { hipLaunchKernelGGL(k_mix, dim3(blocks), dim3(256), 0, 0, out, 1.0f); }, 0, waves * LOOPS * 16.0 * 8.0);
	return 0;
}

// ubench_l1.hip — dev micro-benchmark (MI355X): what does the vector memory path charge for the access pattern of a BVH node step?
// Every lane runs a dependent chain: load a 64-byte record at a pseudo-random 64-byte-aligned address, derive the next address from the
// data. Grid = CUs x 4 blocks of 256 threads (the occupancy of k_pathtrace: 16 waves per CU). Variants:
//   own4      one lane = one record, 4 x global_load_dwordx4 (what stepNode does)
//   own2/own1 the same with 32- / 16-byte records (2 / 1 loads): is the price per instruction or per line?
//   quad4     the 4 lanes of a quad fetch ONE record with ONE instruction (16 bytes each, 64 contiguous bytes per quad); 4 instructions
//             serve the quad's 4 records; every lane ends up with a quarter of each (the exchange is not timed separately here: the
//             next address is taken from the quarter a lane holds, broadcast with a DPP quad permute)
//   lds4      the same through global_load_lds_dwordx4 + ds_read_b128 (records land in LDS, each lane reads back its own 64 bytes)
// x active lanes per wave (64 / 48 / 32 / 16: "is a step priced per lane?") x working set (1 MB .. 1 GB: L2, Infinity Cache, HBM).
//   hipcc --offload-arch=gfx950 -O3 tools/ubench_l1.hip -o gpurun_out/ubench_l1 && gpurun_out/ubench_l1
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

typedef float f4 __attribute__((ext_vector_type(4)));
typedef uint32_t u4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ uint32_t fold(u4 a) { return a.x ^ a.y ^ a.z ^ a.w; }       /* every dword is used: the compiler cannot narrow the loads */
__device__ __forceinline__ uint32_t mix(uint32_t x) { x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16; return x; }

template <int LOADS>
__global__ __launch_bounds__(256) void k_own(const u4 *buf, uint32_t mask, int iters, int active, uint32_t *out) {
	const uint32_t lane = threadIdx.x & 63u;
	uint32_t idx = mix(blockIdx.x * 256u + threadIdx.x) & mask;
	uint32_t acc = 0;
	if ((int)lane < active) {
		for (int i = 0; i < iters; ++i) {
			const u4 *p = buf + (size_t)idx * 4u;
			u4 a = p[0], b = a, c = a, d = a;
			if (LOADS >= 2) b = p[1];
			if (LOADS >= 4) { c = p[2]; d = p[3]; }
			const uint32_t v = fold(a) ^ fold(b) * 3u ^ fold(c) * 5u ^ fold(d) * 7u;
			acc += v;
			idx = mix(v + idx) & mask;
		}
	}
	out[blockIdx.x * 256u + threadIdx.x] = acc;
}

__device__ __forceinline__ uint32_t quadBcast(uint32_t v, int j) {
	switch (j) {
		case 0: return (uint32_t)__builtin_amdgcn_mov_dpp((int)v, 0x00, 0xF, 0xF, true);   // quad_perm [0,0,0,0]
		case 1: return (uint32_t)__builtin_amdgcn_mov_dpp((int)v, 0x55, 0xF, 0xF, true);   // [1,1,1,1]
		case 2: return (uint32_t)__builtin_amdgcn_mov_dpp((int)v, 0xAA, 0xF, 0xF, true);   // [2,2,2,2]
		default: return (uint32_t)__builtin_amdgcn_mov_dpp((int)v, 0xFF, 0xF, 0xF, true);  // [3,3,3,3]
	}
}

// quad-cooperative: whole quads are active or not (active = lanes, rounded down to quads)
__global__ __launch_bounds__(256) void k_quad(const u4 *buf, uint32_t mask, int iters, int active, uint32_t *out) {
	const uint32_t lane = threadIdx.x & 63u, q = lane & 3u;
	uint32_t idx = mix(blockIdx.x * 256u + threadIdx.x) & mask;
	uint32_t acc = 0;
	if ((int)lane < active) {
		for (int i = 0; i < iters; ++i) {
			u4 r[4];
#pragma unroll
			for (int m = 0; m < 4; ++m) {
				const uint32_t im = quadBcast(idx, m);
				r[m] = buf[(size_t)im * 4u + q];
			}
			// the record of member m: lane q holds quarter q. Next index of THIS lane's record = f(first dword of quarter 0) -> from lane 0 of the quad, register r[q]
			uint32_t v0 = quadBcast(fold(r[0]), 0), v1 = quadBcast(fold(r[1]), 0), v2 = quadBcast(fold(r[2]), 0), v3 = quadBcast(fold(r[3]), 0);
			const uint32_t v = q == 0 ? v0 : q == 1 ? v1 : q == 2 ? v2 : v3;
			acc += v;
			idx = mix(v + idx) & mask;
		}
	}
	out[blockIdx.x * 256u + threadIdx.x] = acc;
}

// the same through LDS: global_load_lds_dwordx4 puts lane l's 16 bytes at M0 base + l * 16; 4 instructions = 4 KB per wave; each lane then reads its record (64 contiguous bytes)
__global__ __launch_bounds__(256) void k_lds(const u4 *buf, uint32_t mask, int iters, int active, uint32_t *out) {
	__shared__ u4 s_rec[4 * 4 * 64];          // per wave: 4 members x 64 lanes x 16 B
	const uint32_t lane = threadIdx.x & 63u, q = lane & 3u, w = threadIdx.x >> 6;
	uint32_t idx = mix(blockIdx.x * 256u + threadIdx.x) & mask;
	uint32_t acc = 0;
	if ((int)lane < active) {
		for (int i = 0; i < iters; ++i) {
#pragma unroll
			for (int m = 0; m < 4; ++m) {
				const uint32_t im = quadBcast(idx, m);
				const u4 *src = buf + (size_t)im * 4u + q;
				__builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)src,
												 (__attribute__((address_space(3))) void *)&s_rec[(w * 4 + m) * 64], 16, 0, 0);
			}
			__builtin_amdgcn_s_waitcnt(0x0070);      // vmcnt(0) (gfx9 encoding: vmcnt low bits 3:0 = 0, expcnt 7, lgkmcnt 0 -> conservative)
			__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
			// member q's record of this quad: 64 bytes at slab q, lanes 4g .. 4g+3
			const u4 *mine = &s_rec[(w * 4 + q) * 64 + (lane & ~3u)];
			const u4 a = mine[0], b = mine[1], c = mine[2], d = mine[3];
			const uint32_t v = fold(a) ^ fold(b) * 3u ^ fold(c) * 5u ^ fold(d) * 7u;
			acc += v;
			idx = mix(v + idx) & mask;
		}
	}
	out[blockIdx.x * 256u + threadIdx.x] = acc;
}

// stores: the path-table pattern — 4 x 16-byte stores of one lane into its own 128-byte record (SHADE), or one 64-byte... (no wider store exists)
template <int STORES>
__global__ __launch_bounds__(256) void k_store(u4 *buf, uint32_t mask, int iters, int active, uint32_t *out) {
	const uint32_t lane = threadIdx.x & 63u;
	uint32_t idx = mix(blockIdx.x * 256u + threadIdx.x) & mask;
	if ((int)lane < active) {
		for (int i = 0; i < iters; ++i) {
			u4 *p = buf + (size_t)idx * 8u;           // 128-byte records
			const u4 v = {idx, (uint32_t)i, lane, 7u};
			p[0] = v;
			if (STORES >= 2) p[1] = v;
			if (STORES >= 4) { p[2] = v; p[3] = v; }
			idx = mix(idx + (uint32_t)i) & mask;
		}
	}
	out[blockIdx.x * 256u + threadIdx.x] = idx;
}

int main(int argc, char **argv) {
	int dev = 0;
	CK(hipSetDevice(dev));
	hipDeviceProp_t prop;
	CK(hipGetDeviceProperties(&prop, dev));
	const int cus = prop.multiProcessorCount, grid = cus * 4;
	const size_t maxBytes = (size_t)1 << 30;
	u4 *buf; uint32_t *out;
	CK(hipMalloc((void **)&buf, maxBytes));
	CK(hipMalloc((void **)&out, (size_t)grid * 256 * 4));
	{   // pseudo-random contents
		std::vector<uint32_t> h(maxBytes / 4);
		uint32_t s = 12345u;
		for (auto &x : h) { s = s * 1664525u + 1013904223u; x = s; }
		CK(hipMemcpy(buf, h.data(), maxBytes, hipMemcpyHostToDevice));
	}
	hipEvent_t a, b;
	CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
	const int iters = argc > 1 ? atoi(argv[1]) : 2000;
	printf("# %s, %d CUs, grid %d x 256, %d dependent steps per lane\n", prop.name, cus, grid, iters);
	printf("%-8s %8s %6s %10s %12s %12s %12s\n", "variant", "set", "lanes", "ms", "ns/wave-step", "ns/lane-step", "GB/s useful");
	const size_t sets[] = {(size_t)1 << 20, (size_t)16 << 20, (size_t)160 << 20, (size_t)1 << 30};
	const int lanesList[] = {64, 48, 32, 16};
	struct V { const char *name; int bytes; int kind; } vars[] = {{"own4", 64, 0}, {"own2", 32, 1}, {"own1", 16, 2}, {"quad4", 64, 3}, {"lds4", 64, 4}, {"store4", 64, 5}, {"store2", 32, 6}, {"store1", 16, 7}};
	for (const V &v : vars) for (size_t set : sets) for (int lanes : lanesList) {
		if (v.kind >= 1 && v.kind != 3 && v.kind != 4 && lanes != 64 && lanes != 32) continue;          // fewer combinations for the side variants
		const uint32_t recBytes = v.kind >= 5 ? 128u : 64u;
		const uint32_t mask = (uint32_t)(set / recBytes) - 1u;
		float best = 1e30f;
		for (int rep = 0; rep < 3; ++rep) {
			CK(hipEventRecord(a));
			switch (v.kind) {
				case 0: hipLaunchKernelGGL(k_own<4>, dim3(grid), dim3(256), 0, 0, buf, mask, iters, lanes, out); break;
				case 1: hipLaunchKernelGGL(k_own<2>, dim3(grid), dim3(256), 0, 0, buf, mask, iters, lanes, out); break;
				case 2: hipLaunchKernelGGL(k_own<1>, dim3(grid), dim3(256), 0, 0, buf, mask, iters, lanes, out); break;
				case 3: hipLaunchKernelGGL(k_quad, dim3(grid), dim3(256), 0, 0, buf, mask, iters, lanes, out); break;
				case 4: hipLaunchKernelGGL(k_lds, dim3(grid), dim3(256), 0, 0, buf, mask, iters, lanes, out); break;
				case 5: hipLaunchKernelGGL(k_store<4>, dim3(grid), dim3(256), 0, 0, buf, mask, iters, lanes, out); break;
				case 6: hipLaunchKernelGGL(k_store<2>, dim3(grid), dim3(256), 0, 0, buf, mask, iters, lanes, out); break;
				default: hipLaunchKernelGGL(k_store<1>, dim3(grid), dim3(256), 0, 0, buf, mask, iters, lanes, out); break;
			}
			CK(hipEventRecord(b));
			CK(hipEventSynchronize(b));
			float ms; CK(hipEventElapsedTime(&ms, a, b));
			if (ms < best) best = ms;
		}
		const double waveSteps = (double)iters;          // per wave
		const double nsWave = best * 1e6 / waveSteps;
		const double laneSteps = (double)grid * 4 * lanes * iters;
		printf("%-8s %6zuMB %6d %10.3f %12.1f %12.4f %12.1f\n", v.name, set >> 20, lanes, best, nsWave, best * 1e6 / laneSteps * (double)(grid * 4), laneSteps * v.bytes / (best * 1e-3) / 1e9);
		fflush(stdout);
	}
	return 0;
}

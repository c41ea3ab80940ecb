/*
 * hipemu_selftest.cpp — known answers for the HIP-on-CPU shim itself (tests/emu/hipemu; run by tests/test_kernel_emu.py): ballots with
 * exited lanes, shuffles, readfirstlane, mbcnt ranks, the block barrier, atomics across blocks, a lane-0 spin lock with s_sleep,
 * and the CRH_LOCKSTEP rendezvous.
 */
#include <hip/hip_runtime.h>

#include <cstdio>
#include <vector>

#define CHECK(cond) do { if (!(cond)) { fprintf(stderr, "hipemu selftest: %s failed (line %d)\n", #cond, __LINE__); return 1; } } while (0)

__global__ void k_collectives(unsigned long long *out, unsigned *ranks, unsigned *sums, unsigned *firsts) {
	const unsigned lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
	if (wave == 1 && lane >= 40u) return;                       /* lanes that left the kernel are inactive in later collectives */
	const unsigned long long odd = __ballot(lane & 1u);
	const unsigned rank = __builtin_amdgcn_mbcnt_hi((unsigned)(odd >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)odd, 0u));
	unsigned v = lane;
	for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
	unsigned f = 0;
	if (lane == 0) f = 1000u + wave;
	f = __builtin_amdgcn_readfirstlane(f);
	if (lane == 0) out[blockIdx.x * 4 + wave] = odd;
	ranks[blockIdx.x * 256 + threadIdx.x] = rank;
	if (lane == 0) { sums[blockIdx.x * 4 + wave] = v; firsts[blockIdx.x * 4 + wave] = f; }
}

__global__ void k_block(unsigned *counter, unsigned *perBlock, int *lockWord, unsigned *guarded) {
	__shared__ unsigned s_sum;
	__shared__ unsigned s_turns[4];
	const unsigned lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
	if (threadIdx.x == 0) s_sum = 0;
	__syncthreads();
	atomicAdd(&s_sum, threadIdx.x);
	__syncthreads();
	if (threadIdx.x == 0) perBlock[blockIdx.x] = s_sum;
	atomicAdd(counter, 1u);
	/* "every lane reads, lane 0 writes" with the rendezvous the kernels mark: all 64 reads see the old value */
	if (lane == 0) s_turns[wave] = 7u;
	CRH_LOCKSTEP();
	const unsigned seen = s_turns[wave];
	CRH_LOCKSTEP();
	if (lane == 0) s_turns[wave] = 8u;
	if (seen != 7u) atomicAdd(counter, 1000000u);
	/* a spin lock taken by lane 0 of every wave of every block (global word): the others yield with s_sleep */
	for (int round = 0; round < 3; ++round) {
		if (lane == 0) {
			while (atomicCAS(lockWord, 0, 1) != 0) __builtin_amdgcn_s_sleep(2);
		}
		CRH_LOCKSTEP();
		if (lane == 0) { const unsigned g = *guarded; __builtin_amdgcn_s_sleep(1); *guarded = g + 1u; }       /* not atomic: the lock makes it safe */
		CRH_LOCKSTEP();
		if (lane == 0) __hip_atomic_store(lockWord, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
	}
}

int main() {
	const unsigned blocks = 5;
	unsigned long long *out; unsigned *ranks, *sums, *firsts, *counter, *perBlock, *guarded; int *lockWord;
	CHECK(hipMalloc(&out, blocks * 4 * sizeof(*out)) == hipSuccess);
	CHECK(hipMalloc(&ranks, blocks * 256 * sizeof(unsigned)) == hipSuccess);
	CHECK(hipMalloc(&sums, blocks * 4 * sizeof(unsigned)) == hipSuccess);
	CHECK(hipMalloc(&firsts, blocks * 4 * sizeof(unsigned)) == hipSuccess);
	CHECK(hipMalloc(&counter, sizeof(unsigned)) == hipSuccess && hipMalloc(&perBlock, blocks * sizeof(unsigned)) == hipSuccess);
	CHECK(hipMalloc(&guarded, sizeof(unsigned)) == hipSuccess && hipMalloc(&lockWord, sizeof(int)) == hipSuccess);
	CHECK(hipMemset(counter, 0, sizeof(unsigned)) == hipSuccess && hipMemset(guarded, 0, sizeof(unsigned)) == hipSuccess && hipMemset(lockWord, 0, sizeof(int)) == hipSuccess);
	CHECK(hipMemset(ranks, 0xFF, blocks * 256 * sizeof(unsigned)) == hipSuccess);
	hipLaunchKernelGGL(k_collectives, dim3(blocks), dim3(256), 0, nullptr, out, ranks, sums, firsts);
	for (unsigned b = 0; b < blocks; ++b)
		for (unsigned w = 0; w < 4; ++w) {
			const unsigned live = w == 1 ? 40u : 64u;
			const unsigned long long all = 0xAAAAAAAAAAAAAAAAull, expect = live == 64u ? all : (all & ((1ull << live) - 1ull));
			CHECK(out[b * 4 + w] == expect);
			unsigned sum = 0;
			for (unsigned l = 0; l < live; ++l) sum += l;
			if (live == 64u) CHECK(sums[b * 4 + w] == sum);            /* (a shuffle from an exited lane returns the caller's own value: not checked) */
			CHECK(firsts[b * 4 + w] == 1000u + w);
			for (unsigned l = 0; l < live; ++l) CHECK(ranks[b * 256 + w * 64 + l] == l / 2u);
			for (unsigned l = live; l < 64u; ++l) CHECK(ranks[b * 256 + w * 64 + l] == 0xFFFFFFFFu);
		}
	hipLaunchKernelGGL(k_block, dim3(blocks), dim3(256), 0, nullptr, counter, perBlock, lockWord, guarded);
	CHECK(*counter == blocks * 256u);
	for (unsigned b = 0; b < blocks; ++b) CHECK(perBlock[b] == 255u * 256u / 2u);
	CHECK(*guarded == blocks * 4u * 3u);
	hipEvent_t a, e;
	float ms = -1.0f;
	CHECK(hipEventCreate(&a) == hipSuccess && hipEventCreate(&e) == hipSuccess && hipEventRecord(a, nullptr) == hipSuccess && hipEventRecord(e, nullptr) == hipSuccess);
	CHECK(hipEventElapsedTime(&ms, a, e) == hipSuccess && ms >= 0.0f);
	printf("hipemu selftest ok\n");
	return 0;
}

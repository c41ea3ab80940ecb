/*
 * kernel_emu.cpp — libcray_hip_emu.so: c-ray_amd/csrc/cray_hip.hip (the product's kernels AND its C-ABI host code, the same source
 * file, unmodified) compiled for the host against the HIP-on-CPU shim in hipemu/. TEST INFRASTRUCTURE ONLY: the CPU-only test tier
 * runs the wave machine of k_pathtrace — scheduler, id stacks, path table, work queue, staging, fold — and the host side of
 * crh_render_tiles through the very C-ABI the GPU tests use, and compares the frames with the oracle bit for bit
 * (tests/test_kernel_emu.py). The GPU BVH builder is bvh_emu.cpp. Never loaded by the product, bench.py or the GPU tests.
 */
#include "../../c-ray_amd/csrc/cray_hip.hip"

extern "C" {
/* emulation bookkeeping: {kernel launches, blocks, wave collectives resolved, lane switches} */
void crh_emu_stats(uint64_t out[4]) {
	const hipemu::Stats s = hipemu::stats();
	out[0] = s.launches; out[1] = s.blocks; out[2] = s.collectives; out[3] = s.switches;
}
}

/*
 * bvh_emu.cpp — the GPU BVH builder (c-ray_amd/csrc/bvh_build.hip, unmodified: level-synchronous binning, DPP row scans for the SAH
 * sweeps, one wave per small subtree) compiled for the host against the HIP-on-CPU shim; part of libcray_hip_emu.so. TEST INFRASTRUCTURE.
 */
#include "../../c-ray_amd/csrc/bvh_build.hip"

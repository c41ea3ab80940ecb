"""c-ray_amd — MI355X-native path-tracing hot path for c-ray.

The directory name is not a Python identifier; load it as package `cray_amd`:
    from __graft_entry__ import load_package; cray_amd = load_package()
Contents: csrc/ (HIP kernels + C-ABI, the product), host/ (C flattener that binds to the reference's
`struct world`), abi.py / api.py (ctypes mirror of include/cray_hip.h), tiles.py (tile.c mirror),
render.py (renderFrame()-shaped driver: tile ownership per rank + RCCL framebuffer reduce).
"""

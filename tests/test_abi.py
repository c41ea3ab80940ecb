"""The C-ABI: struct layouts of abi.py match include/cray_hip.h, the library loads and exports every symbol,
and — without a GPU — refuses to compute instead of falling back to a CPU path."""
import ctypes as C
import os
import re
import subprocess

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
STRUCTS = {"crh_bvh_node": "BvhNode", "crh_poly": "Poly", "crh_instance": "Instance", "crh_mesh": "Mesh", "crh_sphere": "Sphere",
           "crh_material": "Material", "crh_gnode": "GNode", "crh_texture": "Texture", "crh_camera": "Camera",
           "crh_scene_desc": "SceneDesc", "crh_render_params": "RenderParams", "crh_counters": "Counters", "crh_hit": "Hit",
           "crh_blob_prefs": "BlobPrefs", "crh_tile": "Tile"}


def test_struct_sizes_match_header(pkg, tmp_path):
    src = tmp_path / "sizes.c"
    body = "\n".join(f'printf("{c} %zu\\n", sizeof({c}));' for c in STRUCTS)
    src.write_text('#include <stdio.h>\n#include "cray_hip.h"\nint main(void){' + body + "return 0;}\n")
    exe = tmp_path / "sizes"
    subprocess.check_call(["gcc", "-I" + os.path.join(REPO, "include"), str(src), "-o", str(exe)])
    out = subprocess.check_output([str(exe)]).decode()
    for line in out.splitlines():
        cname, size = line.split()
        assert C.sizeof(getattr(pkg.abi, STRUCTS[cname])) == int(size), cname
    assert C.sizeof(pkg.abi.Instance) == 128 and C.sizeof(pkg.abi.BvhNode) == 32 and C.sizeof(pkg.abi.Poly) == 40


def test_library_exports_every_declared_symbol(pkg):
    header = open(os.path.join(REPO, "include", "cray_hip.h")).read()
    declared = set(re.findall(r"\b(crh_[a-z0-9_]+)\s*\(", header))
    assert declared == set(pkg.abi.EXPORTED_SYMBOLS), declared ^ set(pkg.abi.EXPORTED_SYMBOLS)
    lib = pkg.api.library()
    for name in declared:
        assert hasattr(lib, name), name
    assert lib.crh_abi_version() == pkg.abi.ABI_VERSION


def test_no_cpu_fallback_without_device(pkg):
    """On a box without a GPU the product must refuse (CRH_ERR_NO_DEVICE), never compute on the CPU."""
    if pkg.api.device_count() > 0:
        pytest.skip("a HIP device is present")
    with pytest.raises(pkg.api.CrhError) as e:
        pkg.api.Context(0)
    assert e.value.code == pkg.abi.ERR_NO_DEVICE


def test_product_does_not_reference_the_oracle():
    """Only tests/, bench.py (cpu_baseline) and __graft_entry__.smoke() may touch oracle/."""
    for root, _, files in os.walk(os.path.join(REPO, "c-ray_amd")):
        for f in files:
            if f.endswith((".py", ".c", ".h", ".cpp", ".hip")):
                text = open(os.path.join(root, f), errors="replace").read()
                assert "oracle_py" not in text and "libcray_oracle" not in text and "cray_oracle.h" not in text, os.path.join(root, f)


def test_default_library_holds_no_experimental_kernel_and_no_emulation(pkg):
    """The library the product loads is the measured one: it holds ONE form of the path-tracing kernel, k_pathtrace_roll (round 4: the other two forms of
    CRH_OPT_KERNEL, k_pathtrace and k_pathtrace_wg, live in csrc/pathtrace_alt.h and are compiled only with -DCRH_WITH_ALT_KERNELS: the emulation tier and A/B
    variant libraries), the product sources carry at most the one dev probe macro (CRH_EXP_ABS_TIMES) — the measured-negative experiments are patches under
    profiles/ —, it is not the CPU emulation, and the product sources never name the emulation library."""
    lib = os.path.join(REPO, "c-ray_amd", "_lib", "libcray_hip.so")
    blob = open(lib, "rb").read()
    assert b"k_pathtrace_roll" in blob and b"k_pathtrace_wgIL" not in blob and b"11k_pathtraceIL" not in blob and b"crh_emu_stats" not in blob and b"hipemu" not in blob
    assert os.path.getsize(lib) < 3 << 20
    exp = 0
    for root, _, files in os.walk(os.path.join(REPO, "c-ray_amd")):
        for f in files:
            if f.endswith((".py", ".c", ".h", ".cpp", ".hip")):
                text = open(os.path.join(root, f), errors="replace").read()
                assert "libcray_hip_emu" not in text, os.path.join(root, f)
                exp += len(re.findall(r"CRH_EXP_", text))
    assert exp <= 8, exp


def test_build_script_knows_every_source_of_the_library(pkg):
    """c-ray_amd/build.py rebuilds when a source is newer than the library: every file the two .hip sources #include (transitively, from csrc/ and include/)
    must be in its dependency list — round 3's list missed csrc/pathtrace_roll.h, the file that holds the hot kernel."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("crh_build", os.path.join(REPO, "c-ray_amd", "build.py"))
    build = importlib.util.module_from_spec(spec); spec.loader.exec_module(build)
    deps = {os.path.realpath(d) for d in build.deps()}
    dirs = [os.path.join(REPO, "c-ray_amd", "csrc"), os.path.join(REPO, "include")]
    seen, todo = set(), [os.path.join(dirs[0], "cray_hip.hip"), os.path.join(dirs[0], "bvh_build.hip"), os.path.join(dirs[0], "scene_compile.cpp")]
    while todo:
        f = todo.pop()
        if f in seen:
            continue
        seen.add(f)
        for inc in re.findall(r'#\s*include\s+"([^"]+)"', open(f).read()):
            for d in dirs:
                if os.path.exists(os.path.join(d, inc)):
                    todo.append(os.path.join(d, inc))
    assert len(seen) >= 8 and any(f.endswith("pathtrace_roll.h") for f in seen)
    for f in seen:
        assert os.path.realpath(f) in deps, f
    # ... and an edit to the hot kernel's file makes the library stale
    roll = os.path.join(dirs[0], "pathtrace_roll.h")
    lib = build.LIB
    if os.path.exists(lib):
        st = os.stat(roll)
        try:
            os.utime(roll, (st.st_atime, os.path.getmtime(lib) + 10))
            assert not build.up_to_date()
        finally:
            os.utime(roll, (st.st_atime, st.st_mtime))


def test_blob_roundtrip(pkg, golden_blob, tmp_path):
    scene = pkg.api.Scene(golden_blob("fence"))
    out = str(tmp_path / "copy.blob")
    rc = pkg.api.library().crh_blob_save(out.encode(), scene.ptr, C.byref(scene.prefs))
    assert rc == 0
    assert open(out, "rb").read() == open(golden_blob("fence"), "rb").read()
    with pytest.raises(pkg.api.CrhError):
        pkg.api.Scene(str(tmp_path / "missing.blob"))


def test_podbuf_allocation_regimes(tmp_path):
    """scene_compile.h's PodBuf — the layout compiler's big arrays — below and above the size from which its blocks sit on 2 MB boundaries, ask for transparent huge
    pages and grow by copying (round 4: the compile was bound by page faults): contents survive every growth, alignment, exact first reservation
    (tests/emu/podbuf_check.cpp). The CPU tier's fixtures never reach that size; hdr.json's texels do."""
    import os
    import subprocess
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = str(tmp_path / "podbuf_check")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-I" + os.path.join(repo, "include"), "-I" + os.path.join(repo, "c-ray_amd", "csrc"),
                           os.path.join(repo, "tests", "emu", "podbuf_check.cpp"), "-o", exe])
    r = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0 and "podbuf_check ok" in r.stdout, r.stdout + r.stderr


"""CPU tier: the kernels of libcray_hip.so themselves — not a restatement of them — run on the CPU and are held to the GPU tier's bar.

tests/emu/libcray_hip_emu.so is c-ray_amd/csrc/cray_hip.hip (k_pathtrace, k_pathtrace_wg, k_trace_rays, k_fold_black, k_to_srgb8 AND the
C-ABI host code around them: work planning, queues, launches, counters) and csrc/bvh_build.hip (the GPU BVH builder) compiled unmodified
against a HIP-on-CPU shim
(tests/emu/hipemu: every lane a fiber, 64-lane waves that meet at ballots / shuffles / readfirstlane, blocks with __syncthreads, LDS as
block-local statics, atomics as atomics). The tests below start the GPU tier's own test functions (`-m gpu`) in a child pytest whose
api.py loads that library (CRH_LIB) — same tests, same source, same C-ABI, no GPU: frames must equal the reference's float buffers bit
for bit, ray records the oracle's, every dispatch decomposition / scheduler option / kernel form the same frame.

What this tier adds to tests/test_emu_parity.py (which pins the LANE code): the wave machine — scheduler, id stacks, path table, shade
class batches, work queue, tapered units, staging and the in-order fold, the workgroup kernel's lock protocol — and the host side of
crh_render_tiles. What it cannot see: anything that depends on the hardware's timing, register allocation or memory model (the GPU tier
keeps that).

TEST INFRASTRUCTURE: the emulation library is never loaded by the product, by bench.py or by the GPU tier.
"""
import os
import re
import subprocess
import sys

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EMU_DIR = os.path.join(REPO, "tests", "emu")
EMU_LIB = os.path.join(EMU_DIR, "libcray_hip_emu.so")


@pytest.fixture(scope="module")
def emu_lib():
    from conftest import locked_make          # (several xdist workers may get here at once)
    locked_make(["make", "-s", "-C", EMU_DIR, "libcray_hip_emu.so"])
    return EMU_LIB


GPU_TIER_JOBS = {     # name -> (files of the GPU tier, -k selection, number of tests that must run and pass)
    "trace_rays": (["test_gpu_parity.py"], "test_trace_rays_bit_exact and not baseline_configs", 6),
    "frames": (["test_gpu_parity.py"], "test_image_parity_vs_reference", 6),
    "schedules": (["test_gpu_parity.py"], "shade_class_batches or dispatch_decompositions or split_pixels or interactive_mode or edge_cases or zero_component or srgb8_matches or error_paths or round_limit or upload_lifecycle or wide_walk_option", 13),
    "stream": (["test_gpu_parity.py"], "streaming_form or wide_walk_meets", 3),          # round 6: the streaming form (walk / shade + refill / fold kernels) against the rolling kernel and the fixtures; the 4-ary walk inside the tolerance gates
    "rare_and_wg": (["test_nodes.py", "test_volumes.py", "test_gpu_parity.py"], "test_gpu_node_zoo or test_gpu_volumes", 3),          # (test_gpu_volumes renders with both kernel forms;
    # test_workgroup_kernel_is_bit_identical_to_the_wave_kernel passes here too, but the lock's polling takes a minute of emulation)
    "bvh": (["test_bvh_build.py"], "not cfg2_hdr and not soup_1m", 8),
    # the reference's own program with renderer.c replaced (c-ray-hip: its main.c, JSON / OBJ loaders, encoders, renderer_hip.c, flatten.c, the GPU
    # BVH builder behind buildBottomLevelBvh) and its cluster worker, bound to the emulation library (LD_LIBRARY_PATH -> tests/emu/_dropin_libs): the drop-in boundary on the CPU
    "dropin": (["test_gpu_parity.py"], "dropin_binary or cluster_worker", 3),
}
ROLL_FIXTURES = ["cfg1_scene", "refraction", "volumes", "nodezoo", "glowmetal"]


@pytest.fixture(scope="module")
def children(emu_lib):
    """Every child process of this module, started at once (they are independent; each runs the emulation on two OS threads): the module
    takes as long as its slowest child instead of the sum."""
    import tempfile
    env = dict(os.environ, CRH_LIB=emu_lib, CRH_ALLOW_EMULATION="1", CRH_DROPIN_LIBDIR=os.path.join(EMU_DIR, "_dropin_libs"), HIPEMU_CUS="2", HIPEMU_THREADS="3")
    procs = {}

    def start(name, cmd, **more_env):
        out = tempfile.TemporaryFile(mode="w+")
        procs[name] = (subprocess.Popen(cmd, env=dict(env, **more_env), cwd=REPO, stdout=out, stderr=subprocess.STDOUT, text=True), out)

    for name, (files, select, _) in GPU_TIER_JOBS.items():
        start(name, [sys.executable, "-m", "pytest", *[os.path.join(REPO, "tests", f) for f in files], "-m", "gpu", "-q", "-x", "-p", "no:cacheprovider", "-k", select])
    start("roll", [sys.executable, os.path.join(EMU_DIR, "render_fixture.py"), "0", *ROLL_FIXTURES])
    start("steps", [sys.executable, os.path.join(EMU_DIR, "sched_counts.py"), *PINNED_STEPS])
    start("fuzz", [sys.executable, os.path.join(REPO, "tools", "emu_fuzz.py"), "--seeds", "0:9"])
    start("fuzz_stream", [sys.executable, os.path.join(REPO, "tools", "emu_fuzz.py"), "--seeds", "0:4", "--fixtures", "fence,refraction,glowmetal,volumes"], FUZZ_KERNEL="3")
    start("fuzz_split", [sys.executable, os.path.join(REPO, "tools", "emu_fuzz_split.py"), "--seeds", "0:8"])
    start("fuzz_bvh", [sys.executable, os.path.join(REPO, "tools", "emu_fuzz_bvh.py"), "--seeds", "0:19"])
    # the top levels' path (per-chunk bin rows + k_fold_bins instead of global atomics) is taken from 64 chunks per node on: CRH_BVH_FOLD_MIN_CHUNKS=1 sends these small meshes through it
    start("fuzz_bvh_fold", [sys.executable, os.path.join(REPO, "tools", "emu_fuzz_bvh.py"), "--seeds", "20:33"], CRH_BVH_FOLD_MIN_CHUNKS="1")
    start("fuzz_rays", [sys.executable, os.path.join(REPO, "tools", "emu_fuzz_rays.py"), "--seeds", "0:6"])

    def finish(name, timeout=1700):
        proc, out = procs[name]
        try:
            proc.wait(timeout=timeout)
        except subprocess.TimeoutExpired:
            proc.kill()
            raise
        out.seek(0)
        return proc.returncode, out.read()

    yield finish
    for proc, out in procs.values():
        if proc.poll() is None:
            proc.kill()
        out.close()


def run_gpu_tier_on_emulation(children, job):
    """The selected `-m gpu` tests, run by a child pytest against the emulation library: every one of them must have run and passed."""
    rc, text = children(job)
    expect_passed = GPU_TIER_JOBS[job][2]
    tail = text[-4000:]
    assert rc == 0, tail
    m = re.search(r"(\d+) passed", text)
    assert m and int(m.group(1)) == expect_passed, f"expected {expect_passed} tests to run on the emulation:\n{tail}"
    assert "skipped" not in text.strip().splitlines()[-1], tail


def test_the_shim_itself(tmp_path):
    """Known answers for the emulation runtime: ballots with exited lanes, shuffles, ranks, barrier, atomics, a spin lock, the rendezvous."""
    exe = str(tmp_path / "hipemu_selftest")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-pthread", "-I" + os.path.join(EMU_DIR, "hipemu"), "-x", "c++",
                           os.path.join(EMU_DIR, "hipemu_selftest.cpp"), os.path.join(EMU_DIR, "hipemu", "hipemu.cpp"), "-o", exe])
    for threads in ("1", "4"):
        r = subprocess.run([exe], env=dict(os.environ, HIPEMU_THREADS=threads), capture_output=True, text=True, timeout=120)
        assert r.returncode == 0 and "selftest ok" in r.stdout, r.stdout + r.stderr


def test_emulation_library_is_the_product_source(emu_lib):
    """The emulation builds the product's own file (no copy of the kernels under tests/) and exports the whole C-ABI."""
    src = open(os.path.join(EMU_DIR, "kernel_emu.cpp")).read()
    assert '#include "../../c-ray_amd/csrc/cray_hip.hip"' in src
    assert "__global__" not in src, "kernel code belongs to c-ray_amd/csrc, not to the emulation harness"
    syms = subprocess.check_output(["nm", "-D", "--defined-only", emu_lib], text=True)
    header = open(os.path.join(REPO, "include", "cray_hip.h")).read()
    declared = set(re.findall(r"\b(crh_[a-z0-9_]+)\s*\(", header))
    exported = set(re.findall(r" T (crh_[a-z0-9_]+)", syms))
    assert declared - exported == set(), sorted(declared - exported)


def test_k_trace_rays_on_emulation(children):
    run_gpu_tier_on_emulation(children, "trace_rays")


def test_k_pathtrace_frames_on_emulation(children):
    """The wave machine renders the six scene fixtures: the reference's float buffer, bit for bit."""
    run_gpu_tier_on_emulation(children, "frames")


def test_schedules_decompositions_and_edge_cases_on_emulation(children):
    """Shade-class batches, tiles / pass chunks / unit sizes / taper levels, split pixels (pass segments folded behind the kernel), the Halton sampler, empty / ragged / single-pixel dispatches,
    bounces <= 0 (k_fold_black), degenerate rays, k_to_srgb8, the error paths of the C-ABI."""
    run_gpu_tier_on_emulation(children, "schedules")


def test_streaming_form_on_emulation(children):
    """CRH_KERNEL_STREAM (csrc/pathtrace_stream.h): the three kernels and the host loop that feeds them, bit-identical to the rolling kernel and to the reference's fixtures
    for pool sizes from one cohort up; the 4-ary walk at SURVEY 8(c)'s gates on two BASELINE configs."""
    run_gpu_tier_on_emulation(children, "stream")


def test_node_programs_volumes_and_the_workgroup_kernel_on_emulation(children):
    """The rare-features instantiations (node programs; volumes: sampler draws inside the walk) and, on the volumes fixture, k_pathtrace_wg
    with its LDS lock."""
    run_gpu_tier_on_emulation(children, "rare_and_wg")


def test_gpu_bvh_builder_on_emulation(children):
    """csrc/bvh_build.hip on the shim (level-synchronous binning with LDS / global 64-bit atomics, the SAH sweeps as DPP row scans, one
    wave per small subtree): the reference's tree — node numbering, bounds bit patterns, primitive order — for the six scene fixtures and
    the degenerate inputs. (The 524 288-triangle stand-in and the 1 M soup pass too, in three minutes: run the GPU tier's file with
    CRH_LIB set to see it.)"""
    run_gpu_tier_on_emulation(children, "bvh")


def test_bvh_builder_fuzz_on_emulation(children):
    """tools/emu_fuzz_bvh.py, 19 seeded meshes that are awkward for a parallel builder (grid-aligned coordinates: every tie rule decides;
    duplicates; clusters with outliers; flat and needle extents; sizes around the phase boundaries): the reference's tree, or — where the
    reference's own node array overflows — a refusal."""
    rc, text = children("fuzz_bvh")
    assert rc == 0, text[-4000:]
    assert text.count('"ok": true') == 19 and '"refused": true' in text, text[-4000:]


def test_bvh_builder_top_level_bin_rows_on_emulation(children):
    """Levels of few, huge nodes (the top of a 10 M-triangle mesh) do not flush their chunks' bins into one node's words with global atomics: every chunk stores a row and
    k_fold_bins folds the columns (bvh_build.hip, round 4). The GPU tier reaches that path with the 524 288-triangle stand-in and the soups; here the threshold is lowered
    so that 13 seeded fuzz meshes take it at every level of at most 16 nodes: the reference's tree."""
    rc, text = children("fuzz_bvh_fold")
    assert rc == 0, text[-4000:]
    assert text.count('"ok": true') == 13, text[-4000:]


def test_one_unit_at_a_time_kernel_on_emulation(children):
    """The DEFAULT kernel form since the end of round 3 is k_pathtrace_roll (csrc/pathtrace_roll.h: a ring of open jobs per wave) — every other test of
    this module that does not say otherwise runs it. This one renders with k_pathtrace, the one-unit-at-a-time form (CRH_KERNEL_WAVE, the default until then):
    the same frames, bit for bit, and the same ray counts — with the default units and with tiny ones."""
    import json
    rc, text = children("roll")
    assert rc == 0, text[-4000:]
    got = [json.loads(l) for l in text.splitlines() if l.startswith("{")]
    assert [g["name"] for g in got] == ROLL_FIXTURES
    assert all(g["equal"] for g in got), got


# The scheduler's step counters on a 2-CU device (HIPEMU_CUS=2 fixes the work plan): node / triangle / control / retire+refill / generate /
# shade steps, lanes served by node and shade steps, scheduling rounds. A unit's schedule depends only on the unit, so these are exact and
# the same on every machine and thread count — and on the GPU (cfg2 at 64 spp, all counters equal). They change only
# when the scheduler, the work plan or a walk's step sequence changes: whoever does that on purpose updates them here (and looks at the
# lanes per step they imply); anything else that moves them is a performance regression caught without a GPU.
PINNED_STEPS = {
    "fence": {"w_node": 75197, "u_node": 2334213, "w_tri": 8527, "w_ctrl": 5469, "n_swap": 28818, "n_gen": 1501, "w_shade": 6221, "u_shade": 297282, "w_round": 67183},
    "cfg1_scene": {"w_node": 305689, "u_node": 11322917, "w_tri": 13460, "w_ctrl": 13431, "n_swap": 44741, "n_gen": 4941, "w_shade": 15615, "u_shade": 932071, "w_round": 133325},       # (round 3: without the shade-class batches, which are off by default now: 16125 -> 15615 shade steps)
}


def test_scheduler_step_counts_are_pinned(children):
    import json
    rc, text = children("steps")
    assert rc == 0, text[-4000:]
    got = {j["name"]: {k: v for k, v in j.items() if k != "name"} for j in (json.loads(l) for l in text.splitlines() if l.startswith("{"))}
    assert got == PINNED_STEPS, got


def test_schedule_fuzz_on_emulation(children):
    """tools/emu_fuzz.py, nine seeded cases (three per kernel form): random work plans, scheduler parameters, kernel forms (wave / workgroup / rolling units),
    device sizes, tile covers and pass splits — every one must give the reference's frame bit for bit and its ray count."""
    rc, text = children("fuzz")
    assert rc == 0, text[-4000:]
    assert text.count('"ok": true') == 9, text[-4000:]


def test_streaming_form_fuzz_on_emulation(children):
    """tools/emu_fuzz.py with the streaming form (CRH_KERNEL_STREAM, round 6): four seeded cases — random pool sizes (a few cohorts: many iterations), device sizes, walk
    scheduler parameters, tile covers (rectangles narrower than the 8 x 8 pixel order's blocks) and pass splits; the volumes fixture is served by the rolling kernel behind
    the same option. 72 cases ran when the form was built (profiles/r06k_emu_fuzz_stream.log)."""
    rc, text = children("fuzz_stream")
    assert rc == 0, text[-4000:]
    assert text.count('"ok": true') == 4, text[-4000:]


def test_split_pixel_fuzz_on_emulation(children):
    """tools/emu_fuzz_split.py, eight seeded cases with 128..330 passes per pixel: the rolling kernel ends its work queue with pass segments of single pixels
    (CRH_OPT_TAIL_SPLIT), staged per pixel and folded behind the kernel — random device sizes, split units per wave, unit sizes, tile covers and pass ranges give
    the one-unit-at-a-time kernel's frame bit for bit and its ray count."""
    rc, text = children("fuzz_split")
    assert rc == 0, text[-4000:]
    assert text.count('"ok": true') == 8, text[-4000:]


def test_adversarial_rays_on_emulation(children):
    """tools/emu_fuzz_rays.py, six seeds x 20 000 rays built for the corner cases of the slab and triangle tests (origins on vertices and grid
    points, direction components 0 / -0 / denormal / 1e9): EVERY ray — the ones whose slabs the reference's arithmetic turns into NaN included — gets the
    oracle's record bit for bit and its node / triangle test counts (crh_trace_rays follows the reference literally since round 3)."""
    import json
    rc, text = children("fuzz_rays")
    assert rc == 0, text[-4000:]
    got = [json.loads(l) for l in text.splitlines() if l.startswith("{")]
    assert len(got) == 6 and all(g["ok"] and g["regular_rays_that_differ"] == 0 and g["more_node_tests"] == 0 and g["fewer_node_tests"] == 0 for g in got), got
    assert sum(g["degenerate_rays_that_differ"] for g in got) == 0 and sum(g["degenerate_rays"] for g in got) > 30000, got


def test_product_entry_points_refuse_the_emulation(emu_lib):
    """api.py binds to the emulation library only for a caller that says CRH_ALLOW_EMULATION=1; bench.py and smoke() refuse it even then."""
    env = dict(os.environ, CRH_LIB=emu_lib)
    env.pop("CRH_ALLOW_EMULATION", None)
    code = "import sys; sys.path.insert(0, %r); from __graft_entry__ import load_package; load_package().api.library()" % REPO
    r = subprocess.run([sys.executable, "-c", code], env=env, cwd=REPO, capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and "CPU emulation" in r.stderr, r.stderr[-1500:]
    env["CRH_ALLOW_EMULATION"] = "1"
    r = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--steps", "1", "--warmup", "0"], env=env, cwd=REPO, capture_output=True, text=True, timeout=600)
    assert r.returncode != 0 and "CPU emulation" in (r.stderr + r.stdout), (r.stdout + r.stderr)[-1500:]
    r = subprocess.run([sys.executable, "-c", "import sys; sys.path.insert(0, %r); import __graft_entry__ as g; g.smoke()" % REPO], env=env, cwd=REPO,
                       capture_output=True, text=True, timeout=600)
    assert r.returncode != 0 and "CPU emulation" in r.stderr, r.stderr[-1500:]


def test_dropin_program_and_cluster_worker_on_emulation(children):
    """SURVEY 8(b) on the CPU: c-ray-hip — the reference's main.c, loaders, encoders and tile / UI code with ONE file replaced — renders
    config 1 through renderFrame() -> flatten -> C-ABI -> kernels (emulated) -> BMP encoder, normal and --iterative, and its `--worker` serves
    the reference's wire protocol to a test master: the reference's frame bit for bit, its tiles byte for byte. Needs the drop-in program and
    the asset overlay (built where /root/reference exists)."""
    if not (os.path.exists(os.path.join(REPO, "c-ray_amd", "_lib", "c-ray-hip")) and os.path.exists(os.path.join(REPO, "oracle", "_ref", "input", "scene.json"))):
        pytest.skip("c-ray-hip or the asset overlay is not built (needs /root/reference at build time)")
    run_gpu_tier_on_emulation(children, "dropin")


def test_dropin_program_on_several_emulated_gpus(emu_lib, manifest, golden_ref, tmp_path):
    """renderer_hip.c with more than one GPU, which no single-GPU box can run: HIPEMU_DEVICES emulated devices (2, 5, 8), one dispatch thread each,
    4-row strips dealt to them (host/share.h), the strips gathered onto GPU 0 through the RCCL entry points (grouped ncclSend / ncclRecv; a stand-in
    library: tests/emu/fake_rccl.c, which checks the call pattern), the 8-bit frame converted on every "device" and assembled from the strips — and,
    with --iterative (3 and 8 devices), the per-dispatch conversion and gather. The frame is the reference's, bit for bit, and so is the BMP, whatever the
    number of GPUs; the one-ncclReduce form gives the same frame; a GPU that cannot be set up or dies mid-frame has its strips re-dealt (round 4)."""
    import hashlib
    import json
    import numpy as np
    exe = os.path.join(REPO, "c-ray_amd", "_lib", "c-ray-hip")
    overlay = os.path.join(REPO, "oracle", "_ref", "input")
    if not (os.path.exists(exe) and os.path.exists(os.path.join(overlay, "scene.json"))):
        pytest.skip("c-ray-hip or the asset overlay is not built (needs /root/reference at build time)")
    sys.path.insert(0, os.path.join(REPO, "tools"))
    import refrun
    libdir = os.path.join(EMU_DIR, "_dropin_libs")
    for mode, key, gpu_counts in (((), "cfg1_scene", (2, 5, 8)), (("--iterative",), "cfg1_scene_iterative", (3, 8))):
        m = manifest[key]
        w, h = manifest["cfg1_scene"]["width"], manifest["cfg1_scene"]["height"]
        scene = refrun.rewrite_scene("scene.json", w, h, m["samples"], m["bounces"], out_dir=str(tmp_path))
        for gpus in gpu_counts:
            dump = str(tmp_path / f"hip_{key}_{gpus}.f32")
            env = dict(os.environ, CRH_DUMP_F32=dump, CRAY_HIP_DEVICES=str(gpus), HIPEMU_DEVICES=str(gpus), HIPEMU_CUS="2", HIPEMU_THREADS="4", CRH_TRACE_UPLOAD="1",
                       LD_LIBRARY_PATH=libdir + ":" + os.environ.get("LD_LIBRARY_PATH", ""))
            proc = subprocess.run([exe, *mode], input=json.dumps(scene).encode(), cwd=overlay, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=900)
            out = proc.stdout.decode(errors="replace")
            assert proc.returncode == 0 and f"on {gpus} GPUs" in out, out[-2000:]
            # ONE layout compile for the frame's GPUs (round 5: every GPU thread used to compile the scene for itself), one copy per GPU
            assert out.count("crh_scene_compile trace:") == 1 and out.count("crh_scene_upload_compiled trace:") == gpus and "crh_scene_upload trace: layout compile" not in out, out[-3000:]
            img = np.fromfile(dump, dtype=np.float32).reshape(h, w, 3)
            assert np.array_equal(img.view(np.uint32), golden_ref(key).view(np.uint32)), (key, gpus)
            bmp = [f for f in os.listdir(tmp_path) if f.endswith(".bmp")]
            assert bmp and hashlib.md5(open(tmp_path / bmp[0], "rb").read()).hexdigest() == m["bmp_md5"], (key, gpus, "the 8-bit frame (converted per GPU, strips gathered) differs from the reference's BMP")
            os.remove(tmp_path / bmp[0])
    # the one-ncclReduce form of the frame assembly (CRH_FRAMES=reduce) gives the same frame as the default strip gather
    m = manifest["cfg1_scene"]
    scene = refrun.rewrite_scene("scene.json", w, h, m["samples"], m["bounces"], out_dir=str(tmp_path))
    dump = str(tmp_path / "hip_reduce.f32")
    env = dict(os.environ, CRH_DUMP_F32=dump, CRAY_HIP_DEVICES="4", HIPEMU_DEVICES="4", HIPEMU_CUS="2", HIPEMU_THREADS="4", CRH_FRAMES="reduce",
               LD_LIBRARY_PATH=libdir + ":" + os.environ.get("LD_LIBRARY_PATH", ""))
    proc = subprocess.run([exe], input=json.dumps(scene).encode(), cwd=overlay, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=900)
    assert proc.returncode == 0, proc.stdout.decode(errors="replace")[-2000:]
    assert np.array_equal(np.fromfile(dump, dtype=np.float32).reshape(h, w, 3).view(np.uint32), golden_ref("cfg1_scene").view(np.uint32))
    # a GPU that cannot be set up (every allocation on device 2 fails), and a GPU that dies in the middle of the frame (its third dispatch's launch fails; one pass
    # per dispatch): the program names it, deals its strips to the GPUs that are left (they render them from pass 0), and the frame is still the reference's — and
    # so are the ray count and the BMP. (Only with no GPU left does renderFrame() end with logr(error): not reachable here — the BVH builder needs device 0 first.)
    env.pop("CRH_FRAMES")
    for hook in ({"HIPEMU_FAIL_DEVICE": "2"}, {"HIPEMU_FAIL_LAUNCH": "1:6", "CRH_DROPIN_PASSES": "1"}):
        os.remove(dump)
        proc = subprocess.run([exe], input=json.dumps(scene).encode(), cwd=overlay, env=dict(env, **hook), stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=900)
        out = proc.stdout.decode(errors="replace")
        gpu = "GPU 2" if "HIPEMU_FAIL_DEVICE" in hook else "GPU 1"
        assert proc.returncode == 0 and gpu in out and "dispatch thread failed" in out and "re-dealing" in out, out[-2000:]
        assert np.array_equal(np.fromfile(dump, dtype=np.float32).reshape(h, w, 3).view(np.uint32), golden_ref("cfg1_scene").view(np.uint32)), hook
        assert f"{manifest['cfg1_scene']['rays']} rays traced" in out, out[-600:]
        bmp = [f for f in os.listdir(tmp_path) if f.endswith(".bmp")]
        assert bmp and hashlib.md5(open(tmp_path / bmp[0], "rb").read()).hexdigest() == m["bmp_md5"], hook
        os.remove(tmp_path / bmp[0])

#!/usr/bin/env python3
"""wide_walk_study.py — dev / evidence (round 5): the 4-ary walk (CRH_OPT_WALK = CRH_WALK_WIDE4) against the binary walk (the contract) in the
LANE emulation (tests/emu/libcray_emu.so: pt_device.h compiled for the host), before any GPU time: per scene the frame of both walks (how many floats differ), the
box tests and walk steps per ray, the deepest stack, and — through emu_trace_rays on camera rays, on rays leaving surfaces and on adversarial rays through the scene's
own vertices and edge midpoints — how many rays change their closest-hit record.

    python tools/wide_walk_study.py [scene[:W:H:SPP:BOUNCES] ...]        default: the fixtures + BASELINE configs 2-5 at reduced size
"""
import ctypes as C, gzip, json, os, sys, tempfile, time
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "oracle")); sys.path.insert(0, os.path.join(REPO, "tests"))
import subprocess
subprocess.check_call(["make", "-s", "-C", os.path.join(REPO, "oracle"), "oracle"])
subprocess.check_call(["make", "-s", "-C", os.path.join(REPO, "tests", "emu"), "libcray_emu.so"])
import numpy as np
import oracle_py
abi = oracle_py.abi
L = C.CDLL(os.path.join(REPO, "tests", "emu", "libcray_emu.so"))
L.emu_render_region.argtypes = [C.POINTER(abi.SceneDesc), C.POINTER(abi.RenderParams), C.c_void_p, C.POINTER(abi.Counters), C.POINTER(C.c_uint32), C.c_int, C.c_int, C.c_int]
L.emu_trace_rays.argtypes = [C.POINTER(abi.SceneDesc), C.c_void_p, C.c_uint64, C.c_void_p]
L.emu_last_error.restype = C.c_char_p
L.emu_stack_high.restype = C.c_uint32

FIXTURES = ["cfg1_scene", "alphanode", "fence", "glowmetal", "refraction", "uvsphere", "nodezoo", "nodezoo_display"]
BIG = {"cfg2_hdr": (320, 180, 8, 8), "cfg3_venus": (320, 180, 4, 32), "cfg4_statues": (320, 180, 2, 30), "soup_1m": (320, 180, 2, 8)}
manifest = json.load(open(os.path.join(REPO, "tests", "golden", "manifest.json")))

def load(name):
    built = os.path.join(REPO, "scenes", "_built", name + ".blob")
    if os.path.exists(built):
        return oracle_py.OracleScene(built)
    with tempfile.NamedTemporaryFile(suffix=".blob", delete=False) as f:
        f.write(gzip.open(os.path.join(REPO, "tests", "golden", name + ".blob.gz")).read())
    return oracle_py.OracleScene(f.name)

def render(scene, w, h, s, b, wide):
    L.emu_set_walk(1 if wide else 0)
    fb = np.zeros((h, w, 3), np.float32)
    cnt, hi = abi.Counters(), C.c_uint32()
    p = abi.RenderParams(0, 0, w, h, w, h, 0, s, s, b)
    t0 = time.time()
    rc = L.emu_render_region(scene.ptr, C.byref(p), fb.ctypes.data, C.byref(cnt), C.byref(hi), 8, 8, 64)
    L.emu_set_walk(0)
    assert rc == 0, L.emu_last_error()
    return fb, cnt.as_dict(), hi.value, time.time() - t0

def trace(scene, rays, wide):
    L.emu_set_walk(1 if wide else 0)
    L.emu_stack_high(1)
    he = np.zeros(len(rays), dtype=abi.HIT_DTYPE)
    rc = L.emu_trace_rays(scene.ptr, rays.ctypes.data, len(rays), he.ctypes.data)
    L.emu_set_walk(0)
    assert rc == 0, L.emu_last_error()
    return he, L.emu_stack_high(1)

def rec_differs(a, b):
    d = np.zeros(len(a), bool)
    for f in ("inst", "poly", "material"):
        d |= a[f] != b[f]
    for f in ("distance", "point", "normal", "uv"):
        d |= (np.ascontiguousarray(a[f]).view(np.uint32) != np.ascontiguousarray(b[f]).view(np.uint32)).reshape(len(a), -1).any(axis=1)
    return d

def study_rays(scene, name, n, seed):
    """camera rays; rays that leave the first hits in random directions (what a path's later bounces are); adversarial: from the camera and from random points
    exactly through vertices and edge midpoints of the scene's meshes (object space = world space for identity instances; elsewhere they are just more rays)."""
    from conftest import camera_rays
    rng = np.random.default_rng(seed)
    d = scene.desc
    out = {}
    rays = camera_rays(d, n, seed)
    hb, _ = trace(scene, rays, False)
    hw, high = trace(scene, rays, True)
    out["camera"] = (n, int(rec_differs(hb, hw).sum()), int((hb["inst"] >= 0).sum()), int(hb["node_tests"].sum()), int(hw["node_tests"].sum()), int(hb["tri_tests"].sum()), int(hw["tri_tests"].sum()), high)
    hit = hb["inst"] >= 0
    if hit.sum() > 10:
        pts = hb["point"][hit]
        k = rng.integers(0, len(pts), n)
        dirs = rng.normal(size=(n, 3)).astype(np.float32)
        dirs /= np.linalg.norm(dirs, axis=1, keepdims=True)
        r2 = np.concatenate([pts[k] + 1e-4 * dirs, dirs], axis=1).astype(np.float32)
        hb2, _ = trace(scene, r2, False)
        hw2, high2 = trace(scene, r2, True)
        out["bounce"] = (n, int(rec_differs(hb2, hw2).sum()), int((hb2["inst"] >= 0).sum()), int(hb2["node_tests"].sum()), int(hw2["node_tests"].sum()), int(hb2["tri_tests"].sum()), int(hw2["tri_tests"].sum()), high2)
    if int(d.vertex_count) > 3 and int(d.poly_count) > 0:
        verts = np.ctypeslib.as_array(d.vertices, shape=(int(d.vertex_count), 3)).astype(np.float32)
        polys = np.frombuffer(C.string_at(d.polys, int(d.poly_count) * C.sizeof(abi.Poly)), dtype=np.int32).reshape(int(d.poly_count), -1)
        pk = rng.integers(0, len(polys), n)
        v = polys[pk, 0:3]                      # crh_poly: v[3] first
        ok = (v >= 0).all(axis=1) & (v < len(verts)).all(axis=1)
        v = v[ok]
        which = rng.integers(0, 3, len(v))
        a = verts[v[np.arange(len(v)), which]]
        bvert = verts[v[np.arange(len(v)), (which + 1) % 3]]
        mode = rng.integers(0, 3, len(v))
        target = np.where((mode == 0)[:, None], a, np.where((mode == 1)[:, None], (a + bvert) * np.float32(0.5), a + (bvert - a) * rng.random((len(v), 1)).astype(np.float32)))
        cam = np.array(list(d.camera.A), dtype=np.float64).reshape(3, 4)[:, 3].astype(np.float32)
        org = np.where((rng.random(len(v)) < 0.5)[:, None], cam[None, :], target + rng.normal(size=(len(v), 3)).astype(np.float32) * 3)
        r3 = np.concatenate([org, target - org], axis=1).astype(np.float32)
        r3 = r3[np.abs(r3[:, 3:6]).sum(axis=1) > 0]
        hb3, _ = trace(scene, r3, False)
        hw3, high3 = trace(scene, r3, True)
        out["through_vertices_and_edges"] = (len(r3), int(rec_differs(hb3, hw3).sum()), int((hb3["inst"] >= 0).sum()), int(hb3["node_tests"].sum()), int(hw3["node_tests"].sum()), int(hb3["tri_tests"].sum()), int(hw3["tri_tests"].sum()), high3)
    return out

args = sys.argv[1:]
cases = []
if not args:
    for f in FIXTURES:
        m = manifest[f]; cases.append((f, m["width"], m["height"], m["samples"], m["bounces"]))
    for k, v in BIG.items(): cases.append((k,) + v)
else:
    for a in args:
        p = a.split(":")
        if len(p) == 5: cases.append((p[0], int(p[1]), int(p[2]), int(p[3]), int(p[4])))
        elif p[0] in BIG: cases.append((p[0],) + BIG[p[0]])
        else:
            m = manifest[p[0]]; cases.append((p[0], m["width"], m["height"], m["samples"], m["bounces"]))
NR = int(os.environ.get("STUDY_RAYS", "200000"))
for name, w, h, s, b in cases:
    scene = load(name)
    if (name in BIG) or os.path.exists(os.path.join(REPO, "scenes", "_built", name + ".blob")):
        cam = scene.desc.camera; cam.width, cam.height = w, h
    fb0, c0, hi0, t0 = render(scene, w, h, s, b, False)
    fb1, c1, hi1, t1 = render(scene, w, h, s, b, True)
    differ = int((fb0.view(np.uint32) != fb1.view(np.uint32)).sum())
    pix = int((fb0.view(np.uint32) != fb1.view(np.uint32)).any(axis=2).sum())
    rays = c0["rays"]
    print(json.dumps({"scene": name, "frame": f"{w}x{h} {s} spp {b} bounces", "floats_that_differ": differ, "pixels_that_differ": pix, "rays": [c0["rays"], c1["rays"]],
                      "binary": {"box_tests_per_ray": round(c0["node_tests"] / rays, 2), "node_steps_per_ray": round(c0["node_tests"] / 2 / rays, 2), "tri_tests_per_ray": round(c0["tri_tests"] / rays, 2), "stack_high": hi0},
                      "wide4": {"box_tests_per_ray": round(c1["node_tests"] / c1["rays"], 2), "node_steps_per_ray": round(c1["node_tests"] / 4 / c1["rays"], 2), "tri_tests_per_ray": round(c1["tri_tests"] / c1["rays"], 2), "stack_high": hi1},
                      "steps_ratio": round((c0["node_tests"] / 2 / rays) / max(c1["node_tests"] / 4 / c1["rays"], 1e-9), 3)}), flush=True)
    for kind, (n, nd, hits, nb, nw, tb, tw, high) in study_rays(scene, name, NR, 7).items():
        print(json.dumps({"scene": name, "rays": kind, "n": n, "hits": hits, "records_that_differ": nd, "steps_ratio": round((nb / 2) / max(nw / 4, 1), 3), "tri_tests_ratio_wide_over_binary": round(tw / max(tb, 1), 3), "wide_stack_high": high}), flush=True)

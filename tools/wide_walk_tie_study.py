#!/usr/bin/env python3
"""wide_walk_tie_study.py — dev (round 5): of the rays whose closest-hit record differs between the binary and the 4-ary walk, how many saw an EXACT tie? Needs the lane emulation built with
profiles/r05c_exp_tie_study.patch as /tmp/libcray_emu_tie.so (g++ ... -DCRH_TIE_STUDY tests/emu/emu.cpp c-ray_amd/csrc/scene_compile.cpp). Result: profiles/r05c_wide_walk_tie_study.log."""
import sys, os
__file__=os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools", "wide_walk_study.py")
src=open(__file__).read().split("args = sys.argv[1:]")[0].replace('L = C.CDLL(os.path.join(REPO, "tests", "emu", "libcray_emu.so"))','L = C.CDLL("/tmp/libcray_emu_tie.so")')
exec(src)
from conftest import camera_rays
def run(name, n, seed, kind):
    scene=load(name)
    d=scene.desc
    rng=np.random.default_rng(seed)
    rays=camera_rays(d, n, seed)
    if kind=="bounce":
        hb,_=trace(scene, rays, False)
        hit=hb["inst"]>=0
        pts=hb["point"][hit]
        k=rng.integers(0,len(pts),n)
        dirs=rng.normal(size=(n,3)).astype(np.float32); dirs/=np.linalg.norm(dirs,axis=1,keepdims=True)
        rays=np.concatenate([pts[k]+1e-4*dirs, dirs],axis=1).astype(np.float32)
    elif kind=="edges":
        verts = np.ctypeslib.as_array(d.vertices, shape=(int(d.vertex_count), 3)).astype(np.float32)
        polys = np.frombuffer(C.string_at(d.polys, int(d.poly_count) * C.sizeof(abi.Poly)), dtype=np.int32).reshape(int(d.poly_count), -1)
        pk = rng.integers(0, len(polys), n)
        v = polys[pk, 0:3]
        which = rng.integers(0, 3, len(v))
        a = verts[v[np.arange(len(v)), which]]; b = verts[v[np.arange(len(v)), (which + 1) % 3]]
        mode = rng.integers(0, 3, len(v))
        target = np.where((mode == 0)[:, None], a, np.where((mode == 1)[:, None], (a + b) * np.float32(0.5), a + (b - a) * rng.random((len(v), 1)).astype(np.float32)))
        cam = np.array(list(d.camera.A), dtype=np.float64).reshape(3, 4)[:, 3].astype(np.float32)
        org = np.where((rng.random(len(v)) < 0.5)[:, None], cam[None, :], target + rng.normal(size=(len(v), 3)).astype(np.float32) * 3)
        rays = np.concatenate([org, target - org], axis=1).astype(np.float32)
    hb,_=trace(scene, rays, False); hw,_=trace(scene, rays, True)
    tb=(hb["tri_tests"]>>31)!=0; tw=(hw["tri_tests"]>>31)!=0
    hb["tri_tests"]&=0x7fffffff; hw["tri_tests"]&=0x7fffffff
    d_=rec_differs(hb,hw)
    print(name, kind, "rays", len(rays), "differ", int(d_.sum()), "of which the WIDE walk saw an exact tie", int((d_&tw).sum()), "binary saw a tie", int((d_&tb).sum()),
          "neither", int((d_&~tw&~tb).sum()), "| walks with a tie: wide", int(tw.sum()), "binary", int(tb.sum()), flush=True)
for a in sys.argv[1:]:
    name,kind,n=a.split(":")
    run(name,int(n),11,kind)

#!/usr/bin/env python3
"""probe_exact.py — dev probe: how many floats of the GPU frame differ from the real reference's frame, per fixture (0 = bit-exact)."""
import gzip, json, os, sys, tempfile
import numpy as np
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from __graft_entry__ import load_package, BUILT
pkg = load_package(); api = pkg.api
G = os.path.join(REPO, "tests", "golden")
man = json.load(open(os.path.join(G, "manifest.json")))
ctx = api.Context(0)
out = {}
for name, m in sorted(man.items()):
    if "iterative" in name:
        continue
    ref = np.frombuffer(gzip.open(os.path.join(G, name + ".ref.f32.gz")).read(), np.float32).reshape(m["height"], m["width"], 3)
    if "built_blob" in m:
        path = os.path.join(BUILT, m["built_blob"] + ".blob")
        scene = api.Scene(path)
        scene.desc.camera.width, scene.desc.camera.height = m["width"], m["height"]
    else:
        tmp = tempfile.NamedTemporaryFile(suffix=".blob", delete=False)
        tmp.write(gzip.open(os.path.join(G, m.get("blob", name) + ".blob.gz")).read()); tmp.close()
        scene = api.Scene(tmp.name)
    ctx.upload(scene)
    fb = ctx.framebuffer(m["width"], m["height"])
    ctx.reset_counters()
    ctx.render_region(fb, m["width"], m["height"], m["samples"], m["bounces"])
    img = ctx.download(fb, m["width"], m["height"])
    cnt = ctx.counters()
    nd = int((img.view(np.uint32) != ref.view(np.uint32)).sum())
    out[name] = {"floats_differ": nd, "pixels_differ": int((img != ref).any(axis=2).sum()), "rays": cnt["rays"], "ref_rays": m.get("rays")}
    print(name, out[name], flush=True)
os.makedirs(os.path.join(REPO, "gpurun_out"), exist_ok=True)
json.dump(out, open(os.path.join(REPO, "gpurun_out", "probe_exact.json"), "w"), indent=1)

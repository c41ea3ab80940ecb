#!/usr/bin/env python3
"""sched_counts.py — child process of tests/test_kernel_emu.py: the wave scheduler's step counters (counter level 2) for golden fixtures
on the kernel emulation (CRH_LIB); one JSON line per fixture. TEST INFRASTRUCTURE."""
import gzip, json, os, sys, tempfile
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO)
from __graft_entry__ import load_package
pkg = load_package(); api, abi = pkg.api, pkg.abi
man = json.load(open(os.path.join(REPO, "tests", "golden", "manifest.json")))
for name in sys.argv[1:]:
    m = man[name]
    with tempfile.NamedTemporaryFile(suffix=".blob") as f:
        f.write(gzip.open(os.path.join(REPO, "tests", "golden", name + ".blob.gz")).read()); f.flush()
        scene = api.Scene(f.name)
    ctx = api.Context(0)
    ctx.set_option(abi.OPT_COUNTER_LEVEL, 2)
    ctx.set_option(abi.OPT_KERNEL, abi.KERNEL_WAVE)          # the pins are the one-unit-at-a-time form's (tests/test_kernel_emu.py)
    ctx.upload(scene)
    fb = ctx.framebuffer(m["width"], m["height"])
    ctx.reset_counters()
    ctx.render_region(fb, m["width"], m["height"], m["samples"], m["bounces"]); ctx.synchronize()
    t = ctx.phase_ticks()
    print(json.dumps({"name": name, **{k: t[k] for k in ("w_node", "u_node", "w_tri", "w_ctrl", "n_swap", "n_gen", "w_shade", "u_shade", "w_round")}}), flush=True)
    ctx.close()

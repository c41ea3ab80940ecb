#!/usr/bin/env python3
"""render_once.py — dev: one dispatch through the C-ABI (no torch), for rocprofv3 --pmc passes on a chosen kernel form / option set.
    [CRH_LIB=<variant .so>] python tools/render_once.py SCENE W H SPP BOUNCES [name=value ...]
    options: kernel (CRH_OPT_KERNEL), unit_items, units_per_wave, pass_chunk, tail, blocks_per_cu, walk (CRH_OPT_WALK: 1 = the 4-ary walk), and the scheduler's
             node tri ctrl swap_min fill_to run_num tri_in_run ctrl_in_run shade_min        (a bare number = kernel, as before)"""
import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from __graft_entry__ import load_package, BUILT
pkg = load_package(); api = pkg.api; abi = pkg.abi
name, w, h, spp, b = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5])
opts = {}
for a in sys.argv[6:]:
    k, _, v = a.partition("=")
    opts.update({k: int(v)} if v else {"kernel": int(k)})
ctx = api.Context(0)
ctx.set_option(abi.OPT_COUNTER_LEVEL, 1)
for key, opt in (("kernel", abi.OPT_KERNEL), ("unit_items", abi.OPT_UNIT_ITEMS), ("units_per_wave", abi.OPT_UNITS_PER_WAVE), ("pass_chunk", abi.OPT_PASS_CHUNK),
                 ("tail", abi.OPT_TAIL_PERCENT), ("blocks_per_cu", abi.OPT_BLOCKS_PER_CU), ("walk", abi.OPT_WALK)):
    if key in opts:
        ctx.set_option(opt, opts[key])
sched = dict(node=70, tri=160, ctrl=120, swap_min=16, fill_to=160, run_num=4, tri_in_run=12, ctrl_in_run=12, shade_min=48)
if any(k in opts for k in sched):
    sched.update({k: v for k, v in opts.items() if k in sched})
    ctx.set_sched(**sched)
ctx.upload(api.Scene(name if os.path.isabs(name) else os.path.join(BUILT, name + ".blob")))        # a name under scenes/_built, or an absolute path
fb = ctx.framebuffer(w, h)
ctx.reset_counters()
ctx.render_region(fb, w, h, spp, b); ctx.synchronize()
print(name, w, h, spp, b, opts or "defaults", f"{ctx.kernel_time_ms()[0]:.2f} ms", ctx.counters()["rays"], "rays", flush=True)

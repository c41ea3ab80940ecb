#!/usr/bin/env python3
"""bench.py — Mray/s of the MI355X path-tracing hot path on BASELINE.json configs[1]
(input/hdr.json, 1280x720, 256 spp, 8 bounces; the venus mesh is the generated 524 288-triangle stand-in).

    python bench.py --gpus N --steps K --warmup W
    (N > 1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...)

One step = one full frame: every rank renders ITS share of the frame (N = 1: the whole frame as one region;
N > 1: every N-th 4-row strip, which balances the ranks where dealing out tiles does not — c-ray_amd/render.py) for
all 256 passes in one dispatch of the persistent kernel, then the float framebuffers are summed onto rank 0 with one
RCCL reduce (non-owned pixels are exactly 0, so the sum is a gather). The frame is a fixed job, so N > 1 is STRONG scaling.
value = rays (getClosestIsect calls, primary + secondary, counted by the kernel) of all ranks / wall time.

Extra objects on the JSON line (N = 1 only for cpu_baseline):
  roofline     HBM roofline of k_pathtrace: algorithmic bytes per launch (DESIGN.md §"Algorithmic bytes")
               / average launch duration measured with HIP events on the launch stream, vs 8 TB/s.
  cpu_baseline the reference's own pthread renderer (oracle/_ref/c-ray-ref, built from the unmodified
               reference sources) timed on this box's host cores on a reduced-spp sample of the same frame.
"""
import argparse
import json
import os
import sys
import time

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)

WORKLOADS = {   # BASELINE.json configs[1..4]; the default (and the headline) is cfg2
    "cfg2": {"scene": "hdr.json", "blob": "cfg2_hdr", "width": 1280, "height": 720, "samples": 256, "bounces": 8, "tile": (64, 64), "tile_order": 1,
             "what": "input/hdr.json {W}x{H}, {SPP} spp, {B} bounces (BASELINE.json configs[1]); venusscaled.obj = generated 524288-triangle stand-in "
                     "(tools/gen_assets.py), HDR env map + 2048^2 grid texture from the reference tree"},
    "cfg3": {"scene": "venus.json", "blob": "cfg3_venus", "width": 1920, "height": 1080, "samples": 1024, "bounces": 32, "tile": (64, 64), "tile_order": 1,
             "what": "input/venus.json {W}x{H}, {SPP} spp, {B} bounces (BASELINE.json configs[2]); venusscaled.obj = generated stand-in"},
    "cfg4": {"scene": "statues.json", "blob": "cfg4_statues", "width": 3840, "height": 2160, "samples": 2048, "bounces": 30, "tile": (64, 64), "tile_order": 1,
             "what": "input/statues.json {W}x{H}, {SPP} spp, {B} bounces (BASELINE.json configs[3]: the 1/2/4/8-GPU scaling scene); 55 instances of the stand-in statue"},
    "soup": {"scene": "soup_1000000.json", "blob": "soup_1m", "width": 2560, "height": 1440, "samples": 512, "bounces": 8, "tile": (64, 64), "tile_order": 1,
             "what": "synthetic 1 M-triangle soup {W}x{H}, {SPP} spp, {B} bounces (BASELINE.json configs[4] at 1 M triangles; the 10 M blob is built on the box by tools/make_soup.sh)"},
}
WORKLOADS["soup10m"] = {"scene": None, "blob": "soup_10m", "width": 2560, "height": 1440, "samples": 512, "bounces": 8, "tile": (64, 64), "tile_order": 1, "triangles": 10000000,
                        "what": "synthetic 10 M-triangle soup {W}x{H}, {SPP} spp, {B} bounces (BASELINE.json configs[4] at its real size; the 1.04 GB scene is built on this box by "
                                "tools/make_soup_blob.py: gen_soup's triangles as the reference's loader reads them + the GPU BVH builder, the reference's tree)"}
WORKLOAD = WORKLOADS["cfg2"]
# what `other_workloads` measures beside the headline: (workload, spp of the timed dispatch = BASELINE's own, spp of the counting dispatch, spp of a short dispatch kept for
# continuity with rounds 1-4). Round 6: configs[3] is timed ONCE at its own 2048 passes too (24 s of one GPU; the seed of a (pixel, pass) depends on the
# sample count, sampler.c:42, and so does the path mix) — its counters come from a 256-pass counting dispatch (bytes per ray carried over: stated in the object), and the
# 256-pass rate of round 5's line stays beside it (`at_256_spp`).
OTHER_WORKLOADS = (("cfg3", 1024, 1024, 32), ("cfg4", 2048, 256, 4), ("soup", 512, 512, 16), ("soup10m", 512, 512, 8))
WIDE4_WORKLOADS = ("cfg3", "cfg4", "soup")          # the 4-ary walk (CRH_OPT_WALK = WIDE4: an option, not bit-exact at near ties) timed beside the contract's walk: `wide4_mrays`


def workload_blob(key, built_dir):
    """Path of a workload's scene blob; the 10 M soup is built here on first use (tools/make_soup_blob.py, kept in the temp directory)."""
    wl = WORKLOADS[key]
    path = os.path.join(built_dir, wl["blob"] + ".blob")
    if os.path.exists(path) or not wl.get("triangles"):
        return path
    import tempfile
    path = os.path.join(tempfile.gettempdir(), f"crh_{wl['blob']}.blob")
    if not os.path.exists(path):
        sys.path.insert(0, os.path.join(REPO, "tools"))
        import make_soup_blob
        make_soup_blob.build(wl["triangles"], path + ".tmp", builder="gpu")
        os.replace(path + ".tmp", path)
    return path
HBM_PEAK_GBS = 8000.0     # MI355X_MICROARCH.md: HBM3E 8 TB/s


PATH_STATE_BYTES_PER_RAY = 152   # SURVEY.md §8(d) B_state: 2 x (28 B ray + 16 B weight + 12 B radiance + 16 B RNG + 4 B pixel)


def algorithmic_bytes(cnt, path_state=True):
    """Device-layout algorithmic bytes of one launch from its own counters (DESIGN.md §Algorithmic bytes).

    path_state adds SURVEY.md §8(d)'s B_state term: the kernel keeps every path in a per-wave table in memory (one
    record written by the shading step and read back by the walk / next shading step per bounce)."""
    mesh_hits = max(cnt["inst_hits"] - 0, 0)
    b = (32 * cnt["node_tests"]             # one 32-B node record per box test (child pairs are one 64-B load)
         + 48 * cnt["tri_tests"]            # prepared triangle v0,e1,e2,n
         + 64 * cnt["inst_visits"]          # enter line of the instance record (Ainv, kind, root, offset, radius)
         + 128 * mesh_hits                  # finishing a hit: second instance line (A, material) + the 64-B shading record
         + 16 * cnt["tex_fetches"]          # one texel (RGB float 12 B / RGBA8 4 B), rounded up
         + 32 * cnt["rays"]                 # material + bsdf records per shaded ray
         + 48 * cnt["paths"])               # sample staged (12 B w + 12 B r) + running mean RMW (24 B / pass chunk)
    if path_state:
        b += PATH_STATE_BYTES_PER_RAY * cnt["rays"]
    return b


def _elf64_sections(data):
    """{name: (type, addr, offset, size, link, entsize)} of a little-endian ELF64 image."""
    import struct
    shoff, = struct.unpack_from("<Q", data, 0x28)
    shentsize, shnum, shstrndx = struct.unpack_from("<HHH", data, 0x3A)
    raw = [struct.unpack_from("<IIQQQQIIQQ", data, shoff + i * shentsize) for i in range(shnum)]
    stroff = raw[shstrndx][4]
    out = {}
    for name, typ, _flags, addr, off, size, link, _info, _align, entsize in raw:
        end = data.index(b"\0", stroff + name)
        out[data[stroff + name:end].decode(errors="replace")] = (typ, addr, off, size, link, entsize)
    return out, raw


def _pathtrace_code_bytes(obj):
    """The machine code and the kernel descriptors of every k_pathtrace* kernel of one gfx950 code object (an ELF64): (symbol name, bytes) pairs, sorted.
    Nothing else of the object enters the fingerprint — notes, build ids, string tables, symbol order and the other kernels' code differ between two builds
    of the same source (a clean rebuild changed 359 bytes outside .text and with them the whole-object md5) or with edits elsewhere."""
    import struct
    secs, raw = _elf64_sections(obj)
    symtab = secs.get(".symtab")
    if not symtab:
        return None
    _typ, _addr, off, size, link, entsize = symtab
    stroff = raw[link][4]
    out = []
    for i in range(size // (entsize or 24)):
        name, _info, _other, shndx, value, ssize = struct.unpack_from("<IBBHQQ", obj, off + i * (entsize or 24))
        end = obj.index(b"\0", stroff + name)
        sym = obj[stroff + name:end]
        if b"k_pathtrace" not in sym or not ssize or shndx == 0 or shndx >= len(raw):
            continue
        _n, styp, _f, saddr, soff, ssz = raw[shndx][:6]
        if styp == 8:          # SHT_NOBITS
            continue
        start = soff + (value - saddr)
        out.append((sym, obj[start:start + ssize]))
    return sorted(out)


def kernel_source_md5():
    """Fingerprint of the DEVICE code of the path-tracing kernels in the built library: the .text bytes and the kernel descriptors of the k_pathtrace* symbols of
    the gfx950 code object of csrc/cray_hip.hip inside the .hip_fatbin section of libcray_hip.so. PMC figures in profiles/ are quoted only while they describe the
    kernels that are running: host-side edits, edits to the other kernels and a REBUILD of the same source do not change the fingerprint (round 3 hashed the whole
    code object, which a rebuild changed with identical machine code); any edit that changes the path-tracing kernels' instructions does."""
    import hashlib
    import struct
    path = os.path.join(REPO, "c-ray_amd", "_lib", "libcray_hip.so")
    with open(path, "rb") as f:
        data = f.read()
    if data[:4] != b"\x7fELF" or data[4] != 2:
        return hashlib.md5(data).hexdigest()
    secs, _ = _elf64_sections(data)
    if ".hip_fatbin" not in secs:
        return hashlib.md5(data).hexdigest()
    _t, _a, off, size = secs[".hip_fatbin"][:4]
    fat = data[off:off + size]
    magic, objects, pos = b"__CLANG_OFFLOAD_BUNDLE__", [], 0
    while True:                                    # one offload bundle per translation unit: {magic, n, n x (offset, size, triple)}
        p = fat.find(magic, pos)
        if p < 0:
            break
        n, = struct.unpack_from("<Q", fat, p + 24)
        q = p + 32
        for _ in range(n):
            o, sz, ts = struct.unpack_from("<QQQ", fat, q)
            triple = fat[q + 24:q + 24 + ts]
            q += 24 + ts
            obj = fat[p + o:p + o + sz]
            if b"gfx" in triple and b"k_pathtrace" in obj:
                objects.append(obj)
        pos = p + len(magic)
    if not objects:
        return hashlib.md5(fat).hexdigest()
    h = hashlib.md5()
    for obj in objects:
        code = _pathtrace_code_bytes(obj) if obj[:4] == b"\x7fELF" else None
        if not code:
            h.update(obj)
            continue
        for sym, blob in code:
            h.update(sym); h.update(blob)
    return h.hexdigest()


def calibration():
    """profiles/calibration.json (tools/calib.sh + tools/calib_table.py, round 5): what rocprofv3's FETCH_SIZE / WRITE_SIZE report for kernels of KNOWN traffic in this
    kernel's own access patterns, and what the VALU-pipe formula reads at KNOWN saturation. {} when absent."""
    try:
        return json.load(open(os.path.join(REPO, "profiles", "calibration.json")))
    except Exception:
        return {}


def calibrated_traffic(t):
    """Measured L2<->fabric bytes per launch from a profiles/hbm_traffic*.json record: the raw FETCH_SIZE bytes times the factor measured on divergent 64-byte gathers
    (the kernel's reads are child pairs, texels and record quarters: 64-byte requests, which FETCH_SIZE counts at face value — NOT the x2 of wide streaming reads,
    which are 128-byte requests tallied at 64) plus WRITE_SIZE (128-byte record writes: exact). Returns (bytes, upper bound as if every read request were 128 bytes)."""
    if not t or t.get("fetch_bytes_raw") is None:
        return None, None
    cal = calibration()
    rf = (cal.get("k_gather64") or {}).get("read_factor") or 1.0
    rf128 = (cal.get("k_gather128") or {}).get("read_factor") or 2.0
    wf = (cal.get("k_write128") or {}).get("write_factor") or 1.0
    return t["fetch_bytes_raw"] * rf + t["write_bytes"] * wf, t["fetch_bytes_raw"] * rf128 + t["write_bytes"] * wf


def measured_profile(workload_key):
    """(traffic bytes per launch, its upper bound, VALU roofline dict) from the committed rocprofv3 PMC summary — or Nones when that summary
    was taken from different kernel sources or another workload (stale numbers are not quoted)."""
    path = os.path.join(REPO, "profiles", "hbm_traffic.json" if workload_key == "cfg2" else f"hbm_traffic_{workload_key}.json")
    try:
        t = json.load(open(path))
    except Exception:
        return None, None, None
    if t.get("source_md5") != kernel_source_md5() or t.get("workload", "cfg2") != workload_key:
        return None, None, None
    lo, hi = calibrated_traffic(t)
    valu = t.get("valu")
    sat = (calibration().get("k_mix") or {}).get("pipe_busy_at_saturation")
    if valu and valu.get("pipe_busy") and sat:
        valu = dict(valu, bound="valu-issue" if valu["pipe_busy"] / sat >= 0.7 else "memory-latency + valu-issue", calibration={"formula_reads_at_known_saturation": {k: round((calibration().get(k) or {}).get("pipe_busy_at_saturation") or 0, 3) for k in ("k_fma", "k_add", "k_mix")},
                                       "of_saturated_rate": round(valu["pipe_busy"] / sat, 3),
                                       "note": "SQ_ACTIVE_INST_VALU counts one busy quad-cycle per vector instruction and SQ_WAVE_CYCLES quad-cycles per resident wave, so the formula reads "
                                               "vector instructions per SIMD quad-cycle — 1.6-1.7 for a saturating loop of the node step's blend (fma / min / max / cmp / cndmask), not 1.0: "
                                               "of_saturated_rate = pipe_busy / that reading is the share of the vector pipe's saturated issue rate the kernel uses (profiles/calibration.json)"})
    return lo, hi, valu


def fractions(alg_bytes, traffic_bytes, ms, traffic_upper=None):
    """The two roofline fractions of one launch, by name, so that neither is mistaken for the other: frac_algorithmic = the bytes a cache-less machine
    would move (scene records touched) / time / HBM peak — it EXCEEDS 1 wherever L2 and the 256 MB Infinity Cache serve most records, and says so;
    frac_measured_traffic = the measured L2<->fabric volume (rocprofv3 FETCH_SIZE and WRITE_SIZE with the factors calibrated on this kernel's access patterns:
    calibrated_traffic(); MALL hits included) / time / HBM peak, null when profiles/ holds no PMC measurement of this device code and workload;
    frac_measured_traffic_upper = the same if every read request had been a 128-byte one."""
    fa = alg_bytes / ms / 1e6 / HBM_PEAK_GBS
    out = {"frac_algorithmic": round(fa, 4), "frac_measured_traffic": round(traffic_bytes / ms / 1e6 / HBM_PEAK_GBS, 4) if traffic_bytes else None}
    if traffic_upper:
        out["frac_measured_traffic_upper"] = round(traffic_upper / ms / 1e6 / HBM_PEAK_GBS, 4)
    if fa > 1.0:
        out["frac_note"] = "algorithmic fraction above 1: the records were served by L2 / Infinity Cache, not by HBM (see frac_measured_traffic)"
    return out


def measure_other_workloads(api, abi, built_dir):
    """BASELINE.json configs[2..4] beside the headline, OUTSIDE the timed region, at BASELINE's own sample counts (OTHER_WORKLOADS): one counting dispatch and one timed
    dispatch of the full frame each (a dispatch of seconds needs no median), then three short dispatches at the reduced sample count of rounds 1-4 (`reduced`). Mray/s,
    algorithmic bytes and fraction of the HBM roofline per workload; `traffic` = the PMC measurement of THAT dispatch (same workload, same sample count, this device code:
    profiles/hbm_traffic_<workload>.json) — a record taken at another sample count is quoted per ray and says so (`traffic_from_spp`); `wide4_mrays` = the same frame with
    the 4-ary walk (CRH_OPT_WALK, an option)."""
    out = {}
    for key, spp, spp_count, spp_reduced in OTHER_WORKLOADS:
        wl = WORKLOADS[key]
        t0 = time.perf_counter()
        try:
            blob = workload_blob(key, built_dir)
            if not os.path.exists(blob):
                out[key] = {"skipped": f"{blob} not built"}
                continue
            scene = api.Scene(blob)
            ctx = api.Context(0)
            ctx.upload(scene)
            w, h, b = wl["width"], wl["height"], wl["bounces"]
            fb = ctx.framebuffer(w, h)

            def timed(n):
                ctx.set_option(abi.OPT_COUNTER_LEVEL, 1)
                ctx.reset_counters()
                ctx.clear(fb, w, h)
                ctx.render_region(fb, w, h, n, b)
                ctx.synchronize()
                return ctx.kernel_time_ms()[0], ctx.counters()["rays"]
            ctx.set_option(abi.OPT_COUNTER_LEVEL, 2)
            ctx.reset_counters()
            ctx.render_region(fb, w, h, spp_count, b)
            ctx.synchronize()
            full = ctx.counters()
            ms, rays = timed(spp)
            spent = ms
            extra = {}
            if spp_count == spp:
                assert rays == full["rays"], "the timed dispatch traced other rays than the counting one"
            else:          # the counters are the shorter dispatch's: quantities per ray carry over, and the shorter dispatch is timed too
                ms_c, rays_c = timed(spp_count)
                spent += ms_c
                assert rays_c == full["rays"], "the timed dispatch traced other rays than the counting one"
                extra[f"at_{spp_count}_spp"] = {"mrays": round(rays_c / ms_c / 1e3, 1), "kernel_ms": round(ms_c, 2), "rays": rays_c,
                                                "note": f"round 5's line quoted this dispatch; the counters of this object (tests per ray, bytes per ray) were taken at {spp_count} spp"}
            times = []
            for _ in range(3):
                m_, reduced_rays = timed(spp_reduced)
                times.append(m_)
            spent += sum(times)
            alg = algorithmic_bytes(full, path_state=False) * (rays / max(full["rays"], 1))
            traffic = upper = None
            traffic_note = {}
            try:
                t = json.load(open(os.path.join(REPO, "profiles", f"hbm_traffic_{key}.json")))
                if t.get("source_md5") == kernel_source_md5() and t.get("rays"):
                    lo, hi = calibrated_traffic(t)
                    if t.get("spp") == spp and t["rays"] == rays:
                        traffic, upper = lo, hi          # the PMC run WAS this dispatch
                    else:
                        traffic, upper = lo * rays / t["rays"], hi * rays / t["rays"]
                        traffic_note = {"traffic_from_spp": t.get("spp"), "traffic_note": "PMC run at another sample count: bytes per ray carried to this dispatch"}
            except Exception:
                pass
            wide = {}
            if key in WIDE4_WORKLOADS:
                try:
                    cw = api.Context(0)
                    cw.set_option(abi.OPT_WALK, abi.WALK_WIDE4)
                    cw.set_option(abi.OPT_COUNTER_LEVEL, 1)
                    cw.upload(scene)
                    fbw = cw.framebuffer(w, h)
                    n_w = spp_count if spp_count != spp else spp
                    cw.render_region(fbw, w, h, min(n_w, 8), b); cw.synchronize(); cw.reset_counters(); cw.clear(fbw, w, h)
                    cw.render_region(fbw, w, h, n_w, b)
                    cw.synchronize()
                    ms_w, rays_w = cw.kernel_time_ms()[0], cw.counters()["rays"]
                    spent += ms_w
                    base_ms, base_rays = (ms, rays) if n_w == spp else (ms_c, rays_c)
                    wide = {"wide4_mrays": round(rays_w / ms_w / 1e3, 1), "wide4": {"spp": n_w, "kernel_ms": round(ms_w, 2), "rays": rays_w, "kernel": cw.last_kernel_name(),
                                                                                     "vs_binary_walk": round((rays_w / ms_w) / (base_rays / base_ms), 3),
                                                                                     "note": "CRH_OPT_WALK = CRH_WALK_WIDE4 (an option; the binary walk is the bit-exact contract): same frame within "
                                                                                             "SURVEY 8(c)'s gates — tests/test_gpu_parity.py::test_wide_walk_meets_the_tolerance_gates — but not bit for bit at near ties"}}
                    cw.close()
                except Exception as e:
                    wide = {"wide4": {"failed": f"{type(e).__name__}: {e}"[:200]}}
            out[key] = {"workload": wl["what"].format(W=w, H=h, SPP=spp if spp == wl["samples"] else f"{spp} of {wl['samples']}", B=b), "spp": spp, "baseline_spp": wl["samples"],
                        "mrays": round(rays / ms / 1e3, 1), "kernel_ms": round(ms, 2),
                        "rays": rays, "rays_per_path": round(full["rays"] / max(full["paths"], 1), 2), "counters_from_spp": spp_count,
                        "node_tests_per_ray": round(full["node_tests"] / max(full["rays"], 1), 1), "tri_tests_per_ray": round(full["tri_tests"] / max(full["rays"], 1), 1),
                        "bytes_per_ray": round(alg / max(rays, 1), 1), "achieved_GBs": round(alg / ms / 1e6, 1),
                        **fractions(alg, traffic, ms, upper),
                        "traffic": traffic, **traffic_note, **extra, **wide,
                        "reduced": {"spp": spp_reduced, "mrays": round(reduced_rays / sorted(times)[1] / 1e3, 1), "kernel_ms": round(sorted(times)[1], 2), "note": "the sample count of rounds 1-4's lines (median of three dispatches)"},
                        "setup_s": round(time.perf_counter() - t0 - spent / 1e3, 2)}
            ctx.close()
            scene.close()
        except Exception as e:      # a missing blob / failed build must not take the headline line with it
            out[key] = {"failed": f"{type(e).__name__}: {e}"[:300]}
    return out


# the scaling objects of the bench line: configs[3] at 128 of its 2048 passes (1.6 s at N = 1: a 1/8 share is 0.2 s, not a drain test) and configs[4] —
# the 10 M-triangle soup, "sharded across 8 MI355X (RCCL framebuffer reduce)" — at 128 of its 512 (0.9 s at N = 1), through the same strips + gather as the headline
SCALING_SPP = {"cfg4": 128, "soup10m": 128}
# (say it in the line) the reduced sample counts do not move the N = 1 rate: round 5's driver line has 2 419 (128 spp, here) against 2 416 Mray/s (256 spp,
# other_workloads) for configs[3] and 1 999 (128 spp) against 1 997 (512 spp, BASELINE's own count) for the 10 M soup
SCALING_FULL_SPP_NOTE = {"cfg4": "; at N = 1 the rate at 128 spp equals the rate at 256 / 2048 spp (BENCH_r05: 2419 vs 2416 Mray/s; other_workloads.cfg4 of this line)",
                         "soup10m": "; at N = 1 the rate at 128 spp equals the rate at BASELINE's 512 spp (BENCH_r05: 1999 vs 1997 Mray/s; other_workloads.soup10m of this line)"}


def all_ranks_ok(ok, torch, dist, world, device):
    """Every rank learns whether ALL ranks came through their last local phase (one tiny all-reduce; nothing at world == 1). A rank that failed locally must not walk away
    from the collectives its peers are about to enter — and must not take the headline line with it (renderer.c:96-117: a worker that fails to start does not abort the frame)."""
    if world == 1:
        return bool(ok)
    t = torch.tensor([1.0 if ok else 0.0], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MIN)
    return bool(float(t[0]))


def scaling_workload(key, api, render, torch, dist, built_dir, local_rank, rank, world, reps=2, device=None):
    """A BASELINE config through the multi-GPU path — 4-row strips per rank, the owned strips gathered on rank 0 — at a stated sample count, emitted at every N (N = 1
    included) so that the points of the driver's scaling run divide. Outside the timed region. The 10 M soup's blob is built by rank 0 (tools/make_soup_blob.py: the
    GPU BVH builder) while the others wait; every rank then uploads its replica. Returns {mrays, ms, gather_ms, rays, spp} on rank 0 (ms = max over ranks).
    Every local phase (scene build, upload, the first dispatch) ends with all_ranks_ok(): a failure on ANY rank makes EVERY rank return {"failed": ...} before the next
    collective — nobody raises, nobody hangs, and the caller's headline line is printed regardless (tests/test_dist_gloo.py)."""
    wl = WORKLOADS[key]
    device = device if device is not None else torch.device("cuda", local_rank)
    err = None
    blob = os.path.join(built_dir, wl["blob"] + ".blob")
    try:
        if wl.get("triangles") and not os.path.exists(blob) and rank == 0:
            blob = workload_blob(key, built_dir)
    except Exception as e:
        err = f"scene build: {type(e).__name__}: {e}"
    if not all_ranks_ok(err is None, torch, dist, world, device):          # (also the barrier the other ranks wait at while rank 0 builds)
        return {"failed": (err or "scene build failed on another rank")[:300]}
    if wl.get("triangles") and not os.path.exists(blob):
        blob = workload_blob(key, built_dir)
    if not all_ranks_ok(os.path.exists(blob), torch, dist, world, device):
        return {"skipped": f"{blob} not built"}
    W, H, B, spp = wl["width"], wl["height"], wl["bounces"], SCALING_SPP[key]
    scene = fr = None
    try:
        scene = api.Scene(blob)
        fr = render.FrameRenderer(api, scene, W, H, device=local_rank, rank=rank, world=world, tile=wl["tile"], order=wl["tile_order"])
        fr.ctx.set_option(api.abi.OPT_COUNTER_LEVEL, 1)
        fr.render(min(spp, 8), B)          # warm-up dispatch (code object, buffers); its gather follows once every rank has come this far
    except Exception as e:
        err = f"upload / first dispatch: {type(e).__name__}: {e}"

    def close():
        for obj in (fr, scene):
            try:
                if obj is not None:
                    obj.close()
            except Exception:
                pass
    if not all_ranks_ok(err is None, torch, dist, world, device):
        close()
        return {"failed": (err or "upload / first dispatch failed on another rank")[:300]}

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    fr.reduce(dist); barrier()          # (the collective's first use)
    fr.ctx.reset_counters()
    e0, e1, e2 = (torch.cuda.Event(enable_timing=True) for _ in range(3))
    best = None
    for _ in range(reps):
        barrier()
        t0 = time.perf_counter()
        with torch.cuda.stream(fr.stream):
            e0.record(fr.stream)
        fr.render(spp, B)
        with torch.cuda.stream(fr.stream):
            e1.record(fr.stream)
        fr.reduce(dist)
        with torch.cuda.stream(fr.stream):
            e2.record(fr.stream)
        barrier()
        wall = (time.perf_counter() - t0) * 1e3
        rec = (wall, e0.elapsed_time(e1), e1.elapsed_time(e2))
        best = rec if best is None or rec[0] < best[0] else best
    rays = fr.ctx.counters()["rays"] / reps
    tt = torch.tensor([best[0], best[1], best[2], rays], dtype=torch.float64, device=fr.device)
    if world > 1:
        tmax = tt.clone()
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dist.all_reduce(tt, op=dist.ReduceOp.SUM)
        wall_ms, render_ms, gather_ms, total_rays = float(tmax[0]), float(tmax[1]), float(tmax[2]), float(tt[3])
    else:
        wall_ms, render_ms, gather_ms, total_rays = (float(v) for v in tt)
    close()
    full_spp = SCALING_FULL_SPP_NOTE.get(key, "")
    return {"workload": wl["what"].format(W=W, H=H, SPP=f"{spp} of {wl['samples']}", B=B), "spp": spp, "n_gpus": world,
            "mrays": round(total_rays / wall_ms / 1e3, 1), "ms": round(wall_ms, 3), "render_ms": round(render_ms, 3), "gather_ms": round(gather_ms, 3),
            "rays": int(total_rays), "scaling": "strong (the frame is a fixed job)",
            "note": "best of %d frames, wall clock between barriers, max over ranks; render_ms / gather_ms = stream events around this rank's dispatch / the strip gather "
                    "(at N > 1 a rank's gather_ms includes waiting for the slowest peer)%s" % (reps, full_spp)}


def parity_columns(img, ref):
    """BASELINE.md section 3's parity columns for two float frames: share of pixels whose 8-bit sRGB value differs by more than 1 LSB in any channel,
    mean and 99.9th percentile of the per-pixel L2 distance (linear values clipped to [0, 1]), RMSE, share of floats that differ at all."""
    import numpy as np
    sys.path.insert(0, os.path.join(REPO, "oracle"))
    import oracle_py
    a8, b8 = oracle_py.to_srgb8(img).astype(np.int16), oracle_py.to_srgb8(ref).astype(np.int16)
    a, b = np.clip(img.astype(np.float64), 0.0, 1.0), np.clip(ref.astype(np.float64), 0.0, 1.0)
    per_px = np.sqrt(((a - b) ** 2).sum(axis=2))
    return {"pixels_gt_1lsb_pct": round(float((np.abs(a8 - b8).max(axis=2) > 1).mean() * 100.0), 4), "mean_l2": float(per_px.mean()),
            "p999_l2": float(np.percentile(per_px, 99.9)), "rmse": float(np.sqrt(((a - b) ** 2).mean())),
            "floats_that_differ": int((img.view(np.uint32) != ref.view(np.uint32)).sum())}


def dropin_timing(workload):
    """The binary a c-ray user runs: c-ray-hip (reference main + loader + encoders, renderer.c replaced) on the same frame. Render phase =
    the timer of src/c-ray.c:279-281 around renderFrame(); CRH_DUMP_STATS splits it (flatten, context + upload, dispatch, download, sRGB)."""
    import subprocess
    import tempfile
    sys.path.insert(0, os.path.join(REPO, "tools"))
    exe = os.path.join(REPO, "c-ray_amd", "_lib", "c-ray-hip")
    overlay = os.path.join(REPO, "oracle", "_ref", "input")
    if not (os.path.exists(exe) and os.path.exists(os.path.join(overlay, workload["scene"]))):
        return {"skipped": "c-ray-hip or the asset overlay is not built"}
    import refrun
    with tempfile.TemporaryDirectory() as tmp:
        scene = refrun.rewrite_scene(workload["scene"], workload["width"], workload["height"], workload["samples"], workload["bounces"],
                                     tile=workload["tile"], out_dir=tmp)
        stats = os.path.join(tmp, "stats.json")
        env = dict(os.environ, CRH_DUMP_STATS=stats, CRAY_HIP_DEVICES="1", CRH_NO_IMAGE="1")
        t0 = time.perf_counter()
        proc = subprocess.run([exe], input=json.dumps(scene).encode(), cwd=overlay, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=900)
        wall = time.perf_counter() - t0
        if proc.returncode != 0 or not os.path.exists(stats):
            return {"failed": proc.stdout.decode(errors="replace")[-300:]}
        st = json.load(open(stats))
        if os.environ.get("CRH_TRACE_SYNC") or os.environ.get("CRH_TRACE_UPLOAD"):            # dev: the library's launch / synchronize / upload trace lines of the child
            st["trace"] = [l for l in proc.stdout.decode(errors="replace").splitlines() if "trace" in l]
    phase = st.get("render_phase_ms", st["render_ms"])
    return {"program": "c-ray_amd/_lib/c-ray-hip < hdr.json (1 GPU)",
            "render_phase_ms": phase, "mrays": round(st["rays"] / phase / 1e3, 1),
            "mrays_over_frame_ms": round(st["rays"] / st["frame_ms"] / 1e3, 1) if st.get("frame_ms") else None,
            "render_ms": st["render_ms"], "kernel_ms": st.get("kernel_ms"), "dispatches": st.get("dispatches"), "launch_host_ms": st.get("launch_host_ms"),
            "resolve_srgb_ms": st["resolve_srgb_ms"], "download_ms": st.get("download_ms"), "gather_ms": st.get("gather_ms"),
            "frame_ms": st.get("frame_ms"), "teardown_ms": st.get("teardown_ms"), "setup_ms": st.get("setup_ms"), "flatten_ms": st["flatten_ms"], "context_ms": st.get("context_ms"),
            "upload_ms": st.get("upload_ms"), "process_wall_s": round(wall, 2), "rays": st["rays"],
            "phase_vs_kernel": round(phase / st["kernel_ms"], 4) if st.get("kernel_ms") else None, **({"trace": st["trace"]} if st.get("trace") else {}),
            "note": "render_phase_ms = SURVEY 8(d)'s phase: the timer of src/c-ray.c:279-281 around renderFrame() (frame_ms: ALL of it, since round 4 the teardown too — teardown_ms: the contexts, "
                    "the framebuffers) minus the set-up and the teardown (setup_ms: everything "
                    "before the GPU was ready to dispatch — context + code objects + per-wave buffers, which run beside the flattener, then the scene upload); it holds "
                    "the dispatch (render_ms; kernel_ms = its GPU time), the 8-bit conversion on the device + its download (resolve_srgb_ms), the float "
                    "buffer's download and the host in between; mrays = rays / render_phase_ms. mrays_over_frame_ms = rays / frame_ms: the rate over what the REFERENCE's own "
                    "timer (c-ray.c:279-281) would show a user of the drop-in — set-up included — and the figure to hold against the reference's CPU rate, whose timer "
                    "includes its (BVH-free, cheap) set-up too. process_wall_s also holds JSON / OBJ parsing and the GPU BVH build"}


def dropin_iterative(workload, samples=129):
    """c-ray-hip --iterative (renderThreadInteractive's replacement) on the same frame, `samples` - 1 passes: how busy the GPU stays while the host converts
    to 8 bit on the device, downloads and redraws between dispatches of about 16 ms."""
    import subprocess
    import tempfile
    sys.path.insert(0, os.path.join(REPO, "tools"))
    exe = os.path.join(REPO, "c-ray_amd", "_lib", "c-ray-hip")
    overlay = os.path.join(REPO, "oracle", "_ref", "input")
    if not (os.path.exists(exe) and os.path.exists(os.path.join(overlay, workload["scene"] or "-"))):
        return {"skipped": "c-ray-hip or the asset overlay is not built"}
    import refrun
    with tempfile.TemporaryDirectory() as tmp:
        scene = refrun.rewrite_scene(workload["scene"], workload["width"], workload["height"], samples, workload["bounces"], tile=workload["tile"], out_dir=tmp)
        stats = os.path.join(tmp, "stats.json")
        proc = subprocess.run([exe, "--iterative"], input=json.dumps(scene).encode(), cwd=overlay, env=dict(os.environ, CRH_DUMP_STATS=stats, CRAY_HIP_DEVICES="1"),
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=900)
        if proc.returncode != 0 or not os.path.exists(stats):
            return {"failed": proc.stdout.decode(errors="replace")[-300:]}
        return json.load(open(stats))


def cpu_baseline(oracle_py, blob_path, w, h, bounces, budget_s=12.0, frames=None):
    """Reference pthread renderer on the host cores, bounded sample of the same frame (reduced spp). `frames` (dict) receives the float frames the CPU
    side rendered — "strict" (the bit-exact restatement of c-ray-ref-strict) and "default" (c-ray-ref, upstream's default flags) — and their spp."""
    frames = {} if frames is None else frames
    import numpy as np
    sys.path.insert(0, os.path.join(REPO, "tools"))
    cores = os.cpu_count() or 1
    oscene = oracle_py.OracleScene(blob_path)
    # size the sample with the (bit-exact, OpenMP) restatement: 1 spp probe
    t = time.time()
    _, c1 = oracle_py.render(oscene, w, h, 1, bounces)
    probe = max(time.time() - t, 1e-3)
    spp = int(max(2, min(64, budget_s / probe)))
    ref_exe = os.path.join(REPO, "oracle", "_ref", "c-ray-ref")
    overlay = os.path.join(REPO, "oracle", "_ref", "input", WORKLOAD["scene"] or "-")       # (the 10 M soup has no scene file on this box: the restatement is its CPU baseline)
    t = time.time()
    port_img, cnt = oracle_py.render(oscene, w, h, spp, bounces)
    port_s = time.time() - t
    frames["spp"] = spp
    frames["strict"] = port_img          # the restatement = c-ray-ref-strict bit for bit (tests/golden: 15 fixtures)
    if os.path.exists(ref_exe) and os.path.exists(overlay):
        try:
            import refrun
            ref_img, st = refrun.render_reference(WORKLOAD["scene"], w, h, spp, bounces, flavour="default", threads=cores)
            frames["default"] = ref_img
            secs = st["render_ms"] / 1e3
            out = {"value": round(cnt["rays"] / secs / 1e6, 3), "unit": "Mray/s", "cores": cores, "kind": "reference",
                   "sample": f"oracle/_ref/c-ray-ref -j {cores}: {WORKLOAD['scene']} {w}x{h}, {spp} spp (of {WORKLOAD['samples']}), {bounces} bounces, the scene file's own "
                             f"{WORKLOAD['tile'][0]}x{WORKLOAD['tile'][1]} tiles, render phase {secs:.2f} s, {cnt['rays']} rays (counted by the bit-exact restatement)",
                   "port_value": round(cnt["rays"] / port_s / 1e6, 3)}
            # a FAIR column: the scene file's 64x64 tiles give a 1280x720 frame 240 tiles — with more threads than tiles every thread gets at most one and
            # the frame lasts as long as its slowest tile (tile.c:66-117). The same sample with 16x16 tiles (args.c: the -t option's effect; here through the scene's
            # renderer.tileWidth / tileHeight) keeps every core busy: the figure to divide by when a "x host CPU" ratio is wanted
            try:
                _img16, st16 = refrun.render_reference(WORKLOAD["scene"], w, h, spp, bounces, flavour="default", threads=cores, tile=(16, 16))
                out["value_tiles16"] = round(cnt["rays"] / (st16["render_ms"] / 1e3) / 1e6, 3)
                out["sample_tiles16"] = f"the same frame and sample with 16x16 tiles: render phase {st16['render_ms'] / 1e3:.2f} s"
            except Exception as e:
                out["value_tiles16"] = None
                out["sample_tiles16"] = f"failed: {type(e).__name__}"
            return out
        except Exception as e:  # fall back to the restatement, say so
            note = f" (reference binary failed: {type(e).__name__})"
    else:
        note = " (oracle/_ref binary or asset overlay absent)"
    return {"value": round(cnt["rays"] / port_s / 1e6, 3), "unit": "Mray/s", "cores": cores, "kind": "port",
            "sample": f"oracle/libcray_oracle.so (OpenMP, {cores} threads): {w}x{h}, {spp} spp (of {WORKLOAD['samples']}), {bounces} bounces, {port_s:.2f} s" + note}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", choices=sorted(WORKLOADS), default="cfg2",
                    help="cfg2 (default, the headline: BASELINE.json configs[1]); cfg4 = configs[3], the scene the 1/2/4/8-GPU curve is quoted on; "
                         "soup10m = configs[4] at its real size (the scene is built on this box on first use)")
    ap.add_argument("--samples", type=int, default=0, help=argparse.SUPPRESS)   # dev only: fewer passes than the config names
    ap.add_argument("--no-cpu", action="store_true", help=argparse.SUPPRESS)
    ap.add_argument("--no-dropin", action="store_true", help=argparse.SUPPRESS)
    ap.add_argument("--no-others", action="store_true", help=argparse.SUPPRESS)
    a = ap.parse_args()
    global WORKLOAD
    WORKLOAD = WORKLOADS[a.workload]
    if not a.samples:
        a.samples = WORKLOAD["samples"]

    import torch
    from __graft_entry__ import load_package, BUILT
    pkg = load_package()
    api, abi, render = pkg.api, pkg.abi, pkg.render

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if a.gpus != world:
        if world == 1 and a.gpus > 1:
            raise SystemExit("bench.py --gpus N > 1 must be launched with torch.distributed.run (one rank per GPU)")
    if hasattr(api.library(), "crh_emu_stats"):
        raise SystemExit("bench.py: CRH_LIB names the CPU emulation of the kernels (tests/emu): bench.py measures the MI355X path only")
    if api.device_count() < 1:
        raise SystemExit("bench.py: no HIP device visible; libcray_hip has no CPU fallback")
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.cuda.set_device(local_rank)
        dist.init_process_group(backend="nccl", rank=rank, world_size=world)

    W, H, SPP, B = WORKLOAD["width"], WORKLOAD["height"], a.samples, WORKLOAD["bounces"]
    blob = workload_blob(a.workload, BUILT) if rank == 0 or not WORKLOAD.get("triangles") else None
    if world > 1 and WORKLOAD.get("triangles"):          # built once, by rank 0
        dist.barrier()
        blob = workload_blob(a.workload, BUILT)
    if not os.path.exists(blob):
        raise SystemExit(f"bench.py: {blob} missing — run __graft_entry__.build() where /root/reference exists")
    scene = api.Scene(blob)
    fr = render.FrameRenderer(api, scene, W, H, device=local_rank, rank=rank, world=world,
                              tile=WORKLOAD["tile"], order=WORKLOAD["tile_order"])
    ctx = fr.ctx

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    gather_events = []

    def step():
        fr.render(SPP, B)
        if world > 1:              # the gather's own stream time, reported beside the frame (it is inside the timed region either way)
            ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
            ev[0].record(fr.stream)
            fr.reduce(dist)
            ev[1].record(fr.stream)
            gather_events.append(ev)
        else:
            fr.reduce(dist)

    # counting pass (outside the timed region): every crh_counters field for the roofline's algorithmic bytes
    ctx.set_option(abi.OPT_COUNTER_LEVEL, 2)
    ctx.reset_counters()
    step()
    barrier()
    full = ctx.counters()

    ctx.set_option(abi.OPT_COUNTER_LEVEL, 1)          # timed runs keep only the ray / path counters
    for _ in range(a.warmup):
        step()
    barrier()
    ctx.reset_counters()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        step()
    barrier()
    elapsed = time.perf_counter() - t0
    cnt = ctx.counters()
    _, kernel_total_ms, launches = ctx.kernel_time_ms()
    kernel_name = ctx.last_kernel_name()
    gather_ms = sum(a_.elapsed_time(b_) for a_, b_ in gather_events[-a.steps:]) / max(a.steps, 1) if gather_events else 0.0

    tt = torch.tensor([elapsed, float(cnt["rays"]), float(cnt["paths"]), gather_ms, kernel_total_ms / max(launches, 1)], dtype=torch.float64, device=fr.device)
    kernel_ms_max = float(tt[4])
    if world > 1:
        tmax = tt.clone()
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dist.all_reduce(tt, op=dist.ReduceOp.SUM)
        elapsed, gather_ms, kernel_ms_max = float(tmax[0]), float(tmax[3]), float(tmax[4])
    total_rays, total_paths = float(tt[1]), float(tt[2])

    out = None
    if rank == 0:
        ms_per_step = elapsed / a.steps * 1e3
        value = total_rays / elapsed / 1e6
        avg_kernel_ms = kernel_total_ms / max(launches, 1)
        alg = algorithmic_bytes(full)
        alg_no_state = algorithmic_bytes(full, path_state=False)
        achieved = alg / (avg_kernel_ms * 1e-3) / 1e9
        traffic, traffic_upper, valu = (None, None, None)
        if world == 1 and SPP == WORKLOAD["samples"]:
            traffic, traffic_upper, valu = measured_profile(a.workload)
        sat = ((valu or {}).get("calibration") or {}).get("of_saturated_rate")
        frac_state = achieved / HBM_PEAK_GBS
        achieved_scene = alg_no_state / (avg_kernel_ms * 1e-3) / 1e9
        out = {
            "metric": "Mray/s (primary+secondary)", "value": round(value, 2), "unit": "Mray/s", "n_gpus": world,
            "steps": a.steps, "warmup": a.warmup, "ms_per_step": round(ms_per_step, 3), "higher_is_better": True,
            "scaling": "strong", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": WORKLOAD["what"].format(W=W, H=H, SPP=SPP, B=B),
                       "rays_per_step": int(total_rays / a.steps), "paths_per_step": int(total_paths / a.steps),
                       "parallelism": ("1 rank: the whole frame in one dispatch" if world == 1 else
                                       f"4-row strips interleaved over {world} ranks; rank 0 gathers the owned strips (1/{world} of the float framebuffer per rank, "
                                       "RCCL send / receive over xGMI; render.py: StripGather — bit-identical to the one-reduce form)"),
                       "baseline_config": {"cfg2": "configs[1] (the 1-GPU headline; at N > 1 a STRONG-scaling run of the same 68 ms frame)", "cfg3": "configs[2]",
                                           "cfg4": "configs[3] (the scene BASELINE.json quotes the 1/2/4/8-GPU curve on: --workload cfg4)",
                                           "soup": "configs[4] at 1 M triangles", "soup10m": "configs[4]"}.get(a.workload)},
            "gather_ms": round(gather_ms, 3) if world > 1 else None, "kernel_ms_max_over_ranks": round(kernel_ms_max, 3),
            # `bound` is set from the evidence: the PMC run of THIS device code (profiles/, fingerprint-gated) shows the vector ALU issuing most of the time
            # -> "valu-issue"; without such a run the label is the metric's nominal one, "hbm", and the note says that nothing measured backs it
            # `bound` is set from the evidence, CALIBRATED since round 5: the vector pipe's formula is held against its reading at known saturation (profiles/calibration.json)
            "roofline": {"bound": ("valu-issue" if sat >= 0.7 else "memory-latency + valu-issue") if sat else ("valu-issue" if (valu or {}).get("pipe_busy", 0) >= 0.7 else "hbm"),
                         "bound_evidence": (("profiles/hbm_traffic.json `valu` (rocprofv3 PMC of this device code): the vector pipe issues at %.0f %% of its saturated rate (formula reading %.2f "
                                             "against %.2f for a saturating loop of the node step's instruction blend), lane utilisation %.0f %%; the waves are parked on memory waits half of their "
                                             "cycles at 4 waves per SIMD (profiles/*_pmc_deep.txt): neither HBM bandwidth nor the issue rate alone binds — a wave waits for the slowest of its "
                                             "lanes' cache misses, and too few of a SIMD's four waves are runnable to fill the pipe")
                                            % (100 * sat, valu["pipe_busy"], valu["pipe_busy"] / sat, 100 * valu.get("lane_utilisation", 0))) if sat else
                                           ("profiles/hbm_traffic.json `valu` (rocprofv3 PMC of this device code): VALU pipe formula %.2f (uncalibrated), lane utilisation %.0f %%"
                                            % (valu["pipe_busy"], 100 * valu.get("lane_utilisation", 0))) if valu and valu.get("pipe_busy") else
                                           "no PMC run of this device code in profiles/ (fingerprint mismatch): nominal label",
                         "achieved": round(achieved_scene, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved_scene / HBM_PEAK_GBS, 5), "traffic": traffic,
                         **fractions(alg_no_state, traffic if world == 1 else None, avg_kernel_ms, traffic_upper if world == 1 else None),
                         "kernel": kernel_name, "avg_launch_ms": round(avg_kernel_ms, 3), "algorithmic_bytes_per_launch": int(alg_no_state),
                         "bytes_per_ray": round(alg_no_state / max(full["rays"], 1), 1),
                         "frac_with_path_state": round(frac_state, 5),
                         "what_it_is": "`achieved` / `frac` (= frac_algorithmic) are the ALGORITHMIC rate, not an HBM measurement: scene records touched (node / triangle / "
                                       "instance / shading / texel), B_state = 0 (SURVEY 8(d): persistent megakernel), divided by the launch time. Most of those bytes are "
                                       "served by L2 and the 256 MB Infinity Cache; `traffic` is the measured L2<->fabric volume per launch (FETCH_SIZE and WRITE_SIZE with the factors "
                                       "calibrated on divergent 64-byte gathers and 128-byte record writes — profiles/calibration.json; until round 4 the read side was doubled, which "
                                       "holds for 128-byte requests only: frac_measured_traffic_upper —, MALL hits included), frac_measured_traffic the same over time and HBM peak; both are "
                                       "null whenever profiles/hbm_traffic.json was not measured on this device code (fingerprint of the k_pathtrace* machine code) / this workload. "
                                       "frac_with_path_state adds 152 B/ray for the per-wave path table. What binds the kernel: `bound_evidence`",
                         "valu": valu,
                         # the one number that says how far the kernel is from the machine: the share of the vector pipe's saturated issue rate it uses
                         # x the share of a wave's 64 lanes that are active in what it issues
                         "lane_fraction": round(sat * valu["lane_utilisation"], 4) if sat and (valu or {}).get("lane_utilisation") else None},
        }
        if world > 1:
            # the headline FIRST: the scaling objects below are the first time RCCL's grouped send / receive meets the 1 GB soup on N real GPUs — whatever
            # happens there, this line is out. The complete line follows when they have run
            print(json.dumps({**out, "provisional": "headline only: the complete line (scaling_cfg4, scaling_soup10m) follows"}), flush=True)
    scaling = {}
    if a.workload == "cfg2" and SPP == WORKLOAD["samples"] and not a.no_others:
        for key in ("cfg4", "soup10m"):
            try:
                scaling[key] = scaling_workload(key, api, render, torch, dist, BUILT, local_rank, rank, world)
            except Exception as e:          # must not take the headline line with it (local failures come back as {"failed": ...} on every rank: scaling_workload)
                scaling[key] = {"failed": f"{type(e).__name__}: {e}"[:300]}
    if rank == 0:
        for key, obj in scaling.items():
            out["scaling_" + key] = obj
        if world == 1 and not a.no_cpu:
            sys.path.insert(0, os.path.join(REPO, "oracle"))
            import oracle_py
            frames = {}
            out["cpu_baseline"] = cpu_baseline(oracle_py, blob, W, H, B, frames=frames)
            if frames.get("spp") and WORKLOAD.get("scene"):
                # BASELINE.md section 3's parity columns: the GPU frame at the CPU sample's spp (the seed depends on the sample count: sampler.c:42) against
                # the strict flavour (the parity contract: 0 everywhere) and against upstream's default flags (FMA contraction on: how far the contract's
                # flavour is from what a user's own build renders — chaos, not error: DESIGN.md section 5)
                try:
                    import numpy as np
                    fbp = ctx.framebuffer(W, H)
                    ctx.render_region(fbp, W, H, frames["spp"], B)
                    gpu_img = ctx.download(fbp, W, H)
                    out["parity"] = {"frame": f"{W}x{H}, {frames['spp']} spp, {B} bounces",
                                     "vs_reference_strict_flavour": parity_columns(gpu_img, frames["strict"])}
                    if "default" in frames:
                        out["parity"]["vs_reference_default_flags"] = parity_columns(gpu_img, frames["default"])
                        out["parity"]["reference_strict_vs_default_flags"] = parity_columns(frames["strict"], frames["default"])
                except Exception as e:
                    out["parity"] = {"failed": f"{type(e).__name__}: {e}"[:200]}
        if world == 1 and not a.no_dropin and SPP == WORKLOAD["samples"] and a.workload == "cfg2":
            try:
                fr.close()                 # the drop-in is its own process with its own context
            except Exception:
                pass
            out["dropin"] = dropin_timing(WORKLOAD)
            out["dropin_iterative"] = dropin_iterative(WORKLOAD)
            if isinstance(out["dropin"].get("render_phase_ms"), (int, float)):
                out["dropin"]["vs_bench_ms_per_step"] = round(out["dropin"]["render_phase_ms"] / ms_per_step, 3)
        if world == 1 and not a.no_others and a.workload == "cfg2" and SPP == WORKLOAD["samples"]:
            out["other_workloads"] = measure_other_workloads(api, abi, BUILT)
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

"""The N > 1 path on CPU: world_size 2, gloo. Each rank takes its share of the reference's ordered tile list,
renders it into a zeroed framebuffer and one reduce(SUM) onto rank 0 assembles the frame — the same host logic
bench.py and c-ray_amd/render.py run with backend nccl (= RCCL) on GPUs. The per-rank renderer here is the
oracle (tests may use it; there is no GPU in this tier)."""
import os
import socket
import sys

import numpy as np
import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, blob, w, h, spp, bounces, out_path, how="reduce"):
    sys.path.insert(0, REPO)
    sys.path.insert(0, os.path.join(REPO, "oracle"))
    import torch
    import torch.distributed as dist
    from __graft_entry__ import load_package
    import oracle_py
    pkg = load_package()
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    scene = oracle_py.OracleScene(blob)
    mine = pkg.render.owned_tiles(w, h, 32, 32, pkg.tiles.ORDER_FROM_MIDDLE, rank, world)
    fb = np.zeros((h, w, 3), np.float32)
    for t in mine:
        oracle_py.render(scene, w, h, spp, bounces, region=t, fb=fb, threads=2)
    tfb = torch.from_numpy(fb)
    if how == "gather":        # the default of FrameRenderer.reduce: only the owned strips travel
        tfb = pkg.render.StripGather(torch, w, h, rank, world, torch.device("cpu"))(tfb, dist)
    else:
        pkg.render.reduce_frame(tfb, world, dist, dst=0)
    if rank == 0:
        np.save(out_path, tfb.numpy())
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_tile_sharding_reproduces_the_frame(manifest, golden_blob, golden_ref, tmp_path):
    import torch.multiprocessing as mp
    m = manifest["fence"]
    out = str(tmp_path / "frame.npy")
    mp.spawn(_worker, args=(2, _free_port(), golden_blob("fence"), m["width"], m["height"], m["samples"], m["bounces"], out),
             nprocs=2, join=True)
    frame = np.load(out)
    assert np.array_equal(frame, golden_ref("fence")), "2-rank frame differs from the reference's single-process frame"


def test_eight_rank_strip_sharding_reproduces_the_frame(manifest, golden_blob, golden_ref, tmp_path):
    """World size 8 with a frame height (100) that is not a multiple of 4 x 8: ragged last strip, ranks with unequal strip counts."""
    import torch.multiprocessing as mp
    m = manifest["fence"]
    assert m["height"] % (4 * 8) != 0
    out = str(tmp_path / "frame8.npy")
    mp.spawn(_worker, args=(8, _free_port(), golden_blob("fence"), m["width"], m["height"], m["samples"], m["bounces"], out),
             nprocs=8, join=True)
    assert np.array_equal(np.load(out), golden_ref("fence")), "8-rank frame differs from the reference's single-process frame"


@pytest.mark.parametrize("world", [2, 8])
def test_strip_gather_reproduces_the_frame(world, manifest, golden_blob, golden_ref, tmp_path):
    """The same frames with the strips gathered instead of the whole buffer reduced (render.py: StripGather, the bench's default):
    ragged height, ranks with unequal row counts (padded to a common shape for the gather)."""
    import torch.multiprocessing as mp
    m = manifest["fence"]
    out = str(tmp_path / f"frame_gather{world}.npy")
    mp.spawn(_worker, args=(world, _free_port(), golden_blob("fence"), m["width"], m["height"], m["samples"], m["bounces"], out, "gather"),
             nprocs=world, join=True)
    assert np.array_equal(np.load(out), golden_ref("fence")), f"{world}-rank gathered frame differs from the reference's single-process frame"


def _cfg4_worker(rank, world, port, w, h, out_path):
    """bench.py: scaling_cfg4's host path at BASELINE configs[3]'s frame size — this rank's strips (owned_tiles), a per-pixel pattern standing in for the
    dispatch (every owned pixel a value only its coordinates determine, every other pixel 0), the strip gather onto rank 0."""
    sys.path.insert(0, REPO)
    import torch
    import torch.distributed as dist
    from __graft_entry__ import load_package
    pkg = load_package()
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    fb = np.zeros((h, w, 3), np.float32)
    ys, xs = np.mgrid[0:h, 0:w]
    full = np.stack([xs * 0.25 + ys + 1.0, ys * 0.5 + xs + 2.0, (xs ^ ys) + 3.0], axis=2).astype(np.float32)          # (never 0: an unowned pixel would show)
    for x0, y0, x1, y1 in pkg.render.owned_tiles(w, h, 64, 64, 1, rank, world):
        fb[h - y1:h - y0, x0:x1] = full[h - y1:h - y0, x0:x1]          # framebuffer rows are stored top-down (texture.c:24-28)
    tfb = pkg.render.StripGather(torch, w, h, rank, world, torch.device("cpu"))(torch.from_numpy(fb), dist)
    if rank == 0:
        np.save(out_path, np.array([np.array_equal(tfb.numpy(), full), float((tfb.numpy() != 0).all())]))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("w,h,world", [(3840, 2160, 2), (2560, 1440, 3)])
def test_scaling_cfg4_share_and_gather_at_full_frame_size(w, h, world, tmp_path):
    """bench.py emits `scaling_cfg4` (statues.json 3840x2160, the scene BASELINE.json quotes the 1/2/4/8-GPU curve on) and — round 5 — `scaling_soup10m` (configs[4]'s
    2560x1440 frame) through the same strips + gather as the headline: two (three: a world size that does not divide the 360 strips) gloo ranks assemble those frame
    sizes exactly — every pixel from exactly one owner."""
    import torch.multiprocessing as mp
    out = str(tmp_path / "ok.npy")
    mp.spawn(_cfg4_worker, args=(world, _free_port(), w, h, out), nprocs=world, join=True)
    ok = np.load(out)
    assert ok[0] == 1.0 and ok[1] == 1.0


def test_strip_rows_match_the_c_hosts_pack_order(pkg):
    """render.py: strip_rows lists, per rank, the framebuffer rows k_strip_rows (cray_hip.hip: crh_frames_gather) packs: strip by strip, bottom row of a strip first."""
    for h, world in ((100, 8), (720, 8), (7, 3), (3, 2), (9, 16)):
        seen = []
        for r in range(world):
            rows = pkg.render.strip_rows(h, r, world)
            ys = [h - 1 - x for x in rows]
            want = [(r + (p // 4) * world) * 4 + p % 4 for p in range(len(rows))]      # k_strip_rows' y of packed row p
            assert ys == want, (h, world, r)
            seen += ys
        assert sorted(seen) == list(range(h))


@pytest.mark.parametrize("width,height,gpus", [(160, 100, 8), (1280, 720, 8), (3840, 2160, 8), (33, 7, 3), (64, 3, 2), (5, 9, 16)])
def test_c_host_partition_equals_the_python_hosts_and_covers_the_frame_once(width, height, gpus, pkg, tmp_path):
    """c-ray-hip (host/share.h: crh_strip_share) and render.py (owned_tiles) must own the same pixels on every rank; together the
    shares cover every pixel exactly once (ragged heights, more GPUs than strips)."""
    import subprocess
    src = tmp_path / "share.c"
    src.write_text('#include <stdio.h>\n#include <stdlib.h>\n#include "share.h"\n'
                   'int main(int c, char **v) { int W = atoi(v[1]), H = atoi(v[2]), G = atoi(v[3]);\n'
                   ' crh_tile *t = calloc(crh_strip_share_max(H, G), sizeof(*t));\n'
                   ' for (int g = 0; g < G; ++g) { uint32_t n = crh_strip_share(W, H, g, G, t); if (n > crh_strip_share_max(H, G)) return 1;\n'
                   '  for (uint32_t i = 0; i < n; ++i) printf("%d %d %d %d %d\\n", g, t[i].x0, t[i].y0, t[i].x1, t[i].y1); }\n return 0; }\n')
    exe = tmp_path / "share"
    subprocess.check_call(["gcc", "-std=gnu99", "-I" + os.path.join(REPO, "include"), "-I" + os.path.join(REPO, "c-ray_amd", "host"), str(src), "-o", str(exe)])
    rows = [tuple(int(x) for x in line.split()) for line in subprocess.check_output([str(exe), str(width), str(height), str(gpus)]).decode().splitlines()]
    cover = np.zeros((height, width), np.int32)
    for g in range(gpus):
        mine = [r[1:] for r in rows if r[0] == g]
        assert mine == [tuple(t) for t in pkg.render.owned_tiles(width, height, 64, 64, pkg.tiles.ORDER_FROM_MIDDLE, g, gpus)], g
        for x0, y0, x1, y1 in mine:
            cover[y0:y1, x0:x1] += 1
    assert (cover == 1).all()


def _failing_side_workload_worker(rank, world, port, built_dir, out_dir):
    """bench.py's scaling_workload with a renderer that cannot start on rank 1 (and starts on rank 0): what every rank gets back."""
    sys.path.insert(0, REPO)
    import importlib.util
    import json
    import types
    import torch
    import torch.distributed as dist
    spec = importlib.util.spec_from_file_location("bench_under_test", os.path.join(REPO, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)

    class Scene:
        def __init__(self, blob): self.blob = blob
        def close(self): pass

    class FrameRenderer:
        def __init__(self, api, scene, w, h, device, rank, world, tile, order):
            if rank == 1:
                raise RuntimeError("hipMalloc: out of memory (a GPU that cannot hold the scene)")
            self.ctx = types.SimpleNamespace(set_option=lambda *a: None)
        def render(self, spp, bounces): pass
        def close(self): pass
    api = types.SimpleNamespace(Scene=Scene, abi=types.SimpleNamespace(OPT_COUNTER_LEVEL=1))
    render = types.SimpleNamespace(FrameRenderer=FrameRenderer)
    got = bench.scaling_workload("cfg4", api, render, torch, dist, built_dir, rank, rank, world, device=torch.device("cpu"))
    json.dump(got, open(os.path.join(out_dir, f"rank{rank}.json"), "w"))
    dist.barrier()          # (every rank is still in step with the others: the next collective of the bench — its closing barrier — goes through)
    dist.destroy_process_group()


def test_a_side_workload_that_fails_on_one_rank_fails_on_all_and_nobody_hangs(tmp_path):
    """at N > 1 a failure inside a scaling object on ONE rank used to re-raise there and leave its peers in the next collective — taking the headline
    line of that N with it. Now the ranks agree (all_ranks_ok) after every local phase; all of them return {"failed": ...} and carry on. bench.py also prints the
    headline line before the scaling objects run (the same dict, marked provisional)."""
    import json
    import torch.multiprocessing as mp
    built = tmp_path / "built"
    built.mkdir()
    (built / "cfg4_statues.blob").write_bytes(b"not a scene: the stand-in renderer never reads it")
    mp.spawn(_failing_side_workload_worker, args=(2, _free_port(), str(built), str(tmp_path)), nprocs=2, join=True)
    r0, r1 = (json.load(open(tmp_path / f"rank{r}.json")) for r in (0, 1))
    assert "failed" in r0 and "another rank" in r0["failed"], r0
    assert "failed" in r1 and "out of memory" in r1["failed"], r1
    src = open(os.path.join(REPO, "bench.py")).read()
    assert src.index('"provisional"') < src.index("scaling[key] = scaling_workload("), "the headline line must be printed before the scaling objects run"

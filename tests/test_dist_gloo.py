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


def _worker(rank, world, port, blob, w, h, spp, bounces, out_path):
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

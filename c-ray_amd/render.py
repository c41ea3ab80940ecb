"""render.py — host-side driver shaped like the reference's renderFrame() (src/renderer/renderer.c:40-180).

One process per GPU (torch.distributed; backend "nccl" = RCCL over xGMI on ROCm). Every rank holds a
replica of the flattened scene, takes its share of the frame — every world-th 4-row strip (owned_tiles below; one
rank: the whole frame as one region) — and renders it with ONE crh_render_tiles dispatch into a zeroed float
framebuffer; a single reduce(SUM) to rank 0 then assembles the frame — pixels a rank does not own are
exactly 0.0f, so the sum is a gather and the result is bit-identical to the 1-GPU frame.
PyTorch is used for device memory, the stream and the collective only; all rendering is behind the C-ABI.
"""
from . import tiles as tiles_mod


STRIP_ROWS = 4


def owned_tiles(width, height, tile_w, tile_h, order, rank, world):
    """This rank's share of the frame, as a tile list for crh_render_tiles.

    One rank: the whole frame as ONE region (the library hands its pixel blocks out bottom-up; the reference's tile orders,
    tile.c:119-241, are a preview preference and cost 2-10 % as work orders: tools/probe_tile_order.py). Several ranks: horizontal
    strips of STRIP_ROWS pixel rows, strip i owned by rank i mod world. Dealing out the reference's tiles instead (tile i -> rank i mod world) looks
    natural but is badly balanced for the orders that start in the middle: on the bench frame two ranks get 236 M and
    346 M rays (45 / 70 ms), eight ranks 9.8 ... 20.4 ms; 4-row strips give 59.2 / 59.2 ms and 16.9 ... 17.6 ms
    (tools/probe_rank_share.py). Any disjoint cover reproduces the frame bit for bit: a pixel's passes fold on its owner."""
    if world <= 1:
        return [(0, 0, width, height)]
    strips = [(0, y, width, min(y + STRIP_ROWS, height)) for y in range(0, height, STRIP_ROWS)]
    return strips[rank::world]


def reduce_frame(fb, world, dist=None, dst=0):
    """Sum the per-rank framebuffers onto `dst` (RCCL reduce on GPU tensors, gloo on CPU tensors)."""
    if world > 1:
        dist.reduce(fb, dst=dst, op=dist.ReduceOp.SUM)
    return fb


class FrameRenderer:
    """renderFrame() for one GPU rank: scene replica + float framebuffer (a torch CUDA tensor) + dispatch."""

    def __init__(self, api, scene, width, height, device=0, rank=0, world=1, tile=(64, 64), order=tiles_mod.ORDER_FROM_MIDDLE):
        import torch
        self.torch = torch
        self.api = api
        self.width, self.height = int(width), int(height)
        self.rank, self.world = rank, world
        self.device = torch.device("cuda", device)
        torch.cuda.set_device(self.device)
        # A dedicated (non-default) torch stream: the kernels are launched on its raw hipStream_t through the C-ABI and the
        # tensor ops / the collective are issued under `with torch.cuda.stream(...)`, so everything is ordered on ONE stream.
        # (torch's default stream has the raw handle 0, which the C-ABI would read as "create your own stream".)
        self.stream = torch.cuda.Stream(device=self.device)
        self.ctx = api.Context(device, self.stream.cuda_stream)
        self.ctx.upload(scene)
        with torch.cuda.stream(self.stream):
            self.fb = torch.zeros((self.height, self.width, 3), dtype=torch.float32, device=self.device)
        self.tiles = owned_tiles(self.width, self.height, tile[0], tile[1], order, rank, world)

    def render(self, samples, bounces, clear=True):
        """Enqueue this rank's tiles (asynchronous on the stream)."""
        with self.torch.cuda.stream(self.stream):
            if clear:
                self.fb.zero_()
            if self.tiles:
                self.ctx.render_tiles(self.fb.data_ptr(), self.width, self.height, samples, bounces, self.tiles)

    def reduce(self, dist=None):
        with self.torch.cuda.stream(self.stream):
            return reduce_frame(self.fb, self.world, dist)

    def close(self):
        self.ctx.close()

"""render.py — host-side driver shaped like the reference's renderFrame() (src/renderer/renderer.c:40-180).

One process per GPU (torch.distributed; backend "nccl" = RCCL over xGMI on ROCm). Every rank holds a
replica of the flattened scene, takes tiles i = rank (mod world) of the reference's own ordered tile list
(tile.c:66-117 via tiles.py) and renders them with ONE crh_render_tiles dispatch into a zeroed float
framebuffer; a single reduce(SUM) to rank 0 then assembles the frame — pixels a rank does not own are
exactly 0.0f, so the sum is a gather and the result is bit-identical to the 1-GPU frame.
PyTorch is used for device memory, the stream and the collective only; all rendering is behind the C-ABI.
"""
from . import tiles as tiles_mod


def owned_tiles(width, height, tile_w, tile_h, order, rank, world):
    """This rank's share of the reference's ordered tile list."""
    all_tiles = tiles_mod.quantize_image(width, height, tile_w, tile_h, order)
    return tiles_mod.tiles_for_rank(all_tiles, rank, world)


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

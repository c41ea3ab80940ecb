"""render.py — host-side driver shaped like the reference's renderFrame() (src/renderer/renderer.c:40-180).

One process per GPU (torch.distributed; backend "nccl" = RCCL over xGMI on ROCm). Every rank holds a
replica of the flattened scene, takes its share of the frame — every world-th 4-row strip (owned_tiles below; one
rank: the whole frame as one region) — and renders it with ONE crh_render_tiles dispatch into a zeroed float
framebuffer; rank 0 then assembles the frame: either a single reduce(SUM) — pixels a rank does not own are
exactly 0.0f, so the sum is a gather — or (default) a gather of the owned strips only, 1 / world of the bytes
(StripGather). Both give the 1-GPU frame bit for bit.
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


def strip_rows(height, rank, world):
    """Framebuffer rows (texture.c:24-28: row H - 1 - y) of the strips `rank` owns, in strip order — the rows crh_frames_gather packs (C host)."""
    rows = []
    for y0 in range(rank * STRIP_ROWS, height, world * STRIP_ROWS):
        rows.extend(height - 1 - y for y in range(y0, min(y0 + STRIP_ROWS, height)))
    return rows


class StripGather:
    """The frame with 1 / world of the bytes on the links: every rank packs the rows of ITS strips (one index_select), one gather brings
    them to `dst` (RCCL: grouped send / receive, the peers' xGMI links in parallel), `dst` writes them into its framebuffer (index_copy_).
    Bit-identical to reduce_frame for strip shares: a sum with 0.0f changes nothing. Torch is plumbing here: copies and the collective."""

    def __init__(self, torch, width, height, rank, world, device, dst=0):
        self.torch, self.rank, self.world, self.dst = torch, rank, world, dst
        self.rows = [torch.tensor(strip_rows(height, r, world), dtype=torch.long, device=device) for r in (range(world) if rank == dst else [rank])]
        self.mine = self.rows[rank] if rank == dst else self.rows[0]
        most = (height + STRIP_ROWS - 1) // STRIP_ROWS            # strips in the frame
        most = ((most + world - 1) // world) * STRIP_ROWS          # rows of the rank with the most strips: gather wants equal shapes
        self.pack = torch.zeros((most, width * 3), dtype=torch.float32, device=device)
        self.recv = [torch.empty_like(self.pack) for _ in range(world)] if rank == dst else None

    def __call__(self, fb, dist):
        flat = fb.view(fb.shape[0], -1)
        n = self.mine.numel()
        if n:
            self.torch.index_select(flat, 0, self.mine, out=self.pack[:n])
        dist.gather(self.pack, self.recv, dst=self.dst)
        if self.rank == self.dst:
            for r in range(self.world):
                if r != self.dst and self.rows[r].numel():
                    flat.index_copy_(0, self.rows[r], self.recv[r][:self.rows[r].numel()])
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
            # (the gather's index / pack tensors too: they are filled on the stream they are used on — ADVICE r03: built on the default stream,
            # their zero-fill could race with the first index_select on this one)
            self.gather = StripGather(torch, self.width, self.height, rank, world, self.device) if world > 1 else None
        self.tiles = owned_tiles(self.width, self.height, tile[0], tile[1], order, rank, world)

    def render(self, samples, bounces, clear=True):
        """Enqueue this rank's tiles (asynchronous on the stream)."""
        with self.torch.cuda.stream(self.stream):
            if clear:
                self.fb.zero_()
            if self.tiles:
                self.ctx.render_tiles(self.fb.data_ptr(), self.width, self.height, samples, bounces, self.tiles)

    def reduce(self, dist=None, how="gather"):
        """Assemble the frame on rank 0: how = "gather" (the owned strips only, 1 / world of the bytes) or "reduce" (one reduce(SUM) of the whole buffer)."""
        with self.torch.cuda.stream(self.stream):
            if self.world > 1 and how == "gather":
                return self.gather(self.fb, dist)
            return reduce_frame(self.fb, self.world, dist)

    def close(self):
        self.ctx.close()

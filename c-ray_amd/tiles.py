"""tiles.py — host mirror of the reference's image quantisation (src/datatypes/tile.c:66-117) and tile
orderings (tile.c:119-241), so that tests, the bench and the cluster-worker test derive the SAME ordered tile list the reference's
`nextTile()` would hand out. (Multi-GPU shares are NOT dealt from this list: render.py: owned_tiles gives rank r the
4-row strips i = r mod world — tile orders such as "from middle" balance badly when dealt round-robin)
Tiles are (x0, y0, x1, y1) in reference coordinates (y counts from the bottom of the image).
"""
import math

ORDER_TOP_TO_BOTTOM, ORDER_FROM_MIDDLE, ORDER_TO_MIDDLE, ORDER_NORMAL, ORDER_RANDOM = 0, 1, 2, 3, 4   # tile.h:15-21

_MASK64 = (1 << 64) - 1


class _Pcg32:
    """pcg_basic.c:42-68 (tile.c:151 seeds it with 3141592, stream 0)."""

    def __init__(self, seed, seq):
        self.state, self.inc = 0, ((seq << 1) | 1) & _MASK64
        self.next()
        self.state = (self.state + seed) & _MASK64
        self.next()

    def next(self):
        old = self.state
        self.state = (old * 6364136223846793005 + self.inc) & _MASK64
        xs = (((old >> 18) ^ old) >> 27) & 0xFFFFFFFF
        rot = old >> 59
        return ((xs >> rot) | (xs << ((-rot) & 31))) & 0xFFFFFFFF


def _rand_interval(lo, hi, rng):          # tile.c:132-147
    rng_range = 1 + hi - lo
    buckets = 0xFFFFFFFF // rng_range
    limit = buckets * rng_range
    while True:
        r = rng.next()
        if r < limit:
            return lo + r // buckets


def quantize_image(width, height, tile_width, tile_height, order=ORDER_FROM_MIDDLE):
    if tile_width >= width:
        tile_width = width
    if tile_height >= height:
        tile_height = height
    tile_width = max(tile_width, 1)
    tile_height = max(tile_height, 1)
    tiles_x = width // tile_width + (1 if width % tile_width else 0)
    tiles_y = height // tile_height + (1 if height % tile_height else 0)
    tiles = []
    for y in range(tiles_y):
        for x in range(tiles_x):
            tiles.append((x * tile_width, y * tile_height, min((x + 1) * tile_width, width), min((y + 1) * tile_height, height)))
    n = len(tiles)
    if order == ORDER_FROM_MIDDLE:         # tile.c:163-186 (ceil(tileCount / 2) is an integer division)
        right = int(math.ceil(n // 2))
        left = right - 1
        out, is_right = [], True
        for _ in range(n):
            if is_right:
                out.append(tiles[right]); right += 1
            else:
                out.append(tiles[left]); left -= 1
            is_right = not is_right
        tiles = out
    elif order == ORDER_TO_MIDDLE:         # tile.c:188-211
        left, right, out, is_right = 0, n - 1, [], True
        for _ in range(n):
            if is_right:
                out.append(tiles[right]); right -= 1
            else:
                out.append(tiles[left]); left += 1
            is_right = not is_right
        tiles = out
    elif order == ORDER_TOP_TO_BOTTOM:     # tile.c:119-130
        tiles = tiles[::-1]
    elif order == ORDER_RANDOM:            # tile.c:149-161
        rng = _Pcg32(3141592, 0)
        for i in range(n):
            j = _rand_interval(0, n - 1, rng)
            tiles[i], tiles[j] = tiles[j], tiles[i]
    return tiles


def tiles_for_rank(tiles, rank, world):
    """Interleaved ownership: rank g of G takes tiles i = g (mod G) of the ordered list."""
    return tiles[rank::world]

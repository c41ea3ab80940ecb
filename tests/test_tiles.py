"""tiles.py mirrors src/datatypes/tile.c:66-241 (quantizeImage + the five orderings)."""
import pytest


def test_quantize_covers_image_once(pkg):
    T = pkg.tiles
    for (w, h, tw, th) in [(1280, 720, 64, 64), (320, 200, 64, 64), (100, 37, 32, 16), (5, 5, 64, 64), (7, 3, 1, 1)]:
        for order in (T.ORDER_TOP_TO_BOTTOM, T.ORDER_FROM_MIDDLE, T.ORDER_TO_MIDDLE, T.ORDER_NORMAL, T.ORDER_RANDOM):
            tiles = T.quantize_image(w, h, tw, th, order)
            seen = set()
            for (x0, y0, x1, y1) in tiles:
                assert 0 <= x0 < x1 <= w and 0 <= y0 < y1 <= h
                for y in range(y0, y1):
                    for x in range(x0, x1):
                        assert (x, y) not in seen
                        seen.add((x, y))
            assert len(seen) == w * h


def test_orderings_known_answers(pkg):
    T = pkg.tiles
    base = T.quantize_image(4, 1, 1, 1, T.ORDER_NORMAL)
    xs = lambda tiles: [t[0] for t in tiles]
    assert xs(base) == [0, 1, 2, 3]
    assert xs(T.quantize_image(4, 1, 1, 1, T.ORDER_TOP_TO_BOTTOM)) == [3, 2, 1, 0]
    assert xs(T.quantize_image(4, 1, 1, 1, T.ORDER_FROM_MIDDLE)) == [2, 1, 3, 0]      # tile.c:163-186
    assert xs(T.quantize_image(5, 1, 1, 1, T.ORDER_FROM_MIDDLE)) == [2, 1, 3, 0, 4]
    assert xs(T.quantize_image(4, 1, 1, 1, T.ORDER_TO_MIDDLE)) == [3, 0, 2, 1]        # tile.c:188-211
    r = xs(T.quantize_image(16, 1, 1, 1, T.ORDER_RANDOM))
    assert sorted(r) == list(range(16)) and r != list(range(16))
    assert r == xs(T.quantize_image(16, 1, 1, 1, T.ORDER_RANDOM))                      # fixed seed 3141592


def test_rank_ownership_partitions(pkg):
    T = pkg.tiles
    tiles = T.quantize_image(1280, 720, 64, 64, T.ORDER_FROM_MIDDLE)
    assert len(tiles) == 240
    for world in (1, 2, 4, 8):
        parts = [T.tiles_for_rank(tiles, r, world) for r in range(world)]
        assert sorted(sum(parts, [])) == sorted(tiles)
        assert max(len(p) for p in parts) - min(len(p) for p in parts) <= 1

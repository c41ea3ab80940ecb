"""CPU tier for the HOST logic that cuts a dispatch into work units (cray_hip.hip: planWork, through crh_debug_plan_units — no device
needed): every pass of every pixel of every tile belongs to exactly one unit, units are handed out in list order with the small ones last, block
shapes respect the tiles, a block unit never holds too few paths to fill a wave's path table, and — with CRH_OPT_TAIL_SPLIT (off by default; the plan's debug
entry takes the process default from CRH_TAIL_SPLIT like a context does) — the queue ends with the rolling kernel's 64-path units: small blocks, or pass
segments of single pixels."""
import numpy as np
import pytest


def cover(units, width, height, passes=None):
    """How often a pixel is covered — counted in passes and divided by the dispatch's (a split pixel's segments add up to 1)."""
    img = np.zeros((height, width), np.int64)
    passes = passes or int(units[:, 7].max())
    for x0, y0, x1, y1, _, _, _, n in units:
        img[y0:y1, x0:x1] += n
    assert (img % passes == 0).all()
    return img // passes


@pytest.fixture
def tail_split(monkeypatch):
    monkeypatch.setenv("CRH_TAIL_SPLIT", "4")


def test_the_default_plan_has_no_split_units(pkg):
    units, chunk = pkg.api.plan_units(1280, 720, 256, [(0, 0, 1280, 720)])
    assert (cover(units, 1280, 720) == 1).all() and units[:, 5].max() == 2 and (units[:, 6] == 0).all() and (units[:, 7] == 256).all() and chunk == 256
    assert set(units[units[:, 5] == 2, 4]) == {1}


def test_full_frame_region_at_256_spp(pkg, tail_split):
    units, chunk = pkg.api.plan_units(1280, 720, 256, [(0, 0, 1280, 720)])
    assert (cover(units, 1280, 720) == 1).all()
    area, level = units[:, 4], units[:, 5]
    assert (np.diff(level) >= 0).all(), "regular blocks first, then quarter blocks, then sixteenth blocks, then the 64-path units"
    assert set(area[level == 0]) == {8} and set(area[level == 1]) == {2} and set(area[level == 2]) == {1}    # 8 px x 256 spp = 2048 paths
    px = (units[:, 2] - units[:, 0]) * (units[:, 3] - units[:, 1])
    blocks = level < 3
    assert 0.10 < px[(level >= 1) & blocks].sum() / (1280 * 720) < 0.22 and 0.02 < px[level == 2].sum() / (1280 * 720) < 0.07
    assert (units[blocks, 6] == 0).all() and (units[blocks, 7] == 256).all()
    # the very end: single pixels in four segments of 64 passes, a pixel's segments next to each other and in pass order; about four units per wave
    split = units[level == 3]
    assert len(split) % 4 == 0 and 3 * 4096 <= len(split) <= 6 * 4096
    assert (split[:, 4] == 1).all() and (split[:, 7] == 64).all() and (split[:, 6].reshape(-1, 4) == [0, 64, 128, 192]).all()
    assert (split[:, :4].reshape(-1, 4, 4) == split[::4, None, :4]).all()
    assert chunk == 256
    # bottom-up, row by row: the first unit sits at the frame's origin
    assert tuple(units[0, :2]) == (0, 0)


def test_units_never_hold_fewer_paths_than_fill_a_wave(pkg, tail_split):
    """8 passes at 4K: 128-pixel blocks (every wave gets >= 8 units); the tail's blocks stop at 64 px (512 paths) and 32 px (256 paths),
    not at a quarter (32 px) and a sixteenth (8 px = 64 paths) of the block."""
    units, chunk = pkg.api.plan_units(3840, 2160, 8, [(0, 0, 3840, 2160)])
    assert (cover(units, 3840, 2160) == 1).all()
    area, level = units[:, 4], units[:, 5]
    assert set(area[level == 0]) == {128} and set(area[level == 1]) == {64} and set(area[level == 2]) == {32}
    assert set(area[level == 3]) == {8} and (units[:, 7] == 8).all()         # the rolling kernel's last units: 8 px x 8 passes, no pixel is split
    assert 3 * 4096 <= (level == 3).sum() <= 6 * 4096
    assert chunk == 8


def test_blocks_are_never_taller_than_the_strips_of_a_multi_gpu_share(pkg):
    strips = pkg.render.owned_tiles(3840, 2160, 64, 64, pkg.tiles.ORDER_FROM_MIDDLE, 3, 8)
    units, _ = pkg.api.plan_units(3840, 2160, 8, strips)
    img = cover(units, 3840, 2160)
    own = np.zeros_like(img)
    for x0, y0, x1, y1 in strips:
        own[y0:y1, x0:x1] = 1
    assert np.array_equal(img, own)
    assert (units[:, 3] - units[:, 1]).max() <= 4
    full = (units[:, 2] - units[:, 0]) * (units[:, 3] - units[:, 1]) == units[:, 4]
    assert full.mean() > 0.99, "no padding: a block's area is all pixels (only ragged right edges clip)"


@pytest.mark.parametrize("order", ["ORDER_FROM_MIDDLE", "ORDER_TOP_TO_BOTTOM", "ORDER_RANDOM"])
def test_the_reference_tile_lists_are_covered_once_in_list_order(order, pkg):
    w, h = 1280, 720
    tiles = pkg.tiles.quantize_image(w, h, 64, 64, getattr(pkg.tiles, order))
    units, _ = pkg.api.plan_units(w, h, 256, tiles)
    assert (cover(units, w, h) == 1).all()
    # every unit lies inside one tile, and the tiles are visited in list order
    def tile_of(u):
        for i, (x0, y0, x1, y1) in enumerate(tiles):
            if x0 <= u[0] and u[2] <= x1 and y0 <= u[1] and u[3] <= y1:
                return i
        return -1
    idx = np.array([tile_of(u) for u in units[:: max(1, len(units) // 4000)]])
    assert (idx >= 0).all() and (np.diff(idx) >= 0).all()


def test_small_dispatches_shrink_their_blocks_to_keep_every_wave_fed(pkg, tail_split):
    units, chunk = pkg.api.plan_units(33, 7, 1, [(0, 0, 33, 7)])
    assert (cover(units, 33, 7) == 1).all()
    assert len(units) == 33 * 7 and chunk == 1                      # 231 pixels for 4096 waves: single pixels
    units, _ = pkg.api.plan_units(1280, 720, 256, [(0, y, 1280, y + 4) for y in range(0, 720, 32)])      # an eighth of the frame
    assert units[:, 4].max() <= 2
    img = cover(units, 1280, 720)
    assert all((img[y:y + 4] == 1).all() for y in range(0, 720, 32)) and img.sum() == 23 * 4 * 1280
    assert 0.03 < ((units[:, 5] == 3).sum() / 4) / (23 * 4 * 1280) <= 0.25, "split pixels: a few units per wave, at most a quarter of the share"


def test_bad_dispatches_are_refused(pkg):
    with pytest.raises(pkg.api.CrhError):
        pkg.api.plan_units(64, 64, 4, [(0, 0, 65, 64)])
    units, _ = pkg.api.plan_units(64, 64, 4, [])
    assert len(units) == 0

"""The tile dealer behind the C-ABI (yart_tiles_*) against the Python restatement of
Renderer::render's enumeration (yart_amd/shard.py, src/renderer.cpp:161-217): same tiles in the same
order, the same pixel order per rank, and every pixel of the frame owned by exactly one rank.  CPU only;
the RCCL gather over these lists is tests/test_multigpu_capi.py (GPU)."""
import numpy as np
import pytest

from yart_amd import api, shard

SIZES = [(1024, 768), (64, 64), (200, 120), (33, 31), (31, 33), (1, 1), (3840, 2160), (100, 32)]


@pytest.mark.parametrize("width,height", SIZES)
def test_tile_list_matches_reference_enumeration(width, height):
    assert np.array_equal(api.tiles_list(width, height), shard.tile_list(width, height))


@pytest.mark.parametrize("width,height", SIZES[:6])
@pytest.mark.parametrize("world", [1, 2, 3, 8])
def test_rank_pixels_partition_the_frame(width, height, world):
    seen = np.zeros(width * height, np.int32)
    for rank in range(world):
        got = api.tiles_rank_pixels(width, height, rank, world)
        want = shard.rank_pixels(width, height, rank, world)
        assert np.array_equal(got.astype(np.int64), want)
        seen[got] += 1
    assert (seen == 1).all()


def test_bad_arguments_are_counts_of_zero():
    L = api.lib()
    assert L.yart_tiles_count(0, 10, 32) == 0
    assert L.yart_tiles_rank_pixels(10, 10, 32, 2, 2, None) == 0      # rank out of range
    assert L.yart_tiles_list(10, 10, 32, None) == 1

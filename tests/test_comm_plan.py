"""CPU: the arithmetic of the all-gatherv exchange (ii2_gatherv_offsets, csrc/comm.cpp) — where each rank's
contribution lands in the rank-order concatenation (inverted_index.go:330-339) and the capacity decision every
rank takes identically.  Pure host code of the product library: no GPU is touched."""
import ctypes as C

import numpy as np
import pytest

from inverted_index_2_amd import _lib


def _offsets(counts, cap):
    lib = _lib.load()
    c = (C.c_uint64 * len(counts))(*counts)
    off = (C.c_uint64 * (len(counts) + 1))()
    rc = lib.ii2_gatherv_offsets(c, len(counts), cap, off)
    return rc, list(off)


@pytest.mark.parametrize("world", [1, 2, 8, 64])
def test_offsets_are_the_exclusive_prefix(world):
    rng = np.random.default_rng(world)
    counts = [int(x) for x in rng.integers(0, 10_000_000, world)]
    rc, off = _offsets(counts, sum(counts))
    assert rc == 0
    assert off == [0] + list(np.cumsum(counts))


def test_empty_ranks_take_no_room():
    rc, off = _offsets([0, 5, 0, 0, 7, 0, 0, 0], 12)
    assert rc == 0 and off == [0, 0, 5, 5, 5, 12, 12, 12, 12]
    rc, off = _offsets([0, 0], 0)
    assert rc == 0 and off == [0, 0, 0]


def test_capacity_is_checked_on_the_total():
    rc, off = _offsets([4, 4], 7)
    assert rc == -4 and off == [0, 4, 8]          # II2_ECAPACITY, offsets still filled
    rc, _ = _offsets([4, 4], 8)
    assert rc == 0
    rc, _ = _offsets([2**63, 2**63], 2**64 - 1)   # the sum wraps: refused, not truncated
    assert rc == -4


def test_world_bounds():
    lib = _lib.load()
    off = (C.c_uint64 * 70)()
    c = (C.c_uint64 * 70)()
    assert lib.ii2_gatherv_offsets(c, 0, 10, off) == -1
    assert lib.ii2_gatherv_offsets(c, 65, 10, off) == -1
    assert lib.ii2_gatherv_offsets(None, 2, 10, off) == -1


def test_segment_gather_plan_is_three_exclusive_prefixes():
    # ii2_seg_allgather: where every rank's lists, blocks and payload bytes land in the concatenated segment
    from inverted_index_2_amd.engine import seg_gather_plan
    rng = np.random.default_rng(11)
    for world in (1, 2, 8, 64):
        shapes = [(int(a), int(b), int(c)) for a, b, c in zip(rng.integers(0, 200_000, world), rng.integers(0, 900_000, world), rng.integers(0, 40_000_000, world))]
        rc, lo, bo, qo = seg_gather_plan(shapes)
        assert rc == 0
        assert lo == [0] + list(np.cumsum([s[0] for s in shapes]))
        assert bo == [0] + list(np.cumsum([s[1] for s in shapes]))
        assert qo == [0] + list(np.cumsum([s[2] for s in shapes]))
    rc, lo, bo, qo = seg_gather_plan([(0, 0, 0), (3, 2, 10), (0, 0, 0)])          # empty ranks take no room
    assert rc == 0 and lo == [0, 0, 3, 3] and bo == [0, 0, 2, 2] and qo == [0, 0, 10, 10]


def test_segment_gather_plan_refuses_what_one_segment_cannot_hold():
    from inverted_index_2_amd.engine import seg_gather_plan
    assert seg_gather_plan([(10, 10, 3 << 30), (10, 10, 2 << 30)])[0] == -5        # II2_ERANGE: >= 4 GiB of payload
    assert seg_gather_plan([(1 << 30, 1, 1), (1 << 30, 1, 1)])[0] == -5            # 2^31 lists
    assert seg_gather_plan([(1, 1 << 30, 1), (1, 1 << 30, 1)])[0] == -5            # 2^31 blocks
    assert seg_gather_plan([])[0] == -1                                            # world 0

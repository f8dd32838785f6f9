"""GPU: intersection kernels against the oracle (bit-exact id sequences) — through the C ABI."""
import numpy as np
import pytest

from inverted_index_2_amd import synth
from oracle import oracle as orc
from tests.gpu_util import ctx, sorted_unique  # noqa: F401

pytestmark = pytest.mark.gpu


def _check(ctx, lists, removed=None):
    seg = ctx.encode_lists(lists)
    tomb = ctx.tombstones(removed) if removed is not None else None
    want = orc.intersect(lists, np.sort(removed) if removed is not None else ())
    for bitmap in (1, 0):      # very dense tiles: per-list bitmaps (default) vs the byte map
        ctx.set_option("intersect.bitmap", bitmap)
        out, n = ctx.intersect([(seg, i) for i in range(len(lists))], tomb=tomb)
        got = out.download(n)
        assert n == want.size, (bitmap, n, want.size)
        assert np.array_equal(got, want), bitmap
    ctx.set_option("intersect.bitmap", 1)
    # split over two segments as well (lists from different segments)
    if len(lists) >= 2:
        s0, s1 = ctx.encode_lists(lists[:1]), ctx.encode_lists(lists[1:])
        out, n = ctx.intersect([(s0, 0)] + [(s1, i) for i in range(len(lists) - 1)], tomb=tomb)
        assert np.array_equal(out.download(n), want)


def test_edge_cases(ctx):
    e = np.empty(0, np.uint32)
    _check(ctx, [e, e])
    _check(ctx, [np.asarray([5], np.uint32), e])
    _check(ctx, [np.asarray([5], np.uint32), np.asarray([5], np.uint32)])
    _check(ctx, [np.asarray([5], np.uint32), np.asarray([6], np.uint32)])
    _check(ctx, [np.asarray([0, 0xFFFFFFFF], np.uint32), np.asarray([0xFFFFFFFF], np.uint32)])
    _check(ctx, [np.asarray([0, 0xFFFFFFFF], np.uint32), np.asarray([0, 7, 0xFFFFFFFF], np.uint32)])
    a = np.arange(0, 3000, 2, dtype=np.uint32)
    b = np.arange(1, 3000, 2, dtype=np.uint32)
    _check(ctx, [a, b])                       # disjoint, interleaved
    _check(ctx, [a, a])                       # identical
    _check(ctx, [a, a[::7].copy()])           # subset
    _check(ctx, [a, (a + 100000).astype(np.uint32)])   # disjoint ranges


@pytest.mark.parametrize("n", [255, 256, 257, 511, 512, 513, 2047, 2048, 2049, 4097])
def test_block_boundary_lengths(ctx, n):
    rng = np.random.default_rng(n)
    a = sorted_unique(rng, n, 4 * n + 10)
    b = sorted_unique(rng, n + 3, 4 * n + 10)
    _check(ctx, [a, b])
    _check(ctx, [a, b], removed=rng.integers(0, 4 * n + 10, n // 3 + 1).astype(np.uint32))


@pytest.mark.parametrize("seed", range(6))
def test_random_dense(ctx, seed):
    rng = np.random.default_rng(100 + seed)
    U = int(rng.integers(10_000, 400_000))
    k = int(rng.integers(2, 6))
    lists = [sorted_unique(rng, int(rng.integers(U // 8, U // 2)), U) for _ in range(k)]
    _check(ctx, lists)
    _check(ctx, lists, removed=rng.integers(0, U, U // 50).astype(np.uint32))


@pytest.mark.parametrize("seed", range(6))
def test_random_sparse_and_skewed(ctx, seed):
    # sparse lists (gaps >> 64) leave the byte-map path; skew forces galloping over skip tables
    rng = np.random.default_rng(200 + seed)
    U = 1 << 31
    core = sorted_unique(rng, 300, U)
    sizes = [500, 5_000, 60_000, 400_000][: 2 + seed % 3]
    lists = [np.union1d(sorted_unique(rng, s, U), core).astype(np.uint32) for s in sizes]
    _check(ctx, lists)
    _check(ctx, lists[::-1], removed=core[::3].copy())


def test_dense_vs_sparse_mix(ctx):
    rng = np.random.default_rng(7)
    dense = sorted_unique(rng, 900_000, 2_000_000)
    sparse = sorted_unique(rng, 700, 2_000_000)
    mid = sorted_unique(rng, 40_000, 2_000_000)
    _check(ctx, [dense, sparse])
    _check(ctx, [dense, mid, sparse])
    _check(ctx, [mid, dense])


def test_eight_terms_zipf(ctx):
    # BASELINE config 5 in miniature: ranks {2,4,...,16384}, common core forced into every list
    D = 20_000_000
    rng = np.random.default_rng(8)
    core = sorted_unique(rng, 2000, D)
    lists = [np.union1d(synth.zipf_list(r, D), core).astype(np.uint32) for r in (2, 4, 16, 64, 256, 1024, 4096, 16384)]
    _check(ctx, lists)


def test_forced_tile_heights(ctx):
    rng = np.random.default_rng(9)
    a = sorted_unique(rng, 50_000, 160_000)
    b = sorted_unique(rng, 70_000, 160_000)
    seg = ctx.encode_lists([a, b])
    want = orc.intersect([a, b])
    for g in (1, 2, 3, 8):
        ctx.set_option("intersect.g", g)
        out, n = ctx.intersect([(seg, 0), (seg, 1)])
        assert np.array_equal(out.download(n), want), g
    ctx.set_option("intersect.g", 0)


def test_config2_scaled_properties(ctx):
    # BASELINE config 2 at 1/10 scale vs the oracle, full scale is covered by bench.py's checksum
    D = 10_000_000
    a, b = synth.zipf_list(2, D), synth.zipf_list(3, D)
    _check(ctx, [a, b])


def _gap_list(rng, n, gaps, probs, start=0):
    g = rng.choice(np.asarray(gaps, np.int64), size=n, p=probs)
    return (start + np.cumsum(g)).astype(np.uint32)


@pytest.mark.parametrize("k", [2, 3, 5, 7, 8])
def test_bitmap_mode_very_dense(ctx, k):
    """Lists dense enough for the bitmap tile path (full blocks of one-byte gaps, <= ~3 docs per posting);
    7 lists is the most a bitmap tile holds, 8 falls back to the byte map."""
    rng = np.random.default_rng(900 + k)
    lists = [_gap_list(rng, 300_000, [1, 2, 3, 4, 5], [0.35, 0.3, 0.2, 0.1, 0.05], start=int(rng.integers(0, 50))) for _ in range(k)]
    _check(ctx, lists)
    _check(ctx, lists, removed=rng.integers(0, 700_000, 20_000).astype(np.uint32))


def test_bitmap_mode_gap_outliers(ctx):
    """Dense lists with outliers: gaps of 31/32 (the 32-bit mask limit of a group of four), runs of 127s that push a
    lane past its window, and a partial last block — the slow lanes and the generic block decoder of the bitmap path."""
    rng = np.random.default_rng(77)
    a = _gap_list(rng, 400_000, [1, 2, 7, 8, 9, 10, 31, 32, 127], [0.5, 0.3, 0.04, 0.04, 0.04, 0.04, 0.02, 0.01, 0.01])
    b = _gap_list(rng, 500_000, [1, 2, 3, 29, 30, 33, 100], [0.45, 0.3, 0.2, 0.02, 0.01, 0.01, 0.01], start=3)
    c = _gap_list(rng, 350_123, [1, 2, 3, 127], [0.4, 0.35, 0.2, 0.05], start=11)
    _check(ctx, [a, b])
    _check(ctx, [a, b, c], removed=a[::5].copy())
    # a long stretch of 127-gaps inside an otherwise dense list: lanes wholly outside the other list's tile ranges
    d = np.concatenate([np.arange(1, 200_000, 2), 200_000 + 127 * np.arange(1, 3000), 600_000 + np.arange(0, 200_000, 3)]).astype(np.uint32)
    _check(ctx, [d, np.arange(0, 800_001, 1, dtype=np.uint32)])
    _check(ctx, [np.arange(0, 800_001, 2, dtype=np.uint32), d])


def test_sixty_four_lists(ctx):
    """II2_MAX_LISTS terms in one conjunction (258 descriptor words per tile: more than one per thread)."""
    rng = np.random.default_rng(64)
    core = sorted_unique(rng, 5000, 300_000)
    lists = [np.union1d(core, sorted_unique(rng, int(rng.integers(20_000, 60_000)), 300_000)).astype(np.uint32) for _ in range(64)]
    _check(ctx, lists)
    _check(ctx, lists[:63], removed=core[::7].copy())


@pytest.mark.parametrize("seed", range(4))
def test_bitmap_mode_lane_jumping_into_range(ctx, seed):
    """Regression (found by scripts/stress.py): in bitmap mode a lane of a non-driver list that starts more than the
    128-doc guard below a tile's range but reaches into it through a wide gap (one-byte gaps up to 127, so 16 postings
    can span 2000 docs) must still mark its postings.  Dense lists with a sprinkling of wide gaps, many tiles."""
    rng = np.random.default_rng(4100 + seed)
    # ~2.9 docs per posting (dense enough for bitmap tiles), 1.5 % gaps of 110..127: ~3 % of the lanes span > 128 docs
    gaps = [1, 2, 110, 127]
    probs = [.6, .385, .0075, .0075]
    lists = [_gap_list(rng, 1_000_000, gaps, probs, start=int(rng.integers(0, 1000))) for _ in range(2 + seed % 2)]
    _check(ctx, lists)
    _check(ctx, lists, removed=lists[0][::11].copy())


@pytest.mark.parametrize("n_rare", [3, 200, 256, 257, 700])
def test_rare_term_against_long_lists(ctx, n_rare):
    """A tiny sparse driver (one to three blocks) against long lists: its blocks are split over several tiles so that
    more than a handful of workgroups decode the long lists' blocks (sub-block tiles, gallop path)."""
    rng = np.random.default_rng(7000 + n_rare)
    U = 40_000_000
    long1 = _gap_list(rng, 2_000_000, [1, 5, 20, 60], [.25, .25, .25, .25])
    long2 = sorted_unique(rng, 900_000, U)
    rare = np.union1d(sorted_unique(rng, n_rare, U), rng.choice(np.intersect1d(long1, long2), min(n_rare, 40), replace=False)).astype(np.uint32)
    _check(ctx, [rare, long1])
    _check(ctx, [long2, rare, long1], removed=rare[::3].copy())


@pytest.mark.parametrize("n_lists", [4, 6, 8])
def test_rare_term_against_many_long_lists_all_at_once(ctx, n_lists):
    """Four or more lists with a tiny sparse driver (split driver blocks): after the second-shortest list, every (surviving
    candidate, longer list) pair is tested in ONE stage (intersect.hip, `combine`).  Built so that the stage has work that the
    list-after-list form would have skipped: the second list holds every driver id (nothing is thinned out), the longer lists
    each drop a different part of them, some ids survive everything, and one list is so short in the tile's range that a
    candidate finds no block."""
    rng = np.random.default_rng(900 + n_lists)
    U = 60_000_000
    rare = sorted_unique(rng, 1500, U)
    second = np.union1d(rare, sorted_unique(rng, 20_000, U)).astype(np.uint32)             # a superset of the driver
    longs = []
    for i in range(n_lists - 2):
        keep = rare[rng.random(rare.size) < 0.7]                                            # each longer list keeps ~70 % of the driver ...
        keep = np.union1d(keep, rare[::5])                                                  # ... and all of them every fifth id
        base = sorted_unique(rng, 300_000 * (i + 1), U) if i else _gap_list(rng, 1_500_000, [1, 5, 20, 60], [.25, .25, .25, .25])
        longs.append(np.union1d(base, keep).astype(np.uint32))
    longs[-1] = longs[-1][longs[-1] >= rare[40]]                                            # no block of this list before the 40th candidate
    lists = [rare, second] + longs
    _check(ctx, lists)
    _check(ctx, lists[::-1], removed=rare[::15].copy())
    got = orc.intersect(lists)
    assert 100 < got.size < rare.size

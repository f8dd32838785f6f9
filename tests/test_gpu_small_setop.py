"""GPU: the single-workgroup kernel for small queries (csrc/setop_small.hip: AND / OR of lists holding <= 8192
postings together) against numpy, with the general paths (option setop.small = 0) as a second opinion."""
import numpy as np
import pytest

from tests.gpu_util import ctx  # noqa: F401

pytestmark = pytest.mark.gpu


def _check(ctx, lists, removed=None):
    lists = [np.asarray(l, dtype=np.uint32) for l in lists]
    want_and = lists[0]
    for x in lists[1:]:
        want_and = np.intersect1d(want_and, x, assume_unique=True)
    want_or = np.unique(np.concatenate(lists)) if lists else np.empty(0, np.uint32)
    if removed is not None:
        want_and = np.setdiff1d(want_and, removed, assume_unique=True)
        want_or = np.setdiff1d(want_or, removed, assume_unique=True)
    tomb = ctx.tombstones(removed) if removed is not None else None
    seg = ctx.encode_lists(lists)
    ls = [(seg, i) for i in range(len(lists))]
    out = ctx.empty(sum(l.size for l in lists) + 8)
    for small in (1, 0):
        ctx.set_option("setop.small", small)
        _, n = ctx.intersect(ls, tomb=tomb, out=out)
        assert n == want_and.size and np.array_equal(out.download(n), want_and.astype(np.uint32)), ("and", small)
        _, n = ctx.union(ls, tomb=tomb, out=out)
        assert n == want_or.size and np.array_equal(out.download(n), want_or.astype(np.uint32)), ("or", small)
    ctx.set_option("setop.small", 1)
    seg.free()
    out.free()


def test_tiny_and_edge_lists(ctx):
    _check(ctx, [[5], [5]])
    _check(ctx, [[5], [6]])
    _check(ctx, [[0, 1, 2, 0xFFFFFFFF], [0, 2, 0xFFFFFFFE, 0xFFFFFFFF]])
    _check(ctx, [[0xFFFFFFFF], [0xFFFFFFFF], [0xFFFFFFFF]])
    _check(ctx, [[1, 2, 3]])                                                     # one list: itself
    _check(ctx, [[1, 2, 3], [1, 2, 3]])                                          # the same list twice
    _check(ctx, [np.arange(0, 600, 3), np.arange(0, 600, 2), np.arange(0, 600, 5)], removed=np.array([0, 30, 60, 599], np.uint32))
    _check(ctx, [[7, 9], [], [9]]) if False else None                            # (an empty list: see below)


def test_empty_lists_in_small_queries(ctx):
    a = np.array([7, 9, 11], np.uint32)
    seg = ctx.encode_lists([a, np.empty(0, np.uint32), np.array([9], np.uint32)])
    out, n = ctx.intersect([(seg, 0), (seg, 1), (seg, 2)])
    assert n == 0
    out, n = ctx.union([(seg, 0), (seg, 1), (seg, 2)])
    assert np.array_equal(out.download(n), a)
    out, n = ctx.union([(seg, 1)])
    assert n == 0


@pytest.mark.parametrize("seed", range(6))
def test_random_small_queries(ctx, seed):
    rng = np.random.default_rng(700 + seed)
    for _ in range(12):
        k = int(rng.choice([1, 2, 3, 5, 12, 31, 32, 40, 64]))
        universe = int(rng.choice([50, 5_000, 1_000_000, (1 << 32) - 1]))
        budget = int(rng.choice([30, 800, 8192, 9000]))                          # 9000: over the kernel's limit -> general paths
        sizes = rng.multinomial(min(budget, universe * k // 2 + 1), np.ones(k) / k)
        lists = [np.unique(rng.integers(0, universe, int(s) + (1 if rng.random() < 0.8 else 0), dtype=np.uint64)).astype(np.uint32) for s in sizes]
        lists = [l for l in lists if l.size] or [np.array([1], np.uint32)]
        if rng.random() < 0.5 and len(lists) > 1:                                # make the intersection non-trivial
            core = lists[0][:: max(1, lists[0].size // 7)]
            lists = [np.union1d(l, core).astype(np.uint32) for l in lists]
        removed = np.unique(rng.integers(0, universe, 20, dtype=np.uint64)).astype(np.uint32) if rng.random() < 0.5 else None
        _check(ctx, lists, removed)


def test_block_boundaries_and_capacity(ctx):
    from inverted_index_2_amd import II2Error
    a = np.arange(0, 256 * 16, dtype=np.uint32)                                  # 16 full blocks
    b = np.arange(0, 256 * 16, dtype=np.uint32) * 2                              # 16 full blocks, multi-byte gaps at the end? no: gap 2
    _check(ctx, [a, b])                                                          # exactly 8192 postings: the kernel's limit
    _check(ctx, [a, b, np.array([3], np.uint32)])                                # 8193: general paths
    # many short lists: up to 64 lists / 128 blocks as long as the postings fit
    rng = np.random.default_rng(9)
    _check(ctx, [np.unique(rng.integers(0, 100_000, 120, dtype=np.uint64)).astype(np.uint32) for _ in range(64)])
    _check(ctx, [np.unique(rng.integers(0, 3_000, 300, dtype=np.uint64)).astype(np.uint32) for _ in range(27)], removed=np.arange(0, 3000, 7, dtype=np.uint32))
    _check(ctx, [np.arange(i, 2000, 17, dtype=np.uint32) for i in range(64)])
    seg = ctx.encode_lists([a[:1000], b[:1000]])
    small = ctx.empty(10)
    with pytest.raises(II2Error):                                                # union larger than the output buffer
        ctx.union([(seg, 0), (seg, 1)], out=small)


# ---- unions of a few medium-size lists by ranking (csrc/union_rank.hip) ----
@pytest.mark.parametrize("seed", range(5))
def test_union_by_ranking(ctx, seed):
    rng = np.random.default_rng(900 + seed)
    for _ in range(4):
        k = int(rng.integers(1, 9))
        universe = int(rng.choice([20_000, 3_000_000, (1 << 32) - 1]))
        lists = []
        for _ in range(k):
            n = int(rng.choice([1, 300, 9_000, 60_000, 250_000]))
            l = np.unique(rng.integers(0, universe, min(n, universe), dtype=np.uint64)).astype(np.uint32)
            if rng.random() < 0.3 and lists:                                   # heavy overlap with an earlier list
                l = np.union1d(l, lists[0][::3]).astype(np.uint32)
            lists.append(l)
        if rng.random() < 0.3:
            lists.append(lists[0].copy())                                      # the same ids twice
        lists = lists[:8]
        removed = np.unique(rng.integers(0, universe, 3000, dtype=np.uint64)).astype(np.uint32) if rng.random() < 0.5 else None
        want = np.unique(np.concatenate(lists))
        if removed is not None:
            want = np.setdiff1d(want, removed, assume_unique=True)
        tomb = ctx.tombstones(removed) if removed is not None else None
        seg = ctx.encode_lists(lists)
        ls = [(seg, i) for i in range(len(lists))]
        out = ctx.empty(sum(l.size for l in lists) + 8)
        for rank in (1, 0):
            ctx.set_option("union.rank", rank)
            _, n = ctx.union(ls, tomb=tomb, out=out)
            assert n == want.size and np.array_equal(out.download(n), want.astype(np.uint32)), rank
        ctx.set_option("union.rank", 1)
        seg.free()
        out.free()


def test_union_by_ranking_limits(ctx):
    from inverted_index_2_amd import II2Error
    a = np.arange(0, 600_000, dtype=np.uint32) * 3
    b = np.arange(0, 448_576, dtype=np.uint32) * 5                               # 1,048,576 postings in all: the limit
    c = np.array([7], np.uint32)
    for lists in ([a, b], [a, b, c]):                                            # ... and one more: the other paths
        seg = ctx.encode_lists(lists)
        out, n = ctx.union([(seg, i) for i in range(len(lists))])
        assert np.array_equal(out.download(n), np.unique(np.concatenate(lists)))
    seg = ctx.encode_lists([a[:50_000], b[:50_000]])
    with pytest.raises(II2Error):
        ctx.union([(seg, 0), (seg, 1)], out=ctx.empty(1000))

"""GPU: the wave-streaming dense intersection (csrc/intersect_dense.hip, chosen by the host for 2..4 lists whose
driver is dense) against numpy / the oracle — bit-exact id sequences, with the general tile kernels (option
intersect.dense = 0) as a second opinion.  Inputs are built to reach every path of the kernel: full blocks of
one-byte gaps, groups of four postings spanning more than 31 docs, multi-byte gaps and sparse stretches inside a dense
list (windows that split), short last blocks, lists from different segments, tombstones, ids next to 2^32."""
import numpy as np
import pytest

from inverted_index_2_amd import synth
from oracle import oracle as orc
from tests.gpu_util import ctx, sorted_unique  # noqa: F401

pytestmark = pytest.mark.gpu
AND2_DEFAULT = 1


def bernoulli(rng, p, lo, hi):
    return (np.flatnonzero(rng.random(hi - lo) < p) + lo).astype(np.uint32)


def _check(ctx, lists, removed=None, split=False):
    want = lists[0]
    for x in lists[1:]:
        want = np.intersect1d(want, x, assume_unique=True)
    if removed is not None:
        want = np.setdiff1d(want, removed, assume_unique=True)
    want = want.astype(np.uint32)
    tomb = ctx.tombstones(removed) if removed is not None else None
    if split:
        segs = [ctx.encode_lists([l]) for l in lists]
        ls = [(s, 0) for s in segs]
    else:
        seg = ctx.encode_lists(lists)
        ls = [(seg, i) for i in range(len(lists))]
    out = ctx.empty(min(l.size for l in lists) + 16)
    # dense = 1: two lists go through intersect_and2.hip (the shorter list tested against the longer one's bitmap; and2 = 1:
    # one launch with a look-back for the output offsets, 2: two kernels) unless intersect.and2 = 0, which sends them through
    # the n-list streaming kernel like three or four lists; dense = 0: general tiles
    for dense, and2 in ((1, 1), (1, 2), (1, 0), (0, 0)):
        ctx.set_option("intersect.dense", dense)
        ctx.set_option("intersect.and2", and2)
        out.upload(np.full(out.count, 0xDEADBEEF, np.uint32))
        _, n = ctx.intersect(ls, tomb=tomb, out=out)
        got = out.download(n)
        assert n == want.size, (dense, and2, n, want.size)
        assert np.array_equal(got, want), (dense, and2)
        assert np.all(out.download()[n:] == 0xDEADBEEF), (dense, and2)       # nothing written past the result
    ctx.set_option("intersect.dense", 1)
    ctx.set_option("intersect.and2", AND2_DEFAULT)
    for bpw in (32, 64):                           # longer waves: several rounds per wave, carried boundary words
        ctx.set_option("intersect.dense_bpw", bpw)
        _, n = ctx.intersect(ls, tomb=tomb, out=out)
        assert n == want.size and np.array_equal(out.download(n), want), bpw
    ctx.set_option("intersect.dense_bpw", 0)
    return want


@pytest.mark.parametrize("ps", [(0.5, 0.33), (0.3, 0.9), (0.26, 0.3), (0.5, 0.4, 0.3), (0.9, 0.5, 0.35, 0.3)])
def test_dense_bernoulli_lists(ctx, ps):
    rng = np.random.default_rng(int(sum(ps) * 1000))
    U = 1_600_000                                   # >= 1024 driver blocks at these densities
    lists = [bernoulli(rng, p, 0, U) for p in ps]
    want = _check(ctx, lists)
    assert want.size > 1000
    removed = bernoulli(rng, 0.02, 0, U)
    _check(ctx, lists, removed=removed)
    _check(ctx, lists, split=True)


def test_dense_with_sparse_stretches_and_multibyte_gaps(ctx):
    # dense lists with holes: empty stretches (gaps of 300 .. 200k docs -> two- and three-byte varints, windows that split),
    # a stretch where one list thins out to 1 % (groups of four postings far wider than 31 docs) and a short last block
    rng = np.random.default_rng(77)
    U = 2_400_000
    a, b = bernoulli(rng, 0.45, 0, U), bernoulli(rng, 0.35, 0, U)
    def punch(x, holes):
        keep = np.ones(x.size, bool)
        for lo, hi in holes:
            keep &= ~((x >= lo) & (x < hi))
        return x[keep]
    a = punch(a, [(100_000, 100_300), (500_000, 700_000), (1_000_000, 1_020_000)])
    b = punch(b, [(90_000, 130_000), (1_500_000, 1_500_900), (2_000_000, 2_000_040)])
    thin = (b >= 1_200_000) & (b < 1_400_000) & (rng.random(b.size) > 0.03)
    b = b[~thin]
    b = np.union1d(b, a[(a >= 1_200_000) & (a < 1_400_000)][::7]).astype(np.uint32)
    _check(ctx, [a, b])
    _check(ctx, [a, b], removed=bernoulli(rng, 0.01, 0, U))
    _check(ctx, [b, a, bernoulli(rng, 0.6, 0, U)])


def test_dense_lists_next_to_the_top_of_the_id_space(ctx):
    rng = np.random.default_rng(5)
    base = (1 << 32) - 1_200_000
    a, b = bernoulli(rng, 0.5, base, (1 << 32)), bernoulli(rng, 0.4, base, (1 << 32))
    a = np.union1d(a, np.asarray([0xFFFFFFFF, 0xFFFFFFFE], np.uint32)).astype(np.uint32)
    b = np.union1d(b, np.asarray([0xFFFFFFFF, 5, 17], np.uint32)).astype(np.uint32)       # one list also starts near 0: a 4G-doc gap
    _check(ctx, [a, b])


def test_dense_zipf_pairs_match_the_oracle(ctx):
    D = 6_000_000
    lists = [synth.zipf_list(r, D) for r in (2, 3)]
    removed = synth.geometric_postings(0.01, D, synth.term_seed(10**6))
    want = _check(ctx, lists, removed=removed)
    assert np.array_equal(want, orc.intersect(lists, removed))
    # identical lists, a subset, disjoint halves
    _check(ctx, [lists[0], lists[0].copy()])
    _check(ctx, [lists[0], lists[0][::3].copy(), lists[1]])
    lo, hi = lists[0][lists[0] < D // 2], lists[1][lists[1] >= D // 2]
    _check(ctx, [lo, hi])


def test_and2_bounded_wait_runs_out(ctx):
    """The one-launch two-list AND orders its output with waits between workgroups (look-back).  Every wait is bounded; when one
    runs out (forced here for workgroup 1: option intersect.and2_spin < 0) the call's count is all ones, ii2_intersect repeats
    the query through the two-kernel form and still returns the exact result, and ii2_ctx_sync reports the asynchronous call."""
    from inverted_index_2_amd.engine import II2Error
    rng = np.random.default_rng(11)
    U = 3_000_000
    a, b = bernoulli(rng, 0.5, 0, U), bernoulli(rng, 0.3, 0, U)
    want = np.intersect1d(a, b, assume_unique=True).astype(np.uint32)
    seg = ctx.encode_lists([a, b])
    ls = [(seg, 0), (seg, 1)]
    out = ctx.empty(b.size + 16)
    dcnt = ctx.empty(8, np.uint64)
    ctx.set_option("intersect.and2", 1)
    try:
        ctx.set_option("intersect.and2_spin", -1)
        before = ctx.counters()[1]
        _, n = ctx.intersect(ls, out=out)                     # synchronous: repeated inside the call
        assert n == want.size and np.array_equal(out.download(n), want)
        assert ctx.counters()[1] == before + 1
        ctx.intersect_async(ls, None, out, dcnt)              # asynchronous: poisoned count, reported at the next sync
        with pytest.raises(II2Error):
            ctx.sync()
        assert int(dcnt.download(1)[0]) == 0xFFFFFFFFFFFFFFFF
        ctx.set_option("intersect.and2_spin", 0)
        ctx.intersect_async(ls, None, out, dcnt)              # and the next call is clean again
        ctx.sync()
        n = int(dcnt.download(1)[0])
        assert n == want.size and np.array_equal(out.download(n), want)
        for spin in (1, 2, 5):                                # a tiny budget: some waits run out on their own; the sync call stays exact
            ctx.set_option("intersect.and2_spin", spin)
            _, n = ctx.intersect(ls, out=out)
            assert n == want.size and np.array_equal(out.download(n), want), spin
    finally:
        ctx.set_option("intersect.and2_spin", 0)
        try:
            ctx.sync()
        except II2Error:
            pass


# ---- the same kernel with OR semantics: unions paced by a long dense list (PrefixSearch, inverted_index.go:274-292) ----
def _check_union(ctx, lists, removed=None, split=False):
    want = lists[0]
    for x in lists[1:]:
        want = np.union1d(want, x)
    if removed is not None:
        want = np.setdiff1d(want, removed, assume_unique=True)
    want = want.astype(np.uint32)
    tomb = ctx.tombstones(removed) if removed is not None else None
    if split:
        segs = [ctx.encode_lists([l]) for l in lists]
        ls = [(s, 0) for s in segs]
    else:
        seg = ctx.encode_lists(lists)
        ls = [(seg, i) for i in range(len(lists))]
    out = ctx.empty(sum(l.size for l in lists) + 16)
    for stream in (1, 0):                          # the streaming kernel, then the fixed-range OR tiles as a second opinion
        ctx.set_option("union.stream", stream)
        _, n = ctx.union(ls, tomb=tomb, out=out)
        assert n == want.size, (stream, n, want.size)
        assert np.array_equal(out.download(n), want), stream
    ctx.set_option("union.stream", 1)
    return want


@pytest.mark.parametrize("ps", [(0.5, 0.33), (0.9, 0.05), (0.3, 0.3, 0.002), (0.6, 0.5, 0.4, 0.01)])
def test_stream_union_bernoulli_lists(ctx, ps):
    rng = np.random.default_rng(int(sum(ps) * 977))
    U = 1_600_000
    lists = [bernoulli(rng, p, 0, U) for p in ps]
    _check_union(ctx, lists)
    _check_union(ctx, lists, removed=bernoulli(rng, 0.05, 0, U))
    _check_union(ctx, lists[::-1], split=True)


def test_stream_union_lists_reaching_past_the_pacer(ctx):
    # the longest list covers [300k, 1.5M); the others start before it and end after it (the first and the last round
    # of the pacer stretch over them), one has holes and multi-byte gaps, one ends next to 2^32 - far outside: that
    # case must fall back to the fixed-range tiles and still be exact
    rng = np.random.default_rng(5)
    pacer = bernoulli(rng, 0.6, 300_000, 1_500_000)
    early = bernoulli(rng, 0.2, 250_000, 900_000)
    late = bernoulli(rng, 0.3, 700_000, 1_560_000)
    late = late[~((late >= 1_000_000) & (late < 1_090_000))]
    _check_union(ctx, [pacer, early, late])
    _check_union(ctx, [early, pacer])
    _check_union(ctx, [late, pacer], removed=bernoulli(rng, 0.03, 0, 1_600_000))
    far = np.concatenate([bernoulli(rng, 0.1, 0, 50_000), np.array([4_000_000_000, 0xFFFFFFFE], np.uint32)]).astype(np.uint32)
    _check_union(ctx, [pacer, far])


def test_stream_union_duplicates_and_identical_lists(ctx):
    rng = np.random.default_rng(6)
    a = bernoulli(rng, 0.5, 0, 1_000_000)
    _check_union(ctx, [a, a])
    _check_union(ctx, [a, a[::2], a[1::3]])

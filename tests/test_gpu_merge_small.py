"""GPU: the one-launch small-shard merge (ii2_merge_small, csrc/merge_small.hip) against the reference's semantics restated
on the host: term alignment in file.CompareTermValues order (file/types.go:24-26), same-term union (file/types.go:14-22),
the removed-list filter and the empty-term drop of Shard.Merge (shard.go:163-212, 219-225) — through the C ABI."""
import functools

import numpy as np
import pytest

from inverted_index_2_amd.engine import II2Error
from oracle import oracle as orc
from tests.gpu_util import ctx, sorted_unique  # noqa: F401

pytestmark = pytest.mark.gpu


def want_merge(dicts, lists, removed):
    """{term: surviving ids} in term order, terms without survivors dropped."""
    union = sorted(set(t for d in dicts for t in d), key=functools.cmp_to_key(orc.compare_terms))
    out = []
    rm = np.unique(np.asarray(removed, np.uint32))
    for t in union:
        parts = [lists[s][d.index(t)] for s, d in enumerate(dicts) if t in d]
        v = np.unique(np.concatenate(parts)) if parts else np.empty(0, np.uint32)
        v = v[~np.isin(v, rm)]
        if v.size:
            out.append((t, v.astype(np.uint32)))
    return out


def check(ctx, dicts, lists, removed=()):
    segs = [ctx.encode_lists(ls) for ls in lists]
    seg, terms, st = ctx.merge_small(segs, dicts, removed)
    want = want_merge(dicts, lists, removed)
    assert terms == [t for t, _ in want]
    assert st.n_terms_out == len(want) and st.n_out == sum(v.size for _, v in want) and st.n_in == sum(l.size for ls in lists for l in ls)
    if not want:
        assert seg is None
        return None
    po, v = seg.decode()
    assert np.array_equal(np.diff(po.astype(np.int64)), [w.size for _, w in want])
    assert np.array_equal(v, np.concatenate([w for _, w in want]))
    # the new segment is a first-class one: it merges, unions and exports like any other
    again, _ = ctx.merge_to_segment([seg, seg])
    po2, v2 = again.decode()
    assert np.array_equal(po2, po) and np.array_equal(v2, v)
    out, n = ctx.union([(seg, 0), (seg, len(want) - 1)])
    assert np.array_equal(out.download(n), np.union1d(want[0][1], want[-1][1]))
    blk, skip, payload = seg.export()
    seg2 = ctx.import_dv1(int(st.n_out), blk, skip, payload)          # passes the import's structural validation
    assert np.array_equal(seg2.decode()[1], v)
    return seg


def test_reference_kat_direct_segments(ctx):
    # shard_test.go:138-190 (TestMerging / TestMergeWithRemoval): three Puts = three direct segments, one value per term
    dicts = [[b"term1", b"term3"], [b"term2"], [b"term3"]]
    lists = [[np.array([1], np.uint32), np.array([1], np.uint32)], [np.array([2], np.uint32)], [np.array([3], np.uint32)]]
    check(ctx, dicts, lists)
    check(ctx, dicts, lists, removed=[2])                                  # term2 loses its only value: dropped (shard.go:192-194)
    # TestMergeEmptySegment (shard_test.go:192-214): nothing survives -> no segment (shard.go:219-225)
    assert check(ctx, [[b"term1"], [b"term1"]], [[np.array([1], np.uint32)]] * 2, removed=[1]) is None


@pytest.mark.parametrize("k,n_terms,max_len", [(2, 3, 1), (8, 3, 1), (32, 5, 1), (8, 40, 30), (64, 6, 3), (3, 100, 40), (16, 30, 1)])
def test_random_small_merges(ctx, k, n_terms, max_len):
    rng = np.random.default_rng(k * 100 + n_terms)
    vocab = sorted({bytes(rng.choice([0x61, 0x62, 0x7A, 0x00, 0xFF], int(rng.integers(0, 12))).tolist()) for _ in range(3 * n_terms)},
                   key=functools.cmp_to_key(orc.compare_terms))
    dicts, lists = [], []
    for s in range(k):
        d = [t for t in vocab if rng.random() < min(1.0, n_terms / len(vocab))] or [vocab[0]]
        dicts.append(d)
        lists.append([sorted_unique(rng, int(rng.integers(1, max_len + 1)), 500) for _ in d])
    check(ctx, dicts, lists)
    check(ctx, dicts, lists, removed=rng.integers(0, 500, 60).astype(np.uint32))     # unsorted, duplicates
    check(ctx, dicts, lists, removed=np.arange(500, dtype=np.uint32))               # everything removed


def test_long_lists_edge_ids_and_long_terms(ctx):
    rng = np.random.default_rng(9)
    # lists that span several DV1 blocks, ids at both ends of the id space, terms that tie on their first 8 bytes
    dicts = [[b"", b"prefix__", b"prefix__a", b"prefix__ab"], [b"prefix__", b"prefix__ab", b"z"], [b"prefix__a\x00"]]
    lists = [[np.array([0, 0xFFFFFFFF], np.uint32), sorted_unique(rng, 700, 1 << 20), sorted_unique(rng, 300, 5000), np.array([7], np.uint32)],
             [sorted_unique(rng, 900, 1 << 20), np.array([7, 8], np.uint32), np.array([0xFFFFFFFE, 0xFFFFFFFF], np.uint32)],
             [sorted_unique(rng, 513, 4000)]]
    check(ctx, dicts, lists)
    check(ctx, dicts, lists, removed=[0, 7, 0xFFFFFFFF, 0xFFFFFFFF, 12])


def test_limits_are_refused_not_truncated(ctx):
    rng = np.random.default_rng(1)
    big = [sorted_unique(rng, 5000, 1 << 24), sorted_unique(rng, 5000, 1 << 24)]
    with pytest.raises(II2Error) as e:
        ctx.merge_small([ctx.encode_lists([big[0]]), ctx.encode_lists([big[1]])], [[b"a"], [b"a"]])
    assert e.value.code == -5                                              # II2_ERANGE: > 8192 postings
    many = [b"t%04d" % i for i in range(300)]
    with pytest.raises(II2Error) as e:
        ctx.merge_small([ctx.encode_lists([np.array([1], np.uint32)] * 300)] * 2, [many, many])
    assert e.value.code == -5                                              # > 512 terms
    with pytest.raises(II2Error):
        ctx.merge_small([ctx.encode_lists([np.array([1], np.uint32)] * 2)], [[b"a"]])      # lists and terms disagree


def want_read(dicts, lists):
    """[(term, merged ids)] in term order, every term kept (a read drops nothing)."""
    union = sorted(set(t for d in dicts for t in d), key=functools.cmp_to_key(orc.compare_terms))
    out = []
    for t in union:
        parts = [lists[s][d.index(t)] for s, d in enumerate(dicts) if t in d]
        out.append((t, np.unique(np.concatenate(parts)).astype(np.uint32)))
    return out


@pytest.mark.parametrize("k,n_terms,max_len", [(1, 4, 3), (2, 3, 1), (8, 30, 20), (64, 6, 3), (3, 100, 40)])
def test_small_read_is_the_merge_without_filter_or_drop(ctx, k, n_terms, max_len):
    """ii2_read_small (Shard.Read of a few small segments in one launch): the k-way merge of the dictionaries, same-term
    union, every term handed back - also the ones whose lists are empty - straight to host memory."""
    rng = np.random.default_rng(4000 + k * 7 + n_terms)
    vocab = [bytes(rng.integers(97, 100, int(rng.integers(1, 12))).astype(np.uint8)) for _ in range(3 * n_terms)]
    vocab = sorted(set(vocab), key=functools.cmp_to_key(orc.compare_terms))
    dicts, lists = [], []
    for s in range(k):
        d = sorted(set(vocab[i] for i in rng.choice(len(vocab), min(n_terms, len(vocab)), replace=False)), key=functools.cmp_to_key(orc.compare_terms))
        dicts.append(d)
        lists.append([sorted_unique(rng, int(rng.integers(0, max_len + 1)), 5000) for _ in d])
    segs = [ctx.encode_lists(ls) for ls in lists]
    terms, po, vals = ctx.read_small(segs, dicts)
    want = want_read(dicts, lists)
    assert terms == [t for t, _ in want]
    assert np.array_equal(np.diff(po.astype(np.int64)), [w.size for _, w in want])
    assert np.array_equal(vals, np.concatenate([w for _, w in want] + [np.empty(0, np.uint32)]))
    # a range-restricted read: a slice of every segment's terms (list_first shifts into the segment)
    lo_t, hi_t = vocab[len(vocab) // 4], vocab[(3 * len(vocab)) // 4]
    cmpf = orc.compare_terms
    sl, lf, sub_lists = [], [], []
    for d, ls in zip(dicts, lists):
        idx = [j for j, t in enumerate(d) if cmpf(t, lo_t) >= 0 and cmpf(t, hi_t) <= 0]
        sl.append([d[j] for j in idx])
        lf.append(idx[0] if idx else 0)
        sub_lists.append([ls[j] for j in idx])
    terms2, po2, vals2 = ctx.read_small(segs, sl, lf)
    want2 = want_read(sl, sub_lists)
    assert terms2 == [t for t, _ in want2]
    assert np.array_equal(np.diff(po2.astype(np.int64)), [w.size for _, w in want2])
    assert np.array_equal(vals2, np.concatenate([w for _, w in want2] + [np.empty(0, np.uint32)]))


def test_small_read_limits(ctx):
    rng = np.random.default_rng(9)
    big = [sorted_unique(rng, 5000, 1 << 20), sorted_unique(rng, 5000, 1 << 20)]
    seg = ctx.encode_lists(big)
    with pytest.raises(II2Error):
        ctx.read_small([seg], [[b"a", b"b"]])                 # 10000 postings: over II2_SMALL_MERGE_POSTINGS
    small = ctx.encode_lists([np.array([1, 2], np.uint32)])
    with pytest.raises(II2Error):
        ctx.read_small([small], [[b"a", b"b"]])               # the dictionary names more lists than the segment has

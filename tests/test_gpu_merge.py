"""GPU: segment merge / union kernels against the oracle's restatement of Shard.Merge
(bit-exact offsets and id sequences) — through the C ABI."""
import numpy as np
import pytest

from inverted_index_2_amd import synth
from oracle import oracle as orc
from tests.gpu_util import ctx, sorted_unique  # noqa: F401

pytestmark = pytest.mark.gpu


def _check_merge(ctx, offs, vals, removed=None):
    segs = [ctx.encode(o, v) for o, v in zip(offs, vals)]
    tomb = ctx.tombstones(removed) if removed is not None and len(removed) else None
    w_off, w_vals, w_terms = orc.merge_segments(offs, vals, np.sort(removed) if removed is not None else ())
    out_off, out_vals, st = ctx.merge(segs, tomb=tomb)
    g_off = out_off.download()
    assert np.array_equal(g_off, w_off)
    assert np.array_equal(out_vals.download(int(w_off[-1])), w_vals)
    assert st.n_out == int(w_off[-1]) and st.n_terms_out == w_terms
    assert st.n_in == sum(int(o[-1]) for o in offs)
    # host-buffer entry point (what the cgo binding calls)
    h_off, h_vals, hst = ctx.merge_host(offs, vals, removed if removed is not None else ())
    assert np.array_equal(h_off, w_off) and np.array_equal(h_vals, w_vals) and hst.n_terms_out == w_terms
    return st


def _rand_segments(rng, k, T, max_len, universe, p_empty=0.3):
    offs, vals = [], []
    for _ in range(k):
        lens = rng.integers(0, max_len + 1, T)
        lens[rng.random(T) < p_empty] = 0
        o = np.concatenate([[0], np.cumsum(lens)]).astype(np.uint64)
        v = [sorted_unique(rng, int(n), universe) for n in lens]
        lens2 = np.array([x.size for x in v])
        o = np.concatenate([[0], np.cumsum(lens2)]).astype(np.uint64)
        offs.append(o)
        vals.append(np.concatenate(v + [np.empty(0, np.uint32)]).astype(np.uint32))
    return offs, vals


def test_reference_kat_tables(ctx):
    # shard_test.go:138-190 as aligned-term CSR: terms term1,term2,term3; three direct segments
    offs = [np.array([0, 1, 1, 2], np.uint64), np.array([0, 0, 1, 1], np.uint64), np.array([0, 0, 0, 1], np.uint64)]
    vals = [np.array([1, 1], np.uint32), np.array([2], np.uint32), np.array([3], np.uint32)]
    _check_merge(ctx, offs, vals)                       # TestMergeWithRemoval before removal
    st = _check_merge(ctx, offs, vals, removed=np.array([2], np.uint32))
    assert st.n_terms_out == 2                          # term2 dropped (shard.go:192-194)
    # TestMergeEmptySegment: two segments {term1:[1]}, value 1 removed -> nothing survives
    offs = [np.array([0, 1], np.uint64)] * 2
    vals = [np.array([1], np.uint32)] * 2
    st = _check_merge(ctx, offs, vals, removed=np.array([1], np.uint32))
    assert st.n_out == 0 and st.n_terms_out == 0
    seg, st2 = ctx.merge_to_segment([ctx.encode(o, v) for o, v in zip(offs, vals)], ctx.tombstones(np.array([1], np.uint32)))
    assert seg is None and st2.n_terms_out == 0         # shard.go:219-225: no segment is written


@pytest.mark.parametrize("k,T,max_len", [(1, 7, 40), (2, 50, 30), (3, 200, 10), (5, 64, 300), (16, 500, 12), (64, 40, 20)])
def test_random_small_terms(ctx, k, T, max_len):
    rng = np.random.default_rng(k * 1000 + T)
    offs, vals = _rand_segments(rng, k, T, max_len, 5000)
    _check_merge(ctx, offs, vals)
    _check_merge(ctx, offs, vals, removed=rng.integers(0, 5000, 400).astype(np.uint32))   # unsorted, duplicates


def test_edge_values_and_all_removed(ctx):
    offs = [np.array([0, 2, 2, 3], np.uint64), np.array([0, 1, 1, 3], np.uint64)]
    vals = [np.array([0, 0xFFFFFFFF, 7], np.uint32), np.array([0xFFFFFFFF, 7, 9], np.uint32)]
    _check_merge(ctx, offs, vals)
    _check_merge(ctx, offs, vals, removed=np.array([0, 7, 9, 0xFFFFFFFF, 7], np.uint32))
    empty = [np.zeros(4, np.uint64)] * 3
    _check_merge(ctx, empty, [np.empty(0, np.uint32)] * 3)


@pytest.mark.parametrize("k", [2, 4, 16])
def test_large_terms_use_doc_range_tiles(ctx, k):
    rng = np.random.default_rng(77 + k)
    T = 12
    offs, vals = [], []
    for s in range(k):
        lists = []
        for t in range(T):
            n = [0, 3, 40_000, 700, 9_000, 1][t % 6]
            lists.append(sorted_unique(rng, n, 1 << 22))
        o = np.concatenate([[0], np.cumsum([x.size for x in lists])]).astype(np.uint64)
        offs.append(o)
        vals.append(np.concatenate(lists).astype(np.uint32))
    st = _check_merge(ctx, offs, vals, removed=rng.integers(0, 1 << 22, 50_000).astype(np.uint32))
    assert st.n_tiles > T


def test_clustered_large_term_replans(ctx):
    # one segment's list is clustered where the others are not: unbalanced splitters must still be exact
    rng = np.random.default_rng(5)
    a = np.arange(1_000_000, 1_030_000, dtype=np.uint32)             # dense cluster
    b = sorted_unique(rng, 30_000, 1 << 30)
    c = np.concatenate([sorted_unique(rng, 100, 1 << 20), np.arange(1_010_000, 1_050_000, dtype=np.uint32)])
    c = np.unique(c).astype(np.uint32)
    offs = [np.array([0, x.size], np.uint64) for x in (a, b, c)]
    _check_merge(ctx, offs, [a, b, c])


def test_zipf_workload_miniature(ctx):
    # BASELINE config 3 in miniature: Zipf term sizes, 16 segments, 10 % duplicated postings, 1 % tombstones
    offs, vals, removed = synth.merge_workload(20_000, 16, 120, 2_000_000)
    st = _check_merge(ctx, offs, vals, removed)
    assert st.n_out < st.n_in


def test_merge_to_segment_roundtrip(ctx):
    rng = np.random.default_rng(9)
    offs, vals = _rand_segments(rng, 4, 300, 50, 100_000)
    removed = rng.integers(0, 100_000, 2000).astype(np.uint32)
    segs = [ctx.encode(o, v) for o, v in zip(offs, vals)]
    merged, st = ctx.merge_to_segment(segs, ctx.tombstones(removed))
    w_off, w_vals, _ = orc.merge_segments(offs, vals, np.sort(removed))
    po, v = merged.decode()
    assert np.array_equal(po, w_off) and np.array_equal(v, w_vals)
    # merging the merged segment with itself is idempotent (cf. shard_test.go:153 idempotency)
    again, _ = ctx.merge_to_segment([merged, merged])
    po2, v2 = again.decode()
    assert np.array_equal(po2, w_off) and np.array_equal(v2, w_vals)


def _segment_equals_oracle_encoding(seg, w_off, w_vals):
    oblk, oskip, opayload = orc.dv1_encode(w_off, w_vals)
    blk, skip, payload = seg.export()
    assert seg.info.n_postings == int(w_off[-1]) and seg.info.n_blocks == oskip.size - 1 and seg.info.n_bytes == opayload.size
    assert np.array_equal(blk, oblk)
    assert np.array_equal(skip["first_doc"], oskip["first_doc"]) and np.array_equal(skip["byte_off"], oskip["byte_off"])
    assert np.array_equal(payload, opayload)
    po, v = seg.decode()
    assert np.array_equal(po, w_off) and np.array_equal(v, w_vals)


@pytest.mark.parametrize("stream", [1, 0, -1])
def test_merged_segment_bytes_equal_the_oracles_encoding(ctx, stream):
    """Shard.Merge ends in Writer.Append of every merged term (shard.go:207, file/writer.go:32-59): the segment that
    ii2_merge_segments_to_seg returns must be, byte for byte, the DV1 encoding of the oracle's merge - through the one-pass
    encoder (encode.stream = 1), the two-pass one (0) and the one-pass one giving up on a look-back wait (-1: it hands over to
    the two-pass encoder).  Shapes: many tiny and empty lists, lists of exactly 255 / 256 / 257 / 512 / 513 postings, sparse lists
    (3- to 5-byte gaps), ids next to 2^32, a list far longer than a workgroup's 4096 ids, everything removed from some terms."""
    rng = np.random.default_rng(123)
    ctx.set_option("encode.stream", stream)
    try:
        # (a) random small terms with empty ones in between, four segments, tombstones
        offs, vals = _rand_segments(rng, 4, 3000, 40, 5_000_000, p_empty=0.5)
        removed = rng.integers(0, 5_000_000, 50_000).astype(np.uint32)
        merged, st = ctx.merge_to_segment([ctx.encode(o, v) for o, v in zip(offs, vals)], ctx.tombstones(removed))
        w_off, w_vals, _ = orc.merge_segments(offs, vals, np.sort(removed))
        _segment_equals_oracle_encoding(merged, w_off, w_vals)
        # (b) block-boundary lengths, a 70,000-posting list, sparse ids up to 2^32 - 1, a term that loses everything
        lens = [255, 256, 257, 0, 512, 513, 1, 70_000, 0, 0, 3, 1024, 5]
        lists = [sorted_unique(rng, n, 1 << 22) for n in lens]
        lists[6] = np.array([0xFFFFFFFF], np.uint32)
        lists[10] = np.array([7, 1 << 31, 0xFFFFFFFE], np.uint32)
        lists[12] = np.array([100, 101, 102, 103, 104], np.uint32)
        other = [x[::2].copy() for x in lists]                     # second segment: half of every list again (duplicates)
        offs2 = [np.concatenate([[0], np.cumsum([x.size for x in ls])]).astype(np.uint64) for ls in (lists, other)]
        vals2 = [np.concatenate(ls).astype(np.uint32) for ls in (lists, other)]
        removed2 = np.array([100, 101, 102, 103, 104, 7], np.uint32)
        merged2, _ = ctx.merge_to_segment([ctx.encode(o, v) for o, v in zip(offs2, vals2)], ctx.tombstones(removed2))
        w_off2, w_vals2, _ = orc.merge_segments(offs2, vals2, np.sort(removed2))
        assert w_off2[13] == w_off2[12]                            # the last term lost everything
        _segment_equals_oracle_encoding(merged2, w_off2, w_vals2)
        # (c) BASELINE config 3 in miniature (Zipf sizes, 16 segments)
        offs3, vals3, removed3 = synth.merge_workload(20_000, 16, 120, 2_000_000)
        merged3, _ = ctx.merge_to_segment([ctx.encode(o, v) for o, v in zip(offs3, vals3)], ctx.tombstones(removed3))
        w_off3, w_vals3, _ = orc.merge_segments(offs3, vals3, removed3)
        _segment_equals_oracle_encoding(merged3, w_off3, w_vals3)
    finally:
        ctx.set_option("encode.stream", 1)


def _one_segment_case(ctx, lists):
    off = np.concatenate([[0], np.cumsum([x.size for x in lists])]).astype(np.uint64)
    vals = np.concatenate(list(lists) + [np.empty(0, np.uint32)]).astype(np.uint32)
    before = ctx.counters()[1]
    merged, st = ctx.merge_to_segment([ctx.encode(off, vals)], None)
    assert st.n_out == vals.size
    assert ctx.counters()[1] == before                          # (the one-pass encoder did it: no hand-over to the two-pass one)
    _segment_equals_oracle_encoding(merged, off, vals)          # (one segment, nothing removed: the merge is the identity)
    merged.free()


def test_one_pass_encoder_on_list_structures_that_stress_its_bookkeeping(ctx):
    """k_enc_stream derives block starts, block numbers and block owners from a bit mask of list starts per 1024-id tile
    (encode_stream.hip); the shapes here aim at every branch of that: long runs of EMPTY lists in front of a list that is
    longer than a block (the owner of a block that starts inside a list is then found by search, not by counting starts),
    more than 64 list starts in one tile (several rounds of the marking loop), tiles filled with five-byte gaps to the
    proven worst case of the LDS stage (lists of 16 ids 2^28 apart: 4800 bytes per 1024 ids), and totals right at the
    tile / wave / workgroup sizes (1, 1023, 1024, 1025, 2048, 8192, 8193 ids)."""
    rng = np.random.default_rng(77)
    e = np.empty(0, np.uint32)
    # (a) empty runs before long lists; the long lists start inside tiles
    lists = [sorted_unique(rng, 300, 1 << 20)] + [e] * 5000 + [sorted_unique(rng, 700, 1 << 24)] + [e] * 3000 + \
            [sorted_unique(rng, 2000, 1 << 22), sorted_unique(rng, 5, 100)] + [e] * 70000 + [sorted_unique(rng, 5000, 1 << 30), e, e, sorted_unique(rng, 257, 1 << 12)]
    _one_segment_case(ctx, lists)
    # (b) thousands of one- to three-posting lists, then a long one
    tiny = [sorted_unique(rng, int(n), 1 << 31) for n in rng.integers(1, 4, 6000)]
    _one_segment_case(ctx, tiny + [sorted_unique(rng, 9000, 1 << 26)] + tiny[:100])
    # (c) five-byte gaps: 16 ids 2^28 apart per list, 300 lists (every tile at the stage's worst case), behind a 15-id tail
    far = [(np.arange(16, dtype=np.uint64) * (1 << 28) + int(b)).astype(np.uint32) for b in rng.integers(0, 1 << 28, 300)]
    _one_segment_case(ctx, [far[0][1:]] + far)
    _one_segment_case(ctx, [sorted_unique(rng, 1009, 1 << 20)] + far)          # (the same lists at another phase of the tiles)
    # (d) totals at the sizes the kernel is cut by
    for n in (1, 1023, 1024, 1025, 2048, 4096, 8191, 8192, 8193, 16384):
        _one_segment_case(ctx, [sorted_unique(rng, n, 1 << 28)])
        _one_segment_case(ctx, [sorted_unique(rng, n - n // 2, 1 << 28), e, sorted_unique(rng, n // 2, 1 << 16)])


def test_direct_placement_wait_runs_out_and_the_merge_is_repeated(ctx):
    """The merge's direct placement (tiles move their survivors to their final place once a scanner workgroup has turned the
    tiles' counts into offsets) rests on workgroups starting in index order.  Every wait is bounded: with a scanner that never
    runs (option debug.merge_skip bit 6) and a short budget (merge.spin) the workers' waits run out, every workgroup leaves
    promptly, and ii2_merge_segments repeats the call through the parking + packing pass - same offsets, same ids."""
    offs, vals, removed = synth.merge_workload(30_000, 8, 150, 3_000_000)
    segs = [ctx.encode(o, v) for o, v in zip(offs, vals)]
    tomb = ctx.tombstones(removed)
    w_off, w_vals, w_terms = orc.merge_segments(offs, vals, removed)
    before = ctx.counters()[0]
    try:
        ctx.set_option("merge.spin", 200)
        ctx.set_option("debug.merge_skip", 64)
        out_off, out_vals, st = ctx.merge(segs, tomb=tomb)
    finally:
        ctx.set_option("debug.merge_skip", 0)
        ctx.set_option("merge.spin", 0)
    assert ctx.counters()[0] == before + 1
    assert np.array_equal(out_off.download(), w_off) and np.array_equal(out_vals.download(int(w_off[-1])), w_vals)
    assert st.n_out == int(w_off[-1]) and st.n_terms_out == w_terms
    out_off2, out_vals2, _ = ctx.merge(segs, tomb=tomb)                      # and the next call takes the direct placement again
    assert ctx.counters()[0] == before + 1
    assert np.array_equal(out_off2.download(), w_off) and np.array_equal(out_vals2.download(int(w_off[-1])), w_vals)


@pytest.mark.parametrize("n", [1, 2, 3, 8, 33])
def test_union_matches_prefix_search_dedupe(ctx, n):
    # inverted_index.go:274-292: append every matching term's values, sort, compact
    rng = np.random.default_rng(300 + n)
    lists = [sorted_unique(rng, int(rng.integers(0, 3000)), 50_000) for _ in range(n)]
    seg = ctx.encode_lists(lists)
    want = orc.union(lists)
    out, cnt = ctx.union([(seg, i) for i in range(n)])
    assert cnt == want.size and np.array_equal(out.download(cnt), want)
    assert np.array_equal(ctx.union_host(lists), want)
    removed = rng.integers(0, 50_000, 500).astype(np.uint32)
    out, cnt = ctx.union([(seg, i) for i in range(n)], tomb=ctx.tombstones(removed))
    assert np.array_equal(out.download(cnt), orc.filter_removed(want, np.sort(removed)))


def test_union_large_lists(ctx):
    rng = np.random.default_rng(31)
    lists = [sorted_unique(rng, 200_000, 3_000_000), sorted_unique(rng, 5, 3_000_000), sorted_unique(rng, 90_000, 3_000_000)]
    seg = ctx.encode_lists(lists)
    want = orc.union(lists)
    out, cnt = ctx.union([(seg, i) for i in range(3)])
    assert cnt == want.size and np.array_equal(out.download(cnt), want)


def test_search_by_prefix_kat(ctx):
    # inverted_index_test.go:196-221: "a1" -> {1,2}; "term" -> {5,6,7}
    assert ctx.union_host([[1], [1, 2]]).tolist() == [1, 2]
    assert ctx.union_host([[5], [6], [7]]).tolist() == [5, 6, 7]
    assert ctx.intersect_host([[1, 2, 3], [2, 3, 4]], removed=[3]).tolist() == [2]


def test_bucket_fold_fallbacks_on_clustered_ids(ctx):
    """The tile kernel folds runs with a bucket sort keyed on the doc id; ids clustered so that a bucket overflows
    send the tile to the pairwise folds instead.  Batches of small terms and range tiles of a large term, with
    duplicates across segments and tombstones inside the clusters."""
    rng = np.random.default_rng(4242)
    k, T = 8, 60
    segs = []
    for s in range(k):
        lists = []
        for t in range(T):
            if t == 7:        # a large term: dense cluster + two far outliers in every segment
                ids = np.concatenate([rng.choice(6000, 700, replace=False), [1_000_000_000 + s, 3_000_000_000 - s]])
            else:             # small terms: ~60 ids inside a 100-doc window + one id far away
                base = int(rng.integers(0, 1 << 20))
                ids = np.concatenate([base + rng.choice(100, int(rng.integers(20, 70)), replace=False), [4_000_000_000 - t]])
            lists.append(np.unique(ids).astype(np.uint32))
        segs.append(lists)
    removed = np.unique(np.concatenate([rng.integers(0, 6000, 500), (1 << 19) + rng.integers(0, 1 << 19, 2000)])).astype(np.uint32)
    offs = [np.concatenate([[0], np.cumsum([x.size for x in lists])]).astype(np.uint64) for lists in segs]
    vals = [np.concatenate(lists).astype(np.uint32) for lists in segs]
    _check_merge(ctx, offs, vals, removed)
    _check_merge(ctx, offs, vals, None)


def _check_union(ctx, lists, removed=None):
    seg = ctx.encode_lists(lists)
    tomb = ctx.tombstones(removed) if removed is not None else None
    want = orc.union(lists)
    if removed is not None:
        want = orc.filter_removed(want, np.sort(removed))
    for dense in (1, 0):       # byte-map tiles with OR (lists dense together) / the merge passes
        ctx.set_option("union.dense", dense)
        out, n = ctx.union([(seg, i) for i in range(len(lists))], tomb=tomb)
        assert n == want.size and np.array_equal(out.download(n), want), dense
    ctx.set_option("union.dense", 1)


@pytest.mark.parametrize("k", [1, 2, 3, 9, 64])
def test_union_dense_path(ctx, k):
    """Lists dense enough together for the OR tiles: ranges that start and end apart, an empty list among them,
    ids at the ends of the id space, tombstones; both paths must give the reference's sort + compact."""
    rng = np.random.default_rng(300 + k)
    lists = []
    for i in range(k):
        lo = int(rng.integers(0, 200_000))
        hi = lo + int(rng.integers(50_000, 600_000))
        lists.append((lo + sorted_unique(rng, int(rng.integers(10_000, 90_000)), hi - lo)).astype(np.uint32))
    if k >= 3:
        lists[1] = np.empty(0, np.uint32)
    _check_union(ctx, lists)
    _check_union(ctx, lists, removed=rng.integers(0, 800_000, 30_000).astype(np.uint32))


def test_union_dense_high_ids_and_sparse_fallback(ctx):
    top = 0xFFFFFFFF
    a = (top - np.arange(0, 300_000, 2, dtype=np.uint64)[::-1]).astype(np.uint32)       # dense run ending at 2^32 - 1
    b = (top - np.arange(1, 300_000, 3, dtype=np.uint64)[::-1]).astype(np.uint32)
    _check_union(ctx, [a, b])
    rng = np.random.default_rng(5)
    sparse = [sorted_unique(rng, 40_000, 1 << 31) for _ in range(3)]                       # far too sparse: merge path
    _check_union(ctx, sparse)
    _check_union(ctx, [a, sparse[0]])                                                       # dense run + ids spread over 2^31


@pytest.mark.parametrize("k", [2, 16, 40])
def test_dense_single_term_tiles_take_the_bitmap_path(ctx, k):
    # terms dense enough for a tile's doc range to fit the LDS bitmap (>= 1 posting per ~64 docs), next to terms that
    # are not; docs clustered so that ranges of one term differ in density; duplicates across segments; tombstones
    # inside and outside the tiles' ranges; ids next to 2^32.  Both paths (option merge.bitmap_tiles) against the oracle.
    rng = np.random.default_rng(100 + k)
    U = 3_000_000
    def docs(p, lo, hi):
        return (np.flatnonzero(rng.random(hi - lo) < p) + lo).astype(np.uint32)
    terms = [
        docs(0.7, 0, 400_000),                                                          # very dense: few words per tile
        np.concatenate([docs(0.05, 0, 1_000_000), docs(0.0005, 1_000_000, U)]),        # dense head, sparse tail (bucket fold)
        docs(1 / 70, 0, U),                                                             # just inside the bitmap's reach
        docs(1 / 400, 0, U),                                                            # outside it
        docs(0.3, (1 << 32) - 200_000, (1 << 32) - 1).astype(np.uint32),               # dense, at the top of the id space
        docs(0.2, 5_000, 5_500),                                                        # small term between the large ones
    ]
    T = len(terms)
    where = [rng.integers(0, k, t.size) for t in terms]
    dup = [rng.random(t.size) < 0.1 for t in terms]
    where2 = [(w + 1 + rng.integers(0, k - 1, w.size)) % k for w in where]
    offs, vals = [], []
    for s in range(k):
        lists = [np.unique(np.concatenate([t[w == s], t[d & (w2 == s)]])) for t, w, d, w2 in zip(terms, where, dup, where2)]
        offs.append(np.concatenate([[0], np.cumsum([x.size for x in lists])]).astype(np.uint64))
        vals.append(np.concatenate(lists).astype(np.uint32))
    removed = np.concatenate([docs(0.02, 0, U), docs(0.05, (1 << 32) - 200_000, (1 << 32) - 1), np.array([0xFFFFFFFE], np.uint32)])
    for on in (1, 0):
        ctx.set_option("merge.bitmap_tiles", on)
        _check_merge(ctx, offs, vals)
        st = _check_merge(ctx, offs, vals, removed=removed)
        assert st.n_tiles > T
    ctx.set_option("merge.bitmap_tiles", 1)


def test_output_too_small_writes_nothing(ctx):
    """II2_ECAPACITY is all-or-nothing for the merge (include/ii2.h): neither the offsets nor the values are touched."""
    from inverted_index_2_amd.engine import II2Error
    rng = np.random.default_rng(8)
    offs, vals = _rand_segments(rng, 3, 200, 40, 1_000_000)
    segs = [ctx.encode(o, v) for o, v in zip(offs, vals)]
    w_off, w_vals, _ = orc.merge_segments(offs, vals, ())
    n_out = int(w_off[-1])
    out_off = ctx.empty(201, np.uint64).upload(np.full(201, 0xABCDEF0123456789, np.uint64))
    out_vals = ctx.empty(n_out - 1).upload(np.full(n_out - 1, 0xDEADBEEF, np.uint32))
    with pytest.raises(II2Error) as e:
        ctx.merge(segs, None, out_off, out_vals)
    assert e.value.code == -4
    assert np.all(out_off.download() == np.uint64(0xABCDEF0123456789)) and np.all(out_vals.download() == 0xDEADBEEF)
    out_vals2 = ctx.empty(n_out)                                  # exactly enough: succeeds
    g_off, g_vals, st = ctx.merge(segs, None, out_off, out_vals2)
    assert np.array_equal(g_off.download(), w_off) and np.array_equal(g_vals.download(n_out), w_vals)


@pytest.mark.parametrize("k,tile", [(3, 300), (16, 700), (64, 3000)])
def test_range_tiles_cut_blocks_at_any_byte(ctx, k, tile):
    """Round 3: a range tile decodes only its own part of a block that straddles its doc bounds (merge.hip: cuts).  Small
    tiles put a cut into nearly every block; the lists mix 1- to 5-byte gaps, repeat ids inside a list (gap 0) and
    share ids across lists, so cuts fall before, inside and after multi-byte varints and runs of equal ids."""
    rng = np.random.default_rng(900 + k)
    ctx.set_option("merge.large_tile", tile)
    ctx.set_option("merge.bitmap_tiles", 0)
    try:
        offs, vals = [], []
        shared = np.sort(rng.integers(0, 1 << 31, 4000).astype(np.uint32))
        for s in range(k):
            lists = []
            for t in range(3):
                n = [20_000, 0, 6_000][t] // (1 + s % 3)
                widths = rng.choice([7, 14, 21, 28], n, p=[0.5, 0.3, 0.15, 0.05])
                gaps = (rng.integers(0, 1 << 30, n) % (1 << widths)).astype(np.uint64)
                gaps[rng.random(n) < 0.05] = 0                                  # the same id twice in one list
                ids = np.cumsum(gaps)
                ids = ids[ids < (1 << 32) - 1].astype(np.uint32)
                if t == 0: ids = np.sort(np.concatenate([ids, shared[rng.random(shared.size) < 0.5]]))
                lists.append(ids)
            offs.append(np.concatenate([[0], np.cumsum([x.size for x in lists])]).astype(np.uint64))
            vals.append(np.concatenate(lists).astype(np.uint32))
        removed = np.unique(np.concatenate([shared[::7], rng.integers(0, 1 << 31, 5000).astype(np.uint32)]))
        st = _check_merge(ctx, offs, vals, removed=removed)
        assert st.n_tiles > 20
    finally:
        ctx.set_option("merge.large_tile", 0)
        ctx.set_option("merge.bitmap_tiles", 1)

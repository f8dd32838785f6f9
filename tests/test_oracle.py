"""CPU: pins the oracle (oracle/) against the reference's own known-answer tables
(tests/golden/ref_kats.json) and against an independent numpy formulation."""
import numpy as np
import pytest

from oracle import oracle as orc
from oracle import ref_model
from tests.kat_runner import run_script


def test_sort_matches_numpy():
    rng = np.random.default_rng(1)
    for n in [0, 1, 2, 3, 23, 24, 25, 100, 1000, 100_000]:
        for hi in [4, 1 << 16, 1 << 32]:
            v = rng.integers(0, hi, n, dtype=np.uint64).astype(np.uint32)
            assert np.array_equal(orc.sort_u32(v), np.sort(v))
    v = np.arange(50_000, dtype=np.uint32)
    assert np.array_equal(orc.sort_u32(v[::-1]), v)
    assert np.array_equal(orc.sort_u32(np.zeros(10_000, np.uint32)), np.zeros(10_000, np.uint32))


def test_merge_term_values_is_set_union():
    rng = np.random.default_rng(2)
    for _ in range(50):
        a = rng.integers(0, 500, rng.integers(0, 200)).astype(np.uint32)
        b = rng.integers(0, 500, rng.integers(0, 200)).astype(np.uint32)
        assert np.array_equal(orc.merge_term_values(a, b), np.union1d(a, b))
    assert orc.merge_term_values([], []).size == 0
    assert orc.merge_term_values([0, 0xFFFFFFFF], [0xFFFFFFFF]).tolist() == [0, 0xFFFFFFFF]


def test_compare_terms():
    assert orc.compare_terms(b"a", b"b") == -1
    assert orc.compare_terms(b"b", b"a") == 1
    assert orc.compare_terms(b"a", b"a") == 0
    assert orc.compare_terms(b"a", b"aa") == -1
    assert orc.compare_terms(b"", b"") == 0
    assert orc.compare_terms(b"\xff", b"a") == 1      # bytes.Compare is unsigned


def test_removed_lists_kat(kats):
    k = kats["removed"]["TestRemovedLists"]
    rl = ref_model.RemovedLists()
    for ts, vals in k["batches"]:
        rl.put(ts, vals)
    assert rl.values().tolist() == k["values"]
    rl.sync(k["sync_timestamps"])
    assert rl.values().tolist() == k["values_after_sync"]
    rl.sync([])                       # removed_list.go:58-60 — empty is a no-op
    assert rl.values().tolist() == k["values_after_sync"]
    # duplicates across batches are kept (no dedupe in Values)
    assert orc.removed_values([[3, 1], [3]]).tolist() == [1, 3, 3]


def test_filter_removed():
    assert orc.filter_removed([1, 2, 3, 4], [2, 2, 4]).tolist() == [1, 3]
    assert orc.filter_removed([1, 2], []).tolist() == [1, 2]
    assert orc.filter_removed([], [1]).tolist() == []
    assert orc.filter_removed([5, 1, 5], [1]).tolist() == [5, 5]      # stable, verbatim order


def test_shard_key_kat(kats):
    for term, key in kats["shard_key"]:
        assert orc.shard_key(term.encode()) == key
    assert orc.shard_key(b"\xff\xff") == 1023


@pytest.mark.parametrize("name", [
    "TestInitFromExistingFiles", "TestIngestion", "TestReadPartial_merged", "TestReadPartial_direct",
    "TestMerging", "TestMergeWithRemoval", "TestMergeEmptySegment", "TestConcurrentAccess_script",
])
def test_shard_scripts(kats, name):
    k = kats["scripts"][name]
    assert k["target"] == "shard"
    run_script(ref_model.Shard(), k["script"], n_segments=lambda s: len(s.segments),
               removed_values=lambda s: s.removed.values())


@pytest.mark.parametrize("name", ["TestPutRemove", "TestPut", "TestSearchByPrefix", "TestReadScoped"])
def test_index_scripts(kats, name):
    k = kats["scripts"][name]
    assert k["target"] == "index"
    run_script(ref_model.InvertedIndex(), k["script"], n_shards=lambda ii: len(ii.shards))


def test_codec_kat_roundtrip(kats):
    # file/writer_test.go:11-46 — unsorted {10,500,300}, empty {}, {66,5513} survive verbatim
    for name, rows in kats["codec"].items():
        lists = [np.asarray(v, np.uint32) for _, v in rows]
        po = np.concatenate([[0], np.cumsum([l.size for l in lists])]).astype(np.uint64)
        flat = np.concatenate(lists) if lists else np.empty(0, np.uint32)
        blk, skip, payload = orc.dv1_encode(po, flat)
        po2, out = orc.dv1_decode(blk, skip, payload, flat.size)
        assert np.array_equal(po2, po) and np.array_equal(out, flat), name


def test_dv1_roundtrip_random():
    rng = np.random.default_rng(3)
    lens = [0, 1, 2, 255, 256, 257, 511, 512, 513, 1000, 0, 5]
    lists = []
    for i, n in enumerate(lens):
        gaps = rng.integers(1, [2, 100, 20000, 3_000_000][i % 4] + 1, n)
        ids = np.cumsum(gaps).astype(np.uint64)
        lists.append((ids % (1 << 32)).astype(np.uint32) if i % 4 != 3 else np.sort(ids.astype(np.uint32)))
    lists.append(np.asarray([0, 0xFFFFFFFF], np.uint32))
    po = np.concatenate([[0], np.cumsum([l.size for l in lists])]).astype(np.uint64)
    flat = np.concatenate(lists)
    blk, skip, payload = orc.dv1_encode(po, flat)
    assert blk[-1] == sum((l.size + 255) // 256 for l in lists)
    po2, out = orc.dv1_decode(blk, skip, payload, flat.size)
    assert np.array_equal(po2, po) and np.array_equal(out, flat)


def _rand_segments(rng, k, T, max_len, universe):
    offs, vals = [], []
    for _ in range(k):
        lens = rng.integers(0, max_len + 1, T)
        lens[rng.random(T) < 0.3] = 0
        o = np.concatenate([[0], np.cumsum(lens)]).astype(np.uint64)
        v = np.concatenate([np.sort(rng.choice(universe, n, replace=False)) for n in lens] + [np.empty(0, np.int64)])
        offs.append(o)
        vals.append(v.astype(np.uint32))
    return offs, vals


def test_merge_segments_vs_numpy():
    rng = np.random.default_rng(4)
    for k, T in [(1, 5), (2, 10), (5, 40), (16, 30)]:
        offs, vals = _rand_segments(rng, k, T, 30, 200)
        removed = np.sort(rng.choice(200, 20)).astype(np.uint32)           # has duplicates
        oo, ov, nsurv = orc.merge_segments(offs, vals, removed)
        surv = 0
        for t in range(T):
            parts = [vals[s][int(offs[s][t]):int(offs[s][t + 1])] for s in range(k)]
            want = np.setdiff1d(np.unique(np.concatenate(parts)), removed)
            assert np.array_equal(ov[int(oo[t]):int(oo[t + 1])], want)
            surv += want.size > 0
        assert nsurv == surv
        o2, v2, n2 = orc.merge_segments(offs, vals, removed, threads=3)
        assert np.array_equal(o2, oo) and np.array_equal(v2, ov) and n2 == nsurv


def test_merge_single_source_passes_verbatim():
    # a term held by exactly one source is not sorted / deduped
    offs = [np.array([0, 3], np.uint64), np.array([0, 0], np.uint64)]
    vals = [np.array([10, 500, 300], np.uint32), np.empty(0, np.uint32)]
    _, ov, _ = orc.merge_segments(offs, vals)
    assert ov.tolist() == [10, 500, 300]
    # present-but-empty second source forces the MergeTermValues fold (sorted)
    pres = [np.array([1], np.uint8), np.array([1], np.uint8)]
    _, ov, _ = orc.merge_segments(offs, vals, present=pres)
    assert ov.tolist() == [10, 300, 500]


def test_union_and_intersect_vs_numpy():
    rng = np.random.default_rng(5)
    for n_lists in [1, 2, 3, 8]:
        lists = [np.sort(rng.choice(3000, rng.integers(0, 1500), replace=False)).astype(np.uint32) for _ in range(n_lists)]
        want_u = np.unique(np.concatenate(lists))
        assert np.array_equal(orc.union(lists), want_u)
        want_i = lists[0]
        for l in lists[1:]:
            want_i = np.intersect1d(want_i, l)
        assert np.array_equal(orc.intersect(lists), want_i)
        removed = np.sort(rng.choice(3000, 300)).astype(np.uint32)
        assert np.array_equal(orc.intersect(lists, removed), np.setdiff1d(want_i, removed))
    assert orc.intersect([[1, 2, 3], []]).size == 0
    assert orc.intersect([[0, 0xFFFFFFFF], [0xFFFFFFFF]]).tolist() == [0xFFFFFFFF]

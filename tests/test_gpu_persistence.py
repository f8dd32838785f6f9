"""GPU: the host mirror on a directory — segment files, removed.list and the segment lifecycle
around them (host/segment_file.h, host_index.cpp), against the reference's own tests of that layer:
shard_test.go:40-63 (TestInitFromExistingFiles), file/writer_test.go:11-84 (TestWriter, TestWriterDirect),
removed_list_test.go:26-37 (TestSerialize), and every replayed shard / index script with the shard re-opened from
its directory before each read.  The file formats are this repo's own (vellum / intcomp bytes are unpinned, SURVEY
§8 c); what is compared is what the reference's tests compare: the decoded (term, values) sequences."""
import os

import numpy as np
import pytest

from oracle import ref_model
from tests.gpu_util import ctx  # noqa: F401
from tests.kat_runner import run_script

pytestmark = pytest.mark.gpu


class Reopening:
    """Forwards the script's operations to a host-mirror target living in `directory`; before every read the target is
    closed and a new one is opened on the same directory (NewShard / NewInvertedIndex load what the files hold)."""

    def __init__(self, cls, ctx, directory):
        self.cls, self.ctx, self.dir = cls, ctx, directory
        self.t = cls(ctx, directory)
        self.reopened = 0

    def reopen(self):
        self.t.close()
        self.t = self.cls(self.ctx, self.dir)
        self.reopened += 1

    def put(self, terms, val):
        self.t.put(terms, val)

    def read(self, lo=None, hi=None):
        self.reopen()
        return self.t.read(lo, hi)

    def merge(self, *a):
        return self.t.merge(*a)

    def remove(self, values):
        self.t.remove(values)

    def put_removed(self, values):
        self.t.put_removed(values)

    def prefix_search(self, prefixes):
        self.reopen()
        return self.t.prefix_search(prefixes)

    def close(self):
        self.t.close()


def seg_files(directory):
    return sorted(f for f in os.listdir(directory) if f.endswith(("_tdx", "_dv1")))


@pytest.mark.parametrize("name", [
    "TestInitFromExistingFiles", "TestIngestion", "TestReadPartial_merged", "TestReadPartial_direct",
    "TestMerging", "TestMergeWithRemoval", "TestMergeEmptySegment", "TestConcurrentAccess_script",
])
def test_shard_scripts_from_files(ctx, kats, name, tmp_path):
    from inverted_index_2_amd.host import Shard
    s = Reopening(Shard, ctx, str(tmp_path))
    run_script(s, kats["scripts"][name]["script"], n_segments=lambda t: t.t.n_segments, removed_values=lambda t: t.t.removed_values())
    assert s.reopened > 0
    assert not [f for f in os.listdir(tmp_path) if f.endswith("_tmp")]
    s.close()


@pytest.mark.parametrize("name", ["TestPutRemove", "TestPut", "TestSearchByPrefix", "TestReadScoped"])
def test_index_scripts_from_files(ctx, kats, name, tmp_path):
    from inverted_index_2_amd.host import InvertedIndex
    ii = Reopening(InvertedIndex, ctx, str(tmp_path))
    run_script(ii, kats["scripts"][name]["script"], n_shards=lambda t: t.t.n_shards)
    assert ii.reopened > 0
    # shards are sub-directories named by shardKey (shard.go:362-378, inverted_index.go:175)
    assert all(len(d) == 4 and d.isdigit() for d in os.listdir(tmp_path))
    ii.close()


def test_writer_reader_roundtrip(ctx, tmp_path):
    # file/writer_test.go:11-46 TestWriter: unsorted values and an empty list come back verbatim
    from inverted_index_2_amd.host import SegmentFiles
    f = SegmentFiles(ctx)
    d = str(tmp_path)
    inp = [(b"term1", [10, 500, 300]), (b"term2", []), (b"term3", [66, 5513])]
    key = f.write(d, inp)
    assert seg_files(d) == [key + "_dv1", key + "_tdx"] and key.isdigit()
    assert f.read(d, key) == inp
    # scoped readers (file/reader.go:136-199): min / max inclusive; nothing in range -> the iterator is done at once
    assert f.read(d, key, b"term2", b"term2") == [inp[1]]
    assert f.read(d, key, b"term10", None) == inp[1:]
    assert f.read(d, key, None, b"term2zzz") == inp[:2]
    assert f.read(d, key, b"u", None) is None and f.read(d, key, None, b"s") is None
    # file/writer_test.go:48-84 TestWriterDirect: one file, one value per term
    key2 = f.write(d, [(b"term1", [10]), (b"term2", [11])], direct=True)
    assert key2 + "_tdx" in seg_files(d) and key2 + "_dv1" not in seg_files(d)
    assert f.read(d, key2) == [(b"term1", [10]), (b"term2", [11])]
    # file.RemoveSegment (file/writer.go:138-147)
    f.remove(d, key)
    f.remove(d, key2)
    assert seg_files(d) == []
    # lists longer than a block, ids up to 2^32 - 1, many terms
    rng = np.random.default_rng(3)
    big = []
    for i in range(300):
        n = int(rng.choice([0, 1, 5, 255, 256, 257, 1000]))
        big.append((b"t%05d" % i, np.unique(rng.integers(0, 1 << 32, n, dtype=np.uint64)).astype(np.int64).tolist()))
    key3 = f.write(d, big)
    assert f.read(d, key3) == big
    f.close()


def test_merge_replaces_the_merged_files(ctx, tmp_path):
    from inverted_index_2_amd.host import Shard
    d = str(tmp_path)
    s = Shard(ctx, d)
    s.put([b"a", b"b"], 1)
    s.put([b"b", b"c"], 2)
    s.put([b"c"], 3)
    direct = seg_files(d)
    assert len(direct) == 3 and all(f.endswith("_tdx") for f in direct)          # direct segments: term file only
    assert s.merge(2, 2) == 2                                                    # the two smallest... by term count: [c] and one more
    after = seg_files(d)
    assert len([f for f in after if f.endswith("_dv1")]) == 1 and len([f for f in after if f.endswith("_tdx")]) == 2
    assert len(set(direct) & set(after)) == 1                                    # two direct files gone, one merged pair written
    want = [(b"a", [1]), (b"b", [1, 2]), (b"c", [2, 3])]
    assert s.read() == want
    s.close()
    s2 = Shard(ctx, d)
    assert s2.n_segments == 2 and s2.read() == want
    assert s2.merge(2, 10) == 2
    assert len(seg_files(d)) == 2 and s2.read() == want
    # a merge that leaves nothing writes nothing (shard.go:219-225) and still removes its inputs
    s2.remove([1, 2, 3])
    s2.put([b"z"], 3)
    assert s2.merge(2, 10) == 2
    assert seg_files(d) == [] and s2.read() == []
    s2.close()
    assert Shard(ctx, d).read() == []


def test_removed_list_survives_reopen(ctx, tmp_path):
    # removed_list_test.go:26-37 TestSerialize + shard.go:78-120: batches are written on every Remove
    from inverted_index_2_amd.host import Shard
    d = str(tmp_path)
    s = Shard(ctx, d)
    assert not os.path.exists(os.path.join(d, "removed.list"))
    s.put([b"x"], 7)
    s.remove([1, 5, 10])
    s.remove([2, 20, 30, 5])
    assert os.path.exists(os.path.join(d, "removed.list"))
    assert s.removed_values() == [1, 2, 5, 5, 10, 20, 30]
    s.close()
    s = Shard(ctx, d)
    assert s.removed_values() == [1, 2, 5, 5, 10, 20, 30]
    # Sync drops batches older than every segment (removed_list.go:57-71): merge away the old segment, remove again
    s.put([b"y"], 8)
    assert s.merge(2, 2) == 2
    s.remove([99])
    assert s.removed_values() == [99]
    s.close()
    assert Shard(ctx, d).removed_values() == [99]


def test_corrupt_and_foreign_files(ctx, tmp_path):
    from inverted_index_2_amd.host import HostError, SegmentFiles, Shard
    d = str(tmp_path)
    f = SegmentFiles(ctx)
    key = f.write(d, [(b"k1", [1, 2, 3]), (b"k2", [4])])
    open(os.path.join(d, "123_tdx_tmp"), "wb").write(b"half written")             # leftovers of a crashed writer: ignored
    open(os.path.join(d, "notes.txt"), "w").write("not a segment")
    assert Shard(ctx, d).read() == [(b"k1", [1, 2, 3]), (b"k2", [4])]
    for suffix in ("_dv1", "_tdx"):
        p = os.path.join(d, key + suffix)
        good = open(p, "rb").read()
        bad = bytearray(good)
        bad[len(bad) // 2] ^= 0x40
        open(p, "wb").write(bytes(bad))
        with pytest.raises(HostError, match="checksum"):
            Shard(ctx, d)
        open(p, "wb").write(good[:-9])
        with pytest.raises(HostError):
            Shard(ctx, d)
        open(p, "wb").write(good)
    os.rename(os.path.join(d, key + "_dv1"), os.path.join(d, key + "_dv1.bak"))
    with pytest.raises(HostError):                                                # value file missing for a non-direct segment
        Shard(ctx, d)
    os.rename(os.path.join(d, key + "_dv1.bak"), os.path.join(d, key + "_dv1"))
    open(os.path.join(d, "notanumber_tdx"), "wb").write(open(os.path.join(d, key + "_tdx"), "rb").read())
    with pytest.raises(HostError, match="key to int"):                            # shard.go:92-96: keys are unix nanoseconds
        Shard(ctx, d)
    os.remove(os.path.join(d, "notanumber_tdx"))
    open(os.path.join(d, "removed.list"), "wb").write(b"garbage garbage garbage")
    with pytest.raises(HostError, match="rem list"):
        Shard(ctx, d)
    with pytest.raises(HostError):
        Shard(ctx, os.path.join(d, "missing-directory"))
    f.close()


def test_random_workload_on_disk_matches_reference_model(ctx, tmp_path):
    from inverted_index_2_amd.host import InvertedIndex
    rng = np.random.default_rng(43)
    vocab = [bytes(rng.choice(list(b"abcdefghijklmnopqrstuvwxyzABCDEFGHIJKLMNOPQRSTUVWXYZ"), int(rng.integers(2, 8))).tolist())
             for _ in range(100)]
    d = str(tmp_path)
    gpu, ref = InvertedIndex(ctx, d), ref_model.InvertedIndex()
    for step in range(120):
        terms = [vocab[i] for i in rng.choice(len(vocab), int(rng.integers(1, 6)), replace=False)]
        val = int(rng.integers(0, 60))
        gpu.put(list(terms), val)
        ref.put(list(terms), val)
        if step % 37 == 36:
            rem = rng.integers(0, 60, 5).tolist()
            gpu.put_removed(rem)
            ref.put_removed(rem)
        if step % 25 == 24:
            assert gpu.merge(2, 4, 2) == ref.merge(2, 4, 2)
        if step % 40 == 39:                                  # a restart in the middle of the run
            gpu.close()
            gpu = InvertedIndex(ctx, d)
            for sh in ref.shards.values():                   # the reference's merging flags are in-memory atomics: a restart clears them
                for sg in sh.segments:
                    sg.merging = False
            assert gpu.read() == [(t, [int(v) for v in vs]) for t, vs in ref.read()]
    gpu.close()
    gpu = InvertedIndex(ctx, d)
    want = [(t, [int(v) for v in vs]) for t, vs in ref.read()]
    assert gpu.read() == want
    assert gpu.read(b"b", b"q") == [(t, [int(v) for v in vs]) for t, vs in ref.read(b"b", b"q")]
    assert gpu.prefix_search([b"a", b"Zq", b"m"]) == ref.prefix_search([b"a", b"Zq", b"m"])
    gpu.close()

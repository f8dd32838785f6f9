"""CPU: the parts of the host mirror's file layer that involve no device (host/segment_file.h through libii2_host.so):
the term dictionary file of a direct segment (file/writer_test.go:48-84 TestWriterDirect at the file level), removed.list
(removed_list_test.go:9-37 TestRemovedLists / TestSerialize), temp-name + rename, checksums."""
import os

import pytest


@pytest.fixture()
def files():
    from inverted_index_2_amd.host import SegmentFiles
    f = SegmentFiles(None)
    yield f
    f.close()


def test_direct_segment_term_file(files, tmp_path):
    from inverted_index_2_amd.host import HostError
    d = str(tmp_path)
    inp = [(b"term1", [10]), (b"term2", [11]), (b"", [0]), (b"\x00\xff", [4294967295])]
    inp.sort()
    key = files.write(d, inp, direct=True)
    assert sorted(os.listdir(d)) == [key + "_tdx"] and key.isdigit()          # one file, final name only
    assert files.read_terms(d, key) == (True, inp)
    with pytest.raises(HostError):                                            # a non-direct writer needs the device encode step
        files.write(d, inp)
    with pytest.raises(HostError, match="one value per term"):
        files.write(d, [(b"t", [])], direct=True)
    p = os.path.join(d, key + "_tdx")
    good = open(p, "rb").read()
    for bad in (good[:20], good[:-1], good[:40] + bytes([good[40] ^ 1]) + good[41:], b"II2XXXX\0" + good[8:], b""):
        open(p, "wb").write(bad)
        with pytest.raises(HostError):
            files.read_terms(d, key)
    open(p, "wb").write(good)
    assert files.read_terms(d, key) == (True, inp)
    files.remove(d, key)
    assert os.listdir(d) == []
    files.remove(d, key)                                                      # RemoveSegment of a missing segment is not an error here
    with pytest.raises(HostError):
        files.read_terms(d, key)


def test_removed_list_file(files, tmp_path):
    from inverted_index_2_amd.host import HostError
    d = str(tmp_path)
    assert files.read_removed(d) == (0, [])                                   # no file: an empty list (shard.go:338-342)
    t1, t2 = 1_700_000_000_000_000_001, 1_700_000_000_000_000_002
    files.write_removed(d, {t1: [1, 5, 10], t2: [2, 20, 30]})                 # removed_list_test.go:12-18
    assert files.read_removed(d) == (2, [1, 2, 5, 10, 20, 30])
    files.write_removed(d, {t2: [2, 20, 30]})                                 # after Sync dropped the older batch (:20-23)
    assert files.read_removed(d) == (1, [2, 20, 30])
    files.write_removed(d, {t2: [], 5: [7, 7]})                               # empty batches and duplicates survive as they are
    assert files.read_removed(d) == (2, [7, 7])
    assert sorted(os.listdir(d)) == ["removed.list"]                          # no temp file left behind
    p = os.path.join(d, "removed.list")
    good = open(p, "rb").read()
    open(p, "wb").write(good[:-3])
    with pytest.raises(HostError):
        files.read_removed(d)
    open(p, "wb").write(good[:24] + bytes([good[24] ^ 0x80]) + good[25:])
    with pytest.raises(HostError, match="checksum"):
        files.read_removed(d)

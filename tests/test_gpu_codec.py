"""GPU: DV1 encode / decode kernels against the oracle codec (byte-exact) — through the C ABI."""
import numpy as np
import pytest

from oracle import oracle as orc
from tests.gpu_util import ctx, sorted_unique  # noqa: F401

pytestmark = pytest.mark.gpu


def test_selftest(ctx):
    ctx.selftest()


def _lists(rng):
    lens = [0, 1, 2, 3, 255, 256, 257, 511, 512, 513, 1000, 0, 5, 70000]
    gaps_hi = [2, 100, 20000, 3_000_000]
    out = []
    for i, n in enumerate(lens):
        g = rng.integers(1, gaps_hi[i % 4] + 1, n, dtype=np.int64)
        ids = np.cumsum(g)
        ids = ids[ids < (1 << 32)]
        out.append(ids.astype(np.uint32))
    out.append(np.asarray([0, 0xFFFFFFFF], np.uint32))
    out.append(np.asarray([0xFFFFFFFF], np.uint32))
    return out


def test_encode_matches_oracle_bytes_and_roundtrips(ctx):
    rng = np.random.default_rng(11)
    lists = _lists(rng)
    po = np.concatenate([[0], np.cumsum([l.size for l in lists])]).astype(np.uint64)
    flat = np.concatenate(lists)
    seg = ctx.encode(po, flat)
    blk, skip, payload = seg.export()
    oblk, oskip, opayload = orc.dv1_encode(po, flat)
    assert np.array_equal(blk, oblk)
    assert np.array_equal(skip["first_doc"][:-1], oskip["first_doc"][:-1])
    assert np.array_equal(skip["byte_off"], oskip["byte_off"])
    assert np.array_equal(payload, opayload)
    po2, vals = seg.decode()
    assert np.array_equal(po2, po) and np.array_equal(vals, flat)


def test_import_oracle_encoded_and_decode(ctx):
    rng = np.random.default_rng(12)
    lists = _lists(rng)
    po = np.concatenate([[0], np.cumsum([l.size for l in lists])]).astype(np.uint64)
    flat = np.concatenate(lists)
    oblk, oskip, opayload = orc.dv1_encode(po, flat)
    seg = ctx.import_dv1(flat.size, oblk, oskip, opayload)
    po2, vals = seg.decode()
    assert np.array_equal(po2, po) and np.array_equal(vals, flat)


def test_codec_kats_verbatim(ctx, kats):
    # file/writer_test.go:11-46 — unsorted {10,500,300}, empty, {66,5513} survive write->read verbatim
    for name, rows in kats["codec"].items():
        lists = [np.asarray(v, np.uint32) for _, v in rows]
        seg = ctx.encode_lists(lists)
        po, vals = seg.decode()
        for i, l in enumerate(lists):
            assert np.array_equal(vals[int(po[i]):int(po[i + 1])], l), (name, i)


def test_many_tiny_lists(ctx):
    rng = np.random.default_rng(13)
    lens = rng.integers(0, 9, 20000)
    lists = [sorted_unique(rng, int(n), 1 << 20) for n in lens]
    po = np.concatenate([[0], np.cumsum([l.size for l in lists])]).astype(np.uint64)
    flat = np.concatenate(lists)
    seg = ctx.encode(po, flat)
    po2, vals = seg.decode()
    assert np.array_equal(po2, po) and np.array_equal(vals, flat)
    oblk, oskip, opayload = orc.dv1_encode(po, flat)
    blk, skip, payload = seg.export()
    assert np.array_equal(blk, oblk) and np.array_equal(payload, opayload)


def test_tombstone_bitmap_filters_like_binary_search(ctx):
    rng = np.random.default_rng(14)
    a = sorted_unique(rng, 5000, 100000)
    removed = rng.integers(0, 100000, 3000).astype(np.uint32)     # unsorted, with duplicates
    seg = ctx.encode_lists([a])
    out, n = ctx.intersect([(seg, 0)], tomb=ctx.tombstones(removed))
    assert np.array_equal(out.download(n), orc.filter_removed(a, np.sort(removed)))


def test_import_rejects_malformed_segment(ctx):
    from inverted_index_2_amd import II2Error
    a = np.arange(0, 3000, 3, dtype=np.uint32)
    po = np.array([0, a.size], np.uint64)
    blk, skip, payload = orc.dv1_encode(po, a)
    bad = skip.copy()
    bad["byte_off"][1] = 10_000_000          # points far outside the payload
    with pytest.raises(II2Error) as e:
        ctx.import_dv1(a.size, blk, bad, payload)
    assert e.value.code == -1
    bad2 = blk.copy()
    bad2[1] = 999
    with pytest.raises(II2Error):
        ctx.import_dv1(a.size, bad2, skip, payload)
    # the arrays' lengths are part of the call: closing entries that disagree with them are refused before anything is
    # read past the arrays (blk_off[n_lists] = 999 above used to index the 5-entry skip table with 999)
    with pytest.raises(II2Error) as e:
        ctx.import_dv1(a.size, blk, skip, payload[:-7])          # payload shorter than skip[n_blocks].byte_off
    assert e.value.code == -1
    with pytest.raises(II2Error) as e:
        ctx.import_dv1(a.size, blk, skip[:-1], payload)          # skip table one entry short
    assert e.value.code == -1
    ctx.import_dv1(a.size, blk, skip, payload)      # the intact one is accepted
    # n_postings sizes the decode buffers: a count that differs from what the blocks hold is refused either way
    for wrong in (a.size - 1, a.size + 1, 0):
        with pytest.raises(II2Error) as e:
            ctx.import_dv1(wrong, blk, skip, payload)
        assert e.value.code == -1
    # block fill rule: every block but a list's last holds 256 postings (the merge places decoded blocks by it).
    # Build a 3-block list by hand whose MIDDLE block is short: structurally fine, but refused.
    def enc(ids):
        gaps = np.diff(ids.astype(np.int64))
        assert gaps.max() < 128
        return gaps.astype(np.uint8).tobytes()
    b0, b1, b2 = np.arange(0, 256), np.arange(1000, 1100), np.arange(5000, 5256)
    pay = enc(b0) + enc(b1) + enc(b2)
    sk = np.zeros(4, dtype=skip.dtype)
    sk["first_doc"] = [0, 1000, 5000, 0]
    sk["byte_off"] = [0, 255, 255 + 99, 255 + 99 + 255]
    with pytest.raises(II2Error) as e:
        ctx.import_dv1(256 + 100 + 256, np.array([0, 3], np.uint32), sk, np.frombuffer(pay, np.uint8))
    assert e.value.code == -1


def test_allgatherv_single_rank_is_a_copy(ctx):
    a = np.arange(10, 5000, 7, dtype=np.uint32)
    src = ctx.empty(a.size).upload(a)
    dst = ctx.empty(a.size + 10)
    counts = ctx.allgatherv(src, a.size, dst, 1)
    assert counts == [a.size] and np.array_equal(dst.download(a.size), a)


def test_segment_allgather_with_one_rank_is_a_copy(ctx):
    """ii2_seg_allgather without a communicator: the same shapes / gather / shift / closing-entry steps as the N-rank
    exchange, with one contribution — the result decodes to the same lists and merges like the original."""
    rng = np.random.default_rng(21)
    lists = [sorted_unique(rng, int(n), 5_000_000) for n in (0, 1, 300, 70_000, 0, 257, 5)]
    seg = ctx.encode_lists(lists)
    got = ctx.seg_allgather(seg)
    assert (got.info.n_lists, got.info.n_postings, got.info.n_blocks, got.info.n_bytes) == \
        (seg.info.n_lists, seg.info.n_postings, seg.info.n_blocks, seg.info.n_bytes)
    po, v = got.decode()
    po0, v0 = seg.decode()
    assert np.array_equal(po, po0) and np.array_equal(v, v0)
    out_off, out_vals, st = ctx.merge([got, seg])
    assert st.n_out == sum(l.size for l in lists)
    # byte-wise all-gatherv, one rank
    a = ctx.empty(1000, np.uint8).upload(np.arange(1000) % 251)
    b = ctx.empty(1200, np.uint8)
    assert ctx.allgatherv_bytes(a, 777, b, 1) == [777]
    assert np.array_equal(b.download(777), (np.arange(777) % 251).astype(np.uint8))


def test_segments_concatenate_like_the_exchange_would(ctx):
    """ii2_seg_concat is the exchange's arithmetic on one device (ii2_seg_gather_plan + copies + the shift of block numbers, byte
    offsets and block owners that ii2_seg_allgather applies to the parts it receives): three segments - one of them empty, one
    with empty lists at its ends - must concatenate into, byte for byte, the encoding of the concatenated lists; views are refused."""
    rng = np.random.default_rng(21)
    parts = []
    for n_lists, max_len in ((40, 900), (0, 0), (25, 3000), (7, 5)):
        lens = rng.integers(0, max_len + 1, n_lists)
        if n_lists:
            lens[0] = 0; lens[-1] = 0
        lists = [sorted_unique(rng, int(n), 1 << 24) for n in lens]
        po = np.concatenate([[0], np.cumsum([x.size for x in lists])]).astype(np.uint64)
        flat = np.concatenate(lists + [np.empty(0, np.uint32)]).astype(np.uint32)
        parts.append((po, flat))
    segs = [ctx.encode(po, flat) for po, flat in parts]
    before = ctx.counters()[2]
    cat = ctx.seg_concat(segs)
    assert ctx.counters()[2] == before + 1                      # one host wait for the whole concatenation
    po_all = np.concatenate([[0]] + [po[1:] + sum(int(q[0][-1]) for q in parts[:i]) for i, (po, _) in enumerate(parts)]).astype(np.uint64)
    flat_all = np.concatenate([f for _, f in parts]).astype(np.uint32)
    oblk, oskip, opayload = orc.dv1_encode(po_all, flat_all)
    blk, skip, payload = cat.export()
    assert np.array_equal(blk, oblk) and np.array_equal(skip["first_doc"][:-1], oskip["first_doc"][:-1]) and np.array_equal(skip["byte_off"], oskip["byte_off"])
    assert np.array_equal(payload, opayload)
    gpo, gv = cat.decode()
    assert np.array_equal(gpo, po_all) and np.array_equal(gv, flat_all)
    # the derived arrays travelled and were shifted too: a query against a list of the last part finds it
    li = parts[0][0].size - 1 + parts[2][0].size - 1 + 3
    a = flat_all[int(po_all[li]):int(po_all[li + 1])]
    out, n = ctx.intersect([(cat, li), (cat, li)])
    assert n == a.size and np.array_equal(out.download(n), a)
    merged, _ = ctx.merge_to_segment([cat, cat])                # and the merge reads it like any other segment
    mpo, mv = merged.decode()
    assert np.array_equal(mpo, po_all) and np.array_equal(mv, flat_all)
    from inverted_index_2_amd.engine import II2Error
    view = ctx.select(segs[0], np.arange(3, 10, dtype=np.int64))
    with pytest.raises(II2Error):
        ctx.seg_concat([view, segs[2]])


def test_a_view_of_some_lists_knows_its_own_size(ctx):
    """ii2_seg_select: a view shares its store's skip table and payload, but what a merge of it can read is its own lists -
    its posting count is exact (a merge sizes its plan, its scratch and its launches by it: a view that reported its store's
    total made every merge of a few lists of a big segment plan for the whole segment), and merging views gives what merging
    the same lists as segments of their own gives."""
    rng = np.random.default_rng(21)
    lens = [0, 5, 300, 0, 7000, 1, 0, 64, 257, 0, 90_000, 3]
    lists = [sorted_unique(rng, n, 1 << 26) for n in lens]
    other = [x[::3].copy() for x in lists]
    a, b = ctx.encode_lists(lists), ctx.encode_lists(other)
    for lo, hi in ((0, len(lens)), (2, 6), (4, 5), (6, 7), (9, 12), (0, 1)):
        idx = np.arange(lo, hi, dtype=np.int64)
        va, vb = ctx.select(a, idx), ctx.select(b, idx)
        assert va.info.n_lists == hi - lo and va.info.n_postings == sum(x.size for x in lists[lo:hi])
        assert vb.info.n_postings == sum(x.size for x in other[lo:hi])
        oo, ov, st = ctx.merge([va, vb])
        sa, sb = ctx.encode_lists(lists[lo:hi]), ctx.encode_lists(other[lo:hi])
        wo, wv, wst = ctx.merge([sa, sb])
        assert st.n_out == wst.n_out and np.array_equal(oo.download(), wo.download()) and np.array_equal(ov.download(int(st.n_out)), wv.download(int(wst.n_out)))
        m, _ = ctx.merge_to_segment([va, vb])
        if st.n_out:
            po, v = m.decode()
            assert np.array_equal(po, wo.download()) and np.array_equal(v, wv.download(int(wst.n_out)))
        else:
            assert m is None


def test_freed_segment_arrays_are_reused_not_returned_to_the_driver(ctx):
    """devmem.cpp: a freed segment's device arrays wait in a size-class cache; making the same segment again takes them out
    of it instead of allocating (ii2_devmem_stats: live bytes come back to the same value, idle bytes do not grow)."""
    import ctypes as C
    rng = np.random.default_rng(3)
    lists = [np.unique(rng.integers(0, 1 << 24, 50_000)).astype(np.uint32) for _ in range(20)]

    def stats():
        live, idle = C.c_uint64(), C.c_uint64()
        ctx.lib.ii2_devmem_stats(C.byref(live), C.byref(idle))
        return live.value, idle.value
    seg = ctx.encode_lists(lists)
    live1, idle1 = stats()
    seg.free()
    live0, idle0 = stats()
    assert live0 < live1 and idle0 > idle1                      # the arrays moved from "handed out" to "waiting"
    for _ in range(5):
        seg = ctx.encode_lists(lists)
        assert stats() == (live1, idle1)                        # the same arrays again: nothing new, nothing left over
        _, v = seg.decode()
        assert np.array_equal(v, np.concatenate(lists))
        seg.free()
        assert stats() == (live0, idle0)


def test_idle_segment_arrays_go_back_to_the_driver_when_another_allocation_needs_the_room(ctx):
    """The cache keeps freed segment arrays; every OTHER device allocation of the library (user buffers, workspaces, staging
    pools, dictionaries) goes through a helper that, when hipMalloc fails, gives the idle arrays back and tries once more.
    Fill HBM with ii2_dev_alloc until it fails: by then the cache must be empty (ii2_devmem_stats), and once the big buffers
    are released the library works as before."""
    import ctypes as C
    from inverted_index_2_amd.engine import II2Error
    rng = np.random.default_rng(4)
    lists = [np.unique(rng.integers(0, 1 << 26, 400_000)).astype(np.uint32) for _ in range(8)]

    def idle():
        live, idl = C.c_uint64(), C.c_uint64()
        ctx.lib.ii2_devmem_stats(C.byref(live), C.byref(idl))
        return idl.value
    ctx.encode_lists(lists).free()
    assert idle() > 0                                            # arrays are waiting in the cache
    hogs, chunk = [], 16 << 30
    try:
        for _ in range(64):                                      # 288 GB of HBM: fails long before 64 x 16 GB
            try:
                hogs.append(ctx.empty(chunk, np.uint8))
            except II2Error as e:
                assert e.code == -2                              # II2_ENOMEM
                break
        else:
            pytest.skip("the device took 1 TB of allocations: nothing to test here")
        assert idle() == 0                                       # the failing attempt gave the cache back before it gave up
    finally:
        for h in hogs:
            h.free()
    seg = ctx.encode_lists(lists)
    _, v = seg.decode()
    assert np.array_equal(v, np.concatenate(lists))
    seg.free()

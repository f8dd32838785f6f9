"""GPU: the C library is re-entrant and segments are shared between contexts (SURVEY §8 b "Threading";
InvertedIndex.Merge's worker pool inverted_index.go:83-103, readers sharing segments segments.go:32-46):
8 threads x 8 contexts merge 8 different shards and intersect THE SAME two lists at the same time."""
import threading

import numpy as np
import pytest

from inverted_index_2_amd import Context, synth
from oracle import oracle as orc
from tests.gpu_util import ctx  # noqa: F401

pytestmark = pytest.mark.gpu


def test_eight_threads_eight_contexts_share_segments(ctx):
    D = 3_000_000
    a, b = synth.zipf_list(2, D), synth.zipf_list(5, D)
    shared = ctx.encode_lists([a, b])                      # made by one context, read by all
    removed = np.arange(0, D, 41, dtype=np.uint32)
    shared_tomb = ctx.tombstones(removed)
    want_and = orc.intersect([a, b], removed)
    shards = []
    for i in range(8):
        offs, vals, rem = synth.merge_workload(3000, 6, 80, 400_000, seed=1000 + i)
        shards.append((offs, vals, rem, [ctx.encode(o, v) for o, v in zip(offs, vals)], ctx.tombstones(rem)))
    serial = []
    for offs, vals, rem, segs, tomb in shards:
        oo, ov, st = ctx.merge(segs, tomb)
        serial.append((oo.download(), ov.download(int(st.n_out))))
        w_off, w_vals, _ = orc.merge_segments(offs, vals, rem)
        assert np.array_equal(serial[-1][0], w_off) and np.array_equal(serial[-1][1], w_vals)
    workers = [Context(0) for _ in range(8)]
    errors, barrier = [], threading.Barrier(8)

    def run(i):
        try:
            c = workers[i]
            barrier.wait()
            for rep in range(6):
                _, _, _, segs, tomb = shards[(i + rep) % 8]
                oo, ov, st = c.merge(segs, tomb)           # segments / tombstones created by `ctx`, used by `c`
                g = (oo.download(), ov.download(int(st.n_out)))
                s = serial[(i + rep) % 8]
                assert np.array_equal(g[0], s[0]) and np.array_equal(g[1], s[1]), ("merge", i, rep)
                out, n = c.intersect([(shared, 0), (shared, 1)], tomb=shared_tomb)
                assert n == want_and.size and np.array_equal(out.download(n), want_and), ("intersect", i, rep)
                m, _ = c.merge_to_segment(segs, tomb)      # a segment born in a worker context ...
                po, v = m.decode()
                assert np.array_equal(po, s[0]) and np.array_equal(v, s[1])
                m.free()
        except BaseException as e:  # noqa: BLE001
            errors.append((i, repr(e)))

    ts = [threading.Thread(target=run, args=(i,)) for i in range(8)]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    assert not errors, errors
    # ... and a segment made by a worker context outlives that context
    seg_w = workers[3].encode_lists([a, b])
    for w in workers:
        w.close()
    out, n = ctx.intersect([(seg_w, 0), (seg_w, 1)], tomb=shared_tomb)
    assert np.array_equal(out.download(n), want_and)


def test_kernels_that_wait_between_workgroups_do_not_starve_each_other_across_contexts(ctx):
    """The one-pass encoder behind a merge, the one-launch two-list AND and the merge's direct placement let workgroups wait for
    lower-numbered workgroups of their own launch.  Two such kernels from two contexts side by side can hold each other's slots
    (seen: three contexts encoding at once - every encoder ran out its bounded waits, seconds each, and handed over to its
    second path): the library orders them per device (api.cpp: ii2_lookback_launch).  Four contexts on four threads, launches big
    enough to fill the chip several times over: right results, and NO launch repeated on its second path."""
    D = 40_000_000
    a, b = synth.zipf_list(2, D), synth.zipf_list(3, D)                 # dense: the one-launch AND (~4000 workgroups... of 256 docs x 16 blocks)
    shared = ctx.encode_lists([a, b])
    want_and = np.intersect1d(a, b, assume_unique=True)
    offs, vals, rem = synth.merge_workload(150_000, 8, 120, 50_000_000, seed=77)       # ~18M postings: ~2000 encoder workgroups
    segs = [ctx.encode(o, v) for o, v in zip(offs, vals)]
    tomb = ctx.tombstones(rem)
    ref_seg, _ = ctx.merge_to_segment(segs, tomb)
    ref_off, ref_vals = ref_seg.decode()
    ref_seg.free()
    workers = [Context(0) for _ in range(4)]
    errors, barrier = [], threading.Barrier(4)

    def run(i):
        try:
            c = workers[i]
            barrier.wait()
            for rep in range(5):
                m, st = c.merge_to_segment(segs, tomb)
                assert st.n_out == ref_vals.size
                out, n = c.intersect([(shared, 0), (shared, 1)])
                assert n == want_and.size
                if rep == 4:
                    po, v = m.decode()
                    assert np.array_equal(po, ref_off) and np.array_equal(v, ref_vals), ("merge_to_segment", i)
                    assert np.array_equal(out.download(n), want_and), ("intersect", i)
                m.free()
        except BaseException as e:  # noqa: BLE001
            errors.append((i, repr(e)))

    ts = [threading.Thread(target=run, args=(i,)) for i in range(4)]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    assert not errors, errors
    repeats = [w.counters()[:2] for w in workers]
    assert all(r == (0, 0) for r in repeats), repeats
    for w in workers:
        w.close()


def test_host_mirror_merge_fans_out_over_workers(ctx):
    # InvertedIndex.Merge(reqCount, mCount, concurrency): same merged count and content with 1 and with 6 workers
    from inverted_index_2_amd.host import InvertedIndex
    rng = np.random.default_rng(5)
    vocab = [bytes([65 + i % 26, 97 + (i * 7) % 26, 97 + (i * 3) % 26]) for i in range(300)]      # ~26 shards
    results = []
    for conc in (1, 6):
        ii = InvertedIndex(ctx)
        r2 = np.random.default_rng(9)
        for d in range(120):
            ii.put([vocab[j] for j in r2.choice(len(vocab), 12, replace=False)], d)
        ii.put_removed([3, 50, 77])
        merged = []
        while True:
            m = ii.merge(2, 8, conc)
            merged.append(m)
            if m == 0:
                break
        results.append((merged, ii.read()))
        assert ii.merge(2, 8, 0) == 0                     # no workers: nothing merged (the reference starts none)
        ii.close()
    assert results[0] == results[1]


# ---- the host mirror's own thread safety (segments.go:26-54 locks, shard.go:134-146 merging flags) ----------------
def test_concurrent_access_on_one_shard(ctx, tmp_path):
    # shard_test.go:216-248 TestConcurrentAccess: many threads run the same ingest / merge / compare sequence on ONE shard
    import threading
    from inverted_index_2_amd.host import Shard
    shard = Shard(ctx, str(tmp_path))
    want = [(b"term1", [1, 2]), (b"term2", [2]), (b"term3", [3])]
    errors = []
    begin = threading.Event()

    def run(sess):
        try:
            begin.wait()
            sess.put([b"term1"], 1)
            sess.put([b"term1", b"term2"], 2)
            sess.put([b"term3"], 3)
            for _ in range(3):
                sess.merge(2, 2)
            got = sess.read()
            assert got == want, got
        except Exception as e:  # noqa: BLE001
            errors.append(repr(e))
        finally:
            sess.close()

    threads = [threading.Thread(target=run, args=(shard.session(),)) for _ in range(16)]
    for t in threads:
        t.start()
    begin.set()
    for t in threads:
        t.join()
    assert not errors, errors[:3]
    while shard.merge(2, 100):
        pass
    assert shard.read() == want
    shard.close()
    # what the directory holds after the storm is the same index
    again = Shard(ctx, str(tmp_path))
    assert again.read() == want
    assert not [f for f in __import__("os").listdir(tmp_path) if f.endswith("_tmp")]
    again.close()


def test_concurrent_put_read_merge_on_an_index(ctx):
    # inverted_index_test.go:84-138 TestConcurrent: writers and readers at once, merges until nothing is left to merge
    import threading
    from inverted_index_2_amd.host import InvertedIndex
    ii = InvertedIndex(ctx)
    rng = np.random.default_rng(9)
    letters = list(b"abcdefghijklmnopqrstuvwxyzABCDEFGHIJKLMNOPQRSTUVWXYZ")
    plan = [[(sorted(bytes(rng.choice(letters, int(rng.integers(10, 20))).tolist()) for _ in range(3)), i) for _ in range(int(rng.integers(1, 12)))]
            for i in range(12)]
    errors = []

    def writer(sess, puts):
        try:
            for terms, val in puts:
                sess.put(list(terms), val)
        except Exception as e:  # noqa: BLE001
            errors.append(repr(e))
        finally:
            sess.close()

    def reader(sess):
        try:
            for _ in range(3):
                got = sess.read()
                terms = [t for t, _ in got]                   # whatever moment the read caught: ascending terms, ascending unique ids
                assert terms == sorted(set(terms)) and all(v == sorted(set(v)) and v for _, v in got)
                sess.merge(2, 4, 2)
        except Exception as e:  # noqa: BLE001
            errors.append(repr(e))
        finally:
            sess.close()

    threads = [threading.Thread(target=writer, args=(ii.session(), p)) for p in plan] + [threading.Thread(target=reader, args=(ii.session(),)) for _ in range(6)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors, errors[:3]
    while ii.merge(2, 100, 2):
        pass
    expect = {}
    for puts in plan:
        for terms, val in puts:
            for t in terms:
                expect.setdefault(t, set()).add(val)
    got = dict(ii.read())
    assert got == {t: sorted(v) for t, v in expect.items()}
    ii.close()

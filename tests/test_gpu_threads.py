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

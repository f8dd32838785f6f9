"""GPU probe: latency of the one-launch small-shard merge (ii2_merge_small) next to the general path (device alignment +
aligned views + tombstones + merge-to-segment + empty-term compaction), on direct segments like the ones Shard.Put writes."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from inverted_index_2_amd import Context
ctx = Context(0)
rng = np.random.default_rng(2)
vocab = sorted(b"term%04d" % i for i in range(400))
for k, terms in ((2, 3), (8, 3), (32, 5), (8, 60)):
    dicts = [sorted(vocab[i] for i in rng.choice(len(vocab), terms, replace=False)) for _ in range(k)]
    lists = [[np.array([d], np.uint32) for _ in dd] for d, dd in enumerate(dicts)]
    segs = [ctx.encode_lists(ls) for ls in lists]
    removed = np.array([1], np.uint32)
    for _ in range(3): ctx.merge_small(segs, dicts, removed)
    ts = []
    for _ in range(30):
        t = time.perf_counter(); seg, tt, st = ctx.merge_small(segs, dicts, removed); ts.append(time.perf_counter() - t)
    tg = []
    for _ in range(10):
        t = time.perf_counter()
        al = ctx.align_terms(dicts)
        views = [ctx.select_aligned(s, al, i) for i, s in enumerate(segs)]
        tomb = ctx.tombstones(removed)
        m, st2 = ctx.merge_to_segment(views, tomb)
        po, _ = m.decode()
        keep = np.flatnonzero(np.diff(po.astype(np.int64)) > 0)
        c = ctx.select(m, keep)
        tg.append(time.perf_counter() - t)
    print(f"{k:3d} segments x {terms:3d} terms: one launch {np.median(ts)*1e6:7.1f} us (min {min(ts)*1e6:6.1f}), general path {np.median(tg)*1e6:8.1f} us; "
          f"{int(st.n_terms_out)} terms, {int(st.n_out)} postings out", flush=True)

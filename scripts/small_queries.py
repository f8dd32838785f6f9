"""GPU probe: latency of small queries (2-term AND / OR of short lists), back to back on one stream."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from inverted_index_2_amd import Context
ctx = Context(0)
rng = np.random.default_rng(1)
U = 100_000_000
for n in (10, 100, 1000, 10_000, 100_000, 1_000_000):
    a = np.unique(rng.integers(0, U, n, dtype=np.uint64)).astype(np.uint32)
    b = np.unique(np.concatenate([rng.integers(0, U, 4 * n, dtype=np.uint64), a[::3].astype(np.uint64)])).astype(np.uint32)
    seg = ctx.encode_lists([a, b])
    ls = [(seg, 0), (seg, 1)]
    out = ctx.empty(a.size + b.size + 16); cnt = ctx.empty(8, np.uint64)
    _, k = ctx.intersect(ls, out=out)
    assert np.array_equal(out.download(k), np.intersect1d(a, b, assume_unique=True))
    res = []
    for op in ("intersect", "union"):
        fn = (lambda: ctx.intersect_async(ls, None, out, cnt)) if op == "intersect" else (lambda: ctx.union(ls, out=out))
        for _ in range(5): fn()
        ctx.sync(); t = time.perf_counter()
        for _ in range(200): fn()
        ctx.sync(); res.append((time.perf_counter() - t) / 200 * 1e6)
    print(f"lists of {a.size:>8d} and {b.size:>8d} postings: AND {res[0]:7.1f} us   OR {res[1]:7.1f} us (OR returns its count: one sync per call)", flush=True)
    seg.free(); out.free(); cnt.free()
ctx.close()
# many short lists (PrefixSearch: the lists of every term with the prefix)
ctx = Context(0)
for k, n in ((8, 100), (32, 100), (64, 100), (64, 20), (16, 500)):
    lists = [np.unique(rng.integers(0, 1_000_000, n, dtype=np.uint64)).astype(np.uint32) for _ in range(k)]
    seg = ctx.encode_lists(lists)
    ls = [(seg, i) for i in range(k)]
    out = ctx.empty(sum(l.size for l in lists) + 16)
    _, m = ctx.union(ls, out=out)
    assert np.array_equal(out.download(m), np.unique(np.concatenate(lists)))
    res, dev = [], []
    ctx.set_option("profile.events", 1)
    for small in (1, 0):
        ctx.set_option("setop.small", small)
        for _ in range(5): ctx.union(ls, out=out)
        ctx.sync(); ctx.profile_read(); t = time.perf_counter()
        for _ in range(100): ctx.union(ls, out=out)
        ctx.sync(); res.append((time.perf_counter() - t) / 100 * 1e6)
        ms, cnt = ctx.profile_read(); dev.append(ms / max(cnt, 1) * 1e3)
    ctx.set_option("setop.small", 1); ctx.set_option("profile.events", 0)
    print(f"OR of {k:2d} lists x {n:4d} postings: {res[0]:7.1f} us wall, {dev[0]:6.1f} us on the device  (general paths: {res[1]:7.1f} / {dev[1]:6.1f} us; "
          f"the wall time includes the Python binding's argument marshalling)", flush=True)
    seg.free(); out.free()
ctx.close()

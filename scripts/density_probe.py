"""2-term AND over 100M docs for several Zipf rank pairs: which tile path they take shows in the rate."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from inverted_index_2_amd import Context, synth
D = 100_000_000
ctx = Context(0)
thresholds = [int(x) for x in sys.argv[1].split(",")] if len(sys.argv) > 1 else [0]
for ra, rb in ((1, 2), (2, 3), (3, 5), (5, 8), (10, 20), (30, 60), (100, 200), (1000, 3), (10000, 2), (100000, 1)):
    a, b = synth.zipf_list(ra, D), synth.zipf_list(rb, D)
    seg = ctx.encode_lists([a, b])
    want = np.intersect1d(a, b, assume_unique=True)
    out = ctx.empty(min(a.size, b.size) + 512)
    dcnt = ctx.empty(8, np.uint64)
    lists = [(seg, 0), (seg, 1)]
    nin = a.size + b.size
    line = f"ranks ({ra},{rb}): sizes {a.size:>9d} {b.size:>9d}"
    for th in thresholds:
        ctx.set_option("intersect.map_docs", th)
        _, n = ctx.intersect(lists, out=out)
        ok = n == want.size and np.array_equal(out.download(n), want)
        for _ in range(3): ctx.intersect_async(lists, None, out, dcnt)
        ctx.sync()
        t = time.time()
        K = 30
        for _ in range(K): ctx.intersect_async(lists, None, out, dcnt)
        ctx.sync()
        dt = (time.time() - t) / K
        line += f" | map_docs={th}: {dt*1e6:7.1f} us ok={ok}"
    print(line, flush=True)
    seg.free()

"""A/B probe: bitmap tile kernel (intersect.bm2) vs the general tile kernel on the C2 workload."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from inverted_index_2_amd import Context, synth
D = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000_000
ctx = Context(0)
a, b = synth.zipf_list(2, D), synth.zipf_list(3, D)
seg = ctx.encode_lists([a, b])
want = np.intersect1d(a, b, assume_unique=True)
out = ctx.empty(min(a.size, b.size) + 512)
dcnt = ctx.empty(8, np.uint64)
lists = [(seg, 0), (seg, 1)]
for bm2, wgs in ((0, 0), (1, 0), (1, 5), (1, 6)):
    ctx.set_option("intersect.bm2", bm2)
    ctx.set_option("intersect.wgs", wgs)
    _, n = ctx.intersect(lists, out=out)
    ok = n == want.size and np.array_equal(out.download(n), want)
    for _ in range(5): ctx.intersect_async(lists, None, out, dcnt)
    ctx.sync()
    t = time.time()
    for _ in range(100): ctx.intersect_async(lists, None, out, dcnt)
    ctx.sync()
    dt = (time.time() - t) / 100
    print(f"bm2={bm2} wgs={wgs} match={ok}: {dt*1e6:.1f} us/step {(a.size+b.size)/dt/1e9:.0f} Gpostings/s", flush=True)

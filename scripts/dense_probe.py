"""A/B probe: wave-streaming dense intersection (intersect.dense) vs the general tile kernels on Zipf rank pairs
over a 100M-doc universe.  Usage: python scripts/dense_probe.py [bpw ...]"""
import sys, time
import numpy as np
sys.path.insert(0, ".")
from inverted_index_2_amd import Context, synth

D = 100_000_000
ctx = Context(0)
bpws = [int(x) for x in sys.argv[1:]] or [16]
for ranks in ((2, 3), (1, 2), (3, 5), (5, 8), (2, 3, 5)):
    lists = [synth.zipf_list(r, D) for r in ranks]
    seg = ctx.encode_lists(lists)
    ls = [(seg, i) for i in range(len(lists))]
    want = lists[0]
    for x in lists[1:]:
        want = np.intersect1d(want, x, assume_unique=True)
    out = ctx.empty(min(l.size for l in lists) + 512)
    cnt = ctx.empty(8, np.uint64)
    for dense, bpw in [(0, 0)] + [(1, b) for b in bpws]:
        ctx.set_option("intersect.dense", dense)
        ctx.set_option("intersect.dense_bpw", bpw)
        _, n = ctx.intersect(ls, out=out)
        ok = n == want.size and np.array_equal(out.download(n), want)
        for _ in range(10):
            ctx.intersect_async(ls, None, out, cnt)
        ctx.sync()
        ctx.profile_region(True)
        t0 = time.perf_counter()
        for _ in range(100):
            ctx.intersect_async(ls, None, out, cnt)
        ctx.profile_region(False)
        ctx.sync()
        dt = (time.perf_counter() - t0) / 100
        dev = ctx.profile_region_ms() / 100
        n_in = sum(l.size for l in lists)
        print(f"ranks={ranks} dense={dense} bpw={bpw} match={ok} n={n}/{want.size}: wall {dt*1e6:.1f} us  device {dev*1e3:.1f} us/step  {n_in/dev/1e6:.0f} G postings/s", flush=True)
    seg.free(); out.free(); cnt.free()

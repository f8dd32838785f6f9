import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from inverted_index_2_amd import Context
ctx = Context(0)
def bernoulli(rng, p, lo, hi):
    return (np.flatnonzero(rng.random(hi - lo) < p) + lo).astype(np.uint32)
rng = np.random.default_rng(77)
U = 2_400_000
a, b = bernoulli(rng, 0.45, 0, U), bernoulli(rng, 0.35, 0, U)
def punch(x, holes):
    keep = np.ones(x.size, bool)
    for lo, hi in holes:
        keep &= ~((x >= lo) & (x < hi))
    return x[keep]
a = punch(a, [(100_000, 100_300), (500_000, 700_000), (1_000_000, 1_020_000)])
b = punch(b, [(90_000, 130_000), (1_500_000, 1_500_900), (2_000_000, 2_000_040)])
thin = (b >= 1_200_000) & (b < 1_400_000) & (rng.random(b.size) > 0.03)
b = b[~thin]
b = np.union1d(b, a[(a >= 1_200_000) & (a < 1_400_000)][::7]).astype(np.uint32)
for lists in ([a, b],):
    want = np.intersect1d(lists[0], lists[1], assume_unique=True).astype(np.uint32)
    seg = ctx.encode_lists(lists)
    ls = [(seg, 0), (seg, 1)]
    out = ctx.empty(min(l.size for l in lists) + 16)
    for and2, dbg in ((1, 0), (2, 0)):
        ctx.set_option("intersect.and2", and2)
        _, n = ctx.intersect(ls, out=out)
        got = out.download(n)
        print("and2", and2, "dbg", dbg, "n", n, want.size, "equal", np.array_equal(got, want))
        if not np.array_equal(got, want):
            m = min(n, want.size)
            bad = np.flatnonzero(got[:m] != want[:m])
            print(" mismatches", bad.size, "first at", bad[:10])
            for i in bad[:6]:
                print("   i", i, "got", got[i], "want", want[i], " around want", want[i-2:i+3], "got", got[i-2:i+3])
            # which B blocks?
            short = lists[0] if lists[0].size < lists[1].size else lists[1]
            for i in bad[:3]:
                pos = np.searchsorted(short, want[i])
                print("   B posting index", pos, "block", pos // 256, "in-block", pos % 256, "wave", pos // 4096)

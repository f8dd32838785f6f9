"""Randomized GPU-vs-numpy runs of the dense streaming intersection and union: 2..4 lists of random densities (with holes, thin
stretches and multi-byte gaps), random tombstones, random driver blocks per wave.  Usage: stress_dense.py [iterations] [seed]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from inverted_index_2_amd import Context


def main(iters=40, seed0=1, ctx=None, quiet=False):
    """`iters` random cases; raises AssertionError on the first mismatch."""
    own = ctx is None
    if own:
        ctx = Context(0)
    for it in range(iters):
        rng = np.random.default_rng(seed0 * 1000 + it)
        U = int(rng.integers(900_000, 3_000_000))
        base = int(rng.choice([0, 0, 5_000_000, (1 << 32) - U - 1]))
        k = int(rng.integers(2, 5))
        lists = []
        for j in range(k):
            p = float(rng.uniform(0.27, 0.95))
            keep = rng.random(U) < p
            for _ in range(int(rng.integers(0, 4))):          # holes and thin stretches
                a = int(rng.integers(0, U)); ln = int(rng.choice([40, 300, 5000, 200_000]))
                if rng.random() < 0.5: keep[a:a + ln] = False
                else: keep[a:a + ln] &= rng.random(min(ln, U - a)) < 0.02
            lists.append((np.flatnonzero(keep) + base).astype(np.uint32))
        want = lists[0]
        for x in lists[1:]:
            want = np.intersect1d(want, x, assume_unique=True)
        removed = None
        if rng.random() < 0.5:
            removed = (np.flatnonzero(rng.random(U) < 0.01) + base).astype(np.uint32)
            want = np.setdiff1d(want, removed, assume_unique=True)
        seg = ctx.encode_lists(lists)
        tomb = ctx.tombstones(removed) if removed is not None else None
        bpw = int(rng.choice([0, 16, 32, 48, 64]))
        ctx.set_option("intersect.dense_bpw", bpw)
        out, n = ctx.intersect([(seg, i) for i in range(k)], tomb=tomb)
        got = out.download(n)
        if n != want.size or not np.array_equal(got, want.astype(np.uint32)):
            raise AssertionError("MISMATCH intersect: " + repr((it, "seed", seed0, "k", k, "U", U, "base", base, "bpw", bpw, n, want.size)))
        out.free()
        # the same lists (some shifted / shortened so they reach past each other) through the streaming union
        ul = [l if rng.random() < 0.5 else l[int(rng.integers(0, l.size // 3)): l.size - int(rng.integers(0, l.size // 3))] for l in lists]
        useg = ctx.encode_lists(ul)
        uw = ul[0]
        for x in ul[1:]:
            uw = np.union1d(uw, x)
        if removed is not None:
            uw = np.setdiff1d(uw, removed, assume_unique=True)
        out, n = ctx.union([(useg, i) for i in range(k)], tomb=tomb)
        if n != uw.size or not np.array_equal(out.download(n), uw.astype(np.uint32)):
            raise AssertionError("MISMATCH union: " + repr((it, "seed", seed0, "k", k, "U", U, "base", base, n, uw.size)))
        seg.free(); useg.free(); out.free()
        if tomb: tomb.free()
        if it % 10 == 9 and not quiet: print("ok", it + 1, flush=True)
    ctx.set_option("intersect.dense_bpw", 0)
    if not quiet:
        print("stress_dense ok", iters)
    if own:
        ctx.close()
    return iters


if __name__ == "__main__":
    main(int(sys.argv[1]) if len(sys.argv) > 1 else 40, int(sys.argv[2]) if len(sys.argv) > 2 else 1)

"""Randomized merge stress (GPU vs oracle): random term counts, segment counts, size skew, clustered ids, duplicates,
tombstones; also merge -> segment -> merge again (ii2_merge_segments_to_seg) and term selection views.
usage: python scripts/stress_merge.py [seconds] [seed]"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from inverted_index_2_amd import Context
from oracle import oracle as orc


def main(budget=120.0, seed=1, ctx=None, quiet=False):
    """Runs for `budget` seconds; raises AssertionError on the first mismatch.  Returns the iteration count."""
    own = ctx is None
    if own:
        ctx = Context(0)
    rng = np.random.default_rng(seed)


    def ids(n, universe, style):
        if n == 0:
            return np.empty(0, np.uint32)
        if style == 0:
            return np.unique(rng.integers(0, universe, n, dtype=np.uint64)).astype(np.uint32)
        if style == 1:
            c = int(rng.integers(0, universe))
            return np.unique((c + rng.integers(0, max(2, n * int(rng.integers(1, 6))), n, dtype=np.uint64)) % universe).astype(np.uint32)
        v = np.concatenate([rng.integers(0, min(universe, 4000), n - n // 8, dtype=np.uint64), rng.integers(0, universe, n // 8 + 1, dtype=np.uint64)])
        return np.unique(v).astype(np.uint32)


    t_end = time.time() + budget
    it = 0
    while time.time() < t_end:
        it += 1
        universe = int(rng.choice([3_000, 100_000, 10_000_000, (1 << 32) - 1]))
        k = int(rng.choice([1, 2, 3, 5, 16, 33, 64]))
        T = int(rng.choice([1, 2, 7, 60, 400, 3000]))
        head = int(rng.choice([0, 1, 3]))                       # how many "large" terms
        offs, vals = [], []
        base_sizes = np.minimum(rng.zipf(1.6, T) * int(rng.choice([1, 5, 40])), 3000)
        for i in range(head):
            base_sizes[int(rng.integers(0, T))] = int(rng.choice([5_000, 40_000, 200_000]))
        styles = rng.integers(0, 3, T)
        shared = [ids(int(min(b, universe)), universe, int(st)) for b, st in zip(base_sizes, styles)]      # overlap across segments
        for s in range(k):
            parts = []
            for t in range(T):
                if rng.random() < 0.3:
                    parts.append(np.empty(0, np.uint32)); continue
                src = shared[t]
                take = src[rng.random(src.size) < rng.choice([0.2, 0.6, 1.0])]
                extra = ids(int(rng.integers(0, 30)), universe, 0)
                parts.append(np.union1d(take, extra).astype(np.uint32))
            offs.append(np.concatenate([[0], np.cumsum([p.size for p in parts])]).astype(np.uint64))
            vals.append(np.concatenate(parts + [np.empty(0, np.uint32)]).astype(np.uint32))
        removed = None
        if rng.random() < 0.6:
            removed = np.unique(rng.integers(0, universe, int(rng.integers(1, 20000)), dtype=np.uint64)).astype(np.uint32)
        tomb = ctx.tombstones(removed) if removed is not None else None
        rm = removed if removed is not None else ()
        w_off, w_vals, w_terms = orc.merge_segments(offs, vals, rm)
        segs = [ctx.encode(o, v) for o, v in zip(offs, vals)]
        out_off, out_vals, st = ctx.merge(segs, tomb=tomb)
        ok = np.array_equal(out_off.download(), w_off) and np.array_equal(out_vals.download(int(w_off[-1])), w_vals) and st.n_terms_out == w_terms
        if not ok:
            raise AssertionError("MISMATCH merge: iteration %d seed %d k %d T %d universe %d n_in %d" % (it, seed, k, T, universe, sum(int(o[-1]) for o in offs)))
        # two-stage merge through a device segment: merge(first half) -> seg, then merge(seg, rest) must equal the one-shot merge
        if k >= 2 and w_off[-1] > 0:
            h = k // 2
            mid, _ = ctx.merge_to_segment(segs[:h], tomb=tomb)
            if mid is not None:
                out_off2, out_vals2, st2 = ctx.merge([mid] + segs[h:], tomb=tomb)
                ok2 = np.array_equal(out_off2.download(), w_off) and np.array_equal(out_vals2.download(int(w_off[-1])), w_vals)
                if not ok2:
                    raise AssertionError("MISMATCH two-stage merge: iteration %d seed %d k %d T %d universe %d" % (it, seed, k, T, universe))
        if it % 20 == 0 and not quiet:
            print("iter", it, "ok", flush=True)
    if not quiet:
        print("merge stress ok:", it, "iterations, seed", seed)
    if own:
        ctx.close()
    return it


if __name__ == "__main__":
    main(float(sys.argv[1]) if len(sys.argv) > 1 else 120.0, int(sys.argv[2]) if len(sys.argv) > 2 else 1)

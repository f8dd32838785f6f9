"""Randomized GPU-vs-oracle stress run (intersect / union / merge over mixed densities); prints the first mismatch.
usage: python scripts/stress.py [seconds] [seed]"""
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


    def rand_list(n, universe, style):
        if n == 0:
            return np.empty(0, np.uint32)
        if style == 0:      # uniform
            return np.unique(rng.integers(0, universe, n, dtype=np.uint64)).astype(np.uint32)
        if style == 1:      # dense run with small gaps (single-byte) and occasional big jumps
            gaps = rng.choice([1, 2, 3, 4, 9, 31, 32, 33, 127, 128, 300, 20000], size=n, p=[.3, .25, .2, .1, .05, .02, .02, .02, .01, .01, .01, .01])
            v = int(rng.integers(0, max(1, universe // 4))) + np.cumsum(gaps)
            return np.unique(v[v < universe]).astype(np.uint32)
        if style == 2:      # very dense (bitmap tiles) with rare wide one-byte gaps
            gaps = rng.choice([1, 2, 3, 110, 127], size=n, p=[.55, .33, .105, .0075, .0075])
            v = int(rng.integers(0, max(1, universe // 8))) + np.cumsum(gaps)
            return np.unique(v[v < universe]).astype(np.uint32)
        # clustered
        centers = rng.integers(0, universe, max(1, n // 500), dtype=np.uint64)
        v = (centers[rng.integers(0, centers.size, n)] + rng.integers(0, 2000, n, dtype=np.uint64)) % universe
        return np.unique(v).astype(np.uint32)


    t_end = time.time() + budget
    it = 0
    while time.time() < t_end:
        it += 1
        universe = int(rng.choice([5_000, 200_000, 5_000_000, 1 << 31, (1 << 32) - 1]))
        k = int(rng.choice([1, 2, 2, 3, 4, 7, 8, 9, 20, 64]))
        sizes = [int(rng.choice([0, 1, 255, 256, 257, 3000, 40_000, 300_000, 1_000_000], p=[.05, .05, .05, .1, .05, .2, .2, .2, .1])) for _ in range(k)]
        lists = [rand_list(min(s, universe), universe, int(rng.integers(0, 4))) for s in sizes]
        removed = None
        if rng.random() < 0.5:
            removed = np.unique(rng.integers(0, universe, int(rng.integers(1, 5000)), dtype=np.uint64)).astype(np.uint32)
        tomb = ctx.tombstones(removed) if removed is not None else None
        seg = ctx.encode_lists(lists)
        ref_rm = removed if removed is not None else ()
        # codec: decode == input, byte-exact with the oracle's encoder, export -> import -> intersect gives the same
        po_w = np.concatenate([[0], np.cumsum([l.size for l in lists])]).astype(np.uint64)
        flat = np.concatenate(lists + [np.empty(0, np.uint32)]).astype(np.uint32)
        po_g, vals_g = seg.decode()
        blk, skip, payload = seg.export()
        o_blk, o_skip, o_payload = orc.dv1_encode(po_w, flat)
        if not (np.array_equal(po_g, po_w) and np.array_equal(vals_g, flat) and np.array_equal(blk, o_blk) and np.array_equal(payload, o_payload)
                and np.array_equal(skip["first_doc"][:-1], o_skip["first_doc"][:-1]) and np.array_equal(skip["byte_off"], o_skip["byte_off"])):
            raise AssertionError("MISMATCH codec: " + repr((it, "seed", seed, "k", k, "universe", universe, "sizes", [l.size for l in lists])))
        if it % 5 == 0:
            seg = ctx.import_dv1(int(flat.size), blk, skip, payload)       # the imported copy serves the rest of the iteration
        # intersect (all option combinations that change the kernel path)
        want = orc.intersect(lists, ref_rm)
        for bm in (1, 0):
            ctx.set_option("intersect.bitmap", bm)
            out, n = ctx.intersect([(seg, i) for i in range(k)], tomb=tomb)
            got = out.download(n)
            if n != want.size or not np.array_equal(got, want):
                print("MISMATCH intersect", it, "seed", seed, "k", k, "universe", universe, "sizes", [l.size for l in lists], "bm", bm, n, want.size)
                os.makedirs("gpurun_out", exist_ok=True)
                np.savez_compressed("gpurun_out/stress_fail.npz", got=got, want=want, removed=removed if removed is not None else np.empty(0, np.uint32), **{"list%d" % i: l for i, l in enumerate(lists)})
                raise AssertionError("MISMATCH intersect (inputs saved to gpurun_out/stress_fail.npz)")
        ctx.set_option("intersect.bitmap", 1)
        # union
        wantu = orc.union(lists)
        if removed is not None:
            wantu = orc.filter_removed(wantu, removed)
        for dense in (1, 0):
            ctx.set_option("union.dense", dense)
            out, n = ctx.union([(seg, i) for i in range(k)], tomb=tomb)
            if n != wantu.size or not np.array_equal(out.download(n), wantu):
                raise AssertionError("MISMATCH union: " + repr((it, "seed", seed, "k", k, "universe", universe, "sizes", [l.size for l in lists], "dense", dense, n, wantu.size)))
        ctx.set_option("union.dense", 1)
        # merge: the k lists as k single-term segments plus a second term built from shuffled halves
        T = 3
        offs, vals = [], []
        for l in lists[: min(k, 16)]:
            parts = [l, l[::2].copy(), rand_list(int(rng.integers(0, 2000)), universe, 0)]
            offs.append(np.concatenate([[0], np.cumsum([p.size for p in parts])]).astype(np.uint64))
            vals.append(np.concatenate(parts).astype(np.uint32))
        w_off, w_vals, w_terms = orc.merge_segments(offs, vals, np.sort(removed) if removed is not None else ())
        segs = [ctx.encode(o, v) for o, v in zip(offs, vals)]
        out_off, out_vals, st = ctx.merge(segs, tomb=tomb)
        if not (np.array_equal(out_off.download(), w_off) and np.array_equal(out_vals.download(int(w_off[-1])), w_vals) and st.n_terms_out == w_terms):
            raise AssertionError("MISMATCH merge: " + repr((it, "seed", seed, "k", len(offs), "universe", universe, "sizes", [l.size for l in lists])))
        if it % 20 == 0 and not quiet:
            print("iter", it, "ok", flush=True)
    if not quiet:
        print("stress ok:", it, "iterations, seed", seed)
    if own:
        ctx.close()
    return it


if __name__ == "__main__":
    main(float(sys.argv[1]) if len(sys.argv) > 1 else 120.0, int(sys.argv[2]) if len(sys.argv) > 2 else 1)

"""Replays one iteration of scripts/stress_merge.py (same RNG stream) and reports where GPU and oracle differ.
usage: python scripts/repro_merge.py <seed> <iteration> [opt=value ...]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from inverted_index_2_amd import Context
from oracle import oracle as orc

seed, target = int(sys.argv[1]), int(sys.argv[2])
ctx = Context(0)
for kv in sys.argv[3:]:
    ctx.set_option(kv.split("=")[0], int(kv.split("=")[1]))
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

for it in range(1, target + 1):
    universe = int(rng.choice([3_000, 100_000, 10_000_000, (1 << 32) - 1]))
    k = int(rng.choice([1, 2, 3, 5, 16, 33, 64]))
    T = int(rng.choice([1, 2, 7, 60, 400, 3000]))
    head = int(rng.choice([0, 1, 3]))
    offs, vals = [], []
    base_sizes = np.minimum(rng.zipf(1.6, T) * int(rng.choice([1, 5, 40])), 3000)
    for i in range(head):
        base_sizes[int(rng.integers(0, T))] = int(rng.choice([5_000, 40_000, 200_000]))
    styles = rng.integers(0, 3, T)
    shared = [ids(int(min(b, universe)), universe, int(st)) for b, st in zip(base_sizes, styles)]
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
    if it < target:
        # the stress loop's later draws do not depend on results, but the two-stage merge draws nothing either: skip the GPU work
        continue
    print("case: k", k, "T", T, "universe", universe, "n_in", sum(int(o[-1]) for o in offs), "removed", None if removed is None else removed.size, flush=True)
    tomb = ctx.tombstones(removed) if removed is not None else None
    rm = removed if removed is not None else ()
    w_off, w_vals, w_terms = orc.merge_segments(offs, vals, rm)
    segs = [ctx.encode(o, v) for o, v in zip(offs, vals)]
    out_off, out_vals, st = ctx.merge(segs, tomb=tomb)
    g_off = out_off.download(); g_vals = out_vals.download(int(g_off[-1]))
    print("n_out gpu", int(g_off[-1]), "oracle", int(w_off[-1]), "tiles", int(st.n_tiles))
    n_t = sum(np.diff(o.astype(np.int64)) for o in offs)
    bad = 0
    for t in range(T):
        a = g_vals[int(g_off[t]):int(g_off[t + 1])]; b = w_vals[int(w_off[t]):int(w_off[t + 1])]
        if a.size != b.size or not np.array_equal(a, b):
            bad += 1
            if bad <= 6:
                lists = [v[int(o[t]):int(o[t + 1])] for o, v in zip(offs, vals)]
                mn = min(int(x[0]) for x in lists if x.size); mx = max(int(x[-1]) for x in lists if x.size)
                only_g = np.setdiff1d(a, b); only_w = np.setdiff1d(b, a)
                print(f"term {t}: n_in {int(n_t[t])} gpu {a.size} oracle {b.size} style {int(styles[t])} min {mn} max {mx} sorted {bool(np.all(a[1:] > a[:-1]))} only_gpu {only_g.size} {only_g[:5]} only_oracle {only_w.size} {only_w[:5]}")
    print("bad terms", bad)

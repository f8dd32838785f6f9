"""Synthetic Zipf postings: splitmix64, per-term seed = hash(global seed,
term rank), doc ids by geometric-gap sampling with p = df/D over [0, D) — sorted-unique by
construction.  Workload generator for tests and bench.py; not part of the data path."""
from __future__ import annotations

import numpy as np

GLOBAL_SEED = 0x1A2B3C4D
_GOLD = np.uint64(0x9E3779B97F4A7C15)
_M1 = np.uint64(0xBF58476D1CE4E5B9)
_M2 = np.uint64(0x94D049BB133111EB)


def splitmix64(x: np.ndarray) -> np.ndarray:
    with np.errstate(over="ignore"):
        z = x.astype(np.uint64) + _GOLD
        z = (z ^ (z >> np.uint64(30))) * _M1
        z = (z ^ (z >> np.uint64(27))) * _M2
        return z ^ (z >> np.uint64(31))


def term_seed(rank: int, global_seed: int = GLOBAL_SEED) -> int:
    with np.errstate(over="ignore"):
        x = np.asarray([rank], np.uint64) * _GOLD ^ np.uint64(global_seed)
    return int(splitmix64(x)[0])


def geometric_postings(p: float, universe: int, seed: int, offset: int = 0) -> np.ndarray:
    """Ascending unique doc ids in [offset, offset+universe), each doc kept with probability p."""
    if p >= 1.0:
        return (np.arange(universe, dtype=np.uint64) + np.uint64(offset)).astype(np.uint32)
    out = []
    start = -1
    ctr = 0
    log1mp = np.log1p(-p)
    while start < universe - 1:
        n = int(min(max((universe - 1 - start) * p * 1.02 + 4096, 4096), 1 << 26))
        with np.errstate(over="ignore"):
            idx = (np.arange(ctr, ctr + n, dtype=np.uint64) * _GOLD) + np.uint64(seed)
        ctr += n
        u = ((splitmix64(idx) >> np.uint64(11)).astype(np.float64) + 1.0) * (1.0 / 9007199254740992.0)
        gaps = np.floor(np.log(u) / log1mp).astype(np.int64) + 1
        ids = start + np.cumsum(gaps)
        keep = ids < universe
        out.append(ids[keep])
        if not keep.all():
            break
        start = int(ids[-1])
    ids = np.concatenate(out) if out else np.empty(0, np.int64)
    return (ids + offset).astype(np.uint32)


def zipf_list(rank: int, universe: int, offset: int = 0, global_seed: int = GLOBAL_SEED) -> np.ndarray:
    """Postings of the term of Zipf rank `rank`: df = floor(D / rank)."""
    df = universe // rank
    return geometric_postings(df / universe, universe, term_seed(rank, global_seed), offset)


def merge_workload(n_terms: int, k: int, mean_len: float, universe: int, dup_frac: float = 0.10,
                   tomb_frac: float = 0.01, seed: int = GLOBAL_SEED):
    """C3-style input: term sizes ∝ 1/rank scaled to `mean_len`, each posting placed in one of k
    segments uniformly, dup_frac of them also in a second segment; tombstones = tomb_frac of the
    universe.  Returns (seg_offs [k][T+1] u64, seg_vals [k] u32, removed u32 sorted)."""
    rng = np.random.default_rng(seed)
    ranks = np.arange(1, n_terms + 1, dtype=np.float64)
    w = 1.0 / ranks
    sizes = np.clip(np.floor(w * (mean_len * n_terms / w.sum())), 1, universe).astype(np.int64)
    term_of = np.repeat(np.arange(n_terms, dtype=np.int64), sizes)
    # doc ids: per term sorted-unique sample; sampling with replacement + unique keeps it cheap
    docs = rng.integers(0, universe, term_of.size, dtype=np.int64)
    key = term_of * universe + docs
    key = np.unique(key)
    term_of, docs = key // universe, key % universe
    seg = rng.integers(0, k, key.size)
    dup = rng.random(key.size) < dup_frac
    seg2 = (seg + 1 + rng.integers(0, max(k - 1, 1), key.size)) % k
    t_all = np.concatenate([term_of, term_of[dup]])
    d_all = np.concatenate([docs, docs[dup]])
    s_all = np.concatenate([seg, seg2[dup]]) if k > 1 else np.concatenate([seg, seg[dup]])
    if k == 1:      # a duplicate inside one list would break sorted-unique: drop the copies
        t_all, d_all, s_all = term_of, docs, seg
    order = np.lexsort((d_all, t_all, s_all))
    t_all, d_all, s_all = t_all[order], d_all[order], s_all[order]
    seg_offs, seg_vals = [], []
    bounds = np.searchsorted(s_all, np.arange(k + 1))
    for s in range(k):
        a, b = bounds[s], bounds[s + 1]
        cnt = np.bincount(t_all[a:b], minlength=n_terms)
        off = np.zeros(n_terms + 1, np.uint64)
        off[1:] = np.cumsum(cnt)
        seg_offs.append(off)
        seg_vals.append(d_all[a:b].astype(np.uint32))
    n_rem = int(universe * tomb_frac)
    removed = np.sort(rng.integers(0, universe, n_rem, dtype=np.int64)).astype(np.uint32)
    return seg_offs, seg_vals, removed

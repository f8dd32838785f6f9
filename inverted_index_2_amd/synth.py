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


def random_terms(n: int, seed: int = GLOBAL_SEED, lo: int = 10, hi: int = 19) -> list:
    """Stand-in for the reference's missing terms.1m.txt (.MISSING_LARGE_BLOBS): n unique [a-zA-Z]{lo..hi}
    byte strings in the manner of randomString (shard_test.go:258-266), seeded."""
    letters = np.frombuffer(b"abcdefghijklmnopqrstuvwxyzABCDEFGHIJKLMNOPQRSTUVWXYZ", np.uint8)
    rng = np.random.default_rng(seed)
    out, seen = [], set()
    while len(out) < n:
        m = n - len(out)
        lens = rng.integers(lo, hi + 1, m)
        chars = letters[rng.integers(0, letters.size, int(lens.sum()))]
        pos = np.concatenate([[0], np.cumsum(lens)])
        blob = chars.tobytes()
        for i in range(m):
            t = blob[pos[i]:pos[i + 1]]
            if t not in seen:
                seen.add(t)
                out.append(t)
    return out


def c1_workload(n_terms: int = 1_000_000, docs_per_segment: int = 10_000, n_segments: int = 2, terms_per_doc: int = 20,
                seed: int = GLOBAL_SEED):
    """BASELINE config 1 (plumbing): docs [s * dps, (s+1) * dps) go to segment s; every doc gets `terms_per_doc`
    terms drawn Zipf(s=1) over the term file (duplicates inside a doc collapse, as a Put's term set does).
    Returns (term_rank [n_pairs] i64, doc [n_pairs] i64) — the distinct (term rank, doc) pairs, sorted by (rank, doc)."""
    rng = np.random.default_rng(seed ^ 0xC1)
    n_docs = docs_per_segment * n_segments
    w = 1.0 / np.arange(1, n_terms + 1, dtype=np.float64)
    cdf = np.cumsum(w)
    cdf /= cdf[-1]
    ranks = np.searchsorted(cdf, rng.random(n_docs * terms_per_doc), side="left").astype(np.int64)
    docs = np.repeat(np.arange(n_docs, dtype=np.int64), terms_per_doc)
    key = np.unique(ranks * n_docs + docs)
    return key // n_docs, key % n_docs


def _merge_chunk(args):
    """One contiguous range of terms of merge_workload_big: per-segment values + per-(segment, term) counts."""
    t0, sizes, k, universe, dup_frac, seed = args
    rng = np.random.default_rng([seed, t0])
    nt = sizes.size
    total = int(sizes.sum())
    term_of = np.repeat(np.arange(nt, dtype=np.int32), sizes)
    p = np.repeat(np.minimum(sizes / float(universe), 1.0), sizes)
    gaps = rng.geometric(p).astype(np.int64)            # >= 1: ids inside a term are strictly ascending
    del p
    cs = np.cumsum(gaps)
    del gaps
    starts = np.concatenate([[0], np.cumsum(sizes)[:-1]])
    base = np.where(starts > 0, cs[np.maximum(starts, 1) - 1], 0)
    ids = cs - np.repeat(base, sizes) - 1
    del cs
    keep = ids < universe
    ids, term_of = ids[keep].astype(np.uint32), term_of[keep]
    total = ids.size
    seg = rng.integers(0, k, total, dtype=np.uint8)
    vals, counts = [], np.zeros((k, nt), np.int64)
    if k > 1:
        dup = rng.random(total, dtype=np.float32) < dup_frac
        seg2 = ((seg.astype(np.int16) + 1 + rng.integers(0, k - 1, total, dtype=np.int16)) % k).astype(np.uint8)
        seg2[~dup] = 255
    for s in range(k):
        m = seg == s
        if k > 1:
            m |= seg2 == s
        vals.append(ids[m])
        counts[s] = np.bincount(term_of[m], minlength=nt)
    return vals, counts


def merge_chunk_bounds(n_terms: int, mean_len: float, universe: int):
    """Term sizes of the big merge workload and the term indices at which its generator cuts chunks (every chunk is
    seeded by its first term, so a rank may generate any range of whole chunks on its own): multiples of
    n_terms / 64, the head chunks cut further so that no chunk is much above 1/256 of the postings."""
    ranks = np.arange(1, n_terms + 1, dtype=np.float64)
    w = 1.0 / ranks
    sizes_all = np.clip(np.floor(w * (mean_len * n_terms / w.sum())), 1, universe).astype(np.int64)
    step = max(n_terms // 64, 1)
    bounds = list(range(0, n_terms, step)) + [n_terms]
    target = max(int(sizes_all.sum()) // 256, 1 << 20)
    fine = [0]
    for a, b in zip(bounds[:-1], bounds[1:]):
        cum = np.cumsum(sizes_all[a:b])
        cuts = np.searchsorted(cum, np.arange(target, int(cum[-1]), target)) + a + 1
        fine.extend(int(c) for c in np.unique(cuts) if a < c < b)
        fine.append(b)
    return sizes_all, sorted(set(fine))


def merge_workload_big(n_terms: int, k: int, mean_len: float, universe: int, dup_frac: float = 0.10, tomb_frac: float = 0.01,
                       seed: int = GLOBAL_SEED, threads: int = 8, term_range=None):
    """BASELINE config 3 / 4 at full size: same shape as merge_workload (term sizes ∝ 1/rank scaled to `mean_len`, each
    posting in one of k segments, dup_frac also in a second one, tombstones = tomb_frac of the universe) but generated
    without a global sort — per term geometric gaps (sorted-unique by construction), chunks of terms on a thread pool,
    every chunk seeded by its first term, so the data does not depend on the thread count.  term_range=(t0, t1)
    generates only those terms' lists (a rank's share of config 4) — identical to the same slice of the full workload
    as long as t0 / t1 fall on chunk boundaries of the full run (multiples of n_terms / 64 do).
    Returns (seg_offs [k][T+1] u64, seg_vals [k] u32, removed u32 sorted)."""
    from concurrent.futures import ThreadPoolExecutor
    sizes_all, fine = merge_chunk_bounds(n_terms, mean_len, universe)
    t_lo, t_hi = term_range if term_range is not None else (0, n_terms)
    jobs = [(a, sizes_all[a:b], k, universe, dup_frac, seed) for a, b in zip(fine[:-1], fine[1:]) if a >= t_lo and b <= t_hi]
    assert jobs and jobs[0][0] == t_lo and jobs[-1][0] + jobs[-1][1].size == t_hi, "term_range must fall on chunk boundaries"
    with ThreadPoolExecutor(max_workers=threads) as pool:
        parts = list(pool.map(_merge_chunk, jobs))
    seg_offs, seg_vals = [], []
    for s in range(k):
        cnt = np.concatenate([p[1][s] for p in parts])
        off = np.zeros(cnt.size + 1, np.uint64)
        off[1:] = np.cumsum(cnt)
        seg_offs.append(off)
        seg_vals.append(np.concatenate([p[0][s] for p in parts]))
        for p in parts:
            p[0][s] = None
    rng = np.random.default_rng([seed, 0x70B])
    removed = np.sort(rng.integers(0, universe, int(universe * tomb_frac), dtype=np.int64)).astype(np.uint32)
    return seg_offs, seg_vals, removed

"""How the hot path is spread over the GPUs of a node (one process per GPU).

Merge: shards are independent in the reference (own segments, own tombstones —
shard.go:19-26, fan-out inverted_index.go:83-103) and shardKey is order-preserving for terms
of >= 2 bytes (shard.go:371-377), so the 1024 shard keys are cut into `world` contiguous
ranges.  One conjunctive query: the doc-id space is cut into `world` contiguous ranges.  Either
way rank-order concatenation of the per-rank results is the global result
(inverted_index.go:330-339) — that concatenation is the only exchange (ii2_allgatherv)."""
from __future__ import annotations

from typing import List, Sequence, Tuple

import numpy as np

N_SHARD_KEYS = 1024          # 10 bits of the first two term bytes (shard.go:375)


def shard_key(term: bytes) -> int:
    """shard.go:362-378."""
    if len(term) < 2:
        return 0
    return ((term[0] << 8) + term[1]) >> 6


def key_range(rank: int, world: int) -> Tuple[int, int]:
    """Contiguous shard-key range [lo, hi) of a rank."""
    return rank * N_SHARD_KEYS // world, (rank + 1) * N_SHARD_KEYS // world


def owner_of_key(key: int, world: int) -> int:
    for r in range(world):
        lo, hi = key_range(r, world)
        if lo <= key < hi:
            return r
    raise ValueError(key)


def term_range(rank: int, world: int, n_terms: int) -> Tuple[int, int]:
    """Contiguous range of aligned term slots of a rank (synthetic workloads: term id order = key order)."""
    return rank * n_terms // world, (rank + 1) * n_terms // world


def doc_range(rank: int, world: int, universe: int) -> Tuple[int, int]:
    return rank * universe // world, (rank + 1) * universe // world


def slice_list_to_docs(ids: np.ndarray, lo: int, hi: int) -> np.ndarray:
    """The part of an ascending id list that falls in [lo, hi)."""
    a, b = np.searchsorted(ids, [lo, hi])
    return ids[a:b]


def concat_in_rank_order(parts: Sequence[np.ndarray]) -> np.ndarray:
    return np.concatenate(list(parts)) if len(parts) else np.empty(0, np.uint32)


# Merge cost per posting by term size, relative to a giant term (measured on one MI355X: every rank's share of the 8-way split of
# C4 - 64 segments - timed through BENCH_PRETEND=r/8 python bench.py --workload strong, ~0.6 ms of per-step fixed cost taken off;
# round 4's kernels): the densest terms go through bitmap tiles (0.7 - 1.0), terms of a few hundred thousand postings through range
# tiles at about twice that, and terms of ~800 - 8,000 postings at more than three times (peak near 2,000: the boundary between
# batches and range tiles): with 64 segments such a term is 12 - 120 postings per list, one mostly empty block each, and every
# (range tile, list) pair needs its cut; batches of small
# terms pay per list on top (block set-up, one 16-byte piece per tiny list).  Sizes are for the 100M-doc universe of the
# configs.  (Round 3's table had everything between a few thousand and a million postings at ~2.0: the two mid ranks of the
# 8-way split then took 3.65 ms against 2.6 - 2.9 for the others.)
_COST_LOG10_SIZE = np.array([1.9, 2.1, 2.4, 2.7, 3.0, 3.3, 3.6, 3.9, 4.2, 4.5, 4.9, 5.4, 5.9, 6.45, 7.1, 7.76])
_COST_PER_POSTING = np.array([4.2, 3.8, 3.1, 2.85, 3.4, 3.9, 3.5, 3.0, 2.3, 2.1, 2.05, 2.0, 1.7, 1.2, 0.75, 1.0])


# DV1-encoding the merged postings (merge -> segment, what Shard.Merge does) costs the same per posting whatever the term:
# ~0.3 - 0.5 ms per 100M postings against 0.82 ms per 100M for merging a giant term; 0.35 balanced the 8-way split of C4 best
# (BENCH_PRETEND=r/8 python bench.py --workload strong: 3.6 - 4.1 ms per rank with 0.67, 3.8 - 4.1 with 0).
_ENCODE_COST_PER_POSTING = 0.35


def merge_cost_weights(sizes: np.ndarray, encode: bool = False) -> np.ndarray:
    """Estimated merge cost of every term (arbitrary unit: postings of a giant term); encode: the result is written as a
    DV1 segment, not as raw ids."""
    sizes = np.asarray(sizes, dtype=np.float64)
    per = np.interp(np.log10(np.maximum(sizes, 1.0)), _COST_LOG10_SIZE, _COST_PER_POSTING)
    return sizes * (per + (_ENCODE_COST_PER_POSTING if encode else 0.0))


def balanced_term_ranges(n_terms: int, mean_len: float, universe: int, world: int, by: str = "cost") -> List[Tuple[int, int]]:
    """Contiguous term ranges of the big synthetic merge workload, one per rank, balanced by estimated merge COST
    (by="cost": posting count weighted by merge_cost_weights — a posting of a 160-posting term costs 2.7x one of a giant
    term, so ranges balanced by posting count alone leave the tail rank 2.5x the head rank's time; by="cost+encode": the
    merged postings are DV1-encoded too, a flat cost per posting on top) or by POSTING count
    (by="postings"; Zipf skew: equal term counts would give rank 0 most of the work — SURVEY §8 e), cut at the
    generator's chunk boundaries so that every rank can generate exactly its own share."""
    from . import synth
    sizes, fine = synth.merge_chunk_bounds(n_terms, mean_len, universe)
    if by not in ("cost", "cost+encode", "postings"):
        raise ValueError("by must be 'cost', 'cost+encode' or 'postings'")
    w = merge_cost_weights(sizes, encode=by == "cost+encode") if by != "postings" else sizes.astype(np.float64)
    cum = np.concatenate([[0.0], np.cumsum(w)])
    total = float(cum[-1])
    fine = np.asarray(fine)
    cuts = [0]
    for r in range(1, world):
        want = total * r / world
        j = int(np.argmin(np.abs(cum[fine] - want)))
        c = int(fine[j])
        c = max(c, cuts[-1])
        cuts.append(c)
    cuts.append(n_terms)
    for r in range(1, world + 1):                 # strictly increasing where possible (tiny inputs may leave empty ranks)
        if cuts[r] < cuts[r - 1]:
            cuts[r] = cuts[r - 1]
    return [(cuts[r], cuts[r + 1]) for r in range(world)]

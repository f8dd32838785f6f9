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


# Merge cost per posting by term size, relative to a giant term (measured range by range on one MI355X, round 2,
# scripts/strong_ranges.py, 64 segments): terms above the tile size run through single-term tiles at the same rate
# whatever their size, except the densest (a tile's doc range fits the LDS bitmap: ~0.75x); batches of small terms pay
# per list (block decode, slice tables, scans over T x k counts).  Sizes are for the 100M-doc universe of the configs.
_COST_LOG10_SIZE = np.array([2.2, 3.0, 3.8, 4.6, 6.0, 6.6])
_COST_PER_POSTING = np.array([2.7, 2.0, 1.18, 1.0, 1.0, 0.75])


def merge_cost_weights(sizes: np.ndarray) -> np.ndarray:
    """Estimated merge cost of every term (arbitrary unit: postings of a giant term)."""
    sizes = np.asarray(sizes, dtype=np.float64)
    return sizes * np.interp(np.log10(np.maximum(sizes, 1.0)), _COST_LOG10_SIZE, _COST_PER_POSTING)


def balanced_term_ranges(n_terms: int, mean_len: float, universe: int, world: int, by: str = "cost") -> List[Tuple[int, int]]:
    """Contiguous term ranges of the big synthetic merge workload, one per rank, balanced by estimated merge COST
    (by="cost": posting count weighted by merge_cost_weights — a posting of a 160-posting term costs 2.7x one of a giant
    term, so ranges balanced by posting count alone leave the tail rank 2.5x the head rank's time) or by POSTING count
    (by="postings"; Zipf skew: equal term counts would give rank 0 most of the work — SURVEY §8 e), cut at the
    generator's chunk boundaries so that every rank can generate exactly its own share."""
    from . import synth
    sizes, fine = synth.merge_chunk_bounds(n_terms, mean_len, universe)
    if by not in ("cost", "postings"):
        raise ValueError("by must be 'cost' or 'postings'")
    w = merge_cost_weights(sizes) if by == "cost" else sizes.astype(np.float64)
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

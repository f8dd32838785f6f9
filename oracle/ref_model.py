"""In-memory restatement of the reference's Shard / InvertedIndex behaviour (small cases).

TEST INFRASTRUCTURE ONLY (see oracle/ii2_oracle.h).  Pure-Python loops; the posting
arithmetic goes through the C oracle (oracle.py).  Files, FSTs and locks are replaced
by dicts — only what decides *which doc ids come out* is kept:

  Segments.add ordering            segments.go:56-64
  Shard.Put (direct segment)       shard.go:33-67
  Shard.Read / makeIterator        shard.go:72-75, 253-278  (k-way merge, MergeTermValues fold)
  Shard.Remove + RemovedLists      shard.go:78-105, removed_list.go:36-71
  Shard.Merge                      shard.go:127-245
  shardKey                         shard.go:362-378
  InvertedIndex.Put/Read/Merge/PutRemoved/PrefixSearch   inverted_index.go:41-340
"""
from __future__ import annotations

from typing import Dict, List, Optional, Tuple

import numpy as np

from . import oracle as orc


class _Clock:
    """Stands in for time.Now().UnixNano(): strictly increasing."""

    def __init__(self) -> None:
        self.t = 1_000_000

    def now(self) -> int:
        self.t += 1000
        return self.t


class Segment:
    def __init__(self, key: int, terms: Dict[bytes, List[int]]):
        self.key = key                      # unix-ns key (file/writer.go names files by it)
        self.postings = terms               # term -> values (verbatim, as written)
        self.terms = len(terms)
        self.merging = False


class RemovedLists:
    def __init__(self) -> None:
        self.lists: Dict[int, List[int]] = {}

    def put(self, ts: int, values) -> None:           # removed_list.go:36-41
        self.lists[ts] = list(values)

    def values(self) -> np.ndarray:                   # removed_list.go:44-54
        return orc.removed_values(list(self.lists.values()))

    def sync(self, timestamps) -> None:               # removed_list.go:57-71
        if not timestamps:
            return
        ts = list(self.lists.keys())
        keep = orc.removed_sync(ts, list(timestamps))
        for t, k in zip(ts, keep):
            if not k:
                del self.lists[t]


def _kway(segments: List[Segment], lo: Optional[bytes], hi: Optional[bytes]) -> List[Tuple[bytes, List[int]]]:
    """makeIterator (shard.go:253-278): merge by term; equal terms folded with MergeTermValues
    in source order; a term held by one source only is yielded verbatim."""
    terms = set()
    for s in segments:
        terms.update(s.postings.keys())
    out = []
    for term in sorted(terms):
        if lo is not None and orc.compare_terms(term, lo) < 0:
            continue
        if hi is not None and orc.compare_terms(term, hi) > 0:   # inclusive max, file/reader.go:54-58
            continue
        acc = None
        for s in segments:
            if term not in s.postings:
                continue
            vals = s.postings[term]
            acc = list(vals) if acc is None else orc.merge_term_values(acc, vals).tolist()
        out.append((term, acc))
    return out


class Shard:
    def __init__(self, clock: Optional[_Clock] = None):
        self.clock = clock or _Clock()
        self.segments: List[Segment] = []
        self.removed = RemovedLists()

    # segments.go:56-64 — insert before the first segment with terms >= new.terms
    def _add(self, seg: Segment) -> None:
        pos = 0
        while pos < len(self.segments) and self.segments[pos].terms < seg.terms:
            pos += 1
        self.segments.insert(pos, seg)

    def put(self, terms: List[bytes], val: int) -> None:        # shard.go:33-67
        terms = sorted(terms)
        self._add(Segment(self.clock.now(), {t: [val] for t in terms}))

    def read(self, lo: Optional[bytes] = None, hi: Optional[bytes] = None):   # shard.go:72-75 — no tombstone filter
        return _kway(list(self.segments), lo, hi)

    def remove(self, values) -> None:                           # shard.go:78-105
        if len(values) == 0:
            return
        timestamps = [self.clock.now()] + [s.key for s in self.segments]
        self.removed.sync(timestamps)
        self.removed.put(self.clock.now(), values)

    def merge(self, req_count: int, m_count: int) -> int:       # shard.go:127-245
        if len(self.segments) < req_count:
            return 0
        picked = []
        for s in self.segments:
            if len(picked) == m_count:
                break
            if not s.merging:
                s.merging = True
                picked.append(s)
        if len(picked) < 2:
            return 0                                            # NB: a lone picked flag stays set (shard.go:149-151)
        removed = self.removed.values()
        merged: Dict[bytes, List[int]] = {}
        for term, vals in _kway(picked, None, None):
            kept = orc.filter_removed(vals, removed).tolist()   # shard.go:181-190
            if not kept:
                continue                                        # shard.go:192-194
            merged[term] = kept
        if merged:                                              # shard.go:219-225 — lazy writer
            self._add(Segment(self.clock.now(), merged))
        self.segments = [s for s in self.segments if s not in picked]
        return len(picked)

    def min_max(self):                                          # shard.go:280-298
        lo = hi = None
        for s in self.segments:
            ks = sorted(s.postings.keys())
            if not ks:
                continue
            lo = ks[0] if lo is None or ks[0] < lo else lo
            hi = ks[-1] if hi is None or ks[-1] > hi else hi
        return lo, hi


class InvertedIndex:
    def __init__(self) -> None:
        self.clock = _Clock()
        self.shards: Dict[int, Shard] = {}

    def put(self, terms: List[bytes], val: int) -> None:        # inverted_index.go:113-145
        groups: Dict[int, List[bytes]] = {}
        for t in terms:
            groups.setdefault(orc.shard_key(t), []).append(t)
        for key in sorted(groups):
            self.shards.setdefault(key, Shard(self.clock)).put(groups[key], val)

    def put_removed(self, values) -> None:                      # inverted_index.go:41-55
        for s in self.shards.values():
            s.remove(values)

    def merge(self, req_count: int, m_count: int, concurrency: int = 1) -> int:   # inverted_index.go:62-109
        return sum(self.shards[k].merge(req_count, m_count) for k in sorted(self.shards))

    def read(self, lo: Optional[bytes] = None, hi: Optional[bytes] = None):       # inverted_index.go:300-340
        out = []
        for key in sorted(self.shards):                         # shards in ascending key order
            s = self.shards[key]
            smin, smax = s.min_max()
            if smin is None:
                continue
            if lo is not None and orc.compare_terms(lo, smax) > 0:
                continue
            if hi is not None and orc.compare_terms(hi, smin) < 0:
                continue
            out.extend(s.read(lo, hi))
        return out

    def prefix_search(self, prefixes: List[bytes]) -> Dict[bytes, List[int]]:     # inverted_index.go:192-295
        prefixes = sorted(prefixes)
        found: Dict[bytes, List[np.ndarray]] = {}
        for key in sorted(self.shards):
            s = self.shards[key]
            smin, smax = s.min_max()
            if smin is None:
                continue
            mine = []
            for p in prefixes:
                l = min(len(p), len(smin))
                if p[:l] < smin[:l]:
                    continue
                l = min(len(p), len(smax))
                if p[:l] > smax[:l]:
                    continue
                mine.append(p)
            if not mine:
                continue
            greatest = mine[-1]
            for term, vals in s.read(mine[0], None):
                if greatest < term[: min(len(term), len(greatest))]:
                    break
                for p in mine:
                    if term.startswith(p):
                        found.setdefault(p, []).append(np.asarray(vals, np.uint32))
        return {p: orc.union(ls).tolist() for p, ls in found.items()}             # :288-292

"""CPU: BASELINE config 1 (plumbing) — a seeded stand-in for the reference's missing terms.1m.txt, 2 segments x 10k
docs, 20 Zipf terms per doc, 2-term AND of the two most frequent terms — through the CPU oracle (the reference's Go
path cannot be built here: no Go toolchain).  The GPU leg is tests/test_gpu_configs.py::test_config1_*."""
import numpy as np

from inverted_index_2_amd import synth
from oracle import oracle as orc
from oracle import ref_model


def c1_segments(n_terms, dps, n_seg=2):
    """Term-aligned CSR segments over the terms that occur, plus the byte strings of those terms in bytes.Compare order."""
    rank, doc = synth.c1_workload(n_terms, dps, n_seg)
    used = np.unique(rank)
    names = synth.random_terms(n_terms)
    order = sorted(range(used.size), key=lambda i: names[used[i]])          # slot order = term byte order (file/types.go:24-26)
    slot_of_rank = np.full(n_terms, -1, np.int64)
    slot_of_rank[used[order]] = np.arange(used.size)
    slot = slot_of_rank[rank]
    offs, vals = [], []
    for s in range(n_seg):
        m = (doc >= s * dps) & (doc < (s + 1) * dps)
        o = np.lexsort((doc[m], slot[m]))
        cnt = np.bincount(slot[m], minlength=used.size)
        off = np.zeros(used.size + 1, np.uint64)
        off[1:] = np.cumsum(cnt)
        offs.append(off)
        vals.append(doc[m][o].astype(np.uint32))
    terms = [names[used[i]] for i in order]
    return terms, offs, vals, (slot_of_rank[0], slot_of_rank[1]), (rank, doc)


def test_config1_two_term_and_full_size():
    terms, offs, vals, (s1, s2), (rank, doc) = c1_segments(1_000_000, 10_000)
    assert all(orc.compare_terms(a, b) < 0 for a, b in zip(terms[:2000], terms[1:2001]))
    m_off, m_vals, n_terms_out = orc.merge_segments(offs, vals, ())
    assert n_terms_out == len(terms)
    a = m_vals[int(m_off[s1]):int(m_off[s1 + 1])]
    b = m_vals[int(m_off[s2]):int(m_off[s2 + 1])]
    got = orc.intersect([a, b])
    want = np.intersect1d(doc[rank == 0], doc[rank == 1]).astype(np.uint32)      # independent: straight from the doc table
    assert np.array_equal(got, want) and 5000 < got.size < 12000
    # every merged list = the docs of that term, ascending (two segments with disjoint doc ranges concatenate)
    assert int(m_off[-1]) == rank.size and np.array_equal(np.sort(m_vals[int(m_off[s1]):int(m_off[s1 + 1])]), a)


def test_config1_reduced_through_the_reference_model():
    # same generator, 2 segments x 300 docs over a 5000-term file, through the Shard/InvertedIndex restatement:
    # Put per doc, merge the first half into one segment per shard, Put the second half, merge those, then AND
    n_terms, dps = 5000, 300
    rank, doc = synth.c1_workload(n_terms, dps, 2)
    names = synth.random_terms(n_terms)
    ii = ref_model.InvertedIndex()
    for half in range(2):
        for d in range(half * dps, (half + 1) * dps):
            ii.put([names[r] for r in rank[doc == d]], d)
        if half == 0:
            while ii.merge(2, 1_000_000):
                pass
        else:
            for sh in ii.shards.values():                       # merge only the direct segments of the second half
                n_direct = sum(1 for s in sh.segments if s.terms and all(len(v) == 1 for v in s.postings.values()) and min(min(v) for v in s.postings.values()) >= dps)
                if n_direct >= 2:
                    sh.merge(2, n_direct)
    lists = dict(ii.read())
    a, b = lists[names[0]], lists[names[1]]
    want = np.intersect1d(doc[rank == 0], doc[rank == 1]).tolist()
    assert orc.intersect([np.asarray(a, np.uint32), np.asarray(b, np.uint32)]).tolist() == want
    assert max(len(sh.segments) for sh in ii.shards.values()) <= 2

"""GPU: the BASELINE.json configurations that round 1 left untested at their stated sizes, against the oracle:
config 1 (terms.1m.txt stand-in, 2 segments x 10k docs, 2-term AND), config 3 (16-way merge of 1M terms, ~1.06e9
postings, 1 % tombstones — offsets and values bit-equal to the oracle) and config 4's one-GPU shape (64 segments x
125k terms).  Configs 2 and 5 are in test_gpu_fullsize.py."""
import gc
import os

import numpy as np
import pytest

from inverted_index_2_amd import synth
from oracle import oracle as orc
from oracle import ref_model
from tests.gpu_util import ctx  # noqa: F401
from tests.test_config1_cpu import c1_segments

pytestmark = pytest.mark.gpu

CORES = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)


def test_config1_two_term_and_full_size(ctx):
    # configs[0] through the GPU path: the two segments' aligned CSR -> merged segment (Read's view) -> AND of the two
    # most frequent terms; the oracle does the same on the CPU (tests/test_config1_cpu.py)
    terms, offs, vals, (s1, s2), (rank, doc) = c1_segments(1_000_000, 10_000)
    segs = [ctx.encode(o, v) for o, v in zip(offs, vals)]
    merged, st = ctx.merge_to_segment(segs)
    w_off, w_vals, w_terms = orc.merge_segments(offs, vals, ())
    po, v = merged.decode()
    assert np.array_equal(po, w_off) and np.array_equal(v, w_vals) and st.n_terms_out == w_terms == len(terms)
    out, n = ctx.intersect([(merged, int(s1)), (merged, int(s2))])
    want = orc.intersect([w_vals[int(w_off[s]):int(w_off[s + 1])] for s in (s1, s2)])
    assert n == want.size and np.array_equal(out.download(n), want)
    assert np.array_equal(want, np.intersect1d(doc[rank == 0], doc[rank == 1]).astype(np.uint32))


def test_config1_reduced_through_the_host_mirror(ctx):
    # the same workload at 2 x 150 docs through Put / Merge / Read / Intersect of the host mirror (every Put is a
    # direct GPU segment), next to the reference model
    from inverted_index_2_amd.host import InvertedIndex
    n_terms, dps = 2000, 150
    rank, doc = synth.c1_workload(n_terms, dps, 2)
    names = synth.random_terms(n_terms)
    gpu, ref = InvertedIndex(ctx), ref_model.InvertedIndex()
    for half in range(2):
        for d in range(half * dps, (half + 1) * dps):
            ts = [names[r] for r in rank[doc == d]]
            gpu.put(list(ts), d)
            ref.put(list(ts), d)
        while True:
            a, b = gpu.merge(2, 1_000_000, 4), ref.merge(2, 1_000_000, 4)
            assert a == b
            if a == 0:
                break
    assert gpu.read() == [(t, [int(x) for x in vs]) for t, vs in ref.read()]
    assert gpu.intersect([names[0], names[1]]) == np.intersect1d(doc[rank == 0], doc[rank == 1]).tolist()
    gpu.close()


def _check_merge(ctx, offs, vals, removed, threads, to_segment=False):
    segs = [ctx.encode(o, v) for o, v in zip(offs, vals)]
    tomb = ctx.tombstones(removed)
    out_off, out_vals, st = ctx.merge(segs, tomb)
    g_off = out_off.download()
    g_vals = out_vals.download(int(st.n_out))
    out_off.free()
    out_vals.free()
    exported = None
    if to_segment:      # Shard.Merge's real output (shard.go:207 -> file/writer.go:32-59): the merged terms as an encoded segment
        merged, st2 = ctx.merge_to_segment(segs, tomb)
        assert int(st2.n_out) == int(st.n_out)
        exported = merged.export()
        merged.free()
    for s in segs:
        s.free()
    w_off, w_vals, w_terms = orc.merge_segments(offs, vals, removed, threads=threads)
    assert int(st.n_out) == int(w_off[-1]) and st.n_terms_out == w_terms
    assert np.array_equal(g_off, w_off)
    assert np.array_equal(g_vals, w_vals)
    if exported is not None:      # byte for byte the DV1 encoding of the oracle's merge
        oblk, oskip, opayload = orc.dv1_encode(w_off, w_vals)
        blk, skip, payload = exported
        assert np.array_equal(blk, oblk) and np.array_equal(skip["first_doc"], oskip["first_doc"]) and np.array_equal(skip["byte_off"], oskip["byte_off"])
        assert np.array_equal(payload, opayload)
    return st


def test_config3_full_size_16way_merge_of_1m_terms(ctx):
    # configs[2] at the size the metric is quoted on: 16 segments x 1,000,000 terms x mean 1000 postings (~1.06e9 in),
    # 10 % duplicated into a second segment, 1 % tombstones; offsets + values bit-equal to the oracle's worker pool
    offs, vals, removed = synth.merge_workload_big(1_000_000, 16, 1000.0, 100_000_000, threads=min(CORES, 16))
    n_in = sum(int(v.size) for v in vals)
    assert n_in > 1_000_000_000
    st = _check_merge(ctx, offs, vals, removed, threads=min(CORES, 32), to_segment=True)
    assert st.n_in == n_in and 0.85 * n_in < st.n_out < n_in
    del offs, vals
    gc.collect()


def test_config4_one_gpu_shape_64way_merge(ctx):
    # configs[3]'s per-GPU shape: 64 segments x 125,000 terms (Zipf, mean 1000, ~1.3e8 postings), tombstones on — the
    # first time the k = 64 tile kernel sees large-term tiles, 33-term batches and 64-run folds at real sizes
    offs, vals, removed = synth.merge_workload_big(125_000, 64, 1000.0, 100_000_000, threads=min(CORES, 16))
    st = _check_merge(ctx, offs, vals, removed, threads=min(CORES, 32), to_segment=True)
    assert st.n_tiles > 10_000
    # and a light share of the 8-way term split of the 1M-term index (terms 375k..500k: all batches of tiny lists)
    offs, vals, removed = synth.merge_workload_big(1_000_000, 64, 1000.0, 100_000_000, threads=min(CORES, 16),
                                                   term_range=(375_000, 500_000))
    _check_merge(ctx, offs, vals, removed, threads=min(CORES, 32))

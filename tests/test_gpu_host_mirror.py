"""GPU: the reference's own test scripts (tests/golden/ref_kats.json — shard_test.go,
inverted_index_test.go) replayed against the host mirror (csrc/host_index.cpp), whose posting
work runs on the GPU through the C ABI; the oracle's model runs the same scripts beside it."""
import numpy as np
import pytest

from oracle import ref_model
from tests.gpu_util import ctx  # noqa: F401
from tests.kat_runner import run_script

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("name", [
    "TestInitFromExistingFiles", "TestIngestion", "TestReadPartial_merged", "TestReadPartial_direct",
    "TestMerging", "TestMergeWithRemoval", "TestMergeEmptySegment", "TestConcurrentAccess_script",
])
def test_shard_scripts(ctx, kats, name):
    from inverted_index_2_amd.host import Shard
    s = Shard(ctx)
    run_script(s, kats["scripts"][name]["script"], n_segments=lambda t: t.n_segments, removed_values=lambda t: t.removed_values())
    s.close()


@pytest.mark.parametrize("name", ["TestPutRemove", "TestPut", "TestSearchByPrefix", "TestReadScoped"])
def test_index_scripts(ctx, kats, name):
    from inverted_index_2_amd.host import InvertedIndex
    ii = InvertedIndex(ctx)
    run_script(ii, kats["scripts"][name]["script"], n_shards=lambda t: t.n_shards)
    ii.close()


def test_random_workload_matches_reference_model(ctx):
    # a TestConcurrent-like workload (inverted_index_test.go:84-138), sequential: random puts, removals, merges
    from inverted_index_2_amd.host import InvertedIndex
    rng = np.random.default_rng(42)
    vocab = [bytes(rng.choice(list(b"abcdefghijklmnopqrstuvwxyzABCDEFGHIJKLMNOPQRSTUVWXYZ"), int(rng.integers(2, 8))).tolist())
             for _ in range(120)]
    gpu, ref = InvertedIndex(ctx), ref_model.InvertedIndex()
    for step in range(150):
        terms = [vocab[i] for i in rng.choice(len(vocab), int(rng.integers(1, 6)), replace=False)]
        val = int(rng.integers(0, 60))
        gpu.put(list(terms), val)
        ref.put(list(terms), val)
        if step % 37 == 36:
            rem = rng.integers(0, 60, 5).tolist()
            gpu.put_removed(rem)
            ref.put_removed(rem)
        if step % 25 == 24:
            assert gpu.merge(2, 4, 2) == ref.merge(2, 4, 2)
    while True:
        a, b = gpu.merge(2, 100, 2), ref.merge(2, 100, 2)
        assert a == b
        if a == 0:
            break
    assert gpu.read() == [(t, [int(v) for v in vs]) for t, vs in ref.read()]
    assert gpu.read(b"b", b"q") == [(t, [int(v) for v in vs]) for t, vs in ref.read(b"b", b"q")]
    assert gpu.prefix_search([b"a", b"Zq", b"m"]) == ref.prefix_search([b"a", b"Zq", b"m"])
    t1, t2 = vocab[3], vocab[7]
    want = sorted(set(dict(ref.read()).get(t1, [])) & set(dict(ref.read()).get(t2, [])))
    assert gpu.intersect([t1, t2]) == want
    gpu.close()

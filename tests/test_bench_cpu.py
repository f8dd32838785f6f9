"""CPU: the bench harness's own helpers (no GPU): the oracle-based CPU baseline must return the exact intersection for
any shard count, since bench.py cross-checks the GPU result against it."""
import importlib.util
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench():
    spec = importlib.util.spec_from_file_location("bench", os.path.join(ROOT, "bench.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_cpu_baseline_intersect_sharded_equals_numpy():
    b = _bench()
    rng = np.random.default_rng(3)
    lists = [np.unique(rng.integers(0, 400_000, n)).astype(np.uint32) for n in (120_000, 90_000)]
    want = np.intersect1d(lists[0], lists[1], assume_unique=True)
    removed = want[::5].copy()
    for threads in (1, 3, 8):
        rate, res, per = b.cpu_baseline_intersect(lists, None, 1, threads=threads)
        assert np.array_equal(res, want) and rate > 0 and per > 0
        _, res2, _ = b.cpu_baseline_intersect(lists, removed, 1, threads=threads)
        assert np.array_equal(res2, np.setdiff1d(want, removed, assume_unique=True))


def test_bench_argument_contract():
    b = _bench()
    import sys
    argv = sys.argv
    try:
        sys.argv = ["bench.py", "--gpus", "1", "--steps", "7", "--warmup", "2"]
        a = b.parse()
    finally:
        sys.argv = argv
    assert (a.gpus, a.steps, a.warmup, a.workload) == (1, 7, 2, "all")

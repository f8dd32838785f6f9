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


def test_gpus_n_starts_n_ranks():
    """`bench.py --gpus 2` without a launcher starts two ranks itself (torch.distributed.run on 127.0.0.1) before touching a
    GPU; --dry-run keeps the ranks off the GPU: they join a gloo group, count themselves and rank 0 prints the frame."""
    import json
    import subprocess
    import sys
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--dry-run"],
                         capture_output=True, text=True, env=env, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1                                   # rank 0 alone prints
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == 2 and rec["ranks_counted"] == 2 and rec["steps"] == 3 and rec["dry_run"] is True


def test_c5_index_is_the_same_for_any_rank_count():
    """configs[4]'s lists are defined on 8 fixed doc shards: 1, 2, 4 and 8 ranks hold the same index."""
    b = _bench()
    D = 4_000_000
    whole, _ = b.c5_lists(D, 1, 0)
    for world in (2, 8):
        parts = [b.c5_lists(D, world, r) for r in range(world)]
        for t in range(len(b.C5_RANKS)):
            cat = np.concatenate([p[0][t] for p in parts])
            assert np.array_equal(cat, whole[t])
        assert [p[1] for p in parts] == [(r * D // world, (r + 1) * D // world) for r in range(world)]
    core = np.unique(np.random.default_rng(55).integers(0, D, 10_000)).astype(np.uint32)
    assert all(np.isin(core, l).all() for l in whole)

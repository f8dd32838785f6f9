"""GPU probe: what a tiny Shard.Merge costs through the host mirror (segments of a few terms, like the direct segments
Shard.Put writes): the floor set by launch and sync latency, not by the amount of data."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from inverted_index_2_amd import Context
from inverted_index_2_amd.host import Shard
ctx = Context(0)
rng = np.random.default_rng(2)
vocab = [b"term%04d" % i for i in range(400)]
for segs, terms in ((2, 3), (8, 3), (32, 5), (8, 200)):
    times = []
    for rep in range(12):
        s = Shard(ctx)
        for d in range(segs):
            s.put([vocab[i] for i in rng.choice(len(vocab), terms, replace=False)], d)
        t = time.perf_counter(); n = s.merge(2, segs); times.append(time.perf_counter() - t)
        assert n == segs and s.n_segments == 1
        s.close()
    t = time.perf_counter(); r = None
    s = Shard(ctx)
    for d in range(segs):
        s.put([vocab[i] for i in rng.choice(len(vocab), terms, replace=False)], d)
    t = time.perf_counter(); r = s.read(); tr = time.perf_counter() - t
    s.close()
    print(f"merge of {segs:3d} segments x {terms:3d} terms: {np.median(times)*1e6:8.0f} us  (Read of the same: {tr*1e6:8.0f} us)", flush=True)
ctx.close()

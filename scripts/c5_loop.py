"""BASELINE configs[4] on one GPU (8-term AND over a 1B-doc index, the bench's lists) in a loop, for rocprofv3.
Usage: python scripts/c5_loop.py [steps=N] [docs=D] [opt=value ...]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bench
from inverted_index_2_amd import Context
steps, D = 20, 1_000_000_000
ctx = Context(0)
for kv in sys.argv[1:]:
    k, v = kv.split("=")
    if k == "steps": steps = int(v)
    elif k == "docs": D = int(v)
    else: ctx.set_option(k, int(v))
lists, _ = bench.c5_lists(D, 1, 0)
seg = ctx.encode_lists(lists)
sel = [(seg, i) for i in range(len(lists))]
out = ctx.empty(int(min(l.size for l in lists)) + 512); cnt = ctx.empty(8, np.uint64)
_, n = ctx.intersect(sel, out=out)
info = seg.info
alg = int(info.n_bytes + 8 * info.n_blocks + 4 * n)
for _ in range(5): ctx.intersect_async(sel, None, out, cnt)
ctx.sync()
ctx.profile_region(True)
for _ in range(steps): ctx.intersect_async(sel, None, out, cnt)
ctx.profile_region(False)
ctx.sync()
us = ctx.profile_region_ms() * 1e3 / steps
print("postings_in", int(sum(l.size for l in lists)), "out", n, "device us/query", round(us, 2), "algorithmic_bytes", alg, flush=True)

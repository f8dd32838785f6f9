"""GPU probe (timing experiments): the C3 merge with phases of the tile kernel left out (option debug.merge_skip)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from inverted_index_2_amd import Context, synth
T, k = 1_000_000, 16
ctx = Context(0)
offs, vals, removed = synth.merge_workload_big(T, k, 1000.0, 100_000_000, threads=min(len(os.sched_getaffinity(0)), 32))
segs = [ctx.encode(o, v) for o, v in zip(offs, vals)]
tomb = ctx.tombstones(removed)
n_in = int(sum(int(o[-1]) for o in offs))
del offs, vals
out_off = ctx.empty(T + 1, np.uint64); out_vals = ctx.empty(n_in)
ctx.merge(segs, tomb, out_off, out_vals); ctx.sync()
ctx.set_option("profile.events", 1)
for mask in [int(a) for a in sys.argv[1:]] or [0, 1, 2, 4, 8]:
    ctx.set_option("debug.merge_skip", mask)
    ctx.merge(segs, tomb, out_off, out_vals); ctx.profile_read()
    for _ in range(3): ctx.merge(segs, tomb, out_off, out_vals)
    ms, n = ctx.profile_read()
    print(f"skip mask {mask}: device ms/merge {ms / n:.3f}", flush=True)

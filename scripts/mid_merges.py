"""Merges between the one-launch small path (<= 8192 postings) and the big ones: wall time per ii2_merge_segments /
ii2_merge_segments_to_seg call by input size - what the general path's fixed cost (launches, host waits) is."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from inverted_index_2_amd import Context, synth
ctx = Context(0)
only = int(sys.argv[1]) if len(sys.argv) > 1 else -1      # (one case, for a kernel trace)
for ci, (T, k, mean) in enumerate(((200, 4, 25), (2000, 4, 25), (20000, 4, 25), (200000, 4, 25), (2000, 16, 60), (20000, 16, 60))):
    if only >= 0 and ci != only:
        continue
    offs, vals, removed = synth.merge_workload(T, k, mean, 5_000_000)
    segs = [ctx.encode(o, v) for o, v in zip(offs, vals)]
    tomb = ctx.tombstones(removed)
    n_in = int(sum(int(o[-1]) for o in offs))
    out_off = ctx.empty(T + 1, np.uint64); out_vals = ctx.empty(n_in + 16)
    for _ in range(5): ctx.merge(segs, tomb, out_off, out_vals)
    t0 = time.perf_counter()
    for _ in range(50): ctx.merge(segs, tomb, out_off, out_vals)
    raw = (time.perf_counter() - t0) / 50 * 1e6
    for _ in range(5):
        s, _ = ctx.merge_to_segment(segs, tomb); s.free()
    t0 = time.perf_counter()
    for _ in range(50):
        s, _ = ctx.merge_to_segment(segs, tomb); s.free()
    seg = (time.perf_counter() - t0) / 50 * 1e6
    print(f"terms {T:7d} segs {k:2d} postings_in {n_in:9d}  raw merge {raw:8.1f} us  to segment {seg:8.1f} us", flush=True)

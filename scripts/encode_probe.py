"""GPU probe: the encode step (ii2_seg_encode from device arrays) and merge-to-segment on the C3 merge's output."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from inverted_index_2_amd import Context, synth, _lib
T, k = (int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000), 16
ctx = Context(0)
offs, vals, removed = synth.merge_workload_big(T, k, 1000.0, 100_000_000, threads=min(len(os.sched_getaffinity(0)), 32))
segs = [ctx.encode(o, v) for o, v in zip(offs, vals)]
tomb = ctx.tombstones(removed)
n_in = int(sum(int(o[-1]) for o in offs))
del offs, vals
out_off = ctx.empty(T + 1, np.uint64); out_vals = ctx.empty(n_in)
_, _, st = ctx.merge(segs, tomb, out_off, out_vals); ctx.sync()
for rep in range(3):
    t = time.perf_counter(); s2 = ctx.encode(out_off, out_vals, where=_lib.II2_DEVICE); ctx.sync(); dt = time.perf_counter() - t
    inf = s2.info
    print(f"encode of {int(st.n_out)} postings / {inf.n_lists} lists / {inf.n_blocks} blocks / {inf.n_bytes} bytes: {dt*1e3:.2f} ms", flush=True)
    t = time.perf_counter(); s2.free(); ctx.sync(); print(f"  free {1e3*(time.perf_counter()-t):.2f} ms")
for rep in range(3):
    t = time.perf_counter(); s3, st3 = ctx.merge_to_segment(segs, tomb); ctx.sync(); dt = time.perf_counter() - t
    print(f"merge_to_segment: {dt*1e3:.2f} ms", flush=True)
    s3.free()

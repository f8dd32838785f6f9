"""The bench's merge workload in a loop, for rocprofv3.  Usage: python scripts/merge_loop.py [terms=N] [segments=K] [steps=N] [tombstones=0|1] [opt=value ...]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from inverted_index_2_amd import Context, synth
T, k, steps, use_tomb = 1_000_000, 16, 5, True
ctx = Context(0)
for kv in sys.argv[1:]:
    key, v = kv.split("=")
    if key == "terms": T = int(v)
    elif key == "segments": k = int(v)
    elif key == "steps": steps = int(v)
    elif key == "tombstones": use_tomb = bool(int(v))
    else: ctx.set_option(key, int(v))
offs, vals, removed = synth.merge_workload_big(T, k, 1000.0, 100_000_000, threads=min(len(os.sched_getaffinity(0)), 32))
segs = [ctx.encode(o, v) for o, v in zip(offs, vals)]
tomb = ctx.tombstones(removed) if use_tomb else None
n_in = int(sum(int(o[-1]) for o in offs))
del offs, vals
out_off = ctx.empty(T + 1, np.uint64); out_vals = ctx.empty(n_in)
_, _, st = ctx.merge(segs, tomb, out_off, out_vals)        # warm-up: the workspace grows to its size here, outside the timed calls
ctx.sync()
ctx.set_option("profile.events", 1); ctx.profile_read()
for _ in range(steps):
    _, _, st = ctx.merge(segs, tomb, out_off, out_vals)
ms, n = ctx.profile_read()
enc = sum(s.info.n_bytes for s in segs); nblk = sum(s.info.n_blocks for s in segs)
alg = enc + 8 * nblk + 4 * k * (T + 1) + 100_000_000 // 8 + 4 * int(st.n_out) + 4 * (T + 1)
print("postings_in", n_in, "out", int(st.n_out), "device ms/merge", ms / n, "algorithmic_bytes", alg, flush=True)
for s in segs: s.free()
out_off.free(); out_vals.free()
if tomb is not None: tomb.free()
ctx.close()

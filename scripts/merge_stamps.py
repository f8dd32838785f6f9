"""GPU probe: the bench's merge workload (C3 shape) timed, plus the cycle shares of the tile kernel's steps by tile kind.
Usage: python scripts/merge_stamps.py [terms=N] [segments=K] [opt=value ...]"""
import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from inverted_index_2_amd import Context, synth
T, k = 1_000_000, 16
ctx = Context(0)
for kv in sys.argv[1:]:
    key, v = kv.split("=")
    if key == "terms": T = int(v)
    elif key == "segments": k = int(v)
    else: ctx.set_option(key, int(v))
offs, vals, removed = synth.merge_workload_big(T, k, 1000.0, 100_000_000, threads=min(len(os.sched_getaffinity(0)), 32))
segs = [ctx.encode(o, v) for o, v in zip(offs, vals)]
tomb = ctx.tombstones(removed)
n_in = int(sum(int(o[-1]) for o in offs))
del offs, vals
out_off = ctx.empty(T + 1, np.uint64); out_vals = ctx.empty(n_in)
_, _, st = ctx.merge(segs, tomb, out_off, out_vals)
ctx.sync()
ctx.set_option("profile.events", 1); ctx.profile_read()
for _ in range(3): ctx.merge(segs, tomb, out_off, out_vals)
ms, n = ctx.profile_read()
print(f"postings_in {n_in} out {int(st.n_out)} tiles {int(st.n_tiles)} device ms/merge {ms / n:.3f}", flush=True)
ctx.set_option("profile.events", 0)
ctx.set_option("debug.stamps", 1)
ctx.merge(segs, tomb, out_off, out_vals)
nwg = 512
buf = (C.c_uint64 * (nwg * 8))()
ctx._ck(ctx.lib.ii2_debug_read(ctx.h, buf, nwg * 8))
arr = np.frombuffer(buf, dtype=np.uint64).reshape(-1, 8).astype(np.float64)
names = ["setup", "batch decode", "batch sort+write", "range decode", "range sort+write", "bitmap decode+mark", "bitmap tomb+extract", "tail"]
print("scanner (direct placement): iterations %d, empty %d, cycles %d, tiles %d" % tuple(int(x) for x in arr[0, :4]))
arr = arr[1:]
tot = arr.sum(axis=1)
print(f"cycles per WG: mean {tot.mean():.0f} max {tot.max():.0f} min {tot.min():.0f}")
for i, nm in enumerate(names):
    print(f"  {nm:22s} {100 * arr[:, i].mean() / tot.mean():5.1f}%")

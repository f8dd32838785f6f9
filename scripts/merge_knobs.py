"""GPU probe: the C3 merge under option settings given as name=value[,name=value...] groups (one timing per group)."""
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
ref = None
ctx.set_option("profile.events", 1)
for grp in sys.argv[1:] or ["merge.bitmap_tiles=1"]:
    for kv in grp.split(","):
        key, v = kv.split("=")
        ctx.set_option(key, int(v))
    _, _, st = ctx.merge(segs, tomb, out_off, out_vals); ctx.profile_read()
    for _ in range(3): ctx.merge(segs, tomb, out_off, out_vals)
    ms, n = ctx.profile_read()
    chk = int(out_vals.download(int(st.n_out)).astype(np.uint64).sum()) ^ int(out_off.download().sum())
    if ref is None: ref = chk
    print(f"{grp}: device ms/merge {ms / n:.3f} tiles {int(st.n_tiles)} n_out {int(st.n_out)} same_result {chk == ref}", flush=True)

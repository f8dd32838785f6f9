"""A mid-size merge of many tiny lists (2000 terms x 4 segments x ~7 postings): where its tile kernel's 180 us go."""
import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from inverted_index_2_amd import Context, synth
ctx = Context(0)
T, k, mean = 2000, 4, 25
for kv in sys.argv[1:]:
    key, v = kv.split("=")
    if key == "terms": T = int(v)
    elif key == "segments": k = int(v)
    elif key == "mean": mean = int(v)
    else: ctx.set_option(key, int(v))
offs, vals, removed = synth.merge_workload(T, k, mean, 5_000_000)
segs = [ctx.encode(o, v) for o, v in zip(offs, vals)]
tomb = ctx.tombstones(removed)
n_in = int(sum(int(o[-1]) for o in offs))
out_off = ctx.empty(T + 1, np.uint64); out_vals = ctx.empty(n_in + 16)
for _ in range(3): _, _, st = ctx.merge(segs, tomb, out_off, out_vals)
ctx.set_option("debug.stamps", 1)
ctx.merge(segs, tomb, out_off, out_vals)
nwg = 256
buf = (C.c_uint64 * (nwg * 8))()
ctx._ck(ctx.lib.ii2_debug_read(ctx.h, buf, nwg * 8))
arr = np.frombuffer(buf, dtype=np.uint64).reshape(-1, 8).astype(np.float64)
print(f"postings_in {n_in} tiles {int(st.n_tiles)} lists {T * k}")
print("scanner: iterations %d, empty %d, cycles %d, tiles %d" % tuple(int(x) for x in arr[0, :4]))
names = ["setup", "batch decode", "batch sort+write", "range decode", "range sort+write", "bitmap decode+mark", "bitmap tomb+extract", "tail"]
order = np.argsort(-arr[1:].sum(axis=1))[:8] + 1
for w in order:
    tot = arr[w].sum()
    if tot: print("wg %2d cycles %8d: " % (w, tot) + ", ".join(f"{nm} {int(arr[w, i])}" for i, nm in enumerate(names) if arr[w, i]))

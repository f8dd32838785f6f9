"""Cycle shares of the dense intersection kernels on C2 (option debug.stamps: 1 = tile kernel, 2 = expand kernel)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ctypes as C
import numpy as np
from inverted_index_2_amd import Context, synth

D = 100_000_000
ctx = Context(0)
for kv in sys.argv[1:]:
    k, v = kv.split("=")
    ctx.set_option(k, int(v))
a, b = synth.zipf_list(2, D), synth.zipf_list(3, D)
seg = ctx.encode_lists([a, b])
out = ctx.empty(min(a.size, b.size) + 512)
dcnt = ctx.empty(8, np.uint64)
lists = [(seg, 0), (seg, 1)]
names = {1: ["prologue", "gen + fetch issue", "wait payload", "clear / fold", "mark", "finalise"],
         2: ["prologue", "scan", "stage", "flush"]}
for mode in (1, 2):
    ctx.set_option("debug.stamps", mode)
    ctx.intersect_async(lists, None, out, dcnt); ctx.sync()
    buf = (C.c_uint64 * (2048 * 8))()
    ctx._ck(ctx.lib.ii2_debug_read(ctx.h, buf, 2048 * 8))
    arr = np.frombuffer(buf, dtype=np.uint64).reshape(-1, 8).astype(np.float64)
    arr = arr[arr.sum(axis=1) > 0]
    tot = arr.sum(axis=1).mean() / 4          # 4 waves add into one row
    print("mode", mode, "rows", arr.shape[0], "mean cycles per wave", tot)
    for i, nm in enumerate(names[mode]):
        print(f"    {nm:22s} {arr[:, i].mean() / 4:10.0f}  {100 * arr[:, i].mean() / 4 / tot:5.1f}%")
ctx.set_option("debug.stamps", 0)

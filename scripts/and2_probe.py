"""C2 (2-term AND over 100M docs) through the two dense paths: intersect.and2 = 1 (intersect_and2.hip) against 0 (the n-list
streaming kernel), device time per pass from an event pair around a run of back-to-back passes, result checked against numpy
once.  Extra arguments `name=value` are set as options first; `stamps` prints the per-phase cycle shares of the and2 kernels."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ctypes as C

import numpy as np

from inverted_index_2_amd import Context, synth

D = int(os.environ.get("D", 100_000_000))
ctx = Context(0)
want_stamps = False
for kv in sys.argv[1:]:
    if kv == "stamps":
        want_stamps = True
        continue
    k, v = kv.split("=")
    ctx.set_option(k, int(v))
a, b = synth.zipf_list(2, D), synth.zipf_list(3, D)
want = np.intersect1d(a, b, assume_unique=True)
seg = ctx.encode_lists([a, b])
out = ctx.empty(min(a.size, b.size) + 512)
dcnt = ctx.empty(8, np.uint64)
lists = [(seg, 0), (seg, 1)]
removed = synth.geometric_postings(0.01, D, synth.term_seed(10**6))
tomb = ctx.tombstones(removed)
want_t = np.setdiff1d(want, removed, assume_unique=True)


def timed(tb, steps=50, warm=10):
    for _ in range(warm):
        ctx.intersect_async(lists, tb, out, dcnt)
    ctx.sync()
    ctx.profile_region(True)
    for _ in range(steps):
        ctx.intersect_async(lists, tb, out, dcnt)
    ctx.profile_region(False)
    ctx.sync()
    return ctx.profile_region_ms() * 1e3 / steps


for and2 in (1, 2, 0, 1):
    ctx.set_option("intersect.and2", and2)
    for tb, w, name in ((None, want, "plain"), (tomb, want_t, "tombstones")):
        ctx.intersect_async(lists, tb, out, dcnt)
        ctx.sync()
        n = int(dcnt.download(1)[0])
        ok = n == w.size and np.array_equal(out.download(n), w)
        us = timed(tb)
        print(f"and2={and2} {name:10s} ok={ok} n={n} {us:8.2f} us per pass", flush=True)

if want_stamps:
    ctx.set_option("intersect.and2", 1)
    names = {1: ["prologue", "clear+fetch", "mark A", "tomb+test B", "count+publish", "stage", "look-back", "flush"]}
    for mode in (1,):
        ctx.set_option("debug.stamps", mode)
        ctx.intersect_async(lists, None, out, dcnt)
        ctx.sync()
        buf = (C.c_uint64 * (2048 * 8))()
        ctx._ck(ctx.lib.ii2_debug_read(ctx.h, buf, 2048 * 8))
        arr = np.frombuffer(buf, dtype=np.uint64).reshape(-1, 8).astype(np.float64)
        arr = arr[arr.sum(axis=1) > 0]
        tot = arr.sum(axis=1).mean() / 4
        print("mode", mode, "rows", arr.shape[0], "mean cycles per wave", tot)
        for i, nm in enumerate(names[mode]):
            print(f"    {nm:22s} {arr[:, i].mean() / 4:10.0f}  {100 * arr[:, i].mean() / 4 / tot:5.1f}%")
    ctx.set_option("debug.stamps", 0)

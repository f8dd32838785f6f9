"""A/B probe of the intersect tile-kernel options on the C2 workload: us/step and the per-part cycle shares."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ctypes as C
import numpy as np
from inverted_index_2_amd import Context, synth

D = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000_000
ctx = Context(0)
a, b = synth.zipf_list(2, D), synth.zipf_list(3, D)
seg = ctx.encode_lists([a, b])
want = np.intersect1d(a, b, assume_unique=True)
out = ctx.empty(min(a.size, b.size) + 512)
dcnt = ctx.empty(8, np.uint64)
lists = [(seg, 0), (seg, 1)]
names = ["barrier+prefetch issue", "decode", "bar-after-decode", "finalise", "clear+commit(wait)"]
for bm in (0, 1):
    ctx.set_option("intersect.bitmap", bm)
    _, n = ctx.intersect(lists, out=out)
    ok = n == want.size and np.array_equal(out.download(n), want)
    for _ in range(5):
        ctx.intersect_async(lists, None, out, dcnt)
    ctx.sync()
    t = time.time()
    K = 50
    for _ in range(K):
        ctx.intersect_async(lists, None, out, dcnt)
    ctx.sync()
    dt = (time.time() - t) / K
    print(f"bitmap={bm} match={ok} {dt*1e6:.1f} us/step {(a.size+b.size)/dt/1e9:.1f} Gpostings/s", flush=True)
    ctx.set_option("debug.stamps", 1)
    ctx.intersect_async(lists, None, out, dcnt); ctx.sync()
    nwg = 1280
    buf = (C.c_uint64 * (nwg * 8))()
    ctx._ck(ctx.lib.ii2_debug_read(ctx.h, buf, nwg * 8))
    arr = np.frombuffer(buf, dtype=np.uint64).reshape(-1, 8).astype(np.float64)
    tot = arr.sum(axis=1).mean()
    print("  mean cycles per WG", tot)
    for i, nm in enumerate(names):
        print(f"    {nm:26s} {arr[:, i].mean():10.0f}  {100 * arr[:, i].mean() / tot:5.1f}%")
    ctx.set_option("debug.stamps", 0)
ctx.set_option("intersect.bitmap", 1)
for wgs in (3, 4, 5, 6):
    ctx.set_option("intersect.wgs", wgs)
    for _ in range(3):
        ctx.intersect_async(lists, None, out, dcnt)
    ctx.sync()
    t = time.time()
    for _ in range(50):
        ctx.intersect_async(lists, None, out, dcnt)
    ctx.sync()
    print(f"bitmap=1 wgs/CU={wgs}: {(time.time()-t)/50*1e6:.1f} us/step", flush=True)
ctx.set_option("intersect.wgs", 0)

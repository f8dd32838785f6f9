"""Quick GPU probe: self-test, C2 intersect timing at full size (HIP events via torch-free path)."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from inverted_index_2_amd import Context, synth
from oracle import oracle as orc

D = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000_000
ctx = Context(0)
ctx.selftest(); print("selftest ok", flush=True)
t = time.time(); a, b = synth.zipf_list(2, D), synth.zipf_list(3, D); print("gen", a.size, b.size, time.time() - t, flush=True)
t = time.time(); seg = ctx.encode_lists([a, b]); print("encode", time.time() - t, seg.info.n_bytes, seg.info.n_blocks, flush=True)
t = time.time(); want = orc.intersect([a, b]); cpu_s = time.time() - t; print("oracle", want.size, cpu_s, flush=True)
out = ctx.empty(min(a.size, b.size) + 512)
dcnt = ctx.empty(8, np.uint64)
for lb in (1,):
    pass
    o, n = ctx.intersect([(seg, 0), (seg, 1)], out=out)
    got = out.download(n)
    print("check", lb, "count", n, "match", n == want.size and np.array_equal(got, want), flush=True)
    for g in (0, 2, 4, 8):
        ctx.set_option("intersect.g", g)
        ctx.intersect_async([(seg, 0), (seg, 1)], None, out, dcnt); ctx.sync()
        t = time.time()
        K = 20
        for _ in range(K):
            ctx.intersect_async([(seg, 0), (seg, 1)], None, out, dcnt)
        ctx.sync()
        dt = (time.time() - t) / K
        alg = seg.info.n_bytes + 8 * seg.info.n_blocks + 4 * n
        print(f"  g={g} {dt*1e6:.1f} us/step  {(a.size+b.size)/dt/1e9:.1f} Gpostings/s  alg {alg/dt/1e12:.3f} TB/s", flush=True)
    ctx.set_option("intersect.g", 0)
for wgs in (3, 4, 5, 6, 8, 10):
    ctx.set_option("intersect.wgs", wgs)
    ctx.intersect_async([(seg, 0), (seg, 1)], None, out, dcnt); ctx.sync()
    t = time.time()
    for _ in range(20): ctx.intersect_async([(seg, 0), (seg, 1)], None, out, dcnt)
    ctx.sync(); print(f"wgs/CU={wgs}: {(time.time()-t)/20*1e6:.1f} us/step", flush=True)
ctx.set_option("intersect.wgs", 0)
# per-phase cycle shares (diagnostic path)
import ctypes as C
ctx.set_option("debug.stamps", 1)
ctx.intersect_async([(seg, 0), (seg, 1)], None, out, dcnt); ctx.sync()
nwg = 1280
buf = (C.c_uint64 * (nwg * 8))()
ctx._ck(ctx.lib.ii2_debug_read(ctx.h, buf, nwg * 8))
arr = np.frombuffer(buf, dtype=np.uint64).reshape(-1, 8).astype(np.float64)
names = ["barrier+prefetch issue", "decode", "bar-after-decode", "finalise", "clear+commit(wait)", "-", "-", "-"]
tot = arr.sum(axis=1).mean()
print("mean cycles per WG", tot)
for i, nm in enumerate(names[:5]):
    print(f"  {nm:26s} {arr[:, i].mean():10.0f}  {100 * arr[:, i].mean() / tot:5.1f}%")
ctx.set_option("debug.stamps", 0)

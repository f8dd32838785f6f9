"""C2 (ranks 2 and 3 over 100M docs) in a loop, for rocprofv3.  Usage: python scripts/c2_loop.py [opt=value ...] [steps=N]"""
import sys, time
import numpy as np
sys.path.insert(0, ".")
from inverted_index_2_amd import Context, synth
D = 100_000_000
ctx = Context(0)
steps = 50
for kv in sys.argv[1:]:
    k, v = kv.split("=")
    if k == "steps":
        steps = int(v)
    else:
        ctx.set_option(k, int(v))
a, b = synth.zipf_list(2, D), synth.zipf_list(3, D)
seg = ctx.encode_lists([a, b])
ls = [(seg, 0), (seg, 1)]
out = ctx.empty(b.size + 512)
cnt = ctx.empty(8, np.uint64)
_, n = ctx.intersect(ls, out=out)
import os
if not os.environ.get("C2_NOCHECK"): assert np.array_equal(out.download(n), np.intersect1d(a, b, assume_unique=True))
ctx.sync()
ctx.profile_region(True)
for _ in range(steps):
    ctx.intersect_async(ls, None, out, cnt)
ctx.profile_region(False)
ctx.sync()
print("device us/step", ctx.profile_region_ms() / steps * 1e3)
seg.free(); out.free(); cnt.free(); ctx.close()

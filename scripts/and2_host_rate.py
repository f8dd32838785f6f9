"""How fast can the host enqueue C2 queries (ii2_intersect_async through the Python binding), against the device time per query."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from inverted_index_2_amd import Context, synth
D = 100_000_000
ctx = Context(0)
a, b = synth.zipf_list(2, D), synth.zipf_list(3, D)
seg = ctx.encode_lists([a, b])
out = ctx.empty(b.size + 512); dcnt = ctx.empty(8, np.uint64)
lists = [(seg, 0), (seg, 1)]
for _ in range(20): ctx.intersect_async(lists, None, out, dcnt)
ctx.sync()
for steps in (50, 200):
    ctx.profile_region(True)
    t0 = time.perf_counter()
    for _ in range(steps): ctx.intersect_async(lists, None, out, dcnt)
    t1 = time.perf_counter()
    ctx.profile_region(False)
    ctx.sync()
    t2 = time.perf_counter()
    print(f"steps {steps}: host enqueue {1e6*(t1-t0)/steps:.1f} us/call, wall {1e6*(t2-t0)/steps:.1f} us/call, device region {ctx.profile_region_ms()*1e3/steps:.2f} us/call")

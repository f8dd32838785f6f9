"""Probe: two passes in flight on two contexts (two HIP streams) vs one."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from inverted_index_2_amd import Context, synth
D = 100_000_000
a, b = synth.zipf_list(2, D), synth.zipf_list(3, D)
ctxs = [Context(0) for _ in range(3)]
state = []
for c in ctxs:
    seg = c.encode_lists([a, b])
    state.append((c, seg, c.empty(min(a.size, b.size) + 512), c.empty(8, np.uint64)))
def run(nctx, K=60):
    for c, seg, out, cnt in state[:nctx]:
        c.intersect_async([(seg, 0), (seg, 1)], None, out, cnt)
    for c, *_ in state[:nctx]: c.sync()
    t = time.time()
    for i in range(K):
        c, seg, out, cnt = state[i % nctx]
        c.intersect_async([(seg, 0), (seg, 1)], None, out, cnt)
    for c, *_ in state[:nctx]: c.sync()
    return (time.time() - t) / K
for n in (1, 2, 3, 1, 2):
    dt = run(n)
    print(f"{n} stream(s): {dt*1e6:.1f} us/step  {(a.size+b.size)/dt/1e9:.1f} Gpostings/s", flush=True)

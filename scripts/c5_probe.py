"""Probe: 8-term conjunctive query with skewed list lengths (BASELINE config 5 layout) on one GPU."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from inverted_index_2_amd import Context, synth
from oracle import oracle as orc
D = int(sys.argv[1]) if len(sys.argv) > 1 else 200_000_000
ranks = (2, 4, 16, 64, 256, 1024, 4096, 16384)
rng = np.random.default_rng(55)
core = np.unique(rng.integers(0, D, 10_000)).astype(np.uint32)
t = time.time(); lists = [np.union1d(synth.zipf_list(r, D), core).astype(np.uint32) for r in ranks]; print("gen", time.time() - t, [l.size for l in lists], flush=True)
ctx = Context(0)
seg = ctx.encode_lists(lists)
n_in = sum(l.size for l in lists)
t = time.time(); want = orc.intersect(lists[::-1]); print("oracle (shortest first)", time.time() - t, want.size, flush=True)
out = ctx.empty(lists[1].size + 512); cnt = ctx.empty(8, np.uint64)
for sel in (list(range(8)), [0, 7], [0, 1], [5, 6, 7], [0, 3, 7]):
    ls = [(seg, i) for i in sel]
    o, n = ctx.intersect(ls, out=out)
    ok = np.array_equal(out.download(n), orc.intersect([lists[i] for i in sel][::-1]))
    ctx.intersect_async(ls, None, out, cnt); ctx.sync()
    t = time.time()
    for _ in range(10): ctx.intersect_async(ls, None, out, cnt)
    ctx.sync(); dt = (time.time() - t) / 10
    nin = sum(lists[i].size for i in sel)
    print(f"lists {sel}: {dt*1e6:9.1f} us  postings {nin:>11d}  {nin/dt/1e9:8.1f} Gpostings/s  out {n} ok={ok}", flush=True)

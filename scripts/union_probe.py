"""Union of the two C2 lists (100M docs): OR tiles (union.dense=1) vs the merge passes."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from inverted_index_2_amd import Context, synth
D = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000_000
ctx = Context(0)
a, b = synth.zipf_list(2, D), synth.zipf_list(3, D)
seg = ctx.encode_lists([a, b])
want = np.union1d(a, b)
out = ctx.empty(a.size + b.size + 512)
for dense in (1, 0):
    ctx.set_option("union.dense", dense)
    _, n = ctx.union([(seg, 0), (seg, 1)], out=out)
    ok = n == want.size and np.array_equal(out.download(n), want)
    t = time.time()
    K = 10
    for _ in range(K):
        ctx.union([(seg, 0), (seg, 1)], out=out)
    dt = (time.time() - t) / K
    print(f"union.dense={dense} match={ok}: {dt*1e6:.0f} us  {(a.size+b.size)/dt/1e9:.1f} Gpostings/s  out {n}", flush=True)

"""GPU probe: union of the two C2 lists (Zipf ranks 2 and 3 over 100M docs): streaming kernel vs the fixed-range OR tiles."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from inverted_index_2_amd import Context, synth
D = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000_000
ctx = Context(0)
lists = [synth.zipf_list(r, D) for r in (2, 3)]
seg = ctx.encode_lists(lists)
ls = [(seg, 0), (seg, 1)]
out = ctx.empty(sum(l.size for l in lists) + 16)
want = np.union1d(lists[0], lists[1]).astype(np.uint32)
for stream in (1, 0):
    ctx.set_option("union.stream", stream)
    _, n = ctx.union(ls, out=out)
    assert n == want.size and np.array_equal(out.download(n), want), stream
    ctx.sync(); t = time.time()
    for _ in range(20): ctx.union(ls, out=out)
    ctx.sync(); dt = (time.time() - t) / 20
    print(f"union.stream={stream}: {dt*1e6:.1f} us  {sum(l.size for l in lists)/dt/1e12:.3f} T postings/s  ids out {n}", flush=True)
seg.free(); ctx.close()

import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from inverted_index_2_amd import Context
ctx = Context(0)
ctx.set_option("debug.stamps",1)
seg = ctx.encode_lists([[1,5,9],[2,5,7,9]])
o, c = ctx.union([(seg,0),(seg,1)]); print("union", c, o.download(c))
offs = [np.array([0, 1, 1, 2], np.uint64), np.array([0, 0, 1, 1], np.uint64), np.array([0, 0, 0, 1], np.uint64)]
vals = [np.array([1, 1], np.uint32), np.array([2], np.uint32), np.array([3], np.uint32)]
segs = [ctx.encode(o, v) for o, v in zip(offs, vals)]
for rep in range(2):
    out_off, out_vals, st = ctx.merge(segs)
    print("off", out_off.download(), "vals", out_vals.download(4), "stats", st.n_in, st.n_out, st.n_terms_out, st.n_tiles)
# T=3 with payload
offs = [np.array([0, 3, 3, 5], np.uint64), np.array([0, 0, 2, 2], np.uint64)]
vals = [np.array([1, 4, 9, 2, 3], np.uint32), np.array([7, 8], np.uint32)]
segs = [ctx.encode(o, v) for o, v in zip(offs, vals)]
out_off, out_vals, st = ctx.merge(segs)
print("off", out_off.download(), "vals", out_vals.download(7), "stats", st.n_in, st.n_out, st.n_terms_out, st.n_tiles)

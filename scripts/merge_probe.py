"""GPU probe: merge workload timing + per-step cycle shares of k_merge_tiles."""
import sys, os, time, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from inverted_index_2_amd import Context, synth
T = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000
k = int(sys.argv[2]) if len(sys.argv) > 2 else 16
ctx = Context(0)
t = time.time(); offs, vals, removed = synth.merge_workload(T, k, 1000.0, 100_000_000); print("gen", time.time() - t, flush=True)
segs = [ctx.encode(o, v) for o, v in zip(offs, vals)]
tomb = ctx.tombstones(removed) if os.environ.get("NO_TOMB") != "1" else None
n_in = sum(int(o[-1]) for o in offs)
out_off = ctx.empty(T + 1, np.uint64); out_vals = ctx.empty(n_in)
_, _, st = ctx.merge(segs, tomb, out_off, out_vals)
print("n_in", n_in, "n_out", st.n_out, "tiles", st.n_tiles, "enc bytes", sum(s.info.n_bytes for s in segs), "blocks", sum(s.info.n_blocks for s in segs), flush=True)
ctx.set_option("profile.events", 1); ctx.profile_read()
t = time.time()
for _ in range(3): ctx.merge(segs, tomb, out_off, out_vals)
ctx.sync(); dt = (time.time() - t) / 3
ms, n = ctx.profile_read()
print(f"step {dt*1e3:.2f} ms  tile kernel {ms/n:.2f} ms  {n_in/dt/1e9:.2f} Gpostings/s", flush=True)
ctx.set_option("debug.stamps", 1)
ctx.merge(segs, tomb, out_off, out_vals)
nwg = 512
buf = (C.c_uint64 * (nwg * 8))()
ctx._ck(ctx.lib.ii2_debug_read(ctx.h, buf, nwg * 8))
arr = np.frombuffer(buf, dtype=np.uint64).reshape(-1, 8).astype(np.float64)
names = ["A ranges/table", "E1 bucket fold", "E2 single-term pairwise", "D gather", "E3 multi-term fold", "F filter/compact", "G leaves", "H scan + write-out"]
tot = arr.sum(axis=1).mean()
print("mean cycles per WG", tot, "tiles per WG", st.n_tiles / nwg)
for i, nm in enumerate(names[:8]):
    print(f"  {nm:18s} {arr[:, i].mean():12.0f}  {100 * arr[:, i].mean() / tot:5.1f}%")

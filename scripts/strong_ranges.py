"""GPU probe: the strong-scaling split of BASELINE configs[3] (64 segments x 1M terms) measured range by range on ONE
GPU — every rank's share is generated, merged and timed in turn, so the slowest range (what an N-GPU run waits for) is
known without an N-GPU node.  Usage: strong_ranges.py [world ...] [k=<segments>] [only=<rank>]   (default: 8 ranges, 64 segments)"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from inverted_index_2_amd import Context, synth, sharding
T, k, mean, D = 1_000_000, 64, 1000.0, 100_000_000
worlds = [int(a) for a in sys.argv[1:] if "=" not in a] or [8]
kk = [int(a.split("=")[1]) for a in sys.argv[1:] if a.startswith("k=")]          # k=<segments> (default 64)
if kk: k = kk[0]
only = [int(a.split("=")[1]) for a in sys.argv[1:] if a.startswith("only=")]      # only=<rank>: just that range (for rocprofv3)
ctx = Context(0)
ctx.set_option("profile.events", 1)
for a in sys.argv[1:]:                                  # any other name=value is a library option
    if "=" in a and not a.startswith(("k=", "only=", "stamps=")):
        ctx.set_option(a.split("=")[0], int(a.split("=")[1]))
for world in worlds:
    ranges = sharding.balanced_term_ranges(T, mean, D, world)
    times, posts = [], []
    for r, (t0, t1) in enumerate(ranges):
        if only and r not in only: continue
        offs, vals, removed = synth.merge_workload_big(T, k, mean, D, threads=16, term_range=(t0, t1))
        n_in = int(sum(int(o[-1]) for o in offs))
        segs = [ctx.encode(o, v) for o, v in zip(offs, vals)]
        tomb = ctx.tombstones(removed)
        out_off = ctx.empty(t1 - t0 + 1, np.uint64); out_vals = ctx.empty(max(n_in, 1))
        del offs, vals
        ctx.merge(segs, tomb, out_off, out_vals); ctx.sync()
        t = time.perf_counter()
        for _ in range(3): ctx.merge(segs, tomb, out_off, out_vals)
        ctx.sync(); dt = (time.perf_counter() - t) / 3
        times.append(dt * 1e3); posts.append(n_in)
        print(f"world {world} range [{t0}, {t1}) terms {t1 - t0} postings {n_in} merge {dt*1e3:.2f} ms", flush=True)
        if "stamps=1" in sys.argv:                          # cycle shares of the tile kernel's steps (see merge_probe.py)
            import ctypes as C
            ctx.set_option("debug.stamps", 1)
            _, _, st = ctx.merge(segs, tomb, out_off, out_vals)
            buf = (C.c_uint64 * (512 * 8))()
            ctx._ck(ctx.lib.ii2_debug_read(ctx.h, buf, 512 * 8))
            arr = np.frombuffer(buf, dtype=np.uint64).reshape(-1, 8).astype(np.float64)
            tot = arr.sum(axis=1).mean()
            names = ["A ranges/table", "E1 single-term fold", "E2 batch fold", "D gather/mark", "E3 pairwise fold", "F filter/compact", "G park", "-"]
            print("   tiles", st.n_tiles, " ".join(f"{nm}: {100 * arr[:, i].mean() / tot:.0f}%" for i, nm in enumerate(names[:7])), flush=True)
            ctx.set_option("debug.stamps", 0)
        for s in segs: s.free()
        tomb.free(); out_off.free(); out_vals.free()
    print(f"world {world}: slowest range {max(times):.2f} ms, sum {sum(times):.2f} ms, postings {sum(posts)} -> "
          f"{sum(posts)/max(times)/1e6:.1f} G postings/s if the ranges ran side by side", flush=True)
ctx.close()

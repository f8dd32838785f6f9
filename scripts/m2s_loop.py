"""C3 / C4-shaped merge into a DV1 segment (ii2_merge_segments_to_seg) next to the raw merge, wall clock per call, for rocprofv3.
Usage: python scripts/m2s_loop.py [terms=N] [segments=K] [steps=N] [check=0|1] [opt=value ...]"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from inverted_index_2_amd import Context, synth
T, k, steps, check = 1_000_000, 16, 5, 0
ctx = Context(0)
want_stamps = False
for kv in sys.argv[1:]:
    if kv == "stamps":
        want_stamps = True
        continue
    key, v = kv.split("=")
    if key == "terms": T = int(v)
    elif key == "segments": k = int(v)
    elif key == "steps": steps = int(v)
    elif key == "check": check = int(v)
    else: ctx.set_option(key, int(v))
offs, vals, removed = synth.merge_workload_big(T, k, 1000.0, 100_000_000, threads=min(len(os.sched_getaffinity(0)), 32))
segs = [ctx.encode(o, v) for o, v in zip(offs, vals)]
tomb = ctx.tombstones(removed)
n_in = int(sum(int(o[-1]) for o in offs))
del offs, vals
out_off = ctx.empty(T + 1, np.uint64); out_vals = ctx.empty(n_in)
_, _, st = ctx.merge(segs, tomb, out_off, out_vals)
ctx.sync()
t0 = time.perf_counter()
for _ in range(steps):
    ctx.merge(segs, tomb, out_off, out_vals)
ctx.sync()
raw_ms = (time.perf_counter() - t0) / steps * 1e3
seg, st2 = ctx.merge_to_segment(segs, tomb)
ctx.sync()
if check:
    import ctypes as C
    want_off = out_off.download(); want_vals = out_vals.download(int(st.n_out))
    got_off, got_vals = seg.decode()
    assert np.array_equal(got_off, want_off) and np.array_equal(got_vals, want_vals), "merge_to_segment differs from the raw merge"
    print("check ok", flush=True)
info = seg.info
seg.free()
t0 = time.perf_counter()
for _ in range(steps):
    s2, _ = ctx.merge_to_segment(segs, tomb)
    s2.free()
ctx.sync()
seg_ms = (time.perf_counter() - t0) / steps * 1e3
print("postings_in", n_in, "out", int(st.n_out), "raw merge ms", round(raw_ms, 3), "to segment ms", round(seg_ms, 3), "ratio", round(seg_ms / raw_ms, 3),
      "out bytes", int(info.n_bytes), "blocks", int(info.n_blocks), flush=True)
if want_stamps:          # where an encoder wave's cycles go (option debug.stamps = 3)
    import ctypes as C
    ctx.set_option("debug.stamps", 3)
    s3, _ = ctx.merge_to_segment(segs, tomb)
    ctx.sync()
    buf = (C.c_uint64 * (2048 * 8))()
    ctx._ck(ctx.lib.ii2_debug_read(ctx.h, buf, 2048 * 8))
    raw = np.frombuffer(buf, dtype=np.uint64).reshape(2048, 8).sum(axis=0)
    a = raw.astype(np.float64)
    polls, pm, a[7] = float(int(raw[7]) & 0xFFFFFFFF), float(int(raw[7]) >> 32), 0.0
    names = ["first loads", "list starts", "walk A", "other waves", "publish + walk B", "offset", "skip entries + flush", "-"]
    print("encoder cycles per phase (all waves): " + ", ".join(f"{n} {100 * x / a.sum():.1f}%" for n, x in zip(names, a) if x), flush=True)
    print("cycles per wave:", a.sum() / (-(-int(st.n_out) // 1024)), "polls per look-back:", polls / (-(-int(st.n_out) // 8192)), "of them with a member's amount missing:", pm / (-(-int(st.n_out) // 8192)), flush=True)
    s3.free()
    ctx.set_option("debug.stamps", 0)

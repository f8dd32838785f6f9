"""GPU probe: the host-buffer entry points (what a cgo binding calls with Go slices) on BASELINE configs[1] and a
200k-term slice of configs[2]: wall time INCLUDING the upload of the raw lists over PCIe, the encode, the operation and
the download — next to the device-resident figures of bench.py.  Pageable host memory, like Go slices."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from inverted_index_2_amd import Context, synth
ctx = Context(0)
D = 100_000_000
a, b = synth.zipf_list(2, D), synth.zipf_list(3, D)
want = np.intersect1d(a, b, assume_unique=True)
for rep in range(3):
    t = time.perf_counter(); got = ctx.intersect_host([a, b]); dt = time.perf_counter() - t
    assert np.array_equal(got, want)
    print(f"ii2_intersect_host C2: {dt*1e3:.1f} ms  ({(a.size+b.size)/dt/1e9:.2f} G postings/s; {4*(a.size+b.size)/1e6:.0f} MB up, {4*got.size/1e6:.0f} MB down)", flush=True)
# the C call alone, on flat buffers that already exist (what a cgo binding hands over: Go slices) - the wrapper above also
# concatenates the lists (a 333 MB host copy) and allocates and copies the result
import ctypes as C
from inverted_index_2_amd.engine import _ptr
off = np.array([0, a.size, a.size + b.size], np.uint64)
flat = np.concatenate([a, b])
out = np.zeros(min(a.size, b.size), np.uint32)
cnt = C.c_uint64()
for rep in range(4):
    t = time.perf_counter()
    ctx._ck(ctx.lib.ii2_intersect_host(ctx.h, 2, _ptr(off), _ptr(flat), None, 0, _ptr(out), out.size, C.byref(cnt)))
    dt = time.perf_counter() - t
    assert cnt.value == want.size and np.array_equal(out[: cnt.value], want)
    print(f"ii2_intersect_host C2, the C call on existing flat buffers: {dt*1e3:.1f} ms  ({(a.size+b.size)/dt/1e9:.2f} G postings/s)", flush=True)
T, k = 200_000, 16
offs, vals, removed = synth.merge_workload_big(T, k, 1000.0, D, threads=16)
n_in = sum(int(o[-1]) for o in offs)
for rep in range(3):
    t = time.perf_counter(); o, v, st = ctx.merge_host(offs, vals, removed); dt = time.perf_counter() - t
    print(f"ii2_merge_host {k} x {T} terms: {dt*1e3:.1f} ms  ({n_in/dt/1e9:.2f} G postings/s; {4*n_in/1e6:.0f} MB up, {4*int(st.n_out)/1e6:.0f} MB down)", flush=True)
# ... and the merge's C call alone on existing flat buffers
from inverted_index_2_amd._lib import MergeStats
flat_off = np.concatenate([np.asarray(o, np.uint64) for o in offs])
base = np.zeros(k + 1, np.uint64); base[1:] = np.cumsum([v.size for v in vals])
flat = np.concatenate([np.asarray(v, np.uint32) for v in vals])
rem = np.asarray(removed, np.uint32)
out_off = np.zeros(T + 1, np.uint64); out_vals = np.zeros(flat.size, np.uint32)
st = MergeStats()
for rep in range(4):
    t = time.perf_counter()
    ctx._ck(ctx.lib.ii2_merge_host(ctx.h, k, T, _ptr(flat_off), _ptr(base), _ptr(flat), _ptr(rem), rem.size, _ptr(out_off), _ptr(out_vals), out_vals.size, C.byref(st)))
    dt = time.perf_counter() - t
    print(f"ii2_merge_host {k} x {T} terms, the C call on existing flat buffers: {dt*1e3:.1f} ms  ({n_in/dt/1e9:.2f} G postings/s)", flush=True)
assert np.array_equal(out_off, o) and np.array_equal(out_vals[: int(out_off[-1])], v)
ctx.close()

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
T, k = 200_000, 16
offs, vals, removed = synth.merge_workload_big(T, k, 1000.0, D, threads=16)
n_in = sum(int(o[-1]) for o in offs)
for rep in range(3):
    t = time.perf_counter(); o, v, st = ctx.merge_host(offs, vals, removed); dt = time.perf_counter() - t
    print(f"ii2_merge_host {k} x {T} terms: {dt*1e3:.1f} ms  ({n_in/dt/1e9:.2f} G postings/s; {4*n_in/1e6:.0f} MB up, {4*int(st.n_out)/1e6:.0f} MB down)", flush=True)
ctx.close()
